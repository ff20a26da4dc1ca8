# Where do the host threads and the GPU sit?  (diagnosis for the spread of host_buffer_path across boxes)
echo "== allowed cpus"; grep -i "cpus_allowed_list\|mems_allowed_list" /proc/self/status
echo "== cpu.max"; cat /sys/fs/cgroup/cpu.max 2>/dev/null
echo "== numa nodes"; for n in /sys/devices/system/node/node*; do echo "$(basename $n): $(cat $n/cpulist)"; done
echo "== gpu pci"; for d in /sys/class/drm/card*/device; do echo "$d -> $(readlink -f $d | xargs basename) numa_node=$(cat $d/numa_node 2>/dev/null) local_cpulist=$(cat $d/local_cpulist 2>/dev/null)"; done
echo "== visible"; echo "HIP_VISIBLE_DEVICES=$HIP_VISIBLE_DEVICES ROCR_VISIBLE_DEVICES=$ROCR_VISIBLE_DEVICES"
nproc

# SQ counters of the LM kernel on a low-ray rig: usage bash tools/_r04_lm_pmc.sh <n_cam>x<n_sys> [...]
set -x
export TMPDIR=/tmp
O=gpurun_out/r4_lm_pmc
mkdir -p $O
for rig in "$@"; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${rig}_stats -- python3 tools/lm_rigs.py pmc $rig > $O/$rig.stats.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES --kernel-trace --output-format csv -d $O/${rig}_pmc1 -- python3 tools/lm_rigs.py pmc $rig > $O/$rig.pmc1.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/${rig}_pmc2 -- python3 tools/lm_rigs.py pmc $rig > $O/$rig.pmc2.log 2>&1
  n=${rig#*x}; c=${rig%x*}
  python3 tools/summarize_sq.py $O/$rig.summary.json "ray_kernel<$c," $n $O/${rig}_stats $O/${rig}_pmc1 $O/${rig}_pmc2 -- "$rig" | grep -E "avg_ns|lanes_active|insts_per_system|WAVE_CYCLES|ACTIVE_INST_ANY|WAIT_INST|SQ_WAVES|issue_slot"
done

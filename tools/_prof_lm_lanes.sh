# Lane utilisation of the LM throughput kernel against the batch size (16 cameras): how much of the idle-lane share is
# the drain (fixed per launch) and how much is steady state.  usage (GPU box): bash tools/_prof_lm_lanes.sh
set -x
export TMPDIR=/tmp
O=gpurun_out/prof_lm_lanes
mkdir -p $O
for n in 250000 1000000 4000000; do
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES --kernel-trace --output-format csv -d $O/n$n -- python3 tools/ab_knobs.py 16 $n > $O/n$n.log 2>&1 || exit 1
done
python3 - <<'PY'
import csv, glob, json, collections
out = {}
for n in (250000, 1000000, 4000000):
    f = max(glob.glob(f"gpurun_out/prof_lm_lanes/n{n}/**/*counter_collection.csv", recursive=True))
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "ray_kernel<16" in r["Kernel_Name"] and ", 0>(" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    c = {k: sum(v) / len(v) for k, v in acc.items()}
    out[n] = {"launches": len(acc["SQ_INSTS_VALU"]), "wave_valu_insts_per_system": c["SQ_INSTS_VALU"] / n,
              "lanes_active_per_valu_inst": c["SQ_THREAD_CYCLES_VALU"] / (64 * c["SQ_ACTIVE_INST_VALU"]),
              "active_lane_insts_per_system": c["SQ_THREAD_CYCLES_VALU"] / 4 / n}
json.dump(out, open("gpurun_out/prof_lm_lanes/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY

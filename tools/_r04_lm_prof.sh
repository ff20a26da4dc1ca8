# per-kernel durations of the LM launches on the low-ray rigs (rocprofv3 --kernel-trace --stats), usage: bash tools/_r04_lm_prof.sh <label> [rig ...]
set -x
export TMPDIR=/tmp
L=${1:-x}; shift
O=gpurun_out/r4_prof_$L
mkdir -p $O
for rig in ${@:-2x200000 4x1000000 6x1000000}; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$rig -- python3 tools/lm_rigs.py $L $rig > $O/$rig.log 2>&1
  f=$(find $O/$rig -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if float(r["Percentage"]) > 0.5:
        print(r["Name"][:90], "calls", r["Calls"], "avg_us", round(float(r["AverageNs"]) / 1e3, 1), "min_us", round(float(r["MinNs"]) / 1e3, 1), "pct", r["Percentage"])
PY
done
python3 tools/lm_iter_hist.py 2 200000 2>&1 | tail -15

# Round 5: L2 -> fabric stall counters of the read-only / read + store microbenchmark kernels (tools/experiments/rw_mix).
# gpurun -- 'bash tools/_r05_rw_pmc.sh'
set -x
export TMPDIR=/tmp
O=gpurun_out/rw_pmc
mkdir -p $O
timeout -k 10 200 rocprofv3 --pmc TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum --kernel-trace --output-format csv -d $O/p1 -- tools/experiments/rw_mix > $O/p1.log 2>&1 && echo p1 ok
timeout -k 10 200 rocprofv3 --pmc TCC_TAG_STALL_sum TCC_BUSY_sum TCC_IB_STALL_sum TCC_EA0_RDREQ_LEVEL_sum --kernel-trace --output-format csv -d $O/p2 -- tools/experiments/rw_mix > $O/p2.log 2>&1 && echo p2 ok
python3 - <<'PY'
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/rw_pmc/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = k[k.find("rw<"):k.find(">") + 1] if "rw<" in k else k[:40]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
json.dump(out, open("gpurun_out/rw_pmc/summary.json", "w"), indent=1)
for k in sorted(out): print(k, {c: round(v) for c, v in out[k].items()})
PY

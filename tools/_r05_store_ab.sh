# Round 5: store-policy A/B of the 8-camera matrix kernel (variants 5-7) against the default, interleaved in one process,
# then the L2->fabric write request mix (64-B vs all) of the default and of the best alternative.
# gpurun -- 'bash tools/_r05_store_ab.sh'
set -x
export TMPDIR=/tmp
O=gpurun_out/store_ab
mkdir -p $O
TDR_VARIANTS=0,5,6,7,2 timeout -k 10 300 python3 tools/ab_variants.py 8 10000000 9 > $O/ab.txt 2>&1 && cat $O/ab.txt
rocprofv3 -L > $O/counters.txt 2>&1; grep -o "TCC_EA0_WRREQ[A-Za-z0-9_]*" $O/counters.txt | sort -u | head -20
for v in 0 5 6; do
  TDR_VARIANTS=$v timeout -k 10 200 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d $O/pmc_v$v -- python3 tools/ab_variants.py 8 10000000 1 > $O/pmc_v$v.log 2>&1 && echo pmc $v ok
done
python3 - <<'PY'
import csv, glob, collections
for v in (0, 5, 6):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/store_ab/pmc_v{v}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "matrix_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(v, {k: sum(x) / len(x) for k, x in acc.items()}, {k: len(x) for k, x in acc.items()})
PY

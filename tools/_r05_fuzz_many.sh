# Round 5: the differential campaign (tests/tools/fuzz_parity.py) over many seeds on the final kernels.
# gpurun --timeout 1100 -- 'bash tools/_r05_fuzz_many.sh'
set -x
mkdir -p gpurun_out/fuzz
for seed in 7001 7002 7003 7004 7005 7006 7007 7008 7009 7010 7011 7012; do
  timeout -k 10 300 python3 tests/tools/fuzz_parity.py 600 400 $seed > gpurun_out/fuzz/seed_$seed.json 2> gpurun_out/fuzz/seed_$seed.err && echo seed $seed done $(date +%s) || { echo seed $seed FAILED; tail -5 gpurun_out/fuzz/seed_$seed.err; break; }
done

set -x
export TMPDIR=/tmp
O=gpurun_out/prof_r03_c5
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5_stats -- python3 bench.py --config c5 --steps 10 --warmup 2 --no-cpu-baseline --repeats 0 > $O/c5_stats.log 2>&1 && echo c5 stats ok
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES --kernel-trace --output-format csv -d $O/c5_pmc1 -- python3 bench.py --config c5 --steps 5 --warmup 1 --no-cpu-baseline --repeats 0 > $O/c5_pmc1.log 2>&1 && echo c5 pmc1 ok
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/c5_pmc2 -- python3 bench.py --config c5 --steps 5 --warmup 1 --no-cpu-baseline --repeats 0 > $O/c5_pmc2.log 2>&1 && echo c5 pmc2 ok
TDR_PROFILE_COMMIT=${TDR_PROFILE_COMMIT:-worktree} python tools/summarize_sq.py $O/r03_c5_summary.json "ray_kernel<16" 1000000 $O/c5_stats $O/c5_pmc1 $O/c5_pmc2 -- "C5: 16 cameras x 1M systems, LM throughput kernel (round 3)" | tail -16
cp $(find $O/c5_stats -name "*kernel_stats.csv" | head -1) $O/r03_c5_kernel_stats.csv

set -x
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/prof_r02d
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5_stats -- python3 bench.py --config c5 --steps 10 --warmup 2 --no-cpu-baseline --repeats 0 > $O/c5_stats.log 2>&1 && echo c5 stats ok
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES --kernel-trace --output-format csv -d $O/c5_pmc1 -- python3 bench.py --config c5 --steps 5 --warmup 1 --no-cpu-baseline --repeats 0 > $O/c5_pmc1.log 2>&1 && echo c5 pmc1 ok
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/c5_pmc2 -- python3 bench.py --config c5 --steps 5 --warmup 1 --no-cpu-baseline --repeats 0 > $O/c5_pmc2.log 2>&1 && echo c5 pmc2 ok
TDR_PROFILE_COMMIT=45e227d python tools/summarize_sq.py profiles/r02_c5_summary.json ray_kernel 1000000 $O/c5_stats $O/c5_pmc1 $O/c5_pmc2 -- "C5: 16 cameras x 1M systems, LM kernel (square-root-free pass: J^T J from p p^T / r^2, J^T r = sum p; origins prefetched; write-out at refill; cooperative drain after the loop)" | tail -14
cp $(find $O/c5_stats -name "*kernel_stats.csv" | head -1) $O/r02_c5_kernel_stats.csv
cp profiles/r02_c5_summary.json $O/
python bench.py --config c5 > $O/bench_c5.json 2> $O/bench_c5.err; tail -c 1200 $O/bench_c5.json

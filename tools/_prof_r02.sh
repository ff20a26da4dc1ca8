set -x
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/prof_r02
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3_stats -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --repeats 0 > $O/c3_stats.log 2>&1 && echo c3 stats ok
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/c3_fetch -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --repeats 0 > $O/c3_fetch.log 2>&1 && echo c3 fetch ok
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/c3_write -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --repeats 0 > $O/c3_write.log 2>&1 && echo c3 write ok
python tools/summarize_prof.py r02 c3 $O/c3_stats $O/c3_fetch $O/c3_write matrix_kernel 31affea | tail -5
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5_stats -- python3 bench.py --config c5 --steps 10 --warmup 2 --no-cpu-baseline --repeats 0 > $O/c5_stats.log 2>&1 && echo c5 stats ok
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES --kernel-trace --output-format csv -d $O/c5_pmc1 -- python3 bench.py --config c5 --steps 5 --warmup 1 --no-cpu-baseline --repeats 0 > $O/c5_pmc1.log 2>&1 && echo c5 pmc1 ok
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/c5_pmc2 -- python3 bench.py --config c5 --steps 5 --warmup 1 --no-cpu-baseline --repeats 0 > $O/c5_pmc2.log 2>&1 && echo c5 pmc2 ok
TDR_PROFILE_COMMIT=31affea python tools/summarize_sq.py profiles/r02_c5_summary.json ray_kernel 1000000 $O/c5_stats $O/c5_pmc1 $O/c5_pmc2 -- "C5: 16 cameras x 1M systems, LM kernel without the FD truncation term" | tail -12
cp $(find $O/c5_stats -name "*kernel_stats.csv" | head -1) profiles/r02_c5_kernel_stats.csv
cp profiles/r02_c3_kernel_stats.csv profiles/r02_c3_summary.json profiles/traffic.json profiles/r02_c5_summary.json profiles/r02_c5_kernel_stats.csv $O/
python bench.py > $O/bench_c3.json 2> $O/bench_c3.err; tail -c 600 $O/bench_c3.json
python bench.py --config c5 > $O/bench_c5.json 2> $O/bench_c5.err; tail -c 900 $O/bench_c5.json

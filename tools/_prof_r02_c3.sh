set -x
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/prof_r02_c3
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3_stats -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --repeats 0 > $O/c3_stats.log 2>&1 && echo c3 stats ok
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/c3_fetch -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --repeats 0 > $O/c3_fetch.log 2>&1 && echo c3 fetch ok
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/c3_write -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --repeats 0 > $O/c3_write.log 2>&1 && echo c3 write ok
python tools/summarize_prof.py r02 c3 $O/c3_stats $O/c3_fetch $O/c3_write matrix_kernel 605d408 | tail -6
grep -o '{"metric".*' $O/c3_stats.log | head -1 > $O/r02_c3_bench_line_under_rocprof.log
cp profiles/r02_c3_kernel_stats.csv profiles/r02_c3_summary.json profiles/traffic.json $O/

# Round 5, second session: the C3 profile passes again on the session's final library (kernels unchanged since c96d33f;
# the launch path gained A/B knobs and the stream timer two modes).  gpurun -- 'bash tools/_prof_r05b.sh <commit>'
set -x
export TMPDIR=/tmp
C=${1:-worktree}
O=gpurun_out/prof_r05b
mkdir -p $O
B="--no-cpu-baseline --repeats 0 --side-records off"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3_stats -- python3 bench.py --steps 200 --warmup 150 $B > $O/c3_stats.log 2>&1 && echo c3 stats ok
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/c3_fetch -- python3 bench.py --steps 20 --warmup 2 $B > $O/c3_fetch.log 2>&1 && echo c3 fetch ok
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/c3_write -- python3 bench.py --steps 20 --warmup 2 $B > $O/c3_write.log 2>&1 && echo c3 write ok
cp profiles/traffic.json $O/traffic_before.json
python tools/summarize_prof.py r05b c3 $O/c3_stats $O/c3_fetch $O/c3_write "matrix_kernel<8" $C | tail -12
grep -o '{"metric".*' $O/c3_stats.log | head -1 > profiles/r05b_c3_bench_line_under_rocprof.log
mkdir -p gpurun_out/profiles_r05b && cp profiles/r05b_* profiles/traffic.json gpurun_out/profiles_r05b/

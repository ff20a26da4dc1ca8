# Round-4 profile passes (run on the GPU box from the repo root: gpurun -- 'bash tools/_prof_r04.sh <commit>').
# Separate rocprofv3 runs: --kernel-trace --stats for durations, --pmc passes for counters (never mixed with traces).
set -x
export TMPDIR=/tmp
C=${1:-worktree}
O=gpurun_out/prof_r04
mkdir -p $O
B="--no-cpu-baseline --repeats 0 --side-records off"
# ---- C3 (default bench line): kernel stats, the bench line under rocprof, HBM traffic
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3_stats -- python3 bench.py --steps 100 --warmup 10 $B > $O/c3_stats.log 2>&1 && echo c3 stats ok
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/c3_fetch -- python3 bench.py --steps 20 --warmup 2 $B > $O/c3_fetch.log 2>&1 && echo c3 fetch ok
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/c3_write -- python3 bench.py --steps 20 --warmup 2 $B > $O/c3_write.log 2>&1 && echo c3 write ok
python tools/summarize_prof.py r04 c3 $O/c3_stats $O/c3_fetch $O/c3_write matrix_kernel $C | tail -8
grep -o '{"metric".*' $O/c3_stats.log | head -1 > profiles/r04_c3_bench_line_under_rocprof.log
# ---- C4 (8 cameras x 4 drones x 10 M frames = 40 M systems, the same kernel on a 6.08 GB footprint; since round 4 a
# step is 4 launches on slices of ~10 Mi systems: per-step figures = sums over the 4)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4_stats -- python3 bench.py --config c4 --steps 50 --warmup 10 $B > $O/c4_stats.log 2>&1 && echo c4 stats ok
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/c4_fetch -- python3 bench.py --config c4 --steps 10 --warmup 2 $B > $O/c4_fetch.log 2>&1 && echo c4 fetch ok
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/c4_write -- python3 bench.py --config c4 --steps 10 --warmup 2 $B > $O/c4_write.log 2>&1 && echo c4 write ok
python tools/summarize_prof.py r04 c4 $O/c4_stats $O/c4_fetch $O/c4_write matrix_kernel $C 4 | tail -12
grep -o '{"metric".*' $O/c4_stats.log | head -1 > profiles/r04_c4_bench_line_under_rocprof.log
# ---- C5 (LM kernel): kernel stats + SQ counters in two passes
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5_stats -- python3 bench.py --config c5 --steps 10 --warmup 2 --no-cpu-baseline --repeats 0 > $O/c5_stats.log 2>&1 && echo c5 stats ok
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES --kernel-trace --output-format csv -d $O/c5_pmc1 -- python3 bench.py --config c5 --steps 5 --warmup 1 --no-cpu-baseline --repeats 0 > $O/c5_pmc1.log 2>&1 && echo c5 pmc1 ok
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/c5_pmc2 -- python3 bench.py --config c5 --steps 5 --warmup 1 --no-cpu-baseline --repeats 0 > $O/c5_pmc2.log 2>&1 && echo c5 pmc2 ok
TDR_PROFILE_COMMIT=$C python tools/summarize_sq.py profiles/r04_c5_summary.json "ray_kernel<16" 1000000 $O/c5_stats $O/c5_pmc1 $O/c5_pmc2 -- "C5: 16 cameras x 1M systems, LM kernel ray_kernel<16, CamR<16>, true, true, 0> (one launch; rank-deficient solves are closed forms inside the kernel since round 4)" | tail -12
cp $(find $O/c5_stats -name "*kernel_stats.csv" | head -1) profiles/r04_c5_kernel_stats.csv
# ---- bench lines (un-profiled)
python bench.py > profiles/r04_bench_default.json 2> $O/bench_default.err; tail -c 300 profiles/r04_bench_default.json
python bench.py --config c5 > profiles/r04_bench_c5.json 2> $O/bench_c5.err; tail -c 200 profiles/r04_bench_c5.json
mkdir -p gpurun_out/profiles_r04 && cp profiles/r04_* profiles/traffic.json gpurun_out/profiles_r04/

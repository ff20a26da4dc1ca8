# Round-5 profile passes (run on the GPU box from the repo root: gpurun -- 'bash tools/_prof_r05.sh <commit>').
# Separate rocprofv3 runs: --kernel-trace --stats for durations, --pmc passes for counters (never mixed with traces).
# Every stats pass warms the clock up for >= 25 ms INSIDE the profiled command (the round-4 c3i16
# profile averaged 110 short launches still on the ramp and sat 13 % above the bench figure).
set -x
export TMPDIR=/tmp
C=${1:-worktree}
O=gpurun_out/prof_r05
mkdir -p $O
B="--no-cpu-baseline --repeats 0 --side-records off"
run_stats() { timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$1_stats -- python3 bench.py $2 $B > $O/$1_stats.log 2>&1 && echo $1 stats ok; }
run_pmc() { timeout -k 10 300 rocprofv3 --pmc $3 --kernel-trace --output-format csv -d $O/$1_$2 -- python3 bench.py $4 $B > $O/$1_$2.log 2>&1 && echo $1 $2 ok; }
# ---- C3 (default bench line)
run_stats c3 "--steps 200 --warmup 150"
run_pmc c3 fetch FETCH_SIZE "--steps 20 --warmup 2"
run_pmc c3 write WRITE_SIZE "--steps 20 --warmup 2"
python tools/summarize_prof.py r05 c3 $O/c3_stats $O/c3_fetch $O/c3_write matrix_kernel $C | tail -8
grep -o '{"metric".*' $O/c3_stats.log | head -1 > profiles/r05_c3_bench_line_under_rocprof.log
# ---- C4 (40 M systems = 4 launches on slices per step)
run_stats c4 "--config c4 --steps 50 --warmup 30"
run_pmc c4 fetch FETCH_SIZE "--config c4 --steps 10 --warmup 2"
run_pmc c4 write WRITE_SIZE "--config c4 --steps 10 --warmup 2"
python tools/summarize_prof.py r05 c4 $O/c4_stats $O/c4_fetch $O/c4_write matrix_kernel $C 4 | tail -12
grep -o '{"metric".*' $O/c4_stats.log | head -1 > profiles/r05_c4_bench_line_under_rocprof.log
# ---- C5 in closed form (new row): 16 cameras x 1 M systems, ray_closest_point_kernel
run_stats c5cf "--config c5cf --steps 400 --warmup 600"
run_pmc c5cf fetch FETCH_SIZE "--config c5cf --steps 20 --warmup 2"
run_pmc c5cf write WRITE_SIZE "--config c5cf --steps 20 --warmup 2"
python tools/summarize_prof.py r05 c5cf $O/c5cf_stats $O/c5cf_fetch $O/c5cf_write ray_closest_point_kernel $C | tail -8
grep -o '{"metric".*' $O/c5cf_stats.log | head -1 > profiles/r05_c5cf_bench_line_under_rocprof.log
run_pmc c5cf sq1 "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES" "--config c5cf --steps 20 --warmup 2"
run_pmc c5cf sq2 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SMEM" "--config c5cf --steps 20 --warmup 2"
TDR_PROFILE_COMMIT=$C python tools/summarize_sq.py profiles/r05_c5cf_sq_summary.json "ray_closest_point_kernel" 1000000 $O/c5cf_stats $O/c5cf_sq1 $O/c5cf_sq2 -- "C5 stream in closed form: 16 cameras x 1M systems, ray_closest_point_kernel<16, false, true, 64> (camera constants through scalar loads, no LDS)" | tail -12
# ---- C5 (LM kernel): kernel stats + SQ counters in two passes
run_stats c5 "--config c5 --steps 20 --warmup 10"
run_pmc c5 pmc1 "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES" "--config c5 --steps 5 --warmup 1"
run_pmc c5 pmc2 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "--config c5 --steps 5 --warmup 1"
TDR_PROFILE_COMMIT=$C python tools/summarize_sq.py profiles/r05_c5_summary.json "ray_kernel<16" 1000000 $O/c5_stats $O/c5_pmc1 $O/c5_pmc2 -- "C5: 16 cameras x 1M systems, LM kernel ray_kernel<16, CamR<16>, true, true, 0, false, 0> (one launch)" | tail -12
cp $(find $O/c5_stats -name "*kernel_stats.csv" | head -1) profiles/r05_c5_kernel_stats.csv
# ---- int16 format row, warmed up
run_stats c3i16 "--config c3i16 --steps 300 --warmup 400"
run_pmc c3i16 pmc1 "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES" "--config c3i16 --steps 10 --warmup 2"
run_pmc c3i16 pmc2 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "--config c3i16 --steps 10 --warmup 2"
run_pmc c3i16 fetch FETCH_SIZE "--config c3i16 --steps 10 --warmup 2"
run_pmc c3i16 write WRITE_SIZE "--config c3i16 --steps 10 --warmup 2"
TDR_PROFILE_COMMIT=$C python tools/summarize_sq.py profiles/r05_c3i16_summary.json "matrix_i16xk_kernel" 10000000 $O/c3i16_stats $O/c3i16_pmc1 $O/c3i16_pmc2 $O/c3i16_fetch $O/c3i16_write -- "c3i16: 8 cameras x 10 M systems, int16 observation format, two systems per lane; 700 launches in the stats pass, the first 400 (56 ms) as warm-up inside the profiled command" | tail -30
cp $(find $O/c3i16_stats -name "*kernel_stats.csv" | head -1) profiles/r05_c3i16_kernel_stats.csv
grep -o '{"metric".*' $O/c3i16_stats.log | head -1 > profiles/r05_c3i16_bench_line_under_rocprof.log
# ---- bench lines (un-profiled)
python bench.py > profiles/r05_bench_default.json 2> $O/bench_default.err; tail -c 300 profiles/r05_bench_default.json
python bench.py --config c5 > profiles/r05_bench_c5.json 2> $O/bench_c5.err; tail -c 200 profiles/r05_bench_c5.json
python bench.py --config c5cf > profiles/r05_bench_c5cf.json 2> $O/bench_c5cf.err; tail -c 200 profiles/r05_bench_c5cf.json
mkdir -p gpurun_out/profiles_r05 && cp profiles/r05_* profiles/traffic.json gpurun_out/profiles_r05/

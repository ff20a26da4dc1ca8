# Counter passes of the int16-format matrix kernel (bench config c3i16) -> profiles/r04_c3i16_summary.json (round 4: reciprocal-seed pivots in solve_spd3).
# usage (GPU box, repo root): bash tools/_prof_i16.sh <commit>      [TDR_I16_X2=0 in the environment: one system per lane]
set -x
export TMPDIR=/tmp
C=${1:-worktree}
O=gpurun_out/prof_i16
mkdir -p $O
B="python3 bench.py --config c3i16 --no-cpu-baseline --repeats 0"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B --steps 100 --warmup 10 > $O/stats.log 2>&1 && echo stats ok
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES --kernel-trace --output-format csv -d $O/pmc1 -- $B --steps 10 --warmup 2 > $O/pmc1.log 2>&1 && echo pmc1 ok
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc2 -- $B --steps 10 --warmup 2 > $O/pmc2.log 2>&1 && echo pmc2 ok
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- $B --steps 10 --warmup 2 > $O/fetch.log 2>&1 && echo fetch ok
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- $B --steps 10 --warmup 2 > $O/write.log 2>&1 && echo write ok
TDR_PROFILE_COMMIT=$C python tools/summarize_sq.py profiles/r04_c3i16_summary.json "matrix_i16xk_kernel" 10000000 $O/stats $O/pmc1 $O/pmc2 $O/fetch $O/write -- "c3i16: 8 cameras x 10 M systems, int16 observation format, two systems per lane, results staged through LDS; FETCH_SIZE / WRITE_SIZE in KiB as reported (FETCH_SIZE x 2 on gfx950: 320 MB read + 240 MB written = the algorithmic 56 B per system)" | tail -40
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) profiles/r04_c3i16_kernel_stats.csv
mkdir -p gpurun_out/profiles_r04 && cp profiles/r04_c3i16_* gpurun_out/profiles_r04/

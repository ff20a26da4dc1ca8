# A/B script of the two-walk queue experiment: needs tools/experiments/lm_two_walk_queue.patch applied and
# the library rebuilt — the knob TDR_LM_LPT exists only in that patch.  Output kept as profiles/r04_ab_lm_two_walk.log.
set -x
export TMPDIR=/tmp
O=gpurun_out/r4_lpt
mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py -x -q -k "ray or cascade or eigen or lm" > $O/ray_tests.log 2>&1; rc=$?; echo "ray tests rc $rc"; tail -5 $O/ray_tests.log
[ $rc -eq 0 ] || exit $rc
for rig in "3 500000" "4 1000000" "6 1000000" "8 1000000" "4 250000" "6 4000000"; do
  timeout -k 10 300 python tools/ab_knobs.py $rig "TDR_LM_LPT=0" "" "TDR_LM_LPT=2" "TDR_LM_LPT=4" 2>&1 | tee -a $O/ab_lpt.log
done
timeout -k 10 300 python tools/lm_rigs.py lpt > $O/lm_rigs.jsonl 2> $O/lm_rigs.err; cat $O/lm_rigs.jsonl
timeout -k 10 300 python tools/time_ray_score.py 4 4 2000 1 > $O/ray_score.json 2>&1; cat $O/ray_score.json

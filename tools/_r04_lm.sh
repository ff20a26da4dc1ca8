# Round-4 LM iteration: the ray tests, then the rigs of VERDICT r3 item 2 and the ray-route scoring.  usage: bash tools/_r04_lm.sh <label>
set -x
export TMPDIR=/tmp
L=${1:-after}
O=gpurun_out/r4_$L
mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py -x -q -k "ray or rank2 or eigen or lm" > $O/ray_tests.log 2>&1; rc=$?; echo "ray tests rc $rc"; tail -5 $O/ray_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/lm_rigs.py $L > $O/lm_rigs.jsonl 2> $O/lm_rigs.err; cat $O/lm_rigs.jsonl
timeout -k 10 300 python tools/time_ray_score.py 4 4 2000 1 > $O/ray_score.json 2>&1; cat $O/ray_score.json

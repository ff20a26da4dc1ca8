# Round-4 first GPU call: the new multi-GPU tests, the "before" timings of the LM rigs and of the ray-route scoring,
# and one default bench line (C4 with its same-footprint stream calibration).
set -x
export TMPDIR=/tmp
O=gpurun_out/r4a
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_sharding_nccl.py tests/test_gpu_bench_rehearsal.py -x -q > $O/tests_multi.log 2>&1; echo "tests rc $?"; tail -5 $O/tests_multi.log
timeout -k 10 300 python tools/lm_rigs.py before > $O/lm_rigs_before.jsonl 2> $O/lm_rigs_before.err; cat $O/lm_rigs_before.jsonl
timeout -k 10 300 python tools/time_ray_score.py 4 4 2000 1 > $O/ray_score_before.json 2>&1; cat $O/ray_score_before.json
timeout -k 10 300 python tests/tools/bench_classifier.py 4 4 2000 100 ray > $O/classifier_ray_before.json 2>&1; tail -3 $O/classifier_ray_before.json
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4a/bench_default.json").read().strip().splitlines()[-1])
print({k:d[k] for k in d if k.startswith("c4_") or k.startswith("c5_")}, d["value"], d["roofline"]["frac"])
print(json.dumps(d["other_configs"]["c4"]["roofline"], indent=1))
PY

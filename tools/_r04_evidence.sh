# Round-4 evidence runs on the final build (GPU box, repo root): the differential fuzz campaign, the direction
# diagnostic of the LM status words on other scenes, the ray-route combination scoring, the low-ray rigs.
set -x
export TMPDIR=/tmp
O=gpurun_out/r4_evidence
mkdir -p $O
timeout -k 10 280 python tools/lm_rigs.py final > $O/lm_rigs.jsonl 2> $O/lm_rigs.err; cat $O/lm_rigs.jsonl | cut -c1-160
for w in "4 4 2000 1" "4 4 2000 0" "8 4 300 1"; do timeout -k 10 200 python tools/time_ray_score.py $w >> $O/tree_sweep.jsonl 2>> $O/tree_sweep.err; done; cat $O/tree_sweep.jsonl
timeout -k 10 200 python tests/tools/bench_classifier.py 4 4 2000 100 ray > $O/classifier_ray.json 2>&1; tail -1 $O/classifier_ray.json | cut -c1-400
timeout -k 10 200 python tools/c4_probe.py > $O/c4_probe.json 2> $O/c4_probe.err; tail -6 $O/c4_probe.err | cut -c1-330
for off in 0 1000 2000; do timeout -k 10 400 python tests/tools/lm_direction_diag.py 1.0 $off > $O/lm_direction_$off.json 2> $O/lm_direction_$off.err; echo "direction $off rc $?"; done
for seed in 2026 404; do timeout -k 10 500 python tests/tools/fuzz_parity.py 1000 1500 $seed > $O/fuzz_$seed.json 2> $O/fuzz_$seed.err; echo "fuzz $seed rc $?"; tail -c 600 $O/fuzz_$seed.json; done

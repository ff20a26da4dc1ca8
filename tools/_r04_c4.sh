set -x
export TMPDIR=/tmp
O=gpurun_out/r4_c4
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -x -q -k "full_size or large_matrix or chunk_boundary or matrix" > $O/tests.log 2>&1; echo "tests rc $?"; tail -4 $O/tests.log
timeout -k 10 200 python tools/c4_probe.py > $O/c4_probe.json 2> $O/c4_probe.err; tail -3 $O/c4_probe.err | cut -c1-330
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench.err; python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r4_c4/bench_default.json").read().strip().splitlines()[-1])
print({k:d[k] for k in d if k.startswith("c4_") or k.startswith("c5_")}, d["value"], d["roofline"]["frac"], d["roofline"]["kernel_ms"])
print(d["other_configs"]["c4"].get("launches_per_step"), d["other_configs"]["c4"]["roofline"])
PY

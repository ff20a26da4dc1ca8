"""build() where it runs: every GPU record so far executed binaries that were
cross-compiled in the build container and pushed to the box.  Here the box compiles libtdr_amd.so itself — the
Makefile's own rule, forced (`make -B`), output redirected into a temp directory so the in-tree library stays what
it was — and a FRESH python process loads that file through the package's loader override (TDR_LIB_PATH) and runs
one matrix and one ray parity case against the CPU oracle plus a bit-for-bit comparison with the shipped library's
results.  hipcc needs about two and a half minutes for the kernel file."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = "3d-reconstruction-triangulation_amd"
pytestmark = pytest.mark.gpu

CHILD = r"""
import importlib, json, os, sys
import numpy as np
root = sys.argv[1]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "oracle"))
tdr = importlib.import_module("3d-reconstruction-triangulation_amd")
sc = importlib.import_module("3d-reconstruction-triangulation_amd.scenes")
import tdr_oracle as oracle
assert os.path.realpath(tdr.LIB_PATH) == os.path.realpath(os.environ["TDR_LIB_PATH"]), tdr.LIB_PATH
tdr.lib()
mapped = [l.split()[-1] for l in open("/proc/self/maps") if "libtdr_amd.so" in l]
cp = sc.ring_camera_params(8, seed=11)
cams = [tdr.createCamera(i, c["width"], c["height"], c["focal"], c["tvec"], c["rquat"]) for i, c in enumerate(cp)]
P, aux = tdr.pack_cameras(cams)
eng = tdr.Engine(P, aux, device=0)
X = sc.smooth_path(20000, seed=11)
pts = sc.make_detections(sc.project_P(P, X), seed=11)
xyz = eng.matrix_triangulate_points(pts)
rc, ref, _, _ = oracle.matrix_triangulate_points(P, pts)
m_rel = sc.rel_err(xyz, ref)
ptsr = sc.make_detections(sc.project_ray_model(aux, X), seed=12, dropout_p=0.3, min_valid=2)
g = eng.ray_triangulate_points_ex(ptsr)
o = oracle.ray_triangulate_points_ex(aux, ptsr, n_threads=os.cpu_count() or 1)
ok = oracle.lm_comparable(g["status"], o["status"])
r_rel = sc.rel_err(g["xyz"][ok], o["xyz"][ok])
np.save(sys.argv[2], np.concatenate([xyz.ravel(), g["xyz"].ravel(), g["status"].astype(np.float64)]))
print(json.dumps({"mapped": sorted(set(mapped)), "matrix_rel": m_rel, "ray_rel": r_rel, "ray_compared": int(ok.sum()),
                  "rc": int(rc)}))
"""


def _run_child(lib_path, out_npy):
    env = dict(os.environ, TDR_LIB_PATH=lib_path)
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT, out_npy], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_library_compiled_on_this_box_runs_and_matches(tmp_path):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip(f"no {hipcc} on this box: nothing to compile with (the shipped library is what the other tests run)")
    import time
    built = str(tmp_path / "libtdr_amd.so")
    t0 = time.time()
    r = subprocess.run(["make", "-C", os.path.join(ROOT, PKG), "-B", "libtdr_amd.so", f"LIBOUT={built}",
                        f"RESOUT={tmp_path / 'resource_usage.txt'}"], capture_output=True, text=True, timeout=1500)
    compile_s = time.time() - t0
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert os.path.getsize(built) > 1 << 20
    here = _run_child(built, str(tmp_path / "built.npy"))
    assert here["mapped"] == [os.path.realpath(built)], here["mapped"]  # the process ran the file compiled here
    assert here["rc"] == 0 and here["matrix_rel"] <= 1e-9 and here["ray_rel"] <= 1e-7 and here["ray_compared"] > 15000
    shipped = _run_child(os.path.join(ROOT, PKG, "libtdr_amd.so"), str(tmp_path / "shipped.npy"))
    assert shipped["mapped"] == [os.path.realpath(os.path.join(ROOT, PKG, "libtdr_amd.so"))]
    # same sources, same compiler, same flags: the two binaries compute the same bits
    np.testing.assert_array_equal(np.load(tmp_path / "built.npy"), np.load(tmp_path / "shipped.npy"))
    # the run's own record (gpurun_out/ travels back from the box; a copy is committed under profiles/ per round)
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "build_on_box.json"), "w") as f:
            json.dump({"test": "tests/test_gpu_build_on_box.py", "hipcc": hipcc, "compile_seconds": round(compile_s, 1),
                       "built_bytes": os.path.getsize(built), "built_on_box": here, "shipped": shipped,
                       "bit_identical": True}, f, indent=1)
    except OSError:
        pass

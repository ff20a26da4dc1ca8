"""Why does C4 (8 cameras x 40 M systems, 6.08 GB) run the matrix kernel a few per cent below C3 (10 M systems,
1.52 GB)?    One process, interleaved: the kernel and the compute-free read stream of the same
shape at 10 / 20 / 40 M systems and 1 / 4 drones (the synthetic stream's interleave), every kernel variant at 40 M,
and the 40 M batch cut into 4 launches of 10 M.  One JSON document on stdout."""
import importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tdr = importlib.import_module("3d-reconstruction-triangulation_amd")
sc = importlib.import_module("3d-reconstruction-triangulation_amd.scenes")
n_cam, n_max = 8, 40_000_000
cp = sc.ring_camera_params(n_cam, seed=4)
cams = [tdr.createCamera(i, c["width"], c["height"], c["focal"], c["tvec"], c["rquat"]) for i, c in enumerate(cp)]
P, aux = tdr.pack_cameras(cams)
eng = tdr.Engine(P, aux)
d_pts = eng.alloc(n_cam * n_max * 16); d_xyz = eng.alloc(n_max * 24); d_sink = eng.alloc(n_max * 8)
out = {"what": "matrix kernel vs footprint (8 cameras), one process", "rows": []}
med = lambda v: float(np.median(v))
for n_drones in (1, 4):
    eng.synth_device(0xC0FFEE + 4, n_drones, 0, n_max, n_max, d_pts.ptr)
    eng.synchronize()
    for n in (10_000_000, 20_000_000, 40_000_000):
        it = max(20, 800_000_000 // n)  # ~20 ms per group: long enough for the clock to sit where it will stay
        eng.time_launches(0, d_pts.ptr, n_cam, n_max, n, d_xyz.ptr, None, warmup=0, iters=2 * it)  # clock settles
        k, r = [], []
        for _ in range(5):  # kernel and stream alternate: whatever drifts, drifts for both
            k.append(eng.time_launches(0, d_pts.ptr, n_cam, n_max, n, d_xyz.ptr, None, warmup=0, iters=it))
            r.append(eng.time_stream(0, n_cam, d_pts.ptr, d_sink.ptr, n, warmup=2, iters=it))
        kb, rb = n * (16 * n_cam + 24), n * (16 * n_cam + 8)
        out["rows"].append({"n_drones": n_drones, "n_sys": n, "cam_stride": n_max, "kernel_ms": med(k), "kernel_TBps": kb / med(k) / 1e9,
                            "hbm_frac": kb / med(k) / 1e9 / 8.0, "stream_ms": med(r), "stream_TBps": rb / med(r) / 1e9,
                            "kernel_over_stream_rate": (kb / med(k)) / (rb / med(r))})
        print(out["rows"][-1], file=sys.stderr, flush=True)
# every variant at 40 M (4 drones)
var = {}
for rep in range(3):
    for v in (0, 1, 2, 3, 4):
        eng.set_matrix_variant(v)
        var.setdefault(v, []).append(eng.time_launches(0, d_pts.ptr, n_cam, n_max, n_max, d_xyz.ptr, None, warmup=1, iters=8))
eng.set_matrix_variant(0)
out["variants_at_40M_ms"] = {str(v): med(ms) for v, ms in var.items()}
# 40 M as four launches of 10 M on slices of the same arrays (cam_stride = 40 M): is it the grid size or the footprint?
import time
def four():
    for q in range(4):
        eng.matrix_device(d_pts.ptr + q * 10_000_000 * 16, n_cam, n_max, 10_000_000, d_xyz.ptr + q * 10_000_000 * 24)
for _ in range(3):
    four()
eng.synchronize()
ts = []
for _ in range(5):
    t0 = time.perf_counter()
    for _ in range(5):
        four()
    eng.synchronize()
    ts.append((time.perf_counter() - t0) / 5 * 1e3)
out["four_launches_of_10M_ms"] = med(ts)
print(json.dumps(out))

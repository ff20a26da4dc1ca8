"""C4 shape (8 cameras x 40 M systems): one launch against the same batch cut into k launches on slices of the same
arrays (cam_stride = 40 M), interleaved in one process, HIP-event timed on the launch stream via time_launches for the
single launch and wall clock + stream sync for the sliced forms (both forms also timed by wall clock)."""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tdr = importlib.import_module("3d-reconstruction-triangulation_amd")
sc = importlib.import_module("3d-reconstruction-triangulation_amd.scenes")
n_cam, n = 8, 40_000_000
cp = sc.ring_camera_params(n_cam, seed=4)
cams = [tdr.createCamera(i, c["width"], c["height"], c["focal"], c["tvec"], c["rquat"]) for i, c in enumerate(cp)]
P, aux = tdr.pack_cameras(cams)
eng = tdr.Engine(P, aux)
d_pts = eng.alloc(n_cam * n * 16); d_xyz = eng.alloc(n * 24)
eng.synth_device(0xC0FFEE + 4, 4, 0, n, n, d_pts.ptr)
eng.synchronize()
def run(k, reps=20):
    step = n // k
    # Since round 4 the library ITSELF cuts a batch beyond 10 Mi systems into launches under the default variant, so the
    # k = 1 and k = 2 cases would silently measure four launches.  Variant 3 is the same streaming configuration
    # (nontemporal, 64-thread blocks) without the slicing: a call is exactly one launch.
    eng.set_matrix_variant(3)
    t0 = time.perf_counter()
    for _ in range(reps):
        for q in range(k):
            eng.matrix_device(d_pts.ptr + q * step * 16, n_cam, n, step, d_xyz.ptr + q * step * 24)
    eng.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
for k in (1, 2, 4, 8):
    run(k, 5)
res = {}
for rep in range(6):
    for k in (1, 2, 4, 8, 16):
        res.setdefault(k, []).append(run(k))
print(json.dumps({"what": "8 cameras x 40 M systems, ms per batch, wall clock over 20 batches, median of 6 interleaved repeats",
                  "ms_by_number_of_launches": {str(k): float(np.median(v)) for k, v in res.items()},
                  "min": {str(k): float(np.min(v)) for k, v in res.items()}}))

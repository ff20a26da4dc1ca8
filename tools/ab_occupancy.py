"""Occupancy sweep of the streaming matrix kernel at C3's shape without touching its code: knob TDR_MATRIX_PAD_LDS adds
unused dynamic LDS per 64-thread block, which caps the blocks (= waves) resident on a CU (160 KB of LDS per CU; the
kernel itself holds 768 B + its registers allow 24 waves per CU).  Interleaved in one process.
usage: python tools/ab_occupancy.py [n_cam] [n_sys] [rounds]"""
import importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tdr = importlib.import_module("3d-reconstruction-triangulation_amd")
sc = importlib.import_module("3d-reconstruction-triangulation_amd.scenes")
n_cam = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 7
cp = sc.ring_camera_params(n_cam, seed=3)
cams = [tdr.createCamera(i, c["width"], c["height"], c["focal"], c["tvec"], c["rquat"]) for i, c in enumerate(cp)]
P, aux = tdr.pack_cameras(cams)
eng = tdr.Engine(P, aux)
d_pts = eng.alloc(n_cam * n * 16)
d_xyz = eng.alloc(n * 24)
eng.synth_device(0xC0FFEE + 3, 1, 0, n, n, d_pts.ptr)
eng.synchronize()
LDS_CU = 160 * 1024
res = {}
waves = [24, 20, 16, 12, 10, 8, 6, 4]  # resident waves per CU wanted
for r in range(rounds):
    for w in waves:
        pad = 0 if w >= 24 else max(0, LDS_CU // w - 768 - 256)  # floor(LDS / per-block) == w
        tdr.set_knob("TDR_MATRIX_PAD_LDS", pad)
        ms = eng.time_launches(0, d_pts.ptr, n_cam, n, n, d_xyz.ptr, None, warmup=2, iters=20)
        res.setdefault(w, []).append(ms)
tdr.set_knob("TDR_MATRIX_PAD_LDS", None)
b = n_cam * 16 + 24
print(json.dumps({"what": f"{n_cam} cameras x {n} systems, matrix kernel, ms per launch by resident waves per CU "
                          "(cap through unused dynamic LDS), median / min of %d interleaved rounds x 20 launches" % rounds,
                  "ms_median": {str(w): float(np.median(v)) for w, v in res.items()},
                  "ms_min": {str(w): float(np.min(v)) for w, v in res.items()},
                  "hbm_frac_at_median": {str(w): n * b / (float(np.median(v)) * 1e-3) / 8e12 for w, v in res.items()}}))
assert eng.fetch_first_bad() == -1

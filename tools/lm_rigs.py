"""LM kernel (device-resident entry point) on the rigs VERDICT r3 item 2 names: 1 M systems with 5 % drop-outs at 4 / 6
/ 8 / 12 cameras, 2 cameras x 200 k (every system through the eigen solve), 16 cameras x 1 M (C5 shape).  One JSON
object per line; `sha` is a digest of the result bits (two builds that agree bit for bit print the same digest).
usage: python tools/lm_rigs.py [label] [rig ...]     rig = <n_cam>x<n_sys>, default: the list above"""
import hashlib, importlib, json, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tdr = importlib.import_module("3d-reconstruction-triangulation_amd")
sc = importlib.import_module("3d-reconstruction-triangulation_amd.scenes")
label = sys.argv[1] if len(sys.argv) > 1 else "worktree"
rigs = sys.argv[2:] or ["2x200000", "3x500000", "4x1000000", "6x1000000", "8x1000000", "12x1000000", "16x1000000"]
for rig in rigs:
    n_cam, n = (int(v) for v in rig.split("x"))
    cp = sc.ring_camera_params(n_cam, seed=5)
    cams = [tdr.createCamera(i, c["width"], c["height"], c["focal"], c["tvec"], c["rquat"]) for i, c in enumerate(cp)]
    P, aux = tdr.pack_cameras(cams)
    eng = tdr.Engine(P, aux)
    d_pts = eng.alloc(n_cam * n * 16)
    d_xyz = eng.alloc(n * 24)
    eng.synth_device(0xC0FFEE + 5, 1, 0, n, n, d_pts.ptr, ray_model=True)
    eng.synchronize()
    eng.time_launches(1, d_pts.ptr, n_cam, n, n, d_xyz.ptr, None, warmup=0, iters=4)
    ms = sorted(eng.time_launches(1, d_pts.ptr, n_cam, n, n, d_xyz.ptr, None, warmup=0, iters=6) for _ in range(5))
    x = d_xyz.download((n, 3))
    print(json.dumps({"label": label, "n_cam": n_cam, "n_sys": n, "dropout": 0.05, "ms_median": ms[2], "ms_min": ms[0],
                      "ms_max": ms[-1], "points_per_s": n / (ms[2] * 1e-3), "finite": int(np.isfinite(x).all(axis=1).sum()),
                      "sha": hashlib.sha256(x.tobytes()).hexdigest()[:16]}), flush=True)
    d_pts.free(); d_xyz.free()
    del eng

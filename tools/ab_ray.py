"""Time the LM kernel (C5 shape by default) for the current TDR_LM_REFILL; prints ms per launch.
usage: python tools/ab_ray.py [n_cam] [n_sys]"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tdr = importlib.import_module("3d-reconstruction-triangulation_amd")
sc = importlib.import_module("3d-reconstruction-triangulation_amd.scenes")

n_cam = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
cp = sc.ring_camera_params(n_cam, seed=5)
cams = [tdr.createCamera(i, c["width"], c["height"], c["focal"], c["tvec"], c["rquat"]) for i, c in enumerate(cp)]
P, aux = tdr.pack_cameras(cams)
eng = tdr.Engine(P, aux)
d_pts = eng.alloc(n_cam * n * 16)
d_xyz = eng.alloc(n * 24)
eng.synth_device(0xC0FFEE + 5, 1, 0, n, n, d_pts.ptr, ray_model=True)
eng.synchronize()
ms = [eng.time_launches(1, d_pts.ptr, n_cam, n, n, d_xyz.ptr, None, warmup=1, iters=5) for _ in range(3)]
print(f"refill={os.environ.get('TDR_LM_REFILL', 'default')} n_cam={n_cam} n={n}: ms {min(ms):.3f} "
      f"(median {np.median(ms):.3f})  {n / min(ms) / 1e3:.1f} M points/s")

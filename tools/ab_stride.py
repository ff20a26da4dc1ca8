"""Does the camera-row stride (where each row starts relative to the HBM channel interleave) matter for the
matrix kernel?  C3 shape, rows padded by various amounts.  usage: python tools/ab_stride.py"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tdr = importlib.import_module("3d-reconstruction-triangulation_amd")
sc = importlib.import_module("3d-reconstruction-triangulation_amd.scenes")
n_cam, n = 8, 10_000_000
cp = sc.ring_camera_params(n_cam, seed=3)
cams = [tdr.createCamera(i, c["width"], c["height"], c["focal"], c["tvec"], c["rquat"]) for i, c in enumerate(cp)]
P, aux = tdr.pack_cameras(cams)
eng = tdr.Engine(P, aux)
d_xyz = eng.alloc(n * 24)
for pad in (0, 16, 64, 256, 1024, 4096, 4096 + 512, 65536, 65536 + 4096 + 256, 1 << 20, 485760):
    stride = n + pad
    d_pts = eng.alloc(n_cam * stride * 16)
    eng.synth_device(0xC0FFEE + 3, 1, 0, n, stride, d_pts.ptr)
    eng.synchronize()
    ms = min(eng.time_launches(0, d_pts.ptr, n_cam, stride, n, d_xyz.ptr, None, warmup=5, iters=50) for _ in range(3))
    print(f"pad {pad:8d} systems ({pad * 16:9d} B): {ms:.4f} ms  {n * 152 / ms / 1e9:.3f} TB/s", flush=True)
    d_pts.free()

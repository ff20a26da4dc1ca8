"""Small-batch LM latency through the device entry point: camera-per-lane kernel (default for small batches)
against the lane-per-system kernel (TDR_LM_COOP_MAX=0), same process, interleaved.
usage: python tools/ab_ray_small.py"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tdr = importlib.import_module("3d-reconstruction-triangulation_amd")
sc = importlib.import_module("3d-reconstruction-triangulation_amd.scenes")
for n_cam in (4, 8, 16):
    cp = sc.ring_camera_params(n_cam, seed=5)
    cams = [tdr.createCamera(i, c["width"], c["height"], c["focal"], c["tvec"], c["rquat"]) for i, c in enumerate(cp)]
    P, aux = tdr.pack_cameras(cams)
    eng = tdr.Engine(P, aux)
    for n in (64, 500, 2000, 8000, 16000, 32000, 64000, 128000):
        d_pts = eng.alloc(n_cam * n * 16)
        d_xyz = eng.alloc(n * 24)
        eng.synth_device(0xC0FFEE + 5, 1, 0, n, n, d_pts.ptr, ray_model=True)
        eng.synchronize()
        res = {}
        for rep in range(3):
            for name, env in (("lane", "0"), ("coop", str(1 << 30))):
                tdr.set_knob("TDR_LM_COOP_MAX", int(env))
                ms = eng.time_launches(1, d_pts.ptr, n_cam, n, n, d_xyz.ptr, None, warmup=1, iters=10)
                res[name] = min(res.get(name, 1e9), ms)
        print(f"n_cam={n_cam} n={n}: lane-per-system {res['lane']*1e3:.1f} us, camera-per-lane {res['coop']*1e3:.1f} us", flush=True)
        d_pts.free(); d_xyz.free()

"""Time the LM kernel (device-resident, same process, interleaved) under several knob settings.
usage: python tools/ab_knobs.py <n_cam> <n_sys> "K1=v,K2=v" "K1=v" ...   ("" = defaults)"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tdr = importlib.import_module("3d-reconstruction-triangulation_amd")
sc = importlib.import_module("3d-reconstruction-triangulation_amd.scenes")
n_cam, n = int(sys.argv[1]), int(sys.argv[2])
settings = sys.argv[3:] or [""]
cp = sc.ring_camera_params(n_cam, seed=5)
cams = [tdr.createCamera(i, c["width"], c["height"], c["focal"], c["tvec"], c["rquat"]) for i, c in enumerate(cp)]
P, aux = tdr.pack_cameras(cams)
eng = tdr.Engine(P, aux)
d_pts = eng.alloc(n_cam * n * 16)
d_xyz = eng.alloc(n * 24)
eng.synth_device(0xC0FFEE + 5, 1, 0, n, n, d_pts.ptr, ray_model=True)
eng.synchronize()
ALL = ["TDR_LM_COOP", "TDR_LM_COOP_MAX", "TDR_LM_FULL", "TDR_LM_CHUNK", "TDR_LM_REFILL", "TDR_LM_NLDS", "TDR_LM_LITERAL_FD"]
res = {s: [] for s in settings}
ref = None
for rep in range(4):
    for s in settings:
        for k in ALL:
            tdr.set_knob(k, None)
        for kv in filter(None, s.split(",")):
            k, v = kv.split("=")
            tdr.set_knob(k, int(v))
        ms = eng.time_launches(1, d_pts.ptr, n_cam, n, n, d_xyz.ptr, None, warmup=2 if rep == 0 else 0, iters=8)
        res[s].append(ms)
        x = d_xyz.download((n, 3))
        if ref is None:
            ref = x
        elif not np.array_equal(x, ref):
            print("RESULTS DIFFER under", s, float(np.nanmax(np.abs(x - ref))))
for s in settings:
    print(f"n_cam={n_cam} n={n} [{s or 'defaults'}]: ms min {min(res[s]):.4f} median {sorted(res[s])[len(res[s]) // 2]:.4f}", flush=True)

"""For 4 / 8 / 16 cameras: the matrix kernel next to a compute-free nontemporal read stream of the same
shape (n_cam 16-B rows in, 8 B out per lane) and a plain copy, all in one process on one device."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tdr = importlib.import_module("3d-reconstruction-triangulation_amd")
sc = importlib.import_module("3d-reconstruction-triangulation_amd.scenes")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
for n_cam in (4, 8, 16):
    cp = sc.ring_camera_params(n_cam, seed=3)
    cams = [tdr.createCamera(i, c["width"], c["height"], c["focal"], c["tvec"], c["rquat"]) for i, c in enumerate(cp)]
    P, aux = tdr.pack_cameras(cams)
    eng = tdr.Engine(P, aux)
    d_pts = eng.alloc(n_cam * n * 16); d_xyz = eng.alloc(n * 24); d_sink = eng.alloc(n * 8)
    eng.synth_device(0xC0FFEE + 3, 1, 0, n, n, d_pts.ptr)
    eng.synchronize()
    k, r = [], []
    for _ in range(5):
        k.append(eng.time_launches(0, d_pts.ptr, n_cam, n, n, d_xyz.ptr, None, warmup=2, iters=20))
        r.append(eng.time_stream(0, n_cam, d_pts.ptr, d_sink.ptr, n, warmup=2, iters=20))
    kb, rb = n * (16 * n_cam + 24), n * (16 * n_cam + 8)
    print(f"{n_cam:2d} cameras: kernel {np.median(k):.4f} ms = {kb/np.median(k)/1e9:.2f} TB/s | read stream "
          f"{np.median(r):.4f} ms = {rb/np.median(r)/1e9:.2f} TB/s | kernel/stream rate = {(kb/np.median(k))/(rb/np.median(r)):.3f}")
    for b in (d_pts, d_xyz, d_sink):
        b.free()

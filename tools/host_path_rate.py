"""PCIe-inclusive rate of the host-buffer entry point (pageable host memory -> H2D -> kernel -> D2H),
C3 shape.  This is NOT the bench `value` (which has inputs resident in HBM); DESIGN.md quotes it."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tdr = importlib.import_module("3d-reconstruction-triangulation_amd")
sc = importlib.import_module("3d-reconstruction-triangulation_amd.scenes")
n_cam = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
cp = sc.ring_camera_params(n_cam, seed=3)
cams = [tdr.createCamera(i, c["width"], c["height"], c["focal"], c["tvec"], c["rquat"]) for i, c in enumerate(cp)]
P, aux = tdr.pack_cameras(cams)
shards = int(os.environ.get('TDR_HOST_SHARDS', '1'))
eng = tdr.Engine(P, aux) if shards == 1 else tdr.Engine(P, aux, devices=[0] * shards)
gen = tdr.Engine(P, aux)
d_pts = gen.alloc(n_cam * n * 16)
gen.synth_device(0xC0FFEE + 3, 1, 0, n, n, d_pts.ptr)
gen.synchronize()
pts = d_pts.download((n_cam, n, 2))
d_pts.free()
eng.matrix_triangulate_points(pts[:, :100000])  # warm the workspace path
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    xyz = eng.matrix_triangulate_points(pts)
    best = min(best, time.perf_counter() - t0)
b = n * (n_cam * 16 + 24)
print(f"shards={shards} host-buffer path {n_cam} cam x {n}: {best*1e3:.1f} ms  {n/best/1e6:.1f} M points/s  {b/best/1e9:.1f} GB/s over PCIe")

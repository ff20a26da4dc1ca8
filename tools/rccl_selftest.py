"""RCCL transport of tdr_multi_matrix_triangulate_gather in a process WITHOUT torch (librccl comes from
/opt/rocm via dlopen): one-shard context, comm init + grouped send/recv to itself."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
assert "torch" not in sys.modules
tdr = importlib.import_module("3d-reconstruction-triangulation_amd")
sc = importlib.import_module("3d-reconstruction-triangulation_amd.scenes")
cp = sc.ring_camera_params(8, seed=3)
cams = [tdr.createCamera(i, c["width"], c["height"], c["focal"], c["tvec"], c["rquat"]) for i, c in enumerate(cp)]
P, aux = tdr.pack_cameras(cams)
multi = tdr.Engine(P, aux, device=-1)
subs = multi.shard_engines()
print("shards (visible GPUs):", len(subs))
n = 100_000
G = len(subs)
bounds = [n * g // G for g in range(G + 1)]
d_pts, d_xyz, keep = [], [], []
for g, e in enumerate(subs):
    m = bounds[g + 1] - bounds[g]
    bp, bx = e.alloc(8 * m * 16), e.alloc(m * 24)
    e.synth_device(1, 1, bounds[g], m, m, bp.ptr)
    e.synchronize()
    d_pts.append(bp.ptr); d_xyz.append(bx.ptr); keep += [bp, bx]
root = subs[0].alloc(n * 24)
ns = [bounds[g + 1] - bounds[g] for g in range(G)]
fb = multi.multi_matrix_gather(d_pts, 8, ns, ns, d_xyz, root.ptr, transport=1)
out = root.download((n, 3))
assert fb == -1 and np.isfinite(out).all()
print("rccl gather ok:", out[:2].tolist())

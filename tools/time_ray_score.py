"""Time tdr_ray_score_combinations / tdr_score_combinations_levelwise on a synthetic multi-drone recording.
usage: python tools/time_ray_score.py [n_cam] [n_drones] [n_frames] [path 0|1] [error_max]"""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tdr = importlib.import_module("3d-reconstruction-triangulation_amd")
sc = importlib.import_module("3d-reconstruction-triangulation_amd.scenes")
n_cam = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n_drones = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n_frames = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
path = int(sys.argv[4]) if len(sys.argv) > 4 else 1
error_max = float(sys.argv[5]) if len(sys.argv) > 5 else (120.0 if path == 1 else 1e5)
cp = sc.ring_camera_params(n_cam, seed=77)
cams = [tdr.createCamera(i, c["width"], c["height"], c["focal"], c["tvec"], c["rquat"]) for i, c in enumerate(cp)]
P, aux = tdr.pack_cameras(cams)
eng = tdr.Engine(P, aux)
counts, dets, _ = sc.multi_drone_frames(P, n_frames, n_drones, seed=11)
eng.score_combinations_levelwise(path, counts[:4], dets[:int(counts[:4].sum())], error_max)  # warm
times = []
for _ in range(5):
    t0 = time.perf_counter()
    got = eng.score_combinations_levelwise(path, counts, dets, error_max, cap=1 << 22)
    times.append(time.perf_counter() - t0)
import hashlib
print(json.dumps({"workload": f"{n_cam} cameras x {n_drones} drones x {n_frames} frames", "path": "ray" if path else "matrix",
                  "error_max": error_max, "survivors": int(len(got)), "ms_best": min(times) * 1e3,
                  "ms_median": sorted(times)[2] * 1e3,
                  "survivor_keys_sha": hashlib.sha256(np.ascontiguousarray(np.stack([got["frame"].astype(np.int64), got["leaf"], got["n_used"].astype(np.int64)])).tobytes()).hexdigest()[:16]}))

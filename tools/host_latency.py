"""Call latency of the host-buffer entry points at the reference's own operating sizes (a few thousand
frames, BASELINE.json configs[0]/[1]): pageable numpy arrays in, numpy arrays out, medians of many calls."""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tdr = importlib.import_module("3d-reconstruction-triangulation_amd")
sc = importlib.import_module("3d-reconstruction-triangulation_amd.scenes")
out = []
for n_cam in (4, 8):
    cp = sc.ring_camera_params(n_cam, seed=3)
    cams = [tdr.createCamera(i, c["width"], c["height"], c["focal"], c["tvec"], c["rquat"]) for i, c in enumerate(cp)]
    P, aux = tdr.pack_cameras(cams)
    eng = tdr.Engine(P, aux)
    for n in (1, 64, 2000, 20000, 100000, 1000000):
        X = sc.smooth_path(n, seed=n)
        pts = np.ascontiguousarray(sc.make_detections(sc.project_P(P, X), seed=n))
        for path, fn in (("matrix", eng.matrix_triangulate_points), ("ray", eng.ray_triangulate_points)):
            if path == "ray" and n > 100000:
                continue
            fn(pts)
            reps = 200 if n <= 20000 else 30
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                fn(pts)
                ts.append(time.perf_counter() - t0)
            ts.sort()
            out.append({"path": path, "n_cam": n_cam, "n_sys": n, "median_us": round(ts[len(ts) // 2] * 1e6, 1),
                        "min_us": round(ts[0] * 1e6, 1)})
            print(out[-1], flush=True)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)

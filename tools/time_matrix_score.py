"""Matrix combination scoring: the brute-force leaf kernel against the breadth-first sweep over tree sizes (where
should tdr_matrix_score_combinations hand over?).  Synthetic multi-drone recordings; best of 5 calls each.
usage: python tools/time_matrix_score.py   -> JSON lines"""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["TDR_SCORE_LEAF_LIMIT"] = str(1 << 40)  # matrix_score_combinations = always the leaf kernel here
tdr = importlib.import_module("3d-reconstruction-triangulation_amd")
sc = importlib.import_module("3d-reconstruction-triangulation_amd.scenes")
for n_cam, n_drones, n_frames in ((4, 4, 2000), (6, 4, 1000), (8, 2, 1000), (8, 3, 1000), (8, 4, 300), (8, 4, 3000), (8, 5, 300), (8, 6, 100)):
    cp = sc.ring_camera_params(n_cam, seed=77)
    cams = [tdr.createCamera(i, c["width"], c["height"], c["focal"], c["tvec"], c["rquat"]) for i, c in enumerate(cp)]
    P, aux = tdr.pack_cameras(cams)
    eng = tdr.Engine(P, aux)
    counts, dets, _ = sc.multi_drone_frames(P, n_frames, n_drones, seed=11)
    leaves = float(np.prod(counts.astype(np.float64) + 1, axis=1).sum())
    cap = 1 << 23
    res = {}
    for name, fn in (("leaf_kernel", lambda: eng.matrix_score_combinations(counts, dets, 1e5, cap=cap)),
                     ("sweep", lambda: eng.score_combinations_levelwise(0, counts, dets, 1e5, cap=cap))):
        fn()
        t = []
        for _ in range(5):
            t0 = time.perf_counter()
            got = fn()
            t.append(time.perf_counter() - t0)
        res[name] = (min(t) * 1e3, len(got))
    assert res["leaf_kernel"][1] == res["sweep"][1]
    print(json.dumps({"workload": f"{n_cam} cameras x {n_drones} drones x {n_frames} frames", "leaves_total": leaves,
                      "leaves_per_frame": leaves / n_frames, "survivors": res["sweep"][1],
                      "leaf_kernel_ms": round(res["leaf_kernel"][0], 2), "sweep_ms": round(res["sweep"][0], 2)}), flush=True)

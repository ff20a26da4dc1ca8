"""A whole recording through the product's CLI (tdr_main: XCP ingest, CSV ingest, DroneClassifier, triangulation, PLY
output) against its ground truth.  The recording directory has the layout tests/golden/make_s02_d4_a.py writes
(cameras.xcp, dl_data/, truth_mm.csv).  Prints one JSON line.
usage: python tools/run_recording.py <dir> <n_drones> [matrix|ray] [repeats]"""
import json
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAIN = os.path.join(ROOT, "3d-reconstruction-triangulation_amd", "tdr_main")
rec, n_drones = sys.argv[1], int(sys.argv[2])
tri = sys.argv[3] if len(sys.argv) > 3 else "matrix"
repeats = int(sys.argv[4]) if len(sys.argv) > 4 else 3
truth = np.loadtxt(os.path.join(rec, "truth_mm.csv"), delimiter=",").reshape(-1, n_drones, 3).transpose(1, 0, 2)
times = []
with tempfile.TemporaryDirectory() as tmp:
    for _ in range(repeats):
        r = subprocess.run([MAIN, os.path.join(rec, "cameras.xcp"), os.path.join(rec, "dl_data"), "--n_drones", str(n_drones),
                            "--triangulator", tri, "--output_dir", os.path.join(tmp, "results")],
                           check=True, timeout=1200, capture_output=True, text=True)
        times.append(float(re.search(r"Execution time: ([0-9.eE+-]+)s", r.stdout).group(1)))
    paths = [np.array([[float(v) for v in l.split()] for l in open(os.path.join(tmp, "results", f"drone{k + 1}.ply")).read().splitlines()[9:]])
             for k in range(n_drones)]
rows = []
for p in paths:
    valid = p.any(axis=1)
    d = np.linalg.norm(p[None, :, :] - truth[:, :len(p)], axis=2)
    best = int(np.argmin(np.median(d[:, valid], axis=1))) if valid.any() else -1
    dd = d[best, valid] if best >= 0 else np.array([np.nan])
    rows.append({"follows_true_drone": best + 1, "frames_tracked": int(valid.sum()), "median_mm": round(float(np.median(dd)), 1),
                 "p90_mm": round(float(np.percentile(dd, 90)), 1), "within_150mm": round(float((dd < 150).mean()), 3)})
n_frames = len(paths[0])
print(json.dumps({"recording": os.path.basename(os.path.normpath(rec)), "n_drones": n_drones, "triangulator": tri,
                  "frames": n_frames, "classifyDrones_s_best": min(times), "classifyDrones_s_all": times,
                  "frames_per_s": round(n_frames / min(times), 1),
                  "distinct_true_drones_followed": len({r["follows_true_drone"] for r in rows}), "paths": rows}))

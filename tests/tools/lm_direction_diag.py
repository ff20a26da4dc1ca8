"""GPU vs oracle LM status words on 2- and 3-ray systems: which bit disagrees, in which direction, at what iteration
counts.  Diagnostic behind test_ray_lm_status_is_two_sided_on_a_million_random_systems' direction assertion.
Round 5: every population is solved twice on the GPU against ONE oracle run — by the default kernels (exact gradient
above 0.1 mm) and by the literal central-difference build (knob TDR_LM_LITERAL_FD, csrc/tdr_lm_literal.hip) — so the
direction of the two-ray disagreements can be attributed on the device.
usage (GPU box): python tests/tools/lm_direction_diag.py [scale] [seed offset] [n_cam,...]  -> JSON on stdout"""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import tdr_oracle as o  # noqa: E402

tdr = importlib.import_module("3d-reconstruction-triangulation_amd")
sc = importlib.import_module("3d-reconstruction-triangulation_amd.scenes")
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0  # other scenes, paths and noise: the direction is not seed luck
rows_by = {"default": [], "literal_fd": []}
only = {int(v) for v in sys.argv[3].split(",")} if len(sys.argv) > 3 else {2, 3, 4}  # populations to run, by n_cam
for n_cam, drop, n_sys in [(2, 0.0, 240000), (3, 0.0, 240000), (4, 0.3, 400000)]:
    if n_cam not in only:
        continue
    n_sys = int(n_sys * scale)
    cp = sc.ring_camera_params(n_cam, seed=200 + n_cam + seed0)
    cams = [tdr.createCamera(i, c["width"], c["height"], c["focal"], c["tvec"], c["rquat"]) for i, c in enumerate(cp)]
    P, aux = tdr.pack_cameras(cams)
    X = sc.smooth_path(n_sys, seed=7 * n_cam + int(drop * 10) + seed0)
    pts = sc.make_detections(sc.project_ray_model(aux, X), seed=7 * n_cam + int(drop * 10) + seed0, dropout_p=drop, min_valid=2)
    nv = ((pts[..., 0] != -1) & (pts[..., 1] != -1)).sum(0)
    eng = tdr.Engine(P, aux)
    r = o.ray_triangulate_points_ex(aux, pts, 0, os.cpu_count() or 1)
    runs = {}
    for form, lit in (("default", None), ("literal_fd", 1)):
        tdr.set_knob("TDR_LM_LITERAL_FD", lit)
        runs[form] = eng.ray_triangulate_points_ex(pts)
    tdr.set_knob("TDR_LM_LITERAL_FD", None)
    # One certificate for both sides: TDR_LM_NOT_CONVERGED is decided by the device's own pass (lm_status_kernel) on the
    # GPU and by the oracle's code on the CPU — two evaluations of |x - x*| <= 5e-8 max(|x*|, 1) with different
    # roundings, and two-ray runs end NEAR that threshold in large numbers.  Here the flag of BOTH results is recomputed
    # by the same numpy code (x* from numpy.linalg.solve, oracle.np_ray_closed_form_batch); the cap bit is exact on
    # both sides.  If a direction survives this, it is in the runs; if it vanishes, it was in the certificates.
    x_star, _, _ = o.np_ray_closed_form_batch(aux, pts)
    tol = 5e-8 * np.maximum(np.abs(x_star).max(axis=1), 1.0)

    def same_certificate(res):
        dist = np.abs(res["xyz"] - x_star).max(axis=1)
        return ((res["status"] & 1) != 0) | ~(dist <= tol)

    for form, g in runs.items():
        rows = rows_by[form]
        sg, so = g["status"] & 3, r["status"] & 3
        fg2, fo2 = same_certificate(g), same_certificate(r)
        ig, io = g["iters"], np.abs(r["iters"])
        for k in sorted(set(nv.tolist())):
            m = nv == k
            row = {"n_cam": n_cam, "rays": int(k), "systems": int(m.sum()),
                   "flagged_gpu": int((sg[m] != 0).sum()), "flagged_oracle": int((so[m] != 0).sum()),
                   "cap_gpu": int(((sg & 1) != 0)[m].sum()), "cap_oracle": int(((so & 1) != 0)[m].sum()),
                   "notconv_gpu": int(((sg & 2) != 0)[m].sum()), "notconv_oracle": int(((so & 2) != 0)[m].sum()),
                   "mean_iters_gpu": float(ig[m].mean()), "mean_iters_oracle": float(io[m].mean())}
            row["same_certificate"] = {"flagged_gpu": int(fg2[m].sum()), "flagged_oracle": int(fo2[m].sum()),
                                       "gpu_only": int((fg2 & ~fo2)[m].sum()), "oracle_only": int((fo2 & ~fg2)[m].sum()),
                                       "gpu_flag_changed_by_certificate": int((fg2 != (sg != 0))[m].sum()),
                                       "oracle_flag_changed_by_certificate": int((fo2 != (so != 0))[m].sum())}
            both = m & (sg == 0) & (so == 0)
            dd = np.abs(ig[both] - io[both])
            row["both_converged"] = {"n": int(both.sum()), "iters_identical": float((dd == 0).mean()),
                                     "iters_within_5": float((dd <= 5).mean()), "iters_within_20": float((dd <= 20).mean()),
                                     "median_iters_gpu": float(np.median(ig[both])), "median_iters_oracle": float(np.median(io[both]))}
            for name, a, b in (("gpu_only", sg, so), ("oracle_only", so, sg)):
                sel = m & (a != 0) & (b == 0)
                ia, ib = (ig, io) if name == "gpu_only" else (io, ig)
                row[name] = {"n": int(sel.sum()),
                             "status_values": {int(v): int((a[sel] == v).sum()) for v in (1, 2, 3)},
                             "flagged_side_iters_quartiles": [float(x) for x in np.percentile(ia[sel], [25, 50, 75])] if sel.any() else [],
                             "other_side_iters_quartiles": [float(x) for x in np.percentile(ib[sel], [25, 50, 75])] if sel.any() else []}
            rows.append(row)
print(json.dumps({"seed_offset": seed0, "rows": rows_by["default"], "rows_literal_fd": rows_by["literal_fd"]}, indent=1))

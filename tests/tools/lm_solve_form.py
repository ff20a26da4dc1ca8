"""Which deviation from the reference biases the LM status words?  
CPU only: the ORACLE is run in several forms on the same 2-, 3- and 4-camera systems —
  Jacobian:      the reference's central difference | the exact gradient (oracle_lm_analytic_jacobian)
  linear solves: the reference's DECOMP_EIG pseudo-inverse | LDL^T + |det| / max|cof| (oracle_lm_ldlt_solve: what the
                 round-2 HIP kernel did)
  S = sum r^2:   over the square-rooted residuals | from the squared cross products (oracle_lm_sum_from_squares = 1) |
                 the same with the cross product d x w fused (fma) as the round-2 kernel had it (= 2)
  normal eqs.:   J^T J, J^T r from the Jacobian | the "H form" J^T J = H - sum (D / r^2) c c^T, J^T r = H x - b without
                 the gradient (oracle_lm_h_form = 1: a cheaper HIP camera term that round 3 built, measured and dropped,
                 DESIGN.md §4.2) | the same routed as the kernels would have to (= 2: fewer than 4 rays, a pivot within
                 1e-5 of the trace or a ray closer than 0.1 mm -> the whole run again in the reference's form)
and every variant's flag set (status & 3 != 0) is compared with the reference form's.  A deviation that is mere
rounding noise flips flags in both directions equally often (variant-only ~ reference-only); a biased one does not.
usage: python tests/tools/lm_solve_form.py  -> one JSON document on stdout"""
import ctypes as C
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import tdr_oracle as o  # noqa: E402

o.allow_experiments(True)  # this script flips the oracle's experiment switches on purpose (oracle/tdr_oracle.py)

sc = importlib.import_module("3d-reconstruction-triangulation_amd.scenes")
jac = C.c_int.in_dll(o.lib(), "oracle_lm_analytic_jacobian")
ldl = C.c_int.in_dll(o.lib(), "oracle_lm_ldlt_solve")
ssq = C.c_int.in_dll(o.lib(), "oracle_lm_sum_from_squares")
hfm = C.c_int.in_dll(o.lib(), "oracle_lm_h_form")
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
rows = []
for n_cam, drop, n_sys in [(2, 0.0, 60000), (3, 0.0, 60000), (4, 0.3, 100000), (8, 0.5, 100000)]:
    n_sys = int(n_sys * scale)
    cp = sc.ring_camera_params(n_cam, seed=200 + n_cam)
    cams = [o.camera_derive(c["width"], c["height"], c["focal"], c["tvec"], c["rquat"]) for c in cp]
    P, aux = o.cameras_to_arrays(cams)
    X = sc.smooth_path(n_sys, seed=7 * n_cam + int(drop * 10))
    pts = sc.make_detections(sc.project_ray_model(aux, X), seed=7 * n_cam + int(drop * 10), dropout_p=drop, min_valid=2)
    nv = ((pts[..., 0] != -1) & (pts[..., 1] != -1)).sum(0)
    res = {}
    for j in (0, 1):
        for l in (0, 1):
            jac.value, ldl.value = j, l
            res[(j, l)] = o.ray_triangulate_points_ex(aux, pts, 0, os.cpu_count() or 1)
    jac.value = ldl.value = 0
    for q in (1, 2):
        jac.value, ssq.value = 1, q
        res[("ssq", q)] = o.ray_triangulate_points_ex(aux, pts, 0, os.cpu_count() or 1)
    for h in (1, 2):
        jac.value, ssq.value, hfm.value = 1, 1, h
        res[("hform", h)] = o.ray_triangulate_points_ex(aux, pts, 0, os.cpu_count() or 1)
    jac.value = ssq.value = hfm.value = 0
    # a second run of the reference form with a 1-ulp input perturbation: the noise floor of the comparison
    res["ulp"] = o.ray_triangulate_points_ex(aux, pts, 1, os.cpu_count() or 1)
    base = (res[(0, 0)]["status"] & 3) != 0
    for key, name in [((1, 0), "exact gradient, DECOMP_EIG"), ((0, 1), "central difference, LDLT"),
                      ((1, 1), "exact gradient, LDLT (= round-2 kernel)"),
                      (("ssq", 1), "exact gradient, DECOMP_EIG, S from the squared cross products"),
                      (("ssq", 2), "exact gradient, DECOMP_EIG, S from FUSED cross products"),
                      (("hform", 1), "exact gradient, S from the squared cross products, H form of J^T J and J^T r"),
                      (("hform", 2), "the same, routed: < 4 rays / near-singular / small ray -> reference form"),
                      ("ulp", "reference form, input moved by 1 ulp")]:
        f = (res[key]["status"] & 3) != 0
        for k in sorted(set(nv.tolist())):
            m = nv == k
            rows.append({"n_cam": n_cam, "dropout": drop, "rays": int(k), "systems": int(m.sum()), "variant": name,
                         "flagged_reference_form": int(base[m].sum()), "flagged_variant": int(f[m].sum()),
                         "variant_only": int((f & ~base & m).sum()), "reference_only": int((base & ~f & m).sum())})
print(json.dumps({"what": "oracle forms vs the reference form: direction of status-flag disagreements", "rows": rows}, indent=1))

"""VERDICT r3 item 2, CPU only: WHEN may the kernels stay on their LDL^T hot path instead of the reference's DECOMP_EIG
form?  The ORACLE is run on two-, three- and four-ray systems in the reference form (every solve through the Jacobi
eigen-decomposition), with round 3's rule (eigen form iff the smallest unpivoted pivot <= 1e-5 x trace), and with
a sharper rule (eigen form iff det <= tau x trace x e2, i.e. a guaranteed lower bound of the smallest eigenvalue is
within tau of the trace) for several tau — each together with the closed-form two-ray step — and with NO
eigen-decomposition at all (rule 3, oracle sym3_cascade: the DECOMP_EIG answer by rank, the rank decided on the
matrix's invariants at the reference's own cut; what the HIP kernels evaluate since round 4), and every variant's
status flags are compared with the reference form's, as in tests/tools/lm_solve_form.py.  Also counted: how many
systems meet the eigen form at all under each rule (= what the HIP kernels park for their second launch).
usage: python tests/tools/lm_hybrid_rule.py [scale]  -> one JSON document on stdout"""
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
r2 = C.c_int.in_dll(o.lib(), "oracle_lm_rank2_closed_form")
rule = C.c_int.in_dll(o.lib(), "oracle_lm_hybrid_rule")
tau = C.c_double.in_dll(o.lib(), "oracle_lm_hybrid_tau")
n_eig = C.c_long.in_dll(o.lib(), "oracle_lm_hybrid_eig_systems")
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
nthr = os.cpu_count() or 1
VARIANTS = [("round-3 rule: pivot <= 1e-5 trace", 1, 1e-5), ("det/e2 rule, tau 1e-8", 2, 1e-8),
            ("det/e2 rule, tau 1e-10", 2, 1e-10), ("det/e2 rule, tau 1e-12", 2, 1e-12),
            ("rule 3: no eigen-decomposition, rank decided on tr / e2 / det at the reference's cut", 3, 0.0)]
rows = []
eig_systems = {}
for n_cam, drop, n_sys, seed in [(2, 0.0, 200000, 5), (2, 0.0, 200000, 105), (3, 0.0, 300000, 7), (3, 0.0, 300000, 107), (4, 0.25, 400000, 9), (6, 0.4, 300000, 13), (8, 0.55, 300000, 17)]:
    n_sys = int(n_sys * scale)
    cp = sc.ring_camera_params(n_cam, seed=200 + n_cam + seed)
    cams = [o.camera_derive(c["width"], c["height"], c["focal"], c["tvec"], c["rquat"]) for c in cp]
    P, aux = o.cameras_to_arrays(cams)
    X = sc.smooth_path(n_sys, seed=seed)
    pts = sc.make_detections(sc.project_ray_model(aux, X), seed=seed, dropout_p=drop, min_valid=2)
    nv = ((pts[..., 0] != -1) & (pts[..., 1] != -1)).sum(0)
    r2.value = rule.value = 0
    ref = o.ray_triangulate_points_ex(aux, pts, 0, nthr)
    ulp = o.ray_triangulate_points_ex(aux, pts, 1, nthr)
    res = {"reference form, input moved by 1 ulp": ulp}
    for name, rl, t in VARIANTS:
        r2.value, rule.value, tau.value, n_eig.value = 1, rl, t, 0
        res[name] = o.ray_triangulate_points_ex(aux, pts, 0, nthr)
        eig_systems[name] = eig_systems.get(name, 0) + n_eig.value
    r2.value = rule.value = 0
    base = (ref["status"] & 3) != 0
    for name, r in res.items():
        f = (r["status"] & 3) != 0
        for k in sorted(set(nv.tolist())):
            if k > 4:
                continue
            m = nv == k
            both = m & ~f & ~base
            rel = np.abs(r["xyz"][both] - ref["xyz"][both]).max(axis=1) / np.maximum(np.abs(ref["xyz"][both]).max(axis=1), 1.0)
            rows.append({"n_cam": n_cam, "dropout": drop, "rays": int(k), "systems": int(m.sum()), "variant": name,
                         "flagged_reference_form": int(base[m].sum()), "flagged_variant": int(f[m].sum()),
                         "variant_only": int((f & ~base & m).sum()), "reference_only": int((base & ~f & m).sum()),
                         "both_converged_max_rel_dp": float(rel.max()) if rel.size else None})
tot = {}
for r in rows:
    t = tot.setdefault(r["variant"], {}).setdefault(str(r["rays"]), [0, 0, 0])
    t[0] += r["variant_only"]; t[1] += r["reference_only"]; t[2] += r["systems"]
print(json.dumps({"what": "LDL^T / eigen-form switching rules vs the reference form (CPU oracle): variant-only : reference-only : systems by ray count",
                  "totals": tot, "systems_that_met_the_eigen_form": eig_systems, "rows": rows}, indent=1))

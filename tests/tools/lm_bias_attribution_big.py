"""Round 5: WHICH departure carries the two-ray status residue?  profiles/r05_lm_literal_fd.json measured it at
-6.8e-5 +- 1.1e-5 per system on 38.4 M two-ray systems (GPU-only : oracle-only 80 872 : 83 498), the same with the
literal central-difference build — so not the Jacobian form — and concentrated in some rigs (seed offsets 8000, 9000,
12000: -1.6e-4 ... -3e-4) and in runs that stop on the STEP test away from the minimiser (status 2).
This script takes those rigs and runs, on 2.4 M two-ray systems each, the ORACLE in its reference form, the ORACLE with
one departure at a time (exact gradient; S from squared cross products; the same with the fused cross product; the
closed-form rank cascade; all three = the kernels' form) and the GPU (default build), and prints every pair's
variant-only : reference-only flags.  If the oracle in the kernels' form is fair against the GPU, the three known
departures are the whole story and the single-departure rows say which one; if not, something the oracle's switches do
not model is.
usage (GPU box): python tests/tools/lm_bias_attribution_big.py [scale] [offset ...]  -> JSON on stdout"""
import ctypes as C
import importlib
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import tdr_oracle as o  # noqa: E402

o.allow_experiments(True)  # this script flips the oracle's experiment switches on purpose (oracle/tdr_oracle.py)
tdr = importlib.import_module("3d-reconstruction-triangulation_amd")
sc = importlib.import_module("3d-reconstruction-triangulation_amd.scenes")
L = o.lib()
jac = C.c_int.in_dll(L, "oracle_lm_analytic_jacobian")
ssq = C.c_int.in_dll(L, "oracle_lm_sum_from_squares")
rule = C.c_int.in_dll(L, "oracle_lm_hybrid_rule")
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
offsets = [int(v) for v in sys.argv[2:]] or [8000, 9000, 12000]
NT = os.cpu_count() or 1
VARIANTS = [("exact gradient", 1, 0, 0), ("S from squares", 0, 1, 0), ("S from squares, fused cross", 0, 2, 0),
            ("rank cascade", 0, 0, 3), ("kernels' form: all three", 1, 1, 3)]
rows = []
for off in offsets:
    n_cam, n_sys = 2, int(240000 * scale)
    cp = sc.ring_camera_params(n_cam, seed=200 + n_cam + off)
    cams = [tdr.createCamera(i, c["width"], c["height"], c["focal"], c["tvec"], c["rquat"]) for i, c in enumerate(cp)]
    P, aux = tdr.pack_cameras(cams)
    X = sc.smooth_path(n_sys, seed=7 * n_cam + off)
    pts = sc.make_detections(sc.project_ray_model(aux, X), seed=7 * n_cam + off, dropout_p=0.0, min_valid=2)
    t0 = time.time()
    flags = {"gpu": (tdr.Engine(P, aux).ray_triangulate_points_ex(pts)["status"] & 3) != 0}
    jac.value = ssq.value = rule.value = 0
    flags["reference form"] = (o.ray_triangulate_points_ex(aux, pts, 0, NT)["status"] & 3) != 0
    for name, j, s, r in VARIANTS:
        jac.value, ssq.value, rule.value = j, s, r
        flags[name] = (o.ray_triangulate_points_ex(aux, pts, 0, NT)["status"] & 3) != 0
    jac.value = ssq.value = rule.value = 0

    def pair(a, b):
        fa, fb = flags[a], flags[b]
        x, y = int((fa & ~fb).sum()), int((fb & ~fa).sum())
        return {"offset": off, "systems": n_sys, "A": a, "B": b, "A_only": x, "B_only": y,
                "sigma": round((x - y) / math.sqrt(x + y), 2) if x + y else 0.0, "bias_per_system": (x - y) / n_sys}

    rows.append(pair("gpu", "reference form"))
    for name, *_ in VARIANTS:
        rows.append(pair(name, "reference form"))
    rows.append(pair("gpu", "kernels' form: all three"))
    for r in rows[-(len(VARIANTS) + 2):]:
        print(r, file=sys.stderr, flush=True)
    print(f"  offset {off}: {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
tot = {}
for r in rows:
    t = tot.setdefault((r["A"], r["B"]), [0, 0, 0])
    t[0] += r["A_only"]
    t[1] += r["B_only"]
    t[2] += r["systems"]
summary = [{"A": a, "B": b, "A_only": x, "B_only": y, "systems": n, "sigma": round((x - y) / math.sqrt(x + y), 2),
            "bias_per_system": (x - y) / n, "resolution_1_sigma": math.sqrt(x + y) / n} for (a, b), (x, y, n) in tot.items()]
print(json.dumps({"scale": scale, "offsets": offsets, "summary": summary, "rows": rows}, indent=1))

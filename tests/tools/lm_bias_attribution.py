"""Is the small residue in the direction of GPU-vs-oracle status disagreements (HISTORY.md §2.2: 2 173 : 2 322 on 2.6 M
low-ray systems) one of the kernel's known deviations from the reference form?  CPU only: the ORACLE in four forms on
2.4 M two- and three-ray systems (two scene seeds), each form's flag set against the reference form's.
usage: python tests/tools/lm_bias_attribution.py  -> rows on stdout, JSON in /tmp/lm_bias.json"""
import ctypes as C, importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import tdr_oracle as o

o.allow_experiments(True)  # this script flips the oracle's experiment switches on purpose (oracle/tdr_oracle.py)
sc = importlib.import_module("3d-reconstruction-triangulation_amd.scenes")
jac = C.c_int.in_dll(o.lib(), "oracle_lm_analytic_jacobian")
ssq = C.c_int.in_dll(o.lib(), "oracle_lm_sum_from_squares")
rule = C.c_int.in_dll(o.lib(), "oracle_lm_hybrid_rule")
rows=[]
for seed0 in (1000, 2000):
  for n_cam, n_sys in [(2, 600000), (3, 600000)]:
    cp = sc.ring_camera_params(n_cam, seed=200 + n_cam + seed0)
    cams = [o.camera_derive(c["width"], c["height"], c["focal"], c["tvec"], c["rquat"]) for c in cp]
    P, aux = o.cameras_to_arrays(cams)
    X = sc.smooth_path(n_sys, seed=7 * n_cam + seed0)
    pts = sc.make_detections(sc.project_ray_model(aux, X), seed=7 * n_cam + seed0, dropout_p=0.0, min_valid=2)
    t0=time.time()
    res={}
    jac.value=ssq.value=0
    res['ref']=o.ray_triangulate_points_ex(aux, pts, 0, 8)
    res['ulp']=o.ray_triangulate_points_ex(aux, pts, 1, 8)
    jac.value=1
    res['exact gradient']=o.ray_triangulate_points_ex(aux, pts, 0, 8)
    ssq.value=1
    res['exact gradient + S from squares (kernel form)']=o.ray_triangulate_points_ex(aux, pts, 0, 8)
    rule.value=3  # round 4: + the eigen-free cascade = everything the kernels do differently from the reference form
    res['exact gradient + S from squares + lm_cascade (round-4 kernel form)']=o.ray_triangulate_points_ex(aux, pts, 0, 8)
    rule.value=0
    jac.value=0
    res['central difference + S from squares']=o.ray_triangulate_points_ex(aux, pts, 0, 8)
    jac.value=ssq.value=0
    base=(res['ref']['status']&3)!=0
    for k,v in res.items():
        if k=='ref': continue
        f=(v['status']&3)!=0
        rows.append({'seed':seed0,'n_cam':n_cam,'variant':k,'flagged_ref':int(base.sum()),'flagged_variant':int(f.sum()),'variant_only':int((f&~base).sum()),'reference_only':int((base&~f).sum())})
        print(rows[-1], round(time.time()-t0,1), flush=True)
json.dump(rows, open('/tmp/lm_bias.json','w'), indent=1)

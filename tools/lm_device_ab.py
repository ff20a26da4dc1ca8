"""Two builds / knob settings of the LM kernels against EACH OTHER on the device, on samples the CPU oracle could never
run: status-flag disagreements (A-only : B-only) and the agreement of the converged results, on synthetic n-camera
streams generated on the device (SURVEY.md §8d generator, ray model, 2 px noise, rounded pixels, no drop-outs unless
asked).  No oracle involved: this measures what ONE departure does, with a resolution of ~1e-6 per system at 500 M
systems (the oracle-side experiments of rounds 3-4 stopped at 4.5 M).

usage: python tools/lm_device_ab.py <n_cam> <million systems> "<knobs A>" "<knobs B>" [chunk millions] [dropout]
       knobs: "K=v,K=v" or "" for the defaults, e.g.  python tools/lm_device_ab.py 2 200 "" TDR_LM_LITERAL_FD=1
prints one JSON object."""
import importlib
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tdr = importlib.import_module("3d-reconstruction-triangulation_amd")
sc = importlib.import_module("3d-reconstruction-triangulation_amd.scenes")

n_cam = int(sys.argv[1])
n_total = int(float(sys.argv[2]) * 1e6)
sets = [sys.argv[3], sys.argv[4]]
chunk = int(float(sys.argv[5]) * 1e6) if len(sys.argv) > 5 else 10_000_000
dropout = float(sys.argv[6]) if len(sys.argv) > 6 else 0.0
ALL = ["TDR_LM_COOP", "TDR_LM_COOP_MAX", "TDR_LM_FULL", "TDR_LM_CHUNK", "TDR_LM_REFILL", "TDR_LM_NLDS", "TDR_LM_LITERAL_FD"]


def apply(s):
    for k in ALL:
        tdr.set_knob(k, None)
    for kv in filter(None, s.split(",")):
        k, v = kv.split("=")
        tdr.set_knob(k, int(v))


tot = {"systems": 0, "flagged_A": 0, "flagged_B": 0, "A_only": 0, "B_only": 0, "cap_A": 0, "cap_B": 0,
       "both_converged": 0, "iters_identical": 0, "iters_within_5": 0}
worst = 0.0
ms = [0.0, 0.0]
t_start = time.time()
scene = 0
done = 0
while done < n_total:
    n = min(chunk, n_total - done)
    # a new rig every chunk (other camera poses), a new stream seed: the direction must not be one scene's luck
    cp = sc.ring_camera_params(n_cam, seed=1000 + scene)
    cams = [tdr.createCamera(i, c["width"], c["height"], c["focal"], c["tvec"], c["rquat"]) for i, c in enumerate(cp)]
    P, aux = tdr.pack_cameras(cams)
    eng = tdr.Engine(P, aux)
    d_pts = eng.alloc(n_cam * n * 16)
    d_xyz = eng.alloc(n * 24)
    d_it = eng.alloc(n * 4)
    d_st = eng.alloc(n * 4)
    eng.synth_device(0xABC000 + scene, 1, 0, n, n, d_pts.ptr, dropout_p=dropout, ray_model=True)
    eng.synchronize()
    res = []
    for k, s in enumerate(sets):
        apply(s)
        t0 = time.perf_counter()
        eng.ray_device_ex(d_pts.ptr, n_cam, n, n, d_xyz.ptr, None, d_it.ptr, d_st.ptr)
        eng.synchronize()
        ms[k] += (time.perf_counter() - t0) * 1e3
        res.append((d_xyz.download((n, 3)), d_it.download((n,), np.int32), d_st.download((n,), np.int32)))
    apply("")
    (xa, ia, sa), (xb, ib, sb) = res
    fa, fb = (sa & 3) != 0, (sb & 3) != 0
    ok = ~fa & ~fb
    tot["systems"] += n
    tot["flagged_A"] += int(fa.sum())
    tot["flagged_B"] += int(fb.sum())
    tot["A_only"] += int((fa & ~fb).sum())
    tot["B_only"] += int((fb & ~fa).sum())
    tot["cap_A"] += int(((sa & 1) != 0).sum())
    tot["cap_B"] += int(((sb & 1) != 0).sum())
    tot["both_converged"] += int(ok.sum())
    dd = np.abs(ia[ok] - ib[ok])
    tot["iters_identical"] += int((dd == 0).sum())
    tot["iters_within_5"] += int((dd <= 5).sum())
    rel = np.abs(xa[ok] - xb[ok]).max(axis=1) / np.maximum(np.abs(xb[ok]).max(axis=1), 1.0)
    worst = max(worst, float(rel.max()) if rel.size else 0.0)
    for b in (d_pts, d_xyz, d_it, d_st):
        b.free()
    eng.close()
    done += n
    scene += 1
    print(f"  {done / 1e6:.0f} M: A-only {tot['A_only']} B-only {tot['B_only']}", file=sys.stderr, flush=True)
a, b = tot["A_only"], tot["B_only"]
out = {"n_cam": n_cam, "dropout": dropout, "A": sets[0] or "defaults", "B": sets[1] or "defaults", **tot,
       "sigma_off_fair": (a - b) / math.sqrt(a + b) if a + b else 0.0,
       "bias_per_system_A_minus_B": (a - b) / tot["systems"],
       "bias_resolution_1_sigma": math.sqrt(a + b) / tot["systems"] if a + b else None,
       "converged_max_rel_difference": worst, "kernel_ms_total_A_B": [round(v, 1) for v in ms],
       "seconds": round(time.time() - t_start, 1)}
print(json.dumps(out))

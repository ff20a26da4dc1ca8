#!/usr/bin/env python3
"""Fixture tests/golden/s02_d4_a/ — the reference's simulated recording dataset/S02_D4_A (8 cameras, 4 drones, ground
truth), the only data it ships with 8 cameras AND known drone positions.  Run once in the build container (where
/root/reference is mounted); the tests never read /root/reference.

DATA copied from the reference's dataset/ tree (inputs and ground truth only):
  dl_data/cam{1..8}i_s.csv      the per-camera detection files, rows with frame number <= 224 (cam8's file ends at 225 with a broken row;
                                the reference's loader refuses recordings whose cameras disagree on the frame count)
  stationary_camera_data.csv    the eight camera records as shipped
  truth_mm.csv                  drone_data/dron{1..4}_pos_25.csv, columns pos_x..pos_z of rows 0..224, in millimetres
WRITTEN BY THIS SCRIPT:
  cameras.xcp                   the eight cameras in the format utils.cpp's loadCamerasXML reads.  The recording ships
                                no .xcp; the reference's workflow derives one by hand from stationary_camera_data.csv
                                (src/generate_xcp.cpp prints position = -R^-1 t for a Rodrigues vector).  Which
                                convention the file's columns follow is decided here FROM THE DATA: the true drone
                                positions are projected through four candidate readings of (cam_x..z, cam_or_x..z) and
                                compared with the detections; the winner's numbers go into convention_probe.json.
                                Positions are scaled to millimetres (the classifier's thresholds — MAX_STEP 200,
                                error limits — are in the real recordings' unit).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/dataset/S02_D4_A"
OUT = os.path.join(HERE, "s02_d4_a")
LAST_FRAME = 224
W, H = 1920, 1080


def rodrigues(r):
    th = np.linalg.norm(r)
    if th < 1e-12:
        return np.eye(3)
    k = r / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def rot2quat(R):
    """(w, x, y, z) with toRotMatrix(q) == R (src/Camera.h:274-287 is the standard Hamilton form)"""
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1) * 2
        return np.array([.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    if R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = np.sqrt(1 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        return np.array([(R[2, 1] - R[1, 2]) / s, .25 * s, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s])
    if R[1, 1] > R[2, 2]:
        s = np.sqrt(1 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        return np.array([(R[0, 2] - R[2, 0]) / s, (R[0, 1] + R[1, 0]) / s, .25 * s, (R[1, 2] + R[2, 1]) / s])
    s = np.sqrt(1 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
    return np.array([(R[1, 0] - R[0, 1]) / s, (R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, .25 * s])


def main(SRC=SRC, OUT=OUT, LAST_FRAME=LAST_FRAME, N_DRONES=4):
    os.makedirs(os.path.join(OUT, "dl_data"), exist_ok=True)
    cam_text = open(os.path.join(SRC, "stationary_camera_data.csv")).read()
    open(os.path.join(OUT, "stationary_camera_data.csv"), "w").write(cam_text)
    cams = []
    for line in cam_text.strip().split("\n")[1:]:
        f = line.split(";")
        cams.append((np.array(f[1:4], float), np.array(f[4:7], float), float(f[7])))
    truth = []
    for k in range(1, N_DRONES + 1):
        rows = [l.split(";") for l in open(os.path.join(SRC, "drone_data", f"dron{k}_pos_25.csv")).read().strip().split("\n")[1:]]
        truth.append(np.array([[float(x) for x in r[1:4]] for r in rows[:LAST_FRAME + 1]]))
    truth = np.array(truth)  # [drone][frame][3] metres
    dets = []
    for c in range(1, 9):
        kept, frames = [], {}
        for l in open(os.path.join(SRC, "dl_data", f"cam{c}i_s.csv")).read().strip().split("\n"):
            f = l.split(",")
            if int(f[0]) > LAST_FRAME:
                continue
            kept.append(l)
            v = [float(x) for x in f[1:] if x != ""]
            frames[int(f[0])] = np.array(v).reshape(-1, 7)[:, 4:6] if len(v) >= 7 else np.zeros((0, 2))
        open(os.path.join(OUT, "dl_data", f"cam{c}i_s.csv"), "w").write("\n".join(kept) + "\n")
        dets.append(frames)

    # ---- which convention?  x_cam for a world point X under four readings of (t, r) = (cam_x..z, cam_or_x..z)
    readings = {
        "A: R(r) X + t      (OpenCV extrinsics: rotation vector world->camera, t = translation)": lambda R, t, X: R @ X + t,
        "B: R(r)^T (X - t)  (t = camera centre, r = camera->world)": lambda R, t, X: R.T @ (X - t),
        "C: R(r) (X - t)    (t = camera centre, r = world->camera)": lambda R, t, X: R @ (X - t),
        "D: R(r)^T X + t": lambda R, t, X: R.T @ X + t,
    }
    probe = {}
    for name, fn in readings.items():
        med = []
        for ci, (t, r, fov) in enumerate(cams):
            f = (W / 2) / np.tan(np.radians(fov) / 2)
            R = rodrigues(r)
            errs = []
            for fr in range(0, LAST_FRAME + 1, 5):
                if fr not in dets[ci] or len(dets[ci][fr]) == 0:
                    continue
                for k in range(N_DRONES):
                    xc = fn(R, t, truth[k, fr])
                    if xc[2] <= 0:
                        errs.append(1e4)  # behind the camera
                        continue
                    u, v = W / 2 + f * xc[0] / xc[2], H / 2 + f * xc[1] / xc[2]
                    errs.append(float(np.min(np.hypot(dets[ci][fr][:, 0] - u, dets[ci][fr][:, 1] - v))))
            med.append(round(float(np.median(errs)), 1))
        probe[name] = med
        print(name, med)
    best = min(probe, key=lambda n: max(probe[n]))
    # (camera 7 is turned by pi, so R = R^T there and reading D agrees with A on that one camera)
    assert best.startswith("A:") and max(probe[best]) < 40 and all(max(v) > 200 for n, v in probe.items() if n != best)
    json.dump({"what": "median distance [px] between the projection of the true drone positions and the nearest detection, "
                       "per camera, for four readings of stationary_camera_data.csv (every 5th frame, 4 drones)",
               "medians_px_per_camera": probe, "chosen": best}, open(os.path.join(OUT, "convention_probe.json"), "w"), indent=1)

    # ---- cameras.xcp in the loader's parametrisation (src/Camera.h:133-170: extrinsic = [Rq^-1 | -Rq^-1 tvec], i.e.
    # the loader's "POSITION" is the camera centre and its quaternion rotates camera -> world)
    lines = ['<?xml version="1.0" encoding="UTF-8" standalone="no" ?>', '<Cameras VERSION="1.0">']
    for i, (t, r, fov) in enumerate(cams):
        R = rodrigues(r)
        centre_mm = -R.T @ t * 1000.0
        q = rot2quat(R.T)
        focal = (W / 2) / np.tan(np.radians(fov) / 2)
        num = lambda v: repr(float(v))  # noqa: E731  (shortest round-trip decimal)
        lines += [f'  <Camera DEVICEID="{i + 1}" NAME="cam{i + 1}" TYPE="simulated">', "    <ControlFrames>",
                  f'      <ControlFrame FOCAL_LENGTH="{num(focal)}" FRAME="0" ORIENTATION="{" ".join(num(v) for v in q)}" '
                  f'POSITION="{" ".join(num(v) for v in centre_mm)}" PRINCIPAL_POINT="{W // 2} {H // 2}" VICON_RADIAL="0 0"/>',
                  "    </ControlFrames>", "  </Camera>"]
    lines.append("</Cameras>")
    open(os.path.join(OUT, "cameras.xcp"), "w").write("\n".join(lines) + "\n")
    with open(os.path.join(OUT, "truth_mm.csv"), "w") as f:
        for fr in range(LAST_FRAME + 1):
            f.write(",".join(repr(float(v)) for k in range(N_DRONES) for v in truth[k, fr] * 1000.0) + "\n")
    print("wrote", sorted(os.listdir(OUT)), sum(os.path.getsize(os.path.join(dp, fn)) for dp, _, fns in os.walk(OUT) for fn in fns), "bytes")


# The committed excerpt tests/golden/s09_d6_excerpt (first 150 frames of dataset/S09_D6: 8 cameras, SIX drones) was
# written by:  python tests/golden/make_s02_d4_a.py S09_D6 tests/golden/s09_d6_excerpt 149 6
# and tests/golden/s01_d2_a_excerpt (first 200 frames of dataset/S01_D2_A, two drones) by:  ... S01_D2_A tests/golden/s01_d2_a_excerpt 199 2
if __name__ == "__main__":
    if len(sys.argv) > 1:  # another recording of the same kind, e.g.  S09_D6 tests/golden/_local_s09_d6 2999 6
        main(os.path.join("/root/reference/dataset", sys.argv[1]), sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
    else:
        main()

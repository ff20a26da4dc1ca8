"""End-to-end classifier measurement (SURVEY.md §8f rows 1-4 together): writes a synthetic multi-drone
recording as the reference's on-disk formats (.xcp + per-camera CSVs), runs tdr_main (prints the
reference's "Execution time" around classifyDrones) and the Python restatement of the reference
classifier on the same files, and checks the PLY outputs agree.
usage: python tests/tools/bench_classifier.py [n_cam] [n_drones] [n_frames] [oracle_frames] [matrix|ray]"""
import importlib, json, os, re, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
sc = importlib.import_module("3d-reconstruction-triangulation_amd.scenes")
import tdr_oracle as oracle, classifier_oracle as co, io_oracle as io

n_cam = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n_drones = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n_frames = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
oracle_frames = int(sys.argv[4]) if len(sys.argv) > 4 else 300
tri = sys.argv[5] if len(sys.argv) > 5 else "matrix"
MAIN = os.path.join(ROOT, "3d-reconstruction-triangulation_amd", "tdr_main")

cp = sc.ring_camera_params(n_cam, seed=77)
cams = [oracle.camera_derive(c["width"], c["height"], c["focal"], c["tvec"], c["rquat"]) for c in cp]
P, aux = oracle.cameras_to_arrays(cams)
counts, dets, truth = sc.multi_drone_frames(P, n_frames, n_drones, seed=11)


def write_recording(d, nf):
    os.makedirs(os.path.join(d, "dl_data"))
    with open(os.path.join(d, "cameras.xcp"), "w") as f:
        f.write('<?xml version="1.0"?>\n<Cameras VERSION="1.0">\n')
        for i, c in enumerate(cp):
            q, t = [float(v) for v in c["rquat"]], [float(v) for v in c["tvec"]]
            f.write(f'  <Camera DEVICEID="{1000 + i}" TYPE="DCAM">\n    <ControlFrames>\n      <ControlFrame '
                    f'FOCAL_LENGTH="{float(c["focal"])!r}" FRAME="0" ORIENTATION="{q[0]!r} {q[1]!r} {q[2]!r} {q[3]!r}" '
                    f'POSITION="{t[0]!r} {t[1]!r} {t[2]!r}" PRINCIPAL_POINT="{c["width"] // 2} {c["height"] // 2}"/>\n'
                    f'    </ControlFrames>\n  </Camera>\n')
        f.write("</Cameras>\n")
    off = np.concatenate([[0], np.cumsum(counts.reshape(-1))]).astype(np.int64)
    for c in range(n_cam):
        with open(os.path.join(d, "dl_data", f"cam{c:02d}.csv"), "w") as f:
            for fr in range(nf):
                rec = dets[off[fr * n_cam + c]:off[fr * n_cam + c + 1]]
                f.write(str(fr) + "".join(f",0,0,10,10,{int(x)},{int(y)},0.9" for x, y in rec) + "\n")


with tempfile.TemporaryDirectory() as tmp:
    full, part = os.path.join(tmp, "full"), os.path.join(tmp, "part")
    write_recording(full, n_frames)
    write_recording(part, oracle_frames)
    t0 = time.perf_counter()
    r = subprocess.run([MAIN, os.path.join(full, "cameras.xcp"), os.path.join(full, "dl_data"), "--n_drones",
                        str(n_drones), "--triangulator", tri, "--output_dir", os.path.join(tmp, "out_full")], check=True,
                       capture_output=True, text=True)
    wall_full = time.perf_counter() - t0
    classify_s = float(re.search(r"Execution time: ([0-9.e+-]+)s", r.stdout).group(1))
    subprocess.run([MAIN, os.path.join(part, "cameras.xcp"), os.path.join(part, "dl_data"), "--n_drones",
                    str(n_drones), "--triangulator", tri, "--output_dir", os.path.join(tmp, "out_part")], check=True,
                   capture_output=True)
    # restated reference flow on the shorter recording (pure Python control flow + C oracle solves)
    cams_o = io.load_cameras_xml(os.path.join(part, "cameras.xcp"))
    Po, auxo = oracle.cameras_to_arrays(cams_o)
    co_counts, co_dets = io.read_detection_csvs(os.path.join(part, "dl_data"))
    t0 = time.perf_counter()
    ref = co.DroneClassifier(Po, auxo, n_drones, tri).classify_drones(co.Container.from_flat(co_counts, co_dets))
    oracle_s = time.perf_counter() - t0
    same = all(open(os.path.join(tmp, "out_part", f"drone{k + 1}.ply")).read() == io.ply_text(ref[k])
               for k in range(n_drones))
    tracked = sum(int(p.any(axis=1).sum()) for p in ref)
print(json.dumps({"workload": f"{n_cam} cameras x {n_drones} drones x {n_frames} frames (synthetic recording on disk)",
                  "triangulator": tri,
                  "tdr_main_classifyDrones_s": classify_s, "tdr_main_wall_s_incl_ingest_and_ply": wall_full,
                  "frames_per_s": n_frames / classify_s,
                  "oracle_frames": oracle_frames, "oracle_classify_s": oracle_s,
                  "oracle_frames_per_s": oracle_frames / oracle_s,
                  "oracle_kind": "Python control flow + C oracle solves (prefix-tree walk), 1 thread",
                  "ply_identical_on_oracle_frames": same, "points_tracked_by_oracle": tracked}))

"""dataset/S02_D4_A — the reference's only recording with 8 cameras AND ground truth (simulated: 4 drones, known
positions) — through the whole flow: loadCamerasXML -> DetectionsContainer -> DroneClassifier -> triangulation.
PHYSICAL SANITY, not parity: the recording ships no camera file in the loader's format; tests/golden/make_s02_d4_a.py
derives cameras.xcp from its stationary_camera_data.csv, choosing the convention from the data itself (the true drone
positions projected through four candidate readings, compared with the detections: convention_probe.json), and the
fixture keeps frames 0..224 (cam8's file ends at frame 225 with a cut-off row, which the loader rejects).  Truth is the drones' body origin, a detection the centre of a
~100 px bounding box: tens of millimetres between the two are expected."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "s02_d4_a")
MAIN = os.path.join(ROOT, "3d-reconstruction-triangulation_amd", "tdr_main")
# the first 150 frames of dataset/S09_D6 (8 cameras, SIX drones, up to 7 detections per camera and frame), same generator
GOLD6 = os.path.join(ROOT, "tests", "golden", "s09_d6_excerpt")


def _truth(gold=GOLD, n_drones=4):
    return np.loadtxt(os.path.join(gold, "truth_mm.csv"), delimiter=",").reshape(-1, n_drones, 3).transpose(1, 0, 2)


def _against_truth(paths, gold=GOLD, n_drones=4):
    """per estimated path: (index of the true drone it follows, distances on its valid frames)"""
    T = _truth(gold, n_drones)
    out = []
    for p in paths:
        p = np.asarray(p)
        valid = p.any(axis=1)
        d = np.linalg.norm(p[None, :, :] - T[:, :len(p)], axis=2)
        best = int(np.argmin(np.median(d[:, valid], axis=1)))
        out.append((best, d[best, valid]))
    return out


def _oracle_paths(oracle, triangulator="matrix", gold=GOLD, n_drones=4, n_frames=225):
    import classifier_oracle as co
    import io_oracle as io
    cams = io.load_cameras_xml(os.path.join(gold, "cameras.xcp"), False)
    P, aux = oracle.cameras_to_arrays(cams)
    counts, dets = io.read_detection_csvs(os.path.join(gold, "dl_data"))
    assert counts.shape == (n_frames, 8)
    paths = co.DroneClassifier(P, aux, n_drones, triangulator).classify_drones(co.Container.from_flat(counts, dets))
    return paths, io


def _check(stats, label):
    assert sorted(b for b, _ in stats) == [0, 1, 2, 3], f"{label}: the four paths must follow four different drones"
    med = [float(np.median(d)) for _, d in stats]
    n_valid = [len(d) for _, d in stats]
    within = [float((d < 150).mean()) for _, d in stats]
    print(f"S02_D4_A {label}: median distance to the true drone {np.round(med, 1)} mm, frames tracked {n_valid}, "
          f"within 150 mm {np.round(within, 3)}")
    # measured with the restated reference flow: medians 43 / 46 / 51 / 79 mm, 184-225 of 225 frames tracked, two paths
    # within 150 mm on every frame (the other two lose their drone for a stretch, as the reference's classifier does)
    assert max(med) < 120 and min(n_valid) >= 150
    assert sorted(within)[-2] > 0.98 and min(within) > 0.5


def test_convention_was_fixed_from_the_data():
    probe = json.load(open(os.path.join(GOLD, "convention_probe.json")))
    med = probe["medians_px_per_camera"]
    assert probe["chosen"].startswith("A:") and max(med[probe["chosen"]]) < 40
    assert all(max(v) > 200 for k, v in med.items() if k != probe["chosen"])


def test_reference_flow_follows_the_true_drones(oracle):
    paths, _ = _oracle_paths(oracle)
    _check(_against_truth(paths), "oracle flow, matrix")


@pytest.mark.gpu
@pytest.mark.parametrize("triangulator", ["matrix", "ray"])
def test_cli_on_the_simulated_recording(tmp_path, oracle, _built, triangulator):
    out_dir = tmp_path / "results"
    r = subprocess.run([MAIN, os.path.join(GOLD, "cameras.xcp"), os.path.join(GOLD, "dl_data"), "--n_drones", "4",
                        "--triangulator", triangulator, "--output_dir", str(out_dir)],
                       check=True, timeout=600, capture_output=True, text=True)
    assert "Execution time:" in r.stdout
    got = [np.array([[float(v) for v in l.split()] for l in open(out_dir / f"drone{k + 1}.ply").read().splitlines()[9:]])
           for k in range(4)]
    _check(_against_truth(got), f"tdr_main, {triangulator}")
    if triangulator == "matrix":  # and the product equals the restated reference flow byte for byte
        ref, io = _oracle_paths(oracle)
        for k in range(4):
            assert open(out_dir / f"drone{k + 1}.ply").read() == io.ply_text(ref[k]), f"drone{k + 1}.ply differs"


@pytest.mark.gpu
def test_cli_on_the_six_drone_recording_if_a_local_copy_exists(tmp_path, _built):
    """dataset/S09_D6 (8 cameras, 6 drones, 3000 frames, up to 7 detections per camera and frame) is too large to
    commit; `python tests/golden/make_s02_d4_a.py S09_D6 tests/golden/_local_s09_d6 2999 6` makes a git-ignored copy
    in the build container, which travels to the GPU box with the snapshot."""
    rec = os.path.join(ROOT, "tests", "golden", "_local_s09_d6")
    if not os.path.exists(os.path.join(rec, "cameras.xcp")):
        pytest.skip("no local copy of S09_D6")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_recording.py"), rec, "6", "matrix", "1"], check=True,
                       timeout=600, capture_output=True, text=True)
    out = json.loads(r.stdout.strip().splitlines()[-1])
    print(f"S09_D6: classifyDrones {out['classifyDrones_s_best']:.3f} s = {out['frames_per_s']:.0f} frames/s; "
          f"medians {[p['median_mm'] for p in out['paths']]} mm")
    assert out["frames"] == 3000 and out["distinct_true_drones_followed"] == 6
    assert max(p["median_mm"] for p in out["paths"]) < 100 and min(p["frames_tracked"] for p in out["paths"]) > 2900


def _check6(stats, label):
    assert sorted(b for b, _ in stats) == [0, 1, 2, 3, 4, 5], f"{label}: six paths, six different drones"
    med = [float(np.median(d)) for _, d in stats]
    within = sorted(float((d < 150).mean()) for _, d in stats)
    print(f"S09_D6 excerpt {label}: median distance to the true drone {np.round(med, 1)} mm, within 150 mm {np.round(within, 3)}")
    # measured with the restated reference flow: medians 40 / 41 / 44 / 46 / 70 / 107 mm, every drone on all 150
    # frames, four paths within 150 mm on >= 99 % of them (two paths trade drones halfway, in the reference's flow too)
    assert max(med) < 150 and min(len(d) for _, d in stats) >= 140 and within[2] > 0.95


def test_reference_flow_follows_six_drones(oracle):
    paths, _ = _oracle_paths(oracle, gold=GOLD6, n_drones=6, n_frames=150)
    _check6(_against_truth(paths, GOLD6, 6), "oracle flow, matrix")


@pytest.mark.gpu
@pytest.mark.parametrize("triangulator", ["matrix", "ray"])
def test_cli_on_the_six_drone_excerpt(tmp_path, oracle, _built, triangulator):
    out_dir = tmp_path / "results"
    subprocess.run([MAIN, os.path.join(GOLD6, "cameras.xcp"), os.path.join(GOLD6, "dl_data"), "--n_drones", "6",
                    "--triangulator", triangulator, "--output_dir", str(out_dir)], check=True, timeout=600,
                   capture_output=True, text=True)
    got = [np.array([[float(v) for v in l.split()] for l in open(out_dir / f"drone{k + 1}.ply").read().splitlines()[9:]])
           for k in range(6)]
    _check6(_against_truth(got, GOLD6, 6), f"tdr_main, {triangulator}")
    if triangulator == "matrix":  # byte for byte the restated reference flow, on trees of up to 8^8 combinations per frame
        ref, io = _oracle_paths(oracle, gold=GOLD6, n_drones=6, n_frames=150)
        for k in range(6):
            assert open(out_dir / f"drone{k + 1}.ply").read() == io.ply_text(ref[k]), f"drone{k + 1}.ply differs"


GOLD2 = os.path.join(ROOT, "tests", "golden", "s01_d2_a_excerpt")  # first 200 frames of dataset/S01_D2_A: two drones


def _check2(stats, label):
    assert sorted(b for b, _ in stats) == [0, 1], f"{label}: two paths, two different drones"
    med = [float(np.median(d)) for _, d in stats]
    within = [float((d < 150).mean()) for _, d in stats]
    print(f"S01_D2_A excerpt {label}: median distance to the true drone {np.round(med, 1)} mm, within 150 mm {np.round(within, 3)}")
    assert max(med) < 130 and min(len(d) for _, d in stats) >= 195 and min(within) > 0.85


def test_reference_flow_follows_two_drones(oracle):
    paths, _ = _oracle_paths(oracle, gold=GOLD2, n_drones=2, n_frames=200)
    _check2(_against_truth(paths, GOLD2, 2), "oracle flow, matrix")


@pytest.mark.gpu
@pytest.mark.parametrize("triangulator", ["matrix", "ray"])
def test_cli_on_the_two_drone_excerpt(tmp_path, oracle, _built, triangulator):
    out_dir = tmp_path / "results"
    subprocess.run([MAIN, os.path.join(GOLD2, "cameras.xcp"), os.path.join(GOLD2, "dl_data"), "--n_drones", "2",
                    "--triangulator", triangulator, "--output_dir", str(out_dir)], check=True, timeout=600,
                   capture_output=True, text=True)
    got = [np.array([[float(v) for v in l.split()] for l in open(out_dir / f"drone{k + 1}.ply").read().splitlines()[9:]])
           for k in range(2)]
    _check2(_against_truth(got, GOLD2, 2), f"tdr_main, {triangulator}")
    if triangulator == "matrix":
        ref, io = _oracle_paths(oracle, gold=GOLD2, n_drones=2, n_frames=200)
        for k in range(2):
            assert open(out_dir / f"drone{k + 1}.ply").read() == io.ply_text(ref[k]), f"drone{k + 1}.ply differs"

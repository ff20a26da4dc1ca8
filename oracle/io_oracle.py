"""TEST INFRASTRUCTURE ONLY — Python restatement of the reference's file ingest/output used by the
CLI test: loadCamerasXML (src/utils.cpp:46-92), DetectionsContainer::readFiles
(src/DetectionsContainer.cpp:19-92) and writeOutputFile (src/utils.cpp:109-131)."""
import os
import re
import xml.etree.ElementTree as ET

import numpy as np

import tdr_oracle as oracle


def load_cameras_xml(path, vicon_order=False):
    """-> list of oracle camera dicts, in file order; cameras without a <ControlFrame> are skipped."""
    root = ET.parse(path).getroot()
    assert root.tag == "Cameras"
    cams = []
    for cam in root.findall("Camera"):
        cfs = cam.find("ControlFrames")
        cf = cfs.find("ControlFrame") if cfs is not None else None
        if cf is None:
            continue
        pp = [int(re.match(r"\s*[+-]?\d+", t).group(0)) for t in cf.get("PRINCIPAL_POINT").split()]  # >> int
        w, h = pp[0] * 2, pp[1] * 2
        pos = np.array([float(t) for t in cf.get("POSITION").split()])
        q = np.array([float(t) for t in cf.get("ORIENTATION").split()])
        if vicon_order:
            q = np.array([q[3], -q[0], -q[1], -q[2]])
        c = oracle.camera_derive(w, h, float(cf.get("FOCAL_LENGTH")), pos, q)
        c["id"] = int(cam.get("DEVICEID"))
        cams.append(c)
    return cams


def _stoi(tok):
    m = re.match(r"\s*[+-]?\d+", tok)
    if m is None:
        raise ValueError("stoi: invalid_argument")  # std::invalid_argument
    v = int(m.group(0))
    if not -2**31 <= v < 2**31:
        raise OverflowError("stoi: out_of_range")  # std::out_of_range
    return v


def read_detection_csvs(path, offset=0, record_size=7, start_frame=0, end_frame=0):
    """-> (det_count [n_frames,n_cam] int32, dets [total,2] frame-major/camera-major).
    src/DetectionsContainer.cpp:19-76: std::stoi over EVERY field of a line first (:33-38; a malformed field raises
    whether or not the line survives), then the [start_frame, end_frame] window (:40-44, off when both are equal),
    then gap filling and the record-size check."""
    files = []
    for d, _, names in os.walk(path):
        for n in names:
            p = os.path.join(d, n)
            if ".csv" in p and ".avi" not in p:
                files.append(p)
    files.sort()
    data = []
    for name in files:
        frames, frame = [], -1
        for n_line, line in enumerate(open(name).read().split("\n")):
            if line == "" and n_line == len(open(name).read().split("\n")) - 1:
                break
            if offset > n_line:
                continue
            fields = [_stoi(t) for t in line.split(",") if t != ""]
            if (fields[0] <= start_frame or fields[0] > end_frame) and start_frame != end_frame:
                frame = fields[0]
                continue
            for _ in range(fields[0] - frame - 1):
                frames.append([])
            frame = fields[0]
            if (len(fields) - 1) % record_size != 0:
                raise RuntimeError("Invalid CSV file!")
            frames.append([(float(fields[j * record_size + 5]), float(fields[j * record_size + 6]))
                           for j in range(len(fields) // record_size)])
        data.append(frames)
    if len(data) < 2:
        raise RuntimeError("There must be at least 2 cameras")
    if len({len(d) for d in data}) != 1:
        raise RuntimeError("Number of frames on all cameras must be the same")
    n_cam, n_frames = len(data), len(data[0])
    counts = np.array([[len(data[c][f]) for c in range(n_cam)] for f in range(n_frames)], dtype=np.int32)
    dets = [p for f in range(n_frames) for c in range(n_cam) for p in data[c][f]]
    return counts, np.array(dets, dtype=np.float64).reshape(-1, 2)


def ply_text(points):
    """writeOutputFile: default ostream formatting = %g with 6 significant digits."""
    lines = ["ply", "format ascii 1.0", f"element vertex {len(points)}", "property float x", "property float y",
             "property float z", "element face 0", "property list uchar int vertex_index", "end_header"]
    for p in points:
        lines.append(" ".join(format(float(v), ".6g") for v in p))
    return "\n".join(lines) + "\n"


# ----------------------------------------------------------------------------- Vicon evaluation (src/utils.cpp:133-340)
def cross3d(four_points, arms=(9.0, 6.0), err=1000.0):
    """src/utils.cpp:275-317 -> [p0, p_a, p_b, p_c, crossing] or []"""
    p = [np.asarray(q, dtype=np.float64) for q in four_points]
    order = None
    for j in ((1, 2, 3), (2, 1, 3), (3, 1, 2)):
        l1 = np.linalg.norm(p[0] - p[j[0]])
        l2 = np.linalg.norm(p[j[1]] - p[j[2]])
        if (abs(l1 - arms[0]) < err and abs(l2 - arms[1]) < err) or (abs(l1 - arms[1]) < err and abs(l2 - arms[0]) < err):
            order = j
            break
    if order is None:
        return []
    dc = p[order[0]] - p[0]
    dn = p[order[1]] - p[0]
    t = (dc @ dn) / (dc @ dc)
    return [p[0], p[order[0]], p[order[1]], p[order[2]], p[0] + dc * t]


def read_input_csv(path, frequency, n_drones=4, markers_per_drone=5):
    """src/utils.cpp:215-273 (its two hard-coded constants as parameters) -> list of [n,3] label paths."""
    paths = [[] for _ in range(n_drones)]
    i = 0
    for line in open(path).read().split("\n")[:-1] if open(path).read().endswith("\n") else open(path).read().split("\n"):
        if "Drone" in line:
            continue
        if i % frequency != 0:
            i += 1
            continue
        i += 1
        f = [float(t) for t in line.split(",")]
        if len(f) != 1 + n_drones * markers_per_drone * 3:
            raise RuntimeError("Error at CSV file")
        for d in range(n_drones):
            m = [f[markers_per_drone * d * 3 + j * 3 + 1: markers_per_drone * d * 3 + j * 3 + 4] for j in range(markers_per_drone)]
            c = cross3d(m)
            if len(c) == 5:
                paths[d].append(c[4])
    return [np.array(p).reshape(-1, 3) for p in paths]


def path_statistics(label, pred):
    """calculateError / calculateMedian / calculateStd / calcualteQuarterDeviation with the reference's quirks
    (src/utils.cpp:133-213): zero predictions skipped, mean and 'std' (a variance) divide by the unfiltered
    count, median of the errors in frame order, quartile indices from the unfiltered size."""
    size = min(len(label), len(pred))
    label, pred = np.asarray(label)[:size], np.asarray(pred)[:size]
    keep = pred.any(axis=1)
    e = np.linalg.norm(pred - label, axis=1)[keep]
    e = e[~np.isnan(e)]
    mean = e.sum() / size
    median = e[(len(e) - 1) // 2] if len(e) % 2 == 1 else (e[len(e) // 2] + e[len(e) // 2 - 1]) / 2
    var = ((e - mean) ** 2).sum() / size
    es = np.sort(e)
    return mean, median, var, (es[3 * (size // 4)] - es[size // 4]) / 2

"""Synthetic N-camera scenes — host side, numpy only.

Cameras sit on a ring (radius 5000 mm, heights 400..3200 mm) looking at the origin +- 200 mm,
with the R02_D1 sensor (1924 x 1082, FOCAL_LENGTH 1081.0810546875, dataset/R02_D1/R02_D1.xcp:318).
The large detection streams of the benchmark are generated on the device by
tdr_synth_detections_device; the numpy generators here serve small parity cases.
"""
import numpy as np

WIDTH, HEIGHT, FOCAL = 1924, 1082, 1081.0810546875


def rotmat_to_quat_wfirst(R):
    """Unit quaternion (w,x,y,z) whose src/Camera.h:274-287 matrix is R."""
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = [0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s]
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = np.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        q = [(R[2, 1] - R[1, 2]) / s, 0.25 * s, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s]
    elif R[1, 1] > R[2, 2]:
        s = np.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        q = [(R[0, 2] - R[2, 0]) / s, (R[0, 1] + R[1, 0]) / s, 0.25 * s, (R[1, 2] + R[2, 1]) / s]
    else:
        s = np.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        q = [(R[1, 0] - R[0, 1]) / s, (R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, 0.25 * s]
    q = np.array(q, dtype=np.float64)
    return q / np.linalg.norm(q)


def ring_camera_params(n_cam, seed=0):
    """-> list of dicts(width,height,focal,tvec,rquat) ; rquat = camera-to-world, w-first."""
    rng = np.random.default_rng(0xC0FFEE + seed)
    out = []
    for i in range(n_cam):
        ang = 2 * np.pi * (i + 0.25 * rng.uniform(-1, 1)) / n_cam
        C = np.array([5000 * np.cos(ang), 5000 * np.sin(ang), rng.uniform(400, 3200)])
        target = np.array([0.0, 0.0, 1250.0]) + rng.uniform(-200, 200, 3)
        z = target - C
        z /= np.linalg.norm(z)
        up = np.array([0.0, 0.0, -1.0])  # image y grows downwards
        x = np.cross(up, z)
        x /= np.linalg.norm(x)
        y = np.cross(z, x)
        R = np.stack([x, y, z], axis=1)  # columns = camera axes in world coordinates
        out.append(dict(width=WIDTH, height=HEIGHT, focal=FOCAL, tvec=C, rquat=rotmat_to_quat_wfirst(R)))
    return out


def smooth_path(n_sys, seed=0, n_drones=1):
    """Bounded smooth path(s), step <= ~45 mm/frame; system index = frame * n_drones + drone."""
    rng = np.random.default_rng(1000 + seed)
    ph = rng.uniform(0, 2 * np.pi, (n_drones, 3))
    g = np.arange(n_sys)
    f = (g // n_drones).astype(np.float64)
    d = g % n_drones
    X = 1500 * np.sin(0.017 * f + ph[d, 0]) + 300 * np.sin(0.0071 * f + ph[d, 1])
    Y = 1500 * np.sin(0.013 * f + ph[d, 1]) + 300 * np.sin(0.0093 * f + ph[d, 2])
    Z = 1250 + 800 * np.sin(0.011 * f + ph[d, 2])
    return np.stack([X, Y, Z], axis=1)


def project_P(P, X):
    """P [n_cam,12|3x4], X [n,3] -> uv [n_cam,n,2] (exact, un-rounded)."""
    P = np.asarray(P, dtype=np.float64).reshape(-1, 3, 4)
    Xh = np.concatenate([X, np.ones((X.shape[0], 1))], axis=1)
    h = np.einsum("cij,nj->cni", P, Xh)
    return h[..., :2] / h[..., 2:3]


def project_ray_model(aux, X):
    """Inverse of the ray path's pixel->ray model (src/Triangulator.cpp:27-55) for UNIT quaternions.
    aux [n_cam,10], X [n,3] -> uv [n_cam,n,2]."""
    aux = np.asarray(aux, dtype=np.float64).reshape(-1, 10)
    out = np.zeros((aux.shape[0], X.shape[0], 2))
    for c, a in enumerate(aux):
        qw, qx, qy, qz = a[:4]
        R = np.array([[1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qw * qz), 2 * (qx * qz + qw * qy)],
                      [2 * (qx * qy + qw * qz), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qw * qx)],
                      [2 * (qx * qz - qw * qy), 2 * (qy * qz + qw * qx), 1 - 2 * (qx * qx + qy * qy)]])
        fovy, w, h = a[7], a[8], a[9]
        d = 1 / np.tan(fovy * 0.0174533 / 2)
        pc = (X - a[4:7]) @ R  # R^T (X - o)
        xn = pc[:, 0] / pc[:, 2] * d
        yn = pc[:, 1] / pc[:, 2] * d
        out[c, :, 0] = (xn / (w / h) + 1) * w / 2 - 0.5
        out[c, :, 1] = (yn + 1) * h / 2 - 0.5
    return out


def make_detections(uv_exact, seed=0, noise_px=2.0, dropout_p=0.05, round_to_int=True, min_valid=2):
    """Noise, rounding (integer-valued doubles like the CSV data) and (-1,-1) drop-outs, every
    system keeping >= min_valid observations."""
    rng = np.random.default_rng(2000 + seed)
    pts = uv_exact + (rng.normal(0, noise_px, uv_exact.shape) if noise_px > 0 else 0)
    if round_to_int:
        pts = np.rint(pts)
    n_cam, n_sys = pts.shape[:2]
    drop = rng.uniform(size=(n_cam, n_sys)) < dropout_p
    if dropout_p > 0:
        for s in np.nonzero((~drop).sum(axis=0) < min_valid)[0]:
            keep = rng.choice(n_cam, size=min_valid, replace=False)
            drop[:, s] = True
            drop[keep, s] = False
    pts = pts.copy()
    pts[drop] = -1.0
    return np.ascontiguousarray(pts)


def rel_err(p, ref):
    """SURVEY.md §8d gate: max_i |p_i - ref_i|_inf / |ref_i|_inf"""
    num = np.abs(p - ref).max(axis=1)
    den = np.abs(ref).max(axis=1)
    return float((num / den).max())


def multi_drone_frames(P, n_frames, n_drones, seed=0, noise_px=2.0, dropout_p=0.1, false_pos_p=0.15,
                       width=WIDTH, height=HEIGHT):
    """Multi-drone detection frames like DetectionsContainer holds them (data[cam][frame][det]):
    every drone projected into every camera (+noise, rounded), detections of a camera shuffled,
    random drop-outs and false positives.  Returns (det_count [n_frames,n_cam] int32,
    dets [total,2] frame-major/camera-major, truth [n_frames,n_drones,3])."""
    rng = np.random.default_rng(3000 + seed)
    P = np.asarray(P, dtype=np.float64).reshape(-1, 12)
    n_cam = P.shape[0]
    X = smooth_path(n_frames * n_drones, seed=seed, n_drones=n_drones).reshape(n_frames, n_drones, 3)
    counts = np.zeros((n_frames, n_cam), dtype=np.int32)
    dets = []
    for f in range(n_frames):
        uv = project_P(P, X[f])  # [n_cam, n_drones, 2]
        for c in range(n_cam):
            lst = [np.rint(uv[c, d] + rng.normal(0, noise_px, 2)) for d in range(n_drones)
                   if rng.uniform() >= dropout_p]
            if rng.uniform() < false_pos_p:
                lst.append(np.rint([rng.uniform(0, width), rng.uniform(0, height)]))
            rng.shuffle(lst)
            counts[f, c] = len(lst)
            dets.extend(lst)
    dets = np.array(dets, dtype=np.float64).reshape(-1, 2)
    return counts, dets, X

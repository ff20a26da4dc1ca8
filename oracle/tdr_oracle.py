"""TEST INFRASTRUCTURE ONLY — ctypes front-end of oracle/tdr_oracle.c plus an independent
numpy/LAPACK restatement used to pin the C restatement.

PARITY UNPINNED (see tdr_oracle.c header): the reference needs OpenCV and cannot be built
here; it ships no fixtures for this path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product path (the HIP library behind include/tdr_amd.h) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libtdr_oracle.so")

OK, E_SHAPE, E_TOO_FEW, E_CAMERA = 0, -1, -2, -3

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_lp = C.POINTER(C.c_int64)


def build(force=False):
    """gcc build of the C restatement (oracle/Makefile)."""
    src = os.path.join(_HERE, "tdr_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libtdr_oracle.so"])
    return _LIB_PATH


class OracleCamera(C.Structure):
    _fields_ = [
        ("width", C.c_int), ("height", C.c_int), ("cx", C.c_int), ("cy", C.c_int),
        ("fx", C.c_double), ("fy", C.c_double), ("fovx", C.c_double), ("fovy", C.c_double),
        ("rquat", C.c_double * 4), ("tvec", C.c_double * 3), ("campos", C.c_double * 3),
        ("K", C.c_double * 9), ("E", C.c_double * 12), ("P", C.c_double * 12),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        assert _lib.oracle_sizeof_camera() == C.sizeof(OracleCamera)
        _lib.oracle_get_dist_from_ray.restype = C.c_double
    return _lib


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


# ----------------------------------------------------------------------------- reference form guard
# oracle/tdr_oracle.c carries process-global EXPERIMENT switches (tests/tools/*.py flip them to measure what a
# departure from the reference's arithmetic does to the LM runs).  With any of them set the library is no longer the
# restatement of the reference that the parity tests compare the GPU with — so every parity entry point below refuses
# to run unless all of them are at their reference values, or the caller says it is running an
# experiment (allow_experiment=True / allow_experiments(True) in a tool).
_EXPERIMENT_SWITCHES = ("oracle_lm_analytic_jacobian", "oracle_lm_sum_from_squares", "oracle_lm_ldlt_solve",
                        "oracle_lm_hybrid_rule", "oracle_lm_rank2_closed_form", "oracle_lm_h_form",
                        "oracle_jacobi_stop_rule")
_EXPERIMENT_THRESHOLDS = {"oracle_lm_hybrid_tau": 1e-5, "oracle_lm_rank2_min_e2": 1e-10, "oracle_lm_singular_rho": 1e-6}
_allow_experiments = False


def allow_experiments(flag=True):
    """For the measurement scripts under tests/tools/: let the entry points run with experiment switches set."""
    global _allow_experiments
    _allow_experiments = bool(flag)


def experiment_state():
    """{switch: value} of everything that is NOT at its reference value (empty = the reference form)."""
    L = lib()
    off = {n: C.c_int.in_dll(L, n).value for n in _EXPERIMENT_SWITCHES if C.c_int.in_dll(L, n).value != 0}
    off.update({n: C.c_double.in_dll(L, n).value for n, v in _EXPERIMENT_THRESHOLDS.items()
                if C.c_double.in_dll(L, n).value != v})
    return off


def assert_reference_form(allow_experiment=None):
    if allow_experiment if allow_experiment is not None else _allow_experiments:
        return
    off = experiment_state()
    if off:
        raise RuntimeError(f"the oracle is not in its reference form: {off} (an experiment switch was left set; parity "
                           "results against it would mean nothing — pass allow_experiment=True if that is intended)")


# ----------------------------------------------------------------------------- cameras
def camera_derive(width, height, focal_length, tvec, rquat):
    """src/Camera.h:177-187 via src/utils.cpp:94-107. Returns dict with P (3x4), aux (10,) …"""
    cam = OracleCamera()
    _, t = _d(tvec)
    _, q = _d(rquat)
    rc = lib().oracle_camera_derive(int(width), int(height), C.c_double(focal_length), t, q, C.byref(cam))
    if rc != OK:
        raise RuntimeError(f"oracle_camera_derive rc={rc}")
    out = {k: getattr(cam, k) for k in ("width", "height", "cx", "cy", "fx", "fy", "fovx", "fovy")}
    for k, shp in (("rquat", (4,)), ("tvec", (3,)), ("campos", (3,)), ("K", (3, 3)), ("E", (3, 4)), ("P", (3, 4))):
        out[k] = np.array(getattr(cam, k), dtype=np.float64).reshape(shp)
    out["aux"] = np.concatenate([out["rquat"], out["tvec"], [out["fovy"], float(width), float(height)]])
    return out


def cameras_to_arrays(cams):
    """list of camera dicts -> (P [n,12], aux [n,10])"""
    P = np.stack([c["P"].reshape(12) for c in cams]).astype(np.float64)
    aux = np.stack([c["aux"] for c in cams]).astype(np.float64)
    return np.ascontiguousarray(P), np.ascontiguousarray(aux)


# ----------------------------------------------------------------------------- matrix path
def matrix_triangulate_point(P, uv, allow_experiment=None):
    """src/MatrixTriangulator.cpp:3-62.  P [n,12], uv [n,2] -> (X[3], err)"""
    assert_reference_form(allow_experiment)
    P, pP = _d(P)
    uv, puv = _d(uv)
    n = uv.shape[0]
    X = np.zeros(3)
    err = C.c_double()
    rc = lib().oracle_matrix_triangulate_point(pP, puv, n, X.ctypes.data_as(_dp), C.byref(err))
    if rc != OK:
        raise RuntimeError(f"oracle_matrix_triangulate_point rc={rc}")
    return X, err.value


def matrix_triangulate_points(P, pts, want_err=False, n_threads=1, fast=False, allow_experiment=None):
    """src/MatrixTriangulator.cpp:70-100.  P [n_cam,12]; pts [n_cam_pts, n_sys, 2].
    Returns (rc, xyz [n_sys,3], err or None, first_bad)."""
    assert_reference_form(allow_experiment)
    P, pP = _d(P)
    pts, ppts = _d(pts)
    n_cam_pts, n_sys = pts.shape[0], pts.shape[1]
    xyz = np.zeros((n_sys, 3))
    err = np.zeros(n_sys) if want_err else None
    fb = C.c_int64(-1)
    fn = lib().oracle_matrix_triangulate_points_fast if fast else lib().oracle_matrix_triangulate_points_mt
    rc = fn(pP, P.shape[0], ppts, n_cam_pts, C.c_int64(n_sys), xyz.ctypes.data_as(_dp),
            err.ctypes.data_as(_dp) if want_err else None, C.byref(fb), int(n_threads))
    return rc, xyz, err, fb.value


_refcost = None


def refcost_matrix_triangulate_points(P, pts):
    """oracle/ref_cost_model.cpp: the reference's batch loop in its own SHAPE (nested vectors by value, a heap-grown
    vector per point, by-value matrix headers with atomic reference counts, a heap matrix per expression) around the
    arithmetic of this oracle — the cost model behind bench.py's `cpu_baseline.reference_shaped_value`.
    Returns (rc, xyz [n_sys,3], seconds spent inside the by-value triangulatePoints call)."""
    global _refcost
    if _refcost is None:
        lib()
        so = os.path.join(_HERE, "libtdr_refcost.so")
        src = os.path.join(_HERE, "ref_cost_model.cpp")
        if not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(so) < os.path.getmtime(src)):
            subprocess.check_call(["make", "-s", "-C", _HERE, "libtdr_refcost.so"])
        _refcost = C.CDLL(so)
    P, pP = _d(P)
    pts, ppts = _d(pts)
    n_sys = pts.shape[1]
    xyz = np.zeros((n_sys, 3))
    secs = C.c_double(0.0)
    rc = _refcost.oracle_refcost_matrix_triangulate_points(pP, P.shape[0], ppts, pts.shape[0], C.c_int64(n_sys),
                                                           xyz.ctypes.data_as(_dp), C.byref(secs))
    return rc, xyz, secs.value


# Which LM results are comparable between implementations.  The reference's solver can stop away from the
# minimiser of its own objective (iteration cap; step test firing on a heavily damped step); where it stops then
# depends on the last bits of hundreds of branchy iterations (tests/tools/lm_sensitivity.py: a 1-ulp change of one
# ray direction moves such results by up to metres), so the reference has no reproducible answer there.  Both the
# oracle (ray_triangulate_points_ex) and the HIP path (tdr_ray_triangulate_points_ex) report per system
#   LM_CAP            stopped by MAX_ITERATIONS
#   LM_NOT_CONVERGED  |x - x*|_inf > 5e-8 max(|x*|_inf, 1), x* = exact minimiser of the quadratic objective
# and a system is compared iff BOTH sides report 0.  (Round 1 used an iteration-count threshold tuned to the
# observed failures; it is gone.)
LM_FLAG_MASK = 3


def lm_comparable(status, other_status=None):
    """Boolean mask: systems whose LM run ended at the minimiser according to `status` (and `other_status`)."""
    ok = (np.asarray(status) & LM_FLAG_MASK) == 0
    if other_status is not None:
        ok &= (np.asarray(other_status) & LM_FLAG_MASK) == 0
    return ok


# ----------------------------------------------------------------------------- ray path
def create_ray(aux, u, v):
    """src/Triangulator.cpp:27-55 -> (origin[3], dir[3])"""
    aux, pa = _d(aux)
    o, d = np.zeros(3), np.zeros(3)
    lib().oracle_create_ray(pa, C.c_double(u), C.c_double(v), o.ctypes.data_as(_dp), d.ctypes.data_as(_dp))
    return o, d


def get_dist_from_ray(aux, u, v, point):
    """src/Triangulator.cpp:57-61"""
    aux, pa = _d(aux)
    point, pp = _d(point)
    return lib().oracle_get_dist_from_ray(pa, C.c_double(u), C.c_double(v), pp)


def ray_triangulate_point(aux, uv, allow_experiment=None):
    """src/RayTriangulator.cpp:83-107 -> (X[3], err, iters)"""
    assert_reference_form(allow_experiment)
    aux, pa = _d(aux)
    uv, puv = _d(uv)
    X = np.zeros(3)
    err, it = C.c_double(), C.c_int()
    rc = lib().oracle_ray_triangulate_point(pa, puv, uv.shape[0], X.ctypes.data_as(_dp), C.byref(err), C.byref(it))
    if rc != OK:
        raise RuntimeError(f"oracle_ray_triangulate_point rc={rc}")
    return X, err.value, it.value


def ray_closed_form(aux, uv):
    aux, pa = _d(aux)
    uv, puv = _d(uv)
    X = np.zeros(3)
    rc = lib().oracle_ray_closed_form(pa, puv, uv.shape[0], X.ctypes.data_as(_dp))
    if rc != OK:
        raise RuntimeError(f"oracle_ray_closed_form rc={rc}")
    return X


def ray_closest_point(aux, uv, allow_experiment=None):
    """oracle_ray_closest_point: closed-form minimiser from the reference's rays, x0 - pinv_EIG(H)(H x0 - g), and
    the mean ray distance there -> (X[3], err)."""
    assert_reference_form(allow_experiment)
    aux, pa = _d(aux)
    uv, puv = _d(uv)
    X = np.zeros(3)
    err = C.c_double()
    rc = lib().oracle_ray_closest_point(pa, puv, uv.shape[0], X.ctypes.data_as(_dp), C.byref(err))
    if rc != OK:
        raise RuntimeError(f"oracle_ray_closest_point rc={rc}")
    return X, err.value


def ray_closest_point_points(aux, pts, want_err=False, allow_experiment=None):
    """Batch form with the shapes / too-few rule of src/RayTriangulator.cpp:51-81 (checker of
    tdr_ray_closest_point_points).  Returns (rc, xyz, err|None, first_bad)."""
    assert_reference_form(allow_experiment)
    aux, pa = _d(aux)
    pts, ppts = _d(pts)
    n_cam_pts, n_sys = pts.shape[0], pts.shape[1]
    xyz = np.zeros((n_sys, 3))
    err = np.zeros(n_sys) if want_err else None
    fb = C.c_int64(-1)
    rc = lib().oracle_ray_closest_point_points(pa, aux.shape[0], ppts, n_cam_pts, C.c_int64(n_sys),
                                               xyz.ctypes.data_as(_dp),
                                               err.ctypes.data_as(_dp) if want_err else None, C.byref(fb))
    return rc, xyz, err, fb.value


def ray_triangulate_points(aux, pts, want_err=False, want_iters=False, n_threads=1, allow_experiment=None):
    """src/RayTriangulator.cpp:51-81.  aux [n_cam,10]; pts [n_cam_pts,n_sys,2].
    Returns (rc, xyz, err|None, iters|None, first_bad)."""
    assert_reference_form(allow_experiment)
    aux, pa = _d(aux)
    pts, ppts = _d(pts)
    n_cam_pts, n_sys = pts.shape[0], pts.shape[1]
    xyz = np.zeros((n_sys, 3))
    err = np.zeros(n_sys) if want_err else None
    iters = np.zeros(n_sys, dtype=np.int32) if want_iters else None
    fb = C.c_int64(-1)
    rc = lib().oracle_ray_triangulate_points_mt(
        pa, aux.shape[0], ppts, n_cam_pts, C.c_int64(n_sys), xyz.ctypes.data_as(_dp),
        err.ctypes.data_as(_dp) if want_err else None,
        iters.ctypes.data_as(_ip) if want_iters else None, C.byref(fb), int(n_threads))
    return rc, xyz, err, iters, fb.value


LM_CAP, LM_NOT_CONVERGED = 1, 2  # ORACLE_LM_* == TDR_LM_* status bits (tdr_oracle.c, include/tdr_amd.h)
LM_SINGULAR_TRIAL, LM_SINGULAR_ACCEPT = 16, 32  # oracle-only diagnostics


def ray_triangulate_points_ex(aux, pts, perturb_ulps=0, n_threads=1, allow_experiment=None):
    """ray_triangulate_points plus the per-system LM diagnostics: returns a dict with xyz, err, iters,
    status (LM_* bits), rho (min over the lambda == 0 trials of det A / (max cof_kk tr A), 1 if none) and
    dist (|x - x*|_inf: distance of the returned point from the minimiser of the solver's objective).
    perturb_ulps moves the x component of every system's first ray direction by that many ulps."""
    assert_reference_form(allow_experiment)
    aux, pa = _d(aux)
    pts, ppts = _d(pts)
    n_cam_pts, n_sys = pts.shape[0], pts.shape[1]
    xyz = np.zeros((n_sys, 3))
    err = np.zeros(n_sys)
    iters = np.zeros(n_sys, dtype=np.int32)
    status = np.zeros(n_sys, dtype=np.int32)
    rho = np.zeros(n_sys)
    dist = np.zeros(n_sys)
    fb = C.c_int64(-1)
    rc = lib().oracle_ray_triangulate_points_ex(
        pa, aux.shape[0], ppts, n_cam_pts, C.c_int64(n_sys), xyz.ctypes.data_as(_dp), err.ctypes.data_as(_dp),
        iters.ctypes.data_as(_ip), status.ctypes.data_as(_ip), rho.ctypes.data_as(_dp), dist.ctypes.data_as(_dp), int(perturb_ulps),
        C.byref(fb), int(n_threads))
    return {"rc": rc, "xyz": xyz, "err": err, "iters": iters, "status": status, "rho": rho, "dist": dist,
            "first_bad": fb.value}


def set_lm_singular_rho(v):
    C.c_double.in_dll(lib(), "oracle_lm_singular_rho").value = float(v)


# ============================================================================= numpy / LAPACK
# Independent restatement (no shared code with the C file) used to pin it.

def np_camera_derive(width, height, focal_length, tvec, rquat):
    """Same chain as camera_derive, written with numpy (numpy.linalg.inv instead of adjugate)."""
    RAD_TO_DEG = 57.29577951308232087679
    DEG_TO_RAD = 0.01745329251994329576
    w, h = int(width), int(height)
    cx = int(np.floor(w / 2.0 + 0.5))
    cy = int(np.floor(h / 2.0 + 0.5))
    fovx = 2 * np.arctan(w / (2 * focal_length)) * 57.2958
    fovy = 2.0 * np.arctan(np.tan(fovx * 0.5 * DEG_TO_RAD) / (float(w) / float(h))) * RAD_TO_DEG
    fx = (w / 2.0) / np.tan((fovx / 2.0) * DEG_TO_RAD)
    fy = (h / 2.0) / np.tan((fovy / 2.0) * DEG_TO_RAD)
    a, b, c, d = [float(x) for x in rquat]
    R = np.array([[1 - 2 * (c * c + d * d), 2 * (b * c - a * d), 2 * (b * d + a * c)],
                  [2 * (b * c + a * d), 1 - 2 * (b * b + d * d), 2 * (c * d - a * b)],
                  [2 * (b * d - a * c), 2 * (c * d + a * b), 1 - 2 * (b * b + c * c)]])
    Rinv = np.linalg.inv(R)
    campos = -Rinv @ np.asarray(tvec, dtype=np.float64)
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
    E = np.hstack([Rinv, campos.reshape(3, 1)])
    return {"cx": cx, "cy": cy, "fovx": fovx, "fovy": fovy, "fx": fx, "fy": fy, "campos": campos,
            "K": K, "E": E, "P": K @ E}


def np_build_system(P, uv):
    """A (2n x 3), b (2n) of src/MatrixTriangulator.cpp:16-49."""
    P = np.asarray(P, dtype=np.float64).reshape(-1, 3, 4)
    uv = np.asarray(uv, dtype=np.float64)
    n = uv.shape[0]
    A = np.zeros((2 * n, 3))
    b = np.zeros(2 * n)
    A[0::2] = P[:, 0, :3] - uv[:, 0:1] * P[:, 2, :3]
    A[1::2] = P[:, 1, :3] - uv[:, 1:2] * P[:, 2, :3]
    b[0::2] = uv[:, 0] * P[:, 2, 3] - P[:, 0, 3]
    b[1::2] = uv[:, 1] * P[:, 2, 3] - P[:, 1, 3]
    return A, b


def np_matrix_triangulate_point(P, uv):
    """LAPACK pinv (gesdd) solve of the same system."""
    A, b = np_build_system(P, uv)
    X = np.linalg.pinv(A) @ b
    r = A @ X - b
    return X, float(np.sqrt(r @ r / A.shape[0]))


def np_matrix_triangulate_points(P, pts):
    """Batch semantics of src/MatrixTriangulator.cpp:70-100 with LAPACK lstsq per system."""
    P = np.asarray(P, dtype=np.float64).reshape(-1, 12)
    pts = np.asarray(pts, dtype=np.float64)
    nc = min(P.shape[0], pts.shape[0])
    n_sys = pts.shape[1]
    xyz = np.zeros((n_sys, 3))
    err = np.zeros(n_sys)
    for s in range(n_sys):
        valid = [c for c in range(nc) if not (pts[c, s, 0] == -1 or pts[c, s, 1] == -1)]
        if len(valid) < 2:
            return E_TOO_FEW, xyz, err, s
        xyz[s], err[s] = np_matrix_triangulate_point(P[valid], pts[valid, s])
    return OK, xyz, err, -1


def np_create_ray(aux, u, v):
    q = np.asarray(aux[:4], dtype=np.float64)
    fovy, width, height = aux[7], aux[8], aux[9]
    px, py = u + 0.5, v + 0.5
    d = 1 / np.tan(fovy * 0.0174533 / 2)
    x = (width / height) * ((2 * px / width) - 1)
    y = (2 * py / height) - 1
    vec = np.array([x, y, d])
    vec = vec * (1.0 / np.linalg.norm(vec))

    def qmul(a, b):
        return np.array([
            a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
            a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
            a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
            a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]])

    conj = q * np.array([1, -1, -1, -1.0])
    r = qmul(qmul(q, np.concatenate([[0.0], vec])), conj)
    return np.asarray(aux[4:7], dtype=np.float64).copy(), r[1:4]


def np_ray_closed_form(aux, uv):
    """argmin sum |d x (x-o)|^2 via numpy.linalg.solve."""
    M = np.zeros((3, 3))
    rhs = np.zeros(3)
    for a, (u, v) in zip(aux, uv):
        o, d = np_create_ray(a, u, v)
        m = (d @ d) * np.eye(3) - np.outer(d, d)
        M += m
        rhs += m @ o
    return np.linalg.solve(M, rhs)


def np_rays_batch(aux, pts):
    """np_create_ray for a whole [n_cam, n_sys, 2] detection block at once (same formulas, vectorised; the sentinel
    (-1) observations get a ray too — mask them with the returned `valid`).  -> (origins [n_cam,3], dirs
    [n_cam,n_sys,3], valid [n_cam,n_sys])."""
    aux = np.asarray(aux, dtype=np.float64)
    pts = np.asarray(pts, dtype=np.float64)
    q = aux[:, None, :4]
    fovy, width, height = aux[:, None, 7], aux[:, None, 8], aux[:, None, 9]
    px, py = pts[..., 0] + 0.5, pts[..., 1] + 0.5
    d = 1 / np.tan(fovy * 0.0174533 / 2) + 0 * px
    x = (width / height) * ((2 * px / width) - 1)
    y = (2 * py / height) - 1
    vec = np.stack([x, y, d], axis=-1)
    vec = vec * (1.0 / np.linalg.norm(vec, axis=-1))[..., None]

    def qmul(a, b):
        return np.stack([
            a[..., 0] * b[..., 0] - a[..., 1] * b[..., 1] - a[..., 2] * b[..., 2] - a[..., 3] * b[..., 3],
            a[..., 0] * b[..., 1] + a[..., 1] * b[..., 0] + a[..., 2] * b[..., 3] - a[..., 3] * b[..., 2],
            a[..., 0] * b[..., 2] - a[..., 1] * b[..., 3] + a[..., 2] * b[..., 0] + a[..., 3] * b[..., 1],
            a[..., 0] * b[..., 3] + a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1] + a[..., 3] * b[..., 0]], axis=-1)

    v4 = np.concatenate([np.zeros(vec.shape[:-1] + (1,)), vec], axis=-1)
    qq = np.broadcast_to(q, v4.shape)
    r = qmul(qmul(qq, v4), qq * np.array([1, -1, -1, -1.0]))
    valid = (pts[..., 0] != -1) & (pts[..., 1] != -1)
    return aux[:, 4:7].copy(), r[..., 1:4], valid


def np_ray_closed_form_batch(aux, pts):
    """np_ray_closed_form for every system of a block (numpy.linalg.solve on the stacked 3x3 systems) plus what a
    comparison needs to be honest about: cond_2(H) per system.  The normal matrix H = sum (|d|^2 I - d d^T) of rays
    that are nearly (anti)parallel is ill-conditioned, and ANY fp64 evaluation of H^-1 g — this one included — carries
    a relative error of a few eps x cond(H).  -> (xyz [n_sys,3], mean ray distance [n_sys], cond [n_sys])."""
    o, d, valid = np_rays_batch(aux, pts)
    w = valid[..., None, None].astype(np.float64)
    dd = (d * d).sum(axis=-1)
    Hi = (dd[..., None, None] * np.eye(3) - d[..., :, None] * d[..., None, :]) * w
    H = Hi.sum(axis=0)
    g = (Hi @ o[:, None, :, None])[..., 0].sum(axis=0)
    xyz = np.linalg.solve(H, g[..., None])[..., 0]
    ev = np.linalg.eigvalsh(H)
    dist = np.linalg.norm(np.cross(d, xyz[None] - o[:, None, :]), axis=-1) * valid
    return xyz, dist.sum(axis=0) / valid.sum(axis=0), ev[:, -1] / ev[:, 0]


def np_eig_pinv(M):
    """cv::solve / cv::invert with DECOMP_EIG on a symmetric matrix, through LAPACK (numpy.linalg.eigh) instead of
    the C file's hand-written Jacobi: V diag(1/w) V^T with the eigenvalues at rounding level dropped.  OpenCV cuts at
    |w_i| <= 2 DBL_EPSILON sum|w|, which its Jacobi eigenvalues (high RELATIVE accuracy on a positive semi-definite
    matrix: a zero eigenvalue comes out well below that) respect; LAPACK's tridiagonal solver resolves a zero
    eigenvalue only to a few eps ||M|| — above OpenCV's cut about one time in three — so this stand-in cuts at
    32 eps sum|w|.  The two agree wherever the matrix is not within rounding of singular-but-kept."""
    w, V = np.linalg.eigh(M)
    thr = 32 * np.finfo(np.float64).eps * np.abs(w).sum()
    inv = np.zeros_like(M)
    for i in range(len(w)):
        if abs(w[i]) > thr:
            inv += np.outer(V[:, i], V[:, i]) / w[i]
    return inv


def np_ray_lm(aux, uv, max_iters=1000):
    """RayTriangulator::triangulatePoint (src/RayTriangulator.cpp:83-107) with cv::LMSolver::run restated a SECOND
    time, in numpy, sharing no code with tdr_oracle.c: central-difference Jacobian (eps = 1e-4), the levmarq.cpp
    lambda schedule, DECOMP_EIG through np_eig_pinv.  Returns (x, iterations).  Independent pin of the C restatement's
    loop and of its eigen solve (tests/test_oracle_cpu.py)."""
    rays = [np_create_ray(a, u, v) for a, (u, v) in zip(aux, uv)]
    O = np.array([r[0] for r in rays])
    D = np.array([r[1] for r in rays])

    def dist(x):
        return np.linalg.norm(np.cross(D, x[None, :] - O), axis=1)

    def jac(x):
        e = 1e-4
        J = np.zeros((len(rays), 3))
        for k in range(3):
            dx = np.zeros(3)
            dx[k] = e
            J[:, k] = (dist(x + dx) - dist(x - dx)) / (2 * e)
        return J

    eps, feps = np.finfo(np.float64).eps, float(np.finfo(np.float32).eps)
    x = O.mean(axis=0)
    r, J = dist(x), jac(x)
    S = float(r @ r)
    A, v = J.T @ J, J.T @ r
    Dg = np.diag(A).copy()
    lam, lc, it = 1.0, 0.75, 0
    while True:
        d = np_eig_pinv(A + lam * np.diag(Dg)) @ v
        xd = x - d
        rd = dist(xd)
        Sd = float(rd @ rd)
        dS = float(d @ (2 * v - A @ d))
        R = (S - Sd) / (dS if abs(dS) > eps else 1.0)
        if R > 0.75:
            lam *= 0.5
            if lam < lc:
                lam = 0.0
        elif R < 0.25:
            t = float(d @ v)
            nu = min(max((Sd - S) / (t if abs(t) > eps else 1.0) + 2, 2.0), 10.0)
            if lam == 0:
                lam = lc = 1.0 / max(eps, float(np.abs(np.diag(np_eig_pinv(A))).max()))
                nu *= 0.5
            lam *= nu
        if Sd < S:
            S, x = Sd, xd
            r, J = dist(x), jac(x)
            A, v = J.T @ J, J.T @ r
        it += 1
        if not (it < max_iters and np.abs(d).max() >= feps and np.abs(r).max() >= feps):
            return x, it


# ----------------------------------------------------------------------------- candidate scoring
def fill_combination_queue(cam_const, det_count, dets_per_cam, error_max=1e5, min_cameras=2, mode=0):
    """DroneClassifier::fillCombinationQueue (src/DroneClassifier.cpp:156-198) for ONE frame.
    cam_const: P [n_cam,12] (mode 0) or aux [n_cam,10] (mode 1); dets_per_cam: list of [k_c,2] arrays.
    Returns (comb [n,n_cam] int32, xyz [n,3], err [n], n_evals) in push order."""
    cam_const, pc = _d(cam_const)
    n_cam = cam_const.shape[0]
    det_count = np.ascontiguousarray(det_count, dtype=np.int32)
    flat = np.concatenate([np.asarray(d, dtype=np.float64).reshape(-1, 2) for d in dets_per_cam] +
                          [np.zeros((1, 2))])
    flat, pf = _d(flat)
    off = np.zeros(n_cam, dtype=np.int64)
    off[1:] = np.cumsum(det_count)[:-1]
    cap = int(np.prod(det_count.astype(np.int64) + 1))
    comb = np.zeros((cap, n_cam), dtype=np.int32)
    xyz = np.zeros((cap, 3))
    err = np.zeros(cap)
    ne = C.c_int64(0)
    fn = lib().oracle_fill_combination_queue
    fn.restype = C.c_int64
    n = fn(pc, int(mode), n_cam, det_count.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p), pf,
           C.c_double(error_max), int(min_cameras), C.c_int64(cap), comb.ctypes.data_as(C.c_void_p),
           xyz.ctypes.data_as(_dp), err.ctypes.data_as(_dp), C.byref(ne))
    return comb[:n], xyz[:n], err[:n], ne.value


def fill_combination_queue_ex(cam_const, det_count, dets_per_cam, error_max=120.0, min_cameras=2, mode=1):
    """fill_combination_queue with the LM status tracked along the walk (oracle_fill_combination_queue_ex):
    returns (comb, xyz, err, flagged [n] bool, cuts [m, n_cam] int32 with -1 beyond the cut node's depth)."""
    cam_const, pc = _d(cam_const)
    n_cam = cam_const.shape[0]
    det_count = np.ascontiguousarray(det_count, dtype=np.int32)
    flat = np.concatenate([np.asarray(d, dtype=np.float64).reshape(-1, 2) for d in dets_per_cam] + [np.zeros((1, 2))])
    flat, pf = _d(flat)
    off = np.zeros(n_cam, dtype=np.int64)
    off[1:] = np.cumsum(det_count)[:-1]
    cap = int(np.prod(det_count.astype(np.int64) + 1))
    comb = np.zeros((cap, n_cam), dtype=np.int32)
    xyz, err = np.zeros((cap, 3)), np.zeros(cap)
    flagged = np.zeros(cap, dtype=np.int32)
    cuts = np.zeros((cap, n_cam), dtype=np.int32)
    ne, nc = C.c_int64(0), C.c_int64(0)
    fn = lib().oracle_fill_combination_queue_ex
    fn.restype = C.c_int64
    n = fn(pc, int(mode), n_cam, det_count.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p), pf,
           C.c_double(error_max), int(min_cameras), C.c_int64(cap), comb.ctypes.data_as(C.c_void_p),
           xyz.ctypes.data_as(_dp), err.ctypes.data_as(_dp), C.byref(ne), flagged.ctypes.data_as(C.c_void_p),
           C.c_int64(cap), cuts.ctypes.data_as(C.c_void_p), C.byref(nc))
    return comb[:n], xyz[:n], err[:n], flagged[:n].astype(bool), cuts[:nc.value]

"""ctypes binding of libtdr_amd.so (include/tdr_amd.h) + a Python mirror of the reference's
`Triangulator` plugin surface (src/Triangulator.h:9-59) for tests and bench.py.

This is plumbing over the C ABI, not a second implementation: every solve is a HIP kernel launch
inside libtdr_amd.so.  There is no CPU fallback — if the library is missing or no gfx950 device
is usable, loading / construction raises.

The directory name is not a Python identifier; import it with
    importlib.import_module("3d-reconstruction-triangulation_amd")
(tests/conftest.py and bench.py do exactly that).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TDR_LIB_PATH") or os.path.join(_HERE, "libtdr_amd.so")  # env override: A/B builds only

TDR_OK, TDR_E_SHAPE, TDR_E_TOO_FEW, TDR_E_CAMERA = 0, -1, -2, -3
TDR_E_HIP, TDR_E_NO_DEVICE, TDR_E_RCCL, TDR_E_ARG = -4, -5, -6, -7
TDR_MAX_CAMERAS = 32
TDR_LM_CAP, TDR_LM_NOT_CONVERGED, TDR_LM_TOO_FEW = 1, 2, 4  # LM status bits (include/tdr_amd.h)

_dp = C.POINTER(C.c_double)
_vp = C.c_void_p
_i64 = C.c_int64

# every symbol include/tdr_amd.h declares: (restype, argtypes)
C_API = {
    "tdr_strerror": (C.c_char_p, [C.c_int]),
    "tdr_last_error": (C.c_char_p, []),
    "tdr_camera_derive": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_double, _dp, _dp, _vp]),
    "tdr_camera_pack": (C.c_int, [_vp, C.c_int, _dp, _dp]),
    "tdr_create": (C.c_int, [_dp, _dp, C.c_int, C.c_int, C.POINTER(_vp)]),
    "tdr_create_multi": (C.c_int, [_dp, _dp, C.c_int, _vp, C.c_int, C.POINTER(_vp)]),
    "tdr_destroy": (None, [_vp]),
    "tdr_multi_shard_count": (C.c_int, [_vp]),
    "tdr_multi_shard": (C.c_int, [_vp, C.c_int, C.POINTER(_vp)]),
    "tdr_multi_matrix_triangulate_gather": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp, _vp, _vp, C.c_int,
                                                      C.POINTER(_i64)]),
    "tdr_n_cameras": (C.c_int, [_vp]),
    "tdr_device_count": (C.c_int, []),
    "tdr_matrix_triangulate_points": (C.c_int, [_vp, _dp, C.c_int, _i64, _dp, _dp, C.POINTER(_i64)]),
    "tdr_ray_triangulate_points": (C.c_int, [_vp, _dp, C.c_int, _i64, _dp, _dp, C.POINTER(_i64)]),
    "tdr_ray_triangulate_points_ex": (C.c_int, [_vp, _dp, C.c_int, _i64, _dp, _dp, _vp, _vp, C.POINTER(_i64)]),
    "tdr_ray_triangulate_points_device_ex": (C.c_int, [_vp, _vp, C.c_int, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "tdr_matrix_triangulate_points_rows": (C.c_int, [_vp, _vp, C.c_int, _i64, _dp, _dp, C.POINTER(_i64)]),
    "tdr_ray_triangulate_points_rows": (C.c_int, [_vp, _vp, C.c_int, _i64, _dp, _dp, C.POINTER(_i64)]),
    "tdr_matrix_triangulate_point": (C.c_int, [_vp, _dp, _dp, C.c_int, _dp, _dp]),
    "tdr_ray_triangulate_point": (C.c_int, [_vp, _dp, _dp, C.c_int, _dp, _dp]),
    "tdr_matrix_triangulate_points_device": (C.c_int, [_vp, _vp, C.c_int, _i64, _i64, _vp, _vp, _vp]),
    "tdr_ray_triangulate_points_device": (C.c_int, [_vp, _vp, C.c_int, _i64, _i64, _vp, _vp, _vp]),
    "tdr_ray_closest_point_points": (C.c_int, [_vp, _dp, C.c_int, _i64, _dp, _dp, C.POINTER(_i64)]),
    "tdr_ray_closest_point_points_device": (C.c_int, [_vp, _vp, C.c_int, _i64, _i64, _vp, _vp, _vp]),
    "tdr_ray_closest_point_points_rows": (C.c_int, [_vp, _vp, C.c_int, _i64, _dp, _dp, C.POINTER(_i64)]),
    "tdr_create_ray": (C.c_int, [_dp, _dp, _dp, _dp]),
    "tdr_ray_direction_for_pixel": (C.c_int, [_dp, _dp, _dp]),
    "tdr_rotate_point_by_quaternion": (C.c_int, [_dp, _dp, _dp]),
    "tdr_dist_to_ray": (C.c_double, [_dp, _dp, _dp, C.c_int]),
    "tdr_pack_detections_i16_host": (C.c_int, [_dp, _i64, _vp, C.POINTER(C.c_int)]),
    "tdr_pack_detections_i16_device": (C.c_int, [_vp, _vp, C.c_int, _i64, _i64, _vp, _i64, _vp]),
    "tdr_matrix_triangulate_points_i16_device": (C.c_int, [_vp, _vp, C.c_int, _i64, _i64, _vp, _vp, _vp]),
    "tdr_matrix_triangulate_points_i16": (C.c_int, [_vp, _vp, C.c_int, _i64, _dp, _dp, C.POINTER(_i64)]),
    "tdr_time_fp64_fma": (C.c_int, [_vp, C.c_int, C.c_int, _dp, _dp]),
    "tdr_debug_set_knob": (C.c_int, [C.c_char_p, C.c_int, C.c_int]),
    "tdr_debug_sym3_eig_device": (C.c_int, [_vp, _vp, _i64, _vp, _vp]),
    "tdr_debug_lm_cascade_device": (C.c_int, [_vp, _vp, _i64, C.c_int, _vp, _vp]),
    "tdr_first_bad": (C.c_int, [_vp, _vp, C.POINTER(_i64)]),
    "tdr_matrix_score_combinations": (C.c_int, [_vp, C.c_int, _vp, _dp, C.c_double, C.c_int, _vp, _i64,
                                                C.POINTER(_i64)]),
    "tdr_ray_score_combinations": (C.c_int, [_vp, C.c_int, _vp, _dp, C.c_double, C.c_int, _vp, _i64,
                                             C.POINTER(_i64)]),
    "tdr_score_combinations_levelwise": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _dp, C.c_double, C.c_int, _vp,
                                                   _i64, C.POINTER(_i64)]),
    "tdr_fetch_scored_combinations": (C.c_int, [_vp, _vp, _i64, C.POINTER(_i64)]),
    "tdr_ray_score_combinations_ex": (C.c_int, [_vp, C.c_int, _vp, _dp, C.c_double, C.c_int, _vp, _i64,
                                                C.POINTER(_i64), _vp, _i64, C.POINTER(_i64)]),
    "tdr_decode_combination": (C.c_int, [C.c_int, _vp, _i64, _vp]),
    "tdr_set_matrix_variant": (C.c_int, [_vp, C.c_int]),
    "tdr_set_ray_jacobian": (C.c_int, [_vp, C.c_int]),
    "tdr_set_host_pipeline": (C.c_int, [_vp, C.c_int, _i64]),
    "tdr_dist_from_ray_device": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "tdr_dist_from_ray_batch": (C.c_int, [_vp, _vp, _dp, _dp, _i64, _dp]),
    "tdr_dist_from_ray": (C.c_int, [_vp, _dp, _dp, _dp, _dp]),
    "tdr_synth_detections_device": (C.c_int, [_vp, C.c_uint64, C.c_int, _i64, _i64, _i64, C.c_double,
                                              C.c_double, C.c_int, C.c_int, _vp, _vp, _vp]),
    "tdr_device_alloc": (C.c_int, [_vp, _i64, C.POINTER(_vp)]),
    "tdr_device_free": (C.c_int, [_vp, _vp]),
    "tdr_memcpy_h2d": (C.c_int, [_vp, _vp, _vp, _i64]),
    "tdr_memcpy_d2h": (C.c_int, [_vp, _vp, _vp, _i64]),
    "tdr_stream_synchronize": (C.c_int, [_vp, _vp]),
    "tdr_host_alloc": (C.c_int, [_i64, C.POINTER(_vp)]),
    "tdr_host_free": (C.c_int, [_vp]),
    "tdr_time_stream": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, _i64, C.c_int, C.c_int, _dp]),
    "tdr_time_launches": (C.c_int, [_vp, C.c_int, _vp, C.c_int, _i64, _i64, _vp, _vp, C.c_int, C.c_int,
                                    _vp, _dp]),
}


CANDIDATE_DTYPE = np.dtype([("frame", "<i4"), ("n_used", "<i2"), ("flags", "<i2"), ("leaf", "<i8"), ("x", "<f8"),
                            ("y", "<f8"), ("z", "<f8"), ("err", "<f8")])  # struct tdr_candidate, 48 B
TDR_CANDIDATE_FLAGGED, TDR_CANDIDATE_FLAGGED_CUT = 1, 2


def decode_combination(det_count_frame, leaf):
    """digits[c]: 0 = no detection, k = detection k-1 (tdr_decode_combination)."""
    dc = np.ascontiguousarray(det_count_frame, dtype=np.int32)
    digits = np.zeros(dc.shape[0], dtype=np.int32)
    _check(lib().tdr_decode_combination(dc.shape[0], dc.ctypes.data, int(leaf), digits.ctypes.data))
    return digits


class TdrCamera(C.Structure):
    """struct tdr_camera (include/tdr_amd.h)"""
    _fields_ = [
        ("id", C.c_int32), ("width", C.c_int32), ("height", C.c_int32), ("cx", C.c_int32),
        ("cy", C.c_int32), ("diagonal", C.c_int32),
        ("fx", C.c_double), ("fy", C.c_double), ("fovx", C.c_double), ("fovy", C.c_double),
        ("rquat", C.c_double * 4), ("tvec", C.c_double * 3), ("rvec", C.c_double * 3),
        ("campos", C.c_double * 3), ("K", C.c_double * 9), ("E", C.c_double * 12), ("P", C.c_double * 12),
    ]


_lib = None


def lib():
    """Load libtdr_amd.so; raise (never fall back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in C_API.items():
            fn = getattr(handle, name)  # AttributeError if the ABI symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


class TdrError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(message)
        self.status = status


def _check(rc, too_few_msg="Too few rays are found"):
    if rc == TDR_OK:
        return
    L = lib()
    if rc == TDR_E_TOO_FEW:
        raise TdrError(rc, too_few_msg)
    msg = L.tdr_strerror(rc).decode()
    if rc in (TDR_E_HIP, TDR_E_NO_DEVICE):
        msg += ": " + L.tdr_last_error().decode()
    raise TdrError(rc, msg)


def set_knob(name, value=None):
    """Execution-form / tuning knob (tdr_debug_set_knob): `value=None` restores the built-in default.  The library
    reads the TDR_* environment variables once per process; tests switch forms through this call."""
    _check(lib().tdr_debug_set_knob(name.encode(), 0 if value is None else int(value), 1 if value is None else 0))


def _as_d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


# ------------------------------------------------------------------------------------------ cameras
class Camera:
    """Mirror of tdr::Camera as built by createCamera (src/utils.cpp:94-107): public fields
    width,height,cx,cy,fx,fy,fovx,fovy,rquat,tvec,camPos,cameraMatrix,cameraExtrinsicMatrix,
    cameraPerspectiveMatrix, derived by the library's tdr_camera_derive."""

    def __init__(self, cam_id, width, height, focal_length, tvec, rquat):
        c = TdrCamera()
        _, t = _as_d(tvec)
        _, q = _as_d(rquat)
        _check(lib().tdr_camera_derive(int(cam_id), int(width), int(height), float(focal_length), t, q,
                                       C.byref(c)))
        self._c = c
        for k in ("id", "width", "height", "cx", "cy", "diagonal", "fx", "fy", "fovx", "fovy"):
            setattr(self, k, getattr(c, k))
        self.rquat = np.array(c.rquat)
        self.tvec = np.array(c.tvec)
        self.rvec = np.array(c.rvec)
        self.camPos = np.array(c.campos)
        self.cameraMatrix = np.array(c.K).reshape(3, 3)
        self.cameraExtrinsicMatrix = np.array(c.E).reshape(3, 4)
        self.cameraPerspectiveMatrix = np.array(c.P).reshape(3, 4)

    def getCamId(self):
        return self.id

    @property
    def aux(self):
        return np.concatenate([self.rquat, self.tvec, [self.fovy, float(self.width), float(self.height)]])


def createCamera(cam_id, width, height, focalLength, translation, rotation):
    """src/utils.cpp:94-107"""
    return Camera(cam_id, width, height, focalLength, translation, rotation)


def pack_cameras(cameras):
    P = np.ascontiguousarray(np.stack([c.cameraPerspectiveMatrix.reshape(12) for c in cameras]))
    aux = np.ascontiguousarray(np.stack([c.aux for c in cameras]))
    return P, aux


# ------------------------------------------------------------------------------------------ engine
class DeviceBuffer:
    def __init__(self, engine, nbytes):
        self.engine = engine
        self.nbytes = int(nbytes)
        p = _vp()
        _check(lib().tdr_device_alloc(engine._ctx, self.nbytes, C.byref(p)))
        self.ptr = p.value

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        _check(lib().tdr_memcpy_h2d(self.engine._ctx, self.ptr, arr.ctypes.data, arr.nbytes))
        return self

    def download(self, shape, dtype=np.float64, offset_bytes=0):
        out = np.empty(shape, dtype=dtype)
        assert offset_bytes + out.nbytes <= self.nbytes
        _check(lib().tdr_memcpy_d2h(self.engine._ctx, out.ctypes.data, self.ptr + offset_bytes, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            lib().tdr_device_free(self.engine._ctx, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Engine:
    """One tdr_ctx: a camera set resident on one GPU."""

    def __init__(self, P, aux=None, device=0, devices=None):
        """device: GPU ordinal, or -1 for every visible GPU; devices: explicit shard list (may repeat
        an ordinal) for the single-process multi-GPU context (tdr_create_multi)."""
        P, pP = _as_d(np.asarray(P).reshape(-1, 12))
        self.n_cam = P.shape[0]
        if aux is not None:
            aux, pa = _as_d(np.asarray(aux).reshape(-1, 10))
            assert aux.shape[0] == self.n_cam
        else:
            pa = None
        ctx = _vp()
        if devices is not None:
            dv = np.ascontiguousarray(devices, dtype=np.int32)
            _check(lib().tdr_create_multi(pP, pa, self.n_cam, dv.ctypes.data, dv.shape[0], C.byref(ctx)))
        else:
            _check(lib().tdr_create(pP, pa, self.n_cam, int(device), C.byref(ctx)))
        self._ctx = ctx
        self.device = device

    def close(self):
        if getattr(self, "_ctx", None):
            if not getattr(self, "_borrowed", False):
                lib().tdr_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- multi-GPU context, device-resident
    def shard_engines(self):
        """Sub-contexts of a multi-GPU context as non-owning Engine views (one per shard)."""
        out = []
        for g in range(lib().tdr_multi_shard_count(self._ctx)):
            p = _vp()
            _check(lib().tdr_multi_shard(self._ctx, g, C.byref(p)))
            e = Engine.__new__(Engine)
            e._ctx, e.n_cam, e.device, e._borrowed = p, self.n_cam, None, True
            out.append(e)
        return out

    def multi_matrix_gather(self, d_pts, n_cam_pts, cam_stride, n_sys, d_xyz, d_xyz_root, transport=0):
        """tdr_multi_matrix_triangulate_gather: per-shard device pointers / sizes (lists), gathered path on
        shard 0's device.  Returns the globally first failing index (-1 if none)."""
        G = len(d_pts)
        pp = (C.c_void_p * G)(*d_pts)
        px = (C.c_void_p * G)(*d_xyz)
        cs = np.ascontiguousarray(cam_stride, dtype=np.int64)
        ns = np.ascontiguousarray(n_sys, dtype=np.int64)
        fb = _i64(-1)
        rc = lib().tdr_multi_matrix_triangulate_gather(self._ctx, C.cast(pp, _vp), int(n_cam_pts), cs.ctypes.data,
                                                       ns.ctypes.data, C.cast(px, _vp), d_xyz_root, int(transport),
                                                       C.byref(fb))
        self.first_bad = fb.value
        _check(rc)
        return fb.value

    # ---- host-buffer batch entry points
    def _batch(self, fn, pts, want_err, too_few_msg):
        pts, pp = _as_d(pts)
        if pts.ndim != 3 or pts.shape[2] != 2:
            raise TdrError(TDR_E_SHAPE, lib().tdr_strerror(TDR_E_SHAPE).decode())
        n_cam_pts, n_sys = pts.shape[0], pts.shape[1]
        xyz = np.empty((n_sys, 3))
        err = np.empty(n_sys) if want_err else None
        fb = _i64(-1)
        rc = fn(self._ctx, pp, n_cam_pts, n_sys, xyz.ctypes.data_as(_dp),
                err.ctypes.data_as(_dp) if want_err else None, C.byref(fb))
        self.first_bad = fb.value
        _check(rc, too_few_msg)
        return (xyz, err) if want_err else xyz

    def triangulate_points_rows(self, rows, ray=False):
        """rows: list of per-camera [n_sys, 2] float64 arrays (separate allocations, like the inner vectors
        of the reference's argument) -> xyz through tdr_*_triangulate_points_rows."""
        rows = [np.ascontiguousarray(r, dtype=np.float64) for r in rows]
        n_sys = rows[0].shape[0]
        ptrs = (C.c_void_p * len(rows))(*[r.ctypes.data for r in rows])
        xyz = np.empty((n_sys, 3))
        fb = _i64(-1)
        fn = lib().tdr_ray_triangulate_points_rows if ray else lib().tdr_matrix_triangulate_points_rows
        rc = fn(self._ctx, C.cast(ptrs, _vp), len(rows), n_sys, xyz.ctypes.data_as(_dp), None, C.byref(fb))
        self.first_bad = fb.value
        _check(rc, "Too few detections are found" if ray else "Too few rays are found")
        return xyz

    def matrix_triangulate_points(self, pts, want_err=False):
        return self._batch(lib().tdr_matrix_triangulate_points, pts, want_err, "Too few rays are found")

    def ray_triangulate_points(self, pts, want_err=False):
        return self._batch(lib().tdr_ray_triangulate_points, pts, want_err, "Too few detections are found")

    def ray_closest_point_points(self, pts, want_err=False):
        """tdr_ray_closest_point_points: the closed-form minimiser of the ray path's objective (not the LM iterate)."""
        return self._batch(lib().tdr_ray_closest_point_points, pts, want_err, "Too few detections are found")

    def ray_triangulate_points_ex(self, pts):
        """tdr_ray_triangulate_points_ex -> dict(xyz, err, iters, status): the LM batch with the per-system
        iteration count and TDR_LM_* status word."""
        pts, pp = _as_d(pts)
        if pts.ndim != 3 or pts.shape[2] != 2:
            raise TdrError(TDR_E_SHAPE, lib().tdr_strerror(TDR_E_SHAPE).decode())
        n_cam_pts, n_sys = pts.shape[0], pts.shape[1]
        xyz, err = np.empty((n_sys, 3)), np.empty(n_sys)
        iters, status = np.empty(n_sys, dtype=np.int32), np.empty(n_sys, dtype=np.int32)
        fb = _i64(-1)
        rc = lib().tdr_ray_triangulate_points_ex(self._ctx, pp, n_cam_pts, n_sys, xyz.ctypes.data_as(_dp),
                                                 err.ctypes.data_as(_dp), iters.ctypes.data, status.ctypes.data,
                                                 C.byref(fb))
        self.first_bad = fb.value
        _check(rc, "Too few detections are found")
        return {"xyz": xyz, "err": err, "iters": iters, "status": status}

    # ---- single system
    def matrix_triangulate_point(self, P_obs, uv):
        P_obs, pP = _as_d(np.asarray(P_obs).reshape(-1, 12))
        uv, pu = _as_d(np.asarray(uv).reshape(-1, 2))
        X = np.empty(3)
        e = C.c_double()
        _check(lib().tdr_matrix_triangulate_point(self._ctx, pP, pu, uv.shape[0], X.ctypes.data_as(_dp),
                                                  C.cast(C.byref(e), _dp)))
        return X, e.value

    def ray_triangulate_point(self, aux_obs, uv):
        aux_obs, pa = _as_d(np.asarray(aux_obs).reshape(-1, 10))
        uv, pu = _as_d(np.asarray(uv).reshape(-1, 2))
        X = np.empty(3)
        e = C.c_double()
        _check(lib().tdr_ray_triangulate_point(self._ctx, pa, pu, uv.shape[0], X.ctypes.data_as(_dp),
                                               C.cast(C.byref(e), _dp)))
        return X, e.value

    # ---- device-resident
    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    # ---- int16 observation format (include/tdr_amd.h: 4 B per observation, fp64 arithmetic)
    def pack_i16_device(self, d_pts, n_cam_pts, cam_stride, n_sys, d_out, out_stride, stream=None):
        _check(lib().tdr_pack_detections_i16_device(self._ctx, d_pts, n_cam_pts, cam_stride, n_sys, d_out,
                                                    out_stride, stream))

    def matrix_i16_device(self, d_pts16, n_cam_pts, cam_stride, n_sys, d_xyz, d_err=None, stream=None):
        _check(lib().tdr_matrix_triangulate_points_i16_device(self._ctx, d_pts16, n_cam_pts, cam_stride, n_sys,
                                                              d_xyz, d_err, stream))

    def matrix_triangulate_points_i16(self, pts16, want_err=False):
        """pts16: [n_cam, n_sys, 2] int16 host array -> xyz (and err)."""
        pts16 = np.ascontiguousarray(pts16, dtype=np.int16)
        if pts16.ndim != 3 or pts16.shape[2] != 2:
            raise TdrError(TDR_E_SHAPE, lib().tdr_strerror(TDR_E_SHAPE).decode())
        n_cam_pts, n_sys = pts16.shape[0], pts16.shape[1]
        xyz = np.empty((n_sys, 3))
        err = np.empty(n_sys) if want_err else None
        fb = _i64(-1)
        rc = lib().tdr_matrix_triangulate_points_i16(self._ctx, pts16.ctypes.data_as(_vp), n_cam_pts, n_sys,
                                                     xyz.ctypes.data_as(_dp),
                                                     err.ctypes.data_as(_dp) if want_err else None, C.byref(fb))
        self.first_bad = fb.value
        _check(rc, "Too few rays are found")
        return (xyz, err) if want_err else xyz

    def matrix_device(self, d_pts, n_cam_pts, cam_stride, n_sys, d_xyz, d_err=None, stream=None):
        _check(lib().tdr_matrix_triangulate_points_device(self._ctx, d_pts, n_cam_pts, cam_stride, n_sys,
                                                          d_xyz, d_err, stream))

    def ray_device(self, d_pts, n_cam_pts, cam_stride, n_sys, d_xyz, d_err=None, stream=None):
        _check(lib().tdr_ray_triangulate_points_device(self._ctx, d_pts, n_cam_pts, cam_stride, n_sys,
                                                       d_xyz, d_err, stream))

    def ray_closest_point_device(self, d_pts, n_cam_pts, cam_stride, n_sys, d_xyz, d_err=None, stream=None):
        _check(lib().tdr_ray_closest_point_points_device(self._ctx, d_pts, n_cam_pts, cam_stride, n_sys,
                                                         d_xyz, d_err, stream))

    def ray_device_ex(self, d_pts, n_cam_pts, cam_stride, n_sys, d_xyz, d_err=None, d_iters=None, d_status=None,
                      stream=None):
        _check(lib().tdr_ray_triangulate_points_device_ex(self._ctx, d_pts, n_cam_pts, cam_stride, n_sys, d_xyz,
                                                          d_err, d_iters, d_status, stream))

    def fetch_first_bad(self, stream=None):
        fb = _i64(-1)
        _check(lib().tdr_first_bad(self._ctx, stream, C.byref(fb)))
        return fb.value

    def set_matrix_variant(self, v):
        _check(lib().tdr_set_matrix_variant(self._ctx, int(v)))

    def set_ray_jacobian(self, literal):
        """tdr_set_ray_jacobian: True = the reference's central difference taken literally (2-4 x the time)."""
        _check(lib().tdr_set_ray_jacobian(self._ctx, 1 if literal else 0))

    def set_host_pipeline(self, mode=-1, chunk_systems=0):
        _check(lib().tdr_set_host_pipeline(self._ctx, int(mode), int(chunk_systems)))

    def dist_from_ray_device(self, d_cam_idx, d_uv, d_point, n, d_dist, stream=None):
        _check(lib().tdr_dist_from_ray_device(self._ctx, d_cam_idx, d_uv, d_point, n, d_dist, stream))

    def time_fp64_fma(self, waves_per_simd=4, iters=20000):
        """(TFLOP/s, ns per wave-level fp64 instruction per SIMD) of a pure v_fma_f64 loop (calibration)."""
        tf, ns = C.c_double(), C.c_double()
        _check(lib().tdr_time_fp64_fma(self._ctx, waves_per_simd, iters, C.cast(C.byref(tf), _dp),
                                       C.cast(C.byref(ns), _dp)))
        return tf.value, ns.value

    def debug_sym3_eig(self, M6v3):
        """Test hook: round 3's Jacobi restatement of DECOMP_EIG on [n, 9] rows {m00 m01 m02 m11 m12 m22 v0 v1 v2} ->
        [n, 4] rows {pinv(M) v, max_k |pinv(M)_kk|} (tdr_debug_sym3_eig_device)."""
        a = np.ascontiguousarray(M6v3, dtype=np.float64).reshape(-1, 9)
        n = a.shape[0]
        d_in, d_out = self.alloc(max(n, 1) * 72), self.alloc(max(n, 1) * 32)
        d_in.upload(a)
        _check(lib().tdr_debug_sym3_eig_device(self._ctx, d_in.ptr, n, d_out.ptr, None))
        self.synchronize()
        out = d_out.download((n, 4))
        d_in.free()
        d_out.free()
        return out

    def debug_lm_cascade(self, M6v3, n_rays):
        """Test hook: the LM loop's eigen-free DECOMP_EIG restatement on [n, 9] rows {m00 m01 m02 m11 m12 m22 v0 v1 v2}
        -> [n, 5] rows {pinv(M) v, max_k |pinv(M)_kk|, rank} (tdr_debug_lm_cascade_device)."""
        a = np.ascontiguousarray(M6v3, dtype=np.float64).reshape(-1, 9)
        n = a.shape[0]
        d_in, d_out = self.alloc(max(n, 1) * 72), self.alloc(max(n, 1) * 40)
        d_in.upload(a)
        _check(lib().tdr_debug_lm_cascade_device(self._ctx, d_in.ptr, n, int(n_rays), d_out.ptr, None))
        self.synchronize()
        out = d_out.download((n, 5))
        d_in.free()
        d_out.free()
        return out

    def dist_from_ray_batch(self, cam_idx, uv, point):
        """getDistFromRay for n (camera ordinal, pixel, 3D point) triples on host arrays -> [n] distances."""
        cam_idx = np.ascontiguousarray(cam_idx, dtype=np.int32)
        uv, pu = _as_d(np.asarray(uv).reshape(-1, 2))
        point, pp = _as_d(np.asarray(point).reshape(-1, 3))
        n = cam_idx.shape[0]
        if uv.shape[0] != n or point.shape[0] != n:
            raise TdrError(TDR_E_SHAPE, lib().tdr_strerror(TDR_E_SHAPE).decode())
        out = np.empty(n)
        _check(lib().tdr_dist_from_ray_batch(self._ctx, cam_idx.ctypes.data_as(_vp), pu, pp, n,
                                             out.ctypes.data_as(_dp)))
        return out

    def synth_device(self, seed, n_drones, sys_offset, n_sys, cam_stride, d_pts, d_truth=None,
                     noise_px=2.0, dropout_p=0.05, round_to_int=True, ray_model=False, stream=None):
        _check(lib().tdr_synth_detections_device(self._ctx, int(seed), int(n_drones), int(sys_offset),
                                                 int(n_sys), int(cam_stride), float(noise_px),
                                                 float(dropout_p), int(round_to_int), int(ray_model),
                                                 d_pts, d_truth, stream))

    def score_combinations_levelwise(self, path, det_count, dets, error_max, min_cameras=2, cap=None):
        """Breadth-first sweep of the reference's prefix tree (tdr_score_combinations_levelwise):
        path 0 = matrix, 1 = ray.  Same return as matrix_score_combinations."""
        det_count = np.ascontiguousarray(det_count, dtype=np.int32).reshape(-1, self.n_cam)
        dets = np.ascontiguousarray(dets, dtype=np.float64).reshape(-1, 2)
        n_frames = det_count.shape[0]
        cap = int(cap) if cap is not None else max(1024, 4 * n_frames)
        while True:
            out = np.zeros(cap, dtype=CANDIDATE_DTYPE)
            n = _i64(0)
            rc = lib().tdr_score_combinations_levelwise(self._ctx, int(path), n_frames, det_count.ctypes.data,
                                                        dets.ctypes.data_as(_dp), float(error_max),
                                                        int(min_cameras), out.ctypes.data, cap, C.byref(n))
            if rc == TDR_E_ARG and n.value > cap:
                cap = n.value
                continue
            _check(rc)
            return out[:n.value]

    def fetch_scored_combinations(self, cap):
        """tdr_fetch_scored_combinations: the survivors a sweep call could not return for lack of capacity.
        Returns (rc, survivors or None, n_needed)."""
        out = np.zeros(int(cap), dtype=CANDIDATE_DTYPE)
        n = _i64(0)
        rc = lib().tdr_fetch_scored_combinations(self._ctx, out.ctypes.data, int(cap), C.byref(n))
        return rc, (out[:n.value] if rc == TDR_OK else None), n.value

    def ray_score_combinations_ex(self, det_count, dets, error_max=120.0, min_cameras=2):
        """tdr_ray_score_combinations_ex -> (survivors, cuts): survivors carry TDR_CANDIDATE_FLAGGED where a solve
        on their prefix path had a non-zero LM status; cuts are the prefix nodes cut on such a solve
        (flags >> 8 = depth, leaf = index over cameras 0..depth)."""
        det_count = np.ascontiguousarray(det_count, dtype=np.int32).reshape(-1, self.n_cam)
        dets = np.ascontiguousarray(dets, dtype=np.float64).reshape(-1, 2)
        n_frames = det_count.shape[0]
        cap, ccap = max(1024, 4 * n_frames), 1024
        while True:
            out = np.zeros(cap, dtype=CANDIDATE_DTYPE)
            cuts = np.zeros(ccap, dtype=CANDIDATE_DTYPE)
            n, nc = _i64(0), _i64(0)
            rc = lib().tdr_ray_score_combinations_ex(self._ctx, n_frames, det_count.ctypes.data,
                                                     dets.ctypes.data_as(_dp), float(error_max), int(min_cameras),
                                                     out.ctypes.data, cap, C.byref(n), cuts.ctypes.data, ccap,
                                                     C.byref(nc))
            if (rc == TDR_E_ARG and n.value > cap) or nc.value > ccap:
                cap, ccap = max(cap, n.value), max(ccap, nc.value)
                continue
            _check(rc)
            return out[:n.value], cuts[:nc.value]

    def matrix_score_combinations(self, det_count, dets, error_max=1e5, min_cameras=2, cap=None):
        """Batched DroneClassifier::fillCombinationQueue scoring (src/DroneClassifier.cpp:156-198).
        det_count [n_frames, n_cam] int32; dets [total, 2].  Returns a structured array of survivors
        (frame, n_used, leaf, x, y, z, err) in the reference's push order."""
        det_count = np.ascontiguousarray(det_count, dtype=np.int32).reshape(-1, self.n_cam)
        dets = np.ascontiguousarray(dets, dtype=np.float64).reshape(-1, 2)
        assert dets.shape[0] == int(det_count.sum())
        n_frames = det_count.shape[0]
        cap = int(cap) if cap is not None else max(1024, 4 * n_frames)
        while True:
            out = np.zeros(cap, dtype=CANDIDATE_DTYPE)
            n = _i64(0)
            rc = lib().tdr_matrix_score_combinations(self._ctx, n_frames, det_count.ctypes.data,
                                                     dets.ctypes.data_as(_dp), float(error_max),
                                                     int(min_cameras), out.ctypes.data, cap, C.byref(n))
            if rc == TDR_E_ARG and n.value > cap:
                cap = n.value
                continue
            _check(rc)
            return out[:n.value]

    def synchronize(self, stream=None):
        _check(lib().tdr_stream_synchronize(self._ctx, stream))

    def time_stream(self, mode, streams, d_src, d_dst, n_lanes, warmup=3, iters=20):
        ms = C.c_double()
        _check(lib().tdr_time_stream(self._ctx, int(mode), int(streams), d_src, d_dst, int(n_lanes), int(warmup),
                                     int(iters), C.cast(C.byref(ms), _dp)))
        return ms.value

    def time_launches(self, path, d_pts, n_cam_pts, cam_stride, n_sys, d_xyz, d_err=None, warmup=3,
                      iters=20, stream=None):
        ms = C.c_double()
        _check(lib().tdr_time_launches(self._ctx, int(path), d_pts, n_cam_pts, cam_stride, n_sys, d_xyz,
                                       d_err, int(warmup), int(iters), stream, C.cast(C.byref(ms), _dp)))
        return ms.value


# ------------------------------------------------------------------------------------------ plugin surface
def create_ray(aux, uv):
    """tdr_create_ray: Triangulator::createRayForPoint (src/Triangulator.cpp:46-55) -> (origin[3], dir[3]); host side."""
    aux, pa = _as_d(aux)
    uv, pu = _as_d(uv)
    o, d = np.empty(3), np.empty(3)
    _check(lib().tdr_create_ray(pa, pu, o.ctypes.data_as(_dp), d.ctypes.data_as(_dp)))
    return o, d


def ray_direction_for_pixel(aux, uv):
    """tdr_ray_direction_for_pixel: Triangulator::calculateRayDirectionForPixel (src/Triangulator.cpp:27-44)."""
    aux, pa = _as_d(aux)
    uv, pu = _as_d(uv)
    d = np.empty(3)
    _check(lib().tdr_ray_direction_for_pixel(pa, pu, d.ctypes.data_as(_dp)))
    return d


def rotate_point_by_quaternion(quat, point):
    """tdr_rotate_point_by_quaternion: Triangulator::rotatePointByQuaternion (src/Triangulator.cpp:15-25)."""
    quat, pq = _as_d(quat)
    point, pp = _as_d(point)
    out = np.empty(3)
    _check(lib().tdr_rotate_point_by_quaternion(pq, pp, out.ctypes.data_as(_dp)))
    return out


def dist_to_ray(origin, direction, point, squared=False):
    """tdr_dist_to_ray: Triangulator::distToRay (src/Triangulator.cpp:3-9)."""
    origin, po = _as_d(origin)
    direction, pd = _as_d(direction)
    point, pp = _as_d(point)
    return lib().tdr_dist_to_ray(po, pd, pp, 1 if squared else 0)


class CamPointPair:
    """src/Triangulator.h:11-15"""

    def __init__(self, camera, point):
        self.camera = camera
        self.point = (float(point[0]), float(point[1]))

    def projectionMatrix(self):
        return self.camera.cameraPerspectiveMatrix


class Triangulator:
    """src/Triangulator.h:9-59 (abstract).  Cameras are snapshotted at construction."""
    type_ = ""
    _too_few = ""

    def __init__(self, cameras_, device=0):
        self.cameras = list(cameras_)
        P, aux = pack_cameras(self.cameras)
        self._engine = Engine(P, aux, device)

    def getCameras(self):
        return self.cameras

    def getCamera(self, camera):
        return self.cameras[camera]

    def getType(self):
        return self.type_

    @staticmethod
    def _check_points(points):
        # src/MatrixTriangulator.cpp:72-76, src/RayTriangulator.cpp:54-58
        if len(points) == 0:  # UB in the reference (points.size()-1 underflow); rejected
            raise TdrError(TDR_E_SHAPE, "Every camera should have the same number of points")
        n = len(points[0])
        for row in points:
            if len(row) != n:
                raise TdrError(TDR_E_SHAPE, "Every camera should have the same number of points")
        return np.asarray(points, dtype=np.float64).reshape(len(points), n, 2)

    def getDistFromRay(self, pair, point):
        """src/Triangulator.cpp:57-61: one observation with the pair's own camera (tdr_dist_from_ray)."""
        _, pa = _as_d(pair.camera.aux)
        _, pu = _as_d(np.array(pair.point, dtype=np.float64))
        _, pp = _as_d(np.asarray(point, dtype=np.float64))
        d = C.c_double()
        _check(lib().tdr_dist_from_ray(self._engine._ctx, pa, pu, pp, C.cast(C.byref(d), _dp)))
        return d.value


class MatrixTriangulator(Triangulator):
    """src/MatrixTriangulator.h:6-25"""
    type_ = "matrix"

    def triangulatePoint(self, images):
        P = np.stack([im.projectionMatrix().reshape(12) for im in images])
        uv = np.array([im.point for im in images])
        X, e = self._engine.matrix_triangulate_point(P, uv)
        return X, e

    def triangulatePoints(self, points):
        return self._engine.matrix_triangulate_points(self._check_points(points))


class RayTriangulator(Triangulator):
    """src/RayTriangulator.h:13-41"""
    type_ = "ray"

    def triangulatePoint(self, images):
        aux = np.stack([im.camera.aux for im in images])
        uv = np.array([im.point for im in images])
        return self._engine.ray_triangulate_point(aux, uv)

    def triangulatePoints(self, points):
        return self._engine.ray_triangulate_points(self._check_points(points))

    # engine-side additions, as on the C++ adapter (host/RayTriangulator.h)
    def closestPoints(self, points, want_err=False):
        """The minimiser of the ray objective in closed form (tdr_ray_closest_point_points) — not the LM iterate."""
        return self._engine.ray_closest_point_points(self._check_points(points), want_err=want_err)

    def setLiteralJacobian(self, literal):
        """tdr_set_ray_jacobian: differentiate exactly as RayClosestPoint::compute does (2-4 x the time)."""
        self._engine.set_ray_jacobian(literal)

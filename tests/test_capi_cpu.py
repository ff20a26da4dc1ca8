"""CPU tests (no GPU) of the drop-in boundary: the C-ABI library loads, exports every symbol
include/tdr_amd.h declares, its host-side pieces (camera derivation, error texts) match the oracle,
and it FAILS LOUDLY without a device (no CPU fallback).  No compute entry point is called here."""
import ctypes as C
import importlib
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sc = importlib.import_module("3d-reconstruction-triangulation_amd.scenes")


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "tdr_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(tdr_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol(tdr):
    names = _declared_symbols()
    assert len(names) >= 20
    handle = C.CDLL(tdr.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), f"{n} declared in include/tdr_amd.h but not exported"
    assert sorted(tdr.C_API) == names, "ctypes table out of sync with the header"
    tdr.lib()


def test_error_texts_are_the_references(tdr):
    L = tdr.lib()
    # src/MatrixTriangulator.cpp:74-75,93 / src/RayTriangulator.cpp:56-57
    assert L.tdr_strerror(tdr.TDR_E_SHAPE) == b"Every camera should have the same number of points"
    assert L.tdr_strerror(tdr.TDR_E_TOO_FEW) == b"Too few rays are found"
    assert b"no CPU fallback" in L.tdr_strerror(tdr.TDR_E_NO_DEVICE)


def test_camera_derive_matches_oracle(tdr, oracle):
    for p in sc.ring_camera_params(7, seed=8):
        cam = tdr.createCamera(3, p["width"], p["height"], p["focal"], p["tvec"], p["rquat"])
        ref = oracle.camera_derive(p["width"], p["height"], p["focal"], p["tvec"], p["rquat"])
        np.testing.assert_allclose(cam.cameraPerspectiveMatrix, ref["P"], rtol=1e-15, atol=1e-12)
        np.testing.assert_allclose(cam.cameraExtrinsicMatrix, ref["E"], rtol=1e-15, atol=1e-12)
        np.testing.assert_allclose(cam.cameraMatrix, ref["K"], rtol=1e-15)
        assert (cam.cx, cam.cy, cam.width, cam.height) == (ref["cx"], ref["cy"], p["width"], p["height"])
        for k in ("fx", "fy", "fovx", "fovy"):
            assert getattr(cam, k) == pytest.approx(ref[k], rel=1e-15)
        np.testing.assert_allclose(cam.aux, ref["aux"], rtol=1e-15)
        assert cam.diagonal == int(np.sqrt(p["width"] ** 2 + p["height"] ** 2))
        assert cam.getCamId() == 3


def test_camera_derive_rejects_unset_fields(tdr):
    """tdr::Camera's 'Set ... first.' throws (src/Camera.h:79,90,123)."""
    for w, h, f in ((0, 1082, 1000.0), (1924, 0, 1000.0), (1924, 1082, 0.0)):
        with pytest.raises(tdr.TdrError) as ei:
            tdr.createCamera(0, w, h, f, [0, 0, 0], [1, 0, 0, 0])
        assert ei.value.status == tdr.TDR_E_CAMERA


def test_no_device_fails_loudly(tdr):
    L = tdr.lib()
    if L.tdr_device_count() > 0:
        pytest.skip("a HIP device is present")
    P = np.zeros((2, 12))
    with pytest.raises(tdr.TdrError) as ei:
        tdr.Engine(P, None)
    assert ei.value.status == tdr.TDR_E_NO_DEVICE


def test_points_shape_check_of_the_plugin_surface(tdr):
    """src/MatrixTriangulator.cpp:72-76: ragged rows -> the reference's message; empty -> rejected."""
    chk = tdr.Triangulator._check_points
    ok = chk([[(1, 2), (3, 4)], [(5, 6), (7, 8)]])
    assert ok.shape == (2, 2, 2)
    with pytest.raises(tdr.TdrError, match="Every camera should have the same number of points"):
        chk([[(1, 2), (3, 4)], [(5, 6)]])
    with pytest.raises(tdr.TdrError):
        chk([])


def test_missing_library_raises(tdr, monkeypatch):
    monkeypatch.setattr(tdr, "_lib", None)
    monkeypatch.setattr(tdr, "LIB_PATH", "/nonexistent/libtdr_amd.so")
    with pytest.raises(ImportError, match="no CPU fallback"):
        tdr.lib()


def test_null_context_is_rejected_without_touching_a_device(tdr):
    """Every compute entry point validates its context before any HIP call: with a NULL context it returns
    TDR_E_ARG (no crash, no device needed).  Covers the entry points added after the first ABI draft as well."""
    L = tdr.lib()
    z = np.zeros(16)
    zp = z.ctypes.data_as(C.POINTER(C.c_double))
    zv = C.c_void_p(z.ctypes.data)
    fb = C.c_int64(0)
    calls = [
        lambda: L.tdr_matrix_triangulate_points(None, zp, 2, 1, zp, None, C.byref(fb)),
        lambda: L.tdr_ray_triangulate_points(None, zp, 2, 1, zp, None, C.byref(fb)),
        lambda: L.tdr_matrix_triangulate_points_device(None, zv, 2, 1, 1, zv, None, None),
        lambda: L.tdr_ray_triangulate_points_device(None, zv, 2, 1, 1, zv, None, None),
        lambda: L.tdr_matrix_triangulate_points_i16(None, zv, 2, 1, zp, None, C.byref(fb)),
        lambda: L.tdr_matrix_triangulate_points_i16_device(None, zv, 2, 1, 1, zv, None, None),
        lambda: L.tdr_pack_detections_i16_device(None, zv, 2, 1, 1, zv, 1, None),
        lambda: L.tdr_dist_from_ray_batch(None, zv, zp, zp, 1, zp),
        lambda: L.tdr_time_fp64_fma(None, 2, 10, zp, zp),
    ]
    for call in calls:
        assert call() == tdr.TDR_E_ARG


def test_host_i16_packer_is_exact_or_says_so(tdr):
    """tdr_pack_detections_i16_host (the conversion behind the narrow transport of the host-buffer path): exact on
    integer-valued int16 data of any length / alignment (vector body + scalar tail), and *all_exact = 0 for a
    fraction, an out-of-range value, NaN or infinity anywhere in the block.  Runs the SSE2 or the AVX2 body,
    whichever this CPU has; needs no GPU."""
    L = tdr.lib()
    rng = np.random.default_rng(7)
    dp = C.POINTER(C.c_double)
    for n in (0, 1, 2, 3, 4, 5, 7, 8, 9, 63, 64, 65, 1000, 4099):
        for off in (0, 1):  # 16-B aligned and misaligned sources
            buf = np.zeros(2 * n + 2 * off + 2)
            src = buf[2 * off: 2 * off + 2 * n]
            src[:] = rng.integers(-32768, 32768, 2 * n).astype(np.float64)
            if n:
                src[0], src[-1] = -32768.0, 32767.0
                if n > 2:
                    src[2] = -1.0  # the missing-observation sentinel survives
            out = np.full(2 * n + 8, 12345, dtype=np.int16)
            ok = C.c_int(-1)
            p_src = C.cast(buf.ctypes.data + 16 * off, dp)
            assert L.tdr_pack_detections_i16_host(p_src, n, out.ctypes.data, C.byref(ok)) == tdr.TDR_OK
            assert ok.value == 1
            np.testing.assert_array_equal(out[:2 * n].astype(np.float64), src)
            assert (out[2 * n:] == 12345).all()  # nothing written past the block
            for bad in (0.5, 32768.0, -32769.0, 1e300, float("nan"), float("inf"), 100.0000001):
                if n == 0:
                    continue
                pos = int(rng.integers(0, 2 * n))
                keep = src[pos]
                src[pos] = bad
                assert L.tdr_pack_detections_i16_host(p_src, n, out.ctypes.data, C.byref(ok)) == tdr.TDR_OK
                assert ok.value == 0, (n, off, bad, pos)
                src[pos] = keep
    assert L.tdr_pack_detections_i16_host(None, 4, out.ctypes.data, None) == tdr.TDR_E_ARG


def test_protected_static_helpers_match_the_oracle_bit_for_bit(tdr, oracle):
    """tdr_create_ray / tdr_ray_direction_for_pixel / tdr_rotate_point_by_quaternion / tdr_dist_to_ray — the host-side
    exports behind the adapter's protected statics (src/Triangulator.h:27-42, src/Triangulator.cpp:3-55; VERDICT r4
    item 4).  Host code, no FMA contraction on baseline x86-64: the same operations in the same order as the oracle's
    restatement, hence the same bits; and the two halves compose to tdr_create_ray exactly."""
    rng = np.random.default_rng(5)
    for p in sc.ring_camera_params(9, seed=4):
        cam = tdr.createCamera(0, p["width"], p["height"], p["focal"], p["tvec"], p["rquat"])
        aux = cam.aux.copy()
        aux[:4] *= rng.uniform(0.7, 1.4)  # the quaternion is used as stored, never normalised
        for _ in range(40):
            uv = rng.uniform([-50, -50], [2000, 1200])
            if rng.random() < 0.3:
                uv = np.rint(uv)
            o, d = tdr.create_ray(aux, uv)
            ro, rd = oracle.create_ray(aux, uv[0], uv[1])
            assert np.array_equal(o, ro) and np.array_equal(d, rd)
            pix = tdr.ray_direction_for_pixel(aux, uv)
            assert abs(np.linalg.norm(pix) - 1.0) < 1e-15
            assert np.array_equal(tdr.rotate_point_by_quaternion(aux[:4], pix), d)
            pt = rng.uniform(-3000, 3000, 3)
            dist = tdr.dist_to_ray(o, d, pt)
            assert dist == oracle.get_dist_from_ray(aux, uv[0], uv[1], pt)
            assert tdr.dist_to_ray(o, d, pt, squared=True) == pytest.approx(dist * dist, rel=4e-16)
            # independent: numpy cross product
            assert dist == pytest.approx(np.linalg.norm(np.cross(d, pt - o)), rel=1e-13)
    L = tdr.lib()
    assert L.tdr_create_ray(None, None, None, None) == tdr.TDR_E_ARG
    assert np.isnan(L.tdr_dist_to_ray(None, None, None, 0))

"""GPU parity of the closed-form ray-closest-point entry points (tdr_ray_closest_point_points[_device],
include/tdr_amd.h) — the HBM-bound cross-check of the ray path.

What is compared with what:
  * numpy / LAPACK: oracle.np_ray_closed_form solves sum (|d|^2 I - d d^T) x = sum (|d|^2 I - d d^T) o with
    numpy.linalg.solve from the reference's literal ray set-up (np_create_ray), EVERY system — tolerance 1e-12
    relative up to cond(H) = 1e3, 4 eps cond(H) beyond (_tol below; the kernel folds the ray set-up into an affine
    map of the pixel: directions differ by a few ulp);
  * the C oracle's oracle_ray_closest_point (reference ray set-up, x0 - pinv_EIG(H)(H x0 - g), mean distToRay):
    every system, point and err;
  * the LM entry points (GPU and oracle) on every system whose status word is 0: <= 1e-7 — that part lives in
    tests/test_gpu_parity.py::test_ray_lm_status_is_two_sided_on_a_million_random_systems, which runs the
    million-system population anyway.
PARITY UNPINNED like everything on this path: the reference holds no fixture for it (oracle/tdr_oracle.c header).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CF_TIGHT = 1e-12
EPS = 2.220446049250313e-16


def _tol(cond):
    """Relative tolerance per system.  H = sum (|d|^2 I - d d^T) is a normal matrix: for rays that are nearly
    (anti)parallel — two cameras facing each other with the point near the line between them — it is ill-conditioned,
    and every fp64 evaluation of H^-1 g (numpy's, the C oracle's, the kernel's) carries a few eps x cond(H): the ray
    weight |d|^2 / |e|^2 is rounded, so (|d|^2 I - d d^T) d is not exactly 0, and that leak lies along the rays —
    H's weak direction for such a system.  1e-12 up to cond 1e3 (99 % of the random populations here
    have cond < 60), 4 eps cond beyond.  The kernel's worst observed case: 1.4e-12 at cond 4140 (1.5 eps cond)."""
    return np.maximum(CF_TIGHT, 4 * EPS * np.asarray(cond))


def _scene(tdr, n_cam, seed=0, quat_scale=1.0):
    sc = __import__("importlib").import_module(tdr.__name__ + ".scenes")
    cp = sc.ring_camera_params(n_cam, seed)
    cams = [tdr.createCamera(i, c["width"], c["height"], c["focal"], c["tvec"],
                             [q * quat_scale for q in c["rquat"]]) for i, c in enumerate(cp)]
    P, aux = tdr.pack_cameras(cams)
    return sc, cams, P, aux


def _rel_each(a, b):
    return np.abs(a - b).max(axis=1) / np.abs(b).max(axis=1)


def _rel(a, b):
    return float(_rel_each(a, b).max())


@pytest.mark.parametrize("n_cam,n_sys,drop", [(2, 65, 0.0), (3, 300, 0.1), (4, 1000, 0.2), (6, 640, 0.3),
                                              (8, 2000, 0.05), (12, 700, 0.4), (16, 1500, 0.05), (24, 333, 0.5),
                                              (32, 515, 0.7)])
def test_closed_form_vs_numpy_and_oracle(tdr, oracle, n_cam, n_sys, drop):
    sc, cams, P, aux = _scene(tdr, n_cam, seed=60 + n_cam)
    X = sc.smooth_path(n_sys, seed=n_sys)
    pts = sc.make_detections(sc.project_ray_model(aux, X), seed=n_sys, dropout_p=drop, min_valid=2)
    eng = tdr.Engine(P, aux)
    xyz, err = eng.ray_closest_point_points(pts, want_err=True)
    np.testing.assert_array_equal(eng.ray_closest_point_points(pts), xyz)  # err on / off: the same point
    rc, ref, ref_err, fb = oracle.ray_closest_point_points(aux, pts, want_err=True)
    assert rc == 0 and fb == -1
    # independent of the C oracle: numpy.linalg.solve on the literal rays, EVERY system, with cond(H) beside it
    np_xyz, np_err, cond = oracle.np_ray_closed_form_batch(aux, pts)
    tol = _tol(cond)
    assert (_rel_each(xyz, np_xyz) <= tol).all(), float((_rel_each(xyz, np_xyz) / tol).max())
    assert (_rel_each(xyz, ref) <= tol).all(), float((_rel_each(xyz, ref) / tol).max())
    assert (cond <= 1e3).mean() >= 0.98  # the 1e-12 bound is the one that applies almost everywhere
    # the mean distance moves by at most the point's displacement
    err_tol = 1e-10 + tol * np.abs(np_xyz).max(axis=1)
    assert (np.abs(err - ref_err) <= err_tol).all() and (np.abs(err - np_err) <= err_tol).all()
    for s in range(0, n_sys, max(1, n_sys // 20)):  # the scalar numpy form on a sample (pins the batch form above)
        v = (pts[:, s, 0] != -1) & (pts[:, s, 1] != -1)
        cf = oracle.np_ray_closed_form(aux[v], pts[v, s])
        assert np.abs(np_xyz[s] - cf).max() / np.abs(cf).max() <= tol[s]


def test_closed_form_known_answer(tdr, oracle):
    """Exact (unrounded, noise-free) pixels of the ray model -> the 3D point itself, residual distance ~ 0."""
    sc, cams, P, aux = _scene(tdr, 8, seed=21)
    X = sc.smooth_path(3000, seed=3)
    xyz, err = tdr.Engine(P, aux).ray_closest_point_points(sc.project_ray_model(aux, X), want_err=True)
    assert sc.rel_err(xyz, X) <= 1e-12
    assert err.max() < 1e-8


def test_closed_form_raw_quaternion_is_not_normalised(tdr, oracle):
    """The reference rotates by the quaternion as stored (src/Triangulator.cpp:15-25): |d| = |q|^2.  The minimiser
    does not depend on the scale when every camera has the same one, the distances do (x |q|^2) — and with a
    different scale per camera the rays are weighted differently, which moves the point."""
    sc, cams, P, aux = _scene(tdr, 6, seed=31)
    X = sc.smooth_path(800, seed=9)
    pts = sc.make_detections(sc.project_ray_model(aux, X), seed=9)
    aux2 = aux.copy()
    aux2[:, :4] *= np.array([1.3, 0.8, 1.0, 1.7, 0.6, 1.1])[:, None]
    xyz, err = tdr.Engine(P, aux2).ray_closest_point_points(pts, want_err=True)
    rc, ref, ref_err, _ = oracle.ray_closest_point_points(aux2, pts, want_err=True)
    assert rc == 0
    np_xyz, np_err, cond = oracle.np_ray_closed_form_batch(aux2, pts)
    assert (_rel_each(xyz, ref) <= _tol(cond)).all() and (_rel_each(xyz, np_xyz) <= _tol(cond)).all()
    np.testing.assert_allclose(err, ref_err, rtol=1e-9)
    np.testing.assert_allclose(err, np_err, rtol=1e-9)
    plain = tdr.Engine(P, aux).ray_closest_point_points(pts)
    assert _rel(xyz, plain) > 1e-6  # the weights matter


def test_closed_form_agrees_with_the_lm_entry_where_its_status_is_zero(tdr, oracle):
    """The small-batch version of the million-system check in tests/test_gpu_parity.py."""
    for n_cam, drop in ((2, 0.0), (4, 0.2), (16, 0.05)):
        sc, cams, P, aux = _scene(tdr, n_cam, seed=70 + n_cam)
        X = sc.smooth_path(4000, seed=n_cam)
        pts = sc.make_detections(sc.project_ray_model(aux, X), seed=n_cam, dropout_p=drop, min_valid=2)
        eng = tdr.Engine(P, aux)
        cf = eng.ray_closest_point_points(pts)
        g = eng.ray_triangulate_points_ex(pts)
        ok = (g["status"] & 3) == 0
        assert ok.mean() > 0.5
        assert _rel(cf[ok], g["xyz"][ok]) <= 1e-7
        # ... and where the LM run did NOT reach the minimiser the two entry points differ: the closed form is not a
        # replacement for the LM entry (include/tdr_amd.h)
        if (~ok).sum() >= 5:
            assert _rel(cf[~ok], g["xyz"][~ok]) > 1e-7


def test_closed_form_too_few_and_shape(tdr, oracle):
    sc, cams, P, aux = _scene(tdr, 4, seed=22)
    X = sc.smooth_path(300, seed=4)
    pts = sc.make_detections(sc.project_ray_model(aux, X), seed=4)
    pts[:3, 200] = -1
    pts[1:, 250] = -1
    eng = tdr.Engine(P, aux)
    with pytest.raises(tdr.TdrError, match="Too few detections are found"):
        eng.ray_closest_point_points(pts)
    assert eng.first_bad == 200
    rc, _, _, fb = oracle.ray_closest_point_points(aux, pts)
    assert (rc, fb) == (oracle.E_TOO_FEW, 200)
    # more point rows than cameras is UB in the reference (src/RayTriangulator.cpp:65,69): rejected, as by the LM entry
    eng3 = tdr.Engine(P[:3], aux[:3])
    with pytest.raises(tdr.TdrError) as ei:
        eng3.ray_closest_point_points(pts)
    assert ei.value.status == tdr.TDR_E_SHAPE
    # fewer rows than cameras: the first rows' cameras
    ok = sc.make_detections(sc.project_ray_model(aux, X), seed=5, dropout_p=0.0)
    np.testing.assert_array_equal(eng.ray_closest_point_points(ok[:3]),
                                  tdr.Engine(P[:3], aux[:3]).ray_closest_point_points(ok[:3]))
    # a context without the ray model cannot run it
    with pytest.raises(tdr.TdrError) as ei:
        tdr.Engine(P).ray_closest_point_points(ok)
    assert ei.value.status == tdr.TDR_E_ARG
    assert eng.ray_closest_point_points(pts[:, :0]).shape == (0, 3)


def test_closed_form_parallel_rays(tdr, oracle):
    """All rays parallel (two cameras with one orientation looking at the same pixel): H has rank 2 and the
    minimiser is a line.  Both sides return the point of that line that pseudo-inverse steps reach from the mean of
    the origins (src/RayTriangulator.cpp:90-98)."""
    sc, cams, P, aux = _scene(tdr, 3, seed=5)
    aux = aux.copy()
    aux[1, :4] = aux[0, :4]
    aux[2, :4] = aux[0, :4]
    aux[1, 7:] = aux[0, 7:]
    aux[2, 7:] = aux[0, 7:]
    n = 200
    rng = np.random.default_rng(3)
    pts = np.empty((3, n, 2))
    pts[0, :, 0] = rng.integers(100, 1800, n)
    pts[0, :, 1] = rng.integers(100, 1000, n)
    pts[1] = pts[0]
    pts[2] = pts[0]
    pts[2, ::3] = -1  # two or three parallel rays
    eng = tdr.Engine(P, aux)
    xyz, err = eng.ray_closest_point_points(pts, want_err=True)
    rc, ref, ref_err, _ = oracle.ray_closest_point_points(aux, pts, want_err=True)
    assert rc == 0
    assert np.isfinite(xyz).all()
    assert _rel(xyz, ref) <= 1e-9
    np.testing.assert_allclose(err, ref_err, rtol=1e-8)
    # it IS a minimiser: moving along the common direction changes nothing, the distances are those of the lines
    o0, d0 = oracle.create_ray(aux[0], *pts[0, 0])
    moved = xyz[0] + 123.0 * d0
    e_moved = np.mean([oracle.get_dist_from_ray(aux[c], *pts[c, 0], moved) for c in range(2)])
    assert e_moved == pytest.approx(err[0], rel=1e-9)
    # ... the one nearest to the mean of the origins
    x0 = aux[:2, 4:7].mean(axis=0)
    assert abs(np.dot(xyz[0] - x0, d0)) <= 1e-6 * np.linalg.norm(x0)


def test_closed_form_device_entry_strides_and_large_batch(tdr, oracle):
    """Device-resident entry point on a slice of a wider [cam][sys] array (cam_stride > n_sys), and a batch past the
    192 MiB switch to nontemporal loads / 64-thread blocks; both bit-identical to the host-buffer call."""
    sc, cams, P, aux = _scene(tdr, 16, seed=77)
    eng = tdr.Engine(P, aux)
    n_all = 900_000  # 16 x 16 B + 24 B = 280 B per system: 252 MB > 192 MiB
    pts, xyz, err = eng.alloc(16 * n_all * 16), eng.alloc(n_all * 24), eng.alloc(n_all * 8)
    eng.synth_device(1234, 1, 0, n_all, n_all, pts.ptr, ray_model=True)
    eng.ray_closest_point_device(pts.ptr, 16, n_all, n_all, xyz.ptr, err.ptr)
    eng.synchronize()
    assert eng.fetch_first_bad() == -1
    n_chk = 50_000
    h_pts = np.stack([pts.download((n_chk, 2), offset_bytes=c * n_all * 16) for c in range(16)])
    h_xyz, h_err = xyz.download((n_all, 3)), err.download((n_all,))
    rc, ref, ref_err, _ = oracle.ray_closest_point_points(aux, h_pts, want_err=True)
    assert rc == 0
    _, _, cond = oracle.np_ray_closed_form_batch(aux, h_pts)
    assert (_rel_each(h_xyz[:n_chk], ref) <= _tol(cond)).all()
    np.testing.assert_allclose(h_err[:n_chk], ref_err, rtol=1e-9)
    # a slice in place: systems [s0, s0 + n), camera rows n_all apart, default cache policy (small batch)
    s0, n = 12_345, 7_001
    xs = eng.alloc(n * 24)
    eng.ray_closest_point_device(pts.ptr + s0 * 16, 16, n_all, n, xs.ptr)
    eng.synchronize()
    h_xs = xs.download((n, 3))
    np.testing.assert_array_equal(h_xs, h_xyz[s0:s0 + n])
    # the host-buffer call (pinned block) on the same systems
    host = eng.ray_closest_point_points(h_pts[:, s0:s0 + n])
    np.testing.assert_array_equal(host, h_xs)
    # too few on the device path: recorded, NaN, first index
    bad = h_pts[:, :1000].copy()
    bad[1:, 700] = -1.0
    d_bad = eng.alloc(bad.nbytes).upload(bad)
    eng.ray_closest_point_device(d_bad.ptr, 16, 1000, 1000, xs.ptr)
    assert eng.fetch_first_bad() == 700
    h_bad = xs.download((1000, 3))
    assert np.isnan(h_bad[700]).all()
    np.testing.assert_array_equal(np.delete(h_bad, 700, axis=0), np.delete(h_xyz[:1000], 700, axis=0))


@pytest.mark.parametrize("mode,chunk", [(0, 0), (1, 4096), (2, 777), (3, 4096)])
def test_closed_form_host_pipeline_modes_are_bit_identical(tdr, oracle, mode, chunk):
    """A host batch too large for the pinned block goes through the chunked pipeline: every transport
    mode returns the bits of the single-launch result."""
    sc, cams, P, aux = _scene(tdr, 8, seed=88)
    n = 40_000  # 8 x 16 + 24 + 8 = 160 B per system: 6.4 MB > the 4 MiB pinned block
    X = sc.smooth_path(n, seed=8)
    pts = sc.make_detections(sc.project_ray_model(aux, X), seed=8)
    eng = tdr.Engine(P, aux)
    ref = eng.ray_closest_point_points(pts[:, :20_000], want_err=True)  # pinned-block route
    eng.set_host_pipeline(mode, chunk)
    xyz, err = eng.ray_closest_point_points(pts, want_err=True)
    np.testing.assert_array_equal(xyz[:20_000], ref[0])
    np.testing.assert_array_equal(err[:20_000], ref[1])
    rc, o_xyz, o_err, _ = oracle.ray_closest_point_points(aux, pts, want_err=True)
    _, _, cond = oracle.np_ray_closed_form_batch(aux, pts)
    assert (_rel_each(xyz, o_xyz) <= _tol(cond)).all()
    bad = pts.copy()
    bad[:7, 31_000] = -1
    with pytest.raises(tdr.TdrError, match="Too few detections are found"):
        eng.ray_closest_point_points(bad)
    assert eng.first_bad == 31_000

"""GPU parity tests proper: the HIP path (through the C ABI, include/tdr_amd.h) against the CPU
oracle on the same inputs.

Gate (BASELINE.json north_star): <= 1e-4 relative on every 3D point,
    max_i |p_gpu_i - p_cpu_i|_inf / |p_cpu_i|_inf          (SURVEY.md §8d).
The kernels compute in fp64, so the tests additionally assert the much tighter bound that is
actually expected (TIGHT): a regression that stays under 1e-4 but loses fp64 accuracy still fails.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GATE = 1e-4     # north_star tolerance
TIGHT = 1e-9    # fp64 normal equations vs fp64 Jacobi-SVD pinv at kappa(A) ~ 2
RAY_TIGHT = 1e-7  # LM exit iterate vs LM exit iterate (both within ~1e-6 mm of the minimiser)


def _scene(tdr, oracle, n_cam, seed=0):
    sc = __import__("importlib").import_module(tdr.__name__ + ".scenes")
    cp = sc.ring_camera_params(n_cam, seed)
    cams = [tdr.createCamera(i, c["width"], c["height"], c["focal"], c["tvec"], c["rquat"])
            for i, c in enumerate(cp)]
    P, aux = tdr.pack_cameras(cams)
    return sc, cams, P, aux


@pytest.fixture(scope="module")
def scene8(tdr, oracle):
    return _scene(tdr, oracle, 8)


def test_camera_derivation_matches_oracle(tdr, oracle):
    sc, cams, P, aux = _scene(tdr, oracle, 6, seed=3)
    for c, p in zip(cams, sc.ring_camera_params(6, 3)):
        o = oracle.camera_derive(p["width"], p["height"], p["focal"], p["tvec"], p["rquat"])
        np.testing.assert_allclose(c.cameraPerspectiveMatrix, o["P"], rtol=1e-14, atol=1e-9)
        assert (c.cx, c.cy) == (o["cx"], o["cy"])
        assert c.fovy == pytest.approx(o["fovy"], rel=1e-15)


@pytest.mark.parametrize("n_cam,n_sys", [(2, 1), (3, 63), (4, 64), (4, 1000), (5, 257), (7, 4097),
                                         (8, 10000), (12, 3000), (16, 2049), (20, 777), (24, 1300), (32, 515)])
@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4])
def test_matrix_batch_vs_oracle(tdr, oracle, n_cam, n_sys, variant):
    sc, cams, P, aux = _scene(tdr, oracle, n_cam, seed=n_cam)
    X = sc.smooth_path(n_sys, seed=n_sys)
    pts = sc.make_detections(sc.project_P(P, X), seed=n_sys)
    eng = tdr.Engine(P, aux)
    eng.set_matrix_variant(variant)
    xyz, err = eng.matrix_triangulate_points(pts, want_err=True)
    xyz_noerr = eng.matrix_triangulate_points(pts)
    rc, ref, ref_err, fb = oracle.matrix_triangulate_points(P, pts, want_err=True)
    assert rc == 0
    rel = sc.rel_err(xyz, ref)
    assert rel <= GATE
    assert rel <= TIGHT
    np.testing.assert_array_equal(xyz, xyz_noerr)  # err on/off must not change the point
    np.testing.assert_allclose(err, ref_err, rtol=1e-9)


def test_matrix_known_answer(tdr, oracle, scene8):
    """Exact projections -> the 3D point itself (implementation-independent truth)."""
    sc, cams, P, aux = scene8
    X = sc.smooth_path(5000, seed=7)
    xyz, err = tdr.Engine(P, aux).matrix_triangulate_points(sc.project_P(P, X), want_err=True)
    assert sc.rel_err(xyz, X) <= 1e-10
    assert err.max() < 1e-6


def test_matrix_c2_full_config(tdr, oracle):
    """BASELINE config 2: 4 cameras x 100k frames, full size, every point against the oracle."""
    sc, cams, P, aux = _scene(tdr, oracle, 4, seed=2)
    X = sc.smooth_path(100_000, seed=2)
    pts = sc.make_detections(sc.project_P(P, X), seed=2)
    xyz = tdr.Engine(P, aux).matrix_triangulate_points(pts)
    rc, ref, _, _ = oracle.matrix_triangulate_points(P, pts, n_threads=8)
    assert rc == 0
    rel = sc.rel_err(xyz, ref)
    print(f"C2 max rel |dp| = {rel:.3e}, max abs = {np.abs(xyz - ref).max():.3e} mm")
    assert rel <= TIGHT


def test_matrix_fewer_point_rows_than_cameras(tdr, oracle, scene8):
    """min(points.size(), cameras.size()) rule, src/MatrixTriangulator.cpp:84."""
    sc, cams, P, aux = scene8
    X = sc.smooth_path(300, seed=1)
    pts = sc.make_detections(sc.project_P(P, X), seed=1)
    eng = tdr.Engine(P, aux)
    xyz = eng.matrix_triangulate_points(pts[:5])
    rc, ref, _, _ = oracle.matrix_triangulate_points(P, pts[:5])
    assert sc.rel_err(xyz, ref) <= TIGHT
    # more point rows than cameras: extra rows ignored
    eng5 = tdr.Engine(P[:5], aux[:5])
    xyz5 = eng5.matrix_triangulate_points(pts)
    np.testing.assert_array_equal(xyz5, xyz)


def test_matrix_too_few_throws_first_index(tdr, oracle, scene8):
    """src/MatrixTriangulator.cpp:92-94: first frame with < 2 valid cameras aborts the batch."""
    sc, cams, P, aux = scene8
    X = sc.smooth_path(2000, seed=5)
    pts = sc.make_detections(sc.project_P(P, X), seed=5)
    pts[1:, 1234] = -1  # one valid camera
    pts[:, 1500] = -1   # none
    pts[0, 77, 0] = -1  # x == -1 alone marks the observation missing; still 7 valid
    eng = tdr.Engine(P, aux)
    with pytest.raises(tdr.TdrError, match="Too few rays are found") as ei:
        eng.matrix_triangulate_points(pts)
    assert ei.value.status == tdr.TDR_E_TOO_FEW
    assert eng.first_bad == 1234
    rc, _, _, fb = oracle.matrix_triangulate_points(P, pts)
    assert rc == oracle.E_TOO_FEW and fb == 1234
    # the context stays usable and clean afterwards
    pts[:, 1234] = pts[:, 1233]
    pts[:, 1500] = pts[:, 1499]
    xyz = eng.matrix_triangulate_points(pts)
    rc, ref, _, _ = oracle.matrix_triangulate_points(P, pts)
    assert rc == 0 and sc.rel_err(xyz, ref) <= TIGHT


def test_matrix_empty_batch(tdr, oracle, scene8):
    sc, cams, P, aux = scene8
    xyz = tdr.Engine(P, aux).matrix_triangulate_points(np.zeros((8, 0, 2)))
    assert xyz.shape == (0, 3)


def test_matrix_single_point_arbitrary_subset(tdr, oracle, scene8):
    """triangulatePoint with CamPointPair semantics (src/MatrixTriangulator.cpp:3-62)."""
    sc, cams, P, aux = scene8
    mt = tdr.MatrixTriangulator(cams)
    assert mt.getType() == "matrix"
    X = sc.smooth_path(20, seed=9)
    uv = sc.make_detections(sc.project_P(P, X), seed=9, dropout_p=0)
    for s, subset in enumerate([[0, 1], [7, 2, 4], [6, 5, 4, 3, 2, 1, 0], list(range(8))]):
        images = [tdr.CamPointPair(cams[c], uv[c, s]) for c in subset]
        Xg, eg = mt.triangulatePoint(images)
        Xo, eo = oracle.matrix_triangulate_point(P[subset], uv[subset, s])
        assert np.abs(Xg - Xo).max() / np.abs(Xo).max() <= TIGHT
        assert eg == pytest.approx(eo, rel=1e-9)


def test_plugin_surface_triangulate_points(tdr, oracle, scene8):
    sc, cams, P, aux = scene8
    mt = tdr.MatrixTriangulator(cams)
    X = sc.smooth_path(500, seed=11)
    pts = sc.make_detections(sc.project_P(P, X), seed=11)
    out = mt.triangulatePoints([[tuple(p) for p in row] for row in pts])
    rc, ref, _, _ = oracle.matrix_triangulate_points(P, pts)
    assert sc.rel_err(out, ref) <= TIGHT
    ragged = [[tuple(p) for p in row] for row in pts]
    ragged[3] = ragged[3][:-1]
    with pytest.raises(tdr.TdrError, match="Every camera should have the same number of points"):
        mt.triangulatePoints(ragged)
    with pytest.raises(tdr.TdrError):
        mt.triangulatePoints([])


# --------------------------------------------------------------------------------------- ray path
@pytest.mark.parametrize("n_cam,n_sys", [(2, 65), (4, 1000), (8, 2000), (16, 1500), (24, 130)])
def test_ray_batch_vs_oracle(tdr, oracle, n_cam, n_sys):
    sc, cams, P, aux = _scene(tdr, oracle, n_cam, seed=40 + n_cam)
    X = sc.smooth_path(n_sys, seed=n_sys)
    pts = sc.make_detections(sc.project_ray_model(aux, X), seed=n_sys)
    eng = tdr.Engine(P, aux)
    g = eng.ray_triangulate_points_ex(pts)
    xyz, err = g["xyz"], g["err"]
    np.testing.assert_array_equal(eng.ray_triangulate_points(pts), xyz)  # asking for the status changes nothing
    o = oracle.ray_triangulate_points_ex(aux, pts, n_threads=8)
    assert o["rc"] == 0
    # A run that stopped away from the minimiser of its own objective (iteration cap / damped step) ends at a
    # chaotic iterate: both sides flag it (TDR_LM_* status) and it is compared on neither.
    ok = oracle.lm_comparable(g["status"], o["status"])
    assert (((g["status"] & 3) != 0) != ((o["status"] & 3) != 0)).mean() < 0.02
    assert ok.mean() > (0.9 if n_cam > 2 else 0.8)
    ref, ref_err = o["xyz"], o["err"]
    nv_ok = ((pts[..., 0] != -1) & (pts[..., 1] != -1)).sum(axis=0)[ok]
    xyz, err, ref, ref_err, pts = xyz[ok], err[ok], ref[ok], ref_err[ok], pts[:, ok]
    # Same schedule, not just the same point.  Near convergence a step lowers S = sum r^2 (~1e4 mm^2) by less
    # than its rounding error, so the last few accept / reject decisions — and with them the exact iteration
    # count — are rounding noise in ANY implementation (measured: identical medians, |difference| <= 1 on 70 %,
    # <= 5 on 96-99 % of the systems).  Compared where J^T J has full rank at the solution (>= 4 rays).
    if (nv_ok >= 4).sum() >= 100:
        gi, oi = g["iters"][ok][nv_ok >= 4], np.abs(o["iters"][ok][nv_ok >= 4])
        print(f"iterations, >= 4 rays: median {np.median(gi)} / {np.median(oi)}, |diff| <= 5 on {(np.abs(gi - oi) <= 5).mean():.4f}")
        # within one iteration; the 24- and 32-camera rigs (medians around 100) get 2 %
        assert abs(np.median(gi) - np.median(oi)) <= (max(1.0, 0.02 * np.median(oi)) if n_cam > 16 else 1.0)
        assert (np.abs(gi - oi) <= 5).mean() >= 0.9
    n_sys = int(ok.sum())
    rel = sc.rel_err(xyz, ref)
    print(f"ray {n_cam}x{n_sys}: max rel {rel:.3e}")
    assert rel <= GATE
    assert rel <= RAY_TIGHT
    # The returned error is the mean ray distance at the solver's LAST compute(), which may be a rejected trial
    # point (src/RayTriangulator.cpp:26,106).  On a run that ended at the minimiser (status 0) that last point is
    # the converged one or a step of < 1e-7 mm away from it, so the value is reproducible for EVERY ray count —
    # tests/test_oracle_cpu.py::test_lm_err_is_reproducible_exactly_where_the_status_word_is_zero shows it on the
    # oracle itself; round 1 compared it for >= 4 rays only.
    np.testing.assert_allclose(err, ref_err, rtol=1e-5, atol=1e-6)
    assert np.isfinite(err).all()
    # independent pin: closed-form minimiser of the same objective — status 0 means the run reached it
    for s in range(0, n_sys, max(1, n_sys // 50)):
        v = pts[:, s, 0] != -1
        v &= pts[:, s, 1] != -1
        cf = oracle.np_ray_closed_form(aux[v], pts[v, s])
        assert np.abs(xyz[s] - cf).max() / np.abs(cf).max() <= 1e-7


def _random_ray_batch(sc, aux, n_sys, seed, dropout):
    X = sc.smooth_path(n_sys, seed=seed)
    return sc.make_detections(sc.project_ray_model(aux, X), seed=seed, dropout_p=dropout, min_valid=2)


def test_ray_lm_status_is_two_sided_on_a_million_random_systems(tdr, oracle):
    """VERDICT r1 item 1.  >= 1 M random 2..32-camera systems (drop-outs up to 80 %, so every ray count from 2
    up occurs) through tdr_ray_triangulate_points_ex and the oracle:
      (i)   EVERY system that both sides report as converged (status 0) agrees to 1e-7 relative — no iteration
            threshold, no tuned constant in the comparison;
      (ii)  GPU and oracle status words agree on >= 99.8 % of this population and on >= 99.9 % of the systems with
            >= 4 rays; the disagreements are 2- and 3-ray systems (0.8 % / 0.2 % of their class), whose J^T J is
            rank-deficient so that the lambda = 0 steps are rounding noise over rounding noise in both
            implementations — the oracle disagrees with a 1-ulp perturbed copy of ITSELF at the same rate
            (tests/test_oracle_cpu.py::test_lm_flagged_systems_are_chaotic);
      (iii) on the converged systems with >= 4 rays the two run the same schedule: iteration counts within 5 of
            each other on >= 95 % (the last accept / reject decisions are rounding noise, see
            test_ray_batch_vs_oracle);
      (iv)  the closed-form entry point agrees to 1e-7 with the oracle's LM result on EVERY system the oracle reports
            as converged, and with the GPU's on every one the GPU does.
    The evidence that flagged systems have no reproducible answer is CPU-side: tests/test_oracle_cpu.py
    test_lm_flagged_systems_are_chaotic and tests/tools/lm_sensitivity.py."""
    plan = [(2, 0.0, 60_000), (3, 0.0, 60_000), (4, 0.0, 100_000), (4, 0.3, 100_000), (5, 0.2, 80_000),
            (6, 0.3, 100_000), (8, 0.05, 100_000), (8, 0.5, 100_000), (12, 0.3, 80_000), (16, 0.05, 60_000),
            (16, 0.6, 100_000), (24, 0.5, 40_000), (32, 0.05, 10_000), (32, 0.8, 40_000)]
    assert sum(p[2] for p in plan) >= 1_000_000
    n_all = n_ok = n_ok4 = n_flag_g = n_flag_o = n_disagree = n_same_iters = 0
    worst = worst_err = worst_cf = 0.0
    n_cf = 0
    by_nv = {}
    only = {}  # ray count -> [flagged by the GPU only, flagged by the oracle only]
    low_iters = {2: ([], []), 3: ([], [])}  # iteration counts (gpu, oracle) of the converged 2- and 3-ray systems
    for n_cam, drop, n_sys in plan:
        sc, cams, P, aux = _scene(tdr, oracle, n_cam, seed=200 + n_cam)
        pts = _random_ray_batch(sc, aux, n_sys, seed=7 * n_cam + int(drop * 10), dropout=drop)
        eng = tdr.Engine(P, aux)
        g = eng.ray_triangulate_points_ex(pts)
        o = oracle.ray_triangulate_points_ex(aux, pts, n_threads=16)
        assert o["rc"] == 0
        fg, fo = (g["status"] & 3) != 0, (o["status"] & 3) != 0
        ok = ~fg & ~fo
        rel = np.abs(g["xyz"][ok] - o["xyz"][ok]).max(axis=1) / np.abs(o["xyz"][ok]).max(axis=1)
        worst = max(worst, float(rel.max()))
        assert rel.max() <= RAY_TIGHT, (n_cam, drop, float(rel.max()))
        # (iv) the closed-form entry point (tdr_ray_closest_point_points, round 5) returns the point every run with
        # status 0 ended at: against the ORACLE's LM result on every system the oracle calls converged, and against
        # the GPU's LM result on every system the GPU calls converged
        cf = eng.ray_closest_point_points(pts)
        for lm_xyz, conv in ((o["xyz"], ~fo), (g["xyz"], ~fg)):
            rel_cf = np.abs(cf[conv] - lm_xyz[conv]).max(axis=1) / np.abs(lm_xyz[conv]).max(axis=1)
            worst_cf = max(worst_cf, float(rel_cf.max()))
            assert rel_cf.max() <= RAY_TIGHT, (n_cam, drop, float(rel_cf.max()))
        n_cf += int((~fo).sum())
        # the cap bit is exact on both sides: iters == 1000 <=> TDR_LM_CAP
        assert np.array_equal((g["status"] & 1) != 0, g["iters"] >= 1000)
        nv = ((pts[..., 0] != -1) & (pts[..., 1] != -1)).sum(axis=0)
        for k in np.unique(nv):
            m = nv == k
            a = by_nv.setdefault(int(k), [0, 0])
            a[0] += int(m.sum())
            a[1] += int((fg[m] != fo[m]).sum())
            b = only.setdefault(int(k), [0, 0])
            b[0] += int((fg[m] & ~fo[m]).sum())
            b[1] += int((fo[m] & ~fg[m]).sum())
        for k in (2, 3):
            m = ok & (nv == k)
            low_iters[k][0].append(g["iters"][m])
            low_iters[k][1].append(np.abs(o["iters"][m]))
        n_all += n_sys
        n_ok += int(ok.sum())
        n_flag_g += int(fg.sum())
        n_flag_o += int(fo.sum())
        n_disagree += int((fg != fo).sum())
        worst_err = max(worst_err, float((np.abs(g["err"][ok] - o["err"][ok]) / np.maximum(o["err"][ok], 1e-3)).max()))
        n_same_iters += int((np.abs(g["iters"][ok & (nv >= 4)] - np.abs(o["iters"][ok & (nv >= 4)])) <= 5).sum())
        n_ok4 += int((ok & (nv >= 4)).sum())
    print(f"{n_all} systems: flagged gpu {n_flag_g} / oracle {n_flag_o}, flags differ on {n_disagree} "
          f"({n_disagree / n_all:.2e}); unflagged max rel {worst:.3e}, max rel err-value difference {worst_err:.2e}; "
          f"iteration counts within 5 on "
          f"{n_same_iters / n_ok4:.4f} (>= 4 rays); per ray count (n, disagreements): {sorted(by_nv.items())[:8]}")
    print(f"closed form vs LM where status == 0: {n_cf} oracle-converged systems, max rel {worst_cf:.3e}")
    assert n_disagree / n_all <= 2e-3
    # DIRECTION of the disagreements.  Round 2 solved the trial steps by LDL^T only: on a
    # rank-deficient J^T J (two rays: always) that gives a NaN step where the reference's DECOMP_EIG takes the
    # minimum-norm one, and the GPU flagged ~900 systems the oracle did not against ~465 the other way round — twelve
    # sigma off a fair split, all of it in the 2- and 3-ray classes.  With the eigen solve restated in the kernel
    # (sym3_eig_pinv, tdr_kernels.hip) the disagreements that remain are rounding noise, and rounding noise has no
    # preferred side: GPU-only and oracle-only counts must be a fair coin's split, overall and in the 2- and 3-ray
    # classes separately, and the flagged totals must agree to 1 %.
    g_only = sum(v[0] for v in only.values())
    o_only = sum(v[1] for v in only.values())
    print(f"direction: gpu-only {g_only} / oracle-only {o_only}; 2 rays {only.get(2)}, 3 rays {only.get(3)}, "
          f">= 4 rays {[sum(v[i] for k, v in only.items() if k >= 4) for i in (0, 1)]}")
    assert g_only + o_only == n_disagree
    assert abs(g_only - o_only) <= 3.0 * np.sqrt(g_only + o_only) + 1, (g_only, o_only)
    for k in (2, 3):
        gk, ok_ = only[k]
        assert abs(gk - ok_) <= 3.0 * np.sqrt(gk + ok_) + 1, (k, gk, ok_)
    assert abs(n_flag_g - n_flag_o) <= 0.01 * n_flag_o, (n_flag_g, n_flag_o)
    # ... and the 2- and 3-ray systems now run the reference's SCHEDULE, not just to its answer: with the eigen solve
    # their iteration counts have the oracle's median and stay within 20 of it on >= 85 % (two rays creep for ~100
    # iterations through accept / reject decisions that are rounding noise) / 93 % (three rays).  Round 2's LDL^T
    # step left a two-ray run after two passes.
    for k, within in ((2, 0.85), (3, 0.93)):
        gi, oi = np.concatenate(low_iters[k][0]), np.concatenate(low_iters[k][1])
        frac = float((np.abs(gi - oi) <= 20).mean())
        print(f"{k} rays, both converged: {len(gi)} systems, median iterations {np.median(gi)} / {np.median(oi)}, "
              f"within 20 on {frac:.4f}, identical on {float((gi == oi).mean()):.4f}")
        assert abs(np.median(gi) - np.median(oi)) <= max(1.0, 0.02 * np.median(oi))
        assert frac >= within, (k, frac)
    n4 = sum(n for k, (n, d) in by_nv.items() if k >= 4)
    d4 = sum(d for k, (n, d) in by_nv.items() if k >= 4)
    assert d4 / n4 <= 1e-3, (n4, d4)
    for k, (n, d) in by_nv.items():
        if n >= 1000:
            assert d / n <= 1.5e-2, (k, n, d)
    assert n_same_iters / n_ok4 >= 0.95
    assert worst_err <= 1e-4  # the `err` output too, for every ray count (2 and 3 included), where status == 0
    assert 0 < n_flag_g < 0.1 * n_all  # the population does contain flagged systems, and they are a minority


def test_device_eigen_solve_matches_the_oracle(tdr, oracle):
    """The device restatement of cv::solve / cv::invert(DECOMP_EIG) (sym3_eig_pinv, the LM loop's solve on (nearly)
    rank-deficient J^T J) through the test hook tdr_debug_sym3_eig_device, against the oracle's sym3_eig_solve and
    against LAPACK: regular matrices (the plain inverse), rank 2 (two rays: the minimum-norm solution, the null
    direction dropped), rank 1, the zero matrix, graded scalings, and a NaN (the loop must end and poison only its
    own lane)."""
    import ctypes as C
    sc, cams, P, aux = _scene(tdr, oracle, 4, seed=33)
    eng = tdr.Engine(P, aux)
    rng = np.random.default_rng(5)
    rows, ranks = [], []
    for k in range(4000):
        r = int(rng.integers(1, 5))
        G = rng.standard_normal((r, 3)) * 10.0 ** rng.integers(-3, 4)
        M = G.T @ G
        b = M @ rng.standard_normal(3)
        rows.append([M[0, 0], M[0, 1], M[0, 2], M[1, 1], M[1, 2], M[2, 2], *b])
        ranks.append(min(r, 3))
    rows.append([0.0] * 9)
    ranks.append(0)
    rows.append([float("nan"), 0, 0, 1, 0, 1, 1, 1, 1])
    ranks.append(-1)
    rows = np.array(rows)
    got = eng.debug_sym3_eig(rows)
    f = oracle.lib().oracle_sym3_eig_solve
    f.argtypes = [C.POINTER(C.c_double)] * 4
    dp = C.POINTER(C.c_double)
    worst = 0.0
    for i, (row, rank) in enumerate(zip(rows, ranks)):
        if rank < 0:
            assert np.isnan(got[i]).any()
            continue
        M = np.array([[row[0], row[1], row[2]], [row[1], row[3], row[4]], [row[2], row[4], row[5]]])
        b = np.ascontiguousarray(row[6:9])
        x, inv = np.zeros(3), np.zeros(9)
        f(np.ascontiguousarray(M).ctypes.data_as(dp), b.ctypes.data_as(dp), x.ctypes.data_as(dp), inv.ctypes.data_as(dp))
        scale = max(np.abs(x).max(), 1e-300)
        worst = max(worst, float(np.abs(got[i, :3] - x).max() / scale) if rank else float(np.abs(got[i, :3]).max()))
        md = np.abs(np.diag(inv.reshape(3, 3))).max()
        assert abs(got[i, 3] - md) <= 1e-10 * max(md, 1e-300), (i, rank)
        if rank == 3:  # ... and it is the plain solve there
            assert np.abs(got[i, :3] - np.linalg.solve(M, b)).max() <= 1e-6 * scale
        elif rank in (1, 2):  # minimum-norm solution: LAPACK's pseudo-inverse with an explicit rank
            ref = np.linalg.pinv(M, rcond=1e-10, hermitian=True) @ b
            assert np.abs(got[i, :3] - ref).max() <= 1e-7 * max(np.abs(ref).max(), 1e-300), (i, rank)
    print(f"device eigen solve vs oracle: max rel {worst:.2e}")
    assert worst <= 1e-10
    assert np.isfinite(got[:-1]).all()  # the NaN lane did not touch its neighbours


def test_device_cascade_matches_its_cpu_restatement_and_the_eigen_solve(tdr, oracle):
    """Round 4: the LM kernels answer cv::solve / cv::invert(DECOMP_EIG) on a (nearly) rank-deficient J^T J WITHOUT an
    eigen-decomposition — the rank is decided on tr / e2 / det at the reference's 2 eps sum|w| cut and each rank has a
    closed form (lm_cascade).  Through the test hook tdr_debug_lm_cascade_device, on matrices of every rank and
    conditioning, as a two-ray system meets them (right-hand side J^T r in the range, the rank-2 solution taken as it
    is) and as a three-ray system does (a small genuine null component in the right-hand side, the rank-2 solution
    cleared of it):
      (a) bit-identical to the oracle's restatement of the same cascade (sym3_cascade — the form
          tests/tools/lm_hybrid_rule.py measured against the reference's Jacobi form at the level of whole LM runs:
          profiles/r04_lm_hybrid_rule.json);
      (b) the same RANK as constructed, and the solution / the inverse's diagonal within a few eps x cond of the
          oracle's Jacobi DECOMP_EIG restatement and of the device's own (tdr_debug_sym3_eig_device);
      (c) zero and NaN matrices end in the zero step (the loop then stops, as the reference's does)."""
    import ctypes as C
    sc, cams, P, aux = _scene(tdr, oracle, 4, seed=33)
    eng = tdr.Engine(P, aux)
    dp = C.POINTER(C.c_double)
    f = oracle.lib().oracle_sym3_cascade
    f.argtypes = [dp, dp, C.c_int, dp, dp]
    f.restype = C.c_int
    g = oracle.lib().oracle_sym3_eig_solve
    g.argtypes = [dp] * 4
    worst = {}
    for n_rays in (2, 3):
        rng = np.random.default_rng(11 + n_rays)
        rows, ranks = [], []
        for k in range(6000):
            r = int(rng.integers(1, 4))
            if r == 2 and k % 3 == 0 and n_rays == 2:  # two nearly parallel gradients: rank 2 with cond up to 1e9
                g1 = rng.standard_normal(3)
                G = np.stack([g1, g1 + 10.0 ** rng.uniform(-4.5, -1) * rng.standard_normal(3)])
            else:
                G = rng.standard_normal((r if r < 3 else int(rng.integers(3, 6)), 3))
            G = G * 10.0 ** rng.integers(-3, 4)
            M = G.T @ G
            b = G.T @ (rng.standard_normal(G.shape[0]) * 50.0)  # J^T r: in the range
            if r == 2 and n_rays == 3:
                b = b + np.abs(M).max() * 1e-7 * rng.standard_normal(3)  # ... three almost coplanar gradients: nearly so
            rows.append([M[0, 0], M[0, 1], M[0, 2], M[1, 1], M[1, 2], M[2, 2], *b])
            ranks.append(r)
        rows.append([0.0] * 9)
        ranks.append(0)
        rows = np.array(rows)
        eig = eng.debug_sym3_eig(rows)
        got = eng.debug_lm_cascade(rows, n_rays)
        for i, (row, rank) in enumerate(zip(rows, ranks)):
            M = np.ascontiguousarray([[row[0], row[1], row[2]], [row[1], row[3], row[4]], [row[2], row[4], row[5]]])
            b = np.ascontiguousarray(row[6:9])
            x, md, xe, inv = np.zeros(3), np.zeros(1), np.zeros(3), np.zeros(9)
            rk = f(M.ctypes.data_as(dp), b.ctypes.data_as(dp), n_rays, x.ctypes.data_as(dp), md.ctypes.data_as(dp))
            assert rk == rank == int(got[i, 4]), (i, rank, rk, got[i, 4])                      # (b) the rank
            assert np.array_equal(got[i, :3], x) and got[i, 3] == md[0], (i, rank, got[i], x, md)  # (a) same operations
            if rank == 0:
                assert not got[i, :4].any()
                continue
            g(M.ctypes.data_as(dp), b.ctypes.data_as(dp), xe.ctypes.data_as(dp), inv.ctypes.data_as(dp))
            w = np.linalg.eigvalsh(M)
            cond = w[2] / w[3 - rank]
            tol = 20 * 2.3e-16 * cond + 1e-14
            scale = max(np.abs(xe).max(), 1e-300)
            key = (n_rays, rank)
            worst[key] = max(worst.get(key, 0.0), float(np.abs(got[i, :3] - xe).max() / scale / tol),
                             float(np.abs(got[i, :3] - eig[i, :3]).max() / scale / tol))
            mde = np.abs(np.diag(inv.reshape(3, 3))).max()
            assert abs(got[i, 3] - mde) <= 50 * tol * mde and abs(got[i, 3] - eig[i, 3]) <= 50 * tol * mde, (i, rank)
    print(f"device cascade vs the Jacobi form (CPU and device), in units of 20 eps x cond, by (rays, rank): {worst}")
    assert max(worst.values()) <= 1.0
    nan_row = np.array([[float("nan"), 0, 0, 1, 0, 1, 1, 1, 1], [1.0, 0, 0, 1, 0, 1, 1, 1, 1]])
    gn = eng.debug_lm_cascade(nan_row, 2)
    assert not gn[0, :4].any() and gn[0, 4] == 0  # (c) a NaN matrix: the zero step, its neighbour untouched
    assert gn[1, 4] == 3 and np.array_equal(gn[1, :3], [1.0, 1.0, 1.0])


def test_ray_coplanar_rays_follow_the_reference_through_the_pseudo_inverse(tdr, oracle):
    """Four cameras with COLLINEAR centres and exact pixels: the rays of a system lie in one plane (the line of
    centres and the point), the start point (mean of the centres) and every LM iterate with them, so J^T J has rank 2
    at every iterate although there are four rays — a factorisation has nothing to solve at lambda = 0, the
    reference's DECOMP_EIG takes the in-plane (minimum-norm) step (src/RayTriangulator.cpp:100-104).  GPU and oracle
    behave alike; run as a large batch (lane-per-system kernel + parked launch)
    and as a small one (camera-per-lane kernel with the eigen solve inlined), bit-identical between the two."""
    sc = __import__("importlib").import_module(tdr.__name__ + ".scenes")
    c0 = sc.ring_camera_params(4, seed=1)[0]
    C0 = np.asarray(c0["tvec"], dtype=np.float64)
    view = -C0 / np.linalg.norm(C0)                       # the ring cameras look at the origin
    side = np.cross(view, [0.0, 0.0, 1.0])
    side /= np.linalg.norm(side)
    centres = [C0 + t * side for t in (-600.0, -150.0, 250.0, 700.0)]
    cams = [tdr.createCamera(i, c0["width"], c0["height"], c0["focal"], C, c0["rquat"]) for i, C in enumerate(centres)]
    P, aux = tdr.pack_cameras(cams)
    rng = np.random.default_rng(8)
    n = 3000
    X = rng.uniform(-250.0, 250.0, size=(n, 3)) + [0.0, 0.0, 900.0]
    pts = sc.project_ray_model(aux, X)                    # exact pixels: every ray passes through its X
    assert np.isfinite(pts).all() and (pts[..., 0] > 0).all() and (pts[..., 0] < c0["width"]).all()
    eng = tdr.Engine(P, aux)
    tdr.set_knob("TDR_LM_COOP_MAX", 0)
    try:
        big = eng.ray_triangulate_points_ex(pts)          # lane-per-system kernel, these systems are parked
    finally:
        tdr.set_knob("TDR_LM_COOP_MAX", None)
    small = eng.ray_triangulate_points_ex(pts)            # camera-per-lane kernel
    np.testing.assert_array_equal(big["xyz"], small["xyz"])
    np.testing.assert_array_equal(big["iters"], small["iters"])
    o = oracle.ray_triangulate_points_ex(aux, pts, n_threads=8)
    ok = oracle.lm_comparable(big["status"], o["status"])
    fg, fo = (big["status"] & 3) != 0, (o["status"] & 3) != 0
    print(f"coplanar rays: both converged on {ok.mean():.3f}; gpu flagged {fg.sum()}, oracle {fo.sum()}, flags differ on "
          f"{(fg != fo).sum()}; median iterations {np.median(big['iters'])} / {np.median(np.abs(o['iters']))}")
    # The reference's solver is bad at this geometry — it creeps inside the plane and mostly runs into its iteration
    # cap — and the kernel is bad at it in the same way: the same share of runs flagged, the same systems, the same
    # iteration counts; where a run does end at the minimiser both sides have the known point.
    assert abs(int(fg.sum()) - int(fo.sum())) <= 0.03 * n
    g_only, o_only = int((fg & ~fo).sum()), int((fo & ~fg).sum())
    assert (fg != fo).mean() <= 0.12  # runs that end within a few iterations of the cap: rounding decides the side ...
    assert abs(g_only - o_only) <= 3.0 * np.sqrt(g_only + o_only) + 1, (g_only, o_only)  # ... and it has no favourite
    assert abs(np.median(big["iters"]) - np.median(np.abs(o["iters"]))) <= 0.05 * np.median(np.abs(o["iters"]))
    assert ok.sum() >= 20
    assert sc.rel_err(big["xyz"][ok], o["xyz"][ok]) <= RAY_TIGHT
    assert sc.rel_err(big["xyz"][ok], X[ok]) <= 1e-6      # the known answer


def test_ray_known_answer(tdr, oracle):
    sc, cams, P, aux = _scene(tdr, oracle, 8, seed=21)
    X = sc.smooth_path(2000, seed=3)
    xyz = tdr.Engine(P, aux).ray_triangulate_points(sc.project_ray_model(aux, X))
    assert sc.rel_err(xyz, X) <= 1e-7


def test_ray_too_few_and_shape(tdr, oracle):
    sc, cams, P, aux = _scene(tdr, oracle, 4, seed=22)
    X = sc.smooth_path(300, seed=4)
    pts = sc.make_detections(sc.project_ray_model(aux, X), seed=4)
    pts[:3, 200] = -1
    eng = tdr.Engine(P, aux)
    with pytest.raises(tdr.TdrError, match="Too few detections are found"):
        eng.ray_triangulate_points(pts)
    assert eng.first_bad == 200
    # more point rows than cameras is UB in the reference (src/RayTriangulator.cpp:65,69): rejected
    eng3 = tdr.Engine(P[:3], aux[:3])
    with pytest.raises(tdr.TdrError) as ei:
        eng3.ray_triangulate_points(pts)
    assert ei.value.status == tdr.TDR_E_SHAPE


def test_ray_single_point_and_dist_from_ray(tdr, oracle):
    sc, cams, P, aux = _scene(tdr, oracle, 6, seed=23)
    rt = tdr.RayTriangulator(cams)
    assert rt.getType() == "ray"
    X = sc.smooth_path(4, seed=5)
    uv = sc.make_detections(sc.project_ray_model(aux, X), seed=5, dropout_p=0)
    for s, subset in enumerate([[0, 3], [5, 1, 2], list(range(6))]):
        images = [tdr.CamPointPair(cams[c], uv[c, s]) for c in subset]
        Xg, eg = rt.triangulatePoint(images)
        Xo, eo, _ = oracle.ray_triangulate_point(aux[subset], uv[subset, s])
        assert np.abs(Xg - Xo).max() / np.abs(Xo).max() <= RAY_TIGHT
        if len(subset) >= 4:  # <= 3 rays: rank-2 J^T J at the optimum, see test_ray_batch_vs_oracle
            assert eg == pytest.approx(eo, rel=1e-6, abs=1e-6)
    d = rt.getDistFromRay(tdr.CamPointPair(cams[2], uv[2, 0]), X[1])
    assert d == pytest.approx(oracle.get_dist_from_ray(aux[2], uv[2, 0, 0], uv[2, 0, 1], X[1]), rel=1e-12)


# --------------------------------------------------------------------------------------- device-resident
def test_device_resident_synth_and_shards(tdr, oracle, scene8):
    """Device-generated stream: sampled systems against the oracle; shard consistency
    (two half-launches on strided views == one launch); determinism."""
    sc, cams, P, aux = scene8
    eng = tdr.Engine(P, aux)
    n = 1_000_003
    d_pts = eng.alloc(8 * n * 16)
    d_xyz = eng.alloc(n * 24)
    d_xyz2 = eng.alloc(n * 24)
    d_truth = eng.alloc(n * 24)
    eng.synth_device(0xC0FFEE + 3, 1, 0, n, n, d_pts.ptr, d_truth.ptr)
    eng.matrix_device(d_pts.ptr, 8, n, n, d_xyz.ptr)
    h = n // 2 + 5
    eng.matrix_device(d_pts.ptr, 8, n, h, d_xyz2.ptr)
    eng.matrix_device(d_pts.ptr + h * 16, 8, n, n - h, d_xyz2.ptr + h * 24)
    assert eng.fetch_first_bad() == -1
    xyz = d_xyz.download((n, 3))
    xyz2 = d_xyz2.download((n, 3))
    np.testing.assert_array_equal(xyz, xyz2)
    pts = d_pts.download((8, n, 2))
    truth = d_truth.download((n, 3))
    assert np.isfinite(xyz).all()
    valid = ((pts[..., 0] != -1) & (pts[..., 1] != -1)).sum(axis=0)
    assert valid.min() >= 2 and 0.03 < (valid < 8).mean() < 0.5
    assert np.array_equal(pts, np.rint(pts))  # integer-valued doubles like the CSV data
    idx = np.unique(np.concatenate([np.arange(0, n, 997), [n - 1, n - 2, 255, 256, 257]]))
    rc, ref, _, _ = oracle.matrix_triangulate_points(P, pts[:, idx])
    assert rc == 0
    assert sc.rel_err(xyz[idx], ref) <= TIGHT
    # 2 px noise, 8 cameras at ~5 m: a few cm from the truth
    assert np.abs(xyz - truth).max() < 100.0


# --------------------------------------------------------------------------------------- single-process multi-GPU
def test_multi_context_shards_host_batches(tdr, oracle, scene8):
    """tdr_create_multi: host batches block-sharded over sub-contexts by host threads.  On a one-GPU
    box the shard list repeats ordinal 0 (three 'virtual' GPUs): same sharding code, same results."""
    sc, cams, P, aux = scene8
    X = sc.smooth_path(10_007, seed=13)
    pts = sc.make_detections(sc.project_P(P, X), seed=13)
    one = tdr.Engine(P, aux)
    multi = tdr.Engine(P, aux, devices=[0, 0, 0])
    xyz1, err1 = one.matrix_triangulate_points(pts, want_err=True)
    xyz3, err3 = multi.matrix_triangulate_points(pts, want_err=True)
    np.testing.assert_array_equal(xyz1, xyz3)  # sharding must not change a bit
    np.testing.assert_array_equal(err1, err3)
    rc, ref, _, _ = oracle.matrix_triangulate_points(P, pts)
    assert sc.rel_err(xyz3, ref) <= TIGHT
    # the globally first failing system is reported, whichever shard it falls in
    bad = pts.copy()
    bad[1:, 9000] = -1
    bad[1:, 4000] = -1
    with pytest.raises(tdr.TdrError, match="Too few rays are found"):
        multi.matrix_triangulate_points(bad)
    assert multi.first_bad == 4000
    # ray path and the tiny / empty cases
    ptsr = sc.make_detections(sc.project_ray_model(aux, X[:500]), seed=14)
    np.testing.assert_array_equal(one.ray_triangulate_points(ptsr), multi.ray_triangulate_points(ptsr))
    assert multi.matrix_triangulate_points(pts[:, :2]).shape == (2, 3)
    assert multi.matrix_triangulate_points(pts[:, :0]).shape == (0, 3)
    # device pointers belong to one GPU: the device entry points refuse a multi-GPU context
    buf = one.alloc(1024)
    with pytest.raises(tdr.TdrError) as ei:
        multi.matrix_device(buf.ptr, 8, 4, 4, buf.ptr)
    assert ei.value.status == tdr.TDR_E_ARG
    # device = -1: every visible GPU
    allgpu = tdr.Engine(P, aux, device=-1)
    np.testing.assert_array_equal(allgpu.matrix_triangulate_points(pts), xyz1)


def _rank_deficient_cases(tdr, oracle, sc):
    """(name, P [n,12], aux, uv [n,2]) of geometries whose 2N x 3 system has rank 2: the reference's
    cv::invert(DECOMP_SVD) (src/MatrixTriangulator.cpp:52-54) returns the minimum-norm point for them."""
    cp = sc.ring_camera_params(4, seed=1)
    c0 = cp[0]
    X = np.array([[300.0, -200.0, 1500.0]])
    C0 = np.asarray(c0["tvec"], dtype=np.float64)

    def cams_at(centres):
        cams = [tdr.createCamera(i, c0["width"], c0["height"], c0["focal"], C, c0["rquat"]) for i, C in enumerate(centres)]
        return tdr.pack_cameras(cams)

    out = []
    P, aux = cams_at([C0, C0])  # the same camera twice, same pixel
    uv = sc.project_P(P, X)[:, 0, :]
    out.append(("duplicate camera, exact pixel", P, aux, uv))
    out.append(("duplicate camera, integer pixel", P, aux, np.rint(uv)))
    P, aux = cams_at([C0, C0 - 0.3 * (X[0] - C0)])  # second camera further back on the first one's line of sight
    out.append(("two cameras seeing the same ray", P, aux, sc.project_P(P, X)[:, 0, :]))
    P, aux = cams_at([C0, C0 - 0.3 * (X[0] - C0), C0 - 0.6 * (X[0] - C0)])
    out.append(("three collinear centres through the point", P, aux, sc.project_P(P, X)[:, 0, :]))
    return out


def test_matrix_rank_deficient_returns_the_reference_pinv_answer(tdr, oracle, scene8):
    """rank-deficient systems get the reference's Moore-Penrose (minimum-norm) answer from
    the one-sided Jacobi slow path, on every route (batch kernel, single-system kernel, int16 format), next to
    ordinary systems of the same launch."""
    sc = scene8[0]
    for name, P, aux, uv in _rank_deficient_cases(tdr, oracle, sc):
        Xo, eo = oracle.matrix_triangulate_point(P, uv)
        Xl, _ = oracle.np_matrix_triangulate_point(P.reshape(-1, 3, 4), uv)
        assert np.abs(Xo - Xl).max() / np.abs(Xl).max() <= 1e-6, name  # oracle == LAPACK pinv on this case
        eng = tdr.Engine(P, aux)
        # a batch in which the degenerate system sits between regular ones (so fast and slow lanes share a wave)
        n = 130
        pts = np.repeat(uv[:, None, :], n, axis=1)
        if P.shape[0] >= 2:
            pts[1, ::2] += [37.0, -11.0]  # every other system: a different pixel in camera 1 -> full rank
        xyz, err = eng.matrix_triangulate_points(pts, want_err=True)
        rc, ref, ref_err, _ = oracle.matrix_triangulate_points(P, pts, want_err=True)
        assert rc == 0
        rel = np.abs(xyz - ref).max(axis=1) / np.abs(ref).max(axis=1)
        print(f"{name}: deficient lanes max rel {rel[1::2].max():.2e}, regular lanes {rel[::2].max():.2e}")
        assert np.isfinite(xyz).all()
        assert rel[1::2].max() <= 1e-6, name
        # the regular neighbours of this rig are themselves badly conditioned (two nearly identical views):
        # they are inside the north_star gate, the deficient ones are the point of the test
        assert rel[::2].max() <= GATE, name
        np.testing.assert_allclose(err[1::2], ref_err[1::2], rtol=1e-5, atol=1e-6)
        Xs, es = eng.matrix_triangulate_point(P, uv)  # triangulatePoint route
        assert np.abs(Xs - Xo).max() / np.abs(Xo).max() <= 1e-6, name
        if np.array_equal(uv, np.rint(uv)):
            x16 = eng.matrix_triangulate_points_i16(np.repeat(uv[:, None, :], 5, axis=1).astype(np.int16))
            assert np.abs(x16 - Xo).max() / np.abs(Xo).max() <= 1e-6, name


def test_matrix_rank_deficient_systems_inside_a_large_batch(tdr, oracle, scene8):
    """The cold path must not disturb its neighbours: 20 000 regular 8-camera systems with degenerate ones
    planted (all cameras but a duplicated pair dropped) — regular systems bit-identical to a launch without
    them, planted ones equal to the oracle's pinv answer."""
    sc, cams, P, aux = scene8
    P2 = np.concatenate([P, P[:1]])  # camera 8 duplicates camera 0
    aux2 = np.concatenate([aux, aux[:1]])
    n = 20_000
    X = sc.smooth_path(n, seed=77)
    pts = sc.make_detections(sc.project_P(P2, X), seed=77, dropout_p=0.0)
    eng = tdr.Engine(P2, aux2)
    base = eng.matrix_triangulate_points(pts)
    planted = np.arange(64, n, 997)
    mod = pts.copy()
    mod[1:8, planted] = -1          # only cameras 0 and 8 (the same camera) remain ...
    mod[8, planted] = mod[0, planted]  # ... with the same pixel: rank 2
    got = eng.matrix_triangulate_points(mod)
    keep = np.ones(n, dtype=bool)
    keep[planted] = False
    np.testing.assert_array_equal(got[keep], base[keep])
    rc, ref, _, _ = oracle.matrix_triangulate_points(P2, mod[:, planted])
    assert rc == 0
    assert sc.rel_err(got[planted], ref) <= 1e-6


def test_c_abi_argument_validation(tdr, scene8):
    """Nothing throws or faults across the C boundary: bad arguments come back as status codes."""
    import ctypes as C
    sc, cams, P, aux = scene8
    L = tdr.lib()
    eng = tdr.Engine(P, aux)
    ctx = eng._ctx
    xyz = np.zeros((4, 3))
    pts = np.zeros((8, 4, 2))
    dp = C.POINTER(C.c_double)
    fb = C.c_int64(0)
    f = L.tdr_matrix_triangulate_points
    assert f(None, pts.ctypes.data_as(dp), 8, 4, xyz.ctypes.data_as(dp), None, C.byref(fb)) == tdr.TDR_E_ARG
    assert f(ctx, None, 8, 4, xyz.ctypes.data_as(dp), None, C.byref(fb)) == tdr.TDR_E_ARG
    assert f(ctx, pts.ctypes.data_as(dp), 8, 4, None, None, C.byref(fb)) == tdr.TDR_E_ARG
    assert f(ctx, pts.ctypes.data_as(dp), 0, 4, xyz.ctypes.data_as(dp), None, C.byref(fb)) == tdr.TDR_E_SHAPE
    assert f(ctx, pts.ctypes.data_as(dp), 8, -1, xyz.ctypes.data_as(dp), None, C.byref(fb)) == tdr.TDR_E_SHAPE
    # device entry points: misaligned detection pointer, stride shorter than the batch
    buf = eng.alloc(4096)
    g = L.tdr_matrix_triangulate_points_device
    assert g(ctx, buf.ptr + 8, 8, 4, 4, buf.ptr + 2048, None, None) == tdr.TDR_E_ARG
    assert g(ctx, buf.ptr, 8, 3, 4, buf.ptr + 2048, None, None) == tdr.TDR_E_SHAPE
    # a ray call on a context created without the ray constants
    eng_m = tdr.Engine(P, None)
    assert L.tdr_ray_triangulate_points(eng_m._ctx, pts.ctypes.data_as(dp), 8, 4, xyz.ctypes.data_as(dp), None,
                                        C.byref(fb)) == tdr.TDR_E_ARG
    # context creation
    out = C.c_void_p()
    assert L.tdr_create(P.ctypes.data_as(dp), None, 0, 0, C.byref(out)) == tdr.TDR_E_ARG
    assert L.tdr_create(P.ctypes.data_as(dp), None, 33, 0, C.byref(out)) == tdr.TDR_E_ARG
    assert L.tdr_create(P.ctypes.data_as(dp), None, 8, 9999, C.byref(out)) == tdr.TDR_E_ARG
    # single system: sentinel observation and too many observations are rejected, not reproduced
    X = np.zeros(3)
    uv = np.array([[-1.0, 5.0], [3.0, 4.0]])
    assert L.tdr_matrix_triangulate_point(ctx, P.ctypes.data_as(dp), uv.ctypes.data_as(dp), 2,
                                          X.ctypes.data_as(dp), None) == tdr.TDR_E_SHAPE
    assert L.tdr_matrix_triangulate_point(ctx, P.ctypes.data_as(dp), uv.ctypes.data_as(dp), 33,
                                          X.ctypes.data_as(dp), None) == tdr.TDR_E_SHAPE
    # and the context still works
    Xs = sc.smooth_path(4, seed=1)
    assert sc.rel_err(eng.matrix_triangulate_points(sc.project_P(P, Xs)), Xs) <= 1e-10


def test_row_pointer_entry_points(tdr, oracle, scene8):
    """tdr_*_triangulate_points_rows: one pointer per camera row (separate allocations, like the inner
    vectors of the reference's argument) == the contiguous-block entry point, incl. through shards."""
    sc, cams, P, aux = scene8
    X = sc.smooth_path(3001, seed=17)
    pts = sc.make_detections(sc.project_P(P, X), seed=17)
    rows = [pts[c].copy() for c in range(8)]
    eng = tdr.Engine(P, aux)
    np.testing.assert_array_equal(eng.triangulate_points_rows(rows), eng.matrix_triangulate_points(pts))
    multi = tdr.Engine(P, aux, devices=[0, 0])
    np.testing.assert_array_equal(multi.triangulate_points_rows(rows), eng.matrix_triangulate_points(pts))
    ptsr = sc.make_detections(sc.project_ray_model(aux, X[:400]), seed=18)
    np.testing.assert_array_equal(eng.triangulate_points_rows([ptsr[c].copy() for c in range(8)], ray=True),
                                  eng.ray_triangulate_points(ptsr))
    rows[3][2500] = -1
    for c in (0, 1, 2, 4, 5, 6):
        rows[c][2500] = -1
    with pytest.raises(tdr.TdrError, match="Too few rays are found"):
        multi.triangulate_points_rows(rows)
    assert multi.first_bad == 2500


def test_matrix_ill_conditioned_geometry_stays_inside_the_gate(tdr, oracle):
    """Normal equations square the condition number of A.  Narrow-baseline rigs (all cameras within a
    few centimetres to decimetres of each other, 5 m from the target) push kappa(A) to 1e2..1e4; the fp64
    kernel must still sit orders of magnitude inside the 1e-4 gate against the SVD-based oracle."""
    sc = __import__("importlib").import_module(tdr.__name__ + ".scenes")
    base = sc.ring_camera_params(1, seed=50)[0]
    worst = 0.0
    for baseline in (1000.0, 100.0, 20.0):
        cp = []
        for i in range(4):
            c = dict(base)
            off = np.array([np.cos(i * 1.7), np.sin(i * 2.3), np.cos(i * 0.9)]) * baseline
            c["tvec"] = base["tvec"] + off
            cp.append(c)
        cams = [tdr.createCamera(i, c["width"], c["height"], c["focal"], c["tvec"], c["rquat"])
                for i, c in enumerate(cp)]
        P, aux = tdr.pack_cameras(cams)
        X = sc.smooth_path(2000, seed=int(baseline)) * 0.3 + np.array([0, 0, 1000.0])
        pts = sc.make_detections(sc.project_P(P, X), seed=3, dropout_p=0.0)
        A, _ = oracle.np_build_system(P, pts[:, 0])
        kappa = np.linalg.cond(A)
        xyz = tdr.Engine(P, aux).matrix_triangulate_points(pts)
        rc, ref, _, _ = oracle.matrix_triangulate_points(P, pts)
        rel = sc.rel_err(xyz, ref)
        print(f"baseline {baseline:7.1f} mm: kappa(A) = {kappa:9.3e}, max rel |dp| = {rel:.3e}")
        worst = max(worst, rel)
        assert rel <= 1e-4 * 1e-3  # three orders inside the gate even at a 2 cm baseline
    assert worst > 0


@pytest.mark.parametrize("shards,transport", [([0, 0, 0], 2), ([0, 0, 0], 0), ([0], 1)])
def test_multi_context_device_resident_gather(tdr, oracle, scene8, shards, transport):
    """tdr_multi_matrix_triangulate_gather: every shard triangulates its slice on its GPU, then the single
    gather of the path onto shard 0.  One-GPU box: three 'virtual' shards exercise the sharding + the
    peer-copy transport; a one-shard context exercises the RCCL transport end to end (comm init, grouped
    send/recv to itself) — the 8-GPU form differs only in the number of ranks."""
    sc, cams, P, aux = scene8
    multi = tdr.Engine(P, aux, devices=shards)
    subs = multi.shard_engines()
    G = len(subs)
    n_total = 50_003
    bounds = [n_total * g // G for g in range(G + 1)]
    d_pts, d_xyz, keep = [], [], []
    for g, e in enumerate(subs):  # every shard generates ITS block of one global stream on its own GPU
        n = bounds[g + 1] - bounds[g]
        bp, bx = e.alloc(8 * n * 16), e.alloc(n * 24)
        e.synth_device(0xC0FFEE + 9, 1, bounds[g], n, n, bp.ptr)
        e.synchronize()
        d_pts.append(bp.ptr)
        d_xyz.append(bx.ptr)
        keep += [bp, bx]
    root = subs[0].alloc(n_total * 24)
    n_sys = [bounds[g + 1] - bounds[g] for g in range(G)]
    fb = multi.multi_matrix_gather(d_pts, 8, n_sys, n_sys, d_xyz, root.ptr, transport=transport)
    assert fb == -1
    got = root.download((n_total, 3))
    # the unsharded answer: one context, the whole stream
    one = tdr.Engine(P, aux)
    bp, bx = one.alloc(8 * n_total * 16), one.alloc(n_total * 24)
    one.synth_device(0xC0FFEE + 9, 1, 0, n_total, n_total, bp.ptr)
    one.matrix_device(bp.ptr, 8, n_total, n_total, bx.ptr)
    one.synchronize()
    np.testing.assert_array_equal(got, bx.download((n_total, 3)))
    # RCCL refuses shards that share a device: reported, not crashed
    if G > 1:
        with pytest.raises(tdr.TdrError) as ei:
            multi.multi_matrix_gather(d_pts, 8, n_sys, n_sys, d_xyz, root.ptr, transport=1)
        assert ei.value.status == tdr.TDR_E_RCCL
    # the gather is pipelined behind the solve in pieces (TDR_GATHER_CHUNKS): a too-few system in a late piece
    # of the last shard must still come back as the GLOBALLY first failing index (pieces report chunk-relative
    # indices; the failure path re-solves unchunked)
    g_bad, local_bad = G - 1, n_sys[G - 1] - 7
    row0 = subs[g_bad].alloc(16)
    pts_last = keep[2 * g_bad].download((8, n_sys[g_bad], 2))
    for c in range(1, 8):  # blank cameras 1..7 of that system
        keep[2 * g_bad].engine  # (same engine object)
        tdr._check(tdr.lib().tdr_memcpy_h2d(subs[g_bad]._ctx, keep[2 * g_bad].ptr + (c * n_sys[g_bad] + local_bad) * 16,
                                            np.array([-1.0, -1.0]).ctypes.data, 16))
    with pytest.raises(tdr.TdrError, match="Too few rays are found"):
        multi.multi_matrix_gather(d_pts, 8, n_sys, n_sys, d_xyz, root.ptr, transport=transport)
    assert multi.first_bad == bounds[g_bad] + local_bad
    keep[2 * g_bad].upload(pts_last)  # restore; the context keeps working
    assert multi.multi_matrix_gather(d_pts, 8, n_sys, n_sys, d_xyz, root.ptr, transport=transport) == -1
    np.testing.assert_array_equal(root.download((n_total, 3)), got)


def test_concurrent_host_calls_on_one_context(tdr, oracle, scene8):
    """The host-buffer entry points serialise on the context's mutex (they share its staging workspace):
    concurrent callers get correct, independent results."""
    import threading
    sc, cams, P, aux = scene8
    eng = tdr.Engine(P, aux)
    jobs = []
    for k in range(6):
        X = sc.smooth_path(20_000 + 997 * k, seed=100 + k)
        pts = sc.make_detections(sc.project_P(P, X), seed=100 + k)
        jobs.append((pts, oracle.matrix_triangulate_points(P, pts, n_threads=4)[1]))
    out = [None] * len(jobs)

    def work(i):
        out[i] = eng.matrix_triangulate_points(jobs[i][0]) if i % 2 == 0 else eng.triangulate_points_rows(
            [jobs[i][0][c].copy() for c in range(8)])

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for (pts, ref), got in zip(jobs, out):
        assert got is not None and sc.rel_err(got, ref) <= TIGHT


def test_ray_batches_in_flight_on_two_streams(tdr, oracle):
    """Two LM batches enqueued back to back on one context without synchronising in between (each launch
    takes its own work-queue head from a ring, zeroed on the launch stream)."""
    sc, cams, P, aux = _scene(tdr, oracle, 8, seed=33)
    eng = tdr.Engine(P, aux)
    n = 20_000
    bufs = []
    for k in range(3):
        d_pts, d_xyz = eng.alloc(8 * n * 16), eng.alloc(n * 24)
        eng.synth_device(500 + k, 1, 0, n, n, d_pts.ptr, ray_model=True)
        bufs.append((d_pts, d_xyz))
    eng.synchronize()
    stat = [eng.alloc(n * 4) for _ in bufs]
    for (d_pts, d_xyz), d_st in zip(bufs, stat):  # three launches (+ their status passes) in flight
        eng.ray_device_ex(d_pts.ptr, 8, n, n, d_xyz.ptr, d_status=d_st.ptr)
    assert eng.fetch_first_bad() == -1
    for (d_pts, d_xyz), d_st in zip(bufs, stat):
        pts = d_pts.download((8, n, 2))
        idx = np.arange(0, n, 199)
        o = oracle.ray_triangulate_points_ex(aux, pts[:, idx], n_threads=8)
        ok = oracle.lm_comparable(d_st.download((n,), dtype=np.int32)[idx], o["status"])
        assert ok.mean() > 0.95
        assert sc.rel_err(d_xyz.download((n, 3))[idx][ok], o["xyz"][ok]) <= RAY_TIGHT


@pytest.mark.parametrize("n_sys", [21000, 21846, 21847, 30000])
def test_host_batch_routes_agree(tdr, oracle, scene8, n_sys):
    """Host-buffer batches up to 4 MiB (input + results) go through one pinned staging block, larger ones through
    per-row copies into the chunk workspace (tdr_capi.cpp run_host_batch).  Sizes on both sides of the
    switch (8 cameras, err requested: 160 B per system + 16 -> 26214 systems; without err 152 B -> 27594), the
    too-few report included, must give the same answers as the oracle."""
    sc, cams, P, aux = scene8
    X = sc.smooth_path(n_sys, seed=n_sys)
    pts = sc.make_detections(sc.project_P(P, X), seed=n_sys)
    eng = tdr.Engine(P, aux)
    for want_err in (False, True):
        for n in (n_sys, n_sys + 6000):  # 27000/27846/... : the second size is past the switch
            Xn = sc.smooth_path(n, seed=n)
            p = sc.make_detections(sc.project_P(P, Xn), seed=n)
            got = eng.matrix_triangulate_points(p, want_err=want_err)
            rc, ref, ref_err, _ = oracle.matrix_triangulate_points(P, p, want_err=True)
            xyz = got[0] if want_err else got
            assert rc == 0 and sc.rel_err(xyz, ref) <= TIGHT
            if want_err:
                np.testing.assert_allclose(got[1], ref_err, rtol=1e-9)
    bad = pts.copy()
    bad[1:, n_sys - 3] = -1
    for p in (bad, np.concatenate([bad, bad[:, :8000]], axis=1)):
        with pytest.raises(tdr.TdrError, match="Too few rays are found"):
            eng.matrix_triangulate_points(p)
        assert eng.first_bad == n_sys - 3
    assert sc.rel_err(eng.matrix_triangulate_points(pts), oracle.matrix_triangulate_points(P, pts)[1]) <= TIGHT


@pytest.mark.parametrize("n_cam,n_sys,noise,dropout", [(16, 3000, True, 0.08), (8, 5000, True, 0.08), (4, 4000, True, 0.08),
                                                       (6, 700, False, 0.0), (12, 900, True, 0.08), (13, 600, True, 0.08),
                                                       (8, 6000, True, 0.65), (16, 2500, True, 0.8), (3, 3000, True, 0.2)])
def test_ray_kernel_forms_agree_bitwise(tdr, oracle, n_cam, n_sys, noise, dropout):
    """The LM path has several execution forms — exec-masked or branch-free camera loop, lane-per-system with or
    without the cooperative drain, the camera-per-lane kernel for small batches — and which one a system runs
    through depends on the batch size, its neighbours in the wave and on scheduling.  They share every formula and the summation order, so point, error value and
    iteration count must agree bit for bit.  Noise-free detections exercise the exact
    central-difference branch (rays closer than 0.1 mm); the heavy drop-out cases are full of two- and three-ray
    systems, which round 3's lane-per-system kernels PARKED for a second launch (eigen solve, HISTORY.md §2.2) while the
    camera-per-lane kernel solves them in place — a system's result must not depend on which of the two it met."""
    sc, cams, P, aux = _scene(tdr, oracle, n_cam, seed=70 + n_cam)
    X = sc.smooth_path(n_sys, seed=n_sys)
    uv = sc.project_ray_model(aux, X)
    pts = sc.make_detections(uv, seed=n_sys, dropout_p=dropout, min_valid=2) if noise else uv.copy()
    eng = tdr.Engine(P, aux)
    if dropout >= 0.2:
        nv = ((pts[..., 0] != -1) & (pts[..., 1] != -1)).sum(axis=0)
        assert (nv == 2).sum() >= 100  # the parked population is there
    results = []
    # COOP_MAX = 0 keeps small batches out of the camera-per-lane kernel, so that the lane-per-system kernel
    # with and without its cooperative drain is compared too
    for coop, full, coop_max in ((1, 1, None), (1, 1, 0), (0, 1, 0), (1, 0, 0), (0, 0, 0)):
        tdr.set_knob("TDR_LM_COOP", coop)
        tdr.set_knob("TDR_LM_FULL", full)
        tdr.set_knob("TDR_LM_COOP_MAX", coop_max)
        try:
            g = eng.ray_triangulate_points_ex(pts)
            results.append((g["xyz"], g["err"], g["iters"], g["status"]))
        finally:
            for k in ("TDR_LM_COOP", "TDR_LM_FULL", "TDR_LM_COOP_MAX"):
                tdr.set_knob(k, None)
    g = eng.ray_triangulate_points_ex(pts)  # the default policy
    results.append((g["xyz"], g["err"], g["iters"], g["status"]))
    for r in results[1:]:
        for a, b in zip(r, results[0]):
            np.testing.assert_array_equal(a, b)
    assert np.isfinite(results[0][0]).all()
    if not noise:
        assert np.abs(results[0][0] - X).max() < 1e-3  # known answer


def _rays(aux, uv):
    """Vectorised pixel -> ray of the ray model (src/Triangulator.cpp:27-55) for unit quaternions:
    aux [n_cam, 10], uv [n_cam, n, 2] -> origins [n_cam, 3], unit directions [n_cam, n, 3]."""
    o, d = np.zeros((aux.shape[0], 3)), np.zeros(uv.shape[:2] + (3,))
    for c, a in enumerate(aux):
        qw, qx, qy, qz = a[:4]
        R = np.array([[1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qw * qz), 2 * (qx * qz + qw * qy)],
                      [2 * (qx * qy + qw * qz), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qw * qx)],
                      [2 * (qx * qz - qw * qy), 2 * (qy * qz + qw * qx), 1 - 2 * (qx * qx + qy * qy)]])
        fovy, w, h = a[7], a[8], a[9]
        pc = np.stack([(w / h) * (2 * (uv[c, :, 0] + 0.5) / w - 1), 2 * (uv[c, :, 1] + 0.5) / h - 1,
                       np.full(uv.shape[1], 1 / np.tan(fovy * 0.0174533 / 2))], axis=1)
        pc /= np.linalg.norm(pc, axis=1, keepdims=True)
        d[c] = pc @ R.T
        o[c] = a[4:7]
    return o, d


@pytest.mark.parametrize("n_cam,n_sys", [(16, 3000), (8, 4000), (12, 2500)])
def test_ray_forms_agree_when_single_rays_pass_through_the_solution(tdr, oracle, n_cam, n_sys):
    """Noisy systems in which ONE ray passes within a micrometre of the least-squares point — what happens to about
    one system in a thousand on real data, here to every fifth: the observation of one camera is moved onto the
    projection of the system's own minimiser until the two agree.  In such a system only that camera takes the
    reference's central-difference branch (src/RayTriangulator.cpp:28-44, rays closer than THRESHOLD), next to lanes
    that never do: the branch-free evaluator (the affected lanes patch that camera's term inside the two-camera block,
    lm_normal_full<NCAM, true>), the exec-masked one (lm_normal) and the camera-per-lane kernel must agree bit for
    bit, and with the oracle where both sides converged."""
    sc, cams, P, aux = _scene(tdr, oracle, n_cam, seed=170 + n_cam)
    X = sc.smooth_path(n_sys, seed=n_sys + 1)
    pts = sc.make_detections(sc.project_ray_model(aux, X), seed=n_sys + 1, dropout_p=0.0)
    rng = np.random.default_rng(n_sys)
    chosen = np.flatnonzero(rng.random(n_sys) < 0.2)
    cam_of = rng.integers(0, n_cam, chosen.size)
    for _ in range(8):  # the minimiser moves by ~1/n_cam of the change each time
        o, d = _rays(aux, pts[:, chosen])
        M = np.eye(3)[None, None] - d[..., :, None] * d[..., None, :]  # [cam, sys, 3, 3]
        x_star = np.linalg.solve(M.sum(axis=0), np.einsum("csij,cj->si", M, o)[..., None])[..., 0]
        uv = sc.project_ray_model(aux, x_star)  # [cam, sys, 2]
        pts[cam_of, chosen] = uv[cam_of, np.arange(chosen.size)]
    o, d = _rays(aux, pts[:, chosen])
    M = np.eye(3)[None, None] - d[..., :, None] * d[..., None, :]
    x_star = np.linalg.solve(M.sum(axis=0), np.einsum("csij,cj->si", M, o)[..., None])[..., 0]
    w = x_star[None] - o[:, None]
    dist = np.linalg.norm(np.cross(d, w), axis=2)  # [cam, sys]
    assert (dist[cam_of, np.arange(chosen.size)] < 1e-3).all()  # the chosen ray: well inside 0.1 mm
    assert (np.sort(dist, axis=0)[1] > 0.1).mean() > 0.99      # and (almost always) the only one
    eng = tdr.Engine(P, aux)
    results = []
    for coop, full, coop_max in ((1, 1, 0), (0, 1, 0), (1, 0, 0), (1, 1, None)):
        tdr.set_knob("TDR_LM_COOP", coop)
        tdr.set_knob("TDR_LM_FULL", full)
        tdr.set_knob("TDR_LM_COOP_MAX", coop_max)
        try:
            g = eng.ray_triangulate_points_ex(pts)
            results.append((g["xyz"], g["err"], g["iters"], g["status"]))
        finally:
            for k in ("TDR_LM_COOP", "TDR_LM_FULL", "TDR_LM_COOP_MAX"):
                tdr.set_knob(k, None)
    tdr.set_knob("TDR_LM_COOP_MAX", 1 << 30)  # the whole batch through the camera-per-lane kernel
    try:
        g = eng.ray_triangulate_points_ex(pts)
        results.append((g["xyz"], g["err"], g["iters"], g["status"]))
    finally:
        tdr.set_knob("TDR_LM_COOP_MAX", None)
    for r in results[1:]:
        for a, b in zip(r, results[0]):
            np.testing.assert_array_equal(a, b)
    o_ex = oracle.ray_triangulate_points_ex(aux, pts, n_threads=8)
    ok = oracle.lm_comparable(results[0][3], o_ex["status"])
    assert ok[chosen].mean() > 0.9
    assert sc.rel_err(results[0][0][ok], o_ex["xyz"][ok]) <= 1e-7
    both = ok[chosen]
    assert np.abs(results[0][0][chosen][both] - x_star[both]).max() < 1e-4  # mm: the closed-form minimiser


def test_dist_from_ray_batch(tdr, oracle, scene8):
    """Host-buffer getDistFromRay batch (tdr_dist_from_ray_batch, used once per frame by the classifier) against
    the oracle's scalar helper (src/Triangulator.cpp:57-61); argument checks."""
    sc, cams, P, aux = scene8
    rng = np.random.default_rng(11)
    n = 777
    idx = rng.integers(0, 8, n).astype(np.int32)
    uv = np.rint(rng.uniform([0, 0], [1924, 1082], (n, 2)))
    pt = rng.uniform(-2000, 2000, (n, 3))
    eng = tdr.Engine(P, aux)
    got = eng.dist_from_ray_batch(idx, uv, pt)
    ref = np.array([oracle.get_dist_from_ray(aux[idx[i]], uv[i, 0], uv[i, 1], pt[i]) for i in range(n)])
    np.testing.assert_allclose(got, ref, rtol=1e-12)
    # the host-side exports behind the adapter's protected statics (createRayForPoint + distToRay)
    # against the device kernel: the same tdr_math.h functions, FMA-contracted on the device only
    host = np.array([tdr.dist_to_ray(*tdr.create_ray(aux[idx[i]], uv[i]), pt[i]) for i in range(n)])
    np.testing.assert_allclose(got, host, rtol=1e-12)
    np.testing.assert_array_equal(host, ref)  # host code: the oracle's bits
    assert eng.dist_from_ray_batch(idx[:0], uv[:0], pt[:0]).shape == (0,)
    bad = idx.copy()
    bad[5] = 8
    with pytest.raises(tdr.TdrError) as ei:
        eng.dist_from_ray_batch(bad, uv, pt)
    assert ei.value.status == tdr.TDR_E_SHAPE


@pytest.mark.parametrize("n_cam,n_sys", [(2, 70), (4, 1000), (8, 5000), (5, 333), (16, 2049), (32, 300)])
def test_matrix_int16_format_is_bit_identical(tdr, oracle, n_cam, n_sys):
    """The narrow observation format (two int16 per observation, include/tdr_amd.h) converts in registers and
    then runs the double kernel's code: on integer-valued detections — what the reference's CSV data are — point
    and error must equal the double entry point's bit for bit, host-buffer and device-resident alike; the
    packer rejects values that are not int16 integers."""
    sc, cams, P, aux = _scene(tdr, oracle, n_cam, seed=90 + n_cam)
    X = sc.smooth_path(n_sys, seed=n_sys)
    pts = sc.make_detections(sc.project_P(P, X), seed=n_sys, dropout_p=0.1)  # rounded to integers, (-1,-1) missing
    eng = tdr.Engine(P, aux)
    xyz, err = eng.matrix_triangulate_points(pts, want_err=True)
    pts16 = pts.astype(np.int16)
    assert np.array_equal(pts16.astype(np.float64), pts)
    x16, e16 = eng.matrix_triangulate_points_i16(pts16, want_err=True)
    np.testing.assert_array_equal(x16, xyz)
    np.testing.assert_array_equal(e16, err)
    np.testing.assert_array_equal(eng.matrix_triangulate_points_i16(pts16), xyz)
    # device-resident: pack on the device, then the int16 kernel
    d_pts = eng.alloc(n_cam * n_sys * 16)
    d_p16 = eng.alloc(n_cam * (n_sys + 3) * 4)
    d_xyz = eng.alloc(n_sys * 24)
    d_pts.upload(pts)
    # row stride n_sys (odd n_sys: one system per lane), an even padded stride (two systems per lane) and a stride
    # padded to a multiple of four, each with and without the two-systems-per-lane kernel
    try:
        for per_lane in (None, 0, 1):
            tdr.set_knob("TDR_I16_X2", per_lane)
            for stride in (n_sys, n_sys + (n_sys & 1), (n_sys + 3) // 4 * 4):
                eng.pack_i16_device(d_pts.ptr, n_cam, n_sys, n_sys, d_p16.ptr, stride)
                d_xyz.upload(np.zeros((n_sys, 3)))
                eng.matrix_i16_device(d_p16.ptr, n_cam, stride, n_sys, d_xyz.ptr)
                assert eng.fetch_first_bad() == -1
                np.testing.assert_array_equal(d_xyz.download((n_sys, 3)), xyz)
    finally:
        tdr.set_knob("TDR_I16_X2", None)
    # a non-integer or out-of-range coordinate is refused by the packer
    for bad_value in (100.5, 40000.0, np.nan):
        bad = pts.copy()
        bad[0, n_sys // 2, 1] = bad_value
        d_pts.upload(bad)
        with pytest.raises(tdr.TdrError) as ei:
            eng.pack_i16_device(d_pts.ptr, n_cam, n_sys, n_sys, d_p16.ptr, n_sys)
        assert ei.value.status == tdr.TDR_E_ARG
    # too-few semantics are the double path's
    few = pts16.copy()
    few[1:, n_sys - 2] = -1
    with pytest.raises(tdr.TdrError, match="Too few rays are found"):
        eng.matrix_triangulate_points_i16(few)
    assert eng.first_bad == n_sys - 2
    for b in (d_pts, d_p16, d_xyz):
        b.free()


@pytest.mark.parametrize("mode,chunk", [(1, 4096), (2, 4096), (2, 777), (1, 30000), (2, 39999), (3, 4096), (3, 777),
                                        (3, 39999)])
def test_host_pipeline_modes_are_bit_identical(tdr, oracle, scene8, mode, chunk):
    """The pipelined host-buffer routes (run_host_pipelined: 4 device slots, upload / kernel / download on three
    streams, pinned staging workers in mode 2) against the serial route on batches cut into many chunks: matrix
    with err, ray with iters + status, the row-pointer entry point, a multi-shard context, and a too-few system
    in a late chunk (the globally first index must come back and the call must not hang or leak threads)."""
    sc, cams, P, aux = scene8
    n = 40_000  # 8 x 40 000 x 16 B = 5.1 MB: past the small-batch pinned block
    X = sc.smooth_path(n, seed=91)
    pts = sc.make_detections(sc.project_P(P, X), seed=91)
    ptsr = sc.make_detections(sc.project_ray_model(aux, X), seed=92, dropout_p=0.3, min_valid=2)
    ref_eng = tdr.Engine(P, aux)
    ref_eng.set_host_pipeline(0, 0)
    eng = tdr.Engine(P, aux)
    eng.set_host_pipeline(mode, chunk)
    xyz0, err0 = ref_eng.matrix_triangulate_points(pts, want_err=True)
    xyz1, err1 = eng.matrix_triangulate_points(pts, want_err=True)
    np.testing.assert_array_equal(xyz0, xyz1)
    np.testing.assert_array_equal(err0, err1)
    np.testing.assert_array_equal(eng.matrix_triangulate_points(pts), xyz0)
    np.testing.assert_array_equal(eng.triangulate_points_rows([pts[c].copy() for c in range(8)]), xyz0)
    r0 = ref_eng.ray_triangulate_points_ex(ptsr)
    r1 = eng.ray_triangulate_points_ex(ptsr)
    for k in ("xyz", "err", "iters", "status"):
        np.testing.assert_array_equal(r0[k], r1[k])
    assert (r1["status"] != 0).any() and (r1["status"] == 0).mean() > 0.8
    np.testing.assert_array_equal(eng.ray_triangulate_points(ptsr), r0["xyz"])
    multi = tdr.Engine(P, aux, devices=[0, 0, 0])
    multi.set_host_pipeline(mode, max(64, chunk // 4))
    np.testing.assert_array_equal(multi.matrix_triangulate_points(pts), xyz0)
    bad = pts.copy()
    bad[1:, 33_333] = -1
    bad[1:, 39_000] = -1
    for e in (eng, multi):
        with pytest.raises(tdr.TdrError, match="Too few rays are found"):
            e.matrix_triangulate_points(bad)
        assert e.first_bad == 33_333
        np.testing.assert_array_equal(e.matrix_triangulate_points(pts), xyz0)  # and the context still works
    badr = ptsr.copy()
    badr[1:, 12_345] = -1
    with pytest.raises(tdr.TdrError, match="Too few detections are found"):
        eng.ray_triangulate_points_ex(badr)
    assert eng.first_bad == 12_345
    # mode 3 sends integer-valued chunks as int16 pairs; chunks holding a fraction, a value beyond int16 or a NaN
    # must travel as doubles — and every result must still equal the serial double route bit for bit
    mixed = pts.copy()
    mixed[2, 5_000] = [700.25, 300.5]       # fractional pixel
    mixed[3, 17_000] = [40000.0, 200.0]     # beyond int16
    mixed[5, 29_500, 0] += 1e-9             # almost an integer
    mixed[0, 39_998] = [-32768.0, 32767.0]  # the int16 limits themselves still convert
    np.testing.assert_array_equal(eng.matrix_triangulate_points(mixed), ref_eng.matrix_triangulate_points(mixed))
    x_un = sc.project_P(P, X)  # exact, un-rounded projections: every chunk travels as doubles
    np.testing.assert_array_equal(eng.matrix_triangulate_points(x_un), ref_eng.matrix_triangulate_points(x_un))
    # the LM path takes the same transport (int16 rows unpacked on the device in front of the kernel)
    mixedr = ptsr.copy()
    mixedr[2, 5_000] = [700.25, 300.5]
    mixedr[3, 17_000] = [40000.0, 200.0]
    mixedr[0, 39_998] = [-32768.0, 32767.0]
    rm0, rm1 = ref_eng.ray_triangulate_points_ex(mixedr), eng.ray_triangulate_points_ex(mixedr)
    for k in ("xyz", "err", "iters", "status"):
        np.testing.assert_array_equal(rm0[k], rm1[k])


def test_lm_status_entry_points_argument_rules(tdr, scene8):
    """tdr_ray_triangulate_points_ex / _device_ex: status and iteration outputs are optional and independent; the
    matrix path has no such outputs; misaligned int32 pointers are refused; a too-few system is TDR_LM_TOO_FEW in
    the device-resident form (where the batch does not fail by itself)."""
    import ctypes as C
    sc, cams, P, aux = scene8
    eng = tdr.Engine(P, aux)
    n = 3000
    X = sc.smooth_path(n, seed=5)
    pts = sc.make_detections(sc.project_ray_model(aux, X), seed=5)
    pts[1:, 77] = -1  # one system with a single observation
    d_pts, d_xyz = eng.alloc(pts.nbytes).upload(pts), eng.alloc(n * 24)
    d_it, d_st = eng.alloc(n * 4 + 8), eng.alloc(n * 4 + 8)
    eng.ray_device_ex(d_pts.ptr, 8, n, n, d_xyz.ptr, d_iters=d_it.ptr, d_status=d_st.ptr)
    assert eng.fetch_first_bad() == 77
    st = d_st.download((n,), dtype=np.int32)
    it = d_it.download((n,), dtype=np.int32)
    assert st[77] == tdr.TDR_LM_TOO_FEW and it[77] == 0 and np.isnan(d_xyz.download((n, 3))[77]).all()
    assert ((st & tdr.TDR_LM_TOO_FEW) != 0).sum() == 1
    eng.ray_device_ex(d_pts.ptr, 8, n, n, d_xyz.ptr, d_status=d_st.ptr)  # status without iters
    assert eng.fetch_first_bad() == 77
    np.testing.assert_array_equal(d_st.download((n,), dtype=np.int32), st)
    eng.ray_device_ex(d_pts.ptr, 8, n, n, d_xyz.ptr, d_iters=d_it.ptr)  # iters without status
    assert eng.fetch_first_bad() == 77
    np.testing.assert_array_equal(d_it.download((n,), dtype=np.int32), it)
    L = tdr.lib()
    assert L.tdr_ray_triangulate_points_device_ex(eng._ctx, d_pts.ptr, 8, n, n, d_xyz.ptr, None, d_it.ptr + 2, None,
                                                  None) == tdr.TDR_E_ARG
    eng_m = tdr.Engine(P, None)
    assert L.tdr_ray_triangulate_points_device_ex(eng_m._ctx, d_pts.ptr, 8, n, n, d_xyz.ptr, None, None, None,
                                                  None) == tdr.TDR_E_ARG  # no ray constants in this context


def test_single_system_entry_points_give_the_batch_path_bits(tdr, oracle):
    """tdr_matrix_triangulate_point / tdr_ray_triangulate_point (one launch per call; round 5: constants in the kernel
    argument, result through a host-coherent pinned slot, the ray call on the camera-per-lane kernel) against the batch
    entry points on the same observations: the same device arithmetic in the same order, so the same bits — under the
    default route and under TDR_POINT_ARGS=0 (round 4's copy-up / launch / copy-down route, kept as the A/B)."""
    sc, cams, P, aux = _scene(tdr, oracle, 8, seed=61)
    X = sc.smooth_path(40, seed=6)
    uv_m = sc.make_detections(sc.project_P(P, X), seed=6, dropout_p=0.0)
    uv_r = sc.make_detections(sc.project_ray_model(aux, X), seed=7, dropout_p=0.0)
    eng = tdr.Engine(P, aux)
    big = tdr.Engine(np.tile(P, (3, 1)), np.tile(aux, (3, 1)))  # 24 "cameras": the > 16-observation fallback
    try:
        for mode in (None, 0):
            tdr.set_knob("TDR_POINT_ARGS", mode)
            for subset in ([0, 1], [5, 2, 7], [0, 1, 2, 3], list(range(8)), [7, 6, 5, 4, 3, 2]):
                e_sub = tdr.Engine(P[subset], aux[subset])
                bm, bm_err = e_sub.matrix_triangulate_points(uv_m[subset], want_err=True)
                br, br_err = e_sub.ray_triangulate_points(uv_r[subset], want_err=True)
                for s in (0, 17, 39):
                    xm, em = eng.matrix_triangulate_point(P[subset], uv_m[subset, s])
                    assert np.array_equal(xm, bm[s]) and em == bm_err[s]
                    xr, er = eng.ray_triangulate_point(aux[subset], uv_r[subset, s])
                    assert np.array_equal(xr, br[s]) and er == br_err[s], (mode, subset, s)
            # 24 observations (three copies of every camera): the generic single-system LM kernel
            xr, er = big.ray_triangulate_point(np.tile(aux, (3, 1)), np.tile(uv_r[:, 3], (3, 1)))
            ref = big.ray_triangulate_points(np.tile(uv_r[:, 3:4], (3, 1, 1)))
            assert np.abs(xr - ref[0]).max() <= 1e-7 * np.abs(ref[0]).max()
            d = eng.dist_from_ray_batch(np.array([2], dtype=np.int32), uv_r[2, 5][None], X[9][None])[0]
            rt = tdr.RayTriangulator(cams)
            assert rt.getDistFromRay(tdr.CamPointPair(cams[2], uv_r[2, 5]), X[9]) == d
    finally:
        tdr.set_knob("TDR_POINT_ARGS", None)

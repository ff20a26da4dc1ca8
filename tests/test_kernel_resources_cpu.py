"""Occupancy cliffs of the hand-written kernels, checked in the container.  The LM kernels are
register-allocated to the edge — the 16-camera throughput kernel needs 246 of the 256 VGPRs that still allow two waves
per SIMD, and one wave fewer costs 40 % of C5 (measured in round 3: 3.1 -> 4.1 ms) — so a source or compiler change
that costs a kernel a wave must fail HERE, not on the bench.  The numbers are the compiler's own
(-Rpass-analysis=kernel-resource-usage), written next to the library by the very build that produced it
(`make libtdr_amd.so` -> build/resource_usage.txt); if that file is missing or older than the sources, the library is
rebuilt (about a minute and a half)."""
import importlib.util
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT, "3d-reconstruction-triangulation_amd")
RES = os.path.join(PKG_DIR, "build", "resource_usage.txt")
SOURCES = [os.path.join(PKG_DIR, "csrc", n) for n in ("tdr_kernels.hip", "tdr_lm_kernels.hip", "tdr_lm_literal.hip",
                                                      "tdr_lm_device.h", "tdr_device_common.h", "tdr_kernels.h",
                                                      "tdr_math.h")]


@pytest.fixture(scope="module")
def usage():
    stale = not os.path.exists(RES) or any(os.path.getmtime(s) > os.path.getmtime(RES) for s in SOURCES)
    if stale:
        if not os.path.exists("/opt/rocm/bin/hipcc"):
            pytest.skip("no hipcc here and no resource_usage.txt from the build")
        subprocess.run(["make", "-C", PKG_DIR, "libtdr_amd.so"], check=True, capture_output=True, timeout=1200)
    spec = importlib.util.spec_from_file_location("resource_usage", os.path.join(ROOT, "tools", "resource_usage.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    u = mod.parse(RES)
    assert len(u) > 200, len(u)  # every instantiation reported
    return u


def _get(usage, name):
    assert name in usage, (name, [k for k in usage if k.startswith(name.split("<")[0])][:8])
    return usage[name]


# kernel -> (minimum waves per SIMD, scratch bytes allowed)
LM_EXPECT = {
    # throughput kernels: FULL (branch-free evaluator, the rig's own camera count) and masked
    "ray_kernel<2, CamR<2>, true, false, 0, false, 0>": (3, 0),
    "ray_kernel<3, CamR<3>, true, false, 0, false, 0>": (3, 0),
    "ray_kernel<4, CamR<4>, true, false, 0, false, 0>": (3, 0),
    "ray_kernel<4, CamR<4>, true, true, 0, false, 0>": (3, 0),
    "ray_kernel<6, CamR<6>, true, true, 0, false, 0>": (3, 0),
    "ray_kernel<6, CamR<6>, true, false, 0, false, 0>": (3, 0),
    "ray_kernel<8, CamR<8>, true, true, 0, false, 0>": (2, 0),
    "ray_kernel<8, CamR<8>, true, false, 0, false, 0>": (2, 0),
    "ray_kernel<12, CamR<12>, true, true, 0, false, 0>": (2, 0),
    "ray_kernel<12, CamR<12>, true, false, 0, false, 0>": (2, 0),
    "ray_kernel<16, CamR<16>, true, true, 0, false, 0>": (2, 0),   # C5
    "ray_kernel<16, CamR<16>, true, false, 0, false, 0>": (2, 0),
    # 24 / 32 cameras keep the last directions in LDS to hold two waves; a few spilled dwords are tolerated there
    "ray_kernel<24, CamR<24>, true, true, 4, false, 0>": (2, 64),
    "ray_kernel<24, CamR<24>, true, false, 4, false, 0>": (2, 64),
    "ray_kernel<32, CamR<32>, true, true, 12, false, 0>": (2, 64),
    "ray_kernel<32, CamR<32>, true, false, 12, false, 0>": (2, 64),
    # the literal build (knob TDR_LM_LITERAL_FD, a diagnostic: six shifted distance evaluations per ray and the Jacobi
    # eigen form inlined): two waves up to 8 cameras, one at 16 — and never scratch
    "ray_kernel<4, CamR<4>, true, false, 0, true, 0>": (2, 0),
    "ray_kernel<8, CamR<8>, true, true, 0, true, 0>": (2, 0),
    "ray_kernel<16, CamR<16>, true, true, 0, true, 0>": (1, 0),
}


@pytest.mark.parametrize("name", sorted(LM_EXPECT))
def test_lm_kernel_keeps_its_waves(usage, name):
    occ, scratch = LM_EXPECT[name]
    r = _get(usage, name)
    assert r["occupancy"] >= occ, (name, r)
    assert r["scratch"] <= scratch, (name, r)
    if occ >= 2:
        assert (r["agprs"] or 0) == 0, (name, r)  # accumulation registers in use = past the 256-VGPR budget


@pytest.mark.parametrize("ncam", [2, 3, 4, 6, 8, 12, 16])
def test_camera_per_lane_kernel_keeps_three_waves(usage, ncam):
    # since round 4 (no Jacobi sweeps inlined) the small-batch kernel fits three waves per SIMD
    r = _get(usage, f"ray_coop_kernel<{ncam}, false>")
    assert r["occupancy"] >= 3 and r["scratch"] == 0, r


def test_matrix_kernels_have_no_scratch_and_c3_keeps_six_waves(usage):
    c3 = _get(usage, "matrix_kernel<8, false, 0, true, true, 64, false>")  # the BENCH kernel
    assert c3["occupancy"] >= 6 and c3["scratch"] == 0 and c3["vgprs"] <= 80, c3
    n = 0
    for name, r in usage.items():
        if name.startswith(("matrix_kernel<", "matrix_i16xk_kernel<", "score_leaves_kernel<", "lm_status_kernel",
                            "dist_from_ray_kernel", "synth_kernel")):
            n += 1
            if name == "score_leaves_kernel<32>":  # 17-32 cameras: the digit / row arrays live in scratch (cold config)
                continue
            assert r["scratch"] == 0, (name, r)
            assert r["occupancy"] >= 2, (name, r)
    assert n > 100, n
    i16 = _get(usage, "matrix_i16xk_kernel<8, false, true, 256, 2, true>")  # the int16 format row of the bench
    assert i16["occupancy"] >= 4, i16

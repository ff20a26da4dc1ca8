"""The torch.distributed side of the multi-GPU path on RCCL itself: `sharding.gather_path`,
`sharding.solve_and_gather_chunked` and `sharding.global_first_bad` with the **nccl** backend (= RCCL on ROCm) at
world size 1 on the one GPU of the box — device tensors, async `dist.gather` into row-block views of the result,
the solves on a side stream with an event per piece — compared bit for bit with the unsharded launch.  The gloo
tests (tests/test_sharding_gloo.py, tests/test_gpu_bench_rehearsal.py) cover world sizes 2 and 4; what they cannot
cover is that RCCL accepts these calls in this form.  Two ranks on one device are refused by RCCL, so world size 1
is what a one-GPU box can do; real 2/4/8-GPU numbers remain the driver's to measure.

The ranks are fresh child processes of pytest (this file run as a script), so a failed communicator cannot leak
into the other GPU tests."""
import importlib
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = "3d-reconstruction-triangulation_amd"

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _child():
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    tdr = importlib.import_module(PKG)
    sc = importlib.import_module(PKG + ".scenes")
    sh = importlib.import_module(PKG + ".sharding")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    # TDR_TEST_DATA_GROUP=1: the layout bench.py uses — the DEFAULT group is gloo (control plane: barriers, timing
    # reductions on host tensors) and RCCL is a group of its own that carries the gather (bench.make_data_group)
    split = os.environ.get("TDR_TEST_DATA_GROUP") == "1"
    grp = None
    if split:
        import bench
        dist.init_process_group("gloo", rank=0, world_size=1)
        grp = bench.make_data_group(torch, dist, 0)
    else:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    out = {"backend": dist.get_backend(), "data_backend": dist.get_backend(grp)}
    try:
        n_cam, n_sys = 8, 300_000
        cp = sc.ring_camera_params(n_cam, seed=4)
        cams = [tdr.createCamera(i, c["width"], c["height"], c["focal"], c["tvec"], c["rquat"]) for i, c in enumerate(cp)]
        P, aux = tdr.pack_cameras(cams)
        eng = tdr.Engine(P, aux, device=0)
        pts = torch.empty((n_cam, n_sys, 2), dtype=torch.float64, device=dev)
        eng.synth_device(0xC0FFEE + 4, 4, 0, n_sys, n_sys, pts.data_ptr())
        ref = torch.empty((n_sys, 3), dtype=torch.float64, device=dev)
        eng.matrix_device(pts.data_ptr(), n_cam, n_sys, n_sys, ref.data_ptr())
        eng.synchronize()
        assert eng.fetch_first_bad() == -1
        assert bool(torch.isfinite(ref).all())

        # the single gather of the path (device tensors through RCCL)
        full = sh.gather_path(ref, n_sys, dst=0, group=grp)
        torch.cuda.synchronize()
        out["gather_path_rows"] = int(full.shape[0])
        out["gather_path_differs"] = int((full != ref).any(dim=1).sum())
        out["gather_path_is_cuda"] = bool(full.is_cuda)

        # the gather pipelined behind the solve: pieces solved on a side stream, gathered asynchronously in place
        out["chunked"] = {}
        for n_chunks in (1, 3, 4, 7):
            xyz = torch.full((n_sys, 3), float("nan"), dtype=torch.float64, device=dev)
            launched = []

            def launch_chunk(c0, c1, stream):
                assert stream is not None  # nccl: a side stream is used
                launched.append((c0, c1))
                eng.matrix_device(pts.data_ptr() + c0 * 16, n_cam, n_sys, c1 - c0, xyz.data_ptr() + c0 * 24,
                                  stream=stream)

            got = sh.solve_and_gather_chunked(launch_chunk, xyz, n_sys, n_chunks=n_chunks, dst=0, group=grp)
            torch.cuda.synchronize()
            out["chunked"][str(n_chunks)] = {
                "pieces": launched == sh.chunk_ranges(n_sys, n_chunks), "rows": int(got.shape[0]),
                "differs": int((got != ref).any(dim=1).sum()), "is_cuda": bool(got.is_cuda)}

        # the MIN all-reduce of the first failing index on a device tensor
        out["first_bad_none"] = sh.global_first_bad(-1, 1000, device=dev, group=grp)
        out["first_bad_some"] = sh.global_first_bad(17, 1000, device=dev, group=grp)
        # a barrier and the MAX reduction bench.py times with (on the control plane: host tensors when it is gloo)
        dist.barrier()
        t = torch.tensor([1.5, 2.5], dtype=torch.float64, device="cpu" if split else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        out["allreduce_max"] = [float(v) for v in t]
    finally:
        dist.destroy_process_group()
    print("NCCL_WORLD1 " + json.dumps(out), flush=True)


@pytest.mark.parametrize("split", [False, True], ids=["default_group_nccl", "gloo_control_plane_plus_nccl_data_group"])
def test_sharding_on_rccl_world_size_1(split):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0", TDR_TEST_DATA_GROUP="1" if split else "0")
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = [l for l in r.stdout.splitlines() if l.startswith("NCCL_WORLD1 ")]
    assert len(line) == 1, r.stdout[-2000:]
    d = json.loads(line[0][len("NCCL_WORLD1 "):])
    assert d["backend"] == ("gloo" if split else "nccl") and d["data_backend"] == "nccl"
    assert d["gather_path_rows"] == 300_000 and d["gather_path_differs"] == 0 and d["gather_path_is_cuda"]
    for k, rec in d["chunked"].items():
        assert rec == {"pieces": True, "rows": 300_000, "differs": 0, "is_cuda": True}, (k, rec)
    assert d["first_bad_none"] == -1 and d["first_bad_some"] == 1017
    assert d["allreduce_max"] == [1.5, 2.5]


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` with no launcher and no RANK / WORLD_SIZE in the environment starts its own ranks
    as fresh child processes (bench.self_launch), relays rank 0's single JSON line and the children's exit code."""
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "LOCAL_WORLD_SIZE", "GROUP_RANK")}
    env.update(TDR_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 2 and d["scaling"] == "weak"
    assert "C3" in d["config"]["workload"] and "REHEARSAL" in d["config"]["workload"]
    c4 = d["c4_strong"]
    assert "error" not in c4, c4
    assert c4["scaling"] == "strong" and c4["gathered_rows_differing_from_a_local_solve"] == 0
    assert 0 < c4["value_with_gather_pipelined"] <= c4["scaling_kernel_only_value"]
    # a failing child's exit code comes back through the launcher
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--config", "c2", "--variant", "987654"], env=env, cwd=ROOT, capture_output=True, text=True,
                         timeout=600)
    assert bad.returncode != 0
    assert not [l for l in bad.stdout.splitlines() if l.startswith("{")]


if __name__ == "__main__" and "--child" in sys.argv:
    _child()

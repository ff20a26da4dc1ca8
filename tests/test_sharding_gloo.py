"""N>1 path on CPU: world_size-2 gloo processes run the sharding / gather / first-bad logic that
bench.py and multi-GPU callers use with RCCL.  The per-shard solve is stood in by the CPU oracle
(this is a test of the host logic around the kernels, not of the kernels)."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_sys, bad_at, out_dir):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import tdr_oracle as oracle
        sh = importlib.import_module("3d-reconstruction-triangulation_amd.sharding")
        sc = importlib.import_module("3d-reconstruction-triangulation_amd.scenes")
        cp = sc.ring_camera_params(5, seed=12)
        cams = [oracle.camera_derive(c["width"], c["height"], c["focal"], c["tvec"], c["rquat"]) for c in cp]
        P, aux = oracle.cameras_to_arrays(cams)
        X = sc.smooth_path(n_sys, seed=12)
        pts = sc.make_detections(sc.project_P(P, X), seed=12)  # identical on every rank
        for b in bad_at:
            pts[1:, b] = -1
        s0, s1 = sh.shard_range(n_sys, world, rank)
        rc, xyz_local, _, fb = oracle.matrix_triangulate_points(P, pts[:, s0:s1])
        gfb = sh.global_first_bad(fb, s0)
        full = sh.gather_path(torch.from_numpy(xyz_local), n_sys, dst=0)
        if rank == 0:
            rc_ref, ref, _, fb_ref = oracle.matrix_triangulate_points(P, pts)
            np.save(os.path.join(out_dir, "result.npy"),
                    np.array([gfb, fb_ref, float(full.shape[0]),
                              0.0 if bad_at else float(np.abs(full.numpy() - ref).max())]))
        else:
            assert full is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_sys,bad_at", [(1001, []), (640, []), (777, [700, 120, 650]), (10, [9])])
def test_shard_gather_first_bad_world2(tmp_path, n_sys, bad_at):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n_sys, bad_at, str(tmp_path)), nprocs=world, join=True)
    gfb, fb_ref, n_rows, max_abs = np.load(tmp_path / "result.npy")
    assert n_rows == n_sys
    assert gfb == fb_ref == (min(bad_at) if bad_at else -1)
    assert max_abs == 0.0  # sharded + gathered == unsharded, bit for bit


def test_shard_ranges_cover_exactly():
    sh = importlib.import_module("3d-reconstruction-triangulation_amd.sharding")
    for n in (0, 1, 7, 8, 1001, 10_000_000):
        for w in (1, 2, 3, 4, 8):
            r = [sh.shard_range(n, w, g) for g in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            assert sum(sh.shard_counts(n, w)) == n
            assert max(sh.shard_counts(n, w)) - min(sh.shard_counts(n, w)) <= 1


def _worker_chunked(rank, world, port, n_sys, n_chunks, out_dir):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import tdr_oracle as oracle
        sh = importlib.import_module("3d-reconstruction-triangulation_amd.sharding")
        sc = importlib.import_module("3d-reconstruction-triangulation_amd.scenes")
        cp = sc.ring_camera_params(4, seed=21)
        cams = [oracle.camera_derive(c["width"], c["height"], c["focal"], c["tvec"], c["rquat"]) for c in cp]
        P, aux = oracle.cameras_to_arrays(cams)
        X = sc.smooth_path(n_sys, seed=21)
        pts = sc.make_detections(sc.project_P(P, X), seed=21)
        s0, s1 = sh.shard_range(n_sys, world, rank)
        xyz_local = torch.full((s1 - s0, 3), float("nan"), dtype=torch.float64)
        launched = []

        def launch_chunk(c0, c1, stream):  # the per-piece solve: the CPU oracle stands in for the kernel
            assert stream is None
            rc, xyz, _, _ = oracle.matrix_triangulate_points(P, pts[:, s0 + c0:s0 + c1])
            assert rc == 0
            xyz_local[c0:c1] = torch.from_numpy(xyz)
            launched.append((c0, c1))

        full = sh.solve_and_gather_chunked(launch_chunk, xyz_local, n_sys, n_chunks=n_chunks, dst=0)
        assert launched == sh.chunk_ranges(s1 - s0, n_chunks)
        # the same step into a receive buffer allocated beforehand (bench.py makes that allocation — the one
        # rank-asymmetric one — a phase of its own): the root gets a buffer, the others None, same result in place
        recv = sh.gather_buffer(xyz_local, n_sys, dst=0)
        assert (recv is not None) == (rank == 0)
        del launched[:]
        again = sh.solve_and_gather_chunked(launch_chunk, xyz_local, n_sys, n_chunks=n_chunks, dst=0, out=recv)
        if rank == 0:
            assert again.data_ptr() == recv.data_ptr() and torch.equal(again, full)
        if rank == 0:
            rc, ref, _, _ = oracle.matrix_triangulate_points(P, pts)
            np.save(os.path.join(out_dir, "chunked.npy"), np.array([float(full.shape[0]), float(np.abs(full.numpy() - ref).max())]))
        else:
            assert full is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_sys,n_chunks", [(1000, 4), (64, 1), (10, 7), (2002, 3)])
def test_chunked_overlapped_gather_world2(tmp_path, n_sys, n_chunks):
    """solve_and_gather_chunked (the gather pipelined behind the solve) == one unsharded solve."""
    world = 2
    mp.spawn(_worker_chunked, args=(world, _free_port(), n_sys, n_chunks, str(tmp_path)), nprocs=world, join=True)
    n_rows, max_abs = np.load(tmp_path / "chunked.npy")
    assert n_rows == n_sys and max_abs == 0.0


def test_chunk_ranges():
    sh = importlib.import_module("3d-reconstruction-triangulation_amd.sharding")
    for n in (0, 1, 5, 1000, 1234567):
        for k in (1, 3, 4, 16):
            r = sh.chunk_ranges(n, k)
            if n == 0:
                assert r == []
                continue
            assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert all(b > a for a, b in r) and len(r) <= k


def _worker_together(rank, world, port, out_dir):
    for p in (ROOT,):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        cpu = torch.device("cpu")
        # a phase that succeeds everywhere returns its value on every rank
        v = bench.phase_together(torch, dist, cpu, "record", "fine", lambda: rank * 10)
        assert v == rank * 10

        def fails_on_rank_1():
            if rank == 1:
                raise MemoryError("no room for the shard")  # what an out-of-memory allocation looks like
            return "ok"

        msg = None
        try:
            bench.phase_together(torch, dist, cpu, "record", "setup", fails_on_rank_1)
        except RuntimeError as exc:
            msg = str(exc)
        # BOTH ranks got out — the healthy one too — and the collectives still work afterwards
        t = torch.tensor([1.0 + rank])
        dist.all_reduce(t)
        with open(os.path.join(out_dir, f"together_{rank}.txt"), "w") as f:
            f.write(f"{msg}\n{float(t[0])}\n")
    finally:
        dist.destroy_process_group()


def test_a_rank_local_failure_takes_every_rank_out_of_the_record_together(tmp_path):
    """a side record of bench.py must never leave healthy ranks blocked inside a collective because
    another rank failed before it.  bench.phase_together: the failing rank raises its own error text, the healthy one
    raises too ("another rank failed"), and the process group is still usable."""
    world = 2
    mp.spawn(_worker_together, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = open(tmp_path / "together_0.txt").read().splitlines()
    r1 = open(tmp_path / "together_1.txt").read().splitlines()
    assert "abandoned by all ranks in phase 'setup'" in r0[0] and "another rank failed" in r0[0]
    assert "abandoned by all ranks in phase 'setup'" in r1[0] and "MemoryError: no room for the shard" in r1[0]
    assert r0[1] == r1[1] == "3.0"


def test_bench_without_a_launcher_starts_its_own_ranks_and_relays_their_exit_code():
    """`python bench.py --gpus 2` with no RANK / WORLD_SIZE: the parent must become the launcher BEFORE it imports torch
    or touches a GPU (bench.self_launch).  Here there is no GPU, so the two fresh ranks refuse to run ("no CPU fallback
    exists") — and that refusal, not a traceback of the parent, is what comes back, with a non-zero exit code."""
    import subprocess
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "LOCAL_WORLD_SIZE", "GROUP_RANK")}
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by tests/test_gpu_sharding_nccl.py")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "bench.py needs an MI355X" in r.stderr, r.stderr[-2000:]
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


_WATCHDOG_RANK = r"""
import json, os, sys, time
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
import bench
rank = int(os.environ["RANK"])
dist.init_process_group("gloo", rank=rank, world_size=2)
line = {{"metric": "m", "value": 42.0, "roofline": {{"frac": 0.8}}}}
wd = bench.LineWatchdog(float(os.environ["WD_S"]), rank, (lambda: line) if rank == 0 else None)
wd.start()
if os.environ["WD_CASE"] == "hang":
    line["gather"] = {{"ms": 1.0}}          # an extra that completed before the hang
    if rank == 1:
        time.sleep(120)                     # never reaches the collective
    t = torch.ones(1)
    dist.all_reduce(t)                      # rank 0 blocks here for good
    print("NOT REACHED", flush=True)
else:
    t = torch.ones(1)
    dist.all_reduce(t)
    wd.finish()
    if rank == 0:
        print(json.dumps(line), flush=True)
    time.sleep(float(os.environ["WD_S"]) + 1.0)   # the disarmed deadline passes: nothing more may be printed
    dist.destroy_process_group()
"""


def _run_watchdog_case(case, seconds):
    import subprocess
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   WD_S=str(seconds), WD_CASE=case)
        procs.append(subprocess.Popen([sys.executable, "-c", _WATCHDOG_RANK.format(root=ROOT)], env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            o, e = p.communicate(timeout=90)
            outs.append((p.returncode, o, e))
    finally:
        for p in procs:  # exactly the processes this test started
            if p.poll() is None:
                p.kill()
    return outs


def test_a_deadline_saves_the_contract_line_when_a_rank_never_returns_from_a_collective():
    """bench.LineWatchdog (N > 1): everything after the timed region of bench.py sits on collectives that have never run
    on more than one real GPU in this pipeline.  Rank 1 never joins an all-reduce; rank 0 is blocked inside it.  At the
    deadline rank 0 prints ONE JSON line — the contract fields, the extras that were complete, `side_records_error` —
    and both ranks leave with exit code 0, nobody waits for a communicator time-out."""
    import json
    (rc0, out0, err0), (rc1, out1, err1) = _run_watchdog_case("hang", 3.0)
    assert rc0 == 0 and rc1 == 0, (err0[-800:], err1[-800:])
    lines = [l for l in out0.splitlines() if l.strip() and not l.startswith("[Gloo]")]  # gloo announces itself on stdout
    assert len(lines) == 1 and "NOT REACHED" not in out0
    line = json.loads(lines[0])
    assert line["value"] == 42.0 and line["roofline"]["frac"] == 0.8 and line["gather"] == {"ms": 1.0}
    assert "extras abandoned 3 s after the timed region" in line["side_records_error"]
    assert [l for l in out1.splitlines() if l.strip() and not l.startswith("[Gloo]")] == [] and "leaving" in err1


def test_a_disarmed_deadline_prints_nothing():
    import json
    (rc0, out0, err0), (rc1, out1, err1) = _run_watchdog_case("fine", 2.0)
    assert rc0 == 0 and rc1 == 0, (err0[-800:], err1[-800:])
    lines = [l for l in out0.splitlines() if l.strip() and not l.startswith("[Gloo]")]
    assert len(lines) == 1 and "side_records_error" not in json.loads(lines[0])
    assert [l for l in out1.splitlines() if l.strip() and not l.startswith("[Gloo]")] == [] and "leaving" not in err0 + err1

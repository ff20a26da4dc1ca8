"""bench.py's N > 1 path on a one-GPU box: `python -m torch.distributed.run` starts 2 and 4
fresh ranks — child processes that have not touched the GPU when they are launched — all on cuda:0 with the gloo
backend (TDR_BENCH_REHEARSAL=1).  The numbers mean nothing; what is checked is the code path the driver runs on a
real 8-GPU node: shard ranges, barriers, max-over-ranks timing, the gather (plain and pipelined behind the solve),
the JSON schema with kernel-only and gather-inclusive values side by side, and the parity checks of the N > 1 path
(gathered rows == a local solve of the same systems; rank 0's shard vs the CPU oracle).  Real 2/4/8-GPU numbers
remain the driver's to measure."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("n_ranks", [2, 4])  # + the pytest process itself: at most 5 processes on the card
def test_bench_multi_rank_rehearsal(n_ranks):
    env = dict(os.environ, TDR_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", str(n_ranks), "--steps", "4", "--warmup", "2", "--config", "c2", "--repeats", "2",
           "--gather-chunks", "3", "--side-records", "on"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 prints ONE JSON line
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "gather", "parity"):
        assert key in d, key
    assert d["n_gpus"] == n_ranks and d["steps"] == 4 and d["warmup"] == 2
    assert d["unit"] == "points/s" and d["dtype"] == "f64" and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "REHEARSAL" in d["config"]["workload"] and d["config"]["systems_per_gpu"] == 100_000
    # whole-job aggregate: all ranks' systems over the max-over-ranks time
    assert d["value"] == pytest.approx(100_000 * n_ranks / (d["ms_per_step"] * 1e-3), rel=1e-9)
    g = d["gather"]
    assert g["scaling_kernel_only_value"] == d["value"]
    assert 0 < g["value_with_gather"] < d["value"]  # never hidden: gather-inclusive is its own, smaller number
    assert g["pipelined"]["chunks"] == 3 and g["pipelined"]["value_with_gather_pipelined"] > 0
    assert g["gathered_rows_checked_per_rank"] == 4096 and g["gathered_rows_differing_from_a_local_solve"] == 0
    assert d["parity"]["max_rel_dp_vs_cpu_oracle"] <= 1e-9 and d["parity"]["systems_checked"] == 100_000
    assert d["roofline"]["repeats"]["n"] == 2
    # the north_star design as its own sub-record: FIXED C4 stream block-sharded over the ranks,
    # the pipelined gather inside the timed step, gathered rows checked against local solves
    c4 = d["c4_strong"]
    for key in ("workload", "scaling", "systems_total", "systems_per_gpu", "steps", "collective", "chunks",
                "ms_per_step_incl_gather_pipelined", "value_with_gather_pipelined", "ms_per_step_incl_gather_plain",
                "value_with_gather_plain", "kernel_ms", "scaling_kernel_only_value", "unit",
                "gathered_rows_checked_per_rank", "gathered_rows_differing_from_a_local_solve", "judged_on"):
        assert key in c4, key
    assert c4["scaling"] == "strong" and c4["systems_total"] == c4["systems_per_gpu"] * n_ranks
    assert c4["chunks"] == 3 and c4["gathered_rows_differing_from_a_local_solve"] == 0
    assert 0 < c4["value_with_gather_pipelined"] <= c4["scaling_kernel_only_value"]
    assert 0 < c4["value_with_gather_plain"] <= c4["scaling_kernel_only_value"]
    assert c4["value_with_gather_pipelined"] == pytest.approx(
        c4["systems_total"] / (c4["ms_per_step_incl_gather_pipelined"] * 1e-3), rel=1e-9)
    assert "c4_strong" in d["scaling_target"]


def test_a_wrong_gathered_row_is_loud():
    """when the gathered path of the c4_strong record differs from a local solve the record carries the
    error, the flat throughput key is null, a top-level c4_strong_error says why — and the run exits non-zero AFTER the
    JSON line is out (TDR_BENCH_PLANT_BAD_ROW=1 corrupts one gathered row on rank 0 before the check)."""
    env = dict(os.environ, TDR_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1",
               TDR_BENCH_PLANT_BAD_ROW="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "2", "--warmup", "1", "--config", "c2", "--repeats", "0", "--side-records", "on"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads(lines[0])
    assert r.returncode != 0
    assert d["c4_strong_value_with_gather_pipelined"] is None
    assert "differs from a local solve on 1 rows" in d["c4_strong_error"]
    assert d["c4_strong"]["gathered_rows_differing_from_a_local_solve"] == 1 and "error" in d["c4_strong"]
    assert d["value"] > 0  # the contract line itself is intact


def test_a_rank_that_stalls_after_the_timed_region_cannot_cost_the_contract_line():
    """bench.LineWatchdog in the real command: rank 1 stalls right after the timed region (TDR_BENCH_TEST_STALL_RANK),
    rank 0 goes on into the repeats' all-reduce and blocks there.  At the deadline (TDR_BENCH_WATCHDOG_S) rank 0 prints
    the contract line it already holds, with `side_records_error`, and both ranks exit 0 — the launcher too."""
    import time
    env = dict(os.environ, TDR_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1",
               TDR_BENCH_TEST_STALL_RANK="1", TDR_BENCH_WATCHDOG_S="15")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "2", "--warmup", "1", "--config", "c2", "--repeats", "1", "--side-records", "on"]
    t0 = time.time()
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert time.time() - t0 < 240
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert "extras abandoned 15 s after the timed region" in d["side_records_error"]
    assert d["n_gpus"] == 2 and d["value"] == pytest.approx(200_000 / (d["ms_per_step"] * 1e-3), rel=1e-9)
    assert d["roofline"]["kernel_ms"] > 0 and "c4_strong" not in d and "gather" not in d


def test_a_failed_exchange_step_costs_the_gather_records_but_not_the_line():
    """The contract's timed region runs over the gloo control plane; RCCL (the data group) serves the gather only and is
    created after the timed region.  If it cannot be created (TDR_BENCH_TEST_FAIL_DATA_GROUP plants the failure on every
    rank) the ranks agree over the control plane, drop the gather records and the c4_strong side record together, and
    the contract line comes out whole, with the error named."""
    env = dict(os.environ, TDR_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1",
               TDR_BENCH_TEST_FAIL_DATA_GROUP="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "2", "--warmup", "1", "--config", "c2", "--repeats", "1", "--side-records", "on"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] == pytest.approx(200_000 / (d["ms_per_step"] * 1e-3), rel=1e-9)
    assert "planted: the data group could not be created" in d["gather_error"] and "error" in d["gather"]
    assert d["c4_strong_value_with_gather_pipelined"] is None and d["c4_strong_error"].startswith("skipped: ")
    assert "c4_strong" not in d and "side_records_error" not in d
    assert d["parity"]["max_rel_dp_vs_cpu_oracle"] <= 1e-9  # rank 0's shard is still checked against the oracle

#!/usr/bin/env python3
"""bench.py — triangulated 3D points/s of the HIP hot path on N MI355X (one process per GPU).

    python bench.py --gpus 1 --steps K --warmup W [--config c3|c2|c4|c5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (one kernel launch through the device-resident C-ABI entry
point) over this rank's shard of a synthetic detection stream that is already resident in HBM
(generated on device).  Default workload = BASELINE config 3 (8 cameras / 1 drone /
10M frames per GPU, MatrixTriangulator path) — the configuration the north_star target is quoted
on.  Frames shard over ranks with no data-path collective (weak scaling: every rank owns 10M
frames of one global stream); the single RCCL gather that collects the 3D path on rank 0 is timed
separately and reported under "gather" (with the gather inside the timed region the
xGMI ingest of the root, not the kernel, bounds 8-GPU throughput).  Process groups at N > 1: the contract's barriers
and timing reductions run over the default gloo group (control plane); RCCL is a group of its own that carries the
gather alone, created after the timed region (make_data_group) under a deadline (LineWatchdog).

With no --config (the driver's command) the line carries two more records next to the contract fields, which stay
what they were:
  N = 1  "other_configs": C2, C4, C5 and the int16 format row, each with kernel_ms (median of 5 repeats), value,
         its roofline fraction (HBM; fp64 issue for C5) and a sampled parity against the CPU oracle — so that every
         BASELINE config is in the driver-run record, not only C3.
  N > 1  "c4_strong": BASELINE config 4 as the north_star states it — a FIXED 40 M-system stream block-sharded over
         the N ranks with the single gather pipelined behind the solve INSIDE the timed step — next to the
         weak-scaling `value` (kernel-only, ~N x by construction).  The >= 6 x target is judged on
         c4_strong.value_with_gather_pipelined.

Prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "3d-reconstruction-triangulation_amd"

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
HBM_ACHIEVABLE_GBPS = 6290.0

CONFIGS = {
    # name: (n_cam, n_drones, frames per GPU, path, description)
    "c2": (4, 1, 100_000, "matrix", "C2: synthetic 4-camera / 1-drone / 100k frames, MatrixTriangulator path"),
    "c3": (8, 1, 10_000_000, "matrix", "C3: synthetic 8-camera / 1-drone / 10M frames, MatrixTriangulator path"),
    # separate FORMAT row: the same workload with observations stored as two int16 (4 B) instead
    # of two doubles (16 B); fp64 arithmetic, bit-identical results.  Never the default, never mixed with c3.
    "c3i16": (8, 1, 10_000_000, "matrix_i16",
              "C3 in the int16 observation format: synthetic 8-camera / 1-drone / 10M frames, MatrixTriangulator path"),
    "c4": (8, 4, 10_000_000, "matrix",
           "C4: synthetic 8-camera / 4-drone / 10M frames total, sharded by frame over the GPUs"),
    "c5": (16, 1, 1_000_000, "ray", "C5: synthetic 16-camera / 1-drone / 1M frames, RayTriangulator LM path"),
    # separate ALGORITHM row: the C5 stream through the closed-form minimiser of the ray path's
    # objective (tdr_ray_closest_point_points_device) — the HBM-bound form of config 5.  NOT what the reference's LM
    # run returns where it stops early, so never the default and never mixed with c5's LM figure.
    "c5cf": (16, 1, 1_000_000, "ray_cf",
             "C5 stream in closed form: synthetic 16-camera / 1-drone / 1M frames, minimiser of the RayTriangulator "
             "objective in one pass (not the LM iterate)"),
}

CFG_ID = {"c2": 2, "c3": 3, "c3i16": 3, "c4": 4, "c5": 5, "c5cf": 5}


def make_engine(tdr, sc, name, device):
    n_cam = CONFIGS[name][0]
    cp = sc.ring_camera_params(n_cam, seed=CFG_ID[name])
    cams = [tdr.createCamera(i, c["width"], c["height"], c["focal"], c["tvec"], c["rquat"])
            for i, c in enumerate(cp)]
    P, aux = tdr.pack_cameras(cams)
    return tdr.Engine(P, aux, device=device), P, aux


def sq_summary_path(config):
    """Newest committed SQ-counter summary (rocprofv3 --pmc passes, tools/summarize_sq.py) of a config's kernel."""
    for rnd in ("r05", "r04", "r03", "r02", "r01"):
        p = os.path.join(ROOT, "profiles", f"{rnd}_{config}_summary.json")
        if os.path.exists(p):
            return p
    return None


def valu_roofline(eng, n_sys, kernel_ms, config="c5", waves_per_simd=2):
    """the LM path is VALU-bound — price it against the vector-issue peak.  Instruction counts come from
    the committed SQ-counter pass (rocprofv3 --pmc, profiles/r0N_<config>_summary.json); kernel time and the fp64 issue
    rate (a pure v_fma_f64 loop at the kernel's occupancy: ~56-61 TFLOP/s at two waves per SIMD, ~68 at five, not the
    data sheet's 78.6) are live.  Also used for the int16 format row, which is VALU / HBM co-bound."""
    spath = sq_summary_path(config)
    if spath is None:
        return None
    with open(spath) as f:
        prof = json.load(f)
    per_sys = prof["wave_valu_insts_per_system"]  # wave64 VALU instructions / system
    simds, clock_hz, cycles_per_inst = 256 * 4, 2.4e9, 4  # 16-lane SIMDs: 4 cycles per wave64 op
    issue = per_sys * n_sys * cycles_per_inst / (kernel_ms * 1e-3 * simds * clock_hz)
    tf, ns = eng.time_fp64_fma(waves_per_simd, 20000)
    bound_ms = per_sys * n_sys / simds * ns * 1e-6
    return {"bound": "fp64 VALU issue", "issue_slot_frac_at_2.4GHz": issue,
            "lanes_active_frac": prof["lanes_active_per_valu_inst"], "wave_valu_insts_per_system": per_sys,
            "instruction_counts_source": {"file": os.path.relpath(spath, ROOT),
                                          "measured_at_commit": prof.get("measured_at_commit")},
            "waves_per_simd": waves_per_simd,
            f"measured_fp64_fma_TFLOPs_at_{waves_per_simd}_waves_per_simd": tf,
            "measured_ns_per_wave_inst_per_simd": ns,
            "issue_bound_ms": bound_ms, "frac_of_measured_issue_bound": bound_ms / kernel_ms,
            "note": "instruction counts from the PMC pass in profiles/; kernel time and the fp64 issue rate (pure FMA "
                    "loop at the kernel's occupancy) measured live"}


def side_config_record(tdr, sc, oracle, torch, name, local_rank, repeats=5, parity_sample=4096):
    """One BASELINE config as a side record of the default N = 1 run: device-resident synthetic stream of the
    config's full size, `repeats` groups of back-to-back launches between HIP events on the launch stream (median),
    parity of the first `parity_sample` systems against the CPU oracle.  No CPU baseline here (that is the main
    line's): the whole record costs about a second."""
    import numpy as np
    n_cam, n_drones, frames, path, desc = CONFIGS[name]
    i16 = path == "matrix_i16"
    ray = path == "ray"
    ray_cf = path == "ray_cf"
    n_sys = frames * n_drones
    eng, P, aux = make_engine(tdr, sc, name, local_rank)
    dev = torch.device("cuda", local_rank)
    pts = torch.empty((n_cam, n_sys, 2), dtype=torch.float64, device=dev)
    xyz = torch.empty((n_sys, 3), dtype=torch.float64, device=dev)
    eng.synth_device(0xC0FFEE + CFG_ID[name], n_drones, 0, n_sys, n_sys, pts.data_ptr(), ray_model=ray or ray_cf)
    eng.synchronize()
    pts_in = pts
    if i16:
        pts_in = torch.empty((n_cam, n_sys, 2), dtype=torch.int16, device=dev)
        eng.pack_i16_device(pts.data_ptr(), n_cam, n_sys, n_sys, pts_in.data_ptr(), n_sys)
    path_id = 2 if i16 else (1 if ray else (3 if ray_cf else 0))
    bytes_per_sys = n_cam * (4 if i16 else 16) + 24
    # launches per group sized for ~5 ms of kernel time; the warm-up is long enough for the clock to settle (the
    # round-2 c3i16 line's first region, 20 warm-up launches = 3 ms, still sat on the ramp: 7 % above its repeats)
    est_ms = {"c2": 0.004, "c3i16": 0.15, "c4": 1.0, "c5": 3.0, "c5cf": 0.05}.get(name, 0.25)
    iters = max(3, min(400, int(round(6.0 / est_ms))))
    eng.time_launches(path_id, pts_in.data_ptr(), n_cam, n_sys, n_sys, xyz.data_ptr(), None, warmup=0,
                      iters=max(3, min(2000, int(round(25.0 / est_ms)))))
    ms = sorted(eng.time_launches(path_id, pts_in.data_ptr(), n_cam, n_sys, n_sys, xyz.data_ptr(), None, warmup=0,
                                  iters=iters) for _ in range(repeats))
    med = ms[len(ms) // 2]
    if eng.fetch_first_bad() != -1:
        raise SystemExit(f"{name}: synthetic stream produced a system with < 2 observations")
    achieved = n_sys * bytes_per_sys / (med * 1e-3) / 1e9
    rec = {"workload": desc, "n_cameras": n_cam,
           ("systems_per_step" if path == "matrix" and n_sys > (10 << 20) else "systems_per_launch"): n_sys,
           "path": path,
           "kernel_ms": med, "kernel_ms_min": ms[0], "kernel_ms_max": ms[-1], "repeats": repeats,
           "launches_each": iters, "value": n_sys / (med * 1e-3), "unit": "points/s",
           "algorithmic_bytes_per_system": bytes_per_sys,
           "roofline": {"bound": "fp64 VALU issue" if ray else ("hbm" if not (i16 or name == "c2") else
                                 ("fp64 VALU / load latency (DESIGN.md §4.4)" if i16 else
                                  "launch latency: 8.8 MB, cache-resident (DESIGN.md §4.1)")),
                        "achieved_GBps": achieved, "peak_GBps": HBM_PEAK_GBPS, "hbm_frac": achieved / HBM_PEAK_GBPS}}
    if path == "matrix" and n_sys > (10 << 20):
        # the library cuts a batch beyond 10 Mi systems into launches on slices of the same arrays (kMatrixSlice)
        rec["launches_per_step"] = -(-n_sys // (10 << 20))
    if name == "c4":
        # the compute-free nontemporal stream of the same shape AND footprint (6.08 GB at C4), in this process: says
        # whether a fraction below C3's is the kernel's or the device's at this size
        sink = torch.empty((n_sys,), dtype=torch.float64, device=dev)
        ms_read = sorted(eng.time_stream(0, n_cam, pts.data_ptr(), sink.data_ptr(), n_sys, warmup=3, iters=12)
                         for _ in range(3))[1]
        read_gbps = n_sys * (16 * n_cam + 8) / ms_read / 1e6
        rec["roofline"]["same_box_stream_calibration"] = {
            "read_stream_GBps": read_gbps, "read_stream_ms": ms_read, "footprint_GB": n_sys * (16 * n_cam + 8) / 1e9,
            "note": "n_cam 16-B read rows + 8 B out per lane, no arithmetic, nontemporal, same footprint as the kernel's"}
        rec["roofline"]["frac_of_same_box_read_stream"] = achieved / read_gbps
        del sink
    if (name == "c4" or ray_cf) and n_cam in (4, 8, 16):
        # this kernel's memory traffic without its arithmetic, in this process (tdr_time_stream mode 3: n_cam 16-B rows
        # read + three 8-B stores per lane; C4's footprint in one launch of the twin, the kernel's in four)
        n_tw = n_sys - n_sys % 64
        twin_dst = torch.empty((n_tw, 3), dtype=torch.float64, device=dev)
        ms_twin = sorted(eng.time_stream(3, n_cam, pts.data_ptr(), twin_dst.data_ptr(), n_tw, warmup=3, iters=max(12, iters // 4))
                         for _ in range(3))[1]
        rec["roofline"]["traffic_twin_ms"] = ms_twin * n_sys / n_tw
        rec["roofline"]["frac_of_same_box_traffic_twin"] = (ms_twin * n_sys / n_tw) / med
        del twin_dst
    if ray:
        rec["roofline"]["valu"] = valu_roofline(eng, n_sys, med)
    elif i16:
        rec["roofline"]["valu"] = valu_roofline(eng, n_sys, med, "c3i16", 5)
    # sampled parity against the CPU oracle
    n_par = int(min(parity_sample, n_sys))
    h_pts = pts[:, :n_par].contiguous().cpu().numpy()
    got = xyz[:n_par].cpu().numpy()
    if ray_cf:
        # (a) the C oracle's closed form (reference ray set-up, eigen pseudo-inverse) on the sample; (b) the oracle's LM
        # run on the same sample: every system it reports as converged (status 0) ended at this point
        rc, ref, ref_err, _ = oracle.ray_closest_point_points(aux, h_pts, want_err=True)
        rel = sc.rel_err(got, ref)
        o = oracle.ray_triangulate_points_ex(aux, h_pts, n_threads=os.cpu_count() or 1)
        conv = oracle.lm_comparable(o["status"])
        rel_lm = sc.rel_err(got[conv], o["xyz"][conv])
        rec["parity"] = {"max_rel_dp_vs_cpu_oracle_closed_form": rel, "gate": 1e-4, "systems_checked": n_par,
                         "max_rel_dp_vs_cpu_oracle_LM_where_status_0": rel_lm, "lm_status_0_systems": int(conv.sum())}
        if rc != 0 or not rel_lm <= 1e-7:
            raise SystemExit(f"{name}: closed form vs LM oracle: rc {rc}, {rel_lm}")
        # the same launch with the mean ray distance requested (8 more bytes written per system)
        err_t = torch.empty((n_sys,), dtype=torch.float64, device=dev)
        ms_e = sorted(eng.time_launches(path_id, pts_in.data_ptr(), n_cam, n_sys, n_sys, xyz.data_ptr(),
                                        err_t.data_ptr(), warmup=0, iters=iters) for _ in range(3))[1]
        rec["with_err_output"] = {"kernel_ms": ms_e, "algorithmic_bytes_per_system": bytes_per_sys + 8,
                                  "hbm_frac": n_sys * (bytes_per_sys + 8) / (ms_e * 1e-3) / 1e9 / HBM_PEAK_GBPS}
        np_err = err_t[:n_par].cpu().numpy()
        if not np.allclose(np_err, ref_err, rtol=1e-9):
            raise SystemExit(f"{name}: err output differs from the oracle's")
        del err_t
    elif ray:
        st = torch.empty((n_sys,), dtype=torch.int32, device=dev)
        eng.ray_device_ex(pts.data_ptr(), n_cam, n_sys, n_sys, xyz.data_ptr(), d_status=st.data_ptr())
        eng.synchronize()
        st = st.cpu().numpy()
        o = oracle.ray_triangulate_points_ex(aux, h_pts, n_threads=os.cpu_count() or 1)
        ok = oracle.lm_comparable(st[:n_par], o["status"])
        rel = sc.rel_err(got[ok], o["xyz"][ok])
        rec["parity"] = {"max_rel_dp_vs_cpu_oracle": rel, "gate": 1e-4, "systems_checked": int(ok.sum()),
                         "lm_status_gpu": {"systems": int(n_sys), "cap": int((st & 1).astype(bool).sum()),
                                           "not_converged": int((st & 2).astype(bool).sum())}}
    else:
        rc, ref, _, _ = oracle.matrix_triangulate_points(P, h_pts)
        rel = sc.rel_err(got, ref)
        rec["parity"] = {"max_rel_dp_vs_cpu_oracle": rel, "gate": 1e-4, "systems_checked": n_par}
        if rc != 0:
            raise SystemExit(f"{name}: oracle rc {rc}")
    if not rel <= 1e-4:
        raise SystemExit(f"{name}: parity gate failed: {rel}")
    del pts, xyz, pts_in
    eng.close() if hasattr(eng, "close") else None
    torch.cuda.empty_cache()
    return rec


def phase_together(torch, dist, red_dev, what, phase, fn):
    """Run a rank-LOCAL phase of a multi-rank record, then agree on its outcome: if it failed anywhere, EVERY rank
    raises here — none is left waiting inside the collectives that follow.  Returns fn()'s value."""
    err = out = None
    try:
        out = fn()
    except (Exception, SystemExit) as exc:
        err = exc
    flag = torch.tensor([0 if err is None else 1], dtype=torch.int32, device=red_dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    if int(flag[0]):
        raise RuntimeError(f"{what} abandoned by all ranks in phase '{phase}': "
                           + (f"{type(err).__name__}: {err}" if err is not None else "another rank failed"))
    return out


def make_data_group(torch, dist, local_rank):
    """The RCCL group of the path's one exchange step (the gather of the 3D path onto rank 0): backend nccl over all
    ranks, next to the default gloo group that carries the control plane.  Collective over the default group — every
    rank calls it at the same point."""
    try:
        return dist.new_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    except TypeError:  # a torch without new_group(device_id=...)
        return dist.new_group(backend="nccl")


class LineWatchdog:
    """A deadline for the extras of an N > 1 run.  `start()` arms a daemon thread; if `finish()` has not been called
    `seconds` later, rank 0 writes ONE JSON line — the dict `get_line()` returns at that moment (the contract fields
    from the timed region plus whatever extras were complete), with `side_records_error` naming the deadline — and
    the process leaves through os._exit(0); the other ranks leave the same way without printing.  os._exit, not an
    exception: the main thread is inside a collective that will not return, and interpreter tear-down would wait for
    the communicator.  Exit code 0: the contract line is whole; only extras were cut."""

    def __init__(self, seconds, rank, get_line, out=None):
        import threading
        self.seconds, self.rank, self.get_line = seconds, rank, get_line
        self.out = out if out is not None else sys.stdout
        self.lock = threading.Lock()
        self.done = False
        self.fallback = None
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.wake = threading.Event()

    def start(self):
        if self.rank == 0 and self.get_line is not None:
            self.fallback = json.dumps(self._with_error(dict(self.get_line())))  # serialised while nothing mutates it
        self.thread.start()

    def _with_error(self, line):
        line["side_records_error"] = (f"extras abandoned {self.seconds:.0f} s after the timed region (a rank did not "
                                      "return from a collective or a side record); contract fields are complete")
        return line

    def _run(self):
        if self.wake.wait(self.seconds):
            return  # finish() came first
        with self.lock:
            if self.done:
                return
            if self.rank == 0 and self.get_line is not None:
                try:
                    text = json.dumps(self._with_error(dict(self.get_line())))
                except Exception:  # the main thread was adding to the dict at this very moment
                    text = self.fallback
                self.out.write(text + "\n")
                self.out.flush()
            sys.stderr.write(f"[bench rank {self.rank}] extras did not finish within {self.seconds:.0f} s: leaving\n")
            sys.stderr.flush()
            os._exit(0)

    def finish(self):
        """The main path is about to print: disarm.  (If the deadline won the race the process is already gone.)"""
        with self.lock:
            self.done = True
        self.wake.set()



def c4_strong_record(tdr, sc, sh, torch, dist, rank, world, local_rank, rehearsal, n_chunks, dgroup=None, steps=5):
    """BASELINE config 4 as the north_star states it: a FIXED stream of 8 cameras x 4 drones x 10 M frames (40 M
    systems) block-sharded by frame over the N ranks, no data-path collective, the single gather of the 3D path onto
    rank 0 pipelined behind the solve and INSIDE the timed step (sharding.solve_and_gather_chunked).  One step =
    solve this rank's shard + gather everything; `steps` steps between barrier + device sync on both sides, max over
    ranks.  Also reported: the same shard's kernel-only time (what the weak-scaling headline counts) and the parity of
    the gathered rows against local solves on rank 0."""
    name = "c4"
    n_cam, n_drones, frames, path, desc = CONFIGS[name]
    n_total = frames * n_drones
    if rehearsal:  # every rank on one GPU, gloo through host memory: the code path, not the size
        n_total = 400_000
    n_total -= n_total % world  # equal shards (the pipelined gather's layout); at most world - 1 systems dropped
    sys_start, sys_end = sh.shard_range(n_total, world, rank)
    n_sys = sys_end - sys_start
    dev = torch.device("cuda", local_rank)
    red_dev = torch.device("cpu")  # control plane: the default gloo group; `dgroup` (RCCL) carries the gathers only
    seed = 0xC0FFEE + CFG_ID[name]

    def together(phase, fn):
        return phase_together(torch, dist, red_dev, "c4_strong", phase, fn)

    def setup():
        eng_, P_, aux_ = make_engine(tdr, sc, name, local_rank)
        pts_ = torch.empty((n_cam, n_sys, 2), dtype=torch.float64, device=dev)
        xyz_ = torch.empty((n_sys, 3), dtype=torch.float64, device=dev)
        eng_.synth_device(seed, n_drones, sys_start, n_sys, n_sys, pts_.data_ptr())
        eng_.synchronize()
        return eng_, pts_, xyz_

    eng, pts, xyz = together("setup", setup)

    def launch_chunk(c0, c1, stream):
        eng.matrix_device(pts.data_ptr() + c0 * 16, n_cam, n_sys, c1 - c0, xyz.data_ptr() + c0 * 24, stream=stream)

    def reduce_max(v):
        t = torch.tensor([v], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0])

    # kernel only (warm, then timed with HIP events on the launch stream)
    def kernel_only():
        eng.time_launches(0, pts.data_ptr(), n_cam, n_sys, n_sys, xyz.data_ptr(), None, warmup=0, iters=10)
        return eng.time_launches(0, pts.data_ptr(), n_cam, n_sys, n_sys, xyz.data_ptr(), None, warmup=0, iters=20)

    k_ms = reduce_max(together("kernel-only timing", kernel_only))
    # the one rank-asymmetric allocation (the root's 960 MB receive buffer at C4) as a phase of its own: if it fails
    # on rank 0, every rank leaves here — nobody is left inside the gather that follows
    recv = together("gather buffer", lambda: sh.gather_buffer(xyz, n_total, dst=0, group=dgroup))
    full = sh.solve_and_gather_chunked(launch_chunk, xyz, n_total, n_chunks=n_chunks, out=recv, group=dgroup)  # warm (communicator)
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        full = sh.solve_and_gather_chunked(launch_chunk, xyz, n_total, n_chunks=n_chunks, out=recv, group=dgroup)
    torch.cuda.synchronize()
    dist.barrier()
    step_ms = reduce_max((time.perf_counter() - t0) / steps * 1e3)
    # the plain (unpipelined) form of the same step: solve, then one gather
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.matrix_device(pts.data_ptr(), n_cam, n_sys, n_sys, xyz.data_ptr())
        eng.synchronize()
        full_plain = sh.gather_path(xyz, n_total, dst=0, group=dgroup)
    torch.cuda.synchronize()
    dist.barrier()
    plain_ms = reduce_max((time.perf_counter() - t0) / steps * 1e3)
    rec = None
    n_chk = int(min(4096, n_sys))

    def verify():  # rank 0 only: the first systems of every rank's shard, solved locally, against the rows that arrived
        if os.environ.get("TDR_BENCH_PLANT_BAD_ROW") == "1":  # test hook: one gathered row is wrong
            full[sh.shard_range(n_total, world, world - 1)[0] + 3, 1] += 1.0
        chk_pts = torch.empty((n_cam, n_chk, 2), dtype=torch.float64, device=dev)
        chk_xyz = torch.empty((n_chk, 3), dtype=torch.float64, device=dev)
        bad = 0
        for r in range(world):
            r0, _ = sh.shard_range(n_total, world, r)
            eng.synth_device(seed, n_drones, r0, n_chk, n_chk, chk_pts.data_ptr())
            eng.matrix_device(chk_pts.data_ptr(), n_cam, n_chk, n_chk, chk_xyz.data_ptr())
            eng.synchronize()
            for got in (full, full_plain):
                bad += int((got[r0:r0 + n_chk].to(dev) != chk_xyz).any(dim=1).sum())
        return bad

    bad_rows = together("verification", lambda: verify() if rank == 0 else 0)
    if rank == 0:
        rec = {"workload": desc + (" [REHEARSAL: all ranks on one GPU, gloo, reduced to 400 k systems]"
                                   if rehearsal else ""),
               "scaling": "strong", "systems_total": n_total, "systems_per_gpu": n_sys, "steps": steps,
               "collective": "gloo gather (rehearsal)" if rehearsal else "rccl gather -> rank 0, pipelined behind the solve",
               "chunks": n_chunks,
               "ms_per_step_incl_gather_pipelined": step_ms,
               "value_with_gather_pipelined": n_total / (step_ms * 1e-3),
               "ms_per_step_incl_gather_plain": plain_ms, "value_with_gather_plain": n_total / (plain_ms * 1e-3),
               "kernel_ms": k_ms, "scaling_kernel_only_value": n_total / (k_ms * 1e-3),
               "unit": "points/s",
               "root_ingest_GBps_pipelined": (n_total - n_sys) * 24 / step_ms / 1e6,
               "gathered_rows_checked_per_rank": n_chk, "gathered_rows_differing_from_a_local_solve": bad_rows,
               **({"error": f"gathered path differs from a local solve on {bad_rows} rows"} if bad_rows else {}),
               "judged_on": "the >= 6 x at 8 GPUs target of the north_star is judged on value_with_gather_pipelined of "
                            "this record against the N = 1 run's other_configs.c4.value; the line's top-level `value` is "
                            "weak-scaling kernel-only throughput and is ~N x by construction",
               "note": "SURVEY.md §8e: result bytes are 16 % of the kernel's traffic but the root ingests them at xGMI "
                       "speed, so the gather, not the kernel, bounds this number at 8 GPUs"}
    del full, full_plain, recv, pts, xyz
    torch.cuda.empty_cache()
    return rec


def self_launch(n_ranks):
    """`python bench.py --gpus N ...` started by hand (no RANK / WORLD_SIZE in the environment): start the N ranks the
    way the driver does — `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port <free> bench.py <same arguments>` — as a child process, pass its stdout (rank 0's single JSON line)
    and stderr through, and return its exit code.  The parent never imports torch and never touches a GPU."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS),
                    help="default: c3 plus the side records (other_configs at N = 1, c4_strong at N > 1)")
    ap.add_argument("--side-records", choices=("auto", "on", "off"), default="auto",
                    help="other_configs (N = 1) / c4_strong (N > 1); auto = on when --config is not given")
    ap.add_argument("--variant", type=int, default=None, help="matrix kernel variant (A/B only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=None, help="systems timed on the CPU oracle")
    ap.add_argument("--repeats", type=int, default=5, help="extra repeats of the timed launches (median/min/max)")
    ap.add_argument("--gather-chunks", type=int, default=4, help="pieces of the pipelined gather (N > 1)")
    args = ap.parse_args()
    side_records = args.side_records == "on" or (args.side_records == "auto" and args.config is None)
    side_failure = 0  # exit code after the JSON line when the c4_strong record of an N > 1 run carries an error
    if args.config is None:
        args.config = "c3"

    # `python bench.py --gpus N` without a launcher: this process becomes the launcher.  Decided before torch is
    # imported and before any GPU call — the ranks are FRESH child processes (one per GPU), never a fork or an exec
    # of a process that has touched the card.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))

    import numpy as np
    import torch  # first: its bundled HIP runtime must be the one libtdr_amd.so binds to
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node must equal --gpus")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists)")
    # TDR_BENCH_REHEARSAL=1: every rank on cuda:0 with the gloo backend — exercises the N>1 code path
    # (sharding, barriers, max-over-ranks, gather) on a one-GPU box; its numbers mean nothing.
    rehearsal = os.environ.get("TDR_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # The path shards with NO data-path collective, so the contract's timed region needs a control plane only: the
        # barriers and the MAX over ranks of two floats.  That runs over gloo — the backend every N > 1 test of this
        # repo has exercised (world 2 and 4) — so that the contract line cannot be lost to the communicator of the one
        # real exchange step, the gather of the 3D path.  RCCL serves exactly that step, in a group of its own
        # (`data_group()`), created after the timed region under the LineWatchdog.
        dist.init_process_group("gloo", rank=rank, world_size=world)

    tdr = importlib.import_module(PKG)
    sc = importlib.import_module(PKG + ".scenes")
    sh = importlib.import_module(PKG + ".sharding")

    n_cam, n_drones, frames, path, desc = CONFIGS[args.config]
    if args.config == "c4":  # strong: 10M frames x 4 drones in total, block-sharded over ranks
        n_total = frames * n_drones
        sys_start, sys_end = sh.shard_range(n_total, world, rank)
        scaling = "strong"
    else:  # weak: every rank owns `frames` frames of one global stream
        n_total = frames * n_drones * world
        sys_start, sys_end = sh.shard_range(n_total, world, rank)
        scaling = "weak"
    n_sys = sys_end - sys_start
    i16 = path == "matrix_i16"
    if i16:
        path = "matrix"
    ray_cf = path == "ray_cf"
    steps = args.steps if args.steps is not None else (200 if path != "ray" else 20)
    warmup = args.warmup if args.warmup is not None else (20 if path != "ray" else 5)
    path_id = 2 if i16 else (0 if path == "matrix" else (3 if ray_cf else 1))
    bytes_per_sys = n_cam * (4 if i16 else 16) + 24  # Point2d reads + Point3d write

    cfg_id = CFG_ID[args.config]
    eng, P, aux = make_engine(tdr, sc, args.config, local_rank)
    if args.variant is not None:
        eng.set_matrix_variant(args.variant)

    dev = torch.device("cuda", local_rank)
    pts = torch.empty((n_cam, n_sys, 2), dtype=torch.float64, device=dev)
    xyz = torch.empty((n_sys, 3), dtype=torch.float64, device=dev)
    seed = 0xC0FFEE + cfg_id
    eng.synth_device(seed, n_drones, sys_start, n_sys, n_sys, pts.data_ptr(),
                     ray_model=(path in ("ray", "ray_cf")))
    eng.synchronize()
    pts_in = pts
    if i16:  # the stream's detections are integer-valued (rounded like the CSV data): pack them once, untimed
        pts_in = torch.empty((n_cam, n_sys, 2), dtype=torch.int16, device=dev)
        eng.pack_i16_device(pts.data_ptr(), n_cam, n_sys, n_sys, pts_in.data_ptr(), n_sys)

    def barrier():
        if world > 1:
            dist.barrier()

    # ---- warmup (untimed)
    if warmup > 0:
        eng.time_launches(path_id, pts_in.data_ptr(), n_cam, n_sys, n_sys, xyz.data_ptr(), None,
                          warmup=0, iters=warmup)
    # ---- timed region: exactly `steps` launches, barrier + device sync on both sides
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kernel_ms = eng.time_launches(path_id, pts_in.data_ptr(), n_cam, n_sys, n_sys, xyz.data_ptr(), None,
                                  warmup=0, iters=steps)  # HIP events on the launch stream
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    if eng.fetch_first_bad() != -1:
        raise SystemExit("synthetic stream produced a system with < 2 observations")
    red_dev = torch.device("cpu")  # the control plane is gloo (see above)
    elapsed = torch.tensor([t1 - t0, kernel_ms], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    wall_s, kernel_ms_max = float(elapsed[0]), float(elapsed[1])
    ms_per_step = wall_s / steps * 1e3
    value = n_total * steps / wall_s

    # ---- the contract line exists from here on (rank 0); everything below only ADDS to it
    result = None
    if rank == 0:
        achieved = n_sys * bytes_per_sys / (kernel_ms_max * 1e-3) / 1e9
        traffic = traffic_src = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):  # HBM bytes/launch from the separate rocprofv3 --pmc passes
            with open(tpath) as f:
                tj = json.load(f).get(args.config, {})
            traffic = tj.get("traffic_bytes_per_launch")
            traffic_src = {"file": "profiles/traffic.json", "measured_at_commit": tj.get("measured_at_commit"),
                           "note": "separate rocprofv3 --pmc passes (FETCH_SIZE x2 on gfx950, WRITE_SIZE), not this run"}
        metric = "triangulated 3D points/sec at 1/2/4/8 MI355X; max |\u0394p| vs CPU ref"
        try:  # BASELINE.json's own wording, when the file travels with the repo
            with open(os.path.join(ROOT, "BASELINE.json")) as f:
                metric = json.load(f).get("metric", metric)
        except OSError:
            pass
        result = {
            "metric": metric,
            "value": value, "unit": "points/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": desc + (" [REHEARSAL: all ranks on one GPU, gloo]" if rehearsal else ""),
                       "n_cameras": n_cam, "n_drones": n_drones,
                       "systems_per_gpu": n_sys, "path": path,
                       "layout": "pts[cam][sys] 4 B (int16 x, y), xyz[sys] 24 B" if i16 else "pts[cam][sys] 16 B, xyz[sys] 24 B",
                       "parallelism": f"frames block-sharded over {world} GPU(s), no data-path collective"},
            "roofline": {
                # the LM path is fp64-VALU/divergence bound; its HBM figure is reported for
                # completeness only
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                "frac_of_achievable_6.29TBps": achieved / HBM_ACHIEVABLE_GBPS,
                "kernel_ms": kernel_ms_max, "algorithmic_bytes_per_system": bytes_per_sys,
                ("systems_per_step" if path == "matrix" and not i16 and n_sys > (10 << 20) else "systems_per_launch"): n_sys,
                "traffic_source": traffic_src,
            },
        }

    # Everything after the timed region is extra: repeats, the gather measurements, calibration, side records.  At
    # N > 1 most of it sits on collectives that have never run on more than one real GPU in this pipeline, so a
    # deadline guards the contract line: if the extras have not finished by then, rank 0 prints the line as it stands
    # (with `side_records_error`) and every rank leaves through os._exit — no rank is left inside a collective.
    watchdog = None
    if world > 1:
        watchdog = LineWatchdog(float(os.environ.get("TDR_BENCH_WATCHDOG_S", "300")), rank,
                                (lambda: result) if rank == 0 else None)
        watchdog.start()
        if os.environ.get("TDR_BENCH_TEST_STALL_RANK") == str(rank):  # test hook: this rank never reaches the extras
            time.sleep(1e6)

    # ---- repeats (median of >= 5): the same `steps` launches again, HIP events on the launch
    # stream, outside the contract's timed region above — `value` / `ms_per_step` are from that one region only
    n_rep = max(0, args.repeats)
    rep_ms = [eng.time_launches(path_id, pts_in.data_ptr(), n_cam, n_sys, n_sys, xyz.data_ptr(), None, warmup=0,
                                iters=steps) for _ in range(n_rep)]
    if world > 1 and n_rep:
        rt = torch.tensor(rep_ms, dtype=torch.float64, device=red_dev)
        dist.all_reduce(rt, op=dist.ReduceOp.MAX)  # per repeat, the slowest rank
        rep_ms = [float(v) for v in rt]

    # ---- the single RCCL gather of the 3D path (timed separately, never hidden)
    gather = None
    dgroup = None  # rehearsal: the default (gloo) group, device tensors staged through the host by sharding.py
    data_err = None  # the exchange step's own failure (RCCL): it may cost the gather records, never the line
    full = full_p = None
    try:
        if os.environ.get("TDR_BENCH_TEST_FAIL_DATA_GROUP") == "1" and world > 1:  # test hook
            raise RuntimeError("planted: the data group could not be created")
        if world > 1 and not rehearsal:
            dgroup = make_data_group(torch, dist, local_rank)  # RCCL: every rank creates it here, after the timed region
        if world > 1:
            full = sh.gather_path(xyz, n_total, dst=0, group=dgroup)  # warm
            torch.cuda.synchronize()
            barrier()
            g0 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                full = sh.gather_path(xyz, n_total, dst=0, group=dgroup)
            torch.cuda.synchronize()
            barrier()
            g_ms = (time.perf_counter() - g0) / reps * 1e3
            gt = torch.tensor([g_ms], dtype=torch.float64, device=red_dev)
            dist.all_reduce(gt, op=dist.ReduceOp.MAX)
            g_ms = float(gt[0])
            gather = {
                "collective": "gloo gather (rehearsal)" if rehearsal else "rccl gather -> rank 0",
                "ms": round(g_ms, 4),
                "root_ingest_GBps": round((n_total - n_sys) * 24 / g_ms / 1e6, 1),
                "value_with_gather": n_total / ((ms_per_step + g_ms) * 1e-3),
                "scaling_kernel_only_value": value,
            }
            if rank == 0:
                result["gather"] = gather  # attached at once: a deadline (LineWatchdog) prints what is complete
            # the same step with the gather PIPELINED behind the solve: the shard is cut into pieces,
            # piece c is gathered while piece c + 1 is solved (sharding.solve_and_gather_chunked).  Equal shards only.
            full_p = None
            if n_sys * world == n_total and path == "matrix" and not i16:
                def launch_chunk(c0, c1, stream):
                    eng.matrix_device(pts.data_ptr() + c0 * 16, n_cam, n_sys, c1 - c0, xyz.data_ptr() + c0 * 24,
                                      stream=stream)
                try:  # an extra measurement: it must never cost the bench line (every rank takes the same branch)
                    full_p = sh.solve_and_gather_chunked(launch_chunk, xyz, n_total, n_chunks=args.gather_chunks, group=dgroup)  # warm
                    torch.cuda.synchronize()
                    barrier()
                    p0 = time.perf_counter()
                    for _ in range(reps):
                        full_p = sh.solve_and_gather_chunked(launch_chunk, xyz, n_total, n_chunks=args.gather_chunks, group=dgroup)
                    torch.cuda.synchronize()
                    barrier()
                    p_ms = (time.perf_counter() - p0) / reps * 1e3
                    pt = torch.tensor([p_ms], dtype=torch.float64, device=red_dev)
                    dist.all_reduce(pt, op=dist.ReduceOp.MAX)
                    gather["pipelined"] = {"chunks": args.gather_chunks, "ms_per_step_incl_gather": round(float(pt[0]), 4),
                                           "value_with_gather_pipelined": n_total / (float(pt[0]) * 1e-3)}
                except RuntimeError as exc:  # pragma: no cover - only on a backend that refuses the async gather
                    full_p = None
                    gather["pipelined"] = {"error": str(exc)[:300]}
            # Sampled parity of the GATHERED path on rank 0 (N > 1 used to skip parity altogether): the synthetic
            # stream is a pure function of the global system index, so rank 0 re-creates the first systems of every
            # rank's shard, solves them locally and compares them with the rows that arrived — bit for bit.
            if rank == 0:
                n_chk = int(min(4096, n_sys))
                chk_pts = torch.empty((n_cam, n_chk, 2), dtype=torch.float64, device=dev)
                chk_xyz = torch.empty((n_chk, 3), dtype=torch.float64, device=dev)
                bad_rows = 0
                for r in range(world):
                    r0, _ = sh.shard_range(n_total, world, r)
                    eng.synth_device(seed, n_drones, r0, n_chk, n_chk, chk_pts.data_ptr(),
                                     ray_model=(path in ("ray", "ray_cf")))
                    if ray_cf:
                        eng.ray_closest_point_device(chk_pts.data_ptr(), n_cam, n_chk, n_chk, chk_xyz.data_ptr())
                    elif path == "ray":
                        eng.ray_device(chk_pts.data_ptr(), n_cam, n_chk, n_chk, chk_xyz.data_ptr())
                    else:
                        eng.matrix_device(chk_pts.data_ptr(), n_cam, n_chk, n_chk, chk_xyz.data_ptr())
                    eng.synchronize()
                    for got in (full, full_p):
                        if got is not None:
                            bad_rows += int((got[r0:r0 + n_chk].to(dev) != chk_xyz).any(dim=1).sum())
                gather["gathered_rows_checked_per_rank"] = n_chk
                gather["gathered_rows_differing_from_a_local_solve"] = bad_rows
                if bad_rows:
                    raise SystemExit(f"gathered path differs from a local solve on {bad_rows} rows")
            del full, full_p
    except Exception as exc:  # SystemExit (a WRONG gathered row) is not caught: that must stay loud
        data_err = f"{type(exc).__name__}: {exc}"[:400]
    if world > 1:
        # the ranks agree over the control plane whether the exchange step worked everywhere; if not, every rank
        # drops the gather records and the c4_strong side record together (a rank still inside a collective of the
        # data group never reaches this point: the LineWatchdog prints the line then)
        flag = torch.tensor([1 if data_err else 0], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag[0]):
            data_err = data_err or "another rank failed in the exchange step"
            gather = {"error": data_err}
            if rank == 0:
                result["gather"] = gather
                result["gather_error"] = data_err

    # ---- N > 1: parity of rank 0's own shard against the CPU oracle on a sample (the N = 1 run checks every
    # system of the workload; the gathered rows were compared with local solves above)
    parity_multi = None
    if world > 1 and rank == 0 and path == "matrix":
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import tdr_oracle as oracle_m
        n_par = int(min(100_000, n_sys))
        rc_m, ref_m, _, _ = oracle_m.matrix_triangulate_points(P, pts[:, :n_par].contiguous().cpu().numpy())
        rel_m = sc.rel_err(xyz[:n_par].cpu().numpy(), ref_m)
        parity_multi = {"max_rel_dp_vs_cpu_oracle": rel_m, "gate": 1e-4, "systems_checked": n_par,
                        "note": "rank 0's shard, sampled; N = 1 checks the whole workload"}
        if rc_m != 0 or not rel_m <= 1e-4:
            raise SystemExit(f"parity gate failed on rank 0: {rel_m}")

    # ---- calibration: a plain nontemporal stream of the same shape on this device (rank 0, matrix path)
    calib = None
    if rank == 0 and path == "matrix" and not i16 and n_cam in (4, 8, 16):
        sink = torch.empty((n_sys,), dtype=torch.float64, device=dev)
        ms_read = eng.time_stream(0, n_cam, pts.data_ptr(), sink.data_ptr(), n_sys, warmup=5, iters=50)
        n_copy = min(n_sys * n_cam, n_sys * 3 // 2)
        copy_dst = torch.empty((n_copy, 2), dtype=torch.float64, device=dev)  # never the result buffer
        ms_copy = eng.time_stream(1, 8, pts.data_ptr(), copy_dst.data_ptr(), n_copy, warmup=5, iters=50)
        calib = {"read_stream_GBps": n_sys * (16 * n_cam + 8) / ms_read / 1e6,
                 "copy_GBps": n_copy * 32 / ms_copy / 1e6,
                 "note": "plain nontemporal streams measured in this process: n_cam 16-B read rows + 8 B out per "
                         "lane (no arithmetic), and a 16-B/lane copy"}
        del sink, copy_dst
        # the kernel's own access pattern without its arithmetic (64-thread blocks): the read side alone, and the
        # "traffic twin" — the same reads + the three 8-B result stores per lane.  kernel_ms / twin_ms ~ 1 says the
        # kernel IS its memory traffic; pure_read vs twin says what interleaving the writes costs the memory system
        n_tw = n_sys - n_sys % 64
        if n_tw > 0:
            twin_dst = torch.empty((n_tw, 3), dtype=torch.float64, device=dev)
            ms_pure = eng.time_stream(2, n_cam, pts.data_ptr(), twin_dst.data_ptr(), n_tw, warmup=5, iters=50)
            ms_twin = eng.time_stream(3, n_cam, pts.data_ptr(), twin_dst.data_ptr(), n_tw, warmup=5, iters=50)
            calib.update({"pure_read_GBps": n_tw * 16 * n_cam / ms_pure / 1e6,
                          "traffic_twin_ms": ms_twin * n_sys / n_tw,
                          "traffic_twin_GBps": n_tw * bytes_per_sys / ms_twin / 1e6,
                          "twin_note": "this kernel's access pattern without its arithmetic: the read side alone, and "
                                       "the reads + three 8-B result stores per lane (its memory traffic, byte for byte)"})
            del twin_dst

    if rank == 0:
        if rep_ms:
            srt = sorted(rep_ms)
            med = srt[len(srt) // 2]
            result["roofline"]["repeats"] = {
                "n": len(srt), "launches_each": steps, "kernel_ms_median": med, "kernel_ms_min": srt[0],
                "kernel_ms_max": srt[-1], "achieved_GBps_at_median": n_sys * bytes_per_sys / (med * 1e-3) / 1e9,
                "value_at_median_kernel_ms": n_total / (med * 1e-3)}
        if calib:
            result["roofline"]["same_box_stream_calibration"] = calib
            result["roofline"]["frac_of_same_box_read_stream"] = achieved / calib["read_stream_GBps"]
            if "traffic_twin_GBps" in calib:
                result["roofline"]["frac_of_same_box_traffic_twin"] = achieved / calib["traffic_twin_GBps"]
        if path == "ray" and args.config == "c5":
            valu = valu_roofline(eng, n_sys, kernel_ms_max)
            if valu:
                result["roofline"]["valu"] = valu
        if gather:
            result["gather"] = gather
        if parity_multi:
            result["parity"] = parity_multi

    # ---- CPU baseline: the oracle (a port; the reference needs OpenCV and cannot be built here)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import tdr_oracle as oracle
        n_threads_all = os.cpu_count() or 1
        if path == "matrix":
            sample = args.cpu_sample or min(n_sys, 10_000_000)
            h_pts = pts[:, :sample].contiguous().cpu().numpy()
            oracle.matrix_triangulate_points(P, h_pts[:, :1000])  # page in / warm
            c0 = time.perf_counter()
            rc, ref, _, _ = oracle.matrix_triangulate_points(P, h_pts, n_threads=1)
            c_s = time.perf_counter() - c0
            assert rc == 0
            f0 = time.perf_counter()
            rc, ref_fast, _, _ = oracle.matrix_triangulate_points(P, h_pts, n_threads=n_threads_all, fast=True)
            f_s = time.perf_counter() - f0
            extra = {"best_effort_value": sample / f_s, "best_effort_cores": n_threads_all,
                     "best_effort_note": "flat arrays + normal equations, all host threads"}
            # SURVEY.md §8d flavour (i): the same arithmetic inside the reference's own data movement (nested vectors by
            # value, a heap-grown vector per point, by-value matrix headers, a heap matrix per expression) — a cost
            # model (oracle/ref_cost_model.cpp), bit-identical results, on a 1 M-system sample
            n_rs = min(sample, 1_000_000)
            rc_r, ref_r, rs_s = oracle.refcost_matrix_triangulate_points(P, h_pts[:, :n_rs])
            assert rc_r == 0 and bool((ref_r == ref[:n_rs]).all())
            extra.update({"reference_shaped_value": n_rs / rs_s, "reference_shaped_cores": 1,
                          "reference_shaped_sample": f"first {n_rs} systems ({rs_s:.1f} s)",
                          "reference_shaped_note": "cost model of src/MatrixTriangulator.cpp:70-100 / :3-62 as written "
                          "(containers by value, vector per point, 14 by-value matrix headers per camera, heap matrix per "
                          "expression) around the port's arithmetic; omits OpenCV's own overheads, so still an upper "
                          "bound on the reference; bit-identical to the port"})
            kind_note = "faithful-cost port: per-system one-sided Jacobi SVD pseudo-inverse, 1 thread"
        elif ray_cf:
            sample = args.cpu_sample or min(n_sys, 1_000_000)
            h_pts = pts[:, :sample].contiguous().cpu().numpy()
            c0 = time.perf_counter()
            rc, ref, _, _ = oracle.ray_closest_point_points(aux, h_pts)
            c_s = time.perf_counter() - c0
            assert rc == 0
            extra = {}
            kind_note = ("closed-form minimiser from the reference's ray set-up (oracle_ray_closest_point), 1 thread; "
                         "the reference itself runs cv::LMSolver here: see config c5")
        else:
            sample = args.cpu_sample or min(n_sys, 100_000)
            h_pts = pts[:, :sample].contiguous().cpu().numpy()
            c0 = time.perf_counter()
            rc, ref, _, _, _ = oracle.ray_triangulate_points(aux, h_pts, n_threads=1)
            c_s = time.perf_counter() - c0
            assert rc == 0
            # Runs that stop away from the minimiser of their own objective (iteration cap / damped step) end at
            # a chaotic iterate.  Both sides flag them; a system is compared iff both report 0.
            o_ex = oracle.ray_triangulate_points_ex(aux, h_pts, n_threads=n_threads_all)  # untimed: status words
            st_gpu = torch.empty((n_sys,), dtype=torch.int32, device=dev)
            eng.ray_device_ex(pts.data_ptr(), n_cam, n_sys, n_sys, xyz.data_ptr(), d_status=st_gpu.data_ptr())
            eng.synchronize()
            st_gpu = st_gpu.cpu().numpy()
            comparable = oracle.lm_comparable(st_gpu[:sample], o_ex["status"])
            extra = {"lm_status_gpu": {"systems": int(n_sys), "cap": int((st_gpu & 1).astype(bool).sum()),
                                       "not_converged": int((st_gpu & 2).astype(bool).sum())},
                     "lm_status_flags_differ_gpu_vs_oracle_in_sample":
                         int((((st_gpu[:sample] & 3) != 0) != ((o_ex["status"] & 3) != 0)).sum())}
            kind_note = "port of the LM loop (cv::LMSolver schedule), 1 thread"
        if path == "matrix" and not i16:
            # the drop-in call itself: pageable host buffers in, pageable host buffer out (PCIe-inclusive; NEVER
            # `value`).  Same detections, result must equal the device-resident launch bit for bit.
            h_times = []
            h_xyz = None
            for _ in range(6):
                del h_xyz  # freeing the previous 240 MB result (munmap, ~10 ms) is not part of the call
                h0 = time.perf_counter()
                h_xyz = eng.matrix_triangulate_points(h_pts)
                h_times.append(time.perf_counter() - h0)
            h_times = sorted(h_times[1:])  # the first call pins fresh pages and sizes the staging slots
            same = bool((h_xyz == xyz[:sample].cpu().numpy()).all())
            # the same call with the detections kept as doubles across the link (mode 2): the default (mode 3) sends
            # integer-valued detections as int16 pairs and solves them with the int16 kernel, bit-identically
            eng.set_host_pipeline(2, 0)
            d_times = []
            for _ in range(4):
                del h_xyz
                h0 = time.perf_counter()
                h_xyz = eng.matrix_triangulate_points(h_pts)
                d_times.append(time.perf_counter() - h0)
            eng.set_host_pipeline(-1, 0)
            d_times = sorted(d_times[1:])
            same = same and bool((h_xyz == xyz[:sample].cpu().numpy()).all())
            link_bytes = n_cam * 4 + 24
            result["host_buffer_path"] = {
                "entry_point": "tdr_matrix_triangulate_points (pageable host memory in / out)",
                "systems": sample, "ms_best": h_times[0] * 1e3, "ms_median": h_times[len(h_times) // 2] * 1e3,
                "points_per_s_best": sample / h_times[0],
                "transport": "int16 pairs up (4 B per observation: the synthetic detections are integer-valued, like "
                             "the reference's std::stoi data), doubles down",
                "link_bytes_per_system": link_bytes,
                "GBps_over_pcie_best": sample * link_bytes / h_times[0] / 1e9,
                "double_transport": {"ms_best": d_times[0] * 1e3, "ms_median": d_times[len(d_times) // 2] * 1e3,
                                     "points_per_s_best": sample / d_times[0],
                                     "GBps_over_pcie_best": sample * bytes_per_sys / d_times[0] / 1e9},
                "bit_identical_to_device_resident_launch": same,
                "note": "PCIe-inclusive rate of the drop-in call; never `value` (DESIGN.md §5)"}
            del h_xyz
        if path == "ray":
            # the LM path through the drop-in call (pageable host memory in / out), whole batch; NEVER `value`
            h_all = pts.contiguous().cpu().numpy()
            dev_xyz = xyz.cpu().numpy()
            rows = {}
            for label, mode in (("default", -1), ("double_transport", 2)):
                eng.set_host_pipeline(mode, 0)
                t_h = []
                h_xyz = None
                for _ in range(5):
                    del h_xyz
                    h0 = time.perf_counter()
                    h_xyz = eng.ray_triangulate_points(h_all)
                    t_h.append(time.perf_counter() - h0)
                t_h = sorted(t_h[1:])
                rows[label] = {"ms_best": t_h[0] * 1e3, "ms_median": t_h[len(t_h) // 2] * 1e3,
                               "points_per_s_best": n_sys / t_h[0],
                               "bit_identical_to_device_resident_launch": bool((h_xyz == dev_xyz).all())}
            eng.set_host_pipeline(-1, 0)
            del h_xyz, h_all, dev_xyz
            result["host_buffer_path"] = {
                "entry_point": "tdr_ray_triangulate_points (pageable host memory in / out)", "systems": n_sys,
                "transport": "int16 pairs up (integer-valued detections), unpacked on the device in front of the LM "
                             "kernel; chunks of 131072 systems alternate between two compute streams",
                **rows["default"], "double_transport": rows["double_transport"],
                "note": "PCIe-inclusive rate of the drop-in call; never `value` (DESIGN.md §5)"}
        got = xyz[:sample].cpu().numpy()
        if path == "ray":
            got, ref = got[comparable], ref[comparable]
        rel = sc.rel_err(got, ref)
        result["cpu_baseline"] = {
            "value": sample / c_s, "unit": "points/s", "cores": 1, "kind": "port",
            "sample": f"first {sample} systems of the same stream ({c_s:.1f} s of CPU work); {kind_note}",
            "host_cpus": n_threads_all, **extra,
        }
        result["parity"] = {"max_rel_dp_vs_cpu_oracle": rel, "gate": 1e-4, "systems_checked": sample}
        if not rel <= 1e-4:
            raise SystemExit(f"parity gate failed: {rel}")

    # ---- side records of the default command (see the module docstring)
    if side_records and world == 1:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import tdr_oracle as oracle_s
        del pts, xyz, pts_in
        torch.cuda.empty_cache()
        t_side = time.perf_counter()
        oc = {}
        for name, key in (("c2", "c2"), ("c4", "c4"), ("c5", "c5"), ("c5cf", "c5_closed_form"), ("c3i16", "c3i16")):
            try:  # a side record must never cost the contract line: a failure becomes the record
                oc[key] = side_config_record(tdr, sc, oracle_s, torch, name, local_rank)
            except (Exception, SystemExit) as exc:
                oc[key] = {"error": f"{type(exc).__name__}: {exc}"[:400]}
                torch.cuda.empty_cache()
        # flat copies of the two records the scaling / LM targets are read from, placed IN FRONT of the long nested
        # object: a consumer that cuts the tail of the line still has them (round 3's driver record was cut inside
        # other_configs)
        for key, (name, field) in {"c4_value": ("c4", "value"), "c4_kernel_ms": ("c4", "kernel_ms"),
                                   "c4_hbm_frac": ("c4", "hbm_frac"), "c5_value": ("c5", "value"),
                                   "c5_kernel_ms": ("c5", "kernel_ms"),
                                   "c5_closed_form_kernel_ms": ("c5_closed_form", "kernel_ms"),
                                   "c5_closed_form_hbm_frac": ("c5_closed_form", "hbm_frac")}.items():
            rec_n = oc[name]
            result[key] = rec_n.get(field, rec_n.get("roofline", {}).get(field)) if "error" not in rec_n else None
        oc["seconds"] = round(time.perf_counter() - t_side, 2)
        oc["note"] = ("every BASELINE config next to the C3 contract line: kernel_ms = median of 5 groups of back-to-back "
                      "launches (HIP events on the launch stream), inputs resident in HBM; c3i16 is a separate FORMAT row "
                      "(4-B observations), never mixed with C3; c5_closed_form is a separate ALGORITHM row (the minimiser "
                      "of the ray objective in one pass, not the reference's LM iterate), never mixed with C5")
        result["other_configs"] = oc
    if side_records and world > 1 and data_err:
        if rank == 0:
            result["c4_strong_value_with_gather_pipelined"] = None
            result["c4_strong_error"] = "skipped: " + data_err
    elif side_records and world > 1:
        del pts, xyz, pts_in
        torch.cuda.empty_cache()
        # a side record must never cost the contract line: c4_strong_record agrees on an ok-flag over all ranks after
        # every rank-local phase, so that the ranks abandon it TOGETHER instead of leaving some inside a collective
        try:
            rec = c4_strong_record(tdr, sc, sh, torch, dist, rank, world, local_rank, rehearsal, args.gather_chunks,
                                   dgroup=dgroup)
        except (Exception, SystemExit) as exc:
            rec = {"error": f"{type(exc).__name__}: {exc}"[:400]}
            torch.cuda.empty_cache()
        if rank == 0:
            # a record that carries an error publishes NO throughput in the flat keys (a wrong gather must not
            # yield a clean-looking scaling number), and the run exits non-zero after the line is out
            failed = "error" in (rec or {"error": "no record"})
            result["c4_strong_value_with_gather_pipelined"] = None if failed else rec.get("value_with_gather_pipelined")
            if failed:
                result["c4_strong_error"] = (rec or {}).get("error", "no record")
                side_failure = 3
            result["c4_strong"] = rec
            result["scaling_target"] = ("north_star '>= 6x at 8 GPUs': judged on c4_strong.value_with_gather_pipelined "
                                        "(fixed 40 M systems, gather inside the step), NOT on the weak-scaling `value`")

    if watchdog is not None:
        watchdog.finish()  # from here on the main path owns stdout (a deadline that fired earlier has already exited)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        barrier()  # rank 0 may still be calibrating: tear the communicator down together
        dist.destroy_process_group()
    if side_failure:
        raise SystemExit(side_failure)


if __name__ == "__main__":
    main()

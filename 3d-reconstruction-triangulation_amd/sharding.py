"""Multi-GPU host logic of the path: one process per GPU, systems block-partitioned
over ranks, no collective on the data path.  The only exchanges are
  - ONE gather of the [n_local][3] result blocks onto the root (RCCL over xGMI on GPUs; the same
    code runs on gloo/CPU tensors in the tests), and
  - an 8-byte MIN all-reduce of the first failing system index, so every rank raises the
    reference's "Too few rays are found" for the same (globally first) frame.
torch.distributed is plumbing here; the solves themselves are the HIP kernels.
"""
import torch
import torch.distributed as dist


def shard_range(n_sys, world, rank):
    """Contiguous block partition: rank g owns [g*n/G, (g+1)*n/G)."""
    return (n_sys * rank) // world, (n_sys * (rank + 1)) // world


def shard_counts(n_sys, world):
    return [shard_range(n_sys, world, r)[1] - shard_range(n_sys, world, r)[0] for r in range(world)]


def gather_path(xyz_local, n_sys, dst=0, group=None):
    """Collect the sharded 3D path on rank `dst` with a single gather.
    xyz_local: [n_local, 3] float64 tensor (n_local = this rank's shard_range size).
    Returns the [n_sys, 3] tensor on dst, None elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if xyz_local.is_cuda and dist.get_backend(group) == "gloo":
        xyz_local = xyz_local.cpu()  # gloo has no device gather (one-GPU rehearsal of the N>1 path)
    counts = shard_counts(n_sys, world)
    assert xyz_local.shape == (counts[rank], 3), (tuple(xyz_local.shape), counts[rank])
    cmax = max(counts)
    if counts[rank] != cmax:  # gather needs equal blocks: pad the (at most one row) difference
        pad = torch.zeros((cmax - counts[rank], 3), dtype=xyz_local.dtype, device=xyz_local.device)
        send = torch.cat([xyz_local, pad], dim=0)
    else:
        send = xyz_local.contiguous()
    if rank == dst:
        # blocks land directly in their place of the [world][cmax][3] result: no second pass over the path
        full = torch.empty((world, cmax, 3), dtype=xyz_local.dtype, device=xyz_local.device)
        recv = list(full.unbind(0))
        dist.gather(send, recv, dst=dst, group=group)
        if min(counts) == cmax:
            return full.view(world * cmax, 3)
        return torch.cat([recv[r][:counts[r]] for r in range(world)], dim=0)
    dist.gather(send, None, dst=dst, group=group)
    return None


def global_first_bad(local_first_bad, shard_start, device=None, group=None):
    """Lowest failing GLOBAL system index over all ranks (-1 if none)."""
    big = torch.iinfo(torch.int64).max
    v = big if local_first_bad < 0 else shard_start + local_first_bad
    t = torch.tensor([v], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    v = int(t[0])
    return -1 if v == big else v


def chunk_ranges(n_local, n_chunks):
    """[(c0, c1)] covering [0, n_local) in n_chunks nearly equal pieces (empty pieces dropped)."""
    n_chunks = max(1, min(int(n_chunks), max(1, n_local)))
    out = [((n_local * c) // n_chunks, (n_local * (c + 1)) // n_chunks) for c in range(n_chunks)]
    return [(a, b) for a, b in out if b > a]


def gather_buffer(xyz_local, n_sys, dst=0, group=None):
    """The root's [world, n_local, 3] receive buffer of solve_and_gather_chunked (None on the other ranks), allocated
    on its own so that a caller can make this — the one rank-ASYMMETRIC allocation of the gather, 960 MB at C4 — a
    step whose failure every rank learns about before anybody enters the collective (bench.py: phase_together)."""
    world = dist.get_world_size(group)
    if dist.get_rank(group) != dst:
        return None
    n_local = xyz_local.shape[0]
    on_gpu = xyz_local.is_cuda and dist.get_backend(group) != "gloo"
    return torch.empty((world, n_local, 3), dtype=xyz_local.dtype, device=xyz_local.device if on_gpu else "cpu")


def solve_and_gather_chunked(launch_chunk, xyz_local, n_sys, n_chunks=4, dst=0, group=None, out=None):
    """The path's single gather, pipelined behind the solve (SURVEY.md §8e "pipeline the gather in chunks"):
    the rank's shard is cut into n_chunks pieces; piece c is solved by `launch_chunk(c0, c1, stream)` (which must
    enqueue the kernel for local systems [c0, c1) on `stream`, writing xyz_local[c0:c1]) and its result block is
    gathered onto `dst` while piece c + 1 is being solved.  Every rank must own the same number of systems
    (n_sys divisible by the world size: the weak-scaling layout of bench.py); otherwise use gather_path().
    Works on CUDA tensors with the nccl (= RCCL) backend — the solves run on a side stream, the collectives on the
    current stream behind an event — and on CPU tensors / gloo (launch_chunk then computes synchronously and
    `stream` is None), which is how the logic is tested without a GPU.
    out: the root's receive buffer from gather_buffer() (reused across calls); None = allocate here.
    Returns the [n_sys, 3] path on dst (None elsewhere)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n_local = xyz_local.shape[0]
    if n_local * world != n_sys:
        raise ValueError("solve_and_gather_chunked needs equal shards; use gather_path for ragged ones")
    on_gpu = xyz_local.is_cuda and dist.get_backend(group) != "gloo"
    full = None
    if rank == dst:
        full = out if out is not None else torch.empty((world, n_local, 3), dtype=xyz_local.dtype,
                                                       device=xyz_local.device if on_gpu else "cpu")
        if tuple(full.shape) != (world, n_local, 3):
            raise ValueError("out must be the [world, n_local, 3] buffer of gather_buffer()")
    pieces = chunk_ranges(n_local, n_chunks)
    works = []
    side = torch.cuda.Stream(device=xyz_local.device) if xyz_local.is_cuda else None
    if side is not None:
        side.wait_stream(torch.cuda.current_stream(xyz_local.device))
    for c0, c1 in pieces:
        if side is not None:
            launch_chunk(c0, c1, side.cuda_stream)
            ev = torch.cuda.Event()
            ev.record(side)
            torch.cuda.current_stream(xyz_local.device).wait_event(ev)  # the collective waits for THIS piece only
        else:
            launch_chunk(c0, c1, None)
        send = xyz_local[c0:c1]
        if not on_gpu and send.is_cuda:  # one-GPU rehearsal over gloo: stage through the host
            send = send.cpu()
        if rank == dst:
            recv = [full[r, c0:c1] for r in range(world)]  # contiguous row blocks: results land in place
            works.append(dist.gather(send, recv, dst=dst, group=group, async_op=True))
        else:
            works.append(dist.gather(send, None, dst=dst, group=group, async_op=True))
    for w in works:
        w.wait()
    return full.view(world * n_local, 3) if rank == dst else None

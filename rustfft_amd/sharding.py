"""Batch-dimension sharding across the GPUs of one node.

A batched `Fft::process` call is `batch` independent length-N transforms stored back to back (the chunk loop of
src/array_utils.rs:164-169), so the path shards with NO data-path collective: rank g owns the contiguous rows
[g * ceil(batch / G), min(batch, (g + 1) * ceil(batch / G))) with its own replica of the (tiny) plan tables.
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests) is used only at the
edges: a barrier around the timed region, MAX-reduction of the elapsed time, SUM-reduction of checksums.
"""
import math


def shard_rows(batch, world, rank):
    """Contiguous row range [lo, hi) of `rank` out of `world` for a batch of `batch` transforms."""
    per = math.ceil(batch / world) if world > 0 else batch
    lo = min(batch, rank * per)
    hi = min(batch, (rank + 1) * per)
    return lo, hi


def all_shards(batch, world):
    return [shard_rows(batch, world, r) for r in range(world)]


def reduce_max(value, dist=None, device="cpu"):
    """MAX over ranks of a python float (the timing rule of bench.py)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch

    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def reduce_sum(value, dist=None, device="cpu"):
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch

    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


# ---- the edges: a batch that originates on ONE rank -------------------------------------------------------------------------
# grouped send / recv (ncclGroupStart .. ncclSend / ncclRecv .. ncclGroupEnd over xGMI on the GPU box =
# torch.distributed.batch_isend_irecv; the same calls run over gloo in the CPU tests).  Root -> rank g message: that rank's
# contiguous rows; bounded by the root's 7 xGMI links (~153 GB/s each), i.e. far slower than the transforms themselves, which
# is why these edges are timed separately from the compute path and why the sharded benchmark keeps the shards resident.
_MAX_MSG_ELEMS = 1 << 27  # complex elements per point-to-point operation (1 GiB of Complex<f32>): bounds staging inside the backend


def _as_real(t):
    import torch

    return torch.view_as_real(t) if t.is_complex() else t


def _p2p_chunks(buf, lo_elem, hi_elem):
    for a in range(lo_elem, hi_elem, _MAX_MSG_ELEMS):
        yield _as_real(buf[a:min(hi_elem, a + _MAX_MSG_ELEMS)])


def scatter_rows(full, n, batch, dist, root=0, device=None, dtype=None):
    """Rows [lo_g, hi_g) of the root's `full` buffer (batch * n complex elements, torch tensor; None on the other ranks) ->
    a fresh local tensor on every rank g.  Returns the local shard."""
    import torch

    world, rank = dist.get_world_size(), dist.get_rank()
    lo, hi = shard_rows(batch, world, rank)
    if rank == root:
        device, dtype = full.device, full.dtype
    elif device is None or dtype is None:
        # torch.empty(dtype=None, device=None) would silently make a CPU float32 buffer of half the bytes the root sends
        raise ValueError("scatter_rows: ranks other than the root must pass the shard's device and dtype")
    local = torch.empty((hi - lo) * n, dtype=dtype, device=device)
    ops = []
    if rank == root:
        for g in range(world):
            glo, ghi = shard_rows(batch, world, g)
            if g == root:
                local.copy_(full[glo * n:ghi * n])
            else:
                ops += [dist.P2POp(dist.isend, c, g) for c in _p2p_chunks(full, glo * n, ghi * n)]
    else:
        ops += [dist.P2POp(dist.irecv, c, root) for c in _p2p_chunks(local, 0, (hi - lo) * n)]
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return local


def gather_rows(local, full, n, batch, dist, root=0):
    """The inverse edge: every rank's transformed rows back into the root's `full` buffer (None elsewhere)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    lo, hi = shard_rows(batch, world, rank)
    ops = []
    if rank == root:
        for g in range(world):
            glo, ghi = shard_rows(batch, world, g)
            if g == root:
                full[glo * n:ghi * n].copy_(local)
            else:
                ops += [dist.P2POp(dist.irecv, c, g) for c in _p2p_chunks(full, glo * n, ghi * n)]
    else:
        ops += [dist.P2POp(dist.isend, c, root) for c in _p2p_chunks(local, 0, (hi - lo) * n)]
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return full


def process_from_root(fft, full, n, batch, dist, root=0, device=None, dtype=None, sync=None):
    """A batched `Fft::process` whose buffer lives on one GPU, run on all of them: scatter rows, transform shards (no
    collective), gather rows.  Returns (full, {"scatter_s", "compute_s", "gather_s"}); `sync` is called before each clock
    read (torch.cuda.synchronize on the GPU box)."""
    import time

    sync = sync or (lambda: None)
    sync()
    t0 = time.perf_counter()
    local = scatter_rows(full, n, batch, dist, root, device, dtype)
    sync()
    t1 = time.perf_counter()
    if local.numel():
        fft.process(local)
    sync()
    t2 = time.perf_counter()
    gather_rows(local, full, n, batch, dist, root)
    sync()
    t3 = time.perf_counter()
    return full, {"scatter_s": t1 - t0, "compute_s": t2 - t1, "gather_s": t3 - t2}


def process_sharded(fft, local_rows_buffer):
    """Transform this rank's rows in place (no communication)."""
    fft.process(local_rows_buffer)
    return local_rows_buffer

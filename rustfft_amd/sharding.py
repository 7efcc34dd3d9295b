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


def process_sharded(fft, local_rows_buffer):
    """Transform this rank's rows in place (no communication)."""
    fft.process(local_rows_buffer)
    return local_rows_buffer

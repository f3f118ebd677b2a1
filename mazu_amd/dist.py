"""Multi-GPU plumbing for the k2u path: one process per GPU, index replicated, queries sharded,
and ONE collective -- the final all-reduce of the {n_hit, n_miss, n_skipped} tallies (RCCL over
xGMI when the backend is "nccl"; "gloo" in the CPU tests).  No data-path collective exists."""
import os

import torch
import torch.distributed as dist


def init(backend=None, device=None, force=False):
    """initialise torch.distributed from the torchrun environment; returns (rank, world).  A single process needs no group
    (`force` makes one anyway: the one-rank RCCL check of the GPU tests)"""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend or ("nccl" if torch.cuda.is_available() else "gloo"), **kw)
    return rank, world


def shard_bounds(n, rank, world):
    """contiguous shard [lo, hi) of n items for `rank` (sizes differ by at most one)"""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_reads(read_offsets, rank, world):
    """reads are split on read boundaries: returns (first_read, last_read_exclusive)"""
    return shard_bounds(len(read_offsets) - 1, rank, world)


def _active():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def _reduce(t, op):
    """all_reduce that also works when the backend is gloo and the tensor lives in HBM (CPU rehearsals)"""
    if dist.get_backend() == "gloo" and t.is_cuda:
        c = t.cpu()
        dist.all_reduce(c, op=op)
        t.copy_(c)
    else:
        dist.all_reduce(t, op=op)
    return t


def allreduce_counts(counts):
    """sum a 3-element int64 tensor {n_hit, n_miss, n_skipped} over all ranks (in place)"""
    if _active():
        _reduce(counts, dist.ReduceOp.SUM)
    return counts


def max_over_ranks(seconds, device="cpu"):
    if _active():
        t = torch.tensor([seconds], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
        _reduce(t, dist.ReduceOp.MAX)
        return float(t.item())
    return seconds


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()

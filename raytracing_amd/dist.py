"""Multi-GPU partition of the path: samples-per-pixel are split across ranks (one process per GPU), every rank
renders every pixel for its own sample-index range (disjoint counter-RNG streams via rt_params.spp_offset), and
the fp32 radiance-sum accumulators are summed onto rank 0 with ONE collective — torch.distributed `reduce`,
which is RCCL over xGMI with backend "nccl" on ROCm (and gloo in the CPU tests).

The reference has no distributed path (it is shared-memory rayon, main.rs:64-67); this is the exchange step
north_star adds. No other collective is used: the scene (a few hundred KiB at most) is replicated by construction.
"""
from __future__ import annotations

from typing import Callable, Tuple


def partition_spp(total_spp: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous split of sample indices [0, total_spp) -> (first index, count) for `rank`.
    The remainder goes to the lowest ranks; a rank may get 0 samples when total_spp < world_size."""
    if world_size <= 0 or not (0 <= rank < world_size) or total_spp < 0:
        raise ValueError("bad partition arguments")
    base, rem = divmod(total_spp, world_size)
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


def weak_range(spp_per_rank: int, rank: int) -> Tuple[int, int]:
    """Weak scaling (bench.py): every rank renders spp_per_rank samples; indices stay disjoint."""
    return rank * spp_per_rank, spp_per_rank


def render_and_reduce(render_sum: Callable[[int, int], "object"], first: int, count: int, dist=None, dst: int = 0):
    """render_sum(first, count) -> tensor of per-pixel radiance SUMS over samples [first, first+count) on this
    rank's device. Returns the tensor after the sum-reduce onto `dst` (valid on `dst` only)."""
    acc = render_sum(first, count)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(acc, dst=dst, op=dist.ReduceOp.SUM)
        if getattr(acc, "is_cuda", False):
            # the library renders on its own HIP stream, which is not ordered against torch's NCCL stream: wait
            # for the collective on the host before the caller can render into (or read) this buffer again
            import torch
            torch.cuda.synchronize(acc.device)
    return acc

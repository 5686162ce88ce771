"""Multi-GPU plumbing for the embedding path: one process per GPU (torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests).

The path shards by utterance with replicated weights and no data-path collective (SURVEY.md
section 8e).  The only exchange is enrol/verify: gather every rank's fused rows before trial
scoring; scoring can then shard by trial range.  Nothing here does arithmetic.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(n: int, rank: Optional[int] = None, world_size: Optional[int] = None) -> Tuple[int, int]:
    """Contiguous block partition of n units: the first n % W ranks get one extra unit."""
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    q, rem = divmod(n, world_size)
    lo = rank * q + min(rank, rem)
    return lo, lo + q + (1 if rank < rem else 0)


def gather_rows(local: torch.Tensor, total_rows: int) -> torch.Tensor:
    """All-gather ragged row shards produced by ``shard_range(total_rows)`` into [total_rows, D] on
    every rank (shards are padded to the largest one so a single all_gather_into_tensor suffices:
    one 256 KB-1 MB message per rank, latency-bound on xGMI)."""
    rank, w = world()
    if w == 1:
        return local
    sizes = [shard_range(total_rows, r, w) for r in range(w)]
    mx = max(hi - lo for lo, hi in sizes)
    D = local.shape[1]
    buf = torch.zeros((mx, D), dtype=local.dtype, device=local.device)
    buf[: local.shape[0]] = local
    out = torch.empty((w * mx, D), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, buf)
    return torch.cat([out[r * mx: r * mx + (hi - lo)] for r, (lo, hi) in enumerate(sizes)], 0)


def score_trials_sharded(scorer: Callable[[torch.Tensor, torch.Tensor], torch.Tensor], idx_a: torch.Tensor,
                         idx_b: torch.Tensor) -> torch.Tensor:
    """Each rank scores its block of the trial list with ``scorer(idx_a_blk, idx_b_blk)`` and the
    score vectors are gathered (order preserved)."""
    n = idx_a.numel()
    lo, hi = shard_range(n)
    local = scorer(idx_a[lo:hi].contiguous(), idx_b[lo:hi].contiguous())
    return gather_rows(local.view(-1, 1), n).view(-1)


def allreduce_metrics(values: List[float], device) -> List[float]:
    """Sum of (loss*n, correct, n)-style scalars over ranks."""
    t = torch.tensor(values, dtype=torch.float64, device=device)
    if world()[1] > 1:
        dist.all_reduce(t)
    return t.tolist()


def allreduce_grads(params, world_size: Optional[int] = None) -> int:
    """Data-parallel gradient exchange: ONE flat all-reduce (sum) of every ``p.grad`` divided by the world size
    (backend nccl = RCCL on the GPU box, gloo in the CPU tests).  One bucket on purpose: the trainable sets here
    are 3.4 MB (fusion head), 25 MB (speech encoder), 144 MB (lip-clip model) -- on the point-to-point xGMI
    ring the largest takes ~2 ms beside a 60 ms step, so neither bucketing nor overlap with backward pays.
    Returns the number of elements reduced (0 when not distributed)."""
    if not (dist.is_available() and dist.is_initialized()):
        return 0
    w = world_size or dist.get_world_size()
    grads = [p.grad for p in params if p.grad is not None]
    if w == 1 or not grads:
        return 0
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat)
    flat /= w
    o = 0
    for g in grads:
        g.copy_(flat[o:o + g.numel()].view_as(g))
        o += g.numel()
    return o

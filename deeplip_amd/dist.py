"""Multi-GPU plumbing for the embedding path: one process per GPU (torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests).

The path shards by utterance with replicated weights and no data-path collective (SURVEY.md
section 8e).  The only exchange is enrol/verify: gather every rank's fused rows before trial
scoring; scoring can then shard by trial range.  Nothing here does arithmetic.
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def active() -> bool:
    """A process group exists: collectives are issued (also in a one-rank job)."""
    return dist.is_available() and dist.is_initialized()


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
    if not active():
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
    if active():
        dist.all_reduce(t)
    return t.tolist()


def init_from_env(device: Optional[torch.device] = None) -> Tuple[int, int]:
    """Join the job torch.distributed.run started (RANK / WORLD_SIZE / MASTER_* in the environment): backend nccl
    (= RCCL over xGMI) bound to ``device`` for a GPU rank, gloo for a CPU rank.  No-op for a plain single process; a ONE-rank
    job (torch.distributed.run --nproc-per-node 1) does join, and every collective below then really goes through the
    backend -- how a single-GPU box exercises the RCCL path (tests/test_rccl_gpu.py).  Returns (rank, world size)."""
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if device is not None and device.type == "cuda":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")
    return world()


def broadcast_params(params, src: int = 0) -> None:
    """Replicas start identical (what nn.DataParallel's per-step broadcast guarantees: train_audio.py:83).  The write goes
    through ``p.data`` (no version-counter bump), so the packed-weight cache and any recorded step plan are invalidated
    explicitly: a forward run before the broadcast must not leave stale packs on the non-source ranks."""
    if active():
        for p in params:
            dist.broadcast(p.data, src)
        from . import holders
        holders.invalidate_packs()


def allreduce_grads(params, world_size: Optional[int] = None) -> int:
    """Data-parallel gradient exchange in ONE flat all-reduce (sum / world size) -- for small trainable sets (the
    3.4 MB fusion head: latency-bound on xGMI, a single message is the right shape).  The flat buffer is built from
    the FIXED parameter list, a parameter without a gradient on this rank contributing zeros, so every rank reduces
    the same layout whatever its batch touched; such a parameter then receives the (mean) gradient of the others.
    Larger models use GradBuckets (overlapped with backward).  Returns the number of elements reduced (0 when not
    distributed)."""
    if not (dist.is_available() and dist.is_initialized()):
        return 0
    w = world_size or dist.get_world_size()
    params = [p for p in params if p.requires_grad]
    if not params:
        return 0
    has = torch.tensor([0.0 if p.grad is None else 1.0 for p in params], device=params[0].device)
    dist.all_reduce(has)                                       # which parameters got a gradient on ANY rank
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
    dist.all_reduce(flat)
    flat /= w
    o = n = 0
    for p, h in zip(params, has.tolist()):
        k = p.numel()
        if h > 0:
            if p.grad is None:
                p.grad = flat[o:o + k].view_as(p).clone()
            else:
                p.grad.copy_(flat[o:o + k].view_as(p))
            n += k
        o += k
    return n


_CAPTURE_GROUP = [None, None]      # the group, and the default group it was made beside (a re-initialised job makes a new one)


def capture_group(device: Optional[torch.device] = None):
    """The process group whose collectives are CAPTURED into recorded training steps (deeplip_amd.train_plan) -- a second
    communicator over the same ranks that never carries an eager collective after its one warm-up.  Why: every process group has
    a watchdog thread polling the end events of its eager collectives; when a recorded step captures a collective of that group,
    the group's own stream enters the capture and HIP refuses the watchdog's query of an earlier event on it -- the job aborts
    ("operation not permitted on an event last recorded in a capturing stream": round 5, once in seven runs).  A group that only
    ever sees captured collectives (which are never enqueued to a watchdog) has nothing to poll; the default group, which the eager
    steps, broadcasts and metric reductions use, is never captured.  Collective on every rank (dist.new_group); None outside a job."""
    if not active():
        return None
    default = dist.distributed_c10d._get_default_group()
    if _CAPTURE_GROUP[0] is None or _CAPTURE_GROUP[1] is not default:
        g = dist.new_group(backend=dist.get_backend())
        t = torch.zeros(1, device=device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu"))
        dist.all_reduce(t, group=g)            # communicator set-up happens HERE, eagerly, long before any capture
        if t.is_cuda:
            torch.cuda.synchronize(t.device)
        _CAPTURE_GROUP[0], _CAPTURE_GROUP[1] = g, default
    return _CAPTURE_GROUP[0]


class GradBuckets:
    """Bucketed gradient all-reduce overlapped with backward (what replaces nn.DataParallel's
    gather of gradients, train_audio.py:83 / train_video.py:206-207, on one process per GPU over RCCL).

    The parameters' ``.grad`` tensors are VIEWS into a few flat buckets (filled in reverse registration order, the
    order backward produces them), so there is no concatenation and no copy-back: a post-accumulate hook counts a
    bucket's gradients as they land, and the moment a bucket is complete its all-reduce is launched asynchronously
    -- on RCCL's own stream, beside the rest of the backward kernels.  ``finish()`` launches whatever is left
    (parameters a batch did not touch keep their zeros: every rank reduces the same layout), waits, and divides by
    the world size.  Use ``optimizer.zero_grad(set_to_none=False)`` (or ``buckets.zero()``) so the views survive.
    xGMI is point-to-point (7 links x ~153 GB/s): a ring all-reduce is bound by one link, so buckets are sized for
    a few hundred microseconds each (default 32 MB ~ 0.5 ms at 8 ranks) -- large enough to be bandwidth- rather than
    latency-bound, small enough that the last one does not stick out behind backward.

    Stream rule: a bucket's all-reduce waits for the stream that is current when its last gradient lands.  Gradients written on
    OTHER streams (deeplip_amd.video.BRANCH_STREAMS: forked branches of the train graph) are not covered -- keep the branches on
    one stream with GradBuckets (train_video.py does).

    One difference from a single-GPU run to know about: a parameter that NO rank's batch reached keeps an all-zero gradient here
    (its ``.grad`` is a bucket view, never None), so an optimizer with momentum or weight decay still updates it, where the
    single-process run -- ``.grad is None`` -- skips it.  The shipped models touch every trainable parameter in every step, so
    the two coincide; a model with truly conditional parameters should exclude them from weight decay."""

    def __init__(self, params, bucket_bytes: int = 32 << 20):
        self.params = [p for p in params if p.requires_grad]
        self.world = world()[1]
        self.active = active()     # a process group exists (a one-rank job included): the collectives are issued
        # all-reduces issued while the stream is being captured (a recorded step) go to the capture group (see capture_group)
        self.capture_grp = capture_group(self.params[0].device) if (self.active and self.params and self.params[0].is_cuda) else None
        self.buckets: List[torch.Tensor] = []
        self._slot = {}            # id(param) -> bucket index
        self._pending: List[int] = []
        self._count: List[int] = []
        self._works: list = []
        self._launched: List[bool] = []
        cur, cur_n = [], 0
        groups = []
        for p in reversed(self.params):
            if cur and (cur_n + p.numel()) * p.element_size() > bucket_bytes:
                groups.append(cur); cur, cur_n = [], 0
            cur.append(p); cur_n += p.numel()
        if cur:
            groups.append(cur)
        for bi, g in enumerate(groups):
            flat = torch.zeros(sum(p.numel() for p in g), dtype=g[0].dtype, device=g[0].device)
            o = 0
            for p in g:
                p.grad = flat[o:o + p.numel()].view_as(p)
                o += p.numel()
                self._slot[id(p)] = bi
                p.register_post_accumulate_grad_hook(self._hook)
            self.buckets.append(flat)
            self._count.append(len(g))
        self._reset()

    def _reset(self):
        self._pending = list(self._count)
        self._launched = [False] * len(self.buckets)
        self._works = []
        self._next = 0             # buckets [0, _next) have been launched: launches go out in INDEX ORDER only

    def _launch_upto(self):
        """Launch, in bucket-index order, every bucket that is complete and whose predecessors have all been launched.
        Collectives are matched across ranks by issue order, not by tensor: if one rank's backward completes buckets in a
        different order, or never completes one (a parameter its batch did not reach), launching "whatever is ready" would
        pair different buckets -- different sizes -- on different ranks and hang or mis-reduce on RCCL.  So a complete bucket
        waits for every lower-numbered one (DDP's rule), and finish() flushes the remainder in the same order."""
        while self._next < len(self.buckets) and self._pending[self._next] == 0:
            self._launch(self._next)
            self._next += 1

    def _launch(self, bi: int):
        if self._launched[bi]:
            return
        self._launched[bi] = True
        if self.active:
            b = self.buckets[bi]
            grp = self.capture_grp if (self.capture_grp is not None and b.is_cuda and torch.cuda.is_current_stream_capturing()) else None
            self._works.append(dist.all_reduce(b, async_op=True, group=grp))

    def _hook(self, p):
        bi = self._slot[id(p)]
        self._pending[bi] -= 1
        if self._pending[bi] < 0:
            raise RuntimeError("GradBuckets: a parameter received a second gradient before finish() -- a second backward() "
                               "(gradient accumulation) would add into a bucket whose all-reduce may already be in flight; "
                               "call finish() after every backward()")
        self._launch_upto()

    def zero(self):
        for b in self.buckets:
            b.zero_()

    def finish(self) -> int:
        """Call after backward: reduce the buckets that are not on their way yet, wait for all, average.  Returns the
        number of elements reduced (0 when not distributed)."""
        for p in self.params:       # an optimizer.zero_grad() with set_to_none=True would have detached the views
            b = self.buckets[self._slot[id(p)]]
            if p.grad is None or not (b.data_ptr() <= p.grad.data_ptr() < b.data_ptr() + b.numel() * b.element_size()):
                raise RuntimeError("GradBuckets: a parameter's .grad no longer lives in its bucket; clear gradients with "
                                   "optimizer.zero_grad(set_to_none=False) or GradBuckets.zero()")
        for bi in range(self._next, len(self.buckets)):       # the rest, still in index order on every rank
            self._launch(bi)
        for w in self._works:
            w.wait()
        n = 0
        if self.active:
            for b in self.buckets:
                if self.world > 1:
                    b.div_(self.world)
                n += b.numel()
        self._reset()
        return n

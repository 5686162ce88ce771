"""Extraction with the host-to-device copies BEHIND the compute.

The reference moves every utterance to the GPU inside its test loop and waits for it (train_fusion.py:338, 346-348:
``.to(device)`` per utterance and per clip, then the forward).  Batched and replayed from a step plan the forward of 64
pairs takes ~4 ms, and a synchronous copy of the batch in front of it costs a third of the throughput although the bytes
fit the link four times over.  So the copies get their own stream and the inputs two homes:

    copy stream :  H2D(batch 0 -> set 0)  H2D(batch 1 -> set 1)            H2D(batch 2 -> set 0)   ...
    run  stream :                         plan[0].run()  rows -> table     plan[1].run()  rows -> table   ...

``depth`` input sets, one recorded StepPlan per set (a plan addresses fixed buffers; 1.7 GB of arena each at B = 64 -- of
288 GB), two events per set: READY (recorded on the copy stream behind the set's copies, awaited by the run stream) and FREE
(recorded on the run stream behind the plan's replay, awaited by the HOST before the set is overwritten: the host stays at most
`depth` batches ahead, see run()).  Per batch the host enqueues a few async copies and one plan replay.

Inputs are whatever the step function takes: normalised float clips [B,1,T,88,88], or the uint8 frames a loader holds
([B,T,3,H,W] RGB / [B,T,H,W] gray), which the lip-clip encoder's pre-pass normalises on the fly (a quarter of the bytes over PCIe).
Host tensors must be pinned for the copies to be asynchronous (``pin()`` helps).
"""
from __future__ import annotations

from typing import Callable, Iterable, List, Optional, Sequence, Tuple

import torch

from . import arith
from ._lib import DeepLipHipError, check_range
from .plan import StalePlanError, StepPlan

Tensor = torch.Tensor


def pin(t: Tensor) -> Tensor:
    return t if t.is_pinned() else t.pin_memory()


class ExtractPipeline:
    """``fn(*inputs) -> rows [B, D]`` recorded once per input set; ``run(batches, table)`` streams host batches through."""

    def __init__(self, fn: Callable, *example_inputs: Tensor, depth: int = 2, device: Optional[torch.device] = None, fallback="auto"):
        """``fallback``: what happens to a batch whose replay left the range of the f16x3 arithmetic.  True: that batch -- still in its
        input set -- is computed again eagerly on the exact f32 pack of the same models and its rows overwritten (counted in
        deeplip_amd.arith.STATS); False: DeepLipRangeError at the next submit / finish; "auto": True under arith mode ``auto``."""
        if depth < 2:
            raise ValueError("ExtractPipeline: depth >= 2 (one set being filled while another is being read)")
        for i, t in enumerate(example_inputs):
            if not isinstance(t, Tensor):
                raise DeepLipHipError(f"ExtractPipeline: example input {i} must be a tensor (shape / dtype template)")
        self.device = device or next((t.device for t in example_inputs if t.is_cuda), torch.device("cuda", torch.cuda.current_device()))
        self.depth = depth
        self.fn = fn
        self.fallback = arith.fallback_enabled() if fallback == "auto" else bool(fallback)
        self.reruns = 0
        self.rerecorded = 0
        self.run_stream = torch.cuda.Stream(device=self.device)
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.sets: List[Tuple[Tensor, ...]] = []
        self.plans: List[StepPlan] = []
        caller = torch.cuda.current_stream(self.device)
        self.run_stream.wait_stream(caller)
        with torch.cuda.stream(self.run_stream):
            for _ in range(depth):
                ins = tuple(torch.zeros(t.shape, dtype=t.dtype, device=self.device) for t in example_inputs)
                for dst, src in zip(ins, example_inputs):      # a recorded pass must see representative values (range guard)
                    dst.copy_(src, non_blocking=True)
                self.sets.append(ins)
                self.plans.append(StepPlan(fn, *ins, stream=self.run_stream, private_status=True))
        self.ready = [torch.cuda.Event() for _ in range(depth)]
        self.free = [torch.cuda.Event() for _ in range(depth)]
        self._pending: list = [None] * depth # (tables, row, rows) of the replay in flight on set k: settled when the set is recycled / at finish()
        # (every report of a replay is settled here, batch by batch: _settle; the plans' status blocks are private)
        self._fifo: list = []                # sets with an unsettled replay, oldest first (wait_next)
        self._rows = [0] * depth             # rows of the batch a set holds (prefetch -> replay)
        self._replayed = [False] * depth     # set k has a replay in flight (or finished) whose FREE event must be awaited before refilling it
        self._next = 0                       # batches submitted so far: batch i uses input set i % depth
        self.batch = int(example_inputs[0].shape[0])
        self.launches = self.plans[0].launches
        self.run_stream.synchronize()
        for p in self.plans:
            p.take_range_error()             # what the RECORDING passes reported about the example values: not an output of this pipeline

    def submit(self, hb: Sequence[Tensor], tables: Sequence[Tensor], row: int) -> int:
        """One batch: ``hb`` (pinned host tensors shaped like the recorded inputs; any of them may be SHORT in its leading
        dimension) -> copies on the copy stream, one plan replay on the run stream, the first ``rows = hb[0].shape[0]`` output rows
        into ``tables[j][row : row + rows]``.  Asynchronous; returns ``rows``.  (= prefetch + replay.)"""
        return self.replay(self.prefetch(hb), tables, row)

    def prefetch(self, hb: Sequence[Tensor]) -> int:
        """The copy half of ``submit``: the batch goes into the next input set (copy stream; the host first waits for the replay
        that last read the set).  Returns the set's index for ``replay``.  A caller that wants the replays in ITS order -- the
        fusion trainer runs encoders(i), head(i), encoders(i + 1) back to back on one stream -- prefetches ahead and replays late."""
        k = self._next % self.depth
        self._next += 1
        ins = self.sets[k]
        rows = int(hb[0].shape[0])
        if rows > self.batch or len(hb) != len(ins) or any(int(h.shape[0]) > int(d.shape[0]) for h, d in zip(hb, ins)):
            raise ValueError("ExtractPipeline: batch does not match the recorded inputs")
        if k in self._fifo:
            self._fifo.remove(k)             # recycled before anybody waited for it: settled right below
        if self._replayed[k]:
            # (Also across calls: a second run() without finish() in between refills sets the previous call's last replays may
            # still be reading -- the copy stream is ordered behind nothing but this wait.)
            # Bounded run-ahead: the HOST waits here until the replay that read this set has finished, so it is never more
            # than `depth` batches ahead of the GPU.  Measured (tools/probes/h2d_timeline.py, B = 64, uint8 RGB): with the
            # host free to enqueue all 40 batches at once the replays behind the enqueue burst take 5.2-6.2 ms instead of
            # 4.2 (and the enqueue itself 2.3 ms per batch); throttled, every batch takes 4.23 ms -- the resident rate.
            self.free[k].synchronize()
            self._settle(k)
            self._replayed[k] = False
        with torch.cuda.stream(self.copy_stream):
            for dst, src in zip(ins, hb):
                (dst if src.shape[0] == dst.shape[0] else dst[:src.shape[0]]).copy_(src, non_blocking=True)
            self.ready[k].record(self.copy_stream)
        self._rows[k] = rows
        return k

    def replay(self, k: int, tables: Sequence[Tensor], row: int) -> int:
        """The run half of ``submit`` for the set ``prefetch`` returned: the plan's replay on the run stream behind the set's copies,
        its first rows into ``tables[j][row : ...]``."""
        rows = self._rows[k]
        with torch.cuda.stream(self.run_stream):
            self.run_stream.wait_event(self.ready[k])
            try:
                out = self.plans[k].run(check_reports=False)
            except StalePlanError:
                # the models' packs changed under the plan.  Under the fallback that is expected once: the exact re-run of an
                # out-of-range batch calibrated activation exponents (arith.rerun_exact), so the f16x3 packs were rebuilt -- record
                # this set's plan again on the batch that has just been copied into it, and carry on in the fast arithmetic
                if not self.fallback:
                    raise
                self._rerecord(k)
                out = self.plans[k].run(check_reports=False)
            outs = [out] if isinstance(out, Tensor) else list(out)
            if len(outs) != len(tables):
                raise ValueError(f"ExtractPipeline: the step returns {len(outs)} tensors, {len(tables)} tables given")
            for t, o in zip(tables, outs):
                t[row: row + rows].copy_(o[:rows], non_blocking=True)
            self.free[k].record(self.run_stream)
        self._replayed[k] = True
        self._pending[k] = (list(tables), row, rows)
        self._fifo.append(k)
        return rows

    def _rerecord(self, k: int) -> None:
        old = self.plans[k]
        self.plans[k] = StepPlan(self.fn, *self.sets[k], stream=self.run_stream, private_status=True)
        self.plans[k].take_range_error()     # (the recording passes ran on this very batch; its replay right behind reports for itself)
        self.rerecorded += 1
        try:
            old.close()
        except Exception:   # noqa: BLE001 -- a stale plan's leftovers must not take the run down
            pass

    def wait_next(self) -> bool:
        """Wait for the OLDEST submitted batch that has not been waited for (host wait on its FREE event), settle its range report
        (f32 re-run under the fallback) and return True; False when nothing is pending.  Lets a caller consume batch i while
        batches i + 1 .. are in flight -- the training loop of train_fusion.py: the head's step on batch i runs while the encoders
        work on batch i + 1 and the copies of batch i + 2 are on their way."""
        if not self._fifo:
            return False
        k = self._fifo.pop(0)
        self.free[k].synchronize()
        self._settle(k)
        return True

    def _settle(self, k: int) -> None:
        """Set k's last replay has finished: did it stay inside the f16x3 range?  If not, the batch is still in the set's buffers --
        compute it again on the exact f32 packs (fallback) or raise.  The reference computes everything in fp32
        (train_fusion.py:338-358), so a batch must never come back wrong because the fast arithmetic could not hold it."""
        pend, self._pending[k] = self._pending[k], None
        if pend is None:
            return
        err = self.plans[k].take_range_error()
        if err is None:
            return
        if not self.fallback:
            raise err
        tables, row, rows = pend
        with torch.cuda.stream(self.run_stream):
            out = arith.rerun_exact(self.fn, *self.sets[k], what=f"ExtractPipeline batch (rows {row}..{row + rows - 1})", err=err)
            outs = [out] if isinstance(out, Tensor) else list(out)
            for t, o in zip(tables, outs):
                t[row: row + rows].copy_(o[:rows], non_blocking=True)
        self.run_stream.synchronize()
        self.reruns += 1

    def run(self, batches: Iterable[Sequence[Tensor]], table, row0: int = 0) -> int:
        """Stream ``batches`` (tuples of pinned host tensors shaped like the recorded inputs; any of them may be SHORT in its
        leading dimension -- the last batch of a list) through the plans.  ``table``: one tensor, or one per output of the step
        function; batch i's rows land in ``table[row0 + i*B : ...]``, as many rows as the batch's FIRST tensor has.  ``batches``
        may be a generator: the host prepares batch i+2 while the GPU works on i and i+1.  Asynchronous: returns the number of
        rows enqueued -- call ``finish()`` (or synchronise the device) before reading the tables."""
        tables = [table] if isinstance(table, Tensor) else list(table)
        n = 0
        for hb in batches:
            n += self.submit(hb, tables, row0 + n)
        return n

    def finish(self) -> None:
        self.run_stream.synchronize()
        self._replayed = [False] * self.depth
        self._fifo = []
        for k in range(self.depth):   # an f16x3 overflow of the LAST batches surfaces (or is repaired) here, not one call late
            self._settle(k)
        check_range(sync=False)

    def close(self) -> None:
        for p in self.plans:
            p.close()
        self.plans = []


class BucketedExtract:
    """ExtractPipeline for RAGGED lists: one pipeline (= ``depth`` input sets + recorded plans) per padded batch shape.

    A length-sorted list cut by deeplip_amd.ragged.plan_batches reaches the GPU as batches of a dozen distinct padded shapes
    (the rungs of the length ladder); the lengths themselves travel as an int32 input tensor, so every batch of a rung replays
    the rung's plan.  Pipelines are created on first use from the batch that needs them (a plan is recorded on representative
    values) and kept up to ``max_arena_bytes`` of plan arenas, least recently used first out -- a second pass over the list
    (the bench, an epoch of evaluation) records nothing.  Rows are written in SUBMISSION order; the caller un-sorts once at the
    end (``table[order] = rows``)."""

    def __init__(self, fn: Callable, depth: int = 2, device: Optional[torch.device] = None, max_arena_bytes: int = 96 << 30, fallback="auto"):
        self.fn, self.depth, self.device, self.max_arena_bytes = fn, depth, device, max_arena_bytes
        self.fallback = fallback
        self.pipes: dict = {}        # shape key -> ExtractPipeline, in LRU order (dicts keep insertion order)
        self.recorded = 0            # pipelines recorded so far (tests, the bench's report)

    @staticmethod
    def _key(hb: Sequence[Tensor]):
        return tuple((tuple(t.shape[1:]), t.dtype) for t in hb)

    def _arena_bytes(self) -> int:
        return sum(pl.arena.nbytes() for p in self.pipes.values() for pl in p.plans)

    def pipe_for(self, hb: Sequence[Tensor], batch: int) -> ExtractPipeline:
        key = (batch,) + self._key(hb)
        p = self.pipes.pop(key, None)
        if p is None:
            full = []
            for t in hb:           # a short first batch still records the plan at the full batch size (rows repeated)
                reps = -(-batch // int(t.shape[0]))
                full.append(torch.cat([t] * reps)[:batch] if int(t.shape[0]) != batch else t)
            dev = self.device or torch.device("cuda", torch.cuda.current_device())
            with torch.no_grad():
                p = ExtractPipeline(self.fn, *(t.to(dev) for t in full), depth=self.depth, device=dev, fallback=self.fallback)
            self.recorded += 1
        self.pipes[key] = p        # most recently used last
        while len(self.pipes) > 1 and self._arena_bytes() > self.max_arena_bytes:
            old_key = next(iter(self.pipes))
            old = self.pipes.pop(old_key)
            old.finish()
            old.close()
        return p

    def submit(self, hb: Sequence[Tensor], tables: Sequence[Tensor], row: int, batch: int) -> int:
        return self.pipe_for(hb, batch).submit(hb, tables, row)

    def finish(self) -> None:
        for p in self.pipes.values():
            p.finish()

    @property
    def reruns(self) -> int:
        return sum(p.reruns for p in self.pipes.values())

    def close(self) -> None:
        for p in self.pipes.values():
            p.close()
        self.pipes = {}

"""A whole optimisation step as ONE replayed HIP graph.

The train-mode path issues ~1 500 launches per step of the lip-clip model (forward, backward and Adam of ResNet-18 + MS-TCN,
every one a ``dlip_*`` kernel behind torch.autograd): at ~20 us of Python / autograd / ctypes per launch the HOST needs 33 ms for a
step whose kernels take 28 ms (B = 32; profiles/r3/train_video_kernel_stats_graph.txt) -- the reference hides the same cost behind
cuDNN's fat kernels, an engine of many small launches cannot.  So the step is recorded once and replayed, the way the extraction
path replays its step plan (deeplip_amd.plan):

    zero_grad -> forward -> loss -> backward -> optimizer.step        = one hipGraphLaunch

Recording uses torch.cuda.CUDAGraph (hipGraph capture of the stream every dlip_* launch already goes to, with torch's caching
allocator in capture mode -- autograd's own buffers, the optimizer's kernels and the dropout generator's Philox offsets are
graph-safe there; the inference StepPlan's arena only knows the engine's allocations).  What the recording fixes: shapes, the
parameter / optimizer-state addresses (updated in place), the input buffers (``step`` copies each batch into them).  What must not
be inside: host reads (``float(loss)``: read the returned tensors after the call), host->device copies of Python data (pass clip
lengths as an int32 device tensor), a learning rate held as a Python float (use a tensor ``lr`` -- torch's schedulers then update it
in place -- and ``capturable=True`` for Adam).

Every step -- eager or replayed -- runs on the plan's own stream; do not keep tensors of an autograd graph built on ANOTHER stream
alive across ``step`` (their AccumulateGrad nodes remember that stream and break the capture).

The first ``eager_steps`` calls run the step eagerly: they ARE optimisation steps (they create the optimizer state, size the
stream-K workspace, set kernel attributes), so every call of ``step`` is exactly one step whichever way it ran.
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence

import torch

from ._lib import DeepLipHipError, check_range

Tensor = torch.Tensor


class TrainStepGraph:
    """``fn(*inputs) -> tensor or tuple of tensors`` performs one complete optimisation step on device tensors ``inputs``.

    ``verify`` (first contact must not depend on luck): the call that would record the step first runs it EAGERLY from a snapshot
    of ``state()`` (every tensor the step mutates: parameters, buffers, optimizer state, gradient buckets) + the generators, keeps
    its outputs and ``witness()`` (tensors that summarise what the step did -- the gradients), restores the snapshot, records,
    replays once and compares.  A capture that fails, or a replay whose outputs / witnesses differ from the eager step's beyond
    ``verify_rtol``, restores the snapshot once more, runs the step eagerly and keeps doing so (``mode`` says "eager: <reason>",
    logged); in a torch.distributed job the ranks agree on the verdict (MIN all-reduce) so that either all replay or none.
    ``verify="auto"``: on when the job has more than one rank (a captured RCCL collective has never met a second rank on this
    pool) or DLIP_VERIFY_GRAPH=1."""

    def __init__(self, fn: Callable, eager_steps: int = 2, device: Optional[torch.device] = None, branch_streams: bool = True,
                 verify="auto", state: Optional[Callable] = None, witness: Optional[Callable] = None, verify_rtol: float = 1e-6):
        if eager_steps < 1:
            raise ValueError("TrainStepGraph: at least one eager step (optimizer state and workspaces must exist before recording)")
        self.fn = fn
        self.branch_streams = branch_streams    # independent branches of the lip-clip model's train graph on side streams (video.BRANCH_STREAMS)
        self.eager_steps = eager_steps
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.stream = torch.cuda.Stream(device=self.device)
        self.calls = 0
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.static: Sequence[Tensor] = ()
        self.outputs = None
        self.state, self.witness, self.verify_rtol = state, witness, float(verify_rtol)
        if verify == "auto":
            import os
            import torch.distributed as _dist
            multi = _dist.is_available() and _dist.is_initialized() and _dist.get_world_size() > 1
            verify = (multi or os.environ.get("DLIP_VERIFY_GRAPH") == "1") and state is not None
        if verify and state is None:
            raise ValueError("TrainStepGraph(verify=True) needs state=: the tensors the step mutates")
        self.verify = bool(verify)
        self.eager_only: Optional[str] = None   # why this plan gave up on the recorded step
        self.verified: Optional[dict] = None    # what the first replay's comparison with the eager step measured

    @property
    def mode(self) -> str:
        """What the NEXT call of step() does: "eager (warm-up)", "graph", or "eager: <why the recorded step was dropped>"."""
        if self.eager_only is not None:
            return "eager: " + self.eager_only
        return "graph" if self.graph is not None else "eager (warm-up)"

    def _check(self, inputs):
        for i, t in enumerate(inputs):
            if not (isinstance(t, Tensor) and t.is_cuda):
                raise DeepLipHipError(f"TrainStepGraph: input {i} must be a device tensor (no host data inside a recorded step)")

    def step(self, *inputs: Tensor):
        """One optimisation step on this batch.  Returns fn's outputs: tensors valid until the next call (the replayed graph
        rewrites them in place)."""
        self._check(inputs)
        cur = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(cur)
        from . import video as _video
        prev, _video.BRANCH_STREAMS = _video.BRANCH_STREAMS, self.branch_streams
        try:
            return self._step(inputs, cur)
        finally:
            _video.BRANCH_STREAMS = prev

    # ---- first-contact verification
    @staticmethod
    def _flat(out):
        if isinstance(out, Tensor):
            return [out]
        return [t for t in (out or ()) if isinstance(t, Tensor)]

    def _snapshot(self):
        return ([t.detach().clone() for t in self.state()], torch.cuda.get_rng_state(self.device), torch.get_rng_state())

    def _restore(self, snap):
        with torch.no_grad():
            for t, s in zip(self.state(), snap[0]):
                t.detach().copy_(s)
        torch.cuda.set_rng_state(snap[1], self.device)
        torch.set_rng_state(snap[2])

    def _agree(self, ok: bool) -> bool:
        """The ranks' common verdict: every one of them saw ``ok`` (eager collective on the default group, outside any capture)."""
        import torch.distributed as _dist
        if not (_dist.is_available() and _dist.is_initialized()):
            return ok
        t = torch.tensor([1.0 if ok else 0.0], device=self.device)
        _dist.all_reduce(t, op=_dist.ReduceOp.MIN)
        return bool(t.item() > 0.5)

    def _compare(self, ref, got):
        worst = 0.0
        for a, b in zip(got, ref):
            if a.shape != b.shape:
                return False, float("inf")
            if a.numel() == 0:
                continue
            a64, b64 = a.detach().double(), b.detach().double()
            scale = float(b64.abs().max())
            err = float((a64 - b64).abs().max())
            if not (err == err):              # NaN on one side only is a mismatch; NaN on both compares below
                if not bool(torch.equal(torch.isnan(a64), torch.isnan(b64))):
                    return False, float("nan")
                continue
            worst = max(worst, err / max(scale, 1e-30))
        return worst <= self.verify_rtol, worst

    def _capture(self):
        g = torch.cuda.CUDAGraph()
        # Inside a torch.distributed job the process group's watchdog THREAD polls the events of earlier collectives;
        # under the default ("global") capture mode such a call from another thread aborts the capture ("operation not
        # permitted when stream is capturing" -- seen in one run of two).  "thread_local" confines the check to this
        # thread; the launches of the autograd worker threads are captured either way (they go to the capturing stream).
        import torch.distributed as _dist
        in_job = _dist.is_available() and _dist.is_initialized()
        mode = "thread_local" if in_job else "global"
        if in_job:
            # ... and a watchdog must hold NO work of a stream that is about to capture: HIP refuses its query of an EARLIER
            # collective's end event (recorded on that stream before the capture, long complete) while the stream is capturing --
            # "operation not permitted on an event last recorded in a capturing stream", the job aborts (round 5, once in seven
            # runs).  Since round 6 the collectives of a recorded step go to a process group OF THEIR OWN (deeplip_amd.dist.
            # capture_group: its one warm-up collective was retired long ago, and captured collectives are never enqueued to a
            # watchdog), so the default group's stream -- the one eager steps use -- is never captured at all.  The wait below
            # (a synchronise + WATCHDOG_POLLS polls of the watchdog's interval) stays as the second line for callers that
            # capture collectives of the default group.
            import os
            import time
            torch.cuda.synchronize(self.device)
            poll_ms = float(os.environ.get("TORCH_NCCL_WATCHDOG_SLEEP_INTERVAL_MS", os.environ.get("DLIP_WATCHDOG_POLL_MS", "100")))
            time.sleep(5 * poll_ms / 1e3)
        with torch.cuda.graph(g, stream=self.stream, capture_error_mode=mode):
            out = self.fn(*self.static)
        return g, out

    def _give_up(self, reason: str):
        import logging
        self.eager_only, self.graph, self.outputs = reason, None, None
        logging.getLogger("deeplip_amd.train_plan").warning("TrainStepGraph: the recorded step is dropped, steps run eagerly from here: %s", reason)

    def _record(self, inputs):
        """The call that records: returns this step's outputs (it IS an optimisation step, whichever way it ran)."""
        self.static = tuple(torch.empty_like(t) for t in inputs)
        for d, s in zip(self.static, inputs):
            d.copy_(s, non_blocking=True)
        self.stream.synchronize()
        snap = ref_out = ref_wit = None
        if self.verify:
            snap = self._snapshot()
            out = self.fn(*self.static)                      # the eager step everybody trusts, from the snapshot
            self.stream.synchronize()
            ref_out = [t.detach().clone() for t in self._flat(out)]
            ref_wit = [t.detach().clone() for t in (self.witness() if self.witness is not None else [])]
            self._restore(snap)
            self.stream.synchronize()
        try:
            g, outputs = self._capture()
            captured, why = True, ""
        except Exception as ex:   # noqa: BLE001 -- whatever the capture died of, the step itself is still runnable eagerly
            if not self.verify:
                raise
            captured, why = False, f"capture failed ({type(ex).__name__}: {str(ex)[:160]})"
        if self.verify and not self._agree(captured):
            self._give_up(why or "capture failed on another rank")
            self._restore(snap)
            return self.fn(*self.static)
        self.graph, self.outputs = g, outputs
        self.graph.replay()
        if self.verify:
            self.stream.synchronize()
            ok_o, err_o = self._compare(ref_out, self._flat(self.outputs))
            ok_w, err_w = self._compare(ref_wit, list(self.witness()) if self.witness is not None else [])
            self.verified = {"outputs_rel_err": err_o, "witness_rel_err": err_w, "rtol": self.verify_rtol, "tensors": len(ref_out) + len(ref_wit)}
            if not self._agree(ok_o and ok_w):
                self._give_up(f"first replay differs from the eager step of the same state (outputs {err_o:.3e}, gradients {err_w:.3e}, "
                              f"bar {self.verify_rtol:.0e}) on this or another rank")
                self._restore(snap)
                return self.fn(*self.static)
        return self.outputs

    def _step(self, inputs, cur):
        with torch.cuda.stream(self.stream):
            if self.calls < self.eager_steps or self.eager_only is not None:
                out = self.fn(*inputs)
            elif self.graph is None:
                out = self._record(inputs)
            else:
                if len(inputs) != len(self.static) or any(a.shape != b.shape or a.dtype != b.dtype for a, b in zip(inputs, self.static)):
                    raise DeepLipHipError("TrainStepGraph: batch shapes / dtypes differ from the recorded step")
                for d, s in zip(self.static, inputs):
                    d.copy_(s, non_blocking=True)
                self.graph.replay()
                out = self.outputs
        cur.wait_stream(self.stream)
        self.calls += 1
        return out

    @property
    def recorded(self) -> bool:
        return self.graph is not None

    def finish(self) -> None:
        """Wait for the last step and surface a range error of the split-fp16 arithmetic (an overflow is an error, not a NaN)."""
        self.stream.synchronize()
        check_range(sync=False)


def step_state(modules=(), optimizers=(), buckets=None) -> Callable:
    """``state`` for TrainStepGraph(verify=...): every tensor an optimisation step of these modules mutates -- parameters, buffers
    (BatchNorm running statistics, num_batches_tracked), the optimizers' state tensors (momentum, Adam moments, step counters) and
    the gradient buckets.  Evaluated lazily (optimizer state exists only after the first eager step)."""
    def state():
        out = []
        for m in modules:
            out += list(m.parameters()) + list(m.buffers())
        for o in optimizers:
            for grp in o.param_groups:
                for p in grp["params"]:
                    st = o.state.get(p, {})
                    out += [v for _, v in sorted(st.items()) if isinstance(v, Tensor)]
                out += [v for _, v in sorted(grp.items()) if isinstance(v, Tensor)]      # a tensor learning rate
        if buckets is not None:
            out += list(buckets.buckets)
        return out
    return state


def grad_witness(modules=(), buckets=None) -> Callable:
    """``witness`` for TrainStepGraph(verify=...): the gradients the step left behind (the buckets, or every parameter's .grad)."""
    def witness():
        if buckets is not None:
            return list(buckets.buckets)
        return [p.grad for m in modules for p in m.parameters() if p.grad is not None]
    return witness


class ShapeKeyedSteps:
    """One TrainStepGraph per batch SHAPE (and per ``key`` the caller adds -- a criterion's margin, anything a recorded step
    bakes in): the loaders of the reference hand over batches padded to their longest clip (pad_packed_collate,
    models/video_models/dataset.py:123-139) or cropped to a random length (models/audio_models/datasets.py:112-115), so a
    training run meets a handful of shapes; each one is recorded the second time it is met (its first step runs eagerly) and
    replayed from then on.  ``max_plans`` bounds the graphs kept (least recently used out: a graph owns a memory pool)."""

    def __init__(self, fn: Callable, max_plans: int = 16, **plan_kwargs):
        self.fn, self.max_plans, self.kw = fn, int(max_plans), plan_kwargs
        self.plans: dict = {}
        self.last: Optional[TrainStepGraph] = None

    def step(self, *inputs: Tensor, key=None):
        k = (key,) + tuple((tuple(t.shape), t.dtype) for t in inputs)
        plan = self.plans.pop(k, None)
        if plan is None:
            plan = TrainStepGraph(self.fn, **self.kw)
            while len(self.plans) >= self.max_plans:
                self.plans.pop(next(iter(self.plans)))
        self.plans[k] = plan
        self.last = plan
        return plan.step(*inputs)

    def finish(self) -> None:
        if self.last is not None:
            self.last.finish()

    @property
    def mode(self) -> str:
        return self.last.mode if self.last is not None else "eager (warm-up)"

    def summary(self) -> dict:
        return {"shapes": len(self.plans), "recorded": sum(1 for p in self.plans.values() if p.recorded),
                "eager_only": [p.eager_only for p in self.plans.values() if p.eager_only]}

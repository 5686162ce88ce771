"""Step plans: record a hot-path step once, replay it with one host call (include/deeplip_hip.h, dlip_plan_*).

The reference walks its test list one utterance at a time from Python (train_fusion.py:338-358); this
engine batches the list, and a batch step is ~45 dlip_* launches.  Issued one by one from Python the GPU
waits for the host between them, so the steady-state loop does not issue them at all:

    plan = StepPlan(step_fn, clips, mels)      # runs step_fn twice on a private stream: warm-up, then record
    fused = plan.run()                         # one dlip_plan_run call; returns the recorded output tensors
    fused = plan(new_clips, new_mels)          # copies into the recorded input buffers, then runs

What is recorded is exactly what ``step_fn`` launches through ``deeplip_amd.ops`` (every op is one dlip_*
call on the current stream).  The tensors of the recorded step -- inputs, every intermediate, outputs, the
packed weights -- are owned by the plan's arena for its lifetime: during the warm-up pass ``ops`` hands out
fresh ``torch.empty`` blocks and the arena keeps them; during the recorded pass it hands out the SAME blocks
in the same order (shapes are checked), so nothing is allocated while the stream is being captured and the
addresses baked into the plan stay valid and private.  ``step_fn`` must be launch-only: no host
synchronisation, no ``.item()``/``.cpu()``, no host->device copies (pass lengths etc. as device tensors).

A plan is tied to the weights it was recorded with: ``plan.run()`` raises ``StalePlanError`` after a
``load_state_dict`` / ``.to()`` / train-eval switch of any model (re-record it), mirroring the pack cache
(deeplip_amd/packing.py).
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, List, Optional, Sequence, Tuple

import torch

from . import holders, ops
from ._lib import DeepLipHipError, StatusBlock, check, check_range, lib, range_scope, scope_slots

Tensor = torch.Tensor


class StalePlanError(DeepLipHipError):
    pass


class Arena:
    """Owns the buffers of one recorded step.  mode 'collect': allocate and remember; 'replay': hand the
    remembered buffers out again in order."""

    def __init__(self):
        self.bufs: List[Tensor] = []
        self.keep: list = []          # packed weights and anything else the recorded launches address
        self.modules: dict = {}       # id(module) -> (module, device, state_version at record time): whose packs those are
        self.mode = "collect"
        self.cursor = 0

    def take(self, shape: Tuple[int, ...], device, dtype) -> Tensor:
        if self.mode == "collect":
            t = torch.empty(shape, device=device, dtype=dtype)
            self.bufs.append(t)
            return t
        if self.cursor >= len(self.bufs):
            raise DeepLipHipError("StepPlan: the recorded pass allocates more tensors than the warm-up pass did "
                                  "(the step function must be deterministic in its launches)")
        t = self.bufs[self.cursor]
        self.cursor += 1
        if tuple(t.shape) != tuple(shape) or t.dtype != dtype:
            raise DeepLipHipError(f"StepPlan: allocation #{self.cursor - 1} was {tuple(t.shape)}/{t.dtype} in the warm-up "
                                  f"pass and is {tuple(shape)}/{dtype} in the recorded pass")
        return t

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.bufs)


def _flatten(out):
    if isinstance(out, Tensor):
        return [out]
    if isinstance(out, (tuple, list)):
        return [t for o in out for t in _flatten(o)]
    if out is None:
        return []
    raise TypeError(f"StepPlan: step function returned {type(out).__name__}; expected tensors")


class _NameHook:
    """ops.LAUNCH_HOOK that only notes (kernel instance, algorithmic FLOPs) per launch, in order."""

    def __init__(self):
        self.seen: list = []

    def begin(self, name, flops):
        self.seen.append((name, flops, torch.cuda.current_stream().cuda_stream))   # + the stream the launch goes to (a step forks)
        return None

    def end(self, tok):
        pass


# kernels that time themselves inside a span scope (dlip_span_next in their launch path), as ops.LAUNCH_HOOK names them
SPAN_KERNELS = ("conv_igemm_f16x3_dma_kernel", "conv_rows_f16x3_kernel", "conv_win_f16x3_kernel", "stem3d_pool_f16x3_kernel")


class StepPlan:
    """``fn(*inputs)`` recorded on a private HIP stream; see the module docstring."""

    def __init__(self, fn: Callable, *inputs: Tensor, stream: Optional[torch.cuda.Stream] = None, spans: bool = False,
                 private_status: bool = False):
        for i, t in enumerate(inputs):
            if not (isinstance(t, Tensor) and t.is_cuda):
                raise DeepLipHipError(f"StepPlan: input {i} must be a CUDA (ROCm) tensor")
        lib()
        self.fn = fn
        self.inputs: Tuple[Tensor, ...] = tuple(t.contiguous() for t in inputs)   # the recorded input buffers
        self.device = self.inputs[0].device if self.inputs else torch.device("cuda", torch.cuda.current_device())
        # The recording stream is ALWAYS a private one (`stream`, if given, is only ordered in front of and behind the recording).
        # Round 4: a plan recorded on a stream that had carried torch.distributed collectives killed the job about one time in
        # five -- "Process group watchdog thread terminated with exception: HIP error: operation not permitted on an event last
        # recorded in a capturing stream".  Established by tools/probes/capture_race.py --long (a capture held open across three polls
        # of the process group's watchdog; profiles/r4/capture_race_probe.txt): collectives issued on the recording stream -> the
        # abort at the first plan; collectives on any other stream -> 12 of 12 plans fine.  The mechanism that fits (inferred, not
        # read in torch's or HIP's source): the events the watchdog polls sit on the stream that was current when a synchronous
        # collective was issued, and HIP refuses a query while THAT STREAM is capturing, although the record predates the capture
        # and the capture is thread-local (csrc/plan.hip).  A stream no collective ever ran on cannot be hit, whatever the timing.
        self.stream = torch.cuda.Stream(device=self.device)
        self._order_with = stream
        self.arena = Arena()
        self._range_slots = scope_slots(self.device)
        # the plan's OWN range-status block (dlip_status_scope): the recorded launches report to it on every replay, so a pipeline
        # knows which batch left the f16x3 range (take_range_error) -- check_range() sees it as well
        self.status = StatusBlock()
        # private_status: the owner settles every report of this plan itself, the recording passes' included (a pipeline records on
        # whatever batch comes first -- possibly the very batch that is out of range -- and repairs it after its replay)
        self.status.private = bool(private_status)
        # spans=True: every launch of the LDS-DMA convolution kernel times itself in-kernel on every replay (dlip_span_scope_*);
        # span_names = those launches in order, as ops.LAUNCH_HOOK names them (instance, algorithmic FLOPs)
        self._spans = None
        self.span_names: list = []
        self.last_stream_us: dict = {}
        if spans:
            n = 256
            pairs = torch.zeros((n, 16), dtype=torch.int64, device=self.device)   # one 128-byte record per launch (include/deeplip_hip.h)
            pairs[:, 0] = -1                                   # start ~0, the eight end words 0: armed
            self._spans = (pairs, torch.zeros((n, 2), dtype=torch.int64, device=self.device), n)
        self._handle = C.c_void_p()
        self.outputs = None
        caller = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(caller)
        if self._order_with is not None:
            self.stream.wait_stream(self._order_with)
        with torch.cuda.stream(self.stream), torch.no_grad():
            self._pass("collect")                 # warm-up: packs weights, registers the workspace, fills the arena
            self.stream.synchronize()
            check(lib().dlip_plan_begin(self.stream.cuda_stream), "dlip_plan_begin")
            try:
                out = self._pass("replay")
            except BaseException:
                h = C.c_void_p()
                lib().dlip_plan_end(self.stream.cuda_stream, C.byref(h))   # leave capture mode before propagating
                lib().dlip_plan_destroy(h)
                raise
            check(lib().dlip_plan_end(self.stream.cuda_stream, C.byref(self._handle)), "dlip_plan_end")
        if self.arena.cursor != len(self.arena.bufs):
            raise DeepLipHipError("StepPlan: the recorded pass made fewer allocations than the warm-up pass")
        self.outputs = out
        self._out_flat = _flatten(out)
        self.launches = int(lib().dlip_plan_launches(self._handle))
        self._gen = holders.PACK_GEN[0]
        caller.wait_stream(self.stream)
        if self._order_with is not None:
            self._order_with.wait_stream(self.stream)

    def _pass(self, mode: str):
        self.arena.mode, self.arena.cursor = mode, 0
        prev = ops.ARENA
        ops.ARENA = self.arena
        names = _NameHook() if self._spans is not None else None
        prev_hook = ops.LAUNCH_HOOK
        try:
            if names is not None:
                ops.LAUNCH_HOOK = names
                check(lib().dlip_span_scope_begin(self._spans[0].data_ptr(), self._spans[1].data_ptr(), self._spans[2]), "dlip_span_scope_begin")
            try:
                # one low-side range scope per pass over the plan's own evidence words: the verdict kernel is the last launch of
                # the recorded step, so every replay reports (and re-zeroes) for itself
                with self.status.scope(), range_scope(self._range_slots):
                    out = self.fn(*self.inputs)
            finally:
                if names is not None:
                    used = C.c_int32()
                    lib().dlip_span_scope_end(torch.cuda.current_stream(self.device).cuda_stream, C.byref(used))
                    self.span_names = [x for x in names.seen if any(k in x[0] for k in SPAN_KERNELS)]
                    if len(self.span_names) != used.value:
                        raise DeepLipHipError(f"StepPlan spans: {used.value} timed launches but {len(self.span_names)} named ones")
            return out
        finally:
            ops.ARENA = prev
            ops.LAUNCH_HOOK = prev_hook

    def run(self, check_reports: bool = True):
        """Replay the step on the current stream (asynchronous); returns the recorded output tensor(s),
        which the next run overwrites.  ``check_reports=False``: the caller settles range reports itself (pipeline.py)."""
        if self._gen != holders.PACK_GEN[0]:
            raise StalePlanError("StepPlan: model weights / placement changed since the plan was recorded; record a new plan")
        # in-place parameter updates (optimizer.step() on an eval-mode model, hand edits, `p.data = ...`) leave the generation
        # alone but change the fingerprint the eager path compares too: a replay must not run on packs baked from older values
        from . import packing
        for module, device, ver in self.arena.modules.values():
            if packing.state_version(module, device) != ver:
                raise StalePlanError("StepPlan: a model's parameters were modified in place since the plan was recorded (its packed "
                                     "weights are stale); record a new plan")
        if check_reports:
            check_range()       # f16x3 overflow reported by an earlier replay (host read, no synchronisation)
        check(lib().dlip_plan_run(self._handle, torch.cuda.current_stream(self.device).cuda_stream), "dlip_plan_run")
        return self.outputs

    def take_range_error(self):
        """The range error the plan's completed replays reported since the last look (cleared), or None.  Host memory read."""
        return self.status.take()

    def span_summary(self, reset: bool = True) -> dict:
        """Per kernel instance: launches per replay, replays seen, mean in-kernel span (first workgroup in -> last workgroup out,
        100 MHz clock) and the algorithmic TFLOP/s over it -- measured by the replayed launches themselves.  Synchronises."""
        if self._spans is None:
            return {}
        torch.cuda.synchronize(self.device)
        acc = self._spans[1].cpu().numpy()
        out = {}
        streams: dict = {}
        self.last_stream_us = {}                        # "stream0" (the first stream a timed launch went to), "stream1", ... -> sum of spans per replay
        for i, (name, flops, stream) in enumerate(self.span_names):
            ticks, cnt = int(acc[i, 0]), int(acc[i, 1])
            if cnt == 0:
                continue
            e = out.setdefault(name, {"launches_per_step": 0, "replays": cnt, "us_sum": 0.0, "flops": 0.0})
            e["launches_per_step"] += 1
            e["us_sum"] += ticks / cnt / 100.0          # 100 MHz ticks -> us, mean over the replays
            e["flops"] += flops
            key = "stream%d" % streams.setdefault(stream, len(streams))
            self.last_stream_us[key] = self.last_stream_us.get(key, 0.0) + ticks / cnt / 100.0
        for e in out.values():
            e["avg_launch_us"] = round(e["us_sum"] / e["launches_per_step"], 2)
            e["tflops"] = round(e["flops"] / (e["us_sum"] * 1e-6) / 1e12, 2)
            e["us_sum"] = round(e["us_sum"], 2)
            del e["flops"]
        if reset:
            self._spans[1].zero_()
        return out

    def __call__(self, *new_inputs: Tensor):
        if len(new_inputs) != len(self.inputs):
            raise ValueError(f"StepPlan: expected {len(self.inputs)} inputs, got {len(new_inputs)}")
        for dst, src in zip(self.inputs, new_inputs):
            if src is not dst:
                if tuple(src.shape) != tuple(dst.shape):
                    raise ValueError(f"StepPlan: input shape {tuple(src.shape)} != recorded {tuple(dst.shape)}")
                dst.copy_(src, non_blocking=True)
        return self.run()

    def close(self):
        h, self._handle = self._handle, C.c_void_p()
        if h:
            torch.cuda.synchronize(self.device)
            lib().dlip_plan_destroy(h)
            check_range()       # a range report of the LAST replay (every earlier one surfaced at the next run())

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

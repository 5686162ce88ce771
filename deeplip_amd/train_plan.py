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
    """``fn(*inputs) -> tensor or tuple of tensors`` performs one complete optimisation step on device tensors ``inputs``."""

    def __init__(self, fn: Callable, eager_steps: int = 2, device: Optional[torch.device] = None, branch_streams: bool = True):
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

    def _step(self, inputs, cur):
        with torch.cuda.stream(self.stream):
            if self.calls < self.eager_steps:
                out = self.fn(*inputs)
            else:
                if self.graph is None:
                    self.static = tuple(torch.empty_like(t) for t in inputs)
                    for d, s in zip(self.static, inputs):
                        d.copy_(s, non_blocking=True)
                    self.stream.synchronize()
                    g = torch.cuda.CUDAGraph()
                    # Inside a torch.distributed job the process group's watchdog THREAD polls the events of earlier collectives;
                    # under the default ("global") capture mode such a call from another thread aborts the capture ("operation not
                    # permitted when stream is capturing" -- seen in one run of two).  "thread_local" confines the check to this
                    # thread; the launches of the autograd worker threads are captured either way (they go to the capturing stream).
                    import torch.distributed as _dist
                    in_job = _dist.is_available() and _dist.is_initialized()
                    mode = "thread_local" if in_job else "global"
                    if in_job:
                        # ... and it must hold NO work of the eager steps when the capture begins: a recorded DP step captures its
                        # bucket all-reduces, which pulls the process group's own stream into the capture, and HIP refuses the
                        # watchdog's query of an EARLIER collective's end event (recorded on that stream before the capture, long
                        # complete) while the stream is capturing -- "operation not permitted on an event last recorded in a
                        # capturing stream", the job aborts (seen once in seven runs of tests/test_rccl_gpu.py, round 5).  The
                        # watchdog retires completed work at its next poll (every 100 ms): everything is complete after the
                        # synchronize, five polls later its list is empty; captured collectives are never enqueued to it.
                        import time
                        torch.cuda.synchronize(self.device)
                        time.sleep(0.5)
                    with torch.cuda.graph(g, stream=self.stream, capture_error_mode=mode):
                        self.outputs = self.fn(*self.static)
                    self.graph = g
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

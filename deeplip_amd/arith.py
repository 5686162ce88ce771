"""Which arithmetic the engine's contractions run in -- ONE switch for the library, the three entry points and the bench.

    f16x3   every product of fp32 values as hi*hi + hi*lo + lo*hi on the f16 matrix core, fp32 accumulate (fp32-grade: 5e-7 against
            the CPU reference where the bar is 1e-4; three times the exact mode's speed).  A value the split format cannot hold
            (|v| >= 65520, or a whole tensor below 2^-2) RAISES DeepLipRangeError at the next check.
    f32     exact fp32 MFMA: the reference's arithmetic (models/video_models/model.py:82-85 computes in fp32 end to end).
    auto    f16x3, and what leaves its range is computed again -- the same batch, the same process, the same model object -- on
            the exact f32 pack, logged and counted (``STATS``).  The default: the measured configuration IS the product's.

Precedence: an explicit argument (``--arith`` of the entry points) > the environment (``DLIP_ARITH``) > the config file's
``model.arith`` (``arith`` in conf/video_config.json) > "auto".  ``configure()`` runs before the first weight pack; the packed-weight
caches are keyed by the mode, so both packs of a model can be alive at once (the f32 one is built the first time it is needed).

The reference has no such switch (everything is torch fp32: train_fusion.py:338-358, train_audio.py:343-373, train_video.py:129);
it exists because the hot path here is hand-written MFMA arithmetic whose fast form has a range.
"""
from __future__ import annotations

import contextlib
import functools
import logging
import os
import threading
from typing import Callable, Optional

import torch

from . import _lib, packing

log = logging.getLogger("deeplip_amd.arith")

MODES = ("auto", "f16x3", "f32")
ENV = "DLIP_ARITH"

# batches / calls computed again in f32 by the auto mode, the last one's story, and how many of those re-runs also CALIBRATED a
# model's activation exponents (packing.act_exponents): from then on that model's batches stay on the f16x3 path
STATS = {"f32_reruns": 0, "calibrations": 0, "last": None}
CALIBRATE = True               # auto: the exact re-run of an out-of-range batch also calibrates activation exponents (False: it only repairs the batch)
_tls = threading.local()


def resolve(explicit: Optional[str] = None, config: Optional[str] = None) -> str:
    """The mode to run in: ``explicit`` (a command-line flag) > $DLIP_ARITH > ``config`` (the file's key) > "auto"."""
    for src, v in (("argument", explicit), ("$" + ENV, os.environ.get(ENV)), ("config", config)):
        if v is None or v == "":
            continue
        v = str(v).lower()
        if v not in MODES:
            raise ValueError(f"arith: {src} says {v!r}; expected one of {', '.join(MODES)}")
        return v
    return "auto"


MODE = resolve()               # the library's default before anybody calls configure(): $DLIP_ARITH, else "auto" (packing.PRECISION agrees)


def configure(mode: Optional[str] = None, config: Optional[str] = None) -> str:
    """Set the arithmetic of every pack made from now on (eval path: packing.PRECISION; train path: autograd_video.TRAIN_CONV)."""
    global MODE
    m = resolve(mode, config)
    packing.set_precision("f32" if m == "f32" else "f16x3")
    from . import autograd_video as av
    av.TRAIN_CONV = "f32" if m == "f32" else "f16x3"
    MODE = m
    return m


def fallback_enabled() -> bool:
    return MODE == "auto" and packing.PRECISION == "f16x3"


@contextlib.contextmanager
def exact():
    """The launches inside run on the exact f32 packs (built on first use, cached beside the f16x3 ones)."""
    prev = packing.PRECISION
    packing.set_precision("f32")
    try:
        yield
    finally:
        packing.set_precision(prev)


def note_rerun(what: str, err: Exception) -> None:
    STATS["f32_reruns"] += 1
    STATS["last"] = {"what": what, "reason": str(err).split(";")[0][:200]}
    log.warning("arith auto: %s left the range of the f16x3 arithmetic (%s); computed again in exact f32 (%d so far)",
                what, STATS["last"]["reason"], STATS["f32_reruns"])


def rerun_exact(fn: Callable, *args, what: str = "a batch", err: Optional[Exception] = None, calibrate: bool = True):
    """``fn(*args)`` eagerly on the f32 packs; the results are complete (synchronised, range-checked) on return.  ``calibrate``: the
    exact pass also measures every tensor the f16x3 path stores split and gives the models it ran activation exponents (powers of two
    folded into their f16x3 packs: packing.calib_finish) -- the next batch of the same kind stays inside the fast arithmetic's range."""
    note_rerun(what, err or _lib.DeepLipRangeError("range"))
    calibrate = calibrate and CALIBRATE
    if calibrate:
        packing.CALIB = {}
    try:
        with exact(), torch.no_grad():
            out = fn(*args)
    finally:
        n = packing.calib_finish() if calibrate else 0
    torch.cuda.synchronize()
    _lib.check_range()          # exact fp32 reports nothing of the split format; a stale word must not outlive the re-run
    if n:
        STATS["calibrations"] += n
        log.warning("arith auto: activation exponents calibrated for %d model(s) on that batch (%d so far): their f16x3 packs are rebuilt, "
                    "recorded plans re-recorded", n, STATS["calibrations"])
    return out


def calibrate(fn: Callable, *args):
    """Explicit calibration (any mode): run ``fn(*args)`` -- eval-mode forwards of the models to calibrate, on a representative batch --
    in exact fp32 and give those models activation exponents for their f16x3 packs (packing.calib_finish).  Returns fn's (exact)
    result.  Under ``auto`` this happens by itself on the first out-of-range batch; under ``f16x3`` it is how a checkpoint whose
    activations live outside the split format is made to run there without an exception."""
    packing.CALIB = {}
    try:
        with exact(), torch.no_grad():
            out = fn(*args)
    finally:
        n = packing.calib_finish()
    STATS["calibrations"] += n
    return out


def guarded_eval(fn):
    """Method decorator of the encoders' eval-mode entry points.  Under ``auto`` an EAGER call (not one being recorded into a step
    plan, not one nested in another guarded call) is followed by a synchronise + range check, and a range error turns into a second
    pass of the same call on the f32 pack.  Recorded steps are guarded per replay by their pipelines instead (pipeline.py)."""

    @functools.wraps(fn)
    def wrapper(self, *a, **k):
        from . import ops
        if (getattr(self, "training", False) or not fallback_enabled() or ops.ARENA is not None or getattr(_tls, "depth", 0) > 0
                or torch.cuda.is_current_stream_capturing()):
            return fn(self, *a, **k)
        _lib.check_range()      # a report of EARLIER launches is an error of theirs, not a reason to re-run this call
        _tls.depth = 1
        try:
            try:
                out = fn(self, *a, **k)
                _lib.check_range(sync=True)
                return out
            except _lib.DeepLipRangeError as ex:
                return rerun_exact(lambda: fn(self, *a, **k), what=f"{type(self).__name__}.{fn.__name__}", err=ex)
        finally:
            _tls.depth = 0
    return wrapper


def add_argument(parser) -> None:
    parser.add_argument("--arith", default=None, choices=list(MODES),
                        help="arithmetic of the engine's contractions: f16x3 (split fp16, 3 MFMAs per product, fp32-grade), f32 (exact "
                             "fp32 MFMA), auto (f16x3; a batch that leaves its range is computed again in f32 in the same process). "
                             f"Default: ${ENV}, else the config's key, else auto")

"""torch.autograd plumbing for the TRAIN-MODE lip-clip encoder.

What torch.autograd does for ``Lipreading.forward`` under ``model.train()`` (train_video.py:108-169 over
models/video_models/model.py:80-105, resnet.py:28-127, tcn.py:28-116), with every forward and backward
step a ``dlip_*`` launch: the convolutions (forward, data gradient, weight gradient) on the implicit-GEMM
kernels, BatchNorm with batch statistics on the row-BN kernels of the speech encoder
(encoder_train_ops.hip), PReLU / max-pool / average-pool / masked-mean / dropout on
video_train_ops.hip.  torch supplies the tape, parameter storage, slicing / concatenation of tensors
and the optimizer.  Activations are channels-last fp32 ([N,H,W,C]; Conv1d as H = 1) throughout;
parameters keep the reference layouts, so the state dict is the reference's.
"""
from __future__ import annotations

import weakref

import torch
from torch.autograd import Function

from . import _lib, ops
from ._lib import LIFT_BCAST, LIFT_WORDS, check, lib, ptr, stream_handle
from .autograd import BNRowsActFn, _permute3, _ws


_CONST = {}


def const_vec(n: int, value: float, device):
    """A read-only [n] vector of ``value`` (unit scales, zero shifts of the conv epilogues): made once per (device, n, value) --
    276 torch fill launches of a B = 32 training step were these."""
    key = (str(device), int(n), float(value))
    t = _CONST.get(key)
    if t is None:
        t = torch.full((n,), float(value), device=device, dtype=torch.float32)
        # The vector is SHARED by every later launch on every stream, but its fill runs on whichever stream is current now --
        # possibly a side stream of a forked branch (video.BRANCH_STREAMS), which nobody else waits for: the first step read
        # garbage scales on the other streams (a range error with the default stream as main stream, silently different weights
        # otherwise).  Created once: wait for it.
        if not torch.cuda.is_current_stream_capturing():
            torch.cuda.current_stream(t.device).synchronize()
        _CONST[key] = t
    return t


def _colsum_rows(x2):
    M, C_ = x2.shape
    _lib.ensure_conv_workspace()
    y = torch.empty((C_,), device=x2.device, dtype=torch.float32)
    check(lib().dlip_colsum_rows_f32(ptr(x2), ptr(y), ptr(_ws(M, C_, x2.device)), M, C_, stream_handle()), "dlip_colsum_rows_f32")
    return y


def wgrad_gemm(dz_rows, tap_rows, n_taps):
    """dW[s, c, k] = sum_j rows_s[j, c] * dz[j, k]: one GEMM per filter tap with the J output positions as the
    reduction.  Both operands are transposed to reduction-major, split into (hi, lo) fp16 pairs (the gradient
    after a power-of-two lift into fp16's normal range) and multiplied by the LDS-DMA kernel, whose balanced
    work split is what fills the chip on a [C x K] product with J ~ 1e5..1e6 (same scheme as the speech
    encoder's Conv1d weight gradient, autograd._conv1d_wgrad).  ``tap_rows(s)`` -> [J, C] contiguous."""
    J, K = dz_rows.shape
    dev = dz_rows.device
    J32 = (J + 31) // 32 * 32
    dzT = torch.empty((1, K, J32), device=dev, dtype=torch.float32)
    check(lib().dlip_nct_to_ntc_f32(ptr(dz_rows), ptr(dzT), 1, J, K, J32, stream_handle()), "dlip_nct_to_ntc_f32")
    scale2 = torch.empty((2,), device=dev, dtype=torch.float32)
    check(lib().dlip_pow2_scale_f32(ptr(dzT), ptr(scale2), dzT.numel(), 1024.0, stream_handle()), "dlip_pow2_scale_f32")
    dzT_s = torch.empty_like(dzT)
    check(lib().dlip_split_pack_scaled_f32(ptr(dzT), ptr(dzT_s), ptr(scale2), K, J32, stream_handle()), "dlip_split_pack_scaled_f32")
    inv = lift_inv(scale2, K)
    ones = const_vec(K, 1.0, dev)
    zeros = const_vec(K, 0.0, dev)
    out = None
    xT = None
    for s in range(n_taps):
        rows = tap_rows(s)
        Cx = rows.shape[1]
        if out is None:
            out = torch.empty((n_taps, Cx, K), device=dev, dtype=torch.float32)
            xT = torch.empty((1, Cx, J32), device=dev, dtype=torch.float32)
        check(lib().dlip_nct_to_ntc_f32(ptr(rows), ptr(xT), 1, J, Cx, J32, stream_handle()), "dlip_nct_to_ntc_f32")
        xT_s = ops.split_pack(xT.view(Cx, J32))
        ops.conv_nhwc(xT_s.view(1, 1, Cx, J32), dzT_s.view(K, 1, 1, J32), None, w_scale=ones, x_split=True,
                      post_scale=inv, post_shift=zeros, out=out[s].view(1, 1, Cx, K))
    return out


# Arithmetic of the convolutions of a TRAINING step (forward and data gradient; the weight gradient has always been split-fp16):
# "f16x3" = the split-fp16 kernels of the extraction path -- activations split once per convolution (dlip_split_pack_f32; a
# gradient after its power-of-two lift), the CURRENT weights split on the device (dlip_split_weights_rows_f32), fp32 out: 2.6x the
# rate of the exact-fp32 MFMA kernel at fp32-grade accuracy -- or "f32", the exact kernel (what the training path used before).
TRAIN_CONV = "f32" if __import__("os").environ.get("DLIP_ARITH", "auto").strip().lower() == "f32" else "f16x3"    # (deeplip_amd.arith.configure sets it)
WGRAD_ODD_PITCH = True


def pow2_lift(t):
    """Device pair (2^e, 2^-e) that lifts a gradient tensor's largest magnitude to ~1024 (dlip_pow2_scale_f32): computed ONCE per
    backward step and shared by the data-gradient convolution and the weight-gradient operand of the same dy."""
    ready = getattr(t, "_dlip_lift", None)       # the producer of this gradient (a BatchNorm backward) formed it while writing t
    if ready is not None:
        return ready
    scale2 = torch.empty((LIFT_WORDS,), device=t.device, dtype=torch.float32)
    check(lib().dlip_pow2_lift_f32(ptr(t), ptr(scale2), t.numel(), 1024.0, stream_handle()), "dlip_pow2_lift_f32")
    return scale2


def lift_inv(scale2, K):
    """[K] copies of 2^-e: the post_scale vector of a convolution over a lifted gradient.  A DLIP_LIFT_WORDS buffer carries 2048
    of them behind the pair (written by the kernel that finalised the lift); anything else gets a fill launch."""
    if scale2.numel() >= 2 + K and K <= LIFT_BCAST:
        return scale2[2:2 + K]
    inv = torch.empty((K,), device=scale2.device, dtype=torch.float32)
    check(lib().dlip_fill_from_scalar_f32(scale2[1:].data_ptr(), ptr(inv), K, stream_handle()), "dlip_fill_from_scalar_f32")
    return inv


# ---- the step's weight images in ONE launch (round 4) ----------------------------------------------------------------------------
# Every convolution of a training step needs the split-fp16 image of its CURRENT weights twice: the forward bank and the flipped /
# transposed data-gradient bank (conv_train below: dlip_split_weights_perm_f32, 94 launches of 8 - 10 us per lip-clip step, each in front
# of the convolution that waits for it).  The first step REGISTERS each (weight, bank) it meets together with persistent output
# buffers; from then on `prepare_weights()` -- called at the start of a train-mode forward -- writes all of them in one launch
# (dlip_split_weights_multi_f32) and conv_train takes the prepared image when the weight has not been modified in place since (tensor
# version counters: an optimizer step between forward and backward simply falls back to the per-convolution launch).
class _WeightPrep:
    def __init__(self):
        self.entries = {}          # (data_ptr, shape, transposed, Cw) -> dict
        self.order = []
        self.table = None          # (descs_dev, blocks_dev, n_blocks) for len(order) entries
        self.table_len = 0
        self.stats = {"hit": 0, "miss": 0, "stale": 0, "launch": 0, "skipped": 0}   # (Python-side counts: a replayed graph adds none)
        # the one launch runs on a stream of its own, beside the stem (whose weights have their own image kernel) and the first
        # BatchNorm passes: the first convolution that takes a prepared image is ~0.6 ms into the forward.  A consumer's stream
        # waits for the launch's event the first time it takes an image in a step.
        self.side = None
        self.event = None
        self.joined = set()
        self.step = 0              # prepare() calls so far (Python-side: a replayed graph adds none)
        self.pinned = False        # a stream capture has seen this registry: a recorded graph may address its buffers -> nothing is dropped

    def __deepcopy__(self, memo):     # (a copied / pickled model starts with an empty registry: addresses, streams and events are this one's)
        return _WeightPrep()

    def __reduce__(self):
        return (_WeightPrep, ())

    def lookup(self, key, w_ref):
        e = self.entries.get(key)
        if e is not None:
            e["seen"] = self.step
        if e is not None and e["ver"] == w_ref._version and e["ver"] >= 0:
            self.stats["hit"] += 1
            st = torch.cuda.current_stream()
            if st.cuda_stream not in self.joined:
                st.wait_event(self.event)
                self.joined.add(st.cuda_stream)
            return e["ws"], e["wsc"]
        self.stats["stale" if e is not None else "miss"] += 1
        return None

    def register(self, key, w_ref, Ko, Ci, T, mode, Cw, ws, wsc):
        if key not in self.entries and not torch.cuda.is_current_stream_capturing():
            rows = Ko if mode == 0 else Ci
            # persistent buffers of the registry (the caller's ws / wsc are step-local temporaries)
            # (a DETACHED alias: it shares storage and version counter with the parameter, and keeps no autograd node alive --
            # holding the view the forward made, grad_fn and all, across steps crashed the end of a later step-graph capture)
            # ... and a WEAK reference to the tensor that owns the storage (the nn.Parameter; for `weight.view(...)` its base):
            # the alias alone pins the old storage for ever, so `param.data = ...` / `model.to()` / a deleted model could never be
            # noticed by comparing the alias's address with the key -- the owner's current address is what tells
            base = w_ref._base if w_ref._base is not None else w_ref
            self.entries[key] = dict(t=w_ref.detach(), base=weakref.ref(base), off=key[0] - base.data_ptr(), Ko=Ko, Ci=Ci, T=T, mode=mode,
                                     Cw=Cw, rows=rows, ver=-1, seen=self.step, ws=torch.empty_like(ws), wsc=torch.empty_like(wsc))
            self.order.append(key)

    def _build(self, device):
        import numpy as np
        dt = np.dtype([("w", "<u8"), ("ws", "<u8"), ("sc", "<u8"), ("K", "<i4"), ("C", "<i4"), ("T", "<i4"), ("Cp", "<i4"), ("mode", "<i4"), ("row0", "<i4")])
        assert dt.itemsize == 48
        descs = np.zeros(len(self.order), dtype=dt)
        blocks = []
        row0 = 0
        for i, k in enumerate(self.order):
            e = self.entries[k]
            descs[i] = (e["t"].data_ptr(), e["ws"].data_ptr(), e["wsc"].data_ptr(), e["Ko"], e["Ci"], e["T"], e["Cw"], e["mode"], row0)
            blocks.append(np.full(e["rows"], i, dtype=np.int32))
            row0 += e["rows"]
        blocks = np.concatenate(blocks)
        d_dev = torch.from_numpy(descs.view(np.uint8).copy()).to(device)
        b_dev = torch.from_numpy(blocks).to(device)
        max_row = max(self.entries[k]["T"] * self.entries[k]["Cw"] for k in self.order)    # floats of the longest output row
        self.table, self.table_len = (d_dev, b_dev, int(blocks.size), int(max_row)), len(self.order)

    def prepare(self):
        if not self.order:
            return
        import os
        if os.environ.get("DLIP_WEIGHT_PREP", "1") == "0":   # A/B switch: every convolution splits its own weights, as until round 4
            for k in self.order:
                self.entries[k]["ver"] = -1
            return
        capturing = torch.cuda.is_current_stream_capturing()
        self.pinned = self.pinned or capturing
        self.step += 1

        # Dropped: entries whose owner is gone or holds other storage now (`param.data = ...`, `.to()`, `.float()`, a deleted
        # model) -- their images would be re-split every step for nobody and pin the old weights plus two image buffers each --
        # and entries no convolution asked for during the last 8 steps (direct conv_train calls on weights that are not
        # this model's: the anonymous registry).  Never under a capture, never once a recorded graph may address the buffers.
        def _dead(k):
            e = self.entries[k]
            b = e["base"]()
            return b is None or b.data_ptr() + e["off"] != k[0] or self.step - e["seen"] > 8
        dead = [k for k in self.order if _dead(k)] if not self.pinned else []
        if dead and not capturing:
            for k in dead:
                del self.entries[k]
            self.order = [k for k in self.order if k in self.entries]
            self.table_len = -1
            if not self.order:
                return
        if self.table_len != len(self.order):
            if torch.cuda.is_current_stream_capturing():
                self.stats["skipped"] += 1
                return                                   # (no host-to-device copy inside a capture: this step splits per convolution)
            self._build(self.entries[self.order[0]]["ws"].device)
        d_dev, b_dev, n, max_row = self.table
        cur = torch.cuda.current_stream()
        if self.side is None:
            self.side = torch.cuda.Stream(device=d_dev.device)
        self.side.wait_stream(cur)                       # the previous step's last readers of the images are behind us
        with torch.cuda.stream(self.side):
            check(lib().dlip_split_weights_multi_f32(ptr(d_dev), ptr(b_dev), n, max_row, stream_handle()), "dlip_split_weights_multi_f32")
            self.event = torch.cuda.Event()
            self.event.record(self.side)
        self.joined = set()
        self.stats["launch"] += 1
        for k in self.order[:self.table_len]:
            e = self.entries[k]
            e["ver"] = e["t"]._version


WEIGHT_PREP = _WeightPrep()     # the registry in use: the one of the model whose train-mode forward ran last (or this anonymous one)


def prepare_weights(owner=None):
    """Start of a train-mode forward: all registered weight images of the step in one launch (see _WeightPrep).  ``owner`` (the
    model): its registry -- images, descriptor table, side stream -- lives on the model object and goes with it; what the step's
    convolutions register and take is the registry of the forward that ran last."""
    global WEIGHT_PREP
    if owner is not None:
        reg = owner.__dict__.get("_dlip_weight_prep")
        if reg is None:
            reg = _WeightPrep()
            owner.__dict__["_dlip_weight_prep"] = reg
        WEIGHT_PREP = reg
    if TRAIN_CONV == "f16x3":
        WEIGHT_PREP.prepare()


def conv_train(x, w_krsc, bias, stride=(1, 1), pad=(0, 0), dil=(1, 1), lift=False, scale2=None, w_ref=None, transposed=False,
               xs_ready=None, stats=None):
    """One convolution of a training step on NHWC fp32 ``x`` with CURRENT weights -> fp32 NHWC.  Weights: ``w_krsc`` [K,R,S,C]
    (kernel layout), or ``w_ref`` = the parameter itself in the reference layout [Ko,Ci,R,S] -- then ``transposed`` False means
    the forward filter bank and True the data gradient's (flipped taps, in / out channels swapped): the split-fp16 image is
    written straight from the parameter by dlip_split_weights_perm_f32, no permuted fp32 copy in between.
    ``lift``: x is a gradient (tiny magnitudes): multiply by a power of two into fp16's normal range before the split and
    divide the result by it (the conv epilogue's post_scale) -- exact.
    ``stats`` (a dict {"chunks": n > 0 from ops.conv_stats_chunks, "ws": float64 [n * K * 2]}): if this call ends on the split-fp16
    kernel, its epilogue also writes the column sums of the output into ws (the BatchNorm behind it then skips its statistics pass) and
    stats["done"] is set; otherwise the dict is left alone."""
    N, H, W, Cx = x.shape
    if w_ref is not None:
        Ko, Ci, R_, S_ = w_ref.shape
        K, Cw = (Ci, Ko) if transposed else (Ko, Ci)
        padded = (not transposed) and Cw < Cx and (Cw + 31) // 32 * 32 == Cx     # x carries zero channels up to the next 32 (a 24-feature input)
        if padded and TRAIN_CONV == "f16x3" and K % 4 == 0:
            Cw = Cx
        # a lifted gradient whose channel count is a multiple of 4 but not of 32 (1500): split with its rows zero-padded to the next 32
        pad_x = Cx
        if transposed and lift and TRAIN_CONV == "f16x3" and Cx % 32 and Cx % 4 == 0 and K % 4 == 0 and Cw == Cx and (xs_ready is None or scale2 is not None):
            pad_x = (Cx + 31) // 32 * 32
            dev = x.device
            if scale2 is None:
                scale2 = pow2_lift(x)
            ws = torch.empty((K, R_, S_, pad_x), device=dev, dtype=torch.float32)
            wsc = torch.empty((K,), device=dev, dtype=torch.float32)
            check(lib().dlip_split_weights_perm_f32(ptr(w_ref.contiguous()), ptr(ws), ptr(wsc), Ko, Ci, R_ * S_, 1, pad_x, stream_handle()),
                  "dlip_split_weights_perm_f32")
            if xs_ready is not None:       # (round 6) the padded split operand [N,H,W,pad_x], lifted by scale2, written by the pass that formed the gradient
                xs = xs_ready
            else:
                xs = torch.empty((N, H, W, pad_x), device=dev, dtype=torch.float32)
                check(lib().dlip_split_pack_scaled_pad_f32(ptr(x), ptr(xs), ptr(scale2), N * H * W, Cx, pad_x, stream_handle()), "dlip_split_pack_scaled_pad_f32")
            return ops.conv_nhwc(xs, ws, None, stride=stride, pad=pad, dil=dil, w_scale=wsc, x_split=True, post_scale=lift_inv(scale2, K),
                                 post_shift=const_vec(K, 0.0, dev))
        if TRAIN_CONV != "f16x3" or Cx % 32 or K % 4 or Cw != Cx:
            w3 = w_ref.contiguous().view(Ko, Ci, R_ * S_)
            w_krsc = (_permute3(w3, (1, 2, 0), flip_axis=2).view(Ci, R_, S_, Ko) if transposed else _permute3(w3, (0, 2, 1)).view(Ko, R_, S_, Ci))
            return ops.conv_nhwc(x, w_krsc, bias, stride=stride, pad=pad, dil=dil)
        dev = x.device
        key = (w_ref.data_ptr(), tuple(w_ref.shape), bool(transposed), Cw)
        hit = WEIGHT_PREP.lookup(key, w_ref) if w_ref.is_contiguous() else None
        if hit is not None:
            ws, wsc = hit                                   # written by prepare_weights() at the start of this step
        else:
            ws = torch.empty((K, R_, S_, Cw), device=dev, dtype=torch.float32)
            wsc = torch.empty((K,), device=dev, dtype=torch.float32)
            check(lib().dlip_split_weights_perm_f32(ptr(w_ref.contiguous()), ptr(ws), ptr(wsc), Ko, Ci, R_ * S_, 1 if transposed else 0, Cw,
                                                    stream_handle()), "dlip_split_weights_perm_f32")
            if w_ref.is_contiguous():
                WEIGHT_PREP.register(key, w_ref, Ko, Ci, R_ * S_, 1 if transposed else 0, Cw, ws, wsc)
    else:
        K = w_krsc.shape[0]
        if TRAIN_CONV != "f16x3" or Cx % 32 or K % 4 or w_krsc.shape[3] != Cx:
            return ops.conv_nhwc(x, w_krsc, bias, stride=stride, pad=pad, dil=dil)
        dev = x.device
        L = w_krsc.numel() // K
        ws = torch.empty_like(w_krsc)
        wsc = torch.empty((K,), device=dev, dtype=torch.float32)
        check(lib().dlip_split_weights_rows_f32(ptr(w_krsc), ptr(ws), ptr(wsc), K, L, stream_handle()), "dlip_split_weights_rows_f32")
    if not lift:
        st = None
        if stats is not None and stats.get("chunks", 0) > 0:
            st = stats["ws"]
            stats["done"] = True
        return ops.conv_nhwc(xs_ready if xs_ready is not None else ops.split_pack(x), ws, bias, stride=stride, pad=pad, dil=dil, w_scale=wsc,
                             x_split=True, stats=st)
    if scale2 is None:
        scale2 = pow2_lift(x)
    if xs_ready is not None:        # (x in the split format, lifted by scale2, written by the pass that formed the weight gradient's image)
        xs = xs_ready
    else:
        xs = torch.empty_like(x)
        check(lib().dlip_split_pack_scaled_f32(ptr(x), ptr(xs), ptr(scale2), x.numel() // Cx, Cx, stream_handle()), "dlip_split_pack_scaled_f32")
    inv = lift_inv(scale2, K)
    zeros = const_vec(K, 0.0, dev)
    return ops.conv_nhwc(xs, ws, None, stride=stride, pad=pad, dil=dil, w_scale=wsc, x_split=True, post_scale=inv, post_shift=zeros)


def _pixel_pitch(t):
    """Floats between two pixels of an NHWC tensor.  NOT simply ``t.stride(2)``: torch ignores the strides of size-1 dimensions when it calls
    a tensor contiguous, so a [N,1,1,C] gradient that arrives as a permuted view ([N,C,1,1] -> NHWC) keeps stride(2) == 1 through
    ``.contiguous()`` -- and the operand producers rejected it (tools/probes/grad_fuzz.py: a Conv1d whose output is one frame wide)."""
    return t.shape[3] if t.is_contiguous() else t.stride(2)


def wgrad_conv_fused(x, dy, R, S, stride, pad, dil, scale2=None):
    """dW[(tap, c), k] of a Conv2d / Conv1d on NHWC tensors with BOTH operands written by dlip_wgrad_operand_f32: one pass over
    x (all R*S taps, reduction-major, split) and one over dy (power-of-two lift, reduction-major, split), then the one GEMM.
    Replaces tap gather + transpose + split (three round trips of a matrix R*S times the activation; 15 of the 61 ms of a
    B = 32 training step were those passes).  Returns [R*S, C, K]."""
    N, H, W, Cx = x.shape
    _, Ho, Wo, K = dy.shape
    dev = x.device
    J = N * Ho * Wo
    J32 = (J + 31) // 32 * 32
    if WGRAD_ODD_PITCH and (J32 // 32) % 2 == 0:
        J32 += 32            # an odd number of 128-B blocks per row: the 128 rows a slice touches spread over the memory channels
    taps = R * S
    if scale2 is None:
        scale2 = pow2_lift(dy)
    dzT_s = torch.empty((K, J32), device=dev, dtype=torch.float32)
    check(lib().dlip_wgrad_operand_f32(ptr(dy), ptr(dzT_s), J32, N, Ho, Wo, K, K, Ho, Wo, 1, 1, 1, 1, 1, 1, 0, 0, ptr(scale2), stream_handle()),
          "dlip_wgrad_operand_f32")
    xT_s = torch.empty((taps * Cx, J32), device=dev, dtype=torch.float32)
    check(lib().dlip_wgrad_operand_f32(ptr(x), ptr(xT_s), J32, N, H, W, Cx, _pixel_pitch(x), Ho, Wo, stride[0], stride[1], R, S, dil[0], dil[1],
                                       pad[0], pad[1], None, stream_handle()), "dlip_wgrad_operand_f32")
    inv = lift_inv(scale2, K)
    ones = const_vec(K, 1.0, dev)
    zeros = const_vec(K, 0.0, dev)
    out = torch.empty((taps * Cx, K), device=dev, dtype=torch.float32)
    ops.conv_nhwc(xT_s.view(1, 1, taps * Cx, J32), dzT_s.view(K, 1, 1, J32), None, w_scale=ones, x_split=True, post_scale=inv,
                  post_shift=zeros, out=out.view(1, 1, taps * Cx, K))
    return out.view(taps, Cx, K)


def operand_and_split(t2, scale2=None, bn=None):
    """A [J, C] row matrix (C % 64 == 0) -> (its reduction-major weight-gradient GEMM operand [C, J32], its split NHWC copy [J, C]) from
    ONE read (dlip_wgrad_operand_split_f32): the forward convolution's operand of a k = 1 layer's input beside its weight-gradient
    image, or -- with the lift ``scale2`` -- the data gradient's operand of dy beside dy's."""
    J, C_ = t2.shape
    J32 = (J + 31) // 32 * 32
    if WGRAD_ODD_PITCH and (J32 // 32) % 2 == 0:
        J32 += 32
    opT = torch.empty((C_, J32), device=t2.device, dtype=torch.float32)
    spl = torch.empty((J, C_), device=t2.device, dtype=torch.float32)
    if bn is not None:      # t2's values were never written: read the raw convolution output and apply its BatchNorm + activation on load
        z, mean, invstd, gamma, beta, slope = bn
        sv, sc = (slope, 0.0) if isinstance(slope, torch.Tensor) else (None, float(slope))
        check(lib().dlip_wgrad_operand_split_bn_f32(ptr(z), ptr(opT), J32, J, C_, ptr(mean), ptr(invstd), ptr(gamma), ptr(beta), ptr(sv), sc,
                                                    ptr(spl), stream_handle()), "dlip_wgrad_operand_split_bn_f32")
        return opT, spl
    check(lib().dlip_wgrad_operand_split_f32(ptr(t2), ptr(opT), J32, J, C_, ptr(scale2) if scale2 is not None else None, ptr(spl), stream_handle()),
          "dlip_wgrad_operand_split_f32")
    return opT, spl


def wgrad_gemm_operands(xT_s, dzT_s, scale2):
    """[rows of x^T, J32] x [K, J32] (both reduction-major, split; dz lifted by ``scale2``) -> [rows, K]: the weight-gradient GEMM."""
    rows, J32 = xT_s.shape
    K = dzT_s.shape[0]
    dev = xT_s.device
    out = torch.empty((rows, K), device=dev, dtype=torch.float32)
    ops.conv_nhwc(xT_s.view(1, 1, rows, J32), dzT_s.view(K, 1, 1, J32), None, w_scale=const_vec(K, 1.0, dev), x_split=True,
                  post_scale=lift_inv(scale2, K), post_shift=const_vec(K, 0.0, dev), out=out.view(1, 1, rows, K))
    return out


# How the convolutions' weight gradients run: "conv" = as a convolution over ONE transposed copy of x and of dy (wgrad_as_conv),
# "gemm" = one GEMM over the R*S shifted copies of x (wgrad_conv_fused: what round 3 started with).
WGRAD = "conv"


# Operand images of the weight gradient run as a convolution: slice-major ([image][32-image slice][H][W][32]: the taps a workgroup
# walks are adjacent 128-byte lines, dlip_wgrad_conv_f16x3) or pixel-major ([image][H][W][N32] through dlip_conv_nhwc_f16x3: every
# piece a DRAM page of its own -- kept for A/B runs and as the second implementation the tests compare).
WGRAD_SLICE_MAJOR = True
# Round 5: the weight-gradient convolution's epilogue stores dW in the reference layout [K, C, R, S] itself (dlip_wgrad_conv_f16x3's
# R, S): no [C, R', S', K] tensor, slice copy and permute launch behind every one of them.  False: round 4's path (tests compare the two).
WGRAD_DIRECT_LAYOUT = __import__("os").environ.get("DLIP_WGRAD_DIRECT", "1") != "0"      # (the environment switch: same-box A/B runs)


def _wgrad_conv_launch(xT, gT, inv, C_, H, W, K, Ho, Wo, N32, stride, pad, dil, RS=None):
    """``RS`` = (R, S): the result straight in the reference layout [K, C, R, S] (the kernel's epilogue stores it transposed and
    drops the positions beyond the layer's filter); None: [C, R', S', K]."""
    dev = xT.device
    Ro = (H + 2 * pad[0] - stride[0] * (Ho - 1) - 1) // dil[0] + 1
    So = (W + 2 * pad[1] - stride[1] * (Wo - 1) - 1) // dil[1] + 1
    out = torch.empty((K, C_, RS[0], RS[1]) if RS is not None else (C_, Ro, So, K), device=dev, dtype=torch.float32)
    check(lib().dlip_wgrad_conv_f16x3(ptr(xT), ptr(gT), ptr(inv), ptr(const_vec(K, 0.0, dev)), ptr(const_vec(K, 1.0, dev)), ptr(out), C_, H, W, K,
                                      Ho, Wo, N32, stride[0], stride[1], pad[0], pad[1], dil[0], dil[1], RS[0] if RS else 0, RS[1] if RS else 0,
                                      stream_handle()), "dlip_wgrad_conv_f16x3")
    return out


def wgrad_image(t, scale2=None, also_nhwc_split=False, bn=None):
    """The weight gradient's operand image of an NHWC tensor ``t`` [N,H,W,C] (dlip_wgrad_chwn_f32; layout per
    WGRAD_SLICE_MAJOR; times scale2[0] when given).  ``also_nhwc_split``: the convolution kernels' split activation format of the
    same (scaled) tensor from the same read -- the forward convolution's operand (t = x) or the data gradient's (t = dy).
    Returns (image [C,H,W,N32], split or None)."""
    N, H, W, C_ = t.shape
    N32 = (N + 31) // 32 * 32
    img = torch.empty((C_, H, W, N32), device=t.device, dtype=torch.float32)    # slice-major: [C][N32/32][H][W][32]
    spl = torch.empty_like(t) if also_nhwc_split else None
    if bn is not None:      # (see operand_and_split)
        z, mean, invstd, gamma, beta, slope = bn
        sv, sc = (slope, 0.0) if isinstance(slope, torch.Tensor) else (None, float(slope))
        check(lib().dlip_wgrad_chwn_bn_f32(ptr(z), ptr(img), N, H, W, C_, N32, ptr(mean), ptr(invstd), ptr(gamma), ptr(beta), ptr(sv), sc, ptr(spl),
                                           stream_handle()), "dlip_wgrad_chwn_bn_f32")
        return img, spl
    check(lib().dlip_wgrad_chwn_f32(ptr(t), ptr(img), N, H, W, C_, _pixel_pitch(t), N32, ptr(scale2) if scale2 is not None else None,
                                    1 if WGRAD_SLICE_MAJOR else 0, ptr(spl) if spl is not None else None, stream_handle()), "dlip_wgrad_chwn_f32")
    return img, spl


def wgrad_as_conv(x, dy, R, S, stride, pad, dil, scale2=None, xT=None, gT=None):
    """dW[c, r, s, k] = sum_{n,h,w} x[n, h*sh + r*dh - ph, w*sw + s*dw - pw, c] * dy[n, h, w, k] run as a CONVOLUTION on the
    engine's own conv kernel: input x' = x as [C][H][W][N] (the C channels play the batch, the N images the channels), filter
    g' = dy as [K][Ho][Wo][N], convolution stride = the layer's dilation, dilation = the layer's stride, same padding; the
    output [C, R', S', K] holds dW in its first R x S positions (R' > R when the layer's stride leaves a remainder).  Both
    operands are ONE split copy of their tensor (dlip_wgrad_chwn_f32; dy after its power-of-two lift) -- the reduction-major
    GEMM operand is R*S shifted copies of x: 1.03 GB written and read back per layer-1 convolution at B = 32, 0.9 ms of a
    0.1 ms data gradient's worth of FLOPs.  The Ho*Wo filter taps (484 on layer 1) are the conv kernel's address walk
    (its variant without the 32-bit tap mask).  ``xT`` / ``gT``: operand images already formed (wgrad_image; gT with the lift
    ``scale2``); then x / dy may be None.  Returns dW in the reference layout [K, C, R, S]."""
    if xT is None:
        xT, _ = wgrad_image(x)
    if gT is None:
        if scale2 is None:
            scale2 = pow2_lift(dy)
        gT, _ = wgrad_image(dy, scale2)
    Cx, H, W, N32 = xT.shape
    K, Ho, Wo, _ = gT.shape
    dev = xT.device
    sm = 1 if WGRAD_SLICE_MAJOR else 0
    inv = lift_inv(scale2, K)
    if sm and WGRAD_DIRECT_LAYOUT:
        return _wgrad_conv_launch(xT, gT, inv, Cx, H, W, K, Ho, Wo, N32, stride, pad, dil, RS=(R, S))   # [K, Cx, R, S] from the epilogue
    if sm:
        out = _wgrad_conv_launch(xT, gT, inv, Cx, H, W, K, Ho, Wo, N32, stride, pad, dil)   # [Cx, R', S', K]
    else:
        out = ops.conv_nhwc(xT, gT, None, stride=dil, pad=pad, dil=stride, w_scale=const_vec(K, 1.0, dev), x_split=True,
                            post_scale=inv, post_shift=const_vec(K, 0.0, dev))
    if out.shape[1] != R or out.shape[2] != S:
        out = out[:, :R, :S].contiguous()
    return _permute3(out.view(Cx, R * S, K), (2, 0, 1)).view(K, Cx, R, S)                   # the reference layout [K, C, R, S]


def wgrad_conv(x, dy, R, S, stride, pad, dil, scale2=None):
    """dW of a Conv2d / Conv1d (H = 1) in the REFERENCE layout [K, C, R, S]."""
    # (a 1x1 convolution has nothing to expand: its GEMM operand is one copy already, and as a convolution a strided one would
    # compute a 2x2 output for the one position it needs -- 326 vs 160 us on layer2.0's shortcut, tools/bench_wgrad.py)
    if WGRAD == "conv" and TRAIN_CONV == "f16x3" and R * S > 1:
        return wgrad_as_conv(x, dy, R, S, stride, pad, dil, scale2)
    dwt = wgrad_conv_fused(x, dy, R, S, stride, pad, dil, scale2)                           # [R*S, C, K]
    return _permute3(dwt, (2, 1, 0)).view(dy.shape[3], x.shape[3], R, S)


def _upsample_zero(dy, Hu, Wu, sh, sw, lift, Ci=4):
    """Zero insertion of a strided layer's output gradient dy [N,Ho,Wo,K] -> [N,Hu,Wu,K] for its data-gradient convolution.  Returns
    (tensor, split): with the split-fp16 arithmetic and K % 32 == 0 the lifted split operand is written directly (round 5:
    dlip_upsample_zero_split_f32) and the fp32 tensor is an UNWRITTEN placeholder that only carries the shape; otherwise the fp32 tensor
    and None."""
    N, Ho, Wo, K = dy.shape
    src = torch.empty((N, Hu, Wu, K), device=dy.device, dtype=torch.float32)
    if TRAIN_CONV == "f16x3" and K % 32 == 0 and Ci % 4 == 0 and lift is not None and UPSAMPLE_SPLIT_FUSED:   # (conv_train's split-kernel conditions)
        xs = torch.empty_like(src)
        check(lib().dlip_upsample_zero_split_f32(ptr(dy), ptr(xs), ptr(lift), N, Ho, Wo, Hu, Wu, K, sh, sw, stream_handle()),
              "dlip_upsample_zero_split_f32")
        return src, xs
    check(lib().dlip_upsample_zero_f32(ptr(dy), ptr(src), N, Ho, Wo, Hu, Wu, K, sh, sw, stream_handle()), "dlip_upsample_zero_f32")
    return src, None


UPSAMPLE_SPLIT_FUSED = __import__("os").environ.get("DLIP_UPSAMPLE_SPLIT", "1") != "0"


class ConvTrainFn(Function):
    """nn.Conv2d / nn.Conv1d (H = 1) on NHWC activations, raw (unfolded) weights in the reference layout
    [K,C,R,S]: forward = the fp32 implicit-GEMM kernel; backward = bias column sum, DATA gradient = the same
    kernel on the flipped / transposed weights (over the zero-inserted dY when strided), WEIGHT gradient = one
    GEMM over the rows of all taps gathered side by side (resnet.py:9-16,55-69; tcn.py:39-41,94,101)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, dil, pending=None):
        """``pending`` = (z, mean, invstd, gamma, beta, slopes) of the BatchNorm + PReLU in front (BNPReLUFn defer): x's values were never
        written -- the operand producer applies the normalisation and the slopes to z on load; a convolution that cannot take them
        on load writes them first."""
        x = x.contiguous()
        N, H, W, Cx = x.shape
        K, Cw, R, S = weight.shape
        if Cw != Cx or Cx % 4 or K % 4:
            raise ValueError(f"conv train path: channels must match and be multiples of 4 (x {Cx}, weight {Cw}, out {K})")
        # One read of x writes BOTH of its split images: the forward convolution's operand and the weight gradient's (kept for the
        # backward instead of x itself -- the data gradient does not need x).
        fused = WGRAD == "conv" and TRAIN_CONV == "f16x3" and R * S > 1 and Cx % 32 == 0 and K % 4 == 0 and ctx.needs_input_grad[1]
        xT = xs = None
        on_load = pending is not None and fused and Cx % 64 == 0 and WGRAD_SLICE_MAJOR
        if pending is not None and not on_load:
            from .autograd import materialize_pending
            materialize_pending(x, pending)
        if fused:
            xT, xs = wgrad_image(x, None, also_nhwc_split=True, bn=pending if on_load else None)
        y = conv_train(x, None, bias.contiguous() if bias is not None else None, stride, pad, dil, w_ref=weight, xs_ready=xs)
        ctx.save_for_backward(xT if fused else x, weight)
        ctx.cfg = (stride, pad, dil, bias is not None)
        ctx.x_shape, ctx.fused = (N, H, W, Cx), fused
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        (sh, sw), (ph, pw), (dh, dw), has_bias = ctx.cfg
        dy = dy.contiguous()
        N, H, W, Cx = ctx.x_shape
        K, _, R, S = weight.shape
        _, Ho, Wo, _ = dy.shape
        J = N * Ho * Wo
        dev = dy.device
        dz_rows = dy.view(J, K)
        dbias = _colsum_rows(dz_rows) if has_bias and ctx.needs_input_grad[2] else None
        dx = None
        lift = pow2_lift(dy) if (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]) else None   # zero insertion does not change max|dy|
        if ctx.fused:
            # (x here is the weight gradient's image of the layer input, formed in the forward.)  One read of dy writes its image AND,
            # for a stride-1 layer, the lifted split operand of the data-gradient convolution.
            want_w, want_x = ctx.needs_input_grad[1], ctx.needs_input_grad[0]
            dense = sh == 1 and sw == 1 and K % 32 == 0
            gT = dys = None
            if want_w:
                gT, dys = wgrad_image(dy, lift, also_nhwc_split=want_x and dense)
            if want_x:
                src = dy
                if not dense:
                    src, dys = _upsample_zero(dy, H + 2 * ph - dh * (R - 1), W + 2 * pw - dw * (S - 1), sh, sw, lift, weight.shape[1])
                dx = conv_train(src, None, None, (1, 1), (dh * (R - 1) - ph, dw * (S - 1) - pw), (dh, dw), lift=True, scale2=lift,
                                w_ref=weight, transposed=True, xs_ready=dys)
            dweight = wgrad_as_conv(None, None, R, S, (sh, sw), (ph, pw), (dh, dw), scale2=lift, xT=x, gT=gT) if want_w else None
            return dx, dweight, dbias, None, None, None, None
        if ctx.needs_input_grad[0]:
            src, dys = dy, None
            if sh != 1 or sw != 1:
                src, dys = _upsample_zero(dy, H + 2 * ph - dh * (R - 1), W + 2 * pw - dw * (S - 1), sh, sw, lift, weight.shape[1])
            dx = conv_train(src, None, None, (1, 1), (dh * (R - 1) - ph, dw * (S - 1) - pw), (dh, dw), lift=True, scale2=lift,
                            w_ref=weight, transposed=True, xs_ready=dys)
        dweight = None
        if ctx.needs_input_grad[1]:
            # all taps side by side in ONE [J, RS*C] matrix -> one GEMM with RS*C output rows (RS times the
            # tiles of a per-tap GEMM: a 64-channel layer would otherwise be a single 64x64 tile)
            taps = R * S
            if Cx % 4 == 0 and K % 4 == 0:
                return dx, wgrad_conv(x, dy, R, S, (sh, sw), (ph, pw), (dh, dw), scale2=lift), dbias, None, None, None, None
            rows = torch.empty((J, taps * Cx), device=dev, dtype=torch.float32)
            for t in range(taps):
                r, s = divmod(t, S)
                check(lib().dlip_tap_gather_f32(ptr(x), rows.data_ptr() + 4 * t * Cx, N, H, W, Cx, Cx, Ho, Wo, sh, sw,
                                                r * dh - ph, s * dw - pw, taps * Cx, stream_handle()), "dlip_tap_gather_f32")
            dwt = wgrad_gemm(dz_rows, lambda _: rows, 1).view(taps, Cx, K)  # [RS, C, K]
            dweight = _permute3(dwt, (2, 1, 0)).view(K, Cx, R, S)
        return dx, dweight, dbias, None, None, None, None


class StemConvTrainFn(Function):
    """Conv3d(1,64,(5,7,7),(1,2,2),(2,3,3), bias=False) on [B,T,H,W] (model.py:82) -> [(B T),H/2,W/2,64]:
    forward = the split-fp16 stem kernel on the CURRENT weights (split on the device; unit scale, zero shift, slope 1) -- the exact
    fp32 stem kernel with TRAIN_CONV = "f32"; backward = weight gradient only (the input is data): both GEMM operands written
    in one pass each (dlip_stem_wgrad_operand_f32 from the clip, dlip_wgrad_operand_f32 from dy), then one GEMM."""

    @staticmethod
    def forward(ctx, x, weight):
        x = x.contiguous()
        B, T, H, W = x.shape
        K = weight.shape[0]
        dev = x.device
        zero, one = const_vec(K, 0.0, dev), const_vec(K, 1.0, dev)
        if TRAIN_CONV == "f16x3" and K == 64:
            img = torch.empty((K * 296,), device=dev, dtype=torch.float32)
            wsc = torch.empty((K,), device=dev, dtype=torch.float32)
            check(lib().dlip_split_stem_weights_f32(ptr(weight.contiguous()), ptr(img), ptr(wsc), K, stream_handle()), "dlip_split_stem_weights_f32")
            y = ops.stem3d(x, img, zero, one, w_scale=wsc)
        else:
            wk = torch.zeros((248, K), device=dev, dtype=torch.float32)
            wk[:245].copy_(_permute3(weight.contiguous().view(1, K, 245), (0, 2, 1)).view(245, K))
            y = ops.stem3d(x, wk, zero, one)
        ctx.save_for_backward(x)
        ctx.K = K
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        B, T, H, W = x.shape
        K = ctx.K
        dev = x.device
        dy = dy.contiguous()
        Ho, Wo = H // 2, W // 2
        if WGRAD == "conv" and TRAIN_CONV == "f16x3" and K % 32 == 0:
            # as ONE 2-D convolution (see wgrad_as_conv): the five temporal taps are five shifted copies of the clip playing the
            # "images", the filter is the output gradient (44 x 44 taps, dilation 2): reads 0.6 GB where the reduction-major GEMM
            # operands are 2.2 GB written and read back
            N = B * T
            N32 = (N + 31) // 32 * 32
            scale2 = pow2_lift(dy)
            sm = 1 if WGRAD_SLICE_MAJOR else 0
            xT = torch.empty((5, H, W, N32), device=dev, dtype=torch.float32)
            check(lib().dlip_stem_wgrad_chwn_f32(ptr(x), ptr(xT), B, T, H, W, N32, sm, stream_handle()), "dlip_stem_wgrad_chwn_f32")
            gT = torch.empty((K, Ho, Wo, N32), device=dev, dtype=torch.float32)
            check(lib().dlip_wgrad_chwn_f32(ptr(dy), ptr(gT), N, Ho, Wo, K, K, N32, ptr(scale2), sm, None, stream_handle()), "dlip_wgrad_chwn_f32")
            inv = lift_inv(scale2, K)
            if sm:
                out = _wgrad_conv_launch(xT, gT, inv, 5, H, W, K, Ho, Wo, N32, (2, 2), (3, 3), (1, 1))   # [5, 8, 8, K] (even H: a spare row / column)
            else:
                out = ops.conv_nhwc(xT, gT, None, stride=(1, 1), pad=(3, 3), dil=(2, 2), w_scale=const_vec(K, 1.0, dev), x_split=True,
                                    post_scale=inv, post_shift=const_vec(K, 0.0, dev))
            out = out[:, :7, :7].contiguous()
            return None, _permute3(out.view(1, 245, K), (0, 2, 1)).view(K, 1, 5, 7, 7)
        J = B * T * Ho * Wo
        J32 = (J + 31) // 32 * 32
        if WGRAD_ODD_PITCH and (J32 // 32) % 2 == 0:
            J32 += 32
        scale2 = pow2_lift(dy)
        dzT_s = torch.empty((K, J32), device=dev, dtype=torch.float32)
        check(lib().dlip_wgrad_operand_f32(ptr(dy), ptr(dzT_s), J32, B * T, Ho, Wo, K, K, Ho, Wo, 1, 1, 1, 1, 1, 1, 0, 0, ptr(scale2), stream_handle()),
              "dlip_wgrad_operand_f32")
        xT_s = torch.empty((248, J32), device=dev, dtype=torch.float32)
        check(lib().dlip_stem_wgrad_operand_f32(ptr(x), ptr(xT_s), J32, B, T, H, W, stream_handle()), "dlip_stem_wgrad_operand_f32")
        inv = lift_inv(scale2, K)
        out = torch.empty((248, K), device=dev, dtype=torch.float32)
        ops.conv_nhwc(xT_s.view(1, 1, 248, J32), dzT_s.view(K, 1, 1, J32), None, w_scale=const_vec(K, 1.0, dev), x_split=True,
                      post_scale=inv, post_shift=const_vec(K, 0.0, dev), out=out.view(1, 1, 248, K))
        dweight = _permute3(out[:245].contiguous().view(1, 245, K), (0, 2, 1)).view(K, 1, 5, 7, 7)
        return None, dweight


class PReLUFn(Function):
    """nn.PReLU(C) on [..., C] channels-last; slope gradient = column sums of (x < 0 ? dy x : 0)."""

    @staticmethod
    def forward(ctx, x, slope):
        x = x.contiguous()
        C_ = x.shape[-1]
        M = x.numel() // C_
        y = torch.empty_like(x)
        check(lib().dlip_prelu_rows_fwd_f32(ptr(x), ptr(slope), ptr(y), M, C_, stream_handle()), "dlip_prelu_rows_fwd_f32")
        ctx.save_for_backward(x, slope)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, slope = ctx.saved_tensors
        dy = dy.contiguous()
        C_ = x.shape[-1]
        M = x.numel() // C_
        dx = torch.empty_like(x)
        terms = torch.empty_like(x)
        check(lib().dlip_prelu_rows_bwd_f32(ptr(dy), ptr(x), ptr(slope), ptr(dx), ptr(terms), M, C_, stream_handle()),
              "dlip_prelu_rows_bwd_f32")
        dslope = _colsum_rows(terms.view(M, C_)) if ctx.needs_input_grad[1] else None
        return dx, dslope


class BNPReLUFn(Function):
    """prelu(bn_train(x)) on [M, C] rows with per-channel slopes in the BatchNorm kernels' own passes
    (dlip_bn_prelu_rows_train_fwd/bwd_f32): no separate PReLU forward / backward pass and no column sum of slope terms."""

    @staticmethod
    def forward(ctx, x, gamma, beta, slope, running_mean, running_var, momentum, eps, nbt=None, defer=False):
        """``defer``: statistics only -- returns (y [values NOT written], mean, invstd): the convolution behind applies the
        normalisation and the slopes on load (ConvTrainFn ``pending``; autograd.TDNNBlockTrainFn does the same for the speech encoder)."""
        shape = tuple(x.shape)                                 # any channels-last shape [..., C]
        x = x.contiguous().view(-1, shape[-1])
        M, C_ = x.shape
        _lib.ensure_conv_workspace()      # (its ticket words: the finalize steps run in the passes' last workgroups)
        y = None if defer else torch.empty_like(x)
        mean = torch.empty((C_,), device=x.device, dtype=torch.float32)
        invstd = torch.empty_like(mean)
        ws = torch.empty((int(lib().dlip_bn_rows_chunks(M)) * C_ * 4,), device=x.device, dtype=torch.float64)
        check(lib().dlip_bn_prelu_rows_train_fwd_f32(ptr(x), ptr(gamma), ptr(beta), ptr(slope), ptr(y), ptr(mean), ptr(invstd),
                                                     ptr(running_mean), ptr(running_var), ptr(ws), M, C_, momentum, eps, ptr(nbt), stream_handle()),
              "dlip_bn_prelu_rows_train_fwd_f32")
        ctx.save_for_backward(x, gamma, beta, slope, mean, invstd)
        ctx.shape = shape
        if defer:
            y = torch.empty(shape, device=x.device, dtype=torch.float32)      # (address and shape only)
            ctx.mark_non_differentiable(mean, invstd)
            ctx.set_materialize_grads(False)
            return y, mean, invstd
        return y.view(shape)

    @staticmethod
    def backward(ctx, dy, *_unused):
        x, gamma, beta, slope, mean, invstd = ctx.saved_tensors
        M, C_ = x.shape
        _lib.ensure_conv_workspace()
        dy = dy.contiguous().view(M, C_)
        dx = torch.empty_like(x)
        dg, db, ds = (torch.empty_like(mean) for _ in range(3))
        ws = torch.empty((int(lib().dlip_bn_rows_chunks(M)) * C_ * 4,), device=x.device, dtype=torch.float64)
        lift = torch.empty((LIFT_WORDS,), device=x.device, dtype=torch.float32)
        check(lib().dlip_bn_prelu_rows_train_bwd_f32(ptr(dy), ptr(x), ptr(gamma), ptr(beta), ptr(slope), ptr(mean), ptr(invstd), ptr(dx),
                                                     ptr(dg), ptr(db), ptr(ds), ptr(ws), M, C_, ptr(lift), stream_handle()),
              "dlip_bn_prelu_rows_train_bwd_f32")
        dx = dx.view(ctx.shape)
        dx._dlip_lift = lift      # see autograd._bn_rows_bwd: travels with the tensor object the convolution backward receives
        return dx, dg, db, (ds if ctx.needs_input_grad[3] else None), None, None, None, None, None, None


class BNPReLUMaxPoolFn(Function):
    """maxpool(prelu(bn_train(x))) of the stem (model.py:83-85) on [N,H,W,C] WITHOUT the full-resolution tensors between the three
    (dlip_bn_prelu_maxpool_train_fwd/bwd_f32): the forward's one pass behind the statistics writes the pooled output and the argmax
    codes, the backward's two passes take the gradient behind the pooling per pixel from the codes and the pooled gradient."""

    @staticmethod
    def forward(ctx, x, gamma, beta, slope, running_mean, running_var, momentum, eps, nbt=None, fork=False):
        """``fork``: the pooled output TWICE (two tensor objects over one storage, as BNAddPReLUFn): the first block's convolution takes
        one, its shortcut the other, and their two gradients are added inside the backward's passes."""
        x = x.contiguous()
        N, H, W, C_ = x.shape
        _lib.ensure_conv_workspace()
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty((N, Ho, Wo, C_), device=x.device, dtype=torch.float32)
        idx = torch.empty((N, Ho, Wo, C_ // 4), device=x.device, dtype=torch.int32)
        mean = torch.empty((C_,), device=x.device, dtype=torch.float32)
        invstd = torch.empty_like(mean)
        ws = torch.empty((int(lib().dlip_bn_rows_chunks(N * H * W)) * C_ * 2,), device=x.device, dtype=torch.float64)
        check(lib().dlip_bn_prelu_maxpool_train_fwd_f32(ptr(x), ptr(gamma), ptr(beta), ptr(slope), ptr(y), idx.data_ptr(), ptr(mean), ptr(invstd),
                                                        ptr(running_mean), ptr(running_var), ptr(ws), N, H, W, C_, momentum, eps, ptr(nbt),
                                                        stream_handle()), "dlip_bn_prelu_maxpool_train_fwd_f32")
        ctx.save_for_backward(x, gamma, beta, slope, mean, invstd, idx)
        ctx.set_materialize_grads(False)
        return (y, y.detach()) if fork else y

    @staticmethod
    def backward(ctx, dy, dy2=None):
        x, gamma, beta, slope, mean, invstd, idx = ctx.saved_tensors
        N, H, W, C_ = x.shape
        _lib.ensure_conv_workspace()
        if dy is None:
            dy, dy2 = dy2, None
        dy = dy.contiguous()
        dy2 = dy2.contiguous() if dy2 is not None else None
        dx = torch.empty_like(x)
        dg, db, ds = (torch.empty_like(mean) for _ in range(3))
        ws = torch.empty((int(lib().dlip_bn_rows_chunks(N * H * W)) * C_ * 4,), device=x.device, dtype=torch.float64)
        lift = torch.empty((LIFT_WORDS,), device=x.device, dtype=torch.float32)
        check(lib().dlip_bn_prelu_maxpool_train_bwd_f32(ptr(dy), ptr(dy2), idx.data_ptr(), ptr(x), ptr(gamma), ptr(beta), ptr(slope), ptr(mean), ptr(invstd),
                                                        ptr(dx), ptr(dg), ptr(db), ptr(ds), ptr(ws), N, H, W, C_, ptr(lift), stream_handle()),
              "dlip_bn_prelu_maxpool_train_bwd_f32")
        dx._dlip_lift = lift
        return dx, dg, db, (ds if ctx.needs_input_grad[3] else None), None, None, None, None, None, None


class AddPReLUFn(Function):
    """prelu(a + b) with per-channel slopes in one launch (the end of a residual block); the sum is kept for the backward, which is
    PReLU's (the same gradient flows to a and to b)."""

    @staticmethod
    def forward(ctx, a, b, slope):
        a, b = a.contiguous(), b.contiguous()
        C_ = a.shape[-1]
        M = a.numel() // C_
        s_ = torch.empty_like(a)
        y = torch.empty_like(a)
        check(lib().dlip_add_prelu_rows_fwd_f32(ptr(a), ptr(b), ptr(slope), ptr(s_), ptr(y), M, C_, stream_handle()), "dlip_add_prelu_rows_fwd_f32")
        ctx.save_for_backward(s_, slope)
        return y

    @staticmethod
    def backward(ctx, dy):
        s_, slope = ctx.saved_tensors
        dy = dy.contiguous()
        C_ = s_.shape[-1]
        M = s_.numel() // C_
        dx = torch.empty_like(s_)
        terms = torch.empty_like(s_)
        check(lib().dlip_prelu_rows_bwd_f32(ptr(dy), ptr(s_), ptr(slope), ptr(dx), ptr(terms), M, C_, stream_handle()), "dlip_prelu_rows_bwd_f32")
        dslope = _colsum_rows(terms.view(M, C_)) if ctx.needs_input_grad[2] else None
        return dx, dx, dslope


class BNAddPReLUFn(Function):
    """The end of a BasicBlock under model.train(): prelu(bn2(x) + residual) (resnet.py:62-69) as dlip_bn_add_prelu_rows_train_fwd/bwd_f32
    -- bn2's output is never stored, the backward's first pass replaces four.  ``fork``: return the output TWICE (two tensor objects
    over one storage) -- the next block's first convolution takes one, its shortcut the other -- so that the two gradients arrive here
    separately and are added inside the backward's first pass instead of by a launch of autograd's."""

    @staticmethod
    def forward(ctx, x, res, gamma, beta, slope, running_mean, running_var, momentum, eps, nbt, fork):
        shape = tuple(x.shape)
        x = x.contiguous().view(-1, shape[-1])
        res = res.contiguous().view(-1, shape[-1])
        M, C_ = x.shape
        _lib.ensure_conv_workspace()
        s_ = torch.empty_like(x)
        y = torch.empty_like(x)
        mean = torch.empty((C_,), device=x.device, dtype=torch.float32)
        invstd = torch.empty_like(mean)
        ws = torch.empty((int(lib().dlip_bn_rows_chunks(M)) * C_ * 2,), device=x.device, dtype=torch.float64)
        check(lib().dlip_bn_add_prelu_rows_train_fwd_f32(ptr(x), ptr(res), ptr(gamma), ptr(beta), ptr(slope), ptr(s_), ptr(y), ptr(mean), ptr(invstd),
                                                         ptr(running_mean), ptr(running_var), ptr(ws), M, C_, momentum, eps, ptr(nbt),
                                                         stream_handle()), "dlip_bn_add_prelu_rows_train_fwd_f32")
        ctx.save_for_backward(x, s_, gamma, beta, slope, mean, invstd)
        ctx.shape = shape
        ctx.set_materialize_grads(False)
        y = y.view(shape)
        return (y, y.detach()) if fork else y

    @staticmethod
    def backward(ctx, dy, dy2=None):
        x, s_, gamma, beta, slope, mean, invstd = ctx.saved_tensors
        M, C_ = x.shape
        _lib.ensure_conv_workspace()
        if dy is None:
            dy, dy2 = dy2, None
        if dy is None:
            dy = torch.zeros_like(x)
        dy = dy.contiguous()
        dy2 = dy2.contiguous() if dy2 is not None else None
        dres = torch.empty_like(x)
        dx = torch.empty_like(x)
        dg, db, ds = (torch.empty_like(mean) for _ in range(3))
        ws = torch.empty((int(lib().dlip_bn_rows_chunks(M)) * C_ * 4,), device=x.device, dtype=torch.float64)
        lift = torch.empty((LIFT_WORDS,), device=x.device, dtype=torch.float32)
        check(lib().dlip_bn_add_prelu_rows_train_bwd_f32(ptr(dy), ptr(dy2), ptr(s_), ptr(x), ptr(gamma), ptr(beta), ptr(slope), ptr(mean), ptr(invstd),
                                                         ptr(dres), ptr(dx), ptr(dg), ptr(db), ptr(ds), ptr(ws), M, C_, ptr(lift), stream_handle()),
              "dlip_bn_add_prelu_rows_train_bwd_f32")
        dx = dx.view(ctx.shape)
        dx._dlip_lift = lift
        return dx, dres.view(ctx.shape), dg, db, (ds if ctx.needs_input_grad[4] else None), None, None, None, None, None, None


class MaxPoolFn(Function):
    """MaxPool3d((1,3,3),(1,2,2),(0,1,1)) on [(B T),H,W,C] (model.py:85).  The forward records each maximum's tap as one byte, the
    backward reads those instead of re-scanning the windows of x (and x itself is not kept alive for it)."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        N, H, W, C_ = x.shape
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty((N, Ho, Wo, C_), device=x.device, dtype=torch.float32)
        idx = torch.empty((N, Ho, Wo, C_ // 4), device=x.device, dtype=torch.int32)
        check(lib().dlip_maxpool3x3s2_idx_f32(ptr(x), ptr(y), idx.data_ptr(), N, H, W, C_, stream_handle()), "dlip_maxpool3x3s2_idx_f32")
        ctx.save_for_backward(idx)
        ctx.shape = (N, H, W, C_)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        N, H, W, C_ = ctx.shape
        dx = torch.empty(ctx.shape, device=dy.device, dtype=torch.float32)
        check(lib().dlip_maxpool3x3s2_bwd_idx_f32(idx.data_ptr(), ptr(dy.contiguous()), ptr(dx), N, H, W, C_, stream_handle()),
              "dlip_maxpool3x3s2_bwd_idx_f32")
        return dx


class AvgPoolFn(Function):
    """AdaptiveAvgPool2d(1) + flatten on [N,H,W,C] -> [N,C] (resnet.py:83,125-126)."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        ctx.shape = tuple(x.shape)
        return ops.avgpool(x)

    @staticmethod
    def backward(ctx, dy):
        N, H, W, C_ = ctx.shape
        dx = torch.empty(ctx.shape, device=dy.device, dtype=torch.float32)
        check(lib().dlip_row_broadcast_f32(ptr(dy.contiguous()), None, ptr(dx), N, H * W, C_, 1.0 / (H * W), stream_handle()),
              "dlip_row_broadcast_f32")
        return dx


class TimeMeanFn(Function):
    """_average_batch (model.py:16-17): mean over t < length of [B,T,C] -> [B,C]."""

    @staticmethod
    def forward(ctx, x, lengths):
        x = x.contiguous()
        ctx.shape = tuple(x.shape)
        ctx.save_for_backward(lengths)
        return ops.time_mean(x, lengths)

    @staticmethod
    def backward(ctx, dy):
        (lengths,) = ctx.saved_tensors
        B, T, C_ = ctx.shape
        dx = torch.empty(ctx.shape, device=dy.device, dtype=torch.float32)
        check(lib().dlip_row_broadcast_f32(ptr(dy.contiguous()), ptr(lengths), ptr(dx), B, T, C_, 0.0, stream_handle()),
              "dlip_row_broadcast_f32")
        return dx, None


class DropoutFn(Function):
    """nn.Dropout (tcn.py:80,85) from the uniform draws u themselves: y = u >= p ? x / (1 - p) : 0 (dlip_dropout_keep_f32), the backward
    the same launch on dy.  torch's generator makes the draws (graph-safe Philox offsets); the keep test, the cast and the product
    were three more launches per dropout."""

    @staticmethod
    def forward(ctx, x, u, p):
        x = x.contiguous()
        ctx.save_for_backward(u)
        ctx.p = float(p)
        y = torch.empty_like(x)
        check(lib().dlip_dropout_keep_f32(ptr(x), ptr(u), ptr(y), x.numel(), ctx.p, 1.0 / (1.0 - ctx.p), stream_handle()), "dlip_dropout_keep_f32")
        return y

    @staticmethod
    def backward(ctx, dy):
        (u,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        check(lib().dlip_dropout_keep_f32(ptr(dy), ptr(u), ptr(dx), dy.numel(), ctx.p, 1.0 / (1.0 - ctx.p), stream_handle()), "dlip_dropout_keep_f32")
        return dx, None, None


class MulMaskFn(Function):
    """y = x * mask * scale: nn.Dropout with an explicit keep-mask (tcn.py:80,85) and, with mask = 1 and two
    calls, nothing else -- the add of the residual branches is torch's own tensor add."""

    @staticmethod
    def forward(ctx, x, mask, scale):
        x = x.contiguous()
        ctx.save_for_backward(mask)
        ctx.scale = scale
        y = torch.empty_like(x)
        check(lib().dlip_mul_mask_f32(ptr(x), ptr(mask), ptr(y), x.numel(), scale, stream_handle()), "dlip_mul_mask_f32")
        return y

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        check(lib().dlip_mul_mask_f32(ptr(dy), ptr(mask), ptr(dx), dy.numel(), ctx.scale, stream_handle()), "dlip_mul_mask_f32")
        return dx, None, None


# ------------------------------------------------------------------------------------------------------
# functional helpers used by deeplip_amd.video in train mode
# ------------------------------------------------------------------------------------------------------
def conv(x, weight, bias=None, stride=(1, 1), pad=(0, 0), dil=(1, 1), pending=None):
    """x NHWC, weight [K,C,R,S] (reference Conv2d layout) or [K,C,S] (Conv1d: R = 1).  ``pending``: batchnorm_prelu(..., defer=True)."""
    if weight.dim() == 3:
        weight = weight.unsqueeze(2)
    return ConvTrainFn.apply(x, weight, bias, tuple(stride), tuple(pad), tuple(dil), pending)


def batchnorm(x, bn):
    """Train-mode BatchNorm over all leading axes of a channels-last tensor; running stats updated in place
    (nn.BatchNorm1d/2d/3d: resnet.py:51,64,16; model.py:83; tcn.py:42)."""
    C_ = x.shape[-1]
    return BNRowsActFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps, 1.0, False,
                             bn.num_batches_tracked)       # (the counter is incremented by the launch that finishes the statistics)


# (round 5) conv1 -> bn1 + relu1 -> conv2: relu1's output is not stored, conv2's operand producer applies bn1 + relu1 on load
BN_ON_LOAD = __import__("os").environ.get("DLIP_BN_ON_LOAD", "1") != "0"


def batchnorm_prelu(x, bn, act, defer=False):
    """prelu(batchnorm(x)) of a channels-last tensor in train mode, fused when the activation has one slope per channel (or is a
    ReLU marker: slope 0, no parameter); a single shared slope keeps the two-step path (its gradient is a sum over channels).
    ``defer``: returns (y, pending) -- y's values are NOT written when pending is not None: hand both to conv(..., pending=pending)."""
    w = getattr(act, "weight", None)
    C_ = x.shape[-1]
    if w is not None and w.numel() != C_:
        y = prelu(batchnorm(x, bn), act)
        return (y, None) if defer else y
    if w is None:
        w = const_vec(C_, 0.0, x.device)
    w = w if w.is_contiguous() else w.contiguous()
    if defer and BN_ON_LOAD:
        y, mean, invstd = BNPReLUFn.apply(x, bn.weight, bn.bias, w, bn.running_mean, bn.running_var, bn.momentum, bn.eps, bn.num_batches_tracked, True)
        return y, (x.detach(), mean, invstd, bn.weight.detach(), bn.bias.detach(), w.detach())
    y = BNPReLUFn.apply(x, bn.weight, bn.bias, w, bn.running_mean, bn.running_var, bn.momentum, bn.eps, bn.num_batches_tracked)
    return (y, None) if defer else y


# (round 5) the stem's BatchNorm + PReLU + max-pool as one Function (False: round 4's three: BNPReLUFn, then MaxPoolFn)
STEM_BN_POOL_FUSED = __import__("os").environ.get("DLIP_STEM_BN_POOL", "1") != "0"


def batchnorm_prelu_maxpool(x, bn, act, fork=False):
    """maxpool(prelu(batchnorm(x))) of a channels-last [N,H,W,C] tensor in train mode (model.py:83-85).  ``fork``: a pair of tensors
    over the one output (see BNPReLUMaxPoolFn)."""
    w = getattr(act, "weight", None)
    C_ = x.shape[-1]
    if not STEM_BN_POOL_FUSED or x.dim() != 4 or (w is not None and w.numel() != C_):
        y = maxpool(batchnorm_prelu(x, bn, act))
        return (y, y) if fork else y
    if w is None:
        w = const_vec(C_, 0.0, x.device)
    return BNPReLUMaxPoolFn.apply(x, bn.weight, bn.bias, w if w.is_contiguous() else w.contiguous(), bn.running_mean, bn.running_var,
                                  bn.momentum, bn.eps, bn.num_batches_tracked, bool(fork))


# (round 5) the end of a BasicBlock as one Function (False: round 4's BNRowsActFn + AddPReLUFn)
BLOCK_TAIL_FUSED = __import__("os").environ.get("DLIP_BLOCK_TAIL", "1") != "0"


def batchnorm_add_prelu(x, bn, res, act, fork=False):
    """prelu(batchnorm(x) + res) in train mode (resnet.py:62-69).  ``fork``: a pair of tensors over the one output (BNAddPReLUFn)."""
    w = getattr(act, "weight", None)
    C_ = x.shape[-1]
    if not BLOCK_TAIL_FUSED or x.shape != res.shape or (w is not None and w.numel() != C_):
        y = add_prelu(batchnorm(x, bn), res, act)
        return (y, y) if fork else y
    if w is None:
        w = const_vec(C_, 0.0, x.device)
    return BNAddPReLUFn.apply(x, res, bn.weight, bn.bias, w if w.is_contiguous() else w.contiguous(), bn.running_mean, bn.running_var,
                              bn.momentum, bn.eps, bn.num_batches_tracked, bool(fork))


def prelu(x, act):
    """act: holders.PReLUParams (learnable per-channel slope) or a ReLU marker (slope 0, no parameter)."""
    w = getattr(act, "weight", None)
    C_ = x.shape[-1]
    if w is None:
        w = const_vec(C_, 0.0, x.device)
    elif w.numel() == 1:
        w = w.expand(C_)
    return PReLUFn.apply(x, w.contiguous() if not w.is_contiguous() else w)


def add_prelu(a, b, act):
    """prelu(a + b); act as in prelu()."""
    w = getattr(act, "weight", None)
    C_ = a.shape[-1]
    if w is None:
        w = const_vec(C_, 0.0, a.device)
    elif w.numel() == 1:
        w = w.expand(C_)
    if a.shape != b.shape:
        return prelu(a + b, act)
    return AddPReLUFn.apply(a, b, w.contiguous() if not w.is_contiguous() else w)


def maxpool(x):
    return MaxPoolFn.apply(x)


def avgpool(x):
    return AvgPoolFn.apply(x)


def time_mean(x, lengths):
    return TimeMeanFn.apply(x, lengths)


class ConcatChannelsFn(Function):
    """torch.cat(branches, dim=-1) of channels-last tensors under autograd as dlip_* launches (the multibranch TCN block's
    concatenation, tcn.py:96-108): forward copies each branch into its channel slice of one output, backward copies the
    slices of dy back out -- the strided row copy of dlip_tap_gather_f32 (one tap, identity gather, ldx / ldo row pitches)."""

    @staticmethod
    def forward(ctx, *xs):
        xs = [x.contiguous() for x in xs]
        lead = xs[0].shape[:-1]
        J = xs[0].numel() // xs[0].shape[-1]
        widths = [x.shape[-1] for x in xs]
        Ct = sum(widths)
        out = torch.empty(tuple(lead) + (Ct,), device=xs[0].device, dtype=torch.float32)
        off = 0
        for x, Cb in zip(xs, widths):
            check(lib().dlip_tap_gather_f32(ptr(x), out.data_ptr() + 4 * off, J, 1, 1, Cb, Cb, 1, 1, 1, 1, 0, 0, Ct, stream_handle()),
                  "dlip_tap_gather_f32")
            off += Cb
        ctx.widths = widths
        return out

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        Ct = dy.shape[-1]
        J = dy.numel() // Ct
        outs, off = [], 0
        for Cb in ctx.widths:
            g = torch.empty(tuple(dy.shape[:-1]) + (Cb,), device=dy.device, dtype=torch.float32)
            check(lib().dlip_tap_gather_f32(dy.data_ptr() + 4 * off, ptr(g), J, 1, 1, Cb, Ct, 1, 1, 1, 1, 0, 0, Cb, stream_handle()),
                  "dlip_tap_gather_f32")
            outs.append(g)
            off += Cb
        return tuple(outs)


def concat_channels(xs):
    return ConcatChannelsFn.apply(*xs)


class ChompConcatFn(Function):
    """The end of a multibranch TCN stage in one launch per branch: symmetric chomp (tcn.py:52-59: rows pad/2 .. pad/2 + T of the
    padded-length branch output [B,1,T+pad,nb]) AND concatenation along channels (tcn.py:96-108) -> [B,T,sum nb].  Backward: each
    branch's gradient is the slice of dy at its channels, placed at rows pad/2.. of a zero tensor -- the same strided row copy with
    a negative row offset (out-of-range rows read as zeros).  Replaces slice + contiguous + concat (two copies forward; a zero
    fill, a copy and a slice copy backward)."""

    @staticmethod
    def forward(ctx, T, *zs):
        zs = [z.contiguous() for z in zs]
        B = zs[0].shape[0]
        widths = [z.shape[-1] for z in zs]
        lens = [z.shape[2] for z in zs]                       # T + pad_j
        Ct = sum(widths)
        out = torch.empty((B, T, Ct), device=zs[0].device, dtype=torch.float32)
        ctx.cfg = (T, widths, lens)
        if TCN_FEWER_LAUNCHES and len(zs) <= 4 and all((L - T) % 2 == 0 for L in lens):
            _chomp_concat_launch(zs, lens, widths, out, B, T, 0)      # (round 5) every branch in one launch
            return out
        off = 0
        for z, Cb, L in zip(zs, widths, lens):
            check(lib().dlip_tap_gather_f32(ptr(z), out.data_ptr() + 4 * off, B, 1, L, Cb, Cb, 1, T, 1, 1, 0, (L - T) // 2, Ct, stream_handle()),
                  "dlip_tap_gather_f32")
            off += Cb
        return out

    @staticmethod
    def backward(ctx, dy):
        T, widths, lens = ctx.cfg
        dy = dy.contiguous()
        B, _, Ct = dy.shape
        if TCN_FEWER_LAUNCHES and len(widths) <= 4 and all((L - T) % 2 == 0 for L in lens):
            outs = [torch.empty((B, 1, L, Cb), device=dy.device, dtype=torch.float32) for Cb, L in zip(widths, lens)]
            _chomp_concat_launch(outs, lens, widths, dy, B, T, 1)
            return (None,) + tuple(outs)
        outs, off = [], 0
        for Cb, L in zip(widths, lens):
            g = torch.empty((B, 1, L, Cb), device=dy.device, dtype=torch.float32)
            check(lib().dlip_tap_gather_f32(dy.data_ptr() + 4 * off, ptr(g), B, 1, T, Cb, Ct, 1, L, 1, 1, 0, -((L - T) // 2), Cb, stream_handle()),
                  "dlip_tap_gather_f32")
            outs.append(g)
            off += Cb
        return (None,) + tuple(outs)


# (round 5) the MS-TCN stage's chomp + concatenation in one launch, its dropout in two (False: round 4's 3 + 4 launches; A/B runs)
TCN_FEWER_LAUNCHES = __import__("os").environ.get("DLIP_TCN_FEWER_LAUNCHES", "1") != "0"


def _chomp_concat_launch(branches, lens, widths, cat, B, T, backward):
    import ctypes as C
    n = len(branches)
    pa = (C.c_void_p * n)(*[b.data_ptr() for b in branches])
    la = (C.c_int32 * n)(*[int(v) for v in lens])
    wa = (C.c_int32 * n)(*[int(v) for v in widths])
    check(lib().dlip_chomp_concat_f32(pa, la, wa, n, ptr(cat), B, T, backward, stream_handle()), "dlip_chomp_concat_f32")


def chomp_concat(zs, T):
    return ChompConcatFn.apply(T, *zs)


def dropout(x, p: float, training: bool = True):
    if not training or p <= 0.0:
        return x
    u = torch.rand(x.shape, device=x.device)                     # torch's generator: the mask stream is build-owned; kept iff u >= p
    if not TCN_FEWER_LAUNCHES:
        return MulMaskFn.apply(x, (u >= p).float(), 1.0 / (1.0 - p))
    return DropoutFn.apply(x, u, p)


def stem_conv(x_bthw, weight):
    return StemConvTrainFn.apply(x_bthw, weight)

"""Thin tensor-level wrappers over the C ABI (include/deeplip_hip.h).

PyTorch is plumbing here: it owns the device memory (``torch.empty``) and the stream; every
arithmetic op is one ``dlip_*`` call on ``torch.cuda.current_stream()``.  Inputs must be fp32
CUDA tensors; nothing in this module computes on the CPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import ConvDesc, check, lib, ptr, stream_handle

Tensor = torch.Tensor

# Optional measurement hook (bench.py): an object with begin(kernel_name, flops) -> token and
# end(token), called around every MFMA kernel launch on the current stream.  None = no overhead.
LAUNCH_HOOK = None

# Set by deeplip_amd.plan.StepPlan while a step is warmed up / recorded: every result tensor then comes from
# the plan's arena (same blocks in both passes), so nothing is allocated while the stream is being captured.
ARENA = None


def _empty(shape, device, dtype=torch.float32) -> Tensor:
    a = ARENA
    if a is None:
        return torch.empty(shape, device=device, dtype=dtype)
    return a.take(tuple(shape), device, dtype)


def _req(t: Optional[Tensor], name: str, dtype=torch.float32) -> None:
    if t is None:
        return
    if not t.is_cuda:
        raise _lib.DeepLipHipError(f"{name}: expected a CUDA (ROCm) tensor; deeplip_amd has no CPU path")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous tensor")


def conv_out_size(n: int, k: int, stride: int, pad: int, dil: int) -> int:
    return (n + 2 * pad - dil * (k - 1) - 1) // stride + 1


def conv_nhwc(x: Tensor, w_krsc: Tensor, bias: Optional[Tensor] = None, *, stride=(1, 1), pad=(0, 0),
              dil=(1, 1), residual: Optional[Tensor] = None, slope: Optional[Tensor] = None,
              post_scale: Optional[Tensor] = None, post_shift: Optional[Tensor] = None,
              out: Optional[Tensor] = None, out_channel_offset: int = 0,
              in_channels: Optional[int] = None, in_channel_offset: int = 0,
              w_scale: Optional[Tensor] = None, x_split: bool = False, out_split: bool = False,
              stats: Optional[Tensor] = None) -> Tensor:
    """x [N,H,W,Cx] (NHWC), w [K,R,S,C] -> y [N,Ho,Wo,Ky].

    ``out`` / ``out_channel_offset`` write the K result channels into a slice of a wider tensor
    (concat-free multibranch blocks); ``in_channels`` / ``in_channel_offset`` read a slice.
    With ``w_scale`` given, ``w_krsc`` is the split (hi, lo) fp16 packing of packing.split_weights
    ([K,R,S,C32] float32 view) and the launch goes to the 3 x f16-MFMA kernel; there ``x_split`` says
    x and residual hold the split activation format (``split_pack``) and ``out_split`` asks for it."""
    for t, n in ((x, "x"), (w_krsc, "w"), (bias, "bias"), (residual, "residual"), (slope, "slope"),
                 (post_scale, "post_scale"), (post_shift, "post_shift"), (out, "out")):
        _req(t, n)
    _req(w_scale, "w_scale")
    N, H, W, Cx = x.shape
    K, R, S, Cw = w_krsc.shape
    if w_scale is not None:        # split weights are channel-padded to a multiple of 32
        if in_channels is None:
            in_channels = min(Cx - in_channel_offset, Cw)
        if (in_channels + 31) // 32 * 32 != Cw:
            raise ValueError(f"conv_nhwc: split weights have C32={Cw}, input slice has {in_channels} channels")
        Cw = in_channels
    Cin = Cw if in_channels is None else in_channels
    if Cin != Cw or in_channel_offset + Cin > Cx:
        raise ValueError(f"conv_nhwc: weight C={Cw} does not match input slice [{in_channel_offset}:+{Cin}] of {Cx}")
    Ho = conv_out_size(H, R, stride[0], pad[0], dil[0])
    Wo = conv_out_size(W, S, stride[1], pad[1], dil[1])
    if out is None:
        out = _empty((N, Ho, Wo, K), x.device)
        out_channel_offset = 0
    if tuple(out.shape[:3]) != (N, Ho, Wo) or out_channel_offset + K > out.shape[3]:
        raise ValueError(f"conv_nhwc: bad output tensor {tuple(out.shape)} for result {(N, Ho, Wo, K)}")
    if residual is not None and (tuple(residual.shape[:3]) != (N, Ho, Wo) or residual.shape[3] < K):
        raise ValueError(f"conv_nhwc: residual {tuple(residual.shape)} does not match output {(N, Ho, Wo, K)}")
    for v, n in ((bias, "bias"), (slope, "slope"), (post_scale, "post_scale"), (post_shift, "post_shift")):
        if v is not None and v.numel() != K:
            raise ValueError(f"conv_nhwc: {n} has {v.numel()} elements, expected K={K}")
    if x_split or out_split:
        if w_scale is None:
            raise ValueError("conv_nhwc: the split activation format exists only on the f16x3 path (w_scale)")
        if x_split and (Cin % 32 or Cx % 32 or in_channel_offset % 32
                        or (residual is not None and (residual.shape[3] % 32 or K % 32))):
            raise ValueError("conv_nhwc: split input needs channel counts / offsets in multiples of 32")
        if out_split and (K % 32 or out.shape[3] % 32 or out_channel_offset % 32):
            raise ValueError("conv_nhwc: split output needs channel counts / offsets in multiples of 32")
    d = ConvDesc(N, H, W, Cin, K, R, S, stride[0], stride[1], pad[0], pad[1], dil[0], dil[1], Ho, Wo,
                 Cx, out.shape[3], residual.shape[3] if residual is not None else 0)
    xp = x.data_ptr() + 4 * in_channel_offset
    yp = out.data_ptr() + 4 * out_channel_offset
    hook = LAUNCH_HOOK
    if hook is not None:
        bm, bn = C.c_int32(), C.c_int32()
        mode = 0 if w_scale is None else (3 if x_split else 1)
        check(lib().dlip_conv_plan(C.byref(d), mode, C.byref(bm), C.byref(bn)), "dlip_conv_plan")
        kname = ("conv_igemm_f32_kernel", "conv_igemm_f16x3_kernel", "", "conv_igemm_f16x3_dma_kernel")[mode]
        kind = lib().dlip_conv_kernel_kind(C.byref(d)) if mode == 3 else 0
        if kind == 1:
            kname, bm.value, bn.value = "conv_win_f16x3_kernel", 128, (128 if K > 64 else 64)
        elif kind == 2:      # the rows kernel (its speech-encoder mode, or the general mode: 2-D filters / residual)
            kname = "conv_rows_f16x3_kernel"
        tok = hook.begin(f"{kname}<{bm.value},{bn.value}>", 2.0 * N * Ho * Wo * K * R * S * Cin)
    if stats is not None:
        # the convolution's epilogue also leaves the column sums of y (dlip_conv_nhwc_stats_f16x3): conv_stats_chunks(...) said it can
        _req(stats, "stats", torch.float64)
        if not (x_split and w_scale is not None) or out_split or residual is not None or post_scale is not None or in_channel_offset or out_channel_offset:
            raise ValueError("conv_nhwc(stats=...): split input, fp32 output, no residual / post-affine / channel slices")
        _lib.ensure_conv_workspace()
        check(lib().dlip_conv_nhwc_stats_f16x3(C.byref(d), xp, ptr(w_krsc), ptr(w_scale), ptr(bias), ptr(slope), yp, ptr(stats),
                                               stats.numel() * 8, stream_handle()), "dlip_conv_nhwc_stats_f16x3")
    elif w_scale is not None:
        if x_split:
            _lib.ensure_conv_workspace()
        check(lib().dlip_conv_nhwc_f16x3(C.byref(d), xp, ptr(w_krsc), ptr(w_scale), ptr(bias), ptr(residual),
                                         ptr(slope), ptr(post_scale), ptr(post_shift), yp,
                                         int(x_split) | 2 * int(out_split), stream_handle()),
              "dlip_conv_nhwc_f16x3")
    else:
        check(lib().dlip_conv_nhwc_f32(C.byref(d), xp, ptr(w_krsc), ptr(bias), ptr(residual), ptr(slope),
                                       ptr(post_scale), ptr(post_shift), yp, stream_handle()), "dlip_conv_nhwc_f32")
    if hook is not None:
        hook.end(tok)
    return out


def conv_stats_chunks(N, H, W, Cin, K, R, S, stride=(1, 1), pad=(0, 0), dil=(1, 1)) -> int:
    """How many partial rows conv_nhwc(..., stats=...) writes for this split-input, fp32-output launch (0: no statistics epilogue
    for the shape -- launch the plain convolution and let the BatchNorm make its own pass)."""
    Ho = conv_out_size(H, R, stride[0], pad[0], dil[0])
    Wo = conv_out_size(W, S, stride[1], pad[1], dil[1])
    d = ConvDesc(N, H, W, Cin, K, R, S, stride[0], stride[1], pad[0], pad[1], dil[0], dil[1], Ho, Wo, Cin, K, 0)
    return int(lib().dlip_conv_stats_chunks(C.byref(d)))


def conv2_nhwc(x: Tensor, x2: Tensor, w_split: Tensor, bias: Tensor, w_scale: Tensor, *, stride=(1, 1), pad=(0, 0),
               dil=(1, 1), stride2=(2, 2), residual: Optional[Tensor] = None, slope: Optional[Tensor] = None,
               out_split: bool = True) -> Tensor:
    """One reduction over two sources (dlip_conv2_nhwc_f16x3): y = act(conv(x, w[..., taps]) + conv1x1(x2[::s2], w[..., tail])
    + bias).  x [N,H,W,C], x2 [N,H2,W2,C2], both in the split activation format; ``w_split`` [K, R*S*C32 + C2] float32 view
    of the jointly scaled split weights (packing.pack_conv2d_shortcut).  The BasicBlock shortcut (resnet.py:13-17,62-66)."""
    for t, n in ((x, "x"), (x2, "x2"), (w_split, "w"), (bias, "bias"), (w_scale, "w_scale"), (residual, "residual"), (slope, "slope")):
        _req(t, n)
    N, H, W, Cx = x.shape
    N2, H2, W2, C2 = x2.shape
    K, RSC = w_split.shape
    R = S = 3 if (RSC - C2) == 9 * Cx else 1
    if N2 != N or RSC != R * S * Cx + C2 or Cx % 32 or C2 % 32:
        raise ValueError(f"conv2_nhwc: weights {tuple(w_split.shape)} do not match x {tuple(x.shape)} + x2 {tuple(x2.shape)}")
    Ho = conv_out_size(H, R, stride[0], pad[0], dil[0])
    Wo = conv_out_size(W, S, stride[1], pad[1], dil[1])
    if (Ho - 1) * stride2[0] >= H2 or (Wo - 1) * stride2[1] >= W2:
        raise ValueError("conv2_nhwc: the shortcut source does not cover the output grid")
    if out_split and K % 32:
        raise ValueError("conv2_nhwc: split output needs K % 32 == 0")
    out = _empty((N, Ho, Wo, K), x.device)
    if residual is not None and tuple(residual.shape) != (N, Ho, Wo, K):
        raise ValueError("conv2_nhwc: residual shape")
    d = ConvDesc(N, H, W, Cx, K, R, S, stride[0], stride[1], pad[0], pad[1], dil[0], dil[1], Ho, Wo, Cx, K,
                 K if residual is not None else 0)
    hook = LAUNCH_HOOK
    if hook is not None:
        bm, bn = C.c_int32(), C.c_int32()
        check(lib().dlip_conv_plan(C.byref(d), 3, C.byref(bm), C.byref(bn)), "dlip_conv_plan")
        kname = "conv_rows_f16x3_kernel" if lib().dlip_conv_kernel_kind(C.byref(d)) == 2 else "conv_igemm_f16x3_dma_kernel"
        tok = hook.begin(f"{kname}<{bm.value},{bn.value},dual>", 2.0 * N * Ho * Wo * K * (R * S * Cx + C2))
    _lib.ensure_conv_workspace()
    check(lib().dlip_conv2_nhwc_f16x3(C.byref(d), ptr(x), ptr(x2), H2, W2, C2, C2, stride2[0], stride2[1], ptr(w_split),
                                      ptr(w_scale), ptr(bias), ptr(residual), ptr(slope), None, None, ptr(out),
                                      1 | 2 * int(out_split), stream_handle()), "dlip_conv2_nhwc_f16x3")
    if hook is not None:
        hook.end(tok)
    return out


class Pooled:
    """Partial column sums of a pooled convolution (conv_pool) + what pool_finish needs to read them.  ``lengths`` (int32 CUDA
    [G] or None) with ``len_mul`` / ``len_add``: a ragged batch -- group g is valid for its first lengths[g] * len_mul + len_add rows."""
    __slots__ = ("partials", "M", "K", "tile_rows", "group_rows", "lengths", "len_mul", "len_add")

    def __init__(self, partials, M, K, tile_rows, group_rows, lengths=None, len_mul=1, len_add=0):
        self.partials, self.M, self.K, self.tile_rows, self.group_rows = partials, M, K, tile_rows, group_rows
        self.lengths, self.len_mul, self.len_add = lengths, len_mul, len_add


def lengths_i32(lengths, device, n: Optional[int] = None, lo: int = 1, hi: Optional[int] = None) -> Optional[Tensor]:
    """The length vector of a ragged batch as the int32 device tensor the kernels read.  A CUDA int32 tensor passes through
    untouched (what a recorded step plan needs: no host->device copy inside the step; its values are clamped on the device and
    are the caller's responsibility); host lists / arrays / CPU tensors are validated here -- ``n`` entries in [lo, hi]."""
    if lengths is None:
        return None
    if isinstance(lengths, Tensor) and lengths.is_cuda:
        if lengths.dtype != torch.int32:
            raise TypeError("lengths: a device tensor must be int32")
        if n is not None and lengths.numel() != n:
            raise ValueError(f"lengths: {lengths.numel()} entries for a batch of {n}")
        return lengths.contiguous()
    vals = [int(l) for l in (lengths.tolist() if hasattr(lengths, "tolist") else lengths)]
    if n is not None and len(vals) != n:
        raise ValueError(f"lengths: {len(vals)} entries for a batch of {n}")
    if vals and (min(vals) < lo or (hi is not None and max(vals) > hi)):
        raise ValueError(f"lengths: values must lie in [{lo}, {hi}] (got {min(vals)} .. {max(vals)})")
    return torch.tensor(vals, dtype=torch.int32).to(device)


def conv_pool_tile_rows(x: Tensor, w_krsc: Tensor, *, stride=(1, 1), pad=(0, 0), dil=(1, 1)) -> int:
    """Rows of the workgroup tile conv_pool would use for this launch (group_rows must be >= it)."""
    N, H, W, Cx = x.shape
    K, R, S, Cw = w_krsc.shape
    Ho = conv_out_size(H, R, stride[0], pad[0], dil[0])
    Wo = conv_out_size(W, S, stride[1], pad[1], dil[1])
    d = ConvDesc(N, H, W, Cx, K, R, S, stride[0], stride[1], pad[0], pad[1], dil[0], dil[1], Ho, Wo, Cx, K, 0)
    bm = C.c_int32()
    if int(lib().dlip_conv_pool_partial_bytes(C.byref(d), C.byref(bm))) <= 0:
        raise _lib.DeepLipHipError("dlip_conv_pool_partial_bytes failed")
    return bm.value


def conv_pool(x: Tensor, w_krsc: Tensor, bias: Optional[Tensor], w_scale: Tensor, group_rows: int, *, stride=(1, 1),
              pad=(0, 0), dil=(1, 1), residual: Optional[Tensor] = None, slope: Optional[Tensor] = None,
              post_scale: Optional[Tensor] = None, post_shift: Optional[Tensor] = None,
              lengths: Optional[Tensor] = None, len_mul: int = 1, len_add: int = 0) -> Pooled:
    """dlip_conv_pool_f16x3: the convolution of conv_nhwc (split-format x / residual) whose output is never written --
    per workgroup tile only the fp64 column sums of y and y^2, cut at the boundaries of consecutive ``group_rows``-row
    groups.  pool_finish turns them into group means (AdaptiveAvgPool + temporal mean, resnet.py:125-126 +
    train_fusion.py:348) or mean | std (MeanStdPooling, pooling.py:24-26).  ``lengths`` (int32 CUDA, one per group): a
    ragged batch -- only the first lengths[g] * len_mul + len_add rows of group g are summed (and counted by pool_finish)."""
    for t, n in ((x, "x"), (w_krsc, "w"), (bias, "bias"), (w_scale, "w_scale"), (residual, "residual"), (slope, "slope"),
                 (post_scale, "post_scale"), (post_shift, "post_shift")):
        _req(t, n)
    _req(lengths, "lengths", torch.int32)
    N, H, W, Cx = x.shape
    K, R, S, Cw = w_krsc.shape
    if Cx % 32 or Cw != Cx:
        raise ValueError("conv_pool: split-format input with C % 32 == 0 and matching split weights required")
    Ho = conv_out_size(H, R, stride[0], pad[0], dil[0])
    Wo = conv_out_size(W, S, stride[1], pad[1], dil[1])
    if residual is not None and (tuple(residual.shape[:3]) != (N, Ho, Wo) or residual.shape[3] != K or K % 32):
        raise ValueError("conv_pool: residual shape")
    d = ConvDesc(N, H, W, Cx, K, R, S, stride[0], stride[1], pad[0], pad[1], dil[0], dil[1], Ho, Wo, Cx, K,
                 K if residual is not None else 0)
    bm = C.c_int32()
    nbytes = int(lib().dlip_conv_pool_partial_bytes(C.byref(d), C.byref(bm)))
    if nbytes <= 0:
        raise _lib.DeepLipHipError("dlip_conv_pool_partial_bytes failed")
    if group_rows < bm.value:
        raise ValueError(f"conv_pool: group_rows {group_rows} < tile rows {bm.value}; use the unfused path")
    part = _empty((nbytes // 8,), x.device, torch.float64)
    hook = LAUNCH_HOOK
    if hook is not None:
        # the rows kernel (pooled launches go there only when it is forced: dlip_debug_set) reports HALF its tile as partial rows
        rows = residual is None and _lib.DEBUG.get(_lib.DBG_ROWS, -1) > 0 and lib().dlip_conv_kernel_kind(C.byref(d)) == 2
        tok = hook.begin(f"conv_rows_f16x3_kernel<{2 * bm.value},256,pool>" if rows else f"conv_igemm_f16x3_dma_kernel<{bm.value},128,pool>",
                         2.0 * N * Ho * Wo * K * R * S * Cx)
    _lib.ensure_conv_workspace()
    check(lib().dlip_conv_pool_f16x3(C.byref(d), ptr(x), ptr(w_krsc), ptr(w_scale), ptr(bias), ptr(residual), ptr(slope),
                                     ptr(post_scale), ptr(post_shift), ptr(part), nbytes, group_rows, ptr(lengths), len_mul, len_add,
                                     stream_handle()), "dlip_conv_pool_f16x3")
    if hook is not None:
        hook.end(tok)
    M = N * Ho * Wo
    if lengths is not None and lengths.numel() != (M + group_rows - 1) // group_rows:
        raise ValueError(f"conv_pool: {lengths.numel()} lengths for {(M + group_rows - 1) // group_rows} groups")
    return Pooled(part, M, K, bm.value, group_rows, lengths, len_mul, len_add)


def pool_finish(p: Pooled, mode: str = "mean", out_split: bool = False) -> Tensor:
    """Pooled partial sums -> [G,K] group means ('mean') or [G,2K] mean | unbiased std ('meanstd'; ``out_split``: [G, 2K
    rounded up to 32] in the split activation format)."""
    G = (p.M + p.group_rows - 1) // p.group_rows
    if mode == "mean":
        y = _empty((G, p.K), p.partials.device)
    else:
        y = _empty((G, (2 * p.K + 31) // 32 * 32 if out_split else 2 * p.K), p.partials.device)
    check(lib().dlip_pool_finish_f32(ptr(p.partials), p.M, p.K, p.tile_rows, p.group_rows, ptr(p.lengths), p.len_mul, p.len_add,
                                     0 if mode == "mean" else 1, int(out_split), ptr(y), stream_handle()), "dlip_pool_finish_f32")
    return y


def linear(x: Tensor, w_kc: Tensor, bias: Optional[Tensor] = None, **kw) -> Tensor:
    """x [M,C] @ w [K,C]^T (+ epilogue) -> [M,K], through the same implicit-GEMM kernel."""
    M, Cx = x.shape
    K, Cw = w_kc.shape
    y = conv_nhwc(x.view(1, 1, M, Cx), w_kc.view(K, 1, 1, Cw), bias, **kw)
    return y.view(M, K)


def conv1d_ntc(x: Tensor, w_ksc: Tensor, bias: Optional[Tensor] = None, *, dilation: int = 1, pad: int = 0,
               **kw) -> Tensor:
    """x [B,T,C] (time-major, channels-last), w [K,S,C] -> [B,T',K]."""
    B, T, Cx = x.shape
    K, S, Cw = w_ksc.shape
    res = kw.pop("residual", None)
    out = kw.pop("out", None)
    if res is not None:
        res = res.view(B, 1, res.shape[1], res.shape[2])
    if out is not None:
        out = out.view(B, 1, out.shape[1], out.shape[2])
    y = conv_nhwc(x.view(B, 1, T, Cx), w_ksc.view(K, 1, S, Cw), bias, dil=(1, dilation), pad=(0, pad),
                  residual=res, out=out, **kw)
    return y.view(B, y.shape[2], y.shape[3])


def stem3d(x_bthw: Tensor, w_248xk: Tensor, bias: Tensor, slope: Optional[Tensor],
           w_scale: Optional[Tensor] = None) -> Tensor:
    """x [B,T,H,W] -> [(B*T), H/2, W/2, 64] (Conv3d 5x7x7 + folded BN + PReLU/ReLU).  With ``w_scale``
    the weights are the split-fp16 image of packing.split_stem_weights (3 x f16 MFMA kernel)."""
    for t, n in ((x_bthw, "x"), (w_248xk, "w"), (bias, "bias"), (slope, "slope"), (w_scale, "w_scale")):
        _req(t, n)
    B, T, H, W = x_bthw.shape
    if w_scale is not None:
        y = _empty((B * T, H // 2, W // 2, 64), x_bthw.device)
        hook = LAUNCH_HOOK
        if hook is not None:
            tok = hook.begin("stem3d_f16x3_kernel", 2.0 * B * T * (H // 2) * (W // 2) * 64 * 245)
        check(lib().dlip_stem3d_bn_act_f16x3(ptr(x_bthw), ptr(w_248xk), ptr(w_scale), ptr(bias), ptr(slope), ptr(y),
                                             B, T, H, W, 64, stream_handle()), "dlip_stem3d_bn_act_f16x3")
        if hook is not None:
            hook.end(tok)
        return y
    K = w_248xk.shape[1]
    if w_248xk.shape[0] != 248:
        raise ValueError("stem3d: weights must be packed [248, K]")
    y = _empty((B * T, H // 2, W // 2, K), x_bthw.device)
    hook = LAUNCH_HOOK
    if hook is not None:
        tok = hook.begin("stem3d_f32_kernel", 2.0 * B * T * (H // 2) * (W // 2) * K * 245)
    check(lib().dlip_stem3d_bn_act_f32(ptr(x_bthw), ptr(w_248xk), ptr(bias), ptr(slope), ptr(y), B, T, H, W, K,
                                       stream_handle()), "dlip_stem3d_bn_act_f32")
    if hook is not None:
        hook.end(tok)
    return y


def stem3d_pool(x_bthw: Tensor, w_img: Tensor, bias: Tensor, slope: Optional[Tensor], w_scale: Tensor,
                lengths: Optional[Tensor] = None) -> Tensor:
    """x [B,T,H,W] -> [(B*T), Hp, Wp, 64] in the split activation format: Conv3d 5x7x7 + folded BN +
    PReLU/ReLU + MaxPool3d((1,3,3),(1,2,2),(0,1,1)) in one kernel (split-fp16 weights image only).  ``lengths`` (int32 CUDA [B]):
    a zero-padded ragged batch (dataset.py:123-139) -- frames t >= lengths[b] are read as zeros whatever x holds there."""
    for t, n in ((x_bthw, "x"), (w_img, "w"), (bias, "bias"), (slope, "slope"), (w_scale, "w_scale")):
        _req(t, n)
    _req(lengths, "lengths", torch.int32)
    B, T, H, W = x_bthw.shape
    if lengths is not None and lengths.numel() != B:
        raise ValueError(f"stem3d_pool: {lengths.numel()} lengths for {B} clips")
    Ho, Wo = H // 2, W // 2
    y = _empty((B * T, (Ho - 1) // 2 + 1, (Wo - 1) // 2 + 1, 64), x_bthw.device)
    ws = _empty((int(lib().dlip_stem3d_pool_workspace_bytes(B, T, H, W)) // 4,), x_bthw.device)   # the clip as (hi, lo) pairs
    hook = LAUNCH_HOOK
    if hook is not None:
        tok = hook.begin("stem3d_pool_f16x3_kernel", 2.0 * B * T * Ho * Wo * 64 * 245)
    check(lib().dlip_stem3d_pool_f16x3(ptr(x_bthw), ptr(lengths), ptr(ws), ptr(w_img), ptr(w_scale), ptr(bias), ptr(slope), ptr(y),
                                       B, T, H, W, 64, stream_handle()), "dlip_stem3d_pool_f16x3")
    if hook is not None:
        hook.end(tok)
    return y


def center_crop_origin(size: int, crop: int) -> int:
    """CenterCrop's offset (models/video_models/preprocess.py:89-90): ``int(round(w - tw) / 2.)`` -- the round() is of the
    integer margin (a no-op), the division's result is truncated: FLOOR of half the margin, as dlip_crop_normalize_u8 does.
    (Round 3 had int(round(margin / 2)), Python's round-half-even: one pixel off for margins of 3, 7, 11 ... -- 91- or
    95-pixel frames.)"""
    return (size - crop) // 2


def draw_clip_params(n_clips: int, Hs: int, Ws: int, crop: int = 88, flip_ratio: float = 0.5, rng=None):
    """One (oy, ox, flip, 0) row per clip as the reference's train pipeline draws them (dataloaders.py:13-17): RandomCrop's
    ``random.randint(0, w - tw)`` then ``randint(0, h - th)`` (preprocess.py:110-111, inclusive bounds), then HorizontalFlip's
    ``random.random() < flip_ratio`` (:134), per clip, from ``rng`` (a ``random.Random``; the module-level generator when None --
    what the reference uses, seeded by the trainer).  Returns an int32 numpy array [n_clips, 4] (host; ``.to(device)`` it)."""
    import random as _random
    import numpy as _np
    r = rng if rng is not None else _random
    out = _np.zeros((n_clips, 4), dtype=_np.int32)
    for i in range(n_clips):
        ox = r.randint(0, Ws - crop)
        oy = r.randint(0, Hs - crop)
        out[i, 0], out[i, 1], out[i, 2] = oy, ox, int(r.random() < flip_ratio)
    return out


def stem3d_pool_u8(frames: Tensor, w_img: Tensor, bias: Tensor, slope: Optional[Tensor], w_scale: Tensor, crop: int = 88,
                   lengths: Optional[Tensor] = None, clip_params: Optional[Tensor] = None) -> Tensor:
    """uint8 frames [B,T,Hs,Ws] (gray) or [B,T,3,Hs,Ws] (RGB) -> [(B*T), Hp, Wp, 64] split format: centre crop + gray +
    (x/255 - 0.421)/0.165 inside the stem's pre-pass (dlip_stem3d_pool_u8_f16x3) -- no fp32 clip in HBM or over PCIe.
    ``lengths`` (int32 CUDA [B]): ragged batch, frames t >= lengths[b] become zeros of the normalised clip.  ``clip_params``
    (int32 CUDA [B,4] = (oy, ox, flip, 0) per clip): the train pipeline's RandomCrop + HorizontalFlip (preprocess.py:95-138)
    instead of the centre crop -- see draw_clip_params."""
    _req(frames, "frames", torch.uint8)
    for t, n in ((w_img, "w"), (bias, "bias"), (slope, "slope"), (w_scale, "w_scale")):
        _req(t, n)
    _req(lengths, "lengths", torch.int32)
    _req(clip_params, "clip_params", torch.int32)
    if lengths is not None and lengths.numel() != frames.shape[0]:
        raise ValueError(f"stem3d_pool_u8: {lengths.numel()} lengths for {frames.shape[0]} clips")
    if clip_params is not None and tuple(clip_params.shape) != (frames.shape[0], 4):
        raise ValueError(f"stem3d_pool_u8: clip_params must be [B,4] = (oy, ox, flip, 0), got {tuple(clip_params.shape)}")
    if frames.dim() not in (4, 5) or (frames.dim() == 5 and frames.shape[2] != 3):
        raise ValueError("stem3d_pool_u8: expected uint8 [B,T,H,W] or [B,T,3,H,W]")
    ch = 3 if frames.dim() == 5 else 1
    B, T = frames.shape[0], frames.shape[1]
    Hs, Ws = frames.shape[-2], frames.shape[-1]
    H = W = crop
    if Hs < H or Ws < W:
        raise ValueError(f"stem3d_pool_u8: frames {Hs}x{Ws} are smaller than the {crop}x{crop} crop")
    oy, ox = center_crop_origin(Hs, H), center_crop_origin(Ws, W)
    Ho, Wo = H // 2, W // 2
    y = _empty((B * T, (Ho - 1) // 2 + 1, (Wo - 1) // 2 + 1, 64), frames.device)
    ws = _empty((int(lib().dlip_stem3d_pool_workspace_bytes(B, T, H, W)) // 4,), frames.device)
    hook = LAUNCH_HOOK
    if hook is not None:
        tok = hook.begin("stem3d_pool_f16x3_kernel", 2.0 * B * T * Ho * Wo * 64 * 245)
    check(lib().dlip_stem3d_pool_u8_f16x3(ptr(frames), ch, Hs, Ws, oy, ox, ptr(clip_params), ptr(lengths), ptr(ws), ptr(w_img), ptr(w_scale), ptr(bias), ptr(slope),
                                          ptr(y), B, T, H, W, 64, stream_handle()), "dlip_stem3d_pool_u8_f16x3")
    if hook is not None:
        hook.end(tok)
    return y


def split_pack(x: Tensor) -> Tensor:
    """fp32 [..., C] -> split activation format (same shape / dtype container; C % 32 == 0)."""
    _req(x, "x")
    y = _empty(x.shape, x.device)
    Cc = x.shape[-1]
    check(lib().dlip_split_pack_f32(ptr(x), ptr(y), x.numel() // Cc, Cc, stream_handle()), "dlip_split_pack_f32")
    return y


def split_pack_scaled(x: Tensor, scale: Tensor) -> Tensor:
    """split_pack(x * scale[0]) -- ``scale``: a device scalar holding a power of two (the activation exponent of a model input whose
    gain lies outside the split format, packing.act_exponents; the multiplication is exact)."""
    _req(x, "x"); _req(scale, "scale")
    y = _empty(x.shape, x.device)
    Cc = x.shape[-1]
    check(lib().dlip_split_pack_scaled_f32(ptr(x), ptr(y), ptr(scale), x.numel() // Cc, Cc, stream_handle()), "dlip_split_pack_scaled_f32")
    return y


def channel_scale(x: Tensor, scale_vec: Tensor, zero_vec: Tensor) -> Tensor:
    """y[..., c] = x[..., c] * scale_vec[c] (dlip_affine_act_f32 with a unit slope): how an fp32 output leaves a model whose last
    split tensor carries an activation exponent -- scale_vec = 2^-e, exact."""
    C_ = x.shape[-1]
    return affine_act(x.reshape(-1, C_), scale_vec, zero_vec, slope=1.0).view(x.shape)


def split_unpack(x: Tensor) -> Tensor:
    """Inverse of split_pack (hi + lo in fp32)."""
    _req(x, "x")
    y = _empty(x.shape, x.device)
    Cc = x.shape[-1]
    check(lib().dlip_split_unpack_f32(ptr(x), ptr(y), x.numel() // Cc, Cc, stream_handle()), "dlip_split_unpack_f32")
    return y


def maxpool3x3s2(x: Tensor, out_split: bool = False) -> Tensor:
    _req(x, "x")
    N, H, W, Cc = x.shape
    y = _empty((N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, Cc), x.device)
    check(lib().dlip_maxpool3x3s2_nhwc_f32(ptr(x), ptr(y), N, H, W, Cc, int(out_split), stream_handle()),
          "dlip_maxpool3x3s2_nhwc_f32")
    return y


def avgpool(x: Tensor) -> Tensor:
    _req(x, "x")
    N, H, W, Cc = x.shape
    y = _empty((N, Cc), x.device)
    check(lib().dlip_avgpool_nhwc_f32(ptr(x), ptr(y), N, H * W, Cc, stream_handle()), "dlip_avgpool_nhwc_f32")
    return y


def time_mean(x: Tensor, lengths: Optional[Tensor] = None, len_add: int = 0) -> Tensor:
    """x [B,T,C] -> [B,C]; ``lengths`` int32 [B] masks the mean to t < len + len_add (model.py:16-17)."""
    _req(x, "x")
    _req(lengths, "lengths", torch.int32)
    B, T, Cc = x.shape
    y = _empty((B, Cc), x.device)
    check(lib().dlip_time_mean_f32(ptr(x), ptr(lengths), len_add, ptr(y), B, T, Cc, Cc, stream_handle()), "dlip_time_mean_f32")
    return y


def mask_frames(x: Tensor, lengths: Tensor) -> Tensor:
    """x [B,T,E] -> same with frames t >= lengths[b] zeroed (the padding of a ragged batch; dlip_mask_frames_f32)."""
    _req(x, "x")
    _req(lengths, "lengths", torch.int32)
    B, T, E = x.shape
    if lengths.numel() != B or E % 4:
        raise ValueError("mask_frames: one length per row of x [B,T,E], E % 4 == 0")
    y = _empty(x.shape, x.device)
    check(lib().dlip_mask_frames_f32(ptr(x), ptr(lengths), ptr(y), B, T, E, stream_handle()), "dlip_mask_frames_f32")
    return y


def l1_sum(w: Tensor) -> Tensor:
    """sum |w| as a 0-d device tensor (dlip_l1_sum_f32: fp64 accumulation, fixed order)."""
    _req(w, "w")
    out = torch.empty((1,), device=w.device, dtype=torch.float32)
    check(lib().dlip_l1_sum_f32(ptr(w.contiguous()), ptr(out), w.numel(), stream_handle()), "dlip_l1_sum_f32")
    return out[0]


def group_mean(x: Tensor, group_ptr: Tensor) -> Tensor:
    _req(x, "x")
    _req(group_ptr, "group_ptr", torch.int32)
    U = group_ptr.numel() - 1
    y = _empty((U, x.shape[1]), x.device)
    check(lib().dlip_group_mean_f32(ptr(x), ptr(group_ptr), ptr(y), U, x.shape[1], stream_handle()), "dlip_group_mean_f32")
    return y


def meanstd_pool(x: Tensor, out_split: bool = False, lengths: Optional[Tensor] = None, len_add: int = 0) -> Tensor:
    """x [B,T,C] -> [B,2C] = cat(mean_t, unbiased std_t)  (pooling.py:24-26).  ``out_split``: the
    result is [B, 2C rounded up to 32] in the split activation format (zero padded).  ``lengths`` (int32 CUDA [B]): ragged
    batch, utterance b's statistics cover its first lengths[b] + len_add frames."""
    _req(x, "x")
    _req(lengths, "lengths", torch.int32)
    B, T, Cc = x.shape
    if lengths is not None and lengths.numel() != B:
        raise ValueError(f"meanstd_pool: {lengths.numel()} lengths for {B} utterances")
    if Cc % 4:
        raise ValueError("meanstd_pool: C must be a multiple of 4")
    width = (2 * Cc + 31) // 32 * 32 if out_split else 2 * Cc
    y = _empty((B, width), x.device)
    check(lib().dlip_meanstd_pool_f32(ptr(x), ptr(lengths), len_add, ptr(y), B, T, Cc, int(out_split), stream_handle()), "dlip_meanstd_pool_f32")
    return y


def nct_to_ntc(x: Tensor, pad_to: Optional[int] = None, out_split: bool = False) -> Tensor:
    """[B,C,T] -> [B,T,Cp] channels-last, zero-padded to Cp; ``out_split``: in the split activation format (Cp % 32 == 0)."""
    _req(x, "x")
    B, Cc, T = x.shape
    Cp = Cc if pad_to is None else pad_to
    y = _empty((B, T, Cp), x.device)
    if out_split:
        check(lib().dlip_nct_to_ntc_split_f32(ptr(x), ptr(y), B, Cc, T, Cp, stream_handle()), "dlip_nct_to_ntc_split_f32")
        return y
    check(lib().dlip_nct_to_ntc_f32(ptr(x), ptr(y), B, Cc, T, Cp, stream_handle()), "dlip_nct_to_ntc_f32")
    return y


def ntc_to_nct(x: Tensor) -> Tensor:
    _req(x, "x")
    B, T, Cc = x.shape
    y = _empty((B, Cc, T), x.device)
    check(lib().dlip_ntc_to_nct_f32(ptr(x), ptr(y), B, T, Cc, stream_handle()), "dlip_ntc_to_nct_f32")
    return y


def ingest_rgb_u8(frames: Tensor) -> Tensor:
    """[B,T,3,H,W] uint8 -> [B,1,T,H,W] float32 normalised gray."""
    _req(frames, "frames", torch.uint8)
    B, T, three, H, W = frames.shape
    if three != 3:
        raise ValueError("ingest_rgb_u8: expected [B,T,3,H,W]")
    y = _empty((B, 1, T, H, W), frames.device)
    check(lib().dlip_ingest_rgb_u8(ptr(frames), ptr(y), B * T, H, W, stream_handle()), "dlip_ingest_rgb_u8")
    return y


def affine_act(x: Tensor, scale: Tensor, shift: Tensor, slope: float = 0.2, act_first: bool = False) -> Tensor:
    """[M,C]: lrelu(x*scale+shift) or lrelu(x)*scale+shift (eval BatchNorm1d + LeakyReLU)."""
    _req(x, "x"); _req(scale, "scale"); _req(shift, "shift")
    y = _empty(x.shape, x.device)
    check(lib().dlip_affine_act_f32(ptr(x), ptr(scale), ptr(shift), ptr(y), x.numel() // x.shape[-1], x.shape[-1],
                                    slope, int(act_first), stream_handle()), "dlip_affine_act_f32")
    return y


def znorm_cat(a: Optional[Tensor], v: Optional[Tensor], biased: bool = False) -> Tensor:
    _req(a, "a"); _req(v, "v")
    U = (a if a is not None else v).shape[0]
    Da = a.shape[1] if a is not None else 0
    Dv = v.shape[1] if v is not None else 0
    y = _empty((U, Da + Dv), (a if a is not None else v).device)
    check(lib().dlip_znorm_cat_f32(ptr(a), Da, ptr(v), Dv, ptr(y), U, int(biased), stream_handle()), "dlip_znorm_cat_f32")
    return y


def znorm_cat_pooled(a: Optional[Tensor], p: "Pooled", biased: bool = False) -> Tensor:
    """znorm_cat(a, pool_finish(p, 'mean')) in one launch (bit-identical to the two)."""
    _req(a, "a")
    U = (p.M + p.group_rows - 1) // p.group_rows
    Da = a.shape[1] if a is not None else 0
    if a is not None and a.shape[0] != U:
        raise ValueError(f"znorm_cat_pooled: {a.shape[0]} rows of a, {U} pooled groups")
    y = _empty((U, Da + p.K), p.partials.device)
    check(lib().dlip_znorm_cat_pooled_f32(ptr(a), Da, ptr(p.partials), p.M, p.K, p.tile_rows, p.group_rows, ptr(p.lengths), p.len_mul,
                                          p.len_add, ptr(y), U, int(biased), stream_handle()), "dlip_znorm_cat_pooled_f32")
    return y


def l2_normalize(x: Tensor, eps: float = 1e-12) -> Tensor:
    _req(x, "x")
    y = _empty(x.shape, x.device)
    check(lib().dlip_l2_normalize_f32(ptr(x), ptr(y), x.shape[0], x.shape[1], eps, stream_handle()), "dlip_l2_normalize_f32")
    return y


def pair_cosine(emb: Tensor, idx_a: Tensor, idx_b: Tensor, mode: int = 0, eps: float = 1e-8,
                weight: float = 1.0, out: Optional[Tensor] = None) -> Tensor:
    _req(emb, "emb"); _req(idx_a, "idx_a", torch.int32); _req(idx_b, "idx_b", torch.int32)
    n = idx_a.numel()
    acc = out is not None
    if out is None:
        out = _empty((n,), emb.device)
    _req(out, "out")
    check(lib().dlip_pair_cosine_f32(ptr(emb), emb.shape[0], emb.shape[1], ptr(idx_a), ptr(idx_b), ptr(out), n, mode,
                                     eps, weight, int(acc), stream_handle()), "dlip_pair_cosine_f32")
    return out


def logits_argmax(e: Tensor, W: Tensor, bias: Optional[Tensor] = None, cosine: bool = False) -> Tuple[Tensor, Tensor]:
    _req(e, "e"); _req(W, "W"); _req(bias, "bias")
    B, D = e.shape
    K = W.shape[0]
    logits = _empty((B, K), e.device)
    amax = _empty((B,), e.device, torch.int64)
    check(lib().dlip_logits_argmax_f32(ptr(e), ptr(W), ptr(bias), ptr(logits), ptr(amax), B, D, K, int(cosine),
                                       stream_handle()), "dlip_logits_argmax_f32")
    return logits, amax


def margin_ce_loss(logits: Tensor, labels: Tensor, scale: float = 1.0, margin: float = 0.0) -> Tensor:
    _req(logits, "logits"); _req(labels, "labels", torch.int64)
    loss = _empty((1,), logits.device)
    check(lib().dlip_margin_ce_loss_f32(ptr(logits), ptr(labels), ptr(loss), logits.shape[0], logits.shape[1], scale,
                                        margin, stream_handle()), "dlip_margin_ce_loss_f32")
    return loss[0]


def lowfer_cat(e1: Tensor, e2: Tensor) -> Tensor:
    _req(e1, "e1"); _req(e2, "e2")
    B, D = e1.shape
    y = _empty((B, 3 * D), e1.device)
    check(lib().dlip_lowfer_cat_f32(ptr(e1), ptr(e2), ptr(y), B, D, stream_handle()), "dlip_lowfer_cat_f32")
    return y

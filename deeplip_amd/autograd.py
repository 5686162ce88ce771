"""torch.autograd plumbing for the TRAINABLE tail of the fusion pipeline (config C5).

The reference freezes both encoders and trains only the fusion head and the criterion
(train_fusion.py:120,198-201).  Autograd is plumbing here: every Function's forward and backward
is one or a few ``dlip_*`` launches (deeplip_amd/csrc/train_ops.hip, plus the MFMA GEMM for the
forward Linear); torch supplies the tape, the optimizer and the RCCL all-reduce.
"""
from __future__ import annotations

import torch
from torch.autograd import Function

from . import _lib, ops
from ._lib import LIFT_WORDS, check, lib, ptr, stream_handle


def _gemm(A, B, M, N, K, ta=False, tb=False):
    C_ = torch.empty((M, N), device=A.device, dtype=torch.float32)
    check(lib().dlip_gemm_small_f32(ptr(A), ptr(B), ptr(C_), M, N, K, int(ta), int(tb), stream_handle()),
          "dlip_gemm_small_f32")
    return C_


def _colsum(x):
    y = torch.empty((x.shape[1],), device=x.device, dtype=torch.float32)
    check(lib().dlip_colsum_f32(ptr(x), ptr(y), x.shape[0], x.shape[1], stream_handle()), "dlip_colsum_f32")
    return y


# (round 6) nn.Linear products of 2^27 multiply-adds or more (the speech encoder's fc1: 256 x 3000 -> 512) on the split-fp16 matrix kernels with
# their balanced work split, like every convolution of the step: the exact-fp32 kernels ran them on 32 - 188 tiles with no split of the
# reduction -- forward 63 us, dx 69 us, dW 44 us (tools/probes: E§R6.17).  An input width that is no multiple of 32 is zero-padded by
# the pass that splits it (3 000 -> 3 008).
LINEAR_ON_MATRIX_KERNELS = __import__("os").environ.get("DLIP_LINEAR_F16X3", "1") != "0"      # (the environment switch: same-box A/B runs)
_ONE_PAIR = {}


def _one_pair(device):
    key = (device.type, device.index)
    if key not in _ONE_PAIR:
        _ONE_PAIR[key] = torch.ones((2,), device=device, dtype=torch.float32)
    return _ONE_PAIR[key]


def _linear_big(x, w) -> bool:
    from . import autograd_video as av
    M, Cin = x.shape
    N = w.shape[0]
    if not (LINEAR_ON_MATRIX_KERNELS and av.TRAIN_CONV == "f16x3" and x.is_cuda and M * N * Cin >= (1 << 27) and Cin % 4 == 0 and N % 32 == 0):
        return False
    # (the (1, 1) scale pair is made by the first eager call: a recorded step's first call is eager -- nothing is allocated under a capture)
    return (x.device.type, x.device.index) in _ONE_PAIR or not torch.cuda.is_current_stream_capturing()


class LinearFn(Function):
    """y = x W^T + b   (nn.Linear; model_fusion.py:19,23, loss.py:14)."""

    @staticmethod
    def forward(ctx, x, w, b):
        x = x.contiguous(); w = w.contiguous()
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        M, Cin = x.shape
        ctx.big = _linear_big(x, w)
        if ctx.big:
            from . import autograd_video as av
            N = w.shape[0]
            Cp = (Cin + 31) // 32 * 32
            xs = torch.empty((M, 1, 1, Cp), device=x.device, dtype=torch.float32)
            check(lib().dlip_split_pack_scaled_pad_f32(ptr(x), ptr(xs), ptr(_one_pair(x.device)), M, Cin, Cp, stream_handle()),
                  "dlip_split_pack_scaled_pad_f32")
            y = av.conv_train(xs, None, b.contiguous() if b is not None else None, w_ref=w.view(N, Cin, 1, 1), xs_ready=xs)   # (xs doubles as the shape holder)
            return y.view(M, N)
        if Cin % 4 == 0:
            return ops.linear(x, w, b.contiguous() if b is not None else None)
        y = _gemm(x, w, M, w.shape[0], Cin, tb=True)
        return y + b if b is not None else y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        M, K = dy.shape
        Cin = x.shape[1]
        if ctx.big:
            from . import autograd_video as av
            lift = av.pow2_lift(dy)
            dy4 = dy.view(M, 1, 1, K)
            dx = None
            if ctx.needs_input_grad[0]:
                dx = av.conv_train(dy4, None, None, lift=True, scale2=lift, w_ref=w.view(K, Cin, 1, 1), transposed=True).view(M, Cin)
            dw = av.wgrad_conv(x.view(M, 1, 1, Cin), dy4, 1, 1, (1, 1), (0, 0), (1, 1), scale2=lift).view(K, Cin) if ctx.needs_input_grad[1] else None
            db = _colsum(dy) if ctx.has_bias and ctx.needs_input_grad[2] else None
            return dx, dw, db
        dx = _gemm(dy, w, M, Cin, K) if ctx.needs_input_grad[0] else None          # dY [M,K] @ W [K,C]
        dw = _gemm(dy, x, K, Cin, M, ta=True) if ctx.needs_input_grad[1] else None  # dY^T [K,M] @ X [M,C]
        db = _colsum(dy) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return dx, dw, db


class BatchNormActTrainFn(Function):
    """y = LeakyReLU_slope(BatchNorm1d_train(x)); updates running stats in place
    (nn.BatchNorm1d + nn.LeakyReLU, model_fusion.py:21-22)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps, slope):
        x = x.contiguous()
        M, C_ = x.shape
        y = torch.empty_like(x)
        mean = torch.empty((C_,), device=x.device, dtype=torch.float32)
        invstd = torch.empty_like(mean)
        check(lib().dlip_bn1d_train_fwd_f32(ptr(x), ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(invstd),
                                            ptr(running_mean), ptr(running_var), M, C_, momentum, eps, slope,
                                            stream_handle()), "dlip_bn1d_train_fwd_f32")
        ctx.save_for_backward(x, y, gamma, mean, invstd)
        ctx.slope = slope
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, gamma, mean, invstd = ctx.saved_tensors
        dy = dy.contiguous()
        M, C_ = x.shape
        if ctx.slope != 1.0:
            g = torch.empty_like(dy)
            check(lib().dlip_lrelu_bwd_f32(ptr(dy), ptr(y), ptr(g), dy.numel(), ctx.slope, stream_handle()),
                  "dlip_lrelu_bwd_f32")
            dy = g
        dx = torch.empty_like(x); dg = torch.empty_like(mean); db = torch.empty_like(mean)
        check(lib().dlip_bn1d_train_bwd_f32(ptr(dy), ptr(x), ptr(mean), ptr(invstd), ptr(gamma), ptr(dx), ptr(dg),
                                            ptr(db), M, C_, stream_handle()), "dlip_bn1d_train_bwd_f32")
        return dx, dg, db, None, None, None, None, None


class L2NormalizeFn(Function):
    """F.normalize(x) (loss.py:44)."""

    @staticmethod
    def forward(ctx, x, eps):
        x = x.contiguous()
        ctx.save_for_backward(x)
        ctx.eps = eps
        return ops.l2_normalize(x, eps)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x)
        check(lib().dlip_l2_normalize_bwd_f32(ptr(x), ptr(dy.contiguous()), ptr(dx), x.shape[0], x.shape[1], ctx.eps,
                                              stream_handle()), "dlip_l2_normalize_bwd_f32")
        return dx, None


class MarginCELossFn(Function):
    """mean_b CE(scale*(logits - margin*onehot(label)) + 1e-8, label)   (loss.py:15,45-48)."""

    @staticmethod
    def forward(ctx, logits, labels, scale, margin):
        logits = logits.contiguous()
        ctx.save_for_backward(logits, labels)
        ctx.scale, ctx.margin = scale, margin
        return ops.margin_ce_loss(logits, labels.contiguous(), scale, margin)

    @staticmethod
    def backward(ctx, dloss):
        logits, labels = ctx.saved_tensors
        g = torch.empty_like(logits)
        dl = dloss.reshape(1).to(torch.float32).contiguous()       # stays on the device: no host read-back per step
        check(lib().dlip_margin_ce_bwd_f32(ptr(logits), ptr(labels), ptr(g), logits.shape[0], logits.shape[1],
                                           ctx.scale, ctx.margin, 1.0, ptr(dl), stream_handle()),
              "dlip_margin_ce_bwd_f32")
        return g, None, None, None


class AAMMarginFn(Function):
    """cos(theta) -> cos(theta + m) on the target column of cosine logits (ArcFace / AAM-softmax)."""

    @staticmethod
    def forward(ctx, logits, labels, margin, easy):
        logits = logits.contiguous()
        ctx.save_for_backward(logits, labels)
        ctx.cfg = (float(margin), int(easy))
        y = torch.empty_like(logits)
        check(lib().dlip_aam_margin_f32(ptr(logits), ptr(labels), None, ptr(y), logits.shape[0], logits.shape[1], float(margin),
                                        int(easy), 0, stream_handle()), "dlip_aam_margin_f32")
        return y

    @staticmethod
    def backward(ctx, dy):
        logits, labels = ctx.saved_tensors
        margin, easy = ctx.cfg
        g = torch.empty_like(logits)
        check(lib().dlip_aam_margin_f32(ptr(logits), ptr(labels), ptr(dy.contiguous()), ptr(g), logits.shape[0], logits.shape[1],
                                        margin, easy, 1, stream_handle()), "dlip_aam_margin_f32")
        return g, None, None, None


def aam_margin(logits, labels, margin, easy=False):
    return AAMMarginFn.apply(logits, labels.contiguous(), margin, easy)


def linear(x, w, b=None):
    return LinearFn.apply(x, w, b)


def bn_act_train(x, bn, slope=1.0):
    """bn: a holders.BatchNormParams (its running stats are updated in place, as torch does)."""
    y = BatchNormActTrainFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps, slope)
    bn.num_batches_tracked += 1
    return y


def l2_normalize(x, eps=1e-12):
    return L2NormalizeFn.apply(x, eps)


def margin_ce_loss(logits, labels, scale=1.0, margin=0.0):
    return MarginCELossFn.apply(logits, labels, float(scale), float(margin))


# ==========================================================================================
# Train-mode speech encoder: what torch.autograd does for
# SpeakerEmbNet.forward under model.train() (train_audio.py:167-183), as dlip_* launches.
# Channels-last [B,T,C] activations throughout; parameters keep the reference layouts.
# ==========================================================================================
# Round 5 (both measured on the speech encoder's training step, tools/bench_train_audio.py; False restores round 4's passes):
BN_STATS_FROM_CONV = __import__("os").environ.get("DLIP_BN_STATS_FROM_CONV", "1") != "0"
# Round 5: a TDNN block's activated output is not stored when the next block can take it on load (TDNNBlockTrainFn: defer / pending)
BN_ON_LOAD = __import__("os").environ.get("DLIP_BN_ON_LOAD", "1") != "0"
ZERO_BIAS_GRAD_BEFORE_BN = __import__("os").environ.get("DLIP_ZERO_BIAS_GRAD", "1") != "0"


def _ws(M: int, C_: int, device):
    """fp64 workspace of the chunked column reductions (encoder_train_ops.hip)."""
    return torch.empty((int(lib().dlip_bn_rows_chunks(M)) * C_ * 2,), device=device, dtype=torch.float64)


class L1NormFn(Function):
    """coef * ||W||_1 (LMCL's regulariser, loss.py:49-50) and its gradient coef * sign(W): two dlip_* launches instead of
    torch.norm(W, 1) under autograd."""

    @staticmethod
    def forward(ctx, w, coef):
        w = w.contiguous()
        out = torch.empty((1,), device=w.device, dtype=torch.float32)
        check(lib().dlip_l1_sum_f32(ptr(w), ptr(out), w.numel(), stream_handle()), "dlip_l1_sum_f32")
        ctx.save_for_backward(w)
        ctx.coef = float(coef)
        return out[0] * float(coef)

    @staticmethod
    def backward(ctx, g):
        (w,) = ctx.saved_tensors
        dw = torch.empty_like(w)
        gs = g.contiguous().view(1)
        check(lib().dlip_l1_sign_f32(ptr(w), ptr(gs), ptr(dw), ctx.coef, w.numel(), stream_handle()), "dlip_l1_sign_f32")
        return dw, None


def l1_norm(w, coef=1.0):
    return L1NormFn.apply(w, coef)


def _permute3(x, perm, flip_axis=-1):
    d0, d1, d2 = x.shape
    y = torch.empty(tuple(x.shape[p] for p in perm), device=x.device, dtype=torch.float32)
    check(lib().dlip_permute3_f32(ptr(x), ptr(y), d0, d1, d2, perm[0], perm[1], perm[2], flip_axis, stream_handle()),
          "dlip_permute3_f32")
    return y


def _bn_rows_fwd(x2, gamma, beta, rm, rv, momentum, eps, slope, act_first, ready=None, nbt=None, stats_only=False):
    """``ready`` = (ws, chunks): the partial column sums of x2 already written by the convolution that produced it
    (conv_train(..., stats=...)): the statistics pass over x2 is skipped.  ``nbt``: the module's num_batches_tracked, incremented by the
    launch that finishes the statistics."""
    M, C_ = x2.shape
    _lib.ensure_conv_workspace()          # (its ticket words: the finalize step runs in the statistics pass's last workgroup)
    y = None if stats_only else torch.empty_like(x2)      # (stats_only: the consumer applies the BatchNorm on load)
    mean = torch.empty((C_,), device=x2.device, dtype=torch.float32)
    invstd = torch.empty_like(mean)
    ws, chunks = ready if ready is not None else (_ws(M, C_, x2.device), 0)
    check(lib().dlip_bn_rows_train_fwd_f32(ptr(x2), ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(invstd), ptr(rm), ptr(rv),
                                           ptr(ws), M, C_, momentum, eps, slope, int(act_first), int(chunks), ptr(nbt),
                                           stream_handle()), "dlip_bn_rows_train_fwd_f32")
    return y, mean, invstd


def _bn_rows_bwd(dy2, x2, gamma, beta, mean, invstd, slope, act_first):
    M, C_ = x2.shape
    _lib.ensure_conv_workspace()
    dx = torch.empty_like(x2)
    dg = torch.empty_like(mean)
    db = torch.empty_like(mean)
    lift = torch.empty((LIFT_WORDS,), device=x2.device, dtype=torch.float32)
    check(lib().dlip_bn_rows_train_bwd_f32(ptr(dy2), ptr(x2), ptr(gamma), ptr(beta), ptr(mean), ptr(invstd), ptr(dx), ptr(dg),
                                           ptr(db), ptr(_ws(M, C_, x2.device)), M, C_, slope, int(act_first), ptr(lift), stream_handle()),
          "dlip_bn_rows_train_bwd_f32")
    dx._dlip_lift = lift      # the power-of-two lift of dx, formed by the pass that wrote it (autograd_video.pow2_lift picks it up)
    return dx, dg, db


# (round 6, ABI 47) conv -> BatchNorm -> activation under backward(): the BatchNorm's input gradient dz is needed only as the operand
# images of the convolution in front (weight gradient, data gradient), so it is formed PER LOADED VALUE by their producers from dy and z
# (dlip_wgrad_*_bnbwd_f32) and never stored: the apply pass's write and the producers' read of an activation-sized tensor per layer.
BN_BWD_ON_LOAD = __import__("os").environ.get("DLIP_BN_BWD_ON_LOAD", "1") != "0"      # (the environment switch: same-box A/B runs)
BN_SMALL_ROWS = 4096          # (encoder_train_ops.hip: at most this many rows go through the one-launch BatchNorm, which writes dz)


def _bn_rows_bwd_sums(dy2, x2, gamma, beta, mean, invstd, slope, act_first, ms=None):
    """dgamma, dbeta and the lift of the dz that the producers will form on load (a DLIP_LIFT_WORDS buffer, as pow2_lift returns it).
    ``ms`` = (coef, T): dy2 itself is formed on load from a statistics pooling's coefficients (MeanStdPoolFn hand_over; dy2 may be None)."""
    M, C_ = x2.shape
    ms_c, ms_T = ms if ms is not None else (None, 0)
    _lib.ensure_conv_workspace()
    dg = torch.empty_like(mean)
    db = torch.empty_like(mean)
    lift = torch.empty((LIFT_WORDS,), device=x2.device, dtype=torch.float32)
    parts = torch.empty((2 * ((C_ + 63) // 64) * int(lib().dlip_bn_rows_chunks(M)),), device=x2.device, dtype=torch.float32)
    check(lib().dlip_bn_rows_train_bwd_sums_f32(ptr(dy2), ptr(x2), ptr(gamma), ptr(beta), ptr(mean), ptr(invstd), ptr(dg), ptr(db),
                                                ptr(_ws(M, C_, x2.device)), ptr(parts), M, C_, slope, int(act_first), ptr(lift), ptr(ms_c),
                                                int(ms_T), stream_handle()), "dlip_bn_rows_train_bwd_sums_f32")
    return dg, db, lift


class BNRowsActFn(Function):
    """[M,C]: lrelu(bn_train(x)) (act_first=False) or bn_train(lrelu(x)) (True); running stats updated in place
    (bn1 / bn2 of SpeakerEmbNet, tdnn.py:92-97,105-110)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps, slope, act_first, nbt=None):
        shape = tuple(x.shape)                                 # any channels-last shape [..., C]: rows = all leading axes
        x = x.contiguous().view(-1, shape[-1])
        y, mean, invstd = _bn_rows_fwd(x, gamma, beta, running_mean, running_var, momentum, eps, slope, act_first, nbt=nbt)
        ctx.save_for_backward(x, gamma, beta, mean, invstd)
        ctx.slope, ctx.act_first, ctx.shape = slope, act_first, shape
        return y.view(shape)

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, mean, invstd = ctx.saved_tensors
        dx, dg, db = _bn_rows_bwd(dy.contiguous().view(x.shape), x, gamma, beta, mean, invstd, ctx.slope, ctx.act_first)
        out = dx.view(ctx.shape)
        out._dlip_lift = dx._dlip_lift                        # travels with the tensor object the next backward receives
        return out, dg, db, None, None, None, None, None, None, None


class TDNNBlockTrainFn(Function):
    """TDNN_Block under autograd (tdnn.py:35-43): Conv1d(k, dilation, no padding, bias) -> BatchNorm1d (batch
    statistics) -> LeakyReLU (or Conv -> LeakyReLU -> BN when bn_first=False), x [B,T,C] -> [B,T',K].

    forward: the fp32 implicit-GEMM kernel on the raw (unfolded) weights + the row-BN kernels.
    backward: BN/activation backward; bias gradient = column sum; DATA gradient = the same conv kernel
    on the flipped, transposed weights with full padding; WEIGHT gradient = one GEMM per filter tap with
    the B*T axis as the reduction -- both operands transposed to reduction-major, split into (hi, lo)
    fp16 pairs (the gradient after a power-of-two lift into fp16's normal range) and sent through the
    LDS-DMA kernel, whose balanced work split is what makes a 512 x 512 x 19 200 product fill the chip."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, running_mean, running_var, momentum, eps, slope, dilation, act_first, nbt=None,
                defer=False, pending=None):
        """``pending`` = (z, mean, invstd, gamma, beta, slope) of the PREVIOUS block: x then is that block's output tensor whose
        VALUES WERE NEVER WRITTEN (defer) -- this block's operand producers read the previous convolution's raw output z and apply
        its BatchNorm + LeakyReLU on load (dlip_wgrad_*_bn_f32); a block that cannot (channel counts, arithmetic mode) writes the
        values first.  ``defer``: do the same for THIS block's output: returns (y [values not written], z, mean, invstd); the
        backward is the same either way (it needs z, never y)."""
        x = x.contiguous()
        B, T, Cx = x.shape
        K, Cw, S = weight.shape
        # (a first layer on 24 features may read its input zero-padded to 32 channels: then the split-fp16 kernels serve it too)
        if (Cw != Cx and (Cw + 31) // 32 * 32 != Cx) or Cx % 4 or K % 4:
            raise ValueError(f"TDNN train path: channels must match and be multiples of 4 (x {Cx}, weight {Cw}, out {K})")
        from . import autograd_video as av
        B_, T_, C_in = x.shape
        # (round 4) ONE read of x writes both of its split images: the forward convolution's operand and the weight gradient's (kept
        # for the backward instead of x): the reduction-major GEMM operand of a k = 1 layer (mode 1), the [C][T][B32] image of the
        # weight gradient run as a convolution otherwise (mode 2) -- what ConvTrainFn does for the lip-clip encoder
        mode, xT, xs = 0, None, None
        images = ctx.needs_input_grad[1] and av.TRAIN_CONV == "f16x3" and K % 4 == 0 and (Cw == Cx or (Cw + 31) // 32 * 32 == Cx)
        on_load = pending is not None and images and Cx % 64 == 0 and (S == 1 or av.WGRAD == "conv") and av.WGRAD_SLICE_MAJOR
        if pending is not None and not on_load:
            materialize_pending(x, pending)               # this block reads x itself: write the values the previous block left out
        if images:
            if S == 1 and Cx % 64 == 0:
                xT, xs = av.operand_and_split(x.view(B_ * T_, C_in), bn=pending if on_load else None)
                mode, xs = 1, xs.view(B_, 1, T_, C_in)
            elif S > 1 and Cx % 32 == 0 and av.WGRAD == "conv":
                xT, xs = av.wgrad_image(x.view(B_, 1, T_, C_in), None, also_nhwc_split=True, bn=pending if on_load else None)
                mode = 2
        # (round 5) conv -> BN: the batch statistics come out of the convolution's epilogue (per half tile the column sums of what it
        # writes) instead of a pass of their own over z -- 155 MB read per layer at B = 256
        stats = None
        if not act_first and BN_STATS_FROM_CONV and av.TRAIN_CONV == "f16x3":
            n_ch = ops.conv_stats_chunks(B_, 1, T_, C_in, K, 1, S, dil=(1, dilation))
            if n_ch > 0:
                stats = {"chunks": n_ch, "ws": torch.empty((n_ch * K * 2,), device=x.device, dtype=torch.float64)}
        z = av.conv_train(x.view(B_, 1, T_, C_in), None, bias.contiguous() if bias is not None else None, (1, 1), (0, 0), (1, dilation),
                          w_ref=weight.view(K, Cw, 1, S), xs_ready=xs, stats=stats)   # reference [K,C,S]: split image written straight from it
        z = z.view(B_, z.shape[2], z.shape[3])
        Tp = z.shape[1]
        ready = (stats["ws"], stats["chunks"]) if stats is not None and stats.get("done") else None
        ctx.expect_ms = defer == 2 and not act_first      # (defer == 2: deferred to a statistics pooling that hands its backward over)
        defer = bool(defer) and not act_first
        y2, mean, invstd = _bn_rows_fwd(z.view(B * Tp, K), gamma, beta, running_mean, running_var, momentum, eps, slope, act_first, ready, nbt,
                                        stats_only=defer)
        ctx.save_for_backward(xT if mode else x, weight, z, gamma, beta, mean, invstd)
        ctx.cfg = (dilation, slope, act_first, bias is not None)
        ctx.x_shape, ctx.mode = (B, T, Cx), mode
        if defer:
            y2 = torch.empty((B, Tp, K), device=x.device, dtype=torch.float32)   # (address and shape only: nobody reads its values)
            ctx.mark_non_differentiable(z, mean, invstd)
            ctx.set_materialize_grads(False)      # (or autograd hands the backward ZERO-FILLED gradients for the three: a 155 MB fill per block)
            return y2, z, mean, invstd
        return y2.view(B, Tp, K)

    @staticmethod
    def backward(ctx, dy, *_unused):
        x, weight, z, gamma, beta, mean, invstd = ctx.saved_tensors
        dilation, slope, act_first, has_bias = ctx.cfg
        B, T, Cx = ctx.x_shape
        mode = ctx.mode
        K, _, S = weight.shape
        Tp = z.shape[1]
        dev = x.device
        from . import autograd_video as av
        ms = getattr(dy, "_dlip_ms", None)
        if ctx.expect_ms and ms is None:
            raise RuntimeError("TDNNBlockTrainFn.backward: the gradient handed over by the statistics pooling (MeanStdPoolFn, hand_over) arrived "
                               "without its source -- autograd copied or accumulated the tensor; set DLIP_POOL_BWD_ON_LOAD=0")
        want_x, want_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        no_dbias_pass = not (has_bias and ctx.needs_input_grad[2]) or (not act_first and ZERO_BIAS_GRAD_BEFORE_BN)
        # (ABI 49) the k = 1 producer takes any K % 4 == 0 (tdnn.9's 1 500): a ragged last channel block, the split copy at a padded row pitch
        fused = (BN_BWD_ON_LOAD and av.TRAIN_CONV == "f16x3" and mode in (1, 2) and want_w and (K % 64 == 0 or (mode == 1 and K % 4 == 0))
                 and B * Tp > BN_SMALL_ROWS and no_dbias_pass and av.WGRAD_SLICE_MAJOR and (mode == 1 or av.WGRAD == "conv")
                 and (ms is None or mode == 1))
        if ms is not None and not fused:
            # dy was never written: formed per loaded value inside the BatchNorm backward from the pooled statistics and their gradient
            coef_ms, T_ms = ms
            M = B * Tp
            _lib.ensure_conv_workspace()
            dz2 = torch.empty((M, K), device=dev, dtype=torch.float32)
            dgamma, dbeta = torch.empty_like(mean), torch.empty_like(mean)
            lift = torch.empty((LIFT_WORDS,), device=dev, dtype=torch.float32)
            check(lib().dlip_bn_rows_train_bwd_ms_f32(ptr(coef_ms), int(T_ms), ptr(z), ptr(gamma), ptr(beta), ptr(mean), ptr(invstd),
                                                      ptr(dz2), ptr(dgamma), ptr(dbeta), ptr(_ws(M, K, dev)), M, K, float(slope), ptr(lift),
                                                      stream_handle()), "dlip_bn_rows_train_bwd_ms_f32")
            dz2._dlip_lift = lift
            return TDNNBlockTrainFn._backward_from_dz(ctx, dz2, dgamma, dbeta)
        if fused:
            # dz is never stored: sums + lift, then the producers form it on load (see BN_BWD_ON_LOAD); with ``ms`` dy is not stored either
            dy2, z2 = (dy.contiguous().view(B * Tp, K) if ms is None else None), z.view(B * Tp, K)
            dgamma, dbeta, lift = _bn_rows_bwd_sums(dy2, z2, gamma, beta, mean, invstd, slope, act_first, ms)
            ms_c, ms_T = ms if ms is not None else (None, 0)
            dbias = torch.zeros((K,), device=dev, dtype=torch.float32) if has_bias and ctx.needs_input_grad[2] else None
            M = B * Tp
            sl = float(slope)
            Kp = (K + 31) // 32 * 32                     # row pitch of the data gradient's split operand (K % 32 != 0: zero-padded channels)
            if mode == 1:
                J32 = x.shape[1]
                gT = torch.empty((K, J32), device=dev, dtype=torch.float32)
                dzs = torch.empty((B, 1, Tp, Kp), device=dev, dtype=torch.float32) if want_x else None
                check(lib().dlip_wgrad_operand_split_bnbwd_f32(ptr(dy2), ptr(z2), ptr(gT), J32, M, K, ptr(mean), ptr(invstd), ptr(gamma), ptr(beta),
                                                               ptr(dgamma), ptr(dbeta), M, sl, int(act_first), ptr(lift), ptr(dzs),
                                                               Kp if Kp != K else 0, ptr(ms_c), int(ms_T), stream_handle()),
                      "dlip_wgrad_operand_split_bnbwd_f32")
            else:
                N32 = (B + 31) // 32 * 32
                gT = torch.empty((K, 1, Tp, N32), device=dev, dtype=torch.float32)
                dzs = torch.empty((B, 1, Tp, K), device=dev, dtype=torch.float32) if want_x else None
                check(lib().dlip_wgrad_chwn_bnbwd_f32(ptr(dy2), ptr(z2), ptr(gT), B, 1, Tp, K, N32, ptr(mean), ptr(invstd), ptr(gamma), ptr(beta),
                                                      ptr(dgamma), ptr(dbeta), M, sl, int(act_first), ptr(lift), ptr(dzs), stream_handle()),
                      "dlip_wgrad_chwn_bnbwd_f32")
            dx = None
            if want_x:
                shape_only = torch.empty((B, 1, Tp, K), device=dev, dtype=torch.float32)      # (address and shape: conv_train reads dzs)
                dx = av.conv_train(shape_only, None, None, (1, 1), (0, (S - 1) * dilation), (1, dilation), lift=True, scale2=lift,
                                   w_ref=weight.view(K, Cx, 1, S), transposed=True, xs_ready=dzs)
                dx = dx.view(B, dx.shape[2], Cx)
            if mode == 1:
                dweight = _permute3(av.wgrad_gemm_operands(x, gT, lift).view(1, Cx, K), (2, 1, 0)).view(K, Cx, 1)
            else:
                dweight = av.wgrad_as_conv(None, None, 1, S, (1, 1), (0, 0), (1, dilation), scale2=lift, xT=x, gT=gT).view(K, Cx, S)
            if dweight.shape[1] != weight.shape[1]:
                dweight = dweight[:, :weight.shape[1]].contiguous()
            return dx, dweight, dbias, dgamma, dbeta, None, None, None, None, None, None, None, None, None, None
        dz2, dgamma, dbeta = _bn_rows_bwd(dy.contiguous().view(B * Tp, K), z.view(B * Tp, K), gamma, beta, mean, invstd,
                                          slope, act_first)
        return TDNNBlockTrainFn._backward_from_dz(ctx, dz2, dgamma, dbeta)

    @staticmethod
    def _backward_from_dz(ctx, dz2, dgamma, dbeta):
        """The convolution's part of the backward, given the BatchNorm's input gradient dz2 [B T', K] (and its lift, as an attribute)."""
        x, weight, z, gamma, beta, mean, invstd = ctx.saved_tensors
        dilation, slope, act_first, has_bias = ctx.cfg
        B, T, Cx = ctx.x_shape
        mode = ctx.mode
        K, _, S = weight.shape
        Tp = z.shape[1]
        dev = x.device
        dbias = None
        if has_bias and ctx.needs_input_grad[2]:
            if not act_first and ZERO_BIAS_GRAD_BEFORE_BN:
                # conv -> batch-statistics BN: the bias gradient is the column sum of the BatchNorm's input gradient, and that is
                # IDENTICALLY zero: dz = gamma * invstd * (g - mean(g) - xhat * mean(g * xhat)) sums to -gamma invstd mean(g xhat)
                # sum(xhat) = 0.  torch's autograd (the reference) returns that sum's rounding noise (~1e-9 |dy|); summing 155 MB per
                # layer to reproduce noise is a pass this engine does not make.
                dbias = torch.zeros((K,), device=dev, dtype=torch.float32)
            else:
                dbias = torch.empty((K,), device=dev, dtype=torch.float32)
                check(lib().dlip_colsum_rows_f32(ptr(dz2), ptr(dbias), ptr(_ws(B * Tp, K, dev)), B * Tp, K, stream_handle()),
                      "dlip_colsum_rows_f32")
        dz = dz2.view(B, Tp, K)
        dx = None
        from . import autograd_video as av
        lift = av.pow2_lift(dz2)                                               # one absmax pass for dgrad and wgrad alike
        # one read of dz writes its weight-gradient image AND the data gradient's lifted split operand (when its channels allow)
        gT = dzs = None
        if mode == 1 and ctx.needs_input_grad[1]:
            if K % 64 == 0 and ctx.needs_input_grad[0]:
                gT, dzs = av.operand_and_split(dz2, lift)
                dzs = dzs.view(B, 1, Tp, K)
            else:
                J32 = x.shape[1]
                gT = torch.empty((K, J32), device=dev, dtype=torch.float32)
                check(lib().dlip_wgrad_operand_f32(ptr(dz2), ptr(gT), J32, B, 1, Tp, K, K, 1, Tp, 1, 1, 1, 1, 1, 1, 0, 0, ptr(lift), stream_handle()),
                      "dlip_wgrad_operand_f32")
        elif mode == 2 and ctx.needs_input_grad[1]:
            gT, dzs = av.wgrad_image(dz2.view(B, 1, Tp, K), lift, also_nhwc_split=ctx.needs_input_grad[0] and K % 32 == 0)
        if ctx.needs_input_grad[0]:
            dzc = dz.contiguous()
            dx = av.conv_train(dzc.view(B, 1, Tp, K), None, None, (1, 1), (0, (S - 1) * dilation), (1, dilation), lift=True, scale2=lift,
                               w_ref=weight.view(K, Cx, 1, S), transposed=True, xs_ready=dzs)   # [K,C,S] -> rows c of [S reversed][K]
            dx = dx.view(B, dx.shape[2], Cx)
        dweight = None
        if ctx.needs_input_grad[1]:
            if mode == 1:
                dweight = _permute3(av.wgrad_gemm_operands(x, gT, lift).view(1, Cx, K), (2, 1, 0)).view(K, Cx, 1)
            elif mode == 2:
                dweight = av.wgrad_as_conv(None, None, 1, S, (1, 1), (0, 0), (1, dilation), scale2=lift, xT=x, gT=gT).view(K, Cx, S)
            else:
                dweight = _conv1d_wgrad(x, dz, S, dilation, lift)
            if dweight.shape[1] != weight.shape[1]:          # zero-padded input channels: their gradient columns are not parameters
                dweight = dweight[:, :weight.shape[1]].contiguous()
        return dx, dweight, dbias, dgamma, dbeta, None, None, None, None, None, None, None, None, None, None


def _conv1d_wgrad(x, dz, S, dilation, lift=None):
    """dW[k,c,s] = sum_{b,t} dz[b,t,k] x[b,t+s*dil,c]  ->  reference layout [K,C,S]."""
    B, T, Cx = x.shape
    _, Tp, K = dz.shape
    dev = x.device
    if Cx % 4 == 0 and K % 4 == 0:
        # a "valid" 1-D convolution is the H = 1 case of the fused operand path (autograd_video.wgrad_conv_fused)
        from .autograd_video import wgrad_conv
        return wgrad_conv(x.contiguous().view(B, 1, T, Cx), dz.contiguous().view(B, 1, Tp, K), 1, S, (1, 1), (0, 0), (1, dilation),
                          scale2=lift).view(K, Cx, S)                         # reference layout [K,C,S]
    dzp = torch.zeros((B, T, K), device=dev, dtype=torch.float32)              # rows t >= T' stay zero: no cross-utterance terms
    dzp[:, :Tp].copy_(dz)
    J = B * T - (S - 1) * dilation                                             # rows every tap can read
    J32 = (J + 31) // 32 * 32
    xf, dzf = x.view(B * T, Cx), dzp.view(B * T, K)
    # reduction-major operands: [rows, J32] with the B*T axis contiguous (zero padded)
    dzT = torch.empty((1, K, J32), device=dev, dtype=torch.float32)
    check(lib().dlip_nct_to_ntc_f32(ptr(dzf), ptr(dzT), 1, J, K, J32, stream_handle()), "dlip_nct_to_ntc_f32")
    scale2 = torch.empty((2,), device=dev, dtype=torch.float32)
    check(lib().dlip_pow2_scale_f32(ptr(dzT), ptr(scale2), dzT.numel(), 1024.0, stream_handle()), "dlip_pow2_scale_f32")
    dzT_s = torch.empty_like(dzT)
    check(lib().dlip_split_pack_scaled_f32(ptr(dzT), ptr(dzT_s), ptr(scale2), K, J32, stream_handle()), "dlip_split_pack_scaled_f32")
    inv = torch.empty((K,), device=dev, dtype=torch.float32)
    check(lib().dlip_fill_from_scalar_f32(scale2[1:].data_ptr(), ptr(inv), K, stream_handle()), "dlip_fill_from_scalar_f32")
    ones = torch.ones((K,), device=dev, dtype=torch.float32)
    zeros = torch.zeros((K,), device=dev, dtype=torch.float32)
    dwt = torch.empty((S, Cx, K), device=dev, dtype=torch.float32)
    xT = torch.empty((1, Cx, J32), device=dev, dtype=torch.float32)
    for s in range(S):
        xs = xf[s * dilation: s * dilation + J]                                # contiguous row slice: no copy
        check(lib().dlip_nct_to_ntc_f32(ptr(xs), ptr(xT), 1, J, Cx, J32, stream_handle()), "dlip_nct_to_ntc_f32")
        xT_s = ops.split_pack(xT.view(Cx, J32))
        ops.conv_nhwc(xT_s.view(1, 1, Cx, J32), dzT_s.view(K, 1, 1, J32), None, w_scale=ones, x_split=True,
                      post_scale=inv, post_shift=zeros, out=dwt[s].view(1, 1, Cx, K))
    return _permute3(dwt, (2, 1, 0))                                           # [S,C,K] -> [K,C,S]


class MeanStdPoolFn(Function):
    """MeanStdPooling on [B,T,C] -> [B,2C] (pooling.py:24-26) and its backward.  ``pending`` = (z, mean, invstd, gamma, beta, slope) of the
    block in front (TDNNBlockTrainFn defer): x's values were never written -- both kernels read the raw convolution output z and apply
    that block's BatchNorm + LeakyReLU per loaded value (dlip_meanstd_pool_bn_f32 / _bwd_bn_f32, ABI 48)."""

    @staticmethod
    def forward(ctx, x, pending=None, hand_over=False):
        """``hand_over`` (with ``pending``): the backward writes nothing -- see backward."""
        x = x.contiguous()
        ctx.hand_over = bool(hand_over) and pending is not None
        if pending is not None:
            z, mean, invstd, gamma, beta, slope = pending
            B, T, C_ = x.shape
            y = torch.empty((B, 2 * C_), device=x.device, dtype=torch.float32)
            check(lib().dlip_meanstd_pool_bn_f32(ptr(z), ptr(mean), ptr(invstd), ptr(gamma), ptr(beta), float(slope), ptr(y), B, T, C_, stream_handle()),
                  "dlip_meanstd_pool_bn_f32")
            ctx.save_for_backward(z, y, mean, invstd, gamma, beta)
            ctx.slope, ctx.on_load = float(slope), True
            return y
        y = ops.meanstd_pool(x)
        ctx.save_for_backward(x, y)
        ctx.on_load = False
        return y

    @staticmethod
    def backward(ctx, dy):
        if ctx.on_load:
            z, y, mean, invstd, gamma, beta = ctx.saved_tensors
            B, T, C_ = z.shape
            dx = torch.empty_like(z)
            if ctx.hand_over:
                # The block in front forms this gradient itself, per loaded value, inside its BatchNorm backward (MsSrc,
                # dlip_bn_rows_train_bwd_ms_f32): dx stays UNWRITTEN and carries what that takes.  TDNNBlockTrainFn.backward raises if the
                # tensor it receives has lost the attribute (it was told at forward time to expect it).
                coef = torch.empty_like(y)            # [B, 2C] = (A | K): dy[b,t,c] = A + K y, one launch over the pooled tensors
                check(lib().dlip_meanstd_bwd_coef_f32(ptr(y), ptr(dy.contiguous()), ptr(coef), B, C_, T, stream_handle()), "dlip_meanstd_bwd_coef_f32")
                dx._dlip_ms = (coef, T)
                return dx, None, None
            check(lib().dlip_meanstd_pool_bwd_bn_f32(ptr(z), ptr(mean), ptr(invstd), ptr(gamma), ptr(beta), ctx.slope, ptr(y), ptr(dy.contiguous()),
                                                     ptr(dx), B, T, C_, stream_handle()), "dlip_meanstd_pool_bwd_bn_f32")
            return dx, None, None
        x, y = ctx.saved_tensors
        B, T, C_ = x.shape
        dx = torch.empty_like(x)
        check(lib().dlip_meanstd_pool_bwd_f32(ptr(x), ptr(y), ptr(dy.contiguous()), ptr(dx), B, T, C_, stream_handle()),
              "dlip_meanstd_pool_bwd_f32")
        return dx, None, None


class AttnStatPoolFn(Function):
    """The tail of AttentiveStatPooling (models/audio_models/pooling.py:87-107) behind its hidden layer: e = relu(hidden) v + k,
    alpha = softmax over frames, y = [sum alpha x | sqrt(sum alpha x^2 - mean^2)].  x [B,T,C], hidden [B,T,H] (= x W^T + b from the
    differentiable GEMM in front: W, b and the second path into x get their gradients there), v [H,1], k [1,1]."""

    @staticmethod
    def forward(ctx, x, hidden, v, k):
        x, hidden = x.contiguous(), hidden.contiguous()
        B, T, C_ = x.shape
        H = hidden.shape[2]
        y = torch.empty((B, 2 * C_), device=x.device, dtype=torch.float32)
        alpha = torch.empty((B, T), device=x.device, dtype=torch.float32)
        vv, kk = v.detach().contiguous(), k.detach().contiguous()
        check(lib().dlip_attentive_stat_pool_f32(ptr(x), ptr(hidden), ptr(vv), ptr(kk), None, 0, ptr(y), ptr(alpha), B, T, C_, H,
                                                 stream_handle()), "dlip_attentive_stat_pool_f32")
        ctx.save_for_backward(x, hidden, vv, alpha, y)
        ctx.shapes = (tuple(v.shape), tuple(k.shape))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, hidden, vv, alpha, y = ctx.saved_tensors
        B, T, C_ = x.shape
        H = hidden.shape[2]
        dx = torch.empty_like(x)
        dh = torch.empty_like(hidden)
        rde = torch.empty_like(hidden)
        de = torch.empty((B, T), device=x.device, dtype=torch.float32)
        check(lib().dlip_attentive_stat_pool_bwd_f32(ptr(x), ptr(hidden), ptr(vv), ptr(alpha), ptr(y), ptr(dy.contiguous()), None, 0, ptr(dx),
                                                     ptr(dh), ptr(rde), ptr(de), B, T, C_, H, stream_handle()), "dlip_attentive_stat_pool_bwd_f32")
        dv = _colsum(rde.view(B * T, H)).view(ctx.shapes[0]) if ctx.needs_input_grad[2] else None
        dk = _colsum(de.view(B * T, 1)).view(ctx.shapes[1]) if ctx.needs_input_grad[3] else None
        return dx, dh, dv, dk


def attentive_stat_pool(x, pool):
    """AttentiveStatPooling.forward under model.train() (pooling.py:99-107) on channels-last x [B,T,C] -> [B,2C]: the hidden layer
    W x + b is the engine's differentiable 1 x 1 convolution (forward, data and weight gradient on the MFMA kernels), the rest
    AttnStatPoolFn."""
    from . import autograd_video as av
    B, T, C_ = x.shape
    H = pool.hidden_size
    hidden = av.conv(x.reshape(B, 1, T, C_), pool.W.view(H, C_, 1, 1), pool.b.reshape(-1)).reshape(B, T, H)
    return AttnStatPoolFn.apply(x, hidden, pool.v, pool.k)


def materialize_pending(y, pending):
    """Write the values of a deferred block output ``y`` (TDNNBlockTrainFn defer): lrelu(bn(z)) from the statistics already formed."""
    z, mean, invstd, gamma, beta, slope = pending
    M, C_ = z.numel() // z.shape[-1], z.shape[-1]
    sv, sc = (slope, 0.0) if isinstance(slope, torch.Tensor) else (None, float(slope))
    check(lib().dlip_bn_apply_rows_f32(ptr(z), ptr(mean), ptr(invstd), ptr(gamma), ptr(beta), ptr(sv), sc, ptr(y), M, C_, stream_handle()),
          "dlip_bn_apply_rows_f32")


def tdnn_block_train(x, blk, pending=None, defer=False):
    """blk: deeplip_amd.audio.TDNN_Block in train mode; x [B,T,C] channels-last.  ``pending`` / ``defer``: TDNNBlockTrainFn --
    returns (y, pending for the next block or None)."""
    bn = blk.bn
    defer = (2 if defer == 2 else 1) if (bool(defer) and BN_ON_LOAD and blk.bn_first) else False     # (2: deferred to a pooling that hands its backward over)
    out = TDNNBlockTrainFn.apply(x, blk.context_layer.weight, blk.context_layer.bias, bn.weight, bn.bias, bn.running_mean,
                                 bn.running_var, bn.momentum, bn.eps, 0.2, blk.dilation, not blk.bn_first, bn.num_batches_tracked,
                                 defer, pending)
    if defer:                             # (num_batches_tracked += 1: done by the launch that finishes the batch statistics)
        y, z, mean, invstd = out
        return y, (z, mean, invstd, bn.weight.detach(), bn.bias.detach(), 0.2)
    return out, None


def bn_rows_act_train(x, bn, slope, act_first):
    return BNRowsActFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps, slope, act_first,
                             bn.num_batches_tracked)


POOL_BN_ON_LOAD = __import__("os").environ.get("DLIP_POOL_BN_ON_LOAD", "1") != "0"      # (the environment switch: same-box A/B runs)


POOL_BWD_ON_LOAD = __import__("os").environ.get("DLIP_POOL_BWD_ON_LOAD", "1") != "0"     # (the environment switch: same-box A/B runs)


def meanstd_pool(x, pending=None, hand_over=False):
    """``pending`` / ``hand_over``: see MeanStdPoolFn (the last TDNN block's output taken on load; its gradient formed by that block)."""
    return MeanStdPoolFn.apply(x, pending, hand_over)

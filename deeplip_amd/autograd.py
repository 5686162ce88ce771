"""torch.autograd plumbing for the TRAINABLE tail of the fusion pipeline (config C5).

The reference freezes both encoders and trains only the fusion head and the criterion
(train_fusion.py:120,198-201).  Autograd is plumbing here: every Function's forward and backward
is one or a few ``dlip_*`` launches (deeplip_amd/csrc/train_ops.hip, plus the MFMA GEMM for the
forward Linear); torch supplies the tape, the optimizer and the RCCL all-reduce.
"""
from __future__ import annotations

import torch
from torch.autograd import Function

from . import ops
from ._lib import check, lib, ptr, stream_handle


def _gemm(A, B, M, N, K, ta=False, tb=False):
    C_ = torch.empty((M, N), device=A.device, dtype=torch.float32)
    check(lib().dlip_gemm_small_f32(ptr(A), ptr(B), ptr(C_), M, N, K, int(ta), int(tb), stream_handle()),
          "dlip_gemm_small_f32")
    return C_


def _colsum(x):
    y = torch.empty((x.shape[1],), device=x.device, dtype=torch.float32)
    check(lib().dlip_colsum_f32(ptr(x), ptr(y), x.shape[0], x.shape[1], stream_handle()), "dlip_colsum_f32")
    return y


class LinearFn(Function):
    """y = x W^T + b   (nn.Linear; model_fusion.py:19,23, loss.py:14)."""

    @staticmethod
    def forward(ctx, x, w, b):
        x = x.contiguous(); w = w.contiguous()
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        M, Cin = x.shape
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
        check(lib().dlip_margin_ce_bwd_f32(ptr(logits), ptr(labels), ptr(g), logits.shape[0], logits.shape[1],
                                           ctx.scale, ctx.margin, float(dloss), stream_handle()),
              "dlip_margin_ce_bwd_f32")
        return g, None, None, None


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

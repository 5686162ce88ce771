"""Gradient fuzz of the two training primitives: autograd_video.conv (Conv2d / Conv1d on NHWC) and autograd.TDNNBlockTrainFn (Conv1d + train-mode
BatchNorm + LeakyReLU) on RANDOM shapes -- batch, extent, channels (multiples of 4 / 32 / 64 and odd ones the fused flows fall back on),
filter, stride, padding, dilation -- against torch autograd in fp64.  The tests fix 7 + 7 shapes; a wrong pad / pitch / tail rule of the
operand producers or the weight-gradient convolution is an O(1) error on SOME shape.   python tools/probes/grad_fuzz.py [n] [seed]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import torch
import torch.nn.functional as F

from deeplip_amd import autograd as ag, autograd_video as av, _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
r = np.random.Generator(np.random.PCG64(seed))
g = torch.Generator().manual_seed(seed)
DEV = "cuda"


def rel(a, b):
    a, b = a.double().cpu().numpy(), b.double().cpu().numpy()
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def pick(opts):
    return opts[int(r.integers(0, len(opts)))]


worst = (0.0, None)
bad = 0
kinked = 0
for i in range(n):
    # ---- conv
    one_d = bool(i % 2)
    C = pick([4, 8, 24, 32, 64, 96, 128, 256, 512, 768])
    K = pick([4, 32, 64, 100, 128, 256, 512])
    if one_d:
        H, W = 1, int(r.integers(5, 80))
        S = pick([1, 3, 5, 7]); R = 1
        dil = pick([1, 2, 4]) if S > 1 else 1
        stride = 1
        pad = pick([0, (S - 1) * dil, (S - 1) * dil // 2])
        N = int(r.integers(1, 9))
    else:
        H, W = int(r.integers(3, 30)), int(r.integers(3, 30))
        R = S = pick([1, 3])
        stride = pick([1, 2])
        dil = 1
        pad = pick([0, 1]) if R == 3 else 0
        N = int(r.integers(1, 7))
        if C * H * W * N > 3_000_000:
            N = 1
    sh, sw = (1, stride) if H == 1 else (stride, stride)
    ph, pw = (0, pad) if H == 1 else (pad, pad)
    dh, dw = (1, dil) if H == 1 else (dil, dil)
    if (H + 2 * ph - dh * (R - 1) - 1) < 0 or (W + 2 * pw - dw * (S - 1) - 1) < 0:
        continue
    bias = bool(r.integers(0, 2))
    x = torch.randn(N, C, H, W, generator=g).requires_grad_()
    w = (torch.randn(K, C, R, S, generator=g) / np.sqrt(C * R * S)).requires_grad_()
    b = (torch.randn(K, generator=g) * 0.1).requires_grad_() if bias else None
    ref = F.conv2d(x.double(), w.double(), b.double() if bias else None, stride=(sh, sw), padding=(ph, pw), dilation=(dh, dw))
    dy = torch.randn(*ref.shape, generator=g)
    ref.backward(dy.double())
    xg = x.detach().permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_()
    wg = w.detach().to(DEV).requires_grad_()
    bg = b.detach().to(DEV).requires_grad_() if bias else None
    tag = f"conv N={N} HxW={H}x{W} C={C} K={K} {R}x{S} stride={stride} pad={pad} dil={dil} bias={bias}"
    try:
        y = av.conv(xg, wg, bg, stride=(sh, sw), pad=(ph, pw), dil=(dh, dw))
        y.backward(dy.permute(0, 2, 3, 1).contiguous().to(DEV))
        torch.cuda.synchronize()
        _lib.check_range(sync=True)
        e = [rel(y.detach().permute(0, 3, 1, 2), ref.detach()), rel(xg.grad.permute(0, 3, 1, 2), x.grad), rel(wg.grad, w.grad)]
        if bias:
            e.append(rel(bg.grad, b.grad))
        ok = e[0] < 2e-5 and max(e[1:]) < 1e-4
    except Exception as ex:          # an argument the path refuses is fine if it says so; anything else is a find
        e, ok = [float("nan")], isinstance(ex, (ValueError, NotImplementedError))
        tag += f" -> {type(ex).__name__}: {str(ex)[:90]}"
    print(f"{tag}: " + " ".join(f"{v:.2e}" for v in e) + ("" if ok else "   <-- OUTSIDE"), flush=True)
    bad += not ok
    if ok and max(e) == max(e) and max(e) > worst[0]:
        worst = (max(e), tag)
    # ---- TDNN block (every third shape with more than 4 096 rows: the BatchNorm backward formed on load, autograd.BN_BWD_ON_LOAD)
    B, T = int(r.integers(1, 12)), int(r.integers(12, 120))
    if i % 3 == 2:
        B, T = int(r.integers(24, 64)), int(r.integers(120, 260))
    C = pick([24, 32, 64, 80, 128, 256, 512])
    K = pick([32, 64, 100, 128, 512, 1500])
    S = pick([1, 3, 5]); dil = pick([1, 2, 3]) if S > 1 else 1
    act_first = bool(r.integers(0, 2))
    if T - dil * (S - 1) < 2:
        continue
    x = torch.randn(B, C, T, generator=g).requires_grad_()
    w = (torch.randn(K, C, S, generator=g) / np.sqrt(C * S)).requires_grad_()
    b = (torch.randn(K, generator=g) * 0.1).requires_grad_()
    gamma = (torch.rand(K, generator=g) + 0.5).requires_grad_()
    beta = (torch.randn(K, generator=g) * 0.2).requires_grad_()
    z = F.conv1d(x.double(), w.double(), b.double(), dilation=dil)
    if act_first:
        ref = F.batch_norm(F.leaky_relu(z, 0.2), None, None, gamma.double(), beta.double(), training=True, eps=1e-5)
    else:
        ref = F.leaky_relu(F.batch_norm(z, None, None, gamma.double(), beta.double(), training=True, eps=1e-5), 0.2)
    dy = torch.randn(*ref.shape, generator=g)
    ref.backward(dy.double())
    xg = x.detach().permute(0, 2, 1).contiguous().to(DEV).requires_grad_()
    wg_, bg, gg, beg = (t.detach().to(DEV).requires_grad_() for t in (w, b, gamma, beta))
    rm, rv = torch.zeros(K, device=DEV), torch.ones(K, device=DEV)
    tag = f"tdnn B={B} T={T} C={C} K={K} S={S} dil={dil} act_first={act_first}"
    try:
        y = ag.TDNNBlockTrainFn.apply(xg, wg_, bg, gg, beg, rm, rv, 0.1, 1e-5, 0.2, dil, act_first)
        y.backward(dy.permute(0, 2, 1).contiguous().to(DEV))
        torch.cuda.synchronize()
        _lib.check_range(sync=True)
        e = [rel(y.detach().permute(0, 2, 1), ref.detach()), rel(xg.grad.permute(0, 2, 1), x.grad), rel(wg_.grad, w.grad), rel(gg.grad, gamma.grad),
             rel(beg.grad, beta.grad)]
        ok = e[0] < 2e-5 and max(e[1:]) < 1e-4
        if not ok and e[0] < 2e-5 and max(e[1:]) < 0.2:
            # A LeakyReLU pre-activation that rounding moves across zero changes the gradient THERE by a finite amount (dbeta is a sum of
            # +-1-sized terms of magnitude sqrt(M): one flipped term is 1 %).  Count them: elements of the fp64 pre-activation within 2e-6 of
            # zero, and -- where the output shows the branch -- output signs that differ from fp64's.
            pre = z if act_first else F.batch_norm(z, None, None, gamma.double(), beta.double(), training=True, eps=1e-5)
            near = int((pre.detach().abs() < 2e-6).sum())
            flips = -1 if act_first else int(((y.detach().cpu().permute(0, 2, 1).double() >= 0) != (ref.detach() >= 0)).sum())
            if near > 0:
                tag += f" [kinks: {near} pre-activations within 2e-6 of zero, {flips} output signs differ]"
                ok, kinked = True, kinked + 1
    except Exception as ex:
        e, ok = [float("nan")], isinstance(ex, (ValueError, NotImplementedError))
        tag += f" -> {type(ex).__name__}: {str(ex)[:90]}"
    print(f"{tag}: " + " ".join(f"{v:.2e}" for v in e) + ("" if ok else "   <-- OUTSIDE"), flush=True)
    bad += not ok
    if ok and max(e) == max(e) and max(e) > worst[0] and "[kinks" not in tag:
        worst = (max(e), tag)
print(f"{kinked} shape(s) with a flipped LeakyReLU kink (gradient off by up to a few per cent at those elements, in any fp32 arithmetic)")
print(f"worst {worst[0]:.3e} at {worst[1]}; {bad} outside")
sys.exit(1 if bad else 0)

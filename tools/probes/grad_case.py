"""One TDNN-block shape of tools/probes/grad_fuzz.py, BatchNorm-backward-on-load on and off, against fp64 autograd -- and how many LeakyReLU
pre-activations sit within 2e-6 of zero (each such element may take the other branch in fp32: a finite change of the gradient there).
   python tools/probes/grad_case.py B T C K S dil act_first [seed]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import torch
import torch.nn.functional as F
from deeplip_amd import autograd as ag

B, T, C, K, S, dil, act_first = (int(v) for v in sys.argv[1:8])
seed = int(sys.argv[8]) if len(sys.argv) > 8 else 0
g = torch.Generator().manual_seed(seed)
x = torch.randn(B, C, T, generator=g).requires_grad_()
w = (torch.randn(K, C, S, generator=g) / np.sqrt(C * S)).requires_grad_()
b = (torch.randn(K, generator=g) * 0.1).requires_grad_()
gamma = (torch.rand(K, generator=g) + 0.5).requires_grad_()
beta = (torch.randn(K, generator=g) * 0.2).requires_grad_()
z = F.conv1d(x.double(), w.double(), b.double(), dilation=dil)
if act_first:
    pre = z
    ref = F.batch_norm(F.leaky_relu(z, 0.2), None, None, gamma.double(), beta.double(), training=True, eps=1e-5)
else:
    pre = F.batch_norm(z, None, None, gamma.double(), beta.double(), training=True, eps=1e-5)
    ref = F.leaky_relu(pre, 0.2)
dy = torch.randn(*ref.shape, generator=g)
ref.backward(dy.double())
near = int((pre.detach().abs() < 2e-6).sum())
print(f"pre-activations within 2e-6 of zero: {near} of {pre.numel()}")


def rel(a, b_):
    return float((a.double().cpu() - b_.double()).abs().max() / b_.double().abs().max())


for fused in (True, False):
    ag.BN_BWD_ON_LOAD = fused
    xg = x.detach().permute(0, 2, 1).contiguous().cuda().requires_grad_()
    wg_, bg, gg, beg = (t.detach().cuda().requires_grad_() for t in (w, b, gamma, beta))
    rm, rv = torch.zeros(K, device="cuda"), torch.ones(K, device="cuda")
    y = ag.TDNNBlockTrainFn.apply(xg, wg_, bg, gg, beg, rm, rv, 0.1, 1e-5, 0.2, dil, bool(act_first))
    y.backward(dy.permute(0, 2, 1).contiguous().cuda())
    torch.cuda.synchronize()
    # where does the engine's activation branch differ from fp64's?
    yb = y.detach().cpu().permute(0, 2, 1).double()
    flips = int(((yb >= 0) != (ref.detach() >= 0)).sum()) if not act_first else -1
    print(f"on_load={fused}: y {rel(y.detach().permute(0, 2, 1), ref.detach()):.2e} dx {rel(xg.grad.permute(0, 2, 1), x.grad):.2e} dW {rel(wg_.grad, w.grad):.2e} "
          f"dgamma {rel(gg.grad, gamma.grad):.2e} dbeta {rel(beg.grad, beta.grad):.2e}; output signs that differ from fp64: {flips}")

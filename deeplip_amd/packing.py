"""Weight packing for the HIP engine: eval-mode BatchNorm folded into the preceding conv/linear
in fp64 on the host (rounded to fp32 once), weights permuted to the kernels' KRSC layout.
Runs once per ``load_state_dict`` / device move (cached by the owning module)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch

from .holders import BatchNormParams

Tensor = torch.Tensor


@dataclass
class Packed:
    w: Tensor                      # kernel-layout weights (device, fp32)
    b: Optional[Tensor] = None     # folded bias
    slope: Optional[Tensor] = None  # per-channel negative slope (PReLU weight / 0.2 / 0)
    post_scale: Optional[Tensor] = None
    post_shift: Optional[Tensor] = None


def bn_scale_shift(bn: BatchNormParams):
    """Eval BatchNorm as y = x*scale + shift, in fp64."""
    g = bn.weight.detach().double().cpu()
    beta = bn.bias.detach().double().cpu()
    mean = bn.running_mean.detach().double().cpu()
    var = bn.running_var.detach().double().cpu()
    scale = g / torch.sqrt(var + bn.eps)
    return scale, beta - mean * scale


def fold(weight: Tensor, bias: Optional[Tensor], bn: Optional[BatchNormParams]):
    """(w, b) of conv/linear followed by eval BN -> folded fp64 (w', b')."""
    w = weight.detach().double().cpu()
    b = bias.detach().double().cpu() if bias is not None else torch.zeros(w.shape[0], dtype=torch.float64)
    if bn is not None:
        scale, shift = bn_scale_shift(bn)
        w = w * scale.view(-1, *([1] * (w.dim() - 1)))
        b = b * scale + shift
    return w, b


def _dev(t: Tensor, device) -> Tensor:
    return t.to(dtype=torch.float32).contiguous().to(device)


def pad_channels(c: int, mult: int = 4) -> int:
    return (c + mult - 1) // mult * mult


def pack_conv2d(weight, bias, bn, device, slope=None) -> Packed:
    """[K,C,R,S] -> KRSC."""
    w, b = fold(weight, bias, bn)
    return Packed(_dev(w.permute(0, 2, 3, 1), device), _dev(b, device), slope)


def pack_conv1d(weight, bias, bn, device, slope=None, cin_pad: Optional[int] = None) -> Packed:
    """[K,C,S] -> [K,S,Cp] (input channels zero-padded to Cp)."""
    w, b = fold(weight, bias, bn)
    w = w.permute(0, 2, 1)
    if cin_pad is not None and cin_pad != w.shape[2]:
        wp = torch.zeros(w.shape[0], w.shape[1], cin_pad, dtype=torch.float64)
        wp[:, :, :w.shape[2]] = w
        w = wp
    return Packed(_dev(w, device), _dev(b, device), slope)


def pack_linear(weight, bias, bn, device, slope=None) -> Packed:
    w, b = fold(weight, bias, bn)
    return Packed(_dev(w, device), _dev(b, device), slope)


def pack_stem3d(weight, bn, device, slope=None) -> Packed:
    """[64,1,5,7,7] -> k-major [248,64] (245 taps + 3 zero rows)."""
    w, b = fold(weight, None, bn)
    K = w.shape[0]
    wp = torch.zeros(248, K, dtype=torch.float64)
    wp[:245] = w.reshape(K, 245).t()
    return Packed(_dev(wp, device), _dev(b, device), slope)


def const_slope(k: int, value: float, device) -> Tensor:
    return torch.full((k,), value, dtype=torch.float32, device=device)


def state_version(module: torch.nn.Module, device) -> tuple:
    """Cheap fingerprint that changes on load_state_dict / in-place updates / device moves."""
    v = [str(device)]
    for t in list(module.parameters()) + list(module.buffers()):
        v.append((t._version, t.data_ptr()))
    return tuple(v)

"""Weight packing for the HIP engine: eval-mode BatchNorm folded into the preceding conv/linear
in fp64 on the host (rounded to fp32 once), weights permuted to the kernels' KRSC layout.
Runs once per ``load_state_dict`` / device move (cached by the owning module)."""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Optional

import torch

from . import holders
from .holders import BatchNormParams

Tensor = torch.Tensor


# Arithmetic mode of the implicit-GEMM kernels for weights packed from now on:
#   "f32"   exact fp32 MFMA (bit-reproducible fma chains)
#   "f16x3" split (hi, lo) fp16 operands, 3 f16 MFMAs per product, fp32 accumulate (~2^-22 relative)
# The library's default is the MEASURED configuration: "f16x3" under the arith mode "auto" (deeplip_amd/arith.py: what leaves its
# range is computed again in f32) unless $DLIP_ARITH says f32; the entry points call arith.configure() with their flag / config key.
PRECISION = "f32" if os.environ.get("DLIP_ARITH", "auto").strip().lower() == "f32" else "f16x3"


def set_precision(mode: str) -> None:
    global PRECISION
    if mode not in ("f32", "f16x3"):
        raise ValueError(mode)
    PRECISION = mode


@dataclass
class Packed:
    w: Tensor                      # kernel-layout weights (device): fp32 KRSC, or split fp16 pairs
    b: Optional[Tensor] = None     # folded bias
    slope: Optional[Tensor] = None  # per-channel negative slope (PReLU weight / 0.2 / 0)
    post_scale: Optional[Tensor] = None
    post_shift: Optional[Tensor] = None
    wscale: Optional[Tensor] = None  # f16x3 only: per-output-channel power-of-two weight scale


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


def split_weights(w: Tensor):
    """fp64 [K, ..., C] (channels last) -> (float32 view [K, ..., C32] of (hi, lo) fp16 pairs, scale [K]).

    Per output channel k the weights are multiplied by 2^e_k so that max|w_k| lands in [512, 1024):
    hi = fp16(w) keeps 11 bits and lo = fp16(w - hi) stays a NORMAL fp16 for every weight down to
    ~1e-4 of the channel maximum (unscaled, lo of a 0.03 weight would be subnormal and lose bits).
    Layout per 32-channel block: 32 hi halves then 32 lo halves (128 B, as one fp32 block)."""
    K, C = w.shape[0], w.shape[-1]
    C32 = (C + 31) // 32 * 32
    flat = w.reshape(K, -1)
    amax = flat.abs().amax(dim=1).clamp_min(1e-30)
    scale = torch.pow(2.0, torch.floor(torch.log2(1023.0 / amax))).to(torch.float64)
    ws = w * scale.view(K, *([1] * (w.dim() - 1)))
    if C32 != C:
        pad = torch.zeros(*w.shape[:-1], C32 - C, dtype=torch.float64)
        ws = torch.cat([ws, pad], dim=-1)
    hi = ws.to(torch.float16)
    lo = (ws - hi.to(torch.float64)).to(torch.float16)
    lead = ws.shape[:-1]
    blk = torch.stack([hi.reshape(*lead, C32 // 32, 32), lo.reshape(*lead, C32 // 32, 32)], dim=-2)  # [..., nb, 2, 32]
    packed = blk.contiguous().view(torch.float32).reshape(*lead, C32)       # 64 halves = 32 dwords per block
    return packed, scale.to(torch.float32)


def _finish(w64: Tensor, b64: Tensor, device, slope) -> "Packed":
    """w64: fp64 weights already in kernel layout [K, (R, S,) C]."""
    if PRECISION == "f16x3":
        ws, sc = split_weights(w64)
        return Packed(ws.contiguous().to(device), _dev(b64, device), slope, wscale=sc.to(device))
    return Packed(_dev(w64, device), _dev(b64, device), slope)


def pad_channels(c: int, mult: int = 4) -> int:
    return (c + mult - 1) // mult * mult


def pack_conv2d(weight, bias, bn, device, slope=None) -> Packed:
    """[K,C,R,S] -> KRSC."""
    w, b = fold(weight, bias, bn)
    return _finish(w.permute(0, 2, 3, 1).contiguous(), b, device, slope)


def pack_conv2d_shortcut(weight, bn, down_weight, down_bn, device, slope=None) -> Packed:
    """conv2 [K,C,3,3] + bn2 and the shortcut's 1x1 stride-2 conv [K,C2,1,1] + BatchNorm (resnet.py:13-17) as ONE
    reduction: rows [K, 9*C32 | C2] under one per-channel power-of-two scale, bias = sum of the folded biases
    (dlip_conv2_nhwc_f16x3).  f16x3 packing only."""
    if PRECISION != "f16x3":
        raise ValueError("pack_conv2d_shortcut: split-fp16 packing only")
    w, b = fold(weight, None, bn)
    wd, bd = fold(down_weight, None, down_bn)
    K, C = w.shape[0], w.shape[1]
    C2 = wd.shape[1]
    if C % 32 or C2 % 32:
        raise ValueError("pack_conv2d_shortcut: channel counts must be multiples of 32")
    rows = torch.cat([w.permute(0, 2, 3, 1).reshape(K, -1), wd.reshape(K, C2)], dim=1)   # [K, 9C + C2], 32-blocks intact
    ws, sc = split_weights(rows)
    return Packed(ws.contiguous().to(device), _dev(b + bd, device), slope, wscale=sc.to(device))


def pack_conv1d(weight, bias, bn, device, slope=None, cin_pad: Optional[int] = None) -> Packed:
    """[K,C,S] -> [K,S,Cp] (input channels zero-padded to Cp)."""
    w, b = fold(weight, bias, bn)
    w = w.permute(0, 2, 1)
    if cin_pad is not None and cin_pad != w.shape[2]:
        wp = torch.zeros(w.shape[0], w.shape[1], cin_pad, dtype=torch.float64)
        wp[:, :, :w.shape[2]] = w
        w = wp
    return _finish(w.contiguous(), b, device, slope)


def pack_linear(weight, bias, bn, device, slope=None) -> Packed:
    w, b = fold(weight, bias, bn)
    return _finish(w.contiguous(), b, device, slope)


def split_stem_weights(w: Tensor):
    """fp64 [64,1,5,7,7] -> (float32 view of the 64 x 1184-byte LDS image of stem3d_f16x3.hip, scale [64]).
    Per channel: [36 kernel rows (kt*7+kh, row 35 zero) x 8 hi halves (tap 7 zero)][36 x 8 lo halves] + 32 B pad
    (hi / lo planes and the 74-slot channel stride keep the kernel's fragment reads bank-conflict free)."""
    K = w.shape[0]
    rows = torch.zeros(K, 36, 8, dtype=torch.float64)
    rows[:, :35, :7] = w.reshape(K, 35, 7)
    amax = rows.reshape(K, -1).abs().amax(dim=1).clamp_min(1e-30)
    scale = torch.pow(2.0, torch.floor(torch.log2(1023.0 / amax))).to(torch.float64)
    ws = rows * scale.view(K, 1, 1)
    hi = ws.to(torch.float16)
    lo = (ws - hi.to(torch.float64)).to(torch.float16)
    img = torch.zeros(K, 1184 // 2, dtype=torch.float16)
    img[:, :36 * 8] = hi.reshape(K, 36 * 8)
    img[:, 36 * 8:36 * 16] = lo.reshape(K, 36 * 8)
    return img.contiguous().view(torch.float32).reshape(-1), scale.to(torch.float32)


def pack_stem3d(weight, bn, device, slope=None) -> Packed:
    """[64,1,5,7,7] -> k-major [248,64] (245 taps + 3 zero rows); f16x3: the split LDS image."""
    w, b = fold(weight, None, bn)
    K = w.shape[0]
    if PRECISION == "f16x3":
        img, sc = split_stem_weights(w)
        return Packed(img.to(device), _dev(b, device), slope, wscale=sc.to(device))
    wp = torch.zeros(248, K, dtype=torch.float64)
    wp[:245] = w.reshape(K, 245).t()
    return Packed(_dev(wp, device), _dev(b, device), slope)


def const_slope(k: int, value: float, device) -> Tensor:
    return torch.full((k,), value, dtype=torch.float32, device=device)


def invalidate() -> None:
    """Drop every packed-weight cache (and mark recorded step plans stale).  Holders call this themselves on
    load_state_dict / device moves / train-eval switches; call it after editing parameters in place by hand
    through ``.data`` views that do not bump the tensors' version counters."""
    holders.invalidate_packs()


def state_version(module: torch.nn.Module, device) -> tuple:
    """Fingerprint of a module's packed state: (pack generation, device, arithmetic mode, sum of the tensors'
    in-place version counters).  The tensor list is cached per generation, so a forward costs one pass over
    ``_version`` attributes (tens of microseconds for the 343-tensor lip-clip model) instead of a walk over
    the module tree; recorded step plans (plan.py) skip even that."""
    gen = holders.PACK_GEN[0]
    c = module.__dict__.get("_dlip_tensors")
    if c is None or c[0] != gen:
        c = (gen, list(module.parameters()) + list(module.buffers()))
        module.__dict__["_dlip_tensors"] = c
    v = 0
    for t in c[1]:
        v += t._version + (t.data_ptr() & 0xFFFFFFFF)     # in-place updates bump _version; `p.data = other` moves data_ptr
    return (gen, device, PRECISION, v)

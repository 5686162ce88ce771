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


def _finish(w64: Tensor, b64: Tensor, device, slope, e_in: int = 0, e_out: int = 0) -> "Packed":
    """w64: fp64 weights already in kernel layout [K, (R, S,) C].  ``e_in`` / ``e_out`` (f16x3 packs only): the ACTIVATION EXPONENTS of
    the layer's input and output tensors -- the stored split tensor is 2^e times the true one (see act_exponents below).  A layer
    y = act(W x + b) with a positively homogeneous activation (ReLU / PReLU / LeakyReLU) maps stored input to stored output with
    W' = 2^(e_out - e_in) W and b' = 2^e_out b: exact powers of two, folded here -- the split weight bits do not change at all (their
    per-channel power-of-two scale absorbs the factor), no kernel knows about exponents."""
    if PRECISION == "f16x3":
        if e_in or e_out:
            w64 = w64 * (2.0 ** int(e_out - e_in))
            b64 = b64 * (2.0 ** int(e_out))
        ws, sc = split_weights(w64)
        return Packed(ws.contiguous().to(device), _dev(b64, device), slope, wscale=sc.to(device))
    return Packed(_dev(w64, device), _dev(b64, device), slope)


def pad_channels(c: int, mult: int = 4) -> int:
    return (c + mult - 1) // mult * mult


def pack_conv2d(weight, bias, bn, device, slope=None, e_in: int = 0, e_out: int = 0) -> Packed:
    """[K,C,R,S] -> KRSC."""
    w, b = fold(weight, bias, bn)
    return _finish(w.permute(0, 2, 3, 1).contiguous(), b, device, slope, e_in, e_out)


def pack_conv2d_shortcut(weight, bn, down_weight, down_bn, device, slope=None, e_h: int = 0, e_x: int = 0, e_out: int = 0) -> Packed:
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
    # activation exponents: conv2 reads h (e_h), the shortcut reads the block's input (e_x), both land in the output's (e_out)
    rows = torch.cat([w.permute(0, 2, 3, 1).reshape(K, -1) * (2.0 ** int(e_out - e_h)), wd.reshape(K, C2) * (2.0 ** int(e_out - e_x))], dim=1)   # [K, 9C + C2], 32-blocks intact
    ws, sc = split_weights(rows)
    return Packed(ws.contiguous().to(device), _dev((b + bd) * (2.0 ** int(e_out)), device), slope, wscale=sc.to(device))


def pack_conv1d(weight, bias, bn, device, slope=None, cin_pad: Optional[int] = None, e_in: int = 0, e_out: int = 0) -> Packed:
    """[K,C,S] -> [K,S,Cp] (input channels zero-padded to Cp)."""
    w, b = fold(weight, bias, bn)
    w = w.permute(0, 2, 1)
    if cin_pad is not None and cin_pad != w.shape[2]:
        wp = torch.zeros(w.shape[0], w.shape[1], cin_pad, dtype=torch.float64)
        wp[:, :, :w.shape[2]] = w
        w = wp
    return _finish(w.contiguous(), b, device, slope, e_in, e_out)


def pack_linear(weight, bias, bn, device, slope=None, e_in: int = 0, e_out: int = 0) -> Packed:
    w, b = fold(weight, bias, bn)
    return _finish(w.contiguous(), b, device, slope, e_in, e_out)


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


def pack_stem3d(weight, bn, device, slope=None, e_in: int = 0, e_out: int = 0) -> Packed:
    """[64,1,5,7,7] -> k-major [248,64] (245 taps + 3 zero rows); f16x3: the split LDS image."""
    w, b = fold(weight, None, bn)
    K = w.shape[0]
    if PRECISION == "f16x3":
        if e_in or e_out:
            w = w * (2.0 ** int(e_out - e_in))
            b = b * (2.0 ** int(e_out))
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
    v = 7919 * module.__dict__.get("_dlip_exp_ver", 0)     # the activation exponents folded into the f16x3 pack (set_act_exponents)
    for t in c[1]:
        v += t._version + (t.data_ptr() & 0xFFFFFFFF)     # in-place updates bump _version; `p.data = other` moves data_ptr
    return (gen, device, PRECISION, v)


# ---- activation exponents of the f16x3 packs -------------------------------------------------------------------------------------
# The split format holds tensors whose largest |v| lies in [2^-2 .. 65520) at fp32 grade.  A checkpoint whose activations live elsewhere -- BatchNorm statistics
# that shrink or grow a layer's output, an input with a gain of 2^+-20 -- used to RAISE (DeepLipRangeError) and, under arith "auto",
# had every batch computed again in exact fp32.  Now the exact re-run of the FIRST such batch doubles as a calibration: it measures
# the largest magnitude of every tensor the f16x3 path stores split (dlip_pow2_scale_f32 on the fp32 tensors of the exact path, one
# synchronisation at the end), and each gets a power-of-two exponent e that puts its largest stored magnitude at 2^11 .. 2^12.  The
# exponents are folded into the packed weights and biases (see _finish: exact, no kernel changes, in-range models keep e = 0 and
# today's bits); tensors that meet in a residual addition share one exponent (the smallest of the group).  Outputs that leave the
# engine as fp32 are brought back by one multiplication with 2^-e, or not at all where the consumer is scale invariant (z-norm).
ACT_TARGET = 4096.0            # 2^12: a factor 16 below fp16's largest value for batches that run hotter than the calibration batch
MAX_CALIBRATIONS = 3           # per model; afterwards an out-of-range batch is simply computed in f32 (arith auto)
CALIB = None                   # while a calibrating exact pass runs: {id(module): (module, {tensor name: device pair from pow2_scale})}


def act_exponents(module) -> dict:
    """{tensor name: e} of a model's f16x3 pack ({} = all zero: nothing was ever calibrated)."""
    if module.__dict__.get("_dlip_exp_load_gen") != holders.LOAD_GEN[0]:
        return {}               # measured on other weights (a load_state_dict since): void, and the calibration budget starts again
    return module.__dict__.get("_dlip_exp", {})


def set_act_exponents(module, exps: dict) -> None:
    if module.__dict__.get("_dlip_exp_load_gen") != holders.LOAD_GEN[0]:
        module.__dict__["_dlip_calibrations"] = 0
    module.__dict__["_dlip_exp_load_gen"] = holders.LOAD_GEN[0]
    module.__dict__["_dlip_exp"] = {k: int(v) for k, v in exps.items() if int(v) != 0}
    module.__dict__["_dlip_exp_ver"] = module.__dict__.get("_dlip_exp_ver", 0) + 1      # packs and recorded plans of the model are stale now


def calib_note(module, name: str, t: Tensor) -> None:
    """Inside a calibrating pass (CALIB is a dict): remember the power of two that would lift tensor ``t`` to ACT_TARGET."""
    if CALIB is None or not t.is_cuda:
        return
    from . import _lib
    ent = CALIB.setdefault(id(module), (module, {}))
    pair = torch.empty(2, device=t.device, dtype=torch.float32)
    tt = t if t.is_contiguous() else t.contiguous()
    _lib.check(_lib.lib().dlip_pow2_scale_f32(tt.data_ptr(), pair.data_ptr(), tt.numel(), float(ACT_TARGET), _lib.stream_handle()), "dlip_pow2_scale_f32")
    prev = ent[1].get(name)
    ent[1][name] = pair if prev is None else torch.minimum(prev, pair)      # (a name noted twice in one pass: the hotter tensor rules)


def calib_finish(groups_of=None) -> int:
    """End of a calibrating pass: read the noted scales (one synchronisation), turn them into exponents, apply each model's
    residual groups (``module.act_exponent_groups()``: lists of names that must share an exponent) and install them.  Returns the number
    of models whose exponents changed."""
    import math
    global CALIB
    notes, CALIB = CALIB, None
    changed = 0
    if not notes:
        return 0
    torch.cuda.synchronize()
    for module, named in notes.values():
        fresh = module.__dict__.get("_dlip_exp_load_gen") != holders.LOAD_GEN[0]
        if not fresh and module.__dict__.get("_dlip_calibrations", 0) >= MAX_CALIBRATIONS:
            continue
        exps = {}
        for name, pair in named.items():
            s = float(pair[0])
            exps[name] = int(round(math.log2(s))) if (s > 0 and math.isfinite(s)) else 0
        for grp in (module.act_exponent_groups() if hasattr(module, "act_exponent_groups") else []):
            have = [exps[n] for n in grp if n in exps]
            if have:
                for n in grp:
                    exps[n] = min(have)
        exps = {k: max(-100, min(100, v)) for k, v in exps.items()}
        # Tensors that sit comfortably inside the format keep e = 0 (and with it the bits they have always had): a largest magnitude
        # of 2^-2 .. 2^14 gives e in [-2, 14] -- overflow is a factor 4 away at least, and the low-side guard (largest magnitude
        # below 2^-2) is where the window ends: a batch that runs colder than the calibration batch is reported and computed in f32,
        # never silently less exact.  Only what the calibration found outside that window is moved (to 2^12: a factor 2^14 above the guard).
        exps = {k: (0 if -2 <= v <= 14 else v) for k, v in exps.items()}
        if {k: v for k, v in exps.items() if v} != act_exponents(module):
            set_act_exponents(module, exps)            # (resets the budget when the weights are new)
            module.__dict__["_dlip_calibrations"] = module.__dict__.get("_dlip_calibrations", 0) + 1
            changed += 1
    return changed

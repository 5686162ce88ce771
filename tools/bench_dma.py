#!/usr/bin/env python3
"""Per-layer timing of the LDS-DMA split-fp16 conv kernel across its tile menu (development tool).
Activations, residual and output in the split activation format, as inside the encoders."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from deeplip_amd import _lib, ops, packing

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--only", default="")
ap.add_argument("--variants", default="-1", help="comma list of DMA tile ids (dlip_debug_set; -1 = built-in choice; 6..9 need the lab build via DLIP_LIB_PATH)")
ap.add_argument("--streamk", type=int, default=-1, help="balanced split: -1 built-in cost model, 0 never, 2 always (dlip_debug_set)")
ap.add_argument("--ninner", type=int, default=-1, help="1: tile order with the output-channel block inner (experiment)")
ap.add_argument("--win", type=int, default=-1, help="window kernel: -1 built-in rule, 0 off (ring kernel everywhere), 1 K <= 64 only")
ap.add_argument("--check", action="store_true", help="compare every variant's output with the first variant's")
ap.add_argument("--xpad", type=int, default=0, help="extra (unused) channels per input pixel: breaks the power-of-two pixel stride")
a = ap.parse_args()
B = a.batch
N = B * 29
L = [
    ("l1.conv", (N, 22, 22, 64), 64, 3, 3, 1, 1, 1, True, 4),
    ("l2.conv1s2", (N, 22, 22, 64), 128, 3, 3, 2, 1, 1, False, 1),
    ("l2.down", (N, 22, 22, 64), 128, 1, 1, 2, 0, 1, False, 1),
    ("l2.conv", (N, 11, 11, 128), 128, 3, 3, 1, 1, 1, True, 3),
    ("l3.conv1s2", (N, 11, 11, 128), 256, 3, 3, 2, 1, 1, False, 1),
    ("l3.down", (N, 11, 11, 128), 256, 1, 1, 2, 0, 1, False, 1),
    ("l3.conv", (N, 6, 6, 256), 256, 3, 3, 1, 1, 1, True, 3),
    ("l4.conv1s2", (N, 6, 6, 256), 512, 3, 3, 2, 1, 1, False, 1),
    ("l4.down", (N, 6, 6, 256), 512, 1, 1, 2, 0, 1, False, 1),
    ("l4.conv", (N, 3, 3, 512), 512, 3, 3, 1, 1, 1, True, 3),
    ("tdnn.k1", (B, 1, 296, 512), 512, 1, 1, 1, 0, 1, False, 4),
    ("tdnn.k3d2", (B, 1, 296, 512), 512, 1, 3, 1, 0, 2, False, 4),
    ("tdnn9", (B, 1, 278, 512), 1500, 1, 1, 1, 0, 1, False, 1),
]
variants = a.variants.split(",")
_lib.debug_set(_lib.DBG_STREAMK, a.streamk)
_lib.debug_set(5, a.ninner)
_lib.debug_set(_lib.DBG_WIN, a.win)
tot = {v: 0.0 for v in variants}
tot_best = 0.0
for name, (n, h, w, c), k, r, s, st, pd, dl, res, count in L:
    if a.only and a.only not in name:
        continue
    x = ops.split_pack(torch.randn(n, h, w, c + a.xpad, device="cuda"))
    wsp, wsc = packing.split_weights((torch.randn(k, r, s, c, dtype=torch.float64) * 0.05))
    wsp, wsc = wsp.cuda(), wsc.cuda()
    b = torch.randn(k, device="cuda")
    sl = torch.rand(k, device="cuda")
    sh = (1, st) if h == 1 else (st, st)
    pp = (0, pd) if h == 1 else (pd, pd)
    dd = (1, dl) if h == 1 else (dl, dl)
    osp = k % 32 == 0
    kw = dict(stride=sh, pad=pp, dil=dd, slope=sl, w_scale=wsc, x_split=True, out_split=osp)
    if a.xpad:
        kw["in_channels"] = c
    y = ops.conv_nhwc(x, wsp, b, **kw)
    rs = ops.split_pack(torch.randn_like(y)) if res else None
    fl = 2.0 * y.shape[0] * y.shape[1] * y.shape[2] * k * r * s * c
    best = {v: 1e30 for v in variants}
    if a.check:
        outs = []
        for v in variants:
            _lib.debug_set(_lib.DBG_DMA_TILE, int(v))
            outs.append(ops.conv_nhwc(x, wsp, b, residual=rs, out=torch.empty_like(y), **kw).clone())
        u0 = ops.split_unpack(outs[0]) if osp else outs[0]
        for v, o in zip(variants[1:], outs[1:]):
            u = ops.split_unpack(o) if osp else o
            print(f"   check v{v} vs v{variants[0]}: max|d|/max = {float((u - u0).abs().max() / u0.abs().max()):.2e}", flush=True)
    for rnd in range(3):
        for v in variants:
            _lib.debug_set(_lib.DBG_DMA_TILE, int(v))
            ops.conv_nhwc(x, wsp, b, residual=rs, out=y, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                ops.conv_nhwc(x, wsp, b, residual=rs, out=y, **kw)
            e1.record()
            torch.cuda.synchronize()
            best[v] = min(best[v], e0.elapsed_time(e1) * 1e3 / a.iters)
    _lib.debug_set(_lib.DBG_DMA_TILE, -1)
    for v in variants:
        tot[v] += best[v] * count
    tot_best += min(best.values()) * count
    print(f"{name:11s} x{count} " + "  ".join(f"v{v}:{best[v]:7.1f}us {fl / best[v] / 1e6:5.0f}TF" for v in variants) + f"  {fl / 1e9:6.2f} GF", flush=True)
print("sum(us x count): " + "  ".join(f"v{v}:{tot[v]:8.0f}" for v in variants) + f"   best-per-layer: {tot_best:8.0f}")

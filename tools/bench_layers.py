#!/usr/bin/env python3
"""Per-layer kernel timing on the GPU (development tool): every distinct conv/stem shape of the
fused A+V step at a given clip batch, timed with HIP events on random data."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from deeplip_amd import _lib, ops

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--only", default="")
ap.add_argument("--split", action="store_true", help="also time the split-fp16 (f16x3) kernel")
ap.add_argument("--variants", default="-1", help="comma list of tile ids of the fp32 / register-staged menu (dlip_debug_set) (-1 = built-in choice)")
a = ap.parse_args()
B = a.batch
N = B * 29
dev = "cuda"
# name, (N,H,W,C), K, R,S, stride, pad, dil, residual
L = [
    ("l1.conv", (N, 22, 22, 64), 64, 3, 3, 1, 1, 1, True),
    ("l2.conv1s2", (N, 22, 22, 64), 128, 3, 3, 2, 1, 1, False),
    ("l2.down", (N, 22, 22, 64), 128, 1, 1, 2, 0, 1, False),
    ("l2.conv", (N, 11, 11, 128), 128, 3, 3, 1, 1, 1, True),
    ("l3.conv1s2", (N, 11, 11, 128), 256, 3, 3, 2, 1, 1, False),
    ("l3.down", (N, 11, 11, 128), 256, 1, 1, 2, 0, 1, False),
    ("l3.conv", (N, 6, 6, 256), 256, 3, 3, 1, 1, 1, True),
    ("l4.conv1s2", (N, 6, 6, 256), 512, 3, 3, 2, 1, 1, False),
    ("l4.down", (N, 6, 6, 256), 512, 1, 1, 2, 0, 1, False),
    ("l4.conv", (N, 3, 3, 512), 512, 3, 3, 1, 1, 1, True),
    ("tdnn0", (B, 1, 300, 80), 512, 1, 5, 1, 0, 1, False),
    ("tdnn.k1", (B, 1, 296, 512), 512, 1, 1, 1, 0, 1, False),
    ("tdnn.k3d2", (B, 1, 296, 512), 512, 1, 3, 1, 0, 2, False),
    ("tdnn9", (B, 1, 278, 512), 1500, 1, 1, 1, 0, 1, False),
    ("fc1", (1, 1, B, 3000), 512, 1, 1, 1, 0, 1, False),
]
print(f"{'layer':12s} {'us':>9s} {'TFLOP/s':>8s} {'GFLOP':>8s}")
tot_us = tot_f = 0.0
for name, (n, h, w, c), k, r, s, st, pd, dl, res in L:
    if a.only and a.only not in name:
        continue
    x = torch.randn(n, h, w, c, device=dev)
    wt = torch.randn(k, r, s, c, device=dev) * 0.05
    b = torch.randn(k, device=dev)
    sl = torch.rand(k, device=dev)
    sh = (1, st) if h == 1 else (st, st)
    pp = (0, pd) if h == 1 else (pd, pd)
    dd = (1, dl) if h == 1 else (dl, dl)
    y = ops.conv_nhwc(x, wt, b, stride=sh, pad=pp, dil=dd, slope=sl)
    rs = torch.randn_like(y) if res else None
    for _ in range(3):
        ops.conv_nhwc(x, wt, b, stride=sh, pad=pp, dil=dd, slope=sl, residual=rs, out=y)
    fl = 2.0 * y.shape[0] * y.shape[1] * y.shape[2] * k * r * s * c
    variants = a.variants.split(",")
    best = {v: 1e30 for v in variants}
    for rnd in range(4):          # interleaved rounds, report the min per variant
        for v in variants:
            _lib.debug_set(_lib.DBG_CONV_TILE, int(v.split(":")[0]))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                ops.conv_nhwc(x, wt, b, stride=sh, pad=pp, dil=dd, slope=sl, residual=rs, out=y)
            e1.record()
            torch.cuda.synchronize()
            best[v] = min(best[v], e0.elapsed_time(e1) * 1e3 / a.iters)
    line = f"{name:12s} " + "  ".join(f"v{v}: {best[v]:8.1f}us {fl / best[v] / 1e6:6.1f}TF" for v in variants)
    if a.split:
        from deeplip_amd import packing
        wsp, wsc = packing.split_weights(wt.double().cpu())
        wsp, wsc = wsp.cuda(), wsc.cuda()
        bs = {v: 1e30 for v in variants}
        y2 = torch.empty_like(y)
        for rnd in range(4):
            for v in variants:
                _lib.debug_set(_lib.DBG_CONV_TILE, int(v.split(":")[0]))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    ops.conv_nhwc(x, wsp, b, stride=sh, pad=pp, dil=dd, slope=sl, residual=rs, out=y2, w_scale=wsc)
                e1.record()
                torch.cuda.synchronize()
                bs[v] = min(bs[v], e0.elapsed_time(e1) * 1e3 / a.iters)
        err = float((y2 - y).abs().max() / y.abs().max())
        line += "  | f16x3 " + "  ".join(f"v{v}: {bs[v]:8.1f}us {fl / bs[v] / 1e6:6.1f}TF" for v in variants) + f"  relerr {err:.1e}"
    print(line + f"  {fl / 1e9:8.2f} GF")
if not a.only or a.only == 'stem':
    x = torch.randn(B, 29, 88, 88, device=dev)
    wp = torch.randn(248, 64, device=dev) * 0.05
    b = torch.randn(64, device=dev); sl = torch.rand(64, device=dev)
    for _ in range(2):
        ops.stem3d(x, wp, b, sl)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        y = ops.stem3d(x, wp, b, sl)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / a.iters
    fl = 2.0 * B * 29 * 44 * 44 * 64 * 245
    print(f"{'stem3d':12s} {us:9.1f} {fl / us / 1e6:8.1f} {fl / 1e9:8.2f}")
    if a.split:
        from deeplip_amd import packing
        img, sc = packing.split_stem_weights(torch.randn(64, 1, 5, 7, 7, dtype=torch.float64) * 0.05)
        img, sc = img.cuda(), sc.cuda()
        for _ in range(2):
            ops.stem3d(x, img, b, sl, w_scale=sc)
        e0.record()
        for _ in range(a.iters):
            ops.stem3d(x, img, b, sl, w_scale=sc)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.iters
        print(f"{'stem3d f16x3':12s} {us:9.1f} {fl / us / 1e6:8.1f} {fl / 1e9:8.2f}")
    for _ in range(2):
        ops.maxpool3x3s2(y)
    e0.record()
    for _ in range(a.iters):
        ops.maxpool3x3s2(y)
    e1.record(); torch.cuda.synchronize()
    print(f"{'maxpool':12s} {e0.elapsed_time(e1) * 1e3 / a.iters:9.1f}")

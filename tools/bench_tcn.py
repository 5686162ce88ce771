#!/usr/bin/env python3
"""The MS-TCN head's convolutions at training shapes (B = 32 clips x 29 frames -> 928 + padding rows, 512 | 768 -> 256 channels,
k = 3 | 5 | 7, dilation 1..8) on the ring kernel's tile menu: these 48 forward / data-gradient launches of a training step are
latency-bound (16 tiles of 128 x 128 with reductions of 48-168 slices).  python tools/bench_tcn.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeplip_amd import _lib, ops, packing
B, T = 32, 29
shapes = [("l0 k3 d1 512", 512, 256, 3, 1), ("l0 k7 d1 512", 512, 256, 7, 1), ("l1 k5 d2 768", 768, 256, 5, 2), ("l3 k7 d8 768", 768, 256, 7, 8),
          ("dgrad k7 256->768", 256, 768, 7, 4)]
for name, C, K, S, dil in shapes:
    pad = (S - 1) * dil
    x = ops.split_pack(torch.randn(B, 1, T, C, device="cuda"))
    w = torch.randn(K, 1, S, C, dtype=torch.float64) * 0.03
    ws, sc = packing.split_weights(w)
    ws, sc = ws.cuda(), sc.cuda()
    b = torch.randn(K, device="cuda")
    res = []
    for tile, sk in ((-1, -1), (0, -1), (2, -1), (2, 2), (4, -1), (1, -1), (5, -1), (0, 0)):
        _lib.debug_set(_lib.DBG_DMA_TILE, tile); _lib.debug_set(_lib.DBG_STREAMK, sk)
        f = lambda: ops.conv_nhwc(x, ws, b, pad=(0, pad), dil=(1, dil), w_scale=sc, x_split=True)
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        res.append(f"t{tile}/sk{sk}:{e0.elapsed_time(e1) * 50:6.1f}")
    _lib.debug_set(_lib.DBG_DMA_TILE, -1); _lib.debug_set(_lib.DBG_STREAMK, -1)
    fl = 2.0 * B * (T + pad) * K * C * S
    print(f"{name:20s} {fl / 1e9:5.2f} GF  us: " + "  ".join(res), flush=True)

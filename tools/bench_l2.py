#!/usr/bin/env python3
"""Same-size 3x3 convolution at 128 -> 128 channels (layer 2, second block): 128-column window instance (dlip_debug_set
DLIP_DBG_WIN = 2, experiment) vs the ring kernel; correctness on a small batch first, then timing at B = 64 clips."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deeplip_amd import _lib, ops, packing

def run(N, H, C, K, res, w):
    _lib.debug_set(_lib.DBG_WIN, w)
    return ops.conv_nhwc(x, wsp, b, residual=res, out=y, **kw)

for (N, H, C, K) in [(40, 11, 128, 128), (64 * 29, 11, 128, 128)]:
    torch.manual_seed(0)
    x = ops.split_pack(torch.randn(N, H, H, C, device="cuda"))
    wsp, wsc = packing.split_weights(torch.randn(K, 3, 3, C, dtype=torch.float64) * 0.03)
    wsp, wsc = wsp.cuda(), wsc.cuda()
    b = torch.randn(K, device="cuda"); sl = torch.rand(K, device="cuda")
    kw = dict(pad=(1, 1), slope=sl, w_scale=wsc, x_split=True, out_split=True)
    y = torch.empty(N, H, H, K, device="cuda")
    rs = ops.split_pack(torch.randn(N, H, H, K, device="cuda"))
    for res in (rs, None):
        a = run(N, H, C, K, res, 2).clone(); r = run(N, H, C, K, res, 0).clone()
        torch.cuda.synchronize()
        ua, ur = ops.split_unpack(a), ops.split_unpack(r)
        print(f"N={N} res={res is not None}: max|win-ring|/max = {float((ua - ur).abs().max() / ur.abs().max()):.2e}", flush=True)
        if N < 100:
            continue
        for rnd in range(2):
            for w in (2, 0):
                run(N, H, C, K, res, w)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    run(N, H, C, K, res, w)
                e1.record(); torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 100
                print("  residual" if res is not None else "  plain   ", "window128" if w else "ring     ", f"{us:7.1f} us {2 * N * H * H * K * 9 * C / us / 1e6:5.0f} TF", flush=True)
_lib.debug_set(_lib.DBG_WIN, -1)

#!/usr/bin/env python3
"""Layer-1 convolution (64 -> 64, 3x3, 22x22 maps, B = 64 clips): window kernel vs ring kernel on one box, interleaved."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeplip_amd import _lib, ops, packing
N = 64 * 29
x = ops.split_pack(torch.randn(N, 22, 22, 64, device="cuda"))
wsp, wsc = packing.split_weights(torch.randn(64, 3, 3, 64, dtype=torch.float64) * 0.05)
wsp, wsc = wsp.cuda(), wsc.cuda()
b = torch.randn(64, device="cuda"); sl = torch.rand(64, device="cuda")
kw = dict(pad=(1, 1), slope=sl, w_scale=wsc, x_split=True, out_split=True)
y = ops.conv_nhwc(x, wsp, b, **kw); rs = ops.split_pack(torch.randn_like(y))
for res in (rs, None):
    for rnd in range(2):
        for w in (-1, 0):
            _lib.debug_set(_lib.DBG_WIN, w)
            ops.conv_nhwc(x, wsp, b, residual=res, out=y, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.conv_nhwc(x, wsp, b, residual=res, out=y, **kw)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 100
            print("residual" if res is not None else "plain   ", {-1: "window", 0: "ring  "}[w], f"{us:7.1f} us {2 * N * 484 * 64 * 576 / us / 1e6:5.0f} TF", flush=True)
_lib.debug_set(_lib.DBG_WIN, -1)

#!/usr/bin/env python3
"""Window-kernel geometries on one box (dlip_debug_set(DLIP_DBG_WIN, v)): layer 1 (64 -> 64, 22x22 maps) on the 128x64 tile with
2 x 2 waves of 64x32 (built-in until round 6) against ONE-column layouts whose waves hold 64x64 tiles (v = 3: 128x64, two waves;
v = 4: 256x64, four waves); layer 2.1 (128 -> 128, 11x11) on 128x128 with 4 x 2 waves of 32x64 against 2 x 2 of 64x64 (v = 5) and
256x128 with 4 x 2 of 64x64 (v = 6).  Interleaved rounds; every geometry must give the built-in one's bits."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeplip_amd import _lib, ops, packing
B = int(os.environ.get("B", 64))
N = B * 29
torch.manual_seed(0)


def case(C, HW, modes):
    x = ops.split_pack(torch.randn(N, HW, HW, C, device="cuda"))
    wsp, wsc = packing.split_weights(torch.randn(C, 3, 3, C, dtype=torch.float64) * 0.05)
    wsp, wsc = wsp.cuda(), wsc.cuda()
    b = torch.randn(C, device="cuda"); sl = torch.rand(C, device="cuda")
    kw = dict(pad=(1, 1), slope=sl, w_scale=wsc, x_split=True, out_split=True)
    _lib.debug_set(_lib.DBG_WIN, -1)
    y0 = ops.conv_nhwc(x, wsp, b, **kw).clone()
    rs = ops.split_pack(torch.randn(N, HW, HW, C, device="cuda"))
    y0r = ops.conv_nhwc(x, wsp, b, residual=rs, **kw).clone()
    fl = 2.0 * N * HW * HW * C * C * 9
    for res, ref in ((None, y0), (rs, y0r)):
        for rnd in range(3):
            for v in modes:
                _lib.debug_set(_lib.DBG_WIN, v)
                y = ops.conv_nhwc(x, wsp, b, residual=res, **kw)
                same = bool(torch.equal(y, ref))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    ops.conv_nhwc(x, wsp, b, residual=res, out=y, **kw)
                e1.record(); torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 100
                print(f"C={C} {HW}x{HW} {'residual' if res is not None else 'plain   '} win={v:2d} {us:7.1f} us {fl / us / 1e6:5.0f} TF  frac {fl / us / 1e6 / 833.3:.3f}  same_bits={same}", flush=True)
    _lib.debug_set(_lib.DBG_WIN, -1)


case(64, 22, (-1, 3, 4))
case(128, 11, (-1, 5, 6))

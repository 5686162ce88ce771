#!/usr/bin/env python3
"""tdnn.9's pooled launch (512 -> 1500 channels, k = 1, statistics pooling in the epilogue) on every kernel / tile that can run it:
the ring kernel's 128 x 128 and 256 x 128 pooled instances (dlip_debug_set(1, 0 | 5)), the rows kernel's pooled epilogue
(dlip_debug_set(6, 3 | 4 | 5)).    python tools/probes/tdnn9_pool.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from deeplip_amd import _lib, ops, packing
wsp, wsc = packing.split_weights(torch.randn(1500, 1, 1, 512, dtype=torch.float64) * 0.05)
wsp, wsc = wsp.cuda(), wsc.cuda()
b = torch.randn(1500, device="cuda"); sl = torch.full((1500,), 0.2, device="cuda")
for B in (64, 256):
    T = 278
    x = ops.split_pack(torch.randn(B, 1, T, 512, device="cuda"))
    fl = 2.0 * B * T * 1500 * 512
    for name, key, val in (("built-in", None, None), ("ring 128x128", _lib.DBG_DMA_TILE, 0), ("ring 256x128", _lib.DBG_DMA_TILE, 5),
                           ("rows mi=3", _lib.DBG_ROWS, 3), ("rows mi=4", _lib.DBG_ROWS, 4), ("rows mi=5", _lib.DBG_ROWS, 5)):
        try:
            if key is not None:
                _lib.debug_set(key, val)
            p = ops.conv_pool(x, wsp, b, wsc, T, slope=sl)
            y = ops.pool_finish(p, "meanstd")
            best = 1e9
            for rnd in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    ops.conv_pool(x, wsp, b, wsc, T, slope=sl)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 100)
            print(f"B={B:3d} {name:13s} tile rows {p.tile_rows:3d}  {best:7.1f} us  {fl / best / 1e6:5.0f} TF  checksum {float(y.double().sum()):.4f}", flush=True)
        except Exception as ex:
            print(f"B={B:3d} {name:13s} failed: {ex}", flush=True)
        finally:
            if key is not None:
                _lib.debug_set(key, -1)

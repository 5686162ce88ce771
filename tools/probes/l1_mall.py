#!/usr/bin/env python3
"""Is layer 1's window kernel held back by HBM?  The same launch at batch sizes whose tensors do / do not fit the 256 MiB Infinity
Cache (x + y [+ residual] = 3.6 MB per clip each): if the rate rises as the working set shrinks into the cache, splitting the batch
into cache-sized chunks (conv1 -> conv2 per chunk, the intermediate never leaving the die) would pay; if it does not, layer 1's
0.36 is not a bandwidth figure.  Tile counts are chosen as whole multiples of the 512 persistent workgroups' ranges.
    python tools/probes/l1_mall.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from deeplip_amd import ops, packing
wsp, wsc = packing.split_weights(torch.randn(64, 3, 3, 64, dtype=torch.float64) * 0.05)
wsp, wsc = wsp.cuda(), wsc.cuda()
b = torch.randn(64, device="cuda"); sl = torch.rand(64, device="cuda")
kw = dict(pad=(1, 1), slope=sl, w_scale=wsc, x_split=True, out_split=True)
for B in (8, 16, 32, 64, 128):
    N = B * 29
    x = ops.split_pack(torch.randn(N, 22, 22, 64, device="cuda"))
    y = ops.conv_nhwc(x, wsp, b, **kw); rs = ops.split_pack(torch.randn_like(y))
    for res in (None, rs):
        best = 1e9
        for rnd in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.conv_nhwc(x, wsp, b, residual=res, out=y, **kw)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 100)
        mb = x.numel() * 4 / 1e6
        tiles = (N * 484 + 127) // 128
        print(f"B={B:4d} frames {N:5d} tiles {tiles:6d} ({tiles / 512:5.2f} per workgroup)  x = {mb:6.1f} MB  {'residual' if res is not None else 'plain   '} "
              f"{best:7.1f} us  {2 * N * 484 * 64 * 576 / best / 1e6:5.0f} TF", flush=True)

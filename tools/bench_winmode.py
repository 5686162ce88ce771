#!/usr/bin/env python3
"""Window mode of the 256x128 LDS-DMA tile (stride-1 layers whose taps share rows; LAB build only:
DLIP_LIB_PATH=deeplip_amd/lib/libdeeplip_hip_lab.so, switched on by dlip_debug_set(DLIP_DBG_WIN, 2)) against the plain ring
on the same launches: correctness first (both against each other), then interleaved timing (development tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeplip_amd import _lib, ops, packing

B = int(os.environ.get("B", 64))
N = B * 29
L = [  # name, input NHWC, K, R, S, pad, dil, residual
    ("l3.conv", (N, 6, 6, 256), 256, 3, 3, 1, 1, True),
    ("l4.conv", (N, 3, 3, 512), 512, 3, 3, 1, 1, True),
    ("l3.conv-nores", (N, 6, 6, 256), 256, 3, 3, 1, 1, False),
    ("tdnn.k3d2", (B, 1, 296, 512), 512, 1, 3, 0, 2, False),
    ("tdnn.k3d3", (B, 1, 292, 512), 512, 1, 3, 0, 3, False),
    ("tdnn.k5", (B, 1, 300, 512), 512, 1, 5, 0, 1, False),
    ("small", (7, 5, 9, 64), 96, 3, 3, 2, 2, True),
]
only = sys.argv[1] if len(sys.argv) > 1 else ""
for name, (n, h, w, c), k, r, s, pd, dl, res in L:
    if only and only not in name:
        continue
    torch.manual_seed(1)
    x = ops.split_pack(torch.randn(n, h, w, c, device="cuda"))
    wsp, wsc = packing.split_weights(torch.randn(k, r, s, c, dtype=torch.float64) * 0.03)
    wsp, wsc = wsp.cuda(), wsc.cuda()
    b = torch.randn(k, device="cuda"); sl = torch.rand(k, device="cuda")
    pp = (0, pd) if h == 1 else (pd, pd)
    dd = (1, dl) if h == 1 else (dl, dl)
    kw = dict(pad=pp, dil=dd, slope=sl, w_scale=wsc, x_split=True, out_split=True)
    _lib.debug_set(_lib.DBG_DMA_TILE, 5)
    y0 = ops.conv_nhwc(x, wsp, b, **kw)
    rs = ops.split_pack(torch.randn_like(y0)) if res else None
    outs = {}
    for wmode in (2, 0):
        _lib.debug_set(_lib.DBG_WIN, wmode)
        outs[wmode] = ops.split_unpack(ops.conv_nhwc(x, wsp, b, residual=rs, **kw).clone())
    torch.cuda.synchronize()
    d = float((outs[2] - outs[0]).abs().max() / outs[0].abs().max())
    fl = 2.0 * y0.shape[0] * y0.shape[1] * y0.shape[2] * k * r * s * c
    line = f"{name:14s} max|win-ring|/max = {d:.2e}  "
    y = torch.empty_like(y0)
    best = {2: 1e30, 0: 1e30}
    for rnd in range(3):
        for wmode in (2, 0):
            _lib.debug_set(_lib.DBG_WIN, wmode)
            ops.conv_nhwc(x, wsp, b, residual=rs, out=y, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.conv_nhwc(x, wsp, b, residual=rs, out=y, **kw)
            e1.record(); torch.cuda.synchronize()
            best[wmode] = min(best[wmode], e0.elapsed_time(e1) * 100)
    print(line + f"window {best[2]:7.1f} us {fl / best[2] / 1e6:4.0f} TF   ring {best[0]:7.1f} us {fl / best[0] / 1e6:4.0f} TF", flush=True)
_lib.debug_set(_lib.DBG_WIN, -1); _lib.debug_set(_lib.DBG_DMA_TILE, -1)

"""Back-to-back timing of the speech encoder's convolutions on the rows kernel vs the ring kernel (same box, interleaved).
    python tools/bench_rows.py [B]        (DLIP_LIB_PATH=.../libdeeplip_hip_lab.so + DLIP_STAMP_PRINT=1 for the in-kernel stamps)"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from deeplip_amd import _lib, ops, packing

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
LAYERS = [("tdnn.0 k5 96->512", 300, 96, 512, 5, 1), ("k1 512->512", 296, 512, 512, 1, 1), ("k3 d2 512->512", 296, 512, 512, 3, 2),
          ("k3 d4 512->512", 286, 512, 512, 3, 4), ("tdnn.8 k1", 278, 512, 512, 1, 1), ("tdnn.9 512->1500", 278, 512, 1500, 1, 1)]
POOLED = ("tdnn.9 pooled", 278, 512, 1500)
g = torch.Generator().manual_seed(1)
for name, T, C, K, S, dil in LAYERS:
    x = ops.split_pack((torch.randn(B, T, C, generator=g) * 1.5).cuda())
    w = torch.randn(K, S, C, generator=g) / np.sqrt(C * S)
    ws, sc = packing.split_weights(w.double())
    ws, sc = ws.cuda(), sc.cuda()
    b = (torch.randn(K, generator=g) * 0.1).cuda()
    slope = torch.full((K,), 0.2).cuda()
    osp = K % 32 == 0
    flops = 2.0 * B * (T - dil * (S - 1)) * K * C * S
    res = {}
    for rnd in range(3):
        for mode in (0, 5, 4, 3):
            _lib.debug_set(_lib.DBG_ROWS, mode)
            for _ in range(3):
                ops.conv1d_ntc(x, ws, b, dilation=dil, slope=slope, w_scale=sc, x_split=True, out_split=osp)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 20
            e0.record()
            for _ in range(n):
                ops.conv1d_ntc(x, ws, b, dilation=dil, slope=slope, w_scale=sc, x_split=True, out_split=osp)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(mode, []).append(e0.elapsed_time(e1) / n * 1e3)
    _lib.debug_set(_lib.DBG_ROWS, -1)
    line = "  ".join(f"{'ring' if m == 0 else 'rows' + str(m)} {min(v):7.1f} us {flops / min(v) / 1e6:6.1f} TF" for m, v in res.items())
    print(f"{name:22s} M={B * (T - dil * (S - 1)):6d}  {line}", flush=True)
# tdnn.9 with the pooled epilogue (what the extraction path runs)
name, T, C, K = POOLED
x = ops.split_pack((torch.randn(B, T, C, generator=g) * 1.5).cuda()).view(B, 1, T, C)
w = torch.randn(K, 1, 1, C, generator=g) / np.sqrt(C)
ws, sc = packing.split_weights(w.double())
ws, sc = ws.cuda(), sc.cuda()
b = (torch.randn(K, generator=g) * 0.1).cuda()
slope = torch.full((K,), 0.2).cuda()
res = {}
for rnd in range(3):
    for mode in (0, 5, 4):
        _lib.debug_set(_lib.DBG_ROWS, mode)
        for _ in range(3):
            ops.conv_pool(x, ws, b, sc, T, slope=slope)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.conv_pool(x, ws, b, sc, T, slope=slope)
        e1.record()
        torch.cuda.synchronize()
        res.setdefault(mode, []).append(e0.elapsed_time(e1) / 20 * 1e3)
_lib.debug_set(_lib.DBG_ROWS, -1)
flops = 2.0 * B * T * K * C
print(f"{name:22s} M={B * T:6d}  " + "  ".join(f"{'ring' if m == 0 else 'rows' + str(m)} {min(v):7.1f} us {flops / min(v) / 1e6:6.1f} TF" for m, v in res.items()), flush=True)

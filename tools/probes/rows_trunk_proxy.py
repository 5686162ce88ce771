"""What the rows kernel's form (160 x 256 tile, continuous slice stream, register epilogue) would buy on the trunk's layers 3 / 4:
1-D "valid" nine-tap proxies with the same M / C / K / slices as the 3x3 convolutions (no masks, no residual), ring vs rows,
interleaved on one box.     python tools/probes/rows_trunk_proxy.py"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from deeplip_amd import _lib, ops, packing

# name, rows per "utterance" T' (M = 64 T'), C, K
CASES = [("layer3 real M", 1044, 256, 256), ("layer3 2 full rounds", 1280, 256, 256), ("layer4 real M", 261, 512, 512),
         ("layer4 one full round", 320, 512, 512)]
g = torch.Generator().manual_seed(1)
S = 9
for name, Tp, C, K in CASES:
    B, T = 64, Tp + S - 1
    x = ops.split_pack((torch.randn(B, T, C, generator=g) * 1.5).cuda())
    w = torch.randn(K, S, C, generator=g) / np.sqrt(C * S)
    ws, sc = packing.split_weights(w.double())
    ws, sc = ws.cuda(), sc.cuda()
    b = (torch.randn(K, generator=g) * 0.1).cuda()
    slope = torch.full((K,), 0.2).cuda()
    flops = 2.0 * B * Tp * K * C * S
    res = {}
    for rnd in range(3):
        for mode in (0, 5, 21):          # ring | rows kernel, speech-encoder mode | rows kernel, GENERAL mode (masks + balanced split)
            _lib.debug_set(_lib.DBG_ROWS, 0 if mode == 21 else mode)
            _lib.debug_set(_lib.DBG_ROWS2D, 1 if mode == 21 else 0)
            for _ in range(3):
                ops.conv1d_ntc(x, ws, b, slope=slope, w_scale=sc, x_split=True, out_split=True)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.conv1d_ntc(x, ws, b, slope=slope, w_scale=sc, x_split=True, out_split=True)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(mode, []).append(e0.elapsed_time(e1) / 20 * 1e3)
    _lib.debug_set(_lib.DBG_ROWS, -1)
    _lib.debug_set(_lib.DBG_ROWS2D, -1)
    print(f"{name:24s} M={B * Tp:6d} C={C} K={K}  " + "  ".join(
        f"{'ring' if m == 0 else 'rows-general' if m == 21 else 'rows' + str(m)} {min(v):7.1f} us {flops / min(v) / 1e6:6.1f} TF" for m, v in res.items()), flush=True)

"""Layer-1 / layer-2 window-kernel launches at B = 64 with the deep weight ring (built-in) vs three stages (dlip_debug_set(4, 3)),
interleaved on one box.     python tools/probes/win_ring_depth.py"""
import sys, torch
sys.path.insert(0, ".")
from deeplip_amd import _lib, ops, packing
N = 64 * 29
for name, hw, C in (("layer 1 (64 -> 64, 22x22)", 22, 64), ("layer 2 (128 -> 128, 11x11)", 11, 128)):
    x = ops.split_pack(torch.randn(N, hw, hw, C, device="cuda"))
    wsp, wsc = packing.split_weights(torch.randn(C, 3, 3, C, dtype=torch.float64) * 0.05)
    wsp, wsc = wsp.cuda(), wsc.cuda()
    b = torch.randn(C, device="cuda"); sl = torch.rand(C, device="cuda")
    kw = dict(pad=(1, 1), slope=sl, w_scale=wsc, x_split=True, out_split=True)
    y = ops.conv_nhwc(x, wsp, b, **kw); rs = ops.split_pack(torch.randn_like(y))
    fl = 2.0 * N * hw * hw * C * C * 9
    for res in (None, rs):
        t = {}
        for rnd in range(3):
            for mode in (3, -1):
                _lib.debug_set(_lib.DBG_WIN, mode)
                for _ in range(3): ops.conv_nhwc(x, wsp, b, residual=res, out=y, **kw)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20): ops.conv_nhwc(x, wsp, b, residual=res, out=y, **kw)
                e1.record(); torch.cuda.synchronize()
                t.setdefault(mode, []).append(e0.elapsed_time(e1) / 20 * 1e3)
        _lib.debug_set(_lib.DBG_WIN, -1)
        print(f"{name} {'+ residual' if res is not None else '          '}: 3 stages {min(t[3]):7.1f} us {fl/min(t[3])/1e6:5.0f} TF   deep ring {min(t[-1]):7.1f} us {fl/min(t[-1])/1e6:5.0f} TF", flush=True)

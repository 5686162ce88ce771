"""(needs the LAB library: DLIP_LIB_PATH=deeplip_amd/lib/libdeeplip_hip_lab.so -- the general mode is not in the product build)
The trunk's layers 3 / 4 at the bench's batch on the rows kernel's general mode vs the ring kernel (dlip_debug_set(7, 1 | 0)),
interleaved on one box.     python tools/probes/rows2d_layers.py [B]"""

import sys
import torch
sys.path.insert(0, ".")
from deeplip_amd import _lib, ops, packing

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = B * 29
L = [  # name, input NHWC, K, stride, residual, dual source channels (0 = none)
    ("l3.conv1s2", (N, 11, 11, 128), 256, 2, False, 0),
    ("l3.conv2+down", (N, 6, 6, 256), 256, 1, False, 128),
    ("l3.conv", (N, 6, 6, 256), 256, 1, False, 0),
    ("l3.conv+res", (N, 6, 6, 256), 256, 1, True, 0),
    ("l4.conv1s2", (N, 6, 6, 256), 512, 2, False, 0),
    ("l4.conv2+down", (N, 3, 3, 512), 512, 1, False, 256),
    ("l4.conv", (N, 3, 3, 512), 512, 1, False, 0),
    ("l4.conv+res", (N, 3, 3, 512), 512, 1, True, 0),
]
tot = {0: 0.0, 1: 0.0}
for name, (n, h, w, c), k, st, res, c2 in L:
    x = ops.split_pack(torch.randn(n, h, w, c, device="cuda"))
    ho = (h + 2 - 3) // st + 1
    rows = torch.randn(k, 9 * c + c2, dtype=torch.float64) * 0.03
    wsp, wsc = packing.split_weights(rows if c2 else rows.view(k, 3, 3, c))
    wsp, wsc = wsp.cuda(), wsc.cuda()
    b = torch.randn(k, device="cuda")
    sl = torch.rand(k, device="cuda")
    r = ops.split_pack(torch.randn(n, ho, ho, k, device="cuda")) if res else None
    x2 = ops.split_pack(torch.randn(n, 2 * h - 1, 2 * h - 1, c2, device="cuda")) if c2 else None

    def run():
        if c2:
            return ops.conv2_nhwc(x, x2, wsp, b, wsc, pad=(1, 1), stride2=(2, 2), slope=sl, out_split=True)
        return ops.conv_nhwc(x, wsp, b, stride=(st, st), pad=(1, 1), slope=sl, w_scale=wsc, residual=r, x_split=True, out_split=True)

    flops = 2.0 * n * ho * ho * k * (9 * c + c2)
    t = {}
    for rnd in range(3):
        for mode in (0, 1):
            _lib.debug_set(_lib.DBG_ROWS2D, mode)
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run()
            e1.record()
            torch.cuda.synchronize()
            t.setdefault(mode, []).append(e0.elapsed_time(e1) / 20 * 1e3)
    _lib.debug_set(_lib.DBG_ROWS2D, -1)
    for m in (0, 1):
        tot[m] += min(t[m])
    print(f"{name:16s} M={n * ho * ho:6d}  ring {min(t[0]):7.1f} us {flops / min(t[0]) / 1e6:6.1f} TF   rows {min(t[1]):7.1f} us {flops / min(t[1]) / 1e6:6.1f} TF   "
          f"{(min(t[0]) / min(t[1]) - 1) * 100:+5.1f} %", flush=True)
print(f"sum ring {tot[0]:.1f} us   rows {tot[1]:.1f} us")

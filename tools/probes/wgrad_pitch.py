#!/usr/bin/env python3
"""A/B of the weight-gradient path on layer-1's shape: even vs odd row pitch of the reduction-major operands."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deeplip_amd import autograd_video as av
for shape, K in (((32 * 29, 22, 22, 64), 64), ((32 * 29, 11, 11, 128), 128), ((32 * 29, 6, 6, 256), 256)):
    x = torch.randn(shape, device="cuda"); dy = torch.randn(shape[:3] + (K,), device="cuda") * 1e-4
    for odd in (False, True, False, True):
        av.WGRAD_ODD_PITCH = odd
        av.wgrad_conv_fused(x, dy, 3, 3, (1, 1), (1, 1), (1, 1))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): av.wgrad_conv_fused(x, dy, 3, 3, (1, 1), (1, 1), (1, 1))
        e1.record(); torch.cuda.synchronize()
        print(shape, K, "odd pitch" if odd else "even pitch", f"{e0.elapsed_time(e1) / 5:.3f} ms")

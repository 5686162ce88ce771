#!/usr/bin/env python3
"""Weight gradients of the trunk's convolutions at training shapes (B = 32 clips x 29 frames): as a convolution over one
transposed copy of x and dy (autograd_video.wgrad_as_conv) vs the reduction-major GEMM (wgrad_conv_fused); optional forced tile."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeplip_amd import _lib, autograd_video as av
ap = argparse.ArgumentParser()
ap.add_argument("--tile", type=int, default=-1)
ap.add_argument("--gemm", action="store_true")
a = ap.parse_args()
N = 32 * 29
shapes = [("layer1 3x3", 22, 64, 64, 3, 1, 1), ("layer2.0 3x3 s2", 22, 64, 128, 3, 2, 1), ("layer2 3x3", 11, 128, 128, 3, 1, 1),
          ("layer3.0 3x3 s2", 11, 128, 256, 3, 2, 1), ("layer3 3x3", 6, 256, 256, 3, 1, 1), ("layer4.0 3x3 s2", 6, 256, 512, 3, 2, 1),
          ("layer4 3x3", 3, 512, 512, 3, 1, 1), ("layer2.0 1x1 s2", 22, 64, 128, 1, 2, 0)]
if a.tile >= 0:
    _lib.debug_set(_lib.DBG_DMA_TILE, a.tile)
for name, H, C, K, R, s, p in shapes:
    Ho = (H + 2 * p - (R - 1) - 1) // s + 1
    x = torch.randn(N, H, H, C, device="cuda"); dy = torch.randn(N, Ho, Ho, K, device="cuda") * 1e-3
    fn = av.wgrad_conv_fused if a.gemm else av.wgrad_as_conv
    lift = av.pow2_lift(dy)
    for _ in range(2):
        fn(x, dy, R, R, (s, s), (p, p), (1, 1), scale2=lift)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn(x, dy, R, R, (s, s), (p, p), (1, 1), scale2=lift)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 200
    fl = 2.0 * N * Ho * Ho * K * C * R * R
    print(f"{name:18s} {us:8.1f} us  {fl / us / 1e6:6.1f} TF/s", flush=True)

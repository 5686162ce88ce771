#!/usr/bin/env python3
"""Timing of the fused stem + max-pool kernel (dlip_stem3d_pool_f16x3) at the bench's shape (development tool).
With the lab build (DLIP_LIB_PATH=.../libdeeplip_hip_lab.so) and DLIP_STAMP_PRINT=1 the first launch prints its
in-kernel phase stamps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeplip_amd import ops, packing

B, T, H, W = int(os.environ.get("B", 64)), 29, 88, 88
torch.manual_seed(0)
x = torch.randn(B, T, H, W, device="cuda")
w = torch.randn(64, 1, 5, 7, 7, dtype=torch.float64) * 0.05
img, sc = packing.split_stem_weights(w)
img, sc = img.cuda(), sc.cuda()
b = torch.randn(64, device="cuda"); sl = torch.rand(64, device="cuda")
y = ops.stem3d_pool(x, img, b, sl, sc)
torch.cuda.synchronize()
os.environ.pop("DLIP_STAMP_PRINT", None)
fl = 2.0 * B * T * 44 * 44 * 64 * 245
for rnd in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.stem3d_pool(x, img, b, sl, sc)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print(f"stem+pool B={B}: {us:7.1f} us  {fl / us / 1e6:5.0f} TF   checksum {float(ops.split_unpack(y).double().sum()):.6f}", flush=True)

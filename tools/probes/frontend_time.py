"""The three power-spectrum routes of AudioFrontend timed: 256 utterances x 3 s, MFCC-24/26.   python tools/probes/frontend_time.py"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from deeplip_amd.frontend import AudioFrontend

x = torch.randn(256, 48000, device="cuda") * 0.1
for d in ("fft64", "gemm32", "direct64"):
    fe = AudioFrontend("mfcc", dft=d)
    for _ in range(3):
        fe(x)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        fe(x)
    b.record()
    torch.cuda.synchronize()
    print(f"{d}: {a.elapsed_time(b) / 10:.3f} ms per batch of 256 x 3 s ({256 * 299} frames)")

"""C3 (speech encoder, B = 256 x 80 x 300) eager, for a kernel trace:
   rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c3 -- python3 tools/probes/c3_trace.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from deeplip_amd import packing
packing.set_precision("f16x3")
video, audio, _ = bench.build_models(torch.device("cuda"), 80)
x = torch.randn(256, 1, 80, 300, device="cuda")
with torch.no_grad():
    for _ in range(12):
        audio.extract_embedding(x)
torch.cuda.synchronize()

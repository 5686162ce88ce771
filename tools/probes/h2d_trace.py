#!/usr/bin/env python3
"""One short pipelined run for a kernel + memory-copy trace: does the H2D copy run as a blit kernel or on SDMA?
   rocprofv3 --kernel-trace --memory-copy-trace --stats -d gpurun_out/h2dtrace -- python3 tools/probes/h2d_trace.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import bench
from deeplip_amd import packing, weightgen as wg
from deeplip_amd.pipeline import ExtractPipeline
from deeplip_amd.synthetic import frames_u8_from_clips
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
packing.set_precision("f16x3")
video, audio, _ = bench.build_models(dev, 80)
B = 64
clips = wg.video_input(B, key="h2d.v")
xa_p = torch.from_numpy(wg.audio_input(B, 80, 300, key="h2d.a")).unsqueeze(1).pin_memory()
rgb_h = torch.from_numpy(frames_u8_from_clips(clips, True)).pin_memory()
step = lambda v, m: bench.local_step(video, audio, v, m)
pipe = ExtractPipeline(step, rgb_h.to(dev), xa_p.to(dev))
table = torch.empty((20 * B, 1024), device=dev)
pipe.run([(rgb_h, xa_p)] * 4, table); pipe.finish()
t0 = time.perf_counter(); pipe.run([(rgb_h, xa_p)] * 20, table); pipe.finish()
print("ms/step", 1e3 * (time.perf_counter() - t0) / 20)
pipe.close()

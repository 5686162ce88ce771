#!/usr/bin/env python3
"""Un-profiled timeline of the double-buffered extraction: HIP events around every batch's copies (copy stream) and replay
(run stream), printed relative to the first replay.  python tools/probes/h2d_timeline.py [gray|rgb|float] [nbatches]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import bench
from deeplip_amd import packing, weightgen as wg
from deeplip_amd.pipeline import ExtractPipeline
from deeplip_amd.synthetic import frames_u8_from_clips
kind = sys.argv[1] if len(sys.argv) > 1 else "rgb"; nb = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
packing.set_precision("f16x3")
video, audio, _ = bench.build_models(dev, 80)
B = 64
clips = wg.video_input(B, key="h2d.v")
xa_p = torch.from_numpy(wg.audio_input(B, 80, 300, key="h2d.a")).unsqueeze(1).pin_memory()
hx = {"rgb": lambda: torch.from_numpy(frames_u8_from_clips(clips, True)), "gray": lambda: torch.from_numpy(frames_u8_from_clips(clips, False)),
      "float": lambda: torch.from_numpy(clips)}[kind]().pin_memory()
step = lambda v, m: bench.local_step(video, audio, v, m)
pipe = ExtractPipeline(step, hx.to(dev), xa_p.to(dev))
table = torch.empty((nb * B, 1024), device=dev)
pipe.run([(hx, xa_p)] * 4, table); pipe.finish()
E = lambda: torch.cuda.Event(enable_timing=True)
evs = []
t0 = time.perf_counter()
for i in range(nb):
    k = i % 2
    if os.environ.get("THROTTLE") and i >= 2: pipe.free[k].synchronize()      # host never more than `depth` batches ahead
    c0, c1, r0, r1 = E(), E(), E(), E()
    with torch.cuda.stream(pipe.copy_stream):
        if i >= 2: pipe.copy_stream.wait_event(pipe.free[k])
        c0.record()
        for dst, src in zip(pipe.sets[k], (hx, xa_p)): dst.copy_(src, non_blocking=True)
        c1.record(); pipe.ready[k].record(pipe.copy_stream)
    with torch.cuda.stream(pipe.run_stream):
        pipe.run_stream.wait_event(pipe.ready[k]); r0.record()
        out = pipe.plans[k].run(); table[i * B:(i + 1) * B].copy_(out, non_blocking=True)
        r1.record(); pipe.free[k].record(pipe.run_stream)
    evs.append((c0, c1, r0, r1))
host_ms = 1e3 * (time.perf_counter() - t0)
pipe.finish()
base = evs[0][2]
print(f"{kind}: host enqueue of {nb} batches {host_ms:.2f} ms")
tot = evs[4][2].elapsed_time(evs[-1][3]) / (nb - 4)
print(f"steady ms/batch (batches 4..{nb - 1}): {tot:.3f}")
for i, (c0, c1, r0, r1) in enumerate(evs):
    if i % 4 == 0: print(f"batch {i:2d}  copy {base.elapsed_time(c0):8.3f} -> {base.elapsed_time(c1):8.3f} ({c0.elapsed_time(c1):6.3f})   replay {base.elapsed_time(r0):8.3f} -> {base.elapsed_time(r1):8.3f} ({r0.elapsed_time(r1):6.3f})")
pipe.close()

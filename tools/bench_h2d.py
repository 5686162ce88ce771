#!/usr/bin/env python3
"""Where the H2D-inclusive step loses against the resident one: plan replay on resident float clips / resident uint8 RGB /
resident uint8 gray (isolates the pre-pass), then the double-buffered pipeline with the copies (isolates the copies), and the
copies alone.  One box, one process.   python tools/bench_h2d.py [--steps 40]"""
import argparse, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from deeplip_amd import packing, weightgen as wg
from deeplip_amd.plan import StepPlan
from deeplip_amd.pipeline import ExtractPipeline
from deeplip_amd.synthetic import frames_u8_from_clips

ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=40); ap.add_argument("--batch", type=int, default=64)
a = ap.parse_args()
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
packing.set_precision("f16x3")
video, audio, _ = bench.build_models(dev, 80)
B = a.batch
clips = wg.video_input(B, key="h2d.v"); xv = torch.from_numpy(clips).to(dev)
xa_h = torch.from_numpy(wg.audio_input(B, 80, 300, key="h2d.a")).unsqueeze(1); xa = xa_h.to(dev)
rgb_h = torch.from_numpy(frames_u8_from_clips(clips, True)).pin_memory(); gray_h = torch.from_numpy(frames_u8_from_clips(clips, False)).pin_memory()
xa_p = xa_h.pin_memory(); xv_p = torch.from_numpy(clips).pin_memory()
step = lambda v, m: bench.local_step(video, audio, v, m)
sync = torch.cuda.synchronize
def replay(x):
    p = StepPlan(step, x, xa); ms = bench._timed_replay(p.run, a.steps, 5, sync); p.close(); return ms
def piped(hx):
    pipe = ExtractPipeline(step, hx.to(dev), xa)
    table = torch.empty((a.steps * B, 1024), device=dev)
    pipe.run([(hx, xa_p)] * 6, table); pipe.finish(); sync()
    t0 = time.perf_counter(); pipe.run([(hx, xa_p)] * a.steps, table); pipe.finish(); dt = time.perf_counter() - t0
    pipe.close(); return 1e3 * dt / a.steps
def copies(hx):
    d = hx.to(dev); da = xa.clone(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): d.copy_(hx, non_blocking=True); da.copy_(xa_p, non_blocking=True)
        s.synchronize(); t0 = time.perf_counter()
        for _ in range(a.steps): d.copy_(hx, non_blocking=True); da.copy_(xa_p, non_blocking=True)
        s.synchronize()
    return 1e3 * (time.perf_counter() - t0) / a.steps
r = {}
for rnd in range(2):
    r[f"replay_float_resident_ms.{rnd}"] = replay(xv)
    r[f"replay_u8rgb_resident_ms.{rnd}"] = replay(rgb_h.to(dev))
    r[f"replay_u8gray_resident_ms.{rnd}"] = replay(gray_h.to(dev))
    r[f"pipeline_u8rgb_ms.{rnd}"] = piped(rgb_h)
    r[f"pipeline_u8gray_ms.{rnd}"] = piped(gray_h)
    r[f"pipeline_float_ms.{rnd}"] = piped(xv_p)
r["copy_only_u8rgb_ms"] = copies(rgb_h); r["copy_only_float_ms"] = copies(xv_p)
r["mb_rgb"] = rgb_h.numel() / 1e6; r["mb_float"] = xv_p.numel() * 4 / 1e6
for k, v in r.items(): print(f"{k:34s} {v:9.4f}")

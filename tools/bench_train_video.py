#!/usr/bin/env python3
"""Step time of full lip-clip model training at the reference's shapes: Lipreading
(ResNet-18 + MS-TCN), 29-frame 88x88 clips, Adam, CrossEntropy; forward + backward + optimiser, HIP events.
Also prints the 5 costliest kernels of one step (torch profiler is not used: HIP events around phases)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from deeplip_amd import autograd as ag, weightgen as wg
from models.video_models.model import Lipreading

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--frames", type=int, default=29)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--eager", action="store_true", help="issue every launch of every step from Python (no recorded step graph)")
ap.add_argument("--dbg", action="append", default=[], metavar="KEY=VALUE", help="dlip_debug_set(KEY, VALUE) before anything is launched (A/B runs)")
a = ap.parse_args()
for _kv in a.dbg:
    from deeplip_amd import _lib as _dl
    _dl.debug_set(int(_kv.split("=")[0]), int(_kv.split("=")[1]))
tcn = {"num_layers": 4, "kernel_size": [3, 5, 7], "dropout": 0.2, "dwpw": False, "width_mult": 1}
net = Lipreading(num_classes=54, relu_type="prelu", tcn_options=tcn, extract_feats=False)
sd = wg.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, prefix="video.")
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
net.cuda().train()
opt = torch.optim.Adam(net.parameters(), lr=torch.tensor(3e-4, device="cuda"), weight_decay=1e-4, capturable=True, fused=True)
x = torch.from_numpy(wg.video_input(a.batch, frames=a.frames, key="bench.vtrain")).cuda()
lab = torch.from_numpy(wg.labels(a.batch, 54)).cuda()
lengths = torch.full((a.batch,), a.frames, dtype=torch.int32, device="cuda")
fwd_gflop = (18.337 + 2.230) * a.batch * a.frames / 29.0


def step():
    opt.zero_grad()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    e[0].record()
    loss = ag.margin_ce_loss(net(x, lengths=lengths), lab)
    e[1].record()
    loss.backward()
    e[2].record()
    opt.step()
    e[3].record()
    return loss, e


step()
tot = [0.0, 0.0, 0.0]
for _ in range(a.steps):
    loss, e = step()
    torch.cuda.synchronize()
    for i in range(3):
        tot[i] += e[i].elapsed_time(e[i + 1]) / a.steps
ms = sum(tot)
print(f"full Lipreading training step, launches issued from Python: batch {a.batch} x {a.frames} frames: {ms:.1f} ms/step = {a.batch / ms * 1e3:.1f} clips/s "
      f"(forward {tot[0]:.1f} ms, backward {tot[1]:.1f} ms, Adam {tot[2]:.1f} ms; ~{3 * fwd_gflop / ms:.1f} TFLOP/s at 3x forward FLOPs); "
      f"loss {float(loss.detach()):.4f}; peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
if not a.eager:
    import gc
    import time
    from deeplip_amd.train_plan import TrainStepGraph
    # the eager phase ran on the default stream: nothing of its autograd graph may stay alive (AccumulateGrad nodes remember their
    # stream, and a node of the default stream inside a capture on the plan's stream breaks the capture)
    del loss, e
    opt.zero_grad(set_to_none=True)
    gc.collect()
    torch.cuda.synchronize()

    def one(xb, lb, ln):
        opt.zero_grad(set_to_none=True)
        l = ag.margin_ce_loss(net(xb, lengths=ln), lb)
        l.backward()
        opt.step()
        return l

    plan = TrainStepGraph(one, eager_steps=1)
    for _ in range(3):                       # 1 eager + recording + 1 replay
        l = plan.step(x, lab, lengths)
    plan.finish()
    n = max(a.steps, 10)
    t0 = time.perf_counter()
    for _ in range(n):
        l = plan.step(x, lab, lengths)
    plan.finish()
    ms = (time.perf_counter() - t0) / n * 1e3
    print(f"the same step recorded once and replayed as one HIP graph (deeplip_amd.train_plan): {ms:.1f} ms/step = {a.batch / ms * 1e3:.1f} clips/s "
          f"(~{3 * fwd_gflop / ms:.1f} TFLOP/s at 3x forward FLOPs); loss {float(l.detach()):.4f}")
from deeplip_amd import autograd_video as _av
print("weight images:", len(_av.WEIGHT_PREP.order), "registered;", _av.WEIGHT_PREP.stats)

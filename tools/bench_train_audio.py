#!/usr/bin/env python3
"""Step time of full-encoder speech training at the reference's shapes: E-TDNN, 300 frames,
batch 64 / 256 (conf/audio_config.yaml bs), LMCL; forward + backward + SGD, HIP events."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from deeplip_amd import weightgen as wg
from models.audio_models.loss import LMCL
from models.audio_models.tdnn import SpeakerEmbNet

ETDNN_CONTEXT = [[-2, -1, 0, 1, 2], [0], [-2, 0, 2], [0], [-3, 0, 3], [0], [-4, 0, 4], [0], [0], [0]]   # conf/audio_config.yaml
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--dim", type=int, default=24)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--dbg", action="append", default=[], metavar="KEY=VALUE", help="dlip_debug_set(KEY, VALUE) before anything is launched (A/B runs)")
a = ap.parse_args()
for _kv in a.dbg:
    from deeplip_amd import _lib as _dl
    _dl.debug_set(int(_kv.split("=")[0]), int(_kv.split("=")[1]))
et = {"input_dim": a.dim, "hidden_dim": [512] * 9 + [1500], "context": ETDNN_CONTEXT, "tdnn_layers": 10, "embedding_dim": 512,
      "pooling": "statistic", "attention_hidden_size": 64, "bn_first": True}
net = SpeakerEmbNet({"arch": "etdnn", "etdnn": et})
sd = wg.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, prefix="audio.")
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
net.cuda().train()
crit = LMCL(512, 57, 30, 0.2).cuda()
opt = torch.optim.SGD([{"params": net.parameters()}, {"params": crit.parameters()}], 0.01, momentum=0.9, weight_decay=1e-5)
x = torch.from_numpy(wg.audio_input(a.batch, a.dim, 300, key="bench.atrain")).cuda()
lab = torch.from_numpy(wg.labels(a.batch, 57)).cuda()


def step():
    opt.zero_grad()
    loss, _ = crit(net(x), lab)
    loss.backward()
    opt.step()
    return loss


for _ in range(2):
    step()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.steps):
    loss = step()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.steps
fwd_gflop = 2.563 * a.batch
print(f"full-encoder E-TDNN training step: batch {a.batch} x 300 frames x {a.dim} dims: {ms:.2f} ms/step = {a.batch / ms * 1e3:.0f} utt/s "
      f"(~{3 * fwd_gflop / ms:.1f} TFLOP/s at 3x forward FLOPs); loss {float(loss.detach()):.4f}")
if os.environ.get("DLIP_AUDIO_GRAPH", "1") != "0":
    import gc
    import time
    from deeplip_amd.train_plan import TrainStepGraph
    del loss
    opt.zero_grad(set_to_none=True)
    gc.collect()
    torch.cuda.synchronize()

    def one(xb, lb):
        opt.zero_grad(set_to_none=True)
        l, _ = crit(net(xb), lb)
        l.backward()
        opt.step()
        return l

    plan = TrainStepGraph(one, eager_steps=1)
    for _ in range(3):
        l = plan.step(x, lab)
    plan.finish()
    n = max(a.steps, 10)
    t0 = time.perf_counter()
    for _ in range(n):
        l = plan.step(x, lab)
    plan.finish()
    ms = (time.perf_counter() - t0) / n * 1e3
    print(f"the same step recorded once and replayed as one HIP graph: {ms:.2f} ms/step = {a.batch / ms * 1e3:.0f} utt/s; loss {float(l.detach()):.4f}")

#!/usr/bin/env python3
"""C3 (speech encoder, [256,1,80,300] -> [256,512]) as ONE pass over 256 utterances vs the same batch in chunks of 128 / 64 inside one
recorded plan: a layer's activations are 155 MB at B = 256 (beyond the 256 MiB Infinity Cache with input + output), 39 MB at 64.
    python3 tools/probes/c3_chunked.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from deeplip_amd import _lib, packing, weightgen as wg
from deeplip_amd.plan import StepPlan
from models.audio_models.tdnn import SpeakerEmbNet

packing.set_precision("f16x3")
ctx = [[-2, -1, 0, 1, 2], [0], [-2, 0, 2], [0], [-3, 0, 3], [0], [-4, 0, 4], [0], [0], [0]]
et = {"input_dim": 80, "hidden_dim": [512] * 9 + [1500], "context": ctx, "tdnn_layers": 10, "embedding_dim": 512,
      "pooling": "statistic", "attention_hidden_size": 64, "bn_first": True}
net = SpeakerEmbNet({"arch": "etdnn", "etdnn": et})
sd = wg.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, prefix="audio.")
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
net.eval().cuda()
x = torch.from_numpy(wg.audio_input(256, 80, 300, key="bench.c3")).cuda()
flop = 256 * (2.563e9 + 0.085e9)


def make(chunk):
    def fn(xx):
        with torch.no_grad():
            return torch.cat([net.extract_embedding(xx[i:i + chunk])[0] for i in range(0, xx.shape[0], chunk)], 0)
    return fn


plans = {c: StepPlan(make(c), x) for c in (256, 128, 64)}
ref = None
for rnd in range(3):
    for c, p in plans.items():
        for _ in range(5):
            out = p.run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            out = p.run()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 30 * 1e3
        if ref is None:
            ref = out.clone()
        same = bool(torch.equal(out, ref))
        print(f"chunk {c:3d}: {ms:.4f} ms / 256 utterances = {256e3 / ms:9.0f} utt/s, {flop / ms / 1e9:6.1f} TFLOP/s = {flop / ms / 1e9 / 833.3:.4f}; same bits as one pass: {same}", flush=True)

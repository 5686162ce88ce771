"""Long-run memory check: device memory (torch allocator + hipMemGetInfo, which also sees what the library allocates behind torch's back) and host
RSS over (a) 40 passes of train_fusion.Trainer('av_test')'s ragged extraction over one list, (b) 300 recorded training steps of train_audio over its crop
ladder, (c) 200 eager out-of-range batches under arith auto (repair path).
   python tools/probes/leak_check.py"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import resource
import tempfile

import numpy as np
import torch


def mem():
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    return dict(alloc=torch.cuda.memory_allocated() >> 20, reserved=torch.cuda.memory_reserved() >> 20, device_used=(total - free) >> 20,
                rss=resource.getrusage(resource.RUSAGE_SELF).ru_maxrss >> 10)


def report(tag, a, b):
    grow = {k: b[k] - a[k] for k in a}
    print(f"{tag}: first {a} last {b} growth {grow} MiB", flush=True)
    return max(grow["alloc"], grow["device_used"]) > 256 or grow["rss"] > 512


os.chdir(tempfile.mkdtemp())
bad = False
import train_fusion
from deeplip_amd import arith, weightgen as wg

tr = train_fusion.Trainer("av_test", overrides={"data.test_speakers": 16, "data.test_utt_per_spk": 8, "data.trials": 2000, "data.trial_targets": 400,
                                                "test.batch": 32, "test.write_store": False})
m0 = None
for i in range(40):
    tr._extract(tr.lomgridtestset)
    if i == 3:
        m0 = mem()
bad |= report("(a) 40 ragged extraction passes", m0, mem())
tr.close()
del tr
torch.cuda.empty_cache()

import train_audio
ta = train_audio.Trainer(overrides={"data.n_spk": 12, "data.utt_per_spk": 8, "train.bs": 32, "train.crop_frames": [120, 200], "data.audio_frames": 200,
                                    "train.steps_per_epoch": 25})
m0 = None
for ep in range(12):
    ta.current_epoch = ep
    ta._train_epoch()
    if ep == 1:
        m0 = mem()
bad |= report("(b) 12 recorded training epochs (300 steps over the crop ladder)", m0, mem())
ta.close()
del ta
torch.cuda.empty_cache()

from models.audio_models.tdnn import SpeakerEmbNet
from oracle import deeplip_oracle as O
opts = {"arch": "tdnn", "tdnn": {"input_dim": 24, "hidden_dim": [512] * 4 + [1500], "context": O.TDNN_CONTEXT, "tdnn_layers": 5,
                                  "embedding_dim": 512, "pooling": "statistic", "attention_hidden_size": 64, "bn_first": True}}
net = SpeakerEmbNet(opts)
sd = wg.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, prefix="audio_tdnn.")
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
net.eval().cuda()
arith.configure("auto")
arith.CALIBRATE = False
x = torch.from_numpy(wg.audio_input(8, 24, 200)).cuda() * 1e-4
import logging
logging.getLogger("deeplip_amd.arith").setLevel(logging.ERROR)
m0 = None
for i in range(200):
    net.extract_embedding(x)
    if i == 5:
        m0 = mem()
bad |= report("(c) 200 repaired batches", m0, mem())
print("LEAK" if bad else "ok")
sys.exit(1 if bad else 0)

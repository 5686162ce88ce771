"""Second shape fuzz: the shipped E-TDNN (input_dim 24 and 80) at batch sizes up to 96 (the rows kernel's shape class), uint8 frames of random
source size (88 .. 112, RGB or gray) with random ragged lengths, f16x3 against the engine's exact mode.
   python tools/probes/shape_fuzz2.py [n_shapes] [seed]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import numpy as np
import torch

from deeplip_amd import _lib, arith, weightgen as wg
from models.audio_models.tdnn import SpeakerEmbNet
from models.video_models.model import Lipreading
from oracle import deeplip_oracle as O

n_shapes = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
r = np.random.Generator(np.random.PCG64(seed))


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


nets = {}
for dim in (24, 80):
    o = {"arch": "etdnn", "etdnn": {"input_dim": dim, "hidden_dim": [512] * 9 + [1500], "context": O.ETDNN_CONTEXT, "tdnn_layers": 10,
                                     "embedding_dim": 512, "pooling": "statistic", "attention_hidden_size": 64, "bn_first": True}}
    n = SpeakerEmbNet(o)
    sd = wg.fill_state_dict({k: tuple(v.shape) for k, v in n.state_dict().items()}, prefix=f"audio{dim}.")
    n.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    nets[dim] = n.eval().cuda()
tcn = {"num_layers": 4, "kernel_size": [3, 5, 7], "dropout": 0.2, "dwpw": False, "width_mult": 1}
vnet = Lipreading(num_classes=54, relu_type="prelu", tcn_options=tcn, extract_feats=True)
vsd = wg.fill_state_dict({k: tuple(v.shape) for k, v in vnet.state_dict().items()}, prefix="video.")
vnet.load_state_dict({k: torch.from_numpy(v) for k, v in vsd.items()})
vnet.eval().cuda()

worst = (0.0, None)
for i in range(n_shapes):
    dim = 24 if i % 3 else 80
    net = nets[dim]
    amin = net.frames_consumed() + 2
    B, T = int(r.integers(1, 97)), int(r.integers(amin, 420))
    x = torch.from_numpy(wg.audio_input(B, dim, T, key=f"fz2.a{seed}.{i}")).cuda()
    L = None
    if i % 2:
        L = torch.from_numpy(r.integers(amin, T + 1, size=B).astype(np.int32))
        L[int(r.integers(0, B))] = T
        L = L.cuda()
    out = {}
    for mode in ("f32", "f16x3"):
        arith.configure(mode)
        out[mode] = net.extract_embedding(x, lengths=L)[0].cpu().numpy()
        _lib.check_range(sync=True)
    e = rel(out["f16x3"], out["f32"])
    tag = f"etdnn F={dim} B={B} T={T} ragged={L is not None}"
    print(f"{tag}: {e:.3e}", flush=True)
    if e > worst[0]:
        worst = (e, tag)
    # uint8 frames
    B, T = int(r.integers(1, 9)), int(r.integers(1, 36))
    Hs, Ws = int(r.integers(88, 113)), int(r.integers(88, 113))
    rgb = bool(i % 2)
    shape = (B, T, 3, Hs, Ws) if rgb else (B, T, Hs, Ws)
    fr = torch.from_numpy(r.integers(0, 256, size=shape, dtype=np.uint8)).cuda()
    L = None
    if i % 3 == 0:
        L = [int(v) for v in r.integers(1, T + 1, size=B)]
        L[int(r.integers(0, B))] = T
    for mode in ("f32", "f16x3"):
        arith.configure(mode)
        out[mode] = vnet.embed(fr, L).cpu().numpy()
        _lib.check_range(sync=True)
    e = rel(out["f16x3"], out["f32"])
    tag = f"u8 frames {tuple(shape)} ragged={L is not None}"
    print(f"{tag}: {e:.3e}", flush=True)
    if e > worst[0]:
        worst = (e, tag)
print(f"worst {worst[0]:.3e} at {worst[1]}")
sys.exit(1 if worst[0] > 5e-6 else 0)

"""Shape fuzz of the two encoders' eval paths: random batch sizes and lengths (rectangular and ragged), f16x3 against the engine's exact
f32 mode (which the fixed-shape tests pin to the oracle), plus a few shapes against the oracle itself.  Prints the worst case; exits 1 on
a row outside the bar.   python tools/probes/shape_fuzz.py [n_shapes] [seed]"""
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

n_shapes = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
r = np.random.Generator(np.random.PCG64(seed))


def bar(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return float(np.max(np.abs(got - want) - 1e-4 * np.abs(want)) / max(np.max(np.abs(want)), 1e-30))      # <= 1e-6 passes


opts = {"arch": "etdnn", "etdnn": None}
tdnn_opts = {"arch": "tdnn", "tdnn": {"input_dim": 24, "hidden_dim": [512] * 4 + [1500], "context": O.TDNN_CONTEXT, "tdnn_layers": 5,
                                        "embedding_dim": 512, "pooling": "statistic", "attention_hidden_size": 64, "bn_first": True}}
anet = SpeakerEmbNet(tdnn_opts)
asd = wg.fill_state_dict({k: tuple(v.shape) for k, v in anet.state_dict().items()}, prefix="audio_tdnn.")
anet.load_state_dict({k: torch.from_numpy(v) for k, v in asd.items()})
anet.eval().cuda()
tcn = {"num_layers": 4, "kernel_size": [3, 5, 7], "dropout": 0.2, "dwpw": False, "width_mult": 1}
vnet = Lipreading(num_classes=54, relu_type="prelu", tcn_options=tcn, extract_feats=True)
vsd = wg.fill_state_dict({k: tuple(v.shape) for k, v in vnet.state_dict().items()}, prefix="video.")
vnet.load_state_dict({k: torch.from_numpy(v) for k, v in vsd.items()})
vnet.eval().cuda()
amin = anet.frames_consumed() + 2 if hasattr(anet, "frames_consumed") else 24

worst = (0.0, None)
bad = 0
for i in range(n_shapes):
    # ---- speech encoder: [B, 24, T], half of the shapes ragged
    B, T = int(r.integers(1, 21)), int(r.integers(amin, 520))
    x = torch.from_numpy(wg.audio_input(B, 24, T, key=f"fuzz.a{seed}.{i}"))
    L = None
    if i % 2:
        L = torch.from_numpy(r.integers(amin, T + 1, size=B).astype(np.int32))
        L[int(r.integers(0, B))] = T
    out = {}
    for mode in ("f32", "f16x3"):
        arith.configure(mode)
        out[mode] = (anet.extract_embedding(x.cuda(), lengths=L.cuda() if L is not None else None)[0]).cpu().numpy()
        _lib.check_range(sync=True)
    e = bar(out["f16x3"], out["f32"])
    tag = f"audio B={B} T={T} ragged={L is not None}"
    if i < 6:
        with torch.no_grad():
            rows = [O.speaker_extract_embedding(O.to_torch_sd(asd), x[j:j + 1, :, :int(L[j]) if L is not None else T], O.TDNN_CONTEXT)[0] for j in range(B)]
        eo = bar(out["f32"], torch.cat(rows).numpy())
        tag += f" (exact vs oracle {eo:.2e})"
        e = max(e, eo)
    print(f"{tag}: {e:.3e}")
    if e > worst[0]:
        worst = (e, tag)
    bad += e > 1e-6
    # ---- lip-clip encoder: [B, 1, T, 88, 88], half ragged
    B, T = int(r.integers(1, 7)), int(r.integers(1, 41))
    x = torch.from_numpy(wg.video_input(B, frames=T, key=f"fuzz.v{seed}.{i}"))
    L = None
    if i % 2:
        L = [int(v) for v in r.integers(1, T + 1, size=B)]
        L[int(r.integers(0, B))] = T
    out = {}
    for mode in ("f32", "f16x3"):
        arith.configure(mode)
        out[mode] = vnet.embed(x.cuda(), L).cpu().numpy()
        out[mode + "_feat"] = vnet(x.cuda(), L).cpu().numpy()
        _lib.check_range(sync=True)
    e = max(bar(out["f16x3"], out["f32"]), bar(out["f16x3_feat"], out["f32_feat"]))
    tag = f"video B={B} T={T} ragged={L is not None}"
    if i < 4:
        with torch.no_grad():
            rows = [O.video_time_mean(O.lipreading_features(O.to_torch_sd(vsd), x[j:j + 1, :, :(L[j] if L is not None else T)])) for j in range(B)]
        eo = bar(out["f32"], torch.cat(rows).numpy())
        tag += f" (exact vs oracle {eo:.2e})"
        e = max(e, eo)
    print(f"{tag}: {e:.3e}")
    if e > worst[0]:
        worst = (e, tag)
    bad += e > 1e-6
print(f"worst {worst[0]:.3e} at {worst[1]}; {bad} outside the bar (1e-4 |b| + 1e-6 max|b|)")
sys.exit(1 if bad else 0)

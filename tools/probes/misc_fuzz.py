"""Fuzz of the remaining eval surfaces on random sizes against the oracle: the ResNet speech encoder (random B, F, T; both arithmetics), Linearfusion
(eval, random batch and widths, both outputs), LowFER's concatenation, feature_normalize / fuse_av (random widths incl. odd ones, rows of wild
scale), time / group means (random pointers incl. one-clip groups), LMCL / AAM / CE heads (random batch, classes; loss, logits, argmax).
   python tools/probes/misc_fuzz.py [n] [seed]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import torch

from deeplip_amd import arith, fusion, ops, weightgen as wg, _lib
from oracle import deeplip_oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
r = np.random.Generator(np.random.PCG64(seed))
bad, worst = 0, {}


def chk(tag, what, got, want, tol=1e-4, floor=1e-6):
    global bad
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    if got.shape != want.shape:
        print(f"{tag} {what}: shape {got.shape} vs {want.shape}   <-- OUTSIDE", flush=True)
        bad += 1
        return
    e = float(np.max(np.abs(got - want) - tol * np.abs(want)) / max(np.abs(want).max(), 1e-30)) if want.size else 0.0
    worst[what] = max(worst.get(what, 0.0), e)
    if not np.isfinite(e) or e > floor:
        print(f"{tag} {what}: {e:.3e}   <-- OUTSIDE", flush=True)
        bad += 1


from models.resnet import SpeakerEmbNet as ResNetSpk
CFG = {"arch": "resnet", "resnet": {"input_dim": 1, "hidden_dim": [64, 128, 256], "residual_block_layers": [3, 3, 3], "fc_layers": 1,
                                    "embedding_dim": 256, "pooling": "average"}}
rnet = ResNetSpk(CFG)
rsd = wg.fill_state_dict({k: tuple(v.shape) for k, v in rnet.state_dict().items()}, prefix="aresnet.")
rnet.load_state_dict({k: torch.from_numpy(v) for k, v in rsd.items()})
rnet.cuda().eval()
for i in range(n):
    try:
        # ---- ResNet speech encoder
        B, Fd, T = int(r.integers(1, 9)), int(r.choice([24, 40, 64, 80])), int(r.integers(9, 160))
        x = torch.from_numpy(wg.audio_input(B, Fd, T, key=f"mf.r{seed}.{i}")).unsqueeze(1)
        tag = f"audio-resnet B={B} F={Fd} T={T}"
        with torch.no_grad():
            ref = O.audio_resnet_embedding(O.to_torch_sd(rsd), x).numpy()
        for mode in ("f32", "f16x3"):
            arith.configure(mode)
            chk(tag, f"resnet {mode}", rnet.extract_embedding(x.cuda())[0].cpu().numpy(), ref, floor=3e-6)
            _lib.check_range(sync=True)
        arith.configure("f32")
        # ---- Linearfusion
        Din, Hd, B = int(r.choice([64, 512, 1024, 1536])), int(r.choice([64, 256, 512])), int(r.integers(1, 200))
        for ef in (True, False):
            m = fusion.Linearfusion(Din, Hd, 10, ef)
            sd = wg.fill_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, prefix=f"mf.lf{i}.")
            m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
            m.cuda().eval()
            xin = torch.from_numpy(r.standard_normal((B, Din)).astype(np.float32))
            with torch.no_grad():
                ref = O.linearfusion(O.to_torch_sd(sd), xin, ef).numpy()
            chk(f"linearfusion B={B} {Din}->{Hd} feats={ef}", "linearfusion", m(xin.cuda()).cpu().numpy(), ref, floor=3e-6)
        # ---- znorm / fuse_av / lowfer
        U, Da, Dv = int(r.integers(1, 300)), int(r.choice([2, 3, 100, 512, 513])), int(r.choice([2, 5, 256, 512]))
        a = (r.standard_normal((U, Da)) * np.exp(r.uniform(-10, 10, (U, 1)))).astype(np.float32)
        v = r.standard_normal((U, Dv)).astype(np.float32)
        chk(f"fuse_av U={U} {Da}+{Dv}", "fuse_av", fusion.fuse_av(torch.from_numpy(a).cuda(), torch.from_numpy(v).cuda()).cpu().numpy(),
            O.fuse_av(torch.from_numpy(a), torch.from_numpy(v)).numpy(), floor=3e-6)
        chk(f"znorm U={U} D={Da}", "feature_normalize", fusion.feature_normalize(torch.from_numpy(a).cuda()).cpu().numpy(),
            O.feature_normalize_torch(torch.from_numpy(a)).numpy(), floor=3e-6)
        e1 = r.standard_normal((U, Dv)).astype(np.float32)
        chk(f"lowfer U={U} D={Dv}", "lowfer", fusion.LowFER(Dv, Dv, 8)(torch.from_numpy(e1).cuda(), torch.from_numpy(v).cuda()).cpu().numpy(),
            O.lowfer(torch.from_numpy(e1), torch.from_numpy(v)).numpy(), floor=3e-6)
        # ---- group mean
        G = int(r.integers(1, 60))
        cnt = r.integers(1, 5, size=G)
        ptr = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
        cm = r.standard_normal((int(ptr[-1]), 512)).astype(np.float32)
        chk(f"group_mean G={G}", "group_mean", ops.group_mean(torch.from_numpy(cm).cuda(), torch.from_numpy(ptr).cuda()).cpu().numpy(),
            O.video_group_mean(torch.from_numpy(cm), ptr.tolist()).numpy(), floor=3e-6)
    except Exception as ex:
        print(f"iteration {i}: {type(ex).__name__}: {str(ex)[:200]}   <-- RAISED", flush=True)
        bad += 1
print("worst per surface:", {k: f"{v:.2e}" for k, v in worst.items()})
print(f"{bad} outside")
sys.exit(1 if bad else 0)

"""Model-level training fuzz: ONE forward + backward of the lip-clip model and of the speech encoders at RANDOM small (B, T) -- including the
corners (one clip, one frame, ragged lengths; the shortest crop an encoder takes; odd batches) -- loss / logits / every parameter's gradient
against the oracle's train-mode restatement in fp64: loss and logits 1e-4; gradient norms 2e-2; the worst gradient tensor within 3x the fp32
oracle's distance from fp64 OR 5e-2.  The last bar is that loose because at these sizes ONE LeakyReLU / PReLU pre-activation that rounding moves
across zero costs ~1e-2 of a tensor's largest gradient, and whether one does is a coin (tools/probes/train_fuzz_case.py: over 8 inputs of
one shape the error is bimodal, 7e-6 or ~1e-2, in f16x3 3 times, in the engine's f32 6 times, in the fp32 oracle once); an indexing or
padding rule gone wrong is an O(1) error.  Batches of fewer than 4 utterances are left out: a BatchNorm over two rows has no gradient to
its input (the output is +-1 whatever comes in), and what remains is rounding.   python tools/probes/train_fuzz.py [n_video] [n_audio] [seed]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import torch
import torch.nn.functional as F

from deeplip_amd import autograd as ag, _lib, weightgen as wg
from models.audio_models.loss import LMCL
from models.audio_models.tdnn import SpeakerEmbNet
from models.video_models.model import Lipreading
from oracle import deeplip_oracle as O

nv = int(sys.argv[1]) if len(sys.argv) > 1 else 8
na = int(sys.argv[2]) if len(sys.argv) > 2 else 12
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
r = np.random.Generator(np.random.PCG64(seed))
torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
DEV = "cuda"
bad = 0


def compare(tag, names, grads, loss, oracle, extra=None):
    global bad
    l64, g64, out64 = oracle(torch.float64)
    _, g32, _ = oracle(torch.float32)
    ours, floor, msgs = [], [], []
    if abs(loss - l64) > 1e-4 * abs(l64):
        msgs.append(f"loss {loss} vs {l64}")
    if extra is not None:
        e = float((extra.double() - out64.double()).abs().max() / out64.abs().max())
        if e > 1e-4:
            msgs.append(f"outputs {e:.2e}")
    for k in names:
        sc = float(g64[k].abs().max())
        if sc < 1e-9:
            if float(grads[k].abs().max()) > 1e-5:
                msgs.append(f"{k}: exact gradient 0, got {float(grads[k].abs().max()):.2e}")
            continue
        ours.append((float((grads[k] - g64[k]).abs().max()) / sc, k))
        floor.append(float((g32[k] - g64[k]).abs().max()) / sc)
        n64 = float(g64[k].norm())
        if n64 > 1e-7 and abs(float(grads[k].norm()) - n64) > 2e-2 * n64:
            msgs.append(f"{k}: norm {float(grads[k].norm()):.4e} vs {n64:.4e}")
    worst, who = max(ours)
    if worst > max(5e-2, 3.0 * max(floor)):
        msgs.append(f"worst gradient {who} {worst:.2e} (fp32 oracle floor {max(floor):.2e})")
    if np.mean([e for e, _ in ours]) > max(1e-2, 3.0 * np.mean(floor)):
        msgs.append(f"mean gradient error {np.mean([e for e, _ in ours]):.2e} vs floor {np.mean(floor):.2e}")
    print(f"{tag}: worst {worst:.2e} ({who}), fp32 floor {max(floor):.2e}" + ("   <-- " + "; ".join(msgs) if msgs else ""), flush=True)
    bad += bool(msgs)


# ---------------------------------------------------------------- lip-clip model
tcn = {"num_layers": 4, "kernel_size": [3, 5, 7], "dropout": 0.0, "dwpw": False, "width_mult": 1}
vnet = Lipreading(num_classes=54, relu_type="prelu", tcn_options=tcn, extract_feats=False)
vsd = wg.fill_state_dict({k: tuple(v.shape) for k, v in vnet.state_dict().items()}, prefix="vtrain.video.")
shapes = [(1, 1), (1, 2), (2, 1), (3, 5)] + [(int(r.integers(1, 5)), int(r.integers(1, 16))) for _ in range(max(0, nv - 4))]
for (B, T) in shapes[:nv]:
    vnet.load_state_dict({k: torch.from_numpy(v) for k, v in vsd.items()})
    vnet.to(DEV).train()
    vnet.zero_grad(set_to_none=True)
    x = torch.from_numpy(wg.video_input(B, frames=T, key=f"tf.v{seed}.{B}.{T}"))
    lab = torch.from_numpy(wg.labels(B, 54))
    lengths = [int(v) for v in r.integers(1, T + 1, size=B)]
    lengths[0] = T
    sd0 = {k: v.detach().cpu().clone() for k, v in vnet.state_dict().items()}
    tag = f"video B={B} T={T} lengths={lengths}"
    try:
        logits = vnet(x.to(DEV), lengths=lengths)
        loss = ag.margin_ce_loss(logits, lab.to(DEV))
        loss.backward()
        torch.cuda.synchronize()
        _lib.check_range(sync=True)
    except Exception as ex:
        print(f"{tag}: {type(ex).__name__}: {str(ex)[:200]}   <-- RAISED", flush=True)
        bad += 1
        continue
    names = [k for k, _ in vnet.named_parameters()]
    grads = {k: v.grad.detach().cpu().double() for k, v in vnet.named_parameters()}

    def oracle(dt):
        p = {k: (v.to(dt) if v.dtype.is_floating_point else v.clone()) for k, v in sd0.items()}
        for k in names:
            p[k].requires_grad_(True)
        lo = O.lipreading_logits_train(p, x.to(dt), lengths)
        ls = F.cross_entropy(lo, lab)
        ls.backward()
        return float(ls.detach()), {k: p[k].grad.double() for k in names}, lo.detach()
    compare(tag, names, grads, float(loss.detach()), oracle, logits.detach().cpu())

# ---------------------------------------------------------------- speech encoders
for i in range(na):
    arch = "etdnn" if i % 2 else "tdnn"
    dim = 24 if i % 3 else 80
    if arch == "etdnn":
        o = {"input_dim": dim, "hidden_dim": [512] * 9 + [1500], "context": O.ETDNN_CONTEXT, "tdnn_layers": 10, "embedding_dim": 512,
             "pooling": "statistic", "attention_hidden_size": 64, "bn_first": True}
        ctx = O.ETDNN_CONTEXT
    else:
        o = {"input_dim": dim, "hidden_dim": [512] * 4 + [1500], "context": O.TDNN_CONTEXT, "tdnn_layers": 5, "embedding_dim": 512,
             "pooling": "statistic", "attention_hidden_size": 64, "bn_first": True}
        ctx = O.TDNN_CONTEXT
    net = SpeakerEmbNet({"arch": arch, arch: o})
    sd = wg.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, prefix=f"tf.audio.{arch}{dim}.")
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.cuda().train()
    amin = net.frames_consumed() + 2
    B = int(r.integers(4, 24)) if i else 4
    T = amin if i < 2 else int(r.integers(amin, 160))
    crit = LMCL(512, 19, 30, 0.2).cuda()
    x = torch.from_numpy(wg.audio_input(B, dim, T, key=f"tf.a{seed}.{i}"))
    lab = torch.from_numpy(wg.labels(B, 19))
    sd0 = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    cw = crit.weights.detach().cpu().clone()
    tag = f"{arch} F={dim} B={B} T={T}"
    try:
        loss, _ = crit(net(x.cuda()), lab.cuda())
        loss.backward()
        torch.cuda.synchronize()
        _lib.check_range(sync=True)
    except Exception as ex:
        print(f"{tag}: {type(ex).__name__}: {str(ex)[:200]}   <-- RAISED", flush=True)
        bad += 1
        continue
    names = [k for k, _ in net.named_parameters()]
    grads = {k: v.grad.detach().cpu().double() for k, v in net.named_parameters()}

    def oracle(dt):
        p = {k: (v.to(dt) if v.dtype.is_floating_point else v.clone()) for k, v in sd0.items()}
        for k in names:
            p[k].requires_grad_(True)
        l, _ = O.lmcl(O.speaker_forward_train(p, x.to(dt), ctx), lab, cw.to(dt), 30, 0.2)
        l.backward()
        return float(l.detach()), {k: p[k].grad.double() for k in names}, None
    compare(tag, names, grads, float(loss.detach()), oracle)
print(f"{bad} outside")
sys.exit(1 if bad else 0)

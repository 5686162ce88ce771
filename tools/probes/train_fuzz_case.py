"""One shape of tools/probes/train_fuzz.py over several inputs and both training arithmetics: is an excess over the fp32 oracle's floor a
flipped LeakyReLU kink (comes and goes with the input, in either arithmetic) or a rule (stays)?
   python tools/probes/train_fuzz_case.py etdnn 24 10 78 [n_inputs]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import torch

from deeplip_amd import arith, _lib, weightgen as wg
from models.audio_models.loss import LMCL
from models.audio_models.tdnn import SpeakerEmbNet
from oracle import deeplip_oracle as O

arch, dim, B, T = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
n = int(sys.argv[5]) if len(sys.argv) > 5 else 6
torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
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
for i in range(n):
    x = torch.from_numpy(wg.audio_input(B, dim, T, key=f"tfc.{i}"))
    lab = torch.from_numpy(wg.labels(B, 19))
    res = {}
    for mode in ("f16x3", "f32"):
        arith.configure(mode)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        net.cuda().train()
        net.zero_grad(set_to_none=True)
        crit = LMCL(512, 19, 30, 0.2).cuda()
        torch.manual_seed(0)
        with torch.no_grad():
            crit.weights.copy_(torch.from_numpy(wg.fill_state_dict({"w": tuple(crit.weights.shape)}, prefix="tfc.crit.")["w"]))
        cw = crit.weights.detach().cpu().clone()
        sd0 = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
        loss, _ = crit(net(x.cuda()), lab.cuda())
        loss.backward()
        torch.cuda.synchronize()
        _lib.check_range(sync=True)
        names = [k for k, _ in net.named_parameters()]
        res[mode] = {k: v.grad.detach().cpu().double() for k, v in net.named_parameters()}

    def oracle(dt):
        p = {k: (v.to(dt) if v.dtype.is_floating_point else v.clone()) for k, v in sd0.items()}
        for k in names:
            p[k].requires_grad_(True)
        l, _ = O.lmcl(O.speaker_forward_train(p, x.to(dt), ctx), lab, cw.to(dt), 30, 0.2)
        l.backward()
        return {k: p[k].grad.double() for k in names}
    g64, g32 = oracle(torch.float64), oracle(torch.float32)

    def worst(g):
        out = []
        for k in names:
            sc = float(g64[k].abs().max())
            if sc > 1e-9:
                out.append((float((g[k] - g64[k]).abs().max()) / sc, k))
        return max(out)
    print(f"input {i}: f16x3 {worst(res['f16x3'])[0]:.2e} ({worst(res['f16x3'])[1]}), engine f32 {worst(res['f32'])[0]:.2e} ({worst(res['f32'])[1]}), "
          f"oracle fp32 {worst(g32)[0]:.2e} ({worst(g32)[1]})", flush=True)

"""One (B, T, lengths) of tools/probes/train_fuzz.py's lip-clip leg over several inputs and both training arithmetics: is an excess over the
fp32 oracle's floor a discrete choice that rounding flips (a PReLU kink, a max-pool winner: comes and goes with the input, in either arithmetic,
in the fp32 oracle too) or a rule (stays)?   python tools/probes/train_fuzz_case_video.py B T [n_inputs]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import torch
import torch.nn.functional as F

from deeplip_amd import arith, autograd as ag, _lib, weightgen as wg
from models.video_models.model import Lipreading
from oracle import deeplip_oracle as O

B, T = int(sys.argv[1]), int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 6
torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
tcn = {"num_layers": 4, "kernel_size": [3, 5, 7], "dropout": 0.0, "dwpw": False, "width_mult": 1}
net = Lipreading(num_classes=54, relu_type="prelu", tcn_options=tcn, extract_feats=False)
sd = wg.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, prefix="vtrain.video.")
lengths = [T] * B
for i in range(n):
    x = torch.from_numpy(wg.video_input(B, frames=T, key=f"tfcv.{i}"))
    lab = torch.from_numpy(wg.labels(B, 54))
    res = {}
    for mode in ("f16x3", "f32"):
        arith.configure(mode)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        net.cuda().train()
        net.zero_grad(set_to_none=True)
        sd0 = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
        loss = ag.margin_ce_loss(net(x.cuda(), lengths=lengths), lab.cuda())
        loss.backward()
        torch.cuda.synchronize()
        _lib.check_range(sync=True)
        names = [k for k, _ in net.named_parameters()]
        res[mode] = {k: v.grad.detach().cpu().double() for k, v in net.named_parameters()}

    def oracle(dt):
        p = {k: (v.to(dt) if v.dtype.is_floating_point else v.clone()) for k, v in sd0.items()}
        for k in names:
            p[k].requires_grad_(True)
        F.cross_entropy(O.lipreading_logits_train(p, x.to(dt), lengths), lab).backward()
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

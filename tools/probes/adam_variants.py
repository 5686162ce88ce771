"""Two eager steps of the lip-clip model: Adam(float lr) vs Adam(tensor lr, capturable=True) -- how far apart do the weights land?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from deeplip_amd import autograd as ag, weightgen as wg
from models.video_models.model import Lipreading
B, T = 4, 9
def run(capt, steps):
    tcn = {"num_layers": 4, "kernel_size": [3, 5, 7], "dropout": 0.0, "dwpw": False, "width_mult": 1}
    net = Lipreading(num_classes=54, relu_type="prelu", tcn_options=tcn, extract_feats=False)
    sd = wg.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, prefix="video.")
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.cuda().train()
    opt = torch.optim.Adam(net.parameters(), lr=torch.tensor(3e-4, device="cuda"), weight_decay=1e-4, capturable=True) if capt else \
          torch.optim.Adam(net.parameters(), lr=3e-4, weight_decay=1e-4)
    ln = torch.full((B,), T, dtype=torch.int32, device="cuda")
    gs = []
    for i in range(steps):
        x = torch.from_numpy(wg.video_input(B, frames=T, key=f"probe.v{i}")).cuda()
        lab = torch.from_numpy(wg.labels(B, 54)).cuda()
        opt.zero_grad(set_to_none=True)
        l = ag.margin_ce_loss(net(x, lengths=ln), lab)
        l.backward()
        gs = [p.grad.detach().clone() for p in net.parameters()]
        opt.step()
    torch.cuda.synchronize()
    return {n: p.detach().clone() for n, p in net.named_parameters()}, gs
for steps in (1, 2, 3):
    (a, ga), (b, gb) = run(False, steps), run(True, steps)
    w = max((float((a[k] - b[k]).abs().max()), k) for k in a)
    names = list(a.keys())
    g = max((float((u - v).abs().max() / (u.abs().max() + 1e-30)), n) for u, v, n in zip(ga, gb, names))
    print(steps, "weights", w, "| last gradients (rel)", g, flush=True)

"""From ONE state: an eager optimisation step vs the same step recorded and replayed -- gradients and weights compared."""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from deeplip_amd import autograd as ag, weightgen as wg
from models.video_models.model import Lipreading
B, T = 4, 9
tcn = {"num_layers": 4, "kernel_size": [3, 5, 7], "dropout": 0.0, "dwpw": False, "width_mult": 1}
net = Lipreading(num_classes=54, relu_type="prelu", tcn_options=tcn, extract_feats=False)
sd = wg.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, prefix="video.")
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
net.cuda().train()
opt = torch.optim.Adam(net.parameters(), lr=torch.tensor(3e-4, device="cuda"), weight_decay=1e-4, capturable=True)
x = torch.from_numpy(wg.video_input(B, frames=T, key="probe.v")).cuda()
lab = torch.from_numpy(wg.labels(B, 54)).cuda()
ln = torch.full((B,), T, dtype=torch.int32, device="cuda")
names = [n for n, _ in net.named_parameters()]
def one(do_opt=True):
    opt.zero_grad(set_to_none=True)
    l = ag.margin_ce_loss(net(x, lengths=ln), lab)
    l.backward()
    if do_opt:
        opt.step()
    return l
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    one()
    torch.cuda.synchronize()
    msd = copy.deepcopy(net.state_dict()); osd = copy.deepcopy(opt.state_dict())
    l_e = one()
    torch.cuda.synchronize()
    g_e = [p.grad.detach().clone() for p in net.parameters()]; w_e = [p.detach().clone() for p in net.parameters()]
    net.load_state_dict(msd); opt.load_state_dict(osd)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        l_g = one()
    g.replay(); torch.cuda.synchronize()
    g_g = [p.grad.detach().clone() for p in net.parameters()]; w_g = [p.detach().clone() for p in net.parameters()]
print("loss eager", float(l_e), "graph", float(l_g))
dg = sorted(((float((a - b).abs().max() / (a.abs().max() + 1e-30)), n) for a, b, n in zip(g_e, g_g, names)), reverse=True)[:5]
dw = sorted(((float((a - b).abs().max()), n) for a, b, n in zip(w_e, w_g, names)), reverse=True)[:5]
print("grad rel diff", dg)
print("weight abs diff", dw)

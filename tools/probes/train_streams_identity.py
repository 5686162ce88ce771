"""B = 32 x 29 frames, dropout off: four optimisation steps through TrainStepGraph with the independent branches on side streams vs
all on one stream -- parameters and BatchNorm statistics must be bit-identical (a race would show here at the real shapes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from deeplip_amd import autograd as ag, weightgen as wg
from deeplip_amd.train_plan import TrainStepGraph
from models.video_models.model import Lipreading
B, T = 32, 29
def run(branch_streams):
    tcn = {"num_layers": 4, "kernel_size": [3, 5, 7], "dropout": 0.0, "dwpw": False, "width_mult": 1}
    net = Lipreading(num_classes=54, relu_type="prelu", tcn_options=tcn, extract_feats=False)
    sd = wg.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, prefix="video.")
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.cuda().train()
    opt = torch.optim.Adam(net.parameters(), lr=torch.tensor(3e-4, device="cuda"), weight_decay=1e-4, capturable=True, fused=True)
    ln = torch.full((B,), T, dtype=torch.int32, device="cuda")
    def one(xb, lb, l_):
        opt.zero_grad(set_to_none=True)
        l = ag.margin_ce_loss(net(xb, lengths=l_), lb)
        l.backward()
        opt.step()
        return l
    plan = TrainStepGraph(one, eager_steps=1, branch_streams=branch_streams)
    ls = []
    for i in range(5):
        x = torch.from_numpy(wg.video_input(B, frames=T, key=f"ident.v{i}")).cuda()
        lab = torch.from_numpy((wg.labels(B, 54) + 5 * i) % 54).cuda()
        ls.append(float(plan.step(x, lab, ln).detach()))
    plan.finish()
    return ls, {k: v.detach().clone() for k, v in net.state_dict().items()}
la, sa = run(True)
lb, sb = run(False)
print("losses", la, lb)
bad = [k for k in sa if not torch.equal(sa[k], sb[k])]
print("bit-identical" if not bad and la == lb else f"DIFFERENT: {len(bad)} tensors, e.g. {bad[:3]}")

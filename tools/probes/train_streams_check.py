"""Forked branches (video.BRANCH_STREAMS) with the DEFAULT stream / another stream as main stream, eagerly and recorded: the range
flag after every step.  (How the shared constant vectors filled on a side stream were found: step 0 on the default stream.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from deeplip_amd import autograd as ag, weightgen as wg, _lib
from deeplip_amd.train_plan import TrainStepGraph
from models.video_models.model import Lipreading
B, T = 32, 29
tcn = {"num_layers": 4, "kernel_size": [3, 5, 7], "dropout": 0.2, "dwpw": False, "width_mult": 1}
net = Lipreading(num_classes=54, relu_type="prelu", tcn_options=tcn, extract_feats=False)
sd = wg.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, prefix="video.")
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
net.cuda().train()
opt = torch.optim.Adam(net.parameters(), lr=torch.tensor(3e-4, device="cuda"), weight_decay=1e-4, capturable=True, fused=True)
x = torch.from_numpy(wg.video_input(B, frames=T, key="bench.vtrain")).cuda()
lab = torch.from_numpy(wg.labels(B, 54)).cuda()
ln = torch.full((B,), T, dtype=torch.int32, device="cuda")
def one(xb, lb, l_):
    opt.zero_grad(set_to_none=True)
    l = ag.margin_ce_loss(net(xb, lengths=l_), lb)
    l.backward()
    opt.step()
    return l
mode = sys.argv[1]
from deeplip_amd import video as _v
_v.BRANCH_STREAMS = True
if mode == "eager-side":
    s_ = torch.cuda.Stream(); s_.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s_):
        for i in range(6):
            l = one(x, lab, ln)
            torch.cuda.synchronize()
            try:
                _lib.check_range(sync=True); print("eager (non-default main stream) step", i, "ok", float(l.detach()))
            except Exception as e:
                print("eager (non-default main stream) step", i, "RANGE", str(e)[:80]); break
elif mode == "eager":
    for i in range(6):
        l = one(x, lab, ln)
        torch.cuda.synchronize()
        try:
            _lib.check_range(sync=True); print("eager step", i, "ok", float(l.detach()))
        except Exception as e:
            print("eager step", i, "RANGE", str(e)[:80]); break
else:
    plan = TrainStepGraph(one, eager_steps=1)
    for i in range(6):
        l = plan.step(x, lab, ln)
        try:
            plan.finish(); print("plan step", i, "recorded" if plan.recorded else "eager", "ok", float(l.detach()))
        except Exception as e:
            print("plan step", i, "RANGE", str(e)[:80]); break

"""Every parameter gradient of the B = 32 x 29 training step against the fp64 oracle (development probe)."""
import os, sys
sys.path.insert(0, ".")
import numpy as np, torch, torch.nn.functional as F
from deeplip_amd import weightgen as wg, autograd as ag
from oracle import deeplip_oracle as O
from models.video_models.model import Lipreading
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = 29
tcn = {"num_layers": 4, "kernel_size": [3, 5, 7], "dropout": 0.0, "dwpw": False, "width_mult": 1}
net = Lipreading(num_classes=54, relu_type="prelu", tcn_options=tcn, extract_feats=False)
sd = wg.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, prefix="vtrain.video.")
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
net.cuda().train()
x = torch.from_numpy(wg.video_input(B, frames=T, key="vtrain.full"))
lab = torch.from_numpy(wg.labels(B, 54))
lengths = [T - (i % 5) for i in range(B)]; lengths[0] = T
sd0 = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
loss = ag.margin_ce_loss(net(x.cuda(), lengths=lengths), lab.cuda())
loss.backward(); torch.cuda.synchronize()
grads = {k: v.grad.detach().cpu().double() for k, v in net.named_parameters()}
torch.set_num_threads(16)
p = {k: v.double() for k, v in sd0.items()}
names = [k for k, _ in net.named_parameters()]
for k in names: p[k].requires_grad_(True)
rl = F.cross_entropy(O.lipreading_logits_train(p, x.double(), lengths), lab); rl.backward()
print("loss", float(loss), float(rl), flush=True)
# the same oracle in fp32 (what the reference itself computes in): its distance from fp64 is the noise floor of this comparison
q = {k: v.float() for k, v in sd0.items()}
for k in names: q[k].requires_grad_(True)
ql = F.cross_entropy(O.lipreading_logits_train(q, x.float(), lengths), lab); ql.backward()
rows = []
for k in names:
    g64 = p[k].grad; sc = float(g64.abs().max())
    if sc < 1e-9: continue
    rows.append((float((grads[k] - g64).abs().max()) / sc, float((q[k].grad.double() - g64).abs().max()) / sc, k, sc))
print("   ours-vs-f64  torch-f32-vs-f64")
for e, e32, k, sc in sorted(rows, reverse=True)[:30]:
    print(f"{e:10.3e}  {e32:10.3e}  {k:55s} max|g64| {sc:.3e}")
import numpy as np
r = np.array([[a, b] for a, b, _, _ in rows])
print("worst ours", r[:, 0].max(), "worst torch fp32", r[:, 1].max(), "median ratio ours/fp32", float(np.median(r[:, 0] / np.maximum(r[:, 1], 1e-12))))

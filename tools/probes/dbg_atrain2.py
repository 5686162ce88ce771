import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch, torch.nn.functional as F
from conftest import rel_err
from deeplip_amd import weightgen as wg, autograd as ag
from models.audio_models.loss import LMCL
from models.audio_models.tdnn import SpeakerEmbNet
from oracle import deeplip_oracle as O
from test_oracle_golden import atrain_shapes, ATRAIN_CONTEXT
def load(module, prefix):
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    sd = wg.fill_state_dict(shapes, prefix=prefix)
    module.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return module.to("cuda")
opts = {"arch": "tdnn", "tdnn": {"input_dim": 24, "hidden_dim": [512] * 4 + [1500], "context": ATRAIN_CONTEXT, "tdnn_layers": 5, "embedding_dim": 512, "pooling": "statistic", "attention_hidden_size": 64, "bn_first": True}}
net = load(SpeakerEmbNet(opts), "atrain.audio.").train(); crit = load(LMCL(512, 57, 30, 0.2), "atrain.lmcl.").train()
x = torch.from_numpy(wg.audio_input(8, 24, 120, key="atrain.x")).cuda(); lab = torch.from_numpy(wg.labels(8, 57)).cuda()
saved = {}
orig_pool = ag.meanstd_pool
def pool_hook(h):
    h.register_hook(lambda g: saved.__setitem__("d_h", g.detach().cpu()))
    saved["h"] = h.detach().cpu()
    y = orig_pool(h)
    y.register_hook(lambda g: saved.__setitem__("d_pooled", g.detach().cpu()))
    return y
ag.meanstd_pool = pool_hook
out = net(x); loss, logits = crit(out, lab); loss.backward()
# fp64 oracle with the same taps
p = O.to_torch_sd(wg.fill_state_dict(atrain_shapes(), prefix="atrain.audio."))
p = {k: (v.double() if v.is_floating_point() else v) for k, v in p.items()}
cw = O.to_torch_sd(wg.fill_state_dict({"weights": (57, 512)}, prefix="atrain.lmcl."))["weights"].double().requires_grad_()
for k in p:
    if p[k].is_floating_point() and "running" not in k: p[k].requires_grad_()
h = x.cpu().double()
for i, ctx in enumerate(ATRAIN_CONTEXT):
    _, d = O.tdnn_dilation(ctx)
    h = F.conv1d(h, p[f"tdnn.{i}.context_layer.weight"], p[f"tdnn.{i}.context_layer.bias"], dilation=d)
    h = F.leaky_relu(O._bn_train(h, p, f"tdnn.{i}.bn"), 0.2)
h.retain_grad(); hh = h
pooled = O.mean_std_pooling(h); pooled.retain_grad()
x_a = F.linear(pooled, p["fc1.weight"], p["fc1.bias"])
t = F.leaky_relu(O._bn_train(x_a, p, "bn1"), 0.2)
xv = F.linear(t, p["fc2.weight"], p["fc2.bias"])
o = F.leaky_relu(O._bn_train(xv, p, "bn2"), 0.2)
l, lg = O.lmcl(o, lab.cpu(), cw, 30, 0.2); l.backward()
print("h (tdnn out) fwd err", rel_err(saved["h"].permute(0, 2, 1).numpy(), hh.detach().numpy()))
print("d_pooled err", rel_err(saved["d_pooled"].numpy(), pooled.grad.numpy()))
dh = saved["d_h"].permute(0, 2, 1).numpy(); rh = hh.grad.numpy()
print("d_h err", rel_err(dh, rh))
e = np.abs(dh - rh); idx = np.unravel_index(e.argmax(), e.shape); print("worst at (b,c,t)", idx, dh[idx], rh[idx], "std of that channel:", float(pooled.detach()[idx[0], 1500 + idx[1]]), "mean", float(pooled.detach()[idx[0], idx[1]]))

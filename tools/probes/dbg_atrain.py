import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from conftest import rel_err
from deeplip_amd import weightgen as wg
from models.audio_models.loss import LMCL
from models.audio_models.tdnn import SpeakerEmbNet
g = np.load(os.path.join(R, "tests/golden/audio_train_golden.npz"))
def load(module, prefix):
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    sd = wg.fill_state_dict(shapes, prefix=prefix)
    module.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return module.to("cuda")
opts = {"arch": "tdnn", "tdnn": {"input_dim": 24, "hidden_dim": [512] * 4 + [1500], "context": [[-2, -1, 0, 1, 2], [-2, 0, 2], [-3, 0, 3], [0], [0]], "tdnn_layers": 5, "embedding_dim": 512, "pooling": "statistic", "attention_hidden_size": 64, "bn_first": True}}
net = load(SpeakerEmbNet(opts), "atrain.audio.").train(); crit = load(LMCL(512, 57, 30, 0.2), "atrain.lmcl.").train()
x = torch.from_numpy(wg.audio_input(8, 24, 120, key="atrain.x")).cuda(); lab = torch.from_numpy(wg.labels(8, 57)).cuda()
out = net(x); loss, logits = crit(out, lab); loss.backward()
print("loss", float(loss), float(g["loss0"]), "out err", rel_err(out.detach().cpu().numpy(), g["output0"]))
for k, v in net.named_parameters():
    ref = g[f"gradnorm_{k}"]
    print(f"{k:34s} norm {float(v.grad.double().norm()):.6e} ref {ref[0]:.6e}  ratio-1 {float(v.grad.double().norm())/max(ref[0],1e-30)-1:+.2e}")
for name, key in (("tdnn.0.context_layer.weight", "grad_tdnn0_w"), ("tdnn.0.bn.weight", "grad_tdnn0_bn_w")):
    print(name, rel_err(dict(net.named_parameters())[name].grad.cpu().numpy(), g[key]))

# three-way: HIP vs fp64 oracle vs fp32 golden
from oracle import deeplip_oracle as O
sys.path.insert(0, os.path.join(R, "tests"))
from test_oracle_golden import atrain_shapes, ATRAIN_CONTEXT
p = O.to_torch_sd(wg.fill_state_dict(atrain_shapes(), prefix="atrain.audio."))
p = {k: (v.double() if v.is_floating_point() else v) for k, v in p.items()}
cw = O.to_torch_sd(wg.fill_state_dict({"weights": (57, 512)}, prefix="atrain.lmcl."))["weights"].double().requires_grad_()
for k in p:
    if p[k].is_floating_point() and "running" not in k: p[k].requires_grad_()
o = O.speaker_forward_train(p, x.cpu().double(), ATRAIN_CONTEXT)
l, lg = O.lmcl(o, lab.cpu(), cw, 30, 0.2); l.backward()
mine = dict(net.named_parameters())
for name, key in (("tdnn.0.context_layer.weight", "grad_tdnn0_w"), ("tdnn.0.bn.weight", "grad_tdnn0_bn_w"), ("bn2.weight", "grad_bn2_w"), ("tdnn.4.bn.bias", "grad_tdnn4_bn_b")):
    t = p[name].grad.numpy()
    print(f"{name:30s} HIP vs fp64 {rel_err(mine[name].grad.cpu().numpy(), t):.2e}   golden(fp32 ref) vs fp64 {rel_err(g[key], t):.2e}   HIP vs golden {rel_err(mine[name].grad.cpu().numpy(), g[key]):.2e}")

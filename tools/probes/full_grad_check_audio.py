"""Every parameter gradient of the B = 256 x 300 speech-encoder training step against the fp64 oracle (development probe)."""
import sys
sys.path.insert(0, ".")
import numpy as np, torch, torch.nn.functional as F
from deeplip_amd import weightgen as wg
from oracle import deeplip_oracle as O
from models.audio_models.loss import LMCL
from models.audio_models.tdnn import SpeakerEmbNet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
CTX = O.ETDNN_CONTEXT
et = {"input_dim": 24, "hidden_dim": [512] * 9 + [1500], "context": CTX, "tdnn_layers": 10, "embedding_dim": 512,
      "pooling": "statistic", "attention_hidden_size": 64, "bn_first": True}
net = SpeakerEmbNet({"arch": "etdnn", "etdnn": et})
sd = wg.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, prefix="audio.")
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
net.cuda().train()
crit = LMCL(512, 57, 30, 0.2).cuda()
x = torch.from_numpy(wg.audio_input(B, 24, 300, key="probe.atrain"))
lab = torch.from_numpy(wg.labels(B, 57))
sd0 = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
cw = crit.weights.detach().cpu().clone()
loss, _ = crit(net(x.cuda()), lab.cuda())
loss.backward(); torch.cuda.synchronize()
grads = {k: v.grad.detach().cpu().double() for k, v in net.named_parameters()}
names = [k for k, _ in net.named_parameters()]
torch.set_num_threads(16)
res = {}
for dt in (torch.float64, torch.float32):
    p = {k: (v.to(dt) if v.dtype.is_floating_point else v.clone()) for k, v in sd0.items()}
    for k in names: p[k].requires_grad_(True)
    emb = O.speaker_forward_train(p, x.to(dt), CTX)
    l, _ = O.lmcl(emb, lab, cw.to(dt), 30, 0.2)
    l.backward()
    res[dt] = (float(l.detach()), {k: p[k].grad.double() for k in names})
print("loss", float(loss.detach()), res[torch.float64][0], res[torch.float32][0])
g64, g32 = res[torch.float64][1], res[torch.float32][1]
rows = []
for k in names:
    sc = float(g64[k].abs().max())
    if sc < 1e-12: continue
    rows.append((float((grads[k] - g64[k]).abs().max()) / sc, float((g32[k] - g64[k]).abs().max()) / sc, k, sc))
for e, e32, k, sc in sorted(rows, reverse=True)[:12]:
    print(f"{e:10.3e}  {e32:10.3e}  {k:45s} max|g64| {sc:.3e}")

import os, sys, time
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/deeplip_amd") else os.environ["GRAFT_REPO_ROOT"])
import torch
from deeplip_amd import autograd as ag, weightgen as wg, _lib
from models.video_models.model import Lipreading
B, T = 32, 29
tcn = {"num_layers": 4, "kernel_size": [3, 5, 7], "dropout": 0.2, "dwpw": False, "width_mult": 1}
net = Lipreading(num_classes=54, relu_type="prelu", tcn_options=tcn, extract_feats=False)
sd = wg.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, prefix="video.")
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
net.cuda().train()
opt = torch.optim.Adam(net.parameters(), lr=3e-4, weight_decay=1e-4, capturable=True)
x = torch.from_numpy(wg.video_input(B, frames=T, key="bench.vtrain")).cuda()
lab = torch.from_numpy(wg.labels(B, 54)).cuda()
lengths = torch.full((B,), T, dtype=torch.int32, device='cuda')
def step():
    opt.zero_grad(set_to_none=True)
    loss = ag.margin_ce_loss(net(x, lengths=lengths), lab)
    loss.backward()
    opt.step()
    return loss
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        l = step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
print("eager loss", float(l))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    sl = step()
torch.cuda.synchronize()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    g.replay()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print(f"graph replay: {dt*1e3:.2f} ms/step = {B/dt:.0f} clips/s, loss {float(sl):.4f}")
_lib.check_range(sync=True)

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from deeplip_amd import ops, packing
torch.manual_seed(0)
def run(N,H,W,C,K,R,S,pad, tile=None, show=False):
    if tile is not None: os.environ["DLIP_CONV_DMA_TILE"]=str(tile)
    x = torch.randn(N,H,W,C).cuda(); w = torch.randn(K,R,S,C, dtype=torch.float64)/np.sqrt(C*R*S)
    ws, sc = packing.split_weights(w); ws, sc = ws.cuda(), sc.cuda()
    b = torch.randn(K).cuda()
    kw = dict(pad=(pad if H > 1 else 0,pad), w_scale=sc)
    xs = ops.split_pack(x); xv = ops.split_unpack(xs)
    base = ops.conv_nhwc(xv, ws, b, **kw)
    y = ops.conv_nhwc(xs, ws, b, x_split=True, **kw)
    torch.cuda.synchronize()
    d = (y-base).abs().reshape(-1, K).amax(dim=1).cpu().numpy()
    bad = np.nonzero(d > 1e-4*float(base.abs().max()))[0]
    print(f"N{N} H{H} W{W} C{C} K{K} R{R}S{S} tile={tile}: maxerr {d.max():.3e} bad rows {len(bad)} of {len(d)}", flush=True)
    if show and len(bad):
        good = np.setdiff1d(np.arange(len(d)), bad)
        print("  good rows:", good[:60])
        dk = (y-base).abs().reshape(-1, K).amax(dim=0).cpu().numpy()
        print("  bad cols:", np.nonzero(dk > 1e-4*float(base.abs().max()))[0][:20], "of", K)
run(3,22,22,64,64,3,3,1,3,True)
run(9,6,6,256,256,3,3,1,3,True)
run(3,22,22,128,64,3,3,1,3)
run(3,22,22,64,128,3,3,1,3)
run(3,22,22,64,64,1,1,0,3)
run(3,6,6,64,64,3,3,1,3)
run(30,6,6,64,64,3,3,1,3)
run(1,1,700,64,64,1,3,1,3)
run(1,1,700,256,64,1,3,1,3)
run(1,1,700,64,256,1,3,1,3)

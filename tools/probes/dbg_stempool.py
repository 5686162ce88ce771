import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from deeplip_amd import ops, packing
torch.manual_seed(0)
def run(B,T,H,W):
    x = (torch.randn(B,T,H,W)*2).cuda()
    w = torch.randn(64,1,5,7,7, dtype=torch.float64)/np.sqrt(245)
    b = (torch.randn(64)*0.1).cuda(); sl = (torch.rand(64)*0.3).cuda()
    img, sc = packing.split_stem_weights(w); img, sc = img.cuda(), sc.cuda()
    ref = ops.split_unpack(ops.maxpool3x3s2(ops.stem3d(x, img, b, sl, w_scale=sc), out_split=True)).cpu().numpy()
    y = ops.split_unpack(ops.stem3d_pool(x, img, b, sl, sc)).cpu().numpy()
    bad = np.abs(y-ref) > 1e-6
    print(f"B{B} T{T} H{H} W{W}: bad {bad.sum()} of {bad.size}")
    if bad.any():
        print("  frames:", np.nonzero(bad.any(axis=(1,2,3)))[0][:10], " rows:", np.nonzero(bad.any(axis=(0,2,3)))[0], "\n  cols:", np.nonzero(bad.any(axis=(0,1,3)))[0], "\n  chans:", np.nonzero(bad.any(axis=(0,1,2)))[0])
        f,r,c,k = [v[0] for v in np.nonzero(bad)]
        print("  first:", (f,r,c,k), y[f,r,c,k], ref[f,r,c,k])
run(1,1,88,88); run(1,1,24,24); run(1,3,16,16); run(2,7,88,88)

def run2(B,T,H,W):
    import torch.nn.functional as F
    x = (torch.randn(B,T,H,W)*2).cuda()
    w = torch.randn(64,1,5,7,7, dtype=torch.float64)/np.sqrt(245)
    b = (torch.randn(64)*0.1).cuda(); sl = (torch.rand(64)*0.3).cuda()
    img, sc = packing.split_stem_weights(w); img, sc = img.cuda(), sc.cuda()
    act = ops.stem3d(x, img, b, sl, w_scale=sc)            # [N,Ho,Wo,64] fp32
    y = ops.split_unpack(ops.stem3d_pool(x, img, b, sl, sc)).cpu()
    a = act.cpu()
    N, Ho, Wo, _ = a.shape
    f, r, k = 0, 0, 0
    print("act row0 ch0 :", a[f, 0, :, k].numpy().round(3))
    print("act row1 ch0 :", a[f, 1, :, k].numpy().round(3))
    print("mine prow0   :", y[f, 0, :, k].numpy().round(3))
    ref = F.max_pool2d(a.permute(0,3,1,2), 3, 2, 1).permute(0,2,3,1)
    print("ref  prow0   :", ref[f, 0, :, k].numpy().round(3))
run2(1,1,16,16)

def run3(B,T,H,W):
    import torch.nn.functional as F
    x = (torch.randn(B,T,H,W)*2).cuda()
    w = torch.randn(64,1,5,7,7, dtype=torch.float64)/np.sqrt(245)
    b = (torch.randn(64)*0.1).cuda(); sl = (torch.rand(64)*0.3).cuda()
    img, sc = packing.split_stem_weights(w); img, sc = img.cuda(), sc.cuda()
    a = ops.stem3d(x, img, b, sl, w_scale=sc).cpu()
    y = ops.split_unpack(ops.stem3d_pool(x, img, b, sl, sc)).cpu()
    ref = F.max_pool2d(a.permute(0,3,1,2), 3, 2, 1).permute(0,2,3,1)
    bad = (y-ref).abs() > 1e-6
    idx = torch.nonzero(bad)
    print("nbad", len(idx))
    for f,r,c,k in idx[:6].tolist():
        win = a[f, max(0,2*r-1):2*r+2, max(0,2*c-1):2*c+2, k]
        print((f,r,c,k), "mine", float(y[f,r,c,k]), "ref", float(ref[f,r,c,k]), "window:\n", win.numpy().round(4))
        # where does mine's value come from?
        loc = torch.nonzero((a[f,:,:,k]-y[f,r,c,k]).abs() < 1e-6)
        print("   mine's value found at stem (row,col):", loc.tolist()[:5])
run3(1,1,16,16)

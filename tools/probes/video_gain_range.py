"""Does the f16x3 range guard see a float lip clip whose gain puts it below the line?  The clip at gains 1, 2^-14, 2^10, 2^16 through embed() and
forward() under plain f16x3, against the exact mode.  (How the stem's shared evidence slot was found: 2^-14 went unreported, EXPERIMENTS R6.12.)
   python tools/probes/video_gain_range.py"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
from deeplip_amd import _lib, arith, packing, weightgen as wg
from models.video_models.model import Lipreading
tcn = {"num_layers": 4, "kernel_size": [3, 5, 7], "dropout": 0.2, "dwpw": False, "width_mult": 1}
net = Lipreading(num_classes=54, relu_type="prelu", tcn_options=tcn, extract_feats=True)
sd = wg.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, prefix="video.")
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
net.eval().cuda()
x = torch.from_numpy(wg.video_input(2, frames=29, key="vg"))
print("clip max", float(x.abs().max()))
for g in (1.0, 2.0**-14, 2.0**10, 2.0**16):
    arith.configure("f32")
    want = net.embed((x*g).cuda()).cpu()
    arith.configure("f16x3")
    try:
        got = net.embed((x*g).cuda()).cpu()
        _lib.check_range(sync=True)
        print(g, "no report; rel err", float((got-want).abs().max()/want.abs().max()))
    except _lib.DeepLipRangeError as e:
        print(g, "raised:", str(e)[:120])
    try:
        got = net(((x*g)[:, :, :9]).cuda(), None).cpu()
        _lib.check_range(sync=True)
        print(g, "forward no report")
    except _lib.DeepLipRangeError as e:
        print(g, "forward raised:", str(e)[:120])

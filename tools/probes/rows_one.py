"""The k = 1 TDNN GEMM (B = 64 x 296 frames, 512 -> 512, split in / out) launched 12 times: python tools/probes/rows_one.py MODE
(MODE = dlip_debug_set(DBG_ROWS): 0 ring kernel, 5 rows kernel at 160 rows)"""
import sys
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import numpy as np, torch
from deeplip_amd import _lib, ops, packing
g = torch.Generator().manual_seed(1)
x = ops.split_pack((torch.randn(64, 296, 512, generator=g) * 1.5).cuda())
ws, sc = packing.split_weights((torch.randn(512, 1, 512, generator=g) / np.sqrt(512)).double())
ws, sc = ws.cuda(), sc.cuda()
b = (torch.randn(512, generator=g) * 0.1).cuda()
slope = torch.full((512,), 0.2).cuda()
_lib.debug_set(_lib.DBG_ROWS, int(sys.argv[1]))
for _ in range(12):
    ops.conv1d_ntc(x, ws, b, slope=slope, w_scale=sc, x_split=True, out_split=True)
torch.cuda.synchronize()

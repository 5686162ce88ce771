"""Which (row, channel) lands where: the rows kernel on an identity GEMM (y[m][k] = x[m][k] = 1000 m + k)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from deeplip_amd import _lib, ops, packing
M, C = 320, 256
x = (torch.arange(M).view(1, M, 1) * 1.0 + torch.arange(C).view(1, 1, C) / 1024.0).float()
w = torch.eye(C).view(C, 1, C)
ws, sc = packing.split_weights(w.double())
xs = ops.split_pack(x.cuda())
for mode in (0, 5):
    _lib.debug_set(_lib.DBG_ROWS, mode)
    for osp in (False, True):
        y = ops.conv1d_ntc(xs, ws.cuda(), None, w_scale=sc.cuda(), x_split=True, out_split=osp)
        y = (ops.split_unpack(y) if osp else y).cpu()[0]
        bad = (y != x[0]).nonzero()
        print("mode", mode, "split" if osp else "fp32", "mismatches", len(bad))
        if len(bad):
            for m in (0, 1, 17):
                got = y[m]
                print("  row", m, "channels got:", [round(float((v - int(v)) * 1024)) for v in got[:40]], "rows got:", sorted(set(int(v) for v in got.tolist()))[:6])

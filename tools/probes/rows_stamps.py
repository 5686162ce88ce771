"""In-kernel stamps of the rows kernel (lab build): one stamped launch per layer shape.
   python -m deeplip_amd.build --lab && DLIP_LIB_PATH=deeplip_amd/lib/libdeeplip_hip_lab.so DLIP_STAMP_PRINT=1 python tools/probes/rows_stamps.py [B]"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from deeplip_amd import _lib, ops, packing
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
g = torch.Generator().manual_seed(1)
for name, T, C, K, S, dil in [("k1", 296, 512, 512, 1, 1), ("k3d2", 296, 512, 512, 3, 2)]:
    x = ops.split_pack((torch.randn(B, T, C, generator=g) * 1.5).cuda())
    w = torch.randn(K, S, C, generator=g) / np.sqrt(C * S)
    ws, sc = packing.split_weights(w.double())
    ws, sc = ws.cuda(), sc.cuda()
    b = (torch.randn(K, generator=g) * 0.1).cuda()
    slope = torch.full((K,), 0.2).cuda()
    for mi in (5, 4):
        _lib.debug_set(_lib.DBG_ROWS, mi)
        for _ in range(3):      # warm, then the stamped one prints
            print(name, "mi", mi, flush=True)
            ops.conv1d_ntc(x, ws, b, dilation=dil, slope=slope, w_scale=sc, x_split=True, out_split=True)
        torch.cuda.synchronize()

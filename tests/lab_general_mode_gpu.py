"""LAB-ONLY kernel tests (not collected by `pytest tests`: the file name does not match test_*.py).

The rows kernel's GENERAL mode (conv_rows_f16x3.hip: 2-D filters, residual, second source, balanced split with a per-wave hand-off)
was built in round 4, measured 2 - 18 % slower than the ring kernel on the trunk layers it was meant for, and is compiled into the lab
library only since round 5 (python -m deeplip_amd.build --lab).  Run on a GPU box with that library:

    DLIP_LIB_PATH=$PWD/deeplip_amd/lib/libdeeplip_hip_lab.so python -m pytest tests/lab_general_mode_gpu.py -q -m gpu
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err
from test_kernels_gpu import TOL, _split_ref_value, ops, rnd  # noqa: F401  (`ops` is a fixture)

pytestmark = pytest.mark.gpu


ROWS2D_CASES = [  # N, H, W, C, K, R (= S), stride, pad, dil, residual, post-affine; what it stands for
    (70, 6, 6, 256, 256, 3, 1, 1, 1, True, False),     # layer 3's block convolution: 16 tiles of 160 rows, 72 slices, many parts per tile
    (33, 3, 3, 512, 512, 3, 1, 1, 1, True, False),     # layer 4: two column blocks, 144 slices, M = 297 (ragged last tile)
    (9, 12, 12, 128, 256, 3, 2, 1, 1, False, False),   # stride 2 (a down-sampling block's first convolution)
    (5, 9, 7, 64, 320, 3, 1, 2, 2, False, True),       # dilation 2, padding 2, K tail (320 = 256 + 64), post-affine order, W != H
    (3, 5, 5, 32, 96, 1, 1, 0, 1, True, False),        # a one-slice reduction on one partial tile: a single workgroup, no split
    (40, 6, 6, 256, 256, 1, 1, 0, 1, True, False),     # 1x1 with residual: 8 slices per tile, several whole tiles per range
    (2, 20, 20, 96, 192, 5, 1, 2, 1, False, False),    # 25 taps: the mask's upper bits
]


@pytest.fixture
def force_rows2d():
    """The rows kernel's general mode is compiled into the LAB library only (round 5: a measured negative result stays out of
    libdeeplip_hip.so).  Run these tests with DLIP_LIB_PATH=deeplip_amd/lib/libdeeplip_hip_lab.so (python -m deeplip_amd.build --lab);
    on the product library dlip_debug_set(7, 1) is refused and they skip."""
    from deeplip_amd import _lib

    def force(v):
        try:
            _lib.debug_set(_lib.DBG_ROWS2D, v)
        except _lib.DeepLipHipError:
            pytest.skip("the rows kernel's general mode exists in the lab library only (DLIP_LIB_PATH=.../libdeeplip_hip_lab.so)")
    yield force
    _lib.debug_set(_lib.DBG_ROWS2D, -1)
    _lib.debug_set(_lib.DBG_STREAMK, -1)


def _rows2d_inputs(case):
    from deeplip_amd import packing
    N, H, W, C, K, R, stride, pad, dil, use_res, post = case
    x = _split_ref_value(rnd(N, H, W, C, seed=81) * 2.0)
    w = rnd(K, R, R, C, seed=82, scale=1.0 / np.sqrt(C * R * R))
    b = rnd(K, seed=83, scale=0.1)
    slope = torch.rand(K, generator=torch.Generator().manual_seed(21)) * 0.3
    ws, sc = packing.split_weights(w.double())
    kw = dict(stride=(stride, stride), pad=(pad, pad), dil=(dil, dil), slope=slope.cuda(), w_scale=sc.cuda(), x_split=True)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), b.double(), stride=stride, padding=pad, dilation=dil)
    res = None
    if use_res:
        res = _split_ref_value(rnd(N, ref.shape[2], ref.shape[3], K, seed=84))
        ref = ref + res.permute(0, 3, 1, 2).double()
    ref = torch.where(ref >= 0, ref, ref * slope.double().view(1, K, 1, 1))
    if post:
        ps = 0.5 + torch.rand(K, generator=torch.Generator().manual_seed(22))
        pt = rnd(K, seed=23, scale=0.1)
        kw.update(post_scale=ps.cuda(), post_shift=pt.cuda())
        ref = ref * ps.double().view(1, K, 1, 1) + pt.double().view(1, K, 1, 1)
    return x, ws, b, res, kw, ref.permute(0, 2, 3, 1)


@pytest.mark.parametrize("case", ROWS2D_CASES, ids=lambda c: "x".join(str(int(v)) for v in c))
def test_conv_rows_general_mode(ops, case, force_rows2d):
    """conv_rows_f16x3_kernel's general mode FORCED onto 2-D convolutions of every kind it takes (padding, stride, dilation, up to
    32 taps, residual in the split format, post-affine without residual, K tails, ragged row tiles) with its balanced split:
    (1) against an fp64 statement of Conv2d + bias (+ residual) + PReLU (+ affine) of resnet.py:55-69; (2) against the ring
    kernel on the same launch at its split-sum tolerance; (3) repeatable bit for bit (parts are added in part order whoever
    finishes); split and fp32 outputs."""
    from deeplip_amd import _lib
    import ctypes as C
    from deeplip_amd.ops import ConvDesc
    x, ws, b, res, kw, ref = _rows2d_inputs(case)
    N, H, W, Cin, K, R, stride, pad, dil, use_res, post = case
    xs = ops.split_pack(x.cuda())
    rs = ops.split_pack(res.cuda()) if res is not None else None
    force_rows2d(1)
    d = ConvDesc(N, H, W, Cin, K, R, R, stride, stride, pad, pad, dil, dil, ref.shape[1], ref.shape[2], Cin, K, K if use_res else 0)
    assert _lib.lib().dlip_conv_kernel_kind(C.byref(d)) == 2
    outs = {}
    for which in ("ring", "rows", "rows-again"):
        force_rows2d(0 if which == "ring" else 1)
        for out_split in (True, False):
            if which != "ring" and use_res and post:
                continue
            y = ops.conv_nhwc(xs, ws.cuda(), b.cuda(), residual=rs, out_split=out_split, **kw)
            outs[which, out_split] = y.clone()
    torch.cuda.synchronize()
    for (which, out_split), y in outs.items():
        yv = ops.split_unpack(y) if out_split else y
        assert yv.shape == ref.shape
        assert rel_err(yv.cpu().numpy(), ref.numpy()) < TOL, (which, out_split)
        assert np.abs(yv.cpu().numpy() - ref.numpy()).max() < 2e-5 * np.abs(ref.numpy()).max(), (which, out_split)
    for out_split in (True, False):
        assert torch.equal(outs["rows", out_split].view(torch.int32), outs["rows-again", out_split].view(torch.int32)), out_split
        a = ops.split_unpack(outs["rows", out_split]) if out_split else outs["rows", out_split]
        r = ops.split_unpack(outs["ring", out_split]) if out_split else outs["ring", out_split]
        assert rel_err(a.cpu().numpy(), r.cpu().numpy()) < 2e-6, out_split


@pytest.mark.parametrize("case", [(40, 6, 256, 512, True), (12, 12, 128, 256, True), (9, 6, 256, 512, False)], ids=str)
def test_conv_rows_general_mode_second_source(ops, case, force_rows2d):
    """dlip_conv2_nhwc_f16x3 (conv3x3 + the 1x1 stride-2 shortcut in one reduction) on the rows kernel's DUAL instances, forced:
    against fp64 and against the ring kernel's DUAL instances."""
    from deeplip_amd import packing
    N, Hin, C2, K, out_split = case
    Ho = (Hin - 1) // 2 + 1
    x = _split_ref_value(rnd(N, Hin, Hin, C2, seed=51) * 2.0)
    h = _split_ref_value(rnd(N, Ho, Ho, K, seed=52) * 2.0)
    w2 = rnd(K, K, 3, 3, seed=53, scale=1.0 / np.sqrt(9 * K))
    wd = rnd(K, C2, 1, 1, seed=54, scale=1.0 / np.sqrt(C2))
    b = rnd(K, seed=55, scale=0.1)
    slope = (torch.rand(K, generator=torch.Generator().manual_seed(5)) * 0.3)
    ref = F.conv2d(h.permute(0, 3, 1, 2).double(), w2.double(), None, padding=1) + \
        F.conv2d(x.permute(0, 3, 1, 2).double(), wd.double(), None, stride=2) + b.double().view(1, K, 1, 1)
    ref = torch.where(ref >= 0, ref, ref * slope.double().view(1, K, 1, 1)).permute(0, 2, 3, 1)
    rows = torch.cat([w2.double().permute(0, 2, 3, 1).reshape(K, -1), wd.double().reshape(K, C2)], dim=1)
    ws, sc = packing.split_weights(rows)
    outs = {}
    for which in ("ring", "rows", "rows-again"):
        force_rows2d(0 if which == "ring" else 1)
        y = ops.conv2_nhwc(ops.split_pack(h.cuda()), ops.split_pack(x.cuda()), ws.cuda(), b.cuda(), sc.cuda(), pad=(1, 1),
                           stride2=(2, 2), slope=slope.cuda(), out_split=out_split)
        outs[which] = (ops.split_unpack(y) if out_split else y).clone()
    torch.cuda.synchronize()
    for which, y in outs.items():
        assert rel_err(y.cpu().numpy(), ref.numpy()) < TOL, which
    assert torch.equal(outs["rows"].view(torch.int32), outs["rows-again"].view(torch.int32))
    assert rel_err(outs["rows"].cpu().numpy(), outs["ring"].cpu().numpy()) < 2e-6

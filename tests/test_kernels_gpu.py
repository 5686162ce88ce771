"""Kernel-level parity (-m gpu): every C-ABI entry point against a plain torch-CPU fp32 (or fp64)
statement of the same op on seeded inputs.  Tolerance: 2e-5 relative-to-max for the MFMA
contractions (exact fp32 fma chains, only the summation order differs from ATen's), 1e-6 for the
reductions, bit-exact for integer outputs.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu

TOL = 2e-5


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "gpu tests need a ROCm device"
    from deeplip_amd import ops as _ops
    return _ops


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return torch.randn(*shape, generator=g) * scale


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


CONV_CASES = [
    # N, H, W, C, K, R, S, stride, pad, dil, residual, slope
    (3, 22, 22, 64, 64, 3, 3, 1, 1, 1, True, True),      # layer1 shape
    (3, 22, 22, 64, 128, 3, 3, 2, 1, 1, False, True),    # layer2.0.conv1
    (3, 22, 22, 64, 128, 1, 1, 2, 0, 1, False, False),   # layer2.0.downsample
    (5, 11, 11, 128, 128, 3, 3, 1, 1, 1, True, True),
    (7, 6, 6, 256, 256, 3, 3, 1, 1, 1, True, True),
    (9, 3, 3, 512, 512, 3, 3, 1, 1, 1, True, True),      # layer4: M = 81 (ragged M tile)
    (40, 3, 3, 256, 512, 3, 3, 2, 1, 1, False, True),
    (2, 1, 50, 24, 512, 1, 5, 1, 0, 1, False, True),     # tdnn.0 (C=24 < BK, conv1d as H=1)
    (2, 1, 60, 512, 512, 1, 3, 1, 0, 3, False, True),    # dilated tdnn
    (2, 1, 40, 512, 1500, 1, 1, 1, 0, 1, False, True),   # tdnn.9 (K tail 1500)
    (1, 1, 37, 3000, 512, 1, 1, 1, 0, 1, False, False),  # fc1 as GEMM (C % 32 != 0)
    (2, 1, 29, 512, 256, 1, 7, 1, 12, 4, False, True),   # TCN branch k=7, dil 4, same padding
    (260, 22, 22, 64, 64, 3, 3, 1, 1, 1, True, True),    # enough tiles for the 128x64 path
    (300, 6, 6, 256, 256, 3, 3, 1, 1, 1, True, True),    # 128x128 path (M=10800, K=256 -> 170 tiles? no: 64x64)
    (1900, 6, 6, 256, 256, 3, 3, 1, 1, 1, False, True),  # 128x128 path (535 x 2 tiles)
]


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "x".join(str(v) for v in c[:10]))
def test_conv_nhwc(ops, case):
    N, H, W, C, K, R, S, stride, pad, dil, use_res, use_slope = case
    x = rnd(N, C, H, W, seed=1)
    w = rnd(K, C, R, S, seed=2, scale=1.0 / np.sqrt(C * R * S))
    b = rnd(K, seed=3, scale=0.1)
    sh, sw = (1, stride) if H == 1 else (stride, stride)
    ph, pw = (0, pad) if H == 1 else (pad, pad)
    dh, dw = (1, dil) if H == 1 else (dil, dil)
    ref = F.conv2d(x, w, b, stride=(sh, sw), padding=(ph, pw), dilation=(dh, dw))
    res = rnd(*ref.shape, seed=4) if use_res else None
    slope = torch.rand(K, generator=torch.Generator().manual_seed(5)) * 0.3 if use_slope else None
    if res is not None:
        ref = ref + res
    if slope is not None:
        ref = F.prelu(ref, slope)
    dev = "cuda"
    y = ops.conv_nhwc(nhwc(x).to(dev), w.permute(0, 2, 3, 1).contiguous().to(dev), b.to(dev),
                      stride=(sh, sw), pad=(ph, pw), dil=(dh, dw),
                      residual=nhwc(res).to(dev) if res is not None else None,
                      slope=slope.to(dev) if slope is not None else None)
    torch.cuda.synchronize()
    assert rel_err(y.cpu().permute(0, 3, 1, 2).numpy(), ref.numpy()) < TOL


def test_conv_post_affine_and_channel_slices(ops):
    # act-first TDNN epilogue (LeakyReLU then BN affine) and concat-free output slices
    x = rnd(2, 1, 40, 64, seed=7)
    w = rnd(96, 1, 3, 64, seed=8, scale=0.1)
    b = rnd(96, seed=9, scale=0.1)
    sc = torch.rand(96) + 0.5
    sf = rnd(96, seed=10, scale=0.1)
    ref = F.leaky_relu(F.conv2d(x.permute(0, 3, 1, 2), w.permute(0, 3, 1, 2), b), 0.2) * sc[None, :, None, None] + sf[None, :, None, None]
    out = torch.zeros(2, 1, 38, 200, device="cuda")
    slope = torch.full((96,), 0.2, device="cuda")
    ops.conv_nhwc(x.cuda(), w.cuda(), b.cuda(), slope=slope, post_scale=sc.cuda(), post_shift=sf.cuda(), out=out,
                  out_channel_offset=100)
    torch.cuda.synchronize()
    o = out.cpu()
    assert rel_err(o[..., 100:196].permute(0, 3, 1, 2).numpy(), ref.numpy()) < TOL
    assert float(o[..., :100].abs().max()) == 0.0 and float(o[..., 196:].abs().max()) == 0.0
    # input channel slice
    y = ops.conv_nhwc(out, w.cuda()[:, :, :, :32].contiguous(), None, in_channels=32, in_channel_offset=100)
    ref2 = F.conv2d(o[..., 100:132].permute(0, 3, 1, 2), w[:, :, :, :32].permute(0, 3, 1, 2))
    torch.cuda.synchronize()
    assert rel_err(y.cpu().permute(0, 3, 1, 2).numpy(), ref2.numpy()) < TOL


def test_conv_rejects_bad_args(ops):
    from deeplip_amd._lib import DeepLipHipError
    x = torch.zeros(1, 4, 4, 6, device="cuda")      # C % 4 != 0
    w = torch.zeros(8, 1, 1, 6, device="cuda")
    with pytest.raises(DeepLipHipError):
        ops.conv_nhwc(x, w)
    with pytest.raises(DeepLipHipError):
        ops.conv_nhwc(torch.zeros(1, 4, 4, 8), torch.zeros(8, 1, 1, 8))  # CPU tensors: no fallback


@pytest.mark.parametrize("B,T,HW", [(2, 7, 88), (1, 3, 24), (1, 2, 96)])
def test_stem3d(ops, B, T, HW):
    x = rnd(B, 1, T, HW, HW, seed=11)
    w = rnd(64, 1, 5, 7, 7, seed=12, scale=1.0 / np.sqrt(245))
    b = rnd(64, seed=13, scale=0.1)
    slope = torch.rand(64, generator=torch.Generator().manual_seed(14)) * 0.3
    ref = F.prelu(F.conv3d(x, w, b, stride=(1, 2, 2), padding=(2, 3, 3)), slope)     # [B,64,T,Ho,Wo]
    wp = torch.zeros(248, 64)
    wp[:245] = w.reshape(64, 245).t()
    y = ops.stem3d(x[:, 0].contiguous().cuda(), wp.cuda(), b.cuda(), slope.cuda())
    torch.cuda.synchronize()
    got = y.cpu().view(B, T, HW // 2, HW // 2, 64).permute(0, 4, 1, 2, 3)
    assert rel_err(got.numpy(), ref.numpy()) < TOL


def test_maxpool_avgpool(ops):
    x = rnd(3, 64, 44, 44, seed=15)
    ref = F.max_pool2d(x, 3, 2, 1)
    y = ops.maxpool3x3s2(nhwc(x).cuda())
    torch.cuda.synchronize()
    assert np.array_equal(y.cpu().permute(0, 3, 1, 2).numpy(), ref.numpy())
    x = rnd(5, 512, 3, 3, seed=16)
    y = ops.avgpool(nhwc(x).cuda())
    torch.cuda.synchronize()
    assert rel_err(y.cpu().numpy(), F.adaptive_avg_pool2d(x, 1).flatten(1).numpy()) < 1e-6
    x = rnd(2, 8, 5, 7, seed=17)      # odd sizes
    y = ops.maxpool3x3s2(nhwc(x).cuda())
    torch.cuda.synchronize()
    assert np.array_equal(y.cpu().permute(0, 3, 1, 2).numpy(), F.max_pool2d(x, 3, 2, 1).numpy())


def test_time_and_group_mean(ops):
    x = rnd(4, 29, 512, seed=18)
    y = ops.time_mean(x.cuda())
    lens = torch.tensor([29, 20, 11, 1], dtype=torch.int32)
    ym = ops.time_mean(x.cuda(), lens.cuda())
    torch.cuda.synchronize()
    assert rel_err(y.cpu().numpy(), x.mean(1).numpy()) < 1e-6
    ref = torch.stack([x[b, :int(l)].mean(0) for b, l in enumerate(lens)])
    assert rel_err(ym.cpu().numpy(), ref.numpy()) < 1e-6
    g = rnd(7, 512, seed=19)
    ptr = torch.tensor([0, 1, 4, 7], dtype=torch.int32)
    yg = ops.group_mean(g.cuda(), ptr.cuda())
    torch.cuda.synchronize()
    ref = torch.stack([g[0:1].mean(0), g[1:4].mean(0), g[4:7].mean(0)])
    assert rel_err(yg.cpu().numpy(), ref.numpy()) < 1e-6


def test_meanstd_pool(ops):
    x = rnd(3, 1500, 278, seed=20) * 0.3 + 2.0     # large mean: catastrophic for an fp32 one-pass sum/sumsq (the kernel's sums are fp64)
    ref = torch.cat([x.double().mean(2), x.double().std(2)], 1)
    xd = x.permute(0, 2, 1).contiguous().cuda()
    y = ops.meanstd_pool(xd)
    ys = ops.meanstd_pool(xd, out_split=True)          # [3, 3008] split format, 8 zero channels of padding
    torch.cuda.synchronize()
    assert rel_err(y.cpu().numpy(), ref.numpy()) < 1e-6
    assert np.abs(y.cpu().numpy()[:, 1500:] - ref.numpy()[:, 1500:]).max() < 1e-6 * 0.3
    assert ys.shape == (3, 3008)
    assert torch.equal(ys.cpu().view(torch.int32), _split_ref(torch.cat([y.cpu(), torch.zeros(3, 8)], 1)).view(torch.int32))
    with pytest.raises(Exception):
        ops.meanstd_pool(rnd(2, 9, 6).cuda())          # C % 4 != 0


def test_layout_adapters(ops):
    x = rnd(3, 24, 301, seed=21)
    y = ops.nct_to_ntc(x.cuda())
    yp = ops.nct_to_ntc(x.cuda(), pad_to=32)
    back = ops.ntc_to_nct(y)
    torch.cuda.synchronize()
    assert np.array_equal(y.cpu().numpy(), x.permute(0, 2, 1).numpy())
    assert np.array_equal(yp.cpu().numpy()[..., :24], x.permute(0, 2, 1).numpy())
    assert float(yp.cpu()[..., 24:].abs().max()) == 0.0
    assert np.array_equal(back.cpu().numpy(), x.numpy())


def test_ingest_rgb(ops):
    """The ingest arithmetic is uncontracted mul / add / IEEE divide in the oracle's order (dlip_common.h): bit-exact."""
    from oracle import deeplip_oracle as O
    u8 = torch.randint(0, 256, (2, 3, 3, 16, 20), dtype=torch.uint8, generator=torch.Generator().manual_seed(22))
    y = ops.ingest_rgb_u8(u8.cuda())
    torch.cuda.synchronize()
    assert np.array_equal(y.cpu().numpy(), O.ingest_rgb_u8(u8.numpy()))
    full = torch.arange(256, dtype=torch.uint8).view(1, 1, 1, 16, 16).expand(1, 1, 3, 16, 16).contiguous()   # every gray level
    assert np.array_equal(ops.ingest_rgb_u8(full.cuda()).cpu().numpy(), O.ingest_rgb_u8(full.numpy()))


@pytest.mark.parametrize("shape", [(2, 5, 3, 88, 88), (2, 5, 88, 88), (1, 3, 96, 96), (1, 2, 3, 100, 97), (1, 2, 3, 91, 95), (1, 2, 99, 91),
                                   (3, 29, 3, 88, 88)],
                         ids=["rgb88", "gray88", "gray96-crop", "rgb100x97-crop", "rgb91x95-crop", "gray99x91-crop", "rgb-clip29"])
def test_stem3d_pool_u8_prepass_bit_identical(ops, shape):
    """uint8 frames straight into the stem's pre-pass (dlip_stem3d_pool_u8_f16x3): centre crop + BT.601 gray +
    (x/255 - 0.421)/0.165 while the split clip is written.  Three statements, all exact:
      * the normalised clip the pre-pass splits == the oracle's ingest_rgb_u8 / video_preprocess_u8 (bit for bit);
      * the stem + pool output == ingest kernel -> fp32 clip -> dlip_stem3d_pool_f16x3 (bit for bit);
      * gray and RGB sources, frames larger than the crop (even and odd margins; margins of 3, 7 and 11 -- 91, 95, 99 pixels --
        where CenterCrop's floor (preprocess.py:89-90) and a round-half-even of margin / 2 differ by one pixel), a whole 29-frame clip."""
    from deeplip_amd import packing
    from deeplip_amd.frontend import VideoFrontend
    from oracle import deeplip_oracle as O
    u8 = torch.randint(0, 256, shape, dtype=torch.uint8, generator=torch.Generator().manual_seed(71))
    w = rnd(64, 1, 5, 7, 7, seed=72, scale=1.0 / np.sqrt(245))
    b = rnd(64, seed=73, scale=0.1).cuda()
    slope = (torch.rand(64, generator=torch.Generator().manual_seed(74)) * 0.3).cuda()
    img, sc = packing.split_stem_weights(w.double())
    img, sc = img.cuda(), sc.cuda()
    clip = VideoFrontend(88)(u8.cuda())                                  # [B,1,T,88,88] fp32 (crop_norm kernel)
    want = np.stack([O.video_preprocess_u8(c) for c in u8.numpy()])      # oracle: dataloaders.py "val" pipeline per clip
    assert np.array_equal(clip.cpu().numpy()[:, 0], want)
    ref = ops.stem3d_pool(clip[:, 0].contiguous(), img, b, slope, sc)
    got = ops.stem3d_pool_u8(u8.cuda(), img, b, slope, sc, crop=88)
    torch.cuda.synchronize()
    assert got.shape == ref.shape
    assert torch.equal(got.cpu().view(torch.int32), ref.cpu().view(torch.int32))


def test_lds_read_beyond_allocation_returns_zero():
    """conv_win_f16x3's masked taps read LDS at addresses with bit 18 set and take the zeros gfx950 returns as the
    convolution's padding (the kernel is only selected on gfx950: dlip_device_is_gfx950).  The hardware behaviour itself,
    under the kernel's conditions: many workgroups per CU, every allocation full of a non-zero pattern."""
    from deeplip_amd import _lib
    from deeplip_amd.ops import ptr, stream_handle
    counts = torch.full((2,), -1, dtype=torch.int32, device="cuda")
    blocks = 4096
    _lib.check(_lib.lib().dlip_selftest_lds_oob(ptr(counts), blocks, stream_handle()), "dlip_selftest_lds_oob")
    torch.cuda.synchronize()
    in_range_ok, oob_nonzero = counts.cpu().tolist()
    assert in_range_ok == 256 * blocks       # the probe really reads LDS: inside the allocation the pattern comes back
    assert oob_nonzero == 0                  # beyond it: zeros, for every lane of every workgroup


def test_znorm_l2_cosine(ops):
    from oracle import deeplip_oracle as O
    a = rnd(6, 512, seed=23) * 0.2 + 0.1
    v = rnd(6, 512, seed=24) * 3.0 - 1.0
    y = ops.znorm_cat(a.cuda(), v.cuda())
    torch.cuda.synchronize()
    assert rel_err(y.cpu().numpy(), O.fuse_av(a, v).numpy()) < 1e-6
    yb = ops.znorm_cat(a.cuda(), None, biased=True)
    torch.cuda.synchronize()
    refb = np.stack([O.feature_normalize_np(r) for r in a.numpy()])
    assert rel_err(yb.cpu().numpy(), refb) < 1e-6
    n = ops.l2_normalize(a.cuda())
    torch.cuda.synchronize()
    assert rel_err(n.cpu().numpy(), F.normalize(a).numpy()) < 1e-6
    table = rnd(40, 1024, seed=25)
    r = np.random.Generator(np.random.PCG64(3))
    ia = torch.from_numpy(r.integers(0, 40, 500).astype(np.int32))
    ib = torch.from_numpy(r.integers(0, 40, 500).astype(np.int32))
    s0 = ops.pair_cosine(table.cuda(), ia.cuda(), ib.cuda(), mode=0)
    s1 = ops.pair_cosine(table.cuda(), ia.cuda(), ib.cuda(), mode=1)
    torch.cuda.synchronize()
    assert np.abs(s0.cpu().numpy() - O.cosine_trial_scores(table.numpy(), ia.numpy(), ib.numpy())).max() < 1e-6
    assert np.abs(s1.cpu().numpy() - O.torch_cosine_rowwise(table.numpy()[ia], table.numpy()[ib])).max() < 1e-6
    # score fusion 0.5/0.5 through the accumulate path
    ta, tv = rnd(40, 512, seed=26), rnd(40, 512, seed=27)
    sf = ops.pair_cosine(ta.cuda(), ia.cuda(), ib.cuda(), mode=0, weight=0.5)
    sf = ops.pair_cosine(tv.cuda(), ia.cuda(), ib.cuda(), mode=1, weight=0.5, out=sf)
    torch.cuda.synchronize()
    assert np.abs(sf.cpu().numpy() - O.score_fusion(ta.numpy(), tv.numpy(), ia.numpy(), ib.numpy())).max() < 1e-6


def test_logits_argmax_and_losses(ops):
    from oracle import deeplip_oracle as O
    e = rnd(32, 512, seed=28)
    W = rnd(57, 512, seed=29)
    lab = torch.arange(32) % 57
    loss_ref, logits_ref = O.lmcl(e, lab, W, 30.0, 0.2)
    logits, amax = ops.logits_argmax(e.cuda(), W.cuda(), cosine=True)
    loss = ops.margin_ce_loss(logits, lab.cuda(), 30.0, 0.2)
    torch.cuda.synchronize()
    assert rel_err(logits.cpu().numpy(), logits_ref.numpy()) < 1e-5
    assert np.array_equal(amax.cpu().numpy(), O.argmax_first(logits_ref).numpy())
    l1 = float(loss_ref) - 1e-5 * float(W.abs().sum())
    assert abs(float(loss.cpu()) - l1) < 1e-4 * abs(l1)
    b = rnd(57, seed=30)
    loss_ref, logits_ref = O.cross_entropy_head(e, lab, W, b)
    logits, amax = ops.logits_argmax(e.cuda(), W.cuda(), b.cuda(), cosine=False)
    loss = ops.margin_ce_loss(logits, lab.cuda())
    torch.cuda.synchronize()
    assert rel_err(logits.cpu().numpy(), logits_ref.numpy()) < 1e-5
    assert np.array_equal(amax.cpu().numpy(), O.argmax_first(logits_ref).numpy())
    assert abs(float(loss.cpu()) - float(loss_ref)) < 1e-4 * abs(float(loss_ref))
    # first-max tie rule
    e2 = torch.zeros(3, 8); W2 = torch.zeros(5, 8)
    e2[:, 0] = 1.0; W2[1, 0] = 2.0; W2[3, 0] = 2.0
    _, am = ops.logits_argmax(e2.cuda(), W2.cuda(), cosine=False)
    torch.cuda.synchronize()
    assert am.cpu().tolist() == [1, 1, 1]


def test_lowfer_cat(ops):
    from oracle import deeplip_oracle as O
    e1, e2 = rnd(4, 512, seed=31), rnd(4, 512, seed=32)
    y = ops.lowfer_cat(e1.cuda(), e2.cuda())
    torch.cuda.synchronize()
    assert rel_err(y.cpu().numpy(), O.lowfer(e1, e2).numpy()) < 1e-6


def test_trial_scoring_full_size_properties(ops):
    """BASELINE config C4 size: 20 000 trials over a 25 834-utterance table of 1024-d rows
    (database/trial_grid_v1.txt shape).  Size-independent properties + an oracle spot check."""
    from oracle import deeplip_oracle as O
    N, D, T = 25834, 1024, 20000
    g = torch.Generator().manual_seed(41)
    table = torch.randn(N, D, generator=g)
    ia = torch.randint(0, N, (T,), generator=g, dtype=torch.int32)
    ib = torch.randint(0, N, (T,), generator=g, dtype=torch.int32)
    ia[:100] = ib[:100]                                   # self trials
    dev = table.cuda()
    s_ab = ops.pair_cosine(dev, ia.cuda(), ib.cuda(), mode=0)
    s_ba = ops.pair_cosine(dev, ib.cuda(), ia.cuda(), mode=0)
    s_scaled = ops.pair_cosine(dev * 3.5, ia.cuda(), ib.cuda(), mode=0)
    torch.cuda.synchronize()
    s = s_ab.cpu().numpy()
    assert np.array_equal(s, s_ba.cpu().numpy())                       # symmetry, bit-exact
    assert np.abs(s[:100] - 1.0).max() < 1e-6                          # cos(x, x) = 1
    assert np.abs(s - s_scaled.cpu().numpy()).max() < 1e-6             # scale invariance
    assert np.abs(s).max() <= 1.0 + 1e-6
    sel = np.arange(0, T, 97)
    ref = O.cosine_trial_scores(table.numpy(), ia.numpy()[sel], ib.numpy()[sel])
    assert np.abs(s[sel] - ref).max() < 1e-6
    bad = ops.pair_cosine(dev, torch.tensor([N], dtype=torch.int32).cuda(), torch.tensor([0], dtype=torch.int32).cuda())
    assert np.isnan(float(bad.cpu()[0]))                               # out-of-table index -> NaN, not a fault


F16X3_CASES = [c for c in CONV_CASES if c[0] <= 300]


@pytest.mark.parametrize("case", F16X3_CASES, ids=lambda c: "x".join(str(v) for v in c[:10]))
def test_conv_nhwc_f16x3(ops, case):
    """Split-fp16 (3 x f16 MFMA) implicit GEMM: same op, same fp32 I/O; tolerance 2e-5 like the fp32 kernel
    (error budget: dropped lo*lo terms ~2^-22 + fp32 accumulation)."""
    from deeplip_amd import packing
    N, H, W, C, K, R, S, stride, pad, dil, use_res, use_slope = case
    x = rnd(N, C, H, W, seed=1) * 3.0          # activations up to ~12
    w = rnd(K, C, R, S, seed=2, scale=1.0 / np.sqrt(C * R * S))
    b = rnd(K, seed=3, scale=0.1)
    sh, sw = (1, stride) if H == 1 else (stride, stride)
    ph, pw = (0, pad) if H == 1 else (pad, pad)
    dh, dw = (1, dil) if H == 1 else (dil, dil)
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=(sh, sw), padding=(ph, pw), dilation=(dh, dw))
    res = rnd(*ref.shape, seed=4) if use_res else None
    slope = torch.rand(K, generator=torch.Generator().manual_seed(5)) * 0.3 if use_slope else None
    if res is not None:
        ref = ref + res.double()
    if slope is not None:
        ref = F.prelu(ref, slope.double())
    ws, sc = packing.split_weights(w.permute(0, 2, 3, 1).contiguous().double())
    y = ops.conv_nhwc(nhwc(x).cuda(), ws.cuda(), b.cuda(), stride=(sh, sw), pad=(ph, pw), dil=(dh, dw),
                      residual=nhwc(res).cuda() if res is not None else None,
                      slope=slope.cuda() if slope is not None else None, w_scale=sc.cuda())
    torch.cuda.synchronize()
    assert rel_err(y.cpu().permute(0, 3, 1, 2).numpy(), ref.numpy()) < TOL


def test_split_weights_roundtrip():
    from deeplip_amd import packing
    w = torch.randn(7, 3, 3, 40, dtype=torch.float64, generator=torch.Generator().manual_seed(7))
    w = w * torch.logspace(-4, 0, 7, dtype=torch.float64).view(7, 1, 1, 1)
    ws, sc = packing.split_weights(w)
    assert ws.shape == (7, 3, 3, 64) and ws.dtype == torch.float32
    h = ws.view(torch.float16).reshape(7, 3, 3, 2, 2, 32).double()       # [.., block, hi/lo, 32]
    rec = (h[..., 0, :] + h[..., 1, :]).reshape(7, 3, 3, 64)[..., :40] / sc.double().view(7, 1, 1, 1)
    # the error of an element is bounded relative to its output channel's largest weight (the scale's anchor)
    assert float(((rec - w).abs().amax(dim=(1, 2, 3)) / w.abs().amax(dim=(1, 2, 3))).max()) < 2.0 ** -21


@pytest.mark.parametrize("B,T,HW", [(2, 7, 88), (1, 3, 24)])
def test_stem3d_f16x3(ops, B, T, HW):
    from deeplip_amd import packing
    x = rnd(B, 1, T, HW, HW, seed=11) * 2.0
    w = rnd(64, 1, 5, 7, 7, seed=12, scale=1.0 / np.sqrt(245))
    b = rnd(64, seed=13, scale=0.1)
    slope = torch.rand(64, generator=torch.Generator().manual_seed(14)) * 0.3
    ref = F.prelu(F.conv3d(x.double(), w.double(), b.double(), stride=(1, 2, 2), padding=(2, 3, 3)), slope.double())
    img, sc = packing.split_stem_weights(w.double())
    y = ops.stem3d(x[:, 0].contiguous().cuda(), img.cuda(), b.cuda(), slope.cuda(), w_scale=sc.cuda())
    torch.cuda.synchronize()
    got = y.cpu().view(B, T, HW // 2, HW // 2, 64).permute(0, 4, 1, 2, 3)
    assert rel_err(got.numpy(), ref.numpy()) < TOL


def _split_ref(x):
    """CPU statement of the split activation format: per 32-channel block, 32 hi halves then 32 lo halves."""
    xs = x.reshape(-1, x.shape[-1] // 32, 32)
    hi = xs.half()
    lo = (xs - hi.float()).half()
    return torch.stack([hi, lo], dim=2).reshape(-1, x.shape[-1] * 2).view(torch.float32).reshape(x.shape)


def test_split_pack_unpack(ops):
    x = rnd(5, 7, 96, seed=21) * torch.logspace(-3, 2, 96)
    p = ops.split_pack(x.cuda())
    torch.cuda.synchronize()
    assert torch.equal(p.cpu().view(torch.int32), _split_ref(x).view(torch.int32))      # bit-exact layout
    u = ops.split_unpack(p).cpu()
    # hi + lo carries ~22 bits; below ~1e-4 the fp16 subnormal quantum (2^-24) bounds the error instead
    assert float(((u - x).abs() - 2.0 ** -21 * x.abs()).max()) < 2.0 ** -24
    with pytest.raises(Exception):
        ops.split_pack(rnd(4, 40).cuda())                                              # C % 32 != 0


SPLIT_FMT_CASES = [c for c in F16X3_CASES if c[3] % 32 == 0 and c[4] % 32 == 0]


@pytest.mark.parametrize("fmt", ["in", "out", "inout"])
@pytest.mark.parametrize("case", SPLIT_FMT_CASES, ids=lambda c: "x".join(str(v) for v in c[:10]))
def test_conv_nhwc_f16x3_split_formats(ops, case, fmt):
    """The same convolution with x / residual and / or y in the split activation format must give
    the fp32-format kernel's result.  Inputs exactly representable as hi + lo make the two paths see
    identical operands; what differs is the summation tree (split-format launches go to the LDS-DMA
    kernel, whose balanced work split adds partial tiles) and the (hi, lo) rounding of a split output."""
    from deeplip_amd import packing
    N, H, W, C, K, R, S, stride, pad, dil, use_res, use_slope = case
    x = _split_ref_value(rnd(N, H, W, C, seed=1) * 3.0)
    w = rnd(K, R, S, C, seed=2, scale=1.0 / np.sqrt(C * R * S))
    b = rnd(K, seed=3, scale=0.1)
    sh, sw = (1, stride) if H == 1 else (stride, stride)
    ph, pw = (0, pad) if H == 1 else (pad, pad)
    dh, dw = (1, dil) if H == 1 else (dil, dil)
    ws, sc = packing.split_weights(w.double())
    slope = (torch.rand(K, generator=torch.Generator().manual_seed(5)) * 0.3).cuda() if use_slope else None
    kw = dict(stride=(sh, sw), pad=(ph, pw), dil=(dh, dw), slope=slope, w_scale=sc.cuda())
    xd = x.cuda()
    base = ops.conv_nhwc(xd, ws.cuda(), b.cuda(), **kw)                 # shape probe + fp32-format result without residual
    res = _split_ref_value(rnd(*base.shape, seed=4)).cuda() if use_res else None
    base = ops.conv_nhwc(xd, ws.cuda(), b.cuda(), residual=res, **kw)
    x_split, out_split = fmt in ("in", "inout"), fmt in ("out", "inout")
    y = ops.conv_nhwc(ops.split_pack(xd) if x_split else xd, ws.cuda(), b.cuda(),
                      residual=(ops.split_pack(res) if x_split else res) if res is not None else None,
                      x_split=x_split, out_split=out_split, **kw)
    if out_split:
        y = ops.split_unpack(y)
    torch.cuda.synchronize()
    assert rel_err(y.cpu().numpy(), base.cpu().numpy()) < 2e-6


@pytest.fixture
def force_dma_tile():
    """Forces a tile of the LDS-DMA kernel through dlip_debug_set (the library reads no environment variable on
    the launch path) and restores the built-in choice afterwards."""
    from deeplip_amd import _lib
    yield lambda tile: _lib.debug_set(_lib.DBG_DMA_TILE, tile)
    _lib.debug_set(_lib.DBG_DMA_TILE, -1)


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize("case", SPLIT_FMT_CASES, ids=lambda c: "x".join(str(v) for v in c[:10]))
def test_conv_dma_tile_variants(ops, case, tile, force_dma_tile):
    """Every instance of the LDS-DMA kernel (tile menu entries 0..5) forced onto every split-format case: same
    result as the fp32-format kernel, whatever tile the shape would normally get."""
    from deeplip_amd import packing
    N, H, W, C, K, R, S, stride, pad, dil, use_res, use_slope = case
    x = _split_ref_value(rnd(N, H, W, C, seed=1) * 3.0)
    w = rnd(K, R, S, C, seed=2, scale=1.0 / np.sqrt(C * R * S))
    b = rnd(K, seed=3, scale=0.1)
    sh, sw = (1, stride) if H == 1 else (stride, stride)
    ph, pw = (0, pad) if H == 1 else (pad, pad)
    dh, dw = (1, dil) if H == 1 else (dil, dil)
    ws, sc = packing.split_weights(w.double())
    slope = (torch.rand(K, generator=torch.Generator().manual_seed(5)) * 0.3).cuda() if use_slope else None
    kw = dict(stride=(sh, sw), pad=(ph, pw), dil=(dh, dw), slope=slope, w_scale=sc.cuda())
    xd = x.cuda()
    base = ops.conv_nhwc(xd, ws.cuda(), b.cuda(), **kw)
    res = _split_ref_value(rnd(*base.shape, seed=4)).cuda() if use_res else None
    base = ops.conv_nhwc(xd, ws.cuda(), b.cuda(), residual=res, **kw)
    force_dma_tile(tile)
    for out_split in (True, False):
        y = ops.conv_nhwc(ops.split_pack(xd), ws.cuda(), b.cuda(), residual=ops.split_pack(res) if res is not None else None,
                          x_split=True, out_split=out_split, **kw)
        if out_split:
            y = ops.split_unpack(y)
        torch.cuda.synchronize()
        assert rel_err(y.cpu().numpy(), base.cpu().numpy()) < 2e-6, (tile, out_split)


def _split_ref_value(x):
    """x rounded to what the split format can hold (hi + lo), so both formats carry identical values."""
    hi = x.half()
    lo = (x - hi.float()).half()
    return hi.float() + lo.float()


@pytest.mark.parametrize("shape", [(1, 1, 40, 512, 512, 3), (2, 1, 300, 512, 512, 1), (9, 6, 6, 256, 256, 3), (700, 6, 6, 256, 256, 3)],
                         ids=lambda c: "x".join(str(v) for v in c))
def test_conv_dma_balanced_split_is_deterministic(ops, shape):
    """Tiles shared between workgroups are summed in part order by whoever arrives last: two launches give
    the same bits, and the ticket counters are back at zero for the next launch (three launches in a row)."""
    from deeplip_amd import packing
    N, H, W, C, K, S = shape
    x = ops.split_pack((rnd(N, H, W, C, seed=41) * 2.0).cuda())
    R = 1 if H == 1 else S
    w = rnd(K, R, S, C, seed=42, scale=1.0 / np.sqrt(C * R * S))
    ws, sc = packing.split_weights(w.double())
    b = rnd(K, seed=43, scale=0.1).cuda()
    kw = dict(pad=(0 if H == 1 else S // 2, S // 2), w_scale=sc.cuda(), x_split=True)
    ys = [ops.conv_nhwc(x, ws.cuda(), b, **kw).clone() for _ in range(3)]
    torch.cuda.synchronize()
    assert torch.equal(ys[0], ys[1]) and torch.equal(ys[0], ys[2])
    ref = F.conv2d(ops.split_unpack(x).cpu().permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), b.cpu().double(),
                   padding=kw["pad"])
    assert rel_err(ys[0].cpu().permute(0, 3, 1, 2).numpy(), ref.numpy()) < TOL


def test_conv_split_format_rejects_bad_args(ops):
    from deeplip_amd import packing
    w = rnd(64, 1, 1, 40, seed=2)
    ws, sc = packing.split_weights(w.double())
    x = rnd(1, 1, 8, 40).cuda()
    with pytest.raises(ValueError):
        ops.conv_nhwc(x, ws.cuda(), w_scale=sc.cuda(), x_split=True)            # C = 40 is not a multiple of 32
    with pytest.raises(ValueError):
        ops.conv_nhwc(rnd(1, 1, 8, 64).cuda(), rnd(64, 1, 1, 64).cuda(), out_split=True)   # fp32 kernel has no split format


def test_maxpool_split_output(ops):
    x = rnd(3, 44, 44, 64, seed=31)
    y = ops.maxpool3x3s2(x.cuda())
    ys = ops.maxpool3x3s2(x.cuda(), out_split=True)
    torch.cuda.synchronize()
    assert torch.equal(ys.cpu().view(torch.int32), _split_ref(y.cpu()).view(torch.int32))


@pytest.mark.parametrize("B,T,H,W", [(2, 7, 88, 88), (1, 3, 24, 24), (1, 2, 40, 56), (3, 1, 72, 88), (300, 1, 16, 16),
                                     (1, 6, 42, 52), (2, 2, 20, 30), (1, 1, 88, 60), (10, 29, 88, 88)])
@pytest.mark.parametrize("neg_slopes", [False, True], ids=["slopes>=0", "some-slopes<0"])
def test_stem3d_pool_f16x3(ops, B, T, H, W, neg_slopes):
    """Stem + MaxPool fused: the same bits as the split-fp16 stem kernel followed by the pooling kernel
    (same MFMA order, max is exact), in the split activation format; ragged last row tiles, widths that are
    not multiples of 8, frames walked by more and by fewer workgroups than there are CUs.  With every PReLU
    slope >= 0 the kernel applies affine + activation AFTER the pooling (they commute with max bit for bit);
    a negative slope makes it activate first.  The last shape has more frames (290) than the chip has workgroup slots: some
    workgroups walk two frames of six row tiles each (carried rows, flags and window refills across a frame boundary)."""
    from deeplip_amd import packing
    x = (rnd(B, T, H, W, seed=51) * 2.0).cuda()
    w = rnd(64, 1, 5, 7, 7, seed=52, scale=1.0 / np.sqrt(245))
    b = rnd(64, seed=53, scale=0.1).cuda()
    slope = torch.rand(64, generator=torch.Generator().manual_seed(54)) * 0.3
    if neg_slopes:
        slope[5::7] = -slope[5::7] - 0.05
    slope = slope.cuda()
    img, sc = packing.split_stem_weights(w.double())
    img, sc = img.cuda(), sc.cuda()
    ref = ops.maxpool3x3s2(ops.stem3d(x, img, b, slope, w_scale=sc), out_split=True)
    y = ops.stem3d_pool(x, img, b, slope, sc)
    torch.cuda.synchronize()
    assert y.shape == ref.shape
    assert torch.equal(y.cpu().view(torch.int32), ref.cpu().view(torch.int32))


def test_conv_split_input_odd_output_width(ops):
    """Split-format input with an output width the LDS-DMA kernel's 16-byte epilogue cannot write
    (K % 4 != 0): the launch must fall back to the register-staged kernel, not write past rows."""
    from deeplip_amd import packing
    x = rnd(2, 1, 50, 64, seed=61)
    w = rnd(30, 1, 3, 64, seed=62, scale=1.0 / np.sqrt(192))
    b = rnd(30, seed=63, scale=0.1)
    ws, sc = packing.split_weights(w.double())
    xs = ops.split_pack(x.cuda())
    y = ops.conv_nhwc(xs, ws.cuda(), b.cuda(), pad=(0, 1), w_scale=sc.cuda(), x_split=True)
    torch.cuda.synchronize()
    ref = F.conv2d(ops.split_unpack(xs).cpu().permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), b.double(), padding=(0, 1))
    assert rel_err(y.cpu().permute(0, 3, 1, 2).numpy(), ref.numpy()) < TOL


def test_conv_workspace_is_caller_owned(ops):
    """The balanced split's workspace is registered per stream by the binding (torch-owned); a stream with
    a block too small for slabs still computes the same result through plain launches."""
    from deeplip_amd import _lib, packing
    lib = _lib.lib()
    need = int(lib.dlip_conv_workspace_bytes())
    assert need > (1 << 18)
    x = ops.split_pack((rnd(700, 6, 6, 256, seed=71) * 2.0).cuda())
    w = rnd(256, 3, 3, 256, seed=72, scale=1.0 / np.sqrt(2304))
    ws, sc = packing.split_weights(w.double())
    ws, sc, b = ws.cuda(), sc.cuda(), rnd(256, seed=73, scale=0.1).cuda()
    y0 = ops.conv_nhwc(x, ws, b, pad=(1, 1), w_scale=sc, x_split=True)          # default stream: full workspace, balanced split
    assert (torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream) in _lib._workspaces
    side = torch.cuda.Stream()
    small = torch.empty((1 << 18) + 4096, dtype=torch.uint8, device="cuda")      # counters + 4 KB: no room for a slab
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        _lib.check(lib.dlip_conv_set_workspace(small.data_ptr(), small.numel(), side.cuda_stream), "dlip_conv_set_workspace")
        _lib._workspaces[(torch.cuda.current_device(), side.cuda_stream)] = small
        y1 = ops.conv_nhwc(x, ws, b, pad=(1, 1), w_scale=sc, x_split=True)
    torch.cuda.synchronize()
    assert rel_err(y1.cpu().numpy(), y0.cpu().numpy()) < 2e-6
    with pytest.raises(Exception):
        _lib.check(lib.dlip_conv_set_workspace(small.data_ptr(), 1024, side.cuda_stream), "dlip_conv_set_workspace")   # too small for the counters
    _lib.check(lib.dlip_conv_set_workspace(None, 0, side.cuda_stream), "dlip_conv_set_workspace")                      # unregister
    del _lib._workspaces[(torch.cuda.current_device(), side.cuda_stream)]


def test_all_pairs_cosine_and_store_group_mean(ops, tmp_path):
    """Dense [N, N] scoring agrees with the per-trial kernel on every pair; the npy store loads back onto the GPU
    with clip groups averaged by the group-mean kernel."""
    from deeplip_amd import scoring
    emb = rnd(37, 1024, seed=71)
    full = scoring.all_pairs_cosine(emb.cuda())
    ia = torch.arange(37, dtype=torch.int32).repeat_interleave(37).cuda()
    ib = torch.arange(37, dtype=torch.int32).repeat(37).cuda()
    per = scoring.cosine_scores(emb.cuda(), ia, ib).view(37, 37)
    torch.cuda.synchronize()
    assert float((full - per).abs().max()) < 2e-6
    ids = [f"u{i}.wav" for i in range(6)]
    scoring.EmbeddingTable(ids, emb[:6]).save_npy_tree(str(tmp_path))
    t = scoring.EmbeddingTable.load_npy_tree(str(tmp_path), ["a", "b"], device="cuda",
                                             groups={"a": ids[:4], "b": ids[4:]})
    assert rel_err(t.emb.cpu().numpy(), torch.stack([emb[:4].mean(0), emb[4:6].mean(0)]).numpy()) < 1e-6


def test_plda_scoring_kernel_and_eer(ops, tmp_path):
    """PLDA back-end on the engine: latent map (one GEMM) + LLR kernel vs the oracle's explicit-density LLR on every
    trial; target trials score higher than non-target ones (EER well below chance on separable synthetic speakers)."""
    from deeplip_amd import scoring
    from deeplip_amd.plda import PLDA, eer_plda
    from oracle import deeplip_oracle as O
    r = np.random.default_rng(7)
    K, n, D = 24, 12, 64
    centers = r.normal(size=(K, D)) * np.linspace(2.0, 0.1, D)
    X = np.concatenate([c + r.normal(size=(n, D)) * 0.6 for c in centers]).astype(np.float32)
    y = np.repeat(np.arange(K), n)
    model = PLDA.fit(X[y < 16], y[y < 16])                      # fit on 16 speakers, score the other 8
    test = X[y >= 16]; ty = y[y >= 16]
    ids = [f"spk{int(s)}/u{i}.wav" for i, s in enumerate(ty)]
    table = scoring.EmbeddingTable(ids, torch.from_numpy(test).cuda())
    pairs, labs = [], []
    for i in range(0, len(ids), 3):
        for j in range(1, len(ids), 7):
            pairs.append((ids[i], ids[j])); labs.append(int(ty[i] == ty[j]))
    ia, ib = table.trial_indices(pairs)
    s = model.score_trials(table.emb, ia, ib).cpu().numpy()
    U = model.transform_np(test)
    psi = model.psi[model.relevant]
    ref = np.array([O.plda_llr_bruteforce(U[a], U[b], psi) for a, b in zip(ia.cpu().numpy()[:60], ib.cpu().numpy()[:60])])
    assert np.abs(s[:60] - ref).max() < 2e-3 * max(1.0, np.abs(ref).max())
    p = tmp_path / "trials.txt"
    p.write_text("".join(f"{l} {a} {b}\n" for l, (a, b) in zip(labs, pairs)))
    eer, _ = eer_plda(table, str(p), model)
    assert 0.0 <= eer < 0.2




# ---- fusions of round 2: shortcut convolution as extra reduction slices, pooled epilogue, fused input adapter ----
SHORTCUT_CASES = [  # N, H (= W) of the block input, C2 = inplanes, K = planes, out_split, forced tile
    (5, 10, 64, 128, True, None), (5, 10, 64, 128, False, None), (3, 11, 128, 256, True, 0), (40, 6, 256, 512, True, 5),
    (7, 9, 64, 64, True, 1), (2, 6, 64, 128, True, 4), (1, 4, 32, 64, False, 3), (1, 6, 32, 128, True, 2),
]


@pytest.mark.parametrize("case", SHORTCUT_CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_conv2_shortcut_matches_two_convolutions(ops, case, force_dma_tile):
    """dlip_conv2_nhwc_f16x3: conv3x3(h) + conv1x1_stride2(x) + bias in ONE reduction == the two launches it replaces
    (BasicBlock with downsample, resnet.py:13-17,62-68), against an fp64 reference; every tile of the menu."""
    from deeplip_amd import packing
    N, Hin, C2, K, out_split, tile = case
    Ho = (Hin - 1) // 2 + 1
    x = _split_ref_value(rnd(N, Hin, Hin, C2, seed=51) * 2.0)            # block input (shortcut source)
    h = _split_ref_value(rnd(N, Ho, Ho, K, seed=52) * 2.0)               # conv1 output (conv2 source)
    w2 = rnd(K, K, 3, 3, seed=53, scale=1.0 / np.sqrt(9 * K))
    wd = rnd(K, C2, 1, 1, seed=54, scale=1.0 / np.sqrt(C2))
    b = rnd(K, seed=55, scale=0.1)
    slope = (torch.rand(K, generator=torch.Generator().manual_seed(5)) * 0.3)
    ref = F.conv2d(h.permute(0, 3, 1, 2).double(), w2.double(), None, padding=1) + \
        F.conv2d(x.permute(0, 3, 1, 2).double(), wd.double(), None, stride=2) + b.double().view(1, K, 1, 1)
    ref = torch.where(ref >= 0, ref, ref * slope.double().view(1, K, 1, 1)).permute(0, 2, 3, 1)
    rows = torch.cat([w2.double().permute(0, 2, 3, 1).reshape(K, -1), wd.double().reshape(K, C2)], dim=1)
    ws, sc = packing.split_weights(rows)
    if tile is not None:
        force_dma_tile(tile)
    y = ops.conv2_nhwc(ops.split_pack(h.cuda()), ops.split_pack(x.cuda()), ws.cuda(), b.cuda(), sc.cuda(), pad=(1, 1),
                       stride2=(2, 2), slope=slope.cuda(), out_split=out_split)
    if out_split:
        y = ops.split_unpack(y)
    torch.cuda.synchronize()
    assert rel_err(y.cpu().numpy(), ref.numpy()) < TOL
    assert np.abs(y.cpu().numpy() - ref.numpy()).max() < 2e-5 * np.abs(ref.numpy()).max()


POOL_CASES = [  # N, H, W, C, K, R, S, pad, dil, group rows (in output pixels), residual, post-affine
    (58, 3, 3, 512, 512, 3, 3, 1, 1, 29 * 9, True, False),      # 2 clips of 29 frames: the trunk's last convolution (128x128 tile)
    (64 * 29, 3, 3, 512, 512, 3, 3, 1, 1, 29 * 9, True, False), # bench shape: 256x128 tile, balanced split, 64 groups
    (5, 1, 278, 512, 1500, 1, 1, 0, 1, 278, False, False),     # tdnn.9: K tail (1500), groups = utterances
    (3, 1, 150, 512, 192, 1, 3, 0, 2, 146, False, True),       # dilated taps, conv -> LReLU -> BN order, ragged last tile
    (2, 1, 300, 64, 96, 1, 1, 0, 1, 128, False, False),        # group == tile rows: every tile starts a new group
]


@pytest.mark.parametrize("case", POOL_CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_conv_pool_epilogue(ops, case):
    """dlip_conv_pool_f16x3 + dlip_pool_finish_f32 == the convolution followed by the group mean (AdaptiveAvgPool +
    temporal mean, resnet.py:125-126 / train_fusion.py:348) and by mean | unbiased std (pooling.py:24-26)."""
    from deeplip_amd import packing
    N, H, W, C, K, R, S, pad, dil, group, use_res, post = case
    x = _split_ref_value(rnd(N, H, W, C, seed=61) * 2.0)
    w = rnd(K, R, S, C, seed=62, scale=1.0 / np.sqrt(C * R * S))
    b = rnd(K, seed=63, scale=0.1)
    slope = (torch.rand(K, generator=torch.Generator().manual_seed(7)) * 0.3).cuda()
    ph, pw = (0, pad) if H == 1 else (pad, pad)
    dh, dw = (1, dil) if H == 1 else (dil, dil)
    ws, sc = packing.split_weights(w.double())
    kw = dict(pad=(ph, pw), dil=(dh, dw), slope=slope)
    if post:
        kw["post_scale"] = (0.5 + torch.rand(K, generator=torch.Generator().manual_seed(8))).cuda()
        kw["post_shift"] = rnd(K, seed=9, scale=0.1).cuda()
    xs = ops.split_pack(x.cuda())
    full = ops.conv_nhwc(xs, ws.cuda(), b.cuda(), w_scale=sc.cuda(), x_split=True, **kw)          # shape probe
    res = _split_ref_value(rnd(*full.shape, seed=64)).cuda() if use_res else None
    rs = ops.split_pack(res) if res is not None else None
    full = ops.conv_nhwc(xs, ws.cuda(), b.cuda(), w_scale=sc.cuda(), x_split=True, residual=rs, **kw)
    rows = full.reshape(-1, K).double().cpu()
    M = rows.shape[0]
    G = (M + group - 1) // group
    pooled = ops.conv_pool(xs, ws.cuda(), b.cuda(), sc.cuda(), group, residual=rs, **kw)
    assert pooled.group_rows >= pooled.tile_rows and pooled.M == M
    mean = ops.pool_finish(pooled, "mean")
    ms = ops.pool_finish(pooled, "meanstd")
    mss = ops.split_unpack(ops.pool_finish(pooled, "meanstd", out_split=True))
    torch.cuda.synchronize()
    ref_mean = torch.stack([rows[g * group:(g + 1) * group].mean(0) for g in range(G)])
    ref_std = torch.stack([rows[g * group:(g + 1) * group].std(0) for g in range(G)])
    assert mean.shape == (G, K) and ms.shape == (G, 2 * K)
    assert rel_err(mean.cpu().numpy(), ref_mean.numpy()) < 1e-6
    assert rel_err(ms[:, :K].cpu().numpy(), ref_mean.numpy()) < 1e-6
    assert rel_err(ms[:, K:].cpu().numpy(), ref_std.numpy()) < 1e-6
    assert rel_err(mss[:, :2 * K].cpu().numpy(), ms.cpu().numpy()) < 1e-6
    assert float(mss[:, 2 * K:].abs().max()) == 0.0 if mss.shape[1] > 2 * K else True
    # deterministic: a second launch gives the same bits
    again = ops.pool_finish(ops.conv_pool(xs, ws.cuda(), b.cuda(), sc.cuda(), group, residual=rs, **kw), "meanstd")
    torch.cuda.synchronize()
    assert torch.equal(again, ms)


def test_conv_pool_rejects_short_groups(ops):
    from deeplip_amd import packing
    x = ops.split_pack(rnd(2, 1, 300, 64, seed=1).cuda())
    ws, sc = packing.split_weights(rnd(96, 1, 1, 64, seed=2).double())
    with pytest.raises(ValueError):
        ops.conv_pool(x, ws.cuda(), None, sc.cuda(), 100)      # a 128-row tile could hold two group boundaries


@pytest.mark.parametrize("shape", [(3, 24, 150), (2, 80, 300), (1, 33, 37)], ids=str)
def test_nct_to_ntc_split_is_transpose_then_split_pack(ops, shape):
    B, Cc, T = shape
    x = (rnd(B, Cc, T, seed=71) * 4.0).cuda()
    Cp = (Cc + 31) // 32 * 32
    two = ops.split_pack(ops.nct_to_ntc(x, pad_to=Cp))
    one = ops.nct_to_ntc(x, pad_to=Cp, out_split=True)
    torch.cuda.synchronize()
    assert torch.equal(one.view(torch.int32), two.view(torch.int32))


def test_conv_dma_balanced_split_stress_under_uneven_load(ops):
    """The slab hand-off of the balanced split (sc1 write-through slab stores -> drained -> barrier -> relaxed agent
    ticket; finisher: agent acquire -> barrier -> sc1 slab loads) over hundreds of launches with the split FORCED on
    (dlip_debug_set DLIP_DBG_STREAMK = 2), a second stream hammering HBM beside it (uneven load, consumers L1-warm from
    the previous launch): every launch must give the same bits as the first, and those must be right."""
    from deeplip_amd import _lib, packing
    try:
        _lib.debug_set(_lib.DBG_STREAMK, 2)
        side = torch.cuda.Stream()
        junk = torch.empty(64 << 20, device="cuda")
        for shape in [(700, 6, 6, 256, 256, 3), (64, 1, 296, 512, 512, 1)]:
            N, H, W, C, K, S = shape
            x = ops.split_pack((rnd(N, H, W, C, seed=91) * 2.0).cuda())
            R = 1 if H == 1 else S
            w = rnd(K, R, S, C, seed=92, scale=1.0 / np.sqrt(C * R * S))
            ws, sc = packing.split_weights(w.double())
            ws, sc = ws.cuda(), sc.cuda()
            b = rnd(K, seed=93, scale=0.1).cuda()
            kw = dict(pad=(0 if H == 1 else S // 2, S // 2), w_scale=sc, x_split=True)
            first = ops.conv_nhwc(x, ws, b, **kw).clone()
            y = torch.empty_like(first)
            bad = 0
            for i in range(150):
                if i % 3 == 0:
                    with torch.cuda.stream(side):
                        junk.add_(1.0)                 # HBM traffic on other CUs while the tiles hand over
                ops.conv_nhwc(x, ws, b, out=y, **kw)
                bad += int(not torch.equal(y, first))
            torch.cuda.synchronize()
            assert bad == 0, (shape, bad)
            ref = F.conv2d(ops.split_unpack(x).cpu().permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), b.cpu().double(),
                           padding=kw["pad"])
            assert rel_err(first.cpu().permute(0, 3, 1, 2).numpy(), ref.numpy()) < TOL
    finally:
        _lib.debug_set(_lib.DBG_STREAMK, -1)


WIN_CASES = [  # N, H, W, C, K, pad (= dil for same-size 3x3), dil, residual, slope, out_split
    (3, 22, 22, 64, 64, 1, 1, True, True, True),        # layer 1's shape
    (260, 22, 22, 64, 64, 1, 1, True, True, True),      # many tiles, M tail inside an image
    (7, 9, 13, 64, 64, 1, 1, False, True, False),       # odd width: tiles start mid-row, fp32 output
    (5, 11, 11, 128, 64, 1, 1, True, False, True),      # four channel slices: windows fetched inside the loop
    (2, 6, 6, 32, 32, 1, 1, False, True, True),         # one channel slice, K = 32 (half-empty column block)
    (4, 11, 11, 96, 48, 2, 2, True, True, False),       # dilation 2 (halo 48 = the window's slack), ragged K
    (1, 3, 3, 64, 64, 1, 1, True, True, True),          # image smaller than a tile: every border case at once
    (9, 1, 20, 64, 64, 1, 1, False, True, True),        # H = 1 with a 3x3 kernel: row taps all fall outside
    (40, 11, 11, 128, 128, 1, 1, True, True, True),     # layer 2's shape: the 128-column instance (eight waves)
    (3, 7, 9, 64, 96, 1, 1, False, True, False),        # 64 < K < 128: ragged column block, fp32 output
    (130, 6, 6, 256, 128, 1, 1, True, False, True),     # eight channel slices, many tiles on the 128-column instance
]


@pytest.mark.parametrize("case", WIN_CASES, ids=lambda c: "x".join(str(int(v)) for v in c))
def test_conv_window_kernel(ops, case):
    """conv_win_f16x3_kernel (same-size stride-1 3x3, K <= 128): against fp64, and against the ring kernel it replaces
    (dlip_debug_set DLIP_DBG_WIN = 0) -- the two differ only in where the activation fragments come from."""
    from deeplip_amd import _lib, packing
    from deeplip_amd._lib import ConvDesc
    N, H, W, C, K, pad, dil, use_res, use_slope, out_split = case
    x = _split_ref_value(rnd(N, H, W, C, seed=81) * 2.0)
    w = rnd(K, 3, 3, C, seed=82, scale=1.0 / np.sqrt(9 * C))
    b = rnd(K, seed=83, scale=0.1)
    ws, sc = packing.split_weights(w.double())
    slope = (torch.rand(K, generator=torch.Generator().manual_seed(5)) * 0.3) if use_slope else None
    res = _split_ref_value(rnd(N, H, W, K, seed=84)) if use_res else None
    d = ConvDesc(N, H, W, C, K, 3, 3, 1, 1, pad, pad, dil, dil, H, W, C, K, K if use_res else 0)
    assert _lib.lib().dlip_conv_kernel_kind(d) == 1                      # this shape IS served by the window kernel
    kw = dict(pad=(pad, pad), dil=(dil, dil), slope=slope.cuda() if use_slope else None, w_scale=sc.cuda(), x_split=True,
              out_split=out_split and K % 32 == 0, residual=ops.split_pack(res.cuda()) if (use_res and K % 32 == 0) else None)
    if use_res and K % 32:
        res = None
    xs = ops.split_pack(x.cuda())
    y = ops.conv_nhwc(xs, ws.cuda(), b.cuda(), **kw)
    try:
        _lib.debug_set(_lib.DBG_WIN, 0)
        assert _lib.lib().dlip_conv_kernel_kind(d) == 0
        y_ring = ops.conv_nhwc(xs, ws.cuda(), b.cuda(), **kw)
    finally:
        _lib.debug_set(_lib.DBG_WIN, -1)
    if kw["out_split"]:
        y, y_ring = ops.split_unpack(y), ops.split_unpack(y_ring)
    torch.cuda.synchronize()
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), b.double(), padding=pad, dilation=dil)
    if res is not None:
        ref = ref + res.permute(0, 3, 1, 2).double()
    if use_slope:
        ref = torch.where(ref >= 0, ref, ref * slope.double().view(1, K, 1, 1))
    ref = ref.permute(0, 2, 3, 1)
    assert rel_err(y.cpu().numpy(), ref.numpy()) < TOL
    assert rel_err(y.cpu().numpy(), y_ring.cpu().numpy()) < 2e-6


# ---- round 3: the weight gradient run as a convolution (filters of more than 32 taps; one-copy operands) ----
BIG_TAP_CASES = [
    # N, H, W, C, K, R, S, stride, pad, dil, tile (-1 = built-in choice)
    (3, 11, 11, 64, 64, 11, 11, 1, 1, 1, -1),     # layer-2-like: the 11x11 "filter" is an output-gradient map, output 3x3
    (2, 22, 22, 32, 64, 22, 22, 1, 1, 1, -1),     # 484 taps (layer 1)
    (4, 22, 22, 64, 128, 11, 11, 1, 1, 2, -1),    # a strided layer's weight gradient: dilation 2, output 4x4 (one spare row / column)
    (5, 6, 6, 96, 128, 6, 6, 1, 1, 1, 5),         # even filter size, 256x128 tile forced
    (5, 6, 6, 96, 128, 6, 6, 1, 1, 1, 0),         # 128x128 tile forced
    (2, 1, 40, 64, 64, 1, 36, 2, 0, 1, -1),       # 1-D: a dilated TDNN layer's weight gradient (conv stride = the layer's dilation)
]


@pytest.mark.parametrize("case", BIG_TAP_CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_conv_more_than_32_taps(ops, case, force_dma_tile):
    """dlip_conv_nhwc_f16x3 with R*S > 32 (split input, fp32 output): the ring kernel's variant that tests a tap against the image
    when the piece is issued instead of reading the 32-bit tap mask -- against F.conv2d in fp64."""
    from deeplip_amd import packing
    N, H, W, C, K, R, S, stride, pad, dil, tile = case
    x = _split_ref_value(rnd(N, H, W, C, seed=31))
    w = rnd(K, R, S, C, seed=32, scale=1.0 / np.sqrt(C * R * S))
    sh, sw = (1, stride) if H == 1 else (stride, stride)
    ph, pw = (0, pad) if H == 1 else (pad, pad)
    dh, dw = (1, dil) if H == 1 else (dil, dil)
    ws, sc = packing.split_weights(w.double())
    ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), None, (sh, sw), (ph, pw), (dh, dw))
    force_dma_tile(tile)
    y = ops.conv_nhwc(ops.split_pack(x.cuda()), ws.cuda(), None, stride=(sh, sw), pad=(ph, pw), dil=(dh, dw), w_scale=sc.cuda(), x_split=True)
    torch.cuda.synchronize()
    assert tuple(y.shape) == (N, ref.shape[2], ref.shape[3], K)
    assert rel_err(y.cpu().numpy(), ref.permute(0, 2, 3, 1).numpy()) < 2e-5
    with pytest.raises(Exception):          # more than 32 taps exist for split input / fp32 output only
        ops.conv_nhwc(x.cuda(), ws.cuda(), None, stride=(sh, sw), pad=(ph, pw), dil=(dh, dw), w_scale=sc.cuda())


@pytest.mark.parametrize("N,H,W,C", [(5, 3, 4, 64), (33, 2, 2, 40), (64, 1, 7, 32), (70, 2, 3, 128), (40, 3, 3, 256), (33, 1, 5, 192)])
def test_wgrad_chwn_operand(ops, N, H, W, C):
    """dlip_wgrad_chwn_f32: [N,H,W,C] -> [C,H,W,N32] split blocks (32 hi halves | 32 lo halves per 32 images), zero beyond N,
    optional scale -- bit for bit."""
    from deeplip_amd._lib import check, lib, ptr, stream_handle
    x = rnd(N, H, W, C, seed=41) * 3.0
    N32 = (N + 31) // 32 * 32
    xd, scale = x.cuda(), torch.tensor([4.0, 0.25]).cuda()        # (held in variables: ptr() of a temporary would dangle)
    for sc in (None, scale):
        out = torch.full((C, H, W, N32), 9.0, device="cuda")
        check(lib().dlip_wgrad_chwn_f32(ptr(xd), ptr(out), N, H, W, C, C, N32, ptr(sc) if sc is not None else None, 0, None, stream_handle()),
              "dlip_wgrad_chwn_f32")
        out_sm = torch.full((C, N32 // 32, H, W, 32), 9.0, device="cuda")
        spl = torch.full((N, H, W, C), 9.0, device="cuda") if C % 32 == 0 else None      # + the NHWC split copy from the same read
        check(lib().dlip_wgrad_chwn_f32(ptr(xd), ptr(out_sm), N, H, W, C, C, N32, ptr(sc) if sc is not None else None, 1,
                                        ptr(spl) if spl is not None else None, stream_handle()), "dlip_wgrad_chwn_f32")
        torch.cuda.synchronize()
        ref = torch.zeros(C, H, W, N32)
        ref[..., :N] = (x * (4.0 if sc is not None else 1.0)).permute(3, 1, 2, 0)
        want = _split_ref(ref).view(torch.int32)
        assert torch.equal(out.cpu().view(torch.int32), want)
        if spl is not None:
            assert torch.equal(spl.cpu().view(torch.int32), _split_ref(x * (4.0 if sc is not None else 1.0)).view(torch.int32))
        # slice-major: the same 128-byte blocks, ordered [c][slice][h][w]
        assert torch.equal(out_sm.cpu().view(torch.int32), want.view(C, H, W, N32 // 32, 32).permute(0, 3, 1, 2, 4).contiguous())


def test_stem_wgrad_chwn_operand(ops):
    """dlip_stem_wgrad_chwn_f32: out[dt][h][w][n = b T + t] = x[b, t + dt - 2, h, w] inside the clip, zero outside."""
    from deeplip_amd._lib import check, lib, ptr, stream_handle
    B, T, H, W = 3, 7, 6, 10
    x = rnd(B, T, H, W, seed=43)
    N32 = 32
    out = torch.full((5, H, W, N32), 9.0, device="cuda")
    xd = x.cuda()
    check(lib().dlip_stem_wgrad_chwn_f32(ptr(xd), ptr(out), B, T, H, W, N32, 0, stream_handle()), "dlip_stem_wgrad_chwn_f32")
    out_sm = torch.full((5, N32 // 32, H, W, 32), 9.0, device="cuda")
    check(lib().dlip_stem_wgrad_chwn_f32(ptr(xd), ptr(out_sm), B, T, H, W, N32, 1, stream_handle()), "dlip_stem_wgrad_chwn_f32")
    torch.cuda.synchronize()
    ref = torch.zeros(5, H, W, N32)
    for dt in range(5):
        for b in range(B):
            for t in range(T):
                if 0 <= t + dt - 2 < T:
                    ref[dt, :, :, b * T + t] = x[b, t + dt - 2]
    want = _split_ref(ref).view(torch.int32)
    assert torch.equal(out.cpu().view(torch.int32), want)
    assert torch.equal(out_sm.cpu().view(torch.int32), want.view(5, H, W, N32 // 32, 32).permute(0, 3, 1, 2, 4).contiguous())


@pytest.mark.parametrize("slice_major", [True, False], ids=["slice-major", "pixel-major"])
@pytest.mark.parametrize("N,H,C,K,R,stride,pad,dil", [(40, 22, 64, 64, 3, 1, 1, 1), (33, 22, 64, 128, 3, 2, 1, 1), (70, 11, 128, 128, 3, 1, 1, 1),
                                                       (9, 11, 32, 64, 3, 2, 1, 1), (5, 6, 256, 256, 3, 1, 1, 1), (4, 12, 32, 32, 3, 1, 2, 2),
                                                       # layer 4: 3x3 maps -> a "filter" of 9 taps (<= 32: not the many-tap walk) over MORE than one
                                                       # 32-image slice -- the case round 3's slice-major default got wrong
                                                       (70, 3, 512, 512, 3, 1, 1, 1), (40, 6, 256, 512, 3, 2, 1, 1), (33, 4, 64, 96, 3, 1, 1, 1)])
def test_wgrad_as_conv_both_layouts(ops, slice_major, N, H, C, K, R, stride, pad, dil):
    """The weight gradient run as a convolution (dlip_wgrad_chwn_f32 + dlip_wgrad_conv_f16x3 on slice-major images, or
    dlip_conv_nhwc_f16x3 on pixel-major ones) against torch autograd in fp64."""
    from deeplip_amd import autograd_video as av
    x = rnd(N, C, H, H, seed=51).double().requires_grad_(False)
    w = (rnd(K, C, R, R, seed=52) * 0.05).double().requires_grad_()
    y = torch.nn.functional.conv2d(x, w, None, stride, pad, dil)
    dy = rnd(*y.shape, seed=53) * 1e-3
    y.backward(dy.double())
    old = av.WGRAD_SLICE_MAJOR
    av.WGRAD_SLICE_MAJOR = slice_major
    try:
        got = av.wgrad_as_conv(x.float().permute(0, 2, 3, 1).contiguous().cuda(), dy.permute(0, 2, 3, 1).contiguous().cuda(), R, R,
                               (stride, stride), (pad, pad), (dil, dil))
        torch.cuda.synchronize()
    finally:
        av.WGRAD_SLICE_MAJOR = old
    assert tuple(got.shape) == (K, C, R, R)
    assert rel_err(got.cpu().numpy(), w.grad.numpy()) < 2e-5


@pytest.mark.parametrize("shape", [(1856, 6, 6, 256, 256, 3, 1, True), (1856, 11, 11, 128, 256, 3, 2, False), (300, 6, 6, 256, 384, 3, 1, False),
                                   (40, 11, 11, 128, 256, 3, 1, True)],
                         ids=["layer3-res", "layer3.0.conv1-stride2", "three-lanes-forced", "small-128-row-tile"])
def test_conv_dma_paired_column_blocks(ops, shape):
    """Round 5: the column blocks of a launch as LANES of the balanced split (a.n_inner == 2): workgroup g works on column block
    g % P, the P workgroups of a row range run side by side on one XCD.  Layer 3's launches (two 128-column blocks of the 256-row
    tile) take it by themselves; dlip_debug_set(5, 2) forces it elsewhere (three lanes, a 128-row tile).  Against Conv2d in fp64,
    against the column-block-INNER order (mode 1: other part boundaries, so fp32 sums in another order: 1e-6), and bit-repeatable
    under a second stream's load (parts are added in part order whoever finishes)."""
    from deeplip_amd import _lib, packing
    N, H, W, C, K, S, stride, use_res = shape
    R = 1 if H == 1 else S
    pad = (0 if H == 1 else S // 2, S // 2)
    x = ops.split_pack((rnd(N, H, W, C, seed=41) * 2.0).cuda())
    w = rnd(K, R, S, C, seed=42, scale=1.0 / np.sqrt(C * R * S))
    ws, sc = packing.split_weights(w.double())
    ws, sc = ws.cuda(), sc.cuda()
    b = rnd(K, seed=43, scale=0.1).cuda()
    kw = dict(stride=(stride, stride) if H > 1 else (1, 1), pad=pad, w_scale=sc, x_split=True, out_split=True)
    probe = ops.conv_nhwc(x, ws, b, **kw)
    res = ops.split_pack((rnd(*probe.shape, seed=44)).cuda()) if use_res else None
    outs = {}
    side = torch.cuda.Stream()
    junk = torch.empty(32 << 20, device="cuda")
    try:
        for mode in (1, 2):
            _lib.debug_set(_lib.DBG_NINNER, mode)
            first = ops.conv_nhwc(x, ws, b, residual=res, **kw).clone()
            y = torch.empty_like(first)
            bad = 0
            for i in range(20):
                if i % 3 == 0:
                    with torch.cuda.stream(side):
                        junk.add_(1.0)
                ops.conv_nhwc(x, ws, b, residual=res, out=y, **kw)
                bad += int(not torch.equal(y.view(torch.int32), first.view(torch.int32)))
            torch.cuda.synchronize()
            assert bad == 0, (mode, bad)
            outs[mode] = ops.split_unpack(first).cpu()
        _lib.debug_set(_lib.DBG_NINNER, -1)
        builtin = ops.split_unpack(ops.conv_nhwc(x, ws, b, residual=res, **kw)).cpu()
    finally:
        _lib.debug_set(_lib.DBG_NINNER, -1)
    ref = F.conv2d(ops.split_unpack(x).cpu().permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), b.cpu().double(),
                   stride=kw["stride"], padding=pad)
    if use_res:
        ref = ref + ops.split_unpack(res).cpu().permute(0, 3, 1, 2).double()
    for mode in (1, 2):
        assert rel_err(outs[mode].permute(0, 3, 1, 2).numpy(), ref.numpy()) < TOL, mode
    assert rel_err(outs[2].numpy(), outs[1].numpy()) < 3e-6       # other part boundaries: fp32 sums in another order
    assert rel_err(builtin.numpy(), outs[1].numpy()) < 3e-6


@pytest.mark.parametrize("N,H,C,K,stride", [(64, 22, 64, 64, 1), (40, 11, 128, 128, 1), (33, 22, 64, 128, 2), (70, 3, 512, 512, 1), (928, 6, 256, 256, 1),
                                            (35, 12, 96, 160, 2)])
def test_wgrad_reference_layout_from_the_epilogue(ops, N, H, C, K, stride):
    """Round 5: dlip_wgrad_conv_f16x3 with the layer's (R, S) stores dW as [K, C, R, S] from its epilogue -- the in-kernel finisher's
    band store, the plain launch's, and slab_reduce_kernel's -- dropping the positions r' >= R the convolution computes beside them
    (stride 2 leaves a remainder row).  Bit for bit what round 4's [C, R', S', K] result gives after its slice copy and permute."""
    from deeplip_amd import _lib, autograd_video as av
    x = rnd(N, H, H, C, seed=61).cuda()
    Ho = (H + 2 - 3) // stride + 1
    dy = (rnd(N, Ho, Ho, K, seed=62) * 1e-3).cuda()
    outs = {}
    try:
        for direct in (True, False):
            for sk in (-1, 4):                       # the reduce launch / the in-kernel finisher
                av.WGRAD_DIRECT_LAYOUT = direct
                _lib.debug_set(_lib.DBG_STREAMK, sk)
                outs[(direct, sk)] = av.wgrad_as_conv(x, dy, 3, 3, (stride, stride), (1, 1), (1, 1)).clone()
        torch.cuda.synchronize()
    finally:
        av.WGRAD_DIRECT_LAYOUT = True
        _lib.debug_set(_lib.DBG_STREAMK, -1)
    for sk in (-1, 4):
        assert tuple(outs[(True, sk)].shape) == (K, C, 3, 3)
        assert torch.equal(outs[(True, sk)].view(torch.int32), outs[(False, sk)].view(torch.int32)), sk
    assert float(outs[(True, -1)].abs().max()) > 0


@pytest.mark.parametrize("N,H,C,K,stride", [(64, 22, 64, 64, 1), (40, 11, 128, 128, 1), (33, 22, 64, 128, 2), (70, 3, 512, 512, 1)])
def test_split_reduce_launch_is_bit_identical_to_the_in_kernel_finisher(ops, N, H, C, K, stride):
    """Few tiles cut many ways (weight gradients run as convolutions: 1 - 36 tiles, 14 - 100 parts each): the parts are only
    published and slab_reduce_kernel -- a second, fully parallel launch -- adds them in part order and runs the epilogue.  Same sums
    in the same order as the in-kernel finisher (dlip_debug_set(3, 4) brings it back): the same bits; repeatable."""
    from deeplip_amd import _lib, autograd_video as av
    x = rnd(N, H, H, C, seed=61).cuda()
    Ho = (H + 2 - 3) // stride + 1
    dy = (rnd(N, Ho, Ho, K, seed=62) * 1e-3).cuda()
    outs = {}
    try:
        for mode in (4, -1, -1):
            _lib.debug_set(_lib.DBG_STREAMK, mode)
            outs.setdefault(mode, []).append(av.wgrad_as_conv(x, dy, 3, 3, (stride, stride), (1, 1), (1, 1)).clone())
        torch.cuda.synchronize()
    finally:
        _lib.debug_set(_lib.DBG_STREAMK, -1)
    assert torch.equal(outs[-1][0].view(torch.int32), outs[4][0].view(torch.int32))
    assert torch.equal(outs[-1][0].view(torch.int32), outs[-1][1].view(torch.int32))
    assert float(outs[-1][0].abs().max()) > 0


@pytest.mark.parametrize("K,C,T", [(64, 64, 9), (128, 32, 3), (96, 64, 1), (32, 128, 5), (32, 1024, 9), (1024, 32, 9)])
def test_split_weights_perm_equals_permute_then_split(ops, K, C, T):
    """dlip_split_weights_perm_f32 (reference [K,C,T] in, split image out) == permute + dlip_split_weights_rows_f32, both modes."""
    from deeplip_amd._lib import check, lib, ptr, stream_handle
    w = (rnd(K, C, T, seed=45) * 0.07).cuda()
    for mode in (0, 1):
        rows, inner = (K, C) if mode == 0 else (C, K)
        perm = w.permute(0, 2, 1).contiguous() if mode == 0 else w.flip(2).permute(1, 2, 0).contiguous()      # [rows, T, inner]
        a, sa = torch.empty((rows, T * inner), device="cuda"), torch.empty((rows,), device="cuda")
        b, sb = torch.empty_like(a), torch.empty_like(sa)
        check(lib().dlip_split_weights_rows_f32(ptr(perm), ptr(a), ptr(sa), rows, T * inner, stream_handle()), "dlip_split_weights_rows_f32")
        check(lib().dlip_split_weights_perm_f32(ptr(w), ptr(b), ptr(sb), K, C, T, mode, 0, stream_handle()), "dlip_split_weights_perm_f32")
        torch.cuda.synchronize()
        assert torch.equal(sa, sb) and torch.equal(a.view(torch.int32), b.view(torch.int32))
    # mode 0 with the channels zero-padded to the next 32 (a 24-feature first layer)
    Cs = C - 8
    w2 = w[:, :Cs].contiguous()
    perm = torch.zeros((K, T, C), device="cuda"); perm[:, :, :Cs] = w2.permute(0, 2, 1)
    a, sa = torch.empty((K, T * C), device="cuda"), torch.empty((K,), device="cuda")
    b, sb = torch.empty_like(a), torch.empty_like(sa)
    check(lib().dlip_split_weights_rows_f32(ptr(perm), ptr(a), ptr(sa), K, T * C, stream_handle()), "dlip_split_weights_rows_f32")
    check(lib().dlip_split_weights_perm_f32(ptr(w2), ptr(b), ptr(sb), K, Cs, T, 0, C, stream_handle()), "dlip_split_weights_perm_f32")
    torch.cuda.synchronize()
    assert torch.equal(sa, sb) and torch.equal(a.view(torch.int32), b.view(torch.int32))


@pytest.mark.parametrize("N,H,W,C", [(5, 44, 44, 64), (3, 9, 7, 8), (2, 1, 5, 4)])
def test_maxpool_with_recorded_taps(ops, N, H, W, C):
    """dlip_maxpool3x3s2_idx_f32 / _bwd_idx_f32 == the pooling kernel and the window-scanning backward, bit for bit (ties included)."""
    from deeplip_amd._lib import check, lib, ptr, stream_handle
    x = torch.round(rnd(N, H, W, C, seed=47) * 2.0).cuda()            # rounded: plenty of ties inside a window
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    dy = rnd(N, Ho, Wo, C, seed=48).cuda()
    y = torch.empty((N, Ho, Wo, C), device="cuda")
    idx = torch.empty((N, Ho, Wo, C // 4), device="cuda", dtype=torch.int32)
    check(lib().dlip_maxpool3x3s2_idx_f32(ptr(x), ptr(y), idx.data_ptr(), N, H, W, C, stream_handle()), "dlip_maxpool3x3s2_idx_f32")
    dx = torch.empty_like(x)
    check(lib().dlip_maxpool3x3s2_bwd_idx_f32(idx.data_ptr(), ptr(dy), ptr(dx), N, H, W, C, stream_handle()), "dlip_maxpool3x3s2_bwd_idx_f32")
    dx_ref = torch.empty_like(x)
    check(lib().dlip_maxpool3x3s2_bwd_f32(ptr(x), ptr(dy), ptr(dx_ref), N, H, W, C, stream_handle()), "dlip_maxpool3x3s2_bwd_f32")
    torch.cuda.synchronize()
    assert torch.equal(y, ops.maxpool3x3s2(x)) and torch.equal(dx, dx_ref)


def test_pow2_lift_buffer(ops):
    """dlip_pow2_lift_f32: (2^e, 2^-e) as dlip_pow2_scale_f32 forms them, then 2^-e repeated 2048 times."""
    from deeplip_amd import _lib
    from deeplip_amd._lib import check, lib, ptr, stream_handle
    for scale in (3e-7, 1.0, 5e4):
        x = (rnd(1000, 37, seed=49) * scale).cuda()
        lift = torch.empty((_lib.LIFT_WORDS,), device="cuda")
        pair = torch.empty((2,), device="cuda")
        check(lib().dlip_pow2_lift_f32(ptr(x), ptr(lift), x.numel(), 1024.0, stream_handle()), "dlip_pow2_lift_f32")
        check(lib().dlip_pow2_scale_f32(ptr(x), ptr(pair), x.numel(), 1024.0, stream_handle()), "dlip_pow2_scale_f32")
        torch.cuda.synchronize()
        assert torch.equal(lift[:2], pair) and bool((lift[2:2 + _lib.LIFT_BCAST] == pair[1]).all())
        m = float(x.abs().max()) * float(pair[0])
        assert 512.0 <= m <= 1024.0


def test_split_pack_scaled_with_padded_rows(ops):
    """dlip_split_pack_scaled_pad_f32 ([rows, 1500] -> [rows, 1504], zero columns) == pad, then dlip_split_pack_scaled_f32."""
    from deeplip_amd._lib import check, lib, ptr, stream_handle
    rows, C, Cp = 77, 1500, 1504
    x = (rnd(rows, C, seed=91) * 1e-3).cuda()
    sc = torch.tensor([2048.0, 1.0 / 2048.0]).cuda()
    xp = torch.zeros((rows, Cp), device="cuda"); xp[:, :C] = x
    a, b = torch.empty((rows, Cp), device="cuda"), torch.full((rows, Cp), 9.0, device="cuda")
    check(lib().dlip_split_pack_scaled_f32(ptr(xp), ptr(a), ptr(sc), rows, Cp, stream_handle()), "dlip_split_pack_scaled_f32")
    check(lib().dlip_split_pack_scaled_pad_f32(ptr(x), ptr(b), ptr(sc), rows, C, Cp, stream_handle()), "dlip_split_pack_scaled_pad_f32")
    torch.cuda.synchronize()
    assert torch.equal(a.view(torch.int32), b.view(torch.int32))


# ---- round 4: the rows kernel (conv_rows_f16x3.hip): 1-D valid convolutions / k = 1 GEMMs over all frames ----
ROWS_CASES = [  # B, T, C, K, S (taps), dilation, post-affine (conv -> LReLU -> BN order), what it stands for
    (3, 300, 512, 512, 1, 1, False),     # k = 1 TDNN layer, ragged last row tile
    (2, 292, 512, 512, 3, 2, False),     # dilated TDNN layer: rows shift by 2 * dil per utterance (T' = T - 4)
    (2, 150, 96, 512, 5, 1, True),       # tdnn.0 with 80 features padded to 96: 3 slices x 5 taps, post-affine order
    (2, 278, 512, 1500, 1, 1, False),    # tdnn.9's shape: K tail (1500 = 5 x 256 + 220)
    (1, 40, 64, 256, 1, 1, False),       # one partial tile, K = one column block
    (1, 700, 32, 288, 2, 3, True),       # a single channel slice, two taps, K tail of one 32-channel block
    (5, 200, 128, 512, 1, 1, False),     # several tiles per workgroup when the grid is capped (see the persistent test)
]


@pytest.fixture
def force_rows():
    from deeplip_amd import _lib
    yield lambda v: _lib.debug_set(_lib.DBG_ROWS, v)
    _lib.debug_set(_lib.DBG_ROWS, -1)
    _lib.debug_set(_lib.DBG_STREAMK, -1)


def _rows_inputs(case, seed=0):
    from deeplip_amd import packing
    B, T, C, K, S, dil, post = case
    x = _split_ref_value(rnd(B, T, C, seed=71 + seed) * 2.0)
    w = rnd(K, S, C, seed=72 + seed, scale=1.0 / np.sqrt(C * S))
    b = rnd(K, seed=73 + seed, scale=0.1)
    slope = torch.rand(K, generator=torch.Generator().manual_seed(11)) * 0.3
    ws, sc = packing.split_weights(w.double())
    kw = dict(dilation=dil, slope=slope.cuda(), w_scale=sc.cuda(), x_split=True)
    ps = pt = None
    if post:
        ps = 0.5 + torch.rand(K, generator=torch.Generator().manual_seed(12))
        pt = rnd(K, seed=13, scale=0.1)
        kw.update(post_scale=ps.cuda(), post_shift=pt.cuda())
    ref = F.conv1d(x.permute(0, 2, 1).double(), w.permute(0, 2, 1).double(), b.double(), dilation=dil)
    ref = torch.where(ref >= 0, ref, ref * slope.double().view(1, K, 1))
    if post:
        ref = ref * ps.double().view(1, K, 1) + pt.double().view(1, K, 1)
    return x, ws, b, kw, ref.permute(0, 2, 1)


@pytest.mark.parametrize("mi", [3, 4, 5])
@pytest.mark.parametrize("case", ROWS_CASES, ids=lambda c: "x".join(str(int(v)) for v in c))
def test_conv_rows_kernel(ops, case, mi, force_rows):
    """conv_rows_f16x3_kernel<MI, split | fp32 output> forced onto 1-D valid convolutions: (1) against an fp64 statement of
    Conv1d + bias + LeakyReLU (+ BatchNorm behind it) of tdnn.py:35-43; (2) BIT FOR BIT the ring kernel's plain launch of
    the same convolution -- same slice order, same product order per accumulator, same epilogue arithmetic -- for the split
    output (v_permlane16_swap pieces) and the fp32 output alike."""
    from deeplip_amd import _lib
    x, ws, b, kw, ref = _rows_inputs(case)
    xs = ops.split_pack(x.cuda())
    K = case[3]
    outs = {}
    for which in ("ring", "rows"):
        force_rows(0 if which == "ring" else mi)
        _lib.debug_set(_lib.DBG_STREAMK, 0)                      # the ring kernel as a plain launch: one workgroup per tile
        for out_split in ((True, False) if K % 32 == 0 else (False,)):
            y = ops.conv1d_ntc(xs, ws.cuda(), b.cuda(), out_split=out_split, **kw)
            outs[which, out_split] = y.clone()
    torch.cuda.synchronize()
    for (which, out_split), y in outs.items():
        yv = ops.split_unpack(y) if out_split else y
        assert yv.shape == ref.shape
        assert rel_err(yv.cpu().numpy(), ref.numpy()) < TOL, (which, out_split)
    for out_split in ((True, False) if K % 32 == 0 else (False,)):
        assert torch.equal(outs["rows", out_split].view(torch.int32), outs["ring", out_split].view(torch.int32)), out_split


def test_conv_rows_kernel_is_chosen_for_the_speech_encoder_shapes(ops):
    """dlip_conv_kernel_kind / dlip_conv_plan: at the bench's batch the TDNN layers go to the rows kernel with a one-round tile
    (B = 64, T' = 296: 119 row tiles of 160 x 2 column blocks on 256 CUs), a fully connected layer on a batch does not."""
    import ctypes as C
    from deeplip_amd import _lib
    from deeplip_amd.ops import ConvDesc

    def kind(N, W, Cin, K, S, dil, ldr=0):
        Wo = W - dil * (S - 1)
        d = ConvDesc(N, 1, W, Cin, K, 1, S, 1, 1, 0, 0, 1, dil, 1, Wo, Cin, K, ldr)
        bm, bn = C.c_int32(), C.c_int32()
        _lib.check(_lib.lib().dlip_conv_plan(C.byref(d), 3, C.byref(bm), C.byref(bn)), "dlip_conv_plan")
        return _lib.lib().dlip_conv_kernel_kind(C.byref(d)), bm.value, bn.value

    assert kind(64, 296, 512, 512, 1, 1) == (2, 160, 256)
    assert kind(64, 300, 512, 512, 3, 2)[0] == 2
    assert kind(256, 278, 512, 1500, 1, 1)[0] == 2
    assert kind(64, 1, 3008, 512, 1, 1)[0] == 0            # fc1 on 64 utterances: M = 64
    assert kind(64, 296, 512, 512, 1, 1, ldr=512)[0] == 0  # a residual: the ring kernel


def test_conv_rows_persistent_stream_across_tiles(ops, force_rows):
    """More tiles than CUs: every workgroup walks several tiles with one continuous slice stream (a tile's last slices have
    the next tile's first ones in flight; the epilogue's stores are counted in the vmcnt waits behind it).  B x T' rows far
    beyond 256 tiles of 96 rows; bit-identical to the ring kernel, repeatable."""
    from deeplip_amd import _lib
    case = (40, 300, 128, 512, 3, 1, False)                  # M = 11 920 -> 125 x 2 = 250 tiles at MI = 3 ... use more rows:
    case = (130, 300, 128, 512, 3, 1, False)                 # M = 38 740 -> 404 x 2 = 808 tiles of 96 x 256: 3.2 per CU
    x, ws, b, kw, ref = _rows_inputs(case, seed=5)
    xs = ops.split_pack(x.cuda())
    force_rows(0)
    _lib.debug_set(_lib.DBG_STREAMK, 0)
    want = ops.conv1d_ntc(xs, ws.cuda(), b.cuda(), out_split=True, **kw).clone()
    force_rows(3)
    got = [ops.conv1d_ntc(xs, ws.cuda(), b.cuda(), out_split=True, **kw).clone() for _ in range(2)]
    torch.cuda.synchronize()
    assert torch.equal(got[0].view(torch.int32), want.view(torch.int32)) and torch.equal(got[0], got[1])
    assert rel_err(ops.split_unpack(got[0]).cpu().numpy(), ref.numpy()) < TOL


@pytest.mark.parametrize("case,mi", [((100, 300, 64, 512, 1, 1, False), 5), ((70, 300, 64, 512, 3, 2, False), 4), ((120, 290, 96, 256, 2, 1, True), 3),
                                     ((256, 300, 32, 512, 1, 1, False), 5)], ids=lambda c: "x".join(str(int(v)) for v in c) if isinstance(c, tuple) else str(c))
def test_conv_rows_short_last_round(ops, case, mi, force_rows):
    """Round 5: the rows left over after a launch's full rounds of tiles run as ONE round of shorter tiles spread over all CUs (a short
    tile keeps a full tile's LDS image; the blocks it does not have are neither fetched nor multiplied nor stored).  Bit for bit the
    launch with every tile at full height (dlip_debug_set(9, 0)) and the ring kernel's plain launch, split and fp32 output; the
    statistics epilogue's chunks (two per row tile of either height) add up to the same column sums; repeatable."""
    from deeplip_amd import _lib
    B, T, C, K, S, dil, post = case
    x, ws, b, kw, ref = _rows_inputs(case, seed=9)
    xs = ops.split_pack(x.cuda())
    outs = {}
    try:
        for which in ("ring", "full", "short", "short2"):
            force_rows(0 if which == "ring" else mi)
            _lib.debug_set(_lib.DBG_STREAMK, 0)
            _lib.debug_set(_lib.DBG_ROWS_TAIL, 0 if which == "full" else -1)
            for out_split in (True, False):
                outs[which, out_split] = ops.conv1d_ntc(xs, ws.cuda(), b.cuda(), out_split=out_split, **kw).clone()
            if which in ("full", "short") and not post:
                n = ops.conv_stats_chunks(B, 1, T, C, K, 1, S, dil=(1, dil))
                st = torch.full((n * K * 2,), float("nan"), device="cuda", dtype=torch.float64)
                y = ops.conv_nhwc(xs.view(B, 1, T, C), ws.cuda().view(K, 1, S, C), b.cuda(), dil=(1, dil), w_scale=kw["w_scale"], x_split=True,
                                  slope=kw["slope"], stats=st)
                outs[which, "stats"] = (n, st.view(n, K, 2).sum(0), y.clone())
    finally:
        _lib.debug_set(_lib.DBG_ROWS_TAIL, -1)
    torch.cuda.synchronize()
    for out_split in (True, False):
        for which in ("full", "short", "short2"):
            assert torch.equal(outs[which, out_split].view(torch.int32), outs["ring", out_split].view(torch.int32)), (which, out_split)
    assert rel_err(outs["short", False].cpu().numpy(), ref.numpy()) < TOL
    if not post:
        (n0, s0, y0), (n1, s1, y1) = outs["full", "stats"], outs["short", "stats"]
        assert n1 > n0                                        # the short round really is in use: more (shorter) row tiles
        assert bool(torch.isfinite(s1).all()) and rel_err(s1.cpu().numpy(), s0.cpu().numpy()) < 2e-6
        assert torch.equal(y0, y1) and torch.equal(y1.view(-1, K), outs["ring", False].view(-1, K))


@pytest.mark.parametrize("mi", [3, 5])
@pytest.mark.parametrize("case", [c for c in POOL_CASES if c[1] == 1], ids=lambda c: "x".join(str(v) for v in c))
def test_conv_rows_pooled_epilogue(ops, case, mi, force_rows):
    """The rows kernel's pooled epilogue (per wave row and row-group segment the fp64 column sums, straight from the
    accumulators) feeds the same finishers as the ring kernel's: group mean, mean | unbiased std (pooling.py:24-26), and the
    two agree to fp32 rounding; repeatable bits."""
    from deeplip_amd import packing
    N, H, W, C, K, R, S, pad, dil, group, use_res, post = case
    x = _split_ref_value(rnd(N, H, W, C, seed=61) * 2.0)
    w = rnd(K, R, S, C, seed=62, scale=1.0 / np.sqrt(C * R * S))
    b = rnd(K, seed=63, scale=0.1)
    slope = (torch.rand(K, generator=torch.Generator().manual_seed(7)) * 0.3).cuda()
    ws, sc = packing.split_weights(w.double())
    kw = dict(pad=(0, pad), dil=(1, dil), slope=slope)
    if post:
        kw["post_scale"] = (0.5 + torch.rand(K, generator=torch.Generator().manual_seed(8))).cuda()
        kw["post_shift"] = rnd(K, seed=9, scale=0.1).cuda()
    xs = ops.split_pack(x.cuda())
    force_rows(0)
    full = ops.conv_nhwc(xs, ws.cuda(), b.cuda(), w_scale=sc.cuda(), x_split=True, **kw)
    ring = ops.conv_pool(xs, ws.cuda(), b.cuda(), sc.cuda(), group, **kw)
    ring_ms = ops.pool_finish(ring, "meanstd")
    force_rows(mi)
    if group < 16 * mi:
        with pytest.raises(ValueError):
            ops.conv_pool(xs, ws.cuda(), b.cuda(), sc.cuda(), group, **kw)
        return
    pooled = ops.conv_pool(xs, ws.cuda(), b.cuda(), sc.cuda(), group, **kw)
    assert pooled.tile_rows == 16 * mi
    ms = ops.pool_finish(pooled, "meanstd")
    mean = ops.pool_finish(pooled, "mean")
    again = ops.pool_finish(ops.conv_pool(xs, ws.cuda(), b.cuda(), sc.cuda(), group, **kw), "meanstd")
    torch.cuda.synchronize()
    rows = full.reshape(-1, K).double().cpu()
    G = (rows.shape[0] + group - 1) // group
    ref_mean = torch.stack([rows[g * group:(g + 1) * group].mean(0) for g in range(G)])
    ref_std = torch.stack([rows[g * group:(g + 1) * group].std(0) for g in range(G)])
    assert rel_err(mean.cpu().numpy(), ref_mean.numpy()) < 1e-6
    assert rel_err(ms[:, :K].cpu().numpy(), ref_mean.numpy()) < 1e-6 and rel_err(ms[:, K:].cpu().numpy(), ref_std.numpy()) < 1e-6
    assert rel_err(ms.cpu().numpy(), ring_ms.cpu().numpy()) < 1e-6      # (the ring launch may split its reduction: another summation order)
    assert torch.equal(again, ms)


# ---- round 4: the rows kernel's GENERAL mode (conv_rows_f16x3_kernel<5, EPI, 1, DUAL>): 2-D filters, residual, balanced split ----
def test_conv_rows_general_mode_is_never_chosen_unforced(ops):
    """The general mode measured slower than the ring kernel on every trunk layer (conv_rows_f16x3.hip): dlip_conv_kernel_kind /
    dlip_conv_plan report the ring kernel's 256 x 128 tile for layers 3 and 4 at the bench's batch, and the rows kernel's
    160 x 256 only while dlip_debug_set(7, 1) forces it."""
    import ctypes as C
    from deeplip_amd import _lib
    from deeplip_amd.ops import ConvDesc

    def kind(N, HW, Cin, K, stride=1, ldr=0):
        Ho = (HW + 2 - 3) // stride + 1
        d = ConvDesc(N, HW, HW, Cin, K, 3, 3, stride, stride, 1, 1, 1, 1, Ho, Ho, Cin, K, ldr)
        bm, bn = C.c_int32(), C.c_int32()
        _lib.check(_lib.lib().dlip_conv_plan(C.byref(d), 3, C.byref(bm), C.byref(bn)), "dlip_conv_plan")
        return _lib.lib().dlip_conv_kernel_kind(C.byref(d)), bm.value, bn.value

    B = 64 * 29
    assert kind(B, 6, 256, 256, ldr=256) == (0, 256, 128)
    assert kind(B, 3, 512, 512, ldr=512) == (0, 256, 128)
    assert kind(B, 11, 128, 128)[0] == 1                         # layer 2's same-size convolutions: the window kernel
    if _lib.lib().dlip_debug_set(_lib.DBG_ROWS2D, 1) != 0:       # the product library: the mode is not even compiled in
        _lib.debug_set(_lib.DBG_ROWS2D, 0)                       # (switching it OFF is always accepted)
        assert kind(B, 6, 256, 256, ldr=256) == (0, 256, 128)
        return
    try:                                                         # the lab library (DLIP_LIB_PATH): forcing routes layers 3 / 4 there
        assert kind(B, 6, 256, 256, ldr=256) == (2, 160, 256)
        assert kind(B, 3, 512, 512, ldr=512) == (2, 160, 256)
        assert kind(B, 11, 128, 128)[0] == 1                     # (the window kernel is asked first)
    finally:
        _lib.debug_set(_lib.DBG_ROWS2D, -1)


@pytest.mark.parametrize("N,H,W,C,R,S,stride,pad,dil", [(3, 5, 6, 64, 3, 3, 1, 1, 1), (2, 1, 40, 128, 1, 3, 1, 0, 2), (2, 4, 4, 192, 3, 3, 2, 1, 1),
                                                         (5, 3, 3, 40, 3, 3, 1, 1, 1), (1, 1, 70, 256, 1, 5, 1, 0, 1)])
def test_wgrad_reduction_major_operand(ops, N, H, W, C, R, S, stride, pad, dil):
    """dlip_wgrad_operand_f32: out[(tap C + c) J32 + j] = scale * x[n, ho s + r d - p, wo s + s' d - p, c] in 128-byte split blocks of 32
    positions (zero outside the image and for j >= J) -- bit for bit, on the 32-channel tiles (C = 40) and the wide ones (round 4:
    C % 64 == 0 / C % 128 == 0)."""
    from deeplip_amd._lib import check, lib, ptr, stream_handle
    sh = (1, stride) if H == 1 else (stride, stride)
    ph = (0, pad) if H == 1 else (pad, pad)
    dh = (1, dil) if H == 1 else (dil, dil)
    Ho = (H + 2 * ph[0] - dh[0] * (R - 1) - 1) // sh[0] + 1
    Wo = (W + 2 * ph[1] - dh[1] * (S - 1) - 1) // sh[1] + 1
    J = N * Ho * Wo
    J32 = (J + 31) // 32 * 32
    x = rnd(N, H, W, C, seed=91) * 3.0
    xd, scale = x.cuda(), torch.tensor([0.5, 2.0]).cuda()
    for sc in (None, scale):
        out = torch.full((R * S * C, J32), 9.0, device="cuda")
        check(lib().dlip_wgrad_operand_f32(ptr(xd), ptr(out), J32, N, H, W, C, C, Ho, Wo, sh[0], sh[1], R, S, dh[0], dh[1], ph[0], ph[1],
                                           ptr(sc) if sc is not None else None, stream_handle()), "dlip_wgrad_operand_f32")
        torch.cuda.synchronize()
        ref = torch.zeros(R * S, C, J32)
        xs = x * (0.5 if sc is not None else 1.0)
        xp = torch.zeros(N, H + 2 * ph[0], W + 2 * ph[1], C)
        xp[:, ph[0]:ph[0] + H, ph[1]:ph[1] + W] = xs
        for r in range(R):
            for s_ in range(S):
                win = xp[:, r * dh[0]: r * dh[0] + (Ho - 1) * sh[0] + 1: sh[0], s_ * dh[1]: s_ * dh[1] + (Wo - 1) * sh[1] + 1: sh[1]]   # [N,Ho,Wo,C]
                ref[r * S + s_, :, :J] = win.reshape(J, C).t()
        assert torch.equal(out.cpu().view(torch.int32), _split_ref(ref.view(R * S * C, J32)).view(torch.int32))


@pytest.mark.parametrize("case", [(64, 296, 512, 512, 1, 1), (64, 300, 512, 512, 3, 2), (256, 120, 64, 1500, 1, 1), (64, 300, 96, 512, 5, 1)],
                         ids=lambda c: "x".join(str(v) for v in c))
def test_conv_stats_epilogue_gives_the_batchnorm_statistics(ops, case):
    """Round 5: dlip_conv_nhwc_stats_f16x3 -- the rows kernel's fp32 epilogue also leaves, per half tile, the column sums {sum y, sum y^2}
    of what it writes (the train-mode BatchNorm behind a TDNN convolution then makes no statistics pass over y).  (1) y is bit for bit
    the plain launch's; (2) the chunks add up to the fp64 column sums of y (fp32 within a chunk of <= 80 rows: 2e-6); (3) rows past M
    and channels past K contribute nothing; (4) repeatable bits; (5) dlip_bn_rows_train_fwd_f32 fed with them == its own pass."""
    from deeplip_amd import autograd as ag, packing
    from deeplip_amd._lib import check, lib, ptr, stream_handle
    B, T, Cc, K, S, dil = case
    x = ops.split_pack((rnd(B, 1, T, Cc, seed=31) * 2.0).cuda())
    w = rnd(K, 1, S, Cc, seed=32, scale=1.0 / np.sqrt(Cc * S))
    ws, sc = packing.split_weights(w.double())
    ws, sc = ws.cuda(), sc.cuda()
    b = (rnd(K, seed=33) * 0.5).cuda()
    kw = dict(dil=(1, dil), w_scale=sc, x_split=True)
    n = ops.conv_stats_chunks(B, 1, T, Cc, K, 1, S, dil=(1, dil))
    assert n > 0
    plain = ops.conv_nhwc(x, ws, b, **kw)
    st = torch.full((n * K * 2,), float("nan"), device="cuda", dtype=torch.float64)
    y = ops.conv_nhwc(x, ws, b, stats=st, **kw)
    st2 = torch.empty_like(st)
    ops.conv_nhwc(x, ws, b, stats=st2, **kw)
    torch.cuda.synchronize()
    assert torch.equal(y, plain) and torch.equal(st, st2) and bool(torch.isfinite(st).all())
    rows = y.reshape(-1, K).double().cpu()
    tot = st.view(n, K, 2).sum(0).cpu()
    assert rel_err(tot[:, 0].numpy(), rows.sum(0).numpy()) < 2e-6
    assert rel_err(tot[:, 1].numpy(), (rows * rows).sum(0).numpy()) < 2e-6
    if K % 4 == 0:
        M = rows.shape[0]
        g, be = (1.0 + 0.1 * rnd(K, seed=34)).cuda(), (0.1 * rnd(K, seed=35)).cuda()
        a0 = ag._bn_rows_fwd(y.view(M, K), g, be, None, None, 0.1, 1e-5, 0.2, False)
        a1 = ag._bn_rows_fwd(y.view(M, K), g, be, None, None, 0.1, 1e-5, 0.2, False, ready=(st, n))
        torch.cuda.synchronize()
        for u, v in zip(a0, a1):
            assert rel_err(v.cpu().numpy(), u.cpu().numpy()) < 2e-6

"""SURVEY §8(f) rank 2 (-m gpu): the lip-clip encoder under model.train() -- stem, ResNet trunk and MS-TCN head
differentiable through dlip_* forward AND backward launches (deeplip_amd/autograd_video.py).  Kernel-level
gradient checks against torch-CPU autograd (fp64) of the same op, then one Adam step of the full Lipreading
model against values captured from the reference class (tests/golden/capture_golden.py: video_train).
Tolerance 1e-4 relative on outputs and on kernel-level gradients; argmax bit-exact; full-model gradients are held
to the fp64 gradients of the reference class (1e-4, or twice the reference's own fp32 error where that is larger)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err
from deeplip_amd import weightgen as wg

pytestmark = pytest.mark.gpu
DEV = "cuda"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "video_train_golden.npz")


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return torch.randn(*shape, generator=g) * scale


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("N,H,W,C,K,R,S,stride,pad,dil,bias", [
    (3, 22, 22, 64, 64, 3, 3, 1, 1, 1, False),     # layer1 conv
    (3, 22, 22, 64, 128, 3, 3, 2, 1, 1, False),    # layer2.0.conv1 (stride 2, even input)
    (5, 11, 11, 128, 256, 3, 3, 2, 1, 1, False),   # layer3.0.conv1 (stride 2, odd input)
    (3, 22, 22, 64, 128, 1, 1, 2, 0, 1, False),    # downsample 1x1 stride 2
    (4, 3, 3, 512, 512, 3, 3, 1, 1, 1, False),     # layer4 conv
    (2, 1, 29, 512, 256, 1, 5, 1, 8, 2, True),     # TCN branch: k = 5, dilation 2, full padding (k-1)d, bias
    (2, 1, 13, 768, 768, 1, 1, 1, 0, 1, True),     # TCN 1x1 downsample with bias
])
def test_conv_train_fn_gradients(N, H, W, C, K, R, S, stride, pad, dil, bias):
    """Conv forward + d/dx, d/dW, d/db vs torch autograd (fp64)."""
    from deeplip_amd import autograd_video as av
    x = rnd(N, C, H, W, seed=1).requires_grad_()
    w = rnd(K, C, R, S, seed=2, scale=1.0 / np.sqrt(C * R * S)).requires_grad_()
    b = rnd(K, seed=3, scale=0.1).requires_grad_() if bias else None
    sh, sw = (1, stride) if H == 1 else (stride, stride)
    ph, pw = (0, pad) if H == 1 else (pad, pad)
    dh, dw = (1, dil) if H == 1 else (dil, dil)
    ref = F.conv2d(x.double(), w.double(), b.double() if bias else None, stride=(sh, sw), padding=(ph, pw), dilation=(dh, dw))
    dy = rnd(*ref.shape, seed=4)
    ref.backward(dy.double())
    xg = nhwc(x.detach()).to(DEV).requires_grad_()
    wgp = w.detach().to(DEV).requires_grad_()
    bg = b.detach().to(DEV).requires_grad_() if bias else None
    y = av.conv(xg, wgp, bg, stride=(sh, sw), pad=(ph, pw), dil=(dh, dw))
    y.backward(nhwc(dy).to(DEV))
    torch.cuda.synchronize()
    assert rel_err(y.detach().cpu().numpy(), nhwc(ref.detach()).numpy()) < 2e-5
    assert rel_err(xg.grad.cpu().numpy(), nhwc(x.grad).numpy()) < 1e-4
    assert rel_err(wgp.grad.cpu().numpy(), w.grad.numpy()) < 1e-4
    if bias:
        assert rel_err(bg.grad.cpu().numpy(), b.grad.numpy()) < 1e-4


def test_stem_conv_train_fn():
    from deeplip_amd import autograd_video as av
    B, T, H, W = 2, 6, 88, 88
    x = rnd(B, 1, T, H, W, seed=5)
    w = rnd(64, 1, 5, 7, 7, seed=6, scale=1.0 / np.sqrt(245)).requires_grad_()
    ref = F.conv3d(x.double(), w.double(), None, stride=(1, 2, 2), padding=(2, 3, 3))       # [B,64,T,44,44]
    dy = rnd(*ref.shape, seed=7)
    ref.backward(dy.double())
    wgp = w.detach().to(DEV).requires_grad_()
    y = av.stem_conv(x.view(B, T, H, W).to(DEV), wgp)                                        # [(B T),44,44,64]
    y.backward(dy.permute(0, 2, 3, 4, 1).reshape(B * T, 44, 44, 64).contiguous().to(DEV))
    torch.cuda.synchronize()
    assert rel_err(y.detach().cpu().numpy(), ref.detach().permute(0, 2, 3, 4, 1).reshape(B * T, 44, 44, 64).numpy()) < 2e-5
    assert rel_err(wgp.grad.cpu().numpy(), w.grad.numpy()) < 1e-4


@pytest.mark.parametrize("B,T,H,W", [(2, 6, 88, 88), (1, 3, 24, 40)])
def test_stem_wgrad_operand_is_the_split_transposed_im2col(B, T, H, W):
    """dlip_stem_wgrad_operand_f32 (one pass) == im2col -> transpose -> split, bit for bit: rows = taps (245 + 3 zero rows),
    per 32 positions 32 hi halves | 32 lo halves, zero columns beyond J."""
    from deeplip_amd._lib import check, lib, ptr, stream_handle
    x = rnd(B, T, H, W, seed=9).to(DEV)
    J = B * T * (H // 2) * (W // 2)
    J32 = (J + 31) // 32 * 32 + 32
    out = torch.full((248, J32), 7.0, device=DEV)
    check(lib().dlip_stem_wgrad_operand_f32(ptr(x), ptr(out), J32, B, T, H, W, stream_handle()), "dlip_stem_wgrad_operand_f32")
    col = torch.empty((J, 248), device=DEV)
    check(lib().dlip_stem_im2col_f32(ptr(x), ptr(col), B, T, H, W, stream_handle()), "dlip_stem_im2col_f32")
    torch.cuda.synchronize()
    ref = np.zeros((248, J32), dtype=np.float32)
    ref[:, :J] = col.cpu().numpy().T
    hi = ref.astype(np.float16)
    lo = (ref - hi.astype(np.float32)).astype(np.float16)
    blocks = np.concatenate([hi.reshape(248, J32 // 32, 32), lo.reshape(248, J32 // 32, 32)], axis=2)     # [248, nb, 64] halves
    got = out.cpu().numpy().view(np.float16).reshape(248, J32 // 32, 64)
    assert np.array_equal(got.view(np.uint16), blocks.view(np.uint16))


def test_split_stem_weights_on_device_equals_host_packing():
    from deeplip_amd import packing
    from deeplip_amd._lib import check, lib, ptr, stream_handle
    w = rnd(64, 1, 5, 7, 7, seed=6, scale=1.0 / np.sqrt(245))
    w[3] *= 1e-3; w[5] = 0.0
    img_h, sc_h = packing.split_stem_weights(w.double())
    img = torch.empty((64 * 296,), device=DEV)
    sc = torch.empty((64,), device=DEV)
    wd = w.to(DEV)
    check(lib().dlip_split_stem_weights_f32(ptr(wd), ptr(img), ptr(sc), 64, stream_handle()), "dlip_split_stem_weights_f32")
    torch.cuda.synchronize()
    assert np.array_equal(sc.cpu().numpy()[:5], sc_h.numpy()[:5]) and np.array_equal(sc.cpu().numpy()[6:], sc_h.numpy()[6:])
    keep = np.ones(64, bool); keep[5] = False                 # an all-zero channel: any scale, zero image
    a = img.cpu().numpy().view(np.uint16).reshape(64, 592)[:, :576]
    b = img_h.numpy().view(np.uint16).reshape(64, 592)[:, :576]
    assert np.array_equal(a[keep], b[keep]) and not a[5].any()


@pytest.mark.parametrize("M,C", [(2 * 22 * 22, 64), (1300, 256), (70, 768)])
def test_fused_batchnorm_prelu_vs_torch_autograd(M, C):
    """prelu(bn_train(x)) with per-channel slopes in the BatchNorm passes: outputs, dx, dgamma, dbeta, dslope and the running
    statistics against torch autograd (fp64) of BatchNorm1d + PReLU."""
    from deeplip_amd import autograd_video as av
    x = (rnd(M, C, seed=21) * 1.7 + 0.3).requires_grad_()
    ga = (1.0 + 0.3 * rnd(C, seed=22)).requires_grad_()
    be = (0.2 * rnd(C, seed=23)).requires_grad_()
    sl = (torch.rand(C, generator=torch.Generator().manual_seed(24)) * 0.5 - 0.05).requires_grad_()     # a few negative slopes too
    dy = rnd(M, C, seed=25)
    bn = torch.nn.BatchNorm1d(C, momentum=0.1).double()
    with torch.no_grad():
        bn.weight.copy_(ga); bn.bias.copy_(be)
    bn.train()
    xd = x.detach().double().requires_grad_()
    sld = sl.detach().double().requires_grad_()
    ref = F.prelu(bn(xd), sld)
    ref.backward(dy.double())
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    xg, gg, bg, sg = (t.detach().to(DEV).requires_grad_() for t in (x, ga, be, sl))
    y = av.BNPReLUFn.apply(xg, gg, bg, sg, rm, rv, 0.1, 1e-5)
    y.backward(dy.to(DEV))
    torch.cuda.synchronize()
    assert rel_err(y.detach().cpu().numpy(), ref.detach().numpy()) < 1e-5
    assert rel_err(xg.grad.cpu().numpy(), xd.grad.numpy()) < 1e-4
    assert rel_err(gg.grad.cpu().numpy(), bn.weight.grad.numpy()) < 1e-4
    assert rel_err(bg.grad.cpu().numpy(), bn.bias.grad.numpy()) < 1e-4
    assert rel_err(sg.grad.cpu().numpy(), sld.grad.numpy()) < 1e-4
    assert rel_err(rm.cpu().numpy(), bn.running_mean.numpy()) < 1e-5 and rel_err(rv.cpu().numpy(), bn.running_var.numpy()) < 1e-5


@pytest.mark.parametrize("M,C", [(70, 768), (1300, 256), (4096, 12), (5000, 64), (300000, 8)])
def test_batchnorm_launch_sequences_agree(M, C):
    """ABI 44: the finalize steps of the train-mode BatchNorm / column-sum entry points run in the LAST workgroup of the pass before
    them (ticket words of the stream's workspace) and tensors of <= 4096 rows take one launch per direction; dlip_debug_set(8, 0)
    restores ABI 43's separate launches.  Both sequences: same results (the fp64 column sums are associated differently: 2e-6), the
    same power-of-two lift of dx, num_batches_tracked incremented once per forward, and repeatable bits.  300 000 rows: parts longer
    than 512 rows (at most 512 parts per launch)."""
    from deeplip_amd import _lib, autograd as ag, autograd_video as av
    x = (rnd(M, C, seed=31) * 1.7 + 0.3).to(DEV)
    ga = (1.0 + 0.3 * rnd(C, seed=32)).to(DEV)
    be = (0.2 * rnd(C, seed=33)).to(DEV)
    sl = (torch.rand(C, generator=torch.Generator().manual_seed(34)) * 0.5 - 0.05).to(DEV)
    dy = (rnd(M, C, seed=35) * 1e-3).to(DEV)

    def run():
        out = {}
        for name in ("prelu", "lrelu", "act_first"):
            rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
            nbt = torch.zeros((), dtype=torch.long, device=DEV)
            xg, gg, bg, sg = (t.clone().requires_grad_() for t in (x, ga, be, sl))
            if name == "prelu":
                y = av.BNPReLUFn.apply(xg, gg, bg, sg, rm, rv, 0.1, 1e-5, nbt)
            else:
                y = ag.BNRowsActFn.apply(xg, gg, bg, rm, rv, 0.1, 1e-5, 0.2, name == "act_first", nbt)
            y.backward(dy)
            out[name] = [t.detach().clone() for t in (y, xg.grad, gg.grad, bg.grad, rm, rv, nbt.float())]
            if name == "prelu":
                out[name].append(sg.grad.clone())
        # the power-of-two lift of dx, formed by the backward's last pass (it travels on the tensor object the helper returns)
        rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
        _, mean, invstd = ag._bn_rows_fwd(x, ga, be, rm, rv, 0.1, 1e-5, 0.2, False)
        dx, _, _ = ag._bn_rows_bwd(dy, x, ga, be, mean, invstd, 0.2, False)
        out["lift"] = [dx._dlip_lift[:6].clone(), dx.abs().max().view(1)]
        out["colsum"] = [av._colsum_rows(x)]
        torch.cuda.synchronize()
        return out

    try:
        a = run()
        a2 = run()
        _lib.debug_set(_lib.DBG_BN_FUSED, 0)
        b = run()
    finally:
        _lib.debug_set(_lib.DBG_BN_FUSED, -1)
    for k in a:
        for u, v, w in zip(a[k], a2[k], b[k]):
            assert torch.equal(u, v), k                                       # repeatable bits
            assert rel_err(u.cpu().numpy(), w.cpu().numpy()) < 2e-6, k        # the two sequences
    for k in ("prelu", "lrelu", "act_first"):
        assert float(a[k][6]) == 1.0 and float(b[k][6]) == 1.0               # num_batches_tracked: one forward
    for r in (a, b):
        e, inv, amax = float(r["lift"][0][0]), float(r["lift"][0][1]), float(r["lift"][1][0])
        assert e * inv == 1.0 and 512.0 <= amax * e <= 1024.0 and bool((r["lift"][0][2:] == inv).all())
    assert torch.equal(a["lift"][0], b["lift"][0])


@pytest.mark.parametrize("N,H,W,C", [(6, 44, 44, 64), (3, 9, 11, 8), (2, 5, 7, 64), (700, 8, 8, 12)])
def test_stem_batchnorm_prelu_maxpool_as_one_function(N, H, W, C):
    """Round 5: maxpool(prelu(bn_train(x))) of the stem (model.py:83-85) without the full-resolution tensors between the three --
    (1) against torch autograd (fp64) of BatchNorm + PReLU + max_pool2d: output, dx, dgamma, dbeta, dslope, running statistics;
    (2) against the three-Function path (BNPReLUFn + MaxPoolFn): the SAME pooled output bit for bit (hence the same argmax codes)
    and gradients to 2e-6; (3) the lift of dx; repeatable bits."""
    from deeplip_amd import autograd_video as av
    x = (rnd(N, H, W, C, seed=41) * 1.7 + 0.3)
    ga = (1.0 + 0.3 * rnd(C, seed=42)); be = 0.2 * rnd(C, seed=43)
    sl = torch.rand(C, generator=torch.Generator().manual_seed(44)) * 0.5 - 0.05
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    dy = rnd(N, Ho, Wo, C, seed=45) * 1e-2
    bn = torch.nn.BatchNorm2d(C, momentum=0.1).double()
    with torch.no_grad():
        bn.weight.copy_(ga); bn.bias.copy_(be)
    bn.train()
    xd = x.double().permute(0, 3, 1, 2).contiguous().requires_grad_()
    sld = sl.double().requires_grad_()
    ref = F.max_pool2d(F.prelu(bn(xd), sld), 3, 2, 1)
    ref.backward(dy.double().permute(0, 3, 1, 2))

    def run(fused):
        rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
        nbt = torch.zeros((), dtype=torch.long, device=DEV)
        xg, gg, bg, sg = (t.clone().to(DEV).requires_grad_() for t in (x, ga, be, sl))
        if fused:
            y = av.BNPReLUMaxPoolFn.apply(xg, gg, bg, sg, rm, rv, 0.1, 1e-5, nbt)
        else:
            y = av.MaxPoolFn.apply(av.BNPReLUFn.apply(xg, gg, bg, sg, rm, rv, 0.1, 1e-5, nbt))
        y.backward(dy.to(DEV))
        torch.cuda.synchronize()
        return [t.detach().clone() for t in (y, xg.grad, gg.grad, bg.grad, sg.grad, rm, rv, nbt.float())]

    a, a2, b = run(True), run(True), run(False)
    # the forked output: two gradients arriving separately == their sum arriving once
    xg, gg, bg, sg = (t.clone().to(DEV).requires_grad_() for t in (x, ga, be, sl))
    y1, y2 = av.BNPReLUMaxPoolFn.apply(xg, gg, bg, sg, torch.zeros(C, device=DEV), torch.ones(C, device=DEV), 0.1, 1e-5, None, True)
    assert y1.data_ptr() == y2.data_ptr()
    torch.autograd.backward([y1, y2], [(0.25 * dy).to(DEV), (0.75 * dy).to(DEV)])
    torch.cuda.synchronize()
    for got, want in zip((xg.grad, gg.grad, bg.grad, sg.grad), a[1:5]):
        assert rel_err(got.cpu().numpy(), want.cpu().numpy()) < 2e-6
    assert rel_err(a[0].cpu().permute(0, 3, 1, 2).numpy(), ref.detach().numpy()) < 1e-5
    assert rel_err(a[1].cpu().permute(0, 3, 1, 2).numpy(), xd.grad.numpy()) < 1e-4
    assert rel_err(a[2].cpu().numpy(), bn.weight.grad.numpy()) < 1e-4 and rel_err(a[3].cpu().numpy(), bn.bias.grad.numpy()) < 1e-4
    assert rel_err(a[4].cpu().numpy(), sld.grad.numpy()) < 1e-4
    assert rel_err(a[5].cpu().numpy(), bn.running_mean.numpy()) < 1e-5 and rel_err(a[6].cpu().numpy(), bn.running_var.numpy()) < 1e-5
    assert float(a[7]) == 1.0
    assert torch.equal(a[0], b[0])                             # the pooled output: the same bits as apply pass + pooling
    for u, v, w in zip(a, a2, b):
        assert torch.equal(u, v) and rel_err(u.cpu().numpy(), w.cpu().numpy()) < 2e-6
    # the lift pair the backward leaves on dx
    xg = x.to(DEV).requires_grad_()
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    y = av.BNPReLUMaxPoolFn.apply(xg, ga.to(DEV), be.to(DEV), sl.to(DEV), rm, rv, 0.1, 1e-5, None)
    fn = y.grad_fn
    dx = av.BNPReLUMaxPoolFn.backward(fn, dy.to(DEV))[0]
    lift = dx._dlip_lift
    torch.cuda.synchronize()
    e, inv, amax = float(lift[0]), float(lift[1]), float(dx.abs().max())
    assert e * inv == 1.0 and 512.0 <= amax * e <= 1024.0 and bool((lift[2:8] == inv).all())


@pytest.mark.parametrize("M,C", [(2 * 22 * 22, 64), (9000, 128), (70, 512), (300000, 8)])
def test_block_tail_batchnorm_add_prelu_as_one_function(M, C):
    """Round 5: prelu(bn2(x) + residual) -- the end of a BasicBlock (resnet.py:62-69) under model.train() -- as one Function:
    (1) against torch autograd (fp64); (2) the SAME output bits as BNRowsActFn + AddPReLUFn and gradients to 2e-6; (3) the forked
    output: two gradients arriving separately == their sum arriving once; (4) repeatable bits, num_batches_tracked, the lift."""
    from deeplip_amd import autograd as ag, autograd_video as av
    x = (rnd(M, C, seed=51) * 1.7 + 0.3)
    res = rnd(M, C, seed=52)
    ga = (1.0 + 0.3 * rnd(C, seed=53)); be = 0.2 * rnd(C, seed=54)
    sl = torch.rand(C, generator=torch.Generator().manual_seed(55)) * 0.5 - 0.05
    d1, d2 = rnd(M, C, seed=56) * 1e-2, rnd(M, C, seed=57) * 1e-2
    bn = torch.nn.BatchNorm1d(C, momentum=0.1).double()
    with torch.no_grad():
        bn.weight.copy_(ga); bn.bias.copy_(be)
    bn.train()
    xd, rd, sld = x.double().requires_grad_(), res.double().requires_grad_(), sl.double().requires_grad_()
    ref = F.prelu(bn(xd) + rd, sld)
    ref.backward((d1 + d2).double())

    def run(mode):
        rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
        nbt = torch.zeros((), dtype=torch.long, device=DEV)
        xg, rg, gg, bg, sg = (t.clone().to(DEV).requires_grad_() for t in (x, res, ga, be, sl))
        if mode == "two":
            y = av.AddPReLUFn.apply(ag.BNRowsActFn.apply(xg, gg, bg, rm, rv, 0.1, 1e-5, 1.0, False, nbt), rg, sg)
            y.backward((d1 + d2).to(DEV))
        elif mode == "one":
            y = av.BNAddPReLUFn.apply(xg, rg, gg, bg, sg, rm, rv, 0.1, 1e-5, nbt, False)
            y.backward((d1 + d2).to(DEV))
        else:
            y, y2 = av.BNAddPReLUFn.apply(xg, rg, gg, bg, sg, rm, rv, 0.1, 1e-5, nbt, True)
            assert y.data_ptr() == y2.data_ptr()
            torch.autograd.backward([y, y2], [d1.to(DEV), d2.to(DEV)])
        torch.cuda.synchronize()
        return [t.detach().clone() for t in (y, xg.grad, rg.grad, gg.grad, bg.grad, sg.grad, rm, rv, nbt.float())]

    two, one, one2, fork = run("two"), run("one"), run("one"), run("fork")
    assert rel_err(one[0].cpu().numpy(), ref.detach().numpy()) < 1e-5
    for got, want in zip(one[1:6], (xd.grad, rd.grad, bn.weight.grad, bn.bias.grad, sld.grad)):
        assert rel_err(got.cpu().numpy(), want.numpy()) < 1e-4
    assert rel_err(one[6].cpu().numpy(), bn.running_mean.numpy()) < 1e-5 and rel_err(one[7].cpu().numpy(), bn.running_var.numpy()) < 1e-5
    assert float(one[8]) == 1.0 and torch.equal(one[0], two[0]) and torch.equal(one[0], fork[0])
    for a, a2, b, f in zip(one, one2, two, fork):
        assert torch.equal(a, a2)
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 2e-6 and rel_err(f.cpu().numpy(), a.cpu().numpy()) < 2e-6


@pytest.mark.parametrize("N,HW,Cin,planes,stride", [(5, 8, 64, 64, 1), (4, 8, 64, 128, 2), (40, 3, 256, 512, 2)])
def test_basic_block_with_bn1_relu1_applied_on_load(N, HW, Cin, planes, stride):
    """Round 5: conv1 -> bn1 + relu1 -> conv2 under model.train(): relu1's output is not stored, conv2's operand producer applies the
    normalisation and the slopes to conv1's raw output on load (dlip_wgrad_chwn_bn_f32).  A whole BasicBlock (with and without the
    down-sampling shortcut), on load against stored: output, input gradient, every parameter gradient and the running statistics
    bit for bit."""
    from deeplip_amd import autograd_video as av, video
    from deeplip_amd.video import BasicBlock, downsample_basic_block
    x = rnd(N, HW, HW, Cin, seed=81)
    Ho = (HW - 1) // stride + 1
    dy = rnd(N, Ho, Ho, planes, seed=82) * 1e-2

    def run(on_load):
        torch.manual_seed(3)
        ds = downsample_basic_block(Cin, planes, stride) if (stride != 1 or Cin != planes) else None
        blk = BasicBlock(Cin, planes, stride, ds, relu_type="prelu").to(DEV).train()
        prev, av.BN_ON_LOAD = av.BN_ON_LOAD, on_load
        try:
            xg = x.clone().to(DEV).requires_grad_()
            y = video._basic_block_train(blk, xg)
            y.backward(dy.to(DEV))
            torch.cuda.synchronize()
        finally:
            av.BN_ON_LOAD = prev
        return [y.detach().clone(), xg.grad.clone()] + [p.grad.clone() for p in blk.parameters()] + [b.clone().float() for b in blk.buffers()]

    a, b = run(True), run(False)
    assert len(a) == len(b) > 8
    for i, (u, v) in enumerate(zip(a, b)):
        assert torch.equal(u, v), i


def test_prelu_maxpool_avgpool_timemean_dropout():
    from deeplip_amd import autograd_video as av
    # PReLU with per-channel slope
    x = rnd(37, 11, 64, seed=8).requires_grad_()
    sl = (torch.rand(64, generator=torch.Generator().manual_seed(9)) * 0.4).requires_grad_()
    dy = rnd(37, 11, 64, seed=10)
    ref = F.prelu(x.double().permute(0, 2, 1), sl.double()).permute(0, 2, 1)
    ref.backward(dy.double())
    xg, sg = x.detach().to(DEV).requires_grad_(), sl.detach().to(DEV).requires_grad_()
    y = av.PReLUFn.apply(xg, sg)
    y.backward(dy.to(DEV))
    assert rel_err(y.detach().cpu().numpy(), ref.detach().numpy()) < 1e-6
    assert rel_err(xg.grad.cpu().numpy(), x.grad.numpy()) < 1e-6
    assert rel_err(sg.grad.cpu().numpy(), sl.grad.numpy()) < 1e-5
    # MaxPool3d (1,3,3)/(1,2,2)/(0,1,1), even and odd sizes
    for (N, H, W, C) in [(5, 44, 44, 64), (3, 9, 7, 8)]:
        x = rnd(N, C, H, W, seed=11).requires_grad_()
        ref = F.max_pool2d(x.double(), 3, 2, 1)
        dy = rnd(*ref.shape, seed=12)
        ref.backward(dy.double())
        xg = nhwc(x.detach()).to(DEV).requires_grad_()
        y = av.maxpool(xg)
        y.backward(nhwc(dy).to(DEV))
        assert torch.equal(y.detach().cpu(), nhwc(ref.detach()).float())
        assert rel_err(xg.grad.cpu().numpy(), nhwc(x.grad).numpy()) < 1e-6
    # AdaptiveAvgPool2d(1)
    x = rnd(6, 32, 3, 3, seed=13).requires_grad_()
    ref = x.double().mean((2, 3))
    dy = rnd(6, 32, seed=14)
    ref.backward(dy.double())
    xg = nhwc(x.detach()).to(DEV).requires_grad_()
    y = av.avgpool(xg)
    y.backward(dy.to(DEV))
    assert rel_err(y.detach().cpu().numpy(), ref.detach().numpy()) < 1e-6
    assert rel_err(xg.grad.cpu().numpy(), nhwc(x.grad).numpy()) < 1e-6
    # masked temporal mean (model.py:16-17)
    x = rnd(3, 9, 768, seed=15).requires_grad_()
    lens = [9, 4, 7]
    ref = torch.stack([x.double()[i, :l].mean(0) for i, l in enumerate(lens)])
    dy = rnd(3, 768, seed=16)
    ref.backward(dy.double())
    xg = x.detach().to(DEV).requires_grad_()
    y = av.time_mean(xg, torch.tensor(lens, dtype=torch.int32, device=DEV))
    y.backward(dy.to(DEV))
    assert rel_err(y.detach().cpu().numpy(), ref.detach().numpy()) < 1e-6
    assert rel_err(xg.grad.cpu().numpy(), x.grad.numpy()) < 1e-6
    # dropout: kept elements are scaled by 1/(1-p), the gradient uses the same mask
    x = torch.ones(64, 100, 32, device=DEV, requires_grad=True)
    y = av.dropout(x, 0.2)
    y.sum().backward()
    kept = float((y.detach() != 0).float().mean())
    assert 0.75 < kept < 0.85 and torch.equal(x.grad, y.detach()) and abs(float(y.detach().max()) - 1.25) < 1e-6
    torch.cuda.synchronize()


def _build(dropout):
    from models.video_models.model import Lipreading
    tcn = {"num_layers": 4, "kernel_size": [3, 5, 7], "dropout": dropout, "dwpw": False, "width_mult": 1}
    net = Lipreading(num_classes=54, relu_type="prelu", tcn_options=tcn, extract_feats=False)
    sd = wg.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, prefix="vtrain.video.")
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return net.to(DEV)


def test_lipreading_train_step_matches_reference_golden():
    """loss, logits, argmax, gradients (every parameter: norm and sum; selected tensors element-wise), BatchNorm
    running statistics after the step and the loss of the next forward vs the reference class on CPU."""
    g = np.load(GOLD)
    net = _build(0.0)
    net.train()
    x = torch.from_numpy(wg.video_input(2, frames=7, key="vtrain.x")).to(DEV)
    lab = torch.from_numpy(wg.labels(2, 54)).to(DEV)
    from deeplip_amd import autograd as ag
    opt = torch.optim.Adam(net.parameters(), lr=3e-4, weight_decay=1e-4)
    opt.zero_grad()
    logits = net(x, lengths=[7, 5])
    loss = ag.margin_ce_loss(logits, lab)          # nn.CrossEntropyLoss (train_video.py:115,143)
    loss.backward()
    torch.cuda.synchronize()
    assert abs(float(loss) - float(g["loss0"])) < 1e-4 * float(g["loss0"])
    assert rel_err(logits.detach().cpu().numpy(), g["logits0"]) < 1e-4
    assert np.array_equal(torch.max(logits, 1)[1].cpu().numpy(), g["argmax0"])
    grads = {k: v.grad for k, v in net.named_parameters()}
    sel = {"grad_stem_w": ("frontend3D.0.weight", None), "grad_stem_bn_w": ("frontend3D.1.weight", None),
           "grad_stem_prelu": ("frontend3D.2.weight", None), "grad_l1_0_conv1_w_rows4": ("trunk.layer1.0.conv1.weight", 4),
           "grad_l2_0_down_w_rows4": ("trunk.layer2.0.downsample.0.weight", 4),
           "grad_l2_0_conv1_w_rows2": ("trunk.layer2.0.conv1.weight", 2), "grad_l4_1_conv2_w_rows2": ("trunk.layer4.1.conv2.weight", 2),
           "grad_l3_1_relu2": ("trunk.layer3.1.relu2.weight", None),
           "grad_tcn0_cbcr0_1_w_rows4": ("tcn.mb_ms_tcn.network.0.cbcr0_1.conv.weight", 4),
           "grad_tcn3_down_b": ("tcn.mb_ms_tcn.network.3.downsample.bias", None), "grad_tcn_out_w_rows4": ("tcn.tcn_output.weight", 4)}
    # Gradients below layer 4 are ill-conditioned on 2 clips (BatchNorm over 126 rows in layer 4 amplifies
    # rounding): the reference's OWN fp32 gradients sit 3e-3 .. 9e-3 from the fp64 gradients of the same class
    # (captured next to them as *_f64).  The bar is therefore the fp64 value: 1e-4 where the reference's fp32
    # holds 1e-4 itself, otherwise no further from fp64 than twice the reference's fp32 is.
    def f64_bar(got, k32, k64):
        ref64 = g[k64]
        scale = np.abs(ref64).max()
        e_ref = float(np.abs(g[k32].astype(np.float64) - ref64).max() / scale)
        e_got = float(np.abs(got.astype(np.float64) - ref64).max() / scale)
        return e_got, max(1e-4, 2.0 * e_ref)
    for gk, (pk, rows) in sel.items():
        got = grads[pk] if rows is None else grads[pk][:rows]
        e, bar = f64_bar(got.cpu().numpy(), gk, gk + "_f64")
        assert e < bar, (gk, e, bar)
    # every parameter's gradient norm: our worst distance from fp64 vs the reference fp32's worst distance
    worst_got, worst_ref, who = 0.0, 0.0, None
    for k, v in grads.items():
        n64, n32 = float(g[f"gradnorm64_{k}"][0]), float(g[f"gradnorm_{k}"][0])
        if n64 < 1e-7:          # conv biases in front of a BatchNorm: the exact gradient is zero
            assert float(v.double().norm()) < 1e-4, k
            continue
        e_got = abs(float(v.double().norm()) - n64) / n64
        worst_ref = max(worst_ref, abs(n32 - n64) / n64)
        if e_got > worst_got:
            worst_got, who = e_got, k
    assert worst_got < max(1e-4, 2.0 * worst_ref), (who, worst_got, worst_ref)
    opt.step()
    assert rel_err(net.frontend3D[1].running_var.cpu().numpy(), g["after1_stem_running_var"]) < 1e-5
    assert rel_err(net.trunk.layer4[1].bn2.running_mean.cpu().numpy(), g["after1_l4_1_bn2_running_mean"]) < 1e-4
    assert rel_err(net.tcn.mb_ms_tcn.network[0].cbcr0_2.batchnorm.running_var.cpu().numpy(), g["after1_tcn0_cbcr0_2_running_var"]) < 1e-4
    with torch.no_grad():
        loss1 = ag.margin_ce_loss(net(x, lengths=[7, 5]), lab)
    # after one Adam step every weight moved by +-lr: the second loss is a full-model function of 36 M sign decisions
    assert abs(float(loss1) - float(g["loss1"])) < 2e-2 * max(float(g["loss1"]), 1e-3)


def test_lipreading_train_mode_with_dropout_and_eval_roundtrip():
    """Dropout on (shipped config, 0.2): loss finite and decreasing over a few Adam steps; the trained model then
    runs the eval-mode (folded BN, fused kernels) path and agrees with its own train-mode graph evaluated with
    running statistics, i.e. the two paths share one set of parameters."""
    net = _build(0.2)
    net.train()
    x = torch.from_numpy(wg.video_input(4, frames=5, key="vtrain.x2")).to(DEV)
    lab = torch.from_numpy(wg.labels(4, 54)).to(DEV)
    from deeplip_amd import autograd as ag
    opt = torch.optim.Adam(net.parameters(), lr=3e-4, weight_decay=1e-4)
    losses = []
    for _ in range(4):
        opt.zero_grad()
        loss = ag.margin_ce_loss(net(x, lengths=[5, 5, 4, 3]), lab)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    net.eval()
    with torch.no_grad():
        out = net(x, lengths=[5, 5, 4, 3])
    torch.cuda.synchronize()
    assert out.shape == (4, 54) and bool(torch.isfinite(out).all())


def test_add_prelu_vs_torch():
    """AddPReLUFn (the end of a residual block in one launch) == F.prelu(a + b) under torch autograd (fp64)."""
    from deeplip_amd import autograd_video as av
    from deeplip_amd.holders import PReLUParams
    a, b = rnd(5, 6, 7, 64, seed=80).requires_grad_(), rnd(5, 6, 7, 64, seed=81).requires_grad_()
    sl = (torch.rand(64, generator=torch.Generator().manual_seed(82)) * 0.5 - 0.05).requires_grad_()
    dy = rnd(5, 6, 7, 64, seed=83)
    ref = F.prelu((a.double() + b.double()).permute(0, 3, 1, 2), sl.double()).permute(0, 2, 3, 1)
    ref.backward(dy.double())
    ag_, bg, sg = (t.detach().to(DEV).requires_grad_() for t in (a, b, sl))
    y = av.AddPReLUFn.apply(ag_, bg, sg)
    y.backward(dy.to(DEV))
    torch.cuda.synchronize()
    assert rel_err(y.detach().cpu().numpy(), ref.detach().numpy()) < 1e-6
    assert rel_err(ag_.grad.cpu().numpy(), a.grad.numpy()) < 1e-6 and torch.equal(ag_.grad, bg.grad)
    assert rel_err(sg.grad.cpu().numpy(), sl.grad.numpy()) < 1e-5


def test_chomp_concat_vs_torch_slice_and_cat():
    """ChompConcatFn (one strided row copy per branch, forward and backward) == slice + cat under torch autograd, bit for bit."""
    from deeplip_amd import autograd_video as av
    B, T = 3, 11
    pads, widths = [4, 8, 12], [8, 16, 12]
    zs = [rnd(B, 1, T + p, w, seed=60 + i).to(DEV).requires_grad_() for i, (p, w) in enumerate(zip(pads, widths))]
    zr = [z.detach().clone().requires_grad_() for z in zs]
    dy = rnd(B, T, sum(widths), seed=70).to(DEV)
    y = av.chomp_concat(zs, T)
    y.backward(dy)
    ref = torch.cat([z[:, 0, p // 2: p // 2 + T] for z, p in zip(zr, pads)], dim=2)
    ref.backward(dy)
    torch.cuda.synchronize()
    assert torch.equal(y, ref)
    for a, b in zip(zs, zr):
        assert torch.equal(a.grad, b.grad)


@pytest.mark.parametrize("B,T", [(4, 9), (8, 29)])
def test_recorded_training_step_is_bit_identical_to_eager(B, T):
    """deeplip_amd.train_plan.TrainStepGraph: five optimisation steps on five different batches -- eager loop on ONE stream vs one
    eager step + a recorded step replayed four times, both with the independent branches of the graph on SIDE streams
    (video.BRANCH_STREAMS) -- leave bit-identical parameters, BatchNorm statistics and losses (same Adam variant in
    both: capturable, learning rate in a device tensor that a cosine scheduler updates every iteration; dropout off -- the
    generator's offsets under capture are torch's business).  Two DIFFERENT Adam variants diverge by 5e-4 after two steps on this
    model -- the biases in front of a BatchNorm have pure-rounding-noise gradients, which Adam turns into +-lr steps
    (tools/probes/adam_variants.py) -- so only like is compared with like."""
    from deeplip_amd import autograd as ag
    from deeplip_amd.train_plan import TrainStepGraph
    from models.video_models.model import Lipreading

    def run(graph):
        tcn = {"num_layers": 4, "kernel_size": [3, 5, 7], "dropout": 0.0, "dwpw": False, "width_mult": 1}
        net = Lipreading(num_classes=54, relu_type="prelu", tcn_options=tcn, extract_feats=False)
        sd = wg.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, prefix="video.")
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        net.to(DEV).train()
        opt = torch.optim.Adam(net.parameters(), lr=torch.tensor(3e-4, device=DEV), weight_decay=1e-4, capturable=True, fused=True)
        sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=5, eta_min=4e-8)

        def one(xb, lb, ln):
            opt.zero_grad(set_to_none=True)
            l = ag.margin_ce_loss(net(xb, lengths=ln), lb)
            l.backward()
            opt.step()
            return l

        plan = TrainStepGraph(one, eager_steps=1) if graph else None
        losses = []
        for i in range(5):
            x = torch.from_numpy(wg.video_input(B, frames=T, key=f"tsg.v{i}")).to(DEV)
            lab = torch.from_numpy((wg.labels(B, 54) + 7 * i) % 54).to(DEV)
            ln = torch.tensor([T - (i % 4) for i in range(B)], dtype=torch.int32, device=DEV)
            l = plan.step(x, lab, ln) if graph else one(x, lab, ln)
            sched.step()
            losses.append(float(l.detach()))
        if graph:
            plan.finish()
            assert plan.recorded
        torch.cuda.synchronize()
        return losses, {k: v.detach().clone() for k, v in net.state_dict().items()}

    le, se = run(False)
    av_mod = __import__("deeplip_amd.autograd_video", fromlist=["x"])
    av_mod._CONST.clear()       # the recorded run must CREATE the shared constant vectors itself -- inside its forked branches
    lg, sg = run(True)
    assert le == lg
    for k in se:
        assert torch.equal(se[k], sg[k]), k


def test_full_size_training_step_loss_and_gradients_vs_fp64_oracle():
    """The optimisation step at the size bench.py times it (configs.F2_train_video_step: B = 32 clips x 29 frames, 54 classes,
    ragged lengths): loss, logits, argmax and EVERY parameter's gradient against the oracle's train-mode restatement
    (oracle.lipreading_logits_train: batch-statistics BatchNorm, tcn.py:52-59 statistics over the padded length) evaluated in
    fp64 with torch autograd on the host cores (~1.8 TFLOP, a quarter of a minute on 16 cores).

    What the bar can be.  The network is piecewise linear (PReLU kinks, max-pool choices): a rounding difference that moves one
    pre-activation across zero changes the gradient by a finite amount, so ANY fp32 evaluation sits 1e-3 .. 3e-2 (largest element
    error over largest element, per tensor) from the fp64 gradient -- measured here with the same oracle run in fp32, which is what
    the reference itself computes in.  Loss and logits hold the plain 1e-4; the gradients are held to the fp32 noise floor: the
    worst tensor no further from fp64 than twice the fp32 oracle's worst, and the average over tensors no more than twice its
    average.  (This test found a real bug: with more than 32 images per batch the weight gradients of layer 4 -- 3x3 maps, a
    nine-tap "filter" -- read their slice-major operand images with pixel-major strides and were off by 130-200 %;
    conv_igemm_f16x3_dma.hip, dlip_conv_f16x3_dma_launch.  The 2-clip golden has a single 32-image slice, where the layouts coincide.)"""
    from deeplip_amd import autograd as ag
    from oracle import deeplip_oracle as O
    B, T = 32, 29
    net = _build(0.0)
    net.train()
    x = torch.from_numpy(wg.video_input(B, frames=T, key="vtrain.full"))
    lab = torch.from_numpy(wg.labels(B, 54))
    lengths = [T - (i % 5) for i in range(B)]
    lengths[0] = T
    sd0 = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    logits = net(x.to(DEV), lengths=lengths)
    loss = ag.margin_ce_loss(logits, lab.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    grads = {k: v.grad.detach().cpu().double() for k, v in net.named_parameters()}
    names = [k for k, _ in net.named_parameters()]
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))     # (a GPU box grants 16 cores; more threads than that thrash)

    def oracle(dtype):
        p = {k: (v.to(dtype) if v.dtype.is_floating_point else v.clone()) for k, v in sd0.items()}
        for k in names:
            p[k].requires_grad_(True)
        lo = O.lipreading_logits_train(p, x.to(dtype), lengths)
        ls = F.cross_entropy(lo, lab)
        ls.backward()
        return lo.detach(), float(ls.detach()), {k: p[k].grad.double() for k in names}

    ref_logits, ref_loss, g64 = oracle(torch.float64)
    _, _, g32 = oracle(torch.float32)
    assert abs(float(loss.detach()) - ref_loss) < 1e-4 * abs(ref_loss)
    assert rel_err(logits.detach().cpu().numpy(), ref_logits.numpy()) < 1e-4
    assert np.array_equal(torch.max(logits, 1)[1].cpu().numpy(), torch.max(ref_logits, 1)[1].numpy())
    ours, floor = [], []
    for k in names:
        scale = float(g64[k].abs().max())
        if scale < 1e-9:           # conv biases in front of a BatchNorm: the exact gradient is zero
            assert float(grads[k].abs().max()) < 1e-6, k
            continue
        ours.append((float((grads[k] - g64[k]).abs().max()) / scale, k))
        floor.append(float((g32[k] - g64[k]).abs().max()) / scale)
    worst, who = max(ours)
    assert worst < max(1e-4, 2.0 * max(floor)), (who, worst, max(floor))
    assert np.mean([e for e, _ in ours]) < max(1e-4, 2.0 * np.mean(floor)), (np.mean([e for e, _ in ours]), np.mean(floor))
    # gradient norms: well conditioned (a flipped kink moves single elements, not a tensor's norm)
    for k in names:
        n64 = float(g64[k].norm())
        if n64 > 1e-7:
            assert abs(float(grads[k].norm()) - n64) < 5e-3 * n64, (k, float(grads[k].norm()), n64)


@pytest.mark.parametrize("N,Ho,Wo,Hu,Wu,C,sh,sw", [(3, 11, 11, 22, 22, 128, 2, 2), (2, 6, 6, 12, 11, 256, 2, 2), (5, 1, 20, 1, 40, 64, 1, 2),
                                                   (2, 3, 3, 7, 6, 8, 2, 2), (1, 4, 5, 10, 13, 4, 3, 3)])
def test_upsample_zero(N, Ho, Wo, Hu, Wu, C, sh, sw):
    """dlip_upsample_zero_f32 (the zero insertion in front of a strided layer's data-gradient convolution): out[n, hu, wu] =
    dz[n, hu / sh, wu / sw] where both divide and the quotient exists, zero elsewhere -- bit for bit."""
    from deeplip_amd._lib import check, lib, ptr, stream_handle
    dz = rnd(N, Ho, Wo, C, seed=71)
    out = torch.full((N, Hu, Wu, C), 9.0, device=DEV)
    dzd = dz.to(DEV)
    check(lib().dlip_upsample_zero_f32(ptr(dzd), ptr(out), N, Ho, Wo, Hu, Wu, C, sh, sw, stream_handle()), "dlip_upsample_zero_f32")
    torch.cuda.synchronize()
    ref = torch.zeros(N, Hu, Wu, C)
    hs, ws = min(Ho, (Hu + sh - 1) // sh), min(Wo, (Wu + sw - 1) // sw)
    ref[:, ::sh, ::sw][:, :hs, :ws] = dz[:, :hs, :ws]
    assert torch.equal(out.cpu(), ref)


def test_prepared_weight_images_equal_the_per_convolution_ones_and_follow_the_weights():
    """Round 4: after the first step registered them, `prepare_weights()` writes every (weight, bank) split image of the step in ONE
    launch.  (1) Bit for bit the images dlip_split_weights_perm_f32 writes one by one (forward bank, data-gradient bank, padded
    rows); (2) conv_train takes a prepared image only while the weight is unmodified: an in-place update in between falls back to
    the per-convolution launch, so a stale image is never used."""
    from deeplip_amd import autograd_video as av
    from deeplip_amd._lib import check, lib, ptr, stream_handle
    av.WEIGHT_PREP.__init__()
    ws_ = [(rnd(64, 32, 3, 3, seed=1) * 0.1).to(DEV), (rnd(96, 64, 1, 5, seed=2) * 0.1).to(DEV), (rnd(128, 64, 1, 1, seed=3) * 0.1).to(DEV)]
    xs = [rnd(2, 6, 6, 32, seed=4).to(DEV), rnd(2, 1, 20, 64, seed=5).to(DEV), rnd(2, 1, 9, 64, seed=6).to(DEV)]
    pads = [(1, 1), (0, 2), (0, 0)]

    def run():
        outs = []
        for w, x, p in zip(ws_, xs, pads):
            y = av.conv_train(x, None, None, (1, 1), p, (1, 1), w_ref=w)
            g = av.conv_train(y, None, None, (1, 1), (w.shape[2] - 1 - p[0], w.shape[3] - 1 - p[1]), (1, 1), lift=True, w_ref=w, transposed=True)
            outs += [y.clone(), g.clone()]
        return outs

    first = run()                                         # registers six images
    assert len(av.WEIGHT_PREP.order) == 6
    av.prepare_weights()
    for k in av.WEIGHT_PREP.order:
        e = av.WEIGHT_PREP.entries[k]
        ws1, sc1 = torch.empty_like(e["ws"]), torch.empty_like(e["wsc"])
        check(lib().dlip_split_weights_perm_f32(ptr(e["t"]), ptr(ws1), ptr(sc1), e["Ko"], e["Ci"], e["T"], e["mode"], e["Cw"], stream_handle()), "split")
        torch.cuda.synchronize()
        assert torch.equal(e["ws"].view(torch.int32), ws1.view(torch.int32)) and torch.equal(e["wsc"], sc1)
    second = run()                                        # from the prepared images
    for a, b in zip(first, second):
        assert torch.equal(a, b)
    with torch.no_grad():
        ws_[0].mul_(2.0)                                  # an in-place update: the prepared image of weight 0 is stale now
    third = run()
    assert torch.allclose(third[0], 2.0 * first[0], rtol=1e-5, atol=1e-6) and torch.equal(third[2], first[2])
    av.prepare_weights()
    fourth = run()
    for a, b in zip(third, fourth):
        assert torch.equal(a, b)
    av.WEIGHT_PREP.__init__()


def test_weight_registry_drops_entries_whose_owner_moved_or_died():
    """Round 5 (advisor): the registry's entries held a detached alias of the weight, which pins the OLD storage -- so neither
    `param.data = ...` nor `model.to()` nor a deleted model could ever be noticed by comparing addresses, and stale entries were
    re-split every step for nobody.  Now every entry knows its owner weakly: reassigned storage / a dead owner drops the entry at the
    next prepare, entries nobody asks for age out, and results follow the new weights."""
    import gc
    from deeplip_amd import autograd_video as av
    av.WEIGHT_PREP.__init__()
    w = torch.nn.Parameter((rnd(64, 32, 3, 3, seed=11) * 0.1).to(DEV))
    w2 = torch.nn.Parameter((rnd(64, 32, 1, 1, seed=12) * 0.1).to(DEV))
    x = rnd(2, 6, 6, 32, seed=13).to(DEV)
    run = lambda: (av.conv_train(x, None, None, (1, 1), (1, 1), (1, 1), w_ref=w).clone(),
                   av.conv_train(x, None, None, (1, 1), (0, 0), (1, 1), w_ref=w2.view(64, 32, 1, 1)).clone())      # a view: owner = its base
    first = run()
    assert len(av.WEIGHT_PREP.order) == 2
    av.prepare_weights()
    h0 = av.WEIGHT_PREP.stats["hit"]
    second = run()
    assert av.WEIGHT_PREP.stats["hit"] == h0 + 2 and torch.equal(first[0], second[0]) and torch.equal(first[1], second[1])
    w.data = (w.data * 2.0).clone()                         # new storage: the registered address is the old one
    av.prepare_weights()
    assert len(av.WEIGHT_PREP.order) == 1                   # the stale entry is gone (and with it the pinned old weights)
    third = run()                                           # re-registers at the new address, computes with the new values
    assert len(av.WEIGHT_PREP.order) == 2 and torch.allclose(third[0], 2.0 * first[0], rtol=1e-5, atol=1e-6)
    del w2, run
    gc.collect()
    av.prepare_weights()
    assert len(av.WEIGHT_PREP.order) == 1                   # owner deleted
    for _ in range(10):                                     # nobody asks for the remaining image: it ages out
        av.prepare_weights()
    assert len(av.WEIGHT_PREP.order) == 0
    av.WEIGHT_PREP.__init__()


def test_conv_train_fn_one_frame_wide_output_from_a_permuted_gradient():
    """A Conv1d whose output is ONE frame wide, its gradient arriving as a permuted view [N,K,1,1] -> NHWC: torch calls that view contiguous
    with stride(2) == 1 (size-1 dimensions' strides do not count), and the operand producers took stride(2) for the pixel pitch
    (dlip_wgrad_chwn_f32: invalid argument; found by tools/probes/grad_fuzz.py -- 590 random shapes of this function and of the TDNN block
    against fp64 autograd, nothing else outside the bars, worst 2.3e-6)."""
    from deeplip_amd import autograd_video as av
    N, W, C, K, S, dil = 2, 9, 768, 100, 3, 4
    x = rnd(N, C, 1, W, seed=41).requires_grad_()
    w = rnd(K, C, 1, S, seed=42, scale=1.0 / np.sqrt(C * S)).requires_grad_()
    b = rnd(K, seed=43, scale=0.1).requires_grad_()
    ref = F.conv2d(x.double(), w.double(), b.double(), dilation=(1, dil))
    assert ref.shape == (N, K, 1, 1)
    dy = rnd(*ref.shape, seed=44)
    ref.backward(dy.double())
    xg = nhwc(x.detach()).to(DEV).requires_grad_()
    wgp, bg = w.detach().to(DEV).requires_grad_(), b.detach().to(DEV).requires_grad_()
    y = av.conv(xg, wgp, bg, dil=(1, dil))
    g = dy.to(DEV).permute(0, 2, 3, 1)                    # NOT copied by .contiguous(): strides (K, 1, 1, 1)
    assert g.is_contiguous() and g.stride(2) == 1
    y.backward(g)
    torch.cuda.synchronize()
    assert rel_err(y.detach().cpu().numpy(), nhwc(ref.detach()).numpy()) < 2e-5
    assert rel_err(xg.grad.cpu().numpy(), nhwc(x.grad).numpy()) < 1e-4
    assert rel_err(wgp.grad.cpu().numpy(), w.grad.numpy()) < 1e-4
    assert rel_err(bg.grad.cpu().numpy(), b.grad.numpy()) < 1e-4

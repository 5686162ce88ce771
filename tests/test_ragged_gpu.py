"""Ragged (variable-length) batches on the embedding path (-m gpu).

The reference's test loop takes every utterance and every lip clip AT ITS OWN LENGTH, batch 1 (train_fusion.py:334-349:
``extract_embedding(audio[1,24,T_i])``, ``model_video(v[1,1,T_j,88,88])`` then the mean over T_j).  The engine takes the
zero-padded batch + length vector of pad_packed_collate (models/video_models/dataset.py:123-139) and must return, row by row, what
that loop returns.  Oracle = the CPU restatement run ONE ITEM AT A TIME, exactly as the reference's loop does; bars: 1e-4
element-wise (conftest.assert_close_rel), speaker argmax equal; both arithmetic modes.

The padding is filled with GARBAGE on purpose wherever the engine promises to ignore it (the speech encoder never reads it for a
valid frame; the lip-clip encoder's pre-pass overwrites it with zeros).
"""
import numpy as np
import pytest
import torch

from conftest import assert_close_rel, rel_err
from deeplip_amd import weightgen as wg
from oracle import deeplip_oracle as O
from test_models_gpu import DEV, TCN_OPTS, TOL, close, etdnn_opts, load, tdnn_opts

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["f32", "f16x3"])
def mode(request):
    from deeplip_amd import packing
    packing.set_precision(request.param)
    yield request.param
    packing.set_precision("f32")


def _ragged_audio(lengths, F, key, garbage=True):
    """[B,F,Tmax]: utterance b = its own seeded [F,T_b] features, the padding behind it garbage (or zeros)."""
    Tmax = max(lengths)
    x = np.zeros((len(lengths), F, Tmax), dtype=np.float32)
    if garbage:
        x[:] = 37.0 * wg.audio_input(len(lengths), F, Tmax, key=key + ".garbage")
    items = []
    for b, T in enumerate(lengths):
        it = wg.audio_input(1, F, T, key=f"{key}.{b}", speakers=[b % 5])[0]
        x[b, :, :T] = it
        items.append(it)
    return torch.from_numpy(x), items


def _ragged_video(lengths, key, garbage=True):
    Tmax = max(lengths)
    x = np.zeros((len(lengths), 1, Tmax, 88, 88), dtype=np.float32)
    if garbage:
        x[:] = 5.0 * wg.video_input(len(lengths), Tmax, 88, key=key + ".garbage")
    items = []
    for b, T in enumerate(lengths):
        it = wg.video_input(1, T, 88, key=f"{key}.{b}", speakers=[b % 5])[0]      # [1,T,88,88]
        x[b, :, :T] = it
        items.append(it)
    return torch.from_numpy(x), items


# ------------------------------------------------------------------------------------------ speech encoder
@pytest.mark.parametrize("fuse", [True, False])
def test_audio_ragged_rows_equal_the_reference_loop(mode, fuse):
    """E-TDNN (input_dim 80), utterances of 137 .. 412 frames in one zero-padded batch: every row against the oracle run on that
    utterance alone, and the speaker argmax through an LMCL head.  ``fuse``: statistics pooling in the last layer's epilogue
    (dlip_conv_pool_f16x3 with the length vector; f16x3 only) / the unfused pooling kernel (dlip_meanstd_pool_f32 with it)."""
    from deeplip_amd import audio as A, ops
    from models.audio_models.tdnn import SpeakerEmbNet
    net, sd = load(SpeakerEmbNet(etdnn_opts(80)), "audio80.")
    lengths = [137, 300, 412, 200, 161, 255, 412, 138]
    x, items = _ragged_audio(lengths, 80, "ragged.audio")
    try:
        A.FUSE_POOL = fuse
        xv, xa = net.extract_embedding(x.to(DEV), lengths=lengths)
        xv_dev, _ = net.extract_embedding(x.to(DEV), lengths=torch.tensor(lengths, dtype=torch.int32, device=DEV))
        one = [net.extract_embedding(torch.from_numpy(it[None]).to(DEV))[0] for it in items]      # the engine, one at a time
    finally:
        A.FUSE_POOL = True
    torch.cuda.synchronize()
    assert torch.equal(xv, xv_dev)                                    # host list and device vector: the same launches
    W = torch.from_numpy(wg.gen("ragged.lmcl.W", (57, 512)))
    with torch.no_grad():
        for b, it in enumerate(items):
            rxv, rxa = O.speaker_extract_embedding(sd, torch.from_numpy(it[None]), O.ETDNN_CONTEXT)
            close(xv[b:b + 1].cpu().numpy(), rxv.numpy(), what=f"xv row {b} (T={lengths[b]})")
            close(xa[b:b + 1].cpu().numpy(), rxa.numpy(), what=f"x_a row {b}")
            assert rel_err(xv[b:b + 1].cpu().numpy(), one[b].cpu().numpy()) < 1e-6
            _, got = ops.logits_argmax(xv[b:b + 1].contiguous(), W.to(DEV), cosine=True)
            want = O.argmax_first(torch.nn.functional.normalize(rxv) @ torch.nn.functional.normalize(W).t())
            assert int(got.cpu()[0]) == int(want[0])
    # and the padding really is ignored: other garbage, same rows
    x2 = x.clone()
    for b, T in enumerate(lengths):
        x2[b, :, T:] = -x2[b, :, T:] + 1.0
    xv2, _ = net.extract_embedding(x2.to(DEV), lengths=lengths)
    torch.cuda.synchronize()
    assert torch.equal(xv2, xv)


def test_audio_ragged_tdnn5_and_attentive_pooling(mode):
    from models.audio_models.tdnn import SpeakerEmbNet
    lengths = [90, 64, 77, 90, 33]
    for opts, ctx, pool in ((tdnn_opts(), O.TDNN_CONTEXT, "statistic"),
                            (etdnn_opts(24, pooling="attentive_statistic"), O.ETDNN_CONTEXT, "attentive_statistic")):
        net, sd = load(SpeakerEmbNet(opts), f"ragged.{pool}.")
        x, items = _ragged_audio(lengths, 24, "ragged.audio24")
        xv, _ = net.extract_embedding(x.to(DEV), lengths=lengths)
        torch.cuda.synchronize()
        with torch.no_grad():
            for b, it in enumerate(items):
                rxv, _ = O.speaker_extract_embedding(sd, torch.from_numpy(it[None]), ctx, pooling=pool)
                close(xv[b:b + 1].cpu().numpy(), rxv.numpy(), what=f"{pool} row {b}")


def test_audio_lengths_are_validated():
    from models.audio_models.tdnn import SpeakerEmbNet
    net, _ = load(SpeakerEmbNet(etdnn_opts(24)), "ragged.val.")
    x = torch.zeros(2, 24, 100, device=DEV)
    with pytest.raises(ValueError):
        net.extract_embedding(x, lengths=[100, 23])            # 23 - 22 consumed frames = 1 pooled frame: no unbiased std
    with pytest.raises(ValueError):
        net.extract_embedding(x, lengths=[100, 101])           # longer than the padded batch
    with pytest.raises(ValueError):
        net.extract_embedding(x, lengths=[100])                # one length per utterance
    with pytest.raises(TypeError):
        net.extract_embedding(x, lengths=torch.tensor([100, 50], device=DEV))   # a device vector must be int32


# ------------------------------------------------------------------------------------------ lip-clip encoder
@pytest.fixture(scope="module")
def video_net():
    from models.video_models.model import Lipreading
    net = Lipreading(hidden_dim=256, backbone_type="resnet", num_classes=54, relu_type="prelu",
                     tcn_options=TCN_OPTS, width_mult=1.0, extract_feats=True)
    return load(net, "video.")


@pytest.mark.slow
@pytest.mark.parametrize("lengths", [[29, 11, 40, 33, 40], [11, 20, 14]], ids=["pooled-epilogue", "short-clips"])
def test_video_ragged_rows_equal_the_reference_loop(mode, video_net, lengths):
    """Clips of 11 .. 40 frames, zero-padded (garbage-padded, in fact) to the longest: row b of embed(x, lengths) against the
    oracle's features of clip b ALONE, averaged over its own frames (train_fusion.py:346-348).  [.., 40]: 40 * 9 rows >= the
    pooled tile, so in f16x3 mode the mean comes out of the last convolution's epilogue with the length vector; [.., 20]: below
    it -> the unfused masked mean."""
    net, sd = video_net
    x, items = _ragged_video(lengths, "ragged.video")
    em = net.embed(x.to(DEV), lengths=lengths)
    one = [net.embed(torch.from_numpy(it[None]).to(DEV)) for it in items]
    torch.cuda.synchronize()
    with torch.no_grad():
        for b, it in enumerate(items):
            ref = O.video_time_mean(O.lipreading_features(sd, torch.from_numpy(it[None])))
            close(em[b:b + 1].cpu().numpy(), ref.numpy(), what=f"clip {b} (T={lengths[b]})")
            assert rel_err(em[b:b + 1].cpu().numpy(), one[b].cpu().numpy()) < 1e-6


def test_video_ragged_uint8_frames(mode, video_net):
    """uint8 RGB frames [B,T,3,96,96] (BASELINE.json's input) in a ragged batch: the padding must be made by the stem's
    pre-pass -- byte 0 is not a zero of the normalised clip.  Against the engine on each clip's own frames."""
    net, _ = video_net
    lengths = [31, 12, 40]
    r = np.random.Generator(np.random.PCG64(11))
    frames = torch.from_numpy(r.integers(0, 256, size=(3, 40, 3, 96, 96), dtype=np.uint8))
    em = net.embed(frames.to(DEV), lengths=lengths)
    one = [net.embed(frames[b:b + 1, :T].contiguous().to(DEV)) for b, T in enumerate(lengths)]
    torch.cuda.synchronize()
    for b in range(3):
        assert rel_err(em[b:b + 1].cpu().numpy(), one[b].cpu().numpy()) < 1e-6, b


def test_fuse_av_on_ragged_pooled_means_bit_identical(video_net):
    """fusion.fuse_av over the still-pooled ragged clip means (dlip_znorm_cat_pooled_f32 with the length vector) == pool_finish +
    znorm_cat."""
    from deeplip_amd import fusion, ops, packing
    packing.set_precision("f16x3")
    try:
        net, _ = video_net
        lengths = [29, 30, 35]
        x, _ = _ragged_video(lengths, "ragged.fuse", garbage=False)
        a = torch.from_numpy(wg.gen("ragged.fuse.a", (3, 512))).to(DEV)
        pooled = net.embed(x.to(DEV), lengths=lengths, finish=False)
        assert isinstance(pooled, ops.Pooled) and pooled.lengths is not None and pooled.len_mul == 9
        one = fusion.fuse_av(a, pooled)
        two = fusion.fuse_av(a, ops.pool_finish(pooled, "mean"))
        torch.cuda.synchronize()
        assert torch.equal(one, two)
    finally:
        packing.set_precision("f32")


def test_one_recorded_plan_serves_a_bucket():
    """Lengths are DEVICE tensors, so a plan recorded on one batch of a length bucket replays on the next: same padded shape,
    other lengths, other clips."""
    from deeplip_amd import fusion, packing
    from deeplip_amd.plan import StepPlan
    from models.audio_models.tdnn import SpeakerEmbNet
    from models.video_models.model import Lipreading
    packing.set_precision("f16x3")
    try:
        vnet, _ = load(Lipreading(hidden_dim=256, num_classes=54, relu_type="prelu", tcn_options=TCN_OPTS, extract_feats=True), "video.")
        anet, _ = load(SpeakerEmbNet(etdnn_opts(80)), "audio80.")
        la, lv = [300, 280, 290], [31, 29, 33]
        xa, _ = _ragged_audio(la, 80, "plan.a0", garbage=False)
        xv, _ = _ragged_video(lv, "plan.v0", garbage=False)
        ins = (xv.to(DEV), torch.tensor(lv, dtype=torch.int32, device=DEV), xa.to(DEV), torch.tensor(la, dtype=torch.int32, device=DEV))

        def step(v, vl, a, al):
            return fusion.fuse_av(anet.extract_embedding(a, lengths=al)[0], vnet.embed(v, lengths=vl, finish=False))

        plan = StepPlan(step, *ins)
        first = plan.run().clone()
        torch.cuda.synchronize()
        assert torch.equal(first, step(*ins))
        la2, lv2 = [285, 300, 281], [33, 30, 29]               # the same bucket: padded to 300 / 33 frames
        xa2, _ = _ragged_audio(la2, 80, "plan.a1", garbage=False)
        xv2, _ = _ragged_video(lv2, "plan.v1", garbage=False)
        ins2 = (xv2.to(DEV), torch.tensor(lv2, dtype=torch.int32, device=DEV), xa2.to(DEV), torch.tensor(la2, dtype=torch.int32, device=DEV))
        got = plan(*ins2).clone()
        torch.cuda.synchronize()
        want = step(*ins2)
        torch.cuda.synchronize()
        assert torch.equal(got, want) and not torch.equal(got, first)
        plan.close()
    finally:
        packing.set_precision("f32")


# ------------------------------------------------------------------------------------------ the kernels themselves
def test_meanstd_time_mean_with_lengths_vs_torch():
    from deeplip_amd import ops
    r = torch.Generator().manual_seed(3)
    x = torch.randn(5, 50, 24, generator=r)
    lens = [50, 7, 33, 2, 49]
    l = torch.tensor(lens, dtype=torch.int32, device=DEV)
    got = ops.meanstd_pool(x.to(DEV), lengths=l).cpu()
    gm = ops.time_mean(x.to(DEV), l).cpu()
    got_add = ops.meanstd_pool(x.to(DEV), lengths=l + 3, len_add=-3).cpu()
    for b, T in enumerate(lens):
        want = torch.cat([x[b, :T].double().mean(0), x[b, :T].double().std(0)]).float()
        assert_close_rel(got[b].numpy(), want.numpy(), rtol=1e-6, what=f"meanstd row {b}")
        assert_close_rel(gm[b].numpy(), x[b, :T].double().mean(0).float().numpy(), rtol=1e-6, what=f"time mean row {b}")
    assert torch.equal(got, got_add)


@pytest.mark.parametrize("rows_kernel", [False, True])
def test_conv_pool_with_lengths_equals_conv_then_masked_pooling(rows_kernel):
    """dlip_conv_pool_f16x3 with a length vector against the same convolution written out + dlip_meanstd_pool_f32 with it:
    group boundaries inside tiles, valid ends inside tiles, a group whose valid rows end before a tile starts.  ``rows_kernel``:
    the rows kernel's pooled epilogue (forced: dlip_debug_set(6, 1))."""
    from deeplip_amd import _lib, ops, packing
    r = np.random.Generator(np.random.PCG64(5))
    B, T, C, K = 5, 300, 64, 256
    x = torch.from_numpy(r.standard_normal((B, T, C)).astype(np.float32)).to(DEV)
    w = torch.from_numpy((r.standard_normal((K, 1, C)) / 8).astype(np.float32))
    bias = torch.from_numpy(r.standard_normal(K).astype(np.float32)).to(DEV)
    ws, wscale = packing.split_weights(w.view(K, 1, 1, C).double())
    ws, wscale = ws.to(DEV), wscale.to(DEV)
    xs = ops.split_pack(x)
    lens = torch.tensor([300, 140, 17, 299, 129], dtype=torch.int32, device=DEV)
    try:
        if rows_kernel:
            _lib.debug_set(_lib.DBG_ROWS, 4)
        pooled = ops.conv_pool(xs.view(B, 1, T, C), ws.view(K, 1, 1, -1), bias, wscale, T, lengths=lens, len_add=-5)
        got = ops.pool_finish(pooled, "meanstd")
    finally:
        _lib.debug_set(_lib.DBG_ROWS, -1)
    y = ops.conv_nhwc(xs.view(B, 1, T, C), ws.view(K, 1, 1, -1), bias, w_scale=wscale, x_split=True)
    want = ops.meanstd_pool(y.view(B, T, K), lengths=lens, len_add=-5)
    torch.cuda.synchronize()
    assert_close_rel(got.cpu().numpy(), want.cpu().numpy(), rtol=2e-6, what="ragged pooled epilogue")


def test_stem_prepass_zeroes_padding_frames_and_flips():
    """The pre-pass contract at the kernel boundary (dlip_stem3d_pool_u8_f16x3): (1) lengths -> the stem output of the valid frames
    equals the clip cut to its length, whatever bytes sit in the padding; (2) clip_params -> crop origin + horizontal flip per
    clip equal the same clip cropped / flipped on the host and fed with the batch-wide origin."""
    from deeplip_amd import ops, packing
    from models.video_models.model import Lipreading
    packing.set_precision("f16x3")
    try:
        net, _ = load(Lipreading(hidden_dim=256, num_classes=54, relu_type="prelu", tcn_options=TCN_OPTS, extract_feats=True), "video.")
        from deeplip_amd.video import _cached_pack
        p = _cached_pack(net, torch.device(DEV), net._pack)["stem"]
        r = np.random.Generator(np.random.PCG64(7))
        fr = r.integers(0, 256, size=(3, 9, 95, 91), dtype=np.uint8)             # gray [B,T,Hs,Ws], odd margins
        lens = [9, 4, 6]
        l = torch.tensor(lens, dtype=torch.int32, device=DEV)
        y = ops.stem3d_pool_u8(torch.from_numpy(fr).to(DEV), p.w, p.b, p.slope, p.wscale, lengths=l).view(3, 9, 22, 22, 64)
        for b, T in enumerate(lens):
            yb = ops.stem3d_pool_u8(torch.from_numpy(fr[b:b + 1, :T].copy()).to(DEV), p.w, p.b, p.slope, p.wscale).view(1, T, 22, 22, 64)
            torch.cuda.synchronize()
            assert torch.equal(y[b, :T], yb[0]), b
        params = np.array([[0, 0, 0, 0], [7, 3, 1, 0], [2, 1, 1, 0]], dtype=np.int32)
        yp = ops.stem3d_pool_u8(torch.from_numpy(fr).to(DEV), p.w, p.b, p.slope, p.wscale,
                                clip_params=torch.from_numpy(params).to(DEV)).view(3, 9, 22, 22, 64)
        for b, (oy, ox, flip, _) in enumerate(params):
            crop = fr[b:b + 1, :, oy:oy + 88, ox:ox + 88]
            if flip:
                crop = crop[..., ::-1]
            yb = ops.stem3d_pool_u8(torch.from_numpy(np.ascontiguousarray(crop)).to(DEV), p.w, p.b, p.slope, p.wscale).view(1, 9, 22, 22, 64)
            torch.cuda.synchronize()
            assert torch.equal(yp[b], yb[0]), b
    finally:
        packing.set_precision("f32")


# ------------------------------------------------------------------------------------------ the extraction flow
def test_ragged_extractor_equals_one_at_a_time(video_net):
    """deeplip_amd.extract.RaggedExtractor over a ragged synthetic list (1-2 clips per utterance, short batches, several length
    rungs, uint8 RGB frames): every row of both tables == the engine run on that utterance's audio / clips alone, in LIST order;
    a second pass over the list records no further plan."""
    from deeplip_amd import ops, packing
    from deeplip_amd.extract import RaggedExtractor
    from deeplip_amd.synthetic import SyntheticAVSet, frames_u8_from_clips
    from models.audio_models.tdnn import SpeakerEmbNet
    packing.set_precision("f16x3")
    try:
        vnet, _ = video_net
        anet, _ = load(SpeakerEmbNet(etdnn_opts(24)), "ragged.ex.")
        ds = SyntheticAVSet(3, 5, 2, audio_dim=24, key="ragged.ex", ragged=True, audio_range=(60, 140), video_range=(6, 16))
        n = len(ds)
        ex = RaggedExtractor(lambda a, l: anet.extract_embedding(a, lengths=l)[0], lambda v, l: vnet.embed(v, lengths=l),
                             torch.device(DEV), batch=4, clip_batch=6, waste=0.25)
        xa, xv = ex.run(ds, 2, n, 512, u8=True)                       # utterances 2 .. n: a shard, as a rank would take
        first = ex.stats["plans_recorded"]
        xa2, xv2 = ex.run(ds, 2, n, 512, u8=True)
        assert ex.stats["plans_recorded"] == first and torch.equal(xa, xa2) and torch.equal(xv, xv2)
        assert ex.stats["audio_shapes"] >= 2 and ex.stats["video_shapes"] >= 2
        ex.close()
        for j, i in enumerate(range(2, n)):
            want_a = anet.extract_embedding(torch.from_numpy(ds.audio_item(i)[None]).to(DEV))[0]
            cm = [vnet.embed(torch.from_numpy(frames_u8_from_clips(ds.clip_item(c)[None, None], rgb=True)).to(DEV))
                  for c in range(int(ds.clip_ptr[i]), int(ds.clip_ptr[i + 1]))]
            want_v = torch.stack(cm).mean(0)
            torch.cuda.synchronize()
            assert rel_err(xa[j:j + 1].cpu().numpy(), want_a.cpu().numpy()) < 1e-6, i
            assert rel_err(xv[j:j + 1].cpu().numpy(), want_v.cpu().numpy()) < 2e-6, i
    finally:
        packing.set_precision("f32")


# ------------------------------------------------------------------------------------------ BASELINE's full sizes, through properties
def test_ragged_full_size_batches_properties(video_net):
    """BASELINE.json's batch sizes with RAGGED contents -- 64 clips of 11 .. 75 frames, 256 utterances of 137 .. 412 frames, f16x3:
    (1) sampled rows equal the engine run on that item alone at its own length (1e-6: the balanced split's
    sum order moves with the batch); (2) what sits in the padding does not matter, bit for bit; (3) two rows with equal content and
    DIFFERENT neighbours / positions agree; (4) a row's result does not depend on the other rows' lengths; (5) every row
    against the oracle run on that item alone at its own length (1e-4 element-wise)."""
    from deeplip_amd import packing
    from models.audio_models.tdnn import SpeakerEmbNet
    packing.set_precision("f16x3")
    try:
        vnet, sdv = video_net
        r = np.random.Generator(np.random.PCG64(21))
        lv = r.integers(11, 76, size=64).tolist()
        lv[0], lv[63] = 75, 33
        lv[40] = lv[7]
        base = torch.from_numpy(wg.video_input(8, 75, 88, key="ragged.full.v", speakers=list(range(8))))     # 8 distinct clips, cut to length
        xv = torch.zeros(64, 1, 75, 88, 88)
        for b in range(64):
            xv[b, :, :lv[b]] = base[b % 8, :, :lv[b]]
        xv[40, :, :lv[40]] = xv[7, :, :lv[7]]                      # the same clip at two places of the batch
        em = vnet.embed(xv.to(DEV), lengths=lv)
        junk = xv.clone()
        for b in range(64):
            junk[b, :, lv[b]:] = 9.0
        em_j = vnet.embed(junk.to(DEV), lengths=lv)
        lv2 = list(lv)
        for b in range(64):
            if b not in (7, 40, 63):
                lv2[b] = max(11, lv[b] - 5)
        em_l = vnet.embed(xv.to(DEV), lengths=lv2)
        torch.cuda.synchronize()
        assert em.shape == (64, 512) and bool(torch.isfinite(em).all())
        assert torch.equal(em, em_j)
        assert rel_err(em[40:41].cpu().numpy(), em[7:8].cpu().numpy()) < 1e-6
        for b in (7, 40, 63):
            assert rel_err(em_l[b:b + 1].cpu().numpy(), em[b:b + 1].cpu().numpy()) < 1e-6, b
        for b in (0, 7, 63):
            one = vnet.embed(xv[b:b + 1, :, :lv[b]].contiguous().to(DEV))
            torch.cuda.synchronize()
            assert rel_err(em[b:b + 1].cpu().numpy(), one.cpu().numpy()) < 1e-6, b
        # (5, round 5) ... and EVERY row against the oracle run on that clip alone at its own length -- the reference's loop
        # (train_fusion.py:346-348); the host cores of the box take ~10 s for the 64 clips
        with torch.no_grad():
            for b in range(64):
                ref = O.video_time_mean(O.lipreading_features(sdv, xv[b:b + 1, :, :lv[b]].contiguous()))
                close(em[b:b + 1].cpu().numpy(), ref.numpy(), what=f"clip {b} of 64 (T={lv[b]})")

        anet, sda = load(SpeakerEmbNet(etdnn_opts(80)), "audio80.")
        la = r.integers(137, 413, size=256).tolist()
        la[0], la[255] = 412, 137
        basea = torch.from_numpy(wg.audio_input(8, 80, 412, key="ragged.full.a", speakers=list(range(8))))
        xa = torch.zeros(256, 80, 412)
        for b in range(256):
            xa[b, :, :la[b]] = basea[b % 8, :, :la[b]]
        la[200] = la[9]
        xa[200] = 0.0
        xa[200, :, :la[200]] = xa[9, :, :la[9]]
        ea, _ = anet.extract_embedding(xa.to(DEV), lengths=la)
        ja = xa.clone()
        for b in range(256):
            ja[b, :, la[b]:] = -4.0
        ea_j, _ = anet.extract_embedding(ja.to(DEV), lengths=la)
        torch.cuda.synchronize()
        assert ea.shape == (256, 512) and torch.equal(ea, ea_j)
        assert rel_err(ea[200:201].cpu().numpy(), ea[9:10].cpu().numpy()) < 1e-6
        for b in (0, 9, 255):
            one, _ = anet.extract_embedding(xa[b:b + 1, :, :la[b]].contiguous().to(DEV))
            torch.cuda.synchronize()
            assert rel_err(ea[b:b + 1].cpu().numpy(), one.cpu().numpy()) < 1e-6, b
        with torch.no_grad():                                 # every one of the 256 utterances against the oracle, one at a time
            for b in range(256):
                ref, _ = O.speaker_extract_embedding(sda, xa[b:b + 1, :, :la[b]].contiguous(), O.ETDNN_CONTEXT)
                close(ea[b:b + 1].cpu().numpy(), ref.numpy(), what=f"utterance {b} of 256 (T={la[b]})")
    finally:
        packing.set_precision("f32")


def test_random_shapes_f16x3_against_the_exact_mode():
    """Shape fuzz (tools/probes/shape_fuzz.py: 340 random shapes over three seeds, worst 1.2e-6 of a row's scale for the speech encoder,
    5.5e-7 for the lip-clip encoder): 12 seeded random (B, T) per encoder, every second one ragged, the f16x3 arithmetic against the
    engine's exact mode -- which the fixed-shape tests pin to the oracle.  A tile-edge or padding bug of the window / ring / rows / pooled
    kernels is an O(1) error on some shape; the arithmetic's own tail is ~1e-6, so the bar here is rel_err < 5e-6 (max error over max
    magnitude per tensor; the fixed shapes keep the element-wise 1e-4 bar)."""
    from conftest import rel_err
    from deeplip_amd import _lib, arith, weightgen as wg
    from models.audio_models.tdnn import SpeakerEmbNet
    from models.video_models.model import Lipreading
    from oracle import deeplip_oracle as O
    opts = {"arch": "tdnn", "tdnn": {"input_dim": 24, "hidden_dim": [512] * 4 + [1500], "context": O.TDNN_CONTEXT, "tdnn_layers": 5,
                                      "embedding_dim": 512, "pooling": "statistic", "attention_hidden_size": 64, "bn_first": True}}
    anet = SpeakerEmbNet(opts)
    sd = wg.fill_state_dict({k: tuple(v.shape) for k, v in anet.state_dict().items()}, prefix="audio_tdnn.")
    anet.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    anet.eval().cuda()
    tcn = {"num_layers": 4, "kernel_size": [3, 5, 7], "dropout": 0.2, "dwpw": False, "width_mult": 1}
    vnet = Lipreading(num_classes=54, relu_type="prelu", tcn_options=tcn, extract_feats=True)
    sd = wg.fill_state_dict({k: tuple(v.shape) for k, v in vnet.state_dict().items()}, prefix="video.")
    vnet.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    vnet.eval().cuda()
    amin = anet.frames_consumed() + 2
    r = np.random.Generator(np.random.PCG64(2026))
    worst = 0.0
    for i in range(12):
        B, T = int(r.integers(1, 21)), int(r.integers(amin, 520))
        x = torch.from_numpy(wg.audio_input(B, 24, T, key=f"fuzz.t.a{i}")).cuda()
        L = None
        if i % 2:
            L = torch.from_numpy(r.integers(amin, T + 1, size=B).astype(np.int32))
            L[int(r.integers(0, B))] = T
            L = L.cuda()
        out = {}
        for mode in ("f32", "f16x3"):
            arith.configure(mode)
            out[mode] = anet.extract_embedding(x, lengths=L)[0].cpu().numpy()
            _lib.check_range(sync=True)
        e = rel_err(out["f16x3"], out["f32"])
        assert e < 5e-6, f"speech encoder B={B} T={T} ragged={L is not None}: {e:.3e}"
        worst = max(worst, e)
        B, T = int(r.integers(1, 7)), int(r.integers(1, 41))
        x = torch.from_numpy(wg.video_input(B, frames=T, key=f"fuzz.t.v{i}")).cuda()
        L = None
        if i % 2:
            L = [int(v) for v in r.integers(1, T + 1, size=B)]
            L[int(r.integers(0, B))] = T
        for mode in ("f32", "f16x3"):
            arith.configure(mode)
            out[mode] = vnet.embed(x, L).cpu().numpy()
            out[mode + "f"] = vnet(x, L).cpu().numpy()
            _lib.check_range(sync=True)
        e = max(rel_err(out["f16x3"], out["f32"]), rel_err(out["f16x3f"], out["f32f"]))
        assert e < 5e-6, f"lip-clip encoder B={B} T={T} ragged={L is not None}: {e:.3e}"
        worst = max(worst, e)
    print(f"\nshape fuzz, 24 shapes: worst f16x3 vs exact {worst:.2e}")

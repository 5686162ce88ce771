"""Preprocessing front-end: GEMM-based MFCC / fbank / logfbank and the
video crop+normalise, against the oracle's numpy restatement of python_speech_features / the
reference's 'val' transform chain.  Tolerance 1e-4 relative-to-max on normalised features."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from deeplip_amd import weightgen as wg
from oracle import deeplip_oracle as O


def test_filterbank_and_oracle_self_consistency():
    from deeplip_amd.frontend import mel_filterbank, num_frames
    fb = mel_filterbank(26, 512, 16000)
    assert fb.shape == (26, 257) and fb.min() >= 0 and abs(fb.max() - 1.0) < 1e-12
    assert num_frames(16000, 400, 160) == 99 and num_frames(300, 400, 160) == 1
    sig = wg.gen("fe.sig", (16000,)) * 0.1
    f = O.audio_features(sig, "mfcc")
    assert f.shape == (24, 99)
    assert np.abs(f.mean(1)).max() < 1e-5 and np.abs(f.std(1) - 1).max() < 1e-4    # CMVN


@pytest.mark.gpu
# 80 bands at nfft=512: 19 one-bin filters sit next to DC where pre-emphasis leaves ~1e-9 of the spectrum, i.e.
# at the fp32 noise floor of a 512-point DFT (the reference's numpy FFT runs in fp64): banks denser than the shipped
# 26 / 24 / 60 bands (conf/fusion_config.yaml:12-40) take the fp64 DFT kernel and hold 1e-4 as well.
@pytest.mark.parametrize("feat_type,num_bin,tol", [("mfcc", 26, 1e-4), ("logfbank", 60, 1e-4), ("fbank", 24, 1e-4),
                                                   ("logfbank", 80, 1e-4)])
def test_audio_frontend_vs_oracle(feat_type, num_bin, tol):
    from deeplip_amd.frontend import AudioFrontend
    B, S = 3, 16000 * 2 + 123
    t = np.arange(S) / 16000.0
    sig = np.stack([0.3 * np.sin(2 * np.pi * (200 + 150 * b) * t) + 0.05 * wg.gen(f"fe.n{b}", (S,)) for b in range(B)]).astype(np.float32)
    fe = AudioFrontend(feat_type, num_bin=num_bin)
    y = fe(torch.from_numpy(sig).cuda())
    torch.cuda.synchronize()
    for b in range(B):
        ref = O.audio_features(sig[b].astype(np.float64), feat_type, nfilt=num_bin)
        assert y[b].shape == ref.shape
        # at nfft=512 / 16 kHz the lowest mel filters of an 80-band bank are EMPTY (floor() collapses their
        # bin edges): constant log(eps) rows whose "CMVN" is rounding noise / 2e-12 in the reference -- skip them
        raw = O.audio_features(sig[b].astype(np.float64), feat_type, nfilt=num_bin, normalize=False)
        # ... and a band whose variation over the utterance is below ~1e-4 of the feature magnitude (log-energies near
        # -27 moving by 1e-5) is CMVN-ed by a standard deviation the size of an fp32 ulp of its values: ill-conditioned in
        # any arithmetic that stores the log filterbank in fp32.  Such bands are compared before the normalisation only.
        live = raw.std(axis=1) > 1e-4 * np.abs(raw).max()
        assert live.sum() >= num_bin - 8
        assert rel_err(y[b].cpu().numpy()[live], ref[live]) < tol, (feat_type, b)
    fe_raw = AudioFrontend(feat_type, num_bin=num_bin, normalize=False)
    yr = fe_raw(torch.from_numpy(sig).cuda())
    torch.cuda.synchronize()
    for b in range(B):
        raw = O.audio_features(sig[b].astype(np.float64), feat_type, nfilt=num_bin, normalize=False)
        nonempty = raw.std(axis=1) > 1e-6
        assert rel_err(yr[b].cpu().numpy()[nonempty], raw[nonempty]) < tol, ("un-normalised", feat_type, b)


@pytest.mark.gpu
def test_mfcc_feeds_the_encoder():
    """waveform -> MFCC-24 -> E-TDNN embedding, GPU end to end vs oracle end to end."""
    from deeplip_amd.frontend import AudioFrontend
    from models.audio_models.tdnn import SpeakerEmbNet
    et = {"input_dim": 24, "hidden_dim": [512] * 9 + [1500], "context": O.ETDNN_CONTEXT, "tdnn_layers": 10,
          "embedding_dim": 512, "pooling": "statistic", "attention_hidden_size": 64, "bn_first": True}
    net = SpeakerEmbNet({"arch": "etdnn", "etdnn": et})
    sd = wg.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, prefix="audio.")
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); net.eval().cuda()
    S = 16000 * 3
    sig = (0.2 * np.sin(2 * np.pi * 310 * np.arange(S) / 16000.0) + 0.05 * wg.gen("fe.enc", (S,))).astype(np.float32)
    feats = AudioFrontend("mfcc")(torch.from_numpy(sig[None]).cuda())
    xv, _ = net.extract_embedding(feats)
    ref_f = torch.from_numpy(O.audio_features(sig.astype(np.float64), "mfcc"))[None]
    with torch.no_grad():
        ref, _ = O.speaker_extract_embedding(O.to_torch_sd(sd), ref_f, O.ETDNN_CONTEXT)
    assert rel_err(xv.cpu().numpy(), ref.numpy()) < 1e-3      # features 1e-4, amplified by 10 layers + pooling


@pytest.mark.gpu
def test_video_frontend_vs_oracle():
    from deeplip_amd.frontend import VideoFrontend
    g = torch.Generator().manual_seed(3)
    gray = torch.randint(0, 256, (2, 5, 96, 96), dtype=torch.uint8, generator=g)
    rgb = torch.randint(0, 256, (2, 4, 3, 96, 100), dtype=torch.uint8, generator=g)
    vf = VideoFrontend(88)
    yg = vf(gray.cuda()); yr = vf(rgb.cuda())
    torch.cuda.synchronize()
    assert yg.shape == (2, 1, 5, 88, 88) and yr.shape == (2, 1, 4, 88, 88)
    for b in range(2):
        assert np.abs(yg[b, 0].cpu().numpy() - O.video_preprocess_u8(gray[b].numpy())).max() < 1e-5
        assert np.abs(yr[b, 0].cpu().numpy() - O.video_preprocess_u8(rgb[b].numpy())).max() < 1e-4
    clips = [gray[0, :3].cuda(), gray[1].cuda()]
    batch, lengths = vf.collate(clips)
    odd = torch.randint(0, 256, (1, 2, 3, 91, 95), dtype=torch.uint8, generator=g)      # margins 3 and 7: CenterCrop takes 1 and 3
    yo = vf(odd.cuda())
    assert np.array_equal(yo[0, 0].cpu().numpy(), O.video_preprocess_u8(odd[0].numpy()))
    ref = ((0.299 * odd[0, :, 0].float() + 0.587 * odd[0, :, 1].float() + 0.114 * odd[0, :, 2].float())[:, 1:89, 3:91] / 255.0 - 0.421) / 0.165
    assert np.abs(yo[0, 0].cpu().numpy() - ref.numpy()).max() < 1e-4
    assert lengths == [5, 3] and batch.shape == (2, 1, 5, 88, 88)
    assert float(batch[1, 0, 3:].abs().max()) == 0.0      # padded AFTER the normalisation (dataset.py:117,130-134): zeros
    assert np.abs(batch[1, 0, :3].cpu().numpy() - O.video_preprocess_u8(gray[0, :3].numpy())).max() < 1e-5


@pytest.mark.gpu
def test_video_frontend_train_pipeline_bit_exact():
    """The 'train' pipeline (dataloaders.py:13-17: RandomCrop(88) + HorizontalFlip(0.5) per clip, preprocess.py:95-138) on 91 x 95
    frames: the crop origins and flips the host draws (ops.draw_clip_params, the reference's draw order) applied by the ingest
    kernel == the oracle's restatement driven by the same generator state, BIT FOR BIT; gray and RGB; with a ragged batch's
    padding frames zeroed."""
    import random
    from deeplip_amd import ops
    from deeplip_amd.frontend import VideoFrontend
    g = torch.Generator().manual_seed(9)
    vf = VideoFrontend(88)
    for shape in ((5, 4, 91, 95), (5, 4, 3, 91, 95)):
        fr = torch.randint(0, 256, shape, dtype=torch.uint8, generator=g)
        params = ops.draw_clip_params(5, 91, 95, crop=88, rng=random.Random(7))
        lengths = torch.tensor([4, 2, 4, 1, 3], dtype=torch.int32)
        y = vf(fr.cuda(), clip_params=torch.from_numpy(params).cuda(), lengths=lengths.cuda())
        torch.cuda.synchronize()
        rng = random.Random(7)
        for b in range(5):
            want, drawn = O.video_preprocess_train_u8(fr[b].numpy(), rng)
            assert tuple(params[b, :3]) == drawn
            T = int(lengths[b])
            assert np.array_equal(y[b, 0, :T].cpu().numpy(), want[:T]), (shape, b)
            assert float(y[b, 0, T:].abs().max()) == 0.0 if T < 4 else True
    assert params[:, 2].min() == 0 and params[:, 2].max() == 1       # both coin faces were exercised


def test_oracle_delta_known_answers():
    """python_speech_features.delta restated: a ramp has slope 1 inside, edge padding flattens the ends."""
    ramp = np.arange(6, dtype=np.float64)[:, None]
    assert np.allclose(O.psf_delta(ramp, 1)[:, 0], [0.5, 1, 1, 1, 1, 0.5])
    assert np.allclose(O.psf_delta(ramp, 2)[:, 0], [0.5, 0.8, 1, 1, 0.8, 0.5])
    f = O.add_deltas(np.hstack([ramp, ramp ** 2]), order=2)
    assert f.shape == (6, 6) and np.allclose(f[:, 2], O.psf_delta(ramp, 1)[:, 0]) and np.allclose(f[:, 4], O.psf_delta(ramp, 2)[:, 0])


@pytest.mark.gpu
def test_delta_features_vs_oracle():
    """`delta: true` (datasets.py:55-63,81-82): [feat | delta N=1 | delta N=2] appended after CMVN."""
    from deeplip_amd.frontend import AudioFrontend
    S = 16000 + 77
    sig = np.stack([0.3 * np.sin(2 * np.pi * 240 * (b + 1) * np.arange(S) / 16000.0) + 0.05 * wg.gen(f"fe.d{b}", (S,)) for b in range(2)]).astype(np.float32)
    fe = AudioFrontend("mfcc", delta=True)
    assert fe.feat_dim == 72
    y = fe(torch.from_numpy(sig).cuda())
    torch.cuda.synchronize()
    for b in range(2):
        ref = O.audio_features(sig[b].astype(np.float64), "mfcc", delta=True)
        assert y[b].shape == ref.shape == (72, ref.shape[1])
        assert rel_err(y[b].cpu().numpy(), ref) < 1e-4


@pytest.mark.gpu
def test_default_route_resolves_the_bands_that_pre_emphasis_empties():
    """tools/probes/frontend_fuzz.py's find: a sine + white noise, logfbank-60 (a shipped bank), a length with a remainder against the shift.
    Band 0 -- a one-bin filter next to DC -- holds ~1e-12 of a frame's energy in some frames (log energy -24 .. -27).  The reference
    pre-emphasises, frames and transforms in fp64; so does the default route since ABI 46 (dlip_powspec_wave_fft64_f32), and EVERY element
    of the un-normalised features is inside 1e-4 of the feature scale.  The fp32 routes are not (0.3 .. 1 % in those elements' energy):
    kept as second implementations, they agree with the default on everything above the floor."""
    from deeplip_amd.frontend import AudioFrontend
    S, B = 56146, 6
    t = np.arange(S) / 16000.0
    r = np.random.Generator(np.random.PCG64(11))
    sig = np.stack([0.3 * np.sin(2 * np.pi * (150 + 170 * b) * t) + 0.05 * r.standard_normal(S) for b in range(B)]).astype(np.float32)
    x = torch.from_numpy(sig).cuda()
    out = {d: AudioFrontend("logfbank", num_bin=60, normalize=False, dft=d)(x).cpu().numpy() for d in ("fft64", "gemm32", "direct64")}
    assert AudioFrontend("logfbank", num_bin=60).dft == "fft64"
    n_floor = 0
    for b in range(B):
        ref = O.audio_features(sig[b].astype(np.float64), "logfbank", nfilt=60, normalize=False)
        assert out["fft64"][b].shape == ref.shape
        scale = np.abs(ref).max()
        assert np.abs(out["fft64"][b] - ref).max() < 1e-4 * scale, b                      # every element, floor included
        floor = ref < -20.0
        n_floor += int(floor.sum())
        for d in ("gemm32", "direct64"):
            assert (np.abs(out[d][b] - ref) * ~floor).max() < 1e-4 * scale, (d, b)
    assert n_floor >= 8                                                                  # (the case really has such elements: 11 of 126 000)
    # CMVN-ed MFCCs + deltas through the default route, against the oracle
    fe = AudioFrontend("mfcc", delta=True)
    y = fe(x).cpu().numpy()
    for b in range(B):
        ref = O.audio_features(sig[b].astype(np.float64), "mfcc", delta=True)
        assert rel_err(y[b], ref) < 1e-4, b

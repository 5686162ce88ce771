"""Range guard of the split-fp16 ("f16x3") arithmetic.  The reference is fp32 end to end (model.py:82-85); the split
activation format stores hi + lo fp16, so |v| >= 65520 cannot be represented.  Every producer of split values reports
such a value to a status block the Python layer turns into DeepLipRangeError; the recourse is the exact "f32" mode."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return torch.randn(*shape, generator=g) * scale


@pytest.fixture(autouse=True)
def _clean_status():
    from deeplip_amd import _lib
    torch.cuda.synchronize()
    _lib.status_words().zero_()
    yield
    torch.cuda.synchronize()
    _lib.status_words().zero_()


def test_split_pack_reports_overflow_and_clears():
    from deeplip_amd import _lib, ops
    x = rnd(4, 8, 64, seed=1).cuda()
    ops.split_pack(x)
    _lib.check_range(sync=True)                     # in range: silent
    x[2, 3, 17] = 7.0e4
    ops.split_pack(x)
    with pytest.raises(_lib.DeepLipRangeError, match="split_pack"):
        _lib.check_range(sync=True)
    _lib.check_range(sync=True)                     # the report was consumed
    x[2, 3, 17] = float("inf")
    ops.nct_to_ntc(x.permute(0, 2, 1).contiguous(), pad_to=64, out_split=True)
    with pytest.raises(_lib.DeepLipRangeError):
        _lib.check_range(sync=True)


def _chain(ops, x, ws, scs, bs, split):
    h = ops.split_pack(x) if split else x
    for i, (w, sc, b) in enumerate(zip(ws, scs, bs)):
        last = i == len(ws) - 1
        h = ops.conv_nhwc(h, w, b, pad=(1, 1), w_scale=sc, x_split=split, out_split=split and not last)
    return h


@pytest.mark.parametrize("scale,gain", [(1.0, 1.0), (1.0e3, 1.0), (1.0e4, 8.0), (1.0e5, 1.0), (1.0e-3, 1.0), (1.0e-6, 1.0)])
def test_three_layer_chain_across_magnitudes(scale, gain):
    """A 3-layer 3x3 chain (64 channels, unit-gain weights) with the activations scaled by `scale`, against fp64.
    1e3: in range, fp32-grade.  1e4 with a gain-8 first layer: the INPUT fits, the first layer's output does not -> the
    conv epilogue reports it; 1e5: the input itself does not fit -> split_pack reports it (loud, never a silent inf/NaN).
    1e-3 / 1e-6: no error is raised -- hi / lo become fp16 subnormals (quantum 2^-24 = 6e-8), so the ABSOLUTE error
    stays ~1e-7 while the relative error grows; the measured numbers are what DESIGN.md quotes."""
    from deeplip_amd import _lib, ops, packing
    x = rnd(3, 12, 12, 64, seed=5) * scale
    ws64 = [rnd(64, 3, 3, 64, seed=6 + i, scale=(gain if i == 0 else 1.0) / np.sqrt(576)).double() for i in range(3)]
    bs = [torch.zeros(64) for _ in range(3)]
    ref = x.double().permute(0, 3, 1, 2)
    for w in ws64:
        ref = F.conv2d(ref, w.permute(0, 3, 1, 2), None, padding=1)
    ref = ref.permute(0, 2, 3, 1)
    packed = [packing.split_weights(w) for w in ws64]
    args = ([p[0].cuda() for p in packed], [p[1].cuda() for p in packed], [b.cuda() for b in bs])
    y = _chain(ops, x.cuda(), *args, split=True)
    if scale >= 1.0e4:
        with pytest.raises(_lib.DeepLipRangeError, match="convolution" if scale < 1.0e5 else "split_pack"):
            _lib.check_range(sync=True)
        return
    _lib.check_range(sync=True)
    err = rel_err(y.cpu().numpy(), ref.numpy())
    print(f"\nf16x3 3-layer chain, activation scale {scale:g}: max|err| / max|ref| = {err:.3e}")
    bound = {1.0: 2e-6, 1.0e3: 2e-6, 1.0e-3: 2e-4, 1.0e-6: 0.2}[scale]
    assert err < bound


def test_bn_gamma_50_checkpoint_is_loud_in_f16x3_and_exact_in_f32():
    """A checkpoint whose BatchNorm gammas are 50 (activations x50 per layer): f16x3 overflows after a few layers and
    says so; the documented recourse -- the same model packed in f32 mode -- matches the CPU oracle."""
    from deeplip_amd import _lib, packing, weightgen as wg
    from models.audio_models.tdnn import SpeakerEmbNet
    from oracle import deeplip_oracle as O
    opts = {"arch": "tdnn", "tdnn": {"input_dim": 24, "hidden_dim": [512] * 4 + [1500], "context": O.TDNN_CONTEXT, "tdnn_layers": 5,
                                      "embedding_dim": 512, "pooling": "statistic", "attention_hidden_size": 64, "bn_first": True}}
    net = SpeakerEmbNet(opts)
    sd = wg.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, prefix="audio_tdnn.")
    one = {k: v.copy() for k, v in sd.items()}
    for i in range(5):
        sd[f"tdnn.{i}.bn.weight"] = np.full_like(sd[f"tdnn.{i}.bn.weight"], 50.0)
    one["tdnn.1.bn.weight"] = np.full_like(one["tdnn.1.bn.weight"], 50.0)
    x = torch.from_numpy(wg.audio_input(3, 24, 200))
    try:
        # gamma = 50 in ONE layer: still inside fp16 range -> f16x3 is silent and right
        packing.set_precision("f16x3")
        net.load_state_dict({k: torch.from_numpy(v) for k, v in one.items()})
        net.eval().cuda()
        xv = net.extract_embedding(x.cuda())[0]
        _lib.check_range(sync=True)
        with torch.no_grad():
            rxv, _ = O.speaker_extract_embedding(O.to_torch_sd(one), x, O.TDNN_CONTEXT)
        assert rel_err(xv.cpu().numpy(), rxv.numpy()) < 1e-4
        # gamma = 50 in all five: 50^5 -> overflow, reported
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        net.extract_embedding(x.cuda())
        with pytest.raises(_lib.DeepLipRangeError):
            _lib.check_range(sync=True)
        # recourse: exact mode, same engine
        packing.set_precision("f32")
        xv = net.extract_embedding(x.cuda())[0]
        _lib.check_range(sync=True)
        with torch.no_grad():
            rxv, _ = O.speaker_extract_embedding(O.to_torch_sd(sd), x, O.TDNN_CONTEXT)
        assert torch.isfinite(xv).all()
        assert rel_err(xv.cpu().numpy(), rxv.numpy()) < 1e-4
    finally:
        packing.set_precision("f32")


def test_next_forward_raises_without_an_explicit_check():
    """The models look at the status block (a host memory read, no synchronisation) when a forward starts: an overflow
    of an earlier, completed forward cannot go unnoticed."""
    from deeplip_amd import _lib, ops, packing, weightgen as wg
    from models.audio_models.tdnn import SpeakerEmbNet
    from oracle import deeplip_oracle as O
    opts = {"arch": "tdnn", "tdnn": {"input_dim": 24, "hidden_dim": [512] * 4 + [1500], "context": O.TDNN_CONTEXT, "tdnn_layers": 5,
                                      "embedding_dim": 512, "pooling": "statistic", "attention_hidden_size": 64, "bn_first": True}}
    try:
        packing.set_precision("f16x3")
        net = SpeakerEmbNet(opts)
        sd = wg.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, prefix="audio_tdnn.")
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        net.eval().cuda()
        x = torch.from_numpy(wg.audio_input(2, 24, 150)).cuda()
        net.extract_embedding(x * 1.0e6)           # input beyond fp16 range
        torch.cuda.synchronize()
        with pytest.raises(_lib.DeepLipRangeError):
            net.extract_embedding(x)
        net.extract_embedding(x)                   # consumed: the next one runs
        _lib.check_range(sync=True)
    finally:
        packing.set_precision("f32")

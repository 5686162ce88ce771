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


# ------------------------------------------------------------------------------------------------------------------
# The LOW side.  lo = v - hi is a normal fp16 number only for |v| >= 2^-3; below it sits on the subnormal grid 2^-24, an
# absolute error of ~3e-8 per element = 3e-8 / max|v| of the tensor's scale.  Inside a range scope (every eval forward of
# the models, every step plan) a produced tensor whose largest magnitude lies in (0, 2^-2) is reported (2^-6 until round 6: dlip_common.h says why).
# ------------------------------------------------------------------------------------------------------------------
def test_low_side_scope_reports_a_tensor_below_2_to_minus_2():
    from deeplip_amd import _lib, ops
    x = rnd(4, 8, 64, seed=2).cuda()
    with _lib.range_scope():
        ops.split_pack(x)                            # max|v| ~ 4: fine
        ops.split_pack(torch.zeros_like(x))          # all zero: exact in any format, not a report
        y = x * 1e-4
        y[1, 2, 3] = 0.5                             # ONE element above the line: the tensor's scale is fine
        ops.split_pack(y)
    _lib.check_range(sync=True)
    with _lib.range_scope():
        ops.split_pack(x * 0.1)                      # max|v| ~ 0.35: just above the line
    _lib.check_range(sync=True)
    with _lib.range_scope():
        ops.split_pack(x * 2e-2)                     # max|v| ~ 7e-2 < 2^-2
    with pytest.raises(_lib.DeepLipRangeError, match="below 2\\^-2"):
        _lib.check_range(sync=True)
    _lib.check_range(sync=True)                      # consumed
    ops.split_pack(x * 2e-2)                         # outside a scope the low side is not guarded (documented)
    _lib.check_range(sync=True)
    with _lib.range_scope():                         # the scope's words were re-zeroed by its verdict: a clean scope stays clean
        ops.split_pack(x)
    _lib.check_range(sync=True)


@pytest.mark.parametrize("scale", [1.0, 1e-3, 1e-6])
def test_three_layer_chain_inside_a_scope_is_fp32_grade_or_raises(scale):
    """The chain of test_three_layer_chain_across_magnitudes inside a scope: scale 1 is silent and fp32-grade; at 1e-3 and 1e-6
    (silent 2.8e-5 and 3.1e-2 outside a scope) the producers report."""
    from deeplip_amd import _lib, ops, packing
    x = rnd(3, 12, 12, 64, seed=5) * scale
    ws64 = [rnd(64, 3, 3, 64, seed=6 + i, scale=1.0 / np.sqrt(576)).double() for i in range(3)]
    packed = [packing.split_weights(w) for w in ws64]
    args = ([p[0].cuda() for p in packed], [p[1].cuda() for p in packed], [torch.zeros(64).cuda() for _ in range(3)])
    with _lib.range_scope():
        y = _chain(ops, x.cuda(), *args, split=True)
    if scale < 1.0:
        with pytest.raises(_lib.DeepLipRangeError, match="below 2\\^-2"):
            _lib.check_range(sync=True)
        return
    _lib.check_range(sync=True)
    ref = x.double().permute(0, 3, 1, 2)
    for w in ws64:
        ref = F.conv2d(ref, w.permute(0, 3, 1, 2), None, padding=1)
    assert rel_err(y.cpu().numpy(), ref.permute(0, 2, 3, 1).numpy()) < 2e-6


def _meets_bar_or_raises(run, oracle, what):
    """The contract of the f16x3 mode on ANY checkpoint: a forward either meets the north star's element-wise 1e-4 bar against
    the oracle, or DeepLipRangeError is raised (and the exact f32 packing then meets the bar).  Returns which happened."""
    from conftest import assert_close_rel
    from deeplip_amd import _lib, packing
    want = oracle()
    try:
        packing.set_precision("f16x3")
        try:
            got = run()
            _lib.check_range(sync=True)
        except _lib.DeepLipRangeError as ex:
            side = "low" if "below 2^-2" in str(ex) else "high"
            packing.set_precision("f32")
            got = run()
            _lib.check_range(sync=True)
            assert_close_rel(got.cpu().numpy(), want.numpy(), rtol=1e-4, what=what + " (f32 recourse)")
            return f"raised ({side} side)"
        assert_close_rel(got.cpu().numpy(), want.numpy(), rtol=1e-4, what=what + " (f16x3)")
        return "met"
    finally:
        packing.set_precision("f32")


def _tdnn(prefix="audio_tdnn."):
    from deeplip_amd import weightgen as wg
    from models.audio_models.tdnn import SpeakerEmbNet
    from oracle import deeplip_oracle as O
    opts = {"arch": "tdnn", "tdnn": {"input_dim": 24, "hidden_dim": [512] * 4 + [1500], "context": O.TDNN_CONTEXT, "tdnn_layers": 5,
                                      "embedding_dim": 512, "pooling": "statistic", "attention_hidden_size": 64, "bn_first": True}}
    net = SpeakerEmbNet(opts)
    return net, wg.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, prefix=prefix)


@pytest.mark.parametrize("layers", [[1], [0, 1, 2, 3, 4]], ids=["gamma1e-3-in-one-layer", "gamma1e-3-in-all-layers"])
def test_small_bn_gamma_checkpoint_meets_the_bar_or_raises(layers):
    """Checkpoint-like statistics on the SMALL side (a BatchNorm whose gammas collapsed to 1e-3): activations behind such a
    layer are ~1e-3 of their usual scale.  No silent path: fp32-grade, or an error."""
    from deeplip_amd import weightgen as wg
    from oracle import deeplip_oracle as O
    net, sd = _tdnn()
    for i in layers:
        sd[f"tdnn.{i}.bn.weight"] = np.full_like(sd[f"tdnn.{i}.bn.weight"], 1.0e-3)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.eval().cuda()
    x = torch.from_numpy(wg.audio_input(3, 24, 200))

    def oracle():
        with torch.no_grad():
            return O.speaker_extract_embedding(O.to_torch_sd(sd), x, O.TDNN_CONTEXT)[0]

    outcome = _meets_bar_or_raises(lambda: net.extract_embedding(x.cuda())[0], oracle, f"x-vector, gamma 1e-3 in layers {layers}")
    print(f"\nBN gamma = 1e-3 in TDNN layers {layers}: f16x3 {outcome}")
    # (with gamma = 1e-3 everywhere the activations are the BatchNorm betas plus a 1e-3 ripple: O(0.1), inside the window)


def _lipreading():
    from deeplip_amd import weightgen as wg
    from models.video_models.model import Lipreading
    tcn = {"num_layers": 4, "kernel_size": [3, 5, 7], "dropout": 0.2, "dwpw": False, "width_mult": 1}
    net = Lipreading(hidden_dim=256, backbone_type="resnet", num_classes=54, relu_type="prelu", tcn_options=tcn, width_mult=1.0,
                     extract_feats=True)
    return net, wg.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, prefix="video.")


def test_stem_running_var_1e3_meets_the_bar_or_raises():
    """frontend3D's BatchNorm with running_var = 1e3 (its output ~1/32 of the usual scale, model.py:82-85)."""
    from deeplip_amd import weightgen as wg
    from oracle import deeplip_oracle as O
    net, sd = _lipreading()
    sd["frontend3D.1.running_var"] = np.full_like(sd["frontend3D.1.running_var"], 1.0e3)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.eval().cuda()
    x = torch.from_numpy(wg.video_input(2, frames=9, key="range.video"))

    def oracle():
        with torch.no_grad():
            return O.lipreading_features(O.to_torch_sd(sd), x)

    outcome = _meets_bar_or_raises(lambda: net(x.cuda(), None), oracle, "features, stem running_var 1e3")
    print(f"\nstem running_var = 1e3: f16x3 {outcome}")


@pytest.mark.parametrize("lo,hi,seed", [(1e-2, 10.0, 0), (1e-2, 10.0, 1), (1e-2, 3.0, 0), (1e-2, 1.0, 0)])
def test_trunk_gamma_log_uniform_meets_the_bar_or_raises(lo, hi, seed):
    """Every BatchNorm gamma of the ResNet trunk drawn log-uniformly from [lo, hi] per channel (resnet.py:28-69): channels orders
    of magnitude apart inside one tensor, layer after layer.  The per-layer gain is the RMS gamma: 2.7 for [1e-2, 10] (17 layers:
    beyond fp16's range -> the high side reports), 0.89 for [1e-2, 3], 0.33 for [1e-2, 1] (1e-8 after 17 layers -> the low side)."""
    from deeplip_amd import weightgen as wg
    from oracle import deeplip_oracle as O
    net, sd = _lipreading()
    r = np.random.Generator(np.random.PCG64(100 + seed))
    for k in sd:
        if k.startswith("trunk.") and (".bn" in k or "downsample.1" in k) and k.endswith(".weight"):
            sd[k] = np.exp(r.uniform(np.log(lo), np.log(hi), sd[k].shape)).astype(np.float32)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.eval().cuda()
    x = torch.from_numpy(wg.video_input(2, frames=9, key="range.video2"))

    def oracle():
        with torch.no_grad():
            return O.lipreading_features(O.to_torch_sd(sd), x)

    outcome = _meets_bar_or_raises(lambda: net(x.cuda(), None), oracle, f"features, trunk gamma log-uniform [{lo}, {hi}] seed {seed}")
    print(f"\ntrunk gamma log-uniform [{lo:g}, {hi:g}] (seed {seed}): f16x3 {outcome}")


def test_plan_replay_reports_an_under_range_batch():
    """The verdict kernel is the last launch of a recorded step: a replay on inputs 1e-4 of the recorded scale reports, and the
    report surfaces at the next run() (or at close())."""
    from deeplip_amd import _lib, packing, weightgen as wg
    from deeplip_amd.plan import StepPlan
    try:
        packing.set_precision("f16x3")
        net, sd = _tdnn()
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        net.eval().cuda()
        x = torch.from_numpy(wg.audio_input(2, 24, 150)).cuda()
        plan = StepPlan(lambda a: net.extract_embedding(a)[0], x.clone())
        plan.run(); plan.run()
        _lib.check_range(sync=True)
        plan(x * 1.0e-4)                            # CMVN-normalised features never look like this
        torch.cuda.synchronize()
        with pytest.raises(_lib.DeepLipRangeError, match="below 2\\^-2"):
            plan.run()
        plan(x)
        plan.close()                                # clean again
    finally:
        packing.set_precision("f32")


def test_a_scope_that_runs_out_of_slots_ends_with_an_error():
    """dlip_range_scope_end: more split-producing launches than slots -> DLIP_ERANGE (the later launches ran unguarded; never quietly)."""
    from deeplip_amd import _lib, ops
    x = rnd(4, 8, 64, seed=3).cuda()
    slots = torch.zeros(_lib._EVID_WORDS * 2, dtype=torch.int32, device="cuda")          # a scope of two slots
    with pytest.raises(_lib.DeepLipHipError):
        with _lib.range_scope(slots):
            for _ in range(3):
                ops.split_pack(x)
    with _lib.range_scope(slots):                                                          # two launches fit
        ops.split_pack(x)
        ops.split_pack(x)
    _lib.check_range(sync=True)


def test_stem_reports_a_clip_below_the_line_even_when_its_output_is_ordinary():
    """The fused stem + pool entry point splits TWO tensors (the clip in its pre-pass, the pooled output in its epilogue): each has its
    evidence slot.  A float clip with a gain of 2^-14 behind an ordinary BatchNorm gives an ordinary output (the BatchNorm's shift) --
    and is reported for its input (until round 6 the output's evidence vouched for it)."""
    from deeplip_amd import _lib, arith, weightgen as wg
    net, sd = _lipreading()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.eval().cuda()
    x = torch.from_numpy(wg.video_input(2, frames=9, key="range.stem.low"))
    arith.configure("f16x3")
    try:
        net(x.cuda(), None)
        _lib.check_range(sync=True)                                                        # an ordinary clip: nothing
        with pytest.raises(_lib.DeepLipRangeError, match="below 2\\^-2"):
            net((x * float(2.0 ** -14)).cuda(), None)
            _lib.check_range(sync=True)
    finally:
        arith.configure("f32")

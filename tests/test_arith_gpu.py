"""The arithmetic switch (deeplip_amd/arith.py) on the GPU: mode ``auto`` = the f16x3 arithmetic, with what leaves its range computed
again on the exact f32 pack of the SAME model in the SAME process -- an eager call, and the offending batch of a recorded pipeline
(per-plan status blocks, dlip_status_scope: ABI 45).  The reference computes in fp32 end to end (train_fusion.py:338-358), so a row
must never come back wrong because the fast arithmetic could not hold it."""
import numpy as np
import pytest
import torch

from conftest import assert_close_rel

pytestmark = pytest.mark.gpu


def _tdnn():
    from deeplip_amd import weightgen as wg
    from models.audio_models.tdnn import SpeakerEmbNet
    from oracle import deeplip_oracle as O
    opts = {"arch": "tdnn", "tdnn": {"input_dim": 24, "hidden_dim": [512] * 4 + [1500], "context": O.TDNN_CONTEXT, "tdnn_layers": 5,
                                      "embedding_dim": 512, "pooling": "statistic", "attention_hidden_size": 64, "bn_first": True}}
    net = SpeakerEmbNet(opts)
    sd = wg.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, prefix="audio_tdnn.")
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return net.eval().cuda(), sd


def _oracle_rows(sd, x):
    from oracle import deeplip_oracle as O
    with torch.no_grad():
        return O.speaker_extract_embedding(O.to_torch_sd(sd), x, O.TDNN_CONTEXT)[0].numpy()


def test_auto_eager_call_is_computed_again_in_f32(monkeypatch):
    """An under-range input (1e-4 of what CMVN-normalised features look like) through the eager entry point: f16x3 raises,
    auto returns oracle-grade rows and counts the re-run; an in-range call afterwards runs f16x3 again and counts nothing.
    (Repair only, arith.CALIBRATE off: one odd batch between normal ones must not re-scale the model; the calibration has its own tests.)"""
    from deeplip_amd import _lib, arith, weightgen as wg
    monkeypatch.setattr(arith, "CALIBRATE", False)
    net, sd = _tdnn()
    x = torch.from_numpy(wg.audio_input(3, 24, 200))
    small = x * 1.0e-4
    arith.configure("f16x3")
    with pytest.raises(_lib.DeepLipRangeError):
        net.extract_embedding(small.cuda())
        _lib.check_range(sync=True)
    arith.configure("auto")
    n0 = arith.STATS["f32_reruns"]
    got = net.extract_embedding(small.cuda())[0]
    assert arith.STATS["f32_reruns"] == n0 + 1 and "SpeakerEmbNet" in arith.STATS["last"]["what"]
    assert_close_rel(got.cpu().numpy(), _oracle_rows(sd, small), rtol=1e-4, what="auto: under-range batch")
    got = net.extract_embedding(x.cuda())[0]
    assert arith.STATS["f32_reruns"] == n0 + 1                      # in range: the fast arithmetic, no re-run
    assert_close_rel(got.cpu().numpy(), _oracle_rows(sd, x), rtol=1e-4, what="auto: in-range batch")
    from deeplip_amd import packing
    assert packing.PRECISION == "f16x3"                              # the exact pack was borrowed for the re-run only


def test_auto_video_embed_is_computed_again_in_f32():
    """The lip-clip encoder on a checkpoint whose trunk BatchNorms shrink the activations below 2^-2 (resnet.py:28-69)."""
    from deeplip_amd import arith, weightgen as wg
    from models.video_models.model import Lipreading
    from oracle import deeplip_oracle as O
    tcn = {"num_layers": 4, "kernel_size": [3, 5, 7], "dropout": 0.2, "dwpw": False, "width_mult": 1}
    net = Lipreading(num_classes=54, relu_type="prelu", tcn_options=tcn, extract_feats=True)
    sd = wg.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, prefix="video.")
    for k in sd:                # every BatchNorm gamma of the trunk at 0.05: the activations shrink layer after layer, far below 2^-2
        if k.startswith("trunk.") and (".bn" in k or "downsample.1" in k) and k.endswith(".weight"):
            sd[k] = np.full_like(sd[k], 0.05)
        if k.startswith("trunk.") and (".bn" in k or "downsample.1" in k) and k.endswith(".bias"):
            sd[k] = np.zeros_like(sd[k])
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.eval().cuda()
    x = torch.from_numpy(wg.video_input(2, frames=9, key="arith.video"))
    with torch.no_grad():
        want = O.video_time_mean(O.lipreading_features(O.to_torch_sd(sd), x)).numpy()
    arith.configure("auto")
    n0 = arith.STATS["f32_reruns"]
    got = net.embed(x.cuda())
    assert arith.STATS["f32_reruns"] == n0 + 1 and "Lipreading" in arith.STATS["last"]["what"]
    assert_close_rel(got.cpu().numpy(), want, rtol=1e-4, what="auto: lip-clip embed re-run in f32")


@pytest.mark.parametrize("mode", ["auto", "f16x3"])
def test_pipeline_repairs_exactly_the_offending_batch(mode, monkeypatch):
    """Five batches through an ExtractPipeline, the third one under-range.  auto: every row oracle-grade, ONE re-run, the other
    batches untouched by it (same bits as a run without the bad batch); f16x3: DeepLipRangeError by finish()."""
    from deeplip_amd import _lib, arith, weightgen as wg
    from deeplip_amd.pipeline import ExtractPipeline, pin
    net, sd = _tdnn()
    arith.configure(mode)
    monkeypatch.setattr(arith, "CALIBRATE", False)      # (repair only: ONE odd batch among normal ones must not re-scale the model)
    B = 4
    xs = [torch.from_numpy(wg.audio_input(B, 24, 160, key=f"arith.pipe{i}")) for i in range(5)]
    xs[2] = xs[2] * 1.0e-4
    with torch.no_grad():
        pipe = ExtractPipeline(lambda a: net.extract_embedding(a)[0], xs[0].cuda())
    table = torch.zeros((5 * B, 512), device="cuda")
    n0 = arith.STATS["f32_reruns"]
    if mode == "f16x3":
        with pytest.raises(_lib.DeepLipRangeError):
            pipe.run([(pin(x),) for x in xs], table)
            pipe.finish()
        pipe.close()
        return
    pipe.run([(pin(x),) for x in xs], table)
    pipe.finish()
    assert pipe.reruns == 1 and arith.STATS["f32_reruns"] == n0 + 1
    want = np.concatenate([_oracle_rows(sd, x) for x in xs])
    assert_close_rel(table.cpu().numpy(), want, rtol=1e-4, what="pipeline rows, one batch repaired")
    # the in-range batches ran the fast arithmetic: bit-identical to a second pass without the bad batch
    t2 = torch.zeros((4 * B, 512), device="cuda")
    pipe.run([(pin(x),) for i, x in enumerate(xs) if i != 2], t2)
    pipe.finish()
    assert pipe.reruns == 1
    keep = torch.cat([table[:2 * B], table[3 * B:]])
    assert torch.equal(keep, t2)
    pipe.close()
    _lib.check_range(sync=True)                                      # nothing left behind in any status block


def test_plan_status_block_is_its_own():
    """dlip_status_scope: a recorded plan reports to ITS block -- two plans side by side, only the one that replayed the bad batch
    holds a report, and the process-wide check still sees it (nothing reported anywhere goes unseen)."""
    from deeplip_amd import _lib, arith, weightgen as wg
    from deeplip_amd.plan import StepPlan
    net, _ = _tdnn()
    arith.configure("f16x3")
    x = torch.from_numpy(wg.audio_input(2, 24, 150)).cuda()
    pa = StepPlan(lambda a: net.extract_embedding(a)[0], x.clone())
    pb = StepPlan(lambda a: net.extract_embedding(a)[0], x.clone())
    pa.take_range_error(); pb.take_range_error()
    pa(x * 1.0e-4)
    pb(x)
    torch.cuda.synchronize()
    assert pb.take_range_error() is None
    err = pa.take_range_error()
    assert isinstance(err, _lib.DeepLipRangeError) and "below 2^-2" in str(err)
    assert pa.take_range_error() is None                             # taken = cleared
    pa(x * 1.0e-4)
    torch.cuda.synchronize()
    with pytest.raises(_lib.DeepLipRangeError):
        _lib.check_range()
    pa.close(); pb.close()


@pytest.mark.parametrize("gain_log2", [-20, -10, 14, 20])
def test_calibration_moves_an_out_of_range_input_gain_onto_the_fast_path(gain_log2):
    """Features with a gain of 2^-20 .. 2^+20 (a front-end without CMVN, a different scaling convention): the FIRST batch leaves the
    f16x3 range, is computed again in exact f32 -- and that pass calibrates activation exponents (powers of two folded into the f16x3
    pack, packing.act_exponents).  The next batches of the same kind run the fast arithmetic, no re-run, and meet the 1e-4 bar."""
    from deeplip_amd import arith, packing, weightgen as wg
    net, sd = _tdnn()
    arith.configure("auto")
    g = float(2.0 ** gain_log2)
    xs = [torch.from_numpy(wg.audio_input(3, 24, 200, key=f"arith.gain{i}")) * g for i in range(3)]
    n0, c0 = arith.STATS["f32_reruns"], arith.STATS["calibrations"]
    got = net.extract_embedding(xs[0].cuda())[0]
    assert arith.STATS["f32_reruns"] == n0 + 1 and arith.STATS["calibrations"] == c0 + 1
    assert_close_rel(got.cpu().numpy(), _oracle_rows(sd, xs[0]), rtol=1e-4, what="first batch (exact re-run)")
    e = packing.act_exponents(net)
    assert e and "in" in e and (e["in"] > 0) == (gain_log2 < 0)
    for x in xs[1:]:
        got = net.extract_embedding(x.cuda())[0]
        assert_close_rel(got.cpu().numpy(), _oracle_rows(sd, x), rtol=1e-4, what=f"gain 2^{gain_log2}: calibrated f16x3 path")
    assert arith.STATS["f32_reruns"] == n0 + 1                       # ... without another exact re-run
    # new weights void the calibration
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    assert packing.act_exponents(net) == {}


def test_calibration_of_the_lip_clip_trunk_respects_the_residual_groups():
    """A checkpoint whose trunk BatchNorms shrink the activations layer after layer: after the calibrating re-run the model embeds on
    the f16x3 path; tensors that meet in a residual addition share their exponent (resnet.py:62-69: out += residual)."""
    from deeplip_amd import arith, packing, weightgen as wg
    from models.video_models.model import Lipreading
    from oracle import deeplip_oracle as O
    tcn = {"num_layers": 4, "kernel_size": [3, 5, 7], "dropout": 0.2, "dwpw": False, "width_mult": 1}
    net = Lipreading(num_classes=54, relu_type="prelu", tcn_options=tcn, extract_feats=True)
    sd = wg.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, prefix="video.")
    for k in sd:
        if k.startswith("trunk.") and (".bn" in k or "downsample.1" in k) and k.endswith(".weight"):
            sd[k] = np.full_like(sd[k], 0.05)
        if k.startswith("trunk.") and (".bn" in k or "downsample.1" in k) and k.endswith(".bias"):
            sd[k] = np.zeros_like(sd[k])
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.eval().cuda()
    arith.configure("auto")
    xs = [torch.from_numpy(wg.video_input(2, frames=29, key=f"arith.vcal{i}")) for i in range(2)]      # (29 frames: the pooled epilogue serves embed())
    n0 = arith.STATS["f32_reruns"]
    net.embed(xs[0].cuda())
    assert arith.STATS["f32_reruns"] == n0 + 1
    e = packing.act_exponents(net)
    assert e and max(e.values()) > 14
    for grp in net.act_exponent_groups():
        assert len({e.get(n, 0) for n in grp}) == 1, (grp, e)
    with torch.no_grad():
        want = O.video_time_mean(O.lipreading_features(O.to_torch_sd(sd), xs[1])).numpy()
        feats = O.lipreading_features(O.to_torch_sd(sd), xs[1]).numpy()
    got = net.embed(xs[1].cuda())
    assert arith.STATS["f32_reruns"] == n0 + 1                       # the second clip batch: fast path, no re-run
    assert_close_rel(got.cpu().numpy(), want, rtol=1e-4, what="calibrated f16x3 lip-clip embed")
    assert_close_rel(net(xs[1].cuda(), None).cpu().numpy(), feats, rtol=1e-4, what="calibrated f16x3 features [B,T,512]")
    # the fused row: the pooled sums reach the z-norm still carrying the pack's output exponent -- invariant under it, bit for bit
    from deeplip_amd import fusion, ops
    xa = torch.randn(2, 512, generator=torch.Generator().manual_seed(1)).cuda()
    pooled = net.embed(xs[1].cuda(), finish=False)
    assert isinstance(pooled, ops.Pooled)
    assert torch.equal(fusion.fuse_av(xa, pooled), fusion.fuse_av(xa, net.embed(xs[1].cuda())))
    assert arith.STATS["f32_reruns"] == n0 + 1


def test_pipeline_re_records_its_plans_after_a_calibration():
    """Six batches of features with a gain of 2^-18 through an ExtractPipeline under auto: the first batches are repaired in exact f32
    (one of those passes calibrates the model), the pipeline re-records its plans on the rebuilt f16x3 pack and the rest of the list
    runs the fast arithmetic; every row oracle-grade."""
    from deeplip_amd import arith, weightgen as wg
    from deeplip_amd.pipeline import ExtractPipeline, pin
    net, sd = _tdnn()
    arith.configure("auto")
    B, g = 4, float(2.0 ** -18)
    xs = [torch.from_numpy(wg.audio_input(B, 24, 160, key=f"arith.pcal{i}")) * g for i in range(6)]
    with torch.no_grad():
        pipe = ExtractPipeline(lambda a: net.extract_embedding(a)[0], xs[0].cuda())
    table = torch.zeros((6 * B, 512), device="cuda")
    pipe.run([(pin(x),) for x in xs], table)
    pipe.finish()
    assert 1 <= pipe.reruns <= 3 and pipe.rerecorded >= 1
    want = np.concatenate([_oracle_rows(sd, x) for x in xs])
    assert_close_rel(table.cpu().numpy(), want, rtol=1e-4, what="pipeline rows across a calibration")
    r0 = pipe.reruns
    pipe.run([(pin(x),) for x in xs], table)                         # a second pass over the list: nothing left to repair
    pipe.finish()
    assert pipe.reruns == r0
    assert_close_rel(table.cpu().numpy(), want, rtol=1e-4, what="second pass")
    pipe.close()


@pytest.mark.parametrize("case", ["tdnn-gamma-1e-3", "stem-var-1e3", "trunk-gamma-1e-2..10", "trunk-gamma-1e-2..1"])
def test_checkpoint_like_statistics_meet_the_bar_under_auto(case):
    """tests/test_range_gpu.py holds the f16x3 mode to "meets the 1e-4 bar OR raises" on checkpoints whose BatchNorm statistics push
    activations out of the split format.  Under ``auto`` the same checkpoints simply MEET THE BAR: in range -> f16x3; out of range ->
    the first call is computed in exact f32 and calibrates activation exponents, the second call runs f16x3 on the re-scaled pack (or
    is repaired again if the calibration budget is spent) -- never an error, never a wrong row."""
    from deeplip_amd import arith, weightgen as wg
    from oracle import deeplip_oracle as O
    import test_range_gpu as R
    arith.configure("auto")
    if case.startswith("tdnn"):
        net, sd = R._tdnn()
        for i in range(5):
            sd[f"tdnn.{i}.bn.weight"] = np.full_like(sd[f"tdnn.{i}.bn.weight"], 1.0e-3)
            sd[f"tdnn.{i}.bn.bias"] = np.zeros_like(sd[f"tdnn.{i}.bn.bias"])
        xs = [torch.from_numpy(wg.audio_input(3, 24, 200, key=f"auto.r{i}")) for i in range(2)]
        run = lambda x: net.extract_embedding(x.cuda())[0]
        oracle = lambda x: O.speaker_extract_embedding(O.to_torch_sd(sd), x, O.TDNN_CONTEXT)[0]
    else:
        net, sd = R._lipreading()
        if case == "stem-var-1e3":
            sd["frontend3D.1.running_var"] = np.full_like(sd["frontend3D.1.running_var"], 1.0e3)
        else:
            lo, hi = (1e-2, 10.0) if case.endswith("10") else (1e-2, 1.0)
            r = np.random.Generator(np.random.PCG64(100))
            for k in sd:
                if k.startswith("trunk.") and (".bn" in k or "downsample.1" in k) and k.endswith(".weight"):
                    sd[k] = np.exp(r.uniform(np.log(lo), np.log(hi), sd[k].shape)).astype(np.float32)
        xs = [torch.from_numpy(wg.video_input(2, frames=9, key=f"auto.rv{i}")) for i in range(2)]
        run = lambda x: net(x.cuda(), None)
        oracle = lambda x: O.lipreading_features(O.to_torch_sd(sd), x)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.eval().cuda()
    n0 = arith.STATS["f32_reruns"]
    for i, x in enumerate(xs):
        with torch.no_grad():
            want = oracle(x).numpy()
        assert_close_rel(run(x).cpu().numpy(), want, rtol=1e-4, what=f"{case}, call {i}")
    print(f"\n{case}: {arith.STATS['f32_reruns'] - n0} exact re-run(s) over 2 calls, exponents {__import__('deeplip_amd.packing', fromlist=['x']).act_exponents(net)}")


def test_explicit_calibration_lets_plain_f16x3_run_an_out_of_range_input():
    """arith.calibrate(fn, batch): under plain ``f16x3`` (which raises on a range report, it never re-runs) a model calibrated on one
    representative batch takes the same kind of input without an exception, at the bar."""
    from deeplip_amd import _lib, arith, packing, weightgen as wg
    net, sd = _tdnn()
    arith.configure("f16x3")
    g = float(2.0 ** -16)
    xs = [torch.from_numpy(wg.audio_input(3, 24, 200, key=f"arith.ecal{i}")) * g for i in range(2)]
    with pytest.raises(_lib.DeepLipRangeError):
        net.extract_embedding(xs[0].cuda())
        _lib.check_range(sync=True)
    exact = arith.calibrate(lambda x: net.extract_embedding(x)[0], xs[0].cuda())
    assert_close_rel(exact.cpu().numpy(), _oracle_rows(sd, xs[0]), rtol=1e-4, what="the calibrating pass itself (exact f32)")
    assert packing.act_exponents(net).get("in", 0) > 14
    got = net.extract_embedding(xs[1].cuda())[0]
    _lib.check_range(sync=True)                                      # nothing reported
    assert_close_rel(got.cpu().numpy(), _oracle_rows(sd, xs[1]), rtol=1e-4, what="plain f16x3 after an explicit calibration")


def test_trainer_extraction_of_an_out_of_range_list_repairs_calibrates_and_reports(tmp_path, monkeypatch):
    """The entry point: train_audio.Trainer (arith auto, the shipped default) extracting a ragged test list whose features carry a
    gain of 2^-18.  The first batch is computed again in exact f32 and calibrates the encoder; the extractor's plans are re-recorded;
    the table is oracle-grade row by row; the trainer's extract_stats and arith.STATS say what happened."""
    import train_audio
    from deeplip_amd import arith, weightgen as wg
    from oracle import deeplip_oracle as O
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("DLIP_ARITH", "auto")
    tr = train_audio.Trainer(overrides={"data.test_speakers": 4, "data.test_utt_per_spk": 4, "data.trials": 100, "data.trial_targets": 20,
                                        "data.n_spk": 6, "data.utt_per_spk": 2, "data.test_audio_frames": [60, 140], "test.write_store": False})
    ds = tr.voxtestset
    item = ds.audio_item
    g = np.float32(2.0 ** -18)
    ds.audio_item = lambda i: item(i) * g
    n0, c0 = arith.STATS["f32_reruns"], arith.STATS["calibrations"]
    table = tr._xvectors(ds, batch=8, normalize=False)
    assert arith.STATS["f32_reruns"] > n0 and arith.STATS["calibrations"] > c0 and tr.extract_stats["f32_reruns"] >= 1
    sd = O.to_torch_sd(wg.fill_state_dict({k: tuple(v.shape) for k, v in tr.model.state_dict().items()}, prefix="audio."))
    rows = []
    with torch.no_grad():
        for i in range(len(ds)):
            rows.append(O.speaker_extract_embedding(sd, torch.from_numpy(ds.audio_item(i)[None]), O.ETDNN_CONTEXT)[0])
    assert_close_rel(table.emb.cpu().numpy(), torch.cat(rows).numpy(), rtol=1e-4, what="x-vectors of an out-of-range list")
    n1 = arith.STATS["f32_reruns"]
    tr._xvectors(ds, batch=8, normalize=False)                       # a second pass over the list: the calibrated fast path
    assert arith.STATS["f32_reruns"] == n1
    tr.close()


def test_pipeline_survives_a_list_whose_gains_jump_between_extremes():
    """Soak: 36 batches through ONE ExtractPipeline under auto with the calibration ON, their gains drawn from {1, 2^-18, 2^+15, 1e-4} in
    a seeded shuffle -- no single set of exponents suits the list.  Whatever the pipeline does about it (repair, calibrate, re-record;
    at most packing.MAX_CALIBRATIONS calibrations per model, then repairs only), EVERY row is at the 1e-4 bar against the engine's own
    exact mode on the same list, nothing is left in any status block, and the counters add up."""
    from deeplip_amd import _lib, arith, packing, weightgen as wg
    from deeplip_amd.pipeline import ExtractPipeline, pin
    net, sd = _tdnn()
    B, n = 4, 36
    r = np.random.Generator(np.random.PCG64(606))
    gains = r.choice(np.array([1.0, 2.0 ** -18, 2.0 ** 15, 1.0e-4], dtype=np.float32), size=n)
    gains[0] = 1.0                                                   # (the recording batch is an ordinary one)
    xs = [torch.from_numpy(wg.audio_input(B, 24, 160, key=f"arith.soak{i}")) * float(gains[i]) for i in range(n)]
    arith.configure("f32")
    with torch.no_grad():
        want = torch.cat([net.extract_embedding(x.cuda())[0] for x in xs]).cpu().numpy()
    # (two of the exact rows against the oracle: the reference of this test is itself pinned)
    for i in (1, n - 1):
        assert_close_rel(want[i * B:(i + 1) * B], _oracle_rows(sd, xs[i]), rtol=1e-4, what=f"exact mode, batch {i}")
    arith.configure("auto")
    c0, n0 = arith.STATS["calibrations"], arith.STATS["f32_reruns"]
    with torch.no_grad():
        pipe = ExtractPipeline(lambda a: net.extract_embedding(a)[0], xs[0].cuda())
    table = torch.zeros((n * B, 512), device="cuda")
    pipe.run([(pin(x),) for x in xs], table)
    pipe.finish()
    got = table.cpu().numpy()
    for i in range(n):                                               # per batch: a relative bar must not be hidden by the 2^15 batches' magnitudes
        assert_close_rel(got[i * B:(i + 1) * B], want[i * B:(i + 1) * B], rtol=1e-4, what=f"soak batch {i} (gain {gains[i]:g})")
    cal = arith.STATS["calibrations"] - c0
    assert cal <= packing.MAX_CALIBRATIONS and pipe.reruns >= 1 and arith.STATS["f32_reruns"] - n0 >= pipe.reruns
    print(f"\nsoak: {n} batches, {pipe.reruns} repaired in f32, {cal} calibration(s), {pipe.rerecorded} re-recording(s), exponents {packing.act_exponents(net)}")
    pipe.close()
    _lib.check_range(sync=True)


def test_lip_clip_pipeline_survives_a_list_whose_gains_jump_between_extremes():
    """The same soak for the lip-clip encoder (float clips; the residual groups of the trunk share exponents, the stem's input carries one,
    the pooled output is de-scaled): 18 batches of two 29-frame clips with gains from {1, 2^-14, 2^+10} through ONE ExtractPipeline
    under auto, against the engine's exact mode; two of the exact batches against the oracle."""
    from deeplip_amd import _lib, arith, packing, weightgen as wg
    from deeplip_amd.pipeline import ExtractPipeline, pin
    from models.video_models.model import Lipreading
    from oracle import deeplip_oracle as O
    tcn = {"num_layers": 4, "kernel_size": [3, 5, 7], "dropout": 0.2, "dwpw": False, "width_mult": 1}
    net = Lipreading(num_classes=54, relu_type="prelu", tcn_options=tcn, extract_feats=True)
    sd = wg.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, prefix="video.")
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.eval().cuda()
    B, n = 2, 18
    r = np.random.Generator(np.random.PCG64(707))
    gains = r.choice(np.array([1.0, 2.0 ** -14, 2.0 ** 10], dtype=np.float32), size=n)
    gains[0] = 1.0
    xs = [torch.from_numpy(wg.video_input(B, frames=29, key=f"arith.vsoak{i}")) * float(gains[i]) for i in range(n)]
    arith.configure("f32")
    want = torch.cat([net.embed(x.cuda()) for x in xs]).cpu().numpy()
    with torch.no_grad():
        for i in (1, n - 1):
            ref = O.video_time_mean(O.lipreading_features(O.to_torch_sd(sd), xs[i])).numpy()
            assert_close_rel(want[i * B:(i + 1) * B], ref, rtol=1e-4, what=f"exact mode, clip batch {i}")
    arith.configure("auto")
    c0 = arith.STATS["calibrations"]
    with torch.no_grad():
        pipe = ExtractPipeline(lambda a: net.embed(a), xs[0].cuda())
    table = torch.zeros((n * B, want.shape[1]), device="cuda")
    pipe.run([(pin(x),) for x in xs], table)
    pipe.finish()
    got = table.cpu().numpy()
    for i in range(n):
        assert_close_rel(got[i * B:(i + 1) * B], want[i * B:(i + 1) * B], rtol=1e-4, what=f"clip soak batch {i} (gain {gains[i]:g})")
    cal = arith.STATS["calibrations"] - c0
    assert cal <= packing.MAX_CALIBRATIONS and pipe.reruns >= 1
    print(f"\nclip soak: {n} batches, {pipe.reruns} repaired in f32, {cal} calibration(s), {pipe.rerecorded} re-recording(s), exponents {packing.act_exponents(net)}")
    pipe.close()
    _lib.check_range(sync=True)


def test_new_weights_void_a_calibration_and_in_place_updates_keep_it():
    """Activation exponents belong to the weights they were measured on: ``load_state_dict`` voids them (and the calibration budget starts
    again), an in-place update of the same parameters (an optimizer step, a fine-tune) re-packs WITH them.  Both ways every row is at
    the bar."""
    from deeplip_amd import arith, packing, weightgen as wg
    net, sd = _tdnn()
    arith.configure("auto")
    g = float(2.0 ** -18)
    xs = [torch.from_numpy(wg.audio_input(3, 24, 200, key=f"arith.void{i}")) for i in range(3)]
    net.extract_embedding((xs[0] * g).cuda())                        # repaired in f32, calibrates
    e1 = dict(packing.act_exponents(net))
    assert e1.get("in", 0) > 14
    # an in-place update of one layer's weights (what an optimizer step does): same exponents, new pack, right rows
    with torch.no_grad():
        net.tdnn[1].context_layer.weight.mul_(1.01)
    sd2 = {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}
    n0 = arith.STATS["f32_reruns"]
    got = net.extract_embedding((xs[1] * g).cuda())[0]
    assert packing.act_exponents(net) == e1 and arith.STATS["f32_reruns"] == n0      # the fast arithmetic on the calibrated, re-packed model
    assert_close_rel(got.cpu().numpy(), _oracle_rows(sd2, xs[1] * g), rtol=1e-4, what="after an in-place weight update")
    # new weights: the exponents are void, an ordinary input runs the ordinary pack ...
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    assert packing.act_exponents(net) == {}
    got = net.extract_embedding(xs[2].cuda())[0]
    assert arith.STATS["f32_reruns"] == n0
    assert_close_rel(got.cpu().numpy(), _oracle_rows(sd, xs[2]), rtol=1e-4, what="ordinary input on reloaded weights")
    # ... and the out-of-range one is repaired and calibrates again (a fresh budget)
    c0 = arith.STATS["calibrations"]
    got = net.extract_embedding((xs[2] * g).cuda())[0]
    assert arith.STATS["f32_reruns"] == n0 + 1 and arith.STATS["calibrations"] == c0 + 1
    assert_close_rel(got.cpu().numpy(), _oracle_rows(sd, xs[2] * g), rtol=1e-4, what="out-of-range input on reloaded weights")

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


def test_auto_eager_call_is_computed_again_in_f32():
    """An under-range input (1e-4 of what CMVN-normalised features look like) through the eager entry point: f16x3 raises,
    auto returns oracle-grade rows and counts the re-run; an in-range call afterwards runs f16x3 again and counts nothing."""
    from deeplip_amd import _lib, arith, weightgen as wg
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
    """The lip-clip encoder on a checkpoint whose trunk BatchNorms shrink the activations below 2^-6 (resnet.py:28-69)."""
    from deeplip_amd import arith, weightgen as wg
    from models.video_models.model import Lipreading
    from oracle import deeplip_oracle as O
    tcn = {"num_layers": 4, "kernel_size": [3, 5, 7], "dropout": 0.2, "dwpw": False, "width_mult": 1}
    net = Lipreading(num_classes=54, relu_type="prelu", tcn_options=tcn, extract_feats=True)
    sd = wg.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, prefix="video.")
    for k in sd:                # every BatchNorm gamma of the trunk at 0.05: the activations shrink layer after layer, far below 2^-6
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
def test_pipeline_repairs_exactly_the_offending_batch(mode):
    """Five batches through an ExtractPipeline, the third one under-range.  auto: every row oracle-grade, ONE re-run, the other
    batches untouched by it (same bits as a run without the bad batch); f16x3: DeepLipRangeError by finish()."""
    from deeplip_amd import _lib, arith, weightgen as wg
    from deeplip_amd.pipeline import ExtractPipeline, pin
    net, sd = _tdnn()
    arith.configure(mode)
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
    assert isinstance(err, _lib.DeepLipRangeError) and "below 2^-6" in str(err)
    assert pa.take_range_error() is None                             # taken = cleared
    pa(x * 1.0e-4)
    torch.cuda.synchronize()
    with pytest.raises(_lib.DeepLipRangeError):
        _lib.check_range()
    pa.close(); pb.close()

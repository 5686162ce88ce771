"""The model-level parity suite again, with the implicit-GEMM layers in split-fp16 mode
(packing.set_precision("f16x3"): 3 x f16 MFMA per product, fp32 accumulate).  Same golden vectors,
same oracle, same bars: 1e-4 relative for features / embeddings / scores, argmax bit-exact."""
import pytest

import test_models_gpu as T
from test_models_gpu import *  # noqa: F401,F403  (re-collect every test of the fp32 suite)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _f16x3_mode():
    from deeplip_amd import packing
    packing.set_precision("f16x3")
    yield
    packing.set_precision("f32")


def test_mode_is_active(video_net):
    """The packed trunk weights really are split fp16 pairs (a silent fp32 run would void this file)."""
    from deeplip_amd import packing
    from deeplip_amd.video import _cached_pack
    net, _ = video_net
    assert packing.PRECISION == "f16x3"
    p = _cached_pack(net, next(net.parameters()).device, net._pack)
    assert p["trunk"][0]["conv1"].wscale is not None and p["trunk"][0]["conv1"].w.shape[-1] == 64


def test_fused_paths_match_their_unfused_twins(video_net):
    """The round-2 fusions of the extraction path -- the shortcut convolution folded into conv2's reduction, the pooled
    epilogue of the trunk's last convolution -- against the launches they replace, on the golden clip shape, and
    against the golden time-mean captured from the reference class."""
    from deeplip_amd import video as V
    net, _ = video_net
    x = torch.from_numpy(wg.video_input(4)).to(DEV)
    assert net._can_pool(x)                       # T*3*3 = 261 >= the tile rows: the pooled path is the one embed() takes
    outs = {}
    try:
        for key, (fs, fp) in {"fused": (True, True), "no_shortcut": (False, True), "no_pool": (True, False),
                              "unfused": (False, False)}.items():
            V.FUSE_SHORTCUT, V.FUSE_POOL = fs, fp
            outs[key] = net.embed(x).clone()
    finally:
        V.FUSE_SHORTCUT, V.FUSE_POOL = True, True
    torch.cuda.synchronize()
    for key in ("no_shortcut", "no_pool", "fused"):
        assert rel_err(outs[key].cpu().numpy(), outs["unfused"].cpu().numpy()) < 1e-6, key


def test_fused_audio_pooling_matches_unfused():
    from deeplip_amd import audio as A
    from models.audio_models.tdnn import SpeakerEmbNet
    net, sd = load(SpeakerEmbNet(etdnn_opts(80)), "audio80.")
    x = torch.from_numpy(wg.audio_input(5, 80, 300)).to(DEV)
    try:
        A.FUSE_POOL = False
        xv0, xa0 = (t.clone() for t in net.extract_embedding(x))
        A.FUSE_POOL = True
        xv1, xa1 = net.extract_embedding(x)
    finally:
        A.FUSE_POOL = True
    torch.cuda.synchronize()
    assert rel_err(xa1.cpu().numpy(), xa0.cpu().numpy()) < 1e-6
    assert rel_err(xv1.cpu().numpy(), xv0.cpu().numpy()) < 1e-6
    with torch.no_grad():
        rxv, rxa = O.speaker_extract_embedding(sd, x.cpu(), O.ETDNN_CONTEXT)
    assert rel_err(xv1.cpu().numpy(), rxv.numpy()) < TOL


def test_fuse_av_finishes_pooled_means_bit_identically(video_net):
    """fusion.fuse_av on the still-pooled clip means (one launch) == pool_finish + znorm_cat (two launches), bit for bit."""
    from deeplip_amd import fusion, ops
    net, _ = video_net
    x = torch.from_numpy(wg.video_input(3)).to(DEV)
    a = torch.from_numpy(wg.audio_input(3, 24, 64)).to(DEV)[:, :, 0].contiguous().repeat(1, 22)[:, :512].contiguous()
    pooled = net.embed(x, finish=False)
    assert isinstance(pooled, ops.Pooled)
    one = fusion.fuse_av(a, pooled)
    two = fusion.fuse_av(a, ops.pool_finish(pooled, "mean"))
    torch.cuda.synchronize()
    assert one.shape == (3, 1024) and torch.equal(one, two)

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

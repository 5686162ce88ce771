"""SURVEY §8(f) rank 4 (-m gpu): the heads the north star names but the reference does not pin -- the ResNet speech
encoder of `arch: resnet` (config only upstream) and AAM-softmax (an empty stub upstream).  Parity is against this
repo's oracle restatement (parity unpinned upstream), 1e-4 relative, argmax bit-exact, both arithmetic modes;
training steps against torch autograd of the oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err
from deeplip_amd import packing, weightgen as wg
from oracle import deeplip_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
CFG = {"arch": "resnet", "resnet": {"input_dim": 1, "hidden_dim": [64, 128, 256], "residual_block_layers": [3, 3, 3],
                                    "fc_layers": 1, "embedding_dim": 256, "pooling": "average"}}


def _net():
    from models.resnet import SpeakerEmbNet
    net = SpeakerEmbNet(CFG)
    sd = wg.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, prefix="aresnet.")
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return net.to(DEV), O.to_torch_sd(sd)


@pytest.mark.parametrize("mode", ["f32", "f16x3"])
def test_audio_resnet_embedding_vs_oracle(mode):
    packing.set_precision(mode)
    try:
        net, sd = _net()
        net.eval()
        x = torch.from_numpy(wg.audio_input(6, 40, 100, key="aresnet.x")).unsqueeze(1)          # [B,1,F,T]
        e, e2 = net.extract_embedding(x.to(DEV))
        e3 = net.extract_embedding(x[2:4].to(DEV))[0]
        torch.cuda.synchronize()
        with torch.no_grad():
            ref = O.audio_resnet_embedding(sd, x)
        assert e.shape == (6, 256) and e is e2
        assert rel_err(e.cpu().numpy(), ref.numpy()) < 1e-4
        assert rel_err(e[2:4].cpu().numpy(), e3.cpu().numpy()) < 1e-6                            # batch invariance
    finally:
        packing.set_precision("f32")


def test_audio_resnet_train_step_vs_oracle_autograd():
    """model.train(): batch-statistics BN, backward through every layer on the engine, vs torch autograd of the
    oracle restatement (same ill-conditioning caveat as the lip-clip trunk: compare against fp64)."""
    net, sd = _net()
    net.train()
    x = torch.from_numpy(wg.audio_input(4, 24, 60, key="aresnet.xt")).unsqueeze(1)
    lab = torch.from_numpy(wg.labels(4, 10))
    from models.audio_models.loss import AAMSoftmax
    crit = AAMSoftmax(256, 10, 30.0, 0.2).to(DEV)
    cw = wg.fill_state_dict({"weights": (10, 256)}, prefix="aresnet.aam.")["weights"]
    crit.load_state_dict({"weights": torch.from_numpy(cw)})
    loss, logits = crit(net(x.to(DEV)), lab.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    p64 = {k: v.double().requires_grad_(v.dtype.is_floating_point and "running" not in k) if v.dtype.is_floating_point else v
           for k, v in sd.items()}
    w64 = torch.from_numpy(cw).double().requires_grad_()
    with O.bn_training():
        ref_loss, ref_logits = O.aam_softmax(O.audio_resnet_embedding(p64, x.double()), lab, w64, 30.0, 0.2)
    ref_loss.backward()
    assert abs(float(loss.detach()) - float(ref_loss)) < 1e-4 * abs(float(ref_loss))
    assert rel_err(logits.detach().cpu().numpy(), ref_logits.detach().numpy()) < 1e-4
    assert np.array_equal(torch.max(logits, 1)[1].cpu().numpy(), O.argmax_first(ref_logits.detach().float()).numpy())
    grads = dict(net.named_parameters())
    for k in ("conv1.weight", "layers.0.1.conv2.weight", "layers.1.0.downsample.0.weight", "layers.2.2.bn2.weight", "fc.weight"):
        assert rel_err(grads[k].grad.cpu().numpy(), p64[k].grad.numpy()) < 2e-3, k            # fp32 engine vs fp64 autograd
    assert rel_err(crit.weights.grad.cpu().numpy(), w64.grad.numpy()) < 1e-4
    assert int(net.bn1.num_batches_tracked) == 1


def test_aam_softmax_forward_backward_vs_oracle():
    from deeplip_amd.loss import AAMSoftmax
    B, D, K = 16, 64, 12
    g = torch.Generator().manual_seed(3)
    emb = torch.randn(B, D, generator=g)
    w = torch.randn(K, D, generator=g)
    emb[0] = -w[3] * 2.0                       # a target near theta = pi: exercises the fallback branch
    lab = torch.arange(B) % K
    lab[0] = 3
    for easy in (False, True):
        crit = AAMSoftmax(D, K, 30.0, 0.3, easy_margin=easy).to(DEV)
        crit.load_state_dict({"weights": w})
        e = emb.clone().to(DEV).requires_grad_()
        loss, logits = crit(e, lab.to(DEV))
        loss.backward()
        er = emb.clone().double().requires_grad_(); wr = w.clone().double().requires_grad_()
        rl, rlog = O.aam_softmax(er, lab, wr, 30.0, 0.3, easy)
        rl.backward()
        torch.cuda.synchronize()
        assert abs(float(loss.detach()) - float(rl)) < 1e-5 * abs(float(rl))
        assert rel_err(logits.detach().cpu().numpy(), rlog.detach().numpy()) < 1e-5
        assert rel_err(e.grad.cpu().numpy(), er.grad.numpy()) < 1e-4
        assert rel_err(crit.weights.grad.cpu().numpy(), wr.grad.numpy()) < 1e-4
        with torch.no_grad():
            l2, lg2, amax = crit.predict(emb.to(DEV), lab.to(DEV))                              # inference path, same numbers
        assert abs(float(l2) - float(rl)) < 1e-5 * abs(float(rl))
        assert np.array_equal(amax.cpu().numpy(), O.argmax_first(rlog.detach().float()).numpy())

"""Config C5 in miniature (-m gpu): two SGD steps of the trainable tail (Linearfusion in train mode
+ CrossEntropy / LMCL) through the HIP forward AND backward kernels, against the golden values
captured from the reference classes; plus gradient checks of each autograd Function against torch.
Tolerance 1e-4 relative (north star); argmax bit-exact."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from deeplip_amd import weightgen as wg

pytestmark = pytest.mark.gpu
DEV = "cuda"


def load(module, prefix):
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    sd = wg.fill_state_dict(shapes, prefix=prefix)
    module.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return module.to(DEV)


@pytest.mark.parametrize("tag", ["ce", "lmcl"])
def test_two_sgd_steps_vs_reference_golden(golden, tag):
    from models.audio_models.loss import LMCL, CrossEntropy
    from models.fusion_models.model_fusion import model_fusion
    g = golden["train"]
    B = 60
    xa = torch.from_numpy(wg.gen("train.xv_audio", (B, 512))).to(DEV)
    ev = torch.from_numpy(wg.gen("train.em_video", (B, 512))).to(DEV)
    lab = torch.from_numpy(wg.labels(B, 57)).to(DEV)
    fus = load(model_fusion(1024, 512, 57, False), "train.lf.").train()
    crit = load(CrossEntropy(512, 57) if tag == "ce" else LMCL(512, 57, 30, 0.2), f"train.{tag}.").train()
    opt = torch.optim.SGD([{"params": fus.parameters()}, {"params": crit.parameters()}], lr=0.5, weight_decay=1e-5,
                          momentum=0.9)
    for step in range(2):
        opt.zero_grad()
        out = fus(torch.cat([xa, ev], dim=1))
        loss, logits = crit(out, lab)
        loss.backward()
        if step == 0:
            assert abs(float(loss) - float(g[f"{tag}_loss0"])) < 1e-4 * float(g[f"{tag}_loss0"])
            assert rel_err(logits.detach().cpu().numpy(), g[f"{tag}_logits0"]) < 1e-4
            assert np.array_equal(torch.max(logits, 1)[1].cpu().numpy(), g[f"{tag}_argmax0"])
            assert rel_err(fus.fc2.weight.grad[:8].cpu().numpy(), g[f"{tag}_grad_fc2_w_rows8"]) < 1e-4
            # a bias in front of train-mode BN has zero gradient analytically: both sides are rounding noise
            assert float(fus.fc1.bias.grad.abs().max()) < 1e-5 and float(np.abs(g[f"{tag}_grad_fc1_b"]).max()) < 1e-5
            assert rel_err(fus.bn1.weight.grad.cpu().numpy(), g[f"{tag}_grad_bn1_w"]) < 1e-4
            cw = crit.fc.weight if tag == "ce" else crit.weights
            assert rel_err(cw.grad.cpu().numpy(), g[f"{tag}_grad_crit_w"]) < 1e-4
        opt.step()
    assert abs(float(loss) - float(g[f"{tag}_loss1"])) < 1e-3 * max(1.0, float(g[f"{tag}_loss1"]))
    assert rel_err(fus.fc2.weight.detach()[:8].cpu().numpy(), g[f"{tag}_after2_fc2_w_rows8"]) < 1e-4
    assert rel_err(fus.bn1.running_var.cpu().numpy(), g[f"{tag}_after2_bn1_running_var"]) < 1e-5
    assert int(fus.bn1.num_batches_tracked) == 2
    for k, v in {**{"fus." + k: v for k, v in fus.state_dict().items()},
                 **{"crit." + k: v for k, v in crit.state_dict().items()}}.items():
        ref = g[f"{tag}_after2_{k}_sum"]
        v = v.detach().double().cpu()
        assert abs(float(v.abs().sum()) - ref[1]) < 1e-4 * max(ref[1], 1e-6), k


def test_autograd_functions_vs_torch():
    from deeplip_amd import autograd as ag
    torch.manual_seed(0)
    x = torch.randn(37, 60, device=DEV, requires_grad=True)       # odd sizes: small-GEMM path
    w = torch.randn(57, 60, device=DEV, requires_grad=True)
    b = torch.randn(57, device=DEV, requires_grad=True)
    lab = torch.randint(0, 57, (37,), device=DEV)
    y = ag.linear(x, w, b)
    loss = ag.margin_ce_loss(ag.l2_normalize(y), lab, 30.0, 0.2)
    loss.backward()
    xc, wc, bc = (t.detach().cpu().double().requires_grad_() for t in (x, w, b))
    yc = torch.nn.functional.linear(xc, wc, bc)
    lg = torch.nn.functional.normalize(yc)
    m = torch.zeros_like(lg); m.scatter_(1, lab.cpu().view(-1, 1), 0.2)
    lc = torch.nn.functional.cross_entropy(30.0 * (lg - m) + 1e-8, lab.cpu())
    lc.backward()
    assert abs(float(loss) - float(lc)) < 1e-5 * abs(float(lc))
    for a, c in ((x, xc), (w, wc), (b, bc)):
        assert rel_err(a.grad.cpu().numpy(), c.grad.numpy()) < 2e-5


@pytest.mark.gpu
def test_large_linear_products_run_on_the_matrix_kernels(monkeypatch):
    """nn.Linear of 2^27 multiply-adds or more under autograd (the speech encoder's fc1: 256 x 3000 -> 512): forward, dx, dW through the
    split-fp16 convolution kernels (the input's 3 000 columns zero-padded to 3 008 by the pass that splits it) against fp64, and against the
    exact-fp32 route it replaces."""
    from deeplip_amd import autograd as ag
    g = torch.Generator().manual_seed(5)
    x = torch.randn(256, 3000, generator=g)
    w = torch.randn(512, 3000, generator=g) * 0.02
    b = torch.randn(512, generator=g) * 0.1
    dy = torch.randn(256, 512, generator=g) * 1e-3
    xr, wr, br = (t.double().requires_grad_() for t in (x, w, b))
    (xr @ wr.t() + br).backward(dy.double())
    out = {}
    for on in (True, False):
        monkeypatch.setattr(ag, "LINEAR_ON_MATRIX_KERNELS", on)
        xg, wg_, bg = (t.clone().cuda().requires_grad_() for t in (x, w, b))
        y = ag.linear(xg, wg_, bg)
        y.backward(dy.cuda())
        torch.cuda.synchronize()
        out[on] = (y.detach().cpu(), xg.grad.cpu(), wg_.grad.cpu(), bg.grad.cpu())
        ref = (xr.detach() @ wr.detach().t() + br.detach())
        for got, want, tol in ((out[on][0], ref, 2e-5), (out[on][1], xr.grad, 1e-4), (out[on][2], wr.grad, 1e-4), (out[on][3], br.grad, 1e-4)):
            assert float((got.double() - want).abs().max() / want.abs().max()) < tol, on
    for a, c in zip(out[True], out[False]):
        assert float((a.double() - c.double()).abs().max() / c.double().abs().max()) < 5e-6

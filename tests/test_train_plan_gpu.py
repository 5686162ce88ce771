"""Recorded training steps (deeplip_amd/train_plan.py) -- first contact must not depend on luck: a recorded step is checked, on its
first replay, against an eager step from the same state and dropped (with the state restored) on any doubt.  Exercised here at one
rank with verify=True forced; at world > 1 it is on by default (train_fusion.py:241-315's step, trained through a captured RCCL
all-reduce, has never met a second rank on this pool)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _head(seed=0):
    from models.audio_models.loss import CrossEntropy
    from models.fusion_models import model_fusion
    torch.manual_seed(seed)
    head = model_fusion.model_fusion(64, 32, 7, extract_feats=False).to(DEV).train()
    crit = CrossEntropy(32, 7).to(DEV)
    opt = torch.optim.SGD([{"params": head.parameters()}, {"params": crit.parameters()}], lr=torch.tensor(0.05, device=DEV), momentum=0.9,
                          weight_decay=1e-5, fused=True)
    return head, crit, opt


def _batches(n, B=16):
    g = torch.Generator().manual_seed(3)
    return [(torch.randn(B, 64, generator=g).to(DEV), torch.randint(0, 7, (B,), generator=g).to(DEV)) for _ in range(n)]


def _make_step(head, crit, opt, poison=False):
    def one(x, lab):
        opt.zero_grad(set_to_none=True)
        loss, logits = crit(head(x), lab)
        if poison and torch.cuda.is_current_stream_capturing():
            loss = loss * 1.001                      # a recorded step that does NOT compute what the eager step computes
        loss.backward()
        opt.step()
        return loss, logits
    return one


def _run(verify, poison=False, steps=5):
    from deeplip_amd.train_plan import TrainStepGraph, grad_witness, step_state
    head, crit, opt = _head()
    plan = TrainStepGraph(_make_step(head, crit, opt, poison), eager_steps=1, device=torch.device(DEV), branch_streams=False,
                          verify=verify, state=step_state([head, crit], [opt]), witness=grad_witness([head, crit]))
    losses = []
    for x, lab in _batches(steps):
        loss, _ = plan.step(x, lab)
        plan.finish()
        losses.append(float(loss))
    params = torch.cat([p.detach().reshape(-1) for m in (head, crit) for p in m.parameters()]).cpu()
    bufs = torch.cat([b.detach().double().reshape(-1) for b in head.buffers()]).cpu()
    return plan, losses, params, bufs


def test_verified_recorded_step_matches_the_unverified_and_the_eager_run():
    plan_v, l_v, p_v, b_v = _run(True)
    assert plan_v.mode == "graph" and plan_v.verified is not None
    assert plan_v.verified["outputs_rel_err"] <= 1e-6 and plan_v.verified["witness_rel_err"] <= 1e-6 and plan_v.verified["tensors"] >= 4
    plan_n, l_n, p_n, b_n = _run(False)
    assert plan_n.mode == "graph" and plan_n.verified is None
    # the verification step ran the recording call's batch twice (eager, then replay) from a restored snapshot: same trajectory
    assert l_v == l_n and torch.equal(p_v, p_n) and torch.equal(b_v, b_n)


def test_a_recorded_step_that_differs_is_dropped_and_the_state_restored():
    plan_p, l_p, p_p, b_p = _run(True, poison=True)
    assert plan_p.mode.startswith("eager: first replay differs") and not plan_p.recorded
    # every step, the recording call's included, was an eager step from the right state: the trajectory of a run that never recorded
    from deeplip_amd.train_plan import TrainStepGraph
    head, crit, opt = _head()
    one = _make_step(head, crit, opt)
    ref = []
    for x, lab in _batches(5):
        loss, _ = one(x, lab)
        ref.append(float(loss))
    torch.cuda.synchronize()
    params = torch.cat([p.detach().reshape(-1) for m in (head, crit) for p in m.parameters()]).cpu()
    bufs = torch.cat([b.detach().double().reshape(-1) for b in head.buffers()]).cpu()
    assert l_p == ref and torch.equal(p_p, params) and torch.equal(b_p, bufs)       # (num_batches_tracked among the buffers: 5, not 6)


def test_a_capture_that_fails_falls_back_to_eager():
    from deeplip_amd.train_plan import TrainStepGraph, grad_witness, step_state
    head, crit, opt = _head()
    inner = _make_step(head, crit, opt)

    def one(x, lab):
        if torch.cuda.is_current_stream_capturing():
            float(x.sum())                           # a host read inside a capture: the capture dies
        return inner(x, lab)

    plan = TrainStepGraph(one, eager_steps=1, device=torch.device(DEV), branch_streams=False, verify=True,
                          state=step_state([head, crit], [opt]), witness=grad_witness([head, crit]))
    for x, lab in _batches(4):
        loss, _ = plan.step(x, lab)
        plan.finish()
        assert np.isfinite(float(loss))
    assert plan.mode.startswith("eager: capture failed")


def test_shape_keyed_steps_records_one_graph_per_shape():
    from deeplip_amd.train_plan import ShapeKeyedSteps
    head, crit, opt = _head()
    steps = ShapeKeyedSteps(_make_step(head, crit, opt), eager_steps=1, device=torch.device(DEV), branch_streams=False)
    g = torch.Generator().manual_seed(9)
    for B in (8, 12, 8, 12, 8, 12):
        x, lab = torch.randn(B, 64, generator=g).to(DEV), torch.randint(0, 7, (B,), generator=g).to(DEV)
        loss, logits = steps.step(x, lab)
        steps.finish()
        assert logits.shape == (B, 7) and np.isfinite(float(loss))
    assert steps.summary() == {"shapes": 2, "recorded": 2, "eager_only": []}

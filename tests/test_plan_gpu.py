"""Step plans (deeplip_amd/plan.py over dlip_plan_*): a recorded step replays bit-identically to the eager
launches it was recorded from, accepts new inputs, and refuses to run on stale weights."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TCN = {"num_layers": 4, "kernel_size": [3, 5, 7], "dropout": 0.2, "dwpw": False, "width_mult": 1}
CTX = [[-2, -1, 0, 1, 2], [0], [-2, 0, 2], [0], [-3, 0, 3], [0], [-4, 0, 4], [0], [0], [0]]


def _models(precision):
    from deeplip_amd import packing, weightgen as wg
    from models.audio_models.tdnn import SpeakerEmbNet
    from models.video_models.model import Lipreading
    packing.set_precision(precision)
    video = Lipreading(num_classes=54, relu_type="prelu", tcn_options=TCN, extract_feats=True)
    et = {"input_dim": 24, "hidden_dim": [512] * 9 + [1500], "context": CTX, "tdnn_layers": 10, "embedding_dim": 512,
          "pooling": "statistic", "attention_hidden_size": 64, "bn_first": True}
    audio = SpeakerEmbNet({"arch": "etdnn", "etdnn": et})
    for name, m in (("video.", video), ("audio.", audio)):
        sd = wg.fill_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, prefix=name)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        m.eval().cuda()
    return video, audio


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_plan_replays_the_eager_step_bit_for_bit(precision):
    from deeplip_amd import fusion, packing, weightgen as wg
    from deeplip_amd.plan import StepPlan
    try:
        video, audio = _models(precision)

        def step(xv, xa):
            return fusion.fuse_av(audio.extract_embedding(xa)[0], video.embed(xv))

        xv = torch.from_numpy(wg.video_input(3, frames=11, key="plan.video")).cuda()
        xa = torch.from_numpy(wg.audio_input(3, 24, 150, key="plan.audio")).cuda()
        ref = step(xv, xa).clone()
        plan = StepPlan(step, xv.clone(), xa.clone())
        assert plan.launches >= 30                       # the whole step is in the plan
        for _ in range(3):
            out = plan.run()
        torch.cuda.synchronize()
        assert torch.equal(out, ref)
        # new inputs go through the recorded input buffers
        xv2 = torch.from_numpy(wg.video_input(3, frames=11, key="plan.video2")).cuda()
        xa2 = torch.from_numpy(wg.audio_input(3, 24, 150, key="plan.audio2")).cuda()
        ref2 = step(xv2, xa2).clone()
        out2 = plan(xv2, xa2)
        torch.cuda.synchronize()
        assert torch.equal(out2, ref2)
        assert not torch.equal(ref2, ref)
        with pytest.raises(ValueError):
            plan(xv2[:2], xa2)
        plan.close()
    finally:
        packing.set_precision("f32")


def test_plan_is_stale_after_a_weight_change():
    from deeplip_amd import fusion, packing, weightgen as wg
    from deeplip_amd.plan import StalePlanError, StepPlan
    video, audio = _models("f32")
    xa = torch.from_numpy(wg.audio_input(2, 24, 120, key="plan.stale")).cuda()
    plan = StepPlan(lambda a: audio.extract_embedding(a)[0], xa)
    plan.run()
    sd = audio.state_dict()
    sd["fc2.bias"] = sd["fc2.bias"] + 1.0
    audio.load_state_dict(sd)
    with pytest.raises(StalePlanError):
        plan.run()
    # eager forwards pick the new weights up (pack cache invalidated by the same event)
    a = audio.extract_embedding(xa)[0]
    plan2 = StepPlan(lambda a: audio.extract_embedding(a)[0], xa)
    torch.cuda.synchronize()
    assert torch.equal(plan2.run(), a)


def test_plan_is_stale_after_in_place_edits_too():
    """What the generation counter does not see -- an optimizer step on an eval-mode model, `p.data = other`, a hand edit --
    changes the fingerprint the eager path compares (version counters + data pointers): a replay refuses the stale packs."""
    from deeplip_amd import weightgen as wg
    from deeplip_amd.plan import StalePlanError, StepPlan
    video, audio = _models("f32")
    xa = torch.from_numpy(wg.audio_input(2, 24, 120, key="plan.stale2")).cuda()
    plan = StepPlan(lambda a: audio.extract_embedding(a)[0], xa)
    plan.run()
    with torch.no_grad():
        audio.fc2.bias.add_(0.5)                                   # in place: _version moves
    with pytest.raises(StalePlanError, match="in place"):
        plan.run()
    plan = StepPlan(lambda a: audio.extract_embedding(a)[0], xa)
    plan.run()
    audio.fc2.bias.data = audio.fc2.bias.data.clone() + 1.0         # through .data: only the pointer moves
    with pytest.raises(StalePlanError, match="in place"):
        plan.run()


def test_pack_cache_sees_in_place_edits():
    """Eager forwards fingerprint the parameters' version counters: an in-place edit without load_state_dict
    (an optimizer step, a broadcast) repacks."""
    from deeplip_amd import weightgen as wg
    video, audio = _models("f32")
    xa = torch.from_numpy(wg.audio_input(2, 24, 120, key="plan.inplace")).cuda()
    a0 = audio.extract_embedding(xa)[0].clone()
    with torch.no_grad():
        audio.fc2.bias.add_(0.5)
    a1 = audio.extract_embedding(xa)[0]
    torch.cuda.synchronize()
    np.testing.assert_allclose((a1 - a0).cpu().numpy(), 0.5, atol=1e-5)


def test_two_stream_step_is_bit_identical_eager_and_recorded():
    """fusion.embed_av issues the speech encoder on a second stream (fork / join); recorded into a plan the two encoders
    are parallel branches of the graph.  Same bits as the sequential step, eagerly and replayed."""
    from deeplip_amd import fusion, packing, weightgen as wg
    from deeplip_amd.plan import StepPlan
    try:
        video, audio = _models("f16x3")
        xv = torch.from_numpy(wg.video_input(3, frames=29, key="plan2s.video")).cuda()
        xa = torch.from_numpy(wg.audio_input(3, 24, 300, key="plan2s.audio")).cuda()
        seq = fusion.embed_av(audio, video, xa, xv, two_streams=False).clone()
        par = fusion.embed_av(audio, video, xa, xv, two_streams=True).clone()
        torch.cuda.synchronize()
        assert torch.equal(seq, par)
        plan = StepPlan(lambda v, a: fusion.embed_av(audio, video, a, v), xv.clone(), xa.clone())
        for _ in range(4):
            out = plan.run()
        torch.cuda.synchronize()
        assert torch.equal(out, seq)
        plan.close()
    finally:
        packing.set_precision("f32")


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
def test_extract_pipeline_rows_equal_the_direct_step(precision):
    """deeplip_amd.pipeline.ExtractPipeline: uint8 RGB frames + mel batches from pinned host memory, H2D on a copy stream into
    one of two input sets while the other set's plan replays.  Seven batches (three distinct, a short last one) -> every
    table row equals the row the direct, un-pipelined step gives for that batch -- bit for bit, in list order."""
    from deeplip_amd import fusion, packing, weightgen as wg
    from deeplip_amd.pipeline import ExtractPipeline
    from deeplip_amd.synthetic import frames_u8_from_clips
    try:
        video, audio = _models(precision)

        def step(frames, xa):
            return fusion.embed_av(audio, video, xa, frames)

        B = 3
        host = []
        for r in range(3):
            fr = torch.from_numpy(frames_u8_from_clips(wg.video_input(B, frames=11, key=f"pipe.v{r}"), rgb=True)).pin_memory()
            xa = torch.from_numpy(wg.audio_input(B, 24, 150, key=f"pipe.a{r}")).pin_memory()
            host.append((fr, xa))
        assert host[0][0].dtype == torch.uint8 and tuple(host[0][0].shape) == (B, 11, 3, 88, 88)
        want = [step(fr.cuda(), xa.cuda()).clone() for fr, xa in host]
        torch.cuda.synchronize()
        order = [0, 1, 2, 1, 0, 2]
        batches = [host[i] for i in order] + [(host[1][0][:2], host[1][1][:2])]      # a short last batch
        pipe = ExtractPipeline(step, host[0][0].cuda(), host[0][1].cuda())
        table = torch.full((len(order) * B + 2, 1024), float("nan"), device="cuda")
        n = pipe.run(batches, table)
        pipe.finish()
        assert n == table.shape[0]
        for j, i in enumerate(order):
            assert torch.equal(table[j * B:(j + 1) * B], want[i]), (j, i)
        assert torch.equal(table[-2:], want[1][:2])
        # a second run reuses the sets (FREE events of the previous run are behind us)
        table2 = torch.empty_like(table)
        pipe.run(batches, table2); pipe.finish()
        assert torch.equal(table2, table)
        pipe.close()
    finally:
        packing.set_precision("f32")


def test_plan_spans_time_the_replayed_launches():
    """spans=True: every MFMA launch of the recorded step (stem + pool with its pre-pass, window, ring and rows kernels) times itself in-kernel on every replay
    (dlip_span_scope_*: first workgroup in -> last workgroup out on the constant 100 MHz clock) -- the per-kernel timing a replayed
    hipGraph cannot give the host.  Results are bit-identical with and without; the spans add up to less than the replay's wall
    time on one stream and each is a plausible duration."""
    from deeplip_amd import fusion, packing, weightgen as wg
    from deeplip_amd.plan import StepPlan
    try:
        video, audio = _models("f16x3")
        xv = torch.from_numpy(wg.video_input(4, frames=29, key="span.video")).cuda()
        xa = torch.from_numpy(wg.audio_input(4, 24, 300, key="span.audio")).cuda()
        step = lambda v, a: fusion.embed_av(audio, video, a, v, two_streams=False)
        ref = StepPlan(step, xv.clone(), xa.clone())
        want = ref.run().clone()
        plan = StepPlan(step, xv.clone(), xa.clone(), spans=True)
        assert plan.launches == ref.launches + 1                   # + the collect kernel
        from deeplip_amd.plan import SPAN_KERNELS
        kinds = {k for n, _, _ in plan.span_names for k in SPAN_KERNELS if k in n}
        assert len(plan.span_names) >= 20 and {"conv_igemm_f16x3_dma_kernel", "conv_win_f16x3_kernel", "stem3d_pool_f16x3_kernel"} <= kinds
        assert len({st for _, _, st in plan.span_names}) == 1      # two_streams=False: one stream
        plan.span_summary()                                        # drop the recording passes
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            out = plan.run()
        e1.record()
        torch.cuda.synchronize()
        assert torch.equal(out, want)
        s = plan.span_summary()
        assert s and all(v["replays"] == 5 for v in s.values())
        assert sum(v["launches_per_step"] for v in s.values()) == len(plan.span_names)
        total_us = sum(v["us_sum"] for v in s.values())
        wall_us = 1e3 * e0.elapsed_time(e1) / 5
        assert 0 < total_us < wall_us, (total_us, wall_us)         # one stream: the MFMA launches are a part of the replay, never more
        # (the summary rounds every kernel's sum to 0.01 us and the per-stream total to 0.001 us)
        assert abs(sum(plan.last_stream_us.values()) - total_us) < 0.01 * len(s) + 1e-6 * total_us and list(plan.last_stream_us) == ["stream0"]
        assert all(1.0 < v["avg_launch_us"] < 5e3 for v in s.values())
        assert plan.span_summary() == {}                           # reset by the previous call
        plan.close(); ref.close()
    finally:
        packing.set_precision("f32")

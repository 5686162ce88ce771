"""Host logic of the arithmetic switch and of the ragged extractor's length validation (no GPU)."""
import numpy as np
import pytest


def test_resolve_precedence(monkeypatch):
    from deeplip_amd import arith
    monkeypatch.delenv("DLIP_ARITH", raising=False)
    assert arith.resolve() == "auto"                                  # the library's default: the measured configuration
    assert arith.resolve(None, "f32") == "f32"                        # the config file's key
    monkeypatch.setenv("DLIP_ARITH", "f16x3")
    assert arith.resolve(None, "f32") == "f16x3"                      # the environment overrides the file
    assert arith.resolve("auto", "f32") == "auto"                     # an explicit flag overrides both
    monkeypatch.setenv("DLIP_ARITH", "fp8")
    with pytest.raises(ValueError, match="DLIP_ARITH"):
        arith.resolve()


def test_configure_sets_both_paths(monkeypatch):
    from deeplip_amd import arith, autograd_video as av, packing
    monkeypatch.delenv("DLIP_ARITH", raising=False)
    assert arith.configure("f32") == "f32" and packing.PRECISION == "f32" and av.TRAIN_CONV == "f32" and not arith.fallback_enabled()
    assert arith.configure(None, "auto") == "auto" and packing.PRECISION == "f16x3" and av.TRAIN_CONV == "f16x3" and arith.fallback_enabled()
    assert arith.configure("f16x3") == "f16x3" and packing.PRECISION == "f16x3" and not arith.fallback_enabled()
    with arith.exact():
        assert packing.PRECISION == "f32"
    assert packing.PRECISION == "f16x3"


def test_entry_points_carry_the_arith_key():
    """All three configs name the arithmetic, and all three command lines take --arith."""
    import json
    import os
    import yaml
    from conftest import ROOT
    assert yaml.safe_load(open(os.path.join(ROOT, "conf/fusion_config.yaml")))["model"]["arith"] == "auto"
    assert yaml.safe_load(open(os.path.join(ROOT, "conf/audio_config.yaml")))["model"]["arith"] == "auto"
    assert json.load(open(os.path.join(ROOT, "conf/video_config.json")))["arith"] == "auto"
    import train_video
    assert train_video.load_args(["--arith", "f32"]).arith == "f32" and train_video.load_args([]).arith is None
    for script in ("train_fusion.py", "train_audio.py"):
        assert "arith.add_argument(ap)" in open(os.path.join(ROOT, script)).read()


class _FakeSet:
    def __init__(self, audio_len, clip_len, clip_ptr):
        self.audio_len, self.clip_len, self.clip_ptr = np.asarray(audio_len), np.asarray(clip_len), np.asarray(clip_ptr, dtype=np.int32)


def test_ragged_extractor_validates_lengths_on_the_host():
    """An utterance shorter than the encoder can pool (frames_consumed() + 2 = 24 for the E-TDNN), a zero-frame clip or an utterance
    without clips is refused BEFORE anything is batched: on the device a pooled count of 0 would come back as NaN rows."""
    from deeplip_amd.extract import RaggedExtractor
    ex = RaggedExtractor(lambda a, l: a, lambda v, l: v, "cpu", batch=4, audio_min_frames=24)
    with pytest.raises(ValueError, match="utterance 1 has 23 frames"):
        ex.run(_FakeSet([100, 23, 50], [5, 5, 5], [0, 1, 2, 3]), 0, 3, 8)
    with pytest.raises(ValueError, match="lip clip 2 has 0 frames"):
        ex.run(_FakeSet([100, 60, 50], [5, 5, 0], [0, 1, 2, 3]), 0, 3, 8)
    with pytest.raises(ValueError, match="utterance 1 has no lip clip"):
        ex.run(_FakeSet([100, 60, 50], [5, 5], [0, 1, 1, 2]), 0, 3, 8)


def test_shape_keyed_steps_bookkeeping():
    """ShapeKeyedSteps keeps one plan per (key, shapes), least recently used out -- host logic, checked with stand-in plans."""
    import torch
    from deeplip_amd import train_plan

    class Fake:
        made = 0

        def __init__(self, fn, **kw):
            Fake.made += 1
            self.recorded, self.eager_only, self.mode = False, None, "eager (warm-up)"

        def step(self, *a):
            return a

        def finish(self):
            pass

    orig, train_plan.TrainStepGraph = train_plan.TrainStepGraph, Fake
    try:
        s = train_plan.ShapeKeyedSteps(lambda *a: a, max_plans=2)
        a, b, c = torch.zeros(2, 3), torch.zeros(2, 4), torch.zeros(2, 5)
        s.step(a); s.step(b); s.step(a)
        assert Fake.made == 2 and len(s.plans) == 2
        s.step(a, key=0.35)                       # same shape, another baked-in constant: its own plan; `b` (least recently used) goes
        assert Fake.made == 3 and len(s.plans) == 2
        s.step(b)
        assert Fake.made == 4
        assert s.summary()["shapes"] == 2
    finally:
        train_plan.TrainStepGraph = orig

"""Entry points: train_video.py plumbing on CPU (BASELINE config C1: no GPU), and its GPU run."""
import os

import numpy as np
import pytest
import torch


def test_train_video_cpu_plumbing(tmp_path):
    """C1: CPU-only plumbing -- config, model with 54 classes, pad_packed_collate batches
    [B=4, T<=29, 88, 88] + lengths, Adam + per-iteration cosine LR, checkpoint round trip."""
    import train_video
    args = train_video.load_args(["--device", "cpu", "--save-path", str(tmp_path / "ck"), "--steps", "2"])
    model = train_video.get_model(args)
    assert len(model.state_dict()) == 343 and model.tcn.tcn_output.weight.shape == (54, 768)
    data, lengths, labels = train_video.synthetic_batch(args, 0)
    assert data.shape == (4, 29, 88, 88) and lengths == sorted(lengths, reverse=True) and lengths[0] == 29
    assert float(data[-1, lengths[-1]:].abs().max()) == 0.0 if lengths[-1] < 29 else True
    rgb, _, _ = train_video.synthetic_batch(args, 0, rgb=True)
    assert rgb.shape == (4, 29, 3, 96, 96) and rgb.dtype == torch.uint8      # --frame-size mouth crops: cropped to 88 on the GPU
    res = train_video.main(["--device", "cpu", "--save-path", str(tmp_path / "ck"), "--steps", "2"])
    assert res is None and os.path.exists(tmp_path / "ck" / "1.pt")


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["full", "full-rgb", "head-only", "c1-size"])
def test_train_video_gpu_two_steps(tmp_path, mode, arith_mode):
    """Default = full-model training as the reference does (model.train(), backward through stem / trunk / TCN on
    the engine); --head-only = classifier layer on frozen eval-mode features.  ``c1-size`` is BASELINE config C1's shape --
    conf/video_config.json, [4, (3 ->) 1, 29, 88, 88] from uint8 RGB, 54 classes, 2 optimisation steps -- whose "finite loss,
    logits [4,54]" half can only be computed on the GPU (the CPU run is plumbing-only by design)."""
    import train_video
    argv = ["--save-path", str(tmp_path / "ck"), "--steps", "2", "--frames", "29" if mode == "c1-size" else "9"]
    argv += {"full": [], "full-rgb": ["--rgb"], "head-only": ["--head-only"], "c1-size": ["--rgb"]}[mode]
    loss, shape = train_video.main(argv)
    assert np.isfinite(loss) and shape == (4, 54)
    sd = torch.load(tmp_path / "ck" / "1.pt", map_location="cpu")
    ref = __import__("deeplip_amd.weightgen", fromlist=["x"]).fill_state_dict({"frontend3D.0.weight": (64, 1, 5, 7, 7)}, prefix="video.")
    moved = not np.array_equal(sd["frontend3D.0.weight"].numpy(), ref["frontend3D.0.weight"])
    assert moved == (mode != "head-only")                 # the stem weights train only in full mode
    assert int(sd["frontend3D.1.num_batches_tracked"]) == (0 if mode == "head-only" else 2)


@pytest.mark.gpu
def test_train_video_graph_step(tmp_path, capsys):
    """--graph-step: the optimisation step recorded once and replayed as one HIP graph (deeplip_amd.train_plan): four steps = one
    eager + recording + replays, per-iteration cosine learning rate read from a device tensor.  (That a replayed step is bit-
    identical to the eager one is tests/test_train_video_gpu.py::test_recorded_training_step_is_bit_identical_to_eager.)"""
    import train_video
    loss, shape = train_video.main(["--save-path", str(tmp_path / "ck"), "--steps", "4", "--frames", "9"])      # recorded steps: the default
    assert np.isfinite(loss) and shape == (4, 54)
    out = capsys.readouterr().out
    assert out.count("(replayed)") == 3 and "recorded steps: {'shapes': 1, 'recorded': 1" in out
    assert train_video.train.last_stats["step_mode"] == "graph"
    lrs = [float(l.split(" lr ")[1].split()[0]) for l in out.splitlines() if l.startswith("epoch 0 it")]
    assert lrs == sorted(lrs, reverse=True) and lrs[0] < 3e-4 and lrs[-1] > 0           # the cosine schedule advanced every iteration
    sd = torch.load(tmp_path / "ck" / "1.pt", map_location="cpu")
    ref = __import__("deeplip_amd.weightgen", fromlist=["x"]).fill_state_dict({"frontend3D.0.weight": (64, 1, 5, 7, 7)}, prefix="video.")
    assert not np.array_equal(sd["frontend3D.0.weight"].numpy(), ref["frontend3D.0.weight"])
    assert int(sd["frontend3D.1.num_batches_tracked"]) == 4


@pytest.mark.gpu
def test_train_audio_test_mode(tmp_path, monkeypatch, arith_mode):
    import train_audio
    monkeypatch.chdir(tmp_path)
    tr = train_audio.Trainer(overrides={"data.test_speakers": 4, "data.test_utt_per_spk": 3, "data.trials": 200,
                                        "data.trial_targets": 40, "data.audio_frames": 120, "data.n_spk": 6,
                                        "data.utt_per_spk": 3, "train.bs": 8, "train.epoch": 2})
    w0 = tr.model.tdnn[0].context_layer.weight.detach().clone()
    tr._train()                                          # full-encoder training: model.train(), backward through every layer
    assert not torch.equal(w0, tr.model.tdnn[0].context_layer.weight.detach())
    assert int(tr.model.tdnn[0].bn.num_batches_tracked) == 4 and int(tr.model.bn2.num_batches_tracked) == 4
    assert np.isfinite(tr.last_epoch_stats["loss"]) and tr.last_epoch_stats["utt_per_s"] > 0
    assert tr.arith == arith_mode and tr.last_epoch_stats["step_mode"] == "graph"       # the recorded step is the default
    assert not tr.model.training                         # back in eval mode for extraction
    assert tr.model_average(2) == 2
    table = tr.extract_test_xv()
    assert table.emb.shape == (12, 512)
    assert np.abs(table.emb.norm(dim=1).cpu().numpy() - 1).max() < 1e-5
    eer, _ = tr.eer()
    assert 0 <= eer <= 1


@pytest.mark.gpu
def test_train_audio_frozen_encoder(tmp_path, monkeypatch):
    import train_audio
    monkeypatch.chdir(tmp_path)
    tr = train_audio.Trainer(overrides={"data.test_speakers": 4, "data.test_utt_per_spk": 3, "data.trials": 200,
                                        "data.trial_targets": 40, "data.audio_frames": 120, "data.n_spk": 6,
                                        "data.utt_per_spk": 3, "train.bs": 8, "train.epoch": 1, "train.freeze_encoder": True})
    w0 = tr.model.tdnn[0].context_layer.weight.detach().clone()
    c0 = tr.criterion.weights.detach().clone()
    tr._train()
    assert torch.equal(w0, tr.model.tdnn[0].context_layer.weight.detach())      # encoder untouched
    assert not torch.equal(c0, tr.criterion.weights.detach())                     # criterion trained


@pytest.mark.gpu
def test_train_audio_resnet_arch_with_aam_softmax(tmp_path, monkeypatch):
    """`arch: resnet` (config-only upstream, train_audio.py:64-66) + the AAM-softmax criterion: full-encoder training
    step on [B,1,F,T] features, then extraction and cosine EER."""
    import train_audio
    monkeypatch.chdir(tmp_path)
    tr = train_audio.Trainer(overrides={"model.arch": "resnet", "train.loss": "AAMSoftmax", "data.feat_dim": 24,
                                        "data.test_speakers": 3, "data.test_utt_per_spk": 2, "data.trials": 30,
                                        "data.trial_targets": 6, "data.audio_frames": 60, "data.n_spk": 5,
                                        "data.utt_per_spk": 2, "train.bs": 4, "train.epoch": 1})
    w0 = tr.model.conv1.weight.detach().clone()
    tr._train()
    assert not torch.equal(w0, tr.model.conv1.weight.detach()) and int(tr.model.bn1.num_batches_tracked) == 2
    assert np.isfinite(tr.last_epoch_stats["loss"])
    table = tr.extract_test_xv()
    assert table.emb.shape == (6, 256)
    eer, _ = tr.eer()
    assert 0 <= eer <= 1


@pytest.mark.gpu
def test_train_audio_loads_reference_style_checkpoints(tmp_path, monkeypatch):
    """Trainer.load takes the three checkpoint forms the reference produces (train_audio.py:209-232,259-266): its own
    epoch files (criterion as a state dict here), a file whose ``criterion`` is the pickled criterion MODULE, and
    ``net_avg.pth`` -- state_dict only, 'module.'-prefixed keys -- which model_average() now also writes."""
    import train_audio
    monkeypatch.chdir(tmp_path)
    ov = {"data.test_speakers": 3, "data.test_utt_per_spk": 2, "data.trials": 30, "data.trial_targets": 6,
          "data.audio_frames": 100, "data.n_spk": 5, "data.utt_per_spk": 2, "train.bs": 4, "train.epoch": 2}
    tr = train_audio.Trainer(overrides=ov)
    tr._train()
    assert tr.model_average(2) == 2
    avg_path = "exp/{}/net_avg.pth".format(tr.log_time)
    assert os.path.exists(avg_path)
    ck = torch.load(avg_path, map_location="cpu")
    assert set(ck) == {"state_dict"} and all(k.startswith("module.") for k in ck["state_dict"])
    want = {k: v.detach().cpu().clone() for k, v in tr.model.state_dict().items()}
    # (1) net_avg.pth: no criterion, no epoch
    tr2 = train_audio.Trainer(overrides=ov)
    tr2.load(avg_path)
    for k, v in tr2.model.state_dict().items():
        assert torch.equal(v.cpu(), want[k]), k
    # (2) criterion pickled as a Module, the way the reference saves it
    p2 = str(tmp_path / "ref_style.pth")
    torch.save({"epoch": 7, "state_dict": {"module." + k: v for k, v in want.items()}, "criterion": tr.criterion.cpu(),
                "optimizer": {}}, p2)
    tr3 = train_audio.Trainer(overrides=ov)
    with pytest.raises(RuntimeError, match="pickled objects"):      # a pickled Module executes code on load: refused by default
        tr3.load(p2)
    tr3.train_opts["allow_pickled_checkpoints"] = True
    tr3.load(p2)
    assert tr3.current_epoch == 7
    assert torch.equal(tr3.criterion.weights.detach().cpu(), tr.criterion.weights.detach().cpu())
    # (3) this trainer's own epoch file
    tr3.load("exp/{}/net_2.pth".format(tr.log_time))
    assert tr3.current_epoch == 2


@pytest.mark.gpu
def test_train_audio_reference_method_names_and_av_test_flow(tmp_path, monkeypatch):
    """The Trainer surface of train_audio.py:234-483 -- ``__call__``, ``extract_train_xv``, ``train_plda``,
    ``extract_test_xv_lomgrid`` / ``_grid``, ``load_finetune`` -- and its ``__main__`` scoring (``utils.eer(log_time)``,
    ``utils.eer_cos_lomgrid(log_time)``, ``utils.eer_plda_lomgrid(log_time)``: one argument, the store on disk)."""
    import train_audio
    from deeplip_amd import scoring_entry as se
    from models.audio_models import utils
    monkeypatch.chdir(tmp_path)
    ov = {"data.test_speakers": 5, "data.test_utt_per_spk": 4, "data.trials": 200, "data.trial_targets": 40, "data.audio_frames": 120,
          "data.n_spk": 6, "data.utt_per_spk": 3, "train.bs": 8, "train.epoch": 1}
    tr = train_audio.Trainer(overrides=ov)
    try:
        tr()                                                              # :473-483
        root = "exp/{}".format(tr.log_time)
        assert os.path.exists(root + "/net_1.pth")
        t = tr.extract_train_xv()                                          # :234-258: not normalised
        assert t.emb.shape == (18, 512) and np.abs(t.emb.norm(dim=1).cpu().numpy() - 1).max() > 1e-3
        assert np.load(root + "/train_xv/s0/s0_u00.npy").shape == (1, 512)
        tr.extract_test_xv()
        e, thr = utils.eer(tr.log_time)                                    # :499-503
        assert (e, thr) == tuple(tr.eer())
        tr.train_plda()                                                    # :298-341
        assert os.path.exists("exp/plda.pkl") and os.path.exists(root + "/dev_xv_lomgrid/s0_u00.npy")
        for name in ("lomgrid", "grid"):
            tab = getattr(tr, "extract_test_xv_" + name)()                 # :375-437
            assert np.abs(tab.emb.norm(dim=1).cpu().numpy() - 1).max() < 1e-5
            assert np.load(root + "/test_xv_{}/s0/s0_u00.npy".format(name)).shape == (1, 512)
            e, _ = getattr(utils, "eer_cos_" + name)(tr.log_time)
            ep, _ = getattr(utils, "eer_plda_" + name)(tr.log_time)
            assert 0 <= e <= 1 and 0 <= ep <= 1
        # load_finetune (:276-296): encoder frozen, optimizer over the criterion alone
        tr.train_opts["type"] = "sgd"
        tr.load_finetune(root + "/net_1.pth", None)
        assert tr.log_time == root.split("/")[1] and not any(p.requires_grad for p in tr.model.parameters())
        assert sum(len(g["params"]) for g in tr.optim.param_groups) == len(list(tr.criterion.parameters()))
        w0 = tr.model.tdnn[0].context_layer.weight.detach().clone(); c0 = tr.criterion.weights.detach().clone()
        tr.current_epoch = 0
        tr._train()
        assert torch.equal(w0, tr.model.tdnn[0].context_layer.weight.detach()) and not torch.equal(c0, tr.criterion.weights.detach())
    finally:
        se._process_paths.clear()


@pytest.mark.gpu
def test_train_video_eager_step_flag_and_ragged_shapes(tmp_path, capsys):
    """--eager-step keeps the loop of eager launches; and a run whose batches come in two padded lengths records one step per
    shape (pad_packed_collate pads to the batch's longest clip, dataset.py:123-139)."""
    import train_video
    loss, shape = train_video.main(["--save-path", str(tmp_path / "ck"), "--steps", "2", "--frames", "9", "--eager-step"])
    assert np.isfinite(loss) and "(replayed)" not in capsys.readouterr().out
    assert train_video.train.last_plan is None
    orig = train_video.synthetic_batch

    def two_lengths(args, it, rgb=False):
        import copy
        a = copy.copy(args)
        a.frames = 9 if it % 2 == 0 else 7
        return orig(a, it, rgb)

    train_video.synthetic_batch = two_lengths
    try:
        loss, shape = train_video.main(["--save-path", str(tmp_path / "ck2"), "--steps", "6", "--frames", "9"])
    finally:
        train_video.synthetic_batch = orig
    assert np.isfinite(loss)
    assert train_video.train.last_plan.summary() == {"shapes": 2, "recorded": 2, "eager_only": []}


@pytest.mark.gpu
def test_train_audio_crop_ladder_records_one_step_per_length(tmp_path, monkeypatch):
    """train.crop_frames: the collate's random crop (models/audio_models/datasets.py:112-115) drawn from a short ladder of
    lengths, one recorded step per rung; the margin a recorded step bakes in is part of its key."""
    import train_audio
    monkeypatch.chdir(tmp_path)
    tr = train_audio.Trainer(overrides={"data.test_speakers": 2, "data.test_utt_per_spk": 2, "data.audio_frames": 160, "data.n_spk": 6,
                                        "data.utt_per_spk": 3, "train.bs": 8, "train.epoch": 1, "train.steps_per_epoch": 12,
                                        "train.crop_frames": [100, 160], "train.margin": [0.2, 0.35]})
    ladder = tr.crop_ladder()
    assert ladder[-1] == 160 and ladder[0] >= 100 and all(b <= 1.1 * a + 4 for a, b in zip(ladder, ladder[1:])) and len(ladder) >= 4
    tr.current_epoch = 1
    tr._adjust_margin()
    tr._train_epoch()
    s1 = tr._steps.summary()
    assert 2 <= s1["shapes"] <= len(ladder) and s1["eager_only"] == []
    tr.current_epoch = 6                        # past epoch 5 the margin steps to its end value (train_audio.py:141-145): new recordings
    tr._adjust_margin()
    tr._train_epoch()
    assert tr._steps.summary()["shapes"] > s1["shapes"]
    assert np.isfinite(tr.last_epoch_stats["loss"]) and tr.last_epoch_stats["crop_ladder"] == [int(t) for t in ladder]
    tr.close()

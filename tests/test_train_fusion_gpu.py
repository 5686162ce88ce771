"""train_fusion.py entry point end to end (-m gpu): a short training run of the trainable tail on
synthetic A+V batches (loss must fall), then the av_test flow: batched extraction -> fused table ->
cosine trials -> EER; the extraction is checked against the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SMALL = {"train.bs": 16, "train.epoch": 2, "train.steps_per_epoch": 3, "data.n_spk": 6, "data.utt_per_spk": 4,
         "data.test_speakers": 4, "data.test_utt_per_spk": 3, "data.trials": 300, "data.trial_targets": 60,
         "data.video_frames": 9, "data.audio_frames": 120, "data.test_audio_frames": [60, 120], "data.test_video_frames": [5, 12],
         "data.test_clips_per_utt": 2, "test.batch": 16}


@pytest.fixture(autouse=True)
def _modes(arith_mode):
    """Every test of this file runs under arith ``auto`` (the shipped configuration: f16x3 + f32 re-run) and under exact ``f32``."""
    return arith_mode


@pytest.mark.parametrize("loss,fus", [("CrossEntropy", "linear"), ("LMCL", "concat")])
def test_train_mode_loss_falls(loss, fus, tmp_path, monkeypatch):
    import train_fusion
    monkeypatch.chdir(tmp_path)
    tr = train_fusion.Trainer("train", overrides=dict(SMALL, **{"train.loss": loss, "model.fusion": fus,
                                                                   "train.sgd.init_lr": 0.05}))
    tr.current_epoch = 1
    l0, _ = tr._train_epoch()
    tr.current_epoch = 2
    l1, a1 = tr._train_epoch()
    assert np.isfinite(l0) and np.isfinite(l1)
    assert tr.last_epoch_stats["step_mode"] == "graph"      # the shipped config: encoders' plan + the head's recorded step
    if loss == "CrossEntropy":     # LMCL at s=30 on random-weight embeddings is too noisy over 3 batches
        assert l1 < l0
    p = tr.save()
    tr.load(p)


def test_av_test_flow_matches_oracle(tmp_path, monkeypatch):
    import train_fusion
    from deeplip_amd import weightgen as wg
    from oracle import deeplip_oracle as O
    monkeypatch.chdir(tmp_path)
    tr = train_fusion.Trainer("av_test", overrides=dict(SMALL, **{"data.clips_per_utt": 2, "data.test_ragged": False}))
    table = tr.extract_test_xv_lomgrid()
    assert table.emb.shape == (12, 1024)
    eer, thr = tr.eer_cos(tr.lomgridtestset, tr.lomgrid_tables, "cos")
    assert 0.0 <= eer <= 1.0
    for mode in ("scorefusion", "featurefusion"):
        e, _ = tr.eer_cos(tr.lomgridtestset, tr.lomgrid_tables, mode)
        assert 0.0 <= e <= 1.0
    # oracle: same synthetic utterances, reference-style per-utterance loops
    ds = tr.lomgridtestset
    idx = list(range(len(ds)))
    vsd = O.to_torch_sd(wg.fill_state_dict({k: tuple(v.shape) for k, v in tr.model_video.state_dict().items()}, prefix="video."))
    asd = O.to_torch_sd(wg.fill_state_dict({k: tuple(v.shape) for k, v in tr.model_audio.state_dict().items()}, prefix="audio."))
    clips, ptr = ds.video(idx)
    with torch.no_grad():
        cm = O.video_time_mean(O.lipreading_features(vsd, torch.from_numpy(clips)))
        emv = O.video_group_mean(cm, ptr.tolist())
        xva, _ = O.speaker_extract_embedding(asd, torch.from_numpy(ds.audio(idx)), O.ETDNN_CONTEXT)
        ref = O.fuse_av(xva, emv)
    err = float((table.emb.cpu() - ref).abs().max() / ref.abs().max())
    assert err < 1e-4, err


def test_av_test_flow_on_a_ragged_list_matches_the_reference_loop(tmp_path, monkeypatch):
    """The shipped configuration: test utterances of differing duration, 1-2 clip files each (data.test_ragged).  The trainer
    extracts them in length-bucketed batches (deeplip_amd/extract.py); the oracle walks the list as the reference does --
    one utterance at a time, its audio and each of its clips at their own lengths, the clips' frame means averaged
    (train_fusion.py:334-358)."""
    import train_fusion
    from deeplip_amd import weightgen as wg
    from oracle import deeplip_oracle as O
    monkeypatch.chdir(tmp_path)
    tr = train_fusion.Trainer("av_test", overrides=dict(SMALL))
    ds = tr.lomgridtestset
    assert ds.ragged and len(set(ds.audio_len.tolist())) > 3 and len(set(ds.clip_len.tolist())) > 3
    table = tr.extract_test_xv_lomgrid()
    assert table.emb.shape == (12, 1024)
    st = tr.extract_stats
    assert st["valid_audio_frames"] == int(ds.audio_len.sum()) and st["valid_video_frames"] == int(ds.clip_len.sum())
    vsd = O.to_torch_sd(wg.fill_state_dict({k: tuple(v.shape) for k, v in tr.model_video.state_dict().items()}, prefix="video."))
    asd = O.to_torch_sd(wg.fill_state_dict({k: tuple(v.shape) for k, v in tr.model_audio.state_dict().items()}, prefix="audio."))
    rows = []
    with torch.no_grad():
        for i in range(len(ds)):
            xva, _ = O.speaker_extract_embedding(asd, torch.from_numpy(ds.audio_item(i)[None]), O.ETDNN_CONTEXT)          # :338
            em = torch.zeros(1, 512)
            cl = range(int(ds.clip_ptr[i]), int(ds.clip_ptr[i + 1]))
            for c in cl:                                                                                                   # :346-348
                em = em + O.video_time_mean(O.lipreading_features(vsd, torch.from_numpy(ds.clip_item(c)[None, None])))
            rows.append(O.fuse_av(xva, em / len(cl)))                                                                      # :349-358
    ref = torch.cat(rows)
    err = float((table.emb.cpu() - ref).abs().max() / ref.abs().max())
    assert err < 1e-4, err
    from conftest import assert_close_rel
    assert_close_rel(table.emb.cpu().numpy(), ref.numpy(), rtol=1e-4, what="ragged fused rows")
    for mode in ("cos", "scorefusion", "featurefusion"):
        e, _ = tr.eer_cos(ds, tr.lomgrid_tables, mode)
        assert 0.0 <= e <= 1.0


def test_main_flow_scores_through_the_reference_entry_points(tmp_path, monkeypatch):
    """train_fusion.py:430-469: extract, then ``utils.eer_cos_*(trainer.log_time)`` -- one argument, the store on disk.  The
    store the trainer leaves is the reference's (fused rows under exp/<run>/test_em/test_em_<set>/, x-vectors, lip clip
    files, trial lists); every entry point over it gives what the in-memory scoring of the same tables gives."""
    import os
    import train_fusion
    from models.fusion_models import utils
    monkeypatch.chdir(tmp_path)
    tr = train_fusion.Trainer("av_test", overrides=dict(SMALL, **{"data.clips_per_utt": 2}))
    try:
        for name, ds_name in (("lomgrid", "lomgridtestset"), ("grid", "gridtestset")):
            getattr(tr, "extract_test_xv_" + name)()
            tables = getattr(tr, name + "_tables")
            root = "exp/{}".format(tr.log_time)
            u0 = getattr(tr, ds_name).utt_ids[0].replace(".wav", ".npy")
            assert np.load(os.path.join(root, "test_em", "test_em_" + name, u0)).shape == (1, 1024)      # train_fusion.py:362-364
            assert np.load(os.path.join(root, "test_xv_" + name, u0)).shape == (1, 512)
            for fn, mode in (("eer_cos_{}", "cos"), ("eer_cos_{}_scorefusion", "scorefusion"), ("eer_cos_{}_featurefusion", "featurefusion")):
                got = getattr(utils, fn.format(name))(tr.log_time)
                want = tr.eer_cos(getattr(tr, ds_name), tables, mode)
                assert abs(got[0] - want[0]) < 1e-6 and abs(got[1] - want[1]) < 1e-5, (fn, name, got, want)
    finally:
        from deeplip_amd import scoring_entry as se
        se._process_paths.clear()


def test_model_average_is_the_mean_of_the_last_checkpoints(tmp_path, monkeypatch):
    """train_fusion.py:158-175."""
    import train_fusion
    monkeypatch.chdir(tmp_path)
    tr = train_fusion.Trainer("train", overrides=dict(SMALL, **{"train.sgd.init_lr": 0.05}))
    tr()                                                 # two epochs -> net_1.pth, net_2.pth
    a = torch.load("exp/{}/net_1.pth".format(tr.log_time), map_location="cpu")["state_dict"]
    b = torch.load("exp/{}/net_2.pth".format(tr.log_time), map_location="cpu")["state_dict"]
    avg = tr.model_average(2)
    ck = torch.load("exp/{}/net_avg.pth".format(tr.log_time), map_location="cpu")
    assert ck["epoch"] == 0 and "optimizer" in ck
    for k, v in tr.model_fusion.state_dict().items():
        if v.dtype.is_floating_point:
            want = ((a[k].double() + b[k].double()) / 2).float()
            assert torch.equal(v.cpu(), want) and torch.equal(ck["state_dict"][k], want), k
    assert any(not torch.equal(a[k], b[k]) for k in a)

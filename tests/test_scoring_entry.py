"""The reference's scoring entry points, called the way its trainers call them -- ``utils.<name>(trainer.log_time)``, ONE
argument (train_fusion.py:430-469, train_audio.py:499-543) -- over a synthetic on-disk store in the reference's formats: a
``.npy`` tree under ``exp/<run>/``, ``label utt1 utt2`` trial lists at the reference's default relative paths, lip-embedding
``.npz`` clip files, ``exp/plda.pkl``.  Each is compared with the oracle's restatement of the same function body
(models/fusion_models/utils.py:234-521)."""
import inspect
import os

import numpy as np
import pytest
import torch

from conftest import assert_close_rel

NAMES = ["eer", "eer_cos_lomgrid", "eer_cos_grid", "eer_plda_lomgrid", "eer_plda_grid", "eer_cos_lomgrid_scorefusion",
         "eer_cos_grid_scorefusion", "eer_cos_lomgrid_featurefusion", "eer_cos_grid_featurefusion"]


def _modules():
    import models.audio_models.utils as au
    import models.fusion_models.utils as fu
    return {"fusion": fu, "audio": au}


# ------------------------------------------------------------------------------------------ CPU: the surface
def test_entry_points_exist_with_the_reference_signature():
    """Every name of utils.py:234-521 in BOTH utils modules; exactly one positional parameter (`exp_dir`), everything else
    keyword-only -- the shape `utils.eer_cos_lomgrid(trainer.log_time)` needs (round 3's alias took two and raised TypeError)."""
    for mod in _modules().values():
        for n in NAMES + ["feature_normalize"]:
            assert callable(getattr(mod, n)), n
        for n in NAMES:
            ps = list(inspect.signature(getattr(mod, n)).parameters.values())
            assert ps[0].name == "exp_dir" and ps[0].kind == ps[0].POSITIONAL_OR_KEYWORD and ps[0].default is ps[0].empty
            assert all(p.kind == p.KEYWORD_ONLY and p.default is not p.empty for p in ps[1:]), n
            assert getattr(mod, n).__name__ == n


def test_default_paths_are_the_references_literals():
    from deeplip_amd import scoring_entry as se
    f, a = se.FUSION_DEFAULTS, se.AUDIO_DEFAULTS
    assert f["eer"] == {"trial": "task.txt", "sub": "test_xv"}
    assert f["eer_cos_lomgrid"] == {"trial": "data/data_audio/trial_lomgrid_2w.txt", "sub": "test_em_lomgrid"}
    assert a["eer_cos_lomgrid"] == {"trial": "data/trial/A_lomgrid_trial_2w", "sub": "test_xv_lomgrid"}
    assert a["eer_cos_grid"] == {"trial": "data/trial/A_grid_trial_2w", "sub": "test_xv_grid"}
    assert f["eer_plda_grid"]["plda"] == "exp/plda.pkl" and f["eer_plda_grid"]["sub"] == "test_xv_grid"
    assert f["eer_cos_grid_scorefusion"]["video_trial"].endswith("preprocessing/grid/grid_trial_2w")
    assert f["eer_cos_lomgrid_featurefusion"]["pattern"] == "spk/utt" and f["eer_cos_grid_featurefusion"]["pattern"] == "utt"
    assert se._pattern("spk/utt", "s2_l_bbal8p.wav") == "s2/s2_l_bbal8p" and se._pattern("utt", "s2_l_bbal8p.wav") == "s2_l_bbal8p"


def test_path_resolution_order(tmp_path, monkeypatch):
    """keyword > set_paths(name) > set_paths() > environment > literal; the embedding directory falls back to where the
    reference's fusion trainer writes (exp/<run>/test_em/test_em_lomgrid, train_fusion.py:332) when the directory its reader
    names (exp/<run>/test_em_lomgrid, utils.py:259) does not exist."""
    from deeplip_amd import scoring_entry as se
    monkeypatch.chdir(tmp_path)
    d = se.FUSION_DEFAULTS["eer_cos_lomgrid"]
    try:
        assert se._resolve("eer_cos_lomgrid", d, "run", {})["trial"] == d["trial"]
        assert se._resolve("eer_cos_lomgrid", d, "run", {})["emb_dir"] == "exp/run/test_em_lomgrid"
        os.makedirs("exp/run/test_em/test_em_lomgrid")
        assert se._resolve("eer_cos_lomgrid", d, "run", {})["emb_dir"] == "exp/run/test_em/test_em_lomgrid"
        os.makedirs("exp/run/test_em_lomgrid")
        assert se._resolve("eer_cos_lomgrid", d, "run", {})["emb_dir"] == "exp/run/test_em_lomgrid"
        monkeypatch.setenv("DLIP_TRIAL_LIST", "env.txt")
        assert se._resolve("eer_cos_lomgrid", d, "run", {})["trial"] == "env.txt"
        se.set_paths(trial="all.txt")
        assert se._resolve("eer_cos_lomgrid", d, "run", {})["trial"] == "all.txt"
        se.set_paths("eer_cos_lomgrid", trial="one.txt")
        assert se._resolve("eer_cos_lomgrid", d, "run", {})["trial"] == "one.txt"
        assert se._resolve("eer_cos_grid", se.FUSION_DEFAULTS["eer_cos_grid"], "run", {})["trial"] == "all.txt"
        assert se._resolve("eer_cos_lomgrid", d, "run", {"trial": "kw.txt"})["trial"] == "kw.txt"
        with pytest.raises(KeyError):
            se.set_paths("eer", nonsense="x")
    finally:
        se.set_paths("eer_cos_lomgrid"); se.set_paths()


def test_plda_file_round_trip_and_foreign_pickle(tmp_path, monkeypatch):
    """exp/plda.pkl written by this build is a numpy archive of arrays read with allow_pickle=False (loading it executes nothing);
    anything else -- the reference's joblib pickle of a `plda.Classifier` -- is loaded only on request (allow_pickle / the
    pickled-checkpoint switch), and a pickle of a package that is not installed is refused with an explanation."""
    import zipfile
    from deeplip_amd.plda import PLDA
    monkeypatch.delenv("DLIP_ALLOW_PICKLED_CHECKPOINTS", raising=False)
    r = np.random.default_rng(3)
    X = np.concatenate([c + 0.5 * r.normal(size=(8, 12)) for c in r.normal(size=(6, 12))])
    m = PLDA.fit(X, np.repeat(np.arange(6), 8), n_principal_components=5)
    p = str(tmp_path / "exp" / "plda.pkl")
    m.save(p)
    assert zipfile.is_zipfile(p) and all(v.dtype != object for v in np.load(p, allow_pickle=False).values())
    m2 = PLDA.load(p)
    np.testing.assert_array_equal(m.transform_np(X), m2.transform_np(X))
    np.testing.assert_array_equal(m.psi, m2.psi)
    # a foreign file: refused unless asked for
    foreign = str(tmp_path / "foreign.pkl")
    import joblib
    joblib.dump({"anything": 1}, foreign)
    with pytest.raises(RuntimeError, match="allow_pickle=True"):
        PLDA.load(foreign)
    # a stand-in for the reference's classifier object: taken over attribute by attribute (m, inv_A, Psi diagonal MATRIX, pca)
    class _Pca:  # noqa: E306
        mean_, components_ = m.pca_mean, m.pca_components
    import types
    fake = types.SimpleNamespace(model=types.SimpleNamespace(m=m.m, inv_A=m.inv_A, Psi=np.diag(m.psi), relevant_U_dims=m.relevant, pca=_Pca))
    import unittest.mock as mock
    with mock.patch("joblib.load", return_value=fake):
        m3 = PLDA.load(foreign, allow_pickle=True)
    np.testing.assert_allclose(m3.transform_np(X), m.transform_np(X), rtol=0, atol=0)
    monkeypatch.setenv("DLIP_ALLOW_PICKLED_CHECKPOINTS", "1")
    with mock.patch("joblib.load", side_effect=ModuleNotFoundError("No module named 'plda'")):
        with pytest.raises(RuntimeError, match="third-party `plda` package"):
            PLDA.load(foreign)


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU box only")
def test_entry_points_have_no_cpu_path(tmp_path, monkeypatch):
    from deeplip_amd._lib import DeepLipHipError
    monkeypatch.chdir(tmp_path)
    for mod in _modules().values():
        with pytest.raises(DeepLipHipError, match="no CPU path"):
            mod.eer_cos_lomgrid("run")


# ------------------------------------------------------------------------------------------ GPU: parity with the oracle
def _store(tmp_path, r, n_spk=9, per=5, D=512, trials=1500):
    """A run directory in the reference's formats.  Speaker-structured embeddings so that the EER is neither 0 nor 0.5."""
    from deeplip_amd import scoring
    spk = np.repeat(np.arange(n_spk), per)
    utts = [f"s{s}_l_u{i}.wav" for i, s in enumerate(spk)]                       # flat names like the A+V trial lists'
    ca, cv = r.normal(size=(n_spk, D)), r.normal(size=(n_spk, D))
    audio = (ca[spk] + 3.0 * r.normal(size=(len(spk), D)) + 0.7).astype(np.float32)
    fused = np.concatenate([audio, (cv[spk] + 3.0 * r.normal(size=(len(spk), D))).astype(np.float32)], 1)
    # lip store: 1-3 clip files per utterance, ragged frame counts, data [1, T, D]
    clips = {}
    for u, s in zip(utts, spk):       # utterance-level scatter (frame noise averages out over the clip files; this does not)
        off = 3.0 * r.normal(size=D)
        clips[u] = [(cv[s] + off + 3.0 * r.normal(size=(1, int(r.integers(3, 40)), D)) - 0.3).astype(np.float32) for _ in range(int(r.integers(1, 4)))]
    pairs, y = [], []
    for _ in range(trials):
        a, b = r.integers(0, len(utts), 2)
        if r.random() < 0.25:
            b = int(r.choice(np.where(spk == spk[a])[0]))
        pairs.append((utts[a], utts[b])); y.append(int(spk[a] == spk[b]))
    run = "Oct__4_10:00:00_2026"
    xv = audio / np.linalg.norm(audio, axis=1, keepdims=True)
    for sub, emb in (("test_xv", xv), ("test_xv_lomgrid", xv), ("test_xv_grid", xv), ("test_em_grid", fused)):
        scoring.EmbeddingTable(utts, torch.from_numpy(emb)).save_npy_tree(str(tmp_path / "exp" / run / sub))
    # the fused LombardGRID rows go where the reference's fusion trainer writes them (the reader's fallback)
    scoring.EmbeddingTable(utts, torch.from_numpy(fused)).save_npy_tree(str(tmp_path / "exp" / run / "test_em" / "test_em_lomgrid"))
    lines = "".join(f"{l} {a} {b}\n" for l, (a, b) in zip(y, pairs))
    for rel in ("task.txt", "data/data_audio/trial_lomgrid_2w.txt", "data/data_audio/trial_grid_2w.txt", "data/trial/A_lomgrid_trial_2w",
                "data/trial/A_grid_trial_2w"):
        os.makedirs(os.path.dirname(str(tmp_path / rel)), exist_ok=True)
        (tmp_path / rel).write_text(lines)
    return dict(run=run, utts=utts, spk=spk, audio=xv, fused=fused, clips=clips, pairs=pairs, y=np.asarray(y))


def _write_lip_store(tmp_path, st, kind):
    """<root>/embedding_x/<pattern>_<k>.npz; the reader is handed <root>/datasets_x/ and swaps the word (utils.py:361)."""
    from deeplip_amd import scoring_entry as se
    vdir = str(tmp_path / "site" / f"datasets_{kind.replace('/', '')}") + "/"
    for u, files in st["clips"].items():
        for k, a in enumerate(files):
            f = (vdir + se._pattern(kind, u) + f"_{k:03d}.npz").replace("datasets", "embedding")
            os.makedirs(os.path.dirname(f), exist_ok=True)
            np.savez_compressed(f, data=a)
    vtrial = str(tmp_path / "site" / f"video_trials_{kind.replace('/', '')}")
    with open(vtrial, "w") as fh:
        fh.writelines(f"{se._pattern(kind, a)}\t{se._pattern(kind, b)}\n" for a, b in st["pairs"])
    return vdir, vtrial


def _oracle_video(st):
    """utils.py:363-370: embedding = sum over the utterance's clip files of np.mean(data.squeeze(-3), 0), divided by their count."""
    out = []
    for u in st["utts"]:
        m = 0
        for a in st["clips"][u]:
            m = m + np.mean(a.squeeze(-3), 0)
        out.append(m / len(st["clips"][u]))
    return np.stack(out).astype(np.float32)


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["fusion", "audio"])
def test_entry_points_match_the_oracle(which, tmp_path, monkeypatch):
    from deeplip_amd.plda import PLDA
    from oracle import deeplip_oracle as O
    mod = _modules()[which]
    monkeypatch.chdir(tmp_path)
    st = _store(tmp_path, np.random.default_rng(11 if which == "fusion" else 12))
    run, y = st["run"], st["y"]
    idx = {u: i for i, u in enumerate(st["utts"])}
    ia = np.array([idx[a] for a, _ in st["pairs"]]); ib = np.array([idx[b] for _, b in st["pairs"]])

    def same(got, want_scores, what):
        e, t, s = got
        assert_close_rel(s, want_scores, what=what + " scores")                    # trial scores within 1e-4 relative
        we, wt = O.eer(y, want_scores)
        assert abs(e - we) < 1e-6 and abs(t - wt) < 1e-4 * max(1.0, abs(wt)), (what, e, we, t, wt)
        e1, t1 = getattr(mod, what)(run)                                           # exactly the reference's call
        assert (e1, t1) == (e, t)
        assert 0.02 < e < 0.48, (what, e)                                          # a non-degenerate operating point

    # cosine on the x-vector store / the fused store (the two modules read different directories: utils.py:254-260 of each)
    cos_x = O.cosine_trial_scores(st["audio"], ia, ib)
    cos_f = O.cosine_trial_scores(st["fused"], ia, ib)
    same(mod.eer(run, return_scores=True), cos_x, "eer")
    for name in ("eer_cos_lomgrid", "eer_cos_grid"):
        same(getattr(mod, name)(run, return_scores=True), cos_f if which == "fusion" else cos_x, name)

    # score fusion / feature fusion over the lip-embedding store
    vid = _oracle_video(st)
    for tag, kind in (("lomgrid", "spk/utt"), ("grid", "utt")):
        vdir, vtrial = _write_lip_store(tmp_path, st, kind)
        monkeypatch.setenv("DLIP_VIDEO_EMBEDDING_DIR", vdir)
        monkeypatch.setenv("DLIP_VIDEO_TRIAL_LIST", vtrial)
        same(getattr(mod, f"eer_cos_{tag}_scorefusion")(run, return_scores=True), O.score_fusion(st["audio"], vid, ia, ib),
             f"eer_cos_{tag}_scorefusion")
        same(getattr(mod, f"eer_cos_{tag}_featurefusion")(run, return_scores=True), O.feature_fusion_scores(st["audio"], vid, ia, ib),
             f"eer_cos_{tag}_featurefusion")

    # PLDA: exp/plda.pkl fitted on half of the speakers
    half = st["spk"] < 5
    model = PLDA.fit(st["audio"][half], st["spk"][half], n_principal_components=20)
    model.save("exp/plda.pkl")
    U, psi = model.transform_np(st["audio"]), model.psi[model.relevant]
    for name in ("eer_plda_lomgrid", "eer_plda_grid"):
        e, t, s = getattr(mod, name)(run, return_scores=True)
        ref = np.array([O.plda_llr_bruteforce(U[a], U[b], psi) for a, b in zip(ia[:80], ib[:80])])
        assert np.abs(s[:80] - ref).max() < 2e-3 * max(1.0, np.abs(ref).max()), name
        we, _ = O.eer(y, s)
        assert abs(e - we) < 1e-6 and getattr(mod, name)(run) == (e, t)


@pytest.mark.gpu
def test_missing_files_name_the_override(tmp_path, monkeypatch):
    import models.fusion_models.utils as fu
    monkeypatch.chdir(tmp_path)
    with pytest.raises(FileNotFoundError, match="DLIP_TRIAL_LIST"):
        fu.eer_cos_lomgrid("nothing")
    st = _store(tmp_path, np.random.default_rng(5), n_spk=3, per=2, trials=20)
    with pytest.raises(FileNotFoundError, match="DLIP_VIDEO_EMBEDDING_DIR"):
        fu.eer_cos_grid_featurefusion(st["run"], video_dir=str(tmp_path / "nowhere") + "/")

"""CPU tests of the host logic: state-dict schema of the drop-in modules equals the reference's
(manifest captured from the reference classes), BN folding identity, EER implementation vs the
golden (sklearn + scipy) value, module guards."""
import numpy as np
import pytest
import torch

from deeplip_amd import packing, scoring, weightgen as wg

TCN_OPTS = {"num_layers": 4, "kernel_size": [3, 5, 7], "dropout": 0.2, "dwpw": False, "width_mult": 1}


def shapes(m):
    return {k: list(v.shape) for k, v in m.state_dict().items()}


def test_video_state_dict_schema(manifest):
    from models.video_models.model import Lipreading
    net = Lipreading(hidden_dim=256, backbone_type="resnet", num_classes=54, relu_type="prelu",
                     tcn_options=TCN_OPTS, width_mult=1.0, extract_feats=True)
    assert shapes(net) == manifest["video_prelu_54"]
    assert len(net.state_dict()) == 343
    net = Lipreading(hidden_dim=256, num_classes=57, relu_type="relu", tcn_options=TCN_OPTS)
    assert shapes(net) == manifest["video_relu_57"]


def test_audio_and_heads_state_dict_schema(manifest):
    from models.audio_models.tdnn import SpeakerEmbNet
    from models.audio_models.loss import LMCL, CrossEntropy
    from models.fusion_models.model_fusion import model_fusion
    from oracle.deeplip_oracle import ETDNN_CONTEXT, TDNN_CONTEXT
    et = {"input_dim": 24, "hidden_dim": [512] * 9 + [1500], "context": ETDNN_CONTEXT, "tdnn_layers": 10,
          "embedding_dim": 512, "pooling": "statistic", "attention_hidden_size": 64, "bn_first": True}
    td = {"input_dim": 24, "hidden_dim": [512] * 4 + [1500], "context": TDNN_CONTEXT, "tdnn_layers": 5,
          "embedding_dim": 512, "pooling": "statistic", "attention_hidden_size": 64, "bn_first": True}
    assert shapes(SpeakerEmbNet({"arch": "etdnn", "etdnn": et})) == manifest["audio_etdnn_24"]
    assert len(SpeakerEmbNet({"arch": "etdnn", "etdnn": et}).state_dict()) == 84
    assert shapes(SpeakerEmbNet({"arch": "tdnn", "tdnn": td})) == manifest["audio_tdnn_24"]
    at = dict(td, pooling="attentive_statistic")
    assert shapes(SpeakerEmbNet({"arch": "tdnn", "tdnn": at})) == manifest["audio_tdnn_24_attentive"]
    assert shapes(LMCL(512, 57, 30, 0.2)) == manifest["lmcl_512_57"]
    assert shapes(CrossEntropy(1024, 57)) == manifest["ce_1024_57"]
    assert shapes(model_fusion(1024, 512, 57, False)) == manifest["linearfusion_1024_512"]
    with pytest.raises(NotImplementedError):
        SpeakerEmbNet({"arch": "tdnn", "tdnn": dict(td, pooling="mono_head_attention")})


def test_bn_fold_identity():
    from deeplip_amd.holders import BatchNormParams
    torch.manual_seed(0)
    bn = BatchNormParams(8)
    sd = wg.fill_state_dict({f"bn.{k}": tuple(v.shape) for k, v in bn.state_dict().items()})
    bn.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in sd.items()})
    w = torch.randn(8, 4, 3, 3); b = torch.randn(8); x = torch.randn(2, 4, 5, 5)
    ref = torch.nn.functional.batch_norm(torch.nn.functional.conv2d(x, w, b), bn.running_mean, bn.running_var,
                                         bn.weight, bn.bias, False, 0.0, bn.eps)
    wf, bf = packing.fold(w, b, bn)
    got = torch.nn.functional.conv2d(x.double(), wf, bf)
    assert float((got - ref.double()).abs().max()) < 1e-5


def test_holders_refuse_to_compute():
    from deeplip_amd.holders import ConvParams
    with pytest.raises(RuntimeError):
        ConvParams(4, 4, (3, 3))(torch.zeros(1, 4, 8, 8))


def test_models_refuse_train_mode_and_cpu():
    from models.fusion_models.model_fusion import model_fusion
    from deeplip_amd._lib import DeepLipHipError
    m = model_fusion(16, 8, 3, False)
    with pytest.raises(RuntimeError):
        m(torch.zeros(2, 16))           # train mode
    m.eval()
    with pytest.raises(DeepLipHipError):
        m(torch.zeros(2, 16))           # CPU tensor: no fallback


def test_eer_matches_golden(golden):
    g = golden["heads"]
    e, thr = scoring.eer_from_scores(g["eer_y_true"].astype(int), g["eer_scores"])
    assert abs(e - float(g["eer"])) < 1e-9
    assert abs(thr - float(g["eer_threshold"])) < 1e-7


def test_roc_curve_matches_sklearn():
    from sklearn.metrics import roc_curve as sk_roc
    r = np.random.Generator(np.random.PCG64(5))
    for n in (10, 257, 5000):
        y = r.integers(0, 2, n)
        s = np.round(r.standard_normal(n), 1).astype(np.float32)   # many ties
        a = scoring.roc_curve(y, s)
        b = sk_roc(y, s, pos_label=1)
        for u, v in zip(a, b):
            assert np.array_equal(np.asarray(u, dtype=np.float64), np.asarray(v, dtype=np.float64))


def test_trial_list_reader(tmp_path):
    p = tmp_path / "t.txt"
    p.write_text("1 a.wav b.wav\n0 a.wav c.wav\n")
    y, pairs = scoring.read_trial_list(str(p))
    assert y.tolist() == [1, 0] and pairs == [("a.wav", "b.wav"), ("a.wav", "c.wav")]


def test_embedding_store_npy_tree_roundtrip(tmp_path):
    """The reference's on-disk store (one [1, D] .npy per utterance, train_fusion.py:361-364; several clip files
    per utterance averaged into one row, utils.py:456-463) <-> the in-memory table, host side only."""
    ids = ["s1/a.wav", "s1/b.wav", "s2/c.wav"]
    emb = torch.arange(3 * 8, dtype=torch.float32).view(3, 8) / 7.0
    t = scoring.EmbeddingTable(ids, emb)
    t.save_npy_tree(str(tmp_path / "em"))
    assert np.load(tmp_path / "em" / "s1" / "a.npy").shape == (1, 8)
    back = scoring.EmbeddingTable.load_npy_tree(str(tmp_path / "em"), ids)
    assert back.utt_ids == ids and torch.equal(back.emb, emb)
    ia, ib = back.trial_indices([("s1/a.wav", "s2/c.wav"), ("s1/b.wav", "s1/a.wav")])
    assert ia.tolist() == [0, 1] and ib.tolist() == [2, 0]
    # several clip files per utterance are AVERAGED: that is arithmetic of the path, and there is no host-side version of it
    from deeplip_amd._lib import DeepLipHipError
    with pytest.raises(DeepLipHipError, match="runs on the GPU"):
        scoring.EmbeddingTable.load_npy_tree(str(tmp_path / "em"), ["u1", "u2"],
                                             groups={"u1": ["s1/a.wav", "s1/b.wav"], "u2": ["s2/c.wav"]})


def test_plda_fit_and_latent_space_properties():
    """Host side of the PLDA back-end (the `plda` package's algorithm, parity unpinned): in the
    fitted latent space the within-class covariance is ~I and the between-class covariance ~diag(psi); the affine
    map (with and without the PCA front) reproduces transform; dims are ordered by decreasing psi."""
    from deeplip_amd.plda import PLDA
    r = np.random.default_rng(3)
    K, n, D = 40, 30, 16
    centers = r.normal(size=(K, D)) * np.linspace(3.0, 0.2, D)
    X = np.concatenate([c + r.normal(size=(n, D)) * 0.7 for c in centers])
    y = np.repeat(np.arange(K), n)
    mixing = r.normal(size=(D, D))
    X = X @ mixing
    m = PLDA.fit(X, y)
    U = m.transform_np(X)
    within = np.concatenate([U[y == k] - U[y == k].mean(0) for k in range(K)])
    cw = within.T @ within / (len(within) - K)
    assert np.abs(cw - np.eye(cw.shape[0])).max() < 0.05
    means = np.stack([U[y == k].mean(0) for k in range(K)])
    psi = m.psi[m.relevant]
    assert np.all(np.diff(psi) <= 1e-12) and psi[0] > 5 * psi[-1]
    assert np.allclose(np.var(means, axis=0), psi + 1.0 / n, rtol=0.35)
    m2 = PLDA.fit(X, y, n_principal_components=8)
    assert m2.transform_np(X).shape[1] <= 8
    Wt, b = m2.affine()
    assert np.allclose(X[:5] @ Wt.T + b, m2.transform_np(X[:5]))


def test_plda_closed_form_matches_bruteforce_density():
    """The closed-form per-dimension LLR the kernel implements vs explicit Gaussian densities of the pair."""
    from oracle import deeplip_oracle as O
    r = np.random.default_rng(5)
    psi = np.abs(r.normal(size=12)) * 3 + 0.01
    for _ in range(5):
        u1, u2 = r.normal(size=12) * 1.5, r.normal(size=12) * 1.5
        closed = np.sum(np.log1p(psi) - 0.5 * np.log1p(2 * psi) + psi * (u1 + u2) ** 2 / (2 * (1 + 2 * psi))
                        - psi * (u1 ** 2 + u2 ** 2) / (2 * (1 + psi)))
        assert abs(closed - O.plda_llr_bruteforce(u1, u2, psi)) < 1e-9 * max(1.0, abs(closed))


def test_pmc_summary_parses_counter_csvs(tmp_path):
    """deeplip_amd.pmc (used by tools/pmc_summary.py and by bench.py's in-run counter passes): per-launch HBM bytes with the gfx950
    FETCH_SIZE x2 correction, kernel names as the bench line spells them."""
    from deeplip_amd import pmc
    d = tmp_path / "pass" / "host"
    d.mkdir(parents=True)
    name = "void (anonymous namespace)::conv_igemm_f16x3_dma_kernel<256, 128, 4, 2, 1, 3, 1, false, 0>((anonymous namespace)::ConvArgs)"
    rows = ["Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value"]
    for i in range(4):
        rows.append(f'{i},"{name}",FETCH_SIZE,1000')
        rows.append(f'{i},"{name}",WRITE_SIZE,500')
    (d / "1_counter_collection.csv").write_text("\n".join(rows) + "\n")
    res = pmc.summarise([str(tmp_path / "pass")], 2)
    k = res["conv_igemm_f16x3_dma_kernel<256,128>"]
    assert k["launches"] == 4 and k["launches_per_step"] == 2
    assert k["hbm_read_bytes_per_launch"] == 2 * 1000 * 1024 and k["hbm_write_bytes_per_launch"] == 500 * 1024
    assert pmc.short("void conv_win_f16x3_kernel<128, 64, 2, 2, true, 2>(ConvArgs)") == "conv_win_f16x3_kernel<128,64>"
    assert pmc.short("x::conv_igemm_f16x3_dma_kernel<256, 128, 4, 2, 2, 3, 1, false, 0>") == "conv_igemm_f16x3_dma_kernel<256,128,pool>"
    assert pmc.short("x::conv_igemm_f16x3_dma_kernel<256, 128, 4, 2, 1, 3, 1, true, 0>") == "conv_igemm_f16x3_dma_kernel<256,128,dual>"

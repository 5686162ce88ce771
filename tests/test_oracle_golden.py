"""Pin the CPU oracle (oracle/deeplip_oracle.py) against golden vectors captured from the
reference's own model classes (tests/golden/capture_golden.py).  CPU only.

Tolerance: 2e-5 relative-to-max for activations (the oracle calls the same ATen primitives but
composes them itself; differences are summation-order noise); integers bit-exact.
"""
import numpy as np
import pytest
import torch

from conftest import rel_err
from deeplip_amd import weightgen as wg
from oracle import deeplip_oracle as O

TOL = 2e-5


def sd_from(manifest, name, prefix):
    shapes = {k: tuple(v) for k, v in manifest[name].items()}
    return O.to_torch_sd(wg.fill_state_dict(shapes, prefix=prefix))


@pytest.fixture(scope="module")
def video_sd(manifest):
    return sd_from(manifest, "video_prelu_54", "video.")


@pytest.fixture(scope="module")
def video_out(video_sd):
    x = torch.from_numpy(wg.video_input(4))
    taps = {}
    with torch.no_grad():
        feats = O.lipreading_features(video_sd, x, "prelu", taps)
    return x, feats, taps


def test_video_features(golden, video_out):
    _, feats, _ = video_out
    g = golden["video"]
    assert rel_err(feats[:2].numpy(), g["feats_b2"]) < TOL
    assert rel_err(O.video_time_mean(feats).numpy(), g["feats_time_mean"]) < TOL


def test_video_taps(golden, video_out):
    _, _, taps = video_out
    g = golden["video"]
    b, t = [int(v) for v in g["tap_frame"]]
    assert rel_err(taps["stem_act"][b, :8, t].numpy(), g["tap_stem_act_c8"]) < TOL
    assert rel_err(taps["stem"][b, :, t].numpy(), g["tap_stem"]) < TOL
    for li in range(1, 5):
        assert rel_err(taps[f"layer{li}"][b * 29 + t].numpy(), g[f"tap_layer{li}"]) < TOL


def test_video_tcn_logits_ragged(golden, video_sd, video_out):
    x, _, _ = video_out
    g = golden["video"]
    lengths = [int(v) for v in g["tcn_lengths"]]
    xp = x.clone()
    for i, l in enumerate(lengths):
        xp[i, :, l:] = 0.0
    with torch.no_grad():
        logits = O.lipreading_logits(video_sd, xp, lengths)
        full = O.lipreading_logits(video_sd, x, [29] * 4)
    assert rel_err(logits.numpy(), g["tcn_logits"]) < TOL
    assert rel_err(full.numpy(), g["tcn_logits_full"]) < TOL
    assert np.array_equal(O.argmax_first(logits).numpy(), g["tcn_argmax"])


def test_video_relu_variant(golden, manifest):
    sd = sd_from(manifest, "video_relu_57", "video_relu.")
    x = torch.from_numpy(wg.video_input(1, frames=5, key="input.video.short"))
    with torch.no_grad():
        f = O.lipreading_features(sd, x, "relu")
    assert rel_err(f.numpy(), golden["video"]["relu_feats_t5"]) < TOL


def test_audio_etdnn(golden, manifest):
    g = golden["audio"]
    sd = sd_from(manifest, "audio_etdnn_24", "audio.")
    x = torch.from_numpy(wg.audio_input(4, 24, 300))
    taps = {}
    with torch.no_grad():
        xv, xa = O.speaker_extract_embedding(sd, x, O.ETDNN_CONTEXT, taps=taps)
        fwd = O.speaker_forward(sd, x, O.ETDNN_CONTEXT)
    assert rel_err(xv.numpy(), g["etdnn_xv"]) < TOL
    assert rel_err(xa.numpy(), g["etdnn_xa"]) < TOL
    assert rel_err(fwd.numpy(), g["etdnn_forward"]) < TOL
    assert rel_err(taps["pooled"].numpy(), g["etdnn_pooled"]) < TOL
    assert rel_err(taps["tdnn_out"][:, :16].numpy(), g["etdnn_tdnn_out_c16"]) < TOL
    x200 = torch.from_numpy(wg.audio_input(2, 24, 200, key="input.audio.t200"))
    with torch.no_grad():
        assert rel_err(O.speaker_extract_embedding(sd, x200, O.ETDNN_CONTEXT)[0].numpy(), g["etdnn_xv_t200"]) < TOL


def test_audio_variants(golden, manifest):
    g = golden["audio"]
    x = torch.from_numpy(wg.audio_input(4, 24, 300))
    with torch.no_grad():
        sd = sd_from(manifest, "audio_tdnn_24", "audio5.")
        assert rel_err(O.speaker_extract_embedding(sd, x, O.TDNN_CONTEXT)[0].numpy(), g["tdnn_xv"]) < TOL
        sd = sd_from(manifest, "audio_etdnn_80", "audio80.")
        x80 = torch.from_numpy(wg.audio_input(2, 80, 300, key="input.audio.f80"))
        assert rel_err(O.speaker_extract_embedding(sd, x80, O.ETDNN_CONTEXT)[0].numpy(), g["etdnn80_xv"]) < TOL
        sd = sd_from(manifest, "audio_tdnn_24_actfirst", "audio_nb.")
        assert rel_err(O.speaker_extract_embedding(sd, x, O.TDNN_CONTEXT, bn_first=False)[0].numpy(), g["tdnn_actfirst_xv"]) < TOL
        assert rel_err(O.speaker_forward(sd, x, O.TDNN_CONTEXT, bn_first=False).numpy(), g["tdnn_actfirst_forward"]) < TOL
        sd = sd_from(manifest, "audio_tdnn_24_attentive", "audio_at.")
        assert rel_err(O.speaker_extract_embedding(sd, x, O.TDNN_CONTEXT, pooling="attentive_statistic")[0].numpy(),
                       g["tdnn_attentive_xv"]) < TOL
    xp = torch.from_numpy(wg.gen("input.pool", (3, 40, 50)))
    assert rel_err(O.mean_std_pooling(xp).numpy(), g["meanstd_pool"]) < 1e-6


def test_heads(golden, manifest):
    g = golden["heads"]
    emb = torch.from_numpy(wg.gen("input.emb", (32, 512)))
    lab = torch.from_numpy(wg.labels(32, 57))
    sd = sd_from(manifest, "lmcl_512_57", "lmcl.")
    loss, logits = O.lmcl(emb, lab, sd["weights"], 30, 0.2)
    assert rel_err(logits.numpy(), g["lmcl_logits"]) < 1e-6
    assert abs(float(loss) - float(g["lmcl_loss"])) < 1e-5
    assert np.array_equal(O.argmax_first(logits).numpy(), g["lmcl_argmax"])
    assert float(g["lmcl_min_top2_gap"]) > 1e-3      # argmax golden is well separated
    emb2 = torch.from_numpy(wg.gen("input.emb1024", (32, 1024)))
    sd = sd_from(manifest, "ce_1024_57", "ce.")
    loss, logits = O.cross_entropy_head(emb2, lab, sd["fc.weight"], sd["fc.bias"])
    assert rel_err(logits.numpy(), g["ce_logits"]) < 1e-6
    assert abs(float(loss) - float(g["ce_loss"])) < 1e-5
    assert np.array_equal(O.argmax_first(logits).numpy(), g["ce_argmax"])
    sd = sd_from(manifest, "linearfusion_1024_512", "lf.")
    assert rel_err(O.linearfusion(sd, emb2, False).numpy(), g["linearfusion_extract0"]) < 1e-6
    assert rel_err(O.linearfusion(sd, emb2, True).numpy(), g["linearfusion_extract1"]) < 1e-6


def test_fusion_and_scores(golden):
    g = golden["heads"]
    xa = torch.from_numpy(wg.gen("input.xv_audio", (4, 512)))
    ev = torch.from_numpy(wg.gen("input.em_video", (4, 512)))
    assert rel_err(O.fuse_av(xa, ev).numpy(), g["fused_av"]) < 1e-6
    table = wg.gen("input.table", (40, 1024))
    ia, ib = g["trial_idx_a"], g["trial_idx_b"]
    assert np.abs(O.cosine_trial_scores(table, ia, ib) - g["trial_cos"]).max() < 2e-7
    ta = wg.gen("input.table_a", (40, 512)); tv = wg.gen("input.table_v", (40, 512))
    assert np.abs(O.score_fusion(ta, tv, ia, ib) - g["trial_scorefusion"]).max() < 2e-7


def test_eer(golden):
    g = golden["heads"]
    e, thr = O.eer(g["eer_y_true"].astype(int), [np.array([s]) for s in g["eer_scores"]])
    assert abs(e - float(g["eer"])) < 1e-12
    assert abs(thr - float(g["eer_threshold"])) < 1e-12


def test_ingest_rgb_shape():
    u8 = (wg.gen("input.rgb", (1, 3, 3, 8, 8), kind="uniform") * 255).astype(np.uint8)
    y = O.ingest_rgb_u8(u8)
    assert y.shape == (1, 1, 3, 8, 8) and y.dtype == np.float32
    # a gray image (R=G=B=v) maps to (v/255 - 0.421)/0.165
    v = np.full((1, 1, 3, 2, 2), 128, np.uint8)
    assert np.allclose(O.ingest_rgb_u8(v), (128 / 255 - 0.421) / 0.165, atol=1e-5)


def test_train_step_oracle(golden, manifest):
    """Two SGD steps of fusion head + criterion (config C5) -- oracle vs the reference classes."""
    g = golden["train"]
    B = 60
    xa = torch.from_numpy(wg.gen("train.xv_audio", (B, 512)))
    ev = torch.from_numpy(wg.gen("train.em_video", (B, 512)))
    lab = torch.from_numpy(wg.labels(B, 57))
    for tag, cname in (("ce", "ce_512_57"), ("lmcl", "lmcl_512_57")):
        lf = sd_from(manifest, "linearfusion_1024_512", "train.lf.")
        if tag == "ce":
            cs = O.to_torch_sd(wg.fill_state_dict({"fc.weight": (57, 512), "fc.bias": (57,)}, prefix="train.ce."))
        else:
            cs = sd_from(manifest, "lmcl_512_57", "train.lmcl.")
        train_keys = [k for k in lf if "running" not in k and "num_batches" not in k]
        params = [lf[k].requires_grad_() for k in train_keys] + [v.requires_grad_() for v in cs.values()]
        bufs = [None] * len(params)
        for step in range(2):
            out = O.linearfusion_train(lf, torch.cat([xa, ev], 1))
            if tag == "ce":
                loss, logits = O.cross_entropy_head(out, lab, cs["fc.weight"], cs["fc.bias"])
            else:
                loss, logits = O.lmcl(out, lab, cs["weights"], 30, 0.2)
            loss.backward()
            if step == 0:
                assert abs(float(loss) - float(g[f"{tag}_loss0"])) < 1e-5
                assert rel_err(logits.detach().numpy(), g[f"{tag}_logits0"]) < 1e-5
                assert np.array_equal(O.argmax_first(logits).numpy(), g[f"{tag}_argmax0"])
                assert rel_err(lf["fc2.weight"].grad[:8].numpy(), g[f"{tag}_grad_fc2_w_rows8"]) < 1e-4
                cw = cs["fc.weight"] if tag == "ce" else cs["weights"]
                assert rel_err(cw.grad.numpy(), g[f"{tag}_grad_crit_w"]) < 1e-4
            O.sgd_momentum_step(params, bufs, 0.5, 0.9, 1e-5)
        assert abs(float(loss) - float(g[f"{tag}_loss1"])) < 1e-3 * max(1.0, float(g[f"{tag}_loss1"]))
        assert rel_err(lf["fc2.weight"].detach()[:8].numpy(), g[f"{tag}_after2_fc2_w_rows8"]) < 1e-4
        assert rel_err(lf["bn1.running_var"].numpy(), g[f"{tag}_after2_bn1_running_var"]) < 1e-5


ATRAIN_CONTEXT = [[-2, -1, 0, 1, 2], [-2, 0, 2], [-3, 0, 3], [0], [0]]


def atrain_shapes():
    s, cin = {}, 24
    for i, (k, cout) in enumerate(zip((5, 3, 3, 1, 1), [512] * 4 + [1500])):
        s[f"tdnn.{i}.context_layer.weight"] = (cout, cin, k); s[f"tdnn.{i}.context_layer.bias"] = (cout,)
        for n in ("weight", "bias", "running_mean", "running_var"):
            s[f"tdnn.{i}.bn.{n}"] = (cout,)
        s[f"tdnn.{i}.bn.num_batches_tracked"] = ()
        cin = cout
    s["fc1.weight"] = (512, 3000); s["fc1.bias"] = (512,); s["fc2.weight"] = (512, 512); s["fc2.bias"] = (512,)
    for b in ("bn1", "bn2"):
        for n in ("weight", "bias", "running_mean", "running_var"):
            s[f"{b}.{n}"] = (512,)
        s[f"{b}.num_batches_tracked"] = ()
    return s


def test_audio_encoder_train_step_oracle(golden):
    """Two SGD steps of the FULL speech encoder + LMCL (train_audio.py:167-183) -- the oracle's train-mode
    restatement under torch autograd vs values captured from the reference classes."""
    g = golden["audio_train"]
    p = O.to_torch_sd(wg.fill_state_dict(atrain_shapes(), prefix="atrain.audio."))
    cw = O.to_torch_sd(wg.fill_state_dict({"weights": (57, 512)}, prefix="atrain.lmcl."))["weights"]
    x = torch.from_numpy(wg.audio_input(8, 24, 120, key="atrain.x"))
    lab = torch.from_numpy(wg.labels(8, 57))
    tkeys = [k for k in p if "running" not in k and "num_batches" not in k]
    params = [p[k].requires_grad_() for k in tkeys] + [cw.requires_grad_()]
    bufs = [None] * len(params)
    for step in range(2):
        out = O.speaker_forward_train(p, x, ATRAIN_CONTEXT)
        loss, logits = O.lmcl(out, lab, cw, 30, 0.2)
        loss.backward()
        if step == 0:
            assert abs(float(loss) - float(g["loss0"])) < 1e-5 * float(g["loss0"])
            assert rel_err(out.detach().numpy(), g["output0"]) < 1e-5
            assert np.array_equal(O.argmax_first(logits).numpy(), g["argmax0"])
            assert rel_err(p["tdnn.0.context_layer.weight"].grad.numpy(), g["grad_tdnn0_w"]) < 1e-4
            assert rel_err(p["tdnn.2.context_layer.weight"].grad[:4].numpy(), g["grad_tdnn2_w_rows4"]) < 1e-4
            assert rel_err(p["fc1.weight"].grad[:4].numpy(), g["grad_fc1_w_rows4"]) < 1e-4
        O.sgd_momentum_step(params, bufs, 0.01, 0.9, 1e-5)
    assert abs(float(loss) - float(g["loss1"])) < 1e-4 * float(g["loss1"])
    assert rel_err(p["tdnn.0.context_layer.weight"].detach().numpy(), g["after2_tdnn0_w"]) < 1e-5
    assert rel_err(p["tdnn.1.bn.running_var"].numpy(), g["after2_tdnn1_running_var"]) < 1e-5


def test_lipreading_train_step_oracle_vs_reference_golden(golden):
    """One training step of the FULL lip-clip model (train_video.py:129-147; TCN dropout 0) -- the oracle's
    train-mode restatement under torch autograd vs values captured from the reference class: loss, logits, argmax,
    gradients from the stem to the classifier, BatchNorm running statistics."""
    import torch.nn.functional as F
    g = golden["video_train"]
    from models.video_models.model import Lipreading
    tcn = {"num_layers": 4, "kernel_size": [3, 5, 7], "dropout": 0.0, "dwpw": False, "width_mult": 1}
    shapes = {k: tuple(v.shape) for k, v in Lipreading(num_classes=54, relu_type="prelu", tcn_options=tcn).state_dict().items()}
    p = O.to_torch_sd(wg.fill_state_dict(shapes, prefix="vtrain.video."))
    for k, v in p.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_()
    x = torch.from_numpy(wg.video_input(2, frames=7, key="vtrain.x"))
    lab = torch.from_numpy(wg.labels(2, 54))
    logits = O.lipreading_logits_train(p, x, [7, 5])
    loss = F.cross_entropy(logits, lab)
    loss.backward()
    assert abs(float(loss) - float(g["loss0"])) < 1e-5 * float(g["loss0"])
    assert rel_err(logits.detach().numpy(), g["logits0"]) < 1e-5
    assert np.array_equal(O.argmax_first(logits).numpy(), g["argmax0"])
    assert rel_err(p["frontend3D.0.weight"].grad.numpy(), g["grad_stem_w"]) < 1e-4
    assert rel_err(p["trunk.layer2.0.downsample.0.weight"].grad[:4].numpy(), g["grad_l2_0_down_w_rows4"]) < 1e-4
    assert rel_err(p["trunk.layer4.1.conv2.weight"].grad[:2].numpy(), g["grad_l4_1_conv2_w_rows2"]) < 1e-4
    assert rel_err(p["tcn.mb_ms_tcn.network.0.cbcr0_1.conv.weight"].grad[:4].numpy(), g["grad_tcn0_cbcr0_1_w_rows4"]) < 1e-4
    assert rel_err(p["tcn.tcn_output.weight"].grad[:4].numpy(), g["grad_tcn_out_w_rows4"]) < 1e-4
    # running statistics after ONE forward (the optimiser step does not touch buffers)
    assert rel_err(p["frontend3D.1.running_var"].numpy(), g["after1_stem_running_var"]) < 1e-5
    assert rel_err(p["tcn.mb_ms_tcn.network.0.cbcr0_2.batchnorm.running_var"].numpy(), g["after1_tcn0_cbcr0_2_running_var"]) < 1e-5

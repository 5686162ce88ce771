#!/usr/bin/env python3
"""Capture golden vectors from the REFERENCE's own model classes (run in the build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/capture_golden.py

Imports ``models.*`` from /root/reference (read-only), overwrites every parameter / buffer with
``deeplip_amd.weightgen.fill_state_dict`` (name-keyed, seed 1), runs the seeded synthetic inputs
of SURVEY.md §8c and writes small ``.npz`` fixtures + a key/shape manifest next to this script.
Only DATA is written (inputs are regenerated from the same generator, outputs are stored);
no reference source travels.  The reference never runs on the GPU box.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("DEEPLIP_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from deeplip_amd import weightgen as wg  # noqa: E402

# import the reference package under its own name, without letting our drop-in `models/` shadow it
# (the reference's models/ is a namespace package: any regular package named `models` on sys.path
# would win, so this repo's root and the cwd are taken off the path first)
sys.path = [p for p in sys.path if os.path.realpath(p or os.getcwd()) != os.path.realpath(ROOT)]
sys.path.insert(0, REF)
for m in [k for k in sys.modules if k == "models" or k.startswith("models.")]:
    del sys.modules[m]
from models.video_models.model import Lipreading  # noqa: E402
from models.audio_models.tdnn import SpeakerEmbNet  # noqa: E402
from models.audio_models.pooling import MeanStdPooling, AttentiveStatPooling  # noqa: E402
from models.audio_models.loss import LMCL, CrossEntropy  # noqa: E402
from models.fusion_models.model_fusion import model_fusion  # noqa: E402
import models  # noqa: E402
assert os.path.realpath(os.path.dirname(models.__path__[0] if hasattr(models, "__path__") else models.__file__)).startswith(os.path.realpath(REF)), "reference not imported"

torch.set_num_threads(8)
torch.manual_seed(1)

VERSIONS = {"torch": torch.__version__, "numpy": np.__version__}
try:
    import sklearn, scipy
    VERSIONS["sklearn"] = sklearn.__version__
    VERSIONS["scipy"] = scipy.__version__
except Exception:  # pragma: no cover
    pass


def fill(module, prefix=""):
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    sd = wg.fill_state_dict(shapes, prefix=prefix)
    module.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    module.eval()
    return shapes


TCN_OPTS = {"num_layers": 4, "kernel_size": [3, 5, 7], "dropout": 0.2, "dwpw": False, "width_mult": 1}
ETDNN = {"input_dim": 24, "hidden_dim": [512] * 9 + [1500],
         "context": [[-2, -1, 0, 1, 2], [0], [-2, 0, 2], [0], [-3, 0, 3], [0], [-4, 0, 4], [0], [0], [0]],
         "tdnn_layers": 10, "fc_layers": 3, "embedding_dim": 512, "pooling": "statistic",
         "attention_hidden_size": 64, "bn_first": True}
TDNN = {"input_dim": 24, "hidden_dim": [512, 512, 512, 512, 1500],
        "context": [[-2, -1, 0, 1, 2], [-2, 0, 2], [-3, 0, 3], [0], [0]],
        "tdnn_layers": 5, "fc_layers": 3, "embedding_dim": 512, "pooling": "statistic",
        "attention_hidden_size": 64, "bn_first": True}

manifest = {"versions": VERSIONS}


def video():
    out = {}
    # ---- extract_feats=True, prelu, 54 classes (config C1/C2 shapes) ----
    net = Lipreading(hidden_dim=256, backbone_type="resnet", num_classes=54, relu_type="prelu",
                     tcn_options=TCN_OPTS, width_mult=1.0, extract_feats=True)
    manifest["video_prelu_54"] = {k: list(s) for k, s in fill(net, "video.").items()}
    x = torch.from_numpy(wg.video_input(4))
    taps = {}
    hooks = []
    hooks.append(net.frontend3D[2].register_forward_hook(lambda m, i, o: taps.__setitem__("stem_act", o.detach())))
    hooks.append(net.frontend3D.register_forward_hook(lambda m, i, o: taps.__setitem__("stem", o.detach())))
    for li in range(1, 5):
        hooks.append(getattr(net.trunk, f"layer{li}").register_forward_hook(
            lambda m, i, o, li=li: taps.__setitem__(f"layer{li}", o.detach())))
    with torch.no_grad():
        feats = net(x, lengths=[29] * 4)
    for h in hooks:
        h.remove()
    out["feats_b2"] = feats[:2].numpy()                      # [2,29,512]
    out["feats_time_mean"] = feats.mean(1).numpy()           # [4,512]
    # taps for one frame (clip 1, t=7 -> row index 1*29+7 after the fold)
    b, t = 1, 7
    out["tap_frame"] = np.array([b, t])
    out["tap_stem_act_c8"] = taps["stem_act"][b, :8, t].numpy()      # [8,44,44] pre-pool, 8 channels
    out["tap_stem"] = taps["stem"][b, :, t].numpy()                  # [64,22,22]
    for li in range(1, 5):
        out[f"tap_layer{li}"] = taps[f"layer{li}"][b * 29 + t].numpy()
    # ---- classifier path (extract_feats=False), ragged lengths with zero padding ----
    net.extract_feats = False
    lengths = [29, 29, 20, 11]
    xp = x.clone()
    for i, l in enumerate(lengths):
        xp[i, :, l:] = 0.0          # pad_packed_collate zero-pads raw frames (dataset.py:131-134)
    with torch.no_grad():
        logits = net(xp, lengths=lengths)
    out["tcn_lengths"] = np.array(lengths)
    out["tcn_logits"] = logits.numpy()                       # [4,54]
    out["tcn_argmax"] = torch.max(logits, 1)[1].numpy()
    with torch.no_grad():
        out["tcn_logits_full"] = net(x, lengths=[29] * 4).numpy()
    # ---- relu variant, short clip ----
    net2 = Lipreading(hidden_dim=256, backbone_type="resnet", num_classes=57, relu_type="relu",
                      tcn_options=TCN_OPTS, width_mult=1.0, extract_feats=True)
    manifest["video_relu_57"] = {k: list(s) for k, s in fill(net2, "video_relu.").items()}
    x2 = torch.from_numpy(wg.video_input(1, frames=5, key="input.video.short"))
    with torch.no_grad():
        out["relu_feats_t5"] = net2(x2, lengths=[5]).numpy()  # [1,5,512]
    np.savez_compressed(os.path.join(HERE, "video_golden.npz"), **out)
    print("video:", {k: v.shape for k, v in out.items()})


def audio():
    out = {}
    opts = {"arch": "etdnn", "etdnn": dict(ETDNN), "tdnn": dict(TDNN)}
    net = SpeakerEmbNet(opts)
    manifest["audio_etdnn_24"] = {k: list(s) for k, s in fill(net, "audio.").items()}
    x = torch.from_numpy(wg.audio_input(4, 24, 300))
    taps = {}
    h1 = net.tdnn.register_forward_hook(lambda m, i, o: taps.__setitem__("tdnn_out", o.detach()))
    h2 = net.pooling.register_forward_hook(lambda m, i, o: taps.__setitem__("pooled", o.detach().clone()))
    with torch.no_grad():
        xv, x_a = net.extract_embedding(x)
        fwd = net(x)
    h1.remove(); h2.remove()
    out["etdnn_xv"] = xv.numpy(); out["etdnn_xa"] = x_a.numpy(); out["etdnn_forward"] = fwd.numpy()
    out["etdnn_pooled"] = taps["pooled"].numpy()                  # [4,3000]
    out["etdnn_tdnn_out_c16"] = taps["tdnn_out"][:, :16].numpy()  # [4,16,278]
    # ragged: 200-frame utterance (train collate draws 200-400, datasets.py:112-115)
    x200 = torch.from_numpy(wg.audio_input(2, 24, 200, key="input.audio.t200"))
    with torch.no_grad():
        out["etdnn_xv_t200"] = net.extract_embedding(x200)[0].numpy()
    # 5-layer TDNN
    opts5 = {"arch": "tdnn", "etdnn": dict(ETDNN), "tdnn": dict(TDNN)}
    net5 = SpeakerEmbNet(opts5)
    manifest["audio_tdnn_24"] = {k: list(s) for k, s in fill(net5, "audio5.").items()}
    with torch.no_grad():
        out["tdnn_xv"] = net5.extract_embedding(x)[0].numpy()
    # north-star input_dim = 80
    o80 = dict(ETDNN); o80["input_dim"] = 80
    net80 = SpeakerEmbNet({"arch": "etdnn", "etdnn": o80})
    manifest["audio_etdnn_80"] = {k: list(s) for k, s in fill(net80, "audio80.").items()}
    x80 = torch.from_numpy(wg.audio_input(2, 80, 300, key="input.audio.f80"))
    with torch.no_grad():
        out["etdnn80_xv"] = net80.extract_embedding(x80)[0].numpy()
    # bn_first = False
    onb = dict(TDNN); onb["bn_first"] = False
    netnb = SpeakerEmbNet({"arch": "tdnn", "tdnn": onb})
    manifest["audio_tdnn_24_actfirst"] = {k: list(s) for k, s in fill(netnb, "audio_nb.").items()}
    with torch.no_grad():
        out["tdnn_actfirst_xv"] = netnb.extract_embedding(x)[0].numpy()
        out["tdnn_actfirst_forward"] = netnb(x).numpy()
    # attentive statistic pooling
    oat = dict(TDNN); oat["pooling"] = "attentive_statistic"
    netat = SpeakerEmbNet({"arch": "tdnn", "tdnn": oat})
    shapes = {k: tuple(v.shape) for k, v in netat.state_dict().items()}
    sd = wg.fill_state_dict(shapes, prefix="audio_at.")
    netat.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); netat.eval()
    manifest["audio_tdnn_24_attentive"] = {k: list(s) for k, s in shapes.items()}
    with torch.no_grad():
        out["tdnn_attentive_xv"] = netat.extract_embedding(x)[0].numpy()
    # standalone pooling modules
    xp = torch.from_numpy(wg.gen("input.pool", (3, 40, 50)))
    out["meanstd_pool"] = MeanStdPooling()(xp).numpy()
    np.savez_compressed(os.path.join(HERE, "audio_golden.npz"), **out)
    print("audio:", {k: v.shape for k, v in out.items()})


def heads():
    out = {}
    emb = torch.from_numpy(wg.gen("input.emb", (32, 512)))
    lab = torch.from_numpy(wg.labels(32, 57))
    crit = LMCL(512, 57, 30, 0.2)
    manifest["lmcl_512_57"] = {k: list(s) for k, s in fill(crit, "lmcl.").items()}
    with torch.no_grad():
        loss, logits = crit(emb, lab)
    out["lmcl_loss"] = loss.numpy(); out["lmcl_logits"] = logits.numpy()
    out["lmcl_argmax"] = torch.max(logits, 1)[1].numpy()
    top2 = torch.topk(logits, 2, dim=1)[0]
    out["lmcl_min_top2_gap"] = (top2[:, 0] - top2[:, 1]).min().numpy()
    ce = CrossEntropy(1024, 57)
    manifest["ce_1024_57"] = {k: list(s) for k, s in fill(ce, "ce.").items()}
    emb2 = torch.from_numpy(wg.gen("input.emb1024", (32, 1024)))
    with torch.no_grad():
        loss, logits = ce(emb2, lab)
    out["ce_loss"] = loss.numpy(); out["ce_logits"] = logits.numpy()
    out["ce_argmax"] = torch.max(logits, 1)[1].numpy()
    for ef in (False, True):
        lf = model_fusion(1024, 512, 57, ef)
        manifest["linearfusion_1024_512"] = {k: list(s) for k, s in fill(lf, "lf.").items()}
        with torch.no_grad():
            out[f"linearfusion_extract{int(ef)}"] = lf(emb2).numpy()
    # test-time fusion exactly as train_fusion.py:233-238,353-358 (method body re-typed here
    # because the Trainer class cannot be imported: module-level side effects)
    xa = torch.from_numpy(wg.gen("input.xv_audio", (4, 512)))
    ev = torch.from_numpy(wg.gen("input.em_video", (4, 512)))

    def feature_normalize(data):
        mu = torch.mean(data, axis=1); std = torch.std(data, axis=1)
        return ((data.transpose(0, 1) - mu) / std).transpose(0, 1)
    out["fused_av"] = torch.cat([feature_normalize(xa), feature_normalize(ev)], dim=1).numpy()
    # sklearn cosine on 64 synthetic trials over a 40-utterance table
    from sklearn.metrics.pairwise import cosine_similarity
    table = wg.gen("input.table", (40, 1024))
    r = np.random.Generator(np.random.PCG64(7))
    ia = r.integers(0, 40, 64); ib = r.integers(0, 40, 64)
    out["trial_idx_a"] = ia; out["trial_idx_b"] = ib
    out["trial_cos"] = np.concatenate([cosine_similarity(table[a].reshape(1, -1), table[b].reshape(1, -1)).reshape(-1)
                                       for a, b in zip(ia, ib)])
    # score fusion (utils.py:343-377): 0.5*sklearn cos(audio) + 0.5*torch cos(video, eps=1e-8)
    ta = wg.gen("input.table_a", (40, 512)); tv = wg.gen("input.table_v", (40, 512))
    import torch.nn.functional as F
    sa = np.concatenate([cosine_similarity(ta[a].reshape(1, -1), ta[b].reshape(1, -1)).reshape(-1) for a, b in zip(ia, ib)])
    sv = np.array([F.cosine_similarity(torch.from_numpy(tv[a]), torch.from_numpy(tv[b]), dim=0, eps=1e-8).numpy()
                   for a, b in zip(ia, ib)])
    out["trial_scorefusion"] = 0.5 * sa + 0.5 * sv
    # EER on a trial_grid_v1-shaped synthetic score set: 4000 target / 16000 non-target
    from scipy.interpolate import interp1d
    from scipy.optimize import brentq
    from sklearn.metrics import roc_curve
    y_true = np.concatenate([np.ones(4000, dtype=np.int64), np.zeros(16000, dtype=np.int64)])
    sc = np.concatenate([wg.gen("input.tar", (4000,)) * 0.15 + 0.55, wg.gen("input.non", (16000,)) * 0.15 + 0.10]).astype(np.float32)
    perm = np.random.Generator(np.random.PCG64(11)).permutation(20000)
    y_true, sc = y_true[perm], sc[perm]
    y_pred = [np.array([s]) for s in sc]                       # list of (1,) arrays as in utils.py:262
    fpr, tpr, thr = roc_curve(list(y_true), y_pred, pos_label=1)
    e = brentq(lambda x: 1. - x - interp1d(fpr, tpr)(x), 0., 1.)
    out["eer_y_true"] = y_true.astype(np.int8); out["eer_scores"] = sc
    out["eer"] = np.float64(e); out["eer_threshold"] = np.float64(interp1d(fpr, thr)(e))
    np.savez_compressed(os.path.join(HERE, "heads_golden.npz"), **out)
    print("heads:", {k: np.shape(v) for k, v in out.items()})


def train():
    """One optimisation step of the trainable tail exactly as train_fusion.py:286-299 composes it
    (frozen-encoder outputs are synthetic [60,512] embeddings; bs 60, SGD lr 0.5 / momentum 0.9 /
    wd 1e-5: conf/fusion_config.yaml:91-99), for CrossEntropy and for LMCL (audio_config: s=30, m=0.2)."""
    out = {}
    B = 60
    xa = torch.from_numpy(wg.gen("train.xv_audio", (B, 512)))
    ev = torch.from_numpy(wg.gen("train.em_video", (B, 512)))
    lab = torch.from_numpy(wg.labels(B, 57))
    for tag, make in (("ce", lambda: CrossEntropy(512, 57)), ("lmcl", lambda: LMCL(512, 57, 30, 0.2))):
        fus = model_fusion(1024, 512, 57, False)
        crit = make()
        fill(fus, "train.lf."); fill(crit, f"train.{tag}.")
        fus.train(); crit.train()
        params = [{"params": fus.parameters()}, {"params": crit.parameters()}]
        opt = torch.optim.SGD(params, lr=0.5, weight_decay=1e-5, momentum=0.9)
        for step in range(2):      # two steps so the momentum buffer matters
            opt.zero_grad()
            output = fus(torch.cat([xa, ev], dim=1))
            loss, logits = crit(output, lab)
            _, pred = torch.max(logits, dim=1)
            loss.backward()
            if step == 0:
                out[f"{tag}_loss0"] = loss.detach().numpy(); out[f"{tag}_logits0"] = logits.detach().numpy()
                out[f"{tag}_argmax0"] = pred.numpy()
                out[f"{tag}_grad_fc2_w_rows8"] = fus.fc2.weight.grad[:8].numpy().copy()
                out[f"{tag}_grad_fc1_b"] = fus.fc1.bias.grad.numpy().copy()
                out[f"{tag}_grad_bn1_w"] = fus.bn1.weight.grad.numpy().copy()
                cw = crit.fc.weight if tag == "ce" else crit.weights
                out[f"{tag}_grad_crit_w"] = cw.grad.numpy().copy()
            opt.step()
        out[f"{tag}_loss1"] = loss.detach().numpy()
        sd = {**{"fus." + k: v for k, v in fus.state_dict().items()}, **{"crit." + k: v for k, v in crit.state_dict().items()}}
        for k, v in sd.items():
            v = v.detach().double()
            out[f"{tag}_after2_{k}_sum"] = np.array([float(v.sum()), float(v.abs().sum())])
        out[f"{tag}_after2_fc2_w_rows8"] = fus.fc2.weight.detach()[:8].numpy().copy()
        out[f"{tag}_after2_bn1_running_var"] = fus.bn1.running_var.numpy().copy()
    np.savez_compressed(os.path.join(HERE, "train_golden.npz"), **out)
    print("train:", {k: np.shape(v) for k, v in out.items()})


def audio_train():
    """Two optimisation steps of the FULL speech encoder + LMCL exactly as train_audio.py:167-183 composes them
    (model.train(): batch-statistics BatchNorm everywhere; SGD lr 0.01 / momentum 0.9 / wd 1e-5:
    conf/audio_config.yaml:134-136; LMCL s=30, m=0.2).  5-layer TDNN, 8 utterances x 120 frames x 24 MFCCs."""
    out = {}
    opts = {"arch": "tdnn", "tdnn": {"input_dim": 24, "hidden_dim": [512] * 4 + [1500],
                                     "context": [[-2, -1, 0, 1, 2], [-2, 0, 2], [-3, 0, 3], [0], [0]], "tdnn_layers": 5,
                                     "embedding_dim": 512, "pooling": "statistic", "attention_hidden_size": 64, "bn_first": True}}
    net = SpeakerEmbNet(opts)
    crit = LMCL(512, 57, 30, 0.2)
    fill(net, "atrain.audio."); fill(crit, "atrain.lmcl.")
    net.train(); crit.train()
    x = torch.from_numpy(wg.audio_input(8, 24, 120, key="atrain.x"))
    lab = torch.from_numpy(wg.labels(8, 57))
    opt = torch.optim.SGD([{"params": net.parameters()}, {"params": crit.parameters()}], 0.01, momentum=0.9, weight_decay=1e-5)
    for step in range(2):
        opt.zero_grad()
        output = net(x)
        loss, logits = crit(output, lab)
        loss.backward()
        if step == 0:
            out["loss0"] = loss.detach().numpy(); out["logits0"] = logits.detach().numpy()
            out["argmax0"] = torch.max(logits, dim=1)[1].numpy(); out["output0"] = output.detach().numpy()
            g = {k: v.grad for k, v in net.named_parameters()}
            out["grad_tdnn0_w"] = g["tdnn.0.context_layer.weight"].numpy().copy()
            out["grad_tdnn0_b"] = g["tdnn.0.context_layer.bias"].numpy().copy()
            out["grad_tdnn0_bn_w"] = g["tdnn.0.bn.weight"].numpy().copy()
            out["grad_tdnn2_w_rows4"] = g["tdnn.2.context_layer.weight"][:4].numpy().copy()
            out["grad_tdnn4_bn_b"] = g["tdnn.4.bn.bias"].numpy().copy()
            out["grad_fc1_w_rows4"] = g["fc1.weight"][:4].numpy().copy()
            out["grad_bn2_w"] = g["bn2.weight"].numpy().copy()
            for k, v in g.items():
                out[f"gradnorm_{k}"] = np.array([float(v.double().norm()), float(v.double().sum())])
        opt.step()
    out["loss1"] = loss.detach().numpy()
    for k, v in net.state_dict().items():
        v = v.detach().double()
        out[f"after2_{k}_sum"] = np.array([float(v.sum()), float(v.abs().sum())])
    out["after2_tdnn0_w"] = net.tdnn[0].context_layer.weight.detach().numpy().copy()
    out["after2_tdnn1_running_var"] = net.tdnn[1].bn.running_var.numpy().copy()
    out["after2_fc2_w_rows4"] = net.fc2.weight.detach()[:4].numpy().copy()
    np.savez_compressed(os.path.join(HERE, "audio_train_golden.npz"), **out)
    print("audio_train:", {k: np.shape(v) for k, v in list(out.items())[:12]}, "...", len(out), "arrays")


def audio_attn_train():
    """Two optimisation steps of the speech encoder with ``pooling: attentive_statistic`` (tdnn.py:66-75; pooling.py:73-107) + LMCL,
    composed as train_audio.py:167-183 composes them (model.train(), SGD lr 0.0001 / momentum 0.9 / wd 1e-5).  5-layer TDNN,
    8 utterances x 120 frames x 24 MFCCs: the attention parameters W, b, v, k train with everything else."""
    out = {}
    opts = {"arch": "tdnn", "tdnn": {"input_dim": 24, "hidden_dim": [512] * 4 + [1500],
                                     "context": [[-2, -1, 0, 1, 2], [-2, 0, 2], [-3, 0, 3], [0], [0]], "tdnn_layers": 5,
                                     "embedding_dim": 512, "pooling": "attentive_statistic", "attention_hidden_size": 64, "bn_first": True}}
    net = SpeakerEmbNet(opts)
    crit = LMCL(512, 57, 30, 0.2)
    fill(net, "attrain.audio."); fill(crit, "attrain.lmcl.")
    net.train(); crit.train()
    x = torch.from_numpy(wg.audio_input(8, 24, 120, key="attrain.x"))
    lab = torch.from_numpy(wg.labels(8, 57))
    # lr 0.0001, not the config's 0.01: from 0.002 up the reference's SECOND forward is NaN on these inputs -- one step at LMCL's s = 30
    # sharpens the attention until some channel's weighted variance q - m^2 rounds below zero in fp32 and pooling.py:104 takes its
    # square root (the reference has no floor there).  A golden must be finite to pin anything.
    opt = torch.optim.SGD([{"params": net.parameters()}, {"params": crit.parameters()}], 0.0001, momentum=0.9, weight_decay=1e-5)
    for step in range(2):
        opt.zero_grad()
        output = net(x)
        loss, logits = crit(output, lab)
        loss.backward()
        if step == 0:
            out["loss0"] = loss.detach().numpy(); out["logits0"] = logits.detach().numpy()
            out["argmax0"] = torch.max(logits, dim=1)[1].numpy(); out["output0"] = output.detach().numpy()
            g = {k: v.grad for k, v in net.named_parameters()}
            for k in ("pooling.W", "pooling.b", "pooling.v", "pooling.k"):
                out["grad_" + k] = g[k].numpy().copy() if k != "pooling.W" else g[k][:6].numpy().copy()      # (six rows of the 64 x 1500: a small fixture)
            out["grad_fc1_w_rows4"] = g["fc1.weight"][:4].numpy().copy()
            out["grad_tdnn4_bn_w"] = g["tdnn.4.bn.weight"].numpy().copy()
            for k, v in g.items():
                out[f"gradnorm_{k}"] = np.array([float(v.double().norm()), float(v.double().sum())])
        opt.step()
    out["loss1"] = loss.detach().numpy()
    for k, v in net.state_dict().items():
        v = v.detach().double()
        out[f"after2_{k}_sum"] = np.array([float(v.sum()), float(v.abs().sum())])
    for k in ("pooling.W", "pooling.v", "pooling.k"):
        v = dict(net.named_parameters())[k].detach()
        out["after2_" + k] = (v[:6] if k == "pooling.W" else v).numpy().copy()
    manifest["audio_attn_train"] = {k: list(np.shape(v)) for k, v in out.items()}
    np.savez_compressed(os.path.join(HERE, "audio_attn_train_golden.npz"), **out)
    print("audio_attn_train:", {k: np.shape(v) for k, v in list(out.items())[:10]}, "...", len(out), "arrays")


def video_train():
    """One optimisation step of the FULL lip-clip model exactly as train_video.py:129-147 composes it
    (model.train(): batch-statistics BatchNorm in stem / trunk / TCN, learnable PReLU slopes; CrossEntropyLoss;
    Adam lr 3e-4 / wd 1e-4: train_video.py:112-113), then the loss of a second forward.  Dropout is set to 0 in
    the TCN options (the keep-masks come from torch's generator, which a re-implementation cannot share);
    2 clips x 7 frames x 88 x 88, lengths [7, 5] (the masked consensus mean, model.py:16-17)."""
    out = {}
    opts = dict(TCN_OPTS, dropout=0.0)
    net = Lipreading(num_classes=54, relu_type="prelu", tcn_options=opts, extract_feats=False)
    fill(net, "vtrain.video.")
    net.train()
    x = torch.from_numpy(wg.video_input(2, frames=7, key="vtrain.x"))
    lengths = [7, 5]
    lab = torch.from_numpy(wg.labels(2, 54))
    opt = torch.optim.Adam(net.parameters(), lr=3e-4, weight_decay=1e-4)
    crit = torch.nn.CrossEntropyLoss()
    opt.zero_grad()
    logits = net(x, lengths=lengths)
    loss = crit(logits, lab)
    loss.backward()
    out["loss0"] = loss.detach().numpy(); out["logits0"] = logits.detach().numpy()
    out["argmax0"] = torch.max(logits, dim=1)[1].numpy()
    g = {k: v.grad for k, v in net.named_parameters()}
    out["grad_stem_w"] = g["frontend3D.0.weight"].numpy().copy()
    out["grad_stem_bn_w"] = g["frontend3D.1.weight"].numpy().copy()
    out["grad_stem_prelu"] = g["frontend3D.2.weight"].numpy().copy()
    out["grad_l1_0_conv1_w_rows4"] = g["trunk.layer1.0.conv1.weight"][:4].numpy().copy()
    out["grad_l2_0_down_w_rows4"] = g["trunk.layer2.0.downsample.0.weight"][:4].numpy().copy()
    out["grad_l2_0_conv1_w_rows2"] = g["trunk.layer2.0.conv1.weight"][:2].numpy().copy()
    out["grad_l4_1_conv2_w_rows2"] = g["trunk.layer4.1.conv2.weight"][:2].numpy().copy()
    out["grad_l3_1_relu2"] = g["trunk.layer3.1.relu2.weight"].numpy().copy()
    out["grad_tcn0_cbcr0_1_w_rows4"] = g["tcn.mb_ms_tcn.network.0.cbcr0_1.conv.weight"][:4].numpy().copy()
    out["grad_tcn3_down_b"] = g["tcn.mb_ms_tcn.network.3.downsample.bias"].numpy().copy()
    out["grad_tcn_out_w_rows4"] = g["tcn.tcn_output.weight"][:4].numpy().copy()
    for k, v in g.items():
        out[f"gradnorm_{k}"] = np.array([float(v.double().norm()), float(v.double().sum())])
    # the same step in fp64 (same class, parameters and inputs cast to double): the yardstick for how far the
    # reference's own fp32 arithmetic is from the exact gradients through 18 BatchNorm'd layers on 2 clips
    import copy
    net64 = copy.deepcopy(net).double()
    net64.zero_grad()
    loss64 = crit(net64(x.double(), lengths=lengths), lab)
    loss64.backward()
    g64 = {k: v.grad for k, v in net64.named_parameters()}
    out["loss0_f64"] = loss64.detach().numpy()
    for gk, pk, rows in (("grad_stem_w", "frontend3D.0.weight", None), ("grad_stem_bn_w", "frontend3D.1.weight", None),
                         ("grad_stem_prelu", "frontend3D.2.weight", None), ("grad_l1_0_conv1_w_rows4", "trunk.layer1.0.conv1.weight", 4),
                         ("grad_l2_0_down_w_rows4", "trunk.layer2.0.downsample.0.weight", 4),
                         ("grad_l2_0_conv1_w_rows2", "trunk.layer2.0.conv1.weight", 2),
                         ("grad_l4_1_conv2_w_rows2", "trunk.layer4.1.conv2.weight", 2), ("grad_l3_1_relu2", "trunk.layer3.1.relu2.weight", None),
                         ("grad_tcn0_cbcr0_1_w_rows4", "tcn.mb_ms_tcn.network.0.cbcr0_1.conv.weight", 4),
                         ("grad_tcn3_down_b", "tcn.mb_ms_tcn.network.3.downsample.bias", None),
                         ("grad_tcn_out_w_rows4", "tcn.tcn_output.weight", 4)):
        v = g64[pk] if rows is None else g64[pk][:rows]
        out[gk + "_f64"] = v.numpy().copy()
    for k, v in g64.items():
        out[f"gradnorm64_{k}"] = np.array([float(v.norm()), float(v.sum())])
    opt.step()
    out["after1_stem_running_var"] = net.frontend3D[1].running_var.numpy().copy()
    out["after1_l4_1_bn2_running_mean"] = net.trunk.layer4[1].bn2.running_mean.numpy().copy()
    out["after1_tcn0_cbcr0_2_running_var"] = net.tcn.mb_ms_tcn.network[0].cbcr0_2.batchnorm.running_var.numpy().copy()
    with torch.no_grad():
        out["loss1"] = crit(net(x, lengths=lengths), lab).numpy()
    np.savez_compressed(os.path.join(HERE, "video_train_golden.npz"), **out)
    print("video_train:", {k: np.shape(v) for k, v in list(out.items())[:14]}, "...", len(out), "arrays; loss0", float(out["loss0"]),
          "loss1", float(out["loss1"]))


if __name__ == "__main__":
    which = sys.argv[1:] or ["video", "audio", "heads", "train", "audio_train", "audio_attn_train", "video_train"]
    mpath = os.path.join(HERE, "manifest.json")
    if os.path.exists(mpath):
        manifest.update(json.load(open(mpath)))
        manifest["versions"] = VERSIONS
    for w in which:
        globals()[w]()
    json.dump(manifest, open(mpath, "w"), indent=0, sort_keys=True)
    print("wrote", mpath)

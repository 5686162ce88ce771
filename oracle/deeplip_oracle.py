"""CPU oracle for the DeepLip audio-visual embedding hot path.   *** TEST INFRASTRUCTURE ***

This file is the checker, never the product: only ``tests/``, ``__graft_entry__.smoke()`` and
the ``cpu_baseline`` leg of ``bench.py`` may import it.  Nothing under ``deeplip_amd/`` does.

It is a plain functional restatement (torch-CPU / numpy, eval mode, fp32) of the reference
algorithm, written from the reference's source text; each function cites the file:line it
follows (paths relative to the reference repo root).  The arithmetic primitives the reference
itself delegates to third-party code (PyTorch ATen conv / batch-norm / linear / std;
scikit-learn ``roc_curve``; SciPy ``brentq`` / ``interp1d``; versions unpinned upstream, see
SURVEY.md §8c) are called from the versions installed in this image (torch 2.10.0, numpy 2.2,
scipy 1.15, scikit-learn 1.7).

Pinning: the reference holds no tests, golden vectors or fixtures for this path,
so the oracle is pinned against outputs of the reference's own model classes run in the build
container: ``tests/golden/capture_golden.py`` imports them from /root/reference, fills them with
the name-keyed deterministic weights of ``deeplip_amd.weightgen`` and commits the outputs as
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks every function below against
those vectors.  Pieces whose reference module cannot be imported anywhere (``LowFER`` ctor needs
a CUDA device; ``models/fusion_models/utils.py`` needs kaldiio) follow the source text and are
marked "parity unpinned (source-text only)".
"""
from __future__ import annotations

from typing import Dict, List, Mapping, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Mapping[str, Tensor]

BN_EPS = 1e-5  # torch.nn.BatchNorm{1,2,3}d default, used everywhere in the reference


def to_torch_sd(sd: Mapping[str, np.ndarray]) -> Dict[str, Tensor]:
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}


_BN_TRAIN = [False]   # set by bn_training(): the video functions below then follow model.train()


class bn_training:
    """Context manager: inside it every BatchNorm of the video restatement uses batch statistics and updates
    its running buffers in place (momentum 0.1), i.e. the functions restate the reference classes under
    model.train() (train_video.py:129).  Dropout is not restated (p = 0 in the pinned configuration)."""

    def __enter__(self):
        _BN_TRAIN.append(True)

    def __exit__(self, *a):
        _BN_TRAIN.pop()


def _bn(x: Tensor, sd: SD, p: str) -> Tensor:
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"],
                        sd[p + ".bias"], training=_BN_TRAIN[-1], momentum=0.1, eps=BN_EPS)


def _act(x: Tensor, sd: SD, p: str, relu_type: str) -> Tensor:
    # nn.PReLU(num_parameters=C) or nn.ReLU  (resnet.py:41-46, model.py:79)
    return F.prelu(x, sd[p + ".weight"]) if relu_type == "prelu" else F.relu(x)


# ----------------------------------------------------------------------------------------
# Video encoder: models/video_models/model.py:61-105, resnet.py:28-127
# ----------------------------------------------------------------------------------------
def frontend3d(sd: SD, x: Tensor, relu_type: str = "prelu", taps: Optional[dict] = None) -> Tensor:
    """model.py:80-85: Conv3d(1,64,(5,7,7),s(1,2,2),p(2,3,3),no bias) -> BN3d -> PReLU(64)|ReLU
    -> MaxPool3d((1,3,3), s(1,2,2), p(0,1,1)).  x [B,1,T,88,88] -> [B,64,T,22,22]."""
    y = F.conv3d(x, sd["frontend3D.0.weight"], None, stride=(1, 2, 2), padding=(2, 3, 3))
    y = _bn(y, sd, "frontend3D.1")
    y = _act(y, sd, "frontend3D.2", relu_type)
    if taps is not None:
        taps["stem_act"] = y
    y = F.max_pool3d(y, kernel_size=(1, 3, 3), stride=(1, 2, 2), padding=(0, 1, 1))
    return y


def basic_block(sd: SD, p: str, x: Tensor, stride: int, relu_type: str) -> Tensor:
    """resnet.py:55-69: conv3x3(stride)-bn1-relu1-conv3x3-bn2, += residual (1x1 stride conv + BN
    when a downsample exists, resnet.py:13-17), relu2."""
    out = F.conv2d(x, sd[p + ".conv1.weight"], None, stride=stride, padding=1)
    out = _bn(out, sd, p + ".bn1")
    out = _act(out, sd, p + ".relu1", relu_type)
    out = F.conv2d(out, sd[p + ".conv2.weight"], None, stride=1, padding=1)
    out = _bn(out, sd, p + ".bn2")
    if (p + ".downsample.0.weight") in sd:
        res = F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride=stride, padding=0)
        res = _bn(res, sd, p + ".downsample.1")
    else:
        res = x
    out = out + res
    return _act(out, sd, p + ".relu2", relu_type)


def resnet18_trunk(sd: SD, x: Tensor, relu_type: str = "prelu", taps: Optional[dict] = None,
                   prefix: str = "trunk") -> Tensor:
    """resnet.py:119-127 with layers [2,2,2,2], planes 64/128/256/512, strides 1/2/2/2
    (resnet.py:79-82,100-115); AdaptiveAvgPool2d(1); flatten.  [N,64,22,22] -> [N,512]."""
    for li, stride in ((1, 1), (2, 2), (3, 2), (4, 2)):
        for bi in range(2):
            x = basic_block(sd, f"{prefix}.layer{li}.{bi}", x, stride if bi == 0 else 1, relu_type)
        if taps is not None:
            taps[f"layer{li}"] = x
    x = F.adaptive_avg_pool2d(x, 1)
    return x.reshape(x.size(0), -1)


def lipreading_features(sd: SD, x: Tensor, relu_type: str = "prelu", taps: Optional[dict] = None) -> Tensor:
    """Lipreading.forward with extract_feats=True (model.py:96-105): stem -> threeD_to_2D_tensor
    (model.py:9-13) -> trunk -> view(B, T, 512)."""
    B, C, T, H, W = x.shape
    y = frontend3d(sd, x, relu_type, taps)
    if taps is not None:
        taps["stem"] = y
    Tn = y.shape[2]
    y = y.transpose(1, 2).reshape(B * Tn, y.shape[1], y.shape[3], y.shape[4])
    y = resnet18_trunk(sd, y, relu_type, taps)
    return y.view(B, Tn, y.size(1))


def _cbcr(sd: SD, p: str, x: Tensor, k: int, dilation: int, relu_type: str) -> Tensor:
    """ConvBatchChompRelu (tcn.py:28-59, dwpw=False): Conv1d(pad=(k-1)*d, dilation d, bias) ->
    BN1d -> symmetric Chomp1d(pad) (tcn.py:12-25) -> PReLU|ReLU."""
    pad = (k - 1) * dilation
    out = F.conv1d(x, sd[p + ".conv.weight"], sd[p + ".conv.bias"], stride=1, padding=pad, dilation=dilation)
    out = _bn(out, sd, p + ".batchnorm")
    if pad > 0:
        out = out[:, :, pad // 2: -pad // 2].contiguous()
    return _act(out, sd, p + ".non_lin", relu_type)


def ms_tcn_head(sd: SD, feats: Tensor, lengths: Sequence[int], kernel_sizes: Sequence[int] = (3, 5, 7),
                num_layers: int = 4, relu_type: str = "prelu", prefix: str = "tcn") -> Tensor:
    """MultiscaleMultibranchTCN.forward (model.py:31-37) in eval mode (dropout = identity):
    4 x MultibranchTemporalBlock (tcn.py:64-116; dilation 2**i, tcn.py:126-134) ->
    _average_batch masked mean over [0:length] (model.py:16-17) -> Linear (model.py:27).
    feats [B,T,512] -> logits [B,num_classes]."""
    x = feats.transpose(1, 2)
    nk = len(kernel_sizes)
    for i in range(num_layers):
        p = f"{prefix}.mb_ms_tcn.network.{i}"
        d = 2 ** i
        out0 = torch.cat([_cbcr(sd, f"{p}.cbcr0_{j}", x, k, d, relu_type) for j, k in enumerate(kernel_sizes)], 1)
        out1 = torch.cat([_cbcr(sd, f"{p}.cbcr1_{j}", out0, k, d, relu_type) for j, k in enumerate(kernel_sizes)], 1)
        if (p + ".downsample.weight") in sd:  # tcn.py:87: always true for the shipped config
            res = F.conv1d(x, sd[p + ".downsample.weight"], sd[p + ".downsample.bias"])
        else:
            res = x
        x = _act(out1 + res, sd, p + ".relu_final", relu_type)
    pooled = torch.stack([torch.mean(x[b][:, 0:int(l)], 1) for b, l in enumerate(lengths)], 0)
    return F.linear(pooled, sd[prefix + ".tcn_output.weight"], sd[prefix + ".tcn_output.bias"])


def lipreading_logits(sd: SD, x: Tensor, lengths: Sequence[int], relu_type: str = "prelu",
                      kernel_sizes: Sequence[int] = (3, 5, 7), num_layers: int = 4) -> Tensor:
    """Lipreading.forward with extract_feats=False (model.py:105)."""
    return ms_tcn_head(sd, lipreading_features(sd, x, relu_type), lengths, kernel_sizes, num_layers, relu_type)


def video_time_mean(feats: Tensor) -> Tensor:
    """train_fusion.py:274,348: torch.mean(model_video(v)[1,T,512].squeeze(-3), dim=0), batched."""
    return feats.mean(dim=1)


def video_group_mean(clip_means: Tensor, group_ptr: Sequence[int]) -> Tensor:
    """train_fusion.py:272-275,346-349: em = sum over the utterance's clip files / len(group).
    ``group_ptr`` is CSR offsets into clip_means [G,512] -> [U,512]."""
    out = []
    for u in range(len(group_ptr) - 1):
        seg = clip_means[group_ptr[u]:group_ptr[u + 1]]
        em = 0
        for row in seg:
            em = em + row
        out.append(em / len(seg))
    return torch.stack(out)


# ----------------------------------------------------------------------------------------
# Audio encoder: models/audio_models/tdnn.py:7-111, pooling.py:7-26,73-107
# ----------------------------------------------------------------------------------------
def tdnn_dilation(context: Sequence[int]) -> Tuple[int, int]:
    """tdnn.py:18-22: kernel = len(context); dilation = (last-first)//(k-1) or 1."""
    k = len(context)
    d = (context[-1] - context[0]) // (k - 1) if k > 1 else 1
    return k, d


def tdnn_block(sd: SD, p: str, x: Tensor, context: Sequence[int], bn_first: bool = True) -> Tensor:
    """TDNN_Block.forward (tdnn.py:35-43): Conv1d(no padding, bias) then BN->LeakyReLU(0.2)
    or LeakyReLU->BN."""
    _, d = tdnn_dilation(context)
    x = F.conv1d(x, sd[p + ".context_layer.weight"], sd[p + ".context_layer.bias"], dilation=d)
    if bn_first:
        return F.leaky_relu(_bn(x, sd, p + ".bn"), 0.2)
    return _bn(F.leaky_relu(x, 0.2), sd, p + ".bn")


def mean_std_pooling(x: Tensor) -> Tensor:
    """MeanStdPooling.forward (pooling.py:24-26): cat(mean(x,2), std(x,2)); torch.std is the
    unbiased (N-1) estimator."""
    return torch.cat([torch.mean(x, dim=2), torch.std(x, dim=2)], dim=1)


def attentive_stat_pooling(sd: SD, p: str, x: Tensor) -> Tensor:
    """AttentiveStatPooling.forward (pooling.py:87-107)."""
    W, b, v, k = sd[p + ".W"], sd[p + ".b"], sd[p + ".v"], sd[p + ".k"]
    hidden = W.matmul(x).transpose(1, 2) + b
    e = F.relu(hidden).matmul(v) + k
    alpha = F.softmax(e, dim=1)
    mean = torch.matmul(x, alpha).squeeze(-1)
    std = torch.sqrt(torch.matmul(x * x, alpha).squeeze(-1) - mean * mean)
    return torch.cat([mean, std], dim=1)


def speaker_tdnn_stack(sd: SD, x: Tensor, context: Sequence[Sequence[int]], bn_first: bool = True) -> Tensor:
    for i, ctx in enumerate(context):
        x = tdnn_block(sd, f"tdnn.{i}", x, ctx, bn_first)
    return x


def speaker_extract_embedding(sd: SD, x: Tensor, context: Sequence[Sequence[int]], bn_first: bool = True,
                              pooling: str = "statistic", taps: Optional[dict] = None) -> Tuple[Tensor, Tensor]:
    """SpeakerEmbNet.extract_embedding (tdnn.py:89-101): tdnn stack -> pooling -> squeeze_(1)
    -> fc1 (= x_a) -> BN/LReLU -> fc2 (= xv)."""
    h = speaker_tdnn_stack(sd, x, context, bn_first)
    if taps is not None:
        taps["tdnn_out"] = h
    if pooling == "statistic":
        h = mean_std_pooling(h)
    elif pooling == "average":
        h = F.adaptive_avg_pool1d(h, 1)
    elif pooling == "attentive_statistic":
        h = attentive_stat_pooling(sd, "pooling", h)
    else:
        raise NotImplementedError("Other pooling method has not implemented.")
    if taps is not None:
        taps["pooled"] = h
    if h.dim() == 3 and h.size(1) == 1:
        h = h.squeeze(1)
    if h.dim() == 3:
        h = h.squeeze(-1)
    x_a = F.linear(h, sd["fc1.weight"], sd["fc1.bias"])
    if bn_first:
        h = F.leaky_relu(_bn(x_a, sd, "bn1"), 0.2)
    else:
        h = _bn(F.leaky_relu(x_a, 0.2), sd, "bn1")
    xv = F.linear(h, sd["fc2.weight"], sd["fc2.bias"])
    return xv, x_a


def speaker_forward(sd: SD, x: Tensor, context, bn_first: bool = True, pooling: str = "statistic") -> Tensor:
    """SpeakerEmbNet.forward (tdnn.py:103-111): extract_embedding()[0] -> bn2/LReLU."""
    xv, _ = speaker_extract_embedding(sd, x, context, bn_first, pooling)
    if bn_first:
        return F.leaky_relu(_bn(xv, sd, "bn2"), 0.2)
    return _bn(F.leaky_relu(xv, 0.2), sd, "bn2")


# ----------------------------------------------------------------------------------------
# Criteria: models/audio_models/loss.py:6-51
# ----------------------------------------------------------------------------------------
def audio_resnet_embedding(sd: SD, x: Tensor, hidden: Sequence[int] = (64, 128, 256), layers: Sequence[int] = (3, 3, 3)) -> Tensor:
    """The build-owned ResNet speech encoder of ``arch: resnet`` (conf/audio_config.yaml:93-102; no source ships
    upstream -- deeplip_amd/audio_resnet.py states the architecture): conv3x3 - BN - ReLU, stages of BasicBlocks
    (resnet-style, stride 2 from the second stage on, 1x1 conv + BN shortcuts), global average pooling, Linear.
    x [B,1,F,T] -> [B,E].  Honours bn_training()."""
    y = F.relu(_bn(F.conv2d(x, sd["conv1.weight"], None, padding=1), sd, "bn1"))
    for i, n in enumerate(layers):
        for j in range(n):
            y = basic_block(sd, f"layers.{i}.{j}", y, 2 if (i > 0 and j == 0) else 1, "relu")
    y = F.adaptive_avg_pool2d(y, 1).flatten(1)
    return F.linear(y, sd["fc.weight"], sd["fc.bias"])


def aam_softmax(emb: Tensor, labels: Tensor, weights: Tensor, s: float, margin: float, easy_margin: bool = False) -> Tuple[Tensor, Tensor]:
    """ArcFace / AAM-softmax as published (the reference's AAMSoftmax is an empty stub, loss.py:62-67):
    cosine logits, cos(theta_y + m) on the target (fallback cos - m sin(pi - m) where theta + m >= pi), CE at scale s.
    -> (loss, cosine logits)."""
    import math
    logits = F.linear(F.normalize(emb), F.normalize(weights))
    sine = torch.sqrt(torch.clamp(1.0 - logits * logits, min=0.0))
    phi = logits * math.cos(margin) - sine * math.sin(margin)
    if easy_margin:
        phi = torch.where(logits > 0, phi, logits)
    else:
        phi = torch.where(logits > math.cos(math.pi - margin), phi, logits - math.sin(math.pi - margin) * margin)
    onehot = F.one_hot(labels, logits.shape[1]).bool()
    out = torch.where(onehot, phi, logits)
    return F.cross_entropy(s * out, labels), logits


def lmcl(emb: Tensor, labels: Tensor, weights: Tensor, s: float, margin: float) -> Tuple[Tensor, Tensor]:
    """LMCL.forward (loss.py:43-51): cosine logits; margin only at the label column;
    CE(s*(cos - m*onehot) + 1e-8) + 1e-5*||W||_1; returns the UN-margined cosine logits."""
    logits = F.linear(F.normalize(emb), F.normalize(weights))
    m = torch.zeros_like(logits)
    m.scatter_(1, labels.view(-1, 1), margin)
    m_logits = s * (logits - m)
    loss = F.cross_entropy(m_logits + 1e-8, labels)
    loss = loss + 0.00001 * torch.norm(weights, 1)
    return loss, logits


def cross_entropy_head(emb: Tensor, labels: Tensor, W: Tensor, b: Tensor) -> Tuple[Tensor, Tensor]:
    """CrossEntropy.forward (loss.py:13-16): Linear then CE(logits + 1e-8)."""
    logits = F.linear(emb, W, b)
    return F.cross_entropy(logits + 1e-8, labels), logits


def argmax_first(logits: Tensor) -> Tensor:
    """torch.max(logits, dim=1)[1] (train_fusion.py:296, train_audio.py:197, train_video.py:145):
    int64 index of the first maximum."""
    return torch.max(logits, dim=1)[1]


# ----------------------------------------------------------------------------------------
# Fusion heads: models/fusion_models/model_fusion.py:10-27, LBP.py:28-54
# ----------------------------------------------------------------------------------------
def linearfusion(sd: SD, x: Tensor, extract_feats: bool) -> Tensor:
    """Linearfusion.forward (model_fusion.py:19-24), eval-mode BN."""
    x1 = F.linear(x, sd["fc1.weight"], sd["fc1.bias"])
    x1 = F.leaky_relu(_bn(x1, sd, "bn1"), 0.2)
    out = F.linear(x1, sd["fc2.weight"], sd["fc2.bias"])
    return x1 if extract_feats else out


def lowfer(e1: Tensor, e2: Tensor) -> Tensor:
    """LowFER.forward (LBP.py:28-54).  The MFB product (LBP.py:38-42) is computed and then
    overwritten (LBP.py:48-50), so the returned value is cat[e1, sigmoid(e2), sigmoid(e2)*e1].
    parity unpinned (source-text only): the ctor hard-codes device='cuda' (LBP.py:12-15)."""
    s = torch.sigmoid(e2)
    return torch.cat([e1, s, s * e1], dim=1)


# ----------------------------------------------------------------------------------------
# Test-time fusion and trial scoring: train_fusion.py:233-238,353-358;
# models/fusion_models/utils.py:234-283,331-527
# ----------------------------------------------------------------------------------------
def feature_normalize_torch(data: Tensor) -> Tensor:
    """Trainer.feature_normalize (train_fusion.py:233-238): per-row (x - mean) / std with
    torch.std's UNBIASED estimator over dim 1."""
    mu = torch.mean(data, dim=1)
    std = torch.std(data, dim=1)
    return ((data.transpose(0, 1) - mu) / std).transpose(0, 1)


def feature_normalize_np(data: np.ndarray) -> np.ndarray:
    """feature_normalize (models/fusion_models/utils.py:524-527): numpy BIASED std, axis 0 of
    a 1-D vector.  parity unpinned (source-text only): the module needs kaldiio to import."""
    mu = np.mean(data, axis=0)
    std = np.std(data, axis=0)
    return (data - mu) / std


def fuse_av(xv_audio: Tensor, em_video: Tensor) -> Tensor:
    """train_fusion.py:353-358: em = cat([znorm(xv_audio), znorm(em_video)], dim=1) -> [U,1024]."""
    return torch.cat([feature_normalize_torch(xv_audio), feature_normalize_torch(em_video)], dim=1)


def sklearn_cosine_rowwise(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """sklearn.metrics.pairwise.cosine_similarity(a.reshape(1,-1), b.reshape(1,-1)) per trial
    (utils.py:244,262): X/||X|| . Y/||Y|| in the input dtype (float32 stays float32)."""
    an = a / np.sqrt(np.einsum("ij,ij->i", a, a))[:, None]
    bn = b / np.sqrt(np.einsum("ij,ij->i", b, b))[:, None]
    return np.einsum("ij,ij->i", an, bn).astype(a.dtype)


def cosine_trial_scores(emb: np.ndarray, idx_a: np.ndarray, idx_b: np.ndarray) -> np.ndarray:
    """eer_cos_* inner loop (utils.py:254-262) with the .npy tree replaced by an [N,D] table."""
    return sklearn_cosine_rowwise(emb[idx_a], emb[idx_b])


def torch_cosine_rowwise(a: np.ndarray, b: np.ndarray, eps: float = 1e-8) -> np.ndarray:
    """F.cosine_similarity(x1, x2, dim=0, eps=1e-8) per trial (utils.py:372)."""
    return F.cosine_similarity(torch.from_numpy(a), torch.from_numpy(b), dim=1, eps=eps).numpy()


def score_fusion(audio_emb: np.ndarray, video_emb: np.ndarray, idx_a: np.ndarray, idx_b: np.ndarray) -> np.ndarray:
    """eer_cos_*_scorefusion (utils.py:331-381): 0.5*sklearn-cos(audio) + 0.5*torch-cos(video)."""
    return (0.5 * cosine_trial_scores(audio_emb, idx_a, idx_b)
            + 0.5 * torch_cosine_rowwise(video_emb[idx_a], video_emb[idx_b]))


def feature_fusion_scores(audio_emb: np.ndarray, video_emb: np.ndarray, idx_a, idx_b) -> np.ndarray:
    """eer_cos_*_featurefusion (utils.py:465-473): hstack(znorm_np(video), znorm_np(audio)) then
    sklearn cosine."""
    def fuse(i):
        v = np.stack([feature_normalize_np(video_emb[j]) for j in i])
        a = np.stack([feature_normalize_np(audio_emb[j]) for j in i])
        return np.hstack((v, a))
    return sklearn_cosine_rowwise(fuse(idx_a), fuse(idx_b))


def plda_llr_bruteforce(u1: np.ndarray, u2: np.ndarray, psi: np.ndarray) -> float:
    """Same/different log-likelihood ratio of one trial in a PLDA model's latent space, computed the long way
    round: explicit Gaussian densities of the stacked pair (models/fusion_models/utils.py:300-304 calls the
    un-vendored `plda` package's calc_same_diff_log_likelihood_ratio -- parity unpinned; the model is
    u = v + e, v ~ N(0, diag(psi)), e ~ N(0, I), same speaker = shared v).  fp64."""
    from scipy.stats import multivariate_normal as mvn
    u1, u2, psi = (np.asarray(a, dtype=np.float64) for a in (u1, u2, psi))
    tot = 0.0
    for d in range(len(psi)):
        cov_same = np.array([[1.0 + psi[d], psi[d]], [psi[d], 1.0 + psi[d]]])
        tot += mvn.logpdf([u1[d], u2[d]], mean=[0.0, 0.0], cov=cov_same)
        tot -= mvn.logpdf(u1[d], mean=0.0, cov=1.0 + psi[d]) + mvn.logpdf(u2[d], mean=0.0, cov=1.0 + psi[d])
    return float(tot)


def eer(y_true: Sequence[int], y_pred: Sequence[float]) -> Tuple[float, float]:
    """utils.py:263-266: roc_curve(y_true, y_pred, pos_label=1); eer = brentq(1-x-interp1d(fpr,tpr)(x),
    0, 1); threshold = interp1d(fpr, thresholds)(eer).  Third-party calls kept verbatim."""
    from scipy.interpolate import interp1d
    from scipy.optimize import brentq
    from sklearn.metrics import roc_curve
    fpr, tpr, threshold = roc_curve(list(y_true), list(y_pred), pos_label=1)
    e = brentq(lambda x: 1. - x - interp1d(fpr, tpr)(x), 0., 1.)
    thr = interp1d(fpr, threshold)(e)
    return float(e), float(thr)


# ----------------------------------------------------------------------------------------
# Build-owned ingest (no reference counterpart on the live path; RgbToGray exists but is unused:
# models/video_models/preprocess.py:32-46; constants dataloaders.py:11-22)
# ----------------------------------------------------------------------------------------
def ingest_rgb_u8(frames_u8: np.ndarray) -> np.ndarray:
    """[B,T,3,H,W] uint8 RGB -> [B,1,T,H,W] float32: gray = 0.299 R + 0.587 G + 0.114 B (the
    BT.601 weights of cv2.COLOR_RGB2GRAY, preprocess.py:44; kept in float, no uint8 rounding),
    /255 then (x - 0.421)/0.165 (dataloaders.py:12,21-22).  Build-owned: parity unpinned."""
    x = frames_u8.astype(np.float32)
    g = np.float32(0.299) * x[:, :, 0] + np.float32(0.587) * x[:, :, 1] + np.float32(0.114) * x[:, :, 2]
    g = g / np.float32(255.0)
    g = (g - np.float32(0.421)) / np.float32(0.165)
    return g[:, None].astype(np.float32)


# ----------------------------------------------------------------------------------------
# End-to-end helpers used by smoke() and bench.py's cpu_baseline leg
# ----------------------------------------------------------------------------------------
ETDNN_CONTEXT = [[-2, -1, 0, 1, 2], [0], [-2, 0, 2], [0], [-3, 0, 3], [0], [-4, 0, 4], [0], [0], [0]]
TDNN_CONTEXT = [[-2, -1, 0, 1, 2], [-2, 0, 2], [-3, 0, 3], [0], [0]]


def fused_av_embedding(video_sd: SD, audio_sd: SD, video: Tensor, audio: Tensor,
                       context=ETDNN_CONTEXT) -> Tensor:
    """One clip per utterance: train_fusion.py:338-358 collapsed to a batch."""
    with torch.no_grad():
        em_video = video_time_mean(lipreading_features(video_sd, video))
        xv, _ = speaker_extract_embedding(audio_sd, audio, context)
        return fuse_av(xv, em_video)


# ----------------------------------------------------------------------------------------
# Trainable tail (config C5): train_fusion.py:286-299 with the encoders frozen
# ----------------------------------------------------------------------------------------
def lipreading_logits_train(p: Dict[str, Tensor], x: Tensor, lengths: Sequence[int], relu_type: str = "prelu") -> Tensor:
    """Lipreading.forward under model.train() with TCN dropout 0 (train_video.py:129,140-146 over model.py:96-105):
    batch-statistics BatchNorm in the stem, every BasicBlock and every ConvBatchChompRelu -- there over the FULL
    padded-length conv output, before the chomp (tcn.py:52-59) -- running buffers of `p` updated in place."""
    with bn_training():
        return lipreading_logits(p, x, lengths, relu_type)


def linearfusion_train(p: Dict[str, Tensor], x: Tensor, extract_feats: bool = False,
                       momentum: float = 0.1) -> Tensor:
    """Linearfusion.forward in TRAIN mode (model_fusion.py:19-24): BatchNorm1d uses batch statistics
    and updates running_mean / running_var in place (torch default momentum 0.1)."""
    x1 = F.linear(x, p["fc1.weight"], p["fc1.bias"])
    x1 = F.batch_norm(x1, p["bn1.running_mean"], p["bn1.running_var"], p["bn1.weight"], p["bn1.bias"],
                      training=True, momentum=momentum, eps=BN_EPS)
    x1 = F.leaky_relu(x1, 0.2)
    return x1 if extract_feats else F.linear(x1, p["fc2.weight"], p["fc2.bias"])


def _bn_train(x: Tensor, p: Dict[str, Tensor], name: str, momentum: float = 0.1) -> Tensor:
    return F.batch_norm(x, p[name + ".running_mean"], p[name + ".running_var"], p[name + ".weight"], p[name + ".bias"],
                        training=True, momentum=momentum, eps=BN_EPS)


def speaker_forward_train(p: Dict[str, Tensor], x: Tensor, context: Sequence[Sequence[int]], bn_first: bool = True,
                          momentum: float = 0.1) -> Tensor:
    """SpeakerEmbNet.forward under model.train() (tdnn.py:35-43,89-111; train_audio.py:167-183): every
    BatchNorm1d uses batch statistics and updates its running stats in place; statistic pooling."""
    h = x
    for i, ctx in enumerate(context):
        _, d = tdnn_dilation(ctx)
        h = F.conv1d(h, p[f"tdnn.{i}.context_layer.weight"], p[f"tdnn.{i}.context_layer.bias"], dilation=d)
        if bn_first:
            h = F.leaky_relu(_bn_train(h, p, f"tdnn.{i}.bn", momentum), 0.2)
        else:
            h = _bn_train(F.leaky_relu(h, 0.2), p, f"tdnn.{i}.bn", momentum)
    h = mean_std_pooling(h)
    x_a = F.linear(h, p["fc1.weight"], p["fc1.bias"])
    h = F.leaky_relu(_bn_train(x_a, p, "bn1", momentum), 0.2) if bn_first else _bn_train(F.leaky_relu(x_a, 0.2), p, "bn1", momentum)
    xv = F.linear(h, p["fc2.weight"], p["fc2.bias"])
    return F.leaky_relu(_bn_train(xv, p, "bn2", momentum), 0.2) if bn_first else _bn_train(F.leaky_relu(xv, 0.2), p, "bn2", momentum)


def sgd_momentum_step(params: Sequence[Tensor], bufs: List[Optional[Tensor]], lr: float, momentum: float,
                      weight_decay: float) -> None:
    """torch.optim.SGD update (train_fusion.py:120-124; conf/fusion_config.yaml:96-99):
    g += wd*p; buf = g (first step) or momentum*buf + g; p -= lr*buf."""
    with torch.no_grad():
        for i, p in enumerate(params):
            g = p.grad + weight_decay * p
            bufs[i] = g.clone() if bufs[i] is None else bufs[i].mul_(momentum).add_(g)
            p.add_(bufs[i], alpha=-lr)
            p.grad = None


# ----------------------------------------------------------------------------------------
# Preprocessing in front of the path.
# The arithmetic lives in the third-party package python_speech_features (pinned nowhere upstream;
# v0.6 is the only release; NOT installed in this image), called at
# models/audio_models/datasets.py:65-83.  Below is a numpy restatement of its published algorithm
# (base.mfcc / base.fbank / base.logfbank / sigproc.*).  parity unpinned (no reference output can be
# generated here); the product's GEMM-based front-end is checked against THIS restatement.
# ----------------------------------------------------------------------------------------
def _psf_frames(signal: np.ndarray, frame_len: int, frame_step: int, preemph: float = 0.97) -> np.ndarray:
    sig = np.append(signal[0], signal[1:] - preemph * signal[:-1])            # sigproc.preemphasis
    slen = len(sig)
    nf = 1 if slen <= frame_len else 1 + int(np.ceil((1.0 * slen - frame_len) / frame_step))
    padlen = int((nf - 1) * frame_step + frame_len)
    pad = np.concatenate((sig, np.zeros((padlen - slen,))))
    idx = np.tile(np.arange(0, frame_len), (nf, 1)) + np.tile(np.arange(0, nf * frame_step, frame_step), (frame_len, 1)).T
    return pad[idx.astype(np.int32)]                                          # rectangular window (winfunc = ones)


def psf_get_filterbanks(nfilt=26, nfft=512, samplerate=16000, lowfreq=0.0, highfreq=None) -> np.ndarray:
    """base.get_filterbanks restated on its own (NOT the product's deeplip_amd.frontend.mel_filterbank, so that a
    filterbank bug in the product cannot cancel out of the front-end parity tests): nfilt triangular filters whose
    corners are nfilt + 2 points equally spaced on the mel scale (hz2mel = 2595 log10(1 + hz / 700)), mapped to FFT
    bins floor((nfft + 1) * hz / samplerate); filter j rises linearly over bins [b_j, b_j+1) and falls over
    [b_j+1, b_j+2).  Built here with vectorised ramps instead of the package's per-bin loops."""
    highfreq = highfreq or samplerate / 2
    mel_lo, mel_hi = 2595.0 * np.log10(1.0 + lowfreq / 700.0), 2595.0 * np.log10(1.0 + highfreq / 700.0)
    mels = mel_lo + (mel_hi - mel_lo) * np.arange(nfilt + 2) / (nfilt + 1)
    hz = 700.0 * (10.0 ** (mels / 2595.0) - 1.0)
    b = np.floor((nfft + 1) * hz / samplerate)
    k = np.arange(nfft // 2 + 1, dtype=np.float64)[None, :]
    lo, mid, hi = b[:-2, None], b[1:-1, None], b[2:, None]
    with np.errstate(divide="ignore", invalid="ignore"):
        up = np.where((k >= lo) & (k < mid), (k - lo) / (mid - lo), 0.0)
        down = np.where((k >= mid) & (k < hi), (hi - k) / (hi - mid), 0.0)
    return np.nan_to_num(up + down)


def psf_fbank(signal, rate=16000, winlen=0.025, winstep=0.01, nfilt=26, nfft=512, preemph=0.97):
    """base.fbank -> (feat [NF, nfilt], energy [NF])."""
    mel_filterbank = psf_get_filterbanks
    frames = _psf_frames(np.asarray(signal, dtype=np.float64), int(round(winlen * rate)), int(round(winstep * rate)), preemph)
    pspec = 1.0 / nfft * np.square(np.absolute(np.fft.rfft(frames, nfft)))    # sigproc.powspec
    energy = np.sum(pspec, 1)
    energy = np.where(energy == 0, np.finfo(float).eps, energy)
    fb = mel_filterbank(nfilt, nfft, rate)
    feat = np.dot(pspec, fb.T)
    feat = np.where(feat == 0, np.finfo(float).eps, feat)
    return feat, energy


def psf_mfcc(signal, rate=16000, winlen=0.025, winstep=0.01, numcep=24, nfilt=26, nfft=512, preemph=0.97,
             ceplifter=22, append_energy=True):
    """base.mfcc (datasets.py:67 passes winlen, winstep, numcep; the rest are package defaults)."""
    from scipy.fftpack import dct
    feat, energy = psf_fbank(signal, rate, winlen, winstep, nfilt, nfft, preemph)
    feat = np.log(feat)
    feat = dct(feat, type=2, axis=1, norm="ortho")[:, :numcep]
    if ceplifter > 0:
        n = np.arange(numcep)
        feat = (1 + (ceplifter / 2.0) * np.sin(np.pi * n / ceplifter)) * feat
    if append_energy:
        feat[:, 0] = np.log(energy)
    return feat


def psf_delta(feat: np.ndarray, N: int) -> np.ndarray:
    """base.delta: feat [NF, F] -> [NF, F]; d[t] = sum_{n=1..N} n (f[t+n] - f[t-n]) / (2 sum n^2), edge-padded."""
    nf = len(feat)
    denom = 2 * sum(i ** 2 for i in range(1, N + 1))
    padded = np.pad(feat, ((N, N), (0, 0)), mode="edge")
    out = np.empty_like(feat)
    for t in range(nf):
        out[t] = np.dot(np.arange(-N, N + 1), padded[t:t + 2 * N + 1]) / denom
    return out


def add_deltas(feat: np.ndarray, order: int = 2) -> np.ndarray:
    """SpkTrainDataset._delta (datasets.py:55-63): hstack of the features, delta(N=1) and -- order 2 -- delta(N=2),
    both taken of the BASE features.  feat [NF, F] -> [NF, (1 + order) F]."""
    parts = [feat, psf_delta(feat, 1)]
    if order == 2:
        parts.append(psf_delta(feat, 2))
    return np.hstack(parts)


def audio_features(signal, feat_type="mfcc", normalize=True, delta=False, **kw) -> np.ndarray:
    """SpkTrainDataset._extract_feature + _normalize (datasets.py:52-53,65-83) -> [F, NF] float32."""
    if feat_type == "mfcc":
        feat = psf_mfcc(signal, **kw)
    elif feat_type == "fbank":
        feat = psf_fbank(signal, **kw)[0]
    elif feat_type == "logfbank":
        feat = np.log(psf_fbank(signal, **kw)[0])
    else:
        raise NotImplementedError("Other features are not implemented!")
    if normalize:
        feat = (feat - feat.mean(axis=0)) / (feat.std(axis=0) + 2e-12)
    if delta:                                     # datasets.py:81-82, after the normalisation
        feat = add_deltas(feat, order=2)
    return feat.T.astype(np.float32)


def video_preprocess_u8(frames_u8: np.ndarray, crop: int = 88) -> np.ndarray:
    """dataloaders.py:17-24 'val' pipeline: Normalize(0,255) -> CenterCrop(crop) -> Normalize(0.421,0.165)
    on [T,H,W] gray (or [T,3,H,W] RGB through the BT.601 gray of preprocess.py:44, kept in float)."""
    x = frames_u8.astype(np.float32)
    if x.ndim == 4:
        x = np.float32(0.299) * x[:, 0] + np.float32(0.587) * x[:, 1] + np.float32(0.114) * x[:, 2]
    h, w = x.shape[-2:]
    dh, dw = int(round((h - crop)) / 2.), int(round((w - crop)) / 2.)         # preprocess.py:89-90 CenterCrop, verbatim: floor(margin / 2)
    x = x[:, dh:dh + crop, dw:dw + crop] / np.float32(255.0)
    return ((x - np.float32(0.421)) / np.float32(0.165)).astype(np.float32)


def video_preprocess_train_u8(frames_u8: np.ndarray, rng, crop: int = 88, flip_ratio: float = 0.5):
    """dataloaders.py:13-17 'train' pipeline: Normalize(0,255) -> RandomCrop(crop) -> HorizontalFlip(0.5) -> Normalize(0.421,0.165)
    on [T,H,W] gray (or [T,3,H,W] RGB through the BT.601 gray).  ``rng``: a ``random.Random`` standing for the module-level generator
    the reference draws from -- RandomCrop takes ``randint(0, w - tw)`` THEN ``randint(0, h - th)`` (preprocess.py:110-111, both ends
    inclusive), HorizontalFlip flips every frame of the clip iff ``random() < flip_ratio`` (:134-136, cv2.flip(frame, 1) = left-right).
    Returns (clip [T,crop,crop] float32, (delta_h, delta_w, flipped))."""
    x = frames_u8.astype(np.float32)
    if x.ndim == 4:
        x = np.float32(0.299) * x[:, 0] + np.float32(0.587) * x[:, 1] + np.float32(0.114) * x[:, 2]
    x = x / np.float32(255.0)                                            # Normalize(0.0, 255.0)
    h, w = x.shape[-2:]
    dw = rng.randint(0, w - crop)
    dh = rng.randint(0, h - crop)
    x = x[:, dh:dh + crop, dw:dw + crop]                                 # RandomCrop
    flipped = rng.random() < flip_ratio
    if flipped:
        x = x[:, :, ::-1]                                                # HorizontalFlip: cv2.flip(frame, 1)
    return ((x - np.float32(0.421)) / np.float32(0.165)).astype(np.float32), (dh, dw, int(flipped))

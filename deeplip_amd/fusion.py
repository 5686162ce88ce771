"""Fusion heads on the HIP engine: mirror of ``models/fusion_models/model_fusion.py`` (Linearfusion)
and ``models/fusion_models/LBP.py`` (LowFER), plus the test-time fusion the reference actually
uses for scoring: per-modality z-norm + concat (train_fusion.py:233-238,353-358)."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from . import ops, packing
from .holders import BatchNormParams, LinearParams, Marker
from .video import _cached_pack


class Linearfusion(nn.Module):
    """model_fusion.py:10-24: fc1 - bn1 - LeakyReLU(0.2) - fc2; returns x1 if extract_feats."""

    def __init__(self, input_size, hidden_size, num_classes, extract_feats):
        super().__init__()
        self.extract_feats = extract_feats
        self.hidden_size = hidden_size
        self.fc1 = LinearParams(input_size, hidden_size)
        self.bn1 = BatchNormParams(hidden_size)
        self.fc2 = LinearParams(hidden_size, hidden_size)
        self.activation = Marker("LeakyReLU(0.2)")

    def _pack(self, device):
        return {"fc1": packing.pack_linear(self.fc1.weight, self.fc1.bias, self.bn1, device,
                                           packing.const_slope(self.hidden_size, 0.2, device)),
                "fc2": packing.pack_linear(self.fc2.weight, self.fc2.bias, None, device)}

    def forward(self, x):
        if self.training:
            # train mode (config C5): batch-statistics BN through the autograd-wrapped HIP kernels
            from . import autograd as ag
            x1 = ag.bn_act_train(ag.linear(x, self.fc1.weight, self.fc1.bias), self.bn1, 0.2)
            return x1 if self.extract_feats else ag.linear(x1, self.fc2.weight, self.fc2.bias)
        p = _cached_pack(self, x.device, self._pack)
        x1 = ops.linear(x.contiguous(), p["fc1"].w, p["fc1"].b, slope=p["fc1"].slope, w_scale=p["fc1"].wscale)  # fc1+bn1+lrelu fused
        if self.extract_feats:
            return x1
        return ops.linear(x1, p["fc2"].w, p["fc2"].b, w_scale=p["fc2"].wscale)


def model_fusion(input_size, hidden_size, num_classes, extract_feats):
    """model_fusion.py:26-27."""
    return Linearfusion(input_size, hidden_size, num_classes, extract_feats)


class LowFER(nn.Module):
    """LBP.py:8-54.  Parameters U, V, bn0, bn1 are kept for state-dict compatibility (created on the
    CPU: the reference hard-codes device='cuda', LBP.py:12-15).  forward returns what the shipped
    code returns -- cat[e1, sigmoid(e2), sigmoid(e2)*e1] -- because the MFB product (LBP.py:38-42) is
    overwritten before use (LBP.py:48-50)."""

    def __init__(self, d1, d2, o):
        super().__init__()
        k = 30
        self.U = nn.Parameter(torch.tensor(np.random.uniform(-1, 1, (d1, k * o)), dtype=torch.float))
        self.V = nn.Parameter(torch.tensor(np.random.uniform(-1, 1, (d2, k * o)), dtype=torch.float))
        self.input_dropout = Marker("Dropout(0.3)")
        self.hidden_dropout1 = Marker("Dropout(0.4)")
        self.hidden_dropout2 = Marker("Dropout(0.5)")
        self.bn0 = BatchNormParams(d1)
        self.bn1 = BatchNormParams(d1)
        self.k, self.o = k, o

    def forward(self, e1, e2):
        return ops.lowfer_cat(e1.contiguous(), e2.contiguous())


def feature_normalize(data: torch.Tensor) -> torch.Tensor:
    """Trainer.feature_normalize (train_fusion.py:233-238): per-row z-norm, unbiased std."""
    return ops.znorm_cat(data.contiguous(), None)


def fuse_av(xv_audio: torch.Tensor, em_video) -> torch.Tensor:
    """train_fusion.py:353-358: cat([znorm(audio), znorm(video)], 1) in one launch -> [U, Da+Dv].  ``em_video`` may be
    the per-clip means [U,512] or what ``Lipreading.embed(x, finish=False)`` returns while they are still pooled partial
    sums (ops.Pooled): the temporal mean of train_fusion.py:348 is then finished inside the same launch."""
    if isinstance(em_video, ops.Pooled):
        return ops.znorm_cat_pooled(xv_audio.contiguous(), em_video)
    return ops.znorm_cat(xv_audio.contiguous(), em_video.contiguous())


_side_streams: dict = {}


def embed_av(model_audio, model_video, feats_audio: torch.Tensor, clips: torch.Tensor, two_streams: bool = True) -> torch.Tensor:
    """One batch of the test-time pipeline (train_fusion.py:338-358): x-vectors, per-clip lip embeddings, z-norm + concat
    -> [B, 1024].  ``two_streams``: the speech encoder is issued on a second HIP stream (fork / join by events), so the
    two encoders' launches -- each of which fills the chip's LDS on its own -- overlap at their heads and tails;
    recorded into a step plan the fork / join becomes two branches of the graph (measured +4.5-5 % on the B = 64 step)."""
    from ._lib import range_scope
    with range_scope():      # one scope over both encoders: its verdict goes out on `cur` behind the join
        return _embed_av(model_audio, model_video, feats_audio, clips, two_streams)


def _embed_av(model_audio, model_video, feats_audio, clips, two_streams):
    if not two_streams:
        return fuse_av(model_audio.extract_embedding(feats_audio)[0], model_video.embed(clips, finish=False))
    cur = torch.cuda.current_stream(clips.device)
    side = _side_streams.get((clips.device, cur.cuda_stream))
    if side is None:
        side = _side_streams[(clips.device, cur.cuda_stream)] = torch.cuda.Stream(device=clips.device)
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        xv_audio = model_audio.extract_embedding(feats_audio)[0]
    em_video = model_video.embed(clips, finish=False)
    cur.wait_stream(side)
    return fuse_av(xv_audio, em_video)

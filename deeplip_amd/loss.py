"""Speaker-classification criteria on the HIP engine: mirror of the reference's
``models/audio_models/loss.py`` (LMCL = CosFace / AM-softmax, CrossEntropy).

``forward(embeddings, labels) -> (loss, logits)`` exactly as the reference; both values come from
``dlip_logits_argmax_f32`` + ``dlip_margin_ce_loss_f32``.  ``predict`` adds the first-max argmax
(torch.max(logits, 1)[1], train_fusion.py:296) from the same launch.  With grad enabled, forward
goes through deeplip_amd/autograd.py (HIP forward + backward kernels) so the criterion trains
(config C5).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from .holders import LinearParams


class LMCL(nn.Module):
    """loss.py:33-51.  ``margin`` is a plain attribute the trainers mutate (train_fusion.py:137-141)."""

    def __init__(self, embedding_size, num_classes, s, margin):
        super().__init__()
        self.embedding_size, self.num_classes = embedding_size, num_classes
        self.s, self.margin = s, margin
        self.weights = nn.Parameter(torch.Tensor(num_classes, embedding_size))
        nn.init.kaiming_normal_(self.weights)

    def predict(self, embeddings, labels=None):
        w = self.weights.detach().contiguous()
        logits, amax = ops.logits_argmax(embeddings.contiguous(), w, cosine=True)
        loss = None
        if labels is not None:
            loss = ops.margin_ce_loss(logits, labels.contiguous(), float(self.s), float(self.margin))
            loss = loss + ops.l1_sum(w) * 0.00001      # L1 regulariser 1e-5*||W||_1 (loss.py:49-50)
        return loss, logits, amax

    def forward(self, embeddings, labels):
        if torch.is_grad_enabled() and (self.weights.requires_grad or embeddings.requires_grad):
            from . import autograd as ag       # differentiable path (training)
            logits = ag.linear(ag.l2_normalize(embeddings), ag.l2_normalize(self.weights))
            loss = ag.margin_ce_loss(logits, labels, self.s, self.margin)
            loss = loss + ag.l1_norm(self.weights, 0.00001)       # L1 term (loss.py:49-50)
            return loss, logits
        loss, logits, _ = self.predict(embeddings, labels)
        return loss, logits


class AAMSoftmax(nn.Module):
    """Additive angular margin softmax (ArcFace).  Upstream `AAMSoftmax` is an empty TODO stub (loss.py:62-67) that
    the north star nevertheless names, so this is the published recipe (parity unpinned) behind the interface of
    the reference's LMCL: ``forward(embeddings, labels) -> (loss, cosine logits)``.
    loss = CE(s * [cos(theta_y + m) on the target, cos(theta_j) elsewhere]); cos(theta + m) where theta + m < pi,
    else cos(theta) - m sin(pi - m); ``easy_margin`` applies the margin only where cos(theta) > 0."""

    def __init__(self, embedding_size, num_classes, s=30.0, margin=0.2, easy_margin=False):
        super().__init__()
        self.embedding_size, self.num_classes = embedding_size, num_classes
        self.s, self.margin, self.easy_margin = s, margin, easy_margin
        self.weights = nn.Parameter(torch.Tensor(num_classes, embedding_size))
        nn.init.kaiming_normal_(self.weights)

    def predict(self, embeddings, labels=None):
        from . import autograd as ag
        w = self.weights.detach().contiguous()
        logits, amax = ops.logits_argmax(embeddings.detach().contiguous(), w, cosine=True)
        loss = None
        if labels is not None:
            with torch.no_grad():
                loss = ops.margin_ce_loss(ag.aam_margin(logits, labels, self.margin, self.easy_margin), labels.contiguous(),
                                          float(self.s), 0.0)
        return loss, logits, amax

    def forward(self, embeddings, labels):
        if torch.is_grad_enabled() and (self.weights.requires_grad or embeddings.requires_grad):
            from . import autograd as ag
            logits = ag.linear(ag.l2_normalize(embeddings), ag.l2_normalize(self.weights))
            loss = ag.margin_ce_loss(ag.aam_margin(logits, labels, self.margin, self.easy_margin), labels, self.s, 0.0)
            return loss, logits
        loss, logits, _ = self.predict(embeddings, labels)
        return loss, logits


class CrossEntropy(nn.Module):
    """loss.py:6-16: Linear + CE(logits + 1e-8)."""

    def __init__(self, embedding_size, num_classes):
        super().__init__()
        self.embedding_size, self.num_classes = embedding_size, num_classes
        self.fc = LinearParams(embedding_size, num_classes)

    def predict(self, embeddings, labels=None):
        logits, amax = ops.logits_argmax(embeddings.contiguous(), self.fc.weight.detach().contiguous(),
                                         self.fc.bias.detach().contiguous(), cosine=False)
        loss = ops.margin_ce_loss(logits, labels.contiguous(), 1.0, 0.0) if labels is not None else None
        return loss, logits, amax

    def forward(self, embeddings, labels):
        if torch.is_grad_enabled() and (self.fc.weight.requires_grad or embeddings.requires_grad):
            from . import autograd as ag
            logits = ag.linear(embeddings, self.fc.weight, self.fc.bias)
            return ag.margin_ce_loss(logits, labels, 1.0, 0.0), logits
        loss, logits, _ = self.predict(embeddings, labels)
        return loss, logits


class _Stub(nn.Module):
    """ASoftmax / AAMSoftmax / Contrastive are empty TODO stubs upstream (loss.py:53-75)."""

    def forward(self, x):
        pass


class ASoftmax(_Stub):
    pass


class Contrastive(_Stub):
    pass

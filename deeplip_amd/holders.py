"""Parameter-holder modules.

The reference builds its models from stock ``torch.nn`` layers; what a drop-in must preserve of
those layers is their *state-dict schema* (key names, shapes, init) -- not their arithmetic, which
the HIP engine replaces.  These classes own parameters/buffers under exactly the names
``nn.Conv*d`` / ``nn.BatchNorm*d`` / ``nn.PReLU`` / ``nn.Linear`` use, so the authors' checkpoints
load with ``load_state_dict``, and they refuse to compute: ``forward``
raises, so no stock-torch (MIOpen / rocBLAS / CPU) path can silently stand in for the kernels.
"""
from __future__ import annotations

import math
from typing import Sequence

import torch
import torch.nn as nn


# Generation of the packed-weight caches (deeplip_amd/packing.py) and of recorded step plans (plan.py): bumped
# whenever a holder's tensors may have been replaced or rewritten wholesale -- load_state_dict, .to()/.cuda()/
# .float(), train()/eval() switches (an optimizer ran in between) -- so a forward only compares one integer.
PACK_GEN = [0]
LOAD_GEN = [0]      # bumped by load_state_dict alone: calibrated activation exponents (packing.act_exponents) belong to the weights they were measured on


def invalidate_packs() -> None:
    PACK_GEN[0] += 1


class _Holder(nn.Module):
    def _apply(self, fn, *a, **k):
        invalidate_packs()
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):
        invalidate_packs()
        LOAD_GEN[0] += 1
        return super()._load_from_state_dict(*a, **k)

    def train(self, mode: bool = True):
        if mode != self.training:
            invalidate_packs()
        return super().train(mode)

    def forward(self, *a, **k):  # pragma: no cover - guard
        raise RuntimeError(
            f"{type(self).__name__} only holds parameters; arithmetic runs in the owning model's HIP engine "
            "(deeplip_amd has no stock-torch compute path)")


class ConvParams(_Holder):
    """Same parameters / default init as nn.Conv{1,2,3}d(in, out, kernel, bias=...)."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: Sequence[int], bias: bool = True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = tuple(kernel_size)
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, *self.kernel_size))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in = self.in_channels * int(math.prod(self.kernel_size))
            bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
            nn.init.uniform_(self.bias, -bound, bound)


class LinearParams(_Holder):
    """Same parameters / default init as nn.Linear."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_features))
        else:
            self.register_parameter("bias", None)
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            bound = 1 / math.sqrt(in_features)
            nn.init.uniform_(self.bias, -bound, bound)


class BatchNormParams(_Holder):
    """Same parameters / buffers as nn.BatchNorm{1,2,3}d(num_features) (eps 1e-5, affine, tracked)."""

    def __init__(self, num_features: int, eps: float = 1e-5, momentum: float = 0.1):
        super().__init__()
        self.num_features, self.eps, self.momentum = num_features, eps, momentum
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class PReLUParams(_Holder):
    """nn.PReLU(num_parameters): weight init 0.25."""

    def __init__(self, num_parameters: int = 1, init: float = 0.25):
        super().__init__()
        self.num_parameters = num_parameters
        self.weight = nn.Parameter(torch.full((num_parameters,), init))


class Marker(_Holder):
    """Parameter-free stage (ReLU / MaxPool3d / Dropout / Chomp1d / LeakyReLU / AdaptiveAvgPool)."""

    def __init__(self, what: str):
        super().__init__()
        self.what = what

    def extra_repr(self):
        return self.what

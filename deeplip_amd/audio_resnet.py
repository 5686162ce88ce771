"""ResNet speech encoder on the HIP engine: ``models.resnet.SpeakerEmbNet`` for ``arch: resnet``.

The reference selects it by config (conf/audio_config.yaml:93-102: ``input_dim 1, hidden_dim [64,128,256],
residual_block_layers [3,3,3], fc_layers 1, embedding_dim 256, pooling average``; train_audio.py:64-66 imports
``models.resnet`` and feeds it ``[B,1,F,T]`` features, :183-184) but ships NO source for it, while the north star
names the "audio_models ResNet ... mel-feature encoder over [B,1,F,T]".  So the architecture is build-owned
(parity unpinned; pinned only to this repo's own oracle restatement, oracle.audio_resnet_embedding), the thin
speaker ResNet those config keys describe:

    Conv2d(1 -> h0, 3x3, pad 1, no bias) - BN - ReLU
    stage i (h_i channels): residual_block_layers[i] BasicBlocks (conv3x3 - BN - ReLU - conv3x3 - BN, + shortcut
        (1x1 stride-2 conv + BN where the shape changes), ReLU); stages 1.. start with stride 2
    global average pooling over (F', T')                      ("pooling: average")
    Linear(h_last -> embedding_dim)                           ("fc_layers: 1")

Engine mapping: exactly the lip-clip trunk's -- every conv is one implicit-GEMM launch with BN / ReLU / residual
fused (deeplip_amd.video.BasicBlock is reused as is), channels-last activations, split activation format
between layers in ``f16x3`` mode.  The 1-channel input is zero-padded to the kernels' channel granule (4, or 32
in split format) once at the boundary.  Train mode runs the same autograd Functions as the lip-clip trunk.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
import torch.nn as nn

from . import ops, packing
from . import arith
from .holders import BatchNormParams, ConvParams, LinearParams, Marker
from .video import BasicBlock, _basic_block_train, _cached_pack, downsample_basic_block

Tensor = torch.Tensor


class SpeakerEmbNet(nn.Module):
    def __init__(self, opts):
        super().__init__()
        o = opts[opts["arch"]] if "arch" in opts else opts
        if o.get("pooling", "average") != "average":
            raise NotImplementedError("resnet speech encoder: only pooling='average' (conf/audio_config.yaml:102)")
        if o.get("fc_layers", 1) != 1:
            raise NotImplementedError("resnet speech encoder: fc_layers must be 1 (conf/audio_config.yaml:99)")
        self.input_dim = o.get("input_dim", 1)
        if self.input_dim != 1:
            raise ValueError("resnet speech encoder takes [B,1,F,T] features (input_dim 1)")
        hidden, layers = list(o["hidden_dim"]), list(o["residual_block_layers"])
        self.embedding_dim = o["embedding_dim"]
        self.conv1 = ConvParams(1, hidden[0], (3, 3), bias=False)
        self.bn1 = BatchNormParams(hidden[0])
        self.relu = Marker("ReLU")
        blocks, inplanes = [], hidden[0]
        for i, (planes, n) in enumerate(zip(hidden, layers)):
            stride = 1 if i == 0 else 2
            stage = []
            for j in range(n):
                s = stride if j == 0 else 1
                down = downsample_basic_block(inplanes, planes, s) if (s != 1 or inplanes != planes) else None
                stage.append(BasicBlock(inplanes, planes, s, down, relu_type="relu"))
                inplanes = planes
            blocks.append(nn.Sequential(*stage))
        self.layers = nn.Sequential(*blocks)
        self.avgpool = Marker("AdaptiveAvgPool2d(1)")
        self.fc = LinearParams(inplanes, self.embedding_dim)
        for m in self.modules():
            if isinstance(m, ConvParams):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))

    def _blocks(self):
        return [b for stage in self.layers for b in stage]

    def _pack(self, device):
        cp = 32 if packing.PRECISION == "f16x3" else 4
        w = torch.zeros(self.conv1.out_channels, cp, 3, 3, dtype=self.conv1.weight.dtype)
        w[:, :1] = self.conv1.weight.detach().cpu()
        return {"stem": packing.pack_conv2d(w, None, self.bn1, device, packing.const_slope(self.conv1.out_channels, 0.0, device)),
                "blocks": [b.pack(device) for b in self._blocks()],
                "fc": packing.pack_linear(self.fc.weight, self.fc.bias, None, device), "cp": cp}

    def _input_nhwc(self, x: Tensor, cp: int) -> Tensor:
        if x.dim() == 3:
            x = x.unsqueeze(1)
        if x.dim() != 4 or x.shape[1] != 1:
            raise ValueError(f"resnet speech encoder expects [B,1,F,T] features, got {tuple(x.shape)}")
        B, _, Fq, T = x.shape
        # [B,1,F,T] -> channels-last [B,F,T,cp] with the one real channel first (a transpose-and-pad launch: [B*F, 1, T] -> [B*F, T, cp])
        return ops.nct_to_ntc(x.contiguous().float().view(B * Fq, 1, T), pad_to=cp).view(B, Fq, T, cp)

    def _embed_train(self, x: Tensor) -> Tensor:
        from . import autograd as ag, autograd_video as av
        h = self._input_nhwc(x, 4)
        w = torch.cat([self.conv1.weight, torch.zeros_like(self.conv1.weight).expand(-1, 3, -1, -1)], dim=1)   # pad C 1 -> 4
        h = av.prelu(av.batchnorm(av.conv(h, w, None, pad=(1, 1)), self.bn1), self.relu)
        for b in self._blocks():
            h = _basic_block_train(b, h)
        return ag.linear(av.avgpool(h), self.fc.weight, self.fc.bias)

    @arith.guarded_eval
    def extract_embedding(self, x: Tensor, lengths=None) -> Tuple[Tensor, Tensor]:
        """[B,1,F,T] (or [B,F,T]) -> (embedding [B,E], the same tensor): the TDNN encoder's (xv, x_a) interface with
        a single fully connected layer."""
        if lengths is not None:
            raise NotImplementedError("the resnet speech encoder's padded convolutions read across an utterance's end: ragged "
                                      "batches (lengths) exist for the TDNN / E-TDNN encoders only")
        if self.training:
            e = self._embed_train(x)
            return e, e
        p = _cached_pack(self, x.device, self._pack)
        split = p["stem"].wscale is not None
        h = self._input_nhwc(x, p["cp"])
        if split:
            h = ops.split_pack(h)
        h = ops.conv_nhwc(h, p["stem"].w, p["stem"].b, pad=(1, 1), slope=p["stem"].slope, w_scale=p["stem"].wscale,
                          x_split=split, out_split=split)
        blocks = self._blocks()
        for i, (b, bp) in enumerate(zip(blocks, p["blocks"])):
            h = b.run(h, bp, split=split, out_split=split and i + 1 < len(blocks))   # the pooling kernel reads fp32
        e = ops.linear(ops.avgpool(h), p["fc"].w, p["fc"].b, w_scale=p["fc"].wscale)
        return e, e

    def forward(self, x: Tensor) -> Tensor:
        return self.extract_embedding(x)[0]

"""Video lip-clip encoder on the HIP engine.  Host-side mirror of the reference's
``models/video_models/{model,resnet,tcn}.py``: same class names, constructor signatures,
attribute (= state-dict key) names and forward signatures, so ``from models.video_models.model
import Lipreading`` keeps working and the authors' ``video_model.pth`` loads unchanged.

Data layout inside the engine is channels-last: the Conv3d stem writes [(B*T),H,W,C] directly, so
``threeD_to_2D_tensor`` (model.py:9-13) costs nothing, every BasicBlock conv is one implicit-GEMM
launch with BN / PReLU / residual fused, and the trunk output [(B*T),512] *is* [B,T,512].
Eval mode (running-statistics BN folded into the packed weights) is the extraction path of the
fusion pipeline, where the encoders are frozen (train_fusion.py:198-201,245-252); under
``model.train()`` the same classes run batch-statistics BN, dropout and a full backward through
the autograd Functions of deeplip_amd/autograd_video.py (train_video.py:108-169).
"""
from __future__ import annotations

import contextlib

import math
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn as nn

from . import _lib, arith, ops, packing
from .holders import BatchNormParams, ConvParams, LinearParams, Marker, PReLUParams

Tensor = torch.Tensor

# Fusions of the split-fp16 extraction path (each has an unfused twin the tests compare it with):
#   FUSE_SHORTCUT  the 1x1 stride-2 shortcut convolution + BN of a down-sampling block runs as extra reduction slices of
#                  the block's conv2 (dlip_conv2_nhwc_f16x3) instead of as its own launch and activation round trip;
#   FUSE_POOL      embed(): AdaptiveAvgPool + the temporal mean come out of the last convolution's epilogue as pooled
#                  partial sums (dlip_conv_pool_f16x3) instead of two more passes over a [B*T,3,3,512] tensor.
FUSE_SHORTCUT = True
FUSE_POOL = True


def _require_eval(m: nn.Module):
    """Standalone sub-module forwards (ResNet, TDNN_Block, the TCN head on their own) exist in eval mode only;
    train mode runs through the owning model's forward (Lipreading / SpeakerEmbNet), which is differentiable."""
    if m.training:
        raise RuntimeError(
            f"{type(m).__name__}: standalone forward is eval-mode only (running-statistics BatchNorm); call .eval(), "
            "or train through the owning model (Lipreading / SpeakerEmbNet), whose forward is differentiable.")


def _act_holder(relu_type: str, channels: int):
    return PReLUParams(channels) if relu_type == "prelu" else Marker("ReLU")


def _slope(act, channels: int, device) -> Tensor:
    if isinstance(act, PReLUParams):
        w = act.weight.detach().float()
        if w.numel() == 1:
            w = w.expand(channels)
        return w.contiguous().to(device)
    return packing.const_slope(channels, 0.0, device)  # ReLU


# ------------------------------------------------------------------------------------------
# resnet.py
# ------------------------------------------------------------------------------------------
def conv3x3(in_planes, out_planes, stride=1):
    """resnet.py:7-9 (parameter holder; stride/padding live in the engine call)."""
    c = ConvParams(in_planes, out_planes, (3, 3), bias=False)
    c.stride = stride
    return c


def downsample_basic_block(inplanes, outplanes, stride):
    """resnet.py:12-16: 1x1 strided conv + BN; keys downsample.0.weight / downsample.1.*"""
    c = ConvParams(inplanes, outplanes, (1, 1), bias=False)
    c.stride = stride
    return nn.Sequential(c, BatchNormParams(outplanes))


class BasicBlock(nn.Module):
    """resnet.py:28-69."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, relu_type="relu"):
        super().__init__()
        assert relu_type in ["relu", "prelu"]
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.bn1 = BatchNormParams(planes)
        self.relu1 = _act_holder(relu_type, planes)
        self.relu2 = _act_holder(relu_type, planes)
        self.conv2 = conv3x3(planes, planes)
        self.bn2 = BatchNormParams(planes)
        self.downsample = downsample
        self.stride = stride
        self.planes = planes

    def pack(self, device, e_x: int = 0, e_h: int = 0, e_o: int = 0) -> Dict[str, packing.Packed]:
        """``e_x`` / ``e_h`` / ``e_o``: activation exponents (f16x3 pack) of the block's input, of conv1's output and of the block's
        output; a block without a shortcut convolution adds its input to its output, so there e_o == e_x (packing.act_exponents)."""
        if self.downsample is None and e_o != e_x:
            raise ValueError("BasicBlock.pack: an identity shortcut needs e_o == e_x")
        p = {
            "conv1": packing.pack_conv2d(self.conv1.weight, None, self.bn1, device, _slope(self.relu1, self.planes, device), e_x, e_h),
            "conv2": packing.pack_conv2d(self.conv2.weight, None, self.bn2, device, _slope(self.relu2, self.planes, device), e_h, e_o),
        }
        if self.downsample is not None:
            p["down"] = packing.pack_conv2d(self.downsample[0].weight, None, self.downsample[1], device, None, e_x, e_o)
            if packing.PRECISION == "f16x3" and self.conv2.weight.shape[1] % 32 == 0 and self.downsample[0].weight.shape[1] % 32 == 0:
                p["conv2+down"] = packing.pack_conv2d_shortcut(self.conv2.weight, self.bn2, self.downsample[0].weight,
                                                              self.downsample[1], device, _slope(self.relu2, self.planes, device),
                                                              e_h, e_x, e_o)
        return p

    def run(self, x: Tensor, p: Dict[str, packing.Packed], split: bool = False, out_split: bool = False,
            pool_group: Optional[int] = None, pool_lengths: Optional[Tensor] = None, pool_len_mul: int = 1, calib=None):
        """x NHWC.  conv1+bn1+relu1 | (1x1 s2 conv + bn) | conv2+bn2 + residual + relu2.
        ``split``: x is in the split activation format (f16x3 packing only) and so are the block's
        internal tensors; ``out_split`` keeps the result in it for the next block; ``pool_group`` (split only):
        return ops.Pooled column sums over groups of that many output pixels instead of the output."""
        s = (self.stride, self.stride)
        h = ops.conv_nhwc(x, p["conv1"].w, p["conv1"].b, stride=s, pad=(1, 1), slope=p["conv1"].slope,
                          w_scale=p["conv1"].wscale, x_split=split, out_split=split)
        if calib is not None and not split:      # a calibrating exact pass: (owner model, block index) -> note |h|, |out|
            packing.calib_note(calib[0], f"h{calib[1]}", h)
        if split and FUSE_SHORTCUT and "conv2+down" in p and pool_group is None:
            q = p["conv2+down"]   # conv2 + bn2 + (1x1 s2 conv + bn)(x) + relu2 in one reduction (resnet.py:62-68)
            return ops.conv2_nhwc(h, x, q.w, q.b, q.wscale, pad=(1, 1), stride2=s, slope=q.slope, out_split=out_split)
        res = ops.conv_nhwc(x, p["down"].w, p["down"].b, stride=s, w_scale=p["down"].wscale,
                            x_split=split, out_split=split) if "down" in p else x
        if pool_group is not None:   # the block's output leaves as pooled partial sums only
            return ops.conv_pool(h, p["conv2"].w, p["conv2"].b, p["conv2"].wscale, pool_group, pad=(1, 1), residual=res,
                                 slope=p["conv2"].slope, lengths=pool_lengths, len_mul=pool_len_mul)
        out = ops.conv_nhwc(h, p["conv2"].w, p["conv2"].b, pad=(1, 1), residual=res, slope=p["conv2"].slope,
                            w_scale=p["conv2"].wscale, x_split=split, out_split=out_split)
        if calib is not None and not split:
            packing.calib_note(calib[0], f"o{calib[1]}", out)
        return out


def _basic_block_train(b: "BasicBlock", x, fork: bool = False):
    """BasicBlock.forward under model.train() (resnet.py:55-69): batch-statistics BN, learnable PReLU slopes,
    every step a differentiable dlip_* launch (deeplip_amd/autograd_video.py).  x NHWC -- or a PAIR of tensors over the same
    values (the previous block's forked output: the convolutions take the first, the shortcut the second, and their two
    gradients meet inside that block's backward instead of in an addition launch of autograd's).  ``fork``: return such a pair."""
    from . import autograd_video as av
    s = (b.stride, b.stride)
    xa, xb = x if isinstance(x, tuple) else (x, x)
    res, side = xb, None
    if b.downsample is not None:
        # the shortcut (1x1 strided convolution + BatchNorm) is independent of the main path: its own stream, forward and -- by
        # autograd's stream rule -- backward (see _tcn_block_train)
        main = _fork_stream(xb)
        side = _branch_streams(xb.device, 1)[0] if main is not None else None
        if side is not None:
            side.wait_stream(main)
        with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
            res = av.batchnorm(av.conv(xb, b.downsample[0].weight, None, stride=s), b.downsample[1])
    # bn1 + relu1's output is not stored: conv2's operand producer applies them on load to conv1's raw output
    h, pend = av.batchnorm_prelu(av.conv(xa, b.conv1.weight, None, stride=s, pad=(1, 1)), b.bn1, b.relu1, defer=True)
    h = av.conv(h, b.conv2.weight, None, pad=(1, 1), pending=pend)
    if side is not None:
        main.wait_stream(side)
    return av.batchnorm_add_prelu(h, b.bn2, res, b.relu2, fork=fork)      # bn2 + shortcut + relu2: one Function


class ResNet(nn.Module):
    """resnet.py:72-127 (BasicBlock, [2,2,2,2]); init as resnet.py:86-99."""

    def __init__(self, block, layers, num_classes=1000, relu_type="relu", gamma_zero=False,
                 avg_pool_downsample=False):
        super().__init__()
        if avg_pool_downsample:
            raise NotImplementedError("downsample_basic_block_v2 is not used by any shipped config")
        self.inplanes = 64
        self.relu_type = relu_type
        self.gamma_zero = gamma_zero
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.avgpool = Marker("AdaptiveAvgPool2d(1)")
        for m in self.modules():
            if isinstance(m, ConvParams):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
        if self.gamma_zero:
            for m in self.modules():
                if isinstance(m, BasicBlock):
                    m.bn2.weight.data.zero_()

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = downsample_basic_block(self.inplanes, planes * block.expansion, stride)
        layers = [block(self.inplanes, planes, stride, downsample, relu_type=self.relu_type)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, relu_type=self.relu_type))
        return nn.Sequential(*layers)

    def blocks(self) -> List[BasicBlock]:
        return [b for l in (self.layer1, self.layer2, self.layer3, self.layer4) for b in l]

    def pack(self, device, exps: Optional[dict] = None):
        """``exps``: {"stem", "h<i>", "o<i>"} activation exponents of the owning model's f16x3 pack (None / {}: all zero)."""
        e = exps or {}
        out, e_x = [], e.get("stem", 0)
        for i, b in enumerate(self.blocks()):
            e_o = e.get(f"o{i}", 0) if b.downsample is not None else e_x
            out.append(b.pack(device, e_x, e.get(f"h{i}", 0), e_o))
            e_x = e_o
        return out

    def exponent_groups(self):
        """Tensors that meet in a residual addition and therefore share an exponent: a run of identity-shortcut blocks and the
        tensor that enters it ("stem" for layer 1, the down-sampling block's output elsewhere)."""
        groups, cur = [], ["stem"]
        for i, b in enumerate(self.blocks()):
            if b.downsample is not None:
                groups.append(cur)
                cur = []
            cur.append(f"o{i}")
        groups.append(cur)
        return groups

    @staticmethod
    def wants_split(packed) -> bool:
        """True when the packing is split-fp16: the trunk then takes (and keeps) its activations in the
        split activation format, so each one is split once by its producer."""
        return packed[0]["conv1"].wscale is not None

    def run(self, x: Tensor, packed, taps: Optional[dict] = None, x_split: bool = False, pool_frames: Optional[int] = None,
            pool_lengths: Optional[Tensor] = None, owner=None):
        """x [N,H,W,64] NHWC (in the split activation format if ``x_split``) -> [N,512]; with ``pool_frames`` = T
        (f16x3 packing): ops.Pooled sums over each clip's T*Ho*Wo output pixels of the last convolution instead
        (finish with ops.pool_finish(..., 'mean') = AdaptiveAvgPool + temporal mean); ``pool_lengths`` (int32 CUDA [B]):
        ragged batch, clip b's sums cover its first pool_lengths[b] frames."""
        split = self.wants_split(packed)
        if split and not x_split:
            x = ops.split_pack(x)
        elif x_split and not split:
            raise ValueError("ResNet.run: split-format input needs the f16x3 packing")
        blocks = self.blocks()
        for i, (b, p) in enumerate(zip(blocks, packed)):
            last = i == len(blocks) - 1
            if last and pool_frames is not None:
                if not split:
                    raise ValueError("ResNet.run: pooled output needs the f16x3 packing")
                hw = ops.conv_out_size(x.shape[1], 3, b.stride, 1, 1) * ops.conv_out_size(x.shape[2], 3, b.stride, 1, 1)
                return b.run(x, p, split=True, pool_group=pool_frames * hw, pool_lengths=pool_lengths, pool_len_mul=hw)
            x = b.run(x, p, split=split, out_split=split and not last,   # avgpool reads fp32
                      calib=(owner, i) if (owner is not None and packing.CALIB is not None) else None)
            if taps is not None and i % 2 == 1:
                taps[f"layer{i // 2 + 1}"] = ops.split_unpack(x) if (split and not last) else x
        return ops.avgpool(x)

    def forward(self, x: Tensor) -> Tensor:
        """x [N,64,H,W] (reference layout) -> [N,512].  Standalone use; Lipreading calls run()."""
        _require_eval(self)
        pk = _cached_pack(self, x.device, self.pack)
        return self.run(x.permute(0, 2, 3, 1).contiguous(), pk)


# ------------------------------------------------------------------------------------------
# tcn.py (multibranch MS-TCN head; the single-branch TCN is on no shipped config)
# ------------------------------------------------------------------------------------------
class Chomp1d(Marker):
    """tcn.py:12-25.  Symmetric chomp of a (k-1)d-padded conv == 'same' padding (k-1)d/2, which is
    how the engine runs it; the module is kept for state-dict/API symmetry."""

    def __init__(self, chomp_size, symm_chomp):
        super().__init__(f"Chomp1d({chomp_size}, symm={symm_chomp})")
        self.chomp_size, self.symm_chomp = chomp_size, symm_chomp
        if symm_chomp:
            assert chomp_size % 2 == 0, "If symmetric chomp, chomp size needs to be even"


class ConvBatchChompRelu(nn.Module):
    """tcn.py:28-59 (dwpw=False)."""

    def __init__(self, n_inputs, n_outputs, kernel_size, stride, dilation, padding, relu_type, dwpw=False):
        super().__init__()
        if dwpw:
            raise NotImplementedError("depthwise-separable TCN (tcn_dwpw) is off in every shipped config")
        assert stride == 1
        self.kernel_size, self.dilation, self.padding = kernel_size, dilation, padding
        self.n_outputs = n_outputs
        self.conv = ConvParams(n_inputs, n_outputs, (kernel_size,), bias=True)
        self.batchnorm = BatchNormParams(n_outputs)
        self.chomp = Chomp1d(padding, True)
        self.non_lin = _act_holder(relu_type, n_outputs)

    def pack(self, device):
        return packing.pack_conv1d(self.conv.weight, self.conv.bias, self.batchnorm, device,
                                   _slope(self.non_lin, self.n_outputs, device))


class MultibranchTemporalBlock(nn.Module):
    """tcn.py:64-116."""

    def __init__(self, n_inputs, n_outputs, kernel_sizes, stride, dilation, padding, dropout=0.2,
                 relu_type="relu", dwpw=False):
        super().__init__()
        self.kernel_sizes = kernel_sizes
        self.num_kernels = len(kernel_sizes)
        self.n_outputs_branch = n_outputs // self.num_kernels
        self.n_outputs = n_outputs
        self.dilation = dilation
        assert n_outputs % self.num_kernels == 0, "Number of output channels needs to be divisible by number of kernels"
        for k_idx, k in enumerate(kernel_sizes):
            setattr(self, f"cbcr0_{k_idx}", ConvBatchChompRelu(n_inputs, self.n_outputs_branch, k, stride, dilation,
                                                                padding[k_idx], relu_type, dwpw=dwpw))
        self.dropout0 = Marker(f"Dropout({dropout})")
        for k_idx, k in enumerate(kernel_sizes):
            setattr(self, f"cbcr1_{k_idx}", ConvBatchChompRelu(n_outputs, self.n_outputs_branch, k, stride, dilation,
                                                                padding[k_idx], relu_type, dwpw=dwpw))
        self.dropout1 = Marker(f"Dropout({dropout})")
        # tcn.py:87 -- the test compares n_inputs//num_kernels with n_outputs, so it is always true for
        # the shipped config and every block owns a 1x1 projection.
        self.downsample = ConvParams(n_inputs, n_outputs, (1,), bias=True) if (n_inputs // self.num_kernels) != n_outputs else None
        self.relu_final = _act_holder(relu_type, n_outputs)

    def pack(self, device):
        p = {f"cbcr{s}_{j}": getattr(self, f"cbcr{s}_{j}").pack(device) for s in (0, 1) for j in range(self.num_kernels)}
        if self.downsample is not None:
            p["down"] = packing.pack_conv1d(self.downsample.weight, self.downsample.bias, None, device)
        p["final_slope"] = _slope(self.relu_final, self.n_outputs, device)
        return p

    def run(self, x: Tensor, p) -> Tensor:
        """x [B,T,Cin] -> [B,T,n_outputs]; branches write channel slices of one buffer (no concat)."""
        B, T, _ = x.shape
        nb = self.n_outputs_branch
        cur = x
        for s in (0, 1):
            out = ops._empty((B, T, self.n_outputs), x.device)
            for j, k in enumerate(self.kernel_sizes):
                pk = p[f"cbcr{s}_{j}"]
                ops.conv1d_ntc(cur, pk.w, pk.b, dilation=self.dilation, pad=(k - 1) * self.dilation // 2,
                               slope=pk.slope, out=out, out_channel_offset=j * nb, w_scale=pk.wscale)
            cur = out  # dropout: identity in eval
        if self.downsample is not None:
            return ops.conv1d_ntc(x, p["down"].w, p["down"].b, residual=cur, slope=p["final_slope"], w_scale=p["down"].wscale)
        raise NotImplementedError("identity-residual multibranch block never occurs (tcn.py:87)")


# Independent branches of the train-mode graph (the three kernel sizes of a TCN stage, a block's shortcut) on side streams.
# OFF by default (issued from Python the extra stream switches cost more host time than the overlap returns);
# deeplip_amd.train_plan.TrainStepGraph turns it on around the step it runs / records.  Bit-identical to the single-stream step
# (tools/probes/train_streams_identity.py at B = 32; tests/test_train_video_gpu.py).  What a forked branch must not do is CREATE a
# tensor that other streams will read without an edge to its fill -- autograd_video.const_vec waits for the ones it makes.
BRANCH_STREAMS = False
_BRANCH_STREAMS = {}


def _fork_stream(x):
    """The stream to fork side streams from, or None: branches stay on the current stream."""
    if not (BRANCH_STREAMS and x.is_cuda):
        return None
    return torch.cuda.current_stream(x.device)


def _branch_streams(device, n: int):
    key = (device.index if device.index is not None else torch.cuda.current_device())
    have = _BRANCH_STREAMS.setdefault(key, [])
    while len(have) < n:
        have.append(torch.cuda.Stream(device=device))
    return have[:n]


def _tcn_block_train(b: "MultibranchTemporalBlock", x: Tensor, p_drop: float) -> Tensor:
    """MultibranchTemporalBlock.forward under model.train() (tcn.py:89-116).  Each branch convolves with
    padding (k-1)d on both sides, normalises with the batch statistics of the FULL padded-length output,
    THEN chomps symmetrically (tcn.py:52-59: conv -> batchnorm -> chomp -> non_lin) -- in train mode the
    statistics therefore include the edge frames the chomp removes.  x [B,T,C] channels-last."""
    from . import autograd_video as av
    B, T, _ = x.shape
    cur = x
    # The branches of a stage are independent and small ([B T = 928 rows] x 256 channels: 16 tiles on a 512-slot chip, 40 us of
    # mostly latency per launch): each runs on its OWN stream between a fork and a join, so that their launches -- and, since
    # autograd runs an op's backward on the stream of its forward, their backward launches too -- overlap, eagerly and as parallel
    # branches of a recorded step graph.  (A side stream's tensors return to that stream's allocator pool; every side-stream
    # phase starts by waiting for the main stream, so a block is never rewritten while the main stream still reads it.)
    main = _fork_stream(x)
    side = _branch_streams(x.device, len(b.kernel_sizes) - 1) if main is not None else []
    for s in (0, 1):
        outs = []
        for j, k in enumerate(b.kernel_sizes):
            m = getattr(b, f"cbcr{s}_{j}")
            pad = (k - 1) * b.dilation
            st = side[j - 1] if (main is not None and j > 0) else None
            if st is not None:
                st.wait_stream(main)
            with torch.cuda.stream(st) if st is not None else contextlib.nullcontext():
                z = av.conv(cur.reshape(B, 1, T, cur.shape[2]), m.conv.weight, m.conv.bias, pad=(0, pad), dil=(1, b.dilation))
                outs.append(av.batchnorm_prelu(z, m.batchnorm, m.non_lin))   # [B,1,T+pad,nb]; the element-wise PReLU commutes with the chomp
        for st in side:
            main.wait_stream(st)
        cur = av.dropout(av.chomp_concat(outs, T), p_drop)      # symmetric chomp + concatenation: one strided row copy per branch
    if b.downsample is None:
        raise NotImplementedError("identity-residual multibranch block never occurs (tcn.py:87)")
    res = av.conv(x.reshape(B, 1, T, x.shape[2]), b.downsample.weight, b.downsample.bias).view(B, T, b.n_outputs)
    return av.add_prelu(cur, res, b.relu_final)


class MultibranchTemporalConvNet(nn.Module):
    """tcn.py:118-140."""

    def __init__(self, num_inputs, num_channels, tcn_options, dropout=0.2, relu_type="relu", dwpw=False):
        super().__init__()
        self.ksizes = tcn_options["kernel_size"]
        layers = []
        for i in range(len(num_channels)):
            d = 2 ** i
            cin = num_inputs if i == 0 else num_channels[i - 1]
            padding = [(s - 1) * d for s in self.ksizes]
            layers.append(MultibranchTemporalBlock(cin, num_channels[i], self.ksizes, stride=1, dilation=d,
                                                   padding=padding, dropout=dropout, relu_type=relu_type, dwpw=dwpw))
        self.network = nn.Sequential(*layers)


class MultiscaleMultibranchTCN(nn.Module):
    """model.py:20-37."""

    def __init__(self, input_size, num_channels, num_classes, tcn_options, dropout, relu_type, dwpw=False):
        super().__init__()
        self.kernel_sizes = tcn_options["kernel_size"]
        self.num_kernels = len(self.kernel_sizes)
        self.mb_ms_tcn = MultibranchTemporalConvNet(input_size, num_channels, tcn_options, dropout=dropout,
                                                    relu_type=relu_type, dwpw=dwpw)
        self.tcn_output = LinearParams(num_channels[-1], num_classes)

    def pack(self, device):
        return {"blocks": [b.pack(device) for b in self.mb_ms_tcn.network],
                "out": packing.pack_linear(self.tcn_output.weight, self.tcn_output.bias, None, device)}

    def pooled(self, x: Tensor, lengths, p) -> Tensor:
        """x [B,T,512] (already time-major channels-last: the reference's transpose(1,2) is a no-op
        here) -> consensus features [B, 768] = _average_batch(mb_ms_tcn(x)) (model.py:16-17,34-36)."""
        for b, bp in zip(self.mb_ms_tcn.network, p["blocks"]):
            x = b.run(x, bp)
        return ops.time_mean(x, _lengths_i32(lengths, x.device))

    def run(self, x: Tensor, lengths, p) -> Tensor:
        return ops.linear(self.pooled(x, lengths, p), p["out"].w, p["out"].b, w_scale=p["out"].wscale)   # tcn_output, model.py:27,37

    def forward(self, x, lengths, B):
        _require_eval(self)
        return self.run(x.contiguous(), lengths, _cached_pack(self, x.device, self.pack))


class TCN(nn.Module):
    """model.py:40-58 single-branch head: selected only when len(kernel_size) == 1, which no
    shipped config does (conf/video_config.json:6-10, conf/fusion_config.yaml:80)."""

    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError("single-branch TCN head is out of scope (no shipped config selects it)")


# ------------------------------------------------------------------------------------------
# model.py
# ------------------------------------------------------------------------------------------
def _lengths_i32(lengths, device) -> Tensor:
    """Clip lengths as the int32 device vector the masked temporal mean reads.  A CUDA int32 tensor passes
    through untouched (what a recorded step plan needs: no host->device copy inside the step)."""
    if isinstance(lengths, Tensor) and lengths.is_cuda and lengths.dtype == torch.int32:
        return lengths.contiguous()
    return torch.as_tensor([int(l) for l in lengths], dtype=torch.int32).to(device)


def _cached_pack(module: nn.Module, device, builder):
    """The module's packed weights in the CURRENT arithmetic mode.  One slot per mode: the f16x3 pack a recorded plan addresses
    and the f32 pack the auto mode re-runs an out-of-range batch on (deeplip_amd/arith.py) live side by side; a slot of the other
    mode that was baked from older parameter values is dropped when this one is rebuilt."""
    ver = packing.state_version(module, device)
    caches = module.__dict__.get("_dlip_pack")
    if caches is None:
        caches = module.__dict__["_dlip_pack"] = {}
    cache = caches.get(ver[2])
    if cache is None or cache[0] != ver:
        for mode in [m for m, c in caches.items() if (c[0][0], c[0][1], c[0][3]) != (ver[0], ver[1], ver[3])]:
            del caches[mode]
        cache = (ver, builder(device))
        caches[ver[2]] = cache
    if ops.ARENA is not None:
        ops.ARENA.keep.append(cache[1])   # a recorded step addresses these weights: the plan keeps them alive
        ops.ARENA.modules[id(module)] = (module, device, ver)   # ... and checks on every replay that they still are the weights
    return cache[1]


class Lipreading(nn.Module):
    """model.py:61-105.  forward(x [B,1,T,H,W], lengths) -> [B,T,512] (extract_feats) or logits."""

    def __init__(self, hidden_dim=256, backbone_type="resnet", num_classes=500, relu_type="prelu",
                 tcn_options={}, width_mult=1.0, extract_feats=False):
        super().__init__()
        self.extract_feats = extract_feats
        self.backbone_type = backbone_type
        self.relu_type = relu_type
        if backbone_type == "resnet":
            self.frontend_nout = 64
            self.backend_out = 512
            self.trunk = ResNet(BasicBlock, [2, 2, 2, 2], relu_type=relu_type)
        elif backbone_type == "shufflenet":
            assert width_mult in [0.5, 1.0, 1.5, 2.0], "Width multiplier not correct"
            raise NotImplementedError("shufflenet backbone is out of scope: both shipped configs select resnet "
                                      "(conf/video_config.json:2, conf/fusion_config.yaml:75)")
        else:
            raise NotImplementedError(backbone_type)
        frontend_relu = PReLUParams(self.frontend_nout) if relu_type == "prelu" else Marker("ReLU")
        stem = ConvParams(1, self.frontend_nout, (5, 7, 7), bias=False)
        self.frontend3D = nn.Sequential(stem, BatchNormParams(self.frontend_nout), frontend_relu,
                                        Marker("MaxPool3d((1,3,3),(1,2,2),(0,1,1))"))
        self.tcn_dropout = float(tcn_options["dropout"])
        tcn_class = TCN if len(tcn_options["kernel_size"]) == 1 else MultiscaleMultibranchTCN
        self.tcn = tcn_class(input_size=self.backend_out,
                             num_channels=[hidden_dim * len(tcn_options["kernel_size"]) * tcn_options["width_mult"]] * tcn_options["num_layers"],
                             num_classes=num_classes, tcn_options=tcn_options, dropout=tcn_options["dropout"],
                             relu_type=relu_type, dwpw=tcn_options["dwpw"])

    def act_exponent_groups(self):
        return self.trunk.exponent_groups()

    def _pack(self, device):
        # activation exponents of the f16x3 pack (packing.act_exponents: all zero unless a calibration set them): "in" the float
        # clip, "stem" the pooled stem output, "h<i>" / "o<i>" the trunk's block tensors
        e = packing.act_exponents(self) if packing.PRECISION == "f16x3" else {}
        n_blocks = len(self.trunk.blocks())
        pk = {
            "stem": packing.pack_stem3d(self.frontend3D[0].weight, self.frontend3D[1], device,
                                        _slope(self.frontend3D[2], self.frontend_nout, device), e.get("in", 0), e.get("stem", 0)),
            "trunk": self.trunk.pack(device, e),
            "tcn": self.tcn.pack(device),
            "e_in": e.get("in", 0), "e_out": 0,
        }
        if e:
            e_x = e.get("stem", 0)       # the exponent the LAST block's output carries (identity blocks inherit their input's)
            for i, b in enumerate(self.trunk.blocks()):
                e_x = e.get(f"o{i}", 0) if b.downsample is not None else e_x
            pk["e_out"] = e_x
            if e_x:
                pk["out_scale"] = (torch.full((self.backend_out,), 2.0 ** (-e_x), dtype=torch.float32, device=device),
                                   torch.zeros((self.backend_out,), dtype=torch.float32, device=device))
            if pk["e_in"]:
                pk["in_scale"] = (torch.full((1,), 2.0 ** pk["e_in"], dtype=torch.float32, device=device),
                                  torch.zeros((1,), dtype=torch.float32, device=device))
        return pk

    def _forward_train(self, x: Tensor, lengths):
        """forward() under model.train() (train_video.py:129,140-146): the whole encoder differentiable, every
        forward and backward step a dlip_* launch (deeplip_amd/autograd_video.py); BatchNorm uses batch
        statistics and updates its running buffers, Dropout draws a fresh keep-mask."""
        from . import autograd as ag, autograd_video as av
        B, C, T, H, W = x.size()
        if C != 1:
            raise ValueError("Lipreading expects grayscale clips [B,1,T,H,W] (model.py:82)")
        stem, bn, act = self.frontend3D[0], self.frontend3D[1], self.frontend3D[2]
        av.prepare_weights(self)                           # the step's split weight images (forward and data-gradient banks): one launch
        y = av.stem_conv(x.contiguous().float().view(B, T, H, W), stem.weight)       # [(B T),H/2,W/2,64]
        y = av.batchnorm_prelu_maxpool(y, bn, act, fork=True)   # (one Function: no full-resolution tensor between the three; a pair: see BNAddPReLUFn)
        blocks = list(self.trunk.blocks())
        for i, blk in enumerate(blocks):
            y = _basic_block_train(blk, y, fork=i + 1 < len(blocks))     # (a block's output feeds the next block twice)
        y = av.avgpool(y).view(B, T, self.backend_out)
        if self.extract_feats:
            return y
        for blk in self.tcn.mb_ms_tcn.network:
            y = _tcn_block_train(blk, y, self.tcn_dropout)
        return ag.linear(av.time_mean(y, _lengths_i32(lengths, x.device)), self.tcn.tcn_output.weight, self.tcn.tcn_output.bias)

    @arith.guarded_eval
    @_lib.scoped_eval
    def forward(self, x: Tensor, lengths, taps: Optional[dict] = None, pooled: bool = False, ragged: Optional[Tensor] = None,
                clip_params: Optional[Tensor] = None):
        """``pooled`` (eval, f16x3 packing, extract path): return the ops.Pooled sums of the last convolution over each
        clip instead of the [B,T,512] features -- what embed() finishes into the per-clip mean.  ``ragged`` (int32 CUDA [B];
        embed()'s): the clip lengths of a zero-padded batch whose padding frames are to be READ AS ZEROS (and left out of the
        pooled sums) so that every clip comes out as if run alone at its own length -- the reference's classifier path
        (``lengths``, model.py:16-17) instead feeds the padding frames through the net as they are and only masks the
        consensus mean, and so does this method without ``ragged``.  ``clip_params``: per-clip crop / flip of uint8 frames."""
        if self.training:
            return self._forward_train(x, lengths)
        _lib.check_range()      # an overflow reported by an earlier f16x3 launch surfaces here (host read, no sync)
        if x.dtype == torch.uint8:
            return self._forward_u8(x, lengths, taps, pooled, ragged, clip_params)
        B, C, T, H, W = x.size()
        if C != 1:
            raise ValueError("Lipreading expects grayscale clips [B,1,T,H,W] (model.py:82); use "
                             "deeplip_amd.ops.ingest_rgb_u8 for [B,T,3,H,W] uint8 RGB")
        p = _cached_pack(self, x.device, self._pack)
        x = x.contiguous().float()
        split = self.trunk.wants_split(p["trunk"])
        packing.calib_note(self, "in", x)                 # (a calibrating exact pass notes the tensors' magnitudes; no-op otherwise)
        if split and (p["e_in"] or p["e_out"] or packing.act_exponents(self)):
            if taps is not None:
                raise NotImplementedError("taps of a model with calibrated activation exponents (the intermediate tensors are scaled)")
            if p["e_in"]:                                  # a float clip whose gain lies outside the split format: 2^e x first (exact)
                sv = p.setdefault(("in_vec", W), (torch.full((W,), 2.0 ** p["e_in"], dtype=torch.float32, device=x.device),
                                                  torch.zeros((W,), dtype=torch.float32, device=x.device)))
                x = ops.channel_scale(x, sv[0], sv[1])
        if split and taps is None and p["stem"].wscale is not None and W <= 88 and W % 8 == 0:
            # stem + max pooling in one kernel: the pre-pool activations (4x the pooled bytes) stay on chip
            y = ops.stem3d_pool(x.view(B, T, H, W), p["stem"].w, p["stem"].b, p["stem"].slope, p["stem"].wscale, lengths=ragged)
        else:
            if ragged is not None:
                x = ops.mask_frames(x.view(B, T, H * W), ragged).view(B, 1, T, H, W)     # padding frames -> zeros
            y = ops.stem3d(x.view(B, T, H, W), p["stem"].w, p["stem"].b, p["stem"].slope,
                           w_scale=p["stem"].wscale if W <= 88 else None)   # [(B*T),H/2,W/2,64]
            if taps is not None:
                taps["stem_act"] = y
            y = ops.maxpool3x3s2(y, out_split=split)
            if not split:
                packing.calib_note(self, "stem", y)
            if taps is not None:
                taps["stem"] = ops.split_unpack(y) if split else y
        if pooled:
            return self.trunk.run(y, p["trunk"], None, x_split=split, pool_frames=T, pool_lengths=ragged)
        y = self._descale(self.trunk.run(y, p["trunk"], taps, x_split=split, owner=self), p).view(B, T, self.backend_out)
        return y if self.extract_feats else self.tcn.run(y, lengths, p["tcn"])

    @staticmethod
    def _descale(y: Tensor, p) -> Tensor:
        """fp32 features / means of a pack whose last trunk tensor carries an activation exponent: times 2^-e (exact)."""
        sc = p.get("out_scale")
        return ops.channel_scale(y, sc[0], sc[1]) if sc is not None else y

    def _forward_u8(self, frames: Tensor, lengths, taps, pooled: bool, ragged: Optional[Tensor] = None,
                    clip_params: Optional[Tensor] = None):
        """forward() for uint8 frames as a loader hands them over -- [B,T,Hs,Ws] gray (the reference's npz mouth crops) or
        [B,T,3,Hs,Ws] RGB (BASELINE.json's input) -- instead of the normalised float clip [B,1,T,88,88].  On the split-format
        path the centre crop, the gray conversion and (x/255 - 0.421)/0.165 (dataloaders.py:11-22) happen inside the stem's
        pre-pass; otherwise (f32 packing, taps) the ingest kernel writes the float clip and forward() carries on with it.
        Either way HIP kernels do it; there is no host-side path."""
        if frames.dim() not in (4, 5) or (frames.dim() == 5 and frames.shape[2] != 3):
            raise ValueError("Lipreading: uint8 input must be [B,T,H,W] (gray) or [B,T,3,H,W] (RGB)")
        frames = frames.contiguous()
        p = _cached_pack(self, frames.device, self._pack)
        split = self.trunk.wants_split(p["trunk"])
        if split and p["e_in"]:
            raise NotImplementedError("this model's f16x3 pack was calibrated on float clips of an unusual gain (input exponent "
                                      f"{p['e_in']}): uint8 frames always normalise into range -- re-pack (load_state_dict) to drop the calibration")
        B, T = frames.shape[0], frames.shape[1]
        if not (split and taps is None and p["stem"].wscale is not None):
            from .frontend import VideoFrontend
            if clip_params is not None:
                raise NotImplementedError("per-clip crop / flip of uint8 frames lives in the f16x3 stem's pre-pass")
            return self.forward(VideoFrontend(88)(frames), lengths, taps, pooled, ragged)
        y = ops.stem3d_pool_u8(frames, p["stem"].w, p["stem"].b, p["stem"].slope, p["stem"].wscale, crop=88, lengths=ragged,
                               clip_params=clip_params)
        if pooled:
            return self.trunk.run(y, p["trunk"], None, x_split=True, pool_frames=T, pool_lengths=ragged)
        y = self._descale(self.trunk.run(y, p["trunk"], None, x_split=True), p).view(B, T, self.backend_out)
        return y if self.extract_feats else self.tcn.run(y, lengths, p["tcn"])

    @arith.guarded_eval
    def classifier_features(self, x: Tensor, lengths) -> Tensor:
        """[B,1,T,H,W] -> [B,768]: everything of forward() except the final tcn_output Linear (the
        input of the trainable classifier layer in train_video.py)."""
        ef, self.extract_feats = self.extract_feats, True
        try:
            feats = self.forward(x, lengths)
        finally:
            self.extract_feats = ef
        return self.tcn.pooled(feats, lengths, _cached_pack(self, x.device, self._pack)["tcn"])

    @arith.guarded_eval
    @_lib.scoped_eval
    def embed(self, x: Tensor, lengths=None, finish: bool = True):
        """[B,1,T,H,W] -> [B,512]: per-clip temporal mean of the features, the quantity the fusion
        pipeline consumes (train_fusion.py:274,348).  ``finish=False`` may return the means as pooled partial sums
        (ops.Pooled) for deeplip_amd.fusion.fuse_av to finish inside its own launch.

        ``lengths`` (list / int32 tensor [B]; build-owned): a RAGGED batch as pad_packed_collate yields it (dataset.py:123-139:
        clips zero-padded to the longest + their lengths).  Row b then equals ``embed(x[b:b+1, :, :lengths[b]])`` -- the
        reference's test loop, one clip at a time at its own length (train_fusion.py:346-348): the padding frames are read as
        zeros (= the Conv3d's own padding behind the clip's last frame; the trunk is per frame) and stay out of the mean.  A
        device tensor is read by the kernels directly, so a recorded plan replays with new lengths."""
        lens = None
        if lengths is not None:
            if self.training:
                raise NotImplementedError("embed(lengths=...) is the eval-mode extraction path")
            lens = ops.lengths_i32(lengths, x.device, n=x.shape[0], lo=1, hi=x.shape[1] if x.dtype == torch.uint8 else x.shape[2])
        if not self.training and FUSE_POOL and self._can_pool(x):
            pooled = self.forward(x, None, pooled=True, ragged=lens)
            # (finish=False: the sums go to fusion.fuse_av, whose z-norm is invariant under the pack's power-of-two output exponent)
            return self._descale(ops.pool_finish(pooled, "mean"), _cached_pack(self, x.device, self._pack)) if finish else pooled
        ef, self.extract_feats = self.extract_feats, True
        try:
            return ops.time_mean(self.forward(x, lengths=None, ragged=lens), lens)
        finally:
            self.extract_feats = ef

    def _can_pool(self, x: Tensor) -> bool:
        """The pooled epilogue serves the f16x3 packing when a clip's T*Ho*Wo output pixels of the last convolution
        are at least one workgroup tile (so a tile holds at most one clip boundary)."""
        p = _cached_pack(self, x.device, self._pack)
        if not self.trunk.wants_split(p["trunk"]) or p["stem"].wscale is None:
            return False
        if x.dtype == torch.uint8:          # [B,T,Hs,Ws] / [B,T,3,Hs,Ws] frames: cropped to 88 x 88 by the pre-pass
            B, T, H, W = x.shape[0], x.shape[1], 88, 88
        else:
            B, _, T, H, W = x.shape
        if not (W <= 88 and W % 8 == 0):
            return False
        h, w = H // 2, W // 2
        h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1        # stem + max pool
        for _ in range(3):
            h, w = ops.conv_out_size(h, 3, 2, 1, 1), ops.conv_out_size(w, 3, 2, 1, 1)
        probe = torch.empty((B * T, h, w, 512), device="meta")
        return T * h * w >= ops.conv_pool_tile_rows(probe, p["trunk"][-1]["conv2"].w, pad=(1, 1))


def threeD_to_2D_tensor(x):
    """model.py:9-13: kept for API completeness; a pure index permutation (no arithmetic)."""
    n_batch, n_channels, s_time, sx, sy = x.shape
    return x.transpose(1, 2).reshape(n_batch * s_time, n_channels, sx, sy)

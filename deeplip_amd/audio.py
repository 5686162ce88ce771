"""Speech encoder (TDNN / E-TDNN x-vector) on the HIP engine.  Host-side mirror of the
reference's ``models/audio_models/{tdnn,pooling}.py``: same class names, ``opts`` schema,
state-dict keys and return tuples.

Engine layout is time-major channels-last [B,T,C]: each TDNN_Block (dilated Conv1d + BN +
LeakyReLU, tdnn.py:35-43) is one implicit-GEMM launch (k=1 layers are plain GEMMs over B*T rows),
statistics pooling is a two-pass fp64 reduction, fc1/fc2 reuse the GEMM kernel.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib, arith, ops, packing
from .holders import BatchNormParams, ConvParams, LinearParams, Marker
from .video import _cached_pack, _require_eval

Tensor = torch.Tensor
LRELU = 0.2

# f16x3 extraction path: statistics pooling comes out of the last TDNN layer's epilogue as pooled partial sums
# (dlip_conv_pool_f16x3 + dlip_pool_finish_f32) instead of a [B,T',1500] tensor written and read back once
# (pooling.py:24-26).  The unfused twin stays for the tests, for taps and for utterances shorter than a tile.
FUSE_POOL = True


class MeanStdPooling(nn.Module):
    """pooling.py:7-26: [B,C,T] -> [B,2C] = cat(mean, unbiased std) over T."""

    def forward(self, x: Tensor) -> Tensor:
        return ops.meanstd_pool(ops.nct_to_ntc(x.contiguous()))

    def run_ntc(self, x_ntc: Tensor) -> Tensor:
        return ops.meanstd_pool(x_ntc)


class AttentiveStatPooling(nn.Module):
    """pooling.py:73-107: attention over frames, then attention-weighted mean and std.
    hidden = W x + b is one GEMM launch; score / softmax / weighted statistics one fused kernel."""

    def __init__(self, input_size, hidden_size):
        super().__init__()
        self.hidden_size, self.input_size = hidden_size, input_size
        self.W = nn.Parameter(torch.Tensor(hidden_size, input_size))
        self.b = nn.Parameter(torch.Tensor(1, hidden_size))
        self.v = nn.Parameter(torch.Tensor(hidden_size, 1))
        self.k = nn.Parameter(torch.Tensor(1, 1))
        for p in self.parameters():
            nn.init.xavier_normal_(p)

    def run_ntc(self, x_ntc: Tensor, lengths: Optional[Tensor] = None, len_add: int = 0) -> Tensor:
        """x [B,T,C] -> [B,2C]; ``lengths`` (int32 CUDA [B]): attention and statistics over the first lengths[b] + len_add frames."""
        from ._lib import check, lib, ptr, stream_handle
        B, T, C_ = x_ntc.shape
        hidden = ops.linear(x_ntc.reshape(B * T, C_), self.W.detach().contiguous(), self.b.detach().reshape(-1).contiguous())
        y = ops._empty((B, 2 * C_), x_ntc.device)
        check(lib().dlip_attentive_stat_pool_f32(ptr(x_ntc), ptr(hidden), ptr(self.v.detach().contiguous()),
                                                 ptr(self.k.detach().contiguous()), ptr(lengths), len_add, ptr(y), None, B, T, C_,
                                                 self.hidden_size, stream_handle()), "dlip_attentive_stat_pool_f32")
        return y

    def forward(self, x):
        """[B,C,T] (reference layout) -> [B,2C]."""
        return self.run_ntc(ops.nct_to_ntc(x.contiguous()))


class TDNN_Block(nn.Module):
    """tdnn.py:7-43.  ``dilation`` is the context list, e.g. [-2,0,2] -> kernel 3, dilation 2."""

    def __init__(self, input_dim, output_dim, dilation, padding=0, stride=1, bn_first=True):
        super().__init__()
        kernel_size = len(dilation)
        if len(dilation) > 1:
            dilation = (dilation[-1] - dilation[0]) // (len(dilation) - 1)
        else:
            dilation = 1
        if stride != 1:
            raise NotImplementedError("TDNN_Block stride != 1 is never used (tdnn.py:62)")
        self.kernel_size, self.dilation, self.padding = kernel_size, dilation, padding
        self.input_dim, self.output_dim = input_dim, output_dim
        self.context_layer = ConvParams(input_dim, output_dim, (kernel_size,), bias=True)
        self.bn = BatchNormParams(output_dim)
        self.activation = Marker("LeakyReLU(0.2)")
        self.bn_first = bn_first

    def pack(self, device, e_in: int = 0, e_out: int = 0) -> packing.Packed:
        """``e_in`` / ``e_out``: activation exponents of the layer's input / output in the f16x3 pack (packing.act_exponents)."""
        cp = packing.pad_channels(self.input_dim)
        slope = packing.const_slope(self.output_dim, LRELU, device)
        if self.bn_first:   # conv -> BN -> LReLU: BN folds into the conv
            return packing.pack_conv1d(self.context_layer.weight, self.context_layer.bias, self.bn, device, slope, cp, e_in, e_out)
        p = packing.pack_conv1d(self.context_layer.weight, self.context_layer.bias, None, device, slope, cp, e_in, e_out)
        sc, sh = packing.bn_scale_shift(self.bn)  # conv -> LReLU -> BN: BN is the post-affine of the epilogue
        if packing.PRECISION == "f16x3" and e_out:
            sh = sh * (2.0 ** int(e_out))          # (the shift lands behind the activation: scaled with the output)
        p.post_scale, p.post_shift = sc.float().to(device), sh.float().to(device)
        return p

    def run_ntc(self, x: Tensor, p: packing.Packed, x_split: bool = False, out_split: bool = False) -> Tensor:
        """x [B,T,C] -> [B,T',K]; x_split / out_split: split activation format (f16x3 packing only)."""
        return ops.conv1d_ntc(x, p.w, p.b, dilation=self.dilation, pad=self.padding, slope=p.slope,
                              post_scale=p.post_scale, post_shift=p.post_shift, w_scale=p.wscale,
                              x_split=x_split, out_split=out_split)

    def run_pooled(self, x: Tensor, p: packing.Packed, lengths: Optional[Tensor] = None, len_add: int = 0):
        """x [B,T,C] split format -> ops.Pooled column sums of the layer's output over each utterance's T' frames (the
        input of statistics pooling), or None when T' is shorter than the workgroup tile the launch would use.  ``lengths``
        (int32 CUDA [B]): ragged batch -- only the first lengths[b] + len_add output frames of utterance b are pooled."""
        B, T, Cx = x.shape
        Tp = ops.conv_out_size(T, self.kernel_size, 1, self.padding, self.dilation)
        xv = x.view(B, 1, T, Cx)
        wv = p.w.view(p.w.shape[0], 1, p.w.shape[1], p.w.shape[2])
        kw = dict(pad=(0, self.padding), dil=(1, self.dilation))
        if Tp < ops.conv_pool_tile_rows(xv, wv, **kw):
            return None
        return ops.conv_pool(xv, wv, p.b, p.wscale, Tp, slope=p.slope, post_scale=p.post_scale, post_shift=p.post_shift,
                             lengths=lengths, len_add=len_add, **kw)

    def forward(self, x: Tensor) -> Tensor:
        """[B,C,T] -> [B,K,T'] (reference layout, standalone use)."""
        _require_eval(self)
        p = _cached_pack(self, x.device, self.pack)
        h = ops.nct_to_ntc(x.contiguous(), pad_to=packing.pad_channels(self.input_dim))
        return ops.ntc_to_nct(self.run_ntc(h, p))


class SpeakerEmbNet(nn.Module):
    """tdnn.py:45-111.  ``opts`` = model options with sub-dict ``opts[opts['arch']]``."""

    def __init__(self, opts):
        super().__init__()
        opts = opts[opts["arch"]]
        context = opts["context"]
        input_dim = opts["input_dim"]
        hidden_dim = opts["hidden_dim"]
        layers_num = opts["tdnn_layers"]
        embedding_dim = opts["embedding_dim"]
        attention_hidden_size = opts["attention_hidden_size"]
        self.bn_first = opts["bn_first"]
        self.input_dim = input_dim
        self.embedding_dim = embedding_dim
        self.activation = Marker("LeakyReLU(0.2)")
        layers = []
        for i in range(layers_num):
            layers.append(TDNN_Block(input_dim, hidden_dim[i], dilation=context[i], stride=1, bn_first=self.bn_first))
            input_dim = hidden_dim[i]
        self.tdnn = nn.Sequential(*layers)
        self.pooling_type = opts["pooling"]
        if opts["pooling"] == "statistic":
            self.pooling = MeanStdPooling()
        elif opts["pooling"] == "average":
            self.pooling = Marker("AdaptiveAvgPool1d(1)")
        elif opts["pooling"] == "attentive_statistic":
            self.pooling = AttentiveStatPooling(hidden_dim[-1], attention_hidden_size)
        else:
            raise NotImplementedError("Other pooling method has not implemented.")
        if opts["pooling"] in ("statistic", "attentive_statistic"):
            self.fc1 = LinearParams(hidden_dim[-1] * 2, embedding_dim)
        elif opts["pooling"] == "average":
            self.fc1 = LinearParams(hidden_dim[-1], embedding_dim)
        else:
            raise ValueError("pooling method is wrong!")
        self.bn1 = BatchNormParams(embedding_dim)
        self.fc2 = LinearParams(embedding_dim, embedding_dim)
        self.bn2 = BatchNormParams(embedding_dim)

    def _exp(self, name: str) -> int:
        """Activation exponent of a tensor of the f16x3 pack: "in" (the features), "t<i>" (output of TDNN layer i = input of
        layer i + 1; the last one also scales the pooled statistics that fc1 reads).  Zero unless a calibration set it."""
        if packing.PRECISION != "f16x3":
            return 0
        if name == f"t{len(self.tdnn) - 1}" and self.pooling_type == "attentive_statistic":
            return 0            # (the attention scores are not homogeneous in their input: that tensor stays unscaled)
        return packing.act_exponents(self).get(name, 0)

    def _pack(self, device):
        def ss(bn):
            sc, sh = packing.bn_scale_shift(bn)
            return sc.float().to(device), sh.float().to(device)
        n = len(self.tdnn)
        e = [self._exp("in")] + [self._exp(f"t{i}") for i in range(n)]
        pk = {"tdnn": [b.pack(device, e[i], e[i + 1]) for i, b in enumerate(self.tdnn)],
              "fc1": packing.pack_linear(self.fc1.weight, self.fc1.bias, None, device, e_in=e[n], e_out=0),
              "fc2": packing.pack_linear(self.fc2.weight, self.fc2.bias, None, device),
              "bn1": ss(self.bn1), "bn2": ss(self.bn2), "e_in": e[0], "e_last": e[n]}
        if e[0]:
            pk["in_scale"] = torch.full((1,), 2.0 ** e[0], dtype=torch.float32, device=device)
        return pk

    def _to_ntc(self, x: Tensor, split: bool = False, pad32: bool = False) -> Tensor:
        """[B,F,T] -> [B,T,Fp] channels-last; ``split``: Fp = F rounded up to 32, split activation format (``pad32``: that width, fp32)."""
        if x.dim() == 4:            # [B,1,F,T] (the north-star / train_audio.py:183-184 resnet layout)
            x = x.squeeze(1)
        if x.dim() != 3 or x.shape[1] != self.input_dim:
            raise ValueError(f"SpeakerEmbNet expects [B,{self.input_dim},T] features, got {tuple(x.shape)}")
        return ops.nct_to_ntc(x.contiguous().float(), pad_to=packing.pad_channels(self.input_dim, 32 if (split or pad32) else 4),
                              out_split=split)

    def _to_ntc_padded(self, x: Tensor) -> Tensor:
        from . import autograd_video as av
        if x.dim() == 4:
            x = x.squeeze(1)
        if x.dim() != 3 or x.shape[1] != self.input_dim:
            raise ValueError(f"SpeakerEmbNet expects [B,{self.input_dim},T] features, got {tuple(x.shape)}")
        pad32 = av.TRAIN_CONV == "f16x3" and self.input_dim % 32 != 0 and self.tdnn[0].output_dim % 4 == 0
        return ops.nct_to_ntc(x.contiguous().float(), pad_to=packing.pad_channels(self.input_dim, 32 if pad32 else 4))

    def _extract_embedding_train(self, x: Tensor) -> Tuple[Tensor, Tensor]:
        """extract_embedding under model.train() (train_audio.py:167-183): batch-statistics BatchNorm, every
        layer differentiable.  Each step is a torch.autograd Function whose forward and backward are dlip_*
        launches (deeplip_amd/autograd.py: TDNNBlockTrainFn, MeanStdPoolFn, LinearFn, BNRowsActFn)."""
        from . import autograd as ag
        if self.pooling_type not in ("statistic", "attentive_statistic"):
            raise NotImplementedError("train-mode encoder: pooling 'statistic' (the shipped configs) or 'attentive_statistic'")
        if self.input_dim % 4:
            raise ValueError("train-mode encoder: input_dim must be a multiple of 4")
        # [B,T,F] channels-last; F zero-padded to a multiple of 32 when the first layer can then run on the split-fp16 kernels
        # (24 features: the exact-fp32 kernel took 0.55 ms per launch on it, 1.1 of a 15 ms step at B = 256)
        h = self._to_ntc_padded(x)
        from . import autograd_video as av
        av.prepare_weights(self)                           # the step's split weight images (forward and data-gradient banks): one launch
        pend = None                                        # (a block's activated output is not stored when the next block takes it on load)
        # (round 6) the LAST block's activated output is not stored either when the statistics pooling takes it on load (ABI 48)
        # ... and the pooling's BACKWARD writes nothing: the last block's BatchNorm backward forms that gradient per loaded value (hand_over)
        last_on_load = self.pooling_type == "statistic" and ag.POOL_BN_ON_LOAD
        n = len(self.tdnn)
        hand_over = False
        for i, blk in enumerate(self.tdnn):
            if i + 1 == n and last_on_load:
                T_out = h.shape[1] - blk.dilation * (blk.kernel_size - 1) + 2 * blk.padding
                hand_over = ag.POOL_BWD_ON_LOAD and h.shape[0] * T_out > ag.BN_SMALL_ROWS and T_out > 1 and blk.bn_first and ag.BN_ON_LOAD
            h, pend = ag.tdnn_block_train(h, blk, pending=pend, defer=(2 if hand_over else True) if (i + 1 < n or last_on_load) else False)
        # (pooling.py:24-26 | :87-107: attention scores, softmax over frames, weighted mean and std, all differentiable)
        h = (ag.meanstd_pool(h, pending=pend, hand_over=hand_over and pend is not None) if self.pooling_type == "statistic"
             else ag.attentive_stat_pool(h, self.pooling))
        x_a = ag.linear(h, self.fc1.weight, self.fc1.bias)
        h = ag.bn_rows_act_train(x_a, self.bn1, LRELU, act_first=not self.bn_first)
        xv = ag.linear(h, self.fc2.weight, self.fc2.bias)
        return xv, x_a

    def frames_consumed(self) -> int:
        """Frames the stack's valid convolutions take off an utterance: T' = T - frames_consumed() (22 for the E-TDNN)."""
        return sum(b.dilation * (b.kernel_size - 1) - 2 * b.padding for b in self.tdnn)

    @arith.guarded_eval
    @_lib.scoped_eval
    def extract_embedding(self, x: Tensor, lengths=None, taps: Optional[dict] = None) -> Tuple[Tensor, Tensor]:
        """[B,F,T] -> (xv [B,E] = fc2 output, x_a [B,E] = fc1 output)   (tdnn.py:89-101).

        ``lengths`` (list / int32 tensor [B]; build-owned): a RAGGED batch -- x is zero-padded to the longest utterance and
        utterance b has lengths[b] frames.  Row b then equals ``extract_embedding(x[b:b+1, :, :lengths[b]])``, the reference's
        one-utterance-at-a-time test loop (train_fusion.py:334-338, train_audio.py:343-373): the convolutions are valid ones, so
        an output frame t < lengths[b] - frames_consumed() never reads the padding, and the statistics pooling covers exactly
        those frames.  A device tensor is read by the kernels directly (a recorded plan replays with new lengths)."""
        if self.training:
            if lengths is not None:
                raise NotImplementedError("train mode crops every utterance of a batch to one length (datasets.py:112-115)")
            return self._extract_embedding_train(x)
        _lib.check_range()      # an overflow reported by an earlier f16x3 launch surfaces here (host read, no sync)
        p = _cached_pack(self, x.device, self._pack)
        shrink = self.frames_consumed()
        lens = ops.lengths_i32(lengths, x.device, n=x.shape[0], lo=shrink + 2, hi=x.shape[-1])   # >= 2 pooled frames (unbiased std)
        len_add = -shrink
        # f16x3 packing: frame-level activations travel between layers as (hi, lo) fp16 pairs, written
        # by the producing layer's epilogue; the last layer hands fp32 to the pooling kernel, which
        # writes the utterance statistics in that format again for fc1.
        f16x3 = p["tdnn"][0].wscale is not None
        packing.calib_note(self, "in", x)                    # (a calibrating exact pass notes the tensors' magnitudes; no-op otherwise)
        if f16x3 and p["e_in"]:
            # an input whose gain put it outside the split format (calibrated: packing.act_exponents): 2^e x, then the split
            h = ops.split_pack_scaled(self._to_ntc(x, split=False, pad32=True), p["in_scale"])
        else:
            h = self._to_ntc(x, split=f16x3)
        if taps is not None and f16x3 and (p["e_in"] or p["e_last"] or packing.act_exponents(self)):
            raise NotImplementedError("taps of a model with calibrated activation exponents (the intermediate tensors are scaled)")
        split, n = f16x3, len(self.tdnn)
        pooled = None
        for i, (blk, bp) in enumerate(zip(self.tdnn, p["tdnn"])):
            nxt = bp.wscale is not None and i + 1 < n and blk.output_dim % 32 == 0
            if (i + 1 == n and split and FUSE_POOL and taps is None and self.pooling_type == "statistic"
                    and blk.output_dim % 4 == 0):
                pooled = blk.run_pooled(h, bp, lens, len_add)       # None when an utterance is shorter than a workgroup tile
                if pooled is not None:
                    break
            h = blk.run_ntc(h, bp, x_split=split, out_split=nxt)
            packing.calib_note(self, f"t{i}", h)
            split = nxt
        if taps is not None:
            taps["tdnn_out"] = h
        pooled_split = False
        if pooled is not None:
            pooled_split = True
            h = ops.pool_finish(pooled, "meanstd", out_split=True)
        elif self.pooling_type == "statistic":
            pooled_split = f16x3 and h.shape[2] % 4 == 0
            h = ops.meanstd_pool(h, out_split=pooled_split, lengths=lens, len_add=len_add)
        elif self.pooling_type == "average":
            h = ops.time_mean(h, lens, len_add)
        else:
            h = self.pooling.run_ntc(h, lens, len_add)
        if taps is not None:
            taps["pooled"] = ops.split_unpack(h)[:, :self.fc1.in_features].contiguous() if pooled_split else h
        x_a = ops.linear(h, p["fc1"].w, p["fc1"].b, w_scale=p["fc1"].wscale, x_split=pooled_split)
        h = ops.affine_act(x_a, p["bn1"][0], p["bn1"][1], LRELU, act_first=not self.bn_first)
        xv = ops.linear(h, p["fc2"].w, p["fc2"].b, w_scale=p["fc2"].wscale)
        return xv, x_a

    @arith.guarded_eval
    def forward(self, x: Tensor) -> Tensor:
        """tdnn.py:103-111: extract_embedding()[0] -> bn2 / LeakyReLU."""
        xv, _ = self.extract_embedding(x)
        if self.training:
            from . import autograd as ag
            return ag.bn_rows_act_train(xv, self.bn2, LRELU, act_first=not self.bn_first)
        p = _cached_pack(self, x.device, self._pack)
        return ops.affine_act(xv, p["bn2"][0], p["bn2"][1], LRELU, act_first=not self.bn_first)

"""GPU-side preprocessing in front of the encoders.

``AudioFrontend`` turns waveforms [B, S] into the [B, F, T] feature tensors the reference's loaders
produce with python_speech_features (models/audio_models/datasets.py:65-83: ``mfcc`` / ``fbank`` /
``logfbank`` with winlen 0.025, winstep 0.01; conf/fusion_config.yaml:8-40), followed by the
per-utterance mean/variance normalisation of datasets.py:52-53.  The 512-point real DFT, the mel
filterbank and the DCT-II(+lifter) are three fp32 MFMA GEMMs against constant matrices built once
on the host in fp64; framing/pre-emphasis, power spectrum, log and CMVN are small HIP kernels.

``VideoFrontend`` is the test-time pipeline of models/video_models/dataloaders.py:11-22
(Normalize(0,255) -> CenterCrop(88) -> Normalize(0.421, 0.165)) on uint8 gray or RGB frames plus the
zero-pad collate of models/video_models/dataset.py:123-139.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import ops
from ._lib import check, lib, ptr, stream_handle


def hz2mel(hz):
    return 2595.0 * np.log10(1.0 + hz / 700.0)


def mel2hz(mel):
    return 700.0 * (10.0 ** (mel / 2595.0) - 1.0)


def mel_filterbank(nfilt: int, nfft: int, rate: int, lowfreq: float = 0.0, highfreq: Optional[float] = None) -> np.ndarray:
    """python_speech_features.base.get_filterbanks: [nfilt, nfft//2+1] triangular filters on FFT bins
    floor((nfft+1)*hz/rate)."""
    highfreq = highfreq or rate / 2
    melpoints = np.linspace(hz2mel(lowfreq), hz2mel(highfreq), nfilt + 2)
    bins = np.floor((nfft + 1) * mel2hz(melpoints) / rate)
    fb = np.zeros([nfilt, nfft // 2 + 1])
    for j in range(nfilt):
        for i in range(int(bins[j]), int(bins[j + 1])):
            fb[j, i] = (i - bins[j]) / (bins[j + 1] - bins[j])
        for i in range(int(bins[j + 1]), int(bins[j + 2])):
            fb[j, i] = (bins[j + 2] - i) / (bins[j + 2] - bins[j + 1])
    return fb


def num_frames(n_samples: int, frame_len: int, frame_step: int) -> int:
    """sigproc.framesig: 1 + ceil((slen - frame_len) / frame_step) for slen > frame_len, else 1."""
    return 1 if n_samples <= frame_len else 1 + int(math.ceil((1.0 * n_samples - frame_len) / frame_step))


class AudioFrontend:
    def __init__(self, feat_type: str = "mfcc", rate: int = 16000, win_len: float = 0.025, win_shift: float = 0.01,
                 nfft: int = 512, num_bin: int = 26, num_cep: int = 24, preemph: float = 0.97, ceplifter: int = 22,
                 energy: bool = True, normalize: bool = True, delta: bool = False, dft64: Optional[bool] = None, device="cuda",
                 dft: Optional[str] = None):
        if feat_type not in ("mfcc", "fbank", "logfbank"):
            raise NotImplementedError("Other features are not implemented!")   # datasets.py:75-76
        self.feat_type, self.rate, self.nfft = feat_type, rate, nfft
        self.frame_len = int(round(win_len * rate))      # sigproc uses round_half_up; 400 / 160 are exact
        self.frame_step = int(round(win_shift * rate))
        self.num_bin, self.num_cep, self.preemph, self.normalize, self.energy = num_bin, num_cep, preemph, normalize, energy
        self.delta = delta            # datasets.py:81-82: [feat | delta(N=1) | delta(N=2)] after the normalisation
        # How the power spectrum is formed (``dft``):
        #   "fft64"    (default since ABI 46) pre-emphasis, framing and a radix-2 FFT in fp64, one launch from the waveform
        #              (dlip_powspec_wave_fft64_f32) -- the reference's own precision (numpy on doubles), every band of every frame at 1e-4;
        #   "gemm32"   fp32 pre-emphasis + the DFT as an fp32 MFMA GEMM (rounds 1 - 5's default): elements holding ~1e-12 of a frame's
        #              energy come out 0.3 .. 1 % off (tools/probes/frontend_fuzz.py); kept for A/B runs and as a second implementation;
        #   "direct64" fp32 pre-emphasis + a direct fp64 DFT (round 2's route for banks denser than 60 bands).
        # ``dft64`` (rounds 2 - 5's switch): True -> "direct64", False -> "gemm32".
        if dft is None:
            dft = "fft64" if dft64 is None else ("direct64" if dft64 else "gemm32")
        if dft not in ("fft64", "gemm32", "direct64"):
            raise ValueError(f"AudioFrontend: dft must be 'fft64', 'gemm32' or 'direct64', got {dft!r}")
        if dft == "fft64" and (nfft < 128 or nfft > 1024 or nfft & (nfft - 1)):
            dft = "gemm32"            # (the FFT kernel takes powers of two 128 .. 1024; any other nfft keeps the GEMM route)
        self.dft = dft
        self.dft64 = dft == "direct64"
        self.device = torch.device(device)
        nb = nfft // 2 + 1
        self.nb, self.nbp = nb, (nb + 3) // 4 * 4
        k = np.arange(nb)[:, None] * np.arange(nfft)[None, :] * (2.0 * np.pi / nfft)
        dft = np.concatenate([np.cos(k), -np.sin(k)], 0)                       # [2*nb, nfft]: re | im
        fb = np.zeros((num_bin, self.nbp)); fb[:, :nb] = mel_filterbank(num_bin, nfft, rate)
        self.nfp = (num_bin + 3) // 4 * 4
        self.w_dft = torch.from_numpy(dft).float().contiguous().to(self.device)
        self.w_mel = torch.from_numpy(fb).float().contiguous().to(self.device)
        if feat_type == "mfcc":
            n = np.arange(num_bin)
            dct = np.cos(np.pi * np.arange(num_cep)[:, None] * (2 * n[None, :] + 1) / (2.0 * num_bin))   # DCT-II
            dct *= np.sqrt(2.0 / num_bin); dct[0] *= np.sqrt(0.5)                                          # norm='ortho'
            lift = 1.0 + (ceplifter / 2.0) * np.sin(np.pi * np.arange(num_cep) / ceplifter) if ceplifter > 0 else np.ones(num_cep)
            d = np.zeros((num_cep, self.nfp)); d[:, :num_bin] = dct * lift[:, None]                        # lifter folded in
            self.w_dct = torch.from_numpy(d).float().contiguous().to(self.device)

    @property
    def feat_dim(self) -> int:
        return (3 if self.delta else 1) * (self.num_cep if self.feat_type == "mfcc" else self.num_bin)

    def __call__(self, wave: torch.Tensor) -> torch.Tensor:
        """wave [B, S] float32 (cuda) -> features [B, F, NF] float32."""
        wave = wave.contiguous().float()
        B, S = wave.shape
        NF = num_frames(S, self.frame_len, self.frame_step)
        R = B * NF
        pw = ops._empty((R, self.nbp), wave.device)
        energy = ops._empty((R,), wave.device)
        if self.dft == "fft64":
            check(lib().dlip_powspec_wave_fft64_f32(ptr(wave), ptr(pw), ptr(energy), B, S, NF, self.frame_len, self.frame_step, self.nfft,
                                                    float(self.preemph), self.nb, self.nbp, stream_handle()), "dlip_powspec_wave_fft64_f32")
        else:
            frames = ops._empty((R, self.nfft), wave.device)
            check(lib().dlip_frame_preemph_f32(ptr(wave), ptr(frames), B, S, NF, self.frame_len, self.frame_step, self.nfft,
                                               self.preemph, stream_handle()), "dlip_frame_preemph_f32")
            if self.dft == "direct64":
                check(lib().dlip_powspec_dft64_f32(ptr(frames), ptr(pw), ptr(energy), R, self.nb, self.nbp, self.nfft, stream_handle()),
                      "dlip_powspec_dft64_f32")
            else:
                spec = ops.linear(frames, self.w_dft)                          # [R, 2*nb]  (DFT as GEMM)
                check(lib().dlip_powspec_f32(ptr(spec), ptr(pw), ptr(energy), R, self.nb, self.nbp, self.nfft, stream_handle()),
                      "dlip_powspec_f32")
        mel = torch.zeros((R, self.nfp), device=wave.device, dtype=torch.float32)
        ops.conv_nhwc(pw.view(1, 1, R, self.nbp), self.w_mel.view(self.num_bin, 1, 1, self.nbp),
                      out=mel.view(1, 1, R, self.nfp))                         # [R, num_bin] (+ zero pad)
        feat, C_, en = mel, self.num_bin, None
        if self.feat_type in ("mfcc", "logfbank"):
            lg = ops._empty(tuple(mel.shape), mel.device)
            check(lib().dlip_log_floor_f32(ptr(mel), ptr(lg), mel.numel(), stream_handle()), "dlip_log_floor_f32")
            feat = lg
        if self.feat_type == "mfcc":
            feat = ops.linear(feat, self.w_dct)                                # [R, num_cep]
            C_ = self.num_cep
            en = energy if self.energy else None                               # appendEnergy: c0 = log(energy)
        out = ops._empty((B, C_, NF), wave.device)
        check(lib().dlip_cmvn_nct_f32(ptr(feat), ptr(en), ptr(out), B, NF, C_, feat.shape[1], int(self.normalize),
                                      stream_handle()), "dlip_cmvn_nct_f32")
        if self.delta:
            out3 = ops._empty((B, 3 * C_, NF), wave.device)
            check(lib().dlip_delta_nct_f32(ptr(out), ptr(out3), B, C_, NF, 2, stream_handle()), "dlip_delta_nct_f32")
            return out3
        return out


class VideoFrontend:
    """uint8 frames -> normalised grayscale clips [B,1,T,88,88] ready for Lipreading."""

    def __init__(self, crop: int = 88):
        self.crop = crop

    def __call__(self, frames: torch.Tensor, clip_params: Optional[torch.Tensor] = None,
                 lengths: Optional[torch.Tensor] = None) -> torch.Tensor:
        """frames [B,T,H,W] (gray) or [B,T,3,H,W] (RGB) uint8 cuda.  Default: the "val" pipeline (CenterCrop).  ``clip_params``
        (int32 cuda [B,4] = (oy, ox, flip, 0), see ops.draw_clip_params): the "train" pipeline's RandomCrop + HorizontalFlip per
        clip (preprocess.py:95-138).  ``lengths`` (int32 cuda [B]): frames t >= lengths[b] become zeros of the NORMALISED clip,
        the padding of pad_packed_collate."""
        if frames.dtype != torch.uint8 or not frames.is_cuda:
            raise TypeError("VideoFrontend expects uint8 CUDA frames")
        frames = frames.contiguous()
        ch = 3 if frames.dim() == 5 else 1
        B, T = frames.shape[0], frames.shape[1]
        H, W = frames.shape[-2], frames.shape[-1]
        for t, n, shape in ((clip_params, "clip_params", (B, 4)), (lengths, "lengths", (B,))):
            if t is not None and (t.dtype != torch.int32 or not t.is_cuda or tuple(t.shape) != shape or not t.is_contiguous()):
                raise ValueError(f"VideoFrontend: {n} must be a contiguous int32 CUDA tensor of shape {shape}")
        y = ops._empty((B, 1, T, self.crop, self.crop), frames.device)   # (arena-aware: this may run inside a recorded step plan)
        check(lib().dlip_crop_normalize_u8(ptr(frames), ptr(clip_params), ptr(lengths), T, ptr(y), B * T, ch, H, W, self.crop,
                                           stream_handle()), "dlip_crop_normalize_u8")
        return y

    def collate(self, clips: Sequence[torch.Tensor]) -> Tuple[torch.Tensor, List[int]]:
        """pad_packed_collate (dataset.py:123-139): clips [T_i,H,W] (or [T_i,3,H,W]) uint8 cuda, sorted by length (desc),
        normalised, and zero-padded to the longest AFTER the normalisation, as the reference does (its dataset normalises in
        __getitem__, dataset.py:117, and the collate pads the normalised clips with zeros, :130-134).  Round 4 padded the raw
        bytes and normalised the padding along: (0 / 255 - 0.421) / 0.165 = -2.55 in every padding pixel instead of 0."""
        order = sorted(range(len(clips)), key=lambda i: clips[i].shape[0], reverse=True)
        lengths = [int(clips[i].shape[0]) for i in order]
        shape = (len(clips), lengths[0]) + tuple(clips[0].shape[1:])
        buf = torch.zeros(shape, dtype=torch.uint8, device=clips[0].device)
        for j, i in enumerate(order):
            buf[j, :lengths[j]] = clips[i]
        return self(buf, lengths=torch.tensor(lengths, dtype=torch.int32).to(buf.device)), lengths

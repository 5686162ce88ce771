"""Embedding extraction over lists of VARIABLE-LENGTH utterances -- the reference's test loops, batched.

The reference walks its trial lists one utterance at a time: the audio at its own length, every lip clip of the utterance at its
own length, batch 1, a ``.to(device)`` in front of every forward (train_fusion.py:334-349; train_audio.py:343-373).  Here the
rank's shard of the list is sorted by length, cut into length-bucketed batches (deeplip_amd/ragged.py: zero-padded to a short
ladder of lengths, padding <= ``waste``), and streamed through one recorded step plan per padded shape with the host-to-device
copies behind the compute (deeplip_amd/pipeline.py: BucketedExtract).  The lengths ride along as int32 device vectors: the
kernels leave the padding out of the pooled statistics (include/deeplip_hip.h, "RAGGED BATCHES"), so every row equals the
reference's one-at-a-time result (tests/test_ragged_gpu.py).  Speech and lip pipelines are fed alternately: each has its own
streams, so the two encoders overlap on the GPU as they do inside the rectangular step.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np
import torch

from . import _lib, ops
from .pipeline import BucketedExtract, pin
from .ragged import Batch, padding_overhead, plan_batches

Tensor = torch.Tensor


class RaggedExtractor:
    """Keeps the recorded plans of both encoders between calls (a second list, a second pass: nothing is recorded again).

    ``audio_fn(x [B,F,T], lengths int32 [B]) -> rows [B,D]`` and ``video_fn(clips [B,1,T,88,88] | uint8 frames, lengths) ->
    rows [B,D]`` are launch-only step functions (deeplip_amd/plan.py)."""

    def __init__(self, audio_fn: Optional[Callable], video_fn: Optional[Callable], device, batch: int = 32, clip_batch: Optional[int] = None,
                 waste: float = 0.10, audio_quantum: int = 4, video_quantum: int = 1, max_arena_bytes: int = 64 << 30,
                 audio_min_frames: int = 1, video_min_frames: int = 1, fallback="auto"):
        """``audio_min_frames`` / ``video_min_frames``: the shortest item the encoders can embed -- for the E-TDNN
        ``frames_consumed() + 2`` = 24 (its valid convolutions take 22 frames off an utterance and the unbiased standard deviation of
        the statistics pooling needs two pooled frames, pooling.py:24-26), one frame for a lip clip.  Checked HERE, on the host,
        where the lengths are: the kernels only clamp a length to [0, T] and a pooled count of 0 (or 1 under the std) would come
        back as NaN / Inf rows in the embedding table with no error.  ``fallback``: see ExtractPipeline (per-batch f32 re-run)."""
        self.device, self.batch, self.clip_batch, self.waste = device, int(batch), int(clip_batch or batch), float(waste)
        self.aq, self.vq = audio_quantum, video_quantum
        self.audio_min_frames, self.video_min_frames = int(audio_min_frames), int(video_min_frames)
        self.pa = BucketedExtract(audio_fn, device=device, max_arena_bytes=max_arena_bytes // 4, fallback=fallback) if audio_fn is not None else None
        self.pv = BucketedExtract(video_fn, device=device, max_arena_bytes=max_arena_bytes, fallback=fallback) if video_fn is not None else None
        self.stats: dict = {}

    def close(self) -> None:
        for p in (self.pa, self.pv):
            if p is not None:
                p.close()

    # ------------------------------------------------------------------ one modality: host batches in submission order
    def _audio_host(self, dataset, lo: int, b: Batch):
        x, L = dataset.audio_padded([lo + int(i) for i in b.idx], T=b.T)
        return (pin(torch.from_numpy(x)), pin(torch.from_numpy(L)))

    def _video_host(self, dataset, c0: int, b: Batch, u8: bool):
        x, L = dataset.clips_padded([c0 + int(i) for i in b.idx], T=b.T)
        if u8:
            from .synthetic import frames_u8_from_clips
            x = frames_u8_from_clips(x, rgb=True)        # [n,T,3,88,88] uint8: the loader's frames (BASELINE.json's input shape)
        return (pin(torch.from_numpy(x)), pin(torch.from_numpy(L)))

    def run(self, dataset, lo: int, hi: int, D: int, u8: bool = False, host_cache: Optional[dict] = None) -> Tuple[Optional[Tensor], Optional[Tensor]]:
        """Utterances lo .. hi of ``dataset`` (deeplip_amd.synthetic.SyntheticAVSet interface: audio_len, clip_len, clip_ptr,
        audio_padded, clips_padded) -> (x-vectors [n,D], per-utterance lip embeddings [n,D] = mean over the utterance's clips of the
        clips' frame means, train_fusion.py:346-349), in list order.  ``host_cache``: keeps the pinned host batches (the bench
        walks one list several times and times the GPU, not numpy)."""
        n = hi - lo
        dev = self.device
        xa = xv = None
        if n and self.pa is not None:
            short = np.flatnonzero(np.asarray(dataset.audio_len[lo:hi]) < self.audio_min_frames)
            if short.size:
                raise ValueError(f"RaggedExtractor: utterance {lo + int(short[0])} has {int(dataset.audio_len[lo + int(short[0])])} frames; the "
                                 f"speech encoder needs >= {self.audio_min_frames} (its valid convolutions + two pooled frames for the "
                                 f"unbiased std); {short.size} such utterance(s) in the list")
        ba = plan_batches(dataset.audio_len[lo:hi], self.batch, self.waste, self.aq) if self.pa is not None and n else []
        c0, c1 = int(dataset.clip_ptr[lo]), int(dataset.clip_ptr[hi])
        if n and self.pv is not None:
            short = np.flatnonzero(np.asarray(dataset.clip_len[c0:c1]) < self.video_min_frames)
            if short.size:
                raise ValueError(f"RaggedExtractor: lip clip {c0 + int(short[0])} has {int(dataset.clip_len[c0 + int(short[0])])} frames; "
                                 f"needs >= {self.video_min_frames}")
            empty = np.flatnonzero(np.diff(np.asarray(dataset.clip_ptr[lo:hi + 1])) < 1)
            if empty.size:
                raise ValueError(f"RaggedExtractor: utterance {lo + int(empty[0])} has no lip clip (the mean over its clip files, "
                                 "train_fusion.py:349, would be 0 / 0)")
        bv = plan_batches(dataset.clip_len[c0:c1], self.clip_batch, self.waste, self.vq) if self.pv is not None and n else []
        ta = torch.empty((len(ba) * self.batch, D), device=dev) if ba else None           # rows in submission order
        tv = torch.empty((len(bv) * self.clip_batch, D), device=dev) if bv else None
        cache = host_cache if host_cache is not None else {}

        def host(kind, i):
            key = (kind, lo, hi, i, u8)
            if key not in cache:
                cache[key] = self._audio_host(dataset, lo, ba[i]) if kind == "a" else self._video_host(dataset, c0, bv[i], u8)
            hb = cache[key]
            if host_cache is None:
                del cache[key]
            return hb

        # feed the two pipelines alternately, in proportion to their batch counts
        ia = iv = 0
        with torch.no_grad():
            while ia < len(ba) or iv < len(bv):
                take_v = iv < len(bv) and (ia >= len(ba) or iv * len(ba) <= ia * len(bv))
                if take_v:
                    self.pv.submit(host("v", iv), [tv], iv * self.clip_batch, self.clip_batch)
                    iv += 1
                else:
                    self.pa.submit(host("a", ia), [ta], ia * self.batch, self.batch)
                    ia += 1
            for p in (self.pa, self.pv):
                if p is not None:
                    p.finish()
        if ba:
            order = torch.from_numpy(np.concatenate([b.idx for b in ba])).to(dev)
            rows = torch.from_numpy(np.concatenate([i * self.batch + np.arange(len(b.idx)) for i, b in enumerate(ba)])).to(dev)
            xa = torch.empty((n, D), device=dev)
            xa[order] = ta[rows]                                                        # un-sort: list order
        if bv:
            order = torch.from_numpy(np.concatenate([b.idx for b in bv])).to(dev)
            rows = torch.from_numpy(np.concatenate([i * self.clip_batch + np.arange(len(b.idx)) for i, b in enumerate(bv)])).to(dev)
            cm = torch.empty((c1 - c0, D), device=dev)
            cm[order] = tv[rows]                                                        # clip means in (utterance, clip) order
            ptr = torch.from_numpy((dataset.clip_ptr[lo:hi + 1] - c0).astype(np.int32)).to(dev)
            xv = ops.group_mean(cm, ptr)                                                # mean over the utterance's clip files
        self.stats = {
            "audio_batches": len(ba), "audio_shapes": len({b.T for b in ba}), "video_batches": len(bv), "video_shapes": len({b.T for b in bv}),
            "audio_padding_overhead": round(padding_overhead(dataset.audio_len[lo:hi], ba, self.batch), 4) if ba else 0.0,
            "video_padding_overhead": round(padding_overhead(dataset.clip_len[c0:c1], bv, self.clip_batch), 4) if bv else 0.0,
            "valid_audio_frames": int(np.sum(dataset.audio_len[lo:hi])) if ba else 0,
            "valid_video_frames": int(np.sum(dataset.clip_len[c0:c1])) if bv else 0,
            "plans_recorded": (self.pa.recorded if self.pa else 0) + (self.pv.recorded if self.pv else 0),
            "f32_reruns": (self.pa.reruns if self.pa else 0) + (self.pv.reruns if self.pv else 0),
        }
        _lib.check_range(sync=True)
        return xa, xv

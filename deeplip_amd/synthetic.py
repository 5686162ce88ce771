"""Seeded synthetic A+V data with the shapes the reference's loaders produce
(models/fusion_models/datasets.py:115-156: ``(feats_video: list[list[np[T,88,88]]], feats_audio
[B,F,T], labels [B])``; test sets yield one utterance at a time) and trial lists shaped like
database/trial_grid_v1.txt (20 000 lines, 4 000 target / 16 000 non-target).  There is no network
and no dataset in this environment; the reference's disk/codec-bound loaders are out of scope
(SURVEY.md section 2.1 #13-15)."""
from __future__ import annotations

from typing import List, Tuple

import zlib

import numpy as np
import torch

from . import weightgen as wg


class SyntheticAVSet:
    """``n_spk`` speakers x ``utt_per_spk`` utterances; utterance u has ``clips_per_utt`` lip clips
    [T,88,88] and one [F,Ta] feature matrix; the speaker shapes both modalities.

    ``ragged=True``: what a real test list looks like (25 834 utterances of differing duration; long videos
    pre-split into several clip files per utterance, models/fusion_models/datasets.py:143-150) -- utterance u has its own number
    of audio frames in ``audio_range`` and 1 .. ``clips_per_utt`` clips, each with its own length in ``video_range``, all drawn
    from a generator keyed by (key, seed).  ``audio_len`` [N], ``clip_len`` [Nc], ``clip_ptr`` [N+1] (CSR: the clips of
    utterance u are clip_ptr[u] .. clip_ptr[u+1]) describe the set; ``audio_item`` / ``clip_item`` yield single items at their own
    length (the reference's loop), ``audio_padded`` / ``clips_padded`` the zero-padded batches of pad_packed_collate."""

    def __init__(self, n_spk: int, utt_per_spk: int, clips_per_utt: int = 1, video_frames: int = 29,
                 audio_dim: int = 24, audio_frames: int = 300, key: str = "synth", seed: int = wg.DEFAULT_SEED,
                 ragged: bool = False, audio_range: Tuple[int, int] = (137, 412), video_range: Tuple[int, int] = (11, 75),
                 session: float = 1.0, jitter: float = 1.0):
        self.n_spk, self.utt_per_spk, self.clips = n_spk, utt_per_spk, clips_per_utt
        self.T, self.F, self.Ta, self.key, self.seed = video_frames, audio_dim, audio_frames, key, seed
        # every utterance of a speaker shares the speaker's generator (weightgen: key + ".spk<s>") and has its own utterance
        # generator (utt_ids); ``session`` / ``jitter`` = the per-utterance variability beside the speaker's (weightgen.audio_input /
        # video_input) -- they set how separable the speakers are, i.e. where a trial list's EER lands
        self.session, self.jitter = float(session), float(jitter)
        self.utts = [(s, u) for s in range(n_spk) for u in range(utt_per_spk)]
        # ids are prefix-free (zero-padded utterance index): the reference's readers glob ``<pattern>*`` for an utterance's clip
        # files (models/fusion_models/utils.py:456-463), so "s3_u1" must not be a prefix of "s3_u10"
        w = max(2, len(str(max(utt_per_spk - 1, 0))))
        self.utt_ids = [f"s{s}/s{s}_u{u:0{w}d}.wav" for s, u in self.utts]
        self.ragged = bool(ragged)
        n = len(self.utts)
        if self.ragged:
            r = np.random.Generator(np.random.PCG64([int(seed) & 0xFFFFFFFF, zlib.crc32((key + ".ragged").encode())]))
            self.audio_len = r.integers(audio_range[0], audio_range[1] + 1, size=n).astype(np.int64)
            n_clips = r.integers(1, clips_per_utt + 1, size=n) if clips_per_utt > 0 else np.zeros((n,), dtype=np.int64)
            self.clip_ptr = np.concatenate([[0], np.cumsum(n_clips)]).astype(np.int32)
            self.clip_len = r.integers(video_range[0], video_range[1] + 1, size=int(self.clip_ptr[-1])).astype(np.int64)
        else:
            self.audio_len = np.full((n,), audio_frames, dtype=np.int64)
            self.clip_ptr = (np.arange(n + 1) * clips_per_utt).astype(np.int32)
            self.clip_len = np.full((n * clips_per_utt,), video_frames, dtype=np.int64)
        self.clip_utt = np.repeat(np.arange(n), np.diff(self.clip_ptr))          # clip -> its utterance

    def __len__(self):
        return len(self.utts)

    def labels(self, idx) -> np.ndarray:
        return np.array([self.utts[i][0] for i in idx], dtype=np.int64)

    # ---- single items at their own length (what the reference's test loop feeds, train_fusion.py:334-349)
    def audio_item(self, i: int) -> np.ndarray:
        """[F, audio_len[i]]"""
        return wg.audio_input(1, self.F, int(self.audio_len[i]), self.seed, key=f"{self.key}.a", speakers=[self.utts[i][0]],
                              utt_ids=[i], session=self.session)[0]

    def clip_item(self, c: int) -> np.ndarray:
        """[clip_len[c], 88, 88] normalised gray"""
        u = int(self.clip_utt[c])
        k = c - int(self.clip_ptr[u])
        return wg.video_input(1, int(self.clip_len[c]), 88, self.seed, key=f"{self.key}.v", speakers=[self.utts[u][0]],
                              utt_ids=[f"{u}.{k}"], jitter=self.jitter)[0, 0]

    # ---- zero-padded batches + lengths (pad_packed_collate, models/video_models/dataset.py:123-139)
    def audio_padded(self, idx, T: int = None, rows: int = None) -> Tuple[np.ndarray, np.ndarray]:
        from .ragged import pad_stack
        items = [self.audio_item(i) for i in idx]
        L = np.array([it.shape[1] for it in items], dtype=np.int32)
        return pad_stack(items, int(T or L.max()), axis=1, rows=rows), L

    def clips_padded(self, clip_idx, T: int = None, rows: int = None) -> Tuple[np.ndarray, np.ndarray]:
        """-> ([n,1,T,88,88], lengths [n])"""
        from .ragged import pad_stack
        items = [self.clip_item(c)[None] for c in clip_idx]                      # [1,T_c,88,88]
        L = np.array([it.shape[1] for it in items], dtype=np.int32)
        return pad_stack(items, int(T or L.max()), axis=1, rows=rows), L

    # ---- rectangular batches (every item the same length)
    def audio(self, idx) -> np.ndarray:
        if self.ragged:
            raise ValueError("SyntheticAVSet.audio: a ragged set has no rectangular batches; use audio_padded / audio_item")
        spk = [self.utts[i][0] for i in idx]
        out = np.empty((len(idx), self.F, self.Ta), dtype=np.float32)
        for j, i in enumerate(idx):
            out[j] = wg.audio_input(1, self.F, self.Ta, self.seed, key=f"{self.key}.a", speakers=[spk[j]], utt_ids=[i],
                                    session=self.session)[0]
        return out

    def video(self, idx) -> Tuple[np.ndarray, np.ndarray]:
        """All clips of the utterances in ``idx`` as one [G,1,T,88,88] batch + CSR group offsets."""
        if self.ragged:
            raise ValueError("SyntheticAVSet.video: a ragged set has no rectangular batches; use clips_padded / clip_item")
        clips, ptr = [], [0]
        for i in idx:
            s = self.utts[i][0]
            for c in range(self.clips):
                clips.append(wg.video_input(1, self.T, 88, self.seed, key=f"{self.key}.v", speakers=[s], utt_ids=[f"{i}.{c}"],
                                            jitter=self.jitter)[0])
            ptr.append(len(clips))
        return np.stack(clips), np.asarray(ptr, dtype=np.int32)


def frames_u8_from_clips(clips: np.ndarray, rgb: bool = True) -> np.ndarray:
    """Normalised float clips [G,1,T,H,W] -> the uint8 frames a loader would have held before dataloaders.py:17-24 normalised
    them: gray = round(255 * (x * 0.165 + 0.421)), as [G,T,3,H,W] RGB with R = G = B (BASELINE.json's input shape) or
    [G,T,H,W] gray.  (The engine's ingest of these frames gives the clip back up to the uint8 quantisation.)"""
    g = np.clip(np.rint((clips[:, 0].astype(np.float64) * 0.165 + 0.421) * 255.0), 0, 255).astype(np.uint8)
    return np.ascontiguousarray(np.repeat(g[:, :, None], 3, axis=2)) if rgb else g


def synthetic_trials(dataset: SyntheticAVSet, n_trials: int = 20000, n_target: int = 4000, seed: int = 3):
    """(labels [n], pairs [(utt1, utt2)]) with the target / non-target split of trial_grid_v1.txt."""
    r = np.random.Generator(np.random.PCG64(seed))
    by_spk = {}
    for i, (s, _) in enumerate(dataset.utts):
        by_spk.setdefault(s, []).append(i)
    spks = sorted(by_spk)
    y, pairs = [], []
    for t in range(n_trials):
        if t < n_target:
            s = spks[r.integers(len(spks))]
            a, b = r.choice(by_spk[s], 2, replace=len(by_spk[s]) < 2)
            y.append(1)
        else:
            s1, s2 = r.choice(len(spks), 2, replace=False)
            a = by_spk[spks[s1]][r.integers(len(by_spk[spks[s1]]))]
            b = by_spk[spks[s2]][r.integers(len(by_spk[spks[s2]]))]
            y.append(0)
        pairs.append((dataset.utt_ids[a], dataset.utt_ids[b]))
    perm = r.permutation(n_trials)
    return np.asarray(y, dtype=np.int64)[perm], [pairs[i] for i in perm]

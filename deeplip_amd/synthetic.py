"""Seeded synthetic A+V data with the shapes the reference's loaders produce
(models/fusion_models/datasets.py:115-156: ``(feats_video: list[list[np[T,88,88]]], feats_audio
[B,F,T], labels [B])``; test sets yield one utterance at a time) and trial lists shaped like
database/trial_grid_v1.txt (20 000 lines, 4 000 target / 16 000 non-target).  There is no network
and no dataset in this environment; the reference's disk/codec-bound loaders are out of scope
(SURVEY.md section 2.1 #13-15)."""
from __future__ import annotations

from typing import List, Tuple

import numpy as np
import torch

from . import weightgen as wg


class SyntheticAVSet:
    """``n_spk`` speakers x ``utt_per_spk`` utterances; utterance u has ``clips_per_utt`` lip clips
    [T,88,88] and one [F,Ta] feature matrix; the speaker shapes both modalities."""

    def __init__(self, n_spk: int, utt_per_spk: int, clips_per_utt: int = 1, video_frames: int = 29,
                 audio_dim: int = 24, audio_frames: int = 300, key: str = "synth", seed: int = wg.DEFAULT_SEED):
        self.n_spk, self.utt_per_spk, self.clips = n_spk, utt_per_spk, clips_per_utt
        self.T, self.F, self.Ta, self.key, self.seed = video_frames, audio_dim, audio_frames, key, seed
        self.utts = [(s, u) for s in range(n_spk) for u in range(utt_per_spk)]
        self.utt_ids = [f"s{s}/s{s}_u{u}.wav" for s, u in self.utts]

    def __len__(self):
        return len(self.utts)

    def labels(self, idx) -> np.ndarray:
        return np.array([self.utts[i][0] for i in idx], dtype=np.int64)

    def audio(self, idx) -> np.ndarray:
        spk = [self.utts[i][0] for i in idx]
        out = np.empty((len(idx), self.F, self.Ta), dtype=np.float32)
        for j, i in enumerate(idx):
            out[j] = wg.audio_input(1, self.F, self.Ta, self.seed, key=f"{self.key}.a.{i}", speakers=[spk[j]])[0]
        return out

    def video(self, idx) -> Tuple[np.ndarray, np.ndarray]:
        """All clips of the utterances in ``idx`` as one [G,1,T,88,88] batch + CSR group offsets."""
        clips, ptr = [], [0]
        for i in idx:
            s = self.utts[i][0]
            for c in range(self.clips):
                clips.append(wg.video_input(1, self.T, 88, self.seed, key=f"{self.key}.v.{i}.{c}", speakers=[s])[0])
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

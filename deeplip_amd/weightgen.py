"""Name-keyed deterministic tensor generator.

The reference ships no checkpoints and no tests, and its weights (144 MB)
can never be committed.  Every synthetic weight / buffer / input used by the golden-vector
capture script, the parity tests, ``smoke()`` and ``bench.py`` is therefore produced by
``gen(key, shape, seed)``: a pure function of the state-dict key, the shape and a seed, so
the GPU box regenerates bit-identical tensors from key names alone.

Value ranges are chosen so activations stay O(1) through the 18 conv layers of the video
encoder and the 10 TDNN layers of the audio encoder (BN gamma in [0.8,1.2], running_var in
[0.5,1.5], PReLU slopes in [0.1,0.3], fan-in scaled conv/linear weights).
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Mapping, Sequence, Tuple

import numpy as np

DEFAULT_SEED = 1  # echoes SEED = 1 of the reference's train_video.py:70


def _rng(key: str, seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64([int(seed) & 0xFFFFFFFF, zlib.crc32(key.encode("utf-8"))]))


def gen(key: str, shape: Sequence[int], seed: int = DEFAULT_SEED, kind: str = "normal",
        lo: float = 0.0, hi: float = 1.0, std: float = 1.0) -> np.ndarray:
    """Deterministic float32 array for ``key``.  kind: 'normal' (0, std) or 'uniform' [lo, hi)."""
    r = _rng(key, seed)
    shape = tuple(int(s) for s in shape)
    if kind == "normal":
        a = r.standard_normal(shape, dtype=np.float64) * std
    elif kind == "uniform":
        a = r.random(shape, dtype=np.float64) * (hi - lo) + lo
    else:
        raise ValueError(kind)
    return a.astype(np.float32)


def _is_bn_prefix(prefix: str, keys: Iterable[str]) -> bool:
    return (prefix + ".running_mean") in keys


def fill_state_dict(shapes: Mapping[str, Tuple[int, ...]], seed: int = DEFAULT_SEED,
                    prefix: str = "") -> Dict[str, np.ndarray]:
    """Generate a whole state dict from ``{key: shape}``.

    Classification is by key suffix only (so it needs no module objects):
      *.running_var          U[0.5, 1.5]
      *.running_mean         N(0, 0.1)
      *.num_batches_tracked  int64 0
      BN .weight / .bias     U[0.8, 1.2] / N(0, 0.1)      (sibling running_mean exists)
      1-D .weight otherwise  U[0.1, 0.3]                   (PReLU slopes)
      1-D .bias otherwise    N(0, 0.05)                    (conv / linear bias)
      >=2-D tensors          N(0, 1/sqrt(fan_in))          (conv, linear, LMCL.weights, ...)
    ``prefix`` is prepended to the key that seeds the generator (lets two instances of the
    same architecture get different weights).
    """
    keys = set(shapes.keys())
    out: Dict[str, np.ndarray] = {}
    for k, shp in shapes.items():
        shp = tuple(int(s) for s in shp)
        gk = prefix + k
        mod, _, leaf = k.rpartition(".")
        if leaf == "num_batches_tracked":
            out[k] = np.zeros(shp, dtype=np.int64)
        elif leaf == "running_var":
            out[k] = gen(gk, shp, seed, "uniform", 0.5, 1.5)
        elif leaf == "running_mean":
            out[k] = gen(gk, shp, seed, "normal", std=0.1)
        elif _is_bn_prefix(mod, keys):
            if leaf == "weight":
                out[k] = gen(gk, shp, seed, "uniform", 0.8, 1.2)
            else:
                out[k] = gen(gk, shp, seed, "normal", std=0.1)
        elif len(shp) <= 1:
            if leaf == "bias":
                out[k] = gen(gk, shp, seed, "normal", std=0.05)
            else:
                out[k] = gen(gk, shp, seed, "uniform", 0.1, 0.3)
        else:
            fan_in = int(np.prod(shp[1:]))
            out[k] = gen(gk, shp, seed, "normal", std=1.0 / np.sqrt(fan_in))
    return out


def video_input(batch: int, frames: int = 29, size: int = 88, seed: int = DEFAULT_SEED,
                key: str = "input.video", speakers: Sequence[int] | None = None, utt_ids: Sequence | None = None,
                jitter: float = 0.3) -> np.ndarray:
    """Synthetic normalised grayscale lip clips ``[B, 1, T, H, W]``.

    Each clip is a speaker-specific sum of three drifting low-frequency gratings plus a small
    per-utterance phase jitter and pixel noise, clipped to [0,1] and normalised with
    (x - 0.421) / 0.165 -- the pixel statistics after the reference's
    Normalize(0,255) -> CenterCrop(88) -> Normalize(0.421, 0.165)
    (models/video_models/dataloaders.py:11-22).  White noise alone gives embeddings that are
    0.9997-correlated across clips (bias dominated), which would make cosine / argmax parity
    tests vacuous; the speaker structure makes target trials score higher than non-target ones.
    ``speakers[i]`` defaults to ``i`` (every clip its own speaker).  ``utt_ids[i]`` (default ``i``) names the utterance-level
    generator: a dataset that draws its clips ONE PER CALL passes its own utterance / clip id here and one shared ``key``, so that
    clips of one speaker share the speaker's gratings and differ only by the utterance jitter (``jitter`` radians of phase) and
    noise -- with the utterance folded into ``key`` instead (as deeplip_amd.synthetic did until round 5) every utterance was its
    own "speaker" and trial scores carried no speaker information: EER 0.50 by construction."""
    out = np.empty((batch, 1, frames, size, size), dtype=np.float32)
    u = (np.arange(size, dtype=np.float64) / size)
    tt = (np.arange(frames, dtype=np.float64) / max(frames, 1))
    for i in range(batch):
        s = i if speakers is None else int(speakers[i])
        rs = _rng(f"{key}.spk{s}", seed)
        ru = _rng(f"{key}.utt{i if utt_ids is None else utt_ids[i]}", seed)
        img = np.full((frames, size, size), 0.45, dtype=np.float64)
        for _ in range(3):
            fx, fy = rs.uniform(0.5, 4.0, 2)
            ft = rs.uniform(0.0, 2.0)
            ph = rs.uniform(0.0, 2 * np.pi) + jitter * ru.standard_normal()
            amp = rs.uniform(0.08, 0.2)
            img += amp * np.sin(2 * np.pi * (fx * u[None, None, :] + fy * u[None, :, None]
                                             + ft * tt[:, None, None]) + ph)
        img += 0.05 * ru.standard_normal((frames, size, size))
        np.clip(img, 0.0, 1.0, out=img)
        out[i, 0] = ((img - 0.421) / 0.165).astype(np.float32)
    return out


def audio_input(batch: int, feat_dim: int = 24, frames: int = 300, seed: int = DEFAULT_SEED,
                key: str = "input.audio", speakers: Sequence[int] | None = None, utt_ids: Sequence | None = None,
                session: float = 0.0) -> np.ndarray:
    """Synthetic acoustic features ``[B, F, T]``: speaker-specific spectral envelope +
    slow per-channel modulation + unit-ish noise (same rationale as ``video_input``; ``utt_ids`` as there).  ``session``: the
    standard deviation of a per-UTTERANCE envelope offset (channel / session variability) beside the speaker's 0.8-sigma envelope
    -- what keeps a trial list's EER away from zero."""
    out = np.empty((batch, feat_dim, frames), dtype=np.float32)
    tt = np.arange(frames, dtype=np.float64) / 100.0
    for i in range(batch):
        s = i if speakers is None else int(speakers[i])
        rs = _rng(f"{key}.spk{s}", seed)
        ru = _rng(f"{key}.utt{i if utt_ids is None else utt_ids[i]}", seed)
        env = rs.standard_normal(feat_dim)
        fm = rs.uniform(0.2, 3.0, feat_dim)
        ph = rs.uniform(0, 2 * np.pi, feat_dim) + 0.3 * ru.standard_normal(feat_dim)
        x = (0.8 * env[:, None] + 0.5 * np.sin(2 * np.pi * fm[:, None] * tt[None, :] + ph[:, None])
             + 0.6 * ru.standard_normal((feat_dim, frames)))
        if session:
            x = x + session * _rng(f"{key}.sess{i if utt_ids is None else utt_ids[i]}", seed).standard_normal(feat_dim)[:, None]
        out[i] = x.astype(np.float32)
    return out


def labels(batch: int, n_spk: int) -> np.ndarray:
    """``arange(B) % n_spk`` int64."""
    return (np.arange(batch) % n_spk).astype(np.int64)

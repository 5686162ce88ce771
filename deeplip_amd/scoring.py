"""Trial scoring and EER: mirror of the scoring half of ``models/fusion_models/utils.py``
(:234-283, :331-527; duplicated in models/audio_models/utils.py).

The reference writes one ``.npy`` per utterance, then re-reads two files per trial and calls
sklearn on a 1x1 problem, 20 000 times.  Here the embeddings stay in HBM as an [N, D] table
(``EmbeddingTable``), a trial list is two int32 index vectors, and all trials are scored by one
``dlip_pair_cosine_f32`` launch.  EER follows the reference's recipe
(roc_curve -> brentq(1 - x - interp1d(fpr, tpr)(x)) -> interp1d(fpr, thresholds)(eer)) with a
build-owned numpy implementation of the ROC (sklearn's drop_intermediate semantics included),
so no sklearn/scipy is needed on the product path.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import ops
from ._lib import DeepLipHipError


class EmbeddingTable:
    """Utterance-id -> row of a device-resident [N, D] fp32 table (replaces exp/<run>/test_em/*.npy,
    train_fusion.py:361-364)."""

    def __init__(self, utt_ids: Sequence[str], emb: torch.Tensor):
        if emb.dim() != 2 or emb.shape[0] != len(utt_ids):
            raise ValueError("EmbeddingTable: emb must be [len(utt_ids), D]")
        self.utt_ids = list(utt_ids)
        self.index: Dict[str, int] = {u: i for i, u in enumerate(self.utt_ids)}
        self.emb = emb.contiguous()

    def trial_indices(self, trials: Iterable[Tuple[str, str]]) -> Tuple[torch.Tensor, torch.Tensor]:
        ia, ib = [], []
        for a, b in trials:
            ia.append(self.index[a]); ib.append(self.index[b])
        dev = self.emb.device
        return (torch.tensor(ia, dtype=torch.int32, device=dev), torch.tensor(ib, dtype=torch.int32, device=dev))

    def save_npy_tree(self, root: str) -> None:
        """Write the reference's on-disk format: one [1, D] .npy per utterance (train_fusion.py:364)."""
        import os
        host = self.emb.cpu().numpy()
        for u, row in zip(self.utt_ids, host):
            p = os.path.join(root, u.replace(".wav", ".npy"))
            os.makedirs(os.path.dirname(p) or ".", exist_ok=True)
            np.save(p, row[None, :])


    @classmethod
    def load_npy_tree(cls, root: str, utt_ids: Sequence[str], device=None, groups: Optional[Dict[str, Sequence[str]]] = None):
        """Read the reference's on-disk store back: one ``[1, D]`` ``.npy`` per utterance under ``root``
        (what ``eer_cos_*`` np.load per trial, utils.py:260-261) -> one device-resident table.  ``groups`` maps
        an utterance id to the clip files whose embeddings are averaged into its row (long videos are stored as
        several clips per utterance: utils.py:456-463, models/fusion_models/datasets.py:143-150); the mean runs
        on the GPU (group-mean kernel) when a device is given."""
        import os
        rows, gptr = [], [0]
        for u in utt_ids:
            files = list(groups[u]) if groups is not None else [u]
            for f in files:
                a = np.load(os.path.join(root, f.replace(".wav", ".npy")))
                rows.append(np.asarray(a, dtype=np.float32).reshape(-1))
            gptr.append(len(rows))
        emb = torch.from_numpy(np.stack(rows, 0))
        if device is not None:
            emb = emb.to(device)
        if groups is not None:
            if not emb.is_cuda:   # the clip-file mean is arithmetic of the path (utils.py:456-463): it runs on the engine or not at all
                raise DeepLipHipError("EmbeddingTable.load_npy_tree: averaging clip files per utterance runs on the GPU "
                                      "(group-mean kernel); pass device=")
            emb = ops.group_mean(emb, torch.tensor(gptr, dtype=torch.int32, device=emb.device))
        return cls(utt_ids, emb)


def all_pairs_cosine(emb: torch.Tensor) -> torch.Tensor:
    """Cosine score of every (enrol, test) pair of a table, [N, N]: rows L2-normalised, then one GEMM on the
    engine -- the dense form of the trial loop (utils.py:251-283) for full score matrices / score
    normalisation cohorts."""
    n = ops.l2_normalize(emb.contiguous())
    if n.shape[1] % 4:
        raise ValueError("all_pairs_cosine: embedding dimension must be a multiple of 4")
    return ops.linear(n, n)


def read_trial_list(path: str) -> Tuple[np.ndarray, List[Tuple[str, str]]]:
    """`label utt1 utt2` per line (database/trial_grid_v1.txt; utils.py:256-259)."""
    y, pairs = [], []
    with open(path) as f:
        for line in f:
            line = line.rstrip()
            if not line:
                continue
            lab, u1, u2 = line.split(" ")
            y.append(int(lab)); pairs.append((u1, u2))
    return np.asarray(y, dtype=np.int64), pairs


def cosine_scores(emb: torch.Tensor, idx_a: torch.Tensor, idx_b: torch.Tensor) -> torch.Tensor:
    """sklearn-cosine of emb[idx_a[i]] and emb[idx_b[i]] for every trial (utils.py:262)."""
    return ops.pair_cosine(emb, idx_a, idx_b, mode=0)


def score_fusion(audio_emb, video_emb, idx_a, idx_b) -> torch.Tensor:
    """eer_cos_*_scorefusion (utils.py:343-377): 0.5*cos(audio) + 0.5*F.cosine_similarity(video, eps=1e-8)."""
    s = ops.pair_cosine(audio_emb, idx_a, idx_b, mode=0, weight=0.5)
    return ops.pair_cosine(video_emb, idx_a, idx_b, mode=1, eps=1e-8, weight=0.5, out=s)


def feature_fusion_scores(audio_emb, video_emb, idx_a, idx_b) -> torch.Tensor:
    """eer_cos_*_featurefusion (utils.py:465-473): hstack(znorm_biased(video), znorm_biased(audio)) then cosine."""
    fused = ops.znorm_cat(video_emb.contiguous(), audio_emb.contiguous(), biased=True)
    return ops.pair_cosine(fused, idx_a, idx_b, mode=0)


# ------------------------------------------------------------------------------------------
# EER on the host (20 000 scalars; the reference does this with sklearn + scipy on the CPU too)
# ------------------------------------------------------------------------------------------
def roc_curve(y_true: np.ndarray, y_score: np.ndarray, pos_label: int = 1):
    """sklearn.metrics.roc_curve(y_true, y_score, pos_label, drop_intermediate=True) semantics:
    stable descending sort, one point per distinct score, collinear interior points dropped,
    (0,0) prepended with threshold +inf."""
    y_true = (np.asarray(y_true).ravel() == pos_label)
    y_score = np.asarray(y_score, dtype=np.float64).ravel()
    order = np.argsort(y_score, kind="mergesort")[::-1]
    y_score, y_true = y_score[order], y_true[order]
    distinct = np.where(np.diff(y_score))[0]
    idx = np.r_[distinct, y_true.size - 1]
    tps = np.cumsum(y_true, dtype=np.float64)[idx]
    fps = 1 + idx - tps
    thr = y_score[idx]
    if len(fps) > 2:
        keep = np.where(np.r_[True, np.logical_or(np.diff(fps, 2), np.diff(tps, 2)), True])[0]
        fps, tps, thr = fps[keep], tps[keep], thr[keep]
    tps = np.r_[0, tps]; fps = np.r_[0, fps]; thr = np.r_[np.inf, thr]
    return fps / fps[-1], tps / tps[-1], thr


def _interp(xs: np.ndarray, ys: np.ndarray, x: float) -> float:
    """scipy interp1d(kind='linear') at one point; for repeated xs it takes the segment
    searchsorted(side='left') selects (clipped to [1, n-1])."""
    i = int(np.clip(np.searchsorted(xs, x, side="left"), 1, len(xs) - 1))
    x0, x1, y0, y1 = xs[i - 1], xs[i], ys[i - 1], ys[i]
    if x1 == x0:
        return float(y0)  # 0 * inf guards
    return float(y0 + (y1 - y0) * ((x - x0) / (x1 - x0)))


def _brentq(f, a: float, b: float, xtol: float = 2e-12, rtol: float = 8.881784197001252e-16, maxiter: int = 100):
    """Brent's method with SciPy's defaults (scipy/optimize/Zeros/brentq.c control flow)."""
    xpre, xcur = a, b
    fpre, fcur = f(xpre), f(xcur)
    if fpre == 0:
        return xpre
    if fcur == 0:
        return xcur
    if np.sign(fpre) == np.sign(fcur):
        raise ValueError("f(a) and f(b) must have different signs")
    xblk = fblk = spre = scur = 0.0
    for _ in range(maxiter):
        if fpre != 0 and fcur != 0 and (np.sign(fpre) != np.sign(fcur)):
            xblk, fblk = xpre, fpre
            spre = scur = xcur - xpre
        if abs(fblk) < abs(fcur):
            xpre, xcur, xblk = xcur, xblk, xcur
            fpre, fcur, fblk = fcur, fblk, fcur
        delta = (xtol + rtol * abs(xcur)) / 2
        sbis = (xblk - xcur) / 2
        if fcur == 0 or abs(sbis) < delta:
            return xcur
        if abs(spre) > delta and abs(fcur) < abs(fpre):
            if xpre == xblk:
                stry = -fcur * (xcur - xpre) / (fcur - fpre)
            else:
                dpre = (fpre - fcur) / (xpre - xcur)
                dblk = (fblk - fcur) / (xblk - xcur)
                stry = -fcur * (fblk * dblk - fpre * dpre) / (dblk * dpre * (fblk - fpre))
            if 2 * abs(stry) < min(abs(spre), 3 * abs(sbis) - delta):
                spre, scur = scur, stry
            else:
                spre = scur = sbis
        else:
            spre = scur = sbis
        xpre, fpre = xcur, fcur
        if abs(scur) > delta:
            xcur += scur
        else:
            xcur += delta if sbis > 0 else -delta
        fcur = f(xcur)
    return xcur


def eer_from_scores(y_true, y_pred) -> Tuple[float, float]:
    """utils.py:263-266 -> (eer, threshold)."""
    y_pred = np.asarray([np.asarray(s).reshape(-1)[0] for s in y_pred], dtype=np.float64) \
        if not isinstance(y_pred, np.ndarray) else y_pred.reshape(-1)
    fpr, tpr, thr = roc_curve(np.asarray(y_true), y_pred, pos_label=1)
    e = _brentq(lambda x: 1. - x - _interp(fpr, tpr, x), 0., 1.)
    return float(e), float(_interp(fpr, thr, e))


def eer_cos(table: EmbeddingTable, trial_path: str) -> Tuple[float, float]:
    """eer_cos_lomgrid / eer_cos_grid (utils.py:251-283) over an in-memory table."""
    y, pairs = read_trial_list(trial_path)
    ia, ib = table.trial_indices(pairs)
    s = cosine_scores(table.emb, ia, ib)
    return eer_from_scores(y, s.cpu().numpy())

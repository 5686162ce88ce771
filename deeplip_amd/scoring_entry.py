"""The reference's scoring ENTRY POINTS, callable the way its trainers call them: ``f(exp_dir) -> (eer, threshold)``
(models/fusion_models/utils.py:234-521, duplicated with other default paths in models/audio_models/utils.py;
callers train_fusion.py:430-469, train_audio.py:499-543).

Every function of the reference walks a trial list and, PER TRIAL, np.loads two ``.npy`` files (plus, for the fusion
variants, globs and np.loads the clip files of two utterances), calls sklearn on a 1 x 1 problem and appends one score:
40 000 file reads for 20 000 trials.  Here the same on-disk store is read ONCE per distinct utterance into a
device-resident ``[N, D]`` table (``EmbeddingTable.load_npy_tree``; clip files of an utterance are averaged by the
group-mean kernel), the trial list becomes two int32 index vectors, all trials are scored by one launch of the pair-cosine
/ z-norm / PLDA kernels, and the EER is computed on the host from the 20 000 scores as the reference does.

Paths.  The reference hard-codes its site's files in the function bodies (the trial list relative to the working
directory, the embedding directory under ``exp/<exp_dir>/``, the lip-embedding store under ``/data/liumeng/...``).  Those
defaults are kept, per module, in ``FUSION_DEFAULTS`` / ``AUDIO_DEFAULTS`` below -- one table instead of ten function
bodies -- and each can be overridden per call by keyword (``trial_path=``, ``emb_dir=``, ``video_dir=``,
``video_trial_path=``, ``plda_path=``), per process through ``set_paths(...)``, or by environment variable
(``DLIP_TRIAL_LIST``, ``DLIP_EMB_DIR``, ``DLIP_VIDEO_EMBEDDING_DIR``, ``DLIP_VIDEO_TRIAL_LIST``, ``DLIP_PLDA_MODEL``).
The positional signature stays the reference's: ONE argument.

There is no CPU path: the tables live on the current ROCm device and the scores come from ``dlip_*`` launches.
"""
from __future__ import annotations

import os
from glob import glob
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import ops, scoring
from ._lib import DeepLipHipError

_SITE = "/data/liumeng/Lipreading_using_Temporal_Convolutional_Networks/"

# name -> defaults, as written in models/fusion_models/utils.py (line numbers in the comments)
FUSION_DEFAULTS: Dict[str, Dict[str, str]] = {
    "eer": dict(trial="task.txt", sub="test_xv"),                                                           # :234-249
    "eer_cos_lomgrid": dict(trial="data/data_audio/trial_lomgrid_2w.txt", sub="test_em_lomgrid"),           # :251-266
    "eer_cos_grid": dict(trial="data/data_audio/trial_grid_2w.txt", sub="test_em_grid"),                    # :268-283
    "eer_plda_lomgrid": dict(trial="data/trial/A_lomgrid_trial_2w", sub="test_xv_lomgrid", plda="exp/plda.pkl"),   # :285-306
    "eer_plda_grid": dict(trial="data/trial/A_grid_trial_2w", sub="test_xv_grid", plda="exp/plda.pkl"),     # :308-329
    "eer_cos_lomgrid_scorefusion": dict(trial="data/trial/A_lomgrid_trial_2w", sub="test_xv_lomgrid",       # :331-382
                                        video_dir=_SITE + "datasets_lombardgrid/",
                                        video_trial=_SITE + "preprocessing/lombardgrid_trial_2w"),
    "eer_cos_grid_scorefusion": dict(trial="data/trial/A_grid_trial_2w", sub="test_xv_grid",                # :384-435
                                     video_dir=_SITE + "datasets_grid/", video_trial=_SITE + "preprocessing/grid/grid_trial_2w"),
    "eer_cos_lomgrid_featurefusion": dict(trial="data/trial/A_lomgrid_trial_2w", sub="test_xv_lomgrid",     # :437-479
                                          video_dir=_SITE + "datasets_lombardgrid/", pattern="spk/utt"),
    "eer_cos_grid_featurefusion": dict(trial="data/trial/A_grid_trial_2w", sub="test_xv_grid",              # :481-522
                                       video_dir=_SITE + "datasets_grid/", pattern="utt"),
}
# models/audio_models/utils.py differs in two functions only (its :254,:259-260,:271,:276-277)
AUDIO_DEFAULTS: Dict[str, Dict[str, str]] = {k: dict(v) for k, v in FUSION_DEFAULTS.items()}
AUDIO_DEFAULTS["eer_cos_lomgrid"] = dict(trial="data/trial/A_lomgrid_trial_2w", sub="test_xv_lomgrid")
AUDIO_DEFAULTS["eer_cos_grid"] = dict(trial="data/trial/A_grid_trial_2w", sub="test_xv_grid")

_ENV = {"trial": "DLIP_TRIAL_LIST", "emb_dir": "DLIP_EMB_DIR", "video_dir": "DLIP_VIDEO_EMBEDDING_DIR",
        "video_trial": "DLIP_VIDEO_TRIAL_LIST", "plda": "DLIP_PLDA_MODEL"}
_process_paths: Dict[str, Dict[str, str]] = {}


def set_paths(name: Optional[str] = None, **paths) -> None:
    """Process-wide override of a function's default files (``name=None``: of every function), e.g.
    ``set_paths("eer_cos_lomgrid", trial="exp/run/trials.txt")``.  Keys: trial, emb_dir, video_dir, video_trial, plda.
    ``set_paths(name)`` without keywords clears the override."""
    key = name or "*"
    if not paths:
        _process_paths.pop(key, None)
        return
    bad = set(paths) - set(_ENV)
    if bad:
        raise KeyError(f"set_paths: unknown keys {sorted(bad)} (known: {sorted(_ENV)})")
    _process_paths.setdefault(key, {}).update({k: str(v) for k, v in paths.items()})


def _resolve(name: str, defaults: Dict[str, str], exp_dir: str, kw: Dict[str, Optional[str]]) -> Dict[str, str]:
    """keyword > set_paths(name) > set_paths() > environment > the reference's literal."""
    out = dict(defaults)
    out["emb_dir"] = None
    for k, env in _ENV.items():
        v = kw.get(k)
        if v is None:
            v = _process_paths.get(name, {}).get(k)
        if v is None:
            v = _process_paths.get("*", {}).get(k)
        if v is None:
            v = os.environ.get(env)
        if v is not None:
            out[k] = v
    if out["emb_dir"] is None:
        d = os.path.join("exp/{}".format(exp_dir), out["sub"])
        if not os.path.isdir(d):
            # the reference's fusion trainer WRITES exp/<run>/test_em/test_em_lomgrid (train_fusion.py:332,362) while its
            # eer_cos_lomgrid READS exp/<run>/test_em_lomgrid (utils.py:259): an upstream mismatch.  Read where the writer wrote.
            alt = os.path.join("exp/{}".format(exp_dir), out["sub"].split("_")[0] + "_" + out["sub"].split("_")[1], out["sub"])
            if os.path.isdir(alt):
                d = alt
        out["emb_dir"] = d
    return out


def _device(device) -> torch.device:
    if device is not None:
        return torch.device(device)
    if not torch.cuda.is_available():
        raise DeepLipHipError("scoring entry points need a ROCm GPU: trials are scored by dlip_* launches, there is no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


def _read_trials(path: str) -> Tuple[np.ndarray, List[Tuple[str, str]]]:
    """`label utt1 utt2` per line (utils.py:254-258; the label is `eval`-ed there: int here)."""
    if not os.path.exists(path):
        raise FileNotFoundError(f"trial list {path!r} not found (the reference's default path, relative to the working directory; "
                                "pass trial_path=, call scoring_entry.set_paths(...) or set DLIP_TRIAL_LIST)")
    return scoring.read_trial_list(path)


def _unique(seq: Sequence[str]) -> List[str]:
    seen, out = set(), []
    for s in seq:
        if s not in seen:
            seen.add(s); out.append(s)
    return out


def _audio_table(p: Dict[str, str], pairs, dev) -> Tuple[scoring.EmbeddingTable, torch.Tensor, torch.Tensor]:
    """One table row per DISTINCT utterance of the trial list (np.load once each; utils.py:260-261 loads per trial)."""
    utts = _unique([u for ab in pairs for u in ab])
    table = scoring.EmbeddingTable.load_npy_tree(p["emb_dir"], utts, device=dev)
    ia, ib = table.trial_indices(pairs)
    return table, ia, ib


def _video_table(video_dir: str, patterns: Sequence[str], dev) -> scoring.EmbeddingTable:
    """The lip-embedding store of the fusion variants (utils.py:352-370, :451-463): for utterance pattern p the files
    ``sorted(glob((video_dir + p + '*').replace('datasets', 'embedding')))``, each an ``.npz`` whose ``data`` is the lip-clip
    model's ``[1, T, 512]`` output (train_video.py:212); the utterance embedding is the mean over files of the mean over
    frames.  Both means run on the GPU: the frames of all files are ONE ``[sum T, 512]`` upload, the group-mean kernel
    reduces it per file, then per utterance."""
    rows, fptr, uptr = [], [0], [0]
    nrow = 0
    for pat in patterns:
        files = sorted(glob((video_dir + pat + "*").replace("datasets", "embedding")))
        if not files:
            raise FileNotFoundError(f"no lip-embedding files match {(video_dir + pat + '*').replace('datasets', 'embedding')!r} "
                                    "(pass video_dir= / set DLIP_VIDEO_EMBEDDING_DIR)")
        for f in files:
            a = np.load(f)["data"]
            a = np.asarray(a, dtype=np.float32)
            a = a.squeeze(-3) if a.ndim >= 3 else a                         # utils.py:365: [1,T,512] -> [T,512]
            a = a.reshape(-1, a.shape[-1])
            rows.append(a); nrow += a.shape[0]; fptr.append(nrow)
        uptr.append(len(fptr) - 1)
    x = torch.from_numpy(np.concatenate(rows, 0)).to(dev)
    per_file = ops.group_mean(x, torch.tensor(fptr, dtype=torch.int32, device=dev))           # np.mean(data, 0)
    per_utt = ops.group_mean(per_file, torch.tensor(uptr, dtype=torch.int32, device=dev))      # sum over files / len(files)
    return scoring.EmbeddingTable(list(patterns), per_utt)


def _pattern(kind: str, utt: str) -> str:
    """utils.py:448-449 (lomgrid: `<spk>/<utt>` with spk = the text before the first '_') / :492-493 (grid: `<utt>`)."""
    stem = utt.replace(".wav", "")
    return utt.split("_")[0] + "/" + stem if kind == "spk/utt" else stem


def _finish(y: np.ndarray, s: torch.Tensor, return_scores: bool):
    scores = s.cpu().numpy()
    e = scoring.eer_from_scores(y, scores)
    return (e[0], e[1], scores) if return_scores else e


def load_plda(path: str):
    """``exp/plda.pkl`` (train_audio.py:339-341): a joblib file.  Written by this build's ``Trainer.train_plda`` it is a plain
    dict of arrays (``PLDA.save``); a file written by the REFERENCE pickles a ``plda.Classifier`` of the third-party package
    -- readable only where that package is installed, in which case its fitted parameters (m, A, Psi, relevant dims, PCA) are
    taken over."""
    from .plda import PLDA
    return PLDA.load(path)


def make_entry_points(defaults: Dict[str, Dict[str, str]]) -> Dict[str, Callable]:
    """The ten functions of one ``utils`` module, bound to that module's default paths."""

    def _cos(name):
        def f(exp_dir, *, trial_path=None, emb_dir=None, device=None, return_scores=False):
            p = _resolve(name, defaults[name], exp_dir, dict(trial=trial_path, emb_dir=emb_dir))
            dev = _device(device)
            y, pairs = _read_trials(p["trial"])
            table, ia, ib = _audio_table(p, pairs, dev)
            return _finish(y, scoring.cosine_scores(table.emb, ia, ib), return_scores)      # utils.py:262
        return f

    def _plda(name):
        def f(exp_dir, *, trial_path=None, emb_dir=None, plda_path=None, device=None, return_scores=False):
            p = _resolve(name, defaults[name], exp_dir, dict(trial=trial_path, emb_dir=emb_dir, plda=plda_path))
            dev = _device(device)
            model = load_plda(p["plda"])                                                    # utils.py:286
            y, pairs = _read_trials(p["trial"])
            table, ia, ib = _audio_table(p, pairs, dev)
            return _finish(y, model.score_trials(table.emb, ia, ib), return_scores)         # utils.py:298-303
        return f

    def _scorefusion(name):
        def f(exp_dir, *, trial_path=None, emb_dir=None, video_dir=None, video_trial_path=None, device=None, return_scores=False):
            p = _resolve(name, defaults[name], exp_dir, dict(trial=trial_path, emb_dir=emb_dir, video_dir=video_dir,
                                                             video_trial=video_trial_path))
            dev = _device(device)
            y, pairs = _read_trials(p["trial"])
            table, ia, ib = _audio_table(p, pairs, dev)
            with open(p["video_trial"]) as fh:                                              # utils.py:347-354: `p1 \t p2` per line
                vpairs = [tuple(line.split("\t")[:2]) for line in fh.read().splitlines() if line]
            if len(vpairs) != len(pairs):
                raise ValueError(f"{p['video_trial']}: {len(vpairs)} lip trials for {len(pairs)} speech trials (utils.py:378 adds "
                                 "the two score lists element by element)")
            vt = _video_table(p["video_dir"], _unique([u for ab in vpairs for u in ab]), dev)
            va, vb = vt.trial_indices(vpairs)
            s = ops.pair_cosine(table.emb, ia, ib, mode=0, weight=0.5)                      # 0.5 * sklearn cosine (:341-345)
            s = ops.pair_cosine(vt.emb, va, vb, mode=1, eps=1e-8, weight=0.5, out=s)        # + 0.5 * F.cosine_similarity (:372-378)
            return _finish(y, s, return_scores)
        return f

    def _featurefusion(name):
        def f(exp_dir, *, trial_path=None, emb_dir=None, video_dir=None, device=None, return_scores=False):
            p = _resolve(name, defaults[name], exp_dir, dict(trial=trial_path, emb_dir=emb_dir, video_dir=video_dir))
            dev = _device(device)
            y, pairs = _read_trials(p["trial"])
            table, ia, ib = _audio_table(p, pairs, dev)
            vt = _video_table(p["video_dir"], [_pattern(p["pattern"], u) for u in table.utt_ids], dev)   # row i <-> audio row i
            return _finish(y, scoring.feature_fusion_scores(table.emb, vt.emb, ia, ib), return_scores)   # utils.py:465-473
        return f

    out = {}
    for name in defaults:
        mk = _plda if "plda" in name else _scorefusion if name.endswith("scorefusion") else \
            _featurefusion if name.endswith("featurefusion") else _cos
        fn = mk(name)
        fn.__name__ = fn.__qualname__ = name
        fn.__doc__ = (f"{name}(exp_dir) -> (eer, threshold): the reference's entry point of that name (module docstring); defaults "
                      f"{defaults[name]}.")
        out[name] = fn
    return out

"""PLDA back-end scoring: mirror of ``eer_plda_lomgrid`` / ``eer_plda_grid``
(models/fusion_models/utils.py:285-329, duplicated in models/audio_models/utils.py).

The reference loads a pickled classifier of the third-party ``plda`` package (RaviSoji/plda, not vendored, no
version pinned upstream, absent from this image) and, per trial, np.loads two embeddings, maps them with
``model.transform(em, from_space='D', to_space='U_model')`` and calls
``model.calc_same_diff_log_likelihood_ratio``.  Parity is therefore UNPINNED: what is restated here is the
package's published algorithm (Ioffe, "Probabilistic Linear Discriminant Analysis", ECCV 2006, as the package's
optimizer / model implement it):

  fit:        m = mean; S_b, S_w = between / within scatter; W = generalised eigenvectors of (S_b, S_w);
              Lambda_b = W' S_b W, Lambda_w = W' S_w W; n = mean samples per class;
              A = W^-T (n/(n-1) Lambda_w)^1/2;  Psi = max(0, (n-1)/n Lambda_b/Lambda_w - 1/n);
              relevant dims = those with Psi > 0 (largest first)
  transform:  u = (x - m) A^-T, restricted to the relevant dims ("U_model" space: within-class covariance I,
              between-class covariance diag(Psi))
  score:      log p(u1, u2 | same) - log p(u1) - log p(u2)   (closed form per dimension, dlip_plda_llr_f32)

Fitting is a one-off host computation (numpy / scipy eigh, like the EER).  Scoring keeps the embeddings in HBM:
the affine map is ONE GEMM on the engine and all trials are scored by one ``dlip_plda_llr_f32`` launch.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import ops
from ._lib import check, lib, ptr, stream_handle


class PLDA:
    def __init__(self, m: np.ndarray, inv_A: np.ndarray, psi: np.ndarray, relevant: np.ndarray,
                 pca_mean: Optional[np.ndarray] = None, pca_components: Optional[np.ndarray] = None):
        self.m, self.inv_A, self.psi, self.relevant = m, inv_A, psi, relevant
        self.pca_mean, self.pca_components = pca_mean, pca_components
        self._dev = None

    # ---- host: fit (the `plda` package's optimize_maximum_likelihood) ----
    @classmethod
    def fit(cls, X: np.ndarray, labels: Sequence[int], n_principal_components: Optional[int] = None) -> "PLDA":
        from scipy.linalg import eigh
        X = np.asarray(X, dtype=np.float64)
        y = np.asarray(labels)
        pca_mean = pca_comp = None
        if n_principal_components is not None and n_principal_components < X.shape[1]:
            pca_mean = X.mean(0)
            _, _, vt = np.linalg.svd(X - pca_mean, full_matrices=False)
            pca_comp = vt[:n_principal_components]
            X = (X - pca_mean) @ pca_comp.T
        N, D = X.shape
        classes = np.unique(y)
        m = X.mean(0)
        S_b = np.zeros((D, D)); S_w = np.zeros((D, D))
        for c in classes:
            Xc = X[y == c]
            mc = Xc.mean(0)
            S_b += len(Xc) / N * np.outer(mc - m, mc - m)
            S_w += (Xc - mc).T @ (Xc - mc) / N
        n_avg = N / len(classes)
        _, W = eigh(S_b, S_w)                                   # columns: generalised eigenvectors
        Lb = np.diag(W.T @ S_b @ W)
        Lw = np.diag(W.T @ S_w @ W)
        A = np.linalg.inv(W.T) * np.sqrt(n_avg / (n_avg - 1.0) * Lw)   # scales column d by its factor
        psi = (n_avg - 1.0) / n_avg * Lb / Lw - 1.0 / n_avg
        psi[psi <= 0] = 0.0
        order = np.argsort(psi)[::-1]
        relevant = order[: int((psi > 0).sum())]
        return cls(m, np.linalg.inv(A), psi, relevant, pca_mean, pca_comp)

    # ---- exp/plda.pkl (train_audio.py:339-341: joblib.dump(classifier, 'exp/plda.pkl')) ----
    FORMAT = "deeplip_amd.plda/2"

    def save(self, path: str) -> None:
        """This build's file: a numpy ``.npz`` archive of plain arrays written under the reference's file NAME (no pickle inside:
        ``load`` reads it with ``allow_pickle=False``, so loading it executes nothing)."""
        import os
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        arrays = {"format": np.array(self.FORMAT), "m": self.m, "inv_A": self.inv_A, "psi": self.psi, "relevant": np.asarray(self.relevant)}
        if self.pca_components is not None:
            arrays.update(pca_mean=self.pca_mean, pca_components=self.pca_components)
        with open(path, "wb") as f:           # (a file object: np.savez would append ".npz" to a path)
            np.savez(f, **arrays)

    @classmethod
    def load(cls, path: str, allow_pickle: bool = None) -> "PLDA":
        """This build's archive (arrays only, nothing is executed), or -- ONLY ON REQUEST -- the reference's file: there the object is
        a pickled ``plda.Classifier`` (RaviSoji/plda), and unpickling runs whatever code the file names.  ``allow_pickle`` (default:
        the environment's DLIP_ALLOW_PICKLED_CHECKPOINTS=1, the same switch as pickled checkpoints, train_audio.py) lets joblib
        rebuild it where that package is installed; its fitted model's parameters are then taken over (``model.m``,
        ``model.inv_A``, ``model.Psi`` (diagonal), ``model.relevant_U_dims``, ``model.pca``)."""
        import os
        import zipfile
        if os.path.exists(path) and zipfile.is_zipfile(path):
            with np.load(path, allow_pickle=False) as z:
                if not str(z["format"]).startswith("deeplip_amd.plda/"):
                    raise RuntimeError(f"{path}: not a deeplip_amd PLDA archive")
                pca = "pca_components" in z.files
                return cls(z["m"], z["inv_A"], z["psi"], z["relevant"], z["pca_mean"] if pca else None, z["pca_components"] if pca else None)
        if allow_pickle is None:
            allow_pickle = os.environ.get("DLIP_ALLOW_PICKLED_CHECKPOINTS") == "1"
        if not allow_pickle:
            raise RuntimeError(f"{path} is not this build's PLDA archive (a numpy .npz of plain arrays under the reference's file name; "
                               "INTEGRATION.md, 'Formats that differ').  Two older forms are pickles and are only loaded on request -- the "
                               "reference's joblib pickle of a `plda.Classifier`, and the joblib dict of arrays round 4 of this build wrote: "
                               "both run code on load; set DLIP_ALLOW_PICKLED_CHECKPOINTS=1 (or allow_pickle=True) if you trust the file.")
        import joblib
        try:
            obj = joblib.load(path)
        except ModuleNotFoundError as ex:
            raise RuntimeError(f"{path} pickles a classifier of the third-party `plda` package, which is not installed here; "
                               "re-fit with Trainer.train_plda() (writes a plain-array file) or install the package") from ex
        if isinstance(obj, dict) and str(obj.get("format", "")).startswith("deeplip_amd.plda/"):      # round 4's joblib dict of arrays
            return cls(obj["m"], obj["inv_A"], obj["psi"], obj["relevant"], obj.get("pca_mean"), obj.get("pca_components"))
        model = getattr(obj, "model", obj)
        psi = np.asarray(model.Psi, dtype=np.float64)
        psi = np.diag(psi) if psi.ndim == 2 else psi
        pca = getattr(model, "pca", None)
        return cls(np.asarray(model.m, dtype=np.float64), np.asarray(model.inv_A, dtype=np.float64), psi,
                   np.asarray(model.relevant_U_dims), None if pca is None else np.asarray(pca.mean_, dtype=np.float64),
                   None if pca is None else np.asarray(pca.components_, dtype=np.float64))

    # ---- affine map D -> U_model as (weight [Dr, D], bias [Dr]) ----
    def affine(self) -> Tuple[np.ndarray, np.ndarray]:
        Wt = self.inv_A[self.relevant]                          # u = inv_A (x - m)
        b = -Wt @ self.m
        if self.pca_components is not None:                     # x = P (d - pca_mean)
            b = b - Wt @ (self.pca_components @ self.pca_mean)
            Wt = Wt @ self.pca_components
        return Wt, b

    def transform_np(self, X: np.ndarray) -> np.ndarray:
        Wt, b = self.affine()
        return np.asarray(X, dtype=np.float64) @ Wt.T + b

    # ---- engine ----
    def _device_params(self, device):
        if self._dev is None or self._dev[0] != device:
            Wt, b = self.affine()
            self._dev = (device, torch.from_numpy(Wt.astype(np.float32)).contiguous().to(device),
                         torch.from_numpy(b.astype(np.float32)).to(device),
                         torch.from_numpy(self.psi[self.relevant].astype(np.float32)).to(device))
        return self._dev[1:]

    def transform(self, emb: torch.Tensor) -> torch.Tensor:
        """[N, D] embeddings (CUDA) -> [N, Dr] latent vectors: one GEMM on the engine."""
        Wt, b, _ = self._device_params(emb.device)
        if emb.shape[1] % 4:
            raise ValueError("PLDA.transform: embedding dimension must be a multiple of 4")
        return ops.linear(emb.contiguous(), Wt, b)

    def llr(self, u: torch.Tensor, idx_a: torch.Tensor, idx_b: torch.Tensor) -> torch.Tensor:
        """Same/different log-likelihood ratio of every trial (utils.py:300-304) in one launch."""
        _, _, psi = self._device_params(u.device)
        n = idx_a.numel()
        out = torch.empty((n,), device=u.device, dtype=torch.float32)
        check(lib().dlip_plda_llr_f32(ptr(u), u.shape[0], u.shape[1], ptr(psi), ptr(idx_a), ptr(idx_b), ptr(out), n,
                                      stream_handle()), "dlip_plda_llr_f32")
        return out

    def score_trials(self, emb: torch.Tensor, idx_a: torch.Tensor, idx_b: torch.Tensor) -> torch.Tensor:
        return self.llr(self.transform(emb), idx_a, idx_b)


def eer_plda(table, trial_path: str, model: PLDA) -> Tuple[float, float]:
    """eer_plda_lomgrid / eer_plda_grid (utils.py:285-329) over an in-memory embedding table."""
    from . import scoring
    y, pairs = scoring.read_trial_list(trial_path)
    ia, ib = table.trial_indices(pairs)
    s = model.score_trials(table.emb, ia, ib)
    return scoring.eer_from_scores(y, s.cpu().numpy())

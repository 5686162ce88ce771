"""Scoring fuzz: cosine trial scores, score fusion, feature fusion, all-pairs cosine and the EER on RANDOM table sizes / embedding widths / trial
counts (incl. one trial, repeated indices, a == b, widths that are no multiple of anything, rows with tiny and huge norms) against the oracle.
   python tools/probes/score_fuzz.py [n] [seed]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import torch

from deeplip_amd import scoring
from oracle import deeplip_oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
r = np.random.Generator(np.random.PCG64(seed))
bad = 0
worst = 0.0
for i in range(n):
    N = int(r.integers(1, 3000))
    D = int(r.choice([1, 2, 3, 7, 32, 100, 256, 511, 512, 513, 1024, 1500]))
    M = int(r.choice([1, 2, 63, 64, 65, 1000, 20000, 100003]))
    a = r.standard_normal((N, D)).astype(np.float32)
    v = r.standard_normal((N, D)).astype(np.float32)
    if i % 4 == 0:
        a *= np.exp(r.uniform(-20, 20, size=(N, 1))).astype(np.float32)      # rows of very different norms: a cosine must not care
    ia = r.integers(0, N, size=M)
    ib = r.integers(0, N, size=M)
    if M > 2:
        ib[0] = ia[0]
    ea, ev = torch.from_numpy(a).cuda(), torch.from_numpy(v).cuda()
    ta, tb = torch.from_numpy(ia.astype(np.int32)).cuda(), torch.from_numpy(ib.astype(np.int32)).cuda()
    checks = {
        "cosine": (scoring.cosine_scores(ea, ta, tb).cpu().numpy(), O.cosine_trial_scores(a, ia, ib)),
        "score_fusion": (scoring.score_fusion(ea, ev, ta, tb).cpu().numpy(), O.score_fusion(a, v, ia, ib)),
        "feature_fusion": (scoring.feature_fusion_scores(ea, ev, ta, tb).cpu().numpy(), O.feature_fusion_scores(a, v, ia, ib)),
    }
    if N <= 600 and D % 4 == 0:      # (all_pairs_cosine takes widths that are multiples of 4, and says so)
        an = a.astype(np.float64)
        an /= np.linalg.norm(an, axis=1, keepdims=True)
        checks["all_pairs"] = (scoring.all_pairs_cosine(ea).cpu().numpy(), an @ an.T)
    tag = f"N={N} D={D} trials={M} wild_norms={i % 4 == 0}"
    for k, (got, want) in checks.items():
        got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
        both_nan = np.isnan(got) & np.isnan(want)        # (a one-dimensional row has no z-norm: NaN in the reference, NaN here)
        e = float(np.max(np.where(both_nan, 0.0, np.abs(got - want))))       # cosines are O(1): absolute
        worst = max(worst, e)
        if not np.isfinite(e) or e > 1e-4:
            print(f"{tag} {k}: {e:.3e}   <-- OUTSIDE", flush=True)
            bad += 1
    if M >= 1000:
        y = r.integers(0, 2, size=M)
        s = checks["cosine"][1] + 0.3 * y
        got = scoring.eer_from_scores(y, s)
        want = O.eer(y, s)
        if abs(got[0] - want[0]) > 1e-9 or abs(got[1] - want[1]) > 1e-6:
            print(f"{tag} eer: {got} vs {want}   <-- OUTSIDE", flush=True)
            bad += 1
print(f"worst absolute score error {worst:.3e}; {bad} outside")
sys.exit(1 if bad else 0)

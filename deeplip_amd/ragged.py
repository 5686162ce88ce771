"""Length-bucketed batching of variable-length utterances / clips (host logic only; no arithmetic).

The reference extracts its test lists one utterance at a time, each at its own length (train_fusion.py:334-349: audio
``[1,24,T_i]``, every clip ``[1,1,T_j,88,88]``).  The engine wants batches, and a batch of unequal lengths is the zero-padded
tensor + length vector of ``pad_packed_collate`` (models/video_models/dataset.py:123-139) -- whose padding frames are still
computed by the convolutions (the kernels only leave them out of the pooled statistics).  So the list is SORTED by length and cut
into batches whose padded length comes from a short geometric ladder: every batch of a rung pads to the rung's top, a rung spans
lengths within ``1 + waste`` of each other, and the number of distinct padded shapes -- each one a recorded step plan
(deeplip_amd/plan.py) -- stays at log(Tmax / Tmin) / log(1 + waste): a dozen for 137 .. 412-frame utterances at 10 %.
"""
from __future__ import annotations

from typing import List, NamedTuple, Sequence

import numpy as np


class Batch(NamedTuple):
    idx: np.ndarray      # indices into the caller's list, at most ``batch`` of them, longest last
    T: int               # the padded length of this batch (a rung top: many batches share it)


def rung_tops(tmin: int, tmax: int, waste: float = 0.10, quantum: int = 1) -> List[int]:
    """The ladder of padded lengths: tops[k+1] <= (1 + waste) * tops[k] (rounded up to ``quantum`` frames), the last = tmax
    rounded up.  Built downwards from tmax so that the longest -- and most expensive -- items pad least."""
    if tmin < 1 or tmax < tmin or waste <= 0 or quantum < 1:
        raise ValueError("rung_tops: need 1 <= tmin <= tmax, waste > 0, quantum >= 1")
    q = lambda t: -(-int(t) // quantum) * quantum
    tops = [q(tmax)]
    while True:
        lo = int(np.floor(tops[-1] / (1.0 + waste)))    # the shortest item that still pads to tops[-1] within `waste`
        nxt = q(lo)
        if nxt >= tops[-1]:                             # the quantum is coarser than the waste bound: step one quantum down
            nxt = tops[-1] - quantum
        if nxt < tmin or nxt < 1:
            break
        tops.append(nxt)
    return tops[::-1]


def plan_batches(lengths: Sequence[int], batch: int, waste: float = 0.10, quantum: int = 1) -> List[Batch]:
    """Cut a list of item lengths into batches of at most ``batch`` items, ascending in length, each padded to a rung top.

    Every item goes to the lowest rung whose top holds it; a rung's items are emitted ``batch`` at a time and its remainder
    (< batch items) moves UP into the next rung instead of making a short batch -- only the very last batch of the whole list
    may be short.  Padding overhead: sum(B_i * T_i) / sum(lengths) - 1 <= waste for the items that stay on their rung (the few
    carried ones may pad up to two rungs)."""
    L = np.asarray(lengths, dtype=np.int64)
    if L.ndim != 1 or L.size == 0:
        return []
    if L.min() < 1 or batch < 1:
        raise ValueError("plan_batches: lengths must be >= 1, batch >= 1")
    order = np.argsort(L, kind="stable")
    tops = rung_tops(int(L.min()), int(L.max()), waste, quantum)
    out: List[Batch] = []
    carry = np.empty((0,), dtype=np.int64)
    pos = 0
    for k, top in enumerate(tops):
        end = int(np.searchsorted(L[order], top, side="right"))
        items = np.concatenate([carry, order[pos:end]])
        pos = end
        last = k == len(tops) - 1
        n_full = items.size // batch
        for b in range(n_full):
            out.append(Batch(items[b * batch:(b + 1) * batch], int(top)))
        carry = items[n_full * batch:]
        if last and carry.size:
            out.append(Batch(carry, int(top)))
            carry = carry[:0]
    return out


def padding_overhead(lengths: Sequence[int], batches: Sequence[Batch], batch: int) -> float:
    """Computed frames / valid frames - 1, counting a short batch at its full ``batch`` rows (what a recorded plan runs)."""
    L = np.asarray(lengths, dtype=np.int64)
    computed = sum(batch * b.T for b in batches)
    return computed / float(L.sum()) - 1.0


def pad_stack(items: Sequence[np.ndarray], T: int, axis: int, rows: int = None) -> np.ndarray:
    """Zero-pad each item to ``T`` along ``axis`` and stack them (``rows`` >= len(items): trailing all-zero rows) -- what
    pad_packed_collate builds (dataset.py:130-134), for any item rank."""
    first = items[0]
    shape = list(first.shape)
    shape[axis] = T
    out = np.zeros([rows or len(items)] + shape, dtype=first.dtype)
    for i, it in enumerate(items):
        sl = [i] + [slice(None)] * it.ndim
        sl[1 + axis] = slice(0, it.shape[axis])
        out[tuple(sl)] = it
    return out

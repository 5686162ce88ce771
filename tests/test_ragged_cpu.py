"""Host logic of the ragged extraction path (no GPU): the length ladder and batch plan (deeplip_amd/ragged.py), the zero-padded
collate (models/video_models/dataset.py:123-139), the ragged synthetic set, the draw of per-clip crop / flip parameters
(models/video_models/preprocess.py:95-138)."""
import random

import numpy as np
import pytest

from deeplip_amd import ragged
from deeplip_amd.synthetic import SyntheticAVSet


def test_rung_tops_bound_the_padding():
    for tmin, tmax, waste, q in ((137, 412, 0.10, 4), (11, 75, 0.10, 1), (11, 75, 0.05, 1), (29, 29, 0.1, 1), (1, 7, 0.5, 4)):
        tops = ragged.rung_tops(tmin, tmax, waste, q)
        assert tops == sorted(set(tops)) and tops[-1] >= tmax and all(t % q == 0 for t in tops)
        for lo, hi in zip(tops[:-1], tops[1:]):
            # an item just above `lo` pads to `hi`: within the waste bound, or one quantum where the quantum is the coarser step
            assert hi <= (lo + 1) * (1 + waste) + 1e-9 or hi - lo <= q
    assert len(ragged.rung_tops(137, 412, 0.10, 4)) <= 14          # a dozen recorded plans, not one per length


def test_plan_batches_partitions_sorts_and_carries():
    r = np.random.default_rng(0)
    L = r.integers(137, 413, size=5000)
    bs = ragged.plan_batches(L, 64, 0.10, quantum=4)
    allidx = np.concatenate([b.idx for b in bs])
    assert sorted(allidx.tolist()) == list(range(5000))                      # every item exactly once
    assert all(len(b.idx) == 64 for b in bs[:-1]) and 1 <= len(bs[-1].idx) <= 64  # only the very last batch may be short
    assert all(int(L[b.idx].max()) <= b.T for b in bs)                        # every item fits its batch
    assert [b.T for b in bs] == sorted(b.T for b in bs)                       # rungs ascend: each plan's batches are consecutive
    assert ragged.padding_overhead(L, bs, 64) < 0.10
    assert len({b.T for b in bs}) <= 14
    # one length only: one rung, no padding
    bs1 = ragged.plan_batches([29] * 130, 64)
    assert [(b.T, len(b.idx)) for b in bs1] == [(29, 64), (29, 64), (29, 2)]
    assert ragged.plan_batches([], 8) == []
    with pytest.raises(ValueError):
        ragged.plan_batches([3, 0, 5], 2)


def test_pad_stack_is_pad_packed_collate():
    """The reference's collate (dataset.py:130-134): np.zeros((n, max_len, h, w)); data_np[idx][:len] = item."""
    items = [np.full((t, 2, 3), float(t), dtype=np.float32) for t in (5, 2, 4)]
    got = ragged.pad_stack(items, 5, axis=0)
    want = np.zeros((3, 5, 2, 3), dtype=np.float32)
    for i, it in enumerate(items):
        want[i][:it.shape[0]] = it
    assert np.array_equal(got, want)
    got = ragged.pad_stack([np.ones((4, t), dtype=np.float32) for t in (3, 6)], 8, axis=1, rows=3)
    assert got.shape == (3, 4, 8) and got[0, :, :3].all() and not got[0, :, 3:].any() and not got[2].any()


def test_ragged_synthetic_set_and_prefix_free_ids():
    ds = SyntheticAVSet(4, 13, 3, audio_dim=24, key="t.ragged", ragged=True)
    n = len(ds)
    assert ds.audio_len.shape == (n,) and 137 <= ds.audio_len.min() and ds.audio_len.max() <= 412
    assert ds.clip_ptr[0] == 0 and ds.clip_ptr[-1] == len(ds.clip_len) and set(np.diff(ds.clip_ptr)) <= {1, 2, 3}
    assert 11 <= ds.clip_len.min() and ds.clip_len.max() <= 75 and len(set(ds.audio_len.tolist())) > 10
    assert ds.audio_item(3).shape == (24, ds.audio_len[3]) and ds.clip_item(5).shape == (ds.clip_len[5], 88, 88)
    x, L = ds.audio_padded([0, 1, 2], rows=4)
    assert x.shape == (4, 24, L.max()) and np.array_equal(x[1, :, :L[1]], ds.audio_item(1)) and not x[1, :, L[1]:].any() and not x[3].any()
    v, Lv = ds.clips_padded([2, 0])
    assert v.shape == (2, 1, Lv.max(), 88, 88) and np.array_equal(v[1, 0, :Lv[1]], ds.clip_item(0))
    again = SyntheticAVSet(4, 13, 3, audio_dim=24, key="t.ragged", ragged=True)
    assert np.array_equal(again.audio_len, ds.audio_len) and np.array_equal(again.clip_len, ds.clip_len)   # seeded
    with pytest.raises(ValueError):
        ds.audio([0])
    # >= 11 utterances per speaker: the reference's readers glob `<pattern>*` for an utterance's clip files
    # (models/fusion_models/utils.py:456-463); no stored name may be a prefix of another ("s3_u1*" must not match "s3_u10_c0.npz")
    from deeplip_amd import scoring_entry as se
    for kind in ("spk/utt", "utt"):
        pats = [se._pattern(kind, u) for u in ds.utt_ids]
        assert len(set(pats)) == n
        assert not any(a != b and b.startswith(a) for a in pats for b in pats)


def test_draw_clip_params_follows_the_references_draw_order():
    """RandomCrop draws delta_w THEN delta_h with randint's inclusive bounds (preprocess.py:110-111), HorizontalFlip then flips iff
    random.random() < ratio (:134): the same generator state gives the same crops and flips as the reference's pipeline would."""
    from deeplip_amd.ops import draw_clip_params
    a = draw_clip_params(6, 96, 100, crop=88, rng=random.Random(5))
    r = random.Random(5)
    for row in a:
        ox = r.randint(0, 100 - 88)
        oy = r.randint(0, 96 - 88)
        flip = int(r.random() < 0.5)
        assert row.tolist() == [oy, ox, flip, 0]
    assert a.dtype == np.int32 and a.shape == (6, 4)
    z = draw_clip_params(3, 88, 88, rng=random.Random(1))
    assert (z[:, :2] == 0).all()

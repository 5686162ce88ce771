import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: a long CPU-oracle leg (tens of seconds); runs by default")


@pytest.fixture(scope="session", autouse=True)
def _arith_baseline():
    """The LIBRARY's default arithmetic is "auto" (f16x3 with an f32 re-run of what leaves its range: deeplip_amd/arith.py).  The
    tests choose their modes explicitly -- packing.set_precision / arith.configure / $DLIP_ARITH per test or module -- and start from
    the exact mode, as they always have: f32 packs, no fallback."""
    os.environ.pop("DLIP_ARITH", None)
    from deeplip_amd import arith, autograd_video as av
    arith.configure("f32")
    av.TRAIN_CONV = "f16x3"      # ... and the train-mode convolutions where they have always run in the tests: the split kernels
    yield


@pytest.fixture(autouse=True)
def _arith_isolated():
    """What a test (a Trainer, arith.configure) sets must not leak into the next one."""
    from deeplip_amd import arith, autograd_video as av, packing
    saved = (packing.PRECISION, arith.MODE, av.TRAIN_CONV, dict(arith.STATS))
    yield
    packing.set_precision(saved[0])
    arith.MODE, av.TRAIN_CONV = saved[1], saved[2]
    arith.STATS.update(saved[3])


@pytest.fixture(params=["auto", "f32"])
def arith_mode(request, monkeypatch):
    """Entry-point tests run once per arithmetic mode: $DLIP_ARITH is what a Trainer's arith.configure() reads first."""
    monkeypatch.setenv("DLIP_ARITH", request.param)
    return request.param


@pytest.fixture(scope="session")
def manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden():
    return {n: np.load(os.path.join(GOLDEN, f"{n}_golden.npz")) for n in ("video", "audio", "heads", "train", "audio_train", "audio_attn_train", "video_train")}


def rel_err(a, b):
    """max|a-b| / max|b| -- the parity measure used throughout (tolerance stated per test)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def assert_close_rel(a, b, rtol=1e-4, afloor=1e-6, what=""):
    """The north star's bar, element by element: |a - b| <= rtol * |b| + afloor * max|b| for EVERY element
    ("within 1e-4 relative"; the absolute floor, one millionth of the tensor's largest magnitude, only keeps
    elements that are themselves ~0 from being held to a relative bar they cannot have).  Stricter than rel_err,
    which compares the largest error with the largest magnitude."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    bound = rtol * np.abs(b) + afloor * max(float(np.abs(b).max()), 1e-30)
    bad = np.abs(a - b) > bound
    if bad.any():
        i = np.unravel_index(int(np.argmax(np.abs(a - b) - bound)), a.shape)
        raise AssertionError(f"{what}: {int(bad.sum())} of {a.size} elements outside |a-b| <= {rtol}|b| + {afloor} max|b|; worst at {i}: "
                             f"got {a[i]!r}, want {b[i]!r}")

"""Audio front-end fuzz: MFCC / fbank / logfbank (+ deltas) on waveforms of RANDOM length -- from shorter than one window up to 4 s, every
remainder against the 160-sample shift -- and random batch sizes, against the oracle (python_speech_features' arithmetic in fp64).
Un-normalised features are compared outright; CMVN-ed ones on the bands that are not constant (tests/test_frontend.py says why).
Elements whose un-normalised LOG energy is below -20 (2e-9 of a unit-energy frame: the lowest one-bin filter next to DC in a frame where
pre-emphasis leaves nothing) are counted separately: they sit below what fp32 framing resolves -- the reference frames and transforms in
fp64 -- and come out 0.3 ... 1 % off in energy (1e-4 ... 3e-4 of the feature scale, band 0, one frame in a few hundred).
   python tools/probes/frontend_fuzz.py [n] [seed] [fft64 | gemm32 | direct64]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import torch

from deeplip_amd import weightgen as wg
from deeplip_amd.frontend import AudioFrontend
from oracle import deeplip_oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dft = sys.argv[3] if len(sys.argv) > 3 else None        # fft64 (the default) | gemm32 | direct64
r = np.random.Generator(np.random.PCG64(seed))
bad, worst = 0, 0.0
floor_elems, floor_worst = 0, 0.0


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


for i in range(n):
    feat_type, num_bin = [("mfcc", 26), ("logfbank", 60), ("fbank", 24), ("logfbank", 80), ("mfcc", 40)][i % 5]
    delta = bool(r.integers(0, 2))
    B = int(r.integers(1, 6))
    S = [int(r.integers(1, 400)), 400, 401, 559, 560, 561][i] if i < 6 else int(r.integers(400, 64000))
    t = np.arange(S) / 16000.0
    sig = np.stack([0.3 * np.sin(2 * np.pi * (150 + 170 * b) * t) + 0.05 * r.standard_normal(S) for b in range(B)]).astype(np.float32)
    tag = f"{feat_type}-{num_bin} delta={delta} B={B} S={S}"
    try:
        for normalize in (False, True):
            fe = AudioFrontend(feat_type, num_bin=num_bin, normalize=normalize, delta=delta, dft=dft)
            y = fe(torch.from_numpy(sig).cuda()).cpu().numpy()
            for b in range(B):
                ref = O.audio_features(sig[b].astype(np.float64), feat_type, nfilt=num_bin, normalize=normalize, delta=delta)
                if y[b].shape != ref.shape:
                    print(f"{tag} normalize={normalize}: shape {y[b].shape} vs {ref.shape}   <-- OUTSIDE", flush=True)
                    bad += 1
                    break
                raw = O.audio_features(sig[b].astype(np.float64), feat_type, nfilt=num_bin, normalize=False, delta=delta)
                live = raw.std(axis=1) > (1e-4 * np.abs(raw).max() if normalize else 1e-6)
                if ref.shape[1] < 3 and normalize:
                    continue                       # CMVN over one or two frames: 0 / 0 in the reference
                if live.sum() == 0:
                    continue
                floor = (raw < -20.0) if feat_type != "fbank" else (raw < 2e-9)
                if feat_type != "fbank" or not normalize:
                    nfloor = int((floor & live[:, None]).sum())
                    floor_elems += nfloor
                    if nfloor:
                        floor_worst = max(floor_worst, float((np.abs(y[b] - ref) * (floor & live[:, None])).max() / np.abs(ref).max()))
                keep = live[:, None] & ~floor
                e = float((np.abs(y[b] - ref) * keep).max() / max(np.abs(ref * keep).max(), 1e-30))
                worst = max(worst, e)
                if not np.isfinite(e) or e > 1e-4:
                    d = np.abs(y[b] - ref) * keep
                    k, f = np.unravel_index(int(np.argmax(d)), d.shape)
                    print(f"{tag} normalize={normalize} row {b}: {e:.3e} (frames {ref.shape[1]}); worst at band {k} frame {f}: got {y[b][k, f]:.6f} want "
                          f"{ref[k, f]:.6f} raw {raw[k, f]:.6f}; max|ref| {np.abs(ref).max():.3f}; frames over 1e-4: {int((d > 1e-4 * np.abs(ref).max()).any(0).sum())}, "
                          f"bands: {sorted(set(np.nonzero((d > 1e-4 * np.abs(ref).max()).any(1))[0].tolist()))}   <-- OUTSIDE", flush=True)
                    bad += 1
    except Exception as ex:
        print(f"{tag}: {type(ex).__name__}: {str(ex)[:160]}   <-- RAISED", flush=True)
        bad += 1
print(f"worst {worst:.3e}; {bad} outside; {floor_elems} elements at the framing floor, worst of them {floor_worst:.3e}")
sys.exit(1 if bad else 0)

#!/usr/bin/env python3
"""Gated experiment (Next #5): Winograd F(2x2, 3x3) for the stride-1 3x3 layers, fp32 transforms + split-fp16 products.

Runs on the CPU (numpy): numerics of the scheme against an fp64 direct convolution on the trunk's layer shapes, next to the same
emulation of the shipped direct split-fp16 path -- and the arithmetic that decides it: what the 16 transformed GEMMs look like
(reduction depth, bytes) beside the measured efficiency of this engine's kernels at those depths.

    python tools/winograd_study.py            # prints a table; profiles/r3/winograd_study.txt is its output
"""
import numpy as np

G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float64)
Bt = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
At = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)


def split(x32):
    """fp32 -> hi + lo fp16 (what the split activation / weight format stores), returned as fp64 of the two halves' sum terms."""
    hi = x32.astype(np.float16)
    lo = (x32 - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float64), lo.astype(np.float64)


def prod3(a32, b32, axes):
    """sum over `axes` of a*b as hi*hi + hi*lo + lo*hi with fp32 accumulation (emulated: products exact, sum rounded to fp32 once
    per 32-term slice, as the MFMA chain does)."""
    ah, al = split(a32); bh, bl = split(b32)
    return (np.tensordot(ah, bh, axes) + np.tensordot(ah, bl, axes) + np.tensordot(al, bh, axes)).astype(np.float32)


def scale_rows(w):      # per-output-channel power of two, as packing.split_weights
    amax = np.abs(w).reshape(w.shape[0], -1).max(1)
    s = 2.0 ** np.floor(np.log2(1023.0 / np.maximum(amax, 1e-30)))
    return s


def direct_ref(x, w):   # x [H,W,C] fp64, w [K,3,3,C] fp64, pad 1 -> [H,W,K]
    H, W, C = x.shape
    xp = np.zeros((H + 2, W + 2, C)); xp[1:-1, 1:-1] = x
    y = np.zeros((H, W, w.shape[0]))
    for r in range(3):
        for s in range(3):
            y += xp[r:r + H, s:s + W] @ w[:, r, s].T
    return y


def direct_f16x3(x32, w64):
    H, W, C = x32.shape
    K = w64.shape[0]
    sc = scale_rows(w64)
    ws = (w64 * sc[:, None, None, None]).astype(np.float32)
    xp = np.zeros((H + 2, W + 2, C), np.float32); xp[1:-1, 1:-1] = x32
    acc = np.zeros((H, W, K), np.float32)
    for r in range(3):
        for s in range(3):
            acc = (acc + prod3(xp[r:r + H, s:s + W], ws[:, r, s], ([2], [1]))).astype(np.float32)
    return (acc / sc.astype(np.float32)).astype(np.float32)


def winograd_f16x3(x32, w64):
    H, W, C = x32.shape
    K = w64.shape[0]
    Hp, Wp = (H + 1) // 2 * 2, (W + 1) // 2 * 2
    xp = np.zeros((Hp + 2, Wp + 2, C), np.float32); xp[1:H + 1, 1:W + 1] = x32
    U64 = np.einsum("ar,krsc,bs->kabc", G, w64, G)                  # [K,4,4,C] filter transform in fp64 (once per checkpoint)
    amax = np.abs(U64).transpose(1, 2, 0, 3).reshape(16, K, -1).max(2)   # per (frequency, output channel) scale
    sc = 2.0 ** np.floor(np.log2(1023.0 / np.maximum(amax, 1e-30)))     # [16,K]
    y = np.zeros((Hp, Wp, K), np.float32)
    th, tw = Hp // 2, Wp // 2
    d = np.stack([np.stack([xp[2 * i:2 * i + 4, 2 * j:2 * j + 4] for j in range(tw)]) for i in range(th)])   # [th,tw,4,4,C]
    Bt32 = Bt.astype(np.float32)
    V = np.einsum("ar,ijrsc,bs->ijabc", Bt32, d, Bt32).astype(np.float32)   # input transform in fp32 (+/- only: exact up to rounding)
    M = np.zeros((th, tw, 4, 4, K), np.float32)
    for a in range(4):
        for b in range(4):
            Us = (U64[:, a, b] * sc[a * 4 + b][:, None]).astype(np.float32)
            M[:, :, a, b] = prod3(V[:, :, a, b], Us, ([2], [1])) / sc[a * 4 + b].astype(np.float32)
    At32 = At.astype(np.float32)
    Y = np.einsum("pa,ijabk,qb->ijpqk", At32, M, At32).astype(np.float32)   # output transform in fp32
    for i in range(th):
        for j in range(tw):
            y[2 * i:2 * i + 2, 2 * j:2 * j + 2] = Y[i, j]
    return y[:H, :W]


def main():
    r = np.random.Generator(np.random.PCG64(7))
    print("numerics vs an fp64 direct convolution: max|err| / max|ref|, and elements outside the north star's bar |err| <= 1e-4 |ref| + 1e-6 max|ref|:")
    for name, H, C, K in (("layer1 22x22 C64", 22, 64, 64), ("layer2 11x11 C128", 11, 128, 128), ("layer3 6x6 C256", 6, 256, 256)):
        x = r.standard_normal((H, H, C)); x = np.maximum(x, 0.2 * x).astype(np.float32)        # PReLU-like activations
        w = r.standard_normal((K, 3, 3, C)) / np.sqrt(9 * C)
        ref = direct_ref(x.astype(np.float64), w)
        for tag, fn in (("direct f16x3 (shipped)", direct_f16x3), ("Winograd F(2x2,3x3) f16x3", winograd_f16x3)):
            y = fn(x, w).astype(np.float64)
            e = np.abs(y - ref)
            bad = int((e > 1e-4 * np.abs(ref) + 1e-6 * np.abs(ref).max()).sum())
            print(f"  {name:20s} {tag:28s} rel {e.max() / np.abs(ref).max():.2e}   outside the bar: {bad} of {ref.size}")
    print()
    print("what the transformed problem looks like (B = 64 clips, 29 frames):")
    for name, H, C, K, cur_us in (("layer1 (4 convs)", 22, 64, 64, 220.0), ("layer2 (3 convs)", 11, 128, 128, 180.0), ("layer3 (3 convs)", 6, 256, 256, 150.0)):
        N = 64 * 29
        tiles = N * ((H + 1) // 2) ** 2
        act_mb = N * H * H * C * 4 / 1e6
        v_mb = tiles * 16 * C * 4 / 1e6
        direct_gf = 2.0 * N * H * H * K * 9 * C / 1e9
        wino_gf = 2.0 * tiles * 16 * K * C / 1e9
        print(f"  {name:18s} direct {direct_gf:7.1f} GFLOP in ~{cur_us:.0f} us measured | Winograd: 16 GEMMs [{tiles} x {C}] x [{C} x {K}], reduction depth "
              f"{C // 32} slices of 32, {wino_gf:6.1f} GFLOP ({direct_gf / wino_gf:.2f}x fewer); transformed input V = {v_mb:6.0f} MB "
              f"(activation {act_mb:.0f} MB): written + read once = {2 * v_mb / 5.0:.0f} us at 5 TB/s if not fused")
    print()
    print("measured efficiency of this engine's split-fp16 GEMM kernels by reduction depth (profiles/r3): >= 32 slices 0.40-0.44 of the 833 TFLOP/s\n"
          "ceiling; 16 slices (k = 1 TDNN layers) 0.24; the Winograd GEMMs have 2 (layer 1), 4 (layer 2) and 8 (layer 3) slices.")


if __name__ == "__main__":
    main()

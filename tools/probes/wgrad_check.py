"""Weight gradients at the training step's real shapes (B = 32 x 29 = 928 images) against an fp64 statement (development probe)."""
import sys
sys.path.insert(0, ".")
import torch, torch.nn.functional as F
from deeplip_amd import _lib, autograd_video as av
N = int(sys.argv[1]) if len(sys.argv) > 1 else 928
for name, H, C, K, R, s, p in [("layer4 3x3", 3, 512, 512, 3, 1, 1), ("layer4.0 s2", 6, 256, 512, 3, 2, 1), ("layer3 3x3", 6, 256, 256, 3, 1, 1),
                               ("layer2 3x3", 11, 128, 128, 3, 1, 1), ("layer1 3x3", 22, 64, 64, 3, 1, 1)]:
    Ho = (H + 2 * p - (R - 1) - 1) // s + 1
    g = torch.Generator().manual_seed(3)
    x = torch.randn(N, H, H, C, generator=g)
    dy = torch.randn(N, Ho, Ho, K, generator=g) * 1e-3
    # fp64 reference: dW[k,c,r,s] = sum x[n, h*s + r - p, w*s + q - p, c] dy[n,h,w,k]
    xp = F.pad(x.double().permute(0, 3, 1, 2), (p, p, p, p))
    ref = torch.zeros(K, C, R, R, dtype=torch.float64)
    for r in range(R):
        for q in range(R):
            patch = xp[:, :, r:r + (Ho - 1) * s + 1:s, q:q + (Ho - 1) * s + 1:s]          # [N,C,Ho,Ho]
            ref[:, :, r, q] = torch.einsum("nchw,nhwk->kc", patch, dy.double())
    for mode in ("conv", "gemm"):
        fn = av.wgrad_as_conv if mode == "conv" else (lambda *a, **k: av._permute3(av.wgrad_conv_fused(*a, **k), (2, 1, 0)).view(K, C, R, R))
        out = fn(x.cuda(), dy.cuda(), R, R, (s, s), (p, p), (1, 1))
        torch.cuda.synchronize()
        e = float((out.cpu().double() - ref).abs().max() / ref.abs().max())
        print(f"{name:14s} N={N} {mode}: rel err {e:.3e}", flush=True)

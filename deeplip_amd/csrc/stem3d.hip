// Video stem: Conv3d(1 -> 64, 5x7x7, stride (1,2,2), pad (2,3,3)) + folded BN + PReLU/ReLU,
// written directly as NDHWC = [(B*T), Ho, Wo, 64] so the reference's threeD_to_2D_tensor copy
// (models/video_models/model.py:9-13) disappears.  Replaces model.py:81-84.
//
// C_in = 1, so the reduction is the 245 filter taps (padded to 248).  One workgroup owns 4 output
// rows of one frame (4 x 44 = 176 pixels = eleven 16-pixel MFMA row tiles, no remainder):
//   * the 5-frame x 13-row x (W+6)-column input window (zero halo included) is staged once in
//     LDS (24 KB) with coalesced row reads;
//   * each of the 4 waves owns 16 output channels and keeps its 248 x 16 filter slice in 62
//     VGPRs for the whole tile (weights packed k-major [248][64]: 64-B coalesced reads);
//   * A fragments are gathered straight from the LDS window with ds_read_b32 (tap offset walked in
//     registers + a per-lane pixel offset; stride-2 pixels x 4 tap quarters = mostly conflict
//     free), v_mfma_f32_16x16x4_f32 accumulates exact fp32, 11 independent accumulators per wave.
#include "dlip_common.h"

namespace {

constexpr int KT = 5, KH = 7, KW = 7, KTAPS = KT * KH * KW;  // 245
constexpr int KPAD = 248, KSTEPS = KPAD / 4;                  // 62
constexpr int ROWS = 4;                                       // output rows per workgroup
constexpr int PR = 2 * ROWS + 5;                              // 13 input rows

struct StemArgs {
  const float* x;
  const float* w;  // [248][64]
  const float* bias;
  const float* slope;
  float* y;
  int T, H, W, Ho, Wo;
  int row_tiles;
  int pwp;  // LDS row pitch (W + 6)
};

template <int MT>
__global__ __launch_bounds__(256) void stem3d_f32_kernel(const StemArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int plane = PR * a.pwp;
  float* patch = smem;                                         // [5][13][pwp]

  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int rt = swz % a.row_tiles;
  const int f = swz / a.row_tiles;       // frame index b*T + t
  const int t = f % a.T;
  const int ho0 = rt * ROWS;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int n = wave * 16 + li;

  {
    // Stage the 5 x 13 x (W+6) window: each wave takes whole window rows (wave-uniform row decode,
    // lanes stride the columns -> coalesced 4-B reads, zero halo written explicitly).
    const int hi0 = 2 * ho0 - 3;
    const float* xf = a.x + (size_t)(f - t) * a.H * a.W;  // clip base
    const int w_id = __builtin_amdgcn_readfirstlane(tid >> 6), ln = tid & 63;
    for (int row = w_id; row < KT * PR; row += 4) {
      const int ft = row / PR, pr = row - ft * PR;
      const int tt = t + ft - 2, hi = hi0 + pr;
      const bool rok = (unsigned)tt < (unsigned)a.T && (unsigned)hi < (unsigned)a.H;
      const float* src = xf + ((size_t)(rok ? tt : 0) * a.H + (rok ? hi : 0)) * a.W;
      float* dst = patch + row * a.pwp;
      for (int pc = ln; pc < a.pwp; pc += 64) {
        const int wi = pc - 3;
        dst[pc] = (rok && (unsigned)wi < (unsigned)a.W) ? src[wi] : 0.f;
      }
    }
  }

  // The wave's 248 x 16 filter slice lives in VGPRs (hoisting all 62 loads above the staging with
  // a scheduling fence costs a wave of occupancy and measured 11 % slower: left to the compiler).
  float breg[KSTEPS];
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) breg[ks] = a.w[(4 * ks + kq) * 64 + n];

  int pixoff[MT];
  const int npix = ROWS * a.Wo;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    int p = mt * 16 + li;
    if (p >= npix) p = 0;
    const int orow = p / a.Wo, ocol = p - orow * a.Wo;
    pixoff[mt] = 2 * orow * a.pwp + 2 * ocol;
  }

  f32x4 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};

  __syncthreads();

  // Tap walk kept in registers: lane quarter kq owns taps k = 4*ks + kq; (kt,kh,kw) advance by 4
  // taps per step with carry, so the LDS offset of the tap needs no table (and no lgkmcnt drain).
  int kw = kq, kh = 0;              // kq < 7
  int ko = kw;                       // ktp*plane + kh*pwp + kw
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) {
    const int kk = (4 * ks + kq) < KTAPS ? ko : 0;   // padded taps read offset 0 (their weights are 0)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const float av = patch[kk + pixoff[mt]];
      acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, breg[ks], acc[mt], 0, 0, 0);
    }
    kw += 4; ko += 4;
    if (kw >= KW) {
      kw -= KW; ko += a.pwp - KW;
      if (++kh == KH) { kh = 0; ko += plane - KH * a.pwp; }
    }
  }

  // C/D map of the 16x16 MFMA: column (channel) = lane & 15, row (pixel) = (lane >> 4)*4 + e.
  const float bias = a.bias ? a.bias[n] : 0.f;
  const float slope = a.slope ? a.slope[n] : 1.f;
  float* yf = a.y + (size_t)f * a.Ho * a.Wo * 64;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int p = mt * 16 + kq * 4 + e;
      const int orow = p / a.Wo, ocol = p - orow * a.Wo;
      if (p < npix && ho0 + orow < a.Ho) {
        float v = acc[mt][e] + bias;
        if (a.slope) v = v >= 0.f ? v : v * slope;
        yf[((size_t)(ho0 + orow) * a.Wo + ocol) * 64 + n] = v;
      }
    }
  }
}

}  // namespace

extern "C" int dlip_stem3d_bn_act_f32(const float* x, const float* w_248xk, const float* bias,
                                      const float* slope, float* y, int32_t B, int32_t T, int32_t H,
                                      int32_t W, int32_t K, dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && w_248xk && y && B > 0 && T > 0 && H > 0 && W > 0);
  DLIP_CHECK_ARG(K == 64 && (H & 1) == 0 && (W & 1) == 0);
  StemArgs a;
  a.x = x; a.w = w_248xk; a.bias = bias; a.slope = slope; a.y = y;
  a.T = T; a.H = H; a.W = W; a.Ho = H / 2; a.Wo = W / 2;
  a.row_tiles = (a.Ho + ROWS - 1) / ROWS;
  a.pwp = W + 6;
  const long long grid = (long long)B * T * a.row_tiles;
  if (grid > 0x7FFFFFFFll) return DLIP_ERANGE;
  const size_t lds = (size_t)(KT * PR * a.pwp) * 4;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int mt = (ROWS * a.Wo + 15) / 16;
  if (mt <= 11) {
    hipLaunchKernelGGL(stem3d_f32_kernel<11>, dim3((unsigned)grid), dim3(256), lds, st, a);
  } else if (mt <= 12) {
    hipLaunchKernelGGL(stem3d_f32_kernel<12>, dim3((unsigned)grid), dim3(256), lds, st, a);
  } else {
    return DLIP_EINVAL;  // frames wider than 96 pixels are outside the reference's crop sizes
  }
  return dlip_launch_status();
}

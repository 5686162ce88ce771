// Implicit-GEMM convolution for NHWC fp32 activations on the gfx950 fp32 matrix core
// (v_mfma_f32_32x32x2_f32: exact fp32 fma chain, 256 FLOP/clk/CU).
//
// GEMM view:  Y[M, K] = A[M, R*S*C] * Wt[K, R*S*C]^T,  M = N*Ho*Wo rows (output pixels),
// A gathered on the fly from x (im2col never materialised), Wt = KRSC-packed weights.
// One workgroup (4 waves) owns a BM x BN output tile and walks the reduction in BK=32-channel
// slices of one filter tap at a time:
//   global (buffer_load_dwordx4, halo / tails -> 0 via the buffer range check)
//     -> registers (prefetch of slice t+1 is in flight while slice t is multiplied)
//     -> LDS [row][36]  (32 floats + 4 pad: conflict-free ds_read_b128 / ds_write_b128)
//     -> one ds_read_b128 per operand row feeds FOUR mfma 32x32x2 steps: lanes 0-31 hold
//        k = 8q..8q+3, lanes 32-63 hold k = 8q+4..8q+7 of the same row, so step j multiplies
//        k = 8q+j (lower half-wave) and k = 8q+4+j (upper half) -- a permutation of the
//        reduction index that A and W share, hence exact.
// Epilogue (fused, per output element): + bias[k] (folded BN), + residual, PReLU/LeakyReLU/ReLU
// by per-channel slope, optional post affine; stores are 2 x 128-B segments per wave instruction.
//
// Replaces the torch.nn layers listed against dlip_conv_nhwc_f32 in include/deeplip_hip.h.
#include "dlip_common.h"

namespace {

struct ConvArgs {
  const float* x;
  const float* w;
  const float* bias;
  const float* res;
  const float* slope;
  const float* pscale;
  const float* pshift;
  float* y;
  int H, W, C, K, R, S;
  int sh, sw, ph, pw, dh, dw;
  int Wo, HoWo;
  int ldx, ldy, ldr;
  int M;
  int tiles_n;
  int cchunks;  // ceil(C / 32)
  int nk;       // R * S * cchunks
  int rsc;      // R * S * C  (weight row length)
  uint32_t x_bytes, w_bytes;
};

constexpr int BK = 32;
constexpr int LDK = BK + 4;

template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void conv_igemm_f32_kernel(const ConvArgs a) {
  static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int MI = WM / 32, NI = WN / 32;
  constexpr int A_PER = BM / 32, B_PER = BN / 32;
  static_assert(MI >= 1 && NI >= 1, "wave tile must hold at least one 32x32 MFMA tile");

  __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LDK];
  float* As = smem;
  float* Bs = smem + BM * LDK;

  // XCD-aware tile order: workgroups b and b+8 share an XCD (round-robin dispatch), so give each
  // XCD a contiguous run of tiles (bijective for any grid size); tile_n is innermost so the
  // tiles that re-read the same activation rows and the same filter slices meet in one L2.
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tile_n = swz % a.tiles_n;
  const int tile_m = swz / a.tiles_n;

  const int tid = threadIdx.x;
  const int cc = (tid & 7) * 4;   // channel offset of this thread's float4 inside a BK slice
  const int rbase = tid >> 3;     // 0..31

  const __amdgpu_buffer_rsrc_t xr = dlip_make_rsrc(a.x, a.x_bytes);
  const __amdgpu_buffer_rsrc_t wr = dlip_make_rsrc(a.w, a.w_bytes);

  // Per-thread im2col row descriptors.
  int a_pix[A_PER], a_hi0[A_PER], a_wi0[A_PER];
#pragma unroll
  for (int j = 0; j < A_PER; ++j) {
    const int m = tile_m * BM + rbase + 32 * j;
    if (m < a.M) {
      const int n = m / a.HoWo;
      const int rem = m - n * a.HoWo;
      const int ho = rem / a.Wo;
      const int wo = rem - ho * a.Wo;
      a_pix[j] = n * a.H * a.W;
      a_hi0[j] = ho * a.sh - a.ph;
      a_wi0[j] = wo * a.sw - a.pw;
    } else {
      a_pix[j] = 0;
      a_hi0[j] = -0x40000000;  // fails every bounds check -> zeros
      a_wi0[j] = 0;
    }
  }
  int b_row[B_PER];
#pragma unroll
  for (int j = 0; j < B_PER; ++j) {
    const int n = tile_n * BN + rbase + 32 * j;
    b_row[j] = n < a.K ? n * a.rsc : -1;
  }

  f32x4 ra[A_PER], rb[B_PER];
  auto load_slice = [&](int r, int s, int c0) {
    const bool cok = (c0 + cc) < a.C;
    const int dh = r * a.dh, dw = s * a.dw;
#pragma unroll
    for (int j = 0; j < A_PER; ++j) {
      const int hi = a_hi0[j] + dh, wi = a_wi0[j] + dw;
      const bool ok = cok && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
      const uint32_t off = ok ? (uint32_t)(((a_pix[j] + hi * a.W + wi) * a.ldx + c0 + cc) * 4) : DLIP_OOB_OFFSET;
      ra[j] = dlip_buffer_load_f4(xr, off);
    }
    const int kw = (r * a.S + s) * a.C + c0 + cc;
#pragma unroll
    for (int j = 0; j < B_PER; ++j) {
      const bool ok = cok && b_row[j] >= 0;
      const uint32_t off = ok ? (uint32_t)((b_row[j] + kw) * 4) : DLIP_OOB_OFFSET;
      rb[j] = dlip_buffer_load_f4(wr, off);
    }
  };
  auto store_slice = [&]() {
#pragma unroll
    for (int j = 0; j < A_PER; ++j) *reinterpret_cast<f32x4*>(&As[(rbase + 32 * j) * LDK + cc]) = ra[j];
#pragma unroll
    for (int j = 0; j < B_PER; ++j) *reinterpret_cast<f32x4*>(&Bs[(rbase + 32 * j) * LDK + cc]) = rb[j];
  };

  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int lrow = lane & 31, khalf = (lane >> 5) * 4;
  const float* Aw = As + (wm * WM + lrow) * LDK + khalf;
  const float* Bw = Bs + (wn * WN + lrow) * LDK + khalf;

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  int r = 0, s = 0, c0 = 0;
  load_slice(r, s, c0);
  store_slice();
  __syncthreads();

  for (int kt = 0; kt < a.nk; ++kt) {
    const bool more = (kt + 1) < a.nk;
    if (more) {
      c0 += BK;
      if (c0 >= a.C) {
        c0 = 0;
        if (++s == a.S) { s = 0; ++r; }
      }
      load_slice(r, s, c0);  // in flight during the MFMAs below
    }
#pragma unroll
    for (int q = 0; q < BK / 8; ++q) {
      f32x4 af[MI], bf[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) af[mi] = *reinterpret_cast<const f32x4*>(Aw + mi * 32 * LDK + q * 8);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) bf[ni] = *reinterpret_cast<const f32x4*>(Bw + ni * 32 * LDK + q * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi][j], bf[ni][j], acc[mi][ni], 0, 0, 0);
    }
    __syncthreads();
    if (more) {
      store_slice();
      __syncthreads();
    }
  }

  // Epilogue.  C/D map of the 32x32 MFMA: column = lane & 31, row = (e & 3) + 8*(e >> 2) + 4*(lane >> 5).
  const int rquad = (lane >> 5) * 4;
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int k = tile_n * BN + wn * WN + ni * 32 + lrow;
    const bool kok = k < a.K;
    const float bias = (kok && a.bias) ? a.bias[k] : 0.f;
    const float slope = (kok && a.slope) ? a.slope[k] : 1.f;
    const float psc = (kok && a.pscale) ? a.pscale[k] : 1.f;
    const float psh = (kok && a.pshift) ? a.pshift[k] : 0.f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int m0 = tile_m * BM + wm * WM + mi * 32 + rquad;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + (e & 3) + 8 * (e >> 2);
        if (kok && m < a.M) {
          float v = acc[mi][ni][e] + bias;
          if (a.res) v += a.res[(size_t)m * a.ldr + k];
          if (a.slope) v = v >= 0.f ? v : v * slope;
          if (a.pscale) v = v * psc + psh;
          a.y[(size_t)m * a.ldy + k] = v;
        }
      }
    }
  }
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
int launch(const ConvArgs& a, hipStream_t st) {
  ConvArgs b = a;
  const int tiles_m = (a.M + BM - 1) / BM;
  b.tiles_n = (a.K + BN - 1) / BN;
  const long long grid = (long long)tiles_m * b.tiles_n;
  if (grid <= 0 || grid > 0x7FFFFFFFll) return DLIP_EINVAL;
  hipLaunchKernelGGL((conv_igemm_f32_kernel<BM, BN, WAVES_M, WAVES_N>), dim3((unsigned)grid), dim3(256), 0, st, b);
  return dlip_launch_status();
}

// Tile choice: 128x128 when there are enough output channels and rows to fill the chip with it,
// 128x64 for the 64-channel stage, 64x64 for short-M GEMMs (linear layers, tiny batches).
int pick_tile(long long M, int K) {
  const long long t128 = ((M + 127) / 128) * ((K + 127) / 128);
  if (K > 64 && t128 >= 512) return 0;
  const long long t12864 = ((M + 127) / 128) * ((K + 63) / 64);
  if (t12864 >= 512) return 1;
  return 2;
}

}  // namespace

extern "C" int dlip_conv_nhwc_f32(const dlip_conv_desc* d, const float* x, const float* w_krsc,
                                  const float* bias, const float* residual, const float* slope,
                                  const float* post_scale, const float* post_shift, float* y,
                                  dlip_stream_t stream) {
  DLIP_CHECK_ARG(d && x && w_krsc && y);
  DLIP_CHECK_ARG(d->N > 0 && d->H > 0 && d->W > 0 && d->C > 0 && d->K > 0 && d->R > 0 && d->S > 0);
  DLIP_CHECK_ARG(d->stride_h > 0 && d->stride_w > 0 && d->dil_h > 0 && d->dil_w > 0 && d->pad_h >= 0 && d->pad_w >= 0);
  DLIP_CHECK_ARG((d->C & 3) == 0 && (d->ldx & 3) == 0 && d->ldx >= d->C && d->ldy >= d->K);
  DLIP_CHECK_ARG((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(w_krsc) & 15) == 0);
  DLIP_CHECK_ARG((post_scale == nullptr) == (post_shift == nullptr));
  DLIP_CHECK_ARG(residual == nullptr || d->ldr >= d->K);
  const int Ho = (d->H + 2 * d->pad_h - d->dil_h * (d->R - 1) - 1) / d->stride_h + 1;
  const int Wo = (d->W + 2 * d->pad_w - d->dil_w * (d->S - 1) - 1) / d->stride_w + 1;
  DLIP_CHECK_ARG(Ho == d->Ho && Wo == d->Wo && Ho > 0 && Wo > 0);

  const long long in_pix = (long long)d->N * d->H * d->W;
  const long long x_bytes = ((in_pix - 1) * d->ldx + d->C) * 4;
  const long long w_bytes = (long long)d->K * d->R * d->S * d->C * 4;
  const long long M = (long long)d->N * Ho * Wo;
  if (x_bytes > DLIP_MAX_BUFFER_BYTES || w_bytes > DLIP_MAX_BUFFER_BYTES || M > 0x7FFFFFFFll) return DLIP_ERANGE;

  ConvArgs a;
  a.x = x; a.w = w_krsc; a.bias = bias; a.res = residual; a.slope = slope;
  a.pscale = post_scale; a.pshift = post_shift; a.y = y;
  a.H = d->H; a.W = d->W; a.C = d->C; a.K = d->K; a.R = d->R; a.S = d->S;
  a.sh = d->stride_h; a.sw = d->stride_w; a.ph = d->pad_h; a.pw = d->pad_w; a.dh = d->dil_h; a.dw = d->dil_w;
  a.Wo = Wo; a.HoWo = Ho * Wo;
  a.ldx = d->ldx; a.ldy = d->ldy; a.ldr = d->ldr;
  a.M = (int)M;
  a.tiles_n = 0;
  a.cchunks = (d->C + BK - 1) / BK;
  a.nk = d->R * d->S * a.cchunks;
  a.rsc = d->R * d->S * d->C;
  a.x_bytes = (uint32_t)x_bytes; a.w_bytes = (uint32_t)w_bytes;

  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (pick_tile(M, d->K)) {
    case 0: return launch<128, 128, 2, 2>(a, st);
    case 1: return launch<128, 64, 2, 2>(a, st);
    default: return launch<64, 64, 2, 2>(a, st);
  }
}

extern "C" int dlip_conv_plan(const dlip_conv_desc* d, int32_t* bm, int32_t* bn) {
  DLIP_CHECK_ARG(d && bm && bn && d->N > 0 && d->Ho > 0 && d->Wo > 0 && d->K > 0);
  static const int tiles[3][2] = {{128, 128}, {128, 64}, {64, 64}};
  const int t = pick_tile((long long)d->N * d->Ho * d->Wo, d->K);
  *bm = tiles[t][0];
  *bn = tiles[t][1];
  return DLIP_OK;
}

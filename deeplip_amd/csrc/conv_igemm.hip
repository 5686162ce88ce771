// Implicit-GEMM convolution for NHWC fp32 activations on the gfx950 fp32 matrix core
// (v_mfma_f32_32x32x2_f32: exact fp32 fma chain, 256 FLOP/clk/CU).
//
// GEMM view:  Y[M, K] = A[M, R*S*C] * Wt[K, R*S*C]^T,  M = N*Ho*Wo rows (output pixels),
// A gathered on the fly from x (im2col never materialised), Wt = KRSC-packed weights.
// One workgroup (4 waves) owns a BM x BN output tile and walks the reduction in BK=32-channel
// slices of one filter tap at a time:
//   global (buffer_load_dwordx4, halo / tails -> 0 via the buffer range check)
//     -> registers (prefetch of slice t+1 is in flight while slice t is multiplied)
//     -> LDS [2 stages][row][36]  (32 floats + 4 pad: conflict-free ds_read_b128 / ds_write_b128)
//     -> one ds_read_b128 per operand row feeds FOUR mfma 32x32x2 steps: lanes 0-31 hold
//        k = 8q..8q+3, lanes 32-63 hold k = 8q+4..8q+7 of the same row, so step j multiplies
//        k = 8q+j (lower half-wave) and k = 8q+4+j (upper half) -- a permutation of the
//        reduction index that A and W share, hence exact.
// Fusion: the accumulators are initialised with bias[k] (folded BN) + residual[m,k]; the epilogue
// applies the per-channel PReLU/LeakyReLU/ReLU slope and the optional post affine and stores
// 2 x 128-B segments per wave instruction, branch-free (tails dropped by the buffer range check).
//
// Replaces the torch.nn layers listed against dlip_conv_nhwc_f32 in include/deeplip_hip.h.
#include "conv_common.h"

namespace {

constexpr int LDK = BK;  // unpadded 128-B rows; bank conflicts are removed by an XOR swizzle of the 16-B chunks

// ------------------------------------------------------------------------------------------
// Software pipeline: two LDS stages, ONE barrier per 32-channel slice, and every
// non-MFMA instruction of the slice (fragment ds_reads of the next k8 step, the global loads of
// the next slice, their ds_writes, the barrier) placed in program order BETWEEN small groups of
// MFMAs, so that an in-order wave issues them in the 64-cycle shadow of an executing MFMA instead
// of in front of the matrix pipe.  Slice t, k8 step q (MI*NI MFMAs per j):
//   q=0: [j0] read frags(q=1) [j1] load A rows of slice t+2 [j2] load A rows [j3] load W rows
//   q=1: [j0] read frags(q=2) [j1..j3]
//   q=2: [j0] read frags(q=3) [j1] ds_write A(t+1) -> other stage [j2] ds_write W(t+1) [j3]
// (global loads run TWO slices ahead through two staging register sets: >= 1.5 slices of MFMA
//  time to land, instead of 0.5)
//   q=3: [j0] barrier; read frags(q=0 of slice t+1) [j1..j3]
// ------------------------------------------------------------------------------------------
template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void conv_igemm_f32_kernel(const ConvArgs a) {
  static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int MI = WM / 32, NI = WN / 32;
  constexpr int A_PER = BM / 32, B_PER = BN / 32;
  constexpr int STAGE = (BM + BN) * LDK;
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tile_n = swz % a.tiles_n;
  const int tile_m = swz / a.tiles_n;

  const int tid = threadIdx.x;
  const int cc = (tid & 7) * 4;
  const int rbase = tid >> 3;
  const __amdgpu_buffer_rsrc_t xr = dlip_make_rsrc(a.x, a.x_bytes);
  const __amdgpu_buffer_rsrc_t wr = dlip_make_rsrc(a.w, a.w_bytes);

  int a_off[A_PER];
  uint32_t a_mask[A_PER];
#pragma unroll
  for (int j = 0; j < A_PER; ++j) {
    const int m = tile_m * BM + rbase + 32 * j;
    a_off[j] = 0;
    a_mask[j] = 0u;
    if (m < a.M) {
      const int n = m / a.HoWo;
      const int rem = m - n * a.HoWo;
      const int ho = rem / a.Wo;
      const int wo = rem - ho * a.Wo;
      const int hi0 = ho * a.sh - a.ph, wi0 = wo * a.sw - a.pw;
      a_off[j] = (((n * a.H + hi0) * a.W + wi0) * a.ldx + cc) * 4;
      uint32_t mk = 0u;
      for (int r = 0; r < a.R; ++r)
        for (int s = 0; s < a.S; ++s)
          if ((unsigned)(hi0 + r * a.dh) < (unsigned)a.H && (unsigned)(wi0 + s * a.dw) < (unsigned)a.W)
            mk |= 1u << (r * a.S + s);
      a_mask[j] = mk;
    }
  }
  int b_off[B_PER];
#pragma unroll
  for (int j = 0; j < B_PER; ++j) {
    const int n = tile_n * BN + rbase + 32 * j;
    b_off[j] = n < a.K ? (n * a.rsc + cc) * 4 : -1;
  }

  f32x4 ra[2][A_PER], rb[2][B_PER];  // two staging sets: slice t+2 is in flight while slice t+1 waits for its ds_write
  int tap = 0, x_tap = 0, w_tap = 0, c0 = 0;   // wave-uniform position of the slice being LOADED
  auto load_a = [&](auto SETC, int j0, int j1) {
    constexpr int SET = decltype(SETC)::value;
    const bool cok = (c0 + cc) < a.C;
#pragma unroll
    for (int j = j0; j < j1; ++j) {
      const bool ok = cok && ((a_mask[j] >> tap) & 1u);
      ra[SET][j] = dlip_buffer_load_f4(xr, ok ? (uint32_t)(a_off[j] + x_tap) : DLIP_OOB_OFFSET);
    }
  };
  auto load_b = [&](auto SETC) {
    constexpr int SET = decltype(SETC)::value;
    const bool cok = (c0 + cc) < a.C;
#pragma unroll
    for (int j = 0; j < B_PER; ++j) {
      const bool ok = cok && b_off[j] >= 0;
      rb[SET][j] = dlip_buffer_load_f4(wr, ok ? (uint32_t)(b_off[j] + w_tap) : DLIP_OOB_OFFSET);
    }
  };
  // LDS image: row-major [row][32 floats], the eight 16-B chunks of a row permuted by
  // chunk ^ ((row >> 1) & 7).  Rows r and r+1 sit in opposite halves of the 64 banks, and the key
  // walks all 8 chunk positions over 16 consecutive rows, so both the ds_write_b128 (8 lanes = one
  // row) and the ds_read_b128 (16 lanes = 16 rows, one logical chunk) are conflict-free.
  const int st_off = rbase * LDK + ((((tid & 7) ^ ((rbase >> 1) & 7))) << 2);
  auto store_a = [&](auto SETC, int stage) {
    constexpr int SET = decltype(SETC)::value;
    float* As = smem + stage * STAGE + st_off;
#pragma unroll
    for (int j = 0; j < A_PER; ++j) *reinterpret_cast<f32x4*>(&As[32 * j * LDK]) = ra[SET][j];
  };
  auto store_b = [&](auto SETC, int stage) {
    constexpr int SET = decltype(SETC)::value;
    float* Bs = smem + stage * STAGE + BM * LDK + st_off;
#pragma unroll
    for (int j = 0; j < B_PER; ++j) *reinterpret_cast<f32x4*>(&Bs[32 * j * LDK]) = rb[SET][j];
  };

  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int lrow = lane & 31;
  const int a_frag = (wm * WM + lrow) * LDK;
  const int b_frag = BM * LDK + (wn * WN + lrow) * LDK;
  const int rquad = (lane >> 5) * 4;
  int kq[BK / 8];  // swizzled float offset of this lane's 16-B chunk for k8 step q
#pragma unroll
  for (int q = 0; q < BK / 8; ++q) kq[q] = (((2 * q + (lane >> 5)) ^ ((lrow >> 1) & 7))) << 2;

  // accumulators = bias (+ residual)
  const __amdgpu_buffer_rsrc_t rr = dlip_make_rsrc(a.res, a.res ? a.r_bytes : 0u);
  f32x16 acc[MI][NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int k = tile_n * BN + wn * WN + ni * 32 + lrow;
    const bool kok = k < a.K;
    const float bias = (kok && a.bias) ? a.bias[k] : 0.f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int m0 = tile_m * BM + wm * WM + mi * 32 + rquad;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + (e & 3) + 8 * (e >> 2);
        const uint32_t off = (kok && m < a.M) ? (uint32_t)((m * a.ldr + k) * 4) : DLIP_OOB_OFFSET;
        // rr has zero records when there is no residual: the load returns 0 without a branch
        acc[mi][ni][e] = bias + __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, (int)off, 0, 0));
      }
    }
  }

  // slice-walk bookkeeping (scalar)
  int s_pos = 0, x_row = 0;
  const int x_dr = a.dh * a.W * a.ldx * 4, x_ds = a.dw * a.ldx * 4;
  // Reduction walk: 32-channel slice OUTER, filter tap INNER.  Consecutive slices then read the same
  // channel slice of neighbouring pixels (tap shifts the window by one pixel / one image row), so the
  // activation rows of a tile are re-served from the 32 KiB L1 instead of L2; the weight stream is
  // the same bytes in a different order.
  const int ntaps = a.R * a.S;
  auto advance = [&]() {
    ++tap;
    if (++s_pos == a.S) { s_pos = 0; x_row += x_dr; }
    if (tap == ntaps) { tap = 0; s_pos = 0; x_row = 0; c0 += BK; }
    x_tap = x_row + s_pos * x_ds + c0 * 4;
    w_tap = (tap * a.C + c0) * 4;
  };

  f32x4 fa[2][MI], fb[2][NI];
  auto read_frags = [&](int set, int stage, int q) {
    const float* Aw = smem + stage * STAGE + a_frag + kq[q];
    const float* Bw = smem + stage * STAGE + b_frag + kq[q];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) fa[set][mi] = *reinterpret_cast<const f32x4*>(Aw + mi * 32 * LDK);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) fb[set][ni] = *reinterpret_cast<const f32x4*>(Bw + ni * 32 * LDK);
  };
  auto mfma_j = [&](int set, int j) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][mi][j], fb[set][ni][j], acc[mi][ni], 0, 0, 0);
  };
#define DLIP_FENCE() __builtin_amdgcn_sched_barrier(0)

  using Set0 = std::integral_constant<int, 0>;
  using Set1 = std::integral_constant<int, 1>;
  // Prologue: slice 0 -> set 0 -> LDS stage 0; slice 1 -> set 1 (stays in flight).
  load_a(Set0{}, 0, A_PER);
  load_b(Set0{});
  if (a.nk > 1) {
    advance();
    load_a(Set1{}, 0, A_PER);
    load_b(Set1{});
  }
  store_a(Set0{}, 0);
  store_b(Set0{}, 0);
  __syncthreads();
  read_frags(0, 0, 0);

  // One slice.  LD = staging set that receives slice kt+2 (free: it held slice kt, already in LDS);
  // ST = the other set, holding slice kt+1, written to LDS stage (kt+1)&1 during q=2.
  auto slice = [&](auto LD, auto ST, int kt) {
    const bool more1 = (kt + 1) < a.nk, more2 = (kt + 2) < a.nk;
    const int cur = kt & 1;
    // ---- q = 0 ----
    mfma_j(0, 0); DLIP_FENCE();
    read_frags(1, cur, 1); DLIP_FENCE();
    mfma_j(0, 1); DLIP_FENCE();
    if (more2) { advance(); load_a(LD, 0, A_PER / 2); } DLIP_FENCE();
    mfma_j(0, 2); DLIP_FENCE();
    if (more2) load_a(LD, A_PER / 2, A_PER); DLIP_FENCE();
    mfma_j(0, 3); DLIP_FENCE();
    if (more2) load_b(LD); DLIP_FENCE();
    // ---- q = 1 ----
    mfma_j(1, 0); DLIP_FENCE();
    read_frags(0, cur, 2); DLIP_FENCE();
    mfma_j(1, 1); mfma_j(1, 2); mfma_j(1, 3); DLIP_FENCE();
    // ---- q = 2 ----
    mfma_j(0, 0); DLIP_FENCE();
    read_frags(1, cur, 3); DLIP_FENCE();
    mfma_j(0, 1); DLIP_FENCE();
    if (more1) store_a(ST, cur ^ 1); DLIP_FENCE();
    mfma_j(0, 2); DLIP_FENCE();
    if (more1) store_b(ST, cur ^ 1); DLIP_FENCE();
    mfma_j(0, 3); DLIP_FENCE();
    // ---- q = 3 ----
    mfma_j(1, 0); DLIP_FENCE();
    __syncthreads();
    if (more1) read_frags(0, cur ^ 1, 0); DLIP_FENCE();
    mfma_j(1, 1); mfma_j(1, 2); mfma_j(1, 3); DLIP_FENCE();
  };
  for (int kt = 0; kt < a.nk; kt += 2) {
    slice(Set0{}, Set1{}, kt);
    if (kt + 1 < a.nk) slice(Set1{}, Set0{}, kt + 1);
  }
#undef DLIP_FENCE

  const __amdgpu_buffer_rsrc_t yr = dlip_make_rsrc(a.y, a.y_bytes);
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int k = tile_n * BN + wn * WN + ni * 32 + lrow;
    const bool kok = k < a.K;
    const float slope = (kok && a.slope) ? a.slope[k] : 1.f;
    const float psc = (kok && a.pscale) ? a.pscale[k] : 1.f;
    const float psh = (kok && a.pshift) ? a.pshift[k] : 0.f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int m0 = tile_m * BM + wm * WM + mi * 32 + rquad;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + (e & 3) + 8 * (e >> 2);
        float v = acc[mi][ni][e];
        v = v >= 0.f ? v : v * slope;
        v = v * psc + psh;
        const uint32_t off = (kok && m < a.M) ? (uint32_t)((m * a.ldy + k) * 4) : DLIP_OOB_OFFSET;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), yr, (int)off, 0, 0);
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
  constexpr size_t lds = 2 * (size_t)(BM + BN) * LDK * sizeof(float);
  auto kern = conv_igemm_f32_kernel<BM, BN, WAVES_M, WAVES_N>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, st, b);
  return dlip_launch_status();
}

// Tile menu.  {BM, BN, WAVES_M, WAVES_N}: the 2x2 wave layouts give each wave a (BM/2)x(BN/2)
// block; the 1x4 layouts give each wave all BM rows of a 32-channel column, which makes BM any
// multiple of 32 -- used to fight tile quantisation: a launch takes ceil(tiles / 256 CUs) "rounds"
// of one tile per CU, so for the short-M layers (layer3/4, TDNN) BM is chosen so the tile count
// lands just under a multiple of 256.
int launch_cfg(int cfg, const ConvArgs& a, hipStream_t st) {
  switch (cfg) {
    case 0: return launch<128, 128, 2, 2>(a, st);
    case 1: return launch<128, 64, 2, 2>(a, st);
    case 2: return launch<64, 64, 2, 2>(a, st);
    case 3: return launch<64, 128, 1, 4>(a, st);
    default: return launch<96, 128, 1, 4>(a, st);
  }
}

}  // namespace

extern "C" int dlip_conv_nhwc_f32(const dlip_conv_desc* d, const float* x, const float* w_krsc,
                                  const float* bias, const float* residual, const float* slope,
                                  const float* post_scale, const float* post_shift, float* y,
                                  dlip_stream_t stream) {
  ConvArgs a;
  const int rc = dlip_fill_conv_args(d, x, w_krsc, bias, residual, slope, post_scale, post_shift, y, d ? d->C : 0, &a);
  if (rc != DLIP_OK) return rc;
  const long long M = a.M;
  hipStream_t st = static_cast<hipStream_t>(stream);
  return launch_cfg(pick_tile(M, d->K), a, st);
}

extern "C" void dlip_conv_dma_tile(long long M, int K, int nk, int epi, int* bm, int* bn);   // conv_igemm_f16x3_dma.hip
extern "C" int dlip_conv_dma_enabled(void);                                  // conv_igemm_f16x3.hip

extern "C" int dlip_conv_rows_plan(const dlip_conv_desc* d, int* bm);   // conv_rows_f16x3.hip
extern "C" int dlip_conv_rows2d_plan(const dlip_conv_desc* d, int c2, int* bm);
extern "C" int dlip_conv_plan(const dlip_conv_desc* d, int32_t split_f16, int32_t* bm, int32_t* bn) {
  DLIP_CHECK_ARG(d && bm && bn && d->N > 0 && d->Ho > 0 && d->Wo > 0 && d->K > 0);
  if ((split_f16 & 3) == 3 && dlip_conv_dma_enabled()) {   // split-format activations: the LDS-DMA kernel's menu
    int rbm = 0;
    if (dlip_conv_rows_plan(d, &rbm)) { *bm = rbm; *bn = 256; return DLIP_OK; }   // conv_rows_f16x3_kernel<BM / 32, ..>: BM x 256
    if (dlip_conv_rows2d_plan(d, 0, &rbm)) { *bm = rbm; *bn = 256; return DLIP_OK; }   // its general mode (2-D filters, residual, second source)
    dlip_conv_dma_tile((long long)d->N * d->Ho * d->Wo, d->K, d->R * d->S * ((d->C + 31) / 32), 0, bm, bn);
    return DLIP_OK;
  }
  const int t = pick_tile((long long)d->N * d->Ho * d->Wo, d->K, split_f16 ? kEffF16x3 : kEffF32);
  *bm = kCfg[t].bm;
  *bn = kCfg[t].bn;
  return DLIP_OK;
}

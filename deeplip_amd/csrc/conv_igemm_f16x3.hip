// Split-precision implicit-GEMM convolution: fp32 operands carried as (hi, lo) fp16 pairs and
// multiplied on the 16x faster f16 matrix core with THREE v_mfma_f32_32x32x16_f16 per product
// (hi*hi + hi*lo + lo*hi, fp32 accumulate).  hi has 11 significant bits and lo the next 11, so
// x ~ hi + lo to ~2^-22, each fp16 x fp16 product is exact in the fp32 accumulator, and only the
// lo*lo term (~2^-22 relative) is dropped: fp32-grade results (measured parity in tests /
// bench) at up to 5.3x the fp32-MFMA rate.  Same GEMM view, tiling, swizzled two-stage LDS,
// two-slice register prefetch and MFMA-shadow program order as conv_igemm.hip; differences:
//   * weights are pre-split on the host (deeplip_amd/packing.py): per 32-channel slice of a tap the
//     packed row holds [hi k0..31 | lo k0..31] fp16 (= 128 B, same bytes as fp32), scaled per output
//     channel by a power of two so lo stays a NORMAL fp16 (the scale is undone exactly in the
//     epilogue);
//   * activations stay fp32 in HBM; each thread splits its float4 of a slice row in registers
//     (cvt / sub / cvt, packed) and writes hi and lo with two ds_write_b64 into the same
//     XOR-swizzled 128-B LDS row: chunks 0-3 = hi(k 0-7, 8-15, 16-23, 24-31), chunks 4-7 = lo;
//   * a k16 step reads 4 fragments (a_hi, a_lo, b_hi, b_lo; lanes 0-31 take k 0-7, lanes 32-63 take
//     k 8-15 of the step) and issues 3 MFMAs per 32x32 tile.
// Replaces the same reference layers as dlip_conv_nhwc_f32 (include/deeplip_hip.h).
#include "conv_common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int LDK = BK;  // 32 dwords = 128 B per LDS row: 64 B of hi + 64 B of lo

// float4 -> 4 hi halves, 4 lo halves
__device__ __forceinline__ void split4(const f32x4 v, f16x4& hi, f16x4& lo) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const _Float16 h = (_Float16)v[i];
    hi[i] = h;
    lo[i] = (_Float16)(v[i] - (float)h);
  }
}

// ASPLIT: activations x (and the residual) are already stored as (hi, lo) fp16 pairs -- per pixel and
// 32-channel block: 32 hi halves then 32 lo halves, the same 128 bytes an fp32 block occupies -- so the
// A operand is a plain 16-B copy like the weights (no conversion, one ds_write_b128).  OSPLIT: the
// epilogue writes that format, i.e. each activation is split ONCE by its producer instead of once
// per filter tap and output-column tile by its consumers.
template <int BM, int BN, int WAVES_M, int WAVES_N, bool ASPLIT, bool OSPLIT>
__global__ __launch_bounds__(256, 2) void conv_igemm_f16x3_kernel(const ConvArgs a) {
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
  const int cq = tid & 7;          // which float4 (4 channels) of the 32-channel slice row
  const int cc = cq * 4;
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
    b_off[j] = n < a.K ? (n * a.rsc + cc) * 4 : -1;   // packed weight rows: 16-B chunk cq of the slice
  }

  f32x4 ra[2][A_PER], rb[2][B_PER];
  int tap = 0, x_tap = 0, w_tap = 0, c0 = 0;
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
#pragma unroll
    for (int j = 0; j < B_PER; ++j)
      rb[SET][j] = dlip_buffer_load_f4(wr, b_off[j] >= 0 ? (uint32_t)(b_off[j] + w_tap) : DLIP_OOB_OFFSET);
  };

  // LDS row = 8 chunks of 16 B, chunk p stored at p ^ ((row >> 1) & 7) (conv_igemm.hip).
  const int key_st = (rbase >> 1) & 7;
  // A: this thread's 4 channels go to hi chunk cq/2 and lo chunk 4 + cq/2, 8-B half cq & 1.
  const int a_hi_off = rbase * LDK + ((((cq >> 1)) ^ key_st) << 2) + ((cq & 1) << 1);
  const int a_lo_off = rbase * LDK + ((((cq >> 1) + 4) ^ key_st) << 2) + ((cq & 1) << 1);
  const int b_st_off = rbase * LDK + ((cq ^ key_st) << 2);   // B rows arrive already in chunk order
  auto store_a = [&](auto SETC, int stage) {
    constexpr int SET = decltype(SETC)::value;
    float* As = smem + stage * STAGE;
#pragma unroll
    for (int j = 0; j < A_PER; ++j) {
      if constexpr (ASPLIT) {
        *reinterpret_cast<f32x4*>(&As[b_st_off + 32 * j * LDK]) = ra[SET][j];   // chunk cq -> position cq ^ key
      } else {
        f16x4 hi, lo;
        split4(ra[SET][j], hi, lo);
        *reinterpret_cast<f16x4*>(&As[a_hi_off + 32 * j * LDK]) = hi;
        *reinterpret_cast<f16x4*>(&As[a_lo_off + 32 * j * LDK]) = lo;
      }
    }
  };
  auto store_b = [&](auto SETC, int stage) {
    constexpr int SET = decltype(SETC)::value;
    float* Bs = smem + stage * STAGE + BM * LDK + b_st_off;
#pragma unroll
    for (int j = 0; j < B_PER; ++j) *reinterpret_cast<f32x4*>(&Bs[32 * j * LDK]) = rb[SET][j];
  };

  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int lrow = lane & 31, half = lane >> 5;
  const int a_frag = (wm * WM + lrow) * LDK;
  const int b_frag = BM * LDK + (wn * WN + lrow) * LDK;
  const int rquad = half * 4;
  const int key_rd = (lrow >> 1) & 7;
  int khi[2], klo[2];   // swizzled dword offsets of this lane's hi / lo chunk for k16 step s
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    khi[s] = ((2 * s + half) ^ key_rd) << 2;
    klo[s] = ((4 + 2 * s + half) ^ key_rd) << 2;
  }

  // accumulators = (bias + residual) * wscale[k]   (the weight scale is undone in the epilogue)
  const __amdgpu_buffer_rsrc_t rr = dlip_make_rsrc(a.res, a.res ? a.r_bytes : 0u);
  f32x16 acc[MI][NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int k = tile_n * BN + wn * WN + ni * 32 + lrow;
    const bool kok = k < a.K;
    const float bias = (kok && a.bias) ? a.bias[k] : 0.f;
    const float ws = kok ? a.wscale[k] : 1.f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int m0 = tile_m * BM + wm * WM + mi * 32 + rquad;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + (e & 3) + 8 * (e >> 2);
        float r;
        if constexpr (ASPLIT) {   // residual in split format: hi at block*128 + 2*(k%32), lo 64 bytes further
          const uint32_t off = (kok && m < a.M) ? (uint32_t)((m * a.ldr + (k & ~31)) * 4 + (k & 31) * 2) : DLIP_OOB_OFFSET;
          const _Float16 rh = __builtin_bit_cast(_Float16, __builtin_amdgcn_raw_buffer_load_b16(rr, (int)off, 0, 0));
          const _Float16 rl = __builtin_bit_cast(_Float16, __builtin_amdgcn_raw_buffer_load_b16(rr, (int)off + 64, 0, 0));
          r = (float)rh + (float)rl;
        } else {
          const uint32_t off = (kok && m < a.M) ? (uint32_t)((m * a.ldr + k) * 4) : DLIP_OOB_OFFSET;
          r = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, (int)off, 0, 0));
        }
        acc[mi][ni][e] = (bias + r) * ws;
      }
    }
  }

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
    w_tap = (tap * a.Cw + c0) * 4;
  };

  f16x8 fah[2][MI], fal[2][MI], fbh[2][NI], fbl[2][NI];
  auto read_frags = [&](int set, int stage, int s) {
    const float* Aw = smem + stage * STAGE + a_frag;
    const float* Bw = smem + stage * STAGE + b_frag;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      fah[set][mi] = *reinterpret_cast<const f16x8*>(Aw + mi * 32 * LDK + khi[s]);
      fal[set][mi] = *reinterpret_cast<const f16x8*>(Aw + mi * 32 * LDK + klo[s]);
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      fbh[set][ni] = *reinterpret_cast<const f16x8*>(Bw + ni * 32 * LDK + khi[s]);
      fbl[set][ni] = *reinterpret_cast<const f16x8*>(Bw + ni * 32 * LDK + klo[s]);
    }
  };
  // g = 0: lo*hi, 1: hi*lo, 2: hi*hi  (small terms first)
  auto mfma_g = [&](int set, int g) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const f16x8 av = g == 0 ? fal[set][mi] : fah[set][mi];
        const f16x8 bv = g == 1 ? fbl[set][ni] : fbh[set][ni];
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc[mi][ni], 0, 0, 0);
      }
  };
#define DLIP_FENCE() __builtin_amdgcn_sched_barrier(0)

  using Set0 = std::integral_constant<int, 0>;
  using Set1 = std::integral_constant<int, 1>;
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

  // One 32-channel slice = two k16 steps x three MFMA groups.  Non-MFMA work sits between groups.
  auto slice = [&](auto LD, auto ST, int kt) {
    const bool more1 = (kt + 1) < a.nk, more2 = (kt + 2) < a.nk;
    const int cur = kt & 1;
    // ---- k16 step 0 (fragment set 0) ----
    mfma_g(0, 0); DLIP_FENCE();
    read_frags(1, cur, 1); DLIP_FENCE();
    mfma_g(0, 1); DLIP_FENCE();
    if (more2) { advance(); load_a(LD, 0, A_PER); } DLIP_FENCE();
    mfma_g(0, 2); DLIP_FENCE();
    if (more2) load_b(LD); DLIP_FENCE();
    // ---- k16 step 1 (fragment set 1) ----
    mfma_g(1, 0); DLIP_FENCE();
    if (more1) store_a(ST, cur ^ 1); DLIP_FENCE();
    mfma_g(1, 1); DLIP_FENCE();
    if (more1) store_b(ST, cur ^ 1);
    __syncthreads();
    if (more1) read_frags(0, cur ^ 1, 0); DLIP_FENCE();
    mfma_g(1, 2); DLIP_FENCE();
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
    const float inv = kok ? 1.f / a.wscale[k] : 1.f;   // power of two: exact
    const float slope = (kok && a.slope) ? a.slope[k] : 1.f;
    const float psc = (kok && a.pscale) ? a.pscale[k] : 1.f;
    const float psh = (kok && a.pshift) ? a.pshift[k] : 0.f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int m0 = tile_m * BM + wm * WM + mi * 32 + rquad;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + (e & 3) + 8 * (e >> 2);
        float v = acc[mi][ni][e] * inv;
        v = v >= 0.f ? v : v * slope;
        v = v * psc + psh;
        if constexpr (OSPLIT) {
          const uint32_t off = (kok && m < a.M) ? (uint32_t)((m * a.ldy + (k & ~31)) * 4 + (k & 31) * 2) : DLIP_OOB_OFFSET;
          const _Float16 h = (_Float16)v;
          const _Float16 l = (_Float16)(v - (float)h);
          __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, h), yr, (int)off, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, l), yr, (int)off + 64, 0, 0);
          if ((e & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // keeps the epilogue's live set (and the kernel's VGPR count) small
        } else {
          const uint32_t off = (kok && m < a.M) ? (uint32_t)((m * a.ldy + k) * 4) : DLIP_OOB_OFFSET;
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), yr, (int)off, 0, 0);
        }
      }
    }
  }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool ASPLIT, bool OSPLIT>
int launch_fmt(const ConvArgs& a, hipStream_t st) {
  ConvArgs b = a;
  const int tiles_m = (a.M + BM - 1) / BM;
  b.tiles_n = (a.K + BN - 1) / BN;
  const long long grid = (long long)tiles_m * b.tiles_n;
  if (grid <= 0 || grid > 0x7FFFFFFFll) return DLIP_EINVAL;
  constexpr size_t lds = 2 * (size_t)(BM + BN) * LDK * sizeof(float);
  auto kern = conv_igemm_f16x3_kernel<BM, BN, WAVES_M, WAVES_N, ASPLIT, OSPLIT>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, st, b);
  return dlip_launch_status();
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
int launch(const ConvArgs& a, hipStream_t st, int flags) {
  switch (flags & 3) {
    case 0: return launch_fmt<BM, BN, WAVES_M, WAVES_N, false, false>(a, st);
    case 1: return launch_fmt<BM, BN, WAVES_M, WAVES_N, true, false>(a, st);
    case 2: return launch_fmt<BM, BN, WAVES_M, WAVES_N, false, true>(a, st);
    default: return launch_fmt<BM, BN, WAVES_M, WAVES_N, true, true>(a, st);
  }
}

}  // namespace

extern "C" int dlip_conv_f16x3_dma_launch(const void* args, void* stream, int epi);   // conv_igemm_f16x3_dma.hip
extern "C" void dlip_conv_dma_tile(long long M, int K, int nk, int epi, int* bm, int* bn);
extern "C" int dlip_conv_win_ok(const void* args);                                          // conv_win_f16x3.hip
extern "C" int dlip_conv_f16x3_win_launch(const void* args, void* stream, int out_split);
extern "C" int dlip_conv_rows_ok(const void* args);                                         // conv_rows_f16x3.hip
extern "C" int dlip_conv_f16x3_rows_launch(const void* args, void* stream, int epi);
extern "C" int dlip_conv_rows_plan(const dlip_conv_desc* d, int* bm);
extern "C" long long dlip_conv_rows_tiles(const dlip_conv_desc* d);
extern "C" int dlip_conv_rows_pool_plan(const dlip_conv_desc* d, int* bm);

// Diagnostic switch (dlip_debug_set DLIP_DBG_DMA_ENABLE = 0): keeps split-format launches on the register-staged kernel.
extern "C" __attribute__((visibility("hidden"))) int dlip_conv_dma_enabled(void) { return dlip_dbg_value[DLIP_DBG_DMA_ENABLE] != 0; }

namespace {

// Shared argument checks of the three split-fp16 entry points; fills `a`.
int fill_f16x3(const dlip_conv_desc* d, const float* x, const void* w_split, const float* w_scale, const float* bias,
               const float* residual, const float* slope, const float* post_scale, const float* post_shift, float* y,
               int32_t flags, ConvArgs* a, int max_taps = 32) {
  DLIP_CHECK_ARG(d && w_scale);
  // flags: DLIP_SPLIT_IN (x and residual hold (hi, lo) pairs; C, ldx, ldr multiples of 32),
  //        DLIP_SPLIT_OUT (y is written in that format; K, ldy multiples of 32)
  if (flags & 1) DLIP_CHECK_ARG((d->C & 31) == 0 && (d->ldx & 31) == 0 && (residual == nullptr || ((d->ldr & 31) == 0 && (d->K & 31) == 0)));
  if (flags & 2) DLIP_CHECK_ARG((d->K & 31) == 0 && (d->ldy & 31) == 0);
  const int Cw = (d->C + 31) / 32 * 32;
  const int rc = dlip_fill_conv_args(d, x, static_cast<const float*>(w_split), bias, residual, slope, post_scale,
                                     post_shift, y, Cw, a, max_taps);
  if (rc != DLIP_OK) return rc;
  a->wscale = w_scale;
  a->status = dlip_range_for(DLIP_ST_CONV);
  return DLIP_OK;
}

}  // namespace

extern "C" int dlip_conv_nhwc_f16x3(const dlip_conv_desc* d, const float* x, const void* w_split,
                                    const float* w_scale, const float* bias, const float* residual,
                                    const float* slope, const float* post_scale, const float* post_shift,
                                    float* y, int32_t flags, dlip_stream_t stream) {
  ConvArgs a;
  // Filters with more than 32 taps (a weight gradient run as a convolution: the "filter" is the output-gradient map) exist on the
  // LDS-DMA kernel only: split input, fp32 output, no residual.
  const bool big = d && d->R > 0 && d->S > 0 && (long long)d->R * d->S > 32;
  if (big) DLIP_CHECK_ARG((flags & 3) == 1 && residual == nullptr && (long long)d->R * d->S <= 65536);
  const int rc = fill_f16x3(d, x, w_split, w_scale, bias, residual, slope, post_scale, post_shift, y, flags, &a, big ? 65536 : 32);
  if (rc != DLIP_OK) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // Split-format activations go to the LDS-DMA kernel.  Its epilogue leaves in 16-byte chunks, so an fp32
  // output needs K, ldy in multiples of 4 and a 16-byte aligned y (a split output already has K, ldy % 32 == 0);
  // anything else stays on the register-staged kernel below.
  const bool dma_ok = (flags & 2) ? (reinterpret_cast<uintptr_t>(y) & 15) == 0
                                  : ((d->K & 3) == 0 && (d->ldy & 3) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0);
  const bool res_ok = residual == nullptr || (reinterpret_cast<uintptr_t>(residual) & 15) == 0;
  if ((flags & 1) && dma_ok && res_ok && dlip_conv_dma_enabled()) {
    // narrow same-size 3x3 layers (layer 1: 64 -> 64): one activation window per channel slice instead of nine tap fetches
    if (dlip_conv_win_ok(&a)) return dlip_conv_f16x3_win_launch(&a, stream, (flags & 2) ? 1 : 0);
    // the speech encoder's 1-D "valid" convolutions and k = 1 GEMMs over all frames: persistent 160 x 256 tiles, continuous slice stream
    if (dlip_conv_rows_ok(&a)) return dlip_conv_f16x3_rows_launch(&a, stream, (flags & 2) ? 1 : 0);
    return dlip_conv_f16x3_dma_launch(&a, stream, (flags & 2) ? 1 : 0);
  }
  if (big) return DLIP_EINVAL;
  switch (pick_tile(a.M, d->K, kEffF16x3)) {
    case 0: return launch<128, 128, 2, 2>(a, st, flags);
    case 1: return launch<128, 64, 2, 2>(a, st, flags);
    case 2: return launch<64, 64, 2, 2>(a, st, flags);
    case 3: return launch<64, 128, 1, 4>(a, st, flags);
    default: return launch<96, 128, 1, 4>(a, st, flags);
  }
}

// dlip_conv_nhwc_f16x3 (split input, fp32 output, no residual) that also leaves the column sums of its output (include/deeplip_hip.h).
extern "C" int32_t dlip_conv_stats_chunks(const dlip_conv_desc* d) {
  if (!d || d->N <= 0 || d->Ho <= 0 || d->Wo <= 0 || (d->C & 31) != 0 || (d->K & 3) != 0 || (d->ldy & 3) != 0 || !dlip_conv_dma_enabled()) return 0;
  const long long chunks = 2 * dlip_conv_rows_tiles(d);        // (the window kernel is asked first by the launch: not a rows shape then)
  return chunks > 0 && chunks <= 0x7FFFFFFF ? (int32_t)chunks : 0;
}

extern "C" int dlip_conv_nhwc_stats_f16x3(const dlip_conv_desc* d, const float* x, const void* w_split, const float* w_scale,
                                          const float* bias, const float* slope, float* y, double* stats, int64_t stats_bytes,
                                          dlip_stream_t stream) {
  DLIP_CHECK_ARG(d && stats && (reinterpret_cast<uintptr_t>(stats) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0);
  const int32_t chunks = dlip_conv_stats_chunks(d);
  DLIP_CHECK_ARG(chunks > 0 && stats_bytes >= (int64_t)chunks * d->K * 2 * 8);
  ConvArgs a;
  const int rc = fill_f16x3(d, x, w_split, w_scale, bias, nullptr, slope, nullptr, nullptr, y, DLIP_SPLIT_IN, &a);
  if (rc != DLIP_OK) return rc;
  DLIP_CHECK_ARG(!dlip_conv_win_ok(&a) && dlip_conv_rows_ok(&a));
  a.stats = stats;
  return dlip_conv_f16x3_rows_launch(&a, stream, 0);
}

// A convolution's weight gradient run as a convolution over SLICE-major operand images (include/deeplip_hip.h).
extern "C" int dlip_wgrad_conv_f16x3(const float* x_img, const float* g_img, const float* post_scale, const float* post_shift,
                                     const float* unit_scale, float* dw, int32_t C, int32_t H, int32_t W, int32_t K, int32_t Ho,
                                     int32_t Wo, int32_t N32, int32_t stride_h, int32_t stride_w, int32_t pad_h, int32_t pad_w,
                                     int32_t dil_h, int32_t dil_w, int32_t R, int32_t S, dlip_stream_t stream) {
  DLIP_CHECK_ARG(x_img && g_img && dw && unit_scale && C > 0 && H > 0 && W > 0 && K > 0 && Ho > 0 && Wo > 0 && N32 > 0 && (N32 & 31) == 0);
  DLIP_CHECK_ARG((K & 3) == 0 && (reinterpret_cast<uintptr_t>(dw) & 15) == 0 && dlip_conv_dma_enabled());
  // the LAYER computes y[n, ho, wo] from x[n, ho * stride + r * dil - pad, ...]; its weight gradient is the convolution of x' with the
  // filter g' whose taps are the (ho, wo): convolution stride = the layer's dilation, convolution dilation = the layer's stride
  dlip_conv_desc d;
  d.N = C; d.H = H; d.W = W; d.C = N32; d.K = K; d.R = Ho; d.S = Wo;
  d.stride_h = dil_h; d.stride_w = dil_w; d.pad_h = pad_h; d.pad_w = pad_w; d.dil_h = stride_h; d.dil_w = stride_w;
  d.Ho = (H + 2 * pad_h - stride_h * (Ho - 1) - 1) / dil_h + 1;
  d.Wo = (W + 2 * pad_w - stride_w * (Wo - 1) - 1) / dil_w + 1;
  d.ldx = N32; d.ldy = K; d.ldr = 0;
  DLIP_CHECK_ARG(d.Ho > 0 && d.Wo > 0 && (long long)Ho * Wo <= 65536);
  ConvArgs a;
  const int rc = fill_f16x3(&d, x_img, g_img, unit_scale, nullptr, nullptr, nullptr, post_scale, post_shift, dw, 1, &a, 65536);
  if (rc != DLIP_OK) return rc;
  // slice-major images: [image][N32 / 32][H][W][32] -- a pixel is 128 B, a slice a whole image plane, a tap the next 128-byte line
  const int NS = N32 / 32;
  a.ldx = 32;
  a.Hs = NS * H;
  a.cs_x = H * W * 128;
  a.wt = 128;
  a.cs_w = Ho * Wo * 128;
  DLIP_CHECK_ARG((long long)C * NS * H * W * 128 < (1ll << 31) && (long long)NS * Ho * Wo * 128 < (1ll << 31));
  if (R > 0 || S > 0) {   // dw = the reference layout [K, C, R, S]: the layer's own filter extent out of the R' x S' the convolution computes
    DLIP_CHECK_ARG(R > 0 && S > 0 && R <= d.Ho && S <= d.Wo && (long long)K * C * R * S * 4 <= DLIP_MAX_BUFFER_BYTES);
    a.wg_R = R; a.wg_S = S;
    a.y_bytes = (uint32_t)((long long)K * C * R * S * 4);
  }
  return dlip_conv_f16x3_dma_launch(&a, stream, 0);
}

// conv + 1x1 strided shortcut convolution in ONE reduction (include/deeplip_hip.h).
extern "C" int dlip_conv2_nhwc_f16x3(const dlip_conv_desc* d, const float* x, const float* x2, int32_t H2, int32_t W2,
                                     int32_t C2, int32_t ldx2, int32_t stride2_h, int32_t stride2_w, const void* w_split,
                                     const float* w_scale, const float* bias, const float* residual, const float* slope,
                                     const float* post_scale, const float* post_shift, float* y, int32_t flags,
                                     dlip_stream_t stream) {
  DLIP_CHECK_ARG(d && x2 && (flags & 1) && H2 > 0 && W2 > 0 && C2 > 0 && (C2 & 31) == 0 && (ldx2 & 31) == 0 && ldx2 >= C2);
  DLIP_CHECK_ARG(stride2_h > 0 && stride2_w > 0 && (reinterpret_cast<uintptr_t>(x2) & 15) == 0);
  ConvArgs a;
  const int rc = fill_f16x3(d, x, w_split, w_scale, bias, residual, slope, post_scale, post_shift, y, flags, &a);
  if (rc != DLIP_OK) return rc;
  // the shortcut reads pixel (ho * s2h, wo * s2w) of x2 for output pixel (ho, wo): it must exist for every one
  DLIP_CHECK_ARG((d->Ho - 1) * stride2_h < H2 && (d->Wo - 1) * stride2_w < W2);
  const long long x2_bytes = (((long long)d->N * H2 * W2 - 1) * ldx2 + C2) * 4;
  const long long w_bytes = (long long)d->K * ((long long)a.rsc + C2) * 4;
  if (x2_bytes > DLIP_MAX_BUFFER_BYTES || w_bytes > DLIP_MAX_BUFFER_BYTES) return DLIP_ERANGE;
  a.x2 = x2; a.x2_bytes = (uint32_t)x2_bytes;
  a.H2 = H2; a.W2 = W2; a.ldx2 = ldx2; a.s2h = stride2_h; a.s2w = stride2_w;
  a.nk2 = C2 / 32;
  a.nk += a.nk2;
  a.rsc += C2;
  a.w_bytes = (uint32_t)w_bytes;
  const bool y_ok = (flags & 2) ? true : ((d->K & 3) == 0 && (d->ldy & 3) == 0);
  DLIP_CHECK_ARG(y_ok && (reinterpret_cast<uintptr_t>(y) & 15) == 0 && (residual == nullptr || (reinterpret_cast<uintptr_t>(residual) & 15) == 0));
  return dlip_conv_f16x3_dma_launch(&a, stream, (flags & 2) ? 1 : 0);
}

// conv whose epilogue keeps only segmented column sums (include/deeplip_hip.h).
extern "C" int64_t dlip_conv_pool_partial_bytes(const dlip_conv_desc* d, int32_t* tile_rows) {
  if (!d || d->N <= 0 || d->Ho <= 0 || d->Wo <= 0 || d->K <= 0) return DLIP_EINVAL;
  int bm = 0, bn = 0;
  const long long M = (long long)d->N * d->Ho * d->Wo;
  dlip_conv_dma_tile(M, d->K, d->R * d->S * ((d->C + 31) / 32), 2, &bm, &bn);
  // the rows kernel's pooled epilogue (conv_rows_f16x3.hip): one partial row set per WAVE ROW = half of its BM x 256 tile; columns
  // rounded up to 128 like the ring kernel's, so both feed the same finishers
  if (int rbm = 0; dlip_conv_rows_pool_plan(d, &rbm)) { bm = rbm / 2; bn = 128; }
  if (tile_rows) *tile_rows = bm;
  const long long tiles_m = (M + bm - 1) / bm, Kp = (long long)(d->K + bn - 1) / bn * bn;
  return tiles_m * 4 * Kp * 8;
}

extern "C" int dlip_conv_pool_f16x3(const dlip_conv_desc* d, const float* x, const void* w_split, const float* w_scale,
                                    const float* bias, const float* residual, const float* slope, const float* post_scale,
                                    const float* post_shift, double* partials, int64_t partial_bytes, int32_t group_rows,
                                    const int32_t* group_len, int32_t len_mul, int32_t len_add, dlip_stream_t stream) {
  DLIP_CHECK_ARG(d && partials && group_rows > 0 && (reinterpret_cast<uintptr_t>(partials) & 7) == 0);
  ConvArgs a;
  dlip_conv_desc dd = *d;
  dd.ldy = d->K;   // no y: the descriptor's output stride is not used
  float* no_y = reinterpret_cast<float*>(partials);   // (never written: the pooled epilogue has no y stores)
  const int rc = fill_f16x3(&dd, x, w_split, w_scale, bias, residual, slope, post_scale, post_shift, no_y, DLIP_SPLIT_IN, &a);
  if (rc != DLIP_OK) return rc;
  int32_t bm = 0;
  const int64_t need = dlip_conv_pool_partial_bytes(d, &bm);
  DLIP_CHECK_ARG(need > 0 && partial_bytes >= need && group_rows >= bm);
  DLIP_CHECK_ARG(residual == nullptr || (reinterpret_cast<uintptr_t>(residual) & 15) == 0);
  a.y = nullptr; a.y_bytes = 0;
  a.pool = partials;
  a.pool_group = group_rows;
  a.pool_len.len = group_len; a.pool_len.mul = len_mul; a.pool_len.add = len_add;
  if (residual == nullptr && dlip_conv_rows_pool_plan(d, nullptr) && dlip_conv_rows_ok(&a)) return dlip_conv_f16x3_rows_launch(&a, stream, 2);
  return dlip_conv_f16x3_dma_launch(&a, stream, 2);
}


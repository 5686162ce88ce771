// Split-fp16 implicit-GEMM convolution, LDS-DMA variant: the kernel behind dlip_conv_nhwc_f16x3 when
// the activations arrive in the split activation format (DLIP_SPLIT_IN).
//
// With x already stored as (hi, lo) fp16 pairs, BOTH operands of a reduction slice are plain 128-B
// row copies, so neither passes through registers: every wave moves its share of the slice with
// `buffer_load_dwordx4 ... lds` (1 KiB = 8 rows per wave-instruction, straight from L2/L1 into LDS).
// That removes, per slice, the staging VGPRs (48-64 per lane in conv_igemm_f16x3.hip), the
// ds_write_b128 pass -- the ~79 B/clk/CU VGPR->LDS store path was the busiest unit after the matrix
// core -- and the vmcnt waits in front of it; the freed registers pay for 128x128 / 256x128 workgroup
// tiles, which halve the L2->LDS bytes per MFMA (at 64x128 the 56 B/clk/CU L2 port needs as many
// cycles per slice as the three f16 MFMAs do).
//
//   * LDS image per stage: BM + BN rows of 128 B (64 B of hi, 64 B of lo), 16-B chunk p of row r at
//     position p ^ ((r >> 1) & 7): the same conflict-free image as the register-staged kernels.  An
//     LDS-DMA writes lane-linear (lane i -> base + 16 i), so the XOR goes on the SOURCE side: lane
//     (r & 7, p) of a piece fetches chunk p ^ key(r) of its row.
//   * Padding taps, rows past M and weight rows past K use the out-of-range buffer offset: an
//     out-of-range LDS-DMA writes zeros (probed on gfx950: tools/probes/ldsdma_oob.hip).
//   * NSTAGE-deep LDS ring, slices issued NSTAGE-1 ahead; one raw s_barrier per slice behind a
//     COUNTED s_waitcnt vmcnt (the newest slices stay in flight across the barrier).
//   * The DMA is issued from inline asm: hipcc would otherwise order every later ds_read behind it
//     with vmcnt(0) (it cannot tell the stages apart) and serialise the ring.
// MFMA program order, fragment layout, accumulator initialisation and epilogue are those of
// conv_igemm_f16x3.hip.
#include "conv_common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int ROWB = 128;  // bytes per LDS row (one 32-channel slice of one pixel / filter)

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Buffer descriptor as four SGPR dwords (what an asm operand can carry): raw buffer, stride 0,
// `bytes` records, the same DATA_FORMAT word as dlip_make_rsrc.
__device__ __forceinline__ u32x4 make_rsrc_words(const void* p, uint32_t bytes) {
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  u32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((uint32_t)v);
  r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32) & 0xFFFFu);
  r[2] = __builtin_amdgcn_readfirstlane(bytes);
  r[3] = 0x00020000u;
  return r;
}

// One LDS-DMA piece: 64 lanes x 16 B from (rsrc, voff) to LDS bytes [lds_base, lds_base + 1024).
__device__ __forceinline__ void dma_piece(const u32x4 rsrc, uint32_t voff, uint32_t lds_base) {
  asm volatile(
      "s_mov_b32 m0, %0\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %1, %2, 0 offen lds"
      :
      : "s"(lds_base), "v"(voff), "s"(rsrc)
      : "memory");   // m0 is reserved: hipcc keeps nothing in it across statements (the ISA dump shows no other m0 use)
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool OSPLIT, int NSTAGE, int OCC>
__global__ __launch_bounds__(64 * WAVES_M* WAVES_N, OCC) void conv_igemm_f16x3_dma_kernel(const ConvArgs a) {
  constexpr int NW = WAVES_M * WAVES_N, NT = 64 * NW;
  constexpr int RPP = NT / 8;   // rows one pass of the workgroup covers (8 lanes x 16 B per row)
  static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must be whole passes");
  static_assert(NSTAGE == 2 || NSTAGE == 3, "ring depth");
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int MI = WM / 32, NI = WN / 32;
  constexpr int A_PER = BM / RPP, B_PER = BN / RPP;
  constexpr int NL = A_PER + B_PER;   // DMA instructions per wave per slice
  constexpr int STAGE_B = (BM + BN) * ROWB;
  constexpr int LDK = 32;             // dwords per LDS row
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tile_n = swz % a.tiles_n;
  const int tile_m = swz / a.tiles_n;

  const int tid = threadIdx.x;
  const int cq = tid & 7;
  const int rbase = tid >> 3;
  const int key_st = (rbase >> 1) & 7;          // RPP is a multiple of 16: the key is the same in every pass
  const int csrc = ((cq ^ key_st) << 2);        // first channel (dword) of the chunk this lane fetches
  const u32x4 xr = make_rsrc_words(a.x, a.x_bytes);
  const u32x4 wr = make_rsrc_words(a.w, a.w_bytes);

  int a_off[A_PER];
  uint32_t a_mask[A_PER];
#pragma unroll
  for (int j = 0; j < A_PER; ++j) {
    const int m = tile_m * BM + rbase + RPP * j;
    a_off[j] = 0;
    a_mask[j] = 0u;
    if (m < a.M) {
      const int n = m / a.HoWo;
      const int rem = m - n * a.HoWo;
      const int ho = rem / a.Wo;
      const int wo = rem - ho * a.Wo;
      const int hi0 = ho * a.sh - a.ph, wi0 = wo * a.sw - a.pw;
      a_off[j] = (((n * a.H + hi0) * a.W + wi0) * a.ldx + csrc) * 4;
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
    const int n = tile_n * BN + rbase + RPP * j;
    b_off[j] = n < a.K ? (n * a.rsc + csrc) * 4 : -1;
  }

  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const uint32_t piece0 = lds0 + wave * 8 * ROWB;   // this wave's 8 rows of pass 0, stage 0, operand A

  int tap = 0, x_tap = 0, w_tap = 0, c0 = 0;
  int s_pos = 0, x_row = 0;
  const int x_dr = a.dh * a.W * a.ldx * 4, x_ds = a.dw * a.ldx * 4;
  const int ntaps = a.R * a.S;
  // Reduction walk: 32-channel slice OUTER, filter tap INNER (conv_igemm_f16x3.hip).
  auto advance = [&]() {
    ++tap;
    if (++s_pos == a.S) { s_pos = 0; x_row += x_dr; }
    if (tap == ntaps) { tap = 0; s_pos = 0; x_row = 0; c0 += BK; }
    x_tap = x_row + s_pos * x_ds + c0 * 4;
    w_tap = (tap * a.Cw + c0) * 4;
  };
  auto issue_a = [&](int stage) {
    const uint32_t base = piece0 + stage * STAGE_B;
#pragma unroll
    for (int j = 0; j < A_PER; ++j) {
      const bool ok = (a_mask[j] >> tap) & 1u;
      dma_piece(xr, ok ? (uint32_t)(a_off[j] + x_tap) : DLIP_OOB_OFFSET, base + j * RPP * ROWB);
    }
  };
  auto issue_b = [&](int stage) {
    const uint32_t base = piece0 + stage * STAGE_B + BM * ROWB;
#pragma unroll
    for (int j = 0; j < B_PER; ++j)
      dma_piece(wr, b_off[j] >= 0 ? (uint32_t)(b_off[j] + w_tap) : DLIP_OOB_OFFSET, base + j * RPP * ROWB);
  };

  const int lane = tid & 63;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int lrow = lane & 31, half = lane >> 5;
  const int a_frag = (wm * WM + lrow) * LDK;
  const int b_frag = BM * LDK + (wn * WN + lrow) * LDK;
  const int rquad = half * 4;
  const int key_rd = (lrow >> 1) & 7;
  int khi[2], klo[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    khi[s] = ((2 * s + half) ^ key_rd) << 2;
    klo[s] = ((4 + 2 * s + half) ^ key_rd) << 2;
  }

  // ---- prologue: put the first NSTAGE-1 slices in flight, then initialise the accumulators ----
  constexpr int PF = NSTAGE - 1;
  issue_a(0);
  issue_b(0);
  if (PF > 1 && a.nk > 1) {
    advance();
    issue_a(1);
    issue_b(1);
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
        const uint32_t off = (kok && m < a.M) ? (uint32_t)((m * a.ldr + (k & ~31)) * 4 + (k & 31) * 2) : DLIP_OOB_OFFSET;
        const _Float16 rh = __builtin_bit_cast(_Float16, __builtin_amdgcn_raw_buffer_load_b16(rr, (int)off, 0, 0));
        const _Float16 rl = __builtin_bit_cast(_Float16, __builtin_amdgcn_raw_buffer_load_b16(rr, (int)off + 64, 0, 0));
        acc[mi][ni][e] = (bias + ((float)rh + (float)rl)) * ws;
      }
    }
  }

  f16x8 fah[2][MI], fal[2][MI], fbh[2][NI], fbl[2][NI];
  auto read_frags = [&](int set, int stage, int s) {
    const float* Aw = smem + stage * (STAGE_B / 4) + a_frag;
    const float* Bw = smem + stage * (STAGE_B / 4) + b_frag;
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

  // slice 0 has landed once at most the (PF - 1) younger slices are outstanding
  if (PF > 1 && a.nk > 1) wait_vmcnt<NL>(); else wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  read_frags(0, 0, 0);

  int st_cur = 0, st_iss = (PF > 1 && a.nk > 1) ? 2 % NSTAGE : 1 % NSTAGE;   // stage the next issue goes to
  for (int kt = 0; kt < a.nk; ++kt) {
    const bool more1 = (kt + 1) < a.nk, moreP = (kt + PF) < a.nk;
    const int st_nxt = st_cur + 1 == NSTAGE ? 0 : st_cur + 1;
    // ---- k16 step 0 (fragment set 0) ----
    mfma_g(0, 0); DLIP_FENCE();
    read_frags(1, st_cur, 1); DLIP_FENCE();
    mfma_g(0, 1); DLIP_FENCE();
    if (moreP) { advance(); issue_a(st_iss); } DLIP_FENCE();
    mfma_g(0, 2); DLIP_FENCE();
    if (moreP) { issue_b(st_iss); st_iss = st_iss + 1 == NSTAGE ? 0 : st_iss + 1; } DLIP_FENCE();
    // ---- k16 step 1 (fragment set 1) ----
    mfma_g(1, 0); DLIP_FENCE();
    mfma_g(1, 1); DLIP_FENCE();
    if (more1) {
      // slice kt+1 must have landed (every wave's share: wait, then barrier); slices beyond it stay in flight
      if (PF > 1 && (kt + 2) < a.nk) wait_vmcnt<NL>(); else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      read_frags(0, st_nxt, 0);
    }
    DLIP_FENCE();
    mfma_g(1, 2); DLIP_FENCE();
    st_cur = st_nxt;
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
          if ((e & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        } else {
          const uint32_t off = (kok && m < a.M) ? (uint32_t)((m * a.ldy + k) * 4) : DLIP_OOB_OFFSET;
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), yr, (int)off, 0, 0);
        }
      }
    }
  }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int NSTAGE, int OCC>
int launch_dma(const ConvArgs& a, hipStream_t st, bool out_split) {
  ConvArgs b = a;
  const int tiles_m = (a.M + BM - 1) / BM;
  b.tiles_n = (a.K + BN - 1) / BN;
  const long long grid = (long long)tiles_m * b.tiles_n;
  if (grid <= 0 || grid > 0x7FFFFFFFll) return DLIP_EINVAL;
  constexpr size_t lds = (size_t)NSTAGE * (BM + BN) * ROWB;
  static_assert(lds <= 160 * 1024, "LDS ring exceeds a CU");
  auto kern = out_split ? conv_igemm_f16x3_dma_kernel<BM, BN, WAVES_M, WAVES_N, true, NSTAGE, OCC>
                        : conv_igemm_f16x3_dma_kernel<BM, BN, WAVES_M, WAVES_N, false, NSTAGE, OCC>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * WAVES_M * WAVES_N), lds, st, b);
  return dlip_launch_status();
}

}  // namespace

// Tile menu of the DMA kernel (index = what dlip_conv_plan reports via dlip_conv_dma_tile).
const TileCfg kDmaCfg[] = {{128, 128}, {128, 64}, {64, 128}, {64, 64}, {256, 128},
                           {128, 128}, {256, 128}, {128, 256}, {256, 64}, {128, 128}};   // 5..9: experiments (env only)
constexpr int NUM_DMA_CFG = 5, NUM_DMA_ALL = 10;
const float kDmaEff[NUM_DMA_CFG] = {1.00f, 0.85f, 0.85f, 0.70f, 1.05f};

static int dma_pick(long long M, int K) {
  if (const char* e = getenv("DLIP_CONV_DMA_TILE")) {
    const int v = atoi(e);
    if (v >= 0 && v < NUM_DMA_ALL) return v;
  }
  int best = 0;
  double best_cost = 1e300;
  for (int i = 0; i < NUM_DMA_CFG; ++i) {
    const TileCfg& c = kDmaCfg[i];
    const long long tiles = ((M + c.bm - 1) / c.bm) * ((K + c.bn - 1) / c.bn);
    const long long slots = c.bm * c.bn >= 256 * 128 ? 256 : 512;   // workgroups resident at once
    const long long rounds = (tiles + slots - 1) / slots;
    const double cost = (double)rounds * slots * c.bm * c.bn / kDmaEff[i];
    if (cost < best_cost * 0.999) { best_cost = cost; best = i; }
  }
  return best;
}

// Library-internal entry points (hidden): ConvArgs lives in an unnamed namespace, so it crosses the
// translation-unit boundary as an opaque pointer.
extern "C" __attribute__((visibility("hidden"))) void dlip_conv_dma_tile(long long M, int K, int* bm, int* bn) {
  const TileCfg& c = kDmaCfg[dma_pick(M, K)];
  *bm = c.bm;
  *bn = c.bn;
}

// Called by dlip_conv_nhwc_f16x3 for DLIP_SPLIT_IN launches (argument checks done there).
extern "C" __attribute__((visibility("hidden"))) int dlip_conv_f16x3_dma_launch(const void* args, void* stream, int out_split) {
  const ConvArgs& a = *static_cast<const ConvArgs*>(args);
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (dma_pick(a.M, a.K)) {
    case 0: return launch_dma<128, 128, 2, 2, 2, 2>(a, st, out_split);
    case 1: return launch_dma<128, 64, 2, 2, 3, 2>(a, st, out_split);
    case 2: return launch_dma<64, 128, 2, 2, 3, 2>(a, st, out_split);
    case 3: return launch_dma<64, 64, 2, 2, 3, 2>(a, st, out_split);
    case 4: return launch_dma<256, 128, 4, 2, 2, 1>(a, st, out_split);
    case 5: return launch_dma<128, 128, 2, 2, 3, 1>(a, st, out_split);
    case 6: return launch_dma<256, 128, 4, 2, 3, 1>(a, st, out_split);
    case 7: return launch_dma<128, 256, 2, 4, 2, 1>(a, st, out_split);
    case 8: return launch_dma<256, 64, 4, 2, 2, 1>(a, st, out_split);
    default: return launch_dma<128, 128, 4, 2, 3, 1>(a, st, out_split);
  }
}

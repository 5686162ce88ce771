// Split-fp16 convolution with the WHOLE filter bank resident in LDS: the kernel behind dlip_conv_nhwc_f16x3
// for the 64 -> 64 channel 3x3 layers of the trunk's layer1 (resnet.py:55-69; 4 launches per step, a quarter
// of the step), split-format activations in and out.
//
// Why a second kernel: on these layers the LDS-DMA ring kernel is neither matrix- nor memory-bound (MFMA-busy
// 0.28): a tile is only 18 slices deep, so its set-up, prologue latency and epilogue weigh as much as its main
// loop, every slice costs a workgroup barrier, and 5.5 vector instructions are issued per MFMA.
// With C = K = 64 the split filter bank is 9 taps x 2 channel slices x 64 rows x 128 B = 144 KB: it fits the
// CU's LDS once and for all.  So:
//   * one persistent 8-wave workgroup per CU stages the weights ONCE (LDS-DMA, the ring kernel's swizzled row
//     image) -- no weight traffic, no ring, NO barrier after that;
//   * every wave works alone on 64-pixel x 64-channel tiles: the activation fragments of a (tap, slice) go
//     straight from L2/L1 to registers (two 16-B buffer loads per 16-pixel block: the lane's hi and lo chunk;
//     halo taps and rows past M = out-of-range offset = zeros), the next slice's loads are in flight while
//     this one is multiplied; weight fragments come from LDS;
//   * the epilogue is per wave too: a lane owns 4 consecutive channels of a pixel per accumulator quad, i.e.
//     one 8-B hi piece and one 8-B lo piece of the split output row -- residual pieces are loaded, results
//     stored, directly (no staging, no barrier).
// STATUS: EXPERIMENT, OFF by default (DLIP_CONV_WRES=1 enables it for large M, =2 whenever the shape allows; the
// parity tests force it).  Measured on layer 1 at the benchmark batch (same box, interleaved): 370 us against the
// ring kernel's 328 us.  The activation fragments of the 16x16x32 MFMA put the 16 lanes of a load pass on 16
// different pixels, so a 16-B-per-lane load becomes 64 separate 64-B requests -- a coalesced lane order (which would
// need a ds_bpermute pass to restore the MFMA layout) measured 348 us, still behind: without LDS sharing every wave
// pulls its own copy of the activations from L2 (2.25 KB per pixel), and that request stream, not the barriers or
// the per-tile overheads it removes, is what bounds the layer.
// Arithmetic, formats and results are those of conv_igemm_f16x3_dma.hip (v_mfma_f32_16x16x32_f16, groups
// lo*hi, hi*hi, hi*lo in fp32 accumulators; reduction order tap-major then channel slice, so the bits differ
// from the ring kernel's only by the fp32 summation order).
#include "conv_common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int ROWB = 128;
constexpr int WR_K = 64, WR_C = 64;          // the instance: 64 -> 64 channels
constexpr int WR_MAXTAPS = 9;
constexpr int WR_SLICE_B = WR_K * ROWB;      // 8 KB: one (tap, channel slice) of the filter bank
constexpr int WR_TILE = 64;                  // pixels per wave tile
constexpr int WR_MI = WR_TILE / 16, WR_NI = WR_K / 16;

__device__ __forceinline__ u32x4 rsrc_words(const void* p, uint32_t bytes) {
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  u32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((uint32_t)v);
  r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32) & 0xFFFFu);
  r[2] = __builtin_amdgcn_readfirstlane(bytes);
  r[3] = 0x00020000u;
  return r;
}

__device__ __forceinline__ void dma_piece(const u32x4 rsrc, uint32_t voff, uint32_t lds_base) {
  const uint32_t base = __builtin_amdgcn_readfirstlane(lds_base);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(base), "v"(voff), "s"(rsrc) : "memory");
}

template <bool OSPLIT>
__global__ __launch_bounds__(512, 1) void conv_wres_f16x3_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntaps = a.R * a.S;
  const int nsl = ntaps * 2;                              // (tap, channel slice) pairs, tap-major
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  // ---- stage the filter bank: slice s = tap * 2 + cb -> LDS [s][64 rows][128 B], chunk p of row k at p ^ ((k >> 1) & 7)
  {
    const u32x4 wr = rsrc_words(a.w, a.w_bytes);
    const int r8 = lane >> 3, cq = lane & 7;
    for (int p = wave; p < nsl * 8; p += 8) {               // piece = 8 rows of one slice
      const int s = p >> 3, k = (p & 7) * 8 + r8;
      const int tap = s >> 1, cb = s & 1;
      const int chunk = cq ^ ((k >> 1) & 7);
      dma_piece(wr, (uint32_t)((k * a.rsc + tap * a.Cw + cb * 32) * 4 + chunk * 16), lds0 + p * 1024);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();

  const __amdgpu_buffer_rsrc_t rr = dlip_make_rsrc(a.res, a.res ? a.r_bytes : 0u);
  const __amdgpu_buffer_rsrc_t yr = dlip_make_rsrc(a.y, a.y_bytes);
  const int px = lane & 15, kg = lane >> 4;
  // weight fragment address of this lane inside a slice (hi chunk; lo = ^ 64): row px of block ni
  const int w_ad = px * ROWB + ((kg ^ ((px >> 1) & 7)) << 4);
  const char* lds_c = reinterpret_cast<const char*>(smem);
  const int x_dr = a.dh * a.W * a.ldx * 4, x_ds = a.dw * a.ldx * 4;

  const int n_tiles = (a.M + WR_TILE - 1) / WR_TILE;
  const int n_waves = gridDim.x * 8;
  for (int wt = blockIdx.x + gridDim.x * wave; wt < n_tiles; wt += n_waves) {   // a workgroup's 8 waves take tiles b, b + G, ...
    const int m0 = wt * WR_TILE;
    // per-block pixel: byte offset of the window origin (tap (0,0), this lane's hi chunk) and tap-validity bits
    int a_off[WR_MI];
    uint32_t a_mask[WR_MI];
#pragma unroll
    for (int mi = 0; mi < WR_MI; ++mi) {
      const int m = m0 + mi * 16 + px;
      const int mc = m < a.M ? m : a.M - 1;
      const int n = mc / a.HoWo;
      const int rem = mc - n * a.HoWo;
      const int ho = rem / a.Wo;
      const int wo = rem - ho * a.Wo;
      const int hi0 = ho - a.ph, wi0 = wo - a.pw;
#ifdef DLIP_WRES_COALESCED_EXPERIMENT   // timing experiment (wrong results): lane -> (pixel lane >> 2, chunk lane & 3), 16 requests per load
      {
        const int m2 = m0 + mi * 16 + (lane >> 2);
        const int mc2 = m2 < a.M ? m2 : a.M - 1;
        const int n2 = mc2 / a.HoWo, rem2 = mc2 - n2 * a.HoWo, ho2 = rem2 / a.Wo, wo2 = rem2 - ho2 * a.Wo;
        a_off[mi] = (((n2 * a.H + ho2 - a.ph) * a.W + wo2 - a.pw) * a.ldx + (lane & 3) * 4) * 4;
      }
#else
      a_off[mi] = (((n * a.H + hi0) * a.W + wi0) * a.ldx + kg * 4) * 4;
#endif
      uint32_t colbits = 0u, mask = 0u;
      for (int sx = 0; sx < a.S; ++sx) colbits |= (uint32_t)((unsigned)(wi0 + sx * a.dw) < (unsigned)a.W) << sx;
      for (int r = 0; r < a.R; ++r) mask |= ((unsigned)(hi0 + r * a.dh) < (unsigned)a.H ? colbits : 0u) << (r * a.S);
      a_mask[mi] = m < a.M ? mask : 0u;
    }

    f32x4 acc[WR_MI][WR_NI];
#pragma unroll
    for (int mi = 0; mi < WR_MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < WR_NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    // activation fragments of slice s: [hi | lo] per pixel block; NBUF register buffers = NBUF - 1 slices of loads in
    // flight ahead of the one being multiplied (one slice is ~770 MFMA cycles per wave: a single slice of lead does
    // not cover an L2 / HBM round trip under load).  The loads and their waits are inline asm: hipcc's own vmcnt
    // bookkeeping gives up on the rotating buffers (it waited for the loads it had just issued), so the loads are
    // hidden from it, ALWAYS issued (out of range past the last slice: the count in flight is static) and each
    // buffer is released to its MFMAs by an `s_waitcnt vmcnt(2 slices)` that names the buffer's registers as
    // in/out operands (a data dependence the scheduler cannot move the MFMAs across).
    constexpr int NBUF = 3;
    u32x4 ah[NBUF][WR_MI], al[NBUF][WR_MI];
    const u32x4 xrw = rsrc_words(a.x, a.x_bytes);
    auto load_a = [&](int buf, int tap, int cb, int toff, bool live) {
#pragma unroll
      for (int mi = 0; mi < WR_MI; ++mi) {
        const bool ok = live && ((a_mask[mi] >> tap) & 1u);
        const uint32_t oh = ok ? (uint32_t)(a_off[mi] + toff + cb * 128) : DLIP_OOB_OFFSET;
        const uint32_t ol = ok ? oh + 64 : DLIP_OOB_OFFSET;
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=&v"(ah[buf][mi]) : "v"(oh), "s"(xrw) : "memory");
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=&v"(al[buf][mi]) : "v"(ol), "s"(xrw) : "memory");
      }
    };
    auto release = [&](int buf) {   // this buffer's 2 * WR_MI loads have landed once at most the 2 younger slices are outstanding
      static_assert(WR_MI == 4, "operand list below");
      asm volatile("s_waitcnt vmcnt(%8)"
                   : "+v"(ah[buf][0]), "+v"(ah[buf][1]), "+v"(ah[buf][2]), "+v"(ah[buf][3]), "+v"(al[buf][0]), "+v"(al[buf][1]),
                     "+v"(al[buf][2]), "+v"(al[buf][3])
                   : "n"((NBUF - 1) * 2 * WR_MI)
                   : "memory");
    };
    int tap_n = 0, cb_n = 0, s_pos = 0, x_row = 0;       // the slice being loaded next
    auto next_slice = [&]() {
      if (++cb_n == 2) {
        cb_n = 0;
        ++tap_n;
        if (++s_pos == a.S) { s_pos = 0; x_row += x_dr; }
      }
    };
#pragma unroll
    for (int b = 0; b < NBUF - 1; ++b) { load_a(b, tap_n, cb_n, x_row + s_pos * x_ds, b < nsl); next_slice(); }
#pragma unroll 1
    for (int s = 0; s < nsl; s += NBUF) {                  // NBUF slices per trip: buffer indices are static
#pragma unroll
      for (int ss = 0; ss < NBUF; ++ss) {
        load_a((ss + NBUF - 1) % NBUF, tap_n, cb_n, x_row + s_pos * x_ds, s + ss + NBUF - 1 < nsl);
        next_slice();
        const int cur = ss;
        // (slices past the end multiply zeros against whatever the weight address holds: clamp it into the bank)
        const int sw = s + ss < nsl ? s + ss : nsl - 1;
        const char* wsl = lds_c + sw * WR_SLICE_B + w_ad;
        f16x8 bh[WR_NI], bl[WR_NI];
#pragma unroll
        for (int ni = 0; ni < WR_NI; ++ni) {
          bh[ni] = *reinterpret_cast<const f16x8*>(wsl + ni * 16 * ROWB);
          bl[ni] = *reinterpret_cast<const f16x8*>(wsl + ni * 16 * ROWB + (((w_ad >> 4) & 4) ? -64 : 64));
        }
        release(cur);
#pragma unroll
        for (int g = 0; g < 3; ++g)                        // lo*hi, hi*hi, hi*lo
#pragma unroll
          for (int mi = 0; mi < WR_MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < WR_NI; ++ni) {
              const f16x8 av = __builtin_bit_cast(f16x8, g == 0 ? al[cur][mi] : ah[cur][mi]);
              const f16x8 bv = g == 2 ? bl[ni] : bh[ni];
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bv, av, acc[mi][ni], 0, 0, 0);
            }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the out-of-range tail loads: drained before the compiler counts again

    // ---- epilogue, per wave: y = act(acc / wscale + bias + residual) * post_scale + post_shift ----
    // split row: 32-channel block b = k / 32: [32 hi halves | 32 lo halves]; a lane's 4 channels k .. k + 3 of block
    // ni: hi at 2 (k % 32) bytes, lo 64 B further.  The residual pieces of block ni + 1 are requested before block ni
    // is stored (a raw buffer load is not moved across a raw buffer store: loaded where used, every piece would
    // pay its full latency alone).
    u32x2 rh[2][WR_MI], rl[2][WR_MI];                     // residual pieces of channel block ni (parity ni & 1)
    auto load_res = [&](int ni) {
      const int k = ni * 16 + kg * 4;
      const int sp_off = (k >> 5) * 128 + (k & 31) * 2;
#pragma unroll
      for (int mi = 0; mi < WR_MI; ++mi) {
        const int m = m0 + mi * 16 + px;
        const uint32_t ro = m < a.M ? (uint32_t)(m * a.ldr * 4 + sp_off) : DLIP_OOB_OFFSET;
        rh[ni & 1][mi] = __builtin_amdgcn_raw_buffer_load_b64(rr, (int)ro, 0, 0);
        rl[ni & 1][mi] = __builtin_amdgcn_raw_buffer_load_b64(rr, (int)(m < a.M ? ro + 64 : DLIP_OOB_OFFSET), 0, 0);
      }
    };
    if (a.res) load_res(0);
#pragma unroll
    for (int ni = 0; ni < WR_NI; ++ni) {
      const int k = ni * 16 + kg * 4;                      // this lane's 4 channels of block ni
      const f32x4 ws4 = *reinterpret_cast<const f32x4*>(a.wscale + k);
      f32x4 inv4;
#pragma unroll
      for (int c = 0; c < 4; ++c) inv4[c] = __builtin_amdgcn_rcpf(ws4[c]);   // wscale is a power of two: the reciprocal is exact
      f32x4 bi4 = {0.f, 0.f, 0.f, 0.f}, sl4 = {1.f, 1.f, 1.f, 1.f}, ps4 = {1.f, 1.f, 1.f, 1.f}, pt4 = {0.f, 0.f, 0.f, 0.f};
      if (a.bias) bi4 = *reinterpret_cast<const f32x4*>(a.bias + k);
      if (a.slope) sl4 = *reinterpret_cast<const f32x4*>(a.slope + k);
      if (a.pscale) { ps4 = *reinterpret_cast<const f32x4*>(a.pscale + k); pt4 = *reinterpret_cast<const f32x4*>(a.pshift + k); }
      const int sp_off = (k >> 5) * 128 + (k & 31) * 2;
      if (a.res && ni + 1 < WR_NI) load_res(ni + 1);       // the next block's pieces, requested before this block's stores
#pragma unroll
      for (int mi = 0; mi < WR_MI; ++mi) {
        const int m = m0 + mi * 16 + px;
        const bool ok = m < a.M;
        float v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = acc[mi][ni][c] * inv4[c] + bi4[c];
        if (a.res) {
          const h4 h = __builtin_bit_cast(h4, rh[ni & 1][mi]), l = __builtin_bit_cast(h4, rl[ni & 1][mi]);
#pragma unroll
          for (int c = 0; c < 4; ++c) v[c] += (float)h[c] + (float)l[c];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          v[c] = v[c] >= 0.f ? v[c] : v[c] * sl4[c];
          if (a.pscale) v[c] = v[c] * ps4[c] + pt4[c];
        }
        if constexpr (OSPLIT) {
          h4 hi, lo;
#pragma unroll
          for (int c = 0; c < 4; ++c) { hi[c] = (_Float16)v[c]; lo[c] = (_Float16)(v[c] - (float)hi[c]); }
          const uint32_t yo = ok ? (uint32_t)(m * a.ldy * 4 + sp_off) : DLIP_OOB_OFFSET;
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hi), yr, (int)yo, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, lo), yr, (int)(ok ? yo + 64 : DLIP_OOB_OFFSET), 0, 0);
        } else {
          const f32x4 o = {v[0], v[1], v[2], v[3]};
          const uint32_t yo = ok ? (uint32_t)((m * a.ldy + k) * 4) : DLIP_OOB_OFFSET;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), yr, (int)yo, 0, 0);
        }
      }
    }
  }
}

}  // namespace

// Eligibility: 64 -> 64 channels, <= 9 taps, stride 1 (any padding / dilation: halo by mask), split input.
extern "C" __attribute__((visibility("hidden"))) int dlip_conv_wres_ok(const void* args) {
  const ConvArgs& a = *static_cast<const ConvArgs*>(args);
  const char* e = getenv("DLIP_CONV_WRES");               // default 0: off (slower than the ring kernel, see the header); 1: large M; 2: any M
  const int mode = e ? atoi(e) : 0;
  return mode && a.K == WR_K && a.C == WR_C && a.Cw == WR_C && a.R * a.S <= WR_MAXTAPS && a.sh == 1 && a.sw == 1 &&
         (a.ldx & 31) == 0 && (mode == 2 || a.M >= 64 * 2048);   // enough 64-pixel tiles to give every wave of the chip one
}

extern "C" __attribute__((visibility("hidden"))) int dlip_conv_f16x3_wres_launch(const void* args, void* stream, int out_split) {
  const ConvArgs& a = *static_cast<const ConvArgs*>(args);
  hipStream_t st = static_cast<hipStream_t>(stream);
  auto kern = out_split ? conv_wres_f16x3_kernel<true> : conv_wres_f16x3_kernel<false>;
  const size_t lds = (size_t)a.R * a.S * 2 * WR_SLICE_B;
  static bool attr[2] = {false, false};
  if (!attr[out_split ? 1 : 0]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr[out_split ? 1 : 0] = true;
  }
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
    return DLIP_EINVAL;
  hipLaunchKernelGGL(kern, dim3((unsigned)cus), dim3(512), lds, st, a);
  return dlip_launch_status();
}

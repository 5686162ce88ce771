// Split-fp16 implicit-GEMM convolution, WINDOW variant: same-size stride-1 R x S convolutions with few output
// channels (K <= 128, one column block: the four 3x3 convolutions of the trunk's layer 1, resnet.py:55-69 at 64 -> 64
// channels -- a quarter of the step's time on the ring kernel -- and the two 128 -> 128 ones of layer 2's second block).
//
// Why a second kernel.  With BN = 64 the ring kernel (conv_igemm_f16x3_dma.hip) moves (128 + 64) x 128 B of operands
// per slice for 24 MFMAs per wave: the address unit (64 B / clk / CU) needs as many cycles per slice as the matrix
// core does, and two thirds of those bytes are the activation rows, which the nine taps of a 32-channel slice read
// NINE times shifted by whole pixels.  Here a workgroup loads, per 32-channel slice, ONE window -- the BM + (R-1) dil W
// + (S-1) dil consecutive input pixels its tile touches -- and every tap reads its fragments from that window at a row
// offset; only the weights go through a per-slice ring.  L2 -> LDS bytes per tile drop 2.3x (layer 1: 442 -> 191 KB).
//
// What round 1's window mode got wrong (and lost 3-10 % with): it stored a window as eight planes, one per 16-B
// chunk of the 128-B pixel row, so that fragment reads stay bank-conflict free at every row offset -- but an LDS-DMA
// piece then gathered one chunk of 64 different pixels = 64 cache-line requests per KiB.  The layout here keeps the
// conflict-free property with 4x fewer requests: rows are grouped in BLOCKS of 16; inside a block, 256-B bank row c
// holds chunk c of the 16 rows (slot = row & 15).  A fragment read (16 consecutive rows of one chunk, any offset)
// touches 16 distinct slots = all 64 banks once; a DMA piece fills 4 bank rows = 4 chunks (the hi or the lo half of
// the slice) x 16 rows, i.e. 64 contiguous bytes of each of 16 pixels.
//
//   Taps that fall outside the image read at an address beyond the workgroup's LDS allocation (bit 18 set per lane): an
//   out-of-range ds_read returns zeros on gfx950 (probed).  Window c+1 is fetched one piece per slice during the first
//   taps of channel slice c (no burst, uniform vmcnt).
//
// With the operand traffic gone, what a 18-slice tile costs is its fixed part: set-up, the latency of its first
// fetches, the epilogue.  So the workgroups are PERSISTENT (two per CU, each walking a contiguous range of row blocks)
// and pipeline consecutive tiles: a tile's last slices issue the next tile's window 0 and first weight slices, and its
// epilogue -- from the accumulator registers straight to memory since round 4, the residual by two 16-B loads per pixel
// block -- runs with that head in flight (LDS map in the kernel).
#include "conv_common.h"
#include "conv_dma_common.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <vector>

namespace {

constexpr int WIN_SLACK = 48;   // rows a window holds beyond the tile's BM pixels: >= (R-1) dil W + (S-1) dil of the launch

template <int BM, int BN, int WAVES_M, int WAVES_N, bool OSPLIT, int OCC>
__global__ __launch_bounds__(64 * WAVES_M* WAVES_N, OCC) void conv_win_f16x3_kernel(const ConvArgs a) {
  constexpr int NW = WAVES_M * WAVES_N, NT = 64 * NW;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int FR = 16;
  constexpr int MI = WM / FR, NI = WN / FR;
  constexpr int RPP = NT / 8, B_PER = BN / RPP;
  constexpr int NSTAGE = 3;
  constexpr int WBLK0 = (BM + WIN_SLACK + 15) / 16;
  constexpr int WBLK = (WBLK0 + NW / 2 - 1) / (NW / 2) * (NW / 2);   // 16-row blocks per window: 2 WBLK pieces, whole per wave
  constexpr int WPER = 2 * WBLK / NW;                                // window pieces per wave
  constexpr int SLOT_B = WBLK * 2048;
  constexpr int BSTAGE_B = BN * ROWB;
  // LDS map.  The workgroup is PERSISTENT and pipelines consecutive tiles: while a tile's epilogue runs (from registers: it
  // touches no LDS but the parameter table), the next tile's window 0 and first two weight slices are landing.
  //   [0, SLOT_B)                      window slot 1 (odd channel slices)
  //   [SLOT_B, +BSTAGE_B)              weight ring stage 2
  //   [HEAD_OFF, +2 BSTAGE_B)          weight ring stages 0, 1 (the next tile's arrive during the last two slices)
  //   then window slot 0 (even channel slices; the next tile's arrives during the last slices), the table
  constexpr int W1_OFF = 0;
  constexpr int HEAD_OFF = SLOT_B + BSTAGE_B;      // (until round 3 the epilogue's output image lay over [0, BM BN 4) and pushed stages 0, 1 behind it)
  constexpr int W0_OFF = HEAD_OFF + 2 * BSTAGE_B;
  constexpr int TAB_OFF = W0_OFF + SLOT_B;
  constexpr int LDK = 32;
  constexpr int NSTORE = OSPLIT ? MI * NI : MI * NI;   // 16-B output stores per lane and tile (split: 2 per channel-block pair and pixel block)
  static_assert(BN % RPP == 0 && WM % 16 == 0 && 2 * WBLK % NW == 0 && NI % 2 == 0, "tile / window layout");
  // Window pieces go out PPT per slice during the first taps of the previous channel slice.  (Round 6: PPT = 2 for the geometries
  // with few waves -- 64 x 64 WAVE tiles: two or four waves own the whole window, 10 - 11 pieces each.  Why those geometries: a wave
  // tile of MI x NI fragments reads 2 KB of LDS per fragment row / column and slice for 3 MI NI MFMAs of 16 cycles; against the
  // CU's 128 B / clk that is (4 / 3)(1 / MI + 1 / NI) LDS cycles per MFMA cycle -- 1.0 for the 64 x 32 wave tiles of the 2 x 2
  // layout (the LDS port is as busy as the matrix pipes: neither can be), 0.67 for 64 x 64.)
  constexpr int PPT = (WPER + 6) / 7;
  static_assert(WPER <= 7 * PPT && PPT <= 2, "the last window piece of a channel slice goes out at least two slices before the slice ends");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* lds_c = reinterpret_cast<char*>(smem);
  auto stage_off = [](int st) { return st == 2 ? SLOT_B : HEAD_OFF + st * BSTAGE_B; };   // stages 0, 1 outside the image
  auto slot_off = [](int slot) { return (slot & 1) ? W1_OFF : W0_OFF; };

  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int g = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);   // neighbours in tile order share an XCD (L2)
  const int tiles_m = (a.M + BM - 1) / BM;                                                     // (one column block: K <= BN)
  const int t_begin = (int)((long long)g * tiles_m / nwg), t_end = (int)((long long)(g + 1) * tiles_m / nwg);
  if (t_begin >= t_end) return;
  dlip_span_enter(a.span, g);

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int lrow = lane & 15, half = lane >> 4;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const u32x4 xr = make_rsrc_words(a.x, a.x_bytes);
  const u32x4 wr = make_rsrc_words(a.w, a.w_bytes);
  const int halo_lo = a.ph * a.W + a.pw;                                 // window row of output pixel m0 at tap (0, 0) is 0
  const int need_rows = BM + (a.R - 1) * a.dh * a.W + (a.S - 1) * a.dw;  // rows a window really holds (<= BM + WIN_SLACK)

  // tap validity of this lane's MI fragment pixels of the tile starting at row m0 (bit r*S + s)
  auto tile_masks = [&](int m0, uint32_t* mask) {
    int hi0[MI], wi0[MI];
    uint32_t colbits[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int m = m0 + wm * WM + mi * 16 + lrow;
      const int mc = m < a.M ? m : a.M - 1;
      const int n = dlip_div(mc, a.div_howo);
      const int rem = mc - n * a.HoWo;
      const int ho = dlip_div(rem, a.div_wo);
      hi0[mi] = ho - a.ph;
      wi0[mi] = rem - ho * a.Wo - a.pw;
      colbits[mi] = 0u;
      mask[mi] = 0u;
    }
    for (int sx = 0; sx < a.S; ++sx)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) colbits[mi] |= (uint32_t)((unsigned)(wi0[mi] + sx * a.dw) < (unsigned)a.W) << sx;
    for (int r = 0; r < a.R; ++r)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
        mask[mi] |= ((unsigned)(hi0[mi] + r * a.dh) < (unsigned)a.H ? colbits[mi] : 0u) << (r * a.S);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
      if (m0 + wm * WM + mi * 16 + lrow >= a.M) mask[mi] = 0u;
  };

  // ---- weight ring addressing (as the ring kernel: XOR swizzle on the source side); the same for every tile ----
  const int cq = tid & 7, rbase = tid >> 3;
  const int key_st = (rbase >> 1) & 7;
  const int csrc = ((cq ^ key_st) << 2);
  uint32_t b_off[B_PER];   // (a weight row past K carries the out-of-range offset itself: one add per piece, no select)
#pragma unroll
  for (int j = 0; j < B_PER; ++j) {
    const int n = rbase + RPP * j;
    b_off[j] = n < a.K ? (uint32_t)((n * a.rsc + csrc) * 4) : DLIP_OOB_OFFSET;
  }
  const uint32_t bpiece0 = lds0 + wave * 8 * ROWB;
  auto issue_b = [&](int stage, int w_tap) {
#pragma unroll
    for (int j = 0; j < B_PER; ++j)
      dma_piece(wr, b_off[j] + (uint32_t)w_tap, bpiece0 + stage_off(stage) + j * RPP * ROWB);
  };
  // window piece q (= 2 block + half) of channel slice cc of the tile whose window starts at input pixel p0 -> slot
  auto issue_win = [&](int slot, int cc, int q, int p0) {
    const int wrow = (q >> 1) * 16 + lrow;
    const int p = p0 + wrow;
    const bool ok = wrow < need_rows && p >= 0 && p < a.M;
    dma_piece(xr, ok ? (uint32_t)((p * a.ldx + cc * BK) * 4 + ((q & 1) * 4 + half) * 16) : DLIP_OOB_OFFSET,
              lds0 + slot_off(slot) + q * 1024);
  };
  // what a tile needs before its first slice: window 0 and the weights of slices 0 and 1
  auto issue_tile_head = [&](int p0) {
#pragma unroll
    for (int j = 0; j < WPER; ++j) issue_win(0, 0, wave + NW * j, p0);
    issue_b(0, 0);
    issue_b(1, a.Cw * 4);   // slice 1 = tap 1 of channel slice 0
  };

  int m0 = t_begin * BM;
  uint32_t fr_mask[MI];
  tile_masks(m0, fr_mask);
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) fr_mask[mi] = ~fr_mask[mi];   // kept INVERTED: bit tap set = that tap falls outside the image
  issue_tile_head(m0 - halo_lo);
  if (tid < BN) {
    const int k = tid;
    const bool kok = k < a.K;
    float* tab = smem + TAB_OFF / 4;
    tab[tid] = kok ? 1.f / a.wscale[k] : 0.f;
    tab[BN + tid] = (kok && a.bias) ? a.bias[k] : 0.f;
    tab[2 * BN + tid] = (kok && a.slope) ? a.slope[k] : 1.f;
    tab[3 * BN + tid] = (kok && a.pscale) ? a.pscale[k] : 1.f;
    tab[4 * BN + tid] = (kok && a.pshift) ? a.pshift[k] : 0.f;
  }

  const int a_lane = (wm * WM / 16) * 2048 + half * 256;    // this lane's chunk row of its first block
  const int b_frag = (wn * WN + lrow) * LDK;
  const int key_rd = (lrow >> 1) & 7;
  const int khi = (half ^ key_rd) << 2, klo = ((4 + half) ^ key_rd) << 2;
  const __amdgpu_buffer_rsrc_t yr = dlip_make_rsrc(a.y, a.y_bytes);
  const bool post = a.pscale != nullptr;
  float amax = 0.f;

#ifdef DLIP_LAB
  unsigned long long* stamps = reinterpret_cast<unsigned long long*>(a.pool);   // lab: [G][8] s_memtime of each workgroup's 2nd tile
#define WIN_STAMP(i) do { if (stamps && tid == 0 && t == t_begin + 1) stamps[(size_t)g * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
// inside ONE slice (channel slice 0, tap 4 of the second tile): [4096 * 8 + g * 8 + i]
#define WIN_SSTAMP(i) do { if (stamps && tid == 0 && t == t_begin + 1 && c == 0 && tap == 4) stamps[(size_t)4096 * 8 + (size_t)g * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WIN_STAMP(i) do { } while (0)
#define WIN_SSTAMP(i) do { } while (0)
#endif
  for (int t = t_begin; t < t_end; ++t) {
    const bool first = t == t_begin, has_next = t + 1 < t_end;
    WIN_STAMP(0);
    // The walk over slices: channel slice outer (runtime), the NINE taps of the 3x3 filter inner and fully unrolled --
    // tap index, window row offset, ring stages and the piece schedule are then compile-time per tap: no tap counters,
    // no per-slice scalar bookkeeping (the rolled loop spent 47 vector + 85 scalar instructions per slice on it; the
    // stamps showed 1 720 cycles per slice against 384 of MFMA issue per wave).
    static_assert(NSTAGE == 3, "stage of a slice = tap % 3 (9 taps per channel slice)");
    constexpr int NTAPS = 9;
    const int cchunks = a.cchunks;
    f32x4 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[mi][ni][e] = 0.f;
    // Per tap: the fragment address of this lane's first 16-row block; block mi is 2048 mi further (an immediate of the read).
    // A tap that falls outside the image gets bit 18 set: the address is then beyond the workgroup's LDS allocation, and an
    // out-of-range ds_read returns ZEROS on gfx950 (tools/probes/lds_oob.hip: 0 of 524 288 lanes read anything else, with
    // several workgroups sharing the CU) -- no zero block, no select, no per-(tap, block) lane masks held in scalar registers
    // (the select form cost 44 vector instructions per slice, 12 of them v_readlane of spilled masks, against 24 MFMAs: the
    // loop was bound by vector issue, PMC: 3.6 VALU per MFMA).  u >> 4 blocks of 2048 B + (u & 15) slots of 16 B == 1792 (u >> 4) + 16 u.
    const int dW = a.dh * a.W, dS = a.dw;
    int a_ad[MI];
    auto set_addr = [&](int tap, int slot_base) {   // tap: compile-time after unrolling
      const int u = lrow + (tap / 3) * dW + (tap % 3) * dS;
      const int a0 = (u >> 4) * 1792 + (u << 4) + (slot_base + a_lane);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a_ad[mi] = a0 | (int)(((fr_mask[mi] >> tap) & 1u) << 18);
    };
    f16x8 fal[MI], fah[MI], fbh[NI], fbl[NI];
    auto read_first = [&](int stage) {   // activation lo, weight hi
      const float* Bw = smem + stage_off(stage) / 4 + b_frag;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) fal[mi] = *reinterpret_cast<const f16x8*>(lds_c + a_ad[mi] + mi * 2048 + 1024);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) fbh[ni] = *reinterpret_cast<const f16x8*>(Bw + ni * 16 * LDK + khi);
    };
    auto read_rest = [&](int stage) {    // activation hi, weight lo
      const float* Bw = smem + stage_off(stage) / 4 + b_frag;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) fah[mi] = *reinterpret_cast<const f16x8*>(lds_c + a_ad[mi] + mi * 2048);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) fbl[ni] = *reinterpret_cast<const f16x8*>(Bw + ni * 16 * LDK + klo);
    };
    auto mfma_p = [&](int grp) {   // grp 0: lo*hi, 1: hi*hi, 2: hi*lo
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const f16x8 av = grp == 0 ? fal[mi] : fah[mi];
          const f16x8 bv = grp == 2 ? fbl[ni] : fbh[ni];
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bv, av, acc[mi][ni], 0, 0, 0);
        }
    };
#define DLIP_FENCE() __builtin_amdgcn_sched_barrier(0)
    set_addr(0, slot_off(0));
    // Slice 0's weights (and, older in the queue, the whole of window 0) have landed once only what was issued after
    // them is outstanding: slice 1's pieces and -- after the first tile -- the previous epilogue's NSTORE output stores.
    // (later tiles: their head was streamed in behind the previous tile's last slices; only its output stores are younger)
    if (first) wait_vmcnt<B_PER>(); else wait_vmcnt<NSTORE>();
    __syncthreads();   // window 0 / weights visible; the previous image fully read before this tile's DMA overwrites it
    WIN_STAMP(1);
    read_first(0);

    const int p0 = m0 - halo_lo;
    const int m0n = m0 + BM;
    uint32_t next_mask[MI];
    for (int c = 0; c < cchunks; ++c) {
      const bool last_c = c + 1 == cchunks;
      // The slice stream does not stop at a tile's end: "the next channel slice" of the LAST one is channel slice 0 of the
      // NEXT tile (window slot 0 is free since this tile's slice 0 ... 8 or, with an odd number of channel slices, is the
      // slot the next tile would use anyway), and the weights "two slices ahead" of the last two taps are the next tile's
      // slices 0 and 1 (ring stages 0 and 1: outside the epilogue's image).  The epilogue then runs with the next tile's
      // head already landed or in flight.
      const bool stream_on = !last_c || has_next;
      const bool stream_win = !last_c || (has_next && (cchunks & 1) == 0);   // (an odd count leaves the last slice in slot 0)
      const int sb = slot_off(c), sb_next = last_c ? slot_off(0) : slot_off(c + 1);
      const int wbase = c * BK * 4;                     // byte offset of this channel slice inside a tap's weights
#pragma unroll
      for (int tap = 0; tap < NTAPS; ++tap) {
        // slice (c, tap) sits in ring stage tap % 3; the weights two slices ahead go to stage (tap + 2) % 3
        const bool more1 = !(last_c && tap == NTAPS - 1);
        const bool moreP = tap + 2 < NTAPS || stream_on;
        // top of the slice, right behind the barrier: one piece of the NEXT channel slice's window (its slot was read last
        // in the previous channel slice), then the weights two slices ahead
        // ORDER MATTERS: vmcnt counts in issue order, and the wait at this slice's end may leave in flight only what was issued
        // AFTER the weights of the next slice.  The window piece is the long fetch (a first touch: HBM or the Infinity Cache;
        // the weights come from L2) and is not needed before the next channel slice -- issued BEHIND the slice's weight pieces
        // it may stay in flight across two slice ends instead of one (in-kernel stamps: with the window piece first the wave
        // stood 1 000 - 1 300 cycles in that wait in six slices of nine).
        // pieces this slice / the previous slice sent out (fold per tap once the loop is unrolled)
        const int n_now = stream_win ? (tap * PPT >= WPER ? 0 : (WPER - tap * PPT < PPT ? WPER - tap * PPT : PPT)) : 0;
        const int n_prev = (stream_win && tap >= 1) ? ((tap - 1) * PPT >= WPER ? 0 : (WPER - (tap - 1) * PPT < PPT ? WPER - (tap - 1) * PPT : PPT)) : 0;
        WIN_SSTAMP(0);
        if (moreP) {
          const int wnext = last_c ? 0 : wbase + BK * 4;   // first channel slice of the next tile, or this tile's next one
          issue_b((tap + 2) % 3, tap + 2 < NTAPS ? (tap + 2) * a.Cw * 4 + wbase : (tap + 2 - NTAPS) * a.Cw * 4 + wnext);
        }
#pragma unroll
        for (int i = 0; i < PPT; ++i)
          if (i < n_now) { if (last_c) issue_win(0, 0, wave + NW * (tap * PPT + i), m0n - halo_lo); else issue_win(c + 1, c + 1, wave + NW * (tap * PPT + i), p0); }
        DLIP_FENCE();
        WIN_SSTAMP(1);
        read_rest(tap % 3); DLIP_FENCE();
        mfma_p(0); DLIP_FENCE();
        WIN_SSTAMP(2);
        // next tap's fragment addresses: plain VALU in the shadow of the matrix instructions (and, once per tile, the next
        // tile's tap masks; the last slice's set_addr runs again at the next tile's start, with those masks)
        if (tap + 1 < NTAPS) set_addr(tap + 1, sb); else set_addr(0, sb_next);
        if (tap == 0 && last_c && has_next) tile_masks(m0n, next_mask);
        DLIP_FENCE();
        mfma_p(1); DLIP_FENCE();
        WIN_SSTAMP(3);
        if (more1) {
          // Slice kt+1's weights must have landed; what was issued after them stays in flight: this slice's pieces and -- in
          // a later tile's very first slice -- the previous epilogue's NSTORE output stores, which sit between slice 1's
          // weights and this slice's pieces in the queue.  Every LDS read of this slice is complete (lgkmcnt) before the
          // barrier releases its stage; the last group's MFMAs then cover the next slice's first fragment reads.
          // (younger than the next slice's weights: the previous slice's window piece, this slice's weights and window piece)
          const int wn = n_now + n_prev;                 // 0 .. 2 PPT
          auto wait_plus = [&](auto base) {              // s_waitcnt vmcnt(base + wn): wn folds per tap, one instruction remains
            constexpr int BASE = decltype(base)::value;
            if (wn == 4) wait_vmcnt<BASE + 4>(); else if (wn == 3) wait_vmcnt<BASE + 3>(); else if (wn == 2) wait_vmcnt<BASE + 2>();
            else if (wn == 1) wait_vmcnt<BASE + 1>(); else wait_vmcnt<BASE>();
          };
#ifdef DLIP_LAB
          // lab probe (DLIP_WIN_NOWAIT=1; WRONG results, timing only): what a slice costs when nobody waits for the next slice's weights --
          // the ceiling of any deeper ring / earlier issue
          if (a.n_inner == 77) wait_vmcnt<40>(); else
#endif
          if (!first && c == 0 && tap == 0) wait_plus(std::integral_constant<int, B_PER + NSTORE>{});
          else if (moreP) wait_plus(std::integral_constant<int, B_PER>{});
          else wait_plus(std::integral_constant<int, 0>{});
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          WIN_SSTAMP(4);
          __builtin_amdgcn_s_barrier();
          WIN_SSTAMP(5);
          read_first((tap + 1) % 3);
        }
        DLIP_FENCE();
        mfma_p(2);
        DLIP_FENCE();
        WIN_SSTAMP(6);
      }
    }
#undef DLIP_FENCE
    WIN_STAMP(2);

    // ---- epilogue straight from the accumulators (round 4; the rows kernel's): y = act(acc / wscale + bias + residual) * ps + pt.
    // Until round 3 the output tile was staged in LDS as its memory image (the ring kernel's epilogue): three workgroup barriers, the
    // residual tile fetched by LDS-DMA into that image -- 1.7-3 k cycles of exposed latency on the two residual launches of every
    // BasicBlock, since no LDS was free for it before the ring and the windows were done -- 16 LDS reads per lane to add it, and the
    // image pinned where the next tile's head could not go.  Now a lane exchanges the odd 16-lane rows of one accumulator with the
    // even rows of its neighbour in the channel direction (v_permlane16_swap_b32), holds 8 consecutive channels of its pixel and
    // stores one 16-B piece of hi halves and one of lo halves; the residual arrives in the same shape by two 16-B loads per pixel
    // block, issued FIRST, so that their latency passes under the parameter reads and the other blocks' arithmetic; no barrier,
    // no LDS traffic, and the next tile's head (in flight since the last slices) is not touched.  Arithmetic and order per value are
    // the old epilogue's: same bits. ----
    {
      typedef _Float16 h4 __attribute__((ext_vector_type(4)));
      typedef _Float16 h8 __attribute__((ext_vector_type(8)));
      const f32x4* tab = reinterpret_cast<const f32x4*>(smem + TAB_OFF / 4);
      const __amdgpu_buffer_rsrc_t rr = dlip_make_rsrc(a.res, a.res ? a.r_bytes : 0u);
      const bool head_here = has_next && (cchunks & 1) != 0;   // odd channel-slice count: window 0 could not be streamed
      if (head_here) {
        __syncthreads();   // (workgroup-uniform) slot 0 held the last channel slice: every wave is done reading it
#pragma unroll
        for (int j = 0; j < WPER; ++j) issue_win(0, 0, wave + NW * j, m0n - halo_lo);
      }
      WIN_STAMP(3);
      const int row0 = m0 + wm * WM + lrow;
      if constexpr (OSPLIT) {
        const int cs = ((half & 1) << 4) | ((half & 2) << 2);   // after the swap: channels 32 p + {0, 16, 8, 24}[half] + 0..7
        h8 rh[NI / 2][MI], rl[NI / 2][MI];
        if (a.res) {
#pragma unroll
          for (int p = 0; p < NI / 2; ++p)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
              const int m = row0 + mi * FR, kb = wn * WN + 32 * p;
              const uint32_t off = (m < a.M && kb < a.K) ? (uint32_t)((m * a.ldr + kb) * 4 + cs * 2) : DLIP_OOB_OFFSET;
              rh[p][mi] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rr, (int)off, 0, 0));
              rl[p][mi] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rr, (int)(off == DLIP_OOB_OFFSET ? off : off + 64u), 0, 0));
            }
        }
        WIN_STAMP(4);
#pragma unroll
        for (int p = 0; p < NI / 2; ++p) {
          const int kb = wn * WN + 32 * p, k0 = kb + cs;
          float inv[8], bi[8], sl[8], ps[8], pt[8];
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const f32x4 i4 = tab[(k0 >> 2) + q], b4 = tab[((BN + k0) >> 2) + q], l4 = tab[((2 * BN + k0) >> 2) + q];
            f32x4 p4 = {1.f, 1.f, 1.f, 1.f}, t4 = {0.f, 0.f, 0.f, 0.f};
            if (post) { p4 = tab[((3 * BN + k0) >> 2) + q]; t4 = tab[((4 * BN + k0) >> 2) + q]; }
#pragma unroll
            for (int c = 0; c < 4; ++c) { inv[4 * q + c] = i4[c]; bi[4 * q + c] = b4[c]; sl[4 * q + c] = l4[c]; ps[4 * q + c] = p4[c]; pt[4 * q + c] = t4[c]; }
          }
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            float v[8];
#pragma unroll
            for (int c = 0; c < 4; ++c) {   // (scalar copies: see conv_rows_f16x3.hip on __builtin_bit_cast of a vector element)
              const float xc = acc[mi][2 * p][c], yc = acc[mi][2 * p + 1][c];
              const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(xc), __float_as_uint(yc), false, false);
              v[c] = __uint_as_float(r[0]);
              v[4 + c] = __uint_as_float(r[1]);
            }
            h8 hi, lo;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
              float t = v[c] * inv[c] + bi[c];
              if (a.res) t += (float)rh[p][mi][c] + (float)rl[p][mi][c];
              t = t >= 0.f ? t : t * sl[c];
              if (post) t = t * ps[c] + pt[c];
              hi[c] = (_Float16)t;
              lo[c] = (_Float16)(t - (float)hi[c]);
              amax = fmaxf(amax, fabsf(t));
            }
            const int m = row0 + mi * FR;
            const bool ok = m < a.M && kb < a.K;
            const uint32_t off = ok ? (uint32_t)((m * a.ldy + kb) * 4 + cs * 2) : DLIP_OOB_OFFSET;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hi), yr, (int)off, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, lo), yr, (int)(ok ? off + 64u : DLIP_OOB_OFFSET), 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
        WIN_STAMP(4);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const int kl = wn * WN + ni * FR + 4 * half;
          const f32x4 inv4 = tab[kl >> 2], bi4 = tab[(BN + kl) >> 2], sl4 = tab[(2 * BN + kl) >> 2];
          f32x4 ps4 = {1.f, 1.f, 1.f, 1.f}, pt4 = {0.f, 0.f, 0.f, 0.f};
          if (post) { ps4 = tab[(3 * BN + kl) >> 2]; pt4 = tab[(4 * BN + kl) >> 2]; }
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            const int m = row0 + mi * FR;
            const bool ok = m < a.M && kl < a.K;
            f32x4 v;
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = acc[mi][ni][c] * inv4[c] + bi4[c];
            if (a.res) {   // the residual is in the split format: 4 hi halves, 4 lo halves of this lane's channels
              const uint32_t ro = ok ? (uint32_t)((m * a.ldr + (kl & ~31)) * 4 + (kl & 31) * 2) : DLIP_OOB_OFFSET;
              const h4 rh = __builtin_bit_cast(h4, __builtin_amdgcn_raw_buffer_load_b64(rr, (int)ro, 0, 0));
              const h4 rl = __builtin_bit_cast(h4, __builtin_amdgcn_raw_buffer_load_b64(rr, (int)(ok ? ro + 64u : DLIP_OOB_OFFSET), 0, 0));
#pragma unroll
              for (int c = 0; c < 4; ++c) v[c] += (float)rh[c] + (float)rl[c];
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              v[c] = v[c] >= 0.f ? v[c] : v[c] * sl4[c];
              if (post) v[c] = v[c] * ps4[c] + pt4[c];
            }
            const uint32_t off = ok ? (uint32_t)((m * a.ldy + kl) * 4) : DLIP_OOB_OFFSET;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yr, (int)off, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      WIN_STAMP(5);
      WIN_STAMP(6);
      if (has_next) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) fr_mask[mi] = ~next_mask[mi];
        m0 = m0n;
      }
    }
  }
  if constexpr (OSPLIT) dlip_report_range(amax, a.status);
  dlip_span_exit(a.span);
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int OCC>
int launch_win(const ConvArgs& a, hipStream_t st, bool out_split) {
  ConvArgs b = a;
  const int tiles_m = (a.M + BM - 1) / BM;
  b.tiles_n = (a.K + BN - 1) / BN;
  const long long tiles = (long long)tiles_m * b.tiles_n;
  if (tiles <= 0 || tiles > 0x7FFFFFFFll) return DLIP_EINVAL;
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int WBLK0 = (BM + WIN_SLACK + 15) / 16;
  constexpr int WBLK = (WBLK0 + NW / 2 - 1) / (NW / 2) * (NW / 2);
  constexpr size_t slot_b = (size_t)WBLK * 2048, bst = (size_t)BN * ROWB;
  constexpr size_t head = slot_b + bst;
  constexpr size_t lds = head + 2 * bst + slot_b + 5 * BN * sizeof(float);
  if (b.tiles_n != 1) return DLIP_EINVAL;   // one column block (K <= BN): the kernel's tile index is the row block
  static_assert(lds <= 160 * 1024, "LDS exceeds a CU");
  auto kern = out_split ? conv_win_f16x3_kernel<BM, BN, WAVES_M, WAVES_N, true, OCC> : conv_win_f16x3_kernel<BM, BN, WAVES_M, WAVES_N, false, OCC>;
  static DlipKernelState state[2];      // per instance (plain | split output); per-device inside
  DlipKernelState& ks = state[out_split ? 1 : 0];
  if (lds > 64 * 1024) {
    const int e = ks.ensure_lds(reinterpret_cast<const void*>(kern), lds);
    if (e != DLIP_OK) return e;
  }
  // persistent workgroups: as many as the device holds at once (2 per CU), each walking a contiguous range of tiles
  int slots = 0;
  {
    const int e = ks.resident(reinterpret_cast<const void*>(kern), 64 * NW, lds, &slots);
    if (e != DLIP_OK) return e;
  }
  const long long grid = tiles < slots ? tiles : slots;
  b.span = dlip_span_next();
#ifdef DLIP_LAB
  b.n_inner = getenv("DLIP_WIN_NOWAIT") ? 77 : 0;
  if (getenv("DLIP_STAMP_PRINT")) {   // median cycles between the phase stamps of every workgroup's second tile
    static unsigned long long* dbuf = nullptr;
    if (!dbuf) (void)hipMalloc(reinterpret_cast<void**>(&dbuf), 2 * 4096 * 8 * 8);
    (void)hipMemsetAsync(dbuf, 0, 2 * 4096 * 8 * 8, st);
    b.pool = reinterpret_cast<double*>(dbuf);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * NW), lds, st, b);
    (void)hipStreamSynchronize(st);
    std::vector<unsigned long long> h((size_t)grid * 8);
    (void)hipMemcpy(h.data(), dbuf, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> d[6];
    for (long long i = 0; i < grid; ++i)
      if (h[i * 8 + 6]) for (int j = 0; j < 6; ++j) d[j].push_back((double)(h[i * 8 + j + 1] - h[i * 8 + j]));
    auto med = [](std::vector<double>& v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    fprintf(stderr, "[win stamps %dx%d M=%d nk=%d res=%d] head-wait %.0f  loop %.0f  issue-next+masks %.0f  residual-wait %.0f  epilogue-compute %.0f  stores %.0f\n",
            BM, BN, b.M, b.nk, b.res != nullptr, med(d[0]), med(d[1]), med(d[2]), med(d[3]), med(d[4]), med(d[5]));
    (void)hipMemcpy(h.data(), dbuf + 4096 * 8, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> e[6];
    for (long long i = 0; i < grid; ++i)
      if (h[i * 8 + 6]) for (int j = 0; j < 6; ++j) e[j].push_back((double)(h[i * 8 + j + 1] - h[i * 8 + j]));
    fprintf(stderr, "[win slice (c 0, tap 4)] piece issue %.0f  rest reads + group 0 %.0f  addresses + group 1 %.0f  vmcnt + lgkmcnt %.0f  barrier %.0f  first reads + group 2 %.0f\n",
            med(e[0]), med(e[1]), med(e[2]), med(e[3]), med(e[4]), med(e[5]));
    return dlip_launch_status();
  }
#endif
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * NW), lds, st, b);
  return dlip_launch_status();
}

}  // namespace

// The window kernel serves same-size stride-1 3x3 convolutions (the nine taps are unrolled; a wave's window pieces go
// out one per slice and must be older in the queue than the weights of the next channel slice's first tap), a halo within WIN_SLACK,
// K <= 128 (ONE column block: layer 1 on the 128x64 instance, layer 2 on the 128x128 one; dlip_debug_set(DLIP_DBG_WIN, 1)
// keeps it to K <= 64) and no second reduction source / pooled epilogue.
static bool win_shape_ok(int sh, int sw, int H, int W, int Ho, int Wo, int R, int S, int dh, int dw, int ph, int pw, int K) {
  if (dlip_dbg_value[DLIP_DBG_WIN] == 0) return false;
  // masked taps read past the workgroup's LDS allocation and rely on the hardware returning zeros: probed on gfx950
  // (tests/test_kernels_gpu.py::test_lds_read_beyond_allocation_returns_zero), never assumed elsewhere
  if (!dlip_device_is_gfx950()) return false;
  return sh == 1 && sw == 1 && Wo == W && Ho == H && R == 3 && S == 3 && K <= (dlip_dbg_value[DLIP_DBG_WIN] == 1 ? 64 : 128) && (K & 3) == 0 &&
         (R - 1) * dh * W + (S - 1) * dw <= WIN_SLACK && ph * W + pw <= WIN_SLACK;
}

extern "C" __attribute__((visibility("hidden"))) int dlip_conv_win_ok(const void* args) {
  const ConvArgs& a = *static_cast<const ConvArgs*>(args);
  return a.x2 == nullptr && a.pool == nullptr &&
         win_shape_ok(a.sh, a.sw, a.H, a.W, a.HoWo / a.Wo, a.Wo, a.R, a.S, a.dh, a.dw, a.ph, a.pw, a.K);
}

// ---- the hardware behaviour the masked taps stand on, as a callable check (tests/test_kernels_gpu.py) ----
// A tap outside the image reads its fragment at an LDS address with bit 18 set -- beyond any workgroup's allocation and beyond
// the CU's 160 KiB -- and the kernel takes the zeros gfx950 returns there as the convolution's padding.  This probe runs
// that access pattern under the kernel's conditions (several workgroups per CU, each with its own allocation full of a
// non-zero pattern, ds_read_b128 at base + 2^18 + lane offsets) and counts what comes back.
namespace {
__global__ __launch_bounds__(256) void lds_oob_probe_kernel(int32_t* counts, int alloc_bytes) {
  extern __shared__ __attribute__((aligned(16))) uint32_t probe_lds[];
  for (int i = threadIdx.x; i < alloc_bytes / 4; i += blockDim.x) probe_lds[i] = 0xABCD0000u + blockIdx.x;
  __syncthreads();
  const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)probe_lds;
  const uint32_t offs[4] = {0u, 1u << 18, (1u << 18) + 40000u, (1u << 18) + 130000u};
  bool in_ok = false, oob_nonzero = false;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t addr = base + offs[k] + (threadIdx.x * 16u) % (uint32_t)alloc_bytes;
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    if (k == 0) in_ok = (v[0] >> 16) == 0xABCDu && (v[3] >> 16) == 0xABCDu;
    else oob_nonzero |= (v[0] | v[1] | v[2] | v[3]) != 0u;
  }
  if (in_ok) atomicAdd(&counts[0], 1);
  if (oob_nonzero) atomicAdd(&counts[1], 1);
}
}  // namespace

extern "C" int dlip_selftest_lds_oob(int32_t* counts, int32_t blocks, dlip_stream_t stream) {
  DLIP_CHECK_ARG(counts && blocks > 0 && blocks <= 65536);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(counts, 0, 2 * sizeof(int32_t), st) != hipSuccess) return DLIP_EINVAL;
  hipLaunchKernelGGL(lds_oob_probe_kernel, dim3((unsigned)blocks), dim3(256), 8192, st, counts, 8192);
  return dlip_launch_status();
}

extern "C" int dlip_conv_dma_enabled(void);   // conv_igemm_f16x3.hip

// Which kernel a split-format launch of `d` goes to (include/deeplip_hip.h): 1 = this one.
extern "C" int dlip_conv_rows_plan(const dlip_conv_desc* d, int* bm);   // conv_rows_f16x3.hip
extern "C" int dlip_conv_rows2d_plan(const dlip_conv_desc* d, int c2, int* bm);
extern "C" int dlip_conv_kernel_kind(const dlip_conv_desc* d) {
  if (!d) return DLIP_EINVAL;
  if (d->C % 32 == 0 && dlip_conv_dma_enabled() &&
      win_shape_ok(d->stride_h, d->stride_w, d->H, d->W, d->Ho, d->Wo, d->R, d->S, d->dil_h, d->dil_w, d->pad_h, d->pad_w, d->K))
    return 1;
  return (dlip_conv_rows_plan(d, nullptr) || dlip_conv_rows2d_plan(d, 0, nullptr)) ? 2 : 0;
}

extern "C" __attribute__((visibility("hidden"))) int dlip_conv_f16x3_win_launch(const void* args, void* stream, int out_split) {
  const ConvArgs& a = *static_cast<const ConvArgs*>(args);
  // K <= 64: four waves, two workgroups per CU (eight waves in one workgroup per CU: 258 vs 218 us on layer 1, same box).
  // 64 < K <= 128 (layer 2, 128 -> 128 channels on 11x11 maps): eight waves on a 128x128 tile, one workgroup per CU (the
  // fp32 epilogue image alone is 64 KB): 178 us vs 200-210 on the 256x128 ring kernel without a residual, 211-223 vs 241-275 with.
  // dlip_debug_set(DLIP_DBG_WIN, v) picks a geometry for same-box A/B runs: 3 = 128x64 tile, TWO waves of 64x64 (two workgroups
  // per CU); 4 = 256x64, four waves of 64x64; 5 = (K > 64) 128x128, four waves of 64x64; 6 = (K > 64) 256x128, eight waves of 64x64;
  // 7 = round 5's 2 x 2 waves of 64x32 on the 128x64 tile.
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int v = dlip_dbg_value[DLIP_DBG_WIN];
  if (a.K > 64) {
    if (v == 5) return launch_win<128, 128, 2, 2, 1>(a, st, out_split != 0);
    if (v == 6) return launch_win<256, 128, 4, 2, 1>(a, st, out_split != 0);
    return launch_win<128, 128, 4, 2, 2>(a, st, out_split != 0);
  }
  // (256x64 with eight waves in ONE workgroup per CU -- the weights fetched once per 256 rows instead of per 128 -- measured
  // 9-20 % slower than the two independent 128x64 workgroups: 239-249 vs 219-225 us, with a residual 348-357 vs 292-297)
  if (v == 3) return launch_win<128, 64, 2, 1, 1>(a, st, out_split != 0);
  if (v == 4) return launch_win<256, 64, 4, 1, 1>(a, st, out_split != 0);
  return launch_win<128, 64, 2, 2, 2>(a, st, out_split != 0);
}

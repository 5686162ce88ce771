// Split-fp16 convolution, ROWS variant: convolutions whose every tap is a plain row offset of the activation matrix --
// the 1-D "valid" convolutions of the speech encoder (models/audio_models/tdnn.py:23-43: Conv1d over time, no padding,
// dilation from the context list) and, with one tap, its k = 1 layers: plain [M x C] . [C x K] GEMMs over the B * T' frames.
//
// Why a third kernel.  On the ring kernel (conv_igemm_f16x3_dma.hip) the k = 1 layers -- 16 slices of 32 channels per
// tile -- ran at 0.21 of the split-fp16 ceiling for three rounds: a third of every 128 x 128 tile was set-up, first fills
// and epilogue around a 16-slice loop (in-kernel stamps: 7.8 k + 9.9 k of 52 k cycles), and 592 tiles on 512 slots made a
// second round that was 16 % full.  What such a layer needs is (1) more matrix work behind every barrier and every byte,
// (2) one round, (3) nothing between two tiles.  So:
//   * tile = (32 MI) x 256, MI = 5 | 4 | 3: eight waves (2 x 4), a wave owns 16 MI x 64 outputs = 4 MI accumulator quads
//     (80 registers at MI = 5).  Per 32-channel slice a workgroup moves (160 + 256) x 128 B = 52 KB for 60 MFMAs per
//     wave (the 256 x 128 ring tile: 48 KB for 48); the host picks MI so that rows x column blocks fill ONE round of the
//     chip's CUs where it can (B = 64, T' = 296: 119 x 2 = 238 tiles of 160 x 256 on 256 CUs).
//   * ONE persistent workgroup per CU walks its tiles with a CONTINUOUS slice stream: "two slices ahead" of a tile's last
//     slices are the next tile's first ones, the LDS ring never drains, and no barrier joins the waves between tiles.
//   * the epilogue goes from the accumulator registers straight to memory -- no LDS image, so the ring is not touched
//     (that is what let the stream run on; the ring kernel stages its output tile in the ring).  A split-format output
//     needs 16-B pieces: v_permlane16_swap_b32 exchanges the odd 16-lane rows of one accumulator with the even rows of
//     its neighbour in the channel direction, after which a lane holds 8 consecutive channels of its pixel = one 16-B
//     piece of hi halves and one of lo halves.  Each half of the waves runs a tile's epilogue at the head of its next LOAD
//     interval, i.e. beside its SIMD partner's matrix phase.
//   * main loop: the ring kernel's PING-PONG (its 256 x 128 instance): waves 0-3 and 4-7 -- one wave of every SIMD in each
//     half -- work on the same slice half a period apart, two barriers per slice; one half multiplies (60 MFMAs, nothing
//     else in its stream) while the other reads the 18 fragments of its next matrix phase, issues its LDS-DMA pieces
//     two slices ahead and waits for the pieces of the next slice.
// Arithmetic, slice order (channel slice outer, tap inner), product order per accumulator (lo*hi, hi*hi, hi*lo) and
// epilogue formula are the ring kernel's: a launch that the ring kernel runs as a plain (un-split) launch gives the
// same bits here (tests/test_kernels_gpu.py::test_conv_rows_*).
#include "conv_common.h"
#include "conv_dma_common.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

// conv_igemm_f16x3_dma.hip: the per-stream workspace of the balanced split (slabs + ticket words); 0 = none / too small
extern "C" int dlip_conv_split_workspace(void* stream, size_t slab_floats, float** slabs, int** counters, int* counter_words);

namespace {

struct RowsSched {
  int items;      // tiles of the launch: tiles_m * tiles_n, row block major (the column blocks of a row block are neighbours)
  int tiles_n;
  unsigned long long* span;   // NULL, or this launch's {first start, last end} in 100 MHz ticks (dlip_span_scope_*)
  // MODE 1 (balanced split, the ring kernel's stream-K): the items * nk slices of the launch cut into G equal ranges, one per
  // workgroup; a tile shared by several ranges goes through fp32 slabs (two per workgroup: its first and its last segment) and
  // ticket words (8 per tile: one per wave position) in the ring kernel's per-stream workspace
  int iters;      // items * nk; iters * G < 2^31 (32-bit arithmetic and magic-number division throughout: the 64-bit quotients'
  int G;          // expansion left the cursor state in vector registers and on the stack)
  FastDiv div_G, div_iters, div_nk;
  float* slabs;
  int* counters;
  // MODE 0 (round 5), the SHORT LAST ROUND: row tiles t_full.. (items item_short0..) are 32 * nmi_short rows high instead of 32 MI, so
  // that the rows left over after the launch's full rounds of tiles spread over ALL workgroups as one round of shorter tiles (B = 256:
  // 948 tiles of 160 rows on 256 CUs = 3 full rounds + a fourth that is 70 % full -> 3 rounds + one of 128-row tiles: 3.8 tile
  // times instead of 4).  A short tile keeps the LDS image of a full one: its wave row w holds rows w * 16 nmi_short .. of the tile
  // in LDS rows w * 16 MI ..; the rest of the image is not fetched (out-of-range offsets), not multiplied and not stored.
  // No short tiles: item_short0 = t_full = INT_MAX.
  int item_short0, t_full, nmi_short, row_short0;
#ifdef DLIP_LAB
  unsigned long long* stamps;
#endif
};

constexpr int ROWS_BN = 256;

// The kernel's ConvArgs as the kernarg segment holds it (first explicit argument: offset 0), behind an opaque asm: what is read
// through this pointer is (re)loaded where it is used -- scalar loads, once per tile -- instead of being kept in scalar registers
// for the whole kernel.  The general mode's tile set-up and epilogue need ~45 argument words and seven buffer descriptors between
// them; held across the main loop they spilled ~300 scalar registers into vector lanes and, from there, loop state into scratch.
typedef const ConvArgs __attribute__((address_space(4))) RowsKArgs;
__device__ __forceinline__ RowsKArgs* rows_kargs() {
  RowsKArgs* p = (RowsKArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return p;
}

// ... and the RowsSched behind it (second explicit argument: at the next 8-byte boundary)
typedef const RowsSched __attribute__((address_space(4))) RowsKSched;
__device__ __forceinline__ RowsKSched* rows_ksched() {
  typedef const char __attribute__((address_space(4))) KChar;
  KChar* p = (KChar*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return (RowsKSched*)(p + ((sizeof(ConvArgs) + 7) & ~(size_t)7));
}

// EPI: 0 fp32 rows of y; 1 split-format rows of y (reports range); 2 no y: per HALF tile (the 16 MI rows of a wave row) and
// row-group segment the fp64 column sums of v and v^2 (a.pool, the ring kernel's pooled epilogue with tile rows = 16 MI)
// MODE: 0 the speech encoder's form (file comment): every tap a plain row offset, whole tiles per workgroup, tiles g, g + G, ...
//       1 (round 4) the trunk's deep layers on the same loop: 2-D filters with padding / stride / dilation (a bit per tap and row,
//         ORed into the piece's offset as bit 31 -- the ring kernel's), an optional residual in the split format (two 16-B loads
//         per pixel block, the window kernel's), the ring kernel's BALANCED SPLIT -- every workgroup one contiguous range of the
//         launch's tiles x slices -- and, DUAL, its second reduction source (dlip_conv2_nhwc_f16x3).  The hand-off of a shared
//         tile is PER WAVE: the ping-pong halves never meet at a common barrier, and wave w of the finisher needs exactly the
//         accumulators of wave w of the other parts -- so each wave publishes its own quads (sc1 stores, vmcnt(0), one relaxed
//         ticket on the tile's word for its position) and the wave that finds its position complete adds the others' in part
//         order and runs the epilogue; different waves of a tile may finish in different workgroups, the bits do not depend on it.
// TAIL: the launch has a short last round (RowsSched); instances without it carry none of its code.
template <int MI, int EPI, int MODE = 0, bool DUAL = false, bool TAIL = false>
__global__ __launch_bounds__(512, 2) void conv_rows_f16x3_kernel(const ConvArgs a, const RowsSched sc) {
  static_assert(MODE == 1 || !DUAL, "the second source exists in the general mode only");
  static_assert(!TAIL || (MODE == 0 && EPI != 2), "short tiles: the speech encoder's form, fp32 or split output");
  // (TAIL) the four schedule words are re-read from the kernel-argument segment (scalar loads) where a tile begins or ends: held in
  // scalar registers across the main loop they cost the split-output instances 40 SGPR spills and the B = 64 step 1 % (the
  // paired-lanes lesson again)
#define ROWS_SC(f) (rows_ksched()->f)
  static_assert(MODE == 0 || EPI != 2, "no pooled epilogue in the general mode");
  constexpr int NW = 8, BN = ROWS_BN, BM = 32 * MI, NI = 4;
  constexpr int WM = BM / 2;                       // rows of a wave's tile (16 MI); WN = 64
  constexpr int A_PIECES = BM / 8, B_PER = 4;      // 1-KiB pieces of a slice's activation rows; weight pieces per wave
  constexpr int NA_A = (A_PIECES + 7) / 8, NA_B = A_PIECES / 8;   // activation pieces of a wave of half A (waves 0-3) / B (4-7)
  static_assert(A_PIECES % 8 == 0 || A_PIECES % 8 == 4, "waves 0-3 take one activation piece more than waves 4-7, or the same");
  constexpr int STAGE_B = (BM + BN) * ROWB, NSTAGE = 3;
  constexpr int LDK = 32;
  // output stores per lane and tile that the counted waits behind an epilogue may leave in flight: a LOWER bound of what the
  // epilogue issues (a larger number would let the wait return before the next slice's pieces have landed).  The pooled
  // epilogue's few stores are issued under lane masks: not counted, i.e. waited for.
  constexpr int NST = EPI == 1 ? MI * 4 : EPI == 0 ? MI * NI : 0;
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int g = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);   // neighbours in tile order share an XCD (L2)
  const int nk = a.nk;
  int total, item0 = g, k0 = 0;                           // slices of this workgroup's stream; its first tile and slice
  if constexpr (MODE == 1) {
    const int it_begin = dlip_div(g * sc.iters, sc.div_G);
    total = __builtin_amdgcn_readfirstlane(dlip_div((g + 1) * sc.iters, sc.div_G) - it_begin);
    if (total <= 0) return;
    item0 = __builtin_amdgcn_readfirstlane(dlip_div(it_begin, sc.div_nk));
    k0 = __builtin_amdgcn_readfirstlane(it_begin - item0 * nk);
  } else {
    if (g >= sc.items) return;
    total = ((sc.items - g + nwg - 1) / nwg) * nk;        // this workgroup's tiles: g, g + nwg, ...
  }

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const bool half_b = wave >= NW / 2;                     // (wave-uniform)
  const int wm = wave >> 2, wn = wave & 3;
  const int lrow = lane & 15, hq = lane >> 4;
  // DMA side: piece q = wave + 8 j covers tile rows 8 q .. 8 q + 7; lane -> (row lane >> 3, chunk position lane & 7); the XOR
  // swizzle of the LDS image (chunk p of row r at position p ^ ((r >> 1) & 7)) goes on the SOURCE chunk, and since
  // 4 q mod 8 is the same for every piece of a wave the key is a per-lane constant.
  const int prow = lane >> 3;
  const int key_st = (4 * (wave & 1) + (lane >> 4)) & 7;
  const int csrc = ((lane & 7) ^ key_st) << 2;            // first channel (dword) of the chunk this lane fetches
  const u32x4 xr = make_rsrc_words(a.x, a.x_bytes);
  const u32x4 wr = make_rsrc_words(a.w, a.w_bytes);
  u32x4 x2r = xr;
  if constexpr (DUAL) x2r = make_rsrc_words(a.x2, a.x2_bytes);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const uint32_t piece0 = lds0 + wave * 1024;
  // fragment side (v_mfma_f32_16x16x32_f16; the weight fragment is the A operand: accumulators hold the transposed tile)
  const int a_frag = (wm * WM + lrow) * LDK;
  const int b_frag = BM * LDK + (wn * 64 + lrow) * LDK;
  const int key_rd = (lrow >> 1) & 7;
  const int khi = (hq ^ key_rd) << 2, klo = ((4 + hq) ^ key_rd) << 2;
  const int x_ds = a.dw * a.ldx * 4;                      // bytes between two taps of a row
  const int x_dr = a.dh * a.W * a.ldx * 4;                // (MODE 1) ... between two filter rows
  const int ntaps = a.R * a.S;
  const int nk1 = nk - (DUAL ? a.nk2 : 0);                // slices of the first source

  dlip_span_enter(sc.span, g);
#ifdef DLIP_LAB
#define ROWS_STAMP(i) do { if (sc.stamps && tid == 0) sc.stamps[(size_t)g * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define ROWS_SSTAMP(i) do { if (sc.stamps && tid == 0 && s == 8) sc.stamps[(size_t)g * 16 + 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
  if (sc.stamps && tid == 0) sc.stamps[(size_t)g * 16 + 7] = __builtin_amdgcn_s_memrealtime();
#else
#define ROWS_STAMP(i) do { } while (0)
#define ROWS_SSTAMP(i) do { } while (0)
#endif

  // ---- issue cursor: the slice the NEXT pieces belong to (two slices ahead of the one being multiplied) ----
  // (rows past M / weight rows past K carry the out-of-range offset itself: 2^31 plus any in-range tap offset is still beyond every
  // buffer, so a piece's address is ONE add -- the select per piece was two of every three vector instructions of a slice)
  uint32_t a_off[NA_A], b_off[B_PER];
  uint32_t a_mask[MODE == 1 ? NA_A : 1];                  // (MODE 1) INVERTED: bit t set = tap t of that row is outside the image, or the row past M
  uint32_t a2_off[DUAL ? NA_A : 1];                       // (DUAL) the row's pixel of the second source; out of range past M
  int c_item = item0, c_k = k0, s_pos = 0, c0 = 0, x_tap = 0, w_tap = 0;
  int tap = 0, x_row = 0;                                 // (MODE 1) tap index r S + s and the byte offset of filter row r
  auto set_item = [&](int item) __attribute__((always_inline)) {
    const int tile_m = item / sc.tiles_n, tile_n = item - tile_m * sc.tiles_n;
    [[maybe_unused]] RowsKArgs* ap = nullptr;
    if constexpr (MODE == 1) ap = rows_kargs();
    [[maybe_unused]] int t_base = tile_m * BM, t_half = WM;   // (MODE 0) the tile's first row, rows per wave row (RowsSched: short last round)
    if constexpr (TAIL) {
      if (item >= ROWS_SC(item_short0)) { t_half = 16 * ROWS_SC(nmi_short); t_base = ROWS_SC(row_short0) + (tile_m - ROWS_SC(t_full)) * 2 * t_half; }
    }
#pragma unroll
    for (int j = 0; j < NA_A; ++j) {
      const int m = tile_m * BM + 8 * (wave + 8 * j) + prow;
      if constexpr (MODE == 1) {
        const int mc = m < ap->M ? m : ap->M - 1;
        const int n = dlip_div(mc, FastDiv{ap->div_howo.mul, ap->div_howo.shift});
        const int rem = mc - n * ap->HoWo;
        const int ho = dlip_div(rem, FastDiv{ap->div_wo.mul, ap->div_wo.shift});
        const int wo = rem - ho * ap->Wo;
        const int hi0 = ho * ap->sh - ap->ph, wi0 = wo * ap->sw - ap->pw;   // window origin (may lie in the padding: the sum with a valid tap does not)
        a_off[j] = (uint32_t)((((n * ap->H + hi0) * ap->W + wi0) * ap->ldx + csrc) * 4);
        if constexpr (DUAL) a2_off[j] = m < ap->M ? (uint32_t)((((n * ap->H2 + ho * ap->s2h) * ap->W2 + wo * ap->s2w) * ap->ldx2 + csrc) * 4) : DLIP_OOB_OFFSET;
        uint32_t colbits = 0u, ok = 0u;                   // columns and rows tested separately: R + S steps, not R x S
        for (int sx = 0; sx < ap->S; ++sx) colbits |= (uint32_t)((unsigned)(wi0 + sx * ap->dw) < (unsigned)ap->W) << sx;
        for (int r = 0; r < ap->R; ++r) ok |= ((unsigned)(hi0 + r * ap->dh) < (unsigned)ap->H ? colbits : 0u) << (r * ap->S);
        a_mask[j] = m < ap->M ? ~ok : ~0u;
      } else {
        const int lr = 8 * (wave + 8 * j) + prow;            // row of the LDS image; hr: within its wave row
        const int hr = lr >= WM ? lr - WM : lr;
        const int ms = t_base + (lr >= WM ? t_half : 0) + hr;
        const bool in = hr < t_half && ms < a.M;
        const int mc = in ? ms : 0;
        const int n = dlip_div(mc, a.div_howo);
        const int t = mc - n * a.HoWo;                       // H = 1: the row's first input pixel is n * W + t
        a_off[j] = in ? (uint32_t)(((n * a.W + t) * a.ldx + csrc) * 4) : DLIP_OOB_OFFSET;
      }
    }
#pragma unroll
    for (int j = 0; j < B_PER; ++j) {
      const int n = tile_n * BN + 8 * (wave + 8 * j) + prow;
      b_off[j] = n < a.K ? (uint32_t)((n * a.rsc + csrc) * 4) : DLIP_OOB_OFFSET;
    }
  };
  auto place = [&]() __attribute__((always_inline)) {     // byte offsets of the slice the cursor stands on
    if constexpr (MODE == 1) {
      x_tap = x_row + s_pos * x_ds + c0 * 4;
      w_tap = (tap * a.Cw + c0) * 4;
      if (DUAL && c0 >= a.Cw) { x_tap = (c0 - a.Cw) * 4; w_tap = (ntaps * a.Cw + c0 - a.Cw) * 4; }   // behind the taps of each weight row
    } else {
      x_tap = s_pos * x_ds + c0 * 4;
      w_tap = (s_pos * a.Cw + c0) * 4;
    }
  };
  auto advance = [&]() __attribute__((always_inline)) {   // channel slice outer, tap inner (the ring kernel's reduction order)
    if constexpr (MODE == 1) {
      // (value selects on local copies, one store per variable: as an if / else chain storing to the cursor variables, the
      // optimiser merged the branches' stores into stores through a pointer phi and the whole cursor stayed on the stack)
      const bool second = DUAL && c0 >= a.Cw;            // second source: one slice per 32 channels
      int ntap = tap + 1, nsp = s_pos + 1, nrow = x_row, nc0 = c0, nk_ = c_k + 1, nitem = c_item;
      if (nsp == a.S) { nsp = 0; nrow += x_dr; }
      if (ntap == ntaps) { ntap = 0; nsp = 0; nrow = 0; nc0 += BK; }
      if (second) { ntap = tap; nsp = s_pos; nrow = x_row; nc0 = c0 + BK; }
      if (nk_ == nk) { nk_ = 0; ntap = 0; nsp = 0; nrow = 0; nc0 = 0; ++nitem; }   // (the caller has run set_item for the next tile: enter_next)
      c_k = nk_; tap = ntap; s_pos = nsp; x_row = nrow; c0 = nc0; c_item = nitem;
      place();
    } else {
      if (++c_k == nk) {
        c_k = 0; s_pos = 0; c0 = 0;
        c_item += nwg;
        set_item(c_item);
      } else if (++s_pos == a.S) {
        s_pos = 0; c0 += BK;
      }
      x_tap = s_pos * x_ds + c0 * 4;
      w_tap = (s_pos * a.Cw + c0) * 4;
    }
  };
  // MODE 1: the row set-up of the NEXT tile (two divisions and R + S mask steps per row) is done apart from advance(), at a point
  // of the caller's choosing -- in the loop BEFORE the slice's fragment reads: behind them it sat on top of 72 live fragment
  // registers and pushed loop state into scratch
  auto enter_next = [&]() __attribute__((always_inline)) {
    if constexpr (MODE == 1) { if (c_k + 1 == nk) set_item(c_item + 1); }
  };
  auto issue = [&](int stage, auto na) __attribute__((always_inline)) {
    const uint32_t base = piece0 + stage * STAGE_B;
    if (DUAL && c0 >= a.Cw) {                              // (wave-uniform)
#pragma unroll
      for (int j = 0; j < na(); ++j)
        dma_piece(x2r, a2_off[DUAL ? j : 0] + (uint32_t)x_tap, base + j * 8192);
    } else {
#pragma unroll
      for (int j = 0; j < na(); ++j) {
        if constexpr (MODE == 1) {
          // a tap outside the image: its bit of the inverted mask, shifted to bit 31, ORed into the offset -- beyond every buffer
          const uint32_t oob = (a_mask[j] << (31 - tap)) & 0x80000000u;
          dma_piece(xr, (a_off[j] + (uint32_t)x_tap) | oob, base + j * 8192);
        } else {
          dma_piece(xr, a_off[j] + (uint32_t)x_tap, base + j * 8192);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < B_PER; ++j)
      dma_piece(wr, b_off[j] + (uint32_t)w_tap, base + BM * ROWB + j * 8192);
  };

  f32x4 acc[MI][NI];
  f16x8 fal[MI], fah[MI], fbh[NI], fbl[NI];
#define DLIP_FENCE() __builtin_amdgcn_sched_barrier(0)
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[mi][ni][e] = 0.f;
  };
  auto read_all = [&](int stage) __attribute__((always_inline)) {
    const float* Aw = smem + stage * (STAGE_B / 4) + a_frag;
    const float* Bw = smem + stage * (STAGE_B / 4) + b_frag;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) fal[mi] = *reinterpret_cast<const f16x8*>(Aw + mi * 16 * LDK + klo);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) fbh[ni] = *reinterpret_cast<const f16x8*>(Bw + ni * 16 * LDK + khi);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) fah[mi] = *reinterpret_cast<const f16x8*>(Aw + mi * 16 * LDK + khi);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) fbl[ni] = *reinterpret_cast<const f16x8*>(Bw + ni * 16 * LDK + klo);
  };
  auto mfma_all = [&](int nmi) __attribute__((always_inline)) {   // nmi (wave-uniform): 16-row blocks of this wave's rows that exist
    __builtin_amdgcn_s_setprio(2);   // the matrix phase outranks its SIMD partner's load phase
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      // (TAIL) a tile of the short last round skips the blocks it does not have: one scalar compare-and-branch per block
      if (!TAIL || mi < nmi) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {   // accumulator-major: the three products of one accumulator back to back (ring kernel)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fbh[ni], fal[mi], acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fbh[ni], fah[mi], acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fbl[ni], fah[mi], acc[mi][ni], 0, 0, 0);
        }
      }
      DLIP_FENCE();
    }
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- epilogue of one tile, from this wave's accumulators straight to memory:
  //      y = act(acc / wscale + bias [+ residual]) * post_scale + post_shift   (a residual in MODE 1 only; split format) ----
  // (MODE 0: the descriptors are built once, here -- plain scalars, as the compiler keeps them best; MODE 1: inside the epilogue,
  // re-read through rows_kargs().  An aggregate carrying the seven descriptors into the epilogue cost the speech-encoder layers
  // 2 - 4 %: it lived in vector registers and came back through v_readfirstlane + hazard s_nops, same-box A/B at B = 256.)
  const uint32_t kbytes0 = (uint32_t)a.K * 4u;
  const __amdgpu_buffer_rsrc_t yr0 = dlip_make_rsrc(a.y, a.y_bytes);
  const __amdgpu_buffer_rsrc_t scr0 = dlip_make_rsrc(a.wscale, kbytes0);
  const __amdgpu_buffer_rsrc_t bir0 = dlip_make_rsrc(a.bias, a.bias ? kbytes0 : 0u);
  const __amdgpu_buffer_rsrc_t slr0 = dlip_make_rsrc(a.slope, a.slope ? kbytes0 : 0u);
  const __amdgpu_buffer_rsrc_t psr0 = dlip_make_rsrc(a.pscale, a.pscale ? kbytes0 : 0u);
  const __amdgpu_buffer_rsrc_t ptr0 = dlip_make_rsrc(a.pshift, a.pshift ? kbytes0 : 0u);
  const bool has_slope0 = a.slope != nullptr;
  const bool has_post0 = a.pscale != nullptr;
  float amax = 0.f;
  // (round 5) MODE 0, fp32 / split output: the epilogue's per-channel parameters {1 / wscale, bias, slope} of this workgroup's column
  // block, 3 KB of LDS behind the ring, written once.  A workgroup's tiles g, g + G, ... share their column block whenever the column
  // blocks divide G (K = 512: two blocks on 256 workgroups), and the epilogue took them from memory per tile: two dependent L2 round
  // trips (~2.4 k cycles) at the head of the interval in which its SIMD partner multiplies for 1.1 k -- in-kernel stamps at B = 256:
  // 16 x 2 632 cycles of slices per k = 1 tile, but 60.6 k per tile, the difference being the two halves' epilogues each stalling the
  // other.  A tile of another column block (tab_n differs) takes the parameters from memory as before.
  constexpr bool TAB = MODE == 0 && EPI != 2;
  float* const ptab = smem + NSTAGE * (STAGE_B / 4);
  [[maybe_unused]] int tab_n = -1;
  if constexpr (TAB) {
    tab_n = __builtin_amdgcn_readfirstlane(item0 - (item0 / sc.tiles_n) * sc.tiles_n);
    if (tid < BN) {
      const int k = tab_n * BN + tid;
      float iv = 0.f, bv = 0.f, sv = 1.f;
      if (k < a.K) {
        iv = 1.f / a.wscale[k];                       // power of two: exact
        if (a.bias) bv = a.bias[k];
        if (a.slope) sv = a.slope[k];
      }
      ptab[tid] = iv; ptab[BN + tid] = bv; ptab[2 * BN + tid] = sv;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // landed before this wave reaches the first barrier of the loop
    }
  }
  auto epilogue_p = [&](int item, auto post_c) __attribute__((always_inline)) {
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    constexpr bool post = decltype(post_c)::value;   // (compile-time inside: the runtime flag selected and computed both forms per value)
    __amdgpu_buffer_rsrc_t yr = yr0, rr = yr0, scr = scr0, bir = bir0, slr = slr0, psr = psr0, ptr_ = ptr0;
    bool has_res = false, has_slope = has_slope0;
    int eM = a.M, eK = a.K, eldy = a.ldy, eldr = 0;
    if constexpr (MODE == 1) {
      RowsKArgs* ap = rows_kargs();
      const uint32_t kbytes = (uint32_t)ap->K * 4u;
      yr = dlip_make_rsrc(ap->y, ap->y_bytes);
      has_res = ap->res != nullptr;
      rr = dlip_make_rsrc(ap->res, has_res ? ap->r_bytes : 0u);
      scr = dlip_make_rsrc(ap->wscale, kbytes);
      bir = dlip_make_rsrc(ap->bias, ap->bias ? kbytes : 0u);
      slr = dlip_make_rsrc(ap->slope, ap->slope ? kbytes : 0u);
      psr = dlip_make_rsrc(ap->pscale, ap->pscale ? kbytes : 0u);
      ptr_ = dlip_make_rsrc(ap->pshift, ap->pshift ? kbytes : 0u);
      has_slope = ap->slope != nullptr;
      eM = ap->M; eK = ap->K; eldy = ap->ldy; eldr = ap->ldr;
    }
    const int tile_m = item / sc.tiles_n, tile_n = item - tile_m * sc.tiles_n;
    int row0 = tile_m * BM + wm * WM + lrow;
    [[maybe_unused]] int e_nmi = MI;                  // (TAIL) 16-row blocks of this wave's rows that exist (short last round)
    if constexpr (TAIL) {
      if (item >= ROWS_SC(item_short0)) { e_nmi = ROWS_SC(nmi_short); row0 = ROWS_SC(row_short0) + ((tile_m - ROWS_SC(t_full)) * 2 + wm) * 16 * e_nmi + lrow; }
    }
    const int col0 = tile_n * BN + wn * 64;
    if constexpr (EPI == 1) {
      // after the swap, 16-lane row hq of the pair (2 p, 2 p + 1) holds channels 32 p + {0, 16, 8, 24}[hq] + 0..7 of its pixel
      const int cs = ((hq & 1) << 4) | ((hq & 2) << 2);
      // the residual arrives in the shape the values leave in: per pixel block one 16-B piece of hi halves and one of lo halves,
      // issued first thing for each 32-channel pair -- their latency passes under the parameter loads and the exchange (all
      // NI / 2 pairs at once, as the window kernel does, is 80 registers at MI = 5: spills)
#pragma unroll
      for (int p = 0; p < NI / 2; ++p) {
        const int kb = col0 + 32 * p;                    // first channel of the 32-channel block
        const int k0 = kb + cs;                          // first of this lane's 8 channels
        // (MODE 1, no post-affine -- a launch with both stays on the ring kernel) residual pieces, RD pixel blocks ahead of their use
        constexpr bool RES = MODE == 1 && !post;
        constexpr int RD = 3;
        h8 rh[RES ? RD : 1], rl[RES ? RD : 1];
        auto load_res = [&](int mi) __attribute__((always_inline)) {
          const int m = row0 + mi * 16;
          const uint32_t off = (m < eM && kb < eK) ? (uint32_t)((m * eldr + kb) * 4 + cs * 2) : DLIP_OOB_OFFSET;
          rh[mi % RD] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rr, (int)off, 0, 0));
          rl[mi % RD] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rr, (int)(off == DLIP_OOB_OFFSET ? off : off + 64u), 0, 0));
        };
        if constexpr (RES) {
          if (has_res) {
#pragma unroll
            for (int mi = 0; mi < RD && mi < MI; ++mi) load_res(mi);
          }
        }
        float inv[8], bi[8], sl[8], ps[8], pt[8];
        bool from_tab = false;
        if constexpr (TAB && !post) from_tab = tile_n == tab_n;      // (wave-uniform)
        if (from_tab) {
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const float* tp = ptab + wn * 64 + 32 * p + cs + 4 * q;
            const f32x4 i4 = *reinterpret_cast<const f32x4*>(tp), b4 = *reinterpret_cast<const f32x4*>(tp + BN);
            const f32x4 l4 = *reinterpret_cast<const f32x4*>(tp + 2 * BN);
#pragma unroll
            for (int c = 0; c < 4; ++c) { inv[4 * q + c] = i4[c]; bi[4 * q + c] = b4[c]; sl[4 * q + c] = l4[c]; ps[4 * q + c] = 1.f; pt[4 * q + c] = 0.f; }
          }
        } else {
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const f32x4 s4 = dlip_buffer_load_f4(scr, (uint32_t)(k0 + 4 * q) * 4u);
            const f32x4 b4 = dlip_buffer_load_f4(bir, (uint32_t)(k0 + 4 * q) * 4u);
            f32x4 l4 = {1.f, 1.f, 1.f, 1.f}, p4 = {1.f, 1.f, 1.f, 1.f}, t4 = {0.f, 0.f, 0.f, 0.f};
            if (has_slope) l4 = dlip_buffer_load_f4(slr, (uint32_t)(k0 + 4 * q) * 4u);
            if (post) { p4 = dlip_buffer_load_f4(psr, (uint32_t)(k0 + 4 * q) * 4u); t4 = dlip_buffer_load_f4(ptr_, (uint32_t)(k0 + 4 * q) * 4u); }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              inv[4 * q + c] = k0 + 4 * q + c < eK ? 1.f / s4[c] : 0.f;   // power of two: exact
              bi[4 * q + c] = b4[c]; sl[4 * q + c] = l4[c]; ps[4 * q + c] = p4[c]; pt[4 * q + c] = t4[c];
            }
          }
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          float v[8];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            // (__float_as_uint of a scalar copy: __builtin_bit_cast applied to the vector ELEMENT expression acc[..][c] makes this
            // clang read element 0 for every c -- tools/probes/permlane16_swap.hip)
            const float xc = acc[mi][2 * p][c], yc = acc[mi][2 * p + 1][c];
            const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(xc), __float_as_uint(yc), false, false);
            v[c] = __uint_as_float(r[0]);
            v[4 + c] = __uint_as_float(r[1]);
          }
          h8 hi, lo;
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            float t = v[c] * inv[c] + bi[c];
            if constexpr (RES) { if (has_res) t += (float)rh[mi % RD][c] + (float)rl[mi % RD][c]; }
            t = t >= 0.f ? t : t * sl[c];
            if (post) t = t * ps[c] + pt[c];
            hi[c] = (_Float16)t;
            lo[c] = (_Float16)(t - (float)hi[c]);
            amax = fmaxf(amax, fabsf(t));
          }
          const int m = row0 + mi * 16;
          const bool ok = m < eM && k0 < eK && (!TAIL || mi < e_nmi);   // K % 32 == 0 for a split output: a block is whole or absent
          const uint32_t off = ok ? (uint32_t)((m * eldy + kb) * 4 + cs * 2) : DLIP_OOB_OFFSET;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hi), yr, (int)off, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, lo), yr, (int)(ok ? off + 64u : DLIP_OOB_OFFSET), 0, 0);
          if constexpr (RES) {
            DLIP_FENCE();
            if (has_res && mi + RD < MI) load_res(mi + RD);
          }
        }
        DLIP_FENCE();
      }
    } else if constexpr (EPI == 2) {
      // Pooled: nothing is written but, per wave ROW (its 16 MI rows are one "tile" of the partials: tile index 2 tile_m + wm) and
      // row-group segment, the column sums of v and v^2 -- fp64, fixed order: a lane adds its MI pixels, the 16 lanes of a
      // channel quad meet by butterfly, lane 0 of each quad stores.  Rows before the half tile's one possible group boundary
      // are segment 0, the rest segment 1 (pool_group >= 16 MI); dlip_pool_finish_f32 / znorm_cat_pooled add the tiles in row order.
      const int m0 = tile_m * BM + wm * WM;
      const int g0 = m0 / a.pool_group;
      const int rb = (g0 + 1) * a.pool_group - m0;
      int e0 = rb, e1 = WM;                           // ragged batches: tile-relative ends of the valid rows of the two segments
      if (a.pool_len.len != nullptr) {
        const int G = (eM + a.pool_group - 1) / a.pool_group;
        e0 = min(rb, g0 * a.pool_group + dlip_valid_rows(a.pool_len, g0, a.pool_group) - m0);
        e1 = g0 + 1 < G ? rb + dlip_valid_rows(a.pool_len, g0 + 1, a.pool_group) : rb;
      }
      const int Kp = (eK + 127) / 128 * 128;
      double* prow = a.pool + (size_t)(2 * tile_m + wm) * 4 * Kp;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int k0 = col0 + 16 * ni + 4 * hq;
        const f32x4 s4 = dlip_buffer_load_f4(scr, (uint32_t)k0 * 4u);
        const f32x4 b4 = dlip_buffer_load_f4(bir, (uint32_t)k0 * 4u);
        f32x4 l4 = {1.f, 1.f, 1.f, 1.f}, p4 = {1.f, 1.f, 1.f, 1.f}, t4 = {0.f, 0.f, 0.f, 0.f};
        if (has_slope) l4 = dlip_buffer_load_f4(slr, (uint32_t)k0 * 4u);
        if (post) { p4 = dlip_buffer_load_f4(psr, (uint32_t)k0 * 4u); t4 = dlip_buffer_load_f4(ptr_, (uint32_t)k0 * 4u); }
        f32x4 inv4;
#pragma unroll
        for (int c = 0; c < 4; ++c) inv4[c] = k0 + c < eK ? 1.f / s4[c] : 0.f;
        double st[4][4];   // [sum0, sumsq0, sum1, sumsq1][channel]
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int c = 0; c < 4; ++c) st[q][c] = 0.0;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const int r = mi * 16 + lrow;
          const bool seg1 = r >= rb, in = m0 + r < eM && r < (seg1 ? e1 : e0);
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            float t = acc[mi][ni][c] * inv4[c] + b4[c];
            t = t >= 0.f ? t : t * l4[c];
            if (post) t = t * p4[c] + t4[c];
            const double d = in ? (double)t : 0.0;
            st[0][c] += seg1 ? 0.0 : d; st[1][c] += seg1 ? 0.0 : d * d;
            st[2][c] += seg1 ? d : 0.0; st[3][c] += seg1 ? d * d : 0.0;
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) st[q][c] += __shfl_xor(st[q][c], o, 64);
        if (lrow == 0 && m0 < eM) {   // (a last tile's second half may lie wholly beyond M: it has no partial row set)
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < 4; ++c)
              if (k0 + c < Kp) prow[(size_t)q * Kp + k0 + c] = st[q][c];
        }
        DLIP_FENCE();
      }
    } else {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int k0 = col0 + 16 * ni + 4 * hq;
        f32x4 b4, inv4, l4 = {1.f, 1.f, 1.f, 1.f}, p4 = {1.f, 1.f, 1.f, 1.f}, t4 = {0.f, 0.f, 0.f, 0.f};
        bool from_tab = false;
        if constexpr (TAB && !post) from_tab = tile_n == tab_n;      // (wave-uniform: the parameter table in LDS, see ptab)
        if (from_tab) {
          const float* tp = ptab + wn * 64 + 16 * ni + 4 * hq;
          inv4 = *reinterpret_cast<const f32x4*>(tp); b4 = *reinterpret_cast<const f32x4*>(tp + BN); l4 = *reinterpret_cast<const f32x4*>(tp + 2 * BN);
        } else {
          const f32x4 s4 = dlip_buffer_load_f4(scr, (uint32_t)k0 * 4u);
          b4 = dlip_buffer_load_f4(bir, (uint32_t)k0 * 4u);
          if (has_slope) l4 = dlip_buffer_load_f4(slr, (uint32_t)k0 * 4u);
          if (post) { p4 = dlip_buffer_load_f4(psr, (uint32_t)k0 * 4u); t4 = dlip_buffer_load_f4(ptr_, (uint32_t)k0 * 4u); }
#pragma unroll
          for (int c = 0; c < 4; ++c) inv4[c] = k0 + c < eK ? 1.f / s4[c] : 0.f;
        }
        // (MODE 0, a.stats: launch-uniform) column sums of what this wave writes -- the batch statistics of the BatchNorm behind a
        // TDNN convolution under model.train() (tdnn.py:35-43) -- taken here instead of by a pass over y: a lane adds its MI pixels
        // (fp32: 5 values), the 16 lanes of a channel quad meet by four DPP row steps (fp32, a fixed tree: deterministic), lane 0 of
        // the quad converts to fp64 and stores {sum, sum of squares} of its 4 channels for chunk 2 tile_m + wm; the finalize kernel
        // adds the chunks in fp64 as it adds col_partial_kernel's.  80 values per fp32 sum: 5e-7 relative per chunk, random over
        // ~10^3 chunks.
        float st_s[4] = {0.f, 0.f, 0.f, 0.f}, st_q[4] = {0.f, 0.f, 0.f, 0.f};
        // (the pointer is re-read from the kernel arguments HERE: held in scalar registers across the main loop it cost the fp32
        // instances 30 extra SGPR spills -- the paired-lanes lesson of conv_igemm_f16x3_dma.hip)
        double* const stats_p = MODE == 0 ? *reinterpret_cast<double* const volatile*>(&a.stats) : nullptr;
        const bool want_stats = stats_p != nullptr;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          f32x4 v;
          const int m = row0 + mi * 16;
#pragma unroll
          for (int c = 0; c < 4; ++c) v[c] = acc[mi][ni][c] * inv4[c] + b4[c];
          if constexpr (MODE == 1) {
            if (has_res) {   // the residual is in the split format: 4 hi halves, 4 lo halves of this lane's channels
              typedef _Float16 h4 __attribute__((ext_vector_type(4)));
              const bool rok = m < eM && k0 < eK;
              const uint32_t ro = rok ? (uint32_t)((m * eldr + (k0 & ~31)) * 4 + (k0 & 31) * 2) : DLIP_OOB_OFFSET;
              const h4 r_hi = __builtin_bit_cast(h4, __builtin_amdgcn_raw_buffer_load_b64(rr, (int)ro, 0, 0));
              const h4 r_lo = __builtin_bit_cast(h4, __builtin_amdgcn_raw_buffer_load_b64(rr, (int)(rok ? ro + 64u : DLIP_OOB_OFFSET), 0, 0));
#pragma unroll
              for (int c = 0; c < 4; ++c) v[c] += (float)r_hi[c] + (float)r_lo[c];
            }
          }
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            float t = v[c];
            t = t >= 0.f ? t : t * l4[c];
            if (post) t = t * p4[c] + t4[c];
            v[c] = t;
          }
          if (want_stats && m < eM && (!TAIL || mi < e_nmi)) {
#pragma unroll
            for (int c = 0; c < 4; ++c) { st_s[c] += v[c]; st_q[c] += v[c] * v[c]; }
          }
          const uint32_t off = (m < eM && k0 < eK && (!TAIL || mi < e_nmi)) ? (uint32_t)((m * eldy + k0) * 4) : DLIP_OOB_OFFSET;   // K % 4 == 0
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yr, (int)off, 0, 0);
        }
        if (want_stats) {
          auto row_sum = [](float x) {   // the sum over the 16 lanes of a DPP row, in every lane: xor 1, xor 2, half mirror, mirror
            x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, true));
            x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, true));
            x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xf, 0xf, true));
            x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xf, 0xf, true));
            return x;
          };
#pragma unroll
          for (int c = 0; c < 4; ++c) { st_s[c] = row_sum(st_s[c]); st_q[c] = row_sum(st_q[c]); }
          if (lrow == 0 && k0 < eK) {
            double* dst = stats_p + ((size_t)(2 * tile_m + wm) * eK + k0) * 2;
#pragma unroll
            for (int c = 0; c < 4; ++c) { dst[2 * c] = (double)st_s[c]; dst[2 * c + 1] = (double)st_q[c]; }
          }
        }
        DLIP_FENCE();
      }
    }
  };
  auto epilogue = [&](int item) __attribute__((always_inline)) {
    bool has_post = has_post0;
    if constexpr (MODE == 1) has_post = rows_kargs()->pscale != nullptr;
    if (has_post) epilogue_p(item, std::true_type{}); else epilogue_p(item, std::false_type{});
  };
  // End of a segment (this wave's view).  A whole tile: its epilogue.  MODE 1, a tile this range shares with others (`part`): the
  // hand-off in the kernel comment -- peek at the tile's ticket word for this wave position; everybody else has published: this
  // wave is the finisher and keeps its part in registers; otherwise publish (write-through stores, drained, then ONE relaxed
  // ticket) and finish only if the ticket says the others arrived meanwhile.  The finisher acquires (agent scope: this CU's L1
  // drops stale slab lines), adds the parts IN PART ORDER (its own from registers) one row of NI quads at a time, and runs the
  // epilogue.  Returns whether the epilogue's stores were issued (the counted waits behind it leave NST of them in flight); on the
  // publishing path everything this wave had in flight has been waited for.
  auto finish_tile = [&](int item, bool part, bool first) __attribute__((always_inline)) -> bool {
    if constexpr (MODE == 1) {
      if (part) {                                          // (wave-uniform)
        constexpr int WSLAB = BM * BN / NW;                // floats of one wave's share of a slab = MI NI 64 quads
        const int t0 = item * nk;
        const int gf = __builtin_amdgcn_readfirstlane(dlip_div((t0 + 1) * sc.G - 1, sc.div_iters));    // owner of the tile's first slice
        const int gl = __builtin_amdgcn_readfirstlane(dlip_div((t0 + nk) * sc.G - 1, sc.div_iters));   // owner of its last slice
        const int others = gl - gf;
        // which of part p's two slabs holds this tile: only the first part's range can have begun before the tile
        const int gf_slot = dlip_div(gf * sc.iters, sc.div_G) < t0 ? 1 : 0;
        int* ctr = sc.counters + (size_t)item * NW + wave;
        int seen = 0;
        if (lane == 0) seen = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool fin = __builtin_amdgcn_readfirstlane(seen) == others;
        if (!fin) {
          const __amdgpu_buffer_rsrc_t sr = dlip_make_rsrc(sc.slabs + ((size_t)(2 * g + (first ? 0 : 1)) * NW + wave) * WSLAB, WSLAB * 4);
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[mi][ni]), sr, ((mi * NI + ni) * 64 + lane) * 16, 0, 16);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          int tk = 0;
          if (lane == 0) tk = __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          fin = __builtin_amdgcn_readfirstlane(tk) == others;   // the other parts arrived between the peek and the ticket
          if (!fin) return false;
        }
        if (lane == 0) __hip_atomic_store(ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          f32x4 t[NI];
          for (int p = gf; p <= gl; ++p) {
            const __amdgpu_buffer_rsrc_t pr =
                dlip_make_rsrc(sc.slabs + ((size_t)(2 * p + (p == gf ? gf_slot : 0)) * NW + wave) * WSLAB, WSLAB * 4);
            f32x4 v[NI];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
              v[ni] = acc[mi][ni];
              if (p != g) v[ni] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(pr, ((mi * NI + ni) * 64 + lane) * 16, 0, 16));
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
              for (int c = 0; c < 4; ++c) t[ni][c] = p == gf ? v[ni][c] : t[ni][c] + v[ni][c];
            DLIP_FENCE();
          }
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = t[ni];
        }
      }
    }
    epilogue(item);
    return true;
  };

  // ---- prologue: the first two slices of the stream in flight ----
  constexpr std::integral_constant<int, NA_A> na_a{};
  constexpr std::integral_constant<int, NA_B> na_b{};
  set_item(c_item);
  if constexpr (MODE == 1) {                               // the range may begin inside a tile: stand the cursor on slice k0
    if (DUAL && k0 >= nk1) { c0 = a.Cw + (k0 - nk1) * BK; tap = 0; } else { c0 = (k0 / ntaps) * BK; tap = k0 % ntaps; }
    s_pos = tap % a.S;
    x_row = (tap / a.S) * x_dr;
    place();
  }
  ROWS_STAMP(0);
  if (!half_b) issue(0, na_a); else issue(0, na_b);
  if (total > 1) {
    enter_next();
    advance();
    if (!half_b) issue(1, na_a); else issue(1, na_b);
  }
  zero_acc();

  // One half of the ping-pong loop; NL = this half's pieces per slice, FIRST = half A (reads a slice first).
  //   interval between barriers   b(2s-1) .. b(2s)      |  b(2s) .. b(2s+1)
  //   half A                      [EPI] LOAD(s)         |  MFMA(s), wait slice s+1
  //   half B                      MFMA(s-1)             |  [EPI] LOAD(s), wait slice s+1
  // LOAD(s): read every fragment of slice s (stage s % 3), issue this wave's pieces of slice s+2 into the stage of slice s-1
  // (whose last reader finished before b(2s-1)), lgkmcnt(0).  [EPI]: when the previous matrix phase finished a tile, its
  // epilogue runs first -- beside the partner's matrix phase -- and its NST stores are then the only vector-memory
  // operations younger than slice s+1's pieces besides slice s+2's: the counted wait leaves them in flight too.
  auto run_half = [&](auto na, auto first_c) __attribute__((always_inline)) {
    constexpr int NL = decltype(na)::value + B_PER;
    constexpr bool FIRST = decltype(first_c)::value;
    static_assert(NL + NST < 64, "vmcnt is a 6-bit counter");
    int st_iss = total > 1 ? 2 : 1;                        // stage the next issue goes to (the prologue issued slices 0, 1)
    // consumer side: the segment being multiplied (MODE 0: always a whole tile)
    int kleft = nk, e_item = item0, done_item = -1;
    [[maybe_unused]] int cur_nmi = MI;                     // (TAIL) the height of the tile being multiplied, in 16-row blocks per wave
    if constexpr (TAIL) { if (e_item >= ROWS_SC(item_short0)) cur_nmi = ROWS_SC(nmi_short); }
    [[maybe_unused]] int rem = 0;
    [[maybe_unused]] bool e_part = false, e_first = true, done_part = false, done_first = false;   // (wave-uniform; MODE 1)
    if constexpr (MODE == 1) {
      kleft = nk - k0 < total ? nk - k0 : total;
      rem = total - kleft;
      e_part = kleft != nk;
    }
    if (total > 1) wait_vmcnt<NL>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();                          // slice 0 is complete
    ROWS_STAMP(1);
    if (!FIRST) __builtin_amdgcn_s_barrier();              // b(0): half A reads slice 0 first
    int st_cur = 0;
    for (int s = 0;; ++s) {
      bool did_epi = false;                                // (wave-uniform)
      ROWS_SSTAMP(0);
      if (done_item >= 0) {
        if constexpr (MODE == 1) {
          did_epi = finish_tile(done_item, done_part, done_first);
        } else {
          epilogue(done_item);
          did_epi = true;
        }
        if (s == total) break;                             // the stream ends on a segment's last slice
        zero_acc(); done_item = -1;
      }
      const bool more2 = s + 2 < total;
      if (more2) enter_next();
      DLIP_FENCE();
      read_all(st_cur);                                    // the reads first: their latency passes under the piece issue below
      st_cur = st_cur + 1 == NSTAGE ? 0 : st_cur + 1;
      DLIP_FENCE();
      if (more2) {
        advance();
        issue(st_iss, na);
        st_iss = st_iss + 1 == NSTAGE ? 0 : st_iss + 1;
      }
      DLIP_FENCE();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the stage may be refilled behind the next barrier: the reads are done
      DLIP_FENCE();
      ROWS_SSTAMP(1);
      auto wait_next = [&]() __attribute__((always_inline)) {   // this wave's pieces of slice s + 1 have landed
        if (s + 1 >= total) return;
        if (more2) { if (did_epi) wait_vmcnt<NL + NST>(); else wait_vmcnt<NL>(); }
        else       { if (did_epi) wait_vmcnt<NST>(); else wait_vmcnt<0>(); }
      };
      if (!FIRST) wait_next();                             // half B waits in front of the barrier that opens its matrix phase
      ROWS_SSTAMP(2);
      __builtin_amdgcn_s_barrier();                        // A: b(2s)      B: b(2s+1)
      ROWS_SSTAMP(3);
      mfma_all(cur_nmi);
      ROWS_SSTAMP(4);
      if constexpr (MODE == 1) {
        if (--kleft == 0) {
          done_item = e_item; done_part = e_part; done_first = e_first;
          ++e_item;
          kleft = nk < rem ? nk : rem;
          rem -= kleft;
          e_part = kleft != nk;
          e_first = false;
        }
      } else {
        if (--kleft == 0) {
          kleft = nk; done_item = e_item; e_item += nwg;
          if constexpr (TAIL) { if (e_item >= ROWS_SC(item_short0)) cur_nmi = ROWS_SC(nmi_short); }
        }
      }
      if (FIRST) {
        __builtin_amdgcn_s_barrier();                      // b(2s+1)
        ROWS_SSTAMP(5);
        // half A is the FIRST reader of slice s + 1 (its next load phase): its own pieces must have landed before ITS reads
        wait_next();
        ROWS_SSTAMP(6);
      } else if (s + 1 < total) {
        __builtin_amdgcn_s_barrier();                      // b(2s+2)
        ROWS_SSTAMP(5);
      }
    }
    ROWS_STAMP(2);
  };
  if (!half_b) run_half(na_a, std::true_type{}); else run_half(na_b, std::false_type{});
#undef DLIP_FENCE
  if constexpr (EPI == 1) dlip_report_range(amax, a.status);
  ROWS_STAMP(3);
  dlip_span_exit(sc.span);
#ifdef DLIP_LAB
  if (sc.stamps && tid == 0) sc.stamps[(size_t)g * 16 + 6] = __builtin_amdgcn_s_memrealtime() - sc.stamps[(size_t)g * 16 + 7];
#endif
}

constexpr int kRowsDeclined = -1000;   // (internal) a MODE 1 launch that found no split workspace: the caller takes the ring kernel

int rows_cus() {
  static int cus = 0;                                  // (one device kind per process: every MI355X has the same count)
  if (cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
  }
  return cus > 0 ? cus : 256;
}

// The short last round (RowsSched): M rows on tiles of 32 mi x 256, `slots` resident workgroups (one per CU).  full = row tiles of the
// full height (whole rounds of the chip), nmi = height of the rest in units of 32 rows (== mi: no short tiles), tiles_m = all row tiles.
struct RowsTail { long long tiles_m, full; int nmi; };
RowsTail rows_tail(long long M, int tiles_n, int mi, int slots) {
  const int bm = 32 * mi;
  RowsTail t{(M + bm - 1) / bm, 0, mi};
  if (dlip_dbg_value[DLIP_DBG_ROWS_TAIL] == 0 || slots <= 0 || tiles_n <= 0 || slots % tiles_n != 0) return t;
  const long long slots_m = slots / tiles_n, per_round = (long long)bm * slots_m;
  const long long rounds = M / per_round, rem = M - rounds * per_round;
  if (rounds == 0 || rem == 0) return t;              // (one round: rows_pick_mi has picked the height; whole rounds: nothing left over)
  const int nmi = (int)((rem + 32 * slots_m - 1) / (32 * slots_m));
  if (nmi >= mi) return t;
  t.full = rounds * slots_m;
  t.nmi = nmi;
  t.tiles_m = t.full + (rem + 32 * nmi - 1) / (32 * nmi);
  return t;
}

template <int MI, int EPI, int MODE = 0, bool DUAL = false, bool TAIL = false>
int launch_rows(const ConvArgs& a, hipStream_t st) {
  constexpr int BM = 32 * MI;
  constexpr size_t lds = (size_t)3 * (BM + ROWS_BN) * ROWB + (MODE == 0 && EPI != 2 ? 3 * ROWS_BN * sizeof(float) : 0);   // ring + the epilogue's parameter table
  static_assert(lds <= 160 * 1024, "LDS ring exceeds a CU");
  ConvArgs b = a;
  b.tiles_n = (a.K + ROWS_BN - 1) / ROWS_BN;
  auto kern = conv_rows_f16x3_kernel<MI, EPI, MODE, DUAL, TAIL>;
  static DlipKernelState ks;
  int e = ks.ensure_lds(reinterpret_cast<const void*>(kern), lds);
  if (e != DLIP_OK) return e;
  int slots = 0;
  e = ks.resident(reinterpret_cast<const void*>(kern), 512, lds, &slots);   // one workgroup per CU
  if (e != DLIP_OK) return e;
  RowsSched sc;
  long long tiles_m = ((long long)a.M + BM - 1) / BM;
  sc.item_short0 = sc.t_full = 0x7FFFFFFF; sc.nmi_short = MI; sc.row_short0 = 0;
  if constexpr (TAIL) {
    // the short last round is laid out for the CUs of the chip -- what dlip_conv_stats_chunks can know without a launch; `slots` is that
    // number whenever the kernel is resident once per CU (if it ever were not, the layout would still be correct, only less even)
    const RowsTail t = rows_tail(a.M, b.tiles_n, MI, rows_cus());
    if (t.nmi < MI) {
      tiles_m = t.tiles_m;
      sc.t_full = (int)t.full; sc.item_short0 = (int)(t.full * b.tiles_n); sc.nmi_short = t.nmi; sc.row_short0 = (int)(t.full * BM);
    }
  }
  const long long items = tiles_m * b.tiles_n;
  if (items <= 0 || items > 0x3FFFFFFFll) return DLIP_EINVAL;
  sc.items = (int)items;
  sc.tiles_n = b.tiles_n;
  sc.iters = 0; sc.G = 0; sc.slabs = nullptr; sc.counters = nullptr;
  sc.div_G = sc.div_iters = sc.div_nk = FastDiv{0u, 0u};
  long long grid = items < slots ? items : slots;
  if constexpr (MODE == 1) {
    // The ring kernel's balanced split: items x nk slices in G equal ranges, G = the resident workgroups (at least 16 slices
    // each).  Ranges that end inside a tile need the split workspace (two slabs per workgroup, 8 ticket words per tile).
    const long long iters = items * (long long)a.nk;
    long long G = slots;
    if (iters < 16 * G) G = iters / 16 > 0 ? iters / 16 : 1;
    if ((iters + a.nk) * (G + 1) >= 0x7FFFFFFFll) return kRowsDeclined;   // the kernel's 32-bit range arithmetic
    sc.iters = (int)iters;
    sc.div_G = dlip_fastdiv((uint32_t)G);
    sc.div_iters = dlip_fastdiv((uint32_t)iters);
    sc.div_nk = dlip_fastdiv((uint32_t)a.nk);
    if (items % G != 0) {
      int max_words = 0;
      if (!dlip_conv_split_workspace(st, (size_t)2 * G * BM * ROWS_BN, &sc.slabs, &sc.counters, &max_words) || items * 8 > max_words)
        return kRowsDeclined;
    }
    sc.G = (int)G;
    grid = G;
  }
  sc.span = dlip_span_next();
#ifdef DLIP_LAB
  sc.stamps = nullptr;
  if (getenv("DLIP_STAMP_PRINT")) {
    static unsigned long long* dbuf = nullptr;
    if (!dbuf) (void)hipMalloc(reinterpret_cast<void**>(&dbuf), 4096 * 16 * 8);
    (void)hipMemsetAsync(dbuf, 0, 4096 * 16 * 8, st);
    sc.stamps = dbuf;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds, st, b, sc);
    (void)hipStreamSynchronize(st);
    std::vector<unsigned long long> h((size_t)grid * 16);
    (void)hipMemcpy(h.data(), dbuf, h.size() * 8, hipMemcpyDeviceToHost);
    auto med = [](std::vector<double>& v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    std::vector<double> d[3], sl[6], clk;
    for (long long i = 0; i < grid; ++i) {
      const unsigned long long* r = &h[(size_t)i * 16];
      if (!r[3]) continue;
      for (int j = 0; j < 3; ++j) d[j].push_back((double)(r[j + 1] - r[j]));
      if (r[6]) clk.push_back((double)(r[3] - r[0]) / (double)r[6] * 100.0);
      if (r[14]) for (int j = 0; j < 6; ++j) sl[j].push_back((double)(r[9 + j] - r[8 + j]));
    }
    fprintf(stderr, "[rows stamps %dx%d M=%d K=%d nk=%d grid=%lld] first fill %.0f  stream %.0f (%.0f / slice)  last epilogue %.0f  clock %.0f MHz\n",
            BM, ROWS_BN, b.M, b.K, b.nk, grid, med(d[0]), med(d[1]), med(d[1]) / (double)(((items + grid - 1) / grid) * b.nk), med(d[2]), med(clk));
    fprintf(stderr, "[rows slice 8, wave 0 (half A)] epi+reads+issue %.0f  (wait) %.0f  barrier %.0f  matrix phase %.0f  barrier %.0f  wait next %.0f\n",
            med(sl[0]), med(sl[1]), med(sl[2]), med(sl[3]), med(sl[4]), med(sl[5]));
    return dlip_launch_status();
  }
#endif
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds, st, b, sc);
  return dlip_launch_status();
}

// Tile height.  A launch takes ceil(items / CUs) rounds of one tile-time ~ MI (+ a fixed part per round: first fill and the
// exposed half of the last epilogue, about a slice and a half of a 16-slice tile); the fewest "row units" wins, the taller
// tile on a tie (fewer bytes per MFMA).
// (round 5) ... and a launch whose last round is SHORT (rows_tail) pays that round at its own height.
int rows_pick_mi(long long M, int K, int nk, int cus, bool tail = true) {   // tail = false: a pooled launch (tiles of one height)
  const int tiles_n = (K + ROWS_BN - 1) / ROWS_BN;
  int best = 5;
  double best_cost = 1e300;
  for (int mi = 5; mi >= 3; --mi) {
    const RowsTail t = tail ? rows_tail(M, tiles_n, mi, cus) : RowsTail{(M + 32 * mi - 1) / (32 * mi), 0, mi};
    const double eff = mi == 5 ? 1.0 : mi == 4 ? 1.04 : 1.10;
    double cost;
    if (t.nmi < mi) {
      const long long rounds = t.full * tiles_n / cus;
      cost = ((double)rounds * (mi * (double)nk + 1.5 * 5.0) + (t.nmi * (double)nk + 1.5 * 5.0)) * eff;
    } else {
      const long long items = t.tiles_m * tiles_n;
      const long long rounds = (items + cus - 1) / cus;
      cost = (double)rounds * (mi * (double)nk + 1.5 * 5.0) * eff;
    }
    if (cost < best_cost * 0.999) { best_cost = cost; best = mi; }
  }
  return best;
}

}  // namespace

extern "C" int dlip_conv_dma_enabled(void);   // conv_igemm_f16x3.hip

// Which launches the rows kernel serves: H = 1, one filter row, stride 1, no padding (every tap of every output row exists: a
// plain row offset), whole 32-channel slices, no residual / second source / pooled epilogue, and enough rows and columns for its
// 256-column tiles to be the right shape (the fully connected layers on a batch of utterances stay on the ring kernel's split).
static bool rows_shape_ok(int H, int R, int S, int sh, int sw, int ph, int pw, int C, int K, long long M) {
  const int v = dlip_dbg_value[DLIP_DBG_ROWS];
  if (v == 0 || !dlip_conv_dma_enabled()) return false;
  if (!(H == 1 && R == 1 && sw == 1 && sh == 1 && ph == 0 && pw == 0 && (C & 31) == 0 && S <= 32)) return false;
  if (v > 0) return true;                                  // forced (tests)
  return K >= 192 && M >= 4096;
}

extern "C" __attribute__((visibility("hidden"))) int dlip_conv_rows_ok(const void* args) {
  const ConvArgs& a = *static_cast<const ConvArgs*>(args);
  return a.Cw == a.C && a.res == nullptr && a.x2 == nullptr && a.wscale != nullptr &&
         rows_shape_ok(a.H, a.R, a.S, a.sh, a.sw, a.ph, a.pw, a.C, a.K, a.M);
}

// The same decision from a descriptor (dlip_conv_kernel_kind / dlip_conv_plan: which kernel and tile a profiler will show for a
// split-format launch of `d` without a residual); *bm = the tile height the launch will use.
static int rows_plan(const dlip_conv_desc* d, int* bm, bool tail) {
  const long long M = (long long)d->N * d->Ho * d->Wo;
  if (d->ldr != 0 || !rows_shape_ok(d->H, d->R, d->S, d->stride_h, d->stride_w, d->pad_h, d->pad_w, d->C, d->K, M)) return 0;
  int mi = rows_pick_mi(M, d->K, d->S * (d->C / 32), rows_cus(), tail);
  if (const int v = dlip_dbg_value[DLIP_DBG_ROWS]; v >= 3 && v <= 5) mi = v;
  if (bm) *bm = 32 * mi;
  return 1;
}

extern "C" __attribute__((visibility("hidden"))) int dlip_conv_rows_plan(const dlip_conv_desc* d, int* bm) { return rows_plan(d, bm, true); }

// Row tiles of the fp32 / split-output launch dlip_conv_rows_plan describes (with its short last round, if any): the statistics
// epilogue writes two partial rows per row tile (dlip_conv_stats_chunks).
extern "C" __attribute__((visibility("hidden"))) long long dlip_conv_rows_tiles(const dlip_conv_desc* d) {
  int bm = 0;
  if (!dlip_conv_rows_plan(d, &bm) || bm <= 0) return 0;
  const long long M = (long long)d->N * d->Ho * d->Wo;
  return rows_tail(M, (d->K + ROWS_BN - 1) / ROWS_BN, bm / 32, rows_cus()).tiles_m;
}

// MODE 1: which launches of the LDS-DMA path (split input; dlip_conv_f16x3_dma_launch asks first) CAN take the rows kernel's general
// mode: pixel-major operands, whole 32-channel slices, at most 32 taps, fp32 or split output (no pooled epilogue).
// NOT CHOSEN BY THE LIBRARY: it runs only when forced (dlip_debug_set(7, 1): tests, A/B runs).  Built in round 4 to carry the
// trunk's layers 3 and 4 (the ring kernel's 256 x 128 launches) on the 160 x 256 tile / continuous stream / register epilogue that
// made the speech encoder's layers 25 % faster; measured on one box at B = 64 (tools/probes/rows2d_layers.py,
// profiles/r4/rows2d_layers.txt): layer 3's convolutions 2 - 4 % SLOWER than the ring kernel (200 vs 196 us, with residual 214
// vs 207), its stride-2 and two-source launches 4 - 11 % slower, layer 4's 6 - 18 % slower.  Why (tools/probes/rows_trunk_proxy.py):
// on rows that fill whole rounds of tiles -- no split -- this mode does 436 - 450 TFLOP/s against the ring kernel's 410 - 423,
// but the trunk's row counts need the balanced split (418 tiles on 256 CUs), and here a shared tile's hand-off costs 25 - 32 us per
// launch against the ring kernel's ~10: each ping-pong half publishes / finishes in its own interval (store acknowledgements, a
// ticket round trip, MI x parts serial slab reads), the other half waiting at the next barrier; and every workgroup enters the
// reduction at its own slice, so layer 4's 4.7 MB of weights per 256-column block no longer stay in an XCD's 4 MiB L2 (the ring
// kernel keeps an XCD on one 128-column block: 2.35 MB).  What would change it is in DESIGN.md section 9.
// ROUND 5: a measured negative result belongs in the lab, not in the product library: the general mode's four instances (220-256
// VGPRs, 65-83 SGPR spills) are compiled ONLY under -DDLIP_LAB (python -m deeplip_amd.build --lab; tests/test_kernels_gpu.py's
// general-mode tests and tools/probes/rows2d_layers.py load that library through DLIP_LIB_PATH).  In libdeeplip_hip.so the three entry
// points below decline every launch and dlip_debug_set refuses key 7.
#ifdef DLIP_LAB
static bool rows2d_ok(const ConvArgs& a, int epi) {
  const int v = dlip_dbg_value[DLIP_DBG_ROWS2D];
  if (v <= 0 || !dlip_conv_dma_enabled() || epi < 0 || epi > 1) return false;
  if (a.Cw != a.C || (a.C & 31) != 0 || a.R * a.S > 32 || a.wscale == nullptr || a.pool != nullptr) return false;
  if (a.cs_x != 128 || a.cs_w != 128 || a.wt != a.Cw * 4 || a.Hs != a.H) return false;       // slice-major images: ring kernel only
  if (a.x2 != nullptr && (a.nk2 <= 0)) return false;
  if (a.res != nullptr && a.pscale != nullptr) return false;                                    // (not compiled: registers)
  const long long items = (((long long)a.M + 159) / 160) * ((a.K + ROWS_BN - 1) / ROWS_BN);
  if (items * 8 > 65536) return false;
  return true;
}

extern "C" __attribute__((visibility("hidden"))) int dlip_conv_rows2d_ok(const void* args, int epi) {
  return rows2d_ok(*static_cast<const ConvArgs*>(args), epi) ? 1 : 0;
}

// Descriptor form (dlip_conv_plan / dlip_conv_kernel_kind, `dual` = a second source of c2 channels): 1 if a split-format launch of `d`
// (with or without a residual) runs the general mode; *bm = its tile height.
extern "C" __attribute__((visibility("hidden"))) int dlip_conv_rows2d_plan(const dlip_conv_desc* d, int c2, int* bm) {
  const int v = dlip_dbg_value[DLIP_DBG_ROWS2D];
  if (v <= 0 || !dlip_conv_dma_enabled() || (d->C & 31) != 0 || d->R * d->S > 32 || (c2 & 31) != 0) return 0;
  const long long M = (long long)d->N * d->Ho * d->Wo;
  const long long items = ((M + 159) / 160) * ((d->K + ROWS_BN - 1) / ROWS_BN);
  if (items * 8 > 65536) return 0;
  if (bm) *bm = 160;
  return 1;
}

// returns kRowsDeclined when the split workspace is missing (stream capture before any eager launch, or an external block that
// is too small): the caller then runs the ring kernel
extern "C" __attribute__((visibility("hidden"))) int dlip_conv_f16x3_rows2d_launch(const void* args, void* stream, int epi) {
  const ConvArgs& a = *static_cast<const ConvArgs*>(args);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (a.x2 != nullptr) return epi ? launch_rows<5, 1, 1, true>(a, st) : launch_rows<5, 0, 1, true>(a, st);
  return epi ? launch_rows<5, 1, 1, false>(a, st) : launch_rows<5, 0, 1, false>(a, st);
}
#else
extern "C" __attribute__((visibility("hidden"))) int dlip_conv_rows2d_ok(const void*, int) { return 0; }
extern "C" __attribute__((visibility("hidden"))) int dlip_conv_rows2d_plan(const dlip_conv_desc*, int, int*) { return 0; }
extern "C" __attribute__((visibility("hidden"))) int dlip_conv_f16x3_rows2d_launch(const void*, void*, int) { return DLIP_EINVAL; }
#endif
extern "C" __attribute__((visibility("hidden"))) int dlip_conv_rows_declined(void) { return kRowsDeclined; }

// The pooled epilogue on this kernel is correct and tested, and slower than the ring kernel's LDS-staged one: the fp64 column
// sums taken straight from the accumulators cost ~13 us per tile in cross-lane work (64 doubles x 4 butterfly steps per channel
// quad), tdnn.9 at B = 64: 130 us against 96 us on the ring kernel's <128,128,pool> instance (tools/bench_rows.py).  So a
// pooled launch comes here only when the rows kernel is FORCED (dlip_debug_set(6, 1 | 3 | 4 | 5): tests, A/B runs).
extern "C" __attribute__((visibility("hidden"))) int dlip_conv_rows_pool_plan(const dlip_conv_desc* d, int* bm) {
  if (dlip_dbg_value[DLIP_DBG_ROWS] <= 0) return 0;
  return rows_plan(d, bm, false);
}

extern "C" __attribute__((visibility("hidden"))) int dlip_conv_f16x3_rows_launch(const void* args, void* stream, int epi) {
  const ConvArgs& a = *static_cast<const ConvArgs*>(args);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int cus = rows_cus();
  int mi = rows_pick_mi(a.M, a.K, a.nk, cus, epi != 2);
  if (const int v = dlip_dbg_value[DLIP_DBG_ROWS]; v >= 3 && v <= 5) mi = v;   // dlip_debug_set: a forced tile height (tests, A/B)
  if (epi == 2) {   // pooled: a half tile (16 mi rows) may contain at most one row-group boundary
    if (a.pool == nullptr || a.pool_group < 16 * mi) return DLIP_EINVAL;
    switch (mi) {
      case 3: return launch_rows<3, 2>(a, st);
      case 4: return launch_rows<4, 2>(a, st);
      default: return launch_rows<5, 2>(a, st);
    }
  }
  // a launch with a short last round (rows_tail) runs the TAIL instance; everything else the instance without that code
  if (rows_tail(a.M, (a.K + ROWS_BN - 1) / ROWS_BN, mi, cus).nmi < mi) {
    switch (mi) {
      case 3: return epi ? launch_rows<3, 1, 0, false, true>(a, st) : launch_rows<3, 0, 0, false, true>(a, st);
      case 4: return epi ? launch_rows<4, 1, 0, false, true>(a, st) : launch_rows<4, 0, 0, false, true>(a, st);
      default: return epi ? launch_rows<5, 1, 0, false, true>(a, st) : launch_rows<5, 0, 0, false, true>(a, st);
    }
  }
  switch (mi) {
    case 3: return epi ? launch_rows<3, 1>(a, st) : launch_rows<3, 0>(a, st);
    case 4: return epi ? launch_rows<4, 1>(a, st) : launch_rows<4, 0>(a, st);
    default: return epi ? launch_rows<5, 1>(a, st) : launch_rows<5, 0>(a, st);
  }
}

// Split-fp16 implicit-GEMM convolution, LDS-DMA variant: the kernel behind dlip_conv_nhwc_f16x3 when
// the activations arrive in the split activation format (DLIP_SPLIT_IN), behind dlip_conv2_nhwc_f16x3
// (a second reduction source: the 1x1 strided shortcut convolution of a down-sampling BasicBlock folded
// into conv2's reduction) and behind dlip_conv_pool_f16x3 (segmented column sums instead of the output).
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
//     COUNTED s_waitcnt vmcnt (the newest slices stay in flight across the barrier).  The eight-wave
//     256x128 tile instead runs its two halves of waves half a slice apart (PING-PONG, two barriers per
//     slice: one half multiplies while the other reads fragments, issues pieces and waits) -- see its loop.
//   * The DMA is issued from inline asm: hipcc would otherwise order every later ds_read behind it
//     with vmcnt(0) (it cannot tell the stages apart) and serialise the ring.
//   * Matrix instruction: v_mfma_f32_16x16x32_f16, one per 32-channel slice and 16x16 block (the shape
//     that holds the higher clock in a dense loop on gfx950; the 32x32x16 twin, the window mode and the
//     weights-resident layer-1 kernel of round 1 were measured losers and are gone: DESIGN.md section 4).
// Fragment layout, accumulator initialisation and the LDS image are those of conv_igemm_f16x3.hip.
#include "conv_common.h"
#include "conv_dma_common.h"
#include "conv_dma_hooks.h"
#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>
#include <mutex>
#include <utility>

namespace {

// Balanced ("stream-K") work split: the launch's reduction work = tiles x nk slices is cut into G equal
// contiguous ranges, one per workgroup, G = the number of workgroups the chip holds at once.  A range
// covers whole tiles plus at most one partial tile at each end; a workgroup that computed only some
// slices of a tile either keeps its part in registers (if every other part of that tile has already been
// published: it is the finisher) or publishes it as an fp32 slab and takes a ticket on the tile's counter;
// the workgroup that finds the tile complete adds the other slabs IN PART ORDER (deterministic bits) and
// runs the epilogue.  Nobody waits for anybody, so residency and dispatch order cannot deadlock it.
//
// Visibility (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility", and
// cdna_hip_programming.md section 5 "Projection GEMM at M = 256" item 2, the write-through form):
//   producer: EVERY slab store is sc1 (write-through: the bytes leave the XCD's L2, so no agent release is
//     needed) -> every storing wave drains them with s_waitcnt vmcnt(0) -> workgroup barrier -> ONE lane
//     takes the ticket with a relaxed agent-scope fetch_add (it signals for all waves, behind the barrier);
//   consumer: the finisher learns that it is last from the value its own add returned, or from ONE relaxed
//     agent-scope load of the counter (the peek); that lane then executes an agent-scope ACQUIRE fence
//     (buffer_inv sc1: this CU's L1 drops any stale slab line) and waits for it, the workgroup passes a
//     barrier, and only then are the slabs loaded -- every one of those loads is sc1 as well, so the bytes
//     come from beyond the non-coherent L2 whichever XCD wrote them.
// With G = tiles every range is exactly one tile and no slab is ever written (the plain launch).
struct StreamK {
  long long iters;   // tiles * nk
  float* slabs;      // [2 * G][BM * BN] fp32
  int* counters;     // [tiles], zero between launches (the finisher resets its tile's word)
  int G;             // workgroups in the grid
  unsigned long long* span;   // NULL, or this launch's {first workgroup start, last workgroup end} in 100 MHz ticks (dlip_span_scope_*)
  int il_tiles;      // > 0: G = il_tiles * parts and neighbours in work order take the SAME part of DIFFERENT tiles (launch_one)
  int l2_local;      // experiment (dlip_debug_set(3, 3)): a tile whose parts all sit on ONE XCD hands its slabs over through that XCD's L2
  int reduce_later;  // every workgroup is one part of one tile (il_tiles mode) and only PUBLISHES it: slab_reduce_kernel, launched behind
                     // this kernel, adds a tile's parts in part order and runs the epilogue (launch_one: few tiles cut many ways)
  DLIP_LAB_STREAMK_FIELDS   // (conv_dma_hooks.h: nothing in the product build)
};

// EPI: what the epilogue does with act(acc / wscale + bias + residual) * post_scale + post_shift
//   0  fp32 rows of y;  1  split-format rows of y (reports |v| >= 65520 to the range-status word);
//   2  no y: per tile and row-group segment, the column sums of v and v^2 in fp64 (a.pool)
// DUAL: the reduction continues over a second source x2 (1x1, strided) after the taps of x.
// VAR: 0 in the product except bit 12 (filters of more than 32 taps: below).  The lab build (conv_dma_hooks.h) uses bit 2 (every
// wave issues its pieces at the top of the slice), 6 (256x256 tile), 7 (window mode), 9 / 11 (ping-pong without priority / with the
// group-major MFMA order), 10 (the lock-step loop on 256x128: the ping-pong loop's reference).  The experiments of rounds 2 / 3 that
// lost -- a static priority for the second half of the waves, that half issuing half a slice late, the halves taking turns as a
// slice's issuer, the pieces spread one by one behind MFMA quarter-groups -- are in DESIGN.md section 4 with their numbers and in
// the history of this file, no longer in its text.
template <int BM, int BN, int WAVES_M, int WAVES_N, int EPI, int NSTAGE, int OCC, bool DUAL, int VAR = 0>
__global__ __launch_bounds__(64 * WAVES_M* WAVES_N, OCC) void conv_igemm_f16x3_dma_kernel(const ConvArgs a, const StreamK sk) {
  constexpr bool OSPLIT = EPI == 1;
  constexpr int NW = WAVES_M * WAVES_N, NT = 64 * NW;
  constexpr int RPP = NT / 8;   // rows one pass of the workgroup covers (8 lanes x 16 B per row)
  static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must be whole passes");
  static_assert(NSTAGE == 2 || NSTAGE == 3, "ring depth");
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int FR = 16;               // rows of one MFMA fragment (v_mfma_f32_16x16x32_f16)
  constexpr int MI = WM / FR, NI = WN / FR;
  constexpr int A_PER = BM / RPP, B_PER = BN / RPP;
  constexpr int NL = A_PER + B_PER;   // DMA instructions per wave per slice
  constexpr int STAGE_B = (BM + BN) * ROWB;
  constexpr int LDK = 32;             // dwords per LDS row
  constexpr int PF = NSTAGE - 1;      // slices in flight ahead of the one being multiplied
#include "conv_dma_hook_consts.inc"   // product: `constexpr bool WIN = false; constexpr int RING_W = 0;`
  constexpr int RING = WIN ? RING_W : NSTAGE * STAGE_B;   // bytes; the epilogue parameter table (5 x BN floats) sits behind it
  // The eight-wave 256x128 tile runs the ping-pong main loop (below); VAR bit 10 keeps the lock-step loop (lab reference),
  // bit 9 drops the matrix phase's priority (lab), bit 11 keeps the group-major MFMA order inside the matrix phase (lab).
  constexpr bool PINGPONG = BM == 256 && BN == 128 && NW == 8 && NSTAGE == 3 && (VAR & ~(512 | 2048 | 4096)) == 0;
  // VAR bit 12: filters with MORE THAN 32 TAPS.  The product's per-row validity mask has one bit per tap (R S <= 32); a weight
  // gradient run as a convolution -- input x as [C][H][W][N], "filter" = the output gradient as [K][Ho][Wo][N], the images as the
  // reduction's channels (deeplip_amd.autograd_video.wgrad_as_conv) -- has Ho x Wo taps (484 on layer 1).  This variant keeps each
  // row's window origin (hi0, wi0) in registers and tests a tap's row / column against the image when the piece is issued.
  constexpr bool BIGTAPS = (VAR & 4096) != 0;
  constexpr bool PP_PRIO = (VAR & 512) == 0;
  constexpr bool PP_ACC_MAJOR = (VAR & 2048) == 0;   // (bit 11: the group-major MFMA order, lab reference)
  constexpr int AQ = A_PER, BQ = B_PER;   // rows a lane addresses: its own passes
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int g_hw = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);   // neighbours in work order share an XCD (L2)
  // Few tiles with reductions of thousands of slices (weight gradients run as convolutions): workgroup (tile t, part p) is
  // number t * parts + p of the split, but in WORK order the tiles of one part sit side by side -- they read the same slices of
  // the streamed "filter" (the output-gradient map, cold in every slice) at the same time on the same XCD, once instead of
  // once per tile.
  const int g = sk.il_tiles > 0 ? (g_hw % sk.il_tiles) * (nwg / sk.il_tiles) + g_hw / sk.il_tiles : g_hw;

  const int tid = threadIdx.x;
  const int cq = tid & 7;
  const int rbase = tid >> 3;
  const int key_st = (rbase >> 1) & 7;          // RPP is a multiple of 16: the key is the same in every pass
  const int csrc = ((cq ^ key_st) << 2);        // first channel (dword) of the chunk this lane fetches
  const u32x4 xr = make_rsrc_words(a.x, a.x_bytes);
  const u32x4 wr = make_rsrc_words(a.w, a.w_bytes);
  u32x4 x2r = xr;
  if constexpr (DUAL) x2r = make_rsrc_words(a.x2, a.x2_bytes);

  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const uint32_t piece0 = lds0 + wave * 8 * ROWB;   // this wave's 8 rows of pass 0, stage 0, operand A
  const int lane = tid & 63;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  // lane = (row 0..15, k group 0..3); the one step of a slice reads hi chunk kgroup, lo chunk 4 + kgroup
  const int lrow = lane & (FR - 1), half = lane / FR;
  const int a_frag = (wm * WM + lrow) * LDK;
  const int b_frag = BM * LDK + (wn * WN + lrow) * LDK;
  const int key_rd = (lrow >> 1) & 7;   // fragment blocks start at multiples of 16 rows: the key depends on lrow only
  const int khi = (half ^ key_rd) << 2, klo = ((4 + half) ^ key_rd) << 2;
  const int x_dr = a.dh * a.W * a.ldx * 4, x_ds = a.dw * a.ldx * 4;
  const int ntaps = a.R * a.S;
  const int nk1 = a.nk - (DUAL ? a.nk2 : 0);   // slices of the first source

  // PAIRED column blocks (a.n_inner == 2; round 5): the launch's P = tiles_n column blocks are P LANES of the split -- workgroup g
  // works on column block g % P only, and the (row tile, slice) space of that lane is cut into G / P equal ranges, range g / P.  The P
  // workgroups of a range are neighbours in work order (one XCD), walk the SAME activation rows at the same time, and so fetch them
  // into that XCD's L2 once; with the column block merely INNER (n_inner == 1) one workgroup ran a row block's column tiles one after
  // the other, ~80 us apart, and the rows had left L2 in between (rocprofv3 FETCH_SIZE: the activations of layer 3 read twice).
  // gv / Gv / itv = workgroup index, grid and slice space of ONE lane (== g, sk.G, sk.iters when P == 1).  They are formed where
  // they are needed -- here, at a segment's tile decode and in the (rare) hand-off -- from an opaque copy of g, so that none of them
  // lives across the main loop: carried through it (the first version) they cost every 256 x 128 launch 4 - 10 % in scalar spills.
  long long it_begin, it_end;
  {
    const int P = a.n_inner == 2 ? a.tiles_n : 1;
    const int gv = P > 1 ? g / P : g, Gv = P > 1 ? sk.G / P : sk.G;
    const long long itv = P > 1 ? sk.iters / P : sk.iters;
    it_begin = (long long)gv * itv / Gv;
    it_end = (long long)(gv + 1) * itv / Gv;
  }
  // In-kernel span of this launch (measurement only, off unless a span scope is open: one scalar test; dlip_common.h): what a
  // replayed step plan cannot give the host (no event can be read back from a graph) the kernel notes itself.
  dlip_span_enter(sk.span, g);
  DLIP_LAB_WG_STAMP(8, true);
  for (long long it = it_begin; it < it_end;) {
    const int tile = (int)(it / a.nk);
    const int k0 = (int)(it - (long long)tile * a.nk);
    const int kn = (int)((it_end - it) < (long long)(a.nk - k0) ? (it_end - it) : (long long)(a.nk - k0));
    // Tile order (chosen per launch, launch_one): output-channel block OUTER -- a workgroup's range, and (through the
    // XCD remap of g) an XCD's eighth of the launch, stays on one 128-channel weight block while the activation rows
    // stream through once per block -- or INNER (a.n_inner) when the whole filter bank fits the XCD's L2.
    const int tiles_m = (a.M + BM - 1) / BM;
    int tile_n = tile / tiles_m;
    int tile_m = tile - tile_n * tiles_m;
    if (a.n_inner == 2) {   // paired lanes: `tile` counts the row tiles of this workgroup's column block g % P
      int g_o = g;
      asm volatile("" : "+s"(g_o));   // (opaque: the division below is redone per segment instead of living across the loop)
      tile_m = tile;
      tile_n = g_o - (g_o / a.tiles_n) * a.tiles_n;
    } else if (a.n_inner) {   // column block INNER: consecutive tiles (one workgroup's range, one XCD's eighth) share their activation rows
      tile_m = tile / a.tiles_n;
      tile_n = tile - tile_m * a.tiles_n;
    }
    // every wave is done reading the previous segment's last stage (and the ticket word) before the ring is refilled
    if (it != it_begin) __syncthreads();
    DLIP_STAMP(0);
    DLIP_LAB_WG_STAMP(7, it == it_begin);

    f32x4 acc[MI][NI];
#define DLIP_FENCE() __builtin_amdgcn_sched_barrier(0)
    if constexpr (WIN) {
#include "conv_dma_hook_window.inc"
    } else {
    // Per-row gather state, branch-free: byte offset of the row's window origin and a bit per filter tap
    // that stays inside the image (columns and rows tested separately: R + S steps, not R x S).
    int a_off[AQ];
    uint32_t a_mask[AQ];         // after set-up: INVERTED (bit tap set = that tap falls outside the image, or the row is past M)
    int a_h0[BIGTAPS ? AQ : 1], a_w0[BIGTAPS ? AQ : 1];   // (BIGTAPS) window origin of the row; rows past M: far outside
    uint32_t a2_off[DUAL ? AQ : 1];   // second source: byte offset of the row's pixel; the out-of-range offset itself past M
    auto row_of = [&](int j) { return rbase + RPP * j; };
    {
      int hi0[AQ], wi0[AQ];
      uint32_t colbits[AQ];
#pragma unroll
      for (int j = 0; j < AQ; ++j) {
        const int m = tile_m * BM + row_of(j);
        const int mc = m < a.M ? m : a.M - 1;
        const int n = dlip_div(mc, a.div_howo);
        const int rem = mc - n * a.HoWo;
        const int ho = dlip_div(rem, a.div_wo);
        const int wo = rem - ho * a.Wo;
        hi0[j] = ho * a.sh - a.ph;
        wi0[j] = wo * a.sw - a.pw;
        a_off[j] = (((n * (BIGTAPS ? a.Hs : a.H) + hi0[j]) * a.W + wi0[j]) * a.ldx + csrc) * 4;
        if constexpr (DUAL) a2_off[j] = m < a.M ? (uint32_t)((((n * a.H2 + ho * a.s2h) * a.W2 + wo * a.s2w) * a.ldx2 + csrc) * 4) : DLIP_OOB_OFFSET;
        colbits[j] = 0u;
        a_mask[j] = 0u;
      }
      if constexpr (BIGTAPS) {
#pragma unroll
        for (int j = 0; j < AQ; ++j) {
          a_h0[j] = tile_m * BM + row_of(j) < a.M ? hi0[j] : -(1 << 28);
          a_w0[j] = wi0[j];
        }
      } else {
        for (int sx = 0; sx < a.S; ++sx)
#pragma unroll
          for (int j = 0; j < AQ; ++j) colbits[j] |= (uint32_t)((unsigned)(wi0[j] + sx * a.dw) < (unsigned)a.W) << sx;
        for (int r = 0; r < a.R; ++r)
#pragma unroll
          for (int j = 0; j < AQ; ++j)
            a_mask[j] |= ((unsigned)(hi0[j] + r * a.dh) < (unsigned)a.H ? colbits[j] : 0u) << (r * a.S);
#pragma unroll
        for (int j = 0; j < AQ; ++j)
          a_mask[j] = tile_m * BM + row_of(j) >= a.M ? ~0u : ~a_mask[j];
      }
    }
    // (a weight row past K carries the out-of-range offset itself: 2^31 plus any tap offset stays out of range -- one add per piece)
    uint32_t b_off[BQ];
#pragma unroll
    for (int j = 0; j < BQ; ++j) {
      const int n = tile_n * BN + rbase + RPP * j;
      b_off[j] = n < a.K ? (uint32_t)((n * a.rsc + csrc) * 4) : DLIP_OOB_OFFSET;
    }

    // Reduction walk: 32-channel slice OUTER, filter tap INNER (conv_igemm_f16x3.hip), entered at slice k0.
    // DUAL: c0 runs on past the a.Cw channels of the first source -- slice nk1 + i is channels 32 i .. of x2,
    // whose weights sit behind the R S Cw tap weights of each output channel's row.
    int c0 = (k0 / ntaps) * BK, tap = k0 % ntaps;
    if constexpr (DUAL) {
      if (k0 >= nk1) { c0 = a.Cw + (k0 - nk1) * BK; tap = 0; }
    }
    int s_pos = tap % a.S, x_row = (tap / a.S) * x_dr;
    int tap_dh = (tap / a.S) * a.dh, tap_dw = s_pos * a.dw;   // (BIGTAPS) the tap's row / column offset in pixels
    int x_tap = x_row + s_pos * x_ds + c0 * 4, w_tap = (tap * a.Cw + c0) * 4;
    if constexpr (BIGTAPS) { x_tap = x_row + s_pos * x_ds + (c0 / BK) * a.cs_x; w_tap = tap * a.wt + (c0 / BK) * a.cs_w; }
    if constexpr (DUAL) {
      if (c0 >= a.Cw) { x_tap = (c0 - a.Cw) * 4; w_tap = (ntaps * a.Cw + c0 - a.Cw) * 4; }
    }
    auto advance = [&]() {
      if (DUAL && c0 >= a.Cw) {          // second source: one slice per 32 channels
        c0 += BK;
      } else {
        ++tap;
        if (++s_pos == a.S) { s_pos = 0; x_row += x_dr; if constexpr (BIGTAPS) tap_dh += a.dh; }
        if (tap == ntaps) { tap = 0; s_pos = 0; x_row = 0; c0 += BK; if constexpr (BIGTAPS) tap_dh = 0; }
        if constexpr (BIGTAPS) tap_dw = s_pos * a.dw;
      }
      x_tap = x_row + s_pos * x_ds + c0 * 4;
      w_tap = (tap * a.Cw + c0) * 4;
      if constexpr (BIGTAPS) { x_tap = x_row + s_pos * x_ds + (c0 / BK) * a.cs_x; w_tap = tap * a.wt + (c0 / BK) * a.cs_w; }
      if (DUAL && c0 >= a.Cw) { x_tap = (c0 - a.Cw) * 4; w_tap = (ntaps * a.Cw + c0 - a.Cw) * 4; }
    };
    auto tap_ok = [&](int j) -> bool {
      if constexpr (BIGTAPS) return (unsigned)(a_h0[j] + tap_dh) < (unsigned)a.H && (unsigned)(a_w0[j] + tap_dw) < (unsigned)a.W;
      else return ((a_mask[j] >> tap) & 1u) == 0u;
    };
    // this wave's A_PER + B_PER pieces of the slice the walk stands on -> ring stage `stage`
    auto issue_a = [&](int stage) {
      const uint32_t base = piece0 + stage * STAGE_B;
      if (DUAL && c0 >= a.Cw) {          // (wave-uniform)
#pragma unroll
        for (int j = 0; j < A_PER; ++j)
          dma_piece(x2r, a2_off[DUAL ? j : 0] + (uint32_t)x_tap, base + j * RPP * ROWB);
        return;
      }
#pragma unroll
      for (int j = 0; j < A_PER; ++j) {
        if constexpr (BIGTAPS) {
          dma_piece(xr, tap_ok(j) ? (uint32_t)(a_off[j] + x_tap) : DLIP_OOB_OFFSET, base + j * RPP * ROWB);
        } else {
          // a tap outside the image: its bit of the inverted mask, shifted to bit 31, ORed into the offset -- beyond every buffer
          // whatever the sum was (add, shift, and-or: the compare + select form was five vector instructions per piece in the
          // load phase, which is the long pole of a ping-pong slice)
          const uint32_t oob = (a_mask[j] << (31 - tap)) & 0x80000000u;
          dma_piece(xr, (uint32_t)(a_off[j] + x_tap) | oob, base + j * RPP * ROWB);
        }
      }
    };
    auto issue_b = [&](int stage) {
      const uint32_t base = piece0 + stage * STAGE_B + BM * ROWB;
#pragma unroll
      for (int j = 0; j < B_PER; ++j)
        dma_piece(wr, b_off[j] + (uint32_t)w_tap, base + j * RPP * ROWB);
    };

    // ---- prologue: put the first NSTAGE-1 slices in flight, then initialise the accumulators ----
    DLIP_STAMP(1);
    issue_a(0);
    issue_b(0);
    if (PF > 1 && kn > 1) {
      advance();
      issue_a(1);
      issue_b(1);
    }

    // Accumulators hold the TRANSPOSED tile (rows = output channels, columns = pixels: the weight
    // fragment is the MFMA's A operand), so a lane owns 4 consecutive channels of one pixel per
    // register quad: 8-B (hi) + 8-B (lo) pieces of an output row for the LDS-staged epilogue.
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[mi][ni][e] = 0.f;
    // Per-channel epilogue parameters of this tile's BN channels -> the LDS table behind the ring (read back
    // as 16-B quads): 1/wscale, bias, slope, post scale, post shift.
    if (tid < BN) {
      const int k = tile_n * BN + tid;
      const bool kok = k < a.K;
      float* tab = smem + RING / 4;
      tab[tid] = kok ? 1.f / a.wscale[k] : 0.f;   // power of two: exact
      tab[BN + tid] = (kok && a.bias) ? a.bias[k] : 0.f;
      tab[2 * BN + tid] = (kok && a.slope) ? a.slope[k] : 1.f;
      tab[3 * BN + tid] = (kok && a.pscale) ? a.pscale[k] : 1.f;
      tab[4 * BN + tid] = (kok && a.pshift) ? a.pshift[k] : 0.f;
    }

    // ---- main loop: per slice 3 groups of MI x NI instructions.  Group order lo*hi, hi*hi, hi*lo:
    // the last group needs neither the activation-lo nor the weight-hi fragments, so the NEXT slice's first
    // group's fragments are read (behind the barrier) into registers the tail of this slice does not use. ----
#include "conv_dma_hook_tile256.inc"   // product: nothing
    if constexpr (PINGPONG) {
      // PING-PONG main loop (round 3; the eight-wave 256x128 tile).  The loop below (still what the four-wave tiles run) keeps the
      // two waves of a SIMD in LOCK STEP: both read their
      // fragments, both issue their 48 MFMAs (sharing the matrix pipe: 1 536 cycles for the pair), both wait at vmcnt and at the
      // slice's barrier together -- ~650 of a slice's 2 520 cycles in which neither issues an MFMA (in-kernel stamps, DESIGN.md
      // section 4).  Here the workgroup's two halves -- waves 0-3 ("A") and 4-7 ("B"), one wave of every SIMD in each
      // (MI355X_MICROARCH.md, item 9: split SIMD partners by wave number >= NW/2) -- run the SAME slice half a period apart, two
      // barriers per slice: while one half issues its 48 MFMAs alone on its SIMDs (768 cycles, nothing else in its stream), the
      // other does everything that is not matrix work: the 16 fragment reads of its next MFMA phase, its 6 LDS-DMA pieces two
      // slices ahead, its vmcnt wait.  Matrix beside memory in every interval (item 5: a rendezvous pays when the paired
      // intervals are complementary).
      //   interval between barriers   b(2s-1) .. b(2s)      |  b(2s) .. b(2s+1)
      //   half A                      LOAD(s)               |  MFMA(s), wait slice s+1
      //   half B                      MFMA(s-1)             |  LOAD(s), wait slice s+1
      //   LOAD(s) = issue this wave's pieces of slice s+2 (stage (s+2) % 3 = that of slice s-1, whose last reader -- B's LOAD(s-1)
      //             -- finished before b(2s-1)), read every fragment of slice s (stage s % 3) into registers, lgkmcnt(0).
      //   "wait slice s+1": this wave's pieces of slice s+1 have landed (vmcnt leaves only slice s+2's NL younger ones), so that
      //             behind the next barrier the whole stage of slice s+1 is complete for whichever half reads it first.
      // Accumulation order per accumulator is the product's (lo*hi, hi*hi, hi*lo per slice): bit-identical results.
      static_assert(NW == 8 && NSTAGE == 3, "ping-pong: eight waves, three stages");
      const bool half_b = wave >= NW / 2;            // (wave-uniform)
      f16x8 fal[MI], fah[MI], fbh[NI], fbl[NI];
      auto read_all = [&](int stage) {
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
      auto mfma_all = [&]() {
        if constexpr (PP_PRIO) __builtin_amdgcn_s_setprio(2);   // the matrix phase outranks its SIMD partner's load phase
        if constexpr (PP_ACC_MAJOR) {
          // ACCUMULATOR-MAJOR: the three products of one accumulator back to back.  A wave that is ALONE on its SIMD's matrix pipe
          // -- as it is here -- issues v_mfma_f32_16x16x32_f16 every 16.4 cycles when consecutive instructions accumulate into the
          // same registers, and only every 24.4 when each goes to another accumulator (tools/probes/mfma_rate.hip: 16 accumulators
          // group by group 24.43, one chain 16.58, chains of three 16.42 cycles per MFMA; two waves per SIMD reach 16.3 either
          // way, which is why the lock-step loop never saw it).  Per accumulator the order is still lo*hi, hi*hi, hi*lo: same bits.
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fbh[ni], fal[mi], acc[mi][ni], 0, 0, 0);
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fbh[ni], fah[mi], acc[mi][ni], 0, 0, 0);
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fbl[ni], fah[mi], acc[mi][ni], 0, 0, 0);
            }
            DLIP_FENCE();
          }
        } else {
#pragma unroll
        for (int grp = 0; grp < 3; ++grp) {
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
              const f16x8 av = grp == 0 ? fal[mi] : fah[mi];
              const f16x8 bv = grp == 2 ? fbl[ni] : fbh[ni];
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bv, av, acc[mi][ni], 0, 0, 0);
            }
          DLIP_FENCE();
        }
        }
        if constexpr (PP_PRIO) __builtin_amdgcn_s_setprio(0);
      };
      int st_iss = (PF > 1 && kn > 1) ? 2 % NSTAGE : 1 % NSTAGE;   // stage the next issue goes to (the prologue issued slices 0, 1)
      auto load_phase = [&](int s) {
        read_all(s % NSTAGE);                        // the reads first: their latency passes under the piece issue below
        DLIP_FENCE();
        if (s + 2 < kn) {
          advance();
          issue_a(st_iss); issue_b(st_iss);
          st_iss = st_iss + 1 == NSTAGE ? 0 : st_iss + 1;
        }
        DLIP_FENCE();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the stage may be refilled behind the next barrier: the reads are done
        DLIP_FENCE();
      };
      auto wait_next = [&](int s) {                  // this wave's pieces of slice s + 1 have landed
        if (s + 1 < kn) { if (s + 2 < kn) wait_vmcnt<NL>(); else wait_vmcnt<0>(); }
      };
      DLIP_STAMP(2);
      if (PF > 1 && kn > 1) wait_vmcnt<NL>(); else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();                  // slice 0 is complete
      DLIP_STAMP(3);
      if (!half_b) {
        for (int s = 0; s < kn; ++s) {
          const int kt = s;   // (the slice stamps name the slice `kt`)
          (void)kt;
          DLIP_SSTAMP(0);
          load_phase(s);
          DLIP_SSTAMP(1);
          __builtin_amdgcn_s_barrier();              // b(2s)
          DLIP_SSTAMP(2);
          mfma_all();
          DLIP_SSTAMP(3);
          DLIP_SSTAMP(4);
          __builtin_amdgcn_s_barrier();              // b(2s+1)
          DLIP_SSTAMP(5);
          // half A is the FIRST reader of slice s + 1 (right here, in its next load phase): its own pieces of that slice need
          // to have landed before ITS reads, not before the barrier -- the wait sits behind it (half B reads an interval later)
          wait_next(s);
          DLIP_SSTAMP(6);
        }
      } else {
        __builtin_amdgcn_s_barrier();                // b(0): half A reads slice 0 first
        for (int s = 0; s < kn; ++s) {
          const int kt = s;
          (void)kt;
          DLIP_BSTAMP(0);
          load_phase(s);
          DLIP_BSTAMP(1);
          wait_next(s);
          DLIP_BSTAMP(2);
          __builtin_amdgcn_s_barrier();              // b(2s+1)
          DLIP_BSTAMP(3);
          mfma_all();
          DLIP_BSTAMP(4);
          if (s + 1 < kn) __builtin_amdgcn_s_barrier();   // b(2s+2)
          DLIP_BSTAMP(5);
        }
      }
    } else
    {
      f16x8 fal[MI], fah[MI], fbh[NI], fbl[NI];
      auto read_first = [&](int stage) {   // what group 0 needs: activation lo, weight hi
        const float* Aw = smem + stage * (STAGE_B / 4) + a_frag;
        const float* Bw = smem + stage * (STAGE_B / 4) + b_frag;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) fal[mi] = *reinterpret_cast<const f16x8*>(Aw + mi * 16 * LDK + klo);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) fbh[ni] = *reinterpret_cast<const f16x8*>(Bw + ni * 16 * LDK + khi);
      };
      auto read_rest = [&](int stage) {    // activation hi, weight lo
        const float* Aw = smem + stage * (STAGE_B / 4) + a_frag;
        const float* Bw = smem + stage * (STAGE_B / 4) + b_frag;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) fah[mi] = *reinterpret_cast<const f16x8*>(Aw + mi * 16 * LDK + khi);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) fbl[ni] = *reinterpret_cast<const f16x8*>(Bw + ni * 16 * LDK + klo);
      };
      // grp 0: lo*hi, 1: hi*hi, 2: hi*lo; activation blocks [m0, m1)
      auto mfma_p = [&](int grp, int m0, int m1) {
#pragma unroll
        for (int mi = m0; mi < m1; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            const f16x8 av = grp == 0 ? fal[mi] : fah[mi];
            const f16x8 bv = grp == 2 ? fbl[ni] : fbh[ni];
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bv, av, acc[mi][ni], 0, 0, 0);
          }
      };
      constexpr int MH = MI / 2 > 0 ? MI / 2 : 1;   // the part of the last group issued before the barrier
      // slice 0 has landed once at most the (PF - 1) younger slices are outstanding
      DLIP_STAMP(2);
      if (PF > 1 && kn > 1) wait_vmcnt<NL>(); else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      DLIP_STAMP(3);
      read_first(0);

      int st_cur = 0, st_iss = (PF > 1 && kn > 1) ? 2 % NSTAGE : 1 % NSTAGE;   // stage the next issue goes to
      for (int kt = 0; kt < kn; ++kt) {
        const bool more1 = (kt + 1) < kn, moreP = (kt + PF) < kn;
        const int st_nxt = st_cur + 1 == NSTAGE ? 0 : st_cur + 1;
        // WHERE the slice's LDS-DMA pieces go out.  At the very top of the slice, right behind the barrier that frees their stage
        // (`early`), a layer that runs back to back takes 2-3 % less time on 128x128 and the two-stage 128x64 than with the pieces in
        // the MFMA shadow of groups 0 / 1 (tools/bench_dma.py; -6.5 % on 128x128's launches in the step); the three-stage 128x64
        // never had it (+1 %).  (The 256x128 tile runs the ping-pong loop above; its lock-step form, the lab build's reference,
        // keeps its pieces in the shadow: at the board's power limit the burst at the top cost the step 1.3 %.)
        constexpr bool early = (VAR & 4) != 0 || (BM == 128 && BN == 128) || (BM == 128 && BN == 64 && NSTAGE == 2);
        DLIP_SSTAMP(0);
        if (early && moreP) {
          advance();
          issue_a(st_iss); issue_b(st_iss);
          st_iss = st_iss + 1 == NSTAGE ? 0 : st_iss + 1;
        }
        DLIP_FENCE();
        DLIP_SSTAMP(1);
        read_rest(st_cur); DLIP_FENCE();
        mfma_p(0, 0, MI); DLIP_FENCE();
        DLIP_SSTAMP(2);
        if (moreP && !early) { advance(); issue_a(st_iss); } DLIP_FENCE();
        mfma_p(1, 0, MH); DLIP_FENCE();
        if (moreP && !early) { issue_b(st_iss); st_iss = st_iss + 1 == NSTAGE ? 0 : st_iss + 1; } DLIP_FENCE();
        if (MH < MI) mfma_p(1, MH, MI);
        DLIP_FENCE();
        // (the barrier one half-group earlier, behind group 1 -- what the window kernel does -- measured 1-7 % SLOWER here)
        mfma_p(2, 0, MH); DLIP_FENCE();
        DLIP_SSTAMP(3);
        if (more1) {
          // slice kt+1 must have landed (every wave's share: wait, then barrier); slices beyond it stay in flight
          if (PF > 1 && (kt + 2) < kn) wait_vmcnt<NL>(); else wait_vmcnt<0>();
          DLIP_SSTAMP(4);
          __builtin_amdgcn_s_barrier();
          DLIP_SSTAMP(5);
          read_first(st_nxt);
        }
        DLIP_FENCE();
        if (MH < MI) mfma_p(2, MH, MI);
        DLIP_FENCE();
        DLIP_SSTAMP(6);
        st_cur = st_nxt;
      }
    }
    }
#undef DLIP_FENCE
    DLIP_STAMP(4);

    // Lane coordinates re-derived behind an opaque asm: otherwise the compiler hoists every address of the
    // hand-off and epilogue code (invariant across segments) out of the segment loop and carries ~100
    // values through the MFMA loop in scratch.
    int tid_e = tid;
    asm volatile("" : "+v"(tid_e));
    const int lane_e = tid_e & 63, lrow_e = tid_e & (FR - 1), half_e = (tid_e & 63) / FR;   // pixel in block; channel quad (k group)

    bool finish = true;
    if (kn != a.nk) {   // split tile (workgroup-uniform branch)
      constexpr int SLAB = BM * BN;   // floats
      volatile int* bcast = reinterpret_cast<volatile int*>(smem);
      const long long t0 = (long long)tile * a.nk;
      // (paired lanes) this lane's workgroup index / grid / slice space, and the tile's ticket word = its number in tile_m-major order
      const int P = a.n_inner == 2 ? a.tiles_n : 1, lane_n = P > 1 ? tile_n : 0;
      const int gv = P > 1 ? g / P : g, Gv = P > 1 ? sk.G / P : sk.G;
      const long long itv = P > 1 ? sk.iters / P : sk.iters;
      const int tile_id = P > 1 ? tile * P + lane_n : tile;
      const int gf = (int)(((t0 + 1) * Gv - 1) / itv);                   // owner of the tile's first slice (index within the lane)
      const int gl = (int)(((t0 + a.nk) * Gv - 1) / itv);                // owner of its last slice
      const int others = gl - gf;                                        // parts besides this one
      // (experiment, round 4, off in the product: VERDICT item "slab traffic") parts on one XCD -- neighbours in work order are,
      // except across the 8 XCD boundaries -- share an L2: plain stores (they stop in L2) and sc0 loads (past this CU's L1) would do
      bool l2_local = false;
      if (sk.l2_local && sk.il_tiles == 0 && q8 > 0) {
        const int big = r8 * (q8 + 1);
        const int pf = gf * P + lane_n, pl = gl * P + lane_n;
        const int xf = pf < big ? pf / (q8 + 1) : r8 + (pf - big) / q8, xl = pl < big ? pl / (q8 + 1) : r8 + (pl - big) / q8;
        l2_local = xf == xl;
      }
      if (sk.reduce_later) {   // (workgroup-uniform) publish and leave: the kernel boundary makes the slab visible to the reduce launch
        const __amdgpu_buffer_rsrc_t sr = dlip_make_rsrc(sk.slabs + (size_t)(2 * g) * SLAB, SLAB * 4);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[mi][ni]), sr, ((mi * NI + ni) * NT + tid_e) * 16, 0, 0);
        finish = false;
      } else {
      // 1. peek: if every other part has already published, this workgroup is the finisher and keeps its
      //    part in registers (the usual case for a range's final, head-of-tile segment: the neighbour
      //    computed the rest of that tile first thing).
      __syncthreads();   // all waves are past their last fragment reads: LDS word 0 is free
      if (tid_e == 0) bcast[0] = __hip_atomic_load(sk.counters + tile_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      finish = bcast[0] == others;
      if (!finish) {
        // 2. publish: write-through (sc1) slab stores, drained by every storing wave, barrier, then ONE ticket
        const __amdgpu_buffer_rsrc_t sr = dlip_make_rsrc(sk.slabs + (size_t)(2 * g + (it != it_begin ? 1 : 0)) * SLAB, SLAB * 4);
        if (l2_local) {
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[mi][ni]), sr, ((mi * NI + ni) * NT + tid_e) * 16, 0, 0);
        } else {
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[mi][ni]), sr, ((mi * NI + ni) * NT + tid_e) * 16, 0, 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid_e == 0) bcast[0] = __hip_atomic_fetch_add(sk.counters + tile_id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        finish = bcast[0] == others;   // the other parts arrived between the peek and the ticket
      }
      if (finish) {
        // 3. acquire on this CU (one lane, then the wait, then the barrier every loading wave passes)
        if (tid_e == 0) {
          __hip_atomic_store(sk.counters + tile_id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // sum the parts in part order (own part from registers): the bits do not depend on who came last.  Row block by row
        // block, in place: one row of NI pieces in flight (one wait per row and part; a part cost ~2 us of serial latency when
        // every piece was waited for singly), and only 2 NI quads of registers beside the accumulators (a full second copy of
        // the tile, as before, does not fit once the accumulators are half the register file).  Same sums, same order.
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          f32x4 t[NI];
          for (int p = gf; p <= gl; ++p) {
            const long long pb = (long long)p * itv / Gv;
            const __amdgpu_buffer_rsrc_t pr = dlip_make_rsrc(sk.slabs + (size_t)(2 * (p * P + lane_n) + (pb < t0 ? 1 : 0)) * SLAB, SLAB * 4);
            f32x4 v[NI];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
              v[ni] = acc[mi][ni];
              if (p != gv) {
                if (l2_local) v[ni] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(pr, ((mi * NI + ni) * NT + tid_e) * 16, 0, 1));
                else v[ni] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(pr, ((mi * NI + ni) * NT + tid_e) * 16, 0, 16));
              }
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
              for (int c = 0; c < 4; ++c) t[ni][c] = p == gf ? v[ni][c] : t[ni][c] + v[ni][c];
            __builtin_amdgcn_sched_barrier(0);
          }
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = t[ni];
        }
      }
      }   // (!reduce_later)
    }

    if (finish) {
      // ---- epilogue through LDS: y = act(acc / wscale + bias + residual) * post_scale + post_shift ----
      // The ring is free now and stages the output tile in its memory layout (BM rows x BN*4 bytes, in
      // EPASS row bands when the tile is larger than the ring), 16-B chunk c of row r at position
      // c ^ (r & 15) (low 4 bits): the residual arrives by LDS-DMA, every lane adds / overwrites the 8-B
      // pieces of its own pixels, and the tile leaves in 16-B stores (a row is one contiguous segment).
      // (Measured and rejected in round 2: the residual fetched as 8-B pieces straight into the accumulators
      // during the prologue -- no epilogue fetch, wait or barrier -- was 2-7 % SLOWER on every layer with a
      // residual: 32 scattered 8-B loads per lane cost the address unit more than 8 row-contiguous DMA pieces.)
      typedef _Float16 h4 __attribute__((ext_vector_type(4)));
      constexpr int PITCH = BN * 4;                  // bytes per image row
      constexpr int CPR = BN / 4;                    // 16-B chunks per row
      constexpr int EPASS = (BM * PITCH + RING - 1) / RING;
      constexpr int PROWS = BM / EPASS;              // rows per band
      static_assert(BM % EPASS == 0 && PROWS % WM == 0 && PROWS * PITCH <= RING, "epilogue bands are whole wave rows");
      static_assert((PROWS * PITCH) % (NW * 1024) == 0, "a band is a whole number of DMA pieces per wave");
      static_assert(EPI != 2 || EPASS == 1, "the pooled epilogue reduces one whole-tile image");
      constexpr int RES_PIECES = PROWS * PITCH / 1024 / NW;   // per wave
      constexpr int RPQ = 1024 / PITCH;              // rows per DMA piece
      const u32x4 rrw = make_rsrc_words(a.res, a.res ? a.r_bytes : 0u);
      const __amdgpu_buffer_rsrc_t yr = dlip_make_rsrc(a.y, a.y_bytes);
      const int kcol0 = tile_n * BN;                 // first output channel of the tile
      char* img = reinterpret_cast<char*>(smem);
      const f32x4* tab = reinterpret_cast<const f32x4*>(smem + RING / 4);
      const bool post = a.pscale != nullptr;
      float amax = 0.f;                              // largest |v| this lane converts to fp16 (EPI 1)
      __syncthreads();                               // every wave is done with the ring (and the table is written)
#pragma unroll
      for (int ep = 0; ep < EPASS; ++ep) {
        const int band0 = ep * PROWS;                // first tile row of this band
        if (a.res) {
#pragma unroll
          for (int i = 0; i < RES_PIECES; ++i) {
            const int piece = i * NW + wave;
            const int r = piece * RPQ + lane_e / CPR;  // band row
            const int pp = lane_e % CPR;
            const int c = (pp & ~15) | ((pp ^ r) & 15);
            const int m = tile_m * BM + band0 + r;
            const bool ok = m < a.M && (kcol0 + (c >> 3) * 32) < a.K;
            dma_piece(rrw, ok ? (uint32_t)((m * a.ldr + kcol0) * 4 + c * 16) : DLIP_OOB_OFFSET, lds0 + piece * 1024);
          }
          wait_vmcnt<0>();
          __syncthreads();
        }
        const bool mine = (wm * WM) / PROWS == ep;   // this wave's rows are in the band (wave-uniform)
        if (mine) {
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            const int kl = wn * WN + ni * FR + 4 * half_e;   // tile-local channel of acc[..][ni][0..3]
            const f32x4 inv4 = tab[kl >> 2], bi4 = tab[(BN + kl) >> 2], sl4 = tab[(2 * BN + kl) >> 2];
            f32x4 ps4 = {1.f, 1.f, 1.f, 1.f}, pt4 = {0.f, 0.f, 0.f, 0.f};
            if (post) { ps4 = tab[(3 * BN + kl) >> 2]; pt4 = tab[(4 * BN + kl) >> 2]; }
            const int ch = (kl >> 5) * 8 + ((kl >> 3) & 3);   // hi chunk of these 4 channels within the row (lo: + 4)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
              const int r = wm * WM - band0 + mi * FR + lrow_e;      // band row of this lane's pixel
              char* row = img + r * PITCH + 2 * (kl & 4);            // which 8-B half of the chunk
              const int phi = (ch & ~15) | ((ch ^ r) & 15), plo = ((ch + 4) & ~15) | (((ch + 4) ^ r) & 15);
              float v[4];
#pragma unroll
              for (int c = 0; c < 4; ++c) v[c] = acc[mi][ni][c] * inv4[c] + bi4[c];
              if (a.res) {
                const h4 rh = *reinterpret_cast<const h4*>(row + phi * 16), rl = *reinterpret_cast<const h4*>(row + plo * 16);
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] += (float)rh[c] + (float)rl[c];
              }
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                v[c] = v[c] >= 0.f ? v[c] : v[c] * sl4[c];
                if (post) v[c] = v[c] * ps4[c] + pt4[c];
              }
              if constexpr (OSPLIT) {   // same 8-B pieces the residual came from: no other lane touches them
                h4 hi, lo;
#pragma unroll
                for (int c = 0; c < 4; ++c) { hi[c] = (_Float16)v[c]; lo[c] = (_Float16)(v[c] - (float)hi[c]); }
                amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
                *reinterpret_cast<h4*>(row + phi * 16) = hi;
                *reinterpret_cast<h4*>(row + plo * 16) = lo;
              } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[mi][ni][c] = v[c];
              }
            }
            __builtin_amdgcn_sched_barrier(0);   // one channel quad at a time
          }
        }
        if constexpr (!OSPLIT) {
          if (a.res) __syncthreads();                // fp32 rows overwrite other lanes' residual pieces
          if (mine) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
              const int kl = wn * WN + ni * FR + 4 * half_e;
              const int ch = kl >> 2;              // the 4 channels are one fp32 chunk
#pragma unroll
              for (int mi = 0; mi < MI; ++mi) {
                const int r = wm * WM - band0 + mi * FR + lrow_e;
                const int pc = (ch & ~15) | ((ch ^ r) & 15);
                *reinterpret_cast<f32x4*>(img + r * PITCH + pc * 16) = acc[mi][ni];
              }
            }
          }
        }
        __syncthreads();
        if constexpr (EPI == 2) {
          // ---- pooled epilogue: column sums of v and v^2 over the tile's rows, split at the one row-group
          // boundary a tile can contain (pool_group >= BM), in fp64, fixed order -> a.pool[tile_m][seg][stat][k] ----
          constexpr int NG = NT / CPR;               // row groups of threads
          constexpr int RPG = BM / NG;               // rows per thread
          static_assert(NT % CPR == 0 && BM % NG == 0 && 4 * NG * BN * 8 <= RING, "pooled epilogue layout");
          const int cqd = tid_e % CPR, rg = tid_e / CPR;
          const int m0 = tile_m * BM;
          const int g0 = m0 / a.pool_group;
          const int rb = (g0 + 1) * a.pool_group - m0;   // first tile row of the next group
          // ragged batches (a.pool_len): a group's rows past its valid length are padding and stay out of both sums; e0 / e1 = the
          // tile-relative ends of the valid rows of segment 0 (group g0) and segment 1 (group g0 + 1)
          int e0 = rb, e1 = BM;
          if (a.pool_len.len != nullptr) {
            const int G = (a.M + a.pool_group - 1) / a.pool_group;
            e0 = min(rb, g0 * a.pool_group + dlip_valid_rows(a.pool_len, g0, a.pool_group) - m0);
            e1 = g0 + 1 < G ? rb + dlip_valid_rows(a.pool_len, g0 + 1, a.pool_group) : rb;
          }
          double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0}, q0[4] = {0, 0, 0, 0}, q1[4] = {0, 0, 0, 0};
          for (int i = 0; i < RPG; ++i) {
            const int r = rg * RPG + i;
            if (m0 + r >= a.M) break;
            const int pc = (cqd & ~15) | ((cqd ^ r) & 15);
            const f32x4 v = *reinterpret_cast<const f32x4*>(img + r * PITCH + pc * 16);
            if (r < rb) {
              if (r < e0) {
#pragma unroll
                for (int c = 0; c < 4; ++c) { const double d = (double)v[c]; s0[c] += d; q0[c] += d * d; }
              }
            } else if (r < e1) {
#pragma unroll
              for (int c = 0; c < 4; ++c) { const double d = (double)v[c]; s1[c] += d; q1[c] += d * d; }
            }
          }
          __syncthreads();                           // the image has been read: its memory takes the partials
          double* red = reinterpret_cast<double*>(smem);   // [seg*2 + stat][NG][BN]
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            red[(0 * NG + rg) * BN + cqd * 4 + c] = s0[c];
            red[(1 * NG + rg) * BN + cqd * 4 + c] = q0[c];
            red[(2 * NG + rg) * BN + cqd * 4 + c] = s1[c];
            red[(3 * NG + rg) * BN + cqd * 4 + c] = q1[c];
          }
          __syncthreads();
          const int Kp = a.tiles_n * BN;
          for (int idx = tid_e; idx < 4 * BN; idx += NT) {
            const int ss = idx / BN, col = idx - ss * BN;
            double t = 0.0;
#pragma unroll
            for (int gq = 0; gq < NG; ++gq) t += red[(ss * NG + gq) * BN + col];
            a.pool[((size_t)tile_m * 4 + ss) * Kp + kcol0 + col] = t;
          }
        } else {
          // band -> global: thread t moves chunks t, t + NT, ... (a wave-instruction covers 64 / CPR whole rows)
#pragma unroll
          for (int i = 0; i < PROWS * CPR / NT; ++i) {
            const int idx = i * NT + tid_e;
            const int r = idx / CPR, pp = idx % CPR;
            const int c = (pp & ~15) | ((pp ^ r) & 15);
            const int m = tile_m * BM + band0 + r;
            const int kfirst = OSPLIT ? kcol0 + (c >> 3) * 32 : kcol0 + c * 4;
            const u32x4 v = *reinterpret_cast<const u32x4*>(img + r * PITCH + pp * 16);
            if constexpr (BIGTAPS && !OSPLIT) {
              if (a.wg_R > 0) {   // (launch-uniform) a weight gradient straight into the reference layout [K, C, R, S]: four 4-byte stores
                if (m < a.M && kfirst < a.K) {
                  const int ci = dlip_div(m, a.div_howo), rem = m - ci * a.HoWo, rr = dlip_div(rem, a.div_wo), ss = rem - rr * a.Wo;
                  if (rr < a.wg_R && ss < a.wg_S) {
                    const int RS = a.wg_R * a.wg_S, crs = (a.M / a.HoWo) * RS;
                    float* dst = a.y + (size_t)kfirst * crs + ci * RS + rr * a.wg_S + ss;
#pragma unroll
                    for (int j = 0; j < 4; ++j) dst[(size_t)j * crs] = __uint_as_float(v[j]);
                  }
                }
                continue;
              }
            }
            const uint32_t off = (m < a.M && kfirst < a.K) ? (uint32_t)((m * a.ldy + kcol0) * 4 + c * 16) : DLIP_OOB_OFFSET;
            __builtin_amdgcn_raw_buffer_store_b128(v, yr, (int)off, 0, 0);
          }
        }
        if (ep + 1 < EPASS) __syncthreads();         // the next band reuses the image
      }
      if constexpr (OSPLIT) dlip_report_range(amax, a.status);
    }
    DLIP_LAB_SEGMENT_END_STAMPS();
    it += kn;
  }
  dlip_span_exit(sk.span);
  DLIP_LAB_WG_END_STAMPS();
}

// Second launch of a "publish only" split (StreamK::reduce_later): one thread per accumulator quad of a tile adds that quad over
// the tile's parts IN PART ORDER -- the same sums in the same order as the in-kernel finisher, so the same bits -- and runs the
// fp32 epilogue y = act(v / wscale + bias) * post_scale + post_shift.  Why: a weight gradient run as a convolution is 1 - 36 tiles
// cut 14 - 100 ways (5 tiles x 14 036 slices on layer 1); the finisher -- ONE workgroup per tile reading its parts' slabs one
// after the other -- then ran 230 of a launch's 370 us alone on the chip (in-kernel stamps, round 4: workgroups busy 136 us in the
// median, 369 us the slowest).  Here the parts' loads are independent and every quad has its own thread: microseconds.
template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void slab_reduce_kernel(const ConvArgs a, const float* __restrict__ slabs, int parts) {
  constexpr int NT = 64 * WAVES_M * WAVES_N, WM = BM / WAVES_M, WN = BN / WAVES_N, FR = 16;
  constexpr int MI = WM / FR, NI = WN / FR, QUADS = MI * NI * NT;   // = BM * BN / 4
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= QUADS) return;
  const int tile = blockIdx.y;
  const f32x4* src = reinterpret_cast<const f32x4*>(slabs) + (size_t)2 * tile * parts * QUADS + q;   // part p of tile t: slab 2 (t parts + p)
  f32x4 t = src[0];
#pragma unroll 8
  for (int p = 1; p < parts; ++p) {
    const f32x4 v = src[(size_t)2 * p * QUADS];
#pragma unroll
    for (int c = 0; c < 4; ++c) t[c] = t[c] + v[c];
  }
  // the quad's place in the tile: the main kernel's accumulator map (acc[mi][ni] of thread tid: pixel row wm WM + 16 mi + lrow,
  // channels wn WN + 16 ni + 4 half ..+3)
  const int tid = q % NT, blk = q / NT, ni = blk % NI, mi = blk / NI;
  const int wave = tid >> 6, wm = wave / WAVES_N, wn = wave % WAVES_N, lrow = tid & (FR - 1), half = (tid & 63) / FR;
  const int tiles_m = (a.M + BM - 1) / BM;
  int tile_n = tile / tiles_m, tile_m = tile - tile_n * tiles_m;
  if (a.n_inner) { tile_m = tile / a.tiles_n; tile_n = tile - tile_m * a.tiles_n; }
  const int m = tile_m * BM + wm * WM + mi * FR + lrow;
  const int k = tile_n * BN + wn * WN + ni * FR + 4 * half;
  if (m >= a.M || k >= a.K) return;                      // K % 4 == 0: a quad is whole or absent
  f32x4 y;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float v = t[c] * (1.f / a.wscale[k + c]) + (a.bias ? a.bias[k + c] : 0.f);   // power of two: exact
    v = v >= 0.f ? v : v * (a.slope ? a.slope[k + c] : 1.f);
    if (a.pscale) v = v * a.pscale[k + c] + a.pshift[k + c];
    y[c] = v;
  }
  if (a.wg_R > 0) {   // a weight gradient straight into the reference layout [K, C, R, S] (kernel comment at the band store)
    const int ci = m / a.HoWo, rem = m - ci * a.HoWo, rr = rem / a.Wo, ss = rem - rr * a.Wo;
    if (rr >= a.wg_R || ss >= a.wg_S) return;
    const int RS = a.wg_R * a.wg_S, crs = (a.M / a.HoWo) * RS;
    float* dst = a.y + (size_t)k * crs + ci * RS + rr * a.wg_S + ss;
#pragma unroll
    for (int c = 0; c < 4; ++c) dst[(size_t)c * crs] = y[c];
    return;
  }
  *reinterpret_cast<f32x4*>(a.y + (size_t)m * a.ldy + k) = y;
}

#include "conv_dma_lab_menu.inc"   // lab build: the stamped launch path (nothing in the product build)

// Per-stream workspace of the balanced split: ticket counters (zeroed once; every launch leaves them
// zero) + slabs.  Launches on one stream are ordered, so they can share it; another stream needs its own.
// The caller normally owns it (dlip_conv_workspace_bytes / dlip_conv_set_workspace: the Python
// binding registers a torch allocation per stream); a stream nobody registered a block for gets a
// library-owned one on first use, and a registered block that is too small for a launch simply makes
// that launch a plain (unbalanced) one.
struct Workspace {
  float* slabs = nullptr;
  int* counters = nullptr;
  size_t slab_floats = 0;
  bool external = false;
};
constexpr int kMaxSplitTiles = 1 << 16;   // counter words per workspace
constexpr double kSlotFlops = 0.85e12;    // algorithmic FLOP/s one resident 128x128 workgroup sustains (measured, 2 per CU)
constexpr int kMaxPartsPerTile = 64;      // balanced split: upper bound on the workgroups sharing one tile (16 before the finisher pipelined its slab reads)
constexpr double kHandoffUs = 10.0;       // cost of the slab hand-off of a launch at 128x128 tiles (measured)
constexpr int kMaxDevices = 16;

std::mutex& ws_mutex() { static std::mutex m; return m; }
std::map<std::pair<int, hipStream_t>, Workspace>& ws_table() {
  static std::map<std::pair<int, hipStream_t>, Workspace> t;
  return t;
}

Workspace* workspace_for(int dev, hipStream_t st, size_t slab_floats) {
  std::lock_guard<std::mutex> lock(ws_mutex());
  Workspace& w = ws_table()[{dev, st}];
  if (w.external) return w.slab_floats >= slab_floats ? &w : nullptr;
  // library-owned block: never created while the stream is being captured into a step plan (allocation and
  // synchronisation are not legal there) -- such a launch runs unbalanced; warm the stream up first.
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap) != hipSuccess) return nullptr;
  const bool capturing = cap != hipStreamCaptureStatusNone;
  if (w.counters == nullptr) {
    if (capturing) return nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&w.counters), kMaxSplitTiles * sizeof(int)) != hipSuccess) return nullptr;
    if (hipMemset(w.counters, 0, kMaxSplitTiles * sizeof(int)) != hipSuccess) return nullptr;
  }
  if (w.slab_floats < slab_floats) {
    if (capturing) return nullptr;
    if (w.slabs) {   // earlier launches on this stream may still read the old block
      if (hipStreamSynchronize(st) != hipSuccess) return nullptr;
      (void)hipFree(w.slabs);
      w.slabs = nullptr;
      w.slab_floats = 0;
    }
    if (hipMalloc(reinterpret_cast<void**>(&w.slabs), slab_floats * sizeof(float)) != hipSuccess) return nullptr;
    w.slab_floats = slab_floats;
  }
  return &w;
}

// Per kernel instance and device, set once under the mutex: the LDS attribute is applied and the number of
// workgroups the chip holds at once is queried (neither is free, and neither belongs on the launch path).
struct KernelSlots {
  int slots[kMaxDevices] = {};
};

template <typename K>
int kernel_slots(KernelSlots& ks, int dev, K kern, int threads, size_t lds) {
  if (dev < 0 || dev >= kMaxDevices) return 0;
  std::lock_guard<std::mutex> lock(ws_mutex());
  if (ks.slots[dev] == 0) {
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return 0;
    int cus = 0, per_cu = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, threads, lds) != hipSuccess) return 0;
    ks.slots[dev] = cus * per_cu;
  }
  return ks.slots[dev];
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int NSTAGE, int OCC, int EPI, bool DUAL, int VAR = 0>
int launch_one(const ConvArgs& a, hipStream_t st) {
  ConvArgs b = a;
  const int tiles_m = (a.M + BM - 1) / BM;
  b.tiles_n = (a.K + BN - 1) / BN;
  const long long tiles = (long long)tiles_m * b.tiles_n;
  if (tiles <= 0 || tiles > 0x7FFFFFFFll) return DLIP_EINVAL;
  if (EPI == 2 && a.pool_group < BM) return DLIP_EINVAL;   // a tile may contain at most one row-group boundary
  constexpr bool WIN = (VAR & 128) != 0;   // window mode: two window slots + NSTAGE weight stages (kernel comment)
  constexpr size_t win_ring0 = (size_t)2 * (BM + 48) * ROWB + (size_t)NSTAGE * BN * ROWB;
  constexpr size_t win_ring = win_ring0 > (size_t)BM * BN * 4 ? win_ring0 : (size_t)BM * BN * 4;
  constexpr size_t lds = (WIN ? win_ring : (size_t)NSTAGE * (BM + BN) * ROWB) + 5 * BN * sizeof(float);   // ring + epilogue parameter table
  constexpr int threads = 64 * WAVES_M * WAVES_N;
  static_assert(lds <= 160 * 1024, "LDS ring exceeds a CU");
  auto kern = conv_igemm_f16x3_dma_kernel<BM, BN, WAVES_M, WAVES_N, EPI, NSTAGE, OCC, DUAL, VAR>;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return DLIP_EINVAL;
  static KernelSlots ks;   // one per instantiation
  const int sl = kernel_slots(ks, dev, kern, threads, lds);
  if (sl <= 0) return DLIP_EINVAL;

  StreamK sk;
  sk.iters = tiles * a.nk;
  sk.slabs = nullptr;
  sk.counters = nullptr;
  long long G = tiles;                                  // plain launch: one tile per workgroup
  const int balanced = dlip_dbg_value[DLIP_DBG_STREAMK] < 0 ? 1 : dlip_dbg_value[DLIP_DBG_STREAMK];   // 0 never, 2 always
  if (balanced && tiles <= kMaxSplitTiles) {
    // Plain: ceil(tiles / slots) rounds of one tile-time.  Balanced: tiles / slots tile-times plus the
    // slab hand-off (one slab written and read per workgroup, all the reads at the very end).
    const double tile_us = 2.0 * BM * BN * 32.0 * a.nk / (kSlotFlops * (WAVES_M * WAVES_N / 4.0) * 1e-6);   // an 8-wave workgroup owns its CU
    const double plain_us = (double)((tiles + sl - 1) / sl) * tile_us;
    long long Gb = sl;
    // small problems: at least 16 slices per workgroup (every extra part costs the finisher a serial slab read)
    if (sk.iters < 16 * Gb) Gb = sk.iters / 16 > 0 ? sk.iters / 16 : 1;
    // few tiles with very long reductions (the weight-gradient GEMMs of training: 1..8 tiles, 10^4 slices):
    // the finisher adds the parts of a tile one after the other (~2 us each), so a tile is cut into at most
    // kMaxPartsPerTile parts -- 512 parts of one 64x64 tile cost 1 ms of serial slab reads for 90 us of MFMA
    // (reductions of thousands of slices -- the weight gradients run as convolutions: 5 tiles x 14 036 slices on layer 1 -- afford
    // more parts: 64 parts of 5 tiles leave 192 of 512 slots idle)
    const long long max_parts = a.nk >= 16384 ? 4 * kMaxPartsPerTile : a.nk >= 4096 ? 2 * kMaxPartsPerTile : kMaxPartsPerTile;
    // (no such bound where the parts go to the parallel reduce launch below: a 1 x 1 shortcut's weight gradient -- ONE 64 x 128
    // tile of 3 509 slices -- used 64 of the chip's 512 slots)
    const bool reduce_ok = EPI == 0 && !DUAL && a.res == nullptr && a.nk >= 256 && tiles <= 64 && dlip_dbg_value[DLIP_DBG_STREAMK] != 4;
    if (Gb > tiles * max_parts && !reduce_ok) Gb = tiles * max_parts;   // (same-box A/B of this rule alone: no change on either training step)
    const double bal_us = (double)sk.iters / Gb / a.nk * tile_us + kHandoffUs * (BM * BN / 16384.0);
    // Short reductions on a full chip (nk <= 16 slices, at least one tile per slot: the k = 1 TDNN layers) run
    // plain: a tile's slab hand-off costs as much as a third of such a tile, and the under-filled last round of a
    // plain launch runs faster than the model's whole tile-time (measured: 53.9 vs 57.6 us on tdnn.k1, 113.9 vs
    // 115.3 on tdnn.9; the deep 3x3 / dilated layers gain 12 % from the split and keep it).
    const bool short_full = a.nk <= 16 && tiles >= sl;
    if ((balanced == 2 || (bal_us < plain_us && !short_full)) && Gb * a.nk != sk.iters) G = Gb;
  }
  if (G != tiles) {
    Workspace* w = workspace_for(dev, st, (size_t)2 * G * BM * BN);
    if (w == nullptr) {
      G = tiles;   // no (or too small a) workspace: plain launch
    } else {
      sk.slabs = w->slabs;
      sk.counters = w->counters;
    }
  }
  sk.il_tiles = 0;
  sk.l2_local = dlip_dbg_value[DLIP_DBG_STREAMK] == 3 ? 1 : 0;
  if (G != tiles && a.nk >= 256 && tiles >= 1 && tiles <= 64 && G / tiles >= 2) {
    G = G / tiles * tiles;          // whole parts: every workgroup stays inside one tile
    sk.il_tiles = (int)tiles;
  }
  // Few tiles cut many ways, plain fp32 output: the parts are only published and a second, fully parallel launch adds them and runs
  // the epilogue (slab_reduce_kernel) -- from 4 parts per tile on (below that the in-kernel finisher's one or two slab reads cost
  // less than a launch); dlip_debug_set(3, 4) keeps the in-kernel finisher (A/B runs, tests)
  sk.reduce_later = 0;
  if constexpr (EPI == 0 && !DUAL) {
    if (sk.il_tiles > 0 && a.res == nullptr && G / tiles >= 4 && dlip_dbg_value[DLIP_DBG_STREAMK] != 4) sk.reduce_later = 1;
  }
  sk.G = (int)G;
  sk.span = dlip_span_next();
  // Tile order.  Column block INNER when the whole filter bank stays in an XCD's 4 MiB L2 beside the activation rows in
  // flight (<= 3.25 MB: the TDNN layers incl. tdnn.9's 12 column blocks, layer 3): the column blocks of one row block run
  // back to back, so its activation rows come from HBM once instead of once per column block (tdnn.9 108 -> 98 us,
  // tdnn k=1 50 -> 48, layer 3 -1..5 %).  Larger banks (layer 4: 9.4 MB) keep the column block OUTER -- there each XCD
  // stays on one 128-channel weight block and re-streams the (L2-sized) activations instead (+4 % if forced inner).
  b.n_inner = (b.tiles_n > 1 && (size_t)a.K * a.rsc * 4 <= (size_t)3407872) ? 1 : 0;
  // ... and PAIRED (kernel comment at `P`) where the launch has exactly two column blocks of the 256-row tile and enough row tiles
  // to split -- layer 3 (K = 256): the two blocks of a row range run side by side on one XCD instead of back to back on one CU
  // (the dominant kernel's FETCH_SIZE per launch 372 -> see profiles/r5/pmc_summary.json; dlip_debug_set(5, 2 | 1 | 0) forces a mode)
  if (b.n_inner == 1 && b.tiles_n == 2 && BM == 256 && sk.il_tiles == 0 && !sk.reduce_later && (G & 1) == 0 && tiles > 64) b.n_inner = 2;
  if (dlip_dbg_value[DLIP_DBG_NINNER] >= 0) {
    const int v = dlip_dbg_value[DLIP_DBG_NINNER];
    if (v == 2) b.n_inner = (b.tiles_n > 1 && sk.il_tiles == 0 && !sk.reduce_later && G % b.tiles_n == 0) ? 2 : (b.tiles_n > 1 ? 1 : 0);
    else b.n_inner = v > 0 ? 1 : 0;
  }
  DLIP_LAB_LAUNCH_HOOK();   // (lab build: DLIP_STAMP_PRINT -> stamped launch + printed medians)
  hipLaunchKernelGGL(kern, dim3((unsigned)G), dim3(threads), lds, st, b, sk);
  if (sk.reduce_later) {
    constexpr int quads = BM * BN / 4;
    hipLaunchKernelGGL((slab_reduce_kernel<BM, BN, WAVES_M, WAVES_N>), dim3((quads + 255) / 256, (unsigned)tiles), dim3(256), 0, st, b,
                       static_cast<const float*>(sk.slabs), (int)(G / tiles));
  }
  return dlip_launch_status();
}

// Instances: every tile x {fp32, split} output, plain and with the second reduction source; the pooled
// epilogue on the two tiles its callers reach (128x128: tdnn.9 and small clips; 256x128: the trunk's last conv).
template <int BM, int BN, int WAVES_M, int WAVES_N, int NSTAGE, int OCC, int VAR = 0>
int launch_dma(const ConvArgs& a, hipStream_t st, int epi) {
  const bool dual = a.x2 != nullptr;
  if constexpr (VAR == 128) {   // window mode (product): fp32 / split output
    if (epi == 2 || dual) return DLIP_EINVAL;
    return epi ? launch_one<BM, BN, WAVES_M, WAVES_N, NSTAGE, OCC, 1, false, VAR>(a, st)
               : launch_one<BM, BN, WAVES_M, WAVES_N, NSTAGE, OCC, 0, false, VAR>(a, st);
  } else if constexpr (VAR == 4096) {   // more than 32 filter taps (the weight gradient as a convolution): fp32 output, one source
    if (epi != 0 || dual) return DLIP_EINVAL;
    return launch_one<BM, BN, WAVES_M, WAVES_N, NSTAGE, OCC, 0, false, VAR>(a, st);
  } else if constexpr (VAR != 0) {   // lab experiments: plain launches only
    if (epi == 2 || dual) return DLIP_EINVAL;
    return epi ? launch_one<BM, BN, WAVES_M, WAVES_N, NSTAGE, OCC, 1, false, VAR>(a, st)
               : launch_one<BM, BN, WAVES_M, WAVES_N, NSTAGE, OCC, 0, false, VAR>(a, st);
  }
  if (epi == 2) {
    // (the product tiles 0 and 5: the whole fp32 tile image must fit the ring in one band)
    if constexpr (BN == 128 && ((BM == 128 && WAVES_M == 2) || (BM == 256 && NSTAGE == 3))) {
      if (!dual) return launch_one<BM, BN, WAVES_M, WAVES_N, NSTAGE, OCC, 2, false>(a, st);
    }
    return DLIP_EINVAL;
  }
  if (dual) return epi ? launch_one<BM, BN, WAVES_M, WAVES_N, NSTAGE, OCC, 1, true>(a, st)
                       : launch_one<BM, BN, WAVES_M, WAVES_N, NSTAGE, OCC, 0, true>(a, st);
  return epi ? launch_one<BM, BN, WAVES_M, WAVES_N, NSTAGE, OCC, 1, false>(a, st)
             : launch_one<BM, BN, WAVES_M, WAVES_N, NSTAGE, OCC, 0, false>(a, st);
}

}  // namespace

// Tile menu of the DMA kernel (index = what dlip_conv_plan reports via dlip_conv_dma_tile; dlip_debug_set
// (DLIP_DBG_DMA_TILE) forces one for tests and A/B runs).  0..5 are the product instances; a lab build
// (python -m deeplip_amd.build --lab: -DDLIP_LAB, libdeeplip_hip_lab.so) appends its experiments (conv_dma_hooks.h).
const TileCfg kDmaCfg[] = {{128, 128}, {128, 64}, {64, 128}, {64, 64}, {128, 64}, {256, 128} DLIP_LAB_TILE_CFGS};
constexpr int NUM_DMA_ALL = (int)(sizeof(kDmaCfg) / sizeof(kDmaCfg[0]));

// Tile choice (measured per layer with tools/bench_dma.py, MI355X, balanced split on).  The cost of a slice
// is set by the bytes it pulls from L2 (ablations: pieces issued out of range cost nothing, pieces that fetch
// cost 30 % of a layer), so the deepest layers take the tile with the fewest bytes per FLOP: 256x128, eight
// waves, one workgroup per CU, three-stage ring (5-11 % less time than two co-resident 128x128 workgroups
// on the 3x3 layers of layer2..4).  128x128 (two per CU) serves mid-depth reductions and the M < 8192
// launches; narrow outputs (K <= 64) do better on 128x64 with a three-stage ring, very short reductions
// (nk <= 8 slices: the 1x1 down-sampling convolutions) on 128x64 with a two-stage ring and three
// workgroups per CU (latency, not MFMA, bounds them); M <= 64 (fully connected layers on a batch) uses
// the 64-row tiles.  A pooled epilogue exists on 128x128 and 256x128 only.
static int dma_pick(long long M, int K, int nk, int epi) {
  if (const int v = dlip_dbg_value[DLIP_DBG_DMA_TILE]; v >= 0 && v < NUM_DMA_ALL && epi != 2) return v;
  if (epi == 2) return (nk >= 32 && M >= 8192) ? 5 : 0;
  if (M <= 64) return K <= 64 ? 3 : 2;
  // (round 4) up to 2 048 rows with a reduction of 16 slices or more -- the MS-TCN head's convolutions of a training step: 928 + padding
  // rows, 512 | 768 -> 256 channels, 48-168 slices -- are a handful of row tiles split many ways: on 64-row tiles there are twice as
  // many tiles to split and half as many parts per tile for the finisher to add (tools/bench_tcn.py, same box: 41.8 -> 27.9 us,
  // 45.9 -> 32.7, 60.7 -> 45.6 against 128 x 128; 128 x 64 and 256 x 128 in between)
  // (round 5) ... and up to 3 072 rows: the head's k = 7, dilation 8 branch is 32 x (29 + 48) = 2 464 rows x 168 slices and had fallen
  // to 128 x 128: 64.7 us against 44.6 on 64 x 128 (tools/bench_tcn.py, same box) -- the longest branch of its stage, four launches a step
  if (M <= 3072 && nk >= 16 && nk < 256) return K <= 64 ? 3 : 2;
  if (K <= 64) return 1;
  if (nk <= 8) return 4;
  if (nk >= 32 && M >= 8192) return 5;
  // (round 4) 18..31 slices over many rows -- layer 2's stride-2 3x3 convolution, 64 -> 128 channels on 22x22 maps: M = 224 576,
  // 18 slices -- took 152 us on 128x128 and takes 137 on 256x128 with the balanced split (same box, tools/bench_dma.py); the
  // k = 1 TDNN layers, where 256x128 had been no better at 16 slices, are on the rows kernel now
  if (nk >= 18 && M >= 65536) return 5;
  // very long reductions over few tiles (the weight gradients run as convolutions: 5-36 tiles of 260-3 500 slices; no forward launch reduces over more than 144): the balanced split
  // fills the chip whatever M is, and the 256x128 tile's loop is the faster one (7-12 % per launch, tools/bench_wgrad.py)
  // (round 4, with the reduce launch in place: from M = 256 -- the speech encoder's 512 x 512 weight gradients, 2 400 slices, had
  // fallen to the 64-row rule above: 13.23 ms per training step on 64 x 128, 12.72 on 128 x 128, 12.60 on 256 x 128, same box)
  if (nk >= 256 && M >= 256) return 5;
  // (Measured and rejected in round 2: 160-row tiles for the short plain launches -- the k = 1 TDNN layers are 592 tiles of
  // 128x128 on 512 slots, two rounds with the second 16 % full, and 476 tiles of 160x128, one round: 47.8 vs 46.8 us.  The
  // under-filled second round is cheap because these layers wait for operands, not for the matrix core.)
  return 0;
}

// The split workspace of `stream` for another kernel of the library (conv_rows_f16x3.hip's general mode): 1 and the block, or 0
// when there is none of that size (same rules as for this kernel's own launches: never allocated inside a stream capture).
extern "C" __attribute__((visibility("hidden"))) int dlip_conv_split_workspace(void* stream, size_t slab_floats, float** slabs, int** counters,
                                                                               int* counter_words) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  Workspace* w = workspace_for(dev, static_cast<hipStream_t>(stream), slab_floats);
  if (w == nullptr) return 0;
  *slabs = w->slabs;
  *counters = w->counters;
  *counter_words = kMaxSplitTiles;
  return 1;
}

extern "C" int dlip_conv_rows2d_ok(const void* args, int epi);                                // conv_rows_f16x3.hip
extern "C" int dlip_conv_f16x3_rows2d_launch(const void* args, void* stream, int epi);
extern "C" int dlip_conv_rows_declined(void);

// Library-internal entry points (hidden): ConvArgs lives in an unnamed namespace, so it crosses the
// translation-unit boundary as an opaque pointer.
extern "C" __attribute__((visibility("hidden"))) void dlip_conv_dma_tile(long long M, int K, int nk, int epi, int* bm, int* bn) {
  const TileCfg& c = kDmaCfg[dma_pick(M, K, nk, epi)];
  *bm = c.bm;
  *bn = c.bn;
}

// Called by dlip_conv_nhwc_f16x3 / dlip_conv2_nhwc_f16x3 / dlip_conv_pool_f16x3 for DLIP_SPLIT_IN launches
// (argument checks done there).  epi: 0 fp32 y, 1 split y, 2 pooled partials (a.pool, a.pool_group).
extern "C" __attribute__((visibility("hidden"))) int dlip_conv_f16x3_dma_launch(const void* args, void* stream, int epi) {
  const ConvArgs& a = *static_cast<const ConvArgs*>(args);
  hipStream_t st = static_cast<hipStream_t>(stream);
  // (round 4) the rows kernel's general mode (160 x 256 tiles, continuous slice stream, epilogue from registers, the same balanced
  // split), when FORCED (dlip_debug_set(7, 1); measured slower than this kernel on the trunk: conv_rows_f16x3.hip); it declines when
  // the stream has no split workspace for it
  if (dlip_conv_rows2d_ok(args, epi)) {
    const int rc = dlip_conv_f16x3_rows2d_launch(args, stream, epi);
    if (rc != dlip_conv_rows_declined()) return rc;
  }
  // The many-tap instances are also the ONLY ones that honour the slice / tap strides of slice-major operand images
  // (dlip_wgrad_conv_f16x3: cs_x, wt, cs_w, Hs); a weight gradient with 32 taps or fewer -- layer 4's 3x3 output-gradient maps: 9 taps --
  // used to fall through to the pixel-major instances below and read its slice-major images as if they were pixel-major: wrong
  // layer-4 weight gradients whenever the batch had more than 32 images (round 3's default layout; the 2-clip golden has one
  // slice, where the two layouts coincide; found by tests/test_train_video_gpu.py's full-size gradient test in round 4).
  const bool slice_major = a.cs_x != 128 || a.cs_w != 128 || a.wt != a.Cw * 4 || a.Hs != a.H;
  if (a.R * a.S > 32 || slice_major) {   // the tap-mask-free variant, on the three tiles long reductions use
    const int t = dma_pick(a.M, a.K, a.nk, epi);
    if (a.K <= 64 || t == 1 || t == 3) return launch_dma<128, 64, 2, 2, 3, 2, 4096>(a, st, epi);
    if (t == 5) return launch_dma<256, 128, 4, 2, 3, 1, 4096>(a, st, epi);
    return launch_dma<128, 128, 2, 2, 2, 2, 4096>(a, st, epi);
  }
  switch (dma_pick(a.M, a.K, a.nk, epi)) {
    case 0: return launch_dma<128, 128, 2, 2, 2, 2>(a, st, epi);
    case 1: return launch_dma<128, 64, 2, 2, 3, 2>(a, st, epi);
    case 2: return launch_dma<64, 128, 2, 2, 3, 2>(a, st, epi);
    case 3: return launch_dma<64, 64, 2, 2, 3, 2>(a, st, epi);
    case 4: return launch_dma<128, 64, 2, 2, 2, 3>(a, st, epi);
    DLIP_LAB_DISPATCH_CASES   // (lab build: tiles 6.. of its menu; nothing in the product build)
    default:
      return launch_dma<256, 128, 4, 2, 3, 1, 0>(a, st, epi);
  }
}

// ---- caller-owned workspace (include/deeplip_hip.h) ----
extern "C" int64_t dlip_conv_workspace_bytes(void) {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
    return -1;
  // counters + two slabs per resident workgroup: one per CU x 160x256 fp32 (the rows kernel's general mode) is the largest product on
  // the menu (this kernel's own: 2 per CU x 128x128)
  return (int64_t)kMaxSplitTiles * 4 + (int64_t)2 * cus * 160 * 256 * 4;
}

extern "C" int dlip_conv_set_workspace(void* ptr, int64_t bytes, dlip_stream_t stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return DLIP_EINVAL;
  std::lock_guard<std::mutex> lock(ws_mutex());
  Workspace& w = ws_table()[{dev, st}];
  if (ptr == nullptr) {   // unregister: the stream falls back to a library-owned block
    if (w.external) w = Workspace();
    return DLIP_OK;
  }
  DLIP_CHECK_ARG((reinterpret_cast<uintptr_t>(ptr) & 15) == 0 && bytes >= (int64_t)kMaxSplitTiles * 4 + 16);
  if (!w.external && (w.slabs || w.counters)) {   // drop the library-owned block this stream used so far
    if (hipStreamSynchronize(st) != hipSuccess) return DLIP_EINVAL;
    if (w.slabs) (void)hipFree(w.slabs);
    if (w.counters) (void)hipFree(w.counters);
  }
  if (hipMemsetAsync(ptr, 0, (size_t)kMaxSplitTiles * 4, st) != hipSuccess) return DLIP_EINVAL;   // tickets start at zero
  w.counters = static_cast<int*>(ptr);
  w.slabs = reinterpret_cast<float*>(static_cast<char*>(ptr) + (size_t)kMaxSplitTiles * 4);
  w.slab_floats = ((size_t)bytes - (size_t)kMaxSplitTiles * 4) / 4;
  w.external = true;
  return DLIP_OK;
}

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
#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>
#include <mutex>
#include <utility>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int ROWB = 128;  // bytes per LDS row (one 32-channel slice of one pixel / filter)

// Window mode: rows a window holds beyond the tile's BM pixels (>= (R-1) dil W + (S-1) dil of the launch).
#define DLIP_WIN_SLACK 48
// Rows per chunk plane of a window: BM + slack + 16 spare rows (always loaded out of range, i.e. zero:
// the rows masked taps are redirected to), rounded up to whole 64-row DMA pieces.
constexpr int dlip_win_rows(int BM) { return (BM + DLIP_WIN_SLACK + 16 + 63) / 64 * 64; }
// Bytes of [weight ring][two window slots].
constexpr int dlip_win_ring_bytes(int BM, int BN, int NSTAGE) { return NSTAGE * BN * ROWB + 2 * dlip_win_rows(BM) * ROWB; }

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
// POL: cache policy of the load -- 0 default, 1 `sc1` (served by L2, does not allocate in this CU's L1), 2 `nt`.
template <int POL = 0>
__device__ __forceinline__ void dma_piece(const u32x4 rsrc, uint32_t voff, uint32_t lds_base) {
  const uint32_t base = __builtin_amdgcn_readfirstlane(lds_base);   // wave-uniform by construction; this pins it to an SGPR
  // m0 is reserved: hipcc keeps nothing in it across statements (the ISA dump shows no other m0 use)
  if constexpr (POL == 1)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen sc1 lds" : : "s"(base), "v"(voff), "s"(rsrc) : "memory");
  else if constexpr (POL == 2)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen nt lds" : : "s"(base), "v"(voff), "s"(rsrc) : "memory");
  else
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(base), "v"(voff), "s"(rsrc) : "memory");
}
#ifndef DLIP_POL_A
#define DLIP_POL_A 0
#endif
#ifndef DLIP_POL_B
#define DLIP_POL_B 0
#endif

// Balanced ("stream-K") work split: the launch's reduction work = tiles x nk slices is cut into G equal
// contiguous ranges, one per workgroup, G = the number of workgroups the chip holds at once.  A range
// covers whole tiles plus at most one partial tile at each end; a workgroup that computed only some
// slices of a tile stores its fp32 accumulators as a slab and takes a ticket on the tile's counter;
// the workgroup drawing the last ticket adds the other slabs (in part order) and runs the epilogue.
// Nobody waits for anybody, so residency and dispatch order cannot deadlock it; visibility follows the
// agent-scope release / acquire counter recipe (slab stores -> vmcnt(0) -> barrier -> release fence ->
// ticket; last ticket -> acquire fence -> barrier -> plain slab loads).  With G = tiles every range
// is exactly one tile and no slab is ever written (the plain data-parallel launch).
struct StreamK {
  long long iters;   // tiles * nk
  float* slabs;      // [2 * G][BM * BN] fp32
  int* counters;     // [tiles], zero between launches (the last arriver resets its tile's word)
  int G;             // workgroups in the grid
  int whole;         // 1: ranges are cut at tile boundaries (persistent workgroups, no tile is shared)
#ifdef DLIP_STAMPS
  unsigned long long* stamps;   // diagnostic build only: [G][10] s_memtime values of each workgroup's first segment
#endif
};

#ifdef DLIP_STAMPS
#define DLIP_STAMP(i) do { if (threadIdx.x == 0 && it == it_begin) sk.stamps[(size_t)g * 10 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define DLIP_STAMP(i) do { } while (0)
#endif

// M16 selects the matrix instruction: v_mfma_f32_16x16x32_f16 (one instruction per 32-channel slice and
// 16x16 block; the shape that holds the higher clock in a dense loop on gfx950) instead of two k16 steps of
// v_mfma_f32_32x32x16_f16.  LDS image, ring, work split and epilogue are shared; only the lane -> (row, k group)
// mapping of the fragment reads and of the accumulator registers differs.
//
// WIN ("window" mode, same-size stride-1 convolutions: every 3x3 layer of the trunk): what a slice costs is
// the bytes it pulls from L2 (ablations in DESIGN.md), and the R x S taps of one 32-channel slice read the
// same activation rows shifted by whole pixels.  So the activation operand is not re-fetched per tap: per
// channel slice the workgroup loads ONE window -- the 128-B rows of the BM + (R-1) dil W + (S-1) dil
// consecutive input pixels its tile touches -- into one of two LDS slots (the next slice's window lands
// while the taps of this one are multiplied), and a tap reads its fragments from the window at a row
// offset; rows that fall outside the image (or past M) are redirected per lane to a row that is always
// zero, by the same per-pixel tap mask the per-tap gather uses.  The LDS ring then carries the weights
// only: L2 -> LDS bytes per slice drop from (BM + BN) x 128 to BN x 128 + ~(BM + 64) x 128 / (R S).
// A window is stored as 8 PLANES, one per 16-B chunk of the 128-B pixel row (plane c, row r at
// (c WRP + r) x 16 B, WRP a multiple of 16): a fragment read takes 16 consecutive rows of one plane,
// i.e. 256 consecutive bytes, so it is bank-conflict free at EVERY row offset -- no XOR swizzle keyed on
// the row can be (the key pattern of a 16-row group is not shift invariant; measured 32 % conflict
// cycles with the row-major image).  An LDS-DMA piece then gathers the same chunk of 64 pixels.
template <int BM, int BN, int WAVES_M, int WAVES_N, bool OSPLIT, int NSTAGE, int OCC, bool M16, bool WIN = false>
__global__ __launch_bounds__(64 * WAVES_M* WAVES_N, OCC) void conv_igemm_f16x3_dma_kernel(const ConvArgs a, const StreamK sk) {
  constexpr int NW = WAVES_M * WAVES_N, NT = 64 * NW;
  constexpr int RPP = NT / 8;   // rows one pass of the workgroup covers (8 lanes x 16 B per row)
  static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must be whole passes");
  static_assert(NSTAGE == 2 || NSTAGE == 3, "ring depth");
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int FR = M16 ? 16 : 32;    // rows of one MFMA fragment
  constexpr int QN = M16 ? 1 : 4;      // 4-register quads per accumulator block
  constexpr int MI = WM / FR, NI = WN / FR;
  using acc_t = typename std::conditional<M16, f32x4, f32x16>::type;
  constexpr int A_PER = BM / RPP, B_PER = BN / RPP;
  constexpr int NL = A_PER + B_PER;   // DMA instructions per wave per slice
  constexpr int STAGE_B = WIN ? BN * ROWB : (BM + BN) * ROWB;
  constexpr int LDK = 32;             // dwords per LDS row
  constexpr int PF = NSTAGE - 1;      // slices in flight ahead of the one being multiplied
  // window mode: [weight ring][window slot 0][window slot 1]; a slot = 8 chunk planes x WRP rows x 16 B
  constexpr int WRP = dlip_win_rows(BM);         // rows per plane (a multiple of 64: whole DMA pieces)
  constexpr int WPIECES = 8 * WRP / 64;          // 1-KiB DMA pieces per window (64 rows of one chunk plane each)
  constexpr int WPER = WPIECES / NW;             // per wave
  constexpr int WIN_B = WRP * ROWB;
  constexpr int WIN0 = NSTAGE * STAGE_B;
  constexpr int RING = WIN ? dlip_win_ring_bytes(BM, BN, NSTAGE)
                           : NSTAGE * STAGE_B;   // bytes; the epilogue parameter table (5 x BN floats) sits behind it
  static_assert(!WIN || (M16 && 8 % NW == 0 && WRP >= BM + DLIP_WIN_SLACK + 16 && B_PER + WPER < 64), "window mode layout");
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int g = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);   // neighbours in work order share an XCD (L2)

  const int tid = threadIdx.x;
  const int cq = tid & 7;
  const int rbase = tid >> 3;
  const int key_st = (rbase >> 1) & 7;          // RPP is a multiple of 16: the key is the same in every pass
  const int csrc = ((cq ^ key_st) << 2);        // first channel (dword) of the chunk this lane fetches
  const u32x4 xr = make_rsrc_words(a.x, a.x_bytes);
  const u32x4 wr = make_rsrc_words(a.w, a.w_bytes);
  const __amdgpu_buffer_rsrc_t yr = dlip_make_rsrc(a.y, a.y_bytes);

  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const uint32_t piece0 = lds0 + wave * 8 * ROWB;   // this wave's 8 rows of pass 0, stage 0, operand A
  const int lane = tid & 63;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  // 32x32x16: lane = (row 0..31, k half); a k16 step s reads hi chunk 2s + half, lo chunk 4 + 2s + half.
  // 16x16x32: lane = (row 0..15, k group 0..3); the one step reads hi chunk kgroup, lo chunk 4 + kgroup.
  const int lrow = lane & (FR - 1), half = lane / FR;
  const int a_frag = (wm * WM + lrow) * LDK;
  const int b_frag = (WIN ? 0 : BM * LDK) + (wn * WN + lrow) * LDK;
  const int key_rd = (lrow >> 1) & 7;   // fragment blocks start at multiples of 16 rows: the key depends on lrow only
  int khi[2], klo[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    khi[s] = (((M16 ? 0 : 2 * s) + half) ^ key_rd) << 2;
    klo[s] = ((4 + (M16 ? 0 : 2 * s) + half) ^ key_rd) << 2;
  }
  const int x_dr = a.dh * a.W * a.ldx * 4, x_ds = a.dw * a.ldx * 4;
  const int ntaps = a.R * a.S;

  long long it_begin = (long long)g * sk.iters / sk.G, it_end = (long long)(g + 1) * sk.iters / sk.G;
  if (sk.whole) {   // whole tiles per workgroup: tiles g T / G .. (g + 1) T / G
    const long long T = sk.iters / a.nk;
    it_begin = ((long long)g * T / sk.G) * a.nk;
    it_end = ((long long)(g + 1) * T / sk.G) * a.nk;
  }
#ifdef DLIP_STAMPS
  if (threadIdx.x == 0) sk.stamps[(size_t)g * 10 + 8] = __builtin_amdgcn_s_memrealtime();
#endif
  for (long long it = it_begin; it < it_end;) {
    const int tile = (int)(it / a.nk);
    const int k0 = (int)(it - (long long)tile * a.nk);
    const int kn = (int)((it_end - it) < (long long)(a.nk - k0) ? (it_end - it) : (long long)(a.nk - k0));
    // Tile order: output-channel block OUTER.  A workgroup's range, and (through the XCD remap of g) an
    // XCD's eighth of the launch, then stays on one 128-channel weight block -- BN x R x S x C x 4 bytes,
    // which fits the XCD's 4 MiB L2 where the whole filter bank of layer3/4 (2.4 / 9.4 MB) does not --
    // while the activation rows stream through once per block.
    const int tiles_m = (a.M + BM - 1) / BM;
    const int tile_n = tile / tiles_m;
    const int tile_m = tile - tile_n * tiles_m;
    // every wave is done reading the previous segment's last stage (and the ticket word) before the ring is refilled
    if (it != it_begin) __syncthreads();
    DLIP_STAMP(0);
#ifdef DLIP_STAMPS
    if (threadIdx.x == 0 && it == it_begin) sk.stamps[(size_t)g * 10 + 7] = __builtin_amdgcn_s_memrealtime();   // 100 MHz
#endif

    acc_t acc[MI][NI];
#define DLIP_FENCE() __builtin_amdgcn_sched_barrier(0)
    if constexpr (WIN) {
      // ================= window mode (16x16x32 program) =================
      const int tap0 = k0 % ntaps, cc0 = k0 / ntaps;
      // tap-validity bits of this lane's MI fragment pixels (the mask of the per-tap gather, for other rows)
      uint32_t fr_mask[MI];
      {
        int hi0[MI], wi0[MI];
        uint32_t colbits[MI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const int m = tile_m * BM + wm * WM + mi * 16 + lrow;
          const int mc = m < a.M ? m : a.M - 1;
          const int n = mc / a.HoWo;
          const int rem = mc - n * a.HoWo;
          const int ho = rem / a.Wo;
          hi0[mi] = ho - a.ph;
          wi0[mi] = rem - ho * a.Wo - a.pw;
          colbits[mi] = 0u;
          fr_mask[mi] = 0u;
        }
        for (int sx = 0; sx < a.S; ++sx)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) colbits[mi] |= (uint32_t)((unsigned)(wi0[mi] + sx * a.dw) < (unsigned)a.W) << sx;
        for (int r = 0; r < a.R; ++r)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
            fr_mask[mi] |= ((unsigned)(hi0[mi] + r * a.dh) < (unsigned)a.H ? colbits[mi] : 0u) << (r * a.S);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          if (tile_m * BM + wm * WM + mi * 16 + lrow >= a.M) fr_mask[mi] = 0u;
      }
      const int fr_row0 = wm * WM + lrow;   // window row of fragment block 0 at tap offset 0
      // window pieces of this wave: chunk planes wave, wave + NW, ..., 64-row blocks 0 .. WRP/64 - 1 of each; window
      // row r = input pixel tile_m BM - (ph W + pw) + r (a same-size convolution: output pixel m reads input
      // pixel m + tap offset).  Rows past BM + slack are never fetched: they stay zero (the masked-tap rows).
      // Offsets are rebuilt at issue time (once per R S slices) rather than held in registers.
      const int w_pix0 = tile_m * BM - (a.ph * a.W + a.pw) + lane;
      int b_off[B_PER];
#pragma unroll
      for (int j = 0; j < B_PER; ++j) {
        const int n = tile_n * BN + rbase + RPP * j;
        b_off[j] = n < a.K ? (n * a.rsc + csrc) * 4 : -1;
      }
      int itap = tap0, ic0 = cc0 * BK;
      int w_tap = (itap * a.Cw + ic0) * 4;
      auto advance = [&]() {
        if (++itap == ntaps) { itap = 0; ic0 += BK; }
        w_tap = (itap * a.Cw + ic0) * 4;
      };
      auto issue_b = [&](int stage) {
        const uint32_t base = piece0 + stage * STAGE_B;
#pragma unroll
        for (int j = 0; j < B_PER; ++j)
          dma_piece<DLIP_POL_B>(wr, b_off[j] >= 0 ? (uint32_t)(b_off[j] + w_tap) : DLIP_OOB_OFFSET, base + j * RPP * ROWB);
      };
      auto issue_win = [&](int slot, int cc) {
#pragma unroll
        for (int pl = 0; pl < 8 / NW; ++pl)
#pragma unroll
          for (int blk = 0; blk < WRP / 64; ++blk) {
            const int plane = pl * NW + wave;
            const int pix = w_pix0 + blk * 64;
            const bool ok = (blk * 64 + lane) < BM + DLIP_WIN_SLACK && pix >= 0 && pix < a.M;
            dma_piece<DLIP_POL_A>(xr, ok ? (uint32_t)((pix * a.ldx + cc * BK) * 4 + plane * 16) : DLIP_OOB_OFFSET,
                                  lds0 + WIN0 + slot * WIN_B + (plane * (WRP / 64) + blk) * 1024);
          }
      };
      auto wait_sel = [&](bool w, bool win) {   // leave the pieces issued after the awaited weight slice in flight
        if (w) { if (win) wait_vmcnt<B_PER + WPER>(); else wait_vmcnt<B_PER>(); }
        else   { if (win) wait_vmcnt<WPER>(); else wait_vmcnt<0>(); }
      };

      // ---- prologue ----
      DLIP_STAMP(1);
      issue_win(0, cc0);
      issue_b(0);
      const bool two = PF > 1 && kn > 1;
      if (two) { advance(); issue_b(1); }
      const bool win1 = (ntaps - tap0) < kn;   // the segment reaches the next channel slice
      if (win1) issue_win(1, cc0 + 1);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[mi][ni][e] = 0.f;
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
      // the tap being multiplied: index, window slot, row offset; a_ad = byte address of each block's hi chunk
      int ctap = tap0, cslot = 0, ccur = cc0;
      int cs = tap0 % a.S, crow = (tap0 / a.S) * a.dh * a.W;
      int a_ad[MI];
      const int a_base = WIN0 + (half * WRP + fr_row0) * 16;           // plane `k group`, this lane's row of block 0
      // masked taps read zeros from the last 16 rows of the plane (never fetched: rows >= BM + slack), at the
      // row with the same index mod 16 as the real one -- the same bank, so a group with masked lanes stays
      // conflict free (one shared zero row cost 35 % conflict cycles: it collides with one live lane per group)
      const int a_zero = WIN0 + (half * WRP + WRP - 16) * 16;
      auto set_addr = [&]() {
        const int toff = crow + cs * a.dw;
        const int off = cslot * WIN_B + toff * 16;
        const int zad = a_zero + (((lrow + toff) & 15) << 4);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          a_ad[mi] = ((fr_mask[mi] >> ctap) & 1u) ? a_base + mi * 256 + off : zad;
      };
      f16x8 fal[MI], fah[MI], fbh[NI], fbl[NI];
      const char* lds_c = reinterpret_cast<const char*>(smem);
      // weight fragment address (hi chunk; lo = ^ 64): rebuilt from the lane id behind an opaque asm where it is
      // used, so that it is not carried (and spilled: a scratch reload is a VMEM operation, and the vmcnt(0) the
      // compiler puts behind it would drain the whole DMA ring every slice) through the loop
      auto b_addr = [&]() {
        int t = tid;
        asm volatile("" : "+v"(t));
        const int lr = t & 15, kgp = (t >> 4) & 3;
        return (wn * WN + lr) * ROWB + ((kgp ^ ((lr >> 1) & 7)) << 4);
      };
      auto read_first = [&](int stage) {   // group 0: activation lo, weight hi
        const char* Bw = lds_c + stage * STAGE_B + b_addr();
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) fal[mi] = *reinterpret_cast<const f16x8*>(lds_c + a_ad[mi] + 4 * WRP * 16);   // lo planes 4..7
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) fbh[ni] = *reinterpret_cast<const f16x8*>(Bw + ni * 16 * ROWB);
      };
      auto read_rest = [&](int stage) {    // activation hi, weight lo
        const char* Bw = lds_c + stage * STAGE_B + (b_addr() ^ 64);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) fah[mi] = *reinterpret_cast<const f16x8*>(lds_c + a_ad[mi]);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) fbl[ni] = *reinterpret_cast<const f16x8*>(Bw + ni * 16 * ROWB);
      };
      auto mfma_p = [&](int grp, int m0, int m1) {   // grp 0: lo*hi, 1: hi*hi, 2: hi*lo
#pragma unroll
        for (int mi = m0; mi < m1; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            const f16x8 av = grp == 0 ? fal[mi] : fah[mi];
            const f16x8 bv = grp == 2 ? fbl[ni] : fbh[ni];
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bv, av, acc[mi][ni], 0, 0, 0);
          }
      };
      constexpr int MH = MI / 2 > 0 ? MI / 2 : 1;
      set_addr();
      DLIP_STAMP(2);
      wait_sel(two, win1);
      __syncthreads();   // (also publishes the parameter table)
      DLIP_STAMP(3);
      read_first(0);

      int st_cur = 0, st_iss = two ? 2 % NSTAGE : 1 % NSTAGE;
      for (int kt = 0; kt < kn; ++kt) {
        const bool more1 = (kt + 1) < kn, moreP = (kt + PF) < kn;
        const int st_nxt = st_cur + 1 == NSTAGE ? 0 : st_cur + 1;
        read_rest(st_cur); DLIP_FENCE();
        mfma_p(0, 0, MI); DLIP_FENCE();
        if (moreP) { advance(); issue_b(st_iss); st_iss = st_iss + 1 == NSTAGE ? 0 : st_iss + 1; }
        // entering a channel slice: fetch the next one's window into the slot the previous slice has left
        const bool winnow = kt > 0 && ctap == 0 && (kt + ntaps) < kn;
#ifdef DLIP_ABLATE_WIN   // timing experiment only (wrong results): in-loop windows are not fetched
        if (winnow) issue_win(cslot ^ 1, 1 << 24);
#else
        if (winnow) issue_win(cslot ^ 1, ccur + 1);
#endif
        DLIP_FENCE();
        // the next tap's fragment addresses: plain VALU, scheduled into the shadow of group 1's instructions
        // (this slice's activation fragments are in registers already)
        if (++cs == a.S) { cs = 0; crow += a.dh * a.W; }
        if (++ctap == ntaps) { ctap = 0; cs = 0; crow = 0; cslot ^= 1; ++ccur; }
        set_addr();
        mfma_p(1, 0, MI); DLIP_FENCE();
        mfma_p(2, 0, MH); DLIP_FENCE();
        if (more1) {
          wait_sel(PF > 1 && (kt + 2) < kn, winnow);
          __builtin_amdgcn_s_barrier();
          read_first(st_nxt);
        }
        DLIP_FENCE();
        if (MH < MI) mfma_p(2, MH, MI);
        DLIP_FENCE();
        st_cur = st_nxt;
      }
    } else {
    // Per-row gather state, branch-free: byte offset of the row's window origin and a bit per filter tap
    // that stays inside the image (columns and rows tested separately: R + S steps, not R x S).
    int a_off[A_PER];
    uint32_t a_mask[A_PER];
    {
      int hi0[A_PER], wi0[A_PER];
      uint32_t colbits[A_PER];
#pragma unroll
      for (int j = 0; j < A_PER; ++j) {
        const int m = tile_m * BM + rbase + RPP * j;
        const int mc = m < a.M ? m : a.M - 1;
        const int n = mc / a.HoWo;
        const int rem = mc - n * a.HoWo;
        const int ho = rem / a.Wo;
        const int wo = rem - ho * a.Wo;
        hi0[j] = ho * a.sh - a.ph;
        wi0[j] = wo * a.sw - a.pw;
        a_off[j] = (((n * a.H + hi0[j]) * a.W + wi0[j]) * a.ldx + csrc) * 4;
        colbits[j] = 0u;
        a_mask[j] = 0u;
      }
      for (int sx = 0; sx < a.S; ++sx)
#pragma unroll
        for (int j = 0; j < A_PER; ++j) colbits[j] |= (uint32_t)((unsigned)(wi0[j] + sx * a.dw) < (unsigned)a.W) << sx;
      for (int r = 0; r < a.R; ++r)
#pragma unroll
        for (int j = 0; j < A_PER; ++j)
          a_mask[j] |= ((unsigned)(hi0[j] + r * a.dh) < (unsigned)a.H ? colbits[j] : 0u) << (r * a.S);
#pragma unroll
      for (int j = 0; j < A_PER; ++j)
        if (tile_m * BM + rbase + RPP * j >= a.M) a_mask[j] = 0u;
    }
    int b_off[B_PER];
#pragma unroll
    for (int j = 0; j < B_PER; ++j) {
      const int n = tile_n * BN + rbase + RPP * j;
      b_off[j] = n < a.K ? (n * a.rsc + csrc) * 4 : -1;
    }

    // Reduction walk: 32-channel slice OUTER, filter tap INNER (conv_igemm_f16x3.hip), entered at slice k0.
    int c0 = (k0 / ntaps) * BK, tap = k0 % ntaps;
    int s_pos = tap % a.S, x_row = (tap / a.S) * x_dr;
    int x_tap = x_row + s_pos * x_ds + c0 * 4, w_tap = (tap * a.Cw + c0) * 4;
    auto advance = [&]() {
      ++tap;
      if (++s_pos == a.S) { s_pos = 0; x_row += x_dr; }
      if (tap == ntaps) { tap = 0; s_pos = 0; x_row = 0; c0 += BK; }
      x_tap = x_row + s_pos * x_ds + c0 * 4;
      w_tap = (tap * a.Cw + c0) * 4;
    };
    auto issue_a = [&](int stage) {
#ifdef DLIP_ABLATE_A   // timing experiment only (wrong results): no activation traffic after the prologue
      if (stage >= 0 && tap + c0 != (k0 % ntaps) + (k0 / ntaps) * BK) return;
#endif
      const uint32_t base = piece0 + stage * STAGE_B;
#pragma unroll
      for (int j = 0; j < A_PER; ++j) {
        bool ok = (a_mask[j] >> tap) & 1u;
#ifdef DLIP_ABLATE_OOB   // timing experiment only (wrong results): every piece after the prologue is issued out of range (zeros, no L2 traffic)
        ok = ok && (tap + c0 == (k0 % ntaps) + (k0 / ntaps) * BK);
#endif
        dma_piece<DLIP_POL_A>(xr, ok ? (uint32_t)(a_off[j] + x_tap) : DLIP_OOB_OFFSET, base + j * RPP * ROWB);
      }
    };
    auto issue_b = [&](int stage) {
#ifdef DLIP_ABLATE_B   // timing experiment only (wrong results): no weight traffic after the prologue
      if (stage >= 0 && tap + c0 != (k0 % ntaps) + (k0 / ntaps) * BK) return;
#endif
      const uint32_t base = piece0 + stage * STAGE_B + BM * ROWB;
#pragma unroll
      for (int j = 0; j < B_PER; ++j) {
        bool bok = b_off[j] >= 0;
#ifdef DLIP_ABLATE_OOB
        bok = bok && (tap + c0 == (k0 % ntaps) + (k0 / ntaps) * BK);
#endif
        dma_piece<DLIP_POL_B>(wr, bok ? (uint32_t)(b_off[j] + w_tap) : DLIP_OOB_OFFSET, base + j * RPP * ROWB);
      }
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
        for (int e = 0; e < 4 * QN; ++e) acc[mi][ni][e] = 0.f;
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

    if constexpr (!M16) {
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
    auto mfma_g = [&](int set, int grp) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const f16x8 av = grp == 0 ? fal[set][mi] : fah[set][mi];
          const f16x8 bv = grp == 1 ? fbl[set][ni] : fbh[set][ni];
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bv, av, acc[mi][ni], 0, 0, 0);
        }
    };

    // slice 0 has landed once at most the (PF - 1) younger slices are outstanding
    DLIP_STAMP(2);
    if (PF > 1 && kn > 1) wait_vmcnt<NL>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    DLIP_STAMP(3);
    read_frags(0, 0, 0);

    int st_cur = 0, st_iss = (PF > 1 && kn > 1) ? 2 % NSTAGE : 1 % NSTAGE;   // stage the next issue goes to
    for (int kt = 0; kt < kn; ++kt) {
      const bool more1 = (kt + 1) < kn, moreP = (kt + PF) < kn;
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
        if (PF > 1 && (kt + 2) < kn) wait_vmcnt<NL>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        read_frags(0, st_nxt, 0);
      }
      DLIP_FENCE();
      mfma_g(1, 2); DLIP_FENCE();
      st_cur = st_nxt;
    }
    } else {
    // ---- 16x16x32 program: per slice 3 groups of MI x NI instructions.  Group order lo*hi, hi*hi, hi*lo:
    // the last group needs neither the activation-lo nor the weight-hi fragments, so the NEXT slice's first
    // group's fragments are read (behind the barrier) into registers the tail of this slice does not use. ----
    f16x8 fal[MI], fah[MI], fbh[NI], fbl[NI];
    auto read_first = [&](int stage) {   // what group 0 needs: activation lo, weight hi
      const float* Aw = smem + stage * (STAGE_B / 4) + a_frag;
      const float* Bw = smem + stage * (STAGE_B / 4) + b_frag;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) fal[mi] = *reinterpret_cast<const f16x8*>(Aw + mi * 16 * LDK + klo[0]);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) fbh[ni] = *reinterpret_cast<const f16x8*>(Bw + ni * 16 * LDK + khi[0]);
    };
    auto read_rest = [&](int stage) {    // activation hi, weight lo
      const float* Aw = smem + stage * (STAGE_B / 4) + a_frag;
      const float* Bw = smem + stage * (STAGE_B / 4) + b_frag;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) fah[mi] = *reinterpret_cast<const f16x8*>(Aw + mi * 16 * LDK + khi[0]);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) fbl[ni] = *reinterpret_cast<const f16x8*>(Bw + ni * 16 * LDK + klo[0]);
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
    DLIP_STAMP(2);
    if (PF > 1 && kn > 1) wait_vmcnt<NL>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    DLIP_STAMP(3);
    read_first(0);

    int st_cur = 0, st_iss = (PF > 1 && kn > 1) ? 2 % NSTAGE : 1 % NSTAGE;
    for (int kt = 0; kt < kn; ++kt) {
      const bool more1 = (kt + 1) < kn, moreP = (kt + PF) < kn;
      const int st_nxt = st_cur + 1 == NSTAGE ? 0 : st_cur + 1;
      read_rest(st_cur); DLIP_FENCE();
      mfma_p(0, 0, MI); DLIP_FENCE();
      if (moreP) { advance(); issue_a(st_iss); } DLIP_FENCE();
      mfma_p(1, 0, MH); DLIP_FENCE();
      if (moreP) { issue_b(st_iss); st_iss = st_iss + 1 == NSTAGE ? 0 : st_iss + 1; } DLIP_FENCE();
      if (MH < MI) mfma_p(1, MH, MI);
      DLIP_FENCE();
      mfma_p(2, 0, MH); DLIP_FENCE();
      if (more1) {
        if (PF > 1 && (kt + 2) < kn) wait_vmcnt<NL>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        read_first(st_nxt);
      }
      DLIP_FENCE();
      if (MH < MI) mfma_p(2, MH, MI);
      DLIP_FENCE();
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
    const int lane_e = tid_e & 63, lrow_e = tid_e & (FR - 1), half_e = (tid_e & 63) / FR;   // pixel in block; channel quad (k half / k group)

    bool finish = true;
    if (kn != a.nk) {   // split tile (workgroup-uniform branch)
      constexpr int SLAB = BM * BN;   // floats
      volatile int* bcast = reinterpret_cast<volatile int*>(smem);
      const long long t0 = (long long)tile * a.nk;
      const int gf = (int)(((t0 + 1) * sk.G - 1) / sk.iters);            // owner of the tile's first slice
      const int gl = (int)(((t0 + a.nk) * sk.G - 1) / sk.iters);         // owner of its last slice
      const int others = gl - gf;                                        // parts besides this one
      // 1. peek: if every other part has already published, this workgroup is the last one and keeps its
      //    part in registers (the usual case for a range's final, head-of-tile segment: the neighbour
      //    computed the rest of that tile first thing).
      __syncthreads();   // all waves are past their last fragment reads: LDS word 0 is free
      if (tid_e == 0) bcast[0] = __hip_atomic_load(sk.counters + tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      finish = bcast[0] == others;
      if (!finish) {
        // 2. publish: write-through (sc1) slab stores, drained by every storing wave, then ONE ticket
        const __amdgpu_buffer_rsrc_t sr = dlip_make_rsrc(sk.slabs + (size_t)(2 * g + (it != it_begin ? 1 : 0)) * SLAB, SLAB * 4);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int q = 0; q < QN; ++q) {
              f32x4 v;   // (bit_cast straight from a vector-element lvalue reads element 0: copy out first)
              v[0] = acc[mi][ni][4 * q]; v[1] = acc[mi][ni][4 * q + 1]; v[2] = acc[mi][ni][4 * q + 2]; v[3] = acc[mi][ni][4 * q + 3];
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), sr, (((mi * NI + ni) * QN + q) * NT + tid_e) * 16, 0, 16);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid_e == 0) bcast[0] = __hip_atomic_fetch_add(sk.counters + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        finish = bcast[0] == others;   // the other parts arrived between the peek and the ticket
      }
      if (finish) {
        if (tid_e == 0) __hip_atomic_store(sk.counters + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // no instruction: keeps the slab loads below the poll
        // sum the parts in part order (own part from registers): the bits do not depend on who came last;
        // every slab load is sc1 (served past this CU's L1), matching the sc1 stores
        acc_t tot[MI][NI];
        for (int p = gf; p <= gl; ++p) {
          const long long pb = (long long)p * sk.iters / sk.G;
          const __amdgpu_buffer_rsrc_t pr = dlip_make_rsrc(sk.slabs + (size_t)(2 * p + (pb < t0 ? 1 : 0)) * SLAB, SLAB * 4);
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
              for (int q = 0; q < QN; ++q) {
                f32x4 v;
                if (p == g) {
                  v[0] = acc[mi][ni][4 * q]; v[1] = acc[mi][ni][4 * q + 1]; v[2] = acc[mi][ni][4 * q + 2]; v[3] = acc[mi][ni][4 * q + 3];
                } else {
                  v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(pr, (((mi * NI + ni) * QN + q) * NT + tid_e) * 16, 0, 16));
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) tot[mi][ni][4 * q + c] = p == gf ? v[c] : tot[mi][ni][4 * q + c] + v[c];
                __builtin_amdgcn_sched_barrier(0);
              }
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = tot[mi][ni];
      }
    }

    if (finish) {
      // ---- epilogue through LDS: y = act(acc / wscale + bias + residual) * post_scale + post_shift ----
      // The ring is free now and stages the output tile in its memory layout (BM rows x BN*4 bytes, in
      // EPASS row bands when the tile is larger than the ring), 16-B chunk c of row r at position
      // c ^ (r & 15) (low 4 bits): the residual arrives by LDS-DMA, every lane adds / overwrites the 8-B
      // pieces of its own pixels, and the tile leaves in 16-B stores (a row is one contiguous segment).
      typedef _Float16 h4 __attribute__((ext_vector_type(4)));
      constexpr int PITCH = BN * 4;                  // bytes per image row
      constexpr int CPR = BN / 4;                    // 16-B chunks per row
      constexpr int EPASS = (BM * PITCH + RING - 1) / RING;
      constexpr int PROWS = BM / EPASS;              // rows per band
      static_assert(BM % EPASS == 0 && PROWS % WM == 0 && PROWS * PITCH <= RING, "epilogue bands are whole wave rows");
      static_assert((PROWS * PITCH) % (NW * 1024) == 0, "a band is a whole number of DMA pieces per wave");
      constexpr int RES_PIECES = PROWS * PITCH / 1024 / NW;   // per wave
      constexpr int RPQ = 1024 / PITCH;              // rows per DMA piece
      const u32x4 rrw = make_rsrc_words(a.res, a.res ? a.r_bytes : 0u);
      const int kcol0 = tile_n * BN;                 // first output channel of the tile
      char* img = reinterpret_cast<char*>(smem);
      const f32x4* tab = reinterpret_cast<const f32x4*>(smem + RING / 4);
      const bool post = a.pscale != nullptr;
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
          for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int j = 0; j < QN; ++j) {
              const int kl = wn * WN + ni * FR + (M16 ? 4 * half_e : 8 * j + 4 * half_e);   // tile-local channel of acc[..][ni][4j..4j+3]
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
                for (int c = 0; c < 4; ++c) v[c] = acc[mi][ni][4 * j + c] * inv4[c] + bi4[c];
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
                  *reinterpret_cast<h4*>(row + phi * 16) = hi;
                  *reinterpret_cast<h4*>(row + plo * 16) = lo;
                } else {
#pragma unroll
                  for (int c = 0; c < 4; ++c) acc[mi][ni][4 * j + c] = v[c];
                }
              }
              __builtin_amdgcn_sched_barrier(0);   // one channel quad at a time
            }
        }
        if constexpr (!OSPLIT) {
          if (a.res) __syncthreads();                // fp32 rows overwrite other lanes' residual pieces
          if (mine) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
              for (int j = 0; j < QN; ++j) {
                const int kl = wn * WN + ni * FR + (M16 ? 4 * half_e : 8 * j + 4 * half_e);
                const int ch = kl >> 2;              // the 4 channels are one fp32 chunk
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                  const int r = wm * WM - band0 + mi * FR + lrow_e;
                  const int pc = (ch & ~15) | ((ch ^ r) & 15);
                  f32x4 v;
#pragma unroll
                  for (int c = 0; c < 4; ++c) v[c] = acc[mi][ni][4 * j + c];
                  *reinterpret_cast<f32x4*>(img + r * PITCH + pc * 16) = v;
                }
              }
          }
        }
        __syncthreads();
        // band -> global: thread t moves chunks t, t + NT, ... (a wave-instruction covers 64 / CPR whole rows)
#pragma unroll
        for (int i = 0; i < PROWS * CPR / NT; ++i) {
          const int idx = i * NT + tid_e;
          const int r = idx / CPR, pp = idx % CPR;
          const int c = (pp & ~15) | ((pp ^ r) & 15);
          const int m = tile_m * BM + band0 + r;
          const int kfirst = OSPLIT ? kcol0 + (c >> 3) * 32 : kcol0 + c * 4;
          const u32x4 v = *reinterpret_cast<const u32x4*>(img + r * PITCH + pp * 16);
          const uint32_t off = (m < a.M && kfirst < a.K) ? (uint32_t)((m * a.ldy + kcol0) * 4 + c * 16) : DLIP_OOB_OFFSET;
          __builtin_amdgcn_raw_buffer_store_b128(v, yr, (int)off, 0, 0);
        }
        if (ep + 1 < EPASS) __syncthreads();         // the next band reuses the image
      }
    }
#ifdef DLIP_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0 && it == it_begin) {
      sk.stamps[(size_t)g * 10 + 5] = __builtin_amdgcn_s_memtime();
      sk.stamps[(size_t)g * 10 + 6] = (unsigned long long)kn;
      sk.stamps[(size_t)g * 10 + 7] = __builtin_amdgcn_s_memrealtime() - sk.stamps[(size_t)g * 10 + 7];
    }
#endif
    it += kn;
  }
#ifdef DLIP_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (threadIdx.x == 0) {
    sk.stamps[(size_t)g * 10 + 9] = __builtin_amdgcn_s_memrealtime();
    sk.stamps[(size_t)g * 10 + 6] |= (unsigned long long)(blockIdx.x & 7) << 32;
  }
#endif
}

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
constexpr int kMaxPartsPerTile = 16;      // balanced split: upper bound on the workgroups sharing one tile
constexpr double kHandoffUs = 10.0;       // cost of the slab hand-off of a launch at 128x128 tiles (measured)

std::mutex& ws_mutex() { static std::mutex m; return m; }
std::map<std::pair<int, hipStream_t>, Workspace>& ws_table() {
  static std::map<std::pair<int, hipStream_t>, Workspace> t;
  return t;
}

Workspace* workspace_for(hipStream_t st, size_t slab_floats) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(ws_mutex());
  Workspace& w = ws_table()[{dev, st}];
  if (w.external) return w.slab_floats >= slab_floats ? &w : nullptr;
  if (w.counters == nullptr) {
    if (hipMalloc(reinterpret_cast<void**>(&w.counters), kMaxSplitTiles * sizeof(int)) != hipSuccess) return nullptr;
    if (hipMemset(w.counters, 0, kMaxSplitTiles * sizeof(int)) != hipSuccess) return nullptr;
  }
  if (w.slab_floats < slab_floats) {
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

int resident_workgroups(const void* kern, int threads, size_t lds) {
  int dev = 0, cus = 0, per_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, threads, lds) != hipSuccess) return 0;
  return cus * per_cu;
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int NSTAGE, int OCC, bool M16 = false, bool WIN = false>
int launch_dma(const ConvArgs& a, hipStream_t st, bool out_split) {
  ConvArgs b = a;
  const int tiles_m = (a.M + BM - 1) / BM;
  b.tiles_n = (a.K + BN - 1) / BN;
  const long long tiles = (long long)tiles_m * b.tiles_n;
  if (tiles <= 0 || tiles > 0x7FFFFFFFll) return DLIP_EINVAL;
  constexpr size_t lds = (WIN ? (size_t)dlip_win_ring_bytes(BM, BN, NSTAGE) : (size_t)NSTAGE * (BM + BN) * ROWB)
                         + 5 * BN * sizeof(float);   // ring (+ windows) + epilogue parameter table
  constexpr int threads = 64 * WAVES_M * WAVES_N;
  static_assert(lds <= 160 * 1024, "LDS ring exceeds a CU");
  auto kern = out_split ? conv_igemm_f16x3_dma_kernel<BM, BN, WAVES_M, WAVES_N, true, NSTAGE, OCC, M16, WIN>
                        : conv_igemm_f16x3_dma_kernel<BM, BN, WAVES_M, WAVES_N, false, NSTAGE, OCC, M16, WIN>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  // One slot table per kernel instance (the occupancy query is not free).
  static int slots[2] = {0, 0};
  int& sl = slots[out_split ? 1 : 0];
  if (sl == 0) sl = resident_workgroups(reinterpret_cast<const void*>(kern), threads, lds);
  if (sl <= 0) return DLIP_EINVAL;

  StreamK sk;
  sk.iters = tiles * a.nk;
  sk.slabs = nullptr;
  sk.counters = nullptr;
  long long G = tiles;                                  // plain launch: one tile per workgroup
  static const int balanced = [] { const char* e = getenv("DLIP_CONV_STREAMK"); return e ? atoi(e) : 1; }();   // 0 never, 2 always
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
    if (Gb > tiles * kMaxPartsPerTile) Gb = tiles * kMaxPartsPerTile;
    const double bal_us = (double)sk.iters / Gb / a.nk * tile_us + kHandoffUs * (BM * BN / 16384.0);
    if ((balanced == 2 || bal_us < plain_us) && Gb * a.nk != sk.iters) G = Gb;
  }
  sk.whole = 0;
  if (balanced == 3 && tiles > 4 * sl) {   // experiment: persistent workgroups over whole tiles (no slabs, no tickets)
    G = sl;
    sk.whole = 1;
  }
  if (G != tiles && !sk.whole) {
    Workspace* w = workspace_for(st, (size_t)2 * G * BM * BN);
    if (w == nullptr) {
      G = tiles;   // no (or too small a) workspace: plain launch
    } else {
      sk.slabs = w->slabs;
      sk.counters = w->counters;
    }
  }
  sk.G = (int)G;
#ifdef DLIP_STAMPS
  {   // diagnostic build (tools/probes/stamps.sh): median cycles between the stamps of each workgroup's first segment
    static unsigned long long* dbuf = nullptr;
    static size_t cap = 0;
    if (cap < (size_t)G * 10) { if (dbuf) (void)hipFree(dbuf); (void)hipMalloc(reinterpret_cast<void**>(&dbuf), (size_t)G * 10 * 8); cap = (size_t)G * 10; }
    (void)hipMemsetAsync(dbuf, 0, (size_t)G * 10 * 8, st);
    sk.stamps = dbuf;
    hipLaunchKernelGGL(kern, dim3((unsigned)G), dim3(threads), lds, st, b, sk);
    (void)hipStreamSynchronize(st);
    std::vector<unsigned long long> h((size_t)G * 10);
    (void)hipMemcpy(h.data(), dbuf, h.size() * 8, hipMemcpyDeviceToHost);
    if (getenv("DLIP_STAMP_PRINT")) {
      std::vector<double> d[5], per, clk;
      for (long long i = 0; i < G; ++i) {
        const unsigned long long* r = &h[(size_t)i * 10];
        if (!r[5]) continue;
        for (int j = 0; j < 5; ++j) d[j].push_back((double)(r[j + 1] - r[j]));
        per.push_back((double)(r[4] - r[3]) / (double)((r[6] & 0xffffffffull) ? (r[6] & 0xffffffffull) : 1));
        if (r[7]) clk.push_back((double)(r[5] - r[0]) / (double)r[7] * 100.0);   // MHz: s_memtime ticks per 100 MHz s_memrealtime tick
      }
      unsigned long long t0 = ~0ull, t1 = 0, s1 = 0;
      std::vector<double> dur;
      for (long long i = 0; i < G; ++i) {
        const unsigned long long* r = &h[(size_t)i * 10];
        if (!r[9]) continue;
        t0 = r[8] < t0 ? r[8] : t0; s1 = r[8] > s1 ? r[8] : s1; t1 = r[9] > t1 ? r[9] : t1;
        dur.push_back((double)(r[9] - r[8]) / 100.0);
      }
      {   // busy time by XCD and by number of segments in the workgroup's range
        double xs[8] = {0}, ss[4] = {0}; int xn[8] = {0}, sn[4] = {0};
        for (long long i = 0; i < G; ++i) {
          const unsigned long long* r = &h[(size_t)i * 10];
          if (!r[9]) continue;
          const double d = (double)(r[9] - r[8]) / 100.0;
          const int x = (int)((r[6] >> 32) & 7);
          xs[x] += d; xn[x]++;
          const long long b0 = i * sk.iters / G, b1 = (i + 1) * sk.iters / G;
          const int nseg = (int)((b1 - 1) / a.nk - b0 / a.nk) + 1;
          ss[nseg < 4 ? nseg : 3] += d; sn[nseg < 4 ? nseg : 3]++;
        }
        fprintf(stderr, "[stamps wall] mean busy by XCD:");
        for (int x = 0; x < 8; ++x) fprintf(stderr, " %.1f", xn[x] ? xs[x] / xn[x] : 0.0);
        fprintf(stderr, "   by segments 1/2/3+:");
        for (int k = 1; k < 4; ++k) fprintf(stderr, " %.1f(n=%d)", sn[k] ? ss[k] / sn[k] : 0.0, sn[k]);
        fprintf(stderr, "\n");
      }
      if (const char* dump = getenv("DLIP_STAMP_DUMP")) {
        if (FILE* fp = fopen(dump, "w")) {
          for (long long i = 0; i < G; ++i) {
            const unsigned long long* r = &h[(size_t)i * 10];
            fprintf(fp, "%lld,%d,%.2f,%.2f\n", i, (int)((r[6] >> 32) & 7), (double)(r[8] - t0) / 100.0, (double)(r[9] - r[8]) / 100.0);
          }
          fclose(fp);
        }
      }
      std::sort(dur.begin(), dur.end());
      if (!dur.empty())
        fprintf(stderr, "[stamps wall] kernel span %.1f us; workgroup starts spread %.1f us; workgroup busy min %.1f med %.1f max %.1f us\n",
                (double)(t1 - t0) / 100.0, (double)(s1 - t0) / 100.0, dur.front(), dur[dur.size() / 2], dur.back());
      auto med = [](std::vector<double>& v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
      fprintf(stderr, "[stamps %dx%d M=%d K=%d nk=%d G=%lld tiles=%lld] setup %.0f  issue+init %.0f  first-wait %.0f  loop %.0f (%.0f/slice)  tail %.0f  clock %.0f MHz\n",
              BM, BN, a.M, a.K, a.nk, G, tiles, med(d[0]), med(d[1]), med(d[2]), med(d[3]), med(per), med(d[4]), med(clk));
    }
    return dlip_launch_status();
  }
#endif
  hipLaunchKernelGGL(kern, dim3((unsigned)G), dim3(threads), lds, st, b, sk);
  return dlip_launch_status();
}

}  // namespace

// Tile menu of the DMA kernel (index = what dlip_conv_plan reports via dlip_conv_dma_tile).  Entries 0..5 are the
// product instances, on v_mfma_f32_16x16x32_f16: same cycles per FLOP as 32x32x16 but the chip holds a
// higher clock under it -- 3-8 % less time per layer, same box, interleaved runs (tools/bench_dma.py
// --variants 0,10,...); 10..14 are tiles 0..4 on v_mfma_f32_32x32x16_f16, the rest are experiments.
const TileCfg kDmaCfg[] = {{128, 128}, {128, 64}, {64, 128}, {64, 64}, {128, 64}, {256, 128},
                           {256, 128}, {128, 256}, {256, 64}, {128, 128},                 // 6..9: experiments (DLIP_CONV_DMA_TILE only)
                           {128, 128}, {128, 64}, {64, 128}, {64, 64}, {128, 64},        // 10..14: 0..4 on v_mfma_f32_32x32x16_f16
                           {128, 128}, {256, 128}, {128, 256}, {256, 64},                 // 15..18: experiments
                           {256, 128}, {128, 64}, {256, 64}, {128, 128}};                 // 19..22: window mode (same-size stride-1 only)
constexpr int NUM_DMA_ALL = 23;

// Window mode applies to same-size stride-1 convolutions whose taps span at most DLIP_WIN_SLACK pixels.
static bool win_ok(const ConvArgs& a) {
  return a.sh == 1 && a.sw == 1 && a.Wo == a.W && a.HoWo == a.H * a.W && a.R * a.S >= 3 &&
         (a.R - 1) * a.dh * a.W + (a.S - 1) * a.dw <= DLIP_WIN_SLACK;
}

// Tile choice (measured per layer with tools/bench_dma.py, MI355X, balanced split on).  The cost of a slice
// is set by the bytes it pulls from L2 (ablations: pieces issued out of range cost nothing, pieces that fetch
// cost 30 % of a layer), so the deepest layers take the tile with the fewest bytes per FLOP: 256x128, eight
// waves, one workgroup per CU, three-stage ring (5-11 % less time than two co-resident 128x128 workgroups
// on the 3x3 layers of layer2..4).  128x128 (two per CU) serves mid-depth reductions and the M < 8192
// launches; narrow outputs (K <= 64) do better on 128x64 with a three-stage ring, very short reductions
// (nk <= 8 slices: the 1x1 down-sampling convolutions) on 128x64 with a two-stage ring and three
// workgroups per CU (latency, not MFMA, bounds them); M <= 64 (fully connected layers on a batch) uses
// the 64-row tiles.
static int dma_pick(long long M, int K, int nk) {
  if (const char* e = getenv("DLIP_CONV_DMA_TILE")) {
    const int v = atoi(e);
    if (v >= 0 && v < NUM_DMA_ALL) return v;
  }
  if (M <= 64) return K <= 64 ? 3 : 2;
  if (K <= 64) return 1;
  if (nk <= 8) return 4;
  if (nk >= 32 && M >= 8192) return 5;
  return 0;
}

// Library-internal entry points (hidden): ConvArgs lives in an unnamed namespace, so it crosses the
// translation-unit boundary as an opaque pointer.
extern "C" __attribute__((visibility("hidden"))) void dlip_conv_dma_tile(long long M, int K, int nk, int* bm, int* bn) {
  const TileCfg& c = kDmaCfg[dma_pick(M, K, nk)];
  *bm = c.bm;
  *bn = c.bn;
}

// Called by dlip_conv_nhwc_f16x3 for DLIP_SPLIT_IN launches (argument checks done there).
extern "C" __attribute__((visibility("hidden"))) int dlip_conv_f16x3_dma_launch(const void* args, void* stream, int out_split) {
  const ConvArgs& a = *static_cast<const ConvArgs*>(args);
  hipStream_t st = static_cast<hipStream_t>(stream);
  int pick = dma_pick(a.M, a.K, a.nk);
  if (pick >= 19 && !win_ok(a)) pick = pick == 19 ? 5 : pick == 20 ? 1 : pick == 21 ? 18 : 0;
  switch (pick) {
    case 0: return launch_dma<128, 128, 2, 2, 2, 2, true>(a, st, out_split);
    case 1: return launch_dma<128, 64, 2, 2, 3, 2, true>(a, st, out_split);
    case 2: return launch_dma<64, 128, 2, 2, 3, 2, true>(a, st, out_split);
    case 3: return launch_dma<64, 64, 2, 2, 3, 2, true>(a, st, out_split);
    case 4: return launch_dma<128, 64, 2, 2, 2, 3, true>(a, st, out_split);
    case 5: return launch_dma<256, 128, 4, 2, 3, 1, true>(a, st, out_split);
    case 6: return launch_dma<256, 128, 4, 2, 3, 1>(a, st, out_split);
    case 7: return launch_dma<128, 256, 2, 4, 2, 1>(a, st, out_split);
    case 8: return launch_dma<256, 64, 4, 2, 2, 1>(a, st, out_split);
    case 9: return launch_dma<128, 128, 4, 2, 3, 1>(a, st, out_split);
    case 10: return launch_dma<128, 128, 2, 2, 2, 2>(a, st, out_split);
    case 11: return launch_dma<128, 64, 2, 2, 3, 2>(a, st, out_split);
    case 12: return launch_dma<64, 128, 2, 2, 3, 2>(a, st, out_split);
    case 13: return launch_dma<64, 64, 2, 2, 3, 2>(a, st, out_split);
    case 14: return launch_dma<128, 64, 2, 2, 2, 3>(a, st, out_split);
    case 15: return launch_dma<128, 128, 2, 2, 3, 1>(a, st, out_split);
    case 16: return launch_dma<256, 128, 4, 2, 2, 1, true>(a, st, out_split);
    case 17: return launch_dma<128, 256, 2, 4, 2, 1, true>(a, st, out_split);
    case 18: return launch_dma<256, 64, 4, 2, 3, 1, true>(a, st, out_split);
    case 19: return launch_dma<256, 128, 4, 2, 3, 1, true, true>(a, st, out_split);
    case 20: return launch_dma<128, 64, 2, 2, 3, 2, true, true>(a, st, out_split);
    case 21: return launch_dma<256, 64, 4, 2, 3, 1, true, true>(a, st, out_split);
    default: return launch_dma<128, 128, 2, 2, 2, 2, true, true>(a, st, out_split);
  }
}

// ---- caller-owned workspace (include/deeplip_hip.h) ----
extern "C" int64_t dlip_conv_workspace_bytes(void) {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
    return -1;
  // counters + two slabs per resident workgroup: 2 per CU x 128x128 fp32 is the largest product on the menu
  return (int64_t)kMaxSplitTiles * 4 + (int64_t)2 * (2 * cus) * 128 * 128 * 4;
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

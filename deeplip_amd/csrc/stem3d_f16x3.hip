// Video stem in split-fp16 arithmetic: Conv3d(1 -> 64, 5x7x7, stride (1,2,2), pad (2,3,3)) + folded
// BN + PReLU/ReLU, output [(B*T), Ho, Wo, 64] fp32 (replaces models/video_models/model.py:81-84, like
// stem3d.hip).  Products are hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_f16 with fp32 accumulation
// (see conv_igemm_f16x3.hip for the numerics argument).
//
// GEMM view per workgroup: M = 8 output rows x Wo pixels (352 px = 22 tiles of 16), N = 64 channels,
// K = 36 "kernel rows" (kt, kh) x 8 taps (7 real kw + 1 zero) = 288, walked in 9 steps of 32.
//   * the 5-frame x 21-row x (W+6)-column input window is split ONCE into (hi, lo) fp16 pairs (one
//     dword per pixel) while it is staged in LDS;
//   * a lane's 8 consecutive k of a step are the 8 taps of ONE kernel row = 8 consecutive window
//     dwords: four ds_read_b64 (8-B aligned because the pixel stride is 2) + eight v_perm_b32 to
//     de-interleave hi and lo -- no per-tap address arithmetic;
//   * the 64 x 288 split weights (host-packed, per-channel power-of-two scale, hi and lo planes, channel
//     stride 1184 B so the 16 lanes of every fragment-read group hit 16 different bank quads) live in LDS for the whole
//     workgroup: B fragments are two ds_read_b128 per 16-channel tile;
//   * 8 waves = 8 pixel groups (3,3,3,3,3,3,2,2 tiles), each against all 64 channels: one gathered A
//     fragment feeds 12 MFMAs (an f16 MFMA leaves only half of its 16 cycles for other issue, and the
//     de-interleave costs 8 VALU per fragment);
//   * persistent workgroups (one per CU, 154 KB LDS): weights staged once, the next tile's window is
//     fetched into registers while the current tile is multiplied and lands in a second LDS buffer.
//
// The fused stem + max-pool kernel (dlip_stem3d_pool_f16x3) is the second kernel of this file: same weights
// image, window layout and tap gather, different pixel -> lane map (chosen for the pooling) and window transport.
#include "conv_dma_common.h"
#include <algorithm>
#include <type_traits>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int KT = 5, KH = 7;
constexpr int KROWS = 36;                 // 35 (kt,kh) kernel rows + 1 zero row
constexpr int STEPS = KROWS / 4;          // 9 k32 steps (4 kernel rows of 8 taps each)
constexpr int ROWS = 8;                   // output rows per workgroup
constexpr int PR = 2 * ROWS + 5;          // 21 input rows
// Per output channel: [36 kernel rows x 16 B of hi][36 x 16 B of lo] + 32 B pad = 1184 B = 74 sixteen-byte
// slots.  A B-fragment read (ds_read_b128) is served in 16-lane groups that mix 8 channels at k group q with
// 8 other channels at k group q + 1: with a channel stride of 74 = 10 (mod 16) slots and a k-group stride of ONE
// slot (hi and lo in separate planes) the 16 addresses of every group fall into 16 different bank quads.  The
// first layout ([36][hi | lo] + 16 B: channel stride 9, k-group stride 2 slots) put two pairs of lanes of
// every group on the same banks -- a third of the kernel's LDS cycles were bank conflicts (PMC).
constexpr int WLO_OFF = KROWS * 16;         // 576: byte offset of the lo plane inside a channel
constexpr int WCH_BYTES = KROWS * 32 + 32;  // 1184
constexpr int WBYTES = 64 * WCH_BYTES;      // 75776
constexpr int MTW = 3;                    // pixel tiles per wave (each against all 4 channel tiles)

struct StemArgs {
  const float* x;
  const uint32_t* w;   // packed split weights, WBYTES per 64 channels (LDS image)
  const float* wscale;
  const float* bias;
  const float* slope;
  float* y;
  int T, H, W, Ho, Wo;
  int row_tiles;
  int n_tiles;         // B*T*row_tiles
  uint32_t x_bytes, y_bytes;
  int pwp;             // window row pitch in dwords (>= W + 6, == 32 mod 64)
  int Hp, Wp;          // pooled output size (stem + pool kernel)
  int n_frames;        // B*T
  DlipRange status;    // range reporting of the split-format output (dlip_common.h)
  unsigned long long* span;     // NULL, or {first start, last end} of this launch in 100 MHz ticks (dlip_span_scope_*): the pre-pass notes the start
#ifdef DLIP_LAB
  unsigned long long* stamps;   // lab build: [grid][8] s_memtime of each workgroup's second tile
  int lab_skip_last_step;       // lab probe (DLIP_STEM_SKIP_STEP8=1; WRONG results, timing only): the reduction without its 9th k-step --
                                // what a K = 256 repack (8 steps instead of 9) could gain at most, before it pays for its gather
#endif
};

#ifdef DLIP_LAB
#define STEM_STAMP(i) do { if (threadIdx.x == 0 && it == 1 && a.stamps) a.stamps[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define STEM_WSTAMP(i) do { if ((threadIdx.x & 63) == 0 && it == 2 && a.stamps) a.stamps[((size_t)blockIdx.x * 12 + (threadIdx.x >> 6)) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STEM_STAMP(i) do { } while (0)
#define STEM_WSTAMP(i) do { } while (0)
#endif

__device__ __forceinline__ uint32_t split_pair(float v) {
  const _Float16 h = (_Float16)v;
  const _Float16 l = (_Float16)(v - (float)h);
  return (uint32_t)__builtin_bit_cast(unsigned short, h) | ((uint32_t)__builtin_bit_cast(unsigned short, l) << 16);
}

__global__ __launch_bounds__(512) void stem3d_f16x3_kernel(const StemArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  uint32_t* wl = lds;                                 // [WBYTES / 4]     split weights, staged once
  const int plane = PR * a.pwp;
  const int psize = KT * plane;                       // dwords per window buffer (even)
  uint32_t* patch0 = lds + WBYTES / 4;                // two window buffers: tile i+1 is fetched while
  uint32_t* patch1 = patch0 + psize;                  // tile i is multiplied

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kq = lane >> 4;
  // 22 pixel tiles over 8 waves: waves 0-5 take 3, waves 6-7 take 2; every wave covers all 64 channels,
  // so one gathered A fragment feeds 12 MFMAs (an f16 MFMA leaves only half its cycles for other issue)
  const int mt0 = wave < 6 ? wave * 3 : 18 + (wave - 6) * 2;
  const int mcnt = wave < 6 ? 3 : 2;
  const int npix = ROWS * a.Wo;
  constexpr int PPER = 20;                                 // 512 * 20 = 10240 >= 5*21*94 window pixels

  // Persistent workgroup (one per CU): tiles tile0, tile0 + grid, ...; a tile = 8 output rows of a frame.
  const int ntiles = a.n_tiles, G = gridDim.x;
  float pv[PPER];
  // window element e = tid + 512*i  ->  (frame tap ft, window row pr, column pc): tile independent
  // packed as ft << 16 | pr << 8 | pc (one register each); ft = 15 marks "beyond the window"
  uint32_t wcode[PPER];
#pragma unroll
  for (int i = 0; i < PPER; ++i) {
    const int e = tid + 512 * i;
    const int row = e / a.pwp;
    wcode[i] = ((uint32_t)(e < psize ? row / PR : 15) << 16) | ((uint32_t)(row % PR) << 8) | (uint32_t)(e - row * a.pwp);
  }
  const __amdgpu_buffer_rsrc_t xr = dlip_make_rsrc(a.x, a.x_bytes);
  auto fetch_window = [&](int tile) {   // issue the global loads of a tile's window: branch-free, no waits
    const int rt = tile % a.row_tiles, f = tile / a.row_tiles, t = f % a.T;
    const int hi0 = 2 * rt * ROWS - 3;
    const int clip = (f - t) * a.H * a.W;              // element offset of the clip (fits 32 bits: x < 2 GiB)
#pragma unroll
    for (int i = 0; i < PPER; ++i) {
      const int tt = t + (int)(wcode[i] >> 16) - 2, hi = hi0 + (int)((wcode[i] >> 8) & 255), wi = (int)(wcode[i] & 255) - 3;
      const bool ok = (unsigned)tt < (unsigned)a.T && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
      const uint32_t off = ok ? (uint32_t)((clip + (tt * a.H + hi) * a.W + wi) * 4) : DLIP_OOB_OFFSET;
      pv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, (int)off, 0, 0));
    }
  };
  float amax = 0.f;   // largest magnitude this lane splits (input pixels)
  auto store_window = [&](uint32_t* patch) {   // split into (hi, lo) pairs and write the LDS window
#pragma unroll
    for (int i = 0; i < PPER; ++i) {
      const int e = tid + 512 * i;
      if (e < psize) patch[e] = split_pair(pv[i]);
      amax = fmaxf(amax, fabsf(pv[i]));
    }
  };

  int tile = blockIdx.x;
  if (tile >= ntiles) return;
  {
    const uint4* src = reinterpret_cast<const uint4*>(a.w);
    uint4* dst = reinterpret_cast<uint4*>(wl);
    constexpr int WCHUNKS = WBYTES / 16, WPER = (WCHUNKS + 511) / 512;
    uint4 wv[WPER];
#pragma unroll
    for (int i = 0; i < WPER; ++i) {
      const int c = tid + 512 * i;
      wv[i] = c < WCHUNKS ? src[c] : uint4{0, 0, 0, 0};
    }
    fetch_window(tile);
#pragma unroll
    for (int i = 0; i < WPER; ++i) {
      const int c = tid + 512 * i;
      if (c < WCHUNKS) dst[c] = wv[i];
    }
    store_window(patch0);
  }

  int pixoff[MTW];   // dword offset of this lane's pixel (tap kw = 0, kernel row (0,0)) in the window
#pragma unroll
  for (int m = 0; m < MTW; ++m) {
    int p = (mt0 + (m < mcnt ? m : 0)) * 16 + li;
    if (p >= npix) p = 0;
    const int orow = p / a.Wo, ocol = p - orow * a.Wo;
    pixoff[m] = 2 * orow * a.pwp + 2 * ocol;
  }
  // B fragment byte offsets of this lane for its two 16-channel tiles (kernel row added per step)
  const int boff = li * WCH_BYTES;                      // channel tile nt adds nt * 16 * WCH_BYTES
  const char* wl8 = reinterpret_cast<const char*>(wl);
  const __amdgpu_buffer_rsrc_t yr = dlip_make_rsrc(a.y, a.y_bytes);
  float inv[4], bias[4], slope[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = j * 16 + li;
    inv[j] = 1.f / a.wscale[n];
    bias[j] = a.bias ? a.bias[n] : 0.f;
    slope[j] = a.slope ? a.slope[n] : 1.f;
  }
  __syncthreads();

  for (int it = 0; tile < ntiles; ++it) {
    const uint32_t* patch = (it & 1) ? patch1 : patch0;
    const int next = tile + G;
    STEM_STAMP(0);
    if (next < ntiles) fetch_window(next);          // in flight during this tile's MFMAs

    f32x4 acc[MTW][4];
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Software pipeline: the raw window dwords of pixel tile m+1 (and the weight fragments of step s+1)
    // are requested from LDS before the MFMAs of tile m are issued.
    auto a_ptr = [&](int s, int m) {
      const int rho = 4 * s + kq;                      // kernel row (kt, kh) of this lane quarter
      const int kt = rho / KH, kh = rho - kt * KH;
      const int koff = rho < KT * KH ? kt * plane + kh * a.pwp : 0;   // zero row: any valid address
      return reinterpret_cast<const u32x2*>(patch + koff + pixoff[m]);
    };
    u32x2 raw[2][4];
    f16x8 bfr[1][8];   // [channel tile nt: hi at 2*nt, lo at 2*nt + 1] (single set: 32 VGPRs)
    auto load_b = [&](int par, int s) {
      const int rho = 4 * s + kq;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        bfr[par][2 * nt] = *reinterpret_cast<const f16x8*>(wl8 + boff + nt * 16 * WCH_BYTES + rho * 16);
        bfr[par][2 * nt + 1] = *reinterpret_cast<const f16x8*>(wl8 + boff + nt * 16 * WCH_BYTES + WLO_OFF + rho * 16);
      }
    };
    {
      const u32x2* p0 = a_ptr(0, 0);
      raw[0][0] = p0[0]; raw[0][1] = p0[1]; raw[0][2] = p0[2]; raw[0][3] = p0[3];
    }
#pragma unroll 1
    for (int s2 = 0; s2 < STEPS + 1; s2 += 2) {        // two steps per trip so buffer parities are static
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        const int s = s2 + ss;
        if (s < STEPS) {
          load_b(0, s);
#pragma unroll
          for (int m = 0; m < MTW; ++m) {
            constexpr int dummy = 0; (void)dummy;
            const int cur = (ss * MTW + m) & 1;         // two steps per trip: parity restarts at 0 each trip
            // prefetch the next tile's raw dwords (next m, or m = 0 of the next step)
            if (m + 1 < MTW) {
              const u32x2* pn = a_ptr(s, m + 1);
              raw[cur ^ 1][0] = pn[0]; raw[cur ^ 1][1] = pn[1]; raw[cur ^ 1][2] = pn[2]; raw[cur ^ 1][3] = pn[3];
            } else if (s + 1 < STEPS) {
              const u32x2* pn = a_ptr(s + 1, 0);
              raw[cur ^ 1][0] = pn[0]; raw[cur ^ 1][1] = pn[1]; raw[cur ^ 1][2] = pn[2]; raw[cur ^ 1][3] = pn[3];
            }
            uint32_t hv[4], lv[4];
            // dword = hi | lo << 16.  v_perm_b32 byte selectors (src0 = bytes 7..4, src1 = bytes 3..0)
#pragma unroll
            for (int d = 0; d < 4; ++d) {
              hv[d] = __builtin_amdgcn_perm(raw[cur][d].y, raw[cur][d].x, 0x05040100u);
              lv[d] = __builtin_amdgcn_perm(raw[cur][d].y, raw[cur][d].x, 0x07060302u);
            }
            u32x4 hq = {hv[0], hv[1], hv[2], hv[3]}, lq = {lv[0], lv[1], lv[2], lv[3]};
            const f16x8 ah = __builtin_bit_cast(f16x8, hq), al = __builtin_bit_cast(f16x8, lq);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bfr[0][2 * nt], acc[m][nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bfr[0][2 * nt + 1], acc[m][nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bfr[0][2 * nt], acc[m][nt], 0, 0, 0);
          }
        }
      }
    }

    STEM_STAMP(1);
    const int rt = tile % a.row_tiles, f = tile / a.row_tiles, ho0 = rt * ROWS;
    const int rows_left = a.Ho - ho0;
    {
    // next tile's window -> the other LDS buffer (its last readers finished before the previous barrier)
    if (next < ntiles) store_window((it & 1) ? patch0 : patch1);

    // C/D map of the 16x16 MFMA: column (channel) = lane & 15, row (pixel) = (lane >> 4)*4 + e.
    const int ybase = (f * a.Ho + ho0) * a.Wo * 64;     // element offset (y < 2 GiB checked on the host)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = j * 16 + li;
#pragma unroll
      for (int m = 0; m < MTW; ++m) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int p = (mt0 + m) * 16 + kq * 4 + e;    // pixel index inside the tile = orow*Wo + ocol
          const bool ok = m < mcnt && p < npix && p < rows_left * a.Wo;
          float v = acc[m][j][e] * inv[j] + bias[j];
          v = v >= 0.f ? v : v * slope[j];
          const uint32_t off = ok ? (uint32_t)((ybase + p * 64 + n) * 4) : DLIP_OOB_OFFSET;
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), yr, (int)off, 0, 0);
        }
      }
    }
    }
    tile = next;
    STEM_STAMP(5);
    __syncthreads();   // window (it+1) complete and window (it) free before the next iteration
    STEM_STAMP(6);
  }
  dlip_report_range(amax, a.status);
}


// ---------------- fused stem + max pooling: the 12-wave kernel (round 2) ----------------
// The POOL variant above spends half of a tile's ~28 k cycles behind its MFMA loop (in-kernel stamps: loop 14.0 k,
// activation + barrier 5.2 k, window store 2.2 k, column pooling 1.0 k, two row passes through LDS 5.5 k; six
// barriers).  Two things make the pooling expensive there: 22 pixel tiles do not divide over 8 waves, and with
// row-major pixel tiles a pooling window's nine members sit in unrelated lanes and registers, so everything goes
// through an LDS row buffer, 32 channels at a time.  Here the PIXEL -> (wave, tile, lane) map is chosen for the pooling
// (it is free: every lane gathers its own pixel's taps from the window anyway):
//   * a tile's 8 stem rows x Wo (<= 44) columns are cut into 3 column STRIPS of 16 (columns 0-15, 14-29, 28-43: two
//     columns of overlap, 24 pixel tiles instead of 22 -- but 24 = 12 waves x 2, where 22 = 8 x 2.75 already cost 24);
//   * wave (strip, row pair rp) owns stem rows 2 rp and 2 rp + 1 of its strip, one pixel tile per row: lane & 15 =
//     column inside the strip.  The accumulators are TRANSPOSED (weights are the MFMA's first operand), so a lane holds
//     4 consecutive channels of ONE pixel per register quad;
//   * column pooling = two DPP row shifts per register (pooled column q = columns 2q-1, 2q, 2q+1: all inside the strip
//     that owns q -- strip 0: q 0-7, strip 1: q 8-14, strip 2: q 15-21; a shift past the strip's edge reads -inf, which is
//     exactly the padding at column -1);
//   * row pooling = v_max3 of registers: pooled row = (row above, row 2 rp, row 2 rp + 1); only the row above comes from
//     another wave -- every wave exports its lower row's column-pooled values (2 KB, even lanes) and imports its
//     neighbour's behind ONE barrier (the last row pair's export is the next tile's carried row: two alternating slots);
//   * the pooled values leave as 8-B (hi) + 8-B (lo) pieces of the split activation format straight from registers.
// The WINDOW no longer passes through registers either (20 loads, 20 splits and 20 ds_write_b32 per lane and tile in the
// kernel above): a pre-pass (stem_split_input_kernel, ~25 us at the bench's batch) writes the clip once as (hi, lo) fp16
// pairs with the window's row pitch and its 3 + 5 zero columns, so a window row is 24 contiguous 16-B chunks and a window
// plane arrives as 8 LDS-DMA pieces whose out-of-frame rows (offset out of range) are zeros.  The reduction walks the
// five frame planes in order, so the NEXT tile's plane p is fetched as soon as the last step that reads plane p is
// behind a barrier (after steps 1, 3, 5, 6 and 8): the fetch latency hides under the remaining steps.
// No row buffer; 3 waves per SIMD (<= 168 VGPRs).
constexpr int PWAVES = 12, PTHREADS = 64 * PWAVES;
constexpr int CARRY_SLOTS = 18, CARRY_B = 2048; // [strip][rp 0..2] 12 same-tile slots (9 used) + [parity][strip] 6 tile-to-tile slots

__device__ __forceinline__ float dpp_from_left(float v, float edge) {    // lane i <- lane i - 1 within its row of 16; lane 0 <- edge
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_from_right(float v, float edge) {   // lane i <- lane i + 1; lane 15 <- edge
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, v), 0x101, 0xf, 0xf, false));
}

// The clip -> xs [F, H, pwp] dwords (hi | lo << 16), columns 3 .. W + 2 = the pixels, the rest zero.
//   SRC 0: x fp32 [F, H, W], already normalised (the reference's [B,1,T,H,W] input, model.py:97);
//   SRC 1: uint8 frames [F, CH, Hs, Ws] as a loader hands them over (CH = 1 gray | 3 RGB = the north star's
//          [B,T,3,H,W]), centre-cropped to H x W at (oy, ox) and normalised HERE, (x / 255 - 0.421) / 0.165 on the BT.601
//          gray (dataloaders.py:11-22, preprocess.py:32-46): the fp32 clip (4 B per pixel written by an ingest kernel and
//          read back by this pre-pass, 4 B per pixel over PCIe) never exists.  Same arithmetic, same order as
//          ingest_rgb_kernel / crop_norm_kernel (dlip_common.h): bit-identical to ingest -> fp32 -> SRC 0.
//   lengths (ragged batches; NULL = every clip has T frames): frame t >= lengths[b] of clip b is PADDING and is written as zeros --
//          what pad_packed_collate puts there (models/video_models/dataset.py:123-139) and, for the frames t < lengths[b], exactly the
//          Conv3d's own zero padding behind the clip's last frame: their stem outputs equal the clip run alone at its own length
//          (train_fusion.py:346-348).  For uint8 frames the padding must be made HERE: a zero BYTE is not a zero of the normalised clip.
//   clip_params (SRC 1; NULL = one centre crop for the batch): int32 [B][4] = (oy, ox, flip, 0) per clip -- RandomCrop's origin and
//          HorizontalFlip's coin (models/video_models/preprocess.py:95-138, dataloaders.py:13-17: one draw per CLIP, all its frames alike).
template <int SRC>
__global__ __launch_bounds__(256) void stem_split_input_kernel(const void* __restrict__ xin, uint32_t* __restrict__ xs, int rows, int H, int W,
                                                               int pwp, int CH, int Hs, int Ws, int oy, int ox, int T,
                                                               const int32_t* __restrict__ lengths, const int32_t* __restrict__ clip_params,
                                                               DlipRange status, unsigned long long* span) {
  // (measurement only, one scalar test when no span scope is open: the stem's span starts with its pre-pass and ends with the pool kernel)
  if (span != nullptr && threadIdx.x == 0 && (blockIdx.x & 63) == 0) atomicMin(span, (unsigned long long)__builtin_amdgcn_s_memrealtime());
  const int cpr = pwp >> 2;                            // 16-B chunks per row
  const long long total = (long long)rows * cpr;
  float amax = 0.f;
  for (long long c = (long long)blockIdx.x * 256 + threadIdx.x; c < total; c += (long long)gridDim.x * 256) {
    const int row = (int)(c / cpr), ch = (int)(c - (long long)row * cpr);
    u32x4 o = {0u, 0u, 0u, 0u};
    int f = 0, h = 0, b = 0;
    if (SRC == 1 || lengths != nullptr) {
      f = row / H; h = row - f * H;
      b = f / T;
      if (lengths != nullptr && f - b * T >= lengths[b]) {   // a padding frame: zeros of the normalised clip
        *reinterpret_cast<u32x4*>(xs + (size_t)c * 4) = o;
        continue;
      }
    }
    if constexpr (SRC == 0) {
      const float* src = static_cast<const float*>(xin) + (size_t)row * W;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int wi = 4 * ch + e - 3;
        const float v = (unsigned)wi < (unsigned)W ? src[wi] : 0.f;
        amax = fmaxf(amax, fabsf(v));
        o[e] = split_pair(v);
      }
    } else {
      int coy = oy, cox = ox, flip = 0;
      if (clip_params != nullptr) {                    // per-clip crop origin (clamped into the frame) and flip
        coy = min(max(clip_params[4 * b], 0), Hs - H);
        cox = min(max(clip_params[4 * b + 1], 0), Ws - W);
        flip = clip_params[4 * b + 2];
      }
      const size_t plane = (size_t)Hs * Ws;
      const uint8_t* src = static_cast<const uint8_t*>(xin) + ((size_t)f * CH * Hs + (size_t)(coy + h)) * Ws + cox;
      // a chunk's four pixels are four consecutive bytes per colour plane: one (unaligned) dword load each where the chunk
      // lies inside the row, byte loads at the row's two ends (12 byte loads per 16-B store made this pre-pass 88 us
      // against the fp32 one's 22 at the bench's batch).  Flipped clips: pixel w of the crop is source column W - 1 - w, so
      // the chunk's four source bytes are again consecutive, in reverse order: the same dword load + a byte swap.
      const int w0 = 4 * ch - 3;
      uint32_t px[3] = {0u, 0u, 0u};
      if (w0 >= 0 && w0 + 3 < W) {
        const int s0 = flip ? W - 4 - w0 : w0;
#pragma unroll
        for (int c = 0; c < 3; ++c)
          if (c < CH) {
            __builtin_memcpy(&px[c], src + (size_t)c * plane + s0, 4);
            if (flip) px[c] = __builtin_bswap32(px[c]);
          }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if ((unsigned)(w0 + e) < (unsigned)W) {
            const int sc = flip ? W - 1 - (w0 + e) : w0 + e;
#pragma unroll
            for (int c = 0; c < 3; ++c)
              if (c < CH) px[c] |= (uint32_t)src[(size_t)c * plane + sc] << (8 * e);
          }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = 0.f;                                 // the convolution's zero padding (of the NORMALISED clip)
        if ((unsigned)(w0 + e) < (unsigned)W) {
          const float r = (float)((px[0] >> (8 * e)) & 255u);
          const float g = CH == 3 ? dlip_gray601(r, (float)((px[1] >> (8 * e)) & 255u), (float)((px[2] >> (8 * e)) & 255u)) : r;
          v = dlip_pixel_norm(g);
        }
        amax = fmaxf(amax, fabsf(v));
        o[e] = split_pair(v);
      }
    }
    *reinterpret_cast<u32x4*>(xs + (size_t)c * 4) = o;
  }
  dlip_report_range_block(amax, status);
}

__global__ __launch_bounds__(PTHREADS) void stem3d_pool_f16x3_kernel(const StemArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  uint32_t* wl = lds;                                 // [WBYTES / 4] split weights, staged once
  const int cpr = a.pwp >> 2;                         // 16-B chunks per window row
  const int np = (PR * cpr + 63) >> 6;                // LDS-DMA pieces per window plane (the plane is padded to whole pieces)
  const int plane = np * 256 + 32;                    // dwords per plane: == 32 (mod 64) like the row pitch, so lane quarters that
                                                      // straddle two planes (kernel rows 6 | 7, 20 | 21) still split the 64 banks
  uint32_t* patch = lds + WBYTES / 4;
  char* carry = reinterpret_cast<char*>(patch + KT * plane);
  float* tab = reinterpret_cast<float*>(carry + CARRY_SLOTS * CARRY_B);   // 1/wscale | bias | slope, 64 each
  // per carry slot: 1 + the iteration whose export it holds (explicit LDS address space: a generic volatile pointer would
  // poll with flat loads, which count in vmcnt and would drain the window pieces in flight)
  volatile __attribute__((address_space(3))) int* flags = (volatile __attribute__((address_space(3))) int*)(__attribute__((address_space(3))) char*)(tab + 192);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  const int strip = wave >> 2, rp = wave & 3;
  const int col = 14 * strip + li;                    // stem column of this lane's pixels
  const bool col_ok = col < a.Wo;
  const int ntiles = a.n_tiles, G = gridDim.x;

  // Window fetch: wave w < np moves piece w of a plane; this lane's chunk = (window row wrow, 16-B chunk wch) of the plane.
  const u32x4 xr = make_rsrc_words(a.x, a.x_bytes);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  const int wchunk = wave * 64 + lane;
  const int wrow = wchunk / cpr, wch = wchunk - wrow * cpr;
  auto fetch_plane = [&](int tile, int kt) {          // (call with wave < np only: wave-uniform); tile >= ntiles: zeros
    const int rt = tile % a.row_tiles, f = tile / a.row_tiles, t = f % a.T;
    const int tt = t + kt - 2, hi = 2 * rt * ROWS - 3 + wrow;
    const bool ok = tile < ntiles && (unsigned)tt < (unsigned)a.T && (unsigned)hi < (unsigned)a.H && wrow < PR;
    const uint32_t off = ok ? (uint32_t)((((f - t + tt) * a.H + hi) * cpr + wch) * 16) : DLIP_OOB_OFFSET;
    dma_piece(xr, off, lds0 + WBYTES + (kt * plane + wave * 256) * 4);
  };

  // Which frames: workgroup number wg_f takes frames wg_f, wg_f + G, ...  Round 5: wg_f is NOT blockIdx.x -- consecutive workgroups
  // sit on consecutive XCDs, and a frame's window is the five input planes f - 2 .. f + 2, so with frames dealt out in blockIdx
  // order every plane was pulled into FIVE different XCDs' L2 (each 4 MiB, not shared): rocprofv3 FETCH_SIZE 302 MB per launch for a
  // 62 MB split clip (profiles/r4/pmc_summary.json).  Here an XCD's workgroups take a CONTIGUOUS block of G / 8 frames of every round,
  // so a plane is fetched by one L2 (two at a block's edge): the ring kernel's remap of its work order, for the same reason.
  const int q8 = G >> 3, r8 = G & 7, xcd = blockIdx.x & 7;
  const int wg_f = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
  int tile = wg_f * a.row_tiles;                      // a workgroup walks whole frames, row tile after row tile
  if (tile >= ntiles) return;
  if (wave < np) {
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) fetch_plane(tile, kt);
  }
  {
    const uint4* src = reinterpret_cast<const uint4*>(a.w);
    uint4* dst = reinterpret_cast<uint4*>(wl);
    constexpr int WCHUNKS = WBYTES / 16, WPER = (WCHUNKS + PTHREADS - 1) / PTHREADS;
#pragma unroll
    for (int i = 0; i < WPER; ++i) {
      const int c = tid + PTHREADS * i;
      if (c < WCHUNKS) dst[c] = src[c];
    }
    if (tid < 64) {
      tab[tid] = 1.f / a.wscale[tid];                 // power of two: exact
      tab[64 + tid] = a.bias ? a.bias[tid] : 0.f;
      tab[128 + tid] = a.slope ? a.slope[tid] : 1.f;
      if (tid < CARRY_SLOTS) flags[tid] = 0;
    }
  }

  int pixoff[2];                                      // dword offset of this lane's pixel (tap 0 of kernel row (0,0)) in the window
#pragma unroll
  for (int m = 0; m < 2; ++m) pixoff[m] = 2 * (2 * rp + m) * a.pwp + 2 * (col_ok ? col : 0);
  int koff[STEPS];                                    // window offset of this lane quarter's kernel row (kt, kh) per step
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    const int rho = 4 * s + kq;
    const int kt = rho / KH, kh = rho - kt * KH;
    koff[s] = rho < KT * KH ? kt * plane + kh * a.pwp : (KT - 1) * plane;   // zero row (zero weights): finite data of a plane live in step 8
  }
  const int boff = li * WCH_BYTES;
  const char* wl8 = reinterpret_cast<const char*>(wl);
  const __amdgpu_buffer_rsrc_t yr = dlip_make_rsrc(a.y, a.y_bytes);
  const f32x4* tab4 = reinterpret_cast<const f32x4*>(tab);
  const float NEG = -__builtin_inff();
  float amax = 0.f;                                   // largest magnitude this lane stores
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const bool mono = __builtin_amdgcn_ballot_w64(tab[128 + lane] < 0.f) == 0;   // every slope >= 0 (workgroup-uniform)

  for (int it = 0; tile < ntiles; ++it) {
    const int rt = tile % a.row_tiles, f = tile / a.row_tiles, ho0 = rt * ROWS;
    const int next = rt + 1 < a.row_tiles ? tile + 1 : (f + G) * a.row_tiles;
    const bool duty = wave < np;                      // this wave moves one piece of every plane (wave-uniform)
    STEM_WSTAMP(0);

    f32x4 acc[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x2 raw[2][4];
    f16x8 bfr[8];                                     // channel tile nt: hi at 2 nt, lo at 2 nt + 1
    {
      const u32x2* p0 = reinterpret_cast<const u32x2*>(patch + koff[0] + pixoff[0]);
      raw[0][0] = p0[0]; raw[0][1] = p0[1]; raw[0][2] = p0[2]; raw[0][3] = p0[3];
    }
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
#ifdef DLIP_LAB
      if (s == STEPS - 1 && a.lab_skip_last_step) continue;
#endif
      const int rho16 = (4 * s + kq) * 16;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        bfr[2 * nt] = *reinterpret_cast<const f16x8*>(wl8 + boff + nt * 16 * WCH_BYTES + rho16);
        bfr[2 * nt + 1] = *reinterpret_cast<const f16x8*>(wl8 + boff + nt * 16 * WCH_BYTES + WLO_OFF + rho16);
      }
      // Frame plane p is read by the steps 4 s + kq in [7 p, 7 p + 6]: first by step 0, 1, 3, 5, 7 and last by step 1, 3, 5, 6, 8
      // for p = 0 .. 4.  The tile's ONLY barriers stand behind the reads of steps 1, 3, 5 and 6; behind each the plane that step
      // frees is refilled for the NEXT tile (out of range -> zeros when there is none), and behind step 1 -- every wave has left
      // the previous tile's reduction -- plane 4 of THIS tile as well.  What a barrier certifies (each duty wave waits for its
      // own pieces first; vmcnt counts in issue order, and every wave issues the same sequence per tile:
      //   [step 1: plane 4, next 0] [step 3: next 1] [step 5: next 2] [step 6: next 3] [epilogue: 8 stores]):
      //   step 1: this tile's planes 2, 3 (issued in the previous tile; only its 8 stores are younger)       -> vmcnt(8)
      //   step 6: this tile's plane 4 and the next tile's planes 0, 1 (only next 2 is younger)                  -> vmcnt(1)
      const bool frees = ((1 << 1 | 1 << 3 | 1 << 5 | 1 << 6) >> s) & 1;   // (folds once the step loop is unrolled)
      const int freed = s == 1 ? 0 : s == 3 ? 1 : s == 5 ? 2 : 3;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        if (m == 0) {                                 // the other pixel tile's taps of this step
          const u32x2* pn = reinterpret_cast<const u32x2*>(patch + koff[s] + pixoff[1]);
          raw[1][0] = pn[0]; raw[1][1] = pn[1]; raw[1][2] = pn[2]; raw[1][3] = pn[3];
        }
        uint32_t hv[4], lv[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          hv[d] = __builtin_amdgcn_perm(raw[m][d].y, raw[m][d].x, 0x05040100u);
          lv[d] = __builtin_amdgcn_perm(raw[m][d].y, raw[m][d].x, 0x07060302u);
        }
        u32x4 hq = {hv[0], hv[1], hv[2], hv[3]}, lq = {lv[0], lv[1], lv[2], lv[3]};
        const f16x8 ah = __builtin_bit_cast(f16x8, hq), al = __builtin_bit_cast(f16x8, lq);
        if (m == 1 && frees) {                        // every read of the freed plane has been consumed (hq, lq above): meet, refill
          if (s == 1) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
          else if (s == 6) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
          else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          if (duty) {
            if (s == 1) fetch_plane(tile, 4);
            fetch_plane(next, freed);
          }
        }
        if (m == 1 && s + 1 < STEPS) {                // the next step's first taps (a plane that is still live)
          const u32x2* pn = reinterpret_cast<const u32x2*>(patch + koff[s + 1 < STEPS ? s + 1 : s] + pixoff[0]);
          raw[0][0] = pn[0]; raw[0][1] = pn[1]; raw[0][2] = pn[2]; raw[0][3] = pn[3];
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bfr[2 * nt], al, acc[m][nt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bfr[2 * nt + 1], ah, acc[m][nt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bfr[2 * nt], ah, acc[m][nt], 0, 0, 0);
      }
    }
    STEM_WSTAMP(1);

    // ---- epilogue: activation, column pooling in registers (acc[m][j][e] = pixel (row 2 rp + m, col), channel
    // 16 j + 4 kq + e), one exported row, row pooling, split-format store.  MONO (every PReLU slope >= 0; 1 / wscale
    // is positive): the folded-BN affine and the activation are non-decreasing, so they commute with the maxima --
    // bit for bit, max only selects -- and are applied to the pooled values, a quarter of the registers. ----
    auto epilogue = [&](auto mono_c) {
      constexpr bool MONO = decltype(mono_c)::value;
      const bool inside = 14 * strip + 15 < a.Wo && ho0 + 2 * rp + 1 < a.Ho;   // the wave's 2 x 16 pixels are all inside the frame (wave-uniform)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 inv4, bi4, sl4;
        if constexpr (!MONO) { inv4 = tab4[4 * j + kq]; bi4 = tab4[16 + 4 * j + kq]; sl4 = tab4[32 + 4 * j + kq]; }
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const bool ok = col_ok && (ho0 + 2 * rp + m) < a.Ho;   // pixels outside the frame count as -inf
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = acc[m][j][e];
            if constexpr (!MONO) {
              v = v * inv4[e] + bi4[e];
              v = v >= 0.f ? v : v * sl4[e];
            }
            if (!inside) v = ok ? v : NEG;
            acc[m][j][e] = fmaxf(fmaxf(v, dpp_from_left(v, NEG)), dpp_from_right(v, NEG));
          }
        }
      }
      // the lower row's column maxima go to the wave below (the last row pair's: to the next tile's first wave)
      {
        const int slot = rp < 3 ? strip * 4 + rp : 12 + 3 * (it & 1) + strip;
        if ((li & 1) == 0) {
          f32x4* dst = reinterpret_cast<f32x4*>(carry + slot * CARRY_B) + kq * 8 + (li >> 1);
#pragma unroll
          for (int j = 0; j < 4; ++j) dst[j * 32] = acc[1][j];
        }
      }
      // (LDS executes one wave's instructions in order: whoever sees the flag sees the row)
      asm volatile("" ::: "memory");
      if (lane == 0) flags[rp < 3 ? strip * 4 + rp : 12 + 3 * (it & 1) + strip] = it + 1;
      STEM_WSTAMP(2);

      // row pooling: (row above, row 2 rp, row 2 rp + 1), then the split-format store of the even lanes
      const int slot = rp > 0 ? strip * 4 + rp - 1 : 12 + 3 * ((it + 1) & 1) + strip;
      const bool has_up = rp > 0 || rt > 0;           // (the frame's first row has nothing above it)
      // No workgroup barrier here: the row above is the ONLY thing this wave needs from another one, so it waits for that
      // wave's flag (its own strip's previous row pair in this tile, or the last row pair of the previous tile) and goes on
      // to its stores and the next tile while its SIMD's other waves are still multiplying.  The exporter never waits for
      // anybody, and the next tile's barriers stand between this import and the slot's next export.
      if (has_up) {
        const int expect = rp > 0 ? it + 1 : it;
        while (flags[slot] < expect) __builtin_amdgcn_s_sleep(4);
      }
      asm volatile("" ::: "memory");
      STEM_WSTAMP(3);
      STEM_WSTAMP(4);
      const f32x4* src = reinterpret_cast<const f32x4*>(carry + slot * CARRY_B) + kq * 8 + (li >> 1);
      const int pr = rt * (ROWS / 2) + rp;            // pooled row
      const bool lane_ok = (li & 1) == 0 && (strip == 0 || li >= 2) && col_ok && pr < a.Hp;
      const uint32_t base = (uint32_t)(((f * a.Hp + pr) * a.Wp + (col >> 1)) * 256 + kq * 8);
      typedef _Float16 h4 __attribute__((ext_vector_type(4)));
      float tmax = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 up = src[j * 32];
        f32x4 inv4, bi4, sl4;
        if constexpr (MONO) { inv4 = tab4[4 * j + kq]; bi4 = tab4[16 + 4 * j + kq]; sl4 = tab4[32 + 4 * j + kq]; }
        h4 hi, lo;
        float v4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = fmaxf(fmaxf(has_up ? up[e] : NEG, acc[0][j][e]), acc[1][j][e]);
          if constexpr (MONO) {
            v = v * inv4[e] + bi4[e];
            v = v >= 0.f ? v : v * sl4[e];
          }
          hi[e] = (_Float16)v;
          lo[e] = (_Float16)(v - (float)hi[e]);
          v4[e] = v;
        }
        tmax = fmaxf(fmaxf(tmax, fabsf(v4[0])), fabsf(v4[1]));   // (v_max3 with |.| operands; lanes that do not store are dropped below)
        tmax = fmaxf(fmaxf(tmax, fabsf(v4[2])), fabsf(v4[3]));
        // split activation format: pixel = 64 channels = two 128-B blocks of (32 hi | 32 lo) halves; channels 16 j + 4 kq ..
        const uint32_t off = lane_ok ? base + (uint32_t)((j >> 1) * 128 + (j & 1) * 32) : DLIP_OOB_OFFSET;
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hi), yr, (int)off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, lo), yr, (int)(lane_ok ? off + 64 : DLIP_OOB_OFFSET), 0, 0);
      }
      amax = lane_ok ? fmaxf(amax, tmax) : amax;
    };
    if (mono) epilogue(std::true_type{}); else epilogue(std::false_type{});
    tile = next;
    STEM_WSTAMP(5);
    STEM_WSTAMP(6);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the last tile's refills -- zeros -- have landed before the LDS is released)
  dlip_report_range(amax, a.status);
  dlip_span_exit(a.span);
}

}  // namespace

extern "C" int dlip_stem3d_bn_act_f16x3(const float* x, const void* w_split, const float* w_scale,
                                        const float* bias, const float* slope, float* y, int32_t B, int32_t T,
                                        int32_t H, int32_t W, int32_t K, dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && w_split && w_scale && y && B > 0 && T > 0 && H > 0 && W > 0);
  DLIP_CHECK_ARG(K == 64 && (H & 1) == 0 && (W & 1) == 0);
  StemArgs a;
  a.x = x; a.w = static_cast<const uint32_t*>(w_split); a.wscale = w_scale; a.bias = bias; a.slope = slope; a.y = y;
  a.T = T; a.H = H; a.W = W; a.Ho = H / 2; a.Wo = W / 2;
  a.row_tiles = (a.Ho + ROWS - 1) / ROWS;
  // window row pitch == 32 (mod 64) dwords: the two lane quarters a ds_read_b64 services together read
  // consecutive kernel rows, which then fall in opposite halves of the 64 LDS banks (conflict-free)
  a.pwp = ((W + 6 - 32 + 63) / 64) * 64 + 32;
  DLIP_CHECK_ARG(ROWS * a.Wo <= 22 * 16);   // 22 M tiles per workgroup: frames up to 88 pixels wide
  DLIP_CHECK_ARG(KT * PR * a.pwp <= 512 * 20);   // window pixels one staging pass covers
  const long long tiles = (long long)B * T * a.row_tiles;
  if (tiles > 0x7FFFFFFFll) return DLIP_ERANGE;
  a.n_tiles = (int)tiles;
  const long long xb = (long long)B * T * H * W * 4, yb = (long long)B * T * a.Ho * a.Wo * 64 * 4;
  if (xb > DLIP_MAX_BUFFER_BYTES || yb > DLIP_MAX_BUFFER_BYTES) return DLIP_ERANGE;
  a.x_bytes = (uint32_t)xb; a.y_bytes = (uint32_t)yb;
  a.Hp = a.Wp = 0; a.n_frames = B * T;
  a.span = nullptr;
  const long long grid = tiles < 256 ? tiles : 256;   // persistent: one workgroup per CU (154 KB of LDS each)
  const size_t ldsb = (size_t)WBYTES + 2 * (size_t)KT * PR * a.pwp * 4;
  auto kern = stem3d_f16x3_kernel;
  a.status = dlip_range_for(DLIP_ST_STEM);
  static DlipKernelState ks;   // the attribute is raised once per device and size (not on every launch)
  {
    const int e = ks.ensure_lds(reinterpret_cast<const void*>(kern), ldsb);
    if (e != DLIP_OK) return e;
  }
#ifdef DLIP_LAB
  a.stamps = nullptr;
  a.lab_skip_last_step = 0;
  if (getenv("DLIP_STAMP_PRINT")) {
    static unsigned long long* dbuf = nullptr;
    if (!dbuf) (void)hipMalloc(reinterpret_cast<void**>(&dbuf), 256 * 8 * 8);
    (void)hipMemset(dbuf, 0, 256 * 8 * 8);
    a.stamps = dbuf;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), ldsb, static_cast<hipStream_t>(stream), a);
    (void)hipDeviceSynchronize();
    unsigned long long h[256 * 8];
    (void)hipMemcpy(h, dbuf, sizeof(h), hipMemcpyDeviceToHost);
    double d[6] = {0, 0, 0, 0, 0, 0};
    int n = 0;
    for (int g = 0; g < (int)grid; ++g)
      if (h[g * 8 + 6]) { for (int j = 0; j < 6; ++j) d[j] += (double)(h[g * 8 + j + 1] - h[g * 8 + j]); ++n; }
    if (n) fprintf(stderr, "[stem stamps, mean of %d workgroups' second tile] mfma loop %.0f  act+B1 %.0f  window store %.0f  left/colpool %.0f  row passes %.0f  end barrier %.0f\n",
                   n, d[0] / n, d[1] / n, d[2] / n, d[3] / n, d[4] / n, d[5] / n);
    return dlip_launch_status();
  }
#endif
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), ldsb, static_cast<hipStream_t>(stream), a);
  return dlip_launch_status();
}

extern "C" int64_t dlip_stem3d_pool_workspace_bytes(int32_t B, int32_t T, int32_t H, int32_t W) {
  if (B <= 0 || T <= 0 || H <= 0 || W <= 0) return -1;
  const int64_t pwp = ((W + 6 - 32 + 63) / 64) * 64 + 32;
  return (int64_t)B * T * H * pwp * 4;
}

// src_kind 0: x = fp32 [B,T,H,W]; 1: x = uint8 [B,T,CH,Hs,Ws] cropped to H x W at (oy, ox)
static int stem_pool_launch(const void* x, int src_kind, int CH, int Hs, int Ws, int oy, int ox, const int32_t* clip_params,
                            const int32_t* lengths, void* x_split, const void* w_split,
                            const float* w_scale, const float* bias, const float* slope, float* y, int32_t B, int32_t T, int32_t H,
                            int32_t W, int32_t K, dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && x_split && w_split && w_scale && y && B > 0 && T > 0 && H > 0 && W > 0);
  DLIP_CHECK_ARG(K == 64 && (H & 1) == 0 && (W & 1) == 0 && (reinterpret_cast<uintptr_t>(x_split) & 15) == 0);
  if (src_kind == 1) DLIP_CHECK_ARG((CH == 1 || CH == 3) && oy >= 0 && ox >= 0 && oy + H <= Hs && ox + W <= Ws);
  StemArgs a;
  a.x = static_cast<const float*>(x_split); a.w = static_cast<const uint32_t*>(w_split); a.wscale = w_scale; a.bias = bias; a.slope = slope; a.y = y;
  a.T = T; a.H = H; a.W = W; a.Ho = H / 2; a.Wo = W / 2;
  a.Hp = (a.Ho - 1) / 2 + 1; a.Wp = (a.Wo - 1) / 2 + 1;
  a.row_tiles = (a.Ho + ROWS - 1) / ROWS;
  a.pwp = ((W + 6 - 32 + 63) / 64) * 64 + 32;
  DLIP_CHECK_ARG(a.Wo <= 44);                          // three column strips of 16 (stride 14): frames up to 88 pixels wide
  const int np = (PR * (a.pwp / 4) + 63) / 64;         // LDS-DMA pieces per window plane: one per wave
  DLIP_CHECK_ARG(np <= PWAVES);
  const long long frames = (long long)B * T, tiles = frames * a.row_tiles;
  if (tiles > 0x7FFFFFFFll) return DLIP_ERANGE;
  a.n_tiles = (int)tiles;
  a.n_frames = (int)frames;
  const long long xb = frames * H * a.pwp * 4, yb = frames * a.Hp * a.Wp * 64 * 4;
  if (xb > DLIP_MAX_BUFFER_BYTES || yb > DLIP_MAX_BUFFER_BYTES) return DLIP_ERANGE;
  a.x_bytes = (uint32_t)xb; a.y_bytes = (uint32_t)yb;
  // TWO evidence slots: the pre-pass reports the largest magnitude of the CLIP it splits, the kernel that of the pooled output it
  // stores.  (One slot for both until round 6: an output of ordinary size then vouched for a clip far below the line -- a float clip
  // with a gain of 2^-14 went unreported, tests/test_arith_gpu.py's clip soak.)
  const DlipRange in_status = dlip_range_for(DLIP_ST_STEM);
  a.status = dlip_range_for(DLIP_ST_STEM);
  a.span = dlip_span_next();
  hipStream_t st = static_cast<hipStream_t>(stream);
  {   // pre-pass: the clip as (hi, lo) pairs at the window's row pitch
    const long long chunks = frames * H * (a.pwp / 4);
    const unsigned pgrid = (unsigned)std::min<long long>((chunks + 255) / 256, 256 * 16);
    if (frames * H > 0x7FFFFFFFll) return DLIP_ERANGE;
    if (src_kind == 0)
      hipLaunchKernelGGL(stem_split_input_kernel<0>, dim3(pgrid), dim3(256), 0, st, x, static_cast<uint32_t*>(x_split), (int)(frames * H), H, W, a.pwp,
                         1, H, W, 0, 0, (int)T, lengths, static_cast<const int32_t*>(nullptr), in_status, a.span);
    else
      hipLaunchKernelGGL(stem_split_input_kernel<1>, dim3(pgrid), dim3(256), 0, st, x, static_cast<uint32_t*>(x_split), (int)(frames * H), H, W, a.pwp,
                         CH, Hs, Ws, oy, ox, (int)T, lengths, clip_params, in_status, a.span);
  }
  const long long grid = frames < 256 ? frames : 256;   // persistent: a workgroup walks whole frames
  const size_t ldsb = (size_t)WBYTES + (size_t)KT * (np * 256 + 32) * 4 + (size_t)CARRY_SLOTS * CARRY_B + 3 * 64 * 4 + 128;
  DLIP_CHECK_ARG(ldsb <= 160 * 1024);
  auto kern = stem3d_pool_f16x3_kernel;
  static DlipKernelState ks;   // the attribute is raised once per device and size (not on every launch)
  {
    const int e = ks.ensure_lds(reinterpret_cast<const void*>(kern), ldsb);
    if (e != DLIP_OK) return e;
  }
#ifdef DLIP_LAB
  a.stamps = nullptr;
  a.lab_skip_last_step = getenv("DLIP_STEM_SKIP_STEP8") ? 1 : 0;
  if (getenv("DLIP_STAMP_PRINT")) {
    static unsigned long long* dbuf = nullptr;
    constexpr size_t NS = 256 * 12 * 8;
    if (!dbuf) (void)hipMalloc(reinterpret_cast<void**>(&dbuf), NS * 8);
    (void)hipMemset(dbuf, 0, NS * 8);
    a.stamps = dbuf;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(PTHREADS), ldsb, st, a);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(NS);
    (void)hipMemcpy(h.data(), dbuf, NS * 8, hipMemcpyDeviceToHost);
    // per wave (mean over workgroups, third tile): cycles from the tile's start (wave 0's stamp 0) to each stamp
    for (int w = 0; w < 12; ++w) {
      double d[7] = {0, 0, 0, 0, 0, 0, 0};
      int n = 0;
      for (int g = 0; g < (int)grid; ++g) {
        const unsigned long long* r = &h[((size_t)g * 12 + w) * 8];
        const unsigned long long t0 = h[(size_t)g * 12 * 8];
        if (r[6] && t0) { for (int j = 0; j < 7; ++j) d[j] += (double)(long long)(r[j] - t0); ++n; }
      }
      if (n) fprintf(stderr, "[stem+pool wave %2d] start %6.0f  loop end %6.0f  exported %6.0f  past B1 %6.0f  p4 issued %6.0f  stores issued %6.0f  past end barrier %6.0f\n",
                     w, d[0] / n, d[1] / n, d[2] / n, d[3] / n, d[4] / n, d[5] / n, d[6] / n);
    }
    return dlip_launch_status();
  }
#endif
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(PTHREADS), ldsb, st, a);
  return dlip_launch_status();
}

extern "C" int dlip_stem3d_pool_f16x3(const float* x, const int32_t* lengths, void* x_split, const void* w_split, const float* w_scale,
                                      const float* bias, const float* slope, float* y, int32_t B, int32_t T, int32_t H, int32_t W,
                                      int32_t K, dlip_stream_t stream) {
  return stem_pool_launch(x, 0, 1, H, W, 0, 0, nullptr, lengths, x_split, w_split, w_scale, bias, slope, y, B, T, H, W, K, stream);
}

extern "C" int dlip_stem3d_pool_u8_f16x3(const uint8_t* frames, int32_t channels, int32_t Hs, int32_t Ws, int32_t oy, int32_t ox,
                                         const int32_t* clip_params, const int32_t* lengths, void* x_split, const void* w_split,
                                         const float* w_scale, const float* bias, const float* slope, float* y, int32_t B, int32_t T,
                                         int32_t H, int32_t W, int32_t K, dlip_stream_t stream) {
  return stem_pool_launch(frames, 1, channels, Hs, Ws, oy, ox, clip_params, lengths, x_split, w_split, w_scale, bias, slope, y, B, T,
                          H, W, K, stream);
}

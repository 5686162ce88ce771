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
// POOL variant (dlip_stem3d_pool_f16x3): MaxPool3d((1,3,3), stride (1,2,2), pad (0,1,1)) of
// model.py:85 is applied to the activations before they leave the CU, so the 4x larger pre-pool tensor
// (0.92 GB at the benchmark batch, written once and read once) never exists:
//   * a workgroup walks whole frames, row tile after row tile (8 stem rows -> 4 pooled rows), so the one
//     stem row a pooling window needs from the tile above is carried in LDS instead of recomputed;
//   * columns: a lane's 4 accumulator values are 4 consecutive pixels of one row, so the odd pooled
//     column is lane-local and the even one needs the pixel to the left (ds_bpermute from the lane
//     16 below, the previous M tile's registers, or -- at the 7 wave boundaries -- a 2 KB LDS exchange);
//   * rows: the column-pooled rows go to an LDS buffer (32 channels at a time), 3-row maxima are taken
//     from there and stored in the split activation format the trunk's LDS-DMA kernels read;
//   * LDS: weights 73 KB + ONE window 40 KB + row buffer 28 KB + carried rows 14 KB + exchange 2 KB; the
//     next window still travels in registers during the MFMAs and is written after the barrier that
//     ends them.
#include "dlip_common.h"

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
  int Hp, Wp;          // POOL: pooled output size
  int n_frames;        // POOL: B*T
  int32_t* status;     // range-status word (NULL: not reported)
};

constexpr int HPITCH = 40;   // floats per (row, pooled column) of the POOL row buffer: 32 channels + 8 (bank spread)

__device__ __forceinline__ uint32_t split_pair(float v) {
  const _Float16 h = (_Float16)v;
  const _Float16 l = (_Float16)(v - (float)h);
  return (uint32_t)__builtin_bit_cast(unsigned short, h) | ((uint32_t)__builtin_bit_cast(unsigned short, l) << 16);
}

template <bool POOL>
__global__ __launch_bounds__(512) void stem3d_f16x3_kernel(const StemArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  uint32_t* wl = lds;                                 // [WBYTES / 4]     split weights, staged once
  const int plane = PR * a.pwp;
  const int psize = KT * plane;                       // dwords per window buffer (even)
  uint32_t* patch0 = lds + WBYTES / 4;                // two window buffers: tile i+1 is fetched while
  uint32_t* patch1 = patch0 + psize;                  // tile i is multiplied (POOL: one buffer)
  // POOL only: column-pooled rows of the tile [8][Wp][HPITCH], the carried row [2 tiles][2 channel halves]
  // [Wp][HPITCH], and the wave-boundary exchange [8 waves][64 channels]
  float* hbuf = reinterpret_cast<float*>(patch0 + psize);
  float* halo = hbuf + ROWS * a.Wp * HPITCH;
  float* xch = halo + 4 * a.Wp * HPITCH;

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
  float amax = 0.f;   // largest magnitude this lane splits (input pixels; POOL: output activations too)
  auto store_window = [&](uint32_t* patch) {   // split into (hi, lo) pairs and write the LDS window
#pragma unroll
    for (int i = 0; i < PPER; ++i) {
      const int e = tid + 512 * i;
      if (e < psize) patch[e] = split_pair(pv[i]);
      amax = fmaxf(amax, fabsf(pv[i]));
    }
  };

  // plain: tiles blockIdx, blockIdx + G, ...; POOL: frames blockIdx, blockIdx + G, ..., each frame's row tiles in order
  int tile = POOL ? blockIdx.x * a.row_tiles : blockIdx.x;
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
    const uint32_t* patch = (!POOL && (it & 1)) ? patch1 : patch0;
    int next;
    if constexpr (POOL) {
      const int rtn = tile % a.row_tiles;
      next = rtn + 1 < a.row_tiles ? tile + 1 : (tile / a.row_tiles + G) * a.row_tiles;
    } else {
      next = tile + G;
    }
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

    const int rt = tile % a.row_tiles, f = tile / a.row_tiles, ho0 = rt * ROWS;
    const int rows_left = a.Ho - ho0;
    if constexpr (!POOL) {
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
    } else {
    // ---------------- fused 3x3 / stride-2 max pooling ----------------
    const float NEG = -__builtin_inff();
    // activation in place; pixels outside the frame count as -inf
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
      const int p0 = (mt0 + m) * 16 + kq * 4;
      const bool ok = m < mcnt && p0 < npix && p0 < rows_left * a.Wo;   // Wo % 4 == 0: a lane's 4 pixels share a row
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = acc[m][j][e] * inv[j] + bias[j];
          v = v >= 0.f ? v : v * slope[j];
          acc[m][j][e] = ok ? v : NEG;
        }
    }
    __syncthreads();   // B1: every wave is done reading the window; the exchange words are free
    if (next < ntiles) store_window(patch0);
    if (kq == 3) {
#pragma unroll
      for (int j = 0; j < 4; ++j) xch[wave * 64 + j * 16 + li] = acc[MTW - 1][j][3];   // waves 6, 7: tile 2 is all -inf, fixed below
    }
    if (kq == 3 && mcnt == 2) {
#pragma unroll
      for (int j = 0; j < 4; ++j) xch[wave * 64 + j * 16 + li] = acc[1][j][3];
    }
    // the pixel left of a lane's first pixel: lane - 16 (same M tile) or lane + 48 of the previous M tile
    float left[MTW][4];
    {
      float prev[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) prev[j] = NEG;
#pragma unroll
      for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float last = acc[m][j][3];   // (a bit_cast straight from a vector-element lvalue reads element 0)
          const float got = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((lane - 16) & 63) * 4, __builtin_bit_cast(int, last)));
          left[m][j] = kq > 0 ? got : prev[j];
          prev[j] = got;   // for kq == 0 lanes: element 3 of lane + 48 = the last pixel of this M tile
        }
    }
    __syncthreads();   // B2: exchange words visible
    if (kq == 0 && wave > 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) left[0][j] = xch[(wave - 1) * 64 + j * 16 + li];
    }
    // column pooling: even pooled column 2u = max(left, v0, v1), odd 2u+1 = max(v1, v2, v3)
    int orow[MTW], q0[MTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
      const int p0 = (mt0 + m) * 16 + kq * 4;
      orow[m] = p0 / a.Wo;
      const int ocol = p0 - orow[m] * a.Wo;
      q0[m] = ocol >> 1;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float l = ocol == 0 ? NEG : left[m][j];
        const float he = fmaxf(fmaxf(l, acc[m][j][0]), acc[m][j][1]);
        const float hod = fmaxf(fmaxf(acc[m][j][1], acc[m][j][2]), acc[m][j][3]);
        acc[m][j][0] = he;
        acc[m][j][1] = hod;
      }
    }
    const int rows_here = rows_left < ROWS ? rows_left : ROWS;
    float* halo_cur = halo + (it & 1) * 2 * a.Wp * HPITCH;          // written by the previous tile of this frame
    float* halo_nxt = halo + ((it + 1) & 1) * 2 * a.Wp * HPITCH;
    const int prow0 = rt * (ROWS / 2);
#pragma unroll
    for (int h = 0; h < 2; ++h) {            // 32 channels per pass
#pragma unroll
      for (int m = 0; m < MTW; ++m) {
        if (m < mcnt && orow[m] < rows_here) {
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            const int j = 2 * h + jj;
            float* dst = hbuf + (orow[m] * a.Wp + q0[m]) * HPITCH + jj * 16 + li;
            dst[0] = acc[m][j][0];
            dst[HPITCH] = acc[m][j][1];
            if (orow[m] == ROWS - 1) {       // the row the next tile's first pooling window reaches up to
              float* hd = halo_nxt + (h * a.Wp + q0[m]) * HPITCH + jj * 16 + li;
              hd[0] = acc[m][j][0];
              hd[HPITCH] = acc[m][j][1];
            }
          }
        }
      }
      __syncthreads();   // B3 / B5: the pass's column-pooled rows are in LDS
      const int items = (ROWS / 2) * a.Wp * 8;          // (pooled row, pooled column, 4-channel group)
      for (int i = tid; i < items; i += 512) {
        const int c4 = i & 7;
        const int q = (i >> 3) % a.Wp;
        const int prl = (i >> 3) / a.Wp;
        const int pr = prow0 + prl;
        if (pr < a.Hp) {
          const int r1 = 2 * prl;             // local stem rows 2 prl - 1, 2 prl, 2 prl + 1
          f32x4 v = *reinterpret_cast<const f32x4*>(hbuf + (r1 * a.Wp + q) * HPITCH + c4 * 4);
          if (r1 + 1 < rows_here) {
            const f32x4 w2 = *reinterpret_cast<const f32x4*>(hbuf + ((r1 + 1) * a.Wp + q) * HPITCH + c4 * 4);
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = fmaxf(v[c], w2[c]);
          }
          if (r1 > 0 || rt > 0) {
            const float* up = r1 > 0 ? hbuf + ((r1 - 1) * a.Wp + q) * HPITCH : halo_cur + (h * a.Wp + q) * HPITCH;
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(up + c4 * 4);
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = fmaxf(v[c], w0[c]);
          }
          typedef _Float16 h4 __attribute__((ext_vector_type(4)));
          h4 hi, lo;
#pragma unroll
          for (int c = 0; c < 4; ++c) { hi[c] = (_Float16)v[c]; lo[c] = (_Float16)(v[c] - (float)hi[c]); amax = fmaxf(amax, fabsf(v[c])); }
          // split activation format: pixel = 64 channels = two 128-B blocks of (32 hi | 32 lo) halves
          const uint32_t off = (uint32_t)(((f * a.Hp + pr) * a.Wp + q) * 256 + h * 128 + c4 * 8);
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hi), yr, (int)off, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, lo), yr, (int)off + 64, 0, 0);
        }
      }
      if (h == 0) __syncthreads();   // B4: pass 1 overwrites the row buffer
    }
    }
    tile = next;
    __syncthreads();   // window (it+1) complete and window (it) free before the next iteration
  }
  dlip_report_range(amax, a.status);
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
  const long long grid = tiles < 256 ? tiles : 256;   // persistent: one workgroup per CU (154 KB of LDS each)
  const size_t ldsb = (size_t)WBYTES + 2 * (size_t)KT * PR * a.pwp * 4;
  auto kern = stem3d_f16x3_kernel<false>;
  a.status = dlip_status_words() ? dlip_status_words() + DLIP_ST_STEM : nullptr;
  static size_t lds_set = 0;   // the attribute is raised once per size (not on every launch)
  if (ldsb > lds_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    if (e != hipSuccess) return (int)e;
    lds_set = ldsb;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), ldsb, static_cast<hipStream_t>(stream), a);
  return dlip_launch_status();
}

extern "C" int dlip_stem3d_pool_f16x3(const float* x, const void* w_split, const float* w_scale, const float* bias,
                                      const float* slope, float* y, int32_t B, int32_t T, int32_t H, int32_t W,
                                      int32_t K, dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && w_split && w_scale && y && B > 0 && T > 0 && H > 0 && W > 0);
  DLIP_CHECK_ARG(K == 64 && (H & 1) == 0 && (W & 7) == 0);   // Wo % 4 == 0: a lane's 4 pixels stay in one row
  StemArgs a;
  a.x = x; a.w = static_cast<const uint32_t*>(w_split); a.wscale = w_scale; a.bias = bias; a.slope = slope; a.y = y;
  a.T = T; a.H = H; a.W = W; a.Ho = H / 2; a.Wo = W / 2;
  a.Hp = (a.Ho - 1) / 2 + 1; a.Wp = (a.Wo - 1) / 2 + 1;
  a.row_tiles = (a.Ho + ROWS - 1) / ROWS;
  a.pwp = ((W + 6 - 32 + 63) / 64) * 64 + 32;
  DLIP_CHECK_ARG(ROWS * a.Wo <= 22 * 16);
  DLIP_CHECK_ARG(KT * PR * a.pwp <= 512 * 20);
  const long long frames = (long long)B * T, tiles = frames * a.row_tiles;
  if (tiles > 0x7FFFFFFFll) return DLIP_ERANGE;
  a.n_tiles = (int)tiles;
  a.n_frames = (int)frames;
  const long long xb = frames * H * W * 4, yb = frames * a.Hp * a.Wp * 64 * 4;
  if (xb > DLIP_MAX_BUFFER_BYTES || yb > DLIP_MAX_BUFFER_BYTES) return DLIP_ERANGE;
  a.x_bytes = (uint32_t)xb; a.y_bytes = (uint32_t)yb;
  const long long grid = frames < 256 ? frames : 256;   // persistent: a workgroup walks whole frames
  const size_t ldsb = (size_t)WBYTES + (size_t)KT * PR * a.pwp * 4 + (size_t)(ROWS + 4) * a.Wp * HPITCH * 4 + 8 * 64 * 4;
  DLIP_CHECK_ARG(ldsb <= 160 * 1024);
  auto kern = stem3d_f16x3_kernel<true>;
  a.status = dlip_status_words() ? dlip_status_words() + DLIP_ST_STEM : nullptr;
  static size_t lds_set = 0;   // the attribute is raised once per size (not on every launch)
  if (ldsb > lds_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    if (e != hipSuccess) return (int)e;
    lds_set = ldsb;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), ldsb, static_cast<hipStream_t>(stream), a);
  return dlip_launch_status();
}

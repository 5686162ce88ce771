// Shared between the fp32 and the split-fp16 implicit-GEMM convolution kernels (internal).
#pragma once
#include "dlip_common.h"
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int BK = 32;  // reduction slice: 32 channels of one filter tap

// Exact unsigned division of n < 2^31 by a launch constant: q = umulhi(n, mul) >> shift (mul = ceil(2^(32+shift) / d),
// shift = ceil(log2 d) - 1; d == 1 is mul == 0).  Two divisions per gathered row in every tile's set-up were a
// quarter of it as plain `/` (hipcc expands a 32-bit division into ~30 instructions).
struct FastDiv {
  uint32_t mul, shift;
};
inline FastDiv dlip_fastdiv(uint32_t d) {
  FastDiv f = {0u, 0u};
  if (d <= 1) return f;
  uint32_t s = 0;
  while ((1ull << (s + 1)) < d) ++s;            // s = ceil(log2 d) - 1
  f.shift = s;
  f.mul = (uint32_t)((((unsigned long long)1 << (32 + s)) + d - 1) / d);
  return f;
}
__device__ __forceinline__ int dlip_div(int n, const FastDiv f) {
  return f.mul == 0u ? n : (int)(__umulhi((uint32_t)n, f.mul) >> f.shift);
}

struct ConvArgs {
  const float* x;
  const float* w;
  const float* bias;
  const float* res;
  const float* slope;
  const float* pscale;
  const float* pshift;
  float* y;
  int H, W, C, K, R, S;
  int sh, sw, ph, pw, dh, dw;
  int Wo, HoWo;
  int ldx, ldy, ldr;
  int M;
  int tiles_n;
  int cchunks;  // ceil(C / 32)
  int nk;       // R * S * cchunks
  int rsc;      // R * S * C  (weight row length)
  uint32_t x_bytes, w_bytes, r_bytes, y_bytes;
  const float* wscale;  // split-fp16 path: per-output-channel power-of-two weight scale (NULL otherwise)
  int Cw;               // channels per tap in the PACKED weights (C, or C rounded up to 32)
  // ---- LDS-DMA kernel only (conv_igemm_f16x3_dma.hip) ----
  // second reduction source (dlip_conv2_nhwc_f16x3): after the nk1 = R S cchunks slices of x, nk2 slices of a
  // 1x1 strided convolution over x2 [N,H2,W2,C2] (same N, same output grid), weights behind the taps in each row
  const float* x2;
  uint32_t x2_bytes;
  int H2, W2, ldx2, s2h, s2w, nk2;
  // pooled epilogue (dlip_conv_pool_f16x3): per tile row band and row-group segment, column sums of y and y^2
  double* pool;
  int pool_group;       // rows per group (>= the tile's BM)
  DlipLen pool_len;     // ragged batches: valid rows of each group, counted from the group's first row (len == NULL: all)
  DlipRange status;     // range reporting of a split-format output (dlip_common.h)
  FastDiv div_howo, div_wo;
  int n_inner;          // LDS-DMA kernel: tile order with the output-channel block inner
  // ---- the LDS-DMA kernel's many-tap instances only (a weight gradient run as a convolution, dlip_wgrad_conv_f16x3): where a
  // 32-channel slice and a filter tap sit.  Pixel-major operands (the default): slice c of a pixel / tap is 128 c bytes further,
  // tap t of a filter row Cw * 4 t; SLICE-major operands ([image][slice][H][W][32]): a slice is a whole image plane further and
  // consecutive taps are adjacent 128-byte lines.
  int Hs;               // rows between two images of x (H, or slices * H)
  int cs_x, wt, cs_w;   // bytes: slice stride of x, tap stride and slice stride of the filter rows
  // ---- the window kernel (conv_win_f16x3.hip): NULL, or this launch's {first start, last end} pair in 100 MHz ticks (dlip_span_scope_*;
  // the ring and rows kernels carry theirs in their schedule blocks)
  unsigned long long* span;
  // ---- a weight gradient run as a convolution (dlip_wgrad_conv_f16x3 with R, S given): the output element (m = (c, r', s'), k) goes
  // to dw[k][c][r'][s'] -- the REFERENCE layout [K, C, R, S] -- for r' < wg_R, s' < wg_S and nowhere otherwise; wg_R == 0: plain rows
  int wg_R, wg_S;
  // ---- dlip_conv_nhwc_stats_f16x3 (the rows kernel's fp32 epilogue): NULL, or [chunks][K][2] fp64 -- per half tile (a wave row:
  // chunk 2 tile_m + wm) the column sums {sum y, sum y^2} of the rows it wrote: the train-mode BatchNorm's statistics without a pass
  // of their own over y
  double* stats;
};


// Validates a dlip_conv_desc and fills the kernel argument block.  `Cw` = channels per filter tap in
// the packed weight tensor (== d->C for the fp32 layout).
inline int dlip_fill_conv_args(const dlip_conv_desc* d, const float* x, const float* w, const float* bias,
                               const float* residual, const float* slope, const float* post_scale,
                               const float* post_shift, float* y, int Cw, ConvArgs* out, int max_taps = 32) {
  DLIP_CHECK_ARG(d && x && w && y);
  DLIP_CHECK_ARG(d->N > 0 && d->H > 0 && d->W > 0 && d->C > 0 && d->K > 0 && d->R > 0 && d->S > 0);
  DLIP_CHECK_ARG(d->stride_h > 0 && d->stride_w > 0 && d->dil_h > 0 && d->dil_w > 0 && d->pad_h >= 0 && d->pad_w >= 0);
  DLIP_CHECK_ARG((d->C & 3) == 0 && (d->ldx & 3) == 0 && d->ldx >= d->C && d->ldy >= d->K && Cw >= d->C);
  DLIP_CHECK_ARG((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(w) & 15) == 0);
  DLIP_CHECK_ARG((post_scale == nullptr) == (post_shift == nullptr));
  DLIP_CHECK_ARG(residual == nullptr || d->ldr >= d->K);
  const int Ho = (d->H + 2 * d->pad_h - d->dil_h * (d->R - 1) - 1) / d->stride_h + 1;
  const int Wo = (d->W + 2 * d->pad_w - d->dil_w * (d->S - 1) - 1) / d->stride_w + 1;
  DLIP_CHECK_ARG(Ho == d->Ho && Wo == d->Wo && Ho > 0 && Wo > 0);
  DLIP_CHECK_ARG(d->R * d->S <= max_taps);  // per-row tap-validity mask is 32 bits (the LDS-DMA kernel has a mask-free variant)

  const long long in_pix = (long long)d->N * d->H * d->W;
  const long long x_bytes = ((in_pix - 1) * d->ldx + d->C) * 4;
  const long long w_bytes = (long long)d->K * d->R * d->S * Cw * 4;
  const long long M = (long long)d->N * Ho * Wo;
  const long long y_bytes = ((M - 1) * d->ldy + d->K) * 4;
  const long long r_bytes = residual ? ((M - 1) * d->ldr + d->K) * 4 : 0;
  if (x_bytes > DLIP_MAX_BUFFER_BYTES || w_bytes > DLIP_MAX_BUFFER_BYTES || y_bytes > DLIP_MAX_BUFFER_BYTES ||
      r_bytes > DLIP_MAX_BUFFER_BYTES || M > 0x7FFFFFFFll)
    return DLIP_ERANGE;

  ConvArgs& a = *out;
  a.x = x; a.w = w; a.bias = bias; a.res = residual; a.slope = slope;
  a.pscale = post_scale; a.pshift = post_shift; a.y = y;
  a.H = d->H; a.W = d->W; a.C = d->C; a.K = d->K; a.R = d->R; a.S = d->S;
  a.sh = d->stride_h; a.sw = d->stride_w; a.ph = d->pad_h; a.pw = d->pad_w; a.dh = d->dil_h; a.dw = d->dil_w;
  a.Wo = Wo; a.HoWo = Ho * Wo;
  a.ldx = d->ldx; a.ldy = d->ldy; a.ldr = d->ldr;
  a.M = (int)M;
  a.tiles_n = 0;
  a.cchunks = (d->C + BK - 1) / BK;
  a.nk = d->R * d->S * a.cchunks;
  a.rsc = d->R * d->S * Cw;
  a.x_bytes = (uint32_t)x_bytes; a.w_bytes = (uint32_t)w_bytes;
  a.r_bytes = (uint32_t)r_bytes; a.y_bytes = (uint32_t)y_bytes;
  a.wscale = nullptr;
  a.Cw = Cw;
  a.x2 = nullptr; a.x2_bytes = 0; a.H2 = a.W2 = a.ldx2 = a.s2h = a.s2w = a.nk2 = 0;
  a.pool = nullptr; a.pool_group = 0; a.pool_len = DlipLen{};
  a.status = DlipRange{};
  a.div_howo = dlip_fastdiv((uint32_t)a.HoWo);
  a.div_wo = dlip_fastdiv((uint32_t)a.Wo);
  a.n_inner = 0;
  a.Hs = d->H; a.cs_x = 128; a.wt = Cw * 4; a.cs_w = 128;
  a.span = nullptr;
  a.wg_R = a.wg_S = 0;
  a.stats = nullptr;
  return DLIP_OK;
}

// Tile menu / picker shared by both kernels (see conv_igemm.hip for the rationale).
struct TileCfg { int bm, bn; };
constexpr int NUM_CFG = 5;
const TileCfg kCfg[NUM_CFG] = {{128, 128}, {128, 64}, {64, 64}, {64, 128}, {96, 128}};
// Measured relative efficiency of a busy CU per tile (tools/bench_layers.py, MI355X, B = 64).
const float kEffF32[NUM_CFG] = {0.90f, 0.97f, 0.93f, 1.00f, 0.97f};
const float kEffF16x3[NUM_CFG] = {0.95f, 0.93f, 0.80f, 1.00f, 0.90f};

inline int pick_tile(long long M, int K, const float* eff = kEffF32) {
  if (const int v = dlip_dbg_value[DLIP_DBG_CONV_TILE]; v >= 0 && v < NUM_CFG) return v;   // dlip_debug_set (A/B runs, tests)
  int best = 0;
  double best_cost = 1e300;
  for (int i = 0; i < NUM_CFG; ++i) {
    const TileCfg& c = kCfg[i];
    const long long tiles = ((M + c.bm - 1) / c.bm) * ((K + c.bn - 1) / c.bn);
    const long long rounds = (tiles + 255) / 256;  // one tile per CU per round
    const double cost = (double)rounds * c.bm * c.bn / eff[i];
    if (cost < best_cost * 0.999) { best_cost = cost; best = i; }
  }
  return best;
}

}  // namespace

// Train-mode kernels of the lip-clip ENCODER (full train_video.py training, i.e.
// what torch.autograd does for Lipreading.forward under model.train(): train_video.py:108-169 over
// models/video_models/model.py:80-105, resnet.py:28-127, tcn.py:28-116).  The convolutions themselves
// (forward, data gradient, weight gradient) run on the implicit-GEMM kernels; this file holds what sits
// around them:
//   * tap gather: the rows one filter tap reads, as a dense [J, C] matrix (zeros for the halo) -- the
//     operand of the per-tap weight-gradient GEMM of a padded / strided convolution;
//   * zero insertion: the stride-s data gradient as a stride-1 convolution over the up-sampled dY;
//   * PReLU with a per-channel learnable slope, forward and backward (resnet.py:52,66; tcn.py:47,105);
//   * MaxPool3d((1,3,3),(1,2,2),(0,1,1)) backward (model.py:85), gather form: every input pixel looks at the
//     <= 4 windows that contain it and takes dY from those whose FIRST maximum (row-major scan, the tie
//     rule of the forward kernel and of ATen) it is -- deterministic, no atomics;
//   * row broadcast: backward of AdaptiveAvgPool2d(1) (resnet.py:83) and of the masked temporal mean
//     (model.py:16-17);
//   * im2col of the 5x7x7 stride-(1,2,2) stem (C_in = 1) for its weight gradient;
//   * mask multiply (Dropout forward / backward, tcn.py:80,85).
#include "dlip_common.h"

namespace {

inline unsigned grid_for(long long n, int cap = 1 << 16) {
  long long g = (n + 255) / 256;
  return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

__global__ __launch_bounds__(256) void tap_gather_kernel(const float* __restrict__ x, float* __restrict__ out, int H, int W,
                                                         int ldx, int C4, int Ho, int Wo, int sh, int sw, int oh, int ow,
                                                         int ldo, long long n4) {
  // oh / ow: input offset of the tap = r * dil_h - pad_h, s * dil_w - pad_w
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    const long long j = i / C4;
    const int wo = (int)(j % Wo);
    const long long t = j / Wo;
    const int ho = (int)(t % Ho);
    const long long n = t / Ho;
    const int hi = ho * sh + oh, wi = wo * sw + ow;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W)
      v = *reinterpret_cast<const f32x4*>(x + ((n * H + hi) * W + wi) * ldx + c4 * 4);
    *reinterpret_cast<f32x4*>(out + j * ldo + c4 * 4) = v;
  }
}

// One operand of a weight-gradient GEMM in ONE pass (replaces tap gather -> transpose -> split: three round trips of a matrix
// RS times the activation).  dW[(tap, c), k] = sum_j x[pixel(j) + tap][c] * dy[j][k] reduces over the J output positions, and
// the LDS-DMA kernel wants both operands reduction-major in the split format (per row, blocks of 32 consecutive j as 32 hi
// halves | 32 lo halves).  This kernel reads the NHWC tensor where it lies and writes that image directly:
//   out[(tap * C + c) * J32 + j]  <-  scale * x[n, ho * sh + r * dh - ph, wo * sw + s * dw - pw, c]   (zero outside the image, j >= J)
// One workgroup = 32 positions x 32 channels of one tap through a 32 x 33 LDS tile: coalesced 128-B reads along c, whole 128-B
// split blocks written along j.  With R = S = 1 and no padding it is the dy operand (scale = the gradient's power-of-two lift).
__global__ __launch_bounds__(256) void wgrad_operand_kernel(const float* __restrict__ x, float* __restrict__ out, int H, int W, int ldx,
                                                            int C, int Ho, int Wo, int sh, int sw, int R, int S, int dh, int dw, int ph, int pw,
                                                            int J, long long ldo, const float* __restrict__ scale, DlipRange status) {
  __shared__ float tile[32][33];
  const int j0 = blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  const float sc = scale ? scale[0] : 1.f;
  // this thread's four positions j0 + ty + {0, 8, 16, 24}: pixel origin (tap 0, 0) and image row / column, once for all taps
  int base[4], hi0[4], wi0[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int j = j0 + ty + 8 * q;
    const int wo = j % Wo, t = j / Wo;
    const int ho = t % Ho, n = t / Ho;
    hi0[q] = j < J ? ho * sh - ph : -(1 << 28);       // beyond J: every tap reads "outside the image" -> zeros
    wi0[q] = wo * sw - pw;
    base[q] = n * H;
  }
  const int cr = threadIdx.x >> 3, pq = threadIdx.x & 7;    // write phase: channel row, 16-B piece of its 128-B block
  const int jb = (pq & 3) * 8;
  float amax = 0.f;
  for (int tap = 0; tap < R * S; ++tap) {
    const int r = tap / S, s_ = tap - r * S;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int hi = hi0[q] + r * dh, wi = wi0[q] + s_ * dw;
      float v = 0.f;
      if ((unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W && c0 + tx < C)
        v = x[((long long)(base[q] + hi) * W + wi) * ldx + c0 + tx] * sc;
      tile[ty + 8 * q][tx] = v;
    }
    __syncthreads();
    // a row's 128-B block = 8 sixteen-byte pieces (4 of hi halves, 4 of lo halves, 8 positions each): one 16-B store per thread
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    h8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = tile[jb + e][cr];
      const _Float16 hi = (_Float16)v;
      o[e] = pq < 4 ? hi : (_Float16)(v - (float)hi);
      amax = fmaxf(amax, fabsf(v));
    }
    if (c0 + cr < C)
      *reinterpret_cast<h8*>(reinterpret_cast<_Float16*>(out + ((long long)tap * C + c0 + cr) * ldo + j0) + (pq < 4 ? 0 : 32) + jb) = o;
    __syncthreads();
  }
  dlip_report_range(amax, status);
}

// (round 5) A train-mode BatchNorm + (Leaky | P)ReLU applied ON LOAD by the operand producers below: the convolution that FOLLOWS a
// conv -> BatchNorm -> activation (tdnn.py:35-43, resnet.py:51-53) needs its input only as split images, so the activated tensor is
// never stored -- the producer reads the previous convolution's raw output z and forms lrelu((z - mean) invstd gamma + beta, slope)
// per value (bn_fwd_apply_kernel's expression: the same bits).  Positions outside the image / beyond N stay exact zeros.
struct BnOnLoad {
  const float* mean;        // nullptr: plain load
  const float* invstd;
  const float* gamma;
  const float* beta;
  const float* slope_vec;   // nullptr: the scalar `slope` for every channel
  float slope;
  // (ABI 47) the BACKWARD on load: the producer reads dy (its first source) and z (`z`) and forms the BatchNorm's input gradient
  // dz = gamma invstd (g - dbeta / M - xhat dgamma / M) per value -- bn_bwd_apply_kernel's expression, the same bits -- so dz is never stored
  const float* z = nullptr;
  const float* dgamma = nullptr;
  const float* dbeta = nullptr;
  float invM = 0.f;
  int act_first = 0;
  // (ABI 49) ... with dy itself formed on load from a MeanStdPooling's pooled statistics and their gradient (encoder_train_ops.hip: MsSrc): then
  // the first source is not read at all
  const float* ms_coef = nullptr;    // [B, 2 C] (A | K): dy = A + K y (dlip_meanstd_bwd_coef_f32)
  int ms_T = 1;
  int C = 0;                         // channels of the tensor (the last CT-wide block of a launch may be ragged)
};
__device__ __forceinline__ float bnl_lrelu(float v, float slope) { return v >= 0.f ? v : v * slope; }
template <int CT>
__device__ __forceinline__ void bn_bwd_on_load_stage(const BnOnLoad& b, int c0, float* tab) {      // tab [6][CT]: mean invstd gamma beta dgamma dbeta
  for (int i = threadIdx.x; i < 6 * (CT / 4); i += 256) {
    const int w = i / (CT / 4), c = (i - w * (CT / 4)) * 4;
    const float* src = w == 0 ? b.mean : w == 1 ? b.invstd : w == 2 ? b.gamma : w == 3 ? b.beta : w == 4 ? b.dgamma : b.dbeta;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (c0 + c < b.C) v = *reinterpret_cast<const f32x4*>(src + c0 + c);      // (C % 4 == 0: a quad is all inside or all outside)
    *reinterpret_cast<f32x4*>(tab + w * CT + c) = v;
  }
  __syncthreads();
}
// dy[row, c ..] = A + K ya from the pooling's coefficients (encoder_train_ops.hip: ms_grad) at the activated values ya
__device__ __forceinline__ f32x4 bnl_ms_grad(const BnOnLoad& b, int row, int c, const f32x4 ya) {
  const int u = row / b.ms_T;
  const float* cb = b.ms_coef + (long long)u * 2 * b.C + c;
  const f32x4 A = *reinterpret_cast<const f32x4*>(cb), K = *reinterpret_cast<const f32x4*>(cb + b.C);
  f32x4 o;
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = fmaf(K[k], ya[k], A[k]);
  return o;
}
template <int CT>
__device__ __forceinline__ f32x4 bn_bwd_on_load(const float* tab, const BnOnLoad& b, f32x4 gv, f32x4 xv, int c, int row = 0, int cabs = 0) {
  const f32x4 mu = *reinterpret_cast<const f32x4*>(tab + c), is = *reinterpret_cast<const f32x4*>(tab + CT + c);
  const f32x4 ga = *reinterpret_cast<const f32x4*>(tab + 2 * CT + c), be = *reinterpret_cast<const f32x4*>(tab + 3 * CT + c);
  const f32x4 dg = *reinterpret_cast<const f32x4*>(tab + 4 * CT + c), db = *reinterpret_cast<const f32x4*>(tab + 5 * CT + c);
  if (b.ms_coef != nullptr) {          // (launch-uniform; act_first == 0)
    f32x4 ya;
#pragma unroll
    for (int k = 0; k < 4; ++k) ya[k] = bnl_lrelu((xv[k] - mu[k]) * is[k] * ga[k] + be[k], b.slope);
    gv = bnl_ms_grad(b, row, cabs, ya);
  }
  f32x4 o;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float a = b.act_first ? bnl_lrelu(xv[k], b.slope) : xv[k];
    const float xh = (a - mu[k]) * is[k];
    float g = gv[k];
    if (!b.act_first) g *= (xh * ga[k] + be[k]) >= 0.f ? 1.f : b.slope;
    float d = ga[k] * is[k] * (g - db[k] * b.invM - xh * dg[k] * b.invM);
    if (b.act_first) d *= xv[k] >= 0.f ? 1.f : b.slope;
    o[k] = d;
  }
  return o;
}
// The workgroup's CT channels of the five parameter vectors, staged in LDS once (tab [5][CT]): read from memory per loaded value
// they were five vector-memory instructions beside every 16-B data load (the producers went from 80 to 95 - 104 us per TDNN layer).
template <int CT>
__device__ __forceinline__ void bn_on_load_stage(const BnOnLoad& b, int c0, float* tab) {
  for (int i = threadIdx.x; i < 5 * (CT / 4); i += 256) {
    const int w = i / (CT / 4), c = (i - w * (CT / 4)) * 4;
    const float* src = w == 0 ? b.mean : w == 1 ? b.invstd : w == 2 ? b.gamma : w == 3 ? b.beta : b.slope_vec;
    f32x4 v = {b.slope, b.slope, b.slope, b.slope};
    if (src) v = *reinterpret_cast<const f32x4*>(src + c0 + c);
    *reinterpret_cast<f32x4*>(tab + w * CT + c) = v;
  }
  __syncthreads();
}
template <int CT>
__device__ __forceinline__ f32x4 bn_on_load(const float* tab, f32x4 v, int c) {   // c: channel within the workgroup's CT
  const f32x4 mu = *reinterpret_cast<const f32x4*>(tab + c), is = *reinterpret_cast<const f32x4*>(tab + CT + c);
  const f32x4 ga = *reinterpret_cast<const f32x4*>(tab + 2 * CT + c), be = *reinterpret_cast<const f32x4*>(tab + 3 * CT + c);
  const f32x4 sl = *reinterpret_cast<const f32x4*>(tab + 4 * CT + c);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float t = (v[k] - mu[k]) * is[k] * ga[k] + be[k];
    v[k] = t >= 0.f ? t : t * sl[k];
  }
  return v;
}

// The same operand from WIDER tiles (round 4; see wgrad_chwn_wide_kernel): 32 positions x CT channels (64 | 128) of one tap, read
// as 16-B quads and written as CT / 32 sixteen-byte pieces per thread.  C % CT == 0, ldx % 4 == 0, x 16-byte aligned.
template <int CT, int AFF = 0>      // AFF: 0 plain, 1 BatchNorm + activation on load, 2 BatchNorm BACKWARD on load (x = dy, bn.z = z)
__global__ __launch_bounds__(256) void wgrad_operand_wide_kernel(const float* __restrict__ x, float* __restrict__ out, int H, int W, int ldx,
                                                                 int C, int Ho, int Wo, int sh, int sw, int R, int S, int dh, int dw, int ph,
                                                                 int pw, int J, long long ldo, const float* __restrict__ scale,
                                                                 DlipRange status, float* __restrict__ nhwc_out = nullptr,
                                                                 const BnOnLoad bn = BnOnLoad{}, int ld_nhwc = 0) {
  // (ABI 49, AFF == 2 only) C need not be a multiple of CT: the last channel block is ragged -- quads beyond C load as zeros, rows beyond C
  // of the image are not written, and the split NHWC copy has the row pitch ld_nhwc >= C (a multiple of 32: its padding channels are zeros)
  constexpr int PITCH = CT + 4, Q = CT / 32;
  __shared__ __attribute__((aligned(16))) float tile[32 * PITCH];
  __shared__ __attribute__((aligned(16))) float bn_tab[AFF == 2 ? 6 * CT : AFF == 1 ? 5 * CT : 4];
  const int j0 = blockIdx.x * 32, c0 = blockIdx.y * CT;
  const int ldn = ld_nhwc > 0 ? ld_nhwc : C;
  const float sc = scale ? scale[0] : 1.f;
  if constexpr (AFF == 1) bn_on_load_stage<CT>(bn, c0, bn_tab);
  if constexpr (AFF == 2) bn_bwd_on_load_stage<CT>(bn, c0, bn_tab);
  // this thread's Q quads: position row (i / (CT / 4)) -- pixel origin of tap (0, 0), once for all taps -- and channel quad
  int base[Q], hi0[Q], wi0[Q], col[Q], row_[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int i = threadIdx.x + 256 * q, row = i / (CT / 4);
    const int j = j0 + row;
    const int wo = j % Wo, t = j / Wo;
    const int ho = t % Ho, n = t / Ho;
    hi0[q] = j < J ? ho * sh - ph : -(1 << 28);        // beyond J: every tap reads "outside the image" -> zeros
    wi0[q] = wo * sw - pw;
    base[q] = n * H;
    col[q] = (i - row * (CT / 4)) * 4;
    row_[q] = row;
  }
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  float amax = 0.f;
  for (int tap = 0; tap < R * S; ++tap) {
    const int r = tap / S, s_ = tap - r * S;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int hi = hi0[q] + r * dh, wi = wi0[q] + s_ * dw;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if ((unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W && (AFF != 2 || c0 + col[q] < C)) {
        const long long off = ((long long)(base[q] + hi) * W + wi) * ldx + c0 + col[q];
        if (AFF != 2 || bn.ms_coef == nullptr) v = *reinterpret_cast<const f32x4*>(x + off);
        if constexpr (AFF == 1) v = bn_on_load<CT>(bn_tab, v, col[q]);
        if constexpr (AFF == 2)      // (one tap, stride 1: position j IS row j of the tensor)
          v = bn_bwd_on_load<CT>(bn_tab, bn, v, *reinterpret_cast<const f32x4*>(bn.z + off), col[q], j0 + row_[q], c0 + col[q]);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] *= sc;
      *reinterpret_cast<f32x4*>(tile + row_[q] * PITCH + col[q]) = v;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int i = threadIdx.x + 256 * q, cr = i >> 3, pq = i & 7, jb = (pq & 3) * 8;
      h8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = tile[(jb + e) * PITCH + cr];
        const _Float16 hi = (_Float16)v;
        o[e] = pq < 4 ? hi : (_Float16)(v - (float)hi);
        amax = fmaxf(amax, fabsf(v));
      }
      if (AFF != 2 || c0 + cr < C)
        *reinterpret_cast<h8*>(reinterpret_cast<_Float16*>(out + ((long long)tap * C + c0 + cr) * ldo + j0) + (pq < 4 ? 0 : 32) + jb) = o;
    }
    if (nhwc_out) {   // (one tap, stride 1, no padding: position j IS row j of x) the same tile as split NHWC rows, from the one read
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const int i = threadIdx.x + 256 * q, pq = i & 7, t = i >> 3, cb = t % (CT / 32), nl = t / (CT / 32), jb = (pq & 3) * 8;
        h8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float v = tile[nl * PITCH + cb * 32 + jb + e];
          const _Float16 hi = (_Float16)v;
          o[e] = pq < 4 ? hi : (_Float16)(v - (float)hi);
        }
        if (j0 + nl < J && c0 + cb * 32 < ldn)
          *reinterpret_cast<h8*>(reinterpret_cast<_Float16*>(nhwc_out + (long long)(j0 + nl) * ldn + c0 + cb * 32) + (pq < 4 ? 0 : 32) + jb) = o;
      }
    }
    __syncthreads();
  }
  dlip_report_range(amax, status);
}

// Operands of the weight gradient RUN AS A CONVOLUTION (deeplip_amd.autograd_video.wgrad_as_conv): an NHWC tensor x [N,H,W,C]
// becomes [C][H][W][N32] in the split format -- the images are the "channels" the convolution reduces over, 32 of them per
// 128-B block (32 hi halves | 32 lo halves), zero for n >= N:
//   out[((c * H + h) * W + w) * N32 + n]  <-  scale * x[n, h, w, c]
// One workgroup = 32 images x 32 channels of one pixel through a 32 x 33 LDS tile: 128-B reads along c, one 16-B store per thread
// along n.  Unlike the reduction-major operand above this is ONE copy of the tensor, not R*S of them: the taps are the
// convolution kernel's own address walk.
// With group > 0 the images are frames of clips of `group` frames and image n reads frame n + shift of ITS clip (zeros when that
// leaves the clip): the temporal taps of the stem's Conv3d as five shifted copies of the clip (dlip_wgrad_chwn_f32's callers pass
// group = 0).
__global__ __launch_bounds__(256) void wgrad_chwn_kernel(const float* __restrict__ x, float* __restrict__ out, int N, int HW, int ldx, int C,
                                                         int N32, const float* __restrict__ scale, DlipRange status, int group = 0,
                                                         int shift = 0, int layout = 0, float* __restrict__ nhwc_out = nullptr) {
  __shared__ float tile[32][33];
  const int n0 = blockIdx.x * 32, c0 = blockIdx.y * 32, p = blockIdx.z;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const float sc = scale ? scale[0] : 1.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int n = n0 + ty + 8 * q;
    float v = 0.f;
    const bool in_clip = group == 0 || (unsigned)(n % group + shift) < (unsigned)group;
    if (n < N && c0 + tx < C && in_clip) v = x[((long long)(n + shift) * HW + p) * ldx + c0 + tx] * sc;
    tile[ty + 8 * q][tx] = v;
  }
  __syncthreads();
  const int cr = threadIdx.x >> 3, pq = threadIdx.x & 7;
  const int jb = (pq & 3) * 8;
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  h8 o;
  float amax = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float v = tile[jb + e][cr];
    const _Float16 hi = (_Float16)v;
    o[e] = pq < 4 ? hi : (_Float16)(v - (float)hi);
    amax = fmaxf(amax, fabsf(v));
  }
  // layout 0: [c][pixel][N32] (pixel-major: a pixel's 32-image slices side by side); 1: [c][slice][pixel][32] (SLICE-major: one
  // slice of all pixels of an image is one contiguous plane -- what dlip_wgrad_conv_f16x3 reads: consecutive filter taps are
  // adjacent 128-byte lines); 2: the stem's clip copies, where the "channels" ARE the pixels: [slice][pixel c][32]
  const long long blk = layout == 0 ? ((long long)(c0 + cr) * HW + p) * N32 + n0
                      : layout == 1 ? (((long long)(c0 + cr) * (N32 / 32) + blockIdx.x) * HW + p) * 32
                                    : ((long long)blockIdx.x * C + (c0 + cr)) * 32;
  if (c0 + cr < C) *reinterpret_cast<h8*>(reinterpret_cast<_Float16*>(out + blk) + (pq < 4 ? 0 : 32) + jb) = o;
  // The SAME tile in the split activation format of the convolution kernels ([N,H,W,C]: per pixel and 32 channels one block of
  // 32 hi | 32 lo halves): what the forward convolution reads of x, and the data-gradient convolution of dy -- written here,
  // from the one read of the tensor, instead of by a split pass of its own (C % 32 == 0).
  if (nhwc_out) {
    const int nl = threadIdx.x >> 3, n = n0 + nl;
    h8 q;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = tile[nl][jb + e];
      const _Float16 hi = (_Float16)v;
      q[e] = pq < 4 ? hi : (_Float16)(v - (float)hi);
    }
    if (n < N) *reinterpret_cast<h8*>(reinterpret_cast<_Float16*>(nhwc_out + ((long long)n * HW + p) * C + c0) + (pq < 4 ? 0 : 32) + jb) = q;
  }
  dlip_report_range(amax, status);
}

// The same images from WIDER tiles (round 4): one workgroup = 32 images x CT channels (64 | 128) of one pixel, read as 16-B
// quads (CT / 32 per thread in flight instead of four 4-B loads) and written as CT / 32 sixteen-byte pieces per thread -- the
// 32 x 32 version moved 4 KB per workgroup between two barriers and held 2.8 TB/s.  group = 0, layouts 0 / 1, C % CT == 0.
template <int CT, int AFF = 0>      // AFF as in wgrad_operand_wide_kernel
__global__ __launch_bounds__(256) void wgrad_chwn_wide_kernel(const float* __restrict__ x, float* __restrict__ out, int N, int HW, int ldx, int C,
                                                              int N32, const float* __restrict__ scale, DlipRange status, int layout,
                                                              float* __restrict__ nhwc_out, const BnOnLoad bn = BnOnLoad{}) {
  constexpr int PITCH = CT + 4, Q = CT / 32;              // floats per LDS row (16-B aligned rows); quads / pieces per thread
  __shared__ __attribute__((aligned(16))) float tile[32 * PITCH];
  __shared__ __attribute__((aligned(16))) float bn_tab[AFF == 2 ? 6 * CT : AFF == 1 ? 5 * CT : 4];
  const int n0 = blockIdx.x * 32, c0 = blockIdx.y * CT, p = blockIdx.z;
  const float sc = scale ? scale[0] : 1.f;
  if constexpr (AFF == 1) bn_on_load_stage<CT>(bn, c0, bn_tab);
  if constexpr (AFF == 2) bn_bwd_on_load_stage<CT>(bn, c0, bn_tab);
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int i = threadIdx.x + 256 * q, row = i / (CT / 4), col4 = i - row * (CT / 4);
    const int n = n0 + row;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (n < N) {
      const long long off = ((long long)n * HW + p) * ldx + c0 + col4 * 4;
      v = *reinterpret_cast<const f32x4*>(x + off);
      if constexpr (AFF == 1) v = bn_on_load<CT>(bn_tab, v, col4 * 4);
      if constexpr (AFF == 2) v = bn_bwd_on_load<CT>(bn_tab, bn, v, *reinterpret_cast<const f32x4*>(bn.z + off), col4 * 4);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] *= sc;
    *reinterpret_cast<f32x4*>(tile + row * PITCH + col4 * 4) = v;
  }
  __syncthreads();
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  float amax = 0.f;
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int i = threadIdx.x + 256 * q, cr = i >> 3, pq = i & 7, jb = (pq & 3) * 8;
    h8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = tile[(jb + e) * PITCH + cr];
      const _Float16 hi = (_Float16)v;
      o[e] = pq < 4 ? hi : (_Float16)(v - (float)hi);
      amax = fmaxf(amax, fabsf(v));
    }
    const long long blk = layout == 0 ? ((long long)(c0 + cr) * HW + p) * N32 + n0
                                      : (((long long)(c0 + cr) * (N32 / 32) + blockIdx.x) * HW + p) * 32;
    *reinterpret_cast<h8*>(reinterpret_cast<_Float16*>(out + blk) + (pq < 4 ? 0 : 32) + jb) = o;
  }
  if (nhwc_out) {   // the same tile as split NHWC rows: image nl, 32-channel block cb: 8 pieces of 16 B
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int i = threadIdx.x + 256 * q, pq = i & 7, t = i >> 3, cb = t % (CT / 32), nl = t / (CT / 32), jb = (pq & 3) * 8;
      const int n = n0 + nl;
      h8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = tile[nl * PITCH + cb * 32 + jb + e];
        const _Float16 hi = (_Float16)v;
        o[e] = pq < 4 ? hi : (_Float16)(v - (float)hi);
      }
      if (n < N) *reinterpret_cast<h8*>(reinterpret_cast<_Float16*>(nhwc_out + ((long long)n * HW + p) * C + c0 + cb * 32) + (pq < 4 ? 0 : 32) + jb) = o;
    }
  }
  dlip_report_range(amax, status);
}

// Zero insertion of a strided layer's output gradient.  One workgroup row = one output row (n, hu): its image coordinates once per
// workgroup, 32-bit arithmetic per element (the first version took three 64-bit quotients per 16 bytes: 80 us per launch on
// layer-2-sized maps, 1.9 TB/s; round 4).
__global__ __launch_bounds__(256) void upsample_zero_kernel(const f32x4* __restrict__ dz, f32x4* __restrict__ out, int Ho, int Wo,
                                                            int Hu, int Wu, int C4, int sh, int sw, int rows) {
  const int rowlen = Wu * C4;
  for (int row = blockIdx.y; row < rows; row += gridDim.y) {
    const int n = row / Hu, hu = row - n * Hu;
    const bool row_ok = hu % sh == 0 && hu / sh < Ho;
    const f32x4* src = dz + ((long long)n * Ho + hu / sh) * Wo * C4;
    f32x4* dst = out + (long long)row * rowlen;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < rowlen; e += gridDim.x * 256) {
      const int wu = e / C4, c4 = e - wu * C4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (row_ok && wu % sw == 0 && wu / sw < Wo) v = src[(wu / sw) * C4 + c4];
      dst[e] = v;
    }
  }
}

// Zero insertion AND the lifted split operand of the strided layer's data-gradient convolution in one pass: the inserted tensor
// (4 x the gradient at stride 2: 230 MB for layer 2 at B = 32) was written as fp32, read back and written again in the split
// format (upsample_zero + split_pack_scaled: 0.75 ms of a lip-clip training step for six strided convolutions).  C % 32 == 0.
__global__ __launch_bounds__(256) void upsample_zero_split_kernel(const f32x4* __restrict__ dz, float* __restrict__ out, const float* __restrict__ scale,
                                                                  int Ho, int Wo, int Hu, int Wu, int C4, int sh, int sw, int rows, DlipRange status) {
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  const float sc = scale[0];
  const int rowlen = Wu * C4;
  float amax = 0.f;
  for (int row = blockIdx.y; row < rows; row += gridDim.y) {
    const int n = row / Hu, hu = row - n * Hu;
    const bool row_ok = hu % sh == 0 && hu / sh < Ho;
    const f32x4* src = dz + ((long long)n * Ho + hu / sh) * Wo * C4;
    float* dst = out + (long long)row * rowlen * 4;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < rowlen; e += gridDim.x * 256) {
      const int wu = e / C4, c4 = e - wu * C4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (row_ok && wu % sw == 0 && wu / sw < Wo) v = src[(wu / sw) * C4 + c4];
      h4 hi, lo;
#pragma unroll
      for (int k = 0; k < 4; ++k) { const float t = v[k] * sc; hi[k] = (_Float16)t; lo[k] = (_Float16)(t - (float)hi[k]); amax = fmaxf(amax, fabsf(t)); }
      float* b = dst + ((long long)wu * C4 + (c4 & ~7)) * 4;        // the pixel's 32-channel block: 32 hi halves | 32 lo halves
      const int q = c4 & 7;
      *reinterpret_cast<h4*>(b + q * 2) = hi;
      *reinterpret_cast<h4*>(b + 16 + q * 2) = lo;
    }
  }
  dlip_report_range_block(amax, status);
}

__global__ __launch_bounds__(256) void prelu_fwd_kernel(const f32x4* __restrict__ x, const float* __restrict__ slope,
                                                        f32x4* __restrict__ y, int C4, long long n4) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C4) * 4;
    const f32x4 v = x[i];
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = v[k] >= 0.f ? v[k] : v[k] * slope[c + k];
    y[i] = o;
  }
}

// The end of a residual block in one pass: s = a + b (kept: the backward needs the pre-activation), y = prelu(s)
// (resnet.py:66-68 `out += residual; out = relu2(out)`; tcn.py:114 `relu_final(out + res)`).
__global__ __launch_bounds__(256) void add_prelu_fwd_kernel(const f32x4* __restrict__ a, const f32x4* __restrict__ b, const float* __restrict__ slope,
                                                            f32x4* __restrict__ s_out, f32x4* __restrict__ y, int C4, long long n4) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C4) * 4;
    const f32x4 u = a[i], v = b[i];
    f32x4 sv, o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      sv[k] = u[k] + v[k];
      o[k] = sv[k] >= 0.f ? sv[k] : sv[k] * slope[c + k];
    }
    s_out[i] = sv;
    y[i] = o;
  }
}

// dx = dy * (x >= 0 ? 1 : slope);  t = (x >= 0 ? 0 : dy * x)  (column sums of t = the slope gradient)
__global__ __launch_bounds__(256) void prelu_bwd_kernel(const f32x4* __restrict__ dy, const f32x4* __restrict__ x,
                                                        const float* __restrict__ slope, f32x4* __restrict__ dx,
                                                        f32x4* __restrict__ t, int C4, long long n4) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C4) * 4;
    const f32x4 v = x[i], g = dy[i];
    f32x4 o, u;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      o[k] = v[k] >= 0.f ? g[k] : g[k] * slope[c + k];
      u[k] = v[k] >= 0.f ? 0.f : g[k] * v[k];
    }
    dx[i] = o;
    t[i] = u;
  }
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          f32x4* __restrict__ dx, int H, int W, int Ho, int Wo, int C4,
                                                          long long n4) {
  const int C = C4 * 4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C4) * 4;
    const long long p = i / C4;
    const int w = (int)(p % W);
    const long long t = p / W;
    const int h = (int)(t % H);
    const long long n = t / H;
    const float* xn = x + n * H * W * C + c;
    const f32x4 me = *reinterpret_cast<const f32x4*>(xn + ((long long)h * W + w) * C);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // windows (ho, wo) with 2 ho - 1 <= h <= 2 ho + 1
    for (int ho = h / 2; ho <= (h + 1) / 2; ++ho) {
      if (ho >= Ho) continue;
      for (int wo = w / 2; wo <= (w + 1) / 2; ++wo) {
        if (wo >= Wo) continue;
        // first maximum of the window in row-major order; does it sit at (h, w)?
        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int bh[4] = {-1, -1, -1, -1}, bw[4] = {-1, -1, -1, -1};
        for (int r = 0; r < 3; ++r) {
          const int hh = 2 * ho - 1 + r;
          if ((unsigned)hh >= (unsigned)H) continue;
          for (int s = 0; s < 3; ++s) {
            const int ww = 2 * wo - 1 + s;
            if ((unsigned)ww >= (unsigned)W) continue;
            const f32x4 v = *reinterpret_cast<const f32x4*>(xn + ((long long)hh * W + ww) * C);
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (v[k] > best[k] || bh[k] < 0) { best[k] = v[k]; bh[k] = hh; bw[k] = ww; }
          }
        }
        const f32x4 g = *reinterpret_cast<const f32x4*>(dy + ((n * Ho + ho) * Wo + wo) * C + c);
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (bh[k] == h && bw[k] == w) acc[k] += g[k];
      }
    }
    (void)me;
    dx[i] = acc;
  }
}

// dx[n, p, :] = dy[n, :] * (lengths ? (p < len[n] ? 1 / len[n] : 0) : scale)
__global__ __launch_bounds__(256) void row_broadcast_kernel(const f32x4* __restrict__ dy, const int* __restrict__ lengths,
                                                            f32x4* __restrict__ dx, int P, int C4, float scale, long long n4) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    const long long t = i / C4;
    const int p = (int)(t % P);
    const long long n = t / P;
    float sc = scale;
    if (lengths) {
      const int len = lengths[n] < P ? lengths[n] : P;
      sc = (p < len && len > 0) ? 1.f / (float)len : 0.f;
    }
    const f32x4 g = dy[n * C4 + c4];
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = g[k] * sc;
    dx[i] = o;
  }
}

// col[j, k]: j = ((b T + t) Ho + ho) Wo + wo, k = (dt 7 + r) 7 + s for k < 245, zero for 245..247
__global__ __launch_bounds__(256) void stem_im2col_kernel(const float* __restrict__ x, float* __restrict__ col, int T, int H,
                                                          int W, int Ho, int Wo, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int k = (int)(i % 248);
    const long long j = i / 248;
    float v = 0.f;
    if (k < 245) {
      const int s = k % 7, r = (k / 7) % 7, dt = k / 49;
      const int wo = (int)(j % Wo);
      const long long q = j / Wo;
      const int ho = (int)(q % Ho);
      const long long bt = q / Ho;
      const int t = (int)(bt % T);
      const long long b = bt / T;
      const int tt = t + dt - 2, hh = 2 * ho + r - 3, ww = 2 * wo + s - 3;
      if ((unsigned)tt < (unsigned)T && (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W)
        v = x[((b * T + tt) * H + hh) * W + ww];
    }
    col[i] = v;
  }
}

// MaxPool3d((1,3,3),(1,2,2),(0,1,1)) of a TRAINING step: the forward also records WHERE each maximum sits (tap r*3+s of the first
// maximum in row-major window order -- the tie rule of the backward kernel above), one byte per output element, so that the
// backward reads at most four (index, dy) pairs per input pixel instead of re-scanning up to four 3x3 windows of x
// (0.84 -> 0.3 ms per step at B = 32: the pooled tensor's input is 460 MB).
__global__ __launch_bounds__(256) void maxpool_idx_fwd_kernel(const float* __restrict__ x, f32x4* __restrict__ y, uint32_t* __restrict__ idx,
                                                              int H, int W, int Ho, int Wo, int C4, long long n4) {
  const int C = C4 * 4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C4) * 4;
    const long long p = i / C4;
    const int wo = (int)(p % Wo);
    const long long t = p / Wo;
    const int ho = (int)(t % Ho);
    const long long n = t / Ho;
    const float* xn = x + n * H * W * C + c;
    f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    uint32_t code = 0xFFFFFFFFu;                      // 0xFF per channel: nothing seen yet
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int hh = 2 * ho - 1 + r;
      if ((unsigned)hh >= (unsigned)H) continue;
#pragma unroll
      for (int s_ = 0; s_ < 3; ++s_) {
        const int ww = 2 * wo - 1 + s_;
        if ((unsigned)ww >= (unsigned)W) continue;
        const f32x4 v = *reinterpret_cast<const f32x4*>(xn + ((long long)hh * W + ww) * C);
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (v[k] > best[k] || ((code >> (8 * k)) & 0xFFu) == 0xFFu) {
            best[k] = v[k];
            code = (code & ~(0xFFu << (8 * k))) | ((uint32_t)(r * 3 + s_) << (8 * k));
          }
      }
    }
    y[i] = best;
    idx[i] = code;
  }
}

__global__ __launch_bounds__(256) void maxpool_idx_bwd_kernel(const uint32_t* __restrict__ idx, const f32x4* __restrict__ dy, f32x4* __restrict__ dx,
                                                              int H, int W, int Ho, int Wo, int C4, long long n4) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    const long long p = i / C4;
    const int w = (int)(p % W);
    const long long t = p / W;
    const int h = (int)(t % H);
    const long long n = t / H;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int ho = h / 2; ho <= (h + 1) / 2; ++ho) {
      if (ho >= Ho) continue;
      const int r = h - (2 * ho - 1);
      for (int wo = w / 2; wo <= (w + 1) / 2; ++wo) {
        if (wo >= Wo) continue;
        const uint32_t me = (uint32_t)(r * 3 + (w - (2 * wo - 1)));
        const long long o = ((n * Ho + ho) * Wo + wo) * C4 + c4;
        const uint32_t code = idx[o];
        const f32x4 g = dy[o];
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (((code >> (8 * k)) & 0xFFu) == me) acc[k] += g[k];
      }
    }
    dx[i] = acc;
  }
}

// The stem's weight-gradient operand in ONE pass (was: im2col [J, 248] -> transpose -> split, three round trips of a 1.8 GB matrix
// at B = 32): out[tap * ldo + j] = x[b, t + dt - 2, 2 ho + r - 3, 2 wo + s - 3] (zero outside the clip, for j >= J and for the
// padding rows tap = 245..247), tap = (dt 7 + r) 7 + s, j = ((b T + t) Ho + ho) Wo + wo, in the GEMM's reduction-major split image
// (per row and 32 positions one 128-B block: 32 hi halves | 32 lo halves).  C = 1: the positions ARE the contiguous axis of the
// input (stride 2), so no LDS transposition -- a thread owns 8 consecutive positions (their clip coordinates computed once)
// and walks the tap rows tap_sub, tap_sub + 64, ...; four neighbouring threads complete a row's 128-B block with 16-B stores.
// The clip is read ~61 times per pixel, from L1 / L2 (29 MB at B = 32); HBM sees the 1.8 GB of the operand once.
__global__ __launch_bounds__(256) void stem_wgrad_operand_kernel(const float* __restrict__ x, float* __restrict__ out, int T, int H, int W,
                                                                 int Ho, int Wo, long long J, long long ldo, DlipRange status) {
  const long long j0 = (long long)blockIdx.x * 32;
  const int pq = threadIdx.x & 3, tap_sub = threadIdx.x >> 2;
  int base[8], h0[8], w0[8], tt[8];
  {
    long long j = j0 + pq * 8;
    int wo = (int)(j % Wo);
    long long q = j / Wo;
    int ho = (int)(q % Ho);
    long long bt = q / Ho;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const bool in = j + e < J;
      base[e] = (int)(bt * H * W);
      tt[e] = in ? (int)(bt % T) : -(1 << 20);
      h0[e] = 2 * ho - 3;
      w0[e] = 2 * wo - 3;
      if (++wo == Wo) { wo = 0; if (++ho == Ho) { ho = 0; ++bt; } }
    }
  }
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  float amax = 0.f;
  for (int tap = tap_sub; tap < 248; tap += 64) {
    h8 hi, lo;
    const int s_ = tap % 7, r = (tap / 7) % 7, dt = tap / 49;      // tap >= 245: dt = 5 -> t + 3 may be < T: masked below
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int t2 = tt[e] + dt - 2, hh = h0[e] + r, ww = w0[e] + s_;
      float v = 0.f;
      if (tap < 245 && (unsigned)t2 < (unsigned)T && (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W)
        v = x[base[e] + ((dt - 2) * H + hh) * W + ww];
      const _Float16 a = (_Float16)v;
      hi[e] = a; lo[e] = (_Float16)(v - (float)a);
      amax = fmaxf(amax, fabsf(v));
    }
    _Float16* row = reinterpret_cast<_Float16*>(out + (long long)tap * ldo + j0);
    *reinterpret_cast<h8*>(row + pq * 8) = hi;
    *reinterpret_cast<h8*>(row + 32 + pq * 8) = lo;
  }
  dlip_report_range(amax, status);
}

// The stem's CURRENT weights [K, 245] -> the split LDS image of stem3d_f16x3.hip on the device (packing.split_stem_weights does
// this on the host once per load_state_dict; under training the weights change every step).  One 64-lane workgroup per output
// channel: row maximum -> scale = 2^floor(log2(1023 / max)) -> per channel 1184 B: [36 kernel rows (dt 7 + r; row 35 zero) x 8 hi
// halves (tap 7 zero)] [the same, lo halves] + 32 B of padding.
__global__ __launch_bounds__(64) void split_stem_weights_kernel(const float* __restrict__ w, float* __restrict__ img, float* __restrict__ scale) {
  const float* row = w + (long long)blockIdx.x * 245;
  float m = 0.f;
  for (int i = threadIdx.x; i < 245; i += 64) m = fmaxf(m, fabsf(row[i]));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  float sc = 1.f;
  if (m > 0.f && m < 3.0e38f) sc = exp2f(floorf(log2f(1023.0f / m)));
  if (threadIdx.x == 0) scale[blockIdx.x] = sc;
  _Float16* out = reinterpret_cast<_Float16*>(img + (long long)blockIdx.x * (1184 / 4));
  for (int i = threadIdx.x; i < 592; i += 64) {
    const int q = i < 288 ? i : i - 288;               // slot (row, tap) of the hi plane / the lo plane; i >= 576: padding
    const int kr = q >> 3, tap = q & 7;
    float t = 0.f;
    if (i < 576 && kr < 35 && tap < 7) t = row[kr * 7 + tap] * sc;
    const _Float16 hi = (_Float16)t;
    out[i] = i < 288 ? hi : (_Float16)(t - (float)hi);
  }
}

__global__ __launch_bounds__(256) void mul_mask_kernel(const float* __restrict__ x, const float* __restrict__ mask,
                                                       float* __restrict__ y, float scale, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) y[i] = x[i] * mask[i] * scale;
}

// nn.Dropout with the keep test in the kernel: y = u >= p ? x * scale : 0 from the uniform draws u themselves (was: a compare, a
// cast and the multiplication as three launches behind the generator's; u is what the backward keeps).
__global__ __launch_bounds__(256) void dropout_keep_kernel(const float* __restrict__ x, const float* __restrict__ u, float* __restrict__ y,
                                                           float p, float scale, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) y[i] = u[i] >= p ? x[i] * scale : 0.f;
}

// The end of a multibranch TCN stage in ONE launch (was one strided row copy per branch): out[b, t, off_j + c] = z_j[b, t + (L_j - T) / 2, c]
// -- the symmetric chomp (tcn.py:52-59) and the concatenation along channels (tcn.py:96-108) of up to four branches [B, L_j, C_j].
// BWD: the reverse -- g_j[b, l, c] = dy[b, l - (L_j - T) / 2, off_j + c] inside the kept rows, 0 in the chomped ones.
struct ChompCat {
  float* z[4];            // branch tensors (forward: sources; backward: destinations)
  int L[4], C4[4], off4[4];
  long long start[5];     // prefix sums of the branches' float4 counts (forward: B T C4_j; backward: B L_j C4_j)
  int nb, T, Ct4;
};
template <bool BWD>
__global__ __launch_bounds__(256) void chomp_concat_kernel(const ChompCat cc, float* __restrict__ cat) {
  const long long n4 = cc.start[cc.nb];
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    int j = 0;
#pragma unroll
    for (int q = 1; q < 4; ++q) j += (q < cc.nb && i >= cc.start[q]) ? 1 : 0;
    const long long k = i - cc.start[j];
    const int C4 = cc.C4[j], L = cc.L[j], sh = (L - cc.T) / 2;
    const int c4 = (int)(k % C4);
    const long long row = k / C4;
    if (!BWD) {
      const int t = (int)(row % cc.T);
      const long long b = row / cc.T;
      const f32x4 v = *reinterpret_cast<const f32x4*>(cc.z[j] + ((b * L + t + sh) * C4 + c4) * 4);
      *reinterpret_cast<f32x4*>(cat + ((b * cc.T + t) * cc.Ct4 + cc.off4[j] + c4) * 4) = v;
    } else {
      const int l = (int)(row % L);
      const long long b = row / L;
      const int t = l - sh;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if ((unsigned)t < (unsigned)cc.T) v = *reinterpret_cast<const f32x4*>(cat + ((b * cc.T + t) * cc.Ct4 + cc.off4[j] + c4) * 4);
      *reinterpret_cast<f32x4*>(cc.z[j] + ((b * L + l) * C4 + c4) * 4) = v;
    }
  }
}

}  // namespace

#define ST(s) static_cast<hipStream_t>(s)

extern "C" int dlip_tap_gather_f32(const float* x, float* out, int64_t N, int32_t H, int32_t W, int32_t C, int32_t ldx,
                                   int32_t Ho, int32_t Wo, int32_t stride_h, int32_t stride_w, int32_t off_h, int32_t off_w,
                                   int32_t ldo, dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && out && N > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0 && (ldx & 3) == 0 && ldx >= C && Ho > 0 && Wo > 0 &&
                 stride_h > 0 && stride_w > 0 && ldo >= C && (ldo & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0);
  const long long n4 = (long long)N * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(tap_gather_kernel, dim3(grid_for(n4)), dim3(256), 0, ST(stream), x, out, H, W, ldx, C / 4, Ho, Wo, stride_h,
                     stride_w, off_h, off_w, ldo, n4);
  return dlip_launch_status();
}

extern "C" int dlip_wgrad_operand_f32(const float* x, float* out, int64_t ld_out, int64_t N, int32_t H, int32_t W, int32_t C, int32_t ldx,
                                      int32_t Ho, int32_t Wo, int32_t stride_h, int32_t stride_w, int32_t R, int32_t S, int32_t dil_h,
                                      int32_t dil_w, int32_t pad_h, int32_t pad_w, const float* scale, dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && out && N > 0 && H > 0 && W > 0 && C > 0 && ldx >= C && Ho > 0 && Wo > 0 && stride_h > 0 &&
                 stride_w > 0 && R > 0 && S > 0 && (reinterpret_cast<uintptr_t>(out) & 127) == 0);
  const long long J = (long long)N * Ho * Wo;
  DLIP_CHECK_ARG(J < (1ll << 28) && (long long)N * H < (1ll << 30) && ld_out >= J && (ld_out & 31) == 0 && (C + 31) / 32 <= 65535);
  const bool quads = (ldx & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
  if (quads && C % 128 == 0)
    hipLaunchKernelGGL(wgrad_operand_wide_kernel<128>, dim3((unsigned)(ld_out / 32), (unsigned)(C / 128)), dim3(256), 0, ST(stream), x, out, H, W,
                       ldx, C, Ho, Wo, stride_h, stride_w, R, S, dil_h, dil_w, pad_h, pad_w, (int)J, (long long)ld_out, scale,
                       dlip_range_for(DLIP_ST_PACK));
  else if (quads && C % 64 == 0)
    hipLaunchKernelGGL(wgrad_operand_wide_kernel<64>, dim3((unsigned)(ld_out / 32), (unsigned)(C / 64)), dim3(256), 0, ST(stream), x, out, H, W,
                       ldx, C, Ho, Wo, stride_h, stride_w, R, S, dil_h, dil_w, pad_h, pad_w, (int)J, (long long)ld_out, scale,
                       dlip_range_for(DLIP_ST_PACK));
  else
    hipLaunchKernelGGL(wgrad_operand_kernel, dim3((unsigned)(ld_out / 32), (unsigned)((C + 31) / 32)), dim3(256), 0, ST(stream), x, out, H, W,
                       ldx, C, Ho, Wo, stride_h, stride_w, R, S, dil_h, dil_w, pad_h, pad_w, (int)J, (long long)ld_out, scale,
                       dlip_range_for(DLIP_ST_PACK));
  return dlip_launch_status();
}

extern "C" int dlip_wgrad_operand_split_f32(const float* x, float* out, int64_t ld_out, int64_t J, int32_t C, const float* scale,
                                            float* nhwc_split_out, dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && out && nhwc_split_out && J > 0 && J < (1ll << 28) && C > 0 && (C & 63) == 0 && ld_out >= J && (ld_out & 31) == 0);
  DLIP_CHECK_ARG(((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(nhwc_split_out)) & 127) == 0 &&
                 (reinterpret_cast<uintptr_t>(x) & 15) == 0 && C / 64 <= 65535);
  // x as [J, 1, 1, C]: one "image" per position, one tap
  if (C % 128 == 0)
    hipLaunchKernelGGL(wgrad_operand_wide_kernel<128>, dim3((unsigned)(ld_out / 32), (unsigned)(C / 128)), dim3(256), 0, ST(stream), x, out, 1, 1, C,
                       C, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, (int)J, (long long)ld_out, scale, dlip_range_for(DLIP_ST_PACK), nhwc_split_out);
  else
    hipLaunchKernelGGL(wgrad_operand_wide_kernel<64>, dim3((unsigned)(ld_out / 32), (unsigned)(C / 64)), dim3(256), 0, ST(stream), x, out, 1, 1, C,
                       C, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, (int)J, (long long)ld_out, scale, dlip_range_for(DLIP_ST_PACK), nhwc_split_out);
  return dlip_launch_status();
}

extern "C" int dlip_wgrad_chwn_f32(const float* x, float* out, int64_t N, int32_t H, int32_t W, int32_t C, int32_t ldx, int32_t N32,
                                   const float* scale, int32_t slice_major, float* nhwc_split_out, dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && out && N > 0 && H > 0 && W > 0 && C > 0 && ldx >= C && N32 >= N && (N32 & 31) == 0 &&
                 (reinterpret_cast<uintptr_t>(out) & 127) == 0);
  DLIP_CHECK_ARG((long long)H * W <= 65535 && (C + 31) / 32 <= 65535 && N < (1ll << 31));
  DLIP_CHECK_ARG(nhwc_split_out == nullptr || ((C & 31) == 0 && (reinterpret_cast<uintptr_t>(nhwc_split_out) & 127) == 0));
  const bool quads = (ldx & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
  if (quads && C % 128 == 0)
    hipLaunchKernelGGL(wgrad_chwn_wide_kernel<128>, dim3((unsigned)(N32 / 32), (unsigned)(C / 128), (unsigned)(H * W)), dim3(256), 0, ST(stream),
                       x, out, (int)N, H * W, ldx, C, N32, scale, dlip_range_for(DLIP_ST_PACK), slice_major ? 1 : 0, nhwc_split_out);
  else if (quads && C % 64 == 0)
    hipLaunchKernelGGL(wgrad_chwn_wide_kernel<64>, dim3((unsigned)(N32 / 32), (unsigned)(C / 64), (unsigned)(H * W)), dim3(256), 0, ST(stream),
                       x, out, (int)N, H * W, ldx, C, N32, scale, dlip_range_for(DLIP_ST_PACK), slice_major ? 1 : 0, nhwc_split_out);
  else
    hipLaunchKernelGGL(wgrad_chwn_kernel, dim3((unsigned)(N32 / 32), (unsigned)((C + 31) / 32), (unsigned)(H * W)), dim3(256), 0, ST(stream), x,
                       out, (int)N, H * W, ldx, C, N32, scale, dlip_range_for(DLIP_ST_PACK), 0, 0, slice_major ? 1 : 0, nhwc_split_out);
  return dlip_launch_status();
}

// (ABI 44) dlip_wgrad_operand_split_f32 / dlip_wgrad_chwn_f32 (with its split NHWC copy) of act(bn(x)) instead of x: BnOnLoad above.
extern "C" int dlip_wgrad_operand_split_bn_f32(const float* x, float* out, int64_t ld_out, int64_t J, int32_t C, const float* mean,
                                               const float* invstd, const float* gamma, const float* beta, const float* slope_vec,
                                               float slope, float* nhwc_split_out, dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && out && nhwc_split_out && mean && invstd && gamma && beta && J > 0 && J < (1ll << 28) && C > 0 && (C & 63) == 0 &&
                 ld_out >= J && (ld_out & 31) == 0);
  DLIP_CHECK_ARG(((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(nhwc_split_out)) & 127) == 0 &&
                 (reinterpret_cast<uintptr_t>(x) & 15) == 0 && C / 64 <= 65535);
  const BnOnLoad bn = {mean, invstd, gamma, beta, slope_vec, slope};
  if (C % 128 == 0)
    hipLaunchKernelGGL((wgrad_operand_wide_kernel<128, true>), dim3((unsigned)(ld_out / 32), (unsigned)(C / 128)), dim3(256), 0, ST(stream), x, out,
                       1, 1, C, C, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, (int)J, (long long)ld_out, nullptr, dlip_range_for(DLIP_ST_PACK), nhwc_split_out, bn);
  else
    hipLaunchKernelGGL((wgrad_operand_wide_kernel<64, true>), dim3((unsigned)(ld_out / 32), (unsigned)(C / 64)), dim3(256), 0, ST(stream), x, out,
                       1, 1, C, C, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, (int)J, (long long)ld_out, nullptr, dlip_range_for(DLIP_ST_PACK), nhwc_split_out, bn);
  return dlip_launch_status();
}

extern "C" int dlip_wgrad_chwn_bn_f32(const float* x, float* out, int64_t N, int32_t H, int32_t W, int32_t C, int32_t N32,
                                      const float* mean, const float* invstd, const float* gamma, const float* beta,
                                      const float* slope_vec, float slope, float* nhwc_split_out, dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && out && nhwc_split_out && mean && invstd && gamma && beta && N > 0 && H > 0 && W > 0 && C > 0 && (C & 63) == 0 &&
                 N32 >= N && (N32 & 31) == 0);
  DLIP_CHECK_ARG(((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(nhwc_split_out)) & 127) == 0 &&
                 (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (long long)H * W <= 65535 && C / 64 <= 65535 && N < (1ll << 31));
  const BnOnLoad bn = {mean, invstd, gamma, beta, slope_vec, slope};
  if (C % 128 == 0)
    hipLaunchKernelGGL((wgrad_chwn_wide_kernel<128, true>), dim3((unsigned)(N32 / 32), (unsigned)(C / 128), (unsigned)(H * W)), dim3(256), 0,
                       ST(stream), x, out, (int)N, H * W, C, C, N32, nullptr, dlip_range_for(DLIP_ST_PACK), 1, nhwc_split_out, bn);
  else
    hipLaunchKernelGGL((wgrad_chwn_wide_kernel<64, true>), dim3((unsigned)(N32 / 32), (unsigned)(C / 64), (unsigned)(H * W)), dim3(256), 0,
                       ST(stream), x, out, (int)N, H * W, C, C, N32, nullptr, dlip_range_for(DLIP_ST_PACK), 1, nhwc_split_out, bn);
  return dlip_launch_status();
}

// (ABI 47) The two producers with a train-mode BatchNorm's BACKWARD applied on load: sources dy and z (the BatchNorm's input: the raw
// convolution output), per-channel mean / invstd / gamma / beta and the two backward sums dgamma / dbeta (dlip_bn_rows_train_bwd_sums_f32),
// M = rows of the statistics; the value split is lift[0] * dz with dz = bn_bwd_apply_kernel's expression (same bits).  dz itself -- an
// activation-sized fp32 tensor written by the apply pass and read back here -- never exists.  nhwc_split_out nullable.
extern "C" int dlip_wgrad_operand_split_bnbwd_f32(const float* dy, const float* z, float* out, int64_t ld_out, int64_t J, int32_t C,
                                                  const float* mean, const float* invstd, const float* gamma, const float* beta,
                                                  const float* dgamma, const float* dbeta, int64_t M, float slope, int32_t act_first,
                                                  const float* lift, float* nhwc_split_out, int32_t ld_nhwc, const float* ms_coef,
                                                  int32_t ms_T, dlip_stream_t stream) {
  // (ABI 49) C % 4 == 0 is enough (a ragged last channel block); ld_nhwc: row pitch of nhwc_split_out (0 = C; else >= C, a multiple of 32);
  // ms_coef / ms_T: dy formed on load from a MeanStdPooling's coefficients (dlip_meanstd_bwd_coef_f32; then dy may be NULL)
  DLIP_CHECK_ARG(z && out && mean && invstd && gamma && beta && dgamma && dbeta && lift && J > 0 && J < (1ll << 28) && M > 0 && C > 0 &&
                 (C & 3) == 0 && ld_out >= J && (ld_out & 31) == 0);
  DLIP_CHECK_ARG((dy != nullptr || ms_coef != nullptr) && (ms_coef == nullptr || (ms_T > 1 && !act_first && J % ms_T == 0)));
  DLIP_CHECK_ARG(ld_nhwc == 0 ? (C & 31) == 0 || nhwc_split_out == nullptr : (ld_nhwc >= C && (ld_nhwc & 31) == 0));
  DLIP_CHECK_ARG(((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(nhwc_split_out)) & 127) == 0 &&
                 ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(ms_coef)) & 15) == 0 &&
                 (C + 63) / 64 <= 65535);
  BnOnLoad bn = {mean, invstd, gamma, beta, nullptr, slope};
  bn.z = z; bn.dgamma = dgamma; bn.dbeta = dbeta; bn.invM = 1.f / (float)M; bn.act_first = act_first;
  bn.ms_coef = ms_coef; bn.ms_T = ms_T > 0 ? ms_T : 1; bn.C = C;
  const float* src = dy ? dy : z;      // (never read when the gradient is formed on load)
  if (C % 128 == 0 || C > 512)      // (a ragged last block of 128 wastes less than the 64-wide tile costs: 461 -> us on the E-TDNN's 1 500 channels)
    hipLaunchKernelGGL((wgrad_operand_wide_kernel<128, 2>), dim3((unsigned)(ld_out / 32), (unsigned)((C + 127) / 128)), dim3(256), 0, ST(stream), src, out,
                       1, 1, C, C, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, (int)J, (long long)ld_out, lift, dlip_range_for(DLIP_ST_PACK), nhwc_split_out, bn,
                       ld_nhwc);
  else
    hipLaunchKernelGGL((wgrad_operand_wide_kernel<64, 2>), dim3((unsigned)(ld_out / 32), (unsigned)((C + 63) / 64)), dim3(256), 0, ST(stream), src,
                       out, 1, 1, C, C, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, (int)J, (long long)ld_out, lift, dlip_range_for(DLIP_ST_PACK), nhwc_split_out,
                       bn, ld_nhwc);
  return dlip_launch_status();
}

extern "C" int dlip_wgrad_chwn_bnbwd_f32(const float* dy, const float* z, float* out, int64_t N, int32_t H, int32_t W, int32_t C, int32_t N32,
                                         const float* mean, const float* invstd, const float* gamma, const float* beta, const float* dgamma,
                                         const float* dbeta, int64_t M, float slope, int32_t act_first, const float* lift,
                                         float* nhwc_split_out, dlip_stream_t stream) {
  DLIP_CHECK_ARG(dy && z && out && mean && invstd && gamma && beta && dgamma && dbeta && lift && N > 0 && H > 0 && W > 0 && M > 0 && C > 0 &&
                 (C & 63) == 0 && N32 >= N && (N32 & 31) == 0);
  DLIP_CHECK_ARG(((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(nhwc_split_out)) & 127) == 0 &&
                 ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(z)) & 15) == 0 && (long long)H * W <= 65535 && C / 64 <= 65535 &&
                 N < (1ll << 31));
  BnOnLoad bn = {mean, invstd, gamma, beta, nullptr, slope};
  bn.z = z; bn.dgamma = dgamma; bn.dbeta = dbeta; bn.invM = 1.f / (float)M; bn.act_first = act_first; bn.C = C;
  if (C % 128 == 0)
    hipLaunchKernelGGL((wgrad_chwn_wide_kernel<128, 2>), dim3((unsigned)(N32 / 32), (unsigned)(C / 128), (unsigned)(H * W)), dim3(256), 0,
                       ST(stream), dy, out, (int)N, H * W, C, C, N32, lift, dlip_range_for(DLIP_ST_PACK), 1, nhwc_split_out, bn);
  else
    hipLaunchKernelGGL((wgrad_chwn_wide_kernel<64, 2>), dim3((unsigned)(N32 / 32), (unsigned)(C / 64), (unsigned)(H * W)), dim3(256), 0,
                       ST(stream), dy, out, (int)N, H * W, C, C, N32, lift, dlip_range_for(DLIP_ST_PACK), 1, nhwc_split_out, bn);
  return dlip_launch_status();
}

extern "C" int dlip_stem_wgrad_chwn_f32(const float* x, float* out, int32_t B, int32_t T, int32_t H, int32_t W, int32_t N32,
                                        int32_t slice_major, dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && out && B > 0 && T > 0 && H > 0 && W > 0 && N32 >= (long long)B * T && (N32 & 31) == 0 &&
                 (reinterpret_cast<uintptr_t>(out) & 127) == 0 && (H * W + 31) / 32 <= 65535);
  // the clip as [frames][1 pixel][H W "channels"]: out[dt][p][n] = x[n + dt - 2][p] inside the clip of frame n
  for (int dt = 0; dt < 5; ++dt)
    hipLaunchKernelGGL(wgrad_chwn_kernel, dim3((unsigned)(N32 / 32), (unsigned)((H * W + 31) / 32), 1u), dim3(256), 0, ST(stream), x,
                       out + (long long)dt * H * W * N32, B * T, 1, H * W, H * W, N32, nullptr, dlip_range_for(DLIP_ST_PACK), T, dt - 2,
                       slice_major ? 2 : 0);
  return dlip_launch_status();
}

extern "C" int dlip_upsample_zero_f32(const float* dz, float* out, int64_t N, int32_t Ho, int32_t Wo, int32_t Hu, int32_t Wu,
                                      int32_t C, int32_t stride_h, int32_t stride_w, dlip_stream_t stream) {
  DLIP_CHECK_ARG(dz && out && N > 0 && Ho > 0 && Wo > 0 && Hu > 0 && Wu > 0 && C > 0 && (C & 3) == 0 && stride_h > 0 && stride_w > 0);
  const long long rows = (long long)N * Hu;
  DLIP_CHECK_ARG(rows < (1ll << 31) && (long long)Wu * (C / 4) < (1ll << 30));
  const int rowlen = Wu * (C / 4);
  const unsigned gx = (unsigned)((rowlen + 255) / 256 > 16 ? 16 : (rowlen + 255) / 256);
  long long gy = 4096 / gx;                              // ~4 096 workgroups, each walking rows / gy output rows
  if (gy > rows) gy = rows;
  hipLaunchKernelGGL(upsample_zero_kernel, dim3(gx, (unsigned)gy), dim3(256), 0, ST(stream), reinterpret_cast<const f32x4*>(dz),
                     reinterpret_cast<f32x4*>(out), Ho, Wo, Hu, Wu, C / 4, stride_h, stride_w, (int)rows);
  return dlip_launch_status();
}

extern "C" int dlip_upsample_zero_split_f32(const float* dz, float* out_split, const float* scale, int64_t N, int32_t Ho, int32_t Wo,
                                            int32_t Hu, int32_t Wu, int32_t C, int32_t stride_h, int32_t stride_w, dlip_stream_t stream) {
  DLIP_CHECK_ARG(dz && out_split && scale && N > 0 && Ho > 0 && Wo > 0 && Hu > 0 && Wu > 0 && C > 0 && (C & 31) == 0 && stride_h > 0 && stride_w > 0);
  DLIP_CHECK_ARG(((reinterpret_cast<uintptr_t>(dz) & 15) | (reinterpret_cast<uintptr_t>(out_split) & 127)) == 0);
  const long long rows = (long long)N * Hu;
  DLIP_CHECK_ARG(rows < (1ll << 31) && (long long)Wu * (C / 4) < (1ll << 30));
  const int rowlen = Wu * (C / 4);
  const unsigned gx = (unsigned)((rowlen + 255) / 256 > 16 ? 16 : (rowlen + 255) / 256);
  long long gy = 4096 / gx;
  if (gy > rows) gy = rows;
  hipLaunchKernelGGL(upsample_zero_split_kernel, dim3(gx, (unsigned)gy), dim3(256), 0, ST(stream), reinterpret_cast<const f32x4*>(dz), out_split,
                     scale, Ho, Wo, Hu, Wu, C / 4, stride_h, stride_w, (int)rows, dlip_range_for(DLIP_ST_PACK));
  return dlip_launch_status();
}

extern "C" int dlip_prelu_rows_fwd_f32(const float* x, const float* slope, float* y, int64_t M, int32_t C, dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && slope && y && M > 0 && C > 0 && (C & 3) == 0);
  const long long n4 = (long long)M * (C / 4);
  hipLaunchKernelGGL(prelu_fwd_kernel, dim3(grid_for(n4)), dim3(256), 0, ST(stream), reinterpret_cast<const f32x4*>(x), slope,
                     reinterpret_cast<f32x4*>(y), C / 4, n4);
  return dlip_launch_status();
}

extern "C" int dlip_prelu_rows_bwd_f32(const float* dy, const float* x, const float* slope, float* dx, float* dslope_terms,
                                       int64_t M, int32_t C, dlip_stream_t stream) {
  DLIP_CHECK_ARG(dy && x && slope && dx && dslope_terms && M > 0 && C > 0 && (C & 3) == 0);
  const long long n4 = (long long)M * (C / 4);
  hipLaunchKernelGGL(prelu_bwd_kernel, dim3(grid_for(n4)), dim3(256), 0, ST(stream), reinterpret_cast<const f32x4*>(dy),
                     reinterpret_cast<const f32x4*>(x), slope, reinterpret_cast<f32x4*>(dx), reinterpret_cast<f32x4*>(dslope_terms),
                     C / 4, n4);
  return dlip_launch_status();
}

extern "C" int dlip_add_prelu_rows_fwd_f32(const float* a, const float* b, const float* slope, float* sum, float* y, int64_t M, int32_t C,
                                           dlip_stream_t stream) {
  DLIP_CHECK_ARG(a && b && slope && sum && y && M > 0 && C > 0 && (C & 3) == 0);
  const long long n4 = (long long)M * (C / 4);
  hipLaunchKernelGGL(add_prelu_fwd_kernel, dim3(grid_for(n4)), dim3(256), 0, ST(stream), reinterpret_cast<const f32x4*>(a),
                     reinterpret_cast<const f32x4*>(b), slope, reinterpret_cast<f32x4*>(sum), reinterpret_cast<f32x4*>(y), C / 4, n4);
  return dlip_launch_status();
}

extern "C" int dlip_maxpool3x3s2_bwd_f32(const float* x, const float* dy, float* dx, int64_t N, int32_t H, int32_t W, int32_t C,
                                         dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && dy && dx && N > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0);
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long long n4 = (long long)N * H * W * (C / 4);
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for(n4)), dim3(256), 0, ST(stream), x, dy, reinterpret_cast<f32x4*>(dx), H, W, Ho,
                     Wo, C / 4, n4);
  return dlip_launch_status();
}

extern "C" int dlip_maxpool3x3s2_idx_f32(const float* x, float* y, uint32_t* idx, int64_t N, int32_t H, int32_t W, int32_t C,
                                        dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && y && idx && N > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0);
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long long n4 = (long long)N * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(maxpool_idx_fwd_kernel, dim3(grid_for(n4)), dim3(256), 0, ST(stream), x, reinterpret_cast<f32x4*>(y), idx, H, W, Ho,
                     Wo, C / 4, n4);
  return dlip_launch_status();
}

extern "C" int dlip_maxpool3x3s2_bwd_idx_f32(const uint32_t* idx, const float* dy, float* dx, int64_t N, int32_t H, int32_t W, int32_t C,
                                            dlip_stream_t stream) {
  DLIP_CHECK_ARG(idx && dy && dx && N > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0);
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long long n4 = (long long)N * H * W * (C / 4);
  hipLaunchKernelGGL(maxpool_idx_bwd_kernel, dim3(grid_for(n4)), dim3(256), 0, ST(stream), idx, reinterpret_cast<const f32x4*>(dy),
                     reinterpret_cast<f32x4*>(dx), H, W, Ho, Wo, C / 4, n4);
  return dlip_launch_status();
}

extern "C" int dlip_row_broadcast_f32(const float* dy, const int32_t* lengths, float* dx, int64_t N, int32_t P, int32_t C,
                                      float scale, dlip_stream_t stream) {
  DLIP_CHECK_ARG(dy && dx && N > 0 && P > 0 && C > 0 && (C & 3) == 0);
  const long long n4 = (long long)N * P * (C / 4);
  hipLaunchKernelGGL(row_broadcast_kernel, dim3(grid_for(n4)), dim3(256), 0, ST(stream), reinterpret_cast<const f32x4*>(dy), lengths,
                     reinterpret_cast<f32x4*>(dx), P, C / 4, scale, n4);
  return dlip_launch_status();
}

extern "C" int dlip_stem_im2col_f32(const float* x, float* col, int32_t B, int32_t T, int32_t H, int32_t W, dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && col && B > 0 && T > 0 && H > 1 && W > 1 && (H & 1) == 0 && (W & 1) == 0);
  const long long n = (long long)B * T * (H / 2) * (W / 2) * 248;
  hipLaunchKernelGGL(stem_im2col_kernel, dim3(grid_for(n)), dim3(256), 0, ST(stream), x, col, T, H, W, H / 2, W / 2, n);
  return dlip_launch_status();
}

extern "C" int dlip_stem_wgrad_operand_f32(const float* x, float* out, int64_t ld_out, int32_t B, int32_t T, int32_t H, int32_t W,
                                           dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && out && B > 0 && T > 0 && H > 1 && W > 1 && (H & 1) == 0 && (W & 1) == 0 &&
                 (reinterpret_cast<uintptr_t>(out) & 127) == 0);
  const long long J = (long long)B * T * (H / 2) * (W / 2);
  DLIP_CHECK_ARG(ld_out >= J && (ld_out & 31) == 0 && (long long)B * T * H * W < (1ll << 31) && ld_out / 32 < (1ll << 31));
  hipLaunchKernelGGL(stem_wgrad_operand_kernel, dim3((unsigned)(ld_out / 32)), dim3(256), 0, ST(stream), x, out, T, H, W, H / 2, W / 2, J,
                     (long long)ld_out, dlip_range_for(DLIP_ST_PACK));
  return dlip_launch_status();
}

extern "C" int dlip_split_stem_weights_f32(const float* w, float* w_img, float* w_scale, int32_t K, dlip_stream_t stream) {
  DLIP_CHECK_ARG(w && w_img && w_scale && K > 0);
  hipLaunchKernelGGL(split_stem_weights_kernel, dim3((unsigned)K), dim3(64), 0, ST(stream), w, w_img, w_scale);
  return dlip_launch_status();
}

extern "C" int dlip_dropout_keep_f32(const float* x, const float* u, float* y, int64_t n, float p, float scale, dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && u && y && n > 0 && p >= 0.f && p < 1.f);
  hipLaunchKernelGGL(dropout_keep_kernel, dim3(grid_for(n)), dim3(256), 0, ST(stream), x, u, y, p, scale, (long long)n);
  return dlip_launch_status();
}

extern "C" int dlip_chomp_concat_f32(const float* const* branches, const int32_t* lengths, const int32_t* widths, int32_t n_branches,
                                     float* cat, int32_t B, int32_t T, int32_t backward, dlip_stream_t stream) {
  DLIP_CHECK_ARG(branches && lengths && widths && cat && n_branches >= 1 && n_branches <= 4 && B > 0 && T > 0);
  ChompCat cc;
  cc.nb = n_branches; cc.T = T;
  int off = 0;
  cc.start[0] = 0;
  for (int j = 0; j < 4; ++j) {
    if (j < n_branches) {
      DLIP_CHECK_ARG(branches[j] && widths[j] > 0 && (widths[j] & 3) == 0 && lengths[j] >= T && ((lengths[j] - T) & 1) == 0 &&
                     (reinterpret_cast<uintptr_t>(branches[j]) & 15) == 0);
      cc.z[j] = const_cast<float*>(branches[j]); cc.L[j] = lengths[j]; cc.C4[j] = widths[j] / 4; cc.off4[j] = off / 4;
      off += widths[j];
      cc.start[j + 1] = cc.start[j] + (long long)B * (backward ? lengths[j] : T) * cc.C4[j];
    } else {
      cc.z[j] = nullptr; cc.L[j] = T; cc.C4[j] = 1; cc.off4[j] = 0; cc.start[j + 1] = cc.start[j];
    }
  }
  cc.Ct4 = off / 4;
  DLIP_CHECK_ARG((reinterpret_cast<uintptr_t>(cat) & 15) == 0);
  if (backward) hipLaunchKernelGGL(chomp_concat_kernel<true>, dim3(grid_for(cc.start[n_branches])), dim3(256), 0, ST(stream), cc, cat);
  else hipLaunchKernelGGL(chomp_concat_kernel<false>, dim3(grid_for(cc.start[n_branches])), dim3(256), 0, ST(stream), cc, cat);
  return dlip_launch_status();
}

extern "C" int dlip_mul_mask_f32(const float* x, const float* mask, float* y, int64_t n, float scale, dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && mask && y && n > 0);
  hipLaunchKernelGGL(mul_mask_kernel, dim3(grid_for(n)), dim3(256), 0, ST(stream), x, mask, y, scale, (long long)n);
  return dlip_launch_status();
}

// Train-mode kernels of the speech ENCODER (full train_audio.py training):
// BatchNorm over the B*T' rows of a TDNN layer in training mode fused with LeakyReLU, forward and
// backward (models/audio_models/tdnn.py:35-43 under autograd), MeanStdPooling backward
// (pooling.py:24-26), and the layout helpers of the convolution gradients (weight flip / permute,
// scaled split for the weight-gradient GEMM).  The convolution gradients themselves run on the
// implicit-GEMM kernels: dgrad = the forward kernel on flipped weights, wgrad = one GEMM per filter tap
// with the B*T axis as the reduction (deeplip_amd/autograd.py).
//
// Column reductions over M = B*T' ~ 2e4 rows: rows are cut into fixed chunks of 512, a workgroup
// reduces one (chunk, 64-column block) in fp64 (16 row groups x 16 lanes x float4, LDS tree), partials go
// to a caller-provided workspace and a second kernel adds them in chunk order: deterministic, no atomics.
#include "dlip_common.h"
#include "conv_common.h"   // FastDiv: exact division by launch constants

namespace {

constexpr int CHUNK_ROWS = 512;
constexpr int DLIP_LIFT_BCAST = 2048;   // include/deeplip_hip.h: DLIP_LIFT_WORDS = 2 + max(4096 workgroup maxima, this many copies of 2^-e)

__device__ __forceinline__ float lrelu(float v, float slope) { return v >= 0.f ? v : v * slope; }

// MODE 0: s0 = sum a, s1 = sum a^2 where a = act_first ? lrelu(x) : x            (BN forward statistics)
// MODE 1: s0 = sum g, s1 = sum g * xhat                                          (BN backward)
//         g = dy * (act_first ? 1 : lrelu'(bn(x))),  xhat = (a - mean) * invstd
// MODE 2: s0 = sum x, s1 unused                                                   (bias gradient)
// MODE 3: MODE 1 for y = prelu(bn(x)) with the per-channel slopes `slope_vec` (act_first = 0), plus
//         s2 = sum (bn(x) < 0 ? dy * bn(x) : 0) = the slope gradient, written as {s2, 0} pairs to a SECOND partial
//         region behind the first (part + parts * C * 2): BatchNorm + PReLU backward sums in one pass over dy and x
//
// (round 5) What follows the partial sums -- adding them per channel and forming mean / 1/std (MODE 0) or the parameter gradients
// (MODE 1 / 3) or the bias gradient (MODE 2) -- used to be a launch of its own (bn_fwd_finalize / col_finalize / col_finalize3:
// 128 launches of a lip-clip training step, 5 - 90 us each, 0.7 ms on the critical path: on the MS-TCN head's 1 MB tensors they
// cost more than the passes they finish).  With `fin.ticket` (this stream's self-resetting ticket words: the convolution split's
// workspace) the workgroup that finishes LAST for a 64-column block -- its ticket says so -- adds that block's partials in part
// order (4 lane groups x parts / 4, then the groups in order: a fixed association whichever workgroup it is) and writes the results.
// Rows per part grow beyond 512 so that a launch has at most 512 parts: the finisher reads them through ONE CU.
// (round 5) The gradient behind a MaxPool3d((1,3,3),(1,2,2),(0,1,1)) WITHOUT the full-resolution tensor: the BatchNorm + PReLU in
// front of the pooling (the stem: model.py:83-85) reads, per input pixel, the at most four windows that cover it -- their argmax bytes
// (dlip_maxpool3x3s2_idx_f32's codes) and pooled gradients -- instead of a 460 MB (B = 32) scatter that one launch writes and two
// passes read back.  idx == nullptr: `dy` is an ordinary dense gradient.
struct PoolSrc {
  const uint32_t* idx;               // [N, Ho, Wo, C / 4] argmax codes, one byte per channel
  const float* dy2;                  // nullptr, or a SECOND pooled gradient of the same output, added on the fly (the pooled tensor feeds the
                                     // first block's convolution AND its shortcut: autograd's own addition was one more 115 MB launch)
  int H, W, Ho, Wo;
  FastDiv div_W, div_H, div_Wo, div_Ho;   // (rows < 2^31: 32-bit index arithmetic, divisions by multiplication)
};

// g[row, c4 * 4 ..] for pixel row = (n H + h) W + w: the sum of the pooled gradients whose argmax is this pixel (maxpool_idx_bwd_kernel)
__device__ __forceinline__ f32x4 pooled_grad(const PoolSrc& ps, const float* __restrict__ dyp, int row, int c, int C) {
  const int t = dlip_div(row, ps.div_W), w = row - t * ps.W;
  const int n = dlip_div(t, ps.div_H), h = t - n * ps.H;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int ho = h / 2; ho <= (h + 1) / 2; ++ho) {
    if (ho >= ps.Ho) continue;
    const int r = h - (2 * ho - 1);
    for (int wo = w / 2; wo <= (w + 1) / 2; ++wo) {
      if (wo >= ps.Wo) continue;
      const uint32_t me = (uint32_t)(r * 3 + (w - (2 * wo - 1)));
      const long long o = ((long long)(n * ps.Ho + ho) * ps.Wo + wo) * C + c;
      const uint32_t code = ps.idx[o >> 2];
      f32x4 g = *reinterpret_cast<const f32x4*>(dyp + o);
      if (ps.dy2) {
        const f32x4 g2 = *reinterpret_cast<const f32x4*>(ps.dy2 + o);
#pragma unroll
        for (int k = 0; k < 4; ++k) g[k] += g2[k];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (((code >> (8 * k)) & 0xFFu) == me) acc[k] += g[k];
    }
  }
  return acc;
}

// (ABI 48 / 49) The gradient of a MeanStdPooling's INPUT formed on load: dy[b, t, c] = dmean[b, c] / T + dstd[b, c] (y[b, t, c] - mean[b, c]) / ((T - 1) std[b, c])
// (meanstd_bwd_kernel's expression) is A[b, c] + K[b, c] y with y the ACTIVATED value the BatchNorm backward recomputes anyway -- the pooling's
// backward then writes nothing: its [B,T,C] gradient (460 MB for the E-TDNN's last layer at B = 256) was one write and two reads.  The two
// coefficients per (utterance, channel) come from ms_coef_kernel (one FMA per value: with the divisions per value the passes were ALU-bound,
// 461 us for a producer launch that moves 1.4 GB).
struct MsSrc {
  const float* coef = nullptr;       // [B, 2 C] (A | K); nullptr: the gradient is read from memory
  int T = 1;                         // frames per utterance: row r belongs to utterance r / T
  FastDiv div_T;
};
__device__ __forceinline__ f32x4 ms_grad(const MsSrc& ms, int row, int c, int C, const f32x4 yact) {
  const int b = dlip_div(row, ms.div_T);
  const float* cb = ms.coef + (long long)b * 2 * C + c;
  const f32x4 A = *reinterpret_cast<const f32x4*>(cb), K = *reinterpret_cast<const f32x4*>(cb + C);
  f32x4 o;
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = fmaf(K[k], yact[k], A[k]);
  return o;
}
// coef[b, c] = A = dmean / T - K mean, coef[b, C + c] = K = dstd / ((T - 1) std) (0 where std == 0: no gradient through it)
__global__ __launch_bounds__(256) void ms_coef_kernel(const float* __restrict__ y, const float* __restrict__ g, float* __restrict__ coef, int B,
                                                      int C, int T) {
  const long long n = (long long)B * C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const long long b = i / C;
    const int c = (int)(i - b * C);
    const float mean = y[b * 2 * C + c], sd = y[b * 2 * C + C + c];
    const float gm = g[b * 2 * C + c], gsd = g[b * 2 * C + C + c];
    const float K = sd > 0.f ? gsd / ((float)(T - 1) * sd) : 0.f;
    coef[b * 2 * C + c] = gm / (float)T - K * mean;
    coef[b * 2 * C + C + c] = K;
  }
}

struct ColFin {
  int* ticket;                       // nullptr: the caller launches the finalize kernel
  float* out0;                       // MODE 0 save_mean   | MODE 1 / 3 dbeta  | MODE 2 the column sums
  float* out1;                       // MODE 0 save_invstd | MODE 1 / 3 dgamma
  float* out2;                       // MODE 3 dslope
  float* running_mean;               // MODE 0 (nullable pair)
  float* running_var;
  long long* nbt;                    // MODE 0 (nullable): num_batches_tracked += 1
  float momentum, eps;
  unsigned* amax_parts = nullptr;    // MODE 1 (nullable): per workgroup (max |g|, max |xhat|) as bit patterns, 2 words each (dlip_bn_rows_train_bwd_sums_f32)
  MsSrc ms;                          // MODE 1, act_first == 0: dy formed on load from a pooled gradient (dlip_bn_rows_train_bwd_ms_f32)
};

__device__ __forceinline__ void bn_stats_finish(double s, double q, int M, int c, float* save_mean, float* save_invstd,
                                                float* running_mean, float* running_var, float momentum, float eps) {
  const double mean = s / M;
  double var = q / M - mean * mean;
  if (var < 0.0) var = 0.0;
  save_mean[c] = (float)mean;
  save_invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) {
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(M > 1 ? var * M / (M - 1) : var);
  }
}

// The calling workgroup has written its share of a result other workgroups' shares complete -- with publish() below: write-through
// stores, visible to the whole device once acknowledged -- and now takes a ticket on `word`; returns whether this workgroup took the
// LAST of `n` (it then re-zeroes the word for the next launch on this stream and drops its caches' stale lines: the others' shares
// are read with plain loads).  Every thread of the workgroup calls it; no workgroup waits for another.
// (NOT __threadfence(): an agent-scope release writes back the whole L2 -- issued by each of a pass's ~1 000 workgroups it doubled
// the training steps' time; the convolution kernels' split hand-off publishes the same way as this.)
template <typename T>
__device__ __forceinline__ void publish(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ bool last_arrival(int* word, int n) {
  __shared__ int s_last;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const int t = __hip_atomic_fetch_add(word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = t == n - 1;
    if (s_last) __hip_atomic_store(word, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  const bool last = s_last != 0;
  if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  return last;
}

// The tail of a column-reduction pass (col_partial_kernel, pool_bn_partial_kernel): with a ticket word, the workgroup that arrives
// LAST for its 64-column block adds the block's partials -- lane group g parts g, g + 4, ... (four loads in flight), then the groups in
// order -- and writes the results (MODE as in col_partial_kernel).  red: the caller's [16][64][2 | 3] LDS block, free by now.
template <int MODE>
__device__ __forceinline__ void col_finish(double* part, int parts, int M, int C, int c0, const ColFin& fin,
                                           double (*red)[64][MODE == 3 ? 3 : 2]) {
  if (fin.ticket == nullptr) return;
  if (!last_arrival(fin.ticket + blockIdx.x, parts)) return;
  const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int cc = c0 + col;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0;
  if (cc < C) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    const double* pa = part + (long long)cc * 2;
    const double* pb = pa + (long long)parts * C * 2;
    const long long step = (long long)C * 2;
    int i = grp;
    for (; i + 12 < parts; i += 16) {
      d2 v[4], w[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        v[u] = *reinterpret_cast<const d2*>(pa + (i + 4 * u) * step);
        if (MODE == 3) w[u] = *reinterpret_cast<const d2*>(pb + (i + 4 * u) * step);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) { a0 += v[u][0]; a1 += v[u][1]; if (MODE == 3) a2 += w[u][0]; }
    }
    for (; i < parts; i += 4) {
      a0 += pa[i * step]; a1 += pa[i * step + 1];
      if (MODE == 3) a2 += pb[i * step];
    }
  }
  __syncthreads();                                   // (red is free: every thread passed last_arrival's barriers)
  red[grp][col][0] = a0; red[grp][col][1] = a1;
  if (MODE == 3) red[grp][col][2] = a2;
  __syncthreads();
  if (grp != 0 || cc >= C) return;
  double s = 0.0, q = 0.0, t = 0.0;
#pragma unroll
  for (int i = 0; i < 4; ++i) { s += red[i][col][0]; q += red[i][col][1]; if (MODE == 3) t += red[i][col][2]; }
  if (MODE == 0) {
    bn_stats_finish(s, q, M, cc, fin.out0, fin.out1, fin.running_mean, fin.running_var, fin.momentum, fin.eps);
    if (fin.nbt && cc == 0) fin.nbt[0] += 1;
  } else {
    if (fin.out0) fin.out0[cc] = (float)s;
    if (MODE != 2 && fin.out1) fin.out1[cc] = (float)q;
    if (MODE == 3) fin.out2[cc] = (float)t;
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void col_partial_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          double* part, int M, int C, float slope, int act_first,
                                                          const float* __restrict__ slope_vec, int rows_per_part, const ColFin fin) {
  __shared__ double red[16][64][MODE == 3 ? 3 : 2];
  const int c0 = blockIdx.x * 64, chunk = blockIdx.y;
  const int lx = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int c = c0 + lx * 4;
  const int r0 = chunk * rows_per_part, r1 = min(M, r0 + rows_per_part);
  double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  float gmax = 0.f, hmax = 0.f;      // MODE 1: the largest |g| and |xhat| this lane saw (fin.amax_parts: the bound behind the fused backward's lift)
  if (c < C) {   // C % 4 == 0
    f32x4 mu = {0, 0, 0, 0}, is = {0, 0, 0, 0}, ga = {0, 0, 0, 0}, be = {0, 0, 0, 0}, sl = {0, 0, 0, 0};
    if (MODE == 1 || MODE == 3) {
      mu = *reinterpret_cast<const f32x4*>(mean + c); is = *reinterpret_cast<const f32x4*>(invstd + c);
      ga = *reinterpret_cast<const f32x4*>(gamma + c); be = *reinterpret_cast<const f32x4*>(beta + c);
    }
    if (MODE == 3) sl = *reinterpret_cast<const f32x4*>(slope_vec + c);
    auto add_row = [&](const f32x4 xv, f32x4 gv, const int row) __attribute__((always_inline)) {
      if (MODE == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const double a = act_first ? lrelu(xv[k], slope) : xv[k];
          s0[k] += a; s1[k] += a * a;
        }
      } else if (MODE == 1) {
        if (fin.ms.coef != nullptr) {      // (workgroup-uniform; act_first == 0) the gradient from the pooled one, at the activated value
          f32x4 ya;
#pragma unroll
          for (int k = 0; k < 4; ++k) ya[k] = lrelu((xv[k] - mu[k]) * is[k] * ga[k] + be[k], slope);
          gv = ms_grad(fin.ms, row, c, C, ya);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float a = act_first ? lrelu(xv[k], slope) : xv[k];
          const float xh = (a - mu[k]) * is[k];
          float g = gv[k];
          if (!act_first) g *= (xh * ga[k] + be[k]) >= 0.f ? 1.f : slope;
          s0[k] += (double)g; s1[k] += (double)g * (double)xh;
          gmax = fmaxf(gmax, fabsf(g)); hmax = fmaxf(hmax, fabsf(xh));
        }
      } else if (MODE == 3) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float xh = (xv[k] - mu[k]) * is[k];
          const float bn = xh * ga[k] + be[k];
          float g = gv[k];
          if (bn < 0.f) { s2[k] += (double)g * (double)bn; g *= sl[k]; }
          s0[k] += (double)g; s1[k] += (double)g * (double)xh;
        }
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) s0[k] += (double)xv[k];
      }
    };
    // four rows of a lane group per trip, their loads issued together (one load in flight per lane held a pass at half of what
    // the memory system gives once a part is longer than a few trips)
    constexpr bool TWO_ = MODE == 1 || MODE == 3;
    const bool TWO = TWO_ && !(MODE == 1 && fin.ms.coef != nullptr);      // (a gradient formed on load is not read)
    int r = r0 + rg;
    for (; r + 48 < r1; r += 64) {
      f32x4 xv[4], gv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        xv[u] = *reinterpret_cast<const f32x4*>(x + (long long)(r + 16 * u) * C + c);
        gv[u] = TWO ? *reinterpret_cast<const f32x4*>(dy + (long long)(r + 16 * u) * C + c) : xv[u];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) add_row(xv[u], gv[u], r + 16 * u);
    }
    for (; r < r1; r += 16) {
      const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (long long)r * C + c);
      const f32x4 gv = TWO ? *reinterpret_cast<const f32x4*>(dy + (long long)r * C + c) : xv;
      add_row(xv, gv, r);
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    red[rg][lx * 4 + k][0] = s0[k]; red[rg][lx * 4 + k][1] = s1[k];
    if (MODE == 3) red[rg][lx * 4 + k][2] = s2[k];
  }
  __syncthreads();
  const int parts = gridDim.y;
  if (threadIdx.x < 64 && c0 + (int)threadIdx.x < C) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      a0 += red[i][threadIdx.x][0]; a1 += red[i][threadIdx.x][1];
      if (MODE == 3) a2 += red[i][threadIdx.x][2];
    }
    double* p = part + ((long long)chunk * C + c0 + threadIdx.x) * 2;
    publish(p, a0); publish(p + 1, a1);
    if (MODE == 3) {
      double* p2 = p + (long long)parts * C * 2;
      publish(p2, a2); publish(p2 + 1, 0.0);
    }
  }
  if (MODE == 1 && fin.amax_parts != nullptr) {       // (workgroup-uniform)
    __shared__ float mx[4][2];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { gmax = fmaxf(gmax, __shfl_xor(gmax, off)); hmax = fmaxf(hmax, __shfl_xor(hmax, off)); }
    if ((threadIdx.x & 63) == 0) { mx[threadIdx.x >> 6][0] = gmax; mx[threadIdx.x >> 6][1] = hmax; }
    __syncthreads();
    if (threadIdx.x < 2) {
      float m = fmaxf(fmaxf(mx[0][threadIdx.x], mx[1][threadIdx.x]), fmaxf(mx[2][threadIdx.x], mx[3][threadIdx.x]));
      if (!(m == m)) m = 3.4e38f;
      publish(fin.amax_parts + 2 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x) + threadIdx.x, __float_as_uint(m));
    }
  }
  col_finish<MODE>(part, parts, M, C, c0, fin, red);
}

// The lift of a BatchNorm backward's dx WITHOUT dx: |dx| = |gamma invstd (g - mean(g) - xhat mean(g xhat))| <= gamma invstd max|g| (2 + max|xhat|)
// (|mean(g xhat)| <= sqrt(mean g^2) sqrt(mean xhat^2) <= max|g|: the batch statistics make mean xhat^2 = 1).  parts: n pairs
// (max |g|, max |xhat|) from col_partial_kernel<1>.  out: a DLIP_LIFT_WORDS buffer as pow2_finalize_parts writes it, the exponent chosen
// so that the BOUND sits at `target`: the true maximum then lies up to (2 + max|xhat|) -- an order of magnitude -- below it, well inside
// the split format (a power-of-two lift is exact: another exponent, the same gradients).
__global__ __launch_bounds__(256) void bn_bwd_lift_bound_kernel(const unsigned* __restrict__ parts, int n, const float* __restrict__ gamma,
                                                                const float* __restrict__ invstd, int C, float* __restrict__ out, float target) {
  __shared__ float red[4][3];
  float g = 0.f, h = 0.f, gi = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) { g = fmaxf(g, __uint_as_float(parts[2 * i])); h = fmaxf(h, __uint_as_float(parts[2 * i + 1])); }
  for (int c = threadIdx.x; c < C; c += 256) gi = fmaxf(gi, fabsf(gamma[c] * invstd[c]));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { g = fmaxf(g, __shfl_xor(g, off)); h = fmaxf(h, __shfl_xor(h, off)); gi = fmaxf(gi, __shfl_xor(gi, off)); }
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = g; red[threadIdx.x >> 6][1] = h; red[threadIdx.x >> 6][2] = gi; }
  __syncthreads();
  g = fmaxf(fmaxf(red[0][0], red[1][0]), fmaxf(red[2][0], red[3][0]));
  h = fmaxf(fmaxf(red[0][1], red[1][1]), fmaxf(red[2][1], red[3][1]));
  gi = fmaxf(fmaxf(red[0][2], red[1][2]), fmaxf(red[2][2], red[3][2]));
  const float bound = gi * g * (2.f + h);
  float s = 1.f;
  if (bound > 0.f && bound < 3.0e38f) s = exp2f(floorf(log2f(target / bound)));
  if (!(s > 0.f) || s > 1.0e30f) s = 1.0e30f;
  if (threadIdx.x == 0) { out[0] = s; out[1] = 1.f / s; }
  for (int i = threadIdx.x; i < DLIP_LIFT_BCAST; i += 256) out[2 + i] = 1.f / s;
}

// The BatchNorm + PReLU backward sums (col_partial_kernel<3>) BEHIND A MAX-POOL, taken over the POOLED rows: the gradient behind
// the pooling is zero except at each window's argmax, so the three sums have one term per pooled element -- its gradient g, and x at
// the pixel its argmax code names (a 4-byte gather per channel) -- instead of one per input element whose g is mostly an exact zero:
// a quarter of the rows, and the full-resolution gradient tensor never exists.  Partials, parts and the finisher as col_partial_kernel
// (rows = pooled pixels n Ho Wo).  (A pixel that is the argmax of two windows contributes g1 xhat + g2 xhat here and (g1 + g2) xhat
// in the dense form: the same to fp64 rounding.)
__global__ __launch_bounds__(256) void pool_bn_partial_kernel(const float* __restrict__ x, const float* __restrict__ dyp, const PoolSrc ps,
                                                              const float* __restrict__ mean, const float* __restrict__ invstd,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              const float* __restrict__ slope_vec, double* part, int Mp, int M, int C,
                                                              int rows_per_part, const ColFin fin) {
  __shared__ double red[16][64][3];
  const int c0 = blockIdx.x * 64, chunk = blockIdx.y;
  const int lx = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int c = c0 + lx * 4;
  const int r0 = chunk * rows_per_part, r1 = min(Mp, r0 + rows_per_part);
  double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  if (c < C) {
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), is = *reinterpret_cast<const f32x4*>(invstd + c);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c), be = *reinterpret_cast<const f32x4*>(beta + c);
    const f32x4 sl = *reinterpret_cast<const f32x4*>(slope_vec + c);
    for (int rb = r0 + rg; rb < r1; rb += 32) {           // two pooled rows of this lane group per trip: their gathers overlap
      f32x4 g[2];
      float xv[2][4];
      bool on[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int r = rb + 16 * u;
        on[u] = r < r1;
        g[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) xv[u][k] = 0.f;
        if (on[u]) {
          const int t = dlip_div(r, ps.div_Wo), wo = r - t * ps.Wo;
          const int n = dlip_div(t, ps.div_Ho), ho = t - n * ps.Ho;
          const long long o = (long long)r * C + c;
          const uint32_t code = ps.idx[o >> 2];
          g[u] = *reinterpret_cast<const f32x4*>(dyp + o);
          if (ps.dy2) {
            const f32x4 g2 = *reinterpret_cast<const f32x4*>(ps.dy2 + o);
#pragma unroll
            for (int k = 0; k < 4; ++k) g[u][k] += g2[k];
          }
          const float* xn = x + ((long long)n * ps.H * ps.W) * C + c;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int tap = (int)((code >> (8 * k)) & 0xFFu);          // r * 3 + s of the window (2 ho - 1 .., 2 wo - 1 ..)
            const int tr = tap / 3, ts = tap - 3 * tr;
            xv[u][k] = xn[((long long)(2 * ho - 1 + tr) * ps.W + (2 * wo - 1 + ts)) * C + k];
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (!on[u]) continue;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float xh = (xv[u][k] - mu[k]) * is[k];
          const float bn = xh * ga[k] + be[k];
          float gg = g[u][k];
          if (bn < 0.f) { s2[k] += (double)gg * (double)bn; gg *= sl[k]; }
          s0[k] += (double)gg; s1[k] += (double)gg * (double)xh;
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) { red[rg][lx * 4 + k][0] = s0[k]; red[rg][lx * 4 + k][1] = s1[k]; red[rg][lx * 4 + k][2] = s2[k]; }
  __syncthreads();
  const int parts = gridDim.y;
  if (threadIdx.x < 64 && c0 + (int)threadIdx.x < C) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) { a0 += red[i][threadIdx.x][0]; a1 += red[i][threadIdx.x][1]; a2 += red[i][threadIdx.x][2]; }
    double* p = part + ((long long)chunk * C + c0 + threadIdx.x) * 2;
    publish(p, a0); publish(p + 1, a1);
    double* p2 = p + (long long)parts * C * 2;
    publish(p2, a2); publish(p2 + 1, 0.0);
  }
  col_finish<3>(part, parts, M, C, c0, fin, red);
}

// ---- (round 5) the END of a BasicBlock under model.train(): out = prelu(bn2(conv2) + residual) (resnet.py:62-69) ------------------
// Forward (after the statistics of x = conv2's output): one pass reads x and the residual and writes the sum (kept for the backward)
// and the output -- bn2's output is never stored (was: written by the BatchNorm apply pass, read back by the add + PReLU pass).
template <bool FIXED>
__global__ __launch_bounds__(256) void bn_add_prelu_fwd_kernel(const f32x4* __restrict__ x, const f32x4* __restrict__ res,
                                                               const float* __restrict__ mean, const float* __restrict__ invstd,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               const float* __restrict__ slope_vec, f32x4* __restrict__ s_out,
                                                               f32x4* __restrict__ y, long long n4, int C4) {
  const long long i0 = (long long)blockIdx.x * 256 + threadIdx.x;
  f32x4 mu, is, ga, be, sl;
  auto load = [&](int c) {
    mu = *reinterpret_cast<const f32x4*>(mean + c); is = *reinterpret_cast<const f32x4*>(invstd + c);
    ga = *reinterpret_cast<const f32x4*>(gamma + c); be = *reinterpret_cast<const f32x4*>(beta + c);
    sl = *reinterpret_cast<const f32x4*>(slope_vec + c);
  };
  if (FIXED) load((int)(i0 % C4) * 4);
  for (long long i = i0; i < n4; i += (long long)gridDim.x * 256) {
    if (!FIXED) load((int)(i % C4) * 4);
    const f32x4 v = x[i], r = res[i];
    f32x4 sv, o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float h = (v[k] - mu[k]) * is[k] * ga[k] + be[k];     // bn_fwd_apply_kernel's expression (slope 1)
      sv[k] = h + r[k];                                            // add_prelu_fwd_kernel's
      o[k] = sv[k] >= 0.f ? sv[k] : sv[k] * sl[k];
    }
    s_out[i] = sv;
    y[i] = o;
  }
}

// Backward, first pass: the PReLU's backward, its slope-gradient terms AND the sums of bn2's backward from one read of the incoming
// gradient(s), the kept sum and x -- was four passes (prelu_bwd writing the gradient and a tensor of slope terms, a column sum of the
// terms, the BatchNorm's sums).  dy2 (nullable): a second gradient of the same output, added on the fly (the block's output feeds the
// next block's first convolution AND its shortcut: autograd's own addition of the two was a 345 MB launch per layer-1 block).
// g = the gradient behind the PReLU = the shortcut's gradient AND bn2's incoming one: written once (g_out).  Partials {sum g,
// sum g xhat | sum slope terms}, parts and finisher as col_partial_kernel<3> (dbeta, dgamma, dslope).
__global__ __launch_bounds__(256) void add_prelu_bn_partial_kernel(const float* __restrict__ dy, const float* __restrict__ dy2,
                                                                   const float* __restrict__ sum, const float* __restrict__ x,
                                                                   const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                   const float* __restrict__ slope_vec, float* __restrict__ g_out,
                                                                   double* part, int M, int C, int rows_per_part, const ColFin fin) {
  __shared__ double red[16][64][3];
  const int c0 = blockIdx.x * 64, chunk = blockIdx.y;
  const int lx = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int c = c0 + lx * 4;
  const int r0 = chunk * rows_per_part, r1 = min(M, r0 + rows_per_part);
  double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  if (c < C) {
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), is = *reinterpret_cast<const f32x4*>(invstd + c);
    const f32x4 sl = *reinterpret_cast<const f32x4*>(slope_vec + c);
    const bool two = dy2 != nullptr;                         // (launch-uniform)
    auto one = [&](long long o, const f32x4 d, const f32x4 sv, const f32x4 xv) __attribute__((always_inline)) {
      f32x4 g;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool neg = !(sv[k] >= 0.f);
        g[k] = neg ? d[k] * sl[k] : d[k];                    // prelu_bwd_kernel
        const float t = neg ? d[k] * sv[k] : 0.f;
        const float xh = (xv[k] - mu[k]) * is[k];            // col_partial_kernel<1> at slope 1
        s0[k] += (double)g[k]; s1[k] += (double)g[k] * (double)xh; s2[k] += (double)t;
      }
      *reinterpret_cast<f32x4*>(g_out + o) = g;
    };
    int r = r0 + rg;
    for (; r + 16 < r1; r += 32) {                           // two rows of a lane group per trip, their loads issued together
      const long long o0 = (long long)r * C + c, o1 = o0 + (long long)16 * C;
      f32x4 d0 = *reinterpret_cast<const f32x4*>(dy + o0), d1 = *reinterpret_cast<const f32x4*>(dy + o1);
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(sum + o0), v1 = *reinterpret_cast<const f32x4*>(sum + o1);
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(x + o0), x1 = *reinterpret_cast<const f32x4*>(x + o1);
      if (two) {
        const f32x4 e0 = *reinterpret_cast<const f32x4*>(dy2 + o0), e1 = *reinterpret_cast<const f32x4*>(dy2 + o1);
#pragma unroll
        for (int k = 0; k < 4; ++k) { d0[k] += e0[k]; d1[k] += e1[k]; }
      }
      one(o0, d0, v0, x0);
      one(o1, d1, v1, x1);
    }
    for (; r < r1; r += 16) {
      const long long o = (long long)r * C + c;
      f32x4 d = *reinterpret_cast<const f32x4*>(dy + o);
      if (two) { const f32x4 e = *reinterpret_cast<const f32x4*>(dy2 + o);
#pragma unroll
        for (int k = 0; k < 4; ++k) d[k] += e[k]; }
      one(o, d, *reinterpret_cast<const f32x4*>(sum + o), *reinterpret_cast<const f32x4*>(x + o));
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) { red[rg][lx * 4 + k][0] = s0[k]; red[rg][lx * 4 + k][1] = s1[k]; red[rg][lx * 4 + k][2] = s2[k]; }
  __syncthreads();
  const int parts = gridDim.y;
  if (threadIdx.x < 64 && c0 + (int)threadIdx.x < C) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) { a0 += red[i][threadIdx.x][0]; a1 += red[i][threadIdx.x][1]; a2 += red[i][threadIdx.x][2]; }
    double* p = part + ((long long)chunk * C + c0 + threadIdx.x) * 2;
    publish(p, a0); publish(p + 1, a1);
    double* p2 = p + (long long)parts * C * 2;
    publish(p2, a2); publish(p2 + 1, 0.0);
  }
  col_finish<3>(part, parts, M, C, c0, fin, red);
}

// forward statistics: mean, biased variance -> invstd; running stats with the unbiased variance (torch)
// Sum of the per-chunk partials {s, q} of 16 channels by one 256-thread workgroup: thread (channel cl = tid & 15, lane group
// g = tid >> 4) adds chunks g, g + 16, ... in order, the 16 groups meet in LDS and are added in group order -- a fixed
// association (deterministic bits) with a dependent chain of chunks / 16 loads.  (One thread per channel walking all chunks
// serially -- 877 of them for layer 1's M = 449 k rows -- took 40-50 us per call, 7 ms of a 61 ms training step.)
__device__ __forceinline__ bool col_reduce16(const double* __restrict__ part, int C, int chunks, double& s, double& q) {
  __shared__ double red[16][16][2];
  const int cl = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  double a = 0.0, b = 0.0;
  if (c < C) {
    // (four loads in flight, added in the same order as one at a time: the ~1 000 half-tile rows a convolution's statistics
    // epilogue leaves took 25 us per TDNN layer as a chain of dependent loads)
    typedef double d2 __attribute__((ext_vector_type(2)));
    const long long step = (long long)C * 2;
    const double* p = part + (long long)c * 2;
    int i = g;
    for (; i + 48 < chunks; i += 64) {
      d2 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const d2*>(p + (i + 16 * u) * step);
#pragma unroll
      for (int u = 0; u < 4; ++u) { a += v[u][0]; b += v[u][1]; }
    }
    for (; i < chunks; i += 16) { a += p[i * step]; b += p[i * step + 1]; }
  }
  red[g][cl][0] = a; red[g][cl][1] = b;
  __syncthreads();
  if (g != 0 || c >= C) return false;
  s = 0.0; q = 0.0;
#pragma unroll
  for (int i = 0; i < 16; ++i) { s += red[i][cl][0]; q += red[i][cl][1]; }
  return true;
}

__global__ __launch_bounds__(256) void bn_fwd_finalize_kernel(const double* __restrict__ part, float* __restrict__ save_mean,
                                                              float* __restrict__ save_invstd, float* __restrict__ running_mean,
                                                              float* __restrict__ running_var, int M, int C, int chunks,
                                                              float momentum, float eps, long long* __restrict__ nbt) {
  double s, q;
  if (!col_reduce16(part, C, chunks, s, q)) return;
  const int c = blockIdx.x * 16 + (threadIdx.x & 15);
  bn_stats_finish(s, q, M, c, save_mean, save_invstd, running_mean, running_var, momentum, eps);
  if (nbt && c == 0) nbt[0] += 1;
}

// FIXED: the launch's stride (gridDim.x * 256) is a multiple of C4, so a thread meets ONE channel group in every iteration and
// keeps its per-channel parameters in registers (loading 4-6 parameter vectors per element from memory held these
// HBM-bound passes at ~2.1 TB/s: 34 % of a speech-encoder training step).
template <bool FIXED>
__global__ __launch_bounds__(256) void bn_fwd_apply_kernel(const f32x4* __restrict__ x, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, f32x4* __restrict__ y, long long n4,
                                                           int C4, float slope, int act_first,
                                                           const float* __restrict__ slope_vec = nullptr) {
  const long long i0 = (long long)blockIdx.x * 256 + threadIdx.x;
  f32x4 mu, is, ga, be, sl = {slope, slope, slope, slope};
  auto load = [&](int c) {
    mu = *reinterpret_cast<const f32x4*>(mean + c); is = *reinterpret_cast<const f32x4*>(invstd + c);
    ga = *reinterpret_cast<const f32x4*>(gamma + c); be = *reinterpret_cast<const f32x4*>(beta + c);
    if (slope_vec) sl = *reinterpret_cast<const f32x4*>(slope_vec + c);     // PReLU behind the BatchNorm (act_first = 0)
  };
  if (FIXED) load((int)(i0 % C4) * 4);
  for (long long i = i0; i < n4; i += (long long)gridDim.x * 256) {
    if (!FIXED) load((int)(i % C4) * 4);
    const f32x4 v = x[i];
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (act_first) o[k] = (lrelu(v[k], slope) - mu[k]) * is[k] * ga[k] + be[k];
      else o[k] = lrelu((v[k] - mu[k]) * is[k] * ga[k] + be[k], sl[k]);
    }
    y[i] = o;
  }
}

// y = maxpool3x3s2(prelu(bn(x))) with the argmax codes of dlip_maxpool3x3s2_idx_f32, from the batch statistics mean / invstd: the stem's
// BatchNorm3d + PReLU + MaxPool3d (model.py:83-85) under model.train() in ONE pass over the convolution's output -- the normalised
// full-resolution tensor (460 MB at B = 32) is never written (was: written by the apply pass, read back by the pooling).
__global__ __launch_bounds__(256) void bn_prelu_maxpool_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                                   const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta, const float* __restrict__ slope_vec,
                                                                   f32x4* __restrict__ y, uint32_t* __restrict__ idx, int H, int W,
                                                                   int Ho, int Wo, int C4, long long n4) {
  const int C = C4 * 4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C4) * 4;
    const long long p = i / C4;
    const int wo = (int)(p % Wo);
    const long long t = p / Wo;
    const int ho = (int)(t % Ho);
    const long long n = t / Ho;
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), is = *reinterpret_cast<const f32x4*>(invstd + c);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c), be = *reinterpret_cast<const f32x4*>(beta + c);
    const f32x4 sl = *reinterpret_cast<const f32x4*>(slope_vec + c);
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
        for (int k = 0; k < 4; ++k) {
          const float a = lrelu((v[k] - mu[k]) * is[k] * ga[k] + be[k], sl[k]);   // bn_fwd_apply_kernel's expression
          if (a > best[k] || ((code >> (8 * k)) & 0xFFu) == 0xFFu) {
            best[k] = a;
            code = (code & ~(0xFFu << (8 * k))) | ((uint32_t)(r * 3 + s_) << (8 * k));
          }
        }
      }
    }
    y[i] = best;
    idx[i] = code;
  }
}

__global__ __launch_bounds__(256) void col_finalize_kernel(const double* __restrict__ part, float* __restrict__ out0,
                                                           float* __restrict__ out1, int C, int chunks) {
  double s, q;
  if (!col_reduce16(part, C, chunks, s, q)) return;
  const int c = blockIdx.x * 16 + (threadIdx.x & 15);
  if (out0) out0[c] = (float)s;
  if (out1) out1[c] = (float)q;
}

// The three column sums of the BatchNorm + PReLU backward (dbeta, dgamma from region A's pairs, dslope from region B's) in ONE launch.
__global__ __launch_bounds__(256) void col_finalize3_kernel(const double* __restrict__ part_a, const double* __restrict__ part_b,
                                                            float* __restrict__ out0, float* __restrict__ out1, float* __restrict__ out2,
                                                            int C, int chunks) {
  __shared__ double red[16][16][3];
  const int cl = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  double a = 0.0, b = 0.0, d = 0.0;
  if (c < C)
    for (int i = g; i < chunks; i += 16) {
      a += part_a[((long long)i * C + c) * 2]; b += part_a[((long long)i * C + c) * 2 + 1]; d += part_b[((long long)i * C + c) * 2];
    }
  red[g][cl][0] = a; red[g][cl][1] = b; red[g][cl][2] = d;
  __syncthreads();
  if (g != 0 || c >= C) return;
  double s = 0.0, q = 0.0, r = 0.0;
#pragma unroll
  for (int i = 0; i < 16; ++i) { s += red[i][cl][0]; q += red[i][cl][1]; r += red[i][cl][2]; }
  out0[c] = (float)s; out1[c] = (float)q; out2[c] = (float)r;
}

// out[0] = 2^floor(log2(target / max(parts))), out[1] = 1 / out[0]: the per-workgroup maxima of a producer pass (non-negative floats
// as bit patterns) reduced by ONE workgroup (every thread of it calls): the kernel of that name, or the last workgroup of the
// producer itself.  `parts` may be out + 2 (the maxima live where the broadcast copies go: all read before the first is written).
__device__ __forceinline__ void pow2_finalize_parts(const unsigned* parts, int n, float* out, float target) {
  __shared__ unsigned red[4];
  unsigned m = 0u;
  for (int i = threadIdx.x; i < n; i += 256) m = max(m, parts[i]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  const float mx = __uint_as_float(max(max(red[0], red[1]), max(red[2], red[3])));
  float s = 1.f;
  if (mx > 0.f && mx < 3.0e38f) s = exp2f(floorf(log2f(target / mx)));
  if (!(s > 0.f) || s > 1.0e30f) s = 1.0e30f;
  if (threadIdx.x == 0) { out[0] = s; out[1] = 1.f / s; }
  // words 2 .. 2 + DLIP_LIFT_BCAST: 1 / s repeated -- the per-output-channel post_scale vector of the convolution that consumes the
  // lifted gradient (its epilogue takes a vector), without a fill launch per convolution.  (The maxima lived there: all read above.)
  for (int i = threadIdx.x; i < DLIP_LIFT_BCAST; i += 256) out[2 + i] = 1.f / s;
}

// dx = gamma * invstd * (g - dbeta / M - xhat * dgamma / M), times lrelu'(x) when the activation came first
template <bool FIXED, bool POOL = false>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const f32x4* __restrict__ dy, const f32x4* __restrict__ x,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                                           f32x4* __restrict__ dx, long long n4, int C4, int M, float slope,
                                                           int act_first, const float* __restrict__ slope_vec = nullptr,
                                                           unsigned* amax_acc = nullptr, int* ticket = nullptr,
                                                           const PoolSrc ps = PoolSrc{}, const MsSrc ms = MsSrc{}) {
  const float invM = 1.f / (float)M;
  float amax = 0.f;     // max |dx| of the launch -> amax_acc (the next convolution backward's power-of-two lift, without its own pass)
  const long long i0 = (long long)blockIdx.x * 256 + threadIdx.x;
  f32x4 mu, is, ga, be, dg, db, sl = {slope, slope, slope, slope};
  auto load = [&](int c) {
    mu = *reinterpret_cast<const f32x4*>(mean + c); is = *reinterpret_cast<const f32x4*>(invstd + c);
    ga = *reinterpret_cast<const f32x4*>(gamma + c); be = *reinterpret_cast<const f32x4*>(beta + c);
    dg = *reinterpret_cast<const f32x4*>(dgamma + c); db = *reinterpret_cast<const f32x4*>(dbeta + c);
    if (slope_vec) sl = *reinterpret_cast<const f32x4*>(slope_vec + c);
  };
  if (FIXED) load((int)(i0 % C4) * 4);
  for (long long i = i0; i < n4; i += (long long)gridDim.x * 256) {
    if (!FIXED) load((int)(i % C4) * 4);
    const f32x4 xv = x[i];
    f32x4 gv;
    if constexpr (POOL) { const int row = (int)(i / C4); gv = pooled_grad(ps, reinterpret_cast<const float*>(dy), row, (int)(i - (long long)row * C4) * 4, C4 * 4); }
    else if (ms.coef != nullptr) {       // (launch-uniform; act_first == 0) see MsSrc
      const int row = (int)(i / C4);
      f32x4 ya;
#pragma unroll
      for (int k = 0; k < 4; ++k) ya[k] = lrelu((xv[k] - mu[k]) * is[k] * ga[k] + be[k], slope);
      gv = ms_grad(ms, row, (int)(i - (long long)row * C4) * 4, C4 * 4, ya);
    }
    else gv = dy[i];
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float a = act_first ? lrelu(xv[k], slope) : xv[k];
      const float xh = (a - mu[k]) * is[k];
      float g = gv[k];
      if (!act_first) g *= (xh * ga[k] + be[k]) >= 0.f ? 1.f : sl[k];
      float d = ga[k] * is[k] * (g - db[k] * invM - xh * dg[k] * invM);
      if (act_first) d *= xv[k] >= 0.f ? 1.f : slope;
      o[k] = d;
      amax = fmaxf(amax, fabsf(d));
    }
    dx[i] = o;
  }
  if (amax_acc) {
    __shared__ float amax_red[4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
    if ((threadIdx.x & 63) == 0) amax_red[threadIdx.x >> 6] = amax;
    __syncthreads();
    if (threadIdx.x == 0) {
      float m = fmaxf(fmaxf(amax_red[0], amax_red[1]), fmaxf(amax_red[2], amax_red[3]));
      if (!(m == m)) m = 3.4e38f;
      publish(amax_acc + blockIdx.x, __float_as_uint(m));   // one word per workgroup, reduced by pow2_finalize_parts: thousands of
    }                                                       // workgroups meeting in ONE atomic cost 14 us per launch (the hot-line effect)
    // (round 5) ... by the workgroup that arrives last (`ticket`), not by a launch of its own
    if (ticket && last_arrival(ticket, (int)gridDim.x))
      pow2_finalize_parts(amax_acc, (int)gridDim.x, reinterpret_cast<float*>(amax_acc) - 2, 1024.0f);
  }
}

// ---- (round 5) BatchNorm over FEW rows in one launch -------------------------------------------------------------------------------
// The MS-TCN head's 24 BatchNorm layers work on [B (T + pad), 256] ~ 1 - 2.5 MB (tcn.py:42-43 under model.train()): their statistics
// pass, finalize, apply pass (forward) and sums, finalize, apply, lift (backward) were 3 + 4 launches of 5 - 15 us each -- launch
// latency, not bytes: ~1 ms of an 18 ms lip-clip training step.  For M <= BN_SMALL_ROWS one workgroup owns FOUR channels: a lane
// takes rows lane, lane + 256, ... (at most 16: R of them, a template parameter) and issues ALL its loads at once -- one memory
// latency for the whole tensor, the values stay in registers --, the sums (fp64) meet by wave butterflies and a four-wave LDS step in
// a fixed order, and y / dx are formed from the registers: one read, one write, one launch.  Same formulas per value as the
// three-launch path; the association of the fp64 column sums differs (fixed here too: deterministic).
constexpr int BN_SMALL_ROWS = 4096;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// sums over the workgroup of NQ quantities x 4 channels held per lane: butterflies within a wave, then the four waves in wave order
template <int NQ>
__device__ __forceinline__ void block_sums(double (&s)[NQ][4], double (*red)[NQ][4]) {
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int k = 0; k < 4; ++k) s[q][k] = wave_sum(s[q][k]);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int k = 0; k < 4; ++k) red[wave][q][k] = s[q][k];
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int k = 0; k < 4; ++k) s[q][k] = ((red[0][q][k] + red[1][q][k]) + red[2][q][k]) + red[3][q][k];
}

template <int R>
__global__ __launch_bounds__(256) void bn_small_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ slope_vec,
                                                           float* __restrict__ y, float* __restrict__ save_mean,
                                                           float* __restrict__ save_invstd, float* __restrict__ running_mean,
                                                           float* __restrict__ running_var, long long* __restrict__ nbt, int M, int C,
                                                           float momentum, float eps, float slope, int act_first) {
  __shared__ double red[4][2][4];
  const int c = blockIdx.x * 4;                              // C % 4 == 0: every workgroup has four live channels
  f32x4 v[R];
#pragma unroll
  for (int u = 0; u < R; ++u) {
    const int r = threadIdx.x + 256 * u;
    v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (r < M) v[u] = *reinterpret_cast<const f32x4*>(x + (long long)r * C + c);
  }
  double s[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
  for (int u = 0; u < R; ++u) {
    if (threadIdx.x + 256 * u < M) {
#pragma unroll
      for (int k = 0; k < 4; ++k) { const double a = act_first ? lrelu(v[u][k], slope) : v[u][k]; s[0][k] += a; s[1][k] += a * a; }
    }
  }
  block_sums<2>(s, red);
  f32x4 mu, is;
#pragma unroll
  for (int k = 0; k < 4; ++k) {                              // every lane forms the same mean / 1/std; lane k writes channel k's
    const double mean = s[0][k] / M;
    double var = s[1][k] / M - mean * mean;
    if (var < 0.0) var = 0.0;
    mu[k] = (float)mean;
    is[k] = (float)(1.0 / sqrt(var + (double)eps));
    if ((int)threadIdx.x == k) {
      save_mean[c + k] = mu[k]; save_invstd[c + k] = is[k];
      if (running_mean) {
        running_mean[c + k] = (1.f - momentum) * running_mean[c + k] + momentum * (float)mean;
        running_var[c + k] = (1.f - momentum) * running_var[c + k] + momentum * (float)(M > 1 ? var * M / (M - 1) : var);
      }
    }
  }
  if (nbt && blockIdx.x == 0 && threadIdx.x == 0) nbt[0] += 1;
  const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c), be = *reinterpret_cast<const f32x4*>(beta + c);
  f32x4 sl = {slope, slope, slope, slope};
  if (slope_vec) sl = *reinterpret_cast<const f32x4*>(slope_vec + c);
#pragma unroll
  for (int u = 0; u < R; ++u) {
    const int r = threadIdx.x + 256 * u;
    if (r < M) {
      f32x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (act_first) o[k] = (lrelu(v[u][k], slope) - mu[k]) * is[k] * ga[k] + be[k];
        else o[k] = lrelu((v[u][k] - mu[k]) * is[k] * ga[k] + be[k], sl[k]);
      }
      *reinterpret_cast<f32x4*>(y + (long long)r * C + c) = o;
    }
  }
}

// Backward of the above in one launch: slope_vec != NULL = y = prelu(bn(x)) with per-channel slopes (dslope written), else the
// LeakyReLU forms of dlip_bn_rows_train_bwd_f32.  amax_acc / ticket: the lift of dx (one word per workgroup; the workgroup that
// arrives last -- at most C / 4 arrivals on the word -- forms the pair); amax_acc without a ticket leaves the words for
// pow2_finalize_parts_kernel.
template <int R>
__global__ __launch_bounds__(256) void bn_small_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ slope_vec, float* __restrict__ dx,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dslope,
                                                           int M, int C, float slope, int act_first, unsigned* amax_acc, int* ticket) {
  __shared__ double red[4][3][4];
  const int c = blockIdx.x * 4;
  const bool vec = slope_vec != nullptr;                     // (launch-uniform)
  f32x4 xv[R], gv[R];
#pragma unroll
  for (int u = 0; u < R; ++u) {
    const int r = threadIdx.x + 256 * u;
    xv[u] = gv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (r < M) {
      xv[u] = *reinterpret_cast<const f32x4*>(x + (long long)r * C + c);
      gv[u] = *reinterpret_cast<const f32x4*>(dy + (long long)r * C + c);
    }
  }
  const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), is = *reinterpret_cast<const f32x4*>(invstd + c);
  const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c), be = *reinterpret_cast<const f32x4*>(beta + c);
  f32x4 sl = {slope, slope, slope, slope};
  if (vec) sl = *reinterpret_cast<const f32x4*>(slope_vec + c);
  double s[3][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  // pass 1 (registers): g = dy through the activation, xhat; the three column sums.  g replaces dy and xhat replaces x in place.
#pragma unroll
  for (int u = 0; u < R; ++u) {
    if (threadIdx.x + 256 * u < M) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float g = gv[u][k], xh;
        if (vec) {                                           // col_partial_kernel<3>
          xh = (xv[u][k] - mu[k]) * is[k];
          const float bn = xh * ga[k] + be[k];
          if (bn < 0.f) { s[2][k] += (double)g * (double)bn; g *= sl[k]; }
        } else {                                             // col_partial_kernel<1>
          const float a = act_first ? lrelu(xv[u][k], slope) : xv[u][k];
          xh = (a - mu[k]) * is[k];
          if (!act_first) g *= (xh * ga[k] + be[k]) >= 0.f ? 1.f : slope;
        }
        s[0][k] += (double)g; s[1][k] += (double)g * (double)xh;
        gv[u][k] = g;
        if (!act_first) xv[u][k] = xh;                       // (act_first keeps x: its sign gates the result below)
      }
    }
  }
  block_sums<3>(s, red);
  f32x4 db, dg;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    db[k] = (float)s[0][k]; dg[k] = (float)s[1][k];
    if ((int)threadIdx.x == k) {
      dbeta[c + k] = db[k]; dgamma[c + k] = dg[k];
      if (vec) dslope[c + k] = (float)s[2][k];
    }
  }
  const float invM = 1.f / (float)M;
  float amax = 0.f;
#pragma unroll
  for (int u = 0; u < R; ++u) {
    const int r = threadIdx.x + 256 * u;
    if (r < M) {
      f32x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) {                          // bn_bwd_apply_kernel
        float xh = xv[u][k];
        if (act_first) xh = (lrelu(xv[u][k], slope) - mu[k]) * is[k];
        float d = ga[k] * is[k] * (gv[u][k] - db[k] * invM - xh * dg[k] * invM);
        if (act_first) d *= xv[u][k] >= 0.f ? 1.f : slope;
        o[k] = d;
        amax = fmaxf(amax, fabsf(d));
      }
      *reinterpret_cast<f32x4*>(dx + (long long)r * C + c) = o;
    }
  }
  if (amax_acc) {
    __shared__ float amax_red[4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
    if ((threadIdx.x & 63) == 0) amax_red[threadIdx.x >> 6] = amax;
    __syncthreads();
    if (threadIdx.x == 0) {
      float m = fmaxf(fmaxf(amax_red[0], amax_red[1]), fmaxf(amax_red[2], amax_red[3]));
      if (!(m == m)) m = 3.4e38f;
      publish(amax_acc + blockIdx.x, __float_as_uint(m));
    }
    if (ticket && last_arrival(ticket, (int)gridDim.x))
      pow2_finalize_parts(amax_acc, (int)gridDim.x, reinterpret_cast<float*>(amax_acc) - 2, 1024.0f);
  }
}

// MeanStdPooling backward: y = [mean_t x, std_t x] (unbiased), dx = dmean / T + dstd * (x - mean) / ((T - 1) * std).
// A thread keeps ONE channel quad (its mean, std and the two gradients in registers) and walks a chunk of frames: no index
// arithmetic per element (the first version divided a 64-bit element index twice per element and re-read its four parameters per
// channel: 529 us for the 854 MB of an E-TDNN step at B = 256 -- 1.6 TB/s; round 4).  Same operations per value, same bits.
// (ABI 48) BN: x is the raw convolution output z in front of a train-mode BatchNorm + LeakyReLU whose activated values were never stored
// (dlip_meanstd_pool_bn_f32): they are formed per loaded value again (the same expression, the same bits).
struct PoolBnB {
  const float* mean = nullptr;
  const float* invstd = nullptr;
  const float* gamma = nullptr;
  const float* beta = nullptr;
  float slope = 1.f;
};
template <bool BN = false>
__global__ __launch_bounds__(256) void meanstd_bwd_kernel(const f32x4* __restrict__ x, const float* __restrict__ y,
                                                          const float* __restrict__ dy, f32x4* __restrict__ dx, int T, int C4,
                                                          int tchunk, const PoolBnB bn = PoolBnB{}) {
  const int c4 = blockIdx.x * blockDim.x + threadIdx.x;   // (64, 128 or 256 threads: whichever wastes the fewest lanes on C4)
  if (c4 >= C4) return;
  const int b = blockIdx.z, t0 = blockIdx.y * tchunk, t1 = min(T, t0 + tchunk);
  const int C = C4 * 4, c = c4 * 4;
  const float* yb = y + (long long)b * 2 * C;
  const float* gb = dy + (long long)b * 2 * C;
  float mean[4], den[4], a[4], gs[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float sd = yb[C + c + k];
    mean[k] = yb[c + k];
    a[k] = gb[c + k] / (float)T;
    gs[k] = sd > 0.f ? gb[C + c + k] : 0.f;               // std == 0: no gradient through it (0 * anything / 1 below)
    den[k] = sd > 0.f ? (float)(T - 1) * sd : 1.f;
  }
  const long long base = (long long)b * T * C4 + c4;
  f32x4 mu = {0, 0, 0, 0}, is = mu, ga = mu, be = mu;
  if constexpr (BN) {
    mu = *reinterpret_cast<const f32x4*>(bn.mean + c); is = *reinterpret_cast<const f32x4*>(bn.invstd + c);
    ga = *reinterpret_cast<const f32x4*>(bn.gamma + c); be = *reinterpret_cast<const f32x4*>(bn.beta + c);
  }
  for (int t = t0; t < t1; ++t) {
    f32x4 v = x[base + (long long)t * C4];
    if constexpr (BN) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float u = (v[k] - mu[k]) * is[k] * ga[k] + be[k];
        v[k] = u >= 0.f ? u : u * bn.slope;
      }
    }
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = a[k] + gs[k] * (v[k] - mean[k]) / den[k];
    dx[base + (long long)t * C4] = o;
  }
}

// y[permuted index] = x[i0, i1, i2] with output axes (p0, p1, p2) naming input axes; flip reverses one INPUT axis.
__global__ __launch_bounds__(256) void permute3_kernel(const float* __restrict__ x, float* __restrict__ y, int d0, int d1, int d2,
                                                       int p0, int p1, int p2, int flip) {
  const long long n = (long long)d0 * d1 * d2;
  const int dims[3] = {d0, d1, d2};
  const int o0 = dims[p0], o1 = dims[p1], o2 = dims[p2];
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    int oi[3];
    oi[2] = (int)(i % o2); oi[1] = (int)((i / o2) % o1); oi[0] = (int)(i / ((long long)o2 * o1));
    int in[3];
    in[p0] = oi[0]; in[p1] = oi[1]; in[p2] = oi[2];
    if (flip >= 0) in[flip] = dims[flip] - 1 - in[flip];
    y[i] = x[((long long)in[0] * d1 + in[1]) * d2 + in[2]];
    (void)o0;
  }
}

// out[0] = 2^floor(log2(target / max|x|)) (1 if x == 0), out[1] = 1 / out[0].  max|x| over the whole grid: a
// non-negative float orders like its bit pattern, so workgroup maxima meet in one atomicMax on the word that
// later holds out[0] (zeroed first); order-independent, hence deterministic.
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, unsigned* __restrict__ acc, long long n) {
  __shared__ float red[4];
  float m = 0.f;
  const long long n4 = n >> 2;
  const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const f32x4 v = x4[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
  }
  if (blockIdx.x == 0)
    for (long long i = (n4 << 2) + threadIdx.x; i < n; i += 256) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    if (!(m == m)) m = 3.4e38f;   // NaN: treated as "huge" (scale 1e-30 path below keeps the product finite)
    atomicMax(acc, __float_as_uint(m));
  }
}

__global__ void pow2_finalize_kernel(float* __restrict__ out, float target) {
  const float m = __uint_as_float(reinterpret_cast<const unsigned*>(out)[0]);
  float s = 1.f;
  if (m > 0.f && m < 3.0e38f) s = exp2f(floorf(log2f(target / m)));
  if (!(s > 0.f) || s > 1.0e30f) s = 1.0e30f;
  out[0] = s;
  out[1] = 1.f / s;
}

__global__ __launch_bounds__(256) void pow2_finalize_parts_kernel(const unsigned* parts, int n, float* out, float target) {
  pow2_finalize_parts(parts, n, out, target);
}

typedef _Float16 h4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void split_pack_scaled_kernel(const f32x4* __restrict__ x, float* __restrict__ y,
                                                                const float* __restrict__ scale, long long n4, DlipRange status) {
  const float s = scale[0];
  float amax = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const f32x4 v = x[i];
    h4 hi, lo;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const float t = v[k] * s; hi[k] = (_Float16)t; lo[k] = (_Float16)(t - (float)hi[k]); amax = fmaxf(amax, fabsf(t)); }
    const long long blk = i >> 3; const int q = (int)(i & 7);
    float* b = y + blk * 32;
    *reinterpret_cast<h4*>(b + q * 2) = hi;
    *reinterpret_cast<h4*>(b + 16 + q * 2) = lo;
  }
  dlip_report_range_block(amax, status);
}

// The same with the rows zero-padded from C to Cp channels (Cp = C rounded up to 32): a 1500-channel gradient becomes an operand
// of the split-fp16 kernels (the data gradient of the last TDNN layer ran on the exact-fp32 kernel: 0.97 of a 15 ms step).
__global__ __launch_bounds__(256) void split_pack_scaled_pad_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ scale,
                                                                    long long rows, int C, int Cp, DlipRange status) {
  const float s = scale[0];
  const int Cp4 = Cp / 4;
  const long long n4 = rows * Cp4;
  float amax = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const long long r = i / Cp4;
    const int c = (int)(i - r * Cp4) * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (c < C) v = *reinterpret_cast<const f32x4*>(x + r * C + c);       // C % 4 == 0
    h4 hi, lo;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const float t = v[k] * s; hi[k] = (_Float16)t; lo[k] = (_Float16)(t - (float)hi[k]); amax = fmaxf(amax, fabsf(t)); }
    const long long blk = i >> 3; const int q = (int)(i & 7);
    float* b = y + blk * 32;
    *reinterpret_cast<h4*>(b + q * 2) = hi;
    *reinterpret_cast<h4*>(b + 16 + q * 2) = lo;
  }
  dlip_report_range_block(amax, status);
}

// Weights of a training step -> the split-fp16 operand image, on the device (packing.split_weights does this once per
// load_state_dict on the host; under training the weights change every step).  One workgroup per output-channel row [L]:
// row maximum -> scale[k] = 2^floor(log2(1023 / max)) (lands the row's largest weight in [512, 1024): lo stays a normal fp16
// down to ~1e-4 of it) -> per 32-value block 32 hi halves | 32 lo halves.  An all-zero row gets scale 1.
__global__ __launch_bounds__(256) void split_weights_rows_kernel(const float* __restrict__ w, float* __restrict__ ws, float* __restrict__ scale,
                                                                 int L) {
  __shared__ float red[4];
  const float* row = w + (long long)blockIdx.x * L;
  float m = 0.f;
  for (int i = threadIdx.x; i < L; i += 256) m = fmaxf(m, fabsf(row[i]));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sc = 1.f;
  if (m > 0.f && m < 3.0e38f) sc = exp2f(floorf(log2f(1023.0f / m)));
  if (threadIdx.x == 0) scale[blockIdx.x] = sc;
  _Float16* out = reinterpret_cast<_Float16*>(ws + (long long)blockIdx.x * L);
  for (int i = threadIdx.x; i < L; i += 256) {
    const float t = row[i] * sc;
    const _Float16 hi = (_Float16)t, lo = (_Float16)(t - (float)hi);
    const int b = i >> 5, q = i & 31;
    out[b * 64 + q] = hi;
    out[b * 64 + 32 + q] = lo;
  }
}

// The same from the REFERENCE layout w [K, C, T] (T = R*S taps; nn.Conv2d / nn.Conv1d weights), permutation included -- one launch
// instead of permute + split, 94 times per training step of the lip-clip model:
//   MODE 0 (forward):        row k = [T][C]          row[t * C + c]             = w[k, c, t]
//   MODE 1 (data gradient):  row c = [T flipped][K]  row[(T - 1 - t) * K + k]   = w[k, c, t]
// One workgroup per output row; the row's blocks of 32 values are (hi | lo) as above.  Rows are a few thousand values read
// through L2 (the whole bank is at most 9.4 MB), so the strided gather of MODE 1 costs nothing measurable.
template <int MODE>
__global__ __launch_bounds__(256) void split_weights_perm_kernel(const float* __restrict__ w, float* __restrict__ ws, float* __restrict__ scale,
                                                                 int K, int C, int T, int Cp) {
  __shared__ float red[4];
  __shared__ float stage[8192];                   // the row's source values in OUTPUT order (rows of up to 8192 values: else from memory)
  const int row = blockIdx.x;
  const int inner = Cp;                           // channels of one tap in the output row: C (MODE 0) resp. K (MODE 1) zero-padded to Cp
  const int L = T * inner;
  const bool staged = L <= 8192;
  if (staged) {
    if (Cp != (MODE == 0 ? C : K)) {
      for (int i = threadIdx.x; i < L; i += 256) stage[i] = 0.f;
      __syncthreads();
    }
    // read in SOURCE order -- the T taps of one (k, c) pair are contiguous, MODE 0's whole row is -- and scatter into LDS: MODE 1
    // read in output order touches one 4-byte word per 128-byte line (12 us per call, 0.6 ms per training step)
    for (int s_ = threadIdx.x; s_ < T * (MODE == 0 ? C : K); s_ += 256) {
      const int j = s_ / T, t = s_ - j * T;       // j: the other channel index (c for MODE 0, k for MODE 1)
      const float v = MODE == 0 ? w[(long long)row * C * T + s_] : w[((long long)j * C + row) * T + t];
      stage[(MODE == 0 ? t : T - 1 - t) * inner + j] = v;
    }
    __syncthreads();
  }
  auto src = [&](int i) -> float {
    if (staged) return stage[i];
    const int t = i / inner, j = i - t * inner;
    if (j >= (MODE == 0 ? C : K)) return 0.f;
    return MODE == 0 ? w[((long long)row * C + j) * T + t] : w[((long long)j * C + row) * T + (T - 1 - t)];
  };
  float m = 0.f;
  for (int i = threadIdx.x; i < L; i += 256) m = fmaxf(m, fabsf(src(i)));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sc = 1.f;
  if (m > 0.f && m < 3.0e38f) sc = exp2f(floorf(log2f(1023.0f / m)));
  if (threadIdx.x == 0) scale[row] = sc;
  _Float16* out = reinterpret_cast<_Float16*>(ws + (long long)row * L);
  for (int i = threadIdx.x; i < L; i += 256) {
    const float t = src(i) * sc;
    const _Float16 hi = (_Float16)t, lo = (_Float16)(t - (float)hi);
    const int b = i >> 5, q = i & 31;
    out[b * 64 + q] = hi;
    out[b * 64 + 32 + q] = lo;
  }
}

// split_weights_perm_kernel for MANY weight tensors in one launch (round 4): a training step split the current weights of each of
// its ~47 convolutions twice (forward bank, data-gradient bank) in 94 launches of 8 - 10 us, each in front of the convolution that
// waits for it; removing them (timing experiment) took 0.63 ms off an 18.6 ms lip-clip step.  One workgroup per output row of
// some tensor: blk2desc names the tensor, the descriptor its shape, mode and the row's offset.  Same arithmetic per row as the
// single-tensor kernel (rows staged through LDS in output order, row maximum -> power-of-two scale -> hi / lo halves): same bits.
struct WSplitDesc {
  const float* w;
  float* ws;
  float* scale;
  int32_t K, C, T, Cp, mode, row0;
};

// (round 5) The launch held the chip for 340 us of an 15.3 ms lip-clip step -- 38 k workgroups of one row each, four per CU (a
// static 32 KB staging buffer), every element written by two 2-byte stores.  Now: the staging buffer is dynamic LDS sized to the
// launch's longest row (stage_floats; rows beyond it take the unstaged path as before), the row maximum is taken from 16-byte LDS
// reads, and a thread writes eight consecutive values as one 16-byte piece of hi halves and one of lo halves.  Same value per
// element, same scale per row: same bits.
__global__ __launch_bounds__(256) void split_weights_multi_kernel(const WSplitDesc* __restrict__ descs, const int32_t* __restrict__ blk2desc,
                                                                  int stage_floats) {
  __shared__ float red[4];
  extern __shared__ __attribute__((aligned(16))) float stage[];
  const WSplitDesc d = descs[blk2desc[blockIdx.x]];
  const int row = (int)blockIdx.x - d.row0;
  const int K = d.K, C = d.C, T = d.T, mode = d.mode;
  const int inner = d.Cp, real = mode == 0 ? C : K;
  const int L = T * inner;                          // a multiple of 32 (Cp is)
  const bool staged = L <= stage_floats;
  const float* __restrict__ w = d.w;
  if (staged) {
    if (inner != real) {
      for (int i = threadIdx.x; i < L; i += 256) stage[i] = 0.f;
      __syncthreads();
    }
    for (int s_ = threadIdx.x; s_ < T * real; s_ += 256) {
      const int j = s_ / T, t = s_ - j * T;
      const float v = mode == 0 ? w[(long long)row * C * T + s_] : w[((long long)j * C + row) * T + t];
      stage[(mode == 0 ? t : T - 1 - t) * inner + j] = v;
    }
    __syncthreads();
  }
  auto src = [&](int i) -> float {
    if (staged) return stage[i];
    const int t = i / inner, j = i - t * inner;
    if (j >= real) return 0.f;
    return mode == 0 ? w[((long long)row * C + j) * T + t] : w[((long long)j * C + row) * T + (T - 1 - t)];
  };
  float m = 0.f;
  if (staged) {
    for (int i = threadIdx.x; i < L / 4; i += 256) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(stage + 4 * i);
      m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
  } else {
    for (int i = threadIdx.x; i < L; i += 256) m = fmaxf(m, fabsf(src(i)));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sc = 1.f;
  if (m > 0.f && m < 3.0e38f) sc = exp2f(floorf(log2f(1023.0f / m)));
  if (threadIdx.x == 0) d.scale[row] = sc;
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  _Float16* out = reinterpret_cast<_Float16*>(d.ws + (long long)row * L);
  for (int i8 = threadIdx.x; i8 < L / 8; i8 += 256) {       // eight consecutive values: a quarter of a 32-value block
    float v[8];
    if (staged) {
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(stage + 8 * i8), a1 = *reinterpret_cast<const f32x4*>(stage + 8 * i8 + 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) { v[k] = a0[k]; v[4 + k] = a1[k]; }
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = src(8 * i8 + k);
    }
    h8 hi, lo;
#pragma unroll
    for (int k = 0; k < 8; ++k) { const float t = v[k] * sc; hi[k] = (_Float16)t; lo[k] = (_Float16)(t - (float)hi[k]); }
    const int blk = i8 >> 2, q = (i8 & 3) * 8;
    *reinterpret_cast<h8*>(out + blk * 64 + q) = hi;
    *reinterpret_cast<h8*>(out + blk * 64 + 32 + q) = lo;
  }
}

__global__ __launch_bounds__(256) void fill_from_scalar_kernel(const float* __restrict__ src, float* __restrict__ y, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] = src[0];
}

inline unsigned grid1d(long long n) {
  long long g = (n + 255) / 256;
  if (g > 4096) g = 4096;
  return (unsigned)(g < 1 ? 1 : g);
}

// Grid for an element pass over [M, C4] float4s whose threads want ONE channel group each: a grid whose stride 256 g is a
// multiple of C4 (g a multiple of C4 / gcd(C4, 256)); 0 when no such grid fits (the kernel then re-reads its parameters).
inline unsigned grid_fixed(long long n4, int C4) {
  int a = C4, b = 256;
  while (b) { const int t = a % b; a = b; b = t; }
  const long long m = C4 / a;
  long long g = (long long)grid1d(n4) / m * m;
  if (g == 0 && m <= 4096 && (n4 + 255) / 256 >= 1) g = m <= (n4 + 255) / 256 ? m : 0;
  return (unsigned)g;
}

// Parts of a column reduction over M rows: 512-row parts up to 512 of them, longer parts beyond (a multiple of the 16 row groups).
inline int bn_rows_per_part(int M) {
  long long rpp = ((long long)M + 511) / 512;
  rpp = (rpp + 15) / 16 * 16;
  return rpp < CHUNK_ROWS ? CHUNK_ROWS : (int)rpp;
}

}  // namespace

extern "C" int32_t dlip_bn_rows_chunks(int32_t M) {
  if (M <= 0) return 0;
  const int rpp = bn_rows_per_part(M);
  return (M + rpp - 1) / rpp;
}

// conv_igemm_f16x3_dma.hip: this stream's self-resetting ticket words (the balanced split's workspace); 0 = none (a capture on a
// stream that never launched eagerly): the finalize steps then run as launches of their own
extern "C" int dlip_conv_split_workspace(void* stream, size_t slab_floats, float** slabs, int** counters, int* counter_words);

namespace {

int* stream_tickets(hipStream_t st, int need) {
  if (dlip_dbg_value[DLIP_DBG_BN_FUSED] == 0) return nullptr;       // dlip_debug_set(8, 0): the round-4 launch sequence (tests, A/B runs)
  float* slabs = nullptr;
  int* ctr = nullptr;
  int words = 0;
  if (!dlip_conv_split_workspace(st, 0, &slabs, &ctr, &words) || words < need) return nullptr;
  return ctr;
}

bool bn_small(int M) { return M <= BN_SMALL_ROWS && dlip_dbg_value[DLIP_DBG_BN_FUSED] != 0; }

// Shared forward: statistics (a pass over x, or `ready_chunks` partial rows from the producing convolution), finalize, apply.
int bn_fwd_launch(const float* x, const float* gamma, const float* beta, const float* slope_vec, float* y, float* save_mean,
                  float* save_invstd, float* running_mean, float* running_var, double* workspace, int M, int C, float momentum,
                  float eps, float slope, int act_first, int ready_chunks, long long* nbt, hipStream_t st) {
  if (ready_chunks == 0 && bn_small(M) && y != nullptr) {
    const dim3 grid(C / 4), block(256);
    if (M <= 1024) hipLaunchKernelGGL(bn_small_fwd_kernel<4>, grid, block, 0, st, x, gamma, beta, slope_vec, y, save_mean, save_invstd,
                                      running_mean, running_var, nbt, M, C, momentum, eps, slope, act_first);
    else if (M <= 2048) hipLaunchKernelGGL(bn_small_fwd_kernel<8>, grid, block, 0, st, x, gamma, beta, slope_vec, y, save_mean, save_invstd,
                                           running_mean, running_var, nbt, M, C, momentum, eps, slope, act_first);
    else hipLaunchKernelGGL(bn_small_fwd_kernel<16>, grid, block, 0, st, x, gamma, beta, slope_vec, y, save_mean, save_invstd,
                            running_mean, running_var, nbt, M, C, momentum, eps, slope, act_first);
    return dlip_launch_status();
  }
  // ready_chunks > 0: `workspace` already holds that many partial rows {sum x, sum x^2} [chunk][C][2] -- written by the convolution
  // that produced x (dlip_conv_nhwc_stats_f16x3) -- and the statistics pass over x is not launched
  const int chunks = ready_chunks > 0 ? ready_chunks : dlip_bn_rows_chunks(M);
  bool finalized = false;
  if (ready_chunks == 0) {
    ColFin fin = {stream_tickets(st, (C + 63) / 64), save_mean, save_invstd, nullptr, running_mean, running_var, nbt, momentum, eps};
    finalized = fin.ticket != nullptr;
    hipLaunchKernelGGL(col_partial_kernel<0>, dim3((C + 63) / 64, chunks), dim3(256), 0, st, x, nullptr, nullptr, nullptr,
                       nullptr, nullptr, workspace, M, C, slope, act_first, nullptr, bn_rows_per_part(M), fin);
  }
  if (!finalized)
    hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3((C + 15) / 16), dim3(256), 0, st, workspace, save_mean, save_invstd,
                       running_mean, running_var, M, C, chunks, momentum, eps, nbt);
  if (y == nullptr) return dlip_launch_status();   // statistics only: the consumer applies the BatchNorm on load (dlip_wgrad_*_bn_f32)
  const long long n4 = (long long)M * (C / 4);
  if (const unsigned gf = grid_fixed(n4, C / 4))
    hipLaunchKernelGGL(bn_fwd_apply_kernel<true>, dim3(gf), dim3(256), 0, st, reinterpret_cast<const f32x4*>(x), save_mean,
                       save_invstd, gamma, beta, reinterpret_cast<f32x4*>(y), n4, C / 4, slope, act_first, slope_vec);
  else
    hipLaunchKernelGGL(bn_fwd_apply_kernel<false>, dim3(grid1d(n4)), dim3(256), 0, st, reinterpret_cast<const f32x4*>(x), save_mean,
                       save_invstd, gamma, beta, reinterpret_cast<f32x4*>(y), n4, C / 4, slope, act_first, slope_vec);
  return dlip_launch_status();
}

// Shared backward: slope_vec != NULL = the PReLU form (dslope written; act_first = 0), else the LeakyReLU forms.
int bn_bwd_launch(const float* dy, const float* x, const float* gamma, const float* beta, const float* slope_vec,
                  const float* save_mean, const float* save_invstd, float* dx, float* dgamma, float* dbeta, float* dslope,
                  double* workspace, int M, int C, float slope, int act_first, float* dx_lift2, hipStream_t st, const MsSrc ms = MsSrc{}) {
  unsigned* acc = dx_lift2 ? reinterpret_cast<unsigned*>(dx_lift2) + 2 : nullptr;   // per-workgroup maxima behind the pair
  int* tickets = stream_tickets(st, (C + 63) / 64);
  if (bn_small(M) && C / 4 <= 4096) {
    const unsigned grid = (unsigned)(C / 4);
    if (M <= 1024) hipLaunchKernelGGL(bn_small_bwd_kernel<4>, dim3(grid), dim3(256), 0, st, dy, x, save_mean, save_invstd, gamma, beta, slope_vec,
                                      dx, dgamma, dbeta, dslope, M, C, slope, act_first, acc, tickets);
    else if (M <= 2048) hipLaunchKernelGGL(bn_small_bwd_kernel<8>, dim3(grid), dim3(256), 0, st, dy, x, save_mean, save_invstd, gamma, beta,
                                           slope_vec, dx, dgamma, dbeta, dslope, M, C, slope, act_first, acc, tickets);
    else hipLaunchKernelGGL(bn_small_bwd_kernel<16>, dim3(grid), dim3(256), 0, st, dy, x, save_mean, save_invstd, gamma, beta, slope_vec, dx,
                            dgamma, dbeta, dslope, M, C, slope, act_first, acc, tickets);
    if (acc && !tickets) hipLaunchKernelGGL(pow2_finalize_parts_kernel, dim3(1), dim3(256), 0, st, acc, (int)grid, dx_lift2, 1024.0f);
    return dlip_launch_status();
  }
  const int chunks = dlip_bn_rows_chunks(M);
  const int rpp = bn_rows_per_part(M);
  ColFin fin = {tickets, dbeta, dgamma, dslope, nullptr, nullptr, nullptr, 0.f, 0.f};
  if (slope_vec) {
    hipLaunchKernelGGL(col_partial_kernel<3>, dim3((C + 63) / 64, chunks), dim3(256), 0, st, x, dy, save_mean, save_invstd,
                       gamma, beta, workspace, M, C, 1.f, 0, slope_vec, rpp, fin);
    if (!tickets)
      hipLaunchKernelGGL(col_finalize3_kernel, dim3((C + 15) / 16), dim3(256), 0, st, workspace, workspace + (long long)chunks * C * 2,
                         dbeta, dgamma, dslope, C, chunks);
  } else {
    fin.ms = ms;
    hipLaunchKernelGGL(col_partial_kernel<1>, dim3((C + 63) / 64, chunks), dim3(256), 0, st, x, dy, save_mean, save_invstd,
                       gamma, beta, workspace, M, C, slope, act_first, nullptr, rpp, fin);
    if (!tickets) hipLaunchKernelGGL(col_finalize_kernel, dim3((C + 15) / 16), dim3(256), 0, st, workspace, dbeta, dgamma, C, chunks);
  }
  const long long n4 = (long long)M * (C / 4);
  const unsigned gf = grid_fixed(n4, C / 4);
  const unsigned grid = gf ? gf : grid1d(n4);
  // (the lift of a LARGE dx stays a launch of its own: a ticket word taking thousands of arrivals, and every workgroup waiting for its
  // stores' acknowledgements in front of it, cost bn_bwd_apply 0.6 ms per lip-clip step -- more than the 20 launches it saved)
  int* lift_ticket = nullptr;
  if (gf)
    hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, dim3(gf), dim3(256), 0, st, reinterpret_cast<const f32x4*>(dy),
                       reinterpret_cast<const f32x4*>(x), save_mean, save_invstd, gamma, beta, dgamma, dbeta,
                       reinterpret_cast<f32x4*>(dx), n4, C / 4, M, slope, act_first, slope_vec, acc, lift_ticket, PoolSrc{}, ms);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, dim3(grid), dim3(256), 0, st, reinterpret_cast<const f32x4*>(dy),
                       reinterpret_cast<const f32x4*>(x), save_mean, save_invstd, gamma, beta, dgamma, dbeta,
                       reinterpret_cast<f32x4*>(dx), n4, C / 4, M, slope, act_first, slope_vec, acc, lift_ticket, PoolSrc{}, ms);
  if (acc && !lift_ticket) hipLaunchKernelGGL(pow2_finalize_parts_kernel, dim3(1), dim3(256), 0, st, acc, (int)grid, dx_lift2, 1024.0f);
  return dlip_launch_status();
}

}  // namespace

extern "C" int dlip_bn_rows_train_fwd_f32(const float* x, const float* gamma, const float* beta, float* y,
                                          float* save_mean, float* save_invstd, float* running_mean,
                                          float* running_var, double* workspace, int32_t M, int32_t C, float momentum,
                                          float eps, float slope, int32_t act_first, int32_t ready_chunks,
                                          int64_t* num_batches_tracked, dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && gamma && beta && save_mean && save_invstd && workspace && M > 0 && C > 0 && (C & 3) == 0);   // (y NULL: statistics only)
  DLIP_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr));
  DLIP_CHECK_ARG(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0);
  DLIP_CHECK_ARG(ready_chunks >= 0 && !(ready_chunks > 0 && act_first));
  return bn_fwd_launch(x, gamma, beta, nullptr, y, save_mean, save_invstd, running_mean, running_var, workspace, M, C, momentum, eps,
                       slope, act_first, ready_chunks, reinterpret_cast<long long*>(num_batches_tracked), static_cast<hipStream_t>(stream));
}

extern "C" int dlip_bn_rows_train_bwd_f32(const float* dy, const float* x, const float* gamma, const float* beta,
                                          const float* save_mean, const float* save_invstd, float* dx, float* dgamma,
                                          float* dbeta, double* workspace, int32_t M, int32_t C, float slope,
                                          int32_t act_first, float* dx_lift2, dlip_stream_t stream) {
  DLIP_CHECK_ARG(dy && x && gamma && beta && save_mean && save_invstd && dx && dgamma && dbeta && workspace);
  DLIP_CHECK_ARG(M > 0 && C > 0 && (C & 3) == 0);
  DLIP_CHECK_ARG(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0);
  return bn_bwd_launch(dy, x, gamma, beta, nullptr, save_mean, save_invstd, dx, dgamma, dbeta, nullptr, workspace, M, C, slope, act_first,
                       dx_lift2, static_cast<hipStream_t>(stream));
}

// (ABI 48) dlip_bn_rows_train_bwd_f32 behind a MeanStdPooling whose backward writes nothing: dy is formed per loaded value from the pooled
// statistics y_pool [B,2C] and their gradient g_pool [B,2C] (MsSrc; M = B T rows, conv -> BatchNorm -> LeakyReLU order, more than
// BN_SMALL_ROWS rows).
extern "C" int dlip_meanstd_bwd_coef_f32(const float* y_pool, const float* g_pool, float* coef, int32_t B, int32_t C, int32_t T,
                                         dlip_stream_t stream) {
  DLIP_CHECK_ARG(y_pool && g_pool && coef && B > 0 && C > 0 && T > 1);
  const long long n = (long long)B * C;
  hipLaunchKernelGGL(ms_coef_kernel, dim3((unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     y_pool, g_pool, coef, B, C, T);
  return dlip_launch_status();
}

extern "C" int dlip_bn_rows_train_bwd_ms_f32(const float* ms_coef, int32_t T, const float* x, const float* gamma, const float* beta,
                                             const float* save_mean, const float* save_invstd, float* dx, float* dgamma, float* dbeta,
                                             double* workspace, int32_t M, int32_t C, float slope, float* dx_lift2, dlip_stream_t stream) {
  DLIP_CHECK_ARG(ms_coef && x && gamma && beta && save_mean && save_invstd && dx && dgamma && dbeta && workspace);
  DLIP_CHECK_ARG(M > 0 && C > 0 && (C & 3) == 0 && T > 1 && M % T == 0 && !bn_small(M));
  DLIP_CHECK_ARG(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(ms_coef)) & 15) == 0);
  MsSrc ms;
  ms.coef = ms_coef; ms.T = T; ms.div_T = dlip_fastdiv((uint32_t)T);
  return bn_bwd_launch(x /* never read as dy */, x, gamma, beta, nullptr, save_mean, save_invstd, dx, dgamma, dbeta, nullptr, workspace, M, C, slope, 0,
                       dx_lift2, static_cast<hipStream_t>(stream), ms);
}

// (ABI 47) The first half of dlip_bn_rows_train_bwd_f32 alone: dgamma, dbeta and the lift of a dx that is never written -- the operand
// producers of the convolution in front form dx per loaded value (dlip_wgrad_operand_split_bnbwd_f32 / dlip_wgrad_chwn_bnbwd_f32).
extern "C" int dlip_bn_rows_train_bwd_sums_f32(const float* dy, const float* x, const float* gamma, const float* beta, const float* save_mean,
                                               const float* save_invstd, float* dgamma, float* dbeta, double* workspace, float* amax_parts,
                                               int32_t M, int32_t C, float slope, int32_t act_first, float* dx_lift2, const float* ms_coef,
                                               int32_t ms_T, dlip_stream_t stream) {
  // (ABI 49) ms_coef / ms_T (nullable): dy formed on load from a MeanStdPooling's coefficients (dlip_meanstd_bwd_coef_f32; MsSrc); dy may then be NULL
  DLIP_CHECK_ARG(x && gamma && beta && save_mean && save_invstd && dgamma && dbeta && workspace && amax_parts && dx_lift2);
  DLIP_CHECK_ARG((dy != nullptr || ms_coef != nullptr) && (ms_coef == nullptr || (ms_T > 1 && !act_first && M % ms_T == 0)));
  DLIP_CHECK_ARG(M > 0 && C > 0 && (C & 3) == 0);
  DLIP_CHECK_ARG(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(ms_coef)) & 15) == 0);
  if (dy == nullptr) dy = x;           // (never read)
  hipStream_t st = static_cast<hipStream_t>(stream);
  int* tickets = stream_tickets(st, (C + 63) / 64);
  const int chunks = dlip_bn_rows_chunks(M);
  const int rpp = bn_rows_per_part(M);
  ColFin fin = {tickets, dbeta, dgamma, nullptr, nullptr, nullptr, nullptr, 0.f, 0.f};
  fin.amax_parts = reinterpret_cast<unsigned*>(amax_parts);
  if (ms_coef != nullptr) { fin.ms.coef = ms_coef; fin.ms.T = ms_T; fin.ms.div_T = dlip_fastdiv((uint32_t)ms_T); }
  hipLaunchKernelGGL(col_partial_kernel<1>, dim3((C + 63) / 64, chunks), dim3(256), 0, st, x, dy, save_mean, save_invstd, gamma, beta, workspace, M,
                     C, slope, act_first, nullptr, rpp, fin);
  if (!tickets) hipLaunchKernelGGL(col_finalize_kernel, dim3((C + 15) / 16), dim3(256), 0, st, workspace, dbeta, dgamma, C, chunks);
  hipLaunchKernelGGL(bn_bwd_lift_bound_kernel, dim3(1), dim3(256), 0, st, reinterpret_cast<const unsigned*>(amax_parts), ((C + 63) / 64) * chunks,
                     gamma, save_invstd, C, dx_lift2, 1024.0f);
  return dlip_launch_status();
}

extern "C" int dlip_bn_prelu_rows_train_fwd_f32(const float* x, const float* gamma, const float* beta, const float* slope, float* y,
                                                float* save_mean, float* save_invstd, float* running_mean, float* running_var,
                                                double* workspace, int32_t M, int32_t C, float momentum, float eps,
                                                int64_t* num_batches_tracked, dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && gamma && beta && slope && save_mean && save_invstd && workspace && M > 0 && C > 0 && (C & 3) == 0);   // (y NULL: statistics only)
  DLIP_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr));
  DLIP_CHECK_ARG(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0);
  return bn_fwd_launch(x, gamma, beta, slope, y, save_mean, save_invstd, running_mean, running_var, workspace, M, C, momentum, eps, 1.f, 0,
                       0, reinterpret_cast<long long*>(num_batches_tracked), static_cast<hipStream_t>(stream));
}

extern "C" int dlip_bn_prelu_rows_train_bwd_f32(const float* dy, const float* x, const float* gamma, const float* beta,
                                                const float* slope, const float* save_mean, const float* save_invstd, float* dx,
                                                float* dgamma, float* dbeta, float* dslope, double* workspace, int32_t M, int32_t C,
                                                float* dx_lift2, dlip_stream_t stream) {
  DLIP_CHECK_ARG(dy && x && gamma && beta && slope && save_mean && save_invstd && dx && dgamma && dbeta && dslope && workspace);
  DLIP_CHECK_ARG(M > 0 && C > 0 && (C & 3) == 0);
  DLIP_CHECK_ARG(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0);
  return bn_bwd_launch(dy, x, gamma, beta, slope, save_mean, save_invstd, dx, dgamma, dbeta, dslope, workspace, M, C, 1.f, 0, dx_lift2,
                       static_cast<hipStream_t>(stream));
}

extern "C" int dlip_bn_prelu_maxpool_train_fwd_f32(const float* x, const float* gamma, const float* beta, const float* slope, float* y,
                                                  uint32_t* idx, float* save_mean, float* save_invstd, float* running_mean,
                                                  float* running_var, double* workspace, int64_t N, int32_t H, int32_t W, int32_t C,
                                                  float momentum, float eps, int64_t* num_batches_tracked, dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && gamma && beta && slope && y && idx && save_mean && save_invstd && workspace && N > 0 && H > 0 && W > 0 && C > 0);
  DLIP_CHECK_ARG((C & 3) == 0 && (running_mean == nullptr) == (running_var == nullptr) && N * H * W < 0x7FFFFFFFll);
  DLIP_CHECK_ARG(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int M = (int)(N * H * W);
  const int chunks = dlip_bn_rows_chunks(M);
  long long* nbt = reinterpret_cast<long long*>(num_batches_tracked);
  ColFin fin = {stream_tickets(st, (C + 63) / 64), save_mean, save_invstd, nullptr, running_mean, running_var, nbt, momentum, eps};
  hipLaunchKernelGGL(col_partial_kernel<0>, dim3((C + 63) / 64, chunks), dim3(256), 0, st, x, nullptr, nullptr, nullptr,
                     nullptr, nullptr, workspace, M, C, 1.f, 0, nullptr, bn_rows_per_part(M), fin);
  if (!fin.ticket)
    hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3((C + 15) / 16), dim3(256), 0, st, workspace, save_mean, save_invstd,
                       running_mean, running_var, M, C, chunks, momentum, eps, nbt);
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long long n4 = N * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(bn_prelu_maxpool_fwd_kernel, dim3(grid1d(n4)), dim3(256), 0, st, x, save_mean, save_invstd, gamma, beta, slope,
                     reinterpret_cast<f32x4*>(y), idx, H, W, Ho, Wo, C / 4, n4);
  return dlip_launch_status();
}

extern "C" int dlip_bn_prelu_maxpool_train_bwd_f32(const float* dy_pooled, const float* dy_pooled2, const uint32_t* idx, const float* x, const float* gamma,
                                                  const float* beta, const float* slope, const float* save_mean,
                                                  const float* save_invstd, float* dx, float* dgamma, float* dbeta, float* dslope,
                                                  double* workspace, int64_t N, int32_t H, int32_t W, int32_t C, float* dx_lift2,
                                                  dlip_stream_t stream) {
  DLIP_CHECK_ARG(dy_pooled && idx && x && gamma && beta && slope && save_mean && save_invstd && dx && dgamma && dbeta && dslope && workspace);
  DLIP_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0 && N * H * W < 0x7FFFFFFFll);
  DLIP_CHECK_ARG(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy_pooled) | reinterpret_cast<uintptr_t>(dy_pooled2) |
                   reinterpret_cast<uintptr_t>(dx)) & 15) == 0);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int M = (int)(N * H * W);
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const PoolSrc ps = {idx, dy_pooled2, H, W, Ho, Wo, dlip_fastdiv((uint32_t)W), dlip_fastdiv((uint32_t)H), dlip_fastdiv((uint32_t)Wo), dlip_fastdiv((uint32_t)Ho)};
  const int Mp = (int)(N * Ho * Wo);
  const int chunks = dlip_bn_rows_chunks(Mp);           // (the sums run over the POOLED rows; workspace sized for M >= Mp)
  ColFin fin = {stream_tickets(st, (C + 63) / 64), dbeta, dgamma, dslope, nullptr, nullptr, nullptr, 0.f, 0.f};
  hipLaunchKernelGGL(pool_bn_partial_kernel, dim3((C + 63) / 64, chunks), dim3(256), 0, st, x, dy_pooled, ps, save_mean, save_invstd,
                     gamma, beta, slope, workspace, Mp, M, C, bn_rows_per_part(Mp), fin);
  if (!fin.ticket)
    hipLaunchKernelGGL(col_finalize3_kernel, dim3((C + 15) / 16), dim3(256), 0, st, workspace, workspace + (long long)chunks * C * 2,
                       dbeta, dgamma, dslope, C, chunks);
  unsigned* acc = dx_lift2 ? reinterpret_cast<unsigned*>(dx_lift2) + 2 : nullptr;
  const long long n4 = (long long)M * (C / 4);
  const unsigned gf = grid_fixed(n4, C / 4);
  const unsigned grid = gf ? gf : grid1d(n4);
  if (gf)
    hipLaunchKernelGGL((bn_bwd_apply_kernel<true, true>), dim3(gf), dim3(256), 0, st, reinterpret_cast<const f32x4*>(dy_pooled),
                       reinterpret_cast<const f32x4*>(x), save_mean, save_invstd, gamma, beta, dgamma, dbeta,
                       reinterpret_cast<f32x4*>(dx), n4, C / 4, M, 1.f, 0, slope, acc, nullptr, ps);
  else
    hipLaunchKernelGGL((bn_bwd_apply_kernel<false, true>), dim3(grid), dim3(256), 0, st, reinterpret_cast<const f32x4*>(dy_pooled),
                       reinterpret_cast<const f32x4*>(x), save_mean, save_invstd, gamma, beta, dgamma, dbeta,
                       reinterpret_cast<f32x4*>(dx), n4, C / 4, M, 1.f, 0, slope, acc, nullptr, ps);
  if (acc) hipLaunchKernelGGL(pow2_finalize_parts_kernel, dim3(1), dim3(256), 0, st, acc, (int)grid, dx_lift2, 1024.0f);
  return dlip_launch_status();
}

extern "C" int dlip_bn_add_prelu_rows_train_fwd_f32(const float* x, const float* residual, const float* gamma, const float* beta,
                                                   const float* slope, float* sum, float* y, float* save_mean, float* save_invstd,
                                                   float* running_mean, float* running_var, double* workspace, int32_t M, int32_t C,
                                                   float momentum, float eps, int64_t* num_batches_tracked, dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && residual && gamma && beta && slope && sum && y && save_mean && save_invstd && workspace && M > 0 && C > 0 && (C & 3) == 0);
  DLIP_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr));
  DLIP_CHECK_ARG(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(residual) | reinterpret_cast<uintptr_t>(sum) |
                   reinterpret_cast<uintptr_t>(y)) & 15) == 0);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int chunks = dlip_bn_rows_chunks(M);
  long long* nbt = reinterpret_cast<long long*>(num_batches_tracked);
  ColFin fin = {stream_tickets(st, (C + 63) / 64), save_mean, save_invstd, nullptr, running_mean, running_var, nbt, momentum, eps};
  hipLaunchKernelGGL(col_partial_kernel<0>, dim3((C + 63) / 64, chunks), dim3(256), 0, st, x, nullptr, nullptr, nullptr,
                     nullptr, nullptr, workspace, M, C, 1.f, 0, nullptr, bn_rows_per_part(M), fin);
  if (!fin.ticket)
    hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3((C + 15) / 16), dim3(256), 0, st, workspace, save_mean, save_invstd,
                       running_mean, running_var, M, C, chunks, momentum, eps, nbt);
  const long long n4 = (long long)M * (C / 4);
  if (const unsigned gf = grid_fixed(n4, C / 4))
    hipLaunchKernelGGL(bn_add_prelu_fwd_kernel<true>, dim3(gf), dim3(256), 0, st, reinterpret_cast<const f32x4*>(x),
                       reinterpret_cast<const f32x4*>(residual), save_mean, save_invstd, gamma, beta, slope,
                       reinterpret_cast<f32x4*>(sum), reinterpret_cast<f32x4*>(y), n4, C / 4);
  else
    hipLaunchKernelGGL(bn_add_prelu_fwd_kernel<false>, dim3(grid1d(n4)), dim3(256), 0, st, reinterpret_cast<const f32x4*>(x),
                       reinterpret_cast<const f32x4*>(residual), save_mean, save_invstd, gamma, beta, slope,
                       reinterpret_cast<f32x4*>(sum), reinterpret_cast<f32x4*>(y), n4, C / 4);
  return dlip_launch_status();
}

extern "C" int dlip_bn_add_prelu_rows_train_bwd_f32(const float* dy, const float* dy2, const float* sum, const float* x,
                                                   const float* gamma, const float* beta, const float* slope, const float* save_mean,
                                                   const float* save_invstd, float* dresidual, float* dx, float* dgamma, float* dbeta,
                                                   float* dslope, double* workspace, int32_t M, int32_t C, float* dx_lift2,
                                                   dlip_stream_t stream) {
  DLIP_CHECK_ARG(dy && sum && x && gamma && beta && slope && save_mean && save_invstd && dresidual && dx && dgamma && dbeta && dslope && workspace);
  DLIP_CHECK_ARG(M > 0 && C > 0 && (C & 3) == 0);
  DLIP_CHECK_ARG(((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dy2) | reinterpret_cast<uintptr_t>(sum) | reinterpret_cast<uintptr_t>(x) |
                   reinterpret_cast<uintptr_t>(dresidual) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int chunks = dlip_bn_rows_chunks(M);
  ColFin fin = {stream_tickets(st, (C + 63) / 64), dbeta, dgamma, dslope, nullptr, nullptr, nullptr, 0.f, 0.f};
  hipLaunchKernelGGL(add_prelu_bn_partial_kernel, dim3((C + 63) / 64, chunks), dim3(256), 0, st, dy, dy2, sum, x, save_mean, save_invstd,
                     slope, dresidual, workspace, M, C, bn_rows_per_part(M), fin);
  if (!fin.ticket)
    hipLaunchKernelGGL(col_finalize3_kernel, dim3((C + 15) / 16), dim3(256), 0, st, workspace, workspace + (long long)chunks * C * 2,
                       dbeta, dgamma, dslope, C, chunks);
  // second pass: bn2's input gradient from g (= dresidual) and x -- the BatchNorm backward's apply pass at slope 1
  unsigned* acc = dx_lift2 ? reinterpret_cast<unsigned*>(dx_lift2) + 2 : nullptr;
  const long long n4 = (long long)M * (C / 4);
  const unsigned gf = grid_fixed(n4, C / 4);
  const unsigned grid = gf ? gf : grid1d(n4);
  if (gf)
    hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, dim3(gf), dim3(256), 0, st, reinterpret_cast<const f32x4*>(dresidual),
                       reinterpret_cast<const f32x4*>(x), save_mean, save_invstd, gamma, beta, dgamma, dbeta,
                       reinterpret_cast<f32x4*>(dx), n4, C / 4, M, 1.f, 0, nullptr, acc, nullptr);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, dim3(grid), dim3(256), 0, st, reinterpret_cast<const f32x4*>(dresidual),
                       reinterpret_cast<const f32x4*>(x), save_mean, save_invstd, gamma, beta, dgamma, dbeta,
                       reinterpret_cast<f32x4*>(dx), n4, C / 4, M, 1.f, 0, nullptr, acc, nullptr);
  if (acc) hipLaunchKernelGGL(pow2_finalize_parts_kernel, dim3(1), dim3(256), 0, st, acc, (int)grid, dx_lift2, 1024.0f);
  return dlip_launch_status();
}

extern "C" int dlip_bn_apply_rows_f32(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                      const float* slope_vec, float slope, float* y, int32_t M, int32_t C, dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && mean && invstd && gamma && beta && y && M > 0 && C > 0 && (C & 3) == 0);
  DLIP_CHECK_ARG(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long long n4 = (long long)M * (C / 4);
  if (const unsigned gf = grid_fixed(n4, C / 4))
    hipLaunchKernelGGL(bn_fwd_apply_kernel<true>, dim3(gf), dim3(256), 0, st, reinterpret_cast<const f32x4*>(x), mean, invstd, gamma, beta,
                       reinterpret_cast<f32x4*>(y), n4, C / 4, slope, 0, slope_vec);
  else
    hipLaunchKernelGGL(bn_fwd_apply_kernel<false>, dim3(grid1d(n4)), dim3(256), 0, st, reinterpret_cast<const f32x4*>(x), mean, invstd, gamma,
                       beta, reinterpret_cast<f32x4*>(y), n4, C / 4, slope, 0, slope_vec);
  return dlip_launch_status();
}

extern "C" int dlip_colsum_rows_f32(const float* x, float* y, double* workspace, int32_t M, int32_t C,
                                    dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && y && workspace && M > 0 && C > 0 && (C & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int chunks = dlip_bn_rows_chunks(M);
  ColFin fin = {stream_tickets(st, (C + 63) / 64), y, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0.f};
  hipLaunchKernelGGL(col_partial_kernel<2>, dim3((C + 63) / 64, chunks), dim3(256), 0, st, x, nullptr, nullptr, nullptr,
                     nullptr, nullptr, workspace, M, C, 1.f, 0, nullptr, bn_rows_per_part(M), fin);
  if (!fin.ticket) hipLaunchKernelGGL(col_finalize_kernel, dim3((C + 15) / 16), dim3(256), 0, st, workspace, y, nullptr, C, chunks);
  return dlip_launch_status();
}

extern "C" int dlip_meanstd_pool_bwd_f32(const float* x, const float* y, const float* dy, float* dx, int32_t B, int32_t T,
                                         int32_t C, dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && y && dy && dx && B > 0 && T > 1 && C > 0 && (C & 3) == 0);
  DLIP_CHECK_ARG(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0);
  DLIP_CHECK_ARG(B <= 65535);
  const int C4 = C / 4;
  // enough workgroups to fill the chip: ceil(C4 / 256) column blocks x frame chunks x utterances
  int bs = 256;
  for (int cand : {128, 64})
    if ((C4 + cand - 1) / cand * cand < (C4 + bs - 1) / bs * bs) bs = cand;   // C4 = 375 (E-TDNN's 1 500 channels): 3 x 128, not 2 x 256
  const int cb = (C4 + bs - 1) / bs;
  int tsplit = (int)((4096 + (long long)cb * B - 1) / ((long long)cb * B));
  if (tsplit < 1) tsplit = 1;
  if (tsplit > T) tsplit = T;
  const int tchunk = (T + tsplit - 1) / tsplit;
  hipLaunchKernelGGL(meanstd_bwd_kernel<false>, dim3(cb, (T + tchunk - 1) / tchunk, B), dim3(bs), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const f32x4*>(x), y, dy, reinterpret_cast<f32x4*>(dx), T, C4, tchunk, PoolBnB{});
  return dlip_launch_status();
}

extern "C" int dlip_meanstd_pool_bwd_bn_f32(const float* z, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                            float slope, const float* y, const float* dy, float* dx, int32_t B, int32_t T, int32_t C,
                                            dlip_stream_t stream) {
  DLIP_CHECK_ARG(z && mean && invstd && gamma && beta && y && dy && dx && B > 0 && T > 1 && C > 0 && (C & 3) == 0);
  DLIP_CHECK_ARG(((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0);
  DLIP_CHECK_ARG(B <= 65535);
  const int C4 = C / 4;
  int bs = 256;
  for (int cand : {128, 64})
    if ((C4 + cand - 1) / cand * cand < (C4 + bs - 1) / bs * bs) bs = cand;
  const int cb = (C4 + bs - 1) / bs;
  int tsplit = (int)((4096 + (long long)cb * B - 1) / ((long long)cb * B));
  if (tsplit < 1) tsplit = 1;
  if (tsplit > T) tsplit = T;
  const int tchunk = (T + tsplit - 1) / tsplit;
  PoolBnB bn; bn.mean = mean; bn.invstd = invstd; bn.gamma = gamma; bn.beta = beta; bn.slope = slope;
  hipLaunchKernelGGL(meanstd_bwd_kernel<true>, dim3(cb, (T + tchunk - 1) / tchunk, B), dim3(bs), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const f32x4*>(z), y, dy, reinterpret_cast<f32x4*>(dx), T, C4, tchunk, bn);
  return dlip_launch_status();
}

extern "C" int dlip_permute3_f32(const float* x, float* y, int32_t d0, int32_t d1, int32_t d2, int32_t p0, int32_t p1,
                                 int32_t p2, int32_t flip_axis, dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && y && d0 > 0 && d1 > 0 && d2 > 0 && flip_axis >= -1 && flip_axis <= 2);
  DLIP_CHECK_ARG(p0 >= 0 && p0 < 3 && p1 >= 0 && p1 < 3 && p2 >= 0 && p2 < 3 && p0 != p1 && p0 != p2 && p1 != p2);
  hipLaunchKernelGGL(permute3_kernel, dim3(grid1d((long long)d0 * d1 * d2)), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                     y, d0, d1, d2, p0, p1, p2, flip_axis);
  return dlip_launch_status();
}

extern "C" int dlip_pow2_scale_f32(const float* x, float* scale2, int64_t n, float target, dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && scale2 && n > 0 && target > 0.f);
  DLIP_CHECK_ARG((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(scale2, 0, 2 * sizeof(float), st) != hipSuccess) return DLIP_EINVAL;
  const long long blocks = (n / 4 + 255) / 256;
  hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)(blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks))), dim3(256), 0, st, x,
                     reinterpret_cast<unsigned*>(scale2), (long long)n);
  hipLaunchKernelGGL(pow2_finalize_kernel, dim3(1), dim3(1), 0, st, scale2, target);
  return dlip_launch_status();
}

extern "C" int dlip_pow2_lift_f32(const float* x, float* lift, int64_t n, float target, dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && lift && n > 0 && target > 0.f && (reinterpret_cast<uintptr_t>(x) & 15) == 0);
  hipStream_t st = static_cast<hipStream_t>(stream);
  long long blocks = (n / 4 + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
  if (hipMemsetAsync(lift + 2, 0, sizeof(float), st) != hipSuccess) return DLIP_EINVAL;
  hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, reinterpret_cast<unsigned*>(lift) + 2, (long long)n);
  hipLaunchKernelGGL(pow2_finalize_parts_kernel, dim3(1), dim3(256), 0, st, reinterpret_cast<const unsigned*>(lift) + 2, 1, lift, target);
  return dlip_launch_status();
}

extern "C" int dlip_split_pack_scaled_f32(const float* x, float* y, const float* scale, int64_t rows, int32_t C,
                                          dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && y && scale && rows > 0 && C > 0 && (C & 31) == 0);
  const long long n4 = rows * (C / 4);
  hipLaunchKernelGGL(split_pack_scaled_kernel, dim3(grid1d(n4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const f32x4*>(x), y, scale, n4, dlip_range_for(DLIP_ST_PACK));
  return dlip_launch_status();
}

extern "C" int dlip_split_pack_scaled_pad_f32(const float* x, float* y, const float* scale, int64_t rows, int32_t C, int32_t C_pad,
                                              dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && y && scale && rows > 0 && C > 0 && (C & 3) == 0 && C_pad >= C && (C_pad & 31) == 0);
  DLIP_CHECK_ARG((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  const long long n4 = rows * (C_pad / 4);
  hipLaunchKernelGGL(split_pack_scaled_pad_kernel, dim3(grid1d(n4)), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, scale,
                     (long long)rows, C, C_pad, dlip_range_for(DLIP_ST_PACK));
  return dlip_launch_status();
}

extern "C" int dlip_split_weights_rows_f32(const float* w, float* w_split, float* w_scale, int32_t K, int32_t L, dlip_stream_t stream) {
  DLIP_CHECK_ARG(w && w_split && w_scale && K > 0 && L > 0 && (L & 31) == 0);
  hipLaunchKernelGGL(split_weights_rows_kernel, dim3((unsigned)K), dim3(256), 0, static_cast<hipStream_t>(stream), w, w_split, w_scale, L);
  return dlip_launch_status();
}

extern "C" int dlip_split_weights_perm_f32(const float* w_kct, float* w_split, float* w_scale, int32_t K, int32_t C, int32_t T,
                                           int32_t mode, int32_t C_pad, dlip_stream_t stream) {
  DLIP_CHECK_ARG(w_kct && w_split && w_scale && K > 0 && C > 0 && T > 0 && (mode == 0 || mode == 1));
  const int inner = mode == 0 ? C : K;
  if (C_pad <= 0) C_pad = inner;
  DLIP_CHECK_ARG(C_pad >= inner && (C_pad & 31) == 0 && (long long)(mode == 0 ? K : C) * C_pad * T < (1ll << 31));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (mode == 0) hipLaunchKernelGGL(split_weights_perm_kernel<0>, dim3((unsigned)K), dim3(256), 0, st, w_kct, w_split, w_scale, K, C, T, C_pad);
  else hipLaunchKernelGGL(split_weights_perm_kernel<1>, dim3((unsigned)C), dim3(256), 0, st, w_kct, w_split, w_scale, K, C, T, C_pad);
  return dlip_launch_status();
}

static_assert(sizeof(WSplitDesc) == 48, "include/deeplip_hip.h: struct dlip_wsplit_desc");
extern "C" int dlip_split_weights_multi_f32(const void* descs, const int32_t* block_desc, int32_t n_blocks, int32_t max_row_floats,
                                            dlip_stream_t stream) {
  DLIP_CHECK_ARG(descs && block_desc && n_blocks > 0 && (reinterpret_cast<uintptr_t>(descs) & 7) == 0 && max_row_floats >= 0);
  // the staging buffer: the launch's longest row (T * C_pad floats), at most 32 KB (longer rows are read from memory twice); dynamic
  // LDS, so that a launch of short rows keeps more workgroups on a CU
  int stage_floats = max_row_floats <= 0 || max_row_floats > 8192 ? 8192 : (max_row_floats + 31) / 32 * 32;
  hipLaunchKernelGGL(split_weights_multi_kernel, dim3((unsigned)n_blocks), dim3(256), (size_t)stage_floats * 4, static_cast<hipStream_t>(stream),
                     static_cast<const WSplitDesc*>(descs), block_desc, stage_floats);
  return dlip_launch_status();
}

extern "C" int dlip_fill_from_scalar_f32(const float* src, float* y, int32_t n, dlip_stream_t stream) {
  DLIP_CHECK_ARG(src && y && n > 0);
  hipLaunchKernelGGL(fill_from_scalar_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), src, y, n);
  return dlip_launch_status();
}

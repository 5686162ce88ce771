// HBM-bound reductions on channels-last activations: max-pool, global average pool, (masked)
// temporal mean, clip-group mean, mean + unbiased-std statistics pooling.  One thread owns one
// channel (or a float4 of channels) so every wave access is a contiguous 256 B - 1 KiB segment;
// statistics accumulate in fp64 (torch's CPU reductions accumulate float inputs in double).
#include "dlip_common.h"

namespace {

// MaxPool3d((1,3,3), s(1,2,2), p(0,1,1)) == per-frame MaxPool2d(3, 2, 1); padding never wins.
template <bool OSPLIT>
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const f32x4* __restrict__ x, f32x4* __restrict__ y,
                                                           int N, int H, int W, int C4, int Ho, int Wo, DlipRange status) {
  const long long total = (long long)N * Ho * Wo * C4;
  float amax = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C4);
    long long p = i / C4;
    const int wo = (int)(p % Wo); p /= Wo;
    const int ho = (int)(p % Ho);
    const int n = (int)(p / Ho);
    const float ninf = -__builtin_inff();
    f32x4 m = {ninf, ninf, ninf, ninf};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int hi = 2 * ho - 1 + r;
      if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int wi = 2 * wo - 1 + s;
        if ((unsigned)wi >= (unsigned)W) continue;
        const f32x4 v = x[((long long)(n * H + hi) * W + wi) * C4 + c];
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
      }
    }
    if constexpr (OSPLIT) {   // (hi, lo) fp16 pairs per 32-channel block for the split-fp16 conv kernels
      typedef _Float16 h4 __attribute__((ext_vector_type(4)));
      h4 hi, lo;
#pragma unroll
      for (int k = 0; k < 4; ++k) { hi[k] = (_Float16)m[k]; lo[k] = (_Float16)(m[k] - (float)hi[k]); amax = fmaxf(amax, fabsf(m[k])); }
      float* b = reinterpret_cast<float*>(y) + (i >> 3) * 32;
      const int q = (int)(i & 7);
      *reinterpret_cast<h4*>(b + q * 2) = hi;
      *reinterpret_cast<h4*>(b + 16 + q * 2) = lo;
    } else {
      y[i] = m;
    }
  }
  if constexpr (OSPLIT) dlip_report_range_block(amax, status);
}

// y[n,c] = mean over hw of x[n,hw,c]   (sequential fp32 sum, then one divide: AdaptiveAvgPool2d(1))
__global__ __launch_bounds__(256) void avgpool_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                      int N, int HW, int C) {
  const long long total = (long long)N * C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const long long n = i / C;
    const float* p = x + n * HW * C + c;
    float s = 0.f;
    for (int h = 0; h < HW; ++h) s += p[(long long)h * C];
    y[i] = s / (float)HW;
  }
}

__global__ __launch_bounds__(256) void time_mean_kernel(const float* __restrict__ x, const DlipLen len,
                                                        float* __restrict__ y, int B, int T, int C, int ldx) {
  const long long total = (long long)B * C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const int b = (int)(i / C);
    const int L = dlip_valid_rows(len, b, T);
    const float* p = x + (long long)b * T * ldx + c;
    double s = 0.0;
    for (int t = 0; t < L; ++t) s += (double)p[(long long)t * ldx];
    y[i] = (float)(s / (double)L);
  }
}

__global__ __launch_bounds__(256) void group_mean_kernel(const float* __restrict__ x, const int32_t* __restrict__ ptr,
                                                         float* __restrict__ y, int U, int C) {
  const long long total = (long long)U * C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const int u = (int)(i / C);
    const int g0 = ptr[u], g1 = ptr[u + 1];
    float s = 0.f;  // the reference accumulates the clip means in fp32 (`em += ...`, train_fusion.py:274)
    for (int g = g0; g < g1; ++g) s += x[(long long)g * C + c];
    y[i] = s / (float)(g1 - g0);
  }
}

// y[b, c] = mean_t x[b,t,c];  y[b, C + c] = sqrt( sum_t (x - mean)^2 / (T - 1) )
// One workgroup per (utterance, 64-channel block): 16 lanes x float4 span the block, 16 row groups
// split T, so every thread streams T/16 independent 16-B loads; sums and sums of squares are kept
// in fp64 (the inputs are fp32, so sum x^2 - (sum x)^2 / T loses nothing a two-pass fp32 result has)
// and the 16 partial rows meet in LDS.  SPLIT writes the [B, ldy] result as (hi, lo) fp16 pairs (the
// split activation format; ldy = 2C rounded up to 32, padding zeroed) for the LDS-DMA GEMM behind it.
// `len` (ragged batches: utterance b is valid for its first len[b] + len_add of the T frames, the rest is padding): the
// statistics cover the valid frames only, as the reference's one-utterance-at-a-time loop computes them (train_fusion.py:334-338).
// (ABI 48) BN: x is the RAW output z of the last TDNN convolution and every loaded value becomes lrelu((z - mean) invstd gamma + beta, slope)
// first (bn_fwd_apply_kernel's expression: the same bits) -- the train-mode BatchNorm + LeakyReLU in front of the pooling, whose activated
// tensor ([B,T,1500]: 460 MB at B = 256) is then never stored.
struct PoolBn {
  const float* mean = nullptr;
  const float* invstd = nullptr;
  const float* gamma = nullptr;
  const float* beta = nullptr;
  float slope = 1.f;
};
template <bool SPLIT, bool BN = false>
__global__ __launch_bounds__(256) void meanstd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                      int Tpad, int C, int ldy, const DlipLen len, DlipRange status, const PoolBn bn = PoolBn{}) {
  __shared__ double part[16][64][2];
  const int b = blockIdx.y, c0 = blockIdx.x * 64;
  const int T = dlip_valid_rows(len, b, Tpad);
  const int lx = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int c = c0 + lx * 4;
  float amax = 0.f;
  double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  if (c < C) {   // C % 4 == 0: a float4 is all inside or all outside
    const float* p = x + (long long)b * Tpad * C + c;
    f32x4 mu = {0, 0, 0, 0}, is = mu, ga = mu, be = mu;
    if constexpr (BN) {
      mu = *reinterpret_cast<const f32x4*>(bn.mean + c); is = *reinterpret_cast<const f32x4*>(bn.invstd + c);
      ga = *reinterpret_cast<const f32x4*>(bn.gamma + c); be = *reinterpret_cast<const f32x4*>(bn.beta + c);
    }
    for (int t = g; t < T; t += 16) {
      f32x4 v = *reinterpret_cast<const f32x4*>(p + (long long)t * C);
      if constexpr (BN) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float u = (v[k] - mu[k]) * is[k] * ga[k] + be[k];
          v[k] = u >= 0.f ? u : u * bn.slope;
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) { s[k] += (double)v[k]; q[k] += (double)v[k] * (double)v[k]; }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) { part[g][lx * 4 + k][0] = s[k]; part[g][lx * 4 + k][1] = q[k]; }
  __syncthreads();
  if (threadIdx.x < 64 && c0 + (int)threadIdx.x < C) {
    const int cc = c0 + threadIdx.x;
    double ss = 0.0, qq = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) { ss += part[i][threadIdx.x][0]; qq += part[i][threadIdx.x][1]; }
    const double mean = ss / (double)T;
    double var = (qq - ss * mean) / (double)(T - 1);   // T == 1 -> NaN, as torch.std
    if (var < 0.0) var = 0.0;
    const float out[2] = {(float)mean, (float)sqrt(var)};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int col = h * C + cc;
      if (SPLIT) {
        const _Float16 hi = (_Float16)out[h], lo = (_Float16)(out[h] - (float)hi);
        _Float16* blk = reinterpret_cast<_Float16*>(y + (long long)b * ldy + (col & ~31));
        blk[col & 31] = hi;
        blk[32 + (col & 31)] = lo;
        amax = fmaxf(amax, fabsf(out[h]));
        if (!(fabsf(out[h]) < DLIP_F16_OVERFLOW)) amax = __builtin_inff();   // (a NaN too: fmaxf would drop it)
      } else {
        y[(long long)b * ldy + col] = out[h];
      }
    }
  }
  if (SPLIT && blockIdx.x == 0 && (int)threadIdx.x < ldy - 2 * C) {   // zero the channel padding (< 32 values)
    const int col = 2 * C + threadIdx.x;
    _Float16* blk = reinterpret_cast<_Float16*>(y + (long long)b * ldy + (col & ~31));
    blk[col & 31] = (_Float16)0.f;
    blk[32 + (col & 31)] = (_Float16)0.f;
  }
  if (SPLIT) dlip_report_range(amax, status);
}

static inline unsigned grid_for(long long total) {
  long long g = (total + 255) / 256;
  if (g > 256 * 8) g = 256 * 8;  // 8 workgroups per CU, grid-stride the rest
  if (g < 1) g = 1;
  return (unsigned)g;
}

}  // namespace

extern "C" int dlip_maxpool3x3s2_nhwc_f32(const float* x, float* y, int32_t N, int32_t H, int32_t W, int32_t C,
                                          int32_t out_split, dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && y && N > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0 && (!out_split || (C & 31) == 0));
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long long total = (long long)N * Ho * Wo * (C / 4);
  if (out_split)
    hipLaunchKernelGGL(maxpool3x3s2_kernel<true>, dim3(grid_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const f32x4*>(x), reinterpret_cast<f32x4*>(y), N, H, W, C / 4, Ho, Wo,
                       dlip_range_for(DLIP_ST_PACK));
  else
    hipLaunchKernelGGL(maxpool3x3s2_kernel<false>, dim3(grid_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const f32x4*>(x), reinterpret_cast<f32x4*>(y), N, H, W, C / 4, Ho, Wo, DlipRange{});
  return dlip_launch_status();
}

extern "C" int dlip_avgpool_nhwc_f32(const float* x, float* y, int32_t N, int32_t HW, int32_t C,
                                     dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && y && N > 0 && HW > 0 && C > 0);
  hipLaunchKernelGGL(avgpool_kernel, dim3(grid_for((long long)N * C)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, y, N, HW, C);
  return dlip_launch_status();
}

extern "C" int dlip_time_mean_f32(const float* x, const int32_t* len, int32_t len_add, float* y, int32_t B, int32_t T, int32_t C,
                                  int32_t ldx, dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && y && B > 0 && T > 0 && C > 0 && ldx >= C);
  DlipLen l; l.len = len; l.mul = 1; l.add = len_add;
  hipLaunchKernelGGL(time_mean_kernel, dim3(grid_for((long long)B * C)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, l, y, B, T, C, ldx);
  return dlip_launch_status();
}

extern "C" int dlip_group_mean_f32(const float* x, const int32_t* group_ptr, float* y, int32_t U, int32_t C,
                                   dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && group_ptr && y && U > 0 && C > 0);
  hipLaunchKernelGGL(group_mean_kernel, dim3(grid_for((long long)U * C)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, group_ptr, y, U, C);
  return dlip_launch_status();
}

extern "C" int dlip_meanstd_pool_f32(const float* x, const int32_t* len, int32_t len_add, float* y, int32_t B, int32_t T, int32_t C,
                                     int32_t out_split, dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && y && B > 0 && B <= 65535 && T > 0 && C > 0 && (C & 3) == 0);
  DLIP_CHECK_ARG((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  const dim3 grid((C + 63) / 64, B);
  DlipLen l; l.len = len; l.mul = 1; l.add = len_add;
  if (out_split)
    hipLaunchKernelGGL(meanstd_kernel<true>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), x, y, T, C,
                       (2 * C + 31) / 32 * 32, l, dlip_range_for(DLIP_ST_POOL));
  else
    hipLaunchKernelGGL(meanstd_kernel<false>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), x, y, T, C, 2 * C, l,
                       DlipRange{});
  return dlip_launch_status();
}

extern "C" int dlip_meanstd_pool_bn_f32(const float* z, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                        float slope, float* y, int32_t B, int32_t T, int32_t C, dlip_stream_t stream) {
  DLIP_CHECK_ARG(z && mean && invstd && gamma && beta && y && B > 0 && B <= 65535 && T > 0 && C > 0 && (C & 3) == 0);
  DLIP_CHECK_ARG((reinterpret_cast<uintptr_t>(z) & 15) == 0);
  DlipLen l; l.len = nullptr; l.mul = 1; l.add = 0;
  PoolBn bn; bn.mean = mean; bn.invstd = invstd; bn.gamma = gamma; bn.beta = beta; bn.slope = slope;
  hipLaunchKernelGGL((meanstd_kernel<false, true>), dim3((C + 63) / 64, B), dim3(256), 0, static_cast<hipStream_t>(stream), z, y, T, C, 2 * C, l,
                     DlipRange{}, bn);
  return dlip_launch_status();
}

namespace {
// AttentiveStatPooling tail (models/audio_models/pooling.py:87-107) on N-T-C activations:
//   e[t]   = relu(hidden[b,t,:]) . v + k            (hidden = x W^T + b comes from the GEMM kernel)
//   alpha  = softmax_t(e)
//   mean_c = sum_t alpha[t] x[b,t,c];  std_c = sqrt(sum_t alpha[t] x[b,t,c]^2 - mean_c^2)
// One workgroup per utterance: phase 1 one wave per frame for e[t] (LDS), phase 2 softmax over T,
// phase 3 one thread per channel (coalesced) with fp64 accumulation.
__global__ __launch_bounds__(256) void attentive_stat_kernel(const float* __restrict__ x, const float* __restrict__ hidden,
                                                             const float* __restrict__ v, const float* __restrict__ kk,
                                                             float* __restrict__ y, float* __restrict__ alpha_out, int Tpad, int C,
                                                             int Hd, const DlipLen len) {
  extern __shared__ float alpha[];   // [T]
  __shared__ float red[2];
  const int b = blockIdx.x;
  const int T = dlip_valid_rows(len, b, Tpad);          // ragged batches: attention and statistics over the valid frames only
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* hb = hidden + (long long)b * Tpad * Hd;
  for (int t = wave; t < T; t += 4) {
    double s = 0.0;
    for (int h = lane; h < Hd; h += 64) s += (double)fmaxf(hb[(long long)t * Hd + h], 0.f) * (double)v[h];
    s = dlip_wave_sum_f64(s);
    if (lane == 0) alpha[t] = (float)s + kk[0];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float mx = -__builtin_inff();
    for (int t = 0; t < T; ++t) mx = fmaxf(mx, alpha[t]);
    double se = 0.0;
    for (int t = 0; t < T; ++t) se += exp((double)(alpha[t] - mx));
    red[0] = mx;
    red[1] = (float)se;
  }
  __syncthreads();
  const float mx = red[0];
  const double se = red[1];
  for (int t = threadIdx.x; t < T; t += 256) alpha[t] = (float)(exp((double)(alpha[t] - mx)) / se);
  __syncthreads();
  if (alpha_out != nullptr)          // the attention weights, kept for the backward pass (zeros behind a ragged utterance's end)
    for (int t = threadIdx.x; t < Tpad; t += 256) alpha_out[(long long)b * Tpad + t] = t < T ? alpha[t] : 0.f;
  const float* xb = x + (long long)b * Tpad * C;
  for (int c = threadIdx.x; c < C; c += 256) {
    double m = 0.0, q = 0.0;
    for (int t = 0; t < T; ++t) {
      const double xv = xb[(long long)t * C + c], a = alpha[t];
      m += a * xv;
      q += a * xv * xv;
    }
    y[(long long)b * 2 * C + c] = (float)m;
    y[(long long)b * 2 * C + C + c] = (float)sqrt(q - m * m);
  }
}
}  // namespace

extern "C" int dlip_attentive_stat_pool_f32(const float* x, const float* hidden, const float* v, const float* k,
                                            const int32_t* len, int32_t len_add, float* y, float* alpha_out, int32_t B, int32_t T,
                                            int32_t C, int32_t Hd, dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && hidden && v && k && y && B > 0 && T > 0 && C > 0 && Hd > 0 && T <= 16000);
  DlipLen l; l.len = len; l.mul = 1; l.add = len_add;
  hipLaunchKernelGGL(attentive_stat_kernel, dim3(B), dim3(256), (size_t)T * sizeof(float),
                     static_cast<hipStream_t>(stream), x, hidden, v, k, y, alpha_out, T, C, Hd, l);
  return dlip_launch_status();
}

namespace {
// Backward of the AttentiveStatPooling tail (models/audio_models/pooling.py:87-107; the forward above) for one utterance per
// workgroup.  With u = q - m^2, s = sqrt(u):   g_q = ds / (2 s),  g_m = dm - 2 m g_q   (torch's own chain through sqrt and the square)
//   dx[t,c]     = alpha_t (g_m[c] + 2 x[t,c] g_q[c])                     the statistics' direct path
//   dalpha_t    = sum_c (g_m[c] x[t,c] + g_q[c] x[t,c]^2)
//   de_t        = alpha_t (dalpha_t - sum_t' alpha_t' dalpha_t')        softmax over the utterance's valid frames
//   dhidden[t,j]= de_t v_j [hidden[t,j] > 0]                            -> W, b and the second path into x, through the GEMM's backward
//   rde[t,j]    = de_t relu(hidden[t,j])                                column sums = dv;  sum(de) = dk
// fp64 accumulation wherever a sum runs over channels or frames.  LDS: g_m [C] | g_q [C] | dalpha [Tpad].
__global__ __launch_bounds__(256) void attentive_stat_bwd_kernel(const float* __restrict__ x, const float* __restrict__ hidden,
                                                                 const float* __restrict__ v, const float* __restrict__ alpha,
                                                                 const float* __restrict__ y, const float* __restrict__ dy,
                                                                 float* __restrict__ dx, float* __restrict__ dhidden,
                                                                 float* __restrict__ rde, float* __restrict__ de_out, int Tpad, int C,
                                                                 int Hd, const DlipLen len) {
  extern __shared__ float sm[];
  float* gm = sm;
  float* gq = sm + C;
  float* dal = sm + 2 * C;          // [Tpad]: dalpha, then de
  __shared__ double red[4];
  const int b = blockIdx.x;
  const int T = dlip_valid_rows(len, b, Tpad);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* xb = x + (long long)b * Tpad * C;
  const float* ab = alpha + (long long)b * Tpad;
  for (int c = threadIdx.x; c < C; c += 256) {
    const double m = y[(long long)b * 2 * C + c], s = y[(long long)b * 2 * C + C + c];
    const double dm = dy[(long long)b * 2 * C + c], ds = dy[(long long)b * 2 * C + C + c];
    const double q = ds / (2.0 * s);
    gq[c] = (float)q;
    gm[c] = (float)(dm - 2.0 * m * q);
  }
  __syncthreads();
  for (int t = wave; t < T; t += 4) {
    double a = 0.0;
    for (int c = lane; c < C; c += 64) {
      const double xv = xb[(long long)t * C + c];
      a += xv * ((double)gm[c] + (double)gq[c] * xv);
    }
    a = dlip_wave_sum_f64(a);
    if (lane == 0) dal[t] = (float)a;
  }
  __syncthreads();
  {
    double part = 0.0;
    for (int t = threadIdx.x; t < T; t += 256) part += (double)ab[t] * (double)dal[t];
    part = dlip_wave_sum_f64(part);
    if (lane == 0) red[wave] = part;
  }
  __syncthreads();
  const double dot = red[0] + red[1] + red[2] + red[3];
  for (int t = threadIdx.x; t < Tpad; t += 256) {
    const float de = t < T ? (float)((double)ab[t] * ((double)dal[t] - dot)) : 0.f;
    dal[t] = de;
    de_out[(long long)b * Tpad + t] = de;
  }
  __syncthreads();
  float* dxb = dx + (long long)b * Tpad * C;
  for (long long i = threadIdx.x; i < (long long)Tpad * C; i += 256) {
    const int t = (int)(i / C), c = (int)(i - (long long)t * C);
    dxb[i] = t < T ? ab[t] * (gm[c] + 2.f * xb[i] * gq[c]) : 0.f;
  }
  const float* hb = hidden + (long long)b * Tpad * Hd;
  float* dhb = dhidden + (long long)b * Tpad * Hd;
  float* rb = rde + (long long)b * Tpad * Hd;
  for (long long i = threadIdx.x; i < (long long)Tpad * Hd; i += 256) {
    const int t = (int)(i / Hd), j = (int)(i - (long long)t * Hd);
    const float h = hb[i], de = dal[t];
    dhb[i] = (t < T && h > 0.f) ? de * v[j] : 0.f;
    rb[i] = t < T ? de * fmaxf(h, 0.f) : 0.f;
  }
}
}  // namespace

extern "C" int dlip_attentive_stat_pool_bwd_f32(const float* x, const float* hidden, const float* v, const float* alpha, const float* y,
                                                const float* dy, const int32_t* len, int32_t len_add, float* dx, float* dhidden,
                                                float* rde, float* de, int32_t B, int32_t T, int32_t C, int32_t Hd,
                                                dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && hidden && v && alpha && y && dy && dx && dhidden && rde && de && B > 0 && T > 0 && C > 0 && Hd > 0);
  const size_t lds = ((size_t)2 * C + T) * sizeof(float);
  DLIP_CHECK_ARG(lds <= 60 * 1024);
  DlipLen l; l.len = len; l.mul = 1; l.add = len_add;
  hipLaunchKernelGGL(attentive_stat_bwd_kernel, dim3(B), dim3(256), lds, static_cast<hipStream_t>(stream), x, hidden, v, alpha, y, dy,
                     dx, dhidden, rde, de, T, C, Hd, l);
  return dlip_launch_status();
}

namespace {
// Finisher of the pooled convolution epilogue (conv_igemm_f16x3_dma.hip, EPI 2): partials [tiles_m][4][Kp] fp64 =
// per tile the column sums {sum, sumsq} of the rows before the tile's group boundary (segment 0) and after it
// (segment 1).  Group g = rows [g Gs, (g+1) Gs) touches tiles g Gs / BM .. ((g+1) Gs - 1) / BM; in each it is
// segment 0 if the tile's first row belongs to g, else segment 1.  Tiles are added in row order.
template <int MODE, bool SPLIT>
__global__ __launch_bounds__(256) void pool_finish_kernel(const double* __restrict__ part, float* __restrict__ y, long long M,
                                                          int K, int Kp, int BM, int Gs, int G, int ldy, const DlipLen len,
                                                          DlipRange status) {
  const long long total = (long long)G * K;
  float amax = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int k = (int)(i % K);
    const long long g = i / K;
    const long long r0 = g * Gs;
    long long r1 = r0 + Gs;
    if (r1 > M) r1 = M;
    double s = 0.0, q = 0.0;
    for (long long tm = r0 / BM; tm <= (r1 - 1) / BM; ++tm) {
      const int seg = (tm * BM) / Gs == g ? 0 : 1;
      const double* p = part + ((size_t)tm * 4 + 2 * seg) * Kp + k;
      s += p[0];
      q += p[Kp];
    }
    const double n = (double)dlip_valid_rows(len, g, (int)(r1 - r0));   // ragged: the epilogue summed the group's valid rows only
    const double mean = s / n;
    if (MODE == 0) {
      y[g * ldy + k] = (float)mean;
    } else {
      double var = (q - s * mean) / (n - 1.0);   // n == 1 -> NaN, as torch.std
      if (var < 0.0) var = 0.0;
      const float out[2] = {(float)mean, (float)sqrt(var)};
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int col = h * K + k;
        if (SPLIT) {
          const _Float16 hi = (_Float16)out[h], lo = (_Float16)(out[h] - (float)hi);
          _Float16* blk = reinterpret_cast<_Float16*>(y + g * ldy + (col & ~31));
          blk[col & 31] = hi;
          blk[32 + (col & 31)] = lo;
          amax = fmaxf(amax, fabsf(out[h]));
        } else {
          y[g * ldy + col] = out[h];
        }
      }
    }
  }
  if (SPLIT) {
    const long long pad_total = (long long)G * (ldy - 2 * K);   // zero the channel padding (< 32 values per row)
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < pad_total; i += (long long)gridDim.x * 256) {
      const long long g = i / (ldy - 2 * K);
      const int col = 2 * K + (int)(i % (ldy - 2 * K));
      _Float16* blk = reinterpret_cast<_Float16*>(y + g * ldy + (col & ~31));
      blk[col & 31] = (_Float16)0.f;
      blk[32 + (col & 31)] = (_Float16)0.f;
    }
    dlip_report_range_block(amax, status);
  }
}
}  // namespace

extern "C" int dlip_pool_finish_f32(const double* partials, int64_t M, int32_t K, int32_t tile_rows, int32_t group_rows,
                                    const int32_t* group_len, int32_t len_mul, int32_t len_add,
                                    int32_t mode, int32_t out_split, float* y, dlip_stream_t stream) {
  DLIP_CHECK_ARG(partials && y && M > 0 && K > 0 && tile_rows > 0 && group_rows >= tile_rows && (mode == 0 || mode == 1));
  DLIP_CHECK_ARG(!(out_split && mode == 0));
  const int Kp = (K + 127) / 128 * 128;   // the pooled epilogue exists on the 128-column tiles only
  const long long G = (M + group_rows - 1) / group_rows;
  DLIP_CHECK_ARG(G <= 0x7FFFFFFF);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned grid = grid_for(G * K);
  const DlipRange status = (mode == 1 && out_split) ? dlip_range_for(DLIP_ST_POOL) : DlipRange{};   // only a split output is a producer
  DlipLen l; l.len = group_len; l.mul = len_mul; l.add = len_add;
  if (mode == 0)
    hipLaunchKernelGGL((pool_finish_kernel<0, false>), dim3(grid), dim3(256), 0, st, partials, y, (long long)M, K, Kp, tile_rows,
                       group_rows, (int)G, K, l, status);
  else if (out_split)
    hipLaunchKernelGGL((pool_finish_kernel<1, true>), dim3(grid), dim3(256), 0, st, partials, y, (long long)M, K, Kp, tile_rows,
                       group_rows, (int)G, (2 * K + 31) / 32 * 32, l, status);
  else
    hipLaunchKernelGGL((pool_finish_kernel<1, false>), dim3(grid), dim3(256), 0, st, partials, y, (long long)M, K, Kp, tile_rows,
                       group_rows, (int)G, 2 * K, l, status);
  return dlip_launch_status();
}

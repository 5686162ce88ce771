// Backward / train-mode kernels for the TRAINABLE part of the fusion pipeline (config C5): the
// reference freezes both encoders (train_fusion.py:198-201) and trains only the fusion head
// (Linear - BatchNorm1d - LeakyReLU - Linear, models/fusion_models/model_fusion.py:10-24) and the
// criterion (CrossEntropy / LMCL, models/audio_models/loss.py:6-51) on [B<=256, <=1536] tensors.
// These are tiny, latency-bound ops: one thread (or one wave) per column / row, fp64 accumulation,
// deterministic summation order (no atomics).  The gradient GEMMs (57-wide / batch-wide reductions)
// use the small general GEMM at the end of this file; torch.autograd is only the tape
// (deeplip_amd/autograd.py).
#include "dlip_common.h"

namespace {

// BatchNorm1d, training mode, x [M,C]: per-channel batch mean / biased variance (normalisation) and
// running-stat update with the unbiased variance (torch semantics).  One thread per channel.
__global__ __launch_bounds__(256) void bn1d_train_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ y,
                                                             float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                             float* __restrict__ running_mean, float* __restrict__ running_var,
                                                             int M, int C, float momentum, float eps, float slope) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double s = 0.0;
  for (int m = 0; m < M; ++m) s += (double)x[(long long)m * C + c];
  const double mean = s / M;
  double q = 0.0;
  for (int m = 0; m < M; ++m) {
    const double d = (double)x[(long long)m * C + c] - mean;
    q += d * d;
  }
  const double var = q / M;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float mu = (float)mean;
  save_mean[c] = mu;
  save_invstd[c] = invstd;
  if (running_mean) {
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(M > 1 ? q / (M - 1) : var);
  }
  const float g = gamma[c], b = beta[c];
  for (int m = 0; m < M; ++m) {
    const long long i = (long long)m * C + c;
    const float v = (x[i] - mu) * invstd * g + b;
    y[i] = v >= 0.f ? v : v * slope;   // slope = 1: plain BatchNorm
  }
}

// dx = gamma*invstd/M * (M*dy - sum(dy) - xhat*sum(dy*xhat));  dgamma = sum(dy*xhat); dbeta = sum(dy)
__global__ __launch_bounds__(256) void bn1d_train_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                             const float* __restrict__ save_mean,
                                                             const float* __restrict__ save_invstd,
                                                             const float* __restrict__ gamma, float* __restrict__ dx,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                             int M, int C) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float mu = save_mean[c], is = save_invstd[c];
  double s1 = 0.0, s2 = 0.0;
  for (int m = 0; m < M; ++m) {
    const long long i = (long long)m * C + c;
    const double g = dy[i];
    s1 += g;
    s2 += g * (double)((x[i] - mu) * is);
  }
  dgamma[c] = (float)s2;
  dbeta[c] = (float)s1;
  const double k = (double)gamma[c] * is / M;
  for (int m = 0; m < M; ++m) {
    const long long i = (long long)m * C + c;
    const double xh = (x[i] - mu) * is;
    dx[i] = (float)(k * ((double)M * dy[i] - s1 - xh * s2));
  }
}

__global__ __launch_bounds__(256) void lrelu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                        float* __restrict__ dx, long long n, float slope) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    dx[i] = y[i] >= 0.f ? dy[i] : dy[i] * slope;   // slope > 0: sign(y) == sign(x)
}

__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, float* __restrict__ y, int M, int C) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double s = 0.0;
  for (int m = 0; m < M; ++m) s += (double)x[(long long)m * C + c];
  y[c] = (float)s;
}

// L1 norm of a weight matrix (the LMCL regulariser 1e-5 * ||W||_1, models/audio_models/loss.py:49-50) and its gradient
// d(out)/dW = gscale * sign(W).  One workgroup: the tensors are the criterion's [n_spk, 512] (29 k elements); fp64 sum in a
// fixed order (lane-strided partials, shuffle tree, wave order): deterministic.
__global__ __launch_bounds__(256) void l1_sum_kernel(const float* __restrict__ w, float* __restrict__ out, long long n) {
  __shared__ double red[4];
  double s = 0.0;
  for (long long i = threadIdx.x; i < n; i += 256) s += (double)fabsf(w[i]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (float)(((red[0] + red[1]) + red[2]) + red[3]);
}

__global__ __launch_bounds__(256) void l1_sign_kernel(const float* __restrict__ w, const float* __restrict__ gscale, float* __restrict__ dw,
                                                      float coef, long long n) {
  const float g = coef * (gscale ? gscale[0] : 1.f);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float v = w[i];
    dw[i] = v > 0.f ? g : (v < 0.f ? -g : 0.f);
  }
}

// d loss / d logits for loss = mean_b CE(z_b, label_b), z = scale*(logits - margin*onehot) + 1e-8:
//   dlogits[b,k] = gscale * scale * (softmax(z_b)[k] - [k == label_b]) / B.   One wave per row.
__global__ __launch_bounds__(256) void margin_ce_bwd_kernel(const float* __restrict__ logits,
                                                            const long long* __restrict__ labels,
                                                            float* __restrict__ dlogits, int B, int K, float scale,
                                                            float margin, float gscale_host,
                                                            const float* __restrict__ gscale_dev) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const float gscale = gscale_dev ? gscale_host * gscale_dev[0] : gscale_host;
  const float* p = logits + (long long)b * K;
  const int lab = (int)labels[b];
  float mx = -__builtin_inff();
  for (int k = lane; k < K; k += 64) mx = fmaxf(mx, scale * (p[k] - (k == lab ? margin : 0.f)) + 1e-8f);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  double se = 0.0;
  for (int k = lane; k < K; k += 64) se += exp((double)(scale * (p[k] - (k == lab ? margin : 0.f)) + 1e-8f - mx));
  se = dlip_wave_sum_f64(se);
  const double f = (double)gscale * scale / B;
  for (int k = lane; k < K; k += 64) {
    const double z = scale * (p[k] - (k == lab ? margin : 0.f)) + 1e-8f;
    const double sm = exp(z - mx) / se;
    dlogits[(long long)b * K + k] = (float)(f * (sm - (k == lab ? 1.0 : 0.0)));
  }
}

// Backward of y = x / max(||x||, eps):  dx = (dy - y*(y.dy)) / max(||x||, eps).  One wave per row.
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                         float* __restrict__ dx, int U, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= U) return;
  const float* px = x + (long long)r * D;
  const float* pg = dy + (long long)r * D;
  double s = 0.0, dot = 0.0;
  for (int i = lane; i < D; i += 64) {
    s += (double)px[i] * px[i];
    dot += (double)px[i] * pg[i];
  }
  s = dlip_wave_sum_f64(s);
  dot = dlip_wave_sum_f64(dot);
  const double nrm = fmax(sqrt(s), (double)eps);
  for (int i = lane; i < D; i += 64)
    dx[(long long)r * D + i] = (float)(((double)pg[i] - (double)px[i] * dot / (nrm * nrm)) / nrm);
}

}  // namespace

// Additive angular margin (ArcFace / AAM-softmax; the reference leaves `AAMSoftmax` an empty stub, loss.py:62-67,
// and the north star names it): on the target column of cosine logits, cos(theta) -> cos(theta + m) where
// theta + m stays below pi (cos > th = cos(pi - m)), else the CosFace-style fallback cos - m sin(m) (the usual
// ArcFace recipe; easy_margin: apply only where cos > 0).  mode 0: y = modified logits; mode 1 (backward):
// y = dL/dlogits given g = dL/dy, i.e. g scaled by d cos(theta + m)/d cos(theta) = cos m + sin m cos/sqrt(1 - cos^2)
// on the target column.
__global__ __launch_bounds__(256) void aam_margin_kernel(const float* __restrict__ logits, const long long* __restrict__ labels,
                                                         const float* __restrict__ g, float* __restrict__ y, int B, int K,
                                                         float cos_m, float sin_m, float th, float mm, int easy, int mode) {
  const long long n = (long long)B * K;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int b = (int)(i / K), k = (int)(i - (long long)b * K);
    const float c = logits[i];
    const bool tgt = labels[b] == k;
    if (mode == 0) {
      float v = c;
      if (tgt) {
        const float sine = sqrtf(fmaxf(0.f, 1.f - c * c));
        const float phi = c * cos_m - sine * sin_m;
        v = easy ? (c > 0.f ? phi : c) : (c > th ? phi : c - mm);
      }
      y[i] = v;
    } else {
      float d = 1.f;
      if (tgt) {
        const bool use_phi = easy ? c > 0.f : c > th;
        if (use_phi) d = cos_m + sin_m * c / sqrtf(fmaxf(1e-12f, 1.f - c * c));
      }
      y[i] = g[i] * d;
    }
  }
}

extern "C" int dlip_bn1d_train_fwd_f32(const float* x, const float* gamma, const float* beta, float* y,
                                       float* save_mean, float* save_invstd, float* running_mean,
                                       float* running_var, int32_t M, int32_t C, float momentum, float eps,
                                       float slope, dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && gamma && beta && y && save_mean && save_invstd && M > 0 && C > 0);
  DLIP_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr));
  hipLaunchKernelGGL(bn1d_train_fwd_kernel, dim3((C + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                     gamma, beta, y, save_mean, save_invstd, running_mean, running_var, M, C, momentum, eps, slope);
  return dlip_launch_status();
}

extern "C" int dlip_bn1d_train_bwd_f32(const float* dy, const float* x, const float* save_mean,
                                       const float* save_invstd, const float* gamma, float* dx, float* dgamma,
                                       float* dbeta, int32_t M, int32_t C, dlip_stream_t stream) {
  DLIP_CHECK_ARG(dy && x && save_mean && save_invstd && gamma && dx && dgamma && dbeta && M > 0 && C > 0);
  hipLaunchKernelGGL(bn1d_train_bwd_kernel, dim3((C + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), dy,
                     x, save_mean, save_invstd, gamma, dx, dgamma, dbeta, M, C);
  return dlip_launch_status();
}

extern "C" int dlip_lrelu_bwd_f32(const float* dy, const float* y, float* dx, int64_t n, float slope,
                                  dlip_stream_t stream) {
  DLIP_CHECK_ARG(dy && y && dx && n > 0 && slope > 0.f);
  long long g = (n + 255) / 256;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(lrelu_bwd_kernel, dim3((unsigned)g), dim3(256), 0, static_cast<hipStream_t>(stream), dy, y, dx,
                     (long long)n, slope);
  return dlip_launch_status();
}

extern "C" int dlip_l1_sum_f32(const float* w, float* out, int64_t n, dlip_stream_t stream) {
  DLIP_CHECK_ARG(w && out && n > 0);
  hipLaunchKernelGGL(l1_sum_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), w, out, (long long)n);
  return dlip_launch_status();
}

extern "C" int dlip_l1_sign_f32(const float* w, const float* grad_scale_dev, float* dw, float coef, int64_t n, dlip_stream_t stream) {
  DLIP_CHECK_ARG(w && dw && n > 0);
  const long long blocks = (n + 255) / 256;
  hipLaunchKernelGGL(l1_sign_kernel, dim3((unsigned)(blocks > 1024 ? 1024 : blocks)), dim3(256), 0, static_cast<hipStream_t>(stream), w,
                     grad_scale_dev, dw, coef, (long long)n);
  return dlip_launch_status();
}

extern "C" int dlip_colsum_f32(const float* x, float* y, int32_t M, int32_t C, dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && y && M > 0 && C > 0);
  hipLaunchKernelGGL(colsum_kernel, dim3((C + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, M, C);
  return dlip_launch_status();
}

extern "C" int dlip_margin_ce_bwd_f32(const float* logits, const int64_t* labels, float* dlogits, int32_t B,
                                      int32_t K, float scale, float margin, float grad_scale,
                                      const float* grad_scale_dev, dlip_stream_t stream) {
  DLIP_CHECK_ARG(logits && labels && dlogits && B > 0 && K > 0);
  hipLaunchKernelGGL(margin_ce_bwd_kernel, dim3((B + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), logits,
                     reinterpret_cast<const long long*>(labels), dlogits, B, K, scale, margin, grad_scale, grad_scale_dev);
  return dlip_launch_status();
}

extern "C" int dlip_l2_normalize_bwd_f32(const float* x, const float* dy, float* dx, int32_t U, int32_t D,
                                         float eps, dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && dy && dx && U > 0 && D > 0);
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((U + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), x, dy, dx,
                     U, D, eps);
  return dlip_launch_status();
}

namespace {
// Small general GEMM for the head's backward pass: C[M,N] = op(A)[M,K] * op(B)[K,N], row-major,
// op = identity or transpose, arbitrary sizes (the MFMA conv kernel needs K % 4 == 0 and K-major
// operands; gradients want 57-wide and batch-wide reductions).  32x32 tile, fp32 accumulate in a
// fixed k order (deterministic).  <= 0.1 GFLOP per call in config C5: latency-bound.
__global__ __launch_bounds__(256) void gemm_small_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                         float* __restrict__ C, int M, int N, int K, int ta, int tb) {
  __shared__ float As[32][33], Bs[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < K; k0 += 32) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = ty + 8 * j;
      const int m = m0 + r, k = k0 + tx;          // As[r][tx] = op(A)[m0+r][k0+tx]
      As[r][tx] = (m < M && k < K) ? (ta ? A[(long long)k * M + m] : A[(long long)m * K + k]) : 0.f;
      const int kb = k0 + r, n = n0 + tx;         // Bs[r][tx] = op(B)[k0+r][n0+tx]
      Bs[r][tx] = (kb < K && n < N) ? (tb ? B[(long long)n * K + kb] : B[(long long)kb * N + n]) : 0.f;
    }
    __syncthreads();
#pragma unroll 8
    for (int k = 0; k < 32; ++k) {
      const float b = Bs[k][tx];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = fmaf(As[ty + 8 * j][k], b, acc[j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = m0 + ty + 8 * j, n = n0 + tx;
    if (m < M && n < N) C[(long long)m * N + n] = acc[j];
  }
}
}  // namespace

extern "C" int dlip_gemm_small_f32(const float* A, const float* B, float* C, int32_t M, int32_t N, int32_t K,
                                   int32_t trans_a, int32_t trans_b, dlip_stream_t stream) {
  DLIP_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0);
  hipLaunchKernelGGL(gemm_small_kernel, dim3((N + 31) / 32, (M + 31) / 32), dim3(256), 0,
                     static_cast<hipStream_t>(stream), A, B, C, M, N, K, trans_a, trans_b);
  return dlip_launch_status();
}

extern "C" int dlip_aam_margin_f32(const float* logits, const int64_t* labels, const float* g, float* y, int32_t B, int32_t K,
                                   float margin, int32_t easy_margin, int32_t backward, dlip_stream_t stream) {
  DLIP_CHECK_ARG(logits && labels && y && B > 0 && K > 0 && (!backward || g));
  const float cm = cosf(margin), sm = sinf(margin);
  const float th = cosf(3.14159265358979323846f - margin), mm = sinf(3.14159265358979323846f - margin) * margin;
  long long blocks = ((long long)B * K + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(aam_margin_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), logits,
                     reinterpret_cast<const long long*>(labels), g, y, B, K, cm, sm, th, mm, easy_margin, backward);
  return dlip_launch_status();
}

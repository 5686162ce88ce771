// Embedding-level tail of the hot path: z-norm + concat (test-time A+V fusion), L2 normalise,
// trial-pair cosine scoring over an in-HBM embedding table, cosine / linear logits with
// first-max argmax, margin softmax cross-entropy value, LowFER concat.
// All rows are short (<= a few thousand floats): one 64-lane wave owns a row, lanes stride the
// row with coalesced loads, reductions are wavefront shuffles (no LDS, no atomics) in fp64.
#include "dlip_common.h"

namespace {

constexpr int WAVES_PER_BLOCK = 4;

__device__ __forceinline__ void znorm_row(const float* __restrict__ src, float* __restrict__ dst, int D,
                                          int lane, bool biased) {
  double s = 0.0;
  for (int i = lane; i < D; i += 64) s += (double)src[i];
  const double mean = dlip_wave_sum_f64(s) / (double)D;
  double q = 0.0;
  for (int i = lane; i < D; i += 64) {
    const double d = (double)src[i] - mean;
    q += d * d;
  }
  q = dlip_wave_sum_f64(q);
  // The reference computes (x - mean)/std in fp32 on fp32 mean/std (train_fusion.py:234-237).
  const float mu = (float)mean;
  const float sd = (float)sqrt(q / (double)(biased ? D : D - 1));
  for (int i = lane; i < D; i += 64) dst[i] = (src[i] - mu) / sd;
}

__global__ __launch_bounds__(256) void znorm_cat_kernel(const float* __restrict__ a, int Da,
                                                        const float* __restrict__ v, int Dv,
                                                        float* __restrict__ y, int U, int biased) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= U) return;
  float* out = y + (long long)row * (Da + Dv);
  if (Da > 0) znorm_row(a + (long long)row * Da, out, Da, lane, biased != 0);
  if (Dv > 0) znorm_row(v + (long long)row * Dv, out + Da, Dv, lane, biased != 0);
}

// znorm_cat whose second table is still the pooled partial sums of the trunk's last convolution (dlip_conv_pool_f16x3):
// the wave first finishes its clip's mean exactly as dlip_pool_finish_f32 mode 0 does (tiles in row order, fp64,
// rounded to fp32 once), into its own slice of the output row, then z-normalises that slice in place.
__global__ __launch_bounds__(256) void znorm_cat_pooled_kernel(const float* __restrict__ a, int Da,
                                                               const double* __restrict__ part, long long M, int K, int Kp,
                                                               int BM, int Gs, const DlipLen len, float* __restrict__ y, int U,
                                                               int biased) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= U) return;
  float* out = y + (long long)row * (Da + K);
  if (Da > 0) znorm_row(a + (long long)row * Da, out, Da, lane, biased != 0);
  const long long r0 = (long long)row * Gs;
  long long r1 = r0 + Gs;
  if (r1 > M) r1 = M;
  const double n = (double)dlip_valid_rows(len, row, (int)(r1 - r0));   // ragged batches: the clip's valid rows (pool_finish's count)
  for (int i = lane; i < K; i += 64) {
    double s = 0.0;
    for (long long tm = r0 / BM; tm <= (r1 - 1) / BM; ++tm) {
      const int seg = (tm * BM) / Gs == row ? 0 : 1;
      s += part[((size_t)tm * 4 + 2 * seg) * Kp + i];
    }
    out[Da + i] = (float)(s / n);
  }
  znorm_row(out + Da, out + Da, K, lane, biased != 0);   // a lane reads back only what it wrote itself
}

__global__ __launch_bounds__(256) void l2norm_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                     int U, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= U) return;
  const float* p = x + (long long)row * D;
  double s = 0.0;
  for (int i = lane; i < D; i += 64) s += (double)p[i] * (double)p[i];
  const float nrm = fmaxf((float)sqrt(dlip_wave_sum_f64(s)), eps);
  for (int i = lane; i < D; i += 64) y[(long long)row * D + i] = p[i] / nrm;
}

__global__ __launch_bounds__(256) void pair_cosine_kernel(const float* __restrict__ emb, int N, int D,
                                                          const int32_t* __restrict__ ia,
                                                          const int32_t* __restrict__ ib,
                                                          float* __restrict__ score, int n_trials, int mode,
                                                          float eps, float weight, int accumulate) {
  const int lane = threadIdx.x & 63;
  const int tr = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  if (tr >= n_trials) return;
  const int ra = ia[tr], rb = ib[tr];
  float c;
  if ((unsigned)ra >= (unsigned)N || (unsigned)rb >= (unsigned)N) {
    c = __builtin_nanf("");
  } else {
    const float* pa = emb + (long long)ra * D;
    const float* pb = emb + (long long)rb * D;
    double saa = 0.0, sbb = 0.0;
    for (int i = lane; i < D; i += 64) {
      const double x = pa[i], z = pb[i];
      saa += x * x;
      sbb += z * z;
    }
    saa = dlip_wave_sum_f64(saa);
    sbb = dlip_wave_sum_f64(sbb);
    if (mode == 0) {
      // sklearn: normalise rows in fp32, then dot
      const float na = (float)sqrt(saa), nb = (float)sqrt(sbb);
      double sab = 0.0;
      for (int i = lane; i < D; i += 64) sab += (double)(pa[i] / na) * (double)(pb[i] / nb);
      c = (float)dlip_wave_sum_f64(sab);
    } else {
      // F.cosine_similarity: dot / max(|a|*|b|, eps)   (ATen clamps the product of the norms)
      double sab = 0.0;
      for (int i = lane; i < D; i += 64) sab += (double)pa[i] * (double)pb[i];
      sab = dlip_wave_sum_f64(sab);
      const double den = fmax(sqrt(saa * sbb), (double)eps);
      c = (float)(sab / den);
    }
  }
  if (lane == 0) score[tr] = accumulate ? score[tr] + weight * c : weight * c;
}

// Two-covariance PLDA same/different log-likelihood ratio of a trial in the model's latent space (u = the
// embedding after the model's affine map, unit within-class covariance, between-class variances psi):
//   llr = sum_d [ log(1 + psi) - log(1 + 2 psi) / 2 + psi (u1 + u2)^2 / (2 (1 + 2 psi)) - psi (u1^2 + u2^2) / (2 (1 + psi)) ]
// One wave per trial, fp64 accumulation, wavefront-shuffle reduction.
__global__ __launch_bounds__(256) void plda_llr_kernel(const float* __restrict__ u, int N, int D,
                                                       const float* __restrict__ psi, const int32_t* __restrict__ ia,
                                                       const int32_t* __restrict__ ib, float* __restrict__ score,
                                                       int n_trials) {
  const int lane = threadIdx.x & 63;
  const int tr = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  if (tr >= n_trials) return;
  const int ra = ia[tr], rb = ib[tr];
  double acc = 0.0;
  const bool ok = (unsigned)ra < (unsigned)N && (unsigned)rb < (unsigned)N;
  if (ok) {
    const float* pa = u + (long long)ra * D;
    const float* pb = u + (long long)rb * D;
    for (int i = lane; i < D; i += 64) {
      const double p = psi[i], a = pa[i], b = pb[i];
      acc += log1p(p) - 0.5 * log1p(2.0 * p) + p * (a + b) * (a + b) / (2.0 * (1.0 + 2.0 * p)) -
             p * (a * a + b * b) / (2.0 * (1.0 + p));
    }
  }
  acc = dlip_wave_sum_f64(acc);
  if (lane == 0) score[tr] = ok ? (float)acc : __builtin_nanf("");
}

// One workgroup per embedding row.  wnorm/en: 1/max(norm, 1e-12) as F.normalize (eps 1e-12).
__global__ __launch_bounds__(256) void logits_argmax_kernel(const float* __restrict__ e, const float* __restrict__ W,
                                                            const float* __restrict__ bias,
                                                            float* __restrict__ logits, long long* __restrict__ amax,
                                                            int D, int K, int cosine) {
  __shared__ float row_logits[1024];
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* pe = e + (long long)b * D;
  float en = 1.f;
  if (cosine) {
    double s = 0.0;
    for (int i = lane; i < D; i += 64) s += (double)pe[i] * (double)pe[i];
    en = fmaxf((float)sqrt(dlip_wave_sum_f64(s)), 1e-12f);
  }
  for (int k = wave; k < K; k += WAVES_PER_BLOCK) {
    const float* pw = W + (long long)k * D;
    double dot = 0.0, sw = 0.0;
    if (cosine) {
      for (int i = lane; i < D; i += 64) sw += (double)pw[i] * (double)pw[i];
      const float wn = fmaxf((float)sqrt(dlip_wave_sum_f64(sw)), 1e-12f);
      for (int i = lane; i < D; i += 64) dot += (double)(pe[i] / en) * (double)(pw[i] / wn);
    } else {
      for (int i = lane; i < D; i += 64) dot += (double)pe[i] * (double)pw[i];
    }
    dot = dlip_wave_sum_f64(dot);
    float v = (float)dot;
    if (!cosine && bias) v += bias[k];
    if (lane == 0) row_logits[k] = v;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += 256) logits[(long long)b * K + k] = row_logits[k];
  if (threadIdx.x == 0 && amax) {
    int best = 0;
    float bv = row_logits[0];
    for (int k = 1; k < K; ++k) {
      const float v = row_logits[k];
      if (v > bv || (v != v && bv == bv)) {  // strict '>' keeps the first maximum; NaN wins as in torch.max
        bv = v;
        best = k;
      }
    }
    amax[b] = best;
  }
}

__global__ __launch_bounds__(256) void margin_ce_kernel(const float* __restrict__ logits,
                                                        const long long* __restrict__ labels,
                                                        float* __restrict__ loss, int B, int K, float scale,
                                                        float margin) {
  __shared__ double part[256];
  double acc = 0.0;
  for (int b = threadIdx.x; b < B; b += 256) {
    const float* p = logits + (long long)b * K;
    const int lab = (int)labels[b];
    float mx = -__builtin_inff();
    for (int k = 0; k < K; ++k) {
      const float z = scale * (p[k] - (k == lab ? margin : 0.f)) + 1e-8f;
      mx = fmaxf(mx, z);
    }
    double se = 0.0;
    float zl = 0.f;
    for (int k = 0; k < K; ++k) {
      const float z = scale * (p[k] - (k == lab ? margin : 0.f)) + 1e-8f;
      se += exp((double)(z - mx));
      if (k == lab) zl = z;
    }
    acc += (double)mx + log(se) - (double)zl;
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = (float)(part[0] / (double)B);
}

__global__ __launch_bounds__(256) void lowfer_cat_kernel(const float* __restrict__ e1, const float* __restrict__ e2,
                                                         float* __restrict__ y, int B, int D) {
  const long long total = (long long)B * D;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long b = i / D;
    const int d = (int)(i - b * D);
    const float a = e1[i];
    const float s = 1.f / (1.f + expf(-e2[i]));
    float* o = y + b * 3 * D;
    o[d] = a;
    o[D + d] = s;
    o[2 * D + d] = s * a;
  }
}

}  // namespace

extern "C" int dlip_znorm_cat_f32(const float* a, int32_t Da, const float* v, int32_t Dv, float* y, int32_t U,
                                  int32_t biased, dlip_stream_t stream) {
  DLIP_CHECK_ARG(y && U > 0 && Da >= 0 && Dv >= 0 && (Da + Dv) > 0);
  DLIP_CHECK_ARG((Da == 0 || a) && (Dv == 0 || v));
  hipLaunchKernelGGL(znorm_cat_kernel, dim3((U + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK), dim3(256), 0,
                     static_cast<hipStream_t>(stream), a, Da, v, Dv, y, U, biased);
  return dlip_launch_status();
}

extern "C" int dlip_znorm_cat_pooled_f32(const float* a, int32_t Da, const double* partials, int64_t M, int32_t K,
                                         int32_t tile_rows, int32_t group_rows, const int32_t* group_len, int32_t len_mul,
                                         int32_t len_add, float* y, int32_t U, int32_t biased, dlip_stream_t stream) {
  DLIP_CHECK_ARG(y && partials && U > 0 && Da >= 0 && K > 0 && M > 0 && tile_rows > 0 && group_rows >= tile_rows);
  DLIP_CHECK_ARG((Da == 0 || a) && (M + group_rows - 1) / group_rows == U);
  DlipLen l; l.len = group_len; l.mul = len_mul; l.add = len_add;
  hipLaunchKernelGGL(znorm_cat_pooled_kernel, dim3((U + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK), dim3(256), 0,
                     static_cast<hipStream_t>(stream), a, Da, partials, (long long)M, K, (K + 127) / 128 * 128, tile_rows,
                     group_rows, l, y, U, biased);
  return dlip_launch_status();
}

extern "C" int dlip_l2_normalize_f32(const float* x, float* y, int32_t U, int32_t D, float eps,
                                     dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && y && U > 0 && D > 0);
  hipLaunchKernelGGL(l2norm_kernel, dim3((U + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, y, U, D, eps);
  return dlip_launch_status();
}

extern "C" int dlip_pair_cosine_f32(const float* emb, int32_t N, int32_t D, const int32_t* idx_a,
                                    const int32_t* idx_b, float* score, int32_t n_trials, int32_t mode, float eps,
                                    float weight, int32_t accumulate, dlip_stream_t stream) {
  DLIP_CHECK_ARG(emb && idx_a && idx_b && score && N > 0 && D > 0 && n_trials > 0 && (mode == 0 || mode == 1));
  hipLaunchKernelGGL(pair_cosine_kernel, dim3((n_trials + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK), dim3(256), 0,
                     static_cast<hipStream_t>(stream), emb, N, D, idx_a, idx_b, score, n_trials, mode, eps, weight,
                     accumulate);
  return dlip_launch_status();
}

extern "C" int dlip_plda_llr_f32(const float* u, int32_t N, int32_t D, const float* psi, const int32_t* idx_a,
                                const int32_t* idx_b, float* score, int32_t n_trials, dlip_stream_t stream) {
  DLIP_CHECK_ARG(u && psi && idx_a && idx_b && score && N > 0 && D > 0 && n_trials > 0);
  hipLaunchKernelGGL(plda_llr_kernel, dim3((n_trials + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK), dim3(256), 0,
                     static_cast<hipStream_t>(stream), u, N, D, psi, idx_a, idx_b, score, n_trials);
  return dlip_launch_status();
}

extern "C" int dlip_logits_argmax_f32(const float* e, const float* W, const float* bias, float* logits,
                                      int64_t* argmax, int32_t B, int32_t D, int32_t K, int32_t cosine,
                                      dlip_stream_t stream) {
  DLIP_CHECK_ARG(e && W && logits && B > 0 && D > 0 && K > 0 && K <= 1024);
  hipLaunchKernelGGL(logits_argmax_kernel, dim3(B), dim3(256), 0, static_cast<hipStream_t>(stream), e, W, bias,
                     logits, reinterpret_cast<long long*>(argmax), D, K, cosine);
  return dlip_launch_status();
}

extern "C" int dlip_margin_ce_loss_f32(const float* logits, const int64_t* labels, float* loss, int32_t B,
                                       int32_t K, float scale, float margin, dlip_stream_t stream) {
  DLIP_CHECK_ARG(logits && labels && loss && B > 0 && K > 0);
  hipLaunchKernelGGL(margin_ce_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), logits,
                     reinterpret_cast<const long long*>(labels), loss, B, K, scale, margin);
  return dlip_launch_status();
}

extern "C" int dlip_lowfer_cat_f32(const float* e1, const float* e2, float* y, int32_t B, int32_t D,
                                   dlip_stream_t stream) {
  DLIP_CHECK_ARG(e1 && e2 && y && B > 0 && D > 0);
  long long g = ((long long)B * D + 255) / 256;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(lowfer_cat_kernel, dim3((unsigned)g), dim3(256), 0, static_cast<hipStream_t>(stream), e1, e2,
                     y, B, D);
  return dlip_launch_status();
}

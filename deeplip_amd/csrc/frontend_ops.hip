// GPU-side preprocessing at the entrance of the hot path.
//
// Audio: the reference extracts MFCC / (log-)fbank features per utterance on CPU workers with the
// third-party python_speech_features package (models/audio_models/datasets.py:65-83: mfcc(winlen 0.025,
// winstep 0.01, numcep 24) with that package's defaults nfilt 26, nfft 512, preemph 0.97,
// ceplifter 22, appendEnergy, rectangular window; then per-utterance mean/variance normalisation,
// datasets.py:52-53).  Here the chain is: framing + pre-emphasis (this file) -> 512-point real DFT as
// ONE fp32 MFMA GEMM against a [514 x 512] cos/sin matrix -> power spectrum + frame energy (this
// file) -> mel filterbank GEMM -> log -> DCT-II(+lifter) GEMM -> c0 := log energy -> CMVN (this file).
// The GEMMs reuse dlip_conv_nhwc_f32 (host side: deeplip_amd/frontend.py).
//
// Video: uint8 gray or RGB frames -> centre crop -> /255 -> (x - 0.421)/0.165
// (models/video_models/dataloaders.py:11-22: Normalize(0,255), CenterCrop(88), Normalize(0.421,0.165)).
#include "dlip_common.h"

namespace {

// frames[b*NF + f][j] = (j < L ? pre(b, f*step + j) : 0), pre(n) = x[n] - coef*x[n-1] (x[-1] -> x[0] alone,
// samples beyond the signal are zero padding as python_speech_features.sigproc.framesig does).
__global__ __launch_bounds__(256) void frame_preemph_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                            int S, int NF, int L, int step, int nfft, float coef,
                                                            long long total) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int j = (int)(i % nfft);
    const long long r = i / nfft;
    const int f = (int)(r % NF);
    const long long b = r / NF;
    float v = 0.f;
    const int n = f * step + j;
    if (j < L && n < S) {
      const float* xb = x + b * S;
      v = n == 0 ? xb[0] : xb[n] - coef * xb[n - 1];
    }
    y[i] = v;
  }
}

// spec [R, 2*NB] = (re_0..re_{NB-1}, im_0..im_{NB-1}) -> pow [R, NBp] = (re^2 + im^2)/nfft (zero padded to NBp),
// energy[r] = sum_k pow (0 -> eps), as python_speech_features.sigproc.powspec / base.fbank.
__global__ __launch_bounds__(256) void powspec_kernel(const float* __restrict__ spec, float* __restrict__ pw,
                                                      float* __restrict__ energy, int R, int NB, int NBp, float inv_nfft) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const float* s = spec + (long long)r * 2 * NB;
  double e = 0.0;
  for (int k = lane; k < NBp; k += 64) {
    float p = 0.f;
    if (k < NB) {
      p = (s[k] * s[k] + s[NB + k] * s[NB + k]) * inv_nfft;
      e += (double)p;
    }
    pw[(long long)r * NBp + k] = p;
  }
  e = dlip_wave_sum_f64(e);
  if (lane == 0) energy[r] = e == 0.0 ? 2.220446049250313e-16f : (float)e;
}

// Power spectrum with the DFT carried in fp64 (numpy's rfft, which python_speech_features calls, runs in double): one
// workgroup per frame, the frame and the nfft twiddles (cos, sin of 2 pi j / nfft) in LDS, one thread per bin.  For banks
// whose lowest filters sit on bins holding ~1e-9 of the spectrum (80 bands at nfft = 512) the fp32-GEMM DFT's rounding
// noise is the signal; this path holds 1e-4 there too (nfft a power of two, <= 1024).
__global__ __launch_bounds__(256) void powspec_dft64_kernel(const float* __restrict__ frames, float* __restrict__ pw,
                                                            float* __restrict__ energy, int NB, int NBp, int nfft) {
  extern __shared__ double sh[];          // [nfft] frame | [nfft] cos | [nfft] sin | [4] wave sums
  double* fr = sh;
  double* cs = sh + nfft;
  double* sn = sh + 2 * nfft;
  double* part = sh + 3 * nfft;
  const int r = blockIdx.x;
  for (int j = threadIdx.x; j < nfft; j += 256) {
    fr[j] = (double)frames[(long long)r * nfft + j];
    double s, c;
    sincospi(2.0 * (double)j / (double)nfft, &s, &c);
    cs[j] = c;
    sn[j] = s;
  }
  __syncthreads();
  double e = 0.0;
  for (int k = threadIdx.x; k < NBp; k += 256) {
    float p = 0.f;
    if (k < NB) {
      double re = 0.0, im = 0.0;
      for (int n = 0; n < nfft; ++n) {
        const int idx = (k * n) & (nfft - 1);
        re += fr[n] * cs[idx];
        im -= fr[n] * sn[idx];
      }
      const double pp = (re * re + im * im) / (double)nfft;
      p = (float)pp;
      e += pp;
    }
    pw[(long long)r * NBp + k] = p;
  }
  e = dlip_wave_sum_f64(e);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = e;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double t = part[0] + part[1] + part[2] + part[3];
    energy[r] = t == 0.0 ? 2.220446049250313e-16f : (float)t;
  }
}

// Power spectrum STRAIGHT FROM THE WAVEFORM, in the reference's own precision: python_speech_features pre-emphasises, frames and
// transforms in fp64 (numpy), so a band that holds 1e-12 of a frame's energy -- the lowest one-bin mel filter next to DC in a frame where
// pre-emphasis leaves nothing -- is still resolved there, while fp32 pre-emphasis alone puts rounding noise at -143 dB of the frame under
// it (tools/probes/frontend_fuzz.py: such elements came out 0.3 .. 1 % off in energy).  One workgroup per frame: the frame's samples are
// pre-emphasised in fp64 (coefficient a double: 0.97, not 0.97f) into LDS in bit-reversed order, a radix-2 decimation-in-time FFT of
// nfft points runs there (log2(nfft) stages of nfft / 2 butterflies, one per thread and stage; twiddles from an LDS table of
// sincospi), and the first NB bins leave as |.|^2 / nfft in fp32 together with the frame's energy (summed in fp64).
__global__ __launch_bounds__(512) void powspec_wave_fft64_kernel(const float* __restrict__ x, float* __restrict__ pw, float* __restrict__ energy,
                                                                 long long S, int NF, int L, int step, int nfft, int lg, double coef, int NB,
                                                                 int NBp) {
  extern __shared__ double sh[];          // [nfft] re | [nfft] im | [nfft / 2] cos | [nfft / 2] sin | [8] wave sums
  double* re = sh;
  double* im = sh + nfft;
  double* cs = sh + 2 * nfft;
  double* sn = cs + nfft / 2;
  double* part = sn + nfft / 2;
  const long long r = blockIdx.x;
  const int f = (int)(r % NF);
  const float* xb = x + (r / NF) * S;
  const int half = nfft >> 1;
  for (int j = threadIdx.x; j < nfft; j += blockDim.x) {
    const long long n = (long long)f * step + j;
    double v = 0.0;
    if (j < L && n < S) v = n == 0 ? (double)xb[0] : (double)xb[n] - coef * (double)xb[n - 1];
    const int br = (int)(__brev((unsigned)j) >> (32 - lg));
    re[br] = v;
    im[br] = 0.0;
    if (j < half) {
      double s, c;
      sincospi(-2.0 * (double)j / (double)nfft, &s, &c);       // exp(-2 pi i j / nfft)
      cs[j] = c;
      sn[j] = s;
    }
  }
  __syncthreads();
  for (int st = 1; st <= lg; ++st) {
    const int len = 1 << st, hl = len >> 1, tw = nfft >> st;
    for (int t = threadIdx.x; t < half; t += blockDim.x) {
      const int pos = t & (hl - 1), i0 = ((t >> (st - 1)) << st) + pos, i1 = i0 + hl;
      const double wr = cs[pos * tw], wi = sn[pos * tw];
      const double br_ = re[i1] * wr - im[i1] * wi, bi_ = re[i1] * wi + im[i1] * wr;
      const double ar = re[i0], ai = im[i0];
      re[i0] = ar + br_; im[i0] = ai + bi_;
      re[i1] = ar - br_; im[i1] = ai - bi_;
    }
    __syncthreads();
  }
  double e = 0.0;
  for (int k = threadIdx.x; k < NBp; k += blockDim.x) {
    float p = 0.f;
    if (k < NB) {
      const double pp = (re[k] * re[k] + im[k] * im[k]) / (double)nfft;
      p = (float)pp;
      e += pp;
    }
    pw[r * NBp + k] = p;
  }
  e = dlip_wave_sum_f64(e);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = e;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < (int)(blockDim.x + 63) / 64; ++w) t += part[w];
    energy[r] = t == 0.0 ? 2.220446049250313e-16f : (float)t;
  }
}

// y = log(x == 0 ? eps : x) elementwise (base.logfbank / base.mfcc)
__global__ __launch_bounds__(256) void log_floor_kernel(const float* __restrict__ x, float* __restrict__ y, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float v = x[i];
    y[i] = logf(v == 0.f ? 2.220446049250313e-16f : v);
  }
}

// feat [B, NF, C] (c0 optionally replaced by log(energy)) -> CMVN over the NF frames of each utterance:
// (x - mean)/(std_biased + 2e-12) (datasets.py:52-53), written channel-first [B, C, NF] like the
// reference loaders (datasets.py:135).  One thread per (b, c), fp64 statistics.
__global__ __launch_bounds__(256) void cmvn_kernel(const float* __restrict__ feat, const float* __restrict__ energy,
                                                   float* __restrict__ y, int B, int NF, int C, int ldf, int normalize) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)B * C) return;
  const int c = (int)(i % C);
  const long long b = i / C;
  const float* p = feat + b * NF * ldf + c;
  const float* en = (energy && c == 0) ? energy + b * NF : nullptr;
  double s = 0.0;
  for (int f = 0; f < NF; ++f) s += (double)(en ? logf(en[f]) : p[(long long)f * ldf]);
  const double mean = s / NF;
  double q = 0.0;
  for (int f = 0; f < NF; ++f) {
    const double d = (double)(en ? logf(en[f]) : p[(long long)f * ldf]) - mean;
    q += d * d;
  }
  const float mu = normalize ? (float)mean : 0.f;
  const float den = normalize ? (float)sqrt(q / NF) + 2e-12f : 1.f;
  float* o = y + (b * C + c) * NF;
  for (int f = 0; f < NF; ++f) o[f] = ((en ? logf(en[f]) : p[(long long)f * ldf]) - mu) / den;
}

// python_speech_features.delta as SpkTrainDataset._delta applies it (datasets.py:55-63): x [B, C, NF] (channel first, as
// the loaders hold features) -> y [B, (1 + order) C, NF] = [x | delta(x, N=1) | delta(x, N=2)] -- BOTH differences are
// taken of the base features (the reference does not difference the deltas again):
//   delta(x, N)[t] = sum_{n=-N..N} n x[clamp(t + n)] / (2 sum_{n=1..N} n^2)        (edge padding)
__global__ __launch_bounds__(256) void delta_kernel(const float* __restrict__ x, float* __restrict__ y, long long rows, int C,
                                                    int NF, int order) {
  const long long total = rows * NF;                 // rows = B * C
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int t = (int)(i % NF);
    const long long r = i / NF;
    const long long b = r / C;
    const int c = (int)(r - b * C);
    const float* p = x + r * NF;
    auto at = [&](int u) { return p[u < 0 ? 0 : (u >= NF ? NF - 1 : u)]; };
    float* o = y + (b * (1 + order) * C + c) * NF + t;
    o[0] = p[t];
    if (order >= 1) o[(long long)C * NF] = (at(t + 1) - at(t - 1)) / 2.0f;
    if (order >= 2) o[(long long)2 * C * NF] = (2.0f * at(t + 2) + at(t + 1) - at(t - 1) - 2.0f * at(t - 2)) / 10.0f;
  }
}

// uint8 frames [N, CH, H, W] (CH = 1 gray or 3 RGB; N = B clips of T frames) -> crop [N, CS, CS] float, normalised.
//   clip_params == NULL: the "val" pipeline's CenterCrop (preprocess.py:89-90: the margin halved and rounded DOWN);
//   clip_params [B][4] = (oy, ox, flip, 0): the "train" pipeline's RandomCrop origin + HorizontalFlip coin of each clip
//     (preprocess.py:95-138, dataloaders.py:13-17), drawn by the host; a flipped clip reads column ox + CS - 1 - cx;
//   lengths [B]: frames t >= lengths[b] are the zero padding of pad_packed_collate -- zeros of the NORMALISED clip
//     (dataset.py:117,130-134: the clip is normalised first, padded second).
__global__ __launch_bounds__(256) void crop_norm_kernel(const uint8_t* __restrict__ x, float* __restrict__ y,
                                                        long long N, int CH, int H, int W, int CS, int T,
                                                        const int32_t* __restrict__ clip_params, const int32_t* __restrict__ lengths) {
  const long long total = N * CS * CS;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    int cx = (int)(i % CS);
    const int cy = (int)((i / CS) % CS);
    const long long n = i / ((long long)CS * CS);
    const int b = (int)(n / T);
    if (lengths != nullptr && (int)(n - (long long)b * T) >= lengths[b]) { y[i] = 0.f; continue; }
    int oy = (H - CS) / 2, ox = (W - CS) / 2;
    if (clip_params != nullptr) {
      oy = min(max(clip_params[4 * b], 0), H - CS);
      ox = min(max(clip_params[4 * b + 1], 0), W - CS);
      if (clip_params[4 * b + 2] != 0) cx = CS - 1 - cx;
    }
    const uint8_t* p = x + (n * CH * H + (oy + cy)) * W + ox + cx;
    float g;
    if (CH == 3) g = dlip_gray601((float)p[0], (float)p[(long long)H * W], (float)p[2LL * H * W]);
    else g = (float)p[0];
    y[i] = dlip_pixel_norm(g);
  }
}

inline unsigned grid_for(long long total) {
  long long g = (total + 255) / 256;
  if (g > 2048) g = 2048;
  return (unsigned)(g < 1 ? 1 : g);
}

}  // namespace

extern "C" int dlip_frame_preemph_f32(const float* x, float* frames, int32_t B, int32_t S, int32_t NF, int32_t frame_len,
                                      int32_t frame_step, int32_t nfft, float preemph, dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && frames && B > 0 && S > 0 && NF > 0 && frame_len > 0 && frame_step > 0 && nfft >= frame_len);
  const long long total = (long long)B * NF * nfft;
  hipLaunchKernelGGL(frame_preemph_kernel, dim3(grid_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                     frames, S, NF, frame_len, frame_step, nfft, preemph, total);
  return dlip_launch_status();
}

extern "C" int dlip_powspec_f32(const float* spec, float* pw, float* energy, int32_t R, int32_t NB, int32_t NBp,
                                int32_t nfft, dlip_stream_t stream) {
  DLIP_CHECK_ARG(spec && pw && energy && R > 0 && NB > 0 && NBp >= NB && nfft > 0);
  hipLaunchKernelGGL(powspec_kernel, dim3((R + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), spec, pw, energy,
                     R, NB, NBp, 1.0f / (float)nfft);
  return dlip_launch_status();
}

extern "C" int dlip_powspec_dft64_f32(const float* frames, float* pw, float* energy, int32_t R, int32_t NB, int32_t NBp,
                                      int32_t nfft, dlip_stream_t stream) {
  DLIP_CHECK_ARG(frames && pw && energy && R > 0 && NB > 0 && NBp >= NB && nfft >= 2 && nfft <= 1024 && (nfft & (nfft - 1)) == 0);
  DLIP_CHECK_ARG(NB <= nfft / 2 + 1);
  hipLaunchKernelGGL(powspec_dft64_kernel, dim3(R), dim3(256), (size_t)(3 * nfft + 4) * sizeof(double),
                     static_cast<hipStream_t>(stream), frames, pw, energy, NB, NBp, nfft);
  return dlip_launch_status();
}

extern "C" int dlip_powspec_wave_fft64_f32(const float* x, float* pw, float* energy, int64_t B, int64_t S, int32_t NF, int32_t frame_len,
                                           int32_t frame_step, int32_t nfft, double preemph, int32_t NB, int32_t NBp, dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && pw && energy && B > 0 && S > 0 && NF > 0 && frame_len > 0 && frame_step > 0 && nfft >= frame_len);
  DLIP_CHECK_ARG(nfft >= 128 && nfft <= 1024 && (nfft & (nfft - 1)) == 0 && NB > 0 && NB <= nfft / 2 + 1 && NBp >= NB);
  DLIP_CHECK_ARG(B * (long long)NF < (1ll << 31));
  int lg = 0;
  while ((1 << lg) < nfft) ++lg;
  const int threads = nfft / 2;            // 64 .. 512: one butterfly per thread and stage
  const size_t lds = (size_t)(3 * nfft + 8) * sizeof(double);
  hipLaunchKernelGGL(powspec_wave_fft64_kernel, dim3((unsigned)(B * NF)), dim3((unsigned)threads), lds, static_cast<hipStream_t>(stream), x, pw,
                     energy, (long long)S, NF, frame_len, frame_step, nfft, lg, preemph, NB, NBp);
  return dlip_launch_status();
}

extern "C" int dlip_log_floor_f32(const float* x, float* y, int64_t n, dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && y && n > 0);
  hipLaunchKernelGGL(log_floor_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, (long long)n);
  return dlip_launch_status();
}

extern "C" int dlip_cmvn_nct_f32(const float* feat, const float* energy, float* y, int32_t B, int32_t NF, int32_t C,
                                 int32_t ldf, int32_t normalize, dlip_stream_t stream) {
  DLIP_CHECK_ARG(feat && y && B > 0 && NF > 0 && C > 0 && ldf >= C);
  hipLaunchKernelGGL(cmvn_kernel, dim3((unsigned)(((long long)B * C + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), feat, energy, y, B, NF, C, ldf, normalize);
  return dlip_launch_status();
}

extern "C" int dlip_delta_nct_f32(const float* x, float* y, int32_t B, int32_t C, int32_t NF, int32_t order,
                                  dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && y && B > 0 && C > 0 && NF > 0 && (order == 1 || order == 2));
  const long long rows = (long long)B * C;
  hipLaunchKernelGGL(delta_kernel, dim3(grid_for(rows * NF)), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, rows, C, NF,
                     order);
  return dlip_launch_status();
}

extern "C" int dlip_crop_normalize_u8(const uint8_t* x, const int32_t* clip_params, const int32_t* lengths, int32_t T, float* y,
                                      int64_t n_frames, int32_t channels, int32_t H, int32_t W, int32_t crop, dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && y && n_frames > 0 && (channels == 1 || channels == 3) && crop > 0 && H >= crop && W >= crop);
  DLIP_CHECK_ARG(T > 0 && n_frames % T == 0);
  hipLaunchKernelGGL(crop_norm_kernel, dim3(grid_for(n_frames * crop * crop)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, y, (long long)n_frames, channels, H, W, crop, T, clip_params, lengths);
  return dlip_launch_status();
}

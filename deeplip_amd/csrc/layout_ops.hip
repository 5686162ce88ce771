// Layout adapters at the drop-in boundary: the reference's modules speak channel-first
// ([B,F,T] acoustic features, tdnn.py:89), the kernels speak channels-last.
#include "dlip_common.h"

namespace {

// x [B,R,Cc] -> y [B,Cc,Rp] (zero-padded columns R..Rp-1); 32x32 LDS tile, +1 pad.
__global__ __launch_bounds__(256) void transpose_pad_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                            int R, int Cc, int Rp) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const float* xb = x + (long long)b * R * Cc;
  float* yb = y + (long long)b * Cc * Rp;
#pragma unroll
  for (int j = 0; j < 32; j += 8) {
    const int r = r0 + ty + j, c = c0 + tx;
    tile[ty + j][tx] = (r < R && c < Cc) ? xb[(long long)r * Cc + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 32; j += 8) {
    const int c = c0 + ty + j, r = r0 + tx;
    if (c < Cc && r < Rp) yb[(long long)c * Rp + r] = tile[tx][ty + j];
  }
}

// x [B,Cc,T] -> y [B,T,Cp] in the split activation format (Cp % 32 == 0, channels Cc..Cp-1 zero): the speech
// encoder's input adapter and its split in one pass (one 32 x 32 tile per workgroup: 32 frames x one channel block).
__global__ __launch_bounds__(256) void transpose_split_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                              int Cc, int T, int Cp, DlipRange status) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int c0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const float* xb = x + (long long)b * Cc * T;
  float amax = 0.f;
#pragma unroll
  for (int j = 0; j < 32; j += 8) {
    const int c = c0 + ty + j, t = t0 + tx;
    tile[ty + j][tx] = (c < Cc && t < T) ? xb[(long long)c * T + t] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 32; j += 8) {
    const int t = t0 + ty + j;
    if (t < T) {
      const float v = tile[tx][ty + j];
      const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
      amax = fmaxf(amax, fabsf(v));
      _Float16* blk = reinterpret_cast<_Float16*>(y + ((long long)b * T + t) * Cp + c0);
      blk[tx] = hi;
      blk[32 + tx] = lo;
    }
  }
  dlip_report_range_block(amax, status);
}

__global__ __launch_bounds__(256) void ingest_rgb_kernel(const uint8_t* __restrict__ x, float* __restrict__ y,
                                                         long long n_frames, int HW) {
  const long long total = n_frames * HW;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long f = i / HW;
    const int p = (int)(i - f * HW);
    const uint8_t* px = x + f * 3 * HW + p;
    y[i] = dlip_pixel_norm(dlip_gray601((float)px[0], (float)px[HW], (float)px[2 * HW]));
  }
}

}  // namespace

extern "C" int dlip_nct_to_ntc_f32(const float* x, float* y, int32_t B, int32_t C, int32_t T, int32_t Cp,
                                   dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && y && B > 0 && C > 0 && T > 0 && Cp >= C && B <= 65535);
  // rows = C, cols = T  ->  y [B, T, Cp]
  dim3 grid((T + 31) / 32, (Cp + 31) / 32, B);
  hipLaunchKernelGGL(transpose_pad_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), x, y, C, T, Cp);
  return dlip_launch_status();
}

extern "C" int dlip_nct_to_ntc_split_f32(const float* x, float* y, int32_t B, int32_t C, int32_t T, int32_t Cp,
                                         dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && y && B > 0 && C > 0 && T > 0 && Cp >= C && (Cp & 31) == 0 && B <= 65535);
  dim3 grid((T + 31) / 32, Cp / 32, B);
  hipLaunchKernelGGL(transpose_split_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), x, y, C, T, Cp,
                     dlip_range_for(DLIP_ST_PACK));
  return dlip_launch_status();
}

extern "C" int dlip_ntc_to_nct_f32(const float* x, float* y, int32_t B, int32_t T, int32_t C,
                                   dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && y && B > 0 && C > 0 && T > 0 && B <= 65535);
  dim3 grid((C + 31) / 32, (T + 31) / 32, B);
  hipLaunchKernelGGL(transpose_pad_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), x, y, T, C, T);
  return dlip_launch_status();
}

extern "C" int dlip_ingest_rgb_u8(const uint8_t* x, float* y, int64_t n_frames, int32_t H, int32_t W,
                                  dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && y && n_frames > 0 && H > 0 && W > 0);
  long long g = (n_frames * H * W + 255) / 256;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(ingest_rgb_kernel, dim3((unsigned)g), dim3(256), 0, static_cast<hipStream_t>(stream), x, y,
                     (long long)n_frames, H * W);
  return dlip_launch_status();
}

namespace {
// y = act(x*scale + shift) (order 0: BN -> LeakyReLU) or act(x)*scale + shift (order 1), per channel.
__global__ __launch_bounds__(256) void affine_act_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, float* __restrict__ y,
                                                         long long total, int C, float slope, int order) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C);
    float v = x[i];
    if (order == 0) {
      v = v * scale[c] + shift[c];
      v = v >= 0.f ? v : v * slope;
    } else {
      v = v >= 0.f ? v : v * slope;
      v = v * scale[c] + shift[c];
    }
    y[i] = v;
  }
}
}  // namespace

extern "C" int dlip_affine_act_f32(const float* x, const float* scale, const float* shift, float* y, int64_t M,
                                   int32_t C, float slope, int32_t order, dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && scale && shift && y && M > 0 && C > 0 && (order == 0 || order == 1));
  long long g = (M * C + 255) / 256;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(affine_act_kernel, dim3((unsigned)g), dim3(256), 0, static_cast<hipStream_t>(stream), x, scale,
                     shift, y, (long long)M * C, C, slope, order);
  return dlip_launch_status();
}

namespace {
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
// fp32 [rows, C] <-> split format [rows, C/32 blocks, (32 hi halves | 32 lo halves)]  (C % 32 == 0).
// One thread per 4 channels: 16 B in, 8 B of hi + 8 B of lo out (or the reverse).
__global__ __launch_bounds__(256) void split_pack_kernel(const f32x4* __restrict__ x, float* __restrict__ y, long long n4,
                                                         DlipRange status) {
  float amax = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const f32x4 v = x[i];
    h4 hi, lo;
#pragma unroll
    for (int k = 0; k < 4; ++k) { hi[k] = (_Float16)v[k]; lo[k] = (_Float16)(v[k] - (float)hi[k]); amax = fmaxf(amax, fabsf(v[k])); }
    const long long blk = i >> 3; const int q = (int)(i & 7);          // 8 float4 per 32-channel block
    float* b = y + blk * 32;
    *reinterpret_cast<h4*>(b + q * 2) = hi;        // halves 4q..4q+3 of the hi half (64 B)
    *reinterpret_cast<h4*>(b + 16 + q * 2) = lo;   // same position in the lo half
  }
  dlip_report_range_block(amax, status);
}
__global__ __launch_bounds__(256) void split_unpack_kernel(const float* __restrict__ x, f32x4* __restrict__ y, long long n4) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const long long blk = i >> 3; const int q = (int)(i & 7);
    const float* b = x + blk * 32;
    const h4 hi = *reinterpret_cast<const h4*>(b + q * 2), lo = *reinterpret_cast<const h4*>(b + 16 + q * 2);
    f32x4 v;
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = (float)hi[k] + (float)lo[k];
    y[i] = v;
  }
}
}  // namespace

extern "C" int dlip_split_pack_f32(const float* x, float* y, int64_t rows, int32_t C, dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && y && rows > 0 && C > 0 && (C & 31) == 0);
  const long long n4 = rows * (C / 4);
  long long g = (n4 + 255) / 256; if (g > 2048) g = 2048;
  hipLaunchKernelGGL(split_pack_kernel, dim3((unsigned)g), dim3(256), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const f32x4*>(x), y, n4, dlip_range_for(DLIP_ST_PACK));
  return dlip_launch_status();
}

extern "C" int dlip_split_unpack_f32(const float* x, float* y, int64_t rows, int32_t C, dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && y && rows > 0 && C > 0 && (C & 31) == 0);
  const long long n4 = rows * (C / 4);
  long long g = (n4 + 255) / 256; if (g > 2048) g = 2048;
  hipLaunchKernelGGL(split_unpack_kernel, dim3((unsigned)g), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                     reinterpret_cast<f32x4*>(y), n4);
  return dlip_launch_status();
}

namespace {
// y[b, t, :] = t < len[b] ? x[b, t, :] : 0 -- the padding frames of a ragged batch made zeros of the normalised clip (what
// pad_packed_collate puts there, models/video_models/dataset.py:123-139) on the paths that have no pre-pass to do it in
// (exact-fp32 packing, taps).  E % 4 == 0; one float4 per thread.
__global__ __launch_bounds__(256) void mask_frames_kernel(const f32x4* __restrict__ x, f32x4* __restrict__ y, const int32_t* __restrict__ len,
                                                          int T, int E4, long long n4) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const long long f = i / E4;
    const int b = (int)(f / T), t = (int)(f - (long long)b * T);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (t < len[b]) v = x[i];
    y[i] = v;
  }
}
}  // namespace

extern "C" int dlip_mask_frames_f32(const float* x, const int32_t* len, float* y, int32_t B, int32_t T, int32_t E, dlip_stream_t stream) {
  DLIP_CHECK_ARG(x && len && y && B > 0 && T > 0 && E > 0 && (E & 3) == 0);
  DLIP_CHECK_ARG(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0);
  const long long n4 = (long long)B * T * (E / 4);
  long long g = (n4 + 255) / 256; if (g > 4096) g = 4096;
  hipLaunchKernelGGL(mask_frames_kernel, dim3((unsigned)g), dim3(256), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const f32x4*>(x), reinterpret_cast<f32x4*>(y), len, T, E / 4, n4);
  return dlip_launch_status();
}

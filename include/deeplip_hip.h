/*
 * deeplip_hip.h -- C ABI of libdeeplip_hip.so: the MI355X (gfx950 / CDNA4) kernels under the
 * DeepLip audio-visual embedding hot path.
 *
 * The reference (DanielMengLiu/DeepLip) has no FFI / operator / plugin layer: its boundary is the
 * Python module API + state-dict schema and all arithmetic is delegated to
 * stock torch.nn layers.  Each entry point below therefore names the reference call site(s) whose
 * torch.nn arithmetic it replaces (paths relative to the reference repo root).  The Python host
 * (the deeplip_amd package, re-exported as the models package) binds these with ctypes; INTEGRATION.md shows the
 * stub a reference maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller
 *     (kernels never allocate, free or synchronise);
 *   - activations are channels-last (NHWC / N-T-C), fp32; weights are pre-packed by the
 *     caller to [K][R][S][C] ("KRSC", BN already folded in fp64 on the host);
 *   - every call is asynchronous on `stream` (a hipStream_t; NULL = the null stream);
 *   - return value: 0 on success, a positive hipError_t value if the launch failed, or a
 *     negative DLIP_E* code for argument errors detected on the host.  No exceptions, no
 *     global mutable state: every function is re-entrant.
 */
#ifndef DEEPLIP_HIP_H
#define DEEPLIP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DLIP_ABI_VERSION 49
#define DLIP_LIFT_WORDS 4098   /* a gradient's power-of-two lift: (2^e, 2^-e), then 2^-e repeated 2048 times (the post_scale vector of
                                  the convolution that consumes the lifted gradient); while it is formed the words behind the pair
                                  hold one maximum per workgroup of the producing pass */

#define DLIP_OK 0
#define DLIP_EINVAL (-1)  /* inconsistent shapes / null pointers / unsupported alignment */
#define DLIP_ERANGE (-2)  /* a tensor exceeds the 2 GiB addressing window of one launch */

typedef void* dlip_stream_t; /* hipStream_t */
typedef void* dlip_plan_t;   /* a recorded step (dlip_plan_end) */

int dlip_abi_version(void);
/* sha256 (hex) over the sources and headers this library was built from, stamped at link time by deeplip_amd/build.py: the
 * build check compares it with the tree's own hash, so a stale prebuilt library is rebuilt and a fresh one is proven fresh. */
const char* dlip_source_sha(void);
/* Human-readable text for a code returned by any dlip_* call. */
const char* dlip_error_string(int code);

/* ------------------------------------------------------------------------------------------
 * Generic implicit-GEMM convolution, NHWC fp32, on v_mfma_f32_32x32x2_f32.
 *   y[n,ho,wo,k] = epilogue( sum_{r,s,c} x[n, ho*sh-ph+r*dh, wo*sw-pw+s*dw, c] * w[k,r,s,c] )
 *   epilogue(v)  = v + bias[k]; v += residual[m*ldr + k]; v = v>=0 ? v : v*slope[k];
 *                  v = v*post_scale[k] + post_shift[k]            (each step skipped when NULL)
 * Replaces, with BN folded into (w, bias):
 *   Conv2d 3x3 / 1x1 + BatchNorm2d + PReLU|ReLU + residual add   models/video_models/resnet.py:9-17,55-69
 *   Conv1d(dilated) + BatchNorm1d + LeakyReLU(0.2) (TDNN_Block)   models/audio_models/tdnn.py:23-43
 *   Conv1d + BatchNorm1d + Chomp1d + PReLU, 1x1 residual Conv1d   models/video_models/tcn.py:46-59,87,113-116
 *   nn.Linear (+BatchNorm1d +LeakyReLU)                           models/audio_models/tdnn.py:85-101,
 *                                                                 models/fusion_models/model_fusion.py:19-24,
 *                                                                 models/audio_models/loss.py:14,
 *                                                                 models/video_models/model.py:27
 * Conv1d is H=1; Linear is H=W=R=S=1.  C % 4 == 0 required (pad on pack); ldx/ldy/ldr are pixel
 * strides in floats (>= C / K), so a call may read or write a channel slice of a wider tensor.
 * ------------------------------------------------------------------------------------------ */
typedef struct dlip_conv_desc {
  int32_t N, H, W, C;          /* input  [N,H,W,C]                 */
  int32_t K;                   /* output channels                  */
  int32_t R, S;                /* filter taps (height, width)      */
  int32_t stride_h, stride_w;
  int32_t pad_h, pad_w;
  int32_t dil_h, dil_w;
  int32_t Ho, Wo;              /* output spatial size (validated)  */
  int32_t ldx, ldy, ldr;       /* pixel strides in floats          */
} dlip_conv_desc;

int dlip_conv_nhwc_f32(const dlip_conv_desc* d, const float* x, const float* w_krsc,
                       const float* bias, const float* residual, const float* slope,
                       const float* post_scale, const float* post_shift, float* y,
                       dlip_stream_t stream);

/* Split-precision variant of dlip_conv_nhwc_f32: same operation, fp32 in / fp32 out, but every
 * product is evaluated as hi*hi + hi*lo + lo*hi on the f16 matrix core (3 x v_mfma_f32_32x32x16_f16,
 * fp32 accumulate; ~2^-22 relative, see conv_igemm_f16x3.hip).  `w_split` = weights pre-split by the
 * host into (hi, lo) fp16: [K][R][S][C32/32][2][32] halves with C32 = C rounded up to 32 (zero
 * filled), each output channel k pre-multiplied by the power of two `w_scale[k]` (undone exactly in
 * the epilogue).  bias, slope, post_* are fp32.  With flags = 0, x / residual / y are plain fp32 as
 * in dlip_conv_nhwc_f32.  DLIP_SPLIT_IN: x and residual are in the *split activation format* -- per
 * pixel and 32-channel block, 32 hi halves then 32 lo halves (value = hi + lo), i.e. the same 128
 * bytes and the same ld* (counted in fp32 slots) as the fp32 block they replace; C, ldx, ldr (and K
 * when a residual is given) must be multiples of 32.  DLIP_SPLIT_OUT: y is written in that format
 * (K, ldy multiples of 32).  Chains of convolutions then split each activation once, in the
 * producer's epilogue, instead of once per filter tap in every consumer. */
#define DLIP_SPLIT_IN 1
#define DLIP_SPLIT_OUT 2
int dlip_conv_nhwc_f16x3(const dlip_conv_desc* d, const float* x, const void* w_split,
                         const float* w_scale, const float* bias, const float* residual,
                         const float* slope, const float* post_scale, const float* post_shift,
                         float* y, int32_t flags, dlip_stream_t stream);

/* dlip_conv_nhwc_f16x3 with a SECOND reduction source: y = epilogue( conv(x, w[:, taps]) + conv1x1_strided(x2, w[:, tail]) ).
 * Replaces conv2 + bn2 and the shortcut's 1x1 stride-2 convolution + BatchNorm of a down-sampling BasicBlock
 * (models/video_models/resnet.py:13-17 downsample_basic_block, :55-69 forward: `out += residual` with
 * residual = self.downsample(x)) in ONE launch: both BatchNorms are folded into the weights, so the sum of the two
 * convolutions is one reduction over [R*S*C32 taps | C2 shortcut channels] (what w_split holds per output channel,
 * under one power-of-two scale w_scale[k]); bias = the sum of the two folded biases.  x2 is [N,H2,W2,C2] in the
 * split activation format (DLIP_SPLIT_IN is mandatory), read at pixel (ho*stride2_h, wo*stride2_w) for output pixel
 * (ho, wo); C2, ldx2 multiples of 32.  residual may still be given (added after both).  Split-format kernel only. */
/* (ABI 43) dlip_conv_nhwc_f16x3 with DLIP_SPLIT_IN, fp32 y, no residual / post-affine, whose epilogue ALSO leaves the column sums of
 * the y it writes: stats [chunks][K][2] fp64 = per chunk of output rows {sum y, sum y^2} -- the batch statistics of the BatchNorm
 * behind a convolution under model.train() (models/audio_models/tdnn.py:35-43: Conv1d -> BatchNorm1d), without the pass over y that
 * dlip_bn_rows_train_fwd_f32 would otherwise make (hand it `stats` as its workspace and `chunks` as ready_chunks).
 * dlip_conv_stats_chunks(d): how many chunks such a launch of `d` writes, or 0 when this shape has no statistics epilogue (today:
 * the rows kernel's shapes -- 1-D valid convolutions and k = 1 GEMMs; the caller then launches the plain convolution).  Within a
 * chunk (half a tile: <= 80 rows) the sums are fp32 in a fixed order; across chunks the finalize kernel adds in fp64. */
int32_t dlip_conv_stats_chunks(const dlip_conv_desc* d);
int dlip_conv_nhwc_stats_f16x3(const dlip_conv_desc* d, const float* x, const void* w_split, const float* w_scale,
                               const float* bias, const float* slope, float* y, double* stats, int64_t stats_bytes,
                               dlip_stream_t stream);
int dlip_conv2_nhwc_f16x3(const dlip_conv_desc* d, const float* x, const float* x2, int32_t H2, int32_t W2,
                          int32_t C2, int32_t ldx2, int32_t stride2_h, int32_t stride2_w, const void* w_split,
                          const float* w_scale, const float* bias, const float* residual, const float* slope,
                          const float* post_scale, const float* post_shift, float* y, int32_t flags,
                          dlip_stream_t stream);

/* dlip_conv_nhwc_f16x3 (DLIP_SPLIT_IN) whose epilogue keeps only POOLED STATISTICS of its output: the rows
 * m = (n, ho, wo) are cut into consecutive groups of `group_rows` (a clip's T*Ho*Wo feature-map pixels; an
 * utterance's T' frames) and every workgroup tile stores, per group segment it contains, the fp64 column sums of
 * y and y^2 -- the [M,K] output itself never reaches memory.  dlip_pool_finish_f32 turns the partial sums into
 *   mode 0: the group means  [G,K]   (AdaptiveAvgPool2d(1) + the temporal mean of the lip-clip features:
 *           models/video_models/resnet.py:125-126, train_fusion.py:274,348 -- a mean of equal-sized means)
 *   mode 1: mean | unbiased std [G,2K] (MeanStdPooling, models/audio_models/pooling.py:24-26), optionally in the
 *           split activation format ([G, 2K rounded up to 32], what fc1's kernel reads).
 * partials: caller-owned, dlip_conv_pool_partial_bytes(d, &tile_rows) bytes (8-byte aligned); group_rows must be
 * >= tile_rows (a tile then holds at most one group boundary) -- callers fall back to the unfused kernels
 * otherwise.  Sums are formed in a fixed order: results do not depend on scheduling.
 * RAGGED BATCHES (ABI 43).  The reference extracts one utterance / one clip at a time at its own length
 * (train_fusion.py:334-349: audio [1,24,T_i], clips [1,1,T_j,88,88]); a batch of them is the zero-padded tensor +
 * length vector of pad_packed_collate (models/video_models/dataset.py:123-139).  group_len (device int32 [G], NULL =
 * every group is whole): group g is valid for its first clamp(group_len[g] * len_mul + len_add, 0, group_rows) rows, the
 * rest is padding and enters neither sum nor count -- len_mul / len_add turn the caller's unit into pooled rows (lip
 * clips: frames * Ho*Wo of the last convolution; utterances: input frames - the frames the valid convolutions consume),
 * so one length vector in HBM serves the whole step and a recorded plan replays with new lengths.  The valid rows'
 * values equal the unpadded run's (a valid convolution / a per-frame trunk never reads past them; the stem's pre-pass
 * zeroes the padding frames, dlip_stem3d_pool_f16x3). */
int64_t dlip_conv_pool_partial_bytes(const dlip_conv_desc* d, int32_t* tile_rows);
int dlip_conv_pool_f16x3(const dlip_conv_desc* d, const float* x, const void* w_split, const float* w_scale,
                         const float* bias, const float* residual, const float* slope, const float* post_scale,
                         const float* post_shift, double* partials, int64_t partial_bytes, int32_t group_rows,
                         const int32_t* group_len, int32_t len_mul, int32_t len_add, dlip_stream_t stream);
/* M = rows of the pooled convolution, K its channels, tile_rows as reported by dlip_conv_pool_partial_bytes; group_len /
 * len_mul / len_add as given to dlip_conv_pool_f16x3 (the divisor of group g is its valid row count: the masked mean of
 * models/video_models/model.py:16-17, the statistics of one utterance at its own length).
 * y: mode 0 [G,K]; mode 1 [G,2K] or, with out_split != 0, [G, 2K rounded up to 32] split format (padding zeroed). */
int dlip_pool_finish_f32(const double* partials, int64_t M, int32_t K, int32_t tile_rows, int32_t group_rows,
                         const int32_t* group_len, int32_t len_mul, int32_t len_add,
                         int32_t mode, int32_t out_split, float* y, dlip_stream_t stream);

/* Workspace of dlip_conv_nhwc_f16x3's balanced ("stream-K") work split on split-format activations:
 * ticket counters + partial-tile slabs, one block PER STREAM (launches on a stream are ordered and share
 * it).  dlip_conv_workspace_bytes() = size that covers every launch on the current device;
 * dlip_conv_set_workspace registers a caller-owned device block for `stream` (16-byte aligned; its counter
 * words are zeroed asynchronously on that stream; it must stay valid until replaced, unregistered with
 * ptr = NULL, or the stream's work has completed).  A smaller block is legal: launches that need more
 * than it holds run as plain launches.  A stream without a registered block gets a library-owned one on
 * first use (the only allocation the library ever makes). */
int64_t dlip_conv_workspace_bytes(void);
int dlip_conv_set_workspace(void* ptr, int64_t bytes, dlip_stream_t stream);

/* fp32 [rows, C] -> split activation format (and back), C % 32 == 0.  Boundary converters for callers
 * that hold fp32 tensors; inside the encoders the producers write the split format directly. */
int dlip_split_pack_f32(const float* x, float* y, int64_t rows, int32_t C, dlip_stream_t stream);
int dlip_split_unpack_f32(const float* x, float* y, int64_t rows, int32_t C, dlip_stream_t stream);

/* Reports the workgroup tile (BM x BN) dlip_conv_nhwc_f32 (split_f16 = 0), dlip_conv_nhwc_f16x3 on
 * fp32 activations (split_f16 = 1) or on split-format activations (split_f16 = 3: the LDS-DMA
 * kernel) will use for `d` -- i.e. which conv_igemm_*_kernel<BM,BN,..> instance a profiler will
 * show.  Host-only, no launch. */
int dlip_conv_plan(const dlip_conv_desc* d, int32_t split_f16, int32_t* bm, int32_t* bn);
/* 1 if a split-format (DLIP_SPLIT_IN) launch of `d` runs on the window kernel (conv_win_f16x3_kernel<128,64> for K <= 64, <128,128> above: same-size
 * stride-1 3x3 convolutions with K <= 128 -- one activation window per channel slice in LDS instead of one fetch per
 * tap), 2 if on the rows kernel (conv_rows_f16x3_kernel; dlip_conv_plan then reports its BM x 256 tile): without a residual the
 * speech encoder's 1-D valid convolutions and k = 1 GEMMs over all frames; also, in the lab library only and only while
 * dlip_debug_set(7, 1) forces its general mode, any convolution of whole 32-channel slices and <= 32 taps (such a launch still falls back to the ring kernel when
 * its stream has no split workspace of dlip_conv_workspace_bytes()), 0 if on the LDS-DMA ring kernel dlip_conv_plan describes.
 * Host-only. */
int dlip_conv_kernel_kind(const dlip_conv_desc* d);

/* ------------------------------------------------------------------------------------------
 * Video stem: Conv3d(1->K, 5x7x7, stride (1,2,2), pad (2,3,3), no bias) + BatchNorm3d + PReLU|ReLU
 * (BN folded into w/bias; slope[k] = PReLU weight or 0 for ReLU), output NDHWC = [(B*T),Ho,Wo,K].
 * Replaces models/video_models/model.py:81-84 (frontend3D.0-.2) and makes threeD_to_2D_tensor
 * (model.py:9-13) a no-op.  x is [B,T,H,W] (the reference's [B,1,T,H,W]); w is [K][5][7][7] padded
 * to [K][248]; H, W even.
 * ------------------------------------------------------------------------------------------ */
int dlip_stem3d_bn_act_f32(const float* x, const float* w_k248, const float* bias, const float* slope,
                           float* y, int32_t B, int32_t T, int32_t H, int32_t W, int32_t K,
                           dlip_stream_t stream);

/* Split-fp16 variant of dlip_stem3d_bn_act_f32 (3 x v_mfma_f32_16x16x32_f16 per product, fp32
 * accumulate; fp32 in / fp32 out).  w_split = 64 x 1184 bytes: per output channel a hi plane [36 kernel
 * rows (kt*7+kh; row 35 zero)][8 taps (kw; tap 7 zero)] of fp16, then the lo plane of the same shape, + 32
 * bytes of padding, pre-multiplied by the power of two w_scale[k] (deeplip_amd.packing.pack_stem3d).  W <= 88. */
int dlip_stem3d_bn_act_f16x3(const float* x, const void* w_split, const float* w_scale, const float* bias,
                             const float* slope, float* y, int32_t B, int32_t T, int32_t H, int32_t W,
                             int32_t K, dlip_stream_t stream);

/* dlip_stem3d_bn_act_f16x3 + MaxPool3d((1,3,3), stride (1,2,2), pad (0,1,1)) (replaces
 * models/video_models/model.py:81-85): the pre-pool activations never reach memory.  y is
 * [(B*T), Hp, Wp, 64] with Hp = (H/2 - 1)/2 + 1, in the split activation format of
 * dlip_conv_nhwc_f16x3 (what the trunk's first layers read).  H, W even, W <= 88.
 * x_split: caller-owned scratch of dlip_stem3d_pool_workspace_bytes(B, T, H, W) bytes, 16-B aligned -- a
 * pre-pass writes the clip there once as (hi, lo) fp16 pairs at the kernel's window row pitch, and the
 * kernel fetches its windows from it by LDS-DMA (two launches on `stream`).
 * lengths (device int32 [B], NULL = every clip has T frames; ABI 43): clip b's frames t >= lengths[b] are padding
 * (pad_packed_collate, models/video_models/dataset.py:123-139) and the pre-pass writes them as ZEROS of the normalised clip
 * whatever x holds there -- for the frames t < lengths[b] that is the Conv3d's own zero padding behind the clip's last frame,
 * so their outputs equal the clip run alone at its own length (train_fusion.py:346-348). */
int64_t dlip_stem3d_pool_workspace_bytes(int32_t B, int32_t T, int32_t H, int32_t W);
int dlip_stem3d_pool_f16x3(const float* x, const int32_t* lengths, void* x_split, const void* w_split, const float* w_scale,
                           const float* bias, const float* slope, float* y, int32_t B, int32_t T, int32_t H,
                           int32_t W, int32_t K, dlip_stream_t stream);

/* dlip_stem3d_pool_f16x3 fed with the frames as a loader hands them over: uint8, `channels` = 1 (gray [B,T,Hs,Ws], what the
 * reference's npz mouth crops hold: models/video_models/dataset.py) or 3 (RGB [B,T,3,Hs,Ws], BASELINE.json's input shape).
 * The pre-pass crops H x W at row oy, column ox (CenterCrop, preprocess.py:89-90: oy = (Hs - H) / 2 rounded DOWN) and
 * normalises while it writes the split clip: gray = 0.299 R + 0.587 G + 0.114 B kept in float (preprocess.py:32-46),
 * (gray / 255 - 0.421) / 0.165 (dataloaders.py:11-22), mul / add / IEEE divide in that order, uncontracted -- bit-identical
 * to dlip_ingest_rgb_u8 (or dlip_crop_normalize_u8) followed by dlip_stem3d_pool_f16x3, without the fp32 clip: a quarter of
 * the host-to-device bytes, one HBM write and one read of 4 B per pixel less.  Replaces train_fusion.py:346-348's
 * `.to(device)` of a float clip + model.py:81-85.
 * clip_params (device int32 [B][4], NULL = (oy, ox) for every clip, no flip; ABI 43): per clip (oy_b, ox_b, flip_b, 0) -- the
 * train pipeline's RandomCrop origin and HorizontalFlip coin (preprocess.py:95-138, dataloaders.py:13-17: ONE draw per clip, all
 * of its frames alike; the host draws them from its seeded generator); flip_b != 0 mirrors the cropped frame left-right
 * (pixel w <- column ox_b + W - 1 - w).  Origins are clamped into the frame.  lengths: as dlip_stem3d_pool_f16x3 (here the
 * only way to pad: a zero byte is not a zero of the normalised clip). */
int dlip_stem3d_pool_u8_f16x3(const uint8_t* frames, int32_t channels, int32_t Hs, int32_t Ws, int32_t oy, int32_t ox,
                              const int32_t* clip_params, const int32_t* lengths,
                              void* x_split, const void* w_split, const float* w_scale, const float* bias,
                              const float* slope, float* y, int32_t B, int32_t T, int32_t H, int32_t W, int32_t K,
                              dlip_stream_t stream);

/* Self-test of a hardware behaviour the layer-1 / layer-2 window kernel (conv_win_f16x3.hip) relies on: a ds_read_b128 at an
 * LDS address with bit 18 set (beyond every allocation) returns zeros on gfx950 -- the kernel uses that as the zero
 * padding of taps outside the image, and is selected on gfx950 devices only.  `blocks` workgroups of 256 lanes, 8 KB of
 * LDS each (several per CU), fill their allocation with a pattern and read inside and beyond it; counts (device int32[2])
 * receives [lanes whose in-range read returned the pattern, lanes whose out-of-range reads returned anything non-zero]:
 * expected [256 * blocks, 0]. */
int dlip_selftest_lds_oob(int32_t* counts, int32_t blocks, dlip_stream_t stream);

/* MaxPool3d((1,3,3), stride (1,2,2), pad (0,1,1)) on NHWC: [N,H,W,C] -> [N,Ho,Wo,C],
 * Ho = (H+2-3)/2+1.  Replaces models/video_models/model.py:85.  C % 4 == 0.  out_split != 0 writes y
 * in the split activation format of dlip_conv_nhwc_f16x3 (C % 32 == 0). */
int dlip_maxpool3x3s2_nhwc_f32(const float* x, float* y, int32_t N, int32_t H, int32_t W, int32_t C,
                               int32_t out_split, dlip_stream_t stream);

/* AdaptiveAvgPool2d(1) + flatten on NHWC: [N,HW,C] -> [N,C].  Replaces resnet.py:83,125-126. */
int dlip_avgpool_nhwc_f32(const float* x, float* y, int32_t N, int32_t HW, int32_t C,
                          dlip_stream_t stream);

/* Masked temporal mean: y[b,c] = mean_{t < len[b] + len_add} x[b,t,c]; len == NULL means T for every b.
 * Replaces torch.mean(..., dim=0) over frames (train_fusion.py:274,348) and _average_batch
 * (models/video_models/model.py:16-17).  x [B,T,C] with row stride ldx floats. */
int dlip_time_mean_f32(const float* x, const int32_t* len, int32_t len_add, float* y, int32_t B, int32_t T, int32_t C,
                       int32_t ldx, dlip_stream_t stream);

/* Ragged batches on the paths without a stem pre-pass (exact-fp32 packing, taps): y[b,t,:] = t < len[b] ? x[b,t,:] : 0 for x
 * [B,T,E] -- the padding frames of pad_packed_collate (models/video_models/dataset.py:123-139) as zeros of the normalised
 * clip.  E % 4 == 0, x and y 16-byte aligned (y may be x). */
int dlip_mask_frames_f32(const float* x, const int32_t* len, float* y, int32_t B, int32_t T, int32_t E, dlip_stream_t stream);

/* Segmented mean over clip groups (CSR offsets, G+1 entries): y[u] = sum(x[ptr[u]:ptr[u+1]]) / count.
 * Replaces the per-utterance clip-file average (train_fusion.py:272-275,346-349). */
int dlip_group_mean_f32(const float* x, const int32_t* group_ptr, float* y, int32_t U, int32_t C,
                        dlip_stream_t stream);

/* MeanStdPooling on N-T-C: y[b, 0:C] = mean_t, y[b, C:2C] = unbiased std_t (N-1); sums of x and x^2
 * in fp64.  Replaces models/audio_models/pooling.py:24-26.  C % 4 == 0.  out_split != 0 writes y as
 * [B, 2C rounded up to 32] in the split activation format of dlip_conv_nhwc_f16x3 (padding zeroed).
 * len (device int32 [B], NULL = T for every b; ABI 43): utterance b's statistics cover its first
 * clamp(len[b] + len_add, 0, T) frames -- a zero-padded batch pooled as the reference pools each utterance at its own
 * length (train_fusion.py:334-338); len_add = -(frames the valid convolutions in front consumed) when len counts input frames. */
int dlip_meanstd_pool_f32(const float* x, const int32_t* len, int32_t len_add, float* y, int32_t B, int32_t T, int32_t C,
                          int32_t out_split, dlip_stream_t stream);

/* AttentiveStatPooling tail (models/audio_models/pooling.py:87-107) on N-T-C: hidden [B,T,Hd] = x W^T + b
 * (from dlip_conv_nhwc_f32), e = relu(hidden).v + k, alpha = softmax over T, y [B,2C] = weighted mean |
 * sqrt(weighted E[x^2] - mean^2).  len / len_add: as dlip_meanstd_pool_f32 (softmax and statistics over the valid frames). */
int dlip_attentive_stat_pool_f32(const float* x, const float* hidden, const float* v, const float* k, const int32_t* len,
                                 int32_t len_add, float* y, float* alpha_out, int32_t B, int32_t T, int32_t C, int32_t Hd,
                                 dlip_stream_t stream);
/* (ABI 45) alpha_out (nullable, [B,T]) of the entry point above: the attention weights, kept for the backward pass (zeros behind a
 * ragged utterance's end).  The backward of that tail for a training step (models/audio_models/pooling.py:87-107 under
 * model.train(), selected by the config's `pooling: attentive_statistic`, tdnn.py:66-75; train_audio.py:185-200): from y [B,2C]
 * and dy [B,2C], with g_q = ds / (2 s), g_m = dm - 2 m g_q:
 *   dx [B,T,C]      = alpha_t (g_m + 2 x g_q)                              (the statistics' direct path; zeros in padding frames)
 *   de [B,T]        = alpha_t (dalpha_t - sum_t' alpha_t' dalpha_t'),  dalpha_t = sum_c (g_m x + g_q x^2)
 *   dhidden [B,T,Hd]= de_t v_j [hidden > 0]      (through the GEMM's own backward: dW, db and the second path into x)
 *   rde [B,T,Hd]    = de_t relu(hidden)          (its column sums over B*T rows = dv; the sum of de = dk)
 * One workgroup per utterance, fp64 wherever a sum runs over channels or frames; (2 C + T) * 4 bytes of LDS <= 60 KB. */
int dlip_attentive_stat_pool_bwd_f32(const float* x, const float* hidden, const float* v, const float* alpha, const float* y,
                                     const float* dy, const int32_t* len, int32_t len_add, float* dx, float* dhidden, float* rde,
                                     float* de, int32_t B, int32_t T, int32_t C, int32_t Hd, dlip_stream_t stream);

/* Layout adapters at the API boundary.
 *   dlip_nct_to_ntc_f32: x [B,C,T] (reference layout, tdnn.py:89) -> y [B,T,Cp] zero-padded to Cp>=C.
 *   dlip_ntc_to_nct_f32: y [B,C,T] <- x [B,T,C].
 *   dlip_ingest_rgb_u8 : [B,T,3,H,W] uint8 RGB -> [B,T,H,W] float = ((0.299R+0.587G+0.114B)/255-0.421)/0.165
 *                        (constants: models/video_models/dataloaders.py:12,21-22; build-owned adapter). */
int dlip_nct_to_ntc_f32(const float* x, float* y, int32_t B, int32_t C, int32_t T, int32_t Cp,
                        dlip_stream_t stream);
int dlip_ntc_to_nct_f32(const float* x, float* y, int32_t B, int32_t T, int32_t C,
                        dlip_stream_t stream);
/* dlip_nct_to_ntc_f32 + dlip_split_pack_f32 in one pass: y [B,T,Cp] in the split activation format, Cp % 32 == 0. */
int dlip_nct_to_ntc_split_f32(const float* x, float* y, int32_t B, int32_t C, int32_t T, int32_t Cp,
                              dlip_stream_t stream);
int dlip_ingest_rgb_u8(const uint8_t* x, float* y, int64_t n_frames, int32_t H, int32_t W,
                       dlip_stream_t stream);

/* Eval-mode BatchNorm1d (as per-channel scale/shift) and LeakyReLU(slope) on [M,C]:
 * order 0: y = lrelu(x*scale + shift)  (bn_first, tdnn.py:92-94,105-107; model_fusion.py:21-22)
 * order 1: y = lrelu(x)*scale + shift  (tdnn.py:96-97,109-110). */
int dlip_affine_act_f32(const float* x, const float* scale, const float* shift, float* y, int64_t M,
                        int32_t C, float slope, int32_t order, dlip_stream_t stream);

/* Per-row z-normalisation with the UNBIASED std (train_fusion.py:233-238) of a [U,Da] and a [U,Dv]
 * table, concatenated into y [U, Da+Dv] (train_fusion.py:353-358).  Either input may be NULL
 * (D = 0) to z-normalise a single table.  biased != 0 selects numpy's biased std
 * (models/fusion_models/utils.py:524-527, feature-fusion scoring). */
int dlip_znorm_cat_f32(const float* a, int32_t Da, const float* v, int32_t Dv, float* y, int32_t U,
                       int32_t biased, dlip_stream_t stream);

/* dlip_znorm_cat_f32 whose second table is the per-clip mean still held as pooled partial sums (dlip_conv_pool_f16x3
 * over groups of T*Ho*Wo rows; M, K, tile_rows, group_rows as for dlip_pool_finish_f32, U = number of groups): the
 * temporal mean of train_fusion.py:348 and the fusion of :353-358 in one launch; bit-identical to
 * dlip_pool_finish_f32 (mode 0) followed by dlip_znorm_cat_f32 (group_len / len_mul / len_add as there). */
int dlip_znorm_cat_pooled_f32(const float* a, int32_t Da, const double* partials, int64_t M, int32_t K,
                              int32_t tile_rows, int32_t group_rows, const int32_t* group_len, int32_t len_mul,
                              int32_t len_add, float* y, int32_t U, int32_t biased, dlip_stream_t stream);

/* y[u,:] = x[u,:] / max(||x[u,:]||_2, eps)   (F.normalize; loss.py:44, train_audio.py:355). */
int dlip_l2_normalize_f32(const float* x, float* y, int32_t U, int32_t D, float eps,
                          dlip_stream_t stream);

/* PLDA trial scoring (eer_plda_lomgrid / eer_plda_grid, models/fusion_models/utils.py:285-329: the `plda`
 * package's model.transform(..., 'D' -> 'U_model') followed by calc_same_diff_log_likelihood_ratio per trial).
 * u [N,D] = the embeddings already mapped to the model's latent space (an affine map: one dlip_conv_nhwc_f32
 * GEMM, deeplip_amd/plda.py), psi [D] = the between-class variances there (within-class covariance = I):
 * score[i] = log p(u_a, u_b | same speaker) - log p(u_a) - log p(u_b). */
int dlip_plda_llr_f32(const float* u, int32_t N, int32_t D, const float* psi, const int32_t* idx_a,
                      const int32_t* idx_b, float* score, int32_t n_trials, dlip_stream_t stream);

/* Trial scoring over an [N,D] embedding table: score[i] = cos(emb[idx_a[i]], emb[idx_b[i]]).
 * mode 0: sklearn cosine_similarity semantics (normalise each row, then dot; utils.py:244,262)
 * mode 1: F.cosine_similarity(eps) semantics  (dot / max(||a||*||b||, eps); utils.py:372).
 * score_io: if accumulate != 0, score[i] = score[i] + weight*cos, else score[i] = weight*cos
 * (score-level fusion 0.5/0.5, utils.py:343-377). */
int dlip_pair_cosine_f32(const float* emb, int32_t N, int32_t D, const int32_t* idx_a,
                         const int32_t* idx_b, float* score, int32_t n_trials, int32_t mode,
                         float eps, float weight, int32_t accumulate, dlip_stream_t stream);

/* logits[b,k] = <normalize(e[b]), normalize(W[k])> (cosine logits, loss.py:44) when cosine != 0,
 * else <e[b], W[k]> + bias[k] (loss.py:14); argmax[b] = first index of the row maximum
 * (torch.max(logits,1)[1]; train_fusion.py:296), int64.  K <= 1024. */
int dlip_logits_argmax_f32(const float* e, const float* W, const float* bias, float* logits,
                           int64_t* argmax, int32_t B, int32_t D, int32_t K, int32_t cosine,
                           dlip_stream_t stream);

/* Softmax cross-entropy over margin-adjusted logits (forward value only):
 *   z[b,k] = scale * (logits[b,k] - margin*[k==label[b]]) + 1e-8;  loss = mean_b( lse(z[b]) - z[b,label] )
 * LMCL: scale=s, margin=m (loss.py:45-48); CrossEntropy: scale=1, margin=0 (loss.py:15).
 * loss is one float on the device. */
int dlip_margin_ce_loss_f32(const float* logits, const int64_t* labels, float* loss, int32_t B,
                            int32_t K, float scale, float margin, dlip_stream_t stream);

/* LowFER.forward as shipped (LBP.py:46-50): y = cat[e1, sigmoid(e2), sigmoid(e2)*e1], [B,3D]. */
int dlip_lowfer_cat_f32(const float* e1, const float* e2, float* y, int32_t B, int32_t D,
                        dlip_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * GPU-side preprocessing in front of the hot path.  Audio features as
 * the reference computes them with python_speech_features (models/audio_models/datasets.py:65-83,52-53);
 * the DFT / mel / DCT matrix products go through dlip_conv_nhwc_f32 (deeplip_amd/frontend.py).
 * ------------------------------------------------------------------------------------------ */
/* x [B,S] waveform -> frames [B*NF, nfft]: frame f = pre-emphasised samples [f*step, f*step+len), zero
 * padded to nfft and past the end of the signal (sigproc.preemphasis + framesig, rectangular window). */
int dlip_frame_preemph_f32(const float* x, float* frames, int32_t B, int32_t S, int32_t NF, int32_t frame_len,
                           int32_t frame_step, int32_t nfft, float preemph, dlip_stream_t stream);
/* spec [R, 2*NB] (re | im) -> pw [R, NBp] = |.|^2 / nfft (zero padded), energy [R] = row sum (0 -> eps). */
int dlip_powspec_f32(const float* spec, float* pw, float* energy, int32_t R, int32_t NB, int32_t NBp,
                     int32_t nfft, dlip_stream_t stream);
/* frames [R, nfft] (dlip_frame_preemph_f32) -> pw / energy as dlip_powspec_f32, the DFT itself evaluated in fp64 (numpy's
 * rfft under python_speech_features is double): for filterbanks whose lowest bands hold ~1e-9 of the spectrum (80 bands at
 * nfft 512) the fp32-GEMM DFT's rounding noise is larger than the signal.  nfft a power of two <= 1024. */
int dlip_powspec_dft64_f32(const float* frames, float* pw, float* energy, int32_t R, int32_t NB, int32_t NBp,
                           int32_t nfft, dlip_stream_t stream);
/* (ABI 46) x [B,S] waveform -> pw [B*NF, NBp] / energy [B*NF] as above, with pre-emphasis, framing and the DFT (a radix-2 FFT in LDS, one
 * workgroup per frame) ALL in fp64, as python_speech_features computes them (sigproc.preemphasis / framesig / numpy rfft on doubles;
 * features/audio front-end of models/audio_models/datasets.py:60-82).  `preemph` is a double: the reference's 0.97 is.  What the fp32 routes
 * lose: elements holding ~1e-12 of a frame's energy (the lowest mel band of a frame that pre-emphasis empties), 0.3 .. 1 % there.  The
 * default route of deeplip_amd.frontend.AudioFrontend since ABI 46.  nfft a power of two, 128 .. 1024. */
int dlip_powspec_wave_fft64_f32(const float* x, float* pw, float* energy, int64_t B, int64_t S, int32_t NF, int32_t frame_len,
                                int32_t frame_step, int32_t nfft, double preemph, int32_t NB, int32_t NBp, dlip_stream_t stream);
/* y = log(x == 0 ? eps : x). */
int dlip_log_floor_f32(const float* x, float* y, int64_t n, dlip_stream_t stream);
/* feat [B,NF,C] (row stride ldf; channel 0 := log(energy) when energy != NULL) -> per-utterance
 * (x - mean)/(std + 2e-12) when normalize != 0 -> y [B,C,NF] (the reference loaders' layout). */
int dlip_cmvn_nct_f32(const float* feat, const float* energy, float* y, int32_t B, int32_t NF, int32_t C,
                      int32_t ldf, int32_t normalize, dlip_stream_t stream);
/* Delta features as SpkTrainDataset._delta appends them (models/audio_models/datasets.py:55-63, `delta: true`):
 * x [B,C,NF] -> y [B,(1+order)C,NF] = [x | delta(x, N=1) | delta(x, N=2)], python_speech_features.delta with edge
 * padding; order 1 or 2. */
int dlip_delta_nct_f32(const float* x, float* y, int32_t B, int32_t C, int32_t NF, int32_t order, dlip_stream_t stream);
/* uint8 frames [n, channels(1|3), H, W] (n = B clips of T frames) -> crop [n, crop, crop] float = ((gray)/255 - 0.421)/0.165
 * (models/video_models/dataloaders.py:11-22; RGB -> gray with the BT.601 weights).  clip_params NULL: the "val" / "test"
 * pipeline's CenterCrop (preprocess.py:89-90); clip_params (device int32 [B][4] = (oy, ox, flip, 0); ABI 43): the "train"
 * pipeline's RandomCrop + HorizontalFlip (preprocess.py:95-138, dataloaders.py:13-17), one host draw per clip.  lengths
 * (device int32 [B], NULL = whole clips): frames t >= lengths[b] are written as zeros of the normalised clip -- the padding
 * of pad_packed_collate, which pads AFTER the normalisation (dataset.py:117,130-134). */
int dlip_crop_normalize_u8(const uint8_t* x, const int32_t* clip_params, const int32_t* lengths, int32_t T, float* y,
                           int64_t n_frames, int32_t channels, int32_t H, int32_t W, int32_t crop, dlip_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Train-mode / backward kernels for the trainable fusion head and criterion (config C5; the
 * encoders stay frozen, train_fusion.py:198-201).  Replace the autograd of
 *   nn.Linear / nn.BatchNorm1d(train) / nn.LeakyReLU   models/fusion_models/model_fusion.py:19-24
 *   F.cross_entropy, F.normalize, F.linear              models/audio_models/loss.py:13-16,43-51
 * ------------------------------------------------------------------------------------------ */
/* y = lrelu_slope(BN_train(x)) on [M,C]: batch mean / biased variance, saves mean and 1/std for the
 * backward, updates running stats (unbiased variance, `momentum`) when the pointers are non-NULL.
 * slope = 1 gives plain BatchNorm1d. */
int dlip_bn1d_train_fwd_f32(const float* x, const float* gamma, const float* beta, float* y,
                            float* save_mean, float* save_invstd, float* running_mean,
                            float* running_var, int32_t M, int32_t C, float momentum, float eps,
                            float slope, dlip_stream_t stream);
int dlip_bn1d_train_bwd_f32(const float* dy, const float* x, const float* save_mean,
                            const float* save_invstd, const float* gamma, float* dx, float* dgamma,
                            float* dbeta, int32_t M, int32_t C, dlip_stream_t stream);
/* dx = dy * (y >= 0 ? 1 : slope)  (LeakyReLU backward from the OUTPUT; slope > 0). */
int dlip_lrelu_bwd_f32(const float* dy, const float* y, float* dx, int64_t n, float slope,
                       dlip_stream_t stream);
/* y[c] = sum_m x[m,c]  (bias gradients). */
int dlip_colsum_f32(const float* x, float* y, int32_t M, int32_t C, dlip_stream_t stream);
/* out[0] = sum |w[i]| (fp64 accumulation, fixed order) -- the L1 regulariser of LMCL, 1e-5 * ||W||_1 (models/audio_models/loss.py:
 * 49-50); dlip_l1_sign_f32: dw[i] = coef * grad_scale_dev[0] * sign(w[i]) (grad_scale_dev NULL: 1), its gradient. */
int dlip_l1_sum_f32(const float* w, float* out, int64_t n, dlip_stream_t stream);
int dlip_l1_sign_f32(const float* w, const float* grad_scale_dev, float* dw, float coef, int64_t n, dlip_stream_t stream);

/* dlogits = grad_scale * grad_scale_dev[0] * d/dlogits mean_b CE(scale*(logits - margin*onehot) + 1e-8, labels);
 * grad_scale_dev = the upstream gradient of the scalar loss as a DEVICE scalar (NULL = 1): autograd's backward
 * does not have to read it back to the host. */
int dlip_margin_ce_bwd_f32(const float* logits, const int64_t* labels, float* dlogits, int32_t B,
                           int32_t K, float scale, float margin, float grad_scale,
                           const float* grad_scale_dev, dlip_stream_t stream);
/* Backward of dlip_l2_normalize_f32. */
int dlip_l2_normalize_bwd_f32(const float* x, const float* dy, float* dx, int32_t U, int32_t D,
                              float eps, dlip_stream_t stream);
/* C[M,N] = op(A)[M,K] * op(B)[K,N], row-major, any sizes (head gradients: dX = dY W, dW = dY^T X). */
int dlip_gemm_small_f32(const float* A, const float* B, float* C, int32_t M, int32_t N, int32_t K,
                        int32_t trans_a, int32_t trans_b, dlip_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Train-mode kernels of the speech encoder (deeplip_amd/csrc/encoder_train_ops.hip).
 * They serve torch.autograd Functions in deeplip_amd/autograd.py that replace what autograd does for
 * SpeakerEmbNet.forward under model.train() (train_audio.py:167-183; models/audio_models/tdnn.py:35-43,
 * 89-111; pooling.py:24-26).  Column reductions over the M = B*T' rows are deterministic (fixed 512-row
 * chunks, fp64 partials in `workspace` = dlip_bn_rows_chunks(M) * C * 2 doubles, added in chunk order).
 * ------------------------------------------------------------------------------------------ */
int32_t dlip_bn_rows_chunks(int32_t M);

/* BatchNorm over rows, training mode, fused with LeakyReLU(slope): x, y [M,C], C % 4 == 0.
 * act_first = 0: y = lrelu(bn(x))  (TDNN_Block bn_first=True, tdnn.py:36-38; bn1/bn2 in tdnn.py:93-94,106-107)
 * act_first = 1: y = bn(lrelu(x))  (bn_first=False, tdnn.py:40-42,96-97,109-110).
 * Batch statistics: biased variance for the normalisation, unbiased for running_var (momentum update,
 * nullable pair), as nn.BatchNorm1d.  save_mean / save_invstd [C] feed the backward. */
/* (ABI 44) y == NULL: statistics only (save_mean, save_invstd, running statistics, the counter) -- the consumer applies the
 * normalisation and the activation on load (dlip_wgrad_operand_split_bn_f32 / dlip_wgrad_chwn_bn_f32) and the activated tensor is
 * never stored.
 * (ABI 44) num_batches_tracked (nullable): the module's int64 counter, incremented by the launch that finishes the statistics
 * (nn.BatchNorm*.forward under model.train(): `self.num_batches_tracked += 1` -- 44 one-element torch launches per lip-clip step).
 * Launch sequence since ABI 44: the finalize steps (partials -> mean / 1/std, -> dgamma / dbeta / dslope, -> the lift of dx) run in
 * the LAST workgroup of the pass before them (a ticket word of the stream's dlip_conv_set_workspace block, when the stream has one)
 * instead of in launches of their own, the parts of a column reduction grow beyond 512 rows so that there are at most 512 of them,
 * and tensors of M <= 4096 rows (the MS-TCN head's, tcn.py:42-43) take ONE launch per direction.  dlip_debug_set(8, 0) restores
 * the ABI 43 sequence.  Results are deterministic either way; the association of the fp64 column sums differs between the two.
 * (ABI 43) ready_chunks > 0 (act_first == 0 only): `workspace` already holds that many partial rows [chunk][C][2] fp64 = {sum x,
 * sum x^2} of x, written by the convolution that produced x (dlip_conv_nhwc_stats_f16x3): the statistics pass over x is skipped. */
int dlip_bn_rows_train_fwd_f32(const float* x, const float* gamma, const float* beta, float* y,
                               float* save_mean, float* save_invstd, float* running_mean,
                               float* running_var, double* workspace, int32_t M, int32_t C, float momentum,
                               float eps, float slope, int32_t act_first, int32_t ready_chunks, int64_t* num_batches_tracked,
                               dlip_stream_t stream);
/* Backward of the above: dy = dL/dy -> dx = dL/dx [M,C], dgamma, dbeta [C].  dx_lift2 (nullable, DLIP_LIFT_WORDS floats: the pair, then per-workgroup scratch): the power-of-two
 * lift of dx, (2^e, 2^-e) with max|dx| * 2^e in [512, 1024] -- what dlip_pow2_scale_f32(dx, ., 1024) would return, formed by the
 * pass that writes dx: the convolution backward that consumes dx needs it and would otherwise read dx once more. */
int dlip_bn_rows_train_bwd_f32(const float* dy, const float* x, const float* gamma, const float* beta,
                               const float* save_mean, const float* save_invstd, float* dx, float* dgamma,
                               float* dbeta, double* workspace, int32_t M, int32_t C, float slope,
                               int32_t act_first, float* dx_lift2, dlip_stream_t stream);
/* BatchNorm (batch statistics) + PReLU with PER-CHANNEL slopes in the same passes: y = prelu(bn(x)) (resnet.py:51-53 bn1 + relu1,
 * tcn.py:42-43, model.py:83-84 under model.train()).  Forward as dlip_bn_rows_train_fwd_f32 with slope[C] applied behind the
 * affine; backward additionally returns dslope[C] = sum over rows of (bn(x) < 0 ? dy * bn(x) : 0) from the SAME pass that
 * forms dgamma / dbeta -- `workspace` holds 2 * dlip_bn_rows_chunks(M) * C * 2 doubles here.  Saves the separate PReLU
 * forward / backward passes and the column sum of its slope terms. */
int dlip_bn_prelu_rows_train_fwd_f32(const float* x, const float* gamma, const float* beta, const float* slope, float* y,
                                     float* save_mean, float* save_invstd, float* running_mean, float* running_var,
                                     double* workspace, int32_t M, int32_t C, float momentum, float eps,
                                     int64_t* num_batches_tracked, dlip_stream_t stream);
int dlip_bn_prelu_rows_train_bwd_f32(const float* dy, const float* x, const float* gamma, const float* beta, const float* slope,
                                     const float* save_mean, const float* save_invstd, float* dx, float* dgamma, float* dbeta,
                                     float* dslope, double* workspace, int32_t M, int32_t C, float* dx_lift2, dlip_stream_t stream);
/* (ABI 44) BatchNorm3d (batch statistics) + PReLU + MaxPool3d((1,3,3),(1,2,2),(0,1,1)) of the stem under model.train()
 * (models/video_models/model.py:83-85) WITHOUT the full-resolution tensors between them: x [N,H,W,C] = the stem convolution's output.
 * (dy_pooled2, nullable: a second gradient of the pooled output, added on the fly -- the first BasicBlock's convolution and its
 * shortcut both consume it.)
 * Forward: statistics of x (workspace as dlip_bn_rows_train_fwd_f32 with M = N H W), then ONE pass writes the pooled y
 * [N,Ho,Wo,C] and the argmax codes idx [N,Ho,Wo,C/4] (dlip_maxpool3x3s2_idx_f32's: tap r*3+s of the first maximum, one byte per
 * channel); prelu(bn(x)) is never stored.  Backward: dy_pooled [N,Ho,Wo,C] -> dx [N,H,W,C], dgamma, dbeta, dslope: both passes form
 * the gradient behind the pooling per input pixel from the <= 4 windows covering it (idx, dy_pooled) instead of reading a scattered
 * full-resolution gradient (workspace: 2 * dlip_bn_rows_chunks(N H W) * C * 2 doubles; dx_lift2 as above).  At B = 32 the stem's
 * train-mode passes move 3.2 GB instead of 5.6 GB. */
int dlip_bn_prelu_maxpool_train_fwd_f32(const float* x, const float* gamma, const float* beta, const float* slope, float* y,
                                        uint32_t* idx, float* save_mean, float* save_invstd, float* running_mean, float* running_var,
                                        double* workspace, int64_t N, int32_t H, int32_t W, int32_t C, float momentum, float eps,
                                        int64_t* num_batches_tracked, dlip_stream_t stream);
int dlip_bn_prelu_maxpool_train_bwd_f32(const float* dy_pooled, const float* dy_pooled2, const uint32_t* idx, const float* x, const float* gamma,
                                        const float* beta, const float* slope, const float* save_mean, const float* save_invstd,
                                        float* dx, float* dgamma, float* dbeta, float* dslope, double* workspace, int64_t N, int32_t H,
                                        int32_t W, int32_t C, float* dx_lift2, dlip_stream_t stream);
/* (ABI 44) The END of a BasicBlock under model.train(): y = prelu(bn2(x) + residual), x = conv2's output
 * (models/video_models/resnet.py:62-69: `out = self.bn2(out); out += residual; out = self.relu2(out)`), rows x [M,C].
 * Forward: statistics of x, then ONE pass writes sum = bn2(x) + residual (kept for the backward) and y; bn2's output is never stored.
 * Backward: dy [M,C] (+ dy2, nullable: a second gradient of y -- the next block's first convolution AND its shortcut both consume y --
 * added on the fly) -> dresidual [M,C] = the gradient behind the PReLU (the shortcut's gradient and bn2's incoming one), dx [M,C],
 * dgamma, dbeta (bn2), dslope (relu2): the first pass forms dresidual, the slope gradient's terms and bn2's two sums from one read of
 * dy, sum and x (was: PReLU backward writing two tensors, a column sum, the BatchNorm's sums pass); the second is the BatchNorm
 * backward's apply pass.  workspace: 2 * dlip_bn_rows_chunks(M) * C * 2 doubles; dx_lift2 as above. */
int dlip_bn_add_prelu_rows_train_fwd_f32(const float* x, const float* residual, const float* gamma, const float* beta,
                                         const float* slope, float* sum, float* y, float* save_mean, float* save_invstd,
                                         float* running_mean, float* running_var, double* workspace, int32_t M, int32_t C,
                                         float momentum, float eps, int64_t* num_batches_tracked, dlip_stream_t stream);
int dlip_bn_add_prelu_rows_train_bwd_f32(const float* dy, const float* dy2, const float* sum, const float* x, const float* gamma,
                                         const float* beta, const float* slope, const float* save_mean, const float* save_invstd,
                                         float* dresidual, float* dx, float* dgamma, float* dbeta, float* dslope, double* workspace,
                                         int32_t M, int32_t C, float* dx_lift2, dlip_stream_t stream);
/* (ABI 44) The apply pass alone: y = act((x - mean) invstd gamma + beta) on [M,C] rows from statistics already formed (a deferred
 * activation -- dlip_bn_rows_train_fwd_f32 with y == NULL -- whose consumer turns out to need the values after all). */
int dlip_bn_apply_rows_f32(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                           const float* slope_vec, float slope, float* y, int32_t M, int32_t C, dlip_stream_t stream);
/* y[c] = sum_m x[m,c] (bias gradients), same chunked reduction and workspace. */
int dlip_colsum_rows_f32(const float* x, float* y, double* workspace, int32_t M, int32_t C, dlip_stream_t stream);

/* MeanStdPooling backward (pooling.py:24-26): x [B,T,C], y = forward output [B,2C] (mean | unbiased std),
 * dy [B,2C] -> dx [B,T,C].  C % 4 == 0, T > 1. */
int dlip_meanstd_pool_bwd_f32(const float* x, const float* y, const float* dy, float* dx, int32_t B, int32_t T,
                              int32_t C, dlip_stream_t stream);

/* y = x permuted: x [d0,d1,d2]; output axis i is input axis p_i; flip_axis (-1 = none) reverses one INPUT
 * axis.  Conv1d weight layouts between the reference's [K,C,S], the forward kernel's [K,S,C], the data
 * gradient's flipped [C,S,K] and the weight gradient's [S,C,K]. */
int dlip_permute3_f32(const float* x, float* y, int32_t d0, int32_t d1, int32_t d2, int32_t p0, int32_t p1,
                      int32_t p2, int32_t flip_axis, dlip_stream_t stream);

/* scale2[0] = 2^floor(log2(target / max|x|)), scale2[1] = 1 / scale2[0] (device scalars; 1 if x == 0):
 * the power-of-two that lifts a gradient tensor into fp16's normal range before it is split. */
int dlip_pow2_scale_f32(const float* x, float* scale2, int64_t n, float target, dlip_stream_t stream);
/* The same into a DLIP_LIFT_WORDS buffer (layout above): lift[0] = 2^e, lift[1] = 2^-e, lift[2 .. 2049] = 2^-e. */
int dlip_pow2_lift_f32(const float* x, float* lift, int64_t n, float target, dlip_stream_t stream);
/* dlip_split_pack_f32 of x * scale[0] (device scalar). */
int dlip_split_pack_scaled_f32(const float* x, float* y, const float* scale, int64_t rows, int32_t C,
                               dlip_stream_t stream);
/* The same with each row zero-padded from C (a multiple of 4) to C_pad (a multiple of 32) channels: y [rows, C_pad]. */
int dlip_split_pack_scaled_pad_f32(const float* x, float* y, const float* scale, int64_t rows, int32_t C, int32_t C_pad,
                                   dlip_stream_t stream);
/* Split-fp16 operand image of a weight matrix on the device (what deeplip_amd.packing.split_weights builds on the host at load
 * time; a training step needs it from the CURRENT weights): w [K rows][L], L % 32 == 0 with every 32-channel block intact ->
 * w_split [K][L] (per block 32 hi halves | 32 lo halves of w * w_scale[k]), w_scale[k] = 2^floor(log2(1023 / max|w[k,:]|))
 * (1 for a zero row) -- the w_split / w_scale pair dlip_conv_nhwc_f16x3 takes. */
int dlip_split_weights_rows_f32(const float* w, float* w_split, float* w_scale, int32_t K, int32_t L, dlip_stream_t stream);
/* The same straight from the reference layout w [K, C, T] (T = R*S taps: nn.Conv2d / nn.Conv1d weights), permutation included:
 * mode 0 = the forward operand, K rows of [T][C]; mode 1 = the data-gradient operand, C rows of [T reversed][K] (the flipped,
 * transposed filter).  w_scale has one entry per output row; the row's inner channel count (C resp. K) must be a multiple of 32. */
int dlip_split_weights_perm_f32(const float* w_kct, float* w_split, float* w_scale, int32_t K, int32_t C, int32_t T, int32_t mode,
                                int32_t C_pad, dlip_stream_t stream);
/* C_pad (0 = none): the rows' inner channel count (C in mode 0, K in mode 1) zero-padded to C_pad -- a first layer whose input has
 * 24 feature channels reads its activations padded to 32; the data gradient of a 1500-channel layer reads its gradient padded
 * to 1504 (tdnn.py:52-62 on conf/audio_config.yaml's input_dim / hidden_dim). */
/* (round 4) dlip_split_weights_perm_f32 for MANY weight tensors in ONE launch -- a training step splits the current weights of every
 * convolution twice (forward and data-gradient operand): `descs` is a DEVICE array of dlip_wsplit_desc (one per output tensor),
 * `block_desc` a device int32[n_blocks] naming, for each output row of each tensor in order, its descriptor (row r of descriptor d is
 * block d.row0 + r).  Same result per row as the single-tensor call (mode, C_pad as there). */
typedef struct dlip_wsplit_desc {
  const float* w;       /* [K, C, T] reference layout */
  float* w_split;       /* mode 0: [K][T][C_pad]; mode 1: [C][T reversed][C_pad = K padded] */
  float* w_scale;       /* one per output row */
  int32_t K, C, T, C_pad, mode, row0;
} dlip_wsplit_desc;
/* (ABI 44) max_row_floats: the longest output row of the launch, T * C_pad floats (0: unknown) -- sizes the kernel's staging buffer
 * (dynamic LDS), so that a launch of short rows keeps more workgroups on a CU. */
int dlip_split_weights_multi_f32(const void* descs, const int32_t* block_desc, int32_t n_blocks, int32_t max_row_floats,
                                 dlip_stream_t stream);
/* y[0:n] = src[0] (device scalar broadcast: the per-channel 1/scale vector of the weight-gradient GEMM). */
int dlip_fill_from_scalar_f32(const float* src, float* y, int32_t n, dlip_stream_t stream);

/* Additive angular margin on cosine logits (ArcFace / AAM-softmax: named by the north star; `AAMSoftmax` is an
 * empty stub upstream, models/audio_models/loss.py:62-67 -- parity unpinned, the published recipe is restated):
 * backward == 0: y = logits with the target column cos(theta) replaced by cos(theta + margin) (where cos > cos(pi - m),
 * else cos - m sin(pi - m); easy_margin: only where cos > 0); backward != 0: y = g * d(modified)/d(logits).
 * logits, g, y [B,K]; labels int64 [B]. */
int dlip_aam_margin_f32(const float* logits, const int64_t* labels, const float* g, float* y, int32_t B, int32_t K,
                        float margin, int32_t easy_margin, int32_t backward, dlip_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Train-mode kernels of the lip-clip encoder (deeplip_amd/csrc/video_train_ops.hip):
 * what sits around the implicit-GEMM convolutions when torch.autograd's work for Lipreading.forward under
 * model.train() (train_video.py:108-169; models/video_models/model.py:80-105, resnet.py:28-127,
 * tcn.py:28-116) is done by dlip_* launches (deeplip_amd/autograd_video.py).  All tensors NHWC fp32,
 * C % 4 == 0.
 * ------------------------------------------------------------------------------------------ */

/* Rows one filter tap reads: out[j*ldo + 0:C] = x[n, ho*stride_h + off_h, wo*stride_w + off_w, 0:C] (zeros outside
 * the image), j = (n*Ho + ho)*Wo + wo; off = tap * dilation - padding.  With out = base + tap*C and
 * ldo = taps*C the taps land side by side in one [J, taps*C] matrix: the operand of the weight-gradient GEMM
 * of a padded / strided Conv2d / Conv1d (resnet.py:9-16, tcn.py:39-41). */
int dlip_tap_gather_f32(const float* x, float* out, int64_t N, int32_t H, int32_t W, int32_t C, int32_t ldx,
                        int32_t Ho, int32_t Wo, int32_t stride_h, int32_t stride_w, int32_t off_h, int32_t off_w,
                        int32_t ldo, dlip_stream_t stream);

/* One operand of the weight-gradient GEMM of a Conv2d / Conv1d in ONE pass (replaces dlip_tap_gather_f32 + dlip_nct_to_ntc_f32 +
 * dlip_split_pack*_f32 on a matrix R*S times the activation): out is [R*S*C rows][ld_out] floats in the split activation format ALONG
 * THE REDUCTION -- row (tap, c) holds scale * x[n, ho*stride_h + r*dil_h - pad_h, wo*stride_w + s*dil_w - pad_w, c] for j = (n*Ho +
 * ho)*Wo + wo = 0 .. J-1 (zeros outside the image and for j in J .. ld_out-1), 32 consecutive j per 128-byte block (32 hi halves |
 * 32 lo halves).  ld_out: row pitch in floats, a multiple of 32, >= J (an ODD number of 128-byte blocks keeps the rows of a slice
 * off one memory channel).  R = S = 1, no padding, scale = the power-of-two lift of dlip_pow2_scale_f32: the dy operand.
 * out 128-byte aligned; scale a device scalar or NULL (1).  x is [N,H,W,ldx] NHWC (Conv1d: H = 1). */
/* (round 4) The one-tap case with BOTH images of a [J, C] row matrix from ONE read: `out` as above (rows c, positions j along the
 * reduction) and `nhwc_split_out` [J, C] = the same values in the convolution kernels' split activation format (per row and 32
 * channels one 128-byte block) -- the forward convolution's operand of x, or (scale = the lift) the data gradient's of dy, which
 * dlip_split_pack*_f32 formed in a pass of its own.  C % 64 == 0; x 16-byte, both outputs 128-byte aligned. */
int dlip_wgrad_operand_split_f32(const float* x, float* out, int64_t ld_out, int64_t J, int32_t C, const float* scale,
                                 float* nhwc_split_out, dlip_stream_t stream);
int dlip_wgrad_operand_f32(const float* x, float* out, int64_t ld_out, int64_t N, int32_t H, int32_t W, int32_t C, int32_t ldx,
                           int32_t Ho, int32_t Wo, int32_t stride_h, int32_t stride_w, int32_t R, int32_t S, int32_t dil_h,
                           int32_t dil_w, int32_t pad_h, int32_t pad_w, const float* scale, dlip_stream_t stream);
/* Operand of a weight gradient RUN AS A CONVOLUTION: x [N,H,W,C] fp32 (row pitch ldx) -> out [C,H,W,N32] in the split format, the
 * images as the reduction's channels (N32 >= N, a multiple of 32; images beyond N are zeros), times *scale when scale != NULL
 * (the gradient's power-of-two lift).  With x' = this image of the layer input and g' = this image of dy,
 *   dW[c, r, s, k] = dlip_conv_nhwc_f16x3(x' as N = C images of H x W x N32, filter g' = [K, Ho, Wo, N32], stride = the layer's
 *                    dilation, dilation = the layer's stride, padding = the layer's) [c, r, s, k]        (r < R, s < S)
 * -- ONE copy of each tensor instead of the R*S shifted copies of dlip_wgrad_operand_f32; dlip_conv_nhwc_f16x3 accepts filters of
 * more than 32 taps for this (split input, fp32 output, no residual).  train_video.py:129-147 (loss.backward()). */
int dlip_wgrad_chwn_f32(const float* x, float* out, int64_t N, int32_t H, int32_t W, int32_t C, int32_t ldx, int32_t N32,
                        const float* scale, int32_t slice_major, float* nhwc_split_out, dlip_stream_t stream);
/* (ABI 44) The two producers above with a train-mode BatchNorm + (Leaky | P)ReLU applied ON LOAD: x is the RAW output z of the
 * previous convolution and every loaded value becomes act((z - mean) invstd gamma + beta) (slope_vec [C] per-channel slopes, or NULL:
 * the scalar `slope`) before it is split -- the convolution behind a conv -> BatchNorm -> activation (tdnn.py:35-43: the next
 * TDNN_Block; resnet.py:51-57: conv2 behind bn1 + relu1) needs its input only as these images, so the activated tensor is never
 * stored (one write and one read of an activation-sized tensor per layer).  Slice-major image layout; C % 64 == 0; the split NHWC
 * copy is mandatory (it is the forward convolution's operand). */
int dlip_wgrad_operand_split_bn_f32(const float* x, float* out, int64_t ld_out, int64_t J, int32_t C, const float* mean,
                                    const float* invstd, const float* gamma, const float* beta, const float* slope_vec, float slope,
                                    float* nhwc_split_out, dlip_stream_t stream);
int dlip_wgrad_chwn_bn_f32(const float* x, float* out, int64_t N, int32_t H, int32_t W, int32_t C, int32_t N32, const float* mean,
                           const float* invstd, const float* gamma, const float* beta, const float* slope_vec, float slope,
                           float* nhwc_split_out, dlip_stream_t stream);
/* (ABI 47) The BACKWARD of a train-mode BatchNorm (+ LeakyReLU) applied ON LOAD, in two parts: (1) dlip_bn_rows_train_bwd_sums_f32 = the first
 * half of dlip_bn_rows_train_bwd_f32 -- dgamma, dbeta -- plus the power-of-two lift of a dx that is never written: from the largest |g| and
 * |xhat| the sums pass sees (amax_parts: 2 * ceil(C / 64) * dlip_bn_rows_chunks(M) floats of scratch) and max gamma invstd, the bound
 * |dx| <= gamma invstd max|g| (2 + max|xhat|) is put at 1024 (a lift is exact: another exponent, the same gradients); (2) the two operand
 * producers read dy and z and form dx = gamma invstd (g - dbeta / M - xhat dgamma / M) per loaded value (the apply pass's expression, the
 * same bits), times lift[0], straight into the weight gradient's image and the data gradient's split operand (nhwc_split_out, nullable).
 * The convolution in front of a BatchNorm (tdnn.py:35-43 under loss.backward(), train_audio.py:189-191) needs the BatchNorm's input
 * gradient only as these images: one write and one read of an activation-sized fp32 tensor per layer less.  C % 64 == 0. */
/* (ABI 48) MeanStdPooling (pooling.py:24-26) and its backward on the RAW output z of the last TDNN convolution, the train-mode BatchNorm +
 * LeakyReLU in front applied per loaded value (tdnn.py:35-43 -> :96 under model.train()): the activated [B,T,1500] tensor -- 460 MB at
 * B = 256 -- is never stored.  dx of the backward is the gradient with respect to the ACTIVATED values (the BatchNorm's backward follows). */
int dlip_meanstd_pool_bn_f32(const float* z, const float* mean, const float* invstd, const float* gamma, const float* beta, float slope,
                             float* y, int32_t B, int32_t T, int32_t C, dlip_stream_t stream);
int dlip_meanstd_pool_bwd_bn_f32(const float* z, const float* mean, const float* invstd, const float* gamma, const float* beta, float slope,
                                 const float* y, const float* dy, float* dx, int32_t B, int32_t T, int32_t C, dlip_stream_t stream);
/* (ABI 48) ... and the BatchNorm backward BEHIND that pooling with the pooling's backward formed on load: dy[b,t,c] = dmean / T + dstd (y - mean) /
 * ((T - 1) std) per loaded value from the pooled statistics y_pool [B,2C] and their gradient g_pool [B,2C] (y = the activated value the
 * BatchNorm backward recomputes anyway) -- the pooling's backward writes nothing (its [B,T,C] gradient was one write and two reads).  Otherwise
 * dlip_bn_rows_train_bwd_f32 (conv -> BatchNorm -> LeakyReLU order; M = B T > 4096 rows).  The gradient is A[b,c] + K[b,c] y: the two
 * coefficients per (utterance, channel) -- coef [B,2C] = (A | K) -- come from dlip_meanstd_bwd_coef_f32 (ABI 49), one launch over [B,2C]. */
int dlip_meanstd_bwd_coef_f32(const float* y_pool, const float* g_pool, float* coef, int32_t B, int32_t C, int32_t T, dlip_stream_t stream);
int dlip_bn_rows_train_bwd_ms_f32(const float* ms_coef, int32_t T, const float* x, const float* gamma, const float* beta, const float* save_mean,
                                  const float* save_invstd, float* dx, float* dgamma, float* dbeta, double* workspace, int32_t M, int32_t C,
                                  float slope, float* dx_lift2, dlip_stream_t stream);
int dlip_bn_rows_train_bwd_sums_f32(const float* dy, const float* x, const float* gamma, const float* beta, const float* save_mean,
                                    const float* save_invstd, float* dgamma, float* dbeta, double* workspace, float* amax_parts, int32_t M,
                                    int32_t C, float slope, int32_t act_first, float* dx_lift2, const float* ms_coef, int32_t ms_T,
                                    dlip_stream_t stream);
/* (ABI 49) both take ms_coef [B,2C] + ms_T (nullable: then as ABI 47): dy is itself formed on load from a MeanStdPooling's coefficients
 * (dlip_bn_rows_train_bwd_ms_f32's rule; dy may be NULL) -- the last TDNN layer's whole backward then reads z alone.  The operand
 * producer also takes a channel count that is no multiple of 64 (C % 4 == 0: the E-TDNN's 1 500) and the row pitch ld_nhwc of its split copy
 * (0 = C; else >= C, a multiple of 32, the padding channels written as zeros). */
int dlip_wgrad_operand_split_bnbwd_f32(const float* dy, const float* z, float* out, int64_t ld_out, int64_t J, int32_t C, const float* mean,
                                       const float* invstd, const float* gamma, const float* beta, const float* dgamma, const float* dbeta,
                                       int64_t M, float slope, int32_t act_first, const float* lift, float* nhwc_split_out, int32_t ld_nhwc,
                                       const float* ms_coef, int32_t ms_T, dlip_stream_t stream);
int dlip_wgrad_chwn_bnbwd_f32(const float* dy, const float* z, float* out, int64_t N, int32_t H, int32_t W, int32_t C, int32_t N32,
                              const float* mean, const float* invstd, const float* gamma, const float* beta, const float* dgamma,
                              const float* dbeta, int64_t M, float slope, int32_t act_first, const float* lift, float* nhwc_split_out,
                              dlip_stream_t stream);
/* nhwc_split_out (nullable; C % 32 == 0): the same tensor (scaled alike) ALSO in the convolution kernels' split activation format
 * [N,H,W,C] -- what dlip_split_pack_f32 / dlip_split_pack_scaled_f32 would write -- from the one read: the forward convolution's
 * operand together with the weight gradient's (x), the data gradient's together with the weight gradient's (dy). */
/* slice_major != 0: the image as [C][N32/32][H][W][32] instead -- one 32-image slice of all pixels of a channel is ONE contiguous
 * plane, so that the filter taps a convolution walks one after the other are adjacent 128-byte lines (pixel-major they are
 * N32 * 4 bytes apart: every piece a DRAM page of its own).  That is the layout dlip_wgrad_conv_f16x3 reads:
 *   dw [C, R', S', K] fp32 = sum over images and output positions, x_img = the layer input's image, g_img = the output gradient's
 *   (lifted), R' = (H + 2 pad - stride (Ho - 1) - 1) / dil + 1 >= R (dW sits in [:, :R, :S, :]); post_scale / post_shift [K]: the
 *   epilogue's affine (2^-e of the lift, 0); unit_scale [K]: ones (the "filter" carries no per-channel scale).
 * stride / pad / dil are the LAYER's. */
int dlip_wgrad_conv_f16x3(const float* x_img, const float* g_img, const float* post_scale, const float* post_shift,
                          const float* unit_scale, float* dw, int32_t C, int32_t H, int32_t W, int32_t K, int32_t Ho, int32_t Wo,
                          int32_t N32, int32_t stride_h, int32_t stride_w, int32_t pad_h, int32_t pad_w, int32_t dil_h, int32_t dil_w,
                          int32_t R, int32_t S, dlip_stream_t stream);
/* R, S (ABI 43; 0, 0 = dw [C, R', S', K] as above): the LAYER's filter extent -- dw is then written in the REFERENCE layout
 * [K, C, R, S] (what Conv2d.weight.grad is: models/video_models/resnet.py:9-16), the positions r' >= R / s' >= S dropped, by the
 * epilogue itself: no [C, R', S', K] tensor, no slice copy and no permute launch behind it (48 launches of a training step). */
/* The stem's input for its weight gradient run as a convolution: the clip x [B,T,H,W] -> out [5][H][W][N32] split format, n = b*T + t,
 * out[dt][h][w][n] = x[b, t + dt - 2, h, w] (zero outside the clip; N32 >= B*T, a multiple of 32): the five temporal taps of
 * Conv3d(1,64,(5,7,7),(1,2,2),(2,3,3)) (model.py:82) as five "images" of ONE 2-D convolution whose filter is the output gradient
 * (dlip_wgrad_chwn_f32 of dy: [64][H/2][W/2][N32]; stride 1, dilation 2, padding 3): dW[k, dt, r, s] = its output [dt, r, s, k]. */
int dlip_stem_wgrad_chwn_f32(const float* x, float* out, int32_t B, int32_t T, int32_t H, int32_t W, int32_t N32, int32_t slice_major,
                             dlip_stream_t stream);
/* out [N,Hu,Wu,C] = dz [N,Ho,Wo,C] with stride-1 zeros inserted (out[n, ho*s, wo*s] = dz[n, ho, wo]): the data
 * gradient of a strided convolution as a stride-1 convolution (resnet.py:9-16 with stride 2). */
int dlip_upsample_zero_f32(const float* dz, float* out, int64_t N, int32_t Ho, int32_t Wo, int32_t Hu, int32_t Wu,
                           int32_t C, int32_t stride_h, int32_t stride_w, dlip_stream_t stream);
/* (ABI 44) The same zero insertion written straight as the lifted split operand of the data-gradient convolution: out_split =
 * dlip_split_pack_scaled_f32(dlip_upsample_zero_f32(dz), scale) without the fp32 tensor in between (C % 32 == 0). */
int dlip_upsample_zero_split_f32(const float* dz, float* out_split, const float* scale, int64_t N, int32_t Ho, int32_t Wo, int32_t Hu,
                                 int32_t Wu, int32_t C, int32_t stride_h, int32_t stride_w, dlip_stream_t stream);
/* nn.PReLU(C) on rows [M,C] (resnet.py:52,66; model.py:84; tcn.py:47,105): y = x >= 0 ? x : slope[c] * x. */
int dlip_prelu_rows_fwd_f32(const float* x, const float* slope, float* y, int64_t M, int32_t C, dlip_stream_t stream);
/* Backward: dx, and dslope_terms [M,C] = (x < 0 ? dy * x : 0) whose column sums (dlip_colsum_rows_f32) are
 * the slope gradient. */
int dlip_prelu_rows_bwd_f32(const float* dy, const float* x, const float* slope, float* dx, float* dslope_terms,
                            int64_t M, int32_t C, dlip_stream_t stream);
/* The end of a residual block in one pass (resnet.py:66-68, tcn.py:114): sum = a + b (the pre-activation dlip_prelu_rows_bwd_f32
 * needs), y = prelu(sum) with per-channel slopes; a, b, sum, y [M,C]. */
int dlip_add_prelu_rows_fwd_f32(const float* a, const float* b, const float* slope, float* sum, float* y, int64_t M, int32_t C,
                                dlip_stream_t stream);
/* MaxPool3d((1,3,3),(1,2,2),(0,1,1)) backward (model.py:85): x [N,H,W,C] = the pooled tensor's input,
 * dy [N,Ho,Wo,C] -> dx; first-maximum tie rule (row-major window scan), deterministic. */
int dlip_maxpool3x3s2_bwd_f32(const float* x, const float* dy, float* dx, int64_t N, int32_t H, int32_t W, int32_t C,
                              dlip_stream_t stream);
/* The training pair that does not re-scan x: the forward also writes, per output element, the tap (r*3+s) of its first maximum
 * as one byte (idx: N*Ho*Wo*C bytes, 4 channels per 32-bit word), the backward reads at most four (byte, dy) pairs per input
 * pixel.  y is bit-identical to dlip_maxpool3x3s2_f32, dx to dlip_maxpool3x3s2_bwd_f32. */
int dlip_maxpool3x3s2_idx_f32(const float* x, float* y, uint32_t* idx, int64_t N, int32_t H, int32_t W, int32_t C, dlip_stream_t stream);
int dlip_maxpool3x3s2_bwd_idx_f32(const uint32_t* idx, const float* dy, float* dx, int64_t N, int32_t H, int32_t W, int32_t C,
                                  dlip_stream_t stream);
/* dx[n, p, :] = dy[n, :] * w, p < P: w = scale (AdaptiveAvgPool2d(1) backward, resnet.py:83: scale = 1/(H W)),
 * or with lengths != NULL w = (p < len[n] ? 1/len[n] : 0) (masked temporal mean backward, model.py:16-17). */
int dlip_row_broadcast_f32(const float* dy, const int32_t* lengths, float* dx, int64_t N, int32_t P, int32_t C,
                           float scale, dlip_stream_t stream);
/* im2col of the stem Conv3d(1,64,(5,7,7),(1,2,2),(2,3,3)) (model.py:82): x [B,T,H,W] -> col [B*T*(H/2)*(W/2), 248]
 * (245 taps + 3 zero columns), the operand of the stem's weight-gradient GEMM. */
int dlip_stem_im2col_f32(const float* x, float* col, int32_t B, int32_t T, int32_t H, int32_t W, dlip_stream_t stream);
/* The same operand WITHOUT the im2col round trip: out [248, ld_out] = the reduction-major split image the weight-gradient GEMM
 * reads (per row and 32 positions one 128-B block: 32 hi halves | 32 lo halves; positions >= J and rows 245..247 zero), written
 * in one pass from the clip x [B,T,H,W].  ld_out >= B*T*(H/2)*(W/2), a multiple of 32; out 128-B aligned.  Reports range like
 * the other split producers.  (train_video.py:129-147: the stem's weight gradient.) */
int dlip_stem_wgrad_operand_f32(const float* x, float* out, int64_t ld_out, int32_t B, int32_t T, int32_t H, int32_t W,
                                dlip_stream_t stream);
/* The stem's CURRENT weights w [K,245] (= [K,1,5,7,7]) -> the split-fp16 weight image of dlip_stem3d_bn_act_f16x3 /
 * dlip_stem3d_pool_f16x3 (K x 1184 B) and its per-channel power-of-two scale [K], on the device: what a training step needs
 * every iteration (the extraction path packs once on the host). */
int dlip_split_stem_weights_f32(const float* w, float* w_img, float* w_scale, int32_t K, dlip_stream_t stream);
/* (ABI 44) nn.Dropout (tcn.py:80,85) with the keep test in the kernel: y = u >= p ? x * scale : 0, u = the uniform draws (the
 * backward applies the same launch to dy with the kept u).  Replaces compare + cast + dlip_mul_mask_f32 behind the generator. */
int dlip_dropout_keep_f32(const float* x, const float* u, float* y, int64_t n, float p, float scale, dlip_stream_t stream);
/* (ABI 44) The end of a multibranch TCN stage in one launch: symmetric chomp (tcn.py:52-59) + concatenation along channels
 * (tcn.py:96-108) of n_branches <= 4 tensors z_j [B, lengths[j], widths[j]] (HOST arrays of pointers / sizes; lengths[j] - T even,
 * widths multiples of 4): cat[b, t, off_j + c] = z_j[b, t + (lengths[j] - T) / 2, c].  backward != 0: the reverse -- `branches` are
 * written from cat (= the gradient of the concatenation), zeros in the chomped rows. */
int dlip_chomp_concat_f32(const float* const* branches, const int32_t* lengths, const int32_t* widths, int32_t n_branches, float* cat,
                          int32_t B, int32_t T, int32_t backward, dlip_stream_t stream);
/* y = x * mask * scale (nn.Dropout forward / backward, tcn.py:80,85). */
int dlip_mul_mask_f32(const float* x, const float* mask, float* y, int64_t n, float scale, dlip_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Range status of the split-fp16 ("f16x3") arithmetic.  The reference computes in fp32 end to end
 * (models/video_models/model.py:82-85); the split activation format stores a value as hi + lo fp16, so a value
 * of magnitude >= 65520 becomes infinite.  Every kernel that PRODUCES split-format values (conv epilogues with
 * DLIP_SPLIT_OUT, dlip_split_pack*_f32, the fused stem + pool, pooling with out_split) stores 1 into its word of
 * a caller-owned status block when it meets such a value; the host reads the words (device memory after a
 * synchronisation, or host-pinned device-visible memory at any time), raises, and the documented recourse is to
 * re-pack that model in the exact "f32" mode (same engine).  words = int32[8] {conv, stem, split_pack, pooling, low side (below),
 * 3 reserved}, zeroed by the caller; NULL unregisters (nothing is reported).  One block per process (one process per GPU).
 * ------------------------------------------------------------------------------------------ */
int dlip_set_status_words(int32_t* words);
/* (ABI 45) A status block for the launches of the CALLING THREAD alone: until dlip_status_scope(NULL), every producer this thread
 * launches (and the verdict kernel of a range scope it closes) reports to `words` (int32[8], host-pinned and device-visible, or
 * device memory; zeroed by the caller) instead of the process-wide block.  The address is handed to the kernels per launch, so a step
 * plan recorded inside such a scope reports to ITS block on every replay: the host can tell WHICH recorded batch left the range
 * (deeplip_amd.pipeline: the offending batch is re-run on the exact "f32" pack of the same model, train_fusion.py:338-358 never
 * saw it) where the process-wide block only says that one of the launches since the last look did. */
int dlip_status_scope(int32_t* words);

/* The LOW side of that range.  hi keeps 11 significant bits down to 6.1e-5, but lo = v - hi is a normal fp16 number only while
 * |v| >= 2^-3; below, lo sits on the subnormal grid 2^-24: an absolute error of up to 3e-8 per element, whatever its size.  Measured
 * on a 3-layer chain of Gaussian activations the result is 5.9e-7 off at sigma 1, 2.8e-5 at sigma 1e-3 (largest element 4.5e-3),
 * 3.1e-2 at sigma 1e-6.  A checkpoint whose BatchNorm statistics put a whole layer's activations down there would lose
 * fp32-grade accuracy silently, so a produced tensor whose largest magnitude lies in (0, 2^-2) is reported -- word 4 of the
 * status block (int32[8]: {conv, stem, split_pack, pooling, LOW, 3 reserved}) receives the kernel family + 1 -- and the host
 * raises exactly as for an overflow (same recourse: "f32" packing).
 * Per-launch evidence needs a word per launch: between dlip_range_scope_begin and dlip_range_scope_end (thread-local, may
 * nest: the outermost pair counts) every producer launched by this thread takes the next 1024 words (32 cache lines: its waves
 * spread over them, so that thousands finishing together do not queue at one line) of `slots` (device int32[1024 n], zeroed once
 * by the caller; flags, not counters: plain stores, no read-modify-write); _end launches a one-block verdict kernel on `stream` -- which must be ordered behind every launch
 * of the scope (join side streams first) -- that reports and re-zeroes the words.  Recorded into a step plan the verdict is
 * part of every replay.  Outside a scope the low side is not guarded (the high side always is).  A scope with more producers than
 * slots ends with DLIP_ERANGE (the launches beyond n ran unguarded: never quietly).  Launches that split TWO tensors take two slots
 * (the stem + pool entry point: one for the clip its pre-pass splits, one for the pooled output). */
/* Span scope (measurement): what a replayed step plan cannot give the host -- no event recorded into a graph can be read back --
 * the kernel notes itself.  Between _begin and _end (thread-local, not nestable) every launch of an MFMA kernel on split-format
 * input (the LDS-DMA ring kernel behind dlip_conv_nhwc_f16x3 / dlip_conv2_nhwc_f16x3 / dlip_conv_pool_f16x3, the window kernel, the
 * rows kernel, the stem + pool kernel together with its pre-pass) takes the next RECORD of `pairs` (device uint64[16 n]: word 0 =
 * start, armed as ~0; words 8..15 = end, armed as 0, one per blockIdx.x % 8 so that a grid's workgroups do not queue at one word;
 * a record is one 128-byte line); its workgroups fold the constant 100 MHz clock (s_memrealtime) into it -- every 16th the minimum at
 * entry (a grid starts within a microsecond), EVERY one the maximum at exit.  _end launches a collect kernel on `stream` (order it
 * behind the scope's launches) that adds max(end) - start and 1 into acc[2 i], acc[2 i + 1] (device uint64[2 n], zeroed by the
 * caller) and re-arms the record -- recorded into a plan, every replay accumulates; *used = records taken.  Outside a scope the
 * kernels time nothing. */
int dlip_span_scope_begin(uint64_t* pairs, uint64_t* acc, int32_t n);
int dlip_span_scope_end(dlip_stream_t stream, int32_t* used);

int dlip_range_scope_begin(int32_t* slots, int32_t n);
int dlip_range_scope_end(dlip_stream_t stream);

/* Diagnostic overrides for tests and A/B runs (the launch path reads no environment variable):
 * key 0 tile of dlip_conv_nhwc_f32 / the register-staged f16x3 kernel, 1 tile of the LDS-DMA kernel,
 * 2 LDS-DMA kernel on/off (0 = off), 3 balanced split (0 never, 2 always), 4 window kernel on/off (0 = off),
 * 5 tile order of the LDS-DMA kernel (0 column block outer, 1 inner, 2 the column blocks as lanes of the balanced split: the blocks
 * of a row range on neighbouring workgroups at the same time -- built in for layer 3's two-block launches), 6 rows kernel (conv_rows_f16x3.hip: 0 = off, 1 = on for
 * every launch of its shape class whatever the size, 3 | 4 | 5 = on with that tile height in units of 32 rows),
 * 7 the rows kernel's general mode for 2-D filters / residual / second source (1 = on for every eligible launch; anything else:
 * off -- measured slower than the ring kernel on the trunk, so since ABI 43 it is compiled into the LAB library only
 * (deeplip_amd.build --lab): the product library returns DLIP_EINVAL for a value > 0); key 3 also takes 3 (experiment: slabs of same-XCD tiles through
 * that XCD's L2) and 4 (the in-kernel finisher where the built-in choice is the reduce launch);
 * 8 the train-mode BatchNorm / column-sum entry points' launch sequence (0 = ABI 43's: statistics, finalize, apply [, lift] as separate
 * launches; otherwise the finalize steps run in the last workgroup of the pass before them and few-row tensors take one launch);
 * 9 the rows kernel's short last round (0 = off: every tile the full height);
 * value -1 restores the built-in choice. */
int dlip_debug_set(int32_t key, int32_t value);

/* ------------------------------------------------------------------------------------------
 * Step plans.  The reference drives its encoders from a Python loop, one utterance and one torch.nn layer
 * at a time (train_fusion.py:338-358 extraction, :262-293 training step); here a step is ~45 short kernels
 * and the host must not sit between them.  A plan records every dlip_* launch made on `stream` between
 * dlip_plan_begin and dlip_plan_end (HIP stream capture, thread-local mode -> an instantiated hipGraph) and
 * dlip_plan_run replays them with ONE host call, asynchronously on the given stream.
 *   - `stream` of begin / end must be a created stream (the null stream cannot be captured);
 *   - between begin and end the caller makes launches only: no allocation, no synchronisation, no host copy
 *     (warm every kernel once on that stream first, so workspace registration and kernel attributes are set);
 *   - the plan addresses exactly the buffers of the recorded step: they stay caller-owned and must outlive it;
 *     new inputs are copied into the recorded input buffers;
 *   - a plan must not run concurrently with itself (replays on one stream are ordered).
 * dlip_plan_launches = kernel launches the plan holds.  dlip_plan_destroy(NULL) is a no-op.
 * ------------------------------------------------------------------------------------------ */
int dlip_plan_begin(dlip_stream_t stream);
int dlip_plan_end(dlip_stream_t stream, dlip_plan_t* plan);
int dlip_plan_run(dlip_plan_t plan, dlip_stream_t stream);
int dlip_plan_launches(dlip_plan_t plan);
int dlip_plan_destroy(dlip_plan_t plan);

#ifdef __cplusplus
}
#endif
#endif /* DEEPLIP_HIP_H */

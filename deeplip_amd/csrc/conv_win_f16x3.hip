// Split-fp16 implicit-GEMM convolution, WINDOW variant: same-size stride-1 R x S convolutions with few output
// channels (K <= 64: the four 3x3 convolutions of the trunk's layer 1, resnet.py:55-69 at 64 -> 64 channels --
// a quarter of the step's time on the ring kernel).
//
// Why a second kernel.  With BN = 64 the ring kernel (conv_igemm_f16x3_dma.hip) moves (128 + 64) x 128 B of operands
// per slice for 24 MFMAs per wave: the address unit (64 B / clk / CU) needs as many cycles per slice as the matrix
// core does, and two thirds of those bytes are the activation rows, which the nine taps of a 32-channel slice read
// NINE times shifted by whole pixels.  Here a workgroup loads, per 32-channel slice, ONE window -- the BM + (R-1) dil W
// + (S-1) dil consecutive input pixels its tile touches -- and every tap reads its fragments from that window at a row
// offset; only the weights go through a per-slice ring.  L2 -> LDS bytes per tile drop 2.3x (layer 1: 442 -> 191 KB).
//
// What round 1's window mode got wrong (and lost 3-10 % with): it stored a window as eight planes, one per 16-B
// chunk of the 128-B pixel row, so that fragment reads stay bank-conflict free at every row offset -- but an LDS-DMA
// piece then gathered one chunk of 64 different pixels = 64 cache-line requests per KiB.  The layout here keeps the
// conflict-free property with 4x fewer requests: rows are grouped in BLOCKS of 16; inside a block, 256-B bank row c
// holds chunk c of the 16 rows (slot = row & 15).  A fragment read (16 consecutive rows of one chunk, any offset)
// touches 16 distinct slots = all 64 banks once; a DMA piece fills 4 bank rows = 4 chunks (the hi or the lo half of
// the slice) x 16 rows, i.e. 64 contiguous bytes of each of 16 pixels.
//
//   LDS: [window slot 0][window slot 1][zero block][weight ring: 3 stages][epilogue table]; the epilogue image
//   overlays the window slots.  Taps that fall outside the image are redirected per lane to the zero block at the
//   slot with the same index (same banks: a group with masked lanes stays conflict free).
//   Window c+1 is fetched one piece per slice during the first taps of channel slice c (no burst, uniform vmcnt).
#include "conv_common.h"
#include "conv_dma_common.h"

#include <mutex>

namespace {

constexpr int WIN_SLACK = 48;   // rows a window holds beyond the tile's BM pixels: >= (R-1) dil W + (S-1) dil of the launch

template <int BM, int BN, int WAVES_M, int WAVES_N, bool OSPLIT, int OCC>
__global__ __launch_bounds__(64 * WAVES_M* WAVES_N, OCC) void conv_win_f16x3_kernel(const ConvArgs a) {
  constexpr int NW = WAVES_M * WAVES_N, NT = 64 * NW;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int FR = 16;
  constexpr int MI = WM / FR, NI = WN / FR;
  constexpr int RPP = NT / 8, B_PER = BN / RPP;
  constexpr int NSTAGE = 3, PF = NSTAGE - 1;
  constexpr int WBLK0 = (BM + WIN_SLACK + 15) / 16;
  constexpr int WBLK = (WBLK0 + NW / 2 - 1) / (NW / 2) * (NW / 2);   // 16-row blocks per window: 2 WBLK pieces, whole per wave
  constexpr int WPER = 2 * WBLK / NW;                                // window pieces per wave
  constexpr int SLOT_B = WBLK * 2048;
  constexpr int ZERO_OFF = 2 * SLOT_B;
  constexpr int BRING_OFF = ZERO_OFF + 2048;
  constexpr int BSTAGE_B = BN * ROWB;
  constexpr int TAB_OFF = BRING_OFF + NSTAGE * BSTAGE_B;
  constexpr int LDK = 32;
  static_assert(BN % RPP == 0 && WM % 16 == 0 && 2 * WBLK % NW == 0, "tile / window layout");
  static_assert(BM * BN * 4 <= 2 * SLOT_B, "the epilogue image overlays the window slots");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* lds_c = reinterpret_cast<char*>(smem);

  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);   // neighbours share an XCD (L2)
  const int tiles_m = (a.M + BM - 1) / BM;
  const int tile_n = tile / tiles_m;
  const int tile_m = tile - tile_n * tiles_m;
  const int m0 = tile_m * BM;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int lrow = lane & 15, half = lane >> 4;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const u32x4 xr = make_rsrc_words(a.x, a.x_bytes);
  const u32x4 wr = make_rsrc_words(a.w, a.w_bytes);
  const int ntaps = a.R * a.S;
  const int halo_lo = a.ph * a.W + a.pw;                                 // window row of output pixel m0 at tap (0, 0) is 0
  const int need_rows = BM + (a.R - 1) * a.dh * a.W + (a.S - 1) * a.dw;  // rows a window really holds (<= BM + WIN_SLACK)
  const int p0 = m0 - halo_lo;                                           // input pixel of window row 0 (may be negative)

  // ---- tap validity of this lane's MI fragment pixels (bit r*S + s), as the ring kernel's per-row gather mask ----
  uint32_t fr_mask[MI];
  {
    int hi0[MI], wi0[MI];
    uint32_t colbits[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int m = m0 + wm * WM + mi * 16 + lrow;
      const int mc = m < a.M ? m : a.M - 1;
      const int n = dlip_div(mc, a.div_howo);
      const int rem = mc - n * a.HoWo;
      const int ho = dlip_div(rem, a.div_wo);
      hi0[mi] = ho - a.ph;
      wi0[mi] = rem - ho * a.Wo - a.pw;
      colbits[mi] = 0u;
      fr_mask[mi] = 0u;
    }
    for (int sx = 0; sx < a.S; ++sx)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) colbits[mi] |= (uint32_t)((unsigned)(wi0[mi] + sx * a.dw) < (unsigned)a.W) << sx;
    for (int r = 0; r < a.R; ++r)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
        fr_mask[mi] |= ((unsigned)(hi0[mi] + r * a.dh) < (unsigned)a.H ? colbits[mi] : 0u) << (r * a.S);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
      if (m0 + wm * WM + mi * 16 + lrow >= a.M) fr_mask[mi] = 0u;
  }

  // ---- weight ring addressing (as the ring kernel: XOR swizzle on the source side) ----
  const int cq = tid & 7, rbase = tid >> 3;
  const int key_st = (rbase >> 1) & 7;
  const int csrc = ((cq ^ key_st) << 2);
  int b_off[B_PER];
#pragma unroll
  for (int j = 0; j < B_PER; ++j) {
    const int n = tile_n * BN + rbase + RPP * j;
    b_off[j] = n < a.K ? (n * a.rsc + csrc) * 4 : -1;
  }
  const uint32_t bpiece0 = lds0 + BRING_OFF + wave * 8 * ROWB;
  auto issue_b = [&](int stage, int w_tap) {
#pragma unroll
    for (int j = 0; j < B_PER; ++j)
      dma_piece(wr, b_off[j] >= 0 ? (uint32_t)(b_off[j] + w_tap) : DLIP_OOB_OFFSET, bpiece0 + stage * BSTAGE_B + j * RPP * ROWB);
  };
  // window piece q (= 2 block + half) of channel slice cc -> slot: 4 chunks (hi or lo half) x 16 rows
  auto issue_win = [&](int slot, int cc, int q) {
    const int wrow = (q >> 1) * 16 + lrow;
    const int p = p0 + wrow;
    const bool ok = wrow < need_rows && p >= 0 && p < a.M;
    dma_piece(xr, ok ? (uint32_t)((p * a.ldx + cc * BK) * 4 + ((q & 1) * 4 + half) * 16) : DLIP_OOB_OFFSET,
              lds0 + slot * SLOT_B + q * 1024);
  };

  // ---- prologue ----
#pragma unroll
  for (int j = 0; j < WPER; ++j) issue_win(0, 0, wave + NW * j);
  const int nk = a.nk;
  // the walk over slices: channel slice outer, tap inner; ks = slice whose weights are issued next
  int is_tap = 0, is_c0 = 0;
  auto w_tap_of = [&]() { return (is_tap * a.Cw + is_c0) * 4; };
  auto is_advance = [&]() { if (++is_tap == ntaps) { is_tap = 0; is_c0 += BK; } };
  issue_b(0, w_tap_of());
  if (nk > 1) { is_advance(); issue_b(1, w_tap_of()); }

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[mi][ni][e] = 0.f;
  if (tid < BN) {
    const int k = tile_n * BN + tid;
    const bool kok = k < a.K;
    float* tab = smem + TAB_OFF / 4;
    tab[tid] = kok ? 1.f / a.wscale[k] : 0.f;
    tab[BN + tid] = (kok && a.bias) ? a.bias[k] : 0.f;
    tab[2 * BN + tid] = (kok && a.slope) ? a.slope[k] : 1.f;
    tab[3 * BN + tid] = (kok && a.pscale) ? a.pscale[k] : 1.f;
    tab[4 * BN + tid] = (kok && a.pshift) ? a.pshift[k] : 0.f;
  }
  if (tid < 128) *reinterpret_cast<f32x4*>(lds_c + ZERO_OFF + tid * 16) = f32x4{0.f, 0.f, 0.f, 0.f};   // the zero block

  // the tap being multiplied: index, its window row offset, channel slice, window slot
  int ctap = 0, cs = 0, crow = 0, cc = 0;
  int a_ad[MI];
  const int a_lane = (wm * WM / 16) * 2048 + half * 256;    // this lane's chunk row of its first block
  const int z_lane = ZERO_OFF + half * 256;
  auto set_addr = [&]() {
    const int u = lrow + crow + cs * a.dw;
    const int a0 = (cc & 1) * SLOT_B + a_lane + (u >> 4) * 2048 + (u & 15) * 16;
    const int zad = z_lane + (u & 15) * 16;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) a_ad[mi] = ((fr_mask[mi] >> ctap) & 1u) ? a0 + mi * 2048 : zad;
  };
  auto tap_advance = [&]() {
    if (++cs == a.S) { cs = 0; crow += a.dh * a.W; }
    if (++ctap == ntaps) { ctap = 0; cs = 0; crow = 0; ++cc; }
  };

  const int b_frag = (wn * WN + lrow) * LDK;
  const int key_rd = (lrow >> 1) & 7;
  const int khi = (half ^ key_rd) << 2, klo = ((4 + half) ^ key_rd) << 2;
  f16x8 fal[MI], fah[MI], fbh[NI], fbl[NI];
  auto read_first = [&](int stage) {   // activation lo, weight hi
    const float* Bw = smem + (BRING_OFF + stage * BSTAGE_B) / 4 + b_frag;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) fal[mi] = *reinterpret_cast<const f16x8*>(lds_c + a_ad[mi] + 1024);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) fbh[ni] = *reinterpret_cast<const f16x8*>(Bw + ni * 16 * LDK + khi);
  };
  auto read_rest = [&](int stage) {    // activation hi, weight lo
    const float* Bw = smem + (BRING_OFF + stage * BSTAGE_B) / 4 + b_frag;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) fah[mi] = *reinterpret_cast<const f16x8*>(lds_c + a_ad[mi]);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) fbl[ni] = *reinterpret_cast<const f16x8*>(Bw + ni * 16 * LDK + klo);
  };
  auto mfma_p = [&](int grp, int m_lo, int m_hi) {   // grp 0: lo*hi, 1: hi*hi, 2: hi*lo
#pragma unroll
    for (int mi = m_lo; mi < m_hi; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const f16x8 av = grp == 0 ? fal[mi] : fah[mi];
        const f16x8 bv = grp == 2 ? fbl[ni] : fbh[ni];
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bv, av, acc[mi][ni], 0, 0, 0);
      }
  };
  constexpr int MH = MI / 2 > 0 ? MI / 2 : 1;
#define DLIP_FENCE() __builtin_amdgcn_sched_barrier(0)
  set_addr();
  // weights of slice 0 (and, older in the queue, the whole of window 0) have landed once only slice 1's pieces are outstanding
  if (nk > 1) wait_vmcnt<B_PER>(); else wait_vmcnt<0>();
  __syncthreads();   // (also publishes the table and the zero block)
  read_first(0);

  int st_cur = 0, st_iss = nk > 1 ? 2 % NSTAGE : 1 % NSTAGE;
  for (int kt = 0; kt < nk; ++kt) {
    const bool more1 = (kt + 1) < nk, moreP = (kt + PF) < nk;
    const int st_nxt = st_cur + 1 == NSTAGE ? 0 : st_cur + 1;
    // top of the slice, right behind the barrier: one piece of the NEXT channel slice's window (its slot was read last in
    // the previous channel slice), then the weights two slices ahead
    const bool win_now = ctap < WPER && (cc + 1) * BK < a.Cw;
    if (win_now) issue_win((cc + 1) & 1, cc + 1, wave + NW * ctap);
    if (moreP) { is_advance(); issue_b(st_iss, w_tap_of()); st_iss = st_iss + 1 == NSTAGE ? 0 : st_iss + 1; }
    DLIP_FENCE();
    read_rest(st_cur); DLIP_FENCE();
    mfma_p(0, 0, MI); DLIP_FENCE();
    tap_advance();
    set_addr();        // next tap's fragment addresses: plain VALU in the shadow of group 1
    DLIP_FENCE();
    mfma_p(1, 0, MI); DLIP_FENCE();
    mfma_p(2, 0, MH); DLIP_FENCE();
    if (more1) {
      // slice kt+1's weights must have landed; what was issued after them (this slice) stays in flight
      if (moreP) { if (win_now) wait_vmcnt<B_PER + 1>(); else wait_vmcnt<B_PER>(); }
      else       { if (win_now) wait_vmcnt<1>(); else wait_vmcnt<0>(); }
      __builtin_amdgcn_s_barrier();
      read_first(st_nxt);
    }
    DLIP_FENCE();
    if (MH < MI) mfma_p(2, MH, MI);
    DLIP_FENCE();
    st_cur = st_nxt;
  }
#undef DLIP_FENCE

  // ---- epilogue through LDS (the ring kernel's, fp32 / split rows): y = act(acc / wscale + bias + residual) * ps + pt ----
  {
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    constexpr int PITCH = BN * 4;
    constexpr int CPR = BN / 4;
    constexpr int RES_PIECES = BM * PITCH / 1024 / NW;
    constexpr int RPQ = 1024 / PITCH;
    static_assert((BM * PITCH) % (NW * 1024) == 0, "the tile image is a whole number of DMA pieces per wave");
    const u32x4 rrw = make_rsrc_words(a.res, a.res ? a.r_bytes : 0u);
    const __amdgpu_buffer_rsrc_t yr = dlip_make_rsrc(a.y, a.y_bytes);
    const int kcol0 = tile_n * BN;
    char* img = lds_c;
    const f32x4* tab = reinterpret_cast<const f32x4*>(smem + TAB_OFF / 4);
    const bool post = a.pscale != nullptr;
    float amax = 0.f;
    __syncthreads();   // every wave is done with the windows
    if (a.res) {
#pragma unroll
      for (int i = 0; i < RES_PIECES; ++i) {
        const int piece = i * NW + wave;
        const int r = piece * RPQ + lane / CPR;
        const int pp = lane % CPR;
        const int c = (pp & ~15) | ((pp ^ r) & 15);
        const int m = m0 + r;
        const bool ok = m < a.M && (kcol0 + (c >> 3) * 32) < a.K;
        dma_piece(rrw, ok ? (uint32_t)((m * a.ldr + kcol0) * 4 + c * 16) : DLIP_OOB_OFFSET, lds0 + piece * 1024);
      }
      wait_vmcnt<0>();
      __syncthreads();
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int kl = wn * WN + ni * FR + 4 * half;
      const f32x4 inv4 = tab[kl >> 2], bi4 = tab[(BN + kl) >> 2], sl4 = tab[(2 * BN + kl) >> 2];
      f32x4 ps4 = {1.f, 1.f, 1.f, 1.f}, pt4 = {0.f, 0.f, 0.f, 0.f};
      if (post) { ps4 = tab[(3 * BN + kl) >> 2]; pt4 = tab[(4 * BN + kl) >> 2]; }
      const int ch = (kl >> 5) * 8 + ((kl >> 3) & 3);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int r = wm * WM + mi * FR + lrow;
        char* row = img + r * PITCH + 2 * (kl & 4);
        const int phi = (ch & ~15) | ((ch ^ r) & 15), plo = ((ch + 4) & ~15) | (((ch + 4) ^ r) & 15);
        float v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = acc[mi][ni][c] * inv4[c] + bi4[c];
        if (a.res) {
          const h4 rh = *reinterpret_cast<const h4*>(row + phi * 16), rl = *reinterpret_cast<const h4*>(row + plo * 16);
#pragma unroll
          for (int c = 0; c < 4; ++c) v[c] += (float)rh[c] + (float)rl[c];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          v[c] = v[c] >= 0.f ? v[c] : v[c] * sl4[c];
          if (post) v[c] = v[c] * ps4[c] + pt4[c];
        }
        if constexpr (OSPLIT) {
          h4 hi, lo;
#pragma unroll
          for (int c = 0; c < 4; ++c) { hi[c] = (_Float16)v[c]; lo[c] = (_Float16)(v[c] - (float)hi[c]); }
          amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
          *reinterpret_cast<h4*>(row + phi * 16) = hi;
          *reinterpret_cast<h4*>(row + plo * 16) = lo;
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[mi][ni][c] = v[c];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (!OSPLIT) {
      if (a.res) __syncthreads();
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int kl = wn * WN + ni * FR + 4 * half;
        const int ch = kl >> 2;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const int r = wm * WM + mi * FR + lrow;
          const int pc = (ch & ~15) | ((ch ^ r) & 15);
          *reinterpret_cast<f32x4*>(img + r * PITCH + pc * 16) = acc[mi][ni];
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < BM * CPR / NT; ++i) {
      const int idx = i * NT + tid;
      const int r = idx / CPR, pp = idx % CPR;
      const int c = (pp & ~15) | ((pp ^ r) & 15);
      const int m = m0 + r;
      const int kfirst = OSPLIT ? kcol0 + (c >> 3) * 32 : kcol0 + c * 4;
      const u32x4 v = *reinterpret_cast<const u32x4*>(img + r * PITCH + pp * 16);
      const uint32_t off = (m < a.M && kfirst < a.K) ? (uint32_t)((m * a.ldy + kcol0) * 4 + c * 16) : DLIP_OOB_OFFSET;
      __builtin_amdgcn_raw_buffer_store_b128(v, yr, (int)off, 0, 0);
    }
    if constexpr (OSPLIT) dlip_report_range(amax, a.status);
  }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int OCC>
int launch_win(const ConvArgs& a, hipStream_t st, bool out_split) {
  ConvArgs b = a;
  const int tiles_m = (a.M + BM - 1) / BM;
  b.tiles_n = (a.K + BN - 1) / BN;
  const long long tiles = (long long)tiles_m * b.tiles_n;
  if (tiles <= 0 || tiles > 0x7FFFFFFFll) return DLIP_EINVAL;
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int WBLK0 = (BM + WIN_SLACK + 15) / 16;
  constexpr int WBLK = (WBLK0 + NW / 2 - 1) / (NW / 2) * (NW / 2);
  constexpr size_t lds = (size_t)2 * WBLK * 2048 + 2048 + (size_t)3 * BN * ROWB + 5 * BN * sizeof(float);
  static_assert(lds <= 160 * 1024, "LDS exceeds a CU");
  auto kern = out_split ? conv_win_f16x3_kernel<BM, BN, WAVES_M, WAVES_N, true, OCC> : conv_win_f16x3_kernel<BM, BN, WAVES_M, WAVES_N, false, OCC>;
  static std::mutex mu;
  static bool attr_set[2] = {false, false};
  if (lds > 64 * 1024) {
    std::lock_guard<std::mutex> lock(mu);
    if (!attr_set[out_split ? 1 : 0]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return (int)e;
      attr_set[out_split ? 1 : 0] = true;
    }
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(64 * NW), lds, st, b);
  return dlip_launch_status();
}

}  // namespace

// The window kernel serves same-size stride-1 convolutions with >= 9 taps (a wave's six window pieces go out one per
// slice and must be older in the queue than the weights of the next channel slice's first tap), a halo within WIN_SLACK,
// K <= 64 (one 64-channel column block: layer 1) and no second reduction source / pooled epilogue.
static bool win_shape_ok(int sh, int sw, int H, int W, int Ho, int Wo, int R, int S, int dh, int dw, int ph, int pw, int K) {
  if (dlip_dbg_value[DLIP_DBG_WIN] == 0) return false;
  return sh == 1 && sw == 1 && Wo == W && Ho == H && R * S >= 9 && K <= 64 && (K & 3) == 0 &&
         (R - 1) * dh * W + (S - 1) * dw <= WIN_SLACK && ph * W + pw <= WIN_SLACK;
}

extern "C" __attribute__((visibility("hidden"))) int dlip_conv_win_ok(const void* args) {
  const ConvArgs& a = *static_cast<const ConvArgs*>(args);
  return a.x2 == nullptr && a.pool == nullptr &&
         win_shape_ok(a.sh, a.sw, a.H, a.W, a.HoWo / a.Wo, a.Wo, a.R, a.S, a.dh, a.dw, a.ph, a.pw, a.K);
}

extern "C" int dlip_conv_dma_enabled(void);   // conv_igemm_f16x3.hip

// Which kernel a split-format launch of `d` goes to (include/deeplip_hip.h): 1 = this one.
extern "C" int dlip_conv_kernel_kind(const dlip_conv_desc* d) {
  if (!d) return DLIP_EINVAL;
  return (d->C % 32 == 0 && dlip_conv_dma_enabled() &&
          win_shape_ok(d->stride_h, d->stride_w, d->H, d->W, d->Ho, d->Wo, d->R, d->S, d->dil_h, d->dil_w, d->pad_h, d->pad_w, d->K))
             ? 1 : 0;
}

extern "C" __attribute__((visibility("hidden"))) int dlip_conv_f16x3_win_launch(const void* args, void* stream, int out_split) {
  const ConvArgs& a = *static_cast<const ConvArgs*>(args);
  return launch_win<128, 64, 2, 2, 2>(a, static_cast<hipStream_t>(stream), out_split != 0);
}

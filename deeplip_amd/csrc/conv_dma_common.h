// LDS-DMA helpers shared by the split-fp16 convolution kernels (conv_igemm_f16x3_dma.hip: operand ring;
// conv_win_f16x3.hip: activation windows).  Internal.
#pragma once
#include "conv_common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int ROWB = 128;  // bytes per LDS row (one 32-channel slice of one pixel / filter)

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Buffer descriptor as four SGPR dwords (what an asm operand can carry): raw buffer, stride 0,
// `bytes` records, the same DATA_FORMAT word as dlip_make_rsrc.
__device__ __forceinline__ u32x4 make_rsrc_words(const void* p, uint32_t bytes) {
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  u32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((uint32_t)v);
  r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32) & 0xFFFFu);
  r[2] = __builtin_amdgcn_readfirstlane(bytes);
  r[3] = 0x00020000u;
  return r;
}

// One LDS-DMA piece: 64 lanes x 16 B from (rsrc, voff) to LDS bytes [lds_base, lds_base + 1024).
// POL: cache policy of the load -- 0 default, 1 `sc1` (served by L2, does not allocate in this CU's L1), 2 `nt`.
template <int POL = 0>
__device__ __forceinline__ void dma_piece(const u32x4 rsrc, uint32_t voff, uint32_t lds_base) {
  const uint32_t base = __builtin_amdgcn_readfirstlane(lds_base);   // wave-uniform by construction; this pins it to an SGPR
  // m0 is reserved: hipcc keeps nothing in it across statements (the ISA dump shows no other m0 use)
  if constexpr (POL == 1)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen sc1 lds" : : "s"(base), "v"(voff), "s"(rsrc) : "memory");
  else if constexpr (POL == 2)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen nt lds" : : "s"(base), "v"(voff), "s"(rsrc) : "memory");
  else
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(base), "v"(voff), "s"(rsrc) : "memory");
}

}  // namespace

// Lab hooks of the LDS-DMA ring kernel (conv_igemm_f16x3_dma.hip).  The PRODUCT build (no -DDLIP_LAB) gets the empty column of
// this table; `python -m deeplip_amd.build --lab` (libdeeplip_hip_lab.so, used by tools/ through DLIP_LIB_PATH, never by the
// product) gets in-kernel stamps, the experimental tiles and loops of conv_dma_lab.inc and a stamped launch path.  The kernel's
// text itself carries no #ifdef: it names these hooks where an experiment attaches.
//
//   hook                         product            lab build
//   DLIP_LAB_STREAMK_FIELDS      (nothing)          unsigned long long* stamps
//   DLIP_STAMP / SSTAMP / BSTAMP (nothing)          s_memtime of thread 0 (wave 4 for BSTAMP) into sk.stamps
//   DLIP_LAB_WG_STAMP(slot, c)   (nothing)          s_memrealtime into slot `slot` under condition c
//   DLIP_LAB_WG_END_STAMPS()     (nothing)          the per-segment / per-workgroup closing stamps
//   conv_dma_hook_consts.inc     WIN = false        conv_dma_lab.inc section 1 (window-mode constants)
//   conv_dma_hook_window.inc     (nothing)          conv_dma_lab.inc section 2 (window-mode tile body)
//   conv_dma_hook_tile256.inc    (nothing)          conv_dma_lab.inc section 3 (256x256 half-column loop)
//   conv_dma_lab_menu.inc        not included       lab tile menu, stamped launch (conv_igemm_f16x3_dma.hip's dispatch)
#pragma once
#ifdef DLIP_LAB
#define DLIP_LAB_STREAMK_FIELDS unsigned long long* stamps;   /* [G][10] s_memtime values of each workgroup's first segment */
#define DLIP_STAMP(i) do { if (threadIdx.x == 0 && it == it_begin && sk.stamps) sk.stamps[(size_t)g * 10 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
// inside ONE slice (the 9th of each workgroup's first segment): [(G + g) * 10 + i]
#define DLIP_SSTAMP(i) do { if (threadIdx.x == 0 && it == it_begin && kt == 8 && sk.stamps) sk.stamps[((size_t)sk.G + g) * 10 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
// the same slice as seen by wave 4 (the ping-pong loop's half B): [(2 G + g) * 10 + i]
#define DLIP_BSTAMP(i) do { if (threadIdx.x == 256 && it == it_begin && kt == 8 && sk.stamps) sk.stamps[((size_t)2 * sk.G + g) * 10 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define DLIP_LAB_WG_STAMP(slot, cond) do { if (threadIdx.x == 0 && (cond) && sk.stamps) sk.stamps[(size_t)g * 10 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define DLIP_LAB_SEGMENT_END_STAMPS() do { \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
    if (threadIdx.x == 0 && it == it_begin && sk.stamps) { \
      sk.stamps[(size_t)g * 10 + 5] = __builtin_amdgcn_s_memtime(); \
      sk.stamps[(size_t)g * 10 + 6] = (unsigned long long)kn; \
      sk.stamps[(size_t)g * 10 + 7] = __builtin_amdgcn_s_memrealtime() - sk.stamps[(size_t)g * 10 + 7]; \
    } } while (0)
#define DLIP_LAB_WG_END_STAMPS() do { \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
    if (threadIdx.x == 0 && sk.stamps) { \
      sk.stamps[(size_t)g * 10 + 9] = __builtin_amdgcn_s_memrealtime(); \
      sk.stamps[(size_t)g * 10 + 6] |= (unsigned long long)(blockIdx.x & 7) << 32; \
    } } while (0)
// host side (launch_one / the tile menu / the dispatch switch of conv_igemm_f16x3_dma.hip)
#define DLIP_LAB_LAUNCH_HOOK() do { sk.stamps = nullptr; \
    if (getenv("DLIP_STAMP_PRINT")) { if (!getenv("DLIP_STAMP_REDUCE_LATER")) sk.reduce_later = 0;   /* (stamped launches skip the reduce launch) */ \
      return dlip_lab_stamped_launch(kern, (unsigned)G, threads, lds, st, b, sk, BM, BN); } } while (0)
// tiles 6.. of the lab menu: 6, 7 retired (the spread-piece variants of round 2); 8, 9: tiles 4, 2 with every wave issuing at the top
// of the slice; 10: the 256x256 experiment; 11: 256x128 with the LOCK-STEP loop (the product until round 3: the ping-pong loop's
// reference); 12: ping-pong without s_setprio around the matrix phase; 13: ping-pong with the group-major MFMA order; window mode
// of the 256x128 tile: dlip_debug_set(DLIP_DBG_WIN, 2) + a forced tile >= 14
#define DLIP_LAB_TILE_CFGS , {256, 128}, {256, 128}, {128, 64}, {64, 128}, {256, 256}, {256, 128}, {256, 128}, {256, 128}, {256, 128}
#define DLIP_LAB_DISPATCH_CASES \
    case 6: case 7: return DLIP_EINVAL; \
    case 8: return launch_dma<128, 64, 2, 2, 2, 3, 4>(a, st, epi); \
    case 9: return launch_dma<64, 128, 2, 2, 3, 2, 4>(a, st, epi); \
    case 10: return launch_dma<256, 256, 4, 2, 2, 1, 64>(a, st, epi); \
    case 11: return launch_dma<256, 128, 4, 2, 3, 1, 1024>(a, st, epi); \
    case 12: return launch_dma<256, 128, 4, 2, 3, 1, 512>(a, st, epi); \
    case 13: return launch_dma<256, 128, 4, 2, 3, 1, 2048>(a, st, epi); \
    case 14: if (dlip_dbg_value[DLIP_DBG_WIN] == 2 && win_mode_ok(a, epi)) return launch_dma<256, 128, 4, 2, 3, 1, 128>(a, st, epi); \
             return launch_dma<256, 128, 4, 2, 3, 1, 0>(a, st, epi);
#else
#define DLIP_LAB_STREAMK_FIELDS
#define DLIP_STAMP(i) do { } while (0)
#define DLIP_SSTAMP(i) do { } while (0)
#define DLIP_BSTAMP(i) do { } while (0)
#define DLIP_LAB_WG_STAMP(slot, cond) do { } while (0)
#define DLIP_LAB_SEGMENT_END_STAMPS() do { } while (0)
#define DLIP_LAB_WG_END_STAMPS() do { } while (0)
#define DLIP_LAB_LAUNCH_HOOK() do { } while (0)
#define DLIP_LAB_TILE_CFGS
#define DLIP_LAB_DISPATCH_CASES
#endif

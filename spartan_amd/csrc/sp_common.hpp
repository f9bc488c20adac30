// Shared host/device helpers for libspartan_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#ifndef __HIPCC_RTC__
#include <stdio.h>
#include <string.h>
#endif

#include "../../include/spartan_hip.h"

// ---- error plumbing ------------------------------------------------------
void sp_set_error(const char* fmt, ...);

#define SP_FAIL(...)          \
  do {                        \
    sp_set_error(__VA_ARGS__); \
    return 1;                 \
  } while (0)

#define SP_HIP(expr)                                                              \
  do {                                                                            \
    hipError_t e_ = (expr);                                                       \
    if (e_ != hipSuccess) {                                                       \
      sp_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr,             \
                   hipGetErrorString(e_));                                        \
      return 1;                                                                   \
    }                                                                             \
  } while (0)

#define SP_CHECK_LAUNCH()                                                         \
  do {                                                                            \
    hipError_t e_ = hipGetLastError();                                            \
    if (e_ != hipSuccess) {                                                       \
      sp_set_error("%s:%d: kernel launch failed: %s", __FILE__, __LINE__,         \
                   hipGetErrorString(e_));                                        \
      return 1;                                                                   \
    }                                                                             \
  } while (0)

static inline size_t sp_dtype_size(int32_t dt) {
  switch (dt) {
    case SP_F32: return 4;
    case SP_F64: return 8;
    case SP_I32: return 4;
    case SP_I64: return 8;
    case SP_BOOL: return 1;
    case SP_U8: return 1;
    default: return 0;
  }
}

// number of CUs on MI355X; grids for streaming kernels are capped at
// SP_CUS * SP_BLOCKS_PER_CU workgroups and grid-stride the rest.
#define SP_CUS 256
#define SP_BLOCKS_PER_CU 8
#define SP_BLOCK 256

// ---- direct-to-LDS loads (gfx950 global_load_lds_dwordx4): every lane fetches 16 B from its own global address
// and the wave's 1 KiB lands at a wave-uniform LDS base + 16 * lane.
#ifndef __HIPCC_RTC__
// An LDS-DMA is complete for OTHER waves only after the issuing wave has waited for it (vmcnt) and a barrier: hipcc
// does not reliably put that wait in front of a barrier inside a loop (seen missing in the PIPE loop's .s), so
// it is stated here.
#define SP_GLDS_LANDED() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
// `global_load_lds_dwordx4 v_off, s[base:base+1]`: a wave-uniform 64-bit base in SGPRs plus an unsigned 32-bit
// per-lane byte offset, LDS target (a wave-uniform byte address) through M0.  Written in asm because hipcc selects
// only the 64-bit-VGPR-address form for __builtin_amdgcn_global_load_lds -- one 64-bit VALU add per load and per
// k-tile, on ONE address register pair, plus a v_readfirstlane for M0, between the MFMAs (8192^3 GEMM: 141 vs 150
// TFLOP/s) -- whereas here nothing but scalar adds precede the load.  A kernel that uses these macros must not use
// the builtin as well (the compiler does not know M0 changed).
#define SP_GLDS_S(base, voff, lds_addr)                                                             \
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"                       \
               :: "s"(lds_addr), "v"(voff), "s"(base) : "memory")
// (64-bit per-lane address form, for gathers that do not fit a 32-bit offset; same M0 rule)
#define SP_GLDS_V(gptr, lds_addr)                                                                   \
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"                      \
               :: "s"(lds_addr), "v"(gptr) : "memory")
#define SP_LDS_ADDR(p) ((unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(p))
#endif

struct sp_inputs {
  const void* p[SP_MAX_INPUTS];
};

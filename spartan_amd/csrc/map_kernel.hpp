// The fused-map kernel template (device code only; also compiled at run time by
// sp_jit.hip for programs outside the prebuilt StaticProg library).
#pragma once
#include "sp_interp.hpp"

// One workgroup handles U x 256 vectors of V elements (16 B each): group u of a
// thread is 256 vectors after group u-1, so every load/store instruction of the
// workgroup is a fully coalesced 4 KiB.  Specialised programs are launched with a
// grid that covers the tile exactly (on MI355X a full grid streams ~25% faster than
// a capped, grid-striding one -- tools/hbm_probe.hip, profiles/); the interpreter
// launches a capped grid and this loop strides.
template <typename T, int V, int U, bool LINEAR, typename P = DynProg, int MASK = -1>
__global__ __launch_bounds__(SP_BLOCK) void sp_map_kernel(const sp_program p, const sp_inputs in,
                                                          void* __restrict__ out, int64_t start,
                                                          int64_t nvec) {
  const int64_t stride = (int64_t)gridDim.x * SP_BLOCK * U;
  for (int64_t i = (int64_t)blockIdx.x * SP_BLOCK * U + threadIdx.x; i < nvec; i += stride) {
    int64_t L[U];
    bool full = true;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t k = i + (int64_t)u * SP_BLOCK;
      L[u] = start + (k < nvec ? k : i) * V;   // tail groups re-evaluate group 0 (never stored)
      full = full && (k < nvec);
    }
    T res[U][V];
    if constexpr (MASK >= 0) {
      // specialised 2-D broadcast addressing (sp_eval_2d): one 32-bit division per lane
      const uint32_t cols = (uint32_t)p.shape[1];
      const uint32_t l32 = (uint32_t)L[0];
      const uint32_t row = l32 / cols;
      sp_eval_2d<T, V, P, MASK>(p, in, row, l32 - row * cols, L[0], res[0]);
    } else {
      sp_eval_u<T, V, U, LINEAR, P>(p, in, L, res);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (u == 0 || full || i + (int64_t)u * SP_BLOCK < nvec) sp_store_vec<T, V>(out, p.out_dtype, L[u], res[u]);
  }
}


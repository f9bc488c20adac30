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
//
// RAGGED (strided programs whose innermost dimension is not a multiple of V, e.g. a map over the
// slice x[1:, 1:]): the index space is walked as rows x ceil(inner / V) groups, so a group never
// crosses a row end; the last group of a row is evaluated element by element.  (U == 1 only.)
// NTM: see SP_STREAMS (sp_interp.hpp); the interpreter kernels (dispatch-bound) do without the hint.
template <typename T, int V, int U, bool LINEAR, typename P = DynProg, int MASK = -1, bool RAGGED = false, int NTM = 0>
__global__ __launch_bounds__(SP_BLOCK) void sp_map_kernel(const sp_program p, const sp_inputs in,
                                                          void* __restrict__ out, int64_t start,
                                                          int64_t nvec) {
  const int64_t stride = (int64_t)gridDim.x * SP_BLOCK * U;
  const sp_dyn dyn = sp_dyn_program<P, T>(p);
  if constexpr (RAGGED) {
    static_assert(U == 1 && !LINEAR, "ragged rows: strided programs, one group per lane");
    const int64_t inner = p.shape[p.ndim - 1];
    const int64_t gpr = (inner + V - 1) / V;   // groups per row
    for (int64_t i = (int64_t)blockIdx.x * SP_BLOCK + threadIdx.x; i < nvec; i += stride) {
      int64_t row, g;
      if (p.pad & SP_PAD_IDX32) {   // index space fits 32 bits
        const uint32_t r32 = (uint32_t)i / (uint32_t)gpr;
        row = r32;
        g = (uint32_t)i - r32 * (uint32_t)gpr;
      } else {
        row = i / gpr;
        g = i - row * gpr;
      }
      const int64_t col = g * V;
      const int64_t L0 = row * inner + col;
      if (col + V <= inner) {
        T res[1][V];
        if constexpr (MASK >= 0) {
          sp_eval_2d<T, V, P, MASK, NTM>(p, in, (uint32_t)row, (uint32_t)col, L0, res[0]);
        } else {
          const int64_t Ls[1] = {L0};
          sp_eval_u<T, V, 1, false, P, NTM>(p, in, Ls, res, nullptr, dyn);
        }
        if (SP_STREAMS(NTM, p)) sp_store_vec<T, V, true>(out, p.out_dtype, L0, res[0]);
        else sp_store_vec<T, V>(out, p.out_dtype, L0, res[0]);
      } else {
        for (int64_t c = col; c < inner; ++c) {
          T one[1][1];
          if constexpr (MASK >= 0) {
            sp_eval_2d<T, 1, P, MASK>(p, in, (uint32_t)row, (uint32_t)c, row * inner + c, one[0]);
          } else {
            const int64_t Ls[1] = {row * inner + c};
            sp_eval_u<T, 1, 1, false, P>(p, in, Ls, one, nullptr, dyn);
          }
          sp_store_vec<T, 1>(out, p.out_dtype, row * inner + c, one[0]);
        }
      }
    }
    return;
  }
  if constexpr (!P::kStatic && LINEAR && V == 4) {
    if (sp_ahead_applies<T, 2>(p) && nvec < (1LL << 31)) {
      // Interpreted dense fp32 programs of one or two operands, software-pipelined over the trips of a lane: the
      // operands of trip t + 1 are requested and the results of trip t - 1 stored between the moment trip t's operands
      // reach the register file and the moment its program runs (sp_ahead).
      sp_ahead<T, V, U, 2> ah;
      const uint32_t n32 = (uint32_t)nvec, step = (uint32_t)stride;
      uint32_t i = blockIdx.x * (SP_BLOCK * U) + threadIdx.x;
      if (i >= n32) return;
      auto place = [&](uint32_t at, int64_t (&Lo)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t k = at + (uint32_t)u * SP_BLOCK;
          Lo[u] = start + (int64_t)(k < n32 ? k : at) * V;   // tail groups re-evaluate group 0 (never stored)
        }
      };
      auto store = [&](uint32_t at, const int64_t (&Lo)[U], T (&val)[U][V]) {
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (u == 0 || at + (uint32_t)u * SP_BLOCK < n32) {
            if (SP_STREAMS(NTM, p)) sp_store_vec<T, V, true>(out, p.out_dtype, Lo[u], val[u]);
            else sp_store_vec<T, V>(out, p.out_dtype, Lo[u], val[u]);
          }
      };
      int64_t L[U];
      place(i, L);
      sp_fetch_ahead<T, V, U, NTM>(p, in, L, ah);
      T held[U][V];
      bool have_held = false;
      for (;;) {
        const bool more = n32 - i > step;            // (i < n32 here)
        T res[U][V];
        auto mid = [&]() {
          if (more) {
            int64_t Ln[U];
            place(i + step, Ln);
            sp_fetch_ahead<T, V, U, NTM>(p, in, Ln, ah);
          }
          if (have_held) {
            int64_t Lp[U];
            place(i - step, Lp);
            store(i - step, Lp, held);
          }
        };
        sp_eval_u<T, V, U, true, P, NTM>(p, in, L, res, nullptr, dyn, &ah, mid);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int v = 0; v < V; ++v) held[u][v] = res[u][v];
        have_held = true;
        if (!more) break;
        i += step;
        place(i, L);
      }
      store(i, L, held);
      return;
    }
  }
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
      sp_eval_2d<T, V, P, MASK, NTM>(p, in, row, l32 - row * cols, L[0], res[0]);
    } else {
      sp_eval_u<T, V, U, LINEAR, P, NTM>(p, in, L, res, nullptr, dyn);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (u == 0 || full || i + (int64_t)u * SP_BLOCK < nvec) {
        if (SP_STREAMS(NTM, p)) sp_store_vec<T, V, true>(out, p.out_dtype, L[u], res[u]);
        else sp_store_vec<T, V>(out, p.out_dtype, L[u], res[u]);
      }
  }
}


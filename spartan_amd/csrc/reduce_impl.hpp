// Fused map -> reduce over one axis of a tile, and single-pass arg-reduce.
//
// Replaces the per-tile local reduction of _reduce_mapper
// (reference spartan/expr/operator/reduce.py:21-70, e.g. data.sum(axis) in
// spartan/expr/mathematics.py:126-127, data.max/min in statistics.py:26-61,
// np.all/np.any in logic.py:25-46) including the ReduceMapFusion prologue
// (optimize.py:190-227), and the three-pass argmax/argmin of
// spartan/expr/sorting.py:67-123.
//
// HBM roofline kernels: every input element is read exactly once
// (algorithmic bytes = sizeof(in) per element).  The index space of
// the fused program is viewed as [outer, axis_len, inner]:
//   inner == 1 : "row" kernels  - lanes run along the reduced (contiguous) axis,
//                64-lane DPP/shuffle reduction, then an LDS stage across the 4
//                waves of the workgroup, then (if the axis was split over
//                several workgroups) a tiny second launch over the partials.
//   inner  > 1 : "column" kernels - lanes run along `inner` (coalesced 16 B per
//                lane), the 4 waves of a workgroup walk interleaved positions of
//                the reduced axis and are combined through LDS.
// Results are deterministic (no atomics; fixed combine order).
#pragma once
#include <float.h>
#include <math.h>

#include "sp_interp.hpp"

// Streaming policy of the evaluators inside the reduction kernels (SP_STREAMS, sp_interp.hpp): the specialised
// kernels follow the program's flag at run time, the interpreter kernels (dispatch-bound) do without the hint.
#define SP_RED_NTM(P) (P::kStatic ? 2 : 0)
// the software-pipelined walk of the interpreted column kernel (sp_reduce_cols_kernel): groups of rows per dispatch
// and operands held ahead (A/B builds: -DSP_RED_AHEAD_UP=.. -DSP_RED_AHEAD_N=..).  Measured on the 8192 x 65536 tile,
// GB/s of sum((x - 0.5)^2, 0) / sum(x, 0) / max(2x + 1, 0) on the interpreter (tools/interp_reduce_time.py):
//   no pipeline (round 4)  2196 / 3856 / 1709       UP 2, N 2 (174 VGPRs: 2 waves per SIMD)  1744 / 3856 / 1709
//   UP 2, N 1 (167: 3)     2229 / 5181 / 2261       UP 1, N 1   1810 / 4350 / 1879           UP 4, N 1 (256: 1)  1283 / 2515 / 1170
// -- one operand ahead at three waves per SIMD; programs of two operands take the plain walk.  What is left is the
// dispatch itself (the interpreted instructions of one trip cost more than its 16 bytes per lane take to arrive).
#ifndef SP_RED_AHEAD_UP
#define SP_RED_AHEAD_UP 2
#endif
#ifndef SP_RED_AHEAD_N
#define SP_RED_AHEAD_N 1
#endif

#ifndef __HIPCC_RTC__
int sp_validate_program(const sp_program* p);
int sp_static_enabled();
#include "sp_jit.hpp"
#endif

// ------------------------------------------------------------------ policies
template <typename T>
__device__ __forceinline__ T sp_red_identity(int op) {
  switch (op) {
    case SP_RED_SUM: return (T)0;
    case SP_RED_PROD: return (T)1;
    case SP_RED_MAX:
      if constexpr (sp_is_integral<T>::value) return (T)INT64_MIN;
      else return (T)(-INFINITY);
    case SP_RED_MIN:
      if constexpr (sp_is_integral<T>::value) return (T)INT64_MAX;
      else return (T)(INFINITY);
    case SP_RED_AND: return (T)1;
    default: return (T)0;
  }
}

template <typename T>
__device__ __forceinline__ T sp_red_combine(int op, T a, T b) {
  switch (op) {
    case SP_RED_SUM: return a + b;
    case SP_RED_PROD: return a * b;
    case SP_RED_MAX: return sp_nanmax<T>(a, b);
    case SP_RED_MIN: return sp_nanmin<T>(a, b);
    case SP_RED_AND: return (T)((a != (T)0) && (b != (T)0));
    default: return (T)((a != (T)0) || (b != (T)0));
  }
}

template <typename T>
__device__ __forceinline__ T sp_shfl_down(T v, int delta) {
  if constexpr (sizeof(T) == 8) {
    union { T t; int2 i; } u, w;
    u.t = v;
    w.i.x = __shfl_down(u.i.x, delta, 64);
    w.i.y = __shfl_down(u.i.y, delta, 64);
    return w.t;
  } else {
    return __shfl_down(v, delta, 64);
  }
}

// Plain reduction state.
// The combine op as a compile-time constant around a block of adds: ONE scalar switch per trip.  With `op` a run-time
// value inside Acc::add, the interpreted kernels (OP = -1) went through the switch once per ELEMENT -- 8 x (15 scalar
// instructions, a handful of branches, one dynamically indexed read of the result) per trip, which is what bound the
// interpreted column reduction (202 scalar instructions per trip of 512 elements; profiles/r05_notes.md section 10).
template <int K>
struct sp_op_c {
  static constexpr int value = K;
};
template <typename F>
__device__ __forceinline__ void sp_with_op(int op, F&& body) {
  switch (op) {
    case 1: body(sp_op_c<1>{}); break;
    case 2: body(sp_op_c<2>{}); break;
    case 3: body(sp_op_c<3>{}); break;
    case 4: body(sp_op_c<4>{}); break;
    case 5: body(sp_op_c<5>{}); break;
    default: body(sp_op_c<0>{}); break;
  }
}

template <typename T>
struct PlainAcc {
  T v;
  __device__ __forceinline__ void init(int op) { v = sp_red_identity<T>(op); }
  __device__ __forceinline__ void add(int op, T x, int64_t) { v = sp_red_combine<T>(op, v, x); }
  __device__ __forceinline__ void merge(int op, const PlainAcc& o) { v = sp_red_combine<T>(op, v, o.v); }
  __device__ __forceinline__ PlainAcc shfl(int d) const {
    PlainAcc r;
    r.v = sp_shfl_down<T>(v, d);
    return r;
  }
};

// (value, first index) state; op: 0 = argmax, 1 = argmin.
template <typename T>
struct ArgAcc {
  T v;
  int64_t i;
  __device__ __forceinline__ void init(int) {
    v = (T)0;
    i = -1;  // empty
  }
  static __device__ __forceinline__ bool better(int op, T bv, int64_t bi, T av, int64_t ai) {
    // is (bv,bi) strictly preferable to (av,ai)?
    if (ai < 0) return bi >= 0;
    if (bi < 0) return false;
    const bool an = sp_math<T>::isnan_(av), bn = sp_math<T>::isnan_(bv);
    if (an || bn) {
      if (an && bn) return bi < ai;
      return bn;  // NaN dominates (np.max / np.min propagate NaN)
    }
    if (op == 0 ? (bv > av) : (bv < av)) return true;
    return bv == av && bi < ai;
  }
  // Per-lane accumulation: every caller feeds one accumulator with INCREASING
  // positions `a`, so "first occurrence" only needs a strict value comparison
  // (the full (value, index) order is needed only when lanes are merged).
  // Branch-free (round 5; the short-circuit form compiled to an exec-mask region and a branch on `op` per element):
  // `!(x <= v)` holds when x is larger OR either is NaN, so with "v is not NaN" it is exactly "x is larger, or x is
  // the first NaN"; an empty accumulator (i < 0: the sign of the index's high word) takes anything.
  __device__ __forceinline__ void add(int op, T x, int64_t a) {
    const bool beats_max = !(x <= v), beats_min = !(x >= v);
    const bool beats = op == 0 ? beats_max : beats_min;
    const bool keeps = !sp_math<T>::isnan_(v);
    const bool take = ((int32_t)(i >> 32) < 0) | (beats & keeps);
    v = take ? x : v;
    i = take ? a : i;
  }
  __device__ __forceinline__ void merge(int op, const ArgAcc& o) {
    if (better(op, o.v, o.i, v, i)) { v = o.v; i = o.i; }
  }
  __device__ __forceinline__ ArgAcc shfl(int d) const {
    ArgAcc r;
    r.v = sp_shfl_down<T>(v, d);
    r.i = sp_shfl_down<int64_t>(i, d);
    return r;
  }
};

template <typename Acc>
__device__ __forceinline__ Acc sp_wave_reduce(int op, Acc a) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    Acc o = a.shfl(d);
    a.merge(op, o);
  }
  return a;
}

// where a finished accumulator goes
struct RedOut {
  void* out;         // final output (converted to out_dtype), used when nsplit == 1
  int32_t out_dtype;
  void* part_val;    // [..] of T, used when nsplit > 1
  int64_t* part_idx; // arg only
  int64_t* out_idx;  // arg only, final
  int64_t index_offset, nan_index;
};

template <typename T>
__device__ __forceinline__ void sp_emit(const RedOut& ro, bool final_, int64_t slot, const PlainAcc<T>& a) {
  if (final_) {
    T v = a.v;
    sp_store_vec<T, 1>(ro.out, ro.out_dtype, slot, &v);
  } else {
    ((T*)ro.part_val)[slot] = a.v;
  }
}
template <typename T>
__device__ __forceinline__ void sp_emit(const RedOut& ro, bool final_, int64_t slot, const ArgAcc<T>& a) {
  if (final_) {
    int64_t idx = a.i < 0 ? ro.nan_index : a.i + ro.index_offset;
    if (sp_math<T>::isnan_(a.v)) idx = ro.nan_index;
    ro.out_idx[slot] = idx;
    if (ro.out) ((T*)ro.out)[slot] = a.v;
  } else {
    ((T*)ro.part_val)[slot] = a.v;
    ro.part_idx[slot] = a.i;
  }
}

template <typename T>
__device__ __forceinline__ void sp_load_partial(const RedOut& ro, int64_t slot, PlainAcc<T>& a) {
  a.v = ((const T*)ro.part_val)[slot];
}
template <typename T>
__device__ __forceinline__ void sp_load_partial(const RedOut& ro, int64_t slot, ArgAcc<T>& a) {
  a.v = ((const T*)ro.part_val)[slot];
  a.i = ro.part_idx[slot];
}

// --------------------------------------------------------------- row kernels
// [O, A] with the reduced axis contiguous.  grid = (nsplit, rows).
// one element of the program at (row, col) / flat index L -- the scalar tails of the vector kernels
template <typename T, bool LINEAR, typename P, int MASK>
__device__ __forceinline__ T sp_eval_one(const sp_program& p, const sp_inputs& in, int64_t row, int64_t col, int64_t L,
                                         bool have_rc, const sp_dyn dyn) {
  T x[1][1];
  if constexpr (MASK >= 0) {
    sp_eval_2d<T, 1, P, MASK, SP_RED_NTM(P)>(p, in, (uint32_t)row, (uint32_t)col, L, x[0]);
  } else {
    const int64_t Ls[1] = {L};
    const int64_t rc[1][2] = {{row, col}};
    sp_eval_u<T, 1, 1, LINEAR, P, SP_RED_NTM(P)>(p, in, Ls, x, have_rc ? rc : nullptr, dyn);
  }
  return x[0][0];
}

template <typename T, int V, bool LINEAR, template <typename> class AccT, typename P = DynProg, int OP = -1, int MASK = -1>
__global__ __launch_bounds__(SP_BLOCK) void sp_reduce_rows_kernel(const sp_program p, const sp_inputs in,
                                                                 int op, int64_t O, int64_t A,
                                                                 int64_t chunk, int nsplit, RedOut ro) {
  using Acc = AccT<T>;
  __shared__ Acc sm[SP_BLOCK / 64];
  if constexpr (OP >= 0) op = OP;   // specialised kernels: the combine op is a constant too
  const sp_dyn dyn = sp_dyn_program<P, T>(p);
  const int s = blockIdx.x;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int64_t o = blockIdx.y; o < O; o += gridDim.y) {
    const int64_t a0 = (int64_t)s * chunk;
    int64_t a1 = a0 + chunk;
    if (a1 > A) a1 = A;
    Acc acc;
    acc.init(op);
    bool walked = false;
    if constexpr (!P::kStatic && LINEAR && V == 4 && sp_is_same<T, float>::value && MASK < 0) {
      // Interpreted dense fp32 programs: the walk along the row pipelined as in sp_reduce_cols_kernel (the operands of
      // trip t + 1 requested before trip t's program is dispatched; UP groups share a dispatch).  The plain loop below
      // issued 0.8 scalar instructions per ELEMENT (addressing and operand-type decisions of every trip) and was bound
      // by scalar issue, not by HBM or the vector ALU (profiles/r05_notes.md section 10).
      if (sp_ahead_applies<T, SP_RED_AHEAD_N>(p)) {
        walked = true;
        constexpr int UP = SP_RED_AHEAD_UP;
        constexpr int64_t step = (int64_t)SP_BLOCK * V;
        int64_t a = a0 + (int64_t)threadIdx.x * V;
        if (a + V <= a1) {
          sp_ahead<T, V, UP, SP_RED_AHEAD_N> ah;
          auto place = [&](int64_t at, int64_t (&Lo)[UP]) {
#pragma unroll
            for (int u = 0; u < UP; ++u) {
              const int64_t au = at + (int64_t)u * step;
              Lo[u] = o * A + (au + V <= a1 ? au : at);      // groups past the chunk re-read group `at` (not added)
            }
          };
          int64_t L[UP];
          place(a, L);
          sp_fetch_ahead<T, V, UP, 0>(p, in, L, ah);
          for (;;) {
            const bool more = a + step * UP + V <= a1;
            auto mid = [&]() {
              if (more) {
                int64_t Ln[UP];
                place(a + step * UP, Ln);
                sp_fetch_ahead<T, V, UP, 0>(p, in, Ln, ah);
              }
            };
            T x[UP][V];
            sp_eval_u<T, V, UP, true, P, 0>(p, in, L, x, nullptr, dyn, &ah, mid);
            sp_with_op(op, [&](auto k) {
#pragma unroll
              for (int u = 0; u < UP; ++u) {
                const int64_t au = a + (int64_t)u * step;
                if (u == 0 || au + V <= a1) {
#pragma unroll
                  for (int v = 0; v < V; ++v) acc.add(decltype(k)::value, x[u][v], au + v);
                }
              }
            });
            if (!more) break;
            a += step * UP;
            place(a, L);
          }
        }
      }
    }
    constexpr int U = 1;   // measured: more groups per lane do not help (profiles/r01_notes.md)
    for (int64_t a = a0 + (int64_t)threadIdx.x * V; !walked && a + V <= a1; a += (int64_t)SP_BLOCK * V * U) {
      int64_t L[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t au = a + (int64_t)u * SP_BLOCK * V;
        L[u] = o * A + (au < a1 ? au : a);
      }
      T x[U][V];
      if constexpr (MASK >= 0) {
        sp_eval_2d<T, V, P, MASK, SP_RED_NTM(P)>(p, in, (uint32_t)o, (uint32_t)a, L[0], x[0]);
      } else {
        const int64_t rc[1][2] = {{o, a}};   // (row, column) when the program space is [O, A]
        sp_eval_u<T, V, U, LINEAR, P, SP_RED_NTM(P)>(p, in, L, x, p.ndim == 2 && p.shape[1] == A ? rc : nullptr, dyn);
      }
      sp_with_op(op, [&](auto k) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int64_t au = a + (int64_t)u * SP_BLOCK * V;
          if (u == 0 || au < a1) {
#pragma unroll
            for (int v = 0; v < V; ++v) acc.add(decltype(k)::value, x[u][v], au + v);
          }
        }
      });
    }
    if constexpr (V > 1) {
      // the row's last (a1 - a0) % V elements (only when the row length is not a multiple of V)
      const int rem = (int)((a1 - a0) % V);
      if ((int)threadIdx.x < rem) {
        const int64_t a = a1 - rem + threadIdx.x;
        acc.add(op, sp_eval_one<T, LINEAR, P, MASK>(p, in, o, a, o * A + a, p.ndim == 2 && p.shape[1] == A, dyn), a);
      }
    }
    acc = sp_wave_reduce(op, acc);
    if (lane == 0) sm[w] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      Acc t = sm[0];
#pragma unroll
      for (int k = 1; k < SP_BLOCK / 64; ++k) t.merge(op, sm[k]);
      sp_emit<T>(ro, nsplit == 1, nsplit == 1 ? o : o * nsplit + s, t);
    }
    __syncthreads();
  }
}

// many short rows: one wave per row
template <typename T, int V, bool LINEAR, template <typename> class AccT, typename P = DynProg, int OP = -1, int MASK = -1>
__global__ __launch_bounds__(SP_BLOCK) void sp_reduce_rows_wave_kernel(const sp_program p,
                                                                      const sp_inputs in, int op,
                                                                      int64_t O, int64_t A, RedOut ro) {
  using Acc = AccT<T>;
  if constexpr (OP >= 0) op = OP;
  const sp_dyn dyn = sp_dyn_program<P, T>(p);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t wstride = (int64_t)gridDim.x * (SP_BLOCK / 64);
  for (int64_t o = (int64_t)blockIdx.x * (SP_BLOCK / 64) + w; o < O; o += wstride) {
    Acc acc;
    acc.init(op);
    constexpr int U = 1;
    for (int64_t a = (int64_t)lane * V; a + V <= A; a += 64 * V * U) {
      int64_t L[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t au = a + (int64_t)u * 64 * V;
        L[u] = o * A + (au < A ? au : a);
      }
      T x[U][V];
      if constexpr (MASK >= 0) {
        sp_eval_2d<T, V, P, MASK, SP_RED_NTM(P)>(p, in, (uint32_t)o, (uint32_t)a, L[0], x[0]);
      } else {
        const int64_t rc[1][2] = {{o, a}};
        sp_eval_u<T, V, U, LINEAR, P, SP_RED_NTM(P)>(p, in, L, x, p.ndim == 2 && p.shape[1] == A ? rc : nullptr, dyn);
      }
      sp_with_op(op, [&](auto k) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int64_t au = a + (int64_t)u * 64 * V;
          if (u == 0 || au < A) {
#pragma unroll
            for (int v = 0; v < V; ++v) acc.add(decltype(k)::value, x[u][v], au + v);
          }
        }
      });
    }
    if constexpr (V > 1) {
      const int rem = (int)(A % V);
      if (lane < rem) {
        const int64_t a = A - rem + lane;
        acc.add(op, sp_eval_one<T, LINEAR, P, MASK>(p, in, o, a, o * A + a, p.ndim == 2 && p.shape[1] == A, dyn), a);
      }
    }
    acc = sp_wave_reduce(op, acc);
    if (lane == 0) sp_emit<T>(ro, true, o, acc);
  }
}

// second stage over [O, nsplit] partials: one wave per row
template <typename T, template <typename> class AccT>
__global__ __launch_bounds__(SP_BLOCK) void sp_finish_rows_kernel(int op, int64_t O, int nsplit, RedOut ro) {
  using Acc = AccT<T>;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t wstride = (int64_t)gridDim.x * (SP_BLOCK / 64);
  for (int64_t o = (int64_t)blockIdx.x * (SP_BLOCK / 64) + w; o < O; o += wstride) {
    Acc acc;
    acc.init(op);
    // eight partials of a lane in flight (one at a time, the 2048 partials of a whole-tile reduction were 32 dependent
    // round trips: 19 us for the index reductions' finish, 6 % of argmax(axis=None) on a 2 GiB tile); merged in the
    // same order as before
    for (int k0 = lane; k0 < nsplit; k0 += 64 * 8) {
      Acc t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = k0 + u * 64;
        t[u].init(op);
        if (k < nsplit) sp_load_partial<T>(ro, o * nsplit + k, t[u]);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (k0 + u * 64 < nsplit) acc.merge(op, t[u]);
    }
    acc = sp_wave_reduce(op, acc);
    if (lane == 0) sp_emit<T>(ro, true, o, acc);
  }
}

// ------------------------------------------------------------ column kernels
// [O, A, I], lanes along I.  grid = (ceil(I / (64 V)), nsplit, O').
template <typename T, int V, bool LINEAR, template <typename> class AccT, typename P = DynProg, int OP = -1, int MASK = -1>
__global__ __launch_bounds__(SP_BLOCK) void sp_reduce_cols_kernel(const sp_program p, const sp_inputs in,
                                                                 int op, int64_t O, int64_t A, int64_t I,
                                                                 int64_t chunk, int nsplit, RedOut ro, int64_t c0) {
  using Acc = AccT<T>;
  constexpr int NW = SP_BLOCK / 64;
  __shared__ Acc sm[NW - 1][64 * V];
  if constexpr (OP >= 0) op = OP;
  const sp_dyn dyn = sp_dyn_program<P, T>(p);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int s = blockIdx.y;
  // columns c0 + ...: only whole groups of V (the I % V columns left over go to sp_reduce_cols_tail_kernel)
  const int64_t c = c0 + ((int64_t)blockIdx.x * 64 + lane) * V;
  const bool active = c + V <= I;
  for (int64_t o = blockIdx.z; o < O; o += gridDim.z) {
    const int64_t a0 = (int64_t)s * chunk;
    int64_t a1 = a0 + chunk;
    if (a1 > A) a1 = A;
    Acc acc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v].init(op);
    bool walked = false;
    if constexpr (!P::kStatic && LINEAR && V == 4 && sp_is_same<T, float>::value && MASK < 0) {
      // Interpreted dense fp32 programs of one or two operands: the walk down the axis is software-pipelined like
      // the interpreted map (sp_ahead, map_kernel.hpp) -- the operands of trip t + 1 are requested once trip t's sit
      // in the register file and before its program is dispatched, so HBM works while the wave interprets.  UP groups
      // of rows share a dispatch.
      if (sp_ahead_applies<T, SP_RED_AHEAD_N>(p)) {
        walked = true;
        constexpr int UP = SP_RED_AHEAD_UP;
        if (active && a0 + w < a1) {
          sp_ahead<T, V, UP, SP_RED_AHEAD_N> ah;
          auto place = [&](int64_t at, int64_t (&Lo)[UP]) {
#pragma unroll
            for (int u = 0; u < UP; ++u) {
              const int64_t au = at + (int64_t)u * NW;
              Lo[u] = (o * A + (au < a1 ? au : at)) * I + c;      // rows past the chunk re-read row `at` (not added)
            }
          };
          int64_t a = a0 + w;
          int64_t L[UP];
          place(a, L);
          sp_fetch_ahead<T, V, UP, 0>(p, in, L, ah);
          for (;;) {
            const bool more = a + NW * UP < a1;
            auto mid = [&]() {
              if (more) {
                int64_t Ln[UP];
                place(a + NW * UP, Ln);
                sp_fetch_ahead<T, V, UP, 0>(p, in, Ln, ah);
              }
            };
            T x[UP][V];
            sp_eval_u<T, V, UP, true, P, 0>(p, in, L, x, nullptr, dyn, &ah, mid);
            sp_with_op(op, [&](auto k) {
#pragma unroll
              for (int u = 0; u < UP; ++u) {
                const int64_t au = a + (int64_t)u * NW;
                if (u == 0 || au < a1) {
#pragma unroll
                  for (int v = 0; v < V; ++v) acc[v].add(decltype(k)::value, x[u][v], au);
                }
              }
            });
            if (!more) break;
            a += NW * UP;
            place(a, L);
          }
        }
      }
    }
    if (active && !walked) {
      constexpr int U = P::kStatic ? 1 : 2;   // interpreted: two groups share a dispatch (as in the map kernel)
      for (int64_t a = a0 + w; a < a1; a += NW * U) {
        int64_t L[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int64_t au = a + (int64_t)u * NW;
          L[u] = (o * A + (au < a1 ? au : a)) * I + c;
        }
        T x[U][V];
        if constexpr (MASK >= 0) {
          sp_eval_2d<T, V, P, MASK, SP_RED_NTM(P)>(p, in, (uint32_t)(o * A + a), (uint32_t)c, L[0], x[0]);
        } else {
          const int64_t rc[1][2] = {{o * A + a, c}};   // (row, column) when the program space is [O*A, I]
          sp_eval_u<T, V, U, LINEAR, P, SP_RED_NTM(P)>(p, in, L, x, p.ndim == 2 && p.shape[1] == I ? rc : nullptr, dyn);
        }
        sp_with_op(op, [&](auto k) {
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int64_t au = a + (int64_t)u * NW;
            if (u == 0 || au < a1) {
#pragma unroll
              for (int v = 0; v < V; ++v) acc[v].add(decltype(k)::value, x[u][v], au);
            }
          }
        });
      }
    }
    if (w > 0) {
#pragma unroll
      for (int v = 0; v < V; ++v) sm[w - 1][lane * V + v] = acc[v];
    }
    __syncthreads();
    if (w == 0 && active) {
#pragma unroll
      for (int v = 0; v < V; ++v) {
#pragma unroll
        for (int k = 0; k < NW - 1; ++k) acc[v].merge(op, sm[k][lane * V + v]);
        const int64_t slot = nsplit == 1 ? o * I + c + v : ((int64_t)s * O + o) * I + c + v;
        sp_emit<T>(ro, nsplit == 1, slot, acc[v]);
      }
    }
    __syncthreads();
  }
}

// the I % V (< 4) columns the vector launch leaves over: the workgroup's threads go down the ROWS of the
// chunk (thread t: column c0 + t % 4, rows t / 4, t / 4 + 64, ...), LDS combine in row order.
// grid = (1, nsplit, O'), same partial layout as sp_reduce_cols_kernel.
template <typename T, bool LINEAR, template <typename> class AccT>
__global__ __launch_bounds__(SP_BLOCK) void sp_reduce_cols_tail_kernel(const sp_program p, const sp_inputs in,
                                                                      int op, int64_t O, int64_t A, int64_t I,
                                                                      int64_t chunk, int nsplit, RedOut ro, int64_t c0) {
  using Acc = AccT<T>;
  constexpr int NR = SP_BLOCK / 4;
  __shared__ Acc sm[SP_BLOCK];
  const sp_dyn dyn = sp_dyn_program<DynProg, T>(p);
  const int col = threadIdx.x & 3, r = threadIdx.x >> 2;
  const int s = blockIdx.y;
  const int64_t c = c0 + col;
  const bool active = c < I;
  for (int64_t o = blockIdx.z; o < O; o += gridDim.z) {
    const int64_t a0 = (int64_t)s * chunk;
    int64_t a1 = a0 + chunk;
    if (a1 > A) a1 = A;
    Acc acc;
    acc.init(op);
    if (active) {
      for (int64_t a = a0 + r; a < a1; a += NR) {
        const int64_t L[1] = {(o * A + a) * I + c};
        const int64_t rc[1][2] = {{o * A + a, c}};
        T x[1][1];
        sp_eval_u<T, 1, 1, LINEAR, DynProg, 0>(p, in, L, x, p.ndim == 2 && p.shape[1] == I ? rc : nullptr, dyn);
        acc.add(op, x[0][0], a);
      }
    }
    sm[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < 4 && active) {
      for (int k = 1; k < NR; ++k) acc.merge(op, sm[k * 4 + col]);
      const int64_t slot = nsplit == 1 ? o * I + c : ((int64_t)s * O + o) * I + c;
      sp_emit<T>(ro, nsplit == 1, slot, acc);
    }
    __syncthreads();
  }
}

// second stage over [nsplit, E] partials (E = O*I): lanes along E (coalesced),
// the 4 waves of a workgroup interleave over the nsplit partials, LDS combine
template <typename T, template <typename> class AccT>
__global__ __launch_bounds__(SP_BLOCK) void sp_finish_cols_kernel(int op, int64_t E, int nsplit, RedOut ro) {
  using Acc = AccT<T>;
  constexpr int NW = SP_BLOCK / 64;
  __shared__ Acc sm[NW - 1][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int64_t e0 = (int64_t)blockIdx.x * 64; e0 < E; e0 += (int64_t)gridDim.x * 64) {
    const int64_t e = e0 + lane;
    Acc acc;
    acc.init(op);
    if (e < E) {
      for (int k = w; k < nsplit; k += NW) {
        Acc t;
        sp_load_partial<T>(ro, (int64_t)k * E + e, t);
        acc.merge(op, t);
      }
    }
    if (w > 0) sm[w - 1][lane] = acc;
    __syncthreads();
    if (w == 0 && e < E) {
#pragma unroll
      for (int k = 0; k < NW - 1; ++k) acc.merge(op, sm[k][lane]);
      sp_emit<T>(ro, true, e, acc);
    }
    __syncthreads();
  }
}

#ifndef __HIPCC_RTC__
// ------------------------------------------------------------------- planner
struct RedPlan {
  int kind;  // 0 rows-split, 1 rows-wave, 2 cols
  int nsplit;
  int64_t chunk;
  int64_t partial_slots;  // number of (val[,idx]) partial slots needed
};

static const int64_t kTargetBlocks = (int64_t)SP_CUS * SP_BLOCKS_PER_CU;

// workgroups the planner aims for when it splits the reduced axis (SP_RED_TARGET: tuning knob)
static int64_t sp_plan_target() {
  static int64_t v = -1;
  if (v < 0) {
    const char* e = getenv("SP_RED_TARGET");
    v = e ? atoll(e) : kTargetBlocks;
  }
  return v;
}

static RedPlan sp_plan(int V, int64_t O, int64_t A, int64_t I) {
  RedPlan pl;
  memset(&pl, 0, sizeof(pl));
  if (I == 1) {
    if (A <= 64 * V * 16 && O >= 1024) {
      pl.kind = 1;
      pl.nsplit = 1;
      pl.chunk = A;
      return pl;
    }
    pl.kind = 0;
    const int64_t unit = (int64_t)SP_BLOCK * V;  // one pass of the workgroup
    const int64_t min_chunk = unit * 8;
    int64_t want = (sp_plan_target() + O - 1) / O;       // splits needed to fill the chip
    int64_t maxs = (A + min_chunk - 1) / min_chunk;   // splits the row can afford
    int64_t ns = want < maxs ? want : maxs;
    if (ns < 1) ns = 1;
    if (ns > 4096) ns = 4096;
    int64_t chunk = (A + ns - 1) / ns;
    chunk = (chunk + unit - 1) / unit * unit;
    ns = (A + chunk - 1) / chunk;
    if (ns < 1) ns = 1;
    pl.nsplit = (int)ns;
    pl.chunk = chunk;
    pl.partial_slots = ns > 1 ? O * ns : 0;
    return pl;
  }
  pl.kind = 2;
  const int64_t bx = (I + 64 * V - 1) / (64 * V);
  const int64_t min_chunk = (SP_BLOCK / 64) * 16;
  int64_t want = (sp_plan_target() + bx * O - 1) / (bx * O);
  int64_t maxs = (A + min_chunk - 1) / min_chunk;
  int64_t ns = want < maxs ? want : maxs;
  if (ns < 1) ns = 1;
  if (ns > 1024) ns = 1024;
  int64_t chunk = (A + ns - 1) / ns;
  chunk = (chunk + 3) / 4 * 4;
  ns = (A + chunk - 1) / chunk;
  if (ns < 1) ns = 1;
  pl.nsplit = (int)ns;
  pl.chunk = chunk;
  pl.partial_slots = ns > 1 ? ns * O * I : 0;
  return pl;
}

// V-wide evaluation: a group of V consecutive elements along the kernel's fastest axis (A for the row
// kernels, I for the column kernels) must stay inside one row of the PROGRAM's index space, where
// broadcast / strided operands keep one address mode.  Linear programs address by the flat index alone.
// Row ends that are not a multiple of V are handled by the kernels' scalar tails; alignment is not
// required (element-aligned vector accesses, sp_interp.hpp).
template <int V>
static bool sp_reduce_can_vec(const sp_program* p, const void* const* in, int64_t A, int64_t I) {
  (void)in;
  if (V == 1 || p->linear) return true;
  const int64_t fast = (I == 1) ? A : I;
  const int64_t last = p->shape[p->ndim - 1];
  return last % V == 0 || last == fast;
}

static inline int cap_dim(int64_t x, int64_t cap) { return (int)(x < cap ? (x < 1 ? 1 : x) : cap); }

template <typename... Args>
static int sp_jit_go(void* fn, dim3 grid, hipStream_t st, Args... args) {
  void* ptrs[] = {(void*)&args...};
  return sp_jit_launch(fn, grid, dim3(SP_BLOCK), ptrs, st);
}

template <typename T, template <typename> class AccT>
static int sp_reduce_launch(const sp_program* p, const sp_inputs& in, const void* const* inp, int op,
                            int64_t O, int64_t A, int64_t I, RedOut ro, void* ws, size_t ws_bytes,
                            hipStream_t st) {
  constexpr int VV = sp_cls<T>::V;
  constexpr bool kArg = sizeof(AccT<T>) > sizeof(T);
  const bool vec = sp_reduce_can_vec<VV>(p, inp, A, I);
  const int V = vec ? VV : 1;
  const RedPlan pl = sp_plan(V, O, A, I);
  if (pl.partial_slots) {
    const size_t need = (size_t)pl.partial_slots * (sizeof(T) + (kArg ? sizeof(int64_t) : 0));
    if (!ws || ws_bytes < need) SP_FAIL("sp_reduce: workspace too small (%zu < %zu)", ws_bytes, need);
    ro.part_val = ws;
    ro.part_idx = kArg ? (int64_t*)((char*)ws + (size_t)pl.partial_slots * sizeof(T)) : nullptr;
  }
  const bool lin = p->linear != 0;
  // hot fp32 shapes: kernels specialised on a compile-time instruction stream
  // (sp_interp.hpp StaticProg) -- plain reductions of x, a*b (matrix.vector),
  // x*x, x*x+x, x*(yp-y); arg-reductions of x.
  int sid = -1;
  if constexpr (sp_is_same<T, float>::value) {
    if (vec && sp_static_enabled()) {
      sid = sp_find_static(p, -1);
      if (kArg ? (sid != 0) : !(sid == 0 || sid == 7 || sid == 9 || sid == 10 || sid == 11)) sid = -1;
    }
  }
  // 2-D specialised addressing: program space must be the kernel's own (row, column) space
  int mask2d = -1;
  if (sid >= 0 && !lin) {
    const int64_t cols = (I == 1) ? A : I;
    if (p->shape[1] == cols) mask2d = sp_mask_2d(p, p->n_inputs);
  }
  // programs outside the library: run-time specialisation for large tiles (sp_jit.hip)
  const bool jit_ok = vec && sid < 0 && sp_jit_enabled() && O * A * I >= sp_jit_min_elems();
  int jit_mask = -1;
  if (jit_ok && !lin && p->shape[1] == ((I == 1) ? A : I)) jit_mask = sp_mask_2d(p, p->n_inputs);
  // the combine op as a template constant: SUM / MAX / MIN of the plain reductions, and BOTH ops of the index
  // reductions (0 argmax, 1 argmin: with `op` a run-time value ArgAcc::add selects between two comparison results
  // per element, which the compiler does in vector registers)
  const int sop = kArg ? ((op == 0 || op == 1) ? op : -1)
                       : ((op == SP_RED_SUM || op == SP_RED_MAX || op == SP_RED_MIN) ? op : -1);
#define SP_GO(KERNEL, GRID, LIN, PROG, OPC, MSK, ...) \
  hipLaunchKernelGGL((KERNEL<T, VV, LIN, AccT, PROG, OPC, MSK>), GRID, dim3(SP_BLOCK), 0, st, __VA_ARGS__)
#define SP_LAUNCH_OP(KERNEL, GRID, LIN, PROG, MSK, ...)                                   \
  do {                                                                                    \
    if constexpr (kArg) {                                                                 \
      if (sop == 0) SP_GO(KERNEL, GRID, LIN, PROG, 0, MSK, __VA_ARGS__);                  \
      else if (sop == 1) SP_GO(KERNEL, GRID, LIN, PROG, 1, MSK, __VA_ARGS__);             \
      else SP_GO(KERNEL, GRID, LIN, PROG, -1, MSK, __VA_ARGS__);                          \
    }                                                                                     \
    else {                                                                                \
      if (sop == SP_RED_SUM) SP_GO(KERNEL, GRID, LIN, PROG, SP_RED_SUM, MSK, __VA_ARGS__); \
      else if (sop == SP_RED_MAX) SP_GO(KERNEL, GRID, LIN, PROG, SP_RED_MAX, MSK, __VA_ARGS__); \
      else if (sop == SP_RED_MIN) SP_GO(KERNEL, GRID, LIN, PROG, SP_RED_MIN, MSK, __VA_ARGS__); \
      else SP_GO(KERNEL, GRID, LIN, PROG, -1, MSK, __VA_ARGS__);                          \
    }                                                                                     \
  } while (0)
#define SP_LAUNCH_P(KERNEL, GRID, PROG, ...)                                              \
  do {                                                                                    \
    if (lin) SP_LAUNCH_OP(KERNEL, GRID, true, PROG, -1, __VA_ARGS__);                     \
    else SP_LAUNCH_OP(KERNEL, GRID, false, PROG, -1, __VA_ARGS__);                        \
  } while (0)
#define SP_LAUNCH(KERNEL, GRID, ...)                                                        \
  do {                                                                                      \
    bool done_ = false;                                                                     \
    if constexpr (sp_is_same<T, float>::value) {                                          \
      if (sid == 0) { SP_LAUNCH_P(KERNEL, GRID, StaticProg<0>, __VA_ARGS__); done_ = true; } \
      if constexpr (!kArg) {                                                                \
        if (sid == 7 && mask2d == 0) { SP_LAUNCH_OP(KERNEL, GRID, false, StaticProg<7>, 0, __VA_ARGS__); done_ = true; } \
        else if (sid == 7 && mask2d == 2) { SP_LAUNCH_OP(KERNEL, GRID, false, StaticProg<7>, 2, __VA_ARGS__); done_ = true; } \
        else if (sid == 7) { SP_LAUNCH_P(KERNEL, GRID, StaticProg<7>, __VA_ARGS__); done_ = true; } \
        if (sid == 9) { SP_LAUNCH_P(KERNEL, GRID, StaticProg<9>, __VA_ARGS__); done_ = true; } \
        if (sid == 10 && mask2d == 6) { SP_LAUNCH_OP(KERNEL, GRID, false, StaticProg<10>, 6, __VA_ARGS__); done_ = true; } \
        else if (sid == 10) { SP_LAUNCH_P(KERNEL, GRID, StaticProg<10>, __VA_ARGS__); done_ = true; } \
        if (sid == 11) { SP_LAUNCH_P(KERNEL, GRID, StaticProg<11>, __VA_ARGS__); done_ = true; } \
      }                                                                                     \
    }                                                                                       \
    if (!done_ && vec && jit_ok) {                                                          \
      char expr_[192];                                                                      \
      snprintf(expr_, sizeof(expr_), #KERNEL "<%s, %d, %s, %s, StaticProg<1000>, %d, %d>", sp_cls<T>::name(), VV, \
               lin ? "true" : "false", kArg ? "ArgAcc" : "PlainAcc", sop, jit_mask);        \
      void* fn_ = sp_jit_get("reduce_impl.hpp", expr_, p);                                  \
      if (fn_) {                                                                            \
        if (sp_jit_go(fn_, GRID, st, __VA_ARGS__)) return 1;                                \
        done_ = true;                                                                       \
      }                                                                                     \
    }                                                                                       \
    if (!done_) {                                                                           \
      if (vec) {                                                                            \
        if (lin) hipLaunchKernelGGL((KERNEL<T, VV, true, AccT>), GRID, dim3(SP_BLOCK), 0, st, __VA_ARGS__); \
        else hipLaunchKernelGGL((KERNEL<T, VV, false, AccT>), GRID, dim3(SP_BLOCK), 0, st, __VA_ARGS__);    \
      } else {                                                                              \
        if (lin) hipLaunchKernelGGL((KERNEL<T, 1, true, AccT>), GRID, dim3(SP_BLOCK), 0, st, __VA_ARGS__);  \
        else hipLaunchKernelGGL((KERNEL<T, 1, false, AccT>), GRID, dim3(SP_BLOCK), 0, st, __VA_ARGS__);     \
      }                                                                                     \
    }                                                                                       \
    SP_CHECK_LAUNCH();                                                                      \
  } while (0)

  if (pl.kind == 1) {
    const int64_t blocks = (O + (SP_BLOCK / 64) - 1) / (SP_BLOCK / 64);
    SP_LAUNCH(sp_reduce_rows_wave_kernel, dim3(cap_dim(blocks, kTargetBlocks * 4)), *p, in, op, O, A, ro);
  } else if (pl.kind == 0) {
    SP_LAUNCH(sp_reduce_rows_kernel, dim3(pl.nsplit, cap_dim(O, 65535)), *p, in, op, O, A, pl.chunk,
              pl.nsplit, ro);
    if (pl.nsplit > 1) {
      const int64_t blocks = (O + (SP_BLOCK / 64) - 1) / (SP_BLOCK / 64);
      hipLaunchKernelGGL((sp_finish_rows_kernel<T, AccT>), dim3(cap_dim(blocks, kTargetBlocks)),
                         dim3(SP_BLOCK), 0, st, op, O, pl.nsplit, ro);
      SP_CHECK_LAUNCH();
    }
  } else {
    const int64_t bx = (I + 64 * V - 1) / (64 * V);
    if (bx > 2147483647LL) SP_FAIL("sp_reduce: inner dimension too large");
    const int64_t i_tail = V > 1 ? I % V : 0;   // columns that do not fill a group of V
    if (I - i_tail > 0)
      SP_LAUNCH(sp_reduce_cols_kernel, dim3((unsigned)bx, pl.nsplit, cap_dim(O, 65535)), *p, in, op, O, A, I,
                pl.chunk, pl.nsplit, ro, (int64_t)0);
    if (i_tail) {
      // the left-over columns: the tail kernel on the same split plan, so its outputs / partials land in
      // the slots the finish kernel reads
      if (lin) hipLaunchKernelGGL((sp_reduce_cols_tail_kernel<T, true, AccT>), dim3(1, pl.nsplit, cap_dim(O, 65535)), dim3(SP_BLOCK), 0, st, *p, in, op, O, A, I, pl.chunk, pl.nsplit, ro, I - i_tail);
      else hipLaunchKernelGGL((sp_reduce_cols_tail_kernel<T, false, AccT>), dim3(1, pl.nsplit, cap_dim(O, 65535)), dim3(SP_BLOCK), 0, st, *p, in, op, O, A, I, pl.chunk, pl.nsplit, ro, I - i_tail);
      SP_CHECK_LAUNCH();
    }
    if (pl.nsplit > 1) {
      const int64_t E = O * I;
      const int64_t blocks = (E + 63) / 64;
      hipLaunchKernelGGL((sp_finish_cols_kernel<T, AccT>), dim3(cap_dim(blocks, kTargetBlocks)),
                         dim3(SP_BLOCK), 0, st, op, E, pl.nsplit, ro);
      SP_CHECK_LAUNCH();
    }
  }
#undef SP_LAUNCH
#undef SP_LAUNCH_P
#undef SP_LAUNCH_OP
#undef SP_GO
  return 0;
}

static size_t sp_ws_bytes(int32_t cls, int64_t O, int64_t A, int64_t I, bool arg) {
  const size_t ts = cls == SP_F32 ? 4 : 8;
  const int VV = cls == SP_F32 ? 4 : 2;
  // the vector / scalar decision is made at launch time: size for the larger plan
  size_t best = 0;
  const int vs[2] = {VV, 1};
  for (int k = 0; k < 2; ++k) {
    RedPlan pl = sp_plan(vs[k], O, A, I);
    size_t need = (size_t)pl.partial_slots * (ts + (arg ? 8 : 0));
    if (need > best) best = need;
  }
  return best + 256;
}

static int sp_check_space(const sp_program* p, int64_t O, int64_t A, int64_t I) {
  if (O < 1 || A < 1 || I < 1) SP_FAIL("sp_reduce: empty index space (outer=%lld axis=%lld inner=%lld)",
                                       (long long)O, (long long)A, (long long)I);
  int64_t n = 1;
  for (int d = 0; d < p->ndim; ++d) n *= p->shape[d];
  if (n != O * A * I) SP_FAIL("sp_reduce: outer*axis*inner=%lld != prod(shape)=%lld",
                              (long long)(O * A * I), (long long)n);
  return 0;
}


#endif  // !__HIPCC_RTC__

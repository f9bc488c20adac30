// sort / argsort along the last axis of a dense tile: the tile bodies of the reference's sort operators
//   spartan/expr/operator/sort.py   _sort_mapper (:68-69)  np.sort(tile, axis)
//                                   _argsort_mapper (:137-138) np.argsort(tile, axis)
//                                   _sample_sort_mapper (:11-25) / _fetch_sort_mapper (:43-65) np.sort(data, axis=None)
// The sort is STABLE (equal keys keep their order, np.argsort(kind='stable')); NaNs go last like NumPy's,
// -0.0 and +0.0 compare equal.  Values are gathered from the input by the sorted positions, so the output holds
// the input's bit patterns.
//   lines of <= 4096 32-bit elements : bitonic network on (key, column) pairs in LDS (packed as doubles: a
//                                      compare-exchange is v_min_f64 + v_max_f64), one HBM pass
//   lines of <= 4096 64-bit elements : the same network on (key64, column) pairs in two LDS arrays
//   lines of >= 2048 elements        : LSD radix sort, sizeof(T) passes over the key bytes, every line a SEGMENT of the
//                                      same launches (sp_radix.hpp: histogram [line][digit][block], one scan)
//   (SP_SORT_ALGO=radix, lines < 2048: the tile sorted as ONE array by key, then by the bytes of each element's ROW,
//    which brings the lines back together -- the fallback the LDS paths replaced, kept as a test knob)
#include <stdlib.h>

#include "sp_common.hpp"
#include "../../include/spartan_hip_extras.h"
#include "sp_radix.hpp"

namespace {

__device__ __forceinline__ uint32_t key32(float v) {
  if (v != v) return 0xFFFFFFFFu;
  uint32_t u = __float_as_uint(v);
  if (u == 0x80000000u) u = 0;   // -0.0 == +0.0
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ uint32_t key32(int32_t v) { return (uint32_t)v ^ 0x80000000u; }

// value back from its key; false when the key does not determine the bit pattern (NaN payloads, the sign of zero)
__device__ __forceinline__ bool unkey32(uint32_t k, float* v) {
  if (k == 0xFFFFFFFFu || k == 0x80000000u) return false;
  *v = __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
  return true;
}
__device__ __forceinline__ bool unkey32(uint32_t k, int32_t* v) {
  *v = (int32_t)(k ^ 0x80000000u);
  return true;
}

template <typename T>
__device__ __forceinline__ uint64_t key64(T v);
template <>
__device__ __forceinline__ uint64_t key64<float>(float v) { return key32(v); }
template <>
__device__ __forceinline__ uint64_t key64<int32_t>(int32_t v) { return key32(v); }
template <>
__device__ __forceinline__ uint64_t key64<double>(double v) {
  if (v != v) return ~0ull;
  uint64_t u = (uint64_t)__double_as_longlong(v);
  if (u == 0x8000000000000000ull) u = 0;
  return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}
template <>
__device__ __forceinline__ uint64_t key64<int64_t>(int64_t v) { return (uint64_t)v ^ 0x8000000000000000ull; }

// the value back from its 64-bit key (false: NaN payload / sign of zero are not in the key)
template <typename T>
__device__ __forceinline__ bool unkey64(uint64_t k, T* v);
template <>
__device__ __forceinline__ bool unkey64<float>(uint64_t k, float* v) { return unkey32((uint32_t)k, v); }
template <>
__device__ __forceinline__ bool unkey64<int32_t>(uint64_t k, int32_t* v) { return unkey32((uint32_t)k, v); }
template <>
__device__ __forceinline__ bool unkey64<double>(uint64_t k, double* v) {
  if (k == ~0ull || k == 0x8000000000000000ull) return false;
  *v = __longlong_as_double((long long)((k & 0x8000000000000000ull) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k));
  return true;
}
template <>
__device__ __forceinline__ bool unkey64<int64_t>(uint64_t k, int64_t* v) {
  *v = (int64_t)(k ^ 0x8000000000000000ull);
  return true;
}

// 32-bit element types sort 32-bit keys: 8-byte (key, position) pairs instead of 12
template <typename T>
struct KeyOf {
  typedef uint64_t type;
};
template <>
struct KeyOf<float> {
  typedef uint32_t type;
};
template <>
struct KeyOf<int32_t> {
  typedef uint32_t type;
};

template <typename T>
__global__ __launch_bounds__(256) void sp_sort_keys_kernel(const T* __restrict__ in, int64_t n,
                                                           typename KeyOf<T>::type* __restrict__ keys,
                                                           int32_t* __restrict__ idx) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    keys[i] = (typename KeyOf<T>::type)key64<T>(in[i]);
    idx[i] = (int32_t)i;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void sp_sort_emit_kernel(const T* __restrict__ in,
                                                           const typename KeyOf<T>::type* __restrict__ keys,
                                                           const int32_t* __restrict__ idx, int64_t n, uint32_t cols,
                                                           T* __restrict__ out_vals, int64_t* __restrict__ out_idx) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const uint32_t p = (uint32_t)idx[i];
    if (out_vals) {
      // the sorted key gives the value back without a random read of the input, except for NaNs and zeros
      T v = 0;
      const bool ok = unkey64<T>(keys[i], &v);
      if (__ballot(!ok) != 0) {        // wave-uniform: the gather stays off the common path
        if (!ok) v = in[p];
      }
      out_vals[i] = v;
    }
    if (out_idx) out_idx[i] = (int64_t)(p % cols);
  }
}

constexpr int LDS_SORT_E = 4096;   // (key, column) pairs per workgroup: 32 KB

// The (key, column) pair of an element as a DOUBLE in [1, 2): exponent 0x3FF, the 32-bit key in mantissa bits 51..20,
// the column in bits 19..7.  Doubles of one exponent order like their mantissas, so a compare-exchange is
// v_min_f64 + v_max_f64 -- two instructions instead of a 64-bit integer compare and four selects.  Padding slots hold
// the largest such value (column field 0x1FFF, above every real column).
__device__ __forceinline__ double sp_pack_pair(uint32_t key, uint32_t col) {
  return __longlong_as_double((long long)((0x3FFull << 52) | ((uint64_t)key << 20) | ((uint64_t)col << 7)));
}
__device__ __forceinline__ void sp_ce(double& lo, double& hi) {
  const double a = __builtin_fmin(lo, hi), b = __builtin_fmax(lo, hi);
  lo = a;
  hi = b;
}

// NB consecutive half-cleaner sub-stages (compare distances 2^(lowbit+NB-1) ... 2^lowbit, partner = i ^ distance, all
// ascending) on the 2^NB elements base + (m << lowbit), held in registers: one trip through LDS for NB sub-stages.
template <int NB, int E>
__device__ __forceinline__ void sp_bitonic_round(double* sm, int tid, int lowbit) {
  constexpr int G = 1 << NB;
  for (int gid = tid; gid < E / G; gid += 256) {
    const int base = ((gid >> lowbit) << (lowbit + NB)) | (gid & ((1 << lowbit) - 1));
    double r[G];
#pragma unroll
    for (int m = 0; m < G; ++m) r[m] = sm[base + (m << lowbit)];
#pragma unroll
    for (int b = NB - 1; b >= 0; --b) {
#pragma unroll
      for (int m = 0; m < G; ++m)
        if ((m & (1 << b)) == 0) sp_ce(r[m], r[m | (1 << b)]);
    }
#pragma unroll
    for (int m = 0; m < G; ++m) sm[base + (m << lowbit)] = r[m];
  }
}

// first sub-stage of level l (runs of 2^(l-1), both ascending, merged into runs of 2^l): element i against its MIRROR
// inside the run of 2^l, i ^ (2^l - 1).  With the mirror first, every later compare-exchange of the level is ascending
// (no direction flags).
template <int E>
__device__ __forceinline__ void sp_bitonic_mirror(double* sm, int tid, int l) {
  const int half = 1 << (l - 1), k = 1 << l;
  for (int t = tid; t < E / 2; t += 256) {
    const int i = ((t >> (l - 1)) << l) | (t & (half - 1));
    const int p = i ^ (k - 1);
    double a = sm[i], b = sm[p];
    sp_ce(a, b);
    sm[i] = a;
    sm[p] = b;
  }
}

// `npad` = cols rounded up to a power of two; a workgroup sorts E / npad rows at once (every compare distance is below
// npad, so rows never mix).  E = 4096 pairs (32 KB of LDS) for lines above 2048 elements, 2048 (16 KB: twice the
// workgroups per CU to hide the barriers) below.
template <typename T, int E>
__global__ __launch_bounds__(256) void sp_sort_rows_lds_kernel(const T* __restrict__ in, int64_t rows, int cols, int npad,
                                                               int log_npad, T* __restrict__ out_vals,
                                                               int64_t* __restrict__ out_idx) {
  __shared__ double sm[E];
  const int tid = threadIdx.x;
  const int rpw = E >> log_npad;
  const int64_t nblocks = (rows + rpw - 1) / rpw;
  for (int64_t rb = blockIdx.x; rb < nblocks; rb += gridDim.x) {
    for (int e = tid; e < E; e += 256) {
      const int c = e & (npad - 1);
      const int64_t r = rb * rpw + (e >> log_npad);
      double v = sp_pack_pair(0xFFFFFFFFu, 0x1FFFu);
      if (r < rows && c < cols) v = sp_pack_pair(key32(in[r * cols + c]), (uint32_t)c);
      sm[e] = v;
    }
    __syncthreads();
    for (int l = 1; l <= log_npad; ++l) {
      sp_bitonic_mirror<E>(sm, tid, l);
      __syncthreads();
      int jbit = l - 2;                    // remaining sub-stages: distances 2^(l-2) ... 1
      while (jbit >= 0) {
        const int nb = jbit + 1 < 4 ? jbit + 1 : 4;
        const int lowbit = jbit - nb + 1;
        switch (nb) {
          case 4: sp_bitonic_round<4, E>(sm, tid, lowbit); break;
          case 3: sp_bitonic_round<3, E>(sm, tid, lowbit); break;
          case 2: sp_bitonic_round<2, E>(sm, tid, lowbit); break;
          default: sp_bitonic_round<1, E>(sm, tid, lowbit); break;
        }
        __syncthreads();
        jbit -= nb;
      }
    }
    int special = 0;
    for (int e = tid; e < E; e += 256) {
      const int c = e & (npad - 1);
      const int64_t r = rb * rpw + (e >> log_npad);
      if (r < rows && c < cols) {
        const uint64_t bits = (uint64_t)__double_as_longlong(sm[e]);
        if (out_vals) {
          T v = 0;
          if (!unkey32((uint32_t)(bits >> 20), &v)) special = 1;
          out_vals[r * cols + c] = v;
        }
        if (out_idx) out_idx[r * cols + c] = (int64_t)((bits >> 7) & 0x1FFFu);
      }
    }
    // NaNs and zeros: the key does not hold their bits (payload, sign) -- a second, rarely taken pass gathers them
    // from the input (kept out of the loop above so that the common case does no gather at all)
    if (__syncthreads_or(special) && out_vals) {
      for (int e = tid; e < E; e += 256) {
        const int c = e & (npad - 1);
        const int64_t r = rb * rpw + (e >> log_npad);
        if (r < rows && c < cols) {
          const uint64_t bits = (uint64_t)__double_as_longlong(sm[e]);
          T v;
          if (!unkey32((uint32_t)(bits >> 20), &v)) out_vals[r * cols + c] = in[r * cols + ((bits >> 7) & 0x1FFFu)];
        }
      }
    }
    __syncthreads();
  }
}

// ---- 64-bit element types, lines of <= 4096: the same mirror-first network on (64-bit key, column) held in two LDS
// arrays (24 or 48 KB per workgroup); a pair is ordered by key, then column (stable).
constexpr int LDS_SORT_E64 = 4096;   // 2048 pairs (24 KB) for lines up to 2048, 4096 (48 KB) above

// compare-exchange of two (key, column) pairs kept in separate scalars (arrays of structs spill)
__device__ __forceinline__ void sp_ce64(uint64_t& ka, uint32_t& ca, uint64_t& kb, uint32_t& cb) {
  const bool sw = ka > kb || (ka == kb && ca > cb);
  const uint64_t k0 = sw ? kb : ka, k1 = sw ? ka : kb;
  const uint32_t c0 = sw ? cb : ca, c1 = sw ? ca : cb;
  ka = k0;
  kb = k1;
  ca = c0;
  cb = c1;
}

template <int NB, int E>
__device__ __forceinline__ void sp_bitonic_round64(uint64_t* sk, uint32_t* sc, int tid, int lowbit) {
  constexpr int G = 1 << NB;
  for (int gid = tid; gid < E / G; gid += 256) {
    const int base = ((gid >> lowbit) << (lowbit + NB)) | (gid & ((1 << lowbit) - 1));
    uint64_t rk[G];
    uint32_t rc[G];
#pragma unroll
    for (int m = 0; m < G; ++m) {
      rk[m] = sk[base + (m << lowbit)];
      rc[m] = sc[base + (m << lowbit)];
    }
#pragma unroll
    for (int b = NB - 1; b >= 0; --b) {
#pragma unroll
      for (int m = 0; m < G; ++m)
        if ((m & (1 << b)) == 0) sp_ce64(rk[m], rc[m], rk[m | (1 << b)], rc[m | (1 << b)]);
    }
#pragma unroll
    for (int m = 0; m < G; ++m) {
      sk[base + (m << lowbit)] = rk[m];
      sc[base + (m << lowbit)] = rc[m];
    }
  }
}

template <typename T, int E>
__global__ __launch_bounds__(256) void sp_sort_rows_lds64_kernel(const T* __restrict__ in, int64_t rows, int cols, int npad,
                                                                 int log_npad, T* __restrict__ out_vals,
                                                                 int64_t* __restrict__ out_idx) {
  __shared__ uint64_t sk[E];
  __shared__ uint32_t sc[E];
  const int tid = threadIdx.x;
  const int rpw = E >> log_npad;
  const int64_t nblocks = (rows + rpw - 1) / rpw;
  for (int64_t rb = blockIdx.x; rb < nblocks; rb += gridDim.x) {
    for (int e = tid; e < E; e += 256) {
      const int c = e & (npad - 1);
      const int64_t r = rb * rpw + (e >> log_npad);
      const bool real = r < rows && c < cols;
      sk[e] = real ? key64<T>(in[r * cols + c]) : ~0ull;
      sc[e] = real ? (uint32_t)c : 0xFFFFFFFFu;          // padding: above every real (key, column)
    }
    __syncthreads();
    for (int l = 1; l <= log_npad; ++l) {
      const int half = 1 << (l - 1), k = 1 << l;
      for (int t = tid; t < E / 2; t += 256) {   // mirror sub-stage
        const int i = ((t >> (l - 1)) << l) | (t & (half - 1));
        const int p = i ^ (k - 1);
        uint64_t ka = sk[i], kb = sk[p];
        uint32_t ca = sc[i], cb = sc[p];
        sp_ce64(ka, ca, kb, cb);
        sk[i] = ka;
        sc[i] = ca;
        sk[p] = kb;
        sc[p] = cb;
      }
      __syncthreads();
      int jbit = l - 2;
      while (jbit >= 0) {
        const int nb = jbit + 1 < 3 ? jbit + 1 : 3;
        const int lowbit = jbit - nb + 1;
        switch (nb) {
          case 3: sp_bitonic_round64<3, E>(sk, sc, tid, lowbit); break;
          case 2: sp_bitonic_round64<2, E>(sk, sc, tid, lowbit); break;
          default: sp_bitonic_round64<1, E>(sk, sc, tid, lowbit); break;
        }
        __syncthreads();
        jbit -= nb;
      }
    }
    for (int e = tid; e < E; e += 256) {
      const int c = e & (npad - 1);
      const int64_t r = rb * rpw + (e >> log_npad);
      if (r < rows && c < cols) {
        const uint32_t src = sc[e];
        if (out_vals) {
          T v = 0;
          const bool ok = unkey64<T>(sk[e], &v);
          if (__ballot(!ok) != 0) {
            if (!ok) v = in[r * cols + src];
          }
          out_vals[r * cols + c] = v;
        }
        if (out_idx) out_idx[r * cols + c] = (int64_t)src;
      }
    }
    __syncthreads();
  }
}

inline unsigned sort_grid(int64_t n, int per_block) {
  int64_t b = (n + per_block - 1) / per_block;
  const int64_t cap = (int64_t)SP_CUS * SP_BLOCKS_PER_CU * 4;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}

inline bool lds_path(int32_t dtype, int64_t cols) {
  const char* e = getenv("SP_SORT_ALGO");   // "radix" | "lds": test / tuning knob
  if (e && e[0] == 'r') return false;
  // (64-bit lines up to 4096: 50 Gkeys/s at 256-wide, 31 at 2048-wide -- against 8 for the radix paths.  The first
  //  version kept its pairs in an array of structs, which spilled to scratch: 10x slower.)
  if (dtype == SP_F64 || dtype == SP_I64) return cols <= LDS_SORT_E64;
  return (dtype == SP_F32 || dtype == SP_I32) && cols <= LDS_SORT_E;
}

template <typename T>
int sort_lds64(const T* in, int64_t rows, int64_t cols, T* out_vals, int64_t* out_idx, hipStream_t st) {
  int npad = 1, lg = 0;
  while (npad < cols) {
    npad <<= 1;
    ++lg;
  }
  if (npad <= LDS_SORT_E64 / 2)
    hipLaunchKernelGGL((sp_sort_rows_lds64_kernel<T, LDS_SORT_E64 / 2>), dim3(sort_grid(rows, LDS_SORT_E64 / 2 / npad)), dim3(256),
                       0, st, in, rows, (int)cols, npad, lg, out_vals, out_idx);
  else
    hipLaunchKernelGGL((sp_sort_rows_lds64_kernel<T, LDS_SORT_E64>), dim3(sort_grid(rows, 1)), dim3(256), 0, st, in, rows,
                       (int)cols, npad, lg, out_vals, out_idx);
  SP_CHECK_LAUNCH();
  return 0;
}

// segmented passes keep one histogram per (line, digit, key block of the line)
inline bool seg_ok(int64_t rows, int64_t cols) {
  const char* e = getenv("SP_SORT_SEGMENTED");     // "0": force the row passes (test knob)
  if (e && e[0] == '0') return false;
  // a segment occupies whole key blocks of 4096: lines below 2048 would leave most of every block empty (256-wide
  // fp64 lines: 120 ms segmented against ~36 ms with the row passes)
  return cols >= 2048 && sp_sort_blocks(rows * cols, cols) * RDX <= rows * cols;
}

template <typename T>
int sort_radix(const T* in, int64_t rows, int64_t cols, T* out_vals, int64_t* out_idx, void* d_ws, hipStream_t st) {
  const int64_t n = rows * cols;
  // lines long enough that per-line histograms stay small next to the data are sorted as SEGMENTS by the key passes
  // alone; otherwise the whole tile is sorted by key and two more passes over the row bytes bring the lines together
  const bool segmented = rows > 1 && seg_ok(rows, cols);
  SortWsT<typename KeyOf<T>::type> ws;
  sp_sort_ws_bytes(n, &ws, (char*)d_ws, segmented ? cols : 0);
  hipLaunchKernelGGL((sp_sort_keys_kernel<T>), dim3(sort_grid(n, 256)), dim3(256), 0, st, in, n, ws.keys[0], ws.idx[0]);
  SP_CHECK_LAUNCH();
  int cur = 0;
  for (int shift = 0; shift < (int)sizeof(T) * 8; shift += RDX_BITS) {
    if (sp_radix_pass(ws, cur, n, DigitOfKey{shift}, st, segmented ? cols : 0)) return 1;
    cur = 1 - cur;
  }
  int row_bits = 0;
  while (!segmented && row_bits < 32 && ((int64_t)1 << row_bits) < rows) ++row_bits;
  for (int shift = 0; shift < row_bits; shift += RDX_BITS) {
    if (sp_radix_pass(ws, cur, n, DigitOfRow{shift, (uint32_t)cols}, st)) return 1;
    cur = 1 - cur;
  }
  hipLaunchKernelGGL((sp_sort_emit_kernel<T>), dim3(sort_grid(n, 256)), dim3(256), 0, st, in, ws.keys[cur], ws.idx[cur],
                     n, (uint32_t)cols, out_vals, out_idx);
  SP_CHECK_LAUNCH();
  return 0;
}

template <typename T>
int sort_lds(const T* in, int64_t rows, int64_t cols, T* out_vals, int64_t* out_idx, hipStream_t st) {
  int npad = 1, lg = 0;
  while (npad < cols) {
    npad <<= 1;
    ++lg;
  }
  if (npad <= LDS_SORT_E / 2) {
    hipLaunchKernelGGL((sp_sort_rows_lds_kernel<T, LDS_SORT_E / 2>), dim3(sort_grid(rows, LDS_SORT_E / 2 / npad)), dim3(256), 0,
                       st, in, rows, (int)cols, npad, lg, out_vals, out_idx);
  } else {
    hipLaunchKernelGGL((sp_sort_rows_lds_kernel<T, LDS_SORT_E>), dim3(sort_grid(rows, LDS_SORT_E / npad)), dim3(256), 0, st,
                       in, rows, (int)cols, npad, lg, out_vals, out_idx);
  }
  SP_CHECK_LAUNCH();
  return 0;
}

}  // namespace

extern "C" size_t sp_sort_rows_workspace_bytes(int32_t dtype, int64_t rows, int64_t cols) {
  if (rows < 1 || cols < 1 || lds_path(dtype, cols)) return 256;
  // (sized for the wider key type; segmented and plain layouts differ only in the histogram, take the larger)
  const size_t a = sp_sort_ws_bytes<uint64_t>(rows * cols, nullptr, nullptr, 0);
  const size_t b = (rows > 1 && seg_ok(rows, cols)) ? sp_sort_ws_bytes<uint64_t>(rows * cols, nullptr, nullptr, cols) : 0;
  return a > b ? a : b;
}

extern "C" int sp_sort_rows(const void* d_in, int32_t dtype, int64_t rows, int64_t cols, void* d_out_vals,
                            int64_t* d_out_idx, void* d_ws, size_t ws_bytes, void* stream) {
  if (dtype != SP_F32 && dtype != SP_F64 && dtype != SP_I32 && dtype != SP_I64)
    SP_FAIL("sp_sort_rows: dtype must be f32, f64, i32 or i64");
  if (rows < 0 || cols < 0) SP_FAIL("sp_sort_rows: bad sizes");
  if (rows == 0 || cols == 0) return 0;
  if (!d_in || (!d_out_vals && !d_out_idx)) SP_FAIL("sp_sort_rows: NULL pointer");
  if (d_out_vals == d_in) SP_FAIL("sp_sort_rows: the sort is out of place");
  if (rows * cols > 2147483647LL - SCAN_CHUNK) SP_FAIL("sp_sort_rows: more than 2^31 elements in one tile");
  hipStream_t st = (hipStream_t)stream;
  if (lds_path(dtype, cols)) {
    if (dtype == SP_F32) return sort_lds<float>((const float*)d_in, rows, cols, (float*)d_out_vals, d_out_idx, st);
    if (dtype == SP_I32) return sort_lds<int32_t>((const int32_t*)d_in, rows, cols, (int32_t*)d_out_vals, d_out_idx, st);
    if (dtype == SP_F64) return sort_lds64<double>((const double*)d_in, rows, cols, (double*)d_out_vals, d_out_idx, st);
    return sort_lds64<int64_t>((const int64_t*)d_in, rows, cols, (int64_t*)d_out_vals, d_out_idx, st);
  }
  if (!d_ws || ws_bytes < sp_sort_rows_workspace_bytes(dtype, rows, cols)) SP_FAIL("sp_sort_rows: workspace too small");
  switch (dtype) {
    case SP_F32: return sort_radix<float>((const float*)d_in, rows, cols, (float*)d_out_vals, d_out_idx, d_ws, st);
    case SP_F64: return sort_radix<double>((const double*)d_in, rows, cols, (double*)d_out_vals, d_out_idx, d_ws, st);
    case SP_I32: return sort_radix<int32_t>((const int32_t*)d_in, rows, cols, (int32_t*)d_out_vals, d_out_idx, d_ws, st);
    default: return sort_radix<int64_t>((const int64_t*)d_in, rows, cols, (int64_t*)d_out_vals, d_out_idx, d_ws, st);
  }
}

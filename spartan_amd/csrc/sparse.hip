// Sparse tiles in HBM: the per-tile bodies of the reference's sparse path
//   spartan/array/sparse.pyx   sparse_to_dense_update (:21-38), dot_coo_dense_unordered_map (:103-158),
//                              slice / multiple_slice / compute_sparse_update (:198-341)
//   spartan/array/tile.pyx     merge, sparse branches (:226-252, :283-295)
//   spartan/expr/dot.py        dot_map2_mapper / dot_outer_mapper on scipy.sparse tiles (:193-240)
// The reference keeps a tile as whatever scipy.sparse format the producing mapper chose (lil, coo, csr ...)
// and converts on every use.  Here a sparse tile has ONE device format -- canonical CSR: int64 indptr,
// int32 column indices ascending inside a row, no duplicate coordinates -- and every structural operation
// (upload of COO/CSR data, transpose, slicing by a box, region updates, A + B, the expansion step of
// sparse x sparse) is "edit a COO list, then sp_coo_to_csr": an LSD radix sort of (row, col) keys that is
// stable, so duplicate coordinates are added in list order, without floating-point atomics.
//   sp_csr_spmm     CSR x dense (the dot of the pagerank / netflix-style programs).  N = 1: 8 B per stored entry
//                   (value + column) + 16 B per row (indptr, y, x once) of algorithmic HBM traffic.
//   sp_csr_scatter  sparse -> dense update (assign / add / the reference's masked first-write rule).
#include <stdlib.h>

#include "sp_common.hpp"
#include "sp_scan.hpp"
#include "sp_radix.hpp"

namespace {

constexpr uint64_t KEY_DROPPED = ~0ull;  // entries with row < 0 are dropped: they sort to the end

// ---------------------------------------------------------------- COO -> keys
__global__ __launch_bounds__(256) void sp_coo_keys_kernel(const int32_t* __restrict__ rows,
                                                          const int32_t* __restrict__ cols, int64_t n,
                                                          int64_t ncols, uint64_t* __restrict__ keys,
                                                          int32_t* __restrict__ idx) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int32_t r = rows[i];
    keys[i] = r < 0 ? KEY_DROPPED : (uint64_t)r * (uint64_t)ncols + (uint64_t)cols[i];
    idx[i] = (int32_t)i;
  }
}

// ---------------------------------------------------------------- compress sorted keys to CSR
__device__ __forceinline__ bool sp_is_head(const uint64_t* keys, int64_t i) {
  const uint64_t k = keys[i];
  return k != KEY_DROPPED && (i == 0 || keys[i - 1] != k);
}

__global__ __launch_bounds__(256) void sp_coo_heads_kernel(const uint64_t* __restrict__ keys, int64_t n,
                                                           int* __restrict__ flags) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    flags[i] = sp_is_head(keys, i) ? 1 : 0;
}

// entries with equal coordinates are added in list order (the sort is stable)
template <typename T>
__global__ __launch_bounds__(256) void sp_coo_compress_kernel(const uint64_t* __restrict__ keys,
                                                              const int32_t* __restrict__ idx,
                                                              const T* __restrict__ vals, int64_t n,
                                                              const int* __restrict__ pos, int64_t ncols,
                                                              int32_t* __restrict__ indices, T* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    if (!sp_is_head(keys, i)) continue;
    const uint64_t k = keys[i];
    T s = vals[idx[i]];
    for (int64_t j = i + 1; j < n && keys[j] == k; ++j) s += vals[idx[j]];
    const int o = pos[i];
    indices[o] = (int32_t)(k % (uint64_t)ncols);
    out[o] = s;
  }
}

// indptr[r] = number of output entries whose key is below r * ncols
__global__ __launch_bounds__(256) void sp_csr_indptr_kernel(const uint64_t* __restrict__ keys, int64_t n,
                                                            const int* __restrict__ pos,
                                                            const int* __restrict__ total, int64_t nrows,
                                                            int64_t ncols, int64_t* __restrict__ indptr) {
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r <= nrows; r += (int64_t)gridDim.x * 256) {
    const uint64_t target = (uint64_t)r * (uint64_t)ncols;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (keys[mid] < target) lo = mid + 1;
      else hi = mid;
    }
    // the first key >= target is a head unless it is a dropped entry (then every later key is dropped too)
    indptr[r] = (lo < n && keys[lo] != KEY_DROPPED) ? pos[lo] : *total;
  }
}

// ---------------------------------------------------------------- CSR -> COO rows, box edits
__device__ __forceinline__ int64_t sp_row_of(const int64_t* __restrict__ indptr, int64_t nrows, int64_t t) {
  int64_t lo = 0, hi = nrows;  // largest r with indptr[r] <= t
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (indptr[mid] <= t) lo = mid;
    else hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(256) void sp_csr_rows_kernel(const int64_t* __restrict__ indptr, int64_t nrows,
                                                          int64_t nnz, int32_t* __restrict__ rows) {
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < nnz; t += (int64_t)gridDim.x * 256)
    rows[t] = (int32_t)sp_row_of(indptr, nrows, t);
}

__global__ __launch_bounds__(256) void sp_coo_box_kernel(int32_t* __restrict__ rows, int32_t* __restrict__ cols,
                                                         int64_t n, int64_t r0, int64_t r1, int64_t c0, int64_t c1,
                                                         int64_t dr, int64_t dc, int drop_inside) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t r = rows[i], c = cols[i];
    if (r < 0) continue;
    const bool inside = r >= r0 && r < r1 && c >= c0 && c < c1;
    if (inside == (drop_inside != 0)) {
      rows[i] = -1;
    } else {
      rows[i] = (int32_t)(r + dr);
      cols[i] = (int32_t)(c + dc);
    }
  }
}

// reshape of a COO list: linear position row * old_cols + col, minus `offset`, re-split by new_cols; entries that
// fall outside [0, new_rows * new_cols) are dropped (the Reshape view fetches a covering rectangle of its base)
__global__ __launch_bounds__(256) void sp_coo_reshape_kernel(int32_t* __restrict__ rows, int32_t* __restrict__ cols,
                                                             int64_t n, int64_t old_cols, int64_t offset,
                                                             int64_t new_rows, int64_t new_cols) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t r = rows[i];
    if (r < 0) continue;
    const int64_t lin = r * old_cols + cols[i] - offset;
    if (lin < 0 || lin >= new_rows * new_cols) {
      rows[i] = -1;
    } else {
      rows[i] = (int32_t)(lin / new_cols);
      cols[i] = (int32_t)(lin % new_cols);
    }
  }
}

// ---------------------------------------------------------------- CSR x dense
// N == 1: G lanes per row, strided over the row's entries, shuffle reduction (fixed order per G).
template <typename T, int G>
__global__ __launch_bounds__(256) void sp_csr_spmv_kernel(const int64_t* __restrict__ indptr,
                                                          const int32_t* __restrict__ indices,
                                                          const T* __restrict__ vals, const T* __restrict__ x,
                                                          int64_t ldx, T* __restrict__ y, int64_t ldy, int64_t m,
                                                          int accumulate) {
  const int g = threadIdx.x % G;
  const int64_t groups = (int64_t)gridDim.x * (256 / G);
  for (int64_t r = (int64_t)blockIdx.x * (256 / G) + threadIdx.x / G; r < m; r += groups) {
    const int64_t a = indptr[r], b = indptr[r + 1];
    T acc = 0;
    if (x) {
      for (int64_t j = a + g; j < b; j += G) acc += vals[j] * x[(int64_t)indices[j] * ldx];
    } else {
      for (int64_t j = a + g; j < b; j += G) acc += vals[j];
    }
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (g == 0) y[r * ldy] = accumulate ? y[r * ldy] + acc : acc;
  }
}

// N == 1, short rows: "stream" formulation.  A workgroup owns SPMV_CH consecutive STORED ENTRIES (not rows):
// it loads their (column, value) pairs fully coalesced -- 8 independent loads per thread in flight, no
// indptr -> entries -> x dependency chain per row -- gathers x, leaves the products in LDS, and then one
// thread per row adds the row's products in storage order.  Rows are assigned by where they START
// (indptr[r] in [e0, e1)), so empty rows are written too; the head of the chunk that belongs to a row
// started earlier goes to carry[b] and a fix-up pass adds the carries of a row in chunk order
// (deterministic, no floating-point atomics).  Chunks are dealt to the 8 XCDs in contiguous ranges so a
// site-local link structure keeps its slice of x in that XCD's L2.
constexpr int SPMV_CH = 2048;

__device__ __forceinline__ int64_t sp_lower_bound(const int64_t* __restrict__ a, int64_t n, int64_t v) {
  int64_t lo = 0, hi = n;   // first i in [0, n) with a[i] >= v, else n
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (a[mid] < v) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

template <typename T>
__global__ __launch_bounds__(256) void sp_csr_spmv_stream_kernel(const int64_t* __restrict__ indptr,
                                                                 const int32_t* __restrict__ indices,
                                                                 const T* __restrict__ vals,
                                                                 const T* __restrict__ x, int64_t ldx,
                                                                 T* __restrict__ y, int64_t ldy, int64_t m,
                                                                 int64_t nnz, int nchunk, int accumulate,
                                                                 T* __restrict__ carry,
                                                                 int64_t* __restrict__ carry_row,
                                                                 const int64_t* __restrict__ plan) {
  __shared__ T prod[SPMV_CH];
  __shared__ int64_t rng[2];
  const int per = (nchunk + 7) >> 3;
  const int c = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);   // XCD k walks chunks [k*per, (k+1)*per)
  if (c >= nchunk) return;
  const int64_t e0 = (int64_t)c * SPMV_CH;
  const int64_t e1 = e0 + SPMV_CH < nnz ? e0 + SPMV_CH : nnz;
  const int tid = threadIdx.x;
  // Without a plan (sp_csr_spmv_plan: first row starting in each chunk, computed once per matrix) two lanes
  // search the row range here: ~20 dependent loads on the critical path of every workgroup.
  // (A workgroup-wide 256-ary search was slower still: 256 cache lines per step.)
  if (plan) {
    if (tid == 0) rng[0] = plan[c];
    if (tid == 64) rng[1] = plan[c + 1];
  } else {
    if (tid == 0) rng[0] = sp_lower_bound(indptr, m, e0);
    if (tid == 64) rng[1] = (c == nchunk - 1) ? m : sp_lower_bound(indptr, m, e1);
  }
#pragma unroll
  for (int u = 0; u < SPMV_CH / 256; ++u) {
    const int64_t e = e0 + u * 256 + tid;
    T p = 0;
    if (e < e1) {
      const T v = vals[e];
      p = x ? v * x[(int64_t)indices[e] * ldx] : v;
    }
    prod[u * 256 + tid] = p;
  }
  __syncthreads();
  const int64_t r_lo = rng[0], r_hi = rng[1];
  // head of the chunk: tail of a row that started in an earlier chunk
  int64_t c_end = r_lo < m ? indptr[r_lo] : nnz;
  if (c_end > e1) c_end = e1;
  if (tid < 64) {
    const int len = (int)(c_end - e0);
    if (len > 0) {
      // one wavefront, lane-strided partial sums then a fixed shuffle tree
      T s = 0;
      for (int i = tid; i < len; i += 64) s += prod[i];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
      if (tid == 0) {
        carry[c] = s;
        carry_row[c] = r_lo - 1;
      }
    } else if (tid == 0) {
      carry_row[c] = -1;
    }
  }
  for (int64_t r = r_lo + tid; r < r_hi; r += 256) {
    const int64_t a = indptr[r] - e0;
    int64_t b = indptr[r + 1];
    b = (b < e1 ? b : e1) - e0;
    T s = 0;
    for (int64_t i = a; i < b; ++i) s += prod[i];
    y[r * ldy] = accumulate ? y[r * ldy] + s : s;
  }
}

// plan[c] = first row starting at or after entry c * SPMV_CH (c < nchunk), plan[nchunk] = m;
// plan[nchunk + 1] = length of the longest row (zeroed by the launcher, atomicMax here);
// plan[nchunk + 2] = arrival counter of the planned kernel's long-row mode (zero between launches).
__global__ __launch_bounds__(256) void sp_csr_spmv_plan_kernel(const int64_t* __restrict__ indptr, int64_t m,
                                                               int nchunk, int64_t* __restrict__ plan) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c <= nchunk) plan[c] = c == nchunk ? m : sp_lower_bound(indptr, m, (int64_t)c * SPMV_CH);
  long long longest = 0;
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < m; r += (int64_t)gridDim.x * 256) {
    const long long len = (long long)(indptr[r + 1] - indptr[r]);
    longest = len > longest ? len : longest;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const long long o = __shfl_xor(longest, off);
    longest = o > longest ? o : longest;
  }
  if ((threadIdx.x & 63) == 0 && longest > 0) atomicMax((long long*)&plan[nchunk + 1], longest);
}

// The planned form of the stream kernel: ONE launch, two dependent memory round trips per workgroup instead of
// four.  Everything whose address does not depend on loaded data is requested up front -- the chunk's entries, the
// SPMV_SPILL entries after the chunk, the plan's row range and, right behind it, the row pointers of the thread's
// own row -- then the gathers of x, one barrier, and the row sums from LDS.
// What bounds it (900 000 pages x 10 links, MI355X): 51.6 us, of which the random gather of x is 29 -- with x read
// at the entry's own position instead (SP_SPMV_ABLATE=1, timing only) the same launch takes 22.5 us = 3.85 TB/s of
// the algorithmic bytes.  Every gathered 4-byte value moves a whole cache line from the L2 to the CU (9 M lines
// per launch); the entry stream itself is not the limit, and non-temporal entry loads (SP_SPMV_NT=1) cut the
// counter traffic from 1.65 x to 1.33 x algorithmic but run slower (62 us).
//   * longest row <= SPMV_SPILL + 1 (the plan knows): a row belongs to the chunk it STARTS in and is summed whole
//     there, in storage order; what spills over the chunk's end is in the extra entries every workgroup loaded.
//     No carries, no second pass.
//   * longer rows: partial sums per chunk and carries as in sp_csr_spmv_stream_kernel; the LAST workgroup to
//     arrive (a ticket in the plan) adds the carries of every row in chunk order -- deterministic, no
//     floating-point atomics, and still one launch.
constexpr int SPMV_SPILL = 64;

template <typename T, bool NT, bool ABLATE_GATHER = false>
__global__ __launch_bounds__(256) void sp_csr_spmv_planned_kernel(const int64_t* __restrict__ indptr,
                                                                  const int32_t* __restrict__ indices,
                                                                  const T* __restrict__ vals,
                                                                  const T* __restrict__ x, int64_t ldx,
                                                                  T* __restrict__ y, int64_t ldy, int64_t m,
                                                                  int64_t nnz, int nchunk, int accumulate, T* carry,
                                                                  int64_t* carry_row, int64_t* plan) {
  __shared__ T prod[SPMV_CH + SPMV_SPILL];
  __shared__ int is_last;
  const int per = (nchunk + 7) >> 3;
  const int c = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);   // XCD k walks chunks [k*per, (k+1)*per)
  if (c >= nchunk) return;
  const int64_t e0 = (int64_t)c * SPMV_CH;
  const int64_t e1 = e0 + SPMV_CH < nnz ? e0 + SPMV_CH : nnz;
  const int tid = threadIdx.x;
  const int64_t r_lo = plan[c], r_hi = plan[c + 1];
  const bool whole_rows = plan[nchunk + 1] <= SPMV_SPILL + 1;
  T v[SPMV_CH / 256];
  int32_t col[SPMV_CH / 256];
#pragma unroll
  for (int u = 0; u < SPMV_CH / 256; ++u) {
    const int64_t e = e0 + u * 256 + tid;
    v[u] = 0;
    col[u] = 0;
    if (e < e1) {
      v[u] = NT ? __builtin_nontemporal_load(vals + e) : vals[e];
      col[u] = NT ? __builtin_nontemporal_load(indices + e) : indices[e];
    }
  }
  T vs = 0;
  int32_t cs = 0;
  const bool spill = whole_rows && tid < SPMV_SPILL && e1 + tid < nnz;
  if (spill) {
    vs = vals[e1 + tid];
    cs = indices[e1 + tid];
  }
  int64_t r = r_lo + tid;
  int64_t ra = 0, rb = 0;
  if (r < r_hi) {
    ra = indptr[r];
    rb = indptr[r + 1];
  }
#pragma unroll
  for (int u = 0; u < SPMV_CH / 256; ++u) {
    // (timing ablation only: x read at the entry's own position instead of its column -- a coalesced stream)
    const int64_t xi = ABLATE_GATHER ? (e0 + u * 256 + tid) % m : (int64_t)col[u];
    prod[u * 256 + tid] = x ? v[u] * x[xi * ldx] : v[u];     // (a non-temporal gather of x: 80 us instead of 52)
  }
  if (tid < SPMV_SPILL) prod[SPMV_CH + tid] = spill ? (x ? vs * x[(int64_t)cs * ldx] : vs) : (T)0;
  __syncthreads();
  if (whole_rows) {
    while (r < r_hi) {
      const int a = (int)(ra - e0), b = (int)(rb - e0);       // b <= SPMV_CH + SPMV_SPILL: the row starts before e1
      T s = 0;
      for (int i = a; i < b; ++i) s += prod[i];
      y[r * ldy] = accumulate ? y[r * ldy] + s : s;
      r += 256;
      if (r < r_hi) {
        ra = indptr[r];
        rb = indptr[r + 1];
      }
    }
    return;
  }
  // ---- long rows: per-chunk partial sums + carries
  int64_t c_end = r_lo < m ? indptr[r_lo] : nnz;
  if (c_end > e1) c_end = e1;
  if (tid < 64) {
    const int len = (int)(c_end - e0);
    if (len > 0) {
      T s = 0;
      for (int i = tid; i < len; i += 64) s += prod[i];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
      if (tid == 0) {
        carry[c] = s;
        carry_row[c] = r_lo - 1;
      }
    } else if (tid == 0) {
      carry_row[c] = -1;
    }
  }
  while (r < r_hi) {
    const int a = (int)(ra - e0);
    const int b = (int)((rb < e1 ? rb : e1) - e0);
    T s = 0;
    for (int i = a; i < b; ++i) s += prod[i];
    y[r * ldy] = accumulate ? y[r * ldy] + s : s;
    r += 256;
    if (r < r_hi) {
      ra = indptr[r];
      rb = indptr[r + 1];
    }
  }
  __threadfence();                       // my rows and my carry are visible device-wide before I take a ticket
  __syncthreads();
  if (tid == 0) {
    const unsigned long long t = atomicAdd((unsigned long long*)&plan[nchunk + 2], 1ull);
    is_last = (t == (unsigned long long)(nchunk - 1));
  }
  __syncthreads();
  if (!is_last) return;
  __threadfence();
  for (int cc = tid; cc < nchunk; cc += 256) {
    const int64_t row = __builtin_nontemporal_load(carry_row + cc);
    if (row < 0) continue;
    // only the FIRST chunk carrying into `row` adds (in chunk order) every carry of that row
    if (cc > 0 && __builtin_nontemporal_load(carry_row + cc - 1) == row) continue;
    const int64_t row_end = indptr[row + 1];
    T s = __builtin_nontemporal_load(y + row * ldy);
    for (int k = cc; k < nchunk && (int64_t)k * SPMV_CH < row_end && __builtin_nontemporal_load(carry_row + k) == row; ++k)
      s += __builtin_nontemporal_load(carry + k);
    y[row * ldy] = s;
  }
  if (tid == 0) plan[nchunk + 2] = 0;    // the next launch starts from zero
}

template <typename T>
__global__ __launch_bounds__(256) void sp_csr_spmv_fixup_kernel(const int64_t* __restrict__ indptr, T* __restrict__ y,
                                                                int64_t ldy, int nchunk,
                                                                const T* __restrict__ carry,
                                                                const int64_t* __restrict__ carry_row) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= nchunk) return;
  const int64_t row = carry_row[c];
  if (row < 0) return;
  // only the FIRST chunk carrying into `row` adds (in chunk order) every carry of that row
  if (c > 0 && carry_row[c - 1] == row) return;
  const int64_t row_end = indptr[row + 1];
  T s = y[row * ldy];
  for (int cc = c; cc < nchunk && (int64_t)cc * SPMV_CH < row_end && carry_row[cc] == row; ++cc) s += carry[cc];
  y[row * ldy] = s;
}

// N > 1: NL lanes along the columns of C (V columns each), 64 / NL rows per wavefront; a row's entries are
// walked in storage order, y += a * B[k, :] (the order of scipy's csr_matvecs).
template <typename T, int V>
__global__ __launch_bounds__(256) void sp_csr_spmm_kernel(const int64_t* __restrict__ indptr,
                                                          const int32_t* __restrict__ indices,
                                                          const T* __restrict__ vals, const T* __restrict__ B,
                                                          int64_t ldb, T* __restrict__ C, int64_t ldc, int64_t m,
                                                          int64_t n, int nl, int accumulate) {
  const int lr = threadIdx.x % nl;            // lane inside the row group
  const int rows_per_block = 256 / nl;
  const int64_t stride = (int64_t)gridDim.x * rows_per_block;
  for (int64_t r = (int64_t)blockIdx.x * rows_per_block + threadIdx.x / nl; r < m; r += stride) {
    const int64_t a = indptr[r], b = indptr[r + 1];
    for (int64_t c0 = (int64_t)lr * V; c0 < n; c0 += (int64_t)nl * V) {
      T acc[V];
#pragma unroll
      for (int v = 0; v < V; ++v) acc[v] = (accumulate && c0 + v < n) ? C[r * ldc + c0 + v] : (T)0;
      // U entries at a time: their (column, value) loads and then their U rows of B are all in flight before the
      // first multiply-add (one entry at a time is a chain of two dependent loads per entry); the additions
      // keep the storage order
      constexpr int U = 4;
      int64_t j = a;
      for (; j + U <= b; j += U) {
        T av[U];
        int32_t kk[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          av[u] = vals[j + u];
          kk[u] = indices[j + u];
        }
        T bv[U][V];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const T* brow = B + (int64_t)kk[u] * ldb + c0;
#pragma unroll
          for (int v = 0; v < V; ++v) bv[u][v] = (c0 + v < n) ? brow[v] : (T)0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
          for (int v = 0; v < V; ++v) acc[v] += av[u] * bv[u][v];
        }
      }
      for (; j < b; ++j) {
        const T av = vals[j];
        const T* brow = B + (int64_t)indices[j] * ldb + c0;
#pragma unroll
        for (int v = 0; v < V; ++v)
          if (c0 + v < n) acc[v] += av * brow[v];
      }
#pragma unroll
      for (int v = 0; v < V; ++v)
        if (c0 + v < n) C[r * ldc + c0 + v] = acc[v];
    }
  }
}

// ---------------------------------------------------------------- sparse -> dense
// mode 0: out = v   1: out += v   2: sparse.pyx:21-38 with REDUCE_ADD -- first write where the mask is clear
// (and set it), add where it is set.  Canonical CSR has no duplicate coordinates: no two threads share a cell.
template <typename T>
__global__ __launch_bounds__(256) void sp_csr_scatter_kernel(const int64_t* __restrict__ indptr,
                                                             const int32_t* __restrict__ indices,
                                                             const T* __restrict__ vals, int64_t m, int64_t nnz,
                                                             T* __restrict__ out, int64_t ld, int64_t row0,
                                                             int64_t col0, uint8_t* __restrict__ mask,
                                                             int64_t ldmask, int mode) {
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < nnz; t += (int64_t)gridDim.x * 256) {
    const int64_t r = sp_row_of(indptr, m, t) + row0;
    const int64_t c = (int64_t)indices[t] + col0;
    T* o = out + r * ld + c;
    if (mode == 0) {
      *o = vals[t];
    } else if (mode == 1) {
      *o += vals[t];
    } else {
      uint8_t* mk = mask + r * ldmask + c;
      if (*mk) {
        *o += vals[t];
      } else {
        *o = vals[t];
        *mk = 1;
      }
    }
  }
}

// ---------------------------------------------------------------- sparse x sparse, expansion step
// counts[t] = entries of B's row indicesA[t]
__global__ __launch_bounds__(256) void sp_spgemm_count_kernel(const int32_t* __restrict__ ia, int64_t nnza,
                                                              const int64_t* __restrict__ pb,
                                                              int* __restrict__ counts,
                                                              unsigned long long* __restrict__ total) {
  unsigned long long mine = 0;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < nnza; t += (int64_t)gridDim.x * 256) {
    const int64_t k = ia[t];
    const int64_t c = pb[k + 1] - pb[k];
    counts[t] = (int)c;
    mine += (unsigned long long)c;
  }
  for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(total, mine);   // integer: order does not matter
}

// products of A's entry t = (i, k, a) with B's row k, written at offs[t] in A's storage order, so that the
// stable sort adds the products of one output cell in the order scipy's csr_matmat does (k ascending)
template <typename T>
__global__ __launch_bounds__(256) void sp_spgemm_expand_kernel(const int64_t* __restrict__ pa, int64_t ma,
                                                               const int32_t* __restrict__ ia,
                                                               const T* __restrict__ va, int64_t nnza,
                                                               const int64_t* __restrict__ pb,
                                                               const int32_t* __restrict__ ib,
                                                               const T* __restrict__ vb,
                                                               const int* __restrict__ offs,
                                                               int32_t* __restrict__ rows,
                                                               int32_t* __restrict__ cols, T* __restrict__ out) {
  constexpr int G = 8;
  const int g = threadIdx.x % G;
  const int64_t stride = (int64_t)gridDim.x * (256 / G);
  for (int64_t t = (int64_t)blockIdx.x * (256 / G) + threadIdx.x / G; t < nnza; t += stride) {
    const int32_t i = (int32_t)sp_row_of(pa, ma, t);
    const int64_t k = ia[t];
    const T a = va[t];
    const int64_t b0 = pb[k], len = pb[k + 1] - b0;
    const int64_t o = offs[t];
    for (int64_t j = g; j < len; j += G) {
      rows[o + j] = i;
      cols[o + j] = ib[b0 + j];
      out[o + j] = a * vb[b0 + j];
    }
  }
}

inline unsigned grid_for(int64_t n, int per_block) {
  int64_t b = (n + per_block - 1) / per_block;
  const int64_t cap = (int64_t)SP_CUS * SP_BLOCKS_PER_CU * 4;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}

inline int key_bits(int64_t nrows, int64_t ncols) {
  // keys are < nrows * ncols; one more value keeps the all-ones pattern of dropped entries above every key
  const unsigned __int128 span = (unsigned __int128)nrows * (unsigned __int128)ncols + 1;
  int bits = 1;
  while (bits < 64 && ((unsigned __int128)1 << bits) < span) ++bits;
  return bits;
}

}  // namespace

extern "C" size_t sp_coo_to_csr_workspace_bytes(int64_t nnz) {
  if (nnz < 1) return 256;
  return sp_sort_ws_bytes<uint64_t>(nnz, nullptr, nullptr);
}

extern "C" int sp_coo_to_csr(int32_t dtype, int64_t nrows, int64_t ncols, int64_t nnz, const int32_t* d_rows,
                             const int32_t* d_cols, const void* d_vals, int64_t* d_indptr, int32_t* d_indices,
                             void* d_vals_out, void* d_ws, size_t ws_bytes, void* stream) {
  if (dtype != SP_F32 && dtype != SP_F64) SP_FAIL("sp_coo_to_csr: values must be f32 or f64");
  if (nrows < 0 || ncols < 0 || nnz < 0) SP_FAIL("sp_coo_to_csr: bad sizes");
  if (nrows > 2147483647LL || ncols > 2147483647LL) SP_FAIL("sp_coo_to_csr: a dimension exceeds the int32 index range");
  if (nnz > 2147483647LL - SCAN_CHUNK) SP_FAIL("sp_coo_to_csr: more than 2^31 entries in one tile");
  if (!d_indptr) SP_FAIL("sp_coo_to_csr: NULL pointer");
  hipStream_t st = (hipStream_t)stream;
  if (nnz == 0 || ncols == 0 || nrows == 0) {
    SP_HIP(hipMemsetAsync(d_indptr, 0, (size_t)(nrows + 1) * 8, st));
    return 0;
  }
  if (!d_rows || !d_cols || !d_vals || !d_indices || !d_vals_out) SP_FAIL("sp_coo_to_csr: NULL pointer");
  if (!d_ws || ws_bytes < sp_coo_to_csr_workspace_bytes(nnz)) SP_FAIL("sp_coo_to_csr: workspace too small");
  SortWs ws;
  sp_sort_ws_bytes(nnz, &ws, (char*)d_ws);
  hipLaunchKernelGGL(sp_coo_keys_kernel, dim3(grid_for(nnz, 256)), dim3(256), 0, st, d_rows, d_cols, nnz, ncols,
                     ws.keys[0], ws.idx[0]);
  SP_CHECK_LAUNCH();
  const int bits = key_bits(nrows, ncols);
  int cur = 0;
  for (int shift = 0; shift < bits; shift += RDX_BITS) {
    if (sp_radix_pass(ws, cur, nnz, DigitOfKey{shift}, st)) return 1;
    cur = 1 - cur;
  }
  int* pos = ws.hist;
  hipLaunchKernelGGL(sp_coo_heads_kernel, dim3(grid_for(nnz, 256)), dim3(256), 0, st, ws.keys[cur], nnz, pos);
  SP_CHECK_LAUNCH();
  if (sp_exscan_int(pos, nnz, ws.sums, ws.total, st)) return 1;
  if (dtype == SP_F32)
    hipLaunchKernelGGL((sp_coo_compress_kernel<float>), dim3(grid_for(nnz, 256)), dim3(256), 0, st, ws.keys[cur],
                       ws.idx[cur], (const float*)d_vals, nnz, pos, ncols, d_indices, (float*)d_vals_out);
  else
    hipLaunchKernelGGL((sp_coo_compress_kernel<double>), dim3(grid_for(nnz, 256)), dim3(256), 0, st, ws.keys[cur],
                       ws.idx[cur], (const double*)d_vals, nnz, pos, ncols, d_indices, (double*)d_vals_out);
  SP_CHECK_LAUNCH();
  hipLaunchKernelGGL(sp_csr_indptr_kernel, dim3(grid_for(nrows + 1, 256)), dim3(256), 0, st, ws.keys[cur], nnz, pos,
                     ws.total, nrows, ncols, d_indptr);
  SP_CHECK_LAUNCH();
  return 0;
}

extern "C" int sp_csr_rows(int64_t nrows, int64_t nnz, const int64_t* d_indptr, int32_t* d_rows, void* stream) {
  if (nrows < 0 || nnz < 0) SP_FAIL("sp_csr_rows: bad sizes");
  if (nnz == 0) return 0;
  if (!d_indptr || !d_rows) SP_FAIL("sp_csr_rows: NULL pointer");
  hipLaunchKernelGGL(sp_csr_rows_kernel, dim3(grid_for(nnz, 256)), dim3(256), 0, (hipStream_t)stream, d_indptr, nrows,
                     nnz, d_rows);
  SP_CHECK_LAUNCH();
  return 0;
}

extern "C" int sp_coo_reshape(int64_t nnz, int32_t* d_rows, int32_t* d_cols, int64_t old_cols, int64_t offset,
                              int64_t new_rows, int64_t new_cols, void* stream) {
  if (nnz < 0 || old_cols < 0 || new_rows < 0 || new_cols < 0) SP_FAIL("sp_coo_reshape: bad sizes");
  if (new_rows > 2147483647LL || new_cols > 2147483647LL) SP_FAIL("sp_coo_reshape: a dimension exceeds the int32 index range");
  if (nnz == 0) return 0;
  if (!d_rows || !d_cols) SP_FAIL("sp_coo_reshape: NULL pointer");
  if (new_cols == 0) SP_FAIL("sp_coo_reshape: empty target with entries");
  hipLaunchKernelGGL(sp_coo_reshape_kernel, dim3(grid_for(nnz, 256)), dim3(256), 0, (hipStream_t)stream, d_rows, d_cols,
                     nnz, old_cols, offset, new_rows, new_cols);
  SP_CHECK_LAUNCH();
  return 0;
}

extern "C" int sp_coo_box(int64_t nnz, int32_t* d_rows, int32_t* d_cols, int64_t r0, int64_t r1, int64_t c0, int64_t c1,
                          int64_t dr, int64_t dc, int32_t drop_inside, void* stream) {
  if (nnz < 0) SP_FAIL("sp_coo_box: bad size");
  if (nnz == 0) return 0;
  if (!d_rows || !d_cols) SP_FAIL("sp_coo_box: NULL pointer");
  hipLaunchKernelGGL(sp_coo_box_kernel, dim3(grid_for(nnz, 256)), dim3(256), 0, (hipStream_t)stream, d_rows, d_cols,
                     nnz, r0, r1, c0, c1, dr, dc, (int)drop_inside);
  SP_CHECK_LAUNCH();
  return 0;
}

namespace {

inline int spmv_chunks(int64_t nnz) { return (int)((nnz + SPMV_CH - 1) / SPMV_CH); }

template <typename T>
int spmm_go(int64_t m, int64_t n, int64_t nnz, const int64_t* indptr, const int32_t* indices, const T* vals, const T* B,
            int64_t ldb, T* C, int64_t ldc, int accumulate, const int64_t* plan, void* ws, size_t ws_bytes,
            hipStream_t st) {
  if ((n == 1 || !B) && nnz > 0) {
    // entry-split stream kernel unless rows are very long on average (its one-thread-per-row LDS sums would
    // serialise; measured faster than the lanes-per-row kernel up to 400 entries per row: 0.44 vs 0.51 ms on
    // 200 000 x 400, 2.27 vs 2.39 ms on 4 000 000 x 40)
    const double mean_len = m > 0 ? (double)nnz / (double)m : 0.0;
    const char* ea = getenv("SP_SPMV_ALGO");      // "stream" | "vector": tuning / test knob
    bool stream = mean_len < 1024.0;
    if (ea && ea[0] == 's') stream = true;
    if (ea && ea[0] == 'v') stream = false;
    const int nchunk = spmv_chunks(nnz);
    const size_t need = sp_al256((size_t)nchunk * sizeof(T)) + sp_al256((size_t)nchunk * 8);
    if (stream && ws && ws_bytes >= need) {
      T* carry = (T*)ws;
      int64_t* carry_row = (int64_t*)((char*)ws + sp_al256((size_t)nchunk * sizeof(T)));
      const int per = (nchunk + 7) / 8;
      const char* ep = getenv("SP_SPMV_PLANNED");      // "0": the two-launch form even with a plan (A/B knob)
      if (plan && !(ep && ep[0] == '0')) {
        const char* en = getenv("SP_SPMV_NT");
        const char* ab = getenv("SP_SPMV_ABLATE");        // timing-only: wrong results
        if (ab && ab[0] == '1')
          hipLaunchKernelGGL((sp_csr_spmv_planned_kernel<T, false, true>), dim3(per * 8), dim3(256), 0, st, indptr, indices, vals, B, ldb,
                             C, ldc, m, nnz, nchunk, accumulate, carry, carry_row, const_cast<int64_t*>(plan));
        else if (!(en && en[0] == '1'))      // default: plain loads (non-temporal entry loads measured slower here: 62 vs 52 us)
          hipLaunchKernelGGL((sp_csr_spmv_planned_kernel<T, false>), dim3(per * 8), dim3(256), 0, st, indptr, indices, vals, B, ldb,
                             C, ldc, m, nnz, nchunk, accumulate, carry, carry_row, const_cast<int64_t*>(plan));
        else
          hipLaunchKernelGGL((sp_csr_spmv_planned_kernel<T, true>), dim3(per * 8), dim3(256), 0, st, indptr, indices, vals, B, ldb,
                             C, ldc, m, nnz, nchunk, accumulate, carry, carry_row, const_cast<int64_t*>(plan));
        SP_CHECK_LAUNCH();
        return 0;
      }
      hipLaunchKernelGGL((sp_csr_spmv_stream_kernel<T>), dim3(per * 8), dim3(256), 0, st, indptr, indices, vals, B, ldb,
                         C, ldc, m, nnz, nchunk, accumulate, carry, carry_row, plan);
      SP_CHECK_LAUNCH();
      hipLaunchKernelGGL((sp_csr_spmv_fixup_kernel<T>), dim3((nchunk + 255) / 256), dim3(256), 0, st, indptr, C, ldc,
                         nchunk, carry, carry_row);
      SP_CHECK_LAUNCH();
      return 0;
    }
  }
  if (n == 1 || !B) {
    // lanes per row: the power of two nearest above the mean row length, 2 .. 64
    const double mean = m > 0 ? (double)nnz / (double)m : 0.0;
    int G = 2;
    while (G < 64 && G < mean) G <<= 1;
    const char* e = getenv("SP_SPMV_G");
    if (e && atoi(e) > 0) G = atoi(e);
    const unsigned grid = grid_for(m, 256 / G);
#define SP_SPMV_GO(GG)                                                                                           \
  hipLaunchKernelGGL((sp_csr_spmv_kernel<T, GG>), dim3(grid), dim3(256), 0, st, indptr, indices, vals, B, ldb, C, ldc, \
                     m, accumulate)
    switch (G) {
      case 2: SP_SPMV_GO(2); break;
      case 4: SP_SPMV_GO(4); break;
      case 8: SP_SPMV_GO(8); break;
      case 16: SP_SPMV_GO(16); break;
      case 32: SP_SPMV_GO(32); break;
      default: SP_SPMV_GO(64); break;
    }
#undef SP_SPMV_GO
    SP_CHECK_LAUNCH();
    return 0;
  }
  int V = n >= 256 ? 4 : (n >= 128 ? 2 : 1);
  int nl = 1;
  while (nl < 64 && (int64_t)nl * V < n) nl <<= 1;
  const unsigned grid = grid_for(m, 256 / nl);
  if (V == 4)
    hipLaunchKernelGGL((sp_csr_spmm_kernel<T, 4>), dim3(grid), dim3(256), 0, st, indptr, indices, vals, B, ldb, C, ldc, m,
                       n, nl, accumulate);
  else if (V == 2)
    hipLaunchKernelGGL((sp_csr_spmm_kernel<T, 2>), dim3(grid), dim3(256), 0, st, indptr, indices, vals, B, ldb, C, ldc, m,
                       n, nl, accumulate);
  else
    hipLaunchKernelGGL((sp_csr_spmm_kernel<T, 1>), dim3(grid), dim3(256), 0, st, indptr, indices, vals, B, ldb, C, ldc, m,
                       n, nl, accumulate);
  SP_CHECK_LAUNCH();
  return 0;
}

}  // namespace

extern "C" size_t sp_csr_spmm_workspace_bytes(int64_t nnz, int64_t n) {
  if (n != 1 || nnz < 1) return 256;
  const size_t nchunk = (size_t)spmv_chunks(nnz);
  return sp_al256(nchunk * 8) + sp_al256(nchunk * 8) + 256;
}

extern "C" int64_t sp_csr_spmv_plan_entries(int64_t nnz) { return nnz < 1 ? 3 : (int64_t)spmv_chunks(nnz) + 3; }

extern "C" int sp_csr_spmv_plan(int64_t m, int64_t nnz, const int64_t* d_indptr, int64_t* d_plan, void* stream) {
  if (m < 0 || nnz < 0) SP_FAIL("sp_csr_spmv_plan: bad sizes");
  if (!d_indptr || !d_plan) SP_FAIL("sp_csr_spmv_plan: NULL pointer");
  const int nchunk = nnz < 1 ? 0 : spmv_chunks(nnz);
  SP_HIP(hipMemsetAsync(d_plan + nchunk + 1, 0, 16, (hipStream_t)stream));      // longest row, arrival counter
  int blocks = (nchunk + 256) / 256;
  const int64_t want = (m + 255) / 256;            // the longest-row scan strides over the rows
  if (want > blocks) blocks = (int)(want < 2048 ? want : 2048);
  hipLaunchKernelGGL(sp_csr_spmv_plan_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_indptr,
                     m, nchunk, d_plan);
  SP_CHECK_LAUNCH();
  return 0;
}

extern "C" int sp_csr_spmm(int32_t dtype, int64_t m, int64_t k, int64_t n, int64_t nnz, const int64_t* d_indptr,
                           const int32_t* d_indices, const void* d_vals, const void* d_b, int64_t ldb, void* d_c,
                           int64_t ldc, int32_t accumulate, const int64_t* d_plan, void* d_ws, size_t ws_bytes,
                           void* stream) {
  if (dtype != SP_F32 && dtype != SP_F64) SP_FAIL("sp_csr_spmm: values must be f32 or f64");
  if (m < 0 || k < 0 || n < 0 || nnz < 0) SP_FAIL("sp_csr_spmm: bad sizes");
  if (m == 0 || n == 0) return 0;
  if (!d_indptr || !d_c || (nnz && (!d_indices || !d_vals))) SP_FAIL("sp_csr_spmm: NULL pointer");
  if (!d_b && n != 1) SP_FAIL("sp_csr_spmm: a NULL right-hand side (row sums) needs n == 1");
  if (ldc < n || (d_b && ldb < n)) SP_FAIL("sp_csr_spmm: leading dimension too small");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SP_F32)
    return spmm_go<float>(m, n, nnz, d_indptr, d_indices, (const float*)d_vals, (const float*)d_b, ldb, (float*)d_c, ldc,
                          accumulate, d_plan, d_ws, ws_bytes, st);
  return spmm_go<double>(m, n, nnz, d_indptr, d_indices, (const double*)d_vals, (const double*)d_b, ldb, (double*)d_c,
                         ldc, accumulate, d_plan, d_ws, ws_bytes, st);
}

extern "C" int sp_csr_scatter(int32_t dtype, int64_t m, int64_t nnz, const int64_t* d_indptr, const int32_t* d_indices,
                              const void* d_vals, void* d_out, int64_t ld, int64_t row0, int64_t col0, uint8_t* d_mask,
                              int64_t ldmask, int32_t mode, void* stream) {
  if (dtype != SP_F32 && dtype != SP_F64) SP_FAIL("sp_csr_scatter: values must be f32 or f64");
  if (m < 0 || nnz < 0 || mode < 0 || mode > 2) SP_FAIL("sp_csr_scatter: bad arguments");
  if (nnz == 0) return 0;
  if (!d_indptr || !d_indices || !d_vals || !d_out) SP_FAIL("sp_csr_scatter: NULL pointer");
  if (mode == 2 && !d_mask) SP_FAIL("sp_csr_scatter: masked mode needs a mask");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SP_F32)
    hipLaunchKernelGGL((sp_csr_scatter_kernel<float>), dim3(grid_for(nnz, 256)), dim3(256), 0, st, d_indptr, d_indices,
                       (const float*)d_vals, m, nnz, (float*)d_out, ld, row0, col0, d_mask, ldmask, (int)mode);
  else
    hipLaunchKernelGGL((sp_csr_scatter_kernel<double>), dim3(grid_for(nnz, 256)), dim3(256), 0, st, d_indptr, d_indices,
                       (const double*)d_vals, m, nnz, (double*)d_out, ld, row0, col0, d_mask, ldmask, (int)mode);
  SP_CHECK_LAUNCH();
  return 0;
}

extern "C" size_t sp_spgemm_count_workspace_bytes(int64_t nnz_a) {
  return sp_al256((size_t)((nnz_a + SCAN_CHUNK - 1) / SCAN_CHUNK + 1) * 4) + 256;
}

extern "C" int sp_spgemm_count(int64_t nnz_a, const int32_t* d_indices_a, const int64_t* d_indptr_b, int32_t* d_offs,
                               int64_t* d_total, void* d_ws, size_t ws_bytes, void* stream) {
  if (nnz_a < 0 || nnz_a > 2147483647LL - SCAN_CHUNK) SP_FAIL("sp_spgemm_count: bad size");
  if (!d_offs || !d_total) SP_FAIL("sp_spgemm_count: NULL pointer");
  hipStream_t st = (hipStream_t)stream;
  SP_HIP(hipMemsetAsync(d_total, 0, 8, st));
  if (nnz_a == 0) {
    SP_HIP(hipMemsetAsync(d_offs, 0, 4, st));
    return 0;
  }
  if (!d_indices_a || !d_indptr_b) SP_FAIL("sp_spgemm_count: NULL pointer");
  if (!d_ws || ws_bytes < sp_spgemm_count_workspace_bytes(nnz_a)) SP_FAIL("sp_spgemm_count: workspace too small");
  hipLaunchKernelGGL(sp_spgemm_count_kernel, dim3(grid_for(nnz_a, 256)), dim3(256), 0, st, d_indices_a, nnz_a,
                     d_indptr_b, (int*)d_offs, (unsigned long long*)d_total);
  SP_CHECK_LAUNCH();
  // offs[t] = exclusive prefix (int); *d_total = the exact 64-bit number of products -- the caller refuses an
  // expansion that does not fit the int offsets before calling sp_spgemm_expand
  return sp_exscan_int((int*)d_offs, nnz_a, (int*)d_ws, (int*)d_offs + nnz_a, st);
}

extern "C" int sp_spgemm_expand(int32_t dtype, int64_t m_a, int64_t nnz_a, const int64_t* d_indptr_a,
                                const int32_t* d_indices_a, const void* d_vals_a, const int64_t* d_indptr_b,
                                const int32_t* d_indices_b, const void* d_vals_b, const int32_t* d_offs,
                                int32_t* d_rows, int32_t* d_cols, void* d_vals, void* stream) {
  if (dtype != SP_F32 && dtype != SP_F64) SP_FAIL("sp_spgemm_expand: values must be f32 or f64");
  if (m_a < 0 || nnz_a < 0) SP_FAIL("sp_spgemm_expand: bad sizes");
  if (nnz_a == 0) return 0;
  if (!d_indptr_a || !d_indices_a || !d_vals_a || !d_indptr_b || !d_offs) SP_FAIL("sp_spgemm_expand: NULL pointer");
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = grid_for(nnz_a, 256 / 8);
  if (dtype == SP_F32)
    hipLaunchKernelGGL((sp_spgemm_expand_kernel<float>), dim3(grid), dim3(256), 0, st, d_indptr_a, m_a, d_indices_a,
                       (const float*)d_vals_a, nnz_a, d_indptr_b, d_indices_b, (const float*)d_vals_b,
                       (const int*)d_offs, d_rows, d_cols, (float*)d_vals);
  else
    hipLaunchKernelGGL((sp_spgemm_expand_kernel<double>), dim3(grid), dim3(256), 0, st, d_indptr_a, m_a, d_indices_a,
                       (const double*)d_vals_a, nnz_a, d_indptr_b, d_indices_b, (const double*)d_vals_b,
                       (const int*)d_offs, d_rows, d_cols, (double*)d_vals);
  SP_CHECK_LAUNCH();
  return 0;
}

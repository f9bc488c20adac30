// In-place exclusive scan of int arrays in HBM (shared by the counting sorts of kmeans.hip and
// sparse.hip).  Each translation unit gets its own copy (anonymous namespace).
#pragma once
#include "sp_common.hpp"

namespace {

// exclusive scan of `m` ints in place, three coalesced phases over chunks of 4096:
//   1. chunk sums   2. one workgroup scans the (<= 4096 per pass) chunk sums   3. local scan + chunk offset
constexpr int SCAN_CHUNK = 4096;

__global__ __launch_bounds__(1024) void sp_scan_sums_kernel(const int* __restrict__ a, int64_t m,
                                                            int* __restrict__ sums) {
  __shared__ int red[16];
  const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK;
  int s = 0;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int64_t i = base + u * 1024 + threadIdx.x;
    if (i < m) s += a[i];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < 16; ++w) t += red[w];
    sums[blockIdx.x] = t;
  }
}

// block-wide exclusive scan of one value per thread (1024 threads); returns the exclusive prefix, *total = sum
__device__ __forceinline__ int sp_block_exscan_1024(int v, int* total) {
  __shared__ int wsum[16];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(inc, off);
    if (lane >= off) inc += o;
  }
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  int woff = 0, tot = 0;
  for (int i = 0; i < 16; ++i) {
    const int t = wsum[i];
    if (i < w) woff += t;
    tot += t;
  }
  __syncthreads();
  if (total) *total = tot;
  return woff + inc - v;
}

// in-place exclusive scan of the n_sums chunk sums by one workgroup (sequential passes of 1024)
__global__ __launch_bounds__(1024) void sp_scan_top_kernel(int* __restrict__ sums, int n_sums,
                                                           int* __restrict__ total_out) {
  int carry = 0;
  for (int base = 0; base < n_sums; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = i < n_sums ? sums[i] : 0;
    int tot;
    const int ex = sp_block_exscan_1024(v, &tot);
    if (i < n_sums) sums[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0 && total_out) *total_out = carry;
}

__global__ __launch_bounds__(1024) void sp_scan_apply_kernel(int* __restrict__ a, int64_t m,
                                                             const int* __restrict__ sums) {
  const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK + (int64_t)threadIdx.x * 4;
  int v[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) v[u] = base + u < m ? a[base + u] : 0;
  const int mine = v[0] + v[1] + v[2] + v[3];
  int run = sums[blockIdx.x] + sp_block_exscan_1024(mine, nullptr);
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (base + u < m) a[base + u] = run;
    run += v[u];
  }
}

// host side: scan a[0..m) in place; `sums` needs ceil(m / SCAN_CHUNK) ints; *total (device, may be null) = sum
static inline int sp_exscan_int(int* a, int64_t m, int* sums, int* total, hipStream_t st) {
  if (m <= 0) {
    if (total) SP_HIP(hipMemsetAsync(total, 0, sizeof(int), st));
    return 0;
  }
  const int n_chunks = (int)((m + SCAN_CHUNK - 1) / SCAN_CHUNK);
  hipLaunchKernelGGL(sp_scan_sums_kernel, dim3(n_chunks), dim3(1024), 0, st, a, m, sums);
  SP_CHECK_LAUNCH();
  hipLaunchKernelGGL(sp_scan_top_kernel, dim3(1), dim3(1024), 0, st, sums, n_chunks, total);
  SP_CHECK_LAUNCH();
  hipLaunchKernelGGL(sp_scan_apply_kernel, dim3(n_chunks), dim3(1024), 0, st, a, m, sums);
  SP_CHECK_LAUNCH();
  return 0;
}

}  // namespace

// One stable LSD radix pass over (64-bit key, 32-bit payload) pairs in HBM, shared by sparse.hip (COO -> CSR)
// and sort.hip (sort / argsort of dense tiles).  The digit comes from a functor so a pass can look at a byte
// of the key or at something derived from the payload (the row of a flat position).
#pragma once
#include "sp_common.hpp"
#include "sp_scan.hpp"

namespace {

constexpr int RDX_BITS = 8;
constexpr int RDX = 1 << RDX_BITS;
constexpr int SORT_RB = 4096;            // keys per workgroup / wavefront of one radix pass

struct DigitOfKey {                      // byte `shift / 8` of the key
  int shift;
  __device__ __forceinline__ int operator()(uint64_t key, int32_t) const { return (int)((key >> shift) & (RDX - 1)); }
};

struct DigitOfRow {                      // byte `shift / 8` of the row (payload / cols) of a flat position
  int shift;
  uint32_t cols;
  __device__ __forceinline__ int operator()(uint64_t, int32_t idx) const {
    return (int)((((uint32_t)idx / cols) >> shift) & (RDX - 1));
  }
};

// hist[d * nblk + b] = number of keys with digit d in key block b
template <typename D>
__global__ __launch_bounds__(256) void sp_radix_hist_kernel(const uint64_t* __restrict__ keys,
                                                            const int32_t* __restrict__ idx, int64_t n, D dig,
                                                            int nblk, int* __restrict__ hist) {
  __shared__ int lh[RDX];
  const int b = blockIdx.x;
  lh[threadIdx.x] = 0;
  __syncthreads();
  const int64_t r0 = (int64_t)b * SORT_RB;
  const int64_t r1 = r0 + SORT_RB < n ? r0 + SORT_RB : n;
  for (int64_t i = r0 + threadIdx.x; i < r1; i += 256) atomicAdd(&lh[dig(keys[i], idx[i])], 1);
  __syncthreads();
  hist[(int64_t)threadIdx.x * nblk + b] = lh[threadIdx.x];
}

// One wavefront per key block walks it in order; the rank of a key among the keys of the same digit in its
// 64-key chunk is the number of LOWER lanes with that digit (64 readlane steps, no divergence), the last
// lane of each digit advances the LDS cursor -- the scheme of sp_label_rank_kernel (kmeans.hip).
template <typename D>
__global__ __launch_bounds__(64) void sp_radix_rank_kernel(const uint64_t* __restrict__ keys,
                                                           const int32_t* __restrict__ idx, int64_t n, D dig,
                                                           int nblk, const int* __restrict__ offs,
                                                           uint64_t* __restrict__ keys_out,
                                                           int32_t* __restrict__ idx_out) {
  __shared__ int cur[RDX];
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  for (int i = lane; i < RDX; i += 64) cur[i] = offs[(int64_t)i * nblk + b];
  __syncthreads();
  const int64_t r0 = (int64_t)b * SORT_RB;
  const int64_t r1 = r0 + SORT_RB < n ? r0 + SORT_RB : n;
  for (int64_t base = r0; base < r1; base += 64) {
    const int64_t i = base + lane;
    const bool valid = i < r1;
    uint64_t key = 0;
    int32_t id = 0;
    int dg = -1 - lane;  // invalid lanes: a value no other lane holds
    if (valid) {
      key = keys[i];
      id = idx[i];
      dg = dig(key, id);
    }
    int lower = 0, same = 0;
#pragma unroll
    for (int j = 0; j < 64; ++j) {
      const int dj = __builtin_amdgcn_readlane(dg, j);
      const int eq = (dj == dg) ? 1 : 0;
      same += eq;
      lower += (j < lane) ? eq : 0;
    }
    int start = 0;
    if (valid) start = cur[dg];
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    if (valid) {
      keys_out[start + lower] = key;
      idx_out[start + lower] = id;
      if (lower == same - 1) cur[dg] = start + same;
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
  }
}

inline size_t sp_al256(size_t v) { return (v + 255) & ~(size_t)255; }

// scratch of a sort of n pairs: two key arrays, two payload arrays, the histogram (reusable as n ints), scan sums
struct SortWs {
  uint64_t* keys[2];
  int32_t* idx[2];
  int* hist;   // [RDX][nblk] or n ints, whichever is larger
  int* sums;   // scan chunk sums
  int* total;
};

inline size_t sp_sort_ws_bytes(int64_t n, SortWs* ws, char* base) {
  const int64_t nblk = (n + SORT_RB - 1) / SORT_RB;
  const int64_t hist_words = (int64_t)RDX * nblk > n ? (int64_t)RDX * nblk : n;
  const int64_t sums_words = (hist_words + SCAN_CHUNK - 1) / SCAN_CHUNK + 1;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* p = base ? base + off : nullptr;
    off += sp_al256(bytes);
    return p;
  };
  char* k0 = take((size_t)n * 8);
  char* k1 = take((size_t)n * 8);
  char* i0 = take((size_t)n * 4);
  char* i1 = take((size_t)n * 4);
  char* h = take((size_t)hist_words * 4);
  char* s = take((size_t)sums_words * 4);
  char* t = take(256);
  if (ws) {
    ws->keys[0] = (uint64_t*)k0;
    ws->keys[1] = (uint64_t*)k1;
    ws->idx[0] = (int32_t*)i0;
    ws->idx[1] = (int32_t*)i1;
    ws->hist = (int*)h;
    ws->sums = (int*)s;
    ws->total = (int*)t;
  }
  return off;
}

// keys[cur] / idx[cur] -> keys[1 - cur] / idx[1 - cur], stable by the digit `dig`
template <typename D>
static inline int sp_radix_pass(SortWs& ws, int cur, int64_t n, D dig, hipStream_t st) {
  const int nblk = (int)((n + SORT_RB - 1) / SORT_RB);
  hipLaunchKernelGGL((sp_radix_hist_kernel<D>), dim3(nblk), dim3(256), 0, st, ws.keys[cur], ws.idx[cur], n, dig, nblk,
                     ws.hist);
  SP_CHECK_LAUNCH();
  if (sp_exscan_int(ws.hist, (int64_t)RDX * nblk, ws.sums, nullptr, st)) return 1;
  hipLaunchKernelGGL((sp_radix_rank_kernel<D>), dim3(nblk), dim3(64), 0, st, ws.keys[cur], ws.idx[cur], n, dig, nblk,
                     ws.hist, ws.keys[1 - cur], ws.idx[1 - cur]);
  SP_CHECK_LAUNCH();
  return 0;
}

}  // namespace

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

// One workgroup (4 wavefronts) per key block.  Wavefront w walks its quarter of the block in order, 64 keys at a
// time: the rank of a key among the keys of the same digit in its 64-key chunk is the number of LOWER lanes with
// that digit, a per-wave LDS counter carries the count from chunk to chunk.  The block then places its keys in LDS in digit order and
// writes every digit's run to its global position as one contiguous piece: the scattered 8 + 4 byte stores of a
// direct scatter cost 4x their bytes in HBM write traffic (rocprofv3 WRITE_SIZE), runs of whole cache lines do not.
constexpr int RANK_WAVES = 4;
constexpr int RANK_PER_WAVE = SORT_RB / RANK_WAVES;     // 1024 keys
constexpr int RANK_CHUNKS = RANK_PER_WAVE / 64;         // 16 chunks per wavefront

template <typename D>
__global__ __launch_bounds__(256) void sp_radix_rank_kernel(const uint64_t* __restrict__ keys,
                                                           const int32_t* __restrict__ idx, int64_t n, D dig,
                                                           int nblk, const int* __restrict__ offs,
                                                           uint64_t* __restrict__ keys_out,
                                                           int32_t* __restrict__ idx_out) {
  __shared__ uint64_t sk[SORT_RB];
  __shared__ int32_t si[SORT_RB];
  __shared__ uint16_t wcnt[RANK_WAVES][RDX];   // keys of digit d seen so far by wavefront w; then its base
  __shared__ int lstart[RDX];                  // first LDS slot of digit d
  __shared__ int goff[RDX];                    // global position of this block's first key of digit d
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int i = tid; i < RANK_WAVES * RDX; i += 256) (&wcnt[0][0])[i] = 0;
  goff[tid] = offs[(int64_t)tid * nblk + b];
  __syncthreads();
  const int64_t r0 = (int64_t)b * SORT_RB;
  const int64_t r1 = r0 + SORT_RB < n ? r0 + SORT_RB : n;
  uint64_t key[RANK_CHUNKS];
  int32_t id[RANK_CHUNKS];
  int32_t pos[RANK_CHUNKS];                    // digit << 16 | rank inside this wavefront's quarter
#pragma unroll
  for (int c = 0; c < RANK_CHUNKS; ++c) {
    const int64_t i = r0 + (int64_t)w * RANK_PER_WAVE + c * 64 + lane;
    const bool valid = i < r1;
    key[c] = 0;
    id[c] = 0;
    int dg = -1 - lane;  // invalid lanes: a value no other lane holds
    if (valid) {
      key[c] = keys[i];
      id[c] = idx[i];
      dg = dig(key[c], id[c]);
    }
    // lanes holding the same digit: intersect, bit by bit, the ballots of "my bit value" (8 ballots instead of
    // 64 readlane / compare steps)
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < RDX_BITS; ++bit) {
      const bool one = (dg >> bit) & 1;
      const uint64_t bal = __ballot(one);
      peers &= one ? bal : ~bal;
    }
    const int lower = __popcll(peers & ((1ull << lane) - 1ull));
    const int same = __popcll(peers);
    int start = 0;
    if (valid) start = wcnt[w][dg];
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    if (valid && lower == same - 1) wcnt[w][dg] = (uint16_t)(start + same);
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    pos[c] = valid ? ((dg << 16) | (start + lower)) : -1;
  }
  __syncthreads();
  // digit d: thread d turns the per-wave counts into per-wave bases inside the block's digit-ordered layout
  {
    int cw[RANK_WAVES], tot = 0;
#pragma unroll
    for (int v = 0; v < RANK_WAVES; ++v) {
      cw[v] = wcnt[v][tid];
      tot += cw[v];
    }
    // exclusive scan of tot over the 256 digits (4 wavefronts of 64)
    int inc = tot;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(inc, off);
      if (lane >= off) inc += o;
    }
    __shared__ int wsum[RANK_WAVES];
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    int base = inc - tot;
    for (int v = 0; v < w; ++v) base += wsum[v];
    lstart[tid] = base;
#pragma unroll
    for (int v = 0; v < RANK_WAVES; ++v) {
      wcnt[v][tid] = (uint16_t)base;
      base += cw[v];
    }
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < RANK_CHUNKS; ++c) {
    if (pos[c] >= 0) {
      const int p = wcnt[w][pos[c] >> 16] + (pos[c] & 0xFFFF);
      sk[p] = key[c];
      si[p] = id[c];
    }
  }
  __syncthreads();
  const int cnt = (int)(r1 - r0);
  for (int p = tid; p < cnt; p += 256) {
    const uint64_t k = sk[p];
    const int32_t v = si[p];
    const int d = dig(k, v);
    const int64_t dst = (int64_t)goff[d] + (p - lstart[d]);
    keys_out[dst] = k;
    idx_out[dst] = v;
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
  hipLaunchKernelGGL((sp_radix_rank_kernel<D>), dim3(nblk), dim3(256), 0, st, ws.keys[cur], ws.idx[cur], n, dig, nblk,
                     ws.hist, ws.keys[1 - cur], ws.idx[1 - cur]);
  SP_CHECK_LAUNCH();
  return 0;
}

}  // namespace

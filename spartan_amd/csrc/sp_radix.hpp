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
  template <typename K>
  __device__ __forceinline__ int operator()(K key, int32_t) const { return (int)((key >> shift) & (RDX - 1)); }
};

struct DigitOfRow {                      // byte `shift / 8` of the row (payload / cols) of a flat position
  int shift;
  uint32_t cols;
  template <typename K>
  __device__ __forceinline__ int operator()(K, int32_t idx) const {
    return (int)((((uint32_t)idx / cols) >> shift) & (RDX - 1));
  }
};

// The keys form segments of seg_len (one segment = the whole array for a plain sort; one LINE of a tile for a
// segmented sort), each cut into bpr key blocks of SORT_RB.  hist[(seg * RDX + d) * bpr + bi] = number of keys with
// digit d in block bi of segment seg: ONE exclusive scan over that layout yields destinations ordered by segment,
// then digit, then block -- every segment is sorted in place, independently, by the same launches.
__device__ __forceinline__ void sp_radix_block_range(int b, int64_t n, int64_t seg_len, int bpr, int64_t* r0,
                                                     int64_t* r1, int64_t* hbase) {
  const int64_t seg = b / bpr;
  const int bi = b - (int)(seg * bpr);
  const int64_t s0 = seg * seg_len;
  int64_t s1 = s0 + seg_len;
  if (s1 > n) s1 = n;
  *r0 = s0 + (int64_t)bi * SORT_RB;
  *r1 = *r0 + SORT_RB < s1 ? *r0 + SORT_RB : s1;
  *hbase = seg * RDX * (int64_t)bpr + bi;      // + d * bpr
}

// One workgroup counts HIST_GROUP consecutive key blocks (one LDS histogram each) and then stores, for every digit,
// the counters of its blocks side by side: with one block per workgroup the table was written as 4-byte stores
// strided by the block count -- 1 GB of HBM writes for a 134 MB table (rocprofv3 WRITE_SIZE).
constexpr int HIST_GROUP = 16;

template <typename K, typename D>
__global__ __launch_bounds__(256) void sp_radix_hist_kernel(const K* __restrict__ keys,
                                                            const int32_t* __restrict__ idx, int64_t n, D dig,
                                                            int64_t seg_len, int bpr, int64_t nblk, int group,
                                                            int* __restrict__ hist) {
  __shared__ int lh[HIST_GROUP][RDX];
  const int tid = threadIdx.x;
  for (int i = tid; i < group * RDX; i += 256) (&lh[0][0])[i] = 0;
  __syncthreads();
  const int64_t b0 = (int64_t)blockIdx.x * group;
  for (int k = 0; k < group && b0 + k < nblk; ++k) {
    int64_t r0, r1, hbase;
    sp_radix_block_range((int)(b0 + k), n, seg_len, bpr, &r0, &r1, &hbase);
    for (int64_t i = r0 + tid; i < r1; i += 256) atomicAdd(&lh[k][dig(keys[i], idx[i])], 1);
  }
  __syncthreads();
  // lane (d % 16, k): 16 digits x 16 blocks per store instruction, a digit's 16 counters contiguous in memory when the
  // blocks belong to one segment
  const int k = tid & (group - 1);
  if (b0 + k < nblk) {
    int64_t r0, r1, hbase;
    sp_radix_block_range((int)(b0 + k), n, seg_len, bpr, &r0, &r1, &hbase);
    for (int d = tid / group; d < RDX; d += 256 / group) hist[hbase + (int64_t)d * bpr] = lh[k][d];
  }
}

// One workgroup (4 wavefronts) per key block.  Wavefront w walks its quarter of the block in order, 64 keys at a
// time: the rank of a key among the keys of the same digit in its 64-key chunk is the number of LOWER lanes with
// that digit, a per-wave LDS counter carries the count from chunk to chunk.  The block then places its keys in LDS in digit order and
// writes every digit's run to its global position as one contiguous piece: the scattered 8 + 4 byte stores of a
// direct scatter cost 4x their bytes in HBM write traffic (rocprofv3 WRITE_SIZE), runs of whole cache lines do not.
constexpr int RANK_WAVES = 4;
constexpr int RANK_PER_WAVE = SORT_RB / RANK_WAVES;     // 1024 keys
constexpr int RANK_CHUNKS = RANK_PER_WAVE / 64;         // 16 chunks per wavefront

template <typename K, typename D>
__global__ __launch_bounds__(256) void sp_radix_rank_kernel(const K* __restrict__ keys,
                                                           const int32_t* __restrict__ idx, int64_t n, D dig,
                                                           int64_t seg_len, int bpr, const int* __restrict__ offs,
                                                           K* __restrict__ keys_out,
                                                           int32_t* __restrict__ idx_out) {
  __shared__ K sk[SORT_RB];
  __shared__ int32_t si[SORT_RB];
  __shared__ uint16_t wcnt[RANK_WAVES][RDX];   // keys of digit d seen so far by wavefront w; then its base
  __shared__ int lstart[RDX];                  // first LDS slot of digit d
  __shared__ int goff[RDX];                    // global position of this block's first key of digit d
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  int64_t r0, r1, hbase;
  sp_radix_block_range(blockIdx.x, n, seg_len, bpr, &r0, &r1, &hbase);
  for (int i = tid; i < RANK_WAVES * RDX; i += 256) (&wcnt[0][0])[i] = 0;
  goff[tid] = offs[hbase + (int64_t)tid * bpr];
  __syncthreads();
  K key[RANK_CHUNKS];
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
    const K k = sk[p];
    const int32_t v = si[p];
    const int d = dig(k, v);
    const int64_t dst = (int64_t)goff[d] + (p - lstart[d]);
    keys_out[dst] = k;
    idx_out[dst] = v;
  }
}

inline size_t sp_al256(size_t v) { return (v + 255) & ~(size_t)255; }

// scratch of a sort of n pairs: two key arrays, two payload arrays, the histogram (reusable as n ints), scan sums
template <typename K>
struct SortWsT {
  K* keys[2];
  int32_t* idx[2];
  int* hist;   // [RDX][nblk] or n ints, whichever is larger
  int* sums;   // scan chunk sums
  int* total;
};

using SortWs = SortWsT<uint64_t>;

// key blocks of a sort of n keys in segments of seg_len
inline int64_t sp_sort_blocks(int64_t n, int64_t seg_len) {
  if (n < 1) return 0;
  if (seg_len < 1 || seg_len > n) seg_len = n;
  const int64_t nseg = (n + seg_len - 1) / seg_len;
  return nseg * ((seg_len + SORT_RB - 1) / SORT_RB);
}

template <typename K>
inline size_t sp_sort_ws_bytes(int64_t n, SortWsT<K>* ws, char* base, int64_t seg_len = 0) {
  const int64_t nblk = sp_sort_blocks(n, seg_len);
  const int64_t hist_words = (int64_t)RDX * nblk > n ? (int64_t)RDX * nblk : n;
  const int64_t sums_words = (hist_words + SCAN_CHUNK - 1) / SCAN_CHUNK + 1;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* p = base ? base + off : nullptr;
    off += sp_al256(bytes);
    return p;
  };
  char* k0 = take((size_t)n * sizeof(K));
  char* k1 = take((size_t)n * sizeof(K));
  char* i0 = take((size_t)n * 4);
  char* i1 = take((size_t)n * 4);
  char* h = take((size_t)hist_words * 4);
  char* s = take((size_t)sums_words * 4);
  char* t = take(256);
  if (ws) {
    ws->keys[0] = (K*)k0;
    ws->keys[1] = (K*)k1;
    ws->idx[0] = (int32_t*)i0;
    ws->idx[1] = (int32_t*)i1;
    ws->hist = (int*)h;
    ws->sums = (int*)s;
    ws->total = (int*)t;
  }
  return off;
}

// keys[cur] / idx[cur] -> keys[1 - cur] / idx[1 - cur], stable by the digit `dig`; with seg_len > 0 every segment of
// seg_len keys is sorted on its own (the workspace must have been sized with the same seg_len)
template <typename K, typename D>
static inline int sp_radix_pass(SortWsT<K>& ws, int cur, int64_t n, D dig, hipStream_t st, int64_t seg_len = 0) {
  if (seg_len < 1 || seg_len > n) seg_len = n;
  const int bpr = (int)((seg_len + SORT_RB - 1) / SORT_RB);
  const int64_t nblk = sp_sort_blocks(n, seg_len);
  if (nblk > 2147483647LL / 2) SP_FAIL("radix sort: too many key blocks");
  // (grouping only once there are enough key blocks to fill the chip with groups)
  const int group = nblk >= (int64_t)HIST_GROUP * SP_CUS * SP_BLOCKS_PER_CU ? HIST_GROUP : 1;
  hipLaunchKernelGGL((sp_radix_hist_kernel<K, D>), dim3((unsigned)((nblk + group - 1) / group)), dim3(256), 0, st,
                     ws.keys[cur], ws.idx[cur], n, dig, seg_len, bpr, nblk, group, ws.hist);
  SP_CHECK_LAUNCH();
  if (sp_exscan_int(ws.hist, (int64_t)RDX * nblk, ws.sums, nullptr, st)) return 1;
  hipLaunchKernelGGL((sp_radix_rank_kernel<K, D>), dim3((unsigned)nblk), dim3(256), 0, st, ws.keys[cur], ws.idx[cur], n, dig,
                     seg_len, bpr, ws.hist, ws.keys[1 - cur], ws.idx[1 - cur]);
  SP_CHECK_LAUNCH();
  return 0;
}

}  // namespace

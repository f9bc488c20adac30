// Column-blocked sparse matrix x vector (fp32): y = A . x with the slices of x a row block uses heavily staged
// through LDS.  Replaces `tile.dot(vector)` of a sparse tile on the multiply of the reference's PageRank
// (spartan/examples/pagerank.py via array/sparse.pyx:103-158 / scipy csr_matvec) where sp_csr_spmv_planned_kernel
// (sparse.hip) is bound by the random gather of x: every gathered 4-byte value moves a 128-byte line from the L2 to
// the CU (9 M lines per launch on 900 000 pages x 10 links, 29 of its 52 us).
//
// The plan (built once per tile, tiles are immutable) cuts the rows into blocks of RB and the columns into slices
// of S = 11 264 values (44 KB).  Per row block, a slice holding >= BSP_STAGE_MIN of the block's entries becomes a
// STAGED segment; each run of the other slices between them becomes one DIRECT segment.  The block's entries are
// stably partitioned by segment -- inside a segment they keep CSR order, (row, column) ascending -- and stored as
// (value, key): key = row-in-block << 16 | column-in-slice for a staged segment, the column itself (and the row
// in a 16-bit side array) for a direct one.
//
// The kernel gives a row block to one workgroup: for each segment in column order it copies the slice of x into
// LDS (one coalesced read; the slice for the NEXT segment is already on its way in registers), multiplies the
// segment's entries against it 2048 at a time (entries prefetched two chunks ahead, the direct segments' gathers
// of x one chunk ahead; a lane owns four consecutive entries), leaves the products in LDS, and the lane holding the
// first entry of a run of equal rows adds the run to the row's accumulator, also in LDS (its own entries from
// registers and without branches, what continues in later lanes' entries from LDS).  A row's entries therefore meet its accumulator in ascending column
// order, segment after segment -- CSR storage order for a tile with sorted rows, which is the order csr_matvec
// adds in: results are bit-identical to sp_csr_spmv_planned_kernel's.  Rows that are not sorted by column (the
// builder checks) make the plan invalid and the caller keeps the stream kernel.
#include "sp_common.hpp"

namespace {

// Geometry (round 4): ONE workgroup of 1024 lanes per CU owning up to 4096 rows.  A staged slice then serves twice
// the entries it served with two 512-lane workgroups of 2048 rows per CU (round 3), so the slice copies -- every row
// block of a site pulls the site's whole x through the L2 -- and the barriers around them halve per entry: 42.2 ->
// 37.4 us on the 900 000-page tile; the row of a direct entry packed into its key: 35.8 us.
// Round 6: the loop's loads really run ahead now (see the kernel: three entry sets, two gather sets and two slice sets
// taking turns, nothing moved between them; two LDS slice buffers, so that a segment's slice is stored while the chunk
// before is multiplied and needs no barrier of its own).  Timed with parts removed (-DBSP_ABLATE, one box, us):
//   whole kernel 33.4 | no chunk loop at all 7.7 | loop without products and run sums 23.7 (27.1 before this round's
//   pipeline) | ... and without its barriers 22.3 | loop without barriers only 28.9
// i.e. 7.7 fixed + 16 for the loop's loads (72 MB of entries, 113 MB of slices through the L2, the direct gathers:
// ~60 % of what a CU's 64 B/clk vector-memory path moves) + 9.7 for products and run sums, which do NOT overlap the
// loads: all sixteen waves of a CU walk the same phase between the same barriers.  35.2 -> 32.2 us with the requests
// issued around the run sums.
constexpr int BSP_THREADS = 1024;
constexpr int BSP_WGS_PER_CU = 1;
constexpr int BSP_CAP = 4 * BSP_THREADS;  // products per pass: 4 per lane
constexpr int BSP_PER = BSP_CAP / BSP_THREADS;
constexpr int BSP_MAXSEG = 64;
constexpr int BSP_XS_BYTES = 44 * 1024;
constexpr int BSP_S = BSP_XS_BYTES / 4;   // columns per slice
constexpr int BSP_MAX_RB = 4096;         // rows per block (16-bit row ids; 4 B of accumulator each)
constexpr int BSP_STAGE_MIN = 768;        // a 44 KB slice is 352 lines: staging pays from about twice as many gathers
constexpr int BSP_MAX_SLICES = 16384;
// direct segments: with at most 2^20 columns the row-in-block (12 bits) rides in the key above the column, and the
// 16-bit row array is never read
constexpr int BSP_PACK_SHIFT = 20;
constexpr int64_t BSP_PACK_COLS = 1LL << BSP_PACK_SHIFT;
constexpr int BSP_XV_N = (BSP_XS_BYTES / 16 + BSP_THREADS - 1) / BSP_THREADS;

struct BspSeg {
  int32_t col_lo, width, count, staged;
};

inline size_t sp_al256(size_t b) { return (b + 255) & ~(size_t)255; }

struct BspLayout {
  int64_t rb, nb, ns;
  size_t nseg_off, seg_off, key_off, drow_off, val_off, total;
};

bool bsp_layout(int32_t dtype, int64_t m, int64_t k, int64_t nnz, BspLayout* L) {
  if (dtype != SP_F32 || nnz < (1 << 18) || m < 4096 || k < 4 || nnz > 64 * m || k > 2147483647LL) return false;
  const int64_t ns = (k + BSP_S - 1) / BSP_S;
  if (ns > BSP_MAX_SLICES) return false;
  int64_t j = 1;
  while ((m + BSP_WGS_PER_CU * SP_CUS * j - 1) / (BSP_WGS_PER_CU * SP_CUS * j) > BSP_MAX_RB) ++j;
  L->rb = (m + BSP_WGS_PER_CU * SP_CUS * j - 1) / (BSP_WGS_PER_CU * SP_CUS * j);
  L->nb = (m + L->rb - 1) / L->rb;
  L->ns = ns;
  size_t at = 256;                                   // header: 8 x int64
  L->nseg_off = at;
  at += sp_al256((size_t)L->nb * 4);
  L->seg_off = at;
  at += sp_al256((size_t)L->nb * BSP_MAXSEG * sizeof(BspSeg));
  L->key_off = at;
  at += sp_al256((size_t)nnz * 4);
  L->drow_off = at;
  at += sp_al256((size_t)nnz * 2);
  L->val_off = at;
  at += sp_al256((size_t)nnz * 4);
  L->total = at;
  return true;
}

__device__ __forceinline__ int bsp_block_exscan(int v, int* wsum) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(inc, off);
    if (lane >= off) inc += o;
  }
  __syncthreads();                 // wsum may still be read from the previous call
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  int woff = 0;
  for (int i = 0; i < w; ++i) woff += wsum[i];
  return woff + inc - v;
}

// One workgroup per row block: slice histogram, segment table, sortedness check, stable partition.
__global__ __launch_bounds__(BSP_THREADS) void sp_bsp_build_kernel(const int64_t* __restrict__ indptr,
                                                                   const int32_t* __restrict__ indices,
                                                                   const float* __restrict__ vals, int64_t m, int64_t k,
                                                                   int rb, int ns, long long* __restrict__ header,
                                                                   int* __restrict__ nseg_out, BspSeg* __restrict__ segtab,
                                                                   uint32_t* __restrict__ keys, uint16_t* __restrict__ drow,
                                                                   float* __restrict__ pvals) {
  extern __shared__ int slice_cnt[];      // [ns]
  __shared__ BspSeg segs[BSP_MAXSEG];
  __shared__ int seg_off[BSP_MAXSEG + 1];
  __shared__ int wsum[BSP_THREADS / 64];
  __shared__ int s_nseg, s_bad;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int64_t r0 = (int64_t)b * rb, r1 = r0 + rb < m ? r0 + rb : m;
  const int64_t e0 = indptr[r0], e1 = indptr[r1];
  for (int i = tid; i < ns; i += BSP_THREADS) slice_cnt[i] = 0;
  if (tid == 0) s_bad = (e1 - e0 > 2147483647LL) ? 1 : 0;
  __syncthreads();
  for (int64_t e = e0 + tid; e < e1; e += BSP_THREADS) {
    const int c = indices[e];
    if (c < 0 || c >= k) s_bad = 1;
    else atomicAdd(&slice_cnt[c / BSP_S], 1);
  }
  // my rows (contiguous, so that thread order is row order) -- and are their columns ascending?
  const int q = (rb + BSP_THREADS - 1) / BSP_THREADS;      // <= 4
  const int64_t ra = r0 + (int64_t)tid * q;
  int64_t cursor[BSP_MAX_RB / BSP_THREADS], rend[BSP_MAX_RB / BSP_THREADS];
#pragma unroll
  for (int j = 0; j < BSP_MAX_RB / BSP_THREADS; ++j) {
    const int64_t r = ra + j;
    if (j < q && r < r1) {
      cursor[j] = indptr[r];
      rend[j] = indptr[r + 1];
      for (int64_t e = cursor[j] + 1; e < rend[j]; ++e)
        if (indices[e] < indices[e - 1]) s_bad = 1;
    } else {
      cursor[j] = rend[j] = 0;
    }
  }
  __syncthreads();
  if (s_bad) {
    if (tid == 0) {
      atomicExch((unsigned long long*)&header[0], 0ull);
      nseg_out[b] = 0;
    }
    return;
  }
  if (tid == 0) {
    int n = 0, open = -1;
    bool overflow = false;
    for (int s = 0; s < ns && !overflow; ++s) {
      const int c = slice_cnt[s];
      // (the matrix's last, narrower slice is staged only if 16-byte loads of it stay inside x)
      const bool may_stage = (int64_t)(s + 1) * BSP_S <= k || (k & 3) == 0;
      if (c >= BSP_STAGE_MIN && may_stage) {
        if (open >= 0) {
          segs[open].width = s * BSP_S - segs[open].col_lo;
          open = -1;
        }
        if (n == BSP_MAXSEG) overflow = true;
        else {
          const int64_t w = k - (int64_t)s * BSP_S;
          segs[n++] = BspSeg{s * BSP_S, (int)(w < BSP_S ? w : BSP_S), c, 1};
        }
      } else if (c > 0 || open >= 0) {
        if (open < 0) {
          if (n == BSP_MAXSEG) overflow = true;
          else {
            open = n;
            segs[n++] = BspSeg{s * BSP_S, 0, 0, 0};
          }
        }
        if (!overflow) segs[open].count += c;
      }
    }
    if (open >= 0 && !overflow) segs[open].width = (int)(k - segs[open].col_lo);
    if (overflow) {       // too many segments: everything through the direct path (always valid)
      n = 1;
      segs[0] = BspSeg{0, (int)k, (int)(e1 - e0), 0};
    }
    int run = 0;
    for (int s = 0; s < n; ++s) {
      seg_off[s] = run;
      run += segs[s].count;
    }
    seg_off[n] = run;
    s_nseg = n;
    nseg_out[b] = n;
  }
  __syncthreads();
  const int n = s_nseg;
  for (int i = tid; i < n * 4; i += BSP_THREADS) ((int*)(segtab + (int64_t)b * BSP_MAXSEG))[i] = ((const int*)segs)[i];
  for (int s = 0; s < n; ++s) {
    const int64_t col_hi = (int64_t)segs[s].col_lo + segs[s].width;
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < BSP_MAX_RB / BSP_THREADS; ++j)
      for (int64_t e = cursor[j]; e < rend[j] && indices[e] < col_hi; ++e) ++cnt;
    int64_t pos = e0 + seg_off[s] + bsp_block_exscan(cnt, wsum);
    const bool staged = segs[s].staged != 0;
    const bool packed = k <= BSP_PACK_COLS;
    const int col_lo = segs[s].col_lo;
#pragma unroll
    for (int j = 0; j < BSP_MAX_RB / BSP_THREADS; ++j) {
      int64_t e = cursor[j];
      const uint32_t rl = (uint32_t)(tid * q + j);
      for (; e < rend[j] && indices[e] < col_hi; ++e, ++pos) {
        const int c = indices[e];
        keys[pos] = staged ? (rl << 16) | (uint32_t)(c - col_lo) : (packed ? (rl << BSP_PACK_SHIFT) | (uint32_t)c : (uint32_t)c);
        drow[pos] = (uint16_t)rl;
        pvals[pos] = vals[e];
      }
      cursor[j] = e;
    }
  }
}

typedef float bsp_f4 __attribute__((ext_vector_type(4)));
#ifndef BSP_ABLATE
#define BSP_ABLATE 0    // timing-only builds (wrong results): 1 no chunk loop, 2 no run sums, 4 no products, 8 no barriers in the loop
#endif

// (second launch bound: waves per SIMD -- two workgroups of 8 waves per CU)
template <bool PACKED>
__global__ __launch_bounds__(BSP_THREADS, 4 / BSP_WGS_PER_CU * BSP_WGS_PER_CU) void sp_bsp_spmv_kernel(const int64_t* __restrict__ indptr, int64_t m, int rb,
                                                                     int nb, const int* __restrict__ nseg,
                                                                     const BspSeg* __restrict__ segtab,
                                                                     const uint32_t* __restrict__ keys,
                                                                     const uint16_t* __restrict__ drow,
                                                                     const float* __restrict__ pvals,
                                                                     const float* __restrict__ x, float* __restrict__ y,
                                                                     int64_t ldy, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  float* xs = (float*)lds;                                  // [2][BSP_S]: the slice in use and the one before / after it
  float* acc = xs + 2 * BSP_S;                              // [BSP_MAX_RB]
  float* prod = acc + BSP_MAX_RB + 4;                       // [2][BSP_CAP]   (acc[BSP_MAX_RB]: the spare slot of the run sums)
  uint16_t* rid = (uint16_t*)(prod + 2 * BSP_CAP);          // [2][BSP_CAP]
  __shared__ BspSeg segs[BSP_MAXSEG];
  // XCD k (blockIdx & 7) walks a contiguous range of row blocks: neighbours share their slices of x in its L2
  const int per = (nb + 7) >> 3;
  const int b = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  if (b >= nb) return;
  const int tid = threadIdx.x;
  const int64_t r0 = (int64_t)b * rb, r1 = r0 + rb < m ? r0 + rb : m;
  const int nrows = (int)(r1 - r0);
  // (the whole row of the segment table, whatever the block's count: one round trip for the count, the table and
  //  the block's first entry together instead of two)
  static_assert(BSP_MAXSEG * 4 <= BSP_THREADS, "one dword of the segment table per lane");
  if (tid < BSP_MAXSEG * 4) ((int*)segs)[tid] = ((const int*)(segtab + (int64_t)b * BSP_MAXSEG))[tid];
  const int n = nseg[b];
  const int64_t e0 = indptr[r0];
  for (int i = tid; i < nrows; i += BSP_THREADS) acc[i] = 0.f;
  __syncthreads();

  // the chunk sequence: segment by segment, BSP_CAP entries at a time; entries are stored in that order from e0 on.
  // A lane owns BSP_PER CONSECUTIVE entries of a chunk (one 16-byte load each of keys and values).
  struct Chunk {
    int seg, cnt, staged, before;      // before: entries of the block in the segments before `seg`
    int64_t e;
    bool fresh;                        // first chunk of a staged segment: the one that needs a new slice of x
  };
  auto next_chunk = [&](const Chunk& c) {       // the chunk after c (seg == n: none)
    Chunk o;
    const int used = (int)(c.e - e0) + c.cnt;
    o.seg = c.seg;
    o.before = c.before;
    if (o.seg < n && used >= o.before + segs[o.seg].count) {
      o.before += segs[o.seg].count;
      ++o.seg;
    }
    o.e = e0 + used;
    o.cnt = 0;
    o.staged = 0;
    if (o.seg < n) {
      const int left = o.before + segs[o.seg].count - used;
      o.cnt = left < BSP_CAP ? left : BSP_CAP;
      o.staged = segs[o.seg].staged;
    }
    o.fresh = o.staged && used == o.before;
    return o;
  };
  Chunk c0;
  c0.seg = 0;
  c0.before = 0;
  c0.e = e0;
  c0.cnt = n > 0 ? (segs[0].count < BSP_CAP ? segs[0].count : BSP_CAP) : 0;
  c0.staged = n > 0 ? segs[0].staged : 0;
  c0.fresh = c0.staged != 0;

  typedef uint32_t u32x4 __attribute__((ext_vector_type(4), aligned(4)));
  typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
  typedef uint16_t u16x4 __attribute__((ext_vector_type(4)));
  static_assert(BSP_PER == 4, "one 16-byte load per lane and chunk");
  const int i0 = tid * BSP_PER;
  // THREE register sets of entries (keys, values, rows) and TWO of direct gathers that take turns: chunk j is
  // multiplied from set j % 3 and gather set j % 2 while the gathers of chunk j + 2 and the entries of chunk j + 3 are
  // requested -- every load has at least one whole iteration (a barrier, a pass over LDS) between its request and
  // its first use.  (Round 4's pipeline asked for the gathers of chunk j + 1 while chunk j was multiplied and rotated
  // the sets with register moves at the loop's latch: the moves waited with vmcnt(0) for the loads just issued, and
  // without them -- two sets taking turns, round 6 -- the wait for the gathers, requested one short phase earlier,
  // took their place: one memory latency per chunk either way, 36 us of which 24 were that.)
  struct Entries {
    u32x4 k;
    bsp_f4 v;
    u16x4 r;
  };
  Entries E0 = {{0, 0, 0, 0}, {0.f, 0.f, 0.f, 0.f}, {0, 0, 0, 0}}, E1 = E0, E2 = E0;
  bsp_f4 X0 = {0.f, 0.f, 0.f, 0.f}, X1 = X0;
  // EVERY load of the loop below is issued by every lane in every iteration, with its address clamped into valid
  // memory where the lane has nothing to fetch -- no load sits under a condition.  That is what lets the compiler
  // count the loads in flight and wait for exactly the ones an instruction needs (s_waitcnt vmcnt(N > 0)): with
  // conditional loads (round 3) it could not, and every wait was vmcnt(0).
  const int64_t e_first = e0;
  auto load_entries = [&](const Chunk& c, Entries& E) {
    u32x4& kk = E.k;
    bsp_f4& vv = E.v;
    u16x4& rr = E.r;
    // (a lane whose entries start inside the chunk loads all four: what lies behind the chunk's end is the next
    // chunk's, or the padding of the plan arrays, and is never used; a lane past the chunk re-reads its start)
    const int64_t at = c.seg < n ? c.e + (i0 < c.cnt ? i0 : 0) : e_first;
    kk = *(const u32x4*)(keys + at);
    vv = (bsp_f4)(*(const f32x4u*)(pvals + at));
    if constexpr (!PACKED) {
#pragma unroll
      for (int u = 0; u < BSP_PER; ++u) rr[u] = drow[at + u];
    }
  };
  auto gather_direct = [&](const Chunk& c, const u32x4& kk, bsp_f4& xx) {
    const bool live = c.seg < n && !c.staged;
#pragma unroll
    for (int u = 0; u < BSP_PER; ++u) {
      const uint32_t col = PACKED ? (kk[u] & (uint32_t)(BSP_PACK_COLS - 1)) : kk[u];
      xx[u] = x[(live && i0 + u < c.cnt) ? col : 0u];
    }
  };
  // slices of x travel global -> registers -> LDS, requested while the chunk before the segment's first is being
  // multiplied and stored into the OTHER of two LDS buffers when that chunk comes up.  16-byte loads; a vector past
  // the slice's end -- every vector, when no slice is due (`real` false) -- reads the start of x instead: one line
  // per wave (the plan stages a narrower last slice only when k is a multiple of 4: no vector straddles the end of x).
  bsp_f4 xr0[BSP_XV_N];
  auto prefetch_slice = [&](int s, bool real, bsp_f4* xr) {
    const int lo = real ? segs[s].col_lo : 0, w = real ? segs[s].width : 0;
    const float* __restrict__ xl = x + lo;
#pragma unroll
    for (int u = 0; u < BSP_XV_N; ++u) {
      const int i = (u * BSP_THREADS + tid) * 4;
      xr[u] = *(const bsp_f4*)(i < w ? xl + i : x);
    }
  };

  // pipeline fill: the slice of chunk 0 into LDS, chunks 0 .. 2 into the three sets, the gathers of chunks 0 and 1,
  // the slices chunks 1 and 2 may start into the two slice sets
  Chunk cur = c0, c1 = next_chunk(c0), c2 = next_chunk(c1), c3 = next_chunk(c2);
  int par = 0;
  bsp_f4 xr1[BSP_XV_N];
  load_entries(cur, E0);
  load_entries(c1, E1);
  load_entries(c2, E2);
  prefetch_slice(cur.seg, cur.fresh, xr0);
  gather_direct(cur, E0.k, X0);
  gather_direct(c1, E1.k, X1);
  if (cur.fresh) {
#pragma unroll
    for (int u = 0; u < BSP_XV_N; ++u) {
      const int i = (u * BSP_THREADS + tid) * 4;
      if (i < BSP_S) *(bsp_f4*)(xs + i) = xr0[u];
    }
  }
  prefetch_slice(c1.seg, c1.fresh, xr1);
  prefetch_slice(c2.seg, c2.fresh, xr0);
  __syncthreads();
  int buf = 0;
  // One chunk.  A: the set with the chunk's entries (reloaded with chunk j + 3's), C: the set with chunk j + 2's (the
  // keys of its gathers), X: the chunk's direct gathers (reloaded with chunk j + 2's), XR: the slice chunk j + 1 may
  // start (stored to LDS here, reloaded with the one chunk j + 3 may start).  Every global load of the loop is issued
  // in every iteration, in the order slice / gathers / entries (a slice request that is not due reads x[0..3]), so
  // the waits the compiler places are counts of younger loads -- and every load is requested two or three chunks
  // before its first use.  products | slice store | barrier | slice request | run sums | gathers, entries.
  auto step = [&](Entries& A, const Entries& C, bsp_f4& X, bsp_f4* XR) {
    // (a fresh chunk's slice was stored into the other buffer while the chunk before was multiplied, ahead of that
    //  iteration's barrier; chunk 0's by the fill above, into buffer 0)
    if (cur.fresh && cur.e != e0) par ^= 1;
    const float* xc = xs + par * BSP_S;
    float* pb = prod + buf * BSP_CAP;
    uint16_t* rbuf = rid + buf * BSP_CAP;
    bsp_f4 p = {0.f, 0.f, 0.f, 0.f};
    u16x4 r = {0, 0, 0, 0};
    if (BSP_ABLATE & 4) {
      asm volatile("" ::"v"(A.k), "v"(A.v), "v"(X), "v"(A.r));
    } else if (i0 < cur.cnt) {
#pragma unroll
      for (int u = 0; u < BSP_PER; ++u) {
        const float xv = cur.staged ? xc[A.k[u] & 0xffffu] : X[u];
        p[u] = A.v[u] * xv;
        r[u] = cur.staged ? (uint16_t)(A.k[u] >> 16) : (PACKED ? (uint16_t)(A.k[u] >> BSP_PACK_SHIFT) : A.r[u]);
      }
      *(bsp_f4*)(pb + i0) = p;       // (for the lanes before this one: a run of theirs may go on in these entries)
      *(u16x4*)(rbuf + i0) = r;
    }
    // the slice the NEXT chunk starts (requested two iterations ago) goes to the buffer nobody reads: the slice before
    // the current one was last read before the barrier of the iteration that switched away from it
    if (c1.fresh) {
      float* xw = xs + (par ^ 1) * BSP_S;
#pragma unroll
      for (int u = 0; u < BSP_XV_N; ++u) {
        const int i = (u * BSP_THREADS + tid) * 4;
        if (i < BSP_S) *(bsp_f4*)(xw + i) = XR[u];
      }
    }
    const Chunk done = cur;
    cur = c1;
    c1 = c2;
    c2 = c3;
    c3 = next_chunk(c3);
    if (!(BSP_ABLATE & 8)) __syncthreads();
    // the iteration's requests, around the run sums rather than in one burst ahead of the barrier (33.0 -> 32.2 us:
    // the vector-memory path works on the slice while the waves are busy in LDS)
    prefetch_slice(c2.seg, c2.fresh, XR);
    __builtin_amdgcn_sched_barrier(0);                   // products of `done` (and the next slice) are in LDS; everybody is past the runs before
    // Runs of equal rows: the lane holding a run's first entry adds the run, in order, to the row's accumulator --
    // its own entries from registers, what continues in later lanes' entries from LDS.
    if (!(BSP_ABLATE & 2) && i0 < done.cnt) {
      // (the lane's own products and rows are still in its registers; LDS holds them for the other lanes)
      const int nvalid = done.cnt - i0;                       // >= 1; entries u >= nvalid are not this chunk's
      const unsigned prev = i0 == 0 ? 0xffffffffu : (unsigned)rbuf[i0 - 1];
      // straight-line over the lane's four entries: selects instead of branches (a row's accumulator is read for
      // every entry and used where a run starts; a run that ends inside the lane is stored, one that does not --
      // or entries that belong to an earlier lane's run -- goes to the spare slot behind the accumulators)
      float a4[BSP_PER];
#pragma unroll
      for (int u = 0; u < BSP_PER; ++u) a4[u] = acc[u < nvalid ? r[u] : r[0]];
      bool active = false;
      float sum = 0.f;
      unsigned row = 0;
#pragma unroll
      for (int u = 0; u < BSP_PER; ++u) {
        const bool valid = u < nvalid;
        const bool start = valid && (u == 0 ? (unsigned)r[0] != prev : r[u] != r[u - 1]);
        sum = start ? a4[u] : sum;
        row = start ? (unsigned)r[u] : row;
        active = active || start;
        sum = (valid && active) ? sum + p[u] : sum;
        if (u + 1 < BSP_PER) {
          // the run ends here if the next entry of the lane starts another one
          const bool ends = active && u + 1 < nvalid && r[u + 1] != r[u];
          acc[ends ? row : (unsigned)BSP_MAX_RB] = sum;
        }
      }
      if (active) {      // the lane's last run: it may go on in later lanes' entries
        for (int j = i0 + (nvalid < BSP_PER ? nvalid : BSP_PER); j < done.cnt && rbuf[j] == row; ++j) sum += pb[j];
        acc[row] = sum;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    gather_direct(c1, C.k, X);
    load_entries(c2, A);
    buf ^= 1;
  };
  // chunk j: entries in set j % 3, gathers in set j % 2; the six combinations in turn, no register ever moves
  for (;;) {
    if (BSP_ABLATE & 1) break;
    if (!(cur.seg < n)) break;
    step(E0, E2, X0, xr1);
    if (!(cur.seg < n)) break;
    step(E1, E0, X1, xr0);
    if (!(cur.seg < n)) break;
    step(E2, E1, X0, xr1);
    if (!(cur.seg < n)) break;
    step(E0, E2, X1, xr0);
    if (!(cur.seg < n)) break;
    step(E1, E0, X0, xr1);
    if (!(cur.seg < n)) break;
    step(E2, E1, X1, xr0);
  }
  __syncthreads();
  for (int i = tid; i < nrows; i += BSP_THREADS) {
    float* p = y + (r0 + i) * ldy;
    *p = accumulate ? *p + acc[i] : acc[i];
  }
}

}  // namespace

extern "C" size_t sp_csr_spmv_blockplan_bytes(int32_t dtype, int64_t m, int64_t k, int64_t nnz) {
  BspLayout L;
  return bsp_layout(dtype, m, k, nnz, &L) ? L.total : 0;
}

extern "C" int sp_csr_spmv_blockplan(int32_t dtype, int64_t m, int64_t k, int64_t nnz, const int64_t* d_indptr,
                                     const int32_t* d_indices, const void* d_vals, void* d_plan, size_t plan_bytes,
                                     void* stream) {
  BspLayout L;
  if (!bsp_layout(dtype, m, k, nnz, &L)) SP_FAIL("sp_csr_spmv_blockplan: no blocked plan for this matrix (see sp_csr_spmv_blockplan_bytes)");
  if (!d_indptr || !d_indices || !d_vals || !d_plan) SP_FAIL("sp_csr_spmv_blockplan: NULL pointer");
  if (plan_bytes < L.total) SP_FAIL("sp_csr_spmv_blockplan: plan buffer too small (%zu < %zu)", plan_bytes, L.total);
  hipStream_t st = (hipStream_t)stream;
  char* P = (char*)d_plan;
  const long long header[8] = {1, (long long)L.rb, (long long)L.nb, BSP_S, (long long)L.ns, (long long)nnz, (long long)m, (long long)k};
  SP_HIP(hipMemcpyAsync(P, header, sizeof(header), hipMemcpyHostToDevice, st));
  SP_HIP(hipStreamSynchronize(st));      // (header is a stack array)
  static bool attr_set = false;
  if (!attr_set) {
    SP_HIP(hipFuncSetAttribute((const void*)sp_bsp_build_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, BSP_MAX_SLICES * 4));
    attr_set = true;
  }
  hipLaunchKernelGGL(sp_bsp_build_kernel, dim3((unsigned)L.nb), dim3(BSP_THREADS), (size_t)L.ns * 4, st, d_indptr, d_indices,
                     (const float*)d_vals, m, k, (int)L.rb, (int)L.ns, (long long*)P, (int*)(P + L.nseg_off),
                     (BspSeg*)(P + L.seg_off), (uint32_t*)(P + L.key_off), (uint16_t*)(P + L.drow_off),
                     (float*)(P + L.val_off));
  SP_CHECK_LAUNCH();
  return 0;
}

extern "C" int sp_csr_spmv_blocked(int32_t dtype, int64_t m, int64_t k, int64_t nnz, const int64_t* d_indptr,
                                   const void* d_plan, const void* d_x, void* d_y, int64_t ldy, int32_t accumulate,
                                   void* stream) {
  BspLayout L;
  if (!bsp_layout(dtype, m, k, nnz, &L)) SP_FAIL("sp_csr_spmv_blocked: no blocked plan for this matrix");
  if (!d_indptr || !d_plan || !d_x || !d_y) SP_FAIL("sp_csr_spmv_blocked: NULL pointer");
  if (((uintptr_t)d_x & 15) != 0) SP_FAIL("sp_csr_spmv_blocked: x must be 16-byte aligned");
  if (ldy < 1) SP_FAIL("sp_csr_spmv_blocked: bad ldy");
  const char* P = (const char*)d_plan;
  constexpr int lds_bytes = 2 * BSP_XS_BYTES + (BSP_MAX_RB + 4) * 4 + 2 * BSP_CAP * 4 + 2 * BSP_CAP * 2;
  static_assert(lds_bytes + BSP_MAXSEG * (int)sizeof(BspSeg) <= 160 * 1024, "LDS budget of a CU");
  static bool attr_set = false;
  if (!attr_set) {
    SP_HIP(hipFuncSetAttribute((const void*)sp_bsp_spmv_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    SP_HIP(hipFuncSetAttribute((const void*)sp_bsp_spmv_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    attr_set = true;
  }
  const int per = (int)((L.nb + 7) / 8);
  if (k <= BSP_PACK_COLS)
    hipLaunchKernelGGL(sp_bsp_spmv_kernel<true>, dim3((unsigned)(per * 8)), dim3(BSP_THREADS), lds_bytes, (hipStream_t)stream,
                       d_indptr, m, (int)L.rb, (int)L.nb, (const int*)(P + L.nseg_off), (const BspSeg*)(P + L.seg_off),
                       (const uint32_t*)(P + L.key_off), (const uint16_t*)(P + L.drow_off), (const float*)(P + L.val_off),
                       (const float*)d_x, (float*)d_y, ldy, (int)accumulate);
  else
    hipLaunchKernelGGL(sp_bsp_spmv_kernel<false>, dim3((unsigned)(per * 8)), dim3(BSP_THREADS), lds_bytes, (hipStream_t)stream,
                       d_indptr, m, (int)L.rb, (int)L.nb, (const int*)(P + L.nseg_off), (const BspSeg*)(P + L.seg_off),
                       (const uint32_t*)(P + L.key_off), (const uint16_t*)(P + L.drow_off), (const float*)(P + L.val_off),
                       (const float*)d_x, (float*)d_y, ldy, (int)accumulate);
  SP_CHECK_LAUNCH();
  return 0;
}

// Split tier of sp_nearest_center: the same filter as kmeans_mfma.hpp -- score[c][i] = |c|^2/2 - c.x_i with the
// per-point (best, second best) folded into the epilogue, undecided points re-done exactly -- with the contraction on
// the bf16 matrix pipe, which on gfx950 runs at 16 x the rate of the fp32 one (v_mfma_f32_32x32x16_bf16: 16 features
// per 32 cycles; v_mfma_f32_32x32x2_f32: 2 per 64).
//
// Every fp32 operand is cut into two bf16 numbers, v = hi + mid + r with hi = bf16(v), mid = bf16(v - hi) (both
// subtractions exact) and |r| <= 2^-16 |v| (bf16 keeps 8 significant bits: its unit roundoff is 2^-8),
// and the dot product is taken as  x.c ~ sum_k (xh ch + xh cm + xm ch):  three MFMAs per 16 features, whose products
// are EXACT in fp32 (8 x 8 significant bits) and which accumulate in fp32.  With u = 2^-24 and
// S = sum_k |x_k c_k| <= |x| |c|, that differs from x.c by
//   * the terms left out, xm cm + (xh + xm) rc + rx c:           <= 3.0001 * 2^-16 S = 768.1 u S
//   * the accumulation of the 3 D exact products in fp32:          <= 3.03 D u S -- the bound of ANY order of adding
//     them with one rounding to nearest per addend.  What the instruction does inside is not documented; measured
//     (tools/mfma_bf16_probe.hip, 20 000 random operand sets with exponents spread over 30 binades): it is
//     neither a k-ordered chain nor an exact sum rounded once, and errs by at most 2.5 u (|c| + sum|p|) per MFMA of
//     16 products -- a seventh of what this bound grants it
//   * |c|^2/2 rounded once, the final subtraction rounded once     (as in the fp32 kernel)
// and the score |c|^2 - 2 x.c, in whose units km_decide and the candidate window work, by twice that: the error
// factor of this tier is
//   F = 6.1 D + 1550,    E = u (F |x| |c|max + 2 |c|max^2),
// in the place of the fp32 kernel's 2 D + 4.  At D = 256 the window is 6 x as wide: without the shift below 8 % of the
// points of configs[3] instead of 1.4 % would go to the re-check (16-27 % inside a fit), with it 2-4 % do -- and the pass
// itself needs a third of the fp32 kernel's time.  (Other cuts end at the same width: a fourth product, xm cm, trades 512 u of left-out terms for 2 D u of
// roundings; all 24 bits -- three bf16 per operand, six products -- 1536 u for 6 D u.)  The labels are the exact
// tier's, as before: tools/fuzz_kmeans.py, tests/test_hip_kernels.py::test_nearest_center_*.
//
// THE SHIFT.  E is proportional to |x| |c|max, and the distances do not change when the same vector mu is taken off
// points and centers: the tier works on x~ = fl32(x - mu), c~ = fl32(c - mu) with mu the column means of the points (of 65 536 of them spread over a larger tile)
// (a prepared buffer: computed once with it) or of the centers (a stand-alone call) -- for k-means data, whose centers
// ARE means of points, that takes the common offset out of both norms (configs[3], uniform [0, 1)^256: |x| |c|max
// 74 -> 4.6 once the centers have settled, and with it the share of points inside the window 16-27 % -> 1-2 %).  The two
// roundings are part of the bound -- |c~|^2/2 is taken of the ROUNDED c~, and (x~, c~) differ from (x - mu, c - mu) by
// at most u relative per element: 2 u |x~| |c~| + u |c~|^2 in the score, the "+ 5" of F -- and mu's value is not: any
// vector is a valid shift.  The exact stage reads the caller's x and c.
//
// Operand images: the points are cut once per call -- or once per fit: sp_kmeans_points_prepare -- into two bf16
// arrays (features padded to 16, zeros beyond d) plus |x - mu|^2 per point; the centers once per call.  An image is
// stored k-tile-major, [dp / KS_BK][rows][KS_BK] (see sp_split_rows_kernel: with row-major images every 128-byte line of a
// point was fetched by four different k-steps -- 2.48 -> 1.97 ms at configs[3] for the layout alone).  A k-tile is
// KS_BK = 32 features = 64 bytes per row and image; the LDS images are [row][4 chunks of 16 B], chunk q of row r in
// slot q ^ ((r >> 2) & 3) -- the fp32 kernel's image, bank for bank (conflict-free 16-B fragment reads), filled by global_load_lds_dwordx4 like the fp32 kernel's, the
// pieces of a request spread over the MFMAs of the k-step before.  A lane's fragment is 8 consecutive features of one
// row for both operands, so whatever order the instruction gives the 16 features of its K dimension, A and B agree.
#pragma once

typedef __bf16 km_bf16x8 __attribute__((ext_vector_type(8)));
#ifndef KS_ABLATE
#define KS_ABLATE 0     // timing-only builds (-DKS_ABLATE=n, tools/km_first.py): 1 no k-tile loads, 2 no fragment reads, 4 a
                        // sixteenth of the epilogue (NB: the MFMAs whose results it no longer reads are dropped too), 8 no MFMAs, 16 point tiles from the L2
#endif

namespace {

// Workgroup tile: 256 centers x (64 WN) points, 2 x WN waves of 128 centers x 64 points each, KS_BK features per
// k-step.  Default: WN = 4 (512 lanes, ONE workgroup per CU, 130 KB of LDS) with 32 features per k-step -- 48 MFMAs per
// wave and barrier; first pass at configs[3], one box: 1.74 ms against 1.82 for the fp32 kernel's geometry (WN = 2, 16
// features, two workgroups per CU; 1.78 with a third LDS stage, k-tiles requested two k-steps ahead) and 1.86 for
// WN = 4 with 16 features: what the wider tile loses to its 8-wave barrier the halved number of barriers more than
// returns.  History of the kernel at configs[3] (profiles/r04_notes.md): 2.42 ms with row-major images; timed with
// parts removed 1.47 without the k-tile loads, 2.32 without the fragment reads, 2.00 without the MFMAs, 0.57 without all
// of them; the k-tile-major layout of the images gave 20 % (the loads were waiting for lines the L2 had dropped, not
// for bytes or issue slots).
#ifndef KS_BK_FEATURES
#define KS_BK_FEATURES 32     // (-DKS_BK_FEATURES=16: the round's first geometry, 256 x 128 tiles on two workgroups per CU)
#endif
constexpr int KS_BK = KS_BK_FEATURES;                          // features per k-step (and per slab of an image): 16 or 32
constexpr int KS_RB = KS_BK * 2;                               // bytes of a row of a k-tile
constexpr int KS_CH = KS_BK / 8;                               // its 16-byte chunks
constexpr int KS_BM = KN_BM;
constexpr int KS_A_BYTES = KS_BM * KS_RB;                      // one center image of a k-tile
static_assert(KS_BK == 16 || KS_BK == 32, "k-tile");
template <int WN>
struct KsCfg {
  static constexpr int BN = 64 * WN, NW = 2 * WN, THREADS = 64 * NW;
  static constexpr int B_BYTES = BN * KS_RB;
  static constexpr int STAGE_BYTES = 2 * KS_A_BYTES + 2 * B_BYTES;     // Ah | Am | Bh | Bm
  static constexpr int STAGES = 2;                                     // (a third gave 2 % with 16-feature k-tiles; with 32 it does not fit)
  static constexpr int SMEM_BYTES = STAGES * STAGE_BYTES + 2 * KS_BM * 4;   // + two |c|^2/2 slices
  static constexpr int APW = KS_A_BYTES / 1024 / NW;                    // 1-KiB pieces of a center image per wave
  static constexpr int BPW = B_BYTES / 1024 / NW;                       // ... of a point image
  static constexpr int SLOTS = SP_CUS * (WN == 2 ? 2 : 1);             // resident workgroups
  static_assert(SMEM_BYTES * (WN == 2 ? 2 : 1) <= 160 * 1024, "LDS budget of a CU");
  static_assert(APW >= 1 && BPW >= 1, "every wave brings pieces of both operands");
  static_assert(SLOTS * BN == KM_TAIL_POINTS, "the tail buffer is sized for one round of either geometry");
};

__device__ __forceinline__ float km_split_factor(int d) { return 6.1f * (float)d + 1550.0f; }

// Column means of the rows of X (any dtype the tiers take), two launches: KM_MEAN_BLOCKS partial sums, then their sum
// over the row count.  (The value is a SHIFT, not a result: whatever it is, the labels are the same.)
template <typename T>
__global__ __launch_bounds__(256) void sp_col_partial_kernel(const T* __restrict__ X, int64_t ldx, int64_t n, int d,
                                                             int dp, float* __restrict__ part) {
  const int64_t per = (n + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * per, r1 = r0 + per < n ? r0 + per : n;
  for (int j = threadIdx.x; j < dp; j += blockDim.x) {
    float s = 0.f;
    if (j < d) {
      int64_t r = r0;
      for (; r + 8 <= r1; r += 8) {          // eight rows in flight per lane
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (float)X[(r + u) * ldx + j];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
      }
      for (; r < r1; ++r) s += (float)X[r * ldx + j];
    }
    part[(int64_t)blockIdx.x * dp + j] = s;
  }
}
// (one wave per column: lane l adds partials l, l + 64, ...; one thread per column walked the 2048 partials in 256
// dependent batches -- 113 us for 2 MB in a fit's set-up)
__global__ __launch_bounds__(256) void sp_col_finish_kernel(const float* __restrict__ part, int blocks, int dp, int64_t n,
                                                            float* __restrict__ mu) {
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= dp) return;
  float s = 0.f;
  int b = lane;
  for (; b + 7 * 64 < blocks; b += 8 * 64) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = part[(int64_t)(b + u * 64) * dp + j];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; b < blocks; b += 64) s += part[(int64_t)b * dp + j];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
  if (lane == 0) mu[j] = n > 0 ? s / (float)n : 0.f;
}

// hi / mid images of the fp32 rows MINUS the shift `mu` (NULL: none): one wavefront per row, zeros beyond d;
// optionally the squared norm of the shifted row (fp32 sum, any order: the bound takes it with slack)
// Layout of an image: [dp / KS_BK k-tiles][n rows][KS_BK features] -- the k-tile of the rows a workgroup brings per
// k-step is ONE contiguous block (row-major [n][dp] images made it 32 bytes out of every row's 512: each
// 128-byte line was fetched by four different k-steps, and between them the L2 of an XCD, 64 workgroups' worth of
// such lines plus the centers, had usually dropped it).
__device__ __forceinline__ int64_t ks_at(int64_t row, int j, int64_t n) { return ((int64_t)(j / KS_BK) * n + row) * KS_BK + (j % KS_BK); }
__global__ __launch_bounds__(256) void sp_split_rows_kernel(const float* __restrict__ X, int64_t ldx, int64_t n, int d,
                                                            int dp, const float* __restrict__ mu,
                                                            __bf16* __restrict__ Xh, __bf16* __restrict__ Xm,
                                                            float* __restrict__ xn2) {
  const int lane = threadIdx.x & 63;
  const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (row >= n) return;
  const float* __restrict__ x = X + row * ldx;
  float s = 0.f;
  typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
  const bool wide = (ldx % 4 == 0) && ((((uintptr_t)X) & 15) == 0) && (d % 4 == 0);   // (dp % 16 == 0 always)
  if (wide) {
    for (int j = 4 * lane; j < dp; j += 256) {
      km_f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (j < d) {
        v = *(const km_f32x4*)(x + j);
        if (mu) v -= *(const km_f32x4*)(mu + j);
      }
      bf4 h, m;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        h[e] = (__bf16)v[e];
        m[e] = (__bf16)(v[e] - (float)h[e]);
        s = __builtin_fmaf(v[e], v[e], s);
      }
      *(bf4*)(Xh + ks_at(row, j, n)) = h;
      *(bf4*)(Xm + ks_at(row, j, n)) = m;
    }
  } else {
    for (int j = lane; j < dp; j += 64) {
      const float v = j < d ? (mu ? x[j] - mu[j] : x[j]) : 0.f;
      const __bf16 h = (__bf16)v;
      const __bf16 m = (__bf16)(v - (float)h);
      Xh[ks_at(row, j, n)] = h;
      Xm[ks_at(row, j, n)] = m;
      s = __builtin_fmaf(v, v, s);
    }
  }
  if (xn2) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) xn2[row] = s;
  }
}

// The three center launches of a first pass in one (a fit runs them every iteration): memset of Cf, the shifted
// fp32 copy Cf[c][j] = fl32(C[c][j] - mu[j]) and its cut into the two bf16 images (sp_split_rows_kernel).  One
// wavefront per center row of the padded table: v = fl32(C[c][j] - mu[j]) (0 beyond d, and on the padding rows),
// hi / mid images written straight from the registers -- the fp32 copy is never stored: nothing but the cut read it --
// cn[c] = |v|^2 / 2 (fp64 sum, rounded; +inf on padding), and ONE atomicMax per workgroup for max |v|^2 (one per
// center was 1024 same-address atomics: most of the old kernel's 15 us).  Values identical to the three launches'.
template <typename TC>
__global__ __launch_bounds__(256) void sp_centers_split_prep_kernel(const TC* __restrict__ C, int64_t ldc, int k, int d,
                                                                    int kp, int dp, const float* __restrict__ mu,
                                                                    __bf16* __restrict__ Ch, __bf16* __restrict__ Cm,
                                                                    float* __restrict__ cn, unsigned* __restrict__ cmax2) {
  __shared__ unsigned wave_max[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  unsigned mine = 0u;
  if (c < kp) {
    double s = 0.0;
    for (int j = lane; j < dp; j += 64) {
      float v = 0.f;
      if (c < k && j < d) {
        v = (float)((double)C[(int64_t)c * ldc + j] - (double)mu[j]);
        s += (double)v * (double)v;
      }
      const __bf16 h = (__bf16)v;
      Ch[ks_at(c, j, kp)] = h;
      Cm[ks_at(c, j, kp)] = (__bf16)(v - (float)h);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) {
      if (c < k) {
        const float sf = (float)s;
        cn[c] = 0.5f * sf;
        mine = __float_as_uint(sf * 1.0000002f);
      } else {
        cn[c] = INFINITY;
      }
    }
  }
  if (lane == 0) wave_max[wv] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned m = wave_max[0];
    for (int i = 1; i < 4; ++i) m = wave_max[i] > m ? wave_max[i] : m;
    if (m) atomicMax(cmax2, m);
  }
}

// The kernel.  Same roles as sp_nearest_nt_kernel's template parameters; same outputs.
template <bool RECHECK, bool PARTIAL, int WN>
__global__ __launch_bounds__(KsCfg<WN>::THREADS, 2) void sp_nearest_split_kernel(
    const __bf16* __restrict__ Xh, const __bf16* __restrict__ Xm, const float* __restrict__ xn2,
    const __bf16* __restrict__ Ch, const __bf16* __restrict__ Cm, const float* __restrict__ chalf,
    const unsigned* __restrict__ cmax2_bits, int n, int d, int dp, int kp, int64_t* __restrict__ labels,
    int* __restrict__ amb_rows, float* __restrict__ amb_best, int* __restrict__ amb_count,
    unsigned* __restrict__ cand_mask, float* __restrict__ part, int ldp, int per_tiles, int first_point,
    int n_total) {
  using K = KsCfg<WN>;
  constexpr int KS_BN = K::BN, KS_B_BYTES = K::B_BYTES, KS_STAGE_BYTES = K::STAGE_BYTES, KS_SMEM_BYTES = K::SMEM_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];      // KS_SMEM_BYTES (above 64 KiB for WN = 4)
  (void)KS_SMEM_BYTES;
  float* chs = (float*)(smem + K::STAGES * KS_STAGE_BYTES);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;         // wm: center half (128 rows), wn: which 64 point columns
  const int l31 = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.x * KS_BN;              // first point (RECHECK: first list slot) of this workgroup
  int listed = 0;
  if constexpr (RECHECK) {
    listed = *amb_count;
    if (listed > n || m0 >= listed) return;       // (n: the capacity of the candidate masks)
  }

  // ---- k-tile pieces: 1 KiB = one wave-wide 16-B load = 32 rows of one image.  Wave `wid` brings pieces
  // APW wid .. APW wid + APW - 1 of each center image and piece wid of each point image.
  constexpr int APW = K::APW;
  unsigned a_off[APW];
#pragma unroll
  for (int j = 0; j < APW; ++j) {
    const int slot = (wid * APW + j) * 64 + lane, row = slot / KS_CH;
    a_off[j] = (unsigned)(row * KS_RB + (((slot % KS_CH) ^ ((row >> 2) & (KS_CH - 1))) * 16));   // (bytes inside a k-tile slab)
  }
  constexpr int BPW = K::BPW;
  typename std::conditional<RECHECK, int64_t, unsigned>::type b_off[BPW];
#pragma unroll
  for (int j = 0; j < BPW; ++j) {
    const int slot = (wid * BPW + j) * 64 + lane;
    int row = slot / KS_CH;
    const int chunk = (slot % KS_CH) ^ ((row >> 2) & (KS_CH - 1));
    if constexpr (RECHECK) {
      const int listed_row = m0 + row < listed ? m0 + row : listed - 1;   // tail: repeat the last listed point
      b_off[j] = (int64_t)amb_rows[listed_row] * KS_RB + chunk * 16;
    } else {
      if (m0 + row > n - 1) row = n - 1 - m0;     // clamp: results of points >= n are discarded
      b_off[j] = (unsigned)(row * KS_RB + chunk * 16);
    }
  }
  // (PARTIAL launches pass the whole images and where their points start: first_point)
  // (KS_ABLATE & 16: every workgroup reads the point tile of its XCD's first workgroup -- what the pass costs when the
  // point images come from the L2)
#ifndef KS_ABLATE_MOD
#define KS_ABLATE_MOD 8
#endif
  const int m0_ld = (KS_ABLATE & 16) ? (int)(blockIdx.x % KS_ABLATE_MOD) * KS_BN : m0;
  const char* __restrict__ Xh_blk = (const char*)(RECHECK ? Xh : Xh + (int64_t)(first_point + m0_ld) * KS_BK);
  const char* __restrict__ Xm_blk = (const char*)(RECHECK ? Xm : Xm + (int64_t)(first_point + m0_ld) * KS_BK);
  const int64_t x_slab = (int64_t)n_total * KS_RB, c_slab = (int64_t)kp * KS_RB;   // bytes per k-tile of an image
  const unsigned s_base = SP_LDS_ADDR(smem);
  const unsigned chs_w = SP_LDS_ADDR(chs);
  const int nt = dp / KS_BK;
  const int tm_first = RECHECK ? (int)blockIdx.y : PARTIAL ? (int)blockIdx.y * per_tiles : 0;
  const int tiles_m = RECHECK ? 1 : PARTIAL ? min(per_tiles, kp / KS_BM - tm_first) : kp / KS_BM;
  const int steps = nt * tiles_m;

  // The request for a k-tile is 2 APW center pieces + 2 point pieces (+ the |c|^2/2 slice at the first k-tile of a
  // block, wave 0).  KS_LOAD_BEGIN fixes which k-tile and where; KS_PIECE(i) issues piece i -- the k-step spreads them
  // over its MFMAs (each is an M0 write and a ~100-cycle issue the matrix pipe works through; issued in one run
  // ahead of the fragment reads, as the fp32 kernel does, they were a third of this kernel's time).
  int ld_tr = 0, ld_kt = 0;
  constexpr int NPIECES = 2 * APW + 2 * BPW + 1;
#define KS_LOAD_BEGIN(stage_of)                                                                        \
  const int tr_ = ld_tr, kt_ = ld_kt;                                                                  \
  if (++ld_kt == nt) {                                                                                 \
    ld_kt = 0;                                                                                         \
    ++ld_tr;                                                                                           \
  }                                                                                                    \
  const int tm_ = tm_first + tr_;                                                                      \
  const int64_t ka_ = kt_ * c_slab + (int64_t)tm_ * (KS_BM * KS_RB);                                   \
  const unsigned st_ = s_base + (unsigned)(stage_of) * KS_STAGE_BYTES;
#define KS_PIECE(i)                                                                                    \
  do {                                                                                                 \
    if ((i) < 2 * APW) {                                                                               \
      const int j_ = (i) >> 1;                                                                         \
      if (((i) & 1) == 0) SP_GLDS_S((const char*)Ch + ka_, a_off[j_], st_ + (wid * APW + j_) * 1024);  \
      else SP_GLDS_S((const char*)Cm + ka_, a_off[j_], st_ + KS_A_BYTES + (wid * APW + j_) * 1024);    \
    } else if ((i) < 2 * APW + 2 * BPW) {                                                              \
      const int j_ = ((i) - 2 * APW) >> 1;                                                             \
      const unsigned dst_ = st_ + 2 * KS_A_BYTES + ((((i) - 2 * APW) & 1) ? KS_B_BYTES : 0) + (wid * BPW + j_) * 1024; \
      const char* src_ = ((((i) - 2 * APW) & 1) ? Xm_blk : Xh_blk) + kt_ * x_slab;                     \
      if constexpr (RECHECK) SP_GLDS_V(src_ + b_off[j_], dst_);                                        \
      else SP_GLDS_S(src_, b_off[j_], dst_);                                                           \
    } else if (kt_ == 0 && wid == 0) {                                                                 \
      SP_GLDS_S(chalf + tm_ * KS_BM, (unsigned)lane * 16u, chs_w + (tr_ & 1) * (KS_BM * 4));           \
    }                                                                                                  \
  } while (0)

  km_f32x16 acc[4][2];
  float best[2] = {INFINITY, INFINITY}, second[2] = {INFINITY, INFINITY};
  int bpos[2] = {0, 0}, bblk[2] = {0, 0};

  int cur = 0, nxt = 1;
  {
    KS_LOAD_BEGIN(0)
#pragma unroll
    for (int i = 0; i < NPIECES; ++i) KS_PIECE(i);
  }
  SP_GLDS_LANDED();
  __syncthreads();
  // fragments: row (wave tile row + l31 [+ 32 i]), 16-B chunk (2 kk + lh) ^ ((row >> 2) & (KS_CH - 1)) -- the same
  // xor for both operands; kk: which 16 of the k-tile's features
  const int sw = (l31 >> 2) & (KS_CH - 1);
  int a_frag[KS_BK / 16], b_frag[KS_BK / 16];
#pragma unroll
  for (int kk = 0; kk < KS_BK / 16; ++kk) {
    a_frag[kk] = (wm * 128 + l31) * KS_RB + (((2 * kk + lh) ^ sw) * 16);
    b_frag[kk] = 2 * KS_A_BYTES + (wn * 64 + l31) * KS_RB + (((2 * kk + lh) ^ sw) * 16);
  }

  int t = 0;
  auto kstep = [&](auto first_of_block) {
    constexpr bool FIRST = decltype(first_of_block)::value;
    const bool more = !(KS_ABLATE & 1) && t + 1 < steps;
    KS_LOAD_BEGIN(nxt)
    const char* st = smem + cur * KS_STAGE_BYTES;
    // per 16 features of the k-tile: 24 MFMAs (mid x hi, hi x mid, hi x hi for the 4 x 2 tiles of the wave); a piece of
    // the next k-tile's request after every third MFMA
    int piece = 0;
#pragma unroll
    for (int kk = 0; kk < KS_BK / 16; ++kk) {
      km_bf16x8 ah[4], am[4], bh[2], bm[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (KS_ABLATE & 2) {
          ah[i] = am[i] = *(const km_bf16x8*)(smem + a_frag[0]);
          asm volatile("" : "+v"(ah[i]), "+v"(am[i]));
          continue;
        }
        ah[i] = *(const km_bf16x8*)(st + a_frag[kk] + i * 32 * KS_RB);
        am[i] = *(const km_bf16x8*)(st + KS_A_BYTES + a_frag[kk] + i * 32 * KS_RB);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (KS_ABLATE & 2) {
          bh[j] = bm[j] = *(const km_bf16x8*)(smem + b_frag[0]);
          asm volatile("" : "+v"(bh[j]), "+v"(bm[j]));
          continue;
        }
        bh[j] = *(const km_bf16x8*)(st + b_frag[kk] + j * 32 * KS_RB);
        bm[j] = *(const km_bf16x8*)(st + KS_B_BYTES + b_frag[kk] + j * 32 * KS_RB);
      }
#pragma unroll
      for (int idx = 0; idx < 24; ++idx) {
        const int term = idx >> 3, i = (idx & 7) >> 1, j = idx & 1;
        if (!(KS_ABLATE & 8)) {
          const km_bf16x8 a = term == 0 ? am[i] : ah[i];
          const km_bf16x8 b = term == 1 ? bm[j] : bh[j];
          if (FIRST && kk == 0 && term == 0) {
            const km_f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, zero, 0, 0, 0);
          } else {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i][j], 0, 0, 0);
          }
        } else {
          asm volatile("" : "+v"(acc[i][j]) : "v"(ah[i]), "v"(am[i]), "v"(bh[j]), "v"(bm[j]));
        }
        if (idx % 3 == 2 && piece < NPIECES) {
          __builtin_amdgcn_sched_barrier(0);
          if (more) KS_PIECE(piece);
          __builtin_amdgcn_sched_barrier(0);
          ++piece;
        }
      }
    }
    static_assert(NPIECES <= 8 * (KS_BK / 16), "one piece after every third MFMA");
    if (more) SP_GLDS_LANDED();
    __syncthreads();
    ++t;
    cur ^= 1;
    nxt ^= 1;
  };

  const float cmax2 = __uint_as_float(*cmax2_bits);
  const float cmax = sqrtf(cmax2) * 1.0000002f;
  const float ef = km_split_factor(d);
  for (int tr = 0; tr < tiles_m; ++tr) {
    const int tm = tm_first + tr;
    kstep(std::true_type());
    for (int kt = 1; kt < nt; ++kt) kstep(std::false_type());
    // ---- epilogue of center block tm (as sp_nearest_nt_kernel's: halved scores h = |c|^2/2 - x.c; the rows of a
    // lane ascend with (i, q, e), `<` keeps the first minimum, the new second best is the median of (best, second, h))
    const float* chb = chs + (tr & 1) * KS_BM + wm * 128 + 4 * lh;
    if constexpr (RECHECK) {
      float thr[2];
      int slot[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        slot[j] = m0 + wn * 64 + j * 32 + l31;
        const int at = slot[j] < listed ? slot[j] : listed - 1;
        const float xnorm = sqrtf(xn2[amb_rows[at]]) * 1.001f;
        const float E = 5.9604645e-8f * (ef * xnorm * cmax + 2.0f * cmax2);
        thr[j] = amb_best[at] + E * 1.001f + 1e-30f;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        unsigned bits[2] = {0u, 0u};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const km_f32x4 ch4 = *(const km_f32x4*)(chb + i * 32 + 8 * q);
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              bits[j] |= (ch4[e] - acc[i][j][4 * q + e] <= thr[j]) ? (1u << (8 * q + e)) : 0u;   // (+ 4 lh below)
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          unsigned w = bits[j] << (4 * lh);
          w |= __shfl_xor(w, 32);
          if (lh == 0 && slot[j] < listed) cand_mask[(int64_t)slot[j] * (kp / 32) + tm * 8 + wm * 4 + i] = w;
        }
      }
      return;
    }
    const float before[2] = {best[0], best[1]};
#pragma unroll
    for (int i = 0; i < ((KS_ABLATE & 4) ? 1 : 4); ++i)
#pragma unroll
      for (int q = 0; q < ((KS_ABLATE & 4) ? 1 : 4); ++q) {
        const km_f32x4 ch4 = *(const km_f32x4*)(chb + i * 32 + 8 * q);   // rows i*32 + 8q + 4lh + (0..3)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v = ch4[e] - acc[i][j][4 * q + e];
            const bool better = v < best[j];
            second[j] = __builtin_amdgcn_fmed3f(best[j], second[j], v);
            bpos[j] = better ? 16 * i + 4 * q + e : bpos[j];
            best[j] = better ? v : best[j];          // (a select on the compare's result: fminf costs a canonicalising v_max first)
          }
      }
#pragma unroll
    for (int j = 0; j < 2; ++j) bblk[j] = best[j] < before[j] ? tm : bblk[j];
  }
#undef KS_LOAD_BEGIN
#undef KS_PIECE

  // ---- merge: the two lane halves of a column (rows differ by 4), then the two center waves (LDS)
  float* mb_s = (float*)smem;         // [128]   (the stages are dead: every wave passed the last barrier)
  float* ms_s = mb_s + KS_BN;
  int* mi_s = (int*)(mb_s + 2 * KS_BN);
  float bb[2], ss[2];
  int ii[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float b = best[j], s = second[j];
    int ix = bblk[j] * KS_BM + wm * 128 + (bpos[j] >> 4) * 32 + ((bpos[j] >> 2) & 3) * 8 + (bpos[j] & 3) + 4 * lh;
    const float ob = __shfl_xor(b, 32), os = __shfl_xor(s, 32);
    const int oi = __shfl_xor(ix, 32);
    if (ob < b || (ob == b && oi < ix)) {
      s = fminf(b, os);
      b = ob;
      ix = oi;
    } else {
      s = fminf(ob, s);
    }
    bb[j] = b;
    ss[j] = s;
    ii[j] = ix;
  }
  if (wm == 1 && lh == 0) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = wn * 64 + j * 32 + l31;
      mb_s[col] = bb[j];
      ms_s[col] = ss[j];
      mi_s[col] = ii[j];
    }
  }
  __syncthreads();
  if (wm == 0 && lh == 0) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = wn * 64 + j * 32 + l31;
      float b = bb[j], s = ss[j];
      int ix = ii[j];
      const float ob = mb_s[col], os = ms_s[col];
      const int oi = mi_s[col];
      if (ob < b || (ob == b && oi < ix)) {
        s = fminf(b, os);
        b = ob;
        ix = oi;
      } else {
        s = fminf(ob, s);
      }
      if (m0 + col < n) {
        if constexpr (PARTIAL) {
          const int64_t at = (int64_t)blockIdx.y * 3 * ldp + m0 + col;
          part[at] = b;
          part[at + ldp] = s;
          ((int*)part)[at + 2 * (int64_t)ldp] = ix;
          if (blockIdx.y == 0) part[(int64_t)gridDim.y * 3 * ldp + m0 + col] = xn2[first_point + m0 + col];
        } else {
          km_decide(b, s, ix, xn2[first_point + m0 + col], first_point + m0 + col, ef, cmax, cmax2, labels, amb_rows, amb_best,
                    amb_count);
        }
      }
    }
  }
}

// mu = column means of `rows` rows of X (points: once per prepared buffer; centers: once per unprepared call)
template <typename T>
static int km_col_means(const T* X, int64_t ldx, int64_t rows, int64_t d, const KmWorkspace& w, hipStream_t st) {
  const int blocks = rows < KM_MEAN_BLOCKS ? (int)(rows < 1 ? 1 : rows) : KM_MEAN_BLOCKS;
  hipLaunchKernelGGL((sp_col_partial_kernel<T>), dim3(blocks), dim3(256), 0, st, X, ldx, rows, (int)d, (int)w.dp, w.colsum);
  hipLaunchKernelGGL(sp_col_finish_kernel, dim3((unsigned)((w.dp + 3) / 4)), dim3(256), 0, st, (const float*)w.colsum,
                     blocks, (int)w.dp, rows, w.mu);
  SP_CHECK_LAUNCH();
  return 0;
}

// hi / mid images and squared norms of the points minus w.mu (what a prepared buffer holds, behind its mu)
static int km_split_points(const float* X, int64_t ldx, int64_t n, int64_t d, const KmWorkspace& w, hipStream_t st) {
  if (n < 1) return 0;
  hipLaunchKernelGGL(sp_split_rows_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, X, ldx, n, (int)d, (int)w.dp,
                     (const float*)w.mu, w.Xh, w.Xm, w.xn2);
  SP_CHECK_LAUNCH();
  return 0;
}

// First pass of the split tier: centers shifted, prepared as for the fp32 tier (Cf, |c|^2/2, max |c|^2) and cut into
// their two images, then the whole rounds, the split last round and its merge -- the launch plan of
// sp_nearest_fused_launch for this kernel's geometry.
template <bool RECHECK, bool PARTIAL, int WN, typename... Args>
static int km_split_go(dim3 grid, hipStream_t st, Args... args) {
  using K = KsCfg<WN>;
  auto kern = sp_nearest_split_kernel<RECHECK, PARTIAL, WN>;
  static bool attr_set = false;
  if (!attr_set) {
    SP_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, K::SMEM_BYTES));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, grid, dim3(K::THREADS), K::SMEM_BYTES, st, args...);
  SP_CHECK_LAUNCH();
  return 0;
}

static int km_split_wn() {
  static int wn = -1;
  if (wn < 0) {
    const char* e = getenv("SP_KM_SPLIT_WN");
    wn = e && atoi(e) == 2 ? 2 : 4;     // (2: only with -DKS_BK_FEATURES=16)
  }
  return wn;
}

template <int WN>
static int sp_nearest_split_launch_wn(const void* C, int32_t cdtype, int64_t ldc, int64_t n, int64_t k, int64_t d,
                                      int64_t* labels, const KmWorkspace& w, hipStream_t st) {
  using K = KsCfg<WN>;
  const int64_t kp = w.kp, dp = w.dp;
  SP_HIP(hipMemsetAsync(w.cmax2, 0, 512, st));   // cmax2 and amb_count
  const unsigned pblocks = (unsigned)((kp + 3) / 4);   // one wavefront per center
  if (cdtype == SP_F32)
    hipLaunchKernelGGL((sp_centers_split_prep_kernel<float>), dim3(pblocks), dim3(256), 0, st, (const float*)C, ldc,
                       (int)k, (int)d, (int)kp, (int)dp, (const float*)w.mu, w.Ch, w.Cm, w.cn, w.cmax2);
  else
    hipLaunchKernelGGL((sp_centers_split_prep_kernel<double>), dim3(pblocks), dim3(256), 0, st, (const double*)C, ldc,
                       (int)k, (int)d, (int)kp, (int)dp, (const float*)w.mu, w.Ch, w.Cm, w.cn, w.cmax2);
  SP_CHECK_LAUNCH();
  const int64_t blocks = (n + K::BN - 1) / K::BN, tiles = kp / KS_BM;
  int64_t rem = blocks % K::SLOTS, split = 1, per = tiles;
  static const bool tail_off = getenv("SP_KM_TAIL_SPLIT") && atoi(getenv("SP_KM_TAIL_SPLIT")) == 0;
  if (rem > 0 && tiles > 1 && !tail_off) {
    split = tiles < KM_TAIL_SPLIT ? tiles : KM_TAIL_SPLIT;
    per = (tiles + split - 1) / split;
    split = (tiles + per - 1) / per;
    const int64_t rounds = (rem * split + K::SLOTS - 1) / K::SLOTS;
    if (rounds * per >= tiles) split = 1;     // no shorter than the plain round
  }
  if (split == 1) rem = 0;
  const int64_t whole = blocks - rem, n_whole = whole * K::BN < n ? whole * K::BN : n;
  if (whole > 0 &&
      km_split_go<false, false, WN>(dim3((unsigned)whole), st, (const __bf16*)w.Xh, (const __bf16*)w.Xm, (const float*)w.xn2,
                                    (const __bf16*)w.Ch, (const __bf16*)w.Cm, (const float*)w.cn, (const unsigned*)w.cmax2,
                                    (int)n_whole, (int)d, (int)dp, (int)kp, labels, w.amb_rows, w.amb_best, w.amb_count,
                                    (unsigned*)nullptr, (float*)nullptr, 0, 0, 0, (int)n))
    return 1;
  if (rem > 0) {
    const int n_tail = (int)(n - n_whole);
    if (km_split_go<false, true, WN>(dim3((unsigned)rem, (unsigned)split), st, (const __bf16*)w.Xh,
                                     (const __bf16*)w.Xm, (const float*)w.xn2,
                                     (const __bf16*)w.Ch, (const __bf16*)w.Cm, (const float*)w.cn,
                                     (const unsigned*)w.cmax2, n_tail, (int)d, (int)dp, (int)kp, (int64_t*)nullptr,
                                     (int*)nullptr, (float*)nullptr, (int*)nullptr, (unsigned*)nullptr, w.part,
                                     KM_TAIL_POINTS, (int)per, (int)n_whole, (int)n))
      return 1;
    hipLaunchKernelGGL(sp_nearest_merge_parts_kernel, dim3((unsigned)((n_tail + 255) / 256)), dim3(256), 0, st, w.part,
                       (int)split, KM_TAIL_POINTS, n_tail, (int)n_whole, 6.1f * (float)d + 1550.0f, w.cmax2, labels,
                       w.amb_rows, w.amb_best, w.amb_count);
    SP_CHECK_LAUNCH();
  }
  return 0;
}

static int sp_nearest_split_launch(const void* C, int32_t cdtype, int64_t ldc, int64_t n, int64_t k, int64_t d,
                                   int64_t* labels, const KmWorkspace& w, hipStream_t st) {
#if KS_BK_FEATURES == 16
  if (km_split_wn() == 2) return sp_nearest_split_launch_wn<2>(C, cdtype, ldc, n, k, d, labels, w, st);
#endif
  return sp_nearest_split_launch_wn<4>(C, cdtype, ldc, n, k, d, labels, w, st);   // (32-feature k-tiles: this geometry only)
}

// Second pass over the listed points: marks the centers inside each point's error window.
template <int WN>
static int sp_nearest_split_mark_wn(int64_t d, const KmWorkspace& w, hipStream_t st) {
  using K = KsCfg<WN>;
  const dim3 grid((unsigned)((w.cand_cap + K::BN - 1) / K::BN), (unsigned)(w.kp / KS_BM));
  return km_split_go<true, false, WN>(grid, st, (const __bf16*)w.Xh, (const __bf16*)w.Xm, (const float*)w.xn2,
                                      (const __bf16*)w.Ch, (const __bf16*)w.Cm, (const float*)w.cn,
                                      (const unsigned*)w.cmax2, (int)w.cand_cap, (int)d, (int)w.dp, (int)w.kp,
                                      (int64_t*)nullptr, w.amb_rows, w.amb_best, w.amb_count, w.cand_mask, (float*)nullptr,
                                      0, 0, 0, (int)w.n_points);
}
static int sp_nearest_split_mark_candidates(int64_t d, const KmWorkspace& w, hipStream_t st) {
#if KS_BK_FEATURES == 16
  if (km_split_wn() == 2) return sp_nearest_split_mark_wn<2>(d, w, st);
#endif
  return sp_nearest_split_mark_wn<4>(d, w, st);
}

}  // namespace

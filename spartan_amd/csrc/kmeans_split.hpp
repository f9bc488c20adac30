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
//     (tools/_exp/mfma_bf16_probe.hip, 20 000 random operand sets with exponents spread over 30 binades): it is
//     neither a k-ordered chain nor an exact sum rounded once, and errs by at most 2.5 u (|c| + sum|p|) per MFMA of
//     16 products -- a seventh of what this bound grants it
//   * |c|^2/2 rounded once, the final subtraction rounded once     (as in the fp32 kernel)
// and the score |c|^2 - 2 x.c, in whose units km_decide and the candidate window work, by twice that: the error
// factor of this tier is
//   F = 6.1 D + 1550,    E = u (F |x| |c|max + 2 |c|max^2),
// in the place of the fp32 kernel's 2 D + 4.  At D = 256 the window is 6 x as wide and 8 % of the points of configs[3]
// instead of 1.4 % go to the re-check -- whose first stage is this same kernel -- while the pass itself needs half the
// time.  (Other cuts end at the same width: a fourth product, xm cm, trades 512 u of left-out terms for 2 D u of
// roundings; all 24 bits -- three bf16 per operand, six products -- 1536 u for 6 D u.)  The labels are the exact
// tier's, as before: tools/fuzz_kmeans.py, tests/test_hip_kernels.py::test_nearest_center_*.
//
// THE SHIFT.  E is proportional to |x| |c|max, and the distances do not change when the same vector mu is taken off
// points and centers: the tier works on x~ = fl32(x - mu), c~ = fl32(c - mu) with mu the column means of the points
// (a prepared buffer: computed once with it) or of the centers (a stand-alone call) -- for k-means data, whose centers
// ARE means of points, that takes the common offset out of both norms (configs[3], uniform [0, 1)^256: |x| |c|max
// 74 -> 4.6 once the centers have settled, and with it the share of points inside the window 16-27 % -> 1-2 %).  The two
// roundings are part of the bound -- |c~|^2/2 is taken of the ROUNDED c~, and (x~, c~) differ from (x - mu, c - mu) by
// at most u relative per element: 2 u |x~| |c~| + u |c~|^2 in the score, the "+ 5" of F -- and mu's value is not: any
// vector is a valid shift.  The exact stage reads the caller's x and c.
//
// Operand images: the points are cut once per call -- or once per fit: sp_kmeans_points_prepare -- into two bf16
// arrays [n][dp] (dp = features padded to 16, zeros beyond d) plus |x|^2 per point; the centers once per call.  A
// k-tile is 16 features = 32 bytes per row and image; the LDS images are [row][2 chunks of 16 B], chunk q of row r
// in slot q ^ ((r >> 2) & 1) (conflict-free 16-B fragment reads), filled by global_load_lds_dwordx4 like the fp32
// kernel's.  A lane's fragment is 8 consecutive features of one row for both operands, so whatever order the
// instruction gives the 16 features of its K dimension, A and B agree on it.
#pragma once

typedef __bf16 km_bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int KS_BK = 16;
constexpr int KS_BM = KN_BM, KS_BN = KN_BN;                    // 256 centers x 128 points, as the fp32 kernel
constexpr int KS_A_BYTES = KS_BM * KS_BK * 2;                  // 8 KiB per image
constexpr int KS_B_BYTES = KS_BN * KS_BK * 2;                  // 4 KiB per image
constexpr int KS_STAGE_BYTES = 2 * KS_A_BYTES + 2 * KS_B_BYTES;   // Ah | Am | Bh | Bm
constexpr int KS_SMEM_BYTES = 2 * KS_STAGE_BYTES + 2 * KS_BM * 4;  // two stages + two |c|^2/2 slices (51 200 B)
static_assert(KS_SMEM_BYTES <= 65536, "static LDS limit");

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
    if (j < d)
      for (int64_t r = r0; r < r1; ++r) s += (float)X[r * ldx + j];
    part[(int64_t)blockIdx.x * dp + j] = s;
  }
}
__global__ __launch_bounds__(256) void sp_col_finish_kernel(const float* __restrict__ part, int blocks, int dp, int64_t n,
                                                            float* __restrict__ mu) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= dp) return;
  float s = 0.f;
  for (int b = 0; b < blocks; ++b) s += part[(int64_t)b * dp + j];
  mu[j] = n > 0 ? s / (float)n : 0.f;
}

// hi / mid images of the fp32 rows MINUS the shift `mu` (NULL: none): one wavefront per row, zeros beyond d;
// optionally the squared norm of the shifted row (fp32 sum, any order: the bound takes it with slack)
__global__ __launch_bounds__(256) void sp_split_rows_kernel(const float* __restrict__ X, int64_t ldx, int64_t n, int d,
                                                            int dp, const float* __restrict__ mu,
                                                            __bf16* __restrict__ Xh, __bf16* __restrict__ Xm,
                                                            float* __restrict__ xn2) {
  const int lane = threadIdx.x & 63;
  const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (row >= n) return;
  const float* __restrict__ x = X + row * ldx;
  float s = 0.f;
  for (int j = lane; j < dp; j += 64) {
    const float v = j < d ? (mu ? x[j] - mu[j] : x[j]) : 0.f;
    const __bf16 h = (__bf16)v;
    const __bf16 m = (__bf16)(v - (float)h);
    Xh[row * dp + j] = h;
    Xm[row * dp + j] = m;
    s = __builtin_fmaf(v, v, s);
  }
  if (xn2) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) xn2[row] = s;
  }
}

// sp_centers_prep_kernel for shifted centers: Cf[c][j] = fl32(C[c][j] - mu[j]) (zero padded), cn[c] = |Cf[c]|^2 / 2
// (of the ROUNDED row: the number the contraction multiplies; fp64 sum, rounded; +inf on padding), *cmax2 = max |Cf[c]|^2
template <typename TC>
__global__ __launch_bounds__(256) void sp_centers_prep_shifted_kernel(const TC* __restrict__ C, int64_t ldc, int k, int d,
                                                                      int kp, int dp, const float* __restrict__ mu,
                                                                      float* __restrict__ Cf, float* __restrict__ cn,
                                                                      unsigned* __restrict__ cmax2) {
  const int lane = threadIdx.x & 63;
  const int c = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  if (c >= kp) return;
  if (c >= k) {
    if (lane == 0) cn[c] = INFINITY;
    return;
  }
  double s = 0.0;
  for (int j = lane; j < d; j += 64) {
    const float v = (float)((double)C[(int64_t)c * ldc + j] - (double)mu[j]);
    s += (double)v * (double)v;
    Cf[(int64_t)c * dp + j] = v;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if (lane == 0) {
    const float sf = (float)s;
    cn[c] = 0.5f * sf;
    atomicMax(cmax2, __float_as_uint(sf * 1.0000002f));
  }
}

// The kernel.  Same roles as sp_nearest_nt_kernel's template parameters; same outputs.
template <bool RECHECK, bool PARTIAL>
__global__ __launch_bounds__(256, 2) void sp_nearest_split_kernel(
    const __bf16* __restrict__ Xh, const __bf16* __restrict__ Xm, const float* __restrict__ xn2,
    const __bf16* __restrict__ Ch, const __bf16* __restrict__ Cm, const float* __restrict__ chalf,
    const unsigned* __restrict__ cmax2_bits, int n, int d, int dp, int kp, int64_t* __restrict__ labels,
    int* __restrict__ amb_rows, float* __restrict__ amb_best, int* __restrict__ amb_count,
    unsigned* __restrict__ cand_mask, float* __restrict__ part, int ldp, int per_tiles, int first_point) {
  __shared__ __attribute__((aligned(16))) char smem[KS_SMEM_BYTES];
  float* chs = (float*)(smem + 2 * KS_STAGE_BYTES);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;          // wm: center half (128 rows), wn: point half (64 columns)
  const int l31 = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.x * KS_BN;              // first point (RECHECK: first list slot) of this workgroup
  int listed = 0;
  if constexpr (RECHECK) {
    listed = *amb_count;
    if (listed > n || m0 >= listed) return;       // (n: the capacity of the candidate masks)
  }

  // ---- k-tile pieces: 1 KiB = one wave-wide 16-B load = 32 rows of one image.  Wave `wid` brings pieces 2 wid and
  // 2 wid + 1 of each center image and piece wid of each point image.
  unsigned a_off[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int slot = (wid * 2 + j) * 64 + lane, row = slot >> 1;
    a_off[j] = (unsigned)(row * dp * 2 + (((slot & 1) ^ ((row >> 2) & 1)) * 16));
  }
  typename std::conditional<RECHECK, int64_t, unsigned>::type b_off;
  {
    const int slot = wid * 64 + lane;
    int row = slot >> 1;
    const int chunk = (slot & 1) ^ ((row >> 2) & 1);
    if constexpr (RECHECK) {
      const int listed_row = m0 + row < listed ? m0 + row : listed - 1;   // tail: repeat the last listed point
      b_off = ((int64_t)amb_rows[listed_row] * dp) * 2 + chunk * 16;
    } else {
      if (m0 + row > n - 1) row = n - 1 - m0;     // clamp: results of points >= n are discarded
      b_off = (unsigned)(row * dp * 2 + chunk * 16);
    }
  }
  const char* __restrict__ Xh_blk = (const char*)(RECHECK ? Xh : Xh + (int64_t)m0 * dp);
  const char* __restrict__ Xm_blk = (const char*)(RECHECK ? Xm : Xm + (int64_t)m0 * dp);
  const unsigned s_base = SP_LDS_ADDR(smem);
  const unsigned chs_w = SP_LDS_ADDR(chs);
  const int nt = dp / KS_BK;
  const int tm_first = RECHECK ? (int)blockIdx.y : PARTIAL ? (int)blockIdx.y * per_tiles : 0;
  const int tiles_m = RECHECK ? 1 : PARTIAL ? min(per_tiles, kp / KS_BM - tm_first) : kp / KS_BM;
  const int steps = nt * tiles_m;

  int ld_tr = 0, ld_kt = 0;
#define KS_LOAD(step)                                                                                  \
  do {                                                                                                 \
    const int tr_ = ld_tr, kt_ = ld_kt;                                                                \
    if (++ld_kt == nt) {                                                                               \
      ld_kt = 0;                                                                                       \
      ++ld_tr;                                                                                         \
    }                                                                                                  \
    const int tm_ = tm_first + tr_;                                                                    \
    const int64_t ka_ = ((int64_t)tm_ * KS_BM * dp + kt_ * KS_BK) * 2;                                 \
    const unsigned st_ = s_base + ((step) & 1) * KS_STAGE_BYTES;                                       \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                    \
      SP_GLDS_S((const char*)Ch + ka_, a_off[j], st_ + (wid * 2 + j) * 1024);                          \
      SP_GLDS_S((const char*)Cm + ka_, a_off[j], st_ + KS_A_BYTES + (wid * 2 + j) * 1024);             \
    }                                                                                                  \
    if constexpr (RECHECK) {                                                                           \
      SP_GLDS_V(Xh_blk + kt_ * (KS_BK * 2) + b_off, st_ + 2 * KS_A_BYTES + wid * 1024);                \
      SP_GLDS_V(Xm_blk + kt_ * (KS_BK * 2) + b_off, st_ + 2 * KS_A_BYTES + KS_B_BYTES + wid * 1024);   \
    } else {                                                                                           \
      SP_GLDS_S(Xh_blk + kt_ * (KS_BK * 2), b_off, st_ + 2 * KS_A_BYTES + wid * 1024);                 \
      SP_GLDS_S(Xm_blk + kt_ * (KS_BK * 2), b_off, st_ + 2 * KS_A_BYTES + KS_B_BYTES + wid * 1024);    \
    }                                                                                                  \
    if (kt_ == 0 && wid == 0) SP_GLDS_S(chalf + tm_ * KS_BM, (unsigned)lane * 16u, chs_w + (tr_ & 1) * (KS_BM * 4)); \
  } while (0)

  km_f32x16 acc[4][2];
  float best[2] = {INFINITY, INFINITY}, second[2] = {INFINITY, INFINITY};
  int bpos[2] = {0, 0}, bblk[2] = {0, 0};

  KS_LOAD(0);
  SP_GLDS_LANDED();
  __syncthreads();
  // fragments: row (wave tile row + l31 [+ 32 i]), 16-B chunk lh ^ ((row >> 2) & 1) -- the same xor for both operands
  const int sw = (l31 >> 2) & 1;
  const int a_frag = (wm * 128 + l31) * 32 + ((lh ^ sw) * 16);
  const int b_frag = 2 * KS_A_BYTES + (wn * 64 + l31) * 32 + ((lh ^ sw) * 16);

  int t = 0;
  auto kstep = [&](auto first_of_block) {
    constexpr bool FIRST = decltype(first_of_block)::value;
    if (t + 1 < steps) KS_LOAD(t + 1);
    const char* st = smem + (t & 1) * KS_STAGE_BYTES;
    km_bf16x8 ah[4], am[4], bh[2], bm[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ah[i] = *(const km_bf16x8*)(st + a_frag + i * 1024);
      am[i] = *(const km_bf16x8*)(st + KS_A_BYTES + a_frag + i * 1024);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bh[j] = *(const km_bf16x8*)(st + b_frag + j * 1024);
      bm[j] = *(const km_bf16x8*)(st + KS_B_BYTES + b_frag + j * 1024);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if constexpr (FIRST) {
          const km_f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bh[j], zero, 0, 0, 0);
        } else {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bh[j], acc[i][j], 0, 0, 0);
        }
      }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bm[j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
    if (t + 1 < steps) SP_GLDS_LANDED();
    __syncthreads();
    ++t;
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
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const km_f32x4 ch4 = *(const km_f32x4*)(chb + i * 32 + 8 * q);   // rows i*32 + 8q + 4lh + (0..3)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v = ch4[e] - acc[i][j][4 * q + e];
            const bool better = v < best[j];
            second[j] = __builtin_amdgcn_fmed3f(best[j], second[j], v);
            bpos[j] = better ? 16 * i + 4 * q + e : bpos[j];
            best[j] = fminf(best[j], v);
          }
      }
#pragma unroll
    for (int j = 0; j < 2; ++j) bblk[j] = best[j] < before[j] ? tm : bblk[j];
  }
#undef KS_LOAD

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
          if (blockIdx.y == 0) part[(int64_t)gridDim.y * 3 * ldp + m0 + col] = xn2[m0 + col];
        } else {
          km_decide(b, s, ix, xn2[m0 + col], first_point + m0 + col, ef, cmax, cmax2, labels, amb_rows, amb_best,
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
  hipLaunchKernelGGL(sp_col_finish_kernel, dim3((unsigned)((w.dp + 255) / 256)), dim3(256), 0, st, (const float*)w.colsum,
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

// First pass of the split tier: centers prepared as for the fp32 tier (Cf, |c|^2/2, max |c|^2) and cut into their two
// images, then the whole rounds, the split last round and its merge -- the launch plan of sp_nearest_fused_launch.
static int sp_nearest_split_launch(const void* C, int32_t cdtype, int64_t ldc, int64_t n, int64_t k, int64_t d,
                                   int64_t* labels, const KmWorkspace& w, hipStream_t st) {
  const int64_t kp = w.kp, dp = w.dp;
  if (dp * 2 * (int64_t)KS_BM > (1LL << 31) / 2) SP_FAIL("sp_nearest_center: too many features for the split tier");
  SP_HIP(hipMemsetAsync(w.Cf, 0, km_align((size_t)dp * kp * 4), st));
  SP_HIP(hipMemsetAsync(w.cmax2, 0, 512, st));   // cmax2 and amb_count
  const unsigned pblocks = (unsigned)((kp + 3) / 4);   // one wavefront per center
  if (cdtype == SP_F32)
    hipLaunchKernelGGL((sp_centers_prep_shifted_kernel<float>), dim3(pblocks), dim3(256), 0, st, (const float*)C, ldc,
                       (int)k, (int)d, (int)kp, (int)dp, (const float*)w.mu, w.Cf, w.cn, w.cmax2);
  else
    hipLaunchKernelGGL((sp_centers_prep_shifted_kernel<double>), dim3(pblocks), dim3(256), 0, st, (const double*)C, ldc,
                       (int)k, (int)d, (int)kp, (int)dp, (const float*)w.mu, w.Cf, w.cn, w.cmax2);
  hipLaunchKernelGGL(sp_split_rows_kernel, dim3(pblocks), dim3(256), 0, st, (const float*)w.Cf, dp, kp, (int)dp, (int)dp,
                     (const float*)nullptr, w.Ch, w.Cm, (float*)nullptr);
  SP_CHECK_LAUNCH();
  const int64_t blocks = (n + KS_BN - 1) / KS_BN, tiles = kp / KS_BM;
  int64_t rem = blocks % KM_WG_SLOTS, split = 1, per = tiles;
  static const bool tail_off = getenv("SP_KM_TAIL_SPLIT") && atoi(getenv("SP_KM_TAIL_SPLIT")) == 0;
  if (rem > 0 && tiles > 1 && !tail_off) {
    split = tiles < KM_TAIL_SPLIT ? tiles : KM_TAIL_SPLIT;
    per = (tiles + split - 1) / split;
    split = (tiles + per - 1) / per;
    const int64_t rounds = (rem * split + KM_WG_SLOTS - 1) / KM_WG_SLOTS;
    if (rounds * per >= tiles) split = 1;     // no shorter than the plain round
  }
  if (split == 1) rem = 0;
  const int64_t whole = blocks - rem, n_whole = whole * KS_BN < n ? whole * KS_BN : n;
  if (whole > 0)
    hipLaunchKernelGGL((sp_nearest_split_kernel<false, false>), dim3((unsigned)whole), dim3(256), 0, st, w.Xh, w.Xm, w.xn2,
                       w.Ch, w.Cm, w.cn, w.cmax2, (int)n_whole, (int)d, (int)dp, (int)kp, labels, w.amb_rows, w.amb_best,
                       w.amb_count, (unsigned*)nullptr, (float*)nullptr, 0, 0, 0);
  if (rem > 0) {
    const int n_tail = (int)(n - n_whole);
    hipLaunchKernelGGL((sp_nearest_split_kernel<false, true>), dim3((unsigned)rem, (unsigned)split), dim3(256), 0, st,
                       w.Xh + n_whole * dp, w.Xm + n_whole * dp, w.xn2 + n_whole, w.Ch, w.Cm, w.cn, w.cmax2, n_tail, (int)d,
                       (int)dp, (int)kp, (int64_t*)nullptr, (int*)nullptr, (float*)nullptr, (int*)nullptr,
                       (unsigned*)nullptr, w.part, KM_TAIL_POINTS, (int)per, 0);
    hipLaunchKernelGGL(sp_nearest_merge_parts_kernel, dim3((unsigned)((n_tail + 255) / 256)), dim3(256), 0, st, w.part,
                       (int)split, KM_TAIL_POINTS, n_tail, (int)n_whole, 6.1f * (float)d + 1550.0f, w.cmax2, labels,
                       w.amb_rows, w.amb_best, w.amb_count);
  }
  SP_CHECK_LAUNCH();
  return 0;
}

// Second pass over the listed points: marks the centers inside each point's error window.
static int sp_nearest_split_mark_candidates(int64_t d, const KmWorkspace& w, hipStream_t st) {
  const dim3 grid((unsigned)((w.cand_cap + KS_BN - 1) / KS_BN), (unsigned)(w.kp / KS_BM));
  hipLaunchKernelGGL((sp_nearest_split_kernel<true, false>), grid, dim3(256), 0, st, w.Xh, w.Xm, w.xn2, w.Ch, w.Cm, w.cn,
                     w.cmax2, (int)w.cand_cap, (int)d, (int)w.dp, (int)w.kp, (int64_t*)nullptr, w.amb_rows, w.amb_best,
                     w.amb_count, w.cand_mask, (float*)nullptr, 0, 0, 0);
  SP_CHECK_LAUNCH();
  return 0;
}

}  // namespace

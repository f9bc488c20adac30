// Fused tier of sp_nearest_center: fp32 MFMA GEMM  score[i][c] = |c|^2 - 2 x_i.c
// with the per-point argmin folded into the epilogue, so the n x k distance matrix
// (40 GB at BASELINE configs[3]) is never materialised.  (|x_i|^2 is constant per
// row and irrelevant to the argmin.)
//
// One workgroup (4 waves, 2x2) owns 128 points and walks ALL centers in blocks
// of 128, each block a full contraction over the features in BK=16 steps; the
// (center block, k-step) pairs form ONE software pipeline (global -> VGPR -> LDS
// double buffer, one barrier per step, as in gemm.hip).  Per accumulator row slot
// every lane keeps (best, second best, best's column) over the columns it has seen;
// lanes and the two column waves are merged once at the end.
//
// Exactness: a point is final only if its two best scores differ by more than
// 4E, E = u((2D+4)|x||c|max + 2|c|max^2), u = 2^-24: the fp32 error bound of one
// score (centers rounded to fp32, fmaf chain of D terms, |c|^2 rounded once, one
// final rounding).  Otherwise its label is written as -1-best and the exact fp64
// kernel (kmeans.hip) re-does that point, so the labels equal the exact tier's.
#pragma once

typedef float km_f32x16 __attribute__((ext_vector_type(16)));
typedef float km_f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int KM_BM = 128, KM_BK = 16;
constexpr int KM_BN_MAX = 256;   // centers are padded to a multiple of this (any column-block width divides it)
constexpr int KM_LDA = KM_BK + 4;
constexpr int KM_A_FLOATS = KM_BM * KM_LDA;

// Ct[j][c] = (float)C[c][j] (zero padded to [dp][kp]); cn[c] = |C[c]|^2 / 2 (fp64 sum, rounded; +inf on padding);
// *cmax2 = max_c |C[c]|^2 (as float bits; non-negative floats order like unsigned ints)
template <typename TC>
__global__ __launch_bounds__(256) void sp_centers_prep_kernel(const TC* __restrict__ C, int64_t ldc, int k, int d,
                                                              int kp, float* __restrict__ Ct,
                                                              float* __restrict__ cn, unsigned* __restrict__ cmax2) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= kp) return;
  if (c >= k) {
    cn[c] = INFINITY;
    return;
  }
  double s = 0.0;
  for (int j = 0; j < d; ++j) {
    const double v = (double)C[(int64_t)c * ldc + j];
    s += v * v;
    Ct[(int64_t)j * kp + c] = (float)v;
  }
  const float sf = (float)s;
  cn[c] = 0.5f * sf;   // the kernel compares halved scores |c|^2/2 - x.c (exact scaling)
  // round the max up so the bound stays a bound
  atomicMax(cmax2, __float_as_uint(sf * 1.0000002f));
}

// TN: 32-column MFMA tiles per wave along the centers (column block of the workgroup = 64 * TN)
template <bool FAST, int TN>
__global__ __launch_bounds__(256, 2) void sp_nearest_fused_kernel(const float* __restrict__ X, int64_t ldx,
                                                                  const float* __restrict__ Ct,
                                                                  const float* __restrict__ chalf,
                                                                  const unsigned* __restrict__ cmax2_bits, int n,
                                                                  int d, int kp, int64_t* __restrict__ labels,
                                                                  int* __restrict__ amb_rows,
                                                                  float* __restrict__ amb_best,
                                                                  int* __restrict__ amb_count) {
  constexpr int KM_BN = 64 * TN, KM_B_FLOATS = KM_BK * KM_BN, KM_STAGE = KM_A_FLOATS + KM_B_FLOATS;
  constexpr int BV = TN;   // float4 of B per thread per k-step
  __shared__ __attribute__((aligned(16))) float smem[2 * KM_STAGE];
  __shared__ float xn_s[KM_BM];
  __shared__ float mb_s[KM_BM], ms_s[KM_BM];
  __shared__ int mi_s[KM_BM];
  constexpr int THREADS = 256;
  constexpr int KQ = KM_BK / 4, NQ = KM_BN / 4;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.x * KM_BM;

  const float* __restrict__ Ablk = X + (int64_t)m0 * ldx;
  int a_off[2], a_lds[2], b_off[BV], b_lds[BV];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int e = tid + j * THREADS;
    int row = e / KQ;
    const int kq = e % KQ;
    a_lds[j] = row * KM_LDA + kq * 4;
    if (m0 + row > n - 1) row = n - 1 - m0;   // clamp: results of rows >= n are discarded
    a_off[j] = row * (int)ldx + kq * 4;
  }
#pragma unroll
  for (int j = 0; j < BV; ++j) {
    const int e = tid + j * THREADS;
    const int brow = e / NQ, nq = e % NQ;
    b_lds[j] = brow * KM_BN + nq * 4;
    b_off[j] = brow * kp + nq * 4;
  }
  const int nt = (d + KM_BK - 1) / KM_BK;
  const int tiles_n = kp / KM_BN;
  const int steps = nt * tiles_n;
  km_f32x4 ra[2], rb[BV];

#define KM_LOAD(step)                                                                    \
  do {                                                                                   \
    const int tn_ = (step) / nt, kt_ = (step) - tn_ * nt;                                \
    const int k0_ = kt_ * KM_BK;                                                         \
    const float* Bk_ = Ct + (int64_t)k0_ * kp + tn_ * KM_BN;                             \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                      \
      if constexpr (FAST) {                                                              \
        ra[j] = *(const km_f32x4*)(Ablk + k0_ + a_off[j]);                               \
      } else {                                                                           \
        const int kk = k0_ + ((tid + j * THREADS) % KQ) * 4;                             \
        const float* p = Ablk + k0_ + a_off[j];                                          \
        ra[j].x = kk + 0 < d ? p[0] : 0.f;                                               \
        ra[j].y = kk + 1 < d ? p[1] : 0.f;                                               \
        ra[j].z = kk + 2 < d ? p[2] : 0.f;                                               \
        ra[j].w = kk + 3 < d ? p[3] : 0.f;                                               \
      }                                                                                  \
    }                                                                                    \
    _Pragma("unroll") for (int j = 0; j < BV; ++j) rb[j] = *(const km_f32x4*)(Bk_ + b_off[j]); \
  } while (0)
#define KM_STORE(buf)                                                                    \
  do {                                                                                   \
    float* sA_ = smem + (buf) * KM_STAGE;                                                \
    float* sB_ = sA_ + KM_A_FLOATS;                                                      \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) *(km_f32x4*)(sA_ + a_lds[j]) = ra[j];  \
    _Pragma("unroll") for (int j = 0; j < BV; ++j) *(km_f32x4*)(sB_ + b_lds[j]) = rb[j]; \
  } while (0)

  km_f32x16 acc[2][TN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float best[2][16], second[2][16];
  // column of `best` = 32 * (its 32-column tile id) + l31
  int btile[2][16];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      best[i][r] = INFINITY;
      second[i][r] = INFINITY;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) btile[i][r] = 0;
  }
  float xs[2] = {0.f, 0.f};   // partial |x|^2 of rows wm*64 + i*32 + l31 (this lane's k slots)

  KM_LOAD(0);
  KM_STORE(0);
  __syncthreads();
  const int a_frag = (wm * 64 + l31) * KM_LDA + 4 * lh;
  const int b_frag = (4 * lh) * KM_BN + wn * (32 * TN) + l31;

  float chv[TN];   // |c|^2 / 2 of this lane's two columns of the current center block (fetched a block ahead of use)
#pragma unroll
  for (int j = 0; j < TN; ++j) chv[j] = chalf[(wn * TN + j) * 32 + l31];
  int t = 0;
  for (int tn = 0; tn < tiles_n; ++tn) {
    // ---- contraction over the features for center block tn: one straight-line body per k-step
    for (int kt = 0; kt < nt; ++kt, ++t) {
      if (t + 1 < steps) KM_LOAD(t + 1);
      const float* sA = smem + (t & 1) * KM_STAGE;
      const float* sB = sA + KM_A_FLOATS;
#pragma unroll
      for (int c = 0; c < KM_BK / 8; ++c) {
        km_f32x4 af[2];
        float bf[TN][4];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = *(const km_f32x4*)(sA + a_frag + i * 32 * KM_LDA + c * 8);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int s = 0; s < 4; ++s) bf[j][s] = sB[b_frag + (c * 8 + s) * KM_BN + j * 32];
        // |x|^2 is accumulated on every pass over the point block (no branch in this loop) and
        // divided by the number of passes at the end; it only feeds the error bound
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int s = 0; s < 4; ++s) xs[i] = __builtin_fmaf(af[i][s], af[i][s], xs[i]);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
      }
      if (t + 1 < steps) KM_STORE((t + 1) & 1);
      __syncthreads();
    }
    // ---- epilogue of center block tn, branch-free (the if/else form compiles to one exec-masked
    // basic block per accumulator).  Scores are kept halved, h = |c|^2/2 - x.c; columns ascend
    // with j, so `<` keeps the first minimum.  Invariant: best <= second.
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int tile = tn * (2 * TN) + wn * TN + j;
      const float ch = chv[j];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = ch - acc[i][j][r];
          const bool better = v < best[i][r];
          second[i][r] = fminf(second[i][r], fmaxf(v, best[i][r]));
          btile[i][r] = better ? tile : btile[i][r];
          best[i][r] = fminf(best[i][r], v);
          acc[i][j][r] = 0.f;
        }
    }
    if (tn + 1 < tiles_n) {
#pragma unroll
      for (int j = 0; j < TN; ++j) chv[j] = chalf[((tn + 1) * (2 * TN) + wn * TN + j) * 32 + l31];
    }
  }
#undef KM_LOAD
#undef KM_STORE

  // |x|^2 per row -> LDS (the two lane halves hold complementary k slots)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float tot = (xs[i] + __shfl_xor(xs[i], 32)) / (float)tiles_n;
    if (wn == 0 && lh == 0) xn_s[wm * 64 + i * 32 + l31] = tot;
  }
  // merge the 32 column lanes of each row slot (lanes with the same lh)
  int bidx[2][16];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float b = best[i][r], s = second[i][r];
      int ix = btile[i][r] * 32 + l31;
#pragma unroll
      for (int off = 1; off < 32; off <<= 1) {
        const float ob = __shfl_xor(b, off), os = __shfl_xor(s, off);
        const int oi = __shfl_xor(ix, off);
        if (ob < b || (ob == b && oi < ix)) {
          s = fminf(b, os);
          b = ob;
          ix = oi;
        } else {
          s = fminf(ob, s);
        }
      }
      best[i][r] = b;
      second[i][r] = s;
      bidx[i][r] = ix;
    }
  // merge the two column waves through LDS, then decide
  if (wn == 1 && l31 == 0) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        mb_s[row] = best[i][r];
        ms_s[row] = second[i][r];
        mi_s[row] = bidx[i][r];
      }
  }
  __syncthreads();
  if (wn == 0 && l31 == 0) {
    const float cmax2 = __uint_as_float(*cmax2_bits);
    const float cmax = sqrtf(cmax2) * 1.0000002f;
    const float u = 5.9604645e-8f;   // 2^-24
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        float b = best[i][r], s = second[i][r];
        int ix = bidx[i][r];
        const float ob = mb_s[row], os = ms_s[row];
        const int oi = mi_s[row];
        if (ob < b || (ob == b && oi < ix)) {
          s = fminf(b, os);
          b = ob;
          ix = oi;
        } else {
          s = fminf(ob, s);
        }
        if (m0 + row < n) {
          const float xnorm = sqrtf(xn_s[row]) * 1.001f;   // fp32 sum of squares: generous slack
          const float E = u * ((2.0f * (float)d + 4.0f) * xnorm * cmax + 2.0f * cmax2);
          const bool sure = 2.0f * (s - b) > 4.0f * E;     // (scores are halved) false for NaN / inf-inf as well
          labels[m0 + row] = sure ? (int64_t)ix : (int64_t)(-1 - ix);
          if (!sure) {   // (order-free: each listed point is re-done on its own)
            const int pos = atomicAdd(amb_count, 1);
            amb_rows[pos] = m0 + row;
            amb_best[pos] = b;   // its best (halved) fp32 score: the re-check's candidate window starts here
          }
        }
      }
  }
}

// ---- point block RESIDENT in LDS (d <= 256) ------------------------------------------------------
// The streaming kernel above re-reads its 128-point block from L2/HBM once per center block (PMC:
// 8.9 GB fetched per call at configs[3] for 1.28 GB of points).  MI355X has 160 KB of LDS per CU: the
// whole block [128][d] (133 KB at d = 256) is loaded ONCE and stays; only the center tiles stream
// (L2-resident, 1 MB in all).  One workgroup per CU, 8 waves (4 x 2), wave tile 32 x 64 -- two waves per
// SIMD, so one wave's LDS reads / epilogue hide behind the other's MFMAs.
// MEASURED (profiles/r01_notes.md): HBM fetch drops from 8.9 GB to 1.56 GB per call (1.28 GB algorithmic), but the
// kernel is SLOWER (95 vs 107 TFLOP/s at d = 256, 82 vs 94 at d = 128): the re-reads were not the limiter,
// one workgroup per CU with 32 x 64 wave tiles is.  Kept selectable (SP_KM_ARES=1), not the default.
constexpr int KA_THREADS = 512, KA_BN = 128;
constexpr int KA_B_FLOATS = KM_BK * KA_BN;

template <bool FAST>
__global__ __launch_bounds__(KA_THREADS, 1) void sp_nearest_ares_kernel(const float* __restrict__ X, int64_t ldx,
                                                                        const float* __restrict__ Ct,
                                                                        const float* __restrict__ chalf,
                                                                        const unsigned* __restrict__ cmax2_bits,
                                                                        int n, int d, int dp, int kp,
                                                                        int64_t* __restrict__ labels,
                                                                        int* __restrict__ amb_rows,
                                                                        float* __restrict__ amb_best,
                                                                        int* __restrict__ amb_count) {
  extern __shared__ __attribute__((aligned(16))) float dsm[];
  __shared__ float xn_s[KM_BM];
  __shared__ float mb_s[KM_BM], ms_s[KM_BM];
  __shared__ int mi_s[KM_BM];
  const int lda = dp + 4;                       // row pad: 8 consecutive rows cover the 32 LDS banks in 16-B units
  float* sAres = dsm;                           // [128][lda]
  float* sBst = dsm + KM_BM * lda;              // 2 x [16][128]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;        // 4 x 2 waves
  const int l31 = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.x * KM_BM;

  // ---- the point block, once: coalesced 16-B loads along the features
  {
    const int q4 = dp / 4;                      // float4 per row
    for (int e = tid; e < KM_BM * q4; e += KA_THREADS) {
      const int row = e / q4, kq = e - row * q4;
      int grow = m0 + row;
      if (grow > n - 1) grow = n - 1;           // clamp: results of rows >= n are discarded
      const float* p = X + (int64_t)grow * ldx + kq * 4;
      km_f32x4 v;
      if constexpr (FAST) {
        v = *(const km_f32x4*)p;
      } else {
        const int kk = kq * 4;
        v.x = kk + 0 < d ? p[0] : 0.f;
        v.y = kk + 1 < d ? p[1] : 0.f;
        v.z = kk + 2 < d ? p[2] : 0.f;
        v.w = kk + 3 < d ? p[3] : 0.f;
      }
      *(km_f32x4*)(sAres + row * lda + kq * 4) = v;
    }
  }
  // B tile loads: 16 x 128 floats = 512 float4, one per thread
  const int b_row = tid / (KA_BN / 4), b_nq = tid % (KA_BN / 4);
  const int b_lds = b_row * KA_BN + b_nq * 4;
  const int b_off = b_row * kp + b_nq * 4;
  const int nt = dp / KM_BK;
  const int tiles_n = kp / KA_BN;
  const int steps = nt * tiles_n;
  km_f32x4 rb;
#define KA_LOAD(step)                                                              \
  do {                                                                             \
    const int tn_ = (step) / nt, kt_ = (step) - tn_ * nt;                          \
    rb = *(const km_f32x4*)(Ct + (int64_t)(kt_ * KM_BK) * kp + tn_ * KA_BN + b_off); \
  } while (0)
#define KA_STORE(buf) *(km_f32x4*)(sBst + (buf) * KA_B_FLOATS + b_lds) = rb

  KA_LOAD(0);
  KA_STORE(0);
  __syncthreads();
  // |x|^2 per row from the resident block (feeds the error bound only)
  for (int r = wid; r < KM_BM; r += KA_THREADS / 64) {
    float s = 0.f;
    for (int k = lane; k < dp; k += 64) {
      const float v = sAres[r * lda + k];
      s = __builtin_fmaf(v, v, s);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) xn_s[r] = s;
  }

  km_f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float best[16], second[16];
  int btile[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    best[r] = INFINITY;
    second[r] = INFINITY;
    btile[r] = 0;
  }
  const int a_frag = (wm * 32 + l31) * lda + 4 * lh;
  const int b_frag = (4 * lh) * KA_BN + wn * 64 + l31;
  float chv[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) chv[j] = chalf[(wn * 2 + j) * 32 + l31];
  int t = 0;
  for (int tn = 0; tn < tiles_n; ++tn) {
    for (int kt = 0; kt < nt; ++kt, ++t) {
      if (t + 1 < steps) KA_LOAD(t + 1);
      const float* sB = sBst + (t & 1) * KA_B_FLOATS;
      const float* sA = sAres + a_frag + kt * KM_BK;
#pragma unroll
      for (int c = 0; c < KM_BK / 8; ++c) {
        const km_f32x4 af = *(const km_f32x4*)(sA + c * 8);
        float bf[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int s = 0; s < 4; ++s) bf[j][s] = sB[b_frag + (c * 8 + s) * KA_BN + j * 32];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], bf[j][s], acc[j], 0, 0, 0);
      }
      if (t + 1 < steps) KA_STORE((t + 1) & 1);
      __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int tile = tn * 4 + wn * 2 + j;
      const float ch = chv[j];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = ch - acc[j][r];
        const bool better = v < best[r];
        second[r] = fminf(second[r], fmaxf(v, best[r]));
        btile[r] = better ? tile : btile[r];
        best[r] = fminf(best[r], v);
        acc[j][r] = 0.f;
      }
    }
    if (tn + 1 < tiles_n) {
#pragma unroll
      for (int j = 0; j < 2; ++j) chv[j] = chalf[((tn + 1) * 4 + wn * 2 + j) * 32 + l31];
    }
  }
#undef KA_LOAD
#undef KA_STORE

  int bidx[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float b = best[r], s = second[r];
    int ix = btile[r] * 32 + l31;
#pragma unroll
    for (int off = 1; off < 32; off <<= 1) {
      const float ob = __shfl_xor(b, off), os = __shfl_xor(s, off);
      const int oi = __shfl_xor(ix, off);
      if (ob < b || (ob == b && oi < ix)) {
        s = fminf(b, os);
        b = ob;
        ix = oi;
      } else {
        s = fminf(ob, s);
      }
    }
    best[r] = b;
    second[r] = s;
    bidx[r] = ix;
  }
  if (wn == 1 && l31 == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      mb_s[row] = best[r];
      ms_s[row] = second[r];
      mi_s[row] = bidx[r];
    }
  }
  __syncthreads();
  if (wn == 0 && l31 == 0) {
    const float cmax2 = __uint_as_float(*cmax2_bits);
    const float cmax = sqrtf(cmax2) * 1.0000002f;
    const float u = 5.9604645e-8f;   // 2^-24
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      float b = best[r], s = second[r];
      int ix = bidx[r];
      const float ob = mb_s[row], os = ms_s[row];
      const int oi = mi_s[row];
      if (ob < b || (ob == b && oi < ix)) {
        s = fminf(b, os);
        b = ob;
        ix = oi;
      } else {
        s = fminf(ob, s);
      }
      if (m0 + row < n) {
        const float xnorm = sqrtf(xn_s[row]) * 1.001f;
        const float E = u * ((2.0f * (float)d + 4.0f) * xnorm * cmax + 2.0f * cmax2);
        const bool sure = 2.0f * (s - b) > 4.0f * E;
        labels[m0 + row] = sure ? (int64_t)ix : (int64_t)(-1 - ix);
        if (!sure) {
          const int pos = atomicAdd(amb_count, 1);
          amb_rows[pos] = m0 + row;
          amb_best[pos] = b;
        }
      }
    }
  }
}

static inline int64_t km_round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

// scratch layout of sp_nearest_center (all 256-B aligned)
struct KmWorkspace {
  int64_t kp, dp;
  double* Ct64;      // [d][kp]   fp64 transposed centers (exact kernel)
  float* Ct;         // [dp][kp]  fp32 transposed centers (fused kernel)
  float* cn;         // [kp]      |c|^2
  unsigned* cmax2;   // [1]       max |c|^2 (float bits)
  int* amb_count;    // [1]
  int* amb_rows;     // [n]       points the fused kernel could not decide
  float* amb_best;   // [n]       their best (halved) fp32 score
};

static inline size_t km_align(size_t v) { return (v + 255) & ~(size_t)255; }

static size_t sp_nearest_fused_ws_bytes(int64_t n, int64_t k, int64_t d) {
  const int64_t kp = km_round_up(k < 1 ? 1 : k, KM_BN_MAX), dp = km_round_up(d < 1 ? 1 : d, KM_BK);
  return 256 + km_align((size_t)(d < 1 ? 1 : d) * kp * 8) + km_align((size_t)dp * kp * 4) + km_align((size_t)kp * 4) + 256 + 256 +
         2 * km_align((size_t)(n < 1 ? 1 : n) * 4);
}

static KmWorkspace km_carve(void* ws, int64_t n, int64_t k, int64_t d) {
  KmWorkspace w;
  w.kp = km_round_up(k, KM_BN_MAX);
  w.dp = km_round_up(d < 1 ? 1 : d, KM_BK);
  char* p = (char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  w.Ct64 = (double*)p;
  p += km_align((size_t)(d < 1 ? 1 : d) * w.kp * 8);
  w.Ct = (float*)p;
  p += km_align((size_t)w.dp * w.kp * 4);
  w.cn = (float*)p;
  p += km_align((size_t)w.kp * 4);
  w.cmax2 = (unsigned*)p;
  p += 256;
  w.amb_count = (int*)p;
  p += 256;
  w.amb_rows = (int*)p;
  p += km_align((size_t)(n < 1 ? 1 : n) * 4);
  w.amb_best = (float*)p;
  return w;
}

// the fused tier pays once the contraction is big enough to hide its fixed costs
static bool sp_nearest_fused_applicable(int64_t n, int64_t k, int64_t d, int tier) {
  if (n > 2147483647LL - KM_BM || d < 1 || k > (1LL << 20)) return false;
  if (tier == SP_NEAREST_FUSED || tier == SP_NEAREST_FUSED_UNCHECKED) return true;
  return n >= 1024 && k >= 16 && d >= 8 && n * k * d >= (1LL << 24);
}

static int sp_nearest_fused_launch(const float* X, int64_t ldx, const void* C, int32_t cdtype, int64_t ldc,
                                   int64_t n, int64_t k, int64_t d, int64_t* labels, const KmWorkspace& w,
                                   hipStream_t st) {
  if (ldx > 2147483647LL / KM_BM) SP_FAIL("sp_nearest_center: leading dimension too large for the fused tier");
  const int64_t kp = w.kp, dp = w.dp;
  float* Ct = w.Ct;
  float* cn = w.cn;
  unsigned* cmax2 = w.cmax2;
  SP_HIP(hipMemsetAsync(Ct, 0, (size_t)dp * kp * 4, st));
  SP_HIP(hipMemsetAsync(cmax2, 0, 512, st));   // cmax2 and amb_count
  const unsigned pblocks = (unsigned)((kp + 255) / 256);
  if (cdtype == SP_F32)
    hipLaunchKernelGGL((sp_centers_prep_kernel<float>), dim3(pblocks), dim3(256), 0, st, (const float*)C, ldc, (int)k,
                       (int)d, (int)kp, Ct, cn, cmax2);
  else
    hipLaunchKernelGGL((sp_centers_prep_kernel<double>), dim3(pblocks), dim3(256), 0, st, (const double*)C, ldc, (int)k,
                       (int)d, (int)kp, Ct, cn, cmax2);
  SP_CHECK_LAUNCH();
  const unsigned blocks = (unsigned)((n + KM_BM - 1) / KM_BM);
  const bool fast = (d % KM_BK == 0) && (ldx % 4 == 0) && ((((uintptr_t)X) & 15) == 0);
  // d <= 256: the point block fits the CU's LDS next to the streaming center tiles -> resident variant
  static int ares_env = -1;
  if (ares_env < 0) {
    const char* e = getenv("SP_KM_ARES");
    ares_env = e ? atoi(e) : 0;   // measured slower than the streaming kernel (see the kernel's comment): opt-in
  }
  if (ares_env && dp <= 256) {
    const size_t lds = (size_t)(KM_BM * (dp + 4) + 2 * KA_B_FLOATS) * 4;
    static bool attr_set[2] = {false, false};
    if (fast) {
      auto kfn = sp_nearest_ares_kernel<true>;
      if (!attr_set[0]) {
        SP_HIP(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096));
        attr_set[0] = true;
      }
      hipLaunchKernelGGL(kfn, dim3(blocks), dim3(KA_THREADS), lds, st, X, ldx, Ct, cn, cmax2, (int)n, (int)d, (int)dp,
                         (int)kp, labels, w.amb_rows, w.amb_best, w.amb_count);
    } else {
      auto kfn = sp_nearest_ares_kernel<false>;
      if (!attr_set[1]) {
        SP_HIP(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096));
        attr_set[1] = true;
      }
      hipLaunchKernelGGL(kfn, dim3(blocks), dim3(KA_THREADS), lds, st, X, ldx, Ct, cn, cmax2, (int)n, (int)d, (int)dp,
                         (int)kp, labels, w.amb_rows, w.amb_best, w.amb_count);
    }
    SP_CHECK_LAUNCH();
    return 0;
  }
  // column block of a workgroup: 128 centers (2 MFMA tiles per wave).  256 (SP_KM_TN=4) needs 47 spilled
  // registers at 2 workgroups/CU and measured 1.7 % slower (profiles/r01_notes.md)
  static int tn_env = -1;
  if (tn_env < 0) {
    const char* e = getenv("SP_KM_TN");
    tn_env = e ? atoi(e) : 0;
  }
  const int tn_sel = tn_env == 4 ? 4 : 2;
#define KM_GO(F, T)                                                                                              \
  hipLaunchKernelGGL((sp_nearest_fused_kernel<F, T>), dim3(blocks), dim3(256), 0, st, X, ldx, Ct, cn, cmax2, (int)n, \
                     (int)d, (int)kp, labels, w.amb_rows, w.amb_best, w.amb_count)
  if (fast) {
    if (tn_sel == 4) KM_GO(true, 4);
    else KM_GO(true, 2);
  } else {
    if (tn_sel == 4) KM_GO(false, 4);
    else KM_GO(false, 2);
  }
#undef KM_GO
  SP_CHECK_LAUNCH();
  return 0;
}

}  // namespace

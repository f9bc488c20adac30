// fp64 GEMM on the CDNA4 f64 MFMA (v_mfma_f64_16x16x4_f64, 78.6 TFLOP/s dense on MI355X):
// `tiles[0].dot(tiles[1])` of the reference's dot mappers (spartan/expr/dot.py:172-238) for the
// reference's DEFAULT dtype -- its builders (rand, zeros ...) make float64 arrays
// (spartan/expr/srandom.py:84, creation.py) -- so that a float64 `spartan.dot` does not fall
// back to the untiled multiply-reduce launch.
//
// Same structure as gemm.hip: 128x128 macro-tile, K in steps of 8, 4 waves (2x2), each wave a
// 4x4 grid of 16x16 MFMA tiles (64 accumulator doubles per lane), global -> VGPR -> LDS double
// buffer with one barrier per k-step, XCD-aware grouped tile order.  A is kept [m][k] in LDS with
// a one-double row pad (row stride 72 B: the 16 rows a quarter-wave reads hit 16 distinct bank
// pairs), B [k][n].
#include "sp_common.hpp"

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int DBM = 128, DBN = 128, DBK = 8;
constexpr int DLDA = DBK + 1;
constexpr int DA_DOUBLES = DBM * DLDA, DB_DOUBLES = DBK * DBN, DSTAGE = DA_DOUBLES + DB_DOUBLES;
constexpr int DGROUP_M = 1;   // row-major walk inside an XCD's range (see SP_GEMM_GROUP_M in gemm.hip)

__device__ __forceinline__ void dgemm_tile_of_block(int bid, int nblk, int tiles_m, int tiles_n, int& tm, int& tn) {
  const int xcd = bid & 7, local = bid >> 3;
  const int q = nblk >> 3, r = nblk & 7;
  const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
  const int per_group = DGROUP_M * tiles_n;
  const int group = t / per_group;
  const int first_m = group * DGROUP_M;
  const int gsize = (tiles_m - first_m) < DGROUP_M ? (tiles_m - first_m) : DGROUP_M;
  const int in_group = t - group * per_group;
  tm = first_m + (in_group % gsize);
  tn = in_group / gsize;
}

// FAST: K % 8 == 0, N % 2 == 0, lda/ldb % 2 == 0, 16-B aligned bases.
template <bool FAST, bool SPLIT = false>
__global__ __launch_bounds__(256, 2) void sp_dgemm_kernel(const double* __restrict__ A, int64_t lda,
                                                          const double* __restrict__ B, int64_t ldb,
                                                          double* __restrict__ C, int64_t ldc, int M, int N, int K,
                                                          int accumulate, int tiles_m, int tiles_n, int ksplit_len,
                                                          int64_t c_split_stride) {
  __shared__ __attribute__((aligned(16))) double smem[2 * DSTAGE];
  constexpr int THREADS = 256;
  constexpr int KQ = DBK / 2;   // 16-B pieces per A row
  constexpr int NQ = DBN / 2;   // 16-B pieces per B row
  int tm, tn;
  dgemm_tile_of_block(blockIdx.x, gridDim.x, tiles_m, tiles_n, tm, tn);
  const int m0 = tm * DBM, n0 = tn * DBN;
  if constexpr (SPLIT) {   // split-K slice (sp_gemm_ws)
    const int kb = blockIdx.y * ksplit_len;
    A += kb;
    B += (int64_t)kb * ldb;
    C += (int64_t)blockIdx.y * c_split_stride;
    K = (K - kb) < ksplit_len ? (K - kb) : ksplit_len;
  }
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int l15 = lane & 15, lq = lane >> 4;   // MFMA: row/col index and k (or row-group) index

  const double* __restrict__ Ablk = A + (int64_t)m0 * lda;
  const double* __restrict__ Bblk = B + n0;
  int a_off[2], a_lds[2], b_off[2], b_lds[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int e = tid + j * THREADS;
    int row = e / KQ;
    const int kq = e % KQ;
    a_lds[j] = row * DLDA + kq * 2;
    if (m0 + row > M - 1) row = M - 1 - m0;
    a_off[j] = row * (int)lda + kq * 2;
    const int brow = e / NQ, nq = e % NQ;
    b_lds[j] = brow * DBN + nq * 2;
    int gc = nq * 2;
    if (FAST && n0 + gc > N - 2) gc = N - 2 - n0;
    b_off[j] = brow * (int)ldb + gc;
  }
  f64x2 ra[2], rb[2];

#define DG_LOAD(kt)                                                                  \
  do {                                                                               \
    const int k0_ = (kt) * DBK;                                                      \
    const double* Ak_ = Ablk + k0_;                                                  \
    const double* Bk_ = Bblk + (int64_t)k0_ * ldb;                                   \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                  \
      if constexpr (FAST) {                                                          \
        ra[j] = *(const f64x2*)(Ak_ + a_off[j]);                                     \
        rb[j] = *(const f64x2*)(Bk_ + b_off[j]);                                     \
      } else {                                                                       \
        const int e_ = tid + j * THREADS;                                            \
        const bool rok = (m0 + e_ / KQ) < M;                                         \
        const int kk = k0_ + (e_ % KQ) * 2;                                          \
        const double* p = Ak_ + a_off[j];                                            \
        ra[j].x = (rok && kk + 0 < K) ? p[0] : 0.0;                                  \
        ra[j].y = (rok && kk + 1 < K) ? p[1] : 0.0;                                  \
        const bool kok = (k0_ + e_ / NQ) < K;                                        \
        const int cc = n0 + (e_ % NQ) * 2;                                           \
        const double* q = Bk_ + b_off[j];                                            \
        rb[j].x = (kok && cc + 0 < N) ? q[0] : 0.0;                                  \
        rb[j].y = (kok && cc + 1 < N) ? q[1] : 0.0;                                  \
      }                                                                              \
    }                                                                                \
  } while (0)
#define DG_STORE(buf)                                                                \
  do {                                                                               \
    double* sA_ = smem + (buf) * DSTAGE;                                             \
    double* sB_ = sA_ + DA_DOUBLES;                                                  \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                  \
      sA_[a_lds[j]] = ra[j].x;                                                       \
      sA_[a_lds[j] + 1] = ra[j].y;                                                   \
      *(f64x2*)(sB_ + b_lds[j]) = rb[j];                                             \
    }                                                                                \
  } while (0)

  f64x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0;

  const int nt = (K + DBK - 1) / DBK;
  DG_LOAD(0);
  DG_STORE(0);
  __syncthreads();
  // A fragment: A[wm*64 + i*16 + l15][k = c*4 + lq]; B fragment: B[k = c*4 + lq][wn*64 + j*16 + l15]
  const int a_frag = (wm * 64 + l15) * DLDA + lq;
  const int b_frag = lq * DBN + wn * 64 + l15;

  for (int t = 0; t < nt; ++t) {
    if (t + 1 < nt) DG_LOAD(t + 1);
    const double* sA = smem + (t & 1) * DSTAGE;
    const double* sB = sA + DA_DOUBLES;
#pragma unroll
    for (int c = 0; c < DBK / 4; ++c) {
      double af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = sA[a_frag + i * 16 * DLDA + c * 4];
#pragma unroll
      for (int j = 0; j < 4; ++j) bf[j] = sB[b_frag + c * 4 * DBN + j * 16];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (t + 1 < nt) DG_STORE((t + 1) & 1);
    __syncthreads();
  }
#undef DG_LOAD
#undef DG_STORE

  // C/D layout of the 16x16 f64 tile: col = lane & 15, row = (lane >> 4) + 4 * r
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = n0 + wn * 64 + j * 16 + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * 64 + i * 16 + lq + 4 * r;
        if (row < M && col < N) {
          double* p = C + (int64_t)row * ldc + col;
          double v = acc[i][j][r];
          if (accumulate) v += *p;
          *p = v;
        }
      }
    }
}

}  // namespace

extern "C" int sp_gemm_f64(const double* d_A, int64_t lda, const double* d_B, int64_t ldb, double* d_C, int64_t ldc,
                           int64_t M, int64_t N, int64_t K, int32_t accumulate, void* stream) {
  if (M < 0 || N < 0 || K < 0) SP_FAIL("sp_gemm_f64: negative dimension");
  if (M == 0 || N == 0) return 0;
  if (!d_A || !d_B || !d_C) SP_FAIL("sp_gemm_f64: NULL pointer");
  if (M > 2147483647LL || N > 2147483647LL || K > 2147483647LL) SP_FAIL("sp_gemm_f64: dimension too large");
  if (lda < K || ldb < N || ldc < N) SP_FAIL("sp_gemm_f64: leading dimension too small");
  if (lda > 2147483647LL / DBM || ldb > 2147483647LL / DBK) SP_FAIL("sp_gemm_f64: leading dimension too large");
  hipStream_t st = (hipStream_t)stream;
  if (K == 0) {
    if (!accumulate) SP_HIP(hipMemset2DAsync(d_C, (size_t)ldc * 8, 0, (size_t)N * 8, (size_t)M, st));
    return 0;
  }
  const int64_t tiles_m = (M + DBM - 1) / DBM, tiles_n = (N + DBN - 1) / DBN;
  const int64_t nblk = tiles_m * tiles_n;
  if (nblk > 2147483647LL) SP_FAIL("sp_gemm_f64: too many tiles");
  const bool fast = (K % DBK == 0) && (N % 2 == 0) && (N >= 2) && (lda % 2 == 0) && (ldb % 2 == 0) &&
                    ((((uintptr_t)d_A) | ((uintptr_t)d_B)) & 15) == 0;
  if (fast)
    hipLaunchKernelGGL((sp_dgemm_kernel<true>), dim3((unsigned)nblk), dim3(256), 0, st, d_A, lda, d_B, ldb, d_C, ldc,
                       (int)M, (int)N, (int)K, accumulate, (int)tiles_m, (int)tiles_n, 0, (int64_t)0);
  else
    hipLaunchKernelGGL((sp_dgemm_kernel<false>), dim3((unsigned)nblk), dim3(256), 0, st, d_A, lda, d_B, ldb, d_C, ldc,
                       (int)M, (int)N, (int)K, accumulate, (int)tiles_m, (int)tiles_n, 0, (int64_t)0);
  SP_CHECK_LAUNCH();
  return 0;
}


// split-K launch used by sp_gemm_ws (gemm.hip): slice s of the contraction -> part[s][M][N]
int sp_dgemm_split_launch(const double* A, int64_t lda, const double* B, int64_t ldb, double* part, int64_t M,
                          int64_t N, int64_t K, int splits, int klen, hipStream_t st) {
  const int64_t tiles_m = (M + DBM - 1) / DBM, tiles_n = (N + DBN - 1) / DBN;
  const int64_t nblk = tiles_m * tiles_n;
  const bool fast = (K % DBK == 0) && (N % 2 == 0) && (N >= 2) && (lda % 2 == 0) && (ldb % 2 == 0) &&
                    ((((uintptr_t)A) | ((uintptr_t)B)) & 15) == 0;
  if (fast)
    hipLaunchKernelGGL((sp_dgemm_kernel<true, true>), dim3((unsigned)nblk, (unsigned)splits), dim3(256), 0, st, A, lda, B, ldb,
                       part, N, (int)M, (int)N, (int)K, 0, (int)tiles_m, (int)tiles_n, klen, M * N);
  else
    hipLaunchKernelGGL((sp_dgemm_kernel<false, true>), dim3((unsigned)nblk, (unsigned)splits), dim3(256), 0, st, A, lda, B, ldb,
                       part, N, (int)M, (int)N, (int)K, 0, (int)tiles_m, (int)tiles_n, klen, M * N);
  SP_CHECK_LAUNCH();
  return 0;
}

// fp32 GEMM on the CDNA4 f32-input MFMA (v_mfma_f32_32x32x2_f32): exact fp32
// (bitwise an fmaf chain in k order), 157.3 TFLOP/s peak on MI355X.
//
// Replaces `tiles[0].dot(tiles[1])` of the reference's dot mappers
// (spartan/expr/dot.py:195-217 dot_map2_mapper, :222-238 dot_outer_mapper,
// :172-187 dot_map2_np_mapper); `accumulate` fuses the np.add reducer of the
// dot target (dot.py:289-294 + tile.pyx:263-266) into the epilogue.
//
// Three kernels share the macro-tile, the MFMA schedule and the epilogue: sp_gemm_glds_kernel (direct-to-LDS
// k-tiles, the default for large aligned problems, see its comment), sp_gemm_glds_sk_kernel (the same k-loop over
// whole data-parallel rounds of tiles plus equal ranges of the remaining tiles' k-tiles, for tile counts that do not
// fill the chip; sp_gemm_ws picks it by a cost model) and sp_gemm_kernel (register-staged, any shape / alignment /
// K tail / split-K).
// Structure of sp_gemm_kernel (per workgroup of NW waves, one 32x32 MFMA tile grid per wave):
//   - BM x BN output macro-tile, K walked in steps of BK=16;
//   - global -> VGPR (16-B loads) -> LDS, LDS double-buffered: the loads of
//     k-tile t+1 are issued before the MFMAs of k-tile t and written to the
//     other LDS buffer after them: one workgroup barrier per k-tile;
//   - A is kept [m][k] in LDS with a 16-B row pad so the per-lane 16-B fragment
//     reads (ds_read_b128) are bank-conflict free; the MFMA k-slots are
//     permuted (lane half h takes k = 4h..4h+3 of each 8-wide chunk for both A
//     and B), which is legal because the contraction index order only has to
//     agree between A and B;
//   - B is kept [k][n]; fragments are 4 x ds_read_b32 (lanes along n);
//   - XCD-aware block -> tile mapping: each XCD walks a contiguous range of the
//     tile order, so the 64 workgroups resident on one XCD are neighbours and share
//     their A panel in that XCD's private L2 (see SP_GEMM_GROUP_M).
#include "sp_common.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Tiles are walked m-fastest inside groups of SP_GEMM_GROUP_M tile rows.  Measured on 8192^3 (PMC, 2 x FETCH_SIZE,
// the direct-to-LDS kernel of rounds 3+; round 6, tools/r06/gemm_group.sh): group 1 / 2 / 3 / 4 / 8 -> 9.19 / 6.76 /
// 7.44 / 9.24 / 17.3 GB of L2 misses at the SAME speed (150.8 / 150.8 / 150.8 / 150.7 / 150.3 TFLOP/s; 32768^3,
// 16384^3, 6144^3, 4096^3 and the pipeline's chunk shapes within 0.3 %): with two tile rows per group the 64
// workgroups resident on an XCD are 2 A panels x 32 B panels instead of 1 x 64 -- half the B traffic for twice the
// (small) A traffic; beyond 2 the workgroups of a group drift too far apart in k for the L2 to serve the shared
// panels.  (Round 2, on the register-staged kernel, had measured 1 as the minimum: 10.1 GB against 10.6 for 2.)
// 2 is the default since round 6: counter traffic at 8192^3 11.7 x -> 8.7 x the 12 n^2 floor.
#ifndef SP_GEMM_ABLATE
#define SP_GEMM_ABLATE 0      // build with -DSP_GEMM_ABLATE=1 for the timing-only variants (tools/gemm_variants.py)
#endif
#ifndef SP_GEMM_GROUP_M
#define SP_GEMM_GROUP_M 2
#endif

template <int BM, int BN, int BK, int WM, int WN>
struct GemmCfg {
  static constexpr int NW = WM * WN;
  static constexpr int THREADS = NW * 64;
  static constexpr int WTM = BM / WM, WTN = BN / WN;  // wave tile
  static constexpr int TM = WTM / 32, TN = WTN / 32;  // 32x32 MFMA tiles per wave
  static constexpr int LDA_S = BK + 4;                // padded LDS row (floats)
  static constexpr int LDB_S = BN;
  static constexpr int A_FLOATS = BM * LDA_S;
  static constexpr int B_FLOATS = BK * LDB_S;
  static constexpr int STAGE_FLOATS = A_FLOATS + B_FLOATS;
  static constexpr int LDS_BYTES = 2 * STAGE_FLOATS * 4;
  static constexpr int A_VEC = (BM * BK / 4) / THREADS;  // float4 per thread per k-tile
  static constexpr int B_VEC = (BK * BN / 4) / THREADS;
  // 2 workgroups per CU when the wave tile needs 128 accumulator registers:
  // asks the allocator for <= 256 VGPR+AGPR so barrier stalls of one workgroup
  // are covered by the other's MFMAs.
  static constexpr int MIN_WAVES = (NW == 4) ? 2 : 1;
  static_assert(BK % 8 == 0, "BK must be a multiple of 8");
  static_assert((BM * BK / 4) % THREADS == 0 && (BK * BN / 4) % THREADS == 0, "tile/threads mismatch");
};

// block id -> (tile_m, tile_n), XCD-aware + grouped
__device__ __forceinline__ void sp_gemm_tile_of_block(int bid, int nblk, int tiles_m, int tiles_n,
                                                      int& tm, int& tn) {
  // bijective XCD remap (guide T1): blocks bid, bid+8, ... run on one XCD
  const int xcd = bid & 7, local = bid >> 3;
  const int q = nblk >> 3, r = nblk & 7;
  const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
  const int per_group = SP_GEMM_GROUP_M * tiles_n;
  const int group = t / per_group;
  const int first_m = group * SP_GEMM_GROUP_M;
  const int gsize = (tiles_m - first_m) < SP_GEMM_GROUP_M ? (tiles_m - first_m) : SP_GEMM_GROUP_M;
  const int in_group = t - group * per_group;
  tm = first_m + (in_group % gsize);
  tn = in_group / gsize;
}

// FAST: N % 4 == 0, lda/ldb % 4 == 0, 16-B aligned bases (a K % BK tail is guarded on its own k-tile).
// Rows beyond M / columns beyond N are clamped on load and masked on store.
// KTAIL: FAST kernel whose last k-tile is partial; SPLIT: split-K slice per blockIdx.y (sp_gemm_ws).
// The hot instantiation (FAST, no tail, no split) carries none of that code.
template <typename Cfg, int BM, int BN, int BK, int WM, int WN, bool FAST, bool KTAIL = false, bool SPLIT = false>
__global__ __launch_bounds__(Cfg::THREADS, Cfg::MIN_WAVES) void sp_gemm_kernel(const float* __restrict__ A, int64_t lda,
                                                               const float* __restrict__ B, int64_t ldb,
                                                               float* __restrict__ C, int64_t ldc, int M,
                                                               int N, int K, int accumulate, int tiles_m,
                                                               int tiles_n, int ksplit_len, int64_t c_split_stride) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int THREADS = Cfg::THREADS;
  constexpr int TM = Cfg::TM, TN = Cfg::TN;
  constexpr int LDA_S = Cfg::LDA_S, LDB_S = Cfg::LDB_S;
  constexpr int A_VEC = Cfg::A_VEC, B_VEC = Cfg::B_VEC;
  constexpr int KQ = BK / 4;  // float4 per A row
  constexpr int NQ = BN / 4;  // float4 per B row

  int tm, tn;
  sp_gemm_tile_of_block(blockIdx.x, gridDim.x, tiles_m, tiles_n, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  if constexpr (SPLIT) {
    // split-K: slice blockIdx.y of the contraction goes to its own partial output (sp_gemm_ws)
    const int kb = blockIdx.y * ksplit_len;
    A += kb;
    B += (int64_t)kb * ldb;
    C += (int64_t)blockIdx.y * c_split_stride;
    K = (K - kb) < ksplit_len ? (K - kb) : ksplit_len;
  }

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int l31 = lane & 31, lh = lane >> 5;

  // ---- global load assignments: block-relative 32-bit offsets, scalar bases
  const float* __restrict__ Ablk = A + (int64_t)m0 * lda;
  const float* __restrict__ Bblk = B + n0;
  int a_off[A_VEC], a_lds[A_VEC];
#pragma unroll
  for (int j = 0; j < A_VEC; ++j) {
    const int e = tid + j * THREADS;
    int row = e / KQ;
    const int kq = e % KQ;
    a_lds[j] = row * LDA_S + kq * 4;
    if (m0 + row > M - 1) row = M - 1 - m0;  // clamp (masked on store)
    a_off[j] = row * (int)lda + kq * 4;
  }
  int b_off[B_VEC], b_lds[B_VEC];
#pragma unroll
  for (int j = 0; j < B_VEC; ++j) {
    const int e = tid + j * THREADS;
    const int row = e / NQ;
    const int nq = e % NQ;
    b_lds[j] = row * LDB_S + nq * 4;
    int gc = nq * 4;
    if (FAST && n0 + gc > N - 4) gc = N - 4 - n0;
    b_off[j] = row * (int)ldb + gc;
  }

  f32x4 ra[A_VEC], rb[B_VEC];

#define SP_GEMM_LOAD_TILE(kt)                                                         \
  do {                                                                                \
    const int k0_ = (kt) * BK;                                                        \
    const float* Ak_ = Ablk + k0_;                                                    \
    const float* Bk_ = Bblk + (int64_t)k0_ * ldb;                                     \
    _Pragma("unroll") for (int j = 0; j < A_VEC; ++j) {                               \
      if constexpr (FAST) {                                                           \
        ra[j] = *(const f32x4*)(Ak_ + a_off[j]);                                     \
      } else {                                                                        \
        const int e_ = tid + j * THREADS;                                             \
        const bool rok = (m0 + e_ / KQ) < M;                                          \
        const int kk = k0_ + (e_ % KQ) * 4;                                           \
        const float* p = Ak_ + a_off[j];                                              \
        ra[j].x = (rok && kk + 0 < K) ? p[0] : 0.f;                                   \
        ra[j].y = (rok && kk + 1 < K) ? p[1] : 0.f;                                   \
        ra[j].z = (rok && kk + 2 < K) ? p[2] : 0.f;                                   \
        ra[j].w = (rok && kk + 3 < K) ? p[3] : 0.f;                                   \
      }                                                                               \
    }                                                                                 \
    _Pragma("unroll") for (int j = 0; j < B_VEC; ++j) {                               \
      if constexpr (FAST) {                                                           \
        rb[j] = *(const f32x4*)(Bk_ + b_off[j]);                                     \
      } else {                                                                        \
        const int e_ = tid + j * THREADS;                                             \
        const bool kok = (k0_ + e_ / NQ) < K;                                         \
        const int cc = n0 + (e_ % NQ) * 4;                                            \
        const float* p = Bk_ + b_off[j];                                              \
        rb[j].x = (kok && cc + 0 < N) ? p[0] : 0.f;                                   \
        rb[j].y = (kok && cc + 1 < N) ? p[1] : 0.f;                                   \
        rb[j].z = (kok && cc + 2 < N) ? p[2] : 0.f;                                   \
        rb[j].w = (kok && cc + 3 < N) ? p[3] : 0.f;                                   \
      }                                                                               \
    }                                                                                 \
  } while (0)

// FAST kernels, last k-tile when K % BK != 0: rows / columns are clamped as usual, elements with
// k >= K are zero and are not read
#define SP_GEMM_LOAD_KTAIL(kt)                                                        \
  do {                                                                                \
    const int k0_ = (kt) * BK;                                                        \
    const float* Ak_ = Ablk + k0_;                                                    \
    const float* Bk_ = Bblk + (int64_t)k0_ * ldb;                                     \
    _Pragma("unroll") for (int j = 0; j < A_VEC; ++j) {                               \
      const int kk = k0_ + ((tid + j * THREADS) % KQ) * 4;                            \
      const float* p = Ak_ + a_off[j];                                                \
      ra[j].x = kk + 0 < K ? p[0] : 0.f;                                              \
      ra[j].y = kk + 1 < K ? p[1] : 0.f;                                              \
      ra[j].z = kk + 2 < K ? p[2] : 0.f;                                              \
      ra[j].w = kk + 3 < K ? p[3] : 0.f;                                              \
    }                                                                                 \
    _Pragma("unroll") for (int j = 0; j < B_VEC; ++j) {                               \
      const bool kok = (k0_ + (tid + j * THREADS) / NQ) < K;                          \
      if (kok) rb[j] = *(const f32x4*)(Bk_ + b_off[j]);                               \
      else rb[j] = f32x4{0.f, 0.f, 0.f, 0.f};                                         \
    }                                                                                 \
  } while (0)

#define SP_GEMM_STORE_TILE(buf)                                                       \
  do {                                                                                \
    float* sA_ = smem + (buf) * Cfg::STAGE_FLOATS;                                    \
    float* sB_ = sA_ + Cfg::A_FLOATS;                                                 \
    _Pragma("unroll") for (int j = 0; j < A_VEC; ++j) *(f32x4*)(sA_ + a_lds[j]) = ra[j]; \
    _Pragma("unroll") for (int j = 0; j < B_VEC; ++j) *(f32x4*)(sB_ + b_lds[j]) = rb[j]; \
  } while (0)

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nt = (K + BK - 1) / BK;
  constexpr bool ktail = FAST && KTAIL;   // (the general path guards every element anyway)
  if (ktail && nt == 1) SP_GEMM_LOAD_KTAIL(0);
  else SP_GEMM_LOAD_TILE(0);
  SP_GEMM_STORE_TILE(0);
  __syncthreads();

  const int a_frag_off = (wm * Cfg::WTM + l31) * LDA_S + 4 * lh;
  const int b_frag_off = (4 * lh) * LDB_S + wn * Cfg::WTN + l31;

  for (int t = 0; t < nt; ++t) {
    if (t + 1 < nt) {
      if (ktail && t + 2 == nt) SP_GEMM_LOAD_KTAIL(t + 1);
      else SP_GEMM_LOAD_TILE(t + 1);
    }
    const float* sA = smem + (t & 1) * Cfg::STAGE_FLOATS;
    const float* sB = sA + Cfg::A_FLOATS;
#pragma unroll
    for (int c = 0; c < BK / 8; ++c) {
      f32x4 af[TM];
      float bf[TN][4];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *(const f32x4*)(sA + a_frag_off + i * 32 * LDA_S + c * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int s = 0; s < 4; ++s) bf[j][s] = sB[b_frag_off + (c * 8 + s) * LDB_S + j * 32];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const float av = af[i][s];
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bf[j][s], acc[i][j], 0, 0, 0);
        }
      }
    }
    if (t + 1 < nt) SP_GEMM_STORE_TILE((t + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * Cfg::WTN + j * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * Cfg::WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < M && col < N) {
          float* p = C + (int64_t)row * ldc + col;
          float v = acc[i][j][r];
          if (accumulate) v += *p;
          *p = v;
        }
      }
    }
  }
}

// ---- direct-to-LDS variant (round 2) -------------------------------------------------------------------
// The same macro-tile and MFMA schedule, but the k-tiles go from global memory straight into LDS
// (global_load_lds_dwordx4: 16 B per lane, LDS address = wave-uniform base + 16 * lane), so there are no staging
// registers, no ds_write pass and no wait for the global loads inside the k-loop: tile t+1 is requested right
// after the barrier that frees its stage and only has to have landed at the NEXT barrier, a whole k-tile of MFMAs
// later.  Such a load cannot pad or scatter its LDS image, so the A tile is kept [m][16] UNPADDED and made
// conflict-free by permuting which 16-B chunk of a row each lane fetches (the permutation is on the SOURCE
// address; the LDS image stays lane-linear): chunk q of row m lives in slot q ^ ((m >> 2) & 3).  A ds_read_b128
// is served in four groups of 16 lanes whose rows fall into the four residues mod 4 four times each; rows of one
// residue share their 16 banks, and the xor sends those four rows to four different chunks = four different bank
// quads.  The B tile [16][BN] is read along n by ds_read_b32 exactly as before.
// Preconditions (sp_gemm_f32 checks them, else the register-staged kernel runs): N % 4 == 0, lda % 4 == 0,
// ldb % 4 == 0, 16-B aligned bases, K % 16 == 0.
template <int BM, int BN, int WM, int WN>
struct GldsCfg {
  static constexpr int BK = 16;
  static constexpr int NW = WM * WN, THREADS = NW * 64;
  static constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
  static constexpr int A_FLOATS = BM * BK, B_FLOATS = BK * BN, STAGE_FLOATS = A_FLOATS + B_FLOATS;
  static constexpr int LDS_BYTES = 2 * STAGE_FLOATS * 4;    // (3 stages for the deep-prefetch variant)
  static constexpr int A_PIECES = (A_FLOATS / 256) / NW;   // 1 KiB wave-loads of A per wave and k-tile
  static constexpr int B_PIECES = (B_FLOATS / 256) / NW;
  static constexpr int MIN_WAVES = (NW == 4) ? 2 : 1;
  static_assert((A_FLOATS / 256) % NW == 0 && (B_FLOATS / 256) % NW == 0, "tile / waves mismatch");
};

// PIPE: the fragments of the two 8-deep halves of a k-tile live in two register sets; the half-tile that follows is
// read from LDS while the current one is multiplied, and the workgroup barrier sits between the two halves (the
// reads that must precede it were issued a half-tile of MFMAs earlier, the reads that follow it have one to land).
// WGS: workgroups per CU the register allocation aims for.
template <typename Cfg, int BM, int BN, int WM, int WN, int PIPE, int WGS>
__global__ __launch_bounds__(Cfg::THREADS, WGS) void sp_gemm_glds_kernel(
    const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb, float* __restrict__ C,
    int64_t ldc, int M, int N, int K, int accumulate, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int BK = Cfg::BK, TM = Cfg::TM, TN = Cfg::TN;
  constexpr int AP = Cfg::A_PIECES, BP = Cfg::B_PIECES;
  int tm, tn;
  sp_gemm_tile_of_block(blockIdx.x, gridDim.x, tiles_m, tiles_n, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // in an SGPR: the LDS targets of the loads are scalar
  const int wm = wid / WN, wn = wid % WN;
  const int l31 = lane & 31, lh = lane >> 5;

  // ---- what each lane fetches: block-relative UNSIGNED 32-bit byte offsets off scalar bases, so that a load is
  // `global_load_lds_dwordx4 v_off, s[base]` with nothing to compute per k-tile but the scalar base
  const char* __restrict__ Ablk = (const char*)(A + (int64_t)m0 * lda);
  const char* __restrict__ Bblk = (const char*)(B + n0);
  unsigned a_off[AP], b_off[BP];
#pragma unroll
  for (int j = 0; j < AP; ++j) {
    const int slot = (wid * AP + j) * 64 + lane;          // 16-B slot of the A image: row = slot / 4
    int row = slot >> 2;
    const int q = (slot & 3) ^ ((row >> 2) & 3);           // the chunk of that row this slot holds
    if (m0 + row > M - 1) row = M - 1 - m0;                // clamp (masked on store)
    a_off[j] = (unsigned)(row * (int)lda + q * 4) * 4u;
  }
#pragma unroll
  for (int j = 0; j < BP; ++j) {
    const int slot = (wid * BP + j) * 64 + lane;          // 16-B slot of the B image: row = slot / (BN / 4)
    const int krow = slot / (BN / 4);
    int gc = (slot % (BN / 4)) * 4;
    if (n0 + gc > N - 4) gc = N - 4 - n0;
    b_off[j] = (unsigned)(krow * (int)ldb + gc) * 4u;
  }
  const unsigned sA_w = SP_LDS_ADDR(smem) + wid * (AP * 1024);            // this wave's pieces inside a stage (bytes)
  const unsigned sB_w = SP_LDS_ADDR(smem) + Cfg::A_FLOATS * 4 + wid * (BP * 1024);

#define SP_GLDS_TILE(kt, stage)                                                        \
  do {                                                                                 \
    const char* Ak_ = Ablk + (int64_t)(kt) * (BK * 4);                                 \
    const char* Bk_ = Bblk + (int64_t)((kt) * BK) * ldb * 4;                           \
    const unsigned dA_ = sA_w + (stage) * (Cfg::STAGE_FLOATS * 4);                     \
    const unsigned dB_ = sB_w + (stage) * (Cfg::STAGE_FLOATS * 4);                     \
    _Pragma("unroll") for (int j = 0; j < AP; ++j) SP_GLDS_S(Ak_, a_off[j], dA_ + j * 1024); \
    _Pragma("unroll") for (int j = 0; j < BP; ++j) SP_GLDS_S(Bk_, b_off[j], dB_ + j * 1024); \
  } while (0)

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment addresses: A row (wave tile row + l31), chunk (lh + 2c) ^ ((row >> 2) & 3); B row 4 lh + 8c + s
  const int sw = (l31 >> 2) & 3;
  const int a_row_off = (wm * Cfg::WTM + l31) * BK;
  int a_chunk[BK / 8];
#pragma unroll
  for (int c = 0; c < BK / 8; ++c) a_chunk[c] = a_row_off + 4 * ((lh + 2 * c) ^ sw);
  const int b_frag_off = (4 * lh) * BN + wn * Cfg::WTN + l31;

  const int nt = K / BK;
#define SP_FRAGS(af_, bf_, stage, c)                                                                \
  do {                                                                                              \
    const float* sA_ = smem + (stage) * Cfg::STAGE_FLOATS;                                          \
    const float* sB_ = sA_ + Cfg::A_FLOATS;                                                         \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) af_[i] = *(const f32x4*)(sA_ + a_chunk[c] + i * 32 * BK); \
    _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                  \
      _Pragma("unroll") for (int s4 = 0; s4 < 4; ++s4) bf_[j][s4] = sB_[b_frag_off + ((c) * 8 + s4) * BN + j * 32]; \
  } while (0)
#define SP_MFMAS(af_, bf_)                                                                          \
  do {                                                                                              \
    _Pragma("unroll") for (int s4 = 0; s4 < 4; ++s4)                                                \
      _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                              \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af_[i][s4], bf_[j][s4], acc[i][j], 0, 0, 0); \
  } while (0)
  if constexpr (PIPE >= 16) {
    SP_GLDS_TILE(0, 0);
    SP_GLDS_LANDED();
    __syncthreads();
  } else if constexpr (PIPE == 3) {
    // Three LDS stages, k-tiles requested TWO ahead: a tile has two k-tiles of MFMAs to land, so the tail of the
    // load latency distribution (one slow 1 KiB piece of 24 holds the whole workgroup at its barrier) is covered.
    // The wait is counted -- the AP + BP pieces of the newest tile stay in flight across the barrier -- which
    // needs the raw barrier: __syncthreads() carries a fence that drains the LDS-DMA queue.
    static_assert(AP + BP == 6 || AP + BP == 4, "vmcnt immediates below");
#define SP_WAIT_BUT_NEWEST()                                                   \
  do {                                                                         \
    if constexpr (AP + BP == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); \
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                      \
  } while (0)
    auto body = [&](int t) {
#pragma unroll
      for (int c = 0; c < BK / 8; ++c) {
        f32x4 af[TM];
        float bf[TN][4];
        SP_FRAGS(af, bf, t % 3, c);
        SP_MFMAS(af, bf);
      }
    };
    SP_GLDS_TILE(0, 0);
    if (nt > 1) SP_GLDS_TILE(1, 1);
    if (nt > 1) SP_WAIT_BUT_NEWEST();
    else SP_GLDS_LANDED();
    __builtin_amdgcn_s_barrier();
    int t = 0;
    for (; t + 2 < nt; ++t) {
      SP_GLDS_TILE(t + 2, (t + 2) % 3);
      body(t);
      SP_WAIT_BUT_NEWEST();      // tile t+1 has landed; tile t+2 may still be on its way
      __builtin_amdgcn_s_barrier();
    }
    for (; t < nt; ++t) {
      body(t);
      SP_GLDS_LANDED();
      __builtin_amdgcn_s_barrier();
    }
  } else if constexpr (PIPE) {
    static_assert(BK == 16, "two fragment halves per k-tile");
    f32x4 af0[TM], af1[TM];
    float bf0[TN][4], bf1[TN][4];
    SP_GLDS_TILE(0, 0);
    if (nt > 1) SP_GLDS_TILE(1, 1);
    SP_GLDS_LANDED();
    __syncthreads();
    SP_FRAGS(af0, bf0, 0, 0);
    for (int t = 0; t < nt; ++t) {
      const int cur = t & 1;
      SP_FRAGS(af1, bf1, cur, 1);
      SP_MFMAS(af0, bf0);
      // keep the first half's MFMAs AHEAD of the barrier (hipcc otherwise sinks them below it and the wave stalls
      // on the LDS reads it has just issued): the barrier's lgkmcnt wait then finds them long finished
      __builtin_amdgcn_sched_barrier(0);
      SP_GLDS_LANDED();
      __syncthreads();    // every wave has read all of stage `cur`; tile t+1 has landed in the other stage
      if (t + 2 < nt) SP_GLDS_TILE(t + 2, cur);
      if (t + 1 < nt) SP_FRAGS(af0, bf0, cur ^ 1, 0);
      SP_MFMAS(af1, bf1);
    }
  } else {
    SP_GLDS_TILE(0, 0);
    SP_GLDS_LANDED();
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
      if (t + 1 < nt) SP_GLDS_TILE(t + 1, (t + 1) & 1);
#pragma unroll
      for (int c = 0; c < BK / 8; ++c) {
        f32x4 af[TM];
        float bf[TN][4];
        SP_FRAGS(af, bf, t & 1, c);
        SP_MFMAS(af, bf);
      }
      SP_GLDS_LANDED();
      __syncthreads();    // tile t+1 has landed and every wave is done reading stage t
    }
    // K tail (K % 16 != 0): the last, partial k-tile goes through registers into stage 0 in the same two images,
    // zero beyond K in BOTH operands (so that whatever lies behind a row end never meets a product)
    const int ktail = K - nt * BK;
    if (ktail > 0) {
      const int k0 = nt * BK;
      for (int e = tid; e < BM * BK; e += Cfg::THREADS) {
        const int r = e / BK, kk = e % BK;
        int row = m0 + r;
        if (row > M - 1) row = M - 1;
        const float v = kk < ktail ? A[(int64_t)row * lda + k0 + kk] : 0.f;
        smem[r * BK + (((kk >> 2) ^ ((r >> 2) & 3)) << 2) + (kk & 3)] = v;
      }
      for (int e = tid; e < BK * BN; e += Cfg::THREADS) {
        const int kk = e / BN, c = e % BN;
        int col = n0 + c;
        if (col > N - 1) col = N - 1;
        smem[Cfg::A_FLOATS + kk * BN + c] = kk < ktail ? B[(int64_t)(k0 + kk) * ldb + col] : 0.f;
      }
      __syncthreads();
#pragma unroll
      for (int c = 0; c < BK / 8; ++c) {
        f32x4 af[TM];
        float bf[TN][4];
        SP_FRAGS(af, bf, 0, c);
        SP_MFMAS(af, bf);
      }
    }
  }
#if SP_GEMM_ABLATE
  // Timing-only variants (SP_GEMM_VARIANT 16..23, results are WRONG by construction): which part of the k-loop the
  // MFMA pipe waits for.  Bit 0: no k-tile loads after the first; bit 1: no barriers; bit 2: fragments read once;
  // bit 3: loads issued but never waited for.
  if constexpr (PIPE >= 16) {
    constexpr int ABL = PIPE - 16;
    f32x4 af[BK / 8][TM];
    float bf[BK / 8][TN][4];
    for (int t = 0; t < nt; ++t) {
      if (!(ABL & 1) && t + 1 < nt) SP_GLDS_TILE(t + 1, (t + 1) & 1);
#pragma unroll
      for (int c = 0; c < BK / 8; ++c) {
        if (!(ABL & 4) || t == 0) SP_FRAGS(af[c], bf[c], t & 1, c);
        SP_MFMAS(af[c], bf[c]);
      }
      if (!(ABL & 2)) {
        if (!(ABL & 8)) SP_GLDS_LANDED();
        __syncthreads();
      }
    }
  }
#endif
#undef SP_FRAGS
#undef SP_MFMAS
#undef SP_GLDS_TILE

#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * Cfg::WTN + j * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * Cfg::WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < M && col < N) {
          float* p = C + (int64_t)row * ldc + col;
          float v = acc[i][j][r];
          if (accumulate) v += *p;
          *p = v;
        }
      }
    }
  }
}

template <int BM, int BN, int WM, int WN, int PIPE, int WGS>
static int sp_gemm_glds_launch(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                               int64_t M, int64_t N, int64_t K, int acc, hipStream_t st) {
  using Cfg = GldsCfg<BM, BN, WM, WN>;
  const int64_t tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const int64_t nblk = tiles_m * tiles_n;
  if (nblk > 2147483647LL) SP_FAIL("sp_gemm_f32: too many tiles");
  auto k = sp_gemm_glds_kernel<Cfg, BM, BN, WM, WN, PIPE, WGS>;
  constexpr int lds_bytes = (PIPE == 3 ? 3 : 2) * Cfg::STAGE_FLOATS * 4;
  static bool attr_set = false;
  if (!attr_set) {
    SP_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    attr_set = true;
  }
  hipLaunchKernelGGL(k, dim3((unsigned)nblk), dim3(Cfg::THREADS), lds_bytes, st, A, lda, B, ldb, C, ldc, (int)M,
                     (int)N, (int)K, acc, (int)tiles_m, (int)tiles_n);
  SP_CHECK_LAUNCH();
  return 0;
}

// ---- balanced ("stream-K") variant of the direct-to-LDS kernel ------------------------------------------
// Data-parallel tiling leaves CUs idle whenever the tile count is not a multiple of the resident workgroups
// (3072^3: 288 tiles of 256 x 128 on 512 slots; 5000^3: 800 on 512 -> the last round is 56 % full).  Here the unit
// of work is one k-tile of one output tile: the tiles * (K / 16) units are laid out tile-major and cut into
// gridDim.x EQUAL contiguous ranges, one per workgroup, W = all resident slots.  A range is a run of segments
// (tile, k-tile interval): a segment that spans its tile's whole contraction stores to C like the data-parallel
// kernel; a partial one (at most the first and the last of a range) stores its accumulators to one of the
// workgroup's two workspace slots, and sp_gemm_sk_fixup_kernel adds the partials of every cut tile in k order.
// No atomics, no inter-workgroup waits: the result does not depend on scheduling.  (Summation order differs from
// the data-parallel kernel's single k-ordered chain only at the cuts.)
template <typename Cfg, int BM, int BN, int WM, int WN, int WGS>
__global__ __launch_bounds__(Cfg::THREADS, WGS) void sp_gemm_glds_sk_kernel(
    const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb, float* __restrict__ C,
    int64_t ldc, int M, int N, int K, int accumulate, int tiles_n, unsigned tiles, float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int BK = Cfg::BK, TM = Cfg::TM, TN = Cfg::TN;
  constexpr int AP = Cfg::A_PIECES, BP = Cfg::B_PIECES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;
  const int l31 = lane & 31, lh = lane >> 5;
  const unsigned KT = (unsigned)(K / BK);
  const int ktail = K - (int)KT * BK;

  // logical workgroup index: the workgroups of one XCD (blockIdx & 7) take a contiguous run of tiles / ranges, so
  // that the tiles an XCD's L2 sees are neighbours (the same remap as sp_gemm_tile_of_block)
  const unsigned W = gridDim.x;
  unsigned lw;
  {
    const unsigned xcd = blockIdx.x & 7, local = blockIdx.x >> 3, q8 = W >> 3, r8 = W & 7;
    lw = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + local;
  }
  const unsigned rounds = tiles / W, tiles_full = rounds * W;
  const unsigned total = (tiles - tiles_full) * KT;
  const unsigned q = total / W, r = total % W;
  const unsigned first = lw * q + (lw < r ? lw : r);
  const unsigned end = first + q + (lw < r ? 1u : 0u);

  const unsigned sA_w = SP_LDS_ADDR(smem) + wid * (AP * 1024);
  const unsigned sB_w = SP_LDS_ADDR(smem) + Cfg::A_FLOATS * 4 + wid * (BP * 1024);
  const int sw = (l31 >> 2) & 3;
  const int a_row_off = (wm * Cfg::WTM + l31) * BK;
  int a_chunk[BK / 8];
#pragma unroll
  for (int c = 0; c < BK / 8; ++c) a_chunk[c] = a_row_off + 4 * ((lh + 2 * c) ^ sw);
  const int b_frag_off = (4 * lh) * BN + wn * Cfg::WTN + l31;

  unsigned round = 0;
  for (unsigned it = first;;) {
    unsigned tile, k0, len;
    const bool whole_rounds = round < rounds;
    if (whole_rounds) {            // the data-parallel rounds: whole tiles, every workgroup at the same k
      tile = round * W + lw;
      k0 = 0;
      len = KT;
      ++round;
    } else if (it < end) {         // this workgroup's range of the remainder
      tile = it / KT;
      k0 = it - tile * KT;
      tile += tiles_full;
      len = KT - k0;
      if (len > end - it) len = end - it;
    } else {
      break;
    }
    const int tm = (int)(tile / (unsigned)tiles_n), tn = (int)(tile - (unsigned)tm * (unsigned)tiles_n);
    const int m0 = tm * BM, n0 = tn * BN;
    const char* __restrict__ Ablk = (const char*)(A + (int64_t)m0 * lda) + (int64_t)k0 * (BK * 4);
    const char* __restrict__ Bblk = (const char*)(B + n0) + (int64_t)k0 * BK * ldb * 4;
    unsigned a_off[AP], b_off[BP];
#pragma unroll
    for (int j = 0; j < AP; ++j) {
      const int slot = (wid * AP + j) * 64 + lane;
      int row = slot >> 2;
      const int qq = (slot & 3) ^ ((row >> 2) & 3);
      if (m0 + row > M - 1) row = M - 1 - m0;
      a_off[j] = (unsigned)(row * (int)lda + qq * 4) * 4u;
    }
#pragma unroll
    for (int j = 0; j < BP; ++j) {
      const int slot = (wid * BP + j) * 64 + lane;
      const int krow = slot / (BN / 4);
      int gc = (slot % (BN / 4)) * 4;
      if (n0 + gc > N - 4) gc = N - 4 - n0;
      b_off[j] = (unsigned)(krow * (int)ldb + gc) * 4u;
    }
#define SP_SK_TILE(kt, stage)                                                          \
  do {                                                                                 \
    const char* Ak_ = Ablk + (int64_t)(kt) * (BK * 4);                                 \
    const char* Bk_ = Bblk + (int64_t)((kt) * BK) * ldb * 4;                           \
    const unsigned dA_ = sA_w + (stage) * (Cfg::STAGE_FLOATS * 4);                     \
    const unsigned dB_ = sB_w + (stage) * (Cfg::STAGE_FLOATS * 4);                     \
    _Pragma("unroll") for (int j = 0; j < AP; ++j) SP_GLDS_S(Ak_, a_off[j], dA_ + j * 1024); \
    _Pragma("unroll") for (int j = 0; j < BP; ++j) SP_GLDS_S(Bk_, b_off[j], dB_ + j * 1024); \
  } while (0)

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    SP_SK_TILE(0, 0);
    SP_GLDS_LANDED();
    __syncthreads();
    const int nt = (int)len;
    for (int t = 0; t < nt; ++t) {
      if (t + 1 < nt) SP_SK_TILE(t + 1, (t + 1) & 1);
      const float* sA_ = smem + (t & 1) * Cfg::STAGE_FLOATS;
      const float* sB_ = sA_ + Cfg::A_FLOATS;
#pragma unroll
      for (int c = 0; c < BK / 8; ++c) {
        f32x4 af[TM];
        float bf[TN][4];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *(const f32x4*)(sA_ + a_chunk[c] + i * 32 * BK);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) bf[j][s4] = sB_[b_frag_off + (c * 8 + s4) * BN + j * 32];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s4], bf[j][s4], acc[i][j], 0, 0, 0);
      }
      SP_GLDS_LANDED();
      __syncthreads();    // tile t+1 has landed and every wave is done reading stage t (and, after the last
                          // k-tile, with both stages: the next segment may overwrite them)
    }
#undef SP_SK_TILE
    if (ktail > 0 && k0 + len == KT) {
      // K % 16 != 0: the segment that ends its tile's contraction also takes the partial k-tile, through registers
      // into stage 0, zero beyond K in both operands (as in sp_gemm_glds_kernel)
      const int kbase = (int)KT * BK;
      for (int e = tid; e < BM * BK; e += Cfg::THREADS) {
        const int rr = e / BK, kk = e % BK;
        int row = m0 + rr;
        if (row > M - 1) row = M - 1;
        const float v = kk < ktail ? A[(int64_t)row * lda + kbase + kk] : 0.f;
        smem[rr * BK + (((kk >> 2) ^ ((rr >> 2) & 3)) << 2) + (kk & 3)] = v;
      }
      for (int e = tid; e < BK * BN; e += Cfg::THREADS) {
        const int kk = e / BN, c = e % BN;
        int col = n0 + c;
        if (col > N - 1) col = N - 1;
        smem[Cfg::A_FLOATS + kk * BN + c] = kk < ktail ? B[(int64_t)(kbase + kk) * ldb + col] : 0.f;
      }
      __syncthreads();
#pragma unroll
      for (int c = 0; c < BK / 8; ++c) {
        f32x4 af[TM];
        float bf[TN][4];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *(const f32x4*)(smem + a_chunk[c] + i * 32 * BK);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) bf[j][s4] = smem[Cfg::A_FLOATS + b_frag_off + (c * 8 + s4) * BN + j * 32];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s4], bf[j][s4], acc[i][j], 0, 0, 0);
      }
      __syncthreads();
    }

    // (the per-lane bases below pass through an empty asm so that the 128 row / column offsets of the two stores
    // are computed here and not hoisted out of the segment loop, where they would live across the k-loop and spill)
    if (len == KT) {
      int64_t row_b = m0 + wm * Cfg::WTM + 4 * lh;
      int col_b = n0 + wn * Cfg::WTN + l31;
      asm volatile("" : "+v"(row_b), "+v"(col_b));
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int col = col_b + j * 32;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int64_t row = row_b + i * 32 + (e & 3) + 8 * (e >> 2);
            if (row < M && col < N) {
              float* p = C + row * ldc + col;
              float v = acc[i][j][e];
              if (accumulate) v += *p;
              *p = v;
            }
          }
        }
      }
    } else {
      // the whole BM x BN tile image, row-major, into this workgroup's slot 0 (its first segment) or 1 (its last)
      int off_b = (wm * Cfg::WTM + 4 * lh) * BN + wn * Cfg::WTN + l31;
      asm volatile("" : "+v"(off_b));
      float* __restrict__ P = part + (size_t)(2 * lw + (it != first ? 1u : 0u)) * (BM * BN) + off_b;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
          for (int e = 0; e < 16; ++e) P[(i * 32 + (e & 3) + 8 * (e >> 2)) * BN + j * 32] = acc[i][j][e];
        }
      }
    }
    if (!whole_rounds) it += len;
  }
}

// Adds the partial images of every cut tile in k order (= workgroup order) and stores the tile; tiles one workgroup
// computed whole were stored by the main kernel.  One workgroup per 4096 tile elements.
template <int BM, int BN>
__global__ __launch_bounds__(256) void sp_gemm_sk_fixup_kernel(const float* __restrict__ part, float* __restrict__ C,
                                                               int64_t ldc, int M, int N, int accumulate, int tiles_n,
                                                               unsigned KT, unsigned tiles_full, unsigned total, unsigned W) {
  constexpr unsigned CHUNKS = BM * BN / 4096;
  const unsigned rt = blockIdx.x / CHUNKS, chunk = blockIdx.x % CHUNKS;     // rt: index among the remainder tiles
  const unsigned q = total / W, r = total % W;
  const unsigned lo = rt * KT, hi = lo + KT - 1;
  const unsigned edge = r * (q + 1);
  const unsigned w0 = lo < edge ? lo / (q + 1) : r + (lo - edge) / q;
  const unsigned w1 = hi < edge ? hi / (q + 1) : r + (hi - edge) / q;
  if (w0 == w1) return;
  const unsigned tile = tiles_full + rt;
  const int tm = (int)(tile / (unsigned)tiles_n), tn = (int)(tile - (unsigned)tm * (unsigned)tiles_n);
  const int m0 = tm * BM, n0 = tn * BN;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned e = chunk * 4096 + (i * 256 + threadIdx.x) * 4;
    const int row = m0 + (int)(e / BN), col = n0 + (int)(e % BN);
    if (row >= M || col >= N) continue;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (unsigned w = w0; w <= w1; ++w) {
      const unsigned start = w * q + (w < r ? w : r);
      const f32x4 p = *(const f32x4*)(part + (size_t)(2 * w + (start >= lo ? 0u : 1u)) * (BM * BN) + e);
      v = (w == w0) ? p : v + p;
    }
    f32x4* dst = (f32x4*)(C + (int64_t)row * ldc + col);
    if (accumulate) v += *dst;
    *dst = v;
  }
}

template <int BM, int BN, int WM, int WN, int WGS>
static int sp_gemm_sk_launch(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M,
                             int64_t N, int64_t K, int acc, float* part, hipStream_t st) {
  using Cfg = GldsCfg<BM, BN, WM, WN>;
  const int64_t tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN, tiles = tiles_m * tiles_n;
  const unsigned W = SP_CUS * WGS;
  const unsigned tiles_full = (unsigned)(tiles / W) * W, rem = (unsigned)tiles - tiles_full;
  auto k = sp_gemm_glds_sk_kernel<Cfg, BM, BN, WM, WN, WGS>;
  constexpr int lds_bytes = 2 * Cfg::STAGE_FLOATS * 4;
  static bool attr_set = false;
  if (!attr_set) {
    SP_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    attr_set = true;
  }
  hipLaunchKernelGGL(k, dim3(W), dim3(Cfg::THREADS), lds_bytes, st, A, lda, B, ldb, C, ldc, (int)M, (int)N, (int)K, acc,
                     (int)tiles_n, (unsigned)tiles, part);
  if (rem)
    hipLaunchKernelGGL((sp_gemm_sk_fixup_kernel<BM, BN>), dim3(rem * (BM * BN / 4096)), dim3(256), 0, st,
                       (const float*)part, C, ldc, (int)M, (int)N, acc, (int)tiles_n, (unsigned)(K / 16), tiles_full,
                       rem * (unsigned)(K / 16), W);
  SP_CHECK_LAUNCH();
  return 0;
}

template <typename Cfg, typename KernelT>
static int sp_gemm_go(KernelT k, bool* attr_set, unsigned nblk, unsigned splits, hipStream_t st, const float* A,
                      int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int M, int N, int K, int acc,
                      int tiles_m, int tiles_n, int ksplit_len, int64_t c_split_stride) {
  if (!*attr_set) {
    SP_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES));
    *attr_set = true;
  }
  hipLaunchKernelGGL(k, dim3(nblk, splits), dim3(Cfg::THREADS), Cfg::LDS_BYTES, st, A, lda, B, ldb, C, ldc, M, N, K,
                     acc, tiles_m, tiles_n, ksplit_len, c_split_stride);
  SP_CHECK_LAUNCH();
  return 0;
}

// SPLIT launches one slice of the contraction per blockIdx.y (only instantiated for the 128x128 tile).
template <int BM, int BN, int BK, int WM, int WN, bool SPLIT = false>
static int sp_gemm_launch(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                          int64_t M, int64_t N, int64_t K, int acc, bool fast, hipStream_t st, int splits = 1,
                          int ksplit_len = 0, int64_t c_split_stride = 0) {
  using Cfg = GemmCfg<BM, BN, BK, WM, WN>;
  const int64_t tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const int64_t nblk = tiles_m * tiles_n;
  if (nblk > 2147483647LL) SP_FAIL("sp_gemm_f32: too many tiles");
  static bool set_fast = false, set_tail = false, set_gen = false;
  // a slice of a split may end in a partial k-tile even when K itself does not: always the tail kernel there
  const bool tail = SPLIT ? true : (K % BK != 0);
#define SP_GEMM_ARGS (unsigned)nblk, (unsigned)splits, st, A, lda, B, ldb, C, ldc, (int)M, (int)N, (int)K, acc, \
                     (int)tiles_m, (int)tiles_n, ksplit_len, c_split_stride
  if (fast && !tail) return sp_gemm_go<Cfg>(sp_gemm_kernel<Cfg, BM, BN, BK, WM, WN, true, false, SPLIT>, &set_fast, SP_GEMM_ARGS);
  if (fast) return sp_gemm_go<Cfg>(sp_gemm_kernel<Cfg, BM, BN, BK, WM, WN, true, true, SPLIT>, &set_tail, SP_GEMM_ARGS);
  return sp_gemm_go<Cfg>(sp_gemm_kernel<Cfg, BM, BN, BK, WM, WN, false, false, SPLIT>, &set_gen, SP_GEMM_ARGS);
#undef SP_GEMM_ARGS
}

// Tuning knob (not part of the ABI contract): SP_GEMM_VARIANT=0..3 picks the
// macro-tile; unset = the default chosen from rocprof measurements
// (profiles/).
static int sp_gemm_variant() {
  static int v = -2;
  if (v == -2) {
    const char* e = getenv("SP_GEMM_VARIANT");
    v = e ? atoi(e) : -1;
  }
  return v;
}

// The data-parallel macro-tile whose workgroups keep the CUs busy for the shorter time, and that time in units of one
// 128 x 128 x K tile at the full CU rate.  A CU runs ceil(tiles / CUs) workgroups of its share; a 128 x 128
// workgroup does half the work of a 256 x 128 one at ~0.93 of its rate (4 resident workgroups per CU instead of 2),
// a 64 x 128 one a quarter at ~0.90, and a workgroup alone on its CU loses ~10 % (nothing covers its barriers).
// Measured (tools/gemm_shapes.py, TFLOP/s, 256x128 / 128x128 / 64x128): 8192^3 150 / 146 / -, 4096^3 147 / 146 / -,
// 3072^3 84 / 109 / -, 1536x8192x4096 111 / 140 / -, 2304^3 - / 91 / 117, 2048^3 67 / 128 / 134,
// 2048x2048x16384 69 / 134 / 140.  All fetch their k-tiles straight into LDS when the operands allow it.
static int sp_gemm_dp_choice(int64_t M, int64_t N, double* cost) {
  const int64_t tb = ((M + 255) / 256) * ((N + 127) / 128), ts = ((M + 127) / 128) * ((N + 127) / 128);
  const int64_t tt = ((M + 63) / 64) * ((N + 127) / 128);
  const int64_t cb = (tb + SP_CUS - 1) / SP_CUS, cs = (ts + SP_CUS - 1) / SP_CUS, ct = (tt + SP_CUS - 1) / SP_CUS;
  const double cost_b = 2.0 * (double)cb * (cb == 1 ? 1.14 : 1.0);
  const double cost_s = (double)cs / 0.93 * (cs == 1 ? 1.08 : 1.0);
  const double cost_t = 0.5 * (double)ct / 0.90 * (ct == 1 ? 1.08 : 1.0);
  int v = 6;
  *cost = cost_b;
  if (cost_s < *cost) v = 7, *cost = cost_s;
  if (cost_t < *cost) v = 8, *cost = cost_t;
  return v;
}

extern "C" int sp_gemm_f32(const float* d_A, int64_t lda, const float* d_B, int64_t ldb, float* d_C,
                           int64_t ldc, int64_t M, int64_t N, int64_t K, int32_t accumulate, void* stream) {
  if (M < 0 || N < 0 || K < 0) SP_FAIL("sp_gemm_f32: negative dimension");
  if (M == 0 || N == 0) return 0;
  if (!d_A || !d_B || !d_C) SP_FAIL("sp_gemm_f32: NULL pointer");
  if (M > 2147483647LL || N > 2147483647LL || K > 2147483647LL) SP_FAIL("sp_gemm_f32: dimension too large");
  if (lda < K || ldb < N || ldc < N) SP_FAIL("sp_gemm_f32: leading dimension too small");
  hipStream_t st = (hipStream_t)stream;
  if (K == 0) {
    if (!accumulate) {
      // empty contraction: C = 0 (numpy: zeros)
      SP_HIP(hipMemset2DAsync(d_C, (size_t)ldc * 4, 0, (size_t)N * 4, (size_t)M, st));
    }
    return 0;
  }
  const bool fast = (N % 4 == 0) && (N >= 4) && (lda % 4 == 0) && (ldb % 4 == 0) &&
                    ((((uintptr_t)d_A) | ((uintptr_t)d_B)) & 15) == 0;
  int v = sp_gemm_variant();
  if (v < 0) {
    double cost;
    v = sp_gemm_dp_choice(M, N, &cost);
  }
  switch (v) {
    case 0: return sp_gemm_launch<256, 128, 16, 2, 2>(d_A, lda, d_B, ldb, d_C, ldc, M, N, K, accumulate, fast, st);
    case 1: return sp_gemm_launch<128, 128, 16, 2, 2>(d_A, lda, d_B, ldb, d_C, ldc, M, N, K, accumulate, fast, st);
    case 2: return sp_gemm_launch<256, 256, 16, 2, 4>(d_A, lda, d_B, ldb, d_C, ldc, M, N, K, accumulate, fast, st);
    case 3: return sp_gemm_launch<128, 256, 16, 2, 2>(d_A, lda, d_B, ldb, d_C, ldc, M, N, K, accumulate, fast, st);
    // (SP_GEMM_VARIANT only) k-tiles of 32 halve the barriers per contraction: slower, profiles/r01_notes.md
    case 4: return sp_gemm_launch<128, 128, 32, 2, 2>(d_A, lda, d_B, ldb, d_C, ldc, M, N, K, accumulate, fast, st);
    case 5: return sp_gemm_launch<256, 128, 32, 2, 2>(d_A, lda, d_B, ldb, d_C, ldc, M, N, K, accumulate, fast, st);
    // direct-to-LDS k-tiles (preconditions checked here; otherwise the register-staged kernel of the same tile)
    case 6: case 9: case 11:
      if (fast && K >= 16 && (K % 16 == 0 || v == 6) && (int64_t)256 * lda < (1LL << 30) && (int64_t)16 * ldb < (1LL << 30)) {
        if (v == 6) return sp_gemm_glds_launch<256, 128, 2, 2, 0, 2>(d_A, lda, d_B, ldb, d_C, ldc, M, N, K, accumulate, st);
        if (v == 11) return sp_gemm_glds_launch<256, 128, 2, 2, 3, 2>(d_A, lda, d_B, ldb, d_C, ldc, M, N, K, accumulate, st);
        // (SP_GEMM_VARIANT only) register double-buffered fragments, barrier between the k-tile's halves:
        // 139.2 vs 141.5 TFLOP/s for case 6 -- profiles/r02_notes.md
        return sp_gemm_glds_launch<256, 128, 2, 2, 1, 2>(d_A, lda, d_B, ldb, d_C, ldc, M, N, K, accumulate, st);
      }
      return sp_gemm_launch<256, 128, 16, 2, 2>(d_A, lda, d_B, ldb, d_C, ldc, M, N, K, accumulate, fast, st);
#if SP_GEMM_ABLATE
#define SP_ABL_CASE(m) case 16 + m: return sp_gemm_glds_launch<256, 128, 2, 2, 16 + m, 2>(d_A, lda, d_B, ldb, d_C, ldc, M, N, K, accumulate, st);
    SP_ABL_CASE(0) SP_ABL_CASE(1) SP_ABL_CASE(2) SP_ABL_CASE(3) SP_ABL_CASE(4) SP_ABL_CASE(5) SP_ABL_CASE(6) SP_ABL_CASE(7)
    SP_ABL_CASE(8) SP_ABL_CASE(10)
#undef SP_ABL_CASE
#endif
    case 8:   // 64 x 128: twice the workgroups where 128 x 128 would leave one (or three) per CU
      if (fast && K >= 16 && (int64_t)128 * lda < (1LL << 30) && (int64_t)16 * ldb < (1LL << 30))
        return sp_gemm_glds_launch<64, 128, 2, 2, 0, 4>(d_A, lda, d_B, ldb, d_C, ldc, M, N, K, accumulate, st);
      return sp_gemm_launch<128, 128, 16, 2, 2>(d_A, lda, d_B, ldb, d_C, ldc, M, N, K, accumulate, fast, st);
    case 7:
      if (fast && K >= 16 && (int64_t)128 * lda < (1LL << 30) && (int64_t)16 * ldb < (1LL << 30))
        return sp_gemm_glds_launch<128, 128, 2, 2, 0, 4>(d_A, lda, d_B, ldb, d_C, ldc, M, N, K, accumulate, st);
      return sp_gemm_launch<128, 128, 16, 2, 2>(d_A, lda, d_B, ldb, d_C, ldc, M, N, K, accumulate, fast, st);
    default: SP_FAIL("sp_gemm_f32: unknown SP_GEMM_VARIANT=%d", v);
  }
}


// ---- split-K (small M x N, long contraction: x^T x of a tall matrix, ridge_regression.py:18-19) ----------
// With fewer than 256 output tiles most CUs would idle while a few walk the whole K; the contraction is
// cut into `splits` slices, each slice writes its own partial product, and the partials are added in slice
// order (deterministic, no atomics).
template <typename T>
__global__ __launch_bounds__(256) void sp_splitk_reduce_kernel(const T* __restrict__ part, int splits, int64_t mn, int N,
                                                               T* __restrict__ C, int64_t ldc, int accumulate) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < mn; e += stride) {
    const int64_t i = e / N, j = e - i * N;
    T* p = C + i * ldc + j;
    T acc = accumulate ? *p : (T)0;
    for (int s = 0; s < splits; ++s) acc += part[(int64_t)s * mn + e];
    *p = acc;
  }
}

struct SplitPlan {
  int splits, klen;
};

static SplitPlan sp_split_plan(int64_t M, int64_t N, int64_t K, int bk) {
  SplitPlan pl = {1, 0};
  const int64_t tiles = ((M + 127) / 128) * ((N + 127) / 128);
  if (tiles >= 256 || K < 2048) return pl;
  int64_t s = (512 + tiles - 1) / tiles;
  if (s > K / 512) s = K / 512;
  if (s > 512) s = 512;
  if (s < 2) return pl;
  int64_t klen = (K + s - 1) / s;
  klen = (klen + bk - 1) / bk * bk;
  s = (K + klen - 1) / klen;
  if (s < 2) return pl;
  pl.splits = (int)s;
  pl.klen = (int)klen;
  return pl;
}

int sp_dgemm_split_launch(const double* A, int64_t lda, const double* B, int64_t ldb, double* part, int64_t M,
                          int64_t N, int64_t K, int splits, int klen, hipStream_t st);

// When the balanced kernel (sp_gemm_glds_sk_kernel, 256 x 128 tiles on 512 workgroups) is expected to beat both
// data-parallel tilings, in the units of sp_gemm_f32's cost model (one 128 x 128 x K tile at the full CU rate):
// every CU gets tiles / 256 of the work (+3 % for the segment prologues) plus the fix-up pass, which moves about
// 1.5 partial images per workgroup out and back in.  SP_GEMM_SK=0 / 1 turns it off / forces it (where it applies).
static const size_t SP_SK_WS_BYTES = (size_t)2 * SP_CUS * 2 * 256 * 128 * 4;
// measured (tools/gemm_shapes.py, SP_GEMM_SK=0 / 1 over 2304^3 .. 10000^3): the whole rounds run at the data-parallel
// kernel's rate, the balanced remainder at ~0.93 of peak (ranges start at different k: less sharing in L2), and
// the last partial stores + the fix-up pass cost ~33 us
#define SP_SK_EFF_ROUNDS 0.95
#define SP_SK_EFF_REM 0.93
#define SP_SK_FIXED_S 33e-6
static bool sp_sk_plan(int64_t M, int64_t N, int64_t K) {
  static int mode = -2;
  if (mode == -2) {
    const char* e = getenv("SP_GEMM_SK");
    mode = e ? atoi(e) : -1;
  }
  if (mode == 0 || sp_gemm_variant() >= 0) return false;
  if (K < 256 || N % 4 != 0 || N < 4) return false;
  const int64_t tb = ((M + 255) / 256) * ((N + 127) / 128);
  if (tb >= 2147483647LL) return false;
  const int64_t W = 2 * SP_CUS, rem = tb % W;
  if (rem * (K / 16) < W * 8) return false;      // nothing (or next to nothing) left over to balance
  if (mode > 0) return true;
  double dp_cost;
  sp_gemm_dp_choice(M, N, &dp_cost);
  const double unit_s = 128.0 * 128.0 * (double)K * 2.0 / (157.3e12 * 0.95 / SP_CUS);
  const double dp_s = dp_cost * unit_s;
  const double tile_flop = 256.0 * 128.0 * (double)K * 2.0;
  const double sk_s = (double)(tb - rem) * tile_flop / (157.3e12 * SP_SK_EFF_ROUNDS) +
                      (double)rem * tile_flop / (157.3e12 * SP_SK_EFF_REM) + SP_SK_FIXED_S;
  return sk_s < 0.97 * dp_s;
}

extern "C" size_t sp_gemm_workspace_bytes(int32_t dtype, int64_t M, int64_t N, int64_t K) {
  if (M < 1 || N < 1 || (dtype != SP_F32 && dtype != SP_F64)) return 0;
  const SplitPlan pl = sp_split_plan(M, N, K, dtype == SP_F32 ? 16 : 8);
  if (pl.splits > 1) return (size_t)pl.splits * M * N * (dtype == SP_F32 ? 4 : 8) + 256;
  if (dtype == SP_F32 && sp_sk_plan(M, N, K)) return SP_SK_WS_BYTES + 256;
  return 0;
}

extern "C" int sp_gemm_f64(const double*, int64_t, const double*, int64_t, double*, int64_t, int64_t, int64_t, int64_t,
                           int32_t, void*);

extern "C" int sp_gemm_ws(int32_t dtype, const void* d_A, int64_t lda, const void* d_B, int64_t ldb, void* d_C,
                          int64_t ldc, int64_t M, int64_t N, int64_t K, int32_t accumulate, void* d_ws,
                          size_t ws_bytes, void* stream) {
  if (dtype != SP_F32 && dtype != SP_F64) SP_FAIL("sp_gemm_ws: dtype must be f32 or f64");
  const SplitPlan pl = (M > 0 && N > 0) ? sp_split_plan(M, N, K, dtype == SP_F32 ? 16 : 8) : SplitPlan{1, 0};
  const size_t need = sp_gemm_workspace_bytes(dtype, M, N, K);
  if (pl.splits <= 1 && need && d_ws && ws_bytes >= need && dtype == SP_F32 && d_A && d_B && d_C && lda >= K &&
      ldb >= N && ldc >= N && lda % 4 == 0 && ldb % 4 == 0 && ldc % 4 == 0 &&
      ((((uintptr_t)d_A) | ((uintptr_t)d_B) | ((uintptr_t)d_C)) & 15) == 0 && (int64_t)256 * lda < (1LL << 30) &&
      (int64_t)16 * ldb < (1LL << 30) && M <= 2147483647LL && N <= 2147483647LL) {
    float* part = (float*)(((uintptr_t)d_ws + 255) & ~(uintptr_t)255);
    return sp_gemm_sk_launch<256, 128, 2, 2, 2>((const float*)d_A, lda, (const float*)d_B, ldb, (float*)d_C, ldc, M, N, K,
                                                accumulate, part, (hipStream_t)stream);
  }
  if (pl.splits <= 1 || !d_ws || ws_bytes < need) {
    return dtype == SP_F32 ? sp_gemm_f32((const float*)d_A, lda, (const float*)d_B, ldb, (float*)d_C, ldc, M, N, K, accumulate, stream)
                           : sp_gemm_f64((const double*)d_A, lda, (const double*)d_B, ldb, (double*)d_C, ldc, M, N, K, accumulate, stream);
  }
  if (!d_A || !d_B || !d_C) SP_FAIL("sp_gemm_ws: NULL pointer");
  if (lda < K || ldb < N || ldc < N) SP_FAIL("sp_gemm_ws: leading dimension too small");
  hipStream_t st = (hipStream_t)stream;
  char* part = (char*)(((uintptr_t)d_ws + 255) & ~(uintptr_t)255);
  const int64_t mn = M * N;
  if (dtype == SP_F32) {
    const float* A = (const float*)d_A;
    const float* B = (const float*)d_B;
    // every slice but the last is a multiple of BK long; the general (guarded) loads cover the last one
    const bool fast = (N % 4 == 0) && (N >= 4) && (lda % 4 == 0) && (ldb % 4 == 0) &&
                      ((((uintptr_t)A) | ((uintptr_t)B)) & 15) == 0;
    if (sp_gemm_launch<128, 128, 16, 2, 2, true>(A, lda, B, ldb, (float*)part, N, M, N, K, 0, fast, st, pl.splits, pl.klen, mn))
      return 1;
    hipLaunchKernelGGL((sp_splitk_reduce_kernel<float>), dim3((unsigned)((mn + 255) / 256 > 4096 ? 4096 : (mn + 255) / 256)),
                       dim3(256), 0, st, (const float*)part, pl.splits, mn, (int)N, (float*)d_C, ldc, accumulate);
  } else {
    if (sp_dgemm_split_launch((const double*)d_A, lda, (const double*)d_B, ldb, (double*)part, M, N, K, pl.splits,
                              pl.klen, st))
      return 1;
    hipLaunchKernelGGL((sp_splitk_reduce_kernel<double>), dim3((unsigned)((mn + 255) / 256 > 4096 ? 4096 : (mn + 255) / 256)),
                       dim3(256), 0, st, (const double*)part, pl.splits, mn, (int)N, (double*)d_C, ldc, accumulate);
  }
  SP_CHECK_LAUNCH();
  return 0;
}

// Fused tier of sp_nearest_center: fp32 MFMA GEMM  score[c][i] = |c|^2/2 - c.x_i  with the per-point argmin folded
// into the epilogue, so the n x k distance matrix (40 GB at BASELINE configs[3]) is never materialised.  (|x_i|^2 is
// constant per point and irrelevant to the argmin.)
//
// Exactness: a point is final only if its two best scores differ by more than 4E,
// E = u((2D+4)|x||c|max + 2|c|max^2), u = 2^-24: the fp32 error bound of one score (centers rounded to fp32, fmaf
// chain of D terms, |c|^2 rounded once, one final rounding).  Otherwise its label is written as -1-best and the
// point is listed; a second MFMA pass over the listed points marks the centers whose score lies within E of the
// point's best, and only those get cdist's exact fp64 distance (kmeans.hip: sp_nearest_candidates_kernel) -- so the
// labels equal the exact tier's.
//
// History (profiles/r01_notes.md, r02_notes.md): round 1 had the points as MFMA rows (128 x 128 tile, (best, second,
// index) per accumulator register: 96 VGPRs of state) at 62 % of the MFMA peak with a VALU re-check, and a variant
// with the point block resident in LDS that was slower still; the kernel below reaches 77 %.
#pragma once
#include <type_traits>

typedef float km_f32x16 __attribute__((ext_vector_type(16)));
typedef float km_f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int KM_BK = 16;
constexpr int KM_BN_MAX = 256;   // centers are padded to a multiple of this (any column-block width divides it)

// Cf[c][j] = (float)C[c][j], row-major, zero padded to [kp][dp]; cn[c] = |C[c]|^2 / 2 (fp64 sum, rounded; +inf on
// padding);
// *cmax2 = max_c |C[c]|^2 (as float bits; non-negative floats order like unsigned ints).
// One wavefront per center, lanes along the features (a thread per center walking its row was a 256-step
// latency chain: 84 us for 1024 x 256, rocprofv3).
template <typename TC>
__global__ __launch_bounds__(256) void sp_centers_prep_kernel(const TC* __restrict__ C, int64_t ldc, int k, int d,
                                                              int kp, int dp, float* __restrict__ Cf,
                                                              float* __restrict__ cn, unsigned* __restrict__ cmax2) {
  const int lane = threadIdx.x & 63;
  const int c = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  if (c >= kp) return;
  if (c >= k) {
    if (lane == 0) cn[c] = INFINITY;
    return;
  }
  double s = 0.0;
  for (int j = lane; j < d; j += 64) {
    const double v = (double)C[(int64_t)c * ldc + j];
    s += v * v;
    Cf[(int64_t)c * dp + j] = (float)v;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if (lane == 0) {
    const float sf = (float)s;
    cn[c] = 0.5f * sf;   // the kernel compares halved scores |c|^2/2 - x.c (exact scaling)
    // round the max up so the bound stays a bound
    atomicMax(cmax2, __float_as_uint(sf * 1.0000002f));
  }
}

// ---- centers as MFMA ROWS, points as MFMA COLUMNS (round 2; the default) --------------------------
// score^T[c][i] = |c|^2/2 - c.x_i as an "NT" GEMM: both operands are row-major over the features
// ([center][k] and [point][k]), both are staged [row][k] (padded) in LDS and read as 16-B fragments.
// The MFMA's D layout puts ONE POINT per lane column (lane & 31) and 16 CENTERS per accumulator, so
// the running (best, second best, best's center) of a point is per LANE: 3 registers per point column
// instead of 3 per accumulator register (96 VGPRs in round 1's points-as-rows kernel).  That pays for the
// GEMM's own 256 x 128 macro-tile (wave tile 128 centers x 64 points, 128 accumulator registers, two
// workgroups per CU): 64 MFMAs per wave per barrier instead of 32, the point block is re-read once per
// 256 centers instead of once per 128, and the epilogue is 4 VALU ops per score, once per 256-center block.
// A score is an fmaf chain over the features from 0 followed by one subtraction from |c|^2/2 (the arithmetic the
// bound E above is derived for).
// The k-tiles go from global memory straight into LDS (global_load_lds_dwordx4, as in gemm.hip's
// sp_gemm_glds_kernel: no staging registers, no ds_write pass): both images are [row][16] unpadded, chunk q of row
// r in slot q ^ ((r >> 2) & 3) so that the 16-B fragment reads stay conflict-free, and the |c|^2/2 slice of a
// center block is one more 1 KiB piece.  The last k-tile of a feature count that is not a multiple of 16, and every
// k-tile of point rows that are not 16-B aligned (FAST = false), take the register path for the point tile only,
// into the same image.
constexpr int KN_BM = 256;                       // centers per block (KM_BN_MAX: kp is a multiple of it)
constexpr int KN_BN = 128;                       // points per workgroup
constexpr int KN_A_FLOATS = KN_BM * KM_BK;       // 4096: rows of 16 floats, UNPADDED (see the k-tile loads below)
constexpr int KN_B_FLOATS = KN_BN * KM_BK;       // 2048
constexpr int KN_STAGE = KN_A_FLOATS + KN_B_FLOATS;
constexpr int KN_SMEM_FLOATS = 2 * KN_STAGE + 2 * KN_BM;   // two stages + two |c|^2/2 slices  (51 200 B)
static_assert(KN_SMEM_FLOATS * 4 <= 65536, "static LDS limit");
static_assert(KN_BM == KM_BN_MAX, "kp must be a multiple of the center block");

// The verdict on one point from its fp32 (best, second best) halved scores: the label when the gap clears the error
// bound, else -1 - label and a place in the list the re-check works through.
// `ef`: the tier's error factor (this file: 2 D + 4, km_fp32_factor; kmeans_split.hpp: km_split_factor).
__device__ __forceinline__ float km_fp32_factor(int d) { return 2.0f * (float)d + 4.0f; }
__device__ __forceinline__ void km_decide(float b, float s, int ix, float xn, int point, float ef, float cmax,
                                          float cmax2, int64_t* __restrict__ labels, int* __restrict__ amb_rows,
                                          float* __restrict__ amb_best, int* __restrict__ amb_count) {
  const float u = 5.9604645e-8f;                     // 2^-24
  const float xnorm = sqrtf(xn) * 1.001f;            // fp32 sum of squares: generous slack
  const float E = u * (ef * xnorm * cmax + 2.0f * cmax2);
  // (magnitudes at which products, the bound itself or the bf16 halves of the split tier leave the normal range: no
  //  verdict from the filter, the exact stage decides)
  const float pn = xnorm * cmax;
  const bool underflows = pn > 0.f && pn < 1e-28f;
  const bool sure = !underflows && 2.0f * (s - b) > 4.0f * E;       // (scores are halved) false for NaN / inf-inf as well
  labels[point] = sure ? (int64_t)ix : (int64_t)(-1 - ix);
  if (!sure) {   // (order-free: each listed point is re-done on its own)
    const int pos = atomicAdd(amb_count, 1);
    amb_rows[pos] = point;
    amb_best[pos] = b;   // its best (halved) fp32 score: the re-check's candidate window starts here
  }
}

//
// RECHECK = true is the second pass over the points the first pass could not decide (amb_rows[0 .. *amb_count),
// gathered through the row list): the same contraction, one workgroup per (128 listed points, 256-center block
// blockIdx.y), whose epilogue marks every center whose score lies within the error window E of the point's best
// score -- the only centers that can be the exact answer (best is within E/2 of its true value, a candidate within
// E/2 of its own) -- as one bit of cand_mask[listed point][kp / 32].  Every word is written by exactly one lane: no
// atomics, no zeroing.  sp_nearest_candidates_kernel (kmeans.hip) then takes cdist's fp64 distance of the marked
// centers only.  (Round 1 re-computed all k scores of a listed point with VALU FMAs: 0.68 ms for 1.4 % of the
// points at configs[3], against 5.5 ms for the whole first pass.)
//
// PARTIAL = true serves the points of the last, partly filled round of workgroups (see sp_nearest_fused_launch):
// workgroup (x, y) walks only center blocks [y * per_tiles, (y + 1) * per_tiles) for its 128 points and leaves its
// (best, second best, best's center) per point in part[y][0..2][point] (+ |x|^2 from y = 0);
// sp_nearest_merge_parts_kernel combines the parts and decides exactly as the tail of this kernel does.
template <bool FAST, bool RECHECK, bool PARTIAL = false>
__global__ __launch_bounds__(256, 2) void sp_nearest_nt_kernel(const float* __restrict__ X, int64_t ldx,
                                                               const float* __restrict__ Cf,   // [kp][dp], zero padded
                                                               const float* __restrict__ chalf,
                                                               const unsigned* __restrict__ cmax2_bits, int n,
                                                               int d, int dp, int kp, int64_t* __restrict__ labels,
                                                               int* __restrict__ amb_rows,
                                                               float* __restrict__ amb_best,
                                                               int* __restrict__ amb_count,
                                                               unsigned* __restrict__ cand_mask,
                                                               float* __restrict__ part, int ldp, int per_tiles) {
  // ONE LDS object (stages, |c|^2/2 slices; the merge arrays alias stage 0 after the main loop)
  __shared__ __attribute__((aligned(16))) float smem[KN_SMEM_FLOATS];
  float* chs = smem + 2 * KN_STAGE;
  constexpr int THREADS = 256;
  constexpr int KQ = KM_BK / 4;
  constexpr int AV = (KN_BM * KQ) / THREADS, BV = (KN_BN * KQ) / THREADS;   // 4, 2 float4 per thread per k-step
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: the LDS targets of the k-tile loads are
  const int wm = wid >> 1, wn = wid & 1;          // wm: center half (128 rows), wn: point half (64 columns)
  const int l31 = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.x * KN_BN;              // first point (RECHECK: first list slot) of this workgroup
  int listed = 0;
  if constexpr (RECHECK) {
    listed = *amb_count;
    // the grid covers `n` = the capacity of the candidate masks; a longer list is left to the exact kernel
    if (listed > n || m0 >= listed) return;
  }

  // ---- k-tile pieces (1 KiB = one wave-wide 16-B load): slot -> (row, chunk) of the swizzled image
  constexpr int AP = (KN_A_FLOATS / 256) / 4, BP = (KN_B_FLOATS / 256) / 4;   // 4 and 2 pieces per wave
  // (loads are `global_load_lds_dwordx4 v_off, s[base]`: per-lane unsigned BYTE offsets off scalar bases, see
  // SP_GLDS_S; the gathered rows of the re-check take the 64-bit address form)
  unsigned a_off[AP];
#pragma unroll
  for (int j = 0; j < AP; ++j) {
    const int slot = (wid * AP + j) * 64 + lane, row = slot >> 2;
    a_off[j] = (unsigned)(row * dp + ((slot & 3) ^ ((row >> 2) & 3)) * 4) * 4u;
  }
  const float* __restrict__ Xblk = RECHECK ? X : X + (int64_t)m0 * ldx;
  // Point tile, two ways.  FAST (rows 16-B aligned): whole k-tiles go straight into LDS like the centers (b_off:
  // piece slots as for A); the last, partial k-tile of a feature count that is not a multiple of 16 -- and every
  // k-tile when FAST is off -- goes through registers with per-element guards (r_off / b_lds: thread e owns chunk
  // e % 4 of row e / 4), zero beyond d.
  typename std::conditional<RECHECK, int64_t, unsigned>::type b_off[BP];
  typename std::conditional<RECHECK, int64_t, int>::type r_off[BP];
  int b_lds[BP];                                  // (register path) where this thread's 16 B go
  auto row_offset = [&](int row) {                // floats from Xblk to the start of tile row `row`
    if constexpr (RECHECK) {
      const int listed_row = m0 + row < listed ? m0 + row : listed - 1;   // tail: repeat the last listed point
      return (int64_t)amb_rows[listed_row] * ldx;
    } else {
      if (m0 + row > n - 1) row = n - 1 - m0;     // clamp: results of points >= n are discarded
      return row * (int)ldx;
    }
  };
#pragma unroll
  for (int j = 0; j < BP; ++j) {
    const int rs = tid + j * THREADS, rrow = rs >> 2;
    b_lds[j] = rrow * KM_BK + ((rs & 3) ^ ((rrow >> 2) & 3)) * 4;
    r_off[j] = row_offset(rrow) + (rs & 3) * 4;
    const int fs = (wid * BP + j) * 64 + lane, frow = fs >> 2;
    const auto f = row_offset(frow) + ((fs & 3) ^ ((frow >> 2) & 3)) * 4;
    if constexpr (RECHECK) b_off[j] = f;          // floats, 64-bit (gathered rows)
    else b_off[j] = (unsigned)f * 4u;             // bytes off the scalar base
  }
  bool b_in_regs = false;                         // the k-tile being requested is a register one
  const unsigned sA_w = SP_LDS_ADDR(smem) + wid * (AP * 1024);      // this wave's pieces inside a stage (bytes)
  const unsigned sB_w = SP_LDS_ADDR(smem) + KN_A_FLOATS * 4 + wid * (BP * 1024);
  const unsigned chs_w = SP_LDS_ADDR(chs);
  const int nt = dp / KM_BK;
  const int tm_first = RECHECK ? (int)blockIdx.y : PARTIAL ? (int)blockIdx.y * per_tiles : 0;
  // center blocks walked by this workgroup
  const int tiles_m = RECHECK ? 1 : PARTIAL ? min(per_tiles, kp / KN_BM - tm_first) : kp / KN_BM;
  const int steps = nt * tiles_m;
  km_f32x4 rb[BP];

  // request k-step `step` (into stage step & 1); KN_STORE completes it on the register path.  The steps are
  // requested in order, so (center block, k-tile) of the next request are two counters (no division per k-step).
  int ld_tr = 0, ld_kt = 0;
#define KN_LOAD(step)                                                                    \
  do {                                                                                   \
    const int tr_ = ld_tr, kt_ = ld_kt;                                                  \
    if (++ld_kt == nt) {                                                                 \
      ld_kt = 0;                                                                         \
      ++ld_tr;                                                                           \
    }                                                                                    \
    const int tm_ = tm_first + tr_;                                                      \
    const int k0_ = kt_ * KM_BK;                                                         \
    const float* Ak_ = Cf + (int64_t)tm_ * KN_BM * dp + k0_;                             \
    const unsigned dA_ = sA_w + ((step) & 1) * (KN_STAGE * 4);                           \
    _Pragma("unroll") for (int j = 0; j < AP; ++j) SP_GLDS_S(Ak_, a_off[j], dA_ + j * 1024); \
    b_in_regs = !FAST || k0_ + KM_BK > d;                                                \
    if (!b_in_regs) {                                                                    \
      _Pragma("unroll") for (int j = 0; j < BP; ++j) {                                   \
        if constexpr (RECHECK) {                                                         \
          SP_GLDS_V(Xblk + k0_ + b_off[j], sB_w + ((step) & 1) * (KN_STAGE * 4) + j * 1024); \
        } else {                                                                         \
          SP_GLDS_S(Xblk + k0_, b_off[j], sB_w + ((step) & 1) * (KN_STAGE * 4) + j * 1024); \
        }                                                                                \
      }                                                                                  \
    } else {                                                                             \
      _Pragma("unroll") for (int j = 0; j < BP; ++j) {                                   \
        const int kk = k0_ + ((tid + j * THREADS) & 3) * 4;                              \
        const float* p = Xblk + k0_ + r_off[j];                                          \
        rb[j].x = kk + 0 < d ? p[0] : 0.f;                                               \
        rb[j].y = kk + 1 < d ? p[1] : 0.f;                                               \
        rb[j].z = kk + 2 < d ? p[2] : 0.f;                                               \
        rb[j].w = kk + 3 < d ? p[3] : 0.f;                                               \
      }                                                                                  \
    }                                                                                    \
    if (kt_ == 0 && wid == 0) SP_GLDS_S(chalf + tm_ * KN_BM, (unsigned)lane * 16u, chs_w + (tr_ & 1) * (KN_BM * 4)); \
  } while (0)
#define KN_STORE(step)                                                                   \
  do {                                                                                   \
    if (b_in_regs) {                                                                     \
      float* sB_ = smem + ((step) & 1) * KN_STAGE + KN_A_FLOATS;                         \
      _Pragma("unroll") for (int j = 0; j < BP; ++j) *(km_f32x4*)(sB_ + b_lds[j]) = rb[j]; \
    }                                                                                    \
    SP_GLDS_LANDED();                                                                    \
  } while (0)

  km_f32x16 acc[4][2];                            // (set by the first MFMA of each center block)
  // per point column j (points m0 + wn*64 + j*32 + l31), over the centers this lane has seen
  float best[2] = {INFINITY, INFINITY}, second[2] = {INFINITY, INFINITY};
  // best's center as (center block, position 16 i + 4 q + e inside this lane's 64 rows of the block): the position
  // is an inline constant of the select (a full center number would cost a v_mov per score)
  int bpos[2] = {0, 0}, bblk[2] = {0, 0};
  float xs[2] = {0.f, 0.f};                        // partial |x|^2 (this lane's k slots), once per pass

  KN_LOAD(0);
  KN_STORE(0);
  __syncthreads();
  // fragments: row (wave tile row + l31), chunk (lh + 2c) ^ ((row >> 2) & 3) -- the same xor for both operands
  const int sw = (l31 >> 2) & 3;
  int a_frag[KM_BK / 8], b_frag[KM_BK / 8];
#pragma unroll
  for (int c = 0; c < KM_BK / 8; ++c) {
    a_frag[c] = (wm * 128 + l31) * KM_BK + 4 * ((lh + 2 * c) ^ sw);
    b_frag[c] = (wn * 64 + l31) * KM_BK + 4 * ((lh + 2 * c) ^ sw);
  }

  // One k-step (k-tile t of center block tr).  FIRST: the block's first k-tile, whose first MFMA per accumulator
  // takes 0 as its C operand -- the accumulators are never zeroed by VALU moves.
  int t = 0;
  // |x|^2 is summed during the FIRST center block a workgroup walks only (it does not depend on the centers; round 3
  // summed it in every block and divided by their number): 28 VALU instructions per k-step less in the other blocks
  // -- and on this chip every VALU instruction issued beside the MFMAs of a SIMD costs MFMA issue time (PMC, round 4:
  // MFMA idle tracks the non-MFMA VALU count, 1.35 per MFMA here against 0.14 in the plain GEMM).
  // (a compile-time switch of the k-step: as a run-time condition the compiler keeps the FMAs and selects)
  const bool xs_first = !PARTIAL || blockIdx.y == 0;
  auto kstep = [&](auto first_of_block, auto with_xs) {
    constexpr bool FIRST = decltype(first_of_block)::value;
    constexpr bool XS = decltype(with_xs)::value;
    if (t + 1 < steps) KN_LOAD(t + 1);
    const float* sA = smem + (t & 1) * KN_STAGE;
    const float* sB = sA + KN_A_FLOATS;
#pragma unroll
    for (int c = 0; c < KM_BK / 8; ++c) {
      km_f32x4 af[4], bf[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = *(const km_f32x4*)(sA + a_frag[c] + i * 32 * KM_BK);
#pragma unroll
      for (int j = 0; j < 2; ++j) bf[j] = *(const km_f32x4*)(sB + b_frag[c] + j * 32 * KM_BK);
      if constexpr (XS) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int s = 0; s < 4; ++s) xs[j] = __builtin_fmaf(bf[j][s], bf[j][s], xs[j]);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            if (FIRST && c == 0 && s == 0) {
              const km_f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], zero, 0, 0, 0);
            } else {
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
            }
          }
    }
    if (t + 1 < steps) KN_STORE(t + 1);
    __syncthreads();
    ++t;
  };
  for (int tr = 0; tr < tiles_m; ++tr) {
    const int tm = tm_first + tr;
    if (tr == 0 && xs_first) {
      kstep(std::true_type(), std::true_type());
      for (int kt = 1; kt < nt; ++kt) kstep(std::false_type(), std::true_type());
    } else {
      kstep(std::true_type(), std::false_type());
      for (int kt = 1; kt < nt; ++kt) kstep(std::false_type(), std::false_type());
    }
    // ---- epilogue of center block tm.  Halved scores h = |c|^2/2 - x.c; the rows of a lane ascend with
    // (i, q, e), so `<` keeps the first minimum; invariant best <= second, and the new second best is the
    // median of (best, second, h).
    const float* chb = chs + (tr & 1) * KN_BM + wm * 128 + 4 * lh;
    if constexpr (RECHECK) {
      // the window of each of this lane's two listed points, then one mask word per (point, 32 centers)
      const float cmax2 = __uint_as_float(*cmax2_bits);
      const float cmax = sqrtf(cmax2) * 1.0000002f;
      float thr[2];
      int slot[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        slot[j] = m0 + wn * 64 + j * 32 + l31;
        const float xnorm = sqrtf(xs[j] + __shfl_xor(xs[j], 32)) * 1.001f;
        const float E = 5.9604645e-8f * ((2.0f * (float)d + 4.0f) * xnorm * cmax + 2.0f * cmax2);
        thr[j] = amb_best[slot[j] < listed ? slot[j] : listed - 1] + E * 1.001f + 1e-30f;
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
          for (int h = 0; h < 2; ++h) {
            // two adjacent rows at a time: |c|^2/2 and the accumulator are both in adjacent registers (one packed
            // subtraction, no operand shuffling)
            typedef float km_f32x2 __attribute__((ext_vector_type(2)));
            const km_f32x2 c2 = {ch4[2 * h], ch4[2 * h + 1]};
            const km_f32x2 a2 = {acc[i][j][4 * q + 2 * h], acc[i][j][4 * q + 2 * h + 1]};
            const km_f32x2 v2 = c2 - a2;
#pragma unroll
            for (int o = 0; o < 2; ++o) {
              const float v = v2[o];
              const bool better = v < best[j];
              second[j] = __builtin_amdgcn_fmed3f(best[j], second[j], v);
              bpos[j] = better ? 16 * i + 4 * q + 2 * h + o : bpos[j];
              best[j] = fminf(best[j], v);
            }
          }
      }
#pragma unroll
    for (int j = 0; j < 2; ++j) bblk[j] = best[j] < before[j] ? tm : bblk[j];
  }
#undef KN_LOAD
#undef KN_STORE

  // ---- merge: the two lane halves of a column (rows differ by 4), then the two center waves (LDS)
  float* mb_s = smem;                 // [128]   (the stages are dead: every wave passed the last barrier)
  float* ms_s = smem + KN_BN;
  int* mi_s = (int*)(smem + 2 * KN_BN);
  float bb[2], ss[2], xn[2];
  int ii[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float b = best[j], s = second[j];
    // center = block * 256 + wave half * 128 + i * 32 + q * 8 + lane half * 4 + e
    int ix = bblk[j] * KN_BM + wm * 128 + (bpos[j] >> 4) * 32 + ((bpos[j] >> 2) & 3) * 8 + (bpos[j] & 3) + 4 * lh;
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
    xn[j] = xs[j] + __shfl_xor(xs[j], 32);
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
    const float cmax2 = __uint_as_float(*cmax2_bits);
    const float cmax = sqrtf(cmax2) * 1.0000002f;
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
          if (blockIdx.y == 0) part[(int64_t)gridDim.y * 3 * ldp + m0 + col] = xn[j];
        } else {
          km_decide(b, s, ix, xn[j], m0 + col, km_fp32_factor(d), cmax, cmax2, labels, amb_rows, amb_best, amb_count);
        }
      }
    }
  }
}

// Combines the per-center-range parts of the PARTIAL launch for tail point t (global point first + t).
__global__ __launch_bounds__(256) void sp_nearest_merge_parts_kernel(const float* __restrict__ part, int S, int ldp,
                                                                     int n_tail, int first, float ef,
                                                                     const unsigned* __restrict__ cmax2_bits,
                                                                     int64_t* __restrict__ labels,
                                                                     int* __restrict__ amb_rows,
                                                                     float* __restrict__ amb_best,
                                                                     int* __restrict__ amb_count) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n_tail) return;
  float b = part[t], s = part[ldp + t];
  int ix = ((const int*)part)[2 * (int64_t)ldp + t];
  for (int y = 1; y < S; ++y) {
    const int64_t at = (int64_t)y * 3 * ldp + t;
    const float ob = part[at], os = part[at + ldp];
    const int oi = ((const int*)part)[at + 2 * (int64_t)ldp];
    if (ob < b || (ob == b && oi < ix)) {
      s = fminf(b, os);
      b = ob;
      ix = oi;
    } else {
      s = fminf(ob, s);
    }
  }
  const float cmax2 = __uint_as_float(*cmax2_bits);
  km_decide(b, s, ix, part[(int64_t)S * 3 * ldp + t], first + t, ef, sqrtf(cmax2) * 1.0000002f, cmax2, labels, amb_rows,
            amb_best, amb_count);
}

static inline int64_t km_round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

// scratch layout of sp_nearest_center (all 256-B aligned)
struct KmWorkspace {
  int64_t kp, dp, n_points;
  double* Ct64;      // [d][kp]   fp64 transposed centers (exact kernel)
  float* Cf;         // [kp][dp]  fp32 row-major centers (sp_nearest_nt_kernel)
  float* cn;         // [kp]      |c|^2
  unsigned* cmax2;   // [1]       max |c|^2 (float bits)
  int* amb_count;    // [1]
  int* amb_rows;     // [n]       points the fused kernel could not decide
  float* amb_best;   // [n]       their best (halved) fp32 score
  int64_t cand_cap;  //           listed points the candidate masks have room for
  unsigned* cand_mask;   // [cand_cap][kp / 32]  centers inside the error window of a listed point
  float* part;       // [KM_TAIL_SPLIT][3][KM_TAIL_POINTS] + [KM_TAIL_POINTS]   parts of the tail points
  // the split tier (kmeans_split.hpp): hi / mid bf16 images of the centers and of the points, |x|^2 per point
  __bf16 *Ch, *Cm;   // [kp][dp] each
  float* colsum;     // [KM_MEAN_BLOCKS][dp]  partial column sums behind `mu`
  float* mu;         // [dp]      the shift: both operands are taken relative to it (kmeans_split.hpp)
  float* xn2;        // [n]       |x - mu|^2
  __bf16 *Xh, *Xm;   // [n][dp] each -- LAST: a caller that brings prepared points leaves them (mu, xn2 too) out
};
constexpr int KM_MEAN_BLOCKS = 2048;
constexpr int64_t KM_MEAN_SAMPLE = 65536;     // rows behind the shift of a prepared buffer (sp_kmeans_points_prepare)

// The last round of first-pass workgroups is rarely full (configs[3]: 9 766 workgroups over 512 slots = 19 rounds and
// 38 workgroups that cost a 20th).  The points of that round are split over up to KM_TAIL_SPLIT ranges of center
// blocks instead (PARTIAL launch + merge), so the round is ~1/split as long.
constexpr int KM_WG_SLOTS = 512;                        // 256 CUs x 2 workgroups (launch bounds of the kernel)
constexpr int KM_TAIL_SPLIT = 8;
constexpr int KM_TAIL_POINTS = KM_WG_SLOTS * KN_BN;

static inline size_t km_align(size_t v) { return (v + 255) & ~(size_t)255; }

// features padded to whole k-steps, and to at least TWO of them: the |c|^2/2 slice of center block tm is stored
// during the last k-step of block tm - 1, which must lie behind a barrier that follows the epilogue of block tm - 2
static inline int64_t km_padded_features(int64_t d) {
  return km_round_up(d < 1 ? 1 : d, 2 * KM_BK);      // (whole k-steps of either tier: 16, or 32 for the split tier's wide one)
}

// The MFMA re-check has room for n / 8 listed points (1.4 % are listed at configs[3]); a longer list -- degenerate
// data such as many coincident points -- goes to the exact kernel instead (decided on the device, no host sync).
static inline int64_t km_cand_cap(int64_t n) {
  int64_t cap = n / 8 < 4096 ? 4096 : n / 8;
  return cap > n ? (n < 1 ? 1 : n) : cap;
}

// bytes of the hi / mid images and the norms of n points (what sp_kmeans_points_prepare fills)
static size_t km_points_split_bytes(int64_t n, int64_t d) {
  const int64_t dp = km_padded_features(d), rows = n < 1 ? 1 : n;
  return km_align((size_t)KM_MEAN_BLOCKS * dp * 4) + km_align((size_t)dp * 4) + km_align((size_t)rows * 4) +
         2 * km_align((size_t)rows * dp * 2);
}

static size_t sp_nearest_fused_ws_bytes(int64_t n, int64_t k, int64_t d, bool with_points = true) {
  const int64_t kp = km_round_up(k < 1 ? 1 : k, KM_BN_MAX), dp = km_padded_features(d);
  return 256 + km_align((size_t)(d < 1 ? 1 : d) * kp * 8) + km_align((size_t)dp * kp * 4) + km_align((size_t)kp * 4) + 256 + 256 +
         2 * km_align((size_t)(n < 1 ? 1 : n) * 4) + km_align((size_t)km_cand_cap(n) * (kp / 32) * 4) +
         km_align((size_t)KM_TAIL_POINTS * (3 * KM_TAIL_SPLIT + 1) * 4) + 2 * km_align((size_t)dp * kp * 2) +
         (with_points ? km_points_split_bytes(n, d) : 0);
}

// the points' part (norms, hi image, mid image) of a workspace or of a prepared buffer
static void km_carve_points(void* at, int64_t n, int64_t d, KmWorkspace* w) {
  const int64_t dp = km_padded_features(d), rows = n < 1 ? 1 : n;
  char* p = (char*)at;
  w->colsum = (float*)p;
  p += km_align((size_t)KM_MEAN_BLOCKS * dp * 4);
  w->mu = (float*)p;
  p += km_align((size_t)dp * 4);
  w->xn2 = (float*)p;
  p += km_align((size_t)rows * 4);
  w->Xh = (__bf16*)p;
  p += km_align((size_t)rows * dp * 2);
  w->Xm = (__bf16*)p;
}

static KmWorkspace km_carve(void* ws, int64_t n, int64_t k, int64_t d) {
  KmWorkspace w;
  w.n_points = n;
  w.kp = km_round_up(k, KM_BN_MAX);
  w.dp = km_padded_features(d);
  char* p = (char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  w.Ct64 = (double*)p;
  p += km_align((size_t)(d < 1 ? 1 : d) * w.kp * 8);
  w.Cf = (float*)p;
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
  p += km_align((size_t)(n < 1 ? 1 : n) * 4);
  w.cand_cap = km_cand_cap(n);
  w.cand_mask = (unsigned*)p;
  p += km_align((size_t)w.cand_cap * (w.kp / 32) * 4);
  w.part = (float*)p;
  p += km_align((size_t)KM_TAIL_POINTS * (3 * KM_TAIL_SPLIT + 1) * 4);
  w.Ch = (__bf16*)p;
  p += km_align((size_t)w.dp * w.kp * 2);
  w.Cm = (__bf16*)p;
  p += km_align((size_t)w.dp * w.kp * 2);
  km_carve_points(p, n, d, &w);
  return w;
}

// the fused tier pays once the contraction is big enough to hide its fixed costs
static bool sp_nearest_fused_applicable(int64_t n, int64_t k, int64_t d, int tier) {
  if (n > 2147483647LL - KN_BN || d < 1 || k > (1LL << 20)) return false;
  if (tier == SP_NEAREST_FUSED || tier == SP_NEAREST_FUSED_UNCHECKED || tier == SP_NEAREST_SPLIT || tier == SP_NEAREST_SPLIT_UNCHECKED) return true;
  return n >= 1024 && k >= 16 && d >= 8 && n * k * d >= (1LL << 24);
}

static int sp_nearest_fused_launch(const float* X, int64_t ldx, const void* C, int32_t cdtype, int64_t ldc,
                                   int64_t n, int64_t k, int64_t d, int64_t* labels, const KmWorkspace& w,
                                   hipStream_t st) {
  if (ldx > (1LL << 30) / KN_BN) SP_FAIL("sp_nearest_center: leading dimension too large for the fused tier");
  const int64_t kp = w.kp, dp = w.dp;
  SP_HIP(hipMemsetAsync(w.Cf, 0, km_align((size_t)dp * kp * 4), st));
  SP_HIP(hipMemsetAsync(w.cmax2, 0, 512, st));   // cmax2 and amb_count
  const unsigned pblocks = (unsigned)((kp + 3) / 4);   // one wavefront per center
  if (cdtype == SP_F32)
    hipLaunchKernelGGL((sp_centers_prep_kernel<float>), dim3(pblocks), dim3(256), 0, st, (const float*)C, ldc, (int)k,
                       (int)d, (int)kp, (int)dp, w.Cf, w.cn, w.cmax2);
  else
    hipLaunchKernelGGL((sp_centers_prep_kernel<double>), dim3(pblocks), dim3(256), 0, st, (const double*)C, ldc, (int)k,
                       (int)d, (int)kp, (int)dp, w.Cf, w.cn, w.cmax2);
  SP_CHECK_LAUNCH();
  // direct-to-LDS loads of the points need 16-B aligned rows (a partial last k-tile goes through registers)
  const bool fast = (ldx % 4 == 0) && ((((uintptr_t)X) & 15) == 0);
  const int64_t blocks = (n + KN_BN - 1) / KN_BN, tiles = kp / KN_BM;
  // whole rounds walk all centers per workgroup; the workgroups of the partial last round are split over the centers
  // when that shortens it: `rem` workgroups of `tiles` blocks become rem * split of `per` blocks
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
  const int64_t whole = blocks - rem, n_whole = whole * KN_BN < n ? whole * KN_BN : n;
#define KN_FIRST_PASS(FAST_)                                                                                          \
  do {                                                                                                                \
    if (whole > 0)                                                                                                    \
      hipLaunchKernelGGL((sp_nearest_nt_kernel<FAST_, false, false>), dim3((unsigned)whole), dim3(256), 0, st, X, ldx, \
                         w.Cf, w.cn, w.cmax2, (int)n_whole, (int)d, (int)dp, (int)kp, labels, w.amb_rows, w.amb_best, \
                         w.amb_count, (unsigned*)nullptr, (float*)nullptr, 0, 0);                                      \
    if (rem > 0)                                                                                                      \
      hipLaunchKernelGGL((sp_nearest_nt_kernel<FAST_, false, true>), dim3((unsigned)rem, (unsigned)split), dim3(256), 0, \
                         st, X + n_whole * ldx, ldx, w.Cf, w.cn, w.cmax2, (int)(n - n_whole), (int)d, (int)dp, (int)kp, \
                         (int64_t*)nullptr, (int*)nullptr, (float*)nullptr, (int*)nullptr, (unsigned*)nullptr, w.part, \
                         KM_TAIL_POINTS, (int)per);                                                                    \
  } while (0)
  if (fast)
    KN_FIRST_PASS(true);
  else
    KN_FIRST_PASS(false);
#undef KN_FIRST_PASS
  SP_CHECK_LAUNCH();
  if (rem > 0) {
    const int n_tail = (int)(n - n_whole);
    hipLaunchKernelGGL(sp_nearest_merge_parts_kernel, dim3((unsigned)((n_tail + 255) / 256)), dim3(256), 0, st, w.part,
                       (int)split, KM_TAIL_POINTS, n_tail, (int)n_whole, 2.0f * (float)d + 4.0f, w.cmax2, labels, w.amb_rows, w.amb_best,
                       w.amb_count);
    SP_CHECK_LAUNCH();
  }
  return 0;
}

// Second pass over the listed points: marks the centers inside each point's error window (see the kernel).
static int sp_nearest_mark_candidates(const float* X, int64_t ldx, int64_t d, const KmWorkspace& w, hipStream_t st) {
  const bool fast = (ldx % 4 == 0) && ((((uintptr_t)X) & 15) == 0);
  const dim3 grid((unsigned)((w.cand_cap + KN_BN - 1) / KN_BN), (unsigned)(w.kp / KN_BM));
  if (fast)
    hipLaunchKernelGGL((sp_nearest_nt_kernel<true, true>), grid, dim3(256), 0, st, X, ldx, w.Cf, w.cn, w.cmax2,
                       (int)w.cand_cap, (int)d, (int)w.dp, (int)w.kp, (int64_t*)nullptr, w.amb_rows, w.amb_best,
                       w.amb_count, w.cand_mask, (float*)nullptr, 0, 0);
  else
    hipLaunchKernelGGL((sp_nearest_nt_kernel<false, true>), grid, dim3(256), 0, st, X, ldx, w.Cf, w.cn, w.cmax2,
                       (int)w.cand_cap, (int)d, (int)w.dp, (int)w.kp, (int64_t*)nullptr, w.amb_rows, w.amb_best,
                       w.amb_count, w.cand_mask, (float*)nullptr, 0, 0);
  SP_CHECK_LAUNCH();
  return 0;
}

}  // namespace

// One pass over a row tile X [n][d] for  g[c] = sum_i X[i][c] * (X[i,:] . w - y[i])  -- the gradient of the
// least-squares workload (BASELINE configs[4]; reference: spartan/examples/linear_regression.py:10-24 through
// tests/benchmark_lreg.py), which the expression API states as
//     residual = dot(x, w) - y ;  g = sum(x * residual, axis=0)
// i.e. a matrix.vector product (one launch, X read once) followed by a fused map -> column reduce (another launch,
// X read again).  Both are HBM-bound on X, so reading it once halves the step: a row fits a wavefront's registers
// (d <= 4096: 64 values per lane), the wave keeps it there between its two uses.
//
//   * a wave owns rows w, w + W, ...; lane l holds columns 256 j + 4 l + (0..3), j < d / 256 (16-byte loads, a row
//     is d / 256 wave-wide loads of 1 KiB, all in flight together);
//   * t = x . w: per-lane products added in column order, then a butterfly over the lanes; r = t - y[i];
//   * g_lane[c] += x[c] * r for the lane's columns (product rounded, then added: -ffp-contract=off);
//   * the four waves of a workgroup add their partial g in LDS (wave order), the workgroup writes ONE partial to the
//     workspace, and a second small kernel adds the partials in workgroup order.
// No atomics: the result does not depend on scheduling.  Summation order differs from the two-launch form (its row
// sums and column sums have their own trees), so results agree to rounding, not bit for bit; the expression
// rewrite that selects this kernel (spartan_amd/expr/optimize.py: RowDotColSumFusion) is applied on the HIP
// backend only and can be turned off (FLAGS['opt_rowdot_fusion']).
#include "sp_common.hpp"

namespace {

constexpr int RD_MAX_D = 4096;
constexpr int RD_J = RD_MAX_D / 256;          // 16-byte pieces of a row per lane
constexpr int RD_WAVES = SP_CUS * 8;          // 192 VGPRs: two waves per SIMD, all resident

typedef float rd_f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 2) void sp_rowdot_colsum_kernel(const float* __restrict__ X, int64_t ldx, int64_t n, int d,
                                                                  const float* __restrict__ w, const float* __restrict__ y,
                                                                  int64_t ldy, float* __restrict__ part) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nj = d >> 8;                       // whole 256-column pieces; the rest (a multiple of 4 columns) is piece nj
  const int tail = d & 255;
  const bool tail_lane = 4 * lane < tail;
  rd_f4 wv[RD_J], g[RD_J];
#pragma unroll
  for (int j = 0; j < RD_J; ++j) {
    const bool live = j < nj || (j == nj && tail_lane);
    wv[j] = live ? *(const rd_f4*)(w + 256 * j + 4 * lane) : rd_f4{0.f, 0.f, 0.f, 0.f};
    g[j] = rd_f4{0.f, 0.f, 0.f, 0.f};
  }
  for (int64_t i = wave; i < n; i += (int64_t)gridDim.x * 4) {
    const float* __restrict__ row = X + i * ldx + 4 * lane;
    rd_f4 x[RD_J];
#pragma unroll
    for (int j = 0; j < RD_J; ++j) {
      const bool live = j < nj || (j == nj && tail_lane);
      // (every row is read exactly once by the whole launch: streaming loads)
      x[j] = live ? __builtin_nontemporal_load((const rd_f4*)(row + 256 * j)) : rd_f4{0.f, 0.f, 0.f, 0.f};
    }
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < RD_J; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) t += x[j][e] * wv[j][e];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
    const float r = y ? t - y[i * ldy] : t;
#pragma unroll
    for (int j = 0; j < RD_J; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) g[j][e] += x[j][e] * r;
  }
  // the four waves of the workgroup add up in LDS, in wave order, and ONE partial per workgroup goes to the workspace
  // (round 3 wrote one per wave: 32 MB written and read again by the finishing kernel at configs[4], 3 % of the bytes
  // of the pass itself)
  __shared__ __attribute__((aligned(16))) float wsum[3][RD_MAX_D];
  const int wv_in_wg = threadIdx.x >> 6;
  if (wv_in_wg > 0) {
#pragma unroll
    for (int j = 0; j < RD_J; ++j)
      if (j < nj || (j == nj && tail_lane)) *(rd_f4*)(&wsum[wv_in_wg - 1][256 * j + 4 * lane]) = g[j];
  }
  __syncthreads();
  if (wv_in_wg == 0) {
    float* __restrict__ out = part + (int64_t)blockIdx.x * d + 4 * lane;
#pragma unroll
    for (int j = 0; j < RD_J; ++j)
      if (j < nj || (j == nj && tail_lane)) {
        rd_f4 t = g[j];
#pragma unroll
        for (int k = 0; k < 3; ++k) t += *(const rd_f4*)(&wsum[k][256 * j + 4 * lane]);
        *(rd_f4*)(out + 256 * j) = t;
      }
  }
}

// out[c] (+)= part[0][c] + part[1][c] + ... in wave order.  Workgroup: 64 columns x 16 groups of waves.
__global__ __launch_bounds__(1024) void sp_rowdot_finish_kernel(const float* __restrict__ part, int nwaves, int d,
                                                                float* __restrict__ out, int accumulate) {
  __shared__ float red[16][64];
  const int cl = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int per = (nwaves + 15) / 16;
  const int w0 = grp * per, w1 = w0 + per < nwaves ? w0 + per : nwaves;
  float s = 0.f;
  if (c < d) {
    int wv = w0;
    for (; wv + 8 <= w1; wv += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(int64_t)(wv + u) * d + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; wv < w1; ++wv) s += part[(int64_t)wv * d + c];
  }
  red[grp][cl] = s;
  __syncthreads();
  if (grp == 0 && c < d) {
    float t = red[0][cl];
    for (int k = 1; k < 16; ++k) t += red[k][cl];
    out[c] = accumulate ? out[c] + t : t;
  }
}

int rd_waves(int64_t n) {
  int64_t wv = (n + 3) / 4 * 4;
  if (wv > RD_WAVES) wv = RD_WAVES;
  return (int)wv;
}

}  // namespace

extern "C" size_t sp_rowdot_colsum_workspace_bytes(int64_t n, int64_t d) {
  if (n < 1 || d < 4 || d > RD_MAX_D || d % 4) return 0;
  return (size_t)rd_waves(n) * (size_t)d * 4 + 256;
}

extern "C" int sp_rowdot_colsum_f32(const float* d_x, int64_t ldx, int64_t n, int64_t d, const float* d_w, const float* d_y,
                                    int64_t ldy, float* d_out, int32_t accumulate, void* d_ws, size_t ws_bytes,
                                    void* stream) {
  if (n < 0 || d < 4 || d > RD_MAX_D || d % 4) SP_FAIL("sp_rowdot_colsum_f32: needs 4 <= d <= %d, d %% 4 == 0 (got %lld)", RD_MAX_D, (long long)d);
  if (!d_out || !d_w || (n && !d_x)) SP_FAIL("sp_rowdot_colsum_f32: NULL pointer");
  if (ldx < d || ldx % 4 || (((uintptr_t)d_x | (uintptr_t)d_w | (uintptr_t)d_out) & 15)) SP_FAIL("sp_rowdot_colsum_f32: X rows, w and out must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) {
    if (!accumulate) SP_HIP(hipMemsetAsync(d_out, 0, (size_t)d * 4, st));
    return 0;
  }
  const size_t need = sp_rowdot_colsum_workspace_bytes(n, d);
  if (!d_ws || ws_bytes < need) SP_FAIL("sp_rowdot_colsum_f32: workspace too small (%zu < %zu)", ws_bytes, need);
  float* part = (float*)(((uintptr_t)d_ws + 255) & ~(uintptr_t)255);
  const int waves = rd_waves(n);
  hipLaunchKernelGGL(sp_rowdot_colsum_kernel, dim3(waves / 4), dim3(256), 0, st, d_x, ldx, n, (int)d, d_w, d_y, ldy < 1 ? 1 : ldy, part);
  SP_CHECK_LAUNCH();
  hipLaunchKernelGGL(sp_rowdot_finish_kernel, dim3((unsigned)((d + 63) / 64)), dim3(1024), 0, st, (const float*)part, waves / 4,
                     (int)d, d_out, (int)accumulate);
  SP_CHECK_LAUNCH();
  return 0;
}

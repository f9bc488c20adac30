// Measured dense matrix-pipe rates of this GPU: every CU runs W waves per SIMD of back-to-back independent MFMAs
// (4 accumulator tiles per wave, nothing else in the loop).  Prints TFLOP/s for v_mfma_f64_16x16x4_f64 (the fp64
// GEMM's instruction; MI355X_MICROARCH.md has no fp64 matrix figure) next to v_mfma_f32_32x32x2_f32 (155 TF measured
// in that guide: calibrates the method) and v_mfma_f32_32x32x16_bf16 (the k-means first pass's instruction; the guide's
// dense bf16 figure is 2.5 PFLOP/s).  hipcc --offload-arch=gfx950 -O3 tools/mfma_peak_probe.hip -o tools/mfma_peak_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k_f64(double* out, int iters) {
  f64x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
  }
  double s = 0;
  for (int t = 0; t < 4; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  if (s == 12345.678) out[0] = s;
}
__global__ __launch_bounds__(256) void k_f32(float* out, int iters) {
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t)
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
  float a = 1.0f + threadIdx.x * 1e-6f, b = 1.0f - threadIdx.x * 1e-6f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
  }
  float s = 0;
  for (int t = 0; t < 4; ++t)
    for (int e = 0; e < 16; ++e) s += acc[t][e];
  if (s == 12345.678f) out[0] = s;
}
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void k_bf16(float* out, int iters) {
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t)
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = (__bf16)(1.0f + (threadIdx.x + e) * 0.0078125f);
    b[e] = (__bf16)(1.0f - (threadIdx.x + e) * 0.00390625f);
  }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
  }
  float s = 0;
  for (int t = 0; t < 4; ++t)
    for (int e = 0; e < 16; ++e) s += acc[t][e];
  if (s == 12345.678f) out[0] = s;
}
int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  void* out;
  hipMalloc(&out, 64);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 20000;
  for (int wpsimd = 1; wpsimd <= 2; ++wpsimd) {
    const dim3 grid(cus * wpsimd), block(256);
    for (int which = 0; which < 3; ++which) {
      double best = 0;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        if (which == 0) hipLaunchKernelGGL(k_f64, grid, block, 0, 0, (double*)out, iters);
        else if (which == 1) hipLaunchKernelGGL(k_f32, grid, block, 0, 0, (float*)out, iters);
        else hipLaunchKernelGGL(k_bf16, grid, block, 0, 0, (float*)out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop_per_mfma = which == 0 ? 2.0 * 16 * 16 * 4 : (which == 1 ? 2.0 * 32 * 32 * 2 : 2.0 * 32 * 32 * 16);
        const double flops = (double)grid.x * 4 /*waves*/ * iters * 16.0 * flop_per_mfma;
        const double tf = flops / (ms * 1e-3) / 1e12;
        if (tf > best) best = tf;
      }
      printf("PROBE %s waves_per_simd=%d cus=%d TFLOPs=%.2f\n", which == 0 ? "v_mfma_f64_16x16x4_f64" : (which == 1 ? "v_mfma_f32_32x32x2_f32" : "v_mfma_f32_32x32x16_bf16"), wpsimd, cus, best);
    }
  }
  return 0;
}

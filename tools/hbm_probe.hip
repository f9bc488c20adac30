// HBM streaming probe: which copy / read-reduce structure reaches the box's
// achievable bandwidth.  Build: hipcc --offload-arch=gfx950 -O3 tools/hbm_probe.hip -o /tmp/hbm_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_k(f4* __restrict__ d, const f4* __restrict__ s, long n) {
  const long stride = (long)gridDim.x * 256 * U;
  for (long i = (long)blockIdx.x * 256 * U + threadIdx.x; i < n; i += stride) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      long j = i + (long)u * 256;
      if (j < n) v[u] = NT ? __builtin_nontemporal_load(s + j) : s[j];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      long j = i + (long)u * 256;
      if (j < n) { if (NT) __builtin_nontemporal_store(v[u], d + j); else d[j] = v[u]; }
    }
  }
}

template <int U, bool NT>
__global__ __launch_bounds__(256) void sum_k(float* __restrict__ out, const f4* __restrict__ s, long n) {
  const long stride = (long)gridDim.x * 256 * U;
  f4 acc[U];
#pragma unroll
  for (int u = 0; u < U; ++u) acc[u] = (f4){0, 0, 0, 0};
  for (long i = (long)blockIdx.x * 256 * U + threadIdx.x; i < n; i += stride) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      long j = i + (long)u * 256;
      if (j < n) acc[u] += NT ? __builtin_nontemporal_load(s + j) : s[j];
    }
  }
  f4 a = acc[0];
#pragma unroll
  for (int u = 1; u < U; ++u) a += acc[u];
  float r = a.x + a.y + a.z + a.w;
  for (int d = 32; d; d >>= 1) r += __shfl_down(r, d, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, r);
}

// contiguous chunk per block, partial per block (no atomics)
template <int U, int T>
__global__ __launch_bounds__(T) void sumc_k(float* __restrict__ part, const f4* __restrict__ s, long n, long chunk) {
  const long b0 = (long)blockIdx.x * chunk;
  long b1 = b0 + chunk; if (b1 > n) b1 = n;
  f4 acc[U];
#pragma unroll
  for (int u = 0; u < U; ++u) acc[u] = (f4){0, 0, 0, 0};
  for (long i = b0 + threadIdx.x; i < b1; i += (long)T * U) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      long j = i + (long)u * T;
      if (j < b1) acc[u] += s[j];
    }
  }
  f4 a = acc[0];
#pragma unroll
  for (int u = 1; u < U; ++u) a += acc[u];
  float r = a.x + a.y + a.z + a.w;
  for (int d = 32; d; d >>= 1) r += __shfl_down(r, d, 64);
  __shared__ float sm[T / 64];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = r;
  __syncthreads();
  if (threadIdx.x == 0) { float t = 0; for (int k = 0; k < T / 64; ++k) t += sm[k]; part[blockIdx.x] = t; }
}

// column sums of an [R][C] fp32 matrix, lanes along C (16 B per lane), 4 waves interleave rows
template <int U>
__global__ __launch_bounds__(256) void colsum_k(float* __restrict__ part, const f4* __restrict__ s, long R, long C4, long rchunk) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long c = (long)blockIdx.x * 64 + lane;
  const long r0 = (long)blockIdx.y * rchunk;
  long r1 = r0 + rchunk; if (r1 > R) r1 = R;
  f4 acc[U];
#pragma unroll
  for (int u = 0; u < U; ++u) acc[u] = (f4){0, 0, 0, 0};
  if (c < C4) {
    for (long r = r0 + w; r < r1; r += 4L * U) {
#pragma unroll
      for (int u = 0; u < U; ++u) { long rr = r + 4L * u; if (rr < r1) acc[u] += s[rr * C4 + c]; }
    }
  }
  f4 a = acc[0];
#pragma unroll
  for (int u = 1; u < U; ++u) a += acc[u];
  __shared__ f4 sm[3][64];
  if (w) sm[w - 1][lane] = a;
  __syncthreads();
  if (w == 0 && c < C4) { a += sm[0][lane] + sm[1][lane] + sm[2][lane]; ((f4*)part)[(long)blockIdx.y * C4 + c] = a; }
}

template <typename F>
float timeit(F f, int iters = 10) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 2; ++i) f();
  hipEventRecord(a);
  for (int i = 0; i < iters; ++i) f();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / iters;
}

int main() {
  const long bytes = 2L << 30;
  const long n = bytes / 16;
  f4 *s, *d; float* out;
  CK(hipMalloc(&s, bytes)); CK(hipMalloc(&d, bytes)); CK(hipMalloc(&out, 4));
  CK(hipMemset(s, 1, bytes)); CK(hipMemset(d, 0, bytes));
  int grids[] = {1024, 2048, 4096, 8192, 16384, 0};
  printf("%-28s %8s %10s\n", "variant", "grid", "GB/s");
#define RUN_COPY(U, NT)                                                                   \
  for (int g : grids) {                                                                   \
    long full = (n + 256L * U - 1) / (256L * U);                                          \
    int grid = g == 0 ? (int)full : g;                                                    \
    float ms = timeit([&] { hipLaunchKernelGGL((copy_k<U, NT>), dim3(grid), dim3(256), 0, 0, d, s, n); }); \
    printf("copy U=%d nt=%d               %8d %10.1f\n", U, (int)NT, grid, 2.0 * bytes / ms / 1e6);       \
  }
#define RUN_SUM(U, NT)                                                                    \
  for (int g : grids) {                                                                   \
    long full = (n + 256L * U - 1) / (256L * U);                                          \
    int grid = g == 0 ? (int)full : g;                                                    \
    float ms = timeit([&] { hipLaunchKernelGGL((sum_k<U, NT>), dim3(grid), dim3(256), 0, 0, out, s, n); }); \
    printf("sum  U=%d nt=%d               %8d %10.1f\n", U, (int)NT, grid, 1.0 * bytes / ms / 1e6);       \
  }
  RUN_COPY(1, false) RUN_COPY(1, true)
  float* part = (float*)d;
#define RUN_SUMC(U, T)                                                                    \
  for (int grid : {512, 1024, 2048, 4096, 8192}) {                                        \
    long chunk = (n + grid - 1) / grid;                                                   \
    float ms = timeit([&] { hipLaunchKernelGGL((sumc_k<U, T>), dim3(grid), dim3(T), 0, 0, part, s, n, chunk); }); \
    printf("sumchunk U=%d T=%d          %8d %10.1f\n", U, T, grid, 1.0 * bytes / ms / 1e6);       \
  }
  RUN_SUMC(1, 256) RUN_SUMC(2, 256) RUN_SUMC(4, 256) RUN_SUMC(8, 256) RUN_SUMC(4, 512) RUN_SUMC(4, 1024)
  {
    const long R = 8192, C4 = 65536 / 4;
#define RUN_COL(U)                                                                        \
    for (int ys : {4, 8, 16, 32}) {                                                       \
      long rchunk = (R + ys - 1) / ys;                                                    \
      float ms = timeit([&] { hipLaunchKernelGGL((colsum_k<U>), dim3(C4 / 64, ys), dim3(256), 0, 0, part, s, R, C4, rchunk); }); \
      printf("colsum U=%d                 %3dx%-4d %10.1f\n", U, (int)(C4 / 64), ys, 1.0 * bytes / ms / 1e6); \
    }
    RUN_COL(1) RUN_COL(2) RUN_COL(4) RUN_COL(8)
  }
  float ms = timeit([&] { hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0); });
  printf("hipMemcpyAsync D2D                     - %10.1f\n", 2.0 * bytes / ms / 1e6);
  return 0;
}

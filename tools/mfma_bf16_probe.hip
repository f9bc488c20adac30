// What does v_mfma_f32_32x32x16_bf16 do with its 16 products and C?  One wave, element (0,0): products a_k*b_k
// (k = 0..15: lane 0 holds k 0..7, lane 32 holds k 8..15) + c.  Compared on the host with: (M1) fp32 chain in k order
// with round-to-nearest after every add, starting from c; (M2) exact sum, one round-to-nearest; (M3) exact, truncated.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <cstdint>
#include <vector>
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(const unsigned short* a, const unsigned short* b, const float* c, float* d, int cases) {
  const int lane = threadIdx.x;
  for (int t = 0; t < cases; ++t) {
    bf8 av, bv;
    for (int i = 0; i < 8; ++i) {
      unsigned short ua = 0, ub = 0;
      if (lane == 0) { ua = a[t * 16 + i]; ub = b[t * 16 + i]; }
      if (lane == 32) { ua = a[t * 16 + 8 + i]; ub = b[t * 16 + 8 + i]; }
      av[i] = __builtin_bit_cast(__bf16, ua);
      bv[i] = __builtin_bit_cast(__bf16, ub);
    }
    f16v acc = {0};
    if (lane == 0) acc[0] = c[t];
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
    if (lane == 0) d[t] = acc[0];
  }
}
static unsigned short to_bf16(float v) { uint32_t u; memcpy(&u, &v, 4); u += 0x7fff + ((u >> 16) & 1); return (unsigned short)(u >> 16); }
static float from_bf16(unsigned short h) { uint32_t u = (uint32_t)h << 16; float v; memcpy(&v, &u, 4); return v; }
int main() {
  const int N = 20000;
  std::vector<unsigned short> a(N * 16), b(N * 16);
  std::vector<float> c(N), d(N);
  srand(1);
  for (int t = 0; t < N; ++t) {
    const int mode = t % 5;
    for (int i = 0; i < 16; ++i) {
      float x = (rand() / (float)RAND_MAX - 0.5f) * 2.f, y = (rand() / (float)RAND_MAX - 0.5f) * 2.f;
      int e = mode == 0 ? 0 : (mode == 1 ? -(rand() % 12) : (mode == 2 ? -(rand() % 30) : (mode == 3 ? (i == 0 ? 0 : -24 - rand() % 4) : -(rand() % 20))));
      x = ldexpf(x, e);
      if (mode == 3 && i == 0) { x = 1.f; y = 1.f; }
      a[t * 16 + i] = to_bf16(x); b[t * 16 + i] = to_bf16(y);
    }
    c[t] = mode == 4 ? ldexpf((rand() / (float)RAND_MAX - 0.5f), rand() % 8) : (mode == 3 ? 0.f : (rand() / (float)RAND_MAX - 0.5f));
  }
  unsigned short *da, *db; float *dc, *dd;
  hipMalloc(&da, N * 32); hipMalloc(&db, N * 32); hipMalloc(&dc, N * 4); hipMalloc(&dd, N * 4);
  hipMemcpy(da, a.data(), N * 32, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), N * 32, hipMemcpyHostToDevice);
  hipMemcpy(dc, c.data(), N * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dc, dd, N);
  hipMemcpy(d.data(), dd, N * 4, hipMemcpyDeviceToHost);
  int m1 = 0, m1r = 0, m2 = 0, m3 = 0; double worst_vs_exact = 0;
  for (int t = 0; t < N; ++t) {
    float s1 = c[t], s1r = 0.f; long double ex = c[t]; long double mag = fabsl((long double)c[t]);
    for (int i = 0; i < 16; ++i) { float p = from_bf16(a[t * 16 + i]) * from_bf16(b[t * 16 + i]); s1 += p; ex += (long double)p; mag += fabsl((long double)p); }
    for (int i = 15; i >= 0; --i) s1r += from_bf16(a[t * 16 + i]) * from_bf16(b[t * 16 + i]);
    s1r += c[t];
    float s2 = (float)ex;
    float s3 = s2; if (fabsl((long double)s3) > fabsl(ex)) s3 = nextafterf(s3, 0.f);
    m1 += d[t] == s1; m1r += d[t] == s1r; m2 += d[t] == s2; m3 += d[t] == s3;
    double rel = (double)(fabsl((long double)d[t] - ex) / mag) / ldexp(1.0, -24);
    if (rel > worst_vs_exact) worst_vs_exact = rel;
  }
  printf("cases %d: == fp32 chain (k order, from c) %d, == chain reversed %d, == exact rounded once %d, == exact truncated %d; worst |d - exact| / (u * sum|terms|) = %.3f\n", N, m1, m1r, m2, m3, worst_vs_exact);
  for (int mode = 0; mode < 5; ++mode) {
    int n = 0, e2 = 0, e3 = 0; double w = 0;
    for (int t = mode; t < N; t += 5) {
      long double ex = c[t], mag = fabsl((long double)c[t]);
      for (int i = 0; i < 16; ++i) { float p = from_bf16(a[t * 16 + i]) * from_bf16(b[t * 16 + i]); ex += (long double)p; mag += fabsl((long double)p); }
      float s2 = (float)ex, s3 = s2; if (fabsl((long double)s3) > fabsl(ex)) s3 = nextafterf(s3, 0.f);
      ++n; e2 += d[t] == s2; e3 += d[t] == s3;
      double rel = (double)(fabsl((long double)d[t] - ex) / mag) / ldexp(1.0, -24); if (rel > w) w = rel;
    }
    printf("  mode %d: %d cases, exact-rounded %d, exact-truncated %d, worst %.3f u*sum|terms|\n", mode, n, e2, e3, w);
  }
  return 0;
}

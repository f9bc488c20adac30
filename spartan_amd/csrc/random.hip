// Counter-based random fills (Philox4x32-10) for the reference's srandom builders
// (spartan/expr/srandom.py:38-55: _make_rand / _make_randn / _make_randint run
// np.random.* per tile).  Element i of a fill is a pure function of (seed, offset + i),
// so a tile's content does not depend on the launch geometry; the host advances
// `offset` by the tile size after every fill.  The values are NOT NumPy's Mersenne
// Twister stream (the reference re-seeds every worker from the clock,
// srandom.py:23-35, so its values are not reproducible either).
#include "sp_common.hpp"

namespace {

struct U4 {
  uint32_t x, y, z, w;
};

__device__ __forceinline__ U4 philox4x32_10(uint64_t ctr_lo, uint64_t ctr_hi, uint64_t key) {
  uint32_t c0 = (uint32_t)ctr_lo, c1 = (uint32_t)(ctr_lo >> 32), c2 = (uint32_t)ctr_hi, c3 = (uint32_t)(ctr_hi >> 32);
  uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return U4{c0, c1, c2, c3};
}

// two 53-bit uniforms in [0, 1) from one Philox block
__device__ __forceinline__ void uniforms53(const U4& r, double& a, double& b) {
  a = (double)((((uint64_t)r.x << 32) | r.y) >> 11) * (1.0 / 9007199254740992.0);
  b = (double)((((uint64_t)r.z << 32) | r.w) >> 11) * (1.0 / 9007199254740992.0);
}

template <typename T> struct kIsI64 { static constexpr bool value = false; };
template <> struct kIsI64<int64_t> { static constexpr bool value = true; };

template <typename T>
__device__ __forceinline__ void put(void* out, int64_t i, double v) { ((T*)out)[i] = (T)v; }

// kind 0: uniform [0,1)   1: standard normal (Box-Muller)   2: integers in [lo, hi)
// One thread produces elements 2t and 2t+1 from Philox block (offset/2 + t).
template <typename T>
__global__ __launch_bounds__(256) void sp_random_kernel(void* __restrict__ out, int64_t n, int kind, uint64_t seed,
                                                        uint64_t offset, int64_t lo, uint64_t range) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t pairs = (n + 1) / 2;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < pairs; t += stride) {
    const U4 r = philox4x32_10(offset + (uint64_t)t, 0x5350415254414eull /* "SPARTAN" */, seed);
    double a, b;
    if (kind == 2) {
      const uint64_t ra = ((uint64_t)r.x << 32) | r.y, rb = ((uint64_t)r.z << 32) | r.w;
      a = (double)(lo + (int64_t)(ra % range));
      b = (double)(lo + (int64_t)(rb % range));
      if constexpr (kIsI64<T>::value) {   // int64 output: keep all 64 bits
        ((int64_t*)out)[2 * t] = lo + (int64_t)(ra % range);
        if (2 * t + 1 < n) ((int64_t*)out)[2 * t + 1] = lo + (int64_t)(rb % range);
        continue;
      }
    } else {
      uniforms53(r, a, b);
      if (kind == 1) {
        const double rad = sqrt(-2.0 * log(1.0 - a));   // 1 - a in (0, 1]
        const double ang = 6.283185307179586476925 * b;
        a = rad * cos(ang);
        b = rad * sin(ang);
      }
    }
    put<T>(out, 2 * t, a);
    if (2 * t + 1 < n) put<T>(out, 2 * t + 1, b);
  }
}

}  // namespace

extern "C" int sp_random_fill(void* d_out, int32_t dtype, int64_t n, int32_t kind, uint64_t seed, uint64_t offset,
                              int64_t lo, int64_t hi, void* stream) {
  if (n < 0) SP_FAIL("sp_random_fill: negative size");
  if (n == 0) return 0;
  if (!d_out) SP_FAIL("sp_random_fill: NULL pointer");
  if (kind < 0 || kind > 2) SP_FAIL("sp_random_fill: unknown kind %d", kind);
  if (kind == 2 && hi <= lo) SP_FAIL("sp_random_fill: empty integer range [%lld, %lld)", (long long)lo, (long long)hi);
  const uint64_t range = kind == 2 ? (uint64_t)(hi - lo) : 1;
  int64_t blocks = ((n + 1) / 2 + 255) / 256;
  if (blocks > SP_CUS * 16) blocks = SP_CUS * 16;
  hipStream_t st = (hipStream_t)stream;
  // element pairs are numbered from offset / 2: fills of even sizes tile one global stream
  const uint64_t base = offset / 2;
#define SP_RAND_GO(T) \
  hipLaunchKernelGGL((sp_random_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, st, d_out, n, kind, seed, base, lo, range)
  switch (dtype) {
    case SP_F32: SP_RAND_GO(float); break;
    case SP_F64: SP_RAND_GO(double); break;
    case SP_I64: SP_RAND_GO(int64_t); break;
    case SP_I32: SP_RAND_GO(int32_t); break;
    default: SP_FAIL("sp_random_fill: unsupported dtype %d", dtype);
  }
#undef SP_RAND_GO
  SP_CHECK_LAUNCH();
  return 0;
}

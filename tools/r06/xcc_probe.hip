#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void probe(int* out) {
  extern __shared__ char smem[];
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  if (threadIdx.x == 0) out[blockIdx.x] = (int)(x & 0xf);
  smem[threadIdx.x] = 1;
  for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(10);
}
int main() {
  int* d; hipMalloc(&d, 1024 * 4);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024);
  for (int grid : {256, 512}) {
    hipLaunchKernelGGL(probe, dim3(grid), dim3(512), 136 * 1024, 0, d);
    int h[1024]; hipMemcpy(h, d, grid * 4, hipMemcpyDeviceToHost);
    printf("grid %d:", grid);
    for (int i = 0; i < 40; ++i) printf(" %d", h[i]);
    int bad = 0; for (int i = 0; i < grid; ++i) bad += h[i] != h[i & 7];
    printf(" ... blocks whose XCC differs from block (i&7)'s: %d\n", bad);
  }
  return 0;
}

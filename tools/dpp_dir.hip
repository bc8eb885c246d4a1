// which way do the wavefront-wide DPP shifts move data on this chip?  (tools/, diagnostics only)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* p) {
  unsigned v = threadIdx.x + 100;
  p[threadIdx.x] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true);
  p[64 + threadIdx.x] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true);
}
int main() {
  unsigned *d, h[128];
  hipMalloc(&d, sizeof h);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  printf("wave_shl:1  lane0 <- %u lane1 <- %u lane15 <- %u lane16 <- %u lane31 <- %u lane62 <- %u lane63 <- %u\n", h[0], h[1], h[15], h[16], h[31], h[62], h[63]);
  printf("wave_shr:1  lane0 <- %u lane1 <- %u lane15 <- %u lane16 <- %u lane32 <- %u lane63 <- %u\n", h[64], h[65], h[79], h[80], h[96], h[127]);
  return 0;
}

// Cycles per pair squaring / product of the split form (csrc/hensel.hpp: pairmul) on lone wavefronts: the unit-quotient
// loop modulus P = p*k (38 limbs for a 1024-bit p) against the prime itself (36 limbs, one more multiplication in front of
// every quotient digit).  Timing only -- the operands are arbitrary limbs.  (tools/, diagnostics only)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include "../pailliercryptolib_amd/csrc/hensel.hpp"   // (-DPGPU_QDIGIT_MAD=0: digits by v_mul_lo_u32)
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
using namespace pgpu;

template <int H, int K, bool UQ>
__global__ __launch_bounds__(kWGThreads, 1) void pm_kernel(uint32_t* io, const uint32_t* nmod, uint32_t n0inv_, int steps, int nsq) {
  constexpr int GS = 2 * H;
  raise_wave_priority();
  const int lane = threadIdx.x % kWave;
  const int xg = lane % GS, x = xg % H;
  const uint32_t halfB = (uint32_t)(xg / H);
  uint32_t selB = xg == H ? 1u : 0u;
  asm("" : "+v"(selB));
  uint32_t n[K], own[K], mreg[K];
  uint32_t* p = io + ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * K;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    n[j] = nmod[x * K + j];
    own[j] = p[j] & kLimbMask;
    mreg[j] = own[j] ^ 0x155u;
  }
  const uint32_t n0inv = UQ ? 0u : n0inv_;
  asm volatile(".p2align 6");
#pragma unroll 1
  for (int s = 0; s < steps; ++s) {
#pragma unroll 1
    for (int i = 0; i < nsq; ++i) pairmul<H, K, true, UQ>(own, own, own, n, n0inv, halfB, selB);
    pairmul<H, K, false, UQ>(own, own, mreg, n, n0inv, halfB, selB);
  }
#pragma unroll
  for (int j = 0; j < K; ++j) p[j] = own[j];
}

typedef void (*kern_t)(uint32_t*, const uint32_t*, uint32_t, int, int);
int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount; const double clk = prop.clockRate * 1e3;
  uint32_t *io, *nm; CK(hipMalloc(&io, (size_t)4 * cus * 2 * 256 * 20)); CK(hipMalloc(&nm, 4 * 256));
  CK(hipMemset(io, 0x5a, (size_t)4 * cus * 2 * 256 * 20)); CK(hipMemset(nm, 0x13, 4 * 256));
  struct B { const char* name; kern_t k; int limbs; } bs[] = {
      {"pair<2,19> unit quotient (P = p*k)", pm_kernel<2, 19, true>, 38},
      {"pair<2,19> digits by multiplication", pm_kernel<2, 19, false>, 38},
      {"pair<2,18> digits by multiplication (P = p)", pm_kernel<2, 18, false>, 36},
      {"pair<2,18> unit quotient (for reference)", pm_kernel<2, 18, true>, 36}};
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int steps = 205, nsq = 5;    // one 1024-bit exponent in 5-bit windows
  for (int wps : {1, 2}) {
    printf("--- %d wavefront(s) per SIMD, %d steps of %d squarings + 1 product ---\n", wps, steps, nsq);
    for (auto& b : bs) {
      float best = 1e9;
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(b.k, dim3(cus * wps), dim3(256), 0, 0, io, nm, 0x12345677u, steps, nsq);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      printf("%-46s %8.3f ms   %.0f cycles per step and wavefront\n", b.name, best, best * 1e-3 * clk / steps / wps);
    }
  }
  return 0;
}

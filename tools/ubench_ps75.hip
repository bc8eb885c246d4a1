// Microbenchmark (tools/, diagnostics only; round 6): feasibility of a ONE-LANE product-scanning form for the n^2 domain of
// 2048-bit keys -- halves of 75 limbs of 28 bits (R = 2^2100 >= 16 P, P = n k < 2^2076) -- on the one-wavefront-per-SIMD
// build (512 registers: 256 VGPRs + 256 AGPRs).  5 pair squarings + 1 pair product per window, like CT x PT / r^n.
// Shader clock read inside the kernel (s_memtime), as in tools/ubench_ps.hip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "hensel_ps.hpp"
using namespace pgpu;
#ifndef UB_K
#define UB_K 75
#endif
#ifndef UB_LB
#define UB_LB 28
#endif

template <int K, int LB>
__global__ __launch_bounds__(256, 1) void sq_kernel(const uint32_t* in, const uint32_t* nn, uint32_t* out, unsigned long long* cyc, int iters) {
  extern __shared__ uint4 park_[];      // [kWavesPerWG][K4][kWave]
  constexpr int K4 = (K + 3) / 4;
  uint32_t a[K], b[K], n[K], c[K], d[K];
  const int lane = threadIdx.x + blockIdx.x * 256;
  uint4* slot = park_ + (size_t)(threadIdx.x / kWave) * K4 * kWave + threadIdx.x % kWave;
  __builtin_amdgcn_s_setprio(3);
#pragma unroll
  for (int j = 0; j < K; ++j) {
    a[j] = in[(size_t)lane * 2 * K + j];
    b[j] = in[(size_t)lane * 2 * K + K + j];
    n[j] = ps_uniform(nn[j]);
  }
  const uint32_t n1p = n[1] + 1;
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int w = 0; w < iters; ++w) {
#pragma unroll 1
    for (int i = 0; i < 5; ++i) ps_pairsqr<K, LB>(a, b, n, n1p);
#pragma unroll
    for (int j = 0; j < K; ++j) {
      c[j] = in[(size_t)lane * 2 * K + j] ^ (w & 1);
      d[j] = in[(size_t)lane * 2 * K + K + j] ^ (w & 2);
    }
    ps_pairmul<K, LB, true>(a, b, c, d, n, n1p, 0, slot);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x % kWave == 0) cyc[lane / kWave] = t1 - t0;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    out[(size_t)lane * 2 * K + j] = a[j];
    out[(size_t)lane * 2 * K + K + j] = b[j];
  }
}

int main(int argc, char** argv) {
  constexpr int K = UB_K, LB = UB_LB;
  const double ipi = argc > 1 ? atof(argv[1]) : 5 * 20500.0 + 29000.0;
  const double mac = argc > 2 ? atof(argv[2]) : 0;
  const int iters = 7;     // 32-bit exponents of CT x PT: 7 windows of 5
  const unsigned lds = sizeof(uint4) * kWavesPerWG * ((K + 3) / 4) * kWave;
  hipFuncSetAttribute((const void*)sq_kernel<K, LB>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  for (int blocks : {64, 256, 1024, 4096}) {
    const size_t lanes = (size_t)blocks * 256, waves = lanes / 64;
    std::vector<uint32_t> h(lanes * 2 * K), hn(K);
    srand(1);
    for (auto& v : h) v = ((uint32_t)rand() * 2654435761u) & ((1u << LB) - 1);
    for (auto& v : hn) v = ((uint32_t)rand() * 2654435761u) & ((1u << LB) - 1);
    hn[0] = (1u << LB) - 1;
    uint32_t *din, *dn, *dout;
    unsigned long long* dcyc;
    hipMalloc(&din, h.size() * 4); hipMalloc(&dn, hn.size() * 4); hipMalloc(&dout, h.size() * 4); hipMalloc(&dcyc, waves * 8);
    hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dn, hn.data(), hn.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<unsigned long long> cyc(waves);
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL((sq_kernel<K, LB>), dim3(blocks), dim3(256), lds, 0, din, dn, dout, dcyc, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(cyc.data(), dcyc, waves * 8, hipMemcpyDeviceToHost);
      std::sort(cyc.begin(), cyc.end());
      double mean = 0;
      for (auto v : cyc) mean += (double)v;
      mean /= waves;
      const double instr = iters * ipi;
      if (rep) printf("K=%d LB=%d blocks=%5d (%7zu elements, %d windows)  wall %8.3f ms | %.3f cycles/instr (s_memtime) | %.2f T MAC32/s executed = %.3f of 39.32 | %.2f us per element-window\n",
                      K, LB, blocks, lanes, iters, ms, mean / instr, mac * iters * lanes / (ms * 1e-3) / 1e12, mac * iters * lanes / (ms * 1e-3) / 1e12 / 39.32,
                      ms * 1e3 / iters);
    }
    hipFree(din); hipFree(dn); hipFree(dout); hipFree(dcyc);
  }
  return 0;
}

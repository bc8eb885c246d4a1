// Microbenchmark (tools/, diagnostics only): the product-scanning pair product of csrc/hensel_ps.hpp as a lone wavefront
// per SIMD and with two per SIMD -- 5 squarings + 1 general product per iteration, 205 iterations = one 1024-bit
// exponentiation of the 2048-bit key class.  Prints ms per launch and cycles per instruction at the clock the box held.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Ipailliercryptolib_amd/csrc tools/ubench_ps.hip -o tools/ubench_ps
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "hensel_ps.hpp"
using namespace pgpu;

template <int K, int LB>
__global__ __launch_bounds__(256, 2) void sq_kernel(const uint32_t* in, const uint32_t* nn, uint32_t* out, int iters, int side) {
  extern __shared__ uint32_t claim[];
  uint32_t a[K], b[K], n[K], c[K], d[K];
  const int lane = threadIdx.x + blockIdx.x * 256;
  const uint32_t* np = nn + __builtin_amdgcn_readfirstlane(side) * K;
  __builtin_amdgcn_s_setprio(3);
#pragma unroll
  for (int j = 0; j < K; ++j) { a[j] = in[(size_t)lane * 2 * K + j]; b[j] = in[(size_t)lane * 2 * K + K + j]; n[j] = np[j]; }
#pragma unroll 1
  for (int w = 0; w < iters; ++w) {
#pragma unroll 1
    for (int i = 0; i < 5; ++i) ps_pairmul<K, LB, true, true>(a, b, a, b, n, 0);
#pragma unroll
    for (int j = 0; j < K; ++j) { c[j] = in[(size_t)lane * 2 * K + j] ^ (w & 1); d[j] = in[(size_t)lane * 2 * K + K + j] ^ (w & 2); }
    ps_pairmul<K, LB, false, true>(a, b, c, d, n, 0);
  }
#pragma unroll
  for (int j = 0; j < K; ++j) { out[(size_t)lane * 2 * K + j] = a[j]; out[(size_t)lane * 2 * K + K + j] = b[j]; }
}

template <int K, int LB>
void run(const char* name, int blocks, unsigned lds, int iters, double instr_per_iter) {
  const size_t lanes = (size_t)blocks * 256;
  std::vector<uint32_t> h(lanes * 2 * K), hn(2 * K);
  srand(1);
  for (auto& v : h) v = ((uint32_t)rand() * 2654435761u) & ((1u << LB) - 1);
  for (auto& v : hn) v = ((uint32_t)rand() * 2654435761u) & ((1u << LB) - 1);
  hn[0] = hn[K] = (1u << LB) - 1;
  uint32_t *din, *dn, *dout;
  hipMalloc(&din, h.size() * 4); hipMalloc(&dn, hn.size() * 4); hipMalloc(&dout, h.size() * 4);
  hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dn, hn.data(), hn.size() * 4, hipMemcpyHostToDevice);
  if (lds) hipFuncSetAttribute((const void*)sq_kernel<K, LB>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((sq_kernel<K, LB>), dim3(blocks), dim3(256), lds, 0, din, dn, dout, iters, 0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep) printf("%-44s K=%d LB=%d blocks=%d lds=%u  %.3f ms   %.2f ns per instruction and wave  (%.2f cycles at 2.4 GHz)\n", name, K, LB, blocks,
                    lds, ms, ms * 1e6 / (iters * instr_per_iter), ms * 1e6 / (iters * instr_per_iter) * 2.4);
  }
  hipFree(din); hipFree(dn); hipFree(dout);
}

int main() {
  // instructions per iteration from tools/asm_stats.py on this file's code object (5 squarings + 1 product)
  const double ipi38 = 5 * 5560.0 + 7700.0, ipi20 = 5 * 1640.0 + 2250.0;
  run<38, 28>("one wavefront per SIMD (CU claim)", 256, 84000, 205, ipi38);
  run<38, 28>("two wavefronts per SIMD", 512, 0, 205, ipi38);
  run<38, 28>("quarter chip (64 workgroups, CU claim)", 64, 84000, 205, ipi38);
  run<20, 29>("K=20: two wavefronts per SIMD", 512, 0, 205, ipi20);
  run<20, 29>("K=20: four wavefronts per SIMD", 1024, 0, 205, ipi20);
  return 0;
}

// Microbenchmark (tools/, diagnostics only; round 6): the pair squaring / pair product of csrc/hensel_ps.hpp exactly as
// hensel_decrypt_ps_kernel runs them -- 5 squarings + 1 general product per window, 205 windows = one 1024-bit exponentiation
// of the 2048-bit key class -- with the SHADER clock read inside the kernel: every wavefront brackets its loop with s_memtime
// (tick = one shader cycle, MI355X_MICROARCH.md), so that cycles per instruction and the clock the chip held are two separate
// numbers:   cycles/instr = s_memtime span / instructions of the loop;   clock = s_memtime span / wall time of the launch.
// Round 5's version printed "cycles at 2.4 GHz" computed from WALL time, which cannot tell cadence from clock.
// Instructions per window: counted by tools/asm_stats.py on this file's code object and passed as argv[1] (default below).
// build: python tools/build_ubench.py ubench_ps [-DPGPU_PS_SPLIT=1]   (the library's compile step, alignment pass included)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "hensel_ps.hpp"
using namespace pgpu;

template <int K, int LB, int MINW>
__global__ __launch_bounds__(256, MINW) void sq_kernel(const uint32_t* in, const uint32_t* nn, uint32_t* out, unsigned long long* cyc,
                                                       int iters, int side) {
  extern __shared__ uint32_t claim[];
  __shared__ uint4 park_[kWavesPerWG][(K + 3) / 4][kWave];
  uint32_t a[K], b[K], n[K], c[K], d[K];
  const int lane = threadIdx.x + blockIdx.x * 256;
  uint4* slot = &park_[threadIdx.x / kWave][0][threadIdx.x % kWave];
  const uint32_t* np = nn + __builtin_amdgcn_readfirstlane(side) * K;
  __builtin_amdgcn_s_setprio(3);
#pragma unroll
  for (int j = 0; j < K; ++j) {
    a[j] = in[(size_t)lane * 2 * K + j];
    b[j] = in[(size_t)lane * 2 * K + K + j];
    n[j] = ps_uniform(np[j]);
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

template <int K, int LB, int MINW>
void run(const char* name, int blocks, unsigned lds, int iters, double instr_per_iter) {
  const size_t lanes = (size_t)blocks * 256, waves = lanes / 64;
  std::vector<uint32_t> h(lanes * 2 * K), hn(2 * K);
  srand(1);
  for (auto& v : h) v = ((uint32_t)rand() * 2654435761u) & ((1u << LB) - 1);
  for (auto& v : hn) v = ((uint32_t)rand() * 2654435761u) & ((1u << LB) - 1);
  hn[0] = hn[K] = (1u << LB) - 1;
  uint32_t *din, *dn, *dout;
  unsigned long long* dcyc;
  hipMalloc(&din, h.size() * 4); hipMalloc(&dn, hn.size() * 4); hipMalloc(&dout, h.size() * 4); hipMalloc(&dcyc, waves * 8);
  hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dn, hn.data(), hn.size() * 4, hipMemcpyHostToDevice);
  if (lds) hipFuncSetAttribute((const void*)sq_kernel<K, LB, MINW>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<unsigned long long> cyc(waves);
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((sq_kernel<K, LB, MINW>), dim3(blocks), dim3(256), lds, 0, din, dn, dout, dcyc, iters, 0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(cyc.data(), dcyc, waves * 8, hipMemcpyDeviceToHost);
    std::sort(cyc.begin(), cyc.end());
    double mean = 0;
    for (auto v : cyc) mean += (double)v;
    mean /= waves;
    const double instr = iters * instr_per_iter;
    if (rep) printf("%-46s K=%d blocks=%4d  wall %7.3f ms | s_memtime span per wave: mean %.4g (min %.4g max %.4g) cycles "
                    "= %.3f cycles/instr | clock held %.3f GHz | wall-time figure %.2f ns/instr\n",
                    name, K, blocks, ms, mean, (double)cyc.front(), (double)cyc.back(), mean / instr, (double)cyc.back() / (ms * 1e6),
                    ms * 1e6 / instr);
  }
  hipFree(din); hipFree(dn); hipFree(dout); hipFree(dcyc);
}

int main(int argc, char** argv) {
  // instructions per window (5 squarings + 1 product + the loop's own) of sq_kernel<38,28,*>: tools/asm_stats.py
  const double ipi38 = argc > 1 ? atof(argv[1]) : 5 * 5560.0 + 7700.0;
  const double ipi38w1 = argc > 2 ? atof(argv[2]) : ipi38;
  printf("# PGPU_PS_SPLIT=%d  instructions per window: %.0f (two wavefronts per SIMD build) / %.0f (one wavefront per SIMD build)\n",
         PGPU_PS_SPLIT, ipi38, ipi38w1);
  run<38, 28, 1>("lone quarter chip (64 WGs, CU claim, MINW=1)", 64, 84000, 205, ipi38w1);
  run<38, 28, 1>("full chip, one wavefront per SIMD (MINW=1)", 256, 84000, 205, ipi38w1);
  run<38, 28, 2>("full chip, one wavefront per SIMD (MINW=2)", 256, 84000, 205, ipi38);
  run<38, 28, 2>("full chip, two wavefronts per SIMD (MINW=2)", 512, 0, 205, ipi38);
  run<38, 28, 1>("one WG only (4 waves on one CU, MINW=1)", 1, 84000, 205, ipi38w1);
  return 0;
}

// Micro-experiment: how do TWO waves resident on one SIMD share the VALU?  One 512-thread
// workgroup per CU (waves w and w+4 share a SIMD); every wave runs the same loop and records
// its own duration (s_memtime).  Variants add stalls (s_nop, DPP) and s_setprio.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int NITER = 20000;

template <int VARIANT>
__global__ __launch_bounds__(512) void k(uint64_t* out, uint64_t* clocks, uint32_t a, uint32_t b) {
  const int wv = threadIdx.x / 64;
  uint64_t acc[8]; uint32_t x = a + threadIdx.x, y = b ^ threadIdx.x, z = 0;
  for (int i = 0; i < 8; ++i) acc[i] = i + threadIdx.x;
  if (VARIANT == 2 || VARIANT == 5) { if (wv >= 4) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0); }
  if (VARIANT == 3) { if (wv < 4) __builtin_amdgcn_s_setprio(3); }
  __syncthreads();
  uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < NITER; ++it) {
#define S(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y) : "vcc");
    S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
    if (VARIANT >= 1) {
      // a dependent chain with a DPP hop and a nop, like the quotient-digit chain of the real kernel
      asm volatile("v_mul_lo_u32 %0, %1, %2\n\ts_nop 1\n\tv_and_b32_dpp %0, %0, %3 row_newbcast:0 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                   : "=&v"(z) : "v"((uint32_t)acc[0]), "v"(y), "v"(x));
      x ^= z & 1;
    }
    if (VARIANT == 4 || VARIANT == 5) { if ((it & 63) == 0) asm volatile("s_waitcnt lgkmcnt(0)"); }
    S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#undef S
  }
  uint64_t t1 = __builtin_readcyclecounter();
  uint64_t s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + z;
  if ((threadIdx.x & 63) == 0) clocks[blockIdx.x * 8 + wv] = t1 - t0;
}

template <int V> void run(const char* name, uint64_t* out, uint64_t* clocks, int cus) {
  hipLaunchKernelGGL(k<V>, dim3(cus), dim3(512), 0, 0, out, clocks, 3u, 5u);
  CK(hipDeviceSynchronize());
  std::vector<uint64_t> h(cus * 8);
  CK(hipMemcpy(h.data(), clocks, h.size() * 8, hipMemcpyDeviceToHost));
  double lo = 0, hi = 0;
  for (int c = 0; c < cus; ++c) for (int w = 0; w < 4; ++w) { lo += h[c * 8 + w]; hi += h[c * 8 + w + 4]; }
  lo /= cus * 4; hi /= cus * 4;
  printf("%-46s waves 0-3: %10.0f cyc   waves 4-7: %10.0f cyc   ratio %.3f   per-iter %.1f / %.1f\n", name, lo, hi, hi / lo, lo / NITER, hi / NITER);
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  uint64_t *out, *clocks; CK(hipMalloc(&out, 8 * 512 * cus)); CK(hipMalloc(&clocks, 8 * 8 * cus));
  run<0>("0 pure MACs (16/iter)", out, clocks, cus);
  run<1>("1 MACs + mul_lo/nop/dpp chain", out, clocks, cus);
  run<2>("2 chain, waves 4-7 s_setprio 3", out, clocks, cus);
  run<3>("3 chain, waves 0-3 s_setprio 3", out, clocks, cus);
  run<4>("4 chain + periodic s_waitcnt", out, clocks, cus);
  run<5>("5 chain + waitcnt, waves 4-7 s_setprio 3", out, clocks, cus);
  return 0;
}

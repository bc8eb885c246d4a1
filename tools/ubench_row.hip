// Single-wave issue behaviour of the Montgomery row pattern (tools/, not part of the product).
// hipcc --offload-arch=gfx950 -O3 -o ubench_row ubench_row.hip && ./ubench_row
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int NITER = 2048;
constexpr int K = 18;

// A: K independent MACs per row, q changes per row through a cheap xor (no carry chain)
__global__ void k_macs_only(uint64_t* out, const uint32_t* in) {
  uint64_t acc[K]; uint32_t n[K];
  for (int j = 0; j < K; ++j) { acc[j] = threadIdx.x + j; n[j] = in[threadIdx.x + 64 * j]; }
  uint32_t q = in[threadIdx.x];
  for (int it = 0; it < NITER; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int j = 0; j < K; ++j) acc[j] += (uint64_t)n[j] * q;
      q ^= 0x9e3779b9u + r;
    }
  }
  uint64_t s = 0; for (int j = 0; j < K; ++j) s += acc[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// B: the full row: q from column 0 over DPP, K MACs, limb hand-over, carry into column 1, window slide
template <int ORDER>
__global__ void k_row(uint64_t* out, const uint32_t* in) {
  uint64_t acc[K + 1]; uint32_t n[K];
  for (int j = 0; j < K; ++j) { acc[j] = threadIdx.x + j; n[j] = in[threadIdx.x + 64 * j]; }
  acc[K] = 0;
  uint32_t m = 0x1fffffff; asm("" : "+v"(m));
  for (int it = 0; it < NITER; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      uint32_t q = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)acc[0], 0x00, 0xf, 0xf, true) & m;
      if (ORDER == 0) {
#pragma unroll
        for (int j = 0; j < K; ++j) acc[j] += (uint64_t)n[j] * q;
        uint32_t recv = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)acc[0], 0x101, 0xf, 0xf, true) & m;
        uint64_t c = acc[0] >> 29;
        acc[1] += c;
        acc[K] += recv;
      } else {
        __builtin_amdgcn_sched_barrier(0x3fc);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += (uint64_t)n[j] * q;
        __builtin_amdgcn_sched_barrier(0x3fc);
        uint32_t recv = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)acc[0], 0x101, 0xf, 0xf, true) & m;
        uint64_t c = acc[0] >> 29;
        asm volatile("" : "+v"(c));
        __builtin_amdgcn_sched_barrier(0x3fc);
#pragma unroll
        for (int j = 4; j < 6; ++j) acc[j] += (uint64_t)n[j] * q;
        __builtin_amdgcn_sched_barrier(0x3fc);
        acc[1] += c;
        acc[K] += recv;
        __builtin_amdgcn_sched_barrier(0x3fc);
#pragma unroll
        for (int j = 6; j < K; ++j) acc[j] += (uint64_t)n[j] * q;
        __builtin_amdgcn_sched_barrier(0x3fc);
      }
      // window slide (register renaming only)
#pragma unroll
      for (int j = 0; j < K; ++j) acc[j] = acc[j + 1];
      acc[K] = 0;
    }
  }
  uint64_t s = 0; for (int j = 0; j < K; ++j) s += acc[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef void (*kern_t)(uint64_t*, const uint32_t*);
int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount; double clk = prop.clockRate * 1e3;
  uint64_t* out; uint32_t* in;
  CK(hipMalloc(&out, 8 * 256 * 4096)); CK(hipMalloc(&in, 4 * 64 * 64)); CK(hipMemset(in, 0x5a, 4 * 64 * 64));
  struct B { const char* name; kern_t k; } bs[] = {
      {"18 MACs/row, no chain", k_macs_only}, {"full row, compiler order", k_row<0>}, {"full row, pinned order", k_row<1>}};
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int wps : {1, 2, 3}) {
    printf("--- %d wave(s)/SIMD ---\n", wps);
    for (auto& b : bs) {
      float best = 1e9;
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(b.k, dim3(cus * wps), dim3(256), 0, 0, out, in);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      double rows = (double)NITER * 8 * wps;   // rows per SIMD
      printf("%-28s %8.3f ms  %7.1f cycles/row/SIMD  (%.2f per MAC)\n", b.name, best, best * 1e-3 * clk / rows,
             best * 1e-3 * clk / rows / K);
    }
  }
  return 0;
}

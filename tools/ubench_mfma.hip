// Can one wave overlap v_mfma_i32_16x16x64_i8 with v_mad_u64_u32?  (feasibility probe for the MFMA-assisted
// reduction of DESIGN.md section 8; diagnostics only)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int NITER = 4096;
typedef int v4i __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: 8 MFMA per iteration; 1: 32 MACs; 2: both interleaved
__global__ void k(uint64_t* out, const int* in) {
  v4i a = {in[threadIdx.x], in[threadIdx.x + 64], 3, 4}, b = {5, in[threadIdx.x + 128], 7, 8};
  v4i c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  uint64_t acc[8]; uint32_t x = in[threadIdx.x] | 1, y = in[threadIdx.x + 64] | 3;
  for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x + i;
  for (int it = 0; it < NITER; ++it) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      if (MODE != 1) {
        c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c1, 0, 0, 0);
      }
      if (MODE != 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y) : "s20", "s21");
      }
      if (MODE != 1) {
        c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c3, 0, 0, 0);
      }
      if (MODE != 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y) : "s20", "s21");
      }
    }
  }
  uint64_t s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
  v4i c = c0 + c1 + c2 + c3;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + c.x + c.y + c.z + c.w;
}
typedef void (*kern_t)(uint64_t*, const int*);
int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount; double clk = prop.clockRate * 1e3;
  uint64_t* out; int* in; CK(hipMalloc(&out, 8 * 256 * 4096)); CK(hipMalloc(&in, 4 * 1024)); CK(hipMemset(in, 0x11, 4 * 1024));
  struct B { const char* name; kern_t k; } bs[] = {{"8 MFMA 16x16x64 i8 per iteration", k<0>}, {"32 v_mad_u64_u32 per iteration", k<1>}, {"both interleaved", k<2>}};
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int wps : {1, 2}) {
    printf("--- %d wave(s)/SIMD ---\n", wps);
    for (auto& b : bs) {
      float best = 1e9;
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(b.k, dim3(cus * wps), dim3(256), 0, 0, out, in);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      printf("%-36s %8.3f ms  %8.1f cycles per iteration per SIMD\n", b.name, best, best * 1e-3 * clk / ((double)NITER * wps));
    }
  }
  return 0;
}

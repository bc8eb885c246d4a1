// Cost of moving a value between 16-lane DPP rows on gfx950 (needed by an element-per-lane-row layout,
// DESIGN.md section 8): v_permlane16_swap / v_permlane32_swap vs ds_bpermute_b32 hidden behind MACs.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int NITER = 8192;
#define CLOB "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v18","v19","v20","v21","v22","v23","s20","s21"
#define MAC(acc) "v_mad_u64_u32 v[" #acc "], s[20:21], v18, v19, v[" #acc "]\n\t"
#define MAC8 MAC(32:33) MAC(34:35) MAC(36:37) MAC(38:39) MAC(40:41) MAC(42:43) MAC(44:45) MAC(46:47)
template <int V>
__global__ void k(uint64_t* out) {
  asm volatile("v_mov_b32 v18, 7\n\tv_mov_b32 v19, 9\n\tv_mov_b32 v20, 1\n\tv_mov_b32 v21, 2\n\t"
               "v_lshlrev_b32 v22, 2, v0\n\tv_xor_b32 v22, 64, v22" ::: CLOB);   // v22: byte address of lane ^ 16
  for (int it = 0; it < NITER; ++it) {
    if (V == 0) asm volatile(MAC8 MAC8 ::: CLOB);
    if (V == 1) asm volatile(MAC8 "v_permlane16_swap_b32 v20, v21\n\t" MAC8 "v_permlane16_swap_b32 v20, v21\n\t" ::: CLOB);
    if (V == 2) asm volatile(MAC8 "v_permlane32_swap_b32 v20, v21\n\t" MAC8 "v_permlane32_swap_b32 v20, v21\n\t" ::: CLOB);
    if (V == 3) asm volatile("ds_bpermute_b32 v23, v22, v20\n\t" MAC8 "s_waitcnt lgkmcnt(0)\n\tv_add_u32 v20, v20, v23\n\t"
                             "ds_bpermute_b32 v23, v22, v20\n\t" MAC8 "s_waitcnt lgkmcnt(0)\n\tv_add_u32 v20, v20, v23\n\t" ::: CLOB);
  }
  uint32_t r; asm volatile("v_add_u32 %0, v32, v20" : "=v"(r));
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
typedef void (*kern_t)(uint64_t*);
int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount; double clk = prop.clockRate * 1e3;
  uint64_t* out; CK(hipMalloc(&out, 8 * 256 * 4096));
  struct B { const char* name; kern_t k; } bs[] = {{"16 MACs", k<0>}, {"16 MACs + 2 v_permlane16_swap", k<1>},
      {"16 MACs + 2 v_permlane32_swap", k<2>}, {"16 MACs + 2 (ds_bpermute, wait, add)", k<3>}};
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int wps : {1, 2}) {
    printf("--- %d wave(s)/SIMD ---\n", wps);
    for (auto& b : bs) {
      float best = 1e9;
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(b.k, dim3(cus * wps), dim3(256), 0, 0, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      printf("%-40s %8.3f ms  %7.1f cycles per iteration per SIMD\n", b.name, best, best * 1e-3 * clk / ((double)NITER * wps));
    }
  }
  return 0;
}

// Does a wave64 VALU instruction with only 32 (or 16) active lanes issue faster on gfx950?  (diagnostics)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int NITER = 8192;
#define MAC(acc) "v_mad_u64_u32 v[" #acc "], s[20:21], v18, v19, v[" #acc "]\n\t"
#define ROW MAC(32:33) MAC(34:35) MAC(36:37) MAC(38:39) MAC(40:41) MAC(42:43) MAC(44:45) MAC(46:47) \
            MAC(48:49) MAC(50:51) MAC(52:53) MAC(54:55) MAC(56:57) MAC(58:59) MAC(60:61) MAC(62:63)
#define CLOB "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47", \
  "v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v18","v19","s20","s21"
__global__ void k(uint64_t* out, int active) {
  asm volatile("v_mov_b32 v18, 7\n\tv_mov_b32 v19, 9" ::: CLOB);
  if ((int)(threadIdx.x % 64) < active) {
    for (int it = 0; it < NITER; ++it) asm volatile(ROW ROW ROW ROW ::: CLOB);
  }
  uint32_t r; asm volatile("v_mov_b32 %0, v32" : "=v"(r));
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount; double clk = prop.clockRate * 1e3;
  uint64_t* out; CK(hipMalloc(&out, 8 * 256 * 4096));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int wps : {1, 2, 4}) {
    for (int active : {64, 32, 16}) {
      float best = 1e9;
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(cus * wps), dim3(256), 0, 0, out, active);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      printf("%d wave(s)/SIMD, %2d active lanes: %8.3f ms  %.2f cycles per wave-instruction per SIMD\n", wps, active, best,
             best * 1e-3 * clk / ((double)NITER * 64 * wps));
    }
  }
  return 0;
}

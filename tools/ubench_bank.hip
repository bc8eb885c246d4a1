// Does VGPR bank placement of v_mad_u64_u32 operands matter for a lone wave?  (tools/, diagnostics only)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int NITER = 8192;

#define MAC(acc, a, b, sd) "v_mad_u64_u32 v[" #acc "], " sd ", v" #a ", v" #b ", v[" #acc "]\n\t"
// 16 MACs, accumulators v[32:33]..v[62:63]; A = src0, B = src1
#define ROW(A, B, SD) \
  MAC(32:33, A, B, SD) MAC(34:35, A, B, SD) MAC(36:37, A, B, SD) MAC(38:39, A, B, SD) \
  MAC(40:41, A, B, SD) MAC(42:43, A, B, SD) MAC(44:45, A, B, SD) MAC(46:47, A, B, SD) \
  MAC(48:49, A, B, SD) MAC(50:51, A, B, SD) MAC(52:53, A, B, SD) MAC(54:55, A, B, SD) \
  MAC(56:57, A, B, SD) MAC(58:59, A, B, SD) MAC(60:61, A, B, SD) MAC(62:63, A, B, SD)
#define CLOB "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47", \
  "v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63", \
  "v16","v17","v18","v19","v20","v21","v22","v23","s20","s21","s22","s23","vcc"

template <int V>
__global__ void k(uint64_t* out) {
  asm volatile("v_mov_b32 v16, 3\n\tv_mov_b32 v17, 5\n\tv_mov_b32 v18, 7\n\tv_mov_b32 v19, 9\n\t"
               "v_mov_b32 v20, 11\n\tv_mov_b32 v21, 13\n\tv_mov_b32 v22, 15\n\tv_mov_b32 v23, 17" ::: CLOB);
  for (int it = 0; it < NITER; ++it) {
    if (V == 0) asm volatile(ROW(18, 19, "s[20:21]") ROW(18, 19, "s[20:21]") ROW(18, 19, "s[20:21]") ROW(18, 19, "s[20:21]") ::: CLOB);        // acc even pair (banks 0,1 / 2,3), src banks 2,3
    if (V == 1) asm volatile(ROW(16, 20, "s[20:21]") ::: CLOB);        // src0, src1 both bank 0
    if (V == 2) asm volatile(ROW(16, 17, "s[20:21]") ::: CLOB);        // src banks 0,1
    if (V == 3) asm volatile(ROW(18, 19, "vcc") ::: CLOB);             // carry-out to vcc
    if (V == 4) asm volatile(ROW(18, 18, "s[20:21]") ::: CLOB);        // same register twice
  }
  uint32_t r; asm volatile("v_mov_b32 %0, v32" : "=v"(r));
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
typedef void (*kern_t)(uint64_t*);
int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount; double clk = prop.clockRate * 1e3;
  uint64_t* out; CK(hipMalloc(&out, 8 * 256 * 4096));
  struct B { const char* name; kern_t k; } bs[] = {
      {"64 MACs/iteration, src banks 2,3", k<0>}, {"src0,src1 same bank 0", k<1>}, {"src banks 0,1", k<2>},
      {"carry-out to vcc", k<3>}, {"src0 == src1", k<4>}};
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int wps : {1, 2, 4, 8}) {
    printf("--- %d wave(s)/SIMD ---\n", wps);
    for (auto& b : bs) {
      float best = 1e9;
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(b.k, dim3(cus * wps), dim3(256), 0, 0, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      double macs = (double)NITER * 16 * wps * (&b == &bs[0] ? 4 : 1);
      printf("%-46s %8.3f ms  %.3f cycles/MAC/SIMD  %.2f T MAC32/s\n", b.name, best, best * 1e-3 * clk / macs, macs * cus * 4 * 64 / (best * 1e-3) / 1e12);
    }
  }
  return 0;
}

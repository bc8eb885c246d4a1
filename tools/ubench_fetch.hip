// Is a lone wave's 5-cycle issue cadence an instruction-fetch limit?  Same operation in the 4-byte (e32) and
// the 8-byte (e64) encoding, 1 wave per SIMD vs 2.  (tools/, diagnostics only)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int NITER = 8192;
#define CLOB "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v16","v17"
#define X32(r) "v_xor_b32_e32 v" #r ", v16, v" #r "\n\t"
#define X64(r) "v_xor_b32_e64 v" #r ", v16, v" #r "\n\t"
#define M32(r) "v_mul_u32_u24_e32 v" #r ", v16, v" #r "\n\t"
#define M64(r) "v_mul_u32_u24_e64 v" #r ", v16, v" #r "\n\t"
#define ROW(OP) OP(32) OP(33) OP(34) OP(35) OP(36) OP(37) OP(38) OP(39) OP(40) OP(41) OP(42) OP(43) OP(44) OP(45) OP(46) OP(47)
#define R4(x) x x x x
template <int V>
__global__ void k(uint64_t* out) {
  asm volatile("v_mov_b32 v16, 3\n\tv_mov_b32 v17, 5" ::: CLOB);
  for (int it = 0; it < NITER; ++it) {
    if (V == 0) asm volatile(R4(ROW(X32)) ::: CLOB);
    if (V == 1) asm volatile(R4(ROW(X64)) ::: CLOB);
    if (V == 2) asm volatile(R4(ROW(M32)) ::: CLOB);
    if (V == 3) asm volatile(R4(ROW(M64)) ::: CLOB);
  }
  uint32_t r; asm volatile("v_mov_b32 %0, v32" : "=v"(r));
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
typedef void (*kern_t)(uint64_t*);
int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount; double clk = prop.clockRate * 1e3;
  uint64_t* out; CK(hipMalloc(&out, 8 * 256 * 4096));
  struct B { const char* name; kern_t k; } bs[] = {{"v_xor_b32 e32 (4 B)", k<0>}, {"v_xor_b32 e64 (8 B)", k<1>},
                                                   {"v_mul_u32_u24 e32 (4 B)", k<2>}, {"v_mul_u32_u24 e64 (8 B)", k<3>}};
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int wps : {1, 2, 4}) {
    printf("--- %d wave(s)/SIMD ---\n", wps);
    for (auto& b : bs) {
      float best = 1e9;
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(b.k, dim3(cus * wps), dim3(256), 0, 0, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      printf("%-28s %8.3f ms  %.2f cycles/instr/SIMD\n", b.name, best, best * 1e-3 * clk / ((double)NITER * 64 * wps));
    }
  }
  return 0;
}

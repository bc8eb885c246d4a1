// Dependent-issue latencies of the Montgomery row's instructions for a lone wave (tools/, diagnostics only).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int NITER = 8192;
#define CLOB "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47", \
  "v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63", \
  "v16","v17","v18","v19","v20","v21","v22","v23","s20","s21","s22","s23","vcc"
#define R2(x) x x
#define R4(x) R2(x) R2(x)
#define R8(x) R4(x) R4(x)
#define R16(x) R8(x) R8(x)
#define MACD "v_mad_u64_u32 v[32:33], s[20:21], v18, v19, v[32:33]\n\t"            /* dependent on itself */
#define MACQ "v_mad_u64_u32 v[32:33], s[20:21], v18, v19, v[32:33]\n\t"
#define DPPQ "v_and_b32_dpp v18, v32, v16 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define SHR "v_lshrrev_b64 v[34:35], 29, v[32:33]\n\t"
#define ADD "v_lshl_add_u64 v[32:33], v[34:35], 0, v[36:37]\n\t"
#define ADDI "v_lshl_add_u64 v[38:39], v[38:39], 0, v[36:37]\n\t"
#define FILL(n) "v_mad_u64_u32 v[" #n "], s[20:21], v20, v21, v[" #n "]\n\t"
#define F4 FILL(40:41) FILL(42:43) FILL(44:45) FILL(46:47)
#define F8 F4 FILL(48:49) FILL(50:51) FILL(52:53) FILL(54:55)
#define F12 F8 FILL(56:57) FILL(58:59) FILL(60:61) FILL(62:63)

template <int V>
__global__ void k(uint64_t* out) {
  asm volatile("v_mov_b32 v16, 0x1fffffff\n\tv_mov_b32 v17, 5\n\tv_mov_b32 v18, 7\n\tv_mov_b32 v19, 9\n\t"
               "v_mov_b32 v20, 11\n\tv_mov_b32 v21, 13\n\tv_mov_b32 v36, 15\n\tv_mov_b32 v37, 0" ::: CLOB);
  for (int it = 0; it < NITER; ++it) {
    if (V == 0) asm volatile(R16(MACD) ::: CLOB);                       // MAC -> MAC through the accumulator
    if (V == 1) asm volatile(R8(DPPQ MACQ) ::: CLOB);                   // dpp -> MAC(src0) -> dpp(acc.lo)
    if (V == 2) asm volatile(R8(SHR ADD) ::: CLOB);                     // shift -> add -> shift
    if (V == 3) asm volatile(R4(DPPQ MACQ SHR ADD) ::: CLOB);           // the whole chain, nothing between
    if (V == 4) asm volatile(R4(DPPQ F4 MACQ F4 SHR ADD F4) ::: CLOB);  // chain + 12 fillers, add right after shift
    if (V == 5) asm volatile(R4(DPPQ F4 MACQ F4 SHR F4 ADD) ::: CLOB);  // chain + 12 fillers, spaced
    if (V == 6) asm volatile(R4(F12 F4) ::: CLOB);                      // 16 fillers only
    if (V == 7) asm volatile(R4(DPPQ MACQ SHR ADD F12) ::: CLOB);       // chain back to back, then 12 fillers
    if (V == 8) asm volatile(R4(DPPQ ADDI F4 MACQ F4 SHR F4 ADD) ::: CLOB);   // spaced + the independent recv add
    if (V == 9) asm volatile(R16(ADDI) ::: CLOB);                       // v_lshl_add_u64 dependent chain
    if (V == 10) asm volatile(R8(DPPQ "v_and_b32_dpp v17, v32, v16 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t") ::: CLOB);
  }
  uint32_t r; asm volatile("v_mov_b32 %0, v32" : "=v"(r));
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
typedef void (*kern_t)(uint64_t*);
int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount; double clk = prop.clockRate * 1e3;
  uint64_t* out; CK(hipMalloc(&out, 8 * 256 * 4096));
  struct B { const char* name; kern_t k; double per; } bs[] = {
      {"MAC->MAC (acc) chain, per MAC", k<0>, 16}, {"dpp->MAC->dpp chain, per pair", k<1>, 8},
      {"shr->add chain, per pair", k<2>, 8}, {"dpp,MAC,shr,add chain, per group", k<3>, 4},
      {"chain+12 fillers (add after shr), per group", k<4>, 4}, {"chain+12 fillers spaced, per group", k<5>, 4},
      {"16 fillers, per group", k<6>, 4}, {"chain b2b then 12 fillers, per group", k<7>, 4},
      {"spaced + recv add, per group", k<8>, 4}, {"lshl_add_u64 chain, per instr", k<9>, 16},
      {"2 dpp (indep), per pair", k<10>, 8}};
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
      printf("%-46s %8.3f ms  %6.1f cycles\n", b.name, best, best * 1e-3 * clk / ((double)NITER * b.per * wps));
    }
  }
  return 0;
}

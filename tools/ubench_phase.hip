// Does the 4-byte PHASE of a stream of 8-byte instructions matter for a lone wavefront?  (MI355X_MICROARCH.md: a
// hand-written stream shifted by 4 mod 8 bytes lost 13 %.)  The inner loop of the decrypt kernel is almost purely
// 8-byte VOP3 instructions, but every v_and_b32_e32 / v_lshlrev_b32_e32 / s_nop in it flips the phase of what follows.
// Loop of 64 v_mad_u64_u32 whose first instruction sits at 0 or 4 mod 8 (64-byte aligned label, optional 4-byte pad),
// 1 and 2 wavefronts per SIMD; plus the kernel's mix with its 4-byte instructions at even / odd positions.
// (tools/, diagnostics only)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int NITER = 4096;
#define CLOB "v16","v17","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","vcc","scc","s20","s21"
#define MAC(r) "v_mad_u64_u32 v[" #r ":" #r "+1], s[20:21], v16, v17, v[" #r ":" #r "+1]\n\t"
#define M16 MAC(32) MAC(34) MAC(36) MAC(38) MAC(40) MAC(42) MAC(44) MAC(46) MAC(48) MAC(50) MAC(52) MAC(54) MAC(56) MAC(58) MAC(60) MAC(62)
#define M64 M16 M16 M16 M16
#define AND32 "v_and_b32_e32 v16, v16, v17\n\t"   /* 4-byte encoding */
#define AND64 "v_and_b32_e64 v16, v16, v17\n\t"   /* 8-byte encoding */
#define TAIL "s_sub_u32 %0, %0, 1\n\ts_cmp_lg_u32 %0, 0\n\ts_cbranch_scc1 1b\n\t"
template <int V>
__global__ void k(uint64_t* out) {
  asm volatile("v_mov_b32 v16, 3\n\tv_mov_b32 v17, 5" ::: CLOB);
  int cnt = NITER;
  if (V == 0) asm volatile(".p2align 6\n\t1:\n\t" M64 TAIL : "+s"(cnt) :: CLOB);                       // phase 0
  if (V == 1) asm volatile(".p2align 6\n\ts_nop 0\n\t1:\n\t" M64 TAIL : "+s"(cnt) :: CLOB);            // phase 4
  // 64 MACs + 4 cheap ops: 4-byte ones in pairs (phase restored at once) ...
  if (V == 2) asm volatile(".p2align 6\n\t1:\n\t" M16 AND32 AND32 M16 M16 AND32 AND32 M16 TAIL "s_nop 0\n\t" : "+s"(cnt) :: CLOB);
  // ... single 4-byte ones: the 16 MACs behind each odd one are out of phase
  if (V == 3) asm volatile(".p2align 6\n\t1:\n\t" M16 AND32 M16 AND32 M16 AND32 M16 AND32 TAIL "s_nop 0\n\t" : "+s"(cnt) :: CLOB);
  // ... the same ops in the 8-byte encoding: never out of phase
  if (V == 4) asm volatile(".p2align 6\n\t1:\n\t" M16 AND64 M16 AND64 M16 AND64 M16 AND64 TAIL "s_nop 0\n\t" : "+s"(cnt) :: CLOB);
  uint32_t r; asm volatile("v_mov_b32 %0, v32" : "=v"(r));
  out[blockIdx.x * blockDim.x + threadIdx.x] = r + cnt;
}
typedef void (*kern_t)(uint64_t*);
int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount; double clk = prop.clockRate * 1e3;
  uint64_t* out; CK(hipMalloc(&out, 8 * 256 * 4096));
  struct B { const char* name; kern_t k; int instr; } bs[] = {
      {"64 MACs, loop at 0 mod 8", k<0>, 64}, {"64 MACs, loop at 4 mod 8", k<1>, 64},
      {"64 MACs + 4 v_and e32 in pairs", k<2>, 68}, {"64 MACs + 4 v_and e32 single", k<3>, 68},
      {"64 MACs + 4 v_and e64", k<4>, 68}};
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int wps : {1, 2}) {
    printf("--- %d wave(s)/SIMD ---\n", wps);
    for (auto& b : bs) {
      float best = 1e9;
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(b.k, dim3(cus * wps), dim3(256), 0, 0, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      printf("%-36s %8.3f ms  %.3f cycles/VALU instr/SIMD\n", b.name, best, best * 1e-3 * clk / ((double)NITER * b.instr * wps));
    }
  }
  return 0;
}

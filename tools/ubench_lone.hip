// Lone-wave issue cadence of the Montgomery inner loop's instruction mix (tools/, diagnostics only).
// Question behind it: the bench's CRT-decrypt launch is 1024 wavefronts on 1024 SIMDs, and
// a wavefront that is alone on its SIMD issues a v_mad_u64_u32 only every ~5.1 cycles (4.1-4.4 with >= 2 waves).
// Can anything a single wave does -- the carry-out register, scalar/LDS instructions in the gaps, priorities,
// an SALU-only helper wave on the same SIMD -- buy the missing cycle back?
// Every variant runs NITER iterations of a 64-instruction hand-written block; time = HIP events, best of 4.
// Run under `rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU`
// to see how the counters book a lone wave's issue bubbles (kernel names carry the variant number).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int NITER = 4096;
#define CLOB "v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31", \
  "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47", \
  "v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63", \
  "s20","s21","s22","s23","s24","s25","s26","s27","s30","s31","vcc","memory"
#define R2(x) x x
#define R4(x) R2(x) R2(x)
#define R8(x) R4(x) R4(x)
#define R16(x) R8(x) R8(x)
// sixteen independent accumulators v[32:63]; multiplicands v18..v21
#define MAC_(acc, sd, a, b) "v_mad_u64_u32 v[" #acc "], " sd ", " a ", " b ", v[" #acc "]\n\t"
#define MAC16(sd) MAC_(32:33, sd, "v18", "v19") MAC_(34:35, sd, "v20", "v21") MAC_(36:37, sd, "v18", "v21") MAC_(38:39, sd, "v20", "v19") \
                  MAC_(40:41, sd, "v18", "v19") MAC_(42:43, sd, "v20", "v21") MAC_(44:45, sd, "v18", "v21") MAC_(46:47, sd, "v20", "v19") \
                  MAC_(48:49, sd, "v18", "v19") MAC_(50:51, sd, "v20", "v21") MAC_(52:53, sd, "v18", "v21") MAC_(54:55, sd, "v20", "v19") \
                  MAC_(56:57, sd, "v18", "v19") MAC_(58:59, sd, "v20", "v21") MAC_(60:61, sd, "v18", "v21") MAC_(62:63, sd, "v20", "v19")
// the same sixteen with the carry-out rotating over four SGPR pairs
#define MAC16ROT MAC_(32:33, "s[20:21]", "v18", "v19") MAC_(34:35, "s[22:23]", "v20", "v21") MAC_(36:37, "s[24:25]", "v18", "v21") MAC_(38:39, "s[26:27]", "v20", "v19") \
                 MAC_(40:41, "s[20:21]", "v18", "v19") MAC_(42:43, "s[22:23]", "v20", "v21") MAC_(44:45, "s[24:25]", "v18", "v21") MAC_(46:47, "s[26:27]", "v20", "v19") \
                 MAC_(48:49, "s[20:21]", "v18", "v19") MAC_(50:51, "s[22:23]", "v20", "v21") MAC_(52:53, "s[24:25]", "v18", "v21") MAC_(54:55, "s[26:27]", "v20", "v19") \
                 MAC_(56:57, "s[20:21]", "v18", "v19") MAC_(58:59, "s[22:23]", "v20", "v21") MAC_(60:61, "s[24:25]", "v18", "v21") MAC_(62:63, "s[26:27]", "v20", "v19")
// sixteen MACs, each followed by X
#define MACX(acc, a, b, X) MAC_(acc, "s[20:21]", a, b) X
#define MAC16X(X) MACX(32:33, "v18", "v19", X) MACX(34:35, "v20", "v21", X) MACX(36:37, "v18", "v21", X) MACX(38:39, "v20", "v19", X) \
                  MACX(40:41, "v18", "v19", X) MACX(42:43, "v20", "v21", X) MACX(44:45, "v18", "v21", X) MACX(46:47, "v20", "v19", X) \
                  MACX(48:49, "v18", "v19", X) MACX(50:51, "v20", "v21", X) MACX(52:53, "v18", "v21", X) MACX(54:55, "v20", "v19", X) \
                  MACX(56:57, "v18", "v19", X) MACX(58:59, "v20", "v21", X) MACX(60:61, "v18", "v21", X) MACX(62:63, "v20", "v19", X)
// SGPR multiplicand (one VGPR read less)
#define MAC16S MAC_(32:33, "s[20:21]", "s30", "v19") MAC_(34:35, "s[20:21]", "s30", "v21") MAC_(36:37, "s[20:21]", "s30", "v21") MAC_(38:39, "s[20:21]", "s30", "v19") \
               MAC_(40:41, "s[20:21]", "s30", "v19") MAC_(42:43, "s[20:21]", "s30", "v21") MAC_(44:45, "s[20:21]", "s30", "v21") MAC_(46:47, "s[20:21]", "s30", "v19") \
               MAC_(48:49, "s[20:21]", "s30", "v19") MAC_(50:51, "s[20:21]", "s30", "v21") MAC_(52:53, "s[20:21]", "s30", "v21") MAC_(54:55, "s[20:21]", "s30", "v19") \
               MAC_(56:57, "s[20:21]", "s30", "v19") MAC_(58:59, "s[20:21]", "s30", "v21") MAC_(60:61, "s[20:21]", "s30", "v21") MAC_(62:63, "s[20:21]", "s30", "v19")
// the row's support instructions as they occur in mont_block (independent copies)
#define SUP5 "v_and_b32_dpp v22, v32, v16 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
             "v_and_b32_dpp v23, v34, v16 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
             "v_lshrrev_b64 v[24:25], 29, v[36:37]\n\t" \
             "v_lshl_add_u64 v[38:39], v[38:39], 0, v[24:25]\n\t" \
             "v_lshl_add_u64 v[40:41], v[40:41], 0, v[26:27]\n\t"

__device__ __forceinline__ void init_regs() {
  asm volatile("v_mov_b32 v16, 0x1fffffff\n\tv_mov_b32 v17, 5\n\tv_mov_b32 v18, 7\n\tv_mov_b32 v19, 9\n\t"
               "v_mov_b32 v20, 11\n\tv_mov_b32 v21, 13\n\tv_mov_b32 v26, 15\n\tv_mov_b32 v27, 0\n\t"
               "s_mov_b32 s30, 77\n\tv_mov_b32 v28, 0" ::: CLOB);
}

// V: variant of the 64-instruction block.  HELPER: waves 4..7 of an 8-wave workgroup do not compute, they run
// the helper loop (1: s_nop, 2: s_sleep, 3: SALU adds) until the workers are done (LDS flag).
template <int V, int HELPER>
__global__ __launch_bounds__(512) void k(uint64_t* out) {
  __shared__ volatile int done;
  if (threadIdx.x == 0) done = 0;
  __syncthreads();
  if (HELPER && threadIdx.x >= 256) {
    while (done < 4) {
      if (HELPER == 1) asm volatile(R16("s_nop 0\n\t") ::: "memory");
      if (HELPER == 2) asm volatile("s_sleep 2" ::: "memory");
      if (HELPER == 3) asm volatile(R16("s_add_u32 s30, s30, 1\n\t") ::: "s30", "scc", "memory");
    }
    return;
  }
  init_regs();
  if (V == 9 || V == 12 || V == 13) asm volatile("s_setprio 3");
  for (int it = 0; it < NITER; ++it) {
    if (V == 0) asm volatile(R4(MAC16("s[20:21]")) ::: CLOB);                    // baseline: 64 MACs
    if (V == 1) asm volatile(R4(MAC16ROT) ::: CLOB);                              // carry-out rotates over 4 SGPR pairs
    if (V == 2) asm volatile(R4(MAC16("vcc")) ::: CLOB);                          // carry-out to vcc
    if (V == 3) asm volatile(R4(MAC16X("s_nop 0\n\t")) ::: CLOB);                 // + one s_nop per MAC   (64 MAC + 64 nop)
    if (V == 4) asm volatile(R4(MAC16X("s_add_u32 s31, s31, 1\n\t")) ::: CLOB, "scc");   // + one SALU op per MAC
    if (V == 5) asm volatile(R4(MAC16X("v_add_u32 v28, v28, v17\n\t")) ::: CLOB); // + one cheap VALU per MAC (64 + 64)
    if (V == 6) asm volatile(R4(MAC16S) ::: CLOB);                                // SGPR multiplicand
    if (V == 7) asm volatile(R4(MAC16("s[20:21]") ) "ds_read_b32 v29, v28\n\tds_read_b32 v30, v28 offset:64\n\t"
                             "ds_read_b32 v31, v28 offset:128\n\tds_read_b32 v17, v28 offset:192\n\ts_waitcnt lgkmcnt(0)\n\t" ::: CLOB);
                                                                                  // 64 MACs + 4 LDS reads waited for at the end
    if (V == 8) asm volatile(R2(MAC16("s[20:21]") MAC16("s[20:21]") SUP5 SUP5) ::: CLOB);   // the kernel's mix: 64 MAC + 20 support
    if (V == 9) asm volatile(R4(MAC16("s[20:21]")) ::: CLOB);                    // baseline at s_setprio 3
    if (V == 10) asm volatile(R4("ds_read_b32 v29, v28\n\t" MAC16("s[20:21]") "s_waitcnt lgkmcnt(0)\n\t") ::: CLOB);
    if (V == 11) asm volatile(R2(MAC16("vcc") MAC16("vcc") SUP5 SUP5) ::: CLOB);            // the kernel's mix, carry-out to vcc
    if (V == 12) asm volatile(R2(MAC16("s[20:21]") MAC16("s[20:21]") SUP5 SUP5) ::: CLOB);  // the kernel's mix at s_setprio 3
    if (V == 13) asm volatile(R2(MAC16("vcc") MAC16("vcc") SUP5 SUP5) ::: CLOB);            // mix, vcc, s_setprio 3
    if (V == 14) asm volatile(R2(MAC16ROT MAC16ROT SUP5 SUP5) ::: CLOB);                    // mix, rotating carry-out
                                                                                  // LDS read issued 16 MACs before its wait
  }
  uint32_t r; asm volatile("v_add_u32 %0, v32, v28" : "=v"(r));
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (HELPER && (threadIdx.x & 63) == 0) atomicAdd((int*)&done, 1);
}

typedef void (*kern_t)(uint64_t*);
int main(int argc, char** argv) {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount; double clk = prop.clockRate * 1e3;
  uint64_t* out; CK(hipMalloc(&out, 8 * 512 * 4096));
  struct B { const char* name; kern_t k; double instr; double macs; int threads; } bs[] = {
      {"64 MACs (carry-out s[20:21])", k<0, 0>, 64, 64, 256},
      {"64 MACs, carry-out rotating over 4 SGPR pairs", k<1, 0>, 64, 64, 256},
      {"64 MACs, carry-out to vcc", k<2, 0>, 64, 64, 256},
      {"64 x (MAC + s_nop 0)", k<3, 0>, 128, 64, 256},
      {"64 x (MAC + s_add_u32)", k<4, 0>, 128, 64, 256},
      {"64 x (MAC + v_add_u32 e32)", k<5, 0>, 128, 64, 256},
      {"64 MACs, SGPR multiplicand", k<6, 0>, 64, 64, 256},
      {"64 MACs + 4 ds_read_b32 + wait", k<7, 0>, 69, 64, 256},
      {"kernel mix: 64 MACs + 20 support ops", k<8, 0>, 84, 64, 256},
      {"64 MACs at s_setprio 3", k<9, 0>, 64, 64, 256},
      {"4 x (ds_read, 16 MACs, wait)", k<10, 0>, 72, 64, 256},
      {"kernel mix, carry-out to vcc", k<11, 0>, 84, 64, 256},
      {"kernel mix at s_setprio 3", k<12, 0>, 84, 64, 256},
      {"kernel mix, vcc, s_setprio 3", k<13, 0>, 84, 64, 256},
      {"kernel mix, rotating carry-out", k<14, 0>, 84, 64, 256},
      {"64 MACs + s_nop helper wave on the SIMD", k<0, 1>, 64, 64, 512},
      {"64 MACs + s_sleep helper wave on the SIMD", k<0, 2>, 64, 64, 512},
      {"64 MACs + SALU helper wave on the SIMD", k<0, 3>, 64, 64, 512},
      {"kernel mix + SALU helper wave", k<8, 3>, 84, 64, 512}};
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int only = argc > 1 ? atoi(argv[1]) : -1;
  for (int wps : {1, 2}) {
    printf("--- %d computing wave(s)/SIMD ---\n", wps);
    int idx = 0;
    for (auto& b : bs) {
      if (only >= 0 && idx++ != only) continue;
      if (b.threads == 512 && wps == 2) continue;
      float best = 1e9;
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(b.k, dim3(cus * wps), dim3(b.threads), 0, 0, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      double cyc = best * 1e-3 * clk / ((double)NITER * wps);
      printf("%-50s %8.3f ms  %7.1f cycles/block  %5.2f per instr  %5.2f per MAC\n", b.name, best, cyc, cyc / b.instr,
             cyc / b.macs);
    }
  }
  return 0;
}

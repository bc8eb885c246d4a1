// Micro-benchmark: issue rate of candidate big-integer multiply primitives on gfx950.
// Each kernel runs NITER iterations of UNROLL independent chains of one instruction per lane.
// Reports wave-instructions/s, lane-ops/s and cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int NITER = 4096;

// 8 independent chains per lane; the asm volatile keeps hipcc from folding them.
#define CHAIN8(STMT) STMT(0) STMT(1) STMT(2) STMT(3) STMT(4) STMT(5) STMT(6) STMT(7)

__global__ void k_mad_u64_u32(uint64_t* out, uint32_t a, uint32_t b) {
  uint64_t acc[8]; uint32_t x = a + threadIdx.x, y = b ^ threadIdx.x;
  for (int i = 0; i < 8; ++i) acc[i] = i + threadIdx.x;
  for (int it = 0; it < NITER; ++it) {
#define S(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y) : "vcc");
    CHAIN8(S) CHAIN8(S)
#undef S
  }
  uint64_t s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mul_lo_u32(uint64_t* out, uint32_t a, uint32_t b) {
  uint32_t acc[8]; uint32_t y = b ^ threadIdx.x | 1;
  for (int i = 0; i < 8; ++i) acc[i] = a + i + threadIdx.x;
  for (int it = 0; it < NITER; ++it) {
#define S(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(acc[i]) : "v"(y));
    CHAIN8(S) CHAIN8(S)
#undef S
  }
  uint64_t s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mul_hi_u32(uint64_t* out, uint32_t a, uint32_t b) {
  uint32_t acc[8]; uint32_t y = b ^ threadIdx.x | 0x80000001u;
  for (int i = 0; i < 8; ++i) acc[i] = a + i + threadIdx.x;
  for (int it = 0; it < NITER; ++it) {
#define S(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(acc[i]) : "v"(y));
    CHAIN8(S) CHAIN8(S)
#undef S
  }
  uint64_t s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mad_u32_u24(uint64_t* out, uint32_t a, uint32_t b) {
  uint32_t acc[8]; uint32_t x = a + threadIdx.x, y = b ^ threadIdx.x;
  for (int i = 0; i < 8; ++i) acc[i] = a + i + threadIdx.x;
  for (int it = 0; it < NITER; ++it) {
#define S(i) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y));
    CHAIN8(S) CHAIN8(S)
#undef S
  }
  uint64_t s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mul_hi_u32_u24(uint64_t* out, uint32_t a, uint32_t b) {
  uint32_t acc[8]; uint32_t y = b ^ threadIdx.x | 0x800001u;
  for (int i = 0; i < 8; ++i) acc[i] = a + i + threadIdx.x;
  for (int it = 0; it < NITER; ++it) {
#define S(i) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(acc[i]) : "v"(y));
    CHAIN8(S) CHAIN8(S)
#undef S
  }
  uint64_t s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_fma_f64(uint64_t* out, uint32_t a, uint32_t b) {
  double acc[8]; double x = 1.0 + 1e-9 * (a + threadIdx.x), y = 1e-12 * (b + threadIdx.x);
  for (int i = 0; i < 8; ++i) acc[i] = i + threadIdx.x;
  for (int it = 0; it < NITER; ++it) {
#define S(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(x), "v"(y));
    CHAIN8(S) CHAIN8(S)
#undef S
  }
  double s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint64_t)s;
}
__global__ void k_add_f64(uint64_t* out, uint32_t a, uint32_t b) {
  double acc[8]; double y = 1e-12 * (b + threadIdx.x);
  for (int i = 0; i < 8; ++i) acc[i] = i + threadIdx.x + a;
  for (int it = 0; it < NITER; ++it) {
#define S(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(acc[i]) : "v"(y));
    CHAIN8(S) CHAIN8(S)
#undef S
  }
  double s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint64_t)s;
}
__global__ void k_mul_f64(uint64_t* out, uint32_t a, uint32_t b) {
  double acc[8]; double y = 1.0 + 1e-12 * (b + threadIdx.x);
  for (int i = 0; i < 8; ++i) acc[i] = i + threadIdx.x + a;
  for (int it = 0; it < NITER; ++it) {
#define S(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(acc[i]) : "v"(y));
    CHAIN8(S) CHAIN8(S)
#undef S
  }
  double s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint64_t)s;
}
__global__ void k_fma_f32(uint64_t* out, uint32_t a, uint32_t b) {
  float acc[8]; float x = 1.0f + 1e-6f * (a + threadIdx.x), y = 1e-6f * (b + threadIdx.x);
  for (int i = 0; i < 8; ++i) acc[i] = i + threadIdx.x;
  for (int it = 0; it < NITER; ++it) {
#define S(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(x), "v"(y));
    CHAIN8(S) CHAIN8(S)
#undef S
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint64_t)s;
}
__global__ void k_add_u32(uint64_t* out, uint32_t a, uint32_t b) {
  uint32_t acc[8]; uint32_t y = b ^ threadIdx.x;
  for (int i = 0; i < 8; ++i) acc[i] = a + i + threadIdx.x;
  for (int it = 0; it < NITER; ++it) {
#define S(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(acc[i]) : "v"(y));
    CHAIN8(S) CHAIN8(S)
#undef S
  }
  uint64_t s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_add_co_pair(uint64_t* out, uint32_t a, uint32_t b) {
  uint32_t lo[8], hi[8]; uint32_t y = b ^ threadIdx.x, z = a;
  for (int i = 0; i < 8; ++i) { lo[i] = a + i + threadIdx.x; hi[i] = i; }
  for (int it = 0; it < NITER; ++it) {
#define S(i) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(lo[i]), "+v"(hi[i]) : "v"(y), "v"(z) : "vcc");
    CHAIN8(S) CHAIN8(S)
#undef S
  }
  uint64_t s = 0; for (int i = 0; i < 8; ++i) s += lo[i] + ((uint64_t)hi[i] << 32);
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_lshl_add_u64(uint64_t* out, uint32_t a, uint32_t b) {
  uint64_t acc[8]; uint64_t y = ((uint64_t)b << 20) ^ threadIdx.x;
  for (int i = 0; i < 8; ++i) acc[i] = a + i + threadIdx.x;
  for (int it = 0; it < NITER; ++it) {
#define S(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"(y));
    CHAIN8(S) CHAIN8(S)
#undef S
  }
  uint64_t s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_cvt_f64_u32(uint64_t* out, uint32_t a, uint32_t b) {
  uint32_t src[8]; double acc[8];
  for (int i = 0; i < 8; ++i) { src[i] = a + i + threadIdx.x; acc[i] = 0; }
  for (int it = 0; it < NITER; ++it) {
#define S(i) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(acc[i]) : "v"(src[i]));
    CHAIN8(S) CHAIN8(S)
#undef S
  }
  double s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint64_t)s;
}
__global__ void k_bpermute(uint64_t* out, uint32_t a, uint32_t b) {
  uint32_t acc[8]; uint32_t addr = ((threadIdx.x + 1 + a) & 63) * 4;
  for (int i = 0; i < 8; ++i) acc[i] = b + i + threadIdx.x;
  for (int it = 0; it < NITER / 4; ++it) {
#define S(i) asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(acc[i]) : "v"(addr));
    CHAIN8(S) CHAIN8(S)
#undef S
  }
  uint64_t s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_bpermute_pipelined(uint64_t* out, uint32_t a, uint32_t b) {
  uint32_t acc[8]; uint32_t addr = ((threadIdx.x + 1 + a) & 63) * 4;
  for (int i = 0; i < 8; ++i) acc[i] = b + i + threadIdx.x;
  for (int it = 0; it < NITER / 4; ++it) {
#define S(i) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(acc[i]) : "v"(addr));
    CHAIN8(S)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    CHAIN8(S)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#undef S
  }
  uint64_t s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_dpp_mov(uint64_t* out, uint32_t a, uint32_t b) {
  uint32_t acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = b + i + threadIdx.x + a;
  for (int it = 0; it < NITER; ++it) {
#define S(i) asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(acc[i]));
    CHAIN8(S) CHAIN8(S)
#undef S
  }
  uint64_t s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_readlane(uint64_t* out, uint32_t a, uint32_t b) {
  uint32_t acc[8]; uint32_t x = b ^ threadIdx.x;
  for (int i = 0; i < 8; ++i) acc[i] = a + i + threadIdx.x;
  for (int it = 0; it < NITER; ++it) {
#define S(i) { uint32_t s_; asm volatile("v_readlane_b32 %0, %1, 5\n\ts_nop 3\n\tv_add_u32 %1, %1, %0" : "=&s"(s_), "+v"(acc[i])); }
    CHAIN8(S) CHAIN8(S)
#undef S
  }
  uint64_t s = 0; for (int i = 0; i < 8; ++i) s += acc[i] + x;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// latency probes: one dependent chain, 1 wave per SIMD
__global__ void k_lat_fma_f64(uint64_t* out, uint32_t a, uint32_t b) {
  double acc = threadIdx.x; double x = 1.0 + 1e-9 * a, y = 1e-12 * b;
  for (int it = 0; it < NITER; ++it) {
#define S(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc) : "v"(x), "v"(y));
    CHAIN8(S) CHAIN8(S)
#undef S
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint64_t)acc;
}
__global__ void k_lat_mad_u64_u32(uint64_t* out, uint32_t a, uint32_t b) {
  uint64_t acc = threadIdx.x; uint32_t x = a + threadIdx.x, y = b ^ threadIdx.x;
  for (int it = 0; it < NITER; ++it) {
#define S(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y) : "vcc");
    CHAIN8(S) CHAIN8(S)
#undef S
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

typedef void (*kern_t)(uint64_t*, uint32_t, uint32_t);
struct Bench { const char* name; kern_t k; double ops_per_iter; };

int main(int argc, char** argv) {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount; double clk = prop.clockRate * 1e3;
  printf("device=%s arch=%s CUs=%d clock=%.0f MHz\n", prop.name, prop.gcnArchName, cus, clk / 1e6);
  uint64_t* out; CK(hipMalloc(&out, sizeof(uint64_t) * 256 * 8 * 1024 * 4));
  std::vector<Bench> B = {
    {"v_mad_u64_u32", k_mad_u64_u32, 16.0 * NITER}, {"v_mul_lo_u32", k_mul_lo_u32, 16.0 * NITER},
    {"v_mul_hi_u32", k_mul_hi_u32, 16.0 * NITER}, {"v_mad_u32_u24", k_mad_u32_u24, 16.0 * NITER},
    {"v_mul_hi_u32_u24", k_mul_hi_u32_u24, 16.0 * NITER}, {"v_fma_f64", k_fma_f64, 16.0 * NITER},
    {"v_add_f64", k_add_f64, 16.0 * NITER}, {"v_mul_f64", k_mul_f64, 16.0 * NITER},
    {"v_fma_f32", k_fma_f32, 16.0 * NITER}, {"v_add_u32", k_add_u32, 16.0 * NITER},
    {"v_add_co+v_addc (pair)", k_add_co_pair, 16.0 * NITER}, {"v_lshl_add_u64", k_lshl_add_u64, 16.0 * NITER},
    {"v_cvt_f64_u32", k_cvt_f64_u32, 16.0 * NITER}, {"ds_bpermute_b32 (wait each)", k_bpermute, 16.0 * NITER / 4},
    {"ds_bpermute_b32 (8 in flight)", k_bpermute_pipelined, 16.0 * NITER / 4},
    {"v_mov_dpp row_shr (+s_nop1)", k_dpp_mov, 16.0 * NITER}, {"v_readlane+nop3+add", k_readlane, 16.0 * NITER},
  };
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  // saturating config: waves_per_simd in {1,2,4,8}
  for (int wps : {1, 2, 4, 8}) {
    int threads = 256, blocks = cus * wps;  // 4 waves per block -> wps blocks per CU
    printf("--- %d wave(s)/SIMD, grid=%d x %d ---\n", wps, blocks, threads);
    for (auto& b : B) {
      hipLaunchKernelGGL(b.k, dim3(blocks), dim3(threads), 0, 0, out, 3u, 5u);
      CK(hipDeviceSynchronize());
      float best = 1e30f;
      for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(b.k, dim3(blocks), dim3(threads), 0, 0, out, 3u, 5u);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      double waves = (double)blocks * threads / 64;
      double winst = waves * b.ops_per_iter;             // wave-instructions
      double t = best * 1e-3;
      double per_simd_cyc = t * clk / (winst / (cus * 4.0));  // cycles per wave-instr per SIMD
      printf("%-32s %8.3f ms  %8.2f T lane-ops/s  %6.2f cyc/wave-instr/SIMD\n", b.name, best, winst * 64 / t / 1e12, per_simd_cyc);
    }
  }
  printf("--- latency probes (1 wave/SIMD, dependent chain) ---\n");
  for (auto& b : std::vector<Bench>{{"lat v_fma_f64", k_lat_fma_f64, 16.0 * NITER}, {"lat v_mad_u64_u32", k_lat_mad_u64_u32, 16.0 * NITER}}) {
    int threads = 256, blocks = cus;
    hipLaunchKernelGGL(b.k, dim3(blocks), dim3(threads), 0, 0, out, 3u, 5u); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(b.k, dim3(blocks), dim3(threads), 0, 0, out, 3u, 5u);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-32s %8.3f ms  %6.2f cyc per dependent instr\n", b.name, ms, ms * 1e-3 * clk / b.ops_per_iter);
  }
  return 0;
}

// Variant of redc_mfma.hip with the shared matrices held in LDS as their distinct Toeplitz diagonals
// (tile(t, s) depends only on 16 t - 64 s), four wavefronts (64 elements) per workgroup, fully unrolled GEMMs.
// Experiment, not product code.  Sized for D <= 320 bytes (2048-bit class).
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int STEPS = 5, TILES = 17, NDIAG = TILES + 4 * (STEPS - 1);   // 17 x 16 = 272 output rows, K = 320

struct RedcLdsArgs {
  const uint8_t* T;      // [elems][2*D]
  uint8_t* U;            // [elems][D + 8]
  const int8_t* diag1;   // [NDIAG][64 lanes][16]: tile with 16 t - 64 s = 16 (j - 4 (STEPS-1)), Toeplitz_low(N')
  const int8_t* diag2;   // same for Toeplitz(N) shifted to start at row c0 (a multiple of 16)
  const int* corr1;      // [TILES*16]
  const int* corr2;
  int D, elems, c0;
  int reps;
  long long* clocks;     // [workgroups*4][5]
};

__device__ __forceinline__ v4i offset128(v4i v) {
  v.x ^= 0x80808080; v.y ^= 0x80808080; v.z ^= 0x80808080; v.w ^= 0x80808080;
  return v;
}

extern "C" __global__ __launch_bounds__(256) void redc_lds_kernel(RedcLdsArgs A) {
  __shared__ v4i tiles1[NDIAG][64];
  __shared__ v4i tiles2[NDIAG][64];
  __shared__ int sums[4][16][TILES * 16];
  __shared__ uint8_t qb[4][16][STEPS * 64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, n = lane & 15, rho = lane >> 4;
  for (int i = threadIdx.x; i < NDIAG * 64; i += 256) {
    tiles1[i / 64][i % 64] = ((const v4i*)A.diag1)[i];
    tiles2[i / 64][i % 64] = ((const v4i*)A.diag2)[i];
  }
  __syncthreads();
  const int wave = blockIdx.x * 4 + wv;
  int elem = wave * 16 + n;
  if (elem >= A.elems) elem = A.elems - 1;
  const int D = A.D;
  const uint8_t* t = A.T + (size_t)elem * 2 * D;
  long long c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0;
  for (int rep = 0; rep < A.reps; ++rep) {
    c0 = __builtin_readcyclecounter();
    v4i B[STEPS];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) B[s] = offset128(*(const v4i*)(t + 64 * s + 16 * rho));
#pragma unroll
    for (int tl = 0; tl < TILES; ++tl) {
      v4i c = *(const v4i*)(A.corr1 + tl * 16 + 4 * rho);
#pragma unroll
      for (int s = 0; s < STEPS; ++s) {
        if (64 * s > 16 * tl + 15) continue;                    // lower-triangular
        c = __builtin_amdgcn_mfma_i32_16x16x64_i8(tiles1[tl - 4 * s + 4 * (STEPS - 1)][lane], B[s], c, 0, 0, 0);
      }
      *(v4i*)(&sums[wv][n][tl * 16 + 4 * rho]) = c;
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    c1 = __builtin_readcyclecounter();
    if (rho == 0) {                                             // serial carry per element (not tuned)
      long long carry = 0;
      for (int i = 0; i < STEPS * 64; ++i) {
        long long v = (i < D ? (long long)sums[wv][n][i] : 0) + carry;
        qb[wv][n][i] = i < D ? (uint8_t)(v & 0xFF) : 0x80;      // padding = 128: zero after the offset
        carry = v >> 8;
      }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    c2 = __builtin_readcyclecounter();
#pragma unroll
    for (int s = 0; s < STEPS; ++s) B[s] = offset128(*(const v4i*)(&qb[wv][n][64 * s + 16 * rho]));
#pragma unroll
    for (int tl = 0; tl < TILES; ++tl) {
      v4i c = *(const v4i*)(A.corr2 + tl * 16 + 4 * rho);
#pragma unroll
      for (int s = 0; s < STEPS; ++s)
        c = __builtin_amdgcn_mfma_i32_16x16x64_i8(tiles2[tl - 4 * s + 4 * (STEPS - 1)][lane], B[s], c, 0, 0, 0);
      *(v4i*)(&sums[wv][n][tl * 16 + 4 * rho]) = c;
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    c3 = __builtin_readcyclecounter();
    if (rho == 0 && wave * 16 + n < A.elems) {
      const int top = D - A.c0;                                 // rows of the low half that GEMM 2 produced
      long long V = 0;
      for (int j = 1; j <= 5; ++j) V += ((long long)sums[wv][n][top - j] + t[D - j]) << (8 * (5 - j));
      long long carry = (V + (1ll << 39)) >> 40;
      uint8_t* u = A.U + (size_t)(wave * 16 + n) * (D + 8);
      for (int c = 0; c < D + 8; ++c) {
        long long v = carry + (c < D ? (long long)sums[wv][n][top + c] + t[D + c] : 0);   // column 2D of Q*N is empty
        u[c] = (uint8_t)(v & 0xFF);
        carry = v >> 8;
      }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    c4 = __builtin_readcyclecounter();
  }
  if (A.clocks && lane == 0) {
    long long* r = A.clocks + (size_t)wave * 5;
    r[0] = c1 - c0; r[1] = c2 - c1; r[2] = c3 - c2; r[3] = c4 - c3; r[4] = c4 - c0;
  }
}

extern "C" int redc_lds_launch(const RedcLdsArgs* a, int workgroups) {
  hipLaunchKernelGGL(redc_lds_kernel, dim3(workgroups), dim3(256), 0, 0, *a);
  return (int)hipDeviceSynchronize();
}

// Byte-radix Montgomery reduction on the matrix cores -- experiment, not product code.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o libredc_mfma.so redc_mfma.hip
// One wavefront reduces 16 elements (the N dimension of v_mfma_i32_16x16x64_i8).  See model.py for the maths.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef int v4i __attribute__((ext_vector_type(4)));

struct RedcArgs {
  const uint8_t* T;      // [elems][2*D] little-endian bytes
  uint8_t* U;            // [elems][D + 8]   (T + Q*N) / R
  int* dump;             // optional raw column sums of both GEMMs: [waves][2][tiles][64][4]
  const int8_t* A1;      // [tiles][steps][64 lanes][16] balanced digits of Toeplitz_low(N'), lane order
  const int8_t* A2;      // [tiles][steps][64][16] Toeplitz(N), rows c0 ..
  const int* corr1;      // [tiles*16]
  const int* corr2;      // [tiles*16]
  int D, tiles, steps, elems, top;
  int reps;              // repeat the reduction in-kernel (timing: tiles come from cache after the first pass)
  long long* clocks;     // optional [waves][5]: cycles of GEMM 1, carry 1, GEMM 2, final stage, total (last repetition)
};

constexpr int kMaxSteps = 9;     // K dimension up to 576 bytes

__device__ __forceinline__ v4i offset128(v4i v) {
  v.x ^= 0x80808080; v.y ^= 0x80808080; v.z ^= 0x80808080; v.w ^= 0x80808080;
  return v;
}

extern "C" __global__ __launch_bounds__(64) void redc_kernel(RedcArgs A) {
  extern __shared__ int lds[];                    // [16 elems][tiles*16] column sums, then bytes
  const int lane = threadIdx.x, n = lane & 15, rho = lane >> 4;
  const int wave = blockIdx.x;
  int elem = wave * 16 + n;
  if (elem >= A.elems) elem = A.elems - 1;
  const int D = A.D, cols = A.tiles * 16;
  const uint8_t* t = A.T + (size_t)elem * 2 * D;
  v4i B[kMaxSteps];
  long long c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0;
  for (int rep = 0; rep < A.reps; ++rep) {
  c0 = __builtin_readcyclecounter();
  // ---- GEMM 1: column sums of T_lo * N' ----
  for (int s = 0; s < A.steps; ++s) B[s] = offset128(*(const v4i*)(t + 64 * s + 16 * rho));
  for (int tl = 0; tl < A.tiles; ++tl) {
    v4i c = *(const v4i*)(A.corr1 + tl * 16 + 4 * rho);
    for (int s = 0; s < A.steps; ++s) {
      if (64 * s > 16 * tl + 15) break;            // lower-triangular: k <= i
      const v4i a = *(const v4i*)(A.A1 + (((size_t)tl * A.steps + s) * 64 + lane) * 16);
      c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, B[s], c, 0, 0, 0);
    }
    if (A.dump) *(v4i*)(A.dump + ((((size_t)wave * 2 + 0) * A.tiles + tl) * 64 + lane) * 4) = c;
    *(v4i*)(lds + n * cols + tl * 16 + 4 * rho) = c;     // rows 4*rho .. 4*rho+3 of this tile, element n
  }
  __syncthreads();
  c1 = __builtin_readcyclecounter();
  // ---- Q = column sums mod R as bytes (serial carry per element: lanes 0..15; an experiment, not tuned) ----
  uint8_t* qb = (uint8_t*)(lds + 16 * cols);      // [16][steps*64]
  const int kbytes = A.steps * 64;
  if (rho == 0) {
    long long carry = 0;
    for (int i = 0; i < kbytes; ++i) {
      long long v = (i < D ? (long long)lds[n * cols + i] : 0) + carry;
      qb[n * kbytes + i] = i < D ? (uint8_t)(v & 0xFF) : 0;
      carry = v >> 8;                              // arithmetic shift: sums may be negative
    }
  }
  __syncthreads();
  c2 = __builtin_readcyclecounter();
  // ---- GEMM 2: column sums c0 .. of Q * N ----
  for (int s = 0; s < A.steps; ++s) B[s] = offset128(*(const v4i*)(qb + n * kbytes + 64 * s + 16 * rho));
  for (int tl = 0; tl < A.tiles; ++tl) {
    v4i c = *(const v4i*)(A.corr2 + tl * 16 + 4 * rho);
    for (int s = 0; s < A.steps; ++s) {
      const v4i a = *(const v4i*)(A.A2 + (((size_t)tl * A.steps + s) * 64 + lane) * 16);
      c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, B[s], c, 0, 0, 0);
    }
    if (A.dump) *(v4i*)(A.dump + ((((size_t)wave * 2 + 1) * A.tiles + tl) * 64 + lane) * 4) = c;
    *(v4i*)(lds + n * cols + tl * 16 + 4 * rho) = c;
  }
  __syncthreads();
  c3 = __builtin_readcyclecounter();
  // ---- U = T_hi + high columns + carry of the low half (serial per element) ----
  if (rho == 0 && wave * 16 + n < A.elems) {
    const int top = A.top;
    long long V = 0;
    for (int j = 1; j <= top; ++j) V += ((long long)lds[n * cols + top - j] + t[D - j]) << (8 * (top - j));
    long long carry = (V + (1ll << (8 * top - 1))) >> (8 * top);
    uint8_t* u = A.U + (size_t)(wave * 16 + n) * (D + 8);
    for (int c = 0; c < D + 8; ++c) {
      long long v = carry + (c <= D ? (long long)lds[n * cols + top + c] : 0) + (c < D ? t[D + c] : 0);
      u[c] = (uint8_t)(v & 0xFF);
      carry = v >> 8;
    }
  }
  __syncthreads();
  c4 = __builtin_readcyclecounter();
  }
  if (A.clocks && lane == 0) {
    long long* r = A.clocks + (size_t)wave * 5;
    r[0] = c1 - c0; r[1] = c2 - c1; r[2] = c3 - c2; r[3] = c4 - c3; r[4] = c4 - c0;
  }
}

extern "C" int redc_launch(const RedcArgs* a, int waves, int lds_bytes) {
  hipLaunchKernelGGL(redc_kernel, dim3(waves), dim3(64), lds_bytes, 0, *a);
  return (int)hipDeviceSynchronize();
}

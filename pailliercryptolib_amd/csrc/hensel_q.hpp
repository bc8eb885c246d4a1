// pailliercryptolib_amd -- reduction rows with visible quotient digits, shared by the sequential-halves kernels
// (hensel_seq.hpp).
#ifndef PAILLIERCRYPTOLIB_AMD_CSRC_HENSEL_Q_HPP_
#define PAILLIERCRYPTOLIB_AMD_CSRC_HENSEL_Q_HPP_

#include "hensel.hpp"

namespace pgpu {

// The K reduction rows of one block (mont_core.hpp: mont_reduce_rows) with the quotient digits made visible:
//   QMODE 1: the digit of row r is recorded in qio[r] (wavefront A);
//   QMODE 2: qio[r] * sel0 is added to column r before the digit of row r is taken (wavefront B: A's digit enters the
//            group's low lane, sel0 = 1 there and 0 elsewhere).
template <class GEO, bool UNITQ, int QMODE>
__device__ __forceinline__ void mont_reduce_rows_q(uint64_t (&LOWC)[GEO::K], uint64_t (&UPC)[GEO::K],
                                                   const uint32_t (&n)[GEO::K], uint32_t n0inv, uint32_t (&qio)[GEO::K],
                                                   uint32_t sel0) {
  constexpr int K = GEO::K;
  uint32_t maskv = kLimbMask;
  asm("" : "+v"(maskv));
  constexpr int kNoValuCross = 0x3fc;
  constexpr bool kSpread = QMODE == 2 && K >= 12;   // one dependent step more in front of the digit: more fillers
  constexpr int kFB = kSpread ? 4 : 0;
  constexpr int kHeld = kSpread ? kFB + 2 : 2;
  constexpr int J1 = K < 4 ? K : 4, J2 = K < 6 ? K : 6, J3 = K - kHeld > J2 ? K - kHeld : J2;
  auto mac = [&](int r, int j, uint32_t q) {
    if (r + j < K) LOWC[r + j] += (uint64_t)n[j] * q;
    else UPC[r + j - K] += (uint64_t)n[j] * q;
  };
  uint32_t recv = 0, qprev = 0;
  uint32_t onev = 1;
  asm("" : "+v"(onev));
#pragma unroll
  for (int r = 0; r < K; ++r) {
    __builtin_amdgcn_sched_barrier(kNoValuCross);
    int jf = J3;
    if constexpr (QMODE == 2) {
      LOWC[r] += (uint64_t)qio[r] * sel0;
      if constexpr (kFB > 0) {
        __builtin_amdgcn_sched_barrier(kNoValuCross);
        if (r > 0) {
#pragma unroll
          for (int t = 0; t < kFB; ++t) mac(r - 1, jf + t, qprev);
          jf += kFB;
        }
        __builtin_amdgcn_sched_barrier(kNoValuCross);
      }
    }
    uint32_t q = and_bcast_lane0<GEO::G>(quot_digit<UNITQ>(LOWC[r], n0inv), maskv);
    if constexpr (QMODE == 1) qio[r] = q;
    __builtin_amdgcn_sched_barrier(kNoValuCross);
    if (r > 0) {
#pragma unroll
      for (int j = jf; j < K; ++j) mac(r - 1, j, qprev);
      UPC[r - 1] += (uint64_t)recv * onev;
    }
    __builtin_amdgcn_sched_barrier(kNoValuCross);
#pragma unroll
    for (int j = 0; j < J1; ++j) mac(r, j, q);
    __builtin_amdgcn_sched_barrier(kNoValuCross);
    recv = and_from_next((uint32_t)LOWC[r], maskv);
    uint64_t c = LOWC[r] >> kLimbBits;
    if constexpr (J1 < K) {
      uint64_t& accj = (r + J1 < K) ? LOWC[r + J1] : UPC[r + J1 - K];
      asm("" : "+v"(c), "+v"(accj));
    }
    __builtin_amdgcn_sched_barrier(kNoValuCross);
    if constexpr (J1 < K) mac(r, J1, q);
#pragma unroll
    for (int j = J1 + 1; j < J2; ++j) mac(r, j, q);
    __builtin_amdgcn_sched_barrier(kNoValuCross);
    if (r + 1 < K) LOWC[r + 1] += c;
    else UPC[0] += c;
    __builtin_amdgcn_sched_barrier(kNoValuCross);
#pragma unroll
    for (int j = J2; j < J3; ++j) mac(r, j, q);
    qprev = q;
  }
#pragma unroll
  for (int j = J3; j < K; ++j) mac(K - 1, j, qprev);
  UPC[K - 1] += (uint64_t)recv * onev;
#pragma unroll
  for (int j = 0; j < K; ++j) LOWC[j] = 0;
}

constexpr int kAbPad = 20;        // digits per block / limbs per lane, padded to whole 16-byte LDS accesses (K <= 20)

// 20 dwords of LDS <-> registers as five 16-byte accesses
template <int K>
__device__ __forceinline__ void ab_store20(uint32_t* dst, const uint32_t (&v)[K]) {
  static_assert(K <= kAbPad, "pad");
  uint4* p = reinterpret_cast<uint4*>(dst);
#pragma unroll
  for (int t = 0; t < kAbPad / 4; ++t) {
    uint4 w;
    w.x = 4 * t + 0 < K ? v[4 * t + 0] : 0u;
    w.y = 4 * t + 1 < K ? v[4 * t + 1] : 0u;
    w.z = 4 * t + 2 < K ? v[4 * t + 2] : 0u;
    w.w = 4 * t + 3 < K ? v[4 * t + 3] : 0u;
    p[t] = w;
  }
}
template <int K>
__device__ __forceinline__ void ab_load20(uint32_t (&v)[K], const uint32_t* src) {
  const uint4* p = reinterpret_cast<const uint4*>(src);
#pragma unroll
  for (int t = 0; t < kAbPad / 4; ++t) {
    const uint4 w = p[t];
    if (4 * t + 0 < K) v[4 * t + 0] = w.x;
    if (4 * t + 1 < K) v[4 * t + 1] = w.y;
    if (4 * t + 2 < K) v[4 * t + 2] = w.z;
    if (4 * t + 3 < K) v[4 * t + 3] = w.w;
  }
}

}  // namespace pgpu

#endif  // PAILLIERCRYPTOLIB_AMD_CSRC_HENSEL_Q_HPP_

// pailliercryptolib_amd -- CDNA4 (gfx950) device core: distributed Montgomery multiplication.
//
// Replaces the arithmetic the reference obtains from IPP-Crypto's mbx_exp_mb8 / ippsMontExp
// (call sites: ipcl/mod_exp.cpp:508-516, 549-579).  Design derived from measured gfx950 issue
// rates (profiles/r01_ubench_*.txt): v_mad_u64_u32 is a FULL-rate instruction
// (one wave64 per 4 cycles per SIMD, 39.3 T MAC32/s chip-wide), every other VALU instruction --
// a carry add, a shift, a DPP move -- costs the same issue slot, dependent issue is free, and
// ds_bpermute_b32 costs 24 cycles.  So the goal is the fewest instructions per product.  Hence:
//
//   * reduced radix: a value is L = G*K limbs of LB = 29 bits, each held in a 32-bit VGPR;
//     column sums live in 64-bit VGPR pairs, so  acc += a*b  is ONE v_mad_u64_u32 with no carry
//     handling at all ((2K + 4) * 2^58 per column lifetime < 2^64 for K <= 29, see montmul);
//   * G lanes (G in {2,4,8,16}, inside one 16-lane DPP row) co-operate on one exponentiation,
//     64/G exponentiations per wavefront; lane x owns limbs [x*K, x*K+K);
//   * word-serial Montgomery (operand scanning) in blocks of K rows: K*K MACs of a*b, then K*K
//     MACs of q*n, the K quotient digits q_r broadcast from the group's lane 0 with one DPP
//     move (row_newbcast / quad_perm), then the window slides down K columns with K DPP
//     row_shl moves; the multiplier operand b is staged in LDS and read back as
//     group-broadcast rows;
//   * values stay in [0, 2N) between multiplications (R = 2^(29*L) >= 256*N), so there is no
//     compare/subtract in the loop; canonical reduction happens once, at the very end.
#ifndef PAILLIERCRYPTOLIB_AMD_CSRC_MONT_CORE_HPP_
#define PAILLIERCRYPTOLIB_AMD_CSRC_MONT_CORE_HPP_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kargs.hpp"

// pair form (mont_reduce_rows<.., PAIR>): MACs of a row held back as fillers of the next row's opening -- FA of them
// between the lower half's digit going up and its MAC, FB between that MAC and the digit broadcast, FC behind it.
// Measured on the bench's decrypt launch (tools/build_variant.py, tools/run_variants_h.sh; same box): none held back
// beyond the usual two 5.15-5.20 ms, (3,4,1) 4.79-4.85, (2,3,3) 4.82-4.84, FC = 0 5.05: a lone wavefront has nobody
// else's instructions to cover the two dependent cross-lane steps.
#ifndef PGPU_PAIR_FA
#define PGPU_PAIR_FA 3
#endif
#ifndef PGPU_PAIR_FB
#define PGPU_PAIR_FB 4
#endif
#ifndef PGPU_PAIR_FC
#define PGPU_PAIR_FC 1
#endif
#ifndef PGPU_REG_RECVMAC
#define PGPU_REG_RECVMAC 1   // register-row form: hand-over limb added by a multiply-accumulate by one (see mont_block_rows)
#endif

namespace pgpu {


template <int G_, int K_>
struct Geo {
  static constexpr int G = G_;            // lanes per exponentiation
  static constexpr int K = K_;            // 29-bit limbs per lane
  static constexpr int L = G_ * K_;       // limbs per value
  static constexpr int RBITS = L * kLimbBits;      // Montgomery R = 2^RBITS
  static constexpr int IPW = kWave / G_;           // exponentiations per wavefront
  static constexpr int W64 = (RBITS + 63) / 64;    // 64-bit words that cover R
  static_assert(G_ == 2 || G_ == 4 || G_ == 8 || G_ == 16, "group must sit inside a DPP row");
  static_assert(2 * K_ + 4 < 64, "column accumulators would overflow 64 bits (2K products + relaxed limbs)");
};

// ---- DPP cross-lane moves (VALU, no LDS traffic) ----
// value of lane x+1 (same 16-lane row); 0 at the row end.
__device__ __forceinline__ uint32_t dpp_from_next(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x101 /*row_shl:1*/, 0xf, 0xf, true);
}
// value of lane x-1 (same 16-lane row); 0 at the row start.
__device__ __forceinline__ uint32_t dpp_from_prev(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111 /*row_shr:1*/, 0xf, 0xf, true);
}
// value of lane x-D (same 16-lane row); 0 in the first D lanes of the row.
template <int D>
__device__ __forceinline__ uint32_t dpp_from_below(uint32_t v) {
  static_assert(D >= 1 && D <= 15, "row_shr distance");
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x110 + D /*row_shr:D*/, 0xf, 0xf, true);
}
// broadcast the value held by lane 0 of every G-lane group to the whole group.
template <int G>
__device__ __forceinline__ uint32_t bcast_lane0(uint32_t v) {
  if constexpr (G == 16) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x150 /*row_newbcast:0*/, 0xf, 0xf, true);
  } else if constexpr (G == 8) {
    int t = __builtin_amdgcn_update_dpp(0, (int)v, 0x00 /*quad_perm:[0,0,0,0]*/, 0xf, 0xf, true);
    // quads 1 and 3 take the value from 4 lanes below (quads 0 and 2 keep theirs)
    return (uint32_t)__builtin_amdgcn_update_dpp(t, t, 0x114 /*row_shr:4*/, 0xf, 0xa, false);
  } else if constexpr (G == 4) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x00 /*quad_perm:[0,0,0,0]*/, 0xf, 0xf, true);
  } else {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xa0 /*quad_perm:[0,0,2,2]*/, 0xf, 0xf, true);
  }
}

// value held by lane S of every G-lane group, in every lane of the group (G in {2, 4, 16}: one DPP move; G = 8:
// two bank-masked row broadcasts into one register)
template <int G, int S>
__device__ __forceinline__ uint32_t bcast_lane(uint32_t v) {
  static_assert(G == 2 || G == 4 || G == 8 || G == 16, "group must sit inside a DPP row");
  static_assert(S >= 0 && S < G, "lane index inside the group");
  // (mov_dpp, not update_dpp: every lane reads a valid source, so there is no "old" value to materialise)
  if constexpr (G == 8) {
    int t = __builtin_amdgcn_update_dpp(0, (int)v, 0x150 + S /*row_newbcast:S*/, 0xf, 0x3, false);
    return (uint32_t)__builtin_amdgcn_update_dpp(t, (int)v, 0x150 + 8 + S /*row_newbcast:8+S*/, 0xf, 0xc, false);
  } else if constexpr (G == 16) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x150 + S /*row_newbcast:S*/, 0xf, 0xf, true);
  } else if constexpr (G == 4) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, S | (S << 2) | (S << 4) | (S << 6) /*quad_perm:[S,S,S,S]*/, 0xf, 0xf, true);
  } else {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, S | (S << 2) | ((2 + S) << 4) | ((2 + S) << 6), 0xf, 0xf,
                                              true);   // quad_perm:[S,S,2+S,2+S]
  }
}

// ---- fused DPP + mask (one VALU slot: v_and_b32_dpp) ----
// m must live in a VGPR (callers launder it through an empty asm): with a register mask hipcc's DPP
// combiner folds the v_and into the DPP move, schedules it freely and inserts the "VALU write -> DPP
// read" wait states only where they are needed.  Only the two-instruction G = 8 broadcast is written
// by hand (bank-masked halves of one destination are beyond the combiner); its leading s_nop 1 covers
// that hazard, which the compiler cannot see inside an asm statement.
// (value of lane x+1) & m; 0 at the row end
__device__ __forceinline__ uint32_t and_from_next(uint32_t v, uint32_t m) {
  return dpp_from_next(v) & m;
}
// (value held by lane 0 of the G-lane group) & m, in every lane of the group
template <int G>
__device__ __forceinline__ uint32_t and_bcast_lane0(uint32_t v, uint32_t m) {
  if constexpr (G == 8) {
    uint32_t r;
    // lanes 0-7 take lane 0, lanes 8-15 take lane 8: two bank-masked broadcasts into one VGPR
    asm("s_nop 1\n\tv_and_b32_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0x3 bound_ctrl:1\n\t"
        "v_and_b32_dpp %0, %1, %2 row_newbcast:8 row_mask:0xf bank_mask:0xc bound_ctrl:1"
        : "=&v"(r) : "v"(v), "v"(m));
    return r;
  } else {
    return bcast_lane0<G>(v) & m;
  }
}

// The quotient digit of a reduction row (before the 29-bit mask): the low limb itself under a unit-quotient modulus,
// else low limb * n0'.  PGPU_QDIGIT_MAD=1 takes the product as the low half of a v_mad_u64_u32 instead of v_mul_lo_u32
// (the empty asm keeps the 64-bit product alive): measured equal within 1 % (profiles/r03_ubench_pairmul.txt) -- what a
// multiplied digit costs is the multiplication AND the two wait states between it and the DPP move that reads it, in
// a pair product 2 x (1 + nop) per row: +9 %, which is what a 36-limb loop modulus (the prime itself instead of
// P = p*k, 38 limbs) would save.  Unit digits stay.
#ifndef PGPU_QDIGIT_MAD
#define PGPU_QDIGIT_MAD 0
#endif
template <bool UNITQ>
__device__ __forceinline__ uint32_t quot_digit(uint64_t col, uint32_t n0inv) {
  if constexpr (UNITQ) {
    return (uint32_t)col;
  } else if constexpr (PGPU_QDIGIT_MAD != 0) {
    uint64_t t = (uint64_t)(uint32_t)col * n0inv;
    asm("" : "+v"(t));
    return (uint32_t)t;
  } else {
    return (uint32_t)col * n0inv;
  }
}

// One K-row block of the word-serial Montgomery product.
//   LOWC: the K columns this lane shares with nobody below it (get reduced / passed down)
//   UPC : the next K columns (become LOWC of the next block; enter as zero)
// On exit UPC holds the new low half (old UPC + the K normalised limbs received from lane x+1)
// and LOWC is zero, i.e. the caller swaps the roles of the two arrays for the next block.
// SQR = true: the multiplier rows ARE the multiplicand (a squaring).  Every cross product
// a_X[j]*a_S[r] (lane chunk X, row block S) then occurs twice in the full square -- once in block
// (X,S), once in block (S,X) at the same column -- so each block computes only the pairs with
// j < r, with one factor doubled (a2 = 2*a), plus the diagonal j == r once:
//   sum_{X,S} [ sum_{j<r} 2 a_X[j] a_S[r] + sum_r a_X[r] a_S[r] ] = a^2   (rename X<->S, j<->r)
// K(K+1)/2 MACs instead of K^2, identical instruction stream in every lane and block.
// b: the K multiplier rows of this block, in registers; a2 = 2*a (squarings only: the doubled cross products take
// the doubling on the multiplicand side, computed once per multiplication instead of once per row and block).
// RECVMAC: the received limb joins its column through a multiply-accumulate by one instead of a 64-bit add (the
// register-row form: its register allocation otherwise spends a v_mov per row on the zero high half of the addend).
// PAIR (hensel.hpp): the G-lane group is one half of a 2G-lane group; the quotient digit of the lower half is also
// added to column r of the upper half's low lane (selB = 1 in lane G of the 2G lanes, else 0) before the upper half
// takes its own digit.
template <class GEO, bool UNITQ, bool RECVMAC, bool PAIR = false>
__device__ __forceinline__ void mont_reduce_rows(uint64_t (&LOWC)[GEO::K], uint64_t (&UPC)[GEO::K],
                                                 const uint32_t (&n)[GEO::K], uint32_t n0inv, uint32_t selB = 0) {
  constexpr int K = GEO::K;
  uint32_t maskv = kLimbMask;
  asm("" : "+v"(maskv));   // keep the mask in a VGPR (v_and_b32_dpp takes no literal)
  // The dependent chain of a row is q -> MAC (columns r, r+1) -> carry -> q of the next row.  A lone
  // wavefront on a SIMD (small batches) stalls when those sit back to back, so the row is issued as
  // q | 4 MACs | limb hand-over + carry shift | 2 MACs | carry add | remaining MACs, pinned with VALU
  // scheduling barriers (memory and scalar instructions may still cross).
  constexpr int kNoValuCross = 0x3fc;
  // (PAIR: two dependent cross-lane steps open a row -- the lower half's digit goes up, then both digits are
  // broadcast -- so more of the previous row's MACs are held back as fillers: two between the steps, the rest after)
  constexpr bool kSpread = PAIR && K >= 12;
  constexpr int kFA = kSpread ? PGPU_PAIR_FA : 0, kFB = kSpread ? PGPU_PAIR_FB : 0;
  constexpr int kHeld = kSpread ? kFA + kFB + PGPU_PAIR_FC : 2;
  constexpr int J1 = K < 4 ? K : 4, J2 = K < 6 ? K : 6, J3 = K - kHeld > J2 ? K - kHeld : J2;
  auto mac = [&](int r, int j, uint32_t q) {
    if (r + j < K) LOWC[r + j] += (uint64_t)n[j] * q;
    else UPC[r + j - K] += (uint64_t)n[j] * q;
  };
  uint32_t recv = 0, qprev = 0;
  uint32_t onev = 1;
  if constexpr (RECVMAC) asm("" : "+v"(onev));   // a VGPR holding 1 (keeps the product a v_mad_u64_u32)
  auto add_recv = [&](uint64_t& col) {
    if constexpr (RECVMAC) col += (uint64_t)recv * onev;
    else col += recv;
  };
#pragma unroll
  for (int r = 0; r < K; ++r) {
    // UNITQ: the modulus is == -1 mod 2^29 (capi.hip: build_modctx scales it), so n0' = 1
    __builtin_amdgcn_sched_barrier(kNoValuCross);
    int jf = J3;   // next held-back MAC of the previous row
    if constexpr (PAIR) {
      static_assert(GEO::G <= 8, "a pair is two G-lane halves of a 2G-lane group inside a DPP row");
      // lane G of the group takes the digit from lane 0, G lanes below it (the other lanes multiply theirs by 0)
      const uint32_t qa = dpp_from_below<GEO::G>(quot_digit<UNITQ>(LOWC[r], n0inv)) & maskv;
      if constexpr (kFA > 0) {
        __builtin_amdgcn_sched_barrier(kNoValuCross);
        if (r > 0) {
#pragma unroll
          for (int t = 0; t < kFA; ++t) mac(r - 1, jf + t, qprev);
          jf += kFA;
        }
        __builtin_amdgcn_sched_barrier(kNoValuCross);
      }
      LOWC[r] += (uint64_t)qa * selB;
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
    // fillers between the broadcast and its first use: the last two MACs of the previous row, and
    // the hand-over of column r-1, final since the previous row: its 29-bit limb belongs to lane x-1,
    // whose window overlaps it at its column K+r-1 -- so every lane adds the limb it received from
    // lane x+1 straight into UPC[r-1] (window slide, no deferred copies).  In the group's lane 0 the
    // limb is 0 by construction of q, so the top lane of the group below receives 0: no masking.
    __builtin_amdgcn_sched_barrier(kNoValuCross);
    if (r > 0) {
#pragma unroll
      for (int j = jf; j < K; ++j) mac(r - 1, j, qprev);
      add_recv(UPC[r - 1]);
    }
    __builtin_amdgcn_sched_barrier(kNoValuCross);
#pragma unroll
    for (int j = 0; j < J1; ++j) mac(r, j, q);
    __builtin_amdgcn_sched_barrier(kNoValuCross);
    recv = and_from_next((uint32_t)LOWC[r], maskv);
    uint64_t c = LOWC[r] >> kLimbBits;             // the rest of column r carries into column r+1
    // anchor the shift here, away from the add below: the accumulator of the next MAC is laundered
    // together with c, so the shift has to precede that MAC (an empty asm is not a VALU op and could
    // itself be moved across the VALU scheduling barriers)
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
  add_recv(UPC[K - 1]);
  // the low half is consumed; it becomes the (zero) upper half of the next block
#pragma unroll
  for (int j = 0; j < K; ++j) LOWC[j] = 0;
}

template <class GEO, bool SQR, bool UNITQ, bool RECVMAC = false>
__device__ __forceinline__ void mont_block_rows(uint64_t (&LOWC)[GEO::K], uint64_t (&UPC)[GEO::K],
                                                const uint32_t (&a)[GEO::K], const uint32_t (&a2)[GEO::K],
                                                const uint32_t (&n)[GEO::K], uint32_t n0inv,
                                                const uint32_t (&b)[GEO::K]) {
  constexpr int K = GEO::K;
  // phase A: acc += a_chunk * b_rows  (v_mad_u64_u32 only, no carries)
#pragma unroll
  for (int r = 0; r < K; ++r) {
#pragma unroll
    for (int j = 0; j < K; ++j) {
      uint64_t p;
      if constexpr (SQR) {
        if (j > r) continue;
        p = (uint64_t)(j < r ? a2[j] : a[j]) * b[r];
      } else {
        p = (uint64_t)a[j] * b[r];
      }
      if (r + j < K) LOWC[r + j] += p;
      else UPC[r + j - K] += p;
    }
  }
  // phase B: K quotient digits, each followed by acc += n_chunk * q
  mont_reduce_rows<GEO, UNITQ, RECVMAC>(LOWC, UPC, n, n0inv);
}

// rows staged in LDS by the caller (group-broadcast reads): the form for launches with several wavefronts per
// SIMD, where somebody else's instructions cover the LDS round trip
template <class GEO, bool SQR, bool UNITQ>
__device__ __forceinline__ void mont_block(uint64_t (&LOWC)[GEO::K], uint64_t (&UPC)[GEO::K],
                                           const uint32_t (&a)[GEO::K], const uint32_t (&a2)[GEO::K],
                                           const uint32_t (&n)[GEO::K], uint32_t n0inv,
                                           const uint32_t* __restrict__ brow) {
  constexpr int K = GEO::K;
  uint32_t b[K];
#pragma unroll
  for (int r = 0; r < K; ++r) b[r] = brow[r];
  mont_block_rows<GEO, SQR, UNITQ>(LOWC, UPC, a, a2, n, n0inv, b);
}

// rows taken straight from the registers of lane S of the group (m: the multiplier's limbs, lane-distributed like
// a): one DPP broadcast per row limb, no LDS.  A wavefront that is ALONE on its SIMD
// (the bench's CRT-decrypt launch, every small batch) pays every LDS round trip in full -- ~110 exposed cycles
// per block plus ~25 ds_read / s_waitcnt issue slots and the staging writes of every multiplication -- which is
// more than these K cheap VALU instructions; with several wavefronts per SIMD the LDS form wins
// (profiles/r02_ubench_lone_wave.txt).
template <class GEO, bool SQR, bool UNITQ, int S>
__device__ __forceinline__ void mont_block_reg(uint64_t (&LOWC)[GEO::K], uint64_t (&UPC)[GEO::K],
                                               const uint32_t (&a)[GEO::K], const uint32_t (&a2)[GEO::K],
                                               const uint32_t (&n)[GEO::K], uint32_t n0inv,
                                               const uint32_t (&m)[GEO::K]) {
  constexpr int K = GEO::K;
  uint32_t b[K];
#pragma unroll
  for (int r = 0; r < K; ++r) b[r] = bcast_lane<GEO::G, S>(m[r]);
  mont_block_rows<GEO, SQR, UNITQ, PGPU_REG_RECVMAC != 0>(LOWC, UPC, a, a2, n, n0inv, b);
}

// Epilogue of a multiplication: the K finished columns -> relaxed 29-bit limbs.
template <class GEO>
__device__ __forceinline__ void montmul_finish(uint32_t (&r)[GEO::K], const uint64_t (&c0)[GEO::K]) {
  constexpr int K = GEO::K;
  // pass 1: local carry propagation
  uint64_t c = 0;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    uint64_t t = c0[j] + c;
    r[j] = (uint32_t)t & kLimbMask;
    c = t >> kLimbBits;
  }
  // pass 2: the carry-out of lane x-1 (< 2^36) enters lane x WITHOUT rippling: its low 29 bits join
  // limb 0 and the rest joins limb 1, so limb 0 < 2^30 and limb 1 < 2^29 + 2^7 on exit -- a relaxed form
  // every consumer accepts: the next multiplication (a column's K + K products then sum to less than
  // (2K + 4) * 2^58 -- one pair of oversized factors per column, doubled in a squaring -- i.e. < 2^64 for
  // K <= 29), the doubled rows (2^31 fits), add_normalise and full_normalise.
  // The value is < R, so the top lane of a group never carries out and nothing leaks into the next group.
  uint32_t maskv = kLimbMask;
  asm("" : "+v"(maskv));
  r[0] += dpp_from_prev((uint32_t)c) & maskv;
  r[1] += dpp_from_prev((uint32_t)(c >> kLimbBits));
}

// r = a * b * R^-1 mod N (lazy: inputs < 4N -> output < 2N), b read from LDS at bl[0..L).
// Output limbs are < 2^29 except limbs 0 and 1 of a lane, which hold an unrippled carry (see pass 2).
template <class GEO, bool SQR = false, bool UNITQ = false>
__device__ __forceinline__ void montmul(uint32_t (&r)[GEO::K], const uint32_t (&a)[GEO::K],
                                        const uint32_t* __restrict__ bl,
                                        const uint32_t (&n)[GEO::K], uint32_t n0inv) {
  constexpr int K = GEO::K;
  uint64_t c0[K], c1[K];
  uint32_t a2[K];
#pragma unroll
  for (int j = 0; j < K; ++j) {
    c0[j] = 0;
    c1[j] = 0;
    a2[j] = SQR ? a[j] << 1 : 0;
  }
#pragma unroll 1
  for (int s = 0; s < GEO::G; s += 2) {
    mont_block<GEO, SQR, UNITQ>(c0, c1, a, a2, n, n0inv, bl + s * K);
    mont_block<GEO, SQR, UNITQ>(c1, c0, a, a2, n, n0inv, bl + (s + 1) * K);
  }
  montmul_finish<GEO>(r, c0);
}

// The same multiplication with the multiplier rows broadcast from registers (mont_block_reg): m holds the
// multiplier's limbs, lane-distributed like a (a squaring passes a itself).  All G blocks are inline.
template <class GEO, bool SQR, bool UNITQ, int S>
__device__ __forceinline__ void montmul_reg_blocks(uint64_t (&c0)[GEO::K], uint64_t (&c1)[GEO::K],
                                                   const uint32_t (&a)[GEO::K], const uint32_t (&a2)[GEO::K],
                                                   const uint32_t (&n)[GEO::K], uint32_t n0inv,
                                                   const uint32_t (&m)[GEO::K]) {
  if constexpr (S < GEO::G) {
    mont_block_reg<GEO, SQR, UNITQ, S>(c0, c1, a, a2, n, n0inv, m);
    mont_block_reg<GEO, SQR, UNITQ, S + 1>(c1, c0, a, a2, n, n0inv, m);
    montmul_reg_blocks<GEO, SQR, UNITQ, S + 2>(c0, c1, a, a2, n, n0inv, m);
  }
}
template <class GEO, bool SQR = false, bool UNITQ = false>
__device__ __forceinline__ void montmul_reg(uint32_t (&r)[GEO::K], const uint32_t (&a)[GEO::K],
                                            const uint32_t (&m)[GEO::K], const uint32_t (&n)[GEO::K],
                                            uint32_t n0inv) {
  constexpr int K = GEO::K;
  uint64_t c0[K], c1[K];
  uint32_t a2[K];
#pragma unroll
  for (int j = 0; j < K; ++j) {
    c0[j] = 0;
    c1[j] = 0;
    a2[j] = SQR ? a[j] << 1 : 0;
  }
  montmul_reg_blocks<GEO, SQR, UNITQ, 0>(c0, c1, a, a2, n, n0inv, m);
  montmul_finish<GEO>(r, c0);
}

// Fully canonical limbs (< 2^29 everywhere).  Data-dependent trip count (<= G+1); used only
// outside the multiplication loop.  Values may carry a signed borrow in r[] limbs on entry?  No:
// unsigned only.  x = lane index inside the group.
template <class GEO>
__device__ __forceinline__ void full_normalise(uint32_t (&r)[GEO::K], int x) {
  constexpr int K = GEO::K;
  for (;;) {
    uint32_t c = 0;
#pragma unroll
    for (int j = 0; j < K; ++j) {
      uint32_t u = r[j] + c;
      r[j] = u & kLimbMask;
      c = u >> kLimbBits;
    }
    uint32_t cin = dpp_from_prev(c);
    if (x == 0) cin = 0;
    r[0] += cin;
    if (__ballot(cin != 0) == 0) break;
  }
}

// Lane-parallel canonical reduction: r (canonical limbs, value V < 3N) -> V mod N.
// Each round forms D = V - N limb-wise with a borrow that ripples inside the lane and hops to the
// next lane over DPP (repeated until no borrow is in flight -- data dependent, a handful of
// instructions, used once per exponentiation / product); the sign of D, known in the group's top
// lane, selects D or V for the whole group.
template <class GEO>
__device__ __forceinline__ void cond_sub_limbs(uint32_t (&r)[GEO::K], const uint32_t (&nt)[GEO::K], int x,
                                               int lane) {
  constexpr int K = GEO::K, G = GEO::G;
  const int top_lane = (lane / G) * G + (G - 1);
#pragma unroll 1
  for (int round = 0; round < 2; ++round) {
    uint32_t d[K];
    uint32_t b = 0;        // borrow into limb j
#pragma unroll
    for (int j = 0; j < K; ++j) {
      uint32_t t = r[j] - nt[j] - b;
      d[j] = t & kLimbMask;
      b = t >> 31;         // limbs are < 2^29, so a negative difference has bit 31 set
    }
    uint32_t top_borrow = b;   // meaningful in the top lane only; accumulates over the ripple passes
    for (;;) {
      uint32_t bin = dpp_from_prev(b);
      if (x == 0) bin = 0;
      if (__ballot(bin != 0) == 0) break;
      b = bin;
#pragma unroll
      for (int j = 0; j < K; ++j) {
        uint32_t t = d[j] - b;
        d[j] = t & kLimbMask;
        b = t >> 31;
      }
      top_borrow |= b;
    }
    const uint32_t negative = (uint32_t)__shfl((int)top_borrow, top_lane);   // V < N: keep V
    if (__ballot(negative == 0) == 0) break;                                // nothing left to subtract
    if (!negative) {
#pragma unroll
      for (int j = 0; j < K; ++j) r[j] = d[j];
    }
  }
}

// limb i (29 bits at bit offset 29*i) of a little-endian u64 array that is zero-padded by one word.
__device__ __forceinline__ uint32_t limb_from_words(const uint64_t* w, int i) {
  int bit = i * kLimbBits;
  int word = bit >> 6, sh = bit & 63;
  uint64_t v = w[word] >> sh;
  if (sh > 64 - kLimbBits) v |= w[word + 1] << (64 - sh);
  return (uint32_t)v & kLimbMask;
}

// 64-bit word wi of the value whose canonical 29-bit limbs are limb[0..L).
__device__ __forceinline__ uint64_t word_from_limbs(const uint32_t* limb, int L, int wi) {
  int bit0 = wi * 64;
  int i0 = bit0 / kLimbBits;
  uint64_t v = 0;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    int i = i0 + t;
    if (i < L) {
      int pos = i * kLimbBits - bit0;  // may be negative for t == 0
      uint64_t lv = limb[i];
      if (pos < 0) v |= lv >> (-pos);
      else if (pos < 64) v |= lv << pos;
    }
  }
  return v;
}

}  // namespace pgpu

#endif  // PAILLIERCRYPTOLIB_AMD_CSRC_MONT_CORE_HPP_

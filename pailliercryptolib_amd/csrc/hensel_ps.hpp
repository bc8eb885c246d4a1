// pailliercryptolib_amd -- the split form with a whole exponentiation in ONE lane, by PRODUCT SCANNING (round 5):
// hensel_decrypt_ps_kernel<K, LB, MINW>, CRT decrypt of resident batches under 2048-bit keys (K = 38 limbs of LB = 28 bits
// per half: the headline's dominant kernel), 3072-bit keys (K = 56, LB = 28; one wavefront per SIMD only) and 1024-bit keys
// (K = 19, LB = 29: the successor of hensel_decrypt_lane_kernel<20>).  Instantiated in k_hensel.hip parts 31 / 33 / 34.
// Headroom: R = 2^(LB*K) >= 16 * P is enough here (canonical limbs after every product; the bounds are pushed through the
// kernel's flow in tests/test_hensel_model.py) -- the multi-lane forms keep 256 for their relaxed limbs.
//
// Round 4's one-lane kernel (hensel_decrypt_lane_kernel<20>, retired in round 6: this kernel's <19,29,2> build serves its
// 1024-bit keys 17-20 % faster) scanned by OPERAND: 2K column accumulators (4K registers) that every reduction row
// normalises with a shift and an add -- at K = 38 those accumulators alone are 152 registers.  Here a
// Montgomery product runs column by column: ONE 64-bit accumulator takes every product a_i * c_j with i + j = col and
// every q_i * n_j of the digits found so far, gives up its low limb (the quotient digit of a low column, the result limb
// of a high one) and shifts down into the next column.  Per column: the products, one v_and_b32, one v_lshrrev_b64 (and
// for a digit column one multiply-accumulate by one: P == -1 mod 2^LB makes q * n_0 a shift and an add) -- no DPP, no LDS,
// no per-row carry add, no window of columns in registers.  A pair squaring of the 2048-bit class is 4 997 products in
// ~5 530 instructions (90 %) against 2 x 3 287 in hensel_decrypt_seq_kernel<2,19> (77 %): 16 % fewer issue slots per
// exponentiation; a general product 7 144 in ~7 640 against 2 x 4 356.
//
// A column of 2*a*b + q*n sums 3K products: with K = 38 that needs limbs of 28 bits (3 * 38 * 2^56 < 2^63); the 1024-bit
// class keeps 29 (3 * 20 * 2^58 < 2^64).  The key image therefore carries a constant set of its own for this kernel
// (capi.cpp: build_hensel_set with the limb width as a parameter); the pair rows it reads are the 29-bit rows every
// other kernel writes (re-limbed on entry), its output the same canonical words (mp, mq) for crt_kernel.
// The modulus limbs are wave-uniform (a wavefront serves ONE side of the key) and live in SGPRs.
//
// The price is the launch size: 64 exponentiations per wavefront -- 8192 ciphertexts are 256 wavefronts, a quarter of the
// chip's SIMDs (capi.cpp picks the form for launches that cover the chip alone or together with their neighbour lanes).
// Results bit-identical with every other decrypt form (tests/test_gpu_round5.py).  Reference: the two half-width
// exponentiations of PrivateKey::decryptCRT, ipcl/pri_key.cpp:114-146.
#ifndef PAILLIERCRYPTOLIB_AMD_CSRC_HENSEL_PS_HPP_
#define PAILLIERCRYPTOLIB_AMD_CSRC_HENSEL_PS_HPP_

#include <utility>

#include "hensel.hpp"

// which multiply-accumulate chains are pinned to the carry of the column below (ps_mac_pinned): 0 none, 1 the q*n chain of
// symmetric columns (default), 2 every chain (A/B, tools/build_variant.py)
#ifndef PGPU_PS_PIN
#define PGPU_PS_PIN 1
#endif
// A/B (tools/ubench_ps.hip, round 6): 1 = the operand products of a general column run as a chain of their own, joined to the
// q*n chain (which sits on the carry of the column below) by ONE 64-bit add per column -- two independent accumulator chains
// instead of one long one.  Measured: profiles/r06_ubench_ps_cycles.txt.
#ifndef PGPU_PS_SPLIT
#define PGPU_PS_SPLIT 0
#endif

namespace pgpu {

template <int LB>
struct PsLimb {
  static constexpr uint32_t mask = (1u << LB) - 1;
};

// compile-time loop: f(std::integral_constant<int, 0>) ... f(std::integral_constant<int, N-1>) -- every index a constant
// expression, so that limb arrays stay in registers whatever the unroller thinks of 2K x K trip counts
template <class F, int... Is>
__device__ __forceinline__ void ps_static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void ps_static_for(F&& f) {
  ps_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// a value every lane of the wavefront holds alike, moved to an SGPR (the compiler does not prove it for loads behind the
// kernel's own stores)
__device__ __forceinline__ uint32_t ps_uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// acc += x * y  (one v_mad_u64_u32)
__device__ __forceinline__ void ps_mac(uint64_t& acc, uint32_t x, uint32_t y) { acc += (uint64_t)x * y; }
// ... as a link of a chain the optimiser must leave in this order: the (empty) second use of the partial sum keeps the
// sum of the column from being re-associated.  Left alone the compiler builds every chain from zero and adds the carry of
// the column below last -- right for latency, but one 64-bit add more per column; a column that has a second chain anyway
// (the cross products of a squaring, joined by the shift-and-add that doubles them) pins the first one to the carry.
__device__ __forceinline__ void ps_mac_pinned(uint64_t& acc, uint32_t x, uint32_t y) {
  acc += (uint64_t)x * y;
  asm volatile("" ::"v"(acc));
}

// One Montgomery product by product scanning, everything in this lane:
//   r = (x1*y1 [+ x2*y2] [+ qio as a number]) * R^-1 mod n      (lazy; canonical limbs in and out)
// NP: number of products (1 or 2); SYM: y1 IS x1 (a squaring: the cross products once, doubled by the shift that adds
// their sum; the diagonal once); QMODE 0: plain; 1: the quotient digits are recorded in qio; 2: qio[col] is added to
// column col before its digit is taken (the correction term of the pair product).
// UNITQ: n == -1 mod 2^LB.  Then the digit is the column's low limb, and  acc + q*n_0 = acc - q + q*2^LB  turns the
// digit step into "+ q one column up" -- which joins the q*n_1 of that column: n1p = n_1 + 1 (<= 2^LB) is passed in, and a
// digit column costs one v_and_b32 and one v_lshrrev_b64.
// r may be x2 or y1 (limb col-K of the operands is dead when column col delivers limb col-K of the result).
template <int K, int LB, int NP, bool SYM, int QMODE, bool UNITQ>
__device__ __forceinline__ void ps_montmul(uint32_t (&r)[K], const uint32_t (&x1)[K], const uint32_t (&y1)[K],
                                           const uint32_t (&x2)[K], const uint32_t (&y2)[K], const uint32_t (&n)[K],
                                           uint32_t n1p, uint32_t n0inv, uint32_t (&qio)[K]) {
  constexpr uint32_t M = PsLimb<LB>::mask;
  static_assert(!(SYM && NP != 1), "a symmetric product is a single one");
  uint32_t q[K];
  uint64_t acc = 0;
  uint32_t onev = 1;
  asm("" : "+v"(onev));   // (keeps "+= 32-bit value" ONE v_mad_u64_u32 instead of an add / add-with-carry pair)
  ps_static_for<2 * K>([&](auto colc) __attribute__((always_inline)) {
    constexpr int col = decltype(colc)::value;
    // ---- q_i * n_j of the digits found so far, j >= 1 (n_0 belongs to the digit step), onto the carry of the column below
    {
      constexpr int ilo = col < K ? 0 : col - K + 1;
      constexpr int ihi = col < K ? col : K;               // i <= col - 1
      if constexpr (ihi > ilo) {
        ps_static_for<ihi - ilo>([&](auto ic) __attribute__((always_inline)) {
          constexpr int i = ilo + decltype(ic)::value;
          if constexpr (PGPU_PS_PIN == 2 || PGPU_PS_SPLIT == 1 || (PGPU_PS_PIN == 1 && SYM)) ps_mac_pinned(acc, q[i], (UNITQ && col - i == 1) ? n1p : n[col - i]);
          else ps_mac(acc, q[i], (UNITQ && col - i == 1) ? n1p : n[col - i]);
        });
      }
    }
    // ---- products of the operands ----
    if constexpr (SYM) {
      constexpr int ilo = col < K ? 0 : col - K + 1;       // pairs i < j, i + j = col, j < K
      constexpr int ihi = (col + 1) / 2;                   // i < col - i
      if constexpr (ihi > ilo) {
        uint64_t cross = 0;
        ps_static_for<ihi - ilo>([&](auto ic) __attribute__((always_inline)) {
          constexpr int i = ilo + decltype(ic)::value;
          ps_mac(cross, x1[i], x1[col - i]);
        });
        acc += cross << 1;
      }
      if constexpr (col % 2 == 0) ps_mac(acc, x1[col / 2], x1[col / 2]);
    } else {
      constexpr int ilo = col < K ? 0 : col - K + 1;
      constexpr int ihi = col < K ? col + 1 : K;
      if constexpr (PGPU_PS_SPLIT == 1 && (ihi - ilo) * NP >= 4) {
        uint64_t prod = 0;
        ps_static_for<ihi - ilo>([&](auto ic) __attribute__((always_inline)) {
          constexpr int i = ilo + decltype(ic)::value;
          ps_mac_pinned(prod, x1[i], y1[col - i]);
          if constexpr (NP == 2) ps_mac_pinned(prod, x2[i], y2[col - i]);
        });
        acc += prod;
      } else
      ps_static_for<ihi - ilo>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = ilo + decltype(ic)::value;
        if constexpr (PGPU_PS_PIN == 2) {
          ps_mac_pinned(acc, x1[i], y1[col - i]);
          if constexpr (NP == 2) ps_mac_pinned(acc, x2[i], y2[col - i]);
        } else {
          ps_mac(acc, x1[i], y1[col - i]);
          if constexpr (NP == 2) ps_mac(acc, x2[i], y2[col - i]);
        }
      });
    }
    if constexpr (col < K) {
      if constexpr (QMODE == 2) ps_mac(acc, qio[col], onev);
      if constexpr (UNITQ) {
        q[col] = (uint32_t)acc & M;
        acc >>= LB;                          // (the "+ q" of acc + q*(2^LB - 1) rides on n1p in the next column)
      } else {
        q[col] = ((uint32_t)acc * n0inv) & M;
        ps_mac(acc, q[col], n[0]);
        acc >>= LB;
      }
      if constexpr (QMODE == 1) qio[col] = q[col];
    } else {
      r[col - K] = (uint32_t)acc & M;
      acc >>= LB;
    }
  });
}

// this lane's slot of the wavefront's parking area: K4 = ceil(K/4) 16-byte rows of 64 lanes (conflict-free b128 accesses)
template <int K>
__device__ __forceinline__ void ps_park_store(uint4* slot, const uint32_t (&v)[K]) {
  constexpr int K4 = (K + 3) / 4;
#pragma unroll
  for (int t = 0; t < K4; ++t) {
    uint4 w;
    w.x = v[4 * t];
    w.y = 4 * t + 1 < K ? v[4 * t + 1] : 0u;
    w.z = 4 * t + 2 < K ? v[4 * t + 2] : 0u;
    w.w = 4 * t + 3 < K ? v[4 * t + 3] : 0u;
    slot[t * kWave] = w;
  }
}
template <int K>
__device__ __forceinline__ void ps_park_load(uint32_t (&v)[K], const uint4* slot) {
  constexpr int K4 = (K + 3) / 4;
#pragma unroll
  for (int t = 0; t < K4; ++t) {
    const uint4 w = slot[t * kWave];
    v[4 * t] = w.x;
    if (4 * t + 1 < K) v[4 * t + 1] = w.y;
    if (4 * t + 2 < K) v[4 * t + 2] = w.z;
    if (4 * t + 3 < K) v[4 * t + 3] = w.w;
  }
}

// out = what the slot holds, then the slot takes in: row by row, so that the registers of `in` are free as soon as they
// are written (LDS keeps one wavefront's accesses in order)
template <int K>
__device__ __forceinline__ void ps_park_swap(uint4* slot, uint32_t (&out)[K], const uint32_t (&in)[K]) {
  constexpr int K4 = (K + 3) / 4;
#pragma unroll
  for (int t = 0; t < K4; ++t) {
    const uint4 r = slot[t * kWave];
    uint4 w;
    w.x = in[4 * t];
    w.y = 4 * t + 1 < K ? in[4 * t + 1] : 0u;
    w.z = 4 * t + 2 < K ? in[4 * t + 2] : 0u;
    w.w = 4 * t + 3 < K ? in[4 * t + 3] : 0u;
    slot[t * kWave] = w;
    out[4 * t] = r.x;
    if (4 * t + 1 < K) out[4 * t + 1] = r.y;
    if (4 * t + 2 < K) out[4 * t + 2] = r.z;
    if (4 * t + 3 < K) out[4 * t + 3] = r.w;
  }
}

// (a, b) = (a, b)^2: the Montgomery square of a pair (hensel.hpp), lazy, canonical limbs in and out:
//   t = a*a with its digits q;  b = (2*a*b + q) reduced;  a = t
template <int K, int LB>
__device__ __forceinline__ void ps_pairsqr(uint32_t (&a)[K], uint32_t (&b)[K], const uint32_t (&n)[K], uint32_t n1p) {
  static_assert(3 * (uint64_t)K * ((uint64_t)1 << (2 * LB - 32)) < ((uint64_t)1 << 32), "a column sums up to 3K products below 2^(2LB): must stay below 2^64");
  uint32_t qd[K], t[K];
  ps_montmul<K, LB, 1, true, 1, true>(t, a, a, a, a, n, n1p, 0, qd);
#pragma unroll
  for (int j = 0; j < K; ++j) b[j] <<= 1;
  ps_montmul<K, LB, 1, false, 2, true>(b, a, b, a, b, n, n1p, 0, qd);
#pragma unroll
  for (int j = 0; j < K; ++j) a[j] = t[j];
}

// (a, b) = (a, b) (x) (c, d): the general product.  Six K-limb values are live in it (a, b, c, d, the digits, t): the one
// that idles waits in LDS -- b through the first product (a*c), t through the second (a*d + b*c + q) -- so that the
// kernel keeps to 256 registers (two wavefronts per SIMD for large launches).  slot: this lane's parking slot.
template <int K, int LB, bool UNITQ>
__device__ __forceinline__ void ps_pairmul(uint32_t (&a)[K], uint32_t (&b)[K], const uint32_t (&c)[K],
                                           const uint32_t (&d)[K], const uint32_t (&n)[K], uint32_t n1p, uint32_t n0inv,
                                           uint4* slot) {
  static_assert(3 * (uint64_t)K * ((uint64_t)1 << (2 * LB - 32)) < ((uint64_t)1 << 32), "a column sums up to 3K products below 2^(2LB): must stay below 2^64");
  uint32_t qd[K];
  {
    uint32_t t[K];
    ps_park_store<K>(slot, b);
    __builtin_amdgcn_sched_barrier(0);
    ps_montmul<K, LB, 1, false, 1, UNITQ>(t, a, c, a, c, n, n1p, n0inv, qd);
    __builtin_amdgcn_sched_barrier(0);
    ps_park_swap<K>(slot, b, t);
    __builtin_amdgcn_sched_barrier(0);
  }
  ps_montmul<K, LB, 2, false, 2, UNITQ>(b, a, d, b, c, n, n1p, n0inv, qd);
  __builtin_amdgcn_sched_barrier(0);
  ps_park_load<K>(a, slot);
}

// r = x*y*R^-1 mod n, one product
template <int K, int LB, bool UNITQ>
__device__ __forceinline__ void ps_mul(uint32_t (&r)[K], const uint32_t (&x)[K], const uint32_t (&y)[K],
                                       const uint32_t (&n)[K], uint32_t n1p, uint32_t n0inv) {
  uint32_t none[K];
  ps_montmul<K, LB, 1, false, 0, UNITQ>(r, x, y, x, y, n, n1p, n0inv, none);
}

// a += k with full carry propagation (limbs may be lazy sums below 2^31); the value must stay < 2^(LB*K)
template <int K, int LB>
__device__ __forceinline__ void ps_add(uint32_t (&a)[K], const uint32_t (&k)[K]) {
  uint32_t c = 0;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const uint32_t u = a[j] + k[j] + c;
    a[j] = u & PsLimb<LB>::mask;
    c = u >> LB;
  }
}
// d = r - s (canonical limbs) modulo 2^(LB*K); returns the borrow (1: r < s)
template <int K, int LB>
__device__ __forceinline__ uint32_t ps_sub(uint32_t (&d)[K], const uint32_t (&r)[K], const uint32_t (&s)[K]) {
  uint32_t b = 0;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const uint32_t t = r[j] - s[j] - b;
    d[j] = t & PsLimb<LB>::mask;
    b = t >> 31;
  }
  return b;
}

// NI limbs of RB bits (lazy: below 2^31) -> K canonical limbs of LB bits; the value must fit
template <int K, int LB, int NI, int RB>
__device__ __forceinline__ void ps_relimb(uint32_t (&out)[K], uint32_t (&in)[NI]) {
  uint32_t c = 0;
#pragma unroll
  for (int j = 0; j < NI; ++j) {          // canonical RB-bit limbs first (the top one keeps what is left)
    const uint32_t u = in[j] + c;
    in[j] = j + 1 < NI ? u & ((1u << RB) - 1) : u;
    c = j + 1 < NI ? u >> RB : 0;
  }
  ps_static_for<K>([&](auto jc) __attribute__((always_inline)) {
    constexpr int j = decltype(jc)::value;
    constexpr int bit = j * LB, i0 = bit / RB, off = bit % RB;
    uint32_t v = 0;
    if constexpr (i0 < NI) v = in[i0] >> off;
    if constexpr (i0 + 1 < NI && off + LB > RB) v |= in[i0 + 1] << (RB - off);
    out[j] = v & PsLimb<LB>::mask;
  });
}

// 64-bit word w of the value whose canonical LB-bit limbs are v[0..K)
template <int K, int LB, int W>
__device__ __forceinline__ uint64_t ps_word(const uint32_t (&v)[K]) {
  uint64_t r = 0;
  ps_static_for<K>([&](auto ic) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    constexpr int pos = i * LB - 64 * W;      // bit position of limb i inside word W
    if constexpr (pos > -LB && pos < 64) {
      if constexpr (pos >= 0) r |= (uint64_t)v[i] << pos;
      else r |= (uint64_t)v[i] >> (-pos);
    }
  });
  return r;
}

// table entry e of this wavefront, part 0 (a) / 1 (b): K4 rows of 64 lanes x 16 bytes -- a wavefront reads and writes
// 1 KB runs (every lane of a wavefront takes the SAME entry: one side of the key, one exponent)
template <int K>
__device__ __forceinline__ size_t ps_table_row(int e, int part) {
  return ((size_t)e * 2 + part) * ((K + 3) / 4) * kWave;
}
template <int K>
__device__ __forceinline__ void ps_table_store(uint4* tw, int e, const uint32_t (&a)[K], const uint32_t (&b)[K]) {
  ps_park_store<K>(tw + ps_table_row<K>(e, 0), a);
  ps_park_store<K>(tw + ps_table_row<K>(e, 1), b);
}
// gather: every entry is read and the wanted one selected (the address stream does not depend on the digit)
template <int K>
__device__ __forceinline__ void ps_table_load(uint32_t (&a)[K], uint32_t (&b)[K], const uint4* tw, int idx, int tsize,
                                              bool gather) {
  if (!gather) {
    ps_park_load<K>(a, tw + ps_table_row<K>(idx, 0));
    ps_park_load<K>(b, tw + ps_table_row<K>(idx, 1));
    return;
  }
  // (every lane of the wavefront wants the SAME entry -- one side of the key, one exponent -- so the selection is a
  // v_cndmask_b32 under a wave-uniform condition per limb: half the instructions of an and / or pair.  Two entries per
  // trip: their loads are in flight together)
#pragma unroll
  for (int j = 0; j < K; ++j) a[j] = b[j] = 0;
#pragma unroll 1
  for (int e = 0; e < tsize; e += 2) {
    uint32_t ta[K], tb[K], ua[K], ub[K];
    const int e1 = e + 1 < tsize ? e + 1 : e;
    ps_park_load<K>(ta, tw + ps_table_row<K>(e, 0));
    ps_park_load<K>(tb, tw + ps_table_row<K>(e, 1));
    ps_park_load<K>(ua, tw + ps_table_row<K>(e1, 0));
    ps_park_load<K>(ub, tw + ps_table_row<K>(e1, 1));
    const bool s0 = e == idx, s1 = e1 == idx;
#pragma unroll
    for (int j = 0; j < K; ++j) {
      a[j] = s0 ? ta[j] : (s1 ? ua[j] : a[j]);
      b[j] = s0 ? tb[j] : (s1 ? ub[j] : b[j]);
    }
  }
}

// c*R modulo the side's P^2 as a pair (a, b) from the pair row of the n^2 domain (hensel_decrypt_kernel: the ct_pair entry):
// per chunk of the row a single product for its b half and a pair product for its a half.  ma / mb: scratch of the caller
// (they leave holding copies of a / b: the base of the window table).
template <int K, int LB>
__device__ __forceinline__ void ps_entry_from_pair_row(const HenselArgs& A, int side, size_t elem, const uint32_t (&n)[K],
                                                       uint32_t n1p, uint4* slot, uint32_t (&a)[K], uint32_t (&b)[K],
                                                       uint32_t (&ma)[K], uint32_t (&mb)[K]) {
  constexpr int RB = kLimbBits;
  constexpr int NI = (K * LB - 2) / RB + 1;      // row limbs per entry chunk that fit a half, plus one for the carry
#define HCTX(field) (side ? A.ctx[1].field : A.ctx[0].field)
  {
    const uint32_t* row = A.ct_pair + elem * A.ct_pair_stride;
    uint32_t acc_a[K], acc_b[K];
#pragma unroll
    for (int j = 0; j < K; ++j) acc_a[j] = acc_b[j] = 0;
#pragma unroll 1
    for (int i = 0; i < A.pchunks; ++i) {
      const int first = i * A.pchunk_limbs;
      uint32_t za[NI], zb[NI], cb[K], tb[K];
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const bool in = j < A.pchunk_limbs && first + j < A.pair_l2;
        za[j] = in ? row[first + j] : 0u;
        zb[j] = in ? row[A.pair_l2 + first + j] : 0u;
      }
#pragma unroll
      for (int j = 0; j < K; ++j) {
        b[j] = 0;
        cb[j] = HCTX(pcb)[(size_t)i * K + j];
        ma[j] = HCTX(pconv)[(size_t)i * 2 * K + j];
        mb[j] = HCTX(pconv)[(size_t)i * 2 * K + K + j];
      }
      // (rows written by the multi-lane kernels hold RELAXED limbs; ps_relimb makes them canonical in this kernel's width)
      uint32_t zl[K];
      ps_relimb<K, LB, NI, RB>(zl, zb);
      ps_mul<K, LB, true>(tb, zl, cb, n, n1p, 0);
      ps_relimb<K, LB, NI, RB>(a, za);
      ps_pairmul<K, LB, true>(a, b, ma, mb, n, n1p, 0, slot);
      ps_add<K, LB>(b, tb);
      ps_add<K, LB>(acc_a, a);
      ps_add<K, LB>(acc_b, b);
    }
#pragma unroll
    for (int j = 0; j < K; ++j) {
      a[j] = ma[j] = acc_a[j];
      b[j] = mb[j] = acc_b[j];
    }
  }
#undef HCTX
}

// Exit under the TRUE prime: (a, k*b mod p) times (hp, 0);  mp = ([a' >= p] - b') mod p, written as canonical words to row
// 2*elem + side of A.out (crt_kernel's input).  a, b: the pair of c^(p-1) R in canonical limbs; ma / mb: scratch.
template <int K, int LB>
__device__ __forceinline__ void ps_exit_words(const HenselArgs& A, int side, size_t elem, bool live, uint4* slot,
                                              uint32_t (&a)[K], uint32_t (&b)[K], uint32_t (&ma)[K], uint32_t (&mb)[K]) {
  constexpr int W64 = (K * LB + 63) / 64;
#define HCTX(field) (side ? A.ctx[1].field : A.ctx[0].field)
  uint32_t np[K];
#pragma unroll
  for (int j = 0; j < K; ++j) {
    np[j] = ps_uniform(HCTX(n)[j]);
    ma[j] = HCTX(kr)[j];
  }
  const uint32_t n0 = HCTX(n0inv);
  {
    uint32_t kb[K];
    ps_mul<K, LB, false>(kb, b, ma, np, 0, n0);
#pragma unroll
    for (int j = 0; j < K; ++j) {
      b[j] = kb[j];
      ma[j] = HCTX(h)[j];
      mb[j] = 0;
    }
  }
  ps_pairmul<K, LB, false>(a, b, ma, mb, np, 0, n0, slot);
  uint32_t d[K];
  const uint32_t below_a = ps_sub<K, LB>(d, a, np);
  const uint32_t jflag = below_a ^ 1u;
  const uint32_t below = ps_sub<K, LB>(d, b, np);
  if (!below) {
#pragma unroll
    for (int j = 0; j < K; ++j) b[j] = d[j];
  }
  (void)ps_sub<K, LB>(d, np, b);
  {
    uint32_t jf[K];
#pragma unroll
    for (int j = 0; j < K; ++j) jf[j] = j == 0 ? jflag : 0u;
    ps_add<K, LB>(d, jf);
  }
  const uint32_t small = ps_sub<K, LB>(b, d, np);
  if (small) {
#pragma unroll
    for (int j = 0; j < K; ++j) b[j] = d[j];
  }
  if (live) {
    uint64_t* out = A.out + (2 * elem + side) * A.out_stride;
    const int ow = A.out_words;
    ps_static_for<W64>([&](auto wc) __attribute__((always_inline)) {
      constexpr int ww = decltype(wc)::value;
      if (ww < ow) out[ww] = ps_word<K, LB, ww>(b);
    });
    for (int ww = W64; ww < ow; ++ww) out[ww] = 0;
  }
#undef HCTX
}

// One wavefront = 64 ciphertexts of ONE side (wave parity: even = p, odd = q).  Output: row 2i = mp, row 2i+1 = mq
// (canonical words) for crt_kernel, like hensel_decrypt_kernel.  K limbs of LB bits per half; the constants of A.ctx are
// in THAT limb width, the pair rows of A.ct_pair in the 29-bit limbs (kLimbBits) every other kernel writes.
// A.table: ps_table_words<K>(entries) 32-bit words per wavefront.
template <int K>
constexpr size_t ps_table_words(size_t entries) { return entries * 2 * ((K + 3) / 4) * kWave * 4; }

template <int K, int LB, int MINW>
__global__ __launch_bounds__(kWGThreads, MINW) void hensel_decrypt_ps_kernel(HenselArgs A) {
  constexpr int K4 = (K + 3) / 4;
  raise_wave_priority();
  __shared__ uint4 park_[kWavesPerWG][K4][kWave];
  const int lane = threadIdx.x % kWave, wv = threadIdx.x / kWave;
  uint4* slot = &park_[wv][0][lane];
  const size_t wave_id = (size_t)blockIdx.x * kWavesPerWG + wv;
  const int side = __builtin_amdgcn_readfirstlane((int)(wave_id & 1));
  const size_t first_elem = (wave_id >> 1) * kWave;
  size_t elem = first_elem + lane;
  if (elem >= A.count) elem = A.count - 1;
#define HCTX(field) (side ? A.ctx[1].field : A.ctx[0].field)
  uint32_t n[K], a[K], b[K], ma[K], mb[K];
#pragma unroll
  for (int j = 0; j < K; ++j) n[j] = ps_uniform(HCTX(nhat)[j]);   // wave-uniform: SGPR operands of the products
  const uint32_t n1p = n[1] + 1;
  const int w = A.window, tsize = 1 << w;
  uint4* tw = reinterpret_cast<uint4*>(A.table + wave_id * ps_table_words<K>((size_t)tsize)) + lane;
  const uint64_t* ep = A.exp + (size_t)side * A.exp_stride;
  const int nwin = (A.exp_bits + w - 1) / w;
  auto digit = [&](int i) -> int {
    int bit = i * w;
    int word = bit >> 6, sh = bit & 63;
    uint64_t v = (word < A.exp_words) ? ep[word] >> sh : 0;
    if (sh + w > 64 && word + 1 < A.exp_words) v |= ep[word + 1] << (64 - sh);
    return (int)(v & (uint64_t)(tsize - 1));
  };
  const bool gather = A.ct_gather != 0;

  // ---- c*R as a pair from the pair row of the n^2 domain (hensel_decrypt_kernel: the ct_pair entry) ----
  ps_entry_from_pair_row<K, LB>(A, side, elem, n, n1p, slot, a, b, ma, mb);
  // ---- window table: entry 0 = one, entry 1 = base, entry e = entry e-1 times base ----
  ps_table_store<K>(tw, 1, a, b);
  {
    uint32_t oa[K], ob[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
      oa[j] = HCTX(one)[j];
      ob[j] = HCTX(one)[K + j];
    }
    ps_table_store<K>(tw, 0, oa, ob);
  }
  // (the base comes back from entry 1 for every product, like the entries of the main loop: held in registers across
  // the loop it cost 92 scratch accesses per product)
#pragma unroll 1
  for (int e = 2; e < tsize; ++e) {
    ps_table_load<K>(ma, mb, tw, 1, tsize, false);
    ps_pairmul<K, LB, true>(a, b, ma, mb, n, n1p, 0, slot);
    ps_table_store<K>(tw, e, a, b);
  }
  // ---- main loop: w squarings, one multiplication by a table entry (always, also entry 0 = one) ----
  int win = nwin - 2;
  if (nwin > 0) {
    ps_table_load<K>(a, b, tw, digit(nwin - 1), tsize, gather);
  } else {
#pragma unroll
    for (int j = 0; j < K; ++j) {
      a[j] = HCTX(one)[j];
      b[j] = HCTX(one)[K + j];
    }
  }
#pragma unroll 1
  for (; nwin > 0 && win >= 0; --win) {
    const int idx = digit(win);
#pragma unroll 1
    for (int i = 0; i < w; ++i) ps_pairsqr<K, LB>(a, b, n, n1p);
    // (the entry is fetched AFTER the squarings: held across them it would cost 2K registers.  Fetching it before them in
    // the build that owns the whole register file, and re-aligning a workgroup's wavefronts at a barrier per window or per
    // squaring -- the unrolled code is larger than the instruction cache -- were both measured at +-0.3 %: DESIGN.md section 4)
    ps_table_load<K>(ma, mb, tw, idx, tsize, gather);
    ps_pairmul<K, LB, true>(a, b, ma, mb, n, n1p, 0, slot);
  }
  // ---- exit under the TRUE prime: (a, k*b mod p) times (hp, 0);  mp = ([a' >= p] - b') mod p ----
  ps_exit_words<K, LB>(A, side, elem, first_elem + lane < A.count, slot, a, b, ma, mb);
#undef HCTX
}

}  // namespace pgpu

#endif  // PAILLIERCRYPTOLIB_AMD_CSRC_HENSEL_PS_HPP_

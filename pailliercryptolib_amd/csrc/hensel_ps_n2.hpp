// pailliercryptolib_amd -- the one-lane product-scanning form for the n^2 domain (round 6):
// hensel_modexp_ps_kernel<K, LB>, base[i]^exp[i] modulo n^2 on resident pair rows -- CT x PT (CipherText * PlainText,
// ipcl/ciphertext.cpp:143-162) and, with one shared exponent, the non-DJN obfuscator r^n (ipcl/pub_key.cpp:66-80) -- with a
// WHOLE exponentiation in one lane.  2048-bit keys: K = 75 limbs of 28 bits per half (k_hensel.hip part 35), on the build
// that owns the whole register file (one wavefront per SIMD: 256 VGPRs + 256 AGPRs).
//
// Why: the multi-lane n^2 kernels (hensel_modexp_seq_kernel<4,18>, 72 limbs of 29 bits per half over 4 lanes) spend 20-23 %
// of their issue slots on moving limbs between lanes (DPP broadcasts, per-row carry adds): 0.63-0.71 of the int-ALU
// roofline.  Product scanning in one lane has no cross-lane traffic at all and its per-column support is O(K)
// against O(K^2) products: a pair squaring here is 19 654 products in 20 900 instructions (94 %), against 18 288 in ~23 300.
//
// The same residues as the rows hold -- no conversion products.  A pair row holds c*R as (a, b), c*R == a - P*b (mod n^2),
// P = n*k == -1 (mod 2^29), R = 2^(29*L2) (kargs.hpp).  P == -1 (mod 2^28) as well, so the SAME P is the loop modulus of a
// 28-bit product scan (unit quotient digits); only the radix differs: R' = 2^(28*K) = R * 2^s, s = 28*K - 29*L2 (2048-bit
// keys: 2100 - 2088 = 12).  A value x of the row domain is carried inside the kernel as x~ = x * 2^s: products stay
// consistent -- x~ y~ / R' = (x y / R) 2^s -- so entry is a left shift by s bits of both halves (fused into the re-limbing
// from 29- to 28-bit limbs; the pair is linear), and exit divides the pair by 2^s with two s-bit Montgomery digit steps
// (ps_pair_shift_out): ~300 instructions per exponentiation, no constants beyond P and the pair one.
// Bounds: a shifted entry value is < 4P * 2^s < 2^2091; the first product it enters leaves < 2^2082 + P, every later one
// < 2P; everything fits the 2100 bits of K limbs.  A column sums at most 3K products below 2^56 plus a 28-bit addend and
// a 36-bit carry: 225 * 2^56 < 2^64.
//
// Registers: a pair squaring keeps a, q, t hot (3K = 225) while b waits -- the compiler parks it in AGPRs by itself
// (tools/ubench_ps75.hip: 181 writes, 106 reads per squaring, no scratch).  The GENERAL product of the 38-limb kernel
// (one scan over a*d + b*c with all four operands hot) would need 5K = 375 hot registers: here it runs as three scans of
// at most 3K hot values (ps_pairmul_mem):  t = a*c reduced, digits q;  U = a*d + q unreduced (2K limbs);  b = (b*c + U)
// reduced -- the same 5K^2 - 2K products; of the values that wait, U sits in AGPRs by explicit v_accvgpr moves ("a" operands:
// the register allocator keeps them there), b and t in LDS (two K-limb slots per lane: 152 KB of a CU's 160 KB for the one
// workgroup that fits it), c and d are read again from the table row in memory.  (t in AGPRs as well -- 225 of the 256 --
// left the allocator 31 for its own parking: 193 scratch accesses per product.)
#ifndef PAILLIERCRYPTOLIB_AMD_CSRC_HENSEL_PS_N2_HPP_
#define PAILLIERCRYPTOLIB_AMD_CSRC_HENSEL_PS_N2_HPP_

#include "hensel_ps.hpp"

namespace pgpu {

// a 32-bit value parked in an accumulation register: one VALU move each way, no memory, no latency to hide
__device__ __forceinline__ uint32_t agpr_put(uint32_t v) {
  uint32_t a;
  asm("v_accvgpr_write_b32 %0, %1" : "=a"(a) : "v"(v));
  return a;
}
__device__ __forceinline__ uint32_t agpr_get(uint32_t a) {
  uint32_t v;
  asm("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a));
  return v;
}

// t = x*y*R'^-1 mod P with its quotient digits recorded in q; the limbs of t leave into this lane's LDS slot `ts` four at a
// time as they are born (16-byte rows of 64 lanes, like ps_park_store)
template <int K, int LB>
__device__ __forceinline__ void psn_mul_digits(uint4* ts, const uint32_t (&x)[K], const uint32_t (&y)[K],
                                               const uint32_t (&n)[K], uint32_t n1p, uint32_t (&q)[K]) {
  constexpr uint32_t M = PsLimb<LB>::mask;
  uint64_t acc = 0;
  uint32_t t4[4] = {0, 0, 0, 0};
  ps_static_for<2 * K>([&](auto colc) __attribute__((always_inline)) {
    constexpr int col = decltype(colc)::value;
    constexpr int ilo = col < K ? 0 : col - K + 1;
    {
      constexpr int ihi = col < K ? col : K;
      if constexpr (ihi > ilo) {
        ps_static_for<ihi - ilo>([&](auto ic) __attribute__((always_inline)) {
          constexpr int i = ilo + decltype(ic)::value;
          ps_mac(acc, q[i], col - i == 1 ? n1p : n[col - i]);
        });
      }
    }
    {
      constexpr int ihi = col < K ? col + 1 : K;
      ps_static_for<ihi - ilo>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = ilo + decltype(ic)::value;
        ps_mac(acc, x[i], y[col - i]);
      });
    }
    if constexpr (col < K) {
      q[col] = (uint32_t)acc & M;        // (P == -1 mod 2^LB: the digit is the low limb, its "+ q" rides on n1p)
      acc >>= LB;
    } else {
      constexpr int j = col - K;
      t4[j % 4] = (uint32_t)acc & M;
      acc >>= LB;
      if constexpr (j % 4 == 3 || j == K - 1) {
        uint4 wv;
        wv.x = t4[0];
        wv.y = j % 4 >= 1 ? t4[1] : 0u;
        wv.z = j % 4 >= 2 ? t4[2] : 0u;
        wv.w = j % 4 >= 3 ? t4[3] : 0u;
        ts[(j / 4) * kWave] = wv;
      }
    }
  });
}

// U = x*y + q as 2K canonical limbs, unreduced; every limb leaves into an AGPR as it is born
template <int K, int LB>
__device__ __forceinline__ void psn_mul_plain(uint32_t (&u_ag)[2 * K], const uint32_t (&x)[K], const uint32_t (&y)[K],
                                              const uint32_t (&q)[K]) {
  constexpr uint32_t M = PsLimb<LB>::mask;
  uint64_t acc = 0;
  uint32_t onev = 1;
  asm("" : "+v"(onev));
  ps_static_for<2 * K>([&](auto colc) __attribute__((always_inline)) {
    constexpr int col = decltype(colc)::value;
    constexpr int ilo = col < K ? 0 : col - K + 1;
    constexpr int ihi = col < K ? col + 1 : K;
    if constexpr (ihi > ilo) {
      ps_static_for<ihi - ilo>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = ilo + decltype(ic)::value;
        ps_mac(acc, x[i], y[col - i]);
      });
    }
    if constexpr (col < K) ps_mac(acc, q[col], onev);
    u_ag[col] = agpr_put((uint32_t)acc & M);
    acc >>= LB;
  });
}

// r = (x*y + U)*R'^-1 mod P, U from the AGPRs; r may be x (limb col-K of x is dead when column col delivers r[col-K])
template <int K, int LB>
__device__ __forceinline__ void psn_mul_add(uint32_t (&r)[K], const uint32_t (&x)[K], const uint32_t (&y)[K],
                                            const uint32_t (&n)[K], uint32_t n1p, const uint32_t (&u_ag)[2 * K]) {
  constexpr uint32_t M = PsLimb<LB>::mask;
  uint32_t q[K];
  uint64_t acc = 0;
  uint32_t onev = 1;
  asm("" : "+v"(onev));
  ps_static_for<2 * K>([&](auto colc) __attribute__((always_inline)) {
    constexpr int col = decltype(colc)::value;
    constexpr int ilo = col < K ? 0 : col - K + 1;
    {
      constexpr int ihi = col < K ? col : K;
      if constexpr (ihi > ilo) {
        ps_static_for<ihi - ilo>([&](auto ic) __attribute__((always_inline)) {
          constexpr int i = ilo + decltype(ic)::value;
          ps_mac(acc, q[i], col - i == 1 ? n1p : n[col - i]);
        });
      }
    }
    {
      constexpr int ihi = col < K ? col + 1 : K;
      if constexpr (ihi > ilo) {
        ps_static_for<ihi - ilo>([&](auto ic) __attribute__((always_inline)) {
          constexpr int i = ilo + decltype(ic)::value;
          ps_mac(acc, x[i], y[col - i]);
        });
      }
    }
    ps_mac(acc, agpr_get(u_ag[col]), onev);
    if constexpr (col < K) {
      q[col] = (uint32_t)acc & M;
      acc >>= LB;
    } else {
      r[col - K] = (uint32_t)acc & M;
      acc >>= LB;
    }
  });
}

// (a, b) = (a, b) (x) (c, d) with the multiplier pair in MEMORY (a table row: K4 rows of 64 lanes x 16 bytes, this lane's
// column; ps_park_load reads it).  slot, slot2: this lane's two K-limb parking slots in LDS (b waits in the first through
// the scans that do not need it, the new a part in the second from its birth to the end); U: 2K AGPRs.
template <int K, int LB>
__device__ __forceinline__ void ps_pairmul_mem(uint32_t (&a)[K], uint32_t (&b)[K], const uint4* ce, const uint4* de,
                                               const uint32_t (&n)[K], uint32_t n1p, uint4* slot, uint4* slot2) {
  static_assert(3 * (uint64_t)K * ((uint64_t)1 << (2 * LB - 32)) < ((uint64_t)1 << 32), "a column sums up to 3K products below 2^(2LB): must stay below 2^64");
  uint32_t u_ag[2 * K];
  {
    uint32_t q[K];
    ps_park_store<K>(slot, b);
    __builtin_amdgcn_sched_barrier(0);
    {
      uint32_t c[K];
      ps_park_load<K>(c, ce);
      psn_mul_digits<K, LB>(slot2, a, c, n, n1p, q);
    }
    __builtin_amdgcn_sched_barrier(0);
    {
      uint32_t d[K];
      ps_park_load<K>(d, de);
      psn_mul_plain<K, LB>(u_ag, a, d, q);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  {
    uint32_t c[K];
    ps_park_load<K>(b, slot);
    ps_park_load<K>(c, ce);
    psn_mul_add<K, LB>(b, b, c, n, n1p, u_ag);
  }
  __builtin_amdgcn_sched_barrier(0);
  ps_park_load<K>(a, slot2);
}

// NI relaxed limbs of RB bits (below 2^31) -> K canonical limbs of LB bits of the value SHIFTED LEFT by S bits
template <int K, int LB, int NI, int RB, int S>
__device__ __forceinline__ void psn_relimb_in(uint32_t (&out)[K], uint32_t (&in)[NI]) {
  uint32_t c = 0;
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const uint32_t u = in[j] + c;
    in[j] = j + 1 < NI ? u & ((1u << RB) - 1) : u;
    c = j + 1 < NI ? u >> RB : 0;
  }
  ps_static_for<K>([&](auto jc) __attribute__((always_inline)) {
    constexpr int j = decltype(jc)::value;
    constexpr int bit = j * LB - S;                  // bit of the (unshifted) value where limb j starts; may be negative
    uint32_t v = 0;
    if constexpr (bit + LB > 0) {
      constexpr int b0 = bit < 0 ? 0 : bit;
      constexpr int i0 = b0 / RB, off = b0 % RB, sh = bit < 0 ? -bit : 0;   // (bit < 0: the limb's low -bit bits are zero)
      if constexpr (i0 < NI) v = (in[i0] >> off) << sh;
      if constexpr (i0 + 1 < NI && off + LB - sh > RB) v |= in[i0 + 1] << (RB - off + sh);
    }
    out[j] = v & PsLimb<LB>::mask;
  });
}

// K canonical limbs of LB bits -> NO canonical limbs of RB bits (the value must fit)
template <int K, int LB, int NO, int RB>
__device__ __forceinline__ void psn_relimb_out(uint32_t (&out)[NO], const uint32_t (&in)[K]) {
  ps_static_for<NO>([&](auto jc) __attribute__((always_inline)) {
    constexpr int j = decltype(jc)::value;
    constexpr int bit = j * RB, i0 = bit / LB, off = bit % LB;
    uint32_t v = 0;
    if constexpr (i0 < K) v = in[i0] >> off;
    if constexpr (i0 + 1 < K && off + RB > LB) v |= in[i0 + 1] << (LB - off);
    if constexpr (i0 + 2 < K && off + RB > 2 * LB) v |= in[i0 + 2] << (2 * LB - off);
    out[j] = v & ((1u << RB) - 1);
  });
}

// x = (x + (x mod 2^S) * P) / 2^S, exact (P == -1 mod 2^S): one S-bit Montgomery digit step; returns the digit
template <int K, int LB, int S>
__device__ __forceinline__ uint32_t psn_digit_step(uint32_t (&x)[K], const uint32_t (&n)[K]) {
  static_assert(S > 0 && S < LB, "one partial digit");
  constexpr uint32_t M = PsLimb<LB>::mask;
  const uint32_t q = x[0] & ((1u << S) - 1);
  uint64_t acc = 0;
  uint32_t prev = 0;
  ps_static_for<K>([&](auto jc) __attribute__((always_inline)) {
    constexpr int j = decltype(jc)::value;
    acc += (uint64_t)q * n[j] + x[j];
    const uint32_t limb = (uint32_t)acc & M;
    acc >>= LB;
    if constexpr (j > 0) x[j - 1] = (prev >> S) | ((limb << (LB - S)) & M);
    prev = limb;
  });
  x[K - 1] = (prev >> S) | (((uint32_t)acc << (LB - S)) & M);
  return q;
}
// (a~, b~) = (a, b) * 2^S  ->  (a, b):  a = (a~ + q1 P) / 2^S,  b = (b~ + q1 + q2 P) / 2^S  (header: exit)
template <int K, int LB, int S>
__device__ __forceinline__ void ps_pair_shift_out(uint32_t (&a)[K], uint32_t (&b)[K], const uint32_t (&n)[K]) {
  const uint32_t q1 = psn_digit_step<K, LB, S>(a, n);
  uint32_t c = q1;
#pragma unroll
  for (int j = 0; j < K; ++j) {       // b += q1 (canonical limbs again)
    const uint32_t u = b[j] + c;
    b[j] = u & PsLimb<LB>::mask;
    c = u >> LB;
  }
  (void)psn_digit_step<K, LB, S>(b, n);
}

template <int K>
constexpr size_t psn_table_words(size_t entries) { return entries * 2 * ((K + 3) / 4) * kWave * 4; }

// One wavefront = 64 exponentiations, one per lane.  Pair rows of 2*L2 29-bit limbs in and out (A.base_pair, A.out_pair);
// A.ctx.nhat / A.ctx.one in 29-bit limbs (the key's pair form); per-element exponents (A.exp_stride > 0) or one shared one;
// fixed window A.window, table in A.table: psn_table_words<K>(2^w + 1) 32-bit words per wavefront (the last entry: the
// masked gather's selected row).
// Dynamic LDS: two parking slots per lane, 2 * sizeof(uint4) * kWavesPerWG * ceil(K/4) * 64 bytes (155 648 for K = 75).
template <int K, int LB, int L2>
__global__ __launch_bounds__(kWGThreads, 1) void hensel_modexp_ps_kernel(HenselModexpArgs A) {
  constexpr int K4 = (K + 3) / 4, RB = kLimbBits, S = LB * K - RB * L2;
  static_assert(S >= 0 && S < LB, "R' = R * 2^S with a partial digit");
  // (a shifted entry value -- below 8P * 2^S -- must fit K limbs: bits(n) + RB + 3 + S <= LB * K, checked by the host:
  // capi.cpp modexp_ps_form)
  raise_wave_priority();
  extern __shared__ uint4 psn_park_[];      // [2][kWavesPerWG][K4][kWave]
  const int lane = threadIdx.x % kWave, wv = threadIdx.x / kWave;
  uint4* slot = psn_park_ + (size_t)wv * K4 * kWave + lane;
  uint4* slot2 = slot + (size_t)kWavesPerWG * K4 * kWave;
  const size_t wave_id = (size_t)blockIdx.x * kWavesPerWG + wv;
  const size_t first = wave_id * kWave;
  size_t inst = first + lane;
  if (inst >= A.count) inst = A.count - 1;
  uint32_t n[K], a[K], b[K];
  {
    uint32_t p29[L2], pv[K];
#pragma unroll
    for (int j = 0; j < L2; ++j) p29[j] = A.ctx.nhat[j];
    psn_relimb_in<K, LB, L2, RB, 0>(pv, p29);
#pragma unroll
    for (int j = 0; j < K; ++j) n[j] = ps_uniform(pv[j]);       // wave-uniform: SGPR operands of the products
  }
  const uint32_t n1p = n[1] + 1;
  const int w = A.window, tsize = 1 << w;
  uint4* tw = reinterpret_cast<uint4*>(A.table + wave_id * psn_table_words<K>((size_t)tsize + 1)) + lane;
  const uint64_t* ep = A.exp + inst * A.exp_stride;
  const int nwin = (A.exp_bits + w - 1) / w;
  auto digit = [&](int i) -> int {
    int bit = i * w;
    int word = bit >> 6, sh = bit & 63;
    uint64_t v = (word < A.exp_words) ? ep[word] >> sh : 0;
    if (sh + w > 64 && word + 1 < A.exp_words) v |= ep[word + 1] << (64 - sh);
    return (int)(v & (uint64_t)(tsize - 1));
  };
  const bool gather = A.ct_gather != 0;
  // ---- entry: the row's pair, shifted into the kernel's radix; table entry 1 = base, entry 0 = one ----
  {
    const uint32_t* row = A.base_pair + inst * A.base_pair_stride;
    uint32_t ra[L2], rb[L2];
#pragma unroll
    for (int j = 0; j < L2; ++j) {
      ra[j] = row[j];
      rb[j] = row[L2 + j];
    }
    psn_relimb_in<K, LB, L2, RB, S>(a, ra);
    psn_relimb_in<K, LB, L2, RB, S>(b, rb);
    ps_table_store<K>(tw, 1, a, b);
    uint32_t oa[K], ob[K];
#pragma unroll
    for (int j = 0; j < L2; ++j) {
      ra[j] = A.ctx.one[j];
      rb[j] = A.ctx.one[L2 + j];
    }
    psn_relimb_in<K, LB, L2, RB, S>(oa, ra);
    psn_relimb_in<K, LB, L2, RB, S>(ob, rb);
    ps_table_store<K>(tw, 0, oa, ob);
  }
#pragma unroll 1
  for (int e = 2; e < tsize; ++e) {
    ps_pairmul_mem<K, LB>(a, b, tw + ps_table_row<K>(1, 0), tw + ps_table_row<K>(1, 1), n, n1p, slot, slot2);
    ps_table_store<K>(tw, e, a, b);
  }
  // ---- main loop: w squarings, one multiplication by a table entry (always, also entry 0 = one) ----
  int win = nwin - 2;
  ps_table_load<K>(a, b, tw, nwin > 0 ? digit(nwin - 1) : 0, tsize, gather && nwin > 0);
#pragma unroll 1
  for (; nwin > 0 && win >= 0; --win) {
    const int idx = digit(win);
#pragma unroll 1
    for (int i = 0; i < w; ++i) ps_pairsqr<K, LB>(a, b, n, n1p);
    int row = idx;
    if (gather) {
      // masked access: every entry is read and the wanted one selected into a scratch entry of this lane (entry tsize),
      // which the product then reads like any other
      uint32_t ma[K], mb[K];
      ps_table_load<K>(ma, mb, tw, idx, tsize, true);
      ps_table_store<K>(tw, tsize, ma, mb);
      row = tsize;
    }
    ps_pairmul_mem<K, LB>(a, b, tw + ps_table_row<K>(row, 0), tw + ps_table_row<K>(row, 1), n, n1p, slot, slot2);
  }
  // ---- exit: out of the kernel's radix, back to 29-bit limbs ----
  ps_pair_shift_out<K, LB, S>(a, b, n);
  if (first + lane < A.count) {
    uint32_t* out = A.out_pair + inst * (size_t)(2 * L2);
    uint32_t oa[L2], ob[L2];
    psn_relimb_out<K, LB, L2, RB>(oa, a);
    psn_relimb_out<K, LB, L2, RB>(ob, b);
#pragma unroll
    for (int j = 0; j < L2; ++j) {
      out[j] = oa[j];
      out[L2 + j] = ob[j];
    }
  }
}

}  // namespace pgpu

#endif  // PAILLIERCRYPTOLIB_AMD_CSRC_HENSEL_PS_N2_HPP_

// pailliercryptolib_amd -- CRT-decrypt exponentiation in split form with the two halves of a residue in DIFFERENT
// wavefronts (round 3).
//
// hensel_decrypt_kernel (hensel.hpp) keeps the pair x == a - P*b of one exponentiation in four lanes of one wavefront:
// two lanes hold a, two hold b, and both halves run ONE instruction stream.  Half A's work per squaring is an ordinary
// half-width Montgomery squaring -- K(K+1)/2 products per block thanks to the squaring symmetry plus the reduction --
// but it has to sit through the K^2 products of half B's 2*a*b, which has no symmetry: 12 % of the multiply-accumulate
// slots of the kernel compute products half A does not need, and in a general product half A idles through d*a.
// With 16384 exponentiations on 65536 lanes there is no lane layout that rebalances the halves inside one wavefront
// (DESIGN.md section 4), but with TWO batches in flight (the library's batch lanes) every SIMD hosts two wavefronts anyway
// -- so the halves may as well be two wavefronts with an instruction stream each:
//
//   wavefront A ("producer"): 32 exponentiations x 2 lanes, the a parts.  Per multiplication a half-width Montgomery
//     product modulo P (symmetric when it is a squaring), which also RECORDS its quotient digits; the digits and the
//     result go into a ring of slots in LDS.
//   wavefront B ("consumer"): the same 32 exponentiations x 2 lanes, the b parts.  Per multiplication
//     w = (a*d + b*c + q) / R mod P with q = A's digits of the same multiplication and a = A's result of the previous one,
//     both read from the ring.
//
// A runs ahead of B by up to kRing - 1 multiplications; the hand-over is two counters in LDS (produced / consumed),
// polled with s_sleep in between -- no s_barrier: the two wavefronts never wait for each other in lock step.  Per
// squaring and exponentiation that is 2 x 1475 + 2 x 1826 instead of 4 x 1851 instructions (-11 %), per general
// product -17 %, 12.4 % fewer instructions per exponentiation; results are the same pairs, bit for bit (the a part is
// the same half-width product, the b part the same sum).  The kernel takes ciphertexts as pair rows (the resident path)
// and the fixed-window scan; other inputs and the scheduled exponents stay with hensel_decrypt_kernel.
#ifndef PAILLIERCRYPTOLIB_AMD_CSRC_HENSEL_AB_HPP_
#define PAILLIERCRYPTOLIB_AMD_CSRC_HENSEL_AB_HPP_

#include "hensel_q.hpp"

namespace pgpu {


constexpr int kAbRing = 3;        // slots of the A -> B ring: A may run two multiplications ahead
constexpr int kAbIPW = 32;        // exponentiations per wavefront (two lanes each)

// LDS of one A/B pair
template <int K>
struct AbShared {
  uint32_t q[kAbRing][kAbIPW][2][kAbPad];   // quotient digits: [slot][exponentiation][block][digit]
  uint32_t a[kAbRing][kAbIPW][2][kAbPad];   // A's result: [slot][exponentiation][lane][limb]
  uint32_t produced, consumed;              // multiplications published by A / retired by B
  uint32_t jflag[kAbIPW];                   // exit: [a' >= p] per exponentiation
  uint32_t out[kAbIPW][2 * K];              // exit: canonical limbs of mp / mq for the word conversion
};

// The two counters of a pair live in LDS, and LDS executes one wavefront's instructions in order: the data stores of a
// multiplication are in the queue before the counter store that publishes them, and the reader's data loads are issued
// after the counter load has come back (the loop condition depends on it).  So no fence is needed -- and none is wanted:
// a release / acquire fence is an s_waitcnt vmcnt(0), which would also wait for the window-table entry that travels from
// HBM while the squarings run (first version of this kernel: +0.5 ms per launch).  Only the compiler has to keep the
// order (the empty asm statements).
// (The counters are read and written with explicit ds_ instructions on their LDS offset -- the low half of the generic
// address: through a volatile generic pointer hipcc emits flat_load + s_waitcnt vmcnt(0) lgkmcnt(0), the very wait this is
// meant to avoid.)
template <int SLEEP = 1>
__device__ __forceinline__ uint32_t ab_wait_ge(const uint32_t* flag, uint32_t want) {
  const uint32_t off = (uint32_t)(uintptr_t)flag;
  uint32_t seen;
  for (;;) {
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(off) : "memory");
    seen = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
    if (seen >= want) break;
    __builtin_amdgcn_s_sleep(SLEEP);
  }
  asm volatile("" ::: "memory");
  return seen;
}
// the counter read in two halves: issued now, looked at later (its round trip hides behind whatever is in between)
__device__ __forceinline__ uint32_t ab_peek_issue(const uint32_t* flag) {
  const uint32_t off = (uint32_t)(uintptr_t)flag;
  uint32_t v;
  asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(off) : "memory");
  return v;
}
__device__ __forceinline__ uint32_t ab_peek_value(uint32_t v) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v) : : "memory");
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ void ab_publish(uint32_t* flag, uint32_t value, int lane) {
  const uint32_t off = (uint32_t)(uintptr_t)flag;
  asm volatile("" ::: "memory");
  if (lane == 0) asm volatile("ds_write_b32 %0, %1" : : "v"(off), "v"(value) : "memory");
  asm volatile("" ::: "memory");
}


// Wavefront A: r = a * m * R^-1 mod P (m: the multiplier's limbs, lane-distributed like a; a squaring passes a itself),
// the digits of block S go to qslot[S][.] of this exponentiation.
template <int K, bool SQR, bool UNITQ, int S>
__device__ __forceinline__ void ab_a_blocks(uint64_t (&c0)[K], uint64_t (&c1)[K], const uint32_t (&a)[K],
                                            const uint32_t (&a2)[K], const uint32_t (&n)[K], uint32_t n0inv,
                                            const uint32_t (&m)[K], uint32_t* qslot, int x) {
  using HG = Geo<2, K>;
  if constexpr (S < 2) {
    uint32_t b[K], qrec[K];
#pragma unroll
    for (int r = 0; r < K; ++r) b[r] = bcast_lane<2, S>(m[r]);
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
        if (r + j < K) c0[r + j] += p;
        else c1[r + j - K] += p;
      }
    }
    mont_reduce_rows_q<HG, UNITQ, 1>(c0, c1, n, n0inv, qrec, 0u);
    ab_store20<K>(qslot + S * kAbPad, qrec);   // (both lanes of a group hold the same digits and store them to the same place)
    ab_a_blocks<K, SQR, UNITQ, S + 1>(c1, c0, a, a2, n, n0inv, m, qslot, x);
  }
}
template <int K, bool SQR, bool UNITQ>
__device__ __forceinline__ void ab_a_mul(uint32_t (&r)[K], const uint32_t (&a)[K], const uint32_t (&m)[K],
                                         const uint32_t (&n)[K], uint32_t n0inv, uint32_t* qslot, int x) {
  using HG = Geo<2, K>;
  uint64_t c0[K], c1[K];
  uint32_t a2[K];
#pragma unroll
  for (int j = 0; j < K; ++j) {
    c0[j] = 0;
    c1[j] = 0;
    a2[j] = SQR ? a[j] << 1 : 0;
  }
  ab_a_blocks<K, SQR, UNITQ, 0>(c0, c1, a, a2, n, n0inv, m, qslot, x);
  montmul_finish<HG>(r, c0);
}

// Wavefront B: r = (aprev * d + b * c + q) * R^-1 mod P.  b: this half's own limbs; d: the multiplier's b part; aprev: a
// copy of the a part of the own value; c: a copy of the a part of the multiplier (a squaring: c = aprev, d = b, so
// r = (2*b*aprev + q) / R).  qslot: A's digits of this multiplication.
template <int K, bool SQR, bool UNITQ, int S>
__device__ __forceinline__ void ab_b_blocks(uint64_t (&c0)[K], uint64_t (&c1)[K], const uint32_t (&mc)[K],
                                            const uint32_t (&md)[K], const uint32_t (&aprev)[K], const uint32_t (&c)[K],
                                            const uint32_t (&n)[K], uint32_t n0inv, const uint32_t* qslot, uint32_t sel0) {
  using HG = Geo<2, K>;
  if constexpr (S < 2) {
    uint32_t qin[K], row[K];
    ab_load20<K>(qin, qslot + S * kAbPad);
#pragma unroll
    for (int r = 0; r < K; ++r) row[r] = bcast_lane<2, S>(SQR ? aprev[r] : c[r]);
#pragma unroll
    for (int r = 0; r < K; ++r) {
#pragma unroll
      for (int j = 0; j < K; ++j) {
        const uint64_t p = (uint64_t)mc[j] * row[r];
        if (r + j < K) c0[r + j] += p;
        else c1[r + j - K] += p;
      }
    }
    if constexpr (!SQR) {
#pragma unroll
      for (int r = 0; r < K; ++r) row[r] = bcast_lane<2, S>(aprev[r]);
#pragma unroll
      for (int r = 0; r < K; ++r) {
#pragma unroll
        for (int j = 0; j < K; ++j) {
          const uint64_t p = (uint64_t)md[j] * row[r];
          if (r + j < K) c0[r + j] += p;
          else c1[r + j - K] += p;
        }
      }
    }
    mont_reduce_rows_q<HG, UNITQ, 2>(c0, c1, n, n0inv, qin, sel0);
    ab_b_blocks<K, SQR, UNITQ, S + 1>(c1, c0, mc, md, aprev, c, n, n0inv, qslot, sel0);
  }
}
template <int K, bool SQR, bool UNITQ>
__device__ __forceinline__ void ab_b_mul(uint32_t (&r)[K], const uint32_t (&b)[K], const uint32_t (&d)[K],
                                         const uint32_t (&aprev)[K], const uint32_t (&c)[K], const uint32_t (&n)[K],
                                         uint32_t n0inv, const uint32_t* qslot, uint32_t sel0) {
  static_assert(3 * K + 6 < 64, "a column receives 3K products (+ relaxed limbs): must stay below 2^64");
  using HG = Geo<2, K>;
  uint64_t c0[K], c1[K];
  uint32_t mc[K], md[K];
#pragma unroll
  for (int j = 0; j < K; ++j) {
    c0[j] = 0;
    c1[j] = 0;
    mc[j] = SQR ? b[j] << 1 : b[j];
    md[j] = SQR ? 0u : d[j];
  }
  ab_b_blocks<K, SQR, UNITQ, 0>(c0, c1, mc, md, aprev, c, n, n0inv, qslot, sel0);
  montmul_finish<HG>(r, c0);
}

// One workgroup = two A/B pairs (four wavefronts, one per SIMD of a CU; with a second batch in flight every SIMD hosts
// one wavefront of each launch).  Pair P serves 32 ciphertexts of ONE side (even pairs p, odd pairs q), like a wavefront
// of hensel_decrypt_kernel<2,K> serves 16.
// PAIRS = 2: four wavefronts, one per SIMD (A, A, B, B) -- the form for a launch that has the chip to itself.
// PAIRS = 4: eight wavefronts, two per SIMD; wavefronts w and w + 4 share a SIMD (MI355X_MICROARCH.md: a 512-thread
// workgroup places two waves on each SIMD), so every SIMD hosts the A wavefront of one pair and the B wavefront of
// another BY CONSTRUCTION -- the form for launches of 16384 ciphertexts and more (two per SIMD anyway) and for a batch
// that shares the GPU with a second one (each launch then covers half the CUs).
template <int K, int PAIRS>
__global__ __launch_bounds__(PAIRS * 2 * kWave, 2) void hensel_decrypt_ab_kernel(HenselArgs A) {
  using HG = Geo<2, K>;
  constexpr int L2 = 2 * K, LQ = 4 * K, IPW = kAbIPW, W64 = HG::W64;
  raise_wave_priority();
  __shared__ AbShared<K> sh_[PAIRS];
  const int lane = threadIdx.x % kWave, wv = threadIdx.x / kWave;
  const int pr = wv % PAIRS;
  const bool roleB = wv >= PAIRS;
  AbShared<K>& sh = sh_[pr];
  const int grp = lane / 2, x = lane % 2;
  uint32_t sel0 = x == 0 ? 1u : 0u;
  asm("" : "+v"(sel0));
  const size_t pair_id = (size_t)blockIdx.x * PAIRS + pr;
  const int side = __builtin_amdgcn_readfirstlane((int)(pair_id & 1));
  const size_t first_elem = (pair_id >> 1) * IPW;
  size_t elem = first_elem + grp;
  if (elem >= A.count) elem = A.count - 1;
#define HCTX(field) (side ? A.ctx[1].field : A.ctx[0].field)
  if (!roleB && lane == 0) {
    sh.produced = 0;
    sh.consumed = 0;
  }
  __syncthreads();

  uint32_t n[K];
#pragma unroll
  for (int j = 0; j < K; ++j) n[j] = HCTX(nhat)[x * K + j];
  const int w = A.window, tsize = 1 << w;
  // window table of this exponentiation: entry e = 4K limbs, the a part (2 lanes x K, written by A) then the b part (by B)
  uint32_t* tbl = A.table + (pair_id * IPW + grp) * (size_t)tsize * LQ + (roleB ? L2 : 0) + x * K;
  const uint32_t* tbl_a = A.table + (pair_id * IPW + grp) * (size_t)tsize * LQ + x * K;   // (B reads a parts too)
  const uint64_t* ep = A.exp + (size_t)side * A.exp_stride;
  const int nwin = (A.exp_bits + w - 1) / w;
  auto digit = [&](int i) -> int {
    int bit = i * w;
    int word = bit >> 6, sh_ = bit & 63;
    uint64_t v = (word < A.exp_words) ? ep[word] >> sh_ : 0;
    if (sh_ + w > 64 && word + 1 < A.exp_words) v |= ep[word + 1] << (64 - sh_);
    return (int)(v & (uint64_t)(tsize - 1));
  };
  const uint32_t* row = A.ct_pair + elem * A.ct_pair_stride;
  uint32_t m = 0;   // multiplications so far (the same count in both wavefronts)
  auto qslot = [&](uint32_t mm) -> uint32_t* { return &sh.q[mm % kAbRing][grp][0][0]; };
  auto aslot = [&](uint32_t mm) -> uint32_t* { return &sh.a[mm % kAbRing][grp][x][0]; };
  const bool gather = A.ct_gather != 0;

  if (!roleB) {
    // ============================ wavefront A: the a parts ============================
    uint32_t own[K], mreg[K], acc[K];
    // one multiplication: wait for a free slot, multiply, publish result + digits
#define AB_A_STEP(SQ, UQ, RES, LHS, RHS, N0, PUBLISH_VALUE)                                    \
  do {                                                                                          \
    if (m >= (uint32_t)kAbRing) ab_wait_ge<8>(&sh.consumed, m - (uint32_t)kAbRing + 1);            \
    ab_a_mul<K, SQ, UQ>(RES, LHS, RHS, n, N0, qslot(m), x);                                     \
    ab_store20<K>(aslot(m), PUBLISH_VALUE);                                                     \
    ++m;                                                                                        \
    ab_publish(&sh.produced, m, lane);                                                          \
  } while (0)
#pragma unroll
    for (int j = 0; j < K; ++j) acc[j] = 0;
#pragma unroll 1
    for (int i = 0; i < A.pchunks; ++i) {
      const int first = i * A.pchunk_limbs;
#pragma unroll
      for (int j = 0; j < K; ++j) {
        const int li = x * K + j;
        own[j] = (li < A.pchunk_limbs && first + li < A.pair_l2) ? row[first + li] : 0u;
        mreg[j] = HCTX(pconv)[(size_t)i * LQ + x * K + j];
      }
      if (m >= (uint32_t)kAbRing) ab_wait_ge<8>(&sh.consumed, m - (uint32_t)kAbRing + 1);
      ab_a_mul<K, false, true>(own, own, mreg, n, 0, qslot(m), x);
      add_normalise<HG>(acc, own);
      ab_store20<K>(aslot(m), acc);        // B's copy of the a part is the running sum
      ++m;
      ab_publish(&sh.produced, m, lane);
    }
    // table: entry 0 = one, entry 1 = base, entry e = entry e-1 times base
#pragma unroll
    for (int j = 0; j < K; ++j) {
      tbl[(size_t)LQ + j] = acc[j];
      tbl[j] = HCTX(one)[x * K + j];
      mreg[j] = own[j] = acc[j];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    // (B reads the a parts of table entries from memory: an entry is stored BEFORE the product that made it is published,
    // with a release fence in between -- here, and only here, the global stores of this wavefront matter to B)
#pragma unroll 1
    for (int e = 2; e < tsize; ++e) {
      if (m >= (uint32_t)kAbRing) ab_wait_ge<8>(&sh.consumed, m - (uint32_t)kAbRing + 1);
      ab_a_mul<K, false, true>(own, own, mreg, n, 0, qslot(m), x);
#pragma unroll
      for (int j = 0; j < K; ++j) tbl[(size_t)e * LQ + j] = own[j];
      ab_store20<K>(aslot(m), own);
      ++m;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // (table build only: the entry's global stores have landed)
      ab_publish(&sh.produced, m, lane);
    }
    int win = nwin - 2;
    if (nwin > 0) {
      load_table_entry<K>(own, tbl, digit(nwin - 1), tsize, LQ, gather);
    } else {
#pragma unroll
      for (int j = 0; j < K; ++j) own[j] = HCTX(one)[x * K + j];
    }
#pragma unroll 1
    for (; nwin > 0 && win >= 0; --win) {
      load_table_entry<K>(mreg, tbl, digit(win), tsize, LQ, gather);
#pragma unroll 1
      for (int i = 0; i < w; ++i) AB_A_STEP(true, true, own, own, own, 0, own);
      AB_A_STEP(false, true, own, own, mreg, 0, own);
    }
    // exit under the true prime: times (hp, 0)
    uint32_t np[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
      np[j] = HCTX(n)[x * K + j];
      mreg[j] = HCTX(h)[x * K + j];
    }
    {
      const uint32_t n0 = HCTX(n0inv);
      if (m >= (uint32_t)kAbRing) ab_wait_ge<8>(&sh.consumed, m - (uint32_t)kAbRing + 1);
      // (n[] is the loop modulus; the exit product reduces modulo p)
      uint32_t res[K];
      {
        using HGL = Geo<2, K>;
        uint64_t c0[K], c1[K];
        uint32_t a2[K];
#pragma unroll
        for (int j = 0; j < K; ++j) {
          c0[j] = 0;
          c1[j] = 0;
          a2[j] = 0;
        }
        ab_a_blocks<K, false, false, 0>(c0, c1, own, a2, np, n0, mreg, qslot(m), x);
        montmul_finish<HGL>(res, c0);
      }
      // a' in {hp, hp + p}: tell B whether a' >= p
      full_normalise<HG>(res, x);
      uint32_t d[K];
      const uint32_t below = sub_limbs<HG>(d, res, np, x, lane);
      if (x == 0) sh.jflag[grp] = below ^ 1u;
      ab_store20<K>(aslot(m), res);
      ++m;
      ab_publish(&sh.produced, m, lane);
    }
#undef AB_A_STEP
    return;
  }

  // ============================ wavefront B: the b parts ============================
  uint32_t own[K], acur[K], mc_[K], md_[K], acc[K];
  // one multiplication: wait for A's digits, multiply, take over A's result as the new a copy, retire the slot
  // (A is usually two multiplications ahead: `seen`, the counter as read at the end of the previous step, then already
  // covers this one and the poll costs nothing; A's result of THIS multiplication is fetched before the product starts
  // and becomes the a copy of the next one, its LDS round trip hidden behind the product)
  uint32_t seen = 0;
#define AB_B_STEP(SQ, UQ, CVAL, DVAL, N, N0)                                                   \
  do {                                                                                          \
    if (seen < m + 1) seen = ab_wait_ge(&sh.produced, m + 1);                                   \
    uint32_t anext[K];                                                                          \
    ab_load20<K>(anext, aslot(m));                                                              \
    const uint32_t peek = ab_peek_issue(&sh.produced);                                          \
    ab_b_mul<K, SQ, UQ>(own, own, DVAL, acur, CVAL, N, N0, qslot(m), sel0);                     \
    _Pragma("unroll") for (int j = 0; j < K; ++j) acur[j] = anext[j];                           \
    seen = ab_peek_value(peek);                                                                 \
    ++m;                                                                                        \
    ab_publish(&sh.consumed, m, lane);                                                          \
  } while (0)
#pragma unroll
  for (int j = 0; j < K; ++j) acc[j] = 0;
  {
    const uint32_t* rowb = row + A.pair_l2;
#pragma unroll 1
    for (int i = 0; i < A.pchunks; ++i) {
      const int first = i * A.pchunk_limbs;
      uint32_t zb[K], cb[K], tb[K];
#pragma unroll
      for (int j = 0; j < K; ++j) {
        const int li = x * K + j;
        const bool in = li < A.pchunk_limbs && first + li < A.pair_l2;
        acur[j] = in ? row[first + li] : 0u;                 // the a part of (z_i, 0)
        zb[j] = in ? rowb[first + li] : 0u;                  // b_i
        cb[j] = HCTX(pcb)[(size_t)i * L2 + x * K + j];
        mc_[j] = HCTX(pconv)[(size_t)i * LQ + x * K + j];        // c: a part of pconv_i
        md_[j] = HCTX(pconv)[(size_t)i * LQ + L2 + x * K + j];   // d: b part of pconv_i
        own[j] = 0;                                              // the b part of (z_i, 0)
      }
      montmul_reg<HG, false, true>(tb, zb, cb, n, 0);
      seen = ab_wait_ge(&sh.produced, m + 1);
      ab_b_mul<K, false, true>(own, own, md_, acur, mc_, n, 0, qslot(m), sel0);
#pragma unroll
      for (int j = 0; j < K; ++j) own[j] += tb[j];
      add_normalise<HG>(acc, own);
      ab_load20<K>(acur, aslot(m));     // the a part of the running sum
      ++m;
      ab_publish(&sh.consumed, m, lane);
    }
  }
#pragma unroll
  for (int j = 0; j < K; ++j) {
    tbl[(size_t)LQ + j] = acc[j];
    tbl[j] = HCTX(one)[L2 + x * K + j];
    own[j] = md_[j] = acc[j];       // base: d = its b part ...
    mc_[j] = acur[j];               // ... c = its a part
  }
#pragma unroll 1
  for (int e = 2; e < tsize; ++e) {
    AB_B_STEP(false, true, mc_, md_, n, 0);
#pragma unroll
    for (int j = 0; j < K; ++j) tbl[(size_t)e * LQ + j] = own[j];
  }
  int win = nwin - 2;
  if (nwin > 0) {
    // A has published its last table product (produced >= m), so every a part of the table is in memory
    const int d0 = digit(nwin - 1);
    load_table_entry<K>(own, tbl, d0, tsize, LQ, gather);
    load_table_entry<K>(acur, tbl_a, d0, tsize, LQ, gather);
  } else {
#pragma unroll
    for (int j = 0; j < K; ++j) {
      own[j] = HCTX(one)[L2 + x * K + j];
      acur[j] = HCTX(one)[x * K + j];
    }
  }
#pragma unroll 1
  for (; nwin > 0 && win >= 0; --win) {
    const int idx = digit(win);
    load_table_entry<K>(md_, tbl, idx, tsize, LQ, gather);
    load_table_entry<K>(mc_, tbl_a, idx, tsize, LQ, gather);
#pragma unroll 1
    for (int i = 0; i < w; ++i) AB_B_STEP(true, true, acur, own, n, 0);
    AB_B_STEP(false, true, mc_, md_, n, 0);
  }
  // exit under the true prime: (a, k*b mod p) times (hp, 0)
  uint32_t np[K];
#pragma unroll
  for (int j = 0; j < K; ++j) {
    np[j] = HCTX(n)[x * K + j];
    mc_[j] = HCTX(kr)[x * K + j];
  }
  const uint32_t n0 = HCTX(n0inv);
  {
    uint32_t kb[K];
    montmul_reg<HG, false, false>(kb, own, mc_, np, n0);
#pragma unroll
    for (int j = 0; j < K; ++j) {
      own[j] = kb[j];
      mc_[j] = HCTX(h)[x * K + j];   // c = hp
      md_[j] = 0;                    // d = 0
    }
  }
  AB_B_STEP(false, false, mc_, md_, np, n0);
#undef AB_B_STEP
  // half B: b' < 2p;  mp = ([a' >= p] - b') mod p   (as in hensel_decrypt_kernel)
  const uint32_t jflag = sh.jflag[grp];
  full_normalise<HG>(own, x);
  uint32_t d[K];
  const uint32_t below = sub_limbs<HG>(d, own, np, x, lane);
  if (!below) {
#pragma unroll
    for (int j = 0; j < K; ++j) own[j] = d[j];
  }
  (void)sub_limbs<HG>(d, np, own, x, lane);
  if (x == 0) d[0] += jflag;
  full_normalise<HG>(d, x);
  const uint32_t small = sub_limbs<HG>(own, d, np, x, lane);
  if (small) {
#pragma unroll
    for (int j = 0; j < K; ++j) own[j] = d[j];
  }
  wave_lds_sync();
#pragma unroll
  for (int j = 0; j < K; ++j) sh.out[grp][x * K + j] = own[j];
  wave_lds_sync();
  const int ow = A.out_words;
  for (int t = lane; t < IPW * ow; t += kWave) {
    const int gg = t / ow, ww = t % ow;
    const size_t oe = first_elem + gg;
    if (oe < A.count) A.out[(2 * oe + side) * A.out_stride + ww] = ww < W64 ? word_from_limbs(sh.out[gg], L2, ww) : 0;
  }
#undef HCTX
}

}  // namespace pgpu

#endif  // PAILLIERCRYPTOLIB_AMD_CSRC_HENSEL_AB_HPP_

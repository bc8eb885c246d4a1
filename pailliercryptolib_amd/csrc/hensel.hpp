// pailliercryptolib_amd -- exponentiation modulo a SQUARE (p^2, q^2, n^2) in split form.
//
// Every modulus of the Paillier path is a square whose root the caller knows: p^2 / q^2 in
// PrivateKey::decryptCRT (ipcl/pri_key.cpp:122-157; the reference hands c^(p-1) mod p^2 to ippMBModExp as a
// 2048-bit black box) and n^2 in PublicKey::encrypt and CipherText * PlainText (pub_key.cpp:51-105,
// ciphertext.cpp:143-162).  Kernels here: hensel_decrypt_kernel (replaces the two half-width exponentiations of
// decryptCRT and the L-function / *hp step behind them), hensel_fb_build_kernel / hensel_fb_encrypt_kernel (DJN
// encrypt with a fixed-base table), hensel_modexp_kernel (per-element bases: CT x PT, the non-DJN obfuscator).
// A residue modulo P^2 is kept as a PAIR of half-width numbers
//        x  ==  a - P*b   (mod P^2),      0 <= a, b < 2P  (lazy),
// and the Montgomery product of two pairs costs three half-width products and two half-width reductions
// instead of one full-width product and one full-width reduction (5 s^2 instead of 8 s^2 limb products; a
// squaring 4 s^2 -- here: 4 K^2 per lane -- instead of 6 s^2):
//        a*c = t*R - q*P            (Montgomery reduction modulo P: t = (a*c + q*P)/R, q = the quotient digits)
//   =>   (a - P*b)(c - P*d) == t*R - P*(a*d + b*c + q)            (mod P^2)
//   =>   x*y*R^-1 == t - P*w,   w = (a*d + b*c + q) * R^-1 mod P  (a second Montgomery reduction modulo P)
// -- the quotient digits of the first reduction are exactly the correction the second one needs, and with the
// negated coefficient (a - P*b rather than a + P*b) every term is added, never subtracted.  R = 2^(29*2K) is the
// Montgomery radix of the HALF width.  tests/test_hensel_model.py restates this with Python integers.
//
// Lanes: a group of 2H lanes holds one exponentiation (H = 2: a quad; 4; 8: a DPP row); its lanes 0..H-1
// ("half A") hold a, lanes H..2H-1 ("half B") hold b, K 29-bit limbs per lane: two Geo<H,K> groups side by side
// that run ONE instruction stream.  The multiplier rows of both halves are limbs of half A of an operand (c, then
// a), broadcast to the whole group by one v_mov_b32_dpp each (quad_perm:[S,S,S,S] / row_newbcast:S; H = 4: from a
// per-quad copy, pair_row_source); half A accumulates a*c while half B accumulates b*c + d*a; the K-row
// reduction blocks of mont_core.hpp run in both halves at once, half B's low lane receiving half A's digit first
// (mont_reduce_rows<.., PAIR>).  The loop modulus is P = p*k == -1 (mod 2^29) (unit quotient digits); the
// multiplication that leaves the Montgomery domain switches to the true prime: (a, k*b mod p) is a pair modulo p^2.
// With u = c^(p-1) == 1 (mod p) that last product, by (hp, 0), leaves  a' in {hp, hp + p}  and the plaintext half
//        mp = L_p(u) * hp mod p  =  ([a' >= p] - b') mod p
// directly -- the L function costs a comparison, and crt_kernel receives mp and mq instead of u*hp mod p^2.
#ifndef PAILLIERCRYPTOLIB_AMD_CSRC_HENSEL_HPP_
#define PAILLIERCRYPTOLIB_AMD_CSRC_HENSEL_HPP_

#include "kernels.hpp"

namespace pgpu {

// Row source of a pair product: the limbs of half A of an operand, readable by ONE broadcast per row in both halves.
// H = 2 (quad_perm:[S,S,S,S]) and H = 8 (row_newbcast:S) broadcast straight from half A; an 8-lane group would need
// two bank-masked moves per row, so for H = 4 the upper quad first takes a copy of the lower quad's limbs (one
// row_shr:4 move per limb and multiplication) and every quad broadcasts from its own lane S.
template <int H, int K>
__device__ __forceinline__ void pair_row_source(uint32_t (&src)[K], const uint32_t (&v)[K]) {
#pragma unroll
  for (int j = 0; j < K; ++j) {
    if constexpr (H == 4)
      src[j] = (uint32_t)__builtin_amdgcn_update_dpp((int)v[j], (int)v[j], 0x114 /*row_shr:4*/, 0xf, 0xa, false);
    else
      src[j] = v[j];
  }
}
template <int H, int S>
__device__ __forceinline__ uint32_t pair_row(uint32_t src) {
  if constexpr (H == 8) return bcast_lane<16, S>(src);
  else return bcast_lane<4, S>(src);
}

// One K-row block of a pair product.  mc: this lane's multiplicand limbs for the rows of csrc (half A: a,
// half B: b; a squaring doubles half B's); md / asrc: the second product of half B (d times the rows of a; md is
// zero in half A, unused in a squaring).
template <int H, int K, bool SQR, bool UNITQ, int S>
__device__ __forceinline__ void pair_block(uint64_t (&LOWC)[K], uint64_t (&UPC)[K], const uint32_t (&mc)[K],
                                           const uint32_t (&md)[K], const uint32_t (&csrc)[K],
                                           const uint32_t (&asrc)[K], const uint32_t (&n)[K], uint32_t n0inv,
                                           uint32_t selB) {
  using GEO = Geo<H, K>;
  uint32_t crow[K];
#pragma unroll
  for (int r = 0; r < K; ++r) crow[r] = pair_row<H, S>(csrc[r]);
#pragma unroll
  for (int r = 0; r < K; ++r) {
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const uint64_t p = (uint64_t)mc[j] * crow[r];
      if (r + j < K) LOWC[r + j] += p;
      else UPC[r + j - K] += p;
    }
  }
  if constexpr (!SQR) {
    uint32_t arow[K];
#pragma unroll
    for (int r = 0; r < K; ++r) arow[r] = pair_row<H, S>(asrc[r]);
#pragma unroll
    for (int r = 0; r < K; ++r) {
#pragma unroll
      for (int j = 0; j < K; ++j) {
        const uint64_t p = (uint64_t)md[j] * arow[r];
        if (r + j < K) LOWC[r + j] += p;
        else UPC[r + j - K] += p;
      }
    }
  }
  mont_reduce_rows<GEO, UNITQ, true, true>(LOWC, UPC, n, n0inv, selB);
}

// r = own (x) m: the Montgomery product of two pairs (lazy: inputs < 8P -> outputs < 2P).  own, m, r: this lane's
// K limbs (half A: the a part, half B: the b part).  halfB: 1 in the upper H lanes; selB: 1 in lane H of the group.
template <int H, int K, bool SQR, bool UNITQ, int S>
__device__ __forceinline__ void pair_blocks(uint64_t (&c0)[K], uint64_t (&c1)[K], const uint32_t (&mc)[K],
                                            const uint32_t (&md)[K], const uint32_t (&csrc)[K],
                                            const uint32_t (&asrc)[K], const uint32_t (&n)[K], uint32_t n0inv,
                                            uint32_t selB) {
  if constexpr (S < H) {
    pair_block<H, K, SQR, UNITQ, S>(c0, c1, mc, md, csrc, asrc, n, n0inv, selB);
    pair_block<H, K, SQR, UNITQ, S + 1>(c1, c0, mc, md, csrc, asrc, n, n0inv, selB);
    pair_blocks<H, K, SQR, UNITQ, S + 2>(c0, c1, mc, md, csrc, asrc, n, n0inv, selB);
  }
}
template <int H, int K, bool SQR, bool UNITQ>
__device__ __forceinline__ void pairmul(uint32_t (&r)[K], const uint32_t (&own)[K], const uint32_t (&m)[K],
                                        const uint32_t (&n)[K], uint32_t n0inv, uint32_t halfB, uint32_t selB) {
  static_assert(3 * K + 6 < 64, "a column of half B receives 3K products (+ relaxed limbs): must stay below 2^64");
  static_assert(H == 2 || H == 4 || H == 8, "two halves inside one 16-lane DPP row");
  using GEO = Geo<H, K>;
  uint64_t c0[K], c1[K];
  uint32_t mc[K], md[K], csrc[K], asrc[K];
  const uint32_t maskB = 0u - halfB;
  pair_row_source<H, K>(asrc, own);                     // rows of a (a squaring: the only rows)
  if constexpr (!SQR) pair_row_source<H, K>(csrc, m);   // rows of c
#pragma unroll
  for (int j = 0; j < K; ++j) {
    c0[j] = 0;
    c1[j] = 0;
    mc[j] = SQR ? own[j] << halfB : own[j];   // squaring: half B accumulates 2*a*b
    md[j] = SQR ? 0 : m[j] & maskB;
  }
  if constexpr (SQR) pair_blocks<H, K, SQR, UNITQ, 0>(c0, c1, mc, md, asrc, asrc, n, n0inv, selB);
  else pair_blocks<H, K, SQR, UNITQ, 0>(c0, c1, mc, md, csrc, asrc, n, n0inv, selB);
  montmul_finish<GEO>(r, c0);
}

// d = r - s limb-wise (canonical limbs in, canonical limbs out, modulo 2^(29*L)); returns the final borrow
// (1: r < s), known to every lane of the group.
template <class GEO>
__device__ __forceinline__ uint32_t sub_limbs(uint32_t (&d)[GEO::K], const uint32_t (&r)[GEO::K],
                                              const uint32_t (&s)[GEO::K], int x, int lane) {
  constexpr int K = GEO::K, G = GEO::G;
  const int top_lane = (lane / G) * G + (G - 1);
  uint32_t b = 0;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    uint32_t t = r[j] - s[j] - b;
    d[j] = t & kLimbMask;
    b = t >> 31;
  }
  uint32_t top_borrow = b;
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
  return (uint32_t)__shfl((int)top_borrow, top_lane);
}

// One wavefront = 64/(2H) groups = that many ciphertexts of ONE side (wave parity: even = p, odd = q), so context,
// exponent and schedule are wave-uniform.  Output: row 2i = mp, row 2i+1 = mq (canonical words) for crt_kernel
// (have_m).  H = 2: the throughput form (16 ciphertexts per wavefront); H = 8: the latency form for small batches
// (the serial chain of a multiplication is its H*K quotient rows, and a row shrinks with K).
// MINW: wavefronts per SIMD the register budget is set for.  Up to 14 limbs per lane the kernel fits the 256
// registers of two wavefronts anyway; (2,19) is compiled both ways: with the full budget for launches of at most one
// wavefront per SIMD (no scratch at all) and squeezed into 256 registers (a few set-up values in scratch, none in the
// loops) for larger batches, where the second wavefront buys 7 % (32768 ciphertexts: 4.76 -> 4.41 ms per 8192).
template <int H, int K, int MINW = (K <= 14 ? 2 : 1)>
__global__ __launch_bounds__(kWGThreads, MINW) void hensel_decrypt_kernel(HenselArgs A) {
  using HG = Geo<H, K>;
  constexpr int GS = 2 * H, IPW = kWave / GS, L2 = H * K, LQ = 2 * H * K, W64 = HG::W64;
  raise_wave_priority();
  __shared__ uint32_t bl_[kWavesPerWG][IPW][L2];
  __shared__ uint64_t io_[kWavesPerWG][IPW][W64 + 1];
  const int lane = threadIdx.x % kWave, wv = threadIdx.x / kWave;
  auto& bl = bl_[wv];
  auto& io = io_[wv];
  const int grp = lane / GS, xg = lane % GS, x = xg % H;   // group (one exponentiation), lane in the group, lane in its half
  const uint32_t halfB = (uint32_t)(xg / H);
  uint32_t selB = xg == H ? 1u : 0u;
  asm("" : "+v"(selB));   // opaque, so that "digit * selB" stays ONE v_mad_u64_u32 (not a select and a 64-bit add)
  const size_t wave_id = (size_t)blockIdx.x * kWavesPerWG + wv;
  const int side = __builtin_amdgcn_readfirstlane((int)(wave_id & 1));
  const size_t first_elem = (wave_id >> 1) * IPW;
  size_t elem = first_elem + grp;
  if (elem >= A.count) elem = A.count - 1;   // padded groups recompute the last element
#define HCTX(field) (side ? A.ctx[1].field : A.ctx[0].field)

  uint32_t n[K], own[K], mreg[K];
#pragma unroll
  for (int j = 0; j < K; ++j) n[j] = HCTX(nhat)[x * K + j];
  uint32_t n0inv = 0;   // unit quotient digits until the exit multiplication

  const bool sched_mode = A.sched[0] != nullptr;
  const uint16_t* sch = side ? A.sched[1] : A.sched[0];
  const int nsteps = side ? A.sched_len[1] : A.sched_len[0];
  const int w = A.window;
  const int tsize = sched_mode ? 1 << (w - 1) : 1 << w;
  uint32_t* tbl = A.table + (wave_id * IPW + grp) * (size_t)tsize * LQ + xg * K;
  const uint64_t* ep = A.exp + (size_t)side * A.exp_stride;
  const int nwin = (A.exp_bits + w - 1) / w;
  auto digit = [&](int i) -> int {
    int bit = i * w;
    int word = bit >> 6, sh = bit & 63;
    uint64_t v = (word < A.exp_words) ? ep[word] >> sh : 0;
    if (sh + w > 64 && word + 1 < A.exp_words) v |= ep[word + 1] << (64 - sh);
    return (int)(v & (uint64_t)(tsize - 1));
  };

  // ---- c * R as a pair: sum over the ciphertext's chunks z_i (half-width, so (z_i, 0) is a pair) of
  //      (z_i, 0) (x) pair(2^(64*cw*i) * R^2); a schedule adds one trip that squares the sum (its table holds the
  //      odd powers and is built by multiplying with base^2) ----
  uint32_t acc[K];
#pragma unroll
  for (int j = 0; j < K; ++j) acc[j] = 0;
  if (A.ct_pair) {
    // the ciphertext is a pair row of the n^2 domain: c*Rn == a - Pn*b.  Modulo p^2 (Pn = p * (n/p) * kn):
    //   c*Rn == a - P*(kappa*b),  kappa = (n/p)*kn*k^-1 mod p   -- the a part enters in pchunks chunks z_i like the words of
    // a plain ciphertext do, (z_i, 0) (x) pconv[i]; the b part only matters modulo p: b_i (x) pcb[i], half-width,
    // added to half B.  Limbs are read in place: no word -> limb conversion, no LDS.
    const uint32_t* row = A.ct_pair + elem * A.ct_pair_stride + (size_t)halfB * A.pair_l2;
    const uint32_t* pconv = HCTX(pconv);
    const uint32_t* pcb = HCTX(pcb);
    const int trips = A.pchunks + ((sched_mode && tsize > 1) ? 1 : 0);
#pragma unroll 1
    for (int i = 0; i < trips; ++i) {
      if (i < A.pchunks) {
        const int first = i * A.pchunk_limbs;
        uint32_t z[K], cb[K], tb[K];
#pragma unroll
        for (int j = 0; j < K; ++j) {
          const int li = x * K + j;
          z[j] = (li < A.pchunk_limbs && first + li < A.pair_l2) ? row[first + li] : 0u;   // half A: a_i, half B: b_i
          cb[j] = pcb[(size_t)i * L2 + x * K + j];
          mreg[j] = pconv[(size_t)i * LQ + xg * K + j];
          own[j] = halfB ? 0u : z[j];
        }
        montmul_reg<HG, false, true>(tb, z, cb, n, 0);   // (half A's result is not used)
        pairmul<H, K, false, true>(own, own, mreg, n, n0inv, halfB, selB);
#pragma unroll
        for (int j = 0; j < K; ++j) own[j] += halfB ? tb[j] : 0u;
        add_normalise<HG>(acc, own);
      } else {
#pragma unroll
        for (int j = 0; j < K; ++j) own[j] = mreg[j] = acc[j];
        pairmul<H, K, false, true>(own, own, mreg, n, n0inv, halfB, selB);
      }
    }
  } else {
    const uint64_t* row = A.ct + elem * A.ct_stride;
    const uint32_t* conv = HCTX(conv);
    const int trips = A.nchunks + ((sched_mode && tsize > 1) ? 1 : 0);
#pragma unroll 1
    for (int i = 0; i < trips; ++i) {
      if (i < A.nchunks) {
        wave_lds_sync();
        const int first = i * A.chunk_words;
        const int words = min(A.chunk_words, A.ct_words - first);
        for (int t = xg; t <= W64; t += GS) io[grp][t] = (t < words) ? row[first + t] : 0;
        wave_lds_sync();
#pragma unroll
        for (int j = 0; j < K; ++j) {
          own[j] = halfB ? 0u : limb_from_words(io[grp], x * K + j);
          mreg[j] = conv[(size_t)i * LQ + xg * K + j];
        }
      } else {
#pragma unroll
        for (int j = 0; j < K; ++j) own[j] = mreg[j] = acc[j];
      }
      pairmul<H, K, false, true>(own, own, mreg, n, n0inv, halfB, selB);
      if (i < A.nchunks) add_normalise<HG>(acc, own);
    }
  }

  // ---- window table.  Fixed window: all powers 0 .. 2^w-1, entry e = entry e-1 times the base;
  //      schedule: the odd powers, entry e = entry e-1 times base^2 ----
  {
    int e;
    if (sched_mode) {
#pragma unroll
      for (int j = 0; j < K; ++j) {
        tbl[j] = acc[j];
        mreg[j] = own[j];      // base^2 (the extra trip above; unused if the table has one entry)
        own[j] = acc[j];
      }
      e = 1;
    } else {
#pragma unroll
      for (int j = 0; j < K; ++j) {
        tbl[LQ + j] = acc[j];
        tbl[j] = HCTX(one)[xg * K + j];
        mreg[j] = own[j] = acc[j];
      }
      e = 2;
    }
#pragma unroll 1
    for (; e < tsize; ++e) {
      pairmul<H, K, false, true>(own, own, mreg, n, n0inv, halfB, selB);
#pragma unroll
      for (int j = 0; j < K; ++j) tbl[(size_t)e * LQ + j] = own[j];
    }
  }

  // ---- main loop: steps of (nsq squarings, one multiplication by a table entry) ----
  // fixed window: nsq = w, entry = the next exponent digit (always multiplies, also by entry 0 = 1);
  // schedule: the steps the host built (kargs.hpp: ModexpArgs::sched)
  int win;
  bool any = true;
  if (sched_mode) {
    if (nsteps == 0) any = false;
    win = 1;
  } else {
    if (nwin == 0) any = false;
    win = nwin - 2;
  }
  if (any) {
    const int d0 = sched_mode ? (__builtin_amdgcn_readfirstlane((int)sch[0]) & 63) - 1 : digit(nwin - 1);
    load_table_entry<K>(own, tbl, d0, tsize, LQ, A.ct_gather != 0);   // (this lane's own earlier stores)
  } else {
#pragma unroll
    for (int j = 0; j < K; ++j) own[j] = HCTX(one)[xg * K + j];
  }
  // Placement of the main loop in the instruction stream (A/B: tools/build_variant.py -DPGPU_PHASE_PAD=n): bit 0 shifts
  // it by one 4-byte instruction, bit 1 starts it on a 64-byte line.  A lone wavefront's issue cadence depends on where
  // its 8-byte instructions sit relative to the fetch granules (profiles/r03_ubench_phase.txt).
#ifndef PGPU_PHASE_PAD
#define PGPU_PHASE_PAD 0
#endif
  if constexpr ((PGPU_PHASE_PAD & 2) != 0) asm volatile(".p2align 6");
  if constexpr ((PGPU_PHASE_PAD & 1) != 0) asm volatile("s_nop 0");
#pragma unroll 1
  for (;;) {
    int nsq, idx;
    if (sched_mode) {
      if (!any || win >= nsteps) break;
      const int st = __builtin_amdgcn_readfirstlane((int)sch[win++]);
      nsq = st >> 6;
      idx = (st & 63) - 1;
    } else {
      if (!any || win < 0) break;
      nsq = w;
      idx = digit(win--);
    }
    const bool mul = !sched_mode || idx >= 0;
    if (mul) load_table_entry<K>(mreg, tbl, idx, tsize, LQ, A.ct_gather != 0);   // the entry travels while the squarings run
#pragma unroll 1
    for (int i = 0; i < nsq; ++i) pairmul<H, K, true, true>(own, own, own, n, n0inv, halfB, selB);
    if (mul) pairmul<H, K, false, true>(own, own, mreg, n, n0inv, halfB, selB);
  }

  // ---- leave the Montgomery domain under the TRUE prime: (a, k*b mod p) is a pair modulo p^2 (only b's residue
  //      modulo p matters); multiply by (hp, 0) ----
  {
#pragma unroll
    for (int j = 0; j < K; ++j) {
      n[j] = HCTX(n)[x * K + j];
      mreg[j] = HCTX(kr)[x * K + j];      // k*R mod p: a half-width Montgomery product by it multiplies by k
    }
    n0inv = HCTX(n0inv);
    uint32_t kb[K];
    montmul_reg<HG, false, false>(kb, own, mreg, n, n0inv);   // (both halves run it; half A keeps a)
#pragma unroll
    for (int j = 0; j < K; ++j) {
      if (halfB) own[j] = kb[j];
      mreg[j] = halfB ? 0u : HCTX(h)[x * K + j];
    }
    pairmul<H, K, false, false>(own, own, mreg, n, n0inv, halfB, selB);
  }
  // half A: a' in {hp, hp + p};  half B: b' < 2p.   mp = ([a' >= p] - b') mod p
  full_normalise<HG>(own, x);
  uint32_t d[K];
  const uint32_t below = sub_limbs<HG>(d, own, n, x, lane);          // half A: a' < p ?   half B: b' < p ?
  const uint32_t jflag = (uint32_t)__shfl((int)(below ^ 1u), (lane / GS) * GS);   // half A's verdict, group-wide
  if (!below) {
#pragma unroll
    for (int j = 0; j < K; ++j) own[j] = d[j];                       // half B: b'' = b' mod p  (half A: unused)
  }
  (void)sub_limbs<HG>(d, n, own, x, lane);                           // p - b''  in (0, p]
  if (x == 0) d[0] += jflag;
  full_normalise<HG>(d, x);                                          // p - b'' + j  in (0, p + 1]
  const uint32_t small = sub_limbs<HG>(own, d, n, x, lane);          // >= p: take the difference
  if (small) {
#pragma unroll
    for (int j = 0; j < K; ++j) own[j] = d[j];
  }
  wave_lds_sync();
  if (halfB) {
#pragma unroll
    for (int j = 0; j < K; ++j) bl[grp][x * K + j] = own[j];
  }
  wave_lds_sync();
  const int ow = A.out_words;
  for (int t = lane; t < IPW * ow; t += kWave) {
    const int gg = t / ow, ww = t % ow;
    const size_t oe = first_elem + gg;
    if (oe < A.count) A.out[(2 * oe + side) * A.out_stride + ww] = ww < W64 ? word_from_limbs(bl[gg], L2, ww) : 0;
  }
#undef HCTX
}

// value of lane x+D (same 16-lane row); 0 in the last D lanes of the row.
template <int D>
__device__ __forceinline__ uint32_t dpp_from_above(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x100 + D /*row_shl:D*/, 0xf, 0xf, true);
}

// ---- pair rows (kargs.hpp): the HBM image of a resident ciphertext IS the register image of these kernels ----
// lane xg of the group holds limbs [xg*K, xg*K + K) of the row's 2*H*K limbs
template <int K>
__device__ __forceinline__ void load_pair_row(uint32_t (&v)[K], const uint32_t* __restrict__ row, int xg) {
  const uint32_t* p = row + xg * K;
  if constexpr (K % 2 == 0) {   // rows are 8-byte aligned (2*H*K*4 bytes per row, K even): 64-bit loads
    const uint2* q = reinterpret_cast<const uint2*>(p);
#pragma unroll
    for (int j = 0; j < K / 2; ++j) {
      const uint2 t = q[j];
      v[2 * j] = t.x;
      v[2 * j + 1] = t.y;
    }
  } else {
#pragma unroll
    for (int j = 0; j < K; ++j) v[j] = p[j];
  }
}
template <int K>
__device__ __forceinline__ void store_pair_row(uint32_t* __restrict__ row, const uint32_t (&v)[K], int xg) {
  uint32_t* p = row + xg * K;
  if constexpr (K % 2 == 0) {
    uint2* q = reinterpret_cast<uint2*>(p);
#pragma unroll
    for (int j = 0; j < K / 2; ++j) q[j] = make_uint2(v[2 * j], v[2 * j + 1]);
  } else {
#pragma unroll
    for (int j = 0; j < K; ++j) p[j] = v[j];
  }
}

// own (pair of x*R, loop modulus P = n*k) times g^m = 1 + n*m, the result again a pair of the same domain:
//   (a - P*b)(1 + n*m) == a - P*b + n*m*a == a - P*(b - k^-1*m*a)   (mod n^2)     [n*m*P*b == 0; n = P/k]
// so only b changes: b += (-k^-1 * m * a) mod n -- two half-width Montgomery products under the TRUE modulus n
// (montmul(m, gm) = -k^-1*m*R, then times a), instead of the pair product by the pair of g^m (CT + PT:
// ciphertext.cpp:75-80; the g^m factor of encrypt: pub_key.cpp:88-105).  mwords: the group's plaintext words in LDS
// (zero padded to W64+1 of the half width); both halves run the same instruction stream.
template <int H, int K>
__device__ __forceinline__ void pair_times_gm(uint32_t (&own)[K], const HenselPubDev& C, const uint64_t* mwords, int x,
                                              uint32_t halfB) {
  using HG = Geo<H, K>;
  uint32_t n[K], mv[K], cg[K], u[K], av[K], v[K];
#pragma unroll
  for (int j = 0; j < K; ++j) {
    n[j] = C.n[x * K + j];
    cg[j] = C.gm[x * K + j];
    mv[j] = limb_from_words(mwords, x * K + j);
  }
  montmul_reg<HG, false, false>(u, mv, cg, n, C.n0inv);
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const uint32_t below = dpp_from_below<H>(own[j]);   // half B takes a copy of a
    av[j] = halfB ? below : own[j];
  }
  montmul_reg<HG, false, false>(v, u, av, n, C.n0inv);
#pragma unroll
  for (int j = 0; j < K; ++j) v[j] = halfB ? v[j] : 0u;
  add_normalise<HG>(own, v);
}

// own = words * R as a pair: sum over the chunks z_i of (z_i, 0) (x) pair(2^(64*cw*i) * R^2).  iorow: the group's
// W64+1 words of LDS; xg: lane in the group.
template <int H, int K>
__device__ __forceinline__ void pair_from_words(uint32_t (&own)[K], const uint64_t* row, int nwords, int chunk_words,
                                                int nchunks, const uint32_t* conv, uint64_t* iorow,
                                                const uint32_t (&n)[K], uint32_t halfB, uint32_t selB, int xg) {
  using HG = Geo<H, K>;
  constexpr int GS = 2 * H, LQ = 2 * H * K, W64 = HG::W64;
  const int x = xg % H;
  uint32_t acc[K], mreg[K];
#pragma unroll
  for (int j = 0; j < K; ++j) acc[j] = 0;
#pragma unroll 1
  for (int i = 0; i < nchunks; ++i) {
    wave_lds_sync();
    const int first = i * chunk_words;
    const int words = min(chunk_words, nwords - first);
    for (int t = xg; t <= W64; t += GS) iorow[t] = (t < words) ? row[first + t] : 0;
    wave_lds_sync();
#pragma unroll
    for (int j = 0; j < K; ++j) {
      own[j] = halfB ? 0u : limb_from_words(iorow, x * K + j);
      mreg[j] = conv[(size_t)i * LQ + xg * K + j];
    }
    pairmul<H, K, false, true>(own, own, mreg, n, 0, halfB, selB);
    add_normalise<HG>(acc, own);
  }
#pragma unroll
  for (int j = 0; j < K; ++j) own[j] = acc[j];
}

// Leaves the Montgomery domain of a pair x*R modulo (n*k)^2 and stores the residue modulo n^2 as 64-bit words:
// the exit product runs under the true modulus n on (a, k*b mod n), by (1, 0) or -- with_gm -- by the pair
// (1, 2n - m) of 1 + n*m (1 + n*m == 1 - n*(2n - m) mod n^2; m < 2^(64 * words of n) <= 2n); the canonical pair
// a'' = a' mod n, b'' = ([a' >= n] - b') mod n is the residue c = a'' + n*b'' < n^2, formed by one full-width Montgomery
// product in Geo<2H,K> (two for a Montgomery-form result, HenselFullDev).  bl / io: the wavefront's LDS arrays in the
// full-width geometry; rows: [2][FG::L] LDS, filled here.
template <int H, int K>
__device__ __forceinline__ void pair_exit_store(uint32_t (&own)[K], const HenselPubDev& C, const HenselFullDev& F,
                                                bool with_gm, const uint64_t* fm_words, size_t fm_stride, int fm_nwords,
                                                uint64_t* out, size_t out_stride, size_t first_inst, size_t count,
                                                uint32_t (*bl)[2 * H * K], uint64_t (*io)[Geo<2 * H, K>::W64 + 1],
                                                uint32_t (*rows)[2 * H * K], int lane, int grp, int xg, uint32_t halfB,
                                                uint32_t selB) {
  using HG = Geo<H, K>;
  using FG = Geo<2 * H, K>;
  constexpr int GS = 2 * H;
  const int x = xg % H;
  for (int t = lane; t < FG::L; t += kWave) {
    rows[0][t] = F.nr[t];
    rows[1][t] = F.r2 ? F.r2[t] : 0;
  }
  if (with_gm) stage_words<FG>(io, fm_words, fm_stride, 0, fm_nwords, first_inst, count, 1, lane);
  uint32_t n[K], mreg[K];
#pragma unroll
  for (int j = 0; j < K; ++j) {
    n[j] = C.n[x * K + j];
    mreg[j] = C.kr[x * K + j];
  }
  const uint32_t n0inv = C.n0inv;
  {
    uint32_t kb[K];
    montmul_reg<HG, false, false>(kb, own, mreg, n, n0inv);
#pragma unroll
    for (int j = 0; j < K; ++j)
      if (halfB) own[j] = kb[j];
  }
  wave_lds_sync();
  uint32_t d[K], e[K];
  if (with_gm) {
#pragma unroll
    for (int j = 0; j < K; ++j) {
      e[j] = limb_from_words(io[grp], x * K + j);     // m (both halves compute it; half B uses it)
      d[j] = 2 * n[j];
    }
    full_normalise<HG>(d, x);
    (void)sub_limbs<HG>(mreg, d, e, x, lane);        // 2n - m
  } else {
#pragma unroll
    for (int j = 0; j < K; ++j) mreg[j] = 0;
  }
#pragma unroll
  for (int j = 0; j < K; ++j)
    if (!halfB) mreg[j] = (xg == 0 && j == 0) ? 1u : 0u;
  pairmul<H, K, false, false>(own, own, mreg, n, n0inv, halfB, selB);
  // ---- canonical pair ----
  full_normalise<HG>(own, x);
  const uint32_t below = sub_limbs<HG>(d, own, n, x, lane);
  const uint32_t jflag = (uint32_t)__shfl((int)(below ^ 1u), (lane / GS) * GS);
  if (!below) {
#pragma unroll
    for (int j = 0; j < K; ++j) own[j] = d[j];
  }
  (void)sub_limbs<HG>(d, n, own, x, lane);          // half B: n - (b' mod n) in (0, n]
  if (x == 0) d[0] += jflag;
  full_normalise<HG>(d, x);
  const uint32_t small = sub_limbs<HG>(e, d, n, x, lane);
  // full-width operands in the lane layout of Geo<2H,K>: lane xg holds limbs [xg*K, xg*K + K); both sit in the low half
  uint32_t X[K], Y[K], nf[K];
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const uint32_t bc = small ? d[j] : e[j];
    const uint32_t down = dpp_from_above<H>(bc);
    X[j] = halfB ? 0u : own[j];
    Y[j] = halfB ? 0u : down;
    nf[j] = F.n[xg * K + j];
  }
  wave_lds_sync();
  uint32_t t[K];
  montmul<FG, false, false>(t, Y, rows[0], nf, F.n0inv);     // n * b''   (its Montgomery form: n*R' * b'')
  if (F.r2) {
    uint32_t u[K];
    montmul<FG, false, false>(u, X, rows[1], nf, F.n0inv);   // a'' * R'
    add_normalise<FG>(t, u);
  } else {
    add_normalise<FG>(t, X);
  }
  store_canonical<FG>(t, nf, F.mod_words, bl, io, out, out_stride, first_inst, count, lane, grp, xg);
}

// Fixed-base table of pairs for the DJN obfuscator hs^r (kernels.hpp: fb_build_kernel is the full-width twin):
// group i builds row i, T[i][d] = hs^(d * 2^(w*i)) * R as a pair.
template <int H, int K>
__global__ __launch_bounds__(kWGThreads, K <= 14 ? 2 : 1) void hensel_fb_build_kernel(HenselFbBuildArgs A) {
  using HG = Geo<H, K>;
  constexpr int GS = 2 * H, IPW = kWave / GS, LQ = 2 * H * K, W64 = HG::W64;
  __shared__ uint64_t io_[kWavesPerWG][IPW][W64 + 1];
  const int lane = threadIdx.x % kWave, wv = threadIdx.x / kWave;
  auto& io = io_[wv];
  const int grp = lane / GS, xg = lane % GS, x = xg % H;
  const uint32_t halfB = (uint32_t)(xg / H);
  uint32_t selB = xg == H ? 1u : 0u;
  asm("" : "+v"(selB));
  const size_t first_inst = ((size_t)blockIdx.x * kWavesPerWG + wv) * IPW;
  size_t inst = first_inst + grp;
  const bool live = inst < (size_t)A.nwin;
  if (!live) inst = (size_t)A.nwin - 1;
  uint32_t n[K], own[K], mreg[K];
#pragma unroll
  for (int j = 0; j < K; ++j) n[j] = A.ctx.nhat[x * K + j];
  pair_from_words<H, K>(own, A.base, A.base_words, A.chunk_words, A.nchunks, A.ctx.conv, io[grp], n, halfB, selB, xg);
  const int tsize = 1 << A.w;
  // every group runs the squaring count of the LAST live row of its wavefront (uniform control flow); a group stops
  // updating once its own count is reached
  const int my_sq = A.w * (int)inst;
  size_t last = first_inst + IPW - 1;
  if (last >= (size_t)A.nwin) last = (size_t)A.nwin - 1;
  const int wave_sq = A.w * (int)last;
#pragma unroll 1
  for (int step = 1; step <= wave_sq; ++step) {
    uint32_t r[K];
    pairmul<H, K, true, true>(r, own, own, n, 0, halfB, selB);
    if (step <= my_sq) {
#pragma unroll
      for (int j = 0; j < K; ++j) own[j] = r[j];
    }
  }
  uint32_t* row = A.table + inst * (size_t)tsize * LQ + xg * K;
  if (live) {
#pragma unroll
    for (int j = 0; j < K; ++j) {
      row[j] = A.ctx.one[xg * K + j];
      row[LQ + j] = own[j];
    }
  }
#pragma unroll
  for (int j = 0; j < K; ++j) mreg[j] = own[j];
#pragma unroll 1
  for (int d = 2; d < tsize; ++d) {
    pairmul<H, K, false, true>(own, own, mreg, n, 0, halfB, selB);
    if (live) {
#pragma unroll
      for (int j = 0; j < K; ++j) row[(size_t)d * LQ + j] = own[j];
    }
  }
}

// DJN encrypt c = (1 + n*m) * hs^r mod n^2 (pub_key.cpp:51-64, 88-105) in split form: hs^r as nwin-1 pair products
// of table entries, the exit product under the true modulus n by the pair (1, -m) = 1 + n*m, then back to a
// full-width residue: for canonical a, b < n the pair is c = a + n*b (< n^2), one full-width Montgomery product in
// the geometry Geo<2H,K> of the n^2 context (two for a Montgomery-form result).
template <int H, int K>
__global__ __launch_bounds__(kWGThreads, K <= 14 ? 2 : 1) void hensel_fb_encrypt_kernel(HenselFbArgs A) {
  using HG = Geo<H, K>;
  using FG = Geo<2 * H, K>;
  constexpr int GS = 2 * H, IPW = kWave / GS, LQ = 2 * H * K;
  raise_wave_priority();
  __shared__ uint32_t bl_[kWavesPerWG][IPW][FG::L];
  __shared__ uint64_t io_[kWavesPerWG][IPW][FG::W64 + 1];
  __shared__ uint32_t rows_[kWavesPerWG][2][FG::L];     // n*R' (n*R'^2) and R'^2 as multiplier rows
  const int lane = threadIdx.x % kWave, wv = threadIdx.x / kWave;
  auto& bl = bl_[wv];
  auto& io = io_[wv];
  auto& rows = rows_[wv];
  const int grp = lane / GS, xg = lane % GS, x = xg % H;
  const uint32_t halfB = (uint32_t)(xg / H);
  uint32_t selB = xg == H ? 1u : 0u;
  asm("" : "+v"(selB));
  const size_t first_inst = ((size_t)blockIdx.x * kWavesPerWG + wv) * IPW;
  size_t inst = first_inst + grp;
  if (inst >= A.count) inst = A.count - 1;
  uint32_t n[K], own[K], mreg[K], nxt[K];
#pragma unroll
  for (int j = 0; j < K; ++j) n[j] = A.ctx.nhat[x * K + j];
  const int w = A.w, tsize = 1 << w;
  const uint64_t* ep = A.exp + inst * A.exp_stride;
  auto digit = [&](int i) -> int {
    int bit = i * w;
    int word = bit >> 6, sh = bit & 63;
    uint64_t v = (word < A.exp_words) ? ep[word] >> sh : 0;
    if (sh + w > 64 && word + 1 < A.exp_words) v |= ep[word + 1] << (64 - sh);
    return (int)(v & (uint64_t)(tsize - 1));
  };
  auto load_entry = [&](uint32_t (&dst)[K], int i) {   // (masked: the address stream does not depend on the digits of r)
    load_table_entry<K>(dst, A.table + (size_t)i * tsize * LQ + xg * K, digit(i), tsize, LQ, A.ct_gather != 0);
  };
  load_entry(own, 0);
  if (A.nwin > 1) load_entry(mreg, 1);
  // the entry of the next step is fetched before the product of this one (latency hidden)
#pragma unroll 1
  for (int i = 1; i < A.nwin; ++i) {
    if (i + 1 < A.nwin) load_entry(nxt, i + 1);
    pairmul<H, K, false, true>(own, own, mreg, n, 0, halfB, selB);
#pragma unroll
    for (int j = 0; j < K; ++j) mreg[j] = nxt[j];
  }
  if (A.out_pair) {
    // resident result: stay a pair.  hs^r * (1 + n*m): two half-width products (pair_times_gm), no way back to words.
    stage_words<FG>(io, A.fm_words, A.fm_stride, 0, A.fm_nwords, first_inst, A.count, 1, lane);
    wave_lds_sync();
    pair_times_gm<H, K>(own, A.ctx, io[grp], x, halfB);
    if (first_inst + grp < A.count) store_pair_row<K>(A.out_pair + inst * (size_t)LQ, own, xg);
    return;
  }
  pair_exit_store<H, K>(own, A.ctx, A.full, true, A.fm_words, A.fm_stride, A.fm_nwords, A.out, A.out_stride, first_inst,
                        A.count, bl, io, rows, lane, grp, xg, halfB, selB);
}

// base[i]^exp[i] modulo n^2 in split form -- CT x PT (ciphertext.cpp:143-162: per-element exponents, fixed window) and
// the non-DJN obfuscator r^n with its g^m product (pub_key.cpp:66-80, 88-105: shared exponent n, the host's schedule).
// Same structure as modexp_kernel; entry as in hensel_decrypt_kernel, exit as in hensel_fb_encrypt_kernel.
// (18 limbs per lane squeezed into the 256 registers of two wavefronts per SIMD spill a few set-up values to scratch,
// none in the loops: a 1 M-element CT x PT batch 55.8 -> 48.4 ms, launches of one wavefront per SIMD unchanged)
#ifndef PGPU_HM_WAVES18
#define PGPU_HM_WAVES18 2
#endif
template <int H, int K>
__global__ __launch_bounds__(kWGThreads, K <= 14 ? 2 : PGPU_HM_WAVES18) void hensel_modexp_kernel(HenselModexpArgs A) {
  using HG = Geo<H, K>;
  using FG = Geo<2 * H, K>;
  constexpr int GS = 2 * H, IPW = kWave / GS, LQ = 2 * H * K;
  raise_wave_priority();
  __shared__ uint32_t bl_[kWavesPerWG][IPW][FG::L];
  __shared__ uint64_t io_[kWavesPerWG][IPW][FG::W64 + 1];
  __shared__ uint32_t rows_[kWavesPerWG][2][FG::L];
  const int lane = threadIdx.x % kWave, wv = threadIdx.x / kWave;
  auto& bl = bl_[wv];
  auto& io = io_[wv];
  auto& rows = rows_[wv];
  const int grp = lane / GS, xg = lane % GS, x = xg % H;
  const uint32_t halfB = (uint32_t)(xg / H);
  uint32_t selB = xg == H ? 1u : 0u;
  asm("" : "+v"(selB));
  const size_t wave_id = (size_t)blockIdx.x * kWavesPerWG + wv;
  const size_t first_inst = wave_id * IPW;
  size_t inst = first_inst + grp;
  if (inst >= A.count) inst = A.count - 1;
  uint32_t n[K], own[K], mreg[K];
#pragma unroll
  for (int j = 0; j < K; ++j) n[j] = A.ctx.nhat[x * K + j];

  const bool sched_mode = A.sched != nullptr;
  const uint16_t* sch = A.sched;
  const int nsteps = A.sched_len;
  const int w = A.window;
  const int tsize = sched_mode ? 1 << (w - 1) : 1 << w;
  uint32_t* tbl = A.table + (wave_id * IPW + grp) * (size_t)tsize * LQ + xg * K;
  const uint64_t* ep = A.exp + inst * A.exp_stride;
  const int nwin = (A.exp_bits + w - 1) / w;
  auto digit = [&](int i) -> int {
    int bit = i * w;
    int word = bit >> 6, sh = bit & 63;
    uint64_t v = (word < A.exp_words) ? ep[word] >> sh : 0;
    if (sh + w > 64 && word + 1 < A.exp_words) v |= ep[word + 1] << (64 - sh);
    return (int)(v & (uint64_t)(tsize - 1));
  };

  // (the FG-sized row of io holds more than the W64+1 words of a half-width chunk)
  if (A.base_pair) load_pair_row<K>(own, A.base_pair + inst * A.base_pair_stride, xg);   // resident base: already c*R as a pair
  else pair_from_words<H, K>(own, A.base + inst * A.base_stride, A.base_words, A.chunk_words, A.nchunks, A.ctx.conv,
                             io[grp], n, halfB, selB, xg);
  // ---- window table (fixed window: all powers; schedule: the odd powers, built with base^2) ----
  {
    int e;
    if (sched_mode) {
#pragma unroll
      for (int j = 0; j < K; ++j) tbl[j] = own[j];
      if (tsize > 1) pairmul<H, K, false, true>(mreg, own, own, n, 0, halfB, selB);
      e = 1;
    } else {
#pragma unroll
      for (int j = 0; j < K; ++j) {
        tbl[LQ + j] = own[j];
        tbl[j] = A.ctx.one[xg * K + j];
        mreg[j] = own[j];
      }
      e = 2;
    }
#pragma unroll 1
    for (; e < tsize; ++e) {
      pairmul<H, K, false, true>(own, own, mreg, n, 0, halfB, selB);
#pragma unroll
      for (int j = 0; j < K; ++j) tbl[(size_t)e * LQ + j] = own[j];
    }
  }
  // ---- main loop: steps of (nsq squarings, one multiplication by a table entry), as in hensel_decrypt_kernel ----
  int win;
  bool any = true;
  if (sched_mode) {
    if (nsteps == 0) any = false;
    win = 1;
  } else {
    if (nwin == 0) any = false;
    win = nwin - 2;
  }
  if (any) {
    const int d0 = sched_mode ? (__builtin_amdgcn_readfirstlane((int)sch[0]) & 63) - 1 : digit(nwin - 1);
    load_table_entry<K>(own, tbl, d0, tsize, LQ, A.ct_gather != 0);
  } else {
#pragma unroll
    for (int j = 0; j < K; ++j) own[j] = A.ctx.one[xg * K + j];
  }
#pragma unroll 1
  for (;;) {
    int nsq, idx;
    if (sched_mode) {
      if (!any || win >= nsteps) break;
      const int st = __builtin_amdgcn_readfirstlane((int)sch[win++]);
      nsq = st >> 6;
      idx = (st & 63) - 1;
    } else {
      if (!any || win < 0) break;
      nsq = w;
      idx = digit(win--);
    }
    const bool mul = !sched_mode || idx >= 0;
    if (mul) load_table_entry<K>(mreg, tbl, idx, tsize, LQ, A.ct_gather != 0);
#pragma unroll 1
    for (int i = 0; i < nsq; ++i) pairmul<H, K, true, true>(own, own, own, n, 0, halfB, selB);
    if (mul) pairmul<H, K, false, true>(own, own, mreg, n, 0, halfB, selB);
  }
  if (A.out_pair) {
    if (A.final_mul == FM_PAILLIER_G) {
      wave_lds_sync();
      stage_words<FG>(io, A.fm_words, A.fm_stride, 0, A.fm_nwords, first_inst, A.count, 1, lane);
      wave_lds_sync();
      pair_times_gm<H, K>(own, A.ctx, io[grp], xg % H, halfB);
    }
    if (first_inst + grp < A.count) store_pair_row<K>(A.out_pair + inst * (size_t)LQ, own, xg);
    return;
  }
  pair_exit_store<H, K>(own, A.ctx, A.full, A.final_mul == FM_PAILLIER_G, A.fm_words, A.fm_stride, A.fm_nwords, A.out,
                        A.out_stride, first_inst, A.count, bl, io, rows, lane, grp, xg, halfB, selB);
}

// Element-wise operations on pair rows (kargs.hpp: PairOp): CT + CT as ONE pair product, CT + PT as two half-width
// products, and the conversions between pair rows and 64-bit words (upload of caller data, pgpu_batch_download).
template <int H, int K>
__global__ __launch_bounds__(kWGThreads, 2) void pair_ops_kernel(PairOpsArgs A) {
  using HG = Geo<H, K>;
  using FG = Geo<2 * H, K>;
  constexpr int GS = 2 * H, IPW = kWave / GS, LQ = 2 * H * K;
  raise_wave_priority();
  __shared__ uint32_t bl_[kWavesPerWG][IPW][FG::L];
  __shared__ uint64_t io_[kWavesPerWG][IPW][FG::W64 + 1];
  __shared__ uint32_t rows_[kWavesPerWG][2][FG::L];
  const int lane = threadIdx.x % kWave, wv = threadIdx.x / kWave;
  auto& bl = bl_[wv];
  auto& io = io_[wv];
  auto& rows = rows_[wv];
  const int grp = lane / GS, xg = lane % GS, x = xg % H;
  const uint32_t halfB = (uint32_t)(xg / H);
  uint32_t selB = xg == H ? 1u : 0u;
  asm("" : "+v"(selB));
  const size_t first_inst = ((size_t)blockIdx.x * kWavesPerWG + wv) * IPW;
  size_t inst = first_inst + grp;
  const bool live = inst < A.count;
  if (!live) inst = A.count - 1;
  uint32_t own[K];
  if (A.op == PO_MUL) {
    uint32_t n[K], mreg[K];
    load_pair_row<K>(own, A.a + inst * (size_t)LQ, xg);
    load_pair_row<K>(mreg, A.b + inst * A.b_stride, xg);
#pragma unroll
    for (int j = 0; j < K; ++j) n[j] = A.ctx.nhat[x * K + j];
    pairmul<H, K, false, true>(own, own, mreg, n, 0, halfB, selB);
    if (live) store_pair_row<K>(A.out + inst * (size_t)LQ, own, xg);
  } else if (A.op == PO_TIMES_GM) {
    load_pair_row<K>(own, A.a + inst * (size_t)LQ, xg);
    stage_words<FG>(io, A.words, A.words_stride, 0, A.nwords, first_inst, A.count, 1, lane);
    wave_lds_sync();
    pair_times_gm<H, K>(own, A.ctx, io[grp], x, halfB);
    if (live) store_pair_row<K>(A.out + inst * (size_t)LQ, own, xg);
  } else if (A.op == PO_FROM_WORDS) {
    uint32_t n[K];
#pragma unroll
    for (int j = 0; j < K; ++j) n[j] = A.ctx.nhat[x * K + j];
    pair_from_words<H, K>(own, A.words + inst * A.words_stride, A.nwords, A.chunk_words, A.nchunks, A.ctx.conv, io[grp], n,
                          halfB, selB, xg);
    if (live) store_pair_row<K>(A.out + inst * (size_t)LQ, own, xg);
  } else {
    load_pair_row<K>(own, A.a + inst * (size_t)LQ, xg);
    pair_exit_store<H, K>(own, A.ctx, A.full, false, nullptr, 0, 0, A.out_words, A.out_stride, first_inst, A.count, bl, io,
                          rows, lane, grp, xg, halfB, selB);
  }
}

}  // namespace pgpu

#endif  // PAILLIERCRYPTOLIB_AMD_CSRC_HENSEL_HPP_

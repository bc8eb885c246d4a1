// pailliercryptolib_amd -- the split form with BOTH halves of a residue in the same lanes (round 3): the form for launches
// that still put a wavefront on every SIMD with half the lanes per residue.  Kernels: hensel_decrypt_seq_kernel (CRT
// decrypt), hensel_modexp_seq_kernel (CT x PT), pair_mul_seq_kernel (CT + CT), hensel_fb_encrypt_seq_kernel (DJN
// encrypt) -- all on resident batches (pair rows, kargs.hpp), all bit-identical with the paired kernels of hensel.hpp
// (tests/test_gpu_pair_rows.py: test_sequential_halves_*); capi.cpp picks them by launch size (PGPU_SEQ_DECRYPT).
//
// hensel_decrypt_kernel (hensel.hpp) gives the a half and the b half of a pair x == a - P*b lanes of their own and runs
// them through one instruction stream; half A then sits through the K^2 products of half B's 2*a*b although its own
// squaring needs K(K+1)/2 (the squaring symmetry), and idles through d*a in a general product: 12.5 % of the kernel's
// VALU instructions (PMC; round 3's A/B-wavefront experiment, retired in round 6).  Here a group of G lanes holds a AND b (K limbs of each per lane) and does
// the two halves one after the other with the same accumulators:
//     t = a*c            half-width Montgomery product modulo P, symmetric when it is a squaring; its quotient digits
//                        are kept in registers (G*K of them)
//     w = a*d + b*c + q  the digits enter column by column (mont_reduce_rows_q, QMODE 2)
// -- the arithmetic of that experiment without its hand-over: no LDS, no counters, no second wavefront.  Per
// exponentiation it issues the instructions of one A and one B stream on HALF the lanes of the paired form (3072-bit keys,
// per squaring and exponentiation: 4 lanes x 3 758 against 8 lanes x ~2 100, -10 %; general products -8 %; 2048-bit
// keys: 2 x 3 287 against 4 x 1 851, -11 %, and -17 %).  Half the lanes means half the wavefronts per batch, so the
// form pays as soon as it still puts a wavefront on every SIMD -- measured (profiles/r03_seq_decrypt.txt): config 4's
// decrypt leg 117.3 -> 103.1 ms, 2048-bit keys 16384 / 32768 / 65536 ciphertexts 4.49 / 4.34 / 4.35 -> 4.20 / 4.18 / 4.00 ms
// per 8192 -- and costs where a batch just fills the chip once in the paired form (the bench's 8192 ciphertexts: 512
// wavefronts, 8.1 ms; stays with hensel_decrypt_kernel).  Ciphertexts as pair rows, fixed-window scan; bit-identical.
#ifndef PAILLIERCRYPTOLIB_AMD_CSRC_HENSEL_SEQ_HPP_
#define PAILLIERCRYPTOLIB_AMD_CSRC_HENSEL_SEQ_HPP_

#include "hensel_q.hpp"

namespace pgpu {

// t = a*m (rows: the limbs of m, lane-distributed like a), digits of block S recorded in qd[S]
// (QLDS: the digits go through qs[S * kAbPad + .] in LDS instead of the registers qd -- forms whose G*K digits do not
// fit beside 12 K other registers; the wavefront reads back what it wrote itself, LDS keeps that in order)
template <int G, int K, bool SQR, bool UNITQ, int S, bool QLDS = false>
__device__ __forceinline__ void seq_a_blocks(uint64_t (&c0)[K], uint64_t (&c1)[K], const uint32_t (&a)[K],
                                             const uint32_t (&a2)[K], const uint32_t (&n)[K], uint32_t n0inv,
                                             const uint32_t (&m)[K], uint32_t (&qd)[QLDS ? 1 : G][K], uint32_t* qs = nullptr) {
  using HG = Geo<G, K>;
  if constexpr (S < G) {
    uint32_t b[K];
#pragma unroll
    for (int r = 0; r < K; ++r) b[r] = bcast_lane<G, S>(m[r]);
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
    if constexpr (QLDS) {
      uint32_t qrec[K];
      mont_reduce_rows_q<HG, UNITQ, 1>(c0, c1, n, n0inv, qrec, 0u);
      ab_store20<K>(qs + S * kAbPad, qrec);   // (every lane of the group holds the same digits)
    } else {
      mont_reduce_rows_q<HG, UNITQ, 1>(c0, c1, n, n0inv, qd[S], 0u);
    }
    seq_a_blocks<G, K, SQR, UNITQ, S + 1, QLDS>(c1, c0, a, a2, n, n0inv, m, qd, qs);
  }
}

// w = a*d + b*c + q: mc (= b, or 2b in a squaring) times the rows of c, md (= d) times the rows of a, the digits of A
template <int G, int K, bool SQR, bool UNITQ, int S, bool QLDS = false>
__device__ __forceinline__ void seq_b_blocks(uint64_t (&c0)[K], uint64_t (&c1)[K], const uint32_t (&mc)[K],
                                             const uint32_t (&md)[K], const uint32_t (&a)[K], const uint32_t (&c)[K],
                                             const uint32_t (&n)[K], uint32_t n0inv, uint32_t (&qd)[QLDS ? 1 : G][K],
                                             uint32_t sel0, const uint32_t* qs = nullptr) {
  using HG = Geo<G, K>;
  if constexpr (S < G) {
    uint32_t row[K];
#pragma unroll
    for (int r = 0; r < K; ++r) row[r] = bcast_lane<G, S>(SQR ? a[r] : c[r]);
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
      for (int r = 0; r < K; ++r) row[r] = bcast_lane<G, S>(a[r]);
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
    if constexpr (QLDS) {
      uint32_t qin[K];
      ab_load20<K>(qin, qs + S * kAbPad);
      mont_reduce_rows_q<HG, UNITQ, 2>(c0, c1, n, n0inv, qin, sel0);
    } else {
      mont_reduce_rows_q<HG, UNITQ, 2>(c0, c1, n, n0inv, qd[S], sel0);
    }
    seq_b_blocks<G, K, SQR, UNITQ, S + 1, QLDS>(c1, c0, mc, md, a, c, n, n0inv, qd, sel0, qs);
  }
}

// (a, b) = (a, b) (x) (c, d): the Montgomery product of two pairs held in the same lanes (lazy: inputs < 8P -> outputs
// < 2P).  A squaring passes c = a, d = b.
// (QLDS, ts != null: the new a part waits in LDS -- kAbPad words of this LANE -- while the b part is computed: K registers)
template <int G, int K, bool SQR, bool UNITQ, bool QLDS = false>
__device__ __forceinline__ void seq_pairmul(uint32_t (&a)[K], uint32_t (&b)[K], const uint32_t (&c)[K],
                                            const uint32_t (&d)[K], const uint32_t (&n)[K], uint32_t n0inv, uint32_t sel0,
                                            uint32_t* qs = nullptr, uint32_t* ts = nullptr) {
  static_assert(3 * K + 6 < 64, "a column receives 3K products (+ relaxed limbs): must stay below 2^64");
  static_assert(G % 2 == 0, "blocks alternate between the two accumulator sets and end in the first");
  using HG = Geo<G, K>;
  uint32_t qd[QLDS ? 1 : G][K];
  uint32_t t[K];
  {
    uint64_t c0[K], c1[K];
    uint32_t a2[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
      c0[j] = 0;
      c1[j] = 0;
      a2[j] = SQR ? a[j] << 1 : 0;
    }
    seq_a_blocks<G, K, SQR, UNITQ, 0, QLDS>(c0, c1, a, a2, n, n0inv, c, qd, qs);
    montmul_finish<HG>(t, c0);
    if constexpr (QLDS) {
      if (ts) ab_store20<K>(ts, t);
    }
  }
  {
    uint64_t c0[K], c1[K];
    uint32_t mc[K], md[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
      c0[j] = 0;
      c1[j] = 0;
      mc[j] = SQR ? b[j] << 1 : b[j];
      md[j] = SQR ? 0u : d[j];
    }
    seq_b_blocks<G, K, SQR, UNITQ, 0, QLDS>(c0, c1, mc, md, a, c, n, n0inv, qd, sel0, qs);
    montmul_finish<HG>(b, c0);
  }
  if constexpr (QLDS) {
    if (ts) {
      ab_load20<K>(a, ts);
      return;
    }
  }
#pragma unroll
  for (int j = 0; j < K; ++j) a[j] = t[j];
}

// One wavefront = 64/G ciphertexts of ONE side (wave parity: even = p, odd = q).  Output: row 2i = mp, row 2i+1 = mq
// (canonical words) for crt_kernel, like hensel_decrypt_kernel.
// MINW: workgroups per CU the build must leave room for (2: 256 registers, the form for launches of two wavefronts per
// SIMD; 1: the whole register file -- the build for launches that claim whole CUs and therefore run ONE wavefront per SIMD by
// construction, round 5: no scratch)
template <int G, int K, int MINW = 2>
__global__ __launch_bounds__(kWGThreads, MINW) void hensel_decrypt_seq_kernel(HenselArgs A) {
  using HG = Geo<G, K>;
  constexpr int IPW = kWave / G, L2 = G * K, LQ = 2 * L2, W64 = HG::W64;
  raise_wave_priority();
  __shared__ uint32_t bl_[kWavesPerWG][IPW][L2];
  const int lane = threadIdx.x % kWave, wv = threadIdx.x / kWave;
  auto& bl = bl_[wv];
  const int grp = lane / G, x = lane % G;
  uint32_t sel0 = x == 0 ? 1u : 0u;
  asm("" : "+v"(sel0));
  const size_t wave_id = (size_t)blockIdx.x * kWavesPerWG + wv;
  const int side = __builtin_amdgcn_readfirstlane((int)(wave_id & 1));
  const size_t first_elem = (wave_id >> 1) * IPW;
  size_t elem = first_elem + grp;
  if (elem >= A.count) elem = A.count - 1;
#define HCTX(field) (side ? A.ctx[1].field : A.ctx[0].field)
  uint32_t n[K], a[K], b[K], ma[K], mb[K];
#pragma unroll
  for (int j = 0; j < K; ++j) n[j] = HCTX(nhat)[x * K + j];
  const int w = A.window, tsize = 1 << w;
  uint32_t* tbl = A.table + (wave_id * IPW + grp) * (size_t)tsize * LQ + x * K;   // entry e: a part at e*LQ, b part at e*LQ + L2
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
  {
    const uint32_t* row = A.ct_pair + elem * A.ct_pair_stride;
    uint32_t acc_a[K], acc_b[K];
#pragma unroll
    for (int j = 0; j < K; ++j) acc_a[j] = acc_b[j] = 0;
#pragma unroll 1
    for (int i = 0; i < A.pchunks; ++i) {
      const int first = i * A.pchunk_limbs;
      uint32_t zb[K], cb[K], tb[K];
#pragma unroll
      for (int j = 0; j < K; ++j) {
        const int li = x * K + j;
        const bool in = li < A.pchunk_limbs && first + li < A.pair_l2;
        a[j] = in ? row[first + li] : 0u;
        zb[j] = in ? row[A.pair_l2 + first + li] : 0u;
        b[j] = 0;
        cb[j] = HCTX(pcb)[(size_t)i * L2 + x * K + j];
        ma[j] = HCTX(pconv)[(size_t)i * LQ + x * K + j];
        mb[j] = HCTX(pconv)[(size_t)i * LQ + L2 + x * K + j];
      }
      montmul_reg<HG, false, true>(tb, zb, cb, n, 0);
      seq_pairmul<G, K, false, true>(a, b, ma, mb, n, 0, sel0);
#pragma unroll
      for (int j = 0; j < K; ++j) b[j] += tb[j];
      add_normalise<HG>(acc_a, a);
      add_normalise<HG>(acc_b, b);
    }
#pragma unroll
    for (int j = 0; j < K; ++j) {
      a[j] = ma[j] = acc_a[j];
      b[j] = mb[j] = acc_b[j];
    }
  }
  // ---- window table: entry 0 = one, entry 1 = base, entry e = entry e-1 times base ----
#pragma unroll
  for (int j = 0; j < K; ++j) {
    tbl[(size_t)LQ + j] = a[j];
    tbl[(size_t)LQ + L2 + j] = b[j];
    tbl[j] = HCTX(one)[x * K + j];
    tbl[L2 + j] = HCTX(one)[L2 + x * K + j];
  }
#pragma unroll 1
  for (int e = 2; e < tsize; ++e) {
    seq_pairmul<G, K, false, true>(a, b, ma, mb, n, 0, sel0);
#pragma unroll
    for (int j = 0; j < K; ++j) {
      tbl[(size_t)e * LQ + j] = a[j];
      tbl[(size_t)e * LQ + L2 + j] = b[j];
    }
  }
  // ---- main loop: w squarings, one multiplication by a table entry (always, also entry 0 = one) ----
  int win = nwin - 2;
  if (nwin > 0) {
    const int d0 = digit(nwin - 1);
    load_table_entry<K>(a, tbl, d0, tsize, LQ, gather);
    load_table_entry<K>(b, tbl + L2, d0, tsize, LQ, gather);
  } else {
#pragma unroll
    for (int j = 0; j < K; ++j) {
      a[j] = HCTX(one)[x * K + j];
      b[j] = HCTX(one)[L2 + x * K + j];
    }
  }
#pragma unroll 1
  for (; nwin > 0 && win >= 0; --win) {
    const int idx = digit(win);
    load_table_entry<K>(ma, tbl, idx, tsize, LQ, gather);        // the entry travels while the squarings run
    load_table_entry<K>(mb, tbl + L2, idx, tsize, LQ, gather);
#pragma unroll 1
    for (int i = 0; i < w; ++i) seq_pairmul<G, K, true, true>(a, b, a, b, n, 0, sel0);
    seq_pairmul<G, K, false, true>(a, b, ma, mb, n, 0, sel0);
  }
  // ---- exit under the TRUE prime: (a, k*b mod p) times (hp, 0);  mp = ([a' >= p] - b') mod p ----
  uint32_t np[K];
#pragma unroll
  for (int j = 0; j < K; ++j) {
    np[j] = HCTX(n)[x * K + j];
    ma[j] = HCTX(kr)[x * K + j];
  }
  const uint32_t n0 = HCTX(n0inv);
  {
    uint32_t kb[K];
    montmul_reg<HG, false, false>(kb, b, ma, np, n0);
#pragma unroll
    for (int j = 0; j < K; ++j) {
      b[j] = kb[j];
      ma[j] = HCTX(h)[x * K + j];
      mb[j] = 0;
    }
  }
  seq_pairmul<G, K, false, false>(a, b, ma, mb, np, n0, sel0);
  full_normalise<HG>(a, x);
  uint32_t d[K];
  const uint32_t below_a = sub_limbs<HG>(d, a, np, x, lane);
  const uint32_t jflag = below_a ^ 1u;
  full_normalise<HG>(b, x);
  const uint32_t below = sub_limbs<HG>(d, b, np, x, lane);
  if (!below) {
#pragma unroll
    for (int j = 0; j < K; ++j) b[j] = d[j];
  }
  (void)sub_limbs<HG>(d, np, b, x, lane);
  if (x == 0) d[0] += jflag;
  full_normalise<HG>(d, x);
  const uint32_t small = sub_limbs<HG>(b, d, np, x, lane);
  if (small) {
#pragma unroll
    for (int j = 0; j < K; ++j) b[j] = d[j];
  }
  wave_lds_sync();
#pragma unroll
  for (int j = 0; j < K; ++j) bl[grp][x * K + j] = b[j];
  wave_lds_sync();
  const int ow = A.out_words;
  for (int t = lane; t < IPW * ow; t += kWave) {
    const int gg = t / ow, ww = t % ow;
    const size_t oe = first_elem + gg;
    if (oe < A.count) A.out[(2 * oe + side) * A.out_stride + ww] = ww < W64 ? word_from_limbs(bl[gg], L2, ww) : 0;
  }
#undef HCTX
}

// base[i]^exp[i] modulo n^2 with both halves in the same lanes: CT x PT on resident batches (hensel_modexp_kernel's
// pair-row entry and exit, per-element exponents, fixed window), large launches.  G*K = 72 digits per product do not
// fit in registers beside the operands: they go through LDS (320 B per exponentiation), and the new a part of a
// product waits there while its b part is computed (80 B per lane) -- no scratch in any loop.
template <int G, int K>
__global__ __launch_bounds__(kWGThreads, 2) void hensel_modexp_seq_kernel(HenselModexpArgs A) {
  constexpr int IPW = kWave / G, L2 = G * K, LQ = 2 * L2;
  raise_wave_priority();
  __shared__ __attribute__((aligned(16))) uint32_t qs_[kWavesPerWG][IPW][G * kAbPad];
  __shared__ __attribute__((aligned(16))) uint32_t ts_[kWavesPerWG][kWave][kAbPad];
  const int lane = threadIdx.x % kWave, wv = threadIdx.x / kWave;
  const int grp = lane / G, x = lane % G;
  uint32_t* qs = qs_[wv][grp];
  uint32_t* ts = ts_[wv][lane];
  uint32_t sel0 = x == 0 ? 1u : 0u;
  asm("" : "+v"(sel0));
  const size_t wave_id = (size_t)blockIdx.x * kWavesPerWG + wv;
  const size_t first_inst = wave_id * IPW;
  size_t inst = first_inst + grp;
  if (inst >= A.count) inst = A.count - 1;
  uint32_t n[K], a[K], b[K], ma[K], mb[K];
#pragma unroll
  for (int j = 0; j < K; ++j) n[j] = A.ctx.nhat[x * K + j];
  const int w = A.window, tsize = 1 << w;
  uint32_t* tbl = A.table + (wave_id * IPW + grp) * (size_t)tsize * LQ + x * K;   // entry e: a part at e*LQ, b part at e*LQ + L2
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
  {
    const uint32_t* row = A.base_pair + inst * A.base_pair_stride;
    load_pair_row<K>(a, row, x);
    load_pair_row<K>(b, row + L2, x);
  }
#pragma unroll
  for (int j = 0; j < K; ++j) {
    ma[j] = a[j];
    mb[j] = b[j];
    tbl[(size_t)LQ + j] = a[j];
    tbl[(size_t)LQ + L2 + j] = b[j];
    tbl[j] = A.ctx.one[x * K + j];
    tbl[L2 + j] = A.ctx.one[L2 + x * K + j];
  }
#pragma unroll 1
  for (int e = 2; e < tsize; ++e) {
    seq_pairmul<G, K, false, true, true>(a, b, ma, mb, n, 0, sel0, qs, ts);
#pragma unroll
    for (int j = 0; j < K; ++j) {
      tbl[(size_t)e * LQ + j] = a[j];
      tbl[(size_t)e * LQ + L2 + j] = b[j];
    }
  }
  int win = nwin - 2;
  if (nwin > 0) {
    const int d0 = digit(nwin - 1);
    load_table_entry<K>(a, tbl, d0, tsize, LQ, gather);
    load_table_entry<K>(b, tbl + L2, d0, tsize, LQ, gather);
  } else {
#pragma unroll
    for (int j = 0; j < K; ++j) {
      a[j] = A.ctx.one[x * K + j];
      b[j] = A.ctx.one[L2 + x * K + j];
    }
  }
#pragma unroll 1
  for (; nwin > 0 && win >= 0; --win) {
    const int idx = digit(win);
    load_table_entry<K>(ma, tbl, idx, tsize, LQ, gather);   // the entry travels while the squarings run
    load_table_entry<K>(mb, tbl + L2, idx, tsize, LQ, gather);
#pragma unroll 1
    for (int i = 0; i < w; ++i) seq_pairmul<G, K, true, true, true>(a, b, a, b, n, 0, sel0, qs, ts);
    seq_pairmul<G, K, false, true, true>(a, b, ma, mb, n, 0, sel0, qs, ts);
  }
  if (first_inst + grp < A.count) {
    uint32_t* out = A.out_pair + inst * (size_t)LQ;
    store_pair_row<K>(out, a, x);
    store_pair_row<K>(out + L2, b, x);
  }
}

// CT + CT on resident batches (pair_ops_kernel's PO_MUL) with both halves in the same lanes: one general pair product
// per element -- the product in which the paired form leaves half A idle for a third of its multiply-accumulates.
template <int G, int K>
__global__ __launch_bounds__(kWGThreads, 2) void pair_mul_seq_kernel(PairOpsArgs A) {
  constexpr int IPW = kWave / G, L2 = G * K, LQ = 2 * L2;
  raise_wave_priority();
  __shared__ __attribute__((aligned(16))) uint32_t qs_[kWavesPerWG][IPW][G * kAbPad];
  __shared__ __attribute__((aligned(16))) uint32_t ts_[kWavesPerWG][kWave][kAbPad];
  const int lane = threadIdx.x % kWave, wv = threadIdx.x / kWave;
  const int grp = lane / G, x = lane % G;
  uint32_t* qs = qs_[wv][grp];
  uint32_t* ts = ts_[wv][lane];
  uint32_t sel0 = x == 0 ? 1u : 0u;
  asm("" : "+v"(sel0));
  size_t inst = ((size_t)blockIdx.x * kWavesPerWG + wv) * IPW + grp;
  const bool live = inst < A.count;
  if (!live) inst = A.count - 1;
  uint32_t n[K], a[K], b[K], c[K], d[K];
  const uint32_t* ra = A.a + inst * (size_t)LQ;
  const uint32_t* rb = A.b + inst * A.b_stride;
  load_pair_row<K>(a, ra, x);
  load_pair_row<K>(b, ra + L2, x);
  load_pair_row<K>(c, rb, x);
  load_pair_row<K>(d, rb + L2, x);
#pragma unroll
  for (int j = 0; j < K; ++j) n[j] = A.ctx.nhat[x * K + j];
  seq_pairmul<G, K, false, true, true>(a, b, c, d, n, 0, sel0, qs, ts);
  if (live) {
    uint32_t* out = A.out + inst * (size_t)LQ;
    store_pair_row<K>(out, a, x);
    store_pair_row<K>(out + L2, b, x);
  }
}

// DJN encrypt of resident batches (hensel_fb_encrypt_kernel's pair-row exit) with both halves in the same lanes:
// hs^r as nwin-1 general pair products of table entries -- every one of them a product in which the paired form
// leaves half A idle for a third of its multiply-accumulates -- then (1 + n*m) as  b += (-k^-1 * m * a) mod n
// (pair_times_gm without the hand-over between halves).  The table is the one hensel_fb_build_kernel wrote.
template <int G, int K, int MINW = 2>   // (MINW: as hensel_decrypt_seq_kernel)
__global__ __launch_bounds__(kWGThreads, MINW) void hensel_fb_encrypt_seq_kernel(HenselFbArgs A) {
  using HG = Geo<G, K>;
  constexpr int IPW = kWave / G, L2 = G * K, LQ = 2 * L2;
  raise_wave_priority();
  __shared__ __attribute__((aligned(16))) uint32_t qs_[kWavesPerWG][IPW][G * kAbPad];
  __shared__ __attribute__((aligned(16))) uint32_t ts_[kWavesPerWG][kWave][kAbPad];
  __shared__ uint64_t io_[kWavesPerWG][IPW][HG::W64 + 1];
  const int lane = threadIdx.x % kWave, wv = threadIdx.x / kWave;
  const int grp = lane / G, x = lane % G;
  uint32_t* qs = qs_[wv][grp];
  uint32_t* ts = ts_[wv][lane];
  auto& io = io_[wv];
  uint32_t sel0 = x == 0 ? 1u : 0u;
  asm("" : "+v"(sel0));
  const size_t first_inst = ((size_t)blockIdx.x * kWavesPerWG + wv) * IPW;
  size_t inst = first_inst + grp;
  if (inst >= A.count) inst = A.count - 1;
  uint32_t n[K], a[K], b[K], ma[K], mb[K];
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
  auto load_entry = [&](uint32_t (&da)[K], uint32_t (&db)[K], int i) {
    if (A.ct_gather) {   // masked: every entry of the window is read, the address stream does not depend on the digits of r
      const uint32_t* wb = A.table + (size_t)i * tsize * LQ + x * K;
      load_table_entry<K>(da, wb, digit(i), tsize, LQ, true);
      load_table_entry<K>(db, wb + L2, digit(i), tsize, LQ, true);
      return;
    }
    const uint32_t* e = A.table + ((size_t)i * tsize + digit(i)) * LQ;
    load_pair_row<K>(da, e, x);
    load_pair_row<K>(db, e + L2, x);
  };
  load_entry(a, b, 0);
  if (A.nwin > 1) load_entry(ma, mb, 1);
  // the entry of the next step is fetched before the product of this one (latency hidden)
#pragma unroll 1
  for (int i = 1; i < A.nwin; ++i) {
    uint32_t na[K], nb[K];
    if (i + 1 < A.nwin) load_entry(na, nb, i + 1);
    seq_pairmul<G, K, false, true, true>(a, b, ma, mb, n, 0, sel0, qs, ts);
#pragma unroll
    for (int j = 0; j < K; ++j) {
      ma[j] = na[j];
      mb[j] = nb[j];
    }
  }
  // times 1 + n*m under the true modulus n:  b += montmul(montmul(m, gm), a)
  stage_words<HG>(io, A.fm_words, A.fm_stride, 0, A.fm_nwords, first_inst, A.count, 1, lane);
  wave_lds_sync();
  {
    uint32_t mv[K], u[K], v[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
      n[j] = A.ctx.n[x * K + j];
      ma[j] = A.ctx.gm[x * K + j];
      mv[j] = limb_from_words(io[grp], x * K + j);
    }
    montmul_reg<HG, false, false>(u, mv, ma, n, A.ctx.n0inv);
    montmul_reg<HG, false, false>(v, u, a, n, A.ctx.n0inv);
    add_normalise<HG>(b, v);
  }
  if (first_inst + grp < A.count) {
    uint32_t* out = A.out_pair + inst * (size_t)LQ;
    store_pair_row<K>(out, a, x);
    store_pair_row<K>(out + L2, b, x);
  }
}

}  // namespace pgpu

#endif  // PAILLIERCRYPTOLIB_AMD_CSRC_HENSEL_SEQ_HPP_

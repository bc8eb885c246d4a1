// pailliercryptolib_amd -- the split form with a whole exponentiation in ONE lane (round 4): hensel_decrypt_lane_kernel,
// CRT decrypt of large resident batches under keys whose half-width fits a lane's registers (1024-bit keys: 20 limbs).
//
// Every other form of the split exponentiation (hensel.hpp, hensel_seq.hpp) spreads a residue over 2 to 16 lanes, and
// what bounds it is not the multiply-accumulates but the 6-8 instructions per quotient row that MOVE things between
// lanes -- the digit broadcast, the limb hand-over to the lane below, the row broadcast: with K = 10
// limbs per lane (1024-bit keys in the (2,10) forms) a row is 20 products and 8 other instructions, 61-64 % of the
// issue slots are products, and the decrypt leg of that key class ran at 0.45-0.54 of the int-ALU peak.  Here a lane
// holds a AND b of the pair x == a - P*b, all K = L2 limbs of each, and the Montgomery products run entirely inside
// the lane:
//     t = a*c           K^2 products into 2K column accumulators (K(K+1)/2 when it is a squaring), K reduction rows of
//                       K products each: digit = low limb (unit quotient digits), carry = one shift and one add
//     w = a*d + b*c + q the same with the digits of the first reduction entering column by column
// -- no DPP, no LDS, no broadcasts: per row 2K products and 3 other instructions (93 % products at K = 20); per squaring
// and exponentiation 1 410 products in ~1 650 instructions against 2 x 1 104 in hensel_decrypt_seq_kernel<2,10>.  The
// price is the launch size: 64 exponentiations per wavefront, so it takes 32768 ciphertexts to put a wavefront on every
// SIMD (capi.cpp picks the form from there; PGPU_LANE_DECRYPT).  Same entry (pair rows of the n^2 domain), window
// table, fixed-window scan, exit and output as hensel_decrypt_seq_kernel; results bit-identical
// (tests/test_gpu_round4.py::test_lane_decrypt_kernel_is_bit_identical).  Reference: the two half-width exponentiations
// of PrivateKey::decryptCRT, ipcl/pri_key.cpp:114-146.
#ifndef PAILLIERCRYPTOLIB_AMD_CSRC_HENSEL_LANE_HPP_
#define PAILLIERCRYPTOLIB_AMD_CSRC_HENSEL_LANE_HPP_

#include "hensel.hpp"

namespace pgpu {

// c[i + j] += x[j] * y[i].  SYM: y IS x (a squaring): the cross products once, doubled (x2 = 2x), the diagonal once.
template <int K, bool SYM>
__device__ __forceinline__ void lane_mac(uint64_t (&c)[2 * K], const uint32_t (&x)[K], const uint32_t (&x2)[K],
                                         const uint32_t (&y)[K]) {
#pragma unroll
  for (int i = 0; i < K; ++i) {
#pragma unroll
    for (int j = 0; j < K; ++j) {
      if constexpr (SYM) {
        if (j > i) continue;
        c[i + j] += (uint64_t)(j < i ? x2[j] : x[j]) * y[i];
      } else {
        c[i + j] += (uint64_t)x[j] * y[i];
      }
    }
  }
}

// K reduction rows: the low K columns become multiples of 2^29 and carry into the upper K.  QMODE 1: the digits are
// recorded in qd; QMODE 2: qd[r] is added to column r before its digit is taken (the correction term of the pair product).
template <int K, bool UNITQ, int QMODE>
__device__ __forceinline__ void lane_reduce(uint64_t (&c)[2 * K], const uint32_t (&n)[K], uint32_t n0inv, uint32_t (&qd)[K]) {
  uint32_t onev = 1;
  asm("" : "+v"(onev));   // (keeps "+= digit" ONE v_mad_u64_u32 instead of an add / add-with-carry pair)
#pragma unroll
  for (int r = 0; r < K; ++r) {
    if constexpr (QMODE == 2) c[r] += (uint64_t)qd[r] * onev;
    const uint32_t q = quot_digit<UNITQ>(c[r], n0inv) & kLimbMask;
    if constexpr (QMODE == 1) qd[r] = q;
#pragma unroll
    for (int j = 0; j < K; ++j) c[r + j] += (uint64_t)n[j] * q;
    c[r + 1] += c[r] >> kLimbBits;
  }
}

// the upper K columns -> canonical 29-bit limbs (the value is < R: nothing carries out of the top limb)
template <int K>
__device__ __forceinline__ void lane_finish(uint32_t (&r)[K], const uint64_t (&c)[2 * K]) {
  uint64_t carry = 0;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const uint64_t t = c[K + j] + carry;
    r[j] = (uint32_t)t & kLimbMask;
    carry = t >> kLimbBits;
  }
}

// r = x * y * R^-1 mod n (lazy), everything in this lane
template <int K, bool UNITQ>
__device__ __forceinline__ void lane_montmul(uint32_t (&r)[K], const uint32_t (&x)[K], const uint32_t (&y)[K],
                                             const uint32_t (&n)[K], uint32_t n0inv) {
  uint64_t c[2 * K];
#pragma unroll
  for (int j = 0; j < 2 * K; ++j) c[j] = 0;
  uint32_t none[K];
  lane_mac<K, false>(c, x, x, y);
  lane_reduce<K, UNITQ, 0>(c, n, n0inv, none);
  lane_finish<K>(r, c);
}

// (a, b) = (a, b) (x) (c, d): the Montgomery product of two pairs (hensel.hpp), lazy: inputs < 8P -> outputs < 2P.
// A squaring passes c = a, d = b.
template <int K, bool SQR, bool UNITQ>
__device__ __forceinline__ void lane_pairmul(uint32_t (&a)[K], uint32_t (&b)[K], const uint32_t (&c)[K],
                                             const uint32_t (&d)[K], const uint32_t (&n)[K], uint32_t n0inv) {
  static_assert(3 * K < 64, "a column receives up to 3K products of canonical limbs (< 2^58 each; 2K of < 2^59 in a squaring) plus carries: must stay below 2^64");
  uint32_t qd[K], t[K], twice[K];
  {
    uint64_t col[2 * K];
#pragma unroll
    for (int j = 0; j < 2 * K; ++j) col[j] = 0;
#pragma unroll
    for (int j = 0; j < K; ++j) twice[j] = SQR ? a[j] << 1 : 0u;
    lane_mac<K, SQR>(col, a, twice, c);
    lane_reduce<K, UNITQ, 1>(col, n, n0inv, qd);
    lane_finish<K>(t, col);
  }
  {
    uint64_t col[2 * K];
#pragma unroll
    for (int j = 0; j < 2 * K; ++j) col[j] = 0;
    if constexpr (SQR) {
#pragma unroll
      for (int j = 0; j < K; ++j) twice[j] = b[j] << 1;
      lane_mac<K, false>(col, twice, twice, a);    // 2*b times the rows of a
    } else {
      lane_mac<K, false>(col, b, b, c);            // b times the rows of c
      lane_mac<K, false>(col, d, d, a);            // d times the rows of a
    }
    lane_reduce<K, UNITQ, 2>(col, n, n0inv, qd);
    lane_finish<K>(b, col);
  }
#pragma unroll
  for (int j = 0; j < K; ++j) a[j] = t[j];
}

// a += k with full carry propagation (limbs of a and k may be lazy sums below 2^31); the value must stay < R
template <int K>
__device__ __forceinline__ void lane_add(uint32_t (&a)[K], const uint32_t (&k)[K]) {
  uint32_t c = 0;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const uint32_t u = a[j] + k[j] + c;
    a[j] = u & kLimbMask;
    c = u >> kLimbBits;
  }
}
template <int K>
__device__ __forceinline__ void lane_normalise(uint32_t (&a)[K]) {
  uint32_t c = 0;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const uint32_t u = a[j] + c;
    a[j] = u & kLimbMask;
    c = u >> kLimbBits;
  }
}
// d = r - s (canonical limbs) modulo 2^(29K); returns the borrow (1: r < s)
template <int K>
__device__ __forceinline__ uint32_t lane_sub(uint32_t (&d)[K], const uint32_t (&r)[K], const uint32_t (&s)[K]) {
  uint32_t b = 0;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const uint32_t t = r[j] - s[j] - b;
    d[j] = t & kLimbMask;
    b = t >> 31;
  }
  return b;
}

// One wavefront = 64 ciphertexts of ONE side (wave parity: even = p, odd = q).  Output: row 2i = mp, row 2i+1 = mq
// (canonical words) for crt_kernel, like hensel_decrypt_kernel.  K = limbs per half (H * K of the key's split form).
template <int K>
__global__ __launch_bounds__(kWGThreads, 2) void hensel_decrypt_lane_kernel(HenselArgs A) {
  constexpr int IPW = kWave, L2 = K, LQ = 2 * K, W64 = (K * kLimbBits + 63) / 64;
  raise_wave_priority();
  __shared__ uint32_t bl_[kWavesPerWG][IPW][L2];
  const int lane = threadIdx.x % kWave, wv = threadIdx.x / kWave;
  auto& bl = bl_[wv];
  const size_t wave_id = (size_t)blockIdx.x * kWavesPerWG + wv;
  const int side = __builtin_amdgcn_readfirstlane((int)(wave_id & 1));
  const size_t first_elem = (wave_id >> 1) * IPW;
  size_t elem = first_elem + lane;
  if (elem >= A.count) elem = A.count - 1;
#define HCTX(field) (side ? A.ctx[1].field : A.ctx[0].field)
  uint32_t n[K], a[K], b[K], ma[K], mb[K];
#pragma unroll
  for (int j = 0; j < K; ++j) n[j] = HCTX(nhat)[j];
  const int w = A.window, tsize = 1 << w;
  uint32_t* tbl = A.table + (wave_id * IPW + lane) * (size_t)tsize * LQ;   // entry e: a part at e*LQ, b part at e*LQ + L2
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
        const bool in = j < A.pchunk_limbs && first + j < A.pair_l2;
        a[j] = in ? row[first + j] : 0u;
        zb[j] = in ? row[A.pair_l2 + first + j] : 0u;
        b[j] = 0;
        cb[j] = HCTX(pcb)[(size_t)i * L2 + j];
        ma[j] = HCTX(pconv)[(size_t)i * LQ + j];
        mb[j] = HCTX(pconv)[(size_t)i * LQ + L2 + j];
      }
      // (rows written by the multi-lane kernels hold RELAXED limbs -- limb 0 below 2^30, limb 1 below 2^29 + 2^7 per lane
      // of the producer -- the column accumulators have the room, but the in-lane reductions assume canonical limbs)
      lane_normalise<K>(a);
      lane_normalise<K>(zb);
      lane_montmul<K, true>(tb, zb, cb, n, 0);
      lane_pairmul<K, false, true>(a, b, ma, mb, n, 0);
      lane_add<K>(b, tb);
      lane_add<K>(acc_a, a);
      lane_add<K>(acc_b, b);
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
    tbl[j] = HCTX(one)[j];
    tbl[L2 + j] = HCTX(one)[L2 + j];
  }
#pragma unroll 1
  for (int e = 2; e < tsize; ++e) {
    lane_pairmul<K, false, true>(a, b, ma, mb, n, 0);
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
      a[j] = HCTX(one)[j];
      b[j] = HCTX(one)[L2 + j];
    }
  }
  // (entries of the table may hold the "one" constants of the key image: canonical; products leave canonical limbs)
#pragma unroll 1
  for (; nwin > 0 && win >= 0; --win) {
    const int idx = digit(win);
#pragma unroll 1
    for (int i = 0; i < w; ++i) lane_pairmul<K, true, true>(a, b, a, b, n, 0);
    // (the entry is fetched AFTER the squarings: held across them it costs 2K registers and the second wavefront of the
    // SIMD with them -- 280 registers, one wavefront per SIMD, 4.8 cycles per instruction; the neighbour wavefront covers
    // the round trip instead)
    load_table_entry<K>(ma, tbl, idx, tsize, LQ, gather);
    load_table_entry<K>(mb, tbl + L2, idx, tsize, LQ, gather);
    lane_pairmul<K, false, true>(a, b, ma, mb, n, 0);
  }
  // ---- exit under the TRUE prime: (a, k*b mod p) times (hp, 0);  mp = ([a' >= p] - b') mod p ----
  uint32_t np[K];
#pragma unroll
  for (int j = 0; j < K; ++j) {
    np[j] = HCTX(n)[j];
    ma[j] = HCTX(kr)[j];
  }
  const uint32_t n0 = HCTX(n0inv);
  {
    uint32_t kb[K];
    lane_montmul<K, false>(kb, b, ma, np, n0);
#pragma unroll
    for (int j = 0; j < K; ++j) {
      b[j] = kb[j];
      ma[j] = HCTX(h)[j];
      mb[j] = 0;
    }
  }
  lane_pairmul<K, false, false>(a, b, ma, mb, np, n0);
  uint32_t d[K];
  const uint32_t below_a = lane_sub<K>(d, a, np);
  const uint32_t jflag = below_a ^ 1u;
  const uint32_t below = lane_sub<K>(d, b, np);
  if (!below) {
#pragma unroll
    for (int j = 0; j < K; ++j) b[j] = d[j];
  }
  (void)lane_sub<K>(d, np, b);
  d[0] += jflag;
  lane_normalise<K>(d);
  const uint32_t small = lane_sub<K>(b, d, np);
  if (small) {
#pragma unroll
    for (int j = 0; j < K; ++j) b[j] = d[j];
  }
  wave_lds_sync();
#pragma unroll
  for (int j = 0; j < K; ++j) bl[lane][j] = b[j];
  wave_lds_sync();
  const int ow = A.out_words;
  for (int t = lane; t < IPW * ow; t += kWave) {
    const int gg = t / ow, ww = t % ow;
    const size_t oe = first_elem + gg;
    if (oe < A.count) A.out[(2 * oe + side) * A.out_stride + ww] = ww < W64 ? word_from_limbs(bl[gg], L2, ww) : 0;
  }
#undef HCTX
}

}  // namespace pgpu

#endif  // PAILLIERCRYPTOLIB_AMD_CSRC_HENSEL_LANE_HPP_

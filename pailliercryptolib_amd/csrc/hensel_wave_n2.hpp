// pailliercryptolib_amd -- the LATENCY forms for the n^2 domain (round 6): ONE exponentiation per WAVEFRONT on resident pair rows,
// hensel_fb_encrypt_wave_kernel<L2, LPL> (DJN encrypt of small batches: at the end of this file) and
// hensel_modexp_wave_kernel<L2, LPL> -- CipherText * PlainText (ipcl/ciphertext.cpp:83-106, 143-162) of small batches: the
// reference's BM_Mul_CTPT sizes 16 ... 1024 (benchmark/bench_ops.cpp:138-149; one wavefront per element, at most one per SIMD).
//
// The same idea as hensel_wave.hpp (operand scan with the accumulators sliding down the lanes, the two scans of a pair product in
// lock-step, whole-word quotient digits), for halves of L2 limbs that do not fit one limb per lane: lane l holds LPL consecutive
// limbs (2048-bit keys: L2 = 72 limbs of 29 bits as 36 lanes x 2; 3072-bit: 112 as 56 x 2; 1024-bit: 38 x 1).  Nothing is
// converted: the limbs ARE the row's limbs (29 bits, relaxed), the loop modulus is the row's P = n k == -1 (mod 2^29), the
// radix the row's R = 2^(29 L2) -- a row goes in, a row comes out, like hensel_modexp_seq_kernel.  The slide moves the window
// by ONE limb per step: inside a lane limb j takes over limb j+1's low part, the lane's top limb takes the low part of the
// next lane's limb 0 (the one cross-lane move of a step and scan).
// Whole-word (32-bit) digits (hensel_wave.hpp: wv_digit): values stay below 9 P instead of 2 P, sound for R >= 2^10 P -- the pair
// rows' radix leaves 2^11 (2048-bit keys: 2^2088 against P < 2^2077); the host checks per key.  Results are relaxed limbs below
// 2^29 + 2^9, which every consumer of pair rows accepts (kargs.hpp); the kernel's LAST product runs with masked digits, so
// that the row it stores is below 2 P like every other producer's.
// Window table in LDS (2^w entries x 2 x L2 limbs per wavefront); per-element exponents are per-WAVEFRONT here: scalar digits;
// masked access selects under a per-lane compare.
#ifndef PAILLIERCRYPTOLIB_AMD_CSRC_HENSEL_WAVE_N2_HPP_
#define PAILLIERCRYPTOLIB_AMD_CSRC_HENSEL_WAVE_N2_HPP_

#include "hensel_wave.hpp"

namespace pgpu {

template <int LPL>
struct WaveCtxN {
  uint32_t nl[LPL];
  uint32_t e0, maskv, onev;
};

// limb i of a lane-distributed value (LPL limbs per lane) as an SGPR
template <int LPL, int I>
__device__ __forceinline__ uint32_t wvn_limb(const uint32_t (&x)[LPL]) {
  return (uint32_t)__builtin_amdgcn_readlane((int)x[I % LPL], I / LPL);
}

template <int LPL, int LB>
__device__ __forceinline__ void wvn_finish(uint32_t (&r)[LPL], const uint64_t (&acc)[LPL]) {
  r[0] = ((uint32_t)acc[0] & ((1u << LB) - 1)) + wv_up((uint32_t)(acc[LPL - 1] >> LB));
#pragma unroll
  for (int j = 1; j < LPL; ++j) r[j] = ((uint32_t)acc[j] & ((1u << LB) - 1)) + (uint32_t)(acc[j - 1] >> LB);
}

// (a, b) = (a, b) (x) (cm, dm), or the square (SQR: cm = a, dm = b): t = a*cm with its digits; b = a*dm + b*cm + q reduced; the
// two scans in lock-step and in the pinned order of hensel_wave.hpp: wv_pairop (its comment names the steps), every step on
// LPL limbs.  The slide by one limb: limb j <- low(limb j+1) + high(limb j); the lane's top limb takes the next lane's limb 0.
#define WV_PIN __builtin_amdgcn_sched_barrier(0)
template <int L2, int LPL, int LB, bool SQR, bool WIDEQ>
__device__ __forceinline__ void wvn_pairop(uint32_t (&a)[LPL], uint32_t (&b)[LPL], const uint32_t (&cm)[LPL],
                                           const uint32_t (&dm)[LPL], const WaveCtxN<LPL>& c) {
  uint64_t acc1[LPL], acc2[LPL];
  uint32_t m1[LPL], m2[LPL], a0[LPL], b0[LPL], lo1[LPL], lo2[LPL];
#pragma unroll
  for (int j = 0; j < LPL; ++j) {
    acc1[j] = acc2[j] = 0;
    a0[j] = a[j];
    b0[j] = b[j];
    m1[j] = SQR ? a[j] : cm[j];
    m2[j] = SQR ? b[j] << 1 : dm[j];
    lo1[j] = lo2[j] = 0;
  }
  uint32_t sa[L2 + 2], sb[L2 + 2];                               // the limbs of a (and b) as SGPRs, fetched two steps ahead
  sa[0] = wvn_limb<LPL, 0>(a0);
  sa[1] = wvn_limb<LPL, (L2 > 1 ? 1 : 0)>(a0);
  if constexpr (!SQR) {
    sb[0] = wvn_limb<LPL, 0>(b0);
    sb[1] = wvn_limb<LPL, (L2 > 1 ? 1 : 0)>(b0);
  }
  WV_PIN;
#pragma unroll
  for (int j = 0; j < LPL; ++j) wv_mac(acc1[j], sa[0], m1[j]);   // A1 of step 0
  WV_PIN;
  ps_static_for<L2>([&](auto ic) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    uint32_t q1, t1 = 0;
    if constexpr (WIDEQ) {
      q1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)acc1[0]);                       // R1
      asm volatile("" : "+v"(acc2[0]) : "s"(q1));       // (what follows on acc2 stays behind the broadcast: it is its filler)
    } else {
      t1 = (uint32_t)acc1[0] & c.maskv;
    }
    WV_PIN;
    if constexpr (i > 0) {
#pragma unroll
      for (int j = 0; j < LPL; ++j) wv_mac(acc2[j], lo2[(j + 1) % LPL], c.onev);                   // S2c of step i-1
      WV_PIN;
    }
    if constexpr (!WIDEQ) {
      q1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)t1);
      asm volatile("" : "+v"(acc2[0]) : "s"(q1));
      WV_PIN;
    }
#pragma unroll
    for (int j = 0; j < LPL; ++j) wv_mac(acc2[j], sa[i], m2[j]);                                   // A2
    WV_PIN;
    if constexpr (!SQR) {
#pragma unroll
      for (int j = 0; j < LPL; ++j) wv_mac(acc2[j], sb[i], m1[j]);
      WV_PIN;
    }
    wv_mac(acc2[0], q1, c.e0);                                                                     // B2
    WV_PIN;
#pragma unroll
    for (int j = 0; j < LPL; ++j) wv_mac(acc1[j], q1, c.nl[j]);                                    // B1
    WV_PIN;
    uint32_t q2, t2 = 0;
    if constexpr (WIDEQ) {
      q2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)acc2[0]);                       // R2
      asm volatile("" : "+v"(acc1[0]) : "s"(q2));
    } else {
      t2 = (uint32_t)acc2[0] & c.maskv;
    }
    WV_PIN;
    if constexpr (i + 2 < L2) {                                                                    // RL
      sa[i + 2] = wvn_limb<LPL, (i + 2 < L2 ? i + 2 : 0)>(a0);
      if constexpr (!SQR) sb[i + 2] = wvn_limb<LPL, (i + 2 < L2 ? i + 2 : 0)>(b0);
    }
    WV_PIN;
    if constexpr (!WIDEQ) {
      q2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)t2);
      asm volatile("" : "+v"(acc1[0]) : "s"(q2));
      WV_PIN;
    }
    lo1[0] = wv_down_and((uint32_t)acc1[0], c.maskv);                                              // S1a
#pragma unroll
    for (int j = 1; j < LPL; ++j) lo1[j] = (uint32_t)acc1[j] & c.maskv;
    WV_PIN;
#pragma unroll
    for (int j = 0; j < LPL; ++j) acc1[j] >>= LB;                                                  // S1b
    WV_PIN;
#pragma unroll
    for (int j = 0; j < LPL; ++j) wv_mac(acc2[j], q2, c.nl[j]);                                    // C2
    WV_PIN;
#pragma unroll
    for (int j = 0; j < LPL; ++j) wv_mac(acc1[j], lo1[(j + 1) % LPL], c.onev);                     // S1c
    WV_PIN;
    if constexpr (i + 1 < L2) {
#pragma unroll
      for (int j = 0; j < LPL; ++j) wv_mac(acc1[j], sa[i + 1], m1[j]);                             // A1 of step i+1
      WV_PIN;
    }
    lo2[0] = wv_down_and((uint32_t)acc2[0], c.maskv);                                              // S2a
#pragma unroll
    for (int j = 1; j < LPL; ++j) lo2[j] = (uint32_t)acc2[j] & c.maskv;
    WV_PIN;
#pragma unroll
    for (int j = 0; j < LPL; ++j) acc2[j] >>= LB;                                                  // S2b
    WV_PIN;
  });
#pragma unroll
  for (int j = 0; j < LPL; ++j) wv_mac(acc2[j], lo2[(j + 1) % LPL], c.onev);                       // S2c of the last step
  wvn_finish<LPL, LB>(a, acc1);
  wvn_finish<LPL, LB>(b, acc2);
}
#undef WV_PIN

// 32-bit words of LDS table per wavefront: entry e, part (a / b), limb slot j, lane
template <int LPL>
constexpr size_t wvn_table_words(size_t entries) { return entries * 2 * LPL * kWave; }

// One wavefront = ONE element.  Pair rows of 2*L2 29-bit limbs in (A.base_pair; stride 0: one shared row) and out (A.out_pair);
// A.ctx.nhat / A.ctx.one in 29-bit limbs; per-element exponents (A.exp_stride > 0) or one shared one; fixed window A.window.
// Dynamic LDS: kWavesPerWG * wvn_table_words<LPL>(2^A.window) * 4 bytes.
template <int L2, int LPL, bool WIDEQ>
__global__ __launch_bounds__(kWGThreads, 1) void hensel_modexp_wave_kernel(HenselModexpArgs A) {
  constexpr int LB = kLimbBits, NL = L2 / LPL;
  static_assert(L2 % LPL == 0 && NL < kWave, "LPL limbs per lane and a zero lane above them");
  raise_wave_priority();
  extern __shared__ uint32_t wvn_tbl_[];
  const int lane = threadIdx.x % kWave, wv = threadIdx.x / kWave;
  const size_t inst = (size_t)blockIdx.x * kWavesPerWG + wv;
  if (inst >= A.count) return;                       // (wave-uniform)
  const bool in = lane < NL;
  const int lk = in ? lane : 0;
  WaveCtxN<LPL> c;
#pragma unroll
  for (int j = 0; j < LPL; ++j) c.nl[j] = in ? A.ctx.nhat[lk * LPL + j] : 0u;
  c.e0 = lane == 0 ? 1u : 0u;
  c.maskv = (1u << LB) - 1;
  c.onev = 1;
  asm("" : "+v"(c.maskv), "+v"(c.onev), "+v"(c.e0));
  const int w = A.window, tsize = 1 << w;
  uint32_t* tbl = wvn_tbl_ + (size_t)wv * wvn_table_words<LPL>((size_t)tsize) + lane;
  const uint64_t* ep = A.exp + inst * A.exp_stride;
  const int nwin = (A.exp_bits + w - 1) / w;
  auto digit = [&](int i) -> int {
    int bit = i * w;
    int word = bit >> 6, sh = bit & 63;
    uint64_t v = (word < A.exp_words) ? ep[word] >> sh : 0;
    if (sh + w > 64 && word + 1 < A.exp_words) v |= ep[word + 1] << (64 - sh);
    return __builtin_amdgcn_readfirstlane((int)(v & (uint64_t)(tsize - 1)));
  };
  const bool gather = A.ct_gather != 0;
  auto entry_load = [&](uint32_t (&x)[LPL], uint32_t (&y)[LPL], int e) {
    if (!gather) {
#pragma unroll
      for (int j = 0; j < LPL; ++j) {
        x[j] = tbl[((size_t)e * 2 * LPL + j) * kWave];
        y[j] = tbl[((size_t)e * 2 * LPL + LPL + j) * kWave];
      }
      return;
    }
    uint32_t ev = (uint32_t)e;          // (a VGPR copy of the digit: per-lane compare and select, never a branch on it)
    asm("" : "+v"(ev));
#pragma unroll
    for (int j = 0; j < LPL; ++j) x[j] = y[j] = 0;
    for (int t = 0; t < tsize; ++t) {
      const bool sel = ev == (uint32_t)t;
#pragma unroll
      for (int j = 0; j < LPL; ++j) {
        const uint32_t tx = tbl[((size_t)t * 2 * LPL + j) * kWave], ty = tbl[((size_t)t * 2 * LPL + LPL + j) * kWave];
        x[j] = sel ? tx : x[j];
        y[j] = sel ? ty : y[j];
      }
    }
  };
  auto entry_store = [&](int e, const uint32_t (&x)[LPL], const uint32_t (&y)[LPL]) {
#pragma unroll
    for (int j = 0; j < LPL; ++j) {     // (lanes above the limbs store their zeros: loads need no lane test)
      tbl[((size_t)e * 2 * LPL + j) * kWave] = x[j];
      tbl[((size_t)e * 2 * LPL + LPL + j) * kWave] = y[j];
    }
  };
  uint32_t a[LPL], b[LPL], ba[LPL], bb[LPL];
  {
    const uint32_t* row = A.base_pair + inst * A.base_pair_stride;
#pragma unroll
    for (int j = 0; j < LPL; ++j) {
      a[j] = ba[j] = in ? row[lk * LPL + j] : 0u;
      b[j] = bb[j] = in ? row[L2 + lk * LPL + j] : 0u;
    }
  }
  // ---- window table: entry 0 = one, entry 1 = base, entry e = entry e-1 times base ----
  entry_store(1, a, b);
  {
    uint32_t oa[LPL], ob[LPL];
#pragma unroll
    for (int j = 0; j < LPL; ++j) {
      oa[j] = in ? A.ctx.one[lk * LPL + j] : 0u;
      ob[j] = in ? A.ctx.one[L2 + lk * LPL + j] : 0u;
    }
    entry_store(0, oa, ob);
  }
#pragma unroll 1
  for (int e = 2; e < tsize; ++e) {
    wvn_pairop<L2, LPL, LB, false, WIDEQ>(a, b, ba, bb, c);
    entry_store(e, a, b);
  }
  // ---- main loop: w squarings, one multiplication by a table entry (always, also entry 0 = one); the LAST product with
  // masked digits, so that the row it leaves is below 2 P like every other producer's ----
  entry_load(a, b, nwin > 0 ? digit(nwin - 1) : 0);
  uint32_t ma[LPL], mb[LPL];
#pragma unroll 1
  for (int win = nwin - 2; win >= 1; --win) {
    const int d = digit(win);
#pragma unroll 1
    for (int i = 0; i < w; ++i) wvn_pairop<L2, LPL, LB, true, WIDEQ>(a, b, a, b, c);
    entry_load(ma, mb, d);
    wvn_pairop<L2, LPL, LB, false, WIDEQ>(a, b, ma, mb, c);
  }
  if (nwin >= 2) {
    const int d = digit(0);
#pragma unroll 1
    for (int i = 0; i < w; ++i) wvn_pairop<L2, LPL, LB, true, WIDEQ>(a, b, a, b, c);
    entry_load(ma, mb, d);
  } else {                              // (a single window: the entry itself, times one)
    entry_load(ma, mb, 0);
  }
  wvn_pairop<L2, LPL, LB, false, false>(a, b, ma, mb, c);
  if (in) {
    uint32_t* out = A.out_pair + inst * (size_t)(2 * L2);
#pragma unroll
    for (int j = 0; j < LPL; ++j) {
      out[lk * LPL + j] = a[j];
      out[L2 + lk * LPL + j] = b[j];
    }
  }
}

// r = x*y*R^-1 mod n under the TRUE modulus n (not == -1 mod 2^29: the digit takes a multiply by n0inv), one scan; x, y, r
// lane-distributed.  Used twice per element by the encrypt exit below: no lock-step partner, the scheduler's order.
template <int L2, int LPL, int LB>
__device__ __forceinline__ void wvn_montmul_true(uint32_t (&r)[LPL], const uint32_t (&x)[LPL], const uint32_t (&y)[LPL],
                                                 const uint32_t (&n)[LPL], uint32_t n0inv, uint32_t maskv, uint32_t onev) {
  uint64_t acc[LPL];
#pragma unroll
  for (int j = 0; j < LPL; ++j) acc[j] = 0;
  ps_static_for<L2>([&](auto ic) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    const uint32_t sx = wvn_limb<LPL, i>(x);
#pragma unroll
    for (int j = 0; j < LPL; ++j) wv_mac(acc[j], sx, y[j]);
    const uint32_t q = (uint32_t)__builtin_amdgcn_readfirstlane((int)(((uint32_t)acc[0] * n0inv) & maskv));
#pragma unroll
    for (int j = 0; j < LPL; ++j) wv_mac(acc[j], q, n[j]);
    uint32_t lo[LPL];
    lo[0] = wv_down_and((uint32_t)acc[0], maskv);
#pragma unroll
    for (int j = 1; j < LPL; ++j) lo[j] = (uint32_t)acc[j] & maskv;
#pragma unroll
    for (int j = 0; j < LPL; ++j) {
      acc[j] >>= LB;
      wv_mac(acc[j], lo[(j + 1) % LPL], onev);
    }
  });
  wvn_finish<LPL, LB>(r, acc);
}

// DJN encrypt of small batches on the fixed-base table of pairs (hensel.hpp: hensel_fb_encrypt_kernel; PublicKey::encrypt,
// ipcl/pub_key.cpp:51-64, 88-105): ONE wavefront per element.  hs^r = the product of one table entry per w-bit digit of r --
// nwin - 1 pair products, the entry of the next step fetched while this one's product runs --, then times g^m = 1 + n*m as
// two half-width products under the true modulus n (hensel.hpp: pair_times_gm: only b changes, b += (-k^-1 m a) mod n);
// the result leaves as a pair row (A.out_pair).  The last pair product runs with masked digits (rows below 2 P).
template <int L2, int LPL, bool WIDEQ>
__global__ __launch_bounds__(kWGThreads, 1) void hensel_fb_encrypt_wave_kernel(HenselFbArgs A) {
  constexpr int LB = kLimbBits, NL = L2 / LPL;
  static_assert(L2 % LPL == 0 && NL < kWave, "LPL limbs per lane and a zero lane above them");
  raise_wave_priority();
  const int lane = threadIdx.x % kWave, wv = threadIdx.x / kWave;
  const size_t inst = (size_t)blockIdx.x * kWavesPerWG + wv;
  if (inst >= A.count) return;                       // (wave-uniform)
  const bool in = lane < NL;
  const int lk = in ? lane : 0;
  WaveCtxN<LPL> c;
#pragma unroll
  for (int j = 0; j < LPL; ++j) c.nl[j] = in ? A.ctx.nhat[lk * LPL + j] : 0u;
  c.e0 = lane == 0 ? 1u : 0u;
  c.maskv = (1u << LB) - 1;
  c.onev = 1;
  asm("" : "+v"(c.maskv), "+v"(c.onev), "+v"(c.e0));
  const int w = A.w, tsize = 1 << w;
  const uint64_t* ep = A.exp + inst * A.exp_stride;
  auto digit = [&](int i) -> int {
    int bit = i * w;
    int word = bit >> 6, sh = bit & 63;
    uint64_t v = (word < A.exp_words) ? ep[word] >> sh : 0;
    if (sh + w > 64 && word + 1 < A.exp_words) v |= ep[word + 1] << (64 - sh);
    return __builtin_amdgcn_readfirstlane((int)(v & (uint64_t)(tsize - 1)));
  };
  const bool gather = A.ct_gather != 0;
  auto load_entry = [&](uint32_t (&x)[LPL], uint32_t (&y)[LPL], int i) {
    const uint32_t* win = A.table + (size_t)i * tsize * (2 * L2);
    const int d = digit(i);
    if (!gather) {
      const uint32_t* row = win + (size_t)d * (2 * L2);
#pragma unroll
      for (int j = 0; j < LPL; ++j) {
        x[j] = in ? row[lk * LPL + j] : 0u;
        y[j] = in ? row[L2 + lk * LPL + j] : 0u;
      }
      return;
    }
    // masked: every entry of the window is read, the wanted one selected under a per-lane compare (no address, no branch
    // follows the digit of r)
    uint32_t dv = (uint32_t)d;
    asm("" : "+v"(dv));
#pragma unroll
    for (int j = 0; j < LPL; ++j) x[j] = y[j] = 0;
    for (int t = 0; t < tsize; ++t) {
      const uint32_t* row = win + (size_t)t * (2 * L2);
      const bool sel = dv == (uint32_t)t;
#pragma unroll
      for (int j = 0; j < LPL; ++j) {
        const uint32_t tx = in ? row[lk * LPL + j] : 0u, ty = in ? row[L2 + lk * LPL + j] : 0u;
        x[j] = sel ? tx : x[j];
        y[j] = sel ? ty : y[j];
      }
    }
  };
  uint32_t a[LPL], b[LPL], ma[LPL], mb[LPL], na[LPL], nb[LPL];
  load_entry(a, b, 0);
  if (A.nwin > 1) load_entry(ma, mb, 1);
#pragma unroll 1
  for (int i = 1; i + 1 < A.nwin; ++i) {
    load_entry(na, nb, i + 1);                      // (in flight while the product runs)
    wvn_pairop<L2, LPL, LB, false, WIDEQ>(a, b, ma, mb, c);
#pragma unroll
    for (int j = 0; j < LPL; ++j) {
      ma[j] = na[j];
      mb[j] = nb[j];
    }
  }
  if (A.nwin > 1) wvn_pairop<L2, LPL, LB, false, false>(a, b, ma, mb, c);
  // ---- times g^m: b += (-k^-1 m a) mod n, two products under the true modulus ----
  {
    uint32_t nt[LPL], gm[LPL], mv[LPL], u[LPL], v[LPL];
    const uint64_t* mw = A.fm_words + inst * A.fm_stride;
#pragma unroll
    for (int j = 0; j < LPL; ++j) {
      nt[j] = in ? A.ctx.n[lk * LPL + j] : 0u;
      gm[j] = in ? A.ctx.gm[lk * LPL + j] : 0u;
      const int bit = (lk * LPL + j) * LB, word = bit >> 6, sh = bit & 63;
      uint64_t val = (in && word < A.fm_nwords) ? mw[word] >> sh : 0;
      if (in && sh > 64 - LB && word + 1 < A.fm_nwords) val |= mw[word + 1] << (64 - sh);
      mv[j] = (uint32_t)val & c.maskv;
    }
    wvn_montmul_true<L2, LPL, LB>(u, mv, gm, nt, A.ctx.n0inv, c.maskv, c.onev);
    wvn_montmul_true<L2, LPL, LB>(v, u, a, nt, A.ctx.n0inv, c.maskv, c.onev);
    // b += v, then one carry round: limbs back below 2^29 + 2 (what the readers of pair rows are sized for)
    uint32_t cy[LPL];
#pragma unroll
    for (int j = 0; j < LPL; ++j) {
      b[j] += v[j];
      cy[j] = b[j] >> LB;
      b[j] &= c.maskv;
    }
    b[0] += wv_up(cy[LPL - 1]);
#pragma unroll
    for (int j = 1; j < LPL; ++j) b[j] += cy[j - 1];
  }
  if (in) {
    uint32_t* out = A.out_pair + inst * (size_t)(2 * L2);
#pragma unroll
    for (int j = 0; j < LPL; ++j) {
      out[lk * LPL + j] = a[j];
      out[L2 + lk * LPL + j] = b[j];
    }
  }
}

}  // namespace pgpu

#endif  // PAILLIERCRYPTOLIB_AMD_CSRC_HENSEL_WAVE_N2_HPP_

// pailliercryptolib_amd -- the LATENCY form of the CRT-decrypt exponentiation (round 6): ONE exponentiation per WAVEFRONT, one
// limb per lane.  hensel_decrypt_wave_kernel<K, LB> with hensel_ps_entry_kernel / hensel_ps_exit_kernel around it; CRT decrypt of
// SMALL batches (up to 512 ciphertexts: 2 x count wavefronts, at most one per SIMD) -- the reference's BM_Decrypt sizes 16 ... 512,
// benchmark/bench_cryptography.cpp:10-19.  The two half-width exponentiations of PrivateKey::decryptCRT, ipcl/pri_key.cpp:114-146.
//
// A launch of fewer wavefronts than SIMDs lasts as long as ONE exponentiation's serial chain: 1024 pair squarings + 235 pair
// products for a 2048-bit key.  The multi-lane latency forms (hensel_decrypt_kernel<8,5>: 16 lanes per exponentiation) spend
// 1.94 us per pair squaring -- block-serial reduction rows with cross-lane digit broadcasts and limb hand-overs.  Here a
// residue's half is spread over the WHOLE wavefront, lane l holding limb l (K <= 63 of the 64 lanes), and a Montgomery product is
// the textbook operand scan with the accumulator sliding down the lanes:
//     step i:   acc_l += a_i * b_l                      a_i wave-uniform: an SGPR (v_readlane, once per product and limb)
//               q = acc_0 mod 2^LB                      v_and + v_readfirstlane: the digit is an SGPR as well (P == -1: no multiply)
//               acc_l += q * P_l                        lane 0's low limb becomes zero
//               acc_l = (acc_{l+1} mod 2^LB) + (acc_l >> LB)       one v_lshrrev_b64, one v_and_b32_dpp wave_shl:1, one multiply-add
// -- six or seven instructions per step whatever K is, K steps per half-width product, no LDS, no waiting for memory; the two
// scans of a pair product run in lock-step and in a pinned order (wv_pairop), so that each one's digit broadcast is covered by
// the other's instructions.  The slide both moves the window and keeps every accumulator below 2^37 (each lane passes its own
// carry one column up while it takes over its neighbour's low limb), so limbs need not be canonical anywhere: products leave
// RELAXED limbs (below 2^LB + 2^9; with 32-bit digits -- wv_digit -- values stay below 17 P instead of 2 P).  The pair product
// of hensel.hpp on top of it: t = a*c with its digits; w = a*d + b*c + q with digit q_i added to lane 0 in step i.  550
// instructions per pair squaring at K = 38 against ~900 of the 16-lane form, on a chain without LDS round trips.
// The same constants as hensel_decrypt_ps_kernel (the key's hs_ps set: K limbs of LB bits, P == -1 mod 2^LB), and its entry
// and exit CODE: hensel_ps_entry_kernel runs the products of ps_entry_from_pair_row one lane per exponentiation and product and
// leaves the partial pairs in a buffer, hensel_ps_exit_kernel picks the result up, makes its limbs canonical and runs
// ps_exit_words -- ~40 us of one-lane work around 1.4 ms, in exchange for not restating either in the limb-per-lane layout.
// The window table lives in LDS (2^w entries x 2 x K limbs per wavefront); the exponent is the side's secret p-1 / q-1, the
// same for every wavefront of a side: its digits are scalar, and under the masked-access policy every entry is read and the
// wanted one selected with a per-lane compare (no branch on the digit).
#ifndef PAILLIERCRYPTOLIB_AMD_CSRC_HENSEL_WAVE_HPP_
#define PAILLIERCRYPTOLIB_AMD_CSRC_HENSEL_WAVE_HPP_

#include "hensel_ps.hpp"

namespace pgpu {

// lane l <- lane l+1 of the wavefront (lane 63 <- 0), ANDed with a mask held in a VGPR: ONE v_and_b32_dpp wave_shl:1
__device__ __forceinline__ uint32_t wv_down_and(uint32_t x, uint32_t maskv) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x130 /* wave_shl:1 */, 0xf, 0xf, true) & maskv;
}
// lane l <- lane l-1 (lane 0 <- 0): v_mov_b32_dpp wave_shr:1
__device__ __forceinline__ uint32_t wv_up(uint32_t x) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
}

// what every lane of the kernel carries along: its limb of the loop modulus, the lane-0 indicator, the limb mask and a one
// (VGPRs: so that "+= 32-bit value" stays one v_mad_u64_u32 and the AND fuses into the DPP move)
struct WaveCtx {
  uint32_t nl, e0, maskv, onev;
};

template <int K>
__device__ __forceinline__ void wv_bcast_limbs(uint32_t (&s)[K], uint32_t x) {
  ps_static_for<K>([&](auto ic) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    s[i] = (uint32_t)__builtin_amdgcn_readlane((int)x, i);
  });
}

// acc += x * y as a link of a chain the optimiser must leave in this order (hensel_ps.hpp: ps_mac_pinned): left alone it
// starts every step's sum from zero and adds the slid accumulator last -- one more 64-bit add per step
__device__ __forceinline__ void wv_mac(uint64_t& acc, uint32_t x, uint32_t y) {
  acc += (uint64_t)x * y;
  asm volatile("" ::"v"(acc));
}
// the digit of a step: lane 0's low limb, masked in the VALU and broadcast through an SGPR (no scalar-ALU hop in the chain).
// WIDEQ: lane 0's whole low WORD is the digit.  Any q == acc_0 (mod 2^LB) clears lane 0's low limb (P_0 = 2^LB - 1), and the
// bits above LB only add a larger multiple of P: with 32-bit digits a product stays below (2^(32-LB) + 1) P instead of 2P
// -- sound as long as R >= 2^10 P (the host checks: capi.cpp decrypt_on), and one v_and less in every step of both scans.
template <int LB, bool WIDEQ>
__device__ __forceinline__ uint32_t wv_digit(uint64_t acc, const WaveCtx& c) {
  if constexpr (WIDEQ) return (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)acc);
  else return (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)acc & c.maskv));
}
// accumulators -> relaxed limbs: own low limb plus the carry of the lane below
template <int LB>
__device__ __forceinline__ uint32_t wv_finish(uint64_t acc) {
  return ((uint32_t)acc & PsLimb<LB>::mask) + wv_up((uint32_t)(acc >> LB));
}

// (a, b) = (a, b) (x) (cm, dm), or the square (SQR: cm = a, dm = b): t = a*cm with its digits; b = a*dm + b*cm + q reduced.
// The two scans run in LOCK-STEP -- step i of the second needs digit i of the first and nothing else of it --, so that one
// chain's broadcast round trip (VALU -> SGPR -> VALU) is covered by the other chain's instructions: a lone wavefront has
// nobody else to issue from.  And in a PINNED order (a scheduling barrier behind every instruction, the first product of
// step i+1 pulled into step i): gfx950 wants two instructions between a VALU write of an SGPR and a VALU read of it, and
// between a VALU write of a VGPR and a DPP read of it (one before a v_readfirstlane); in this order every such pair has them --
// 14 instructions per step and pair, the broadcast of the limb a_(i+2) among them as a filler (16 with the b_i broadcast of a
// general product), no s_nop.  Left to the scheduler: 14 plus 3 s_nop.
#define WV_PIN __builtin_amdgcn_sched_barrier(0)
template <int K, int LB, bool SQR, bool WIDEQ>
__device__ __forceinline__ void wv_pairop(uint32_t& a, uint32_t& b, uint32_t cm, uint32_t dm, const WaveCtx& c) {
  const uint32_t m1 = SQR ? a : cm, m2 = SQR ? b << 1 : dm, a0 = a, b0 = b;
  uint32_t sa[K + 2];                                            // the limbs of a as SGPRs, fetched two steps ahead (RL below)
  sa[0] = (uint32_t)__builtin_amdgcn_readlane((int)a0, 0);
  sa[1] = (uint32_t)__builtin_amdgcn_readlane((int)a0, K > 1 ? 1 : 0);
  uint32_t sb[K + 2];                                            // ... and of b (general product)
  if constexpr (!SQR) {
    sb[0] = (uint32_t)__builtin_amdgcn_readlane((int)b0, 0);
    sb[1] = (uint32_t)__builtin_amdgcn_readlane((int)b0, K > 1 ? 1 : 0);
  }
  uint64_t acc1 = 0, acc2 = 0;
  uint32_t lo2 = 0;
  WV_PIN;
  wv_mac(acc1, sa[0], m1);                                       // A1 of step 0
  WV_PIN;
  ps_static_for<K>([&](auto ic) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    uint32_t q1;
    if constexpr (WIDEQ) {
      q1 = wv_digit<LB, true>(acc1, c);                          // R1
      WV_PIN;
      if constexpr (i > 0) {
        wv_mac(acc2, lo2, c.onev);                               // S2c of step i-1
        WV_PIN;
      }
    } else {                                                     // (masked digits: the v_and one instruction ahead of the broadcast)
      const uint32_t t1 = (uint32_t)acc1 & c.maskv;
      WV_PIN;
      if constexpr (i > 0) {
        wv_mac(acc2, lo2, c.onev);
        WV_PIN;
      }
      q1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)t1);
      WV_PIN;
    }
    wv_mac(acc2, sa[i], m2);                                     // A2
    WV_PIN;
    if constexpr (!SQR) {
      wv_mac(acc2, sb[i], m1);
      WV_PIN;
    }
    wv_mac(acc2, q1, c.e0);                                      // B2
    WV_PIN;
    wv_mac(acc1, q1, c.nl);                                      // B1
    WV_PIN;
    uint32_t q2;
    if constexpr (WIDEQ) {
      q2 = wv_digit<LB, true>(acc2, c);                          // R2
      WV_PIN;
      if constexpr (i + 2 < K) sa[i + 2] = (uint32_t)__builtin_amdgcn_readlane((int)a0, i + 2);   // RL: fills the slot a DPP
      WV_PIN;                                                    //     read of acc1 needs behind B1
    } else {
      const uint32_t t2 = (uint32_t)acc2 & c.maskv;
      WV_PIN;
      if constexpr (i + 2 < K) sa[i + 2] = (uint32_t)__builtin_amdgcn_readlane((int)a0, i + 2);
      WV_PIN;
      q2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)t2);
      WV_PIN;
    }
    if constexpr (!SQR && i + 2 < K) {
      sb[i + 2] = (uint32_t)__builtin_amdgcn_readlane((int)b0, i + 2);
      WV_PIN;
    }
    const uint32_t lo1 = wv_down_and((uint32_t)acc1, c.maskv);   // S1a
    WV_PIN;
    acc1 >>= LB;                                                 // S1b
    WV_PIN;
    wv_mac(acc2, q2, c.nl);                                      // C2
    WV_PIN;
    wv_mac(acc1, lo1, c.onev);                                   // S1c
    WV_PIN;
    if constexpr (i + 1 < K) {
      wv_mac(acc1, sa[i + 1], m1);                               // A1 of step i+1
      WV_PIN;
    }
    lo2 = wv_down_and((uint32_t)acc2, c.maskv);                  // S2a
    WV_PIN;
    acc2 >>= LB;                                                 // S2b
    WV_PIN;
  });
  wv_mac(acc2, lo2, c.onev);                                     // S2c of the last step
  a = wv_finish<LB>(acc1);
  b = wv_finish<LB>(acc2);
}
#undef WV_PIN
template <int K, int LB, bool WIDEQ>
__device__ __forceinline__ void wv_pairsqr(uint32_t& a, uint32_t& b, const WaveCtx& c) {
  wv_pairop<K, LB, true, WIDEQ>(a, b, a, b, c);
}
template <int K, int LB, bool WIDEQ>
__device__ __forceinline__ void wv_pairmul(uint32_t& a, uint32_t& b, uint32_t cm, uint32_t dm, const WaveCtx& c) {
  wv_pairop<K, LB, false, WIDEQ>(a, b, cm, dm, c);
}

// 32-bit words of pair buffer per exponentiation (entry kernel -> wave kernel -> exit kernel): a then b, K limbs each
template <int K>
constexpr size_t wv_pair_words() { return 2 * (size_t)K; }
// 32-bit words of LDS table per wavefront
template <int K>
constexpr size_t wv_table_words(size_t entries) { return entries * 2 * (size_t)K; }

// One wavefront = ONE exponentiation: wavefront 2*i + side serves ciphertext i under side (0: p, 1: q).
// A.table: the pair buffer ([2*count][2 * pchunks roles][2][K] 32-bit limbs of LB bits): the entry's partial pairs on entry,
// the result on exit in slot 0 (relaxed limbs).
// Dynamic LDS: kWavesPerWG * wv_table_words<K>(2^A.window) * 4 bytes.
template <int K, int LB, bool WIDEQ>
__global__ __launch_bounds__(kWGThreads, 1) void hensel_decrypt_wave_kernel(HenselArgs A) {
  static_assert(K < kWave, "one limb per lane and a zero lane above them");
  raise_wave_priority();
  extern __shared__ uint32_t wv_tbl_[];
  const int lane = threadIdx.x % kWave, wv = threadIdx.x / kWave;
  const size_t idx = (size_t)blockIdx.x * kWavesPerWG + wv;
  if (idx >= 2 * A.count) return;                       // (wave-uniform)
  const int side = __builtin_amdgcn_readfirstlane((int)(idx & 1));
#define HCTX(field) (side ? A.ctx[1].field : A.ctx[0].field)
  const bool in = lane < K;
  const int lk = in ? lane : 0;
  WaveCtx c;
  c.nl = in ? HCTX(nhat)[lk] : 0u;
  c.e0 = lane == 0 ? 1u : 0u;
  c.maskv = PsLimb<LB>::mask;
  c.onev = 1;
  asm("" : "+v"(c.maskv), "+v"(c.onev), "+v"(c.e0));
  const int w = A.window, tsize = 1 << w;
  uint32_t* tbl = wv_tbl_ + (size_t)wv * wv_table_words<K>((size_t)tsize);     // entry e: a limbs at e*2K, b limbs at e*2K + K
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
  auto entry_load = [&](uint32_t& x, uint32_t& y, int e) {
    if (!gather) {
      x = in ? tbl[(size_t)e * 2 * K + lk] : 0u;
      y = in ? tbl[(size_t)e * 2 * K + K + lk] : 0u;
      return;
    }
    // masked access: every entry read, the wanted one selected under a PER-LANE compare (a VGPR copy of the digit: a
    // v_cmp / v_cndmask pair per entry, never a scalar branch on the secret digit)
    uint32_t ev = (uint32_t)e;
    asm("" : "+v"(ev));
    x = y = 0;
    for (int t = 0; t < tsize; ++t) {
      const uint32_t tx = in ? tbl[(size_t)t * 2 * K + lk] : 0u, ty = in ? tbl[(size_t)t * 2 * K + K + lk] : 0u;
      const bool sel = ev == (uint32_t)t;
      x = sel ? tx : x;
      y = sel ? ty : y;
    }
  };
  auto entry_store = [&](int e, uint32_t x, uint32_t y) {
    if (in) {
      tbl[(size_t)e * 2 * K + lk] = x;
      tbl[(size_t)e * 2 * K + K + lk] = y;
    }
  };
  // the base: the entry kernel leaves one partial pair per ROLE (chunk of the row x {pair product of its a half, single product
  // of its b half}); their sum is c*R -- limbs simply add up (relaxed limbs are fine here), the result goes back into slot 0
  const int nroles = 2 * A.pchunks;
  uint32_t* buf = A.table + idx * (size_t)nroles * wv_pair_words<K>();
  uint32_t a = 0, b = 0;
  for (int r = 0; r < nroles; ++r) {
    if ((r & 1) == 0) a += in ? buf[(size_t)r * wv_pair_words<K>() + lk] : 0u;
    b += in ? buf[(size_t)r * wv_pair_words<K>() + K + lk] : 0u;
  }
  // ---- window table: entry 0 = one, entry 1 = base, entry e = entry e-1 times base ----
  const uint32_t ba = a, bb = b;
  entry_store(1, a, b);
  entry_store(0, in ? HCTX(one)[lk] : 0u, in ? HCTX(one)[K + lk] : 0u);
#pragma unroll 1
  for (int e = 2; e < tsize; ++e) {
    wv_pairmul<K, LB, WIDEQ>(a, b, ba, bb, c);
    entry_store(e, a, b);
  }
  // ---- main loop: w squarings, one multiplication by a table entry (always, also entry 0 = one) ----
  entry_load(a, b, nwin > 0 ? digit(nwin - 1) : 0);
#pragma unroll 1
  for (int win = nwin - 2; win >= 0; --win) {
    const int d = digit(win);
#pragma unroll 1
    for (int i = 0; i < w; ++i) wv_pairsqr<K, LB, WIDEQ>(a, b, c);
    uint32_t ma, mb;
    entry_load(ma, mb, d);
    wv_pairmul<K, LB, WIDEQ>(a, b, ma, mb, c);
  }
  if (in) {
    buf[lk] = a;
    buf[K + lk] = b;
  }
#undef HCTX
}

// The entry of hensel_decrypt_ps_kernel (ps_entry_from_pair_row: per chunk of the pair row a single product for its b half and
// a pair product for its a half, all summed) as a kernel of its own, one lane per exponentiation and ROLE: the 2 * pchunks
// products of an entry are independent, so each runs in a lane of its own -- of a wavefront of its own: the role is
// wave-uniform (wavefront w: role w mod nroles, side (w / nroles) & 1, ciphertexts 64 * (w / (2 nroles)) ...) -- and leaves its
// partial pair in A.table + ((2*elem + side) * nroles + role) * 2K; the wave kernel adds them up.  18 us instead of 72 for the
// whole entry in one lane (a launch this small leaves the chip idle anyway).
template <int K, int LB>
__global__ __launch_bounds__(kWGThreads, 1) void hensel_ps_entry_kernel(HenselArgs A) {
  constexpr int K4 = (K + 3) / 4, RB = kLimbBits;
  constexpr int NI = (K * LB - 2) / RB + 1;      // row limbs per entry chunk that fit a half, plus one for the carry
  __shared__ uint4 park_[kWavesPerWG][K4][kWave];
  const int lane = threadIdx.x % kWave, wv = threadIdx.x / kWave;
  uint4* slot = &park_[wv][0][lane];
  const size_t wave_id = (size_t)blockIdx.x * kWavesPerWG + wv;
  const int nroles = 2 * A.pchunks;
  const int role = __builtin_amdgcn_readfirstlane((int)(wave_id % (size_t)nroles));
  const size_t grp = wave_id / (size_t)nroles;
  const int side = __builtin_amdgcn_readfirstlane((int)(grp & 1));
  const size_t first_elem = (grp >> 1) * kWave;
  size_t elem = first_elem + lane;
  const bool live = elem < A.count;
  if (first_elem >= A.count) return;             // (wave-uniform)
  if (!live) elem = A.count - 1;
#define HCTX(field) (side ? A.ctx[1].field : A.ctx[0].field)
  uint32_t n[K], a[K], b[K];
#pragma unroll
  for (int j = 0; j < K; ++j) n[j] = ps_uniform(HCTX(nhat)[j]);
  const uint32_t n1p = n[1] + 1;
  const int i = role >> 1, first = i * A.pchunk_limbs;
  const uint32_t* row = A.ct_pair + elem * A.ct_pair_stride + ((role & 1) ? A.pair_l2 : 0);
  uint32_t z[NI], zl[K];
#pragma unroll
  for (int j = 0; j < NI; ++j) z[j] = (j < A.pchunk_limbs && first + j < A.pair_l2) ? row[first + j] : 0u;
  // (rows written by the multi-lane kernels hold RELAXED limbs; ps_relimb makes them canonical in this kernel's width)
  ps_relimb<K, LB, NI, RB>(zl, z);
  if (role & 1) {                                // the b half of the chunk: one product, a contribution to b only
    uint32_t cb[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
      cb[j] = HCTX(pcb)[(size_t)i * K + j];
      a[j] = 0;
    }
    ps_mul<K, LB, true>(b, zl, cb, n, n1p, 0);
  } else {                                       // the a half: (z, 0) (x) pconv_i
    uint32_t ma[K], mb[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
      a[j] = zl[j];
      b[j] = 0;
      ma[j] = HCTX(pconv)[(size_t)i * 2 * K + j];
      mb[j] = HCTX(pconv)[(size_t)i * 2 * K + K + j];
    }
    ps_pairmul<K, LB, true>(a, b, ma, mb, n, n1p, 0, slot);
  }
#undef HCTX
  if (live) {
    uint32_t* buf = A.table + ((2 * elem + side) * (size_t)nroles + role) * wv_pair_words<K>();
#pragma unroll
    for (int j = 0; j < K; ++j) {
      buf[j] = a[j];
      buf[K + j] = b[j];
    }
  }
}

// ... and its exit: the pair the wave kernel left (relaxed limbs) made canonical, then ps_exit_words -> mp / mq words
template <int K, int LB>
__global__ __launch_bounds__(kWGThreads, 1) void hensel_ps_exit_kernel(HenselArgs A) {
  constexpr int K4 = (K + 3) / 4;
  __shared__ uint4 park_[kWavesPerWG][K4][kWave];
  const int lane = threadIdx.x % kWave, wv = threadIdx.x / kWave;
  uint4* slot = &park_[wv][0][lane];
  const size_t wave_id = (size_t)blockIdx.x * kWavesPerWG + wv;
  const int side = __builtin_amdgcn_readfirstlane((int)(wave_id & 1));
  const size_t first_elem = (wave_id >> 1) * kWave;
  size_t elem = first_elem + lane;
  const bool live = elem < A.count;
  if (!live) elem = A.count - 1;
  uint32_t a[K], b[K], ma[K], mb[K];
  const uint32_t* buf = A.table + (2 * elem + side) * (size_t)(2 * A.pchunks) * wv_pair_words<K>();   // (slot 0 of the roles)
#pragma unroll
  for (int j = 0; j < K; ++j) {
    ma[j] = buf[j];
    mb[j] = buf[K + j];
  }
  ps_relimb<K, LB, K, LB>(a, ma);
  ps_relimb<K, LB, K, LB>(b, mb);
  ps_exit_words<K, LB>(A, side, elem, live, slot, a, b, ma, mb);
}

}  // namespace pgpu

#endif  // PAILLIERCRYPTOLIB_AMD_CSRC_HENSEL_WAVE_HPP_

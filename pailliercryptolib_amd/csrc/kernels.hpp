// pailliercryptolib_amd -- gfx950 kernels of the batched-modexp hot path.
//
//   modexp_kernel : out[i] = F_i * base[i]^exp[i] mod N_i
//        plain form  -> ipcl::modExp (mod_exp.cpp:680-737, replaces mbx_exp_mb8 :508-516)
//        encrypt form-> PublicKey::raw_encrypt + applyObfuscator (pub_key.cpp:82-110):
//                       F_i = (1 + n*m_i) mod n^2 is computed in-kernel and becomes the
//                       multiplication that leaves the Montgomery domain (zero extra cost)
//        decrypt form-> first half of PrivateKey::decryptCRT (pri_key.cpp:119-134): two
//                       contexts (p^2, q^2; a wavefront serves one), the 2k-bit ciphertext is reduced on load
//                       (c mod p^2 / q^2, pri_key.cpp:128-129), F = hp / hq
//   crt_kernel    : second half of decryptCRT (pri_key.cpp:136-157): L-function by exact
//                   division, *hp mod p, CRT recombination
//   modmul_kernel : out[i] = a[i]*b[i] mod N     (CipherText::raw_add, ciphertext.cpp:135-141)
//
// One wavefront handles 64/G exponentiations (a workgroup is 4 independent wavefronts, one per SIMD);
// nothing is shared between wavefronts, so the only synchronisation is the single-wave LDS hand-off.
#ifndef PAILLIERCRYPTOLIB_AMD_CSRC_KERNELS_HPP_
#define PAILLIERCRYPTOLIB_AMD_CSRC_KERNELS_HPP_

#include "kargs.hpp"
#include "mont_core.hpp"

namespace pgpu {

// Workgroups are kWavesPerWG independent wavefronts (one per SIMD of a CU, for even SIMD load);
// each wavefront owns its slice of the LDS arrays and never talks to the others, so the LDS
// hand-off is wave-scope: LDS executes one wave's instructions in order, only the compiler must
// not reorder across the hand-off (no s_barrier).
// A wavefront that is alone on its SIMD (the bench's batch: 1024 wavefronts on 1024 SIMDs) issues a
// v_mad_u64_u32 every 5.2 cycles at the default priority and every 4.65 at priority 3
// (profiles/r02_ubench_lone_wave.txt; the CU's issue arbiter serves a raised wave sooner); with several
// waves per SIMD all of them are raised alike and nothing changes.  Measured on the decrypt launch: -2.5 %.
#ifndef PGPU_SETPRIO
#define PGPU_SETPRIO 3
#endif
__device__ __forceinline__ void raise_wave_priority() {
  if (PGPU_SETPRIO > 0) __builtin_amdgcn_s_setprio(PGPU_SETPRIO);
}

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// limbs (any lazy form) -> canonical 64-bit words in io[g][0..W64], zero padded
template <class GEO>
__device__ __forceinline__ void limbs_to_words(uint32_t (&r)[GEO::K], uint32_t (*bl)[GEO::L],
                                               uint64_t (*io)[GEO::W64 + 1], int lane, int g, int x) {
  constexpr int K = GEO::K, L = GEO::L, W64 = GEO::W64;
  full_normalise<GEO>(r, x);
  wave_lds_sync();
#pragma unroll
  for (int j = 0; j < K; ++j) bl[g][x * K + j] = r[j];
  wave_lds_sync();
  for (int t = lane; t < GEO::IPW * (W64 + 1); t += kWave) {
    int gg = t / (W64 + 1), w = t % (W64 + 1);
    io[gg][w] = (w < W64) ? word_from_limbs(bl[gg], L, w) : 0;
  }
  wave_lds_sync();
}

// one lane: while (v >= mod) v -= mod, at most `rounds` times.  tmp: W scratch words.
__device__ __forceinline__ void words_reduce(uint64_t* v, const uint64_t* mod, uint64_t* tmp, int W,
                                             int rounds) {
#pragma unroll 1
  for (int round = 0; round < rounds; ++round) {
    uint64_t borrow = 0;
#pragma unroll 1
    for (int w = 0; w < W; ++w) {
      uint64_t vw = v[w], nw = mod[w];
      uint64_t d = vw - nw - borrow;
      borrow = ((vw < nw) | ((vw == nw) & (borrow != 0))) ? 1 : 0;
      tmp[w] = d;
    }
    if (borrow) break;
#pragma unroll 1
    for (int w = 0; w < W; ++w) v[w] = tmp[w];
  }
}

// one lane: v = (a - b) mod m for canonical a, b in [0, m)
__device__ __forceinline__ void words_submod(uint64_t* v, const uint64_t* a, const uint64_t* b,
                                             const uint64_t* m, int W) {
  uint64_t borrow = 0;
#pragma unroll 1
  for (int w = 0; w < W; ++w) {
    uint64_t aw = a[w], bw = b[w];
    uint64_t d = aw - bw - borrow;
    borrow = ((aw < bw) | ((aw == bw) & (borrow != 0))) ? 1 : 0;
    v[w] = d;
  }
  if (borrow) {
    uint64_t carry = 0;
#pragma unroll 1
    for (int w = 0; w < W; ++w) {
      uint64_t s = v[w] + m[w];
      uint64_t c1 = s < v[w];
      uint64_t s2 = s + carry;
      carry = c1 | (s2 < s);
      v[w] = s2;
    }
  }
}

// Load one element per group (64-bit words, C-ABI layout) into LDS, zero padded to W64+1.
// Instance i reads source row (i / div); words [first, first+words) of that row.
// Group g of the wave works on instance first_inst + g * step (step 2: one parity of a two-context
// launch); out-of-range groups recompute the last instance of the same parity.
template <class GEO>
__device__ __forceinline__ void stage_words(uint64_t (*io)[GEO::W64 + 1], const uint64_t* src,
                                            size_t stride, int first, int words, size_t first_inst,
                                            size_t count, int div, int lane, int step = 1) {
  constexpr int WW = GEO::W64 + 1;
  for (int t = lane; t < GEO::IPW * WW; t += kWave) {
    int g = t / WW, w = t % WW;
    size_t inst = first_inst + (size_t)g * step;
    if (inst >= count) inst = count - step + first_inst % step;
    io[g][w] = (w < words) ? src[(inst / div) * stride + first + w] : 0;
  }
}

// Canonicalise r (value < 3N -> [0, N)) and store it as 64-bit words.
// nt: limbs of the TRUE modulus held by this lane.
template <class GEO>
__device__ __forceinline__ void store_canonical(uint32_t (&r)[GEO::K], const uint32_t (&nt)[GEO::K],
                                                int mod_words, uint32_t (*bl)[GEO::L],
                                                uint64_t (*io)[GEO::W64 + 1], uint64_t* out,
                                                size_t out_stride, size_t first_inst, size_t count,
                                                int lane, int g, int x, int step = 1) {
  full_normalise<GEO>(r, x);
  cond_sub_limbs<GEO>(r, nt, x, lane);
  limbs_to_words<GEO>(r, bl, io, lane, g, x);   // (already canonical: the normalise inside is a no-op pass)
  for (int t = lane; t < GEO::IPW * mod_words; t += kWave) {
    int gg = t / mod_words, w = t % mod_words;
    size_t inst = first_inst + (size_t)gg * step;
    if (inst < count) out[inst * out_stride + w] = io[gg][w];
  }
}

// a[j] += k[j], then relaxed normalisation (limbs <= 2^29).  Value must stay < R.
template <class GEO>
__device__ __forceinline__ void add_normalise(uint32_t (&a)[GEO::K], const uint32_t (&k)[GEO::K]) {
  constexpr int K = GEO::K;
  uint32_t c = 0;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    uint32_t u = a[j] + k[j] + c;
    a[j] = u & kLimbMask;
    c = u >> kLimbBits;
  }
  uint32_t cc = dpp_from_prev(c);
#pragma unroll
  for (int j = 0; j < K; ++j) {
    uint32_t u = a[j] + cc;
    a[j] = u & kLimbMask;
    cc = u >> kLimbBits;
  }
  a[0] += dpp_from_prev(cc);
}

#ifndef PGPU_MODEXP_MIN_WAVES
#define PGPU_MODEXP_MIN_WAVES 2
#endif
// REGROWS: the multiplier rows of every multiplication come from registers (DPP broadcasts, mont_core.hpp:
// mont_block_reg) instead of LDS -- the form for launches that leave a wavefront alone on its SIMD; the host
// picks it by wavefront count (capi.cpp: run_modexp).  Results are bit-identical.
template <class GEO, bool REGROWS = false>
__global__ __launch_bounds__(kWGThreads, REGROWS ? 1 : PGPU_MODEXP_MIN_WAVES) void modexp_kernel(ModexpArgs A) {
  constexpr int K = GEO::K, L = GEO::L, G = GEO::G, IPW = GEO::IPW;
  raise_wave_priority();
  __shared__ uint32_t bl_[kWavesPerWG][IPW][L];
  __shared__ uint64_t io_[kWavesPerWG][IPW][GEO::W64 + 1];

  const int lane = threadIdx.x % kWave, wv = threadIdx.x / kWave;
  auto& bl = bl_[wv];
  auto& io = io_[wv];
  const int g = lane / G, x = lane % G;
  const int nctx = A.nctx;
  const size_t wave_id = (size_t)blockIdx.x * kWavesPerWG + wv;
  const size_t tinst = wave_id * IPW + g;    // table slot (padded instances own a slot too)
  // instance of group g: first_inst + g * istep (two contexts: this wave takes one parity)
  const int istep = A.parity_waves ? 2 : 1;
  const size_t first_inst = A.parity_waves ? (wave_id >> 1) * IPW * 2 + (wave_id & 1) : wave_id * IPW;
  size_t inst = first_inst + (size_t)g * istep;
  if (inst >= A.count) inst = A.count - istep + first_inst % istep;   // padded lanes recompute the last element
  const uint64_t t_start = A.wave_clocks ? __builtin_readcyclecounter() : 0;
  const bool second = (nctx == 2) && (inst & 1);   // explicit selects: no dynamic arg indexing
  // Context fields are selected where they are used (all outside the multiplication loop) rather than
  // held in a per-lane copy: eight 64-bit pointers would otherwise stay live across every montmul.
#define PGPU_CTX(field) (second ? A.ctx[1].field : A.ctx[0].field)

  // loop modulus: Nhat (unit quotient digits) when the context provides it, else N
  const bool unitq = A.ctx[0].nhat != nullptr;     // wave-uniform (both contexts agree)
  uint32_t n[K], a[K];
  uint32_t mreg[REGROWS ? K : 1];   // REGROWS: the multiplier of the next multiplication (squarings use a itself)
  // the multiplier limb j of this lane goes to its LDS row or, with REGROWS, stays in a register
#define PGPU_STAGE(j, value)                                         \
  do {                                                               \
    if constexpr (REGROWS) mreg[j] = (value);                        \
    else bl[g][x * K + (j)] = (value);                               \
  } while (0)
  // one multiplication a = a * multiplier (SQ: the multiplier is a itself)
#define PGPU_MONTMUL(SQ, UQ)                                                              \
  do {                                                                                    \
    if constexpr (REGROWS) {                                                              \
      if constexpr (SQ) montmul_reg<GEO, true, UQ>(a, a, a, n, n0inv);                    \
      else montmul_reg<GEO, false, UQ>(a, a, mreg, n, n0inv);                             \
    } else {                                                                              \
      if constexpr (SQ) montmul<GEO, true, UQ>(a, a, bl[g], n, n0inv);                    \
      else montmul<GEO, false, UQ>(a, a, bl[g], n, n0inv);                                \
    }                                                                                     \
  } while (0)
  // (g^m = 1 + n*m is formed under the TRUE modulus so that it stays < 2N; see GMUL below)
  const bool gm_first = A.final_mul == FM_PAILLIER_G;
#pragma unroll
  for (int j = 0; j < K; ++j) n[j] = (unitq && !gm_first) ? PGPU_CTX(nhat)[x * K + j] : PGPU_CTX(n)[x * K + j];
  const uint32_t n0inv = PGPU_CTX(n0inv);
  const int mw = A.ctx[0].mod_words;                // both contexts share the row width
  const bool wide = A.base_words > mw;

  const int w = A.window;
  // A schedule implies parity waves, so the context index is wave-uniform: say so (readfirstlane), or
  // the step counters and with them the whole phase machine would be compiled as divergent control flow.
  const bool sched_mode = A.sched[0] != nullptr;
  const bool second_u = __builtin_amdgcn_readfirstlane((int)second) != 0;
  const uint16_t* sch = second_u ? A.sched[1] : A.sched[0];
  const int nsteps = second_u ? A.sched_len[1] : A.sched_len[0];
  const int tsize = sched_mode ? 1 << (w - 1) : 1 << w;     // odd powers only under a schedule
  // this lane's slice of entry 0; one spare entry behind the table parks the value that waits for a later
  // multiplication (hi part of a wide base, g^m) -- in HBM rather than in K registers for the whole loop
  uint32_t* tbl = A.table + tinst * (size_t)(tsize + 1) * L + x * K;
  uint32_t* keep = tbl + (size_t)tsize * L;
  const uint64_t* ep = A.exp + (A.exp_per_ctx ? (inst % nctx) : inst) * A.exp_stride;
  const int nwin = (A.exp_bits + w - 1) / w;

  auto digit = [&](int i) -> int {
    int bit = i * w;
    int word = bit >> 6, sh = bit & 63;
    uint64_t v = (word < A.exp_words) ? ep[word] >> sh : 0;
    if (sh + w > 64 && word + 1 < A.exp_words) v |= ep[word + 1] << (64 - sh);
    return (int)(v & (uint64_t)(tsize - 1));
  };

  // The whole exponentiation is one loop around a single montmul call site; the wave-uniform
  // phase variable selects how the multiplier operand is staged and what happens to the result.
  enum { GMUL, TOMONT_HI, TOMONT, X2, TABLE, SQR, MUL, FINAL };
  int phase;
  int e = 2;         // next table entry to build
  int win = 0;       // current window index (schedule: next step)
  int sq = 0;        // squarings left in the current window
  int mul_idx = 0;   // schedule: table entry of the pending multiplication, -1 = none

  // ---- first multiplication: operand a from global words, multiplier from the context ----
  if (A.final_mul == FM_PAILLIER_G) {
    phase = GMUL;    // keep = m * (n*R) * R^-1 = n*m mod n^2 (plain domain, lazy)
    stage_words<GEO>(io, A.fm_words, A.fm_stride, 0, A.fm_nwords, first_inst, A.count, 1, lane, istep);
#pragma unroll
    for (int j = 0; j < K; ++j) PGPU_STAGE(j, PGPU_CTX(nr)[x * K + j]);
  } else if (wide) {
    phase = TOMONT_HI;  // keep = hi(base) * 2^(64 mw) * R mod N
    stage_words<GEO>(io, A.base, A.base_stride, mw, A.base_words - mw, first_inst, A.count, nctx, lane, istep);
#pragma unroll
    for (int j = 0; j < K; ++j) PGPU_STAGE(j, PGPU_CTX(r2s)[x * K + j]);
  } else {
    phase = TOMONT;
    stage_words<GEO>(io, A.base, A.base_stride, 0, A.base_words, first_inst, A.count, nctx, lane, istep);
#pragma unroll
    for (int j = 0; j < K; ++j) PGPU_STAGE(j, PGPU_CTX(r2)[x * K + j]);
  }
  wave_lds_sync();
#pragma unroll
  for (int j = 0; j < K; ++j) a[j] = limb_from_words(io[g], x * K + j);

  for (;;) {
    if (phase == FINAL) {
      // leave the Montgomery domain modulo the TRUE modulus (result < N + 1)
      if (unitq) {
#pragma unroll
        for (int j = 0; j < K; ++j) n[j] = PGPU_CTX(n)[x * K + j];
      }
      PGPU_MONTMUL(false, false);
      break;
    }
    // A run of squarings (83 % of all multiplications) is a tight loop around ONE call site: nothing between two
    // squarings but the counter -- and, in the LDS form, the re-staging of the operand.  (Routing every squaring
    // through the phase machine below cost 50-100 instructions of register shuffling and scalar bookkeeping each,
    // and a scalar instruction is a 6-18 cycle issue slot for a wavefront that is alone on its SIMD.)
#define PGPU_SQUARING_RUN(UQ)                                                  \
  do {                                                                          \
    int run_ = (phase == X2) ? 1 : sq;                                          \
    for (;;) {                                                                  \
      PGPU_MONTMUL(true, UQ);                                                   \
      if (--run_ == 0) break;                                                   \
      if constexpr (!REGROWS) {                                                 \
        wave_lds_sync();                                                        \
        _Pragma("unroll") for (int j = 0; j < K; ++j) bl[g][x * K + j] = a[j];  \
        wave_lds_sync();                                                        \
      }                                                                         \
    }                                                                           \
    if (phase == SQR) sq = 0;                                                   \
  } while (0)
    if (unitq && phase != GMUL) {
      if (phase == SQR || phase == X2) PGPU_SQUARING_RUN(true);
      else PGPU_MONTMUL(false, true);
    } else {
      if (phase == SQR || phase == X2) PGPU_SQUARING_RUN(false);
      else PGPU_MONTMUL(false, false);
    }
#undef PGPU_SQUARING_RUN

    bool start_main = false;
    if (phase == GMUL || phase == TOMONT_HI) {
      // park the result, then run the (low-part) to-Montgomery multiplication of the base
      if (phase == GMUL && A.ctx[0].gadd) {
        // Montgomery-form output: the staged multiplier was n*R^2, so a = n*m*R; g^m*R = a + (R mod N)
        uint32_t gk[K];
#pragma unroll
        for (int j = 0; j < K; ++j) gk[j] = PGPU_CTX(gadd)[x * K + j];
        add_normalise<GEO>(a, gk);
      }
#pragma unroll
      for (int j = 0; j < K; ++j)
        keep[j] = a[j] + ((phase == GMUL && !A.ctx[0].gadd && x == 0 && j == 0) ? 1u : 0u);   // g^m = 1 + n*m
      if (phase == GMUL && unitq) {                        // the loop itself runs modulo Nhat
#pragma unroll
        for (int j = 0; j < K; ++j) n[j] = PGPU_CTX(nhat)[x * K + j];
      }
      wave_lds_sync();
      stage_words<GEO>(io, A.base, A.base_stride, 0, wide ? mw : A.base_words, first_inst, A.count,
                       nctx, lane, istep);
#pragma unroll
      for (int j = 0; j < K; ++j) PGPU_STAGE(j, PGPU_CTX(r2)[x * K + j]);
      wave_lds_sync();
#pragma unroll
      for (int j = 0; j < K; ++j) a[j] = limb_from_words(io[g], x * K + j);
      phase = TOMONT;
      continue;
    }
    // schedule mode: load step `win` (nsq squarings, then maybe a multiplication); FINAL after the last
    auto next_step = [&]() {
      if (win >= nsteps) { phase = FINAL; return; }
      const int st = __builtin_amdgcn_readfirstlane((int)sch[win++]);
      sq = st >> 6;
      mul_idx = (st & 63) - 1;
      phase = SQR;
    };
    if (phase == TOMONT) {
      if (wide) {                                           // (lo + hi*2^S) * R, lazy < 4N
        uint32_t hi[K];
#pragma unroll
        for (int j = 0; j < K; ++j) hi[j] = keep[j];
        add_normalise<GEO>(a, hi);
      }
      // a = base*R.  Fixed window: table[1] = a, table[0] = R mod N, multiplier of the table build = a.
      // Schedule: table[0] = a (odd powers base^(2i+1)), multiplier of the table build = a^2.
      wave_lds_sync();
      if (sched_mode) {
#pragma unroll
        for (int j = 0; j < K; ++j) tbl[j] = a[j];
        if (tsize > 1) phase = X2; else start_main = true;
      } else {
#pragma unroll
        for (int j = 0; j < K; ++j) {
          tbl[L + j] = a[j];
          tbl[j] = PGPU_CTX(one)[x * K + j];
          PGPU_STAGE(j, a[j]);
        }
        wave_lds_sync();
        if (tsize > 2) phase = TABLE; else start_main = true;
      }
    } else if (phase == X2) {
      // a = base^2 * R: it becomes the staged multiplier; the running value restarts from base
      wave_lds_sync();
#pragma unroll
      for (int j = 0; j < K; ++j) { PGPU_STAGE(j, a[j]); a[j] = tbl[j]; }
      wave_lds_sync();
      e = 1;
      phase = TABLE;
    } else if (phase == TABLE) {
#pragma unroll
      for (int j = 0; j < K; ++j) tbl[(size_t)e * L + j] = a[j];
      if (++e == tsize) start_main = true;
    } else if (phase == SQR) {   // the whole run of squarings is done (sq == 0)
      if (!sched_mode || mul_idx >= 0) phase = MUL; else next_step();
    } else {  // MUL
      if (sched_mode) next_step();
      else if (--win < 0) phase = FINAL; else { phase = SQR; sq = w; }
    }

    if (start_main) {
      if (sched_mode ? nsteps == 0 : nwin == 0) {
#pragma unroll
        for (int j = 0; j < K; ++j) a[j] = PGPU_CTX(one)[x * K + j];
        phase = FINAL;
      } else if (sched_mode) {
        const int d = (__builtin_amdgcn_readfirstlane((int)sch[0]) & 63) - 1;   // step 0: a = table[d] (this lane's own earlier stores)
#pragma unroll
        for (int j = 0; j < K; ++j) a[j] = tbl[(size_t)d * L + j];
        win = 1;
        next_step();
      } else {
        int d = digit(nwin - 1);   // top window: a = table[d] (this lane's own earlier stores)
#pragma unroll
        for (int j = 0; j < K; ++j) a[j] = tbl[(size_t)d * L + j];
        win = nwin - 2;
        if (win < 0) phase = FINAL; else { phase = SQR; sq = w; }
      }
    }

    // ---- stage the multiplier operand of the next multiplication ----
    if (phase == SQR || phase == X2) {
      if constexpr (!REGROWS) {     // (with REGROWS a squaring reads its rows from a: nothing to stage)
        wave_lds_sync();
#pragma unroll
        for (int j = 0; j < K; ++j) bl[g][x * K + j] = a[j];
        wave_lds_sync();
      }
    } else if (phase == MUL) {
      int d = sched_mode ? mul_idx : digit(win);
      uint32_t t[K];
#pragma unroll
      for (int j = 0; j < K; ++j) t[j] = tbl[(size_t)d * L + j];
      wave_lds_sync();
#pragma unroll
      for (int j = 0; j < K; ++j) PGPU_STAGE(j, t[j]);
      wave_lds_sync();
    } else if (phase == FINAL) {
      // leave the Montgomery domain: multiply by 1, by the context constant, or by g^m
      wave_lds_sync();
      if (A.final_mul == FM_UNIT) {
#pragma unroll
        for (int j = 0; j < K; ++j) PGPU_STAGE(j, (x == 0 && j == 0) ? 1u : 0u);
      } else if (A.final_mul == FM_CTX_CONST) {
#pragma unroll
        for (int j = 0; j < K; ++j) PGPU_STAGE(j, PGPU_CTX(fc)[x * K + j]);
      } else {
#pragma unroll
        for (int j = 0; j < K; ++j) PGPU_STAGE(j, keep[j]);
      }
      wave_lds_sync();
    }
    // phase == TABLE: multiplier (base*R) is already staged
  }
  // canonical reduction modulo the TRUE modulus of this group's context (n[] holds it after FINAL)
  store_canonical<GEO>(a, n, mw, bl, io, A.out, A.out_stride, first_inst, A.count, lane, g, x, istep);
  if (A.wave_clocks && lane == 0) {
    uint64_t* rec = A.wave_clocks + ((size_t)blockIdx.x * kWavesPerWG + wv) * 3;
    rec[0] = t_start;
    rec[1] = __builtin_readcyclecounter();
    uint64_t xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));   // HW_REG_XCC_ID
    uint64_t hwid = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_REG_HW_ID
    rec[2] = xcc | (hwid << 8);
  }
}

#undef PGPU_STAGE
#undef PGPU_MONTMUL
#undef PGPU_CTX

// ---------------------------------------------------------------------------------------------
// Fixed-base exponentiation for the DJN obfuscator hs^r (pub_key.cpp:51-64): the base hs is a key
// constant, so   hs^r = prod_i T[i][d_i],   T[i][d] = hs^(d * 2^(w*i)) * R mod N,   d_i = i-th
// w-bit digit of r   -- nwin-1 multiplications and NO squarings (1024-bit r, w = 8: 127 instead
// of 1259).  The table (nwin * 2^w entries of L limbs) is built once per key by fb_build_kernel and
// lives in HBM / Infinity Cache; an entry is loaded straight into the lane-resident operand.
// instance i builds row i of the table: B = hs^(2^(w*i)) by w*i squarings, then T[i][d] = T[i][d-1]*B
template <class GEO>
__global__ __launch_bounds__(kWGThreads) void fb_build_kernel(FixedBaseBuildArgs A) {
  constexpr int K = GEO::K, L = GEO::L, G = GEO::G, IPW = GEO::IPW;
  __shared__ uint32_t bl_[kWavesPerWG][IPW][L];
  __shared__ uint64_t io_[kWavesPerWG][IPW][GEO::W64 + 1];
  const int lane = threadIdx.x % kWave, wv = threadIdx.x / kWave;
  auto& bl = bl_[wv];
  auto& io = io_[wv];
  const int g = lane / G, x = lane % G;
  const size_t first_inst = ((size_t)blockIdx.x * kWavesPerWG + wv) * IPW;
  size_t inst = first_inst + g;
  const bool live = inst < (size_t)A.nwin;
  if (!live) inst = (size_t)A.nwin - 1;
  const bool unitq = A.ctx.nhat != nullptr;
  uint32_t n[K], a[K];
#pragma unroll
  for (int j = 0; j < K; ++j) n[j] = unitq ? A.ctx.nhat[x * K + j] : A.ctx.n[x * K + j];
  const uint32_t n0inv = A.ctx.n0inv;
  stage_words<GEO>(io, A.base, 0, 0, A.ctx.mod_words, 0, 1, 1, lane);
#pragma unroll
  for (int j = 0; j < K; ++j) bl[g][x * K + j] = A.ctx.r2[x * K + j];
  wave_lds_sync();
#pragma unroll
  for (int j = 0; j < K; ++j) a[j] = limb_from_words(io[g], x * K + j);
  const int tsize = 1 << A.w;
  // every group runs the squaring count of the LAST live instance of its wave so that control
  // flow stays wave-uniform; a group stops updating once its own count is reached
  const int my_sq = A.w * (int)inst;
  size_t last = first_inst + IPW - 1;
  if (last >= (size_t)A.nwin) last = (size_t)A.nwin - 1;
  const int wave_sq = A.w * (int)last;
  uint32_t* row = A.table + inst * (size_t)tsize * L + x * K;
  // step 0: to Montgomery form; steps 1..wave_sq: squarings; then tsize-2 table multiplications
  const int total = 1 + wave_sq + (tsize - 2);
#pragma unroll 1
  for (int step = 0; step < total; ++step) {
    uint32_t r[K];
    if (unitq) montmul<GEO, false, true>(r, a, bl[g], n, n0inv);
    else montmul<GEO, false, false>(r, a, bl[g], n, n0inv);
    const bool squaring_phase = step <= wave_sq;   // result of step is hs*R (0) or a square
    if (squaring_phase) {
      if (step <= my_sq) {
#pragma unroll
        for (int j = 0; j < K; ++j) a[j] = r[j];
      }
      if (step == wave_sq) {
        // a = B_i * R: table entries 0 and 1, multiplier for the rest of the row
        if (live) {
#pragma unroll
          for (int j = 0; j < K; ++j) { row[j] = A.ctx.one[x * K + j]; row[L + j] = a[j]; }
        }
      }
      wave_lds_sync();
#pragma unroll
      for (int j = 0; j < K; ++j) bl[g][x * K + j] = a[j];
      wave_lds_sync();
    } else {
      const int d = step - wave_sq + 1;   // entry index 2..tsize-1
#pragma unroll
      for (int j = 0; j < K; ++j) a[j] = r[j];
      if (live) {
#pragma unroll
        for (int j = 0; j < K; ++j) row[(size_t)d * L + j] = a[j];
      }
    }
  }
}

// Window-table entry `idx` of this lane's slice (entries are LQ limbs apart).  gather: read EVERY entry and keep the one
// wanted -- the address stream is then the same for every exponent, as in the reference's mbx_exp_mb8, which gathers its
// table in constant time; costs tsize*K loads and selects per multiplication instead of K loads.
template <int K>
__device__ __forceinline__ void load_table_entry(uint32_t (&dst)[K], const uint32_t* tbl, int idx, int tsize, size_t stride,
                                                 bool gather) {
  if (!gather) {
#pragma unroll
    for (int j = 0; j < K; ++j) dst[j] = tbl[(size_t)idx * stride + j];
    return;
  }
#pragma unroll
  for (int j = 0; j < K; ++j) dst[j] = 0;
  // two entries per trip: their loads are in flight together (a trip is latency-bound otherwise: 2^w round trips to
  // L2 per multiplication); tsize is a power of two >= 2 whenever a table is used
#pragma unroll 1
  for (int e = 0; e < tsize; e += 2) {
    uint32_t t0[K], t1[K];
    const int e1 = e + 1 < tsize ? e + 1 : e;
#pragma unroll
    for (int j = 0; j < K; ++j) {
      t0[j] = tbl[(size_t)e * stride + j];
      t1[j] = tbl[(size_t)e1 * stride + j];
    }
    // (one v_cndmask_b32 per limb and candidate -- round 4 masked and or-ed: three to four instructions per limb and pair)
    const bool s0 = e == idx, s1 = e1 == idx;
#pragma unroll
    for (int j = 0; j < K; ++j) dst[j] = s0 ? t0[j] : (s1 ? t1[j] : dst[j]);
  }
}

template <class GEO>
__global__ __launch_bounds__(kWGThreads, 2) void fb_encrypt_kernel(FixedBaseArgs A) {
  constexpr int K = GEO::K, L = GEO::L, G = GEO::G, IPW = GEO::IPW;
  raise_wave_priority();
  __shared__ uint32_t bl_[kWavesPerWG][IPW][L];
  __shared__ uint64_t io_[kWavesPerWG][IPW][GEO::W64 + 1];
  const int lane = threadIdx.x % kWave, wv = threadIdx.x / kWave;
  auto& bl = bl_[wv];
  auto& io = io_[wv];
  const int g = lane / G, x = lane % G;
  const size_t first_inst = ((size_t)blockIdx.x * kWavesPerWG + wv) * IPW;
  size_t inst = first_inst + g;
  if (inst >= A.count) inst = A.count - 1;
  const bool unitq = A.ctx.nhat != nullptr;
  uint32_t n[K], a[K], acc[K], nxt[K];
#pragma unroll
  for (int j = 0; j < K; ++j) n[j] = A.ctx.n[x * K + j];   // step 0 (g^m) runs under the TRUE modulus
  const uint32_t n0inv = A.ctx.n0inv;
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
    load_table_entry<K>(dst, A.table + (size_t)i * tsize * L + x * K, digit(i), tsize, L, A.ct_gather != 0);
  };
  // step 0: acc = m * (n*R) * R^-1 + 1 = g^m (plain domain, lazy); multiplier staged = n*R
  stage_words<GEO>(io, A.fm_words, A.fm_stride, 0, A.fm_nwords, first_inst, A.count, 1, lane);
#pragma unroll
  for (int j = 0; j < K; ++j) bl[g][x * K + j] = A.ctx.nr[x * K + j];
  wave_lds_sync();
#pragma unroll
  for (int j = 0; j < K; ++j) a[j] = limb_from_words(io[g], x * K + j);
  load_entry(acc, 0);   // T[0][d_0]: the running product starts here
  uint32_t gm[K];
  const int nwin = A.nwin;
  // steps: 0 (g^m), 1..nwin-1 (acc *= T[i][d_i]), nwin (acc *= g^m, leaves Montgomery form).
  // The table entry of step+1 is fetched before the multiplication of step (latency hidden).
#pragma unroll 1
  for (int step = 0; step <= nwin; ++step) {
    if (step + 1 < nwin) load_entry(nxt, step + 1);
    uint32_t r[K];
    if (step == 0 || step == nwin) {
      // g^m (so that it stays < 2N) and the multiplication that leaves the Montgomery domain run
      // modulo the TRUE modulus; the table products in between modulo Nhat (unit quotient digits)
      if (unitq && step == nwin) {
#pragma unroll
        for (int j = 0; j < K; ++j) n[j] = A.ctx.n[x * K + j];
      }
      montmul<GEO, false, false>(r, a, bl[g], n, n0inv);
      if (unitq && step == 0) {
#pragma unroll
        for (int j = 0; j < K; ++j) n[j] = A.ctx.nhat[x * K + j];
      }
    } else if (unitq) {
      montmul<GEO, false, true>(r, a, bl[g], n, n0inv);
    } else {
      montmul<GEO, false, false>(r, a, bl[g], n, n0inv);
    }
    if (step == nwin) {
#pragma unroll
      for (int j = 0; j < K; ++j) a[j] = r[j];
      break;
    }
    if (step == 0) {
#pragma unroll
      for (int j = 0; j < K; ++j) gm[j] = r[j];
      if (A.ctx.gadd) {       // Montgomery-form output (see modexp_kernel GMUL): g^m*R = n*m*R + (R mod N)
        uint32_t gk[K];
#pragma unroll
        for (int j = 0; j < K; ++j) gk[j] = A.ctx.gadd[x * K + j];
        add_normalise<GEO>(gm, gk);
      } else if (x == 0) {
        gm[0] += 1;
      }
    } else {
#pragma unroll
      for (int j = 0; j < K; ++j) acc[j] = r[j];
    }
    // next multiplication: lane-resident operand = table entry (g^m at the end), row operand = acc
    wave_lds_sync();
#pragma unroll
    for (int j = 0; j < K; ++j) bl[g][x * K + j] = acc[j];
    wave_lds_sync();
#pragma unroll
    for (int j = 0; j < K; ++j) a[j] = (step + 1 < nwin) ? nxt[j] : gm[j];
  }
  store_canonical<GEO>(a, n, A.ctx.mod_words, bl, io, A.out, A.out_stride, first_inst, A.count, lane, g, x);
}

// Batched modular product (CipherText::raw_add, ciphertext.cpp:135-141) in four flavours (ModmulMode):
//   MM_PLAIN   out = a*b mod N          montmul(montmul(a, R^2), b): plain operands, plain result
//   MM_SINGLE  out = montmul(a, b)      operands in Montgomery form -> result in Montgomery form (ONE product;
//                                       device-resident ciphertext chains stay in the Montgomery domain)
//   MM_BY_R2   out = montmul(a, R^2)    plain -> Montgomery form
//   MM_BY_ONE  out = montmul(a, 1)      Montgomery form -> plain
//   MM_GM      out = a * (1 + n*b)      CT + PT with the plaintext b (ctx.nr = n*R^2, ctx.gadd = R mod N)
// Results are canonical (< N) in every mode.
template <class GEO>
__global__ __launch_bounds__(kWGThreads) void modmul_kernel(ModmulArgs A) {
  constexpr int K = GEO::K, L = GEO::L, G = GEO::G, IPW = GEO::IPW;
  raise_wave_priority();
  __shared__ uint32_t bl_[kWavesPerWG][IPW][L];
  __shared__ uint64_t io_[kWavesPerWG][IPW][GEO::W64 + 1];
  const int lane = threadIdx.x % kWave, wv = threadIdx.x / kWave;
  auto& bl = bl_[wv];
  auto& io = io_[wv];
  const int g = lane / G, x = lane % G;
  const size_t first_inst = ((size_t)blockIdx.x * kWavesPerWG + wv) * IPW;

  uint32_t n[K], a[K];
#pragma unroll
  for (int j = 0; j < K; ++j) n[j] = A.ctx.n[x * K + j];
  const uint32_t n0inv = A.ctx.n0inv;
  const int mode = A.mode;

  if (mode == MM_GM) {
    // CT + PT (ciphertext.cpp:75-80) without the host loop of raw_encrypt: g^m*R = m*(n*R^2)*R^-1 + (R mod N),
    // then ONE product with the ciphertext rows -- the result has the form (Montgomery / plain) of the ciphertext
    stage_words<GEO>(io, A.b, A.b_stride, 0, A.b_words, first_inst, A.count, 1, lane);
#pragma unroll
    for (int j = 0; j < K; ++j) bl[g][x * K + j] = A.ctx.nr[x * K + j];
    wave_lds_sync();
#pragma unroll
    for (int j = 0; j < K; ++j) a[j] = limb_from_words(io[g], x * K + j);
    montmul<GEO>(a, a, bl[g], n, n0inv);
    uint32_t gk[K];
#pragma unroll
    for (int j = 0; j < K; ++j) gk[j] = A.ctx.gadd[x * K + j];
    add_normalise<GEO>(a, gk);
    wave_lds_sync();
    stage_words<GEO>(io, A.a, A.a_stride, 0, A.in_words, first_inst, A.count, 1, lane);
    wave_lds_sync();
#pragma unroll
    for (int j = 0; j < K; ++j) bl[g][x * K + j] = limb_from_words(io[g], x * K + j);
    wave_lds_sync();
    montmul<GEO>(a, a, bl[g], n, n0inv);
  } else {
    stage_words<GEO>(io, A.a, A.a_stride, 0, A.in_words, first_inst, A.count, 1, lane);
    if (mode == MM_PLAIN || mode == MM_BY_R2) {
#pragma unroll
      for (int j = 0; j < K; ++j) bl[g][x * K + j] = A.ctx.r2[x * K + j];
    } else if (mode == MM_BY_ONE) {
#pragma unroll
      for (int j = 0; j < K; ++j) bl[g][x * K + j] = (x == 0 && j == 0) ? 1u : 0u;
    }
    wave_lds_sync();
#pragma unroll
    for (int j = 0; j < K; ++j) a[j] = limb_from_words(io[g], x * K + j);
    const int nsteps = mode == MM_PLAIN ? 2 : 1;
#pragma unroll 1
    for (int step = 0; step < nsteps; ++step) {
      if (mode == MM_SINGLE || step == 1) {     // the multiplier is the b operand
        wave_lds_sync();
        stage_words<GEO>(io, A.b, A.b_stride, 0, A.in_words, first_inst, A.count, 1, lane);
        wave_lds_sync();
#pragma unroll
        for (int j = 0; j < K; ++j) bl[g][x * K + j] = limb_from_words(io[g], x * K + j);
        wave_lds_sync();
      }
      montmul<GEO>(a, a, bl[g], n, n0inv);      // a*R (any a < R) / a*b*R^-1 mod N, lazy
    }
  }
  store_canonical<GEO>(a, n, A.ctx.mod_words, bl, io, A.out, (size_t)A.ctx.mod_words, first_inst, A.count, lane, g, x);
}

// CRT recombination, one G-lane group per ciphertext (pri_key.cpp:136-157):
//   Zp = (Vp - hp) mod p^2 = p * mp           (Vp = xp*hp mod p^2, xp = c^(p-1) mod p^2;
//                                              L(xp)*hp mod p == mp  <=>  (xp-1)*hp mod p^2 == p*mp)
//   mp = Zp * p^-1 mod M  (exact division, M coprime to p, M > p)         likewise mq
//   u  = (mq - mp) * (p^-1 mod q) mod q,  m = mp + u*p  (u*p as an exact product mod M > n)
template <class GEO>
__global__ __launch_bounds__(kWGThreads) void crt_kernel(CrtArgs A) {
  constexpr int K = GEO::K, L = GEO::L, G = GEO::G, IPW = GEO::IPW, W64 = GEO::W64;
  __shared__ uint32_t bl_[kWavesPerWG][IPW][L];
  __shared__ uint64_t io_[kWavesPerWG][IPW][W64 + 1];    // working value
  __shared__ uint64_t ymp_[kWavesPerWG][IPW][W64 + 1];   // mp
  __shared__ uint64_t tmp_[kWavesPerWG][IPW][W64 + 1];   // scratch / mq
  const int lane = threadIdx.x % kWave, wv = threadIdx.x / kWave;
  auto& bl = bl_[wv];
  auto& io = io_[wv];
  auto& ymp = ymp_[wv];
  auto& tmp = tmp_[wv];
  const int g = lane / G, x = lane % G;
  const size_t first_inst = ((size_t)blockIdx.x * kWavesPerWG + wv) * IPW;
  size_t inst = first_inst + g;
  if (inst >= A.count) inst = A.count - 1;
  const int vw = A.vw;

  uint32_t n[K], a[K];
  // steps: 0: mp = (Vp-hp)/p   1: mq = (Vq-hq)/q   2: u = (mq-mp)*pinv mod q   3: t = u*p
#pragma unroll 1
  for (int step = 0; step < 4; ++step) {
    const ModCtxDev& C = (step == 2) ? A.ctxQ : A.ctxM;
    const uint32_t* mulc = (step == 0) ? A.cp : (step == 1) ? A.cq : (step == 2) ? A.pinvR : A.pRM;
    wave_lds_sync();
    if (A.have_m && step < 2) {   // the exponentiation kernel delivered mp / mq themselves (hensel.hpp)
      const uint64_t* V = A.v + (2 * inst + step) * (size_t)vw;
      uint64_t* dst = step ? tmp[g] : ymp[g];
      for (int t = x; t <= W64; t += G) {
        dst[t] = (t < vw) ? V[t] : 0;
        io[g][t] = 0;   // (step 2 writes vw words of the difference and reads limbs from all of them)
      }
      continue;
    }
    if (step < 2) {
      // io = (V - h) mod sq, exactly divisible by the prime
      const uint64_t* V = A.v + (2 * inst + step) * (size_t)vw;
      for (int t = x; t <= W64; t += G) io[g][t] = (t < vw) ? V[t] : 0;
      wave_lds_sync();
      if (x == 0) {
        const uint64_t* h = step ? A.hq64 : A.hp64;
        const uint64_t* sq = step ? A.q2_64 : A.p2_64;
        words_submod(io[g], io[g], h, sq, vw);
      }
    } else if (step == 2) {
      // io = (mq - mp) mod q    (mq sits in tmp, mp in ymp; both canonical, mp < p < q)
      if (x == 0) words_submod(io[g], tmp[g], ymp[g], A.q64, vw);
    }
    // step 3: io already holds u
    wave_lds_sync();
#pragma unroll
    for (int j = 0; j < K; ++j) {
      a[j] = limb_from_words(io[g], x * K + j);
      n[j] = C.n[x * K + j];
      bl[g][x * K + j] = mulc[x * K + j];
    }
    wave_lds_sync();
    montmul<GEO>(a, a, bl[g], n, C.n0inv);
    limbs_to_words<GEO>(a, bl, io, lane, g, x);
    if (x == 0) {
      words_reduce(io[g], C.n64, reinterpret_cast<uint64_t*>(bl[g]), W64, 2);
      if (step == 0) { for (int t = 0; t <= W64; ++t) ymp[g][t] = io[g][t]; }
      if (step == 1) { for (int t = 0; t <= W64; ++t) tmp[g][t] = io[g][t]; }
      if (step == 3) {
        // m = mp + u*p  (< n, no reduction needed)
        uint64_t carry = 0;
#pragma unroll 1
        for (int t = 0; t < W64; ++t) {
          uint64_t s = io[g][t] + ymp[g][t];
          uint64_t c1 = s < io[g][t];
          uint64_t s2 = s + carry;
          carry = c1 | (s2 < s);
          io[g][t] = s2;
        }
      }
    }
  }
  wave_lds_sync();
  const int ow = A.out_words;
  for (int t = lane; t < IPW * ow; t += kWave) {
    int gg = t / ow, ww = t % ow;
    size_t oi = first_inst + gg;
    if (oi < A.count) A.out[oi * (size_t)ow + ww] = io[gg][ww];
  }
}

}  // namespace pgpu

#endif  // PAILLIERCRYPTOLIB_AMD_CSRC_KERNELS_HPP_

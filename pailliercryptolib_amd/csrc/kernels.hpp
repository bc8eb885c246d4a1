// pailliercryptolib_amd -- gfx950 kernels of the batched-modexp hot path.
//
//   modexp_kernel : out[i] = base[i]^exp[i] mod N          (ipcl::modExp, mod_exp.cpp:680-737)
//   modmul_kernel : out[i] = a[i]*b[i] mod N               (CipherText::raw_add, ciphertext.cpp:135-141)
//
// One wavefront (= one 64-thread workgroup) handles 64/G exponentiations; nothing is shared
// between wavefronts, so there are no workgroup barriers beyond the single-wave LDS hand-offs.
#ifndef PAILLIERCRYPTOLIB_AMD_CSRC_KERNELS_HPP_
#define PAILLIERCRYPTOLIB_AMD_CSRC_KERNELS_HPP_

#include "mont_core.hpp"

namespace pgpu {

// Montgomery context of one odd modulus N, resident in device memory (built by the host,
// see capi.hip: build_modctx).  R = 2^(29*L) for the geometry the context was built for.
struct ModCtxDev {
  const uint32_t* n;    // [L]   N, 29-bit limbs
  const uint32_t* r2;   // [L]   R^2 mod N
  const uint32_t* one;  // [L]   R mod N
  const uint64_t* n64;  // [W64+1] N as little-endian 64-bit words, zero padded
  uint32_t n0inv;       // -N^-1 mod 2^29
  int mod_words;        // 64-bit words per element in the C-ABI layout (ceil(mod_bits/64))
};

struct ModexpArgs {
  ModCtxDev ctx;
  const uint64_t* base;  // [count][base_stride] (base_stride == 0: one shared base)
  size_t base_stride;
  int base_words;        // valid words per base
  const uint64_t* exp;   // [count][exp_stride]  (exp_stride == 0: one shared exponent)
  size_t exp_stride;
  int exp_words;
  int exp_bits;          // max exponent bit length over the batch (mod_exp.cpp:484)
  int window;            // fixed window width w, 1..5
  uint64_t* out;         // [count][ctx.mod_words]
  uint32_t* table;       // [count rounded up to IPW][2^w][L] workspace
  size_t count;
};

struct ModmulArgs {
  ModCtxDev ctx;
  const uint64_t* a;     // [count][a_stride]
  size_t a_stride;
  const uint64_t* b;     // [count][b_stride]  (b_stride == 0: scalar broadcast)
  size_t b_stride;
  int in_words;          // valid words per operand
  uint64_t* out;         // [count][ctx.mod_words]
  size_t count;
};

// wave-level LDS hand-off: all lanes of the (single-wave) workgroup have finished their LDS
// writes before any lane reads.
__device__ __forceinline__ void wave_lds_sync() { __syncthreads(); }

// Load one element per group (64-bit words, C-ABI layout) into LDS, zero padded to W64+1.
template <class GEO>
__device__ __forceinline__ void stage_words(uint64_t (*io)[GEO::W64 + 1], const uint64_t* src,
                                            size_t stride, int words, size_t first_inst,
                                            size_t count, int lane) {
  constexpr int WW = GEO::W64 + 1;
  for (int t = lane; t < GEO::IPW * WW; t += kWave) {
    int g = t / WW, w = t % WW;
    size_t inst = first_inst + g;
    if (inst >= count) inst = count - 1;
    io[g][w] = (w < words) ? src[inst * stride + w] : 0;
  }
}

// Canonicalise r (value in [0, 2N) -> [0, N)) and store it as 64-bit words.
// Uses bl (limb scratch, [IPW][L]) and io ([IPW][W64+1]).
template <class GEO>
__device__ __forceinline__ void store_canonical(uint32_t (&r)[GEO::K], const ModCtxDev& ctx,
                                                uint32_t (*bl)[GEO::L], uint64_t (*io)[GEO::W64 + 1],
                                                uint64_t* out, size_t first_inst, size_t count,
                                                int lane, int g, int x) {
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
  // canonical reduction on the 64-bit words, one lane per group: while (V >= N) V -= N.
  // V < 3N on every path that reaches here, so two rounds suffice; loops stay rolled.
  if (x == 0) {
    uint64_t* v = io[g];
#pragma unroll 1
    for (int round = 0; round < 2; ++round) {
      // D = V - N into the spare half of the limb scratch; the final borrow says V < N
      uint64_t* dtmp = reinterpret_cast<uint64_t*>(bl[g]);
      uint64_t borrow = 0;
#pragma unroll 1
      for (int w = 0; w < W64; ++w) {
        uint64_t vw = v[w], nw = ctx.n64[w];
        uint64_t d = vw - nw - borrow;
        borrow = ((vw < nw) | ((vw == nw) & (borrow != 0))) ? 1 : 0;
        dtmp[w] = d;
      }
      if (borrow) break;
#pragma unroll 1
      for (int w = 0; w < W64; ++w) v[w] = dtmp[w];
    }
  }
  wave_lds_sync();
  const int mw = ctx.mod_words;
  for (int t = lane; t < GEO::IPW * mw; t += kWave) {
    int gg = t / mw, w = t % mw;
    size_t inst = first_inst + gg;
    if (inst < count) out[inst * (size_t)mw + w] = io[gg][w];
  }
}

template <class GEO>
__global__ __launch_bounds__(kWave) void modexp_kernel(ModexpArgs A) {
  constexpr int K = GEO::K, L = GEO::L, G = GEO::G, IPW = GEO::IPW;
  __shared__ uint32_t bl[IPW][L];
  __shared__ uint64_t io[IPW][GEO::W64 + 1];

  const int lane = threadIdx.x;
  const int g = lane / G, x = lane % G;
  const size_t first_inst = (size_t)blockIdx.x * IPW;
  size_t inst = first_inst + g;
  const size_t tinst = inst;                 // table slot (padded instances own a slot too)
  if (inst >= A.count) inst = A.count - 1;   // padded lanes recompute the last element

  uint32_t n[K], a[K];
#pragma unroll
  for (int j = 0; j < K; ++j) n[j] = A.ctx.n[x * K + j];
  const uint32_t n0inv = A.ctx.n0inv;

  // ---- base -> limbs, R^2 -> LDS ----
  stage_words<GEO>(io, A.base, A.base_stride, A.base_words, first_inst, A.count, lane);
#pragma unroll
  for (int j = 0; j < K; ++j) bl[g][x * K + j] = A.ctx.r2[x * K + j];
  wave_lds_sync();
#pragma unroll
  for (int j = 0; j < K; ++j) a[j] = limb_from_words(io[g], x * K + j);

  const int w = A.window;
  const int tsize = 1 << w;
  uint32_t* tbl = A.table + tinst * (size_t)tsize * L + x * K;   // this lane's slice of entry 0
  const uint64_t* ep = A.exp + inst * A.exp_stride;
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
  enum { TOMONT, TABLE, SQR, MUL, FINAL };
  int phase = TOMONT;
  int e = 2;         // next table entry to build
  int win = 0;       // current window index
  int sq = 0;        // squarings left in the current window
  for (;;) {
    montmul<GEO>(a, a, bl[g], n, n0inv);
    if (phase == FINAL) break;

    bool start_main = false;
    if (phase == TOMONT) {
      // a = base*R.  table[1] = a, table[0] = R mod N; multiplier for the table build = a.
      wave_lds_sync();
#pragma unroll
      for (int j = 0; j < K; ++j) {
        tbl[L + j] = a[j];
        tbl[j] = A.ctx.one[x * K + j];
        bl[g][x * K + j] = a[j];
      }
      wave_lds_sync();
      if (tsize > 2) phase = TABLE; else start_main = true;
    } else if (phase == TABLE) {
#pragma unroll
      for (int j = 0; j < K; ++j) tbl[(size_t)e * L + j] = a[j];
      if (++e == tsize) start_main = true;
    } else if (phase == SQR) {
      if (--sq == 0) phase = MUL;
    } else {  // MUL
      if (--win < 0) phase = FINAL; else { phase = SQR; sq = w; }
    }

    if (start_main) {
      // top window: a = table[d]
      if (nwin == 0) {
#pragma unroll
        for (int j = 0; j < K; ++j) a[j] = A.ctx.one[x * K + j];
        phase = FINAL;
      } else {
        // make this lane's table stores visible to its own loads (same lane wrote them)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        int d = digit(nwin - 1);
#pragma unroll
        for (int j = 0; j < K; ++j) a[j] = tbl[(size_t)d * L + j];
        win = nwin - 2;
        if (win < 0) phase = FINAL; else { phase = SQR; sq = w; }
      }
    }

    // ---- stage the multiplier operand of the next multiplication ----
    if (phase == SQR) {
      wave_lds_sync();
#pragma unroll
      for (int j = 0; j < K; ++j) bl[g][x * K + j] = a[j];
      wave_lds_sync();
    } else if (phase == MUL) {
      int d = digit(win);
      uint32_t t[K];
#pragma unroll
      for (int j = 0; j < K; ++j) t[j] = tbl[(size_t)d * L + j];
      wave_lds_sync();
#pragma unroll
      for (int j = 0; j < K; ++j) bl[g][x * K + j] = t[j];
      wave_lds_sync();
    } else if (phase == FINAL) {
      // leave the Montgomery domain: multiply by 1
      wave_lds_sync();
#pragma unroll
      for (int j = 0; j < K; ++j) bl[g][x * K + j] = (x == 0 && j == 0) ? 1u : 0u;
      wave_lds_sync();
    }
    // phase == TABLE: multiplier (base*R) is already staged
  }
  store_canonical<GEO>(a, A.ctx, bl, io, A.out, first_inst, A.count, lane, g, x);
}

// out = a*b mod N: montmul(montmul(a, R^2), b) -- two Montgomery multiplications.
template <class GEO>
__global__ __launch_bounds__(kWave) void modmul_kernel(ModmulArgs A) {
  constexpr int K = GEO::K, L = GEO::L, G = GEO::G, IPW = GEO::IPW;
  __shared__ uint32_t bl[IPW][L];
  __shared__ uint64_t io[IPW][GEO::W64 + 1];
  const int lane = threadIdx.x;
  const int g = lane / G, x = lane % G;
  const size_t first_inst = (size_t)blockIdx.x * IPW;

  uint32_t n[K], a[K];
#pragma unroll
  for (int j = 0; j < K; ++j) n[j] = A.ctx.n[x * K + j];
  const uint32_t n0inv = A.ctx.n0inv;

  stage_words<GEO>(io, A.a, A.a_stride, A.in_words, first_inst, A.count, lane);
#pragma unroll
  for (int j = 0; j < K; ++j) bl[g][x * K + j] = A.ctx.r2[x * K + j];
  wave_lds_sync();
#pragma unroll
  for (int j = 0; j < K; ++j) a[j] = limb_from_words(io[g], x * K + j);
  montmul<GEO>(a, a, bl[g], n, n0inv);      // a*R  (any a < R is fine)

  wave_lds_sync();
  stage_words<GEO>(io, A.b, A.b_stride, A.in_words, first_inst, A.count, lane);
  wave_lds_sync();
#pragma unroll
  for (int j = 0; j < K; ++j) bl[g][x * K + j] = limb_from_words(io[g], x * K + j);
  wave_lds_sync();
  montmul<GEO>(a, a, bl[g], n, n0inv);      // a*b mod N, lazy
  store_canonical<GEO>(a, A.ctx, bl, io, A.out, first_inst, A.count, lane, g, x);
}

}  // namespace pgpu

#endif  // PAILLIERCRYPTOLIB_AMD_CSRC_KERNELS_HPP_

/* ifma_oracle.c -- 8-lane AVX512-IFMA batched modexp.  TEST INFRASTRUCTURE ONLY (CPU baseline "B3").
 *
 * A restatement -- NOT IPP-Crypto -- of the kind of computation the reference's CPU hot path runs:
 * ipcl/mod_exp.cpp:446-533 marshals 8 (base, exp, mod) triples and calls crypto_mb's mbx_exp_mb8, which
 * (per the public IPP-Crypto documentation) evaluates 8 exponentiations in the
 * 8 x 64-bit lanes of a zmm register on radix-2^52 digits with vpmadd52{lo,hi}uq, fixed 5-bit window.
 * ipcl/mod_exp.cpp:597-636 runs the 8-element chunks under "omp parallel for"; so does this file.
 *
 * Arithmetic: "almost Montgomery multiplication" (AMM) with R = 2^(52 L), L = ceil((bits + 2) / 52), so
 * R >= 4 N: operands < 2N give results < 2N and no conditional subtraction is needed between
 * multiplications.  Column sums are kept in 64-bit lanes (12 bits of headroom, enough for 4 L <= 2^11
 * additions of < 2^52 terms) and normalised to 52-bit digits once per multiplication.
 *
 * Layout at the boundary: little-endian uint64 limbs, [element][limb], shared modulus (include/pgpu.h).
 * Built with -mavx512f -mavx512ifma; oracle/c_oracle.py loads it only when /proc/cpuinfo reports
 * avx512ifma. */
#include <immintrin.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef __m512i v8;
typedef uint64_t u64;
#define MASK52 ((1ULL << 52) - 1)
#define WIN 5
#define MAXL 160 /* moduli up to 8192 bits */

static inline v8 lo52(v8 acc, v8 a, v8 b) { return _mm512_madd52lo_epu64(acc, a, b); }
static inline v8 hi52(v8 acc, v8 a, v8 b) { return _mm512_madd52hi_epu64(acc, a, b); }

/* r = a * b / R  (almost) mod n; a, b, n: L vectors of digits < 2^52; t: scratch of L + 2 vectors.
 * Two multiplier digits per sweep over the columns (halves the loads / stores per multiply-add: the sweep
 * is memory-operand bound otherwise); the second quotient digit comes from the column the first row has
 * just completed.  Column j of row i lands in slot j-1, of row i+1 in slot j-2 (division by 2^52 each). */
static void amm(v8* r, const v8* a, const v8* b, const v8* n, v8 k0, int L, v8* t) {
  const v8 zero = _mm512_setzero_si512();
  for (int j = 0; j < L + 2; ++j) t[j] = zero;
  int i = 0;
  for (; i + 1 < L; i += 2) {
    const v8 b0 = b[i], b1 = b[i + 1];
    /* row i, column 0 */
    v8 c = lo52(t[0], a[0], b0);
    const v8 q0 = lo52(zero, c, k0);
    c = lo52(c, n[0], q0);
    v8 up0 = _mm512_srli_epi64(c, 52);                    /* into column 1 of row i */
    up0 = hi52(hi52(up0, a[0], b0), n[0], q0);
    /* row i, column 1 = first column of row i+1 */
    v8 d = lo52(lo52(t[1], a[1], b0), n[1], q0);
    d = _mm512_add_epi64(d, up0);
    up0 = hi52(hi52(zero, a[1], b0), n[1], q0);           /* into column 2 of row i */
    d = lo52(d, a[0], b1);
    const v8 q1 = lo52(zero, d, k0);
    d = lo52(d, n[0], q1);
    v8 up1 = _mm512_srli_epi64(d, 52);                    /* into column 1 of row i+1 */
    up1 = hi52(hi52(up1, a[0], b1), n[0], q1);
    for (int j = 2; j < L; ++j) {
      /* row i, column j  ->  its value is column j-1 of row i+1 */
      v8 s = lo52(lo52(t[j], a[j], b0), n[j], q0);
      s = _mm512_add_epi64(s, up0);
      up0 = hi52(hi52(zero, a[j], b0), n[j], q0);
      s = lo52(lo52(s, a[j - 1], b1), n[j - 1], q1);
      t[j - 2] = _mm512_add_epi64(s, up1);
      up1 = hi52(hi52(zero, a[j - 1], b1), n[j - 1], q1);
    }
    /* row i column L (= up0) is column L-1 of row i+1 */
    v8 s = lo52(lo52(up0, a[L - 1], b1), n[L - 1], q1);
    t[L - 2] = _mm512_add_epi64(s, up1);
    t[L - 1] = hi52(hi52(zero, a[L - 1], b1), n[L - 1], q1);
  }
  for (; i < L; ++i) {                                    /* odd L: the last row alone */
    const v8 bi = b[i];
    v8 c0 = lo52(t[0], a[0], bi);
    const v8 q = lo52(zero, c0, k0);          /* (column 0 mod 2^52) * (-1/n mod 2^52) mod 2^52 */
    c0 = lo52(c0, n[0], q);                   /* now a multiple of 2^52 */
    v8 up = _mm512_srli_epi64(c0, 52);        /* carry into column 1 */
    up = hi52(hi52(up, a[0], bi), n[0], q);
    for (int j = 1; j < L; ++j) {             /* column j lands in slot j-1: division by 2^52 */
      v8 s = lo52(lo52(t[j], a[j], bi), n[j], q);
      t[j - 1] = _mm512_add_epi64(s, up);
      up = hi52(hi52(zero, a[j], bi), n[j], q);
    }
    t[L - 1] = up;
  }
  v8 c = zero;
  const v8 mask = _mm512_set1_epi64((long long)MASK52);
  for (int j = 0; j < L; ++j) {
    const v8 s = _mm512_add_epi64(t[j], c);
    r[j] = _mm512_and_si512(s, mask);
    c = _mm512_srli_epi64(s, 52);
  }
}

/* dedicated squaring (3 L^2 multiply-adds instead of 4 L^2): cross products once and doubled, the diagonal,
 * then a reduction sweep that retires two columns per pass (the same blocking as amm). */
static void ams(v8* r, const v8* a, const v8* n, v8 k0, int L, v8* t /* 2L+2 */) {
  const v8 zero = _mm512_setzero_si512();
  for (int j = 0; j < 2 * L; ++j) t[j] = zero;
  /* cross products a_i a_j, i < j, once; two rows (i, i+1) per sweep over the columns:
   * column i+j collects lo(a_i a_j), hi(a_i a_{j-1}), lo(a_{i+1} a_{j-1}), hi(a_{i+1} a_{j-2}) */
  int i = 0;
  for (; i + 5 < L; i += 2) {
    const v8 x = a[i], y = a[i + 1];
    t[2 * i + 1] = lo52(t[2 * i + 1], a[i + 1], x);
    t[2 * i + 2] = hi52(lo52(t[2 * i + 2], a[i + 2], x), a[i + 1], x);
    t[2 * i + 3] = lo52(hi52(lo52(t[2 * i + 3], a[i + 3], x), a[i + 2], x), a[i + 2], y);
    for (int j = i + 4; j < L; ++j) {
      v8 s = hi52(lo52(t[i + j], a[j], x), a[j - 1], x);
      t[i + j] = hi52(lo52(s, a[j - 1], y), a[j - 2], y);
    }
    t[i + L] = hi52(lo52(hi52(t[i + L], a[L - 1], x), a[L - 1], y), a[L - 2], y);
    t[i + L + 1] = hi52(t[i + L + 1], a[L - 1], y);
  }
  for (; i < L; ++i) {                              /* remaining rows one at a time */
    const v8 ai = a[i];
    for (int j = i + 1; j < L; ++j) {
      t[i + j] = lo52(t[i + j], a[j], ai);
      t[i + j + 1] = hi52(t[i + j + 1], a[j], ai);
    }
  }
  for (int j = 0; j < 2 * L; ++j) t[j] = _mm512_slli_epi64(t[j], 1);
  for (int d = 0; d < L; ++d) {                     /* diagonal */
    t[2 * d] = lo52(t[2 * d], a[d], a[d]);
    t[2 * d + 1] = hi52(t[2 * d + 1], a[d], a[d]);
  }
  t[2 * L] = zero;
  t[2 * L + 1] = zero;
  i = 0;
  for (; i + 1 < L; i += 2) {                       /* reduction: clears columns i and i+1 */
    const v8 q0 = lo52(zero, t[i], k0);
    v8 c = lo52(t[i], n[0], q0);
    v8 d = _mm512_add_epi64(t[i + 1], _mm512_srli_epi64(c, 52));
    d = lo52(hi52(d, n[0], q0), n[1], q0);
    const v8 q1 = lo52(zero, d, k0);
    d = lo52(d, n[0], q1);
    /* column i+2 receives: carry of d, hi(n1 q0), lo(n2 q0), hi(n0 q1), lo(n1 q1) */
    v8 e = _mm512_add_epi64(t[i + 2], _mm512_srli_epi64(d, 52));
    e = hi52(hi52(e, n[1], q0), n[0], q1);
    if (L > 2) e = lo52(e, n[2], q0);
    e = lo52(e, n[1], q1);
    t[i + 2] = e;
    for (int j = 3; j < L; ++j) {                   /* column i+j: lo(n_j q0) hi(n_{j-1} q0) lo(n_{j-1} q1) hi(n_{j-2} q1) */
      v8 s = lo52(hi52(t[i + j], n[j - 1], q0), n[j], q0);
      t[i + j] = lo52(hi52(s, n[j - 2], q1), n[j - 1], q1);
    }
    if (L > 2) {
      /* column i+L: hi(n_{L-1} q0), lo(n_{L-1} q1), hi(n_{L-2} q1);  column i+L+1: hi(n_{L-1} q1) */
      v8 s = hi52(t[i + L], n[L - 1], q0);
      t[i + L] = lo52(hi52(s, n[L - 2], q1), n[L - 1], q1);
      t[i + L + 1] = hi52(t[i + L + 1], n[L - 1], q1);
    }
  }
  for (; i < L; ++i) {                              /* odd L: the last column alone */
    const v8 q = lo52(zero, t[i], k0);
    v8 c0 = lo52(t[i], n[0], q);
    t[i + 1] = _mm512_add_epi64(t[i + 1], _mm512_srli_epi64(c0, 52));
    t[i + 1] = hi52(t[i + 1], n[0], q);
    for (int j = 1; j < L; ++j) {
      t[i + j] = lo52(t[i + j], n[j], q);
      t[i + j + 1] = hi52(t[i + j + 1], n[j], q);
    }
  }
  v8 c = zero;
  const v8 mask = _mm512_set1_epi64((long long)MASK52);
  for (int j = 0; j < L; ++j) {
    const v8 s = _mm512_add_epi64(t[L + j], c);
    r[j] = _mm512_and_si512(s, mask);
    c = _mm512_srli_epi64(s, 52);
  }
}

static void to52(u64* d, int L, const u64* w, int words) {
  for (int i = 0; i < L; ++i) {
    const int bit = 52 * i, k = bit >> 6, s = bit & 63;
    u64 v = k < words ? w[k] >> s : 0;
    if (s > 12 && k + 1 < words) v |= w[k + 1] << (64 - s);
    d[i] = v & MASK52;
  }
}

static void from52(u64* w, int words, const u64* d, int L) {
  memset(w, 0, (size_t)words * 8);
  for (int i = 0; i < L; ++i) {
    const int bit = 52 * i, k = bit >> 6, s = bit & 63;
    if (k < words) w[k] |= d[i] << s;
    if (s > 12 && k + 1 < words) w[k + 1] |= d[i] >> (64 - s);
  }
}

static int ge_words(const u64* a, const u64* b, int n) {
  for (int i = n - 1; i >= 0; --i)
    if (a[i] != b[i]) return a[i] > b[i];
  return 1;
}

static void sub_words(u64* a, const u64* b, int n) {
  unsigned char br = 0;
  for (int i = 0; i < n; ++i) br = _subborrow_u64(br, a[i], b[i], (unsigned long long*)&a[i]);
}

static int top_bit(const u64* w, int words) {
  for (int i = words - 1; i >= 0; --i)
    if (w[i]) return 64 * i + 64 - __builtin_clzll(w[i]);
  return 0;
}

static inline unsigned window_at(const u64* e, int words, int pos) {   /* WIN bits starting at bit pos */
  const int k = pos >> 6, s = pos & 63;
  u64 v = k < words ? e[k] >> s : 0;
  if (s > 64 - WIN && k + 1 < words) v |= e[k + 1] << (64 - s);
  return (unsigned)(v & ((1u << WIN) - 1));
}

int orc_ifma_available(void) { return __builtin_cpu_supports("avx512ifma") ? 1 : 0; }

int orc_ifma_modexp_batch(const u64* base, size_t base_stride, const u64* exp, size_t exp_stride,
                          int exp_words, const u64* mod, int mod_words, u64* out, size_t count) {
  if (!(mod[0] & 1) || mod_words < 1) return -1;
  const int nbits = top_bit(mod, mod_words);
  const int L = (nbits + 2 + 51) / 52;
  if (L > MAXL || mod_words > 2 * MAXL - 2) return -1;
  /* AMM(base, R^2 mod N) stays < 2N for any base < R.  A wider base never reaches this seam from the
   * path's callers (they guarantee base < mod); refuse instead of reducing */
  if (64 * mod_words > 52 * L)
    for (size_t i = 0; i < (base_stride ? count : 1); ++i)
      if (top_bit(base + i * base_stride, mod_words) > 52 * L) return -2;
  int ebits = 0;                                   /* max over the batch, as mod_exp.cpp:480-484 */
  for (size_t i = 0; i < (exp_stride ? count : 1); ++i) {
    const int b = top_bit(exp + i * exp_stride, exp_words);
    if (b > ebits) ebits = b;
  }
  const int nwin = ebits ? (ebits + WIN - 1) / WIN : 1;

  /* shared constants in radix 2^52 */
  const int W1 = mod_words + 2;
  u64* nd = (u64*)calloc((size_t)L, 8);
  u64* r2d = (u64*)calloc((size_t)L, 8);
  u64* x = (u64*)calloc((size_t)W1, 8);
  u64* nw = (u64*)calloc((size_t)W1, 8);
  memcpy(nw, mod, (size_t)mod_words * 8);
  to52(nd, L, mod, mod_words);
  x[0] = 1;                                        /* x = 2^(2*52*L) mod N by doubling */
  for (int i = 0; i < 2 * 52 * L; ++i) {
    u64 carry = 0;
    for (int k = 0; k < W1; ++k) {
      const u64 v = x[k];
      x[k] = (v << 1) | carry;
      carry = v >> 63;
    }
    if (ge_words(x, nw, W1)) sub_words(x, nw, W1);
  }
  to52(r2d, L, x, mod_words);
  u64 inv = 1;                                     /* -1/n mod 2^52 by Newton */
  for (int i = 0; i < 6; ++i) inv *= 2 - mod[0] * inv;
  const u64 k0s = (0 - inv) & MASK52;
  free(x);
  free(nw);

  const size_t groups = (count + 7) / 8;
  int bad = 0;
#pragma omp parallel
  {
    const size_t vec = (size_t)L;
    v8* mem = (v8*)aligned_alloc(64, sizeof(v8) * vec * (5 + (1u << WIN)) + sizeof(v8) * (2 * vec + 4));
    if (!mem) {
#pragma omp atomic write
      bad = 1;
    }
    v8 *N = mem, *R2 = N + vec, *acc = R2 + vec, *one = acc + vec, *mul = one + vec, *tab = mul + vec,
       *t = tab + vec * (1u << WIN);
    u64 lane[8][MAXL], wtmp[2 * MAXL];
    const v8 k0 = _mm512_set1_epi64((long long)k0s);
    if (mem) {
      for (int j = 0; j < L; ++j) {
        N[j] = _mm512_set1_epi64((long long)nd[j]);
        R2[j] = _mm512_set1_epi64((long long)r2d[j]);
        one[j] = _mm512_set1_epi64(j == 0 ? 1 : 0);
      }
    }
#pragma omp for schedule(dynamic, 1)
    for (long g = 0; g < (long)groups; ++g) {
      if (!mem) continue;
      const size_t first = (size_t)g * 8;
      const int live = count - first < 8 ? (int)(count - first) : 8;
      const u64* ep[8];
      for (int l = 0; l < 8; ++l) {
        const size_t e = first + (l < live ? l : 0);
        to52(lane[l], L, base + e * base_stride, mod_words);
        ep[l] = exp + e * exp_stride;
      }
      for (int j = 0; j < L; ++j)
        acc[j] = _mm512_set_epi64((long long)lane[7][j], (long long)lane[6][j], (long long)lane[5][j],
                                  (long long)lane[4][j], (long long)lane[3][j], (long long)lane[2][j],
                                  (long long)lane[1][j], (long long)lane[0][j]);
      /* table: tab[k] = base^k in the Montgomery domain */
      amm(tab + vec, acc, R2, N, k0, L, t);
      amm(tab, R2, one, N, k0, L, t);
      for (unsigned k = 2; k < (1u << WIN); ++k)
        (k & 1) ? amm(tab + vec * k, tab + vec * (k - 1), tab + vec, N, k0, L, t)
                : ams(tab + vec * k, tab + vec * (k / 2), N, k0, L, t);
      const v8 lane_id = _mm512_set_epi64(7, 6, 5, 4, 3, 2, 1, 0);
      for (int w = nwin - 1; w >= 0; --w) {
        long long idx[8];
        for (int l = 0; l < 8; ++l) idx[l] = (long long)window_at(ep[l], exp_words, w * WIN) * (long long)(vec * 8);
        const v8 vidx = _mm512_add_epi64(_mm512_loadu_si512((const void*)idx), lane_id);
        for (int j = 0; j < L; ++j)
          mul[j] = _mm512_i64gather_epi64(_mm512_add_epi64(vidx, _mm512_set1_epi64(8LL * j)), (const void*)tab, 8);
        if (w == nwin - 1) {
          memcpy(acc, mul, sizeof(v8) * vec);
        } else {
          for (int s = 0; s < WIN; ++s) ams(acc, acc, N, k0, L, t);
          amm(acc, acc, mul, N, k0, L, t);
        }
      }
      amm(acc, acc, one, N, k0, L, t);             /* leave the Montgomery domain: result <= N */
      for (int l = 0; l < live; ++l) {
        u64 d[MAXL];
        for (int j = 0; j < L; ++j) d[j] = ((const u64*)&acc[j])[l];
        from52(wtmp, mod_words + 1, d, L);
        if (wtmp[mod_words] || ge_words(wtmp, mod, mod_words)) sub_words(wtmp, mod, mod_words);
        memcpy(out + (first + (size_t)l) * (size_t)mod_words, wtmp, (size_t)mod_words * 8);
      }
    }
    free(mem);
  }
  free(nd);
  free(r2d);
  return bad ? -3 : 0;
}

/* ---- fixed-base exponentiation base^exp[i] for ONE base (round 4: cpu_baseline.legs.ifma_fixed_base) ----
 * What the GPU step does for the DJN obfuscator hs^r (pub_key.cpp:51-64 computes it by square-and-multiply through
 * ippMBModExp): hs is a key constant, so hs^r = prod_i T[i][d_i] with T[i][d] = hs^(d * 2^(w i)) and d_i the i-th w-bit
 * digit of r -- ceil(bits / w) - 1 products, no squarings, against a per-key table built once.  This is the same
 * algorithm on the 8-lane IFMA arithmetic above, so that the encrypt third of bench.py's CPU baseline can be compared
 * like for like.  Table: nwin x 2^w entries of L radix-2^52 digits (Montgomery form), built by orc_ifma_fb_build --
 * outside the timed region, like the GPU's -- and gathered per lane.  Results are canonical and bit-identical with
 * orc_ifma_modexp_batch (tests/test_oracle_c.py). */
typedef struct orc_fb_table {
  int L, w, nwin, mod_words;
  u64 k0s;
  u64* nd;   /* L digits of the modulus */
  u64* tab;  /* [nwin][2^w][L] */
} orc_fb_table;

void orc_ifma_fb_free(orc_fb_table* t) {
  if (!t) return;
  free(t->nd);
  free(t->tab);
  free(t);
}

orc_fb_table* orc_ifma_fb_build(const u64* base, const u64* mod, int mod_words, int exp_bits, int w) {
  if (!(mod[0] & 1) || mod_words < 1 || w < 1 || w > 12 || exp_bits < 1) return NULL;
  const int nbits = top_bit(mod, mod_words);
  const int L = (nbits + 2 + 51) / 52;
  if (L > MAXL || mod_words > 2 * MAXL - 2) return NULL;
  if (64 * mod_words > 52 * L && top_bit(base, mod_words) > 52 * L) return NULL;
  orc_fb_table* T = (orc_fb_table*)calloc(1, sizeof(*T));
  T->L = L;
  T->w = w;
  T->nwin = (exp_bits + w - 1) / w;
  T->mod_words = mod_words;
  T->nd = (u64*)calloc((size_t)L, 8);
  const size_t ent = (size_t)1 << w;
  T->tab = (u64*)aligned_alloc(64, (((size_t)T->nwin * ent * (size_t)L * 8) + 63) & ~(size_t)63);
  const int W1 = mod_words + 2;
  u64* r2d = (u64*)calloc((size_t)L, 8);
  u64* x = (u64*)calloc((size_t)W1, 8);
  u64* nw = (u64*)calloc((size_t)W1, 8);
  memcpy(nw, mod, (size_t)mod_words * 8);
  to52(T->nd, L, mod, mod_words);
  x[0] = 1;
  for (int i = 0; i < 2 * 52 * L; ++i) {
    u64 carry = 0;
    for (int k = 0; k < W1; ++k) {
      const u64 v = x[k];
      x[k] = (v << 1) | carry;
      carry = v >> 63;
    }
    if (ge_words(x, nw, W1)) sub_words(x, nw, W1);
  }
  to52(r2d, L, x, mod_words);
  u64 inv = 1;
  for (int i = 0; i < 6; ++i) inv *= 2 - mod[0] * inv;
  T->k0s = (0 - inv) & MASK52;
  free(x);
  free(nw);
  const size_t vec = (size_t)L;
  /* g[i] = base^(2^(w i)) in the Montgomery domain: one chain of squarings (every lane the same value) */
  u64* g = (u64*)calloc((size_t)T->nwin * vec, 8);
  u64 bd[MAXL];
  to52(bd, L, base, mod_words);
  {
    v8* mem = (v8*)aligned_alloc(64, sizeof(v8) * (vec * 4 + 2 * vec + 4));
    v8 *N = mem, *R2 = N + vec, *acc = R2 + vec, *one = acc + vec, *t = one + vec;
    const v8 k0 = _mm512_set1_epi64((long long)T->k0s);
    for (int j = 0; j < L; ++j) {
      N[j] = _mm512_set1_epi64((long long)T->nd[j]);
      R2[j] = _mm512_set1_epi64((long long)r2d[j]);
      acc[j] = _mm512_set1_epi64((long long)bd[j]);
    }
    amm(acc, acc, R2, N, k0, L, t);
    for (int i = 0; i < T->nwin; ++i) {
      for (int j = 0; j < L; ++j) g[(size_t)i * vec + j] = ((const u64*)&acc[j])[0];
      if (i + 1 < T->nwin)
        for (int s = 0; s < w; ++s) ams(acc, acc, N, k0, L, t);
    }
    free(mem);
  }
  /* rows: eight windows per vector, T[i][d] = T[i][d-1] * g[i]; T[i][0] = R mod N */
  const int groups = (T->nwin + 7) / 8;
#pragma omp parallel
  {
    v8* mem = (v8*)aligned_alloc(64, sizeof(v8) * (vec * 5 + 2 * vec + 4));
    v8 *N = mem, *R2 = N + vec, *acc = R2 + vec, *gv = acc + vec, *one = gv + vec, *t = one + vec;
    const v8 k0 = _mm512_set1_epi64((long long)T->k0s);
    for (int j = 0; j < L; ++j) {
      N[j] = _mm512_set1_epi64((long long)T->nd[j]);
      R2[j] = _mm512_set1_epi64((long long)r2d[j]);
      one[j] = _mm512_set1_epi64(j == 0 ? 1 : 0);
    }
#pragma omp for schedule(dynamic, 1)
    for (int gr = 0; gr < groups; ++gr) {
      int win[8];
      for (int l = 0; l < 8; ++l) win[l] = gr * 8 + l < T->nwin ? gr * 8 + l : T->nwin - 1;
      for (int j = 0; j < L; ++j) {
        u64 lane[8];
        for (int l = 0; l < 8; ++l) lane[l] = g[(size_t)win[l] * vec + j];
        gv[j] = _mm512_loadu_si512((const void*)lane);
      }
      amm(acc, R2, one, N, k0, L, t);                 /* R mod N: entry 0 */
      for (size_t d = 0; d < ent; ++d) {
        if (d == 1) memcpy(acc, gv, sizeof(v8) * vec);
        else if (d > 1) amm(acc, acc, gv, N, k0, L, t);
        for (int l = 0; l < 8; ++l) {
          if (gr * 8 + l >= T->nwin) continue;
          u64* dst = T->tab + ((size_t)win[l] * ent + d) * vec;
          for (int j = 0; j < L; ++j) dst[j] = ((const u64*)&acc[j])[l];
        }
      }
    }
    free(mem);
  }
  free(g);
  free(r2d);
  return T;
}

int orc_ifma_fb_modexp_batch(const orc_fb_table* T, const u64* exp, size_t exp_stride, int exp_words, u64* out, size_t count) {
  if (!T) return -1;
  const int L = T->L, w = T->w, nwin = T->nwin, mod_words = T->mod_words;
  const size_t vec = (size_t)L, ent = (size_t)1 << w;
  u64 modw[2 * MAXL];
  from52(modw, mod_words, T->nd, L);
  for (size_t i = 0; i < count; ++i)
    if (top_bit(exp + i * exp_stride, exp_words) > w * nwin) return -2;   /* exponent wider than the table covers */
  const size_t groups = (count + 7) / 8;
#pragma omp parallel
  {
    v8* mem = (v8*)aligned_alloc(64, sizeof(v8) * (vec * 4 + 2 * vec + 4));
    v8 *N = mem, *acc = N + vec, *mul = acc + vec, *one = mul + vec, *t = one + vec;
    const v8 k0 = _mm512_set1_epi64((long long)T->k0s);
    u64 wtmp[2 * MAXL];
    for (int j = 0; j < L; ++j) {
      N[j] = _mm512_set1_epi64((long long)T->nd[j]);
      one[j] = _mm512_set1_epi64(j == 0 ? 1 : 0);
    }
#pragma omp for schedule(dynamic, 4)
    for (long g = 0; g < (long)groups; ++g) {
      const size_t first = (size_t)g * 8;
      const int live = count - first < 8 ? (int)(count - first) : 8;
      const u64* ep[8];
      for (int l = 0; l < 8; ++l) ep[l] = exp + (first + (size_t)(l < live ? l : 0)) * exp_stride;
      for (int i = 0; i < nwin; ++i) {
        long long idx[8];
        for (int l = 0; l < 8; ++l) {
          const int pos = i * w, k = pos >> 6, s = pos & 63;
          u64 v = k < exp_words ? ep[l][k] >> s : 0;
          if (s + w > 64 && k + 1 < exp_words) v |= ep[l][k + 1] << (64 - s);
          idx[l] = (long long)(((size_t)i * ent + (size_t)(v & (ent - 1))) * vec);
        }
        const v8 vidx = _mm512_loadu_si512((const void*)idx);
        v8* dst = i == 0 ? acc : mul;
        for (int j = 0; j < L; ++j)
          dst[j] = _mm512_i64gather_epi64(_mm512_add_epi64(vidx, _mm512_set1_epi64(j)), (const void*)T->tab, 8);
        if (i > 0) amm(acc, acc, mul, N, k0, L, t);
      }
      amm(acc, acc, one, N, k0, L, t);
      for (int l = 0; l < live; ++l) {
        u64 d[MAXL];
        for (int j = 0; j < L; ++j) d[j] = ((const u64*)&acc[j])[l];
        from52(wtmp, mod_words + 1, d, L);
        if (wtmp[mod_words] || ge_words(wtmp, modw, mod_words)) sub_words(wtmp, modw, mod_words);
        memcpy(out + (first + (size_t)l) * (size_t)mod_words, wtmp, (size_t)mod_words * 8);
      }
    }
    free(mem);
  }
  return 0;
}

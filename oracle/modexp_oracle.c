/* modexp_oracle.c -- CPU restatement of the reference's hot path in plain C.
 * TEST INFRASTRUCTURE ONLY: used by tests/ (as an independent checker next to CPython pow) and
 * by bench.py's cpu_baseline leg (kind "port").  The product (libpgpu.so) never links it.
 *
 * What it restates (file:line under the reference tree):
 *   orc_modexp_batch   -- ipcl::modExp over a batch (ipcl/mod_exp.cpp:597-636,680-737): chunks
 *                         under `#pragma omp parallel for`, each element a Montgomery fixed-window
 *                         exponentiation.  The arithmetic itself is IPP-Crypto's mbx_exp_mb8
 *                         (not in the tree: intel/ipp-crypto tag ippcp_2021.6); restated here from
 *                         the published algorithm: CIOS Montgomery multiplication on 64-bit limbs
 *                         (Koc/Acar/Kaliski 1996) + fixed 5-bit window, result converted out of
 *                         Montgomery form by a multiplication with 1 (mod_exp.cpp:576-579).
 *   orc_modmul_batch   -- CipherText::raw_add  a*b % n^2 (ciphertext.cpp:135-141)
 *   orc_paillier_encrypt     -- PublicKey::raw_encrypt + applyObfuscator (pub_key.cpp:51-110)
 *   orc_paillier_decrypt_crt -- PrivateKey::decryptCRT (pri_key.cpp:114-157)
 *
 * Pinning: checked in tests/test_oracle_c.py against the reference's ISO/IEC 18033-6 vectors
 * (tests/golden/iso_kat.json) and against CPython pow on seeded inputs.
 *
 * Layout: little-endian uint64 limbs, [element][limb] row-major, identical to include/pgpu.h.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef uint64_t u64;

#define MAXW 200 /* limbs; 12800 bits */

/* ---------- small multi-precision helpers (fixed width w limbs unless noted) ---------- */
static int bn_cmp(const u64* a, const u64* b, int w) {
  for (int i = w - 1; i >= 0; --i)
    if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
  return 0;
}
static u64 bn_add(u64* r, const u64* a, const u64* b, int w) {
  u64 c = 0;
  for (int i = 0; i < w; ++i) {
    u128 s = (u128)a[i] + b[i] + c;
    r[i] = (u64)s;
    c = (u64)(s >> 64);
  }
  return c;
}
static u64 bn_sub(u64* r, const u64* a, const u64* b, int w) {
  u64 br = 0;
  for (int i = 0; i < w; ++i) {
    u128 d = (u128)a[i] - b[i] - br;
    r[i] = (u64)d;
    br = (u64)(d >> 64) & 1;
  }
  return br;
}
/* r[0..wa+wb) = a * b */
static void bn_mul(u64* r, const u64* a, int wa, const u64* b, int wb) {
  memset(r, 0, sizeof(u64) * (size_t)(wa + wb));
  for (int i = 0; i < wa; ++i) {
    u64 c = 0;
    for (int j = 0; j < wb; ++j) {
      u128 t = (u128)a[i] * b[j] + r[i + j] + c;
      r[i + j] = (u64)t;
      c = (u64)(t >> 64);
    }
    r[i + wb] = c;
  }
}
static int bn_bits(const u64* a, int w) {
  for (int i = w - 1; i >= 0; --i)
    if (a[i]) return i * 64 + 64 - __builtin_clzll(a[i]);
  return 0;
}
/* q = a / d (wa limbs, may be NULL), r = a % d (wd limbs, may be NULL): schoolbook long division on 64-bit
 * digits (Knuth 4.3.1 D).  This is the host "glue" arithmetic of the reference (BigNumber / and %,
 * bignum.cpp:204-308 -> ippsDiv_BN / ippsMod_BN), so it must not be slower than a word-level division. */
static void bn_divrem(u64* q, u64* r, const u64* a, int wa, const u64* d, int wd) {
  typedef unsigned __int128 u128;
  int n = wd;
  while (n > 0 && d[n - 1] == 0) --n;               /* significant limbs of the divisor (n > 0 required) */
  int m = wa;
  while (m > 0 && a[m - 1] == 0) --m;
  if (q) memset(q, 0, sizeof(u64) * (size_t)wa);
  if (m < n) {
    if (r) { memset(r, 0, sizeof(u64) * (size_t)wd); memcpy(r, a, sizeof(u64) * (size_t)m); }
    return;
  }
  if (n == 1) {
    u128 rem = 0;
    for (int i = m - 1; i >= 0; --i) {
      u128 cur = (rem << 64) | a[i];
      u64 qd = (u64)(cur / d[0]);
      rem = cur - (u128)qd * d[0];
      if (q) q[i] = qd;
    }
    if (r) { memset(r, 0, sizeof(u64) * (size_t)wd); r[0] = (u64)rem; }
    return;
  }
  u64 dn[MAXW + 1], an[2 * MAXW + 3];
  const int sh = __builtin_clzll(d[n - 1]);
  for (int i = n - 1; i > 0; --i) dn[i] = sh ? (d[i] << sh) | (d[i - 1] >> (64 - sh)) : d[i];
  dn[0] = d[0] << sh;
  an[m] = sh ? a[m - 1] >> (64 - sh) : 0;
  for (int i = m - 1; i > 0; --i) an[i] = sh ? (a[i] << sh) | (a[i - 1] >> (64 - sh)) : a[i];
  an[0] = a[0] << sh;
  for (int j = m - n; j >= 0; --j) {
    u128 num = ((u128)an[j + n] << 64) | an[j + n - 1];
    u128 qhat = num / dn[n - 1], rhat = num - qhat * dn[n - 1];
    while (qhat >> 64 || (u128)(u64)qhat * dn[n - 2] > ((rhat << 64) | an[j + n - 2])) {
      --qhat;
      rhat += dn[n - 1];
      if (rhat >> 64) break;
    }
    u128 borrow = 0, carry = 0;                     /* an[j..j+n] -= qhat * dn */
    for (int i = 0; i < n; ++i) {
      u128 pr = (u128)(u64)qhat * dn[i] + carry;
      carry = pr >> 64;
      u128 sub = (u128)an[i + j] - (u64)pr - borrow;
      an[i + j] = (u64)sub;
      borrow = (sub >> 64) & 1;
    }
    u128 sub = (u128)an[j + n] - carry - borrow;
    an[j + n] = (u64)sub;
    if ((sub >> 64) & 1) {                          /* one too many: add the divisor back */
      --qhat;
      u128 c = 0;
      for (int i = 0; i < n; ++i) {
        u128 t = (u128)an[i + j] + dn[i] + c;
        an[i + j] = (u64)t;
        c = t >> 64;
      }
      an[j + n] += (u64)c;
    }
    if (q) q[j] = (u64)qhat;
  }
  if (r) {
    memset(r, 0, sizeof(u64) * (size_t)wd);
    for (int i = 0; i < n; ++i) r[i] = sh ? (an[i] >> sh) | (an[i + 1] << (64 - sh)) : an[i];
  }
}

/* r = a mod m; a has wa limbs, m has w limbs */
static void bn_mod(u64* r, const u64* a, int wa, const u64* m, int w) { bn_divrem(NULL, r, a, wa, m, w); }

/* ---------- Montgomery context ---------- */
typedef struct {
  int w;          /* limbs */
  u64 n[MAXW];    /* modulus (odd) */
  u64 r2[MAXW];   /* R^2 mod n, R = 2^(64w) */
  u64 one[MAXW];  /* R mod n */
  u64 n0inv;      /* -n^-1 mod 2^64 */
} mont_t;

static void mont_init(mont_t* M, const u64* n, int w) {
  M->w = w;
  memcpy(M->n, n, sizeof(u64) * (size_t)w);
  u64 inv = n[0];
  for (int i = 0; i < 6; ++i) inv *= 2 - n[0] * inv;
  M->n0inv = 0 - inv;
  u64 t[2 * MAXW + 1];
  memset(t, 0, sizeof(u64) * (size_t)(2 * w + 1));
  t[w] = 1;
  bn_mod(M->one, t, w + 1, n, w);
  t[w] = 0;
  t[2 * w] = 1;
  bn_mod(M->r2, t, 2 * w + 1, n, w);
}

/* CIOS Montgomery multiplication: r = a*b*R^-1 mod n, a,b < n */
static void mont_mul(const mont_t* M, u64* r, const u64* a, const u64* b) {
  const int w = M->w;
  u64 t[MAXW + 2];
  memset(t, 0, sizeof(u64) * (size_t)(w + 2));
  for (int i = 0; i < w; ++i) {
    u64 c = 0;
    for (int j = 0; j < w; ++j) {
      u128 s = (u128)a[j] * b[i] + t[j] + c;
      t[j] = (u64)s;
      c = (u64)(s >> 64);
    }
    u128 s = (u128)t[w] + c;
    t[w] = (u64)s;
    t[w + 1] = (u64)(s >> 64);
    u64 q = t[0] * M->n0inv;
    s = (u128)q * M->n[0] + t[0];
    c = (u64)(s >> 64);
    for (int j = 1; j < w; ++j) {
      s = (u128)q * M->n[j] + t[j] + c;
      t[j - 1] = (u64)s;
      c = (u64)(s >> 64);
    }
    s = (u128)t[w] + c;
    t[w - 1] = (u64)s;
    t[w] = t[w + 1] + (u64)(s >> 64);
  }
  if (t[w] || bn_cmp(t, M->n, w) >= 0) bn_sub(t, t, M->n, w);
  memcpy(r, t, sizeof(u64) * (size_t)w);
}

/* out = base^exp mod n; base < 2^(64w) arbitrary (reduced), exp has ew limbs */
static void mont_modexp(const mont_t* M, u64* out, const u64* base, const u64* exp, int ew) {
  const int w = M->w, WB = 5;
  u64 (*tbl)[MAXW] = malloc(sizeof(u64[MAXW]) * 32);
  u64 b[MAXW], acc[MAXW], unit[MAXW];
  if (bn_cmp(base, M->n, w) < 0) memcpy(b, base, sizeof(u64) * (size_t)w);
  else bn_mod(b, base, w, M->n, w);
  mont_mul(M, tbl[1], b, M->r2);
  memcpy(tbl[0], M->one, sizeof(u64) * (size_t)w);
  for (int i = 2; i < 32; ++i) mont_mul(M, tbl[i], tbl[i - 1], tbl[1]);
  int ebits = bn_bits(exp, ew);
  int nwin = (ebits + WB - 1) / WB;
  memcpy(acc, M->one, sizeof(u64) * (size_t)w);
  for (int i = nwin - 1; i >= 0; --i) {
    if (i != nwin - 1)
      for (int s = 0; s < WB; ++s) mont_mul(M, acc, acc, acc);
    int bit = i * WB, word = bit >> 6, sh = bit & 63;
    u64 v = exp[word] >> sh;
    if (sh + WB > 64 && word + 1 < ew) v |= exp[word + 1] << (64 - sh);
    mont_mul(M, acc, acc, tbl[v & 31]);
  }
  memset(unit, 0, sizeof(u64) * (size_t)w);
  unit[0] = 1;
  mont_mul(M, out, acc, unit);
  free(tbl);
}

/* ---------- exported batch entry points ---------- */
int orc_modexp_batch(const u64* base, size_t base_stride, const u64* exp, size_t exp_stride,
                     int exp_words, const u64* mod, int mod_words, u64* out, size_t count) {
  if (mod_words > MAXW || !(mod[0] & 1)) return -1;
  mont_t* M = malloc(sizeof(mont_t));
  mont_init(M, mod, mod_words);
#pragma omp parallel for schedule(dynamic, 1)
  for (long i = 0; i < (long)count; ++i)
    mont_modexp(M, out + (size_t)i * mod_words, base + (size_t)i * base_stride,
                exp + (size_t)i * exp_stride, exp_words);
  free(M);
  return 0;
}

int orc_modmul_batch(const u64* a, const u64* b, size_t b_stride, const u64* mod, int mod_words,
                     u64* out, size_t count) {
  if (mod_words > MAXW) return -1;
#pragma omp parallel for schedule(static)
  for (long i = 0; i < (long)count; ++i) {
    u64 t[2 * MAXW];
    bn_mul(t, a + (size_t)i * mod_words, mod_words, b + (size_t)i * b_stride, mod_words);
    bn_mod(out + (size_t)i * mod_words, t, 2 * mod_words, mod, mod_words);
  }
  return 0;
}

/* c = obf * ((n*m + 1) % n^2) % n^2     pub_key.cpp:88-89,105 */
static void enc_finish_one(const u64* n, int nw, const u64* nsq, const u64* m, int m_words, const u64* obf,
                           u64* c) {
  const int W = 2 * nw;
  u64 g[2 * MAXW + 1], gm[MAXW], t[2 * MAXW];
  memset(g, 0, sizeof(g));
  bn_mul(g, n, nw, m, m_words);
  int gw = nw + m_words;
  for (int j = 0; j < gw + 1; ++j)
    if (++g[j]) break;
  bn_mod(gm, g, gw + 1, nsq, W);
  bn_mul(t, gm, W, obf, W);
  bn_mod(c, t, 2 * W, nsq, W);
}

/* c[i] = obf[i] * (1 + n*m[i]) mod n^2, obf = hs^r (djn) or r^n.   nw = limbs of n. */
int orc_paillier_encrypt(const u64* n, int nw, const u64* hs_or_null, const u64* m, int m_words,
                         const u64* r, int r_words, u64* c, size_t count) {
  const int W = 2 * nw;
  if (W > MAXW) return -1;
  u64 nsq[MAXW];
  bn_mul(nsq, n, nw, n, nw);
  mont_t* M = malloc(sizeof(mont_t));
  mont_init(M, nsq, W);
#pragma omp parallel for schedule(dynamic, 1)
  for (long i = 0; i < (long)count; ++i) {
    u64 obf[MAXW], base[MAXW];
    if (hs_or_null) {
      mont_modexp(M, obf, hs_or_null, r + (size_t)i * r_words, r_words);   /* pub_key.cpp:63 */
    } else {
      memset(base, 0, sizeof(u64) * (size_t)W);
      memcpy(base, r + (size_t)i * r_words, sizeof(u64) * (size_t)(r_words < W ? r_words : W));
      mont_modexp(M, obf, base, n, nw);                                   /* pub_key.cpp:79 */
    }
    enc_finish_one(n, nw, nsq, m + (size_t)i * m_words, m_words, obf, c + (size_t)i * W);
  }
  free(M);
  return 0;
}

/* exact division helper: q = a / d for a an exact multiple (the L function, pri_key.cpp:154-157) */
static void bn_divexact(u64* q, const u64* a, int wa, const u64* d, int wd) { bn_divrem(q, NULL, a, wa, d, wd); }

/* from rp = c^(p-1) mod p^2, rq = c^(q-1) mod q^2 (clobbered) to the plaintext: L-function, *hp / *hq,
 * CRT recombination   pri_key.cpp:142-157 */
static void dec_finish_one(const u64* p, const u64* q, int pw, const u64* hp, const u64* hq, const u64* pinv,
                           u64* rp, u64* rq, u64* mi) {
  const int hw = 2 * pw;
  u64 lp[MAXW], lq[MAXW], t[2 * MAXW], dp[MAXW], dq[MAXW], onev[MAXW];
  memset(onev, 0, sizeof(u64) * (size_t)hw);
  onev[0] = 1;
  bn_sub(rp, rp, onev, hw);                            /* L(x) = (x-1)/p  pri_key.cpp:154-157 */
  bn_sub(rq, rq, onev, hw);
  bn_divexact(lp, rp, hw, p, pw);
  bn_divexact(lq, rq, hw, q, pw);
  bn_mul(t, lp, pw, hp, pw);
  bn_mod(dp, t, 2 * pw, p, pw);                        /* pri_key.cpp:142 */
  bn_mul(t, lq, pw, hq, pw);
  bn_mod(dq, t, 2 * pw, q, pw);                        /* pri_key.cpp:143 */
  /* u = (dq - dp) * pinv mod q, non-negative residue   pri_key.cpp:150 */
  u64 d[MAXW], u[MAXW];
  u64 dpq[MAXW];
  bn_mod(dpq, dp, pw, q, pw);
  if (bn_cmp(dq, dpq, pw) >= 0) {
    bn_sub(d, dq, dpq, pw);
  } else {
    bn_add(d, dq, q, pw);
    bn_sub(d, d, dpq, pw);
  }
  bn_mul(t, d, pw, pinv, pw);
  bn_mod(u, t, 2 * pw, q, pw);
  bn_mul(t, u, pw, p, pw);                             /* pri_key.cpp:151 */
  memset(mi, 0, sizeof(u64) * (size_t)hw);
  memcpy(mi, dp, sizeof(u64) * (size_t)pw);
  bn_add(mi, mi, t, hw);
}

/* CRT decrypt.  p < q REQUIRED (caller orders them, pri_key.cpp:19-22); pw limbs each;
 * hp, hq, pinv (= p^-1 mod q) host-precomputed by the caller (pri_key.cpp:27-29), pw limbs.
 * c: 4*pw limbs per element, m: 2*pw limbs per element. */
int orc_paillier_decrypt_crt(const u64* p, const u64* q, int pw, const u64* hp, const u64* hq,
                             const u64* pinv, const u64* c, u64* m, size_t count) {
  const int hw = 2 * pw, cw = 4 * pw;
  if (cw > MAXW) return -1;
  u64 psq[MAXW], qsq[MAXW], pm1[MAXW], qm1[MAXW], onev[MAXW];
  bn_mul(psq, p, pw, p, pw);
  bn_mul(qsq, q, pw, q, pw);
  memset(onev, 0, sizeof(onev));
  onev[0] = 1;
  bn_sub(pm1, p, onev, pw);
  bn_sub(qm1, q, onev, pw);
  mont_t* Mp = malloc(sizeof(mont_t));
  mont_t* Mq = malloc(sizeof(mont_t));
  mont_init(Mp, psq, hw);
  mont_init(Mq, qsq, hw);
#pragma omp parallel for schedule(dynamic, 1)
  for (long i = 0; i < (long)count; ++i) {
    const u64* ci = c + (size_t)i * cw;
    u64 bp[MAXW], bq[MAXW], rp[MAXW], rq[MAXW];
    bn_mod(bp, ci, cw, psq, hw);                         /* pri_key.cpp:128 */
    bn_mod(bq, ci, cw, qsq, hw);                         /* pri_key.cpp:129 */
    mont_modexp(Mp, rp, bp, pm1, pw);                    /* pri_key.cpp:133 */
    mont_modexp(Mq, rq, bq, qm1, pw);                    /* pri_key.cpp:134 */
    dec_finish_one(p, q, pw, hp, hq, pinv, rp, rq, m + (size_t)i * hw);
  }
  free(Mp);
  free(Mq);
  return 0;
}

/* ---------- split entry points: the same flows with the modexps done by another backend ----------
 * (oracle/ifma_oracle.c or OpenSSL compute obf / rp / rq; these do the reference's host glue around them) */
int orc_paillier_encrypt_finish(const u64* n, int nw, const u64* m, int m_words, const u64* obf, u64* c,
                                size_t count) {
  const int W = 2 * nw;
  if (W > MAXW) return -1;
  u64 nsq[MAXW];
  bn_mul(nsq, n, nw, n, nw);
#pragma omp parallel for schedule(static)
  for (long i = 0; i < (long)count; ++i)
    enc_finish_one(n, nw, nsq, m + (size_t)i * m_words, m_words, obf + (size_t)i * W, c + (size_t)i * W);
  return 0;
}

/* bp = c mod p^2, bq = c mod q^2 (2*pw limbs each)   pri_key.cpp:128-129 */
int orc_paillier_decrypt_prepare(const u64* p, const u64* q, int pw, const u64* c, u64* bp, u64* bq,
                                 size_t count) {
  const int hw = 2 * pw, cw = 4 * pw;
  if (cw > MAXW) return -1;
  u64 psq[MAXW], qsq[MAXW];
  bn_mul(psq, p, pw, p, pw);
  bn_mul(qsq, q, pw, q, pw);
#pragma omp parallel for schedule(static)
  for (long i = 0; i < (long)count; ++i) {
    bn_mod(bp + (size_t)i * hw, c + (size_t)i * cw, cw, psq, hw);
    bn_mod(bq + (size_t)i * hw, c + (size_t)i * cw, cw, qsq, hw);
  }
  return 0;
}

int orc_paillier_decrypt_finish(const u64* p, const u64* q, int pw, const u64* hp, const u64* hq,
                                const u64* pinv, u64* rp, u64* rq, u64* m, size_t count) {
  const int hw = 2 * pw;
  if (4 * pw > MAXW) return -1;
#pragma omp parallel for schedule(static)
  for (long i = 0; i < (long)count; ++i)
    dec_finish_one(p, q, pw, hp, hq, pinv, rp + (size_t)i * hw, rq + (size_t)i * hw, m + (size_t)i * hw);
  return 0;
}

void orc_set_threads(int n) {
#ifdef _OPENMP
  extern void omp_set_num_threads(int);
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

int orc_max_threads(void) {
#ifdef _OPENMP
  extern int omp_get_max_threads(void);
  return omp_get_max_threads();
#else
  return 1;
#endif
}

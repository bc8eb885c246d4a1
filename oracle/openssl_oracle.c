/* openssl_oracle.c -- batched modexp through OpenSSL BN_mod_exp_mont.  TEST INFRASTRUCTURE ONLY.
 * The reference's own accelerator tests use OpenSSL as the independent modexp oracle
 * (module/heqat/test/test_bnModExp.cpp:57-60, 205-208); BASELINE.md lists it as CPU baseline B2.
 * Built only when libcrypto and its headers are present (oracle/Makefile probes).
 * Layout: little-endian uint64 limbs, [element][limb], as include/pgpu.h. */
#include <openssl/bn.h>
#include <stdint.h>
#include <stdlib.h>

int orc_openssl_modexp_batch(const uint64_t* base, size_t base_stride, const uint64_t* exp,
                             size_t exp_stride, int exp_words, const uint64_t* mod, int mod_words,
                             uint64_t* out, size_t count) {
  int bad = 0;
#pragma omp parallel
  {
    BN_CTX* ctx = BN_CTX_new();
    BIGNUM* m = BN_lebin2bn((const unsigned char*)mod, mod_words * 8, NULL);
    BN_MONT_CTX* mont = BN_MONT_CTX_new();
    BIGNUM *b = BN_new(), *e = BN_new(), *r = BN_new();
    int ok = ctx && m && mont && b && e && r && BN_MONT_CTX_set(mont, m, ctx);
    if (!ok) {
#pragma omp atomic write
      bad = 1;
    }
#pragma omp for schedule(dynamic, 1)
    for (long i = 0; i < (long)count; ++i) {
      if (!ok) continue;
      BN_lebin2bn((const unsigned char*)(base + (size_t)i * base_stride), (int)base_stride * 8, b);  /* whole row */
      BN_lebin2bn((const unsigned char*)(exp + (size_t)i * exp_stride), exp_words * 8, e);
      if (BN_cmp(b, m) >= 0) BN_mod(b, b, m, ctx);
      if (!BN_mod_exp_mont(r, b, e, m, ctx, mont) ||
          BN_bn2lebinpad(r, (unsigned char*)(out + (size_t)i * mod_words), mod_words * 8) < 0) {
#pragma omp atomic write
        bad = 1;
      }
    }
    BN_free(b); BN_free(e); BN_free(r); BN_free(m);
    BN_MONT_CTX_free(mont);
    BN_CTX_free(ctx);
  }
  return bad ? -1 : 0;
}

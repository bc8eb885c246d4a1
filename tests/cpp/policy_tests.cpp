// CPU unit test of the kernel-form policy (pailliercryptolib_amd/csrc/policy.cpp): sizes x busy lanes -> form, LDS claim,
// window.  Pure host logic -- built with g++ from policy.cpp alone, no device, no HIP call.  The expectations are the
// thresholds DESIGN.md sections 3-4 document for the MI355X (1024 SIMDs); the operations they steer are the reference's
// PrivateKey::decryptCRT (ipcl/pri_key.cpp:114-146), PublicKey::encrypt (pub_key.cpp:99-129) and the CipherText
// operators (ciphertext.cpp:135-162).
#include <cstdio>

#include "policy.hpp"

namespace pol = pgpu::policy;
static int g_failed = 0, g_checks = 0;
#define CHECK(cond)                                                        \
  do {                                                                     \
    ++g_checks;                                                            \
    if (!(cond)) {                                                         \
      ++g_failed;                                                          \
      std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond);          \
    }                                                                      \
  } while (0)

int main() {
  // ---- defaults: adaptive (4), the one-lane form by size / neighbours (1) ----
  CHECK(pol::seq_policy() == 4 && pol::ps_policy() == 1 && pol::adapt_claim_busy() == 3);
  CHECK(pol::seq_policy_by_size() == 1);
  // CRT decrypt, 2048-bit keys, split form (2,19): 32 ciphertexts per sequential-halves wavefront and side
  CHECK(!pol::seq_form_pays(2, 19, 8192, 0));      // 512 wavefronts: a lone caller keeps the paired kernel
  CHECK(pol::seq_form_pays(2, 19, 8192, 1));       // beside one busy lane: two half-chip launches
  CHECK(pol::seq_form_pays(2, 19, 16384, 0));      // 1024 wavefronts: by size
  CHECK(!pol::seq_form_pays(2, 19, 4096, 1));      // 256 x 2 < 1024
  CHECK(pol::seq_form_pays(2, 19, 4096, 3));
  CHECK(!pol::seq_form_pays(3, 19, 1 << 20, 3));   // not a compiled form
  CHECK(pol::seq_form_pays(4, 14, 8192, 0));       // 3072-bit keys: 16 per wavefront -> 1024 wavefronts
  // the one-lane product-scanning kernel: 64 exponentiations per wavefront
  CHECK(!pol::ps_form_pays(8192, 0) && !pol::ps_form_pays(8192, 1) && !pol::ps_form_pays(8192, 2));
  CHECK(pol::ps_form_pays(8192, 3));               // four lanes: a quarter of the chip each
  CHECK(pol::ps_form_pays(16384, 1) && !pol::ps_form_pays(16384, 0));   // (alone: one round of the sequential-halves form)
  CHECK(pol::ps_form_pays(32768, 0));              // covers the SIMDs alone
  // a lone launch runs in rounds of 32768 ciphertexts whatever their fill: one round beats the multi-lane forms above
  // 16384 ciphertexts of a 2048-bit key (K = 38), above 24576 of a 3072-bit key (K = 56); a form without measurements (any other K): full rounds only
  CHECK(pol::ps_form_pays(16385, 0) && pol::ps_form_pays(20000, 0, 38) && !pol::ps_form_pays(20000, 0, 56));
  CHECK(pol::ps_form_pays(24577, 0, 56) && pol::ps_form_pays(16385, 0, 19) && !pol::ps_form_pays(32767, 0, 20) && pol::ps_form_pays(32768, 0, 20));
  // ... and a mostly empty last round is cut off as a launch of its own
  CHECK(pol::ps_split_head(38, 32768) == 0 && pol::ps_split_head(38, 65536) == 0 && pol::ps_split_head(38, 20000) == 0);
  CHECK(pol::ps_split_head(38, 36000) == 32768 && pol::ps_split_head(38, 65536 + 16384) == 65536);
  CHECK(pol::ps_split_head(38, 32768 + 16385) == 0 && pol::ps_split_head(56, 32768 + 24576) == 32768);
  CHECK(pol::ps_split_head(56, 65536 + 24577) == 0 && pol::ps_split_head(19, 40000) == 32768);
  CHECK(!pol::ps_form_pays(1, 3) && !pol::ps_form_pays(8191 - 64, 3));   // 254 wavefronts x 4 < 1024
  // the decrypt exponent's window: 5 bits up to 2048-bit keys, 6 from 1280-bit exponents up (3072-bit keys: 318 against 338 products)
  CHECK(pol::pick_decrypt_window(512) == pol::pick_window(512) && pol::pick_decrypt_window(1024) == 5);
  CHECK(pol::pick_decrypt_window(1536) == 6 && pol::pick_decrypt_window(2048) == 6 && pol::pick_window(1536) == 5);
  // ... while the launch's table stays under 4 GiB: 65536 ciphertexts of a 3072-bit key (2 x 65536 entries of 448 B) keep w = 6, 1 M fall back
  CHECK(pol::pick_decrypt_window(1536, (size_t)2 * 65536 * 448) == 6 && pol::pick_decrypt_window(1536, (size_t)2 * (1 << 20) * 448) == 5);
  // whole-CU claims of part-chip launches
  CHECK(pol::adaptive_cu_claim(512, 1) == 84000u && pol::adaptive_cu_claim(256, 3) == 84000u);
  CHECK(pol::adaptive_cu_claim(512, 0) == 0u && pol::adaptive_cu_claim(1024, 1) == 0u && pol::adaptive_cu_claim(256, 4) == 0u);
  // DJN encrypt onto pair rows, form (4,18): 16 elements per sequential-halves wavefront
  CHECK(!pol::fb_encrypt_seq_pays(4, 18, 8192, 0));
  CHECK(pol::fb_encrypt_seq_pays(4, 18, 8192, 1) && pol::fb_encrypt_seq_pays(4, 18, 8192, 3));
  CHECK(!pol::fb_encrypt_seq_pays(4, 18, 8192, 2));        // beside two: the paired full-chip kernel
  CHECK(pol::fb_encrypt_seq_pays(4, 18, 16384, 0));
  CHECK(!pol::fb_encrypt_seq_pays(2, 19, 1 << 20, 0));     // 2-lane groups: measured behind the paired kernel
  CHECK(!pol::fb_encrypt_seq_pays(8, 14, 16384 - 8, 0));
  CHECK(pol::fb_encrypt_seq_pays(8, 14, 16384, 0));        // 8-lane groups: from two wavefronts per SIMD (8 per wavefront)
  // CT x PT and CT + CT
  CHECK(pol::modexp_seq_form_pays(4, 18, 1 << 20) && !pol::modexp_seq_form_pays(4, 18, 8192));
  CHECK(pol::pair_mul_seq_pays(4, 18, 1 << 20) && !pol::pair_mul_seq_pays(4, 18, 16383 - 16));
  // the latency form (one exponentiation per wavefront): small LONE launches only, and never while the one-lane form is forced / off
  CHECK(pol::wave_form_pays(1, 0) && pol::wave_form_pays(512, 0) && !pol::wave_form_pays(513, 0) && !pol::wave_form_pays(16, 1));
  CHECK(pol::modexp_wave_form_pays(1) && pol::modexp_wave_form_pays(1024) && !pol::modexp_wave_form_pays(1025));
  pol::set_ps_policy(2);
  CHECK(!pol::wave_form_pays(16, 0) && !pol::modexp_wave_form_pays(16));
  pol::set_ps_policy(0);
  CHECK(!pol::wave_form_pays(16, 0));
  pol::set_ps_policy(1);
  pol::set_wave_policy(2);
  CHECK(pol::wave_form_pays(100000, 3));
  pol::set_wave_policy(0);
  CHECK(!pol::wave_form_pays(16, 0));
  pol::set_wave_policy(1);
  // threads on round-robin lanes keep the lone caller's forms below this launch size (capi.cpp: busy_other_lanes)
  CHECK(pol::kRrAdaptMinCount == 4096 && pol::kRrAdaptMinCount <= 8192);   // (8192: the four-thread headline mode must stay adaptive)
  // windows
  CHECK(pol::pick_window(1024) == 5 && pol::pick_window(512) == 5 && pol::pick_window(33) == 3 && pol::pick_window(1) == 1);
  CHECK(pol::masked_decrypt_window() == 3);
  // ---- knobs ----
  pol::set_seq_policy(1);                                   // by size only: no adaptive forms, no claims
  CHECK(!pol::seq_form_pays(2, 19, 8192, 1) && pol::seq_form_pays(2, 19, 16384, 1) && pol::adaptive_cu_claim(512, 1) == 0u);
  CHECK(!pol::ps_form_pays(8192, 3) && pol::ps_form_pays(32768, 0));
  pol::set_seq_policy(3);                                   // round 3: half-chip launches take the form
  CHECK(pol::seq_form_pays(2, 19, 8192, 0) && !pol::seq_form_pays(2, 19, 4096, 0));
  pol::set_seq_policy(0);
  CHECK(!pol::seq_form_pays(2, 19, 1 << 20, 3) && !pol::modexp_seq_form_pays(4, 18, 1 << 20));
  pol::set_seq_policy(2);
  CHECK(pol::seq_form_pays(2, 19, 1, 0) && pol::fb_encrypt_seq_pays(2, 19, 1, 0));
  pol::set_seq_policy(4);
  pol::set_ps_policy(2);
  CHECK(pol::ps_form_pays(1, 0) && pol::ps_split_head(38, 36000) == 0);    // forced: one launch
  pol::set_ps_policy(0);
  CHECK(!pol::ps_form_pays(1 << 20, 3) && pol::ps_split_head(38, 36000) == 0);
  pol::set_ps_policy(1);
  CHECK(pol::set_rr_adapt(0) == 3 && pol::rr_adapt() == 0 && pol::set_rr_adapt(3) == 0);
  std::printf("%d checks, %d failed\n", g_checks, g_failed);
  return g_failed ? 1 : 0;
}

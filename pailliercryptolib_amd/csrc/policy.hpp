// pailliercryptolib_amd -- the kernel-FORM policy of the host runtime: which form of a split-form kernel a launch takes,
// from its size and from what the GPU's other batch lanes are doing.  Pure host logic (sizes x busy lanes -> form, LDS
// claim, window): no device, no HIP call -- tests/cpp/policy_tests.cpp checks it on the CPU.  The measurements behind
// every threshold are in DESIGN.md sections 3-4; the environment knobs are read once.
//
// The forms (include/pgpu.h: pgpu_kernel_form; reference: the operations of ipcl/pri_key.cpp:114-146, pub_key.cpp:51-129,
// ciphertext.cpp:135-162 all funnel into one exponentiation primitive, which is what takes these forms here):
//   paired        hensel.hpp      the two halves of a residue in neighbouring lanes -- most lanes per residue: a lone caller
//   sequential    hensel_seq.hpp  both halves in the same lanes -- half the wavefronts: launches that still cover the SIMDs,
//                                 alone or together with one busy neighbour lane (each launch then claims half the chip)
//   one-lane      hensel_ps.hpp   a whole exponentiation per lane (product scanning) -- a quarter of the wavefronts again:
//                                 large launches, or 8192-ciphertext launches beside three busy lanes (a quarter chip each)
#ifndef PAILLIERCRYPTOLIB_AMD_CSRC_POLICY_HPP_
#define PAILLIERCRYPTOLIB_AMD_CSRC_POLICY_HPP_

#include <stddef.h>

namespace pgpu {
namespace policy {

constexpr size_t kSimds = 256 * 4;   // MI355X: 256 CUs x 4 SIMDs; a launch of fewer wavefronts leaves SIMDs empty
// threads on round-robin lanes enter the adaptive policy only with launches of at least this many elements (capi.cpp:
// busy_other_lanes)
constexpr size_t kRrAdaptMinCount = 4096;

// ---- knobs: PGPU_SEQ_DECRYPT, PGPU_PS_DECRYPT, PGPU_RR_ADAPT, PGPU_FIXED_WINDOW, PGPU_WAVE_FORMS (environment at start-up; the
// setters are what pgpu_debug_set_* and the tests use) ----
int seq_policy();            // PGPU_SEQ_DECRYPT: 0 never, 1 by launch size, 2 always, 3 two-lane mode (r03), 4 adaptive (default)
void set_seq_policy(int p);
int ps_policy();             // PGPU_PS_DECRYPT (hensel_ps.hpp): 0 never, 1 by launch size / neighbour lanes (default), 2 always
void set_ps_policy(int p);
int adapt_claim_busy();      // up to how many busy neighbours a part-chip launch claims whole CUs (3; a constant since round 6)
int wave_policy();           // the latency form (hensel_wave.hpp): PGPU_WAVE_FORMS: 0 never, 1 small launches while SIMDs are spare (default), 2 always
void set_wave_policy(int p);
int rr_adapt();              // PGPU_RR_ADAPT: from how many active neighbours on threads on round-robin lanes adapt (3; 0 never)
int set_rr_adapt(int min_busy);   // returns the previous value

// ---- decisions ----
// the by-size part of the sequential-halves policy (modes 3 and 4 decide by size like mode 1, plus their extras)
int seq_policy_by_size();
// with `busy` other batch lanes at work, does a launch of `waves` wavefronts of a part-chip form fill its share of the chip?
bool seq_adaptive(size_t waves, int busy);
// LDS bytes a part-chip launch claims beyond its needs under the adaptive policy (more than half a CU's LDS: one workgroup
// per CU, so that the launches of neighbour lanes spread over the chip); 0: no claim
unsigned adaptive_cu_claim(size_t waves, int busy_lanes);
// CRT decrypt of `count` resident ciphertexts in split form (H, K): the sequential-halves kernel?
bool seq_form_pays(int H, int K, size_t count, int busy = 0);
// ... the one-lane product-scanning kernel of hensel_ps.hpp with K limbs per half (the key must have its constant set)?
// A lone launch runs in ROUNDS of kPsRound ciphertexts (one wavefront per SIMD: 2 x 32768 / 64 = 1024 wavefronts), and a
// round costs the same whether it is full or not -- so the form pays from ps_min_count(K) ciphertexts up, the size from
// which one round beats the multi-lane forms (whose rounds are 16384 / 8192 ciphertexts for 2048- / 3072-bit keys):
// 16385 for 1024- and 2048-bit keys, 24577 for 3072-bit keys ...
constexpr size_t kPsRound = 32768;
size_t ps_min_count(int K);
bool ps_form_pays(size_t count, int busy, int K = 38);
// ... and a lone launch of more than a round whose LAST round would be mostly empty is cut in two: the full rounds take
// this form, the rest a launch of its own in whatever form its size takes (ps_split_head: ciphertexts of the first
// launch; 0: one launch).  65536 + 4464 ciphertexts of a 2048-bit key: 42.7 ms in three rounds, 36 ms as 28 + 8.
size_t ps_split_head(int K, size_t count);
// ... the latency form (hensel_wave.hpp: one exponentiation per wavefront): a lone launch that leaves SIMDs empty even at two
// wavefronts per ciphertext -- up to kSimds / 2 = 512 ciphertexts; not while the one-lane form is forced or switched off
bool wave_form_pays(size_t count, int busy);
// ... and for CT x PT on resident rows (hensel_wave_n2.hpp: one wavefront per ELEMENT): launches of at most one wavefront per SIMD
bool modexp_wave_form_pays(size_t count);
// DJN encrypt onto pair rows / CT x PT / CT + CT of `count` elements in form (H, K): the sequential-halves kernels?
bool fb_encrypt_seq_pays(int H, int K, size_t count, int busy = 0);
bool modexp_seq_form_pays(int H, int K, size_t count);
bool pair_mul_seq_pays(int H, int K, size_t count);
// fixed window of a per-element / secret exponent of exp_bits bits: the w in 1..5 with the fewest products
int pick_window(int exp_bits);
// ... of the CRT-decrypt exponentiation (a secret exponent shared by the launch): w = 6 as well from 1280 bits up
// (entry_bytes: one table entry of every exponentiation of the launch; w = 6 only while the whole table stays under 4 GiB)
int pick_decrypt_window(int exp_bits, size_t entry_bytes = 0);
// ... under the masked table gather (every entry of the table read at every window product: small on purpose)
int masked_decrypt_window();

}  // namespace policy
}  // namespace pgpu

#endif  // PAILLIERCRYPTOLIB_AMD_CSRC_POLICY_HPP_

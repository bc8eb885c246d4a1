// pailliercryptolib_amd -- kernel-form policy (policy.hpp).  Split out of capi.cpp in round 5; no behaviour change.
#include "policy.hpp"

#include <algorithm>
#include <atomic>
#include <cstdlib>

#include "launch.hpp"

namespace pgpu {
namespace policy {

namespace {
int env_int(const char* name, int dflt, int lo, int hi) {
  const char* e = std::getenv(name);
  return e ? std::max(lo, std::min(hi, std::atoi(e))) : dflt;
}
// PGPU_SEQ_DECRYPT: 0 = never, 1 = for launches that put at least one wavefront of that form on every SIMD (16384
// ciphertexts under a 2048-bit key, 8192 under a 3072-bit key), 2 = whenever it applies.  hensel_seq.hpp: both halves of a
// residue in the same lanes, one after the other -- 10-13 % fewer instructions per exponentiation on half the lanes.
// 3: the round-3 opt-in mode for two batch lanes that are both kept busy -- a CRT decrypt also takes the form when it
// fills HALF the chip, its workgroups claiming more than half a CU's LDS so that the two lanes' launches spread over all
// CUs; every other operation as under 1.
// 4 (default since round 4): ADAPTIVE -- by launch size as under 1, and a launch that would leave SIMDs empty in this
// form takes it all the same when the GPU's OTHER batch lanes have work queued at launch time, i.e. when this launch
// will share the chip anyway: with b busy neighbours it needs waves * (1 + b) >= SIMDs, and claims the LDS that keeps
// a second workgroup off its CUs.  A lone caller -- nothing queued beside it -- keeps the full-chip paired kernels.  The
// probe is a hipStreamQuery per lane: what it costs when it is wrong is bounded by one launch (a neighbour that drains
// early leaves a half-chip launch to finish alone: 8.1 instead of 4.6 ms for 8192 ciphertexts).
std::atomic<int> g_seq_policy{env_int("PGPU_SEQ_DECRYPT", 4, 0, 4)};
// the one-lane product-scanning form (hensel_ps.hpp; 2048-bit keys).  64 exponentiations per wavefront: 8192 ciphertexts
// are 256 wavefronts -- a quarter of the SIMDs.  0 never; 1 (default) launches that put a wavefront on every SIMD that way
// (32768 ciphertexts), or -- adaptive, like the sequential-halves form -- do so together with the busy neighbour lanes:
// waves * (1 + busy) >= SIMDs (16384 ciphertexts beside one busy lane, 8192 beside three); 2 whenever compiled (tests)
std::atomic<int> g_ps_policy{env_int("PGPU_PS_DECRYPT", 1, 0, 2)};
// Two constants of the adaptive policy that were knobs (PGPU_ADAPT_ENC_SEQ / PGPU_ADAPT_CLAIM_BUSY) in rounds 4-5 and have
// sat at these values since: up to how many busy neighbours the DJN encrypt follows the decrypt into the sequential-halves
// form with a CU claim -- round 4 stopped at ONE (two lanes, each owning half the chip: 4.92 -> 4.87 ms per step); round 5:
// three, because beside one-lane decrypts a co-resident encrypt wavefront only gets the issue slots the older decrypt
// wavefront leaves (0.8 -> 10 ms), so the encrypt takes the lane's own quarter of the chip (capi.cpp: encrypt_on) -- and up
// to how many busy neighbours a part-chip launch claims whole CUs.
constexpr int kAdaptEncSeq = 3;
constexpr int kAdaptClaimBusy = 3;
// PGPU_RR_ADAPT = k > 0 (round 5; default 3): threads on ROUND-ROBIN lanes (synchronous callers of the ipcl:: API side by
// side) enter the adaptive policy as well, but only when at least k other lanes are active -- with all four lanes busy each
// caller's launches take a quarter of the chip (17 ms per encrypt + decrypt instead of 5.4 on the whole chip, four of them
// side by side: 5.3 against 5.9 ms per pair), which pays although every caller's quarter idles through its copies and host
// work; with one busy neighbour (half-chip forms) it does not (r04: 11.8 against 7.2 ms per pair).  0: lone-caller forms.
// PGPU_WAVE_FORMS: the latency forms (hensel_wave.hpp, hensel_wave_n2.hpp: one wavefront per exponentiation) -- 0 never,
// 1 (default) for small launches of a caller that has the chip to itself, 2 always where a kernel exists (tests)
std::atomic<int> g_wave_policy{env_int("PGPU_WAVE_FORMS", 1, 0, 2)};
std::atomic<int> g_rr_adapt{env_int("PGPU_RR_ADAPT", 3, 0, 100)};
}  // namespace

int seq_policy() { return g_seq_policy.load(); }
void set_seq_policy(int p) { g_seq_policy.store(p < 0 ? 0 : (p > 4 ? 4 : p)); }
int ps_policy() { return g_ps_policy.load(); }
void set_ps_policy(int p) { g_ps_policy.store(p < 0 ? 0 : (p > 2 ? 2 : p)); }
int adapt_claim_busy() { return kAdaptClaimBusy; }
int wave_policy() { return g_wave_policy.load(); }
void set_wave_policy(int p) { g_wave_policy.store(p < 0 ? 0 : (p > 2 ? 2 : p)); }
int rr_adapt() { return g_rr_adapt.load(); }
int set_rr_adapt(int min_busy) { return g_rr_adapt.exchange(min_busy < 0 ? 0 : min_busy); }

int seq_policy_by_size() {
  const int p = g_seq_policy.load();
  return (p == 3 || p == 4) ? 1 : p;
}
bool seq_adaptive(size_t waves, int busy) {
  return g_seq_policy.load() == 4 && busy >= 1 && waves * (size_t)(1 + busy) >= kSimds;
}
unsigned adaptive_cu_claim(size_t waves, int busy_lanes) {
  return (g_seq_policy.load() == 4 && waves < kSimds && busy_lanes >= 1 && busy_lanes <= kAdaptClaimBusy) ? 84000u : 0u;
}

bool fb_encrypt_seq_pays(int H, int K, size_t count, int busy) {
  if (!pgpu::hensel_fb_encrypt_seq_has(H, K)) return false;
  const size_t ipw = 64 / (size_t)H;
  const size_t waves = (count + ipw - 1) / ipw;
  const int pol = seq_policy_by_size();
  // (adaptive: beside busy neighbour lanes the form of half the wavefronts -- and a sixth fewer multiply-accumulates --
  // also for launches that would not fill the chip alone; 4-lane groups only, see below for the others)
  // beside ONE neighbour (half the chip each) and beside THREE (a quarter each: the lane's own CUs between its one-lane
  // decrypts); beside two the lanes' sequential-halves decrypts hold every CU and the paired full-chip encrypt squeezes
  // in better than a launch that waits for whole CUs (k = 3 of the lane sweep: 4.9-5.1 against 5.2 ms per step)
  if (H == 4 && busy != 2 && busy <= kAdaptEncSeq && seq_adaptive(waves, busy)) return true;
  // (2-lane groups, 1024-bit keys: measured equal or behind the paired kernel at 65536 elements -- 1.04 against 1.02 ms,
  // 0.96 against 0.92 ms: 19 limbs per lane and the LDS staging leave no register room -- so only when forced)
  if (H == 2 && pol != 2) return false;
  // (8-lane groups pay two DPP moves per row broadcast: alone on a SIMD the form is 2 % behind the paired kernel --
  // 3072-bit keys, 8192 elements: 3.60 against 3.52 ms -- and 7 % ahead with two wavefronts per SIMD: 6.2 against 6.7 ms)
  return pol == 2 || (pol == 1 && waves >= (H >= 8 ? 2 : 1) * kSimds);
}
bool modexp_seq_form_pays(int H, int K, size_t count) {
  if (!pgpu::hensel_modexp_seq_has(H, K)) return false;
  const size_t ipw = 64 / (size_t)H;
  const size_t waves = (count + ipw - 1) / ipw;
  const int pol = seq_policy_by_size();
  if (H == 2 && pol != 2) return false;   // (see fb_encrypt_seq_pays)
  return pol == 2 || (pol == 1 && waves >= (H >= 8 ? 2 : 1) * kSimds);   // (8-lane groups: see fb_encrypt_seq_pays)
}
bool seq_form_pays(int H, int K, size_t count, int busy) {
  if (!pgpu::hensel_seq_has(H, K)) return false;
  const size_t ipw = 64 / (size_t)H;
  const size_t waves = 2 * ((count + ipw - 1) / ipw);
  const int pol = g_seq_policy.load();
  return pol == 2 || (pol == 3 && 2 * waves >= kSimds) || ((pol == 1 || pol == 4) && waves >= kSimds) || seq_adaptive(waves, busy);
}
// Measured on lone launches (profiles/r05_decrypt_partial_rounds.txt).  2048-bit keys (K = 38): a round 14.2 ms against
// 8.1 ms per 16384 ciphertexts of the sequential-halves form and 4.6 ms for up to 8192 in the paired one -- one round beats
// them above 16384 ciphertexts, and a tail of up to 16384 is cheaper as a launch of its own.  3072-bit keys (K = 56): a
// round 47.6 ms against 13.3-15.2 ms per 8192 of the (4,14) sequential-halves form: above 24576.  1024-bit keys (K = 19):
// a round 2.2-2.3 ms against 1.8 ms for 16384 ciphertexts and 3.0 ms for 24576 in the multi-lane forms: above 16384.
size_t ps_min_count(int K) {
  return K == 56 ? 24576 + 1 : (K == 38 || K == 19) ? 16384 + 1 : kPsRound;
}
static size_t ps_split_tail(int K) { return K == 56 ? 24576 : 16384; }
bool wave_form_pays(size_t count, int busy) {
  const int pol = g_wave_policy.load();
  if (pol == 2) return true;
  return pol == 1 && g_ps_policy.load() == 1 && busy == 0 && 2 * count <= kSimds;
}
bool modexp_wave_form_pays(size_t count) {
  const int pol = g_wave_policy.load();
  if (pol == 2) return true;
  return pol == 1 && g_ps_policy.load() == 1 && count <= kSimds;
}
bool ps_form_pays(size_t count, int busy, int K) {
  const size_t waves = 2 * ((count + 63) / 64);
  const int pol = g_ps_policy.load();
  return pol == 2 || (pol == 1 && (count >= ps_min_count(K) || seq_adaptive(waves, busy)));
}
size_t ps_split_head(int K, size_t count) {
  if (g_ps_policy.load() != 1 || count <= kPsRound) return 0;
  const size_t tail = count % kPsRound;
  return (tail != 0 && tail <= ps_split_tail(K)) ? count - tail : 0;
}
bool pair_mul_seq_pays(int H, int K, size_t count) {
  if (!pgpu::pair_mul_seq_has(H, K)) return false;
  const size_t ipw = 64 / (size_t)H;
  const size_t waves = (count + ipw - 1) / ipw;
  const int pol = seq_policy_by_size();
  return pol == 2 || (pol == 1 && waves >= (H >= 8 ? 2 : 1) * kSimds);   // (8-lane groups: see fb_encrypt_seq_pays)
}

int pick_window(int exp_bits) {
  // PGPU_FIXED_WINDOW=w: A/B measurements of the window width (what an LDS-resident table, which
  // holds 8 entries per exponentiation at most, would have to beat)
  static const int forced = env_int("PGPU_FIXED_WINDOW", 0, 1, 6);
  if (forced) return forced;
  int best = 1;
  long best_cost = 1L << 60;
  for (int w = 1; w <= 5; ++w) {
    long cost = ((1L << w) - 2) + (exp_bits + w - 1) / w;
    if (cost < best_cost) { best_cost = cost; best = w; }
  }
  return best;
}
// Window of the CRT-decrypt exponentiation (one secret exponent per side, shared by the launch): exponents of 1280 bits and
// more -- 3072-bit keys up -- also try w = 6 (1536 bits: 62 + 256 products instead of 30 + 308; 65536 ciphertexts 95.9 ->
// 94.7 ms).  Not for shorter ones: 1024 bits would tie on products and double the tables of the four-lane headline; and not
// in pick_window, whose per-element tables of a 1 M-element CT x PT would double with it.
// The window table of a launch is `entry_bytes` (all exponentiations of the launch, one entry each) x 2^w: w = 6 only while
// that stays under kDecryptTableCap -- a 65536-ciphertext launch of a 3072-bit key is 3.8 GB at w = 6 (config 4: kept), a
// 1 M-ciphertext one would be 60 GB for a gain of 1.2 %; entry_bytes = 0: no cap (the by-exponent rule alone).
constexpr size_t kDecryptTableCap = (size_t)4 << 30;
int pick_decrypt_window(int exp_bits, size_t entry_bytes) {
  const int w5 = pick_window(exp_bits);
  static const int forced = env_int("PGPU_FIXED_WINDOW", 0, 1, 6);
  if (forced || exp_bits < 1280) return w5;
  if (entry_bytes > (kDecryptTableCap >> 6)) return w5;
  const long c5 = ((1L << w5) - 2) + (exp_bits + w5 - 1) / w5, c6 = ((1L << 6) - 2) + (exp_bits + 5) / 6;
  return c6 < c5 ? 6 : w5;
}
// Window of the CRT-decrypt exponentiation under the masked table gather (round 5): every one of the 2^w entries of an
// exponentiation's table is read at every window product, so the table is what the launch streams -- 9.7 KB per
// exponentiation at w = 5, 32 entries x 235 products; four batches in flight (636 MB of tables) fall out of the 256 MB
// Infinity Cache and the one-lane decrypt went from 14.5 to 41 ms.  w = 3: 8 entries x 348 products, a third of the bytes,
// 11 % more products: 26 ms (lone paired launch: 6.6 -> 6.0 ms against 4.6 indexed).
int masked_decrypt_window() { return 3; }

}  // namespace policy
}  // namespace pgpu

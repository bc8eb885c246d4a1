// pailliercryptolib_amd -- implementation of the C-ABI declared in include/pgpu.h.
// Host side: Montgomery-constant precomputation (host BigNumber), geometry selection, kernel launches,
// sharding over the device pool (runtime.hpp).  No CPU fallback: everything computes on the GPU.
#include "pgpu.h"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <list>
#include <map>
#include <set>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "ipcl/bignum.h"
#include "kargs.hpp"
#include "launch.hpp"
#include "policy.hpp"
#include "runtime.hpp"

namespace rt = pgpu::rt;
using rt::fail;
// the kernel-form policy (policy.cpp: pure host logic, unit-tested on the CPU)
namespace policy = pgpu::policy;
using pgpu::policy::adaptive_cu_claim;
using pgpu::policy::fb_encrypt_seq_pays;
using pgpu::policy::kSimds;
using pgpu::policy::masked_decrypt_window;
using pgpu::policy::modexp_seq_form_pays;
using pgpu::policy::pair_mul_seq_pays;
using pgpu::policy::pick_window;
using pgpu::policy::seq_adaptive;
using pgpu::policy::seq_form_pays;
using pgpu::policy::seq_policy_by_size;

namespace {

// forget a host buffer that held key material (p-1, q-1, hp, hq, private Montgomery constants): volatile stores,
// which the compiler may not elide the way it may a std::fill on a buffer that is about to die
void secure_wipe(void* p, size_t bytes) {
  volatile unsigned char* w = static_cast<volatile unsigned char*>(p);
  for (size_t i = 0; i < bytes; ++i) w[i] = 0;
}

// ---------- geometry table ----------
struct GeoInfo {
  int G, K;
  int L() const { return G * K; }
  int rbits() const { return pgpu::kLimbBits * G * K; }
  int w64() const { return (rbits() + 63) / 64; }
  int ipw() const { return pgpu::kWave / G; }
};
const GeoInfo kGeos[] = {{2, 9}, {4, 9}, {4, 10}, {4, 14}, {8, 9}, {8, 14}, {16, 9}, {16, 14}, {16, 18}};

// (4,10): the 1024-bit class with room for unit quotient digits (R >= 256 * Nhat needs 37 bits above the
// modulus; (4,9) offers 20).  Measured (tools/probe_geo410.py, profiles/r02_geo410.txt): the 23 % more MACs of
// L = 40 cost more than the per-row n0' multiply they remove (65536 x 512-bit exponents: 6.7 ms vs 5.8 ms), so
// it is OFF by default; PGPU_GEO_410=1 selects it.
bool geo410_enabled() {
  static const bool on = [] { const char* e = std::getenv("PGPU_GEO_410"); return e && std::atoi(e) != 0; }();
  return on;
}

bool unit_fits(const GeoInfo& geo, int mod_bits) { return geo.rbits() >= mod_bits + pgpu::kLimbBits + 8; }

// smallest geometry with R = 2^(29*G*K) >= 2^(64*in_words) (any input row fits) and R >= 256*N.
// unit_q: the caller wants unit quotient digits; a (4,9)-class modulus without the headroom moves to (4,10)
const GeoInfo* pick_geo(int in_words, int mod_bits, bool unit_q = false) {
  const GeoInfo* fit = nullptr;
  const GeoInfo* g410 = nullptr;
  for (const GeoInfo& g : kGeos) {
    if (g.G == 4 && g.K == 10) { g410 = &g; continue; }
    if (!fit && g.rbits() >= 64 * in_words && g.rbits() >= mod_bits + 8) fit = &g;
  }
  if (fit && unit_q && !unit_fits(*fit, mod_bits) && fit->G == 4 && fit->K == 9 && geo410_enabled() && g410 &&
      g410->rbits() >= 64 * in_words && unit_fits(*g410, mod_bits))
    return g410;
  return fit;
}

// v as L limbs of lb bits (the kernels' 29; hensel_ps.hpp: 28 for the 2048-bit key class)
void to_limbs(const BigNumber& v, int L, uint32_t* out, int lb) {
  const BigNumber::Limbs& w = v.limbs64();
  const uint32_t mask = (1u << lb) - 1;
  for (int i = 0; i < L; ++i) {
    int bit = i * lb;
    size_t word = (size_t)bit >> 6;
    int sh = bit & 63;
    uint64_t x = word < w.size() ? w[word] >> sh : 0;
    if (sh > 64 - lb && word + 1 < w.size()) x |= w[word + 1] << (64 - sh);
    out[i] = (uint32_t)x & mask;
  }
}
void to_limbs29(const BigNumber& v, int L, uint32_t* out) { to_limbs(v, L, out, pgpu::kLimbBits); }

BigNumber pow2(int bits) {
  std::vector<uint64_t> w((size_t)bits / 64 + 1, 0);
  w[(size_t)bits / 64] = 1ull << (bits % 64);
  return BigNumber::fromLimbs64(w.data(), w.size());
}

// ---------- modulus context ----------
// Host description of a pgpu::ModCtxDev: the constants live in ONE position-independent image that is
// replicated to every pool device (runtime.hpp: Replicated); view() turns offsets into the pointers of a device.
enum CtxSlot { S_N, S_R2, S_ONE, S_R2S, S_FC, S_NR, S_NHAT, S_ONE_N, S_NR2, S_R2M, S_R2SM, S_COUNT };
enum ViewFlags : int {
  VF_NONE = 0,
  VF_BASE_MONT = 1,   // the (wide) base arrives in the Montgomery domain of another modulus: r2/r2s := r2m/r2sm
  VF_GM_MONT = 2,     // Paillier g^m in Montgomery form: nr := n*R^2, gadd := R mod N  (Montgomery-form result)
  VF_IN_MONT = 4,     // base already in THIS context's Montgomery form: r2 := R (the first product is an identity)
  VF_OUT_MONT = 8     // leave the result in Montgomery form: fc := R mod N (use with FM_CTX_CONST)
};

struct ModCtx {
  GeoInfo geo{};
  int mod_words = 0;
  BigNumber N;
  rt::Replicated blob;
  bool has[S_COUNT] = {};
  size_t off64 = 0;
  uint32_t n0inv = 0;
  bool unit = false;
  pgpu::ModCtxDev view(int dev, int flags = VF_NONE) const {
    const uint32_t* d32 = (const uint32_t*)blob.d[(size_t)dev];
    const int L = geo.L();
    auto at = [&](int slot) -> const uint32_t* { return has[slot] ? d32 + (size_t)slot * L : nullptr; };
    pgpu::ModCtxDev v{};
    v.n = at(S_N);
    v.r2 = at((flags & VF_BASE_MONT) ? S_R2M : (flags & VF_IN_MONT) ? S_ONE : S_R2);
    v.one = at(S_ONE);
    v.r2s = at((flags & VF_BASE_MONT) ? S_R2SM : S_R2S);
    v.fc = at((flags & VF_OUT_MONT) ? S_ONE_N : S_FC);
    v.nr = at((flags & VF_GM_MONT) ? S_NR2 : S_NR);
    v.nhat = at(S_NHAT);
    v.gadd = (flags & VF_GM_MONT) ? at(S_ONE_N) : nullptr;
    v.n64 = (const uint64_t*)((const char*)blob.d[(size_t)dev] + off64);
    v.n0inv = n0inv;
    v.mod_words = mod_words;
    return v;
  }
};

// extras: optional constants of the context image
struct CtxExtras {
  bool unit_q = false;        // scale the loop modulus to Nhat = N*k == -1 mod 2^29 (if R has room)
  int force_unit = -1;        // -1: decide from the headroom; 0 / 1: both contexts of a pair must agree
  bool want_r2s = false;      // R^2 * 2^(64*mod_words) mod N
  const BigNumber* fc = nullptr;    // plain final multiplier
  const BigNumber* nr_n = nullptr;  // n  ->  n*R mod N and n*R^2 mod N
  int mont_src_rbits = 0;     // > 0: bases arrive as x * 2^mont_src_rbits mod (another modulus): r2m / r2sm
  bool secret = false;        // zero the device copies before they are freed
};

int build_modctx(const BigNumber& N, int mod_words, const GeoInfo& geo, const CtxExtras& ex,
                 std::shared_ptr<ModCtx>* out) {
  if (N.isZero() || N.isNegative() || N == BigNumber::One())
    return fail(PGPU_ERR_INVALID_PARAM, "modulus must be > 1");
  if (!N.IsOdd()) return fail(PGPU_ERR_EVEN_MODULUS, "modulus must be odd");
  if (geo.rbits() < N.BitSize() + 8)
    return fail(PGPU_ERR_UNSUPPORTED, "geometry too small for modulus");
  const int L = geo.L(), W64 = geo.w64();
  BigNumber R = pow2(geo.rbits());
  uint32_t n0 = (uint32_t)(N.limbs64()[0] & pgpu::kLimbMask);
  uint32_t inv = n0;  // Newton iteration for n0^-1 mod 2^32 (n0 odd: correct to 3 bits)
  for (int i = 0; i < 5; ++i) inv *= 2u - n0 * inv;
  uint32_t n0inv = (0u - inv) & pgpu::kLimbMask;
  // Unit quotient digits: Nhat = N * k with k = -N^-1 mod 2^29 is == -1 mod 2^29, so its
  // Montgomery constant is 1 and q = low limb.  All loop constants are then taken modulo Nhat
  // (a multiple of N: lazy values stay correct modulo N).  Needs R >= 256 * Nhat.
  bool unit = ex.unit_q && unit_fits(geo, N.BitSize());
  if (ex.force_unit >= 0) unit = ex.force_unit != 0 && unit_fits(geo, N.BitSize());
  const BigNumber M = unit ? N * BigNumber((Ipp32u)n0inv) : N;   // loop modulus
  BigNumber Rm = R % M;
  BigNumber R2 = (Rm * Rm) % M;

  auto ctx = std::make_shared<ModCtx>();
  std::vector<uint32_t> h((size_t)S_COUNT * L, 0);
  auto put = [&](int slot, const BigNumber& v) {
    to_limbs29(v, L, h.data() + (size_t)slot * L);
    ctx->has[slot] = true;
  };
  put(S_N, N);
  put(S_R2, R2);
  put(S_ONE, Rm);
  put(S_ONE_N, R % N);
  const BigNumber shift = pow2(64 * mod_words) % M;
  if (ex.want_r2s) put(S_R2S, (R2 * shift) % M);
  if (ex.fc) put(S_FC, *ex.fc % N);
  if (ex.nr_n) {
    const BigNumber Rn = R % N;
    put(S_NR, (*ex.nr_n * Rn) % N);              // used under the TRUE modulus
    put(S_NR2, (((*ex.nr_n * Rn) % N) * Rn) % N);
  }
  if (unit) put(S_NHAT, M);
  if (ex.mont_src_rbits > 0) {
    // x arrives as x*Rs mod (a multiple of N); to-Montgomery constants that cancel Rs: R^2 * Rs^-1
    const BigNumber Rs_inv = M.InverseMul(pow2(ex.mont_src_rbits) % M);
    const BigNumber r2m = (R2 * Rs_inv) % M;
    put(S_R2M, r2m);
    put(S_R2SM, (r2m * shift) % M);
  }
  std::vector<uint64_t> n64((size_t)W64 + 1, 0);
  N.toLimbs64(n64.data(), n64.size());

  ctx->geo = geo;
  ctx->mod_words = mod_words;
  ctx->N = N;
  ctx->n0inv = n0inv;
  ctx->unit = unit;
  const size_t bytes32 = h.size() * sizeof(uint32_t);
  ctx->off64 = (bytes32 + 15) & ~(size_t)15;
  std::vector<uint8_t> host(ctx->off64 + n64.size() * 8, 0);
  std::memcpy(host.data(), h.data(), bytes32);
  std::memcpy(host.data() + ctx->off64, n64.data(), n64.size() * 8);
  RC_TRY(ctx->blob.upload(host.data(), host.size(), ex.secret));
  if (ex.secret) {
    secure_wipe(host.data(), host.size());
    secure_wipe(h.data(), h.size() * sizeof(uint32_t));
  }
  *out = ctx;
  return PGPU_OK;
}

// Caches of per-modulus device constants (the key-less seams).  Least-recently-used entries are evicted ONE at a
// time; an evicted entry is not destroyed on the spot -- a `_dev` call only enqueues its kernel, so the device image
// may still be read -- but parked: parked entries are destroyed by pgpu_synchronize / pgpu_shutdown (after the
// devices have drained), or, when more than kParkMax have piled up, oldest first by the evicting call itself
// (hipFree then waits for the device: correct, merely slow, and only for workloads that cycle through more than
// kCacheCap + kParkMax moduli without ever synchronising).
constexpr size_t kCacheCap = 64, kParkMax = 64;
std::mutex g_park_mu;
std::deque<std::shared_ptr<void>> g_parked;
void park(std::shared_ptr<void> victim) {
  if (!victim) return;
  std::shared_ptr<void> overflow;
  {
    std::lock_guard<std::mutex> lk(g_park_mu);
    g_parked.push_back(std::move(victim));
    if (g_parked.size() > kParkMax) {
      overflow = std::move(g_parked.front());
      g_parked.pop_front();
    }
  }
  // `overflow` dies here, outside the lock (its hipFree synchronises the device)
}
void drain_parked() {
  std::deque<std::shared_ptr<void>> dead;
  {
    std::lock_guard<std::mutex> lk(g_park_mu);
    dead.swap(g_parked);
  }
}
template <class V>
struct LruCache {
  struct Slot {
    std::shared_ptr<V> value;
    bool known = false;   // a null value may be a cached verdict ("not a square")
    uint64_t tick = 0;
  };
  std::map<std::vector<uint64_t>, Slot> slots;
  uint64_t clock = 0;
  bool find(const std::vector<uint64_t>& key, std::shared_ptr<V>* out) {
    auto it = slots.find(key);
    if (it == slots.end()) return false;
    it->second.tick = ++clock;
    *out = it->second.value;
    return true;
  }
  void insert(const std::vector<uint64_t>& key, std::shared_ptr<V> value) {
    if (slots.size() >= kCacheCap) {
      auto victim = slots.begin();
      for (auto it = slots.begin(); it != slots.end(); ++it)
        if (it->second.tick < victim->second.tick) victim = it;
      park(std::move(victim->second.value));
      slots.erase(victim);
    }
    Slot s;
    s.value = std::move(value);
    s.known = true;
    s.tick = ++clock;
    slots[key] = std::move(s);
  }
  void clear() { slots.clear(); }
};

std::mutex g_ctx_mu;
LruCache<ModCtx> g_ctx_cache;

GeoInfo latency_geo(const GeoInfo& geo);
// cached plain context for the generic seam: unit_q = true for pgpu_modexp (loop modulo Nhat),
// false for pgpu_modmul (true modulus throughout).  latency: build it for the 16-lane latency geometry of the
// modulus' class (same context when L is unchanged or the class has none)
int get_modctx(const uint64_t* mod, int mod_words, bool unit_q, std::shared_ptr<ModCtx>* out,
               bool latency = false) {
  if (!mod || mod_words <= 0) return fail(PGPU_ERR_INVALID_PARAM, "modulus is null/empty");
  if (!(mod[0] & 1)) return fail(PGPU_ERR_EVEN_MODULUS, "modulus must be odd");
  std::vector<uint64_t> key(mod, mod + mod_words);
  key.push_back((unit_q ? 1 : 0) | (latency ? 2 : 0));
  std::lock_guard<std::mutex> lk(g_ctx_mu);
  if (g_ctx_cache.find(key, out)) return PGPU_OK;
  BigNumber N = BigNumber::fromLimbs64(mod, (size_t)mod_words);
  if (N.isZero() || N == BigNumber::One())
    return fail(PGPU_ERR_INVALID_PARAM, "modulus must be > 1");
  const GeoInfo* geo = pick_geo(mod_words, N.BitSize(), unit_q);
  if (!geo) return fail(PGPU_ERR_UNSUPPORTED, "modulus wider than the compiled kernel geometries");
  CtxExtras ex;
  ex.unit_q = unit_q;
  RC_TRY(build_modctx(N, mod_words, latency ? latency_geo(*geo) : *geo, ex, out));
  g_ctx_cache.insert(key, *out);
  return PGPU_OK;
}

// ---------- exponent scanning ----------
// PGPU_SLIDING=0 keeps the fixed-window digit scan for host-known PUBLIC exponents too (A/B measurements)
bool sliding_enabled() {
  static const bool on = [] { const char* e = std::getenv("PGPU_SLIDING"); return !e || std::atoi(e) != 0; }();
  return on;
}
// private-key exponents (p-1, q-1): include/pgpu.h, SIDE CHANNELS
std::atomic<int> g_secret_policy{-1};
int secret_policy() {
  int p = g_secret_policy.load();
  if (p < 0) {
    const char* e = std::getenv("PGPU_SECRET_EXP");
    p = (e && (std::strcmp(e, "sliding") == 0 || std::strcmp(e, "1") == 0)) ? PGPU_EXP_SLIDING : PGPU_EXP_FIXED_WINDOW;
    g_secret_policy.store(p);
  }
  return p;
}


// Sliding-window schedule of an exponent the host knows: odd powers base^(2i+1), i < 2^(w-1);
// step = (nsq << 6) | (idx + 1) -- nsq squarings then * base^(2 idx + 1) (idx + 1 == 0: squarings only);
// step 0 has nsq == 0 and loads its entry (kargs.hpp ModexpArgs).
struct ExpSchedule {
  rt::Replicated dev;
  int len = 0;
  int w = 0;
};
std::vector<uint16_t> sliding_schedule(const BigNumber& e, int w) {
  std::vector<uint16_t> st;
  int i = e.BitSize() - 1, pending = 0;
  auto emit = [&](int nsq, int idx_plus1) {
    while (nsq > 1023) { st.push_back((uint16_t)(1023 << 6)); nsq -= 1023; }
    st.push_back((uint16_t)((nsq << 6) | idx_plus1));
  };
  while (i >= 0) {
    if (!e.TestBit(i)) { ++pending; --i; continue; }
    int j = std::max(i - w + 1, 0);
    while (!e.TestBit(j)) ++j;              // window [i..j], odd value
    int v = 0;
    for (int b = i; b >= j; --b) v = (v << 1) | (e.TestBit(b) ? 1 : 0);
    emit(st.empty() ? 0 : pending + (i - j + 1), (v - 1) / 2 + 1);
    pending = 0;
    i = j - 1;
  }
  if (pending) emit(pending, 0);
  return st;
}
// window of the cheapest sliding schedule for a random exponent of this length (w <= 6: 32 odd powers,
// the same table footprint as the fixed 5-bit window)
int pick_sliding_window(int exp_bits) {
  int best = 1;
  long best_cost = 1L << 60;
  for (int w = 1; w <= 6; ++w) {
    long cost = (1L << (w - 1)) + exp_bits / (w + 1);
    if (cost < best_cost) { best_cost = cost; best = w; }
  }
  return best;
}
int make_schedule(const BigNumber& e, int w, ExpSchedule* out, bool secret) {
  std::vector<uint16_t> st = sliding_schedule(e, w);
  out->len = (int)st.size();
  out->w = w;
  if (st.empty()) st.push_back(0);
  return out->dev.upload(st.data(), st.size() * sizeof(uint16_t), secret);
}

// window width of the fixed-base table for the DJN obfuscator; PGPU_FB_WINDOW=0 selects the
// generic (per-instance table, square-and-multiply) kernel instead.
constexpr int kFbMaxW = 14;        // (the kernels take any width; the table budgets narrow it -- fb_fit_window)
// 13 since round 4: 79 products for a 1024-bit r instead of 86 (encrypt launch of the bench 0.835 -> 0.775 ms, step +1 %),
// 373 MB of table per 2048-bit key and GPU instead of 203 MB, built in 73 ms instead of 40
constexpr int kFbDefaultW = 13;
constexpr int kFbMaskedMaxW = 5;   // tables of windows up to this width belong to the masked product (kept beside the indexed table)
std::atomic<int> g_fb_window{-1};
std::atomic<bool> g_fb_window_explicit{false};   // set through the environment or pgpu_set_fixed_base_window
int fixed_base_window() {
  if (g_fb_window.load() < 0) {
    const char* e = std::getenv("PGPU_FB_WINDOW");
    int v = e ? std::atoi(e) : kFbDefaultW;
    g_fb_window.store((v < 0 || v > kFbMaxW) ? kFbDefaultW : v);
    if (e) g_fb_window_explicit.store(true);
  }
  return g_fb_window.load();
}
// window of the MASKED fixed-base product (every entry of a window is read and selected): small on purpose
int masked_fb_window() {
  static const int w = [] {
    const char* e = std::getenv("PGPU_FB_MASKED_WINDOW");
    return e ? std::max(1, std::min(kFbMaskedMaxW, std::atoi(e))) : 4;
  }();
  return w;
}
// A key that has encrypted little so far starts with an 8-bit window (table 16x smaller, built in ~2 ms
// instead of ~37 ms) and moves to the configured one once this many elements have gone through it.
constexpr size_t kFbGrowAfter = 4096;

// ---------- live launch timing ----------
std::atomic<bool> g_timing{false};
struct TimerScope {
  rt::Device& d;
  hipStream_t s;
  bool on;
  rt::TimedLaunch t{};
  TimerScope(rt::Device& dev, hipStream_t st, int kind, int form = 0) : d(dev), s(st), on(g_timing.load()) {
    if (!on) return;
    t.kind = kind;
    t.form = form;
    t.stream = st;
    t.e0 = d.pool_event();
    t.e1 = d.pool_event();
    (void)hipEventRecord(t.e0, s);
  }
  void set_form(int form) { t.form = form; }
  void stop() {
    if (!on) return;
    (void)hipEventRecord(t.e1, s);
    std::lock_guard<std::mutex> lk(d.mu);
    if (d.timed.size() < 65536) d.timed.push_back(t);
  }
};

// ---------- launch geometry ----------
// wavefronts of a modexp launch: with parity waves every wave takes one parity of the instances
// (kargs.hpp: ModexpArgs::nctx), i.e. 2 * ceil(elements / IPW)
size_t modexp_waves(size_t count, int parity_waves, int ipw) {
  return parity_waves ? 2 * ((count / 2 + ipw - 1) / ipw) : (count + ipw - 1) / ipw;
}
unsigned blocks_for(size_t count, const GeoInfo& g) {
  return (unsigned)((count + (size_t)g.ipw() * pgpu::kWavesPerWG - 1) / ((size_t)g.ipw() * pgpu::kWavesPerWG));
}

// Small batches do not fill the chip: a launch of fewer wavefronts than SIMDs runs as long as ONE wavefront's
// serial chain of ~1200 multiplications.  Spreading an element over 16 lanes instead of 8 shortens every
// multiplication (2048-bit class: (16,5), 1190 instructions instead of 1590 for (8,9); 3072-bit class: (16,7)
// for (8,14), same L and therefore the same context).  Used while the 16-lane split still fits one wavefront
// per SIMD; (16,5) has L = 80, so it needs its own Montgomery context.
GeoInfo latency_geo(const GeoInfo& geo) {
  static const bool allow = [] { const char* e = std::getenv("PGPU_LATENCY_GEO"); return !e || std::atoi(e) != 0; }();
  if (!allow) return geo;
  if (geo.G == 8 && geo.K == 9) return GeoInfo{16, 5};
  if (geo.G == 8 && geo.K == 14) return GeoInfo{16, 7};
  return geo;
}
bool use_latency_geo(const GeoInfo& lat, const GeoInfo& geo, size_t instances) {
  return lat.G != geo.G && (instances + lat.ipw() - 1) / lat.ipw() <= kSimds;
}

// A context built for (G, 9) also serves the "wide" split (G/2, 18): same L, same R, same limb
// arrays, half the lanes per exponentiation and twice the limbs per lane -- the per-row support
// instructions are amortised over twice as many MACs.  It pays as soon as the wide split still puts one
// wavefront on every SIMD: measured on the bench's CRT-decrypt launch, (4,18) at 1 wave/SIMD 7.2 ms vs
// (8,9) at 2 waves/SIMD 8.1 ms.
constexpr size_t kMinWavesForWide = 256 * 4;
GeoInfo launch_geo(const GeoInfo& geo, size_t count) {
  static const bool allow = [] { const char* e = std::getenv("PGPU_WIDE"); return !e || std::atoi(e) != 0; }();
  if (!allow || geo.K != 9 || geo.G < 4) return geo;
  GeoInfo wide{geo.G / 2, 18};
  size_t waves = (count + wide.ipw() - 1) / wide.ipw();
  static const size_t min_waves = [] {
    const char* e = std::getenv("PGPU_WIDE_MIN_WAVES");     // tuning knob (tools/quick_bench.py)
    return e && std::atol(e) > 0 ? (size_t)std::atol(e) : kMinWavesForWide;
  }();
  return waves >= min_waves ? wide : geo;
}

// A launch of at most one wavefront per SIMD takes the kernel form whose multiplier rows are broadcast from registers
// (kernels.hpp: REGROWS): nobody covers a lone wavefront's LDS round trips.  That form holds one wavefront per
// SIMD (it parks a few values in AGPRs), so anything larger runs the LDS form at two per SIMD.
// PGPU_REGROWS=0 / 1 forces the LDS / register form (A/B measurements).
std::atomic<int> g_row_source{-2};   // -1 auto, 0 LDS, 1 registers (-2: not read from the environment yet)
bool use_regrows(const GeoInfo& geo, size_t waves) {
  int mode = g_row_source.load();
  if (mode == -2) {
    const char* e = std::getenv("PGPU_REGROWS");
    mode = e ? (std::atoi(e) != 0 ? 1 : 0) : -1;
    g_row_source.store(mode);
  }
  if (!pgpu::modexp_has_regrows(geo.G, geo.K)) return false;
  if (mode >= 0) return mode != 0;
  return waves <= kSimds;
}

uint64_t* g_wave_clocks = nullptr;   // diagnostics (tools/wave_spread.py)

// CRT decrypt: exponentiation modulo p^2 / q^2 in split form (hensel.hpp) where it is compiled for the key size;
// PGPU_HENSEL=0 keeps the full-width modexp_kernel (A/B measurements, parity tests of both paths).
// two batches in flight (the two batch lanes, callers that keep two streams busy): take the 256-register build of the
// (2,19) decrypt kernel also for launches of one wavefront per SIMD, so that the launches of two streams can share a
// SIMD (the full-budget build holds 291 registers: nothing else fits beside it).  pgpu_debug_set_packed_decrypt /
// PGPU_PACKED_DECRYPT=0 selects the full-budget build again.
std::atomic<bool> g_packed_decrypt{[] {
  const char* e = std::getenv("PGPU_PACKED_DECRYPT");
  return !(e && std::atoi(e) == 0);   // default since round 3: with aligned code both builds run a lone launch equally fast
}()};
// Window-table access of the split-form kernels (include/pgpu.h, SIDE CHANNELS): 0 = indexed by the exponent digit
// (default), 1 = every entry is read and the wanted one selected (pgpu_set_table_gather_policy / PGPU_CT_GATHER=1)
std::atomic<int> g_ct_gather{[] {
  const char* e = std::getenv("PGPU_CT_GATHER");
  return e && std::atoi(e) != 0 ? 1 : 0;
}()};
// PGPU_PAIR_ROWS=0: resident ciphertext batches stay Montgomery-form words (the round-2 representation; A/B)
bool pair_rows_enabled() {
  static const bool on = [] {
    const char* e = std::getenv("PGPU_PAIR_ROWS");
    return !(e && std::atoi(e) == 0);
  }();
  return on;
}
std::atomic<int> g_hensel{-2};
bool hensel_enabled() {
  int mode = g_hensel.load();
  if (mode == -2) {
    const char* e = std::getenv("PGPU_HENSEL");
    mode = (e && std::atoi(e) == 0) ? 0 : 1;
    g_hensel.store(mode);
  }
  return mode != 0;
}

// common launcher of modexp_kernel on device `d`, stream `s`: sizes the window table of the stream's
// workspace and fills the shared fields.  sched: per-context sliding-window schedules (device arrays) or null.
struct SchedRef {
  const uint16_t* p[2] = {nullptr, nullptr};
  int len[2] = {0, 0};
  int w = 0;
};
int run_modexp(rt::Device& d, pgpu::ModexpArgs& a, const GeoInfo& ctx_geo, hipStream_t s,
               const SchedRef* sched = nullptr, rt::StreamWork* held = nullptr) {
  const GeoInfo geo = launch_geo(ctx_geo, a.count);
  size_t entries;
  a.parity_waves = 0;
  if (sched && sched->p[0]) {
    a.parity_waves = a.nctx == 2;
    a.window = sched->w;
    for (int i = 0; i < 2; ++i) {
      a.sched[i] = sched->p[a.nctx == 2 ? i : 0];
      a.sched_len[i] = sched->len[a.nctx == 2 ? i : 0];
    }
    entries = (size_t)1 << (a.window - 1);
  } else {
    a.window = pick_window(a.exp_bits);
    entries = (size_t)1 << a.window;
  }
  const size_t waves = modexp_waves(a.count, a.parity_waves, geo.ipw());
  const size_t padded = (waves + pgpu::kWavesPerWG - 1) / pgpu::kWavesPerWG * pgpu::kWavesPerWG * geo.ipw();
  rt::StreamWork& w = held ? *held : d.work_for(s);
  std::unique_lock<std::mutex> lk(w.mu, std::defer_lock);
  if (!held) lk.lock();
  RC_TRY(w.table.ensure(padded * (entries + 1) * geo.L() * sizeof(uint32_t), s));   // + the parking slot
  a.table = (uint32_t*)w.table.p;
  a.wave_clocks = g_wave_clocks;
  TimerScope t(d, s, PGPU_KERNEL_MODEXP);
  const unsigned blocks = (unsigned)((waves + pgpu::kWavesPerWG - 1) / pgpu::kWavesPerWG);
  if (!pgpu::launch_modexp(geo.G, geo.K, use_regrows(geo, waves), a, blocks, s))
    return fail(PGPU_ERR_UNSUPPORTED, "modexp kernel geometry not compiled");
  HIP_TRY(hipGetLastError());
  t.stop();
  return PGPU_OK;
}

}  // namespace

// ---------- Paillier key objects ----------
struct FbTable {   // immutable once built: hs^(d * 2^(w*i)) * R, [nwin][2^w][L]
  void* p = nullptr;
  int w = 0, nwin = 0;
  hipEvent_t ready = nullptr;   // recorded behind the build; launches on other streams wait for it
  size_t bytes = 0;
  uint64_t tick = 0;            // last use (LRU over all keys of a device)
  int pins = 0;                 // lookups whose launch has not been queued yet: never evicted
  double build_ms = 0;          // filled in lazily from the events below (bench / diagnostics)
  hipEvent_t t0 = nullptr;
};
struct pgpu_pubkey {
  int n_words = 0;
  BigNumber n;
  bool djn = false;
  std::shared_ptr<ModCtx> nsq;  // modulus n^2, with nr = n*R mod n^2
  rt::Replicated d_hs;          // DJN: hs, 2*n_words words
  rt::Replicated d_n;           // plain: the exponent n, n_words words
  ExpSchedule sched_n;          // plain: sliding-window schedule of n (r^n mod n^2; n is public)
  // fixed-base tables for hs^r, per pool device; built lazily, kept until the key dies
  mutable std::mutex mu;
  // [device]; entries never move once handed out (std::list).  Guarded by the process-wide g_fb_mu, not by `mu`:
  // a table may be evicted to make room for ANOTHER key's table (fb_budget)
  mutable std::vector<std::list<FbTable>> fb;
  mutable size_t fb_elems = 0;  // elements encrypted with this key so far (window policy)
  // split forms of n^2 = (n)^2 (hensel.hpp) compiled for this key size, most lanes per element first; empty: none
  struct PubForm {   // (shared: resident pair-row batches keep the form they were produced under alive)
    int H = 0, K = 0, chunk_words = 0, nchunks = 0;
    int n_words = 0;
    BigNumber n;
    rt::Replicated pub;      // P | n | k*R mod n (L2 limbs each) | pair one | pairs conv | pairs conv (Montgomery input)
    rt::Replicated full;     // the way back in Geo<2H,K>, R' = 2^(29*2*L2): n^2 | n*R' | n*R'*Rs | R'*Rs  (2*L2 limbs
                             // each; Rs = the radix of the n^2 context, which Montgomery-form batches carry)
    uint32_t n0inv = 0, n0inv_full = 0;
  };
  std::vector<std::shared_ptr<PubForm>> hforms;
  mutable std::vector<std::list<FbTable>> fbh;   // [device]: fixed-base tables of pairs (form hforms.back())
  uint64_t gen = 0;             // pool generation the device images belong to
  ~pgpu_pubkey();
};

struct pgpu_privkey {
  uint64_t gen = 0;             // pool generation the device images belong to
  int n_words = 0;              // words of n (= words of p^2, q^2 rows)
  int pq_words = 0;
  GeoInfo geo_exp{};            // geometry of the two half-width exponentiations
  GeoInfo geo_crt{};            // geometry of the recombination kernel
  std::shared_ptr<ModCtx> p2, q2;   // moduli p^2, q^2 (fc = hp / hq, r2s set)
  GeoInfo geo_lat{};            // latency geometry of the exponentiations (== geo_exp if there is none)
  std::shared_ptr<ModCtx> p2l, q2l; // the same moduli for geo_lat (aliases of p2, q2 when L is equal)
  std::shared_ptr<ModCtx> cM, cQ;   // auxiliary modulus M, modulus q (CRT geometry)
  rt::Replicated d_exps;        // [2][pq_words]: p-1, q-1
  int exp_bits = 0;
  ExpSchedule sched[2];         // sliding-window schedules of p-1 and q-1 (used under PGPU_EXP_SLIDING only)
  rt::Replicated d_crt32;       // cp | cq | pinvR | pRM  (29-bit limbs, CRT geometry)
  rt::Replicated d_crt64;       // hp | hq | p^2 | q^2 | q   (n_words words each)
  int nsq_rbits = 0;            // R of the n^2 context: Montgomery-form ciphertexts carry this factor
  // split-form exponentiation (hensel.hpp): the forms compiled for this key size, most lanes per ciphertext
  // (shortest serial chain) first; a launch takes the first one that still puts at most one wavefront on a SIMD
  struct HenselSet {
    int H = 0, K = 0;           // 2H lanes per ciphertext side, K limbs per lane; L2 = H*K limbs per half
    int chunk_words = 0, nchunks = 0;
    rt::Replicated blob;        // per side: P | p | h | k*R mod p (L2 limbs each) | pair one | pairs conv | pairs conv (Montgomery input)
                                //           | pairs pconv[pchunks] | pcb[pchunks] (L2 limbs each): entry from pair rows
    uint32_t n0inv[2] = {0, 0};
    int pair_l2 = 0, pchunk_limbs = 0, pchunks = 0;   // pair rows of this key's n^2 domain (0: none)
    int lb = pgpu::kLimbBits;   // bits per limb of the constants (the pair ROWS are 29-bit limbs whatever this says)
    size_t side_words() const { return (size_t)H * K * (6 + 4 * (size_t)nchunks + 3 * (size_t)pchunks); }
  };
  std::vector<std::unique_ptr<HenselSet>> hs;
  // the constants of hensel_decrypt_ps_kernel (hensel_ps.hpp: one lane per exponentiation, product scanning): H = 1,
  // K limbs of lb bits per half (2048-bit keys: 38 x 28); null: no such kernel for this key size
  std::unique_ptr<HenselSet> hs_ps;
  // constants of the pair rows of n^2 = (p*q)^2 -- what a public key over n holds as its pair form: word ciphertexts are
  // brought into pair rows with it when the launch then takes a kernel that reads only those (decrypt_on)
  std::shared_ptr<pgpu_pubkey::PubForm> conv_form;
};

// sharded device-resident batch
struct pgpu_batch {
  size_t count = 0;
  int words = 0;
  int ndev = 1;                        // devices the shards are spread over; count == 1: a copy everywhere
  bool replicated = false;
  std::vector<rt::DevMem> shard;
  std::shared_ptr<ModCtx> mont;        // non-null: values are x*R mod N (canonical) for this context
  // pair rows (kargs.hpp): pair_l2 > 0 -> a row is 2*pair_l2 29-bit limbs (pair_l2 64-bit words of storage), the pair of
  // c*R modulo pair_form->n squared; `words` stays the logical width of the values (download)
  int pair_l2 = 0;
  std::shared_ptr<const pgpu_pubkey::PubForm> pair_form;   // constants of the domain (conversions need no key)
  const uint32_t* prow(int d) const { return (const uint32_t*)shard[(size_t)d].p; }
  uint32_t* prow(int d) { return (uint32_t*)shard[(size_t)d].p; }
  uint64_t gen = 0;                    // pool generation of the shards
  int lane = 0;                        // batch lane (stream) its shards are ordered on: results inherit the lane of
                                       // their first operand, uploads take the calling thread's lane (pgpu_set_batch_lane)
  uint64_t* ptr(int d) const { return (uint64_t*)shard[(size_t)d].p; }
  void bounds(int d, size_t* lo, size_t* hi) const {
    if (replicated) { *lo = 0; *hi = count; }
    else rt::shard_bounds(count, ndev, d, lo, hi);
  }
  // Cross-lane use (lane_acquire / lane_release): per device ONE event that says "produced" -- recorded on the batch's lane
  // the first time another lane consumes it, never again (batches are immutable), so later consumers wait on an event
  // that has long completed -- and per consuming lane an event recorded behind its latest reader.  The batch's own lane
  // waits for the readers only when the batch DIES (before its memory returns to that lane's allocator), not after every
  // use: operands shared between lanes -- a cached randomness batch, a ciphertext several threads read -- no longer chain
  // the lanes to each other (round 4: four API threads sharing injected randomness ran their kernels two at a time).
  struct XLane {
    hipEvent_t ready = nullptr;
    hipEvent_t reader[rt::kBatchLanes] = {};
  };
  mutable std::mutex xmu;
  mutable std::vector<XLane> xlane;    // [device], sized on first cross-lane use
  ~pgpu_batch();
};

namespace {

// The batch lane of the calling thread's uploads: set with pgpu_set_batch_lane, else handed out round-robin the first
// time a thread uploads or creates a batch -- the first thread of the process gets lane 0, so single-threaded callers see
// what they always saw, and the threads of a multi-threaded caller (the reference's own tests encrypt / decrypt from
// four OpenMP threads, test_cryptography.cpp:45-57) land on different lanes: their chains overlap on the GPU and the
// adaptive kernel-form policy places their launches side by side.
std::atomic<int> g_next_thread_lane{0};
thread_local int t_batch_lane = -1;
// true once the thread has chosen a lane itself (pgpu_set_batch_lane): a caller that PIPELINES -- keeps several lanes fed
// without waiting in between.  Only such callers enter the adaptive kernel-form policy (busy_other_lanes).  Threads on
// the lanes handed out round-robin are synchronous API callers (upload, operation, download, wait): their lanes fall idle
// for the host part of every call, and a part-chip launch beside an idle neighbour is the worst case of that policy
// (measured r04, tests/cpp/ipcl_bench.cpp --threads 2: 11.8 ms per encrypt + decrypt against 7.2 with full-chip launches).
thread_local bool t_lane_explicit = false;
std::atomic<int> g_host_adapt{[] { const char* e = std::getenv("PGPU_HOST_ADAPT"); return e && std::atoi(e) != 0 ? 1 : 0; }()};
bool host_adapt() { return g_host_adapt.load() != 0; }
int thread_batch_lane() {
  if (t_batch_lane < 0) t_batch_lane = g_next_thread_lane.fetch_add(1) % rt::kBatchLanes;
  return t_batch_lane;
}
int new_batch(size_t count, int words, std::unique_ptr<pgpu_batch>* out, int pair_l2 = 0, int lane = -1) {
  if (count == 0 || words <= 0) return fail(PGPU_ERR_INVALID_PARAM, "batch needs count > 0 and words > 0");
  std::unique_ptr<pgpu_batch> b(new pgpu_batch);
  b->count = count;
  b->words = words;
  b->gen = rt::pool_generation();
  b->lane = lane < 0 ? thread_batch_lane() : (lane % rt::kBatchLanes);
  b->replicated = count == 1 && rt::pool_size() > 1;
  b->ndev = b->replicated ? rt::pool_size() : rt::shard_devices(count);
  b->shard.resize((size_t)b->ndev);
  for (int d = 0; d < b->ndev; ++d) {
    size_t lo, hi;
    b->bounds(d, &lo, &hi);
    rt::Device& dev = rt::device(d);
    RC_TRY(b->shard[(size_t)d].alloc(dev, dev.bs(b->lane), (hi - lo) * (size_t)(pair_l2 ? pair_l2 : words) * 8));
  }
  b->pair_l2 = pair_l2;
  *out = std::move(b);
  return PGPU_OK;
}

// A Montgomery-form batch fits a key when it was produced under the same modulus and geometry -- the same key
// object, or another object built from the same n (copies of an ipcl::PublicKey each own a device key).
bool same_domain(const std::shared_ptr<ModCtx>& batch_ctx, const std::shared_ptr<ModCtx>& key_ctx) {
  if (!batch_ctx || batch_ctx == key_ctx) return true;
  return batch_ctx->geo.L() == key_ctx->geo.L() && batch_ctx->geo.G == key_ctx->geo.G && batch_ctx->N == key_ctx->N;
}

// operands of one operation must be cut the same way (they are, unless min_shard changed in between)
int same_layout(const pgpu_batch* a, const pgpu_batch* b) {
  if (b->count == 1 && a->count != 1) return PGPU_OK;   // broadcast operand: a copy everywhere (or on device 0)
  if (a->ndev != b->ndev || a->replicated != b->replicated)
    return fail(PGPU_ERR_INVALID_PARAM, "batches were sharded differently (pgpu_set_min_shard changed in between)");
  return PGPU_OK;
}

// ---- batch lanes ----
// Two independent chains of resident batches may be in flight on a GPU (Device::bs(0/1)): a result lives in the lane of
// the operation's first operand.  An operand of the OTHER lane is ordered in before the launch (its producer has to be
// done) and its lane is ordered behind the launch afterwards (its memory may be recycled by that lane's allocator only
// after this reader is done).
int lane_acquire(rt::Device& dev, const pgpu_batch* x, int lane) {
  if (!x || x->lane == lane) return PGPU_OK;
  hipEvent_t ready;
  {
    std::lock_guard<std::mutex> lk(x->xmu);
    if (x->xlane.size() < (size_t)rt::pool_size()) x->xlane.resize((size_t)rt::pool_size());
    pgpu_batch::XLane& xl = x->xlane[(size_t)dev.index];
    if (!xl.ready) {
      HIP_TRY(hipEventCreateWithFlags(&xl.ready, hipEventDisableTiming));
      HIP_TRY(hipEventRecord(xl.ready, dev.bs(x->lane)));   // everything queued on its lane so far: its producer among it
    }
    ready = xl.ready;
  }
  HIP_TRY(hipStreamWaitEvent(dev.bs(lane), ready, 0));
  return PGPU_OK;
}
int lane_release(rt::Device& dev, const pgpu_batch* x, int lane) {
  if (!x || x->lane == lane) return PGPU_OK;
  std::lock_guard<std::mutex> lk(x->xmu);
  if (x->xlane.size() < (size_t)rt::pool_size()) x->xlane.resize((size_t)rt::pool_size());
  hipEvent_t& ev = x->xlane[(size_t)dev.index].reader[lane % rt::kBatchLanes];
  if (!ev) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(ev, dev.bs(lane)));   // (the batch's lane waits for it when the batch dies: ~pgpu_batch)
  return PGPU_OK;
}

// the same for every device an operand lives on (acquire before the launches of an operation, release after them)
int lanes_order(const pgpu_batch* x, int lane, bool acquire) {
  if (!x || x->lane == lane) return PGPU_OK;
  for (int d = 0; d < x->ndev; ++d) {
    rt::Device& dev = rt::device(d);
    rt::DeviceGuard g(dev.ordinal);
    RC_TRY(acquire ? lane_acquire(dev, x, lane) : lane_release(dev, x, lane));
  }
  return PGPU_OK;
}

}  // namespace
pgpu_batch::~pgpu_batch() {
  const bool live = gen == rt::pool_generation() && rt::initialized();
  for (size_t d = 0; d < xlane.size(); ++d) {
    XLane& xl = xlane[d];
    const bool dev_ok = live && (int)d < rt::pool_size();
    std::unique_ptr<rt::DeviceGuard> g(dev_ok ? new rt::DeviceGuard(rt::device((int)d).ordinal) : nullptr);
    for (hipEvent_t ev : xl.reader)
      if (ev) {
        if (dev_ok) (void)hipStreamWaitEvent(rt::device((int)d).bs(lane), ev, 0);   // readers first, then the memory goes back
        (void)hipEventDestroy(ev);
      }
    if (xl.ready) (void)hipEventDestroy(xl.ready);
  }
  if (!xlane.empty()) (void)hipGetLastError();
}
namespace {
// ---- launch helpers on one device ----
int modmul_on(rt::Device& d, const ModCtx& ctx, int mode, const uint64_t* a, const uint64_t* b, size_t b_stride,
              int b_words, uint64_t* out, size_t count, hipStream_t s, int view_flags = VF_NONE) {
  pgpu::ModmulArgs m{};
  m.ctx = ctx.view(d.index, view_flags);
  m.a = a;
  m.a_stride = (size_t)ctx.mod_words;
  m.b = b;
  m.b_stride = b_stride;
  m.in_words = ctx.mod_words;
  m.out = out;
  m.count = count;
  m.mode = mode;
  m.b_words = b_words;
  TimerScope t(d, s, PGPU_KERNEL_MODMUL);
  const GeoInfo lgeo = launch_geo(ctx.geo, count);
  if (!pgpu::launch_modmul(lgeo.G, lgeo.K, m, blocks_for(count, lgeo), s))
    return fail(PGPU_ERR_UNSUPPORTED, "modmul kernel geometry not compiled");
  HIP_TRY(hipGetLastError());
  t.stop();
  return PGPU_OK;
}

// ---------- perfect-square moduli through the generic seam ----------
// ipcl::modExp is the seam the reference's own encrypt / decrypt / CT x PT reach with moduli n^2, p^2, q^2
// (pub_key.cpp:51-80, pri_key.cpp:128-134, ciphertext.cpp:143-162): a modulus that is a perfect square of an odd
// root takes the split form (hensel.hpp: hensel_modexp_kernel) here too, whoever calls.  One throughput and one latency
// form per root size; the way back to a full-width residue has constants of its own (geometry Geo<2H,K>), so no
// relation to the full-width context of the modulus is needed.
struct SquareForm {
  int H = 0, K = 0, chunk_words = 0, nchunks = 0;
  rt::Replicated pub;    // P | root | k*R mod root (L2 limbs each) | pair one | pairs conv
  rt::Replicated full;   // N | root*R' mod N   (2*L2 limbs each)
  uint32_t n0inv_root = 0, n0inv_full = 0;
};
struct SquareCtx {
  int mod_words = 0;
  std::vector<std::unique_ptr<SquareForm>> forms;   // most lanes per element (shortest serial chain) first
};
std::mutex g_sq_mu;
LruCache<SquareCtx> g_sq_cache;   // null entry: not a (supported) square

// floor(sqrt(N)) by Newton's iteration from above
BigNumber isqrt(const BigNumber& N) {
  if (N.isZero()) return N;
  BigNumber x = pow2((N.BitSize() + 1) / 2);
  for (;;) {
    BigNumber y = (x + N / x) / BigNumber((Ipp32u)2);
    if (y >= x) return x;
    x = y;
  }
}

int build_square_form(const BigNumber& N, const BigNumber& root, int mod_words, int H, int K,
                      std::unique_ptr<SquareForm>* out) {
  const int L2 = H * K;
  const int cw = std::min(mod_words, root.BitSize() / 64);
  if (cw <= 0) return PGPU_OK;
  const int nch = (mod_words + cw - 1) / cw;
  std::unique_ptr<SquareForm> f(new SquareForm);
  f->H = H;
  f->K = K;
  f->chunk_words = cw;
  f->nchunks = nch;
  auto n0inv_of = [](const BigNumber& v) {
    uint32_t n0 = (uint32_t)(v.limbs64()[0] & pgpu::kLimbMask), inv = n0;
    for (int i = 0; i < 5; ++i) inv *= 2u - n0 * inv;
    return (0u - inv) & pgpu::kLimbMask;
  };
  f->n0inv_root = n0inv_of(root);
  f->n0inv_full = n0inv_of(N);
  const BigNumber P = root * BigNumber((Ipp32u)f->n0inv_root), P2 = P * P;
  const BigNumber R = pow2(L2 * pgpu::kLimbBits);
  std::vector<uint32_t> h((size_t)L2 * (5 + 2 * (size_t)nch), 0);
  auto put_pair = [&](uint32_t* dst, const BigNumber& z) {
    const BigNumber zr = z % P2;
    const BigNumber q = zr / P;
    to_limbs29(zr % P, L2, dst);
    to_limbs29(q.isZero() ? q : P - q, L2, dst + L2);
  };
  to_limbs29(P, L2, h.data());
  to_limbs29(root, L2, h.data() + L2);
  to_limbs29((R % root) * BigNumber((Ipp32u)f->n0inv_root) % root, L2, h.data() + 2 * L2);
  const BigNumber Rm = R % P2, R2 = (Rm * Rm) % P2;
  put_pair(h.data() + 3 * L2, Rm);
  for (int i = 0; i < nch; ++i) put_pair(h.data() + 5 * L2 + (size_t)i * 2 * L2, (R2 * (pow2(64 * cw * i) % P2)) % P2);
  // The caller of the key-less seam may be the reference's own decryptCRT (pri_key.cpp:128-134: moduli p^2, q^2): the
  // root is then a private prime.  The images are therefore treated as key material: zeroed on the devices before
  // they are freed, the host staging copies wiped, and the cache entry dropped at pgpu_shutdown.
  RC_TRY(f->pub.upload(h.data(), h.size() * sizeof(uint32_t), true));
  secure_wipe(h.data(), h.size() * sizeof(uint32_t));
  std::vector<uint32_t> g((size_t)4 * L2, 0);
  to_limbs29(N, 2 * L2, g.data());
  to_limbs29((root * (pow2(2 * L2 * pgpu::kLimbBits) % N)) % N, 2 * L2, g.data() + 2 * L2);
  RC_TRY(f->full.upload(g.data(), g.size() * sizeof(uint32_t), true));
  secure_wipe(g.data(), g.size() * sizeof(uint32_t));
  *out = std::move(f);
  return PGPU_OK;
}

// the split-form description of a modulus, or null (not an odd perfect square, or no compiled form fits its root)
int get_square_ctx(const uint64_t* mod, int mod_words, std::shared_ptr<SquareCtx>* out) {
  out->reset();
  if (!hensel_enabled() || !(mod[0] & 1)) return PGPU_OK;
  std::vector<uint64_t> key(mod, mod + mod_words);
  std::lock_guard<std::mutex> lk(g_sq_mu);
  if (g_sq_cache.find(key, out)) return PGPU_OK;
  std::shared_ptr<SquareCtx> ctx;
  const BigNumber N = BigNumber::fromLimbs64(mod, (size_t)mod_words);
  // (a square is 0, 1, 4 or 9 mod 16 -- an odd one is 1 or 9: cheap rejection of almost every other modulus)
  const unsigned low = (unsigned)(mod[0] & 15);
  if ((low == 1 || low == 9) && N.BitSize() > 128) {
    const BigNumber root = isqrt(N);
    if (root * root == N && root.IsOdd()) {
      const int need = root.BitSize() + 29 + 8;
      ctx = std::make_shared<SquareCtx>();
      ctx->mod_words = mod_words;
      for (int H : {8, 4, 2})   // per lane count the smallest compiled form with enough limbs
        for (int K = 1; K <= 19; ++K)
          if (pgpu::hensel_modexp_has(H, K) && pgpu::kLimbBits * H * K >= need &&
              2 * pgpu::kLimbBits * H * K >= N.BitSize() + 8) {
            std::unique_ptr<SquareForm> f;
            RC_TRY(build_square_form(N, root, mod_words, H, K, &f));
            if (f) ctx->forms.push_back(std::move(f));
            break;
          }
      if (ctx->forms.empty()) ctx.reset();
    }
  }
  g_sq_cache.insert(key, ctx);
  *out = ctx;
  return PGPU_OK;
}

int modexp_square_on(rt::Device& d, const SquareCtx& sq, const uint64_t* d_base, size_t base_stride,
                     const uint64_t* d_exp, size_t exp_stride, int exp_words, int exp_bits, uint64_t* d_out,
                     size_t count, hipStream_t s, const SchedRef* sched, size_t total_count) {
  // total_count: the batch this launch is a sub-batch of (sub-batches of one call run side by side on the chip)
  const SquareForm* f = sq.forms.back().get();
  for (const auto& c : sq.forms) {   // the form with the most lanes per element that still fits one wavefront per SIMD
    const size_t ipw = 64 / (2 * (size_t)c->H);
    if ((total_count + ipw - 1) / ipw <= kSimds) { f = c.get(); break; }
  }
  const int L2 = f->H * f->K;
  const uint32_t* pb = (const uint32_t*)f->pub.d[(size_t)d.index];
  const uint32_t* fb = (const uint32_t*)f->full.d[(size_t)d.index];
  pgpu::HenselModexpArgs a{};
  a.ctx.nhat = pb;
  a.ctx.n = pb + L2;
  a.ctx.kr = pb + 2 * L2;
  a.ctx.one = pb + 3 * L2;
  a.ctx.conv = pb + 5 * L2;
  a.ctx.n0inv = f->n0inv_root;
  a.full.n = fb;
  a.full.nr = fb + 2 * L2;
  a.full.r2 = nullptr;
  a.full.n0inv = f->n0inv_full;
  a.full.mod_words = sq.mod_words;
  a.base = d_base;
  a.base_stride = base_stride;
  a.base_words = sq.mod_words;
  a.chunk_words = f->chunk_words;
  a.nchunks = f->nchunks;
  a.exp = d_exp;
  a.exp_stride = exp_stride;
  a.exp_words = exp_words;
  a.exp_bits = exp_bits;
  size_t entries;
  if (sched && sched->p[0]) {
    a.sched = sched->p[0];
    a.sched_len = sched->len[0];
    a.window = sched->w;
    entries = (size_t)1 << (a.window - 1);
  } else {
    a.window = pick_window(exp_bits);
    entries = (size_t)1 << a.window;
  }
  a.final_mul = pgpu::FM_UNIT;
  a.ct_gather = (sched && sched->p[0]) ? 0 : g_ct_gather.load();
  a.out = d_out;
  a.out_stride = (size_t)sq.mod_words;
  a.count = count;
  const size_t ipw = 64 / (2 * (size_t)f->H);
  const size_t waves = (count + ipw - 1) / ipw;
  const unsigned blocks = (unsigned)((waves + pgpu::kWavesPerWG - 1) / pgpu::kWavesPerWG);
  rt::StreamWork& w = d.work_for(s);
  std::lock_guard<std::mutex> lk(w.mu);
  RC_TRY(w.table.ensure((size_t)blocks * pgpu::kWavesPerWG * ipw * entries * 2 * L2 * sizeof(uint32_t), s));
  a.table = (uint32_t*)w.table.p;
  TimerScope t(d, s, PGPU_KERNEL_MODEXP, PGPU_FORM_PAIRED);
  if (!pgpu::launch_hensel_modexp(f->H, f->K, a, blocks, s)) return fail(PGPU_ERR_UNSUPPORTED, "split-form modexp kernel not compiled");
  HIP_TRY(hipGetLastError());
  t.stop();
  return PGPU_OK;
}

// generic modexp on one device (bases / result plain or in the context's Montgomery form)
int modexp_on(rt::Device& d, const uint64_t* d_base, size_t base_stride, const uint64_t* d_exp, size_t exp_stride,
              int exp_words, int exp_bits, const uint64_t* h_mod, int mod_words, uint64_t* d_out, size_t count,
              hipStream_t s, const SchedRef* sched, bool in_mont, bool out_mont, std::shared_ptr<ModCtx> ctx_in,
              size_t total_count = 0) {
  std::shared_ptr<ModCtx> ctx = ctx_in;
  if (!in_mont && !out_mont) {   // the key-less seam (plain in, plain out): a perfect-square modulus takes the split form
    std::shared_ptr<SquareCtx> sq;
    RC_TRY(get_square_ctx(h_mod, mod_words, &sq));
    if (sq)
      return modexp_square_on(d, *sq, d_base, base_stride, d_exp, exp_stride, exp_words, exp_bits, d_out, count, s, sched,
                              std::max(count, total_count));
  }
  if (!ctx) RC_TRY(get_modctx(h_mod, mod_words, true, &ctx));
  GeoInfo run_geo = ctx->geo;
  const GeoInfo lat = latency_geo(ctx->geo);
  if (!in_mont && !out_mont && use_latency_geo(lat, ctx->geo, count)) {   // small batch: 16 lanes per element
    if (lat.L() != ctx->geo.L()) {
      std::vector<uint64_t> mw((size_t)mod_words);
      ctx->N.toLimbs64(mw.data(), mw.size());
      RC_TRY(get_modctx(mw.data(), mod_words, true, &ctx, true));
    }
    run_geo = lat;
  }
  pgpu::ModexpArgs a{};
  a.ctx[0] = a.ctx[1] = ctx->view(d.index, (in_mont ? VF_IN_MONT : 0) | (out_mont ? VF_OUT_MONT : 0));
  a.nctx = 1;
  a.base = d_base;
  a.base_stride = base_stride;
  a.base_words = mod_words;
  a.exp = d_exp;
  a.exp_stride = exp_stride;
  a.exp_per_ctx = 0;
  a.exp_words = exp_words;
  a.exp_bits = exp_bits;
  a.final_mul = out_mont ? pgpu::FM_CTX_CONST : pgpu::FM_UNIT;
  a.out = d_out;
  a.out_stride = (size_t)mod_words;
  a.count = count;
  return run_modexp(d, a, run_geo, s, sched);
}

// ---------- fixed-base tables: budget and LRU ----------
// A table is 13 MB (w = 8) to 203 MB (w = 12) per 2048-bit key and GPU.  Two limits keep a server with thousands of
// keys inside its memory (pgpu_set_fixed_base_budget / PGPU_FB_MAX_BYTES, PGPU_FB_KEY_MAX_BYTES):
//   * per key and GPU: the widest window <= the configured one whose table fits kFbKeyMax (default 512 MiB: w = 12
//     also for the 2047-bit randomness of the reference's benchmark fixture, 403 MB);
//   * per GPU, over all keys: kFbDevMax (default 2 GiB).  A new table that would exceed it evicts the least
//     recently used tables of any key on that GPU first (never one whose launch is still being queued); if the
//     budget cannot hold the table at all, the window shrinks until it does.
// An evicted table is freed on the spot (hipFree waits for the kernels that still read it): making room costs a
// device synchronisation, staying inside the budget costs nothing.
std::mutex g_fb_mu;                       // all FbTable lists, the registry, the counters
std::set<const pgpu_pubkey*> g_fb_keys;   // keys that own tables
std::vector<size_t> g_fb_dev_bytes;       // [device] bytes of live tables
std::atomic<size_t> g_fb_dev_max{0}, g_fb_key_max{0};
std::atomic<uint64_t> g_fb_tick{0}, g_fb_evictions{0};
size_t fb_dev_max() {
  size_t v = g_fb_dev_max.load();
  if (v == 0) {
    const char* e = std::getenv("PGPU_FB_MAX_BYTES");
    v = e && std::atoll(e) > 0 ? (size_t)std::atoll(e) : (size_t)2 << 30;
    g_fb_dev_max.store(v);
  }
  return v;
}
size_t fb_key_max() {
  size_t v = g_fb_key_max.load();
  if (v == 0) {
    const char* e = std::getenv("PGPU_FB_KEY_MAX_BYTES");
    v = e && std::atoll(e) > 0 ? (size_t)std::atoll(e) : (size_t)512 << 20;
    g_fb_key_max.store(v);
  }
  return v;
}
size_t fb_table_bytes(int r_bits, int w, size_t entry_bytes) {
  return (size_t)std::max(1, (r_bits + w - 1) / w) * ((size_t)1 << w) * entry_bytes;
}
// the window a table of this key may use on a GPU: <= w_cfg, fits the per-key limit and the per-GPU budget
int fb_fit_window(int w_cfg, int r_bits, size_t entry_bytes) {
  int w = w_cfg;
  const size_t cap = std::min(fb_key_max(), fb_dev_max());
  while (w > 1 && fb_table_bytes(r_bits, w, entry_bytes) > cap) --w;
  return w;
}
void fb_free_table(FbTable& t) {
  if (t.ready) (void)hipEventDestroy(t.ready);
  if (t.t0) (void)hipEventDestroy(t.t0);
  if (t.p) (void)hipFree(t.p);   // (synchronises the device: kernels that read the table have finished)
  t.p = nullptr;
}
// g_fb_mu held.  Frees least-recently-used, unpinned tables on device `dev` until `need` more bytes fit.
void fb_make_room(int dev, size_t need) {
  if (g_fb_dev_bytes.size() <= (size_t)dev) g_fb_dev_bytes.resize((size_t)dev + 1, 0);
  while (g_fb_dev_bytes[(size_t)dev] + need > fb_dev_max()) {
    std::list<FbTable>* vlist = nullptr;
    std::list<FbTable>::iterator victim;
    for (const pgpu_pubkey* k : g_fb_keys)
      for (auto* lists : {&k->fb, &k->fbh}) {
        if (lists->size() <= (size_t)dev) continue;
        auto& l = (*lists)[(size_t)dev];
        for (auto it = l.begin(); it != l.end(); ++it)
          if (it->pins == 0 && (!vlist || it->tick < victim->tick)) {
            vlist = &l;
            victim = it;
          }
      }
    if (!vlist) return;   // everything left is pinned: the caller allocates beyond the budget rather than fail
    g_fb_dev_bytes[(size_t)dev] -= std::min(g_fb_dev_bytes[(size_t)dev], victim->bytes);
    fb_free_table(*victim);
    vlist->erase(victim);
    g_fb_evictions.fetch_add(1);
  }
}
// RAII pin: the table cannot be evicted between the lookup and the moment its consumer kernel is queued
struct FbPin {
  FbTable* t = nullptr;
  FbPin() = default;
  FbPin(const FbPin&) = delete;
  FbPin& operator=(const FbPin&) = delete;
  ~FbPin() { release(); }
  void release() {
    if (!t) return;
    std::lock_guard<std::mutex> lk(g_fb_mu);
    --t->pins;
    t = nullptr;
  }
  const FbTable* operator->() const { return t; }
};

// Looks up / builds the table of (key, device) for window w covering nwin windows.  `build` queues the build kernel
// on s for a freshly allocated table; entry_bytes: bytes per table entry.
template <class Build>
int fb_table_get(const pgpu_pubkey* key, std::vector<std::list<FbTable>>& lists, rt::Device& d, int w, int nwin,
                 size_t entry_bytes, hipStream_t s, FbPin* out, Build build) {
  std::lock_guard<std::mutex> lk(g_fb_mu);
  if (lists.size() < (size_t)rt::pool_size()) lists.resize((size_t)rt::pool_size());
  auto& list = lists[(size_t)d.index];
  for (FbTable& t : list)
    if (t.w == w && t.nwin >= nwin) {
      HIP_TRY(hipStreamWaitEvent(s, t.ready, 0));
      t.tick = g_fb_tick.fetch_add(1) + 1;
      ++t.pins;
      out->t = &t;
      return PGPU_OK;
    }
  // a key keeps ONE table per device and access mode: a wider window replaces the young key's narrow one (the small table
  // of the masked product lives beside the indexed one, so that switching the gather policy does not rebuild tables)
  for (auto it = list.begin(); it != list.end();) {
    if (it->pins == 0 && (it->w <= kFbMaskedMaxW) == (w <= kFbMaskedMaxW)) {
      g_fb_dev_bytes.resize(std::max(g_fb_dev_bytes.size(), (size_t)d.index + 1), 0);
      g_fb_dev_bytes[(size_t)d.index] -= std::min(g_fb_dev_bytes[(size_t)d.index], it->bytes);
      fb_free_table(*it);
      it = list.erase(it);
    } else {
      ++it;
    }
  }
  FbTable t;
  t.w = w;
  t.nwin = nwin;
  t.bytes = (size_t)nwin * ((size_t)1 << w) * entry_bytes;
  fb_make_room(d.index, t.bytes);
  hipError_t e = hipMalloc(&t.p, t.bytes);
  if (e != hipSuccess) return fail(PGPU_ERR_HIP, std::string("fixed-base table: ") + hipGetErrorString(e));
  if (hipEventCreateWithFlags(&t.ready, hipEventDefault) != hipSuccess || hipEventCreateWithFlags(&t.t0, hipEventDefault) != hipSuccess) {
    fb_free_table(t);
    return fail(PGPU_ERR_HIP, "fixed-base table: event creation failed");
  }
  (void)hipEventRecord(t.t0, s);
  int rc = build(t);
  if (rc == PGPU_OK && hipGetLastError() != hipSuccess) rc = fail(PGPU_ERR_HIP, "fixed-base build launch failed");
  if (rc == PGPU_OK && hipEventRecord(t.ready, s) != hipSuccess) rc = fail(PGPU_ERR_HIP, "fixed-base table: event record failed");
  if (rc != PGPU_OK) {
    fb_free_table(t);   // (round-2 advisor: the error paths leaked the table and its event)
    return rc;
  }
  t.tick = g_fb_tick.fetch_add(1) + 1;
  t.pins = 1;
  g_fb_dev_bytes.resize(std::max(g_fb_dev_bytes.size(), (size_t)d.index + 1), 0);
  g_fb_dev_bytes[(size_t)d.index] += t.bytes;
  g_fb_keys.insert(key);
  list.push_back(t);
  out->t = &list.back();
  return PGPU_OK;
}

// fixed-base table of (key, device) for window w covering nwin windows: built on first use
int fb_table_for(const pgpu_pubkey* key, rt::Device& d, int w, int nwin, hipStream_t s, FbPin* out) {
  const GeoInfo& geo = key->nsq->geo;
  return fb_table_get(key, key->fb, d, w, nwin, (size_t)geo.L() * sizeof(uint32_t), s, out, [&](FbTable& t) -> int {
    pgpu::FixedBaseBuildArgs b{};
    b.ctx = key->nsq->view(d.index);
    b.base = (const uint64_t*)key->d_hs.d[(size_t)d.index];
    b.table = (uint32_t*)t.p;
    b.nwin = nwin;
    b.w = w;
    if (!pgpu::launch_fb_build(geo.G, geo.K, b, blocks_for((size_t)nwin, geo), s))
      return fail(PGPU_ERR_UNSUPPORTED, "fixed-base build kernel geometry not compiled");
    return PGPU_OK;
  });
}

// fused encrypt on one device; out_mont: ciphertexts leave in the Montgomery domain of n^2
// DJN encrypt in split form (hensel.hpp): needs m < 2n to form the pair of 1 + n*m, i.e. plaintext rows no wider
// than n; pays once the batch fills the chip in 2H-lane groups -- smaller batches run the full-width kernel in its
// 16-lane split, whose serial chain per product is shorter (Encrypt(16): 1.4 vs 1.8 ms)
const pgpu_pubkey::PubForm* use_split_encrypt(const pgpu_pubkey* key, int m_words, size_t count) {
  if (key->hforms.empty() || !hensel_enabled() || 64 * m_words > key->n.BitSize()) return nullptr;
  const pgpu_pubkey::PubForm* f = key->hforms.back().get();   // the form of fewest lanes per element
  if (!pgpu::hensel_fb_has(f->H, f->K)) return nullptr;
  if (g_hensel.load() >= 2) return f;   // tests: whatever the batch size
  const GeoInfo g = launch_geo(key->nsq->geo, count);
  const size_t ipw = 64 / (2 * (size_t)f->H);
  return (g.G <= 2 * f->H || (count + ipw - 1) / ipw >= kSimds) ? f : nullptr;
}
pgpu::HenselPubDev hensel_pub_view(const pgpu_pubkey::PubForm* f, int dev, bool base_mont = false) {
  const int L2 = f->H * f->K;
  const uint32_t* b = (const uint32_t*)f->pub.d[(size_t)dev];
  pgpu::HenselPubDev v{};
  v.nhat = b;
  v.n = b + L2;
  v.kr = b + 2 * L2;
  v.one = b + 3 * L2;
  v.conv = b + 5 * L2 + (base_mont ? (size_t)f->nchunks * 2 * L2 : 0);
  v.gm = b + (size_t)L2 * (5 + 4 * (size_t)f->nchunks);
  v.n0inv = f->n0inv;
  return v;
}
// ---- pair rows (kargs.hpp): the resident form of ciphertext batches of keys that have a split form ----
// the form whose limb count defines the rows of this key (fewest lanes per element), or null: no pair domain
const pgpu_pubkey::PubForm* pair_form(const pgpu_pubkey* key) {
  if (key->hforms.empty() || !hensel_enabled() || !pair_rows_enabled()) return nullptr;
  const pgpu_pubkey::PubForm* f = key->hforms.back().get();
  return pgpu::pair_ops_has(f->H, f->K) ? f : nullptr;
}
int pair_l2(const pgpu_pubkey* key) {
  const pgpu_pubkey::PubForm* f = pair_form(key);
  return f ? f->H * f->K : 0;
}
std::shared_ptr<const pgpu_pubkey::PubForm> pair_form_shared(const pgpu_pubkey* key) {
  return pair_form(key) ? key->hforms.back() : nullptr;
}
pgpu::HenselFullDev hensel_full_view(const pgpu_pubkey* /*key*/, const pgpu_pubkey::PubForm* f, int dev, bool out_mont) {
  const size_t LF = (size_t)2 * f->H * f->K;
  const uint32_t* b = (const uint32_t*)f->full.d[(size_t)dev];
  pgpu::HenselFullDev v{};
  v.n = b;
  v.nr = out_mont ? b + 2 * LF : b + LF;     // n*R'*Rs (Montgomery-form result) or n*R'
  v.r2 = out_mont ? b + 3 * LF : nullptr;
  v.n0inv = f->n0inv_full;
  v.mod_words = 2 * f->n_words;
  return v;
}

// base^exp modulo n^2 in split form (hensel.hpp: hensel_modexp_kernel): the form with the most lanes per element that
// still leaves at most one wavefront per SIMD, else the one of fewest lanes.  Null when the key has no split form.
// (PGPU_SPLIT_MODEXP_MAX_WAVES: larger launches take the full-width kernel -- A/B measurements; a 1 M-element CT x PT
// batch: 60.0 ms full width, 48.4 ms split.)
const pgpu_pubkey::PubForm* split_modexp_form(const pgpu_pubkey* key, size_t count) {
  if (key->hforms.empty() || !hensel_enabled()) return nullptr;
  static const size_t max_waves = [] {
    const char* e = std::getenv("PGPU_SPLIT_MODEXP_MAX_WAVES");
    return e && std::atol(e) > 0 ? (size_t)std::atol(e) : ~(size_t)0;
  }();
  const pgpu_pubkey::PubForm* last = nullptr;
  for (const auto& f : key->hforms) {
    if (!pgpu::hensel_modexp_has(f->H, f->K)) continue;
    last = f.get();
    const size_t ipw = 64 / (2 * (size_t)f->H);
    if ((count + ipw - 1) / ipw <= kSimds) return last;
  }
  if (!last) return nullptr;
  const size_t ipw = 64 / (2 * (size_t)last->H);
  return (count + ipw - 1) / ipw > max_waves ? nullptr : last;
}
// Other batch lanes of `dev` with work queued right now or fed within the activity window, seen from the lane that owns stream
// `s` (0 when `s` is no batch lane); no PGPU_RR_ADAPT threshold: the latency forms spend a whole wavefront per
// exponentiation, which only pays while the chip has SIMDs to spare -- two API threads with 700-element batches already fill it
// (measured: four threads x 700 elements 1.78 ms per encrypt + decrypt with the throughput forms, 2.01 with the latency forms)
int wave_neighbours(rt::Device& dev, hipStream_t s) {
  int lane = -1;
  for (int k = 0; k < rt::kBatchLanes; ++k)
    if (dev.bs(k) == s) lane = k;
  if (lane < 0) return 0;
  static const int64_t window_ns = [] {
    const char* e = std::getenv("PGPU_LANE_ACTIVE_MS");
    return (int64_t)(e ? std::max(0, std::atoi(e)) : 50) * 1000000;
  }();
  const int64_t now = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
  dev.lane_fed_ns[lane].store(now, std::memory_order_relaxed);   // (CT x PT callers are seen by their neighbours through this stamp)
  int busy = 0;
  for (int k = 0; k < rt::kBatchLanes; ++k) {
    if (k == lane) continue;
    const int64_t fed = std::max(dev.lane_fed_ns[k].load(std::memory_order_relaxed), dev.host_fed_ns[k].load(std::memory_order_relaxed));
    if ((fed != 0 && now - fed < window_ns) || hipStreamQuery(dev.bs(k)) == hipErrorNotReady) ++busy;
  }
  (void)hipGetLastError();
  return busy;
}
// the latency forms of the n^2 domain (hensel_wave_n2.hpp: one wavefront per element on pair rows of this form) for a launch
// of `count` elements?  They run with 32-bit quotient digits (values below 9 P instead of 2 P): the rows' radix must leave
// room, R = 2^(29 L2) >= 2^10 P with P = n k < 2^(bits(n) + 29) -- true for every key size that has these rows but its
// largest two or three bit lengths.
bool wave_n2_applies(const pgpu_pubkey::PubForm* form, size_t count) {
  const int L2 = form->H * form->K;
  return hensel_enabled() && pair_rows_enabled() && pgpu::hensel_modexp_wave_has(L2) && policy::modexp_wave_form_pays(count) &&
         pgpu::kLimbBits * L2 - (form->n.BitSize() + pgpu::kLimbBits) >= 10;
}
int modexp_split_on(rt::Device& d, const pgpu_pubkey* key, const pgpu_pubkey::PubForm* form, const uint64_t* d_base, size_t base_stride,
                    int base_words, bool base_mont, const uint64_t* d_exp, size_t exp_stride, int exp_words,
                    int exp_bits, const SchedRef* sched, int final_mul, const uint64_t* d_m, size_t m_stride,
                    int m_words, uint64_t* d_out, bool out_mont, size_t count, hipStream_t s,
                    const uint32_t* base_pair = nullptr, size_t base_pair_stride = 0, uint32_t* out_pair = nullptr) {
  const int H = form->H, K = form->K;
  pgpu::HenselModexpArgs a{};
  a.base_pair = base_pair;
  a.base_pair_stride = base_pair_stride;
  a.out_pair = out_pair;
  a.ctx = hensel_pub_view(form, d.index, base_mont);
  a.full = hensel_full_view(key, form, d.index, out_mont);
  a.base = d_base;
  a.base_stride = base_stride;
  a.base_words = base_words;
  a.chunk_words = form->chunk_words;
  a.nchunks = (base_words + form->chunk_words - 1) / form->chunk_words;
  a.exp = d_exp;
  a.exp_stride = exp_stride;
  a.exp_words = exp_words;
  a.exp_bits = exp_bits;
  size_t entries;
  if (sched && sched->p[0]) {
    a.sched = sched->p[0];
    a.sched_len = sched->len[0];
    a.window = sched->w;
    entries = (size_t)1 << (a.window - 1);
  } else {
    a.window = pick_window(exp_bits);
    entries = (size_t)1 << a.window;
  }
  a.final_mul = final_mul;
  a.ct_gather = (sched && sched->p[0]) ? 0 : g_ct_gather.load();   // (a host-built schedule belongs to a PUBLIC exponent)
  a.fm_words = d_m;
  a.fm_stride = m_stride;
  a.fm_nwords = m_words;
  a.out = d_out;
  a.out_stride = (size_t)2 * key->n_words;
  a.count = count;
  // The latency form (hensel_wave_n2.hpp; round 6): small launches on resident rows -- one wavefront per element, the window
  // table in its LDS
  if (base_pair && out_pair && !a.sched && final_mul == pgpu::FM_UNIT && wave_n2_applies(form, count) && a.window <= 5 &&
      (policy::wave_policy() == 2 || count * (size_t)(1 + wave_neighbours(d, s)) <= kSimds)) {
    TimerScope t(d, s, PGPU_KERNEL_MODEXP, PGPU_FORM_WAVE);
    if (!pgpu::launch_hensel_modexp_wave(H * K, a, s))
      return fail(PGPU_ERR_UNSUPPORTED, "wavefront-wide modexp kernel not compiled");
    HIP_TRY(hipGetLastError());
    t.stop();
    return PGPU_OK;
  }
  // both halves of a residue in the same lanes (hensel_seq.hpp) when the launch still puts a wavefront on every SIMD
  // that way: resident rows in and out, per-element exponents
  const bool seq = modexp_seq_form_pays(H, K, count) && base_pair && out_pair && !a.sched && final_mul == pgpu::FM_UNIT;
  const size_t ipw = seq ? 64 / (size_t)H : 64 / (2 * (size_t)H);
  const size_t waves = (count + ipw - 1) / ipw;
  const unsigned blocks = (unsigned)((waves + pgpu::kWavesPerWG - 1) / pgpu::kWavesPerWG);
  rt::StreamWork& w = d.work_for(s);
  std::lock_guard<std::mutex> lk(w.mu);
  RC_TRY(w.table.ensure((size_t)blocks * pgpu::kWavesPerWG * ipw * entries * 2 * H * K * sizeof(uint32_t), s));
  a.table = (uint32_t*)w.table.p;
  TimerScope t(d, s, PGPU_KERNEL_MODEXP, seq ? PGPU_FORM_SEQ : PGPU_FORM_PAIRED);
  if (seq) {
    if (!pgpu::launch_hensel_modexp_seq(H, K, a, blocks, s))
      return fail(PGPU_ERR_UNSUPPORTED, "sequential-halves modexp kernel not compiled");
  } else if (!pgpu::launch_hensel_modexp(H, K, a, blocks, s)) return fail(PGPU_ERR_UNSUPPORTED, "split-form modexp kernel not compiled");
  HIP_TRY(hipGetLastError());
  t.stop();
  return PGPU_OK;
}
// the fixed-base table of pairs (hensel.hpp: hensel_fb_build_kernel); same size as the full-width one
int fb_table_for_split(const pgpu_pubkey* key, const pgpu_pubkey::PubForm* form, rt::Device& d, int w, int nwin,
                       hipStream_t s, FbPin* out) {
  const int H = form->H, K = form->K;
  return fb_table_get(key, key->fbh, d, w, nwin, (size_t)2 * H * K * sizeof(uint32_t), s, out, [&](FbTable& t) -> int {
    pgpu::HenselFbBuildArgs b{};
    b.ctx = hensel_pub_view(form, d.index);
    b.base = (const uint64_t*)key->d_hs.d[(size_t)d.index];
    b.base_words = 2 * key->n_words;
    b.chunk_words = form->chunk_words;
    b.nchunks = form->nchunks;
    b.table = (uint32_t*)t.p;
    b.nwin = nwin;
    b.w = w;
    const int ipw = 64 / (2 * H);
    const unsigned blocks = (unsigned)((((size_t)nwin + ipw - 1) / ipw + pgpu::kWavesPerWG - 1) / pgpu::kWavesPerWG);
    if (!pgpu::launch_hensel_fb_build(H, K, b, blocks, s))
      return fail(PGPU_ERR_UNSUPPORTED, "split-form fixed-base build kernel not compiled");
    return PGPU_OK;
  });
}

int encrypt_on(rt::Device& d, const pgpu_pubkey* key, const uint64_t* d_m, size_t m_stride, int m_words,
               const uint64_t* d_r, size_t r_stride, int r_words, int r_bits, uint64_t* d_c, size_t count,
               hipStream_t s, bool out_mont, size_t total_count, uint32_t* d_pair = nullptr, int busy_lanes = 0) {
  // d_pair: the ciphertexts leave as pair rows (resident batches; the caller has checked that the key has a pair form,
  // that the plaintext rows are no wider than n and that the obfuscator runs through a split-form kernel)
  const int W = 2 * key->n_words;
  if (m_words <= 0 || m_words > W || m_stride < (size_t)m_words)
    return fail(PGPU_ERR_INVALID_PARAM, "plaintext width/stride invalid");
  if (r_words <= 0 || r_stride < (size_t)r_words)
    return fail(PGPU_ERR_INVALID_PARAM, "random width/stride invalid");
  if (key->djn && (r_bits < 0 || r_bits > 64 * r_words))
    return fail(PGPU_ERR_INVALID_PARAM, "r_bits/r_words inconsistent");
  const int vflags = out_mont ? VF_GM_MONT : VF_NONE;
  int fbw = fixed_base_window();
  // Masked gather (pgpu_set_table_gather_policy(1); round 4): the fixed-base product runs over a table of its own with a
  // SMALL window -- every one of its 2^w entries is read at each step and the wanted one selected, so the address stream
  // does not depend on the digits of r (what the reference's mbx_exp_mb8 does with its window table, mod_exp.cpp:508-516).
  // w = 4: 256 products of 16 candidates each for a 1024-bit r instead of 85 indexed ones (PGPU_FB_MASKED_WINDOW: 1..5).
  const bool masked = key->djn && fbw > 0 && g_ct_gather.load() != 0;
  if (masked) fbw = masked_fb_window();
  if (key->djn && fbw > 0) {
    if (!masked) {
      std::lock_guard<std::mutex> lk(key->mu);
      if (fbw > 8 && !g_fb_window_explicit.load() && key->fb_elems + total_count < kFbGrowAfter) fbw = 8;
    }
    // hs is a key constant: fixed-base windowing, no squarings (kernels.hpp: fb_encrypt_kernel)
    const GeoInfo& geo = key->nsq->geo;
    const pgpu_pubkey::PubForm* sform = d_pair ? pair_form(key) : use_split_encrypt(key, m_words, count);
    // the per-key / per-GPU table limits may narrow the window (fb_fit_window)
    fbw = fb_fit_window(fbw, r_bits, sform ? (size_t)2 * sform->H * sform->K * sizeof(uint32_t) : (size_t)geo.L() * sizeof(uint32_t));
    const int nwin = std::max(1, (r_bits + fbw - 1) / fbw);
    FbPin tab;
    // split form (hensel.hpp): pairs modulo (n*k)^2; needs m < 2n to form the pair of 1 + n*m, i.e. plaintext rows
    // no wider than n, and a batch that fills the chip in 8-lane groups no worse than the full-width kernel does
    if (const pgpu_pubkey::PubForm* form = sform) {
      RC_TRY(fb_table_for_split(key, form, d, fbw, nwin, s, &tab));
      // pair-row output of a batch that leaves SIMDs idle: the encrypt kernel of a form with more lanes per element and
      // the SAME limbs per half (2048-bit keys: (8,9) beside (4,18)) -- same table, same rows, shorter serial chain
      if (d_pair) {
        for (const auto& alt : key->hforms) {
          const size_t ipw_alt = 64 / (2 * (size_t)alt->H);
          if (alt->H * alt->K == form->H * form->K && alt->H > form->H && pgpu::hensel_fb_encrypt_has(alt->H, alt->K) &&
              (total_count + ipw_alt - 1) / ipw_alt <= kSimds) {
            form = alt.get();
            break;
          }
        }
      }
      pgpu::HenselFbArgs f{};
      f.ctx = hensel_pub_view(form, d.index);
      f.full = hensel_full_view(key, form, d.index, out_mont);
      f.table = (const uint32_t*)tab->p;
      f.nwin = nwin;
      f.w = fbw;
      f.exp = d_r;
      f.exp_stride = r_stride;
      f.exp_words = r_words;
      f.fm_words = d_m;
      f.fm_stride = m_stride;
      f.fm_nwords = m_words;
      f.out = d_c;
      f.out_stride = (size_t)W;
      f.count = count;
      f.out_pair = d_pair;
      f.ct_gather = masked ? 1 : 0;
      TimerScope t(d, s, PGPU_KERNEL_FB_ENCRYPT);
      // The latency form (hensel_wave_n2.hpp; round 6): small launches onto pair rows -- one wavefront per element
      if (d_pair && busy_lanes == 0 && wave_n2_applies(form, count) &&
          (policy::wave_policy() == 2 || count * (size_t)(1 + wave_neighbours(d, s)) <= kSimds)) {
        t.set_form(PGPU_FORM_WAVE);
        if (!pgpu::launch_hensel_fb_encrypt_wave(form->H * form->K, f, s))
          return fail(PGPU_ERR_UNSUPPORTED, "wavefront-wide fixed-base kernel not compiled");
        HIP_TRY(hipGetLastError());
        t.stop();
        return PGPU_OK;
      }
      // resident results of launches that still put a wavefront on every SIMD with half the lanes per element: both
      // halves of a residue in the same lanes (hensel_seq.hpp)
      const bool seq = d_pair && fb_encrypt_seq_pays(form->H, form->K, count, busy_lanes);
      t.set_form(seq ? PGPU_FORM_SEQ : PGPU_FORM_PAIRED);
      const int ipw = seq ? 64 / form->H : 64 / (2 * form->H);
      const unsigned blocks = (unsigned)(((count + ipw - 1) / ipw + pgpu::kWavesPerWG - 1) / pgpu::kWavesPerWG);
      if (seq) {
        // a part-chip launch beside busy neighbour lanes claims whole CUs, like the decrypt it feeds (decrypt_on)
        const size_t seq_waves = (count + ipw - 1) / ipw;
        // beside ONE busy lane: one workgroup per CU, half the chip each (round 4).  Beside two or three (round 5): the
        // workgroup owns 80 000 bytes in all -- two of them share a CU (two wavefronts per SIMD on a quarter of the chip),
        // but none fits beside a neighbour's decrypt workgroup (84 000 bytes claimed): an encrypt wavefront that shares a
        // SIMD with an older decrypt wavefront only gets the issue slots that one leaves (measured: 0.8 -> 10 ms)
        unsigned lds_pad = adaptive_cu_claim(seq_waves, busy_lanes);
        if (lds_pad && busy_lanes >= 3) lds_pad = pgpu::kLdsTotalFlag | 80000u;
        if (lds_pad) t.set_form(PGPU_FORM_SEQ | PGPU_FORM_CU_CLAIM);
        if (!pgpu::launch_hensel_fb_encrypt_seq(form->H, form->K, f, blocks, s, lds_pad))
          return fail(PGPU_ERR_UNSUPPORTED, "sequential-halves fixed-base kernel not compiled");
      } else if (!pgpu::launch_hensel_fb_encrypt(form->H, form->K, f, blocks, s))
        return fail(PGPU_ERR_UNSUPPORTED, "split-form fixed-base kernel not compiled");
      HIP_TRY(hipGetLastError());
      t.stop();
      return PGPU_OK;
    }
    RC_TRY(fb_table_for(key, d, fbw, nwin, s, &tab));
    pgpu::FixedBaseArgs f{};
    f.ctx = key->nsq->view(d.index, vflags);
    f.table = (const uint32_t*)tab->p;
    f.nwin = nwin;
    f.w = fbw;
    f.exp = d_r;
    f.exp_stride = r_stride;
    f.exp_words = r_words;
    f.fm_words = d_m;
    f.fm_stride = m_stride;
    f.fm_nwords = m_words;
    f.out = d_c;
    f.out_stride = (size_t)W;
    f.count = count;
    f.ct_gather = masked ? 1 : 0;
    TimerScope t(d, s, PGPU_KERNEL_FB_ENCRYPT);
    const GeoInfo lgeo = launch_geo(geo, count);
    if (!pgpu::launch_fb_encrypt(lgeo.G, lgeo.K, f, blocks_for(count, lgeo), s))
      return fail(PGPU_ERR_UNSUPPORTED, "fixed-base kernel geometry not compiled");
    HIP_TRY(hipGetLastError());
    t.stop();
    return PGPU_OK;
  }
  pgpu::ModexpArgs a{};
  a.ctx[0] = a.ctx[1] = key->nsq->view(d.index, vflags);
  a.nctx = 1;
  if (key->djn) {  // hs^r: shared base, per-element exponent (pub_key.cpp:51-64)
    a.base = (const uint64_t*)key->d_hs.d[(size_t)d.index];
    a.base_stride = 0;
    a.base_words = W;
    a.exp = d_r;
    a.exp_stride = r_stride;
    a.exp_words = r_words;
    a.exp_bits = r_bits;
  } else {         // r^n: per-element base, shared exponent n (pub_key.cpp:66-80)
    if (r_words > W) return fail(PGPU_ERR_INVALID_PARAM, "random wider than n^2");
    const pgpu_pubkey::PubForm* form = d_pair ? pair_form(key)
                                       : (64 * m_words <= key->n.BitSize() ? split_modexp_form(key, count) : nullptr);
    if (form) {
      SchedRef srn;
      if (key->sched_n.dev.bytes) {
        srn.p[0] = (const uint16_t*)key->sched_n.dev.d[(size_t)d.index];
        srn.len[0] = key->sched_n.len;
        srn.w = key->sched_n.w;
      }
      return modexp_split_on(d, key, form, d_r, r_stride, r_words, false,
                             (const uint64_t*)key->d_n.d[(size_t)d.index], 0, key->n_words, key->n.BitSize(),
                             srn.p[0] ? &srn : nullptr, pgpu::FM_PAILLIER_G, d_m, m_stride, m_words, d_c, out_mont,
                             count, s, nullptr, 0, d_pair);
    }
    a.base = d_r;
    a.base_stride = r_stride;
    a.base_words = r_words;
    a.exp = (const uint64_t*)key->d_n.d[(size_t)d.index];
    a.exp_stride = 0;
    a.exp_words = key->n_words;
    a.exp_bits = key->n.BitSize();
  }
  if (d_pair) return fail(PGPU_ERR_UNSUPPORTED, "pair-row output needs a split-form encrypt kernel");
  a.exp_per_ctx = 0;
  a.final_mul = pgpu::FM_PAILLIER_G;
  a.fm_words = d_m;
  a.fm_stride = m_stride;
  a.fm_nwords = m_words;
  a.out = d_c;
  a.out_stride = (size_t)W;
  a.count = count;
  SchedRef sr;
  if (!key->djn && key->sched_n.dev.bytes) {     // r^n: the exponent is the PUBLIC key constant n
    sr.p[0] = (const uint16_t*)key->sched_n.dev.d[(size_t)d.index];
    sr.len[0] = key->sched_n.len;
    sr.w = key->sched_n.w;
  }
  return run_modexp(d, a, key->nsq->geo, s, sr.p[0] ? &sr : nullptr);
}

// split form of the CRT-decrypt exponentiation for a batch of `count` ciphertexts on one device (null: the
// full-width modexp_kernel)
const pgpu_privkey::HenselSet* pick_hensel(const pgpu_privkey* key, size_t count) {
  if (!hensel_enabled() || key->hs.empty()) return nullptr;
  const int mode = g_hensel.load();   // 1: by batch size; 2 / 3 (tests): always the form of fewest / most lanes
  if (mode == 2) return key->hs.back().get();
  if (mode == 3) return key->hs.front().get();
  for (const auto& f : key->hs) {
    const size_t ipw = 64 / (2 * (size_t)f->H);
    if (2 * ((count + ipw - 1) / ipw) <= kSimds) return f.get();
  }
  return key->hs.back().get();
}

// fused CRT decrypt on one device; in_mont: ciphertexts arrive in the Montgomery domain of n^2
// Batch lanes of `dev` other than `lane` that are ACTIVE: work queued right now, or fed within the last few tens of
// milliseconds (PGPU_LANE_ACTIVE_MS, default 50).  The second clause is what keeps a caller that rotates over the lanes
// in one mode: right after a synchronisation every lane is empty for a moment, and a policy that only looked at the
// queues would start each burst with a full-chip launch that the next lane's half-chip launch then has to share CUs
// with (measured: ~7 ms lost at the head of a 20-step run, 5.31 instead of 4.95 ms per step).  Also stamps `lane`.
// `count`: threads on round-robin lanes with launches under policy::kRrAdaptMinCount elements keep the lone caller's forms
// (stamped all the same): since the placement pad (launch.hpp) spreads small launches over the CUs those beat the part-chip
// forms -- four threads x 64 elements 0.76 against 1.03 ms per encrypt + decrypt, x 256 0.97 against 1.05, x 2048 2.18 against
// 2.26, level at 4096, 5.2 against 5.8 at 8192 (profiles/r06_place_pad.txt)
int busy_other_lanes(rt::Device& dev, int lane, bool force = false, size_t count = (size_t)-1) {
  const int rr_min = (!force && !t_lane_explicit) ? policy::rr_adapt() : 1;
  if (rr_min <= 0) return 0;   // (synchronous callers on round-robin lanes: lone-caller forms, no stamp)
  static const int64_t window_ns = [] {
    const char* e = std::getenv("PGPU_LANE_ACTIVE_MS");
    return (int64_t)(e ? std::max(0, std::atoi(e)) : 50) * 1000000;
  }();
  const int64_t now = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
  dev.lane_fed_ns[lane % rt::kBatchLanes].store(now, std::memory_order_relaxed);
  int busy = 0;
  for (int k = 0; k < rt::kBatchLanes; ++k) {
    if (k == lane) continue;
    const int64_t fed = dev.lane_fed_ns[k].load(std::memory_order_relaxed);
    if ((fed != 0 && now - fed < window_ns) || hipStreamQuery(dev.bs(k)) == hipErrorNotReady) ++busy;
  }
  (void)hipGetLastError();
  if (!force && !t_lane_explicit && count < policy::kRrAdaptMinCount) return 0;
  return busy >= rr_min ? busy : 0;
}
// The same for a synchronous caller of the HOST-ARRAY entry points (pgpu_paillier_encrypt / _decrypt_crt) on its thread's
// lane.  PGPU_HOST_ADAPT=1: the full adaptive policy (measured slower with one neighbour, kept for A/B).  Otherwise, round 5:
// the callers see each other through stamps of their own, and only when at least PGPU_RR_ADAPT (3) others have been calling
// within the activity window do their launches take the quarter-chip forms -- four callers side by side, as four API threads.
int host_busy(rt::Device& dev, int lane, size_t count = (size_t)-1) {
  if (host_adapt()) return busy_other_lanes(dev, lane, true);
  const int k = policy::rr_adapt();
  if (k <= 0) return 0;
  static const int64_t window_ns = [] {
    const char* e = std::getenv("PGPU_LANE_ACTIVE_MS");
    return (int64_t)(e ? std::max(0, std::atoi(e)) : 50) * 1000000;
  }();
  const int64_t now = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
  dev.host_fed_ns[lane % rt::kBatchLanes].store(now, std::memory_order_relaxed);
  int busy = 0;
  for (int j = 0; j < rt::kBatchLanes; ++j) {
    if (j == lane % rt::kBatchLanes) continue;
    const int64_t fed = std::max(dev.host_fed_ns[j].load(std::memory_order_relaxed), dev.lane_fed_ns[j].load(std::memory_order_relaxed));
    if (fed != 0 && now - fed < window_ns) ++busy;
  }
  if (count < policy::kRrAdaptMinCount) return 0;   // (busy_other_lanes: small launches keep the lone caller's forms)
  return busy >= k ? busy : 0;
}

// the one-lane product-scanning form (csrc/hensel_ps.hpp) for a decrypt of `count` resident ciphertexts under this key?
bool ps_form_pays(const pgpu_privkey* key, size_t count, int busy) {
  return key->hs_ps && hensel_enabled() && policy::ps_form_pays(count, busy, key->hs_ps->K);
}
// the latency form on the same constants (csrc/hensel_wave.hpp: one exponentiation per wavefront) for a small lone decrypt?
bool wave_form_pays(const pgpu_privkey* key, size_t count, int busy) {
  return key->hs_ps && hensel_enabled() && pgpu::hensel_wave_has(key->hs_ps->K, key->hs_ps->lb) && policy::wave_form_pays(count, busy);
}
int words_to_pair_on(rt::Device& d, const pgpu_pubkey::PubForm* f, const uint64_t* words, size_t stride, int nwords,
                     bool src_mont, uint32_t* out, size_t count, hipStream_t s);
int decrypt_on(rt::Device& d, const pgpu_privkey* key, const uint64_t* d_c, uint64_t* d_m, size_t count,
               hipStream_t s, bool in_mont, const uint32_t* d_pair = nullptr, int in_pair_l2 = 0, int busy_lanes = 0) {
  // d_pair: the ciphertexts are pair rows of 2*in_pair_l2 limbs (d_c unused); needs a split form of the key
  const int nw = key->n_words;
  // A lone launch of more than a round of the one-lane product-scanning form whose last round would be mostly empty: the
  // full rounds first, the rest as a launch of its own in the form ITS size takes (policy.hpp: ps_split_head)
  if (busy_lanes == 0 && key->hs_ps && hensel_enabled() && secret_policy() != PGPU_EXP_SLIDING &&
      (d_pair ? key->hs_ps->pair_l2 == in_pair_l2 : key->conv_form != nullptr)) {
    if (const size_t head = policy::ps_split_head(key->hs_ps->K, count)) {
      RC_TRY(decrypt_on(d, key, d_c, d_m, head, s, in_mont, d_pair, in_pair_l2, busy_lanes));
      return decrypt_on(d, key, d_c ? d_c + head * (size_t)2 * nw : nullptr, d_m + head * (size_t)nw, count - head, s, in_mont,
                        d_pair ? d_pair + head * (size_t)2 * in_pair_l2 : nullptr, in_pair_l2, busy_lanes);
    }
  }
  rt::StreamWork& w = d.work_for(s);
  std::lock_guard<std::mutex> lk(w.mu);   // the hand-over buffer is ours until both stages are queued
  RC_TRY(w.vbuf.ensure(2 * count * (size_t)nw * 8, s));
  const bool lat = use_latency_geo(key->geo_lat, key->geo_exp, 2 * count);
  const bool sliding = secret_policy() == PGPU_EXP_SLIDING && key->sched[0].dev.bytes;
  bool have_m = false;
  const pgpu_privkey::HenselSet* hset = pick_hensel(key, count);
  // Word ciphertexts (host arrays, uploaded batches) become pair rows first when the launch then takes a kernel that reads
  // only those: the one-lane form, the sequential-halves form by size, or -- beside busy neighbours -- the half-chip
  // launch of the adaptive policy.  One pair_ops_kernel pass (a product per chunk), ~1 % of the exponentiation.
  rt::DevMem conv_rows;
  if (!d_pair && !sliding && hset && key->conv_form) {
    const pgpu_privkey::HenselSet* cand = hset;
    const int cl2 = key->conv_form->H * key->conv_form->K;
    if (cand->pair_l2 == cl2 && (seq_form_pays(cand->H, cand->K, count, busy_lanes) || ps_form_pays(key, count, busy_lanes) ||
                                 wave_form_pays(key, count, busy_lanes))) {
      RC_TRY(conv_rows.alloc(d, s, count * (size_t)2 * cl2 * sizeof(uint32_t)));
      RC_TRY(words_to_pair_on(d, key->conv_form.get(), d_c, (size_t)2 * nw, 2 * nw, in_mont, (uint32_t*)conv_rows.p, count, s));
      d_pair = (const uint32_t*)conv_rows.p;
      in_pair_l2 = cl2;
    }
  }
  const bool wavef = d_pair && !sliding && hset && wave_form_pays(key, count, busy_lanes) && key->hs_ps->pair_l2 == in_pair_l2 &&
                     (policy::wave_policy() == 2 || 2 * count * (size_t)(1 + wave_neighbours(d, s)) <= kSimds);
  const bool psf = wavef || (d_pair && !sliding && hset && ps_form_pays(key, count, busy_lanes) &&
                             key->hs_ps->pair_l2 == in_pair_l2);
  if (psf) hset = key->hs_ps.get();
  if (d_pair && (!hset || hset->pair_l2 != in_pair_l2))
    return fail(PGPU_ERR_UNSUPPORTED, "decrypt: pair-row ciphertexts need the split-form kernel of this key size");
  if (hset) {
    // stage 1, split form: M[2i] = mp, M[2i+1] = mq  (hensel.hpp)
    const int L2 = hset->H * hset->K, nch = hset->nchunks, ipw = 64 / (2 * hset->H);
    const size_t side_words = hset->side_words();
    const uint32_t* blob = (const uint32_t*)hset->blob.d[(size_t)d.index];
    pgpu::HenselArgs h{};
    for (int sd = 0; sd < 2; ++sd) {
      const uint32_t* b = blob + sd * side_words;
      h.ctx[sd].nhat = b;
      h.ctx[sd].n = b + L2;
      h.ctx[sd].h = b + 2 * L2;
      h.ctx[sd].kr = b + 3 * L2;
      h.ctx[sd].one = b + 4 * L2;
      h.ctx[sd].conv = b + 6 * L2 + (in_mont ? (size_t)nch * 2 * L2 : 0);
      h.ctx[sd].n0inv = hset->n0inv[sd];
    }
    h.ct = d_c;
    h.ct_stride = (size_t)2 * nw;
    h.ct_words = 2 * nw;
    if (d_pair) {
      h.ct_pair = d_pair;
      h.ct_pair_stride = (size_t)2 * in_pair_l2;
      h.pair_l2 = in_pair_l2;
      h.pchunk_limbs = hset->pchunk_limbs;
      h.pchunks = hset->pchunks;
      for (int sd = 0; sd < 2; ++sd) {
        const uint32_t* pc = blob + sd * side_words + (size_t)L2 * (6 + 4 * (size_t)nch);
        h.ctx[sd].pconv = pc;
        h.ctx[sd].pcb = pc + (size_t)hset->pchunks * 2 * L2;
      }
    }
    h.chunk_words = hset->chunk_words;
    h.nchunks = nch;
    h.exp = (const uint64_t*)key->d_exps.d[(size_t)d.index];
    h.exp_stride = (size_t)key->pq_words;
    h.exp_words = key->pq_words;
    h.exp_bits = key->exp_bits;
    size_t entries;
    if (sliding) {
      for (int i = 0; i < 2; ++i) {
        h.sched[i] = (const uint16_t*)key->sched[i].dev.d[(size_t)d.index];
        h.sched_len[i] = key->sched[i].len;
      }
      h.window = key->sched[0].w;
      entries = (size_t)1 << (h.window - 1);
    } else {
      h.window = g_ct_gather.load() ? std::min(masked_decrypt_window(), pick_window(h.exp_bits)) : policy::pick_decrypt_window(h.exp_bits, 2 * count * (size_t)2 * L2 * sizeof(uint32_t));
      if (wavef) h.window = std::min(h.window, 5);      // (its table is LDS: 32 entries per wavefront at most)
      entries = (size_t)1 << h.window;
    }
    h.ct_gather = g_ct_gather.load();
    h.out = (uint64_t*)w.vbuf.p;
    h.out_stride = (size_t)key->pq_words;
    h.out_words = key->pq_words;
    h.count = count;
    const size_t waves = 2 * ((count + ipw - 1) / ipw);
    const unsigned blocks = (unsigned)((waves + pgpu::kWavesPerWG - 1) / pgpu::kWavesPerWG);
    // (the window-table workspace is sized by the form that runs: one allocation per launch at most -- a growing
    // workspace is an allocation of a few hundred MB, milliseconds of host time when the arena has to go to the driver)
    TimerScope t(d, s, PGPU_KERNEL_MODEXP);
    const size_t seq_ipw = 64 / (size_t)hset->H;
    const size_t seq_waves = 2 * ((count + seq_ipw - 1) / seq_ipw);
    const bool seq = !psf && d_pair && !sliding && seq_form_pays(hset->H, hset->K, count, busy_lanes);
    t.set_form(psf ? PGPU_FORM_LANE | PGPU_FORM_PS : seq ? PGPU_FORM_SEQ : PGPU_FORM_PAIRED);
    if (wavef) {
      // the latency form: entry (one lane per exponentiation) -> one wavefront per exponentiation -> exit; the workspace
      // is the pair buffer between the three launches, the window table lives in the wave kernel's LDS
      t.set_form(PGPU_FORM_WAVE | PGPU_FORM_PS);
      RC_TRY(w.table.ensure(2 * count * (size_t)(2 * hset->pchunks) * pgpu::hensel_wave_pair_words(hset->K) * sizeof(uint32_t), s));   // (a partial pair per entry role)
      h.table = (uint32_t*)w.table.p;
      // 32-bit quotient digits (one instruction less per step and scan) where the radix leaves room: values then stay below
      // 17 P instead of 2 P, which needs R = 2^(lb K) >= 2^10 P; P = p k < 2^(exp_bits + lb)  (3072-bit keys: R >= 16 P only)
      static const bool wide_ok = [] { const char* e = getenv("PGPU_WAVE_WIDEQ"); return !e || atoi(e) != 0; }();
      const bool wide = wide_ok && hset->lb * hset->K - (key->exp_bits + hset->lb) >= 10;
      if (!pgpu::launch_hensel_wave(hset->K, hset->lb, wide, h, s))
        return fail(PGPU_ERR_UNSUPPORTED, "wavefront-wide decrypt kernel not compiled");
    } else if (psf) {
      const size_t lwaves = 2 * ((count + 63) / 64);
      const unsigned lblocks = (unsigned)((lwaves + pgpu::kWavesPerWG - 1) / pgpu::kWavesPerWG);
      RC_TRY(w.table.ensure((size_t)lblocks * pgpu::kWavesPerWG * pgpu::hensel_ps_table_words(hset->K, entries) * sizeof(uint32_t), s));
      h.table = (uint32_t*)w.table.p;
      // a part-chip launch beside busy neighbour lanes claims whole CUs (one workgroup per CU: the launches of the lanes
      // spread over the chip, one wavefront per SIMD each)
      const unsigned lds_pad = adaptive_cu_claim(lwaves, busy_lanes);
      if (lds_pad) t.set_form(PGPU_FORM_LANE | PGPU_FORM_PS | PGPU_FORM_CU_CLAIM);
      if (!pgpu::launch_hensel_ps(hset->K, hset->lb, h, lblocks, s, lds_pad))
        return fail(PGPU_ERR_UNSUPPORTED, "one-lane product-scanning decrypt kernel not compiled");
    } else if (seq) {
      const unsigned sblocks = (unsigned)((seq_waves + pgpu::kWavesPerWG - 1) / pgpu::kWavesPerWG);
      RC_TRY(w.table.ensure((size_t)sblocks * pgpu::kWavesPerWG * seq_ipw * entries * 2 * L2 * sizeof(uint32_t), s));
      h.table = (uint32_t*)w.table.p;
      static const int env_pad = [] {
        const char* e = getenv("PGPU_SEQ_LDS_PAD");
        return e ? atoi(e) : -1;
      }();
      // a launch that covers less than the chip leaves the other CUs to the neighbour lane's launch (policy 3)
      const int pol = policy::seq_policy();
      const unsigned lds_pad = env_pad >= 0 ? (unsigned)env_pad
                               : (seq_waves < kSimds && (pol == 3 || (pol == 4 && busy_lanes >= 1 && busy_lanes <= policy::adapt_claim_busy())) ? 84000u : 0u);
      if (lds_pad) t.set_form(PGPU_FORM_SEQ | PGPU_FORM_CU_CLAIM);
      static const bool w1 = [] { const char* e = getenv("PGPU_SEQ_W1"); return !e || atoi(e) != 0; }();
      if (!pgpu::launch_hensel_seq(hset->H, hset->K, h, sblocks, s, lds_pad, w1 && lds_pad >= 82000u))
        return fail(PGPU_ERR_UNSUPPORTED, "sequential-halves decrypt kernel not compiled");
    } else {
      RC_TRY(w.table.ensure((size_t)blocks * pgpu::kWavesPerWG * ipw * entries * 2 * L2 * sizeof(uint32_t), s));
      h.table = (uint32_t*)w.table.p;
      if (!pgpu::launch_hensel(hset->H, hset->K, waves > kSimds || g_packed_decrypt.load(), h, blocks, s))
        return fail(PGPU_ERR_UNSUPPORTED, "split-form kernel not compiled");
    }
    HIP_TRY(hipGetLastError());
    t.stop();
    have_m = true;
  }
  // stage 1: V[2i] = c^(p-1)*hp mod p^2, V[2i+1] = c^(q-1)*hq mod q^2   (2*count instances)
  pgpu::ModexpArgs a{};
  const int vf = in_mont ? VF_BASE_MONT : VF_NONE;
  a.ctx[0] = (lat ? key->p2l : key->p2)->view(d.index, vf);
  a.ctx[1] = (lat ? key->q2l : key->q2)->view(d.index, vf);
  a.nctx = 2;
  a.base = d_c;
  a.base_stride = (size_t)2 * nw;
  a.base_words = 2 * nw;
  a.exp = (const uint64_t*)key->d_exps.d[(size_t)d.index];
  a.exp_stride = (size_t)key->pq_words;
  a.exp_per_ctx = 1;
  a.exp_words = key->pq_words;
  a.exp_bits = key->exp_bits;
  a.final_mul = pgpu::FM_CTX_CONST;
  a.out = (uint64_t*)w.vbuf.p;
  a.out_stride = (size_t)nw;
  a.count = 2 * count;
  SchedRef sr;
  if (sliding) {
    for (int i = 0; i < 2; ++i) {
      sr.p[i] = (const uint16_t*)key->sched[i].dev.d[(size_t)d.index];
      sr.len[i] = key->sched[i].len;
    }
    sr.w = key->sched[0].w;
  }
  if (!have_m) RC_TRY(run_modexp(d, a, lat ? key->geo_lat : key->geo_exp, s, sliding ? &sr : nullptr, &w));
  // stage 2: L function, CRT
  const int Lc = key->geo_crt.L(), pad = key->geo_crt.w64() + 1;
  pgpu::CrtArgs c{};
  c.ctxM = key->cM->view(d.index);
  c.ctxQ = key->cQ->view(d.index);
  const uint32_t* c32 = (const uint32_t*)key->d_crt32.d[(size_t)d.index];
  c.cp = c32;
  c.cq = c32 + Lc;
  c.pinvR = c32 + 2 * Lc;
  c.pRM = c32 + 3 * Lc;
  const uint64_t* c64 = (const uint64_t*)key->d_crt64.d[(size_t)d.index];
  c.hp64 = c64;
  c.hq64 = c64 + pad;
  c.p2_64 = c64 + 2 * pad;
  c.q2_64 = c64 + 3 * pad;
  c.q64 = c64 + 4 * pad;
  c.v = (const uint64_t*)w.vbuf.p;
  c.vw = have_m ? key->pq_words : nw;
  c.have_m = have_m ? 1 : 0;
  c.out = d_m;
  c.out_words = nw;
  c.count = count;
  TimerScope tc(d, s, PGPU_KERNEL_CRT);
  // beside three busy neighbour lanes (each lane owns a quarter of the chip) the recombination stays on the CUs its
  // decrypt has just left: 80 000 bytes per workgroup, as the encrypt of that mode (encrypt_on) -- 0.23 -> 0.07 ms.  Not
  // beside two: their half-chip decrypts hold every CU and the launch would wait for one (1.6 ms)
  const unsigned crt_claim = (busy_lanes >= 3 && adaptive_cu_claim(1, busy_lanes)) ? 80000u : 0u;
  if (!pgpu::launch_crt(key->geo_crt.G, key->geo_crt.K, c, blocks_for(count, key->geo_crt), s, crt_claim))
    return fail(PGPU_ERR_UNSUPPORTED, "crt kernel geometry not compiled");
  HIP_TRY(hipGetLastError());
  tc.stop();
  return PGPU_OK;
}

// ---------- pair rows: launches and conversions ----------
// the form an element-wise pair-row operation of `count` elements runs in: the key's pair form f, or -- while the
// launch leaves SIMDs idle -- a compiled form with the same limbs per half on more lanes (2048-bit keys: (8,9))
const pgpu_pubkey::PubForm* pair_op_form(const pgpu_pubkey* key, const pgpu_pubkey::PubForm* f, size_t count) {
  for (const auto& alt : key->hforms) {
    const size_t ipw = 64 / (2 * (size_t)alt->H);
    if (alt->H * alt->K == f->H * f->K && alt->H > f->H && pgpu::pair_ops_alt_has(alt->H, alt->K) &&
        (count + ipw - 1) / ipw <= kSimds)
      return alt.get();
  }
  return f;
}
int pair_op_launch(rt::Device& d, const pgpu_pubkey::PubForm* f, pgpu::PairOpsArgs& a, hipStream_t s, int kind) {
  // CT + CT of launches that still put a wavefront on every SIMD with half the lanes per element: both halves of a
  // residue in the same lanes (hensel_seq.hpp: pair_mul_seq_kernel)
  const bool seq = a.op == pgpu::PO_MUL && pair_mul_seq_pays(f->H, f->K, a.count);
  const size_t ipw = seq ? 64 / (size_t)f->H : 64 / (2 * (size_t)f->H);
  const size_t waves = (a.count + ipw - 1) / ipw;
  const unsigned blocks = (unsigned)((waves + pgpu::kWavesPerWG - 1) / pgpu::kWavesPerWG);
  TimerScope t(d, s, kind, seq ? PGPU_FORM_SEQ : PGPU_FORM_PAIRED);
  if (seq) {
    if (!pgpu::launch_pair_mul_seq(f->H, f->K, a, blocks, s)) return fail(PGPU_ERR_UNSUPPORTED, "pair-row kernel not compiled");
  } else if (!pgpu::launch_pair_ops(f->H, f->K, a, blocks, s)) return fail(PGPU_ERR_UNSUPPORTED, "pair-row kernel not compiled");
  HIP_TRY(hipGetLastError());
  t.stop();
  return PGPU_OK;
}
// one shard: 64-bit words (plain, or c*Rs mod n^2 when src_mont) -> pair rows
int words_to_pair_on(rt::Device& d, const pgpu_pubkey::PubForm* f, const uint64_t* words, size_t stride, int nwords,
                     bool src_mont, uint32_t* out, size_t count, hipStream_t s) {
  pgpu::PairOpsArgs a{};
  a.ctx = hensel_pub_view(f, d.index, src_mont);
  a.op = pgpu::PO_FROM_WORDS;
  a.words = words;
  a.words_stride = stride;
  a.nwords = nwords;
  a.chunk_words = f->chunk_words;
  a.nchunks = (nwords + f->chunk_words - 1) / f->chunk_words;
  a.out = out;
  a.count = count;
  return pair_op_launch(d, f, a, s, PGPU_KERNEL_MODMUL);
}
// one shard: pair rows -> canonical plain words
int pair_to_words_on(rt::Device& d, const pgpu_pubkey::PubForm* f, const uint32_t* rows, uint64_t* out, size_t count,
                     hipStream_t s, size_t out_stride = 0) {
  pgpu::PairOpsArgs a{};
  a.ctx = hensel_pub_view(f, d.index);
  a.full = hensel_full_view(nullptr, f, d.index, false);
  a.op = pgpu::PO_TO_WORDS;
  a.a = rows;
  a.out_words = out;
  a.out_stride = out_stride ? out_stride : (size_t)2 * f->n_words;
  a.count = count;
  return pair_op_launch(d, f, a, s, PGPU_KERNEL_MODMUL);
}
// a ciphertext batch of the key in pair rows: `a` itself when it already is one, else a converted copy held by *tmp
int as_pair_batch(const pgpu_pubkey* key, const pgpu_batch* a, const pgpu_batch** out, std::unique_ptr<pgpu_batch>* tmp) {
  const pgpu_pubkey::PubForm* f = pair_form(key);
  if (!f) return fail(PGPU_ERR_UNSUPPORTED, "key has no pair form");
  const int l2 = f->H * f->K;
  if (a->pair_l2) {
    if (a->pair_l2 != l2 || !(a->pair_form->n == key->n))
      return fail(PGPU_ERR_INVALID_PARAM, "ciphertext batch belongs to a different key");
    *out = a;
    return PGPU_OK;
  }
  std::unique_ptr<pgpu_batch> t;
  RC_TRY(new_batch(a->count, a->words, &t, l2, a->lane));
  t->pair_form = pair_form_shared(key);
  for (int d = 0; d < t->ndev; ++d) {
    size_t lo, hi;
    t->bounds(d, &lo, &hi);
    rt::Device& dev = rt::device(d);
    rt::DeviceGuard g(dev.ordinal);
    RC_TRY(words_to_pair_on(dev, f, a->ptr(d), (size_t)a->words, a->words, a->mont != nullptr, t->prow(d), hi - lo, dev.bs(a->lane)));
  }
  *tmp = std::move(t);
  *out = tmp->get();
  return PGPU_OK;
}
// a ciphertext batch as 64-bit words (plain): `a` itself unless it is in pair rows
int as_word_batch(const pgpu_batch* a, const pgpu_batch** out, std::unique_ptr<pgpu_batch>* tmp) {
  if (!a->pair_l2) {
    *out = a;
    return PGPU_OK;
  }
  std::unique_ptr<pgpu_batch> t;
  RC_TRY(new_batch(a->count, a->words, &t, 0, a->lane));
  for (int d = 0; d < t->ndev; ++d) {
    size_t lo, hi;
    t->bounds(d, &lo, &hi);
    rt::Device& dev = rt::device(d);
    rt::DeviceGuard g(dev.ordinal);
    RC_TRY(pair_to_words_on(dev, a->pair_form.get(), a->prow(d), t->ptr(d), hi - lo, dev.bs(a->lane)));
  }
  *tmp = std::move(t);
  *out = tmp->get();
  return PGPU_OK;
}

// ---- sharding of the host-pointer entry points ----
// fn(lane, lo, hi) handles elements [lo, hi) on lane.dev; sub_min: smallest sub-batch worth its own task
template <class F>
int run_sharded(size_t count, size_t sub_min, F fn) {
  rt::note_caller();
  const int D = rt::shard_devices(count);
  rt::TaskGroup tg;
  for (int d = 0; d < D; ++d) {
    size_t lo, hi;
    rt::shard_bounds(count, D, d, &lo, &hi);
    if (hi == lo) continue;
    const size_t n = hi - lo;
    const int subs = (int)std::max<size_t>(1, std::min<size_t>(4, n / std::max<size_t>(sub_min, 1)));
    for (int k = 0; k < subs; ++k) {
      size_t slo, shi;
      rt::shard_bounds(n, subs, k, &slo, &shi);
      slo += lo;
      shi += lo;
      tg.run(rt::device(d), [fn, slo, shi](rt::Lane& lane) { return fn(lane, slo, shi); });
    }
  }
  return tg.wait();
}
// sub-batches: long exponentiations keep their launches whole (a half-sized launch runs a slower geometry and
// the copies are < 5 % of the time); products and short exponents are copy-bound and pipeline in pieces
constexpr size_t kSubMinHeavy = (size_t)1 << 17, kSubMinLight = (size_t)1 << 14;

// objects of a pool that has been shut down are refused (their device images went with it)
int check_gen(uint64_t gen, const char* what) {
  if (gen != rt::pool_generation())
    return fail(PGPU_ERR_INVALID_PARAM, std::string(what) + " was created under a device pool that has been shut down");
  return PGPU_OK;
}

}  // namespace

pgpu_pubkey::~pgpu_pubkey() {
  std::lock_guard<std::mutex> lk(g_fb_mu);
  g_fb_keys.erase(this);
  const bool live = gen == rt::pool_generation();
  for (auto* lists : {&fb, &fbh})
    for (size_t d = 0; d < lists->size(); ++d) {
      if ((*lists)[d].empty()) continue;
      if (live && (int)d < rt::pool_size()) {
        rt::DeviceGuard g(rt::device((int)d).ordinal);
        for (FbTable& t : (*lists)[d]) {
          if (d < g_fb_dev_bytes.size()) g_fb_dev_bytes[d] -= std::min(g_fb_dev_bytes[d], t.bytes);
          fb_free_table(t);
        }
      } else {
        for (FbTable& t : (*lists)[d]) {
          fb_free_table(t);
          (void)hipGetLastError();
        }
      }
    }
}

extern "C" {

int pgpu_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int pgpu_init(int device) {
  int n = pgpu_device_count();
  if (n <= 0) return fail(PGPU_ERR_NO_DEVICE, "no HIP device visible");
  if (device >= n) return fail(PGPU_ERR_INVALID_PARAM, "device ordinal out of range");
  if (device < 0) {
    if (hipGetDevice(&device) != hipSuccess) device = 0;
  }
  return rt::pool_init(std::vector<int>{device});
}

int pgpu_init_all(int n_devices) {
  int n = pgpu_device_count();
  if (n <= 0) return fail(PGPU_ERR_NO_DEVICE, "no HIP device visible");
  if (n_devices < 0) return fail(PGPU_ERR_INVALID_PARAM, "negative device count");
  const char* over = std::getenv("PGPU_POOL_OVERSUBSCRIBE");
  const bool wrap = over && std::atoi(over) != 0;
  if (n_devices == 0) n_devices = n;
  if (n_devices > n && !wrap)
    return fail(PGPU_ERR_INVALID_PARAM, "more pool entries requested than GPUs visible (PGPU_POOL_OVERSUBSCRIBE=1 wraps)");
  std::vector<int> ord;
  for (int i = 0; i < n_devices; ++i) ord.push_back(i % n);
  return rt::pool_init(ord);
}

void pgpu_shutdown(void) {
  if (!rt::initialized()) return;
  // every cached device image was sized and placed for THIS pool: none may survive it (a later pgpu_init with
  // another device set would index past its copies or read another GPU's memory)
  (void)pgpu_synchronize();
  {
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    g_ctx_cache.clear();
  }
  {
    std::lock_guard<std::mutex> lk(g_sq_mu);
    g_sq_cache.clear();
  }
  drain_parked();
  {
    // Keys may outlive the pool (a caller's objects are destroyed after terminateContext), their fixed-base tables may
    // not: a table of the old pool left registered would be picked as an LRU victim by the NEXT pool's fb_make_room,
    // its bytes subtracted from a counter that never held them (size_t underflow: the budget then looks exceeded for
    // good) and hipFree called on memory of a pool that is gone (round-3 advisor).  So every registered table is freed
    // here, while its device is still up, and the registry starts empty; a key that is used again under a new pool is
    // refused by check_gen anyway.
    std::lock_guard<std::mutex> lk(g_fb_mu);
    for (const pgpu_pubkey* k : g_fb_keys)
      for (auto* lists : {&k->fb, &k->fbh})
        for (size_t d = 0; d < lists->size(); ++d) {
          if ((*lists)[d].empty()) continue;
          if ((int)d < rt::pool_size()) {
            rt::DeviceGuard g(rt::device((int)d).ordinal);
            for (FbTable& t : (*lists)[d]) fb_free_table(t);
          } else {
            for (FbTable& t : (*lists)[d]) fb_free_table(t);
          }
          (*lists)[d].clear();
        }
    g_fb_keys.clear();
    g_fb_dev_bytes.assign(g_fb_dev_bytes.size(), 0);
  }
  rt::pool_shutdown();
}

int pgpu_is_initialized(void) { return rt::initialized() ? 1 : 0; }
int pgpu_build_features(void) {
  return PGPU_WITH_4096 ? PGPU_FEATURE_4096_SPLIT : 0;
}
const char* pgpu_last_error(void) { return rt::g_err.c_str(); }
const char* pgpu_device_name(void) { return rt::initialized() ? rt::current().name.c_str() : ""; }
int pgpu_pool_size(void) { return rt::initialized() ? rt::pool_size() : 0; }
int pgpu_set_device(int pool_index) { return rt::set_current(pool_index); }
int pgpu_get_device(void) { return rt::initialized() ? rt::current_index() : 0; }
const char* pgpu_pool_transport(void) { return rt::replicate_transport(); }
int pgpu_set_min_shard(size_t n) {
  rt::set_min_shard(n);
  return PGPU_OK;
}

int pgpu_synchronize(void) {
  RC_TRY(rt::check_ready());
  for (int i = 0; i < rt::pool_size(); ++i) {
    rt::Device& d = rt::device(i);
    rt::DeviceGuard g(d.ordinal);
    // a batch lane that was still working when the caller came here has been active until NOW: the adaptive kernel-form
    // policy (busy_other_lanes) counts it as active for a little longer, so that a caller who synchronises between two
    // bursts over several lanes does not start every burst with the lone caller's kernel
    bool was_busy[rt::kBatchLanes];
    for (int k = 0; k < rt::kBatchLanes; ++k) was_busy[k] = hipStreamQuery(d.bs(k)) == hipErrorNotReady;
    (void)hipGetLastError();
    for (int k = 0; k < rt::kBatchLanes; ++k) HIP_TRY(hipStreamSynchronize(d.bs(k)));
    const int64_t now = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
    for (int k = 0; k < rt::kBatchLanes; ++k)
      if (was_busy[k] && d.lane_fed_ns[k].load(std::memory_order_relaxed) != 0)   // (lanes of pipelining callers only)
        d.lane_fed_ns[k].store(now, std::memory_order_relaxed);
    for (auto& lane : d.lanes) HIP_TRY(hipStreamSynchronize(lane->stream));
  }
  drain_parked();   // evicted per-modulus contexts (their hipFree waits for whatever still reads them)
  return PGPU_OK;
}

int pgpu_shard_plan(size_t count, int pool_size, int* n_shards, size_t* bounds) {
  if (pool_size <= 0 || !n_shards || !bounds) return fail(PGPU_ERR_INVALID_PARAM, "pgpu_shard_plan: bad argument");
  const int D = (int)std::max<size_t>(1, std::min<size_t>((size_t)pool_size, count / rt::min_shard()));
  *n_shards = D;
  for (int d = 0; d < D; ++d) {
    size_t lo, hi;
    rt::shard_bounds(count, D, d, &lo, &hi);
    bounds[d] = lo;
    bounds[d + 1] = hi;
  }
  return PGPU_OK;
}

int pgpu_kernel_geometry(int in_words, int mod_bits, size_t count, int* lanes, int* limbs) {
  if (in_words <= 0 || mod_bits <= 1 || !lanes || !limbs)
    return fail(PGPU_ERR_INVALID_PARAM, "pgpu_kernel_geometry: bad argument");
  const GeoInfo* geo = pick_geo(in_words, mod_bits, true);
  if (!geo) return fail(PGPU_ERR_UNSUPPORTED, "modulus wider than the compiled kernel geometries");
  const GeoInfo lat = latency_geo(*geo);
  const GeoInfo g = use_latency_geo(lat, *geo, count) ? lat : launch_geo(*geo, count);
  *lanes = g.G;
  *limbs = g.K;
  return PGPU_OK;
}

int pgpu_encrypt_kernel_form(const pgpu_pubkey* key, int m_words, size_t count, int* split, int* lanes, int* limbs) {
  return pgpu_encrypt_kernel_form_ex(key, m_words, count, 0, split, lanes, limbs);
}
int pgpu_encrypt_kernel_form_ex(const pgpu_pubkey* key, int m_words, size_t count, int busy_lanes, int* split, int* lanes,
                                int* limbs) {
  if (!key || !split || !lanes || !limbs) return fail(PGPU_ERR_INVALID_PARAM, "pgpu_encrypt_kernel_form: bad argument");
  RC_TRY(check_gen(key->gen, "key"));
  // busy_lanes < 0: the question is about RESIDENT results (pgpu_batch_encrypt: pair rows) of a lone caller
  const bool resident = busy_lanes < 0;
  if (resident) busy_lanes = 0;
  const pgpu_pubkey::PubForm* ef = key->djn && fixed_base_window() > 0 ? use_split_encrypt(key, m_words, count) : nullptr;
  if (resident && key->djn && fixed_base_window() > 0) {
    // resident results (pair rows) of small launches: one wavefront per element (hensel_wave_n2.hpp), whatever form a launch
    // from host arrays of this size would take
    const pgpu_pubkey::PubForm* pf = pair_form(key);
    if (pf && m_words <= key->n_words && wave_n2_applies(pf, count)) {
      *split = 5;
      *lanes = 64;
      *limbs = pf->H * pf->K;
      return PGPU_OK;
    }
  }
  if (ef) {
    // resident results (pair rows) take the key's pair form, and from a launch size on both halves in the same lanes
    const pgpu_pubkey::PubForm* pf = pair_form(key);
    if (pf && m_words <= key->n_words && fb_encrypt_seq_pays(pf->H, pf->K, count, busy_lanes)) {
      *split = 2;
      *lanes = pf->H;
      *limbs = pf->K;
      return PGPU_OK;
    }
    *split = 1;
    *lanes = 2 * ef->H;
    *limbs = ef->K;
    return PGPU_OK;
  }
  const GeoInfo g = launch_geo(key->nsq->geo, count);
  *split = 0;
  *lanes = g.G;
  *limbs = g.K;
  return PGPU_OK;
}

int pgpu_modexp_n2_kernel_form(const pgpu_pubkey* key, size_t count, int* split, int* lanes, int* limbs) {
  if (!key || !split || !lanes || !limbs) return fail(PGPU_ERR_INVALID_PARAM, "pgpu_modexp_n2_kernel_form: bad argument");
  RC_TRY(check_gen(key->gen, "key"));
  if (const pgpu_pubkey::PubForm* mf = split_modexp_form(key, count)) {
    if (const pgpu_pubkey::PubForm* pf = pair_form(key))                       // (resident rows, small launches: the latency form)
      if (wave_n2_applies(pf, count)) {
        *split = 5;
        *lanes = 64;
        *limbs = pf->H * pf->K;
        return PGPU_OK;
      }
    if (pair_rows_enabled() && modexp_seq_form_pays(mf->H, mf->K, count)) {   // (resident rows, per-element exponents)
      *split = 2;
      *lanes = mf->H;
    } else {
      *split = 1;
      *lanes = 2 * mf->H;
    }
    *limbs = mf->K;
    return PGPU_OK;
  }
  const GeoInfo lat = latency_geo(key->nsq->geo);
  const GeoInfo g = use_latency_geo(lat, key->nsq->geo, count) ? lat : launch_geo(key->nsq->geo, count);
  *split = 0;
  *lanes = g.G;
  *limbs = g.K;
  return PGPU_OK;
}

int pgpu_ct_add_kernel_form(const pgpu_pubkey* key, size_t count, int* split, int* lanes, int* limbs) {
  if (!key || !split || !lanes || !limbs) return fail(PGPU_ERR_INVALID_PARAM, "pgpu_ct_add_kernel_form: bad argument");
  RC_TRY(check_gen(key->gen, "key"));
  if (const pgpu_pubkey::PubForm* f = pair_form(key)) {
    f = pair_op_form(key, f, count);
    const bool seq = pair_mul_seq_pays(f->H, f->K, count);
    *split = seq ? 2 : 1;
    *lanes = seq ? f->H : 2 * f->H;
    *limbs = f->K;
    return PGPU_OK;
  }
  const GeoInfo g = launch_geo(key->nsq->geo, count);
  *split = 0;
  *lanes = g.G;
  *limbs = g.K;
  return PGPU_OK;
}

int pgpu_decrypt_kernel_form(const pgpu_privkey* key, size_t count, int* split, int* lanes, int* limbs) {
  return pgpu_decrypt_kernel_form_ex(key, count, 0, split, lanes, limbs);
}
int pgpu_decrypt_kernel_form_ex(const pgpu_privkey* key, size_t count, int busy_lanes, int* split, int* lanes, int* limbs) {
  if (!key || !split || !lanes || !limbs) return fail(PGPU_ERR_INVALID_PARAM, "pgpu_decrypt_kernel_form: bad argument");
  RC_TRY(check_gen(key->gen, "key"));
  if (const pgpu_privkey::HenselSet* f = pick_hensel(key, count)) {
    if (pair_rows_enabled() && secret_policy() != PGPU_EXP_SLIDING && wave_form_pays(key, count, busy_lanes) &&
        key->hs_ps->pair_l2 == f->pair_l2) {
      *split = 5;
      *lanes = 64;
      *limbs = key->hs_ps->K;
      return PGPU_OK;
    }
    if (pair_rows_enabled() && secret_policy() != PGPU_EXP_SLIDING && ps_form_pays(key, count, busy_lanes) &&
        key->hs_ps->pair_l2 == f->pair_l2) {
      *split = 4;
      *lanes = 1;
      *limbs = key->hs_ps->K;
      return PGPU_OK;
    }
    if (pair_rows_enabled() && secret_policy() != PGPU_EXP_SLIDING && seq_form_pays(f->H, f->K, count, busy_lanes)) {
      *split = 2;
      *lanes = f->H;
    } else {
      *split = 1;
      *lanes = 2 * f->H;
    }
    *limbs = f->K;
    return PGPU_OK;
  }
  const bool lat = use_latency_geo(key->geo_lat, key->geo_exp, 2 * count);
  const GeoInfo g = lat ? key->geo_lat : launch_geo(key->geo_exp, 2 * count);
  *split = 0;
  *lanes = g.G;
  *limbs = g.K;
  return PGPU_OK;
}

int pgpu_set_fixed_base_window(int w) {
  if (w < 0 || w > kFbMaxW) return fail(PGPU_ERR_INVALID_PARAM, "fixed-base window must be 0..14");
  g_fb_window.store(w);
  g_fb_window_explicit.store(true);
  return PGPU_OK;
}

int pgpu_set_secret_exponent_policy(int policy) {
  if (policy != PGPU_EXP_FIXED_WINDOW && policy != PGPU_EXP_SLIDING)
    return fail(PGPU_ERR_INVALID_PARAM, "unknown exponent policy");
  g_secret_policy.store(policy);
  return PGPU_OK;
}
int pgpu_get_secret_exponent_policy(void) { return secret_policy(); }
int pgpu_set_table_gather_policy(int masked) {
  g_ct_gather.store(masked ? 1 : 0);
  return PGPU_OK;
}
int pgpu_get_table_gather_policy(void) { return g_ct_gather.load(); }

// diagnostics (tools/wave_spread.py): device buffer that receives per-wave start/end clocks of the
// next modexp_kernel launches; null switches it off.  Not part of the public header.
void pgpu_debug_set_wave_clocks(uint64_t* d_buf) { g_wave_clocks = d_buf; }
// tests / A-B measurements: force where modexp_kernel takes its multiplier rows from (-1 auto by wavefront count,
// 0 LDS, 1 registers; geometries without a register form keep LDS).  Not part of the public header.
void pgpu_debug_set_row_source(int mode) { g_row_source.store(mode < 0 ? -1 : (mode ? 1 : 0)); }
// tests / A-B measurements: 0 = CRT decrypt through the full-width modexp_kernel, 1 = split form where compiled
// (throughput or latency form by batch size), 2 / 3 = always its throughput / latency form
void pgpu_debug_set_hensel(int mode) { g_hensel.store(mode < 0 ? 0 : (mode > 3 ? 3 : mode)); }
// A/B measurements (bench.py two_streams): 1 = the two-wavefronts-per-SIMD build of the (2,19) decrypt kernel for
// every launch.  Not part of the public header.
void pgpu_debug_set_packed_decrypt(int on) { g_packed_decrypt.store(on != 0); }
// tests / A-B measurements: hensel_seq.hpp (0 never, 1 by launch size, 2 whenever it applies).  Not part of the public header.
void pgpu_debug_set_seq_decrypt(int policy) { pgpu::policy::set_seq_policy(policy); }
int pgpu_debug_set_host_adapt(int on) {
  const int was = g_host_adapt.exchange(on != 0 ? 1 : 0);
  return was;
}
int pgpu_debug_get_seq_decrypt(void) { return pgpu::policy::seq_policy(); }
// tests / A-B measurements: hensel_ps.hpp (0 never, 1 by launch size and neighbour lanes, 2 whenever it is compiled)
void pgpu_debug_set_ps_decrypt(int policy) { pgpu::policy::set_ps_policy(policy); }
// tests / A-B measurements: hensel_wave.hpp (0 never, 1 small lone launches, 2 whenever it is compiled)
void pgpu_debug_set_wave_decrypt(int policy) { pgpu::policy::set_wave_policy(policy); }
int pgpu_debug_get_wave_decrypt(void) { return pgpu::policy::wave_policy(); }
int pgpu_debug_get_ps_decrypt(void) { return pgpu::policy::ps_policy(); }
// tests / A-B measurements: from how many active neighbour lanes on threads on round-robin lanes take the adaptive forms (0 never)
int pgpu_debug_set_rr_adapt(int min_busy) { return pgpu::policy::set_rr_adapt(min_busy); }

#include "capi_timing.inc"   // pgpu_set_timing, pgpu_timing_collect*

// ===================== device buffers (current pool entry) =====================
// Blocks handed out here are used on the default stream and on caller streams: their free list is the one of
// the null stream, and a freed block may be handed out again at once -- the caller orders its work on it.
int pgpu_dev_alloc(size_t bytes, void** out) {
  RC_TRY(rt::check_ready());
  return rt::current().alloc(bytes, nullptr, out);
}
void pgpu_dev_free(void* d_ptr) {
  if (!d_ptr) return;
  if (!rt::initialized()) {
    (void)hipFree(d_ptr);
    return;
  }
  rt::current().free(d_ptr, nullptr);
}
int pgpu_copy_h2d(void* d_dst, const void* h_src, size_t bytes) {
  RC_TRY(rt::check_ready());
  rt::TaskGroup tg;
  // ordered behind the default stream (the buffer may still be read by queued work), complete on return
  tg.run(rt::current(), [=](rt::Lane& lane) -> int {
    HIP_TRY(hipStreamSynchronize(nullptr));
    RC_TRY(lane.h2d(d_dst, h_src, bytes, lane.stream));
    HIP_TRY(hipStreamSynchronize(lane.stream));
    return PGPU_OK;
  });
  return tg.wait();
}
int pgpu_copy_d2h(void* h_dst, const void* d_src, size_t bytes) {
  RC_TRY(rt::check_ready());
  rt::TaskGroup tg;
  tg.run(rt::current(), [=](rt::Lane& lane) -> int {
    hipError_t e = hipStreamSynchronize(nullptr);   // producers on the default stream
    if (e != hipSuccess) return fail(PGPU_ERR_HIP, std::string("kernel failed: ") + hipGetErrorString(e));
    return lane.d2h(h_dst, d_src, bytes, lane.stream);
  });
  return tg.wait();
}

// ---- pinned host memory (include/pgpu.h) ----
int pgpu_host_alloc(size_t bytes, void** out) { return rt::host_alloc(bytes, out); }
void pgpu_host_free(void* p) { rt::host_free(p); }
int pgpu_host_wait(const void* p) { return rt::host_wait(p); }

// ===================== generic modexp =====================
int pgpu_modexp_dev(const uint64_t* d_base, size_t base_stride, const uint64_t* d_exp,
                    size_t exp_stride, int exp_words, int exp_bits, const uint64_t* h_mod,
                    int mod_words, uint64_t* d_out, size_t count, void* hip_stream) {
  RC_TRY(rt::check_ready());
  if (count == 0) return PGPU_OK;
  if (!d_base || !d_exp || !d_out) return fail(PGPU_ERR_INVALID_PARAM, "null batch pointer");
  if (exp_words <= 0 || exp_bits < 0 || exp_bits > 64 * exp_words)
    return fail(PGPU_ERR_INVALID_PARAM, "exp_bits/exp_words inconsistent");
  if (base_stride != 0 && base_stride < (size_t)mod_words)
    return fail(PGPU_ERR_INVALID_PARAM, "base stride smaller than the modulus width");
  if (exp_stride != 0 && exp_stride < (size_t)exp_words)
    return fail(PGPU_ERR_INVALID_PARAM, "exponent stride smaller than exp_words");
  rt::Device& d = rt::current();
  rt::DeviceGuard g(d.ordinal);
  return modexp_on(d, d_base, base_stride, d_exp, exp_stride, exp_words, exp_bits, h_mod, mod_words, d_out, count,
                   (hipStream_t)hip_stream, nullptr, false, false, nullptr);
}

int pgpu_modexp(const uint64_t* base, size_t base_stride, const uint64_t* exp, size_t exp_stride,
                int exp_words, int exp_bits, const uint64_t* mod, int mod_words, uint64_t* out,
                size_t count) {
  RC_TRY(rt::check_ready());
  if (count == 0) return PGPU_OK;
  if (!base || !exp || !out || mod_words <= 0 || exp_words <= 0)
    return fail(PGPU_ERR_INVALID_PARAM, "null batch pointer or zero width");
  if (exp_bits < 0 || exp_bits > 64 * exp_words)
    return fail(PGPU_ERR_INVALID_PARAM, "exp_bits/exp_words inconsistent");
  if (base_stride != 0 && base_stride < (size_t)mod_words)
    return fail(PGPU_ERR_INVALID_PARAM, "base stride smaller than the modulus width");
  if (exp_stride != 0 && exp_stride < (size_t)exp_words)
    return fail(PGPU_ERR_INVALID_PARAM, "exponent stride smaller than exp_words");
  std::shared_ptr<ModCtx> ctx;
  RC_TRY(get_modctx(mod, mod_words, true, &ctx));   // validates the modulus before any task starts
  // one exponent for the whole batch, and the host has it: sliding-window schedule instead of a digit scan
  // (the caller handed the exponent over in the clear: not a secret of this library)
  std::vector<uint16_t> sched;
  int sched_w = 0;
  if (exp_stride == 0 && count >= 16 && exp_bits > 8 && sliding_enabled()) {
    BigNumber e = BigNumber::fromLimbs64(exp, (size_t)exp_words);
    if (!e.isZero()) {
      sched_w = pick_sliding_window(e.BitSize());
      sched = sliding_schedule(e, sched_w);
    }
  }
  const size_t sub_min = exp_bits >= 256 ? kSubMinHeavy : kSubMinLight;
  return run_sharded(count, sub_min, [=, &sched](rt::Lane& lane, size_t lo, size_t hi) -> int {
    rt::Device& d = *lane.dev;
    hipStream_t s = lane.stream;
    const size_t n = hi - lo;
    rt::DevMem db, de, dout, dsched;
    const size_t nb = (base_stride ? n * base_stride : (size_t)mod_words) * 8;
    const size_t ne = (exp_stride ? n * exp_stride : (size_t)exp_words) * 8;
    RC_TRY(db.alloc(d, s, nb));
    RC_TRY(de.alloc(d, s, ne));
    RC_TRY(dout.alloc(d, s, n * (size_t)mod_words * 8));
    RC_TRY(lane.h2d(db.p, base + lo * base_stride, nb, s));
    RC_TRY(lane.h2d(de.p, exp + lo * exp_stride, ne, s));
    SchedRef sr;
    if (!sched.empty()) {
      RC_TRY(dsched.alloc(d, s, sched.size() * sizeof(uint16_t)));
      RC_TRY(lane.h2d(dsched.p, sched.data(), sched.size() * sizeof(uint16_t), s));
      sr.p[0] = (const uint16_t*)dsched.p;
      sr.len[0] = (int)sched.size();
      sr.w = sched_w;
    }
    RC_TRY(modexp_on(d, (const uint64_t*)db.p, base_stride, (const uint64_t*)de.p, exp_stride, exp_words, exp_bits,
                     mod, mod_words, (uint64_t*)dout.p, n, s, sr.p[0] ? &sr : nullptr, false, false, ctx, count));
    return lane.d2h(out + lo * (size_t)mod_words, dout.p, n * (size_t)mod_words * 8, s);
  });
}

// ===================== modmul =====================
int pgpu_modmul_dev(const uint64_t* d_a, const uint64_t* d_b, size_t b_stride,
                    const uint64_t* h_mod, int mod_words, uint64_t* d_out, size_t count,
                    void* hip_stream) {
  RC_TRY(rt::check_ready());
  if (count == 0) return PGPU_OK;
  if (!d_a || !d_b || !d_out) return fail(PGPU_ERR_INVALID_PARAM, "null batch pointer");
  if (b_stride != 0 && b_stride < (size_t)mod_words)
    return fail(PGPU_ERR_INVALID_PARAM, "b stride smaller than the modulus width");
  std::shared_ptr<ModCtx> ctx;
  RC_TRY(get_modctx(h_mod, mod_words, false, &ctx));
  rt::Device& d = rt::current();
  rt::DeviceGuard g(d.ordinal);
  return modmul_on(d, *ctx, pgpu::MM_PLAIN, d_a, d_b, b_stride, 0, d_out, count, (hipStream_t)hip_stream);
}

int pgpu_modmul(const uint64_t* a, const uint64_t* b, size_t b_stride, const uint64_t* mod,
                int mod_words, uint64_t* out, size_t count) {
  RC_TRY(rt::check_ready());
  if (count == 0) return PGPU_OK;
  if (!a || !b || !out || mod_words <= 0)
    return fail(PGPU_ERR_INVALID_PARAM, "null batch pointer or zero width");
  if (b_stride != 0 && b_stride < (size_t)mod_words)
    return fail(PGPU_ERR_INVALID_PARAM, "b stride smaller than the modulus width");
  std::shared_ptr<ModCtx> ctx;
  RC_TRY(get_modctx(mod, mod_words, false, &ctx));
  return run_sharded(count, kSubMinLight, [=](rt::Lane& lane, size_t lo, size_t hi) -> int {
    rt::Device& d = *lane.dev;
    hipStream_t s = lane.stream;
    const size_t n = hi - lo, row = (size_t)mod_words * 8;
    rt::DevMem da, db, dout;
    RC_TRY(da.alloc(d, s, n * row));
    RC_TRY(db.alloc(d, s, b_stride ? n * b_stride * 8 : row));
    RC_TRY(dout.alloc(d, s, n * row));
    RC_TRY(lane.h2d(da.p, a + lo * (size_t)mod_words, n * row, s));
    RC_TRY(lane.h2d(db.p, b + lo * b_stride, b_stride ? n * b_stride * 8 : row, s));
    RC_TRY(modmul_on(d, *ctx, pgpu::MM_PLAIN, (const uint64_t*)da.p, (const uint64_t*)db.p, b_stride, 0,
                     (uint64_t*)dout.p, n, s));
    return lane.d2h(out + lo * (size_t)mod_words, dout.p, n * row, s);
  });
}

#include "capi_keys.inc"   // pgpu_pubkey_create, pgpu_privkey_create and their constants

int pgpu_paillier_encrypt_dev(const pgpu_pubkey* key, const uint64_t* d_m, size_t m_stride,
                              int m_words, const uint64_t* d_r, size_t r_stride, int r_words,
                              int r_bits, uint64_t* d_c, size_t count, void* hip_stream) {
  RC_TRY(rt::check_ready());
  if (!key) return fail(PGPU_ERR_INVALID_PARAM, "null key");
  RC_TRY(check_gen(key->gen, "key"));
  if (count == 0) return PGPU_OK;
  if (!d_m || !d_r || !d_c) return fail(PGPU_ERR_INVALID_PARAM, "null batch pointer");
  rt::Device& d = rt::current();
  rt::DeviceGuard g(d.ordinal);
  RC_TRY(encrypt_on(d, key, d_m, m_stride, m_words, d_r, r_stride, r_words, r_bits, d_c, count,
                    (hipStream_t)hip_stream, false, count));
  if (key->djn) {
    std::lock_guard<std::mutex> lk(key->mu);
    key->fb_elems += count;
  }
  return PGPU_OK;
}

int pgpu_paillier_encrypt(const pgpu_pubkey* key, const uint64_t* m, size_t m_stride, int m_words,
                          const uint64_t* r, size_t r_stride, int r_words, int r_bits,
                          uint64_t* c, size_t count) {
  RC_TRY(rt::check_ready());
  if (!key) return fail(PGPU_ERR_INVALID_PARAM, "null key");
  RC_TRY(check_gen(key->gen, "key"));
  if (count == 0) return PGPU_OK;
  if (!m || !r || !c) return fail(PGPU_ERR_INVALID_PARAM, "null batch pointer");
  const int W = 2 * key->n_words;
  if (m_words <= 0 || m_words > W || m_stride < (size_t)m_words)
    return fail(PGPU_ERR_INVALID_PARAM, "plaintext width/stride invalid");
  if (r_words <= 0 || r_stride < (size_t)r_words)
    return fail(PGPU_ERR_INVALID_PARAM, "random width/stride invalid");
  const size_t sub_min = (key->djn && fixed_base_window() > 0) ? kSubMinLight : kSubMinHeavy;
  // PGPU_HOST_ADAPT=1 (default 0): callers of the host-array entry points count as users of their thread's batch lane
  // (new_batch: thread_batch_lane), see each other -- and the resident batches of other lanes -- through the lane
  // activity stamps, and their launches take the part-chip forms of the adaptive policy like those of resident batches.
  // Measured r04 (tools/probe_two_callers.py, profiles/r04_two_callers.txt): it does NOT pay for synchronous callers.
  // Two of them keep the GPU busy with full-chip kernels back to back already (2 x (0.82 + 4.58) ms per pair of calls,
  // the copies of one under the kernels of the other: 5.45 ms per encrypt + decrypt); on half the chip each, a caller's
  // half idles while its copies and host work run, and the pair of calls takes 6.0 ms.  The part-chip forms need an
  // asynchronous feed -- resident batches on the batch lanes.
  const int caller_lane = thread_batch_lane();
  const pgpu_pubkey::PubForm* pf =
      (key->djn && fixed_base_window() > 0 && 64 * m_words <= key->n.BitSize()) ? pair_form(key) : nullptr;
  int rc = run_sharded(count, sub_min, [=](rt::Lane& lane, size_t lo, size_t hi) -> int {
    rt::Device& d = *lane.dev;
    hipStream_t s = lane.stream;
    const size_t n = hi - lo;
    rt::DevMem dm, dr, dc, rows;
    RC_TRY(dm.alloc(d, s, n * m_stride * 8));
    RC_TRY(dr.alloc(d, s, n * r_stride * 8));
    RC_TRY(dc.alloc(d, s, n * (size_t)W * 8));
    RC_TRY(lane.h2d(dm.p, m + lo * m_stride, n * m_stride * 8, s));
    RC_TRY(lane.h2d(dr.p, r + lo * r_stride, n * r_stride * 8, s));
    const int busy = host_busy(d, caller_lane, n);
    if (pf && busy > 0 && fb_encrypt_seq_pays(pf->H, pf->K, n, busy)) {
      // the sequential-halves kernel leaves pair rows: one pair_ops_kernel pass brings them back to words
      RC_TRY(rows.alloc(d, s, n * (size_t)2 * pf->H * pf->K * sizeof(uint32_t)));
      RC_TRY(encrypt_on(d, key, (const uint64_t*)dm.p, m_stride, m_words, (const uint64_t*)dr.p, r_stride, r_words,
                        r_bits, nullptr, n, s, true, count, (uint32_t*)rows.p, busy));
      RC_TRY(pair_to_words_on(d, pf, (const uint32_t*)rows.p, (uint64_t*)dc.p, n, s));
    } else {
      RC_TRY(encrypt_on(d, key, (const uint64_t*)dm.p, m_stride, m_words, (const uint64_t*)dr.p, r_stride, r_words,
                        r_bits, (uint64_t*)dc.p, n, s, false, count));
    }
    const int rcd = lane.d2h(c + lo * (size_t)W, dc.p, n * (size_t)W * 8, s);
    (void)host_busy(d, caller_lane);   // (stamp: a long call stays visible until it ends)
    return rcd;
  });
  if (rc == PGPU_OK && key->djn) {
    std::lock_guard<std::mutex> lk(key->mu);
    key->fb_elems += count;
  }
  return rc;
}


int pgpu_paillier_decrypt_crt_dev(const pgpu_privkey* key, const uint64_t* d_c, uint64_t* d_m,
                                  size_t count, void* hip_stream) {
  RC_TRY(rt::check_ready());
  if (!key) return fail(PGPU_ERR_INVALID_PARAM, "null key");
  RC_TRY(check_gen(key->gen, "key"));
  if (count == 0) return PGPU_OK;
  if (!d_c || !d_m) return fail(PGPU_ERR_INVALID_PARAM, "null batch pointer");
  rt::Device& d = rt::current();
  rt::DeviceGuard g(d.ordinal);
  return decrypt_on(d, key, d_c, d_m, count, (hipStream_t)hip_stream, false);
}

int pgpu_paillier_decrypt_crt(const pgpu_privkey* key, const uint64_t* c, uint64_t* m,
                              size_t count) {
  RC_TRY(rt::check_ready());
  if (!key) return fail(PGPU_ERR_INVALID_PARAM, "null key");
  RC_TRY(check_gen(key->gen, "key"));
  if (count == 0) return PGPU_OK;
  if (!c || !m) return fail(PGPU_ERR_INVALID_PARAM, "null batch pointer");
  const int nw = key->n_words;
  const int caller_lane = thread_batch_lane();   // (see pgpu_paillier_encrypt)
  return run_sharded(count, kSubMinHeavy, [=](rt::Lane& lane, size_t lo, size_t hi) -> int {
    rt::Device& d = *lane.dev;
    hipStream_t s = lane.stream;
    const size_t n = hi - lo;
    rt::DevMem dc, dm;
    RC_TRY(dc.alloc(d, s, n * (size_t)2 * nw * 8));
    RC_TRY(dm.alloc(d, s, n * (size_t)nw * 8));
    RC_TRY(lane.h2d(dc.p, c + lo * (size_t)2 * nw, n * (size_t)2 * nw * 8, s));
    RC_TRY(decrypt_on(d, key, (const uint64_t*)dc.p, (uint64_t*)dm.p, n, s, false, nullptr, 0, host_busy(d, caller_lane, n)));
    const int rcd = lane.d2h(m + lo * (size_t)nw, dm.p, n * (size_t)nw * 8, s);
    (void)host_busy(d, caller_lane);
    return rcd;
  });
}

#include "capi_batches.inc"   // pgpu_batch_*

// ---- diagnostics of the pool's self-checks and table budget ----
int pgpu_replication_stats(uint64_t* images_verified, uint64_t* copies_repaired) {
  rt::replication_stats(images_verified, copies_repaired);
  return PGPU_OK;
}
const char* pgpu_rccl_note(void) {
  static thread_local std::string note;
  note = rt::rccl_note();
  return note.c_str();
}
int pgpu_debug_corrupt_next_replica(int pool_index) {
  RC_TRY(rt::check_ready());
  if (pool_index < 0 || pool_index >= rt::pool_size()) return fail(PGPU_ERR_INVALID_PARAM, "pool index out of range");
  rt::debug_corrupt_next_replica(pool_index);
  return PGPU_OK;
}
int pgpu_set_fixed_base_budget(size_t max_bytes_per_device, size_t max_bytes_per_key) {
  if (max_bytes_per_device) g_fb_dev_max.store(max_bytes_per_device);
  if (max_bytes_per_key) g_fb_key_max.store(max_bytes_per_key);
  return PGPU_OK;
}
int pgpu_fixed_base_stats(int pool_index, size_t* live_bytes, uint64_t* evictions) {
  std::lock_guard<std::mutex> lk(g_fb_mu);
  if (live_bytes) *live_bytes = (pool_index >= 0 && (size_t)pool_index < g_fb_dev_bytes.size()) ? g_fb_dev_bytes[(size_t)pool_index] : 0;
  if (evictions) *evictions = g_fb_evictions.load();
  return PGPU_OK;
}
int pgpu_pubkey_fixed_base_info(const pgpu_pubkey* key, int pool_index, int* window, size_t* bytes, double* build_ms) {
  if (!key) return fail(PGPU_ERR_INVALID_PARAM, "null key");
  std::lock_guard<std::mutex> lk(g_fb_mu);
  int w = 0;
  size_t b = 0;
  double ms = 0;
  for (auto* lists : {&key->fb, &key->fbh}) {
    if (pool_index < 0 || (size_t)pool_index >= lists->size()) continue;
    for (FbTable& t : (*lists)[(size_t)pool_index])
      if (t.w >= w) {
        w = t.w;
        b = t.bytes;
        if (t.build_ms == 0 && t.t0 && t.ready && hipEventQuery(t.ready) == hipSuccess) {
          float f = 0;
          if (hipEventElapsedTime(&f, t.t0, t.ready) == hipSuccess) t.build_ms = f;
        }
        ms = t.build_ms;
      }
  }
  if (window) *window = w;
  if (bytes) *bytes = b;
  if (build_ms) *build_ms = ms;
  return PGPU_OK;
}

}  // extern "C"

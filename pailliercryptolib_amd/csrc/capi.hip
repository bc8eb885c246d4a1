// pailliercryptolib_amd -- implementation of the C-ABI declared in include/pgpu.h.
// Host side: device/context management, Montgomery-constant precomputation (host BigNumber),
// geometry selection and kernel launches.  No CPU fallback: everything computes on the GPU.
#include "pgpu.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <condition_variable>
#include <thread>
#include <vector>

#include "ipcl/bignum.h"
#include "kernels.hpp"

namespace {

thread_local std::string g_err;
std::recursive_mutex g_mu;
bool g_init = false;
int g_device = -1;
std::string g_devname;
// live per-launch timing (bench.py roofline): when enabled every kernel launch is bracketed by a
// pair of HIP events recorded on the launch stream, WITHOUT synchronising; pgpu_timing_collect
// synchronises once and returns the durations.
bool g_timing = false;
struct TimedLaunch { int kind; hipEvent_t e0, e1; };
std::vector<TimedLaunch> g_timed;
std::vector<hipEvent_t> g_event_pool;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess)                                                                      \
      return fail(PGPU_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));            \
  } while (0)
#define RC_TRY(expr)          \
  do {                        \
    int rc_ = (expr);         \
    if (rc_) return rc_;      \
  } while (0)

// ---------- geometry table ----------
struct GeoInfo {
  int G, K;
  int L() const { return G * K; }
  int rbits() const { return pgpu::kLimbBits * G * K; }
  int w64() const { return (rbits() + 63) / 64; }
  int ipw() const { return pgpu::kWave / G; }
};
const GeoInfo kGeos[] = {{2, 9}, {4, 9}, {4, 14}, {8, 9}, {8, 14}, {16, 9}, {16, 14}, {16, 18}};

// smallest geometry with R = 2^(29*G*K) >= 2^(64*in_words) (any input row fits) and R >= 256*N
const GeoInfo* pick_geo(int in_words, int mod_bits) {
  for (const GeoInfo& g : kGeos)
    if (g.rbits() >= 64 * in_words && g.rbits() >= mod_bits + 8) return &g;
  return nullptr;
}

// ---------- device blobs ----------
struct DevBlob {
  void* p = nullptr;
  ~DevBlob() {
    if (p) (void)hipFree(p);
  }
  int upload(const void* src, size_t bytes) {
    HIP_TRY(hipMalloc(&p, bytes));
    HIP_TRY(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
    return PGPU_OK;
  }
};

void to_limbs29(const BigNumber& v, int L, uint32_t* out) {
  const std::vector<uint64_t>& w = v.limbs64();
  for (int i = 0; i < L; ++i) {
    int bit = i * pgpu::kLimbBits;
    size_t word = (size_t)bit >> 6;
    int sh = bit & 63;
    uint64_t x = word < w.size() ? w[word] >> sh : 0;
    if (sh > 64 - pgpu::kLimbBits && word + 1 < w.size()) x |= w[word + 1] << (64 - sh);
    out[i] = (uint32_t)x & pgpu::kLimbMask;
  }
}

BigNumber pow2(int bits) {
  std::vector<uint64_t> w((size_t)bits / 64 + 1, 0);
  w[(size_t)bits / 64] = 1ull << (bits % 64);
  return BigNumber::fromLimbs64(w.data(), w.size());
}

// ---------- modulus context ----------
// Host description of what goes into a pgpu::ModCtxDev; every BigNumber is a plain integer,
// the builder converts to the Montgomery constants of the chosen geometry.
struct ModCtx {
  GeoInfo geo{};
  int mod_words = 0;
  BigNumber N;
  DevBlob blob;
  pgpu::ModCtxDev dev{};
};

// extras: optional constants appended to the context blob
struct CtxExtras {
  bool unit_q = false;       // scale the loop modulus to Nhat = N*k == -1 mod 2^29 (if R has room)
  bool want_r2s = false;     // R^2 * 2^(64*mod_words) mod N
  const BigNumber* fc = nullptr;   // plain final multiplier
  const BigNumber* nr_n = nullptr; // n  ->  n*R mod N
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
  const bool unit = ex.unit_q && geo.rbits() >= N.BitSize() + pgpu::kLimbBits + 8;
  const BigNumber M = unit ? N * BigNumber((Ipp32u)n0inv) : N;   // loop modulus
  BigNumber Rm = R % M;
  BigNumber R2 = (Rm * Rm) % M;

  std::vector<uint32_t> h((size_t)7 * L, 0);
  to_limbs29(N, L, h.data());
  to_limbs29(R2, L, h.data() + L);
  to_limbs29(Rm, L, h.data() + 2 * L);
  if (ex.want_r2s) to_limbs29((R2 * (pow2(64 * mod_words) % M)) % M, L, h.data() + 3 * L);
  if (ex.fc) to_limbs29(*ex.fc % N, L, h.data() + 4 * L);
  if (ex.nr_n) to_limbs29((*ex.nr_n * (R % N)) % N, L, h.data() + 5 * L);   // used under the TRUE modulus
  if (unit) to_limbs29(M, L, h.data() + 6 * L);
  std::vector<uint64_t> n64((size_t)W64 + 1, 0);
  N.toLimbs64(n64.data(), n64.size());

  auto ctx = std::make_shared<ModCtx>();
  ctx->geo = geo;
  ctx->mod_words = mod_words;
  ctx->N = N;
  size_t bytes32 = h.size() * sizeof(uint32_t);
  size_t off64 = (bytes32 + 15) & ~(size_t)15;
  std::vector<uint8_t> host(off64 + n64.size() * 8, 0);
  std::memcpy(host.data(), h.data(), bytes32);
  std::memcpy(host.data() + off64, n64.data(), n64.size() * 8);
  RC_TRY(ctx->blob.upload(host.data(), host.size()));
  uint32_t* d32 = (uint32_t*)ctx->blob.p;
  ctx->dev.n = d32;
  ctx->dev.r2 = d32 + L;
  ctx->dev.one = d32 + 2 * L;
  ctx->dev.r2s = ex.want_r2s ? d32 + 3 * L : nullptr;
  ctx->dev.fc = ex.fc ? d32 + 4 * L : nullptr;
  ctx->dev.nr = ex.nr_n ? d32 + 5 * L : nullptr;
  ctx->dev.nhat = unit ? d32 + 6 * L : nullptr;
  ctx->dev.n64 = (const uint64_t*)((char*)ctx->blob.p + off64);
  ctx->dev.n0inv = n0inv;
  ctx->dev.mod_words = mod_words;
  *out = ctx;
  return PGPU_OK;
}

std::map<std::vector<uint64_t>, std::shared_ptr<ModCtx>> g_ctx_cache;

// cached plain context for the generic seam: unit_q = true for pgpu_modexp (loop modulo Nhat),
// false for pgpu_modmul (two multiplications, true modulus throughout)
GeoInfo latency_geo(const GeoInfo& geo);
// latency: build the context for the 16-lane latency geometry of the modulus' class (same context when L
// is unchanged or the class has none)
int get_modctx(const uint64_t* mod, int mod_words, bool unit_q, std::shared_ptr<ModCtx>* out,
               bool latency = false) {
  if (!mod || mod_words <= 0) return fail(PGPU_ERR_INVALID_PARAM, "modulus is null/empty");
  if (!(mod[0] & 1)) return fail(PGPU_ERR_EVEN_MODULUS, "modulus must be odd");
  std::vector<uint64_t> key(mod, mod + mod_words);
  key.push_back((unit_q ? 1 : 0) | (latency ? 2 : 0));
  auto it = g_ctx_cache.find(key);
  if (it != g_ctx_cache.end()) {
    *out = it->second;
    return PGPU_OK;
  }
  BigNumber N = BigNumber::fromLimbs64(mod, (size_t)mod_words);
  if (N.isZero() || N == BigNumber::One())
    return fail(PGPU_ERR_INVALID_PARAM, "modulus must be > 1");
  const GeoInfo* geo = pick_geo(mod_words, N.BitSize());
  if (!geo) return fail(PGPU_ERR_UNSUPPORTED, "modulus wider than the compiled kernel geometries");
  CtxExtras ex;
  ex.unit_q = unit_q;
  RC_TRY(build_modctx(N, mod_words, latency ? latency_geo(*geo) : *geo, ex, out));
  if (g_ctx_cache.size() > 64) g_ctx_cache.clear();
  g_ctx_cache[key] = *out;
  return PGPU_OK;
}

// ---------- grow-only device workspaces ----------
struct Workspace {
  void* p = nullptr;
  size_t bytes = 0;
  int ensure(size_t need) {
    if (need <= bytes) return PGPU_OK;
    if (p) {
      HIP_TRY(hipDeviceSynchronize());
      HIP_TRY(hipFree(p));
      p = nullptr;
      bytes = 0;
    }
    HIP_TRY(hipMalloc(&p, need));
    bytes = need;
    return PGPU_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
};
Workspace g_table, g_vbuf;

// fixed-window width: the w in 1..5 that minimises (2^w - 2) table multiplications +
// ceil(e/w) window multiplications (w = 5 for e >= ~240 bits).
// PGPU_SLIDING=0 keeps the fixed-window digit scan for key exponents too (A/B measurements)
bool sliding_enabled() {
  static const bool on = [] { const char* e = std::getenv("PGPU_SLIDING"); return !e || std::atoi(e) != 0; }();
  return on;
}

int pick_window(int exp_bits) {
  int best = 1;
  long best_cost = 1L << 60;
  for (int w = 1; w <= 5; ++w) {
    long cost = ((1L << w) - 2) + (exp_bits + w - 1) / w;
    if (cost < best_cost) { best_cost = cost; best = w; }
  }
  return best;
}

// Sliding-window schedule of an exponent the host knows (key constants: p-1, q-1): odd powers
// base^(2i+1), i < 2^(w-1); step = (nsq << 6) | (idx + 1) -- nsq squarings then * base^(2 idx + 1)
// (idx + 1 == 0: squarings only); step 0 has nsq == 0 and loads its entry (kernels.hpp ModexpArgs).
struct ExpSchedule {
  DevBlob dev;
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
int make_schedule(const BigNumber& e, int w, ExpSchedule* out) {
  std::vector<uint16_t> st = sliding_schedule(e, w);
  out->len = (int)st.size();
  out->w = w;
  if (st.empty()) st.push_back(0);
  return out->dev.upload(st.data(), st.size() * sizeof(uint16_t));
}

// window width of the fixed-base table for the DJN obfuscator; PGPU_FB_WINDOW=0 selects the
// generic (per-instance table, square-and-multiply) kernel instead.
int g_fb_window = -1;
bool g_fb_window_explicit = false;   // set through the environment or pgpu_set_fixed_base_window
int fixed_base_window() {
  if (g_fb_window < 0) {
    const char* e = std::getenv("PGPU_FB_WINDOW");
    int v = e ? std::atoi(e) : 12;
    g_fb_window = (v < 0 || v > 12) ? 12 : v;
    g_fb_window_explicit = e != nullptr;
  }
  return g_fb_window;
}
// A key that has encrypted little so far starts with an 8-bit window (table 16x smaller, built in ~2 ms
// instead of ~37 ms) and moves to the configured one once this many elements have gone through it.
constexpr size_t kFbGrowAfter = 4096;

hipEvent_t pool_event() {
  if (!g_event_pool.empty()) {
    hipEvent_t e = g_event_pool.back();
    g_event_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

struct TimerScope {
  hipStream_t s;
  bool on;
  TimedLaunch t{};
  TimerScope(hipStream_t st, int kind) : s(st), on(g_timing && g_timed.size() < 65536) {
    if (!on) return;
    t.kind = kind;
    t.e0 = pool_event();
    t.e1 = pool_event();
    (void)hipEventRecord(t.e0, s);
  }
  void stop() {
    if (!on) return;
    (void)hipEventRecord(t.e1, s);
    g_timed.push_back(t);
  }
};

// wavefronts of a modexp launch: with two contexts every wave takes one parity of the instances
// (kernels.hpp: ModexpArgs::nctx), i.e. 2 * ceil(elements / IPW)
size_t modexp_waves(size_t count, int parity_waves, int ipw) {
  return parity_waves ? 2 * ((count / 2 + ipw - 1) / ipw) : (count + ipw - 1) / ipw;
}
template <int G, int K>
void launch_modexp(const pgpu::ModexpArgs& a, hipStream_t s) {
  typedef pgpu::Geo<G, K> GEO;
  unsigned blocks = (unsigned)((modexp_waves(a.count, a.parity_waves, GEO::IPW) + pgpu::kWavesPerWG - 1) / pgpu::kWavesPerWG);
  hipLaunchKernelGGL((pgpu::modexp_kernel<GEO>), dim3(blocks), dim3(pgpu::kWGThreads), 0, s, a);
}
template <int G, int K>
void launch_modmul(const pgpu::ModmulArgs& a, hipStream_t s) {
  typedef pgpu::Geo<G, K> GEO;
  unsigned blocks = (unsigned)((a.count + GEO::IPW * pgpu::kWavesPerWG - 1) / (GEO::IPW * pgpu::kWavesPerWG));
  hipLaunchKernelGGL((pgpu::modmul_kernel<GEO>), dim3(blocks), dim3(pgpu::kWGThreads), 0, s, a);
}
template <int G, int K>
void launch_crt(const pgpu::CrtArgs& a, hipStream_t s) {
  typedef pgpu::Geo<G, K> GEO;
  unsigned blocks = (unsigned)((a.count + GEO::IPW * pgpu::kWavesPerWG - 1) / (GEO::IPW * pgpu::kWavesPerWG));
  hipLaunchKernelGGL((pgpu::crt_kernel<GEO>), dim3(blocks), dim3(pgpu::kWGThreads), 0, s, a);
}

template <int G, int K>
void launch_fb_build(const pgpu::FixedBaseBuildArgs& a, hipStream_t s) {
  typedef pgpu::Geo<G, K> GEO;
  unsigned blocks = (unsigned)((a.nwin + GEO::IPW * pgpu::kWavesPerWG - 1) / (GEO::IPW * pgpu::kWavesPerWG));
  hipLaunchKernelGGL((pgpu::fb_build_kernel<GEO>), dim3(blocks), dim3(pgpu::kWGThreads), 0, s, a);
}
template <int G, int K>
void launch_fb_encrypt(const pgpu::FixedBaseArgs& a, hipStream_t s) {
  typedef pgpu::Geo<G, K> GEO;
  unsigned blocks = (unsigned)((a.count + GEO::IPW * pgpu::kWavesPerWG - 1) / (GEO::IPW * pgpu::kWavesPerWG));
  hipLaunchKernelGGL((pgpu::fb_encrypt_kernel<GEO>), dim3(blocks), dim3(pgpu::kWGThreads), 0, s, a);
}

#define GEO_DISPATCH(FN, geo, ...)                                  \
  do {                                                              \
    if (geo.G == 2 && geo.K == 18) FN<2, 18>(__VA_ARGS__);          \
    else if (geo.G == 4 && geo.K == 18) FN<4, 18>(__VA_ARGS__);     \
    else if (geo.G == 8 && geo.K == 18) FN<8, 18>(__VA_ARGS__);     \
    else if (geo.G == 2 && geo.K == 9) FN<2, 9>(__VA_ARGS__);       \
    else if (geo.G == 4 && geo.K == 9) FN<4, 9>(__VA_ARGS__);       \
    else if (geo.G == 4 && geo.K == 14) FN<4, 14>(__VA_ARGS__);     \
    else if (geo.G == 8 && geo.K == 9) FN<8, 9>(__VA_ARGS__);       \
    else if (geo.G == 8 && geo.K == 14) FN<8, 14>(__VA_ARGS__);     \
    else if (geo.G == 16 && geo.K == 9) FN<16, 9>(__VA_ARGS__);     \
    else if (geo.G == 16 && geo.K == 14) FN<16, 14>(__VA_ARGS__);   \
    else FN<16, 18>(__VA_ARGS__);                                   \
  } while (0)

// modexp_kernel additionally exists in two "latency" geometries, 16 lanes per element with 5 / 7 limbs each
#define GEO_DISPATCH_MODEXP(FN, geo, ...)                           \
  do {                                                              \
    if (geo.G == 16 && geo.K == 5) FN<16, 5>(__VA_ARGS__);          \
    else if (geo.G == 16 && geo.K == 7) FN<16, 7>(__VA_ARGS__);     \
    else GEO_DISPATCH(FN, geo, __VA_ARGS__);                        \
  } while (0)

// Small batches do not fill the chip: a launch of fewer wavefronts than SIMDs runs as long as ONE wavefront's
// serial chain of ~1200 multiplications.  Spreading an element over 16 lanes instead of 8 shortens every
// multiplication (2048-bit class: (16,5), 1190 instructions instead of 1590 for (8,9); 3072-bit class: (16,7)
// for (8,14), same L and therefore the same context).  Used while the 16-lane split still fits one wavefront
// per SIMD; (16,5) has L = 80, so it needs its own Montgomery context.
constexpr size_t kSimds = 256 * 4;
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
// instructions are amortised over twice as many MACs (78-80 % of the VALU slots are MACs instead of
// 61-67 %).  It pays as soon as the wide split still puts one wavefront on every SIMD: measured on the
// bench's CRT-decrypt launch (fixed-window build), (4,18) at 1 wave/SIMD 7.2 ms vs (8,9) at 2 waves/SIMD 8.1 ms.
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

uint64_t* g_wave_clocks_ptr();
void pool_release_all();

int check_ready() {
  if (!g_init) return fail(PGPU_ERR_NO_DEVICE, "pgpu_init has not been called (no GPU context)");
  return PGPU_OK;
}

// common launcher of modexp_kernel: sizes the window table and fills the shared fields
// sched: per-context sliding-window schedules of a host-known shared exponent (device arrays), or null
struct SchedRef {
  const uint16_t* p[2] = {nullptr, nullptr};
  int len[2] = {0, 0};
  int w = 0;
};
int run_modexp(pgpu::ModexpArgs& a, const GeoInfo& ctx_geo, hipStream_t s, const SchedRef* sched = nullptr) {
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
  size_t waves = modexp_waves(a.count, a.parity_waves, geo.ipw());
  size_t padded = (waves + pgpu::kWavesPerWG - 1) / pgpu::kWavesPerWG * pgpu::kWavesPerWG * geo.ipw();
  RC_TRY(g_table.ensure(padded * (entries + 1) * geo.L() * sizeof(uint32_t)));   // + the parking slot
  a.table = (uint32_t*)g_table.p;
  a.wave_clocks = g_wave_clocks_ptr();
  TimerScope t(s, PGPU_KERNEL_MODEXP);
  GEO_DISPATCH_MODEXP(launch_modexp, geo, a, s);
  HIP_TRY(hipGetLastError());
  t.stop();
  return PGPU_OK;
}

// Host <-> device copies go through a persistent pinned staging buffer: hipMemcpy straight from
// pageable memory takes an erratic 10+ ms for transfers just above 1 MiB (user-pointer pinning),
// measured with tests/cpp/ipcl_bench.cpp; a pinned bounce buffer is steady at PCIe speed.
constexpr size_t kStageBytes = (size_t)16 << 20;
void* g_stage[2] = {nullptr, nullptr};
hipStream_t g_copy_stream = nullptr;       // DMA of the staged copies, concurrent with the host-side memcpy
hipEvent_t g_stage_ev[2] = {nullptr, nullptr};
hipEvent_t g_order_ev = nullptr;           // default-stream work a staged copy has to wait for
int ensure_stage() {
  for (int i = 0; i < 2; ++i) {
    if (!g_stage[i]) HIP_TRY(hipHostMalloc(&g_stage[i], kStageBytes, hipHostMallocDefault));
    if (!g_stage_ev[i]) HIP_TRY(hipEventCreateWithFlags(&g_stage_ev[i], hipEventDisableTiming));
  }
  if (!g_copy_stream) HIP_TRY(hipStreamCreateWithFlags(&g_copy_stream, hipStreamNonBlocking));
  if (!g_order_ev) HIP_TRY(hipEventCreateWithFlags(&g_order_ev, hipEventDisableTiming));
  return PGPU_OK;
}

// A single thread moves ~8 GB/s between pageable memory and the pinned bounce buffer -- less than the DMA
// engine behind it -- so large copies are split over a few persistent helper threads (never joined: they
// sleep on a condition variable and die with the process).
class CopyPool {
 public:
  static constexpr int kHelpers = 3;
  static CopyPool& get() {
    static CopyPool* p = new CopyPool();
    return *p;
  }
  void copy(void* dst, const void* src, size_t n) {
    if (n < ((size_t)2 << 20)) { std::memcpy(dst, src, n); return; }
    const size_t per = (n / (kHelpers + 1) + 63) & ~(size_t)63;
    {
      std::lock_guard<std::mutex> lk(m_);
      dst_ = (char*)dst; src_ = (const char*)src; n_ = n; per_ = per;
      pending_ = kHelpers;
      ++gen_;
    }
    cv_.notify_all();
    std::memcpy(dst, src, std::min(per, n));                     // slice 0 on the calling thread
    std::unique_lock<std::mutex> lk(m_);
    done_.wait(lk, [&] { return pending_ == 0; });
  }

 private:
  CopyPool() {
    for (int i = 0; i < kHelpers; ++i) std::thread([this, i] { run(i + 1); }).detach();
  }
  void run(int slice) {
    unsigned seen = 0;
    for (;;) {
      std::unique_lock<std::mutex> lk(m_);
      cv_.wait(lk, [&] { return gen_ != seen; });
      seen = gen_;
      char* d = dst_; const char* s = src_; const size_t n = n_, per = per_;
      lk.unlock();
      const size_t lo = std::min(n, per * (size_t)slice), hi = std::min(n, per * (size_t)(slice + 1));
      if (hi > lo) std::memcpy(d + lo, s + lo, hi - lo);
      lk.lock();
      if (--pending_ == 0) done_.notify_one();
    }
  }
  std::mutex m_;
  std::condition_variable cv_, done_;
  char* dst_ = nullptr;
  const char* src_ = nullptr;
  size_t n_ = 0, per_ = 0;
  unsigned gen_ = 0;
  int pending_ = 0;
};

// host -> device: chunk i+1 is packed into the other pinned buffer while chunk i is on the wire
int staged_h2d(void* d_dst, const void* h_src, size_t bytes) {
  RC_TRY(ensure_stage());
  // d_dst may be a pooled buffer that a kernel still queued on the default stream reads (device buffers
  // go back to the pool at once): the copy stream must not overtake that work
  HIP_TRY(hipEventRecord(g_order_ev, nullptr));
  HIP_TRY(hipStreamWaitEvent(g_copy_stream, g_order_ev, 0));
  size_t off = 0;
  for (int i = 0; off < bytes; ++i) {
    const int b = i & 1;
    const size_t n = std::min(kStageBytes, bytes - off);
    if (i >= 2) HIP_TRY(hipEventSynchronize(g_stage_ev[b]));      // the DMA that last read this buffer is done
    CopyPool::get().copy(g_stage[b], (const char*)h_src + off, n);
    HIP_TRY(hipMemcpyAsync((char*)d_dst + off, g_stage[b], n, hipMemcpyHostToDevice, g_copy_stream));
    HIP_TRY(hipEventRecord(g_stage_ev[b], g_copy_stream));
    off += n;
  }
  HIP_TRY(hipStreamSynchronize(g_copy_stream));
  return PGPU_OK;
}
// device -> host (after the work queued on the default stream): chunk i+1 is on the wire while chunk i is
// unpacked from its pinned buffer
int staged_d2h(void* h_dst, const void* d_src, size_t bytes) {
  RC_TRY(ensure_stage());
  hipError_t e = hipStreamSynchronize(nullptr);
  if (e != hipSuccess) return fail(PGPU_ERR_HIP, std::string("kernel failed: ") + hipGetErrorString(e));
  auto issue = [&](int i) -> hipError_t {
    const size_t off = (size_t)i * kStageBytes, n = std::min(kStageBytes, bytes - off);
    hipError_t r = hipMemcpyAsync(g_stage[i & 1], (const char*)d_src + off, n, hipMemcpyDeviceToHost, g_copy_stream);
    return r == hipSuccess ? hipEventRecord(g_stage_ev[i & 1], g_copy_stream) : r;
  };
  const int chunks = (int)((bytes + kStageBytes - 1) / kStageBytes);
  if (chunks > 0) e = issue(0);
  for (int i = 0; i < chunks && e == hipSuccess; ++i) {
    if (i + 1 < chunks) e = issue(i + 1);                         // its buffer was unpacked in iteration i-1
    if (e == hipSuccess) e = hipEventSynchronize(g_stage_ev[i & 1]);
    if (e != hipSuccess) break;
    const size_t off = (size_t)i * kStageBytes, n = std::min(kStageBytes, bytes - off);
    CopyPool::get().copy((char*)h_dst + off, g_stage[i & 1], n);
  }
  if (e != hipSuccess) return fail(PGPU_ERR_HIP, std::string("D2H copy failed: ") + hipGetErrorString(e));
  return PGPU_OK;
}

// staging helper for the synchronous host-pointer entry points
struct DevBuf {   // staging buffers come from the caching allocator (pgpu_dev_alloc)
  void* p = nullptr;
  ~DevBuf() {
    if (p) pgpu_dev_free(p);
  }
  int alloc(size_t bytes) { return pgpu_dev_alloc(bytes, &p); }
  int from_host(const void* src, size_t bytes) {
    RC_TRY(alloc(bytes));
    return staged_h2d(p, src, bytes);
  }
  int to_host(void* dst, size_t bytes) { return staged_d2h(dst, p, bytes); }
};

}  // namespace

// ---------- Paillier key objects ----------
struct pgpu_pubkey {
  int n_words = 0;
  BigNumber n;
  bool djn = false;
  std::shared_ptr<ModCtx> nsq;  // modulus n^2, with nr = n*R mod n^2
  DevBlob d_hs;                 // DJN: hs, 2*n_words words
  DevBlob d_n;                  // plain: the exponent n, n_words words
  ExpSchedule sched_n;          // plain: sliding-window schedule of n (r^n mod n^2)
  // fixed-base table for hs^r (built lazily, grown when a longer exponent shows up)
  mutable Workspace fb_table;
  mutable int fb_nwin = 0;
  mutable int fb_w = 0;
  mutable size_t fb_elems = 0;  // elements encrypted with this key so far (window policy)
  ~pgpu_pubkey() { fb_table.release(); }
};

struct pgpu_privkey {
  int n_words = 0;              // words of n (= words of p^2, q^2 rows)
  int pq_words = 0;
  GeoInfo geo_exp{};            // geometry of the two half-width exponentiations
  GeoInfo geo_crt{};            // geometry of the recombination kernel
  std::shared_ptr<ModCtx> p2, q2;   // moduli p^2, q^2 (fc = hp / hq, r2s set)
  GeoInfo geo_lat{};            // latency geometry of the exponentiations (== geo_exp if there is none)
  std::shared_ptr<ModCtx> p2l, q2l; // the same moduli for geo_lat (aliases of p2, q2 when L is equal)
  std::shared_ptr<ModCtx> cM, cQ;   // auxiliary modulus M, modulus q (CRT geometry)
  DevBlob d_exps;               // [2][pq_words]: p-1, q-1
  int exp_bits = 0;
  ExpSchedule sched[2];         // sliding-window schedules of p-1 and q-1
  DevBlob d_crt32;              // cp | cq | pinvR | pRM  (29-bit limbs, CRT geometry)
  DevBlob d_crt64;              // hp | hq | p^2 | q^2 | q   (n_words words each)
};

extern "C" {

int pgpu_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int pgpu_init(int device) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  int n = pgpu_device_count();
  if (n <= 0) return fail(PGPU_ERR_NO_DEVICE, "no HIP device visible");
  if (device >= n) return fail(PGPU_ERR_INVALID_PARAM, "device ordinal out of range");
  if (device >= 0) HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipGetDevice(&g_device));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, g_device));
  g_devname = std::string(prop.name) + " (" + prop.gcnArchName + ")";
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(PGPU_ERR_NO_DEVICE, "device is not gfx950: " + g_devname);
  g_init = true;
  return PGPU_OK;
}

void pgpu_shutdown(void) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  if (!g_init) return;
  (void)hipDeviceSynchronize();
  g_ctx_cache.clear();
  g_table.release();
  g_vbuf.release();
  pool_release_all();
  for (int i = 0; i < 2; ++i) {
    if (g_stage[i]) (void)hipHostFree(g_stage[i]);
    g_stage[i] = nullptr;
    if (g_stage_ev[i]) (void)hipEventDestroy(g_stage_ev[i]);
    g_stage_ev[i] = nullptr;
  }
  if (g_copy_stream) (void)hipStreamDestroy(g_copy_stream);
  g_copy_stream = nullptr;
  if (g_order_ev) (void)hipEventDestroy(g_order_ev);
  g_order_ev = nullptr;
  for (auto& t : g_timed) { g_event_pool.push_back(t.e0); g_event_pool.push_back(t.e1); }
  g_timed.clear();
  for (hipEvent_t e : g_event_pool) (void)hipEventDestroy(e);
  g_event_pool.clear();
  g_init = false;
}

int pgpu_is_initialized(void) { return g_init ? 1 : 0; }
const char* pgpu_last_error(void) { return g_err.c_str(); }
const char* pgpu_device_name(void) { return g_devname.c_str(); }

int pgpu_kernel_geometry(int in_words, int mod_bits, size_t count, int* lanes, int* limbs) {
  if (in_words <= 0 || mod_bits <= 1 || !lanes || !limbs)
    return fail(PGPU_ERR_INVALID_PARAM, "pgpu_kernel_geometry: bad argument");
  const GeoInfo* geo = pick_geo(in_words, mod_bits);
  if (!geo) return fail(PGPU_ERR_UNSUPPORTED, "modulus wider than the compiled kernel geometries");
  const GeoInfo lat = latency_geo(*geo);
  const GeoInfo g = use_latency_geo(lat, *geo, count) ? lat : launch_geo(*geo, count);
  *lanes = g.G;
  *limbs = g.K;
  return PGPU_OK;
}

int pgpu_set_fixed_base_window(int w) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  if (w < 0 || w > 12) return fail(PGPU_ERR_INVALID_PARAM, "fixed-base window must be 0..12");
  g_fb_window = w;
  g_fb_window_explicit = true;
  return PGPU_OK;
}

// diagnostics (tools/wave_spread.py): device buffer that receives per-wave start/end clocks of the
// next modexp_kernel launches; null switches it off.  Not part of the public header.
uint64_t* g_wave_clocks = nullptr;
extern "C" void pgpu_debug_set_wave_clocks(uint64_t* d_buf) { g_wave_clocks = d_buf; }
}  // extern "C" (reopened below)
namespace { uint64_t* g_wave_clocks_ptr() { return g_wave_clocks; } }
extern "C" {

int pgpu_set_timing(int enabled) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  g_timing = enabled != 0;
  return PGPU_OK;
}

int pgpu_timing_collect(int* kinds, double* ms, int max_entries) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  int n = 0;
  for (auto& t : g_timed) {
    float v = 0;
    if (hipEventSynchronize(t.e1) == hipSuccess && hipEventElapsedTime(&v, t.e0, t.e1) == hipSuccess &&
        n < max_entries && kinds && ms) {
      kinds[n] = t.kind;
      ms[n] = v;
      ++n;
    }
    g_event_pool.push_back(t.e0);
    g_event_pool.push_back(t.e1);
  }
  g_timed.clear();
  return n;
}

// ===================== device buffers =====================
// Caching allocator: batches come and go at the same few sizes, and hipMalloc/hipFree of > 1 MiB
// blocks cost milliseconds (hipFree also synchronises the device).  Freed blocks are kept in
// per-size free lists (sizes rounded to 64 KiB) and handed out again; all work on them is stream
// ordered on the default stream, so reuse is safe.  At most kPoolCap bytes are kept idle.
namespace {
constexpr size_t kPoolGranule = 64 * 1024, kPoolCap = (size_t)4 << 30;
std::map<size_t, std::vector<void*>> g_pool_free;
std::map<void*, size_t> g_pool_size;   // live + cached blocks -> rounded size
size_t g_pool_idle_bytes = 0;
void pool_release_all() {
  for (auto& kv : g_pool_free)
    for (void* p : kv.second) { (void)hipFree(p); g_pool_size.erase(p); }
  g_pool_free.clear();
  g_pool_idle_bytes = 0;
}
}  // namespace

int pgpu_dev_alloc(size_t bytes, void** out) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  RC_TRY(check_ready());
  if (!out) return fail(PGPU_ERR_INVALID_PARAM, "null output pointer");
  size_t rounded = (std::max<size_t>(bytes, 1) + kPoolGranule - 1) / kPoolGranule * kPoolGranule;
  auto it = g_pool_free.find(rounded);
  if (it != g_pool_free.end() && !it->second.empty()) {
    *out = it->second.back();
    it->second.pop_back();
    g_pool_idle_bytes -= rounded;
    return PGPU_OK;
  }
  hipError_t e = hipMalloc(out, rounded);
  if (e != hipSuccess) {   // out of memory: drop the idle blocks and retry once
    pool_release_all();
    HIP_TRY(hipMalloc(out, rounded));
  }
  g_pool_size[*out] = rounded;
  return PGPU_OK;
}
void pgpu_dev_free(void* d_ptr) {
  if (!d_ptr) return;
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  auto it = g_pool_size.find(d_ptr);
  if (it == g_pool_size.end() || !g_init) {   // not ours / after shutdown
    if (it != g_pool_size.end()) g_pool_size.erase(it);
    (void)hipFree(d_ptr);
    return;
  }
  if (g_pool_idle_bytes + it->second > kPoolCap) {
    g_pool_size.erase(it);
    (void)hipFree(d_ptr);
    return;
  }
  g_pool_free[it->second].push_back(d_ptr);
  g_pool_idle_bytes += it->second;
}
int pgpu_copy_h2d(void* d_dst, const void* h_src, size_t bytes) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  RC_TRY(check_ready());
  return staged_h2d(d_dst, h_src, bytes);
}
int pgpu_copy_d2h(void* h_dst, const void* d_src, size_t bytes) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  RC_TRY(check_ready());
  return staged_d2h(h_dst, d_src, bytes);
}

// ===================== generic modexp =====================
static int modexp_dev_impl(const uint64_t* d_base, size_t base_stride, const uint64_t* d_exp,
                           size_t exp_stride, int exp_words, int exp_bits, const uint64_t* h_mod,
                           int mod_words, uint64_t* d_out, size_t count, void* hip_stream,
                           const SchedRef* sched) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  RC_TRY(check_ready());
  if (count == 0) return PGPU_OK;
  if (!d_base || !d_exp || !d_out) return fail(PGPU_ERR_INVALID_PARAM, "null batch pointer");
  if (exp_words <= 0 || exp_bits < 0 || exp_bits > 64 * exp_words)
    return fail(PGPU_ERR_INVALID_PARAM, "exp_bits/exp_words inconsistent");
  if (base_stride != 0 && base_stride < (size_t)mod_words)
    return fail(PGPU_ERR_INVALID_PARAM, "base stride smaller than the modulus width");
  if (exp_stride != 0 && exp_stride < (size_t)exp_words)
    return fail(PGPU_ERR_INVALID_PARAM, "exponent stride smaller than exp_words");
  std::shared_ptr<ModCtx> ctx;
  RC_TRY(get_modctx(h_mod, mod_words, true, &ctx));
  GeoInfo run_geo = ctx->geo;
  const GeoInfo lat = latency_geo(ctx->geo);
  if (use_latency_geo(lat, ctx->geo, count)) {       // small batch: 16 lanes per element
    if (lat.L() != ctx->geo.L()) RC_TRY(get_modctx(h_mod, mod_words, true, &ctx, true));
    run_geo = lat;
  }
  pgpu::ModexpArgs a{};
  a.ctx[0] = a.ctx[1] = ctx->dev;
  a.nctx = 1;
  a.base = d_base;
  a.base_stride = base_stride;
  a.base_words = mod_words;
  a.exp = d_exp;
  a.exp_stride = exp_stride;
  a.exp_per_ctx = 0;
  a.exp_words = exp_words;
  a.exp_bits = exp_bits;
  a.final_mul = pgpu::FM_UNIT;
  a.out = d_out;
  a.out_stride = (size_t)mod_words;
  a.count = count;
  return run_modexp(a, run_geo, (hipStream_t)hip_stream, sched);
}

int pgpu_modexp_dev(const uint64_t* d_base, size_t base_stride, const uint64_t* d_exp,
                    size_t exp_stride, int exp_words, int exp_bits, const uint64_t* h_mod,
                    int mod_words, uint64_t* d_out, size_t count, void* hip_stream) {
  return modexp_dev_impl(d_base, base_stride, d_exp, exp_stride, exp_words, exp_bits, h_mod, mod_words,
                         d_out, count, hip_stream, nullptr);
}

int pgpu_modexp(const uint64_t* base, size_t base_stride, const uint64_t* exp, size_t exp_stride,
                int exp_words, int exp_bits, const uint64_t* mod, int mod_words, uint64_t* out,
                size_t count) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  RC_TRY(check_ready());
  if (count == 0) return PGPU_OK;
  if (!base || !exp || !out || mod_words <= 0 || exp_words <= 0)
    return fail(PGPU_ERR_INVALID_PARAM, "null batch pointer or zero width");
  size_t nb = (base_stride ? count * base_stride : (size_t)mod_words) * 8;
  size_t ne = (exp_stride ? count * exp_stride : (size_t)exp_words) * 8;
  size_t no = count * (size_t)mod_words * 8;
  DevBuf db, de, dout;
  RC_TRY(db.from_host(base, nb));
  RC_TRY(de.from_host(exp, ne));
  RC_TRY(dout.alloc(no));
  // one exponent for the whole batch, and the host has it: sliding-window schedule instead of a digit scan
  DevBuf dsched;
  SchedRef sr;
  if (exp_stride == 0 && count >= 16 && exp_bits > 8 && sliding_enabled()) {
    BigNumber e = BigNumber::fromLimbs64(exp, (size_t)exp_words);
    if (!e.isZero()) {
      sr.w = pick_sliding_window(e.BitSize());
      std::vector<uint16_t> st = sliding_schedule(e, sr.w);
      RC_TRY(dsched.from_host(st.data(), st.size() * sizeof(uint16_t)));
      sr.p[0] = (const uint16_t*)dsched.p;
      sr.len[0] = (int)st.size();
    }
  }
  RC_TRY(modexp_dev_impl((const uint64_t*)db.p, base_stride, (const uint64_t*)de.p, exp_stride, exp_words,
                         exp_bits, mod, mod_words, (uint64_t*)dout.p, count, nullptr, sr.p[0] ? &sr : nullptr));
  return dout.to_host(out, no);
}

// ===================== modmul =====================
int pgpu_modmul_dev(const uint64_t* d_a, const uint64_t* d_b, size_t b_stride,
                    const uint64_t* h_mod, int mod_words, uint64_t* d_out, size_t count,
                    void* hip_stream) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  RC_TRY(check_ready());
  if (count == 0) return PGPU_OK;
  if (!d_a || !d_b || !d_out) return fail(PGPU_ERR_INVALID_PARAM, "null batch pointer");
  if (b_stride != 0 && b_stride < (size_t)mod_words)
    return fail(PGPU_ERR_INVALID_PARAM, "b stride smaller than the modulus width");
  std::shared_ptr<ModCtx> ctx;
  RC_TRY(get_modctx(h_mod, mod_words, false, &ctx));
  pgpu::ModmulArgs a{};
  a.ctx = ctx->dev;
  a.a = d_a;
  a.a_stride = (size_t)mod_words;
  a.b = d_b;
  a.b_stride = b_stride;
  a.in_words = mod_words;
  a.out = d_out;
  a.count = count;
  hipStream_t s = (hipStream_t)hip_stream;
  TimerScope t(s, PGPU_KERNEL_MODMUL);
  const GeoInfo lgeo = launch_geo(ctx->geo, count);
  GEO_DISPATCH(launch_modmul, lgeo, a, s);
  HIP_TRY(hipGetLastError());
  t.stop();
  return PGPU_OK;
}

int pgpu_modmul(const uint64_t* a, const uint64_t* b, size_t b_stride, const uint64_t* mod,
                int mod_words, uint64_t* out, size_t count) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  RC_TRY(check_ready());
  if (count == 0) return PGPU_OK;
  if (!a || !b || !out || mod_words <= 0)
    return fail(PGPU_ERR_INVALID_PARAM, "null batch pointer or zero width");
  size_t na = count * (size_t)mod_words * 8;
  size_t nb = (b_stride ? count * b_stride : (size_t)mod_words) * 8;
  DevBuf da, db, dout;
  RC_TRY(da.from_host(a, na));
  RC_TRY(db.from_host(b, nb));
  RC_TRY(dout.alloc(na));
  RC_TRY(pgpu_modmul_dev((const uint64_t*)da.p, (const uint64_t*)db.p, b_stride, mod, mod_words,
                         (uint64_t*)dout.p, count, nullptr));
  return dout.to_host(out, na);
}

// ===================== Paillier public key / encrypt =====================
int pgpu_pubkey_create(const uint64_t* n, int n_words, const uint64_t* hs_or_null,
                       pgpu_pubkey** out) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  RC_TRY(check_ready());
  if (!n || n_words <= 0 || !out) return fail(PGPU_ERR_INVALID_PARAM, "null key material");
  std::unique_ptr<pgpu_pubkey> k(new pgpu_pubkey);
  k->n_words = n_words;
  k->n = BigNumber::fromLimbs64(n, (size_t)n_words);
  if (!k->n.IsOdd()) return fail(PGPU_ERR_EVEN_MODULUS, "n must be odd");
  BigNumber nsq = k->n * k->n;
  const GeoInfo* geo = pick_geo(2 * n_words, nsq.BitSize());
  if (!geo) return fail(PGPU_ERR_UNSUPPORTED, "key wider than the compiled kernel geometries");
  CtxExtras ex;
  ex.nr_n = &k->n;
  ex.unit_q = true;
  RC_TRY(build_modctx(nsq, 2 * n_words, *geo, ex, &k->nsq));
  if (hs_or_null) {
    k->djn = true;
    RC_TRY(k->d_hs.upload(hs_or_null, (size_t)2 * n_words * 8));
  }
  RC_TRY(k->d_n.upload(n, (size_t)n_words * 8));
  if (sliding_enabled()) RC_TRY(make_schedule(k->n, pick_sliding_window(k->n.BitSize()), &k->sched_n));
  *out = k.release();
  return PGPU_OK;
}

void pgpu_pubkey_destroy(pgpu_pubkey* key) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  delete key;
}

int pgpu_paillier_encrypt_dev(const pgpu_pubkey* key, const uint64_t* d_m, size_t m_stride,
                              int m_words, const uint64_t* d_r, size_t r_stride, int r_words,
                              int r_bits, uint64_t* d_c, size_t count, void* hip_stream) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  RC_TRY(check_ready());
  if (!key) return fail(PGPU_ERR_INVALID_PARAM, "null key");
  if (count == 0) return PGPU_OK;
  if (!d_m || !d_r || !d_c) return fail(PGPU_ERR_INVALID_PARAM, "null batch pointer");
  const int W = 2 * key->n_words;
  if (m_words <= 0 || m_words > W || m_stride < (size_t)m_words)
    return fail(PGPU_ERR_INVALID_PARAM, "plaintext width/stride invalid");
  if (r_words <= 0 || r_stride < (size_t)r_words)
    return fail(PGPU_ERR_INVALID_PARAM, "random width/stride invalid");
  if (key->djn && (r_bits < 0 || r_bits > 64 * r_words))
    return fail(PGPU_ERR_INVALID_PARAM, "r_bits/r_words inconsistent");
  int fbw = fixed_base_window();
  if (key->djn && fbw > 8 && !g_fb_window_explicit && key->fb_elems + count < kFbGrowAfter) fbw = 8;
  if (key->djn) key->fb_elems += count;
  if (key->djn && fbw > 0) {
    // hs is a key constant: fixed-base windowing, no squarings (kernels.hpp: fb_encrypt_kernel)
    hipStream_t s = (hipStream_t)hip_stream;
    const GeoInfo& geo = key->nsq->geo;
    const int nwin = std::max(1, (r_bits + fbw - 1) / fbw);
    if (key->fb_w != fbw || key->fb_nwin < nwin) {
      RC_TRY(key->fb_table.ensure((size_t)nwin * ((size_t)1 << fbw) * geo.L() * sizeof(uint32_t)));
      pgpu::FixedBaseBuildArgs b{};
      b.ctx = key->nsq->dev;
      b.base = (const uint64_t*)key->d_hs.p;
      b.table = (uint32_t*)key->fb_table.p;
      b.nwin = nwin;
      b.w = fbw;
      GEO_DISPATCH(launch_fb_build, geo, b, s);
      HIP_TRY(hipGetLastError());
      key->fb_nwin = nwin;
      key->fb_w = fbw;
    }
    pgpu::FixedBaseArgs f{};
    f.ctx = key->nsq->dev;
    f.table = (const uint32_t*)key->fb_table.p;
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
    TimerScope t(s, PGPU_KERNEL_FB_ENCRYPT);
    const GeoInfo lgeo = launch_geo(geo, count);
    GEO_DISPATCH(launch_fb_encrypt, lgeo, f, s);
    HIP_TRY(hipGetLastError());
    t.stop();
    return PGPU_OK;
  }
  pgpu::ModexpArgs a{};
  a.ctx[0] = a.ctx[1] = key->nsq->dev;
  a.nctx = 1;
  if (key->djn) {  // hs^r: shared base, per-element exponent (pub_key.cpp:51-64)
    a.base = (const uint64_t*)key->d_hs.p;
    a.base_stride = 0;
    a.base_words = W;
    a.exp = d_r;
    a.exp_stride = r_stride;
    a.exp_words = r_words;
    a.exp_bits = r_bits;
  } else {         // r^n: per-element base, shared exponent n (pub_key.cpp:66-80)
    if (r_words > W) return fail(PGPU_ERR_INVALID_PARAM, "random wider than n^2");
    a.base = d_r;
    a.base_stride = r_stride;
    a.base_words = r_words;
    a.exp = (const uint64_t*)key->d_n.p;
    a.exp_stride = 0;
    a.exp_words = key->n_words;
    a.exp_bits = key->n.BitSize();
  }
  a.exp_per_ctx = 0;
  a.final_mul = pgpu::FM_PAILLIER_G;
  a.fm_words = d_m;
  a.fm_stride = m_stride;
  a.fm_nwords = m_words;
  a.out = d_c;
  a.out_stride = (size_t)W;
  a.count = count;
  SchedRef sr;
  if (!key->djn && key->sched_n.dev.p) {     // r^n: the exponent is the key constant n
    sr.p[0] = (const uint16_t*)key->sched_n.dev.p;
    sr.len[0] = key->sched_n.len;
    sr.w = key->sched_n.w;
  }
  return run_modexp(a, key->nsq->geo, (hipStream_t)hip_stream, sr.p[0] ? &sr : nullptr);
}

int pgpu_paillier_encrypt(const pgpu_pubkey* key, const uint64_t* m, size_t m_stride, int m_words,
                          const uint64_t* r, size_t r_stride, int r_words, int r_bits,
                          uint64_t* c, size_t count) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  RC_TRY(check_ready());
  if (!key) return fail(PGPU_ERR_INVALID_PARAM, "null key");
  if (count == 0) return PGPU_OK;
  if (!m || !r || !c) return fail(PGPU_ERR_INVALID_PARAM, "null batch pointer");
  size_t no = count * (size_t)2 * key->n_words * 8;
  DevBuf dm, dr, dc;
  RC_TRY(dm.from_host(m, count * m_stride * 8));
  RC_TRY(dr.from_host(r, count * r_stride * 8));
  RC_TRY(dc.alloc(no));
  RC_TRY(pgpu_paillier_encrypt_dev(key, (const uint64_t*)dm.p, m_stride, m_words,
                                   (const uint64_t*)dr.p, r_stride, r_words, r_bits,
                                   (uint64_t*)dc.p, count, nullptr));
  return dc.to_host(c, no);
}

// ===================== Paillier private key / CRT decrypt =====================
// host-side single modexp through the GPU engine (key precomputation, pri_key.cpp:159-167)
static int host_modexp(const BigNumber& base, const BigNumber& exp, const BigNumber& mod,
                       BigNumber* out) {
  int mw = (mod.BitSize() + 63) / 64, ew = std::max(1, (exp.BitSize() + 63) / 64);
  std::vector<uint64_t> b(mw), e(ew), m(mw), o(mw);
  (base % mod).toLimbs64(b.data(), mw);
  exp.toLimbs64(e.data(), ew);
  mod.toLimbs64(m.data(), mw);
  RC_TRY(pgpu_modexp(b.data(), mw, e.data(), ew, ew, exp.isZero() ? 0 : exp.BitSize(), m.data(), mw,
                     o.data(), 1));
  *out = BigNumber::fromLimbs64(o.data(), mw);
  return PGPU_OK;
}

int pgpu_privkey_create(const uint64_t* p_in, const uint64_t* q_in, int pq_words,
                        pgpu_privkey** out) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  RC_TRY(check_ready());
  if (!p_in || !q_in || pq_words <= 0 || !out) return fail(PGPU_ERR_INVALID_PARAM, "null key material");
  BigNumber p = BigNumber::fromLimbs64(p_in, (size_t)pq_words);
  BigNumber q = BigNumber::fromLimbs64(q_in, (size_t)pq_words);
  if (q < p) std::swap(p, q);  // pri_key.cpp:19-22
  if (p == q) return fail(PGPU_ERR_NOT_INVERTIBLE, "PrivateKey: p and q are same");
  if (!p.IsOdd() || !q.IsOdd() || p <= BigNumber::Two())
    return fail(PGPU_ERR_EVEN_MODULUS, "p and q must be odd primes");
  std::unique_ptr<pgpu_privkey> k(new pgpu_privkey);
  const BigNumber n = p * q, g = n + 1;
  const int nw = (n.BitSize() + 63) / 64;  // words of n; p^2, q^2 rows use the same width
  k->n_words = nw;
  k->pq_words = pq_words;
  const BigNumber psq = p * p, qsq = q * q;
  const BigNumber pm1 = p - 1, qm1 = q - 1;
  if (psq.BitSize() > 64 * nw || qsq.BitSize() > 64 * nw)
    return fail(PGPU_ERR_INVALID_PARAM, "p and q differ too much in size");
  // hp = L_p(g^(p-1) mod p^2)^-1 mod p   (computeHfun, pri_key.cpp:159-167)
  BigNumber hp, hq;
  try {
    BigNumber t;
    RC_TRY(host_modexp(g % psq, pm1, psq, &t));
    hp = p.InverseMul((t - 1) / p);
    RC_TRY(host_modexp(g % qsq, qm1, qsq, &t));
    hq = q.InverseMul((t - 1) / q);
  } catch (const std::exception& e) {
    return fail(PGPU_ERR_NOT_INVERTIBLE, std::string("PrivateKey precompute: ") + e.what());
  }
  // half-width exponentiation contexts: inputs are 2*nw-word ciphertexts reduced on load
  const GeoInfo* ge = pick_geo(nw, std::max(psq.BitSize(), qsq.BitSize()));
  if (!ge) return fail(PGPU_ERR_UNSUPPORTED, "key wider than the compiled kernel geometries");
  k->geo_exp = *ge;
  CtxExtras exp_p, exp_q;
  exp_p.want_r2s = exp_q.want_r2s = true;
  exp_p.unit_q = exp_q.unit_q = true;
  exp_p.fc = &hp;
  exp_q.fc = &hq;
  RC_TRY(build_modctx(psq, nw, *ge, exp_p, &k->p2));
  RC_TRY(build_modctx(qsq, nw, *ge, exp_q, &k->q2));
  k->geo_lat = latency_geo(*ge);
  if (k->geo_lat.L() == ge->L()) {
    k->p2l = k->p2;
    k->q2l = k->q2;
  } else {
    RC_TRY(build_modctx(psq, nw, k->geo_lat, exp_p, &k->p2l));
    RC_TRY(build_modctx(qsq, nw, k->geo_lat, exp_q, &k->q2l));
  }
  std::vector<uint64_t> exps((size_t)2 * pq_words, 0);
  pm1.toLimbs64(exps.data(), pq_words);
  qm1.toLimbs64(exps.data() + pq_words, pq_words);
  RC_TRY(k->d_exps.upload(exps.data(), exps.size() * 8));
  k->exp_bits = std::max(pm1.BitSize(), qm1.BitSize());
  if (sliding_enabled()) {
    const int sw = pick_sliding_window(k->exp_bits);
    RC_TRY(make_schedule(pm1, sw, &k->sched[0]));
    RC_TRY(make_schedule(qm1, sw, &k->sched[1]));
  }

  // recombination: auxiliary modulus M = 2^(29*(L-1)) - 1 must exceed n (exact u*p product)
  const GeoInfo* gc = nullptr;
  for (const GeoInfo& gg : kGeos)
    if (pgpu::kLimbBits * (gg.L() - 1) >= n.BitSize() + 2 && gg.rbits() >= 64 * nw) { gc = &gg; break; }
  if (!gc) return fail(PGPU_ERR_UNSUPPORTED, "key wider than the compiled kernel geometries");
  k->geo_crt = *gc;
  const int Lc = gc->L();
  const BigNumber M = pow2(pgpu::kLimbBits * (Lc - 1)) - 1;
  if (M.gcd(p) != BigNumber::One() || M.gcd(q) != BigNumber::One())
    return fail(PGPU_ERR_NOT_INVERTIBLE, "auxiliary modulus shares a factor with the key");
  const int mwM = (M.BitSize() + 63) / 64;
  RC_TRY(build_modctx(M, mwM, *gc, CtxExtras(), &k->cM));
  RC_TRY(build_modctx(q, nw, *gc, CtxExtras(), &k->cQ));
  const BigNumber Rc = pow2(gc->rbits());
  const BigNumber pinv_q = q.InverseMul(p);  // p^-1 mod q (pri_key.cpp:27)
  std::vector<uint32_t> c32((size_t)4 * Lc);
  to_limbs29((M.InverseMul(p) * Rc) % M, Lc, c32.data());
  to_limbs29((M.InverseMul(q) * Rc) % M, Lc, c32.data() + Lc);
  to_limbs29((pinv_q * Rc) % q, Lc, c32.data() + 2 * Lc);
  to_limbs29((p * Rc) % M, Lc, c32.data() + 3 * Lc);
  RC_TRY(k->d_crt32.upload(c32.data(), c32.size() * 4));
  const int pad = gc->w64() + 1;  // rows padded so word helpers can run over W64 words
  std::vector<uint64_t> c64((size_t)5 * pad, 0);
  hp.toLimbs64(c64.data(), pad);
  hq.toLimbs64(c64.data() + pad, pad);
  psq.toLimbs64(c64.data() + 2 * pad, pad);
  qsq.toLimbs64(c64.data() + 3 * pad, pad);
  q.toLimbs64(c64.data() + 4 * pad, pad);
  RC_TRY(k->d_crt64.upload(c64.data(), c64.size() * 8));
  *out = k.release();
  return PGPU_OK;
}

void pgpu_privkey_destroy(pgpu_privkey* key) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  delete key;
}

int pgpu_paillier_decrypt_crt_dev(const pgpu_privkey* key, const uint64_t* d_c, uint64_t* d_m,
                                  size_t count, void* hip_stream) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  RC_TRY(check_ready());
  if (!key) return fail(PGPU_ERR_INVALID_PARAM, "null key");
  if (count == 0) return PGPU_OK;
  if (!d_c || !d_m) return fail(PGPU_ERR_INVALID_PARAM, "null batch pointer");
  const int nw = key->n_words;
  hipStream_t s = (hipStream_t)hip_stream;
  RC_TRY(g_vbuf.ensure(2 * count * (size_t)nw * 8));
  // stage 1: V[2i] = c^(p-1)*hp mod p^2, V[2i+1] = c^(q-1)*hq mod q^2   (2*count instances)
  pgpu::ModexpArgs a{};
  const bool lat = use_latency_geo(key->geo_lat, key->geo_exp, 2 * count);
  a.ctx[0] = lat ? key->p2l->dev : key->p2->dev;
  a.ctx[1] = lat ? key->q2l->dev : key->q2->dev;
  a.nctx = 2;
  a.base = d_c;
  a.base_stride = (size_t)2 * nw;
  a.base_words = 2 * nw;
  a.exp = (const uint64_t*)key->d_exps.p;
  a.exp_stride = (size_t)key->pq_words;
  a.exp_per_ctx = 1;
  a.exp_words = key->pq_words;
  a.exp_bits = key->exp_bits;
  a.final_mul = pgpu::FM_CTX_CONST;
  a.out = (uint64_t*)g_vbuf.p;
  a.out_stride = (size_t)nw;
  a.count = 2 * count;
  SchedRef sr;
  for (int i = 0; i < 2; ++i) {
    sr.p[i] = (const uint16_t*)key->sched[i].dev.p;
    sr.len[i] = key->sched[i].len;
  }
  sr.w = key->sched[0].w;
  RC_TRY(run_modexp(a, lat ? key->geo_lat : key->geo_exp, s, &sr));
  // stage 2: L function, CRT
  const int Lc = key->geo_crt.L(), pad = key->geo_crt.w64() + 1;
  pgpu::CrtArgs c{};
  c.ctxM = key->cM->dev;
  c.ctxQ = key->cQ->dev;
  const uint32_t* c32 = (const uint32_t*)key->d_crt32.p;
  c.cp = c32;
  c.cq = c32 + Lc;
  c.pinvR = c32 + 2 * Lc;
  c.pRM = c32 + 3 * Lc;
  const uint64_t* c64 = (const uint64_t*)key->d_crt64.p;
  c.hp64 = c64;
  c.hq64 = c64 + pad;
  c.p2_64 = c64 + 2 * pad;
  c.q2_64 = c64 + 3 * pad;
  c.q64 = c64 + 4 * pad;
  c.v = (const uint64_t*)g_vbuf.p;
  c.vw = nw;
  c.out = d_m;
  c.out_words = nw;
  c.count = count;
  TimerScope tc(s, PGPU_KERNEL_CRT);
  GEO_DISPATCH(launch_crt, key->geo_crt, c, s);
  HIP_TRY(hipGetLastError());
  tc.stop();
  return PGPU_OK;
}

int pgpu_paillier_decrypt_crt(const pgpu_privkey* key, const uint64_t* c, uint64_t* m,
                              size_t count) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  RC_TRY(check_ready());
  if (!key) return fail(PGPU_ERR_INVALID_PARAM, "null key");
  if (count == 0) return PGPU_OK;
  if (!c || !m) return fail(PGPU_ERR_INVALID_PARAM, "null batch pointer");
  size_t nc = count * (size_t)2 * key->n_words * 8, nm = count * (size_t)key->n_words * 8;
  DevBuf dc, dm;
  RC_TRY(dc.from_host(c, nc));
  RC_TRY(dm.alloc(nm));
  RC_TRY(pgpu_paillier_decrypt_crt_dev(key, (const uint64_t*)dc.p, (uint64_t*)dm.p, count, nullptr));
  return dm.to_host(m, nm);
}

}  // extern "C"

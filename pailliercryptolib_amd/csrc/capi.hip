// pailliercryptolib_amd -- implementation of the C-ABI declared in include/pgpu.h.
// Host side: device/context management, Montgomery-constant precomputation (host BigNumber),
// geometry selection and kernel launches.  No CPU fallback: everything computes on the GPU.
#include "pgpu.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "ipcl/bignum.h"
#include "kernels.hpp"

namespace {

thread_local std::string g_err;
std::recursive_mutex g_mu;
bool g_init = false;
int g_device = -1;
std::string g_devname;
bool g_timing = false;
double g_last_ms = 0.0;
hipEvent_t g_ev0 = nullptr, g_ev1 = nullptr;

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

// ---------- geometry table ----------
struct GeoInfo { int G, K; };
const GeoInfo kGeos[] = {{2, 9}, {4, 9}, {4, 14}, {8, 9}, {8, 14}, {16, 9}, {16, 14}, {16, 18}};

// smallest geometry with R = 2^(29*G*K) >= 2^(64*mod_words) and R >= 256*N
const GeoInfo* pick_geo(int mod_words, int mod_bits) {
  for (const GeoInfo& g : kGeos) {
    int rbits = pgpu::kLimbBits * g.G * g.K;
    if (rbits >= 64 * mod_words && rbits >= mod_bits + 8) return &g;
  }
  return nullptr;
}

// ---------- modulus context ----------
struct ModCtx {
  GeoInfo geo;
  int mod_words = 0;
  void* d_blob = nullptr;  // one allocation: n | r2 | one | n64
  pgpu::ModCtxDev dev{};
  ~ModCtx() {
    if (d_blob) (void)hipFree(d_blob);
  }
};

std::map<std::vector<uint64_t>, std::shared_ptr<ModCtx>> g_ctx_cache;

// workspace for window tables (grow-only)
void* g_table = nullptr;
size_t g_table_bytes = 0;

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

int get_modctx(const uint64_t* mod, int mod_words, std::shared_ptr<ModCtx>* out) {
  if (!mod || mod_words <= 0) return fail(PGPU_ERR_INVALID_PARAM, "modulus is null/empty");
  if (!(mod[0] & 1)) return fail(PGPU_ERR_EVEN_MODULUS, "modulus must be odd");
  std::vector<uint64_t> key(mod, mod + mod_words);
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
  const int L = geo->G * geo->K;
  const int rbits = pgpu::kLimbBits * L;
  const int W64 = (rbits + 63) / 64;

  BigNumber R = pow2(rbits);
  BigNumber Rm = R % N;
  BigNumber R2 = (Rm * Rm) % N;
  // n0inv = -N^-1 mod 2^29 by Newton iteration on the low word
  uint32_t n0 = (uint32_t)(mod[0] & pgpu::kLimbMask);
  uint32_t inv = n0;                       // correct to 3 bits (n0 odd)
  for (int i = 0; i < 5; ++i) inv *= 2u - n0 * inv;
  uint32_t n0inv = (0u - inv) & pgpu::kLimbMask;

  std::vector<uint32_t> h((size_t)3 * L);
  to_limbs29(N, L, h.data());
  to_limbs29(R2, L, h.data() + L);
  to_limbs29(Rm, L, h.data() + 2 * L);
  std::vector<uint64_t> n64((size_t)W64 + 1, 0);
  for (int i = 0; i < mod_words && i <= W64; ++i) n64[i] = mod[i];

  auto ctx = std::make_shared<ModCtx>();
  ctx->geo = *geo;
  ctx->mod_words = mod_words;
  size_t bytes32 = h.size() * sizeof(uint32_t);
  size_t off64 = (bytes32 + 15) & ~(size_t)15;
  size_t total = off64 + n64.size() * sizeof(uint64_t);
  HIP_TRY(hipMalloc(&ctx->d_blob, total));
  HIP_TRY(hipMemcpy(ctx->d_blob, h.data(), bytes32, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy((char*)ctx->d_blob + off64, n64.data(), n64.size() * sizeof(uint64_t),
                    hipMemcpyHostToDevice));
  uint32_t* d32 = (uint32_t*)ctx->d_blob;
  ctx->dev.n = d32;
  ctx->dev.r2 = d32 + L;
  ctx->dev.one = d32 + 2 * L;
  ctx->dev.n64 = (const uint64_t*)((char*)ctx->d_blob + off64);
  ctx->dev.n0inv = n0inv;
  ctx->dev.mod_words = mod_words;
  g_ctx_cache[key] = ctx;
  *out = ctx;
  return PGPU_OK;
}

int ensure_table(size_t bytes) {
  if (bytes <= g_table_bytes) return PGPU_OK;
  if (g_table) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipFree(g_table));
    g_table = nullptr;
    g_table_bytes = 0;
  }
  HIP_TRY(hipMalloc(&g_table, bytes));
  g_table_bytes = bytes;
  return PGPU_OK;
}

// fixed-window width: canonical w = 5 for long exponents; shorter exponents pick the w that
// minimises (2^w - 2) table multiplications + ceil(e/w) window multiplications.
int pick_window(int exp_bits) {
  int best = 1;
  long best_cost = 1L << 60;
  for (int w = 1; w <= 5; ++w) {
    long cost = ((1L << w) - 2) + (exp_bits + w - 1) / w;
    if (cost < best_cost) { best_cost = cost; best = w; }
  }
  return best;
}

struct TimerScope {
  hipStream_t s;
  bool on;
  explicit TimerScope(hipStream_t st) : s(st), on(g_timing) {
    if (on) (void)hipEventRecord(g_ev0, s);
  }
  void stop() {
    if (!on) return;
    (void)hipEventRecord(g_ev1, s);
    (void)hipEventSynchronize(g_ev1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, g_ev0, g_ev1);
    g_last_ms = ms;
  }
};

template <int G, int K>
void launch_modexp(const pgpu::ModexpArgs& a, hipStream_t s) {
  typedef pgpu::Geo<G, K> GEO;
  unsigned blocks = (unsigned)((a.count + GEO::IPW - 1) / GEO::IPW);
  hipLaunchKernelGGL((pgpu::modexp_kernel<GEO>), dim3(blocks), dim3(pgpu::kWave), 0, s, a);
}
template <int G, int K>
void launch_modmul(const pgpu::ModmulArgs& a, hipStream_t s) {
  typedef pgpu::Geo<G, K> GEO;
  unsigned blocks = (unsigned)((a.count + GEO::IPW - 1) / GEO::IPW);
  hipLaunchKernelGGL((pgpu::modmul_kernel<GEO>), dim3(blocks), dim3(pgpu::kWave), 0, s, a);
}

#define GEO_DISPATCH(FN, geo, ...)                                  \
  do {                                                              \
    if (geo.G == 2 && geo.K == 9) FN<2, 9>(__VA_ARGS__);            \
    else if (geo.G == 4 && geo.K == 9) FN<4, 9>(__VA_ARGS__);       \
    else if (geo.G == 4 && geo.K == 14) FN<4, 14>(__VA_ARGS__);     \
    else if (geo.G == 8 && geo.K == 9) FN<8, 9>(__VA_ARGS__);       \
    else if (geo.G == 8 && geo.K == 14) FN<8, 14>(__VA_ARGS__);     \
    else if (geo.G == 16 && geo.K == 9) FN<16, 9>(__VA_ARGS__);     \
    else if (geo.G == 16 && geo.K == 14) FN<16, 14>(__VA_ARGS__);   \
    else FN<16, 18>(__VA_ARGS__);                                   \
  } while (0)

int check_ready() {
  if (!g_init) return fail(PGPU_ERR_NO_DEVICE, "pgpu_init has not been called (no GPU context)");
  return PGPU_OK;
}

}  // namespace

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
  if (!g_ev0) {
    HIP_TRY(hipEventCreate(&g_ev0));
    HIP_TRY(hipEventCreate(&g_ev1));
  }
  g_init = true;
  return PGPU_OK;
}

void pgpu_shutdown(void) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  if (!g_init) return;
  (void)hipDeviceSynchronize();
  g_ctx_cache.clear();
  if (g_table) (void)hipFree(g_table);
  g_table = nullptr;
  g_table_bytes = 0;
  if (g_ev0) { (void)hipEventDestroy(g_ev0); (void)hipEventDestroy(g_ev1); g_ev0 = g_ev1 = nullptr; }
  g_init = false;
}

int pgpu_is_initialized(void) { return g_init ? 1 : 0; }
const char* pgpu_last_error(void) { return g_err.c_str(); }
const char* pgpu_device_name(void) { return g_devname.c_str(); }

int pgpu_set_timing(int enabled) {
  g_timing = enabled != 0;
  return PGPU_OK;
}
double pgpu_last_kernel_ms(void) { return g_last_ms; }

int pgpu_modexp_dev(const uint64_t* d_base, size_t base_stride, const uint64_t* d_exp,
                    size_t exp_stride, int exp_words, int exp_bits, const uint64_t* h_mod,
                    int mod_words, uint64_t* d_out, size_t count, void* hip_stream) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  int rc = check_ready();
  if (rc) return rc;
  if (count == 0) return PGPU_OK;
  if (!d_base || !d_exp || !d_out) return fail(PGPU_ERR_INVALID_PARAM, "null batch pointer");
  if (exp_words <= 0 || exp_bits < 0 || exp_bits > 64 * exp_words)
    return fail(PGPU_ERR_INVALID_PARAM, "exp_bits/exp_words inconsistent");
  if (base_stride != 0 && base_stride < (size_t)mod_words)
    return fail(PGPU_ERR_INVALID_PARAM, "base stride smaller than the modulus width");
  if (exp_stride != 0 && exp_stride < (size_t)exp_words)
    return fail(PGPU_ERR_INVALID_PARAM, "exponent stride smaller than exp_words");
  std::shared_ptr<ModCtx> ctx;
  rc = get_modctx(h_mod, mod_words, &ctx);
  if (rc) return rc;
  const int L = ctx->geo.G * ctx->geo.K, ipw = pgpu::kWave / ctx->geo.G;
  const int w = pick_window(exp_bits);
  size_t padded = (count + ipw - 1) / ipw * ipw;
  rc = ensure_table(padded * ((size_t)1 << w) * L * sizeof(uint32_t));
  if (rc) return rc;
  pgpu::ModexpArgs a;
  a.ctx = ctx->dev;
  a.base = d_base;
  a.base_stride = base_stride;
  a.base_words = mod_words;
  a.exp = d_exp;
  a.exp_stride = exp_stride;
  a.exp_words = exp_words;
  a.exp_bits = exp_bits;
  a.window = w;
  a.out = d_out;
  a.table = (uint32_t*)g_table;
  a.count = count;
  hipStream_t s = (hipStream_t)hip_stream;
  TimerScope t(s);
  GEO_DISPATCH(launch_modexp, ctx->geo, a, s);
  HIP_TRY(hipGetLastError());
  t.stop();
  return PGPU_OK;
}

int pgpu_modexp(const uint64_t* base, size_t base_stride, const uint64_t* exp, size_t exp_stride,
                int exp_words, int exp_bits, const uint64_t* mod, int mod_words, uint64_t* out,
                size_t count) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  int rc = check_ready();
  if (rc) return rc;
  if (count == 0) return PGPU_OK;
  if (!base || !exp || !out || mod_words <= 0 || exp_words <= 0)
    return fail(PGPU_ERR_INVALID_PARAM, "null batch pointer or zero width");
  size_t nb = (base_stride ? count * base_stride : (size_t)mod_words) * 8;
  size_t ne = (exp_stride ? count * exp_stride : (size_t)exp_words) * 8;
  size_t no = count * (size_t)mod_words * 8;
  void *db = nullptr, *de = nullptr, *dout = nullptr;
  HIP_TRY(hipMalloc(&db, nb));
  HIP_TRY(hipMalloc(&de, ne));
  HIP_TRY(hipMalloc(&dout, no));
  rc = PGPU_OK;
  if (hipMemcpy(db, base, nb, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(de, exp, ne, hipMemcpyHostToDevice) != hipSuccess)
    rc = fail(PGPU_ERR_HIP, "H2D copy failed");
  if (!rc)
    rc = pgpu_modexp_dev((const uint64_t*)db, base_stride, (const uint64_t*)de, exp_stride,
                         exp_words, exp_bits, mod, mod_words, (uint64_t*)dout, count, nullptr);
  if (!rc && hipMemcpy(out, dout, no, hipMemcpyDeviceToHost) != hipSuccess)
    rc = fail(PGPU_ERR_HIP, std::string("D2H copy / kernel failed: ") + hipGetErrorString(hipGetLastError()));
  (void)hipFree(db);
  (void)hipFree(de);
  (void)hipFree(dout);
  return rc;
}

int pgpu_modmul_dev(const uint64_t* d_a, const uint64_t* d_b, size_t b_stride,
                    const uint64_t* h_mod, int mod_words, uint64_t* d_out, size_t count,
                    void* hip_stream) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  int rc = check_ready();
  if (rc) return rc;
  if (count == 0) return PGPU_OK;
  if (!d_a || !d_b || !d_out) return fail(PGPU_ERR_INVALID_PARAM, "null batch pointer");
  if (b_stride != 0 && b_stride < (size_t)mod_words)
    return fail(PGPU_ERR_INVALID_PARAM, "b stride smaller than the modulus width");
  std::shared_ptr<ModCtx> ctx;
  rc = get_modctx(h_mod, mod_words, &ctx);
  if (rc) return rc;
  pgpu::ModmulArgs a;
  a.ctx = ctx->dev;
  a.a = d_a;
  a.a_stride = (size_t)mod_words;
  a.b = d_b;
  a.b_stride = b_stride;
  a.in_words = mod_words;
  a.out = d_out;
  a.count = count;
  hipStream_t s = (hipStream_t)hip_stream;
  TimerScope t(s);
  GEO_DISPATCH(launch_modmul, ctx->geo, a, s);
  HIP_TRY(hipGetLastError());
  t.stop();
  return PGPU_OK;
}

int pgpu_modmul(const uint64_t* a, const uint64_t* b, size_t b_stride, const uint64_t* mod,
                int mod_words, uint64_t* out, size_t count) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  int rc = check_ready();
  if (rc) return rc;
  if (count == 0) return PGPU_OK;
  if (!a || !b || !out || mod_words <= 0)
    return fail(PGPU_ERR_INVALID_PARAM, "null batch pointer or zero width");
  size_t na = count * (size_t)mod_words * 8;
  size_t nb = (b_stride ? count * b_stride : (size_t)mod_words) * 8;
  void *da = nullptr, *db = nullptr, *dout = nullptr;
  HIP_TRY(hipMalloc(&da, na));
  HIP_TRY(hipMalloc(&db, nb));
  HIP_TRY(hipMalloc(&dout, na));
  rc = PGPU_OK;
  if (hipMemcpy(da, a, na, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(db, b, nb, hipMemcpyHostToDevice) != hipSuccess)
    rc = fail(PGPU_ERR_HIP, "H2D copy failed");
  if (!rc)
    rc = pgpu_modmul_dev((const uint64_t*)da, (const uint64_t*)db, b_stride, mod, mod_words,
                         (uint64_t*)dout, count, nullptr);
  if (!rc && hipMemcpy(out, dout, na, hipMemcpyDeviceToHost) != hipSuccess)
    rc = fail(PGPU_ERR_HIP, std::string("D2H copy / kernel failed: ") + hipGetErrorString(hipGetLastError()));
  (void)hipFree(da);
  (void)hipFree(db);
  (void)hipFree(dout);
  return rc;
}

}  // extern "C"

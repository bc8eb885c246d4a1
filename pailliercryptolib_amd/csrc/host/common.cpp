// Host randomness (reference ipcl/utils/common.cpp): the OS CSPRNG (expanded by ChaCha20 for bulk requests) replaces
// the RDSEED / RDRAND / IPP-PRNG chain.  Randomness never reaches the GPU path except as data.
#include "ipcl/utils/common.hpp"

#include <sched.h>
#include <sys/random.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <condition_variable>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <random>
#include <thread>

#include "chacha20.hpp"
#include "detail.hpp"
#include "ipcl/utils/util.hpp"

namespace ipcl {

namespace detail {
namespace {
// n bytes straight from the kernel CSPRNG
void os_random(void* dst, std::size_t n) {
  unsigned char* p = static_cast<unsigned char*>(dst);
  while (n > 0) {
    ssize_t got = getrandom(p, n, 0);
    if (got < 0) {
      if (errno == EINTR) continue;
      std::random_device dev;            // (no getrandom: fall back to the C++ source, word by word)
      for (; n >= 4; n -= 4, p += 4) {
        unsigned v = dev();
        std::memcpy(p, &v, 4);
      }
      for (; n > 0; --n, ++p) *p = (unsigned char)dev();
      return;
    }
    p += got;
    n -= (std::size_t)got;
  }
}
}  // namespace

// Random bytes for obfuscator exponents and key material.  Small requests come straight from the kernel CSPRNG; a
// bulk request (a batch of 8192 DJN exponents is 1 MiB, ~3 ms through getrandom(): three times the GPU kernel it
// feeds) is the ChaCha20 key stream of a fresh 256-bit key and 96-bit nonce drawn from the kernel for this request
// alone -- the arc4random / randombytes construction; nothing is kept between calls.  (Blocks are 64 bytes and the
// counter is 32 bits: requests beyond 256 GiB would wrap, far above any batch.)
void fill_random(void* dst, std::size_t n) {
  constexpr std::size_t kBulk = 4096;
  if (n < kBulk) {
    os_random(dst, n);
    return;
  }
  unsigned char seed[44];
  os_random(seed, sizeof(seed));
  chacha20_stream(seed, seed + 32, 0, static_cast<unsigned char*>(dst), n);
  wipe(seed, sizeof(seed));
}
}  // namespace detail

void rand32u(std::vector<Ipp32u>& addr) { detail::fill_random(addr.data(), addr.size() * sizeof(Ipp32u)); }

BigNumber getRandomBN(int bits) {
  ERROR_CHECK(bits > 0, "getRandomBN: bit length must be positive");
  std::vector<Ipp32u> w((size_t)BITSIZE_WORD(bits));
  rand32u(w);
  if (bits % 32) w.back() &= (1u << (bits % 32)) - 1;
  return BigNumber(w.data(), (int)w.size());
}

namespace detail {

// ---- host thread budget (detail.hpp) ----
namespace {
int compute_thread_budget() {
  int n = 1;
  cpu_set_t set;
  CPU_ZERO(&set);
  if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::max(1, CPU_COUNT(&set));
  // cgroup v2 / v1 CPU quota: a container that sees 256 CPUs may own 16 of them
  if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char quota[32] = {0};
    long period = 0;
    if (std::fscanf(f, "%31s %ld", quota, &period) == 2 && period > 0 && std::strcmp(quota, "max") != 0) {
      const long q = std::atol(quota);
      if (q > 0) n = std::min<long>(n, std::max<long>(1, (q + period - 1) / period));
    }
    std::fclose(f);
  } else {
    long q = -1, period = -1;
    if (FILE* fq = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
      if (std::fscanf(fq, "%ld", &q) != 1) q = -1;
      std::fclose(fq);
    }
    if (FILE* fp = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
      if (std::fscanf(fp, "%ld", &period) != 1) period = -1;
      std::fclose(fp);
    }
    if (q > 0 && period > 0) n = std::min<long>(n, std::max<long>(1, (q + period - 1) / period));
  }
  n = std::min(n, 16);
  // Default: ONE thread.  Measured on the GPU hosts of this pool (EPYC 9575F, 16 cores under the quota;
  // profiles/r03_host_glue_threads.txt): the loops the team would split -- pack, unpack, copy of 8192 BigNumbers -- take
  // 60-260 us on one core and get SLOWER with more (pack 93 -> 144 us, unpack 190 -> 227 us at 16 threads; only the plain
  // copy gains, 103 -> 70 us): they are bound by the allocator and by first touches of fresh pages, not by arithmetic.
  // IPCL_NUM_THREADS=k (or OMP_NUM_THREADS) opts into the team, up to the budget above.
  int want = 1;
  for (const char* var : {"IPCL_NUM_THREADS", "OMP_NUM_THREADS"}) {
    const char* e = std::getenv(var);
    if (e && std::atoi(e) > 0) {
      want = std::atoi(e);
      break;
    }
  }
  return std::max(1, std::min(n, want));
}
}  // namespace

int max_host_threads() {
  static const int budget = compute_thread_budget();
  return budget;
}

int threads_for(std::size_t n, std::size_t grain) {
  const std::size_t by_work = n / std::max<std::size_t>(grain, 1);
  return (int)std::max<std::size_t>(1, std::min<std::size_t>((std::size_t)max_host_threads(), by_work));
}

namespace {
// The sleeping team: max_host_threads() - 1 workers, started on first use, parked on a condition variable.
// One loop at a time owns the team (try_lock: a second concurrent loop runs serially on its caller).
class Team {
 public:
  static Team& get() {
    static Team* t = new Team;   // intentionally leaked: workers may outlive static destruction order
    return *t;
  }
  bool run(std::size_t n, std::size_t chunk, int threads, const std::function<void(std::size_t, std::size_t)>& body) {
    std::unique_lock<std::mutex> owner(owner_mu_, std::try_to_lock);
    if (!owner.owns_lock()) return false;
    ensure_workers(threads - 1);
    {
      std::lock_guard<std::mutex> lk(mu_);
      body_ = &body;
      n_ = n;
      chunk_ = chunk;
      next_.store(0, std::memory_order_relaxed);
      active_ = std::min<int>(threads - 1, (int)workers_.size());
      pending_ = active_;
      ++generation_;
    }
    cv_.notify_all();
    work();   // the caller is a member of the team
    std::unique_lock<std::mutex> lk(mu_);
    done_cv_.wait(lk, [this] { return pending_ == 0; });
    body_ = nullptr;
    return true;
  }

 private:
  void work() {
    for (;;) {
      const std::size_t lo = next_.fetch_add(chunk_, std::memory_order_relaxed);
      if (lo >= n_) return;
      (*body_)(lo, std::min(n_, lo + chunk_));
    }
  }
  void ensure_workers(int want) {
    // (called by the owner of the team before it publishes a new generation: a new worker starts out having
    // "seen" the current one, so it neither replays a finished loop nor misses the next)
    uint64_t now;
    {
      std::lock_guard<std::mutex> lk(mu_);
      now = generation_;
    }
    while ((int)workers_.size() < want) {
      const int id = (int)workers_.size();
      workers_.emplace_back([this, id, now] { loop(id, now); });
      workers_.back().detach();
    }
  }
  void loop(int id, uint64_t seen) {
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return generation_ != seen; });
        seen = generation_;
        if (id >= active_) continue;   // not part of this loop's team
      }
      work();
      std::lock_guard<std::mutex> lk(mu_);
      if (--pending_ == 0) done_cv_.notify_one();
    }
  }
  std::mutex owner_mu_, mu_;
  std::condition_variable cv_, done_cv_;
  std::vector<std::thread> workers_;
  const std::function<void(std::size_t, std::size_t)>* body_ = nullptr;
  std::size_t n_ = 0, chunk_ = 1;
  std::atomic<std::size_t> next_{0};
  int active_ = 0, pending_ = 0;
  uint64_t generation_ = 0;
};
thread_local bool t_in_team_loop = false;
}  // namespace

void parallel_chunks(std::size_t n, std::size_t grain, const std::function<void(std::size_t, std::size_t)>& body) {
  const int t = threads_for(n, grain);
  if (t > 1 && !t_in_team_loop) {
    // chunks of at least `grain` elements, about four per thread (dynamic hand-out evens out uneven elements)
    const std::size_t chunk = std::max<std::size_t>(grain, (n + 4 * (std::size_t)t - 1) / (4 * (std::size_t)t));
    t_in_team_loop = true;
    const bool ran = Team::get().run(n, chunk, t, body);
    t_in_team_loop = false;
    if (ran) return;
  }
  body(0, n);
}

void wipe(void* p, std::size_t bytes) {
  volatile unsigned char* w = static_cast<volatile unsigned char*>(p);
  for (std::size_t i = 0; i < bytes; ++i) w[i] = 0;
}

// BigNumber <-> flat limb arrays: per-element loops on the sleeping team (detail.hpp).
constexpr std::size_t kGrain = 512;   // elements per thread below which a team is not worth waking

int max_bits(const std::vector<BigNumber>& v) {
  std::atomic<int> best{0};
  parallel_chunks(v.size(), 8 * kGrain, [&](std::size_t lo, std::size_t hi) {
    int b = 0;
    for (std::size_t i = lo; i < hi; ++i) b = std::max(b, v[i].isZero() ? 0 : v[i].BitSize());
    int cur = best.load(std::memory_order_relaxed);
    while (b > cur && !best.compare_exchange_weak(cur, b, std::memory_order_relaxed)) {}
  });
  return best.load();
}

bool all_fit(const std::vector<BigNumber>& v, int words) {
  std::atomic<bool> bad{false};
  parallel_chunks(v.size(), 8 * kGrain, [&](std::size_t lo, std::size_t hi) {
    for (std::size_t i = lo; i < hi; ++i)
      if (v[i].isNegative() || v[i].limbs64().size() > (size_t)words) {
        bad.store(true, std::memory_order_relaxed);
        return;
      }
  });
  return !bad.load();
}

namespace {
// bytes of limb storage (incl. block headers) a copy / unpack of `count` values of `words` limbs takes
std::size_t arena_hint(std::size_t count, std::size_t words) { return count * (words * 8 + 32); }
}  // namespace

std::vector<BigNumber> copy_texts(const std::vector<BigNumber>& v) {
  if (threads_for(v.size(), kGrain) <= 1) {
    if (v.size() < 64) return v;
    std::size_t limbs = 0;
    for (const BigNumber& x : v) limbs += x.limbs64().size();
    LimbBulkScope arena(limbs * 8 + v.size() * 32);   // one arena instead of v.size() heap blocks
    return v;
  }
  std::vector<BigNumber> out(v.size());
  parallel_chunks(v.size(), kGrain, [&](std::size_t lo, std::size_t hi) {
    std::size_t limbs = 0;
    for (std::size_t i = lo; i < hi; ++i) limbs += v[i].limbs64().size();
    LimbBulkScope arena(limbs * 8 + (hi - lo) * 32);
    for (std::size_t i = lo; i < hi; ++i) out[i] = v[i];
  });
  return out;
}

std::vector<uint64_t> pack(const std::vector<BigNumber>& v, int words) {
  std::vector<uint64_t> flat(v.size() * (size_t)words);
  std::atomic<bool> fits{true};
  parallel_for(v.size(), kGrain, [&](std::size_t i) {
    if (!v[i].toLimbs64(flat.data() + i * (size_t)words, (size_t)words)) fits.store(false, std::memory_order_relaxed);
  });
  ERROR_CHECK(fits.load(), "pack: value wider than the batch stride");
  return flat;
}

void pack_into(uint64_t* flat, const std::vector<BigNumber>& v, int words) {
  std::atomic<bool> fits{true};
  parallel_for(v.size(), kGrain, [&](std::size_t i) {
    if (!v[i].toLimbs64(flat + i * (size_t)words, (size_t)words)) fits.store(false, std::memory_order_relaxed);
  });
  ERROR_CHECK(fits.load(), "pack: value wider than the batch stride");
}

std::vector<BigNumber> unpack(const uint64_t* flat, std::size_t count, int words) {
  std::vector<BigNumber> v(count);
  if (threads_for(count, kGrain) <= 1) {
    LimbBulkScope arena(count >= 64 ? arena_hint(count, (std::size_t)words) : 0);
    for (std::size_t i = 0; i < count; ++i) v[i] = BigNumber::fromLimbs64(flat + i * (size_t)words, (size_t)words);
    return v;
  }
  // (each member of the team carves the limb blocks of its chunk out of an arena of its own)
  parallel_chunks(count, kGrain, [&](std::size_t lo, std::size_t hi) {
    LimbBulkScope arena(arena_hint(hi - lo, (std::size_t)words));
    for (std::size_t i = lo; i < hi; ++i) v[i] = BigNumber::fromLimbs64(flat + i * (size_t)words, (size_t)words);
  });
  return v;
}
std::vector<BigNumber> unpack(const std::vector<uint64_t>& flat, std::size_t count, int words) {
  return unpack(flat.data(), count, words);
}

// ---- pinned staging blocks ----
namespace {
struct PinnedPool {
  std::mutex mu;
  std::map<std::size_t, std::vector<uint64_t*>> idle;   // [size class] -> blocks
  std::size_t idle_bytes = 0;
};
PinnedPool& pinned_pool() {
  static PinnedPool* p = new PinnedPool;   // (leaked on purpose: texts may die after static destruction began)
  return *p;
}
constexpr std::size_t kPinnedIdleCap = (std::size_t)256 << 20;
std::size_t size_class(std::size_t bytes) {
  std::size_t c = 64 * 1024;
  while (c < bytes) c <<= 1;
  return c;
}
}  // namespace

std::shared_ptr<PinnedBlock> PinnedBlock::acquire(std::size_t bytes) {
  if (!pgpu_is_initialized()) return nullptr;
  const std::size_t cls = size_class(bytes);
  uint64_t* p = nullptr;
  {
    PinnedPool& pool = pinned_pool();
    std::lock_guard<std::mutex> lk(pool.mu);
    auto it = pool.idle.find(cls);
    if (it != pool.idle.end() && !it->second.empty()) {
      p = it->second.back();
      it->second.pop_back();
      pool.idle_bytes -= cls;
    }
  }
  if (p) {
    if (pgpu_host_wait(p) != PGPU_OK) {   // an upload that still reads the block: wait it out (normally long done)
      pgpu_host_free(p);
      p = nullptr;
    }
  }
  if (!p) {
    void* q = nullptr;
    if (pgpu_host_alloc(cls, &q) != PGPU_OK) return nullptr;
    p = static_cast<uint64_t*>(q);
  }
  auto b = std::make_shared<PinnedBlock>();
  b->p = p;
  b->bytes = cls;
  return b;
}

void release_pinned_pool() {
  PinnedPool& pool = pinned_pool();
  std::map<std::size_t, std::vector<uint64_t*>> dead;
  {
    std::lock_guard<std::mutex> lk(pool.mu);
    dead.swap(pool.idle);
    pool.idle_bytes = 0;
  }
  for (auto& kv : dead)
    for (uint64_t* p : kv.second) pgpu_host_free(p);
}

PinnedBlock::~PinnedBlock() {
  if (!p) return;
  PinnedPool& pool = pinned_pool();
  {
    std::lock_guard<std::mutex> lk(pool.mu);
    if (pgpu_is_initialized() && pool.idle_bytes + bytes <= kPinnedIdleCap) {
      pool.idle[bytes].push_back(p);
      pool.idle_bytes += bytes;
      return;
    }
  }
  pgpu_host_free(p);
}

std::shared_ptr<DeviceBatch> DeviceBatch::adopt(pgpu_batch* h) {
  auto b = std::make_shared<DeviceBatch>();
  b->h = h;
  b->count = pgpu_batch_count(h);
  b->words = pgpu_batch_words(h);
  return b;
}

std::shared_ptr<DeviceBatch> DeviceBatch::upload(const std::vector<uint64_t>& flat, std::size_t count, int words) {
  ensure_context();
  pgpu_batch* h = nullptr;
  IPCL_GPU_CHECK(pgpu_batch_upload(flat.data(), count, words, (size_t)words, &h), "device upload");
  return adopt(h);
}

std::shared_ptr<DeviceBatch> DeviceBatch::upload_values(const std::vector<BigNumber>& v, int words) {
  ensure_context();
  const std::size_t bytes = v.size() * (std::size_t)words * 8;
  std::shared_ptr<PinnedBlock> blk = bytes >= kEagerUploadBytes ? PinnedBlock::acquire(bytes) : nullptr;
  if (!blk) return upload(pack(v, words), v.size(), words);
  pack_into(blk->p, v, words);
  pgpu_batch* h = nullptr;
  IPCL_GPU_CHECK(pgpu_batch_upload(blk->p, v.size(), words, (size_t)words, &h), "device upload");
  auto b = adopt(h);
  b->src = std::move(blk);
  return b;
}

namespace {
// Results used in place: the pinned block is laid out as blocks of the limb allocator -- 64 bytes of arena control, then
// per value a 16-byte header and its row -- the rows arrive by ONE strided download, and the BigNumbers adopt them
// (bignum.h: LimbVec::adopt).  Nobody copies 2-4 MB out of the block; it returns to the pool when the last value that
// points into it has died.  Pinned memory a caller may hold that way is capped (beyond it: the copying path below);
// IPCL_ADOPT_RESULTS=0 turns it off.
constexpr std::size_t kAdoptCtrlWords = 8, kAdoptHeaderWords = 2;
constexpr std::size_t kAdoptCapBytes = (std::size_t)512 << 20;
constexpr std::size_t kAdoptMinBytes = (std::size_t)2 << 20;
std::atomic<std::size_t> g_adopted_bytes{0};
bool adopt_enabled() {
  static const bool on = [] { const char* e = std::getenv("IPCL_ADOPT_RESULTS"); return !e || std::atoi(e) != 0; }();
  return on;
}
struct AdoptCookie {
  std::shared_ptr<PinnedBlock> blk;
  std::size_t bytes;
};
void adopt_release(void* ck) {
  AdoptCookie* c = static_cast<AdoptCookie*>(ck);
  g_adopted_bytes.fetch_sub(c->bytes, std::memory_order_relaxed);
  delete c;   // (the block goes back to the pool, or to the driver when the pool is gone)
}
bool download_adopting(pgpu_batch* h, std::size_t count, int words, std::vector<BigNumber>* out) {
  const std::size_t stride = (std::size_t)words + kAdoptHeaderWords;
  const std::size_t bytes = (kAdoptCtrlWords + count * stride) * 8;
  if (!adopt_enabled() || g_adopted_bytes.load(std::memory_order_relaxed) + bytes > kAdoptCapBytes) return false;
  std::shared_ptr<PinnedBlock> blk = PinnedBlock::acquire(bytes);
  if (!blk) return false;
  // what is charged against the cap is what is checked: the block's size CLASS (up to twice the request), taken with one
  // fetch_add and given back when the cap would be passed (a load followed by a later add let racing threads overshoot).
  // Retention: one surviving BigNumber keeps its whole pinned block alive ("Results used in place")
  const std::size_t charge = blk->bytes;
  if (g_adopted_bytes.fetch_add(charge, std::memory_order_relaxed) + charge > kAdoptCapBytes) {
    g_adopted_bytes.fetch_sub(charge, std::memory_order_relaxed);
    return false;
  }
  uint64_t* rows = blk->p + kAdoptCtrlWords + kAdoptHeaderWords;   // row i at rows + i * stride, its header in front
  if (pgpu_batch_download_strided(h, rows, stride) != PGPU_OK) {   // (not this kind of batch / pool)
    g_adopted_bytes.fetch_sub(charge, std::memory_order_relaxed);
    return false;
  }
  AdoptCookie* ck = new AdoptCookie{blk, charge};
  LimbArenaExt* arena = limb_arena_open(blk->p, adopt_release, ck);
  blk.reset();
  std::vector<BigNumber> v;
  v.reserve(count);
  limb_blocks_adopt(arena, rows, stride, count);
  for (std::size_t i = 0; i < count; ++i) v.push_back(BigNumber::adoptLimbs64(rows + i * stride, (std::size_t)words));
  limb_arena_close(arena);
  *out = std::move(v);
  return true;
}
}  // namespace

std::vector<BigNumber> DeviceBatch::download() const {
  const std::size_t bytes = count * (size_t)words * 8;
  // a batch whose pool is gone (terminateContext, also followed by a new context): values a caller handed in and that
  // went to the GPU at construction are still in the pinned block the upload read -- the caller gets its own input back
  if (src && !pgpu_batch_is_current(h)) {
    (void)pgpu_host_wait(src->p);
    return unpack(src->p, count, words);
  }
  if (bytes >= kAdoptMinBytes) {   // (below: the copying path is as fast -- Add_CTCT(2048) 141 against 175 us)
    std::vector<BigNumber> v;
    if (download_adopting(h, count, words, &v)) return v;
  }
  // the results land in a pinned block (one DMA, no unpacking copy) and become BigNumbers whose limbs share one arena
  if (std::shared_ptr<PinnedBlock> blk = bytes >= kEagerUploadBytes ? PinnedBlock::acquire(bytes) : nullptr) {
    IPCL_GPU_CHECK(pgpu_batch_download(h, blk->p), "device download");
    return unpack(blk->p, count, words);
  }
  std::vector<uint64_t> flat(count * (size_t)words);
  IPCL_GPU_CHECK(pgpu_batch_download(h, flat.data()), "device download");
  return unpack(flat, count, words);
}

}  // namespace detail
}  // namespace ipcl

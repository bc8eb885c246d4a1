// pailliercryptolib_amd -- device pool, worker lanes, allocator, staging copies, key replication (runtime.hpp).
#include "runtime.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>   // types only: the library is dlopen'ed (replicate over xGMI), never linked

#include <algorithm>
#include <functional>
#include <chrono>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <set>

namespace pgpu {
namespace rt {

thread_local std::string g_err;
int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

namespace {

std::mutex g_pool_mu;   // init / shutdown
// (never destroyed: a process may exit without pgpu_shutdown, and worker threads must not find their Device gone)
std::vector<std::unique_ptr<Device>>& g_pool = *new std::vector<std::unique_ptr<Device>>();
// Devices of earlier pools: batches, keys and buffers may outlive pgpu_shutdown (a caller's objects are destroyed
// after terminateContext); the (small) Device records they point to are parked here instead of being deleted.
std::vector<std::unique_ptr<Device>>& g_retired = *new std::vector<std::unique_ptr<Device>>();
bool g_init = false;
std::atomic<uint64_t> g_generation{0};
std::atomic<uint64_t> g_repl_verified{0}, g_repl_repaired{0};
std::atomic<int> g_corrupt_next{-1};
thread_local int t_current = 0;
size_t g_min_shard = 0;
const char* g_transport = "single";

// ---- RCCL, loaded at run time ----
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::vector<ncclComm_t> comms;
  bool ready = false;
  std::string note;
} g_rccl;

bool rccl_load() {
  if (g_rccl.lib) return true;
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    g_rccl.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (g_rccl.lib) break;
  }
  if (!g_rccl.lib) {
    g_rccl.note = "librccl not found";
    return false;
  }
#define PGPU_SYM(field, sym)                                                        \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(g_rccl.lib, sym)); \
  if (!g_rccl.field) {                                                              \
    g_rccl.note = std::string("librccl lacks ") + sym;                              \
    return false;                                                                   \
  }
  PGPU_SYM(CommInitAll, "ncclCommInitAll")
  PGPU_SYM(CommDestroy, "ncclCommDestroy")
  PGPU_SYM(Broadcast, "ncclBroadcast")
  PGPU_SYM(GroupStart, "ncclGroupStart")
  PGPU_SYM(GroupEnd, "ncclGroupEnd")
  PGPU_SYM(GetErrorString, "ncclGetErrorString")
#undef PGPU_SYM
  return true;
}

// one communicator per pool device (single process, ncclCommInitAll); needs distinct physical devices
void rccl_init(const std::vector<int>& ordinals) {
  g_rccl.ready = false;
  const char* no = std::getenv("PGPU_NO_RCCL");
  if (no && std::atoi(no) != 0) {
    g_rccl.note = "disabled by PGPU_NO_RCCL";
    return;
  }
  const char* force = std::getenv("PGPU_RCCL_FORCE");   // also bring it up for a pool of one (self-test)
  if (ordinals.size() < 2 && !(force && std::atoi(force) != 0)) return;
  if (std::set<int>(ordinals.begin(), ordinals.end()).size() != ordinals.size()) {
    g_rccl.note = "pool entries share a physical device";
    return;
  }
  if (!rccl_load()) return;
  g_rccl.comms.assign(ordinals.size(), nullptr);
  ncclResult_t r = g_rccl.CommInitAll(g_rccl.comms.data(), (int)ordinals.size(), ordinals.data());
  if (r != ncclSuccess) {
    g_rccl.note = std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r);
    g_rccl.comms.clear();
    return;
  }
  g_rccl.ready = true;
  g_rccl.note = "ok";
}

void rccl_shutdown() {
  if (g_rccl.ready)
    for (ncclComm_t c : g_rccl.comms)
      if (c) (void)g_rccl.CommDestroy(c);
  g_rccl.comms.clear();
  g_rccl.ready = false;
}

void lane_main(Device* d, Lane* lane) {
  (void)hipSetDevice(d->ordinal);
  for (;;) {
    std::function<void(Lane&)> fn;
    {
      std::unique_lock<std::mutex> lk(d->mu);
      d->cv.wait(lk, [&] { return d->stop || !d->queue.empty(); });
      if (d->queue.empty()) return;   // stop requested and nothing left
      fn = std::move(d->queue.front());
      d->queue.pop_front();
    }
    fn(*lane);
  }
}

}  // namespace

// ---------------- Workspace ----------------
int Workspace::ensure(size_t need, hipStream_t s) {
  if (need <= bytes) return PGPU_OK;
  // Growth through the device's block arena (Device::alloc / free, free lists keyed by stream): the old block goes back to
  // the list of ITS stream, where only later launches of that stream can pick it up -- behind the ones that still read it.
  // Rounds 3-6 grew with hipMallocAsync / hipFreeAsync on the owning stream; round 6 found four API threads that grew their
  // workspaces at the same moment (first decrypt of 1024 ciphertexts each, launches sharing CUs) delivering zeros for the tail
  // of a batch in 2 % of the runs -- device memory checked by a second copy, inputs intact, a repeated decrypt right -- and none
  // in 300 runs without the stream-ordered pool (profiles/r06_thread_race.txt).  Grow by half at least, so that a slowly
  // growing batch does not reallocate on every call.
  if (!dev) return fail(PGPU_ERR_INVALID_PARAM, "workspace without a device");
  const size_t want = std::max(need, bytes + bytes / 2);
  void* np = nullptr;
  RC_TRY(dev->alloc(want, s, &np));
  if (p) dev->free(p, s);
  p = np;
  bytes = want;
  return PGPU_OK;
}
void Workspace::release() {
  if (p) (void)hipFree(p);   // (synchronises the device; the caller has dropped the arena's record of the block)
  p = nullptr;
  bytes = 0;
}

// ---------------- Device ----------------
int Device::bind() const {
  HIP_TRY(hipSetDevice(ordinal));
  return PGPU_OK;
}

namespace {
constexpr size_t kGranule = 64 * 1024, kIdleCap = (size_t)4 << 30;
}

int Device::alloc(size_t bytes, hipStream_t tag, void** out) {
  if (!out) return fail(PGPU_ERR_INVALID_PARAM, "null output pointer");
  const size_t rounded = (std::max<size_t>(bytes, 1) + kGranule - 1) / kGranule * kGranule;
  {
    std::lock_guard<std::mutex> lk(mu);
    auto it = free_blocks.find({tag, rounded});
    if (it != free_blocks.end() && !it->second.empty()) {
      *out = it->second.back();
      it->second.pop_back();
      idle_bytes -= rounded;
      return PGPU_OK;
    }
  }
  DeviceGuard g(ordinal);
  hipError_t e = hipMalloc(out, rounded);
  if (e != hipSuccess) {   // out of memory: drop the idle blocks and retry once
    release_idle();
    HIP_TRY(hipMalloc(out, rounded));
  }
  std::lock_guard<std::mutex> lk(mu);
  block_size[*out] = rounded;
  return PGPU_OK;
}

void Device::free(void* p, hipStream_t tag) {
  if (!p) return;
  size_t sz = 0;
  {
    std::lock_guard<std::mutex> lk(mu);
    auto it = block_size.find(p);
    if (it != block_size.end()) {
      sz = it->second;
      if (alive && idle_bytes + sz <= kIdleCap) {
        free_blocks[{tag, sz}].push_back(p);
        idle_bytes += sz;
        return;
      }
      block_size.erase(it);
    }
  }
  DeviceGuard g(ordinal);
  (void)hipFree(p);   // (synchronises the device)
}

void Device::release_idle() {
  DeviceGuard g(ordinal);
  (void)hipDeviceSynchronize();
  std::lock_guard<std::mutex> lk(mu);
  for (auto& kv : free_blocks)
    for (void* p : kv.second) {
      (void)hipFree(p);
      block_size.erase(p);
    }
  free_blocks.clear();
  idle_bytes = 0;
}

StreamWork& Device::work_for(hipStream_t s) {
  std::lock_guard<std::mutex> lk(mu);
  auto it = work.find(s);
  if (it != work.end()) return *it->second;
  if (work.size() >= 32 && work.size() % 32 == 0) {
    // streams come and go (callers create and destroy them): give back the scratch memory of the idle ones.
    // The (small) entries themselves stay, so a reference handed out earlier never dangles; an entry whose
    // mutex is taken is in use right now and keeps its memory.
    DeviceGuard g(ordinal);
    (void)hipDeviceSynchronize();
    for (auto& w : work) {
      if (w.second->mu.try_lock()) {
        block_size.erase(w.second->table.p);
        block_size.erase(w.second->vbuf.p);
        w.second->table.release();
        w.second->vbuf.release();
        w.second->mu.unlock();
      }
    }
  }
  auto& slot = work[s];
  slot.reset(new StreamWork);
  slot->table.dev = this;
  slot->vbuf.dev = this;
  return *slot;
}

hipEvent_t Device::pool_event() {
  {
    std::lock_guard<std::mutex> lk(mu);
    if (!event_pool.empty()) {
      hipEvent_t e = event_pool.back();
      event_pool.pop_back();
      return e;
    }
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

void Device::post(std::function<void(Lane&)> fn) {
  {
    std::lock_guard<std::mutex> lk(mu);
    queue.push_back(std::move(fn));
  }
  cv.notify_one();
}

// ---------------- staging copies ----------------
void big_copy(void* dst, const void* src, size_t n) {
  // A single thread moves ~8 GB/s between pageable memory and a pinned buffer -- less than the DMA engine behind
  // it -- so large copies are split over a few persistent helper threads (never joined: they sleep on a
  // condition variable and die with the process).  One job at a time; a second caller copies on its own.
  struct Pool {
    enum { kHelpers = 3 };
    std::mutex job;   // held for the duration of one copy
    std::mutex m;
    std::condition_variable cv, done;
    char* dst = nullptr;
    const char* src = nullptr;
    size_t n = 0, per = 0;
    unsigned gen = 0;
    int pending = 0;
    Pool() {
      for (int i = 0; i < kHelpers; ++i) std::thread([this, i] { run(i + 1); }).detach();
    }
    void run(int slice) {
      unsigned seen = 0;
      for (;;) {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return gen != seen; });
        seen = gen;
        char* d = dst;
        const char* s = src;
        const size_t nn = n, pp = per;
        lk.unlock();
        const size_t lo = std::min(nn, pp * (size_t)slice), hi = std::min(nn, pp * (size_t)(slice + 1));
        if (hi > lo) std::memcpy(d + lo, s + lo, hi - lo);
        lk.lock();
        if (--pending == 0) done.notify_one();
      }
    }
  };
  static Pool* pool = new Pool();
  if (n < ((size_t)2 << 20) || !pool->job.try_lock()) {
    std::memcpy(dst, src, n);
    return;
  }
  const size_t per = (n / (Pool::kHelpers + 1) + 63) & ~(size_t)63;
  {
    std::lock_guard<std::mutex> lk(pool->m);
    pool->dst = (char*)dst;
    pool->src = (const char*)src;
    pool->n = n;
    pool->per = per;
    pool->pending = Pool::kHelpers;
    ++pool->gen;
  }
  pool->cv.notify_all();
  std::memcpy(dst, src, std::min(per, n));   // slice 0 on the calling thread
  {
    std::unique_lock<std::mutex> lk(pool->m);
    pool->done.wait(lk, [&] { return pool->pending == 0; });
  }
  pool->job.unlock();
}

// ---------------- pinned host blocks ----------------
namespace {
struct HostBlock {
  size_t bytes = 0;
  // one event per (pool device, stream) that ever uploaded from the block, recorded behind that stream's LATEST copy: a
  // caller that feeds several batch lanes from one block (the documented one-thread pipeline) has copies pending on
  // several streams at once, and an earlier copy queued behind another lane's kernels may not have read the block yet
  // when the last-recorded stream is done -- pgpu_host_wait / pgpu_host_free wait for every one of them
  struct Read {
    int dev;
    hipStream_t stream;
    hipEvent_t ev;
  };
  std::vector<Read> last_read;
};
std::mutex g_host_mu;
std::map<uintptr_t, HostBlock>& g_host_blocks = *new std::map<uintptr_t, HostBlock>();
// g_host_mu held: the block that contains [p, p+bytes), or end()
std::map<uintptr_t, HostBlock>::iterator host_lookup(const void* p, size_t bytes) {
  const uintptr_t a = (uintptr_t)p;
  auto it = g_host_blocks.upper_bound(a);
  if (it == g_host_blocks.begin()) return g_host_blocks.end();
  --it;
  if (a >= it->first && a + bytes <= it->first + it->second.bytes) return it;
  return g_host_blocks.end();
}
}  // namespace

int host_alloc(size_t bytes, void** out) {
  if (!out || bytes == 0) return fail(PGPU_ERR_INVALID_PARAM, "pgpu_host_alloc: null pointer or zero size");
  RC_TRY(check_ready());
  void* p = nullptr;
  HIP_TRY(hipHostMalloc(&p, bytes, hipHostMallocPortable));
  std::lock_guard<std::mutex> lk(g_host_mu);
  g_host_blocks[(uintptr_t)p].bytes = bytes;
  *out = p;
  return PGPU_OK;
}

int host_wait(const void* p) {
  std::vector<hipEvent_t> evs;
  {
    std::lock_guard<std::mutex> lk(g_host_mu);
    auto it = host_lookup(p, 1);
    if (it == g_host_blocks.end()) return PGPU_OK;   // not ours: nothing is ever pending on it
    for (const auto& r : it->second.last_read) evs.push_back(r.ev);
  }
  for (hipEvent_t e : evs)
    if (e) HIP_TRY(hipEventSynchronize(e));
  return PGPU_OK;
}

void host_free(void* p) {
  if (!p) return;
  (void)host_wait(p);
  HostBlock b;
  {
    std::lock_guard<std::mutex> lk(g_host_mu);
    auto it = g_host_blocks.find((uintptr_t)p);
    if (it == g_host_blocks.end()) return;
    b = it->second;
    g_host_blocks.erase(it);
  }
  for (const auto& r : b.last_read)
    if (r.ev) (void)hipEventDestroy(r.ev);
  (void)hipHostFree(p);
}

bool host_is_pinned(const void* p, size_t bytes) {
  std::lock_guard<std::mutex> lk(g_host_mu);
  return host_lookup(p, bytes) != g_host_blocks.end();
}

void host_note_read(const void* p, size_t bytes, int dev, hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_host_mu);
  auto it = host_lookup(p, bytes);
  if (it == g_host_blocks.end() || dev < 0) return;
  auto& reads = it->second.last_read;
  for (auto& r : reads)
    if (r.dev == dev && r.stream == s) {
      (void)hipEventRecord(r.ev, s);
      return;
    }
  hipEvent_t ev = nullptr;
  if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipStreamSynchronize(s);   // no event to remember the copy by: wait for it here
    return;
  }
  (void)hipEventRecord(ev, s);
  reads.push_back({dev, s, ev});
}

namespace {
size_t stage_piece(size_t bytes) {
  // Whole staging buffers.  (Round 4 tried quarter-transfer pieces so that packing piece i+1 overlaps the DMA of piece i
  // for the 1-4 MB operands of an 8192-element call: a lone caller gained nothing -- the per-piece event waits cost what
  // the overlap saved -- and with a second caller's kernels on the chip the small copies queued behind them: two callers
  // 9.1 instead of 6.0 ms per encrypt + decrypt.  PGPU_STAGE_PIECES=4 brings the pieces back for A/B runs.)
  static const int pieces = [] { const char* e = std::getenv("PGPU_STAGE_PIECES"); return e ? std::max(1, std::atoi(e)) : 1; }();
  if (pieces <= 1) return kStageBytes;
  const size_t part = ((bytes + pieces - 1) / pieces + 4095) & ~(size_t)4095;
  return std::min(kStageBytes, std::max<size_t>((size_t)512 << 10, part));
}
hipError_t wait_event(hipEvent_t ev) { return hipEventSynchronize(ev); }
int ensure_stage(Lane& l) {
  for (int i = 0; i < 2; ++i) {
    if (!l.stage[i]) HIP_TRY(hipHostMalloc(&l.stage[i], kStageBytes, hipHostMallocDefault));
    if (!l.stage_ev[i]) HIP_TRY(hipEventCreateWithFlags(&l.stage_ev[i], hipEventDisableTiming));
  }
  return PGPU_OK;
}
}  // namespace

// A device -> host copy is handed to the copy engine only once the kernels in front of it on its stream have run.
// The engines work through their rings in order: a copy queued behind an unfinished kernel parks the ring on that
// kernel's completion signal, and every copy queued after it -- of ANY stream -- waits there too.  Measured r04 with two
// synchronous callers (tools/probe_two_callers.py, rocprofv3 --memory-copy-trace --hip-runtime-trace): one caller's
// 24 us upload sat 0.9 ms behind the other's download (which was waiting for a 0.8 ms encrypt), 4.6 ms behind a decrypt;
// the callers' kernels ended up strictly one after the other with the copies exposed in between (6.1 ms per encrypt +
// decrypt instead of 5.4).  One host round trip (~10 us) per download.  PGPU_D2H_PRESYNC=0 turns it off.
// The wait costs a lone caller one more host round trip per download (Add_CTCT(16) at the ipcl:: API 57 -> 67 us) and buys
// it nothing -- its own copies are in order anyway.  So it is taken only while more than one host thread has been calling
// into the transfer entry points (note_caller: another thread within the last 50 ms), and by the asynchronous downloads
// (their point is that the SAME thread keeps other lanes busy meanwhile).  PGPU_D2H_PRESYNC: 0 never, 1 (default) as
// described, 2 always.
namespace {
std::atomic<uint64_t> g_last_caller{0};
std::atomic<int64_t> g_other_caller_ns{0};
thread_local bool t_force_presync = false;
int64_t now_ns() {
  return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
}  // namespace
void note_caller() {
  static thread_local const uint64_t me = std::hash<std::thread::id>()(std::this_thread::get_id()) | 1;
  const uint64_t prev = g_last_caller.exchange(me, std::memory_order_relaxed);
  if (prev != me && prev != 0) g_other_caller_ns.store(now_ns(), std::memory_order_relaxed);
}
void force_presync(bool on) { t_force_presync = on; }
hipError_t drain_before_copy(hipStream_t s) {
  static const int mode = [] { const char* e = std::getenv("PGPU_D2H_PRESYNC"); return e ? std::atoi(e) : 1; }();
  if (mode <= 0) return hipSuccess;
  if (mode == 1 && !t_force_presync) {
    const int64_t other = g_other_caller_ns.load(std::memory_order_relaxed);
    if (other == 0 || now_ns() - other > 50 * 1000000ll) return hipSuccess;
  }
  return hipStreamSynchronize(s);
}

// host -> device: chunk i+1 is packed into the other pinned buffer while chunk i is on the wire.  The copies
// are ordered on `s` (the stream the consumer kernels run on); nothing is waited for at the end.
int Lane::h2d(void* d_dst, const void* h_src, size_t bytes, hipStream_t s) {
  if (bytes && host_is_pinned(h_src, bytes)) {   // a caller buffer from pgpu_host_alloc: it IS the DMA source
    HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, s));
    host_note_read(h_src, bytes, dev ? dev->index : -1, s);
    return PGPU_OK;
  }
  RC_TRY(ensure_stage(*this));
  const size_t piece = stage_piece(bytes);
  size_t off = 0;
  for (int i = 0; off < bytes; ++i) {
    const int b = i & 1;
    const size_t n = std::min(piece, bytes - off);
    HIP_TRY(wait_event(stage_ev[b]));   // the DMA that last used this buffer is done (no-op if never recorded)
    big_copy(stage[b], (const char*)h_src + off, n);
    HIP_TRY(hipMemcpyAsync((char*)d_dst + off, stage[b], n, hipMemcpyHostToDevice, s));
    HIP_TRY(hipEventRecord(stage_ev[b], s));
    off += n;
  }
  return PGPU_OK;
}

// device -> host, ordered behind the work queued on `s`: chunk i+1 is on the wire while chunk i is unpacked
int Lane::d2h(void* h_dst, const void* d_src, size_t bytes, hipStream_t s) {
  HIP_TRY(drain_before_copy(s));
  if (bytes && host_is_pinned(h_dst, bytes)) {   // pinned target: one DMA, no unpacking copy
    HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, s));
    hipError_t e = hipStreamSynchronize(s);
    if (e != hipSuccess) return fail(PGPU_ERR_HIP, std::string("device -> host copy failed: ") + hipGetErrorString(e));
    return PGPU_OK;
  }
  RC_TRY(ensure_stage(*this));
  const size_t piece = stage_piece(bytes);
  const int chunks = (int)((bytes + piece - 1) / piece);
  auto issue = [&](int i) -> hipError_t {
    const size_t off = (size_t)i * piece, n = std::min(piece, bytes - off);
    hipError_t r = wait_event(stage_ev[i & 1]);
    if (r == hipSuccess) r = hipMemcpyAsync(stage[i & 1], (const char*)d_src + off, n, hipMemcpyDeviceToHost, s);
    return r == hipSuccess ? hipEventRecord(stage_ev[i & 1], s) : r;
  };
  hipError_t e = hipSuccess;
  if (chunks > 0) e = issue(0);
  for (int i = 0; i < chunks && e == hipSuccess; ++i) {
    if (i + 1 < chunks) e = issue(i + 1);   // its buffer was unpacked in iteration i-1
    if (e == hipSuccess) e = wait_event(stage_ev[i & 1]);
    if (e != hipSuccess) break;
    const size_t off = (size_t)i * piece, n = std::min(piece, bytes - off);
    big_copy((char*)h_dst + off, stage[i & 1], n);
  }
  if (e != hipSuccess) return fail(PGPU_ERR_HIP, std::string("device -> host copy failed: ") + hipGetErrorString(e));
  return PGPU_OK;
}

// ---------------- tasks ----------------
void TaskGroup::run(Device& d, std::function<int(Lane&)> fn) {
  {
    std::lock_guard<std::mutex> lk(mu);
    ++pending;
  }
  d.post([this, fn](Lane& lane) {
    int rc = fn(lane);
    std::lock_guard<std::mutex> lk(mu);
    if (rc != 0 && status == 0) {
      status = rc;
      err = g_err;   // the lane thread's message
    }
    if (--pending == 0) cv.notify_all();
  });
}

int TaskGroup::wait() {
  std::unique_lock<std::mutex> lk(mu);
  cv.wait(lk, [&] { return pending == 0; });
  if (status != 0) g_err = err;
  return status;
}

// ---------------- pool ----------------
bool initialized() { return g_init; }

int check_ready() {
  if (!g_init) return fail(PGPU_ERR_NO_DEVICE, "pgpu_init has not been called (no GPU context)");
  return PGPU_OK;
}

int pool_size() { return (int)g_pool.size(); }
uint64_t pool_generation() { return g_generation.load(std::memory_order_acquire); }
void replication_stats(uint64_t* verified, uint64_t* repaired) {
  if (verified) *verified = g_repl_verified.load();
  if (repaired) *repaired = g_repl_repaired.load();
}
void debug_corrupt_next_replica(int index) { g_corrupt_next.store(index); }
Device& device(int i) { return *g_pool[(size_t)i]; }
Device& current() {
  int i = t_current;
  if (i < 0 || i >= (int)g_pool.size()) i = 0;
  return *g_pool[(size_t)i];
}
int current_index() { return (t_current >= 0 && t_current < (int)g_pool.size()) ? t_current : 0; }
int set_current(int index) {
  RC_TRY(check_ready());
  if (index < 0 || index >= (int)g_pool.size()) return fail(PGPU_ERR_INVALID_PARAM, "pool index out of range");
  t_current = index;
  return PGPU_OK;
}
const char* replicate_transport() { return g_transport; }

size_t min_shard() {
  if (g_min_shard == 0) {
    const char* e = std::getenv("PGPU_MIN_SHARD");
    g_min_shard = e && std::atol(e) > 0 ? (size_t)std::atol(e) : 256;
  }
  return g_min_shard;
}
void set_min_shard(size_t n) { g_min_shard = n ? n : 256; }
int shard_devices(size_t count) {
  const size_t D = (size_t)std::max(1, pool_size());
  return (int)std::max<size_t>(1, std::min(D, count / min_shard()));
}

// A process that exits without pgpu_shutdown (scripts, a crashed test) must not hang in its static destructors:
// stop and join the worker lanes, and leave every GPU resource to the driver (the HIP runtime may already be
// shutting down at this point, so no HIP call is made here).
static void stop_lanes_at_exit() {
  for (auto& d : g_pool) {
    {
      std::lock_guard<std::mutex> l2(d->mu);
      d->stop = true;
    }
    d->cv.notify_all();
    for (auto& lane : d->lanes)
      if (lane->th.joinable()) lane->th.join();
  }
}

int pool_init(const std::vector<int>& ordinals) {
  std::lock_guard<std::mutex> lk(g_pool_mu);
  static const bool hooked = [] { return std::atexit(stop_lanes_at_exit) == 0; }();
  (void)hooked;
  if (g_init) {
    bool same = ordinals.size() == g_pool.size();
    for (size_t i = 0; same && i < ordinals.size(); ++i) same = g_pool[i]->ordinal == ordinals[i];
    if (same) return PGPU_OK;
    return fail(PGPU_ERR_INVALID_PARAM, "already initialised with a different device set (call pgpu_shutdown first)");
  }
  int visible = 0;
  if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0)
    return fail(PGPU_ERR_NO_DEVICE, "no HIP device visible");
  if (ordinals.empty()) return fail(PGPU_ERR_INVALID_PARAM, "empty device list");
  std::vector<std::unique_ptr<Device>> pool;
  for (size_t i = 0; i < ordinals.size(); ++i) {
    if (ordinals[i] < 0 || ordinals[i] >= visible)
      return fail(PGPU_ERR_INVALID_PARAM, "device ordinal out of range");
    std::unique_ptr<Device> d(new Device);
    d->index = (int)i;
    d->ordinal = ordinals[i];
    HIP_TRY(hipSetDevice(d->ordinal));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, d->ordinal));
    d->name = std::string(prop.name) + " (" + prop.gcnArchName + ")";
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
      return fail(PGPU_ERR_NO_DEVICE, "device is not gfx950: " + d->name);
    for (int k = 0; k < kBatchLanes; ++k) {
      HIP_TRY(hipStreamCreateWithFlags(&d->bstreams[k], hipStreamNonBlocking));
      HIP_TRY(hipEventCreateWithFlags(&d->xlane_ev[k], hipEventDisableTiming));
    }
    // (worker lanes: host-array calls and asynchronous downloads run on them, each blocks its lane until its kernels are
    //  done -- four, so that two pipelined downloads leave room for two synchronous callers; pinned staging buffers are
    //  only allocated by a lane that ever moves pageable memory)
    for (int l = 0; l < 4; ++l) {
      std::unique_ptr<Lane> lane(new Lane);
      lane->dev = d.get();
      lane->id = l;
      HIP_TRY(hipStreamCreateWithFlags(&lane->stream, hipStreamNonBlocking));
      d->lanes.push_back(std::move(lane));
    }
    pool.push_back(std::move(d));
  }
  HIP_TRY(hipSetDevice(ordinals[0]));
  g_pool = std::move(pool);
  for (auto& d : g_pool)
    for (auto& lane : d->lanes) lane->th = std::thread(lane_main, d.get(), lane.get());
  rccl_init(ordinals);
  g_transport = g_pool.size() == 1 && !g_rccl.ready ? "single" : (g_rccl.ready ? "rccl" : "memcpy");
  t_current = 0;
  g_generation.fetch_add(1, std::memory_order_acq_rel);
  g_repl_verified.store(0);
  g_repl_repaired.store(0);
  g_init = true;
  return PGPU_OK;
}

void pool_shutdown() {
  std::lock_guard<std::mutex> lk(g_pool_mu);
  if (!g_init) return;
  g_init = false;
  for (auto& d : g_pool) {
    {
      std::lock_guard<std::mutex> l2(d->mu);
      d->stop = true;
    }
    d->cv.notify_all();
    for (auto& lane : d->lanes)
      if (lane->th.joinable()) lane->th.join();
  }
  rccl_shutdown();
  for (auto& d : g_pool) {
    (void)hipSetDevice(d->ordinal);
    (void)hipDeviceSynchronize();
    for (auto& kv : d->work) {
      d->block_size.erase(kv.second->table.p);
      d->block_size.erase(kv.second->vbuf.p);
      kv.second->table.release();
      kv.second->vbuf.release();
    }
    d->work.clear();
    for (auto& kv : d->free_blocks)
      for (void* p : kv.second) {
        (void)hipFree(p);
        d->block_size.erase(p);
      }
    d->free_blocks.clear();
    d->idle_bytes = 0;
    for (auto& t : d->timed) {
      d->event_pool.push_back(t.e0);
      d->event_pool.push_back(t.e1);
    }
    d->timed.clear();
    for (hipEvent_t e : d->event_pool) (void)hipEventDestroy(e);
    d->event_pool.clear();
    for (auto& lane : d->lanes) {
      for (int i = 0; i < 2; ++i) {
        if (lane->stage[i]) (void)hipHostFree(lane->stage[i]);
        if (lane->stage_ev[i]) (void)hipEventDestroy(lane->stage_ev[i]);
      }
      if (lane->stream) (void)hipStreamDestroy(lane->stream);
    }
    for (int k = 0; k < kBatchLanes; ++k) {
      if (d->bstreams[k]) (void)hipStreamDestroy(d->bstreams[k]);
      if (d->xlane_ev[k]) (void)hipEventDestroy(d->xlane_ev[k]);
    }
  }
  {
    // pinned host blocks belong to their callers and survive; the events that remember uploads from them do not (the
    // devices have drained above, and the next pool may number its entries differently)
    std::lock_guard<std::mutex> hl(g_host_mu);
    for (auto& kv : g_host_blocks) {
      for (const auto& r : kv.second.last_read)
        if (r.ev) (void)hipEventDestroy(r.ev);
      kv.second.last_read.clear();
    }
  }
  if (!g_pool.empty()) (void)hipSetDevice(g_pool[0]->ordinal);
  for (auto& d : g_pool) {
    d->alive = false;
    for (int k = 0; k < kBatchLanes; ++k) {
      d->bstreams[k] = nullptr;
      d->xlane_ev[k] = nullptr;
    }
    d->lanes.clear();
    g_retired.push_back(std::move(d));
  }
  g_pool.clear();
  g_transport = "single";
}

// ---------------- replicated constant data ----------------
Replicated::~Replicated() { scrub_and_free(); }

void Replicated::scrub_and_free() {
  for (size_t i = 0; i < d.size(); ++i) {
    if (!d[i]) continue;
    if (gen == pool_generation() && i < g_pool.size()) {
      DeviceGuard g(g_pool[i]->ordinal);
      // Every stream of the library is hipStreamNonBlocking: a null-stream memset does NOT wait for kernels still queued
      // on them, and a `_dev` call or a resident-batch operation only enqueues its kernel.  A secret image is therefore
      // zeroed only after the device has drained (round-3 advisor: an LRU eviction or a key destroyed with operations in
      // flight could otherwise hand zeroed constants to a queued kernel).  hipFree below synchronises anyway; the wait
      // merely moves in front of the memset.
      if (secret_) {
        (void)hipDeviceSynchronize();
        (void)hipMemset(d[i], 0, bytes);   // key material does not outlive the key object
      }
      (void)hipFree(d[i]);
    } else {
      // the pool this copy was made for is gone (its devices were reset or re-enumerated): the address identifies
      // its device to the driver; a failure here means the memory went with the old context
      if (secret_) (void)hipMemset(d[i], 0, bytes);
      (void)hipFree(d[i]);
      (void)hipGetLastError();
    }
  }
  d.clear();
  bytes = 0;
}

namespace {
uint64_t fnv1a64(const unsigned char* p, size_t n) {
  uint64_t h = 1469598103934665603ull;
  for (size_t i = 0; i < n; ++i) {
    h ^= p[i];
    h *= 1099511628211ull;
  }
  return h;
}
}  // namespace

int Replicated::upload(const void* host, size_t nbytes, bool secret) {
  static std::mutex collective_mu;   // one collective at a time on the pool's communicators
  std::lock_guard<std::mutex> clk(collective_mu);
  scrub_and_free();
  secret_ = secret;
  bytes = nbytes;
  gen = pool_generation();
  const int D = pool_size();
  d.assign((size_t)D, nullptr);
  for (int i = 0; i < D; ++i) {
    DeviceGuard g(device(i).ordinal);
    HIP_TRY(hipMalloc(&d[(size_t)i], nbytes));
  }
  {
    DeviceGuard g(device(0).ordinal);
    HIP_TRY(hipMemcpy(d[0], host, nbytes, hipMemcpyHostToDevice));
  }
  if (D == 1 && !g_rccl.ready) return PGPU_OK;
  bool by_rccl = false;
  if (g_rccl.ready) {
    // ONE broadcast from device 0 over xGMI (single process: a group call over all communicators)
    ncclResult_t r = g_rccl.GroupStart();
    for (int i = 0; i < D && r == ncclSuccess; ++i) {
      DeviceGuard g(device(i).ordinal);
      r = g_rccl.Broadcast(d[0], d[(size_t)i], nbytes, ncclUint8, 0, g_rccl.comms[(size_t)i], device(i).bs(0));
    }
    ncclResult_t r2 = g_rccl.GroupEnd();
    if (r == ncclSuccess) r = r2;
    if (r == ncclSuccess) {
      for (int i = 0; i < D; ++i) {
        DeviceGuard g(device(i).ordinal);
        HIP_TRY(hipStreamSynchronize(device(i).bs(0)));
      }
      by_rccl = true;
    } else {
      // a failed collective leaves RCCL unusable: remember it and copy per device from here on
      g_rccl.note = std::string("ncclBroadcast: ") + g_rccl.GetErrorString(r);
      g_rccl.ready = false;
      g_transport = "memcpy";
    }
  }
  if (!by_rccl) {
    for (int i = 1; i < D; ++i) {
      DeviceGuard g(device(i).ordinal);
      HIP_TRY(hipMemcpy(d[(size_t)i], host, nbytes, hipMemcpyHostToDevice));
    }
  }
  const int hit = g_corrupt_next.exchange(-1);   // test hook: what a wrong-but-successful collective leaves behind
  if (hit >= 0 && hit < D && nbytes > 0) {
    DeviceGuard g(device(hit).ordinal);
    const size_t off = nbytes / 2;
    unsigned char x = (unsigned char)(((const unsigned char*)host)[off] ^ 0x5a);
    HIP_TRY(hipMemcpy((char*)d[(size_t)hit] + off, &x, 1, hipMemcpyHostToDevice));
  }
  // ---- self-check: every copy is read back once and compared with the host image ----
  // (key images are a few KB to a few hundred KB and are uploaded once per key: the read-back is noise next to the
  // hipMallocs above.  The first run on real multi-GPU hardware diagnoses itself: a copy that differs is rewritten
  // from the host, counted, and RCCL is retired for the rest of the process if it produced it.)
  const uint64_t want = fnv1a64((const unsigned char*)host, nbytes);
  std::vector<unsigned char> back(nbytes);
  for (int i = 0; i < D; ++i) {
    DeviceGuard g(device(i).ordinal);
    HIP_TRY(hipMemcpy(back.data(), d[(size_t)i], nbytes, hipMemcpyDeviceToHost));
    const bool ok = fnv1a64(back.data(), nbytes) == want;
    g_repl_verified.fetch_add(1);
    if (!ok) {
      g_repl_repaired.fetch_add(1);
      HIP_TRY(hipMemcpy(d[(size_t)i], host, nbytes, hipMemcpyHostToDevice));
      HIP_TRY(hipMemcpy(back.data(), d[(size_t)i], nbytes, hipMemcpyDeviceToHost));
      if (fnv1a64(back.data(), nbytes) != want)
        return fail(PGPU_ERR_HIP, "replicated key image does not read back correctly on pool entry " + std::to_string(i));
      if (by_rccl) {
        g_rccl.note = "ncclBroadcast delivered a wrong image to pool entry " + std::to_string(i) + ": retired, copying per device";
        g_rccl.ready = false;
        g_transport = "memcpy";
      }
    }
  }
  if (secret) {
    volatile unsigned char* w = back.data();
    for (size_t i = 0; i < nbytes; ++i) w[i] = 0;
  }
  return PGPU_OK;
}

std::string rccl_note() { return g_rccl.note; }

}  // namespace rt
}  // namespace pgpu

// pailliercryptolib_amd -- host runtime behind the C-ABI (include/pgpu.h): the device pool.
//
// One process may drive several GPUs (pgpu_init_all): every pool entry ("Device") owns
//   * two worker lanes (host thread + HIP stream + pinned staging buffers): the host-pointer entry points cut a
//     batch into contiguous shards, one per device, and run each shard (or its sub-batches) as a task on a
//     lane, so that the H2D copy, the kernels and the D2H copy of different tasks overlap -- the in-process
//     fan-out the reference has in ipcl/mod_exp.cpp:700-731 (second std::thread) and in the QAT library's
//     round-robin over device instances (module/heqat/heqat/ctrl.c:500-529);
//   * a batch stream on which device-resident sharded batches (pgpu_batch) are processed;
//   * a caching allocator whose free lists are per stream (a block is reused only by work that is ordered
//     behind its last use);
//   * per-stream launch workspaces (window tables, CRT stage-1 buffer), so that `_dev` calls on different
//     streams never share scratch memory.
// Key material is built once on the host as a position-independent image and replicated to every device:
// uploaded to device 0 and broadcast with ONE RCCL ncclBroadcast over xGMI (librccl is dlopen'ed; when it is
// unavailable, or the pool maps several entries onto one physical GPU, the image is copied per device).
#ifndef PAILLIERCRYPTOLIB_AMD_CSRC_RUNTIME_HPP_
#define PAILLIERCRYPTOLIB_AMD_CSRC_RUNTIME_HPP_

#include <hip/hip_runtime_api.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "pgpu.h"

namespace pgpu {
namespace rt {

extern thread_local std::string g_err;
int fail(int code, const std::string& msg);

#define HIP_TRY(expr)                                                                            \
  do {                                                                                           \
    hipError_t e_ = (expr);                                                                      \
    if (e_ != hipSuccess)                                                                        \
      return ::pgpu::rt::fail(PGPU_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));  \
  } while (0)
#define RC_TRY(expr)     \
  do {                   \
    int rc_ = (expr);    \
    if (rc_) return rc_; \
  } while (0)

struct Device;

// makes `ordinal` the calling thread's HIP device for the scope (kernels and allocations address the
// current device) and restores the previous one: a caller's own device selection is left untouched
struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(int ordinal) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != ordinal) (void)hipSetDevice(ordinal);
    else prev = -1;
  }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

// grow-only device buffer owned by one (device, stream) pair.  Growth goes through the device's block arena: the old
// block returns to the free list of ITS stream (only later launches of that stream can reuse it, behind the ones that
// still read it), so a busy pool never sees a device-wide synchronisation because one stream's batch grew.
struct Workspace {
  void* p = nullptr;
  size_t bytes = 0;
  Device* dev = nullptr;   // whose block arena the memory comes from (set by Device::work_for)
  int ensure(size_t need, hipStream_t s);
  void release();
};

// launch scratch of one stream: `mu` is held from sizing the workspace until the launch is queued
struct StreamWork {
  std::mutex mu;
  Workspace table;   // per-instance window tables of modexp_kernel
  Workspace vbuf;    // CRT stage 1 -> stage 2 hand-over
};

struct TimedLaunch {
  int kind;
  int form;   // pgpu_kernel_form bits (include/pgpu.h): what the launcher picked for this launch
  hipStream_t stream;
  hipEvent_t e0, e1;
};

constexpr size_t kStageBytes = (size_t)8 << 20;
constexpr int kBatchLanes = 4;

struct Lane {
  Device* dev = nullptr;
  int id = 0;
  hipStream_t stream = nullptr;
  void* stage[2] = {nullptr, nullptr};
  hipEvent_t stage_ev[2] = {nullptr, nullptr};
  std::thread th;
  // host <-> device copies through this lane's pinned staging buffers, ordered on stream s
  int h2d(void* d_dst, const void* h_src, size_t bytes, hipStream_t s);   // queued: nothing is waited for
  int d2h(void* h_dst, const void* d_src, size_t bytes, hipStream_t s);   // returns after the data has arrived
};

struct Device {
  int index = 0;     // position in the pool
  int ordinal = 0;   // HIP device ordinal (several pool entries may share one when oversubscribed)
  std::string name;
  bool alive = true;               // false once its pool has been shut down (objects may still point here)
  // resident-batch operations run on the batch lanes: kBatchLanes independent chains of resident batches may be in
  // flight on a GPU (lane 0 also carries the key replication).  Round 3 had two; the sequential-halves kernels need
  // 16384 exponentiation pairs in flight to put a wavefront on every SIMD and 32768 for two, so an 8192-element step
  // wants up to four of them side by side.
  hipStream_t bstreams[kBatchLanes] = {};
  hipStream_t bs(int lane) const { return bstreams[lane % kBatchLanes]; }
  // cross-lane ordering (a batch of one lane consumed by an operation on another): one event per source lane
  hipEvent_t xlane_ev[kBatchLanes] = {};
  // host clock (ns, steady) of the last operation queued on each batch lane: a lane that was fed a moment ago counts as
  // active for the adaptive kernel-form policy even if the GPU has just drained it (capi.cpp: busy_other_lanes)
  std::atomic<int64_t> lane_fed_ns[kBatchLanes] = {};
  // ... and when a synchronous caller of the host-array entry points last ran a call on its thread's lane: seen by other
  // such callers only (capi.cpp: host_busy), not by pipelines of resident batches
  std::atomic<int64_t> host_fed_ns[kBatchLanes] = {};
  std::mutex mu;                   // allocator, work map, timing, queue

  // ---- caching allocator (sizes rounded to 64 KiB; free lists per stream tag) ----
  std::map<std::pair<hipStream_t, size_t>, std::vector<void*>> free_blocks;
  std::map<void*, size_t> block_size;
  size_t idle_bytes = 0;
  int alloc(size_t bytes, hipStream_t tag, void** out);
  void free(void* p, hipStream_t tag);
  void release_idle();

  // ---- per-stream launch scratch ----
  std::map<hipStream_t, std::unique_ptr<StreamWork>> work;
  StreamWork& work_for(hipStream_t s);

  // ---- live kernel timing (bench.py roofline) ----
  std::vector<TimedLaunch> timed;
  std::vector<hipEvent_t> event_pool;
  hipEvent_t pool_event();

  // ---- worker lanes ----
  std::vector<std::unique_ptr<Lane>> lanes;
  std::deque<std::function<void(Lane&)>> queue;
  std::condition_variable cv;
  bool stop = false;
  void post(std::function<void(Lane&)> fn);

  int bind() const;   // hipSetDevice(ordinal) for the calling thread
};

// waits for the work queued on `s` before a device -> host copy is issued on it (runtime.cpp: why)
hipError_t drain_before_copy(hipStream_t s);
void note_caller();             // called by the transfer entry points on the CALLING thread (multi-caller detection)
void force_presync(bool on);    // the calling (worker) thread's downloads always wait first (asynchronous downloads)

// RAII device allocation from a Device's allocator
struct DevMem {
  Device* dev = nullptr;
  hipStream_t tag = nullptr;
  void* p = nullptr;
  DevMem() = default;
  DevMem(const DevMem&) = delete;
  DevMem& operator=(const DevMem&) = delete;
  DevMem(DevMem&& o) noexcept : dev(o.dev), tag(o.tag), p(o.p) { o.p = nullptr; }
  ~DevMem() { reset(); }
  int alloc(Device& d, hipStream_t t, size_t bytes) {
    reset();
    dev = &d;
    tag = t;
    return d.alloc(bytes, t, &p);
  }
  void reset() {
    if (p && dev) dev->free(p, tag);
    p = nullptr;
  }
};

// ---- pool ----
bool initialized();
int check_ready();
int pool_init(const std::vector<int>& ordinals);   // one pool entry per listed HIP ordinal
void pool_shutdown();
int pool_size();
// bumped by every successful pool_init: device-side objects (key images, batches) remember the generation they
// were created in and are refused once the pool they belong to has been shut down
uint64_t pool_generation();
Device& device(int i);
Device& current();              // pool entry the calling thread addresses with the `_dev` entry points
int set_current(int index);
int current_index();
const char* replicate_transport();   // "rccl" | "memcpy" | "single"
std::string rccl_note();             // why RCCL is (not) in use
// replication self-check (Replicated::upload): images verified / copies found wrong and rewritten since pool_init
void replication_stats(uint64_t* verified, uint64_t* repaired);
// test hook: corrupt the copy on pool entry `index` of the NEXT replicated upload right after the broadcast
// (before verification), as a wrong-but-successful collective would
void debug_corrupt_next_replica(int index);

// ---- tasks ----
// runs fn(lane) for every item on a lane of items[i].first; returns the first non-zero status
struct TaskGroup {
  std::mutex mu;
  std::condition_variable cv;
  int pending = 0;
  int status = 0;
  std::string err;
  void run(Device& d, std::function<int(Lane&)> fn);
  int wait();
};

// ---- replicated constant data (key images) ----
struct Replicated {
  std::vector<void*> d;   // one copy per pool device
  size_t bytes = 0;
  uint64_t gen = 0;       // pool generation of the upload
  // usable by the pool as it is now (same generation, one copy per device)?
  bool current() const { return !d.empty() && gen == pool_generation() && (int)d.size() == pool_size(); }
  const void* at(int index) const { return (index >= 0 && (size_t)index < d.size() && gen == pool_generation()) ? d[(size_t)index] : nullptr; }
  // What the copies looked like after replication: the image is read back from EVERY device once per upload and
  // compared with the host bytes (64-bit FNV-1a); a mismatching copy is rewritten by a plain host-to-device copy
  // and counted (replication_repairs()).  A collective that "succeeds" with wrong bytes on some rank would
  // otherwise decrypt garbage on that GPU only.
  Replicated() = default;
  Replicated(const Replicated&) = delete;
  Replicated& operator=(const Replicated&) = delete;
  ~Replicated();
  int upload(const void* host, size_t nbytes, bool secret);   // allocate everywhere + broadcast
  void scrub_and_free();
  bool secret_ = false;
};

// contiguous shard of [0, count) for pool entry d of D (balanced: the first count % D shards get one more)
inline void shard_bounds(size_t count, int D, int d, size_t* lo, size_t* hi) {
  const size_t per = count / (size_t)D, rem = count % (size_t)D;
  *lo = per * (size_t)d + std::min<size_t>((size_t)d, rem);
  *hi = *lo + per + ((size_t)d < rem ? 1 : 0);
}
// number of devices a batch of `count` elements is spread over (tiny batches stay on few devices)
int shard_devices(size_t count);
size_t min_shard();
void set_min_shard(size_t n);

// multi-threaded memcpy between pageable and pinned memory (one job at a time; falls back to memcpy when busy)
void big_copy(void* dst, const void* src, size_t n);

// ---- pinned host blocks handed to callers (pgpu_host_alloc) ----
// A caller buffer that lies inside one of these blocks is the source / target of the DMA itself: Lane::h2d / d2h skip
// their staging copies (8-10 GB/s per host thread against ~50 GB/s on the link).  Uploads from a block are not waited
// for: the block remembers the last copy queued from it per pool device (an event) and host_wait / host_free wait for
// those -- the owner of the block decides when it has to be reusable.
int host_alloc(size_t bytes, void** out);
void host_free(void* p);
bool host_is_pinned(const void* p, size_t bytes);
int host_wait(const void* p);
// called by Lane::h2d after queueing a copy that READS [p, p+bytes) on stream s of pool entry `dev`
void host_note_read(const void* p, size_t bytes, int dev, hipStream_t s);

}  // namespace rt
}  // namespace pgpu

#endif  // PAILLIERCRYPTOLIB_AMD_CSRC_RUNTIME_HPP_

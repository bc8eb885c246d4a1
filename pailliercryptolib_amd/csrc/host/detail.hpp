// Internal helpers of the ipcl:: host layer: BigNumber <-> flat limb batches, device key holders.
#ifndef PAILLIERCRYPTOLIB_AMD_CSRC_HOST_DETAIL_HPP_
#define PAILLIERCRYPTOLIB_AMD_CSRC_HOST_DETAIL_HPP_

#include <cstddef>
#include <cstdint>
#include <functional>
#include <memory>
#include <vector>

#include "ipcl/bignum.h"
#include "pgpu.h"

namespace ipcl {
namespace detail {

// makes sure the process owns a GPU context (initializeContext may not have been called:
// the reference's CPU path needs no initialisation either)
void ensure_context();

inline int words_for_bits(int bits) { return bits <= 0 ? 1 : (bits + 63) / 64; }

// ---- host threads of the per-element loops (reference ipcl/include/ipcl/utils/util.hpp:77-113:
// OMPUtilities::MaxThreads / assignOMPThreads; loops at mod_exp.cpp:607-612, pri_key.cpp:122-145,
// ciphertext.cpp:53-68).  The reference opens an OpenMP region per loop; here a small persistent team of worker
// threads that SLEEP between loops (condition variable) does the same job: an OpenMP team spins at its barriers,
// which under a CPU quota (16 cores of a 256-thread host on this pool's GPU boxes, fewer in CI sandboxes) turns a
// 0.5 ms loop into tens of milliseconds -- measured, round 2 and again round 3 with a bounded team (pack of 8192
// values in a CI sandbox: 0.57 ms serial, 36 ms on 8 spinning threads).  The team is OPT-IN (IPCL_NUM_THREADS /
// OMP_NUM_THREADS > 1): on the GPU hosts of this pool the loops it would split take 60-260 us on one core and do not
// get faster on more (common.cpp: compute_thread_budget); the budget is capped by what the process may actually run
// on -- the CPU affinity mask cut by the cgroup quota -- and at 16.  A loop that finds the team busy (another application thread is inside a
// loop: the reference's application-level OpenMP pattern) runs on the calling thread alone -- the "remaining
// threads" rule of assignOMPThreads.
int max_host_threads();
int threads_for(std::size_t n, std::size_t grain);
// body(lo, hi) over disjoint chunks covering [0, n); body must not throw
void parallel_chunks(std::size_t n, std::size_t grain, const std::function<void(std::size_t, std::size_t)>& body);

// f(i) for i in [0, n)
template <class F>
inline void parallel_for(std::size_t n, std::size_t grain, F&& f) {
  if (threads_for(n, grain) <= 1) {
    for (std::size_t i = 0; i < n; ++i) f(i);
    return;
  }
  parallel_chunks(n, grain, [&f](std::size_t lo, std::size_t hi) {
    for (std::size_t i = lo; i < hi; ++i) f(i);
  });
}

// element-wise copy of a vector of BigNumbers (one heap block per value) on the host thread budget
std::vector<BigNumber> copy_texts(const std::vector<BigNumber>& v);
// true when every value is non-negative and fits `words` 64-bit limbs
bool all_fit(const std::vector<BigNumber>& v, int words);
// securely forget a host buffer that held key or obfuscator material (not elidable, unlike std::fill)
void wipe(void* p, std::size_t bytes);

// row-major [count][words] little-endian limbs of |v[i]|; every value must fit
std::vector<uint64_t> pack(const std::vector<BigNumber>& v, int words);
void pack_into(uint64_t* flat, const std::vector<BigNumber>& v, int words);
std::vector<BigNumber> unpack(const std::vector<uint64_t>& flat, std::size_t count, int words);
std::vector<BigNumber> unpack(const uint64_t* flat, std::size_t count, int words);   // (limb blocks from one arena)

// Pinned staging block of the host layer (pgpu_host_alloc: the DMA reads / writes it directly).  Blocks are pooled by
// size class; a block that still feeds an upload is waited for (pgpu_host_wait) before it is handed out again.
struct PinnedBlock {
  uint64_t* p = nullptr;
  std::size_t bytes = 0;
  ~PinnedBlock();
  static std::shared_ptr<PinnedBlock> acquire(std::size_t bytes);   // null: no pinned memory (callers use the heap)
};
int max_bits(const std::vector<BigNumber>& v);
void fill_random(void* dst, std::size_t n);   // kernel CSPRNG, bulk

// A batch resident in GPU memory: [count][words] little-endian 64-bit limbs, cut into contiguous shards over
// the device pool (pgpu_batch).  Immutable once produced (results are always written to fresh batches), so
// copies of a text can share it.  Ciphertext batches produced on the device are in the Montgomery domain of
// n^2; download() returns plain values either way.
struct DeviceBatch {
  pgpu_batch* h = nullptr;
  std::size_t count = 0;
  int words = 0;
  std::shared_ptr<PinnedBlock> src;   // the pinned block the upload reads (kept until the batch dies; the pool waits
                                      // for the copy before it recycles the block)
  ~DeviceBatch() {
    if (h) pgpu_batch_destroy(h);
  }
  static std::shared_ptr<DeviceBatch> adopt(pgpu_batch* h);
  static std::shared_ptr<DeviceBatch> upload(const std::vector<uint64_t>& flat, std::size_t count, int words);
  // packs v (every value non-negative and no wider than `words`) into a pinned block and queues ONE copy to the GPU;
  // returns without waiting for it
  static std::shared_ptr<DeviceBatch> upload_values(const std::vector<BigNumber>& v, int words);
  std::vector<BigNumber> download() const;
};
void release_pinned_pool();
// batches of at least this many bytes are uploaded as soon as a text is constructed around host values (base_text.cpp)
constexpr std::size_t kEagerUploadBytes = 64 * 1024;

struct PubKeyDevice {
  pgpu_pubkey* h = nullptr;
  BigNumber n, hs;
  bool djn = false;
  ~PubKeyDevice() {
    if (h) pgpu_pubkey_destroy(h);
  }
};

struct PrivKeyDevice {
  pgpu_privkey* h = nullptr;
  ~PrivKeyDevice() {
    if (h) pgpu_privkey_destroy(h);
  }
};

}  // namespace detail
}  // namespace ipcl
#endif

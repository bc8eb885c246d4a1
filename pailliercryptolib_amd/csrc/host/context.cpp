// ipcl::initializeContext / terminateContext over pgpu_init / pgpu_shutdown
// (reference ipcl/utils/context.cpp:40-88; there "QAT" acquires the accelerator, here every
// runtime choice does, because the GPU is the only compute path).
#include "ipcl/utils/context.hpp"

#include <malloc.h>

#include <cstdlib>
#include <map>
#include <mutex>
#include <stdexcept>

#include "detail.hpp"
#include "ipcl/utils/util.hpp"

namespace ipcl {

namespace detail {

void check_gpu(int status, const char* what, const char* file, int line) {
  if (status != PGPU_OK)
    throw std::runtime_error(build_log(file, line, std::string(what) + ": GPU engine error " +
                                                       std::to_string(status) + " (" +
                                                       pgpu_last_error() + ")"));
}

// Which GPUs the process drives (the reference's QAT runtime acquires every instance, heqat/context.h:18-26):
//   IPCL_GPU_DEVICE=<ordinal>   or a torchrun-style LOCAL_RANK  -> that one GPU (one process per GPU)
//   IPCL_GPU_DEVICES=<n>|all    -> an in-process pool over the first n / all visible GPUs (default: all)
// The ipcl:: API hands batches over as std::vector<BigNumber>: 8192 ciphertexts are 8192 heap blocks of 512 bytes plus
// a quarter-megabyte vector, created and destroyed once per call.  With glibc's defaults every call maps fresh pages
// for them and trims the heap again afterwards (M_TRIM_THRESHOLD / M_MMAP_THRESHOLD are 128 KiB): ~1.5 ms of page
// faults and munmap per 8192-element encrypt or decrypt, a third of the GPU time it wraps.  Once per process the heap
// is told to keep what it has been given (IPCL_MALLOC_TUNING=0 leaves the allocator alone).
static void tune_allocator_once() {
  static const bool done = [] {
    const char* e = std::getenv("IPCL_MALLOC_TUNING");
    if (e && std::atoi(e) == 0) return true;
    (void)mallopt(M_MMAP_THRESHOLD, 1 << 30);
    (void)mallopt(M_TRIM_THRESHOLD, 1 << 30);
    (void)mallopt(M_TOP_PAD, 64 << 20);
    return true;
  }();
  (void)done;
}

void ensure_context() {
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  tune_allocator_once();
  if (pgpu_is_initialized()) return;
  for (const char* var : {"IPCL_GPU_DEVICE", "LOCAL_RANK"}) {
    const char* v = std::getenv(var);
    if (v && *v) {
      IPCL_GPU_CHECK(pgpu_init(std::atoi(v)), "initializeContext");
      return;
    }
  }
  const char* v = std::getenv("IPCL_GPU_DEVICES");
  const int n = (v && *v && std::string(v) != "all") ? std::atoi(v) : 0;
  IPCL_GPU_CHECK(pgpu_init_all(n), "initializeContext");
}

}  // namespace detail

enum class RuntimeValue { DEFAULT, CPU, QAT, HYBRID, GPU };
static const std::map<std::string, RuntimeValue> runtimeMap = {
    {"DEFAULT", RuntimeValue::DEFAULT}, {"default", RuntimeValue::DEFAULT},
    {"CPU", RuntimeValue::CPU},         {"cpu", RuntimeValue::CPU},
    {"QAT", RuntimeValue::QAT},         {"qat", RuntimeValue::QAT},
    {"HYBRID", RuntimeValue::HYBRID},   {"hybrid", RuntimeValue::HYBRID},
    {"GPU", RuntimeValue::GPU},         {"gpu", RuntimeValue::GPU}};

bool initializeContext(const std::string runtime_choice) {
  (void)runtimeMap.at(runtime_choice);  // unknown spelling throws std::out_of_range
  detail::ensure_context();
  return true;
}

bool terminateContext() {
  detail::limb_cache_trim();       // retired limb arenas kept for reuse (bignum.h)
  detail::release_pinned_pool();   // idle staging blocks of the host layer (blocks still held by texts follow when those die)
  pgpu_shutdown();
  return true;
}

bool isQATRunning() { return false; }
bool isQATActive() { return false; }
bool isGPUActive() { return pgpu_is_initialized() != 0; }

}  // namespace ipcl

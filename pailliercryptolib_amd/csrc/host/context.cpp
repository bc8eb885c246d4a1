// ipcl::initializeContext / terminateContext over pgpu_init / pgpu_shutdown
// (reference ipcl/utils/context.cpp:40-88; there "QAT" acquires the accelerator, here every
// runtime choice does, because the GPU is the only compute path).
#include "ipcl/utils/context.hpp"

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

static int pick_device() {
  for (const char* var : {"IPCL_GPU_DEVICE", "LOCAL_RANK"}) {
    const char* v = std::getenv(var);
    if (v && *v) return std::atoi(v);
  }
  return 0;
}

void ensure_context() {
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  if (!pgpu_is_initialized()) IPCL_GPU_CHECK(pgpu_init(pick_device()), "initializeContext");
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
  pgpu_shutdown();
  return true;
}

bool isQATRunning() { return false; }
bool isQATActive() { return false; }
bool isGPUActive() { return pgpu_is_initialized() != 0; }

}  // namespace ipcl

// pailliercryptolib_amd -- error convention of the ipcl:: API.
// Mirrors reference ipcl/include/ipcl/utils/util.hpp:23-34: every precondition failure throws
// std::runtime_error("\nFile: ...\nLine: ...\nError: ...").  The x86 feature probing and OpenMP
// thread budgeting of the reference (util.hpp:49-113) have no counterpart: the batch runs on the GPU.
#ifndef PAILLIERCRYPTOLIB_AMD_IPCL_UTILS_UTIL_HPP_
#define PAILLIERCRYPTOLIB_AMD_IPCL_UTILS_UTIL_HPP_

#include <sstream>
#include <stdexcept>
#include <string>

#include "ipcl/utils/common.hpp"

namespace ipcl {

inline std::string build_log(const char* file, int line, const std::string& msg) {
  std::ostringstream log;
  log << "\nFile: " << file << "\nLine: " << line << "\nError: " << msg;
  return log.str();
}

#define ERROR_CHECK(e, ...)                                                        \
  do {                                                                             \
    if (!(e)) throw std::runtime_error(ipcl::build_log(__FILE__, __LINE__, __VA_ARGS__)); \
  } while (0)

namespace detail {
// turns a non-zero pgpu status into the ERROR_CHECK exception (never exit()).
void check_gpu(int status, const char* what, const char* file, int line);
#define IPCL_GPU_CHECK(call, what) ::ipcl::detail::check_gpu((call), (what), __FILE__, __LINE__)
}  // namespace detail

}  // namespace ipcl
#endif  // PAILLIERCRYPTOLIB_AMD_IPCL_UTILS_UTIL_HPP_

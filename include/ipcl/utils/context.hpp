// pailliercryptolib_amd -- runtime context (reference ipcl/include/ipcl/utils/context.hpp:25-44).
// In the reference only "QAT" does anything (acquire_qat_devices); here EVERY choice binds the
// process to the MI355X engine, because there is no CPU path.
#ifndef PAILLIERCRYPTOLIB_AMD_IPCL_UTILS_CONTEXT_HPP_
#define PAILLIERCRYPTOLIB_AMD_IPCL_UTILS_CONTEXT_HPP_

#include <string>

namespace ipcl {

// runtime_choice in {"DEFAULT","CPU","QAT","HYBRID","GPU"} (any case of the reference's
// spellings); an unknown string throws std::out_of_range like the reference's map lookup.
// The GPU is selected by env IPCL_GPU_DEVICE (default: LOCAL_RANK, else 0).
bool initializeContext(const std::string runtime_choice);
bool terminateContext(void);
bool isQATRunning(void);  // always false
bool isQATActive(void);   // always false
bool isGPUActive(void);

}  // namespace ipcl
#endif  // PAILLIERCRYPTOLIB_AMD_IPCL_UTILS_CONTEXT_HPP_

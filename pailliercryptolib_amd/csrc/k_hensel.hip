// pailliercryptolib_amd -- instantiations of the split-form CRT-decrypt exponentiation (hensel.hpp).
#include "hensel.hpp"
#include "launch.hpp"

namespace pgpu {

bool launch_hensel(int K, const HenselArgs& a, unsigned blocks, hipStream_t s) {
  if (K == 10) {
    hipLaunchKernelGGL((hensel_decrypt_kernel<10>), dim3(blocks), dim3(kWGThreads), 0, s, a);
    return true;
  }
  if (K == 19) {
    hipLaunchKernelGGL((hensel_decrypt_kernel<19>), dim3(blocks), dim3(kWGThreads), 0, s, a);
    return true;
  }
  return false;
}

}  // namespace pgpu

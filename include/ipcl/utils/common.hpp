// pailliercryptolib_amd -- constants and host randomness of the ipcl:: API
// (reference ipcl/include/ipcl/utils/common.hpp:13-58, ipcl/utils/common.cpp).
#ifndef PAILLIERCRYPTOLIB_AMD_IPCL_UTILS_COMMON_HPP_
#define PAILLIERCRYPTOLIB_AMD_IPCL_UTILS_COMMON_HPP_

#include <vector>

#include "ipcl/bignum.h"

namespace ipcl {

// kept for source compatibility; the GPU engine has no 8-lane or 1024-request granularity
constexpr int IPCL_CRYPTO_MB_SIZE = 8;
constexpr int IPCL_QAT_MODEXP_BATCH_SIZE = 1024;
constexpr int IPCL_WORKLOAD_SIZE_THRESHOLD = 128;
constexpr float IPCL_HYBRID_MODEXP_RATIO_FULL = 1.0;
constexpr float IPCL_HYBRID_MODEXP_RATIO_ENCRYPT = 0.25;
constexpr float IPCL_HYBRID_MODEXP_RATIO_DECRYPT = 0.12;
constexpr float IPCL_HYBRID_MODEXP_RATIO_MULTIPLY = 0.18;

// fills addr with OS entropy (reference: mt19937 seeded from random_device, common.cpp:96-101)
void rand32u(std::vector<Ipp32u>& addr);

// uniformly random integer of at most `bits` bits from the OS CSPRNG
// (reference: RDSEED -> RDRAND -> IPP PRNG chain, common.cpp:18-94)
BigNumber getRandomBN(int bits);

}  // namespace ipcl
#endif  // PAILLIERCRYPTOLIB_AMD_IPCL_UTILS_COMMON_HPP_

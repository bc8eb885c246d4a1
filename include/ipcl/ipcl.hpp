// pailliercryptolib_amd -- umbrella header (reference ipcl/include/ipcl/ipcl.hpp:19-37).
#ifndef PAILLIERCRYPTOLIB_AMD_IPCL_IPCL_HPP_
#define PAILLIERCRYPTOLIB_AMD_IPCL_IPCL_HPP_

#include "ipcl/mod_exp.hpp"
#include "ipcl/pri_key.hpp"
#include "ipcl/utils/context.hpp"
#include "ipcl/utils/serialize.hpp"

namespace ipcl {

struct KeyPair {
  PublicKey pub_key;
  PrivateKey priv_key;
};

// random probable prime of exactly maxBitSize bits (trial division + 10 Miller-Rabin rounds;
// reference keygen.cpp:13-40 asks ippsPrimeGen_BN for 10 trials too)
BigNumber getPrimeBN(int maxBitSize);

// n_length in [200, 4096], divisible by 4.  The reference caps at 2048 because mbx_exp_mb8 stops
// at 4096-bit moduli (keygen.cpp:10,93-100); the GPU kernels go to 8192-bit n^2, so the cap is
// lifted on purpose (BASELINE config 4 uses a 3072-bit key).
KeyPair generateKeypair(int64_t n_length, bool enable_DJN = true);

}  // namespace ipcl
#endif  // PAILLIERCRYPTOLIB_AMD_IPCL_IPCL_HPP_

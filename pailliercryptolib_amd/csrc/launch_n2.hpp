// pailliercryptolib_amd -- launcher of the one-lane product-scanning kernel of the n^2 domain (hensel_ps_n2.hpp; k_hensel.hip
// part 35).  A header of its own: launch.hpp is a dependency of every device translation unit.
#ifndef PAILLIERCRYPTOLIB_AMD_CSRC_LAUNCH_N2_HPP_
#define PAILLIERCRYPTOLIB_AMD_CSRC_LAUNCH_N2_HPP_

#include "launch.hpp"

namespace pgpu {

// base^exp modulo n^2 on resident pair rows with a whole exponentiation per lane: L2 = limbs per half of the key's pair rows
// (72: 2048-bit keys -> K = 75 limbs of 28 bits inside the kernel)
inline bool hensel_modexp_ps_has(int L2) { return L2 == 72; }
// 32-bit words of window table per wavefront: 2^w entries and one more (the masked gather's selected row), two parts of
// ceil(K/4) 16-byte rows of 64 lanes each
inline size_t hensel_modexp_ps_table_words(int L2, size_t entries) {
  const size_t K = L2 == 72 ? 75 : 0;
  return (entries + 1) * 2 * ((K + 3) / 4) * 64 * 4;
}
bool launch_hensel_modexp_ps_part35(int L2, const HenselModexpArgs& a, unsigned blocks, hipStream_t s);
inline bool launch_hensel_modexp_ps(int L2, const HenselModexpArgs& a, unsigned blocks, hipStream_t s) {
  return launch_hensel_modexp_ps_part35(L2, a, blocks, s);
}

}  // namespace pgpu

#endif  // PAILLIERCRYPTOLIB_AMD_CSRC_LAUNCH_N2_HPP_

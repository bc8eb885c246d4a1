// pailliercryptolib_amd -- the modexp seam (reference ipcl/include/ipcl/mod_exp.hpp:16-115).
#ifndef PAILLIERCRYPTOLIB_AMD_IPCL_MOD_EXP_HPP_
#define PAILLIERCRYPTOLIB_AMD_IPCL_MOD_EXP_HPP_

#include <vector>

#include "ipcl/bignum.h"

namespace ipcl {

// Hybrid CPU/QAT split knobs of the reference (mod_exp.hpp:16-63).  Kept so client code
// compiles and behaves the same (thread-local state, default {0.0, OPTIMAL}); like the
// reference built without QAT (mod_exp.cpp:35-56) the setters do not steer anything: the whole
// batch always goes to the GPU.
enum class HybridMode {
  OPTIMAL = 95, QAT = 100, PREF_QAT90 = 90, PREF_QAT80 = 80, PREF_QAT70 = 70, PREF_QAT60 = 60,
  HALF = 50, PREF_IPP60 = 40, PREF_IPP70 = 30, PREF_IPP80 = 20, PREF_IPP90 = 10, IPP = 0,
  UNDEFINED = -1
};
void setHybridMode(HybridMode mode);
void setHybridRatio(float qat_ratio, bool reset_mode = true);
void setHybridOff();
float getHybridRatio();
HybridMode getHybridMode();
bool isHybridOptimal();

// element-wise base[i]^exp[i] mod mod[i] on the GPU (mod_exp.hpp:72-74).  The three vectors must
// have equal sizes; elements sharing a modulus are batched into one kernel launch.
std::vector<BigNumber> modExp(const std::vector<BigNumber>& base, const std::vector<BigNumber>& exp,
                              const std::vector<BigNumber>& mod);
BigNumber modExp(const BigNumber& base, const BigNumber& exp, const BigNumber& mod);

// source-compatibility aliases of the reference's backend-specific entry points
// (mod_exp.hpp:85-115): there is one backend here.
std::vector<BigNumber> ippModExp(const std::vector<BigNumber>& base, const std::vector<BigNumber>& exp,
                                 const std::vector<BigNumber>& mod);
BigNumber ippModExp(const BigNumber& base, const BigNumber& exp, const BigNumber& mod);
std::vector<BigNumber> qatModExp(const std::vector<BigNumber>& base, const std::vector<BigNumber>& exp,
                                 const std::vector<BigNumber>& mod);  // throws: no QAT

// batched a[i]*b[i] mod m on the GPU (b may have size 1: scalar broadcast); used by CipherText
std::vector<BigNumber> modMul(const std::vector<BigNumber>& a, const std::vector<BigNumber>& b,
                              const BigNumber& mod);

}  // namespace ipcl
#endif  // PAILLIERCRYPTOLIB_AMD_IPCL_MOD_EXP_HPP_

// pailliercryptolib_amd -- Paillier public key (reference ipcl/include/ipcl/pub_key.hpp:18-190).
#ifndef PAILLIERCRYPTOLIB_AMD_IPCL_PUB_KEY_HPP_
#define PAILLIERCRYPTOLIB_AMD_IPCL_PUB_KEY_HPP_

#include <memory>
#include <vector>

#include "ipcl/bignum.h"
#include "ipcl/plaintext.hpp"

namespace ipcl {

class CipherText;
namespace detail { struct PubKeyDevice; }

class PublicKey {
 public:
  PublicKey() = default;
  ~PublicKey() = default;
  explicit PublicKey(const BigNumber& n, int bits = 1024, bool enableDJN_ = false);
  explicit PublicKey(const Ipp32u n, int bits = 1024, bool enableDJN_ = false)
      : PublicKey(BigNumber(n), bits, enableDJN_) {}

  void enableDJN();
  void setDJN(const BigNumber& hs, int randbit);

  // c = (1 + n*m) * obfuscator mod n^2 on the GPU (one fused kernel); make_secure=false skips
  // the obfuscator (pub_key.cpp:99-129)
  CipherText encrypt(const PlainText& plaintext, bool make_secure = true) const;

  std::shared_ptr<BigNumber> getN() const { return m_n; }
  std::shared_ptr<BigNumber> getNSQ() const { return m_nsquare; }
  std::shared_ptr<BigNumber> getG() const { return m_g; }
  int getBits() const { return m_bits; }
  int getDwords() const { return m_dwords; }

  void applyObfuscator(std::vector<BigNumber>& ciphertext) const;

  // inject the randomness (ISO/IEC 18033-6 compliance check; appends, pub_key.cpp:92-95)
  void setRandom(const std::vector<BigNumber>& r);
  void setHS(const BigNumber& hs);

  bool isDJN() const { return m_enable_DJN; }
  BigNumber getHS() const { return m_enable_DJN ? m_hs : BigNumber::Zero(); }
  int getRandBits() const { return m_enable_DJN ? m_randbits : -1; }
  bool isInitialized() { return m_isInitialized; }

  void create(const BigNumber& n, int bits, bool enableDJN_ = false);
  void create(const BigNumber& n, int bits, const BigNumber& hs, int randbits);

  // reference pub_key.hpp:133-164: "bits", "enable_DJN", "randbits", "n", "hs"
  void save(serializer::OutputArchive& ar) const;
  void load(serializer::InputArchive& ar);

 private:
  bool m_isInitialized = false;
  std::shared_ptr<BigNumber> m_n;
  std::shared_ptr<BigNumber> m_g;
  std::shared_ptr<BigNumber> m_nsquare;
  int m_bits = 0;
  int m_dwords = 0;
  BigNumber m_hs;
  int m_randbits = 0;
  bool m_enable_DJN = false;
  // injected randomness (setRandom); shared between copies of the key -- every CipherText holds a copy of its
  // PublicKey, as in the reference (ciphertext.cpp:12-22), and must not drag a batch of BigNumbers along
  std::shared_ptr<const std::vector<BigNumber>> m_r;
  bool m_testv = false;
  // device copy of the injected randomness (setRandom), built at the first encrypt that uses it: the same values feed
  // every encrypt until setRandom is called again (pub_key.cpp:92-95), so they are packed and uploaded once
  struct InjectedRandom;
  mutable std::shared_ptr<InjectedRandom> m_r_dev;
  // device-side key (n^2 Montgomery context, hs; one copy per pool GPU): rebuilt by every mutator, read-only
  // in between, shared by copies of the key
  std::shared_ptr<detail::PubKeyDevice> m_dev;
  friend class CipherText;

  std::vector<BigNumber> raw_encrypt(const std::vector<BigNumber>& pt, bool make_secure = true) const;
  std::vector<BigNumber> drawRandom(std::size_t sz) const;
  std::shared_ptr<detail::PubKeyDevice> device() const;
  void setFields(const BigNumber& n, int bits);
  void rebuildDevice();
};

}  // namespace ipcl
#endif  // PAILLIERCRYPTOLIB_AMD_IPCL_PUB_KEY_HPP_

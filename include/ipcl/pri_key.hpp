// pailliercryptolib_amd -- Paillier private key (reference ipcl/include/ipcl/pri_key.hpp:29-193).
#ifndef PAILLIERCRYPTOLIB_AMD_IPCL_PRI_KEY_HPP_
#define PAILLIERCRYPTOLIB_AMD_IPCL_PRI_KEY_HPP_

#include <memory>
#include <vector>

#include "ipcl/ciphertext.hpp"
#include "ipcl/mod_exp.hpp"
#include "ipcl/plaintext.hpp"

namespace ipcl {

namespace detail { struct PrivKeyDevice; }

// lcm(p, q) (reference pri_key.hpp:23-27)
BigNumber lcm(const BigNumber& p, const BigNumber& q);

class PrivateKey {
 public:
  PrivateKey() = default;
  ~PrivateKey() = default;
  PrivateKey(const PublicKey& pk, const BigNumber& p, const BigNumber& q);
  PrivateKey(const BigNumber& n, const BigNumber& p, const BigNumber& q);

  void enableCRT(bool crt) { m_enable_crt = crt; }  // default: CRT on (pri_key.cpp:18,44)

  // CRT path: one fused GPU pipeline (two half-width modexps + L function + recombination);
  // RAW path: c^lambda mod n^2 on the GPU, then the host L function (pri_key.cpp:92-111)
  PlainText decrypt(const CipherText& ciphertext) const;

  std::shared_ptr<BigNumber> getN() const { return m_n; }
  std::shared_ptr<BigNumber> getP() const { return m_p; }
  std::shared_ptr<BigNumber> getQ() const { return m_q; }
  BigNumber getLambda() const { return m_lambda; }
  bool isInitialized() { return m_isInitialized; }

  // reference pri_key.hpp:93-133: "bits" (of p), "p", "q"; load re-derives every constant
  void save(serializer::OutputArchive& ar) const;
  void load(serializer::InputArchive& ar);

 private:
  bool m_isInitialized = false;
  bool m_enable_crt = false;
  std::shared_ptr<BigNumber> m_n, m_nsquare, m_g, m_p, m_q;
  BigNumber m_pminusone, m_qminusone, m_psquare, m_qsquare, m_pinverse, m_hp, m_hq, m_lambda, m_x;
  std::shared_ptr<detail::PrivKeyDevice> m_dev;

  void precompute(const BigNumber& p, const BigNumber& q);
  BigNumber computeLfun(const BigNumber& a, const BigNumber& b) const;
  BigNumber computeHfun(const BigNumber& a, const BigNumber& b) const;
  void decryptRAW(std::vector<BigNumber>& plaintext, const std::vector<BigNumber>& ciphertext) const;
  void decryptCRT(std::vector<BigNumber>& plaintext, const std::vector<BigNumber>& ciphertext) const;
};

}  // namespace ipcl
#endif  // PAILLIERCRYPTOLIB_AMD_IPCL_PRI_KEY_HPP_

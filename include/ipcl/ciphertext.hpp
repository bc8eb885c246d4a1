// pailliercryptolib_amd -- CipherText (reference ipcl/include/ipcl/ciphertext.hpp:16-75).
#ifndef PAILLIERCRYPTOLIB_AMD_IPCL_CIPHERTEXT_HPP_
#define PAILLIERCRYPTOLIB_AMD_IPCL_CIPHERTEXT_HPP_

#include <memory>
#include <vector>

#include "ipcl/plaintext.hpp"
#include "ipcl/pub_key.hpp"
#include "ipcl/utils/util.hpp"

namespace ipcl {

class CipherText : public BaseText {
 public:
  CipherText() = default;
  ~CipherText() = default;
  CipherText(const PublicKey& pk, const uint32_t& n);
  CipherText(const PublicKey& pk, const std::vector<uint32_t>& n_v);
  CipherText(const PublicKey& pk, const BigNumber& bn);
  CipherText(const PublicKey& pk, const std::vector<BigNumber>& bn_vec);
  CipherText(const CipherText& ct) = default;
  CipherText& operator=(const CipherText& other) = default;

  CipherText operator+(const CipherText& other) const;  // CT+CT: batched modmul mod n^2 on the GPU
  CipherText operator+(const PlainText& other) const;   // CT+PT
  CipherText operator*(const PlainText& other) const;   // CT*PT: batched modexp on the GPU

  CipherText getCipherText(const size_t& idx) const;
  std::shared_ptr<PublicKey> getPubKey() const;
  CipherText rotate(int shift) const;

  void save(serializer::OutputArchive& ar) const;   // reference ciphertext.hpp:69-74: base, "pk"
  void load(serializer::InputArchive& ar);

 private:
  friend class PublicKey;
  CipherText(const PublicKey& pk, std::shared_ptr<detail::DeviceBatch> dev);
  CipherText(std::shared_ptr<PublicKey> pk, std::shared_ptr<detail::DeviceBatch> dev);
  std::shared_ptr<PublicKey> m_pk;
};

}  // namespace ipcl
#endif  // PAILLIERCRYPTOLIB_AMD_IPCL_CIPHERTEXT_HPP_

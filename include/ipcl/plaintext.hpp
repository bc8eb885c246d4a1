// pailliercryptolib_amd -- PlainText (reference ipcl/include/ipcl/plaintext.hpp:18-98).
#ifndef PAILLIERCRYPTOLIB_AMD_IPCL_PLAINTEXT_HPP_
#define PAILLIERCRYPTOLIB_AMD_IPCL_PLAINTEXT_HPP_

#include <vector>

#include "ipcl/base_text.hpp"

namespace ipcl {

class CipherText;

class PlainText : public BaseText {
 public:
  PlainText() = default;
  ~PlainText() = default;
  explicit PlainText(const uint32_t& n);
  explicit PlainText(const std::vector<uint32_t>& n_v);
  explicit PlainText(const BigNumber& bn);
  explicit PlainText(const std::vector<BigNumber>& bn_v);
  PlainText(const PlainText& pt) = default;
  PlainText& operator=(const PlainText& other) = default;

  operator std::vector<uint32_t>() const;  // first 32-bit word of element 0 .. (plaintext.cpp:37-46)
  operator BigNumber() const;              // element 0
  operator std::vector<BigNumber>() const;

  CipherText operator+(const CipherText& other) const;  // PT + CT
  CipherText operator*(const CipherText& other) const;  // PT * CT
  PlainText rotate(int shift) const;

  void save(serializer::OutputArchive& ar) const;   // reference plaintext.hpp:92-98
  void load(serializer::InputArchive& ar);

 private:
  friend class PrivateKey;
  explicit PlainText(std::shared_ptr<detail::DeviceBatch> dev);
};

}  // namespace ipcl
#endif  // PAILLIERCRYPTOLIB_AMD_IPCL_PLAINTEXT_HPP_

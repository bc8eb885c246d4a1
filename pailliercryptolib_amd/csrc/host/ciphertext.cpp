// ipcl::CipherText -- homomorphic operations (reference ipcl/ciphertext.cpp).
// CT+CT is one batched GPU modmul (the reference: per-element BigNumber multiply + divide under
// OpenMP, ciphertext.cpp:35-72,135-141); CT*PT is one batched GPU modexp (ciphertext.cpp:83-106).
#include "ipcl/ciphertext.hpp"

#include "detail.hpp"
#include "ipcl/mod_exp.hpp"

namespace ipcl {

namespace detail {
std::vector<BigNumber> rotated(const std::vector<BigNumber>& v, int shift);
}

CipherText::CipherText(const PublicKey& pk, const uint32_t& n)
    : BaseText(n), m_pk(std::make_shared<PublicKey>(pk)) {}
CipherText::CipherText(const PublicKey& pk, const std::vector<uint32_t>& n_v)
    : BaseText(n_v), m_pk(std::make_shared<PublicKey>(pk)) {}
CipherText::CipherText(const PublicKey& pk, const BigNumber& bn)
    : BaseText(bn), m_pk(std::make_shared<PublicKey>(pk)) {}
CipherText::CipherText(const PublicKey& pk, const std::vector<BigNumber>& bn_v)
    : BaseText(bn_v), m_pk(std::make_shared<PublicKey>(pk)) {}

CipherText CipherText::operator+(const CipherText& other) const {
  std::size_t b_size = other.getSize();
  ERROR_CHECK(this->m_size == b_size || b_size == 1, "CT + CT error: Size mismatch!");
  ERROR_CHECK(*(m_pk->getN()) == *(other.m_pk->getN()),
              "CT + CT error: 2 different public keys detected!");
  return CipherText(*m_pk, modMul(m_texts, other.m_texts, *(m_pk->getNSQ())));
}

CipherText CipherText::operator+(const PlainText& other) const {
  CipherText b = m_pk->encrypt(other, false);  // g^m without obfuscator (ciphertext.cpp:75-80)
  return *this + b;
}

CipherText CipherText::operator*(const PlainText& other) const {
  std::size_t b_size = other.getSize();
  ERROR_CHECK(this->m_size == b_size || b_size == 1, "CT * PT error: Size mismatch!");
  std::vector<BigNumber> e = other.getTexts();
  if (b_size == 1 && m_size > 1) e.assign(m_size, other.getElement(0));
  std::vector<BigNumber> sq(m_size, *(m_pk->getNSQ()));
  return CipherText(*m_pk, modExp(m_texts, e, sq));
}

CipherText CipherText::getCipherText(const size_t& idx) const {
  ERROR_CHECK(idx < m_size, "CipherText::getCipherText index is out of range");
  return CipherText(*m_pk, m_texts[idx]);
}

std::shared_ptr<PublicKey> CipherText::getPubKey() const { return m_pk; }

CipherText CipherText::rotate(int shift) const { return CipherText(*m_pk, detail::rotated(m_texts, shift)); }

}  // namespace ipcl

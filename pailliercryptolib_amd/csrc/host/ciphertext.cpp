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
CipherText::CipherText(const PublicKey& pk, std::shared_ptr<detail::DeviceBatch> dev)
    : BaseText(std::move(dev)), m_pk(std::make_shared<PublicKey>(pk)) {}
CipherText::CipherText(std::shared_ptr<PublicKey> pk, std::shared_ptr<detail::DeviceBatch> dev)
    : BaseText(std::move(dev)), m_pk(std::move(pk)) {}

CipherText CipherText::operator+(const CipherText& other) const {
  std::size_t b_size = other.getSize();
  ERROR_CHECK(this->m_size == b_size || b_size == 1, "CT + CT error: Size mismatch!");
  ERROR_CHECK(*(m_pk->getN()) == *(other.m_pk->getN()),
              "CT + CT error: 2 different public keys detected!");
  ERROR_CHECK(m_size > 0, "CT + CT error: empty CipherText");
  // one batched modmul mod n^2 on resident batches; the sum stays resident
  const BigNumber& nsq = *(m_pk->getNSQ());
  const int W = detail::words_for_bits(nsq.BitSize());
  auto da = deviceBatch(W, &nsq), db = other.deviceBatch(W, &nsq);
  auto dout = detail::DeviceBatch::alloc(m_size, W);
  std::vector<uint64_t> mod((size_t)W);
  nsq.toLimbs64(mod.data(), (size_t)W);
  const size_t bstride = (b_size == m_size) ? (size_t)W : 0;   // size-1 right operand: broadcast
  IPCL_GPU_CHECK(pgpu_modmul_dev(da->ptr(), db->ptr(), bstride, mod.data(), W, dout->ptr(), m_size, nullptr),
                 "CT + CT");
  return CipherText(m_pk, dout);
}

CipherText CipherText::operator+(const PlainText& other) const {
  CipherText b = m_pk->encrypt(other, false);  // g^m without obfuscator (ciphertext.cpp:75-80)
  return *this + b;
}

CipherText CipherText::operator*(const PlainText& other) const {
  std::size_t b_size = other.getSize();
  ERROR_CHECK(this->m_size == b_size || b_size == 1, "CT * PT error: Size mismatch!");
  ERROR_CHECK(m_size > 0, "CT * PT error: empty CipherText");
  // one batched modexp (plaintexts as exponents) on resident batches
  const BigNumber& nsq = *(m_pk->getNSQ());
  const int W = detail::words_for_bits(nsq.BitSize());
  if (!other.isDeviceResident())
    for (const auto& e : other.m_texts) ERROR_CHECK(!e.isNegative(), "CT * PT error: negative plaintext");
  const int ebits = other.maxBitsHint();
  const int ew = other.isDeviceResident() ? other.m_dev->words : detail::words_for_bits(ebits);
  auto dbase = deviceBatch(W, &nsq), dexp = other.deviceBatch(ew);
  auto dout = detail::DeviceBatch::alloc(m_size, W);
  std::vector<uint64_t> mod((size_t)W);
  nsq.toLimbs64(mod.data(), (size_t)W);
  const size_t estride = (b_size == m_size) ? (size_t)ew : 0;   // scalar plaintext: shared exponent
  IPCL_GPU_CHECK(pgpu_modexp_dev(dbase->ptr(), (size_t)W, dexp->ptr(), estride, ew, ebits, mod.data(), W,
                                 dout->ptr(), m_size, nullptr),
                 "CT * PT");
  return CipherText(m_pk, dout);
}

void CipherText::save(serializer::OutputArchive& ar) const {
  ERROR_CHECK(m_pk != nullptr, "CipherText: cannot serialize without a public key");
  ar.class_version("ipcl::CipherText");
  BaseText::save(ar);
  m_pk->save(ar);
}

void CipherText::load(serializer::InputArchive& ar) {
  (void)ar.class_version("ipcl::CipherText");
  BaseText::load(ar);
  auto pk = std::make_shared<PublicKey>();
  pk->load(ar);
  m_pk = pk;
}

CipherText CipherText::getCipherText(const size_t& idx) const {
  ERROR_CHECK(idx < m_size, "CipherText::getCipherText index is out of range");
  ensureHost();
  return CipherText(*m_pk, m_texts[idx]);
}

std::shared_ptr<PublicKey> CipherText::getPubKey() const { return m_pk; }

CipherText CipherText::rotate(int shift) const {
  ensureHost();
  return CipherText(*m_pk, detail::rotated(m_texts, shift));
}

}  // namespace ipcl

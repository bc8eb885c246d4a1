// ipcl::CipherText -- homomorphic operations (reference ipcl/ciphertext.cpp).
// CT+CT is one batched GPU modmul (the reference: per-element BigNumber multiply + divide under
// OpenMP, ciphertext.cpp:35-72,135-141); CT*PT is one batched GPU modexp (ciphertext.cpp:83-106).
#include "ipcl/ciphertext.hpp"

#include <algorithm>

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
  // one batched Montgomery product mod n^2 on resident batches (sharded over the pool); the sum stays resident
  const BigNumber& nsq = *(m_pk->getNSQ());
  const int W = detail::words_for_bits(nsq.BitSize());
  auto da = deviceBatch(W, &nsq), db = other.deviceBatch(W, &nsq);
  pgpu_batch* o = nullptr;
  IPCL_GPU_CHECK(pgpu_batch_ct_add(m_pk->device()->h, da->h, db->h, &o), "CT + CT");
  return CipherText(m_pk, detail::DeviceBatch::adopt(o));
}

// CT + PT: multiply by g^m = 1 + n*m without obfuscator (ciphertext.cpp:75-80 builds it with a host loop in
// raw_encrypt and then runs CT + CT; here one fused launch forms g^m and the product)
CipherText CipherText::operator+(const PlainText& other) const {
  std::size_t b_size = other.getSize();
  ERROR_CHECK(b_size > 0, "encrypt: Cannot encrypt empty PlainText");   // what encrypt(other, false) reports
  ERROR_CHECK(this->m_size == b_size || b_size == 1, "CT + CT error: Size mismatch!");
  ERROR_CHECK(m_size > 0, "CT + CT error: empty CipherText");
  const BigNumber& nsq = *(m_pk->getNSQ());
  const int W = detail::words_for_bits(nsq.BitSize());
  // g^m only depends on m mod n: reduce plaintexts that are negative or wider than n^2
  auto da = deviceBatch(W, &nsq);
  auto dm = other.operandBatch(W, m_pk->getN().get());
  pgpu_batch* o = nullptr;
  IPCL_GPU_CHECK(pgpu_batch_ct_add_plain(m_pk->device()->h, da->h, dm->h, &o), "CT + PT");
  return CipherText(m_pk, detail::DeviceBatch::adopt(o));
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
  pgpu_batch* o = nullptr;
  IPCL_GPU_CHECK(pgpu_batch_ct_mul(m_pk->device()->h, dbase->h, dexp->h, ebits, &o), "CT * PT");
  return CipherText(m_pk, detail::DeviceBatch::adopt(o));
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

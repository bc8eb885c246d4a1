// ipcl::PrivateKey -- decrypt path (reference ipcl/pri_key.cpp).
// The reference reduces c mod p^2 / q^2 on the host under OpenMP, calls modExp twice and runs the
// L function + CRT in another host loop (pri_key.cpp:114-157); here ONE GPU pipeline does all of it.
#include "ipcl/pri_key.hpp"

#include "detail.hpp"
#include "ipcl/utils/util.hpp"

namespace ipcl {

BigNumber lcm(const BigNumber& p, const BigNumber& q) { return p * q / p.gcd(q); }

PrivateKey::PrivateKey(const PublicKey& pk, const BigNumber& p, const BigNumber& q)
    : m_n(pk.getN()), m_nsquare(pk.getNSQ()), m_g(pk.getG()) {
  ERROR_CHECK(m_n != nullptr, "PrivateKey ctor: Public key is NOT initialized.");
  precompute(p, q);
}

PrivateKey::PrivateKey(const BigNumber& n, const BigNumber& p, const BigNumber& q)
    : m_n(std::make_shared<BigNumber>(n)),
      m_nsquare(std::make_shared<BigNumber>(n * n)),
      m_g(std::make_shared<BigNumber>(n + 1)) {
  precompute(p, q);
}

// key constants (reference pri_key.cpp:13-37): p < q, p^-1 mod q, hp, hq, lambda, x
void PrivateKey::precompute(const BigNumber& p_in, const BigNumber& q_in) {
  const bool swap = q_in < p_in;
  m_p = std::make_shared<BigNumber>(swap ? q_in : p_in);
  m_q = std::make_shared<BigNumber>(swap ? p_in : q_in);
  const BigNumber &p = *m_p, &q = *m_q;
  ERROR_CHECK(p * q == *m_n, "PrivateKey ctor: Public key does not match p * q.");
  ERROR_CHECK(p != q, "PrivateKey ctor: p and q are same");
  m_enable_crt = true;
  m_pminusone = p - 1;
  m_qminusone = q - 1;
  m_psquare = p * p;
  m_qsquare = q * q;
  m_pinverse = q.InverseMul(p);
  m_hp = computeHfun(p, m_psquare);
  m_hq = computeHfun(q, m_qsquare);
  m_lambda = lcm(m_pminusone, m_qminusone);
  // x = L(g^lambda mod n^2)^-1 mod n (pri_key.cpp:33-34); g = n + 1, so g^lambda = 1 + lambda*n (mod n^2) and
  // L(.) = lambda mod n: the same value without an exponentiation
  m_x = m_n->InverseMul(m_lambda % *m_n);
  // device-side key: Montgomery contexts mod p^2, q^2, recombination constants
  detail::ensure_context();
  auto d = std::make_shared<detail::PrivKeyDevice>();
  const int pw = detail::words_for_bits(q.BitSize());
  std::vector<uint64_t> pl((size_t)pw), ql((size_t)pw);
  p.toLimbs64(pl.data(), (size_t)pw);
  q.toLimbs64(ql.data(), (size_t)pw);
  IPCL_GPU_CHECK(pgpu_privkey_create(pl.data(), ql.data(), pw, &d->h), "PrivateKey");
  m_dev = d;
  m_isInitialized = true;
}

void PrivateKey::save(serializer::OutputArchive& ar) const {
  ERROR_CHECK(m_isInitialized, "PrivateKey: cannot serialize an uninitialized key");
  ar.class_version("ipcl::PrivateKey");
  ar.i32(m_p->BitSize());
  m_p->save(ar);
  m_q->save(ar);
}

void PrivateKey::load(serializer::InputArchive& ar) {
  (void)ar.class_version("ipcl::PrivateKey");
  (void)ar.i32();
  BigNumber p, q;
  p.load(ar);
  q.load(ar);
  ERROR_CHECK(!p.isNegative() && !q.isNegative() && p.IsOdd() && q.IsOdd() && p > BigNumber::Two() && q > BigNumber::Two(),
              "PrivateKey: corrupt archive (p and q must be odd primes)");
  m_n = std::make_shared<BigNumber>(p * q);
  m_nsquare = std::make_shared<BigNumber>((*m_n) * (*m_n));
  m_g = std::make_shared<BigNumber>((*m_n) + 1);
  precompute(p, q);
}

BigNumber PrivateKey::computeLfun(const BigNumber& a, const BigNumber& b) const { return (a - 1) / b; }

// h = L_a(g^(a-1) mod a^2)^-1 mod a  (reference pri_key.cpp:159-167).  The reference's g is n + 1
// (pri_key.cpp:47) and n^2 == 0 (mod a^2) for a in {p, q}, so the binomial expansion of (1 + n)^(a-1) stops after
// two terms: g^(a-1) mod a^2 = 1 + (a-1)*n mod a^2 -- identical value, no exponentiation.
BigNumber PrivateKey::computeHfun(const BigNumber& a, const BigNumber& b) const {
  BigNumber pm = (BigNumber::One() + (a - 1) * *m_n) % b;
  return a.InverseMul(computeLfun(pm, a));
}

PlainText PrivateKey::decrypt(const CipherText& ct) const {
  ERROR_CHECK(m_isInitialized, "decrypt: Private key is NOT initialized.");
  ERROR_CHECK(*(ct.getPubKey()->getN()) == *(this->getN()),
              "decrypt: The value of N in public key mismatch.");
  std::size_t ct_size = ct.getSize();
  ERROR_CHECK(ct_size > 0, "decrypt: Cannot decrypt empty CipherText");
  if (m_enable_crt) {
    // fused GPU pipeline on the (possibly already resident) ciphertext batch; the plaintexts stay
    // resident until an accessor needs them
    const int nw = detail::words_for_bits(m_n->BitSize());
    auto dc = ct.deviceBatch(2 * nw, m_nsquare.get());
    pgpu_batch* m = nullptr;
    IPCL_GPU_CHECK(pgpu_batch_decrypt_crt(m_dev->h, dc->h, &m), "decrypt");
    auto dm = detail::DeviceBatch::adopt(m);
    return PlainText(dm);
  }
  std::vector<BigNumber> pt_bn(ct_size);
  decryptRAW(pt_bn, ct.getTexts());
  return PlainText(pt_bn);
}

// m = L(c^lambda mod n^2) * x mod n   (reference pri_key.cpp:92-111); modexp on the GPU
void PrivateKey::decryptRAW(std::vector<BigNumber>& plaintext, const std::vector<BigNumber>& ciphertext) const {
  const std::size_t sz = ciphertext.size();
  std::vector<BigNumber> res = modExp(ciphertext, std::vector<BigNumber>(sz, m_lambda),
                                      std::vector<BigNumber>(sz, *m_nsquare));
  // L function and * x on the host team (reference pri_key.cpp:104-110, under OpenMP there as well)
  detail::parallel_for(sz, 32, [&](std::size_t i) { plaintext[i] = (computeLfun(res[i], *m_n) * m_x) % *m_n; });
}

// host-vector variant of the CRT path (kept for the private interface of the reference class)
void PrivateKey::decryptCRT(std::vector<BigNumber>& plaintext, const std::vector<BigNumber>& ciphertext) const {
  PublicKey pk(*m_n, m_n->BitSize());
  plaintext = decrypt(CipherText(pk, ciphertext)).getTexts();
}

}  // namespace ipcl

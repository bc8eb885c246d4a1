// ipcl::PublicKey -- encrypt path (reference ipcl/pub_key.cpp).
// The reference computes (n*m+1) % n^2 in a serial host loop, builds N-element vectors of hs and
// n^2, calls modExp, then multiplies in another serial host loop (pub_key.cpp:51-110).  Here the
// plaintexts and the randomness are marshalled once and ONE fused GPU call returns the ciphertexts.
#include "ipcl/pub_key.hpp"

#include <algorithm>
#include <cstddef>

#include "detail.hpp"
#include "ipcl/ciphertext.hpp"
#include "ipcl/mod_exp.hpp"
#include "ipcl/utils/util.hpp"

namespace ipcl {

// (pub_key.hpp) what an encrypt needs from the injected randomness, derived once per setRandom
struct PublicKey::InjectedRandom {
  std::shared_ptr<const std::vector<BigNumber>> from;   // the vector it was derived from
  std::shared_ptr<detail::DeviceBatch> dev;
  int bits = 0;
};

PublicKey::PublicKey(const BigNumber& n, int bits, bool enableDJN_) { create(n, bits, enableDJN_); }

void PublicKey::setFields(const BigNumber& n, int bits) {
  m_n = std::make_shared<BigNumber>(n);
  m_g = std::make_shared<BigNumber>(n + 1);
  m_nsquare = std::make_shared<BigNumber>(n * n);
  m_bits = bits;
  m_dwords = BITSIZE_DWORD(bits * 2);
  m_enable_DJN = false;
  m_hs = BigNumber::Zero();
  m_randbits = 0;
  m_testv = false;
  m_r.reset();
  m_dev.reset();
}

void PublicKey::create(const BigNumber& n, int bits, bool enableDJN_) {
  setFields(n, bits);
  if (enableDJN_) enableDJN();   // (draws hs, then builds the device key)
  else rebuildDevice();
  m_isInitialized = true;
}

// (the path PublicKey::load takes for DJN keys: fields first, ONE device-key build)
void PublicKey::create(const BigNumber& n, int bits, const BigNumber& hs, int randbits) {
  setFields(n, bits);
  m_enable_DJN = true;
  m_hs = hs;
  m_randbits = randbits;
  rebuildDevice();
  m_isInitialized = true;
}

// hs = (-x^2 mod n)^n mod n^2 for a random x coprime to n (reference pub_key.cpp:29-49)
void PublicKey::enableDJN() {
  const BigNumber& n = *m_n;
  BigNumber x;
  do {
    x = getRandomBN(n.BitSize() + 128);
  } while (x.gcd(n) != BigNumber::One());
  BigNumber xm = x % n;
  BigNumber h = (BigNumber::Zero() - xm * xm) % n;  // non-negative residue
  m_hs = modExp(h, n, *m_nsquare);
  m_randbits = m_bits >> 1;
  m_enable_DJN = true;
  rebuildDevice();
}

void PublicKey::setDJN(const BigNumber& hs, int randbit) {
  if (m_enable_DJN) return;
  m_hs = hs;
  m_randbits = randbit;
  m_enable_DJN = true;
  rebuildDevice();
}

void PublicKey::setRandom(const std::vector<BigNumber>& r) {
  auto all = std::make_shared<std::vector<BigNumber>>();
  if (m_r) *all = *m_r;
  all->insert(all->end(), r.begin(), r.end());   // appends, like pub_key.cpp:92-95
  m_r = std::move(all);
  m_r_dev.reset();
  m_testv = true;
}

void PublicKey::setHS(const BigNumber& hs) {
  m_hs = hs;
  rebuildDevice();
}

// The device-side key (n^2 Montgomery context, hs, one copy per pool GPU) is built EAGERLY by every mutator, so a
// key that is shared between threads afterwards (the reference's APPLEVEL_OMP pattern: generateKeypair, then four
// threads encrypt) is only ever read: no lazy initialisation from const methods.
void PublicKey::rebuildDevice() {
  detail::ensure_context();
  auto d = std::make_shared<detail::PubKeyDevice>();
  d->n = *m_n;
  d->hs = m_hs;
  d->djn = m_enable_DJN;
  const int nw = detail::words_for_bits(m_n->BitSize());
  std::vector<uint64_t> n_l((size_t)nw), hs_l((size_t)2 * nw);
  m_n->toLimbs64(n_l.data(), (size_t)nw);
  if (m_enable_DJN)
    ERROR_CHECK((m_hs % *m_nsquare).toLimbs64(hs_l.data(), hs_l.size()), "PublicKey: hs does not fit n^2");
  IPCL_GPU_CHECK(pgpu_pubkey_create(n_l.data(), nw, m_enable_DJN ? hs_l.data() : nullptr, &d->h),
                 "PublicKey");
  m_dev = d;
}

std::shared_ptr<detail::PubKeyDevice> PublicKey::device() const {
  ERROR_CHECK(m_dev != nullptr, "PublicKey: key is NOT initialized.");
  return m_dev;
}

// the per-element randomness: injected (setRandom) or drawn on the host like the reference
// (DJN: randbits random bits, pub_key.cpp:59-61; otherwise uniform in [1, n-1], pub_key.cpp:74-76)
std::vector<BigNumber> PublicKey::drawRandom(std::size_t sz) const {
  if (m_testv) return m_r ? *m_r : std::vector<BigNumber>();  // used as is: size is checked by the caller like ippMBModExp does
  std::vector<BigNumber> r(sz);
  // one bulk read of the kernel CSPRNG for the whole batch
  const int bits = m_enable_DJN ? m_randbits : m_bits;
  ERROR_CHECK(bits > 0, "getRandomBN: bit length must be positive");
  const std::size_t w32 = (std::size_t)BITSIZE_WORD(bits);
  std::vector<Ipp32u> pool(sz * w32);
  detail::fill_random(pool.data(), pool.size() * sizeof(Ipp32u));
  const BigNumber nm1 = *m_n - 1;
  const bool djn = m_enable_DJN;
  detail::parallel_for(sz, 256, [&](std::size_t i) {
    Ipp32u* w = pool.data() + i * w32;
    if (bits % 32) w[w32 - 1] &= (1u << (bits % 32)) - 1;
    BigNumber x(w, (int)w32);
    r[i] = djn ? x : x % nm1 + 1;
  });
  detail::wipe(pool.data(), pool.size() * sizeof(Ipp32u));
  return r;
}

// g^m only (make_secure = false, used by CT+PT): cheap, stays on the host (pub_key.cpp:105)
std::vector<BigNumber> PublicKey::raw_encrypt(const std::vector<BigNumber>& pt, bool make_secure) const {
  ERROR_CHECK(!make_secure, "raw_encrypt: the obfuscated path runs through encrypt()");
  const BigNumber& n = *m_n;
  const BigNumber& nsq = *m_nsquare;
  std::vector<BigNumber> ct(pt.size());
  detail::parallel_for(pt.size(), 64, [&](std::size_t i) { ct[i] = (n * pt[i] + 1) % nsq; });   // pub_key.cpp:105
  return ct;
}

void PublicKey::save(serializer::OutputArchive& ar) const {
  ERROR_CHECK(m_isInitialized, "PublicKey: cannot serialize an uninitialized key");
  ar.class_version("ipcl::PublicKey");
  ar.i32(m_bits);
  ar.boolean(m_enable_DJN);
  ar.i32(m_randbits);
  m_n->save(ar);
  m_hs.save(ar);
}

void PublicKey::load(serializer::InputArchive& ar) {
  (void)ar.class_version("ipcl::PublicKey");
  int bits = ar.i32();
  bool enable_DJN = ar.boolean();
  int randbits = ar.i32();
  BigNumber n, hs;
  n.load(ar);
  hs.load(ar);
  // a key from an archive is untrusted input: refuse what no generateKeypair / create could have produced
  ERROR_CHECK(bits > 0 && randbits >= 0, "PublicKey: corrupt archive (bits / randbits)");
  ERROR_CHECK(!n.isNegative() && n.IsOdd() && n > BigNumber::One(), "PublicKey: corrupt archive (n must be odd and > 1)");
  ERROR_CHECK(!hs.isNegative(), "PublicKey: corrupt archive (negative hs)");
  if (enable_DJN) create(n, bits, hs, randbits);
  else create(n, bits);
}

// kept for API compatibility: multiplies the given values by fresh obfuscators in place
void PublicKey::applyObfuscator(std::vector<BigNumber>& ciphertext) const {
  const std::size_t sz = ciphertext.size();
  std::vector<BigNumber> r = drawRandom(sz);
  ERROR_CHECK(r.size() == sz, "ippMBModExp: input vector size error");
  std::vector<BigNumber> sq(sz, *m_nsquare), obf;
  if (m_enable_DJN) obf = modExp(std::vector<BigNumber>(sz, m_hs), r, sq);
  else obf = modExp(r, std::vector<BigNumber>(sz, *m_n), sq);
  ciphertext = modMul(ciphertext, obf, *m_nsquare);
}

CipherText PublicKey::encrypt(const PlainText& pt, bool make_secure) const {
  ERROR_CHECK(m_isInitialized, "encrypt: Public key is NOT initialized.");
  const std::size_t sz = pt.getSize();
  ERROR_CHECK(sz > 0, "encrypt: Cannot encrypt empty PlainText");
  if (!make_secure) return CipherText(*this, raw_encrypt(pt.getTexts(), false));

  // One fused GPU launch; plaintexts may already be resident (e.g. the output of decrypt), the
  // ciphertexts stay resident until somebody asks for their BigNumbers.
  if (m_enable_DJN && !m_testv && m_randbits > 0) {
    // DJN exponents are plain randbits-bit strings (pub_key.cpp:59-61): drawn straight into the limb batch the GPU
    // reads, without a BigNumber per element in between
    auto dev = device();
    const int nw = detail::words_for_bits(m_n->BitSize());
    std::shared_ptr<detail::DeviceBatch> dm = pt.operandBatch(2 * nw, m_n.get());
    const int rw = detail::words_for_bits(m_randbits);
    std::vector<uint64_t> flat(sz * (std::size_t)rw);
    detail::fill_random(flat.data(), flat.size() * sizeof(uint64_t));
    if (m_randbits % 64) {
      const uint64_t top = (~(uint64_t)0) >> (64 - m_randbits % 64);
      for (std::size_t i = 0; i < sz; ++i) flat[i * (std::size_t)rw + (std::size_t)rw - 1] &= top;
    }
    auto dr = detail::DeviceBatch::upload(flat, sz, rw);
    detail::wipe(flat.data(), flat.size() * sizeof(uint64_t));
    pgpu_batch* c = nullptr;
    IPCL_GPU_CHECK(pgpu_batch_encrypt(dev->h, dm->h, dr->h, m_randbits, &c), "encrypt");
    return CipherText(*this, detail::DeviceBatch::adopt(c));
  }
  // injected randomness (setRandom) is read in place, without a copy of the batch
  std::vector<BigNumber> drawn;
  if (!m_testv) drawn = drawRandom(sz);
  static const std::vector<BigNumber> kNone;
  const std::vector<BigNumber>* rp = m_testv ? (m_r ? m_r.get() : &kNone) : &drawn;
  ERROR_CHECK(rp->size() == sz, "ippMBModExp: input vector size error");  // reference mod_exp.cpp:452-454
  std::vector<BigNumber> reduced;   // non-DJN bases wider than n^2, reduced copies
  const int nw = detail::words_for_bits(m_n->BitSize());
  // injected randomness that has been through here before: its device copy feeds this encrypt as well
  std::shared_ptr<InjectedRandom> cached = m_testv ? std::atomic_load(&m_r_dev) : nullptr;
  if (cached && cached->from == m_r && cached->dev->count == sz && pgpu_batch_is_current(cached->dev->h)) {   // (a pool restart drops it: re-upload below)
    auto dev = device();
    std::shared_ptr<detail::DeviceBatch> dm = pt.operandBatch(2 * nw, m_n.get());
    pgpu_batch* c = nullptr;
    IPCL_GPU_CHECK(pgpu_batch_encrypt(dev->h, dm->h, cached->dev->h, cached->bits, &c), "encrypt");
    return CipherText(*this, detail::DeviceBatch::adopt(c));
  }
  {
    bool neg = false, wide = false;
    for (const auto& x : *rp) {
      neg = neg || x.isNegative();
      wide = wide || (!m_enable_DJN && x.BitSize() > 64 * 2 * nw);
    }
    ERROR_CHECK(!neg, "encrypt: negative random value");
    if (wide) {
      reduced = *rp;
      for (auto& x : reduced)
        if (x.BitSize() > 64 * 2 * nw) x = x % *m_nsquare;  // base wider than n^2
      rp = &reduced;
    }
  }
  const std::vector<BigNumber>& r = *rp;
  auto dev = device();
  // (n*m+1) % n^2 only depends on m mod n: reduce plaintexts that are negative or wider than n^2
  std::shared_ptr<detail::DeviceBatch> dm = pt.operandBatch(2 * nw, m_n.get());
  const int rbits = detail::max_bits(r);
  const int rw = detail::words_for_bits(rbits);
  auto dr = detail::DeviceBatch::upload_values(r, rw);
  if (m_testv && rp == m_r.get()) {   // (not the reduced copies: those are rebuilt per call)
    auto keep = std::make_shared<InjectedRandom>();
    keep->from = m_r;
    keep->dev = dr;
    keep->bits = rbits;
    std::atomic_store(&m_r_dev, keep);
  }
  pgpu_batch* c = nullptr;
  IPCL_GPU_CHECK(pgpu_batch_encrypt(dev->h, dm->h, dr->h, rbits, &c), "encrypt");
  auto dc = detail::DeviceBatch::adopt(c);
  return CipherText(*this, dc);
}

}  // namespace ipcl

// BaseText / PlainText containers (reference ipcl/base_text.cpp, ipcl/plaintext.cpp).
#include "ipcl/base_text.hpp"

#include <algorithm>
#include <cstdint>
#include <mutex>

#include "detail.hpp"
#include "ipcl/ciphertext.hpp"
#include "ipcl/plaintext.hpp"
#include "ipcl/utils/util.hpp"

namespace ipcl {

// ---- device-resident values ----
namespace {
// Lazy host / device copies are created inside const methods, under a lock picked by the object's address out of a small
// table.  (Round 3 had ONE lock for all texts: a thread materialising its results -- which waits for its kernels -- held
// up every other thread's accessor for that long; four API threads ran no faster than one.)
constexpr std::size_t kTextLocks = 64;
std::mutex g_text_mu[kTextLocks];
std::mutex& text_mu(const void* obj) {
  const std::uintptr_t a = reinterpret_cast<std::uintptr_t>(obj);
  return g_text_mu[((a >> 4) ^ (a >> 12)) % kTextLocks];
}
// both objects of an assignment (one lock when they share a slot; address order otherwise)
struct PairLock {
  std::mutex *a, *b;
  PairLock(const void* x, const void* y) : a(&text_mu(x)), b(&text_mu(y)) {
    if (a == b) b = nullptr;
    else if (b < a) std::swap(a, b);
    a->lock();
    if (b) b->lock();
  }
  ~PairLock() {
    if (b) b->unlock();
    a->unlock();
  }
};
}

BaseText::BaseText(std::shared_ptr<detail::DeviceBatch> dev)
    : m_size(dev->count), m_dev(std::move(dev)), m_host_valid(false) {}

// a text may be copied while another thread materialises its host values (const accessors download lazily)
BaseText::BaseText(const BaseText& o) {
  std::lock_guard<std::mutex> lk(text_mu(&o));
  m_texts = detail::copy_texts(o.m_texts);
  m_size = o.m_size;
  m_dev = o.m_dev;
  m_host_valid = (bool)o.m_host_valid;
  m_bits_hint = o.m_bits_hint;
}

BaseText& BaseText::operator=(const BaseText& o) {
  if (this == &o) return *this;
  PairLock lk(this, &o);
  m_texts = detail::copy_texts(o.m_texts);
  m_size = o.m_size;
  m_dev = o.m_dev;
  m_host_valid = (bool)o.m_host_valid;
  m_bits_hint = o.m_bits_hint;
  return *this;
}

bool BaseText::adoptValues(const std::vector<BigNumber>& v) {
  if (v.size() < 64 || !pgpu_is_initialized()) return false;   // (no GPU context yet: containers work without one)
  std::size_t limbs = 0;
  int bits = 0;
  for (const BigNumber& x : v) {
    if (x.isNegative()) return false;        // signs only live in BigNumbers
    if (x.limbs64().size() > limbs) limbs = x.limbs64().size();
  }
  if (limbs == 0) limbs = 1;
  if (v.size() * limbs * 8 < detail::kEagerUploadBytes) return false;
  for (const BigNumber& x : v)
    if (x.limbs64().size() == limbs) bits = std::max(bits, x.BitSize());
  try {
    m_dev = detail::DeviceBatch::upload_values(v, (int)limbs);
  } catch (const std::exception&) {
    m_dev.reset();
    return false;
  }
  m_bits_hint = bits;
  m_host_valid = false;
  return true;
}

void BaseText::ensureHost() const {
  if (m_host_valid) return;
  std::shared_ptr<detail::DeviceBatch> dev;
  {
    std::lock_guard<std::mutex> lk(text_mu(this));
    if (m_host_valid) return;
    dev = m_dev;
  }
  // The download waits for the batch's kernels -- milliseconds -- so it runs OUTSIDE the lock: other objects share the slot
  // (thread stacks lay the same variables out alike), and a thread that downloads in a loop holds the slot nearly all the
  // time.  Round 6, four threads x CipherText * PlainText + getElement: in every second run one thread's operator* (which
  // takes its operands' slots for an instant) starved behind another thread's downloads for 60-350 ms at a time, 1.45 against
  // 2.0-2.7 ms per call (profiles/r06_place_pad.txt).  Two threads materialising the SAME text both download; one copy stays.
  std::vector<BigNumber> texts = dev->download();
  std::lock_guard<std::mutex> lk(text_mu(this));
  if (m_host_valid) return;
  m_texts = std::move(texts);
  m_host_valid = true;
}

void BaseText::invalidateDevice() {
  ensureHost();
  m_dev.reset();
  m_bits_hint = -1;
}

int BaseText::maxBitsHint() const {
  if (!m_host_valid) return m_bits_hint >= 0 ? m_bits_hint : 64 * m_dev->words;
  return detail::max_bits(m_texts);
}

// A device copy made under a pool that has been shut down since (terminateContext, possibly followed by a new context) is
// no operand any more: the values come back to the host -- from the pinned block the upload read, DeviceBatch::download --
// and the copy is dropped, so that the next use uploads to the pool of today.  (The reference's containers own their
// BigNumbers and keep working across a context restart: base_text.cpp:10-40.)
void BaseText::dropStaleDevice() const {
  std::lock_guard<std::mutex> lk(text_mu(this));
  if (!m_dev || pgpu_batch_is_current(m_dev->h)) return;
  if (!m_host_valid) {
    m_texts = m_dev->download();
    m_host_valid = true;
  }
  m_dev.reset();
  m_bits_hint = -1;
}

std::shared_ptr<detail::DeviceBatch> BaseText::deviceBatch(int words, const BigNumber* reduce_mod) const {
  dropStaleDevice();
  {
    std::lock_guard<std::mutex> lk(text_mu(this));
    if (m_dev && m_dev->words == words) return m_dev;
  }
  ensureHost();
  if (detail::all_fit(m_texts, words)) {
    auto b = detail::DeviceBatch::upload(detail::pack(m_texts, words), m_size, words);
    std::lock_guard<std::mutex> lk(text_mu(this));
    m_dev = b;   // the device copy mirrors m_texts exactly: cache it
    return b;
  }
  ERROR_CHECK(reduce_mod != nullptr, "BaseText: value does not fit the device batch width");
  // a reduced value (negative: the non-negative residue, as IPP's mod gives the reference) may be as wide as the
  // modulus, whatever width the caller derived from the magnitudes: widen the batch rather than fail in pack()
  const int rw = std::max(words, detail::words_for_bits(reduce_mod->BitSize()));
  std::vector<BigNumber> red = detail::copy_texts(m_texts);
  const BigNumber& mod = *reduce_mod;
  detail::parallel_for(red.size(), 64, [&](std::size_t i) {
    BigNumber& x = red[i];
    if (x.isNegative() || x.limbs64().size() > (size_t)words) x = x % mod;
  });
  return detail::DeviceBatch::upload(detail::pack(red, rw), m_size, rw);  // not cached
}

std::shared_ptr<detail::DeviceBatch> BaseText::operandBatch(int max_words, const BigNumber* reduce_mod) const {
  dropStaleDevice();
  if (!m_host_valid) {
    std::lock_guard<std::mutex> lk(text_mu(this));
    if (m_dev && m_dev->words <= max_words) return m_dev;
  }
  // (a text adopted onto the device at construction keeps the width of its widest value: wider than the operation takes,
  // it comes back to the host once and is reduced like any other over-wide input)
  return deviceBatch(std::min(max_words, detail::words_for_bits(maxBitsHint())), reduce_mod);
}

BaseText::BaseText(const uint32_t& n) : m_texts(1, BigNumber((Ipp32u)n)), m_size(1) {}

BaseText::BaseText(const std::vector<uint32_t>& n_v) {
  m_texts.reserve(n_v.size());
  for (uint32_t n : n_v) m_texts.emplace_back((Ipp32u)n);
  m_size = m_texts.size();
}

BaseText::BaseText(const BigNumber& bn) : m_texts(1, bn), m_size(1) {}

BaseText::BaseText(const std::vector<BigNumber>& bn_v) : m_size(bn_v.size()) {
  if (!adoptValues(bn_v)) m_texts = detail::copy_texts(bn_v);
}

BigNumber& BaseText::operator[](const std::size_t idx) {
  ERROR_CHECK(idx < m_size, "BaseText:operator[] index is out of range");
  invalidateDevice();   // the caller may write through the reference
  return m_texts[idx];
}

void BaseText::insert(const std::size_t pos, BigNumber& bn) {
  ERROR_CHECK(pos <= m_size, "BaseText: insert position is out of range");
  invalidateDevice();
  m_texts.insert(m_texts.begin() + (std::ptrdiff_t)pos, bn);
  m_size++;
}

void BaseText::clear() {
  m_host_valid = true;
  m_bits_hint = -1;
  m_dev.reset();
  m_texts.clear();
  m_size = 0;
}

// strict '<' as in the reference (base_text.cpp:58): the last element cannot be removed this way
void BaseText::remove(const std::size_t pos, const std::size_t length) {
  ERROR_CHECK(pos + length < m_size, "BaseText: remove position is out of range");
  invalidateDevice();
  m_texts.erase(m_texts.begin() + (std::ptrdiff_t)pos, m_texts.begin() + (std::ptrdiff_t)(pos + length));
  m_size -= length;
}

BigNumber BaseText::getElement(const std::size_t& idx) const {
  ERROR_CHECK(idx < m_size, "BaseText: getElement index is out of range");
  ensureHost();
  return m_texts[idx];
}

std::vector<uint32_t> BaseText::getElementVec(const std::size_t& idx) const {
  ERROR_CHECK(idx < m_size, "BaseText: getElementVec index is out of range");
  ensureHost();
  std::vector<uint32_t> v;
  m_texts[idx].num2vec(v);
  return v;
}

std::string BaseText::getElementHex(const std::size_t& idx) const {
  ERROR_CHECK(idx < m_size, "BaseText: getElementHex index is out of range");
  ensureHost();
  std::string s;
  m_texts[idx].num2hex(s);
  return s;
}

std::vector<BigNumber> BaseText::getChunk(const std::size_t& start, const std::size_t& size) const {
  ERROR_CHECK(start + size <= m_size, "BaseText: getChunk parameter is incorrect");
  ensureHost();
  return std::vector<BigNumber>(m_texts.begin() + (std::ptrdiff_t)start,
                                m_texts.begin() + (std::ptrdiff_t)(start + size));
}

// ---- serialization (reference base_text.hpp:108-114, plaintext.hpp:92-98) ----
void BaseText::save(serializer::OutputArchive& ar) const {
  ensureHost();
  ar.class_version("ipcl::BaseText");
  ar.u64((uint64_t)m_size);
  ar.u64((uint64_t)m_texts.size());
  for (const auto& t : m_texts) t.save(ar);
}

void BaseText::load(serializer::InputArchive& ar) {
  (void)ar.class_version("ipcl::BaseText");
  uint64_t size = ar.u64();
  uint64_t count = ar.u64();
  ERROR_CHECK(size == count && count < (1u << 28), "BaseText: corrupt archive");
  m_texts.assign((size_t)count, BigNumber());
  for (auto& t : m_texts) t.load(ar);
  m_size = (size_t)size;
  m_host_valid = true;
  m_bits_hint = -1;
  m_dev.reset();
}

void PlainText::save(serializer::OutputArchive& ar) const {
  ar.class_version("ipcl::PlainText");
  BaseText::save(ar);
}

void PlainText::load(serializer::InputArchive& ar) {
  (void)ar.class_version("ipcl::PlainText");
  BaseText::load(ar);
}

std::vector<BigNumber> BaseText::getTexts() const& {
  ensureHost();
  return detail::copy_texts(m_texts);
}
// on a temporary (`pk.encrypt(pt).getTexts()`): hand the values over instead of copying them a second time
std::vector<BigNumber> BaseText::getTexts() && {
  ensureHost();
  std::lock_guard<std::mutex> lk(text_mu(this));
  std::vector<BigNumber> out = std::move(m_texts);
  m_texts.clear();
  if (m_dev) m_host_valid = false;   // the device copy is still there: a later accessor downloads again
  else m_size = 0;
  return out;
}
std::size_t BaseText::getSize() const { return m_size; }

// ---- PlainText ----
PlainText::PlainText(const uint32_t& n) : BaseText(n) {}
PlainText::PlainText(const std::vector<uint32_t>& n_v) : BaseText(n_v) {}
PlainText::PlainText(const BigNumber& bn) : BaseText(bn) {}
PlainText::PlainText(const std::vector<BigNumber>& bn_v) : BaseText(bn_v) {}
PlainText::PlainText(std::shared_ptr<detail::DeviceBatch> dev) : BaseText(std::move(dev)) {}

CipherText PlainText::operator+(const CipherText& other) const { return other + *this; }
CipherText PlainText::operator*(const CipherText& other) const { return other * *this; }

PlainText::operator std::vector<uint32_t>() const {
  ERROR_CHECK(m_size > 0, "PlainText: type conversion to uint32_t vector error");
  ensureHost();
  std::vector<uint32_t> v;
  m_texts[0].num2vec(v);
  return v;
}

PlainText::operator BigNumber() const {
  ERROR_CHECK(m_size > 0, "PlainText: type conversion to BigNumber error");
  ensureHost();
  return m_texts[0];
}

PlainText::operator std::vector<BigNumber>() const {
  ERROR_CHECK(m_size > 0, "PlainText: type conversion to BigNumber vector error");
  ensureHost();
  return m_texts;
}

namespace detail {
// shared by PlainText::rotate and CipherText::rotate (plaintext.cpp:57-73, ciphertext.cpp:117-133):
// positive shift moves elements towards higher indices
std::vector<BigNumber> rotated(const std::vector<BigNumber>& v, int shift) {
  const int size = (int)v.size();
  ERROR_CHECK(size != 1, "rotate: Cannot rotate single CipherText");
  ERROR_CHECK(shift >= -size && shift <= size, "rotate: Cannot shift more than the test size");
  std::vector<BigNumber> out(v);
  if (size == 0 || shift == 0 || shift == size || shift == -size) return out;
  int left = shift > 0 ? size - shift : -shift;
  std::rotate(out.begin(), out.begin() + left, out.end());
  return out;
}
}  // namespace detail

PlainText PlainText::rotate(int shift) const {
  ensureHost();
  return PlainText(detail::rotated(m_texts, shift));
}

}  // namespace ipcl

// pailliercryptolib_amd -- BaseText container (reference ipcl/include/ipcl/base_text.hpp:14-115).
#ifndef PAILLIERCRYPTOLIB_AMD_IPCL_BASE_TEXT_HPP_
#define PAILLIERCRYPTOLIB_AMD_IPCL_BASE_TEXT_HPP_

#include <atomic>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "ipcl/bignum.h"

namespace ipcl {

namespace detail {
struct DeviceBatch;
// std::atomic<bool> with value-copy semantics: "the host copy exists" is published by one thread (lazy
// download) while others may be reading or copying the text
struct AtomicFlag {
  std::atomic<bool> v;
  AtomicFlag(bool b = false) : v(b) {}
  AtomicFlag(const AtomicFlag& o) : v(o.v.load(std::memory_order_acquire)) {}
  AtomicFlag& operator=(const AtomicFlag& o) {
    v.store(o.v.load(std::memory_order_acquire), std::memory_order_release);
    return *this;
  }
  AtomicFlag& operator=(bool b) {
    v.store(b, std::memory_order_release);
    return *this;
  }
  operator bool() const { return v.load(std::memory_order_acquire); }
};
}  // namespace detail

// Container of BigNumbers (reference base_text.hpp:14-115) with one addition: the values may live
// in GPU memory as a flat limb batch (the result of encrypt / decrypt / CT+CT / CT*PT) and are
// only copied back into host BigNumbers when an accessor needs them.  Chained homomorphic
// operations therefore never round-trip through std::vector<BigNumber>; the
// observable API is unchanged.
class BaseText {
 public:
  BaseText() = default;
  ~BaseText() = default;
  explicit BaseText(const uint32_t& n);
  explicit BaseText(const std::vector<uint32_t>& n_v);
  explicit BaseText(const BigNumber& bn);
  explicit BaseText(const std::vector<BigNumber>& bn_v);
  BaseText(const BaseText& bt);               // (copies take the materialisation lock: see base_text.cpp)
  BaseText& operator=(const BaseText& other);

  BigNumber& operator[](const std::size_t idx);
  void insert(const std::size_t pos, BigNumber& bn);
  void clear();
  void remove(const std::size_t pos, const std::size_t length = 1);
  BigNumber getElement(const std::size_t& idx) const;
  std::vector<uint32_t> getElementVec(const std::size_t& idx) const;
  std::string getElementHex(const std::size_t& idx) const;
  std::vector<BigNumber> getChunk(const std::size_t& start, const std::size_t& size) const;
  // (reference base_text.hpp: "std::vector<BigNumber> getTexts() const".  On a temporary -- the usual
  // `pk.encrypt(pt).getTexts()` -- the values are moved out instead of copied a second time.)
  std::vector<BigNumber> getTexts() const&;
  std::vector<BigNumber> getTexts() &&;
  std::size_t getSize() const;

  // reference base_text.hpp:108-114: "size", "texts"
  void save(serializer::OutputArchive& ar) const;
  void load(serializer::InputArchive& ar);

  // true while the values exist only in GPU memory (no host BigNumbers materialised yet)
  bool isDeviceResident() const { return !m_host_valid; }

 protected:
  mutable std::vector<BigNumber> m_texts;   // host values; valid iff m_host_valid
  std::size_t m_size = 0;
  mutable std::shared_ptr<detail::DeviceBatch> m_dev;  // immutable device copy (may be the only copy)
  mutable detail::AtomicFlag m_host_valid{true};
  mutable int m_bits_hint = -1; // exact bit length of the widest value when the text was built around host values that
                                // went straight to the GPU (adoptValues); -1: unknown

  // construct around a device batch (no host copy yet)
  explicit BaseText(std::shared_ptr<detail::DeviceBatch> dev);
  // Large batches of non-negative host values go straight into a pinned block and on to the GPU (ONE pass over the
  // values instead of a BigNumber copy now and a pack + staged upload at first use); the BigNumbers are re-created
  // from the device copy if an accessor ever asks for them.  False: not applicable, the caller copies the values.
  bool adoptValues(const std::vector<BigNumber>& v);
  void ensureHost() const;      // download + unpack if the host copy is missing
  void dropStaleDevice() const; // a device copy of a pool that has been shut down: values back to the host, copy dropped
  void invalidateDevice();      // before any mutation of m_texts
  // device copy with exactly `words` 64-bit limbs per element (uploaded and cached on demand);
  // values that are negative / too wide are reduced mod *reduce_mod (else: error)
  std::shared_ptr<detail::DeviceBatch> deviceBatch(int words, const BigNumber* reduce_mod = nullptr) const;
  // the values as an operand no wider than max_words: the resident copy when it fits (no host round trip), otherwise a
  // batch of the values reduced modulo *reduce_mod (over-wide plaintexts: (n*m + 1) % n^2 only depends on m mod n)
  std::shared_ptr<detail::DeviceBatch> operandBatch(int max_words, const BigNumber* reduce_mod) const;
  int maxBitsHint() const;      // exact for host values, 64*words for device-only values

  friend class PublicKey;
  friend class PrivateKey;
  friend class CipherText;
};

}  // namespace ipcl
#endif  // PAILLIERCRYPTOLIB_AMD_IPCL_BASE_TEXT_HPP_

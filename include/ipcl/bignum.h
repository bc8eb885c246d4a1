// pailliercryptolib_amd -- host-side arbitrary-precision integer.
//
// IPP-free re-creation of the public surface of the reference's global ::BigNumber
// (reference: ipcl/include/ipcl/bignum.h:28-161, ipcl/bignum.cpp).  The reference wraps an
// opaque IppsBigNumState (sign + little-endian 32-bit words); this one is a plain value type
// over little-endian 64-bit limbs -- the same limb layout the GPU C-ABI (include/pgpu.h) uses,
// so a BigNumber's magnitude can be memcpy'd straight into a batch buffer.
//
// Behaviours the reference's tests pin and that are reproduced here:
//   Q1  operator% returns the NON-NEGATIVE residue for a negative left operand;
//   Q7  num2hex is lowercase, "0x"-prefixed, no leading zeros, and zero prints as "0x";
//       the string ctor understands lowercase hex ("0x...") and decimal;
//   Q8  num2vec returns at least one 32-bit word, also for the value 0.
#ifndef PAILLIERCRYPTOLIB_AMD_IPCL_BIGNUM_H_
#define PAILLIERCRYPTOLIB_AMD_IPCL_BIGNUM_H_

#include <cstddef>
#include <cstdint>
#include <ostream>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "ipcl/utils/serialize.hpp"

namespace ipcl {
namespace detail {
// Limb storage of BigNumber.  An API call creates and destroys vectors of thousands of BigNumbers (8192 plaintexts in,
// 8192 ciphertexts out): with one heap block per value, malloc / free are the bulk of the host time of such a call
// (profiles/r03_api_stage_probe.txt).  A thread may therefore open a BULK SCOPE: while it is open, the limb blocks this
// thread allocates are carved out of one arena with a bump pointer, and freed by a counter -- the arena returns to the
// heap when its last block has died (so one surviving BigNumber of a batch keeps that batch's arena alive: a few MB).
// Up to four retired arenas (64 KB - 64 MB each) are kept for the next scope instead of going back to the heap: their
// pages are already mapped; limb_cache_trim() (terminateContext calls it) frees them.
// Outside a scope, and when an arena is full, blocks come from malloc as before.  Every block carries a 16-byte header
// that says where it came from, so any thread may free any block.
void* limb_alloc(std::size_t bytes);
void limb_free(void* p) noexcept;
void limb_bulk_begin(std::size_t bytes_hint);
void limb_bulk_end() noexcept;
void limb_cache_trim() noexcept;
struct LimbBulkScope {
  explicit LimbBulkScope(std::size_t bytes_hint) { limb_bulk_begin(bytes_hint); }
  ~LimbBulkScope() { limb_bulk_end(); }
  LimbBulkScope(const LimbBulkScope&) = delete;
  LimbBulkScope& operator=(const LimbBulkScope&) = delete;
};
// The limb container of BigNumber: the subset of std::vector<uint64_t> the arithmetic uses, over limb_alloc / limb_free, plus
// ONE thing a vector cannot do: take over limbs that already lie in a block of the limb allocator (adopt).  Results of
// a GPU call arrive by DMA in a pinned block laid out as such blocks (16-byte header, row, header, row, ...); their
// BigNumbers point into it instead of copying 2-4 MB out of it (csrc/host/common.cpp: DeviceBatch::download).
class LimbVec {
 public:
  using value_type = uint64_t;
  using size_type = std::size_t;
  using iterator = uint64_t*;
  using const_iterator = const uint64_t*;
  LimbVec() noexcept = default;
  explicit LimbVec(std::size_t n, uint64_t v = 0) { grow_to(n); for (std::size_t i = 0; i < n; ++i) p_[i] = v; n_ = n; }
  LimbVec(const LimbVec& o) { if (o.n_) { grow_to(o.n_); std::memcpy(p_, o.p_, o.n_ * sizeof(uint64_t)); n_ = o.n_; } }
  LimbVec(LimbVec&& o) noexcept : p_(o.p_), n_(o.n_), cap_(o.cap_) { o.p_ = nullptr; o.n_ = o.cap_ = 0; }
  LimbVec& operator=(const LimbVec& o) {
    if (this != &o) {
      if (o.n_ > cap_) { release(); grow_to(o.n_); }
      if (o.n_) std::memcpy(p_, o.p_, o.n_ * sizeof(uint64_t));
      n_ = o.n_;
    }
    return *this;
  }
  LimbVec& operator=(LimbVec&& o) noexcept {
    if (this != &o) {
      release();
      p_ = o.p_; n_ = o.n_; cap_ = o.cap_;
      o.p_ = nullptr; o.n_ = o.cap_ = 0;
    }
    return *this;
  }
  ~LimbVec() { release(); }
  // limbs[0 .. n) lie in a limb-allocator block of room for cap limbs (header in front, see limb_block_adopt)
  static LimbVec adopt(uint64_t* limbs, std::size_t n, std::size_t cap) noexcept {
    LimbVec v;
    v.p_ = limbs; v.n_ = n; v.cap_ = cap;
    return v;
  }
  std::size_t size() const noexcept { return n_; }
  std::size_t capacity() const noexcept { return cap_; }
  bool empty() const noexcept { return n_ == 0; }
  uint64_t* data() noexcept { return p_; }
  const uint64_t* data() const noexcept { return p_; }
  uint64_t& operator[](std::size_t i) noexcept { return p_[i]; }
  const uint64_t& operator[](std::size_t i) const noexcept { return p_[i]; }
  uint64_t& back() noexcept { return p_[n_ - 1]; }
  const uint64_t& back() const noexcept { return p_[n_ - 1]; }
  iterator begin() noexcept { return p_; }
  iterator end() noexcept { return p_ + n_; }
  const_iterator begin() const noexcept { return p_; }
  const_iterator end() const noexcept { return p_ + n_; }
  void clear() noexcept { n_ = 0; }
  void pop_back() noexcept { --n_; }
  void push_back(uint64_t v) { if (n_ == cap_) reserve(cap_ ? 2 * cap_ : 4); p_[n_++] = v; }
  void reserve(std::size_t n) {
    if (n <= cap_) return;
    uint64_t* q = static_cast<uint64_t*>(limb_alloc(n * sizeof(uint64_t)));
    if (n_) std::memcpy(q, p_, n_ * sizeof(uint64_t));
    limb_free(p_);
    p_ = q; cap_ = n;
  }
  void resize(std::size_t n) {            // (new limbs are zero)
    reserve(n);
    for (std::size_t i = n_; i < n; ++i) p_[i] = 0;
    n_ = n;
  }
  void assign(std::size_t n, uint64_t v) {
    if (n > cap_) { release(); grow_to(n); }
    for (std::size_t i = 0; i < n; ++i) p_[i] = v;
    n_ = n;
  }
  template <class It>
  void assign(It first, It last) {
    const std::size_t n = (std::size_t)(last - first);
    if (n > cap_) { release(); grow_to(n); }
    for (std::size_t i = 0; i < n; ++i) p_[i] = (uint64_t)first[i];
    n_ = n;
  }
  void swap(LimbVec& o) noexcept { std::swap(p_, o.p_); std::swap(n_, o.n_); std::swap(cap_, o.cap_); }
  // (ordering and equality as std::vector's: lexicographic over the limbs -- a map key in mod_exp.cpp)
  friend bool operator==(const LimbVec& a, const LimbVec& b) noexcept {
    return a.n_ == b.n_ && (a.n_ == 0 || std::memcmp(a.p_, b.p_, a.n_ * sizeof(uint64_t)) == 0);
  }
  friend bool operator!=(const LimbVec& a, const LimbVec& b) noexcept { return !(a == b); }
  friend bool operator<(const LimbVec& a, const LimbVec& b) noexcept {
    const std::size_t n = a.n_ < b.n_ ? a.n_ : b.n_;
    for (std::size_t i = 0; i < n; ++i)
      if (a.p_[i] != b.p_[i]) return a.p_[i] < b.p_[i];
    return a.n_ < b.n_;
  }

 private:
  uint64_t* p_ = nullptr;
  std::size_t n_ = 0, cap_ = 0;
  void grow_to(std::size_t n) { p_ = static_cast<uint64_t*>(limb_alloc(n * sizeof(uint64_t))); cap_ = n; }   // (p_ is null)
  void release() noexcept { limb_free(p_); p_ = nullptr; n_ = cap_ = 0; }
};

// Blocks that live in caller-provided memory (a pinned result block): `ctrl` (64 bytes, 16-aligned) becomes the control
// structure of an arena that owns no memory itself; limb_block_adopt stamps the 16 bytes in front of `limbs` as the header
// of a block of that arena; when the last block has been freed and the opener has closed, release(cookie) runs (once, on
// whichever thread frees last).  limb_free / LimbVec treat such blocks like any other.
struct LimbArenaExt;
LimbArenaExt* limb_arena_open(void* ctrl, void (*release)(void*), void* cookie) noexcept;
void limb_block_adopt(LimbArenaExt* a, uint64_t* limbs) noexcept;
void limb_blocks_adopt(LimbArenaExt* a, uint64_t* first, std::size_t stride_limbs, std::size_t count) noexcept;   // rows of a batch
void limb_arena_close(LimbArenaExt* a) noexcept;
}  // namespace detail
}  // namespace ipcl

typedef uint8_t Ipp8u;
typedef uint32_t Ipp32u;
typedef int32_t Ipp32s;
typedef uint64_t Ipp64u;
// Same enumerator values as ippcp's IppsBigNumSGN so client code that spells them compiles.
typedef enum { IppsBigNumNEG = 0, IppsBigNumPOS = 1 } IppsBigNumSGN;

class BigNumber {
 public:
  using Limbs = ipcl::detail::LimbVec;
  BigNumber(Ipp32u value = 0);
  BigNumber(Ipp32s value);
  BigNumber(const Ipp32u* pData, int length = 1, IppsBigNumSGN sgn = IppsBigNumPOS);
  BigNumber(const BigNumber& bn) = default;
  BigNumber(BigNumber&& bn) = default;
  BigNumber(const char* s);
  virtual ~BigNumber() = default;

  // set value from little-endian 32-bit words (reference: bignum.cpp:109-112)
  void Set(const Ipp32u* pData, int length = 1, IppsBigNumSGN sgn = IppsBigNumPOS);

  static const BigNumber& Zero();
  static const BigNumber& One();
  static const BigNumber& Two();

  BigNumber& operator=(const BigNumber& bn) = default;
  BigNumber& operator=(BigNumber&& bn) = default;
  BigNumber& operator+=(Ipp32u n);
  BigNumber& operator+=(const BigNumber& bn);
  BigNumber& operator-=(Ipp32u n);
  BigNumber& operator-=(const BigNumber& bn);
  BigNumber& operator*=(Ipp32u n);
  BigNumber& operator*=(const BigNumber& bn);
  BigNumber& operator/=(Ipp32u n);
  BigNumber& operator/=(const BigNumber& bn);
  BigNumber& operator%=(Ipp32u n);
  BigNumber& operator%=(const BigNumber& bn);
  friend BigNumber operator+(const BigNumber& a, const BigNumber& b);
  friend BigNumber operator+(const BigNumber& a, Ipp32u);
  friend BigNumber operator-(const BigNumber& a, const BigNumber& b);
  friend BigNumber operator-(const BigNumber& a, Ipp32u);
  friend BigNumber operator*(const BigNumber& a, const BigNumber& b);
  friend BigNumber operator*(const BigNumber& a, Ipp32u);
  friend BigNumber operator%(const BigNumber& a, const BigNumber& b);
  friend BigNumber operator%(const BigNumber& a, Ipp32u);
  friend BigNumber operator/(const BigNumber& a, const BigNumber& b);
  friend BigNumber operator/(const BigNumber& a, Ipp32u);

  // modulo arithmetic, *this is the modulus (reference: bignum.cpp:318-350)
  BigNumber Modulo(const BigNumber& a) const;
  BigNumber ModAdd(const BigNumber& a, const BigNumber& b) const;
  BigNumber ModSub(const BigNumber& a, const BigNumber& b) const;
  BigNumber ModMul(const BigNumber& a, const BigNumber& b) const;
  BigNumber InverseAdd(const BigNumber& a) const;
  BigNumber InverseMul(const BigNumber& a) const;
  BigNumber gcd(const BigNumber& q) const;
  int compare(const BigNumber&) const;

  friend bool operator<(const BigNumber& a, const BigNumber& b) { return a.compare(b) < 0; }
  friend bool operator>(const BigNumber& a, const BigNumber& b) { return a.compare(b) > 0; }
  friend bool operator==(const BigNumber& a, const BigNumber& b) { return a.compare(b) == 0; }
  friend bool operator!=(const BigNumber& a, const BigNumber& b) { return a.compare(b) != 0; }
  friend bool operator<=(const BigNumber& a, const BigNumber& b) { return !(a > b); }
  friend bool operator>=(const BigNumber& a, const BigNumber& b) { return !(a < b); }

  bool IsOdd() const;
  bool IsEven() const { return !IsOdd(); }
  bool TestBit(int index) const;

  int MSB() const;
  int LSB() const;
  int BitSize() const { return MSB() + 1; }
  int DwordSize() const { return (BitSize() + 31) >> 5; }
  friend int Bit(const std::vector<Ipp32u>& v, int n);

  void num2hex(std::string& s) const;
  void num2vec(std::vector<Ipp32u>& v) const;
  friend std::ostream& operator<<(std::ostream& os, const BigNumber& a);
  void num2char(std::vector<Ipp8u>& dest) const;

  // big-endian fixed-length octet strings (the reference's QAT wire format, bignum.cpp:511-565)
  static bool fromBin(BigNumber& bn, const unsigned char* data, int len);
  static bool toBin(unsigned char* data, int len, const BigNumber& bn);
  static bool toBin(unsigned char** data, int* len, const BigNumber& bn);

  // ---- GPU-side layout helpers (north_star: "bignum.cpp gains a GPU-side 64-bit-limb layout") ----
  // Little-endian 64-bit limbs of |*this|, zero-padded/truncation-checked to exactly nlimbs.
  // Returns false if the magnitude does not fit.
  bool toLimbs64(uint64_t* out, std::size_t nlimbs) const;
  static BigNumber fromLimbs64(const uint64_t* limbs, std::size_t nlimbs);
  // the same without a copy: limbs[0 .. nlimbs) lie in a block of the limb allocator with room for nlimbs
  // (ipcl::detail::limb_block_adopt has stamped its header); leading zero limbs are trimmed
  static BigNumber adoptLimbs64(uint64_t* limbs, std::size_t nlimbs);
  const Limbs& limbs64() const { return m_mag; }
  bool isNegative() const { return m_neg; }
  bool isZero() const { return m_mag.empty(); }

  // wire format of the reference's cereal save/load (bignum.h:131-153): class version once per
  // archive, the 32-bit words (num2vec), the sign enum
  void save(ipcl::serializer::OutputArchive& ar) const;
  void load(ipcl::serializer::InputArchive& ar);

  // divide with remainder: *this = q*d + r, q truncated toward zero, r has the sign of *this
  static void divmod(const BigNumber& a, const BigNumber& d, BigNumber* q, BigNumber* r);

 protected:
  Limbs m_mag;                  // magnitude, little-endian, no leading zero limbs; empty == 0
  bool m_neg = false;           // sign; never set for zero
  void trim();
};

constexpr int BITSIZE_WORD(int n) { return (((n) + 31) >> 5); }
constexpr int BITSIZE_DWORD(int n) { return (((n) + 63) >> 6); }

#endif  // PAILLIERCRYPTOLIB_AMD_IPCL_BIGNUM_H_

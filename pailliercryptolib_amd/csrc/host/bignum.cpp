// pailliercryptolib_amd -- host BigNumber implementation (see include/ipcl/bignum.h).
//
// Own design: sign-magnitude over 64-bit limbs, schoolbook multiply, Knuth algorithm D
// division, Euclid gcd and extended-Euclid modular inverse built on divmod.  It is the
// IPP-free counterpart of the reference's ipcl/bignum.cpp (which forwards every operation to
// ipps*_BN); only the observable behaviour is mirrored, cited per function.
#include "ipcl/bignum.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

typedef unsigned __int128 u128;

// ---- limb storage (bignum.h: limb_alloc / LimbVec / LimbBulkScope) ----
#include <atomic>
#include <new>
namespace ipcl {
namespace detail {
namespace {
struct LimbArena {
  std::atomic<std::size_t> live;   // blocks handed out and not yet freed, + 1 while the scope that owns the arena is open
  std::size_t cap, used;           // bytes behind the header
  std::size_t room;                // bytes the allocation really has behind the header (>= cap; a recycled arena keeps its size)
  void (*release)(void*);          // non-null: the arena lives in caller memory (limb_arena_open); called instead of free
  void* cookie;
};
static_assert(sizeof(LimbArena) <= 64, "limb_arena_open: the control structure fits the 64 bytes callers reserve");
struct LimbHeader {                // in front of every block
  LimbArena* arena;                // null: the block came from malloc
  std::size_t pad;
};
static_assert(sizeof(LimbHeader) == 16, "blocks stay 16-byte aligned behind their header");
constexpr std::size_t kArenaHead = (sizeof(LimbArena) + 15) & ~(std::size_t)15;
thread_local LimbArena* t_arena = nullptr;
thread_local int t_bulk_depth = 0;
// Retired arenas are kept for the next bulk scope: a fresh multi-megabyte allocation comes from mmap and pays a page
// fault per 4 KB on first touch -- for the 8192 plaintexts of a decrypt call that was 2/3 of the unpacking time.  A
// caller that drops one batch of results before (or while) it asks for the next gets the same pages back.
constexpr int kArenaKeep = 4;
constexpr std::size_t kArenaKeepMin = 64 << 10, kArenaKeepMax = 64 << 20;
std::atomic_flag g_keep_lock = ATOMIC_FLAG_INIT;
LimbArena* g_keep[kArenaKeep] = {};
struct KeepGuard {
  KeepGuard() noexcept { while (g_keep_lock.test_and_set(std::memory_order_acquire)) {} }
  ~KeepGuard() { g_keep_lock.clear(std::memory_order_release); }
};
void arena_release(LimbArena* a) noexcept {
  if (a->live.fetch_sub(1, std::memory_order_acq_rel) != 1) return;
  if (a->release) {                // caller memory: hand it back (the control structure dies with it)
    void (*fn)(void*) = a->release;
    void* ck = a->cookie;
    fn(ck);
    return;
  }
  if (a->room >= kArenaKeepMin && a->room <= kArenaKeepMax) {
    KeepGuard g;
    for (LimbArena*& slot : g_keep)
      if (!slot) { slot = a; return; }
  }
  std::free(a);
}
// a retired arena with room for `cap` bytes that is not more than four times too large
LimbArena* arena_reuse(std::size_t cap) noexcept {
  KeepGuard g;
  LimbArena** best = nullptr;
  for (LimbArena*& slot : g_keep)
    if (slot && slot->room >= cap && slot->room / 4 <= cap && (!best || slot->room < (*best)->room)) best = &slot;
  if (!best) return nullptr;
  LimbArena* a = *best;
  *best = nullptr;
  return a;
}
}  // namespace

void limb_cache_trim() noexcept {
  KeepGuard g;
  for (LimbArena*& slot : g_keep) {
    std::free(slot);
    slot = nullptr;
  }
}

void* limb_alloc(std::size_t bytes) {
  const std::size_t need = sizeof(LimbHeader) + ((bytes + 15) & ~(std::size_t)15);
  LimbArena* a = t_arena;
  if (a && a->used + need <= a->cap) {
    LimbHeader* h = reinterpret_cast<LimbHeader*>(reinterpret_cast<char*>(a) + kArenaHead + a->used);
    a->used += need;
    a->live.fetch_add(1, std::memory_order_relaxed);
    h->arena = a;
    return h + 1;
  }
  LimbHeader* h = static_cast<LimbHeader*>(std::malloc(need));
  if (!h) throw std::bad_alloc();
  h->arena = nullptr;
  return h + 1;
}

void limb_free(void* p) noexcept {
  if (!p) return;
  LimbHeader* h = static_cast<LimbHeader*>(p) - 1;
  if (h->arena) arena_release(h->arena);
  else std::free(h);
}

void limb_bulk_begin(std::size_t bytes_hint) {
  if (t_bulk_depth++ > 0) return;   // nested scopes share the outer arena
  if (bytes_hint < 4096) return;    // not worth an arena
  const std::size_t cap = (bytes_hint + 4095) & ~(std::size_t)4095;
  LimbArena* a = arena_reuse(cap);
  if (!a) {
    a = static_cast<LimbArena*>(std::malloc(kArenaHead + cap));
    if (!a) return;                 // no arena: blocks come from malloc
    a->room = cap;
  }
  a->release = nullptr;
  a->cookie = nullptr;
  new (&a->live) std::atomic<std::size_t>(1);
  a->cap = a->room;
  a->used = 0;
  t_arena = a;
}

struct LimbArenaExt : LimbArena {};

LimbArenaExt* limb_arena_open(void* ctrl, void (*release)(void*), void* cookie) noexcept {
  LimbArena* a = static_cast<LimbArena*>(ctrl);
  new (&a->live) std::atomic<std::size_t>(1);   // the opener's reference (limb_arena_close drops it)
  a->cap = a->used = a->room = 0;
  a->release = release;
  a->cookie = cookie;
  return static_cast<LimbArenaExt*>(a);
}
void limb_block_adopt(LimbArenaExt* a, uint64_t* limbs) noexcept {
  LimbHeader* h = reinterpret_cast<LimbHeader*>(limbs) - 1;
  h->arena = a;
  h->pad = 0;
  a->live.fetch_add(1, std::memory_order_relaxed);
}
void limb_blocks_adopt(LimbArenaExt* a, uint64_t* first, std::size_t stride_limbs, std::size_t count) noexcept {
  a->live.fetch_add(count, std::memory_order_relaxed);   // (one atomic for the batch; the headers are plain stores)
  for (std::size_t i = 0; i < count; ++i) {
    LimbHeader* h = reinterpret_cast<LimbHeader*>(first + i * stride_limbs) - 1;
    h->arena = a;
    h->pad = 0;
  }
}
void limb_arena_close(LimbArenaExt* a) noexcept { arena_release(a); }

void limb_bulk_end() noexcept {
  if (--t_bulk_depth > 0) return;
  t_bulk_depth = 0;
  if (LimbArena* a = t_arena) {
    t_arena = nullptr;
    arena_release(a);
  }
}
}  // namespace detail
}  // namespace ipcl

namespace {

typedef BigNumber::Limbs Mag;

void mag_trim(Mag& a) {
  while (!a.empty() && a.back() == 0) a.pop_back();
}

int mag_cmp(const Mag& a, const Mag& b) {
  if (a.size() != b.size()) return a.size() < b.size() ? -1 : 1;
  for (size_t i = a.size(); i-- > 0;)
    if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
  return 0;
}

Mag mag_add(const Mag& a, const Mag& b) {
  const Mag& x = a.size() >= b.size() ? a : b;
  const Mag& y = a.size() >= b.size() ? b : a;
  Mag r(x.size() + 1);
  uint64_t c = 0;
  for (size_t i = 0; i < x.size(); ++i) {
    u128 s = (u128)x[i] + (i < y.size() ? y[i] : 0) + c;
    r[i] = (uint64_t)s;
    c = (uint64_t)(s >> 64);
  }
  r[x.size()] = c;
  mag_trim(r);
  return r;
}

// a - b, requires a >= b
Mag mag_sub(const Mag& a, const Mag& b) {
  Mag r(a.size());
  uint64_t br = 0;
  for (size_t i = 0; i < a.size(); ++i) {
    uint64_t bi = i < b.size() ? b[i] : 0;
    u128 d = (u128)a[i] - bi - br;
    r[i] = (uint64_t)d;
    br = (uint64_t)(d >> 64) & 1;
  }
  mag_trim(r);
  return r;
}

Mag mag_mul(const Mag& a, const Mag& b) {
  if (a.empty() || b.empty()) return Mag();
  Mag r(a.size() + b.size(), 0);
  for (size_t i = 0; i < a.size(); ++i) {
    uint64_t c = 0;
    for (size_t j = 0; j < b.size(); ++j) {
      u128 t = (u128)a[i] * b[j] + r[i + j] + c;
      r[i + j] = (uint64_t)t;
      c = (uint64_t)(t >> 64);
    }
    r[i + b.size()] = c;
  }
  mag_trim(r);
  return r;
}

// Knuth TAOCP vol.2 4.3.1 algorithm D on 64-bit limbs. q = a / d, r = a % d (magnitudes).
void mag_divmod(const Mag& a, const Mag& d, Mag* q, Mag* r) {
  if (d.empty()) throw std::runtime_error("BigNumber: division by zero");
  if (mag_cmp(a, d) < 0) {
    if (q) q->clear();
    if (r) *r = a;
    return;
  }
  if (d.size() == 1) {
    uint64_t dv = d[0], rem = 0;
    Mag qq(a.size());
    for (size_t i = a.size(); i-- > 0;) {
      u128 cur = ((u128)rem << 64) | a[i];
      qq[i] = (uint64_t)(cur / dv);
      rem = (uint64_t)(cur % dv);
    }
    mag_trim(qq);
    if (q) *q = qq;
    if (r) {
      r->clear();
      if (rem) r->push_back(rem);
    }
    return;
  }
  const int s = __builtin_clzll(d.back());
  const size_t n = d.size(), m = a.size() - n;
  Mag v(n), u(a.size() + 1);
  for (size_t i = n; i-- > 0;)
    v[i] = s ? (d[i] << s) | (i ? d[i - 1] >> (64 - s) : 0) : d[i];
  u[a.size()] = s ? a.back() >> (64 - s) : 0;
  for (size_t i = a.size(); i-- > 0;)
    u[i] = s ? (a[i] << s) | (i ? a[i - 1] >> (64 - s) : 0) : a[i];
  Mag qq(m + 1, 0);
  for (size_t j = m + 1; j-- > 0;) {
    u128 num = ((u128)u[j + n] << 64) | u[j + n - 1];
    u128 qhat = num / v[n - 1], rhat = num % v[n - 1];
    while ((qhat >> 64) != 0 ||
           (u128)(uint64_t)qhat * v[n - 2] > ((rhat << 64) | u[j + n - 2])) {
      --qhat;
      rhat += v[n - 1];
      if ((rhat >> 64) != 0) break;
    }
    // multiply and subtract
    uint64_t borrow = 0, carry = 0;
    for (size_t i = 0; i < n; ++i) {
      u128 p = (u128)(uint64_t)qhat * v[i] + carry;
      carry = (uint64_t)(p >> 64);
      u128 sub = (u128)u[i + j] - (uint64_t)p - borrow;
      u[i + j] = (uint64_t)sub;
      borrow = (uint64_t)(sub >> 64) & 1;
    }
    u128 sub = (u128)u[j + n] - carry - borrow;
    u[j + n] = (uint64_t)sub;
    borrow = (uint64_t)(sub >> 64) & 1;
    qq[j] = (uint64_t)qhat;
    if (borrow) {  // add back
      --qq[j];
      uint64_t c = 0;
      for (size_t i = 0; i < n; ++i) {
        u128 t = (u128)u[i + j] + v[i] + c;
        u[i + j] = (uint64_t)t;
        c = (uint64_t)(t >> 64);
      }
      u[j + n] += c;
    }
  }
  mag_trim(qq);
  if (q) *q = qq;
  if (r) {
    Mag rr(n);
    for (size_t i = 0; i < n; ++i)
      rr[i] = s ? (u[i] >> s) | (u[i + 1] << (64 - s)) : u[i];
    mag_trim(rr);
    *r = rr;
  }
}

}  // namespace

void BigNumber::trim() {
  mag_trim(m_mag);
  if (m_mag.empty()) m_neg = false;
}

// ---- constructors (reference: bignum.cpp:44-103) ----
BigNumber::BigNumber(Ipp32u value) {
  if (value) m_mag.push_back(value);
}

BigNumber::BigNumber(Ipp32s value) {
  if (value) {
    m_mag.push_back((uint64_t)std::llabs((long long)value));
    m_neg = value < 0;
  }
}

BigNumber::BigNumber(const Ipp32u* pData, int length, IppsBigNumSGN sgn) {
  Set(pData, length, sgn);
}

void BigNumber::Set(const Ipp32u* pData, int length, IppsBigNumSGN sgn) {
  m_mag.assign((size_t)(length + 1) / 2, 0);
  if (pData)
    for (int i = 0; i < length; ++i) m_mag[i / 2] |= (uint64_t)pData[i] << (32 * (i & 1));
  m_neg = (sgn == IppsBigNumNEG);
  trim();
}

// String ctor: optional '-', then "0x"/"0X" + lowercase hex digits, or decimal digits
// (reference: bignum.cpp:67-93; digits are looked up in "0123456789abcdef").
BigNumber::BigNumber(const char* s) {
  bool neg = '-' == s[0];
  if (neg) s++;
  bool hex = ('0' == s[0]) && (('x' == s[1]) || ('X' == s[1]));
  if (hex) {
    s += 2;
    size_t len = std::strlen(s);
    m_mag.assign((len + 15) / 16, 0);
    for (size_t i = 0; i < len; ++i) {
      char c = s[len - 1 - i];
      uint64_t d = (c >= '0' && c <= '9') ? (uint64_t)(c - '0')
                   : (c >= 'a' && c <= 'f') ? (uint64_t)(c - 'a' + 10)
                                            : 16;  // same out-of-table index the reference yields
      if (d > 15) throw std::runtime_error("BigNumber: invalid hex digit (lowercase hex only)");
      m_mag[i / 16] |= d << (4 * (i % 16));
    }
    trim();
  } else {
    for (; *s; ++s) {
      if (*s < '0' || *s > '9') throw std::runtime_error("BigNumber: invalid decimal digit");
      *this *= (Ipp32u)10;
      *this += (Ipp32u)(*s - '0');
    }
  }
  if (neg && !m_mag.empty()) m_neg = true;
}

const BigNumber& BigNumber::Zero() {
  static const BigNumber zero((Ipp32u)0);
  return zero;
}
const BigNumber& BigNumber::One() {
  static const BigNumber one((Ipp32u)1);
  return one;
}
const BigNumber& BigNumber::Two() {
  static const BigNumber two((Ipp32u)2);
  return two;
}

// ---- arithmetic (reference: bignum.cpp:146-316) ----
BigNumber& BigNumber::operator+=(const BigNumber& bn) {
  if (m_neg == bn.m_neg) {
    m_mag = mag_add(m_mag, bn.m_mag);
  } else {
    int c = mag_cmp(m_mag, bn.m_mag);
    if (c == 0) {
      m_mag.clear();
    } else if (c > 0) {
      m_mag = mag_sub(m_mag, bn.m_mag);
    } else {
      m_mag = mag_sub(bn.m_mag, m_mag);
      m_neg = bn.m_neg;
    }
  }
  trim();
  return *this;
}
BigNumber& BigNumber::operator+=(Ipp32u n) { return *this += BigNumber(n); }

BigNumber& BigNumber::operator-=(const BigNumber& bn) {
  BigNumber t(bn);
  if (!t.m_mag.empty()) t.m_neg = !t.m_neg;
  return *this += t;
}
BigNumber& BigNumber::operator-=(Ipp32u n) { return *this -= BigNumber(n); }

BigNumber& BigNumber::operator*=(const BigNumber& bn) {
  m_mag = mag_mul(m_mag, bn.m_mag);
  m_neg = (m_neg != bn.m_neg);
  trim();
  return *this;
}
BigNumber& BigNumber::operator*=(Ipp32u n) { return *this *= BigNumber(n); }

void BigNumber::divmod(const BigNumber& a, const BigNumber& d, BigNumber* q, BigNumber* r) {
  Mag qm, rm;
  mag_divmod(a.m_mag, d.m_mag, q ? &qm : nullptr, r ? &rm : nullptr);
  if (q) {
    q->m_mag = qm;
    q->m_neg = (a.m_neg != d.m_neg);
    q->trim();
  }
  if (r) {
    r->m_mag = rm;
    r->m_neg = a.m_neg;
    r->trim();
  }
}

BigNumber& BigNumber::operator/=(const BigNumber& bn) {
  BigNumber q;
  divmod(*this, bn, &q, nullptr);
  *this = q;
  return *this;
}
BigNumber& BigNumber::operator/=(Ipp32u n) { return *this /= BigNumber(n); }

// Non-negative residue, also for a negative left operand (ippsMod_BN semantics; needed by
// pub_key.cpp:42-44 and pri_key.cpp:150 -- SURVEY Appendix A Q1).  Modulus must be positive.
BigNumber operator%(const BigNumber& a, const BigNumber& b) {
  if (b.m_neg || b.m_mag.empty())
    throw std::runtime_error("BigNumber: modulus must be positive");
  BigNumber r;
  BigNumber::divmod(a, b, nullptr, &r);
  if (r.m_neg) r += b;
  return r;
}
BigNumber operator%(const BigNumber& a, Ipp32u n) { return a % BigNumber(n); }
BigNumber& BigNumber::operator%=(const BigNumber& bn) {
  *this = *this % bn;
  return *this;
}
BigNumber& BigNumber::operator%=(Ipp32u n) { return *this %= BigNumber(n); }

BigNumber operator+(const BigNumber& a, const BigNumber& b) {
  BigNumber r(a);
  return r += b;
}
BigNumber operator+(const BigNumber& a, Ipp32u n) {
  BigNumber r(a);
  return r += n;
}
BigNumber operator-(const BigNumber& a, const BigNumber& b) {
  BigNumber r(a);
  return r -= b;
}
BigNumber operator-(const BigNumber& a, Ipp32u n) {
  BigNumber r(a);
  return r -= n;
}
BigNumber operator*(const BigNumber& a, const BigNumber& b) {
  BigNumber r(a);
  return r *= b;
}
BigNumber operator*(const BigNumber& a, Ipp32u n) {
  BigNumber r(a);
  return r *= n;
}
BigNumber operator/(const BigNumber& a, const BigNumber& b) {
  BigNumber q(a);
  return q /= b;
}
BigNumber operator/(const BigNumber& a, Ipp32u n) {
  BigNumber q(a);
  return q /= n;
}

// ---- modulo arithmetic (reference: bignum.cpp:318-360) ----
BigNumber BigNumber::Modulo(const BigNumber& a) const { return a % *this; }

BigNumber BigNumber::InverseAdd(const BigNumber& a) const {
  BigNumber t = Modulo(a);
  if (t.isZero()) return t;
  return *this - t;
}

// a^-1 mod *this by the extended Euclidean algorithm (reference: ippsModInv_BN, bignum.cpp:331-335)
BigNumber BigNumber::InverseMul(const BigNumber& a) const {
  const BigNumber& m = *this;
  BigNumber r0 = m, r1 = a % m;
  BigNumber t0 = Zero(), t1 = One();
  while (!r1.isZero()) {
    BigNumber q, r;
    divmod(r0, r1, &q, &r);
    BigNumber t2 = t0 - q * t1;
    r0 = r1;
    r1 = r;
    t0 = t1;
    t1 = t2;
  }
  if (r0 != One()) throw std::runtime_error("BigNumber::InverseMul: value is not invertible");
  return t0 % m;
}

BigNumber BigNumber::ModAdd(const BigNumber& a, const BigNumber& b) const { return Modulo(a + b); }
BigNumber BigNumber::ModSub(const BigNumber& a, const BigNumber& b) const {
  return Modulo(a + InverseAdd(b));
}
BigNumber BigNumber::ModMul(const BigNumber& a, const BigNumber& b) const { return Modulo(a * b); }

BigNumber BigNumber::gcd(const BigNumber& q) const {
  BigNumber a(*this), b(q);
  a.m_neg = b.m_neg = false;
  while (!b.isZero()) {
    BigNumber r;
    divmod(a, b, nullptr, &r);
    a = b;
    b = r;
  }
  return a;
}

int BigNumber::compare(const BigNumber& bn) const {
  if (m_neg != bn.m_neg) return m_neg ? -1 : 1;
  int c = mag_cmp(m_mag, bn.m_mag);
  return m_neg ? -c : c;
}

// ---- tests and sizes (reference: bignum.cpp:391-458) ----
bool BigNumber::IsOdd() const { return !m_mag.empty() && (m_mag[0] & 1); }

bool BigNumber::TestBit(int index) const {
  if (index < 0) return false;
  size_t limb = (size_t)index / 64;
  if (limb >= m_mag.size()) return false;
  return (m_mag[limb] >> (index % 64)) & 1;
}

int BigNumber::MSB() const {
  if (m_mag.empty()) return 0;
  return (int)(m_mag.size() * 64 - 1 - __builtin_clzll(m_mag.back()));
}

int BigNumber::LSB() const {
  if (m_mag.empty()) return 0;
  int lsb = 0;
  for (size_t i = 0; i < m_mag.size(); ++i) {
    if (m_mag[i] == 0) {
      lsb += 64;
    } else {
      lsb += __builtin_ctzll(m_mag[i]);
      break;
    }
  }
  return lsb;
}

int Bit(const std::vector<Ipp32u>& v, int n) { return 0 != (v[n >> 5] & (1u << (n & 0x1F))); }

// ---- conversions (reference: bignum.cpp:460-509) ----
void BigNumber::num2vec(std::vector<Ipp32u>& v) const {
  int len = BITSIZE_WORD(m_mag.empty() ? 1 : BitSize());  // >= 1 word, also for zero (Q8)
  for (int n = 0; n < len; n++) {
    uint64_t limb = (size_t)(n / 2) < m_mag.size() ? m_mag[n / 2] : 0;
    v.push_back((Ipp32u)(limb >> (32 * (n & 1))));
  }
}

void BigNumber::num2hex(std::string& s) const {
  static const char digits[] = "0123456789abcdef";
  if (m_neg) s.append(1, '-');
  s.append("0x");
  bool started = false;
  for (size_t i = m_mag.size(); i-- > 0;)
    for (int nd = 16; nd > 0; nd--) {
      char c = digits[(m_mag[i] >> ((nd - 1) * 4)) & 0xF];
      if (c != '0' || started) {
        started = true;
        s.append(1, c);
      }
    }
}

std::ostream& operator<<(std::ostream& os, const BigNumber& a) {
  std::string s;
  a.num2hex(s);
  os << s;
  return os;
}

void BigNumber::num2char(std::vector<Ipp8u>& dest) const {
  int len = ((m_mag.empty() ? 1 : BitSize()) + 7) >> 3;
  dest.resize(len);
  for (int i = 0; i < len; ++i) {
    uint64_t limb = (size_t)(i / 8) < m_mag.size() ? m_mag[i / 8] : 0;
    dest[i] = (Ipp8u)(limb >> (8 * (i % 8)));
  }
}

bool BigNumber::fromBin(BigNumber& bn, const unsigned char* data, int len) {
  if (len <= 0) return false;
  Mag m((size_t)(len + 7) / 8, 0);
  for (int i = 0; i < len; i++) m[i / 8] |= (uint64_t)data[len - 1 - i] << (8 * (i % 8));
  bn.m_mag = m;
  bn.m_neg = false;
  bn.trim();
  return true;
}

bool BigNumber::toBin(unsigned char* data, int len, const BigNumber& bn) {
  if (len <= 0) return false;
  int bitSizeLen = BITSIZE_WORD(bn.m_mag.empty() ? 1 : bn.BitSize()) * 4;
  if (bitSizeLen > len) return false;  // the reference would write out of bounds here
  for (int i = 0; i < bitSizeLen; i++) {
    uint64_t limb = (size_t)(i / 8) < bn.m_mag.size() ? bn.m_mag[i / 8] : 0;
    data[len - 1 - i] = (unsigned char)(limb >> (8 * (i % 8)));
  }
  return true;
}

bool BigNumber::toBin(unsigned char** bin, int* len, const BigNumber& bn) {
  if (NULL == bin || NULL == len) return false;
  int bitSizeLen = BITSIZE_WORD(bn.m_mag.empty() ? 1 : bn.BitSize()) * 4;
  *len = bitSizeLen;
  bin[0] = reinterpret_cast<unsigned char*>(std::calloc(bitSizeLen, 1));
  if (NULL == bin[0]) return false;
  return toBin(bin[0], bitSizeLen, bn);
}

// ---- GPU limb layout ----
bool BigNumber::toLimbs64(uint64_t* out, std::size_t nlimbs) const {
  if (m_mag.size() > nlimbs) return false;
  std::copy(m_mag.begin(), m_mag.end(), out);
  std::fill(out + m_mag.size(), out + nlimbs, 0);
  return true;
}

BigNumber BigNumber::fromLimbs64(const uint64_t* limbs, std::size_t nlimbs) {
  BigNumber r;
  r.m_mag.assign(limbs, limbs + nlimbs);
  r.trim();
  return r;
}

BigNumber BigNumber::adoptLimbs64(uint64_t* limbs, std::size_t nlimbs) {
  std::size_t n = nlimbs;
  while (n > 0 && limbs[n - 1] == 0) --n;
  BigNumber r;
  r.m_mag = Limbs::adopt(limbs, n, nlimbs);   // (a zero keeps the block too: it goes back with the value)
  return r;
}

// ---- serialization (reference bignum.h:131-153) ----
void BigNumber::save(ipcl::serializer::OutputArchive& ar) const {
  ar.class_version("BigNumber");
  std::vector<Ipp32u> vec;
  num2vec(vec);
  ar.vec_u32(vec);
  ar.i32(m_neg ? IppsBigNumNEG : IppsBigNumPOS);
}

void BigNumber::load(ipcl::serializer::InputArchive& ar) {
  (void)ar.class_version("BigNumber");
  std::vector<Ipp32u> vec = ar.vec_u32();
  int sign = ar.i32();
  Set(vec.data(), (int)vec.size(), sign == IppsBigNumNEG ? IppsBigNumNEG : IppsBigNumPOS);
}

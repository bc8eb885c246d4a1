// Host key generation (reference ipcl/keygen.cpp).  One-time, sequential: stays on the CPU
// (SURVEY 8(f) N2).  Primality: trial division by small primes, then Miller-Rabin.
#include <algorithm>

#include "ipcl/ipcl.hpp"
#include "ipcl/utils/util.hpp"

namespace ipcl {

namespace {

constexpr int N_BIT_SIZE_MAX = 4096;  // reference: 2048 (keygen.cpp:10); lifted, see ipcl.hpp
constexpr int N_BIT_SIZE_MIN = 200;

const std::vector<Ipp32u>& small_primes() {
  static std::vector<Ipp32u> primes = [] {
    std::vector<Ipp32u> p;
    std::vector<bool> sieve(8192, true);
    for (Ipp32u i = 2; i < 8192; ++i) {
      if (!sieve[i]) continue;
      p.push_back(i);
      for (Ipp32u j = i * i; j < 8192; j += i) sieve[j] = false;
    }
    return p;
  }();
  return primes;
}

// host modexp for Miller-Rabin: left-to-right binary method over BigNumber (moduli differ per
// candidate, so the shared-modulus GPU batch does not apply)
BigNumber host_powmod(const BigNumber& base, const BigNumber& exp, const BigNumber& mod) {
  BigNumber result = BigNumber::One(), b = base % mod;
  for (int i = exp.BitSize() - 1; i >= 0; --i) {
    result = (result * result) % mod;
    if (exp.TestBit(i)) result = (result * b) % mod;
  }
  return result;
}

bool is_probable_prime(const BigNumber& n, int rounds) {
  for (Ipp32u sp : small_primes()) {
    if (n == BigNumber(sp)) return true;
    if ((n % sp).isZero()) return false;
  }
  const BigNumber nm1 = n - 1;
  const int s = nm1.LSB();
  BigNumber d = nm1;
  for (int i = 0; i < s; ++i) d /= (Ipp32u)2;
  for (int r = 0; r < rounds; ++r) {
    BigNumber a = getRandomBN(n.BitSize() + 64) % (n - 3) + 2;
    BigNumber x = host_powmod(a, d, n);
    if (x == BigNumber::One() || x == nm1) continue;
    bool witness = true;
    for (int i = 1; i < s && witness; ++i) {
      x = (x * x) % n;
      if (x == nm1) witness = false;
    }
    if (witness) return false;
  }
  return true;
}

// primes closer than 2^(key/2 - 100) are rejected (reference keygen.cpp:43-59)
BigNumber prime_distance(int64_t key_size) {
  int bit = (int)(key_size / 2 - 100);
  std::vector<Ipp32u> w((size_t)bit / 32 + 1, 0);
  w[(size_t)bit / 32] = 1u << (bit % 32);
  return BigNumber(w.data(), (int)w.size());
}

bool too_close(const BigNumber& p, const BigNumber& q, const BigNumber& ref) {
  BigNumber d = (p >= q) ? (p - q) : (q - p);
  return !(d > ref);
}

}  // namespace

BigNumber getPrimeBN(int max_bits) {
  ERROR_CHECK(max_bits >= 16, "getPrimeBN: bit size too small");
  for (;;) {
    BigNumber c = getRandomBN(max_bits);
    // force exact bit length and oddness
    std::vector<Ipp32u> w;
    c.num2vec(w);
    w.resize((size_t)BITSIZE_WORD(max_bits), 0);
    w[(size_t)(max_bits - 1) / 32] |= 1u << ((max_bits - 1) % 32);
    w[0] |= 1u;
    c = BigNumber(w.data(), (int)w.size());
    if (is_probable_prime(c, 10)) return c;
  }
}

KeyPair generateKeypair(int64_t n_length, bool enable_DJN) {
  ERROR_CHECK(n_length <= N_BIT_SIZE_MAX,
              "generateKeyPair: key size exceeds the supported range (n^2 must fit 8192 bits)");
  ERROR_CHECK((n_length >= N_BIT_SIZE_MIN) && (n_length % 4 == 0),
              "generateKeyPair: key size should >=200, and divisible by 4");
  const BigNumber ref_dist = prime_distance(n_length);
  BigNumber p, q, n;
  const int half = (int)(n_length / 2);
  for (;;) {
    if (enable_DJN) {
      // DJN: p = q = 3 (mod 4) and gcd(p-1, q-1) = 2  (reference keygen.cpp:73-90; the stated
      // intent "q mod 4 = 3" is enforced here, the reference re-tests p by mistake)
      do { p = getPrimeBN(half); } while (!p.TestBit(1));
      do { q = getPrimeBN(half); } while (q == p || !q.TestBit(1));
      if ((p - 1).gcd(q - 1) != BigNumber::Two()) continue;
    } else {
      p = getPrimeBN(half);
      do { q = getPrimeBN(half); } while (q == p);
    }
    n = p * q;
    if (n.BitSize() == n_length && !too_close(p, q, ref_dist)) break;
  }
  PublicKey pk(n, (int)n_length, enable_DJN);
  PrivateKey sk(pk, p, q);
  return KeyPair{pk, sk};
}

}  // namespace ipcl

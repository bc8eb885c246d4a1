// Host randomness (reference ipcl/utils/common.cpp): the OS CSPRNG (expanded by ChaCha20 for bulk requests) replaces
// the RDSEED / RDRAND / IPP-PRNG chain.  Randomness never reaches the GPU path except as data.
#include "ipcl/utils/common.hpp"

#include <sys/random.h>

#include <algorithm>
#include <cerrno>
#include <cstddef>
#include <cstring>
#include <random>

#include "chacha20.hpp"
#include "detail.hpp"
#include "ipcl/utils/util.hpp"

namespace ipcl {

namespace detail {
namespace {
// n bytes straight from the kernel CSPRNG
void os_random(void* dst, std::size_t n) {
  unsigned char* p = static_cast<unsigned char*>(dst);
  while (n > 0) {
    ssize_t got = getrandom(p, n, 0);
    if (got < 0) {
      if (errno == EINTR) continue;
      std::random_device dev;            // (no getrandom: fall back to the C++ source, word by word)
      for (; n >= 4; n -= 4, p += 4) {
        unsigned v = dev();
        std::memcpy(p, &v, 4);
      }
      for (; n > 0; --n, ++p) *p = (unsigned char)dev();
      return;
    }
    p += got;
    n -= (std::size_t)got;
  }
}
}  // namespace

// Random bytes for obfuscator exponents and key material.  Small requests come straight from the kernel CSPRNG; a
// bulk request (a batch of 8192 DJN exponents is 1 MiB, ~3 ms through getrandom(): three times the GPU kernel it
// feeds) is the ChaCha20 key stream of a fresh 256-bit key and 96-bit nonce drawn from the kernel for this request
// alone -- the arc4random / randombytes construction; nothing is kept between calls.  (Blocks are 64 bytes and the
// counter is 32 bits: requests beyond 256 GiB would wrap, far above any batch.)
void fill_random(void* dst, std::size_t n) {
  constexpr std::size_t kBulk = 4096;
  if (n < kBulk) {
    os_random(dst, n);
    return;
  }
  unsigned char seed[44];
  os_random(seed, sizeof(seed));
  chacha20_stream(seed, seed + 32, 0, static_cast<unsigned char*>(dst), n);
  volatile unsigned char* w = seed;
  for (std::size_t i = 0; i < sizeof(seed); ++i) w[i] = 0;
}
}  // namespace detail

void rand32u(std::vector<Ipp32u>& addr) { detail::fill_random(addr.data(), addr.size() * sizeof(Ipp32u)); }

BigNumber getRandomBN(int bits) {
  ERROR_CHECK(bits > 0, "getRandomBN: bit length must be positive");
  std::vector<Ipp32u> w((size_t)BITSIZE_WORD(bits));
  rand32u(w);
  if (bits % 32) w.back() &= (1u << (bits % 32)) - 1;
  return BigNumber(w.data(), (int)w.size());
}

namespace detail {

// BigNumber <-> flat limb arrays.  Deliberately serial: an OpenMP team was measured SLOWER here (fork/join of a
// sleeping team costs more than the ~0.3 ms a conversion of 8192 elements takes, and OpenMP's default team of one
// thread per visible CPU turns each region into hundreds of milliseconds on a 256-thread host under a 16-core quota).

int max_bits(const std::vector<BigNumber>& v) {
  int b = 0;
  const std::ptrdiff_t n = (std::ptrdiff_t)v.size();
  for (std::ptrdiff_t i = 0; i < n; ++i) b = std::max(b, v[(size_t)i].isZero() ? 0 : v[(size_t)i].BitSize());
  return b;
}

std::vector<uint64_t> pack(const std::vector<BigNumber>& v, int words) {
  std::vector<uint64_t> flat(v.size() * (size_t)words);
  const std::ptrdiff_t n = (std::ptrdiff_t)v.size();
  bool fits = true;
  for (std::ptrdiff_t i = 0; i < n; ++i)
    fits = fits && v[(size_t)i].toLimbs64(flat.data() + (size_t)i * (size_t)words, (size_t)words);
  ERROR_CHECK(fits, "pack: value wider than the batch stride");
  return flat;
}

std::vector<BigNumber> unpack(const std::vector<uint64_t>& flat, std::size_t count, int words) {
  std::vector<BigNumber> v(count);
  const std::ptrdiff_t n = (std::ptrdiff_t)count;
  for (std::ptrdiff_t i = 0; i < n; ++i)
    v[(size_t)i] = BigNumber::fromLimbs64(flat.data() + (size_t)i * (size_t)words, (size_t)words);
  return v;
}

std::shared_ptr<DeviceBatch> DeviceBatch::adopt(pgpu_batch* h) {
  auto b = std::make_shared<DeviceBatch>();
  b->h = h;
  b->count = pgpu_batch_count(h);
  b->words = pgpu_batch_words(h);
  return b;
}

std::shared_ptr<DeviceBatch> DeviceBatch::upload(const std::vector<uint64_t>& flat, std::size_t count, int words) {
  ensure_context();
  pgpu_batch* h = nullptr;
  IPCL_GPU_CHECK(pgpu_batch_upload(flat.data(), count, words, (size_t)words, &h), "device upload");
  return adopt(h);
}

std::vector<BigNumber> DeviceBatch::download() const {
  std::vector<uint64_t> flat(count * (size_t)words);
  IPCL_GPU_CHECK(pgpu_batch_download(h, flat.data()), "device download");
  return unpack(flat, count, words);
}

}  // namespace detail
}  // namespace ipcl

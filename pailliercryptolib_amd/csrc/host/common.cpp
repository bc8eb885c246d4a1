// Host randomness (reference ipcl/utils/common.cpp): the OS CSPRNG replaces the
// RDSEED / RDRAND / IPP-PRNG chain.  Randomness never reaches the GPU path except as data.
#include "ipcl/utils/common.hpp"

#include <random>

#include "detail.hpp"
#include "ipcl/utils/util.hpp"

namespace ipcl {

void rand32u(std::vector<Ipp32u>& addr) {
  std::random_device dev;
  for (auto& x : addr) x = dev();
}

BigNumber getRandomBN(int bits) {
  ERROR_CHECK(bits > 0, "getRandomBN: bit length must be positive");
  std::vector<Ipp32u> w((size_t)BITSIZE_WORD(bits));
  rand32u(w);
  if (bits % 32) w.back() &= (1u << (bits % 32)) - 1;
  return BigNumber(w.data(), (int)w.size());
}

namespace detail {

int max_bits(const std::vector<BigNumber>& v) {
  int b = 0;
  for (const auto& x : v) b = std::max(b, x.isZero() ? 0 : x.BitSize());
  return b;
}

std::vector<uint64_t> pack(const std::vector<BigNumber>& v, int words) {
  std::vector<uint64_t> flat(v.size() * (size_t)words);
  for (size_t i = 0; i < v.size(); ++i)
    ERROR_CHECK(v[i].toLimbs64(flat.data() + i * (size_t)words, (size_t)words),
                "pack: value wider than the batch stride");
  return flat;
}

std::vector<BigNumber> unpack(const std::vector<uint64_t>& flat, std::size_t count, int words) {
  std::vector<BigNumber> v(count);
  for (size_t i = 0; i < count; ++i) v[i] = BigNumber::fromLimbs64(flat.data() + i * (size_t)words, (size_t)words);
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

// Internal helpers of the ipcl:: host layer: BigNumber <-> flat limb batches, device key holders.
#ifndef PAILLIERCRYPTOLIB_AMD_CSRC_HOST_DETAIL_HPP_
#define PAILLIERCRYPTOLIB_AMD_CSRC_HOST_DETAIL_HPP_

#include <cstdint>
#include <memory>
#include <vector>

#include "ipcl/bignum.h"
#include "pgpu.h"

namespace ipcl {
namespace detail {

// makes sure the process owns a GPU context (initializeContext may not have been called:
// the reference's CPU path needs no initialisation either)
void ensure_context();

inline int words_for_bits(int bits) { return bits <= 0 ? 1 : (bits + 63) / 64; }

// row-major [count][words] little-endian limbs of |v[i]|; every value must fit
std::vector<uint64_t> pack(const std::vector<BigNumber>& v, int words);
std::vector<BigNumber> unpack(const std::vector<uint64_t>& flat, std::size_t count, int words);
int max_bits(const std::vector<BigNumber>& v);
void fill_random(void* dst, std::size_t n);   // kernel CSPRNG, bulk

// A batch resident in GPU memory: [count][words] little-endian 64-bit limbs, cut into contiguous shards over
// the device pool (pgpu_batch).  Immutable once produced (results are always written to fresh batches), so
// copies of a text can share it.  Ciphertext batches produced on the device are in the Montgomery domain of
// n^2; download() returns plain values either way.
struct DeviceBatch {
  pgpu_batch* h = nullptr;
  std::size_t count = 0;
  int words = 0;
  ~DeviceBatch() {
    if (h) pgpu_batch_destroy(h);
  }
  static std::shared_ptr<DeviceBatch> adopt(pgpu_batch* h);
  static std::shared_ptr<DeviceBatch> upload(const std::vector<uint64_t>& flat, std::size_t count, int words);
  std::vector<BigNumber> download() const;
};

struct PubKeyDevice {
  pgpu_pubkey* h = nullptr;
  BigNumber n, hs;
  bool djn = false;
  ~PubKeyDevice() {
    if (h) pgpu_pubkey_destroy(h);
  }
};

struct PrivKeyDevice {
  pgpu_privkey* h = nullptr;
  ~PrivKeyDevice() {
    if (h) pgpu_privkey_destroy(h);
  }
};

}  // namespace detail
}  // namespace ipcl
#endif

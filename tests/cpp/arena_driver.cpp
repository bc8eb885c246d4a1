// Test driver for the limb storage of the host BigNumber (include/ipcl/bignum.h: LimbVec, LimbBulkScope): blocks
// carved out of an arena inside a bulk scope, freed in any order and from any thread, mixed with heap blocks.
#include <atomic>
#include <cstdio>
#include <thread>
#include <vector>
#include "ipcl/bignum.h"

static BigNumber value(unsigned i, int limbs) {
  std::vector<uint64_t> w((size_t)limbs);
  for (int k = 0; k < limbs; ++k) w[(size_t)k] = 0x9e3779b97f4a7c15ull * (i + 1) + (uint64_t)k;
  return BigNumber::fromLimbs64(w.data(), w.size());
}

int main() {
  int bad = 0;
  // 1. values created inside a scope equal the ones created outside; the arena outlives the scope while blocks live
  std::vector<BigNumber> in_scope(5000), plain(5000);
  {
    ipcl::detail::LimbBulkScope arena(5000 * (64 * 8 + 32));
    for (unsigned i = 0; i < 5000; ++i) in_scope[i] = value(i, 1 + (int)(i % 64));
  }
  for (unsigned i = 0; i < 5000; ++i) plain[i] = value(i, 1 + (int)(i % 64));
  for (unsigned i = 0; i < 5000; ++i) bad += in_scope[i] != plain[i];
  // 2. arithmetic on arena-backed values (results come from the heap outside a scope), growth beyond the arena's hint
  {
    ipcl::detail::LimbBulkScope tiny(4096);      // far too small: the rest silently comes from malloc
    std::vector<BigNumber> v(300);
    for (unsigned i = 0; i < 300; ++i) v[i] = value(i, 32) * value(i + 1, 32) % value(i + 2, 31);
    for (unsigned i = 0; i < 300; ++i) bad += v[i] != value(i, 32) * value(i + 1, 32) % value(i + 2, 31);
  }
  // 3. freed from other threads, in a scrambled order, while the creating thread is already in another scope
  std::vector<std::thread> th;
  std::atomic<int> sum{0};
  for (int t = 0; t < 4; ++t)
    th.emplace_back([&, t] {
      for (unsigned i = (unsigned)t; i < 5000; i += 4) {
        sum += in_scope[(i * 7919u) % 5000u].IsOdd() ? 1 : 0;
      }
    });
  for (auto& x : th) x.join();
  th.clear();
  for (int t = 0; t < 4; ++t)
    th.emplace_back([&, t] {
      for (unsigned i = (unsigned)t; i < 5000; i += 4) in_scope[(i * 7919u + 13u) % 5000u] = BigNumber();   // releases its block
    });
  {
    ipcl::detail::LimbBulkScope another(1 << 20);
    std::vector<BigNumber> w(1000);
    for (unsigned i = 0; i < 1000; ++i) w[i] = value(i, 16);
    for (auto& x : th) x.join();
    for (unsigned i = 0; i < 1000; ++i) bad += w[i] != value(i, 16);
    // nested scopes share the outer arena
    ipcl::detail::LimbBulkScope inner(1 << 20);
    BigNumber z = value(3, 8);
    bad += z != value(3, 8);
  }
  // 4. copies and moves between arena and heap storage
  BigNumber a;
  {
    ipcl::detail::LimbBulkScope s(1 << 16);
    BigNumber b = value(42, 20);
    a = b;                 // copy inside the scope: arena block
    BigNumber c = std::move(b);
    bad += c != value(42, 20);
  }
  bad += a != value(42, 20);
  a += 1u;
  bad += a != value(42, 20) + 1u;
  // 5. retired arenas are recycled: a batch dropped before the next one is built hands its pages on; a survivor of the
  //    old batch keeps ITS arena out of the cache (so nothing is overwritten under it); trimming the cache is harmless
  const void* first = nullptr;
  BigNumber survivor;
  for (int round = 0; round < 6; ++round) {
    std::vector<BigNumber> v(2000);
    {
      ipcl::detail::LimbBulkScope s(2000 * (32 * 8 + 32));
      for (unsigned i = 0; i < 2000; ++i) v[i] = value(i + (unsigned)round, 32);
    }
    if (round == 0) first = v[0].limbs64().data();
    if (round == 1) bad += v[0].limbs64().data() != first;     // round 0's arena died with its vector: same pages again
    if (round == 2) survivor = std::move(v[7]);                // pins round 2's arena
    if (round == 4) ipcl::detail::limb_cache_trim();
    for (unsigned i = 0; i < 2000; ++i)
      if (!(round == 2 && i == 7)) bad += v[i] != value(i + (unsigned)round, 32);
    bad += round > 2 && survivor != value(7 + 2, 32);
  }
  // 6. values that ADOPT rows of a caller buffer (results used in place): 64 bytes of arena control, then per value a
  //    16-byte header and its row; the release hook runs once, after the last value that points into the buffer has died
  //    (or grown out of it), whatever thread that happens on
  {
    static std::atomic<int> released{0};
    const size_t rows = 300, words = 32, stride = words + 2;
    std::vector<uint64_t>* backing = new std::vector<uint64_t>(8 + rows * stride, 0xdeadbeefdeadbeefull);
    uint64_t* base = backing->data();
    for (size_t i = 0; i < rows; ++i)
      for (size_t k = 0; k < words; ++k)
        base[8 + 2 + i * stride + k] = (i == 5) ? 0 : (k < 1 + i % words ? 0x9e3779b97f4a7c15ull * (i + 1) + k : 0);   // ragged, one zero
    ipcl::detail::LimbArenaExt* ar = ipcl::detail::limb_arena_open(base, [](void* ck) {
      released++;
      delete static_cast<std::vector<uint64_t>*>(ck);
    }, backing);
    std::vector<BigNumber> v(rows);
    for (size_t i = 0; i < rows; ++i) {
      uint64_t* row = base + 8 + 2 + i * stride;
      ipcl::detail::limb_block_adopt(ar, row);
      v[i] = BigNumber::adoptLimbs64(row, words);
    }
    ipcl::detail::limb_arena_close(ar);
    bad += released.load() != 0;
    for (size_t i = 0; i < rows; ++i) {
      std::vector<uint64_t> w(words, 0);
      for (size_t k = 0; k < words; ++k) w[k] = (i == 5) ? 0 : (k < 1 + i % words ? 0x9e3779b97f4a7c15ull * (i + 1) + k : 0);
      bad += v[i] != BigNumber::fromLimbs64(w.data(), words);
      bad += v[i].limbs64().size() != (i == 5 ? 0u : 1 + i % words);
      bad += (i != 5) && v[i].limbs64().data() != base + 8 + 2 + i * stride;     // in place, not copied
    }
    BigNumber copy = v[7];                                   // a copy owns limbs of its own
    bad += copy.limbs64().data() == v[7].limbs64().data();
    BigNumber grown = std::move(v[8]);                       // a move keeps the row ...
    bad += grown.limbs64().data() != base + 8 + 2 + 8 * stride;
    for (int k = 0; k < 40; ++k) grown = grown * grown % (v[9] + 12345u) + v[10];   // ... arithmetic leaves it behind
    bad += grown.isZero();
    std::vector<std::thread> th2;
    for (int t = 0; t < 3; ++t)
      th2.emplace_back([&, t] {
        for (size_t i = (size_t)t; i < rows; i += 3) v[i] = BigNumber();
      });
    for (auto& x : th2) x.join();
    bad += released.load() != 1;                             // every row has been given back: the buffer is gone
    bad += copy.isZero();                                    // (the copy outlives the buffer)
  }
  std::printf("%s %d\n", bad ? "FAIL" : "OK", bad);
  return bad ? 1 : 0;
}

// Test driver for the limb storage of the host BigNumber (include/ipcl/bignum.h: LimbAllocator, LimbBulkScope): blocks
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
  std::printf("%s %d\n", bad ? "FAIL" : "OK", bad);
  return bad ? 1 : 0;
}

// Syntax-check stand-in for <benchmark/benchmark.h> (tests/test_dropin_compile.py): the slice of google-benchmark the
// reference's benchmark/*.cpp use, so that those files -- compiled in place, never copied -- parse and type-check
// against include/ipcl.  Nothing here measures anything.
#ifndef PAILLIERCRYPTOLIB_AMD_TESTS_SHIMS_BENCHMARK_H_
#define PAILLIERCRYPTOLIB_AMD_TESTS_SHIMS_BENCHMARK_H_
#include <cstdint>
#include <vector>

namespace benchmark {
enum TimeUnit { kNanosecond, kMicrosecond, kMillisecond, kSecond };
class State {
 public:
  int64_t range(int = 0) const { return 0; }
  struct It {
    bool operator!=(const It&) const { return false; }
    void operator++() {}
    int operator*() const { return 0; }
  };
  It begin() { return It(); }
  It end() { return It(); }
  bool KeepRunning() { return false; }
  void PauseTiming() {}
  void ResumeTiming() {}
  void SetItemsProcessed(int64_t) {}
};
namespace internal {
class Benchmark {
 public:
  Benchmark* Args(const std::vector<int64_t>&) { return this; }
  Benchmark* Arg(int64_t) { return this; }
  Benchmark* Unit(TimeUnit) { return this; }
  Benchmark* Apply(void (*)(Benchmark*)) { return this; }
  Benchmark* Iterations(int64_t) { return this; }
  Benchmark* Repetitions(int) { return this; }
  Benchmark* UseRealTime() { return this; }
};
inline Benchmark* RegisterBenchmark(const char*, void (*)(State&)) {
  static Benchmark b;
  return &b;
}
}  // namespace internal
inline void Initialize(int*, char**) {}
inline void RunSpecifiedBenchmarks() {}
inline void Shutdown() {}
template <class T>
inline void DoNotOptimize(T&&) {}
inline void ClobberMemory() {}
}  // namespace benchmark
#define PGPU_SHIM_CAT2(a, b) a##b
#define PGPU_SHIM_CAT(a, b) PGPU_SHIM_CAT2(a, b)
#define BENCHMARK(fn) static ::benchmark::internal::Benchmark* PGPU_SHIM_CAT(shim_bench_, __LINE__) = ::benchmark::internal::RegisterBenchmark(#fn, fn)
#define BENCHMARK_MAIN() int main() { return 0; }
#endif

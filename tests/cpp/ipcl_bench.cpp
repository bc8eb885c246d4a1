// API-level benchmark of the ipcl:: mirror, after the reference's google-benchmark suite
// (benchmark/bench_cryptography.cpp:65-121, benchmark/bench_ops.cpp:65-153): same fixed ISO key,
// DJN with injected hs / r, same batch sizes -- BM_Encrypt, BM_Decrypt, BM_Add_CTCT, BM_Add_CTPT,
// BM_Mul_CTPT.  Times the API-visible call (marshalling + H2D + kernels; results stay resident,
// like the reference's results stay in BigNumbers) and, per op, the chained variant.
// Prints one line per (op, batch): microseconds per call (best of N) and elements/s.
#include <chrono>
#include <cstdio>
#include <functional>
#include <vector>

#include "ipcl/ipcl.hpp"
#include "kat_vectors.inc"

static double best_us(const std::function<void()>& f, int reps) {
  double best = 1e30;
  for (int i = 0; i < reps; ++i) {
    auto t0 = std::chrono::steady_clock::now();
    f();
    auto t1 = std::chrono::steady_clock::now();
    best = std::min(best, std::chrono::duration<double, std::micro>(t1 - t0).count());
  }
  return best;
}

int main(int argc, char** argv) {
  ipcl::initializeContext("default");
  BigNumber P(KAT_P), Q(KAT_Q), n = P * Q;
  std::vector<size_t> sizes = {16, 64, 128, 256, 512, 1024, 2048, 2100};   // bench_cryptography.cpp:12-19
  if (argc > 1) sizes = {8192};
  std::printf("%-14s %8s %14s %16s\n", "op", "batch", "us/call", "elements/s");
  for (size_t dsize : sizes) {
    ipcl::PublicKey pk(n, 2048, true);
    ipcl::PrivateKey sk(pk, P, Q);
    pk.setRandom(std::vector<BigNumber>(dsize, BigNumber(KAT_BENCH_R)));
    pk.setHS(BigNumber(KAT_BENCH_HS));
    std::vector<BigNumber> m(dsize), m2(dsize);
    for (size_t i = 0; i < dsize; i++) {
      m[i] = P - BigNumber((unsigned int)(i * 1024));      // bench_cryptography.cpp:87
      m2[i] = Q + BigNumber((unsigned int)(i * 1024));     // bench_ops.cpp
    }
    ipcl::PlainText pt(m), pt2(m2);
    ipcl::CipherText ct = pk.encrypt(pt), ct2 = pk.encrypt(pt2), out;
    ipcl::PlainText dt;
    (void)sk.decrypt(ct).getElement(0);   // warm up (tables, workspaces)
    // steady state of a key in service: past its first 4096 elements the fixed-base table has its final
    // window (include/pgpu.h: pgpu_set_fixed_base_window)
    for (size_t done = 2 * dsize; done < 4096 + dsize; done += dsize) (void)pk.encrypt(pt).getElement(0);
    auto report = [&](const char* name, double us) {
      std::printf("%-14s %8zu %14.1f %16.0f\n", name, dsize, us, dsize / us * 1e6);
    };
    report("Encrypt", best_us([&] { out = pk.encrypt(pt); (void)out.getElement(0); }, 5));
    report("Decrypt", best_us([&] { dt = sk.decrypt(ct); (void)dt.getElement(0); }, 5));
    report("Add_CTCT", best_us([&] { out = ct + ct2; (void)out.getElement(0); }, 5));
    report("Add_CTPT", best_us([&] { out = ct + pt2; (void)out.getElement(0); }, 5));
    report("Mul_CTPT", best_us([&] { out = ct * pt2; (void)out.getElement(0); }, 3));
    report("Chain(E,+,D)", best_us([&] { dt = sk.decrypt(pk.encrypt(pt) + ct2); (void)dt.getElement(0); }, 3));
  }
  ipcl::terminateContext();
  return 0;
}

// API-level benchmark of the ipcl:: mirror, after the reference's google-benchmark suite
// (benchmark/bench_cryptography.cpp:65-121, benchmark/bench_ops.cpp:65-153): same fixed ISO key,
// DJN with injected hs / r, same batch sizes -- BM_Encrypt, BM_Decrypt, BM_Add_CTCT, BM_Add_CTPT,
// BM_Mul_CTPT.  Times the API-visible call (marshalling + H2D + kernels; results stay resident,
// like the reference's results stay in BigNumbers) and, per op, the chained variant.
// Prints one line per (op, batch): microseconds per call (best of N) and elements/s.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <atomic>
#include <thread>
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

// `--json <batch>`: the API-visible timed region of the reference's BM_Encrypt / BM_Decrypt
// (benchmark/bench_cryptography.cpp:73-121) for bench.py: std::vector<BigNumber> in, std::vector<BigNumber> out
// (the reference's results ARE host BigNumbers, so getTexts() is inside the timed call), one JSON line.
static int json_mode(size_t dsize) {
  ipcl::initializeContext("default");
  BigNumber P(KAT_P), Q(KAT_Q), n = P * Q;
  ipcl::PublicKey pk(n, 2048, true);
  ipcl::PrivateKey sk(pk, P, Q);
  pk.setRandom(std::vector<BigNumber>(dsize, BigNumber(KAT_BENCH_R)));
  pk.setHS(BigNumber(KAT_BENCH_HS));
  std::vector<BigNumber> m(dsize);
  for (size_t i = 0; i < dsize; i++) m[i] = P - BigNumber((unsigned int)(i * 1024));   // bench_cryptography.cpp:87
  std::vector<BigNumber> c, d;
  for (size_t done = 0; done < 4096 + dsize; done += dsize) c = pk.encrypt(ipcl::PlainText(m)).getTexts();   // warm-up
  d = sk.decrypt(ipcl::CipherText(pk, c)).getTexts();
  const double enc = best_us([&] { c = pk.encrypt(ipcl::PlainText(m)).getTexts(); }, 5);
  const double dec = best_us([&] { d = sk.decrypt(ipcl::CipherText(pk, c)).getTexts(); }, 5);
  bool ok = d.size() == m.size();
  for (size_t i = 0; ok && i < dsize; ++i) ok = d[i] == m[i];
  // resident chaining (the texts never become BigNumbers in between): encrypt -> CT+CT -> decrypt -> first element
  ipcl::PlainText pt(m);
  ipcl::CipherText ct2 = pk.encrypt(pt);
  const double chain = best_us([&] { (void)sk.decrypt(pk.encrypt(pt) + ct2).getElement(0); }, 8);   // (the first calls after new texts run slow)
  std::printf("{\"what\": \"ipcl::PublicKey::encrypt / PrivateKey::decrypt, vector<BigNumber> in and out, batch %zu\", "
              "\"encrypt_us\": %.1f, \"decrypt_us\": %.1f, \"chain_enc_add_dec_us\": %.1f, \"round_trip_ok\": %s}\n",
              dsize, enc, dec, chain, ok ? "true" : "false");
  ipcl::terminateContext();
  return ok ? 0 : 1;
}

// `--threads <T> <batch> [rounds]`: T host threads, each encrypting and decrypting vectors of its own through the API
// (vector<BigNumber> in and out), side by side -- the shape of the reference's OpenMP tests (test_cryptography.cpp:45-57);
// aggregate rate, one JSON line.
static int threads_mode(int T, size_t dsize, int rounds) {
  ipcl::initializeContext("default");
  BigNumber P(KAT_P), Q(KAT_Q), n = P * Q;
  ipcl::PublicKey pk(n, 2048, true);
  ipcl::PrivateKey sk(pk, P, Q);
  pk.setRandom(std::vector<BigNumber>(dsize, BigNumber(KAT_BENCH_R)));
  pk.setHS(BigNumber(KAT_BENCH_HS));
  std::vector<std::vector<BigNumber>> m((size_t)T, std::vector<BigNumber>(dsize));
  for (int t = 0; t < T; ++t)
    for (size_t i = 0; i < dsize; i++) m[(size_t)t][i] = P - BigNumber((unsigned int)(i * 1024 + (size_t)t));
  std::vector<int> ok((size_t)T, 1);
  std::vector<double> slowest((size_t)T, 0.0);
  std::vector<std::chrono::steady_clock::time_point> done_at((size_t)T);
  std::atomic<int> ready{0}, go{0};
  std::vector<std::thread> th;
  std::chrono::steady_clock::time_point t0;
  for (int t = 0; t < T; ++t)
    th.emplace_back([&, t] {
      std::vector<BigNumber> c, d;
      for (int r = 0; r < rounds + 2; ++r) {
        if (r == 2) {   // two untimed rounds (tables, workspaces, arenas), then all threads start together
          ready.fetch_add(1);
          while (!go.load()) std::this_thread::yield();
        }
        const auto r0 = std::chrono::steady_clock::now();
        c = pk.encrypt(ipcl::PlainText(m[(size_t)t])).getTexts();
        d = sk.decrypt(ipcl::CipherText(pk, c)).getTexts();
        if (r >= 2) slowest[(size_t)t] = std::max(slowest[(size_t)t], std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - r0).count());
      }
      done_at[(size_t)t] = std::chrono::steady_clock::now();
      for (size_t i = 0; i < dsize; ++i) ok[(size_t)t] &= d[i] == m[(size_t)t][i];
    });
  while (ready.load() < T) std::this_thread::yield();
  t0 = std::chrono::steady_clock::now();
  go.store(1);
  for (auto& x : th) x.join();
  double us = 0, worst = 0;
  for (int t = 0; t < T; ++t) {
    us = std::max(us, std::chrono::duration<double, std::micro>(done_at[(size_t)t] - t0).count());
    worst = std::max(worst, slowest[(size_t)t]);
  }
  bool all = true;
  for (int v : ok) all = all && v;
  std::printf("{\"what\": \"%d host threads, each ipcl::PublicKey::encrypt + PrivateKey::decrypt with vector<BigNumber> in and out, "
              "batch %zu, %d rounds each\", \"threads\": %d, \"us_per_encrypt_plus_decrypt\": %.1f, \"modexps_per_s\": %.1f, "
              "\"slowest_round_us\": %.1f, \"round_trip_ok\": %s}\n",
              T, dsize, rounds, T, us / (rounds * T), 3.0 * dsize * rounds * T / (us * 1e-6), worst, all ? "true" : "false");
  ipcl::terminateContext();
  return all ? 0 : 1;
}

// `--threads-mul <T> <batch> [rounds]`: T host threads, each multiplying a ciphertext vector of its own by a plaintext vector
// (CipherText * PlainText, a full-width exponent per element; the product stays resident, one element fetched); aggregate rate
static int threads_mul_mode(int T, size_t dsize, int rounds) {
  ipcl::initializeContext("default");
  BigNumber P(KAT_P), Q(KAT_Q), n = P * Q;
  ipcl::PublicKey pk(n, 2048, true);
  ipcl::PrivateKey sk(pk, P, Q);
  pk.setRandom(std::vector<BigNumber>(dsize, BigNumber(KAT_BENCH_R)));
  pk.setHS(BigNumber(KAT_BENCH_HS));
  std::vector<int> ok((size_t)T, 1);
  std::vector<double> slowest((size_t)T, 0.0), slowest_call((size_t)T, 0.0);
  std::vector<std::chrono::steady_clock::time_point> done_at((size_t)T);
  std::atomic<int> ready{0}, go{0};
  std::vector<std::thread> th;
  for (int t = 0; t < T; ++t)
    th.emplace_back([&, t] {
      std::vector<BigNumber> m(dsize), e(dsize);
      for (size_t i = 0; i < dsize; i++) {
        m[i] = BigNumber((unsigned int)(i * 1024 + (size_t)t + 1));
        e[i] = Q - BigNumber((unsigned int)(i + 7 * (size_t)t));
      }
      ipcl::CipherText ct = pk.encrypt(ipcl::PlainText(m));
      ipcl::PlainText pt(e);
      ipcl::CipherText out;
      for (int r = 0; r < rounds + 2; ++r) {
        if (r == 2) {
          ready.fetch_add(1);
          while (!go.load()) std::this_thread::yield();
        }
        const auto r0 = std::chrono::steady_clock::now();
        out = ct * pt;
        const auto r1 = std::chrono::steady_clock::now();
        (void)out.getElement(0);
        if (r >= 2) {
          slowest[(size_t)t] = std::max(slowest[(size_t)t], std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - r0).count());
          slowest_call[(size_t)t] = std::max(slowest_call[(size_t)t], std::chrono::duration<double, std::micro>(r1 - r0).count());
        }
      }
      done_at[(size_t)t] = std::chrono::steady_clock::now();   // (the check below is not part of the measurement)
      std::vector<BigNumber> d = sk.decrypt(out).getTexts();
      size_t bad = 0, first = 0;
      for (size_t i = 0; i < dsize; ++i)
        if (!(d[i] == (m[i] * e[i]) % n)) {
          if (!bad) first = i;
          ++bad;
        }
      ok[(size_t)t] = bad == 0;
      if (bad) {   // which of the two was wrong -- the product or the decrypt beside three others?  Once more, and from host copies
        std::vector<BigNumber> d2 = sk.decrypt(out).getTexts();
        std::vector<BigNumber> d3 = sk.decrypt(ipcl::CipherText(pk, out.getTexts())).getTexts();
        size_t bad2 = 0, bad3 = 0;
        for (size_t i = 0; i < dsize; ++i) {
          bad2 += !(d2[i] == (m[i] * e[i]) % n);
          bad3 += !(d3[i] == (m[i] * e[i]) % n);
        }
        std::fprintf(stderr, "thread %d: %zu of %zu products wrong (first at %zu); decrypted again: %zu wrong; from the host copy: %zu wrong\n",
                     t, bad, dsize, first, bad2, bad3);
        // what do the wrong values look like?  ranges, zeros, another thread's plaintexts (expected values are a function of t and i)
        std::string ranges;
        size_t zeros = 0, other = 0, again_same = 0;
        for (size_t i = 0; i < dsize; ++i) {
          const bool w = !(d[i] == (m[i] * e[i]) % n);
          if (w && (i == 0 || d[i - 1] == (m[i - 1] * e[i - 1]) % n)) ranges += " " + std::to_string(i) + "-";
          if (!w && i > 0 && !(d[i - 1] == (m[i - 1] * e[i - 1]) % n)) ranges += std::to_string(i - 1);
          if (!w) continue;
          zeros += d[i] == BigNumber::Zero();
          again_same += d[i] == d2[i];
          for (int u = 0; u < T; ++u) {
            if (u == t) continue;
            const BigNumber mu((unsigned int)(i * 1024 + (size_t)u + 1)), eu = Q - BigNumber((unsigned int)(i + 7 * (size_t)u));
            other += d[i] == (mu * eu) % n;
          }
        }
        std::fprintf(stderr, "thread %d: wrong ranges%s; zeros %zu, another thread's value at the same index %zu, same wrong value the second time %zu\n",
                     t, ranges.c_str(), zeros, other, again_same);
      }
    });
  while (ready.load() < T) std::this_thread::yield();
  const auto t0 = std::chrono::steady_clock::now();
  go.store(1);
  for (auto& x : th) x.join();
  double us = 0, worst = 0, worst_call = 0;
  for (int t = 0; t < T; ++t) {
    us = std::max(us, std::chrono::duration<double, std::micro>(done_at[(size_t)t] - t0).count());
    worst = std::max(worst, slowest[(size_t)t]);
    worst_call = std::max(worst_call, slowest_call[(size_t)t]);
  }
  bool all = true;
  for (int v : ok) all = all && v;
  std::printf("{\"what\": \"%d host threads, each CipherText * PlainText on a resident vector of %zu, %d rounds each\", "
              "\"threads\": %d, \"us_per_mul\": %.1f, \"modexps_per_s\": %.1f, \"slowest_round_us\": %.1f, \"slowest_operator_call_us\": %.1f, "
              "\"products_ok\": %s}\n",
              T, dsize, rounds, T, us / (rounds * T), 1.0 * dsize * rounds * T / (us * 1e-6), worst, worst_call, all ? "true" : "false");
  ipcl::terminateContext();
  return all ? 0 : 1;
}

int main(int argc, char** argv) {
  if (argc > 2 && std::string(argv[1]) == "--json") return json_mode((size_t)std::atol(argv[2]));
  if (argc > 3 && std::string(argv[1]) == "--threads-mul")
    return threads_mul_mode(std::atoi(argv[2]), (size_t)std::atol(argv[3]), argc > 4 ? std::atoi(argv[4]) : 8);
  if (argc > 3 && std::string(argv[1]) == "--threads")
    return threads_mode(std::atoi(argv[2]), (size_t)std::atol(argv[3]), argc > 4 ? std::atoi(argv[4]) : 8);
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

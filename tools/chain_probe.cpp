// Where does Chain(E,+,D).getElement(0) spend its time after result vectors have been taken with getTexts()?  (diagnostics)
#include <chrono>
#include <cstdio>
#include <functional>
#include <vector>
#include "ipcl/ipcl.hpp"
#include "kat_vectors.inc"
#include "pgpu.h"
using clk = std::chrono::steady_clock;
static double us(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }
int main(int argc, char** argv) {
  const size_t N = 8192;
  const bool hold = argc > 1;
  ipcl::initializeContext("default");
  BigNumber P(KAT_P), Q(KAT_Q), n = P * Q;
  ipcl::PublicKey pk(n, 2048, true);
  ipcl::PrivateKey sk(pk, P, Q);
  pk.setRandom(std::vector<BigNumber>(N, BigNumber(KAT_BENCH_R)));
  pk.setHS(BigNumber(KAT_BENCH_HS));
  std::vector<BigNumber> m(N);
  for (size_t i = 0; i < N; i++) m[i] = P - BigNumber((unsigned int)(i * 1024));
  std::vector<BigNumber> c, d;
  const bool late = argc > 2;
  ipcl::PlainText pt0(m);
  ipcl::CipherText ct20 = pk.encrypt(pt0);
  if (hold) {
    for (int k = 0; k < 3; ++k) c = pk.encrypt(ipcl::PlainText(m)).getTexts();
    for (int k = 0; k < 3; ++k) d = sk.decrypt(ipcl::CipherText(pk, c)).getTexts();
  }
  ipcl::PlainText pt = late ? ipcl::PlainText(m) : pt0;
  ipcl::CipherText ct2 = late ? pk.encrypt(pt) : ct20;
  for (int rep = 0; rep < 5; ++rep) {
    auto t0 = clk::now();
    ipcl::PlainText r = sk.decrypt(pk.encrypt(pt) + ct2);
    auto t1 = clk::now();
    pgpu_synchronize();
    auto t2 = clk::now();
    BigNumber e0 = r.getElement(0);
    auto t3 = clk::now();
    r = ipcl::PlainText();
    auto t4 = clk::now();
    std::printf("hold=%d issue %.1f kernels %.1f getElement %.1f drop %.1f\n", (int)hold, us(t0, t1), us(t1, t2), us(t2, t3), us(t3, t4));
  }
  ipcl::terminateContext();
}

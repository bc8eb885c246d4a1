// Where does the API-level time go?  Stage timings of ipcl::PublicKey::encrypt / PrivateKey::decrypt with
// std::vector<BigNumber> in and out (the timed region of the reference's BM_Encrypt / BM_Decrypt), batch 8192,
// next to the same work through the C-ABI on prepared flat arrays.  (tools/, diagnostics only)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>

#include "detail.hpp"
#include "ipcl/ipcl.hpp"
#include "kat_vectors.inc"
#include "pgpu.h"

static double best_us(const std::function<void()>& f, int reps = 7) {
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
  const size_t N = argc > 1 ? (size_t)std::atol(argv[1]) : 8192;
  ipcl::initializeContext("default");
  BigNumber P(KAT_P), Q(KAT_Q), n = P * Q;
  ipcl::PublicKey pk(n, 2048, true);
  ipcl::PrivateKey sk(pk, P, Q);
  pk.setRandom(std::vector<BigNumber>(N, BigNumber(KAT_BENCH_R)));
  pk.setHS(BigNumber(KAT_BENCH_HS));
  std::vector<BigNumber> m(N);
  for (size_t i = 0; i < N; i++) m[i] = P - BigNumber((unsigned int)(i * 1024));
  std::vector<BigNumber> c, d;
  for (size_t done = 0; done < 4096 + N; done += N) c = pk.encrypt(ipcl::PlainText(m)).getTexts();
  d = sk.decrypt(ipcl::CipherText(pk, c)).getTexts();
  std::printf("threads %d\n", ipcl::detail::max_host_threads());
  std::printf("%-44s %9.1f us\n", "encrypt total (vector in, vector out)", best_us([&] { c = pk.encrypt(ipcl::PlainText(m)).getTexts(); }));
  std::printf("%-44s %9.1f us\n", "decrypt total (vector in, vector out)", best_us([&] { d = sk.decrypt(ipcl::CipherText(pk, c)).getTexts(); }));
  {
    ipcl::PlainText pt(m);
    ipcl::CipherText ct = pk.encrypt(pt);
    std::printf("%-44s %9.1f us\n", "  PlainText(m) ctor", best_us([&] { ipcl::PlainText t(m); (void)t.getSize(); }));
    std::printf("%-44s %9.1f us\n", "  pk.encrypt(pt) + getElement(0)", best_us([&] { ipcl::CipherText t = pk.encrypt(pt); (void)t.getElement(0); }));
    std::printf("%-44s %9.1f us\n", "  pk.encrypt(pt) only (async)", best_us([&] { ipcl::CipherText t = pk.encrypt(pt); (void)t.getSize(); pgpu_synchronize(); }));
    std::printf("%-44s %9.1f us\n", "  encrypt(pt).getTexts()", best_us([&] { c = pk.encrypt(pt).getTexts(); }));
    std::printf("%-44s %9.1f us\n", "  CipherText(pk, c) ctor", best_us([&] { ipcl::CipherText t(pk, c); (void)t.getSize(); }));
    ipcl::CipherText cth(pk, c);
    std::printf("%-44s %9.1f us\n", "  sk.decrypt(host ct) + sync", best_us([&] { ipcl::PlainText t = sk.decrypt(cth); (void)t.getSize(); pgpu_synchronize(); }));
    std::printf("%-44s %9.1f us\n", "  sk.decrypt(resident ct) + sync", best_us([&] { ipcl::PlainText t = sk.decrypt(ct); (void)t.getSize(); pgpu_synchronize(); }));
    std::printf("%-44s %9.1f us\n", "  sk.decrypt(resident ct).getTexts()", best_us([&] { d = sk.decrypt(ct).getTexts(); }));
  }
  // the same through the C-ABI on flat arrays
  {
    using namespace ipcl::detail;
    std::vector<uint64_t> fm = pack(m, 32), fr = pack(std::vector<BigNumber>(N, BigNumber(KAT_BENCH_R)), 32), fc(N * 64), fo(N * 32);
    std::printf("%-44s %9.1f us\n", "  pack(m, 32)", best_us([&] { fm = pack(m, 32); }));
    std::printf("%-44s %9.1f us\n", "  unpack(c, 64)", best_us([&] { c = unpack(fc, N, 64); }));
    pgpu_batch *bm = nullptr, *br = nullptr, *bc = nullptr, *bo = nullptr;
    std::printf("%-44s %9.1f us\n", "  pgpu_batch_upload(m) 2 MB", best_us([&] { if (bm) pgpu_batch_destroy(bm); pgpu_batch_upload(fm.data(), N, 32, 32, &bm); }));
    std::printf("%-44s %9.1f us\n", "  pgpu_batch_upload(r) 2 MB", best_us([&] { if (br) pgpu_batch_destroy(br); pgpu_batch_upload(fr.data(), N, 32, 32, &br); }));
    // (pubkey handle is private to the ipcl layer: time the C-ABI batch ops through a key of our own)
    std::vector<uint64_t> nl(32), hsl(64);
    n.toLimbs64(nl.data(), 32);
    BigNumber(KAT_BENCH_HS).toLimbs64(hsl.data(), 64);
    pgpu_pubkey* k = nullptr;
    pgpu_pubkey_create(nl.data(), 32, hsl.data(), &k);
    std::vector<uint64_t> pl(16), ql(16);
    P.toLimbs64(pl.data(), 16);
    Q.toLimbs64(ql.data(), 16);
    pgpu_privkey* s = nullptr;
    pgpu_privkey_create(pl.data(), ql.data(), 16, &s);
    for (int i = 0; i < 3; ++i) { if (bc) pgpu_batch_destroy(bc); pgpu_batch_encrypt(k, bm, br, 2047, &bc); }
    pgpu_synchronize();
    std::printf("%-44s %9.1f us\n", "  pgpu_batch_encrypt + synchronize", best_us([&] { pgpu_batch_destroy(bc); pgpu_batch_encrypt(k, bm, br, 2047, &bc); pgpu_synchronize(); }));
    std::printf("%-44s %9.1f us\n", "  pgpu_batch_download(c) 4 MB", best_us([&] { pgpu_batch_download(bc, fc.data()); }));
    pgpu_batch_decrypt_crt(s, bc, &bo);
    pgpu_synchronize();
    std::printf("%-44s %9.1f us\n", "  pgpu_batch_decrypt_crt + synchronize", best_us([&] { pgpu_batch_destroy(bo); pgpu_batch_decrypt_crt(s, bc, &bo); pgpu_synchronize(); }));
    std::printf("%-44s %9.1f us\n", "  pgpu_batch_download(m) 2 MB", best_us([&] { pgpu_batch_download(bo, fo.data()); }));
    pgpu_batch* bc2 = nullptr;
    std::printf("%-44s %9.1f us\n", "  pgpu_batch_upload(c) 4 MB", best_us([&] { if (bc2) pgpu_batch_destroy(bc2); pgpu_batch_upload(fc.data(), N, 64, 64, &bc2); }));
  }
  return 0;
}

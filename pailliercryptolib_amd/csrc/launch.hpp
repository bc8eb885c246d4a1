// pailliercryptolib_amd -- host-callable launchers of the gfx950 kernels (kernels.hpp).
// The kernels are instantiated per geometry in several translation units (k_modexp.hip compiled
// once per PGPU_PART, k_misc.hip) so that they build in parallel and a change of the host runtime
// does not recompile device code.  Every launcher returns false when the geometry is not compiled.
#ifndef PAILLIERCRYPTOLIB_AMD_CSRC_LAUNCH_HPP_
#define PAILLIERCRYPTOLIB_AMD_CSRC_LAUNCH_HPP_

#include <hip/hip_runtime_api.h>

#include "kargs.hpp"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <mutex>

// Build switch (pailliercryptolib_amd/build.py passes it to every translation unit, device and host alike):
//   PGPU_WITH_4096 (env PGPU_BUILD_4096=1)  the split forms of the 4096-bit key class -- (8,18) fixed-base / modexp / pair
//                  rows, (4,18) and (8,9) CRT decrypt: beyond every BASELINE config and the reference's own 2048-bit cap
//                  (ipcl/keygen.cpp:10), 10-15 minutes of compile time per translation unit.  Off: such keys run the
//                  full-width kernels (Montgomery-form words), as all keys did before round 2.
#ifndef PGPU_WITH_4096
#define PGPU_WITH_4096 0
#endif

namespace pgpu {

// hipFuncAttributeMaxDynamicSharedMemorySize, raised ONCE per (kernel, device): the attribute belongs to the device's copy
// of the function, so a process-wide once-flag (rounds 3-5) served only the first GPU of a pool that launched the kernel --
// every other pool entry would have failed its first whole-CU-claim launch (no multi-GPU box had run it).  Set
// at the FIRST launch of the kernel on that device, whatever the launch asks for, and never changed afterwards: a thread
// that changes the attribute while another thread launches the same function races inside the HIP runtime (seen as a
// segfault with four API threads, round 5).  `done`: one bit per HIP device ordinal, owned by the call site.
inline bool lds_attr_once(const void* fn, int bytes, std::atomic<uint64_t>& done) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  const uint64_t bit = (uint64_t)1 << dev;
  if (done.load(std::memory_order_acquire) & bit) return true;
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  if (done.load(std::memory_order_acquire) & bit) return true;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  done.fetch_or(bit, std::memory_order_release);
  return true;
}
#define PGPU_LDS_ATTR_ONCE(kernel_expr, bytes)                                        \
  ([]() -> bool {                                                                    \
    static std::atomic<uint64_t> done_{0};                                           \
    return ::pgpu::lds_attr_once((const void*)(kernel_expr), (bytes), done_);        \
  }())

// Placement pad (round 6).  The workgroup dispatcher starts every launch on the same CUs: two launches of 32 workgroups side by
// side (two API threads with 512-element batches) land on the same SIMDs and slow each other down although seven eighths of
// the chip idle (rocprofv3 kernel trace: a CT x PT kernel 5.6 ms alone, 6-9.6 ms beside a second one; aggregate 3.6 ms per
// call).  A workgroup that owns more than half of a CU's 160 KB of LDS cannot share its CU, so small launches of the
// multi-lane forms ask for unused dynamic LDS up to 81 KB per workgroup and the dispatcher has to spread them: 2.87 ms per
// call, the two launches fully overlapped; four threads x 700-element encrypt + decrypt 1.85 -> 1.24 ms, four x 1024-element
// CT x PT 2.5-3.6 -> 1.48 ms (profiles/r06_place_pad.txt).  Only launches of at most PGPU_PLACE_PAD workgroups (default 192,
// three quarters of the CUs; 0: never): a launch that puts a workgroup on every CU gains from a neighbour's second wavefront
// on its SIMDs (256: encrypt + decrypt of 2048 elements from two threads 2.8 -> 3.15 ms; 128 and 192 measure alike).  The
// attribute is raised at the kernel's first launch on the device whatever its size (lds_attr_once).  Returns the dynamic LDS
// bytes to launch with.
inline int place_pad_limit() {
  static const int limit = [] {
    const char* e = std::getenv("PGPU_PLACE_PAD");
    return e ? std::max(0, std::atoi(e)) : 192;
  }();
  return limit;
}
inline unsigned place_pad(const void* fn, unsigned blocks, std::atomic<uint64_t>& done, std::atomic<int>& pad_bytes) {
  const int limit = place_pad_limit();
  if (limit == 0) return 0;
  int pad = pad_bytes.load(std::memory_order_acquire);
  if (pad < 0) {
    hipFuncAttributes fa{};
    if (hipFuncGetAttributes(&fa, fn) != hipSuccess) {
      (void)hipGetLastError();
      return 0;
    }
    const int want = 81 * 1024, have = (int)fa.sharedSizeBytes;
    pad = have >= want ? 0 : ((want - have + 1023) / 1024) * 1024;
    pad_bytes.store(pad, std::memory_order_release);
  }
  if (pad == 0 || !lds_attr_once(fn, pad, done)) return 0;
  return blocks <= (unsigned)limit ? (unsigned)pad : 0;
}
#define PGPU_PLACE_PAD(kernel_expr, blocks)                                                      \
  ([&]() -> unsigned {                                                                          \
    static std::atomic<uint64_t> done_{0};                                                      \
    static std::atomic<int> pad_{-1};                                                           \
    return ::pgpu::place_pad((const void*)(kernel_expr), (blocks), done_, pad_);                \
  }())

// modexp_kernel lives in eight translation units (k_modexp.hip, PGPU_PART 0..7); launch_modexp tries each.
// regrows: the kernel form whose multiplier rows come from registers (kernels.hpp: modexp_kernel<GEO, true>).
bool launch_modexp_part0(int G, int K, const ModexpArgs& a, unsigned blocks, hipStream_t s);
bool launch_modexp_part1(int G, int K, const ModexpArgs& a, unsigned blocks, hipStream_t s);
bool launch_modexp_part2(int G, int K, const ModexpArgs& a, unsigned blocks, hipStream_t s);
bool launch_modexp_part3(int G, int K, const ModexpArgs& a, unsigned blocks, hipStream_t s);
bool launch_modexp_reg_part4(int G, int K, const ModexpArgs& a, unsigned blocks, hipStream_t s);
bool launch_modexp_reg_part5(int G, int K, const ModexpArgs& a, unsigned blocks, hipStream_t s);
bool launch_modexp_reg_part6(int G, int K, const ModexpArgs& a, unsigned blocks, hipStream_t s);
bool launch_modexp_reg_part7(int G, int K, const ModexpArgs& a, unsigned blocks, hipStream_t s);
inline bool modexp_has_regrows(int G, int K) { return G == 2 || G == 4 || (G == 16 && K <= 7); }
inline bool launch_modexp(int G, int K, bool regrows, const ModexpArgs& a, unsigned blocks, hipStream_t s) {
  if (regrows)
    return launch_modexp_reg_part4(G, K, a, blocks, s) || launch_modexp_reg_part5(G, K, a, blocks, s) ||
           launch_modexp_reg_part6(G, K, a, blocks, s) || launch_modexp_reg_part7(G, K, a, blocks, s);
  return launch_modexp_part0(G, K, a, blocks, s) || launch_modexp_part1(G, K, a, blocks, s) ||
         launch_modexp_part2(G, K, a, blocks, s) || launch_modexp_part3(G, K, a, blocks, s);
}

// hensel_decrypt_kernel (hensel.hpp; k_hensel.hip compiled once per PGPU_PART 0..2): 2H lanes per ciphertext side, K limbs
// per lane, H*K >= limbs of (prime * 2^37).  Per key class the forms from few lanes (throughput) to many (latency:
// batches that leave SIMDs idle): 1024-bit keys (2,10) (4,5) (8,3); 2048: (2,19) (4,10) (8,5); 3072: (4,14) (8,7);
// 4096: (4,18) (8,9).
inline bool hensel_has(int H, int K) {
  return (H == 2 && (K == 10 || K == 19)) || (H == 4 && (K == 5 || K == 10 || K == 14 || (PGPU_WITH_4096 && K == 18))) ||
         (H == 8 && (K == 3 || K == 5 || K == 7 || (PGPU_WITH_4096 && K == 9)));
}
bool launch_hensel_part0(int H, int K, const HenselArgs& a, unsigned blocks, hipStream_t s);
bool launch_hensel_part1(int H, int K, const HenselArgs& a, unsigned blocks, hipStream_t s);
bool launch_hensel_part2(int H, int K, const HenselArgs& a, unsigned blocks, hipStream_t s);
bool launch_hensel_part7(int H, int K, const HenselArgs& a, unsigned blocks, hipStream_t s);   // (2,19) for two waves per SIMD
// packed: the launch puts more than one wavefront on a SIMD
inline bool launch_hensel(int H, int K, bool packed, const HenselArgs& a, unsigned blocks, hipStream_t s) {
  if (packed && launch_hensel_part7(H, K, a, blocks, s)) return true;
  return launch_hensel_part0(H, K, a, blocks, s) || launch_hensel_part1(H, K, a, blocks, s) ||
         launch_hensel_part2(H, K, a, blocks, s);
}

// split-form fixed-base DJN encrypt (hensel.hpp: hensel_fb_build_kernel / hensel_fb_encrypt_kernel; k_hensel.hip part 3)
// (2,19): 1024-bit keys, (4,18): 2048, (8,14): 3072, (8,18): 4096 (k_hensel.hip parts 10, 3, 4, 22)
inline bool hensel_fb_has(int H, int K) {
  return (H == 2 && K == 19) || (H == 4 && K == 18) || (H == 8 && (K == 14 || (PGPU_WITH_4096 && K == 18)));
}
bool launch_hensel_fb_build_part3(int H, int K, const HenselFbBuildArgs& a, unsigned blocks, hipStream_t s);
bool launch_hensel_fb_build_part4(int H, int K, const HenselFbBuildArgs& a, unsigned blocks, hipStream_t s);
bool launch_hensel_fb_build_part10(int H, int K, const HenselFbBuildArgs& a, unsigned blocks, hipStream_t s);
bool launch_hensel_fb_build_part22(int H, int K, const HenselFbBuildArgs& a, unsigned blocks, hipStream_t s);
bool launch_hensel_fb_encrypt_part22(int H, int K, const HenselFbArgs& a, unsigned blocks, hipStream_t s);
bool launch_hensel_fb_encrypt_part3(int H, int K, const HenselFbArgs& a, unsigned blocks, hipStream_t s);
bool launch_hensel_fb_encrypt_part4(int H, int K, const HenselFbArgs& a, unsigned blocks, hipStream_t s);
bool launch_hensel_fb_encrypt_part10(int H, int K, const HenselFbArgs& a, unsigned blocks, hipStream_t s);
inline bool launch_hensel_fb_build(int H, int K, const HenselFbBuildArgs& a, unsigned blocks, hipStream_t s) {
  return launch_hensel_fb_build_part3(H, K, a, blocks, s) || launch_hensel_fb_build_part4(H, K, a, blocks, s) ||
         launch_hensel_fb_build_part10(H, K, a, blocks, s)
#if PGPU_WITH_4096
         || launch_hensel_fb_build_part22(H, K, a, blocks, s)
#endif
      ;
}
// (8,9): the ENCRYPT kernel only, for batches that leave SIMDs idle under a 2048-bit key -- 16 lanes per element, the
// same 72 limbs per half as (4,18): it reads the table the (4,18) build kernel wrote and writes the same pair rows
bool launch_hensel_fb_encrypt_part14(int H, int K, const HenselFbArgs& a, unsigned blocks, hipStream_t s);
inline bool hensel_fb_encrypt_has(int H, int K) { return hensel_fb_has(H, K) || (H == 8 && K == 9); }
inline bool launch_hensel_fb_encrypt(int H, int K, const HenselFbArgs& a, unsigned blocks, hipStream_t s) {
  return launch_hensel_fb_encrypt_part3(H, K, a, blocks, s) || launch_hensel_fb_encrypt_part4(H, K, a, blocks, s) ||
         launch_hensel_fb_encrypt_part10(H, K, a, blocks, s) || launch_hensel_fb_encrypt_part14(H, K, a, blocks, s)
#if PGPU_WITH_4096
         || launch_hensel_fb_encrypt_part22(H, K, a, blocks, s)
#endif
      ;
}

// split-form generic modexp modulo a square (hensel.hpp: hensel_modexp_kernel; k_hensel.hip parts 5, 6, 8, 9): roots of
// up to 1024 bits -- (2,19) / (4,10) / (8,5), fewest to most lanes --, 2048 bits -- (4,18) / (8,9) --, 3072 bits -- (8,14),
// 4096 bits -- (8,18) (part 23)
inline bool hensel_modexp_has(int H, int K) {
  return (H == 4 && (K == 18 || K == 10)) || (H == 8 && (K == 9 || K == 5 || K == 14 || (PGPU_WITH_4096 && K == 18))) || (H == 2 && K == 19);
}
bool launch_hensel_modexp_part5(int H, int K, const HenselModexpArgs& a, unsigned blocks, hipStream_t s);
bool launch_hensel_modexp_part6(int H, int K, const HenselModexpArgs& a, unsigned blocks, hipStream_t s);
bool launch_hensel_modexp_part8(int H, int K, const HenselModexpArgs& a, unsigned blocks, hipStream_t s);
bool launch_hensel_modexp_part9(int H, int K, const HenselModexpArgs& a, unsigned blocks, hipStream_t s);
bool launch_hensel_modexp_part23(int H, int K, const HenselModexpArgs& a, unsigned blocks, hipStream_t s);
inline bool launch_hensel_modexp(int H, int K, const HenselModexpArgs& a, unsigned blocks, hipStream_t s) {
  return launch_hensel_modexp_part5(H, K, a, blocks, s) || launch_hensel_modexp_part6(H, K, a, blocks, s) ||
         launch_hensel_modexp_part8(H, K, a, blocks, s) || launch_hensel_modexp_part9(H, K, a, blocks, s)
#if PGPU_WITH_4096
         || launch_hensel_modexp_part23(H, K, a, blocks, s)
#endif
      ;
}

// element-wise operations on pair rows (hensel.hpp: pair_ops_kernel; k_hensel.hip parts 11-13): the throughput form of
// each key class -- (2,19): 1024-bit keys, (4,18): 2048, (8,14): 3072, (8,18): 4096 (part 24)
inline bool pair_ops_has(int H, int K) {
  return (H == 2 && K == 19) || (H == 4 && K == 18) || (H == 8 && (K == 14 || (PGPU_WITH_4096 && K == 18)));
}
// (8,9) (part 25): CT + CT / CT + PT of batches that leave SIMDs idle under a 2048-bit key -- 16 lanes per element on
// the same 144-limb rows as (4,18), a product's serial chain 40 % shorter
inline bool pair_ops_alt_has(int H, int K) { return H == 8 && K == 9; }
bool launch_pair_ops_part25(int H, int K, const PairOpsArgs& a, unsigned blocks, hipStream_t s);
bool launch_pair_ops_part11(int H, int K, const PairOpsArgs& a, unsigned blocks, hipStream_t s);
bool launch_pair_ops_part12(int H, int K, const PairOpsArgs& a, unsigned blocks, hipStream_t s);
bool launch_pair_ops_part13(int H, int K, const PairOpsArgs& a, unsigned blocks, hipStream_t s);
bool launch_pair_ops_part24(int H, int K, const PairOpsArgs& a, unsigned blocks, hipStream_t s);
inline bool launch_pair_ops(int H, int K, const PairOpsArgs& a, unsigned blocks, hipStream_t s) {
  return launch_pair_ops_part11(H, K, a, blocks, s) || launch_pair_ops_part12(H, K, a, blocks, s) ||
         launch_pair_ops_part13(H, K, a, blocks, s) || launch_pair_ops_part25(H, K, a, blocks, s)
#if PGPU_WITH_4096
         || launch_pair_ops_part24(H, K, a, blocks, s)
#endif
      ;
}

// CRT decrypt with both halves of a residue in the same lanes (hensel_seq.hpp; k_hensel.hip parts 16, 17): pair-row
// ciphertexts, fixed-window scan, launches of two or more wavefronts per SIMD; (4,14): 3072-bit keys, (2,19): 2048-bit
inline bool hensel_seq_has(int G, int K) { return (G == 4 && K == 14) || (G == 2 && (K == 19 || K == 10)); }   // (2,10): 1024-bit keys (part 29)
// lds_pad: bytes of LDS the workgroup claims beyond what it uses (0: none) -- more than half a CU's LDS keeps a second
// workgroup off the CU, so two half-chip launches on different streams spread over all CUs instead of stacking
bool launch_hensel_seq_part16(int G, int K, const HenselArgs& a, unsigned blocks, hipStream_t s, unsigned lds_pad);
bool launch_hensel_seq_part17(int G, int K, const HenselArgs& a, unsigned blocks, hipStream_t s, unsigned lds_pad);
bool launch_hensel_seq_part29(int G, int K, const HenselArgs& a, unsigned blocks, hipStream_t s, unsigned lds_pad);
bool launch_hensel_seq_w1_part32(int G, int K, const HenselArgs& a, unsigned blocks, hipStream_t s, unsigned lds_pad);
// one_per_simd: the launch claims whole CUs and runs one wavefront per SIMD: the (2,19) form then has a build of its own
// that may use the whole register file (part 32; PGPU_SEQ_W1=0 keeps the 256-register build)
inline bool launch_hensel_seq(int G, int K, const HenselArgs& a, unsigned blocks, hipStream_t s, unsigned lds_pad = 0,
                              bool one_per_simd = false) {
  if (one_per_simd && launch_hensel_seq_w1_part32(G, K, a, blocks, s, lds_pad)) return true;
  return launch_hensel_seq_part16(G, K, a, blocks, s, lds_pad) || launch_hensel_seq_part17(G, K, a, blocks, s, lds_pad) ||
         launch_hensel_seq_part29(G, K, a, blocks, s, lds_pad);
}

// CRT decrypt with a whole exponentiation per lane by product scanning (hensel_ps.hpp; k_hensel.hip part 31): pair-row
// ciphertexts, fixed-window scan, constants in limbs of `lb` bits: K = 38 limbs of 28 bits (2048-bit keys)
// (38, 28): 2048-bit keys (k_hensel.hip part 31); (56, 28): 3072-bit keys (part 33 -- the whole register file, one wavefront per SIMD)
// (19, 29): 1024-bit keys (part 34; a column sums 3 x 19 products of 58 bits)
inline bool hensel_ps_has(int K, int lb) { return ((K == 38 || K == 56) && lb == 28) || (K == 19 && lb == 29); }
bool launch_hensel_ps_part31(int K, int lb, const HenselArgs& a, unsigned blocks, hipStream_t s, unsigned lds_pad);
bool launch_hensel_ps_part33(int K, int lb, const HenselArgs& a, unsigned blocks, hipStream_t s, unsigned lds_pad);
bool launch_hensel_ps_part34(int K, int lb, const HenselArgs& a, unsigned blocks, hipStream_t s, unsigned lds_pad);
inline bool launch_hensel_ps(int K, int lb, const HenselArgs& a, unsigned blocks, hipStream_t s, unsigned lds_pad = 0) {
  return launch_hensel_ps_part31(K, lb, a, blocks, s, lds_pad) || launch_hensel_ps_part33(K, lb, a, blocks, s, lds_pad) ||
         launch_hensel_ps_part34(K, lb, a, blocks, s, lds_pad);
}
// the latency form on the same constants (hensel_wave.hpp; k_hensel.hip part 35): one exponentiation per wavefront, a limb per
// lane, between the one-lane entry and exit of the kernel above; a.table = the pair buffer, 2K 32-bit words per exponentiation
inline bool hensel_wave_has(int K, int lb) { return (K == 38 && lb == 28) || (K == 56 && lb == 28) || (K == 19 && lb == 29); }
inline size_t hensel_wave_pair_words(int K) { return 2 * (size_t)K; }
bool launch_hensel_wave_part35(int K, int lb, bool wide_digits, const HenselArgs& a, hipStream_t s);
inline bool launch_hensel_wave(int K, int lb, bool wide_digits, const HenselArgs& a, hipStream_t s) {
  return launch_hensel_wave_part35(K, lb, wide_digits, a, s);
}
// ... and for the n^2 domain (hensel_wave_n2.hpp): CT x PT of small batches on pair rows, one wavefront per element; L2 = limbs
// per half of the rows (72: 2048-bit keys, 112: 3072, 38: 1024; k_hensel.hip parts 36 / 37); 32-bit quotient digits only: the
// caller checks that the rows' radix leaves room (R >= 2^10 P)
inline bool hensel_modexp_wave_has(int L2) { return L2 == 72 || L2 == 112 || L2 == 38; }
bool launch_hensel_modexp_wave_part36(int L2, const HenselModexpArgs& a, hipStream_t s);
bool launch_hensel_modexp_wave_part37(int L2, const HenselModexpArgs& a, hipStream_t s);
inline bool launch_hensel_modexp_wave(int L2, const HenselModexpArgs& a, hipStream_t s) {
  return launch_hensel_modexp_wave_part36(L2, a, s) || launch_hensel_modexp_wave_part37(L2, a, s);
}
// ... and DJN encrypt of small batches onto pair rows (hensel_fb_encrypt_wave_kernel), the same L2
bool launch_hensel_fb_encrypt_wave_part36(int L2, const HenselFbArgs& a, hipStream_t s);
bool launch_hensel_fb_encrypt_wave_part37(int L2, const HenselFbArgs& a, hipStream_t s);
inline bool launch_hensel_fb_encrypt_wave(int L2, const HenselFbArgs& a, hipStream_t s) {
  return launch_hensel_fb_encrypt_wave_part36(L2, a, s) || launch_hensel_fb_encrypt_wave_part37(L2, a, s);
}
// 32-bit words of window table per wavefront (hensel_ps.hpp: ps_table_words -- per entry two parts of ceil(K/4) 16-byte rows of 64 lanes)
inline size_t hensel_ps_table_words(int K, size_t entries) { return entries * 2 * (size_t)((K + 3) / 4) * 64 * 4; }

// per-element bases modulo n^2 in the same form (k_hensel.hip part 18): resident pair rows in and out, fixed window
// (4,18): 2048-bit keys, (8,14): 3072 (part 26), (2,19): 1024 (part 28)
inline bool hensel_modexp_seq_has(int G, int K) { return (G == 4 && K == 18) || (G == 8 && K == 14) || (G == 2 && K == 19); }
bool launch_hensel_modexp_seq_part18(int G, int K, const HenselModexpArgs& a, unsigned blocks, hipStream_t s);
bool launch_hensel_modexp_seq_part26(int G, int K, const HenselModexpArgs& a, unsigned blocks, hipStream_t s);
bool launch_hensel_modexp_seq_part28(int G, int K, const HenselModexpArgs& a, unsigned blocks, hipStream_t s);
inline bool launch_hensel_modexp_seq(int G, int K, const HenselModexpArgs& a, unsigned blocks, hipStream_t s) {
  return launch_hensel_modexp_seq_part18(G, K, a, blocks, s) || launch_hensel_modexp_seq_part26(G, K, a, blocks, s) ||
         launch_hensel_modexp_seq_part28(G, K, a, blocks, s);
}

// CT + CT on pair rows in the same form (k_hensel.hip part 19)
inline bool pair_mul_seq_has(int G, int K) { return (G == 4 && K == 18) || (G == 8 && K == 14) || (G == 2 && K == 19); }
bool launch_pair_mul_seq_part19(int G, int K, const PairOpsArgs& a, unsigned blocks, hipStream_t s);
bool launch_pair_mul_seq_part27(int G, int K, const PairOpsArgs& a, unsigned blocks, hipStream_t s);
inline bool launch_pair_mul_seq(int G, int K, const PairOpsArgs& a, unsigned blocks, hipStream_t s) {
  return launch_pair_mul_seq_part19(G, K, a, blocks, s) || launch_pair_mul_seq_part27(G, K, a, blocks, s);
}

// DJN encrypt to pair rows in the same form (k_hensel.hip parts 20, 21, 28): (4,18) 2048-bit keys, (8,14) 3072, (2,19) 1024
inline bool hensel_fb_encrypt_seq_has(int G, int K) { return (G == 4 && K == 18) || (G == 8 && K == 14) || (G == 2 && K == 19); }
bool launch_hensel_fb_encrypt_seq_part28(int G, int K, const HenselFbArgs& a, unsigned blocks, hipStream_t s);
bool launch_hensel_fb_encrypt_seq_part20(int G, int K, const HenselFbArgs& a, unsigned blocks, hipStream_t s, unsigned lds_pad);
bool launch_hensel_fb_encrypt_seq_part21(int G, int K, const HenselFbArgs& a, unsigned blocks, hipStream_t s);
// lds_pad: whole-CU claim (see launch_hensel_seq); honoured by the (4,18) form, the one the adaptive policy uses part-chip.
// kLdsTotalFlag | bytes: the workgroup owns `bytes` of LDS in all (its static part included)
constexpr unsigned kLdsTotalFlag = 0x80000000u;
inline bool launch_hensel_fb_encrypt_seq(int G, int K, const HenselFbArgs& a, unsigned blocks, hipStream_t s, unsigned lds_pad = 0) {
  return launch_hensel_fb_encrypt_seq_part20(G, K, a, blocks, s, lds_pad) || launch_hensel_fb_encrypt_seq_part21(G, K, a, blocks, s) ||
         launch_hensel_fb_encrypt_seq_part28(G, K, a, blocks, s);
}

bool launch_modmul(int G, int K, const ModmulArgs& a, unsigned blocks, hipStream_t s);
bool launch_crt(int G, int K, const CrtArgs& a, unsigned blocks, hipStream_t s, unsigned lds_total = 0);
bool launch_fb_build(int G, int K, const FixedBaseBuildArgs& a, unsigned blocks, hipStream_t s);
bool launch_fb_encrypt(int G, int K, const FixedBaseArgs& a, unsigned blocks, hipStream_t s);

}  // namespace pgpu

#endif  // PAILLIERCRYPTOLIB_AMD_CSRC_LAUNCH_HPP_

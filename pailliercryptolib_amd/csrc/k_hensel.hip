// pailliercryptolib_amd -- instantiations of the split-form CRT-decrypt exponentiation (hensel.hpp), split over
// PGPU_PART = 0..13 so that they compile in parallel (3, 4, 10: the fixed-base DJN encrypt; 5, 6, 8, 9: the generic modexp; 7:
// the two-wavefronts-per-SIMD build of the (2,19) decrypt form; 11-13: element-wise operations on pair rows).
#include "hensel_seq.hpp"
#include "launch.hpp"
#if defined(PGPU_PART) && (PGPU_PART == 31 || PGPU_PART == 33 || PGPU_PART == 34)
#include "hensel_ps.hpp"     // whole exponentiations in one lane by product scanning (2048-bit keys; round 5)
#endif
#if defined(PGPU_PART) && PGPU_PART == 35
#include "hensel_wave.hpp"   // one exponentiation per WAVEFRONT, a limb per lane: the latency form of small batches (round 6)
#endif
#if defined(PGPU_PART) && (PGPU_PART == 36 || PGPU_PART == 37)
#include "hensel_wave_n2.hpp"   // ... for the n^2 domain: DJN encrypt and CT x PT of small batches on pair rows
#endif

#ifndef PGPU_PART
#error "compile with -DPGPU_PART=0..37 (15 and 30 are retired)"
#endif

namespace pgpu {

#define PGPU_HENSEL_ONE(h, k)                                                                             \
  if (H == h && K == k) {                                                                                 \
    hipLaunchKernelGGL((hensel_decrypt_kernel<h, k>), dim3(blocks), dim3(kWGThreads), PGPU_PLACE_PAD((hensel_decrypt_kernel<h, k>), blocks), s, a);           \
    return true;                                                                                          \
  }

#if PGPU_PART == 0
bool launch_hensel_part0(int H, int K, const HenselArgs& a, unsigned blocks, hipStream_t s) {
  PGPU_HENSEL_ONE(2, 19) PGPU_HENSEL_ONE(2, 10) PGPU_HENSEL_ONE(4, 5)
  return false;
}
#elif PGPU_PART == 1
bool launch_hensel_part1(int H, int K, const HenselArgs& a, unsigned blocks, hipStream_t s) {
#if PGPU_WITH_4096
  PGPU_HENSEL_ONE(4, 18)
#endif
  PGPU_HENSEL_ONE(4, 14) PGPU_HENSEL_ONE(4, 10)
  return false;
}
#elif PGPU_PART == 3 || PGPU_PART == 4 || PGPU_PART == 10 || PGPU_PART == 22
#if PGPU_PART == 22
#define PGPU_FB_H 8
#define PGPU_FB_K 18
#define PGPU_FB_NAME(f) f##_part22
#elif PGPU_PART == 3
#define PGPU_FB_H 4
#define PGPU_FB_K 18
#define PGPU_FB_NAME(f) f##_part3
#elif PGPU_PART == 4
#define PGPU_FB_H 8
#define PGPU_FB_K 14
#define PGPU_FB_NAME(f) f##_part4
#else
#define PGPU_FB_H 2
#define PGPU_FB_K 19
#define PGPU_FB_NAME(f) f##_part10
#endif
bool PGPU_FB_NAME(launch_hensel_fb_build)(int H, int K, const HenselFbBuildArgs& a, unsigned blocks, hipStream_t s) {
  if (H == PGPU_FB_H && K == PGPU_FB_K) {
    hipLaunchKernelGGL((hensel_fb_build_kernel<PGPU_FB_H, PGPU_FB_K>), dim3(blocks), dim3(kWGThreads), 0, s, a);
    return true;
  }
  return false;
}
bool PGPU_FB_NAME(launch_hensel_fb_encrypt)(int H, int K, const HenselFbArgs& a, unsigned blocks, hipStream_t s) {
  if (H == PGPU_FB_H && K == PGPU_FB_K) {
    hipLaunchKernelGGL((hensel_fb_encrypt_kernel<PGPU_FB_H, PGPU_FB_K>), dim3(blocks), dim3(kWGThreads), PGPU_PLACE_PAD((hensel_fb_encrypt_kernel<PGPU_FB_H, PGPU_FB_K>), blocks), s, a);
    return true;
  }
  return false;
}
#elif PGPU_PART == 11 || PGPU_PART == 12 || PGPU_PART == 13 || PGPU_PART == 24 || PGPU_PART == 25
#if PGPU_PART == 25
#define PGPU_PO_H 8
#define PGPU_PO_K 9
#define PGPU_PO_NAME launch_pair_ops_part25
#elif PGPU_PART == 24
#define PGPU_PO_H 8
#define PGPU_PO_K 18
#define PGPU_PO_NAME launch_pair_ops_part24
#elif PGPU_PART == 11
#define PGPU_PO_H 4
#define PGPU_PO_K 18
#define PGPU_PO_NAME launch_pair_ops_part11
#elif PGPU_PART == 12
#define PGPU_PO_H 2
#define PGPU_PO_K 19
#define PGPU_PO_NAME launch_pair_ops_part12
#else
#define PGPU_PO_H 8
#define PGPU_PO_K 14
#define PGPU_PO_NAME launch_pair_ops_part13
#endif
bool PGPU_PO_NAME(int H, int K, const PairOpsArgs& a, unsigned blocks, hipStream_t s) {
  if (H == PGPU_PO_H && K == PGPU_PO_K) {
    hipLaunchKernelGGL((pair_ops_kernel<PGPU_PO_H, PGPU_PO_K>), dim3(blocks), dim3(kWGThreads), 0, s, a);
    return true;
  }
  return false;
}
#elif PGPU_PART == 16
bool launch_hensel_seq_part16(int G, int K, const HenselArgs& a, unsigned blocks, hipStream_t s, unsigned lds_pad) {
  if (G == 4 && K == 14) {
    // (set at the FIRST launch of the kernel, whatever that launch asks for: a thread that changes the attribute while
    // another thread launches the same function races inside the HIP runtime -- seen as a segfault with four API threads)
    const bool once = PGPU_LDS_ATTR_ONCE((hensel_decrypt_seq_kernel<4, 14>), 128 * 1024);
    if (lds_pad && !once) return false;
    hipLaunchKernelGGL((hensel_decrypt_seq_kernel<4, 14>), dim3(blocks), dim3(kWGThreads), lds_pad, s, a);
    return true;
  }
  return false;
}
#elif PGPU_PART == 17
bool launch_hensel_seq_part17(int G, int K, const HenselArgs& a, unsigned blocks, hipStream_t s, unsigned lds_pad) {
  if (G == 2 && K == 19) {
    // (set at the FIRST launch of the kernel, whatever that launch asks for: a thread that changes the attribute while
    // another thread launches the same function races inside the HIP runtime -- seen as a segfault with four API threads)
    const bool once = PGPU_LDS_ATTR_ONCE((hensel_decrypt_seq_kernel<2, 19>), 128 * 1024);
    if (lds_pad && !once) return false;
    hipLaunchKernelGGL((hensel_decrypt_seq_kernel<2, 19>), dim3(blocks), dim3(kWGThreads), lds_pad, s, a);
    return true;
  }
  return false;
}
#elif PGPU_PART == 32
// the one-wavefront-per-SIMD build of the (2,19) sequential-halves decrypt: launches under a CU claim only
bool launch_hensel_seq_w1_part32(int G, int K, const HenselArgs& a, unsigned blocks, hipStream_t s, unsigned lds_pad) {
  if (G == 2 && K == 19 && lds_pad) {
    const bool once = PGPU_LDS_ATTR_ONCE((hensel_decrypt_seq_kernel<2, 19, 1>), 128 * 1024);
    if (!once) return false;
    hipLaunchKernelGGL((hensel_decrypt_seq_kernel<2, 19, 1>), dim3(blocks), dim3(kWGThreads), lds_pad, s, a);
    return true;
  }
  return false;
}
#elif PGPU_PART == 18
bool launch_hensel_modexp_seq_part18(int G, int K, const HenselModexpArgs& a, unsigned blocks, hipStream_t s) {
  if (G == 4 && K == 18) {
    hipLaunchKernelGGL((hensel_modexp_seq_kernel<4, 18>), dim3(blocks), dim3(kWGThreads), 0, s, a);
    return true;
  }
  return false;
}
#elif PGPU_PART == 19
bool launch_pair_mul_seq_part19(int G, int K, const PairOpsArgs& a, unsigned blocks, hipStream_t s) {
  if (G == 4 && K == 18) {
    hipLaunchKernelGGL((pair_mul_seq_kernel<4, 18>), dim3(blocks), dim3(kWGThreads), 0, s, a);
    return true;
  }
  return false;
}
#elif PGPU_PART == 20
bool launch_hensel_fb_encrypt_seq_part20(int G, int K, const HenselFbArgs& a, unsigned blocks, hipStream_t s, unsigned lds_pad) {
  if (G == 4 && K == 18) {
    // lds_pad: whole-CU claim of a part-chip launch beside busy neighbour lanes, as launch_hensel_seq.  With
    // kLdsTotalFlag set the rest of the value is what the workgroup shall own IN ALL (its own LDS included): the claim
    // that lets two of these workgroups share a CU but keeps them off the CUs of a neighbour lane's decrypt
    unsigned dyn = lds_pad;
    static const unsigned own = [] {
      hipFuncAttributes fa{};
      return hipFuncGetAttributes(&fa, (const void*)hensel_fb_encrypt_seq_kernel<4, 18>) == hipSuccess ? (unsigned)fa.sharedSizeBytes : 0u;
    }();
    if (lds_pad & kLdsTotalFlag) {
      const unsigned total = lds_pad & ~kLdsTotalFlag;
      dyn = own && total > own ? total - own : 0;
    }
    // (the kernel's own ~57 KB of LDS + the 84 000-byte claim fit a CU's 160 KB once, not twice; a 128 KB allowance
    // on top of the static part would exceed the CU and the attribute call fails.  Set at the first launch, whatever it
    // asks for: see launch_hensel_seq_part16)
    const bool once = PGPU_LDS_ATTR_ONCE((hensel_fb_encrypt_seq_kernel<4, 18>), 96 * 1024);
    const bool once1 = PGPU_LDS_ATTR_ONCE((hensel_fb_encrypt_seq_kernel<4, 18, 1>), 96 * 1024);
    if (dyn && (!once || !once1)) return false;
    // a claim of more than half a CU's LDS (beside ONE busy lane) means one workgroup per CU, one wavefront per SIMD: the
    // build that may use the whole register file.  The 80 000-byte claim of the quarter-chip mode puts TWO workgroups on a
    // CU: the 256-register build.  PGPU_SEQ_W1=0 keeps the 256-register build everywhere
    static const bool w1 = [] { const char* e = getenv("PGPU_SEQ_W1"); return !e || atoi(e) != 0; }();
    if (w1 && dyn + own > 82000u && !(lds_pad & kLdsTotalFlag))
      hipLaunchKernelGGL((hensel_fb_encrypt_seq_kernel<4, 18, 1>), dim3(blocks), dim3(kWGThreads), dyn, s, a);
    else
      hipLaunchKernelGGL((hensel_fb_encrypt_seq_kernel<4, 18>), dim3(blocks), dim3(kWGThreads), dyn, s, a);
    return true;
  }
  return false;
}
#elif PGPU_PART == 21
bool launch_hensel_fb_encrypt_seq_part21(int G, int K, const HenselFbArgs& a, unsigned blocks, hipStream_t s) {
  if (G == 8 && K == 14) {
    hipLaunchKernelGGL((hensel_fb_encrypt_seq_kernel<8, 14>), dim3(blocks), dim3(kWGThreads), 0, s, a);
    return true;
  }
  return false;
}
#elif PGPU_PART == 26
bool launch_hensel_modexp_seq_part26(int G, int K, const HenselModexpArgs& a, unsigned blocks, hipStream_t s) {
  if (G == 8 && K == 14) {
    hipLaunchKernelGGL((hensel_modexp_seq_kernel<8, 14>), dim3(blocks), dim3(kWGThreads), 0, s, a);
    return true;
  }
  return false;
}
#elif PGPU_PART == 27
bool launch_pair_mul_seq_part27(int G, int K, const PairOpsArgs& a, unsigned blocks, hipStream_t s) {
  if (G == 8 && K == 14) {
    hipLaunchKernelGGL((pair_mul_seq_kernel<8, 14>), dim3(blocks), dim3(kWGThreads), 0, s, a);
    return true;
  }
  if (G == 2 && K == 19) {
    hipLaunchKernelGGL((pair_mul_seq_kernel<2, 19>), dim3(blocks), dim3(kWGThreads), 0, s, a);
    return true;
  }
  return false;
}
#elif PGPU_PART == 28
bool launch_hensel_modexp_seq_part28(int G, int K, const HenselModexpArgs& a, unsigned blocks, hipStream_t s) {
  if (G == 2 && K == 19) {
    hipLaunchKernelGGL((hensel_modexp_seq_kernel<2, 19>), dim3(blocks), dim3(kWGThreads), 0, s, a);
    return true;
  }
  return false;
}
bool launch_hensel_fb_encrypt_seq_part28(int G, int K, const HenselFbArgs& a, unsigned blocks, hipStream_t s) {
  if (G == 2 && K == 19) {
    hipLaunchKernelGGL((hensel_fb_encrypt_seq_kernel<2, 19>), dim3(blocks), dim3(kWGThreads), 0, s, a);
    return true;
  }
  return false;
}
#elif PGPU_PART == 29
bool launch_hensel_seq_part29(int G, int K, const HenselArgs& a, unsigned blocks, hipStream_t s, unsigned lds_pad) {
  if (G == 2 && K == 10) {
    // (set at the FIRST launch of the kernel, whatever that launch asks for: a thread that changes the attribute while
    // another thread launches the same function races inside the HIP runtime -- seen as a segfault with four API threads)
    const bool once = PGPU_LDS_ATTR_ONCE((hensel_decrypt_seq_kernel<2, 10>), 128 * 1024);
    if (lds_pad && !once) return false;
    hipLaunchKernelGGL((hensel_decrypt_seq_kernel<2, 10>), dim3(blocks), dim3(kWGThreads), lds_pad, s, a);
    return true;
  }
  return false;
}
#elif PGPU_PART == 31
// lds_pad: whole-CU claim (launch_hensel_seq); one_per_simd: the launch runs one wavefront per SIMD by construction
bool launch_hensel_ps_part31(int K, int lb, const HenselArgs& a, unsigned blocks, hipStream_t s, unsigned lds_pad) {
  if (K == 38 && lb == 28) {
    // lds_pad asks for that many bytes of LDS per workgroup in all (more than half a CU's: one workgroup per CU); the
    // kernel's own parking area (40 KB) counts towards it
    constexpr unsigned kStatic = sizeof(uint4) * kWavesPerWG * ((38 + 3) / 4) * kWave;
    const unsigned dyn = lds_pad > kStatic ? lds_pad - kStatic : 0;
    const bool once = PGPU_LDS_ATTR_ONCE((hensel_decrypt_ps_kernel<38, 28, 2>), 96 * 1024);   // (first launch: see part 16)
    const bool once1 = PGPU_LDS_ATTR_ONCE((hensel_decrypt_ps_kernel<38, 28, 1>), 96 * 1024);
    if (dyn && (!once || !once1)) return false;
    // a launch that claims whole CUs runs ONE wavefront per SIMD by construction: the build that may use the whole register
    // file (no scratch; nothing else fits beside it on the SIMD).  PGPU_PS_W1=0 keeps the 256-register build
    static const bool w1 = [] { const char* e = getenv("PGPU_PS_W1"); return !e || atoi(e) != 0; }();
    if (dyn && w1) hipLaunchKernelGGL((hensel_decrypt_ps_kernel<38, 28, 1>), dim3(blocks), dim3(kWGThreads), dyn, s, a);
    else hipLaunchKernelGGL((hensel_decrypt_ps_kernel<38, 28, 2>), dim3(blocks), dim3(kWGThreads), dyn, s, a);
    return true;
  }
  return false;
}
static_assert(ps_table_words<38>(32) == 32 * 2 * ((38 + 3) / 4) * 64 * 4, "launch.hpp: hensel_ps_table_words");
#elif PGPU_PART == 33
// 3072-bit keys: 56 limbs of 28 bits per half.  Six 56-limb values live in a pair product: only the build that may use the
// whole register file (one wavefront per SIMD) -- large launches run as rounds of one wavefront per SIMD
bool launch_hensel_ps_part33(int K, int lb, const HenselArgs& a, unsigned blocks, hipStream_t s, unsigned lds_pad) {
  if (K == 56 && lb == 28) {
    constexpr unsigned kStatic = sizeof(uint4) * kWavesPerWG * ((56 + 3) / 4) * kWave;
    const unsigned dyn = lds_pad > kStatic ? lds_pad - kStatic : 0;
    const bool once = PGPU_LDS_ATTR_ONCE((hensel_decrypt_ps_kernel<56, 28, 1>), 96 * 1024);   // (first launch: see part 16)
    if (dyn && !once) return false;
    hipLaunchKernelGGL((hensel_decrypt_ps_kernel<56, 28, 1>), dim3(blocks), dim3(kWGThreads), dyn, s, a);
    return true;
  }
  return false;
}
static_assert(ps_table_words<56>(32) == 32 * 2 * ((56 + 3) / 4) * 64 * 4, "launch.hpp: hensel_ps_table_words");
#elif PGPU_PART == 34
// 1024-bit keys: 19 limbs of 29 bits per half (3 x 19 products of 58 bits fit a 64-bit column); ~150 registers, one build
bool launch_hensel_ps_part34(int K, int lb, const HenselArgs& a, unsigned blocks, hipStream_t s, unsigned lds_pad) {
  if (K == 19 && lb == 29) {
    constexpr unsigned kStatic = sizeof(uint4) * kWavesPerWG * ((19 + 3) / 4) * kWave;
    const unsigned dyn = lds_pad > kStatic ? lds_pad - kStatic : 0;
    const bool once = PGPU_LDS_ATTR_ONCE((hensel_decrypt_ps_kernel<19, 29, 2>), 96 * 1024);   // (first launch: see part 16)
    if (dyn && !once) return false;
    hipLaunchKernelGGL((hensel_decrypt_ps_kernel<19, 29, 2>), dim3(blocks), dim3(kWGThreads), dyn, s, a);
    return true;
  }
  return false;
}
static_assert(ps_table_words<19>(32) == 32 * 2 * ((19 + 3) / 4) * 64 * 4, "launch.hpp: hensel_ps_table_words");
#elif PGPU_PART == 35
// the latency form: entry (one lane per exponentiation) -> wave kernel (one wavefront per exponentiation) -> exit; a.table is
// the pair buffer between them (launch.hpp: hensel_wave_pair_words), the window table is dynamic LDS of the wave kernel
template <int K, int LB>
static bool launch_wave_one(bool wide, const HenselArgs& a, hipStream_t s) {
  const size_t lane_waves = 2 * ((a.count + kWave - 1) / kWave);
  const unsigned lane_blocks = (unsigned)((lane_waves + kWavesPerWG - 1) / kWavesPerWG);
  const unsigned entry_blocks = (unsigned)((lane_waves * 2 * (size_t)a.pchunks + kWavesPerWG - 1) / kWavesPerWG);   // a wavefront per role
  const unsigned wave_blocks = (unsigned)((2 * a.count + kWavesPerWG - 1) / kWavesPerWG);
  const unsigned lds = (unsigned)(kWavesPerWG * wv_table_words<K>((size_t)1 << a.window) * sizeof(uint32_t));
  if (lds > 64 * 1024) return false;
  hipLaunchKernelGGL((hensel_ps_entry_kernel<K, LB>), dim3(entry_blocks), dim3(kWGThreads), 0, s, a);
  // (placement pad, launch.hpp: two callers' launches of this form on different CUs -- the table alone lets two workgroups
  //  share a CU, two wavefronts per SIMD at 1.7x the time while other CUs idle)
  const unsigned claim = 82 * 1024;
  unsigned dyn = lds;
  if (place_pad_limit() > 0) {
    const bool ok = wide ? PGPU_LDS_ATTR_ONCE((hensel_decrypt_wave_kernel<K, LB, true>), (int)claim)
                         : PGPU_LDS_ATTR_ONCE((hensel_decrypt_wave_kernel<K, LB, false>), (int)claim);
    if (ok && wave_blocks <= (unsigned)place_pad_limit()) dyn = claim;
  }
  if (wide) hipLaunchKernelGGL((hensel_decrypt_wave_kernel<K, LB, true>), dim3(wave_blocks), dim3(kWGThreads), dyn, s, a);
  else hipLaunchKernelGGL((hensel_decrypt_wave_kernel<K, LB, false>), dim3(wave_blocks), dim3(kWGThreads), dyn, s, a);
  hipLaunchKernelGGL((hensel_ps_exit_kernel<K, LB>), dim3(lane_blocks), dim3(kWGThreads), 0, s, a);
  return true;
}
// wide: 32-bit quotient digits (hensel_wave.hpp: wv_digit) -- the caller has checked R >= 2^10 P
bool launch_hensel_wave_part35(int K, int lb, bool wide, const HenselArgs& a, hipStream_t s) {
  if (K == 38 && lb == 28) return launch_wave_one<38, 28>(wide, a, s);
  if (K == 56 && lb == 28) return launch_wave_one<56, 28>(wide, a, s);
  if (K == 19 && lb == 29) return launch_wave_one<19, 29>(wide, a, s);
  return false;
}
#elif PGPU_PART == 36 || PGPU_PART == 37
// The latency forms of the n^2 domain (hensel_wave_n2.hpp), one wavefront per element; part 36: 2048-bit keys (rows of 72 limbs
// per half), part 37: 3072- and 1024-bit keys (112, 38).  Only the builds with 32-bit quotient digits: the caller takes these
// forms only where the rows' radix leaves room for them (capi.cpp).
// CT x PT on pair rows; the window table is dynamic LDS (up to 128 KB)
template <int L2, int LPL>
static bool launch_modexp_wave_one(const HenselModexpArgs& a, hipStream_t s) {
  const unsigned blocks = (unsigned)((a.count + kWavesPerWG - 1) / kWavesPerWG);
  const unsigned lds = (unsigned)(kWavesPerWG * wvn_table_words<LPL>((size_t)1 << a.window) * sizeof(uint32_t));
  if (lds > 144 * 1024) return false;
  const bool once = PGPU_LDS_ATTR_ONCE((hensel_modexp_wave_kernel<L2, LPL, true>), 144 * 1024);
  if (!once) return false;
  hipLaunchKernelGGL((hensel_modexp_wave_kernel<L2, LPL, true>), dim3(blocks), dim3(kWGThreads), lds, s, a);
  return true;
}
// DJN encrypt onto pair rows (no LDS)
template <int L2, int LPL>
static bool launch_fb_encrypt_wave_one(const HenselFbArgs& a, hipStream_t s) {
  const unsigned blocks = (unsigned)((a.count + kWavesPerWG - 1) / kWavesPerWG);
  hipLaunchKernelGGL((hensel_fb_encrypt_wave_kernel<L2, LPL, true>), dim3(blocks), dim3(kWGThreads),
                     PGPU_PLACE_PAD((hensel_fb_encrypt_wave_kernel<L2, LPL, true>), blocks), s, a);
  return true;
}
#if PGPU_PART == 36
bool launch_hensel_modexp_wave_part36(int L2, const HenselModexpArgs& a, hipStream_t s) {
  return L2 == 72 && launch_modexp_wave_one<72, 2>(a, s);
}
bool launch_hensel_fb_encrypt_wave_part36(int L2, const HenselFbArgs& a, hipStream_t s) {
  return L2 == 72 && launch_fb_encrypt_wave_one<72, 2>(a, s);
}
#else
bool launch_hensel_modexp_wave_part37(int L2, const HenselModexpArgs& a, hipStream_t s) {
  if (L2 == 112) return launch_modexp_wave_one<112, 2>(a, s);
  if (L2 == 38) return launch_modexp_wave_one<38, 1>(a, s);
  return false;
}
bool launch_hensel_fb_encrypt_wave_part37(int L2, const HenselFbArgs& a, hipStream_t s) {
  if (L2 == 112) return launch_fb_encrypt_wave_one<112, 2>(a, s);
  if (L2 == 38) return launch_fb_encrypt_wave_one<38, 1>(a, s);
  return false;
}
#endif
#elif PGPU_PART == 14
bool launch_hensel_fb_encrypt_part14(int H, int K, const HenselFbArgs& a, unsigned blocks, hipStream_t s) {
  if (H == 8 && K == 9) {
    hipLaunchKernelGGL((hensel_fb_encrypt_kernel<8, 9>), dim3(blocks), dim3(kWGThreads), PGPU_PLACE_PAD((hensel_fb_encrypt_kernel<8, 9>), blocks), s, a);
    return true;
  }
  return false;
}
#elif PGPU_PART == 8
bool launch_hensel_modexp_part8(int H, int K, const HenselModexpArgs& a, unsigned blocks, hipStream_t s) {
  if (H == 2 && K == 19) {
    hipLaunchKernelGGL((hensel_modexp_kernel<2, 19>), dim3(blocks), dim3(kWGThreads), PGPU_PLACE_PAD((hensel_modexp_kernel<2, 19>), blocks), s, a);
    return true;
  }
  if (H == 8 && K == 5) {
    hipLaunchKernelGGL((hensel_modexp_kernel<8, 5>), dim3(blocks), dim3(kWGThreads), PGPU_PLACE_PAD((hensel_modexp_kernel<8, 5>), blocks), s, a);
    return true;
  }
  if (H == 4 && K == 10) {
    hipLaunchKernelGGL((hensel_modexp_kernel<4, 10>), dim3(blocks), dim3(kWGThreads), PGPU_PLACE_PAD((hensel_modexp_kernel<4, 10>), blocks), s, a);
    return true;
  }
  return false;
}
#elif PGPU_PART == 9
bool launch_hensel_modexp_part9(int H, int K, const HenselModexpArgs& a, unsigned blocks, hipStream_t s) {
  if (H == 8 && K == 14) {
    hipLaunchKernelGGL((hensel_modexp_kernel<8, 14>), dim3(blocks), dim3(kWGThreads), PGPU_PLACE_PAD((hensel_modexp_kernel<8, 14>), blocks), s, a);
    return true;
  }
  return false;
}
#elif PGPU_PART == 7
bool launch_hensel_part7(int H, int K, const HenselArgs& a, unsigned blocks, hipStream_t s) {
  if (H == 2 && K == 19) {
    hipLaunchKernelGGL((hensel_decrypt_kernel<2, 19, 2>), dim3(blocks), dim3(kWGThreads), PGPU_PLACE_PAD((hensel_decrypt_kernel<2, 19, 2>), blocks), s, a);
    return true;
  }
  return false;
}
#elif PGPU_PART == 5
bool launch_hensel_modexp_part5(int H, int K, const HenselModexpArgs& a, unsigned blocks, hipStream_t s) {
  if (H == 4 && K == 18) {
    hipLaunchKernelGGL((hensel_modexp_kernel<4, 18>), dim3(blocks), dim3(kWGThreads), PGPU_PLACE_PAD((hensel_modexp_kernel<4, 18>), blocks), s, a);
    return true;
  }
  return false;
}
#elif PGPU_PART == 23
bool launch_hensel_modexp_part23(int H, int K, const HenselModexpArgs& a, unsigned blocks, hipStream_t s) {
  if (H == 8 && K == 18) {
    hipLaunchKernelGGL((hensel_modexp_kernel<8, 18>), dim3(blocks), dim3(kWGThreads), PGPU_PLACE_PAD((hensel_modexp_kernel<8, 18>), blocks), s, a);
    return true;
  }
  return false;
}
#elif PGPU_PART == 6
bool launch_hensel_modexp_part6(int H, int K, const HenselModexpArgs& a, unsigned blocks, hipStream_t s) {
  if (H == 8 && K == 9) {
    hipLaunchKernelGGL((hensel_modexp_kernel<8, 9>), dim3(blocks), dim3(kWGThreads), PGPU_PLACE_PAD((hensel_modexp_kernel<8, 9>), blocks), s, a);
    return true;
  }
  return false;
}
#else
bool launch_hensel_part2(int H, int K, const HenselArgs& a, unsigned blocks, hipStream_t s) {
  PGPU_HENSEL_ONE(8, 3) PGPU_HENSEL_ONE(8, 5) PGPU_HENSEL_ONE(8, 7)
#if PGPU_WITH_4096
  PGPU_HENSEL_ONE(8, 9)
#endif
  return false;
}
#endif

}  // namespace pgpu

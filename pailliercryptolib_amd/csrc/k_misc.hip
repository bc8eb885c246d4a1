// pailliercryptolib_amd -- instantiations of the modmul / crt / fixed-base kernels for every throughput geometry.
#include "kernels.hpp"
#include "launch.hpp"

namespace pgpu {

#define PGPU_FOR_GEOS(X) X(2, 18) X(4, 18) X(8, 18) X(16, 18) X(2, 9) X(4, 9) X(8, 9) X(16, 9) X(4, 10) \
  X(4, 14) X(8, 14) X(16, 14)

#define PGPU_ONE(KERNEL, g, k)                                                                      \
  if (G == g && K == k) {                                                                           \
    hipLaunchKernelGGL((KERNEL<Geo<g, k>>), dim3(blocks), dim3(kWGThreads), 0, s, a);               \
    return true;                                                                                    \
  }

#define PGPU_ONE_MODMUL(g, k) PGPU_ONE(modmul_kernel, g, k)
bool launch_modmul(int G, int K, const ModmulArgs& a, unsigned blocks, hipStream_t s) {
  PGPU_FOR_GEOS(PGPU_ONE_MODMUL)
  return false;
}
// lds_total: bytes of LDS a workgroup shall own in all (0: what it needs) -- the claim that keeps the recombination of a
// batch lane on the CUs its decrypt has just left instead of squeezed beside the neighbour lanes' decrypt wavefronts
// (capi.cpp: decrypt_on; launch_hensel_fb_encrypt_seq has the same claim)
#define PGPU_ONE_CRT(g, k)                                                                                        \
  if (G == g && K == k) {                                                                                         \
    /* (both at the FIRST launch of the instantiation, whatever it asks for: changing a function's attributes while  \
       another thread launches it races inside the HIP runtime) */                                                  \
    static const unsigned own = [] {                                                                              \
      hipFuncAttributes fa{};                                                                                     \
      return hipFuncGetAttributes(&fa, (const void*)crt_kernel<Geo<g, k>>) == hipSuccess ? (unsigned)fa.sharedSizeBytes : ~0u; \
    }();                                                                                                          \
    const bool once = PGPU_LDS_ATTR_ONCE((crt_kernel<Geo<g, k>>), 96 * 1024);                                      \
    unsigned dyn = 0;                                                                                             \
    if (lds_total && own != ~0u && once && lds_total > own) dyn = lds_total - own;                  \
    hipLaunchKernelGGL((crt_kernel<Geo<g, k>>), dim3(blocks), dim3(kWGThreads), dyn, s, a);                       \
    return true;                                                                                                  \
  }
bool launch_crt(int G, int K, const CrtArgs& a, unsigned blocks, hipStream_t s, unsigned lds_total) {
  PGPU_FOR_GEOS(PGPU_ONE_CRT)
  return false;
}
#define PGPU_ONE_FBB(g, k) PGPU_ONE(fb_build_kernel, g, k)
bool launch_fb_build(int G, int K, const FixedBaseBuildArgs& a, unsigned blocks, hipStream_t s) {
  PGPU_FOR_GEOS(PGPU_ONE_FBB)
  return false;
}
#define PGPU_ONE_FBE(g, k) PGPU_ONE(fb_encrypt_kernel, g, k)
bool launch_fb_encrypt(int G, int K, const FixedBaseArgs& a, unsigned blocks, hipStream_t s) {
  PGPU_FOR_GEOS(PGPU_ONE_FBE)
  return false;
}

}  // namespace pgpu

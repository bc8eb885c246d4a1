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
#define PGPU_ONE_CRT(g, k) PGPU_ONE(crt_kernel, g, k)
bool launch_crt(int G, int K, const CrtArgs& a, unsigned blocks, hipStream_t s) {
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

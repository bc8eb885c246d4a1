// pailliercryptolib_amd -- instantiations of modexp_kernel, split over PGPU_PART = 0..3 so that the
// geometries compile in parallel (build.py compiles this file once per part).
#include "kernels.hpp"
#include "launch.hpp"

#ifndef PGPU_PART
#error "compile with -DPGPU_PART=0..3"
#endif

namespace pgpu {

#define PGPU_TRY_GEO(g, k)                                                                          \
  if (G == g && K == k) {                                                                           \
    hipLaunchKernelGGL((modexp_kernel<Geo<g, k>>), dim3(blocks), dim3(kWGThreads), 0, s, a);        \
    return true;                                                                                    \
  }

#if PGPU_PART == 0
bool launch_modexp_part0(int G, int K, const ModexpArgs& a, unsigned blocks, hipStream_t s) {
  PGPU_TRY_GEO(2, 18) PGPU_TRY_GEO(4, 18)
  return false;
}
#elif PGPU_PART == 1
bool launch_modexp_part1(int G, int K, const ModexpArgs& a, unsigned blocks, hipStream_t s) {
  PGPU_TRY_GEO(8, 18) PGPU_TRY_GEO(16, 18)
  return false;
}
#elif PGPU_PART == 2
bool launch_modexp_part2(int G, int K, const ModexpArgs& a, unsigned blocks, hipStream_t s) {
  PGPU_TRY_GEO(4, 14) PGPU_TRY_GEO(8, 14) PGPU_TRY_GEO(16, 14) PGPU_TRY_GEO(16, 7)
  return false;
}
#else
bool launch_modexp_part3(int G, int K, const ModexpArgs& a, unsigned blocks, hipStream_t s) {
  PGPU_TRY_GEO(2, 9) PGPU_TRY_GEO(4, 9) PGPU_TRY_GEO(8, 9) PGPU_TRY_GEO(16, 9) PGPU_TRY_GEO(16, 5) PGPU_TRY_GEO(4, 10)
  return false;
}
#endif

}  // namespace pgpu

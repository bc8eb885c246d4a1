// pailliercryptolib_amd -- instantiations of modexp_kernel, split over PGPU_PART = 0..7 so that the
// geometries compile in parallel (build.py compiles this file once per part).  Parts 0-3: multiplier rows in LDS
// (launches with several wavefronts per SIMD); parts 4-7: rows broadcast from registers (REGROWS, launches that
// leave a wavefront alone on its SIMD; one-instruction broadcasts exist for 2-, 4- and 16-lane groups; of the
// 16-lane splits only the two latency geometries are built -- every block of a REGROWS multiplication is inline, and
// 16 blocks of 9+ limbs compile for tens of minutes).
#include "kernels.hpp"
#include "launch.hpp"

#ifndef PGPU_PART
#error "compile with -DPGPU_PART=0..7"
#endif

namespace pgpu {

#define PGPU_TRY_GEO(g, k, regrows)                                                                        \
  if (G == g && K == k) {                                                                                  \
    hipLaunchKernelGGL((modexp_kernel<Geo<g, k>, regrows>), dim3(blocks), dim3(kWGThreads), 0, s, a);      \
    return true;                                                                                           \
  }

#if PGPU_PART == 0
bool launch_modexp_part0(int G, int K, const ModexpArgs& a, unsigned blocks, hipStream_t s) {
  PGPU_TRY_GEO(2, 18, false) PGPU_TRY_GEO(4, 18, false)
  return false;
}
#elif PGPU_PART == 1
bool launch_modexp_part1(int G, int K, const ModexpArgs& a, unsigned blocks, hipStream_t s) {
  PGPU_TRY_GEO(8, 18, false) PGPU_TRY_GEO(16, 18, false)
  return false;
}
#elif PGPU_PART == 2
bool launch_modexp_part2(int G, int K, const ModexpArgs& a, unsigned blocks, hipStream_t s) {
  PGPU_TRY_GEO(4, 14, false) PGPU_TRY_GEO(8, 14, false) PGPU_TRY_GEO(16, 14, false) PGPU_TRY_GEO(16, 7, false)
  return false;
}
#elif PGPU_PART == 3
bool launch_modexp_part3(int G, int K, const ModexpArgs& a, unsigned blocks, hipStream_t s) {
  PGPU_TRY_GEO(2, 9, false) PGPU_TRY_GEO(4, 9, false) PGPU_TRY_GEO(8, 9, false) PGPU_TRY_GEO(16, 9, false)
  PGPU_TRY_GEO(16, 5, false) PGPU_TRY_GEO(4, 10, false)
  return false;
}
#elif PGPU_PART == 4
bool launch_modexp_reg_part4(int G, int K, const ModexpArgs& a, unsigned blocks, hipStream_t s) {
  PGPU_TRY_GEO(4, 18, true) PGPU_TRY_GEO(2, 18, true)
  return false;
}
#elif PGPU_PART == 5
bool launch_modexp_reg_part5(int G, int K, const ModexpArgs& a, unsigned blocks, hipStream_t s) {
  PGPU_TRY_GEO(4, 14, true) PGPU_TRY_GEO(16, 7, true)
  return false;
}
#elif PGPU_PART == 6
bool launch_modexp_reg_part6(int G, int K, const ModexpArgs& a, unsigned blocks, hipStream_t s) {
  PGPU_TRY_GEO(16, 5, true)
  return false;
}
#else
bool launch_modexp_reg_part7(int G, int K, const ModexpArgs& a, unsigned blocks, hipStream_t s) {
  PGPU_TRY_GEO(2, 9, true) PGPU_TRY_GEO(4, 9, true) PGPU_TRY_GEO(4, 10, true)
  return false;
}
#endif

}  // namespace pgpu

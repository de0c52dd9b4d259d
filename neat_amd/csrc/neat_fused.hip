// Translation unit of the fused chain kernels (see fused_launch.hpp for why it is separate).
#if defined(NEAT_HALF) && NEAT_HALF      // the f16 twin of this translation unit (see neat_api.hip)
#define neat neat_f16
#endif
#include "kernels_fused.hpp"

namespace neat {

template <int RT> hipError_t launch_rt(hipStream_t st, const FusedArgs& a, int ntiles, int nwg, bool full, bool interleave) {
  typedef F6Cfg<4, RT> C;
  static bool attr_done = false;      // not a stream operation: keep it out of graph capture
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&sdf_fused_w64_kernel<4, false, RT>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&sdf_fused_w64_kernel<4, true, RT>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  if (full) hipLaunchKernelGGL((sdf_fused_w64_kernel<4, false, RT>), dim3(nwg), dim3(C::THREADS), C::LDS, st, a, ntiles, interleave ? -nwg : nwg);
  else hipLaunchKernelGGL((sdf_fused_w64_kernel<4, true, RT>), dim3(nwg), dim3(C::THREADS), C::LDS, st, a, ntiles, interleave ? -nwg : nwg);
  return hipGetLastError();
}

hipError_t launch_ph(hipStream_t st, const FusedArgs& a, int ntiles, int nwg, bool full, bool interleave) {
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&sdf_fused_ph_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, PhCfg::LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&sdf_fused_ph_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, PhCfg::LDS);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  if (full) hipLaunchKernelGGL((sdf_fused_ph_kernel<false>), dim3(nwg), dim3(PHT), PhCfg::LDS, st, a, ntiles, interleave ? -nwg : nwg);
  else hipLaunchKernelGGL((sdf_fused_ph_kernel<true>), dim3(nwg), dim3(PHT), PhCfg::LDS, st, a, ntiles, interleave ? -nwg : nwg);
  return hipGetLastError();
}

hipError_t launch_sdf_fused_w64(hipStream_t st, const FusedArgs& a, int ntiles, int nwg, bool full, bool interleave, int rows_per_wave) {
  if (rows_per_wave == 0) return launch_ph(st, a, ntiles, nwg, full, interleave);
  return rows_per_wave == 64 ? launch_rt<2>(st, a, ntiles, nwg, full, interleave) : launch_rt<1>(st, a, ntiles, nwg, full, interleave);
}

hipError_t launch_sdf_adjoint_w64(hipStream_t st, const AdjArgs& a, int ntiles, int nwg, bool save) {
  typedef F6Cfg<4, 1> C;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&sdf_adjoint_w64_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&sdf_adjoint_w64_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  if (save) hipLaunchKernelGGL((sdf_adjoint_w64_kernel<true>), dim3(nwg), dim3(C::THREADS), C::LDS, st, a, ntiles, nwg);
  else hipLaunchKernelGGL((sdf_adjoint_w64_kernel<false>), dim3(nwg), dim3(C::THREADS), C::LDS, st, a, ntiles, nwg);
  return hipGetLastError();
}

}  // namespace neat

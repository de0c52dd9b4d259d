// Translation unit of the fused chain kernels (see fused_launch.hpp for why it is separate).
#include "kernels_fused.hpp"

namespace neat {

hipError_t launch_sdf_fused_w64(hipStream_t st, const FusedArgs& a, int ntiles, int nwg, bool full, bool interleave) {
  static bool attr_done = false;      // not a stream operation: keep it out of graph capture
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&sdf_fused_w64_kernel<4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, F6Cfg<4>::LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&sdf_fused_w64_kernel<4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, F6Cfg<4>::LDS);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  if (full) hipLaunchKernelGGL((sdf_fused_w64_kernel<4, false>), dim3(nwg), dim3(F6T), F6Cfg<4>::LDS, st, a, ntiles, interleave ? -nwg : nwg);
  else hipLaunchKernelGGL((sdf_fused_w64_kernel<4, true>), dim3(nwg), dim3(F6T), F6Cfg<4>::LDS, st, a, ntiles, interleave ? -nwg : nwg);
  return hipGetLastError();
}

}  // namespace neat

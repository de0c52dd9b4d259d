// Translation unit of the fused chain kernels (see fused_launch.hpp for why it is separate).
#if defined(NEAT_HALF) && NEAT_HALF      // the f16 twin of this translation unit (see neat_api.hip)
#define neat neat_f16
#endif
#include "kernels_fused.hpp"
#include "kernels_x3.hpp"
#include "kernels_heads.hpp"

namespace neat {

template <int RT> hipError_t launch_rt(hipStream_t st, const FusedArgs& a, int ntiles, int nwg, bool full, bool interleave) {
  typedef F6Cfg<4, RT> C;
  static DevOnce attr_done;      // not a stream operation: keep it out of graph capture
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
  static DevOnce attr_done;
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
  static DevOnce attr_done;
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

template <class K, class A> static hipError_t x3_launch(K kern, DevOnce& attr_done, hipStream_t st, int nbatches, int nwg, const A& args) {
  if (!attr_done) {      // not a stream operation: keep it out of graph capture
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, X3::LDS);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  const int grid = nbatches < nwg ? nbatches : nwg;
  if (grid <= 0) return hipSuccess;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(X3::THREADS), X3::LDS, st, args, nbatches);
  return hipGetLastError();
}

hipError_t launch_sdf_chain_x3(hipStream_t st, const FusedArgs& a, int nbatches, int nwg, bool full) {
  static DevOnce d0, d1;
  return full ? x3_launch(&sdf_chain_x3_kernel<false>, d0, st, nbatches, nwg, a) : x3_launch(&sdf_chain_x3_kernel<true>, d1, st, nbatches, nwg, a);
}

hipError_t launch_sdf_adjoint_x3(hipStream_t st, const AdjArgs& a, int nbatches, int nwg, bool save) {
  static DevOnce d0, d1;
  return save ? x3_launch(&sdf_adjoint_x3_kernel<true>, d0, st, nbatches, nwg, a) : x3_launch(&sdf_adjoint_x3_kernel<false>, d1, st, nbatches, nwg, a);
}

hipError_t launch_head_chain_x3(hipStream_t st, const HeadX3Args& a, int head, int nbatches, int nwg, bool save) {
  static DevOnce d[4];
  if (head == 0) return save ? x3_launch(&head_chain_x3_kernel<0, true>, d[0], st, nbatches, nwg, a) : x3_launch(&head_chain_x3_kernel<0, false>, d[1], st, nbatches, nwg, a);
  return save ? x3_launch(&head_chain_x3_kernel<1, true>, d[2], st, nbatches, nwg, a) : x3_launch(&head_chain_x3_kernel<1, false>, d[3], st, nbatches, nwg, a);
}

template <class K, class A> static hipError_t hc_launch(K kern, DevOnce& attr_done, hipStream_t st, int npairs, int nwg, const A& args) {
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, HC::LDS);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  const int nb = (npairs + 1) / 2;
  const int grid = nb < nwg ? nb : nwg;
  if (grid <= 0) return hipSuccess;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(HC::THREADS), HC::LDS, st, args, npairs);
  return hipGetLastError();
}

hipError_t launch_head_chain(hipStream_t st, const HeadX3Args& a, int head, int npairs, int nwg, bool save) {
  static DevOnce d[4];
  if (head == 0) return save ? hc_launch(&head_chain_kernel<0, true>, d[0], st, npairs, nwg, a) : hc_launch(&head_chain_kernel<0, false>, d[1], st, npairs, nwg, a);
  return save ? hc_launch(&head_chain_kernel<1, true>, d[2], st, npairs, nwg, a) : hc_launch(&head_chain_kernel<1, false>, d[3], st, npairs, nwg, a);
}

hipError_t launch_head_bwd_chain(hipStream_t st, const HeadBwdArgs& a, int head, int npairs, int nwg) {
  static DevOnce d[2];
  return head == 0 ? hc_launch(&head_bwd_chain_kernel<0>, d[0], st, npairs, nwg, a) : hc_launch(&head_bwd_chain_kernel<1>, d[1], st, npairs, nwg, a);
}

}  // namespace neat

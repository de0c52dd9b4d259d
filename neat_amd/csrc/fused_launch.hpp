// Host-side launchers of the fused chain kernels (kernels_fused.hpp).  They live in their own translation unit
// (neat_fused.hip), which is compiled with -mllvm -amdgpu-mfma-vgpr-form: these kernels run ONE wave per SIMD with
// 512 registers, keep 256 registers of weight fragments and read every accumulator on the VALU (activation epilogue),
// so the accumulators belong in the VGPR half and the weights in the AGPR half -- hipcc's default puts them the other
// way round and pays one v_accvgpr_read per accumulator element.
#pragma once
#include "bf16_common.hpp"

namespace neat {

// fused SDF primal chain, 4 waves x 64 output rows, 128-point batches.  full: save h_1..h_8, PE, lin8 outputs for backward;
// otherwise only the clamped sdf (sampler).  nwg persistent workgroups over ntiles 32-point tiles; interleave: batches
// interleaved over the workgroups instead of one contiguous range each.
// rows_per_wave: 0 = phase-staggered kernel (sdf_fused_ph_kernel); 64 / 32 = stage-pipelined kernel with four waves (one per
// SIMD) / eight waves (two per SIMD).
hipError_t launch_sdf_fused_w64(hipStream_t st, const FusedArgs& a, int ntiles, int nwg, bool full, bool interleave, int rows_per_wave);

// fused adjoint chain (normals): seed + 8 transposed layers in one launch; save: write u_0 .. u_7 (training)
hipError_t launch_sdf_adjoint_w64(hipStream_t st, const AdjArgs& a, int ntiles, int nwg, bool save);

// split-precision forward chains (kernels_x3.hpp; precision NEAT_F16X3): batches of X3_BATCH points over nwg persistent workgroups
constexpr int X3_BATCH = 64;
hipError_t launch_sdf_chain_x3(hipStream_t st, const FusedArgs& a, int nbatches, int nwg, bool full);
hipError_t launch_sdf_adjoint_x3(hipStream_t st, const AdjArgs& a, int nbatches, int nwg, bool save);
hipError_t launch_head_chain_x3(hipStream_t st, const HeadX3Args& a, int head, int nbatches, int nwg, bool save);

// the heads of the 16-bit builds as fused chains (kernels_heads.hpp): npairs pairs of 32-point tiles over nwg persistent workgroups
constexpr int HC_BATCH = 128;
hipError_t launch_head_chain(hipStream_t st, const HeadX3Args& a, int head, int npairs, int nwg, bool save);
hipError_t launch_head_bwd_chain(hipStream_t st, const HeadBwdArgs& a, int head, int npairs, int nwg);

}  // namespace neat

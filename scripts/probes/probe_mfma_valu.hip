// How many VALU instructions hide under one v_mfma_f32_32x32x16_bf16 when ONE wave runs per SIMD?
// Loop body = 1 MFMA + NV plain VALU (v_fma_f32, NC independent chains) + NT transcendentals (v_exp_f32 / v_log_f32).
// Prints cycles per loop iteration (s_memtime) for a grid of (NV, NT).  Build twice: default (accumulators in AGPRs) and
// with -mllvm -amdgpu-mfma-vgpr-form=1 (accumulators in VGPRs):
//   hipcc --offload-arch=gfx950 -O3 probe_mfma_valu.hip -o probe_mfma_valu_a
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 probe_mfma_valu.hip -o probe_mfma_valu_v
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NM, int NV, int NT, bool READACC>
__global__ __launch_bounds__(256, 1) void k(float* out, long long* cyc, int iters) {
  f32x16 acc[2];
  for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }
  uint4 a = make_uint4(threadIdx.x, 1, 2, 3), b = make_uint4(4, 5, threadIdx.x, 7);
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = 1.0f + 0.001f * (threadIdx.x + i);
  float tr[4] = {1.1f, 1.2f, 1.3f, 1.4f};
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < (NM ? NM : 1); ++m) {      // NM slots of {1 MFMA (accumulators alternate), NV plain VALU, NT transcendentals}
      if (NM) acc[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8*>(&a), *reinterpret_cast<bf16x8*>(&b), acc[m & 1], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        float x = v[j & 7];
        if (READACC && j < 4) x += acc[1][j];
        v[j & 7] = fmaf(x, 1.0001f, 0.5f);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) tr[j & 3] = (j & 1) ? __builtin_amdgcn_logf(tr[j & 3] + 1.0f) : __builtin_amdgcn_exp2f(-tr[j & 3]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += v[i];
  for (int i = 0; i < 4; ++i) s += tr[i];
  for (int i = 0; i < 16; ++i) s += acc[0][i] + acc[1][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NM, int NV, int NT, bool RA = false> void run(float* out, long long* cyc, const char* what) {
  const int iters = 2000;
  hipLaunchKernelGGL((k<NM, NV, NT, RA>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
  hipLaunchKernelGGL((k<NM, NV, NT, RA>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  long long h[256];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < 256; ++i) s += h[i];
  printf("%-14s MFMA %d  VALU %2d  TRANS %d : %7.1f counter ticks / iteration\n", what, NM, NV, NT, s / 256 / iters);
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
  printf("per loop iteration = NM slots of {1 MFMA, NV VALU, NT trans}\n");
  run<0, 0, 0>(out, cyc, "empty loop");
  run<0, 8, 0>(out, cyc, "valu only");  run<0, 16, 0>(out, cyc, "valu only"); run<0, 32, 0>(out, cyc, "valu only");
  run<0, 0, 2>(out, cyc, "trans only"); run<0, 0, 4>(out, cyc, "trans only"); run<0, 6, 2>(out, cyc, "valu+trans");
  run<1, 0, 0>(out, cyc, "1 acc chain"); run<2, 0, 0>(out, cyc, "2 acc"); run<4, 0, 0>(out, cyc, "2 acc x2");
  run<4, 1, 0>(out, cyc, "mix"); run<4, 2, 0>(out, cyc, "mix"); run<4, 3, 0>(out, cyc, "mix"); run<4, 4, 0>(out, cyc, "mix"); run<4, 5, 0>(out, cyc, "mix");
  run<4, 6, 0>(out, cyc, "mix"); run<4, 8, 0>(out, cyc, "mix"); run<4, 10, 0>(out, cyc, "mix");
  run<4, 0, 1>(out, cyc, "mix"); run<4, 0, 2>(out, cyc, "mix"); run<4, 2, 2>(out, cyc, "mix"); run<4, 4, 2>(out, cyc, "mix"); run<4, 5, 2>(out, cyc, "mix");
  run<4, 6, 2>(out, cyc, "mix");
  run<4, 4, 0, true>(out, cyc, "mix, acc read"); run<4, 5, 2, true>(out, cyc, "mix, acc read");
  return 0;
}

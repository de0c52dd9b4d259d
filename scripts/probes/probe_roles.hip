// Do a matrix-only wave and a vector-only wave on the SAME SIMD overlap?  512-thread workgroups, one per CU: waves 0..3 run
// MFMA_ITERS x {8 back-to-back v_mfma_f32_32x32x16_bf16}, waves 4..7 run VALU_ITERS x {VALU body}; reports the time of each role
// alone and of both together (s_memtime of the whole workgroup via wall clock of the kernel).
//   hipcc --offload-arch=gfx950 -O3 probe_roles.hip -o probe_roles
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// MODE bit 0: matrix waves active, bit 1: vector waves active.  VK: 0 = fma only, 1 = softplus-like (fma, exp2, add, log2, max, add, mul),
// 2 = LDS write/read traffic, PAIR: 0 = roles by wave >> 2 (w, w+4 pairs), 1 = roles by wave & 1
template <int VK>
__global__ __launch_bounds__(512, 2) void k(float* out, int mode, int pair, int mi, int vi) {
  __shared__ float lds[8192];
  const int wave = threadIdx.x >> 6;
  const int role = pair ? (wave & 1) : (wave >> 2);
  float res = 0.f;
  if (role == 0) {
    if (mode & 1) {
      f32x16 a0, a1;
      for (int i = 0; i < 16; ++i) { a0[i] = 0.f; a1[i] = 0.f; }
      u32x4 A = {threadIdx.x, 1u, 2u, 3u}, B = {4u, 5u, threadIdx.x, 7u};
      for (int it = 0; it < mi; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8*>(&A), *reinterpret_cast<bf16x8*>(&B), a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8*>(&A), *reinterpret_cast<bf16x8*>(&B), a1, 0, 0, 0);
        }
      }
      for (int i = 0; i < 16; ++i) res += a0[i] + a1[i];
    }
  } else {
    if (mode & 2) {
      float v[8];
      for (int i = 0; i < 8; ++i) v[i] = 0.01f * (threadIdx.x + i);
      for (int it = 0; it < vi; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (VK == 0) { v[i] = fmaf(v[i], 1.0001f, 0.5f); v[i] = fmaf(v[i], 0.9999f, -0.5f); v[i] = fmaf(v[i], 1.0001f, 0.25f); v[i] = fmaf(v[i], 0.9999f, -0.25f); }
          else if (VK == 1) {
            const float u = fmaf(v[i], 144.27f, 0.3f);
            v[i] = (fmaxf(u, 0.f) + __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(-fabsf(u)))) * 0.00693f - 0.01f;
          } else {
            lds[(threadIdx.x * 8 + i) & 8191] = v[i];
            v[i] = lds[(threadIdx.x * 8 + i + 64) & 8191] + 1.0f;
          }
        }
      }
      for (int i = 0; i < 8; ++i) res += v[i];
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = res;
}

template <int VK> float run(float* out, int mode, int pair, int mi, int vi) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<VK>), dim3(256), dim3(512), 0, 0, out, mode, pair, mi, vi);
  (void)hipEventRecord(e0, 0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<VK>), dim3(256), dim3(512), 0, 0, out, mode, pair, mi, vi);
  (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / 5 * 1e3f;
}
int main() {
  float* out; (void)hipMalloc(&out, 256 * 512 * 4);
  const int mi = 4000;      // 32000 MFMAs per matrix wave: ~1 M cycles
  const char* names[3] = {"fma chain", "softplus-like", "lds write+read"};
  for (int vk = 0; vk < 3; ++vk) {
    for (int pair = 0; pair < 2; ++pair) {
      const int vi = vk == 0 ? 7000 : (vk == 1 ? 3500 : 6000);
      float tm, tv, tb;
      if (vk == 0) { tm = run<0>(out, 1, pair, mi, vi); tv = run<0>(out, 2, pair, mi, vi); tb = run<0>(out, 3, pair, mi, vi); }
      else if (vk == 1) { tm = run<1>(out, 1, pair, mi, vi); tv = run<1>(out, 2, pair, mi, vi); tb = run<1>(out, 3, pair, mi, vi); }
      else { tm = run<2>(out, 1, pair, mi, vi); tv = run<2>(out, 2, pair, mi, vi); tb = run<2>(out, 3, pair, mi, vi); }
      printf("%-16s roles by %s: matrix waves alone %7.1f us, vector waves alone %7.1f us, both %7.1f us (sum %7.1f, max %7.1f)\n", names[vk],
             pair ? "wave & 1 " : "wave >> 2", tm, tv, tb, tm + tv, tm > tv ? tm : tv);
    }
  }
  return 0;
}

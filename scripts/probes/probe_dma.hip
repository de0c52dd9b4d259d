// How fast can ONE octet-major array [32][P][16 B] be streamed through an LDS ring by LDS-DMA with the structure of
// layer_kernel_ws (256 persistent 8-wave workgroups, 32-point stages, counted vmcnt + s_barrier per stage)?
// hipcc --offload-arch=gfx950 -O3 probe_dma.hip -o probe_dma
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int P = 133120, WSP = 32;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// MODE 0: DMA ring; 1: same addresses with plain VGPR loads (no LDS); 2: DMA ring without the barrier (vmcnt only);
// 3: ring + barrier + ONE chain of 16 dependent MFMAs per stage (B fragments from LDS); 4: two chains of 8; 5: four chains of 4
template <int NS, int MODE>
__global__ __launch_bounds__(512) void k(const uint4* __restrict__ src, float* __restrict__ sink, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int T = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const unsigned lds_base = (unsigned)(size_t)(lds_ptr)lds;
  unsigned acc = 0;
  auto issue = [&](int tau) {
    const int tile = blockIdx.x + tau * gridDim.x;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int oct = 4 * wave + 2 * i + (lane >> 5);
      const uint4* s2 = src + (size_t)oct * P + (size_t)tile * WSP + (lane & 31);
      if (MODE == 1) { const uint4 v = *s2; acc ^= v.x ^ v.y ^ v.z ^ v.w; continue; }
      const unsigned d2 = __builtin_amdgcn_readfirstlane(lds_base + (tau % NS) * (32 * WSP * 16) + (4 * wave + 2 * i) * (WSP * 16));
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(s2), "s"(d2) : "memory");
    }
  };
  if (MODE == 1) { for (int tau = 0; tau < T; ++tau) issue(tau); if (acc == 0x12345u) sink[tid] = 1.f; return; }
#pragma unroll
  for (int t = 0; t < NS - 1; ++t) if (t < T) issue(t);
  for (int tau = 0; tau < T; ++tau) {
    const int ahead = min(NS - 2, T - 1 - tau);
    // wait until at most 2*ahead DMA instructions of this wave are outstanding
    switch (2 * ahead) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
      case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
    if (MODE != 2 && MODE != 8) asm volatile("s_barrier" ::: "memory");
    if (MODE != 6 && tau + NS - 1 < T) issue(tau + NS - 1);
    if (MODE == 7) {
      float x = __uint_as_float(0x3f800000u + (unsigned)tau + lane);
#pragma unroll
      for (int it = 0; it < 128; ++it) x = fmaf(x, 1.0000001f, 1e-7f);      // ~128 dependent VALU ops
      acc ^= __float_as_uint(x) ^ *reinterpret_cast<const unsigned*>(lds + (tau % NS) * (32 * WSP * 16) + tid * 4);
    } else
    if (MODE >= 3) {
      constexpr int NC = MODE == 3 || MODE == 6 || MODE == 8 || MODE == 9 || MODE == 10 ? 1 : (MODE == 4 ? 2 : 4);
      f32x16 c[NC];
#pragma unroll
      for (int j = 0; j < NC; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) c[j][r] = 0.0f;
      const unsigned char* slot = lds + (tau % NS) * (32 * WSP * 16) + ((lane >> 5) * WSP + (lane & 31)) * 16 + (MODE == 8 ? 4 * wave * (WSP * 16) : 0);
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        uint4 bv = make_uint4(0x3f803f80u, 0x3f803f80u + tau, 0x3f803f80u, 0x3f803f80u);
        if (MODE != 9) bv = *reinterpret_cast<const uint4*>(slot + (MODE == 8 ? (ks & 1) : ks) * (2 * WSP * 16));
        if (MODE == 10) { acc ^= bv.x ^ bv.y ^ bv.z ^ bv.w; continue; }
        uint4 av = make_uint4(0x3f803f80u + ks, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
        c[ks % NC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&av), *reinterpret_cast<const bf16x8*>(&bv), c[ks % NC], 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < NC; ++j) acc ^= __float_as_uint(c[j][0]) ^ __float_as_uint(c[j][7]);
    } else
    acc ^= *reinterpret_cast<const unsigned*>(lds + (tau % NS) * (32 * WSP * 16) + tid * 4);
    if (MODE == 6 && tau + NS - 1 < T) issue(tau + NS - 1);
  }
  if (acc == 0x12345u) sink[tid] = 1.f;
}
template <int NS, int MODE> void run(const uint4* src, float* sink, const char* name) {
  const int ntiles = P / WSP; const size_t lds = (size_t)NS * 32 * WSP * 16;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<NS, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<NS, MODE>), dim3(256), dim3(512), lds, 0, src, sink, ntiles);
  hipEventRecord(e0, 0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<NS, MODE>), dim3(256), dim3(512), lds, 0, src + (size_t)(i % 6) * 32 * P, sink, ntiles);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-40s NS=%d: %6.1f us  %6.0f GB/s\n", name, NS, ms / 20 * 1e3, 32.0 * P * 16 / (ms / 20 * 1e-3) / 1e9);
}
int main() {
  uint4* src; float* sink; hipMalloc(&src, (size_t)6 * 32 * P * 16); hipMemset(src, 1, (size_t)6 * 32 * P * 16); hipMalloc(&sink, 4096);
  run<4, 1>(src, sink, "plain VGPR loads, no ring");
  run<2, 0>(src, sink, "DMA ring + barrier"); run<3, 0>(src, sink, "DMA ring + barrier"); run<4, 0>(src, sink, "DMA ring + barrier");
  run<6, 0>(src, sink, "DMA ring + barrier"); run<8, 0>(src, sink, "DMA ring + barrier");
  run<4, 2>(src, sink, "DMA ring, no barrier"); run<8, 2>(src, sink, "DMA ring, no barrier");
  run<4, 3>(src, sink, "ring + 16 dependent MFMAs / stage"); run<8, 3>(src, sink, "ring + 16 dependent MFMAs / stage");
  run<3, 3>(src, sink, "ring + 16 dependent MFMAs / stage"); run<6, 3>(src, sink, "ring + 16 dependent MFMAs / stage");
  run<4, 6>(src, sink, "ring + MFMAs, issue AFTER compute"); run<4, 7>(src, sink, "ring + 128 dependent VALU ops");
  run<4, 8>(src, sink, "ring + MFMAs on own octets, NO barrier");
  run<4, 9>(src, sink, "ring + 16 MFMAs, operands in registers"); run<4, 10>(src, sink, "ring + 16 ds_read_b128, no MFMA");
  run<4, 4>(src, sink, "ring + 2 chains of 8 MFMAs"); run<4, 5>(src, sink, "ring + 4 chains of 4 MFMAs"); run<8, 5>(src, sink, "ring + 4 chains of 4 MFMAs");
  return 0;
}

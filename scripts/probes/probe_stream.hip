// Practical HBM roof on this box for the access mix of the streaming layer kernels: NR input arrays + NW output arrays of
// 68 MB each (256 rows x 133 120 points x bf16), 16 B per lane, grid-stride.  hipcc --offload-arch=gfx950 -O3 probe_stream.hip -o probe_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int NR, int NW>
__global__ __launch_bounds__(256) void stream_kernel(const uint4* const* in, uint4* const* out, size_t n) {
  const uint4* ip[NR]; uint4* op[NW ? NW : 1];
#pragma unroll
  for (int i = 0; i < NR; ++i) ip[i] = in[i];
#pragma unroll
  for (int i = 0; i < NW; ++i) op[i] = out[i];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < NR; ++r) { const uint4 v = ip[r][i]; acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w; }
#pragma unroll
    for (int w = 0; w < NW; ++w) { acc.x += w; op[w][i] = acc; }
    if (NW == 0 && acc.x == 0x12345679u && acc.y == 0x9abcdef1u) out[0][0] = acc;      // read-only variants: keep the loads alive
  }
}
template <int NR, int NW>
void run(const std::vector<uint4*>& bufs, size_t n, int grid) {
  const uint4** din; uint4** dout;
  hipMalloc(&din, NR * sizeof(void*)); hipMalloc(&dout, (NW ? NW : 1) * sizeof(void*));
  hipMemcpy(din, bufs.data(), NR * sizeof(void*), hipMemcpyHostToDevice);
  hipMemcpy(dout, bufs.data() + NR, (NW ? NW : 1) * sizeof(void*), hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((stream_kernel<NR, NW>), dim3(grid), dim3(256), 0, 0, din, dout, n);
  hipEventRecord(e0, 0);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((stream_kernel<NR, NW>), dim3(grid), dim3(256), 0, 0, din, dout, n);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)(NR + NW) * n * 16.0;
  printf("%dR %dW grid %5d: %7.1f us  %6.0f GB/s\n", NR, NW, grid, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e9);
  hipFree(din); hipFree(dout);
}
int main() {
  const size_t n = (size_t)32 * 133120;          // uint4 elements per array: 256 rows x 133120 points x 2 B / 16
  std::vector<uint4*> bufs(8);
  for (auto& b : bufs) { hipMalloc(&b, n * 16); hipMemset(b, 1, n * 16); }
  for (int grid : {256 * 4, 256 * 8, 256 * 16, 256 * 32}) {
    run<1, 1>(bufs, n, grid); run<2, 1>(bufs, n, grid); run<3, 1>(bufs, n, grid); run<3, 2>(bufs, n, grid); run<2, 2>(bufs, n, grid);
    // read-only mixes (round 5): what the weight-gradient launches, the gather of the in-kernel partials, the row dots and the finish launches are
    run<2, 0>(bufs, n, grid); run<4, 0>(bufs, n, grid); run<7, 0>(bufs, n, grid);
  }
  return 0;
}

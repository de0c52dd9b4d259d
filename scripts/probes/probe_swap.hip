#include <hip/hip_runtime.h>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* o) {
  unsigned a = threadIdx.x, b = threadIdx.x * 3u;
  v2u r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  o[threadIdx.x] = r.x; o[64 + threadIdx.x] = r.y;
}
int main() {
  unsigned* d; hipMalloc(&d, 128 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int i : {0, 1, 31, 32, 33, 63}) printf("lane %d: x=%u y=%u\n", i, h[i], h[64 + i]);
  return 0;
}

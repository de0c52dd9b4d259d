#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void probe(const int* addr_elems, unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  typedef __attribute__((address_space(3))) v4s* lp;
  v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(lds + addr_elems[threadIdx.x]));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)r[j];
}
int main() {
  int h[64]; unsigned short o[256];
  int* d; unsigned short* dout;
  hipMalloc(&d, sizeof(h)); hipMalloc(&dout, sizeof(o));
  for (int mode = 0; mode < 2; ++mode) {
    for (int l = 0; l < 64; ++l) h[l] = mode == 0 ? l * 4 : (l * 37 % 64) * 16 + 4 * (l & 3);   // elements; 8-B aligned
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d, dout);
    hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) printf("lane %2d addr %4d -> %4d %4d %4d %4d\n", l, h[l], o[4*l], o[4*l+1], o[4*l+2], o[4*l+3]);
  }
  return 0;
}

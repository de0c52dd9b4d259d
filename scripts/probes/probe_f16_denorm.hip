// Does v_mfma_f32_32x32x16_f16 flush f16 subnormal inputs, and does v_cvt_f16_f32 produce them?  (precision fp16x3 keeps the
// low plane of a hi/lo split as plain f16: lo ~ 2^-11 |hi| is subnormal for |hi| < 2^-3.)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void probe(const float* av, const float* bv, float* out, unsigned short* cv) {
  const int lane = threadIdx.x;
  h8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)0.0f; b[e] = (_Float16)0.0f; }
  // A[row = lane&31][k = 8*(lane>>5) + e], B[k][col = lane&31]: put a single product on k = 0
  if (lane < 32) { a[0] = (_Float16)av[0]; b[0] = (_Float16)bv[0]; }
  f16v acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  if (lane == 0) { out[0] = acc[0]; cv[0] = __builtin_bit_cast(unsigned short, (_Float16)av[0]); cv[1] = __builtin_bit_cast(unsigned short, (_Float16)bv[0]); }
}
int main() {
  float *da, *db, *dout; unsigned short* dc;
  hipMalloc(&da, 4); hipMalloc(&db, 4); hipMalloc(&dout, 4); hipMalloc(&dc, 4);
  const float as[] = {1.0f, 3.0e-5f, 1.0e-6f, 6.0e-8f, 2.5e-7f, 3.0e-5f};
  const float bs[] = {1.0f, 1.0f, 1.0f, 1.0f, 1024.0f, 3.0e-5f};
  for (int i = 0; i < 6; ++i) {
    float o; unsigned short c[2];
    hipMemcpy(da, &as[i], 4, hipMemcpyHostToDevice); hipMemcpy(db, &bs[i], 4, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(da, db, dout, dc);
    hipMemcpy(&o, dout, 4, hipMemcpyDeviceToHost); hipMemcpy(c, dc, 4, hipMemcpyDeviceToHost);
    printf("a=%.4e b=%.4e  f16 bits a=0x%04x b=0x%04x  mfma=%.6e  exact=%.6e\n", as[i], bs[i], c[0], c[1], o, (double)as[i] * bs[i]);
  }
  return 0;
}

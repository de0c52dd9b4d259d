// Types and device helpers shared by the bf16 kernels (kernels_bf16.hpp) and the fused chains (kernels_fused.hpp, its own
// translation unit): bf16 packing, the branch-free softplus_100, and the argument block of the fused SDF primal chain.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// NEAT_HALF = 1: the same kernels with IEEE half (f16) as the 16-bit storage / MFMA operand type instead of bf16 (precision NEAT_F16,
// BASELINE config 5 "fp16 MFMA with fp32 accumulate").  Only this block knows the format: the conversions and the MFMA opcode; every
// kernel moves the 16-bit values as opaque u16 / packed words.  The names keep their "bf" (bf16x8, f2bf, bf_lo ...): "the 16-bit type
// of this build".  The f16 twin of each translation unit is compiled into its own namespace (see neat_api.hip).
#ifndef NEAT_HALF
#define NEAT_HALF 0
#endif

namespace neat {

// "Done on the CURRENT device?"  The dynamic-LDS limit of a kernel (hipFuncAttributeMaxDynamicSharedMemorySize) is a per-device
// attribute: a process-wide flag would leave a second GPU driven by the same process (or a device switched to after the first
// call) launching 150 KB kernels without it.  Drop-in for the `static bool` of the launch helpers: one bit per device.
struct DevOnce {
  unsigned long long mask = 0;
  static int dev() { int d = 0; if (hipGetDevice(&d) != hipSuccess || d < 0 || d > 63) d = 0; return d; }
  operator bool() const { return ((mask >> dev()) & 1ull) != 0; }
  DevOnce& operator=(bool v) { if (v) mask |= 1ull << dev(); else mask &= ~(1ull << dev()); return *this; }
};

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short u16;

constexpr int BMH = 128;         // point-stride granule of the bf16 build (ldp is a multiple of this)

typedef float f32x2_t __attribute__((ext_vector_type(2)));
#if NEAT_HALF
typedef _Float16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 bf16x2_t __attribute__((ext_vector_type(2)));
#define NEAT_MFMA16 __builtin_amdgcn_mfma_f32_32x32x16_f16
// float -> f16 round-to-nearest-even (v_cvt_f16_f32 x 2 + v_pack_b32_f16); f16 -> float is v_cvt_f32_f16 (op_sel picks the half)
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
  f32x2_t v = {lo, hi};
  bf16x2_t r = __builtin_convertvector(v, bf16x2_t);
  return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ u16 f2bf(float f) { return __builtin_bit_cast(u16, (_Float16)f); }
__device__ __forceinline__ float bf2f(u16 h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ float bf_lo(unsigned w) { return (float)__builtin_bit_cast(bf16x2_t, w).x; }
__device__ __forceinline__ float bf_hi(unsigned w) { return (float)__builtin_bit_cast(bf16x2_t, w).y; }
#else
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
#define NEAT_MFMA16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
// float -> bf16 round-to-nearest-even through the compiler's native conversion (v_cvt_pk_bf16_f32 on gfx950: one
// instruction per PAIR; the hand-rolled add/shift form cost ~5 VALU per value and made the epilogues issue-bound)
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
  f32x2_t v = {lo, hi};
  bf16x2_t r = __builtin_convertvector(v, bf16x2_t);
  return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ u16 f2bf(float f) { return (u16)(pack2(f, 0.0f) & 0xFFFFu); }
__device__ __forceinline__ float bf2f(u16 h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ float bf_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __uint_as_float(w & 0xFFFF0000u); }
#endif

// Packed 16-bit helpers of the fused head chains (kernels_heads.hpp, kernels_x3.hpp).  Both 16-bit formats are sign-magnitude, so on a
// packed pair ReLU is a signed 16-bit max with 0 (v_pk_max_i16: negative values and -0 have the top bit set) and "is positive" is an
// unsigned min with 1 (v_pk_min_u16), one instruction per PAIR each.
typedef short s16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_relu16(unsigned p) {
  const s16x2_t z = {0, 0};
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, p), z));
}
__device__ __forceinline__ unsigned pk_nonzero16(unsigned p) {          // halves -> 0 / 1  (the builtin min is expanded into compares and selects)
  unsigned r;
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(p), "s"(0x00010001u));
  return r;
}
typedef float pk2f_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pk2f_t pk_add_f32(pk2f_t a, pk2f_t b) {       // two fp32 adds per issue slot
  pk2f_t r;
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// ReLU-mask word of one lane and stage (16 accumulator elements = 8 pairs): the forward chains shift the pairs' 0 / 1 halves in,
// pair 0 first (relu_mask_push), and store relu_mask_word: pair i's even / odd element at bits 15 - i / 31 - i.  The backward chain
// walks the pairs in the same order: the two sign bits select the current pair (relu_mask_apply), then the word moves up by one.
__device__ __forceinline__ void relu_mask_push(unsigned& m, unsigned relu_pair) { m = (m << 1) | pk_nonzero16(relu_pair); }
__device__ __forceinline__ unsigned relu_mask_word(unsigned m) { return m << 8; }
__device__ __forceinline__ unsigned relu_mask_apply(unsigned& m, unsigned pair) {
  const s16x2_t sh = {15, 15};
  const unsigned keep = __builtin_bit_cast(unsigned, __builtin_bit_cast(s16x2_t, m) >> sh);      // v_pk_ashrrev_i16: halves -> 0 / 0xFFFF
  m <<= 1;
  return pair & keep;
}

// branch-free activation math for the bf16 build (hardware exp/log; absolute error ~1e-7, far below bf16 resolution)
__device__ __forceinline__ float softplus100_fast(float a) {
  // base-2 form on the raw v_exp_f32 / v_log_f32 (8 VALU ops; `__logf` expands to ~15 with its denormal and ln2 fix-ups):
  // softplus_100(a) = ln2/100 * (max(u,0) + log2(1 + 2^-|u|)),  u = 100 a log2(e);  the log argument lies in [1, 2]
  const float u = a * 144.26950408889634f;
  const float y = __builtin_amdgcn_exp2f(-fabsf(u));
  return (fmaxf(u, 0.0f) + __builtin_amdgcn_logf(1.0f + y)) * 0.0069314718055994531f;
}
// two activations at once on the packed fp32 ALU (v_pk_fma/add/mul_f32: two lanes of work per issue slot; exp2 / log2 and
// max have no packed form): u = 100 log2(e) (acc + b) arrives as acc * C + b_scaled with the bias pre-scaled by C
typedef float v2f_t __attribute__((ext_vector_type(2)));
constexpr float SOFTPLUS_C = 144.26950408889634f;
__device__ __forceinline__ v2f_t softplus100_pk(v2f_t acc, v2f_t bias_scaled) {
  const v2f_t cc = {SOFTPLUS_C, SOFTPLUS_C};
  const v2f_t u = acc * cc + bias_scaled;
  v2f_t y = {__builtin_amdgcn_exp2f(-fabsf(u.x)), __builtin_amdgcn_exp2f(-fabsf(u.y))};
  const v2f_t one = {1.0f, 1.0f};
  y = y + one;
  const v2f_t lg = {__builtin_amdgcn_logf(y.x), __builtin_amdgcn_logf(y.y)};
  const v2f_t m = {fmaxf(u.x, 0.0f), fmaxf(u.y, 0.0f)};
  const v2f_t k = {0.0069314718055994531f, 0.0069314718055994531f};
  return (m + lg) * k;
}
__device__ __forceinline__ float dphi_fast(float h) { return 1.0f - __expf(-100.0f * h); }

// ---------------------------------------------------------------------------------------------
// Fused SDF primal chain (ImplicitNetwork.forward, rend_a :78-96, + get_sdf_vals' clamp :131-137)
// ---------------------------------------------------------------------------------------------
struct FusedArgs {
  const float* x_fm; int P, ldp;
  const uint4* Wp[9]; int KS[9]; const float* bias[9];
  int save, values_only;
  u16* h[9];                  // h[1..8], octet-major (save mode)
  float* E;                   // [39][ldp] fp32 (save mode; the split-precision chain READS it: kernels_x3.hpp)
  u16* feat; float* sdfraw;   // lin8 outputs (save mode): 256 feature rows (octet-major) + raw sdf row
  const uint4* Wlo[9]; u16* hlo[9]; u16* featlo;      // split-precision chain only: lo packs, lo planes of h[1..8] and of the feature rows
  float* sdf_out;             // values mode: clamped sdf, row-major [P]
  float radius, scale;
  int bias8_rot, bias8_n;     // lin8 rows are packed [feature | sdf]
  const int* gate; int gate_value;      // device-side gate (sync-free sampler): the launch does nothing unless *gate == gate_value (null: run)
};

// argument block of the fused adjoint chain (kernels_fused.hpp: sdf_adjoint_w64_kernel)
struct AdjArgs {
  int P, ldp;
  const uint4* Wp[8];            // transposed packs of layers 0 .. 7 (A fragments [tile][16][64])
  const u16* h[9];               // h[1..8]: saved post-activations (octet-major)
  u16* u[8];                     // u[0..7] (octet-major; written in SAVE mode)
  const float* w8; const float* rs8;      // the sdf row of lin8 (weight_v row 0) and its weight-norm scale: the seed
  float* es; float* e0;          // fp32 feature-major [39][ldp] each
  const uint4* Wlo[8]; const u16* hlo[9];  // split-precision chain only: lo packs, lo planes of h[1..8]
};

// one head of the split-precision forward (kernels_x3.hpp: head_chain_x3_kernel)
struct HeadX3Args {
  int P, ldp;
  int nvalid;                                 // batches of 64 points that hold ray samples; later ones (eikonal points, padding) are zero-filled
  const u16* feat; const u16* featlo;         // octet-major, 256 rows
  const float* small; int srows;              // fp32 feature-major [srows][ldp]
  const uint4* Wp[5]; const uint4* Wlo[5];    // packs: lin0 has 20 k-steps ([256 feature | small rows, padded to 64])
  const float* bias[5];
  u16* hid[5];                                // hid[1..4]: hi planes of the hidden activations (save mode)
  float* out;                                 // [3 | 6][ldp] fp32
  const u16* smallbf;                         // one-product chain (kernels_heads.hpp): the small inputs octet-major, 16-bit (oct_pack)
  u16* mask[5];                               // mask[1..4] (save mode): ReLU masks, one 32-bit word per (tile, wave, lane) in the accumulator layout (relu_mask_word)
};

// one head of the fused backward chain (kernels_heads.hpp: head_bwd_chain_kernel)
struct HeadBwdArgs {
  int P, ldp;
  int nvalid;                        // pairs of 32-point tiles that hold ray samples
  const u16* top;                    // cotangent of the head's output, octet-major (one octet: rows 0..2 / 0..5, the rest zero)
  const uint4* Wt[5];                // transposed packs of lin0 .. lin4 (Wt[4]: 4 k-steps per row tile, only the first is non-zero;
                                     // Wt[0]: 256 feature rows, then the small-input rows)
  const u16* mask[5];                // mask[1..4]: ReLU masks written by the forward chain
  u16* ab[4];                        // ab[l]: cotangent of the pre-activation of hid[l + 1] (octet-major; the weight gradients read them)
  u16* featc;                        // feature cotangent, 256 rows octet-major
  int accumulate;                    // 1: featc += (the second head), scaled by cot_scale_of(rho_num) / cot_scale_of(rho_den)
  const float* rho_num; const float* rho_den;
  float* sc; int srows;              // small-input cotangents, fp32 [srows][ldp]
};

}  // namespace neat

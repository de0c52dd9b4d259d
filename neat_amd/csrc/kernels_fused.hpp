// Fused primal chain of the bf16 build, generations 2-4 (generation 1 = sdf_fused_ws_kernel in kernels_bf16.hpp):
//   sdf_fused_w64_kernel<NT = 4, VALUES, RT>: stage pipeline; RT = 2 -> four waves x 64 output rows, RT = 1 (default, tuning key 4 = 3)
//   -> eight waves x 32 rows, two per SIMD;  sdf_fused_ph_kernel: phase-staggered matrix / vector waves (slower, kept selectable).
// The notes below were written for the RT = 2 variant; what carried over to RT = 1 is the pipeline, not the 64-row slice.
//
// sdf_fused_ws_kernel (kernels_bf16.hpp) gives a wave 32 output rows, so every 1 KiB B fragment it reads from LDS
// feeds ONE MFMA, all eight waves re-read the same activation tile, and the 128 KiB weight matrix of the next layer
// can only be fetched after the last MFMA of this one (one register set).  Here
//   * a wave holds a 64 x 256 slice of W_l as 32 A fragments (128 VGPRs; one wave per SIMD, 512 registers): every
//     B fragment read from LDS feeds TWO MFMAs -> half the LDS read traffic per flop, and the two MFMAs of a k-step
//     go to independent accumulators;
//   * the slice of layer l+1 is prefetched from L2 into a SECOND register set at the start of layer l (32 x 1 KiB
//     loads per wave in flight under ~128 MFMAs), so no weight latency is exposed between layers;
//   * batches of 128 points (activations ping-pong between two 64 KiB LDS buffers): 1 KiB of weights per point and
//     layer from L2 instead of 1.4;
//   * the epilogue of point tile t-1 (bias, softplus, pack, LDS + HBM stores) is interleaved quad by quad with the
//     2 x 16 MFMAs of point tile t.
// Layout conventions (octet-major bf16, packed A fragments, lin8 rows [feature | sdf]) are those of kernels_bf16.hpp.
#pragma once
#include "bf16_common.hpp"
#include "fused_launch.hpp"
#include <type_traits>

namespace neat {

// NEAT_F6_ABLATE (probe builds only; results are WRONG): 1 = epilogue without the softplus math, 2 = no MFMAs, 3 = no B-fragment
// LDS reads, 4 = no stage barriers, 5 = no epilogue LDS writes, 9 = no HBM stores of the hidden activations (save mode)
#ifndef NEAT_F6_ABLATE
#define NEAT_F6_ABLATE 0
#endif
#ifndef NEAT_ADJ_ABLATE
#define NEAT_ADJ_ABLATE 0   // probe builds of the adjoint chain only (results WRONG): 1 = no stores of u, 2 = no loads of the saved h quads
#endif
#ifndef NEAT_ADJ_NT_LOAD
#define NEAT_ADJ_NT_LOAD 1      // adjoint chain: the saved h quads arrive with non-temporal loads (round 5: 284 -> 271 us at C2)
#endif
#ifndef NEAT_F6_WIDE
#define NEAT_F6_WIDE 1         // primal chain (save mode): the hidden activations leave as 16-byte stores (as NEAT_ADJ_WIDE)
#endif
#ifndef NEAT_ADJ_WIDE
#define NEAT_ADJ_WIDE 1       // adjoint chain: the saved h quads arrive and the u quads leave as 16-byte accesses (whole octets: lanes 0-31 the octet of
                              // quad 2j, lanes 32-63 the octet of quad 2j+1, halves exchanged with v_permlane32_swap) instead of 8-byte ones
#endif
#ifndef NEAT_F6_SPLITK
#define NEAT_F6_SPLITK 0    // RT = 1: k-steps alternate between two accumulator chains (measured neutral)
#endif
#ifndef NEAT_F6_GROUP
#define NEAT_F6_GROUP 1     // k-steps whose MFMAs are issued back to back before their share of the epilogue
#endif
#ifndef NEAT_F6_RING
#define NEAT_F6_RING 3      // B-fragment ring: registers (k-steps in flight + 1)
#endif

// RT = 32-row output tiles per wave: 2 -> four waves (one per SIMD, 512 registers), 1 -> eight waves (two per SIMD, 256 registers:
// the two waves of a SIMD fill each other's matrix-pipe and VALU gaps)
template <int NT, int RT = 2> struct F6Cfg {
  static constexpr int NW = 8 / RT, THREADS = 64 * NW;
  static constexpr int BP = 32 * NT;
  static constexpr int XBYTES = 32 * BP * 16;            // one activation buffer [32 octets][BP][16 B]
  static constexpr int PEBYTES = 8 * BP * 16;            // PE octets (K padded to 64)
  static constexpr int BIAS_FLOATS = 9 * 256 + 8;
  static constexpr int XA = 0, XB = XBYTES, PE = 2 * XBYTES, BIAS = 2 * XBYTES + PEBYTES, RED = BIAS + BIAS_FLOATS * 4;     // byte offsets
  static constexpr int LDS = RED + NW * BP * 4;          // + partial sums of the sdf row [waves][BP]
};

// Per-lane base addresses of one wave; every access of the stage loop is  base + compile-time offset  that fits the
// instruction's immediate field (ds: 16 bits, so one base per 64 KiB LDS region; global: wave-uniform SGPR base + 32-bit
// per-lane offset + 12 bits), so nothing address-like is recomputed -- or hoisted out of the batch loop and spilled --
// per quad, tile or layer.  The bases are made opaque (empty asm) so that the compiler does not re-derive them from one
// another with constants that do not fit.
struct F6Lane {
  const unsigned char* frag[3];   // B-fragment reads from XA / XB / PE:  region + (hi * BP + (lane & 31)) * 16
  unsigned char* quad[2];         // accumulator-quad writes into XA / XB: region + ((4 RT wave) * BP + (lane & 31)) * 16 + 8 hi
  const unsigned char* bias;      // bias float4 reads: BIAS + (64 wave + 4 hi) * 4
  const unsigned char* pe0;       // PE octet 0 of this lane's point: PE + (lane & 31) * 16        (+ t * 512)
  unsigned gquad;                 // HBM quad store:   ((8 wave) * ldp + p0 + (lane & 31)) * 16 + 8 hi      (per batch)
  unsigned goct;                  // adjoint chain, 16-byte accesses: ((4 wave + hi) * ldp + p0 + (lane & 31)) * 16 -- lanes 32-63 one octet row further, whole octets
  unsigned ldp16;                 // ldp * 16
  float* frows;                   // adjoint chain: fp32 feature-major rows [.][ldp] for the rows >= SPLIT of the current layer
  unsigned fcol;                  // p0 + (lane & 31)                                                             (per batch)
};

// ---------------------------------------------------------------------------------------------------------------
// Stage pipeline.  A STAGE = the 2 x KS MFMAs of one (layer, point tile) interleaved, slot by slot, with the epilogue of
// the PREVIOUS stage (which may belong to the previous layer): after every MFMA the wave issues one "half unit" of
// epilogue work (~7 VALU instructions) -- with one wave per SIMD nothing else can fill the 32 cycles the matrix pipe
// needs per MFMA.  One workgroup barrier per stage; layer l+1 starts on tile 0 while tile 3 of layer l is still in its
// epilogue, so there is no bubble between layers.  Accumulators ping-pong between two register sets (static indices:
// everything is unrolled).
// ---------------------------------------------------------------------------------------------------------------
struct F6EpiState { float m0, m1, w0, w1; unsigned lo; uint2 vprev; unsigned hs[4]; };       // what travels from half unit A to half unit B of a value pair (vprev: adjoint chain, the even quad of a 16-byte store)
#ifndef NEAT_ADJ_NT_FROWS
#define NEAT_ADJ_NT_FROWS 0     // the adjoint chain's fp32 rows (PE cotangents e0 / es, read by sdf_finalize_kernel right behind it)
#endif
__device__ __forceinline__ void f6_storef(float* p, float v) { if (NEAT_ADJ_NT_FROWS) __builtin_nontemporal_store(v, p); else *p = v; }
#ifndef NEAT_F6_NT_E
#define NEAT_F6_NT_E 0         // the fp32 PE rows the primal chain saves
#endif
#ifndef NEAT_F6_NT
#define NEAT_F6_NT 1           // the saved arrays (h_l, u_l: read again only by the backward pass) leave with non-temporal stores: they no longer
                               // displace the weight fragments the rolling refills fetch from L2 (round 5: adjoint chain 300 -> 255 us at C2)
#endif
// quads 2j (prev) and 2j+1 (cur) of this lane -> the octet this lane stores: quad 2j's octet for lanes 0-31, quad 2j+1's for lanes 32-63
__device__ __forceinline__ uint4 f6_octet(uint2 prev, uint2 cur) {
  typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
  const v2u_t s0 = __builtin_amdgcn_permlane32_swap(prev.x, cur.x, false, false);
  const v2u_t s1 = __builtin_amdgcn_permlane32_swap(prev.y, cur.y, false, false);
  return make_uint4(s0.x, s1.x, s0.y, s1.y);
}
__device__ __forceinline__ void f6_store16(void* p, uint4 v) {
  typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
  const v4u_t w = {v.x, v.y, v.z, v.w};
  if (NEAT_F6_NT) __builtin_nontemporal_store(w, reinterpret_cast<v4u_t*>(p));
  else *reinterpret_cast<v4u_t*>(p) = w;
}
__device__ __forceinline__ void f6_store8(void* p, uint2 v) {
  typedef unsigned long long u64_t;
  if (NEAT_F6_NT) __builtin_nontemporal_store(__builtin_bit_cast(u64_t, v), reinterpret_cast<u64_t*>(p));
  else *reinterpret_cast<uint2*>(p) = v;
}

// epilogue of one layer (compile-time description): activation, destination, row count, bias rows
// UDOM (round 6, values mode): the activations stay in the softplus's own scale -- h' = max(u, 0) + log2(1 + 2^-|u|) with u = 100 log2(e) x,
// i.e. h' = (100 / ln 2) h.  The next layer's pre-activation in that scale is then simply acc' + b' (the inputs are already multiplied by
// 100 log2 e), so the epilogue needs neither the fma's scale nor the output multiply: two vector instructions less per value pair of a
// kernel whose vector pipe is the longer one.  The PE rows enter scaled the same way, the sdf sum is divided once per point.  Nothing is
// saved in values mode, so no other kernel sees the scale.
template <bool ACT_, bool SAVE_, int N_, int DST_, int BIASOFF_, bool UDOM_ = false> struct F6EpiCfg {
  static constexpr bool ACT = ACT_, SAVE = SAVE_, NONE = false, REV = false, UDOM = UDOM_;
  static constexpr int N = N_, DST = DST_, BIASOFF = BIASOFF_, SPLIT = 1 << 30;
};
struct F6NoEpi { static constexpr bool ACT = false, SAVE = false, NONE = true, REV = false, UDOM = false; static constexpr int N = 256, DST = 0, BIASOFF = 0, SPLIT = 1 << 30; };
// adjoint chain (fused EPI_REV): value = acc * phi'(h), phi'(a) = 1 - exp(-100 h), h = the saved post-activation of the layer below,
// which travels in the `bq` argument as raw bf16 quads (bq[g].x / .y = the two pairs of quad g).  Rows >= SPLIT leave as fp32 rows
// (row - SPLIT) of F6Lane::frows without the phi' factor and are zero on chip and in the bf16 array: the PE cotangent of the skip
// layer (SPLIT = 217) and the whole output of the last layer (SPLIT = 0, N = 39).
template <bool SAVE_, int N_, int DST_, int SPLIT_> struct F6RevCfg {
  static constexpr bool ACT = false, SAVE = SAVE_, NONE = false, REV = true, UDOM = false;
  static constexpr int N = N_, DST = DST_, BIASOFF = 0, SPLIT = SPLIT_;
};

// the bias rows of this lane for a layer: 8 quads x float4 (pre-scaled by SOFTPLUS_C for the activated layers)
template <class E, int RT> __device__ __forceinline__ void f6_load_bias(const F6Lane& L, float4 (&bq)[4 * RT]) {
#pragma unroll
  for (int g = 0; g < 4 * RT; ++g) bq[g] = *reinterpret_cast<const float4*>(L.bias + (E::BIASOFF + (g >> 2) * 32 + 8 * (g & 3)) * 4);
}

// the same rows in accumulator order (row tile i, quad q, element j -> c0[i][4 q + j]): operand C of a layer's first MFMA (f6_stage CINIT)
template <class E, int RT> __device__ __forceinline__ void f6_load_bias16(const F6Lane& L, f32x16 (&c0)[RT]) {
#pragma unroll
  for (int g = 0; g < 4 * RT; ++g) {
    const float4 b = *reinterpret_cast<const float4*>(L.bias + (E::BIASOFF + (g >> 2) * 32 + 8 * (g & 3)) * 4);
    c0[g >> 2][4 * (g & 3) + 0] = b.x; c0[g >> 2][4 * (g & 3) + 1] = b.y; c0[g >> 2][4 * (g & 3) + 2] = b.z; c0[g >> 2][4 * (g & 3) + 3] = b.w;
  }
}

// half unit h (0 = A: bias, exp2, max; 1 = B: log2, scale, pack, store) of value pair e (0..15): quad g = e >> 1 = (row tile
// i = g >> 2, quad q = g & 3), pair e & 1 of the quad; t = point tile the accumulators `ae` belong to
template <int NT, int RT, bool FULL, class E>
__device__ __forceinline__ void f6_epi_half(const F6Lane& L, const f32x16 (&ae)[RT], const float4 (&bq)[4 * RT], F6EpiState& st, int e, int h, int t,
                                            int nt, u16* hout, int wave, int hi) {
  typedef F6Cfg<NT, RT> C;
  constexpr int BP = C::BP;
  if (NEAT_F6_ABLATE == 7 || NEAT_F6_ABLATE == 8) { if (e == 0 && h == 0) asm volatile("" :: "v"(ae[0][0])); return; }
  const int g = e >> 1, i = g >> 2, q = g & 3, pr = e & 1;
  const float x0 = ae[i][4 * q + 2 * pr], x1 = ae[i][4 * q + 2 * pr + 1];
  const float b0 = pr ? bq[g].z : bq[g].x, b1 = pr ? bq[g].w : bq[g].y;
  if (E::REV) {
    const int nb = 32 * (RT * wave + i) + 8 * q + 4 * hi + 2 * pr;          // rows nb, nb + 1 of the layer's output
    if (h == 0) {
      if (E::SPLIT > 0) {
        unsigned hw;
        if (NEAT_ADJ_WIDE) {      // bq[q even] = the raw octet this lane loaded (load_h): first use of a quad pair swaps the halves (hs = quad q | quad q+1)
          if ((q & 1) == 0 && pr == 0) {
            typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
            const v2u_t s0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(bq[g].x), __float_as_uint(bq[g].z), false, false);
            const v2u_t s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(bq[g].y), __float_as_uint(bq[g].w), false, false);
            st.hs[0] = s0.x; st.hs[1] = s1.x; st.hs[2] = s0.y; st.hs[3] = s1.y;
          }
          hw = st.hs[2 * (q & 1) + pr];
        } else hw = __float_as_uint(pr ? bq[g].y : bq[g].x);
        st.m0 = x0 * dphi_fast(bf_lo(hw));      // (the streaming EPI_REV epilogue's expression: bit-identical results)
        st.m1 = x1 * dphi_fast(bf_hi(hw));
      }
      return;
    }
    float r0 = st.m0, r1 = st.m1;
    if (E::SPLIT < 256 && 32 * (RT * wave + i) + 8 * q + 7 >= E::SPLIT) {      // (wave-uniform: only the row tiles that reach the split)
      // row indices from an opaque copy of the lane's half: the per-(quad, pair) rows, masks and offsets are loop invariants that
      // would otherwise be hoisted out of the batch loop -- 64 values -- and spilled
      int hio = hi;
      asm volatile("" : "+v"(hio));
      const int n0 = 32 * (RT * wave + i) + 8 * q + 4 * hio + 2 * pr;
      const bool st_ok = FULL || t < nt;
      const unsigned col = L.fcol + t * 32, ld = L.ldp16 >> 4;
      if (n0 >= E::SPLIT) { r0 = 0.0f; if (st_ok && n0 < E::N) f6_storef(L.frows + (unsigned)(n0 - E::SPLIT) * ld + col, x0); }
      if (n0 + 1 >= E::SPLIT) { r1 = 0.0f; if (st_ok && n0 + 1 < E::N) f6_storef(L.frows + (unsigned)(n0 + 1 - E::SPLIT) * ld + col, x1); }
    }
    const unsigned pk = pack2(r0, r1);
    if (pr == 0) { st.lo = pk; return; }
    const uint2 v = make_uint2(st.lo, pk);
    if (E::SPLIT > 0) *reinterpret_cast<uint2*>(L.quad[E::DST] + ((i * 4 + q) * BP + t * 32) * 16) = v;
    if (E::SAVE && (FULL || t < nt) && !(NEAT_ADJ_ABLATE & 1)) {
      if (NEAT_ADJ_WIDE) {
        if (q & 1) f6_store16(reinterpret_cast<char*>(hout) + ((unsigned)(i * 4 + q - 1) * L.ldp16 + L.goct) + t * 512, f6_octet(st.vprev, v));
        else st.vprev = v;
      } else f6_store8(reinterpret_cast<char*>(hout) + ((unsigned)(i * 4 + q) * L.ldp16 + L.gquad) + t * 512, v);
    }
    return;
  }
  // The bias is already in the accumulator (f6_stage CINIT): x = W in + b.  Softplus with beta = 100 (rend_a :94):
  //   standard scale:  h = max(x, 0) + (ln 2 / 100) log2(1 + 2^(-|x| 100 log2 e))     -- max, mul, exp2, 1 +, log2, fma
  //   UDOM (x is u = 100 log2 e * pre-activation):  h' = max(u, 0) + log2(1 + 2^-|u|)  -- max, exp2, 1 +, log2, +
  // The maximum is inline asm: fmaxf on a raw MFMA result is preceded by a canonicalising v_max_f32 x, x, x (and the median of (x, 0, inf)
  // is folded into the same pair), one instruction per value more.
  if (h == 0) {
    if (E::ACT && NEAT_F6_ABLATE != 1) {
      st.w0 = 1.0f + __builtin_amdgcn_exp2f(-fabsf(E::UDOM || NEAT_F6_ABLATE == 6 ? x0 : x0 * SOFTPLUS_C));
      st.w1 = 1.0f + __builtin_amdgcn_exp2f(-fabsf(E::UDOM || NEAT_F6_ABLATE == 6 ? x1 : x1 * SOFTPLUS_C));
      asm("v_max_f32 %0, 0, %1" : "=v"(st.m0) : "v"(x0)); asm("v_max_f32 %0, 0, %1" : "=v"(st.m1) : "v"(x1));
    } else {
      st.m0 = x0; st.m1 = x1;
    }
    return;
  }
  float r0 = st.m0, r1 = st.m1;
  if (E::ACT && NEAT_F6_ABLATE != 1) {
    if (E::UDOM || NEAT_F6_ABLATE == 6) { r0 = st.m0 + __builtin_amdgcn_logf(st.w0); r1 = st.m1 + __builtin_amdgcn_logf(st.w1); }
    else { r0 = fmaf(__builtin_amdgcn_logf(st.w0), 0.0069314718055994531f, st.m0); r1 = fmaf(__builtin_amdgcn_logf(st.w1), 0.0069314718055994531f, st.m1); }
  }
  const unsigned pk = pack2(r0, r1);
  if (pr == 0) { st.lo = pk; return; }
  uint2 v = make_uint2(st.lo, pk);
  if (E::N == 217 && i == 0 && q == 3) {       // (row tile 6 is the first tile of its wave for RT = 1 and RT = 2)
    // lin3, rows 216..223 (wave 3 only): [h216 | PE rows 0..6] -- what lin4 (skip connection, rend_a :87-88) and the saved h4
    // expect in the last octet of the 217-row array.  (Row tile 7 holds don't-care values until the skip copy.)
    if (RT * wave == 6) {
      const uint4 w = *reinterpret_cast<const uint4*>(L.pe0 + t * 512);       // PE rows 0..7 of this point, bf16
      if (hi == 0) v = make_uint2((st.lo & 0xFFFFu) | (w.x << 16), (w.x >> 16) | (w.y << 16));
      else v = make_uint2((w.y >> 16) | (w.z << 16), (w.z >> 16) | (w.w << 16));
    }
  }
  if (E::ACT && NEAT_F6_ABLATE != 5) *reinterpret_cast<uint2*>(L.quad[E::DST] + ((i * 4 + q) * BP + t * 32) * 16) = v;
  if (NEAT_F6_ABLATE == 5) asm volatile("" :: "v"(v.x), "v"(v.y));
  if (E::SAVE && NEAT_F6_ABLATE != 9 && (FULL || t < nt)) {                   // wave-uniform row base + per-lane 32-bit offset + immediate
    if (NEAT_F6_WIDE) {      // 16-byte stores: the even quad waits for its odd neighbour, lanes 0-31 store the even quad's octet, lanes 32-63 the odd one's
      if (q & 1) f6_store16(reinterpret_cast<char*>(hout) + ((size_t)((i * 4 + q - 1) * L.ldp16) + t * 512) + (size_t)L.goct, f6_octet(st.vprev, v));
      else st.vprev = v;
    } else f6_store8(reinterpret_cast<char*>(hout) + ((size_t)((i * 4 + q) * L.ldp16) + t * 512) + (size_t)L.gquad, v);
  }
}

// MMA = false: drain stage (epilogue only).  KS k-steps of layer input region SRC (0 = XA, 1 = XB, 2 = PE), tile t; the epilogue
// E works on tile te of the accumulators `ae`.  B fragments travel through a ring of three registers, two k-steps ahead; the
// first two fragments of the NEXT stage (base address nfr, or null) are requested during the last two k-steps, i.e. before the
// barrier that ends the stage: they were written at least two barriers ago.  ROFF = ring slot of this stage's k-step 0
// (every KS is 1 mod 3, so it advances by one per stage).  BAR: end the stage with the workgroup barrier.
// WROLL (adjoint chain): this is the layer's last tile -- slot ks of the weight registers is dead after its MFMA and is refilled right
// there with the NEXT layer's slot ks (wnx = that layer's per-lane fragment address), so one set of 16 fragment registers serves
// the whole chain instead of two (the request is a full stage, >= 1000 cycles, ahead of its first use; the weights sit in L2).
// CINIT (values mode, F6EpiCfg UDOM): the accumulators start from c0 = the layer's bias rows of this lane instead of zero (operand C of the
// first k-step's MFMA), so the epilogue has no bias add left.
template <int NT, int RT, bool FULL, bool MMA, int KS, int SRC, class E, int ROFF, bool BAR, bool WROLL = false, bool CINIT = false>
__device__ __forceinline__ void f6_stage(const F6Lane& L, std::conditional_t<WROLL, uint4, const uint4> (&wc)[RT][16], int t, f32x16 (&am)[RT],
                                         const f32x16 (&ae)[RT], const float4 (&bq)[4 * RT], int te, int nt, u16* hout, int wave, int hi,
                                         uint4 (&ring)[NEAT_F6_RING], const unsigned char* nfr, const unsigned char* wnx = nullptr,
                                         const f32x16* c0 = nullptr) {
  typedef F6Cfg<NT, RT> C;
  constexpr int STEP = 2 * C::BP * 16;                 // bytes between k-steps of a fragment column
  constexpr int HU = 16 * RT;                          // epilogue half units of a tile: 8 RT value pairs x {A, B}
  constexpr int SLOTS = MMA ? RT * KS : 1;
  constexpr int UPS = E::NONE ? 0 : HU / SLOTS;        // epilogue half units per slot
  constexpr int RD = NEAT_F6_RING, AH = RD - 1;       // ring size / k-steps of look-ahead
  F6EpiState st{};
  if (!MMA) {
    if (nfr) {
#pragma unroll
      for (int j = 0; j < AH; ++j) ring[(ROFF + j) % RD] = *reinterpret_cast<const uint4*>(nfr + j * STEP);
    }
#pragma unroll
    for (int u = 0; u < HU; ++u) {
      if (!E::NONE) f6_epi_half<NT, RT, FULL, E>(L, ae, bq, st, u >> 1, u & 1, te, nt, hout, wave, hi);
      if ((u & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    if (BAR) __syncthreads();
    return;
  }
  const unsigned char* fr = L.frag[SRC] + t * 512;
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  f32x16 odd = zero;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    if (NEAT_F6_ABLATE != 3 && NEAT_F6_ABLATE != 8) {
      if (ks + AH < KS) ring[(ks + AH + ROFF) % RD] = *reinterpret_cast<const uint4*>(fr + (ks + AH) * STEP);
      else if (nfr) ring[(ks + AH + ROFF) % RD] = *reinterpret_cast<const uint4*>(nfr + (ks + AH - KS) * STEP);
    }
    const uint4 cur = ring[(ks + ROFF) % RD];
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      // RT = 1: the k-steps alternate between two accumulator chains (a chain of DEPENDENT 32x32x16 MFMAs with other instructions in
      // between runs at ~75 cycles per MFMA instead of 32); they are summed after the last k-step
      f32x16& dst = (NEAT_F6_SPLITK && RT == 1 && (ks & 1)) ? odd : am[i];
      if (NEAT_F6_ABLATE != 2)
        dst = NEAT_MFMA16(*reinterpret_cast<const bf16x8*>(&wc[i][ks]), *reinterpret_cast<const bf16x8*>(&cur), ks >= (NEAT_F6_SPLITK && RT == 1 ? 2 : 1) ? dst : ((CINIT && ks == 0) ? c0[i] : zero), 0, 0, 0);
      else if (ks == 0) { am[i] = zero; am[i][0] = __uint_as_float(cur.x ^ wc[i][ks].x); }
      if constexpr (WROLL) wc[i][ks] = *reinterpret_cast<const uint4*>(wnx + ks * 1024);
      if (NEAT_F6_GROUP == 1) {
#pragma unroll
        for (int u = 0; u < UPS; ++u) {
          const int hu = (RT * ks + i) * UPS + u;
          f6_epi_half<NT, RT, FULL, E>(L, ae, bq, st, hu >> 1, hu & 1, te, nt, hout, wave, hi);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (NEAT_F6_GROUP > 1 && ((ks + 1) % NEAT_F6_GROUP == 0 || ks == KS - 1)) {
      // coarse schedule: the MFMAs of the last GROUP k-steps are in the matrix pipe's queue; now their share of the epilogue
      constexpr int G = NEAT_F6_GROUP;
      const int ks0 = ks - (ks % G);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int hu = 0; hu < HU; ++hu)
        if (hu >= ks0 * RT * UPS && hu < (ks + 1) * RT * UPS) {
          f6_epi_half<NT, RT, FULL, E>(L, ae, bq, st, hu >> 1, hu & 1, te, nt, hout, wave, hi);
          if ((hu & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (NEAT_F6_SPLITK && RT == 1 && NEAT_F6_ABLATE != 2) {
#pragma unroll
    for (int r = 0; r < 16; ++r) am[0][r] += odd[r];
  }
  if (BAR && NEAT_F6_ABLATE != 4) __syncthreads();
}

template <int NT, bool VALUES, int RT>
__global__ __launch_bounds__(64 * (8 / RT), 2 / RT) void sdf_fused_w64_kernel(FusedArgs a, int ntiles, int nwg) {
  typedef F6Cfg<NT, RT> C;
  constexpr int BP = C::BP, F6T = C::THREADS, NW = C::NW;
  static_assert(NT == 4 && (RT == 1 || RT == 2), "the PE phase maps threads to (point of a 128-point batch, frequency group)");
  if (a.gate && *a.gate != a.gate_value) return;
#ifndef NEAT_F6_PRIO
#define NEAT_F6_PRIO 0      // probe: static priority for one half of the workgroup's waves (MI355X_MICROARCH.md, pairing item 4: the second-dispatched half)
#endif
  if (NEAT_F6_PRIO == 1 && threadIdx.x >= 256) __builtin_amdgcn_s_setprio(1);
  if (NEAT_F6_PRIO == 2 && threadIdx.x < 256) __builtin_amdgcn_s_setprio(1);
  extern __shared__ __attribute__((aligned(16))) unsigned char f6lds[];
  float* biasl = reinterpret_cast<float*>(f6lds + C::BIAS);     // [l][256]; lin8 in packed row order
  float* red = reinterpret_cast<float*>(f6lds + C::RED);        // [waves][BP]: partial sums of the sdf row
  u16* pe16 = reinterpret_cast<u16*>(f6lds + C::PE);
  const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr bool SAVE = !VALUES;

  for (int idx = tid; idx < 8 * 256; idx += F6T) {
    const int l = idx >> 8, n = idx & 255;
    float v = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) if (l == k && n < (k == 3 ? 217 : 256)) v = a.bias[k][n];
    biasl[idx] = VALUES ? v * SOFTPLUS_C : v;      // hidden layers (values mode: in the softplus's scale, F6EpiCfg UDOM); operand C of the layer's first MFMAs
  }
  for (int n = tid; n < 257; n += F6T) {
    int bi = n + a.bias8_rot; if (bi >= a.bias8_n) bi -= a.bias8_n;
    biasl[8 * 256 + n] = VALUES ? (n == 0 ? a.bias[8][0] : 0.0f) : a.bias[8][bi];
  }
  // PE rows 39..63 (octets 4..7; row 32..38 are rewritten per batch) stay zero for the whole launch
  for (int idx = tid; idx < 4 * BP; idx += F6T) reinterpret_cast<uint4*>(f6lds + C::PE)[4 * BP + idx] = make_uint4(0u, 0u, 0u, 0u);

  // two register sets for the weight slices (this layer / next layer); lin0's short slice (K = 64: 4 k-steps) and this wave's
  // k-steps of the sdf row of lin8 are fetched per batch into whichever set is idle
  uint4 wA[RT][16], wB[RT][16];
  auto load_w = [&](uint4 (&dst)[RT][16], const uint4* Wl, int KS, int N) {
    // kernarg pointer (SGPR pair) + ONE 32-bit per-lane offset (made opaque: otherwise the fragment addresses of all layers are
    // loop-invariant 64-bit values that get hoisted out of the batch loop and spilled) + immediate.  Dead row tiles (lin3:
    // rows >= 224) re-read a live one; their products are never used.
    const char* base = reinterpret_cast<const char*>(Wl);
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      const int tile = (RT * wave + i) * 32 < N ? RT * wave + i : 0;
      unsigned voff = (unsigned)((tile * KS) * 64 + lane) * 16u;
      asm volatile("" : "+v"(voff));
#pragma unroll
      for (int ks = 0; ks < 16; ++ks)
        if (ks < KS) dst[i][ks] = *reinterpret_cast<const uint4*>(base + voff + ks * 1024);
    }
  };

  F6Lane L;
  {
    const unsigned fo = (unsigned)(hi * BP + (lane & 31)) * 16u, qo = (unsigned)((4 * RT * wave) * BP + (lane & 31)) * 16u + 8u * hi;
    unsigned b0 = fo, b1 = fo + C::XB, b2 = fo + C::PE, q0 = qo, q1 = qo + C::XB, bb = C::BIAS + (unsigned)(32 * RT * wave + 4 * hi) * 4u;
    unsigned pz = C::PE + (unsigned)(lane & 31) * 16u;
    asm volatile("" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(q0), "+v"(q1), "+v"(bb), "+v"(pz));
    L.frag[0] = f6lds + b0; L.frag[1] = f6lds + b1; L.frag[2] = f6lds + b2;
    L.quad[0] = f6lds + q0; L.quad[1] = f6lds + q1; L.bias = f6lds + bb; L.pe0 = f6lds + pz;
  }
  L.ldp16 = (unsigned)a.ldp * 16u;

  const bool inter = nwg < 0;
  const int ng = inter ? -nwg : nwg;
  const int t_begin = inter ? (int)blockIdx.x * NT : (int)(((long long)blockIdx.x * ntiles) / ng);
  const int t_end = inter ? ntiles : (int)(((long long)(blockIdx.x + 1) * ntiles) / ng);
  const int t_step = inter ? ng * NT : NT;
  // A batch = NT tiles through the stage pipeline.  What is left of a workgroup's range (1 .. NT-1 tiles: the ragged end of a big
  // launch, or everything a workgroup gets in a small one -- get_outputs on the R junction points) goes tile by tile through the
  // SINGLE path: one tile per layer, MFMAs then epilogue then barrier, no pipeline.  A pipeline batch costs its full latency
  // (~45 us for nine layers) whatever it holds; a single tile costs ~1/6 of that.
  for (int tile0 = t_begin; tile0 < t_end; tile0 += t_step) {
   const int ntb = min(NT, t_end - tile0);
   for (int sub = 0; sub < (ntb == NT ? 1 : ntb); ++sub) {
    const int p0 = (tile0 + sub) * 32;
    const int nt = ntb == NT ? NT : 1;
    L.gquad = ((unsigned)(4 * RT * wave) * (unsigned)a.ldp + (unsigned)(p0 + (lane & 31))) * 16u + 8u * hi;
    L.goct = ((unsigned)(4 * RT * wave + hi) * (unsigned)a.ldp + (unsigned)(p0 + (lane & 31))) * 16u;

    auto chain = [&](auto single_tag) {
      constexpr bool FULL = true, SINGLE = decltype(single_tag)::value;
      // the slices of lin0 and lin1 travel while the positional encoding is computed
      load_w(wB, a.Wp[0], 4, 256);
      load_w(wA, a.Wp[1], 16, 256);
      // ---- positional encoding (embedder.py:12-36): thread = (point, group of NFQ frequencies); rows 0..2 = x, then per frequency
      // k: rows 3+6k.. = sin(2^k xyz), rows 6+6k.. = cos(2^k xyz)
      {
        constexpr int NG = F6T / BP, NFQ = 6 / (NG < 6 ? (NG == 4 ? 3 : NG) : 6);     // 2 groups x 3 frequencies, or 3 (of 4) groups x 2
        const int p = tid & (BP - 1), fg = __builtin_amdgcn_readfirstlane(tid / BP);      // (BP >= 64: the group is wave-uniform)
        const bool ok = p < nt * 32;
        float xc[3];
        // rows of x_fm / E: kernarg base (SGPR pair) + ONE 32-bit per-lane byte offset computed where it is used from an opaque
        // copy of ldp.  (With 64-bit row addresses the 13 row pointers of a thread -- its rows depend on its frequency group --
        // were hoisted out of the batch loop and spilled, and every reload carried an s_waitcnt vmcnt(0) that serialised the E
        // stores: +75 us per launch in save mode.)
        unsigned pvo = (unsigned)(p0 + p) * 4u, ldp4 = (unsigned)a.ldp * 4u;
        asm volatile("" : "+v"(pvo), "+v"(ldp4));
#pragma unroll
        for (int c = 0; c < 3; ++c)
          xc[c] = ok ? *reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.x_fm) + ((unsigned)c * ldp4 + pvo)) : 0.0f;
        auto put = [&](int j, float v) {
          pe16[((j >> 3) * BP + p) * 8 + (j & 7)] = f2bf(VALUES ? v * SOFTPLUS_C : v);      // (values mode: the softplus's scale, F6EpiCfg UDOM)
          if (SAVE && ok) { if (NEAT_F6_NT_E) __builtin_nontemporal_store(v, reinterpret_cast<float*>(reinterpret_cast<char*>(a.E) + ((unsigned)j * ldp4 + pvo))); else *reinterpret_cast<float*>(reinterpret_cast<char*>(a.E) + ((unsigned)j * ldp4 + pvo)) = v; }
        };
        if (fg == NG - 1) {
#pragma unroll
          for (int c = 0; c < 3; ++c) put(c, xc[c]);
        }
        if (fg * NFQ < 6) {
#pragma unroll
          for (int kk = 0; kk < NFQ; ++kk) {
            const int k = fg * NFQ + kk;
            const float f = (float)(1 << k);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              const float arg = xc[c] * f;
              put(3 + 6 * k + c, ok ? __sinf(arg) : 0.0f);
              put(6 + 6 * k + c, ok ? __cosf(arg) : 0.0f);
            }
          }
        }
      }
      __syncthreads();

      typedef F6EpiCfg<true, SAVE, 256, 0, 0 * 256, VALUES> E0;      // lin0 -> XA
      typedef F6EpiCfg<true, SAVE, 256, 1, 1 * 256, VALUES> E1;      // lin1 -> XB
      typedef F6EpiCfg<true, SAVE, 256, 0, 2 * 256, VALUES> E2;
      typedef F6EpiCfg<true, SAVE, 217, 1, 3 * 256, VALUES> E3;      // lin3 -> XB rows 0..216 (+ PE rows 0..6 in the last octet)
      typedef F6EpiCfg<true, SAVE, 256, 0, 4 * 256, VALUES> E4;
      typedef F6EpiCfg<true, SAVE, 256, 1, 5 * 256, VALUES> E5;
      typedef F6EpiCfg<true, SAVE, 256, 0, 6 * 256, VALUES> E6;
      typedef F6EpiCfg<true, SAVE, 256, 1, 7 * 256, VALUES> E7;
      typedef F6EpiCfg<false, true, 256, 0, 8 * 256> E8;     // lin8 feature rows -> HBM only
      f32x16 acc[2][RT];
      float4 bq[4 * RT];
      f32x16 c0[RT];                 // the bias rows of the layer whose MFMAs run, as their first operand C
      uint4 ring[NEAT_F6_RING];
      constexpr int RD = NEAT_F6_RING;
#define F6_KSUM(S_) ((S_) < 4 ? 4 * (S_) : ((S_) < 33 ? 16 * (S_) - 48 : 16 * (S_) - 64))      /* k-steps before stage S_ (stage 32 = drain) */
      // skip connection (rend_a :87-88): rows 224..255 of lin4's input (octets 28..31 of XB) = PE rows 7..38 (the 1/sqrt2 is folded
      // into W4).  One thread per (point of tile tc, octet): PE rows 7+8k .. 14+8k straddle PE octets k and k+1.
      auto skip_copy = [&](int tc) {
        if (tid < 128) {
          const int pp = tc * 32 + (tid & 31), k = tid >> 5;
          const uint4* pe = reinterpret_cast<const uint4*>(f6lds + C::PE);
          const uint4 lo = pe[k * BP + pp], hi4 = pe[(k + 1) * BP + pp];
          reinterpret_cast<uint4*>(f6lds + C::XB)[(28 + k) * BP + pp] =
              make_uint4((lo.w >> 16) | (hi4.x << 16), (hi4.x >> 16) | (hi4.y << 16), (hi4.y >> 16) | (hi4.z << 16), (hi4.z >> 16) | (hi4.w << 16));
        }
      };
      // One layer = NT stages.  S0_ = index of its first stage (accumulator set = stage & 1, ring offset = stage % 3); the last
      // stage of a layer prefetches fragments of the next layer's input NSRC_ (3 = none).  A barrier ends every second stage
      // (a tile written in stage w is read in stage w + 3) and every stage where ALLBAR_ says so.
#define F6_STAGE(S_, T_, KS_, SRC_, WC_, E_, H_, TE_, NFR_, BAR_)                                                                   \
      f6_stage<NT, RT, FULL, true, KS_, SRC_, E_, F6_KSUM(S_) % RD, BAR_, false, true>(L, WC_, T_, acc[(S_) & 1], acc[((S_) + 1) & 1], bq, TE_, nt, H_, wave, hi, ring, NFR_, nullptr, c0);
#define F6_LAYER(S0_, KS_, SRC_, WC_, EPREV_, ECUR_, HPREV_, HCUR_, NSRC_, ALLBAR_, PRE_)                                              \
      { f6_load_bias16<ECUR_, RT>(L, c0);                                                                                        \
        PRE_(0) F6_STAGE((S0_) + 0, 0, KS_, SRC_, WC_, EPREV_, HPREV_, NT - 1, L.frag[SRC_] + 1 * 512, (ALLBAR_ || ((S0_) + 0) % 2 == 1))  \
                                                                                                \
        PRE_(1) F6_STAGE((S0_) + 1, 1, KS_, SRC_, WC_, ECUR_, HCUR_, 0, L.frag[SRC_] + 2 * 512, (ALLBAR_ || ((S0_) + 1) % 2 == 1))  \
        PRE_(2) F6_STAGE((S0_) + 2, 2, KS_, SRC_, WC_, ECUR_, HCUR_, 1, L.frag[SRC_] + 3 * 512, (ALLBAR_ || ((S0_) + 2) % 2 == 1))  \
        PRE_(3) F6_STAGE((S0_) + 3, 3, KS_, SRC_, WC_, ECUR_, HCUR_, 2, ((NSRC_) < 3 ? L.frag[(NSRC_) < 3 ? (NSRC_) : 0] : nullptr), (ALLBAR_ || ((S0_) + 3) % 2 == 1)) }
#define F6_NOPRE(T_)
#define F6_SKIPPRE(T_) skip_copy((T_) < 3 ? (T_) + 1 : 3);
      if constexpr (SINGLE) {
        // ---- one tile: per layer the 16 (4) k-steps, then the whole epilogue, then the barrier that publishes the tile
#define F6_RING0(SRC_)                                                                                                              \
        _Pragma("unroll") for (int j = 0; j < RD - 1; ++j) ring[j] = *reinterpret_cast<const uint4*>(L.frag[SRC_] + j * 2 * BP * 16);
#define F6_LAYER1(KS_, SRC_, WC_, ECUR_, HCUR_)                                                                                     \
        { F6_RING0(SRC_)                                                                                                            \
          f6_load_bias16<ECUR_, RT>(L, c0);                                                                                         \
          f6_stage<NT, RT, true, true, KS_, SRC_, F6NoEpi, 0, false, false, true>(L, WC_, 0, acc[0], acc[1], bq, 0, 1, nullptr, wave, hi, ring, nullptr, nullptr, c0); \
                                                                                                   \
          f6_stage<NT, RT, true, false, 16, 0, ECUR_, 0, true>(L, WC_, 0, acc[1], acc[0], bq, 0, 1, HCUR_, wave, hi, ring, nullptr); }
        F6_LAYER1(4, 2, wB, E0, a.h[1])
        load_w(wB, a.Wp[2], 16, 256);
        F6_LAYER1(16, 0, wA, E1, a.h[2])
        load_w(wA, a.Wp[3], 16, 217);
        F6_LAYER1(16, 1, wB, E2, a.h[3])
        load_w(wB, a.Wp[4], 16, 256);
        F6_LAYER1(16, 0, wA, E3, a.h[4])
        load_w(wA, a.Wp[5], 16, 256);
        skip_copy(0);
        __syncthreads();
        F6_LAYER1(16, 1, wB, E4, a.h[5])
        load_w(wB, a.Wp[6], 16, 256);
        F6_LAYER1(16, 0, wA, E5, a.h[6])
        load_w(wA, a.Wp[7], 16, 256);
        F6_LAYER1(16, 1, wB, E6, a.h[7])
        if (!VALUES) load_w(wB, a.Wp[8], 16, 256);
        F6_LAYER1(16, 0, wA, E7, a.h[8])
        {                                                    // lin8's sdf row for tile 0: split over the waves' k-steps, reduced through LDS
          constexpr int KW = 16 / NW;
          unsigned voff = (unsigned)((((VALUES ? 0 : 8) * 16 + KW * wave) * 64 + lane) * 16);
          asm volatile("" : "+v"(voff));
#pragma unroll
          for (int j = 0; j < KW; ++j) wA[0][j] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(a.Wp[8]) + voff + j * 1024);
          const unsigned char* fr = L.frag[1] + (unsigned)(KW * wave) * (2 * BP * 16);
          f32x16 accs;
#pragma unroll
          for (int r = 0; r < 16; ++r) accs[r] = 0.0f;
#pragma unroll
          for (int j = 0; j < KW; ++j) {
            const uint4 bv = *reinterpret_cast<const uint4*>(fr + j * 2 * BP * 16);
            accs = NEAT_MFMA16(*reinterpret_cast<const bf16x8*>(&wA[0][j]), *reinterpret_cast<const bf16x8*>(&bv), accs, 0, 0, 0);
          }
          if (hi == 0) red[wave * BP + lane] = accs[0];
        }
        if (!VALUES) {
          F6_LAYER1(16, 1, wB, E8, a.feat)
        } else {
          __syncthreads();
        }
#undef F6_LAYER1
#undef F6_RING0
      } else {
      // stage 0's first two fragments
#pragma unroll
      for (int j = 0; j < RD - 1; ++j) ring[j] = *reinterpret_cast<const uint4*>(L.frag[2] + j * 2 * BP * 16);
      F6_LAYER(0, 4, 2, wB, F6NoEpi, E0, nullptr, a.h[1], 0, false, F6_NOPRE)
      load_w(wB, a.Wp[2], 16, 256);
      F6_LAYER(4, 16, 0, wA, E0, E1, a.h[1], a.h[2], 1, false, F6_NOPRE)
      load_w(wA, a.Wp[3], 16, 217);
      F6_LAYER(8, 16, 1, wB, E1, E2, a.h[2], a.h[3], 0, false, F6_NOPRE)
      load_w(wB, a.Wp[4], 16, 256);
      // lin3 and lin4 keep a barrier per stage: lin4's tile t reads XB rows 224..255 of tile t, which the skip copy fills one to two
      // stages earlier -- after lin3's epilogue of that tile has written its don't-care rows there
      F6_LAYER(12, 16, 0, wA, E2, E3, a.h[3], a.h[4], 3, true, F6_NOPRE)
      load_w(wA, a.Wp[5], 16, 256);
      skip_copy(0);
      __syncthreads();
#pragma unroll
      for (int j = 0; j < RD - 1; ++j) ring[(F6_KSUM(16) + j) % RD] = *reinterpret_cast<const uint4*>(L.frag[1] + j * 2 * BP * 16);
      F6_LAYER(16, 16, 1, wB, E3, E4, a.h[4], a.h[5], 0, true, F6_SKIPPRE)
      load_w(wB, a.Wp[6], 16, 256);
      F6_LAYER(20, 16, 0, wA, E4, E5, a.h[5], a.h[6], 1, false, F6_NOPRE)
      load_w(wA, a.Wp[7], 16, 256);
      F6_LAYER(24, 16, 1, wB, E5, E6, a.h[6], a.h[7], 0, false, F6_NOPRE)
      if (!VALUES) load_w(wB, a.Wp[8], 16, 256);             // the 256 feature rows of lin8 (tiles 0..7 of the [feature | sdf] pack)
      F6_LAYER(28, 16, 0, wA, E6, E7, a.h[7], a.h[8], 3, false, F6_NOPRE)
      // drain: epilogue of lin7's last tile (h8 complete in XB after the barrier)
      f6_stage<NT, RT, FULL, false, 16, 0, E7, 0, true>(L, wA, 0, acc[0], acc[1], bq, NT - 1, nt, a.h[8], wave, hi, ring, nullptr);
      // ---- lin8: the sdf row, split over the waves' k-steps and reduced through LDS; then (save mode) the 256 feature rows
      {
        constexpr int KW = 16 / NW;                          // k-steps of the sdf row per wave
        {                                                    // -> the idle set wA
          unsigned voff = (unsigned)((((VALUES ? 0 : 8) * 16 + KW * wave) * 64 + lane) * 16);
          asm volatile("" : "+v"(voff));
#pragma unroll
          for (int j = 0; j < KW; ++j) wA[0][j] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(a.Wp[8]) + voff + j * 1024);
        }
        const unsigned char* fr = L.frag[1] + (unsigned)(KW * wave) * (2 * BP * 16);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          f32x16 accs;
#pragma unroll
          for (int r = 0; r < 16; ++r) accs[r] = 0.0f;
#pragma unroll
          for (int j = 0; j < KW; ++j) {
            const uint4 bv = *reinterpret_cast<const uint4*>(fr + (j * 2 * BP * 16 + t * 512));
            accs = NEAT_MFMA16(*reinterpret_cast<const bf16x8*>(&wA[0][j]), *reinterpret_cast<const bf16x8*>(&bv), accs, 0, 0, 0);
          }
          if (hi == 0) red[wave * BP + t * 32 + lane] = accs[0];
        }
      }
      if (!VALUES) {
#pragma unroll
        for (int j = 0; j < RD - 1; ++j) ring[(F6_KSUM(33) + j) % RD] = *reinterpret_cast<const uint4*>(L.frag[1] + j * 2 * BP * 16);
        F6_LAYER(33, 16, 1, wB, F6NoEpi, E8, nullptr, a.feat, 3, false, F6_NOPRE)
        f6_stage<NT, RT, FULL, false, 16, 0, E8, 0, true>(L, wB, 0, acc[37 & 1], acc[(37 + 1) & 1], bq, NT - 1, nt, a.feat, wave, hi, ring, nullptr);
      } else {
        __syncthreads();
      }
      }
#undef F6_STAGE
#undef F6_KSUM
#undef F6_NOPRE
#undef F6_SKIPPRE
#undef F6_LAYER
    };
    if (nt == NT) chain(std::false_type{});
    else chain(std::true_type{});
    if (tid < nt * 32) {
      float sv = biasl[8 * 256 + (VALUES ? 0 : 256)];
      if (VALUES) {            // h8 arrived in the softplus's scale: the row sum back to the sdf's
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += red[w * BP + tid];
        sv += t * (1.0f / SOFTPLUS_C);
      } else {
#pragma unroll
        for (int w = 0; w < NW; ++w) sv += red[w * BP + tid];
      }
      const int p = p0 + tid;
      if (VALUES) {
        if (a.radius > 0.0f) {
          const float x0 = a.x_fm[p], x1 = a.x_fm[(size_t)a.ldp + p], x2 = a.x_fm[(size_t)2 * a.ldp + p];
          sv = fminf(sv, a.scale * (a.radius - sqrtf(x0 * x0 + x1 * x1 + x2 * x2)));
        }
        if (p < a.P) a.sdf_out[p] = sv;
      } else {
        a.sdfraw[p] = sv;
      }
    }
    __syncthreads();
   }
  }
}

// ===============================================================================================================
// Fused ADJOINT chain (normals: d sdf / d x through the SDF MLP, reference: autograd.grad at neat_wfr_rend_a.py:121-127).
//   u_7 = w8 (.) phi'(h_8);   u_{l-1} = (W_l^T u_l) (.) phi'(h_l)  for l = 7 .. 1  (l = 4: rows 217.. are the PE cotangent of the skip,
//   fp32, no phi');   e0 = W_0^T u_0 (39 fp32 rows).
// Same stage pipeline, LDS ping-pong and register-resident double-buffered weight slices as sdf_fused_w64_kernel (RT = 1), with
// the transposed packs; the chain variable u never leaves the chip between layers.  Per layer and point the kernel reads the
// saved h_l once (8-byte quads in accumulator layout, requested one stage before the epilogue that uses them) and, in SAVE mode
// (training: the tangent chain and the weight gradient read u_l), writes u_{l-1} once: 2 streams per layer where the streaming
// EPI_REV launches move 3 (in, aux, out) -- and 8 launches + the seed kernel become one.
// ===============================================================================================================
#ifndef NEAT_ADJ_SINGLE
#define NEAT_ADJ_SINGLE 1      // 1: the ragged end of a workgroup's tile range goes tile by tile through chain_single (0: as one partial pipeline batch)
#endif
template <bool SAVE>
__global__ __launch_bounds__(512, 2) void sdf_adjoint_w64_kernel(AdjArgs a, int ntiles, int nwg) {
  constexpr int NT = 4, RT = 1;
  typedef F6Cfg<NT, RT> C;
  constexpr int BP = C::BP, F6T = C::THREADS;
  extern __shared__ __attribute__((aligned(16))) unsigned char f6lds[];
  float* seedw = reinterpret_cast<float*>(f6lds + C::BIAS);      // [256]: w8[k] * rs8
  if (NEAT_F6_PRIO == 1 && threadIdx.x >= 256) __builtin_amdgcn_s_setprio(1);
  if (NEAT_F6_PRIO == 2 && threadIdx.x < 256) __builtin_amdgcn_s_setprio(1);
  const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int k = tid; k < 256; k += F6T) seedw[k] = a.w8[k] * a.rs8[0];

  // ONE set of weight fragment registers (64 VGPRs): a layer's last tile refills each slot with the next layer's fragment right after
  // the slot's last MFMA (f6_stage WROLL) -- the double-buffered form of the primal kernel (128 VGPRs) left this kernel, which also
  // carries the h quads, 19 scratch reloads, each an s_waitcnt vmcnt(0) that drains the stores in flight
  uint4 wA[RT][16];
  auto w_addr = [&](const uint4* Wl, int N) -> const unsigned char* {
    const int tile = (RT * wave) * 32 < N ? RT * wave : 0;          // dead row tiles re-read a live one
    unsigned voff = (unsigned)((tile * 16) * 64 + lane) * 16u;
    asm volatile("" : "+v"(voff));
    return reinterpret_cast<const unsigned char*>(Wl) + voff;
  };
  {
    const unsigned char* w7 = w_addr(a.Wp[7], 256);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) wA[0][ks] = *reinterpret_cast<const uint4*>(w7 + ks * 1024);
  }

  F6Lane L;
  {
    const unsigned fo = (unsigned)(hi * BP + (lane & 31)) * 16u, qo = (unsigned)((4 * RT * wave) * BP + (lane & 31)) * 16u + 8u * hi;
    unsigned b0 = fo, b1 = fo + C::XB, q0 = qo, q1 = qo + C::XB, bb = C::BIAS + (unsigned)(32 * RT * wave + 4 * hi) * 4u;
    asm volatile("" : "+v"(b0), "+v"(b1), "+v"(q0), "+v"(q1), "+v"(bb));
    L.frag[0] = f6lds + b0; L.frag[1] = f6lds + b1; L.frag[2] = f6lds + b0;
    L.quad[0] = f6lds + q0; L.quad[1] = f6lds + q1; L.bias = f6lds + bb; L.pe0 = f6lds;
  }
  L.ldp16 = (unsigned)a.ldp * 16u;
  __syncthreads();

  const int t_begin = (int)(((long long)blockIdx.x * ntiles) / nwg), t_end = (int)(((long long)(blockIdx.x + 1) * ntiles) / nwg);
  // A batch = NT tiles through the stage pipeline.  What is left of a workgroup's range (1 .. NT-1 tiles: at 133 120 points 64 of the
  // 256 workgroups own 17 tiles) goes tile by tile through `chain_single` below: a pipeline batch costs its full latency whatever it
  // holds, a single tile about a sixth of it (round 4; the primal chain has had this path since round 2).
  for (int tile0 = t_begin; tile0 < t_end; tile0 += NT) {
   const int ntb = min(NT, t_end - tile0);
   for (int sub = 0; sub < ((ntb == NT || !NEAT_ADJ_SINGLE) ? 1 : ntb); ++sub) {
    const int p0 = (tile0 + sub) * 32;
    const int nt = (ntb == NT || !NEAT_ADJ_SINGLE) ? ntb : 1;
    L.gquad = ((unsigned)(4 * RT * wave) * (unsigned)a.ldp + (unsigned)(p0 + (lane & 31))) * 16u + 8u * hi;
    L.goct = ((unsigned)(4 * RT * wave + hi) * (unsigned)a.ldp + (unsigned)(p0 + (lane & 31))) * 16u;
    L.fcol = (unsigned)(p0 + (lane & 31));
    // the row stride as an opaque per-batch VGPR: with a loop-invariant stride the 64 (array, quad) row bases of the sixteen h / u
    // arrays are hoisted out of the batch loop as scalar pairs and spilled
    L.ldp16 = (unsigned)a.ldp * 16u;
    asm volatile("" : "+v"(L.ldp16));

    auto chain = [&](auto full_tag) {
      constexpr bool FULL = decltype(full_tag)::value;
      // the saved activation quads of this lane for point tile t: rows 32 wave + 8 q + 4 hi .. + 3, raw bf16 (see F6RevCfg)
      auto load_h = [&](float4 (&dst)[4 * RT], const u16* hsrc, int t, bool wide = NEAT_ADJ_WIDE != 0) {
        if (!(FULL || t < nt)) return;
        if (NEAT_ADJ_ABLATE & 2) { _Pragma("unroll") for (int q = 0; q < 4; ++q) { dst[q].x = 0.0f; dst[q].y = 0.0f; } return; }
        if (wide) {      // two 16-byte loads: lanes 0-31 the octets of quads 0 / 2, lanes 32-63 those of quads 1 / 3, kept RAW in dst[0] / dst[2]; the
          // halves the other lane needs are swapped where the epilogue uses them (f6_epi_half) -- a swap here would wait for the load at once
          typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
#pragma unroll
          for (int qq = 0; qq < 4; qq += 2) {
            const v4u_t* hp = reinterpret_cast<const v4u_t*>(reinterpret_cast<const char*>(hsrc) + ((unsigned)qq * L.ldp16 + L.goct) + t * 512);
            const v4u_t o = NEAT_ADJ_NT_LOAD ? __builtin_nontemporal_load(hp) : *hp;
            dst[qq].x = __uint_as_float(o.x); dst[qq].y = __uint_as_float(o.y); dst[qq].z = __uint_as_float(o.z); dst[qq].w = __uint_as_float(o.w);
          }
          return;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {      // kernarg base + one 32-bit per-lane offset (the arrays are < 4 GB) + immediate
          typedef unsigned long long u64_t;
          const u64_t* hp = reinterpret_cast<const u64_t*>(reinterpret_cast<const char*>(hsrc) + ((unsigned)q * L.ldp16 + L.gquad) + t * 512);
          const uint2 v = NEAT_ADJ_NT_LOAD ? __builtin_bit_cast(uint2, __builtin_nontemporal_load(hp)) : *reinterpret_cast<const uint2*>(hp);
          dst[q].x = __uint_as_float(v.x); dst[q].y = __uint_as_float(v.y);
        }
      };
      // Two sets: the quads of a stage's tile are requested one stage before the epilogue that multiplies with them runs.  NEAT_ADJ_HDEPTH = 3
      // (probe): three sets, requested two stages ahead -- 32 KiB instead of 16 KiB per CU in flight; measured 315 -> 338 us per full-size
      // launch (the rolling weight refills share the in-order vector-memory counter with the longer queue of h requests)
#ifndef NEAT_ADJ_HDEPTH
#define NEAT_ADJ_HDEPTH 2
#endif
      constexpr int HD = NEAT_ADJ_HDEPTH;
      float4 hq[HD][4 * RT];
      // ---- seed: u_7 = w8 (.) phi'(h_8) -> XA (+ HBM)
      {
        float4 wq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) wq[q] = *reinterpret_cast<const float4*>(L.bias + (8 * q) * 4);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          if (!(FULL || t < nt)) {
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<uint2*>(L.quad[0] + (q * BP + t * 32) * 16) = make_uint2(0u, 0u);
            continue;
          }
          load_h(hq[0], a.h[8], t, false);      // (the seed reads its quads directly: 8-byte loads)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const unsigned h0 = __float_as_uint(hq[0][q].x), h1 = __float_as_uint(hq[0][q].y);
            const uint2 v = make_uint2(pack2(wq[q].x * dphi_fast(bf_lo(h0)), wq[q].y * dphi_fast(bf_hi(h0))),
                                       pack2(wq[q].z * dphi_fast(bf_lo(h1)), wq[q].w * dphi_fast(bf_hi(h1))));
            *reinterpret_cast<uint2*>(L.quad[0] + (q * BP + t * 32) * 16) = v;
            if (SAVE) *reinterpret_cast<uint2*>(reinterpret_cast<char*>(a.u[7]) + ((unsigned)q * L.ldp16 + L.gquad) + t * 512) = v;
          }
        }
      }
      __syncthreads();

      typedef F6RevCfg<SAVE, 256, 1, 1 << 30> R7;      // l = 7: XA -> XB, * phi'(h_7), -> u_6
      typedef F6RevCfg<SAVE, 256, 0, 1 << 30> R6;
      typedef F6RevCfg<SAVE, 256, 1, 1 << 30> R5;
      typedef F6RevCfg<SAVE, 256, 0, 217> R4;          // l = 4: rows 217 .. 255 -> es (PE cotangent of the skip)
      typedef F6RevCfg<SAVE, 256, 1, 1 << 30> R3;
      typedef F6RevCfg<SAVE, 256, 0, 1 << 30> R2;
      typedef F6RevCfg<SAVE, 256, 1, 1 << 30> R1;
      typedef F6RevCfg<false, 39, 0, 0> R0;            // l = 0: e0 = W_0^T u_0, 39 fp32 rows
      f32x16 acc[2][RT];
      uint4 ring[NEAT_F6_RING];
      constexpr int RD = NEAT_F6_RING;
#define ADJ_STAGE(S_, T_, SRC_, WC_, E_, H_, TE_, NFR_)                                                                              \
      f6_stage<NT, RT, FULL, true, 16, SRC_, E_, (16 * (S_)) % RD, ((S_) % 2 == 1)>(L, WC_, T_, acc[(S_) & 1], acc[((S_) + 1) & 1], hq[((S_) + HD - 1) % HD], TE_, nt, H_, wave, hi, ring, NFR_);
#define ADJ_STAGE_ROLL(S_, T_, SRC_, WC_, E_, H_, TE_, NFR_, WNX_)                                                                   \
      f6_stage<NT, RT, FULL, true, 16, SRC_, E_, (16 * (S_)) % RD, ((S_) % 2 == 1), true>(L, WC_, T_, acc[(S_) & 1], acc[((S_) + 1) & 1], hq[((S_) + HD - 1) % HD], TE_, nt, H_, wave, hi, ring, NFR_, WNX_);
      // HSRC_: the h array whose phi' multiplies this layer's output (null: none); FPREV_ / FCUR_: fp32 row destinations of the previous / this layer
#define ADJ_LAYER(S0_, SRC_, WC_, EPREV_, ECUR_, HPREV_, HCUR_, HAS_H_, HSRC_, NSRC_, FPREV_, FCUR_, WNEXT_, NNEXT_, HASN_, HNEXT_)             \
      { L.frows = FPREV_;                                                                                                         \
        if (HD == 2 && HAS_H_) load_h(hq[((S0_) + 0) % HD], HSRC_, 0);                                                              \
        if (HD == 3 && HAS_H_) load_h(hq[((S0_) + 1) % HD], HSRC_, 1);                                                              \
        ADJ_STAGE((S0_) + 0, 0, SRC_, WC_, EPREV_, HPREV_, NT - 1, L.frag[SRC_] + 1 * 512)                                         \
        if (HD == 4 && HAS_H_) load_h(hq[((S0_) + 2) % HD], HSRC_, 2);                                                              \
        L.frows = FCUR_;                                                                                                          \
        if (HD == 2 && HAS_H_) load_h(hq[((S0_) + 1) % HD], HSRC_, 1);                                                              \
        if (HD == 3 && HAS_H_) load_h(hq[((S0_) + 2) % HD], HSRC_, 2);                                                              \
        ADJ_STAGE((S0_) + 1, 1, SRC_, WC_, ECUR_, HCUR_, 0, L.frag[SRC_] + 2 * 512)                                                \
        if (HD == 4 && HAS_H_) load_h(hq[((S0_) + 3) % HD], HSRC_, 3);                                                              \
        if (HD == 2 && HAS_H_) load_h(hq[((S0_) + 2) % HD], HSRC_, 2);                                                              \
        if (HD == 3 && HAS_H_) load_h(hq[((S0_) + 3) % HD], HSRC_, 3);                                                              \
        ADJ_STAGE((S0_) + 2, 2, SRC_, WC_, ECUR_, HCUR_, 1, L.frag[SRC_] + 3 * 512)                                                \
        if (HD == 4 && HASN_) load_h(hq[((S0_) + 4) % HD], HNEXT_, 0);                                                              \
        if (HD == 2 && HAS_H_) load_h(hq[((S0_) + 3) % HD], HSRC_, 3);                                                              \
        if (HD == 3 && HASN_) load_h(hq[((S0_) + 4) % HD], HNEXT_, 0);                                                              \
        ADJ_STAGE_ROLL((S0_) + 3, 3, SRC_, WC_, ECUR_, HCUR_, 2, ((NSRC_) < 2 ? L.frag[(NSRC_) < 2 ? (NSRC_) : 0] : nullptr), w_addr(WNEXT_, NNEXT_)) \
        if (HD == 4 && HASN_) load_h(hq[((S0_) + 5) % HD], HNEXT_, 1); }
#pragma unroll
      for (int j = 0; j < RD - 1; ++j) ring[j] = *reinterpret_cast<const uint4*>(L.frag[0] + j * 2 * BP * 16);
      if (HD == 3) load_h(hq[0], a.h[7], 0);
      if (HD == 4) { load_h(hq[0], a.h[7], 0); load_h(hq[1], a.h[7], 1); }      // HD = 4: the quads of the tile whose MFMAs run two stages later, requested BEHIND a stage (and its rolling weight refills)
      ADJ_LAYER(0, 0, wA, F6NoEpi, R7, nullptr, a.u[6], 1, a.h[7], 1, nullptr, nullptr, a.Wp[6], 256, 1, a.h[6])
      ADJ_LAYER(4, 1, wA, R7, R6, a.u[6], a.u[5], 1, a.h[6], 0, nullptr, nullptr, a.Wp[5], 256, 1, a.h[5])
      ADJ_LAYER(8, 0, wA, R6, R5, a.u[5], a.u[4], 1, a.h[5], 1, nullptr, nullptr, a.Wp[4], 256, 1, a.h[4])
      ADJ_LAYER(12, 1, wA, R5, R4, a.u[4], a.u[3], 1, a.h[4], 0, nullptr, a.es, a.Wp[3], 256, 1, a.h[3])
      ADJ_LAYER(16, 0, wA, R4, R3, a.u[3], a.u[2], 1, a.h[3], 1, a.es, nullptr, a.Wp[2], 256, 1, a.h[2])
      ADJ_LAYER(20, 1, wA, R3, R2, a.u[2], a.u[1], 1, a.h[2], 0, nullptr, nullptr, a.Wp[1], 256, 1, a.h[1])
      ADJ_LAYER(24, 0, wA, R2, R1, a.u[1], a.u[0], 1, a.h[1], 1, nullptr, nullptr, a.Wp[0], 39, 0, a.h[1])
      ADJ_LAYER(28, 1, wA, R1, R0, a.u[0], nullptr, 0, a.h[1], 2, nullptr, a.e0, a.Wp[7], 256, 0, a.h[1])      // (the next batch starts with W_7 again)
      // drain: the last tile of the last layer
      f6_stage<NT, RT, FULL, false, 16, 0, R0, 0, true>(L, wA, 0, acc[0], acc[1], hq[(32 + HD - 1) % HD], NT - 1, nt, nullptr, wave, hi, ring, nullptr);
#undef ADJ_LAYER
#undef ADJ_STAGE_ROLL
#undef ADJ_STAGE
    };
    // ---- one tile: per layer the 16 k-steps (the next layer's weight slice rolls into the registers behind them), then the whole
    // epilogue, then the barrier that publishes the tile
    auto chain_single = [&]() {
      constexpr bool FULL = false;
      auto load_h = [&](float4 (&dst)[4 * RT], const u16* hsrc, bool wide = NEAT_ADJ_WIDE != 0) {
        if (wide) {      // the raw octets f6_epi_half expects (see `chain` above)
#pragma unroll
          for (int qq = 0; qq < 4; qq += 2) {
            const uint4 o = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(hsrc) + ((unsigned)qq * L.ldp16 + L.goct));
            dst[qq].x = __uint_as_float(o.x); dst[qq].y = __uint_as_float(o.y); dst[qq].z = __uint_as_float(o.z); dst[qq].w = __uint_as_float(o.w);
          }
          return;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint2 v = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(hsrc) + ((unsigned)q * L.ldp16 + L.gquad));
          dst[q].x = __uint_as_float(v.x); dst[q].y = __uint_as_float(v.y);
        }
      };
      float4 hq[4 * RT];
      {
        float4 wq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) wq[q] = *reinterpret_cast<const float4*>(L.bias + (8 * q) * 4);
        load_h(hq, a.h[8], false);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const unsigned h0 = __float_as_uint(hq[q].x), h1 = __float_as_uint(hq[q].y);
          const uint2 v = make_uint2(pack2(wq[q].x * dphi_fast(bf_lo(h0)), wq[q].y * dphi_fast(bf_hi(h0))),
                                     pack2(wq[q].z * dphi_fast(bf_lo(h1)), wq[q].w * dphi_fast(bf_hi(h1))));
          *reinterpret_cast<uint2*>(L.quad[0] + (q * BP) * 16) = v;
          if (SAVE) *reinterpret_cast<uint2*>(reinterpret_cast<char*>(a.u[7]) + ((unsigned)q * L.ldp16 + L.gquad)) = v;
        }
      }
      __syncthreads();
      typedef F6RevCfg<SAVE, 256, 1, 1 << 30> R7;
      typedef F6RevCfg<SAVE, 256, 0, 1 << 30> R6;
      typedef F6RevCfg<SAVE, 256, 1, 1 << 30> R5;
      typedef F6RevCfg<SAVE, 256, 0, 217> R4;
      typedef F6RevCfg<SAVE, 256, 1, 1 << 30> R3;
      typedef F6RevCfg<SAVE, 256, 0, 1 << 30> R2;
      typedef F6RevCfg<SAVE, 256, 1, 1 << 30> R1;
      typedef F6RevCfg<false, 39, 0, 0> R0;
      f32x16 acc[2][RT];
      uint4 ring[NEAT_F6_RING];
      constexpr int RD = NEAT_F6_RING;
#define ADJ_LAYER1(SRC_, ECUR_, HCUR_, HAS_H_, HSRC_, FCUR_, WNEXT_, NNEXT_)                                                          \
      { _Pragma("unroll") for (int j = 0; j < RD - 1; ++j) ring[j] = *reinterpret_cast<const uint4*>(L.frag[SRC_] + j * 2 * BP * 16);  \
        if (HAS_H_) load_h(hq, HSRC_);                                                                                              \
        L.frows = FCUR_;                                                                                                            \
        f6_stage<NT, RT, FULL, true, 16, SRC_, F6NoEpi, 0, false, true>(L, wA, 0, acc[0], acc[1], hq, 0, 1, nullptr, wave, hi, ring, nullptr, w_addr(WNEXT_, NNEXT_)); \
        f6_stage<NT, RT, FULL, false, 16, 0, ECUR_, 0, true>(L, wA, 0, acc[1], acc[0], hq, 0, 1, HCUR_, wave, hi, ring, nullptr); }
      ADJ_LAYER1(0, R7, a.u[6], 1, a.h[7], nullptr, a.Wp[6], 256)
      ADJ_LAYER1(1, R6, a.u[5], 1, a.h[6], nullptr, a.Wp[5], 256)
      ADJ_LAYER1(0, R5, a.u[4], 1, a.h[5], nullptr, a.Wp[4], 256)
      ADJ_LAYER1(1, R4, a.u[3], 1, a.h[4], a.es, a.Wp[3], 256)
      ADJ_LAYER1(0, R3, a.u[2], 1, a.h[3], nullptr, a.Wp[2], 256)
      ADJ_LAYER1(1, R2, a.u[1], 1, a.h[2], nullptr, a.Wp[1], 256)
      ADJ_LAYER1(0, R1, a.u[0], 1, a.h[1], nullptr, a.Wp[0], 39)
      ADJ_LAYER1(1, R0, nullptr, 0, a.h[1], a.e0, a.Wp[7], 256)       // (the next tile or batch starts with W_7 again)
#undef ADJ_LAYER1
    };
#ifndef NEAT_ADJ_ONE_VARIANT
#define NEAT_ADJ_ONE_VARIANT 1
#endif
    // one variant of the chain for full and partial batches (per-tile `t < nt` predicates on the loads / stores): with a second,
    // predicate-free copy for full batches the rolling weight registers are carried into both copies and the allocator spills them
    if (NEAT_ADJ_SINGLE && ntb != NT) chain_single();
    else if (NEAT_ADJ_ONE_VARIANT) chain(std::false_type{});
    else if (nt == NT) chain(std::true_type{});
    else chain(std::false_type{});
    __syncthreads();
   }
  }
}

// ===============================================================================================================
// Phase-staggered fused primal chain (generation 4).
//
// What the ablations of the stage-pipelined kernel above showed (scripts/probes/probe_fused_abl.py, values mode, P = 133 120):
// MFMAs + B-fragment reads alone 81 us (4 waves x 64 rows) = the matrix-pipe time; the epilogue alone 74 us; both in one
// instruction stream 236 us -- more than their sum, whatever the interleave (fine, per 2 / 4 / 16 k-steps), the fragment
// ring depth, the accumulator register class or the barrier count.  A wave that mixes MFMAs with VALU / LDS work runs
// both at a fraction of their rates; two such waves on one SIMD do not fix it, because they mix in lock step.
// What does overlap on this machine is a wave in a DENSE matrix phase next to a wave in a DENSE vector phase on the same
// SIMD ("matrix beside VALU", MI355X_MICROARCH.md, two waves per SIMD).  So here:
//   * eight waves, wave w owns output rows 32 w .. 32 w + 31 (weights: 16 fragments, double-buffered = 128 registers);
//   * waves 0..3 (role 0) and 4..7 (role 1) -- one of each per SIMD -- run the same sequence of phases one phase apart:
//       phase 2t   : role 0 multiplies tile t (16 back-to-back MFMAs on one accumulator tile, B fragments through a ring),
//                    role 1 runs the epilogue of ITS previous tile (bias, softplus, pack, LDS / HBM stores);
//       phase 2t+1 : the other way round;
//     one workgroup barrier per phase keeps the two roles in opposite phases;
//   * layer l+1 follows layer l without a bubble (its tile 0 was finished six phases earlier).
// ===============================================================================================================
constexpr int PHT = 512;
typedef F6Cfg<4, 1> PhCfg;

// dense matrix phase: acc = W_slice (32 x 16 KS) x input tile t of region SRC.  The first RD-1 fragments are already in `ring`
// (requested at the end of this wave's previous vector phase).
template <int KS, int SRC>
__device__ __forceinline__ void ph_mma(const F6Lane& L, const uint4 (&wc)[16], int t, f32x16& acc, uint4 (&ring)[4]) {
  constexpr int STEP = 2 * PhCfg::BP * 16, RD = 4, AH = 3;
  const unsigned char* fr = L.frag[SRC] + t * 512;
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    if (ks + AH < KS) ring[(ks + AH) % RD] = *reinterpret_cast<const uint4*>(fr + (ks + AH) * STEP);
    const uint4 cur = ring[ks % RD];
    acc = NEAT_MFMA16(*reinterpret_cast<const bf16x8*>(&wc[ks]), *reinterpret_cast<const bf16x8*>(&cur), ks ? acc : zero, 0, 0, 0);
  }
}
// request the first fragments of the next matrix phase (tile t of region SRC; KS >= 3)
template <int SRC> __device__ __forceinline__ void ph_prefetch(const F6Lane& L, int t, uint4 (&ring)[4]) {
  constexpr int STEP = 2 * PhCfg::BP * 16;
  const unsigned char* fr = L.frag[SRC] + t * 512;
#pragma unroll
  for (int j = 0; j < 3; ++j) ring[j] = *reinterpret_cast<const uint4*>(fr + j * STEP);
}

// dense vector phase: epilogue E of the accumulator tile `acc` (point tile t)
template <bool FULL, class E>
__device__ __forceinline__ void ph_epi(const F6Lane& L, const f32x16& acc, const float4 (&bq)[4], int t, int nt, u16* hout, int wave, int hi) {
  constexpr int BP = PhCfg::BP;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float b[4] = {bq[q].x, bq[q].y, bq[q].z, bq[q].w};
    float r[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float x = acc[4 * q + e];
      if (E::ACT) {
        const float u = fmaf(x, SOFTPLUS_C, b[e]);
        r[e] = (fmaxf(u, 0.0f) + __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(-fabsf(u)))) * 0.0069314718055994531f;
      } else {
        r[e] = x + b[e];
      }
    }
    uint2 v = make_uint2(pack2(r[0], r[1]), pack2(r[2], r[3]));
    if (E::N == 217 && q == 3) {
      // lin3, rows 216..223 (wave 6): [h216 | PE rows 0..6] -- what lin4 (skip connection, rend_a :87-88) and the saved h4 expect in
      // the last octet of the 217-row array
      if (wave == 6) {
        const uint4 w = *reinterpret_cast<const uint4*>(L.pe0 + t * 512);       // PE rows 0..7 of this point, bf16
        if (hi == 0) v = make_uint2((v.x & 0xFFFFu) | (w.x << 16), (w.x >> 16) | (w.y << 16));
        else v = make_uint2((w.y >> 16) | (w.z << 16), (w.z >> 16) | (w.w << 16));
      }
    }
    if (E::ACT) *reinterpret_cast<uint2*>(L.quad[E::DST] + (q * BP + t * 32) * 16) = v;
    if (E::SAVE && NEAT_F6_ABLATE != 9 && (FULL || t < nt))                     // wave-uniform row base + per-lane 32-bit offset + immediate
      *reinterpret_cast<uint2*>(reinterpret_cast<char*>(hout) + ((size_t)(q * L.ldp16) + t * 512) + (size_t)L.gquad) = v;
  }
}
template <class E> __device__ __forceinline__ void ph_load_bias(const F6Lane& L, float4 (&bq)[4]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const float4*>(L.bias + (E::BIASOFF + 8 * q) * 4);
}

template <bool VALUES>
__global__ __launch_bounds__(PHT, 2) void sdf_fused_ph_kernel(FusedArgs a, int ntiles, int nwg) {
  typedef PhCfg C;
  constexpr int BP = C::BP, NT = 4, NW = 8;
  if (a.gate && *a.gate != a.gate_value) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char f6lds[];
  float* biasl = reinterpret_cast<float*>(f6lds + C::BIAS);     // [l][256]; lin8 in packed row order
  float* red = reinterpret_cast<float*>(f6lds + C::RED);        // [waves][BP]: partial sums of the sdf row
  u16* pe16 = reinterpret_cast<u16*>(f6lds + C::PE);
  const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int role = wave >> 2;                  // waves w and w + 4 share a SIMD (round-robin placement): one of each role per SIMD
  constexpr bool SAVE = !VALUES;

  for (int idx = tid; idx < 8 * 256; idx += PHT) {
    const int l = idx >> 8, n = idx & 255;
    float v = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) if (l == k && n < (k == 3 ? 217 : 256)) v = a.bias[k][n];
    biasl[idx] = v * SOFTPLUS_C;             // hidden layers: pre-scaled for the softplus epilogue
  }
  for (int n = tid; n < 257; n += PHT) {
    int bi = n + a.bias8_rot; if (bi >= a.bias8_n) bi -= a.bias8_n;
    biasl[8 * 256 + n] = VALUES ? (n == 0 ? a.bias[8][0] : 0.0f) : a.bias[8][bi];
  }
  // PE rows 39..63 (octets 4..7; rows 32..38 are rewritten per batch) stay zero for the whole launch
  for (int idx = tid; idx < 4 * BP; idx += PHT) reinterpret_cast<uint4*>(f6lds + C::PE)[4 * BP + idx] = make_uint4(0u, 0u, 0u, 0u);

  uint4 wA[16], wB[16];
  auto load_w = [&](uint4 (&dst)[16], const uint4* Wl, int KS, int N) {
    // kernarg pointer (SGPR pair) + ONE opaque 32-bit per-lane offset + immediate (see sdf_fused_w64_kernel).  The dead row tile
    // of lin3 (rows 224..255) re-reads tile 0; it is never multiplied.
    const char* base = reinterpret_cast<const char*>(Wl);
    const int tile = wave * 32 < N ? wave : 0;
    unsigned voff = (unsigned)((tile * KS) * 64 + lane) * 16u;
    asm volatile("" : "+v"(voff));
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
      if (ks < KS) dst[ks] = *reinterpret_cast<const uint4*>(base + voff + ks * 1024);
  };

  F6Lane L;
  {
    const unsigned fo = (unsigned)(hi * BP + (lane & 31)) * 16u, qo = (unsigned)((4 * wave) * BP + (lane & 31)) * 16u + 8u * hi;
    unsigned b0 = fo, b1 = fo + C::XB, b2 = fo + C::PE, q0 = qo, q1 = qo + C::XB, bb = C::BIAS + (unsigned)(32 * wave + 4 * hi) * 4u;
    unsigned pz = C::PE + (unsigned)(lane & 31) * 16u;
    asm volatile("" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(q0), "+v"(q1), "+v"(bb), "+v"(pz));
    L.frag[0] = f6lds + b0; L.frag[1] = f6lds + b1; L.frag[2] = f6lds + b2;
    L.quad[0] = f6lds + q0; L.quad[1] = f6lds + q1; L.bias = f6lds + bb; L.pe0 = f6lds + pz;
  }
  L.ldp16 = (unsigned)a.ldp * 16u;

  const bool inter = nwg < 0;
  const int ng = inter ? -nwg : nwg;
  const int t_begin = inter ? (int)blockIdx.x * NT : (int)(((long long)blockIdx.x * ntiles) / ng);
  const int t_end = inter ? ntiles : (int)(((long long)(blockIdx.x + 1) * ntiles) / ng);
  const int t_step = inter ? ng * NT : NT;
  for (int tile0 = t_begin; tile0 < t_end; tile0 += t_step) {
    const int p0 = tile0 * 32;
    const int nt = min(NT, t_end - tile0);
    L.gquad = ((unsigned)(4 * wave) * (unsigned)a.ldp + (unsigned)(p0 + (lane & 31))) * 16u + 8u * hi;

    auto chain = [&](auto full_tag) {
      constexpr bool FULL = decltype(full_tag)::value;
      load_w(wB, a.Wp[0], 4, 256);
      load_w(wA, a.Wp[1], 16, 256);
      // ---- positional encoding (embedder.py:12-36): thread = (point, group of 2 frequencies; the 4th group writes x itself)
      {
        const int p = tid & (BP - 1), fg = __builtin_amdgcn_readfirstlane(tid / BP);      // (BP >= 64: the group is wave-uniform)
        const bool ok = p < nt * 32;
        float xc[3];
        // rows of x_fm / E: kernarg base (SGPR pair) + ONE 32-bit per-lane byte offset computed where it is used from an opaque
        // copy of ldp.  (With 64-bit row addresses the 13 row pointers of a thread -- its rows depend on its frequency group --
        // were hoisted out of the batch loop and spilled, and every reload carried an s_waitcnt vmcnt(0) that serialised the E
        // stores: +75 us per launch in save mode.)
        unsigned pvo = (unsigned)(p0 + p) * 4u, ldp4 = (unsigned)a.ldp * 4u;
        asm volatile("" : "+v"(pvo), "+v"(ldp4));
#pragma unroll
        for (int c = 0; c < 3; ++c)
          xc[c] = ok ? *reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.x_fm) + ((unsigned)c * ldp4 + pvo)) : 0.0f;
        auto put = [&](int j, float v) {
          pe16[((j >> 3) * BP + p) * 8 + (j & 7)] = f2bf(v);
          if (SAVE && ok) { if (NEAT_F6_NT_E) __builtin_nontemporal_store(v, reinterpret_cast<float*>(reinterpret_cast<char*>(a.E) + ((unsigned)j * ldp4 + pvo))); else *reinterpret_cast<float*>(reinterpret_cast<char*>(a.E) + ((unsigned)j * ldp4 + pvo)) = v; }
        };
        if (fg == 3) {
#pragma unroll
          for (int c = 0; c < 3; ++c) put(c, xc[c]);
        } else {
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            const int k = fg * 2 + kk;
            const float f = (float)(1 << k);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              const float arg = xc[c] * f;
              put(3 + 6 * k + c, ok ? __sinf(arg) : 0.0f);
              put(6 + 6 * k + c, ok ? __cosf(arg) : 0.0f);
            }
          }
        }
      }
      __syncthreads();

      typedef F6EpiCfg<true, SAVE, 256, 0, 0 * 256> E0;      // lin0 -> XA
      typedef F6EpiCfg<true, SAVE, 256, 1, 1 * 256> E1;      // lin1 -> XB
      typedef F6EpiCfg<true, SAVE, 256, 0, 2 * 256> E2;
      typedef F6EpiCfg<true, SAVE, 217, 1, 3 * 256> E3;      // lin3 -> XB rows 0..216 (+ PE rows 0..6 in the last octet)
      typedef F6EpiCfg<true, SAVE, 256, 0, 4 * 256> E4;
      typedef F6EpiCfg<true, SAVE, 256, 1, 5 * 256> E5;
      typedef F6EpiCfg<true, SAVE, 256, 0, 6 * 256> E6;
      typedef F6EpiCfg<true, SAVE, 256, 1, 7 * 256> E7;
      typedef F6EpiCfg<false, true, 256, 0, 8 * 256> E8;     // lin8 feature rows -> HBM only
      f32x16 acc;
      float4 bq[4];
      uint4 ring[4];
      // One layer = 2 NT phases.  Role 0: phase 2t = matrix phase of tile t, phase 2t+1 = its epilogue.  Role 1 runs one phase
      // behind: phase 2t = epilogue of its previous tile (tile t-1, or tile 3 of the previous layer: EPREV_ with the previous
      // layer's bias still in bq), phase 2t+1 = matrix phase of tile t.  LIVE_ / LIVEPREV_: this wave's row tile exists in the layer /
      // in the previous one.
      // After a vector phase the wave requests the first fragments of its next matrix phase (NSRC_ = input region of the next layer).
#define PH_LAYER(KS_, SRC_, WC_, EPREV_, ECUR_, HPREV_, HCUR_, NSRC_, LIVE_, LIVEPREV_)                                                     \
      _Pragma("unroll") for (int t = 0; t < NT; ++t) {                                                                          \
        if (role == 0) {                                                                                                        \
          if (LIVE_) ph_mma<KS_, SRC_>(L, WC_, t, acc, ring);                                                                   \
          if (t == 0) ph_load_bias<ECUR_>(L, bq);                                                                               \
        } else {                                                                                                                \
          if (t == 0) { if (!EPREV_::NONE && (LIVEPREV_)) ph_epi<FULL, EPREV_>(L, acc, bq, NT - 1, nt, HPREV_, wave, hi); ph_load_bias<ECUR_>(L, bq); } \
          else if (LIVE_) ph_epi<FULL, ECUR_>(L, acc, bq, t - 1, nt, HCUR_, wave, hi);                                          \
          ph_prefetch<SRC_>(L, t, ring);                                                                                        \
        }                                                                                                                       \
        __syncthreads();                                                                                                        \
        if (role == 0) {                                                                                                        \
          if (LIVE_) ph_epi<FULL, ECUR_>(L, acc, bq, t, nt, HCUR_, wave, hi);                                                   \
          if (t + 1 < NT) ph_prefetch<SRC_>(L, t + 1, ring); else if ((NSRC_) < 3) ph_prefetch<((NSRC_) < 3 ? (NSRC_) : 0)>(L, 0, ring); \
        } else {                                                                                                                \
          if (LIVE_) ph_mma<KS_, SRC_>(L, WC_, t, acc, ring);                                                                   \
        }                                                                                                                       \
        __syncthreads();                                                                                                        \
      }
      if (role == 0) ph_prefetch<2>(L, 0, ring);
      PH_LAYER(4, 2, wB, F6NoEpi, E0, nullptr, a.h[1], 0, true, true)
      load_w(wB, a.Wp[2], 16, 256);
      PH_LAYER(16, 0, wA, E0, E1, a.h[1], a.h[2], 1, true, true)
      load_w(wA, a.Wp[3], 16, 217);
      PH_LAYER(16, 1, wB, E1, E2, a.h[2], a.h[3], 0, true, true)
      load_w(wB, a.Wp[4], 16, 256);
      // skip connection (rend_a :87-88): rows 224..255 of lin4's input (octets 28..31 of XB) = PE rows 7..38 (the 1/sqrt2 is folded
      // into W4); nobody reads XB between lin2 (done) and lin4, and lin3 writes only rows 0..223 of it (its row tile 7 is dead).
      // One thread per (point, octet): PE rows 7+8k .. 14+8k straddle PE octets k and k+1.
      {
        const int pp = tid & (BP - 1), k = tid >> 7;
        const uint4* pe = reinterpret_cast<const uint4*>(f6lds + C::PE);
        const uint4 lo = pe[k * BP + pp], hi4 = pe[(k + 1) * BP + pp];
        reinterpret_cast<uint4*>(f6lds + C::XB)[(28 + k) * BP + pp] =
            make_uint4((lo.w >> 16) | (hi4.x << 16), (hi4.x >> 16) | (hi4.y << 16), (hi4.y >> 16) | (hi4.z << 16), (hi4.z >> 16) | (hi4.w << 16));
      }
      PH_LAYER(16, 0, wA, E2, E3, a.h[3], a.h[4], 1, (wave != 7), true)
      load_w(wA, a.Wp[5], 16, 256);
      PH_LAYER(16, 1, wB, E3, E4, a.h[4], a.h[5], 0, true, (wave != 7))
      load_w(wB, a.Wp[6], 16, 256);
      PH_LAYER(16, 0, wA, E4, E5, a.h[5], a.h[6], 1, true, true)
      load_w(wA, a.Wp[7], 16, 256);
      PH_LAYER(16, 1, wB, E5, E6, a.h[6], a.h[7], 0, true, true)
      if (!VALUES) load_w(wB, a.Wp[8], 16, 256);             // the 256 feature rows of lin8 (tiles 0..7 of the [feature | sdf] pack)
      PH_LAYER(16, 0, wA, E6, E7, a.h[7], a.h[8], 3, true, true)
      // role 1's last epilogue of lin7 (h8 complete in XB after the barrier)
      if (role == 1) ph_epi<FULL, E7>(L, acc, bq, NT - 1, nt, a.h[8], wave, hi);
      __syncthreads();
      // ---- lin8: the sdf row, split over the waves' k-steps and reduced through LDS; then (save mode) the 256 feature rows
      {
        constexpr int KW = 16 / NW;                          // k-steps of the sdf row per wave
        {                                                    // -> the idle set wA
          unsigned voff = (unsigned)((((VALUES ? 0 : 8) * 16 + KW * wave) * 64 + lane) * 16);
          asm volatile("" : "+v"(voff));
#pragma unroll
          for (int j = 0; j < KW; ++j) wA[j] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(a.Wp[8]) + voff + j * 1024);
        }
        const unsigned char* fr = L.frag[1] + (unsigned)(KW * wave) * (2 * BP * 16);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          f32x16 accs;
#pragma unroll
          for (int r = 0; r < 16; ++r) accs[r] = 0.0f;
#pragma unroll
          for (int j = 0; j < KW; ++j) {
            const uint4 bv = *reinterpret_cast<const uint4*>(fr + (j * 2 * BP * 16 + t * 512));
            accs = NEAT_MFMA16(*reinterpret_cast<const bf16x8*>(&wA[j]), *reinterpret_cast<const bf16x8*>(&bv), accs, 0, 0, 0);
          }
          if (hi == 0) red[wave * BP + t * 32 + lane] = accs[0];
        }
      }
      if (!VALUES) {
        if (role == 0) ph_prefetch<1>(L, 0, ring);
        PH_LAYER(16, 1, wB, F6NoEpi, E8, nullptr, a.feat, 3, true, true)
        if (role == 1) ph_epi<FULL, E8>(L, acc, bq, NT - 1, nt, a.feat, wave, hi);
      }
      __syncthreads();
#undef PH_LAYER
    };
    if (VALUES || nt == NT) chain(std::true_type{});       // (values mode stores nothing per tile: the full-batch code serves every batch)
    else chain(std::false_type{});
    if (tid < nt * 32) {
      float sv = biasl[8 * 256 + (VALUES ? 0 : 256)];
#pragma unroll
      for (int w = 0; w < NW; ++w) sv += red[w * BP + tid];
      const int p = p0 + tid;
      if (VALUES) {
        if (a.radius > 0.0f) {
          const float x0 = a.x_fm[p], x1 = a.x_fm[(size_t)a.ldp + p], x2 = a.x_fm[(size_t)2 * a.ldp + p];
          sv = fminf(sv, a.scale * (a.radius - sqrtf(x0 * x0 + x1 * x1 + x2 * x2)));
        }
        if (p < a.P) a.sdf_out[p] = sv;
      } else {
        a.sdfraw[p] = sv;
      }
    }
    __syncthreads();
  }
}

}  // namespace neat

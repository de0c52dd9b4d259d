// Second-generation fused chains of the bf16 build: FOUR waves per workgroup, one workgroup per CU, each wave owns
// 64 output rows of every 256-wide layer.
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

constexpr int F6T = 256;
template <int NT> struct F6Cfg {
  static constexpr int BP = 32 * NT;
  static constexpr int XBYTES = 32 * BP * 16;            // one activation buffer [32 octets][BP][16 B]
  static constexpr int PEBYTES = 8 * BP * 16;            // PE octets (K padded to 64); reused for the sdf partial sums
  static constexpr int BIAS_FLOATS = 9 * 256 + 8;
  static constexpr int XA = 0, XB = XBYTES, PE = 2 * XBYTES, BIAS = 2 * XBYTES + PEBYTES;     // byte offsets in the LDS block
  static constexpr int LDS = 2 * XBYTES + PEBYTES + BIAS_FLOATS * 4;
};

// Per-lane base addresses of one wave; every access of the layer loop is  base + compile-time offset  that fits the
// instruction's immediate field (ds: 16 bits, so one base per 64 KiB LDS region; global: wave-uniform SGPR base + 32-bit
// per-lane offset + 12 bits), so nothing address-like is recomputed -- or hoisted out of the batch loop and spilled --
// per quad, tile or layer.  The bases are made opaque (empty asm) so that the compiler does not re-derive them from one
// another with constants that do not fit.
struct F6Lane {
  const unsigned char* frag[3];   // B-fragment reads from XA / XB / PE:  region + (hi * BP + (lane & 31)) * 16
  unsigned char* quad[2];         // accumulator-quad writes into XA / XB: region + ((8 wave) * BP + (lane & 31)) * 16 + 8 hi
  const unsigned char* bias;      // bias float4 reads: BIAS + (64 wave + 4 hi) * 4
  unsigned gquad;                 // HBM quad store:   ((8 wave) * ldp + p0 + (lane & 31)) * 16 + 8 hi      (per batch)
  unsigned ldp16;                 // ldp * 16
};

// One layer for one wave: rows 64 wave .. 64 wave + 63 of  act(W src + b)  over the nt point tiles of the batch.
//   ACT: softplus_100 -> LDS buffer DST (input of the next layer) and, if hout, HBM;  !ACT (lin8's feature rows): bias only,
//   HBM only.  N = 217 (lin3) masks the rows the PE copy will occupy.  The epilogue of point tile t-1 is interleaved quad by
//   quad with the 2 x KS MFMAs of point tile t.
// FULL: all NT point tiles are valid (straight-line code, no per-tile guards); otherwise the first nt (last batch of a workgroup).
template <int NT, bool FULL, int KS, int N, bool ACT, bool SAVE, int SRC, int DST, int BIASOFF>      // SRC: 0 = XA, 1 = XB, 2 = PE; DST: 0 = XA, 1 = XB
__device__ __forceinline__ void f6_layer(const F6Lane& L, const uint4 (&wc)[2][16], int nt, u16* hout, int wave, int hi) {
  typedef F6Cfg<NT> C;
  constexpr int BP = C::BP;
  const bool live1 = N == 256 || (2 * wave + 1) * 32 < N;      // lin3: row tile 7 (rows 224..255) is dead
  f32x16 acc[2][2];
  const unsigned char* fr = L.frag[SRC];
  auto epi_quad = [&](const f32x16& ac, int t, int i, int q) {
    if (N != 256 && (2 * wave + i) * 32 + 8 * q + 4 * hi >= N) return;
    const float4 bb = *reinterpret_cast<const float4*>(L.bias + (BIASOFF + i * 32 + 8 * q) * 4);
    float o[4];
    if (ACT) {
      const v2f_t o01 = softplus100_pk(v2f_t{ac[4 * q], ac[4 * q + 1]}, v2f_t{bb.x, bb.y});
      const v2f_t o23 = softplus100_pk(v2f_t{ac[4 * q + 2], ac[4 * q + 3]}, v2f_t{bb.z, bb.w});
      o[0] = o01.x; o[1] = o01.y; o[2] = o23.x; o[3] = o23.y;
    } else {
      o[0] = ac[4 * q] + bb.x; o[1] = ac[4 * q + 1] + bb.y; o[2] = ac[4 * q + 2] + bb.z; o[3] = ac[4 * q + 3] + bb.w;
    }
    unsigned char* lq = L.quad[DST] + ((i * 4 + q) * BP + t * 32) * 16;
    if (N == 256 || (2 * wave + i) * 32 + 8 * q + 4 * hi + 3 < N) {
      const uint2 pk = make_uint2(pack2(o[0], o[1]), pack2(o[2], o[3]));
      if (ACT) *reinterpret_cast<uint2*>(lq) = pk;
      if (SAVE)                                        // wave-uniform row base + per-lane 32-bit offset + immediate
        *reinterpret_cast<uint2*>(reinterpret_cast<char*>(hout) + ((size_t)((i * 4 + q) * L.ldp16) + t * 512) + (size_t)L.gquad) = pk;
    } else {                                           // lin3: the quad holding row 216 (rows 217.. receive the PE copy)
      const int n0 = (2 * wave + i) * 32 + 8 * q + 4 * hi;
#pragma unroll
      for (int e = 0; e < 4; ++e) if (n0 + e < N) reinterpret_cast<u16*>(lq)[e] = f2bf(o[e]);
    }
  };
#pragma unroll
  for (int t = 0; t <= NT; ++t) {
    const bool mma_on = t < NT && (FULL || t < nt), epi_on = t >= 1 && (FULL || t - 1 < nt);
    f32x16(&am)[2] = acc[t & 1];
    const f32x16(&ae)[2] = acc[(t + 1) & 1];
    if (mma_on) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { am[0][r] = 0.0f; am[1][r] = 0.0f; }
    }
    uint4 bv = make_uint4(0u, 0u, 0u, 0u);
    if (mma_on) bv = *reinterpret_cast<const uint4*>(fr + t * 512);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (mma_on) {
        const uint4 cur = bv;
        if (ks + 1 < KS) bv = *reinterpret_cast<const uint4*>(fr + ((ks + 1) * 2 * BP * 16 + t * 512));
        am[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&wc[0][ks]), *reinterpret_cast<const bf16x8*>(&cur), am[0], 0, 0, 0);
        if (live1) am[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&wc[1][ks]), *reinterpret_cast<const bf16x8*>(&cur), am[1], 0, 0, 0);
      }
      if (epi_on) {
#pragma unroll
        for (int g = 0; g < 8; ++g)
          if (g >= (8 * ks) / KS && g < (8 * (ks + 1)) / KS) epi_quad(ae[g >> 2], t - 1, g >> 2, g & 3);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __syncthreads();
}

template <int NT, bool VALUES>
__global__ __launch_bounds__(F6T, 1) void sdf_fused_w64_kernel(FusedArgs a, int ntiles, int nwg) {
  typedef F6Cfg<NT> C;
  constexpr int BP = C::BP;
  extern __shared__ __attribute__((aligned(16))) unsigned char f6lds[];
  unsigned char* PE = f6lds + C::PE;
  float* biasl = reinterpret_cast<float*>(f6lds + C::BIAS);     // [l][256]; lin8 in packed row order
  u16* pe16 = reinterpret_cast<u16*>(PE);
  const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool save = !VALUES && a.save;

  for (int idx = tid; idx < 8 * 256; idx += F6T) {
    const int l = idx >> 8, n = idx & 255;
    float v = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) if (l == k && n < (k == 3 ? 217 : 256)) v = a.bias[k][n];
    biasl[idx] = v * SOFTPLUS_C;             // hidden layers: pre-scaled for softplus100_pk
  }
  for (int n = tid; n < 257; n += F6T) {
    int bi = n + a.bias8_rot; if (bi >= a.bias8_n) bi -= a.bias8_n;
    biasl[8 * 256 + n] = VALUES ? (n == 0 ? a.bias[8][0] : 0.0f) : a.bias[8][bi];
  }

  // two register sets for the weight slices (this layer / next layer); lin0's short slice (K = 64: 4 k-steps) and this wave's four
  // k-steps of the sdf row of lin8 are re-fetched per batch into whichever set is idle
  uint4 wA[2][16], wB[2][16];
  auto load_w = [&](uint4 (&dst)[2][16], const uint4* Wl, int KS, int N) {
    // kernarg pointer (SGPR pair) + ONE 32-bit per-lane offset (made opaque: otherwise the fragment addresses of all layers are
    // loop-invariant 64-bit values that get hoisted out of the batch loop and spilled) + immediate.  Dead row tiles (lin3:
    // rows >= 224) re-read a live one.
    const char* base = reinterpret_cast<const char*>(Wl);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int tile = (2 * wave + i) * 32 < N ? 2 * wave + i : 2 * wave;
      unsigned voff = (unsigned)((tile * KS) * 64 + lane) * 16u;
      asm volatile("" : "+v"(voff));
#pragma unroll
      for (int ks = 0; ks < 16; ++ks)
        if (ks < KS) dst[i][ks] = *reinterpret_cast<const uint4*>(base + voff + ks * 1024);
    }
  };

  F6Lane L;
  {
    const unsigned fo = (unsigned)(hi * BP + (lane & 31)) * 16u, qo = (unsigned)((8 * wave) * BP + (lane & 31)) * 16u + 8u * hi;
    unsigned b0 = fo, b1 = fo + C::XB, b2 = fo + C::PE, q0 = qo, q1 = qo + C::XB, bb = C::BIAS + (unsigned)(64 * wave + 4 * hi) * 4u;
    asm volatile("" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(q0), "+v"(q1), "+v"(bb));
    L.frag[0] = f6lds + b0; L.frag[1] = f6lds + b1; L.frag[2] = f6lds + b2;
    L.quad[0] = f6lds + q0; L.quad[1] = f6lds + q1; L.bias = f6lds + bb;
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
    L.gquad = ((unsigned)(8 * wave) * (unsigned)a.ldp + (unsigned)(p0 + (lane & 31))) * 16u + 8u * hi;
    // ---- positional encoding (embedder.py:12-36) into the PE octets (rows 39..63 zero)
    for (int idx = tid; idx < 64 * BP; idx += F6T) {
      const int j = idx / BP, p = idx % BP;
      float v = 0.0f;
      if (j < 39 && p < nt * 32) {
        const int c = j < 3 ? j : (j - 3) % 3;
        const float xc = a.x_fm[(size_t)c * a.ldp + p0 + p];
        if (j < 3) v = xc;
        else {
          const int k = (j - 3) / 6, is_cos = ((j - 3) % 6) >= 3;
          const float f = (float)(1 << k);
          v = is_cos ? __cosf(xc * f) : __sinf(xc * f);
        }
        if (save) a.E[(size_t)j * a.ldp + p0 + p] = v;
      }
      pe16[((j >> 3) * BP + p) * 8 + (j & 7)] = f2bf(v);
    }
    __syncthreads();

    float* red = reinterpret_cast<float*>(PE);               // [4 waves][BP]: partial sums of the sdf row
    auto chain = [&](auto full_tag) {
      constexpr bool FULLB = decltype(full_tag)::value;
      load_w(wB, a.Wp[0], 4, 256);
      load_w(wA, a.Wp[1], 16, 256);
      f6_layer<NT, FULLB, 4, 256, true, !VALUES, 2, 0, 0 * 256>(L, wB, nt, a.h[1], wave, hi);
      load_w(wB, a.Wp[2], 16, 256);
      f6_layer<NT, FULLB, 16, 256, true, !VALUES, 0, 1, 1 * 256>(L, wA, nt, a.h[2], wave, hi);
      load_w(wA, a.Wp[3], 16, 217);
      f6_layer<NT, FULLB, 16, 256, true, !VALUES, 1, 0, 2 * 256>(L, wB, nt, a.h[3], wave, hi);
      load_w(wB, a.Wp[4], 16, 256);
      f6_layer<NT, FULLB, 16, 217, true, false, 0, 1, 3 * 256>(L, wA, nt, nullptr, wave, hi);
      // skip connection (rend_a :87-88): rows 217..255 of lin4's input are the 39 PE rows (1/sqrt2 folded into W4)
      {
        u16* xb16 = reinterpret_cast<u16*>(f6lds + C::XB);
        for (int idx = tid; idx < 39 * BP; idx += F6T) {
          const int j = idx / BP, p = idx % BP, row = 217 + j;
          xb16[((row >> 3) * BP + p) * 8 + (row & 7)] = pe16[((j >> 3) * BP + p) * 8 + (j & 7)];
        }
        __syncthreads();
        if (save) {                                          // h4 as the unfused consumers expect it: 217 rows + PE[0..6] in the pad
          for (int idx = tid; idx < 28 * nt * 32; idx += F6T) {
            const int o8 = idx / (nt * 32), p = idx % (nt * 32);
            reinterpret_cast<uint4*>(a.h[4])[(size_t)o8 * a.ldp + p0 + p] = reinterpret_cast<const uint4*>(f6lds + C::XB)[o8 * BP + p];
          }
        }
      }
      load_w(wA, a.Wp[5], 16, 256);
      f6_layer<NT, FULLB, 16, 256, true, !VALUES, 1, 0, 4 * 256>(L, wB, nt, a.h[5], wave, hi);
      load_w(wB, a.Wp[6], 16, 256);
      f6_layer<NT, FULLB, 16, 256, true, !VALUES, 0, 1, 5 * 256>(L, wA, nt, a.h[6], wave, hi);
      load_w(wA, a.Wp[7], 16, 256);
      f6_layer<NT, FULLB, 16, 256, true, !VALUES, 1, 0, 6 * 256>(L, wB, nt, a.h[7], wave, hi);
      if (!VALUES) load_w(wB, a.Wp[8], 16, 256);             // the 256 feature rows of lin8 (tiles 0..7 of the [feature | sdf] pack)
      f6_layer<NT, FULLB, 16, 256, true, !VALUES, 0, 1, 7 * 256>(L, wA, nt, a.h[8], wave, hi);

      // ---- lin8: the sdf row, split over the waves' k-steps and reduced through LDS; then (save mode) the 256 feature rows
      {
        {                                                    // this wave's four k-steps of the sdf row -> the idle set wA
          unsigned voff = (unsigned)((((VALUES ? 0 : 8) * 16 + 4 * wave) * 64 + lane) * 16);
          asm volatile("" : "+v"(voff));
#pragma unroll
          for (int j = 0; j < 4; ++j) wA[0][j] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(a.Wp[8]) + voff + j * 1024);
        }
        const unsigned char* fr = L.frag[1] + (unsigned)(4 * wave) * (2 * BP * 16);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          if (!FULLB && t >= nt) break;
          f32x16 accs;
#pragma unroll
          for (int r = 0; r < 16; ++r) accs[r] = 0.0f;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const uint4 bv = *reinterpret_cast<const uint4*>(fr + (j * 2 * BP * 16 + t * 512));
            accs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&wA[0][j]), *reinterpret_cast<const bf16x8*>(&bv), accs, 0, 0, 0);
          }
          if (hi == 0) red[wave * BP + t * 32 + lane] = accs[0];
        }
      }
      if (!VALUES) f6_layer<NT, FULLB, 16, 256, false, true, 1, 0, 8 * 256>(L, wB, nt, a.feat, wave, hi);      // (ends with a barrier)
      else __syncthreads();
    };
    if (nt == NT) chain(std::true_type{});
    else chain(std::false_type{});
    if (tid < nt * 32) {
      float sv = biasl[8 * 256 + (VALUES ? 0 : 256)];
#pragma unroll
      for (int w = 0; w < 4; ++w) sv += red[w * BP + tid];
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

// Device kernels for the NEAT hot path on gfx950 (MI355X).  fp32 build: exact-f32 MFMA
// (v_mfma_f32_32x32x2_f32), activations kept FEATURE-MAJOR in HBM/LDS ([feature][point], points
// contiguous) so that (i) the point tile is the MFMA B operand with conflict-free ds_read_b32,
// (ii) accumulator rows go straight back to memory as 128-byte row segments, no transposes.
//
//   out[n][p] = epi( sum_k Wm[n][k] * in[k][p] + bias[n] )       one workgroup = 64 points x all n
//
// Wm is W (forward / tangent chains) or W^T (adjoint / reverse chains); both are pre-packed in MFMA
// A-fragment order by pack_kernel so that a wave's weight fetch is one coalesced 256-byte load.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "bf16_common.hpp"

namespace neat {

constexpr int BM = 64;          // points per workgroup tile
constexpr int WG = 256;         // threads per workgroup (4 waves, one per SIMD)

// ---------------------------------------------------------------------------------------------
// activation helpers (reference: nn.Softplus(beta=100), threshold 20 -- neat_wfr_rend_a.py:76)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float softplus100(float a) {
  const float t = 100.0f * a;
  return t > 20.0f ? a : log1pf(expf(t)) * 0.01f;
}
// softplus'(a) = sigmoid(100 a) = 1 - exp(-100 h), h = softplus(a)
__device__ __forceinline__ float dphi_from_h(float h) { return -expm1f(-100.0f * h); }

enum Epi : int {
  EPI_LINEAR = 0,    // out = acc (+bias) ; rows >= n_split go to out1 ; optional accumulate into out0
  EPI_SOFTPLUS = 1,  // out0 = softplus100(acc + bias)
  EPI_RELU = 2,      // out0 = max(acc + bias, 0)
  EPI_SIGMOID = 3,   // out0 = sigmoid(acc + bias)
  EPI_REV = 4,       // adjoint chain: rows < n_split: out0 = acc * phi'(aux0) ; rows >= n_split: out1 = acc
  EPI_TAN = 5,       // tangent chain: s = phi'(aux0); out0 = acc*s ; out1 = acc * aux1 * 100 (1-s)
  EPI_BWD = 6,       // reverse chain: out0 = acc * phi'(aux0) + aux1
  EPI_BWD_RELU = 7,  // out0 = aux0 > 0 ? acc : 0
};

struct LayerArgs {
  const float* in0; const float* in1;   // input row segments, feature-major [rows][ldp]
  int rows0, rows1;                     // K = rows0 + rows1
  int Kpad;                             // K rounded up to a multiple of 8 (zero rows / zero weights)
  const float* Wp;                      // packed weights [NT][Kpad/2][64]
  const float* bias;                    // [N] or null
  int N, NT;                            // valid output rows, number of 32-row tiles
  int ldp;                              // point stride of every array (multiple of 64)
  float* out0; float* out1; int n_split; int accumulate;
  const float* aux0; const float* aux1;
  int x3;                               // 1: the contraction runs as three bf16 MFMAs on hi/lo splits of both operands (NEAT_BF16X3)
};

template <int NTW>
__device__ __forceinline__ void mma_rows(f32x16 (&acc)[3][2], const float* __restrict__ wp0, int tile_stride,
                                         const float* __restrict__ bl, int s_begin, int s_end) {
  // wp0: this wave's first tile, already offset by lane; consecutive owned tiles are tile_stride floats apart.
  for (int s = s_begin; s < s_end; s += 4) {
    float av[NTW][4], bv[2][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int i = 0; i < NTW; ++i) av[i][u] = wp0[(size_t)i * tile_stride + (s + u) * 64];
      bv[0][u] = bl[(2 * (s + u)) * BM];
      bv[1][u] = bl[(2 * (s + u)) * BM + 32];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int i = 0; i < NTW; ++i) {
        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][u], bv[0][u], acc[i][0], 0, 0, 0);
        acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][u], bv[1][u], acc[i][1], 0, 0, 0);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// NEAT_BF16X3: fp32 storage, fp32 accumulation, every product x*y evaluated as  hi(x) hi(y) + hi(x) lo(y) + lo(x) hi(y)  with
// hi = bf16(x), lo = bf16(x - hi): three v_mfma_f32_32x32x16_bf16 (16 k per 32 cycles each) instead of eight
// v_mfma_f32_32x32x2_f32 (2 k per 64 cycles).  The dropped lo*lo term and the rounding of lo are ~2^-17 relative per product;
// everything else of the fp32 build (layouts, epilogues, every other kernel) is shared.  Operands are split in registers on
// the fly: the fp32 weight pack and the fp32 LDS tile are read in the k-order the bf16 fragment wants (8 consecutive k per lane).
// ---------------------------------------------------------------------------------------------
struct X3Frag { bf16x8 hi, lo; };
__device__ __forceinline__ X3Frag x3_split(const float (&f)[8]) {
  unsigned h[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    h[j] = pack2(f[2 * j], f[2 * j + 1]);
    l[j] = pack2(f[2 * j] - bf_lo(h[j]), f[2 * j + 1] - bf_hi(h[j]));
  }
  X3Frag r;
  const uint4 hv = make_uint4(h[0], h[1], h[2], h[3]), lv = make_uint4(l[0], l[1], l[2], l[3]);
  r.hi = *reinterpret_cast<const bf16x8*>(&hv);
  r.lo = *reinterpret_cast<const bf16x8*>(&lv);
  return r;
}
__device__ __forceinline__ void x3_mfma(f32x16& acc, const X3Frag& A, const X3Frag& B) {
  acc = NEAT_MFMA16(A.lo, B.hi, acc, 0, 0, 0);      // small terms first
  acc = NEAT_MFMA16(A.hi, B.lo, acc, 0, 0, 0);
  acc = NEAT_MFMA16(A.hi, B.hi, acc, 0, 0, 0);
}

// wbase: this wave's first tile of the fp32 pack [tile][Kpad/2][64] (element (s, lane) = W[lane & 31][2 s + (lane >> 5)]);
// tile: the fp32 LDS input tile [Kpad][BM].  16-k chunks go through the split product, a remaining 8-k chunk through the f32 MFMA.
template <int NTW>
__device__ __forceinline__ void mma_rows_x3(f32x16 (&acc)[3][2], const float* __restrict__ wbase, int tile_stride,
                                            const float* __restrict__ tile, int KS) {
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const int SF = KS >> 3;
  // the weight floats of chunk S+1 are requested before chunk S is multiplied: one L2 round trip per chunk was most of the kernel
  const float* wl = wbase + (4 * h) * 64 + r;
  float fa[NTW][8], fn[NTW][8];
  auto fetch = [&](float (&dst)[NTW][8], int S) {
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) dst[i][e] = wl[(size_t)i * tile_stride + (8 * S + (e >> 1)) * 64 + 32 * (e & 1)];
  };
  if (SF > 0) fetch(fn, 0);
  for (int S = 0; S < SF; ++S) {
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) fa[i][e] = fn[i][e];
    if (S + 1 < SF) fetch(fn, S + 1);
    X3Frag B[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = tile[(16 * S + 8 * h + e) * BM + r + 32 * q];
      B[q] = x3_split(f);
    }
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
      const X3Frag A = x3_split(fa[i]);
      x3_mfma(acc[i][0], A, B[0]);
      x3_mfma(acc[i][1], A, B[1]);
    }
  }
  if (8 * SF < KS) mma_rows<NTW>(acc, wbase + lane, tile_stride, tile + h * BM + r, 8 * SF, KS);
}

template <int EPI>
__device__ __forceinline__ void epilogue_tile(const LayerArgs& a, const f32x16& acc, int nt, int pt, int lane, int p0) {
  const int p = p0 + pt * 32 + (lane & 31);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int n = nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (n >= a.N) continue;
    float v = acc[r];
    const size_t idx = (size_t)n * a.ldp + p;
    if (EPI == EPI_LINEAR) {
      if (a.bias) v += a.bias[n];
      if (n < a.n_split) {
        if (a.accumulate) v += a.out0[idx];
        a.out0[idx] = v;
      } else {
        a.out1[(size_t)(n - a.n_split) * a.ldp + p] = v;
      }
    } else if (EPI == EPI_SOFTPLUS) {
      a.out0[idx] = softplus100(v + a.bias[n]);
    } else if (EPI == EPI_RELU) {
      a.out0[idx] = fmaxf(v + a.bias[n], 0.0f);
    } else if (EPI == EPI_SIGMOID) {
      a.out0[idx] = 1.0f / (1.0f + expf(-(v + a.bias[n])));
    } else if (EPI == EPI_REV) {
      if (n < a.n_split) a.out0[idx] = v * dphi_from_h(a.aux0[idx]);
      else a.out1[(size_t)(n - a.n_split) * a.ldp + p] = v;
    } else if (EPI == EPI_TAN) {
      const float s = dphi_from_h(a.aux0[idx]);
      const float u = a.aux1[idx];
      a.out0[idx] = v * s;
      a.out1[idx] = v * u * (100.0f * (1.0f - s));
    } else if (EPI == EPI_BWD) {
      a.out0[idx] = v * dphi_from_h(a.aux0[idx]) + a.aux1[idx];
    } else if (EPI == EPI_BWD_RELU) {
      a.out0[idx] = a.aux0[idx] > 0.0f ? v : 0.0f;
    }
  }
}

template <int EPI>
__global__ __launch_bounds__(WG, 2) void layer_kernel(LayerArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [Kpad][BM]; reused for split-K reduce
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p0 = blockIdx.x * BM;
  const int K = a.rows0 + a.rows1;
  {  // stage the 64-point input tile: 16 rows x 64 points per pass, float4 per lane
    const int c4 = (tid & 15) * 4;
    for (int r = tid >> 4; r < a.Kpad; r += 16) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < a.rows0) v = *reinterpret_cast<const float4*>(a.in0 + (size_t)r * a.ldp + p0 + c4);
      else if (r < K) v = *reinterpret_cast<const float4*>(a.in1 + (size_t)(r - a.rows0) * a.ldp + p0 + c4);
      *reinterpret_cast<float4*>(lds + r * BM + c4) = v;
    }
  }
  __syncthreads();
  const int KS = a.Kpad >> 1;
  const float* bl = lds + (lane >> 5) * BM + (lane & 31);
  f32x16 acc[3][2];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][q][r] = 0.0f;

  if (a.NT > 2) {
    // wave w owns tiles w, w+4, w+8
    const int ntw = (a.NT - wave + 3) >> 2;          // tiles owned (wave-uniform, 1..3 here)
    const float* wp0 = a.Wp + (size_t)wave * KS * 64 + lane;
    const int tstride = 4 * KS * 64;
    if (a.x3) {
      const float* wb = a.Wp + (size_t)wave * KS * 64;
      if (ntw >= 3) mma_rows_x3<3>(acc, wb, tstride, lds, KS);
      else if (ntw == 2) mma_rows_x3<2>(acc, wb, tstride, lds, KS);
      else if (ntw == 1) mma_rows_x3<1>(acc, wb, tstride, lds, KS);
    } else if (ntw >= 3) mma_rows<3>(acc, wp0, tstride, bl, 0, KS);
    else if (ntw == 2) mma_rows<2>(acc, wp0, tstride, bl, 0, KS);
    else if (ntw == 1) mma_rows<1>(acc, wp0, tstride, bl, 0, KS);
#pragma unroll
    for (int i = 0; i < 3; ++i) {            // static indices: a runtime-indexed accumulator array would live in scratch
      if (i < ntw) {
        epilogue_tile<EPI>(a, acc[i][0], wave + 4 * i, 0, lane, p0);
        epilogue_tile<EPI>(a, acc[i][1], wave + 4 * i, 1, lane, p0);
      }
    }
  } else {
    // narrow outputs (N <= 64): split K across waves, then reduce through LDS
    const int ksplit = 4 / a.NT;                     // 4 (NT=1) or 2 (NT=2)
    const int tile = wave % a.NT, kpart = wave / a.NT;
    const int per = ((KS / 4 + ksplit - 1) / ksplit) * 4;
    const int sb = kpart * per, se = min(KS, sb + per);
    const float* wp0 = a.Wp + (size_t)tile * KS * 64 + lane;
    if (sb < se) mma_rows<1>(acc, wp0, 0, bl, sb, se);
    __syncthreads();                                 // everyone is done reading the input tile
    float* red = lds;                                // [4 waves][2 ptiles][16][64]
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[((wave * 2 + q) * 16 + r) * 64 + lane] = acc[0][q][r];
    __syncthreads();
    if (kpart == 0) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        f32x16 sum;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = 0.0f;
          for (int kp = 0; kp < ksplit; ++kp) v += red[(((kp * a.NT + tile) * 2 + q) * 16 + r) * 64 + lane];
          sum[r] = v;
        }
        epilogue_tile<EPI>(a, sum, tile, q, lane, p0);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// weight gradient:  dW[n][k] = sum_p A[n][p] * B[k][p]   (both feature-major, reduction over points)
// up to two (A,B) pairs accumulate into the same tile (primal term + double-backward term);
// B may be the concatenation of up to 3 row segments (e.g. [h4 | PE | ones] -> the ones row yields
// the bias gradient as column K).  Split over points: grid.y chunks write partial tiles.
// ---------------------------------------------------------------------------------------------
struct WgradPair {
  const float* A; int rowsA;
  const float* B[3]; int rowsB[3];
};
struct WgradArgs {
  WgradPair pair[2]; int npairs;
  int N, Kt;                 // valid rows / cols (Kt includes the ones column if present)
  int P, ldp, chunk;         // points per grid.y slice (multiple of 32)
  float* partial;            // element (split, n, k) at n*row_stride + split*split_stride + k
  size_t row_stride, split_stride; int ktiles;
  int x3;                    // NEAT_BF16X3: split-bf16 products (see x3_mfma)
};

constexpr int WBP = 32;      // points per staging step
constexpr int WLD = WBP + 1; // padded LDS row

__global__ __launch_bounds__(WG) void wgrad_kernel(WgradArgs a) {
  __shared__ float As[128 * WLD];
  __shared__ float Bs[128 * WLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tn = blockIdx.x / a.ktiles, tk = blockIdx.x % a.ktiles;
  const int n0 = tn * 128, k0 = tk * 128;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);      // provably wave-uniform: the live[][] tests below become
  const int wr = (wave_u >> 1) * 64, wc = (wave_u & 1) * 64;     // scalar branches instead of exec-mask branches per MFMA
  // which 32x32 sub-tiles of this wave hold any valid output
  bool live[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) live[i][j] = (n0 + wr + 32 * i < a.N) && (k0 + wc + 32 * j < a.Kt);
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int pbeg = blockIdx.y * a.chunk;
  const int pend = min(a.P, pbeg + a.chunk);
  const int lrow = tid >> 3, lc4 = (tid & 7) * 4;       // 32 rows x 8 float4 per pass
  for (int q = 0; q < a.npairs; ++q) {
    const WgradPair& pr = a.pair[q];
    for (int pb = pbeg; pb < pend; pb += WBP) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int r = lrow + 32 * it;
        float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
        const int p = pb + lc4;
        const int n = n0 + r;
        if (n < pr.rowsA) va = *reinterpret_cast<const float4*>(pr.A + (size_t)n * a.ldp + p);
        int k = k0 + r;
        const float* src = nullptr;
        if (k < pr.rowsB[0]) src = pr.B[0] ? pr.B[0] + (size_t)k * a.ldp : nullptr;
        else if ((k -= pr.rowsB[0]) < pr.rowsB[1]) src = pr.B[1] ? pr.B[1] + (size_t)k * a.ldp : nullptr;
        else if ((k -= pr.rowsB[1]) < pr.rowsB[2]) src = pr.B[2] ? pr.B[2] + (size_t)k * a.ldp : nullptr;
        if (src) vb = *reinterpret_cast<const float4*>(src + p);
        if (p + 3 >= pend) {      // ragged tail of the point range: mask columns >= pend
          if (p + 0 >= pend) { va.x = 0.f; vb.x = 0.f; }
          if (p + 1 >= pend) { va.y = 0.f; vb.y = 0.f; }
          if (p + 2 >= pend) { va.z = 0.f; vb.z = 0.f; }
          if (p + 3 >= pend) { va.w = 0.f; vb.w = 0.f; }
        }
        float* da = As + r * WLD + lc4; float* db = Bs + r * WLD + lc4;
        da[0] = va.x; da[1] = va.y; da[2] = va.z; da[3] = va.w;
        db[0] = vb.x; db[1] = vb.y; db[2] = vb.z; db[3] = vb.w;
      }
      __syncthreads();
      if (a.x3) {
#pragma unroll
        for (int S = 0; S < WBP / 16; ++S) {
          X3Frag A[2], B[2];
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            float fa[8], fb[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              fa[e] = As[(wr + 32 * i + (lane & 31)) * WLD + 16 * S + 8 * (lane >> 5) + e];
              fb[e] = Bs[(wc + 32 * i + (lane & 31)) * WLD + 16 * S + 8 * (lane >> 5) + e];
            }
            A[i] = x3_split(fa); B[i] = x3_split(fb);
          }
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              if (live[i][j]) x3_mfma(acc[i][j], A[i], B[j]);
        }
        __syncthreads();
        continue;
      }
      const float* ap = As + (wr + (lane & 31)) * WLD + (lane >> 5);
      const float* bp = Bs + (wc + (lane & 31)) * WLD + (lane >> 5);
#pragma unroll
      for (int s = 0; s < WBP / 2; ++s) {
        const float a0 = ap[2 * s], a1 = ap[32 * WLD + 2 * s];
        const float b0 = bp[2 * s], b1 = bp[32 * WLD + 2 * s];
        if (live[0][0]) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        if (live[0][1]) acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        if (live[1][0]) acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        if (live[1][1]) acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      }
      __syncthreads();
    }
  }
  float* dst = a.partial + (size_t)blockIdx.y * a.split_stride;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (!live[i][j]) continue;
      const int k = k0 + wc + 32 * j + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wr + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (n < a.N && k < a.Kt) dst[(size_t)n * a.row_stride + k] = acc[i][j][r];
      }
    }
}

// stage 1 of the split reduction: sum groups of consecutive splits, fully parallel and bandwidth-bound
//   out[n][g][k] = sum_{s in group g} partial[n][s][k]        (both with row stride = (#splits) * Kld)
__global__ void wpartial_group_sum_kernel(const float* __restrict__ partial, int splits, int Kld4, int groups, int per,
                                          int rows, float* __restrict__ out) {
  const int k4 = blockIdx.x * blockDim.x + threadIdx.x;     // float4 column
  const int g = blockIdx.y, n = blockIdx.z;
  if (k4 >= Kld4 || n >= rows) return;
  const float4* src = reinterpret_cast<const float4*>(partial) + ((size_t)n * splits + (size_t)g * per) * Kld4 + k4;
  const int cnt = min(per, splits - g * per);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
  for (int s = 0; s < cnt; ++s) {
    const float4 v = src[(size_t)s * Kld4];
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  reinterpret_cast<float4*>(out)[((size_t)n * groups + g) * Kld4 + k4] = acc;
}

// Reduce the split partials for one output row, undo the input-column permutation / fold scale, and
// apply the weight-norm backward:  W = g v/|v|  =>  dg = <dW, v>/|v| ;  dv = g/|v| (dW - <dW,v> v/|v|^2).
struct WreduceArgs {
  const float* partial; int splits; size_t row_stride, split_stride;   // (split, n, k) at n*row_stride + split*split_stride + k
  int O, I;                  // layer dims (torch layout [O][I])
  int s0, s0p, off0, off1;   // packed input order: [source cols off0..off0+s0) | pad to s0p | source cols off1.. )
  int rot;                   // packed output row n holds source row (n + rot) mod O
  float scale;               // folded input scale (1/sqrt2 for the skip layer)
  const float* v; const float* g;      // weight_v [O][I], weight_g [O]
  float* dv; float* dg; float* db;     // outputs (db may be null -> no bias column)
  int bias_col;              // column of the partial holding the bias gradient (= packed K), or -1
};

// NW waves per workgroup: 4 for a handful of partials (after a group-sum stage, fp32 build), 16 to sum the ~245 split
// partials of the bf16 build directly (one pass over the partial buffer instead of group-sum + finish: the row is
// 245 x ~1 KiB, 16 waves keep 16 x 4 independent loads in flight)
template <int NW>
__device__ __forceinline__ void wreduce_wnorm_body(const WreduceArgs& a, const int o) {
  // one workgroup per output row: wave w sums splits w, w+NW, ... (independent loads, unrolled), LDS combine,
  // then wave 0 applies the weight-norm backward.
  constexpr int MAXC = 5;                 // up to 320 input columns (+ bias column handled by lane 0 of each wave)
  __shared__ float red[NW][MAXC * 64 + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc[MAXC], bacc = 0.0f;
  int jcol[MAXC];
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int i = lane + 64 * c;          // source column
    acc[c] = 0.0f;
    jcol[c] = (i < a.I) ? ((i >= a.off0 && i < a.off0 + a.s0) ? i - a.off0 : a.s0p + (i - a.off1)) : -1;   // packed column
  }
  const int on = (o - a.rot + a.O) % a.O;                    // packed row of source row o
  const size_t row_off = (size_t)on * a.row_stride, split_stride = a.split_stride;
#pragma unroll 4
  for (int sp = wave; sp < a.splits; sp += NW) {
    const float* src = a.partial + sp * split_stride + row_off;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) if (jcol[c] >= 0) acc[c] += src[jcol[c]];
    if (lane == 0 && a.bias_col >= 0) bacc += src[a.bias_col];
  }
#pragma unroll
  for (int c = 0; c < MAXC; ++c) red[wave][c * 64 + lane] = acc[c];
  if (lane == 0) red[wave][MAXC * 64] = bacc;
  __syncthreads();
  if (wave != 0) return;
  float dw[MAXC], vv[MAXC];
  float dot = 0.0f, nrm2 = 0.0f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int i = lane + 64 * c;
    dw[c] = 0.0f; vv[c] = 0.0f;
    if (i < a.I) {
      const int k = c * 64 + lane;
      float t = 0.0f;
#pragma unroll
      for (int w = 0; w < NW; ++w) t += red[w][k];
      dw[c] = t * a.scale;
      vv[c] = a.v[(size_t)o * a.I + i];
      dot += dw[c] * vv[c];
      nrm2 += vv[c] * vv[c];
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { dot += __shfl_xor(dot, off); nrm2 += __shfl_xor(nrm2, off); }
  const float inv = 1.0f / sqrtf(nrm2);
  const float go = a.g[o];
  const float coef = dot * inv * inv;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int i = lane + 64 * c;
    if (i < a.I) a.dv[(size_t)o * a.I + i] = go * inv * (dw[c] - coef * vv[c]);
  }
  if (lane == 0) {
    a.dg[o] = dot * inv;
    if (a.db && a.bias_col >= 0) {
      float t = 0.0f;
#pragma unroll
      for (int w = 0; w < NW; ++w) t += red[w][MAXC * 64];
      a.db[o] = t;
    }
  }
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void wreduce_wnorm_kernel(WreduceArgs a) { wreduce_wnorm_body<NW>(a, blockIdx.x); }

// One-launch reduction for many split partials (bf16 build: 82..245 splits of 1 KiB rows): the workgroup of output row o
// sums ITS packed row over all splits with 16-byte loads -- thread = (float4 column, one of up to 4 split sub-sequences),
// eight independent loads in flight per thread -- combines the sub-sums in LDS in a fixed order (deterministic), and wave 0
// applies the weight-norm backward.  Replaces wpartial_group_sum_kernel + wreduce_wnorm_kernel<4> (one launch and a
// stage buffer less per layer).
constexpr int WRD_MAXK4 = 80;            // partial row length in float4 (K + 1 <= 320)
__device__ __forceinline__ void wreduce_direct_body(const WreduceArgs& a, const int o) {
  constexpr int MAXC = 5;
  __shared__ __attribute__((aligned(16))) float part[4][WRD_MAXK4 * 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Kld4 = (int)(a.split_stride >> 2);
  const int nsub = min(4, (int)blockDim.x / Kld4);
  const int sub = tid / Kld4, c4 = tid - sub * Kld4;
  const int on = (o - a.rot + a.O) % a.O;                    // packed row of source row o
  if (sub < nsub) {
    const float4* src = reinterpret_cast<const float4*>(a.partial + (size_t)on * a.row_stride) + c4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
    for (int sp = sub; sp < a.splits; sp += nsub) {
      const float4 v = src[(size_t)sp * Kld4];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *reinterpret_cast<float4*>(&part[sub][4 * c4]) = acc;
  }
  __syncthreads();
  if (wave != 0) return;
  float dw[MAXC], vv[MAXC];
  float dot = 0.0f, nrm2 = 0.0f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int i = lane + 64 * c;          // source column
    dw[c] = 0.0f; vv[c] = 0.0f;
    if (i < a.I) {
      const int j = (i >= a.off0 && i < a.off0 + a.s0) ? i - a.off0 : a.s0p + (i - a.off1);   // packed column
      float t = 0.0f;
      for (int w = 0; w < nsub; ++w) t += part[w][j];
      dw[c] = t * a.scale;
      vv[c] = a.v[(size_t)o * a.I + i];
      dot += dw[c] * vv[c];
      nrm2 += vv[c] * vv[c];
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { dot += __shfl_xor(dot, off); nrm2 += __shfl_xor(nrm2, off); }
  const float inv = 1.0f / sqrtf(nrm2);
  const float go = a.g[o];
  const float coef = dot * inv * inv;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int i = lane + 64 * c;
    if (i < a.I) a.dv[(size_t)o * a.I + i] = go * inv * (dw[c] - coef * vv[c]);
  }
  if (lane == 0) {
    a.dg[o] = dot * inv;
    if (a.db && a.bias_col >= 0) {
      float t = 0.0f;
      for (int w = 0; w < nsub; ++w) t += part[w][a.bias_col];
      a.db[o] = t;
    }
  }
}
__global__ __launch_bounds__(256) void wreduce_direct_kernel(WreduceArgs a) { wreduce_direct_body(a, blockIdx.x); }

// the finish of several layers in one launch (the problems of one batched weight-gradient launch): blockIdx.y = layer.
// The argument table is read from the kernel-argument segment through a pointer -- indexing the by-value array with
// blockIdx.y would copy it to scratch.
constexpr int WREDUCE_BATCH = 8;
struct WreduceBatch { WreduceArgs a[WREDUCE_BATCH]; };
__global__ __launch_bounds__(256) void wreduce_wnorm_batch_kernel(WreduceBatch) {
  const WreduceArgs* tab = (const WreduceArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  const WreduceArgs a = tab[blockIdx.y];
  if ((int)blockIdx.x >= a.O) return;          // lin3 has 217 rows
  wreduce_wnorm_body<4>(a, blockIdx.x);
}
__global__ __launch_bounds__(256) void wreduce_direct_batch_kernel(WreduceBatch) {
  const WreduceArgs* tab = (const WreduceArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  const WreduceArgs a = tab[blockIdx.y];
  if ((int)blockIdx.x >= a.O) return;
  wreduce_direct_body(a, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------
// weight preparation: row scale g/|v| (weight norm), then gather into MFMA A-fragment order
// ---------------------------------------------------------------------------------------------
constexpr int NLAYERS = 19;     // 0..8 SDF, 9..13 render head, 14..18 attraction head
struct NetPtrs {
  const float* v[NLAYERS]; const float* g[NLAYERS]; const float* b[NLAYERS];
  int O[NLAYERS], I[NLAYERS];
};
struct RowScaleArgs { NetPtrs net; float* rowscale; int row_off[NLAYERS + 1]; };

__global__ __launch_bounds__(WG) void rowscale_kernel(RowScaleArgs a) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.row_off[NLAYERS]) return;
  int l = 0;
  while (row >= a.row_off[l + 1]) ++l;
  const int o = row - a.row_off[l], I = a.net.I[l];
  if (!a.net.v[l]) return;                      // layer not supplied (e.g. SDF network used without the heads)
  const float* v = a.net.v[l] + (size_t)o * I;
  float s = 0.0f;
  for (int i = lane; i < I; i += 64) s += v[i] * v[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if (lane == 0) a.rowscale[row] = a.net.g[l][o] / sqrtf(s);
}

// ---------------------------------------------------------------------------------------------
// small per-point kernels (feature-major outputs)
// ---------------------------------------------------------------------------------------------
// x = o + z d for p = r*S + i ; also writes row-major points if requested
// A value the compiler may not fuse into an fma with its consumer.  (HIP's __fmul_rn / __fadd_rn are plain operators under the default
// -ffp-contract=fast and DO get contracted: the round-5 form of the two kernels below compiled to v_fmac_f32 / v_pk_fma_f32.)
__device__ __forceinline__ float rounded(float x) { asm volatile("" : "+v"(x)); return x; }

__global__ void points_from_rays_kernel(const float* __restrict__ o, const float* __restrict__ d,
                                        const float* __restrict__ z, int R, int S, int ldp,
                                        float* __restrict__ x_fm, float* __restrict__ pts_rm,
                                        const float* __restrict__ extra, int E) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= ldp) return;
  float xv[3] = {0.f, 0.f, 0.f};
  if (p >= R * S && p < R * S + E) {
    const int e = p - R * S;
    xv[0] = extra[e * 3]; xv[1] = extra[e * 3 + 1]; xv[2] = extra[e * 3 + 2];
  }
  if (p < R * S) {
    const int r = p / S;
    const float zz = z[p];
#pragma unroll
    for (int c = 0; c < 3; ++c) xv[c] = o[r * 3 + c] + rounded(zz * d[r * 3 + c]);      // torch's `cam_loc + z * dirs` (rend_a :395-396): product rounded, then the sum -- no fma
    if (pts_rm) { pts_rm[p * 3 + 0] = xv[0]; pts_rm[p * 3 + 1] = xv[1]; pts_rm[p * 3 + 2] = xv[2]; }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) x_fm[(size_t)c * ldp + p] = xv[c];
}

__global__ void rm_to_fm_kernel(const float* __restrict__ src, int P, int C, int ldp, float* __restrict__ dst) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= ldp) return;
  for (int c = 0; c < C; ++c) dst[(size_t)c * ldp + p] = (p < P) ? src[(size_t)p * C + c] : 0.0f;
}
__global__ void fm_to_rm_kernel(const float* __restrict__ src, int P, int C, int ldp, float* __restrict__ dst, int accumulate) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  for (int c = 0; c < C; ++c) {
    const float v = src[(size_t)c * ldp + p];
    if (accumulate) dst[(size_t)p * C + c] += v; else dst[(size_t)p * C + c] = v;
  }
}
__global__ void fill_kernel(float* __restrict__ dst, size_t n, float v) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = v;
}
__global__ void ones_kernel(float* __restrict__ dst, int P, int ldp) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < ldp) dst[p] = p < P ? 1.0f : 0.0f;
}

// PE-6 rows (embedder.py:12-36): [x, sin(2^k x), cos(2^k x)]_k  -> E[39][ldp]
__global__ void posenc6_kernel(const float* __restrict__ x_fm, int ldp, float* __restrict__ E) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= ldp) return;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float xc = x_fm[(size_t)c * ldp + p];
    E[(size_t)c * ldp + p] = xc;
    float f = 1.0f;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      E[(size_t)(3 + 6 * k + c) * ldp + p] = sinf(xc * f);
      E[(size_t)(6 + 6 * k + c) * ldp + p] = cosf(xc * f);
      f *= 2.0f;
    }
  }
}

// tangent seed  E^ = J g^ : derivative of PE-6 along g^ (rows as posenc6_kernel)
template <bool FAST>
__global__ void posenc6_tangent_kernel(const float* __restrict__ x_fm, const float* __restrict__ gh_fm, int ldp,
                                       float* __restrict__ Eh) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= ldp) return;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float xc = x_fm[(size_t)c * ldp + p], gc = gh_fm[(size_t)c * ldp + p];
    Eh[(size_t)c * ldp + p] = gc;
    float f = 1.0f;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      Eh[(size_t)(3 + 6 * k + c) * ldp + p] = f * (FAST ? __cosf(xc * f) : cosf(xc * f)) * gc;
      Eh[(size_t)(6 + 6 * k + c) * ldp + p] = -f * (FAST ? __sinf(xc * f) : sinf(xc * f)) * gc;
      f *= 2.0f;
    }
  }
}

// adjoint seed: u7 = W8[0,:] * phi'(h8)
__global__ void adjoint_seed_kernel(const float* __restrict__ v8, const float* __restrict__ rs8,
                                    const float* __restrict__ h8, int ldp, float* __restrict__ u7) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = blockIdx.y;
  if (p >= ldp) return;
  const float w = v8[n] * rs8[0];             // row 0 of lin8's effective weight
  const size_t idx = (size_t)n * ldp + p;
  u7[idx] = w * dphi_from_h(h8[idx]);
}

// normals + sphere clamp:  g = J^T (e0 + eskip) ; sdf = min(raw, scale (radius - |x|))  (rend_a :111-129)
// FAST (bf16 build): hardware sin/cos (|arg| <= 96, abs error ~1e-6); the fp32 build keeps libm's for the 1e-4 parity bar.
// E (optional): the PE rows [39][ldp] of the same points (posenc6_kernel: sin / cos by libm) -- read instead of recomputed
// hin (main pass; small_r == null: off): the heads' small inputs of the same point in the same launch (what head_inputs_kernel,
// below, does in a launch of its own for the stand-alone heads entry): render [p(3), PE4(view)(27), normal(3)], attraction
// [p(3), view(3), normal(3)], fp32 rows + octet-major 16-bit copies
struct HeadInArgs { const float* dirs; int P, S; float* small_r; float* small_a; u16* bf_r; u16* bf_a;
                    int skip_fp32 = 0; };     // 1: only the 16-bit copies are read (bf16 / f16 builds: every consumer takes the octets)
template <bool FAST>
__global__ void sdf_finalize_kernel(const float* __restrict__ x_fm, const float* __restrict__ out8,
                                    const float* __restrict__ e0, const float* __restrict__ es, int P, int ldp,
                                    float radius, float scale, float* __restrict__ sdf, float* __restrict__ g_fm,
                                    float* __restrict__ mask, float* __restrict__ sdf_rm, float* __restrict__ g_rm,
                                    int n_clamp, float* __restrict__ g_extra_rm, const float* __restrict__ E, HeadInArgs hin) {
  // points [0, n_clamp) get the bounding-sphere clamp (get_outputs); points [n_clamp, P) are eikonal points:
  // raw network gradient (ImplicitNetwork.gradient), written to g_extra_rm
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= ldp) return;
  float xv[3], gv[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 3; ++c) xv[c] = x_fm[(size_t)c * ldp + p];
  float s = out8[p], m = 0.0f;
  if (e0) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float acc = e0[(size_t)c * ldp + p] + es[(size_t)c * ldp + p];
      float f = 1.0f;
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const float es_ = e0[(size_t)(3 + 6 * k + c) * ldp + p] + es[(size_t)(3 + 6 * k + c) * ldp + p];
        const float ec_ = e0[(size_t)(6 + 6 * k + c) * ldp + p] + es[(size_t)(6 + 6 * k + c) * ldp + p];
        const float ang = xv[c] * f;
        if (E) acc += f * E[(size_t)(6 + 6 * k + c) * ldp + p] * es_ - f * E[(size_t)(3 + 6 * k + c) * ldp + p] * ec_;
        else acc += f * (FAST ? __cosf(ang) : cosf(ang)) * es_ - f * (FAST ? __sinf(ang) : sinf(ang)) * ec_;
        f *= 2.0f;
      }
      gv[c] = acc;
    }
  }
  if (radius > 0.0f && p < n_clamp) {
    const float nr = sqrtf(xv[0] * xv[0] + xv[1] * xv[1] + xv[2] * xv[2]);
    const float sph = scale * (radius - nr);
    if (sph < s) {
      s = sph; m = 1.0f;
      const float inv = nr > 0.0f ? -scale / nr : 0.0f;
      gv[0] = xv[0] * inv; gv[1] = xv[1] * inv; gv[2] = xv[2] * inv;
    }
  }
  sdf[p] = s;
  if (mask) mask[p] = m;
  if (g_fm) { g_fm[p] = gv[0]; g_fm[(size_t)ldp + p] = gv[1]; g_fm[(size_t)2 * ldp + p] = gv[2]; }
  if (hin.small_r) {
    const int r = (p < hin.P ? p : 0) / hin.S;
    float vr[40], va[16];
#pragma unroll
    for (int i = 0; i < 40; ++i) vr[i] = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) va[i] = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float xc = xv[c], gc = gv[c];
      const float dc = (p < hin.P) ? hin.dirs[r * 3 + c] : 0.0f;
      vr[c] = xc; vr[3 + c] = dc;
      float f = 1.0f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        vr[6 + 6 * k + c] = sinf(dc * f);
        vr[9 + 6 * k + c] = cosf(dc * f);
        f *= 2.0f;
      }
      vr[30 + c] = gc;
      va[c] = xc; va[3 + c] = dc; va[6 + c] = gc;
    }
    if (!hin.skip_fp32) {
#pragma unroll
      for (int i = 0; i < 33; ++i) hin.small_r[(size_t)i * ldp + p] = vr[i];
#pragma unroll
      for (int i = 0; i < 9; ++i) hin.small_a[(size_t)i * ldp + p] = va[i];
    }
    if (hin.bf_r) {
#pragma unroll
      for (int o = 0; o < 5; ++o)
        reinterpret_cast<uint4*>(hin.bf_r)[(size_t)o * ldp + p] = make_uint4(pack2(vr[8 * o], vr[8 * o + 1]), pack2(vr[8 * o + 2], vr[8 * o + 3]),
                                                                            pack2(vr[8 * o + 4], vr[8 * o + 5]), pack2(vr[8 * o + 6], vr[8 * o + 7]));
    }
    if (hin.bf_a) {
#pragma unroll
      for (int o = 0; o < 2; ++o)
        reinterpret_cast<uint4*>(hin.bf_a)[(size_t)o * ldp + p] = make_uint4(pack2(va[8 * o], va[8 * o + 1]), pack2(va[8 * o + 2], va[8 * o + 3]),
                                                                            pack2(va[8 * o + 4], va[8 * o + 5]), pack2(va[8 * o + 6], va[8 * o + 7]));
    }
  }
  if (p < P) {
    if (p < n_clamp) {
      if (sdf_rm) sdf_rm[p] = s;
      if (g_rm && e0) { g_rm[p * 3 + 0] = gv[0]; g_rm[p * 3 + 1] = gv[1]; g_rm[p * 3 + 2] = gv[2]; }
    } else if (g_extra_rm) {
      const int e = p - n_clamp;
      g_extra_rm[e * 3 + 0] = gv[0]; g_extra_rm[e * 3 + 1] = gv[1]; g_extra_rm[e * 3 + 2] = gv[2];
    }
  }
}

// head inputs besides the feature rows:
//   render  (rend_a :235-241): [p(3), PE4(view)(27), normal(3)] = 33 rows
//   attract (rend_a :175-181): [p(3), view(3), normal(3)]      =  9 rows
// view dir of point p is dirs[p / S] (S=1: per-point dirs)
// bf_r / bf_a (16-bit builds; null otherwise): the same rows as octet-major 16-bit copies [5 | 2 octets][ldp][8], zero padded -- what the
// heads' chains and weight gradients read (a separate oct_pack launch before)
__global__ void head_inputs_kernel(const float* __restrict__ x_fm, const float* __restrict__ g_fm,
                                   const float* __restrict__ dirs, int P, int S, int ldp,
                                   float* __restrict__ small_r, float* __restrict__ small_a,
                                   u16* __restrict__ bf_r, u16* __restrict__ bf_a) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= ldp) return;
  const int r = (p < P ? p : 0) / S;
  float vr[40], va[16];
#pragma unroll
  for (int i = 0; i < 40; ++i) vr[i] = 0.0f;
#pragma unroll
  for (int i = 0; i < 16; ++i) va[i] = 0.0f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float xc = x_fm[(size_t)c * ldp + p], gc = g_fm[(size_t)c * ldp + p];
    const float dc = (p < P) ? dirs[r * 3 + c] : 0.0f;
    vr[c] = xc; vr[3 + c] = dc;
    float f = 1.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      vr[6 + 6 * k + c] = sinf(dc * f);
      vr[9 + 6 * k + c] = cosf(dc * f);
      f *= 2.0f;
    }
    vr[30 + c] = gc;
    va[c] = xc; va[3 + c] = dc; va[6 + c] = gc;
  }
#pragma unroll
  for (int i = 0; i < 33; ++i) small_r[(size_t)i * ldp + p] = vr[i];
#pragma unroll
  for (int i = 0; i < 9; ++i) small_a[(size_t)i * ldp + p] = va[i];
  if (bf_r) {
#pragma unroll
    for (int o = 0; o < 5; ++o)
      reinterpret_cast<uint4*>(bf_r)[(size_t)o * ldp + p] = make_uint4(pack2(vr[8 * o], vr[8 * o + 1]), pack2(vr[8 * o + 2], vr[8 * o + 3]),
                                                                      pack2(vr[8 * o + 4], vr[8 * o + 5]), pack2(vr[8 * o + 6], vr[8 * o + 7]));
  }
  if (bf_a) {
#pragma unroll
    for (int o = 0; o < 2; ++o)
      reinterpret_cast<uint4*>(bf_a)[(size_t)o * ldp + p] = make_uint4(pack2(va[8 * o], va[8 * o + 1]), pack2(va[8 * o + 2], va[8 * o + 3]),
                                                                      pack2(va[8 * o + 4], va[8 * o + 5]), pack2(va[8 * o + 6], va[8 * o + 7]));
  }
}

// f16 build: the backward pass runs on cotangents normalised by a power of two so that the largest incoming one lies in [1, 2)
// (every backward kernel is linear in them).  Per-point cotangents of a train step are ~1e-7 ... 1e-3, below f16's normal range
// (6e-5); a caller may just as well pass O(1) cotangents -- a fixed scale would underflow one or overflow the other.
// slot[0] holds max |cotangent| as float bits (cot_max_kernel, atomicMax on the bit pattern: non-negative floats order like
// unsigned ints).  slot == nullptr (bf16 / fp32 builds): scale 1.
__device__ __forceinline__ float cot_scale_of(const float* slot) {
  if (!slot) return 1.0f;
  const float m = *slot;
  if (!(m > 0.0f) || !(m < 3.0e38f)) return 1.0f;
  int e;
  (void)frexpf(m, &e);                       // m = f 2^e, f in [0.5, 1)
  e = e < -100 ? -100 : (e > 100 ? 100 : e);
  return ldexpf(1.0f, 1 - e);
}
constexpr int COT_MAX_ARRAYS = 8;
struct CotMaxArgs { const float* p[COT_MAX_ARRAYS]; long long n[COT_MAX_ARRAYS]; int narr; float* slot; };
__global__ __launch_bounds__(256) void cot_max_kernel(CotMaxArgs a) {
  float m = 0.0f;
  for (int q = 0; q < a.narr; ++q) {
    const float* p = a.p[q];
    if (!p) continue;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n[q]; i += (long long)gridDim.x * blockDim.x) {
      const float v = fabsf(p[i]);
      if (v < 3.0e38f) m = fmaxf(m, v);      // (inf / nan stay out of the scale; they propagate through the pass as they are)
    }
  }
  for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  if ((threadIdx.x & 63) == 0 && m > 0.0f) atomicMax(reinterpret_cast<unsigned*>(a.slot), __float_as_uint(m));
}

// cotangent of the normals: g^ = (1-mask) (sc_r[30..32] + sc_a[6..8] + extra) ; also masks the sdf cotangent row
__global__ void normal_cotangent_kernel(const float* __restrict__ sc_r, const float* __restrict__ sc_a,
                                        const float* __restrict__ extra_rm, const float* __restrict__ mask,
                                        int P, int ldp, float* __restrict__ gh_fm,
                                        int P_main, const float* __restrict__ d_tail_rm, const float* __restrict__ cot_slot,
                                        const float* __restrict__ cot_slot_a = nullptr) {
  // the scale multiplies the caller's cotangents (extra_rm, d_tail_rm); sc_r / sc_a come out of the backward pass and carry it already
  // (sc_a in the attraction head's own scale, cot_slot_a: brought to the common one here)
  const float ext_scale = cot_scale_of(cot_slot);
  const float rho_a = cot_slot_a ? ext_scale / cot_scale_of(cot_slot_a) : 1.0f;
  // points [0, P_main): heads' normal cotangents (+ optional row-major extra), masked where the sphere clamp won;
  // points [P_main, P): appended eikonal points, cotangent d_tail_rm[p - P_main]
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= ldp) return;
  const float keep = (p < P) ? 1.0f - (mask ? mask[p] : 0.0f) : 0.0f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float v = 0.0f;
    if (p < P_main) {
      if (sc_r) v += sc_r[(size_t)(30 + c) * ldp + p];
      if (sc_a) v += sc_a[(size_t)(6 + c) * ldp + p] * rho_a;
      if (extra_rm) v += extra_rm[p * 3 + c] * ext_scale;
    } else if (p < P && d_tail_rm) {
      v = d_tail_rm[(p - P_main) * 3 + c] * ext_scale;
    }
    gh_fm[(size_t)c * ldp + p] = v * keep;
  }
}

// zero columns [p_from, ldp) of `rows` feature-major rows
__global__ void zero_tail_kernel(float* __restrict__ a, int rows, int p_from, int ldp) {
  const int p = p_from + blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= ldp) return;
  for (int r = 0; r < rows; ++r) a[(size_t)r * ldp + p] = 0.0f;
}
// ---------------------------------------------------------------------------------------------
// compositing (rend_a :540-554 volume_rendering, :406-426 integrals); one wave per ray
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_incl_scan(float v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const float t = __shfl_up(v, off);
    if (lane >= off) v += t;
  }
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
// Laplace density (density.py:21-26) and its derivative pieces
__device__ __forceinline__ float laplace_sigma(float s, float beta) {
  const float sg = (s > 0.0f) ? 1.0f : ((s < 0.0f) ? -1.0f : 0.0f);
  return (1.0f / beta) * (0.5f + 0.5f * sg * expm1f(-fabsf(s) / beta));
}

struct CompositeArgs {
  const float* z; const float* sdf; const float* dirs;   // [R,S], [P], [R,3]
  const float* x_fm; const float* rgb_fm; const float* lin_fm; const float* g_fm;   // [3|3|6|3][ldp]
  int R, S, ldp; const float* beta_ptr;
  float* weights; float* rgb; float* lines3d; float* depth; float* xyz; float* normal_map;   // outputs (row-major)
  float beta_min = 0.0f;      // the density's beta = |*beta_ptr| + beta_min (LaplaceDensity.get_beta, density.py:29-30)
};

__global__ __launch_bounds__(WG) void composite_fwd_kernel(CompositeArgs a) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= a.R) return;
  const float dn = sqrtf(a.dirs[r * 3] * a.dirs[r * 3] + a.dirs[r * 3 + 1] * a.dirs[r * 3 + 1] + a.dirs[r * 3 + 2] * a.dirs[r * 3 + 2]);
  const float beta = fabsf(*a.beta_ptr) + a.beta_min;
  float carry = 0.0f;
  float acc[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = 0.0f;
  for (int i0 = 0; i0 < a.S; i0 += 64) {
    const int i = i0 + lane;
    const bool ok = i < a.S;
    const int p = r * a.S + (ok ? i : a.S - 1);
    const float zi = a.z[p];
    const float delta = (i + 1 < a.S) ? a.z[p + 1] - zi : 1e10f;
    const float e = ok ? delta * laplace_sigma(a.sdf[p], beta) : 0.0f;
    const float incl = wave_incl_scan(e, lane);
    float excl = __shfl_up(incl, 1);             // exclusive prefix: never contains the 1e10 tail interval
    if (lane == 0) excl = 0.0f;
    const float T = expf(-(carry + excl));
    const float w = ok ? (1.0f - expf(-e)) * T : 0.0f;
    carry += __shfl(incl, 63);
    if (ok) {
      if (a.weights) a.weights[p] = w;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float xc = a.x_fm[(size_t)c * a.ldp + p];
        acc[c] += w * a.rgb_fm[(size_t)c * a.ldp + p];
        acc[3 + c] += w * (xc + a.lin_fm[(size_t)c * a.ldp + p]);            // endpoint 0 = p + offset[0:3]
        acc[6 + c] += w * (xc + a.lin_fm[(size_t)(3 + c) * a.ldp + p]);      // endpoint 1
        acc[10 + c] += w * xc;
      }
      acc[9] += w * fabsf(zi) * dn;                                           // |z d|
      if (a.normal_map) {
        const float g0 = a.g_fm[p], g1 = a.g_fm[(size_t)a.ldp + p], g2 = a.g_fm[(size_t)2 * a.ldp + p];
        const float gn = sqrtf(g0 * g0 + g1 * g1 + g2 * g2);
        acc[13] += w * g0 / gn; acc[14] += w * g1 / gn; acc[15] += w * g2 / gn;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = wave_sum(acc[k]);
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      a.rgb[r * 3 + c] = acc[c];
      a.lines3d[r * 6 + c] = acc[3 + c];
      a.lines3d[r * 6 + 3 + c] = acc[6 + c];
      a.xyz[r * 3 + c] = acc[10 + c];
      if (a.normal_map) a.normal_map[r * 3 + c] = acc[13 + c];
    }
    a.depth[r] = acc[9];
  }
}

struct CompositeBwdArgs {
  const float* z; const float* sdf; const float* dirs; const float* mask;
  const float* x_fm; const float* rgb_fm;
  int R, S, ldp; const float* beta_ptr;
  const float* d_rgb; const float* d_lines3d; const float* d_depth; const float* d_xyz;   // [R,3],[R,6],[R],[R,3] (null = 0)
  const float* d_acc = nullptr;   // [R] cotangent of the ray's opacity sum_i w_i (white_bkgd, rend_a :411-413); null = 0
  float* zrgb_fm;      // [3][ldp]  cotangent of the pre-sigmoid colour logits
  float* dlin_fm;      // [6][ldp]  cotangent of the attraction offsets
  u16* zrgb_oct = nullptr;   // 16-bit builds: the same two as one zero-padded octet per point [ldp][8] (what the heads' backward chain and
  u16* dlin_oct = nullptr;   // output-layer weight gradients read; a separate oct_pack launch before)
  float* dsdf_row;     // [ldp]     cotangent of raw sdf (0 where the sphere clamp is active)
  float* dbeta_ray;    // [R]       per-ray partial of d loss / d beta
  const float* cot_slot = nullptr;   // f16 build: the incoming cotangents are scaled on load, d beta is scaled back (cot_scale_of)
  const float* cot_slot_a = nullptr; // f16 build: the attraction head's backward chain runs in its own power-of-two scale (its cotangents,
                                     // line-loss weight 0.01 and detached weights, are orders of magnitude below the colour ones: in the
                                     // common scale they sit in f16's subnormal range); null = the common one
  float beta_min = 0.0f;             // beta = |*beta_ptr| + beta_min, as in CompositeArgs
  int tail_from = 0;                 // > 0: the workgroups behind the rays' zero the columns [tail_from, ldp) of zrgb / dlin / dsdf_row and of the
                                     // octet copies (eikonal points, padding: no head cotangents) -- what zero_tail3_kernel did in a launch of its own
};

__global__ __launch_bounds__(WG) void composite_bwd_kernel(CompositeBwdArgs a) {
  const int lane = threadIdx.x & 63;
  const int ray_blocks = (a.R + 3) / 4;
  if ((int)blockIdx.x >= ray_blocks) {
    const int p = a.tail_from + ((int)blockIdx.x - ray_blocks) * WG + (int)threadIdx.x;
    if (a.tail_from <= 0 || p >= a.ldp) return;
    for (int c = 0; c < 3; ++c) a.zrgb_fm[(size_t)c * a.ldp + p] = 0.0f;
    for (int c = 0; c < 6; ++c) a.dlin_fm[(size_t)c * a.ldp + p] = 0.0f;
    a.dsdf_row[p] = 0.0f;
    if (a.zrgb_oct) reinterpret_cast<uint4*>(a.zrgb_oct)[p] = make_uint4(0u, 0u, 0u, 0u);
    if (a.dlin_oct) reinterpret_cast<uint4*>(a.dlin_oct)[p] = make_uint4(0u, 0u, 0u, 0u);
    return;
  }
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= a.R) return;
  const float dn = sqrtf(a.dirs[r * 3] * a.dirs[r * 3] + a.dirs[r * 3 + 1] * a.dirs[r * 3 + 1] + a.dirs[r * 3 + 2] * a.dirs[r * 3 + 2]);
  const float beta = fabsf(*a.beta_ptr) + a.beta_min;
  float drgb[3] = {0.f, 0.f, 0.f}, dxyz[3] = {0.f, 0.f, 0.f}, dl[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const float cs = cot_scale_of(a.cot_slot);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    if (a.d_rgb) drgb[c] = a.d_rgb[r * 3 + c] * cs;
    if (a.d_xyz) dxyz[c] = a.d_xyz[r * 3 + c] * cs;
  }
  const float cs_a = a.cot_slot_a ? cot_scale_of(a.cot_slot_a) : cs;
#pragma unroll
  for (int c = 0; c < 6; ++c) if (a.d_lines3d) dl[c] = a.d_lines3d[r * 6 + c] * cs_a;
  const float ddepth = a.d_depth ? a.d_depth[r] * cs : 0.0f;
  const float dacc = a.d_acc ? a.d_acc[r] * cs : 0.0f;
  // pass 1 (forward over the ray): transmittance needs the exclusive prefix of E.  Park T_i and w^_i w_i in the
  // output rows (same thread reads them back in pass 2, so no hazard).
  const int nchunk = (a.S + 63) / 64;
  float carry = 0.0f;
  for (int ch = 0; ch < nchunk; ++ch) {
    const int i = ch * 64 + lane;
    const bool ok = i < a.S;
    const int p = r * a.S + (ok ? i : a.S - 1);
    const float zi = a.z[p];
    const float delta = (i + 1 < a.S) ? a.z[p + 1] - zi : 1e10f;
    const float e = ok ? delta * laplace_sigma(a.sdf[p], beta) : 0.0f;
    const float incl = wave_incl_scan(e, lane);
    float excl = __shfl_up(incl, 1);
    if (lane == 0) excl = 0.0f;
    const float T = expf(-(carry + excl));
    const float w = ok ? (1.0f - expf(-e)) * T : 0.0f;
    carry += __shfl(incl, 63);
    float wh = ddepth * fabsf(zi) * dn + dacc;
#pragma unroll
    for (int c = 0; c < 3; ++c)
      wh += drgb[c] * a.rgb_fm[(size_t)c * a.ldp + p] + dxyz[c] * a.x_fm[(size_t)c * a.ldp + p];
    if (ok) { a.dsdf_row[p] = T; a.dlin_fm[p] = wh; }
  }
  // pass 2 (backward over the ray): suffix_i = sum_{j>i} w^_j w_j by a reverse exclusive scan -- exactly the
  // reverse cumsum autograd performs; in particular it is exactly 0 for the last sample, whose 1e10 interval
  // would otherwise amplify any rounding residue.
  float carry_rev = 0.0f, dbeta = 0.0f;
  for (int ch = nchunk - 1; ch >= 0; --ch) {
    const int i = ch * 64 + lane;
    const bool ok = i < a.S;
    const int p = r * a.S + (ok ? i : a.S - 1);
    const float zi = a.z[p];
    const float delta = (i + 1 < a.S) ? a.z[p + 1] - zi : 1e10f;
    const float s = a.sdf[p];
    const float sigma = laplace_sigma(s, beta);
    const float e = ok ? delta * sigma : 0.0f;
    const float T = ok ? a.dsdf_row[p] : 0.0f;
    const float wh = ok ? a.dlin_fm[p] : 0.0f;
    const float em = expf(-e);
    const float w = ok ? (1.0f - em) * T : 0.0f;
    float rgbv[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) rgbv[c] = a.rgb_fm[(size_t)c * a.ldp + p];
    float rs = wh * w;                                   // reverse inclusive scan over lanes
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const float t = __shfl_down(rs, off);
      if (lane + off < 64) rs += t;
    }
    float excl = __shfl_down(rs, 1);
    if (lane == 63) excl = 0.0f;
    const float suffix = carry_rev + excl;               // sum_{j>i} w^_j w_j
    carry_rev += __shfl(rs, 0);
    if (ok) {
      const float dE = wh * em * T - suffix;             // d w_i/d E_i = e^-E_i T_i ; d w_j/d E_i = -w_j (j>i)
      const float dsig = dE * delta;
      const float sg = (s > 0.0f) ? 1.0f : ((s < 0.0f) ? -1.0f : 0.0f);
      const float q = expf(-fabsf(s) / beta);
      const float ib = 1.0f / beta;
      const float dsig_ds = -0.5f * sg * sg * q * ib * ib;
      const float dsig_db = -sigma * ib + 0.5f * sg * q * fabsf(s) * ib * ib * ib;
      const float keep = 1.0f - (a.mask ? a.mask[p] : 0.0f);
      a.dsdf_row[p] = dsig * dsig_ds * keep;
      dbeta += dsig * dsig_db;
      float zo[3], lo6[6];
#pragma unroll
      for (int c = 0; c < 3; ++c) { zo[c] = w * drgb[c] * rgbv[c] * (1.0f - rgbv[c]); a.zrgb_fm[(size_t)c * a.ldp + p] = zo[c]; }
#pragma unroll
      for (int c = 0; c < 6; ++c) { lo6[c] = w * dl[c]; a.dlin_fm[(size_t)c * a.ldp + p] = lo6[c]; }
      if (a.zrgb_oct) reinterpret_cast<uint4*>(a.zrgb_oct)[p] = make_uint4(pack2(zo[0], zo[1]), pack2(zo[2], 0.0f), 0u, 0u);
      if (a.dlin_oct) reinterpret_cast<uint4*>(a.dlin_oct)[p] = make_uint4(pack2(lo6[0], lo6[1]), pack2(lo6[2], lo6[3]), pack2(lo6[4], lo6[5]), 0u);
    }
  }
  dbeta = wave_sum(dbeta);
  if (lane == 0 && a.dbeta_ray) a.dbeta_ray[r] = dbeta / cs;
}

// pixel -> ray (rend_util.py:55-81,95-108)
__device__ __forceinline__ void camera_ray(const float* __restrict__ uv, const float* __restrict__ pose, const float* __restrict__ Kin,
                                           int kstride, int r, float* __restrict__ dirs, float* __restrict__ origins) {
  const float fx = Kin[0], sk = Kin[1], cx = Kin[2], fy = Kin[kstride + 1], cy = Kin[kstride + 2];
  const float u = uv[r * 2], v = uv[r * 2 + 1];
  const float xl = (u - cx + cy * sk / fy - sk * v / fy) / fx;
  const float yl = (v - cy) / fy;
  float w[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float wc = pose[c * 4 + 0] * xl + pose[c * 4 + 1] * yl + pose[c * 4 + 2] * 1.0f + pose[c * 4 + 3] * 1.0f;
    w[c] = wc - pose[c * 4 + 3];
  }
  const float n = fmaxf(sqrtf(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]), 1e-12f);
  dirs[r * 3 + 0] = w[0] / n; dirs[r * 3 + 1] = w[1] / n; dirs[r * 3 + 2] = w[2] / n;
  if (origins) {          // the camera centre once per ray (the callers' `cam_loc.unsqueeze(1).repeat(1, R, 1)`, rend_a :395)
#pragma unroll
    for (int c = 0; c < 3; ++c) origins[r * 3 + c] = pose[c * 4 + 3];
  }
}
__global__ void camera_rays_kernel(const float* __restrict__ uv, const float* __restrict__ pose, const float* __restrict__ Kin,
                                   int kstride, int R, float* __restrict__ dirs, float* __restrict__ origins) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  camera_ray(uv, pose, Kin, kstride, r, dirs, origins);
}

// eikonal points of a training step (rend_a :515-527): [uniform draws in the bounding cube | one point per ray at its drawn depth
// o + z d | optional extra points (the global junctions)] as one [2R + J, 3] array, one launch instead of addcmul + cat (+ cat)
__global__ void eik_points_kernel(const float* __restrict__ uniform, const float* __restrict__ o, const float* __restrict__ d,
                                  const float* __restrict__ z_eik, const float* __restrict__ extra, int R, int J, float* __restrict__ out,
                                  const float* __restrict__ z, int S, const long long* __restrict__ idx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = (2 * R + J) * 3;
  if (i >= n) return;
  const int p = i / 3, c = i - 3 * p;
  float v;
  if (p < R) v = uniform[i];
  else if (p < 2 * R) {
    const int r = p - R;
    const float ze = z_eik ? z_eik[r] : z[(size_t)r * S + idx[r]];     // the drawn depth, or the draw's index into the ray's depths
    v = o[3 * r + c] + rounded(ze * d[3 * r + c]);      // `cam_loc + z_samples_eik * ray_dirs` (rend_a :519-520): product rounded, then the sum, like the main pass's points
  }
  else v = extra[i - 6 * R];
  out[i] = v;
}

// d loss / d beta_param = sgn(beta_param) * sum over the rays of composite_bwd_kernel's per-ray partials (fixed order: one workgroup,
// strided per-thread sums, wave shuffles, waves in order) -- the `.sum()` and the backward of `.abs()` of density.py:29-30 in one launch
__device__ __forceinline__ void beta_grad_body(const float* __restrict__ dbeta_ray, int R, const float* __restrict__ beta_ptr,
                                               float* __restrict__ out) {
  __shared__ float s_w[4];
  float acc = 0.0f;
  for (int r = threadIdx.x; r < R; r += 256) acc += dbeta_ray[r];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float b = *beta_ptr;
    const float t = ((s_w[0] + s_w[1]) + s_w[2]) + s_w[3];
    out[0] = b > 0.0f ? t : (b < 0.0f ? -t : 0.0f);
  }
}
__global__ __launch_bounds__(256) void beta_grad_kernel(const float* __restrict__ dbeta_ray, int R, const float* __restrict__ beta_ptr,
                                                        float* __restrict__ out) {
  beta_grad_body(dbeta_ray, R, beta_ptr, out);
}

}  // namespace neat

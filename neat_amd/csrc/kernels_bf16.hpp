// bf16 build of the two GEMM-class kernels: v_mfma_f32_32x32x16_bf16 (fp32 accumulate).
//
// Hidden activations live in HBM as bf16 in OCTET-MAJOR layout  [feature/8][point][8]  (16 B = 8 consecutive
// features of one point).  That is at once
//   * the MFMA B-operand fragment (lane = point, 8 consecutive k) -> LDS staging is a straight 16-byte copy and
//     the operand fetch is one conflict-free ds_read_b128, and
//   * what the accumulator layout produces naturally (lane = point, 4 consecutive output rows per register quad
//     -> one 8-byte store), so a chain needs no transposes at all.
// Small arrays (PE rows, normals, rgb/offset cotangents, the sdf row) stay fp32 feature-major and are converted
// while staging.  The weight-gradient kernel, whose reduction runs over POINTS, transposes 8x8 blocks in registers
// (v_perm_b32) while staging, so a single HBM layout serves both consumers.
#pragma once
#include "kernels.hpp"

namespace neat {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

constexpr int BMH = 128;         // point-stride granule of the bf16 build (ldp is a multiple of this)

__device__ __forceinline__ u16 f2bf(float f) {            // round to nearest even (finite inputs)
  unsigned u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (u16)(u >> 16);
}
__device__ __forceinline__ float bf2f(u16 h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ unsigned pack2(float lo, float hi) { return (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16); }

// branch-free activation math for the bf16 build (hardware exp/log; absolute error ~1e-7, far below bf16 resolution)
__device__ __forceinline__ float softplus100_fast(float a) {
  const float t = 100.0f * a;
  return (fmaxf(t, 0.0f) + __logf(1.0f + __expf(-fabsf(t)))) * 0.01f;
}
__device__ __forceinline__ float dphi_fast(float h) { return 1.0f - __expf(-100.0f * h); }

// element (f, p) of an octet-major bf16 array
__device__ __forceinline__ size_t oct_index(int f, int p, int ldp) { return ((size_t)(f >> 3) * ldp + p) * 8 + (f & 7); }

struct SegH { const void* p; int rows; int bf16; };        // rows [rows][ldp]: fp32 feature-major or bf16 octet-major

struct LayerArgsH {
  SegH in[2];                 // packed input = [seg0 rows, zero pad to a multiple of 8 | seg1 rows | zero pad to Kpad]
  int Kpad;                   // multiple of 16
  const uint4* Wp;            // packed bf16 weights [NT][Kpad/16][64 lanes] x 8 bf16
  const float* bias;          // source bias; packed row n reads bias[(n + bias_rot) mod bias_n]; null = none
  int bias_rot, bias_n;
  int N, NT;                  // valid output rows / 32-row tiles computed
  int ldp;                    // multiple of 128
  void* out0; void* out1;     // out1 receives rows >= n_split (row - n_split)
  int out0_bf16, out1_bf16, n_split, accumulate;
  const u16* aux0; const u16* aux1;     // bf16 octet-major, same row indexing as out0
};

__device__ __forceinline__ float seg_read_f32(const SegH& s, int row, int p, int ldp) {
  return reinterpret_cast<const float*>(s.p)[(size_t)row * ldp + p];
}

// stage one 8-row octet x 128 points of a segment into the LDS tile (dst = &tile[octet][0][0])
template <int BMT>
__device__ __forceinline__ void stage_octet(const SegH& s, int oct, int p0, int ldp, uint4* dst, int tid) {
  // BMT points per octet row; 256/BMT octet rows are staged per pass (caller strides octets accordingly)
  const int p = tid & (BMT - 1);
  uint4 v = make_uint4(0u, 0u, 0u, 0u);
  if (s.bf16) {
    if (oct * 8 < s.rows) v = reinterpret_cast<const uint4*>(s.p)[(size_t)oct * ldp + p0 + p];
  } else {
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int row = oct * 8 + e;
      f[e] = row < s.rows ? seg_read_f32(s, row, p0 + p, ldp) : 0.0f;
    }
    v = make_uint4(pack2(f[0], f[1]), pack2(f[2], f[3]), pack2(f[4], f[5]), pack2(f[6], f[7]));
  }
  dst[p] = v;
}

template <int EPI>
__device__ __forceinline__ void epilogue_tile_h(const LayerArgsH& a, const f32x16& acc, int nt, int pt, int lane, int p0) {
  const int p = p0 + pt * 32 + (lane & 31);
  const int hi = lane >> 5;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int n0 = nt * 32 + 8 * q + 4 * hi;               // 4 consecutive rows n0..n0+3
    if (n0 >= ((a.N + 7) & ~7)) continue;                  // rows inside the last valid octet are still written (as zeros)
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = acc[4 * q + e];
    const size_t oidx = ((size_t)(n0 >> 3) * a.ldp + p) * 8 + (n0 & 7);     // bf16 element index of row n0 (octet-major)
    float x0[4] = {0.f, 0.f, 0.f, 0.f}, x1[4] = {0.f, 0.f, 0.f, 0.f};
    if (EPI == EPI_REV || EPI == EPI_TAN || EPI == EPI_BWD || EPI == EPI_BWD_RELU) {
      const uint2 r = *reinterpret_cast<const uint2*>(a.aux0 + oidx);
      x0[0] = bf2f((u16)(r.x & 0xFFFF)); x0[1] = bf2f((u16)(r.x >> 16)); x0[2] = bf2f((u16)(r.y & 0xFFFF)); x0[3] = bf2f((u16)(r.y >> 16));
    }
    if (EPI == EPI_TAN || EPI == EPI_BWD) {
      const uint2 r = *reinterpret_cast<const uint2*>(a.aux1 + oidx);
      x1[0] = bf2f((u16)(r.x & 0xFFFF)); x1[1] = bf2f((u16)(r.x >> 16)); x1[2] = bf2f((u16)(r.y & 0xFFFF)); x1[3] = bf2f((u16)(r.y >> 16));
    }
    float o0[4], o1[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = n0 + e;
      float b = 0.0f;
      if (a.bias && n < a.N) { int bi = n + a.bias_rot; if (bi >= a.bias_n) bi -= a.bias_n; b = a.bias[bi]; }
      float r0 = 0.0f, r1 = 0.0f;
      if (EPI == EPI_LINEAR) r0 = v[e] + b;
      else if (EPI == EPI_SOFTPLUS) r0 = softplus100_fast(v[e] + b);
      else if (EPI == EPI_RELU) r0 = fmaxf(v[e] + b, 0.0f);
      else if (EPI == EPI_SIGMOID) r0 = 1.0f / (1.0f + __expf(-(v[e] + b)));
      else if (EPI == EPI_REV) r0 = (n < a.n_split) ? v[e] * dphi_fast(x0[e]) : v[e];
      else if (EPI == EPI_TAN) { const float s = dphi_fast(x0[e]); r0 = v[e] * s; r1 = v[e] * x1[e] * (100.0f * (1.0f - s)); }
      else if (EPI == EPI_BWD) r0 = v[e] * dphi_fast(x0[e]) + x1[e];
      else if (EPI == EPI_BWD_RELU) r0 = x0[e] > 0.0f ? v[e] : 0.0f;
      if (n >= a.N) { r0 = 0.0f; r1 = 0.0f; }             // padded rows of the last octet stay finite zeros
      o0[e] = r0; o1[e] = r1;
    }
    if (n0 < a.n_split) {
      if (a.out0_bf16) {
        *reinterpret_cast<uint2*>(reinterpret_cast<u16*>(a.out0) + oidx) = make_uint2(pack2(o0[0], o0[1]), pack2(o0[2], o0[3]));
      } else {
        float* o = reinterpret_cast<float*>(a.out0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int n = n0 + e;
          if (n < a.N && n < a.n_split) {
            const size_t idx = (size_t)n * a.ldp + p;
            o[idx] = a.accumulate ? o[idx] + o0[e] : o0[e];
          }
        }
      }
      if (EPI == EPI_TAN)
        *reinterpret_cast<uint2*>(reinterpret_cast<u16*>(a.out1) + oidx) = make_uint2(pack2(o1[0], o1[1]), pack2(o1[2], o1[3]));
    }
    if (EPI == EPI_LINEAR || EPI == EPI_REV) {             // split outputs: rows >= n_split go to out1 (fp32 feature-major)
      if (n0 + 3 >= a.n_split && a.out1) {
        float* o = reinterpret_cast<float*>(a.out1);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int n = n0 + e;
          if (n >= a.n_split && n < a.N) o[(size_t)(n - a.n_split) * a.ldp + p] = o0[e];
        }
      }
    }
  }
}

template <int NTW, int PT>
__device__ __forceinline__ void mma_rows_h(f32x16 (&acc)[2][PT], const uint4* __restrict__ wp0, int tile_stride,
                                           const uint4* __restrict__ bl, int s_begin, int s_end) {
  constexpr int BMT = 32 * PT;
  // Weight fragments come from L2 (packed, 1 KiB per wave-load).  One k-step is only 4*NTW MFMAs (~130-260 cycles),
  // shorter than an L2 round trip, so keep a 4-deep register ring: the load for step s+3 is issued before the MFMAs of
  // step s.  Ring slots are compile-time indices (the loop advances by 4).
  uint4 ring[4][NTW];
#pragma unroll
  for (int u = 0; u < 3; ++u)
    if (s_begin + u < s_end) {
#pragma unroll
      for (int i = 0; i < NTW; ++i) ring[u][i] = wp0[(size_t)i * tile_stride + (s_begin + u) * 64];
    }
  for (int s = s_begin; s < s_end; s += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (s + u >= s_end) break;
      if (s + u + 3 < s_end) {
#pragma unroll
        for (int i = 0; i < NTW; ++i) ring[(u + 3) & 3][i] = wp0[(size_t)i * tile_stride + (s + u + 3) * 64];
      }
      uint4 bv[PT];
#pragma unroll
      for (int q = 0; q < PT; ++q) bv[q] = bl[(2 * (s + u)) * BMT + q * 32];
#pragma unroll
      for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int q = 0; q < PT; ++q)
          acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8*>(&ring[u][i]), *reinterpret_cast<bf16x8*>(&bv[q]),
                                                              acc[i][q], 0, 0, 0);
    }
  }
}

template <int EPI, int PT>
__global__ __launch_bounds__(WG, (PT == 4 ? 2 : 3)) void layer_kernel_h(LayerArgsH a) {
  constexpr int BMT = 32 * PT;                                       // points per workgroup: 128 (PT=4) or 64 (PT=2)
  extern __shared__ __attribute__((aligned(16))) uint4 ldsq[];     // [Kpad/8][BMT] octets (16 B each); reused for split-K reduce
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p0 = blockIdx.x * BMT;
  const int K8 = a.Kpad >> 3;
  const int oct0 = (a.in[0].rows + 7) >> 3;                         // octets of segment 0 (padded)
  const int oct1 = (a.in[1].rows + 7) >> 3;
  {
    // bf16 octet-major segments: the tile image is a straight copy -> global_load_lds DMA, 1 KiB per wave-instruction
    // (LDS destination = wave-uniform base + lane*16, exactly the [octet][point] tile row).
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* gbl_ptr;
    for (int o = wave; o < K8; o += 4) {
      const SegH& sg = (o < oct0) ? a.in[0] : a.in[1];
      const int so = (o < oct0) ? o : o - oct0;
      const bool has = (o < oct0 + oct1) && sg.bf16 && (so * 8 < sg.rows);
      if (has) {
        const uint4* src = reinterpret_cast<const uint4*>(sg.p) + (size_t)so * a.ldp + p0 + lane;
        uint4* dst = ldsq + (size_t)o * BMT;
#pragma unroll
        for (int hh = 0; hh < BMT / 64; ++hh)
          __builtin_amdgcn_global_load_lds((gbl_ptr)(src + 64 * hh), (lds_ptr)(dst + 64 * hh), 16, 0, 0);
      }
    }
    // fp32 feature-major segments (few rows: PE, head inputs, cotangents) and zero padding: convert through registers
    constexpr int OPP = 256 / BMT;                                    // octet rows per pass
    for (int o = tid / BMT; o < K8; o += OPP) {
      const SegH& sg = (o < oct0) ? a.in[0] : a.in[1];
      const int so = (o < oct0) ? o : o - oct0;
      const bool dma = (o < oct0 + oct1) && sg.bf16 && (so * 8 < sg.rows);
      if (dma) continue;
      uint4* dst = ldsq + (size_t)o * BMT;
      if (o < oct0 + oct1 && !sg.bf16) stage_octet<BMT>(sg, so, p0, a.ldp, dst, tid);
      else dst[tid & (BMT - 1)] = make_uint4(0u, 0u, 0u, 0u);
    }
  }
  __syncthreads();
  const int KS = a.Kpad >> 4;
  const uint4* bl = ldsq + (size_t)(lane >> 5) * BMT + (lane & 31);
  f32x16 acc[2][PT];
  if (a.NT > 2) {
    const int tstride = 4 * KS * 64;
    for (int round = 0; round * 8 < a.NT; ++round) {                // wave w owns tiles 8r+w, 8r+w+4
      const int t0 = round * 8 + wave;
      const int ntw = (t0 + 4 < a.NT) ? 2 : (t0 < a.NT ? 1 : 0);
      if (ntw == 0) continue;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < PT; ++q)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][q][r] = 0.0f;
      const uint4* wp0 = a.Wp + (size_t)t0 * KS * 64 + lane;
      if (ntw == 2) mma_rows_h<2, PT>(acc, wp0, tstride, bl, 0, KS);
      else mma_rows_h<1, PT>(acc, wp0, tstride, bl, 0, KS);
#pragma unroll
      for (int i = 0; i < 2; ++i)
        if (i < ntw) {
#pragma unroll
          for (int q = 0; q < PT; ++q) epilogue_tile_h<EPI>(a, acc[i][q], t0 + 4 * i, q, lane, p0);
        }
    }
  } else {
    // narrow outputs (N <= 64): split K over the waves, reduce through LDS
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < PT; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][q][r] = 0.0f;
    const int ksplit = 4 / a.NT;
    const int tile = wave % a.NT, kpart = wave / a.NT;
    const int per = (KS + ksplit - 1) / ksplit;
    const int sb = kpart * per, se = min(KS, sb + per);
    const uint4* wp0 = a.Wp + (size_t)tile * KS * 64 + lane;
    if (sb < se) mma_rows_h<1, PT>(acc, wp0, 0, bl, sb, se);
    __syncthreads();
    float* red = reinterpret_cast<float*>(ldsq);                    // [4 waves][PT ptiles][16][64]
#pragma unroll
    for (int q = 0; q < PT; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[((wave * PT + q) * 16 + r) * 64 + lane] = acc[0][q][r];
    __syncthreads();
    if (kpart == 0) {
#pragma unroll
      for (int q = 0; q < PT; ++q) {
        f32x16 sum;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = 0.0f;
          for (int kp = 0; kp < ksplit; ++kp) v += red[(((kp * a.NT + tile) * PT + q) * 16 + r) * 64 + lane];
          sum[r] = v;
        }
        epilogue_tile_h<EPI>(a, sum, tile, q, lane, p0);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// bf16 weight gradient: dW[n][k] = sum_p A[n][p] B[k][p], fp32 partial tiles as in the fp32 build
// ---------------------------------------------------------------------------------------------
struct WgradPairH {
  SegH A; int A_rot, A_mod;     // fp32 A only: packed row n reads A row (n + A_rot) mod A_mod (A_mod = 0: identity)
  SegH B[3];
  int padB0;                    // packed columns occupied by B[0] (rows rounded up to 8 when B[0] is octet-major)
};
struct WgradArgsH {
  WgradPairH pair[2]; int npairs;
  int N, Kt, P, ldp, chunk;      // chunk: points per grid.y slice (multiple of 64)
  float* partial; size_t row_stride, split_stride; int ktiles;    // (split, n, k) at n*row_stride + split*split_stride + k
  int bias_col;                  // wgrad_kernel_h2: partial column receiving the row sums of pair 0's A (or -1)
};

constexpr int HBP = 64;                 // points per staging step
constexpr int HLD = HBP * 2 + 16;       // LDS row stride in bytes (128 B data + 16 B pad -> conflict-free b128 reads)

// rows [row0, row0+8) x points [p, p+8) of a segment -> 8 uint4 (out[f] = 8 consecutive points of row row0+f)
__device__ __forceinline__ void load_block_T(const SegH& s, int row0, int p, int ldp, int pend, uint4 (&out)[8]) {
  if (s.p == nullptr || row0 >= s.rows) {
#pragma unroll
    for (int f = 0; f < 8; ++f) out[f] = make_uint4(0u, 0u, 0u, 0u);
    return;
  }
  if (s.bf16) {
    unsigned in[8][4];
    const uint4* src = reinterpret_cast<const uint4*>(s.p) + (size_t)(row0 >> 3) * ldp + p;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const uint4 v = (p + t < pend) ? src[t] : make_uint4(0u, 0u, 0u, 0u);
      in[t][0] = v.x; in[t][1] = v.y; in[t][2] = v.z; in[t][3] = v.w;
    }
    // 8x8 transpose of 16-bit elements inside the lane: out[f].dword[j] = { in[2j].feat f , in[2j+1].feat f }
#pragma unroll
    for (int f = 0; f < 8; ++f) {
      unsigned d[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned lo = in[2 * j][f >> 1], hi = in[2 * j + 1][f >> 1];
        d[j] = (f & 1) ? __builtin_amdgcn_perm(hi, lo, 0x07060302u) : __builtin_amdgcn_perm(hi, lo, 0x05040100u);
      }
      out[f] = make_uint4(d[0], d[1], d[2], d[3]);
    }
    // rows beyond the segment inside its last octet are zero by construction of the producers
  } else {
    const float* base = reinterpret_cast<const float*>(s.p);
#pragma unroll
    for (int f = 0; f < 8; ++f) {
      const int row = row0 + f;
      float v[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) v[t] = (row < s.rows && p + t < pend) ? base[(size_t)row * ldp + p + t] : 0.0f;
      out[f] = make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
    }
  }
}

__global__ __launch_bounds__(WG) void wgrad_kernel_h(WgradArgsH a) {
  __shared__ __attribute__((aligned(16))) unsigned char As[128 * HLD];
  __shared__ __attribute__((aligned(16))) unsigned char Bs[128 * HLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tn = blockIdx.x / a.ktiles, tk = blockIdx.x % a.ktiles;
  const int n0 = tn * 128, k0 = tk * 128;
  const int wr = (wave >> 1) * 64, wc = (wave & 1) * 64;
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
  // staging role: threads 0..127 -> A blocks, 128..255 -> B blocks; block = (row octet 0..15, point group 0..7)
  const int sid = tid & 127;
  const int boct = sid >> 3, bpg = sid & 7;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    if (q >= a.npairs) break;
    const WgradPairH& pr = a.pair[q];
    for (int pb = pbeg; pb < pend; pb += HBP) {
      uint4 blk[8];
      const int p = pb + bpg * 8;
      if (tid < 128) {
        const int row = n0 + boct * 8;
        if (pr.A.bf16 || pr.A_mod == 0) {
          load_block_T(pr.A, row, p, a.ldp, pend, blk);
        } else {
#pragma unroll
          for (int f = 0; f < 8; ++f) {
            int rr = row + f + pr.A_rot;
            if (rr >= pr.A_mod) rr -= pr.A_mod;
            const bool okr = (row + f < pr.A_mod) && rr < pr.A.rows;
            float v[8];
#pragma unroll
            for (int t = 0; t < 8; ++t)
              v[t] = (okr && p + t < pend) ? reinterpret_cast<const float*>(pr.A.p)[(size_t)rr * a.ldp + p + t] : 0.0f;
            blk[f] = make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
          }
        }
      } else {
        const int col = k0 + boct * 8;               // packed column of this 8-row block
        if (pr.B[0].bf16 && col < pr.padB0) {
          load_block_T(pr.B[0], col, p, a.ldp, pend, blk);
        } else {                                     // fp32 segments, possibly straddling two of them: row by row
#pragma unroll
          for (int f = 0; f < 8; ++f) {
            int cc = col + f;
            const float* sp = nullptr;
            if (cc < pr.padB0) { if (!pr.B[0].bf16 && cc < pr.B[0].rows) sp = reinterpret_cast<const float*>(pr.B[0].p); }
            else {
              cc -= pr.padB0;
              if (cc < pr.B[1].rows) sp = reinterpret_cast<const float*>(pr.B[1].p);
              else { cc -= pr.B[1].rows; if (cc < pr.B[2].rows) sp = reinterpret_cast<const float*>(pr.B[2].p); }
            }
            float v[8];
#pragma unroll
            for (int t = 0; t < 8; ++t)
              v[t] = (sp && p + t < pend) ? sp[(size_t)cc * a.ldp + p + t] : 0.0f;
            blk[f] = make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
          }
        }
      }
      unsigned char* dst = (tid < 128 ? As : Bs) + (boct * 8) * HLD + bpg * 16;
#pragma unroll
      for (int f = 0; f < 8; ++f) *reinterpret_cast<uint4*>(dst + f * HLD) = blk[f];
      __syncthreads();
      const unsigned char* ap = As + (wr + (lane & 31)) * HLD + (lane >> 5) * 16;
      const unsigned char* bp = Bs + (wc + (lane & 31)) * HLD + (lane >> 5) * 16;
#pragma unroll
      for (int s = 0; s < HBP / 16; ++s) {
        uint4 a0 = *reinterpret_cast<const uint4*>(ap + s * 32), a1 = *reinterpret_cast<const uint4*>(ap + 32 * HLD + s * 32);
        uint4 b0 = *reinterpret_cast<const uint4*>(bp + s * 32), b1 = *reinterpret_cast<const uint4*>(bp + 32 * HLD + s * 32);
        if (live[0][0]) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8*>(&a0), *reinterpret_cast<bf16x8*>(&b0), acc[0][0], 0, 0, 0);
        if (live[0][1]) acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8*>(&a0), *reinterpret_cast<bf16x8*>(&b1), acc[0][1], 0, 0, 0);
        if (live[1][0]) acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8*>(&a1), *reinterpret_cast<bf16x8*>(&b0), acc[1][0], 0, 0, 0);
        if (live[1][1]) acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8*>(&a1), *reinterpret_cast<bf16x8*>(&b1), acc[1][1], 0, 0, 0);
      }
      __syncthreads();
    }
  }
  float* dstp = a.partial + (size_t)blockIdx.y * a.split_stride;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (!live[i][j]) continue;
      const int k = k0 + wc + 32 * j + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wr + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (n < a.N && k < a.Kt) dstp[(size_t)n * a.row_stride + k] = acc[i][j][r];
      }
    }
}

// ---------------------------------------------------------------------------------------------
// bf16 weight gradient, second generation: ONE workgroup (8 waves) owns a full 256x256 output tile, so every
// operand element is read from HBM exactly once per launch (the 128x128 version re-read A 3x and B 2x and was
// HBM-bound at ~800 MB per layer).  The bias gradient (row sums of A) comes from an extra MFMA against an all-ones
// B fragment -- no ones row in memory, no 257th column tile.  Staging loads for step t+1 are issued right after the
// LDS image of step t is written, so HBM latency hides under the 32..40 MFMAs of step t.
// ---------------------------------------------------------------------------------------------
constexpr int W2T = 512;                       // threads
__device__ __forceinline__ void w2_issue(const SegH& s, int row0, int p, int ldp, int pend, uint4 (&raw)[8]) {
  // raw octet loads of an 8-row x 8-point block of a bf16 octet-major segment (transposed later)
  const bool ok = s.p != nullptr && s.bf16 && row0 < s.rows;
  const uint4* src = reinterpret_cast<const uint4*>(s.p) + (size_t)(row0 >> 3) * ldp + p;
#pragma unroll
  for (int t = 0; t < 8; ++t) raw[t] = (ok && p + t < pend) ? src[t] : make_uint4(0u, 0u, 0u, 0u);
}
__device__ __forceinline__ void w2_transpose(const uint4 (&raw)[8], uint4 (&out)[8]) {
  unsigned in[8][4];
#pragma unroll
  for (int t = 0; t < 8; ++t) { in[t][0] = raw[t].x; in[t][1] = raw[t].y; in[t][2] = raw[t].z; in[t][3] = raw[t].w; }
#pragma unroll
  for (int f = 0; f < 8; ++f) {
    unsigned d[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned lo = in[2 * j][f >> 1], hi = in[2 * j + 1][f >> 1];
      d[j] = (f & 1) ? __builtin_amdgcn_perm(hi, lo, 0x07060302u) : __builtin_amdgcn_perm(hi, lo, 0x05040100u);
    }
    out[f] = make_uint4(d[0], d[1], d[2], d[3]);
  }
}
// fp32 feature-major rows (possibly spanning B[0..2] / rotated A) -> packed block, row by row
__device__ __forceinline__ void w2_rows_f32(const WgradPairH& pr, bool isA, int row0, int p, int ldp, int pend, uint4 (&out)[8]) {
#pragma unroll
  for (int f = 0; f < 8; ++f) {
    const float* sp = nullptr;
    int rr = row0 + f;
    if (isA) {
      if (pr.A_mod) { const bool in_range = rr < pr.A_mod; rr += pr.A_rot; if (rr >= pr.A_mod) rr -= pr.A_mod; if (!in_range) rr = 1 << 30; }
      if (!pr.A.bf16 && rr < pr.A.rows) sp = reinterpret_cast<const float*>(pr.A.p);
    } else {
      if (rr < pr.padB0) { if (!pr.B[0].bf16 && rr < pr.B[0].rows) sp = reinterpret_cast<const float*>(pr.B[0].p); }
      else {
        rr -= pr.padB0;
        if (rr < pr.B[1].rows) sp = reinterpret_cast<const float*>(pr.B[1].p);
        else { rr -= pr.B[1].rows; if (rr < pr.B[2].rows) sp = reinterpret_cast<const float*>(pr.B[2].p); }
      }
    }
    float v[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) v[t] = (sp && p + t < pend) ? sp[(size_t)rr * ldp + p + t] : 0.0f;
    out[f] = make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
  }
}

__global__ __launch_bounds__(W2T, 2) void wgrad_kernel_h2(WgradArgsH a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char w2lds[];     // As[256][HLD] | Bs[256][HLD]
  unsigned char* As = w2lds;
  unsigned char* Bs = w2lds + 256 * HLD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tn = blockIdx.x / a.ktiles, tk = blockIdx.x % a.ktiles;
  const int n0 = tn * 256, k0 = tk * 256;
  const int wr = (wave >> 1) * 64, wc = (wave & 1) * 128;
  bool liveR[2], liveC[4];
#pragma unroll
  for (int i = 0; i < 2; ++i) liveR[i] = n0 + wr + 32 * i < a.N;
#pragma unroll
  for (int j = 0; j < 4; ++j) liveC[j] = k0 + wc + 32 * j < a.Kt;
  const bool do_bias = (tk == 0) && a.bias_col >= 0;
  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  float rsum[8];                                            // bias gradient: row sums of pair 0's A, kept by the A-staging threads
#pragma unroll
  for (int f = 0; f < 8; ++f) rsum[f] = 0.0f;
  const int pbeg = blockIdx.y * a.chunk;
  const int pend = min(a.P, pbeg + a.chunk);
  const int nsteps = (pend - pbeg + HBP - 1) / HBP;
  const bool isA = tid < 256;
  const int sid = tid & 255;
  const int boct = sid >> 3, bpg = sid & 7;                 // 32 row octets x 8 point groups
  const int row0 = (isA ? n0 : k0) + boct * 8;
  // raw staging ring in LDS, filled by LDS-DMA (no VGPRs held across the MFMA phase): [wave][load t][lane] x 16 B
  uint4* rawl = reinterpret_cast<uint4*>(w2lds + 2 * 256 * HLD) + (size_t)wave * 8 * 64;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* gbl_ptr;
  // all kernarg accesses below use compile-time pair indices (a runtime-indexed a.pair[q] would be copied to scratch)
#define W2_FAST(pr) (isA ? ((pr).A.bf16 != 0) : ((pr).B[0].bf16 && row0 < (pr).padB0))
#define W2_SEG(pr) (isA ? (pr).A : (pr).B[0])
#define W2_ISSUE(pr, pb)                                                                                         \
  do {                                                                                                           \
    if (W2_FAST(pr) && W2_SEG(pr).p && row0 < W2_SEG(pr).rows) {                                                 \
      const uint4* src_ = reinterpret_cast<const uint4*>(W2_SEG(pr).p) + (size_t)(row0 >> 3) * a.ldp + (pb) + bpg * 8; \
      _Pragma("unroll") for (int t = 0; t < 8; ++t)                                                              \
        __builtin_amdgcn_global_load_lds((gbl_ptr)(src_ + t), (lds_ptr)(rawl + t * 64), 16, 0, 0);               \
    }                                                                                                            \
  } while (0)
  if (nsteps > 0) W2_ISSUE(a.pair[0], pbeg);
  __syncthreads();                                          // (the DMA is drained by the barrier's vmcnt(0))
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    if (q >= a.npairs) break;
    const WgradPairH& pr = a.pair[q];
    const bool fast = W2_FAST(pr);
    const bool okr = fast && W2_SEG(pr).p != nullptr && row0 < W2_SEG(pr).rows;
    const bool bias_now = do_bias && q == 0 && isA;
    for (int st = 0; st < nsteps; ++st) {
      const int pb = pbeg + st * HBP;
      uint4 blk[8];
      if (fast) {
        const int p = pb + bpg * 8;
        uint4 raw[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) raw[t] = (okr && p + t < pend) ? rawl[t * 64 + lane] : make_uint4(0u, 0u, 0u, 0u);
        w2_transpose(raw, blk);
      } else {
        w2_rows_f32(pr, isA, row0, pb + bpg * 8, a.ldp, pend, blk);
      }
      unsigned char* dst = (isA ? As : Bs) + (boct * 8) * HLD + bpg * 16;
#pragma unroll
      for (int f = 0; f < 8; ++f) *reinterpret_cast<uint4*>(dst + f * HLD) = blk[f];
      if (bias_now) {
#pragma unroll
        for (int f = 0; f < 8; ++f) {
          const unsigned w4[4] = {blk[f].x, blk[f].y, blk[f].z, blk[f].w};
          float t = 0.0f;
#pragma unroll
          for (int j = 0; j < 4; ++j) t += bf2f((u16)(w4[j] & 0xFFFF)) + bf2f((u16)(w4[j] >> 16));
          rsum[f] += t;
        }
      }
      __syncthreads();
      // next block's HBM latency hides under the MFMAs below
      if (st + 1 < nsteps) W2_ISSUE(pr, pb + HBP);
      else if (q == 0 && a.npairs > 1) W2_ISSUE(a.pair[1], pbeg);
      const unsigned char* ap = As + (wr + (lane & 31)) * HLD + (lane >> 5) * 16;
      const unsigned char* bp = Bs + (wc + (lane & 31)) * HLD + (lane >> 5) * 16;
#pragma unroll
      for (int s = 0; s < HBP / 16; ++s) {
        uint4 av[2], bv[4];
#pragma unroll
        for (int i = 0; i < 2; ++i) av[i] = *reinterpret_cast<const uint4*>(ap + i * 32 * HLD + s * 32);
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[j] = *reinterpret_cast<const uint4*>(bp + j * 32 * HLD + s * 32);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          if (!liveR[i]) continue;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (liveC[j])
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8*>(&av[i]), *reinterpret_cast<bf16x8*>(&bv[j]), acc[i][j], 0, 0, 0);
        }
      }
      __syncthreads();
    }
  }
#undef W2_ISSUE
#undef W2_SEG
#undef W2_FAST
  float* dstp = a.partial + (size_t)blockIdx.y * a.split_stride;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (!liveR[i]) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n0 + wr + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (n >= a.N) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = k0 + wc + 32 * j + (lane & 31);
        if (liveC[j] && k < a.Kt) dstp[(size_t)n * a.row_stride + k] = acc[i][j][r];
      }
    }
  }
  if (do_bias && isA) {                                     // combine the 8 point-group partials (adjacent lanes) of each row
#pragma unroll
    for (int f = 0; f < 8; ++f) {
      float v = rsum[f];
      v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
      if (bpg == 0 && row0 + f < a.N) dstp[(size_t)(row0 + f) * a.row_stride + a.bias_col] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// bf16 pack:  out[((nt*KS + s)*64 + lane)*8 + e] = Wm[nt*32 + (lane&31)][16 s + 8 (lane>>5) + e]
// (index maps shared with the fp32 pack via PackDesc2)
// ---------------------------------------------------------------------------------------------
struct PackDesc2 {
  int layer, transpose, N, K, Kpad, NT;
  int s0, s0p, off0, off1;      // packed input index j -> source column: j<s0p ? (j<s0 ? off0+j : none) : off1 + (j-s0p)
  int rot;                      // packed output index n -> source row (n + rot) mod O
  float scale; int offset; int blk0; int bf16;
};
constexpr int MAXPACKS2 = 48;
struct PackArgs2 { NetPtrs net; const float* rowscale; int row_off[NLAYERS + 1]; PackDesc2 d[MAXPACKS2]; int npacks; float* out; };

__device__ __forceinline__ float packed_weight(const PackDesc2& d, const float* v, const float* rs, int O, int I, int n, int k) {
  if (!v || n >= d.N || k >= d.K) return 0.0f;
  const int po = d.transpose ? k : n;        // packed output index
  const int pj = d.transpose ? n : k;        // packed input index
  if (po >= O) return 0.0f;
  const int o = (po + d.rot) % O;
  int i;
  if (pj < d.s0p) { if (pj >= d.s0) return 0.0f; i = d.off0 + pj; }
  else { i = d.off1 + (pj - d.s0p); if (pj - d.s0p >= I - d.s0) return 0.0f; }
  if (i >= I) return 0.0f;
  return v[(size_t)o * I + i] * rs[o] * d.scale;
}

__global__ __launch_bounds__(WG) void pack_kernel2(PackArgs2 a) {
  int pk = 0;
  while (pk + 1 < a.npacks && (int)blockIdx.x >= a.d[pk + 1].blk0) ++pk;
  const PackDesc2 d = a.d[pk];
  const int nt = blockIdx.x - d.blk0;
  const int O = a.net.O[d.layer], I = a.net.I[d.layer];
  const float* v = a.net.v[d.layer];
  const float* rs = a.rowscale + a.row_off[d.layer];
  if (d.bf16) {
    const int KS = d.Kpad >> 4;
    uint4* out = reinterpret_cast<uint4*>(a.out + d.offset) + (size_t)nt * KS * 64;
    for (int e = threadIdx.x; e < KS * 64; e += WG) {
      const int s = e >> 6, ln = e & 63;
      const int n = nt * 32 + (ln & 31), kb = 16 * s + 8 * (ln >> 5);
      float w[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) w[t] = packed_weight(d, v, rs, O, I, n, kb + t);
      out[e] = make_uint4(pack2(w[0], w[1]), pack2(w[2], w[3]), pack2(w[4], w[5]), pack2(w[6], w[7]));
    }
  } else {
    const int KS = d.Kpad >> 1;
    float* out = a.out + d.offset + (size_t)nt * KS * 64;
    for (int e = threadIdx.x; e < KS * 64; e += WG) {
      const int s = e >> 6, ln = e & 63;
      out[e] = packed_weight(d, v, rs, O, I, nt * 32 + (ln & 31), 2 * s + (ln >> 5));
    }
  }
}

// u7 = W8[0,:] * phi'(h8), bf16 octet-major in/out
__global__ void adjoint_seed_kernel_h(const float* __restrict__ v8, const float* __restrict__ rs8, const u16* __restrict__ h8,
                                      int ldp, u16* __restrict__ u7) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int oct = blockIdx.y;
  if (p >= ldp) return;
  const size_t idx = ((size_t)oct * ldp + p) * 8;
  const uint4 hv = *reinterpret_cast<const uint4*>(h8 + idx);
  const unsigned hw[4] = {hv.x, hv.y, hv.z, hv.w};
  unsigned o[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float w0 = v8[oct * 8 + 2 * j] * rs8[0], w1 = v8[oct * 8 + 2 * j + 1] * rs8[0];
    o[j] = pack2(w0 * dphi_fast(bf2f((u16)(hw[j] & 0xFFFF))), w1 * dphi_fast(bf2f((u16)(hw[j] >> 16))));
  }
  *reinterpret_cast<uint4*>(u7 + idx) = make_uint4(o[0], o[1], o[2], o[3]);
}

// bf16 octet-major [C rows] -> row-major fp32 [P, C] at column offset col0 of a [P, ldc] matrix
__global__ void oct_to_rm_kernel(const u16* __restrict__ src, int P, int C, int ldp, float* __restrict__ dst, int ldc, int col0) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  for (int c = 0; c < C; ++c) dst[(size_t)p * ldc + col0 + c] = bf2f(src[oct_index(c, p, ldp)]);
}
__global__ void rm_to_oct_kernel(const float* __restrict__ src, int P, int C, int ldp, u16* __restrict__ dst) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= ldp) return;
  const int Cp = (C + 7) & ~7;
  for (int c = 0; c < Cp; ++c) dst[oct_index(c, p, ldp)] = (p < P && c < C) ? f2bf(src[(size_t)p * C + c]) : (u16)0;
}
__global__ void fm_col_to_rm_kernel(const float* __restrict__ src, int P, float* __restrict__ dst, int ldc, int col) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < P) dst[(size_t)p * ldc + col] = src[p];
}

}  // namespace neat

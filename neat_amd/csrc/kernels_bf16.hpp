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
#include "bf16_common.hpp"

namespace neat {

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
  const u16* padfill_oct;                   // the same rows as an octet-major bf16 array (what layer_kernel_ws reads), or null
  const float* padfill; int padfill_rows;   // rows [N, N+padfill_rows) of out0 receive these fp32 rows (skip layer: the
                                            // first 7 PE rows ride in the padding of the 217-row h4 octets); out1 gets 0
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

template <int EPI, int PT, bool OBF>      // OBF: out0 is bf16 octet-major (compile-time: sheds the fp32 / split / accumulate code)
__device__ __forceinline__ void epilogue_rows_h(const LayerArgsH& a, const f32x16 (&acc)[PT], int nt, int lane, int p0) {
  // one 32-row output tile x PT point tiles.  Row-dependent state (validity, bias) is built once per 4-row quad and
  // reused over the point tiles, which keeps the number of live compare masks small (they used to spill by the hundred).
  const int hi = lane >> 5;
  const int Npad = (a.N + 7) & ~7;
  const int split_pad = (a.n_split + 7) & ~7;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int n0 = nt * 32 + 8 * q + 4 * hi;               // 4 consecutive rows n0..n0+3
    if (n0 >= Npad) continue;                              // rows inside the last valid octet are still written (as zeros)
    float b[4];
    bool valid[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = n0 + e;
      valid[e] = n < a.N;
      b[e] = 0.0f;
      if (a.bias && valid[e]) { int bi = n + a.bias_rot; if (bi >= a.bias_n) bi -= a.bias_n; b[e] = a.bias[bi]; }
    }
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      const int p = p0 + pt * 32 + (lane & 31);
      // bf16 element index of row n0 (octet-major); arrays are < 2^31 elements, keep the address math 32-bit
      const unsigned oidx = ((unsigned)(n0 >> 3) * (unsigned)a.ldp + (unsigned)p) * 8u + (unsigned)(n0 & 7);
      float x0[4] = {0.f, 0.f, 0.f, 0.f}, x1[4] = {0.f, 0.f, 0.f, 0.f};
      if (EPI == EPI_REV || EPI == EPI_TAN || EPI == EPI_BWD || EPI == EPI_BWD_RELU) {
        const uint2 r = *reinterpret_cast<const uint2*>(a.aux0 + oidx);
        x0[0] = bf_lo(r.x); x0[1] = bf_hi(r.x); x0[2] = bf_lo(r.y); x0[3] = bf_hi(r.y);
      }
      if (EPI == EPI_TAN || EPI == EPI_BWD) {
        const uint2 r = *reinterpret_cast<const uint2*>(a.aux1 + oidx);
        x1[0] = bf_lo(r.x); x1[1] = bf_hi(r.x); x1[2] = bf_lo(r.y); x1[3] = bf_hi(r.y);
      }
      float o0[4], o1[4], ox[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = n0 + e;
        const float v = acc[pt][4 * q + e];
        float r0 = 0.0f, r1 = 0.0f;
        if (EPI == EPI_LINEAR) r0 = v + b[e];
        else if (EPI == EPI_SOFTPLUS) r0 = softplus100_fast(v + b[e]);
        else if (EPI == EPI_RELU) r0 = fmaxf(v + b[e], 0.0f);
        else if (EPI == EPI_SIGMOID) r0 = 1.0f / (1.0f + __expf(-(v + b[e])));
        else if (EPI == EPI_REV) r0 = (n < a.n_split) ? v * dphi_fast(x0[e]) : v;
        else if (EPI == EPI_TAN) { const float sg = dphi_fast(x0[e]); r0 = v * sg; r1 = v * x1[e] * (100.0f * (1.0f - sg)); }
        else if (EPI == EPI_BWD) r0 = v * dphi_fast(x0[e]) + x1[e];
        else if (EPI == EPI_BWD_RELU) r0 = x0[e] > 0.0f ? v : 0.0f;
        if (!valid[e]) {                                     // padded rows of the last octet: finite zeros, or the pad-fill rows
          r0 = (a.padfill && n < a.N + a.padfill_rows) ? a.padfill[(unsigned)(n - a.N) * (unsigned)a.ldp + (unsigned)p] : 0.0f;
          r1 = 0.0f;
        }
        ox[e] = r0;                                          // value for out1 when this row lies beyond n_split
        if (EPI == EPI_REV && OBF && n >= a.n_split) r0 = 0.0f;    // ... and out0's octet stays clean (finite zeros)
        o0[e] = r0; o1[e] = r1;
      }
      if (n0 < split_pad) {                                  // (bf16 out0: finish the octet that contains the split row)
        if (OBF) {
          *reinterpret_cast<uint2*>(reinterpret_cast<u16*>(a.out0) + oidx) = make_uint2(pack2(o0[0], o0[1]), pack2(o0[2], o0[3]));
        } else {
          float* o = reinterpret_cast<float*>(a.out0);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int n = n0 + e;
            if (valid[e] && n < a.n_split) {
              const unsigned idx = (unsigned)n * (unsigned)a.ldp + (unsigned)p;
              o[idx] = a.accumulate ? o[idx] + o0[e] : o0[e];
            }
          }
        }
        if (EPI == EPI_TAN)
          *reinterpret_cast<uint2*>(reinterpret_cast<u16*>(a.out1) + oidx) = make_uint2(pack2(o1[0], o1[1]), pack2(o1[2], o1[3]));
      }
      if ((EPI == EPI_LINEAR || EPI == EPI_REV) && (EPI == EPI_REV || !OBF || true)) {   // split outputs: rows >= n_split go to out1 (fp32 feature-major)
        if (n0 + 3 >= a.n_split && a.out1) {
          float* o = reinterpret_cast<float*>(a.out1);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int n = n0 + e;
            if (n >= a.n_split && valid[e]) o[(unsigned)(n - a.n_split) * (unsigned)a.ldp + (unsigned)p] = ox[e];
          }
        }
      }
    }
  }
}

template <int NTW, int PT>
__device__ __forceinline__ void mma_rows_h(f32x16 (&acc)[2][PT], const uint4* __restrict__ wp0, int tile_stride,
                                           const uint4* __restrict__ bl, int s_begin, int s_end) {
  // Weight fragments come from L2 (packed, 1 KiB per wave-load).  One k-step is only PT*NTW MFMAs (~130-260 cycles),
  // shorter than an L2 round trip, so keep a 4-deep register ring: the load for step s+3 is issued before the MFMAs of
  // step s.  The loop is BRANCH-FREE (trip count a multiple of 4, prefetch index clamped at the tail): with guards the
  // compiler joins control flow with s_waitcnt vmcnt(0) and the ring degenerates into load->wait->use.
  constexpr int BMT = 32 * PT;
  uint4 ring[4][NTW];
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int sp = min(s_begin + u, s_end - 1);
#pragma unroll
    for (int i = 0; i < NTW; ++i) ring[u][i] = wp0[(size_t)i * tile_stride + sp * 64];
  }
  for (int s = s_begin; s < s_end; s += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int sp = min(s + u + 3, s_end - 1);
#pragma unroll
      for (int i = 0; i < NTW; ++i) ring[(u + 3) & 3][i] = wp0[(size_t)i * tile_stride + sp * 64];
      __builtin_amdgcn_sched_barrier(0);      // pin the prefetch HERE: the scheduler otherwise sinks it next to its use 3 steps later
      uint4 bv[PT];
#pragma unroll
      for (int q = 0; q < PT; ++q) bv[q] = bl[(2 * (s + u)) * BMT + q * 32];
#pragma unroll
      for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int q = 0; q < PT; ++q)
          acc[i][q] = NEAT_MFMA16(*reinterpret_cast<bf16x8*>(&ring[u][i]), *reinterpret_cast<bf16x8*>(&bv[q]),
                                                              acc[i][q], 0, 0, 0);
    }
  }
}

template <int EPI, int PT, bool OBF>
__global__ __launch_bounds__(WG, 2) void layer_kernel_h(LayerArgsH a) {
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
        if (i < ntw) epilogue_rows_h<EPI, PT, OBF>(a, acc[i], t0 + 4 * i, lane, p0);
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
    const int per = ((KS / 4 + ksplit - 1) / ksplit) * 4;            // k-steps per wave, a multiple of 4 (KS is one too)
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
      f32x16 sum[PT];
#pragma unroll
      for (int q = 0; q < PT; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = 0.0f;
          for (int kp = 0; kp < ksplit; ++kp) v += red[(((kp * a.NT + tile) * PT + q) * 16 + r) * 64 + lane];
          sum[q][r] = v;
        }
      epilogue_rows_h<EPI, PT, OBF>(a, sum, tile, lane, p0);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// layer_kernel_ws: weight-stationary streaming version of the hidden 256 -> 256 layers of every chain (all arrays bf16
// octet-major, K = 256, N <= 256).  layer_kernel_h re-fetches its 128 KiB weight matrix from L2 for every 64-point
// workgroup through a 4 KiB-per-wave window and is bound by that latency; here
//  * one persistent workgroup (8 waves) per CU; wave w keeps rows 32w..32w+31 of W as 16 MFMA A fragments in 64 VGPRs
//    for the whole launch -- no weight traffic in the loop;
//  * points stream through an LDS ring of 32-point stages filled by LDS-DMA (issued from inline asm, see
//    wgrad_kernel_h3): the input tile [32 octets][32 points][16 B] is shared by the waves (B fragments by ds_read_b128),
//    the epilogue operands (h for phi', u / m) are DMA'd per wave for its own 32 rows and read back in accumulator
//    layout by ds_read_b64 -- no VGPR-destination global load inside the loop, so the counted waits stay exact;
//  * D = NS-1 stages in flight across raw s_barriers; outputs leave as 8-byte stores straight from the accumulators.
// Stores share the vector-memory counter with the DMA and may retire out of order with it: waiting for
// "at most (later DMA instructions) outstanding" is sufficient whatever the stores do (DMAs retire in order).
// ---------------------------------------------------------------------------------------------
constexpr int WST = 512, WSP = 32;
constexpr int WS_TILE = 32 * WSP * 16;                   // 16 KiB: [32 octets][32 points][16 B]
struct LayerArgsWS {
  const u16* in; const uint4* Wp; const float* bias;
  const float* srow; const float* wrow; const float* wrow_scale;     // EPI_BWD8: per-point scalar, per-row weight and its scale
  const u16* aux0; const u16* aux1;
  u16* out0; u16* out1;
  int N, in_octs, ldp, ntiles, per_wg;                   // 32-point tiles in total / per workgroup (contiguous)
  int kstride;                                           // k-steps per row tile in the pack (16, or 17 when a 257th column follows)
  const u16* in2; int split_oct;                         // input octets >= split_oct come from in2 (skip layer: [vh4 | PE^]); 32 = none
  int n_split; float* out1f;                             // EPI_REV: rows >= n_split leave as fp32 feature-major rows (n - n_split), no phi'
  float* out0f;                                          // OUTF variants: out0 is fp32 feature-major [N][ldp]
  int x_octs;                                            // KS = 20: valid octets of in2 (the rest of its 8-octet slot is zero weight)
  const u16* padfill;                                    // EPI_TAN_PF: octet-major array whose rows 0..6 fill rows N..N+6 of out0
  int xcd_major;                                         // interleaved tiles: workgroup -> tile slot permuted so that an XCD owns a contiguous run
  int wide_store;                                        // 1: bf16 outputs leave as 16-byte stores (v_permlane32_swap), 0: 8-byte
  int aux_nt;                                            // non-temporal: bit 0 / 1 fetch of aux0 / aux1, bit 2 fetch of `in`, bit 3 store of out1
  int tile_stride;                                       // 1: workgroup w owns tiles [w per_wg, (w+1) per_wg); gridDim.x: tiles w, w + grid, ...
  const float* rho_num; const float* rho_den;            // EPI_LINACC: acc is scaled by cot_scale_of(rho_num) / cot_scale_of(rho_den) before aux0 is added
                                                         // (the two heads' backward chains run in different cotangent scales; null = 1)
};
// epilogues that exist only in the weight-stationary kernel
constexpr int EPI_LINACC = 8;      // out0 = acc + aux0                      (feature cotangent: second head adds to the first)
constexpr int EPI_BWD8 = 9;        // out0 = (acc + wrow[n] s[p]) phi'(aux0) + aux1   (first layer of the reverse chain: the
                                   // sdf row of lin8 enters as a rank-1 term, s = cotangent of the raw sdf, fp32 per point)
constexpr int EPI_TAN_PF = 10;     // EPI_TAN for lin3 (217 rows): rows 217..223 of out0 <- rows 0..6 of a small octet-major array
                                   // (the PE tangent rows that ride in the padding of the skip layer's input), out1 <- 0
template <int EPI, int KS = 16> struct WsCfg {
  static constexpr int NAUX = (EPI == EPI_TAN || EPI == EPI_TAN_PF || EPI == EPI_BWD || EPI == EPI_BWD8) ? 2
                              : ((EPI == EPI_REV || EPI == EPI_BWD_RELU || EPI == EPI_LINACC) ? 1 : 0);
  static constexpr int HAS_S = EPI == EPI_BWD8 ? 1 : 0;
  static constexpr int HAS_PF = EPI == EPI_TAN_PF ? 1 : 0;
  static constexpr int XOCT = KS > 16 ? 8 : 0;           // input octets beyond the 32 of the main array (K = 320: [256 | 64])
  static constexpr int IN_TILE = (32 + XOCT) * WSP * 16;
  static constexpr int NS = NAUX == 2 ? 3 : 4;           // ring depth: 3 x 48 KiB or 4 x (16|20|32) KiB (8 x 16 KiB / 6 x 20 KiB measured slower, twice)
  static constexpr int STAGE = IN_TILE + WS_TILE * NAUX + HAS_S * 8 * 256 + HAS_PF * 1024;
  static constexpr int G = 2 * (1 + NAUX) + HAS_S + HAS_PF + (XOCT ? 1 : 0);       // DMA instructions per stage per wave
  static constexpr int LDS = NS * STAGE;
};

__device__ __forceinline__ void ws_wait_barrier(int n) {     // n = DMA instructions of this wave allowed to stay in flight
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)\n\ts_barrier" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)\n\ts_barrier" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)\n\ts_barrier" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)\n\ts_barrier" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); break;      // (not reached: every G x depth is listed)
  }
}

// KS: k-steps (16: K = 256; 1: K <= 16, the narrow cotangents entering the heads' backward); OUTF: narrow fp32 output
template <int EPI, int KS = 16, bool OUTF = false>
__global__ __launch_bounds__(WST, 2) void layer_kernel_ws(LayerArgsWS a) {
  typedef WsCfg<EPI, KS> C;
  extern __shared__ __attribute__((aligned(16))) unsigned char wslds[];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // tile of step tau: contiguous ranges (tile_stride 1) or interleaved over the workgroups (tile_stride = gridDim.x: at any
  // moment the machine works on one contiguous window of every array)
  // (xcd_major: consecutive workgroups go to different XCDs; give each XCD a contiguous run of the interleaved tiles instead)
  const int wq = (a.xcd_major && (gridDim.x & 7) == 0) ? (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int t_begin = a.tile_stride == 1 ? blockIdx.x * a.per_wg : wq;
  const int T = a.tile_stride == 1 ? min(a.per_wg, a.ntiles - t_begin) : (a.ntiles - t_begin + a.tile_stride - 1) / a.tile_stride;
  if (T <= 0) return;
  const unsigned lds_base = (unsigned)(size_t)(lds_ptr)wslds;
  // DMA: per stage this wave moves octets 4w..4w+3 of the input and of each epilogue operand (2 instructions each:
  // 2 octets x 32 points); LDS image [octet][point] x 16 B is lane-linear per instruction
  const unsigned dma_off = (unsigned)(4 * wave) * (WSP * 16);
  auto dma = [&](const u16* base, int tile, unsigned dst, int max_oct, const u16* base2 = nullptr, int split = 1 << 30, bool nt = false) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      // octets past the end of a 217-row array would be uninitialised memory (x zero weight = NaN): re-read a valid one
      int oct = min(4 * wave + 2 * i + (lane >> 5), max_oct);
      const u16* b = base;
      if (oct >= split) { b = base2; oct -= split; }
      const u16* s2 = b + ((size_t)oct * a.ldp + (size_t)tile * WSP + (lane & 31)) * 8;
      const unsigned d2 = __builtin_amdgcn_readfirstlane(dst + i * (2 * WSP * 16));
      unsigned keep;
      if (nt)       // read-once operand written long ago: non-temporal, so that it does not displace the chain's own arrays from the caches
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(s2), "s"(d2) : "memory");
      else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(s2), "s"(d2) : "memory");
    }
  };
  auto dma1 = [&](const void* src, unsigned dst, bool wide) {
    const unsigned d2 = __builtin_amdgcn_readfirstlane(dst);
    unsigned keep;
    if (wide)
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(src), "s"(d2) : "memory");
    else
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(src), "s"(d2) : "memory");
  };
  constexpr unsigned AUX0 = C::IN_TILE, AUX1 = C::IN_TILE + WS_TILE, EXTRA = C::IN_TILE + WS_TILE * C::NAUX;
  auto issue = [&](int tau) {
    const unsigned stage = lds_base + (unsigned)(tau % C::NS) * C::STAGE;
    const unsigned slot = stage + dma_off;
    const int tile = t_begin + tau * a.tile_stride;
    dma(a.in, tile, slot, a.in_octs - 1, a.in2, a.split_oct, (a.aux_nt & 4) != 0);
    if (C::XOCT) {           // octets 32..39 of the packed input come from the small second array (2 octets x 32 points per
                             // instruction; waves 4..7 repeat what waves 0..3 fetch so that every wave issues the same count)
      const int oct = min(2 * (wave & 3) + (lane >> 5), a.x_octs - 1);
      dma1(a.in2 + ((size_t)oct * a.ldp + (size_t)tile * WSP + (lane & 31)) * 8, stage + (32 + 2 * (wave & 3)) * (WSP * 16), true);
    }
    if (C::NAUX >= 1) dma(a.aux0, tile, slot + AUX0, 31, nullptr, 1 << 30, (a.aux_nt & 1) != 0);
    if (C::NAUX >= 2) dma(a.aux1, tile, slot + AUX1, 31, nullptr, 1 << 30, (a.aux_nt & 2) != 0);
    if (C::HAS_S)            // this wave's private copy of the 32 per-point scalars (both half-waves fetch the same 128 B)
      dma1(a.srow + (size_t)tile * WSP + (lane & 31), stage + EXTRA + wave * 256, false);
    if (C::HAS_PF)           // octet 0 of the pad-fill array: every wave writes the SAME bytes to the same 1 KiB (both half-waves
                             // fetch the same 512 B), so each wave's own counted wait covers the copy it reads
      dma1(a.padfill + ((size_t)tile * WSP + (lane & 31)) * 8, stage + EXTRA, true);
  };
#pragma unroll
  for (int t = 0; t < C::NS - 1; ++t)
    if (t < T) issue(t);

  // stationary operands: this wave's 32 x 256 slice of W (A fragments) and its per-row constants in accumulator layout
  const bool live = wave * 32 < a.N;
  uint4 wreg[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) wreg[ks] = live ? a.Wp[((size_t)wave * a.kstride + ks) * 64 + lane] : make_uint4(0u, 0u, 0u, 0u);
  float bias[16];             // EPI_RELU / EPI_LINEAR / EPI_SIGMOID: bias ; EPI_BWD8: the sdf row of lin8 (effective weight)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int n = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    float b = 0.0f;
    if ((EPI == EPI_RELU || EPI == EPI_LINEAR || EPI == EPI_SIGMOID) && a.bias && n < a.N) b = a.bias[n];
    if (EPI == EPI_BWD8 && n < a.N) b = a.wrow[n] * a.wrow_scale[0];
    bias[r] = b;
  }
  const float rho = (EPI == EPI_LINACC && a.rho_num && a.rho_den) ? cot_scale_of(a.rho_num) / cot_scale_of(a.rho_den) : 1.0f;
  const unsigned bfrag = (unsigned)((lane >> 5) * WSP + (lane & 31)) * 16;        // + ks * 2 * WSP * 16
  const unsigned efrag = dma_off + (unsigned)(lane & 31) * 16 + (unsigned)(lane >> 5) * 8;   // + q * WSP * 16
  const int Npad = (a.N + 7) & ~7;

  for (int tau = 0; tau < T; ++tau) {
    const int ahead = min(C::NS - 2, T - 1 - tau);
    ws_wait_barrier(ahead * C::G);
    if (tau + C::NS - 1 < T) issue(tau + C::NS - 1);
    const unsigned char* slot = wslds + (tau % C::NS) * C::STAGE;
    // a wave without live output rows (narrow layers: N = 3, 6, 39 ... leave waves 1..7 idle) only moves its share of the DMA and
    // keeps the barriers: its fragment reads and MFMAs would compete with the stream for the LDS port and the issue slots
    if (!live) continue;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const uint4 bv = *reinterpret_cast<const uint4*>(slot + bfrag + ks * (2 * WSP * 16));
      acc = NEAT_MFMA16(*reinterpret_cast<const bf16x8*>(&wreg[ks]), *reinterpret_cast<const bf16x8*>(&bv), acc, 0, 0, 0);
    }
    const int p = (t_begin + tau * a.tile_stride) * WSP + (lane & 31);
    float sp = 0.0f;
    if (C::HAS_S) sp = *reinterpret_cast<const float*>(slot + EXTRA + wave * 256 + (lane & 31) * 4);
    uint2 pk0[4], pk1[4];            // packed quads of out0 / out1 (wide stores: two quads travel as one 16-byte octet per lane)
#pragma unroll
    for (int q = 0; q < 4; ++q) { pk0[q] = make_uint2(0u, 0u); pk1[q] = make_uint2(0u, 0u); }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n0 = wave * 32 + 8 * q + 4 * (lane >> 5);
      if (n0 >= Npad) continue;
      float x0[4] = {0.f, 0.f, 0.f, 0.f}, x1[4] = {0.f, 0.f, 0.f, 0.f};
      if (C::NAUX >= 1) {
        const uint2 r = *reinterpret_cast<const uint2*>(slot + AUX0 + efrag + q * (WSP * 16));
        x0[0] = bf_lo(r.x); x0[1] = bf_hi(r.x); x0[2] = bf_lo(r.y); x0[3] = bf_hi(r.y);
      }
      if (C::NAUX >= 2) {
        const uint2 r = *reinterpret_cast<const uint2*>(slot + AUX1 + efrag + q * (WSP * 16));
        x1[0] = bf_lo(r.x); x1[1] = bf_hi(r.x); x1[2] = bf_lo(r.y); x1[3] = bf_hi(r.y);
      }
      float o0[4], o1[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = acc[4 * q + e];
        float r0 = 0.0f, r1 = 0.0f;
        if (EPI == EPI_RELU) r0 = fmaxf(v + bias[4 * q + e], 0.0f);
        else if (EPI == EPI_LINEAR) r0 = v + bias[4 * q + e];
        else if (EPI == EPI_SIGMOID) r0 = 1.0f / (1.0f + __expf(-(v + bias[4 * q + e])));
        else if (EPI == EPI_LINACC) r0 = v * rho + x0[e];
        else if (EPI == EPI_REV) { r0 = (n0 + e < a.n_split) ? v * dphi_fast(x0[e]) : 0.0f; r1 = v; }
        else if (EPI == EPI_TAN || EPI == EPI_TAN_PF) { const float sg = dphi_fast(x0[e]); r0 = v * sg; r1 = v * x1[e] * (100.0f * (1.0f - sg)); }
        else if (EPI == EPI_BWD) r0 = v * dphi_fast(x0[e]) + x1[e];
        else if (EPI == EPI_BWD8) r0 = (v + bias[4 * q + e] * sp) * dphi_fast(x0[e]) + x1[e];
        else if (EPI == EPI_BWD_RELU) r0 = x0[e] > 0.0f ? v : 0.0f;
        if (n0 + e >= a.N) {                                // padded rows of the last octet: finite zeros, or the pad-fill rows
          r0 = 0.0f; r1 = 0.0f;
          if (C::HAS_PF && n0 + e < a.N + 7)
            r0 = bf2f(*reinterpret_cast<const u16*>(slot + EXTRA + (lane & 31) * 16 + (n0 + e - a.N) * 2));
        }
        o0[e] = r0; o1[e] = r1;
      }
      if (OUTF) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n0 + e < a.N) a.out0f[(unsigned)(n0 + e) * (unsigned)a.ldp + (unsigned)p] = o0[e];
        continue;
      }
      pk0[q] = make_uint2(pack2(o0[0], o0[1]), pack2(o0[2], o0[3]));
      if (EPI == EPI_TAN || EPI == EPI_TAN_PF) pk1[q] = make_uint2(pack2(o1[0], o1[1]), pack2(o1[2], o1[3]));
      if (!a.wide_store) {
        const unsigned oidx = ((unsigned)(n0 >> 3) * (unsigned)a.ldp + (unsigned)p) * 8u + (unsigned)(n0 & 7);
        if (EPI != EPI_REV || n0 < ((a.n_split + 7) & ~7))        // (split layer: out0 ends with the octet that holds row n_split-1)
          *reinterpret_cast<uint2*>(a.out0 + oidx) = pk0[q];
        if (EPI == EPI_TAN || EPI == EPI_TAN_PF) {
          typedef unsigned long long u64_t;
          if (a.aux_nt & 8) __builtin_nontemporal_store(__builtin_bit_cast(u64_t, pk1[q]), reinterpret_cast<u64_t*>(a.out1 + oidx));   // m_l: next read by the weight gradient
          else *reinterpret_cast<uint2*>(a.out1 + oidx) = pk1[q];
        }
      }
      if (EPI == EPI_REV && n0 + 3 >= a.n_split && a.out1f) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n0 + e >= a.n_split && n0 + e < a.N) a.out1f[(unsigned)(n0 + e - a.n_split) * (unsigned)a.ldp + (unsigned)p] = o1[e];
      }
    }
    if (!OUTF && a.wide_store) {
      // 16-byte stores: the accumulator layout gives lane l rows 8q..8q+3 and lane l+32 rows 8q+4..8q+7 of a point; one
      // v_permlane32_swap per dword hands lanes 0-31 the whole octet of quad 2j and lanes 32-63 the whole octet of quad 2j+1
      // (half as many store instructions, 16 B per lane like a plain copy kernel)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int ob = wave * 32 + 16 * j;
        if (ob >= Npad) continue;
        const int oct = (ob >> 3) + (lane >> 5);
        const unsigned oidx = ((unsigned)oct * (unsigned)a.ldp + (unsigned)p) * 8u;
        const bool in_range = oct * 8 < Npad;
        {
          typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
          const v2u_t s0 = __builtin_amdgcn_permlane32_swap(pk0[2 * j].x, pk0[2 * j + 1].x, false, false);
          const v2u_t s1 = __builtin_amdgcn_permlane32_swap(pk0[2 * j].y, pk0[2 * j + 1].y, false, false);
          if (in_range && (EPI != EPI_REV || oct * 8 < ((a.n_split + 7) & ~7))) {
            typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
            const v4u_t v = {s0.x, s1.x, s0.y, s1.y};
            if (a.aux_nt & 16) __builtin_nontemporal_store(v, reinterpret_cast<v4u_t*>(a.out0 + oidx));
            else *reinterpret_cast<v4u_t*>(a.out0 + oidx) = v;
          }
        }
        if (EPI == EPI_TAN || EPI == EPI_TAN_PF) {
          typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
          const v2u_t s0 = __builtin_amdgcn_permlane32_swap(pk1[2 * j].x, pk1[2 * j + 1].x, false, false);
          const v2u_t s1 = __builtin_amdgcn_permlane32_swap(pk1[2 * j].y, pk1[2 * j + 1].y, false, false);
          if (in_range) *reinterpret_cast<uint4*>(a.out1 + oidx) = make_uint4(s0.x, s1.x, s0.y, s1.y);
        }
      }
    }
  }
}

// out[split][k] = sum over the split's points of  s[p] B0[k][p] + B1[k][p]  (k < 256) and out[split][256] = sum s[p]:
// the lin8 row of the weight gradient that belongs to the raw sdf (cotangent s, fp32 per point) -- the feature rows go
// through wgrad_kernel_h3; same point chunks, same partial layout, so the reduction kernels see one more row.
__global__ __launch_bounds__(256) void rowdot_kernel(const float* __restrict__ s, const u16* __restrict__ B0, const u16* __restrict__ B1,
                                                     int P, int ldp, int chunk, float* __restrict__ out, size_t split_stride) {
  __shared__ float red[8][264];
  const int tid = threadIdx.x, oct = tid >> 3, pg = tid & 7;           // 32 octets x 8 point lanes: 8 lanes = one 128-byte line of one octet row
  const int pbeg = blockIdx.x * chunk, pend = min(P, pbeg + chunk);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float ssum = 0.0f;
  const uint4* b0 = reinterpret_cast<const uint4*>(B0) + (size_t)oct * ldp;
  const uint4* b1 = reinterpret_cast<const uint4*>(B1) + (size_t)oct * ldp;
  for (int pb = pbeg + pg; pb < pend; pb += 32) {       // four points per trip: eight independent 16-byte loads in flight
    float sv[4]; uint4 x[4], y[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int p = pb + 8 * u;
      const bool ok = p < pend;
      sv[u] = ok ? s[p] : 0.0f;
      x[u] = ok ? b0[p] : make_uint4(0u, 0u, 0u, 0u);
      y[u] = ok ? b1[p] : make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const unsigned xw[4] = {x[u].x, x[u].y, x[u].z, x[u].w}, yw[4] = {y[u].x, y[u].y, y[u].z, y[u].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[2 * j] += sv[u] * bf_lo(xw[j]) + bf_lo(yw[j]);
        acc[2 * j + 1] += sv[u] * bf_hi(xw[j]) + bf_hi(yw[j]);
      }
      if (oct == 0) ssum += sv[u];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[pg][oct * 8 + e] = acc[e];
  if (oct == 0) red[pg][256] = ssum;
  __syncthreads();
  for (int k = tid; k < 257; k += 256) {
    float v = 0.0f;
#pragma unroll
    for (int g = 0; g < 8; ++g) v += red[g][k];
    out[(size_t)blockIdx.x * split_stride + k] = v;
  }
}

// ---------------------------------------------------------------------------------------------
// Fused SDF primal chain (ImplicitNetwork.forward, rend_a :78-96, + get_sdf_vals' clamp :131-137):
// PE-6 -> lin0..lin8 for a tile of 32*PT points in ONE launch.  The activation tile lives in LDS (octet-major
// bf16) and is updated in place layer after layer; weights stream from L2 through the register ring of mma_rows_h.
//   values mode : only the clamped sdf leaves the chip (the sampler's 128..640 queries per ray: no HBM round trips)
//   save mode   : every post-activation h_l, the PE rows and the lin8 output are also written for the backward pass
// ---------------------------------------------------------------------------------------------

template <int PT, bool VALUES>
__global__ __launch_bounds__(WG, 2) void sdf_fused_kernel_h(FusedArgs a) {
  constexpr int BMT = 32 * PT;
  if (a.gate && *a.gate != a.gate_value) return;
  extern __shared__ __attribute__((aligned(16))) uint4 ldsq[];        // tile [32][BMT] | PE octets [8][BMT] (K padded to 64)
  uint4* tile = ldsq;
  uint4* pe = ldsq + 32 * BMT;
  u16* tile16 = reinterpret_cast<u16*>(tile);
  u16* pe16 = reinterpret_cast<u16*>(pe);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p0 = blockIdx.x * BMT;
  // ---- positional encoding (embedder.py:12-36) into the PE octets (rows 39..63 zero)
  for (int idx = tid; idx < 64 * BMT; idx += WG) {
    const int j = idx / BMT, p = idx % BMT;
    float v = 0.0f;
    if (j < 39) {
      const int c = j < 3 ? j : (j - 3) % 3;
      const float xc = a.x_fm[(size_t)c * a.ldp + p0 + p];
      if (j < 3) v = xc;
      else {
        const int k = (j - 3) / 6, is_cos = ((j - 3) % 6) >= 3;
        const float f = (float)(1 << k);
        v = is_cos ? __cosf(xc * f) : __sinf(xc * f);      // hardware sin/cos: |arg| <= 96, abs error ~1e-6 << bf16 resolution
      }
      if (!VALUES && a.save) a.E[(size_t)j * a.ldp + p0 + p] = v;
    }
    pe16[((size_t)(j >> 3) * BMT + p) * 8 + (j & 7)] = f2bf(v);
  }
  __syncthreads();
  f32x16 acc[2][PT];
  const int hi = lane >> 5;
  // kernarg tables are read through compile-time indices only (a runtime-indexed by-value array would be copied to scratch)
#define FUSED_SEL(T, field, l, out)                                                                   \
  T out = a.field[0];                                                                                \
  _Pragma("unroll") for (int k_ = 1; k_ < 9; ++k_) if ((l) == k_) out = a.field[k_];
  // ---- hidden layers lin0..lin7
#pragma unroll 1
  for (int l = 0; l < 8; ++l) {
    const int N = (l == 3) ? 217 : 256;
    const int NT = (N + 31) >> 5;
    FUSED_SEL(int, KS, l, KS)
    FUSED_SEL(const uint4*, Wp, l, Wl)
    FUSED_SEL(const float*, bias, l, bias)
    FUSED_SEL(u16*, h, l + 1, hsel)
    const uint4* bl = (l == 0 ? pe : tile) + (size_t)hi * BMT + (lane & 31);
    const int t0 = wave;
    const int ntw = (t0 + 4 < NT) ? 2 : 1;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < PT; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][q][r] = 0.0f;
    const uint4* wp0 = Wl + (size_t)t0 * KS * 64 + lane;
    if (ntw == 2) mma_rows_h<2, PT>(acc, wp0, 4 * KS * 64, bl, 0, KS);
    else mma_rows_h<1, PT>(acc, wp0, 4 * KS * 64, bl, 0, KS);
    __syncthreads();                                               // everyone has read the tile: update it in place
    u16* hout = (!VALUES && a.save) ? hsel : nullptr;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (i >= ntw) continue;
      const int nt = t0 + 4 * i;
#pragma unroll
      for (int q = 0; q < PT; ++q) {
        const int p = q * 32 + (lane & 31);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n0 = nt * 32 + 8 * g + 4 * hi;
          if (n0 >= N) continue;
          float o[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (n0 + e < N) ? softplus100_fast(acc[i][q][4 * g + e] + bias[n0 + e]) : 0.0f;
          const size_t li = ((size_t)(n0 >> 3) * BMT + p) * 8 + (n0 & 7);
          if (n0 + 3 < N) {
            const uint2 pk = make_uint2(pack2(o[0], o[1]), pack2(o[2], o[3]));
            *reinterpret_cast<uint2*>(tile16 + li) = pk;
            if (hout && l != 3) *reinterpret_cast<uint2*>(hout + ((size_t)(n0 >> 3) * a.ldp + p0 + p) * 8 + (n0 & 7)) = pk;
          } else {                                                   // lin3: the quad holding row 216 (rows 217.. belong to the PE copy)
#pragma unroll
            for (int e = 0; e < 4; ++e) if (n0 + e < N) tile16[li + e] = f2bf(o[e]);
          }
        }
      }
    }
    if (l == 3) {
      __syncthreads();
      // skip connection (rend_a :87-88): rows 217..255 of lin4's input are the 39 PE rows (the 1/sqrt2 is folded into W4)
      for (int idx = tid; idx < 39 * BMT; idx += WG) {
        const int j = idx / BMT, p = idx % BMT, row = 217 + j;
        tile16[((size_t)(row >> 3) * BMT + p) * 8 + (row & 7)] = pe16[((size_t)(j >> 3) * BMT + p) * 8 + (j & 7)];
      }
      if (hout) {                                                    // h4 as the unfused consumers expect it: 217 rows + PE[0..6] in the pad
        __syncthreads();
        for (int idx = tid; idx < 28 * BMT; idx += WG) {
          const int o8 = idx / BMT, p = idx % BMT;
          reinterpret_cast<uint4*>(hout)[(size_t)o8 * a.ldp + p0 + p] = tile[(size_t)o8 * BMT + p];
        }
      }
    }
    __syncthreads();
  }
#undef FUSED_SEL
  // ---- lin8
  const uint4* bl = tile + (size_t)hi * BMT + (lane & 31);
  const int KS = a.KS[8];
  if (!VALUES) {
#pragma unroll 1
    for (int round = 0; round < 2; ++round) {                      // tiles 0..7 = features, tile 8 = [sdf, 31 x padding]
      const int t0 = round * 8 + wave;
      const int ntw = (round == 0) ? 2 : (t0 < 9 ? 1 : 0);
      if (ntw == 0) continue;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < PT; ++q)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][q][r] = 0.0f;
      const uint4* wp0 = a.Wp[8] + (size_t)t0 * KS * 64 + lane;
      if (ntw == 2) mma_rows_h<2, PT>(acc, wp0, 4 * KS * 64, bl, 0, KS);
      else mma_rows_h<1, PT>(acc, wp0, 4 * KS * 64, bl, 0, KS);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (i >= ntw) continue;
        const int nt = t0 + 4 * i;
#pragma unroll
        for (int q = 0; q < PT; ++q) {
          const int p = p0 + q * 32 + (lane & 31);
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int n0 = nt * 32 + 8 * g + 4 * hi;
            if (n0 >= 257) continue;
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              int bi = n0 + e + a.bias8_rot; if (bi >= a.bias8_n) bi -= a.bias8_n;
              o[e] = (n0 + e < 257) ? acc[i][q][4 * g + e] + a.bias[8][bi] : 0.0f;
            }
            if (n0 < 256) *reinterpret_cast<uint2*>(a.feat + ((size_t)(n0 >> 3) * a.ldp + p) * 8 + (n0 & 7)) = make_uint2(pack2(o[0], o[1]), pack2(o[2], o[3]));
            else a.sdfraw[p] = o[0];
          }
        }
      }
    }
  } else {
    // values mode: lin8 restricted to the sdf row (pack with one 32-row tile): split K over the 4 waves
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < PT; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][q][r] = 0.0f;
    const int per = ((KS / 4 + 3) / 4) * 4;
    const int sb = wave * per, se = min(KS, sb + per);
    if (sb < se) mma_rows_h<1, PT>(acc, a.Wp[8] + lane, 0, bl, sb, se);
    __syncthreads();
    float* red = reinterpret_cast<float*>(ldsq);                    // [4 waves][PT][64]: only accumulator register 0 of lanes 0..31 matters
    if (hi == 0) {
#pragma unroll
      for (int q = 0; q < PT; ++q) red[(wave * PT + q) * 32 + lane] = acc[0][q][0];
    }
    __syncthreads();
    if (tid < BMT) {
      const int q = tid >> 5, ln = tid & 31;
      float s = a.bias[8][0];
#pragma unroll
      for (int w = 0; w < 4; ++w) s += red[(w * PT + q) * 32 + ln];
      const int p = p0 + tid;
      if (a.radius > 0.0f) {
        const float x0 = a.x_fm[p], x1 = a.x_fm[(size_t)a.ldp + p], x2 = a.x_fm[(size_t)2 * a.ldp + p];
        s = fminf(s, a.scale * (a.radius - sqrtf(x0 * x0 + x1 * x1 + x2 * x2)));
      }
      if (p < a.P) a.sdf_out[p] = s;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// sdf_fused_ws_kernel: the fused primal chain, weight-stationary.  sdf_fused_kernel_h streams every layer's weights
// from L2 for each 64-point workgroup (9 x 128 KiB per 64 points, latency bound, ~200 TF/s); here one persistent
// 8-wave workgroup per CU walks batches of 32*NT points layer by layer:
//  * wave w owns output rows 32w..32w+31 of every hidden layer and holds that 32 x 256 slice of W_l as 16 A fragments
//    in VGPRs; the slice of layer l+1 is prefetched into a second register set while layer l computes (lin0's small
//    slice and the sdf-row fragments of lin8 stay resident for the whole launch);
//  * activations ping-pong between two LDS buffers [32 octets][points][16 B]: one barrier per layer, no in-place
//    hazards; biases of all layers sit in LDS;
//  * save mode streams h_1..h_8, PE and the lin8 outputs to HBM with 8-byte stores straight from the accumulators;
//    values mode (sampler) writes nothing but the clamped sdf.
// ---------------------------------------------------------------------------------------------
constexpr int FWT = 512;
template <int NT> struct FwsCfg {
  static constexpr int BP = 32 * NT;
  static constexpr int XBYTES = 32 * BP * 16;            // one activation buffer
  static constexpr int PEBYTES = 8 * BP * 16;            // PE octets (K padded to 64); reused for the sdf partial sums
  static constexpr int BIAS_FLOATS = 9 * 256 + 8;
  static constexpr int LDS = 2 * XBYTES + PEBYTES + BIAS_FLOATS * 4;
};

template <int NT, bool VALUES>
__global__ __launch_bounds__(FWT, 2) void sdf_fused_ws_kernel(FusedArgs a, int ntiles, int nwg) {
  typedef FwsCfg<NT> C;
  constexpr int BP = C::BP;
  if (a.gate && *a.gate != a.gate_value) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char fws[];
  unsigned char* XA = fws;
  unsigned char* XB = fws + C::XBYTES;
  unsigned char* PE = fws + 2 * C::XBYTES;
  float* biasl = reinterpret_cast<float*>(fws + 2 * C::XBYTES + C::PEBYTES);     // [l][256]; lin8 in packed row order
  u16* pe16 = reinterpret_cast<u16*>(PE);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5;
  const bool save = !VALUES && a.save;

  // biases -> LDS (lin3 has 217 rows; lin8: packed row n <- bias[(n + rot) mod 257], the sdf row's bias at [8][256])
  for (int idx = tid; idx < 8 * 256; idx += FWT) {
    const int l = idx >> 8, n = idx & 255;
    float v = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) if (l == k && n < (k == 3 ? 217 : 256)) v = a.bias[k][n];
    biasl[idx] = v * SOFTPLUS_C;             // hidden layers: pre-scaled for softplus100_pk
  }
  for (int n = tid; n < 257; n += FWT) {
    int bi = n + a.bias8_rot; if (bi >= a.bias8_n) bi -= a.bias8_n;
    biasl[8 * 256 + n] = VALUES ? (n == 0 ? a.bias[8][0] : 0.0f) : a.bias[8][bi];
  }
  // ONE register set for the weight slice: all NT point tiles of a layer are multiplied first (NT independent accumulator
  // chains: a single chain of 16 dependent MFMAs runs at the MFMA latency and exposes every LDS read), then the slice of
  // the NEXT layer is loaded into the same registers and its L2 latency hides under this layer's epilogue.
  uint4 ws8[2], wreg[16];
  {
    unsigned voff = (unsigned)lane * 16u;
    asm volatile("" : "+v"(voff));
#pragma unroll
    for (int j = 0; j < 2; ++j)          // this wave's two k-steps of the sdf row of lin8 stay resident
      ws8[j] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(a.Wp[8]) + voff + (((VALUES ? 0 : 8) * 16 + 2 * wave + j) * 1024));
  }
  auto load_w = [&](const uint4* Wl, int KS, bool on) {
    // The per-lane offset is laundered through an empty asm: otherwise the fragment addresses of all layers are
    // loop-invariant, get hoisted out of the batch loop and spilled (2 VGPRs each).  SGPR base + this offset + immediate.
    unsigned voff = (unsigned)(wave * KS * 64 + lane) * 16u;
    asm volatile("" : "+v"(voff));
    const char* base = reinterpret_cast<const char*>(Wl);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
      if (ks < KS) wreg[ks] = on ? *reinterpret_cast<const uint4*>(base + voff + ks * 1024) : make_uint4(0u, 0u, 0u, 0u);
  };
  load_w(a.Wp[0], 4, true);

  // the 32-point tiles are split evenly over the workgroups (tile counts differ by at most one); each workgroup walks its
  // range in batches of NT tiles, the last batch may be shorter (its missing tiles skip MFMAs, epilogue and stores)
  // nwg < 0: batches of NT tiles interleaved over the -nwg workgroups (batch b -> workgroup b mod grid) instead of one
  // contiguous tile range per workgroup
  const bool inter = nwg < 0;
  const int ng = inter ? -nwg : nwg;
  const int t_begin = inter ? (int)blockIdx.x * NT : (int)(((long long)blockIdx.x * ntiles) / ng);
  const int t_end = inter ? ntiles : (int)(((long long)(blockIdx.x + 1) * ntiles) / ng);
  const int t_step = inter ? ng * NT : NT;
  for (int tile0 = t_begin; tile0 < t_end; tile0 += t_step) {
    const int p0 = tile0 * 32;
    const int nt = min(NT, t_end - tile0);
    // ---- positional encoding (embedder.py:12-36) into the PE octets (rows 39..63 zero)
    for (int idx = tid; idx < 64 * BP; idx += FWT) {
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
      pe16[((size_t)(j >> 3) * BP + p) * 8 + (j & 7)] = f2bf(v);
    }
    __syncthreads();

    // one hidden layer: dst[rows of this wave] = softplus(W src + b), optionally streamed to HBM; then `next` is loaded.
    // Software pipeline over the point tiles: the 16 MFMAs of tile t are interleaved, k-step by k-step, with the epilogue
    // of tile t-1 (bias, softplus, pack, LDS + HBM stores: ~190 VALU instructions per tile against 16 x 8 MFMA passes), so
    // the matrix core and the vector ALU of the SIMD work at the same time instead of one after the other; only two
    // accumulator tiles are live.  The B fragment of the next k-step is read from LDS one step ahead.
    auto hidden = [&](int KS, const unsigned char* src, unsigned char* dst, int l, int N, u16* hout, const uint4* next, int nextKS, bool next_on) {
      const float* bl = biasl + l * 256;
      const unsigned char* bp = src + ((size_t)hi * BP + (lane & 31)) * 16;
      const bool rows_live = wave * 32 < N;
      f32x16 acc[2];
      auto epi_quad = [&](const f32x16& ac, int t, int g) {
        const int n0 = wave * 32 + 8 * g + 4 * hi;
        if (!rows_live || n0 >= N) return;
        const int pl = t * 32 + (lane & 31);
        const float4 bb = *reinterpret_cast<const float4*>(bl + n0);
        const v2f_t o01 = softplus100_pk(v2f_t{ac[4 * g], ac[4 * g + 1]}, v2f_t{bb.x, bb.y});
        const v2f_t o23 = softplus100_pk(v2f_t{ac[4 * g + 2], ac[4 * g + 3]}, v2f_t{bb.z, bb.w});
        float o[4] = {o01.x, o01.y, o23.x, o23.y};
#pragma unroll
        for (int e = 0; e < 4; ++e) if (n0 + e >= N) o[e] = 0.0f;
        u16* lp = reinterpret_cast<u16*>(dst) + ((n0 >> 3) * BP + pl) * 8 + (n0 & 7);
        if (n0 + 3 < N) {
          const uint2 pk = make_uint2(pack2(o[0], o[1]), pack2(o[2], o[3]));
          *reinterpret_cast<uint2*>(lp) = pk;
          if (hout) {                                    // 32-bit byte offset (arrays < 4 GiB) off the layer's base pointer
            const unsigned go = ((unsigned)(n0 >> 3) * (unsigned)a.ldp + (unsigned)(p0 + pl)) * 16u + (unsigned)(n0 & 7) * 2u;
            *reinterpret_cast<uint2*>(reinterpret_cast<char*>(hout) + go) = pk;
          }
        } else {                                         // lin3: the quad holding row 216 (rows 217.. receive the PE copy)
#pragma unroll
          for (int e = 0; e < 4; ++e) if (n0 + e < N) lp[e] = f2bf(o[e]);
        }
      };
#pragma unroll
      for (int t = 0; t <= NT; ++t) {
        const bool mma_on = t < NT && t < nt, epi_on = t >= 1 && t - 1 < nt;
        if (t == NT) {                                   // every MFMA of this layer has been issued: the next slice may land
          __builtin_amdgcn_sched_barrier(0);
          if (next) load_w(next, nextKS, next_on);
          __builtin_amdgcn_sched_barrier(0);
        }
        f32x16& am = acc[t & 1];
        const f32x16& ae = acc[(t + 1) & 1];
        if (mma_on) {
#pragma unroll
          for (int r = 0; r < 16; ++r) am[r] = 0.0f;
        }
        uint4 bv = make_uint4(0u, 0u, 0u, 0u);
        if (mma_on) bv = *reinterpret_cast<const uint4*>(bp + t * 32 * 16);
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
          if (mma_on && ks < KS) {
            const uint4 cur = bv;
            if (ks + 1 < KS) bv = *reinterpret_cast<const uint4*>(bp + (size_t)(ks + 1) * 2 * BP * 16 + t * 32 * 16);
            am = NEAT_MFMA16(*reinterpret_cast<const bf16x8*>(&wreg[ks]), *reinterpret_cast<const bf16x8*>(&cur), am, 0, 0, 0);
          }
          if (epi_on && (ks & 3) == 3) epi_quad(ae, t - 1, ks >> 2);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __syncthreads();
    };

    hidden(4, PE, XA, 0, 256, save ? a.h[1] : nullptr, a.Wp[1], 16, true);
    hidden(16, XA, XB, 1, 256, save ? a.h[2] : nullptr, a.Wp[2], 16, true);
    hidden(16, XB, XA, 2, 256, save ? a.h[3] : nullptr, a.Wp[3], 16, wave * 32 < 217);
    hidden(16, XA, XB, 3, 217, nullptr, a.Wp[4], 16, true);
    // skip connection (rend_a :87-88): rows 217..255 of lin4's input are the 39 PE rows (1/sqrt2 folded into W4)
    {
      u16* xb16 = reinterpret_cast<u16*>(XB);
      for (int idx = tid; idx < 39 * BP; idx += FWT) {
        const int j = idx / BP, p = idx % BP, row = 217 + j;
        xb16[((size_t)(row >> 3) * BP + p) * 8 + (row & 7)] = pe16[((size_t)(j >> 3) * BP + p) * 8 + (j & 7)];
      }
      __syncthreads();
      if (save) {                                          // h4 as the unfused consumers expect it: 217 rows + PE[0..6] in the pad
        for (int idx = tid; idx < 28 * nt * 32; idx += FWT) {
          const int o8 = idx / (nt * 32), p = idx % (nt * 32);
          reinterpret_cast<uint4*>(a.h[4])[(size_t)o8 * a.ldp + p0 + p] = reinterpret_cast<const uint4*>(XB)[(size_t)o8 * BP + p];
        }
      }
    }
    hidden(16, XB, XA, 4, 256, save ? a.h[5] : nullptr, a.Wp[5], 16, true);
    hidden(16, XA, XB, 5, 256, save ? a.h[6] : nullptr, a.Wp[6], 16, true);
    hidden(16, XB, XA, 6, 256, save ? a.h[7] : nullptr, a.Wp[7], 16, true);
    hidden(16, XA, XB, 7, 256, save ? a.h[8] : nullptr, VALUES ? a.Wp[0] : a.Wp[8], VALUES ? 4 : 16, true);

    // ---- lin8: 256 feature rows (save mode) + the sdf row, split over the waves' k-steps and reduced through LDS
    float* red = reinterpret_cast<float*>(PE);             // [8 waves][BP]
    {
      const unsigned char* bp = XB + ((size_t)hi * BP + (lane & 31)) * 16;
      f32x16 accs[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) accs[t][r] = 0.0f;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          if (t >= nt) break;
          const uint4 bv = *reinterpret_cast<const uint4*>(bp + (size_t)(2 * wave + j) * 2 * BP * 16 + t * 32 * 16);
          accs[t] = NEAT_MFMA16(*reinterpret_cast<const bf16x8*>(&ws8[j]), *reinterpret_cast<const bf16x8*>(&bv), accs[t], 0, 0, 0);
        }
      if (hi == 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t) red[wave * BP + t * 32 + lane] = accs[t][0];
      }
      if (!VALUES) {
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            if (t >= nt) break;
            const uint4 bv = *reinterpret_cast<const uint4*>(bp + (size_t)ks * 2 * BP * 16 + t * 32 * 16);
            acc[t] = NEAT_MFMA16(*reinterpret_cast<const bf16x8*>(&wreg[ks]), *reinterpret_cast<const bf16x8*>(&bv), acc[t], 0, 0, 0);
          }
          if ((ks & 1) == 1) __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (tile0 + t_step < t_end) load_w(a.Wp[0], 4, true);  // next batch's lin0 slice
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          if (t >= nt) break;
          const int pl = t * 32 + (lane & 31);
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int n0 = wave * 32 + 8 * g + 4 * hi;
            const float4 bb = *reinterpret_cast<const float4*>(biasl + 8 * 256 + n0);
            const unsigned go = ((unsigned)(n0 >> 3) * (unsigned)a.ldp + (unsigned)(p0 + pl)) * 16u + (unsigned)(n0 & 7) * 2u;
            *reinterpret_cast<uint2*>(reinterpret_cast<char*>(a.feat) + go) =
                make_uint2(pack2(acc[t][4 * g] + bb.x, acc[t][4 * g + 1] + bb.y), pack2(acc[t][4 * g + 2] + bb.z, acc[t][4 * g + 3] + bb.w));
          }
        }
      }
    }
    __syncthreads();
    if (tid < nt * 32) {
      float sv = biasl[8 * 256 + (VALUES ? 0 : 256)];
#pragma unroll
      for (int w = 0; w < 8; ++w) sv += red[w * BP + tid];
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

// ---------------------------------------------------------------------------------------------
// bf16 weight gradient: ONE workgroup (8 waves) owns a full 256x256 output tile, so every
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
          for (int j = 0; j < 4; ++j) t += bf_lo(w4[j]) + bf_hi(w4[j]);
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
              acc[i][j] = NEAT_MFMA16(*reinterpret_cast<bf16x8*>(&av[i]), *reinterpret_cast<bf16x8*>(&bv[j]), acc[i][j], 0, 0, 0);
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

// fp32 feature-major rows -> bf16 octet-major copies (zero-padded to whole octets) of the few small operands (PE rows,
// head inputs, narrow cotangents), so that the streaming kernels see nothing but octet-major bf16
struct OctPackJob { const float* src; int rows; u16* dst; };
struct OctPackArgs { OctPackJob job[4]; int njobs, ldp; };
__global__ void oct_pack_kernel(OctPackArgs a) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= a.ldp) return;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (j >= a.njobs) break;
    const OctPackJob jb = a.job[j];
    for (int o = 0; o * 8 < jb.rows; ++o) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (o * 8 + e < jb.rows) ? jb.src[(size_t)(o * 8 + e) * a.ldp + p] : 0.0f;
      reinterpret_cast<uint4*>(jb.dst)[(size_t)o * a.ldp + p] = make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
    }
  }
}

// Prologue of the SDF backward pass of the 16-bit builds in ONE launch per point (round 4: a launch costs ~5 us inside a replayed
// graph whatever it does): the cotangent of the normals (normal_cotangent_kernel), its PE tangent E^ = J g^ (posenc6_tangent_kernel)
// and the octet-major 16-bit copies of the PE rows and their tangents that the streaming kernels read (oct_pack_kernel: rows 0..38
// and rows 7..38 of each) -- same arithmetic, three launches less.
struct BwdPrologueArgs {
  const float* sc_r; const float* sc_a; const float* extra_rm; const float* mask; const float* d_tail_rm;
  const float* cot_slot; const float* cot_slot_a;
  int P, ldp, P_main;
  const float* x_fm; const float* E;          // points, their PE rows [39][ldp] (saved by the forward pass)
  float* gh; float* Eh;                       // fp32 feature-major outputs [3][ldp], [39][ldp]
  u16* Ebf; u16* Ebf4; u16* Ehbf; u16* Ehbf4; // octet-major copies: rows 0..38 (5 octets) / rows 7..38 (4 octets)
  // round 6: one more workgroup (the last) sums the per-ray partials of d loss / d beta (beta_grad_body; null: not wanted) -- they were
  // written by composite_bwd_kernel, several launches earlier; one launch less per step
  const float* dbeta_ray; int R; const float* beta_ptr; float* dbeta;
};
__global__ __launch_bounds__(256) void bwd_prologue_kernel(BwdPrologueArgs a) {
  if (a.dbeta && blockIdx.x == gridDim.x - 1) { beta_grad_body(a.dbeta_ray, a.R, a.beta_ptr, a.dbeta); return; }
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= a.ldp) return;
  const size_t ldp = (size_t)a.ldp;
  const float ext_scale = cot_scale_of(a.cot_slot);
  const float rho_a = a.cot_slot_a ? ext_scale / cot_scale_of(a.cot_slot_a) : 1.0f;
  const float keep = (p < a.P) ? 1.0f - (a.mask ? a.mask[p] : 0.0f) : 0.0f;
  float eh[40], ev[40];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float v = 0.0f;
    if (p < a.P_main) {
      if (a.sc_r) v += a.sc_r[(size_t)(30 + c) * ldp + p];
      if (a.sc_a) v += a.sc_a[(size_t)(6 + c) * ldp + p] * rho_a;
      if (a.extra_rm) v += a.extra_rm[p * 3 + c] * ext_scale;
    } else if (p < a.P && a.d_tail_rm) {
      v = a.d_tail_rm[(p - a.P_main) * 3 + c] * ext_scale;
    }
    const float gc = v * keep;
    a.gh[(size_t)c * ldp + p] = gc;
    const float xc = a.x_fm[(size_t)c * ldp + p];
    eh[c] = gc;
    float f = 1.0f;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      eh[3 + 6 * k + c] = f * __cosf(xc * f) * gc;
      eh[6 + 6 * k + c] = -f * __sinf(xc * f) * gc;
      f *= 2.0f;
    }
  }
  eh[39] = 0.0f; ev[39] = 0.0f;
#pragma unroll
  for (int j = 0; j < 39; ++j) { a.Eh[(size_t)j * ldp + p] = eh[j]; ev[j] = a.E[(size_t)j * ldp + p]; }
  auto octet = [](const float (&v)[40], int r0) {      // rows r0 .. r0 + 7 (rows >= 39: zero)
    float t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = (r0 + e < 39) ? v[r0 + e] : 0.0f;
    return make_uint4(pack2(t[0], t[1]), pack2(t[2], t[3]), pack2(t[4], t[5]), pack2(t[6], t[7]));
  };
#pragma unroll
  for (int o = 0; o < 5; ++o) {
    reinterpret_cast<uint4*>(a.Ebf)[(size_t)o * ldp + p] = octet(ev, 8 * o);
    reinterpret_cast<uint4*>(a.Ehbf)[(size_t)o * ldp + p] = octet(eh, 8 * o);
  }
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    reinterpret_cast<uint4*>(a.Ebf4)[(size_t)o * ldp + p] = octet(ev, 7 + 8 * o);
    reinterpret_cast<uint4*>(a.Ehbf4)[(size_t)o * ldp + p] = octet(eh, 7 + 8 * o);
  }
}

// ---------------------------------------------------------------------------------------------
// wgrad_kernel_h3: the all-bf16 256x256 case (hidden layers: both operands octet-major, K = 256, N <= 256).
// No register staging, no second LDS image, no v_perm transposes:
//  * LDS-DMA (`global_load_lds`, 16 B/lane) copies 32-point stages straight from HBM into a 4-deep ring; the per-lane
//    SOURCE address interleaves four feature octets, so the ring holds [32-feature quad][point][32 features] (64 B per
//    point and quad) -- the image `ds_read_b64_tr_b16` needs;
//  * `ds_read_b64_tr_b16` (gfx950 transpose read) delivers, per lane, 4 consecutive points of one feature = half an
//    MFMA A/B fragment of v_mfma_f32_32x32x16_bf16 (reduction index = points).  A 32-lane group touches 256 contiguous
//    bytes: conflict free;
//  * three stages (96 KB per CU) stay in flight across raw `s_barrier`s with counted `s_waitcnt vmcnt(N)`
//    (`__syncthreads()` would drain the DMA queue); nothing else in the loop uses the vector-memory counter.
// Bias gradient = row sums of pair 0's A, taken from the A fragments on the VALU by the waves of column half 0.
// ---------------------------------------------------------------------------------------------
// NEAT_W3_ABLATE (probe builds only; results are WRONG): 1 = no MFMAs, 2 = no LDS fragment reads, 3 = no partial-tile stores, 4 = no DMA
#ifndef NEAT_W3_ABLATE
#define NEAT_W3_ABLATE 0
#endif
constexpr int W3T = 512, W3P = 32, W3NS = 4;
// NCB = 32-column blocks of the output per wave: 4 (K <= 256 packed columns), 5 (K <= 320: the heads' input layers [256 feature | <= 64
// small rows] in ONE launch instead of two that each read the whole A operand; round 4) or 1 (K <= 64: lin0, whose B operand is the 39 PE
// rows -- the 4-block variant moves 32 octets of B per stage whatever K is)
template <int NCB> struct W3Cfg {
  static constexpr int QA = 8, QB = 2 * NCB;           // 32-row quads of the A / B operand in a stage
  static constexpr int STAGE = (QA + QB) * 2048;       // bytes per stage: A quads, then B quads, each 32 points x 64 B
  static constexpr int LDS = W3NS * STAGE;
  static constexpr int QPW = NCB == 5 ? 3 : 2;         // quads a wave moves per stage (NCB = 5: 18 quads over 8 waves x 3, NCB = 1: 10 over 8 x 2; the surplus re-reads the last)
  static constexpr int G = 2 * QPW;                    // LDS-DMA instructions per stage and wave
};
constexpr int W3_STAGE = W3Cfg<4>::STAGE;
constexpr int W3_LDS_BYTES = W3Cfg<4>::LDS;
constexpr int W3_SLOTS = 12;         // up to 6 problems x 2 pairs per launch
struct WgradArgsH3 {
  const unsigned short* A[W3_SLOTS]; const unsigned short* B[W3_SLOTS]; int rowsA[W3_SLOTS];   // operand set of (problem p, pair q) at p * npairs + q
  const unsigned short* B2[W3_SLOTS];  // octets >= splitB of the B operand come from B2 (skip layer: [h4 | PE]); 32 = none
  int splitB, octsB;                 // octets of B in total (K = packed columns <= 256)
  int npairs, N, K, P, ldp, chunk;
  float* partial; size_t row_stride, split_stride; int col_off, bias_col;   // partial column of B column 0 / of the bias (-1: none)
  int nt_loads;                      // operand fetches non-temporal
  int interleave;                    // stages of a workgroup: 0 = one contiguous chunk of points, 1 = every gridDim.y-th 32-point stage
  int nprob; size_t prob_stride;     // nprob > 1 (grid.x = nprob): INDEPENDENT problems of npairs pairs each, problem p = blockIdx.x, its partials
                                     // at partial + p * prob_stride (same-shaped layers in one launch: 1/nprob of the partial tiles per layer)
};
typedef short v4s16 __attribute__((ext_vector_type(4)));

template <int NCB>
__global__ __launch_bounds__(W3T, 2) void wgrad_kernel_h3(WgradArgsH3 a) {
  typedef W3Cfg<NCB> C;
  extern __shared__ __attribute__((aligned(16))) unsigned char w3lds[];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef __attribute__((address_space(3))) v4s16* lds_v4;
  typedef const __attribute__((address_space(1))) void* gbl_ptr;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = (wave >> 1) * 64, wc = (wave & 1) * 32 * NCB;
  // point stages of this workgroup: one contiguous chunk, or (interleave) stages y, y + splits, ... -- at any moment the
  // machine then sweeps one contiguous window of every operand array
  const int nsplit = gridDim.y;
  const int pstep = a.interleave ? nsplit * W3P : W3P;
  const int pbeg = a.interleave ? blockIdx.y * W3P : blockIdx.y * a.chunk;
  const int pend = a.interleave ? a.P : min(a.P, pbeg + a.chunk);
  const int nsteps = a.interleave ? ((a.P + W3P - 1) / W3P - (int)blockIdx.y + nsplit - 1) / nsplit : (pend - pbeg + W3P - 1) / W3P;
  const int prob = a.nprob > 1 ? (int)blockIdx.x : 0;
  const int T = nsteps * a.npairs;
  bool liveR[2], liveC[NCB];
#pragma unroll
  for (int i = 0; i < 2; ++i) liveR[i] = wr + 32 * i < a.N;
#pragma unroll
  for (int j = 0; j < NCB; ++j) liveC[j] = wc + 32 * j < a.K;
  f32x16 acc[2][NCB];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NCB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  float rsum[2] = {0.0f, 0.0f};

  // DMA role of this wave: QPW of the stage's QA + QB quads (global quad g: A quads first), two 16-point halves each
  const int dl_oct = lane & 3, dl_pt = lane >> 2;
  // The DMA is issued from inline asm so that hipcc's waitcnt pass does not know about it: otherwise it puts
  // `s_waitcnt vmcnt(0)` in front of every ds_read that follows an LDS-DMA and the ring never holds more than one stage.
  const unsigned lds_base = (unsigned)(size_t)(lds_ptr)w3lds;
  // operand set of (this problem, pair q) per quad of this wave, selected ONCE with compile-time kernarg indices (a runtime
  // index would go through scratch; selecting inside the streaming loop costs ~150 scalar instructions per stage)
  const unsigned short* opbase[2][C::QPW]; const unsigned short* opbase2[2]; int opmax[2][C::QPW]; int opq[C::QPW]; bool opB[C::QPW];
#pragma unroll
  for (int hq = 0; hq < C::QPW; ++hq) {
    const int g = min(C::QPW * wave + hq, C::QA + C::QB - 1);
    opB[hq] = g >= C::QA; opq[hq] = opB[hq] ? g - C::QA : g;
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int idx = prob * a.npairs + q;
    const unsigned short* pA = a.A[0]; const unsigned short* pB = a.B[0]; const unsigned short* p2 = a.B2[0];
    int rA = a.rowsA[0];
#pragma unroll
    for (int k = 1; k < W3_SLOTS; ++k)
      if (idx == k) { pA = a.A[k]; pB = a.B[k]; p2 = a.B2[k]; rA = a.rowsA[k]; }
    opbase2[q] = p2;
#pragma unroll
    for (int hq = 0; hq < C::QPW; ++hq) { opbase[q][hq] = opB[hq] ? pB : pA; opmax[q][hq] = opB[hq] ? a.octsB - 1 : (rA + 7) / 8 - 1; }
  }
  auto issue = [&](int tau) {
    const int q = tau >= nsteps ? 1 : 0;
    const int st = tau - q * nsteps;
    if (NEAT_W3_ABLATE == 4) return;
    const unsigned short* base2 = q ? opbase2[1] : opbase2[0];
    const unsigned stage = lds_base + (tau % W3NS) * C::STAGE;
#pragma unroll
    for (int hq = 0; hq < C::QPW; ++hq) {
      const unsigned short* base = q ? opbase[1][hq] : opbase[0][hq];
      const int maxoct = q ? opmax[1][hq] : opmax[0][hq];
      const int oct = min(4 * opq[hq] + dl_oct, maxoct);       // rows past the array: re-read a valid octet (dropped later)
      const bool second = opB[hq] && oct >= a.splitB;
      const uint4* seg = reinterpret_cast<const uint4*>(second ? base2 : base) + (size_t)(second ? oct - a.splitB : oct) * a.ldp;
      const int g = min(C::QPW * wave + hq, C::QA + C::QB - 1);
#ifndef NEAT_W3_SKIP_SURPLUS
#define NEAT_W3_SKIP_SURPLUS 1      // round 6: the waves whose DMA slots lie beyond the stage's quads (NCB = 5: waves 6, 7; NCB = 1: waves 5..7) issue
#endif                              // nothing instead of re-reading the last quad -- their counted waits are then trivially met.  Same-box A/B, three
                                    // passes: <5> 49.7-50.8 -> 47.9-48.5 us, <1> 40.0-40.3 -> 38.2-38.7 us.  (Compile-time for NCB = 4, which has no
                                    // surplus: a runtime test there cost its launch 2-3 us.)
      if constexpr (NEAT_W3_SKIP_SURPLUS && 8 * C::QPW > C::QA + C::QB) { if (C::QPW * wave + hq >= C::QA + C::QB) continue; }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int p = pbeg + st * pstep + 16 * i + dl_pt;
        const uint4* src = seg + p;
        const unsigned dst = __builtin_amdgcn_readfirstlane(stage + g * 2048 + i * 1024);
        unsigned keep;
        if (a.nt_loads)      // operands are read once, long after they were written: keep them out of the way of the partial tiles
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
        else
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
      }
    }
  };
#pragma unroll
  for (int t = 0; t < W3NS - 1; ++t)
    if (t < T) issue(t);

  // fragment read offsets: lane -> (16-feature half G&1, k-block G>>1, point i>>2, feature quad i&3)
  const int G = lane >> 4, li = lane & 15;
  const int frag_off = ((8 * (G >> 1) + (li >> 2)) * 64) + (G & 1) * 32 + (li & 3) * 8;
  const bool do_bias = a.bias_col >= 0 && wc == 0;

  for (int tau = 0; tau < T; ++tau) {
    // stage tau has landed when at most the 2 x 4 newer DMA instructions of this wave are outstanding; the barrier
    // then publishes every wave's part and retires the reads of stage tau-1 (whose slot is refilled next)
    if (tau + 2 < T) { if (C::G == 4) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory"); else asm volatile("s_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory"); }
    else if (tau + 1 < T) { if (C::G == 4) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (tau + W3NS - 1 < T) issue(tau + W3NS - 1);
    const unsigned char* slot = w3lds + (tau % W3NS) * C::STAGE;
    const int st = tau >= nsteps ? tau - nsteps : tau;
    const int pb = pbeg + st * pstep;
    const bool tail = pb + W3P > pend;
    const bool bias_now = do_bias && tau < nsteps;
#pragma unroll
    for (int s2 = 0; s2 < W3P / 16; ++s2) {
      uint4 av[2], bv[NCB];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (NEAT_W3_ABLATE == 2) { av[i] = make_uint4(lane, tau, i, s2); continue; }
        const unsigned char* ap = slot + ((wr >> 5) + i) * 2048 + s2 * 1024 + frag_off;
        const v4s16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(ap));
        const v4s16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(ap + 256));
        av[i] = make_uint4(((const unsigned*)&lo)[0], ((const unsigned*)&lo)[1], ((const unsigned*)&hi)[0], ((const unsigned*)&hi)[1]);
      }
#pragma unroll
      for (int j = 0; j < NCB; ++j) {
        if (NEAT_W3_ABLATE == 2) { bv[j] = make_uint4(lane, tau, j, s2); continue; }
        const unsigned char* bp = slot + C::QA * 2048 + ((wc >> 5) + j) * 2048 + s2 * 1024 + frag_off;
        const v4s16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(bp));
        const v4s16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(bp + 256));
        bv[j] = make_uint4(((const unsigned*)&lo)[0], ((const unsigned*)&lo)[1], ((const unsigned*)&hi)[0], ((const unsigned*)&hi)[1]);
      }
      if (tail) {          // last stage of the last slice: points >= P hold whatever the padding holds -> zero them
        const int p0 = pb + 16 * s2 + 8 * (G >> 1);
        auto mask4 = [&](uint4& v) {
          unsigned w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bool k0 = p0 + 2 * e < pend, k1 = p0 + 2 * e + 1 < pend;
            w4[e] = (k0 ? (w4[e] & 0x0000FFFFu) : 0u) | (k1 ? (w4[e] & 0xFFFF0000u) : 0u);
          }
          v = make_uint4(w4[0], w4[1], w4[2], w4[3]);
        };
#pragma unroll
        for (int i = 0; i < 2; ++i) mask4(av[i]);
#pragma unroll
        for (int j = 0; j < NCB; ++j) mask4(bv[j]);
      }
      if (bias_now) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const unsigned w4[4] = {av[i].x, av[i].y, av[i].z, av[i].w};
          // one serial chain per row, fenced: the SLP vectoriser otherwise pairs the sums into v_pk_add_f32, and a packed fp32 op
          // beside MFMAs costs far more than its issue slot (-6 us per hidden-layer launch, round 5)
          float t = rsum[i];
#pragma unroll
          for (int e = 0; e < 4; ++e) { t += bf_lo(w4[e]); t += bf_hi(w4[e]); }
          asm volatile("" : "+v"(t));
          rsum[i] = t;
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (!liveR[i]) continue;
#pragma unroll
        for (int j = 0; j < NCB; ++j)
          if (liveC[j]) {
            if (NEAT_W3_ABLATE == 1) { asm volatile("" :: "v"(av[i].x), "v"(av[i].w), "v"(bv[j].x), "v"(bv[j].w)); continue; }
            acc[i][j] = NEAT_MFMA16(*reinterpret_cast<bf16x8*>(&av[i]), *reinterpret_cast<bf16x8*>(&bv[j]), acc[i][j], 0, 0, 0);
          }
      }
    }
  }
  float* dstp = a.partial + (size_t)prob * a.prob_stride + (size_t)blockIdx.y * a.split_stride;
  auto put = [&](float* q, float v) {
    if (NEAT_W3_ABLATE == 3 && v != 123.0f) return;
    *q = v;
  };
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (!liveR[i]) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = wr + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (n >= a.N) continue;
#pragma unroll
      for (int j = 0; j < NCB; ++j) {
        const int k = wc + 32 * j + (lane & 31);
        if (liveC[j] && k < a.K) put(dstp + (size_t)n * a.row_stride + a.col_off + k, acc[i][j][r]);
      }
    }
  }
  if (do_bias) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float v = rsum[i];
      v += __shfl_xor(v, 32);
      const int n = wr + 32 * i + (lane & 31);
      if (lane < 32 && n < a.N) put(dstp + (size_t)n * a.row_stride + a.bias_col, v);
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
  int lo;                       // 1: the pack holds the LOW plane of the hi/lo split, 16-bit(w - 16-bit(w)) (split-precision forward, kernels_x3.hpp)
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
    for (int e = blockIdx.y * WG + threadIdx.x; e < KS * 64; e += WG * gridDim.y) {     // grid.y slices of a tile: short dependent chains
      const int s = e >> 6, ln = e & 63;
      const int n = nt * 32 + (ln & 31), kb = 16 * s + 8 * (ln >> 5);
      float w[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) w[t] = packed_weight(d, v, rs, O, I, n, kb + t);
      if (d.lo) {
#pragma unroll
        for (int t = 0; t < 8; ++t) w[t] -= bf2f(f2bf(w[t]));
      }
      out[e] = make_uint4(pack2(w[0], w[1]), pack2(w[2], w[3]), pack2(w[4], w[5]), pack2(w[6], w[7]));
    }
  } else {
    const int KS = d.Kpad >> 1;
    float* out = a.out + d.offset + (size_t)nt * KS * 64;
    for (int e = blockIdx.y * WG + threadIdx.x; e < KS * 64; e += WG * gridDim.y) {
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
    o[j] = pack2(w0 * dphi_fast(bf_lo(hw[j])), w1 * dphi_fast(bf_hi(hw[j])));
  }
  *reinterpret_cast<uint4*>(u7 + idx) = make_uint4(o[0], o[1], o[2], o[3]);
}

// bf16 octet-major [C rows] -> row-major fp32 [P, C] at column offset col0 of a [P, ldc] matrix
// (srclo: the low plane of a hi/lo split, added in; null = none)
__global__ void oct_to_rm_kernel(const u16* __restrict__ src, int P, int C, int ldp, float* __restrict__ dst, int ldc, int col0,
                                 const u16* __restrict__ srclo = nullptr) {
  // one thread per (point, octet): 16-byte read, 8 consecutive floats of the row out (grid.y = octets)
  const int p = blockIdx.x * blockDim.x + threadIdx.x, o = blockIdx.y;
  if (p >= P) return;
  const uint4 v = reinterpret_cast<const uint4*>(src)[(size_t)o * ldp + p];
  uint4 vl = make_uint4(0u, 0u, 0u, 0u);
  if (srclo) vl = reinterpret_cast<const uint4*>(srclo)[(size_t)o * ldp + p];
  const unsigned w4[4] = {v.x, v.y, v.z, v.w}, l4[4] = {vl.x, vl.y, vl.z, vl.w};
  float* d = dst + (size_t)p * ldc + col0 + o * 8;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (o * 8 + 2 * j < C) d[2 * j] = bf_lo(w4[j]) + bf_lo(l4[j]);
    if (o * 8 + 2 * j + 1 < C) d[2 * j + 1] = bf_hi(w4[j]) + bf_hi(l4[j]);
  }
}
// the lin8 output of the stand-alone SDF entry in ONE launch: raw sdf -> out257[:, 0], the 256 feature rows (octet-major, + the low
// plane of a hi/lo split) -> out257[:, 1:] and / or feat [P, 256]
__global__ void export_out8_kernel(const float* __restrict__ sdfraw, const u16* __restrict__ src, const u16* __restrict__ srclo, int P, int ldp,
                                   float* __restrict__ out257, float* __restrict__ feat) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x, o = blockIdx.y;
  if (p >= P) return;
  const uint4 v = reinterpret_cast<const uint4*>(src)[(size_t)o * ldp + p];
  uint4 vl = make_uint4(0u, 0u, 0u, 0u);
  if (srclo) vl = reinterpret_cast<const uint4*>(srclo)[(size_t)o * ldp + p];
  const unsigned w4[4] = {v.x, v.y, v.z, v.w}, l4[4] = {vl.x, vl.y, vl.z, vl.w};
  float f[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) { f[2 * j] = bf_lo(w4[j]) + bf_lo(l4[j]); f[2 * j + 1] = bf_hi(w4[j]) + bf_hi(l4[j]); }
  if (out257) {
    float* d = out257 + (size_t)p * 257 + 1 + o * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) d[j] = f[j];
    if (o == 0) out257[(size_t)p * 257] = sdfraw[p];
  }
  if (feat) {
    float* d = feat + (size_t)p * 256 + o * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) d[j] = f[j];
  }
}
// (dstlo: also write the low plane of the hi/lo split)
__global__ void rm_to_oct_kernel(const float* __restrict__ src, int P, int C, int ldp, u16* __restrict__ dst, u16* __restrict__ dstlo = nullptr) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= ldp) return;
  const int Cp = (C + 7) & ~7;
  for (int c = 0; c < Cp; ++c) {
    const float v = (p < P && c < C) ? src[(size_t)p * C + c] : 0.0f;
    const u16 h = f2bf(v);
    dst[oct_index(c, p, ldp)] = h;
    if (dstlo) dstlo[oct_index(c, p, ldp)] = f2bf(v - bf2f(h));
  }
}
__global__ void fm_col_to_rm_kernel(const float* __restrict__ src, int P, float* __restrict__ dst, int ldc, int col) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < P) dst[(size_t)p * ldc + col] = src[p];
}

}  // namespace neat

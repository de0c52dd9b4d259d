// Streaming layer kernels of the SDF network's tangent / reverse chains WITH the layer's weight gradient accumulated on chip
// (round 4; VERDICT r3 "next" #1).
//
// The weight gradient of hidden layer l is  dW_l = a^_l (x) in_l  +  u_l (x) vhat_l  (contraction over all points).  Both operands
// of the second pair are in LDS inside the tangent launch of layer l (vhat_l = its input tile, u_l = an epilogue operand), both
// operands of the first pair inside the reverse launch of layer l (a^_l = its input tile, h_l = an epilogue operand).  Up to
// round 3 a separate launch (wgrad_kernel_h3) read all four 68 MB arrays again.  Here the launch that already holds them
// contracts them:
//  * wave w keeps a 64 x 128 fp32 block of the product in 128 accumulator registers for the whole launch:
//      tangent launch:  D[n][k] = sum_p u_l[n][p] vhat_l[k][p],  n in the wave's own 32 output rows;
//      reverse launch:  D[n][k] = sum_p a^_l[n][p] h_l[k][p],    n = rows 32w..32w+31 of the launch's INPUT tile, k over the h_l image
//                       that the eight waves' epilogue-operand slices form together -- the same orientation;
//    per 32-point stage that is 16 more MFMAs per wave (the layer itself: 16), fed by `ds_read_b64_tr_b16` from the stage the
//    LDS-DMA ring delivered anyway -- no additional HBM byte is read;
//  * the octet-major stage image [octet][32 points][16 B] would serve the transposing reads with 4-way bank conflicts (an octet
//    row is 512 B = 2 bank rows, a fragment touches 4 octets at the same points), so the image is SWIZZLED: octet o keeps
//    point p at slot p ^ 4 (o & 3).  The LDS-DMA is lane-linear on the LDS side only -- the permutation is applied to the
//    SOURCE addresses; the fragment reads of the layer (ds_read_b128), the epilogue reads (ds_read_b64) and the transposing
//    reads all stay conflict free;
//  * at the end of the launch every workgroup owns one 256 x 256 partial.  256 workgroups x 256 KiB of fp32 would cost as
//    much traffic as the operand reads they replace (the objection recorded in DESIGN.md in rounds 2 and 3), so the partial
//    leaves as BLOCK-SCALED f16: one power-of-two scale per (workgroup, wave), 11-bit mantissas relative to the block
//    maximum, 128 KiB per workgroup; `dw_gather_kernel` sums the 256 partials of a (layer, pair) in fp32 in a fixed order
//    (deterministic) into the partial-tile format the weight-norm finish (wreduce_*) already reads.
//  * reverse launches also take the bias gradient (row sums of a^_l) from the fragments they read anyway.
#pragma once
#include "kernels_bf16.hpp"

namespace neat {

struct LayerArgsDW {
  LayerArgsWS w;                       // the layer itself (see layer_kernel_ws)
  const u16* auxA2; int auxA_split;    // own-row operand of the gradient: octets >= auxA_split come from auxA2 (skip layer: [h4 | PE 7..38]); 32 = none
  int rowsA;                           // its valid rows (later rows count as zero)
  int P;                               // valid points (the last tile is masked)
  uint4* partial;                      // [grid][8 waves][8 column blocks][2][64 lanes] x 8 f16
  float* pscale;                       // [grid][8]: factor that takes the stored values back
  float* pbias;                        // reverse launches: [grid][256] row sums of the input tile (bias gradient); else null
  int ablate;                          // probe runs only (results WRONG; tuning key 17): 4 = no partial stores
};

constexpr int DW_MAXSEG = 4;                      // layers one launch may run back to back (same epilogue variant; round 5)
struct LayerArgsDWSeg { LayerArgsDW l[DW_MAXSEG]; int n; };

constexpr int DW_XLDS = 1024;                     // bytes behind the ring: lin8's sdf weight row (EPI_BWD8)
constexpr int DW_WG_UINT4 = 8 * 8 * 2 * 64;      // uint4 per workgroup partial (128 KiB)
#ifndef DW_VALU_PER_MFMA
#define DW_VALU_PER_MFMA 14
#endif

// A uniform 64-bit value as an SGPR pair (the compiler cannot see that a pointer derived from threadIdx.x >> 6 is wave-uniform)
__device__ __forceinline__ unsigned long long dw_uniform64(unsigned long long v) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}
// LDS-DMA, 64 lanes x 16 B: source = SGPR base + 32-bit lane offset, LDS destination = M0 + lane * 16 (non-temporal fetch: every
// operand of these launches is read once)
__device__ __forceinline__ void dw_dma16(unsigned long long sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3 nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(sbase) : "memory");
}
__device__ __forceinline__ void dw_dma4(unsigned long long sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(sbase) : "memory");
}

// A launch runs seg.n consecutive layers of one chain (same epilogue variant) back to back: every workgroup walks its OWN tiles layer
// after layer -- layer l + 1 reads, tile by tile, what the same workgroup wrote in layer l (through the XCD's L2: the DMA fetches are
// non-temporal, i.e. they bypass this CU's L1), so no workgroup waits for another and the launch boundaries between the layers (and the
// ragged last round of each: 64 of 256 workgroups own a 17th tile at C2) disappear.  Between two layers: every wave drains its stores,
// one workgroup barrier.
// (Only the FULL variants carry the layer loop: the row-tested variants and lin8's are single launches anyway, and with the loop around them
// their register allocation spilled 16-52 B more -- lin3's tangent launch 70 -> 96 us, lin4's reverse launch 63 -> 87 us.)
template <int EPI, bool FULL>     // FULL: N = 256 output rows and 256 valid rows of the gradient's row operand (no row tests anywhere)
__global__ __launch_bounds__(WST, 2) void layer_kernel_wsdw(LayerArgsDWSeg seg) {
  constexpr bool SEG = FULL && EPI != EPI_BWD8;
  typedef WsCfg<EPI, 16> C;
  static_assert(C::NAUX == 2, "tangent / reverse epilogues only");
  constexpr bool TANK = (EPI == EPI_TAN || EPI == EPI_TAN_PF);      // gradient operand = aux1 (u_l); reverse: aux0 (h_l)
  // (the per-layer arguments are read from the kernel-argument segment through a pointer: indexing the by-value struct with a run-time
  // index would copy it to scratch)
  const LayerArgsDW* tab = (const LayerArgsDW*)__builtin_amdgcn_kernarg_segment_ptr();
  const int nlayers = SEG ? seg.n : 1;
  for (int li = 0; li < nlayers; ++li) {
  if (SEG && li > 0) {                                   // layer li - 1's stores (all waves') before layer li's fetches; the ring is free again
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  const LayerArgsDW d = SEG ? tab[li] : seg.l[0];
  const LayerArgsWS& a = d.w;
  extern __shared__ __attribute__((aligned(16))) unsigned char wslds[];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef __attribute__((address_space(3))) v4s16* lds_v4;
  const int lane0 = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int t_begin = (int)blockIdx.x, tstride = (int)gridDim.x;
  const int T = (a.ntiles - t_begin + tstride - 1) / tstride;
  const unsigned lds_base = (unsigned)(size_t)(lds_ptr)wslds;
  const unsigned dma_off = (unsigned)(4 * wave) * (WSP * 16);
  constexpr unsigned AUX0 = C::IN_TILE, AUX1 = C::IN_TILE + WS_TILE, EXTRA = C::IN_TILE + WS_TILE * C::NAUX;
  constexpr unsigned AUXA = TANK ? AUX1 : AUX0;
  // Every lane-dependent offset is derived inside the loop from a lane id the optimiser cannot see through: hoisted out of the
  // loop they would sit in ~20 registers next to the 64 weight and 128 accumulator registers and spill.
  auto lane_now = [&]() { int l = lane0; asm volatile("" : "+v"(l)); return l; };
  // LDS-DMA of two octet rows of an array (this wave's octets 4w..4w+3 as two instructions); the source addresses carry the
  // swizzle of the image.  The octet rows of one instruction come from ONE array (split points are multiples of 4 octets).
  auto dma = [&](const u16* base, int ln, int tile, unsigned dst, int max_oct, const u16* base2, int split) {
    const bool second = 4 * wave >= split;
    const unsigned long long sb = dw_uniform64((unsigned long long)(second ? base2 : base) + (unsigned long long)tile * (WSP * 16));
    const int o0 = second ? 4 * wave - split : 4 * wave;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int opos = 2 * i + (ln >> 5);                                 // octet row of the image, modulo 4
      const int oct = min(o0 + opos, second ? 31 : max_oct);              // (rows past a 217-row array: re-read a valid octet)
      const unsigned voff = ((unsigned)oct * (unsigned)a.ldp + (unsigned)((ln & 31) ^ (opos << 2))) * 16u;
      dw_dma16(sb, voff, __builtin_amdgcn_readfirstlane(dst + i * (2 * WSP * 16)));
    }
  };
  auto issue = [&](int tau) {
    const unsigned stage = lds_base + (unsigned)(tau % C::NS) * C::STAGE;
    const unsigned slot = stage + dma_off;
    const int tile = t_begin + tau * tstride;
    const int ln = lane_now();
    dma(a.in, ln, tile, slot, a.in_octs - 1, a.in2, a.split_oct);
    dma(a.aux0, ln, tile, slot + AUX0, 31, TANK ? nullptr : d.auxA2, TANK ? (1 << 30) : d.auxA_split);
    dma(a.aux1, ln, tile, slot + AUX1, 31, TANK ? d.auxA2 : nullptr, TANK ? d.auxA_split : (1 << 30));
    if (C::HAS_S) dw_dma4(dw_uniform64((unsigned long long)(a.srow + (size_t)tile * WSP)), (unsigned)(ln & 31) * 4u, __builtin_amdgcn_readfirstlane(stage + EXTRA + wave * 256));
    if (C::HAS_PF) dw_dma16(dw_uniform64((unsigned long long)(a.padfill + (size_t)tile * WSP * 8)), (unsigned)(ln & 31) * 16u, __builtin_amdgcn_readfirstlane(stage + EXTRA));
  };
#pragma unroll
  for (int t = 0; t < C::NS - 1; ++t)
    if (t < T) issue(t);

  const bool live = FULL || wave * 32 < a.N;
  uint4 wreg[16];
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) wreg[ks] = live ? a.Wp[((size_t)wave * a.kstride + ks) * 64 + lane0] : make_uint4(0u, 0u, 0u, 0u);
  // EPI_BWD8 only: the sdf row of lin8 (effective weight), 256 floats behind the ring (16 more registers per lane would spill); the
  // first stage's barrier publishes it
  float* w8 = reinterpret_cast<float*>(wslds + C::LDS);
  if (EPI == EPI_BWD8 && threadIdx.x < 256) w8[threadIdx.x] = (int)threadIdx.x < a.N ? a.wrow[threadIdx.x] * a.wrow_scale[0] : 0.0f;
  const int Npad = (a.N + 7) & ~7;

  f32x16 dacc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) dacc[j][r] = 0.0f;
  float bsum[2] = {0.0f, 0.0f};
  const int wr = 64 * (wave >> 1), wc = 128 * (wave & 1);       // this wave's block of the gradient
  const bool BIAS = d.pbias != nullptr && wc == 0;

  for (int tau = 0; tau < T; ++tau) {
    const int ahead = min(C::NS - 2, T - 1 - tau);
    ws_wait_barrier(ahead * C::G);
    if (tau + C::NS - 1 < T) issue(tau + C::NS - 1);
    const unsigned char* slot = wslds + (tau % C::NS) * C::STAGE;
    const int tile = t_begin + tau * tstride;
    const int ln = lane_now();
    const unsigned hi5 = (unsigned)(ln >> 5), l31 = (unsigned)(ln & 31);
    // ---- weight gradient: 2 k-steps of 16 points; this wave's 64 x 128 block = 2 row fragments x 4 column fragments per k-step
    // (6 transposing fragment reads per 8 MFMAs; a 1 x 8 blocking reads 9 -- the reads, not the MFMAs, are what the gradient costs)
    // transposing reads: lane -> (16-feature half G & 1, k-group G >> 1, point li >> 2, feature quad li & 3) of a 32-feature block
    const unsigned G = (unsigned)(ln >> 4), li = (unsigned)(ln & 15);
    const unsigned ol = 2 * (G & 1) + ((li >> 1) & 1);                            // octet of the block
    const unsigned tro0 = ol * 512 + (((2 * (G >> 1)) ^ ol) * 4 + (li >> 2)) * 16 + (li & 1) * 8;       // points 8 (G >> 1) + 0..3
    const unsigned tro1 = ol * 512 + (((2 * (G >> 1) + 1) ^ ol) * 4 + (li >> 2)) * 16 + (li & 1) * 8;   // points 8 (G >> 1) + 4..7
    const bool tail = (tile + 1) * WSP > d.P;
    // tangent: rows = u_l (the AUX1 image, complete after the barrier), columns = the input tile vhat_l;
    // reverse: rows = the input tile a^_l, columns = h_l (the AUX0 image)
    const unsigned char* abase = (TANK ? slot + AUX1 : slot) + (wr >> 5) * 2048;
    const unsigned char* bbase = (TANK ? slot : slot + AUX0) + (wc >> 5) * 2048;
    auto frag = [&](const unsigned char* base, int s2) {
      const v4s16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(base + tro0 + s2 * 256));
      const v4s16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(base + tro1 + s2 * 256));
      return make_uint4(((const unsigned*)&lo)[0], ((const unsigned*)&lo)[1], ((const unsigned*)&hi)[0], ((const unsigned*)&hi)[1]);
    };
    auto load_frags = [&](int s2, uint4 (&av)[2], uint4 (&bv)[4]) {
#pragma unroll
      for (int i = 0; i < 2; ++i) av[i] = frag(abase + i * 2048, s2);
#pragma unroll
      for (int j = 0; j < 4; ++j) bv[j] = frag(bbase + j * 2048, s2);
    };
    auto mask_frags = [&](int s2, uint4 (&av)[2], uint4 (&bv)[4], bool points) {
      // rows past the operand's last one count as zero; on the last tile (points) so do the points >= P of both operands
      const int arow = wr + (int)(16 * (G & 1) + li);
      const unsigned amask0 = arow < d.rowsA ? 0xFFFFFFFFu : 0u, amask1 = arow + 32 < d.rowsA ? 0xFFFFFFFFu : 0u;
      av[0].x &= amask0; av[0].y &= amask0; av[0].z &= amask0; av[0].w &= amask0;
      av[1].x &= amask1; av[1].y &= amask1; av[1].z &= amask1; av[1].w &= amask1;
      if (!points) return;
      auto mask4 = [&](uint4& v) {
        const int p0 = tile * WSP + 16 * s2 + 8 * (int)(G >> 1);
        unsigned w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool k0 = p0 + 2 * e < d.P, k1 = p0 + 2 * e + 1 < d.P;
          w4[e] = (k0 ? (w4[e] & 0x0000FFFFu) : 0u) | (k1 ? (w4[e] & 0xFFFF0000u) : 0u);
        }
        v = make_uint4(w4[0], w4[1], w4[2], w4[3]);
      };
#pragma unroll
      for (int i = 0; i < 2; ++i) mask4(av[i]);
#pragma unroll
      for (int j = 0; j < 4; ++j) mask4(bv[j]);
    };
    auto dw_mma = [&](const uint4 (&av)[2], const uint4 (&bv)[4], bool mfma = true) {
      if (!TANK && BIAS) {                            // bias gradient: row sums of a^_l (kept by the waves of column half 0)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const unsigned w4[4] = {av[i].x, av[i].y, av[i].z, av[i].w};
          float t = 0.0f;
#pragma unroll
          for (int e = 0; e < 4; ++e) t += bf_lo(w4[e]) + bf_hi(w4[e]);
          bsum[i] += t;
        }
      }
      if (!mfma) return;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          dacc[i * 4 + j] = NEAT_MFMA16(*reinterpret_cast<const bf16x8*>(&av[i]), *reinterpret_cast<const bf16x8*>(&bv[j]), dacc[i * 4 + j], 0, 0, 0);
    };
    // ---- the layer: 16 dependent MFMAs on the weight slice in registers
    f32x16 acc;
    auto chain = [&]() {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
      const unsigned bfrag_e = (hi5 * WSP + (l31 ^ (hi5 << 2))) * 16;               // even k-steps: octet 2 ks + hi5, (octet & 3) = hi5
      const unsigned bfrag_o = (hi5 * WSP + (l31 ^ ((2 + hi5) << 2))) * 16;         // odd k-steps: (octet & 3) = 2 + hi5
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const uint4 bv = *reinterpret_cast<const uint4*>(slot + ((ks & 1) ? bfrag_o : bfrag_e) + ks * (2 * WSP * 16));
        acc = NEAT_MFMA16(*reinterpret_cast<const bf16x8*>(&wreg[ks]), *reinterpret_cast<const bf16x8*>(&bv), acc, 0, 0, 0);
      }
    };
    const int p = tile * WSP + (int)l31;
    // epilogue of output quads 2j, 2j+1 (one 16-byte store per lane and output array); FULLROWS: all 256 rows exist (no row tests:
    // straight-line code, so that the scheduler may place the gradient MFMAs between its vector instructions)
    auto epi_half = [&](int j, bool fullrows, const uint4* fa = nullptr, const uint4* fb = nullptr) {
      float sp = 0.0f;
      if (C::HAS_S) sp = *reinterpret_cast<const float*>(slot + EXTRA + wave * 256 + l31 * 4);
      uint2 pk0[2], pk1[2];
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) {
        const int q = 2 * j + qq;
        pk0[qq] = make_uint2(0u, 0u); pk1[qq] = make_uint2(0u, 0u);
        const int n0 = wave * 32 + 8 * q + 4 * (int)hi5;
        if (!fullrows && fa) {       // layers with fewer rows: the quad's four gradient MFMAs up front, whether or not it has output rows
                                     // (the accumulators must not be assigned in alternative branches: the allocator then spills them)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int t = qq * 4 + e;
            dacc[t] = NEAT_MFMA16(*reinterpret_cast<const bf16x8*>(&fa[t >> 2]), *reinterpret_cast<const bf16x8*>(&fb[t & 3]), dacc[t], 0, 0, 0);
          }
        }
        if (!fullrows && wave * 32 + 8 * q >= Npad) continue;      // (wave-uniform) no output rows in this quad
        const unsigned eoff = dma_off + q * (WSP * 16) + ((l31 ^ ((unsigned)q << 2)) * 16) + hi5 * 8;
        const uint2 r0v = *reinterpret_cast<const uint2*>(slot + AUX0 + eoff);
        const uint2 r1v = *reinterpret_cast<const uint2*>(slot + AUX1 + eoff);
        const float x0[4] = {bf_lo(r0v.x), bf_hi(r0v.x), bf_lo(r0v.y), bf_hi(r0v.y)};
        const float x1[4] = {bf_lo(r1v.x), bf_hi(r1v.x), bf_lo(r1v.y), bf_hi(r1v.y)};
        float o0[4], o1[4];
        float4 b8 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (EPI == EPI_BWD8) b8 = *reinterpret_cast<const float4*>(w8 + n0);
        const float bias8q[4] = {b8.x, b8.y, b8.z, b8.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = acc[4 * q + e];
          if (fa && fullrows) {             // one gradient MFMA per output element: the matrix pipe works under this element's vector instructions
            const int t = qq * 4 + e;
            dacc[t] = NEAT_MFMA16(*reinterpret_cast<const bf16x8*>(&fa[t >> 2]), *reinterpret_cast<const bf16x8*>(&fb[t & 3]), dacc[t], 0, 0, 0);
          }
          float r0 = 0.0f, r1 = 0.0f;
          if (TANK) { const float sg = dphi_fast(x0[e]); r0 = v * sg; r1 = v * x1[e] * (100.0f * (1.0f - sg)); }
          else if (EPI == EPI_BWD) r0 = v * dphi_fast(x0[e]) + x1[e];
          else if (EPI == EPI_BWD8) r0 = (v + bias8q[e] * sp) * dphi_fast(x0[e]) + x1[e];
          if (!fullrows && n0 + e >= a.N) {
            r0 = 0.0f; r1 = 0.0f;
            if (C::HAS_PF && n0 + e < a.N + 7)
              r0 = bf2f(*reinterpret_cast<const u16*>(slot + EXTRA + l31 * 16 + (n0 + e - a.N) * 2));
          }
          o0[e] = r0; o1[e] = r1;
        }
        pk0[qq] = make_uint2(pack2(o0[0], o0[1]), pack2(o0[2], o0[3]));
        if (TANK) pk1[qq] = make_uint2(pack2(o1[0], o1[1]), pack2(o1[2], o1[3]));
      }
      // 16-byte stores: lanes 0-31 get the whole octet of quad 2j, lanes 32-63 that of quad 2j+1 (v_permlane32_swap)
      const int ob = wave * 32 + 16 * j;
      if (!fullrows && ob >= Npad) return;
      const int oct = (ob >> 3) + (int)hi5;
      const unsigned oidx = ((unsigned)oct * (unsigned)a.ldp + (unsigned)p) * 8u;
      const bool in_range = fullrows || oct * 8 < Npad;
      typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
      typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
      {
        const v2u_t s0 = __builtin_amdgcn_permlane32_swap(pk0[0].x, pk0[1].x, false, false);
        const v2u_t s1 = __builtin_amdgcn_permlane32_swap(pk0[0].y, pk0[1].y, false, false);
        if (in_range) {
          const v4u_t v = {s0.x, s1.x, s0.y, s1.y};
          *reinterpret_cast<v4u_t*>(a.out0 + oidx) = v;
        }
      }
      if (TANK) {
        const v2u_t s0 = __builtin_amdgcn_permlane32_swap(pk1[0].x, pk1[1].x, false, false);
        const v2u_t s1 = __builtin_amdgcn_permlane32_swap(pk1[0].y, pk1[1].y, false, false);
        if (in_range) *reinterpret_cast<uint4*>(a.out1 + oidx) = make_uint4(s0.x, s1.x, s0.y, s1.y);
      }
    };
    // Order of a stage: fragments of the first k-step, the layer's MFMA chain, then per epilogue half (two output quads) the eight
    // gradient MFMAs of one k-step -- in the same basic block as the half's vector instructions, so that they go out between them
    // (FULL: no row tests, straight-line code; the hints below ask for one MFMA per DW_VALU_PER_MFMA vector instructions) -- and
    // the fragments of the second k-step requested at the start of the second half.
    uint4 av[2], bv[4];
    load_frags(0, av, bv);
    if (!FULL || tail) mask_frags(0, av, bv, tail);
    if (live) chain();            // (waves without output rows -- lin3's 217 -- keep their share of the gradient: epi_half issues it)
    __builtin_amdgcn_sched_barrier(0);
    dw_mma(av, bv, false);
    epi_half(0, FULL, av, bv);
    __builtin_amdgcn_sched_barrier(0);
    load_frags(1, av, bv);
    if (!FULL || tail) mask_frags(1, av, bv, tail);
    dw_mma(av, bv, false);
    epi_half(1, FULL, av, bv);
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- the workgroup's partial: block-scaled f16, one scale per wave
  const int lane = lane0;
  float amax = 0.0f;
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) amax = fmaxf(amax, fabsf(dacc[j][r]));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
  unsigned ex = (__float_as_uint(amax) >> 23) & 255u;
  ex = ex < 15u ? 15u : (ex > 253u ? 253u : ex);
  const float sc = __uint_as_float((268u - ex) << 23);          // block maximum -> [2^14, 2^15)
  uint4* dst = d.partial + (size_t)blockIdx.x * DW_WG_UINT4 + (size_t)wave * (8 * 2 * 64) + lane;
  typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      if (d.ablate & 4) continue;
      unsigned w4[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const f32x2_t v = {dacc[j][8 * hf + 2 * e] * sc, dacc[j][8 * hf + 2 * e + 1] * sc};
        w4[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, h2_t));
      }
#ifndef NEAT_DW_NT_PART
#define NEAT_DW_NT_PART 0
#endif
      if (NEAT_DW_NT_PART) {
        typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
        const v4u_t wv = {w4[0], w4[1], w4[2], w4[3]};
        __builtin_nontemporal_store(wv, reinterpret_cast<v4u_t*>(dst + (j * 2 + hf) * 64));
      } else dst[(j * 2 + hf) * 64] = make_uint4(w4[0], w4[1], w4[2], w4[3]);
    }
  if (lane == 0) d.pscale[blockIdx.x * 8 + wave] = __uint_as_float((ex - 14u) << 23);
  if (!TANK && BIAS) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float t = bsum[i] + __shfl_xor(bsum[i], 32);
      if (lane < 32) d.pbias[blockIdx.x * 256 + wr + 32 * i + lane] = t;
    }
  }
  }      // layers of the segment
}

// ---------------------------------------------------------------------------------------------
// dw_gather_kernel: sum of the block-scaled f16 partials of one (layer, pair) over a range of workgroups, in fp32 and in a fixed
// order, written as one split of the partial-tile format of wreduce_* ((split, n, k) at n * row_stride + split * split_stride + k).
// grid = (32, NSUB, jobs): thread = one uint4 position of a workgroup partial (8 values), blockIdx.y = which 1/NSUB of the
// workgroups.  `transposed` jobs (reverse launches) hold D[k][n].  The bias row sums of a reverse job go to column bias_col of
// split `split0` (sub-range 0 only... every sub-range adds its own workgroups).
// ---------------------------------------------------------------------------------------------
struct DwGatherJob {
  const uint4* partial; const float* pscale; const float* pbias;     // nwg workgroup partials; pbias: [nwg][256] or null
  const uint4* partial2; const float* pscale2;                        // a second set of nwg partials summed into the same splits (or null)
  int nwg, transposed;
  float* out; size_t row_stride, split_stride; int split0;       // sub-range y writes split split0 + y
  int bias_col;
  int rows, cols;                                                 // valid packed rows n / packed columns k (others are not written)
  const float* xrow; int xrow_n; size_t xrow_stride;              // one more packed row `rows` (lin8: the sdf row, rowdot_kernel's xrow_n block
                                                                  // partials of cols + 1 values -- bias column included); null: none
};
constexpr int DW_MAXJOBS = 16;
struct DwGatherArgs { DwGatherJob job[DW_MAXJOBS]; };

__global__ __launch_bounds__(256) void dw_gather_kernel(DwGatherArgs) {
  const DwGatherJob* tab = (const DwGatherJob*)__builtin_amdgcn_kernarg_segment_ptr();
  const DwGatherJob jb = tab[blockIdx.z];
  const int nsub = gridDim.y, sub = blockIdx.y;
  const int per = (jb.nwg + nsub - 1) / nsub;
  const int g0 = sub * per, g1 = min(jb.nwg, g0 + per);
  const int pos = blockIdx.x * 256 + threadIdx.x;             // uint4 position within a workgroup partial: ((wave * 8 + j) * 2 + hf) * 64 + lane
  const int lane = pos & 63, hf = (pos >> 6) & 1, j = (pos >> 7) & 7, wave = pos >> 10;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int set = 0; set < 2; ++set) {
    const uint4* part = set ? jb.partial2 : jb.partial;
    const float* psc = set ? jb.pscale2 : jb.pscale;
    if (!part) break;
#pragma unroll 8
    for (int g = g0; g < g1; ++g) {
      typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
      const v4u_t v = __builtin_nontemporal_load(reinterpret_cast<const v4u_t*>(part + (size_t)g * DW_WG_UINT4 + pos));
      const float s = psc[g * 8 + wave];
      const unsigned w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const h2_t h = __builtin_bit_cast(h2_t, w4[e]);
        acc[2 * e] += (float)h.x * s;
        acc[2 * e + 1] += (float)h.y * s;
      }
    }
  }
  float* out = jb.out + (size_t)(jb.split0 + sub) * jb.split_stride;
  const int col = 128 * (wave & 1) + 32 * (j & 3) + (lane & 31);             // tile j of wave w: rows 64 (w >> 1) + 32 (j >> 2).., columns 128 (w & 1) + 32 (j & 3)..
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int r = 8 * hf + e;
    const int row = 64 * (wave >> 1) + 32 * (j >> 2) + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    const int n = jb.transposed ? col : row, k = jb.transposed ? row : col;
    if (n < jb.rows && k < jb.cols) out[(size_t)n * jb.row_stride + k] = acc[e];
  }
  if (jb.bias_col >= 0 && blockIdx.x == 0) {                   // bias column of this split: sum of the sub-range's row sums (or zero)
    const int n = threadIdx.x;
    float t = 0.0f;
    if (jb.pbias)
      for (int g = g0; g < g1; ++g) t += jb.pbias[g * 256 + n];
    if (n < jb.rows) out[(size_t)n * jb.row_stride + jb.bias_col] = t;
  }
  if (jb.xrow && blockIdx.x == 1) {                            // extra row: this sub-range's share of the block partials, in block order
    const int perx = (jb.xrow_n + nsub - 1) / nsub;
    const int b0 = sub * perx, b1 = min(jb.xrow_n, b0 + perx);
    for (int k = threadIdx.x; k <= jb.cols; k += 256) {
      float t = 0.0f;
#pragma unroll 8
      for (int b = b0; b < b1; ++b) t += jb.xrow[(size_t)b * jb.xrow_stride + k];
      out[(size_t)jb.rows * jb.row_stride + k] = t;
    }
  }
}

}  // namespace neat

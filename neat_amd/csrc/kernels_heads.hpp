// The two heads of the 16-bit builds as fused chains: one launch per head and direction.
//   head_chain_kernel      rendering network (rend_a :199-255) / attraction field (:139-197) forward:
//                          [256 feature rows | small inputs] -> 4 x (256, ReLU) -> 3 (sigmoid) / 6 (linear)
//   head_bwd_chain_kernel  their backward chains: cotangent of the outputs -> 4 x (W^T, ReLU mask) -> feature cotangent (256 rows,
//                          16-bit, the second head adds to the first) + the small inputs' cotangents (fp32 rows)
// Per-layer launches (layer_kernel_ws) move every hidden array through HBM once as an output and once more as the next launch's
// input, and the backward kernel reads the saved activation only for its sign.  Here the activations stay in LDS between layers;
// HBM sees each hidden array once (written: the weight gradients contract them later) and a 1-bit ReLU mask per element (written by
// the forward chain in the accumulator layout, read back by the lane that owns the same elements in the backward chain).
//
// Shape of both kernels (kernels_x3.hpp's, with one product per k-step):
//   * persistent 8-wave workgroup per CU; wave w owns output rows 32w .. 32w+31; weights of the layer in registers (16 A fragments,
//     64 VGPRs), refilled slot by slot with the next layer's during the layer's last tile;
//   * a batch = 1 or 2 pairs of 32-point tiles (64 / 128 points).  Full rounds of 2-pair batches are interleaved over the workgroups;
//     what is left (< 2 batches per workgroup) goes out as single pairs, so the last round costs half a batch, not a whole one;
//   * stage = 16 k-steps of one (layer, tile) with the epilogue of the PREVIOUS stage's accumulators issued element by element behind
//     the MFMAs; B fragments are read three k-steps ahead; one workgroup barrier per PAIR of stages in a 2-pair batch (tile t of layer
//     l + 1 only needs tile t of layer l, whose epilogue ran at least two stages earlier), one per stage in a single-pair batch.
#pragma once
#include "bf16_common.hpp"
#include "fused_launch.hpp"

namespace neat {

struct HC {
  static constexpr int BP = HC_BATCH, THREADS = 512;
  static constexpr int XPL = 32 * BP * 16;              // an activation buffer [32 octets][BP][16 B] = 64 KiB
  static constexpr int SPL = 8 * BP * 16;               // small inputs / output cotangents, K padded to 64 rows = 16 KiB
  static constexpr int XA = 0, XB = XPL, S = 2 * XPL;
  static constexpr int BIAS = S + SPL;
  static constexpr int BIAS_FLOATS = 4 * 256 + 8;
  static constexpr int LDS = BIAS + BIAS_FLOATS * 4;     // 151 584 B
  static constexpr int KSTEP = 2 * BP * 16;             // bytes between the k-steps of a fragment column
};

#ifndef NEAT_HC_RING
#define NEAT_HC_RING 3        // B fragments in flight ahead of the MFMA that consumes them
#endif
#ifndef NEAT_HC_ABLATE
#define NEAT_HC_ABLATE 0      // probe builds only (results are WRONG), bit mask: 1 = no epilogue, 2 = no MFMAs, 4 = no HBM stores of the hidden
                              // arrays, 8 = no masks, 16 = no LDS writes of the epilogue, 32 = no bias reads, 64 = one add per element instead of the epilogue,
                              // 128 = no barriers inside the layers, 256 = only a workgroup's first batch loads its inputs (forward kernel)
#endif

#ifndef NEAT_HC_NT_FWD
#define NEAT_HC_NT_FWD 1      // the forward chain's hidden arrays (read again only by the backward pass's weight gradients) leave with non-temporal
                              // stores: they no longer push the layers' weight fragments, which every batch re-reads, out of the XCD's L2 (round 5:
                              // forward chains 113 -> 105 us)
#endif
#ifndef NEAT_HC_NT_BWD
#define NEAT_HC_NT_BWD 0      // the backward chain's cotangent arrays (read by the weight-gradient launch right behind it)
#endif
template <bool NT = false>
__device__ __forceinline__ void hc_store8(void* p, uint2 v) {
  typedef unsigned long long u64_t;
  if (NT) __builtin_nontemporal_store(__builtin_bit_cast(u64_t, v), reinterpret_cast<u64_t*>(p));
  else *reinterpret_cast<uint2*>(p) = v;
}
#ifndef NEAT_HC_WIDE
#define NEAT_HC_WIDE 1        // the hidden arrays leave in 16-byte stores: v_permlane32_swap hands lanes 0-31 the whole octet of an even quad and
                              // lanes 32-63 the whole octet of the odd quad next to it (the accumulator layout gives a lane half an octet)
#endif
// quads q - 1 (prev) and q (cur, odd) of this lane -> the octet this lane stores: rows 8 (q - 1) .. + 7 for lanes 0-31, 8 q .. + 7 for lanes 32-63
__device__ __forceinline__ uint4 hc_octet(uint2 prev, uint2 cur) {
  typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
  const v2u_t s0 = __builtin_amdgcn_permlane32_swap(prev.x, cur.x, false, false);
  const v2u_t s1 = __builtin_amdgcn_permlane32_swap(prev.y, cur.y, false, false);
  return make_uint4(s0.x, s1.x, s0.y, s1.y);
}
template <bool NT = false>
__device__ __forceinline__ void hc_store16(void* p, uint4 v) {
  typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
  const v4u_t w = {v.x, v.y, v.z, v.w};
  if (NT) __builtin_nontemporal_store(w, reinterpret_cast<v4u_t*>(p));
  else *reinterpret_cast<v4u_t*>(p) = w;
}
#ifndef NEAT_HC_TIMING
#define NEAT_HC_TIMING 0      // probe builds only: workgroup 0 prints the cycle counts of its first batches' phases (forward kernel)
#endif
#if NEAT_HC_TIMING
#define HC_STAMP(i) do { if (blockIdx.x == 0 && it < 3) { __builtin_amdgcn_s_waitcnt(0); stamp[i] = __builtin_readcyclecounter(); } } while (0)
#else
#define HC_STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ uint4 hc_ldg(const void* base, unsigned off) {
  return *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(base) + off);
}
#ifndef NEAT_HC_NT_IN
#define NEAT_HC_NT_IN 0       // the chains' input tiles (feature rows, small inputs, output cotangents) arrive with non-temporal loads
#endif
__device__ __forceinline__ uint4 hc_ldg_in(const void* base, unsigned off) {
  typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
  if (NEAT_HC_NT_IN) {
    const v4u_t v = __builtin_nontemporal_load(reinterpret_cast<const v4u_t*>(reinterpret_cast<const unsigned char*>(base) + off));
    return make_uint4(v.x, v.y, v.z, v.w);
  }
  return hc_ldg(base, off);
}

// One stage: KS k-steps of `acc` over the fragment column `fr`, epi(e) called 16 / KS times per k-step for e = 0 .. 15.
// ZERO: the accumulator starts at zero.  ROLL: slot ks of the weight registers is refilled with the next layer's fragment
// (wn + woff + ks KiB, NKS slots) right after its MFMA; slots KS .. NKS-1 (free in this layer) are requested up front.
template <int KS, bool ZERO, bool ROLL, int NKS, int WN, class Epi>
__device__ __forceinline__ void hc_stage(const unsigned char* fr, uint4 (&w)[WN], f32x16& acc, const void* wn, unsigned woff, Epi&& epi) {
  constexpr int RING = NEAT_HC_RING;
  uint4 b[KS];
#pragma unroll
  for (int i = 0; i < RING && i < KS; ++i) b[i] = *reinterpret_cast<const uint4*>(fr + i * HC::KSTEP);
  if (ROLL) {
#pragma unroll
    for (int ks = KS; ks < NKS; ++ks) w[ks] = hc_ldg(wn, woff + ks * 1024);
  }
  if (ZERO) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  }
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    if (ks + RING < KS) b[ks + RING] = *reinterpret_cast<const uint4*>(fr + (ks + RING) * HC::KSTEP);
    if (!(NEAT_HC_ABLATE & 2))
      acc = NEAT_MFMA16(*reinterpret_cast<const bf16x8*>(&w[ks]), *reinterpret_cast<const bf16x8*>(&b[ks]), acc, 0, 0, 0);
    else acc[0] += __uint_as_float(w[ks].x ^ b[ks].x);
    if (ROLL && ks < NKS) w[ks] = hc_ldg(wn, woff + ks * 1024);
    if (!(NEAT_HC_ABLATE & 1)) {
#pragma unroll
      for (int e = ks * (16 / KS); e < (ks + 1) * (16 / KS); ++e) epi(e);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}
template <class Epi> __device__ __forceinline__ void hc_drain(Epi&& epi) {
#pragma unroll
  for (int e = 0; e < 16; ++e) { epi(e); if ((e & 3) == 3) __builtin_amdgcn_sched_barrier(0); }
}

// Work of a workgroup: iteration `it` -> first pair and number of pairs of its batch (np = 0: done).
// NEAT_HC_STAGGER (probe, off): odd workgroups run half a batch out of phase with the even ones -- their first batch goes out as two
// single pairs, one at the start and one after the full rounds.  (In lock step all 256 workgroups fetch their 64 KiB of inputs at the
// same moment and wait for HBM together: 9 000 - 32 000 cycles per batch by the cycle counter, against ~8 000 per layer.  Measured:
// 115 -> 130 us, the two single-pair batches cost more than the stagger returns.)
#ifndef NEAT_HC_STAGGER
#define NEAT_HC_STAGGER 0
#endif
struct HcWork {
  int rounds, tail0, grid, nvp;
  __device__ __forceinline__ HcWork(int nvalid_pairs, int g) : grid(g), nvp(nvalid_pairs) {
    rounds = (nvalid_pairs / 2) / g;          // full rounds of 2-pair batches
    tail0 = rounds * g * 2;
  }
  __device__ __forceinline__ int batch(int it, int bid, int& pair0) const {
    if (NEAT_HC_STAGGER && (bid & 1) && rounds > 0) {
      if (it == 0) { pair0 = 2 * bid; return 1; }
      if (it == rounds) { pair0 = 2 * bid + 1; return 1; }
      if (it > rounds) --it;
    }
    if (it < rounds) { pair0 = 2 * (it * grid + bid); return 2; }
    pair0 = tail0 + (it - rounds) * grid + bid;
    return pair0 < nvp ? 1 : 0;
  }
};

// ---------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------
template <int HEAD, bool SAVE>
__global__ __launch_bounds__(512, 2) void head_chain_kernel(HeadX3Args a, int npairs) {
  typedef HC C;
  constexpr int BP = C::BP, NOUT = HEAD ? 6 : 3;
  extern __shared__ __attribute__((aligned(16))) unsigned char hclds[];
  float* biasl = reinterpret_cast<float*>(hclds + C::BIAS);      // [4][256] + [8]
  const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int idx = tid; idx < 4 * 256; idx += C::THREADS) {
    const int l = idx >> 8, n = idx & 255;
    float v = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) if (l == k) v = a.bias[k][n];
    biasl[idx] = v;
  }
  if (tid < 8) biasl[4 * 256 + tid] = tid < NOUT ? a.bias[4][tid] : 0.0f;
  for (int idx = tid; idx < C::SPL / 16; idx += C::THREADS) reinterpret_cast<uint4*>(hclds + C::S)[idx] = make_uint4(0u, 0u, 0u, 0u);

  const int nvp = npairs < a.nvalid ? npairs : a.nvalid;
  // ---- pairs no ray sample lives in (eikonal points, padding): nothing is computed; the saved activations are zeroed (the backward
  // pass contracts them with zero cotangents: they must be finite), and so are the masks and the outputs
  for (int pair = nvp + blockIdx.x; pair < npairs; pair += gridDim.x) {
    const unsigned p0 = (unsigned)pair * 64u;
    if (SAVE) {
#pragma unroll
      for (int l = 1; l <= 4; ++l) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int idx = tid + i * C::THREADS, oct = idx >> 6, pp = idx & 63;
          *reinterpret_cast<uint4*>(reinterpret_cast<char*>(a.hid[l]) + ((unsigned)oct * (unsigned)a.ldp + p0 + pp) * 16u) = make_uint4(0u, 0u, 0u, 0u);
        }
        if (tid < 256) reinterpret_cast<uint4*>(reinterpret_cast<char*>(a.mask[l]) + (size_t)pair * 4096)[tid] = make_uint4(0u, 0u, 0u, 0u);
      }
    }
    for (int idx = tid; idx < NOUT * 64; idx += C::THREADS) a.out[(size_t)(idx >> 6) * a.ldp + p0 + (idx & 63)] = 0.0f;
  }

  uint4 w[20], ow[2];   // the layer's 16 k-steps; [16..19]: lin0's small-input k-steps, ow: lin4's two k-steps of this wave (both stay for the whole launch)
  auto w_off = [&](int KS) -> unsigned {
    unsigned v = (unsigned)((wave * KS) * 64 + lane) * 16u;
    asm volatile("" : "+v"(v));
    return v;
  };
  {
    // lin0's 20 k-steps (later batches get the 16 feature k-steps through lin3's rolling refill)
    const unsigned o0 = w_off(20);
#pragma unroll
    for (int ks = 0; ks < 20; ++ks) w[ks] = hc_ldg(a.Wp[0], o0 + ks * 1024);
    unsigned so = (unsigned)(((2 * wave) * 64 + lane) * 16);
    asm volatile("" : "+v"(so));
#pragma unroll
    for (int j = 0; j < 2; ++j) ow[j] = hc_ldg(a.Wp[4], so + j * 1024);
  }
  // per-lane LDS bases (opaque: every access is base + immediate (+ tile offset))
  const unsigned char* frag[3];
  unsigned char* quad[2];
  const unsigned char* biasp;
  {
    const unsigned fo = (unsigned)(hi * BP + (lane & 31)) * 16u, qo = (unsigned)((4 * wave) * BP + (lane & 31)) * 16u + 8u * hi;
    unsigned b0 = C::XA + fo, b1 = C::XB + fo, b2 = C::S + fo, q0 = C::XA + qo, q1 = C::XB + qo, bb = C::BIAS + (unsigned)(32 * wave + 4 * hi) * 4u;
    asm volatile("" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(q0), "+v"(q1), "+v"(bb));
    frag[0] = hclds + b0; frag[1] = hclds + b1; frag[2] = hclds + b2;
    quad[0] = hclds + q0; quad[1] = hclds + q1; biasp = hclds + bb;
  }
  __syncthreads();

  const HcWork work(nvp, gridDim.x);
  // (Tried, twice: requesting a batch's inputs into 40 registers during the previous batch -- right after its lin0, or at the start of
  //  its lin4 behind the last weight refill -- so that the workgroups do not all wait for HBM at the top of a batch (9 000 - 32 000 of
  //  a batch's ~55 000 cycles by the cycle counter): 115 -> 138 / 135 us.  The vector-memory counter is in order -- the rolling weight
  //  refills of the layers wait behind the prefetch --, and the loads then compete with the hidden arrays' stores they overlap.)
  for (int it = 0;; ++it) {
    int pair0;
    const int np = work.batch(it, blockIdx.x, pair0);
    if (np == 0) break;
    const int nt = 2 * np, npts = 64 * np;
#if NEAT_HC_TIMING
    unsigned long long stamp[8];
#endif
    HC_STAMP(0);
    const unsigned p0 = (unsigned)pair0 * 64u;
    unsigned ldp16 = (unsigned)a.ldp * 16u;      // (scalar: the row-quad part of a store address goes into the scalar base)
    asm volatile("" : "+s"(ldp16));
    const unsigned gquad = ((unsigned)(4 * wave) * (unsigned)a.ldp + p0 + (unsigned)(lane & 31)) * 16u + 8u * hi;
    const unsigned goct = ((unsigned)(4 * wave + hi) * (unsigned)a.ldp + p0 + (unsigned)(lane & 31)) * 16u;
    const unsigned mlane = (((p0 >> 5) * 8u + (unsigned)wave) * 64u + (unsigned)lane) * 4u;       // + t * 2048: this lane's mask word of tile t
    int tb = tid;
    asm volatile("" : "+v"(tb));
    // ---- the feature tile -> XA: 32 octet rows x npts points x 16 B; the small inputs (octet-major copies made by oct_pack) -> S
    {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int idx = tb + i * C::THREADS, oct = idx >> 7, pp = idx & 127;
        if (pp < npts) reinterpret_cast<uint4*>(hclds + C::XA)[oct * BP + pp] = hc_ldg_in(a.feat, ((unsigned)oct * (unsigned)a.ldp + p0 + pp) * 16u);
      }
      const int so = tb >> 7, sp = tb & 127;       // 4 octet rows per pass
      for (int o = so; o * 8 < a.srows; o += 4)
        if (sp < npts) reinterpret_cast<uint4*>(hclds + C::S)[o * BP + sp] = hc_ldg_in(a.smallbf, ((unsigned)o * (unsigned)a.ldp + p0 + sp) * 16u);
    }
    __syncthreads();
    HC_STAMP(1);

    f32x16 acc[2];
    // pending epilogue = the previous stage's: where its accumulators go
    unsigned pboff = 0;                 // bias row block of its layer (bytes)
    unsigned char* plq = quad[0];       // LDS quad base (+ tile)
    unsigned pg = 0;                    // HBM quad offset (+ tile)
    u16* phout = nullptr; u16* pmout = nullptr; unsigned pmoff = 0;
    float4 bqv = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned ph[2]; float keep = 0.0f; unsigned mbits = 0;
    uint2 vprev = make_uint2(0u, 0u);   // wide stores: the even quad waits for its odd neighbour
    unsigned pgw = 0;                   // ... and the lane's octet offset: lanes 32-63 one octet row further, no half-octet offset
    // one call per accumulator element; the work is done per PAIR (odd e): bias, rounding (v_cvt_pk), ReLU and mask bit on the
    // packed 16-bit pair
    auto epi_elem = [&](const f32x16& ap, int e) {
      if (NEAT_HC_ABLATE & 64) {          // matrix side alone: the accumulators stay live through one add per element
        keep += ap[e];
        if (e == 15 && keep == 12345.0f) *reinterpret_cast<float*>(plq) = keep;
        return;
      }
      const int q = e >> 2, j = e & 3;
      if (j == 0 && !(NEAT_HC_ABLATE & 32)) bqv = *reinterpret_cast<const float4*>(biasp + pboff + q * 32);
      if ((j & 1) == 0) return;
      const v2f_t av = {ap[e - 1], ap[e]};
      const v2f_t bv = (j == 1) ? v2f_t{bqv.x, bqv.y} : v2f_t{bqv.z, bqv.w};
#ifndef NEAT_HC_PK_BIAS
#define NEAT_HC_PK_BIAS 0     // 1: the bias add as one v_pk_add_f32 per pair (rounds 3-4).  A packed fp32 op beside MFMAs costs far more than its
                              // issue slot (MI355X_MICROARCH.md): two v_add_f32 are 2-3.5 us per forward chain faster (round 5)
#endif
      const v2f_t rv = NEAT_HC_PK_BIAS ? pk_add_f32(av, bv) : v2f_t{av.x + bv.x, av.y + bv.y};
      const unsigned pr = pk_relu16(pack2(rv.x, rv.y));
      ph[j >> 1] = pr;
      if (SAVE && !(NEAT_HC_ABLATE & 8)) relu_mask_push(mbits, pr);
      if (j != 3) return;
      const uint2 vh = make_uint2(ph[0], ph[1]);
      if (!(NEAT_HC_ABLATE & 16)) *reinterpret_cast<uint2*>(plq + q * (BP * 16)) = vh;
      if (SAVE) {
        if (NEAT_HC_WIDE) {
          if (q & 1) hc_store16<NEAT_HC_NT_FWD != 0>(reinterpret_cast<char*>(phout) + (size_t)(q - 1) * ldp16 + pgw, hc_octet(vprev, vh));
          else vprev = vh;
        } else if (!(NEAT_HC_ABLATE & 4)) hc_store8<NEAT_HC_NT_FWD != 0>(reinterpret_cast<char*>(phout) + (size_t)q * ldp16 + pg, vh);
#ifndef NEAT_HC_NT_MASK
#define NEAT_HC_NT_MASK 1     // mask words: non-temporal stores (forward chain) and loads (backward chain); round 5: -2..3 us per backward chain
#endif
        if (q == 3 && !(NEAT_HC_ABLATE & 8)) {
          if (NEAT_HC_NT_MASK) __builtin_nontemporal_store(relu_mask_word(mbits), reinterpret_cast<unsigned*>(reinterpret_cast<char*>(pmout) + pmoff));
          else *reinterpret_cast<unsigned*>(reinterpret_cast<char*>(pmout) + pmoff) = relu_mask_word(mbits);
          mbits = 0;
        }
      }
    };
    auto set_pend = [&](int l, int dst, int t) {        // layer l (0..3) writes buffer dst and hid[l + 1] / mask[l + 1]
      pboff = (unsigned)l * 1024u;
      plq = quad[dst] + t * 512;
      pg = gquad + (unsigned)t * 512u;
      pgw = goct + (unsigned)t * 512u;
      phout = a.hid[l + 1]; pmout = a.mask[l + 1]; pmoff = mlane + (unsigned)t * 2048u;
    };
    auto none = [](int) {};
    auto seed = [&](f32x16& ac, int t) {                 // the small-input part of lin0
#pragma unroll
      for (int r = 0; r < 16; ++r) ac[r] = 0.0f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const uint4 bs = *reinterpret_cast<const uint4*>(frag[2] + t * 512 + ks * C::KSTEP);
        ac = NEAT_MFMA16(*reinterpret_cast<const bf16x8*>(&w[16 + ks]), *reinterpret_cast<const bf16x8*>(&bs), ac, 0, 0, 0);
      }
    };
#define HC_EPI(ACC_) [&](int e) { epi_elem(ACC_, e); }
    // layer LL: buffer SRC_ -> buffer DST_; its last tile refills the weight registers from WN_ (KSN_ k-steps per row tile)
#define HC_LAYER(LL, SRC_, DST_, WN_, KSN_)                                                                                        \
    for (int pp = 0; pp < np; ++pp) {                                                                                             \
      const int t0 = 2 * pp;                                                                                                      \
      if (LL == 0) seed(acc[0], t0);                                                                                              \
      if (LL == 0 && pp == 0) {                                                                                                   \
        hc_stage<16, false, false, 16>(frag[SRC_] + t0 * 512, w, acc[0], nullptr, 0u, none);                                       \
      } else {                                                                                                                    \
        if (pp == 0) set_pend(LL - 1, SRC_, nt - 1); else set_pend(LL, DST_, t0 - 1);                                              \
        hc_stage<16, LL != 0, false, 16>(frag[SRC_] + t0 * 512, w, acc[0], nullptr, 0u, HC_EPI(acc[1]));                           \
      }                                                                                                                           \
      if (np == 1 && !(NEAT_HC_ABLATE & 128)) __syncthreads();                                                                    \
      if (LL == 0) seed(acc[1], t0 + 1);                                                                                          \
      set_pend(LL, DST_, t0);                                                                                                     \
      if (pp == np - 1) hc_stage<16, LL != 0, true, 16>(frag[SRC_] + (t0 + 1) * 512, w, acc[1], WN_, w_off(KSN_), HC_EPI(acc[0])); \
      else hc_stage<16, LL != 0, false, 16>(frag[SRC_] + (t0 + 1) * 512, w, acc[1], nullptr, 0u, HC_EPI(acc[0]));                  \
      if (!(NEAT_HC_ABLATE & 128)) __syncthreads();                                                                               \
    }
#define HC_LAYER3 HC_LAYER(3, 1, 0, a.Wp[0], 20)
    HC_LAYER(0, 0, 1, a.Wp[1], 16)       // lin0: XA (+ small part from S) -> XB
    HC_STAMP(2);
    HC_LAYER(1, 1, 0, a.Wp[2], 16)       // lin1: XB -> XA
    HC_STAMP(3);
    HC_LAYER(2, 0, 1, a.Wp[3], 16)       // lin2: XA -> XB
    HC_STAMP(4);
    HC_LAYER3                            // lin3: XB -> XA; refills with the next batch's lin0 (its 16 feature k-steps)
    set_pend(3, 0, nt - 1);
    hc_drain(HC_EPI(acc[1]));
    __syncthreads();
    HC_STAMP(5);
#undef HC_LAYER3
#undef HC_LAYER
#undef HC_EPI
    // ---- lin4 (3 / 6 rows): K split over the waves (2 k-steps each), partial sums through the B-role buffer (idle since lin3's
    // last read), then bias (+ sigmoid)
    {
      float* red = reinterpret_cast<float*>(hclds + C::XB);       // [8 waves][8 rows][BP] = 32 KiB
      const unsigned char* fr = frag[0] + (unsigned)(2 * wave) * C::KSTEP;
      for (int t = 0; t < nt; ++t) {
        f32x16 accs;
#pragma unroll
        for (int r = 0; r < 16; ++r) accs[r] = 0.0f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const uint4 bh = *reinterpret_cast<const uint4*>(fr + j * C::KSTEP + t * 512);
          accs = NEAT_MFMA16(*reinterpret_cast<const bf16x8*>(&ow[j]), *reinterpret_cast<const bf16x8*>(&bh), accs, 0, 0, 0);
        }
        // rows 0..3 in lanes 0-31 (registers 0..3), rows 4..7 in lanes 32-63
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(wave * 8 + 4 * hi + r) * BP + t * 32 + (lane & 31)] = accs[r];
      }
      __syncthreads();
      int tf = tid;
      asm volatile("" : "+v"(tf));
      for (int idx = tf; idx < NOUT * BP; idx += C::THREADS) {
        const int n = idx >> 7, pp = idx & 127;
        if (pp < npts) {
          float v = biasl[4 * 256 + n];
#pragma unroll
          for (int wv = 0; wv < 8; ++wv) v += red[(wv * 8 + n) * BP + pp];
          if (HEAD == 0) v = 1.0f / (1.0f + __expf(-v));
          a.out[(size_t)n * a.ldp + p0 + pp] = v;
        }
      }
    }
    __syncthreads();
    HC_STAMP(6);
#if NEAT_HC_TIMING
    if (blockIdx.x == 0 && it < 3 && (tid == 0 || tid == 448))
      printf("hc fwd head %d wave %d batch %d (np %d): load %llu lin0 %llu lin1 %llu lin2 %llu lin3 %llu lin4 %llu cycles\n", HEAD, wave, it, np,
             stamp[1] - stamp[0], stamp[2] - stamp[1], stamp[3] - stamp[2], stamp[4] - stamp[3], stamp[5] - stamp[4], stamp[6] - stamp[5]);
#endif
  }
}

// ---------------------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------------------

__device__ __forceinline__ float hc_cot_scale_of(const float* slot) {      // kernels.hpp: cot_scale_of (that header is not part of this unit)
  if (!slot) return 1.0f;
  const float m = *slot;
  if (!(m > 0.0f) || !(m < 3.0e38f)) return 1.0f;
  int e;
  (void)frexpf(m, &e);
  e = e < -100 ? -100 : (e > 100 ? 100 : e);
  return ldexpf(1.0f, 1 - e);
}

template <int HEAD>
__global__ __launch_bounds__(512, 2) void head_bwd_chain_kernel(HeadBwdArgs a, int npairs) {
  typedef HC C;
  constexpr int BP = C::BP;
  extern __shared__ __attribute__((aligned(16))) unsigned char hclds[];
  const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int idx = tid; idx < C::SPL / 16; idx += C::THREADS) reinterpret_cast<uint4*>(hclds + C::S)[idx] = make_uint4(0u, 0u, 0u, 0u);
  const float rho = a.accumulate ? hc_cot_scale_of(a.rho_num) / hc_cot_scale_of(a.rho_den) : 1.0f;

  const int nvp = npairs < a.nvalid ? npairs : a.nvalid;
  // ---- pairs without ray samples: zero cotangents
  for (int pair = nvp + blockIdx.x; pair < npairs; pair += gridDim.x) {
    const unsigned p0 = (unsigned)pair * 64u;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + i * C::THREADS, oct = idx >> 6, pp = idx & 63;
      const unsigned off = ((unsigned)oct * (unsigned)a.ldp + p0 + pp) * 16u;
#pragma unroll
      for (int l = 0; l < 4; ++l) *reinterpret_cast<uint4*>(reinterpret_cast<char*>(a.ab[l]) + off) = make_uint4(0u, 0u, 0u, 0u);
      if (!a.accumulate) *reinterpret_cast<uint4*>(reinterpret_cast<char*>(a.featc) + off) = make_uint4(0u, 0u, 0u, 0u);
    }
    for (int idx = tid; idx < a.srows * 64; idx += C::THREADS) a.sc[(size_t)(idx >> 6) * a.ldp + p0 + (idx & 63)] = 0.0f;
  }

  uint4 w[16], wtop;
  auto w_off = [&](int tile, int KS) -> unsigned {
    unsigned v = (unsigned)((tile * KS) * 64 + lane) * 16u;
    asm volatile("" : "+v"(v));
    return v;
  };
  wtop = hc_ldg(a.Wt[4], w_off(wave, 4));
  {
    const unsigned o3 = w_off(wave, 16);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) w[ks] = hc_ldg(a.Wt[3], o3 + ks * 1024);
  }
  // the row tile of the small-input cotangents this wave computes at the end of a batch: (row tile 8 + wave / 4, point tile wave % 4)
  const int srt = 8 + (wave >> 2), stile = wave & 3;
  const bool s_live = (srt - 8) * 32 < a.srows;
  const unsigned char* frag[3];
  unsigned char* quad[2];
  {
    const unsigned fo = (unsigned)(hi * BP + (lane & 31)) * 16u, qo = (unsigned)((4 * wave) * BP + (lane & 31)) * 16u + 8u * hi;
    unsigned b0 = C::XA + fo, b1 = C::XB + fo, b2 = C::S + fo, q0 = C::XA + qo, q1 = C::XB + qo;
    asm volatile("" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(q0), "+v"(q1));
    frag[0] = hclds + b0; frag[1] = hclds + b1; frag[2] = hclds + b2;
    quad[0] = hclds + q0; quad[1] = hclds + q1;
  }
  __syncthreads();

  const HcWork work(nvp, gridDim.x);
  for (int it = 0;; ++it) {
    int pair0;
    const int np = work.batch(it, blockIdx.x, pair0);
    if (np == 0) break;
    const int nt = 2 * np, npts = 64 * np;
    const unsigned p0 = (unsigned)pair0 * 64u;
    unsigned ldp16 = (unsigned)a.ldp * 16u;      // (scalar: the row-quad part of a store address goes into the scalar base)
    asm volatile("" : "+s"(ldp16));
    const unsigned gquad = ((unsigned)(4 * wave) * (unsigned)a.ldp + p0 + (unsigned)(lane & 31)) * 16u + 8u * hi;
    const unsigned goct = ((unsigned)(4 * wave + hi) * (unsigned)a.ldp + p0 + (unsigned)(lane & 31)) * 16u;
    const unsigned mlane = (((p0 >> 5) * 8u + (unsigned)wave) * 64u + (unsigned)lane) * 4u;
    int tb = tid;
    asm volatile("" : "+v"(tb));
#if NEAT_HC_TIMING
    unsigned long long stamp[8];
#endif
    HC_STAMP(0);
    // ---- the output cotangent (one octet per point) -> S
    if (tb < npts) reinterpret_cast<uint4*>(hclds + C::S)[tb] = hc_ldg_in(a.top, (p0 + (unsigned)tb) * 16u);
    __syncthreads();

    f32x16 acc[2];
    unsigned char* plq = quad[0];
    unsigned pg = 0;
    u16* phout = nullptr;
    unsigned pmask = 0, nmask = 0;        // mask words of the pending stage / of the stage being computed (requested a stage ahead)
    uint2 pf[4], nf[4];                   // accumulate: the feature cotangent already there, same schedule
    unsigned ph[2];
    uint2 vprev = make_uint2(0u, 0u);   // wide stores (hc_octet): the even quad waits for its odd neighbour
    unsigned pgw = 0;
    auto ldmask = [&](int l, int t) -> unsigned {
      if (NEAT_HC_NT_MASK) return __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(a.mask[l]) + (mlane + (unsigned)t * 2048u)));
      return *reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(a.mask[l]) + (mlane + (unsigned)t * 2048u));
    };
    // masked epilogue: quad -> LDS buffer + ab array.  Per pair: rounding, then the pair's two mask bits (the sign bits of the mask
    // word's halves) clear the packed values
    auto epi_mask = [&](const f32x16& ap, int e) {
      const int q = e >> 2, j = e & 3;
      if ((j & 1) == 0) return;
      ph[j >> 1] = relu_mask_apply(pmask, pack2(ap[e - 1], ap[e]));
      if (j != 3) return;
      const uint2 vh = make_uint2(ph[0], ph[1]);
      *reinterpret_cast<uint2*>(plq + q * (BP * 16)) = vh;
      if (NEAT_HC_WIDE) {
        if (q & 1) hc_store16<NEAT_HC_NT_BWD != 0>(reinterpret_cast<char*>(phout) + (size_t)(q - 1) * ldp16 + pgw, hc_octet(vprev, vh));
        else vprev = vh;
      } else hc_store8<NEAT_HC_NT_BWD != 0>(reinterpret_cast<char*>(phout) + (size_t)q * ldp16 + pg, vh);
    };
    // feature-cotangent epilogue: quad -> featc (the second head adds, in the first head's scale)
    auto epi_feat = [&](const f32x16& ap, int e) {
      const int q = e >> 2, j = e & 3;
      if ((j & 1) == 0) return;
      v2f_t rv = {ap[e - 1], ap[e]};
      if (HEAD == 1) {
        const unsigned wd = (j & 2) ? pf[q].y : pf[q].x;
        rv = v2f_t{fmaf(rv.x, rho, bf_lo(wd)), fmaf(rv.y, rho, bf_hi(wd))};      // (two v_fma_f32, not one v_pk_fma_f32: see NEAT_HC_PK_BIAS)
      }
      ph[j >> 1] = pack2(rv.x, rv.y);
      if (j != 3) return;
#ifndef NEAT_HC_NT_FEATC
#define NEAT_HC_NT_FEATC 0
#endif
      hc_store8<NEAT_HC_NT_FEATC != 0>(reinterpret_cast<char*>(a.featc) + (size_t)q * ldp16 + pg, make_uint2(ph[0], ph[1]));
    };
    auto set_pend = [&](int l, int dst, int t) {        // the stage that produced the cotangent ab[l] (masked by mask[l + 1]) into buffer dst
      plq = quad[dst] + t * 512;
      pg = gquad + (unsigned)t * 512u;
      pgw = goct + (unsigned)t * 512u;
      phout = a.ab[l];
    };
    auto ldfeat = [&](int t) {
      if (HEAD == 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          nf[q] = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(a.featc) + (size_t)q * ldp16 + (gquad + (unsigned)t * 512u));
      }
    };
    HC_STAMP(1);
    // ---- lin4^T: one k-step per tile (the output cotangent has 3 / 6 rows), not pipelined: S -> XA, ab[3]
    {
      unsigned m4[4];               // all masks first: one HBM latency, not one per tile
#pragma unroll
      for (int t = 0; t < 4; ++t) m4[t] = t < nt ? ldmask(4, t) : 0u;
      auto top_mma = [&](f32x16& ac, int t) {
        const uint4 bs = *reinterpret_cast<const uint4*>(frag[2] + t * 512);
#pragma unroll
        for (int r = 0; r < 16; ++r) ac[r] = 0.0f;
        ac = NEAT_MFMA16(*reinterpret_cast<const bf16x8*>(&wtop), *reinterpret_cast<const bf16x8*>(&bs), ac, 0, 0, 0);
      };
      top_mma(acc[0], 0);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (t < nt) {
          if (t + 1 < nt) top_mma(acc[(t + 1) & 1], t + 1);       // the next tile's MFMA is in flight during this tile's epilogue
          pmask = m4[t];
          set_pend(3, 0, t);
          hc_drain([&](int e) { epi_mask(acc[t & 1], e); });
        }
      }
    }
    __syncthreads();
#define HB_MASK(ACC_) [&](int e) { epi_mask(ACC_, e); }
#define HB_FEAT(ACC_) [&](int e) { epi_feat(ACC_, e); }
    // step for lin{LW}^T, LW = 3, 2, 1: buffer SRC_ -> buffer DST_, masked by mask[LW], cotangent array ab[LW - 1]; the refill after
    // the last tile takes row tile WT_ of pack WN_
#define HB_LAYER(LW, SRC_, DST_, WN_, WT_)                                                                                         \
    for (int pp = 0; pp < np; ++pp) {                                                                                             \
      const int t0 = 2 * pp;                                                                                                      \
      nmask = ldmask(LW, t0);                                                                                                     \
      if (LW == 3 && pp == 0) {                                                                                                   \
        hc_stage<16, true, false, 16>(frag[SRC_] + t0 * 512, w, acc[0], nullptr, 0u, [](int) {});                                  \
      } else {                                                                                                                    \
        if (pp == 0) set_pend(LW, SRC_, nt - 1); else set_pend(LW - 1, DST_, t0 - 1);                                              \
        hc_stage<16, true, false, 16>(frag[SRC_] + t0 * 512, w, acc[0], nullptr, 0u, HB_MASK(acc[1]));                             \
      }                                                                                                                           \
      pmask = nmask;                                                                                                              \
      if (np == 1) __syncthreads();                                                                                               \
      nmask = ldmask(LW, t0 + 1);                                                                                                 \
      set_pend(LW - 1, DST_, t0);                                                                                                 \
      if (pp == np - 1) hc_stage<16, true, true, 16>(frag[SRC_] + (t0 + 1) * 512, w, acc[1], WN_, w_off(WT_, 16), HB_MASK(acc[0])); \
      else hc_stage<16, true, false, 16>(frag[SRC_] + (t0 + 1) * 512, w, acc[1], nullptr, 0u, HB_MASK(acc[0]));                    \
      pmask = nmask;                                                                                                              \
      __syncthreads();                                                                                                            \
    }
    HC_STAMP(2);
    HB_LAYER(3, 0, 1, a.Wt[2], wave)      // lin3^T: XA -> XB, ab[2]
    HC_STAMP(3);
    HB_LAYER(2, 1, 0, a.Wt[1], wave)      // lin2^T: XB -> XA, ab[1]
    HB_LAYER(1, 0, 1, a.Wt[0], wave)      // lin1^T: XA -> XB, ab[0]
    HC_STAMP(4);
    // ---- lin0^T, feature rows: XB -> featc (no LDS output); the last tile's refill fetches the small-input rows' tile
    for (int pp = 0; pp < np; ++pp) {
      const int t0 = 2 * pp;
      ldfeat(t0);
      if (pp == 0) {
        set_pend(0, 1, nt - 1);
        hc_stage<16, true, false, 16>(frag[1] + t0 * 512, w, acc[0], nullptr, 0u, HB_MASK(acc[1]));
      } else {
        pg = gquad + (unsigned)(t0 - 1) * 512u;
        hc_stage<16, true, false, 16>(frag[1] + t0 * 512, w, acc[0], nullptr, 0u, HB_FEAT(acc[1]));
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) pf[q] = nf[q];
      if (pp == 0) __syncthreads();        // lin1^T's last tile (the epilogue just issued) is read by this layer's last stage
      ldfeat(t0 + 1);
      pg = gquad + (unsigned)t0 * 512u;
      if (pp == np - 1) hc_stage<16, true, true, 16>(frag[1] + (t0 + 1) * 512, w, acc[1], a.Wt[0], w_off(s_live ? srt : 8, 16), HB_FEAT(acc[0]));
      else hc_stage<16, true, false, 16>(frag[1] + (t0 + 1) * 512, w, acc[1], nullptr, 0u, HB_FEAT(acc[0]));
#pragma unroll
      for (int q = 0; q < 4; ++q) pf[q] = nf[q];
      // (no other barrier: this layer writes nothing to LDS)
    }
    pg = gquad + (unsigned)(nt - 1) * 512u;
    hc_drain(HB_FEAT(acc[1]));
#undef HB_LAYER
#undef HB_MASK
#undef HB_FEAT
    HC_STAMP(5);
    // ---- lin0^T, small-input rows (packed rows 256 ..): one (row tile, point tile) unit per wave, fp32 rows out
    if (s_live && stile < nt) {
      hc_stage<16, true, true, 16>(frag[1] + stile * 512, w, acc[0], a.Wt[3], w_off(wave, 16), [](int) {});
      const int rbase = (srt - 8) * 32 + 4 * hi;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int r = rbase + 8 * (e >> 2) + (e & 3);
        if (r < a.srows) a.sc[(size_t)r * a.ldp + p0 + stile * 32 + (lane & 31)] = acc[0][e];
      }
    } else {
      const unsigned o3 = w_off(wave, 16);
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) w[ks] = hc_ldg(a.Wt[3], o3 + ks * 1024);
    }
    __syncthreads();
    HC_STAMP(6);
#if NEAT_HC_TIMING
    if (blockIdx.x == 0 && it < 3 && tid == 0)
      printf("hc bwd head %d batch %d (np %d): load %llu lin4T %llu lin3T %llu lin2T+lin1T %llu lin0T %llu small %llu cycles\n", HEAD, it, np,
             stamp[1] - stamp[0], stamp[2] - stamp[1], stamp[3] - stamp[2], stamp[4] - stamp[3], stamp[5] - stamp[4], stamp[6] - stamp[5]);
#endif
  }
}

}  // namespace neat

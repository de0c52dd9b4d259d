// Split-precision FORWARD chains (precision NEAT_F16X3): every product of the three forward chains -- SDF primal (rend_a :78-96),
// SDF adjoint = the normals (:121-127), the two heads (:139-255) -- is evaluated as  hi*hi + lo*hi + hi*lo  on 16-bit hi/lo splits
// of BOTH operands (three v_mfma_f32_32x32x16 per 16 k, fp32 accumulate): ~22 mantissa bits, which is what north_star's 1e-4 needs.
// The backward pass of that precision is the plain 16-bit build's and reads the HI planes only: a hi plane is exactly the array the
// 16-bit build saves (f16(h)), the lo plane f16(h - hi) exists on chip and -- only where a later forward kernel needs it -- in HBM
// (h_1..h_8 for the adjoint chain's phi', the 256 feature rows for the heads).  scripts/precision_emul.py: on the oracle, 3-product
// forward + 16-bit backward gives the same gradient error as 3-product everything.
//
// One design for the three kernels (why fused: with 3 MFMAs per product the matrix time of a fused chain covers its epilogue's
// vector work, while a streamed layer would move twice the bytes):
//   * persistent 8-wave workgroup per CU, batches of 64 points = 2 tiles of 32; wave w owns output rows 32w .. 32w+31;
//   * activations ping-pong between two LDS buffers, each a hi plane and a lo plane in the octet-major layout of kernels_bf16.hpp
//     (2 x 2 x 32 KiB), small inputs (PE rows / the heads' few non-feature rows) in a third region;
//   * the weight slice of a layer lives in registers as 16 hi + 16 lo A fragments (128 VGPRs, ONE set): the layer's last tile
//     refills each slot with the next layer's fragment right after the slot's last MFMA (as sdf_adjoint_w64_kernel does);
//   * stage pipeline as in kernels_fused.hpp: a stage = the 16 k-steps x 3 MFMAs of one (layer, tile), the epilogue of the PREVIOUS
//     stage's accumulators is issued element by element behind the k-steps; one workgroup barrier per stage.
#pragma once
#include "bf16_common.hpp"
#include "fused_launch.hpp"
#include <type_traits>

namespace neat {

#ifndef NEAT_X3_NT
#define NEAT_X3_NT 1          // saved planes of the split-precision chains (read again only by later passes) leave with non-temporal stores, the
                              // adjoint chain's saved-activation quads arrive with non-temporal loads: the weight fragments every batch
                              // re-reads stay in the XCD's L2 (round 5; as NEAT_F6_NT / NEAT_ADJ_NT_LOAD in kernels_fused.hpp)
#endif
__device__ __forceinline__ void x3_store8(void* p, uint2 v) {
  typedef unsigned long long u64_t;
  if (NEAT_X3_NT) __builtin_nontemporal_store(__builtin_bit_cast(u64_t, v), reinterpret_cast<u64_t*>(p));
  else *reinterpret_cast<uint2*>(p) = v;
}
__device__ __forceinline__ uint2 x3_load8(const void* p) {
  typedef unsigned long long u64_t;
  if (NEAT_X3_NT) return __builtin_bit_cast(uint2, __builtin_nontemporal_load(reinterpret_cast<const u64_t*>(p)));
  return *reinterpret_cast<const uint2*>(p);
}

struct X3 {
  static constexpr int BP = X3_BATCH, THREADS = 512;
  static constexpr int XPL = 32 * BP * 16;              // one plane of an activation buffer [32 octets][BP][16 B]
  static constexpr int SPL = 8 * BP * 16;               // one plane of the small-input region (K padded to 64)
  static constexpr int XA = 0, XB = 2 * XPL, S = 4 * XPL;       // hi plane at the offset, lo plane XPL (SPL) behind it
  static constexpr int BIAS = S + 2 * SPL;
  static constexpr int BIAS_FLOATS = 9 * 256 + 8;
  static constexpr int RED = BIAS + BIAS_FLOATS * 4;    // [8 waves][BP] partial sums of the sdf row
  static constexpr int LDS = RED + 8 * BP * 4;          // 158 752 B of the CU's 160 KiB
  static constexpr int KSTEP = 2 * BP * 16;             // bytes between the k-steps of a fragment column
};

// hi / lo split of a pair of fp32 values: hi = 16-bit(v) round-to-nearest, lo = 16-bit(v - hi)
__device__ __forceinline__ void x3_split2(float v0, float v1, unsigned& hi, unsigned& lo) {
  hi = pack2(v0, v1);
  lo = pack2(v0 - bf_lo(hi), v1 - bf_hi(hi));
}

// Per-lane LDS bases (opaque to the compiler; every access is base + compile-time immediate, see F6Lane)
struct X3Lane {
  const unsigned char* frag[3];      // fragment reads of XA / XB / S:  region + (hi * BP + (lane & 31)) * 16
  unsigned char* quad[2];            // accumulator-quad writes to XA / XB: region + (4 wave * BP + (lane & 31)) * 16 + 8 hi
  const unsigned char* bias;         // BIAS + (32 wave + 4 hi) * 4
  unsigned gquad;                    // HBM quad offset of this lane for the batch: ((4 wave) ldp + p0 + (lane & 31)) * 16 + 8 hi
  unsigned ldp16;                    // ldp * 16 (opaque per batch)
};

__device__ __forceinline__ uint4 x3_ldg(const void* base, unsigned off) {
  return *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(base) + off);
}

// One stage: KS k-steps (3 MFMAs each) of `acc` over the fragment column `fr` (hi plane; lo plane LO bytes behind), with epi(e)
// called 16 / KS times per k-step for e = 0 .. 15.  ZERO: the accumulator starts at zero.  ROLL: slot ks of the weight registers is
// refilled with the next layer's fragment (wnh / wnl = its hi / lo packs + this lane's offset, NKS slots) right after its MFMAs;
// slots KS .. NKS-1 (free in this layer) are requested up front.
#ifndef NEAT_X3_CHAINS
#define NEAT_X3_CHAINS 1      // independent accumulator chains the 3 MFMAs of a k-step rotate through (summed after the last k-step)
#endif
#ifndef NEAT_X3_SGB
#define NEAT_X3_SGB 0         // > 0: sched_group_barrier pattern per k-step with this many VALU instructions per MFMA (probe)
#endif
#ifndef NEAT_X3_TIMING
#define NEAT_X3_TIMING 0      // probe builds only: workgroup 0 prints the cycle counts of its first batches' phases (SDF primal chain)
#endif
#if NEAT_X3_TIMING
#define X3_STAMP(i) do { if (blockIdx.x == 0 && nb_done < 3) { __builtin_amdgcn_s_waitcnt(0); stamp[i] = __builtin_readcyclecounter(); } } while (0)
#else
#define X3_STAMP(i) do { } while (0)
#endif
#ifndef NEAT_X3_HALF
#define NEAT_X3_HALF 1        // the ragged last round of the two SDF chains as half batches (0: whole batches, as in round 3)
#endif
#ifndef NEAT_X3_ABLATE
#define NEAT_X3_ABLATE 0      // probe builds only (results are WRONG): 1 = no epilogue, 2 = no MFMAs
#endif
template <int KS, int LO, bool ZERO, bool ROLL, int NKS, class Epi>
__device__ __forceinline__ void x3_stage(const unsigned char* fr, uint4 (&wh)[16], uint4 (&wl)[16], f32x16& acc,
                                         const void* wnh, const void* wnl, unsigned woff, Epi&& epi) {
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  constexpr int NC = NEAT_X3_CHAINS;
  f32x16 ch[NC];                 // chain 0 continues `acc` (ZERO: starts at zero), the others start at zero
  bool started[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) started[c] = false;
  if (!ZERO) { ch[0] = acc; started[0] = true; }
  uint4 bh = *reinterpret_cast<const uint4*>(fr), bl = *reinterpret_cast<const uint4*>(fr + LO);
  if (ROLL) {
#pragma unroll
    for (int ks = KS; ks < NKS; ++ks) { wh[ks] = x3_ldg(wnh, woff + ks * 1024); wl[ks] = x3_ldg(wnl, woff + ks * 1024); }
  }
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    uint4 nh = bh, nl = bl;
    if (ks + 1 < KS) {
      nh = *reinterpret_cast<const uint4*>(fr + (ks + 1) * X3::KSTEP);
      nl = *reinterpret_cast<const uint4*>(fr + LO + (ks + 1) * X3::KSTEP);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int c = (3 * ks + i) % NC;
      const uint4& wa = (i == 1) ? wl[ks] : wh[ks];
      const uint4& bb = (i == 2) ? bl : bh;
      if (NEAT_X3_ABLATE != 2)
        ch[c] = NEAT_MFMA16(*reinterpret_cast<const bf16x8*>(&wa), *reinterpret_cast<const bf16x8*>(&bb), started[c] ? ch[c] : zero, 0, 0, 0);
      else if (!started[c]) { ch[c] = zero; ch[c][0] = __uint_as_float(wa.x ^ bb.x); }
      started[c] = true;
    }
    if (ROLL && ks < NKS) { wh[ks] = x3_ldg(wnh, woff + ks * 1024); wl[ks] = x3_ldg(wnl, woff + ks * 1024); }
    if (NEAT_X3_ABLATE != 1) {
#pragma unroll
      for (int e = ks * (16 / KS); e < (ks + 1) * (16 / KS); ++e) epi(e);
    }
#if NEAT_X3_SGB
    // ask the scheduler for an even spread of the k-step's other work behind its three MFMAs: (1 MFMA, up to NEAT_X3_SGB VALU, 1 LDS
    // read, 1 global load, 1 LDS / global write) x 3
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, NEAT_X3_SGB, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x240, 1, 0);
    }
#endif
    __builtin_amdgcn_sched_barrier(0);
    bh = nh; bl = nl;
  }
  acc = ch[0];
#pragma unroll
  for (int c = 1; c < NC; ++c)
    if (started[c]) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] += ch[c][r];
    }
}
// drain: the epilogue alone
template <class Epi> __device__ __forceinline__ void x3_drain(Epi&& epi) {
#pragma unroll
  for (int e = 0; e < 16; ++e) { epi(e); if ((e & 3) == 3) __builtin_amdgcn_sched_barrier(0); }
}

// ---------------------------------------------------------------------------------------------------------------
// SDF primal chain: E (PE rows, fp32, written by posenc6_kernel with libm's sin / cos) -> lin0 .. lin8.
//   VALUES: only the clamped sdf leaves the chip (sampler queries);  otherwise: h_1..h_8 (hi to a.h[l] -- the arrays the 16-bit
//   backward reads -- and lo to a.hlo[l]), the 256 feature rows (a.feat / a.featlo) and the raw sdf row.
// FusedArgs: Wp = hi packs, Wlo = lo packs, KS as in the 16-bit build; E is an INPUT here.
// ---------------------------------------------------------------------------------------------------------------
template <bool VALUES>
__global__ __launch_bounds__(512, 2) void sdf_chain_x3_kernel(FusedArgs a, int nbatches) {
  typedef X3 C;
  constexpr int BP = C::BP, LO = C::XPL, SLO = C::SPL;
  if (a.gate && *a.gate != a.gate_value) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char x3lds[];
  float* biasl = reinterpret_cast<float*>(x3lds + C::BIAS);
  float* red = reinterpret_cast<float*>(x3lds + C::RED);
  const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr bool SAVE = !VALUES;

  for (int idx = tid; idx < 8 * 256; idx += C::THREADS) {
    const int l = idx >> 8, n = idx & 255;
    float v = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) if (l == k && n < (k == 3 ? 217 : 256)) v = a.bias[k][n];
    biasl[idx] = v * SOFTPLUS_C;             // hidden layers: pre-scaled for the softplus epilogue
  }
  for (int n = tid; n < 257; n += C::THREADS) {
    int bi = n + a.bias8_rot; if (bi >= a.bias8_n) bi -= a.bias8_n;
    biasl[8 * 256 + n] = VALUES ? (n == 0 ? a.bias[8][0] : 0.0f) : a.bias[8][bi];
  }
  // small-input region: rows 39..63 of both planes stay zero for the whole launch
  for (int idx = tid; idx < 2 * C::SPL / 16; idx += C::THREADS) reinterpret_cast<uint4*>(x3lds + C::S)[idx] = make_uint4(0u, 0u, 0u, 0u);

  uint4 wh[16], wl[16];
  // per-lane offset of this wave's fragments inside a pack [tile][KS][64 lanes] x 16 B (dead row tiles re-read tile 0)
  auto w_off = [&](int KS, int N) -> unsigned {
    const int tile = wave * 32 < N ? wave : 0;
    unsigned v = (unsigned)((tile * KS) * 64 + lane) * 16u;
    asm volatile("" : "+v"(v));
    return v;
  };
  X3Lane L;
  {
    const unsigned fo = (unsigned)(hi * BP + (lane & 31)) * 16u, qo = (unsigned)((4 * wave) * BP + (lane & 31)) * 16u + 8u * hi;
    unsigned b0 = C::XA + fo, b1 = C::XB + fo, b2 = C::S + fo, q0 = C::XA + qo, q1 = C::XB + qo, bb = C::BIAS + (unsigned)(32 * wave + 4 * hi) * 4u;
    asm volatile("" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(q0), "+v"(q1), "+v"(bb));
    L.frag[0] = x3lds + b0; L.frag[1] = x3lds + b1; L.frag[2] = x3lds + b2;
    L.quad[0] = x3lds + q0; L.quad[1] = x3lds + q1; L.bias = x3lds + bb;
  }
  // ---- PE rows of a batch -> small-input region, hi / lo planes: thread = (point, row group): rows g, g + 8, ...  Requested (e_load)
  // for the NEXT batch once skip_fix has made the last use of the region, written (e_store) at the end of the batch: no batch but a
  // workgroup's first waits for HBM before its first MFMA (PREFETCH; values mode only: the training variant has no five registers to
  // spare -- they spill right behind the load, which waits for it there).  lin0's four k-steps of weights arrive through the last
  // layer's rolling refill in both modes.
  constexpr bool PREFETCH = VALUES;
  float en[5];
  auto e_load = [&](int pfirst, bool half) {      // half: only the first 32 points exist for this workgroup
    int te = tid;
    asm volatile("" : "+v"(te));
    const int p = te & (BP - 1), g = te >> 6;
    unsigned pvo = (unsigned)(pfirst + p) * 4u, ldp4 = (unsigned)a.ldp * 4u;
    asm volatile("" : "+v"(pvo), "+v"(ldp4));
    const bool live = !half || p < 32;
#pragma unroll
    for (int jj = 0; jj < 5; ++jj) {
      const int j = g + 8 * jj;
      en[jj] = (j < 39 && live) ? *reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.E) + ((unsigned)j * ldp4 + pvo)) : 0.0f;
    }
  };
  auto e_store = [&]() {
    int te = tid;
    asm volatile("" : "+v"(te));
    const int p = te & (BP - 1), g = te >> 6;
    u16* shi = reinterpret_cast<u16*>(x3lds + C::S);
    u16* slo = reinterpret_cast<u16*>(x3lds + C::S + SLO);
#pragma unroll
    for (int jj = 0; jj < 5; ++jj) {
      const int j = g + 8 * jj;
      if (j < 39) {
        const u16 h = f2bf(en[jj]);
        shi[((j >> 3) * BP + p) * 8 + (j & 7)] = h;
        slo[((j >> 3) * BP + p) * 8 + (j & 7)] = f2bf(en[jj] - bf2f(h));
      }
    }
  };
  if ((int)blockIdx.x < nbatches) {
    const unsigned o0 = w_off(4, 256);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { wh[ks] = x3_ldg(a.Wp[0], o0 + ks * 1024); wl[ks] = x3_ldg(a.Wlo[0], o0 + ks * 1024); }
  }
  __syncthreads();
  // The ragged last round.  nbatches over G workgroups leaves rem = nbatches % G batches for a last round that would keep G - rem
  // workgroups idle for a whole batch; when 2 rem <= G the round is run as 2 rem HALF batches (one tile of 32 points per workgroup:
  // per layer the k-steps, then the epilogue, one barrier -- no pipeline, about half a batch's latency).
  const int G = (int)gridDim.x;
  int nfull = nbatches, nhalf = 0;
  if (NEAT_X3_HALF && nbatches > G) {
    const int rem = nbatches % G;
    if (rem > 0 && 2 * rem <= G) { nfull = nbatches - rem; nhalf = rem; }
  }

#if NEAT_X3_TIMING
  int nb_done = 0;
  unsigned long long stamp[12];
#endif
  auto run_batch = [&](const int p0, const bool more, const int pnext, auto half_c) {
    constexpr bool HALF = decltype(half_c)::value;
    X3_STAMP(0);
    L.gquad = ((unsigned)(4 * wave) * (unsigned)a.ldp + (unsigned)(p0 + (lane & 31))) * 16u + 8u * hi;
    L.ldp16 = (unsigned)a.ldp * 16u;
    asm volatile("" : "+v"(L.ldp16));
    if (!PREFETCH || HALF) { e_load(p0, HALF); e_store(); __syncthreads(); }
    X3_STAMP(1);

    f32x16 acc[2];
    // the bias rows of a quad are read from LDS when its first element comes up (4 registers instead of 16 held over the stage)
    float4 bqv = make_float4(0.f, 0.f, 0.f, 0.f);
    // epilogue element e of accumulator tile `ap` (point tile t) of a layer with N rows: activation, hi / lo split, the quad
    // (4 consecutive rows of one point) goes to LDS buffer DST (both planes) and, SAVE, to the HBM arrays hout / lout
    unsigned ph[2], pl[2];
    float keep = 0.0f;
    auto epi_elem = [&](const f32x16& ap, int e, int t, int lb, bool act, int N, int DST, bool to_lds, bool save, u16* hout, u16* lout) {
      const int q = e >> 2, j = e & 3;
      if (j == 0) bqv = *reinterpret_cast<const float4*>(L.bias + (lb * 256 + 8 * q) * 4);
      const float b = j == 0 ? bqv.x : (j == 1 ? bqv.y : (j == 2 ? bqv.z : bqv.w));
      float r;
      if (act) {
        const float u = fmaf(ap[e], SOFTPLUS_C, b);
        r = (fmaxf(u, 0.0f) + __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(-fabsf(u)))) * 0.0069314718055994531f;
      } else {
        r = ap[e] + b;
      }
      if ((j & 1) == 0) { keep = r; return; }
      x3_split2(keep, r, ph[j >> 1], pl[j >> 1]);
      if (j != 3) return;
      uint2 vh = make_uint2(ph[0], ph[1]), vl = make_uint2(pl[0], pl[1]);
      if (to_lds) {
        *reinterpret_cast<uint2*>(L.quad[DST] + (q * BP + t * 32) * 16) = vh;
        *reinterpret_cast<uint2*>(L.quad[DST] + LO + (q * BP + t * 32) * 16) = vl;
      }
      // (lin3, N = 217: the octet of rows 216..223 = [h216 | PE rows 0..6] is completed and stored by skip_fix below)
      if (save && !(N == 217 && q == 3 && wave == 6)) {
        const unsigned off = (unsigned)q * L.ldp16 + L.gquad + t * 512;
        x3_store8(reinterpret_cast<char*>(hout) + off, vh);
        x3_store8(reinterpret_cast<char*>(lout) + off, vl);
      }
    };
    auto none = [](int) {};
    // skip connection (rend_a :87-88): lin4's input is [h4 (217 rows) | PE (39 rows)] (the 1/sqrt2 is folded into W4).  After lin3's
    // epilogues: rows 224..255 (octets 28..31 of XB) = PE rows 7..38 -- one thread per (point, octet, plane); PE rows 7+8k .. 14+8k
    // straddle PE octets k and k+1 --, and octet 27 = [h216 | PE rows 0..6] (read-modify-write, threads 0..127), which also goes to
    // the saved h4 planes in that form (what lin4's consumers in the backward pass expect in the last octet of the 217-row array).
    auto skip_fix = [&]() {
      int ts = tid;
      asm volatile("" : "+v"(ts));
      const int pp = ts & (BP - 1), k = (ts >> 6) & 3, pln = ts >> 8;
      const uint4* pe = reinterpret_cast<const uint4*>(x3lds + C::S + pln * SLO);
      const uint4 lo4 = pe[k * BP + pp], hi4 = pe[(k + 1) * BP + pp];
      reinterpret_cast<uint4*>(x3lds + C::XB + pln * LO)[(28 + k) * BP + pp] =
          make_uint4((lo4.w >> 16) | (hi4.x << 16), (hi4.x >> 16) | (hi4.y << 16), (hi4.y >> 16) | (hi4.z << 16), (hi4.z >> 16) | (hi4.w << 16));
      if (ts < 2 * BP) {
        const int p2 = ts & (BP - 1), pl2 = ts >> 6;
        const uint4 e = reinterpret_cast<const uint4*>(x3lds + C::S + pl2 * SLO)[p2];
        uint4* dst = reinterpret_cast<uint4*>(x3lds + C::XB + pl2 * LO) + 27 * BP + p2;
        const unsigned h216 = dst->x & 0xFFFFu;
        const uint4 v = make_uint4(h216 | (e.x << 16), (e.x >> 16) | (e.y << 16), (e.y >> 16) | (e.z << 16), (e.z >> 16) | (e.w << 16));
        *dst = v;
        if (SAVE && (!HALF || p2 < 32)) {
          u16* arr = pl2 ? a.hlo[4] : a.h[4];
          *reinterpret_cast<uint4*>(reinterpret_cast<char*>(arr) + ((unsigned)27 * (unsigned)a.ldp + (unsigned)(p0 + p2)) * 16u) = v;
        }
      }
    };

    // One layer = 2 stages.  SRC / DST: input / output LDS buffer (0 = XA, 1 = XB, 2 = S); the epilogue of the previous stage
    // belongs to layer LP (activated, NP rows, written to buffer SRC -- this layer's input buffer is the previous layer's output).
#define X3_LAYER(LCUR, KS_, SRC_, SRCLO_, LIVE_, FIRST_, NKS_, WNH_, WNL_, NOFF_, PREV_EPI0_, CUR_EPI_)                                   \
    if (HALF) {                                                                                                                     \
      x3_stage<KS_, SRCLO_, true, true, NKS_>(L.frag[SRC_], wh, wl, acc[0], WNH_, WNL_, NOFF_, none);                               \
      x3_drain(CUR_EPI_);                                                                                                           \
      __syncthreads();                                                                                                              \
    } else {                                                                                                                        \
      if (LIVE_) x3_stage<KS_, SRCLO_, true, false, 16>(L.frag[SRC_], wh, wl, acc[0], nullptr, nullptr, 0u, PREV_EPI0_);             \
      else x3_drain(PREV_EPI0_);                                                                                                   \
      __syncthreads();                                                                                                              \
      if (LIVE_) x3_stage<KS_, SRCLO_, true, true, NKS_>(L.frag[SRC_] + 512, wh, wl, acc[1], WNH_, WNL_, NOFF_, CUR_EPI_);           \
      else { const unsigned o_ = NOFF_; _Pragma("unroll") for (int ks = 0; ks < NKS_; ++ks) { wh[ks] = x3_ldg(WNH_, o_ + ks * 1024); wl[ks] = x3_ldg(WNL_, o_ + ks * 1024); } x3_drain(CUR_EPI_); } \
      __syncthreads();                                                                                                              \
    }
    // epilogues: E_<l>(tile) = epilogue of layer l's tile
#define X3_EPI(LB_, ACC_, T_, ACT_, N_, DST_, TOLDS_, SAVE_, H_, HL_) [&](int e) { epi_elem(ACC_, e, T_, LB_, ACT_, N_, DST_, TOLDS_, SAVE_, H_, HL_); }
    // lin0: S -> XA
    X3_LAYER(0, 4, 2, SLO, true, true, 16, a.Wp[1], a.Wlo[1], w_off(16, 256), none, X3_EPI(0, acc[0], 0, true, 256, 0, true, SAVE, a.h[1], a.hlo[1]))
    X3_STAMP(2);
    // lin1: XA -> XB
    X3_LAYER(1, 16, 0, LO, true, false, 16, a.Wp[2], a.Wlo[2], w_off(16, 256), X3_EPI(0, acc[1], 1, true, 256, 0, true, SAVE, a.h[1], a.hlo[1]), X3_EPI(1, acc[0], 0, true, 256, 1, true, SAVE, a.h[2], a.hlo[2]))
    X3_STAMP(3);
    // lin2: XB -> XA
    X3_LAYER(2, 16, 1, LO, true, false, 16, a.Wp[3], a.Wlo[3], w_off(16, 217), X3_EPI(1, acc[1], 1, true, 256, 1, true, SAVE, a.h[2], a.hlo[2]), X3_EPI(2, acc[0], 0, true, 256, 0, true, SAVE, a.h[3], a.hlo[3]))
    // lin3 (217 rows): XA -> XB like every other layer (wave 7 multiplies a duplicate of tile 0: its rows 224.. do not exist and are
    // overwritten below; nothing of it is stored), drained, then the PE rows complete lin4's input (skip_fix)
    const bool w7 = wave != 7;
    X3_LAYER(3, 16, 0, LO, true, false, 16, a.Wp[4], a.Wlo[4], w_off(16, 256), X3_EPI(2, acc[1], 1, true, 256, 0, true, SAVE, a.h[3], a.hlo[3]), X3_EPI(3, acc[0], 0, true, 217, 1, true, SAVE && w7, a.h[4], a.hlo[4]))
    if (!HALF) {
      x3_drain(X3_EPI(3, acc[1], 1, true, 217, 1, true, SAVE && w7, a.h[4], a.hlo[4]));
      __syncthreads();
    }
    skip_fix();
    __syncthreads();
    X3_STAMP(4);
    if (PREFETCH && more) e_load(pnext, false);
    // lin4: XB -> XA
    X3_LAYER(4, 16, 1, LO, true, false, 16, a.Wp[5], a.Wlo[5], w_off(16, 256), none, X3_EPI(4, acc[0], 0, true, 256, 0, true, SAVE, a.h[5], a.hlo[5]))
    X3_STAMP(5);
    // lin5: XA -> XB
    X3_LAYER(5, 16, 0, LO, true, false, 16, a.Wp[6], a.Wlo[6], w_off(16, 256), X3_EPI(4, acc[1], 1, true, 256, 0, true, SAVE, a.h[5], a.hlo[5]), X3_EPI(5, acc[0], 0, true, 256, 1, true, SAVE, a.h[6], a.hlo[6]))
    // lin6: XB -> XA
    X3_LAYER(6, 16, 1, LO, true, false, 16, a.Wp[7], a.Wlo[7], w_off(16, 256), X3_EPI(5, acc[1], 1, true, 256, 1, true, SAVE, a.h[6], a.hlo[6]), X3_EPI(6, acc[0], 0, true, 256, 0, true, SAVE, a.h[7], a.hlo[7]))
    // lin7: XA -> XB (h8); its last tile fetches the 256 feature rows of lin8 (save mode; values mode: nothing to prefetch but the
    // macro refills anyway -- from lin8's pack, which exists in both modes)
    X3_LAYER(7, 16, 0, LO, true, false, (VALUES ? 4 : 16), a.Wp[VALUES ? 0 : 8], a.Wlo[VALUES ? 0 : 8], w_off(VALUES ? 4 : 16, 256), X3_EPI(6, acc[1], 1, true, 256, 0, true, SAVE, a.h[7], a.hlo[7]), X3_EPI(7, acc[0], 0, true, 256, 1, true, SAVE, a.h[8], a.hlo[8]))
    // drain: h8's second tile
    if (!HALF) {
      x3_drain(X3_EPI(7, acc[1], 1, true, 256, 1, true, SAVE, a.h[8], a.hlo[8]));
      __syncthreads();
    }
    X3_STAMP(6);
    // ---- lin8: the sdf row, split over the waves' k-steps (2 each) and reduced through LDS
    {
      unsigned so = (unsigned)((((VALUES ? 0 : 8) * 16 + 2 * wave) * 64 + lane) * 16);
      asm volatile("" : "+v"(so));
      uint4 sh[2], sl[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) { sh[j] = x3_ldg(a.Wp[8], so + j * 1024); sl[j] = x3_ldg(a.Wlo[8], so + j * 1024); }
      const unsigned char* fr = L.frag[1] + (unsigned)(2 * wave) * C::KSTEP;
#pragma unroll
      for (int t = 0; t < (HALF ? 1 : 2); ++t) {
        f32x16 accs;
#pragma unroll
        for (int r = 0; r < 16; ++r) accs[r] = 0.0f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const uint4 bh = *reinterpret_cast<const uint4*>(fr + j * C::KSTEP + t * 512);
          const uint4 bl = *reinterpret_cast<const uint4*>(fr + LO + j * C::KSTEP + t * 512);
          accs = NEAT_MFMA16(*reinterpret_cast<const bf16x8*>(&sh[j]), *reinterpret_cast<const bf16x8*>(&bh), accs, 0, 0, 0);
          accs = NEAT_MFMA16(*reinterpret_cast<const bf16x8*>(&sl[j]), *reinterpret_cast<const bf16x8*>(&bh), accs, 0, 0, 0);
          accs = NEAT_MFMA16(*reinterpret_cast<const bf16x8*>(&sh[j]), *reinterpret_cast<const bf16x8*>(&bl), accs, 0, 0, 0);
        }
        if (hi == 0) red[wave * BP + t * 32 + lane] = accs[0];
      }
    }
    if (SAVE) {
      // ---- the 256 feature rows of lin8 (linear): XB -> HBM only
      if (HALF) {
        x3_stage<16, LO, true, true, 4>(L.frag[1], wh, wl, acc[0], a.Wp[0], a.Wlo[0], w_off(4, 256), none);
        x3_drain(X3_EPI(8, acc[0], 0, false, 256, 0, false, true, a.feat, a.featlo));
      } else {
        x3_stage<16, LO, true, false, 16>(L.frag[1], wh, wl, acc[0], nullptr, nullptr, 0u, none);
        x3_stage<16, LO, true, true, 4>(L.frag[1] + 512, wh, wl, acc[1], a.Wp[0], a.Wlo[0], w_off(4, 256), X3_EPI(8, acc[0], 0, false, 256, 0, false, true, a.feat, a.featlo));      // (rolls in lin0 of the next batch)
        x3_drain(X3_EPI(8, acc[1], 1, false, 256, 0, false, true, a.feat, a.featlo));
      }
    }
    __syncthreads();
#undef X3_EPI
#undef X3_LAYER
    int tf = tid;
    asm volatile("" : "+v"(tf));
    if (tf < (HALF ? 32 : BP)) {
      float sv = biasl[8 * 256 + (VALUES ? 0 : 256)];
#pragma unroll
      for (int w = 0; w < 8; ++w) sv += red[w * BP + tf];
      const int p = p0 + tf;
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
    if (PREFETCH && more) e_store();
    __syncthreads();
    X3_STAMP(7);
#if NEAT_X3_TIMING
    if (blockIdx.x == 0 && nb_done < 3 && tid == 0)
      printf("x3 primal save %d batch %d: load %llu lin0 %llu lin1 %llu lin2+3 %llu lin4 %llu lin5-7 %llu lin8+end %llu cycles\n", (int)SAVE, nb_done,
             stamp[1] - stamp[0], stamp[2] - stamp[1], stamp[3] - stamp[2], stamp[4] - stamp[3], stamp[5] - stamp[4], stamp[6] - stamp[5], stamp[7] - stamp[6]);
    ++nb_done;
#endif
  };
  // (the half batch FIRST: behind the loop it would keep its pointers alive through the loop)
  if ((int)blockIdx.x < 2 * nhalf) run_batch((nfull + ((int)blockIdx.x >> 1)) * BP + 32 * ((int)blockIdx.x & 1), false, 0, std::true_type{});
  if (PREFETCH && (int)blockIdx.x < nfull) { e_load(blockIdx.x * BP, false); e_store(); __syncthreads(); }
  for (int batch = blockIdx.x; batch < nfull; batch += G)
    run_batch(batch * BP, batch + G < nfull, (batch + G) * BP, std::false_type{});
}

// ---------------------------------------------------------------------------------------------------------------
// SDF adjoint chain (the normals, autograd.grad at rend_a :121-127) with 3-product arithmetic:
//   u_7 = w8 (.) phi'(h_8);  u_{l-1} = (W_l^T u_l) (.) phi'(h_l), l = 7 .. 1 (l = 4: rows 217.. = PE cotangent of the skip, fp32, no phi');
//   e0 = W_0^T u_0 (39 fp32 rows).   phi'(h) = 1 - exp(-100 h) from h = hi + lo (both planes are read: 2^-11 on h would be 1e-3 on the normals).
// u stays on chip with both planes; SAVE writes the hi planes u_0 .. u_7 the 16-bit backward reads.
// ---------------------------------------------------------------------------------------------------------------
template <bool SAVE>
__global__ __launch_bounds__(512, 2) void sdf_adjoint_x3_kernel(AdjArgs a, int nbatches) {
  typedef X3 C;
  constexpr int BP = C::BP, LO = C::XPL;
  extern __shared__ __attribute__((aligned(16))) unsigned char x3lds[];
  float* seedw = reinterpret_cast<float*>(x3lds + C::BIAS);      // [256]: w8[k] * rs8
  const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int k = tid; k < 256; k += C::THREADS) seedw[k] = a.w8[k] * a.rs8[0];

  uint4 wh[16], wl[16];
  auto w_off = [&](int N) -> unsigned {
    const int tile = wave * 32 < N ? wave : 0;
    unsigned v = (unsigned)((tile * 16) * 64 + lane) * 16u;
    asm volatile("" : "+v"(v));
    return v;
  };
  {
    const unsigned o7 = w_off(256);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) { wh[ks] = x3_ldg(a.Wp[7], o7 + ks * 1024); wl[ks] = x3_ldg(a.Wlo[7], o7 + ks * 1024); }
  }
  X3Lane L;
  {
    const unsigned fo = (unsigned)(hi * BP + (lane & 31)) * 16u, qo = (unsigned)((4 * wave) * BP + (lane & 31)) * 16u + 8u * hi;
    unsigned b0 = C::XA + fo, b1 = C::XB + fo, q0 = C::XA + qo, q1 = C::XB + qo, bb = C::BIAS + (unsigned)(32 * wave + 4 * hi) * 4u;
    asm volatile("" : "+v"(b0), "+v"(b1), "+v"(q0), "+v"(q1), "+v"(bb));
    L.frag[0] = x3lds + b0; L.frag[1] = x3lds + b1; L.frag[2] = x3lds + b0;
    L.quad[0] = x3lds + q0; L.quad[1] = x3lds + q1; L.bias = x3lds + bb;
  }
  __syncthreads();

  // NEAT_X3_ADJ_REVERSE = 1 (probe): batches in the REVERSE of the primal chain's order, so that what the primal launch wrote last
  // (~1.7 rounds' worth of h planes fit the 256 MB Infinity Cache) is read first.  Measured neutral (631 / 642 us against 615 / 693 us
  // in forward order, two runs each on one box): off.
#ifndef NEAT_X3_ADJ_REVERSE
#define NEAT_X3_ADJ_REVERSE 0
#endif
#if NEAT_X3_TIMING
  int nb_done = 0;
  unsigned long long stamp[12];
#endif
  // the ragged last round as half batches: see sdf_chain_x3_kernel
  const int G = (int)gridDim.x;
  int nfull = nbatches, nhalf = 0;
  if (NEAT_X3_HALF && !NEAT_X3_ADJ_REVERSE && nbatches > G) {
    const int rem = nbatches % G;
    if (rem > 0 && 2 * rem <= G) { nfull = nbatches - rem; nhalf = rem; }
  }
  auto run_batch = [&](const int p0, auto half_c) {
    constexpr bool HALF = decltype(half_c)::value;
    X3_STAMP(0);
    L.gquad = ((unsigned)(4 * wave) * (unsigned)a.ldp + (unsigned)(p0 + (lane & 31))) * 16u + 8u * hi;
    L.ldp16 = (unsigned)a.ldp * 16u;
    asm volatile("" : "+v"(L.ldp16));
    // fp32 rows (es / e0): row n = nu + 4 hi with nu wave-uniform -> address = array + (nu - split) ldp4 [scalar] + fcol, where fcol
    // carries this lane's point and its half's four rows
    unsigned ldp4 = (unsigned)a.ldp * 4u;
    asm volatile("" : "+s"(ldp4));
    unsigned fcol = (unsigned)(p0 + (lane & 31)) * 4u + (unsigned)hi * 4u * ldp4;
    asm volatile("" : "+v"(fcol));

    // saved activation quads of this lane (rows 32 wave + 8 q + 4 hi .. + 3 of point tile t), both planes, raw 16-bit
    // ONE set (16 registers): an epilogue refills each quad right after its last use with the quad its successor (the epilogue of the
    // tile being multiplied in the same stage) needs a full stage later -- two sets did not fit beside 128 weight registers
    uint2 hqh[4], hql[4];
    auto load_hq = [&](int q, const u16* hs, const u16* ls, int t) {
      const unsigned off = (unsigned)q * L.ldp16 + L.gquad + t * 512;
#ifdef NEAT_X3_ADJ_NOLOAD     // probe (results are WRONG): what the adjoint chain costs without its h loads
      hqh[q] = make_uint2(off, off); hql[q] = make_uint2(off >> 8, off >> 9);
      return;
#endif
      hqh[q] = x3_load8(reinterpret_cast<const char*>(hs) + off);
      hql[q] = x3_load8(reinterpret_cast<const char*>(ls) + off);
    };
    auto load_h = [&](const u16* hs, const u16* ls, int t) {
#pragma unroll
      for (int q = 0; q < 4; ++q) load_hq(q, hs, ls, t);
    };
    auto dphi2 = [](unsigned hw, unsigned lw, float& d0, float& d1) {      // phi' of the two values of a packed pair
      const float h0 = bf_lo(hw) + bf_lo(lw), h1 = bf_hi(hw) + bf_hi(lw);
      d0 = 1.0f - __builtin_amdgcn_exp2f(-SOFTPLUS_C * h0);
      d1 = 1.0f - __builtin_amdgcn_exp2f(-SOFTPLUS_C * h1);
    };
    // ---- seed: u_7 = w8 (.) phi'(h_8) -> XA (+ HBM hi plane)
    {
      float4 wq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) wq[q] = *reinterpret_cast<const float4*>(L.bias + (8 * q) * 4);
#pragma unroll
      for (int t = 0; t < (HALF ? 1 : 2); ++t) {
        load_h(a.h[8], a.hlo[8], t);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float d0, d1, d2, d3;
          dphi2(hqh[q].x, hql[q].x, d0, d1);
          dphi2(hqh[q].y, hql[q].y, d2, d3);
          uint2 vh, vl;
          x3_split2(wq[q].x * d0, wq[q].y * d1, vh.x, vl.x);
          x3_split2(wq[q].z * d2, wq[q].w * d3, vh.y, vl.y);
          *reinterpret_cast<uint2*>(L.quad[0] + (q * BP + t * 32) * 16) = vh;
          *reinterpret_cast<uint2*>(L.quad[0] + LO + (q * BP + t * 32) * 16) = vl;
          if (SAVE) x3_store8(reinterpret_cast<char*>(a.u[7]) + ((unsigned)q * L.ldp16 + L.gquad) + t * 512, vh);
        }
      }
    }
    load_h(a.h[7], a.hlo[7], 0);            // what the first epilogue (layer 7, tile 0) needs; every later quad set is requested by an epilogue
    __syncthreads();
    X3_STAMP(1);

    f32x16 acc[2];
    unsigned ph[2], pl[2];
    float keep = 0.0f;
    // epilogue element e of `ap` (tile t): MODE 0: * phi'(h) -> buffer DST (+ hi plane to uout); MODE 1 (l = 4): rows < 217 as MODE 0,
    // rows >= 217 -> fp32 rows (n - 217) of frows and zero on chip; MODE 2 (l = 0): rows < 39 -> fp32 rows of frows, nothing on chip
    // nhs / nls / nt: the h planes and tile whose quads replace this epilogue's (null: none)
    auto epi_elem = [&](const f32x16& ap, int e, int t, int MODE, int DST, u16* uout, float* frows, const u16* nhs, const u16* nls, int nt) {
      const int q = e >> 2, j = e & 3;
      const int nu = 32 * wave + 8 * q + j;                       // output row of the lower half (hi = 0); the upper half's is nu + 4
      if (MODE == 2) {                                            // rows < 39 -> e0 (waves 0, 1)
        if (nu + 4 < 39 || (nu < 39 && hi == 0))
          *reinterpret_cast<float*>(reinterpret_cast<char*>(frows) + ((unsigned)nu * ldp4 + fcol + t * 128)) = ap[e];
        return;
      }
      float r;
      {
        const unsigned hw = (j < 2) ? hqh[q].x : hqh[q].y, lw = (j < 2) ? hql[q].x : hql[q].y;
#ifdef NEAT_X3_ADJ_HI_ONLY      // probe: phi' from the hi plane alone
        const float h = (j & 1) ? bf_hi(hw) : bf_lo(hw);
        (void)lw;
#else
        const float h = (j & 1) ? (bf_hi(hw) + bf_hi(lw)) : (bf_lo(hw) + bf_lo(lw));
#endif
        r = ap[e] * (1.0f - __builtin_amdgcn_exp2f(-SOFTPLUS_C * h));
      }
      if (MODE == 1 && nu + 4 >= 217) {                           // (wave-uniform: waves 6 and 7 only)
        if (nu >= 217 || hi == 1) {
          *reinterpret_cast<float*>(reinterpret_cast<char*>(frows) + ((unsigned)(nu - 217) * ldp4 + fcol + t * 128)) = ap[e];
          r = 0.0f;
        }
      }
      if ((j & 1) == 0) { keep = r; return; }
      x3_split2(keep, r, ph[j >> 1], pl[j >> 1]);
      if (j != 3) return;
      if (nhs) load_hq(q, nhs, nls, nt);                          // (this quad's h values are dead now)
      const uint2 vh = make_uint2(ph[0], ph[1]), vl = make_uint2(pl[0], pl[1]);
      *reinterpret_cast<uint2*>(L.quad[DST] + (q * BP + t * 32) * 16) = vh;
      *reinterpret_cast<uint2*>(L.quad[DST] + LO + (q * BP + t * 32) * 16) = vl;
#ifndef NEAT_X3_ADJ_NOSTORE   // probe (results are WRONG): what the adjoint chain costs without its u stores
      if (SAVE && (MODE == 0 || wave < 7)) x3_store8(reinterpret_cast<char*>(uout) + ((unsigned)q * L.ldp16 + L.gquad) + t * 512, vh);
#endif
    };
    auto none = [](int) {};
    // ADJ_EPI(accumulators, tile, mode, destination buffer, u array, fp32 rows, [h planes, tile] of the NEXT epilogue's quads)
#define ADJ_EPI(ACC_, T_, MODE_, DST_, U_, F_, NH_, NL_, NT_) [&](int e) { epi_elem(ACC_, e, T_, MODE_, DST_, U_, F_, NH_, NL_, NT_); }
    // one layer: stage (tile 0) with the previous layer's tile-1 epilogue, stage (tile 1, rolling in the next layer's weights) with
    // this layer's tile-0 epilogue.  HS / HL: the h planes whose phi' multiplies THIS layer's output (requested a stage ahead).
    // HALF (one tile): the k-steps, then the next layer's weights are requested and the tile's epilogue HALF_EPI_ runs -- CUR_EPI_
    // with the h quads of the NEXT layer's tile 0 as its refill.
#define ADJ_LAYER(SRC_, WNH_, WNL_, NNEXT_, PREV_EPI_, CUR_EPI_, HALF_EPI_)                                                          \
    if (HALF) {                                                                                                                   \
      x3_stage<16, LO, true, true, 16>(L.frag[SRC_], wh, wl, acc[0], WNH_, WNL_, w_off(NNEXT_), none);                            \
      x3_drain(HALF_EPI_);                                                                                                        \
      __syncthreads();                                                                                                            \
    } else {                                                                                                                      \
      x3_stage<16, LO, true, false, 16>(L.frag[SRC_], wh, wl, acc[0], nullptr, nullptr, 0u, PREV_EPI_);                           \
      __syncthreads();                                                                                                            \
      x3_stage<16, LO, true, true, 16>(L.frag[SRC_] + 512, wh, wl, acc[1], WNH_, WNL_, w_off(NNEXT_), CUR_EPI_);                  \
      __syncthreads();                                                                                                            \
    }
    // stage (l, tile 0) runs the epilogue of (l + 1, tile 1), whose quads make room for (l, tile 0)'s; stage (l, tile 1) runs the
    // epilogue of (l, tile 0), whose quads make room for (l, tile 1)'s.  (The first stage has no epilogue: its successor's quads
    // were requested before the loop; layer 0's epilogues read no h.)
    ADJ_LAYER(0, a.Wp[6], a.Wlo[6], 256, none,
              ADJ_EPI(acc[0], 0, 0, 1, a.u[6], nullptr, a.h[7], a.hlo[7], 1),
              ADJ_EPI(acc[0], 0, 0, 1, a.u[6], nullptr, a.h[6], a.hlo[6], 0))                                                     // l = 7: XA -> XB
    X3_STAMP(2);
    ADJ_LAYER(1, a.Wp[5], a.Wlo[5], 256, ADJ_EPI(acc[1], 1, 0, 1, a.u[6], nullptr, a.h[6], a.hlo[6], 0),
              ADJ_EPI(acc[0], 0, 0, 0, a.u[5], nullptr, a.h[6], a.hlo[6], 1),
              ADJ_EPI(acc[0], 0, 0, 0, a.u[5], nullptr, a.h[5], a.hlo[5], 0))                                                     // l = 6: XB -> XA
    X3_STAMP(3);
    ADJ_LAYER(0, a.Wp[4], a.Wlo[4], 256, ADJ_EPI(acc[1], 1, 0, 0, a.u[5], nullptr, a.h[5], a.hlo[5], 0),
              ADJ_EPI(acc[0], 0, 0, 1, a.u[4], nullptr, a.h[5], a.hlo[5], 1),
              ADJ_EPI(acc[0], 0, 0, 1, a.u[4], nullptr, a.h[4], a.hlo[4], 0))                                                     // l = 5
    ADJ_LAYER(1, a.Wp[3], a.Wlo[3], 256, ADJ_EPI(acc[1], 1, 0, 1, a.u[4], nullptr, a.h[4], a.hlo[4], 0),
              ADJ_EPI(acc[0], 0, 1, 0, a.u[3], a.es, a.h[4], a.hlo[4], 1),
              ADJ_EPI(acc[0], 0, 1, 0, a.u[3], a.es, a.h[3], a.hlo[3], 0))                                                        // l = 4: rows 217.. -> es
    X3_STAMP(4);
    ADJ_LAYER(0, a.Wp[2], a.Wlo[2], 256, ADJ_EPI(acc[1], 1, 1, 0, a.u[3], a.es, a.h[3], a.hlo[3], 0),
              ADJ_EPI(acc[0], 0, 0, 1, a.u[2], nullptr, a.h[3], a.hlo[3], 1),
              ADJ_EPI(acc[0], 0, 0, 1, a.u[2], nullptr, a.h[2], a.hlo[2], 0))                                                     // l = 3
    ADJ_LAYER(1, a.Wp[1], a.Wlo[1], 256, ADJ_EPI(acc[1], 1, 0, 1, a.u[2], nullptr, a.h[2], a.hlo[2], 0),
              ADJ_EPI(acc[0], 0, 0, 0, a.u[1], nullptr, a.h[2], a.hlo[2], 1),
              ADJ_EPI(acc[0], 0, 0, 0, a.u[1], nullptr, a.h[1], a.hlo[1], 0))                                                     // l = 2
    ADJ_LAYER(0, a.Wp[0], a.Wlo[0], 39, ADJ_EPI(acc[1], 1, 0, 0, a.u[1], nullptr, a.h[1], a.hlo[1], 0),
              ADJ_EPI(acc[0], 0, 0, 1, a.u[0], nullptr, a.h[1], a.hlo[1], 1),
              ADJ_EPI(acc[0], 0, 0, 1, a.u[0], nullptr, nullptr, nullptr, 0))                                                     // l = 1
    X3_STAMP(5);
    ADJ_LAYER(1, a.Wp[7], a.Wlo[7], 256, ADJ_EPI(acc[1], 1, 0, 1, a.u[0], nullptr, nullptr, nullptr, 0),
              ADJ_EPI(acc[0], 0, 2, 0, nullptr, a.e0, nullptr, nullptr, 0),
              ADJ_EPI(acc[0], 0, 2, 0, nullptr, a.e0, nullptr, nullptr, 0))                                                       // l = 0: e0 (fp32)
    if (!HALF) {
      x3_drain(ADJ_EPI(acc[1], 1, 2, 0, nullptr, a.e0, nullptr, nullptr, 0));
      __syncthreads();
    }
    X3_STAMP(6);
#if NEAT_X3_TIMING
    if (blockIdx.x == 0 && nb_done < 3 && tid == 0)
      printf("x3 adjoint save %d batch %d: seed %llu l7 %llu l6 %llu l5+l4 %llu l3..l1 %llu l0 %llu cycles\n", (int)SAVE, nb_done,
             stamp[1] - stamp[0], stamp[2] - stamp[1], stamp[3] - stamp[2], stamp[4] - stamp[3], stamp[5] - stamp[4], stamp[6] - stamp[5]);
    ++nb_done;
#endif
#undef ADJ_LAYER
#undef ADJ_EPI
  };
  if ((int)blockIdx.x < 2 * nhalf) run_batch((nfull + ((int)blockIdx.x >> 1)) * BP + 32 * ((int)blockIdx.x & 1), std::true_type{});
  for (int bi = blockIdx.x; bi < nfull; bi += G) run_batch((NEAT_X3_ADJ_REVERSE ? nbatches - 1 - bi : bi) * BP, std::false_type{});
}

// ---------------------------------------------------------------------------------------------------------------
// One head (HEAD 0: rendering network rend_a :199-255, sigmoid, 3 rows; HEAD 1: attraction field :139-197, linear, 6 rows) with
// 3-product arithmetic: [feature rows (hi / lo from the primal chain) | small inputs (fp32 rows, split here)] -> 4 x (256, ReLU) -> out.
// SAVE: the hi planes of the four hidden activations go to a.hid[1..4] (what the 16-bit backward reads).
// ---------------------------------------------------------------------------------------------------------------

template <int HEAD, bool SAVE>
__global__ __launch_bounds__(512, 2) void head_chain_x3_kernel(HeadX3Args a, int nbatches) {
  typedef X3 C;
  constexpr int BP = C::BP, LO = C::XPL, SLO = C::SPL, NOUT = HEAD ? 6 : 3;
  extern __shared__ __attribute__((aligned(16))) unsigned char x3lds[];
  float* biasl = reinterpret_cast<float*>(x3lds + C::BIAS);      // [4][256] + [8]
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
  for (int idx = tid; idx < 2 * C::SPL / 16; idx += C::THREADS) reinterpret_cast<uint4*>(x3lds + C::S)[idx] = make_uint4(0u, 0u, 0u, 0u);

  uint4 wh[16], wl[16];
  auto w_off = [&](int KS, int N) -> unsigned {
    const int tile = wave * 32 < N ? wave : 0;
    unsigned v = (unsigned)((tile * KS) * 64 + lane) * 16u;
    asm volatile("" : "+v"(v));
    return v;
  };
  X3Lane L;
  {
    const unsigned fo = (unsigned)(hi * BP + (lane & 31)) * 16u, qo = (unsigned)((4 * wave) * BP + (lane & 31)) * 16u + 8u * hi;
    unsigned b0 = C::XA + fo, b1 = C::XB + fo, b2 = C::S + fo, q0 = C::XA + qo, q1 = C::XB + qo, bb = C::BIAS + (unsigned)(32 * wave + 4 * hi) * 4u;
    asm volatile("" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(q0), "+v"(q1), "+v"(bb));
    L.frag[0] = x3lds + b0; L.frag[1] = x3lds + b1; L.frag[2] = x3lds + b2;
    L.quad[0] = x3lds + q0; L.quad[1] = x3lds + q1; L.bias = x3lds + bb;
  }
  __syncthreads();

  const int nwork = nbatches < a.nvalid ? nbatches : a.nvalid;      // batches >= nvalid hold no ray sample (eikonal points, padding)
  // (Tried: requesting the next batch's lin0 weights and feature tile during lin4 -- registers idle there -- with the two activation
  //  buffers swapping roles per batch: 280 -> 380 us per launch.  The requests sit in front of lin4's own weight loads in the in-order
  //  vector-memory queue and the extra live registers spill; kept simple.)
  for (int batch = blockIdx.x; batch < nbatches; batch += gridDim.x) {
    const int p0 = batch * BP;
    L.gquad = ((unsigned)(4 * wave) * (unsigned)a.ldp + (unsigned)(p0 + (lane & 31))) * 16u + 8u * hi;
    L.ldp16 = (unsigned)a.ldp * 16u;
    asm volatile("" : "+v"(L.ldp16));
    int tb = tid;
    asm volatile("" : "+v"(tb));
    if (batch >= nwork) {
      // columns no ray sample lives in: nothing is computed; the saved hidden activations are zeroed (the backward pass contracts
      // them with zero cotangents: they must be finite) and so are the outputs
      if (SAVE) {
#pragma unroll
        for (int l = 1; l <= 4; ++l)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int idx = tb + i * C::THREADS, oct = idx >> 6, pp = idx & 63;
            *reinterpret_cast<uint4*>(reinterpret_cast<char*>(a.hid[l]) + ((unsigned)oct * (unsigned)a.ldp + (unsigned)(p0 + pp)) * 16u) = make_uint4(0u, 0u, 0u, 0u);
          }
#pragma unroll
        for (int l = 1; l <= 4; ++l)
          if (a.mask[l] && tb < 256) reinterpret_cast<uint4*>(reinterpret_cast<char*>(a.mask[l]) + (size_t)batch * 4096)[tb] = make_uint4(0u, 0u, 0u, 0u);
      }
      for (int idx = tb; idx < NOUT * BP; idx += C::THREADS) a.out[(size_t)(idx / BP) * a.ldp + p0 + idx % BP] = 0.0f;
      continue;
    }
    // lin0's weights: the 16 feature k-steps into the stationary set, the 4 small-input k-steps into a short-lived one
    uint4 sh[4], sl[4];
    {
      const unsigned o0 = w_off(20, 256);
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) { wh[ks] = x3_ldg(a.Wp[0], o0 + ks * 1024); wl[ks] = x3_ldg(a.Wlo[0], o0 + ks * 1024); }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) { sh[ks] = x3_ldg(a.Wp[0], o0 + (16 + ks) * 1024); sl[ks] = x3_ldg(a.Wlo[0], o0 + (16 + ks) * 1024); }
    }
    // ---- the feature tile (both planes) -> XA: a straight copy of 32 octet rows x 64 points x 16 B per plane
    {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int idx = tb + i * C::THREADS, oct = idx >> 6, pp = idx & 63;
        const unsigned go = ((unsigned)oct * (unsigned)a.ldp + (unsigned)(p0 + pp)) * 16u;
        const uint4 vh = x3_ldg(a.feat, go), vl = x3_ldg(a.featlo, go);
        reinterpret_cast<uint4*>(x3lds + C::XA)[idx] = vh;
        reinterpret_cast<uint4*>(x3lds + C::XA + LO)[idx] = vl;
      }
    }
    // ---- the small inputs -> S region, split
    {
      const int p = tb & (BP - 1), g = tb >> 6;
      unsigned pvo = (unsigned)(p0 + p) * 4u, ldp4 = (unsigned)a.ldp * 4u;
      asm volatile("" : "+v"(pvo), "+v"(ldp4));
      u16* shi = reinterpret_cast<u16*>(x3lds + C::S);
      u16* slo = reinterpret_cast<u16*>(x3lds + C::S + SLO);
#pragma unroll
      for (int jj = 0; jj < 5; ++jj) {
        const int j = g + 8 * jj;
        if (j < a.srows) {
          const float v = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.small) + ((unsigned)j * ldp4 + pvo));
          const u16 h = f2bf(v);
          shi[((j >> 3) * BP + p) * 8 + (j & 7)] = h;
          slo[((j >> 3) * BP + p) * 8 + (j & 7)] = f2bf(v - bf2f(h));
        }
      }
    }
    __syncthreads();

    f32x16 acc[2];
    float4 bq[4];
    auto load_bias = [&](int l) {
#pragma unroll
      for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const float4*>(L.bias + (l * 256 + 8 * q) * 4);
    };
    // the small-input part of lin0 seeds both accumulator tiles
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      acc[t] = zero;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const uint4 bh = *reinterpret_cast<const uint4*>(L.frag[2] + t * 512 + ks * C::KSTEP);
        const uint4 bl = *reinterpret_cast<const uint4*>(L.frag[2] + SLO + t * 512 + ks * C::KSTEP);
        acc[t] = NEAT_MFMA16(*reinterpret_cast<const bf16x8*>(&sh[ks]), *reinterpret_cast<const bf16x8*>(&bh), acc[t], 0, 0, 0);
        acc[t] = NEAT_MFMA16(*reinterpret_cast<const bf16x8*>(&sl[ks]), *reinterpret_cast<const bf16x8*>(&bh), acc[t], 0, 0, 0);
        acc[t] = NEAT_MFMA16(*reinterpret_cast<const bf16x8*>(&sh[ks]), *reinterpret_cast<const bf16x8*>(&bl), acc[t], 0, 0, 0);
      }
    }
    unsigned ph[2], pl[2];
    float keep = 0.0f;
    unsigned mbits = 0;
    // this lane's ReLU-mask word of tile 0 of the batch (kernels_heads.hpp: the fused backward chain reads it); + 2048 for tile 1
    const unsigned mlane = ((((unsigned)p0 >> 5) * 8u + (unsigned)wave) * 64u + (unsigned)lane) * 4u;
    auto epi_elem = [&](const f32x16& ap, int e, int t, int DST, u16* hout, u16* mout) {      // ReLU, split, quad -> buffer DST (+ hi plane to HBM)
      const int q = e >> 2, j = e & 3;
      const float b = j == 0 ? bq[q].x : (j == 1 ? bq[q].y : (j == 2 ? bq[q].z : bq[q].w));
      const float r = fmaxf(ap[e] + b, 0.0f);
      if ((j & 1) == 0) { keep = r; return; }
      x3_split2(keep, r, ph[j >> 1], pl[j >> 1]);
      if (SAVE) relu_mask_push(mbits, ph[j >> 1]);          // (r >= 0 here: the hi halves are zero or positive)
      if (SAVE && e == 15) {
        if (mout) *reinterpret_cast<unsigned*>(reinterpret_cast<char*>(mout) + (mlane + t * 2048)) = relu_mask_word(mbits);
        mbits = 0;
      }
      if (j != 3) return;
      const uint2 vh = make_uint2(ph[0], ph[1]), vl = make_uint2(pl[0], pl[1]);
      *reinterpret_cast<uint2*>(L.quad[DST] + (q * BP + t * 32) * 16) = vh;
      *reinterpret_cast<uint2*>(L.quad[DST] + LO + (q * BP + t * 32) * 16) = vl;
#ifndef NEAT_X3_NT_HEADS
#define NEAT_X3_NT_HEADS 0    // (the heads' hidden planes: measured +4 us per launch with non-temporal stores)
#endif
      if (SAVE) {
        if (NEAT_X3_NT_HEADS) x3_store8(reinterpret_cast<char*>(hout) + ((unsigned)q * L.ldp16 + L.gquad) + t * 512, vh);
        else *reinterpret_cast<uint2*>(reinterpret_cast<char*>(hout) + ((unsigned)q * L.ldp16 + L.gquad) + t * 512) = vh;
      }
    };
    auto none = [](int) {};
#define HD_EPI(ACC_, T_, DST_, L_) [&](int e) { epi_elem(ACC_, e, T_, DST_, a.hid[L_], a.mask[L_]); }
#define HD_LAYER(LCUR, SRC_, ZERO_, ROLL_, WNH_, WNL_, PREV_EPI_, CUR_EPI_)                                                         \
    {                                                                                                                             \
      x3_stage<16, LO, ZERO_, false, 16>(L.frag[SRC_], wh, wl, acc[0], nullptr, nullptr, 0u, PREV_EPI_);                          \
      __syncthreads();                                                                                                            \
      load_bias(LCUR);                                                                                                            \
      x3_stage<16, LO, ZERO_, ROLL_, 16>(L.frag[SRC_] + 512, wh, wl, acc[1], WNH_, WNL_, w_off(16, 256), CUR_EPI_);               \
      __syncthreads();                                                                                                            \
    }
    HD_LAYER(0, 0, false, true, a.Wp[1], a.Wlo[1], none, HD_EPI(acc[0], 0, 1, 1))                                   // lin0: XA (+ seeded small part) -> XB
    HD_LAYER(1, 1, true, true, a.Wp[2], a.Wlo[2], HD_EPI(acc[1], 1, 1, 1), HD_EPI(acc[0], 0, 0, 2))          // lin1: XB -> XA
    HD_LAYER(2, 0, true, true, a.Wp[3], a.Wlo[3], HD_EPI(acc[1], 1, 0, 2), HD_EPI(acc[0], 0, 1, 3))          // lin2: XA -> XB
    HD_LAYER(3, 1, true, false, nullptr, nullptr, HD_EPI(acc[1], 1, 1, 3), HD_EPI(acc[0], 0, 0, 4))          // lin3: XB -> XA
    x3_drain(HD_EPI(acc[1], 1, 0, 4));
    __syncthreads();
#undef HD_LAYER
#undef HD_EPI
    // ---- lin4 (3 / 6 rows): K split over the waves (2 k-steps each), partial sums through the B-role buffer (idle since lin3's
    // last read), then bias (+ sigmoid).  (Not through the small-input region: its rows beyond the inputs must stay zero.)
    {
      float* red = reinterpret_cast<float*>(x3lds + C::XB);       // [8 waves][8 rows][BP] = 16 KiB
      unsigned so = (unsigned)(((2 * wave) * 64 + lane) * 16);
      asm volatile("" : "+v"(so));
      uint4 oh[2], ol[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) { oh[j] = x3_ldg(a.Wp[4], so + j * 1024); ol[j] = x3_ldg(a.Wlo[4], so + j * 1024); }
      const unsigned char* fr = L.frag[0] + (unsigned)(2 * wave) * C::KSTEP;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f32x16 accs;
#pragma unroll
        for (int r = 0; r < 16; ++r) accs[r] = 0.0f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const uint4 bh = *reinterpret_cast<const uint4*>(fr + j * C::KSTEP + t * 512);
          const uint4 bl = *reinterpret_cast<const uint4*>(fr + LO + j * C::KSTEP + t * 512);
          accs = NEAT_MFMA16(*reinterpret_cast<const bf16x8*>(&oh[j]), *reinterpret_cast<const bf16x8*>(&bh), accs, 0, 0, 0);
          accs = NEAT_MFMA16(*reinterpret_cast<const bf16x8*>(&ol[j]), *reinterpret_cast<const bf16x8*>(&bh), accs, 0, 0, 0);
          accs = NEAT_MFMA16(*reinterpret_cast<const bf16x8*>(&oh[j]), *reinterpret_cast<const bf16x8*>(&bl), accs, 0, 0, 0);
        }
        // rows 0..3 in lanes 0-31 (registers 0..3), rows 4..7 in lanes 32-63
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(wave * 8 + 4 * hi + r) * BP + t * 32 + (lane & 31)] = accs[r];
      }
      __syncthreads();
      int tf = tid;
      asm volatile("" : "+v"(tf));
      for (int idx = tf; idx < NOUT * BP; idx += C::THREADS) {
        const int n = idx / BP, pp = idx % BP;
        float v = biasl[4 * 256 + n];
#pragma unroll
        for (int w = 0; w < 8; ++w) v += red[(w * 8 + n) * BP + pp];
        if (HEAD == 0) v = 1.0f / (1.0f + __expf(-v));
        a.out[(size_t)n * a.ldp + p0 + pp] = v;
      }
    }
    __syncthreads();
  }
}

}  // namespace neat

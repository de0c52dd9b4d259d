// Host orchestration + C ABI of libneat_hip.so (see include/neat_hip.h).
// One stream-ordered sequence of kernel launches per entry point; no allocation, no sync.
// Two builds of the GEMM-class kernels share the orchestration: precision 0 = exact-f32 MFMA with fp32
// feature-major activations (parity build), precision 1 = bf16 MFMA (fp32 accumulate) with bf16 octet-major
// hidden activations (throughput build).  Small arrays are fp32 feature-major in both.
// NEAT_F16 (precision 3) is this same file compiled a second time with -DNEAT_HALF=1 (build.sh): namespace neat becomes neat_f16, every
// C entry point gets the prefix f16_ (f16_symbols.h, generated from include/neat_hip.h), bf16_common.hpp switches the 16-bit format
// to IEEE half.  The primary build's entry points forward precision 3 to that twin as ITS precision 1 (NEAT_F16_FWD below).
#include <mutex>
#include <unordered_map>
#if defined(NEAT_HALF) && NEAT_HALF
#define neat neat_f16
#include "f16_symbols.h"
#endif
#include "kernels_bf16.hpp"
#include "kernels_dw.hpp"
#include "fused_launch.hpp"
#include "kernels_sampler.hpp"
#include "kernels_junction.hpp"
#include "../../include/neat_hip.h"
#include <algorithm>
#include <initializer_list>
#include <utility>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

using namespace neat;

namespace {

// ------------------------------------------------------------------------------------------------
// architecture (identical in all shipped confs: confs/abc-neat-a.conf:42-70, dtu.conf, bmvs.conf)
// ------------------------------------------------------------------------------------------------
constexpr int L_REND = 9, L_ATTR = 14;
const int kO[NLAYERS] = {256, 256, 256, 217, 256, 256, 256, 256, 257, 256, 256, 256, 256, 3, 256, 256, 256, 256, 6};
const int kI[NLAYERS] = {39, 256, 256, 256, 256, 256, 256, 256, 256, 289, 256, 256, 256, 256, 265, 256, 256, 256, 256};
constexpr int PE_ROWS = 39, SMALL_R = 33, SMALL_A = 9;
enum { F32 = 0, BF16 = 1, BF16X3 = 2, HX3 = 4, HX3_FASTVALUES = 5 };
// HX3 (NEAT_F16X3, served by the f16 twin only) = the 16-bit build (layouts, backward pass) whose three FORWARD chains run with
// 3-product hi/lo arithmetic (kernels_x3.hpp) and save lo planes where a later forward kernel needs them: every entry point maps it
// to BF16 + Ctx::hx3 (take_hx3)
// HX3_FASTVALUES (5): values-mode calls only (neat_sdf_forward mode 0, neat_sdf_values_gated) on a NEAT_F16X3 pack: the ONE-product
// f16 chain of the fp16 build evaluates the query (3x faster; for a sampler that may trade the reference's exact depths for speed)
inline int take_hx3(int& precision) {
  if (NEAT_HALF && precision == HX3) { precision = BF16; return 1; }
  if (NEAT_HALF && precision == HX3_FASTVALUES) { precision = BF16; return 2; }
  return 0;
}
// BF16X3 = the F32 build (layouts, kernels, workspaces) with split-bf16 products in its two GEMM kernels: every entry point maps it
// to F32 + Ctx::x3 (take_x3)
inline int take_x3(int& precision) { if (precision == BF16X3) { precision = F32; return 1; } return 0; }

inline int padk(int k, int prec) { return prec ? (k + 63) & ~63 : (k + 7) & ~7; }   // bf16: k-steps of 16, unrolled by 4
inline int pad8(int k) { return (k + 7) & ~7; }
inline int tiles32(int n) { return (n + 31) / 32; }

struct PackLayout {
  PackDesc2 d[2 * MAXPACKS2];
  int npacks, nblocks;                         // the packs of the plain builds: d[0 .. npacks)
  int npacks_lo, nblocks_lo;                   // HX3: lo-plane packs d[npacks .. npacks + npacks_lo), blk0 counted from 0 again (second launch)
  int fwd[NLAYERS], tr[NLAYERS], sdf_row;      // pack ids (sdf_row: lin8 restricted to the sdf output row)
  int fwd_lo[NLAYERS], tr_lo[NLAYERS], sdf_row_lo;
  int row_off[NLAYERS + 1];
  size_t rowscale_off, total;
};

void build_layout(PackLayout& L, int prec, int hx3) {
  size_t off = 0;
  int np = 0, blk = 0;
  L.row_off[0] = 0;
  for (int l = 0; l < NLAYERS; ++l) L.row_off[l + 1] = L.row_off[l] + kO[l];
  int lo_plane = 0;
  auto add = [&](int l, int t, int nrows_limit, int rot) {
    PackDesc2& d = L.d[np];
    d.layer = l; d.transpose = t; d.bf16 = prec; d.lo = lo_plane;
    d.s0 = kI[l]; d.s0p = kI[l]; d.off0 = 0; d.off1 = 0;
    if (l == L_REND) { d.s0 = 256; d.s0p = 256; d.off0 = SMALL_R; d.off1 = 0; }          // [feature | p, PE4(view), normal]
    if (l == L_ATTR) { d.s0 = 256; d.s0p = 256; d.off0 = SMALL_A; d.off1 = 0; }          // [feature | p, view, normal]
    if (l == 4) { d.s0 = 217; d.s0p = 217; d.off0 = 0; d.off1 = 217; }                     // [h4 | PE]  (skip, rend_a :87-88)
    d.rot = rot;
    d.scale = (l == 4) ? (float)(1.0 / sqrt(2.0)) : 1.0f;                                 // /sqrt2 of the skip concat folded in
    const int kin = d.s0p + (kI[l] - d.s0);
    d.N = t ? kin : (nrows_limit > 0 ? nrows_limit : kO[l]);
    d.K = t ? kO[l] : kin;
    d.Kpad = padk(d.K, prec); d.NT = tiles32(d.N);
    d.offset = (int)off; d.blk0 = blk;
    off += prec ? (size_t)d.NT * (d.Kpad / 16) * 64 * 4 : (size_t)d.NT * (d.Kpad / 2) * 64;
    blk += d.NT;
    return np++;
  };
  for (int l = 0; l < NLAYERS; ++l) {
    const int rot = (l == 8 && prec) ? 1 : 0;         // bf16: lin8 rows reordered to [feature(256) | sdf] (octet aligned)
    L.fwd[l] = add(l, 0, 0, rot);
    L.tr[l] = add(l, 1, 0, rot);
  }
  L.sdf_row = add(8, 0, 1, 0);
  L.npacks = np; L.nblocks = blk;
  L.npacks_lo = 0; L.nblocks_lo = 0;
  if (hx3) {           // lo planes: every forward pack, the transposed packs of the SDF layers the adjoint chain walks, the sdf row
    lo_plane = 1; blk = 0;
    for (int l = 0; l < NLAYERS; ++l) {
      L.fwd_lo[l] = add(l, 0, 0, (l == 8) ? 1 : 0);
      if (l < 8) L.tr_lo[l] = add(l, 1, 0, 0);
    }
    L.sdf_row_lo = add(8, 0, 1, 0);
    L.npacks_lo = np - L.npacks; L.nblocks_lo = blk;
  }
  L.rowscale_off = off;
  off += (size_t)((L.row_off[NLAYERS] + 63) & ~63);
  L.total = off;
}

const PackLayout& pack_layout(int prec, int hx3 = 0) {
  static PackLayout L[3];
  static bool init[3] = {false, false, false};
  const int i = hx3 ? 2 : prec;
  if (!init[i]) { build_layout(L[i], prec, hx3); init[i] = true; }
  return L[i];
}

NetPtrs to_ptrs(const neat_net_params* net) {
  NetPtrs p;
  for (int l = 0; l < NLAYERS; ++l) { p.v[l] = net->v[l]; p.g[l] = net->g[l]; p.b[l] = net->b[l]; p.O[l] = kO[l]; p.I[l] = kI[l]; }
  return p;
}

// NEAT_DEBUG=1: synchronise after every GEMM-class launch and log it (fault localisation only)
inline bool dbg() { static int v = -1; if (v < 0) { const char* e = getenv("NEAT_DEBUG"); v = (e && e[0] == '1') ? 1 : 0; } return v == 1; }
inline void dbg_sync(hipStream_t st, const char* what, int a, int b, int c2) {
  if (!dbg()) return;
  hipError_t e = hipStreamSynchronize(st);
  fprintf(stderr, "[neat] %s %d %d %d -> %s\n", what, a, b, c2, hipGetErrorString(e));
  fflush(stderr);
}

#define NEAT_CHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return (int)e_; } while (0)

// ------------------------------------------------------------------------------------------------
// optional in-stream kernel timing (bench.py): HIP events around every launch of the two GEMM-class kernels
// ------------------------------------------------------------------------------------------------
struct ProfSlot { hipEvent_t e0, e1; double flops, bytes; int cls; };
struct Prof {
  bool on = false;
  std::vector<ProfSlot> pool;
  size_t used = 0;
} g_prof;

inline ProfSlot* prof_begin(hipStream_t st, int cls, double flops, double bytes = 0.0) {
  if (!g_prof.on) return nullptr;
  if (g_prof.used == g_prof.pool.size()) {
    ProfSlot s{};
    if (hipEventCreate(&s.e0) != hipSuccess || hipEventCreate(&s.e1) != hipSuccess) return nullptr;
    g_prof.pool.push_back(s);
  }
  ProfSlot* s = &g_prof.pool[g_prof.used++];
  s->flops = flops; s->bytes = bytes; s->cls = cls;
  hipEventRecord(s->e0, st);
  return s;
}
inline void prof_end(hipStream_t st, ProfSlot* s) { if (s) hipEventRecord(s->e1, st); }

// ------------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------------
template <int EPI> hipError_t launch_layer_f(hipStream_t st, const LayerArgs& a, int ntiles_p) {
  static DevOnce attr_set;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&layer_kernel<EPI>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  size_t lds = (size_t)a.Kpad * BM * sizeof(float);
  if (a.NT <= 2 && lds < 32768) lds = 32768;
  hipLaunchKernelGGL(layer_kernel<EPI>, dim3(ntiles_p), dim3(WG), lds, st, a);
  return hipGetLastError();
}
int g_wgrad_h3 = 1;         // bf16 weight gradient of the all-bf16 256x256 layers: 1 = tr16/4-stage DMA kernel, 0 = wgrad_kernel_h2
int g_pt_bf16 = 2;          // 32-point column tiles per workgroup in the bf16 layer kernel: 2 (64 pts, higher occupancy) or 4

template <int EPI, int PT, bool OBF> hipError_t launch_layer_h_pt(hipStream_t st, const LayerArgsH& a) {
  static DevOnce attr_set;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&layer_kernel_h<EPI, PT, OBF>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  constexpr int BMT = 32 * PT;
  size_t lds = (size_t)(a.Kpad / 8) * BMT * 16;
  const size_t red = (size_t)4 * PT * 16 * 64 * 4;
  if (a.NT <= 2 && lds < red) lds = red;
  hipLaunchKernelGGL((layer_kernel_h<EPI, PT, OBF>), dim3(a.ldp / BMT), dim3(WG), lds, st, a);
  return hipGetLastError();
}
template <int EPI> hipError_t launch_layer_h(hipStream_t st, const LayerArgsH& a, int) {
  // the 128-point tile (PT = 4) is kept as a tuning option for the bf16-output variants only
  if (a.out0_bf16) return g_pt_bf16 == 4 ? launch_layer_h_pt<EPI, 4, true>(st, a) : launch_layer_h_pt<EPI, 2, true>(st, a);
  return launch_layer_h_pt<EPI, 2, false>(st, a);
}
int g_wgrad_batch = -1;     // same-shaped hidden-layer weight gradients per launch: -1 = by operand footprint (below), 0 = one launch
                            // per layer, 2 / 3 / 6 = forced
// Measured on MI355X (ms/step, same box): 131 k points: 2 -> 4.06, 3 -> 4.01, 6 -> 4.25; 262 k points: 0 -> 7.44, 2 -> 7.29,
// 3 -> 7.75.  Fewer, longer point ranges per workgroup save partial tiles, but a launch whose operand arrays add up to
// more than ~1 GB (six two-pair problems at 131 k points, three at 262 k) runs 2.5-2.9x slower per byte.
inline int wgrad_batch_size(int ldp) {
  if (g_wgrad_batch >= 0) return g_wgrad_batch;
  return ldp <= 200000 ? 3 : (ldp <= 400000 ? 2 : 0);
}
int g_wgrad_interleave = 0; // 1: launch each SDF layer's weight gradient right after the reverse step that produced its cotangent (Infinity-Cache reuse; measured neutral)
int g_wreduce_direct = 2;   // bf16 weight-gradient reduction: 2 = one launch per layer / batch with 16-byte loads (wreduce_direct_kernel), 0 = group
                            // sums + finish (two launches, stage buffer), 1 = one 16-wave pass with 4-byte loads (slowest)
int g_head_chain = 2;       // 16-bit builds: the heads as fused chains (kernels_heads.hpp): 2 = forward and backward, 1 = forward only, 0 = per-layer
                            // launches of layer_kernel_ws (tuning key 14)
int g_head_wgrad_order = 1; // fused head backward: a head's weight gradients right after its backward chain (tuning key 15; 0 = after both chains, hidden layers of the two heads batched together)
int g_fused_adj = 1;        // bf16 build: adjoint chain (normals) as one fused launch (sdf_adjoint_w64_kernel); 0 = seed + eight streaming EPI_REV launches (tuning key 13)
int g_fused_ws = 3;         // fused primal chain: 4 = phase-staggered kernel (sdf_fused_ph_kernel; measured slower: a matrix wave and a
                            // vector wave on one SIMD do not overlap on this machine, scripts/probes/probe_roles.hip), 3 / 2 = stage-pipelined kernel of kernels_fused.hpp with 8 waves x 32 rows / 4 waves x 64 rows,
                            // 1 = first weight-stationary kernel (8 waves x 32 rows), 0 = sdf_fused_kernel_h
int g_fused_nt = 0;         // its batch: 4 = 128 points, 2 = 64 points, 0 = whichever balances the CUs better
int g_layer_ws = 1;         // hidden 256x256 bf16 layers: 1 = weight-stationary streaming kernel, 0 = layer_kernel_h
int g_ws_grid = 256;        // persistent workgroups of layer_kernel_ws (one per CU)
constexpr int DW_MAXGRID = 256;     // workgroup partials the workspace holds per (layer, pair)
int g_dw_grid = DW_MAXGRID;  // workgroups (= partials per set) of the launches that contract weight gradients on chip (tuning key 23); 0 = balanced (dw_grid)
inline int dw_slots(int ldp) {          // workgroup partials the workspace holds per (layer, pair): what dw_grid() can reach at this size
  const int nt = ldp / WSP;
  return nt < DW_MAXGRID ? (nt > 0 ? nt : 1) : DW_MAXGRID;
}
inline int dw_grid(int ldp) {
  int g = g_ws_grid < DW_MAXGRID ? g_ws_grid : DW_MAXGRID;
  const int nt = ldp / WSP;
  if (g_dw_grid > 0) g = g < g_dw_grid ? g : g_dw_grid;
  else if (nt > g) {
    // key 23 = 0 (round 6, measured, NOT the default): the persistent workgroups walk ceil(nt / g) tiles, so 4160 tiles over 256 workgroups cost
    // 17 rounds for 16.25 of work; this takes the grid in [208, 256] (multiples of 8) that leaves the fewest idle tile slots -- 208 at C2 (20 tiles
    // each) and at C4's rank shape (10 each), with 19 % fewer partials for the gather.  Same-box A/B, two passes, 256 vs balanced: C2 2.721 / 2.709
    // vs 2.726 / 2.718 ms, C4 rank 1.764 / 1.774 vs 1.761 / 1.760, real step 3.350 / 3.378 vs 3.403 / 3.350, C3 4.954 / 4.995 vs 5.015 / 5.005:
    // what the balance wins the 48 idle compute units lose in bandwidth.  (160 workgroups: +4 % on C2, 128: +12 %.)
    int best = g, waste = ((nt + g - 1) / g) * g - nt;
    for (int c = g - 8; c >= 208; c -= 8) {
      const int w = ((nt + c - 1) / c) * c - nt;
      if (w < waste) { waste = w; best = c; }
    }
    g = best;
  }
  return nt < g ? nt : g;
}
int g_ws_aux_nt = 15;       // non-temporal accesses (tuning key 11): bit 0 / 1 = fetch of aux0 / aux1 of the streaming layer kernels, bit 2 =
                            // weight-gradient operands, bit 3 = `in` of the layer kernels, bit 4 = store of out1 (m_l)
int g_ws_wide_store = 1;    // streaming layer kernels: 16-byte output stores (tuning key 12)
int g_sampler_ablate = 0;
const int* g_gate = nullptr; int g_gate_value = 0;      // set around one neat_sdf_forward call by neat_sdf_values_gated
int g_fused_interleave = 0; // fused primal chain: batches interleaved over the workgroups (tuning key 10)
int g_ws_interleave = 1;    // 1: tiles interleaved over the workgroups instead of one contiguous range each
int g_wgrad_narrow = 1;     // 16-bit builds: lin0's weight gradient (K = 39) on the one-column-block variant of wgrad_kernel_h3 (tuning key 21)
int g_wgrad_k320 = 1;       // 16-bit builds: the heads' input layers (K = 256 + <= 64) as one five-column-block weight-gradient launch (tuning key 19)
int g_head_l4_batched = 1;  // a head's output-layer weight gradient as a fourth problem of its hidden layers' launch (tuning key 20)
int g_dw_ablate = 0;        // probe runs (tuning key 17): see LayerArgsDW::ablate
int g_dw_fused = 1;         // 16-bit builds: weight gradients of the SDF layers 1..7 accumulated inside the tangent / reverse launches
                            // (kernels_dw.hpp; tuning key 16): 1 = where it pays (>= DW_MIN_POINTS points: the partials and their gather cost
                            // the same ~1 GB of traffic per pass whatever the point count, the operand reads they replace scale with it),
                            // 2 = always, 0 = never (the separate wgrad_kernel_h3 launches of round 3)
constexpr int DW_MIN_POINTS = 49152;
constexpr int DW8_XBLOCKS = 512;     // block partials of the sdf row of dW8 (rowdot_kernel)
int g_dw_lin8 = 1;          // with key 16: lin8's feature-row gradient (featc x h8) inside its reverse launch too (tuning key 22)
int g_chain_pp = 1;         // with key 16: the chain variables of the tangent / reverse launches (vhat_2..7, a^_6..1) ping-pong between two buffers each
                            // instead of one array per layer: since the weight gradients are contracted in the launch that holds them no later
                            // kernel reads them, and a line rewritten while it is still in the Infinity Cache never costs an HBM write (tuning key 24)
int g_dw_nsub = 16;         // sub-ranges of workgroup partials summed by dw_gather_kernel (= fp32 splits per layer seen by the finish; tuning key 18)
int g_ffn_mfma = 1;         // the 256 x 256 layers of the global-junction MLP (forward and data backward) on the fp32 matrix pipe (ffn_mfma_kernel) instead
                            // of the vector-ALU kernel (tuning key 28)
int g_dw_segments = 1;      // with key 16: consecutive layers of a chain that share an epilogue variant run as ONE launch with a per-workgroup layer loop
                            // (tangent 1-2 | 3 | 4-7, reverse 8 | 7-5 | 4 | 3-1: 15 launches -> 7; tuning key 25)
template <int EPI, bool FULL> hipError_t launch_layer_wsdw(hipStream_t st, const LayerArgsDW* d0, int n = 1) {
  static DevOnce attr_set;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&layer_kernel_wsdw<EPI, FULL>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, WsCfg<EPI, 16>::LDS + DW_XLDS);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  if (n < 1 || n > DW_MAXSEG || (n > 1 && !(FULL && EPI != EPI_BWD8))) return hipErrorInvalidValue;      // (only the FULL variants carry the layer loop)
  LayerArgsDWSeg seg{};
  seg.n = n;
  for (int i = 0; i < n; ++i) {
    LayerArgsDW& d = seg.l[i];
    d = d0[i];
    d.w.ntiles = d.w.ldp / WSP;
    d.w.aux_nt = (g_ws_aux_nt & 3) | ((g_ws_aux_nt >> 1) & 28);
    d.ablate = g_dw_ablate;
    if (FULL && (d.w.N != 256 || d.rowsA != 256)) return hipErrorInvalidValue;
    if (d.w.ldp != d0[0].w.ldp) return hipErrorInvalidValue;
  }
  hipLaunchKernelGGL((layer_kernel_wsdw<EPI, FULL>), dim3(dw_grid(seg.l[0].w.ldp)), dim3(WST), (WsCfg<EPI, 16>::LDS + DW_XLDS), st, seg);
  return hipGetLastError();
}
template <int EPI, bool FULL> hipError_t launch_layer_wsdw(hipStream_t st, const LayerArgsDW& d0) { return launch_layer_wsdw<EPI, FULL>(st, &d0, 1); }
template <int EPI, int KS = 16, bool OUTF = false> hipError_t launch_layer_ws(hipStream_t st, const LayerArgsWS& a0) {
  static DevOnce attr_set;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&layer_kernel_ws<EPI, KS, OUTF>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, WsCfg<EPI, KS>::LDS);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  LayerArgsWS a = a0;
  a.ntiles = a.ldp / WSP;
  a.per_wg = (a.ntiles + g_ws_grid - 1) / g_ws_grid;
  int grid = (a.ntiles + a.per_wg - 1) / a.per_wg;
  a.tile_stride = 1;
  a.wide_store = g_ws_wide_store;
  a.aux_nt = (g_ws_aux_nt & 3) | ((g_ws_aux_nt >> 1) & 28);      // key bits 3 / 4 / 5 -> kernel bits 2 / 3 / 4
  a.xcd_major = g_ws_interleave == 2;
  if (g_ws_interleave) { grid = a.ntiles < g_ws_grid ? a.ntiles : g_ws_grid; a.tile_stride = grid; }
  hipLaunchKernelGGL((layer_kernel_ws<EPI, KS, OUTF>), dim3(grid), dim3(WST), (WsCfg<EPI, KS>::LDS), st, a);
  return hipGetLastError();
}
// the weight-stationary kernel covers: one bf16 octet-major input of up to 256 rows (K = 256 packed columns), bf16
// outputs without split / accumulate / pad-fill, and the epilogues of the hidden layers
// (c) K <= 64 bf16 input (PE tangent rows into EPI_TAN, the 3/6-row cotangents into EPI_BWD_RELU).
// shapes: (a) hidden: one bf16 input of 256 (or 217/224) rows, optionally continued by a 32-row bf16 array at octet 28
// (skip layer), bf16 output(s); EPI_REV may split rows >= n_split into fp32 rows; (b) narrow fp32 outputs (N <= 256,
// LINEAR / SIGMOID) from a 256-row bf16 input.
inline int ws_shape(int epi, const LayerArgsH& a) {
  if (!g_layer_ws) return 0;
  const bool seg1 = a.in[1].rows != 0;
  if (!a.in[0].bf16 || a.accumulate || (a.bias && a.bias_rot != 0)) return 0;
  if (a.padfill) {                                                       // (d) lin3's tangent layer: pad-fill rows from an octet-major copy
    return (epi == EPI_TAN && a.padfill_oct && a.padfill_rows == 7 && a.N == 217 && a.Kpad == 256 && !seg1 && a.in[0].rows == 256 &&
            a.out0_bf16 && a.out1 && a.out1_bf16 && a.n_split >= a.N && !a.bias) ? 5 : 0;
  }
  if (a.Kpad == 320) {                                                   // (e) head input layers: [256 feature rows | few small rows]
    return (epi == EPI_RELU && seg1 && a.in[1].bf16 && a.in[0].rows == 256 && a.in[1].rows <= 64 && a.out0_bf16 && a.N == 256 &&
            a.n_split >= a.N && !a.out1) ? 4 : 0;
  }
  if (a.Kpad == 64 && !seg1) {                                           // (c) K <= 64: narrow cotangents / PE tangent rows
    if (!a.out0_bf16 || a.N != 256 || a.n_split < a.N || a.bias) return 0;
    if (epi == EPI_BWD_RELU) return a.out1 ? 0 : 3;
    if (epi == EPI_TAN) return (a.out1 && a.out1_bf16) ? 3 : 0;
    return 0;
  }
  if (a.Kpad != 256) return 0;
  const int oct0 = (a.in[0].rows + 7) / 8;
  if (seg1) { if (!(a.in[1].bf16 && a.in[1].rows == 32 && oct0 == 28)) return 0; }
  else if (!(a.in[0].rows == 256 || oct0 == 28)) return 0;
  if (!a.out0_bf16) {                                                    // (b)
    return ((epi == EPI_LINEAR || epi == EPI_SIGMOID) && !seg1 && a.in[0].rows == 256 && a.N <= 256 && a.n_split >= a.N && !a.out1) ? 2 : 0;
  }
  if (a.N > 256 || a.N < 193) return 0;
  if (epi == EPI_REV) {
    if (a.n_split < a.N) { if (!a.out1 || a.out1_bf16 || a.n_split < 193) return 0; }
    else if (a.out1) return 0;
    return a.bias ? 0 : 1;
  }
  if (a.n_split < a.N) return 0;
  if (epi == EPI_TAN) return (a.out1 && a.out1_bf16 && !a.bias) ? 1 : 0;
  if (a.out1) return 0;
  if (epi == EPI_RELU) return 1;
  if (epi == EPI_BWD || epi == EPI_BWD_RELU) return a.bias ? 0 : 1;
  return 0;
}
hipError_t dispatch_ws_args(hipStream_t st, int epi, const LayerArgsWS& a) {
  switch (epi) {
    case EPI_LINEAR: return launch_layer_ws<EPI_LINEAR>(st, a);
    case EPI_LINACC: return launch_layer_ws<EPI_LINACC>(st, a);
    case EPI_RELU: return launch_layer_ws<EPI_RELU>(st, a);
    case EPI_REV: return launch_layer_ws<EPI_REV>(st, a);
    case EPI_TAN: return launch_layer_ws<EPI_TAN>(st, a);
    case EPI_BWD: return launch_layer_ws<EPI_BWD>(st, a);
    case EPI_BWD8: return launch_layer_ws<EPI_BWD8>(st, a);
    case EPI_BWD_RELU: return launch_layer_ws<EPI_BWD_RELU>(st, a);
  }
  return hipErrorInvalidValue;
}
hipError_t dispatch_ws(hipStream_t st, int epi, const LayerArgsH& h, int shape) {
  LayerArgsWS a{};
  a.in = reinterpret_cast<const u16*>(h.in[0].p); a.Wp = h.Wp; a.bias = h.bias;
  a.in2 = reinterpret_cast<const u16*>(h.in[1].p); a.split_oct = h.in[1].rows ? 28 : 32;
  a.aux0 = h.aux0; a.aux1 = h.aux1;
  a.out0 = reinterpret_cast<u16*>(h.out0); a.out1 = reinterpret_cast<u16*>(h.out1);
  a.out0f = reinterpret_cast<float*>(h.out0); a.out1f = reinterpret_cast<float*>(h.out1); a.n_split = h.n_split;
  a.N = h.N; a.in_octs = (h.in[0].rows + 7) / 8 + (h.in[1].rows ? 4 : 0); a.ldp = h.ldp; a.kstride = h.Kpad / 16;
  if (shape == 5) { a.padfill = h.padfill_oct; return launch_layer_ws<EPI_TAN_PF, 16, false>(st, a); }
  if (shape == 4) { a.split_oct = 32; a.in_octs = 32; a.x_octs = (h.in[1].rows + 7) / 8; return launch_layer_ws<EPI_RELU, 20, false>(st, a); }
  if (shape == 3) return epi == EPI_TAN ? launch_layer_ws<EPI_TAN, 4, false>(st, a) : launch_layer_ws<EPI_BWD_RELU, 4, false>(st, a);
  if (shape == 2) return epi == EPI_SIGMOID ? launch_layer_ws<EPI_SIGMOID, 16, true>(st, a) : launch_layer_ws<EPI_LINEAR, 16, true>(st, a);
  if (epi == EPI_REV && h.n_split >= h.N) { a.out1f = nullptr; a.n_split = 1 << 30; }
  return dispatch_ws_args(st, epi, a);
}
#define EPI_SWITCH(FN, st, epi, a, nt)                                         \
  switch (epi) {                                                              \
    case EPI_LINEAR: return FN<EPI_LINEAR>(st, a, nt);                        \
    case EPI_SOFTPLUS: return FN<EPI_SOFTPLUS>(st, a, nt);                    \
    case EPI_RELU: return FN<EPI_RELU>(st, a, nt);                            \
    case EPI_SIGMOID: return FN<EPI_SIGMOID>(st, a, nt);                      \
    case EPI_REV: return FN<EPI_REV>(st, a, nt);                              \
    case EPI_TAN: return FN<EPI_TAN>(st, a, nt);                              \
    case EPI_BWD: return FN<EPI_BWD>(st, a, nt);                              \
    case EPI_BWD_RELU: return FN<EPI_BWD_RELU>(st, a, nt);                    \
  }                                                                           \
  return hipErrorInvalidValue;
hipError_t dispatch_f(hipStream_t st, int epi, const LayerArgs& a, int nt) { EPI_SWITCH(launch_layer_f, st, epi, a, nt) }
hipError_t dispatch_h(hipStream_t st, int epi, const LayerArgsH& a, int nt) { EPI_SWITCH(launch_layer_h, st, epi, a, nt) }

// sdf_finalize_kernel<FAST>: hardware sin/cos in the bf16 build
// (the split-precision forward has the PE rows of the points in w.E -- libm's sin / cos, posenc6_kernel -- and reads them back)
#define FINALIZE_LAUNCH_H(c, hin, ...)                                                                                  \
  do {                                                                                                                   \
    if ((c).hx3) hipLaunchKernelGGL(sdf_finalize_kernel<false>, grid1((c).ldp), dim3(256), 0, (c).st, __VA_ARGS__, (const float*)w.E, hin); \
    else if ((c).prec) hipLaunchKernelGGL(sdf_finalize_kernel<true>, grid1((c).ldp), dim3(256), 0, (c).st, __VA_ARGS__, (const float*)nullptr, hin); \
    else hipLaunchKernelGGL(sdf_finalize_kernel<false>, grid1((c).ldp), dim3(256), 0, (c).st, __VA_ARGS__, (const float*)nullptr, hin); \
  } while (0)
#define FINALIZE_LAUNCH(c, ...) FINALIZE_LAUNCH_H(c, HeadInArgs{}, __VA_ARGS__)

struct Ctx {
  hipStream_t st;
  const float* packed;
  const neat_net_params* net;
  int P, ldp, prec;
  int x3 = 0;                // NEAT_BF16X3: fp32 layouts, split-bf16 products (kernels.hpp, x3_mfma)
  int hx3 = 0;               // NEAT_F16X3: 16-bit layouts and backward, 3-product forward chains (kernels_x3.hpp)
  const PackLayout& L() const { return pack_layout(prec, hx3 != 0); }
  const float* rowscale(int l) const { return packed + L().rowscale_off + L().row_off[l]; }
};

// an array in the workspace: fp32 feature-major, or (big hidden activations in the bf16 build) bf16 octet-major
struct Arr {
  void* p = nullptr; int bf16 = 0;
  float* f() const { return reinterpret_cast<float*>(p); }
};
inline Arr F(float* p) { Arr a; a.p = p; a.bf16 = 0; return a; }
inline Arr F(const float* p) { return F(const_cast<float*>(p)); }
struct In { Arr a; int rows; };
inline In in(Arr a, int rows) { return In{a, rows}; }
const In NOIN = In{Arr{}, 0};

// out[n][p] = epi(Wm in + bias) with Wm = pack `pid`; N <= pack rows (only the leading rows are computed)
hipError_t layer(const Ctx& c, int pid, int epi, In in0, In in1, const float* bias, int N, Arr out0, Arr out1 = Arr{},
                 int n_split = 1 << 30, Arr aux0 = Arr{}, Arr aux1 = Arr{}, int accumulate = 0, int bias_rot = 0, int bias_n = 1 << 30,
                 const float* padfill = nullptr, int padfill_rows = 0, int tile0 = 0, Arr padfill_oct = Arr{}) {
  // tile0: compute packed rows [32 tile0, 32 tile0 + N) only (bf16 build: the small-input rows of the transposed head layer)
  const PackDesc2& d = c.L().d[pid];
  const float* wp = c.packed + d.offset + (size_t)tile0 * (d.Kpad / 16) * 64 * 4;
  const int k_in = (c.prec && in1.rows > 0 ? pad8(in0.rows) : in0.rows) + in1.rows;
  if (k_in != d.K || N + 32 * tile0 > d.N || (tile0 && !c.prec)) return hipErrorInvalidValue;
  // algorithmic flops (true N, K, P) and algorithmic HBM bytes (every operand row once, weights once)
  double bytes = 0.0;
  {
    const double P = (double)c.P;
    auto esz = [](const Arr& a) { return a.bf16 ? 2.0 : 4.0; };
    bytes += in0.rows * P * esz(in0.a) + in1.rows * P * esz(in1.a);
    const int n0rows = N < n_split ? N : n_split;
    bytes += n0rows * P * esz(out0) * (accumulate ? 2.0 : 1.0);
    if (out1.p) bytes += (epi == EPI_TAN ? N : (N > n_split ? N - n_split : 0)) * P * esz(out1);
    if (aux0.p) bytes += n0rows * P * esz(aux0);
    if (aux1.p) bytes += n0rows * P * esz(aux1);
    bytes += (double)N * (in0.rows + in1.rows) * (c.prec ? 2.0 : 4.0);
  }
  ProfSlot* ps = prof_begin(c.st, 0, 2.0 * N * (in0.rows + in1.rows) * (double)c.P, bytes);
  hipError_t e;
  if (!c.prec) {
    if (in0.a.bf16 || in1.a.bf16 || out0.bf16 || aux0.bf16 || aux1.bf16) return hipErrorInvalidValue;
    LayerArgs a;
    a.in0 = in0.a.f(); a.in1 = in1.a.f(); a.rows0 = in0.rows; a.rows1 = in1.rows;
    a.Kpad = d.Kpad; a.Wp = wp; a.bias = bias;
    a.N = N; a.NT = tiles32(N); a.ldp = c.ldp;
    a.out0 = out0.f(); a.out1 = out1.f(); a.n_split = n_split; a.accumulate = accumulate;
    a.aux0 = aux0.f(); a.aux1 = aux1.f(); a.x3 = c.x3;
    e = dispatch_f(c.st, epi, a, c.ldp / BM);
  } else {
    LayerArgsH a;
    a.in[0] = SegH{in0.a.p, in0.rows, in0.a.bf16}; a.in[1] = SegH{in1.a.p, in1.rows, in1.a.bf16};
    a.Kpad = d.Kpad; a.Wp = reinterpret_cast<const uint4*>(wp);
    a.bias = bias; a.bias_rot = bias_rot; a.bias_n = bias_n;
    a.N = N; a.NT = tiles32(N); a.ldp = c.ldp;
    a.out0 = out0.p; a.out1 = out1.p; a.out0_bf16 = out0.bf16; a.out1_bf16 = out1.bf16;
    a.n_split = n_split; a.accumulate = accumulate;
    a.aux0 = reinterpret_cast<const u16*>(aux0.p); a.aux1 = reinterpret_cast<const u16*>(aux1.p);
    a.padfill = padfill; a.padfill_rows = padfill_rows; a.padfill_oct = reinterpret_cast<const u16*>(padfill_oct.p);
    if ((aux0.p && !aux0.bf16) || (aux1.p && !aux1.bf16) || (out1.p && (out1.bf16 != (epi == EPI_TAN)))) return hipErrorInvalidValue;
    const int shape = ws_shape(epi, a);
    e = shape ? dispatch_ws(c.st, epi, a, shape) : dispatch_h(c.st, epi, a, c.ldp / BMH);
  }
  prof_end(c.st, ps);
  dbg_sync(c.st, "layer pid/epi/N", pid, epi, N);
  return e;
}

// a hidden-size layer straight on the weight-stationary kernel (bf16 build; rows [0, N) of pack `pid`, K = 256 columns)
hipError_t layer_ws(const Ctx& c, int pid, int epi, Arr in0, int N, Arr out0, Arr aux0 = Arr{}, Arr aux1 = Arr{},
                    const float* srow = nullptr, const float* wrow = nullptr, const float* wrow_scale = nullptr,
                    const float* rho_num = nullptr, const float* rho_den = nullptr) {
  const PackDesc2& d = c.L().d[pid];
  if (!c.prec || !in0.bf16 || !out0.bf16 || N > 256 || N > d.N || d.Kpad < 256) return hipErrorInvalidValue;
  LayerArgsWS a{};
  a.in = reinterpret_cast<const u16*>(in0.p); a.Wp = reinterpret_cast<const uint4*>(c.packed + d.offset);
  a.aux0 = reinterpret_cast<const u16*>(aux0.p); a.aux1 = reinterpret_cast<const u16*>(aux1.p);
  a.out0 = reinterpret_cast<u16*>(out0.p);
  a.srow = srow; a.wrow = wrow; a.wrow_scale = wrow_scale; a.rho_num = rho_num; a.rho_den = rho_den;
  a.N = N; a.in_octs = 32; a.ldp = c.ldp; a.kstride = d.Kpad / 16;
  a.split_oct = 32; a.n_split = 1 << 30;
  const double P = (double)c.P;
  const double bytes = 256 * P * 2.0 + N * P * 2.0 * (1 + (aux0.p ? 1 : 0) + (aux1.p ? 1 : 0)) + (srow ? P * 4.0 : 0.0) + N * 256 * 2.0;
  ProfSlot* ps = prof_begin(c.st, 0, 2.0 * N * 256 * P, bytes);
  hipError_t e = dispatch_ws_args(c.st, epi, a);
  prof_end(c.st, ps);
  dbg_sync(c.st, "layer_ws pid/epi/N", pid, epi, N);
  return e;
}

inline dim3 grid1(int n, int b = 256) { return dim3((n + b - 1) / b); }

// ------------------------------------------------------------------------------------------------
// workspaces (float offsets; every array is [rows][ldp] floats of space; bf16 arrays use the first half)
// ------------------------------------------------------------------------------------------------
struct SdfWs {
  float *x, *E, *sdfraw, *sdf, *mask, *g, *e0, *es, *Eh, *gh, *abar8, *ones, *partial;   // fp32 feature-major
  Arr featc;                      // bf16 build: cotangent of the 256 feature rows of lin8 (octet-major); abar8 row 0 keeps the sdf row
  Arr Ebf, Ebf4, Ehbf, Ehbf4;     // bf16 build: octet-major copies of the PE rows 0..38 / 7..38 and of their tangents
  Arr h[9], feat, u[8], vh[9], m[8];                                                   // big (bf16 in the bf16 build)
  Arr hlo[9], featlo;             // HX3: lo planes of h_1..h_8 (read by the adjoint chain) and of the feature rows (read by the heads)
  float *dwp, *dwscale, *dwbias;  // 16-bit builds: block-scaled f16 partials of the in-kernel weight gradients (kernels_dw.hpp): 14 jobs
  size_t total;
};
constexpr int DW_JOBS = 15;                 // (layer 1..7) x (tangent pair, reverse pair) + lin8's reverse pair
constexpr int WSPLIT = 128;                 // fp32 build: point-splits of the weight-gradient reduction
constexpr int WLDN = 384, WLDK = 384;       //             partial tile leading dims (>= 296+1, multiple of 128)
constexpr int W2SPLIT = 256;                // bf16 build (256x256 tiles): splits and leading dims
constexpr int W2LDN = 288, W2LDK = 320;
constexpr int W2_LDS_BYTES = 2 * 256 * HLD + 8 * 8 * 64 * 16;      // operand tiles + raw DMA ring
constexpr int WGROUPS = 8;                  // stage-1 groups of the split reduction
constexpr size_t WSTAGE_FLOATS = (size_t)WGROUPS * WREDUCE_BATCH * 256 * 264;   // >= WGROUPS*WLDN*WLDK; the group sums of up to 6 batched 256x(256+1) layers
constexpr size_t WPARTIAL_FLOATS = (size_t)W2SPLIT * W2LDN * W2LDK + WSTAGE_FLOATS;   // >= WSPLIT*WLDN*WLDK + stage

SdfWs sdf_ws(float* base, int ldp, int mode, int prec, int hx3 = 0) {
  SdfWs w{};
  size_t off = 0;
  auto take = [&](int rows) { float* p = base ? base + off : nullptr; off += (size_t)rows * ldp; return p; };
  auto big = [&](int rows) { Arr a; a.p = take(rows); a.bf16 = prec; return a; };
  w.x = take(3); w.E = take(PE_ROWS + 1); w.sdf = take(1); w.mask = take(1); w.g = take(3); w.sdfraw = take(1);
  if (mode == 0) {
    Arr a = big(256), b = big(256);
    for (int l = 1; l <= 8; ++l) w.h[l] = (l & 1) ? a : b;
  } else if (mode == 2) {
    // forward only (eval / inference, neat-final-parsing.py:203-218): h_1..h_8 are kept for the adjoint chain (the normals feed both
    // heads and the normal map), the adjoint's u_l ping-pong between two buffers, nothing is kept for a backward pass
    for (int l = 1; l <= 8; ++l) w.h[l] = big(256);
    if (prec) w.feat = big(256);
    else { float* out8 = take(257); w.sdfraw = out8; w.feat = F(out8 + ldp); }
    Arr a = big(256), b = big(256);
    for (int l = 0; l < 8; ++l) w.u[l] = (l & 1) ? a : b;
    w.e0 = take(PE_ROWS); w.es = take(PE_ROWS);
  } else {
    for (int l = 1; l <= 8; ++l) w.h[l] = big(256);
    if (prec) w.feat = big(256);
    else { float* out8 = take(257); w.sdfraw = out8; w.feat = F(out8 + ldp); }      // fp32: lin8 output [sdf | feature] in place
    for (int l = 0; l < 8; ++l) w.u[l] = big(256);
    w.e0 = take(PE_ROWS); w.es = take(PE_ROWS);
    w.Eh = take(PE_ROWS); w.gh = take(3);
    for (int l = 1; l <= 8; ++l) w.vh[l] = big(256);
    for (int l = 0; l < 8; ++l) w.m[l] = big(256);
    w.abar8 = take(257); w.ones = take(1);
    if (prec) { w.Ebf = big(20); w.Ebf4 = big(16); w.Ehbf = big(20); w.Ehbf4 = big(16); w.featc = big(256); }
    w.partial = base ? base + off : nullptr;
    off += WPARTIAL_FLOATS;
    if (prec) {
      // (sized by the workgroups a launch can have at this point count, not by the machine: a 2048-point step reserves 120 MB, not 480)
      const size_t slots = (size_t)dw_slots(ldp);
      w.dwp = base ? base + off : nullptr; off += (size_t)DW_JOBS * slots * DW_WG_UINT4 * 4;
      w.dwscale = base ? base + off : nullptr; off += (size_t)DW_JOBS * slots * 8;
      w.dwbias = base ? base + off : nullptr; off += (size_t)DW_JOBS * slots * 256;
    }
  }
  if (hx3 && mode != 0) {
    for (int l = 1; l <= 8; ++l) w.hlo[l] = big(128);      // (a 16-bit plane of 256 rows = 128 float rows)
    w.featlo = big(128);
  }
  w.total = off;
  return w;
}

struct HeadWs {
  float *small_r, *small_a, *rgb, *lin, *zrgb, *dlin, *sc_r, *sc_a;      // fp32
  Arr hr[5], ha[5], ar[4], aa[4];                                       // big
  Arr smallbf_r, smallbf_a, topbf_r, topbf_a;                           // bf16 build: octet-major copies of small_* / zrgb / dlin
  float *mr[5], *ma[5];                                                 // 16-bit builds: ReLU masks of hr / ha [1..4] (fused head chains), one word per 16 elements
  size_t total;
};
HeadWs head_ws(float* base, int ldp, int prec, bool fwd_only = false) {
  HeadWs w{};
  size_t off = 0;
  auto take = [&](int rows) { float* p = base ? base + off : nullptr; off += (size_t)rows * ldp; return p; };
  auto big = [&](int rows) { Arr a; a.p = take(rows); a.bf16 = prec; return a; };
  w.small_r = take(SMALL_R); w.small_a = take(SMALL_A);
  if (fwd_only) {      // hidden activations of both heads ping-pong between two buffers; no cotangent arrays
    Arr a = big(256), b = big(256);
    for (int l = 1; l <= 4; ++l) { w.hr[l] = (l & 1) ? a : b; w.ha[l] = (l & 1) ? a : b; }
    w.rgb = take(3); w.lin = take(6);
    if (prec) { w.smallbf_r = big(20); w.smallbf_a = big(8); }
    w.total = off;
    return w;
  }
  for (int l = 1; l <= 4; ++l) { w.hr[l] = big(256); w.ha[l] = big(256); }
  w.rgb = take(3); w.lin = take(6);
  w.zrgb = take(3); w.dlin = take(6);
  for (int l = 0; l < 4; ++l) { w.ar[l] = big(256); w.aa[l] = big(256); }
  w.sc_r = take(SMALL_R); w.sc_a = take(SMALL_A);
  if (prec) { w.smallbf_r = big(20); w.smallbf_a = big(8); w.topbf_r = big(4); w.topbf_a = big(4); }
  if (prec) for (int l = 1; l <= 4; ++l) { w.mr[l] = take(16); w.ma[l] = take(16); }
  w.total = off;
  return w;
}

inline int round_ldp(int P, int prec) { const int t = prec ? BMH : BM; return (P + t - 1) / t * t; }

// ------------------------------------------------------------------------------------------------
// SDF network chains
// ------------------------------------------------------------------------------------------------
// primal chain  (ImplicitNetwork.forward, rend_a :78-96); full = also the 256 feature rows and everything backward needs
// sdf_out (values mode only): clamped sdf, row-major
hipError_t sdf_primal(const Ctx& c, const SdfWs& w, bool full, float radius = 0.f, float scale = 0.f, float* sdf_out = nullptr) {
  const PackLayout& L = c.L();
  if (c.prec) {
    // bf16 build: ONE fused launch, activation tile resident in LDS
    FusedArgs a{};
    a.x_fm = w.x; a.P = c.P; a.ldp = c.ldp;
    for (int l = 0; l < 9; ++l) {
      const PackDesc2& d = L.d[(l == 8 && !full) ? L.sdf_row : L.fwd[l]];
      a.Wp[l] = reinterpret_cast<const uint4*>(c.packed + d.offset);
      a.KS[l] = d.Kpad >> 4;
      a.bias[l] = c.net->b[l];
      a.h[l] = l ? reinterpret_cast<u16*>(w.h[l].p) : nullptr;
    }
    a.save = full; a.values_only = !full;
    a.E = w.E; a.feat = reinterpret_cast<u16*>(w.feat.p); a.sdfraw = w.sdfraw; a.sdf_out = sdf_out;
    a.radius = radius; a.scale = scale; a.bias8_rot = 1; a.bias8_n = 257;
    a.gate = g_gate; a.gate_value = g_gate_value;
    if (c.hx3 == 2 && full) return hipErrorInvalidValue;
    if (c.hx3 == 1) {
      // split-precision forward: PE rows by posenc6_kernel (libm sin / cos: the hardware forms are ~1e-6 off at |arg| ~ 100), then
      // ONE launch of the 3-product chain, which also writes the lo planes the adjoint chain and the heads read
      for (int l = 0; l < 9; ++l) {
        const PackDesc2& d = L.d[(l == 8 && !full) ? L.sdf_row_lo : L.fwd_lo[l]];
        a.Wlo[l] = reinterpret_cast<const uint4*>(c.packed + d.offset);
        a.hlo[l] = (l && full) ? reinterpret_cast<u16*>(w.hlo[l].p) : nullptr;
      }
      a.featlo = full ? reinterpret_cast<u16*>(w.featlo.p) : nullptr;
      hipLaunchKernelGGL(posenc6_kernel, grid1(c.ldp), dim3(256), 0, c.st, w.x, c.ldp, w.E);
      double fl3 = 0.0;
      for (int l = 0; l < 8; ++l) fl3 += 2.0 * kO[l] * kI[l] * (double)c.P;
      fl3 += 2.0 * (full ? 257 : 1) * 256 * (double)c.P;
      ProfSlot* ps3 = prof_begin(c.st, 2, fl3, (double)c.P * (12.0 + 39 * 4.0 + (full ? 2.0 * (7 * 256 + 224 + 256) * 2.0 + 4.0 : 4.0)) + 4.0 * 589000.0);
      hipError_t e3 = launch_sdf_chain_x3(c.st, a, c.ldp / X3_BATCH, g_ws_grid, full);
      prof_end(c.st, ps3);
      return e3;
    }
    constexpr int PT = 2;
    const size_t lds = (size_t)(32 + 8) * (32 * PT) * 16;
    double fl = 0.0;
    for (int l = 0; l < 8; ++l) fl += 2.0 * kO[l] * kI[l] * (double)c.P;
    fl += 2.0 * (full ? 257 : 1) * 256 * (double)c.P;
    const double fbytes = (double)c.P * (12.0 + (full ? (39 * 4.0 + (7 * 256 + 224 + 256) * 2.0 + 4.0) : 4.0)) + 2.0 * 589000.0;
    ProfSlot* ps = prof_begin(c.st, 2, fl, fbytes);
    if (g_fused_ws) {
      // weight-stationary persistent kernel: batches of 128 points (64 when that balances the CUs better)
      static DevOnce attr_done;      // raise the dynamic-LDS limit of all variants once (not a stream operation: keep it out of graph capture)
      if (!attr_done) {
        hipError_t e0 = hipSuccess;
        auto raise = [&](auto kern, int bytes) { if (e0 == hipSuccess) e0 = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes); };
        raise(&sdf_fused_ws_kernel<4, false>, FwsCfg<4>::LDS); raise(&sdf_fused_ws_kernel<4, true>, FwsCfg<4>::LDS);
        raise(&sdf_fused_ws_kernel<3, false>, FwsCfg<3>::LDS); raise(&sdf_fused_ws_kernel<3, true>, FwsCfg<3>::LDS);
        raise(&sdf_fused_ws_kernel<2, false>, FwsCfg<2>::LDS); raise(&sdf_fused_ws_kernel<2, true>, FwsCfg<2>::LDS);
        if (e0 != hipSuccess) return e0;
        attr_done = true;
      }
      const int ntiles = c.ldp / 32;                 // ldp is a multiple of 64
      const int nwg = ntiles < g_ws_grid ? ntiles : g_ws_grid;
      if (g_fused_ws >= 2) {
        hipError_t e6 = launch_sdf_fused_w64(c.st, a, ntiles, nwg, full, g_fused_interleave != 0, g_fused_ws == 2 ? 64 : (g_fused_ws == 3 ? 32 : 0));
        prof_end(c.st, ps);
        return e6;
      }
      auto go = [&](auto kern, int BP, int lds_bytes) -> hipError_t {
        hipLaunchKernelGGL(kern, dim3(nwg), dim3(FWT), lds_bytes, c.st, a, ntiles, g_fused_interleave ? -nwg : nwg);
        return hipGetLastError();
      };
      // batch = up to 32*NT points: larger batches re-read the weights from L2 less often.  The tiles are split evenly over
      // the workgroups; the busiest one walks ceil(ntiles / nwg) tiles as full batches + one shorter batch
      auto cost = [&](int BP) {
        const int tiles = (ntiles + nwg - 1) / nwg, nt = BP / 32, rem = tiles % nt;
        return (tiles / nt) * (BP + 48.0) + (rem ? rem * 32 + 48.0 : 0.0);     // per-point work + per-batch weight streaming, in point units
      };
      int nt_sel = g_fused_nt;
      // (NT = 4 has fewer batches still, but its save-mode variant spills registers and measures slower: tuning key 5 only)
      if (nt_sel == 0) nt_sel = cost(96) < cost(64) ? 3 : 2;
      hipError_t e;
      if (nt_sel == 4) e = full ? go(&sdf_fused_ws_kernel<4, false>, 128, FwsCfg<4>::LDS) : go(&sdf_fused_ws_kernel<4, true>, 128, FwsCfg<4>::LDS);
      else if (nt_sel == 3) e = full ? go(&sdf_fused_ws_kernel<3, false>, 96, FwsCfg<3>::LDS) : go(&sdf_fused_ws_kernel<3, true>, 96, FwsCfg<3>::LDS);
      else e = full ? go(&sdf_fused_ws_kernel<2, false>, 64, FwsCfg<2>::LDS) : go(&sdf_fused_ws_kernel<2, true>, 64, FwsCfg<2>::LDS);
      prof_end(c.st, ps);
      return e;
    }
    if (full) hipLaunchKernelGGL((sdf_fused_kernel_h<PT, false>), dim3(c.ldp / (32 * PT)), dim3(WG), lds, c.st, a);
    else hipLaunchKernelGGL((sdf_fused_kernel_h<PT, true>), dim3(c.ldp / (32 * PT)), dim3(WG), lds, c.st, a);
    prof_end(c.st, ps);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(posenc6_kernel, grid1(c.ldp), dim3(256), 0, c.st, w.x, c.ldp, w.E);
  hipError_t e;
  for (int l = 0; l < 8; ++l) {
    In a = l == 0 ? in(F(w.E), PE_ROWS) : in(w.h[l], l == 4 ? 217 : 256);
    In b = l == 4 ? in(F(w.E), PE_ROWS) : NOIN;
    if ((e = layer(c, L.fwd[l], EPI_SOFTPLUS, a, b, c.net->b[l], kO[l], w.h[l + 1])) != hipSuccess) return e;
  }
  if (!full) return layer(c, L.sdf_row, EPI_LINEAR, in(w.h[8], 256), NOIN, c.net->b[8], 1, F(w.sdfraw));
  return layer(c, L.fwd[8], EPI_LINEAR, in(w.h[8], 256), NOIN, c.net->b[8], 257, F(w.sdfraw));
}

// adjoint chain: u_l = d sdf_raw / d a_l, then e0/es = cotangent of the PE rows (autograd.grad at rend_a :121-127)
hipError_t sdf_adjoint(const Ctx& c, const SdfWs& w, bool save = true) {
  const PackLayout& L = c.L();
  if (c.prec && (g_fused_adj || c.hx3)) {
    // bf16 build: the seed and the eight transposed layers as ONE launch (sdf_adjoint_w64_kernel): u stays on chip between the
    // layers; `save` = the training pass, where the tangent chain and the weight gradients read u_0 .. u_7 later
    AdjArgs a{};
    a.P = c.P; a.ldp = c.ldp;
    for (int l = 0; l < 8; ++l) {
      const PackDesc2& d = L.d[L.tr[l]];
      if (d.Kpad != 256) return hipErrorInvalidValue;
      a.Wp[l] = reinterpret_cast<const uint4*>(c.packed + d.offset);
      a.u[l] = reinterpret_cast<u16*>(w.u[l].p);
    }
    for (int l = 1; l <= 8; ++l) a.h[l] = reinterpret_cast<const u16*>(w.h[l].p);
    a.w8 = c.net->v[8]; a.rs8 = c.rowscale(8); a.es = w.es; a.e0 = w.e0;
    const int ntiles = c.ldp / 32;
    const int nwg = ntiles < g_ws_grid ? ntiles : g_ws_grid;
    double fl = 0.0;
    for (int l = 0; l < 8; ++l) fl += 2.0 * kO[l] * kI[l] * (double)c.P;
    if (c.hx3) {
      for (int l = 0; l < 8; ++l) a.Wlo[l] = reinterpret_cast<const uint4*>(c.packed + L.d[L.tr_lo[l]].offset);
      for (int l = 1; l <= 8; ++l) a.hlo[l] = reinterpret_cast<const u16*>(w.hlo[l].p);
      ProfSlot* ps3 = prof_begin(c.st, 3, fl, (double)c.P * (2 * 8 * 512.0 + (save ? 8 * 512.0 : 0.0) + 2 * 39 * 4.0) + 4.0 * 589000.0);
      hipError_t e3 = launch_sdf_adjoint_x3(c.st, a, c.ldp / X3_BATCH, g_ws_grid, save);
      prof_end(c.st, ps3);
      return e3;
    }
    ProfSlot* ps = prof_begin(c.st, 3, fl, (double)c.P * (8 * 512.0 + (save ? 8 * 512.0 : 0.0) + 2 * 39 * 4.0) + 2.0 * 589000.0);
    hipError_t e = launch_sdf_adjoint_w64(c.st, a, ntiles, nwg, save);
    prof_end(c.st, ps);
    return e;
  }
  if (c.prec)
    hipLaunchKernelGGL(adjoint_seed_kernel_h, dim3((c.ldp + 255) / 256, 32), dim3(256), 0, c.st, c.net->v[8], c.rowscale(8),
                       reinterpret_cast<const u16*>(w.h[8].p), c.ldp, reinterpret_cast<u16*>(w.u[7].p));
  else
    hipLaunchKernelGGL(adjoint_seed_kernel, dim3((c.ldp + 255) / 256, 256), dim3(256), 0, c.st, c.net->v[8], c.rowscale(8),
                       w.h[8].f(), c.ldp, w.u[7].f());
  hipError_t e;
  for (int l = 7; l >= 1; --l) {
    if (l == 4) {
      e = layer(c, L.tr[l], EPI_REV, in(w.u[l], kO[l]), NOIN, nullptr, 256, w.u[l - 1], F(w.es), 217, w.h[l]);
    } else {
      e = layer(c, L.tr[l], EPI_REV, in(w.u[l], kO[l]), NOIN, nullptr, kI[l], w.u[l - 1], Arr{}, 1 << 30, w.h[l]);
    }
    if (e != hipSuccess) return e;
  }
  return layer(c, L.tr[0], EPI_LINEAR, in(w.u[0], 256), NOIN, nullptr, PE_ROWS, F(w.e0));
}

struct WPair { Arr A; int rowsA; int A_rot, A_mod; Arr B[3]; int rowsB[3]; };

struct PackJob { const float* src; int rows; Arr dst; };
void oct_pack(const Ctx& c, std::initializer_list<PackJob> jobs) {
  OctPackArgs a{};
  for (const PackJob& j : jobs) a.job[a.njobs++] = OctPackJob{j.src, j.rows, reinterpret_cast<u16*>(j.dst.p)};
  a.ldp = c.ldp;
  hipLaunchKernelGGL(oct_pack_kernel, dim3((c.ldp + 255) / 256), dim3(256), 0, c.st, a);
}
inline bool oct_operands(const Ctx& c) { return c.prec && g_wgrad_h3; }

// second half of a weight gradient: deterministic reduction of the split partials described by `r` (partial, splits, strides),
// un-permutation of the packed columns and the weight-norm backward into dv / dg / db of `layer_id`
hipError_t wgrad_reduce(const Ctx& c, const SdfWs& w, int layer_id, WreduceArgs r, int Nred, int N, int K, const neat_net_grads* gr) {
  const PackDesc2& d = c.L().d[c.L().fwd[layer_id]];
  const int splits = r.splits;
  const float* partial = r.partial;
  const bool direct = c.prec && g_wreduce_direct == 1;    // bf16 build: one 16-wave pass over all split partials
  const bool direct4 = c.prec && g_wreduce_direct == 2 && splits > 2 * WGROUPS && (r.split_stride & 3) == 0 &&
                       r.split_stride / 4 <= WRD_MAXK4;    // one launch, 16-byte loads (wreduce_direct_kernel)
  if (splits > 2 * WGROUPS && !direct && !direct4) {
    // two-stage, deterministic: bandwidth-bound group sums first, then the per-row finish on 8 partials
    const int Kld = (int)r.split_stride, per = (splits + WGROUPS - 1) / WGROUPS;
    float* stage = w.partial + (WPARTIAL_FLOATS - WSTAGE_FLOATS);
    hipLaunchKernelGGL(wpartial_group_sum_kernel, dim3((Kld / 4 + 127) / 128, WGROUPS, Nred), dim3(128), 0, c.st,
                       partial, splits, Kld / 4, WGROUPS, per, Nred, stage);
    r.partial = stage; r.splits = (splits + per - 1) / per;
    r.row_stride = (size_t)WGROUPS * Kld; r.split_stride = Kld;
  }
  r.O = kO[layer_id]; r.I = kI[layer_id];
  r.s0 = d.s0; r.s0p = d.s0p; r.off0 = d.off0; r.off1 = d.off1; r.rot = d.rot; r.scale = d.scale;
  r.v = c.net->v[layer_id]; r.g = c.net->g[layer_id];
  r.dv = gr->dv[layer_id]; r.dg = gr->dg[layer_id]; r.db = gr->db[layer_id];
  r.bias_col = K;
  dbg_sync(c.st, "wgrad layer/N/K", layer_id, N, K);
  if (direct4) hipLaunchKernelGGL(wreduce_direct_kernel, dim3(r.O), dim3(WG), 0, c.st, r);
  else if (direct && splits > 2 * WGROUPS) hipLaunchKernelGGL(wreduce_wnorm_kernel<16>, dim3(r.O), dim3(1024), 0, c.st, r);
  else hipLaunchKernelGGL(wreduce_wnorm_kernel<4>, dim3(r.O), dim3(WG), 0, c.st, r);
  dbg_sync(c.st, "wreduce layer/splits", layer_id, splits, 0);
  return hipGetLastError();
}

// several same-shaped weight gradients (K = 256 bf16 octet-major operands, up to two pairs each: hidden layers of the two heads,
// hidden SDF layers) in ONE launch of wgrad_kernel_h3: each of the nprob problems gets 1/nprob of the workgroups and nprob times
// the points per workgroup, so the machine is as full as with one problem but only 1/nprob of the fp32 partial tiles per layer
// are written and reduced
struct WProb { int layer_id; Arr A[2]; int rowsA[2]; Arr B[2]; };
hipError_t wgrad_multi(const Ctx& c, const SdfWs& w, const WProb* pb, int nprob, int npairs, const neat_net_grads* gr) {
  if (nprob < 2 || npairs < 1 || npairs > 2 || nprob * npairs > W3_SLOTS) return hipErrorInvalidValue;
  for (int p = 0; p < nprob; ++p) if (!gr->dv[pb[p].layer_id]) return hipErrorInvalidValue;
  const int N = 256, K = 256, Kld2 = (K + 1 + 7) / 8 * 8;
  const int per_prob = (W2SPLIT + nprob - 1) / nprob;
  int chunk = ((c.ldp + per_prob - 1) / per_prob + W3P - 1) / W3P * W3P;
  if (chunk < 2 * W3P) chunk = 2 * W3P;
  const int splits = (c.P + chunk - 1) / chunk;
  const size_t region = (size_t)N * splits * Kld2;
  if ((size_t)nprob * region > WPARTIAL_FLOATS - WSTAGE_FLOATS) return hipErrorInvalidValue;
  static DevOnce attr3;
  if (!attr3) {
    hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel_h3<4>), hipFuncAttributeMaxDynamicSharedMemorySize, W3_LDS_BYTES);
    if (e0 != hipSuccess) return e0;
    attr3 = true;
  }
  WgradArgsH3 a{};
  double flops = 0.0, bytes = 0.0;
  for (int p = 0; p < nprob; ++p)
    for (int q = 0; q < npairs; ++q) {
      if (!pb[p].A[q].bf16 || !pb[p].B[q].bf16 || pb[p].rowsA[q] > 256) return hipErrorInvalidValue;
      const int i = p * npairs + q;
      a.A[i] = reinterpret_cast<const unsigned short*>(pb[p].A[q].p); a.B[i] = reinterpret_cast<const unsigned short*>(pb[p].B[q].p);
      a.B2[i] = nullptr; a.rowsA[i] = pb[p].rowsA[q];
      flops += 2.0 * pb[p].rowsA[q] * 256.0 * (double)c.P;
      bytes += (pb[p].rowsA[q] + 256.0) * (double)c.P * 2.0;
    }
  a.splitB = 32; a.octsB = 32; a.npairs = npairs; a.N = N; a.K = K; a.P = c.P; a.ldp = c.ldp; a.chunk = chunk;
  a.partial = w.partial; a.row_stride = (size_t)splits * Kld2; a.split_stride = Kld2; a.col_off = 0; a.bias_col = K;
  a.nprob = nprob; a.prob_stride = region; a.interleave = g_ws_interleave != 0; a.nt_loads = (g_ws_aux_nt >> 2) & 1;
  ProfSlot* ps = prof_begin(c.st, 1, flops, bytes + (double)nprob * splits * N * (K + 1) * 4.0);
  hipLaunchKernelGGL(wgrad_kernel_h3<4>, dim3(nprob, splits), dim3(W3T), W3_LDS_BYTES, c.st, a);
  prof_end(c.st, ps);
  hipError_t e = hipGetLastError();
  if (splits <= 2 * WGROUPS || nprob > WREDUCE_BATCH) {        // few points: finish each layer on its own
    for (int q = 0; q < nprob && e == hipSuccess; ++q) {
      WreduceArgs r{};
      r.partial = w.partial + q * region; r.splits = splits; r.row_stride = (size_t)splits * Kld2; r.split_stride = Kld2;
      e = wgrad_reduce(c, w, pb[q].layer_id, r, N, N, K, gr);
    }
    return e;
  }
  if (e != hipSuccess) return e;
  // the reduction of all nprob layers in two launches: the regions are contiguous, so the group sums see nprob * N rows;
  // then one finish launch (weight-norm backward) with blockIdx.y = layer
  const bool direct4 = g_wreduce_direct == 2;
  const int per = (splits + WGROUPS - 1) / WGROUPS, groups_used = (splits + per - 1) / per;
  float* stage = w.partial + (WPARTIAL_FLOATS - WSTAGE_FLOATS);
  if (!direct4)
    hipLaunchKernelGGL(wpartial_group_sum_kernel, dim3((Kld2 / 4 + 127) / 128, WGROUPS, nprob * N), dim3(128), 0, c.st,
                       (const float*)w.partial, splits, Kld2 / 4, WGROUPS, per, nprob * N, stage);
  WreduceBatch b{};
  for (int q = 0; q < nprob; ++q) {
    const int layer_id = pb[q].layer_id;
    const PackDesc2& d = c.L().d[c.L().fwd[layer_id]];
    WreduceArgs& r = b.a[q];
    if (direct4) {
      r.partial = w.partial + (size_t)q * region; r.splits = splits; r.row_stride = (size_t)splits * Kld2; r.split_stride = Kld2;
    } else {
      r.partial = stage + (size_t)q * N * WGROUPS * Kld2; r.splits = groups_used;
      r.row_stride = (size_t)WGROUPS * Kld2; r.split_stride = Kld2;
    }
    r.O = kO[layer_id]; r.I = kI[layer_id];
    r.s0 = d.s0; r.s0p = d.s0p; r.off0 = d.off0; r.off1 = d.off1; r.rot = d.rot; r.scale = d.scale;
    r.v = c.net->v[layer_id]; r.g = c.net->g[layer_id];
    r.dv = gr->dv[layer_id]; r.dg = gr->dg[layer_id]; r.db = gr->db[layer_id];
    r.bias_col = K;
  }
  if (direct4) hipLaunchKernelGGL(wreduce_direct_batch_kernel, dim3(256, nprob), dim3(WG), 0, c.st, b);
  else hipLaunchKernelGGL(wreduce_wnorm_batch_kernel, dim3(256, nprob), dim3(WG), 0, c.st, b);
  return hipGetLastError();
}

struct RowDot { const float* s; Arr B0, B1; };     // one more gradient row: sum_p s[p] B0[k][p] + B1[k][p] (the sdf row of lin8)

hipError_t wgrad(const Ctx& c, const SdfWs& w, int layer_id, const WPair* pairs_in, int npairs, int N, const neat_net_grads* gr,
                 const RowDot* rd = nullptr) {
  if (!gr->dv[layer_id]) return hipSuccess;
  const PackDesc2& d = c.L().d[c.L().fwd[layer_id]];
  const int K = d.s0p + (kI[layer_id] - d.s0);      // packed column count of the B operand
  double wflops = 0.0;
  for (int q = 0; q < npairs; ++q)
    wflops += 2.0 * pairs_in[q].rowsA * (pairs_in[q].rowsB[0] + pairs_in[q].rowsB[1] + pairs_in[q].rowsB[2]) * (double)c.P;
  double wbytes = 0.0;
  for (int q = 0; q < npairs; ++q) {
    wbytes += pairs_in[q].rowsA * (double)c.P * (pairs_in[q].A.bf16 ? 2.0 : 4.0);
    for (int t = 0; t < 3; ++t) wbytes += pairs_in[q].rowsB[t] * (double)c.P * (pairs_in[q].B[t].bf16 ? 2.0 : 4.0);
  }
  WreduceArgs r{};
  int splits, chunk_used = 0;
  if (!c.prec) {
    // fp32: 128x128 tiles; the bias gradient is the column of the `ones` row appended to pair 0's B
    WPair pairs[2] = {pairs_in[0], npairs > 1 ? pairs_in[1] : WPair{}};
    for (int s = 0; s < 3; ++s)
      if (pairs[0].rowsB[s] == 0) { pairs[0].B[s] = F(w.ones); pairs[0].rowsB[s] = 1; break; }
    const int Kt = K + 1;
    int chunk = ((c.ldp + WSPLIT - 1) / WSPLIT + WBP - 1) / WBP * WBP;
    if (chunk < 2 * WBP) chunk = 2 * WBP;
    splits = (c.P + chunk - 1) / chunk;
    const int ntile = (N + 127) / 128, ktiles = (Kt + 127) / 128;
    WgradArgs a{};
    for (int q = 0; q < npairs; ++q) {
      a.pair[q].A = pairs[q].A.f(); a.pair[q].rowsA = pairs[q].rowsA;
      for (int s = 0; s < 3; ++s) { a.pair[q].B[s] = pairs[q].B[s].f(); a.pair[q].rowsB[s] = pairs[q].rowsB[s]; }
    }
    a.npairs = npairs; a.N = N; a.Kt = Kt; a.P = c.P; a.ldp = c.ldp; a.chunk = chunk;
    a.partial = w.partial; a.row_stride = (size_t)splits * WLDK; a.split_stride = WLDK; a.ktiles = ktiles; a.x3 = c.x3;
    ProfSlot* ps = prof_begin(c.st, 1, wflops, wbytes + (double)splits * N * Kt * 4.0);
    hipLaunchKernelGGL(wgrad_kernel, dim3(ntile * ktiles, splits), dim3(WG), 0, c.st, a);
    prof_end(c.st, ps);
    r.row_stride = (size_t)splits * WLDK; r.split_stride = WLDK;
  } else {
    static DevOnce attr_set;
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel_h2), hipFuncAttributeMaxDynamicSharedMemorySize, W2_LDS_BYTES);
      if (e != hipSuccess) return e;
      attr_set = true;
    }
    const int Kld2 = (K + 1 + 7) / 8 * 8;      // partial row length: K columns + bias, not the 320 of the widest layer (the reduction reads all of it)
    if (Kld2 > W2LDK) return hipErrorInvalidValue;
    // all-bf16 octet-major operands: the streaming tr16 kernel.  Shapes: K <= 256 in one launch (B may continue in a
    // second array at an octet boundary: skip layer), or [256 | few rows] as two launches into disjoint partial columns
    bool h3 = g_wgrad_h3 && N <= 256;
    for (int q = 0; q < npairs && h3; ++q) {
      const WPair& s = pairs_in[q];
      h3 = s.A.bf16 && s.A_mod == 0 && s.rowsA <= 256 && s.B[0].bf16 && s.rowsB[2] == 0 &&
           (s.rowsB[1] == 0 || (s.B[1].bf16 && s.rowsB[0] % 8 == 0)) &&
           s.rowsB[0] == pairs_in[0].rowsB[0] && s.rowsB[1] == pairs_in[0].rowsB[1];
    }
    const int r0 = pairs_in[0].rowsB[0], r1 = pairs_in[0].rowsB[1];
    // [256 | r1 <= 64 rows] (the heads' input layers): ONE launch of the five-column-block variant (g_wgrad_k320, tuning key 19) -- two
    // launches, [256] and [r1] into disjoint partial columns, each read the whole A operand
    const bool wide = h3 && g_wgrad_k320 && K > 256 && K <= 320 && r0 == 256 && r1 > 0 && r1 <= 64;
    const bool two = K > 256 && !wide;
    if (two && (r0 != 256 || r1 == 0 || r1 > 256)) h3 = false;
    const bool narrow = h3 && g_wgrad_narrow && K <= 64 && r1 == 0;      // lin0: one 32-column block per wave (tuning key 21)
    if (h3) {
      static DevOnce attr3_set;
      if (!attr3_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel_h3<4>), hipFuncAttributeMaxDynamicSharedMemorySize, W3_LDS_BYTES);
        if (e != hipSuccess) return e;
        attr3_set = true;
      }
      // (narrow: 80 KB of LDS per workgroup -> two co-resident workgroups per CU when there are twice as many splits)
      const int nsplit_target = narrow ? W2SPLIT * g_wgrad_narrow : W2SPLIT;
      int chunk = ((c.ldp + nsplit_target - 1) / nsplit_target + W3P - 1) / W3P * W3P;
      if (chunk < 2 * W3P) chunk = 2 * W3P;
      splits = (c.P + chunk - 1) / chunk;
      chunk_used = chunk;
      ProfSlot* ps = prof_begin(c.st, 1, wflops, wbytes + (double)splits * N * (K + 1) * 4.0);
      if (wide) {
        static DevOnce attr5_set;
        if (!attr5_set) {
          hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel_h3<5>), hipFuncAttributeMaxDynamicSharedMemorySize, W3Cfg<5>::LDS);
          if (e != hipSuccess) return e;
          attr5_set = true;
        }
      }
      if (narrow) {
        static DevOnce attr1_set;
        if (!attr1_set) {
          hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel_h3<1>), hipFuncAttributeMaxDynamicSharedMemorySize, W3Cfg<1>::LDS);
          if (e != hipSuccess) return e;
          attr1_set = true;
        }
      }
      for (int part = 0; part < (two ? 2 : 1); ++part) {
        WgradArgsH3 a{};
        for (int q = 0; q < npairs; ++q) {
          const WPair& s = pairs_in[q];
          a.A[q] = reinterpret_cast<const unsigned short*>(s.A.p);
          a.B[q] = reinterpret_cast<const unsigned short*>(part ? s.B[1].p : s.B[0].p);
          a.B2[q] = reinterpret_cast<const unsigned short*>(s.B[1].p);
          a.rowsA[q] = s.rowsA;
        }
        if (part == 0) {
          a.K = two ? 256 : K;
          a.splitB = wide ? 32 : ((!two && r1 > 0) ? r0 / 8 : 32);
          a.col_off = 0; a.bias_col = K;
        } else {
          a.K = r1; a.splitB = 32; a.col_off = 256; a.bias_col = -1;
        }
        a.octsB = (a.K + 7) / 8;
        a.npairs = npairs; a.N = N; a.P = c.P; a.ldp = c.ldp; a.chunk = chunk;
        a.partial = w.partial; a.row_stride = (size_t)splits * Kld2; a.split_stride = Kld2; a.interleave = g_ws_interleave != 0; a.nt_loads = (g_ws_aux_nt >> 2) & 1;
        if (wide) hipLaunchKernelGGL(wgrad_kernel_h3<5>, dim3(1, splits), dim3(W3T), W3Cfg<5>::LDS, c.st, a);
        else if (narrow) hipLaunchKernelGGL(wgrad_kernel_h3<1>, dim3(1, splits), dim3(W3T), W3Cfg<1>::LDS, c.st, a);
        else hipLaunchKernelGGL(wgrad_kernel_h3<4>, dim3(1, splits), dim3(W3T), W3_LDS_BYTES, c.st, a);
      }
      prof_end(c.st, ps);
      r.row_stride = (size_t)splits * Kld2; r.split_stride = Kld2;
    } else {
    int chunk = ((c.ldp + W2SPLIT - 1) / W2SPLIT + HBP - 1) / HBP * HBP;
    if (chunk < 2 * HBP) chunk = 2 * HBP;
    splits = (c.P + chunk - 1) / chunk;
    chunk_used = chunk;
    const int ntile = (N + 255) / 256, ktiles = (K + 255) / 256;
    WgradArgsH a{};
    for (int q = 0; q < npairs; ++q) {
      const WPair& s = pairs_in[q];
      a.pair[q].A = SegH{s.A.p, s.rowsA, s.A.bf16}; a.pair[q].A_rot = s.A_rot; a.pair[q].A_mod = s.A_mod;
      for (int t = 0; t < 3; ++t) a.pair[q].B[t] = SegH{s.B[t].p, s.rowsB[t], s.B[t].bf16};
      a.pair[q].padB0 = s.B[0].bf16 ? pad8(s.rowsB[0]) : s.rowsB[0];
    }
    a.npairs = npairs; a.N = N; a.Kt = K; a.P = c.P; a.ldp = c.ldp; a.chunk = chunk;
    a.partial = w.partial; a.row_stride = (size_t)splits * Kld2; a.split_stride = Kld2; a.ktiles = ktiles; a.bias_col = K;
    ProfSlot* ps = prof_begin(c.st, 1, wflops, wbytes + (double)splits * N * (K + 1) * 4.0);
    hipLaunchKernelGGL(wgrad_kernel_h2, dim3(ntile * ktiles, splits), dim3(W2T), W2_LDS_BYTES, c.st, a);
    prof_end(c.st, ps);
    r.row_stride = (size_t)splits * Kld2; r.split_stride = Kld2;
    }
  }
  int Nred = N;
  if (rd) {            // extra row N of the partial tiles (bf16 build, K = 256)
    if (!c.prec || K != 256 || !rd->B0.bf16 || !rd->B1.bf16) return hipErrorInvalidValue;
    hipLaunchKernelGGL(rowdot_kernel, dim3(splits), dim3(256), 0, c.st, rd->s, reinterpret_cast<const u16*>(rd->B0.p),
                       reinterpret_cast<const u16*>(rd->B1.p), c.P, c.ldp, chunk_used, w.partial + (size_t)N * r.row_stride, r.split_stride);
    Nred = N + 1;
  }
  r.partial = w.partial; r.splits = splits;
  return wgrad_reduce(c, w, layer_id, r, Nred, N, K, gr);
}

// double backward + backward: w.gh (cotangent of normals, masked) and w.abar8 (cotangent of lin8 output) are set
// nc: the cotangent of the normals still has to be formed (its arguments); null = w.gh is set
struct NormalCot { const float* sc_r; const float* sc_a; const float* extra_rm; int P_main; const float* d_tail_rm; const float* slot; const float* slot_a;
                   const float* dbeta_ray = nullptr; int R = 0; const float* beta_ptr = nullptr; float* dbeta = nullptr; };
hipError_t sdf_backward_chains(const Ctx& c, const SdfWs& w, const neat_net_grads* gr, const NormalCot* nc = nullptr) {
  const PackLayout& L = c.L();
  hipError_t e;
  if (c.prec && nc) {
    // 16-bit builds: normal cotangent + PE tangent + the octet-major copies the streaming kernels read, one launch (bwd_prologue_kernel)
    BwdPrologueArgs a{};
    a.sc_r = nc->sc_r; a.sc_a = nc->sc_a; a.extra_rm = nc->extra_rm; a.mask = w.mask; a.d_tail_rm = nc->d_tail_rm;
    a.cot_slot = nc->slot; a.cot_slot_a = nc->slot_a; a.P = c.P; a.ldp = c.ldp; a.P_main = nc->P_main;
    a.x_fm = w.x; a.E = w.E; a.gh = w.gh; a.Eh = w.Eh;
    a.Ebf = reinterpret_cast<u16*>(w.Ebf.p); a.Ebf4 = reinterpret_cast<u16*>(w.Ebf4.p);
    a.Ehbf = reinterpret_cast<u16*>(w.Ehbf.p); a.Ehbf4 = reinterpret_cast<u16*>(w.Ehbf4.p);
    a.dbeta_ray = nc->dbeta_ray; a.R = nc->R; a.beta_ptr = nc->beta_ptr; a.dbeta = nc->dbeta;
    hipLaunchKernelGGL(bwd_prologue_kernel, dim3(grid1(c.ldp).x + (a.dbeta ? 1 : 0)), dim3(256), 0, c.st, a);
  } else {
    if (nc) hipLaunchKernelGGL(normal_cotangent_kernel, grid1(c.ldp), dim3(256), 0, c.st, nc->sc_r, nc->sc_a, nc->extra_rm, w.mask, c.P, c.ldp,
                               w.gh, nc->P_main, nc->d_tail_rm, nc->slot, nc->slot_a);
    if (c.prec) hipLaunchKernelGGL(posenc6_tangent_kernel<true>, grid1(c.ldp), dim3(256), 0, c.st, w.x, w.gh, c.ldp, w.Eh);
    else hipLaunchKernelGGL(posenc6_tangent_kernel<false>, grid1(c.ldp), dim3(256), 0, c.st, w.x, w.gh, c.ldp, w.Eh);
    // bf16 build: octet-major copies of the PE rows and their tangents for the streaming kernels (skip layer, lin0 / lin4 gradients)
    if (c.prec) oct_pack(c, {{w.E, PE_ROWS, w.Ebf}, {w.E + 7 * (size_t)c.ldp, 32, w.Ebf4}, {w.Eh, PE_ROWS, w.Ehbf}, {w.Eh + 7 * (size_t)c.ldp, 32, w.Ehbf4}});
  }
  if (!c.prec) hipLaunchKernelGGL(ones_kernel, grid1(c.ldp), dim3(256), 0, c.st, w.ones, c.P, c.ldp);
  // tangent chain (forward-mode along g^): vh_{l+1} = tangent of h_{l+1}, m_l = extra cotangent of a_l
  // bf16 build: the 217-row arrays of the skip layer carry the first 7 PE rows in the padding of their last octet, so
  // lin4's input is [h4 | PE0..6] (224 rows, octet aligned) + PE7..38 (32 rows) = 256 columns
  // 16-bit builds: layers 1..7 run on the streaming kernels that also contract the layer's weight gradient on chip (kernels_dw.hpp):
  // the tangent launch of layer l leaves u_l (x) vhat_l, the reverse launch a^_l (x) in_l (transposed) and the bias gradient, as
  // block-scaled f16 partials per workgroup; one gather launch and one batched weight-norm finish follow the chains
  bool dw = c.prec && g_dw_fused && g_layer_ws && g_wgrad_h3 && c.ldp >= 2 * WSP && (g_dw_fused == 2 || c.ldp >= DW_MIN_POINTS);
  for (int l = 1; l <= 7; ++l) dw = dw && gr->dv[l] != nullptr && L.d[L.fwd[l]].Kpad == 256 && L.d[L.tr[l]].Kpad == 256;
  const int dwg = dw_grid(c.ldp);
  const bool dw8 = dw && g_dw_lin8 && gr->dv[8] != nullptr && L.d[L.tr[8]].Kpad >= 256 && w.featc.bf16;
  auto dw_job = [&](int l, int pair, LayerArgsDW& d) {
    const size_t j = (size_t)(2 * (l - 1) + pair);
    d.partial = reinterpret_cast<uint4*>(w.dwp) + j * dw_slots(c.ldp) * DW_WG_UINT4;
    d.pscale = w.dwscale + j * dw_slots(c.ldp) * 8;
    d.pbias = pair ? w.dwbias + j * dw_slots(c.ldp) * 256 : nullptr;
    d.P = c.P;
  };
  auto u16p = [](const Arr& a) { return reinterpret_cast<u16*>(a.p); };
  // chain variables of the in-kernel-gradient launches: vhat_1 (lin0's launch) and vhat_8 (rowdot_kernel reads it) keep their arrays, vhat_2..7
  // alternate between vh[2] / vh[3]; the reverse chain's a^_7..1 alternate between vh[4] / vh[5] (free in this mode), a^_0 lands in m[0] (lin0's gradient)
  const bool pp = dw && g_chain_pp;
  auto VH = [&](int l) -> const Arr& { return (pp && l >= 2 && l <= 7) ? w.vh[2 + (l & 1)] : w.vh[l]; };
  auto RB = [&](int l) -> const Arr& { return (pp && l >= 1 && l <= 7) ? w.vh[4 + (l & 1)] : w.m[l]; };
  auto dw_prof = [&](int N, int rowsA) {
    const double P = (double)c.P;
    // the layer (2 N 256 P) + the gradient (2 rowsA 256 P); bytes: input, two epilogue operands, outputs, the partial
    return prof_begin(c.st, 0, 2.0 * N * 256 * P + 2.0 * rowsA * 256 * P, 256 * P * 2.0 + N * P * 2.0 * 4 + (double)dwg * DW_WG_UINT4 * 16.0);
  };
  LayerArgsDW pend[DW_MAXSEG];
  int npend = 0, pend_variant = 0;
  double pend_flops = 0.0, pend_bytes = 0.0;
  auto flush_tan = [&]() -> hipError_t {
    if (!npend) return hipSuccess;
    ProfSlot* ps = prof_begin(c.st, 0, pend_flops, pend_bytes);
    const hipError_t r = pend_variant == 1 ? launch_layer_wsdw<EPI_TAN_PF, false>(c.st, pend, npend) : launch_layer_wsdw<EPI_TAN, true>(c.st, pend, npend);
    prof_end(c.st, ps);
    dbg_sync(c.st, "tangent+dW layers", npend, 0, 0);
    npend = 0; pend_flops = pend_bytes = 0.0;
    return r;
  };
  for (int l = 0; l < 8; ++l) {
    In a = l == 0 ? in(c.prec ? w.Ehbf : F(w.Eh), PE_ROWS) : in(w.vh[l], l == 4 ? (c.prec ? 224 : 217) : 256);
    In b = l == 4 ? (c.prec ? in(w.Ehbf4, 32) : in(F(w.Eh), PE_ROWS)) : NOIN;
    const bool fill = c.prec && l == 3;
    if (dw && l >= 1) {
      LayerArgsDW d{};
      LayerArgsWS& s = d.w;
      const PackDesc2& pd = L.d[L.fwd[l]];
      s.in = u16p(VH(l)); s.Wp = reinterpret_cast<const uint4*>(c.packed + pd.offset);
      s.aux0 = u16p(w.h[l + 1]); s.aux1 = u16p(w.u[l]); s.out0 = u16p(VH(l + 1)); s.out1 = u16p(w.m[l]);
      s.N = kO[l]; s.ldp = c.ldp; s.kstride = pd.Kpad / 16; s.n_split = 1 << 30;
      s.in_octs = 32; s.split_oct = 32;
      if (l == 4) { s.in2 = u16p(w.Ehbf4); s.split_oct = 28; }
      if (l == 3) s.padfill = u16p(w.Ehbf);
      d.auxA2 = nullptr; d.auxA_split = 1 << 30; d.rowsA = kO[l];
      dw_job(l, 0, d);
      // consecutive layers with the same epilogue variant go out as ONE launch (layer loop per workgroup, kernels_dw.hpp): 1-2 | 3 | 4-7
      const int variant = l == 3 ? 1 : 0;
      if (npend && (variant != pend_variant || npend == DW_MAXSEG || !g_dw_segments)) { if ((e = flush_tan()) != hipSuccess) return e; }
      pend[npend++] = d; pend_variant = variant; pend_flops += 2.0 * kO[l] * 256 * (double)c.P + 2.0 * kO[l] * 256 * (double)c.P;
      pend_bytes += 256 * (double)c.P * 2.0 + kO[l] * (double)c.P * 2.0 * 4 + (double)dwg * DW_WG_UINT4 * 16.0;
      if (l == 7 || !g_dw_segments) { if ((e = flush_tan()) != hipSuccess) return e; }
      continue;
    }
    if ((e = layer(c, L.fwd[l], EPI_TAN, a, b, nullptr, kO[l], w.vh[l + 1], w.m[l], 1 << 30, w.h[l + 1], w.u[l], 0, 0, 1 << 30,
                   fill ? w.Eh : nullptr, fill ? 7 : 0, 0, fill ? w.Ehbf : Arr{})) != hipSuccess) return e;
  }
  // reverse chain: a^_{l-1} = (W_l^T a^_l) phi'(a_{l-1}) + m_{l-1}   (in place in m)
  if (!c.prec) e = layer(c, L.tr[8], EPI_BWD, in(F(w.abar8), 257), NOIN, nullptr, 256, w.m[7], Arr{}, 1 << 30, w.h[8], w.m[7]);
  else if (dw8) {
    // the sdf row of dW8 first: its operands h8 and vh8 are what the last tangent launch just read / wrote (Infinity Cache)
    {
      const int K = 256, Kld2 = (K + 1 + 7) / 8 * 8, splits = g_dw_nsub;
      const size_t region = (size_t)256 * splits * Kld2, region8 = (size_t)257 * splits * Kld2;
      float* xrow = w.partial + (size_t)7 * region + region8;
      const int xchunk = ((c.ldp + DW8_XBLOCKS - 1) / DW8_XBLOCKS + 31) / 32 * 32;
      const int xn = (c.P + xchunk - 1) / xchunk;
      if ((size_t)7 * region + region8 + (size_t)DW8_XBLOCKS * Kld2 > WPARTIAL_FLOATS) return hipErrorInvalidValue;
      hipLaunchKernelGGL(rowdot_kernel, dim3(xn), dim3(256), 0, c.st, w.abar8, reinterpret_cast<const u16*>(w.h[8].p),
                         reinterpret_cast<const u16*>(w.vh[8].p), c.P, c.ldp, xchunk, xrow, (size_t)Kld2);
    }
    // lin8's reverse launch contracts featc (x) h8 (the 256 feature rows of dW8) and the row sums of featc on chip as well
    LayerArgsDW d{};
    LayerArgsWS& s = d.w;
    const PackDesc2& pd = L.d[L.tr[8]];
    s.in = u16p(w.featc); s.Wp = reinterpret_cast<const uint4*>(c.packed + pd.offset);
    s.aux0 = u16p(w.h[8]); s.aux1 = u16p(w.m[7]); s.out0 = u16p(RB(7));
    s.srow = w.abar8; s.wrow = c.net->v[8]; s.wrow_scale = c.rowscale(8);
    s.N = 256; s.ldp = c.ldp; s.kstride = pd.Kpad / 16; s.n_split = 1 << 30;
    s.in_octs = 32; s.split_oct = 32;
    d.auxA2 = nullptr; d.auxA_split = 1 << 30; d.rowsA = 256;
    const size_t j = 14;
    d.partial = reinterpret_cast<uint4*>(w.dwp) + j * dw_slots(c.ldp) * DW_WG_UINT4;
    d.pscale = w.dwscale + j * dw_slots(c.ldp) * 8;
    d.pbias = w.dwbias + j * dw_slots(c.ldp) * 256;
    d.P = c.P;
    ProfSlot* ps = dw_prof(256, 256);
    e = launch_layer_wsdw<EPI_BWD8, true>(c.st, d);
    prof_end(c.st, ps);
    dbg_sync(c.st, "reverse+dW layer", 8, 0, 0);
  }
  else if (g_layer_ws) e = layer_ws(c, L.tr[8], EPI_BWD8, w.featc, 256, w.m[7], w.h[8], w.m[7], w.abar8, c.net->v[8], c.rowscale(8));
  else e = layer(c, L.tr[8], EPI_BWD, in(w.featc, 256), in(F(w.abar8), 1), nullptr, 256, w.m[7], Arr{}, 1 << 30, w.h[8], w.m[7]);
  if (e != hipSuccess) return e;
  // weight gradients: dW_l = a^_l in_l^T + u_l vhat_l^T  (+ bias column).  The gradient of layer l is launched right
  // after the reverse step that produced a^_l (g_wgrad_interleave): a^_l is then read twice (weight gradient, next reverse
  // step) while it is still in the 256 MB Infinity Cache instead of after all eight reverse steps.
  const bool oct = oct_operands(c);
  auto wgrad_layer = [&](int l) -> hipError_t {
    WPair pr[2] = {};
    if (l == 8 && c.prec) {
      // feature rows: featc x h8 on the streaming kernel; the sdf row (cotangent abar8 row 0; its second-order part is
      // the plain row sum of vh8, the adjoint seed being 1) as one extra partial row
      pr[0].A = w.featc; pr[0].rowsA = 256; pr[0].B[0] = w.h[8]; pr[0].rowsB[0] = 256;
      RowDot rd{w.abar8, w.h[8], w.vh[8]};
      return wgrad(c, w, 8, pr, 1, 256, gr, &rd);
    }
    pr[0].A = l == 8 ? F(w.abar8) : w.m[l]; pr[0].rowsA = kO[l];
    pr[1].A = l == 8 ? F(w.ones) : w.u[l];  pr[1].rowsA = l == 8 ? 1 : kO[l];
    if (l == 0) {
      pr[0].B[0] = oct ? w.Ebf : F(w.E); pr[0].rowsB[0] = PE_ROWS;
      pr[1].B[0] = oct ? w.Ehbf : F(w.Eh); pr[1].rowsB[0] = PE_ROWS;
    } else if (l == 4) {
      if (c.prec) {
        pr[0].B[0] = w.h[4]; pr[0].rowsB[0] = 224; pr[0].B[1] = oct ? w.Ebf4 : F(w.E + 7 * (size_t)c.ldp); pr[0].rowsB[1] = 32;
        pr[1].B[0] = w.vh[4]; pr[1].rowsB[0] = 224; pr[1].B[1] = oct ? w.Ehbf4 : F(w.Eh + 7 * (size_t)c.ldp); pr[1].rowsB[1] = 32;
      } else {
        pr[0].B[0] = w.h[4]; pr[0].rowsB[0] = 217; pr[0].B[1] = F(w.E); pr[0].rowsB[1] = PE_ROWS;
        pr[1].B[0] = w.vh[4]; pr[1].rowsB[0] = 217; pr[1].B[1] = F(w.Eh); pr[1].rowsB[1] = PE_ROWS;
      }
    } else {
      pr[0].B[0] = w.h[l]; pr[0].rowsB[0] = 256;
      pr[1].B[0] = w.vh[l]; pr[1].rowsB[0] = 256;
    }
    return wgrad(c, w, l, pr, 2, kO[l], gr);
  };
  const bool inter = g_wgrad_interleave != 0 && !dw;
  if (inter && (e = wgrad_layer(8)) != hipSuccess) return e;
  auto flush_rev = [&]() -> hipError_t {
    if (!npend) return hipSuccess;
    ProfSlot* ps = prof_begin(c.st, 0, pend_flops, pend_bytes);
    const hipError_t r = pend_variant == 0 ? launch_layer_wsdw<EPI_BWD, true>(c.st, pend, npend) : launch_layer_wsdw<EPI_BWD, false>(c.st, pend, npend);
    prof_end(c.st, ps);
    dbg_sync(c.st, "reverse+dW layers", npend, 0, 0);
    npend = 0; pend_flops = pend_bytes = 0.0;
    return r;
  };
  for (int l = 7; l >= 1; --l) {
    if (inter && (e = wgrad_layer(l)) != hipSuccess) return e;
    const int N = l == 4 ? 217 : kI[l];
    if (dw) {
      LayerArgsDW d{};
      LayerArgsWS& s = d.w;
      const PackDesc2& pd = L.d[L.tr[l]];
      s.in = u16p(dw8 ? RB(l) : w.m[l]); s.Wp = reinterpret_cast<const uint4*>(c.packed + pd.offset);
      s.aux0 = u16p(w.h[l]); s.aux1 = u16p(w.m[l - 1]); s.out0 = u16p(dw8 ? RB(l - 1) : w.m[l - 1]);
      s.N = N; s.ldp = c.ldp; s.kstride = pd.Kpad / 16; s.n_split = 1 << 30;
      s.in_octs = (kO[l] + 7) / 8; s.split_oct = 32;
      // second gradient operand = the layer's input rows = the aux0 image: h_l, for the skip layer [h4 (217) | PE 0..6 in the padding | PE 7..38]
      d.auxA2 = l == 4 ? u16p(w.Ebf4) : nullptr; d.auxA_split = l == 4 ? 28 : (1 << 30);
      // rows of the gradient = rows of a^_l.  lin3's 217: the rows up to 223 of m[3] are zeros and the octets past them are re-reads of
      // octet 27 (finite), so rows >= 217 of the product are finite garbage that the gather never writes -- no row mask needed
      d.rowsA = 256;
      dw_job(l, 1, d);
      // 7-5 | 4 (217 output rows: the row-tested variant) | 3-1 as one launch each
      const int variant = (N == 256 && d.rowsA == 256) ? 0 : 1;
      if (npend && (variant != pend_variant || npend == DW_MAXSEG || !g_dw_segments)) { if ((e = flush_rev()) != hipSuccess) return e; }
      pend[npend++] = d; pend_variant = variant; pend_flops += 2.0 * N * 256 * (double)c.P + 2.0 * d.rowsA * 256 * (double)c.P;
      pend_bytes += 256 * (double)c.P * 2.0 + N * (double)c.P * 2.0 * 4 + (double)dwg * DW_WG_UINT4 * 16.0;
      if (l == 1 || !g_dw_segments) { if ((e = flush_rev()) != hipSuccess) return e; }
      continue;
    }
    if ((e = layer(c, L.tr[l], EPI_BWD, in(w.m[l], kO[l]), NOIN, nullptr, N, w.m[l - 1], Arr{}, 1 << 30, w.h[l], w.m[l - 1])) != hipSuccess) return e;
  }
  if (inter) return wgrad_layer(0);
  bool done[9] = {};
  if (dw) {
    // one gather launch: 7 layers (tangent + reverse partial sets each) x g_dw_nsub sub-ranges of workgroups -> g_dw_nsub fp32 splits per layer in the partial-tile format;
    // then the weight-norm finish of the seven layers in one launch
    const int K = 256, Kld2 = (K + 1 + 7) / 8 * 8, splits = g_dw_nsub;
    const size_t region = (size_t)256 * splits * Kld2;
    DwGatherArgs ga{};
    WreduceBatch wb{};
    for (int l = 1; l <= 7; ++l) {
      float* out = w.partial + (size_t)(l - 1) * region;
      const size_t j0 = (size_t)(2 * (l - 1)), j1 = j0 + 1;          // tangent / reverse partial sets of the layer
      DwGatherJob& jb = ga.job[l - 1];
      jb.partial = reinterpret_cast<const uint4*>(w.dwp) + j0 * dw_slots(c.ldp) * DW_WG_UINT4;
      jb.pscale = w.dwscale + j0 * dw_slots(c.ldp) * 8;
      jb.partial2 = reinterpret_cast<const uint4*>(w.dwp) + j1 * dw_slots(c.ldp) * DW_WG_UINT4;
      jb.pscale2 = w.dwscale + j1 * dw_slots(c.ldp) * 8;
      jb.pbias = w.dwbias + j1 * dw_slots(c.ldp) * 256;
      jb.nwg = dwg; jb.transposed = 0;
      jb.out = out; jb.row_stride = (size_t)splits * Kld2; jb.split_stride = Kld2; jb.split0 = 0;
      jb.bias_col = K; jb.rows = kO[l]; jb.cols = K;
      const PackDesc2& pd = L.d[L.fwd[l]];
      WreduceArgs& r = wb.a[l - 1];
      r.partial = out; r.splits = splits; r.row_stride = (size_t)splits * Kld2; r.split_stride = Kld2;
      r.O = kO[l]; r.I = kI[l];
      r.s0 = pd.s0; r.s0p = pd.s0p; r.off0 = pd.off0; r.off1 = pd.off1; r.rot = pd.rot; r.scale = pd.scale;
      r.v = c.net->v[l]; r.g = c.net->g[l];
      r.dv = gr->dv[l]; r.dg = gr->dg[l]; r.db = gr->db[l];
      r.bias_col = K;
      done[l] = true;
    }
    int njobs = 7;
    if (dw8) {
      // lin8: packed rows 0..255 = the feature rows (one partial set, reverse launch only), packed row 256 = the sdf row: cotangent abar8
      // row 0 against h8, plus the plain row sums of vh8 (second-order part, the adjoint seed being 1), from rowdot_kernel's block partials
      float* out = w.partial + (size_t)7 * region;
      const size_t region8 = (size_t)257 * splits * Kld2;
      float* xrow = out + region8;
      const int xchunk = ((c.ldp + DW8_XBLOCKS - 1) / DW8_XBLOCKS + 31) / 32 * 32;
      const int xn = (c.P + xchunk - 1) / xchunk;
      const size_t j = 14;
      DwGatherJob& jb = ga.job[7];
      jb.partial = reinterpret_cast<const uint4*>(w.dwp) + j * dw_slots(c.ldp) * DW_WG_UINT4;
      jb.pscale = w.dwscale + j * dw_slots(c.ldp) * 8;
      jb.partial2 = nullptr; jb.pscale2 = nullptr;
      jb.pbias = w.dwbias + j * dw_slots(c.ldp) * 256;
      jb.nwg = dwg; jb.transposed = 0;
      jb.out = out; jb.row_stride = (size_t)splits * Kld2; jb.split_stride = Kld2; jb.split0 = 0;
      jb.bias_col = K; jb.rows = 256; jb.cols = K;
      jb.xrow = xrow; jb.xrow_n = xn; jb.xrow_stride = Kld2;
      const PackDesc2& pd = L.d[L.fwd[8]];
      WreduceArgs& r = wb.a[7];
      r.partial = out; r.splits = splits; r.row_stride = (size_t)splits * Kld2; r.split_stride = Kld2;
      r.O = kO[8]; r.I = kI[8];
      r.s0 = pd.s0; r.s0p = pd.s0p; r.off0 = pd.off0; r.off1 = pd.off1; r.rot = pd.rot; r.scale = pd.scale;
      r.v = c.net->v[8]; r.g = c.net->g[8];
      r.dv = gr->dv[8]; r.dg = gr->dg[8]; r.db = gr->db[8];
      r.bias_col = K;
      done[8] = true;
      njobs = 8;
    }
    // (accounted with the weight-gradient class: no flops, the partials read once + the fp32 splits written and read once)
    ProfSlot* psg = prof_begin(c.st, 1, 0.0, (14.0 + (dw8 ? 1.0 : 0.0)) * dwg * DW_WG_UINT4 * 16.0 + 2.0 * njobs * (double)region * 4.0);
    hipLaunchKernelGGL(dw_gather_kernel, dim3(DW_WG_UINT4 / 256, g_dw_nsub, njobs), dim3(256), 0, c.st, ga);
    dbg_sync(c.st, "dw gather", 0, 0, 0);
    hipLaunchKernelGGL(wreduce_wnorm_batch_kernel, dim3(dw8 ? 257 : 256, njobs), dim3(WG), 0, c.st, wb);
    prof_end(c.st, psg);
    dbg_sync(c.st, "dw finish", 0, 0, 0);
    if ((e = hipGetLastError()) != hipSuccess) return e;
  }
  const int nb2 = wgrad_batch_size(c.ldp);
  if (oct && nb2 && g_wgrad_h3 && !dw) {
    // the six hidden layers with 256 packed input columns as six problems (two (A, B) pairs each) of one launch
    const int ls[6] = {1, 2, 3, 5, 6, 7};
    WProb pb[6];
    for (int t = 0; t < 6; ++t) {
      const int l = ls[t];
      pb[t] = WProb{l, {w.m[l], w.u[l]}, {kO[l], kO[l]}, {w.h[l], w.vh[l]}};
      done[l] = true;
    }
    for (int t = 0; t < 6; t += nb2)
      if ((e = wgrad_multi(c, w, pb + t, nb2, 2, gr)) != hipSuccess) return e;
  }
  for (int l = 0; l <= 8; ++l)
    if (!done[l] && (e = wgrad_layer(l)) != hipSuccess) return e;
  return hipSuccess;
}

// ------------------------------------------------------------------------------------------------
// heads
// ------------------------------------------------------------------------------------------------
// n_main: points that hold ray samples (the heads' outputs beyond them are never read)
// Which forward path filled a heads workspace: the fused backward chain reads the ReLU-mask arrays, which only the fused forward
// chains write (and the split-precision forward only under tuning key 14 == 2).  The switch is a mutable process global; a
// backward that runs under another setting than its forward (a key changed in between, two models sharing the process) must not
// read masks that were never written.  Keyed by the first array of the workspace; host-side, so a captured graph records the
// pairing that was valid at capture.
std::mutex g_head_path_mutex;
std::unordered_map<const void*, int> g_head_path;      // workspace -> 1: masks written by the forward of this workspace
void note_head_path(const HeadWs& h, bool masks_written) {
  std::lock_guard<std::mutex> lock(g_head_path_mutex);
  // (workspaces come and go with the caller's allocator: bound the table by dropping ONE other entry -- never the whole table, which
  // could take the entry of a forward whose backward has not run yet and send that backward down the per-layer path)
  if (g_head_path.size() > 4096 && g_head_path.find(h.small_r) == g_head_path.end()) {
    auto victim = g_head_path.begin();
    g_head_path.erase(victim);
  }
  g_head_path[h.small_r] = masks_written ? 1 : 0;
}
bool head_masks_written(const HeadWs& h) {
  std::lock_guard<std::mutex> lock(g_head_path_mutex);
  auto it = g_head_path.find(h.small_r);
  return it != g_head_path.end() && it->second == 1;
}

hipError_t heads_forward(const Ctx& c, const HeadWs& h, Arr feat, Arr featlo = Arr{}, bool save = true, int n_main = -1) {
  const PackLayout& L = c.L();
  hipError_t e;
  note_head_path(h, save && c.prec && ((c.hx3 && g_head_chain == 2) || (!c.hx3 && g_head_chain != 0)));
  // (16-bit builds: the octet-major copies of the small head inputs, smallbf_r / smallbf_a, were written by head_inputs_kernel)
  if (c.hx3) {
    // split-precision forward: one fused launch per head (kernels_x3.hpp); the hidden activations stay on chip, their hi planes
    // go to hr / ha for the 16-bit backward (save)
    if (!featlo.p) return hipErrorInvalidValue;
    for (int head = 0; head < 2; ++head) {
      const int base = head ? L_ATTR : L_REND;
      const Arr* hh = head ? h.ha : h.hr;
      HeadX3Args a{};
      a.P = c.P; a.ldp = c.ldp;
      a.nvalid = ((n_main < 0 ? c.P : n_main) + X3_BATCH - 1) / X3_BATCH;
      a.feat = reinterpret_cast<const u16*>(feat.p); a.featlo = reinterpret_cast<const u16*>(featlo.p);
      a.small = head ? h.small_a : h.small_r; a.srows = head ? SMALL_A : SMALL_R;
      for (int l = 0; l < 5; ++l) {
        a.Wp[l] = reinterpret_cast<const uint4*>(c.packed + L.d[L.fwd[base + l]].offset);
        a.Wlo[l] = reinterpret_cast<const uint4*>(c.packed + L.d[L.fwd_lo[base + l]].offset);
        a.bias[l] = c.net->b[base + l];
        a.hid[l] = (l && save) ? reinterpret_cast<u16*>(hh[l].p) : nullptr;
      }
      if (L.d[L.fwd[base]].Kpad != 320) return hipErrorInvalidValue;
      a.out = head ? h.lin : h.rgb;
      if (save && g_head_chain == 2) for (int l = 1; l <= 4; ++l) a.mask[l] = reinterpret_cast<u16*>(head ? h.ma[l] : h.mr[l]);
      double fl3 = 0.0;
      for (int l = 0; l < 5; ++l) fl3 += 2.0 * kO[base + l] * kI[base + l] * (double)c.P;
      ProfSlot* ps3 = prof_begin(c.st, 4, fl3, (double)c.P * (2 * 512.0 + a.srows * 4.0 + (save ? 4 * 512.0 : 0.0) + (head ? 24.0 : 12.0)) + 4.0 * 540000.0);
      e = launch_head_chain_x3(c.st, a, head, c.ldp / X3_BATCH, g_ws_grid, save);
      prof_end(c.st, ps3);
      if (e != hipSuccess) return e;
    }
    return hipSuccess;
  }
  if (c.prec && g_head_chain) {
    // 16-bit builds: one fused launch per head (kernels_heads.hpp)
    for (int head = 0; head < 2; ++head) {
      const int base = head ? L_ATTR : L_REND;
      const Arr* hh = head ? h.ha : h.hr;
      HeadX3Args a{};
      a.P = c.P; a.ldp = c.ldp;
      a.nvalid = ((n_main < 0 ? c.P : n_main) + 63) / 64;
      a.feat = reinterpret_cast<const u16*>(feat.p);
      a.smallbf = reinterpret_cast<const u16*>((head ? h.smallbf_a : h.smallbf_r).p); a.srows = head ? SMALL_A : SMALL_R;
      for (int l = 0; l < 5; ++l) {
        a.Wp[l] = reinterpret_cast<const uint4*>(c.packed + L.d[L.fwd[base + l]].offset);
        a.bias[l] = c.net->b[base + l];
        a.hid[l] = (l && save) ? reinterpret_cast<u16*>(hh[l].p) : nullptr;
        a.mask[l] = (l && save) ? reinterpret_cast<u16*>(head ? h.ma[l] : h.mr[l]) : nullptr;
      }
      if (L.d[L.fwd[base]].Kpad != 320 || L.d[L.fwd[base + 4]].Kpad != 256) return hipErrorInvalidValue;
      a.out = head ? h.lin : h.rgb;
      double fl = 0.0;
      for (int l = 0; l < 5; ++l) fl += 2.0 * kO[base + l] * kI[base + l] * (double)c.P;
      ProfSlot* ps = prof_begin(c.st, 4, fl, (double)c.P * (512.0 + a.srows * 2.0 + (save ? 4 * (512.0 + 64.0) : 0.0) + (head ? 24.0 : 12.0)) + 2.0 * 540000.0);
      e = launch_head_chain(c.st, a, head, c.ldp / 64, g_ws_grid, save);
      prof_end(c.st, ps);
      if (e != hipSuccess) return e;
    }
    return hipSuccess;
  }
  for (int head = 0; head < 2; ++head) {
    const int base = head ? L_ATTR : L_REND;
    const Arr* hh = head ? h.ha : h.hr;
    const float* small = head ? h.small_a : h.small_r;
    const int srows = head ? SMALL_A : SMALL_R;
    const In small_in = c.prec ? in(head ? h.smallbf_a : h.smallbf_r, srows) : in(F(small), srows);
    if ((e = layer(c, L.fwd[base], EPI_RELU, in(feat, 256), small_in, c.net->b[base], 256, hh[1])) != hipSuccess) return e;
    for (int l = 1; l < 4; ++l)
      if ((e = layer(c, L.fwd[base + l], EPI_RELU, in(hh[l], 256), NOIN, c.net->b[base + l], 256, hh[l + 1])) != hipSuccess) return e;
    if (head == 0) e = layer(c, L.fwd[base + 4], EPI_SIGMOID, in(hh[4], 256), NOIN, c.net->b[base + 4], 3, F(h.rgb));
    else e = layer(c, L.fwd[base + 4], EPI_LINEAR, in(hh[4], 256), NOIN, c.net->b[base + 4], 6, F(h.lin));
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

// zrgb / dlin hold the cotangents of the heads' last linear outputs; accumulates the feature cotangent into
// abar8 rows 1..256 (render overwrites, attraction adds) and the small-input cotangents into sc_r / sc_a.
// slot / slot_a (f16 build): the common cotangent scale and the attraction head's own (null: one scale)
hipError_t heads_backward(const Ctx& c, const HeadWs& h, const SdfWs& w, const neat_net_grads* gr, const float* slot = nullptr,
                          const float* slot_a = nullptr, int n_main = -1) {
  const PackLayout& L = c.L();
  hipError_t e;
  const bool oct = oct_operands(c);
  // (16-bit builds: the octet copies of zrgb / dlin, topbf_r / topbf_a, were written by composite_bwd_kernel, tail columns included)
  const int nb1 = wgrad_batch_size(c.ldp);
  const bool batch = oct && nb1 && g_wgrad_h3;
  bool wdone[2] = {false, false};
  // all weight gradients of one head (per_head_batch: its three hidden layers as one multi-problem launch)
  auto head_wgrads = [&](int head, bool per_head_batch) -> hipError_t {
    const int base = head ? L_ATTR : L_REND;
    const Arr* hh = head ? h.ha : h.hr;
    const Arr* ab = head ? h.aa : h.ar;
    const float* top = head ? h.dlin : h.zrgb;
    const float* small = head ? h.small_a : h.small_r;
    const int srows = head ? SMALL_A : SMALL_R;
    hipError_t e2;
    const bool hb = per_head_batch && batch;
    bool l4_batched = false;
    if (hb) {
      // the three hidden layers and (16-bit operands) the output layer -- 3 / 6 rows against the same kind of 256-row B operand -- as
      // problems of one launch: one weight-gradient launch and one finish less per head
      WProb pb[4];
      for (int l = 1; l <= 3; ++l) pb[l - 1] = WProb{base + l, {ab[l], Arr{}}, {256, 0}, {hh[l], Arr{}}};
      l4_batched = oct && g_head_l4_batched;
      if (l4_batched) pb[3] = WProb{base + 4, {head ? h.topbf_a : h.topbf_r, Arr{}}, {kO[base + 4], 0}, {hh[4], Arr{}}};
      if ((e2 = wgrad_multi(c, w, pb, l4_batched ? 4 : 3, 1, gr)) != hipSuccess) return e2;
    }
    for (int l = 0; l <= 4; ++l) {
      if ((hb || (!per_head_batch && batch)) && l >= 1 && l <= 3) continue;
      if (l == 4 && l4_batched) continue;
      WPair pr[1] = {};
      pr[0].A = l == 4 ? (oct ? (head ? h.topbf_a : h.topbf_r) : F(top)) : ab[l]; pr[0].rowsA = kO[base + l];
      if (l == 0) {
        pr[0].B[0] = w.feat; pr[0].rowsB[0] = 256;
        pr[0].B[1] = oct ? (head ? h.smallbf_a : h.smallbf_r) : F(small); pr[0].rowsB[1] = srows;
      } else {
        pr[0].B[0] = hh[l]; pr[0].rowsB[0] = 256;
      }
      if ((e2 = wgrad(c, w, base + l, pr, 1, kO[base + l], gr)) != hipSuccess) return e2;
    }
    wdone[head] = true;
    return hipSuccess;
  };
  for (int head = 0; head < 2; ++head) {
    const int base = head ? L_ATTR : L_REND;
    const Arr* hh = head ? h.ha : h.hr;
    const Arr* ab = head ? h.aa : h.ar;
    const float* top = head ? h.dlin : h.zrgb;
    const int top_rows = head ? 6 : 3;
    const float* small = head ? h.small_a : h.small_r;
    const int srows = head ? SMALL_A : SMALL_R;
    if (oct && g_head_chain == 2 && head_masks_written(h)) {
      // fused backward chain (kernels_heads.hpp): the ReLU masks come from the forward chain of this step (a forward that took
      // another path leaves none: per-layer launches below, which read the saved activations)
      HeadBwdArgs a{};
      a.P = c.P; a.ldp = c.ldp;
      a.nvalid = ((n_main < 0 ? c.P : n_main) + 63) / 64;
      a.top = reinterpret_cast<const u16*>((head ? h.topbf_a : h.topbf_r).p);
      for (int l = 0; l < 5; ++l) {
        a.Wt[l] = reinterpret_cast<const uint4*>(c.packed + L.d[L.tr[base + l]].offset);
        a.mask[l] = l ? reinterpret_cast<const u16*>(head ? h.ma[l] : h.mr[l]) : nullptr;
      }
      for (int l = 0; l < 4; ++l) a.ab[l] = reinterpret_cast<u16*>(ab[l].p);
      if (L.d[L.tr[base + 4]].Kpad != 64 || L.d[L.tr[base]].Kpad != 256) return hipErrorInvalidValue;
      a.featc = reinterpret_cast<u16*>(w.featc.p);
      a.accumulate = head;
      a.rho_num = (head && slot_a) ? slot : nullptr; a.rho_den = (head && slot_a) ? slot_a : nullptr;
      a.sc = head ? h.sc_a : h.sc_r; a.srows = srows;
      double fl = 0.0;
      for (int l = 0; l < 5; ++l) fl += 2.0 * kO[base + l] * kI[base + l] * (double)c.P;
      ProfSlot* ps = prof_begin(c.st, 4, fl, (double)c.P * (16.0 + 4 * (512.0 + 64.0) + (head ? 1024.0 : 512.0) + srows * 4.0) + 2.0 * 540000.0);
      e = launch_head_bwd_chain(c.st, a, head, c.ldp / 64, g_ws_grid);
      prof_end(c.st, ps);
      if (e != hipSuccess) return e;
      // ... while its cotangents are the last thing written (unless the batch size in force splits three problems 2 + 1: a multi-problem
      // launch takes at least two, and the batch of two is what large point counts want)
      if (g_head_wgrad_order && (!batch || nb1 >= 3) && (e = head_wgrads(head, true)) != hipSuccess) return e;
      continue;
    }
    if ((e = layer(c, L.tr[base + 4], EPI_BWD_RELU, in(oct ? (head ? h.topbf_a : h.topbf_r) : F(top), top_rows), NOIN, nullptr, 256, ab[3], Arr{}, 1 << 30, hh[4])) != hipSuccess) return e;
    for (int l = 3; l >= 1; --l)
      if ((e = layer(c, L.tr[base + l], EPI_BWD_RELU, in(ab[l], 256), NOIN, nullptr, 256, ab[l - 1], Arr{}, 1 << 30, hh[l])) != hipSuccess) return e;
    if (c.prec) {
      // feature cotangent (256 rows, octet-major bf16: render writes, attraction adds) on the streaming kernel; the few
      // small-input rows (packed rows 256..) with the narrow-output path
      if ((e = layer_ws(c, L.tr[base], head ? EPI_LINACC : EPI_LINEAR, ab[0], 256, w.featc, head ? w.featc : Arr{}, Arr{}, nullptr, nullptr, nullptr,
                        (head && slot_a) ? slot : nullptr, (head && slot_a) ? slot_a : nullptr)) != hipSuccess) return e;
      if ((e = layer(c, L.tr[base], EPI_LINEAR, in(ab[0], 256), NOIN, nullptr, srows, F(head ? h.sc_a : h.sc_r), Arr{}, 1 << 30,
                     Arr{}, Arr{}, 0, 0, 1 << 30, nullptr, 0, 8)) != hipSuccess) return e;
    } else if ((e = layer(c, L.tr[base], EPI_LINEAR, in(ab[0], 256), NOIN, nullptr, 256 + srows, F(w.abar8 + c.ldp),
                          F(head ? h.sc_a : h.sc_r), 256, Arr{}, Arr{}, head)) != hipSuccess) return e;
  }
  // weight gradients not yet launched; the hidden layers l = 1..3 of the two heads have identical shapes: one launch per layer for both
  if (!wdone[0] && !wdone[1] && batch) {          // 3 hidden layers x 2 heads: six single-pair problems, g_wgrad_batch per launch
    WProb pb[6];
    for (int l = 1; l <= 3; ++l) {
      pb[2 * (l - 1)] = WProb{L_REND + l, {h.ar[l], Arr{}}, {256, 0}, {h.hr[l], Arr{}}};
      pb[2 * (l - 1) + 1] = WProb{L_ATTR + l, {h.aa[l], Arr{}}, {256, 0}, {h.ha[l], Arr{}}};
    }
    for (int t = 0; t < 6; t += nb1)
      if ((e = wgrad_multi(c, w, pb + t, nb1, 1, gr)) != hipSuccess) return e;
  }
  for (int head = 0; head < 2; ++head)
    if (!wdone[head] && (e = head_wgrads(head, false)) != hipSuccess) return e;
  return hipSuccess;
}

__global__ void volume_weights_kernel(const float* __restrict__ z, const float* __restrict__ sdf, int R, int S,
                                      const float* __restrict__ beta_ptr, float* __restrict__ weights) {
  const float beta = *beta_ptr;
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  float carry = 0.0f;
  for (int i0 = 0; i0 < S; i0 += 64) {
    const int i = i0 + lane;
    const bool ok = i < S;
    const int p = r * S + (ok ? i : S - 1);
    const float delta = (i + 1 < S) ? z[p + 1] - z[p] : 1e10f;
    const float e = ok ? delta * laplace_sigma(sdf[p], beta) : 0.0f;
    const float incl = wave_incl_scan(e, lane);
    float excl = __shfl_up(incl, 1);
    if (lane == 0) excl = 0.0f;
    if (ok) weights[p] = (1.0f - expf(-e)) * expf(-(carry + excl));
    carry += __shfl(incl, 63);
  }
}

__global__ void lines_from_offsets_kernel(const float* __restrict__ lin_fm, const float* __restrict__ x_fm, int P, int ldp,
                                          float* __restrict__ lines) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  for (int c = 0; c < 6; ++c) lines[(size_t)p * 6 + c] = x_fm[(size_t)(c % 3) * ldp + p] + lin_fm[(size_t)c * ldp + p];
}

// abar8 (source row order) row 0 <- (1-mask) d_sdf ; rows 1.. <- d_feat ; (+ d_out257)
__global__ void build_abar8_kernel(const float* __restrict__ d_out257, const float* __restrict__ d_sdf,
                                   const float* __restrict__ d_feat, const float* __restrict__ mask, int P, int ldp,
                                   float* __restrict__ abar8, const float* __restrict__ cot_slot) {
  const float scale = cot_scale_of(cot_slot);
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = blockIdx.y;
  if (p >= ldp) return;
  float v = 0.0f;
  if (p < P) {
    if (d_out257) v += d_out257[(size_t)p * 257 + n];
    if (n == 0) { if (d_sdf) v += d_sdf[p] * (1.0f - mask[p]); }
    else if (d_feat) v += d_feat[(size_t)p * 256 + (n - 1)];
  }
  abar8[(size_t)n * ldp + p] = v * scale;
}

// f16 build: cotangent normalisation (kernels.hpp: cot_scale_of).  cot_scale_begin reduces max |cotangent| over the caller's arrays
// into `slot` (a float of the workspace: the `ones` row, which only the fp32 build uses); the kernels that read the caller's
// cotangents scale by 2^k; grad_unscale takes the factor out of the finished parameter gradients.  Other builds: slot = nullptr.
// (a kernel, not hipMemsetAsync: inside a captured graph every memset node cost 35-45 us of idle GPU before the next kernel node --
// two of them per f16 / fp16x3 step, found in the step's launch-ordered trace)
__global__ void zero_word_kernel(float* p) { if (threadIdx.x == 0) *p = 0.0f; }
const float* cot_scale_begin(const Ctx& c, float* slot, std::initializer_list<std::pair<const float*, long long>> arrays) {
  if (!NEAT_HALF) return nullptr;
  if (arrays.size() > (size_t)COT_MAX_ARRAYS) abort();      // (a programming error: enlarge COT_MAX_ARRAYS)
  static const bool use_memset = getenv("NEAT_COT_MEMSET") != nullptr;      // probe: the memset node of rounds 2-4
  if (use_memset) (void)hipMemsetAsync(slot, 0, sizeof(float), c.st);
  else hipLaunchKernelGGL(zero_word_kernel, dim3(1), dim3(64), 0, c.st, slot);
  CotMaxArgs a{};
  long long total = 0;
  for (const auto& it : arrays) { if (a.narr >= COT_MAX_ARRAYS) break; a.p[a.narr] = it.first; a.n[a.narr] = it.first ? it.second : 0; total += a.n[a.narr]; ++a.narr; }
  a.slot = slot;
  const int blocks = (int)std::min<long long>(1024, (total + 2047) / 2048);
  if (blocks > 0) hipLaunchKernelGGL(cot_max_kernel, dim3(blocks), dim3(256), 0, c.st, a);
  return slot;
}
struct GradUnscaleArgs { float* p[3 * NLAYERS]; int n[3 * NLAYERS]; const float* slot; };
__global__ void grad_unscale_kernel(GradUnscaleArgs a) {
  float* p = a.p[blockIdx.y];
  const int n = a.n[blockIdx.y];
  if (!p) return;
  const float inv = 1.0f / cot_scale_of(a.slot);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] *= inv;
}
void grad_unscale(const Ctx& c, const neat_net_grads* gr, int first, int count, const float* slot) {
  if (!slot) return;
  GradUnscaleArgs a{};
  for (int l = 0; l < NLAYERS; ++l) {
    const bool on = l >= first && l < first + count && gr->dv[l];
    a.p[3 * l] = on ? gr->dv[l] : nullptr; a.n[3 * l] = kO[l] * kI[l];
    a.p[3 * l + 1] = on ? gr->dg[l] : nullptr; a.n[3 * l + 1] = kO[l];
    a.p[3 * l + 2] = on ? gr->db[l] : nullptr; a.n[3 * l + 2] = kO[l];
  }
  a.slot = slot;
  hipLaunchKernelGGL(grad_unscale_kernel, dim3(64, 3 * NLAYERS), dim3(256), 0, c.st, a);
}

// row-major copies of the lin8 output for the stand-alone module API
void export_out8(const Ctx& c, const SdfWs& w, float* out257, float* feat) {
  const int P = c.P;
  if (c.prec) {
    if (out257 || feat)
      hipLaunchKernelGGL(export_out8_kernel, dim3((P + 255) / 256, 32), dim3(256), 0, c.st, (const float*)w.sdfraw, reinterpret_cast<const u16*>(w.feat.p),
                         reinterpret_cast<const u16*>(w.featlo.p), P, c.ldp, out257, feat);
  } else {
    if (out257) hipLaunchKernelGGL(fm_to_rm_kernel, grid1(P), dim3(256), 0, c.st, w.sdfraw, P, 257, c.ldp, out257, 0);
    if (feat) hipLaunchKernelGGL(fm_to_rm_kernel, grid1(P), dim3(256), 0, c.st, w.feat.f(), P, 256, c.ldp, feat, 0);
  }
}

bool bad_prec(int p) { return p != F32 && p != BF16 && p != BF16X3 && !(NEAT_HALF && (p == HX3 || p == HX3_FASTVALUES)); }

}  // namespace

// ================================================================================================
// C ABI
// ================================================================================================
#if !NEAT_HALF
#define NEAT_TWIN(fn) extern "C" decltype(fn) f16_##fn;
NEAT_TWIN(neat_set_tuning) NEAT_TWIN(neat_prof_enable) NEAT_TWIN(neat_prof_collect)
NEAT_TWIN(neat_packed_floats) NEAT_TWIN(neat_pack_weights) NEAT_TWIN(neat_sdf_ws_floats) NEAT_TWIN(neat_sdf_forward)
NEAT_TWIN(neat_sdf_backward) NEAT_TWIN(neat_heads_ws_floats) NEAT_TWIN(neat_heads_forward) NEAT_TWIN(neat_render_ws_floats)
NEAT_TWIN(neat_render_forward) NEAT_TWIN(neat_render_backward) NEAT_TWIN(neat_render_eval_ws_floats)
NEAT_TWIN(neat_render_forward_eval) NEAT_TWIN(neat_sdf_values_gated) NEAT_TWIN(neat_sdf_values_rays) NEAT_TWIN(neat_sdf_values_laid_out)
#define NEAT_F16_FWD(call) if (precision == 3) { precision = BF16; return f16_##call; } if (precision == HX3 || precision == HX3_FASTVALUES) return f16_##call;
#else
#define NEAT_F16_FWD(call)
#endif

extern "C" {

int neat_abi_version(void) { return 13; }

int neat_set_tuning(int key, int value) {          /* 0: bf16 layer-kernel point tile (2 -> 64 points, 4 -> 128 points) */
#if !NEAT_HALF
  f16_neat_set_tuning(key, value);                 /* the f16 twin keeps its own copies of the switches */
#endif
  if (key == 0 && (value == 2 || value == 4)) { g_pt_bf16 = value; return 0; }
  if (key == 1 && (value == 0 || value == 1)) { g_wgrad_h3 = value; return 0; }
  if (key == 2 && (value == 0 || value == 1)) { g_layer_ws = value; return 0; }
  if (key == 3 && value >= 1 && value <= 4096) { g_ws_grid = value; return 0; }
  if (key == 4 && value >= 0 && value <= 4) { g_fused_ws = value; return 0; }
  if (key == 6 && value >= 0 && value <= 2) { g_wreduce_direct = value; return 0; }
  if (key == 7 && (value == 0 || value == 1)) { g_wgrad_interleave = value; return 0; }
  if (key == 8 && (value == -1 || value == 0 || value == 2 || value == 3 || value == 6)) { g_wgrad_batch = value; return 0; }
  if (key == 5 && (value == 0 || (value >= 2 && value <= 4))) { g_fused_nt = value; return 0; }
  if (key == 9 && value >= 0 && value <= 2) { g_ws_interleave = value; return 0; }
  if (key == 10 && (value == 0 || value == 1)) { g_fused_interleave = value; return 0; }
  if (key == 11 && value >= 0 && value <= 63) { g_ws_aux_nt = value; return 0; }
  if (key == 12 && (value == 0 || value == 1)) { g_ws_wide_store = value; return 0; }
  if (key == 13 && (value == 0 || value == 1)) { g_fused_adj = value; return 0; }
  if (key == 14 && value >= 0 && value <= 2) { g_head_chain = value; return 0; }
  if (key == 15 && (value == 0 || value == 1)) { g_head_wgrad_order = value; return 0; }
  if (key == 16 && value >= 0 && value <= 2) { g_dw_fused = value; return 0; }
  if (key == 17 && value >= 0 && value <= 7) { g_dw_ablate = value; return 0; }
  if (key == 18 && value >= 1 && value <= 16) { g_dw_nsub = value; return 0; }
  if (key == 19 && (value == 0 || value == 1)) { g_wgrad_k320 = value; return 0; }
  if (key == 20 && (value == 0 || value == 1)) { g_head_l4_batched = value; return 0; }
  if (key == 21 && value >= 0 && value <= 4) { g_wgrad_narrow = value; return 0; }
  if (key == 22 && (value == 0 || value == 1)) { g_dw_lin8 = value; return 0; }
  if (key == 23 && (value == 0 || (value >= 16 && value <= DW_MAXGRID))) { g_dw_grid = value; return 0; }
  if (key == 24 && (value == 0 || value == 1)) { g_chain_pp = value; return 0; }
  if (key == 25 && (value == 0 || value == 1)) { g_dw_segments = value; return 0; }
  if (key == 28 && (value == 0 || value == 1)) { g_ffn_mfma = value; return 0; }
  if (key == 30 && value >= 0 && value <= 7) { g_sampler_ablate = value; return 0; }      /* probes: sampler_round_kernel without its bisection (1) / refine (2) / final (4) part */
  return -1;
}

int neat_prof_enable(int on) {
#if !NEAT_HALF
  f16_neat_prof_enable(on);
#endif
  g_prof.on = on != 0;
  g_prof.used = 0;
  return 0;
}

int neat_prof_collect(int cls, double* total_ms, double* total_flops, int* launches, double* total_bytes) {
  double ms = 0.0, fl = 0.0, by = 0.0;
  int n = 0;
  for (size_t i = 0; i < g_prof.used; ++i) {
    ProfSlot& s = g_prof.pool[i];
    if (s.cls != cls) continue;
    hipError_t e = hipEventSynchronize(s.e1);
    if (e != hipSuccess) return (int)e;
    float t = 0.0f;
    if ((e = hipEventElapsedTime(&t, s.e0, s.e1)) != hipSuccess) return (int)e;
    ms += t; fl += s.flops; by += s.bytes; ++n;
  }
#if !NEAT_HALF
  {                                                  /* launches made through the f16 twin are counted there */
    double ms2 = 0.0, fl2 = 0.0, by2 = 0.0;
    int n2 = 0;
    const int rc = f16_neat_prof_collect(cls, &ms2, &fl2, &n2, &by2);
    if (rc != 0) return rc;
    ms += ms2; fl += fl2; by += by2; n += n2;
  }
#endif
  if (total_ms) *total_ms = ms;
  if (total_flops) *total_flops = fl;
  if (launches) *launches = n;
  if (total_bytes) *total_bytes = by;
  return 0;
}

size_t neat_packed_floats(int precision) {
  NEAT_F16_FWD(neat_packed_floats(precision))
  const int x3 = take_x3(precision); (void)x3;
  const int hx3 = take_hx3(precision);
  return (bad_prec(precision) || hx3 == 2) ? 0 : pack_layout(precision, hx3 != 0).total; }

int neat_pack_weights(const neat_net_params* net, float* packed, int precision, void* stream) {
  NEAT_F16_FWD(neat_pack_weights(net, packed, precision, stream))
  const int x3 = take_x3(precision); (void)x3;
  const int hx3 = take_hx3(precision); (void)hx3;
  if (hx3 == 2) return -1;
  if (!net || !packed || bad_prec(precision)) return -1;
  hipStream_t st = (hipStream_t)stream;
  const PackLayout& L = pack_layout(precision, hx3 != 0);
  RowScaleArgs ra;
  ra.net = to_ptrs(net);
  ra.rowscale = packed + L.rowscale_off;
  for (int l = 0; l <= NLAYERS; ++l) ra.row_off[l] = L.row_off[l];
  hipLaunchKernelGGL(rowscale_kernel, dim3((L.row_off[NLAYERS] + 3) / 4), dim3(WG), 0, st, ra);
  PackArgs2 pa;
  pa.net = ra.net; pa.rowscale = ra.rowscale;
  for (int l = 0; l <= NLAYERS; ++l) pa.row_off[l] = L.row_off[l];
  for (int i = 0; i < L.npacks; ++i) pa.d[i] = L.d[i];
  pa.npacks = L.npacks; pa.out = packed;
  hipLaunchKernelGGL(pack_kernel2, dim3(L.nblocks, 4), dim3(WG), 0, st, pa);      // 4 slices per 32-row tile: the pack is latency-bound
  if (L.npacks_lo > 0) {          // HX3: the lo planes, a second launch (the descriptor table is a kernel argument of bounded size)
    if (L.npacks_lo > MAXPACKS2) return -1;
    for (int i = 0; i < L.npacks_lo; ++i) pa.d[i] = L.d[L.npacks + i];
    pa.npacks = L.npacks_lo;
    hipLaunchKernelGGL(pack_kernel2, dim3(L.nblocks_lo, 4), dim3(WG), 0, st, pa);
  }
  return (int)hipGetLastError();
}

int neat_camera_rays(const float* uv, const float* pose, const float* K, int kstride, int R, float* dirs, float* origins, void* stream) {
  if (R <= 0) return 0;
  hipLaunchKernelGGL(camera_rays_kernel, grid1(R), dim3(256), 0, (hipStream_t)stream, uv, pose, K, kstride, R, dirs, origins);
  return (int)hipGetLastError();
}

int neat_eik_points(const float* uniform, const float* origins, const float* dirs, const float* z_eik, const float* extra, int R, int J,
                    float* out, const float* z, int S, const long long* idx, void* stream) {
  if (R <= 0 || J < 0) return R == 0 && J == 0 ? 0 : -1;
  if (!uniform || !origins || !dirs || !out || (J > 0 && !extra)) return -1;
  if (!z_eik && (!z || !idx || S <= 0)) return -1;           // the depth per ray: given, or picked from the ray's S depths by idx
  hipLaunchKernelGGL(eik_points_kernel, grid1((2 * R + J) * 3), dim3(256), 0, (hipStream_t)stream, uniform, origins, dirs, z_eik, extra, R, J, out,
                     z, S, idx);
  return (int)hipGetLastError();
}

size_t neat_sdf_ws_floats(int P, int mode, int precision) {
  NEAT_F16_FWD(neat_sdf_ws_floats(P, mode, precision))
  const int x3 = take_x3(precision); (void)x3;
  const int hx3 = take_hx3(precision); (void)hx3;
  return bad_prec(precision) ? 0 : sdf_ws(nullptr, round_ldp(P, precision), mode, precision, hx3 == 1).total;
}

// x [P,3] row-major, or (x == null) the points o + z d of R rays x S depths (P = R S)
struct RayPoints { const float* origins; const float* dirs; const float* z; int R, S; };
static int sdf_forward_impl(const float* packed, const neat_net_params* net, const float* x, const RayPoints* rp, int P, int mode, int precision,
                            float radius, float scale, float* ws, float* out257, float* sdf, float* feat, float* grad, void* stream,
                            bool laid_out = false) {
  const int x3 = take_x3(precision); (void)x3;
  const int hx3 = take_hx3(precision); (void)hx3;
  if (hx3 == 2 && mode != 0) return -1;      // NEAT_F16X3 fast values: values-mode calls only
  if (P <= 0) return 0;
  if (!packed || !net || (!x && !rp && !laid_out) || !ws || bad_prec(precision)) return -1;
  Ctx c{(hipStream_t)stream, packed, net, P, round_ldp(P, precision), precision};
  c.x3 = x3; c.hx3 = hx3;
  SdfWs w = sdf_ws(ws, c.ldp, mode, precision, hx3);
  if (laid_out) {}      // the caller (sampler_init_kernel / sampler_round_kernel) wrote w.x = the first 3 rows of the workspace
  else if (x) hipLaunchKernelGGL(rm_to_fm_kernel, grid1(c.ldp), dim3(256), 0, c.st, x, P, 3, c.ldp, w.x);
  else hipLaunchKernelGGL(points_from_rays_kernel, grid1(c.ldp), dim3(256), 0, c.st, rp->origins, rp->dirs, rp->z, rp->R, rp->S, c.ldp, w.x,
                          (float*)nullptr, (const float*)nullptr, 0);
  if (mode == 0 && precision) {
    NEAT_CHECK(sdf_primal(c, w, false, radius, scale, sdf));      // fused: PE -> 9 layers -> clamp, one launch
    return (int)hipGetLastError();
  }
  if (mode == 0) {
    NEAT_CHECK(sdf_primal(c, w, false));
    FINALIZE_LAUNCH(c, w.x, w.sdfraw, (const float*)nullptr,
                       (const float*)nullptr, P, c.ldp, radius, scale, w.sdf, (float*)nullptr, (float*)nullptr, sdf, (float*)nullptr,
                       P, (float*)nullptr);
    return (int)hipGetLastError();
  }
  NEAT_CHECK(sdf_primal(c, w, true));
  NEAT_CHECK(sdf_adjoint(c, w));
  FINALIZE_LAUNCH(c, w.x, w.sdfraw, w.e0, w.es, P, c.ldp, radius, scale,
                     w.sdf, w.g, w.mask, sdf, grad, P, (float*)nullptr);
  export_out8(c, w, out257, feat);
  return (int)hipGetLastError();
}

int neat_sdf_forward(const float* packed, const neat_net_params* net, const float* x, int P, int mode, int precision,
                     float radius, float scale, float* ws, float* out257, float* sdf, float* feat, float* grad,
                     void* stream) {
  NEAT_F16_FWD(neat_sdf_forward(packed, net, x, P, mode, precision, radius, scale, ws, out257, sdf, feat, grad, stream))
  if (P > 0 && !x) return -1;
  return sdf_forward_impl(packed, net, x, nullptr, P, mode, precision, radius, scale, ws, out257, sdf, feat, grad, stream);
}

int neat_sdf_values_rays(const float* packed, const neat_net_params* net, const float* origins, const float* dirs, const float* z, int R,
                         int S, int precision, float radius, float scale, float* ws, float* sdf, const int* gate, int gate_value,
                         void* stream) {
  NEAT_F16_FWD(neat_sdf_values_rays(packed, net, origins, dirs, z, R, S, precision, radius, scale, ws, sdf, gate, gate_value, stream))
  if (R <= 0 || S <= 0) return 0;
  if (!packed || !net || !origins || !dirs || !z || !ws || !sdf || bad_prec(precision)) return -1;
  const RayPoints rp{origins, dirs, z, R, S};
  g_gate = gate; g_gate_value = gate_value;
  const int rc = sdf_forward_impl(packed, net, nullptr, &rp, R * S, 0, precision, radius, scale, ws, nullptr, sdf, nullptr, nullptr, stream);
  g_gate = nullptr;
  return rc;
}

int neat_sdf_ldp(int P, int precision) { return P <= 0 ? 0 : round_ldp(P, precision); }

int neat_sdf_values_laid_out(const float* packed, const neat_net_params* net, int P, int precision, float radius, float scale, float* ws,
                             float* sdf, const int* gate, int gate_value, void* stream) {
  NEAT_F16_FWD(neat_sdf_values_laid_out(packed, net, P, precision, radius, scale, ws, sdf, gate, gate_value, stream))
  if (P <= 0) return 0;
  if (!packed || !net || !ws || !sdf || bad_prec(precision)) return -1;
  g_gate = gate; g_gate_value = gate_value;
  const int rc = sdf_forward_impl(packed, net, nullptr, nullptr, P, 0, precision, radius, scale, ws, nullptr, sdf, nullptr, nullptr, stream, true);
  g_gate = nullptr;
  return rc;
}

int neat_sdf_values_gated(const float* packed, const neat_net_params* net, const float* x, int P, int precision, float radius,
                          float scale, float* ws, float* sdf, const int* gate, int gate_value, void* stream) {
  NEAT_F16_FWD(neat_sdf_values_gated(packed, net, x, P, precision, radius, scale, ws, sdf, gate, gate_value, stream))
  if (P <= 0) return 0;
  if (!packed || !net || !x || !ws || !sdf || bad_prec(precision)) return -1;
  g_gate = gate; g_gate_value = gate_value;
  const int rc = neat_sdf_forward(packed, net, x, P, 0, precision, radius, scale, ws, nullptr, sdf, nullptr, nullptr, stream);
  g_gate = nullptr;
  return rc;
}

int neat_sdf_backward(const float* packed, const neat_net_params* net, float* ws, int P, int precision,
                      const float* d_out257, const float* d_sdf, const float* d_feat, const float* d_grad,
                      const neat_net_grads* grads, void* stream) {
  NEAT_F16_FWD(neat_sdf_backward(packed, net, ws, P, precision, d_out257, d_sdf, d_feat, d_grad, grads, stream))
  const int x3 = take_x3(precision); (void)x3;
  const int hx3 = take_hx3(precision); (void)hx3;
  if (hx3 == 2) return -1;
  if (P <= 0) return 0;
  if (!packed || !net || !ws || !grads || bad_prec(precision)) return -1;
  Ctx c{(hipStream_t)stream, packed, net, P, round_ldp(P, precision), precision};
  c.x3 = x3; c.hx3 = hx3;
  SdfWs w = sdf_ws(ws, c.ldp, 1, precision, hx3);
  const float* slot = cot_scale_begin(c, w.ones, {{d_out257, 257LL * P}, {d_sdf, (long long)P}, {d_feat, 256LL * P}, {d_grad, 3LL * P}});
  hipLaunchKernelGGL(build_abar8_kernel, dim3((c.ldp + 255) / 256, 257), dim3(256), 0, c.st, d_out257, d_sdf, d_feat, w.mask, P, c.ldp, w.abar8, slot);
  if (precision) oct_pack(c, {{w.abar8 + c.ldp, 256, w.featc}});
  const NormalCot nc{nullptr, nullptr, d_grad, P, nullptr, slot, nullptr};
  NEAT_CHECK(sdf_backward_chains(c, w, grads, &nc));
  grad_unscale(c, grads, 0, 9, slot);
  return (int)hipGetLastError();
}

size_t neat_heads_ws_floats(int P, int precision) {
  NEAT_F16_FWD(neat_heads_ws_floats(P, precision))
  const int x3 = take_x3(precision); (void)x3;
  const int hx3 = take_hx3(precision); (void)hx3;
  if (hx3 == 2) return 0;
  if (bad_prec(precision)) return 0;
  const int ldp = round_ldp(P, precision);
  return head_ws(nullptr, ldp, precision).total + (size_t)(3 + 3 + 256) * ldp;
}

int neat_heads_forward(const float* packed, const neat_net_params* net, const float* points, const float* normals,
                       const float* view_dirs, const float* feats, int P, int precision, float* ws, float* rgb, float* lines,
                       void* stream) {
  NEAT_F16_FWD(neat_heads_forward(packed, net, points, normals, view_dirs, feats, P, precision, ws, rgb, lines, stream))
  const int x3 = take_x3(precision); (void)x3;
  const int hx3 = take_hx3(precision); (void)hx3;
  if (hx3 == 2) return -1;
  if (P <= 0) return 0;
  if (!packed || !net || !ws || bad_prec(precision)) return -1;
  Ctx c{(hipStream_t)stream, packed, net, P, round_ldp(P, precision), precision};
  c.x3 = x3; c.hx3 = hx3;
  HeadWs h = head_ws(ws, c.ldp, precision);
  float* x_fm = ws + h.total; float* g_fm = x_fm + 3 * (size_t)c.ldp; float* f_fm = g_fm + 3 * (size_t)c.ldp;
  hipLaunchKernelGGL(rm_to_fm_kernel, grid1(c.ldp), dim3(256), 0, c.st, points, P, 3, c.ldp, x_fm);
  hipLaunchKernelGGL(rm_to_fm_kernel, grid1(c.ldp), dim3(256), 0, c.st, normals, P, 3, c.ldp, g_fm);
  Arr feat;
  feat.p = f_fm; feat.bf16 = precision;
  Arr featlo;      // HX3: the lo plane of the feature rows in the second half of the same 256 float rows
  featlo.p = hx3 ? reinterpret_cast<u16*>(f_fm) + (size_t)256 * c.ldp : nullptr; featlo.bf16 = 1;
  if (precision) hipLaunchKernelGGL(rm_to_oct_kernel, grid1(c.ldp), dim3(256), 0, c.st, feats, P, 256, c.ldp, reinterpret_cast<u16*>(f_fm),
                                    reinterpret_cast<u16*>(featlo.p));
  else hipLaunchKernelGGL(rm_to_fm_kernel, grid1(c.ldp), dim3(256), 0, c.st, feats, P, 256, c.ldp, f_fm);
  hipLaunchKernelGGL(head_inputs_kernel, grid1(c.ldp), dim3(256), 0, c.st, x_fm, g_fm, view_dirs, P, 1, c.ldp, h.small_r, h.small_a,
                     reinterpret_cast<u16*>(h.smallbf_r.p), reinterpret_cast<u16*>(h.smallbf_a.p));
  NEAT_CHECK(heads_forward(c, h, feat, featlo));
  if (rgb) hipLaunchKernelGGL(fm_to_rm_kernel, grid1(P), dim3(256), 0, c.st, h.rgb, P, 3, c.ldp, rgb, 0);
  if (lines) hipLaunchKernelGGL(lines_from_offsets_kernel, grid1(P), dim3(256), 0, c.st, h.lin, x_fm, P, c.ldp, lines);   // y = p + offsets (rend_a :195)
  return (int)hipGetLastError();
}

size_t neat_render_ws_floats(int R, int S, int E, int precision) {
  NEAT_F16_FWD(neat_render_ws_floats(R, S, E, precision))
  const int x3 = take_x3(precision); (void)x3;
  const int hx3 = take_hx3(precision); (void)hx3;
  if (hx3 == 2) return 0;
  if (bad_prec(precision)) return 0;
  const int ldp = round_ldp(R * S + E, precision);
  return sdf_ws(nullptr, ldp, 1, precision, hx3).total + head_ws(nullptr, ldp, precision).total;
}

static int render_forward_impl(const float* packed, const neat_net_params* net, const float* origins, const float* dirs,
                               const float* z, int R, int S, int precision, const float* beta, float beta_min, float radius, float scale, float* ws,
                               float* points, float* weights, float* sdf, float* rgb, float* lines3d, float* depth,
                               float* xyz, float* normal_map, const float* eik_points, int E, float* eik_grad, void* stream, bool fwd_only) {
  const int x3 = take_x3(precision); (void)x3;
  const int hx3 = take_hx3(precision); (void)hx3;
  if (hx3 == 2) return -1;
  if (R <= 0 || S <= 0) return 0;
  if (!packed || !net || !ws || !origins || !dirs || !z || !rgb || !lines3d || !depth || !xyz || bad_prec(precision)) return -1;
  if (E < 0 || (E > 0 && (!eik_points || !eik_grad))) return -1;
  const int Pm = R * S, P = Pm + E;
  Ctx c{(hipStream_t)stream, packed, net, P, round_ldp(P, precision), precision};
  c.x3 = x3; c.hx3 = hx3;
  SdfWs w = sdf_ws(ws, c.ldp, fwd_only ? 2 : 1, precision, hx3);
  HeadWs h = head_ws(ws + w.total, c.ldp, precision, fwd_only);
  hipLaunchKernelGGL(points_from_rays_kernel, grid1(c.ldp), dim3(256), 0, c.st, origins, dirs, z, R, S, c.ldp, w.x, points, eik_points, E);
  NEAT_CHECK(sdf_primal(c, w, true));
  NEAT_CHECK(sdf_adjoint(c, w, !fwd_only));
  // normals + sphere clamp, and in the same launch the heads' small inputs (the heads run over every column of the tile grid; only
  // the first R*S columns are consumed)
  HeadInArgs hin{dirs, Pm, S, h.small_r, h.small_a, reinterpret_cast<u16*>(h.smallbf_r.p), reinterpret_cast<u16*>(h.smallbf_a.p)};
  hin.skip_fp32 = (oct_operands(c) && !c.hx3) ? 1 : 0;      // (the split-precision head chains read the fp32 rows; nothing else does in a 16-bit build)
  FINALIZE_LAUNCH_H(c, hin, w.x, w.sdfraw, w.e0, w.es, P, c.ldp, radius, scale,
                    w.sdf, w.g, w.mask, sdf, (float*)nullptr, Pm, eik_grad);
  NEAT_CHECK(heads_forward(c, h, w.feat, w.featlo, !fwd_only, Pm));
  CompositeArgs ca;
  ca.z = z; ca.sdf = w.sdf; ca.dirs = dirs; ca.x_fm = w.x; ca.rgb_fm = h.rgb; ca.lin_fm = h.lin; ca.g_fm = w.g;
  ca.R = R; ca.S = S; ca.ldp = c.ldp; ca.beta_ptr = beta; ca.beta_min = beta_min;
  ca.weights = weights; ca.rgb = rgb; ca.lines3d = lines3d; ca.depth = depth; ca.xyz = xyz; ca.normal_map = normal_map;
  hipLaunchKernelGGL(composite_fwd_kernel, dim3((R + 3) / 4), dim3(WG), 0, c.st, ca);
  return (int)hipGetLastError();
}

int neat_render_forward(const float* packed, const neat_net_params* net, const float* origins, const float* dirs,
                        const float* z, int R, int S, int precision, const float* beta, float beta_min, float radius, float scale, float* ws,
                        float* points, float* weights, float* sdf, float* rgb, float* lines3d, float* depth,
                        float* xyz, float* normal_map, const float* eik_points, int E, float* eik_grad, void* stream) {
  NEAT_F16_FWD(neat_render_forward(packed, net, origins, dirs, z, R, S, precision, beta, beta_min, radius, scale, ws, points, weights, sdf, rgb, lines3d, depth, xyz, normal_map, eik_points, E, eik_grad, stream))
  return render_forward_impl(packed, net, origins, dirs, z, R, S, precision, beta, beta_min, radius, scale, ws, points, weights, sdf, rgb, lines3d,
                             depth, xyz, normal_map, eik_points, E, eik_grad, stream, false);
}

size_t neat_render_eval_ws_floats(int R, int S, int precision) {
  NEAT_F16_FWD(neat_render_eval_ws_floats(R, S, precision))
  const int x3 = take_x3(precision); (void)x3;
  const int hx3 = take_hx3(precision); (void)hx3;
  if (hx3 == 2) return 0;
  if (bad_prec(precision)) return 0;
  const int ldp = round_ldp(R * S, precision);
  return sdf_ws(nullptr, ldp, 2, precision, hx3).total + head_ws(nullptr, ldp, precision, true).total;
}

int neat_render_forward_eval(const float* packed, const neat_net_params* net, const float* origins, const float* dirs,
                             const float* z, int R, int S, int precision, const float* beta, float beta_min, float radius, float scale, float* ws,
                             float* points, float* weights, float* sdf, float* rgb, float* lines3d, float* depth,
                             float* xyz, float* normal_map, void* stream) {
  NEAT_F16_FWD(neat_render_forward_eval(packed, net, origins, dirs, z, R, S, precision, beta, beta_min, radius, scale, ws, points, weights, sdf, rgb, lines3d, depth, xyz, normal_map, stream))
  return render_forward_impl(packed, net, origins, dirs, z, R, S, precision, beta, beta_min, radius, scale, ws, points, weights, sdf, rgb, lines3d,
                             depth, xyz, normal_map, nullptr, 0, nullptr, stream, true);
}

int neat_render_backward(const float* packed, const neat_net_params* net, float* ws, const float* dirs, const float* z,
                         int R, int S, int E, int precision, const float* beta, float beta_min, const float* d_rgb, const float* d_lines3d,
                         const float* d_depth, const float* d_xyz, const float* d_eik_grad, const float* d_acc,
                         const neat_net_grads* grads, float* dbeta_ray, float* dbeta, void* stream) {
  NEAT_F16_FWD(neat_render_backward(packed, net, ws, dirs, z, R, S, E, precision, beta, beta_min, d_rgb, d_lines3d, d_depth, d_xyz, d_eik_grad, d_acc, grads, dbeta_ray, dbeta, stream))
  const int x3 = take_x3(precision); (void)x3;
  const int hx3 = take_hx3(precision); (void)hx3;
  if (hx3 == 2) return -1;
  if (R <= 0 || S <= 0) return 0;
  if (!packed || !net || !ws || !grads || bad_prec(precision) || E < 0 || (dbeta && (!dbeta_ray || !beta))) return -1;
  const int Pm = R * S, P = Pm + E;
  Ctx c{(hipStream_t)stream, packed, net, P, round_ldp(P, precision), precision};
  c.x3 = x3; c.hx3 = hx3;
  SdfWs w = sdf_ws(ws, c.ldp, 1, precision, hx3);
  HeadWs h = head_ws(ws + w.total, c.ldp, precision);
  CompositeBwdArgs cb;
  cb.z = z; cb.sdf = w.sdf; cb.dirs = dirs; cb.mask = w.mask; cb.x_fm = w.x; cb.rgb_fm = h.rgb;
  cb.R = R; cb.S = S; cb.ldp = c.ldp; cb.beta_ptr = beta; cb.beta_min = beta_min;
  cb.d_rgb = d_rgb; cb.d_lines3d = d_lines3d; cb.d_depth = d_depth; cb.d_xyz = d_xyz; cb.d_acc = d_acc;
  cb.zrgb_fm = h.zrgb; cb.dlin_fm = h.dlin; cb.dsdf_row = w.abar8; cb.dbeta_ray = dbeta_ray;
  cb.zrgb_oct = reinterpret_cast<u16*>(h.topbf_r.p); cb.dlin_oct = reinterpret_cast<u16*>(h.topbf_a.p);      // (null in the fp32 build)
  const float* slot = cot_scale_begin(c, w.ones, {{d_rgb, 3LL * R}, {d_lines3d, 6LL * R}, {d_depth, (long long)R}, {d_xyz, 3LL * R},
                                                  {d_eik_grad, 3LL * E}, {d_acc, (long long)R}});
  // the attraction head's chain in its own scale (only its top cotangent d_lines3d feeds it)
  static const bool one_scale = getenv("NEAT_ONE_COT_SCALE") != nullptr;      // probe: the attraction head in the common scale
  const float* slot_a = (slot && d_lines3d && !one_scale) ? cot_scale_begin(c, w.ones + 1, {{d_lines3d, 6LL * R}}) : nullptr;
  cb.cot_slot = slot; cb.cot_slot_a = slot_a;
  if (!c.prec)      // the ones row is the bias column of the fp32 weight-gradient kernel; the bf16 kernels sum the rows of A themselves
    hipLaunchKernelGGL(ones_kernel, grid1(c.ldp), dim3(256), 0, c.st, w.ones, P, c.ldp);
  // columns beyond the ray samples (eikonal points, padding) carry zero head cotangents: zeroed by extra workgroups of the same launch
  cb.tail_from = c.ldp > Pm ? Pm : 0;
  hipLaunchKernelGGL(composite_bwd_kernel, dim3((R + 3) / 4 + (c.ldp > Pm ? (c.ldp - Pm + WG - 1) / WG : 0)), dim3(WG), 0, c.st, cb);
  // d loss / d beta: the per-ray partials are summed by one more workgroup of bwd_prologue_kernel (16-bit builds; fp32: a launch of its own)
  if (dbeta && !c.prec) hipLaunchKernelGGL(beta_grad_kernel, dim3(1), dim3(256), 0, c.st, dbeta_ray, R, beta, dbeta);
  NEAT_CHECK(heads_backward(c, h, w, grads, slot, slot_a, Pm));
  NormalCot nc{h.sc_r, h.sc_a, nullptr, Pm, d_eik_grad, slot, slot_a};
  if (dbeta && c.prec) { nc.dbeta_ray = dbeta_ray; nc.R = R; nc.beta_ptr = beta; nc.dbeta = dbeta; }
  NEAT_CHECK(sdf_backward_chains(c, w, grads, &nc));
  if (slot_a) { grad_unscale(c, grads, 0, L_ATTR, slot); grad_unscale(c, grads, L_ATTR, NLAYERS - L_ATTR, slot_a); }
  else grad_unscale(c, grads, 0, NLAYERS, slot);
  return (int)hipGetLastError();
}

int neat_sampler_bound(const float* z, int n, int R, const float* sdf_old, const float* sdf_new, const int* order, int n_old,
                       const float* beta_in, const float* beta0, float eps, int iters, float* sdf_out, float* beta_out,
                       int* flag, void* stream) {
  if (R <= 0) return 0;
  if (n < 2 || n > SMAX || !z || !sdf_new || !beta_in || !beta0 || !sdf_out || !beta_out || !flag) return -1;
  SamplerBoundArgs a{z, n, R, sdf_old, sdf_new, order, n_old, beta_in, beta0, eps, iters, sdf_out, beta_out, flag, nullptr, 0};
  hipLaunchKernelGGL(sampler_bound_kernel, dim3(R), dim3(BOUND_T), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

int neat_sampler_resample(const float* z, const float* sdf, int n, int R, const float* beta, int refine, float add_tiny,
                          const float* u, int u_stride, int N, float* samples, float* z_merged, int* order, void* stream) {
  if (R <= 0) return 0;
  if (n < 2 || n > SMAX || N < 1 || N > SMAX || !z || !sdf || !beta || !u || !samples || (refine && (!z_merged || !order))) return -1;
  SamplerResampleArgs a{};
  a.z = z; a.sdf = sdf; a.n = n; a.R = R; a.beta = beta; a.refine = refine; a.add_tiny = add_tiny; a.u = u; a.u_stride = u_stride; a.N = N;
  a.samples = samples; a.z_merged = z_merged; a.order = order;
  hipLaunchKernelGGL(sampler_resample_kernel, dim3(R), dim3(BOUND_T), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

int neat_sampler_finish(const float* samples, int N, const float* z, int n, const int* pick, int n_extra, float near, float far,
                        int R, const int* eik_idx, float* z_vals, float* z_eik, void* stream) {
  if (R <= 0) return 0;
  if (N + 2 + n_extra > SMAX || !samples || !z || (n_extra > 0 && !pick) || !eik_idx || !z_vals || !z_eik) return -1;
  SamplerFinishArgs a{samples, N, z, n, pick, n_extra, near, far, R, eik_idx, z_vals, z_eik, n};
  hipLaunchKernelGGL(sampler_finish_kernel, dim3(R), dim3(64), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

int neat_sample_pdf(const float* bins, const float* weights, int nb, int R, const float* u, int u_stride, int N, float* samples,
                    const float* z_merge, int nz, float* z_out, void* stream) {
  if (R <= 0) return 0;
  if (nb < 2 || nb > SMAX || N < 1 || N > SMAX || !bins || !weights || !u || !samples || (z_merge && (!z_out || nz < 0 || nz + N > SMAX))) return -1;
  SamplePdfArgs a{bins, weights, nb, R, u, u_stride, N, samples, z_merge, nz, z_out};
  hipLaunchKernelGGL(sample_pdf_kernel, dim3(R), dim3(64), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

int neat_uniform_depths(const float* near_r, float near_s, const float* far_r, float far_s, const float* t, const float* rnd, int R, int N,
                        float* z, void* stream) {
  if (R <= 0 || N <= 0) return 0;
  if (!t || !z) return -1;
  hipLaunchKernelGGL(uniform_depths_kernel, dim3((R * N + 255) / 256), dim3(256), 0, (hipStream_t)stream, near_r, near_s, far_r, far_s, t, rnd, R, N, z);
  return (int)hipGetLastError();
}

int neat_sampler_init(const float* z, int R, int n, const float* beta, float beta_min, float beta_c, float* beta0, float* beta_ray,
                      int* ctl, int nctl, void* stream) {
  return neat_sampler_init_rays(z, R, n, beta, beta_min, beta_c, beta0, beta_ray, ctl, nctl, nullptr, nullptr, nullptr, 0, nullptr, 0, 0, 0,
                                nullptr, stream);
}

int neat_sampler_init_rays(const float* z, int R, int n, const float* beta, float beta_min, float beta_c, float* beta0, float* beta_ray,
                           int* ctl, int nctl, const float* origins, const float* dirs, float* x_fm, int ldp, const float* keys, int n_step,
                           int n_cand, int n_extra, int* pick_all, void* stream) {
  if (R <= 0 || n < 2 || n > SMAX || !z || !beta || !beta0 || !beta_ray || nctl < 0 || (nctl > 0 && !ctl)) return -1;
  if (x_fm && (!origins || !dirs || (long long)ldp < (long long)R * n)) return -1;
  const bool picks = keys != nullptr;
  if (picks && (!pick_all || n_step < 1 || n_cand < 1 || n_extra < 2 || n_extra > n_step || n_step * n_cand > SMAX)) return -1;
  SamplerInitArgs a{z, R, n, beta, beta_min, beta_c, beta0, beta_ray, ctl, nctl, origins, dirs, x_fm, ldp, keys, n_step, n_cand, n_extra, pick_all,
                    (R + 3) / 4, picks ? (n_step * n_cand + 255) / 256 : 0};
  hipLaunchKernelGGL(sampler_init_kernel, dim3(a.init_blocks + n_cand * a.pick_parts), dim3(256), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

int neat_sampler_round(const float* z, int n, int R, const float* sdf_old, const float* sdf_new, const int* order, int n_old,
                       const float* beta_in, const float* beta0, float eps, int iters, float* sdf_out, float* beta_out, int* ctl, int round,
                       int max_rounds, float add_tiny, const float* u_refine, int N_refine, float* samples_refine, float* z_merged,
                       int* order_out, const float* origins, const float* dirs, float* x_fm, int ldp, const float* u_final,
                       int u_final_stride, int N_final, float* samples_final, float* z_final, int ld_final, void* stream) {
  if (R <= 0) return 0;
  const bool last = round + 1 >= max_rounds;
  if (n < 2 || n > SMAX || !z || !sdf_new || (order && !sdf_old) || !beta_in || !beta0 || !sdf_out || !beta_out || !ctl || round < 0 ||
      round >= max_rounds || N_final < 1 || N_final > SMAX || !u_final || !samples_final || !z_final || ld_final < n) return -1;
  if (!last && (N_refine < 1 || n + N_refine > SMAX || !u_refine || !samples_refine || !z_merged || !order_out ||
                (x_fm && (!origins || !dirs || (long long)ldp < (long long)R * N_refine)))) return -1;
  SamplerRoundArgs a{z, n, R, sdf_old, sdf_new, order, n_old, beta_in, beta0, eps, iters, sdf_out, beta_out, ctl, round, max_rounds, add_tiny,
                     u_refine, N_refine, samples_refine, z_merged, order_out, origins, dirs, x_fm, ldp, u_final, u_final_stride, N_final,
                     samples_final, z_final, ld_final, g_sampler_ablate};
  hipLaunchKernelGGL(sampler_round_kernel, dim3(R), dim3(BOUND_T), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

int neat_sampler_finish_picked(const float* samples, int N, const float* z_final, int ld_final, const int* n_final, const int* pick_all,
                               int n_step, int n_extra, float near, float far, int R, const int* eik_idx, float* z_vals, float* z_eik,
                               void* stream) {
  if (R <= 0) return 0;
  if (N + 2 + n_extra > SMAX || ld_final > SMAX || !samples || !z_final || !n_final || n_extra == 1 || n_extra < 0 || n_step < 1 || !eik_idx ||
      !z_vals || !z_eik) return -1;
  SamplerFinishArgs a{samples, N, z_final, 0, pick_all, n_extra, near, far, R, eik_idx, z_vals, z_eik, ld_final};
  a.n_final = n_final; a.n_step = n_step;
  hipLaunchKernelGGL(sampler_finish_kernel, dim3(R), dim3(64), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

int neat_sampler_bound_dev(const float* z, int n, int R, const float* sdf_old, const float* sdf_new, const int* order, int n_old,
                           const float* beta_in, const float* beta0, float eps, int iters, float* sdf_out, float* beta_out,
                           int* open, const int* gate, int gate_value, void* stream) {
  if (R <= 0) return 0;
  if (n < 2 || n > SMAX || !z || !sdf_new || !beta_in || !beta0 || !sdf_out || !beta_out || !open) return -1;
  SamplerBoundArgs a{z, n, R, sdf_old, sdf_new, order, n_old, beta_in, beta0, eps, iters, sdf_out, beta_out, open, gate, gate_value};
  hipLaunchKernelGGL(sampler_bound_kernel, dim3(R), dim3(BOUND_T), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

int neat_sampler_resample_dev(const float* z, const float* sdf, int n, int R, const float* beta, float add_tiny,
                              const float* u_refine, int N_refine, float* samples_refine, float* z_merged, int* order,
                              const float* u_final, int u_final_stride, int N_final, float* samples_final, float* z_final, int ld_final,
                              int* n_final, const int* open, int* cont, int round, int max_rounds, void* stream) {
  if (R <= 0) return 0;
  if (n < 2 || n > SMAX || N_refine < 1 || N_final < 1 || n + N_refine > SMAX || ld_final < n || !z || !sdf || !beta || !u_refine || !u_final ||
      !samples_refine || !z_merged || !order || !samples_final || !z_final || !n_final || !open || !cont || round < 0 || round >= max_rounds)
    return -1;
  SamplerResampleArgs a{};
  a.z = z; a.sdf = sdf; a.n = n; a.R = R; a.beta = beta; a.refine = 1; a.add_tiny = add_tiny;
  a.u = u_refine; a.u_stride = 0; a.N = N_refine; a.samples = samples_refine; a.z_merged = z_merged; a.order = order;
  a.open = open; a.cont = cont; a.round = round; a.max_rounds = max_rounds;
  a.u_final = u_final; a.u_final_stride = u_final_stride; a.N_final = N_final; a.samples_final = samples_final;
  a.z_final = z_final; a.ld_final = ld_final; a.n_final = n_final;
  hipLaunchKernelGGL(sampler_resample_kernel, dim3(R), dim3(BOUND_T), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

int neat_sampler_finish_dev(const float* samples, int N, const float* z_final, int ld_final, const int* n_final, const float* keys,
                            int n_extra, int* pick, float near, float far, int R, const int* eik_idx, float* z_vals, float* z_eik,
                            void* stream) {
  if (R <= 0) return 0;
  if (N + 2 + n_extra > SMAX || ld_final > SMAX || !samples || !z_final || !n_final || (n_extra > 0 && !pick) || n_extra == 1 || !eik_idx ||
      !z_vals || !z_eik) return -1;
  if (n_extra > 0) hipLaunchKernelGGL(sampler_pick_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, n_final, keys, n_extra, pick);
  SamplerFinishArgs a{samples, N, z_final, 0, pick, n_extra, near, far, R, eik_idx, z_vals, z_eik, ld_final};
  hipLaunchKernelGGL(sampler_finish_kernel, dim3(R), dim3(64), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

int neat_encode_lines(const float* lines, int N, int H, int W, float* lmap, int* label, unsigned char* valid, void* stream) {
  if (N < 0 || H <= 0 || W <= 0 || (N > 0 && !lines) || !lmap || !label) return -1;
  hipLaunchKernelGGL(encode_lines_kernel, grid1(H * W), dim3(256), 0, (hipStream_t)stream, lines, N, H, W, lmap, label, valid);
  return (int)hipGetLastError();
}

int neat_gather_batch(const int* pool, int npool, const long long* draw, int n, int W, const float* att, const float* rgb, const int* labels,
                      const float* lines, int nlines, float* uv, float* uv_proj, float* rgb_out, float* lines_out, long long* labels_out,
                      long long* pixel_out, void* stream) {
  if (n < 0 || npool <= 0 || W <= 0 || nlines <= 0 || !pool || !att || !rgb || !labels || !lines) return -1;
  if (n == 0) return 0;
  if (!draw || !uv || !uv_proj || !rgb_out || !lines_out || !labels_out || !pixel_out) return -1;
  GatherBatchArgs a{pool, draw, n, W, npool, att, rgb, labels, lines, nlines, uv, uv_proj, rgb_out, lines_out, labels_out, pixel_out};
  hipLaunchKernelGGL(gather_batch_kernel, grid1(n), dim3(256), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

int neat_copy_batch(const void* const* src, void* const* dst, const long long* nbytes, int n, void* stream) {
  if (n < 0 || n > COPY_BATCH_MAX) return -1;
  if (n == 0) return 0;
  if (!src || !dst || !nbytes) return -1;
  CopyBatchArgs a{};
  long long mx = 0;
  for (int i = 0; i < n; ++i) {
    if (!src[i] || !dst[i] || nbytes[i] < 0 || (nbytes[i] & 3) || ((size_t)src[i] & 3) || ((size_t)dst[i] & 3)) return -1;
    a.src[i] = (const unsigned*)src[i]; a.dst[i] = (unsigned*)dst[i]; a.words[i] = nbytes[i] >> 2;
    mx = nbytes[i] > mx ? nbytes[i] : mx;
  }
  a.n = n;
  const int by = (int)((mx / 4 + 1023) / 1024);          // 256 threads x 4 words per block
  hipLaunchKernelGGL(copy_batch_kernel, dim3(by < 1 ? 1 : (by > 1024 ? 1024 : by), n), dim3(256), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

int neat_adam_step(float* params, const float* const* grads, const long long* seg_offsets, const int* seg_steps, int nseg,
                   float* exp_avg, float* exp_avg_sq, float lr, float beta1, float beta2, float eps, void* stream) {
  if (nseg <= 0) return 0;
  if (!params || !grads || !seg_offsets || !seg_steps || !exp_avg || !exp_avg_sq || nseg > ADAM_MAXSEG) return -1;
  AdamSegs segs{};
  for (int s = 0; s < nseg; ++s) {
    segs.g[s] = grads[s]; segs.off[s] = seg_offsets[s];
    if (grads[s]) {
      if (seg_steps[s] < 1) return -1;
      const double bc1 = 1.0 - pow((double)beta1, (double)seg_steps[s]), bc2 = 1.0 - pow((double)beta2, (double)seg_steps[s]);
      segs.lr_over_bc1[s] = (float)((double)lr / bc1); segs.inv_sqrt_bc2[s] = (float)(1.0 / sqrt(bc2));
    }
  }
  segs.off[nseg] = seg_offsets[nseg];
  segs.nseg = nseg;
  const long long n = seg_offsets[nseg];
  if (n <= 0) return 0;
  const long long per_block = 1024LL * ADAM_PASSES;    // ADAM_PASSES float4 passes of 256 threads
  const long long blocks = (n + per_block - 1) / per_block;
  hipLaunchKernelGGL(adam_flat_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, params, segs, exp_avg,
                     exp_avg_sq, n, beta1, beta2, eps, (const float*)nullptr);
  return (int)hipGetLastError();
}

int neat_adam_step_coef(float* params, const float* const* grads, const long long* seg_offsets, int nseg, float* exp_avg, float* exp_avg_sq,
                        const float* coef, float beta1, float beta2, float eps, void* stream) {
  if (nseg <= 0) return 0;
  if (!params || !grads || !seg_offsets || !exp_avg || !exp_avg_sq || !coef || nseg > ADAM_MAXSEG) return -1;
  AdamSegs segs{};
  for (int s = 0; s < nseg; ++s) { segs.g[s] = grads[s]; segs.off[s] = seg_offsets[s]; }
  segs.off[nseg] = seg_offsets[nseg];
  segs.nseg = nseg;
  const long long n = seg_offsets[nseg];
  if (n <= 0) return 0;
  const long long per_block = 1024LL * ADAM_PASSES;
  const long long blocks = (n + per_block - 1) / per_block;
  hipLaunchKernelGGL(adam_flat_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, params, segs, exp_avg,
                     exp_avg_sq, n, beta1, beta2, eps, coef);
  return (int)hipGetLastError();
}

int neat_ffn_forward(const float* x, int J, const float* W0, const float* b0, const float* W1, const float* b1, const float* W2,
                     const float* b2, float* h1, float* h2, float* y, void* stream) {
  if (J <= 0) return 0;
  if (!x || !W0 || !b0 || !W1 || !b1 || !W2 || !b2 || !h1 || !h2 || !y) return -1;
  hipStream_t st = (hipStream_t)stream;
  if (J > FFN_FUSED_MIN_ROWS) {       // many rows: one fused launch, 8 rows per workgroup
    hipLaunchKernelGGL(ffn_forward_kernel, dim3((J + FFN_RB - 1) / FFN_RB), dim3(FFN_H), 0, st, x, J, W0, b0, W1, b1, W2, b2, h1, h2, y);
    return (int)hipGetLastError();
  }
  // few rows (the 64 latents of the ABC scenes): one launch per layer, each spread over rows x 32-output blocks
  const dim3 gh((J + FFN_RB - 1) / FFN_RB, FFN_H / 32), g3((J + FFN_RB - 1) / FFN_RB, 1);
  const float* nogate = nullptr; float* noy2 = nullptr;
  if (g_ffn_mfma) {      // the two 256 x 256 layers on the fp32 matrix pipe (tuning key 28)
    const dim3 gm((J + 31) / 32, FFN_H / 32);
    hipLaunchKernelGGL(ffn_mfma_kernel<false>, gm, dim3(256), 0, st, x, J, W0, b0, nogate, 1, h1);
    hipLaunchKernelGGL(ffn_mfma_kernel<false>, gm, dim3(256), 0, st, (const float*)h1, J, W1, b1, nogate, 1, h2);
  } else {
    hipLaunchKernelGGL(ffn_dense_kernel<false>, gh, dim3(256), 0, st, x, J, FFN_H, FFN_H, W0, b0, nogate, 1, h1, noy2);
    hipLaunchKernelGGL(ffn_dense_kernel<false>, gh, dim3(256), 0, st, (const float*)h1, J, FFN_H, FFN_H, W1, b1, nogate, 1, h2, noy2);
  }
  hipLaunchKernelGGL(ffn_dense_kernel<false>, g3, dim3(256), 0, st, (const float*)h2, J, FFN_H, 3, W2, b2, nogate, 0, y, noy2);
  return (int)hipGetLastError();
}

int neat_ffn_backward(const float* x, int J, const float* W0, const float* W1, const float* W2, const float* h1, const float* h2,
                      const float* dy, float* ws2, float* dx, float* dW0, float* db0, float* dW1, float* db1, float* dW2, float* db2,
                      void* stream) {
  if (J <= 0) return 0;
  if (!x || !W0 || !W1 || !W2 || !h1 || !h2 || !dy || !ws2 || !dx || !dW0 || !db0 || !dW1 || !db1 || !dW2 || !db2) return -1;
  float* d_a1 = ws2; float* d_a2 = ws2 + (size_t)J * FFN_H;
  hipStream_t st = (hipStream_t)stream;
  if (J > FFN_FUSED_MIN_ROWS)
    hipLaunchKernelGGL(ffn_backward_data_kernel, dim3((J + FFN_RB - 1) / FFN_RB), dim3(FFN_H), 0, st, dy, J, W0, W1, W2, h1, h2, d_a1, d_a2, dx);
  else {
    const dim3 gh((J + FFN_RB - 1) / FFN_RB, FFN_H / 32);
    const float* nobias = nullptr; const float* nogate = nullptr; float* noy2 = nullptr;
    hipLaunchKernelGGL(ffn_dense_kernel<true>, gh, dim3(256), 0, st, dy, J, 3, FFN_H, W2, nobias, h2, 0, d_a2, noy2);
    if (g_ffn_mfma) {
      const dim3 gm((J + 31) / 32, FFN_H / 32);
      hipLaunchKernelGGL(ffn_mfma_kernel<true>, gm, dim3(256), 0, st, (const float*)d_a2, J, W1, nobias, h1, 0, d_a1);
      hipLaunchKernelGGL(ffn_mfma_kernel<true>, gm, dim3(256), 0, st, (const float*)d_a1, J, W0, nobias, nogate, 0, dx);
    } else {
      hipLaunchKernelGGL(ffn_dense_kernel<true>, gh, dim3(256), 0, st, (const float*)d_a2, J, FFN_H, FFN_H, W1, nobias, h1, 0, d_a1, noy2);
      hipLaunchKernelGGL(ffn_dense_kernel<true>, gh, dim3(256), 0, st, (const float*)d_a1, J, FFN_H, FFN_H, W0, nobias, nogate, 0, dx, noy2);
    }
  }
  if (g_ffn_mfma)
    hipLaunchKernelGGL(ffn_wgrad_mfma_kernel, dim3(FFN_H / 32, FFN_H / 32, 3), dim3(64 * FFN_WNW), 0, (hipStream_t)stream, x, h1, h2, d_a1, d_a2, dy, J, dW0, db0,
                       dW1, db1, dW2, db2);
  else
  hipLaunchKernelGGL(ffn_backward_weights_kernel, dim3(FFN_H / FFN_RN, 3), dim3(64 * FFN_JG), 0, (hipStream_t)stream, x, h1, h2, d_a1, d_a2, dy, J, dW0, db0,
                     dW1, db1, dW2, db2);
  return (int)hipGetLastError();
}

int neat_loss_terms(const float* rgb, const float* rgb_gt, int R, const float* gtheta, int E, const float* loc3, const float* loc2c, int K,
                    const float* glo3, const float* glo2c, int J, float* scal, float* d_rgb, float* d_gtheta, float* pair_cost,
                    float eik_grad_scale, void* stream) {
  if (R <= 0 || !rgb || !rgb_gt || !scal || !d_rgb || E < 0 || K < 0 || J < 0) return -1;
  if ((E > 0 && (!gtheta || !d_gtheta)) || (K > 0 && J > 0 && (!loc3 || !loc2c || !glo3 || !glo2c || !pair_cost))) return -1;
  LossTermsArgs a{rgb, rgb_gt, R, gtheta, E, loc3, loc2c, (J > 0 ? K : 0), glo3, glo2c, J, scal, d_rgb, d_gtheta, pair_cost, eik_grad_scale};
  hipLaunchKernelGGL(loss_terms_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

int neat_loss_lines_terms(const float* pred_px, const float* pred_calib, const float* gt5, const float* Kmat, int L, float threshold, float* out3,
                          float* d_pred_calib, float grad_scale, const float* rgb, const float* rgb_gt, int R, const float* gtheta, int E,
                          const float* loc3, const float* loc2c, int K, const float* glo3, const float* glo2c, int J, float* scal, float* d_rgb,
                          float* d_gtheta, float* pair_cost, float eik_grad_scale, const float* w2c, const float* lines3d, float* d_lines3d,
                          void* stream) {
  if (L <= 0 || !pred_px || !pred_calib || !gt5 || !Kmat || !out3 || !d_pred_calib) return -1;
  if (d_lines3d && (!w2c || !lines3d)) return -1;
  if (R <= 0 || !rgb || !rgb_gt || !scal || !d_rgb || E < 0 || K < 0 || J < 0) return -1;
  if ((E > 0 && (!gtheta || !d_gtheta)) || (K > 0 && J > 0 && (!loc3 || !loc2c || !glo3 || !glo2c || !pair_cost))) return -1;
  LineLossesArgs l{pred_px, pred_calib, gt5, Kmat, L, threshold, out3, d_pred_calib, grad_scale, w2c, lines3d, d_lines3d};
  LossTermsArgs a{rgb, rgb_gt, R, gtheta, E, loc3, loc2c, (J > 0 ? K : 0), glo3, glo2c, J, scal, d_rgb, d_gtheta, pair_cost, eik_grad_scale};
  hipLaunchKernelGGL(loss_lines_terms_kernel, dim3(2), dim3(1024), 0, (hipStream_t)stream, l, a);
  return (int)hipGetLastError();
}

int neat_loss_pairs(const long long* ri, const long long* ci, const int* n_match, int Kmax, const float* loc3, const float* loc2c,
                    const float* loc2, const float* glo3, const float* glo2c, const float* glo2, int J, const float* pair_cost, float* scal,
                    float* d_glo3, float* d_glo2c, const float* line_loss, float w_eik, float w_line, float w_j3, float w_j2, int weighted_grads,
                    float* total, const float* w2c, void* stream) {
  if (Kmax < 0 || J <= 0 || !ri || !ci || !n_match || !loc3 || !loc2c || !loc2 || !glo3 || !glo2c || !glo2 || !pair_cost || !scal ||
      !d_glo3 || !d_glo2c || !line_loss) return -1;
  LossPairsArgs a{ri, ci, n_match, Kmax, loc3, loc2c, loc2, glo3, glo2c, glo2, J, pair_cost, scal, d_glo3, d_glo2c, line_loss, w_eik, w_line,
                  w_j3, w_j2, weighted_grads, total, w2c};
  hipLaunchKernelGGL(loss_pairs_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

int neat_l3d(const float* x, const float* o, const float* d, const float* normal, int R, float* l3d, void* stream) {
  if (R <= 0) return 0;
  if (!x || !o || !d || !normal || !l3d) return -1;
  hipLaunchKernelGGL(l3d_kernel, dim3((R + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, o, d, normal, R, l3d);
  return (int)hipGetLastError();
}

int neat_junction_cost(const float* cand2d, const float* gt2d, int V, int C, float* cost, void* stream) {
  if (V <= 0 || C <= 0) return 0;
  if (!cand2d || !gt2d || !cost) return -1;
  hipLaunchKernelGGL(junction_cost_kernel, dim3((V * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, cand2d, gt2d, V, C, cost);
  return (int)hipGetLastError();
}

int neat_junction_gate(const long long* rows, const long long* cols, int K, const float* cost, int C, const float* cand3d,
                       const float* cand2d, const float* cand2d_calib, int use_median, float* median, unsigned char* good, float* j3d,
                       float* j2d, float* j2d_calib, void* stream) {
  if (K <= 0) return 0;
  if (K > 2048 || !rows || !cols || !cost || !cand3d || !cand2d || !cand2d_calib || !good || !j3d || !j2d || !j2d_calib ||
      (use_median && !median)) return -1;
  JunctionGateArgs a{rows, cols, K, cost, C, cand3d, cand2d, cand2d_calib, use_median, median, good, j3d, j2d, j2d_calib};
  hipLaunchKernelGGL(junction_gate_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

int neat_inv_small(const float* A, int n, int lda, float* out, void* stream) {
  if (!A || !out || n < 1 || n > 4 || lda < n) return -1;
  hipLaunchKernelGGL(inv_small_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, A, n, lda, out);
  return (int)hipGetLastError();
}

int neat_camera_mats(const float* pose, const float* K, int kstride, float* w2c, float* K3, void* stream) {
  if (!pose || !K || !w2c || !K3 || kstride < 3) return -1;
  hipLaunchKernelGGL(camera_mats_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, pose, K, kstride, w2c, K3);
  return (int)hipGetLastError();
}

int neat_project2d(const float* K, const float* w2c, const float* X, int N, float* uv, void* stream) {
  if (N <= 0) return 0;
  if (!K || !w2c || !X || !uv) return -1;
  hipLaunchKernelGGL(project2d_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, K, w2c, X, N, uv);
  return (int)hipGetLastError();
}

int neat_project2d_pair(const float* K, const float* K2, const float* w2c, const float* X, int N, float* uv, float* uv2, void* stream) {
  if (N <= 0) return 0;
  if (!K || !K2 || !w2c || !X || !uv || !uv2) return -1;
  hipLaunchKernelGGL(project2d_pair_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, K, K2, w2c, X, N, uv, uv2);
  return (int)hipGetLastError();
}

int neat_camera_setup(const float* uv, const float* uv2, const float* pose, const float* K, int kstride, int R, float* dirs, float* origins,
                      float* dirs2, float* w2c, float* K3, void* stream) {
  if (R <= 0 || !uv || !pose || !K || !dirs || !origins || (uv2 && !dirs2) || !w2c || !K3 || kstride < 3) return -1;
  hipLaunchKernelGGL(camera_setup_kernel, grid1(R), dim3(256), 0, (hipStream_t)stream, uv, uv2, pose, K, kstride, R, dirs, origins, dirs2, w2c, K3);
  return (int)hipGetLastError();
}

int neat_project2d_backward(const float* K, const float* w2c, const float* X, int N, const float* d_uv, float* d_X, void* stream) {
  if (N <= 0) return 0;
  if (!K || !w2c || !X || !d_uv || !d_X) return -1;
  hipLaunchKernelGGL(project2d_bwd_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, K, w2c, X, N, d_uv, d_X);
  return (int)hipGetLastError();
}

int neat_line_loss(const float* pred, const float* gt, const float* weight, int R, float threshold, float* out2, float* per_line,
                   float* d_pred, void* stream) {
  if (R <= 0 || !pred || !gt || !weight || !out2 || !per_line || !d_pred) return -1;
  hipLaunchKernelGGL(line_loss_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, pred, gt, weight, R, threshold, out2, per_line, d_pred);
  return (int)hipGetLastError();
}

int neat_line_losses(const float* pred_px, const float* pred_calib, const float* gt5, const float* K, int R, float threshold, float* out3,
                     float* d_pred_calib, float grad_scale, void* stream) {
  if (R <= 0 || !pred_px || !pred_calib || !gt5 || !K || !out3 || !d_pred_calib) return -1;
  hipLaunchKernelGGL(line_losses_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, pred_px, pred_calib, gt5, K, R, threshold, out3, d_pred_calib, grad_scale);
  return (int)hipGetLastError();
}

size_t neat_lsap_ws_bytes(int nr, int nc) {
  const size_t mx = (size_t)(nr > nc ? nr : nc), mn = (size_t)(nr < nc ? nr : nc);
  return (mn + 2 * mx) * sizeof(double) + ((size_t)nr + (size_t)nc + 5 * mx + 2 * mn) * sizeof(int);
}

int neat_lsap(const float* cost, int nr, int nc, const unsigned char* row_mask, const unsigned char* col_mask, long long* row_ind,
              long long* col_ind, int* n_match, void* ws, void* stream) {
  if (nr < 0 || nc < 0 || !n_match) return -1;
  if (nr == 0 || nc == 0) return (int)hipMemsetAsync(n_match, 0, sizeof(int), (hipStream_t)stream);
  if (!cost || !row_ind || !col_ind || !ws) return -1;
  const size_t mx = (size_t)(nr > nc ? nr : nc), mn = (size_t)(nr < nc ? nr : nc);
  LsapArgs a{cost, nr, nc, row_mask, row_ind, col_ind, n_match, (double*)ws, (int*)((double*)ws + mn + 2 * mx)};
  a.col_mask = col_mask;
  const size_t dbytes = (mn + 2 * mx) * sizeof(double), ibytes = ((size_t)nr + (size_t)nc + 5 * mx + 2 * mn) * sizeof(int);
  size_t lds = 0;
  constexpr size_t LSAP_LDS_MAX = 156 * 1024;
  a.cost_lds_off = -1;
  if (dbytes + ibytes <= LSAP_LDS_MAX) {
    static DevOnce attr_set;
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lsap_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LSAP_LDS_MAX);
      if (e != hipSuccess) return (int)e;
      attr_set = true;
    }
    a.use_lds = 1; a.lds_int_off = (int)dbytes; lds = (dbytes + ibytes + 15) & ~(size_t)15;
    const size_t cbytes = (size_t)nr * nc * sizeof(float);
    if (lds + cbytes <= LSAP_LDS_MAX) { a.cost_lds_off = (int)lds; lds += cbytes; }      // the cost matrix too (8 x 2048 fits)
  }
  // threads: two columns per thread, whole waves (a small problem does not pay 16-wave barriers; eight columns per thread measured slower)
  int threads = (int)((mx + 1) / 2 + 63) / 64 * 64;
  threads = threads < 64 ? 64 : (threads > LSAP_WG ? LSAP_WG : threads);
  hipLaunchKernelGGL(lsap_kernel, dim3(1), dim3(threads), lds, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

size_t neat_dbscan_ws_bytes(int n) { return (size_t)n * 2 * sizeof(int); }

int neat_dbscan_means(const float* points, int n, double eps, float* centres, unsigned char* valid, int* count, void* ws, void* stream) {
  if (n <= 0 || n > DBSCAN_MAXN || !points || !centres || !valid || !count || !ws || !(eps > 0.0)) return -1;
  int* parent = (int*)ws; int* has_nb = parent + n;
  hipLaunchKernelGGL(dbscan_init_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, parent, has_nb, n);
  hipLaunchKernelGGL(dbscan_union_kernel, dim3(n), dim3(128), 0, (hipStream_t)stream, points, n, eps * eps, parent, has_nb);
  // mode 2 (18 n bytes of dynamic LDS fit next to the 32 KB of labels): O(n) fixed-point sums; else one wavefront per cluster over all
  // points, read from LDS (mode 1: 12 n bytes fit) or from global memory (mode 0)
  const size_t acc_bytes = (size_t)(n / 2) * 28 + (size_t)n * 4, pbytes = (size_t)n * 12;
  const int mode = acc_bytes <= 96 * 1024 ? 2 : (pbytes <= 96 * 1024 ? 1 : 0);
  if (mode) {
    static DevOnce attr_set;
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dbscan_finish_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
      if (e != hipSuccess) return (int)e;
      attr_set = true;
    }
  }
  hipLaunchKernelGGL(dbscan_finish_kernel, dim3(1), dim3(1024), mode == 2 ? acc_bytes : (mode == 1 ? pbytes : 0), (hipStream_t)stream, points, n,
                     parent, has_nb, centres, valid, count, mode);
  return (int)hipGetLastError();
}

int neat_volume_weights(const float* z, const float* sdf, int R, int S, const float* beta, float* weights, void* stream) {
  if (R <= 0 || S <= 0) return 0;
  hipLaunchKernelGGL(volume_weights_kernel, dim3((R + 3) / 4), dim3(WG), 0, (hipStream_t)stream, z, sdf, R, S, beta, weights);
  return (int)hipGetLastError();
}

}  // extern "C"

// Host orchestration + C ABI of libneat_hip.so (see include/neat_hip.h).
// One stream-ordered sequence of kernel launches per entry point; no allocation, no sync.
#include "kernels.hpp"
#include "../../include/neat_hip.h"
#include <math.h>
#include <stdio.h>
#include <vector>

using namespace neat;

namespace {

// ------------------------------------------------------------------------------------------------
// architecture (identical in all shipped confs: confs/abc-neat-a.conf:42-70, dtu.conf, bmvs.conf)
// ------------------------------------------------------------------------------------------------
constexpr int L_SDF = 0, L_REND = 9, L_ATTR = 14;
const int kO[NLAYERS] = {256, 256, 256, 217, 256, 256, 256, 256, 257, 256, 256, 256, 256, 3, 256, 256, 256, 256, 6};
const int kI[NLAYERS] = {39, 256, 256, 256, 256, 256, 256, 256, 256, 289, 256, 256, 256, 256, 265, 256, 256, 256, 256};
constexpr int PE_ROWS = 39, SMALL_R = 33, SMALL_A = 9;

inline int pad8(int k) { return (k + 7) & ~7; }
inline int tiles32(int n) { return (n + 31) / 32; }

struct PackLayout {
  PackDesc d[MAXPACKS];
  int npacks, nblocks;
  int fwd[NLAYERS], tr[NLAYERS];      // pack ids
  int row_off[NLAYERS + 1];
  size_t rowscale_off, total;
};

const PackLayout& pack_layout() {
  static PackLayout L;
  static bool init = false;
  if (init) return L;
  size_t off = 0;
  int np = 0, blk = 0;
  L.row_off[0] = 0;
  for (int l = 0; l < NLAYERS; ++l) L.row_off[l + 1] = L.row_off[l] + kO[l];
  for (int l = 0; l < NLAYERS; ++l) {
    const int perm = (l == L_REND) ? SMALL_R : (l == L_ATTR) ? SMALL_A : 0;
    const float scale = (l == 4) ? (float)(1.0 / sqrt(2.0)) : 1.0f;   // skip concat /sqrt2 (rend_a :87-88) folded in
    for (int t = 0; t < 2; ++t) {
      PackDesc& d = L.d[np];
      d.layer = l; d.transpose = t;
      d.N = t ? kI[l] : kO[l];
      d.K = t ? kO[l] : kI[l];
      d.Kpad = pad8(d.K); d.NT = tiles32(d.N);
      d.perm_split = perm; d.scale = scale;
      d.offset = (int)off; d.blk0 = blk;
      off += (size_t)d.NT * (d.Kpad / 2) * 64;
      blk += d.NT;
      (t ? L.tr : L.fwd)[l] = np;
      ++np;
    }
  }
  L.npacks = np; L.nblocks = blk;
  L.rowscale_off = off;
  off += (size_t)((L.row_off[NLAYERS] + 63) & ~63);
  L.total = off;
  init = true;
  return L;
}

NetPtrs to_ptrs(const neat_net_params* net) {
  NetPtrs p;
  for (int l = 0; l < NLAYERS; ++l) { p.v[l] = net->v[l]; p.g[l] = net->g[l]; p.b[l] = net->b[l]; p.O[l] = kO[l]; p.I[l] = kI[l]; }
  return p;
}

// ------------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------------
#define NEAT_CHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return (int)e_; } while (0)

// ------------------------------------------------------------------------------------------------
// optional in-stream kernel timing (bench.py): HIP events around every launch of the two GEMM-class kernels
// ------------------------------------------------------------------------------------------------
struct ProfSlot { hipEvent_t e0, e1; double flops; int cls; };
struct Prof {
  bool on = false;
  std::vector<ProfSlot> pool;
  size_t used = 0;
} g_prof;

inline ProfSlot* prof_begin(hipStream_t st, int cls, double flops) {
  if (!g_prof.on) return nullptr;
  if (g_prof.used == g_prof.pool.size()) {
    ProfSlot s{};
    if (hipEventCreate(&s.e0) != hipSuccess || hipEventCreate(&s.e1) != hipSuccess) return nullptr;
    g_prof.pool.push_back(s);
  }
  ProfSlot* s = &g_prof.pool[g_prof.used++];
  s->flops = flops; s->cls = cls;
  hipEventRecord(s->e0, st);
  return s;
}
inline void prof_end(hipStream_t st, ProfSlot* s) { if (s) hipEventRecord(s->e1, st); }

template <int EPI> hipError_t launch_layer_t(hipStream_t st, const LayerArgs& a, int ntiles_p) {
  static bool attr_set = false;
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

hipError_t launch_layer(hipStream_t st, int epi, const LayerArgs& a, int ntiles_p) {
  switch (epi) {
    case EPI_LINEAR: return launch_layer_t<EPI_LINEAR>(st, a, ntiles_p);
    case EPI_SOFTPLUS: return launch_layer_t<EPI_SOFTPLUS>(st, a, ntiles_p);
    case EPI_RELU: return launch_layer_t<EPI_RELU>(st, a, ntiles_p);
    case EPI_SIGMOID: return launch_layer_t<EPI_SIGMOID>(st, a, ntiles_p);
    case EPI_REV: return launch_layer_t<EPI_REV>(st, a, ntiles_p);
    case EPI_TAN: return launch_layer_t<EPI_TAN>(st, a, ntiles_p);
    case EPI_BWD: return launch_layer_t<EPI_BWD>(st, a, ntiles_p);
    case EPI_BWD_RELU: return launch_layer_t<EPI_BWD_RELU>(st, a, ntiles_p);
  }
  return hipErrorInvalidValue;
}

struct Ctx {
  hipStream_t st;
  const float* packed;
  const neat_net_params* net;
  int P, ldp;
  const float* pack(int id) const { return packed + pack_layout().d[id].offset; }
  const float* rowscale(int l) const { return packed + pack_layout().rowscale_off + pack_layout().row_off[l]; }
};

// out[n][p] = epi(Wm in + bias) with Wm = pack `pid`; N may be < pack N (only the leading rows are computed)
hipError_t layer(const Ctx& c, int pid, int epi, const float* in0, int rows0, const float* in1, int rows1,
                 const float* bias, int N, float* out0, float* out1 = nullptr, int n_split = 1 << 30,
                 const float* aux0 = nullptr, const float* aux1 = nullptr, int accumulate = 0) {
  const PackDesc& d = pack_layout().d[pid];
  LayerArgs a;
  a.in0 = in0; a.in1 = in1; a.rows0 = rows0; a.rows1 = rows1;
  a.Kpad = d.Kpad; a.Wp = c.pack(pid); a.bias = bias;
  a.N = N; a.NT = tiles32(N);
  a.ldp = c.ldp; a.out0 = out0; a.out1 = out1; a.n_split = n_split; a.accumulate = accumulate;
  a.aux0 = aux0; a.aux1 = aux1;
  if (rows0 + rows1 != d.K || N > d.N) return hipErrorInvalidValue;
  // tile stride inside the pack is Kpad/2*64 per 32 rows, independent of how many tiles we compute
  ProfSlot* ps = prof_begin(c.st, 0, 2.0 * N * d.K * (double)c.P);      // algorithmic flops: true N, K and point count
  hipError_t e = launch_layer(c.st, epi, a, c.ldp / BM);
  prof_end(c.st, ps);
  return e;
}

inline dim3 grid1(int n, int b = 256) { return dim3((n + b - 1) / b); }

// ------------------------------------------------------------------------------------------------
// workspaces (float offsets; every array is [rows][ldp])
// ------------------------------------------------------------------------------------------------
struct SdfWs {
  float *x, *E, *h[9], *out8, *sdf, *mask, *g, *u[8], *e0, *es;      // forward + adjoint
  float *Eh, *gh, *vh[9], *m[8], *abar8, *ones, *partial;            // backward
  size_t total;
};
constexpr int WSPLIT = 128;                 // point-splits of the weight-gradient reduction
constexpr int WLDN = 384, WLDK = 384;       // partial tile leading dims (>= 289+1, multiple of 128)

SdfWs sdf_ws(float* base, int ldp, int mode) {
  SdfWs w{};
  size_t off = 0;
  auto take = [&](int rows) { float* p = base ? base + off : nullptr; off += (size_t)rows * ldp; return p; };
  w.x = take(3); w.E = take(PE_ROWS + 1); w.sdf = take(1); w.mask = take(1); w.g = take(3);
  if (mode == 0) {
    float* a = take(256); float* b = take(256);
    for (int l = 1; l <= 8; ++l) w.h[l] = (l & 1) ? a : b;
    w.out8 = take(1);
  } else {
    for (int l = 1; l <= 8; ++l) w.h[l] = take(256);
    w.out8 = take(257);
    for (int l = 0; l < 8; ++l) w.u[l] = take(256);
    w.e0 = take(PE_ROWS); w.es = take(PE_ROWS);
    w.Eh = take(PE_ROWS); w.gh = take(3);
    for (int l = 1; l <= 8; ++l) w.vh[l] = take(256);
    for (int l = 0; l < 8; ++l) w.m[l] = take(256);
    w.abar8 = take(257); w.ones = take(1);
    w.partial = base ? base + off : nullptr;
    off += (size_t)WSPLIT * WLDN * WLDK;
  }
  w.total = off;
  return w;
}

struct HeadWs {
  float *small_r, *small_a, *hr[5], *ha[5], *rgb, *lin;             // forward
  float *zrgb, *dlin, *ar[4], *aa[4], *sc_r, *sc_a;                 // backward
  size_t total;
};
HeadWs head_ws(float* base, int ldp) {
  HeadWs w{};
  size_t off = 0;
  auto take = [&](int rows) { float* p = base ? base + off : nullptr; off += (size_t)rows * ldp; return p; };
  w.small_r = take(SMALL_R); w.small_a = take(SMALL_A);
  for (int l = 1; l <= 4; ++l) { w.hr[l] = take(256); w.ha[l] = take(256); }
  w.rgb = take(3); w.lin = take(6);
  w.zrgb = take(3); w.dlin = take(6);
  for (int l = 0; l < 4; ++l) { w.ar[l] = take(256); w.aa[l] = take(256); }
  w.sc_r = take(SMALL_R); w.sc_a = take(SMALL_A);
  w.total = off;
  return w;
}

inline int round_ldp(int P) { return (P + BM - 1) / BM * BM; }

// ------------------------------------------------------------------------------------------------
// SDF network chains
// ------------------------------------------------------------------------------------------------
// primal chain  (ImplicitNetwork.forward, rend_a :78-96)
hipError_t sdf_primal(const Ctx& c, const SdfWs& w, int out_rows) {
  const PackLayout& L = pack_layout();
  hipLaunchKernelGGL(posenc6_kernel, grid1(c.ldp), dim3(256), 0, c.st, w.x, c.ldp, w.E);
  hipError_t e;
  for (int l = 0; l < 8; ++l) {
    const float* in0 = l == 0 ? w.E : w.h[l];
    const int rows0 = l == 0 ? PE_ROWS : (l == 4 ? 217 : 256);
    const float* in1 = l == 4 ? w.E : nullptr;
    const int rows1 = l == 4 ? PE_ROWS : 0;
    if ((e = layer(c, L.fwd[l], EPI_SOFTPLUS, in0, rows0, in1, rows1, c.net->b[l], kO[l], w.h[l + 1])) != hipSuccess) return e;
  }
  return layer(c, L.fwd[8], EPI_LINEAR, w.h[8], 256, nullptr, 0, c.net->b[8], out_rows, w.out8);
}

// adjoint chain: u_l = d sdf_raw / d a_l, then e0/es = cotangent of the PE rows (autograd.grad at rend_a :121-127)
hipError_t sdf_adjoint(const Ctx& c, const SdfWs& w) {
  const PackLayout& L = pack_layout();
  hipLaunchKernelGGL(adjoint_seed_kernel, dim3((c.ldp + 255) / 256, 256), dim3(256), 0, c.st,
                     c.net->v[8], c.rowscale(8), w.h[8], c.ldp, w.u[7]);
  hipError_t e;
  for (int l = 7; l >= 1; --l) {
    const int rows = kO[l];
    if (l == 4) e = layer(c, L.tr[l], EPI_REV, w.u[l], rows, nullptr, 0, nullptr, 256, w.u[l - 1], w.es, 217, w.h[l]);
    else e = layer(c, L.tr[l], EPI_REV, w.u[l], rows, nullptr, 0, nullptr, kI[l], w.u[l - 1], nullptr, 1 << 30, w.h[l]);
    if (e != hipSuccess) return e;
  }
  return layer(c, L.tr[0], EPI_LINEAR, w.u[0], 256, nullptr, 0, nullptr, PE_ROWS, w.e0);
}

hipError_t wgrad(const Ctx& c, const SdfWs& w, int layer_id, const WgradPair* pairs, int npairs, int N, int Kt,
                 const neat_net_grads* gr) {
  if (!gr->dv[layer_id]) return hipSuccess;
  WgradArgs a{};
  for (int q = 0; q < npairs; ++q) a.pair[q] = pairs[q];
  a.npairs = npairs; a.N = N; a.Kt = Kt; a.P = c.P; a.ldp = c.ldp;
  int splits = WSPLIT;
  int chunk = ((c.ldp + splits - 1) / splits + WBP - 1) / WBP * WBP;
  if (chunk < 2 * WBP) chunk = 2 * WBP;
  splits = (c.P + chunk - 1) / chunk;
  a.chunk = chunk; a.partial = w.partial; a.Nld = WLDN; a.Kld = WLDK;
  const int ntile = (N + 127) / 128;
  a.ktiles = (Kt + 127) / 128;
  double wflops = 0.0;
  for (int q = 0; q < npairs; ++q) {
    const int kb = pairs[q].rowsB[0] + pairs[q].rowsB[1] + pairs[q].rowsB[2];
    wflops += 2.0 * pairs[q].rowsA * kb * (double)c.P;
  }
  ProfSlot* ps = prof_begin(c.st, 1, wflops);
  hipLaunchKernelGGL(wgrad_kernel, dim3(ntile * a.ktiles, splits), dim3(WG), 0, c.st, a);
  prof_end(c.st, ps);
  WreduceArgs r{};
  r.partial = w.partial; r.splits = splits; r.Nld = WLDN; r.Kld = WLDK;
  r.O = kO[layer_id]; r.I = kI[layer_id];
  r.perm_split = layer_id == L_REND ? SMALL_R : layer_id == L_ATTR ? SMALL_A : 0;
  r.scale = layer_id == 4 ? (float)(1.0 / sqrt(2.0)) : 1.0f;
  r.v = c.net->v[layer_id]; r.g = c.net->g[layer_id];
  r.dv = gr->dv[layer_id]; r.dg = gr->dg[layer_id]; r.db = gr->db[layer_id];
  r.bias_col = Kt - 1;
  hipLaunchKernelGGL(wreduce_wnorm_kernel, dim3(r.O), dim3(WG), 0, c.st, r);
  return hipGetLastError();
}

// double backward + backward: w.gh (cotangent of normals, masked) and w.abar8 (cotangent of lin8 output) are set
hipError_t sdf_backward_chains(const Ctx& c, const SdfWs& w, const neat_net_grads* gr) {
  const PackLayout& L = pack_layout();
  hipError_t e;
  hipLaunchKernelGGL(posenc6_tangent_kernel, grid1(c.ldp), dim3(256), 0, c.st, w.x, w.gh, c.ldp, w.Eh);
  hipLaunchKernelGGL(ones_kernel, grid1(c.ldp), dim3(256), 0, c.st, w.ones, c.P, c.ldp);
  // tangent chain (forward-mode along g^): vh_{l+1} = tangent of h_{l+1}, m_l = extra cotangent of a_l
  for (int l = 0; l < 8; ++l) {
    const float* in0 = l == 0 ? w.Eh : w.vh[l];
    const int rows0 = l == 0 ? PE_ROWS : (l == 4 ? 217 : 256);
    const float* in1 = l == 4 ? w.Eh : nullptr;
    const int rows1 = l == 4 ? PE_ROWS : 0;
    if ((e = layer(c, L.fwd[l], EPI_TAN, in0, rows0, in1, rows1, nullptr, kO[l], w.vh[l + 1], w.m[l], 1 << 30,
                   w.h[l + 1], w.u[l])) != hipSuccess) return e;
  }
  // reverse chain: a^_{l-1} = (W_l^T a^_l) phi'(a_{l-1}) + m_{l-1}   (in place in m)
  if ((e = layer(c, L.tr[8], EPI_BWD, w.abar8, 257, nullptr, 0, nullptr, 256, w.m[7], nullptr, 1 << 30, w.h[8], w.m[7])) != hipSuccess) return e;
  for (int l = 7; l >= 1; --l) {
    const int N = l == 4 ? 217 : kI[l];
    if ((e = layer(c, L.tr[l], EPI_BWD, w.m[l], kO[l], nullptr, 0, nullptr, N, w.m[l - 1], nullptr, 1 << 30, w.h[l], w.m[l - 1])) != hipSuccess) return e;
  }
  // weight gradients: dW_l = a^_l in_l^T + u_l vhat_l^T  (+ bias column from the ones row)
  for (int l = 0; l <= 8; ++l) {
    WgradPair pr[2] = {};
    pr[0].A = l == 8 ? w.abar8 : w.m[l]; pr[0].rowsA = kO[l];
    pr[1].A = l == 8 ? w.ones : w.u[l];  pr[1].rowsA = l == 8 ? 1 : kO[l];
    int Kt;
    if (l == 0) {
      pr[0].B[0] = w.E; pr[0].rowsB[0] = PE_ROWS; pr[0].B[1] = w.ones; pr[0].rowsB[1] = 1;
      pr[1].B[0] = w.Eh; pr[1].rowsB[0] = PE_ROWS;
      Kt = PE_ROWS + 1;
    } else if (l == 4) {
      pr[0].B[0] = w.h[4]; pr[0].rowsB[0] = 217; pr[0].B[1] = w.E; pr[0].rowsB[1] = PE_ROWS; pr[0].B[2] = w.ones; pr[0].rowsB[2] = 1;
      pr[1].B[0] = w.vh[4]; pr[1].rowsB[0] = 217; pr[1].B[1] = w.Eh; pr[1].rowsB[1] = PE_ROWS;
      Kt = 257;
    } else {
      pr[0].B[0] = w.h[l]; pr[0].rowsB[0] = 256; pr[0].B[1] = w.ones; pr[0].rowsB[1] = 1;
      pr[1].B[0] = w.vh[l]; pr[1].rowsB[0] = 256;
      Kt = 257;
    }
    if ((e = wgrad(c, w, l, pr, 2, kO[l], Kt, gr)) != hipSuccess) return e;
  }
  return hipSuccess;
}

// ------------------------------------------------------------------------------------------------
// heads
// ------------------------------------------------------------------------------------------------
hipError_t heads_forward(const Ctx& c, const HeadWs& h, const float* feat_fm) {
  const PackLayout& L = pack_layout();
  hipError_t e;
  for (int head = 0; head < 2; ++head) {
    const int base = head ? L_ATTR : L_REND;
    float* const* hh = head ? h.ha : h.hr;
    const float* small = head ? h.small_a : h.small_r;
    const int srows = head ? SMALL_A : SMALL_R;
    if ((e = layer(c, L.fwd[base], EPI_RELU, feat_fm, 256, small, srows, c.net->b[base], 256, hh[1])) != hipSuccess) return e;
    for (int l = 1; l < 4; ++l)
      if ((e = layer(c, L.fwd[base + l], EPI_RELU, hh[l], 256, nullptr, 0, c.net->b[base + l], 256, hh[l + 1])) != hipSuccess) return e;
    if (head == 0) e = layer(c, L.fwd[base + 4], EPI_SIGMOID, hh[4], 256, nullptr, 0, c.net->b[base + 4], 3, h.rgb);
    else e = layer(c, L.fwd[base + 4], EPI_LINEAR, hh[4], 256, nullptr, 0, c.net->b[base + 4], 6, h.lin);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

// zrgb / dlin hold the cotangents of the heads' last linear outputs; accumulates the feature cotangent into
// abar8 rows 1..256 (render overwrites, attraction adds) and the small-input cotangents into sc_r / sc_a.
hipError_t heads_backward(const Ctx& c, const HeadWs& h, const SdfWs& w, const float* feat_fm, const neat_net_grads* gr) {
  const PackLayout& L = pack_layout();
  hipError_t e;
  for (int head = 0; head < 2; ++head) {
    const int base = head ? L_ATTR : L_REND;
    float* const* hh = head ? h.ha : h.hr;
    float* const* ab = head ? h.aa : h.ar;
    const float* top = head ? h.dlin : h.zrgb;
    const int top_rows = head ? 6 : 3;
    const float* small = head ? h.small_a : h.small_r;
    const int srows = head ? SMALL_A : SMALL_R;
    if ((e = layer(c, L.tr[base + 4], EPI_BWD_RELU, top, top_rows, nullptr, 0, nullptr, 256, ab[3], nullptr, 1 << 30, hh[4])) != hipSuccess) return e;
    for (int l = 3; l >= 1; --l)
      if ((e = layer(c, L.tr[base + l], EPI_BWD_RELU, ab[l], 256, nullptr, 0, nullptr, 256, ab[l - 1], nullptr, 1 << 30, hh[l])) != hipSuccess) return e;
    if ((e = layer(c, L.tr[base], EPI_LINEAR, ab[0], 256, nullptr, 0, nullptr, 256 + srows, w.abar8 + c.ldp,
                   head ? h.sc_a : h.sc_r, 256, nullptr, nullptr, head)) != hipSuccess) return e;
    for (int l = 0; l <= 4; ++l) {
      WgradPair pr[1] = {};
      pr[0].A = l == 4 ? top : ab[l]; pr[0].rowsA = kO[base + l];
      int Kt;
      if (l == 0) {
        pr[0].B[0] = feat_fm; pr[0].rowsB[0] = 256; pr[0].B[1] = small; pr[0].rowsB[1] = srows; pr[0].B[2] = w.ones; pr[0].rowsB[2] = 1;
        Kt = 256 + srows + 1;
      } else {
        pr[0].B[0] = hh[l]; pr[0].rowsB[0] = 256; pr[0].B[1] = w.ones; pr[0].rowsB[1] = 1;
        Kt = 257;
      }
      if ((e = wgrad(c, w, base + l, pr, 1, kO[base + l], Kt, gr)) != hipSuccess) return e;
    }
  }
  return hipSuccess;
}

__global__ void volume_weights_kernel(const float* __restrict__ z, const float* __restrict__ sdf, int R, int S, const float* __restrict__ beta_ptr,
                                      float* __restrict__ weights) {
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

// abar8 row 0 <- (1-mask) d_sdf ; rows 1.. <- d_feat ; (+ d_out257)
__global__ void build_abar8_kernel(const float* __restrict__ d_out257, const float* __restrict__ d_sdf,
                                   const float* __restrict__ d_feat, const float* __restrict__ mask, int P, int ldp,
                                   float* __restrict__ abar8) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = blockIdx.y;
  if (p >= ldp) return;
  float v = 0.0f;
  if (p < P) {
    if (d_out257) v += d_out257[(size_t)p * 257 + n];
    if (n == 0) { if (d_sdf) v += d_sdf[p] * (1.0f - mask[p]); }
    else if (d_feat) v += d_feat[(size_t)p * 256 + (n - 1)];
  }
  abar8[(size_t)n * ldp + p] = v;
}

}  // namespace

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

int neat_abi_version(void) { return 1; }

int neat_prof_enable(int on) {
  g_prof.on = on != 0;
  g_prof.used = 0;
  return 0;
}

int neat_prof_collect(int cls, double* total_ms, double* total_flops, int* launches) {
  double ms = 0.0, fl = 0.0;
  int n = 0;
  for (size_t i = 0; i < g_prof.used; ++i) {
    ProfSlot& s = g_prof.pool[i];
    if (s.cls != cls) continue;
    hipError_t e = hipEventSynchronize(s.e1);
    if (e != hipSuccess) return (int)e;
    float t = 0.0f;
    if ((e = hipEventElapsedTime(&t, s.e0, s.e1)) != hipSuccess) return (int)e;
    ms += t; fl += s.flops; ++n;
  }
  if (total_ms) *total_ms = ms;
  if (total_flops) *total_flops = fl;
  if (launches) *launches = n;
  return 0;
}

size_t neat_packed_floats(void) { return pack_layout().total; }

int neat_pack_weights(const neat_net_params* net, float* packed, void* stream) {
  if (!net || !packed) return -1;
  hipStream_t st = (hipStream_t)stream;
  const PackLayout& L = pack_layout();
  RowScaleArgs ra;
  ra.net = to_ptrs(net);
  ra.rowscale = packed + L.rowscale_off;
  for (int l = 0; l <= NLAYERS; ++l) ra.row_off[l] = L.row_off[l];
  hipLaunchKernelGGL(rowscale_kernel, dim3((L.row_off[NLAYERS] + 3) / 4), dim3(WG), 0, st, ra);
  PackArgs pa;
  pa.net = ra.net; pa.rowscale = ra.rowscale;
  for (int l = 0; l <= NLAYERS; ++l) pa.row_off[l] = L.row_off[l];
  for (int i = 0; i < L.npacks; ++i) pa.d[i] = L.d[i];
  pa.npacks = L.npacks; pa.out = packed;
  hipLaunchKernelGGL(pack_kernel, dim3(L.nblocks), dim3(WG), 0, st, pa);
  return (int)hipGetLastError();
}

int neat_camera_rays(const float* uv, const float* pose, const float* K, int kstride, int R, float* dirs, void* stream) {
  if (R <= 0) return 0;
  hipLaunchKernelGGL(camera_rays_kernel, grid1(R), dim3(256), 0, (hipStream_t)stream, uv, pose, K, kstride, R, dirs);
  return (int)hipGetLastError();
}

size_t neat_sdf_ws_floats(int P, int mode) { return sdf_ws(nullptr, round_ldp(P), mode).total; }

int neat_sdf_forward(const float* packed, const neat_net_params* net, const float* x, int P, int mode,
                     float radius, float scale, float* ws, float* out257, float* sdf, float* feat, float* grad,
                     void* stream) {
  if (P <= 0) return 0;
  if (!packed || !net || !x || !ws) return -1;
  Ctx c{(hipStream_t)stream, packed, net, P, round_ldp(P)};
  SdfWs w = sdf_ws(ws, c.ldp, mode);
  hipLaunchKernelGGL(rm_to_fm_kernel, grid1(c.ldp), dim3(256), 0, c.st, x, P, 3, c.ldp, w.x);
  if (mode == 0) {
    NEAT_CHECK(sdf_primal(c, w, 1));
    hipLaunchKernelGGL(sdf_finalize_kernel, grid1(c.ldp), dim3(256), 0, c.st, w.x, w.out8, (const float*)nullptr,
                       (const float*)nullptr, P, c.ldp, radius, scale, w.sdf, (float*)nullptr, (float*)nullptr, sdf, (float*)nullptr);
    return (int)hipGetLastError();
  }
  NEAT_CHECK(sdf_primal(c, w, 257));
  NEAT_CHECK(sdf_adjoint(c, w));
  hipLaunchKernelGGL(sdf_finalize_kernel, grid1(c.ldp), dim3(256), 0, c.st, w.x, w.out8, w.e0, w.es, P, c.ldp, radius, scale,
                     w.sdf, w.g, w.mask, sdf, grad);
  if (out257) hipLaunchKernelGGL(fm_to_rm_kernel, grid1(P), dim3(256), 0, c.st, w.out8, P, 257, c.ldp, out257, 0);
  if (feat) hipLaunchKernelGGL(fm_to_rm_kernel, grid1(P), dim3(256), 0, c.st, w.out8 + c.ldp, P, 256, c.ldp, feat, 0);
  return (int)hipGetLastError();
}

int neat_sdf_backward(const float* packed, const neat_net_params* net, float* ws, int P,
                      const float* d_out257, const float* d_sdf, const float* d_feat, const float* d_grad,
                      const neat_net_grads* grads, void* stream) {
  if (P <= 0) return 0;
  if (!packed || !net || !ws || !grads) return -1;
  Ctx c{(hipStream_t)stream, packed, net, P, round_ldp(P)};
  SdfWs w = sdf_ws(ws, c.ldp, 1);
  hipLaunchKernelGGL(build_abar8_kernel, dim3((c.ldp + 255) / 256, 257), dim3(256), 0, c.st, d_out257, d_sdf, d_feat, w.mask, P, c.ldp, w.abar8);
  hipLaunchKernelGGL(normal_cotangent_kernel, grid1(c.ldp), dim3(256), 0, c.st, (const float*)nullptr, (const float*)nullptr,
                     d_grad, w.mask, P, c.ldp, w.gh);
  NEAT_CHECK(sdf_backward_chains(c, w, grads));
  return (int)hipGetLastError();
}

size_t neat_heads_ws_floats(int P) {
  const int ldp = round_ldp(P);
  return head_ws(nullptr, ldp).total + (size_t)(3 + 3 + 256) * ldp;
}

int neat_heads_forward(const float* packed, const neat_net_params* net, const float* points, const float* normals,
                       const float* view_dirs, const float* feats, int P, float* ws, float* rgb, float* lines, void* stream) {
  if (P <= 0) return 0;
  if (!packed || !net || !ws) return -1;
  Ctx c{(hipStream_t)stream, packed, net, P, round_ldp(P)};
  HeadWs h = head_ws(ws, c.ldp);
  float* x_fm = ws + h.total; float* g_fm = x_fm + 3 * (size_t)c.ldp; float* f_fm = g_fm + 3 * (size_t)c.ldp;
  hipLaunchKernelGGL(rm_to_fm_kernel, grid1(c.ldp), dim3(256), 0, c.st, points, P, 3, c.ldp, x_fm);
  hipLaunchKernelGGL(rm_to_fm_kernel, grid1(c.ldp), dim3(256), 0, c.st, normals, P, 3, c.ldp, g_fm);
  hipLaunchKernelGGL(rm_to_fm_kernel, grid1(c.ldp), dim3(256), 0, c.st, feats, P, 256, c.ldp, f_fm);
  hipLaunchKernelGGL(head_inputs_kernel, grid1(c.ldp), dim3(256), 0, c.st, x_fm, g_fm, view_dirs, P, 1, c.ldp, h.small_r, h.small_a);
  NEAT_CHECK(heads_forward(c, h, f_fm));
  if (rgb) hipLaunchKernelGGL(fm_to_rm_kernel, grid1(P), dim3(256), 0, c.st, h.rgb, P, 3, c.ldp, rgb, 0);
  if (lines) {   // y = p + offsets.reshape(2,3)   (rend_a :195)
    hipLaunchKernelGGL(lines_from_offsets_kernel, grid1(P), dim3(256), 0, c.st, h.lin, x_fm, P, c.ldp, lines);
  }
  return (int)hipGetLastError();
}

size_t neat_render_ws_floats(int R, int S) {
  const int ldp = round_ldp(R * S);
  return sdf_ws(nullptr, ldp, 1).total + head_ws(nullptr, ldp).total;
}

int neat_render_forward(const float* packed, const neat_net_params* net, const float* origins, const float* dirs,
                        const float* z, int R, int S, const float* beta, float radius, float scale, float* ws,
                        float* points, float* weights, float* sdf, float* rgb, float* lines3d, float* depth,
                        float* xyz, float* normal_map, void* stream) {
  if (R <= 0 || S <= 0) return 0;
  if (!packed || !net || !ws || !origins || !dirs || !z || !rgb || !lines3d || !depth || !xyz) return -1;
  const int P = R * S;
  Ctx c{(hipStream_t)stream, packed, net, P, round_ldp(P)};
  SdfWs w = sdf_ws(ws, c.ldp, 1);
  HeadWs h = head_ws(ws + w.total, c.ldp);
  hipLaunchKernelGGL(points_from_rays_kernel, grid1(c.ldp), dim3(256), 0, c.st, origins, dirs, z, R, S, c.ldp, w.x, points);
  NEAT_CHECK(sdf_primal(c, w, 257));
  NEAT_CHECK(sdf_adjoint(c, w));
  hipLaunchKernelGGL(sdf_finalize_kernel, grid1(c.ldp), dim3(256), 0, c.st, w.x, w.out8, w.e0, w.es, P, c.ldp, radius, scale,
                     w.sdf, w.g, w.mask, sdf, (float*)nullptr);
  hipLaunchKernelGGL(head_inputs_kernel, grid1(c.ldp), dim3(256), 0, c.st, w.x, w.g, dirs, P, S, c.ldp, h.small_r, h.small_a);
  NEAT_CHECK(heads_forward(c, h, w.out8 + c.ldp));
  CompositeArgs ca;
  ca.z = z; ca.sdf = w.sdf; ca.dirs = dirs; ca.x_fm = w.x; ca.rgb_fm = h.rgb; ca.lin_fm = h.lin; ca.g_fm = w.g;
  ca.R = R; ca.S = S; ca.ldp = c.ldp; ca.beta_ptr = beta;
  ca.weights = weights; ca.rgb = rgb; ca.lines3d = lines3d; ca.depth = depth; ca.xyz = xyz; ca.normal_map = normal_map;
  hipLaunchKernelGGL(composite_fwd_kernel, dim3((R + 3) / 4), dim3(WG), 0, c.st, ca);
  return (int)hipGetLastError();
}

int neat_render_backward(const float* packed, const neat_net_params* net, float* ws, const float* dirs, const float* z,
                         int R, int S, const float* beta, const float* d_rgb, const float* d_lines3d, const float* d_depth,
                         const float* d_xyz, const neat_net_grads* grads, float* dbeta_ray, void* stream) {
  if (R <= 0 || S <= 0) return 0;
  if (!packed || !net || !ws || !grads) return -1;
  const int P = R * S;
  Ctx c{(hipStream_t)stream, packed, net, P, round_ldp(P)};
  SdfWs w = sdf_ws(ws, c.ldp, 1);
  HeadWs h = head_ws(ws + w.total, c.ldp);
  CompositeBwdArgs cb;
  cb.z = z; cb.sdf = w.sdf; cb.dirs = dirs; cb.mask = w.mask; cb.x_fm = w.x; cb.rgb_fm = h.rgb;
  cb.R = R; cb.S = S; cb.ldp = c.ldp; cb.beta_ptr = beta;
  cb.d_rgb = d_rgb; cb.d_lines3d = d_lines3d; cb.d_depth = d_depth; cb.d_xyz = d_xyz;
  cb.zrgb_fm = h.zrgb; cb.dlin_fm = h.dlin; cb.dsdf_row = w.abar8; cb.dbeta_ray = dbeta_ray;
  // padded columns of the cotangent arrays must be finite zeros (the weight-gradient kernels mask p >= P anyway)
  hipLaunchKernelGGL(ones_kernel, grid1(c.ldp), dim3(256), 0, c.st, w.ones, P, c.ldp);
  hipLaunchKernelGGL(composite_bwd_kernel, dim3((R + 3) / 4), dim3(WG), 0, c.st, cb);
  NEAT_CHECK(heads_backward(c, h, w, w.out8 + c.ldp, grads));
  hipLaunchKernelGGL(normal_cotangent_kernel, grid1(c.ldp), dim3(256), 0, c.st, h.sc_r, h.sc_a, (const float*)nullptr, w.mask, P, c.ldp, w.gh);
  NEAT_CHECK(sdf_backward_chains(c, w, grads));
  return (int)hipGetLastError();
}

int neat_volume_weights(const float* z, const float* sdf, int R, int S, const float* beta, float* weights, void* stream) {
  if (R <= 0 || S <= 0) return 0;
  hipLaunchKernelGGL(volume_weights_kernel, dim3((R + 3) / 4), dim3(WG), 0, (hipStream_t)stream, z, sdf, R, S, beta, weights);
  return (int)hipGetLastError();
}

}  // extern "C"

// Per-ray kernels of the depth samplers (reference: code/model/ray_sampler.py).
//  * ErrorBoundSampler (VolSDF Algorithm 1, :130-293): sampler_bound_kernel (one 4-wave workgroup per ray: d*, beta bisection),
//    sampler_resample_kernel / sampler_finish_kernel / sampler_pick_kernel (one wavefront per ray); the ray's samples (<= SMAX) live
//    in LDS, prefix sums are fp64 wave scans on the DPP network with a cross-chunk carry.  The batch-global convergence test of the
//    reference (`beta.max() > beta0`, :200) is a device flag: read by the host once per round (the reference's single sync), or --
//    device-decided mode -- consumed by the next round's launches themselves (gates, see SamplerResampleArgs).
//  * sample_pdf + the sort of get_z_vals_fine (:16-59, :97-106): sample_pdf_kernel.
//  * UniformSampler.get_z_vals (:61-95): uniform_depths_kernel.
#pragma once
#include "kernels.hpp"

namespace neat {

constexpr int SMAX = 1024;      // max samples per ray inside the sampler (reference: 128 * max_total_iters = 640)

// fp64 wave scan / sum.  The reference runs on torch-CPU, whose cumsum accumulates float rows in double and rounds every output once
// (at::acc_type<float, false> = double); an fp32 scan rounds at every step and moves CDF knots by a few 1e-8, enough to flip the
// `denom < 1e-5` rule of the inverse-CDF step for bins whose mass sits at the threshold (every empty bin of `weights + 1e-5`).
// The scans run on the DPP network (row shifts inside 16-lane rows, then the two row broadcasts), two 32-bit moves per fp64 step:
// a __shfl_up of a double is two ds_bpermute round trips through the LDS crossbar per step, and with one to four waves per SIMD
// (a ray per workgroup) that latency was most of the bound kernel's time.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_zero_d(double v) {      // lanes without a source (or outside ROW_MASK) receive 0
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_incl_scan_d(double v, int /*lane*/) {
  v += dpp_zero_d<0x111, 0xf>(v);      // row_shr:1
  v += dpp_zero_d<0x112, 0xf>(v);      // row_shr:2
  v += dpp_zero_d<0x114, 0xf>(v);      // row_shr:4
  v += dpp_zero_d<0x118, 0xf>(v);      // row_shr:8   -> inclusive scan inside every row of 16 lanes
  v += dpp_zero_d<0x142, 0xa>(v);      // row_bcast:15 into rows 1 and 3
  v += dpp_zero_d<0x143, 0xc>(v);      // row_bcast:31 into rows 2 and 3
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
  v = wave_incl_scan_d(v, 0);
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_max_f(float v) {         // lanes without a source keep their own value
  const int t = __builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
  return fmaxf(v, __int_as_float(t));
}
__device__ __forceinline__ float wave_max(float v) {
  v = dpp_max_f<0x111, 0xf>(v);
  v = dpp_max_f<0x112, 0xf>(v);
  v = dpp_max_f<0x114, 0xf>(v);
  v = dpp_max_f<0x118, 0xf>(v);
  v = dpp_max_f<0x142, 0xa>(v);
  v = dpp_max_f<0x143, 0xc>(v);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// d* of Theorem 1 per interval (:161-173)
__device__ __forceinline__ float interval_bound(float a, float d0, float d1) {
  const float b = fabsf(d0), c = fabsf(d1);
  const bool c1 = a * a + b * b <= c * c;
  const bool c2 = a * a + c * c <= b * b;
  float ds = 0.0f;
  if (c1) ds = b;
  if (c2) ds = c;
  const float s = (a + b + c) / 2.0f;
  const float area = s * (s - a) * (s - b) * (s - c);
  if (!c1 && !c2 && (b + c - a > 0.0f)) ds = 2.0f * sqrtf(area) / a;
  const float sg0 = (d0 > 0.f) ? 1.f : ((d0 < 0.f) ? -1.f : 0.f), sg1 = (d1 > 0.f) ? 1.f : ((d1 < 0.f) ? -1.f : 0.f);
  return (sg1 * sg0 == 1.0f) ? ds : 0.0f;
}

// max_i of the opacity error bound for one beta (get_error_bound, :285-293); arrays in LDS, m = n-1 intervals.  One workgroup of
// BOUND_T threads per ray: thread t owns the PER consecutive intervals [t*PER, (t+1)*PER): it sums its terms in fp64, a wave scan
// plus the totals of the lower waves (through LDS) give its offset, a second pass over the thread's registers applies it.  PER is
// a template parameter so that the element chains (three expf and two IEEE divisions each) are unrolled and overlap.  History: one
// wave per ray scanning 64-interval chunks took 105 us per launch at n = 640 (1024 rays = one wave per SIMD, everything exposed
// latency); lane-blocked with unrolled chains 76 us; four waves per ray ... see DESIGN.md.
constexpr int BOUND_T = 256;
constexpr int BOUND_NB = 1;       // betas evaluated per pass of the bisection
struct BoundScratch { double e[BOUND_NB][BOUND_T / 64], s[BOUND_NB][BOUND_T / 64]; float mx[BOUND_NB][BOUND_T / 64]; };

// NB betas per pass.  Round 6 measured NB = 3 (the midpoint and both candidates for the next one: five passes instead of the bisection's
// ten dependent evaluations, same bits): sampler_bound_kernel 20-26 -> 27-33 us -- the launch is bound by its VALU instructions (16
// waves per compute unit keep the pipes busy), not by the dependent chain, so three times the arithmetic per pass costs more than the
// halved pass count saves.  NB = 1 is what runs.
template <int PER, int NB>
__device__ __forceinline__ void error_bound_t(const float* sdf, const float* dist, const float* dstar, int m, const float (&beta)[NB], int tid,
                                              BoundScratch* sc, float (&out)[NB]) {
  const int lane = tid & 63, wave = tid >> 6;
  const int j0 = tid * PER;
  // The bisection only compares this bound with eps, so its terms use the hardware exp2 / reciprocal forms (each within ~2 ulp of
  // the library functions, 1 or 2 instructions instead of 15 .. 40): the kernel is VALU-bound -- 11 evaluations x n intervals x
  // 1024 rays -- and the exact expf / expm1f / IEEE-division sequences were 5/6 of its instructions.  The pdf the samples are
  // drawn from (resample_cdf_t) keeps the exact functions.
  float e[NB][PER], sv[NB][PER];
  double te[NB], ts[NB], ie[NB], is[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) { te[b] = 0.0; ts[b] = 0.0; }
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const bool ok = j0 + k < m;
    const int j = min(j0 + k, m - 1);
    const float d = dist[j], sd = sdf[j], ds = dstar[j];
    const float sg = (sd > 0.0f) ? 0.5f : ((sd < 0.0f) ? -0.5f : 0.0f);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const float inv_b = 1.0f / beta[b], inv_q = 1.0f / (4.0f * beta[b] * beta[b]);
      e[b][k] = ok ? d * inv_b * (0.5f + sg * (__expf(-fabsf(sd) * inv_b) - 1.0f)) : 0.0f;      // d * laplace_sigma(sd, beta)
      sv[b][k] = ok ? __expf(-ds * inv_b) * (d * d) * inv_q : 0.0f;
      te[b] += (double)e[b][k];
      ts[b] += (double)sv[b][k];
    }
  }
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    ie[b] = wave_incl_scan_d(te[b], lane); is[b] = wave_incl_scan_d(ts[b], lane);
    if (lane == 63) { sc->e[b][wave] = ie[b]; sc->s[b][wave] = is[b]; }
  }
  __syncthreads();
  float best[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    double re = ie[b] - te[b], rs = is[b] - ts[b];                       // exclusive offsets of this thread
#pragma unroll
    for (int w = 0; w < BOUND_T / 64; ++w)
      if (w < wave) { re += sc->e[b][w]; rs += sc->s[b][w]; }
    float bb = -INFINITY;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      rs += (double)sv[b][k];
      const float v = (fminf(__expf((float)rs), 1.0e6f) - 1.0f) * __expf(-(float)re);
      if (j0 + k < m) bb = fmaxf(bb, v);
      re += (double)e[b][k];
    }
    bb = wave_max(bb);
    if (lane == 0) sc->mx[b][wave] = bb;
    best[b] = bb;
  }
  __syncthreads();
#pragma unroll
  for (int b = 0; b < NB; ++b) {
#pragma unroll
    for (int w = 0; w < BOUND_T / 64; ++w) best[b] = fmaxf(best[b], sc->mx[b][w]);
    out[b] = best[b];                                      // (the next evaluation's first barrier orders the reuse of *sc)
  }
}

template <int NB>
__device__ __forceinline__ void error_bound(const float* sdf, const float* dist, const float* dstar, int m, const float (&beta)[NB], int tid,
                                            BoundScratch* sc, float (&out)[NB]) {
  switch (((m + BOUND_T - 1) / BOUND_T) | 1) {      // n = 128 k samples (the reference's grids): PER = 1 (n <= 256) or 3; SMAX = 1024 -> 5
    case 1: error_bound_t<1, NB>(sdf, dist, dstar, m, beta, tid, sc, out); break;
    case 3: error_bound_t<3, NB>(sdf, dist, dstar, m, beta, tid, sc, out); break;
    default: error_bound_t<5, NB>(sdf, dist, dstar, m, beta, tid, sc, out); break;
  }
}
__device__ __forceinline__ float error_bound(const float* sdf, const float* dist, const float* dstar, int m, float beta, int tid, BoundScratch* sc) {
  const float b1[1] = {beta};
  float o1[1];
  error_bound<1>(sdf, dist, dstar, m, b1, tid, sc, o1);
  return o1[0];
}

struct SamplerBoundArgs {
  const float* z; int n, R;                  // [R,n] sorted depths
  const float* sdf_old; const float* sdf_new; const int* order; int n_old;   // merged sdf = gather(cat[old,new], order); order==null: sdf_new is [R,n]
  const float* beta_in; const float* beta0;  // [R], device scalar
  float eps; int iters;
  float* sdf_out; float* beta_out; int* flag;   // merged sdf [R,n], new beta [R], flag |= (beta > beta0)
  const int* gate; int gate_value;              // device-decided rounds (sync-free sampler): run only if *gate == gate_value (null: always)
};

// The ray's merged sdf values (:152-157), interval lengths and d* into LDS; `sdf_out` keeps the merged row for the next round.
__device__ __forceinline__ void bound_load(const float* __restrict__ z, int n, int r, const float* __restrict__ sdf_old,
                                           const float* __restrict__ sdf_new, const int* __restrict__ order, int n_old,
                                           float* __restrict__ sdf_out, float* ssdf, float* sdist, float* sdstar, float* sz, int tid) {
  for (int j = tid; j < n; j += BOUND_T) {
    float v;
    if (order) {
      const int o = order[(size_t)r * n + j];
      v = o < n_old ? sdf_old[(size_t)r * n_old + o] : sdf_new[(size_t)r * (n - n_old) + (o - n_old)];
    } else {
      v = sdf_new[(size_t)r * n + j];
    }
    ssdf[j] = v;
    sdf_out[(size_t)r * n + j] = v;
    if (sz) sz[j] = z[(size_t)r * n + j];
  }
  __syncthreads();
  const int m = n - 1;
  for (int j = tid; j < m; j += BOUND_T) {
    const float d = z[(size_t)r * n + j + 1] - z[(size_t)r * n + j];
    sdist[j] = d;
    sdstar[j] = interval_bound(d, ssdf[j], ssdf[j + 1]);
  }
  __syncthreads();
}

// Per-ray bisection of beta on [beta0, beta_in] (:177-185); every thread of the workgroup holds the same lo / hi.
__device__ __forceinline__ float bisect_beta(const float* ssdf, const float* sdist, const float* sdstar, int m, float beta0, float hi,
                                             float eps, int iters, int tid, BoundScratch* sc) {
  if (error_bound(ssdf, sdist, sdstar, m, beta0, tid, sc) <= eps) hi = beta0;      // (:177-178)
  float lo = beta0;
  // hi == lo (a ray whose bound already holds at beta0): every further midpoint is (beta0 + beta0) / 2 = beta0 exactly and passes the
  // test again -- the reference's ten iterations change nothing, so they are skipped (round 6; same bits)
  if (hi == lo) return hi;
  int it = 0;
  for (; it < iters; ++it) {
    const float mid = (lo + hi) / 2.0f;
    const float err = error_bound(ssdf, sdist, sdstar, m, mid, tid, sc);
    if (err <= eps) hi = mid;
    if (err > eps) lo = mid;
  }
  return hi;
}

__global__ __launch_bounds__(BOUND_T) void sampler_bound_kernel(SamplerBoundArgs a) {
  __shared__ float ssdf[SMAX], sdist[SMAX], sdstar[SMAX];
  __shared__ BoundScratch sc;
  if (a.gate && *a.gate != a.gate_value) return;
  const int r = blockIdx.x, tid = threadIdx.x, n = a.n;
  bound_load(a.z, n, r, a.sdf_old, a.sdf_new, a.order, a.n_old, a.sdf_out, ssdf, sdist, sdstar, nullptr, tid);
  const float beta0 = *a.beta0;
  const float hi = bisect_beta(ssdf, sdist, sdstar, n - 1, beta0, a.beta_in[r], a.eps, a.iters, tid, &sc);
  if (tid == 0) {
    a.beta_out[r] = hi;
    if (hi > beta0) atomicOr(a.flag, 1);
  }
}

struct SamplerResampleArgs {
  const float* z; const float* sdf; int n, R;   // [R,n]
  const float* beta;                            // [R]
  int refine; float add_tiny;                   // refine: error-bound pdf + merge; else: weights pdf (final set)
  const float* u; int u_stride; int N;          // u [N] (stride 0) or [R,N]
  float* samples;                               // [R,N]
  float* z_merged; int* order;                  // refine: [R,n+N] sorted union and its source index into cat[z, samples]
  // ---- device-decided mode (sync-free sampler; all null / 0 in the host-decided mode) -------------------------------------------
  // The batch-global test of ray_sampler.py:200 (`beta.max() > beta0`) is read from *open instead of by the host: refine iff *open and
  // round + 1 < max_rounds.  Both outcomes have their own u / N / output: refine -> (u, N, samples, z_merged, order) above; final ->
  // (u_final, N_final, samples_final) plus a copy of the grid into z_final [R, ld_final] and its size into *n_final.  cont[round] <- 1
  // (refined) or 2 (finalised); the launch runs only if round == 0 or cont[round - 1] == 1.
  const int* open; int* cont; int round, max_rounds;
  const float* u_final; int u_final_stride, N_final; float* samples_final;
  float* z_final; int ld_final; int* n_final;
};

// ---- resampling of one ray by a workgroup of BOUND_T threads (round 6; before: one wavefront per ray, 64-interval chunks) -----------
// Thread t owns the PER consecutive intervals [t PER, (t + 1) PER): fp64 partial sums, a DPP wave scan and the totals of the lower
// waves give its prefix (as error_bound_t).  The reference's cumsums run on torch-CPU in double and round every knot once; so do these,
// in another summation order (exact to ~1e-16 before the one rounding).
struct ResampleScratch { double a[BOUND_T / 64], b[BOUND_T / 64]; };

__device__ __forceinline__ double block_excl_offset(double incl, double own, int lane, int wave, double* tot, double* total) {
  if (lane == 63) tot[wave] = incl;
  __syncthreads();
  double off = incl - own, all = 0.0;
#pragma unroll
  for (int w = 0; w < BOUND_T / 64; ++w) { if (w < wave) off += tot[w]; all += tot[w]; }
  if (total) *total = all;
  return off;
}

// pdf over the n-1 intervals (error-bound opacity :205-211 if refine, rendering weights + 1e-5 :220-222 otherwise) -> normalised ->
// cdf[0..n-1] in scdf (cdf[0] = 0).  sz / ssdf: the ray's grid and sdf values in LDS.
template <int PER>
__device__ __forceinline__ void resample_cdf_t(const float* sz, const float* ssdf, float* scdf, int n, float beta, bool refine,
                                               float add_tiny, int tid, ResampleScratch* sc) {
  const int lane = tid & 63, wave = tid >> 6, m = n - 1, j0 = tid * PER;
  float e[PER], sv[PER], pdf[PER];
  double te = 0.0, ts = 0.0;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int j = j0 + k;
    e[k] = 0.0f; sv[k] = 0.0f;
    if (j < n) {
      const float d = (j < m) ? sz[j + 1] - sz[j] : 1e10f;
      e[k] = d * laplace_sigma(ssdf[j], beta);
      if (refine && j < m) sv[k] = expf(-interval_bound(d, ssdf[j], ssdf[j + 1]) / beta) * (d * d) / (4.0f * beta * beta);
    }
    te += (double)e[k];
    ts += (double)sv[k];
  }
  const double ie = wave_incl_scan_d(te, lane), is = wave_incl_scan_d(ts, lane);
  if (lane == 63) { sc->a[wave] = ie; sc->b[wave] = is; }
  __syncthreads();
  double re = ie - te, rs = is - ts;
#pragma unroll
  for (int w = 0; w < BOUND_T / 64; ++w)
    if (w < wave) { re += sc->a[w]; rs += sc->b[w]; }
  double tp = 0.0;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int j = j0 + k;
    const float T = expf(-(float)re);                  // exclusive prefix: never contains the 1e10 tail interval
    rs += (double)sv[k];
    float pv = 0.0f;
    if (j < m) pv = refine ? (fminf(expf((float)rs), 1.0e6f) - 1.0f) * T + add_tiny : (1.0f - expf(-e[k])) * T + 1e-5f;
    pdf[k] = pv;
    tp += (double)pv;
    re += (double)e[k];
  }
  __syncthreads();                                      // (sc->a / b are reused)
  double sum;
  const double ip = wave_incl_scan_d(tp, lane);
  block_excl_offset(ip, tp, lane, wave, sc->a, &sum);
  // normalise (pdf / sum in fp32, like `pdf / torch.sum(pdf)`), inclusive cumsum in fp64, each knot rounded once
  const float sumf = (float)sum;
  double tq = 0.0;
#pragma unroll
  for (int k = 0; k < PER; ++k) { pdf[k] = (j0 + k < m) ? pdf[k] / sumf : 0.0f; tq += (double)pdf[k]; }
  const double iq = wave_incl_scan_d(tq, lane);
  double run = block_excl_offset(iq, tq, lane, wave, sc->b, nullptr);
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    run += (double)pdf[k];
    if (j0 + k < m) scdf[j0 + k + 1] = (float)run;
  }
  if (tid == 0) scdf[0] = 0.0f;
  __syncthreads();
}

__device__ __forceinline__ void resample_cdf(const float* sz, const float* ssdf, float* scdf, int n, float beta, bool refine, float add_tiny,
                                             int tid, ResampleScratch* sc) {
  switch (((n + BOUND_T - 1) / BOUND_T) | 1) {
    case 1: resample_cdf_t<1>(sz, ssdf, scdf, n, beta, refine, add_tiny, tid, sc); break;
    case 3: resample_cdf_t<3>(sz, ssdf, scdf, n, beta, refine, add_tiny, tid, sc); break;
    default: resample_cdf_t<5>(sz, ssdf, scdf, n, beta, refine, add_tiny, tid, sc); break;
  }
}

// inverse CDF (:237-249): searchsorted(right) + lerp with the 1e-5 denominator guard; N samples at u -> ssmp (LDS) and `samples`
__device__ __forceinline__ void resample_draw(const float* sz, const float* scdf, int n, const float* __restrict__ u, int N,
                                              float* ssmp, float* __restrict__ samples, int tid) {
  for (int k = tid; k < N; k += BOUND_T) {
    const float uv = u[k];
    int lo = 0, hi = n;                          // first index with cdf[idx] > u
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (scdf[mid] > uv) hi = mid; else lo = mid + 1; }
    const int below = max(lo - 1, 0), above = min(lo, n - 1);
    float den = scdf[above] - scdf[below];
    if (den < 1e-5f) den = 1.0f;
    const float t = (uv - scdf[below]) / den;
    const float smp = sz[below] + t * (sz[above] - sz[below]);
    if (ssmp) ssmp[k] = smp;
    samples[k] = smp;
  }
}

// sorted union of z (sorted) and the new samples (sorted: u is increasing): rank by binary search, old first on ties (:254)
__device__ __forceinline__ void resample_merge(const float* sz, const float* ssmp, int n, int N, float* __restrict__ z_merged,
                                               int* __restrict__ order, int tid) {
  for (int j = tid; j < n; j += BOUND_T) {
    const float v = sz[j];
    int lo = 0, hi = N;                          // #samples < v
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (ssmp[mid] < v) lo = mid + 1; else hi = mid; }
    z_merged[j + lo] = v;
    order[j + lo] = j;
  }
  for (int k = tid; k < N; k += BOUND_T) {
    const float v = ssmp[k];
    int lo = 0, hi = n;                          // #z <= v
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (sz[mid] <= v) lo = mid + 1; else hi = mid; }
    z_merged[k + lo] = v;
    order[k + lo] = n + k;
  }
}

__global__ __launch_bounds__(BOUND_T) void sampler_resample_kernel(SamplerResampleArgs a) {
  __shared__ float sz[SMAX], scdf[SMAX], ssmp[SMAX];
  __shared__ float ssdf[SMAX];
  __shared__ ResampleScratch sc;
  const int r = blockIdx.x, tid = threadIdx.x, n = a.n;
  bool refine = a.refine != 0;
  const float* uptr = a.u; int ustride = a.u_stride, N = a.N;
  float* samples = a.samples;
  if (a.cont) {                                   // device-decided round
    if (a.round > 0 && a.cont[a.round - 1] != 1) return;
    refine = (*a.open != 0) && (a.round + 1 < a.max_rounds);
    if (!refine) { uptr = a.u_final; ustride = a.u_final_stride; N = a.N_final; samples = a.samples_final; }
  }
  const float beta = a.beta[r];
  for (int j = tid; j < n; j += BOUND_T) { sz[j] = a.z[(size_t)r * n + j]; ssdf[j] = a.sdf[(size_t)r * n + j]; }
  __syncthreads();
  if (a.cont && !refine) {                        // the grid of the final round is what the tail of Algorithm 1 picks its extra samples from
    for (int j = tid; j < n; j += BOUND_T) a.z_final[(size_t)r * a.ld_final + j] = sz[j];
    if (r == 0 && tid == 0) { *a.n_final = n; }
  }
  if (a.cont && tid == 0) a.cont[a.round] = refine ? 1 : 2;      // (every ray writes the same value; read by later launches only)
  resample_cdf(sz, ssdf, scdf, n, beta, refine, a.add_tiny, tid, &sc);
  resample_draw(sz, scdf, n, uptr + (size_t)r * ustride, N, ssmp, samples + (size_t)r * N, tid);
  if (!refine) return;
  __syncthreads();
  const int tot = n + N;
  resample_merge(sz, ssmp, n, N, a.z_merged + (size_t)r * tot, a.order + (size_t)r * tot, tid);
}

// ---- one round of Algorithm 1 as ONE launch (round 6; device-decided rounds) ------------------------------------------------------
// bound (merge, d*, bisection) -> refine resampling + merge + the NEXT round's query points in the SDF kernels' layout -> the final
// resampling.  The batch-global test of ray_sampler.py:200 cannot be read inside the launch that produces it, so both outcomes are
// prepared: every ray of a round before the last refines (used iff some ray sets open[round]); the final samples are drawn by the
// rays that can still be part of a final round -- all rays in the last round, otherwise those whose own beta has reached beta0 (a
// round is final only if every ray's has).  A launch of round k > 0 runs iff open[k-1] was set; cont[k] <- 1 marks a round that ran.
struct SamplerRoundArgs {
  const float* z; int n, R;                   // [R,n] sorted grid of this round
  const float* sdf_old; const float* sdf_new; const int* order; int n_old;
  const float* beta_in; const float* beta0; float eps; int iters;
  float* sdf_out; float* beta_out;            // merged sdf [R,n], beta [R]
  int* ctl; int round, max_rounds;            // open[max_rounds] | cont[max_rounds] | n_final
  float add_tiny; const float* u_refine; int N_refine;
  float* samples_refine; float* z_merged; int* order_out;      // [R,N_refine], [R,n+N_refine] x2
  const float* origins; const float* dirs; float* x_fm; int ldp;   // next query: x_fm[c][r N_refine + k] = o + smp d (feature-major, stride ldp)
  const float* u_final; int u_final_stride, N_final; float* samples_final; float* z_final; int ld_final;
  int ablate;                                 // probes only (tuning key 30; results WRONG): 1 = no bisection, 2 = no refine path, 4 = no final path
};

__global__ __launch_bounds__(BOUND_T) void sampler_round_kernel(SamplerRoundArgs a) {
  __shared__ float ssdf[SMAX], sdist[SMAX], sdstar[SMAX], sz[SMAX], scdf[SMAX], ssmp[SMAX];
  __shared__ BoundScratch bsc;
  __shared__ ResampleScratch rsc;
  const int K = a.max_rounds;
  if (a.round > 0 && a.ctl[a.round - 1] == 0) return;          // the previous round closed the sampler
  const int r = blockIdx.x, tid = threadIdx.x, n = a.n;
  bound_load(a.z, n, r, a.sdf_old, a.sdf_new, a.order, a.n_old, a.sdf_out, ssdf, sdist, sdstar, sz, tid);
  const float beta0 = *a.beta0;
  const float beta = (a.ablate & 1) ? a.beta_in[r] : bisect_beta(ssdf, sdist, sdstar, n - 1, beta0, a.beta_in[r], a.eps, a.iters, tid, &bsc);
  const bool open = beta > beta0, last = a.round + 1 >= K;
  if (tid == 0) {
    a.beta_out[r] = beta;
    if (open) atomicOr(a.ctl + a.round, 1);
    a.ctl[K + a.round] = 1;
  }
  if (!last && !(a.ablate & 2)) {
    const int N = a.N_refine, tot = n + N;
    resample_cdf(sz, ssdf, scdf, n, beta, true, a.add_tiny, tid, &rsc);
    resample_draw(sz, scdf, n, a.u_refine, N, ssmp, a.samples_refine + (size_t)r * N, tid);
    __syncthreads();
    resample_merge(sz, ssmp, n, N, a.z_merged + (size_t)r * tot, a.order_out + (size_t)r * tot, tid);
    if (a.x_fm) {
      const float o0 = a.origins[3 * r], o1 = a.origins[3 * r + 1], o2 = a.origins[3 * r + 2];
      const float d0 = a.dirs[3 * r], d1 = a.dirs[3 * r + 1], d2 = a.dirs[3 * r + 2];
      for (int k = tid; k < N; k += BOUND_T) {
        const float zz = ssmp[k];
        const size_t p = (size_t)r * N + k;
        a.x_fm[p] = o0 + rounded(zz * d0);                      // `cam_loc + samples * dirs` (:146): product rounded, then the sum
        a.x_fm[(size_t)a.ldp + p] = o1 + rounded(zz * d1);
        a.x_fm[2 * (size_t)a.ldp + p] = o2 + rounded(zz * d2);
      }
      if (r == a.R - 1)                                         // the layout's padding columns
        for (int p = a.R * N + tid; p < a.ldp; p += BOUND_T) { a.x_fm[p] = 0.f; a.x_fm[(size_t)a.ldp + p] = 0.f; a.x_fm[2 * (size_t)a.ldp + p] = 0.f; }
    }
  }
  if ((last || !open) && !(a.ablate & 4)) {
    __syncthreads();
    for (int j = tid; j < n; j += BOUND_T) a.z_final[(size_t)r * a.ld_final + j] = sz[j];
    if (tid == 0) a.ctl[2 * K] = n;                             // (every ray that writes, writes this round's n; the last round to run wins)
    resample_cdf(sz, ssdf, scdf, n, beta, false, a.add_tiny, tid, &rsc);
    resample_draw(sz, scdf, n, a.u_final + (size_t)r * a.u_final_stride, a.N_final, nullptr, a.samples_final + (size_t)r * a.N_final, tid);
  }
}

// What Algorithm 1 computes before its first round (ray_sampler.py:131-143), one wavefront per ray: beta0 = |beta_param| + beta_min
// (density.py:29-30; written once), the ray's initial beta = sqrt(beta_c * sum_i (z[i+1] - z[i])^2) with beta_c = 1 / (4 log(1 + eps))
// (Lemma 2), and the control words of the device-decided rounds zeroed -- nine elementwise / reduction launches otherwise.
// Round 6 adds two optional jobs to the same launch: (i) the FIRST round's query points o + z d in the SDF kernels' layout (x_fm
// [3][ldp], as sampler_round_kernel writes the later rounds'), (ii) in extra workgroups, the training-mode picks of every possible final
// grid size (sampler_pick_block) -- off the critical path instead of one 16 us launch behind the last round.
struct SamplerInitArgs {
  const float* z; int R, n; const float* beta_ptr; float beta_min, beta_c; float* beta0; float* beta_out; int* ctl; int nctl;
  const float* origins; const float* dirs; float* x_fm; int ldp;      // (i): null = not wanted
  const float* keys; int n_step, n_cand, n_extra; int* pick_all;       // (ii): pick_all [n_cand][n_extra] for grids of n_step (c + 1) depths
  int init_blocks, pick_parts;
};

// The n_extra smallest of the first n keys, in key order (equal keys: the lower index first).  One key per thread; a candidate's keys
// are spread over ceil(n / 256) workgroups (`part` = which 256 keys this workgroup ranks): on ONE compute unit the 8 compares per four
// keys x n / 4 broadcast reads x n threads were 16 us (sampler_pick_kernel) resp. 25 us (256 threads, 3 keys each).
__device__ __forceinline__ void sampler_pick_block(const float* __restrict__ keys, int n, int n_extra, int* __restrict__ pick, float* sk, int part) {
  const int tid = threadIdx.x;
  if (part * 256 >= n) return;
  for (int i = tid; i < SMAX; i += 256) sk[i] = i < n ? keys[i] : INFINITY;      // padding never ranks below a key
  __syncthreads();
  const float4* sk4 = (const float4*)sk;
  const int n4 = (n + 3) >> 2, i = part * 256 + tid;
  if (i >= n) return;
  const float x = sk[i];
  int lt = 0, le = 0;
#pragma unroll 4
  for (int j = 0; j < n4; ++j) {                // (one broadcast ds_read_b128 per four keys)
    const float4 k = sk4[j];
    lt += (k.x < x) + (k.y < x) + (k.z < x) + (k.w < x);
    le += (k.x <= x) + (k.y <= x) + (k.z <= x) + (k.w <= x);
  }
  int rank = lt;
  if (le - lt > 1)                               // equal keys (24-bit uniforms: about 1 call in 80 has a pair)
    for (int j = 0; j < i; ++j) rank += sk[j] == x;
  if (rank < n_extra) pick[rank] = i;
}

__global__ __launch_bounds__(256) void sampler_init_kernel(SamplerInitArgs a) {
  __shared__ __attribute__((aligned(16))) float sk[SMAX];
  if ((int)blockIdx.x >= a.init_blocks) {          // pick role: pick_parts workgroups per candidate final grid
    const int b = blockIdx.x - a.init_blocks, c = b / a.pick_parts;
    sampler_pick_block(a.keys, min(a.n_step * (c + 1), SMAX), a.n_extra, a.pick_all + (size_t)c * a.n_extra, sk, b - c * a.pick_parts);
    return;
  }
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (blockIdx.x == 0) {
    for (int i = threadIdx.x; i < a.nctl; i += 256) a.ctl[i] = 0;
    if (threadIdx.x == 0) a.beta0[0] = fabsf(*a.beta_ptr) + a.beta_min;
  }
  if (a.x_fm && blockIdx.x == 0)                   // the layout's padding columns
    for (int p = a.R * a.n + threadIdx.x; p < a.ldp; p += 256) { a.x_fm[p] = 0.f; a.x_fm[(size_t)a.ldp + p] = 0.f; a.x_fm[2 * (size_t)a.ldp + p] = 0.f; }
  if (r >= a.R) return;
  const float* zr = a.z + (size_t)r * a.n;
  float acc = 0.0f;
  for (int j = lane; j + 1 < a.n; j += 64) { const float g = zr[j + 1] - zr[j]; acc += g * g; }
  acc = wave_sum(acc);
  if (lane == 0) a.beta_out[r] = sqrtf(a.beta_c * acc);
  if (a.x_fm) {
    const float o0 = a.origins[3 * r], o1 = a.origins[3 * r + 1], o2 = a.origins[3 * r + 2];
    const float d0 = a.dirs[3 * r], d1 = a.dirs[3 * r + 1], d2 = a.dirs[3 * r + 2];
    for (int j = lane; j < a.n; j += 64) {
      const float zz = zr[j];
      const size_t p = (size_t)r * a.n + j;
      a.x_fm[p] = o0 + rounded(zz * d0);
      a.x_fm[(size_t)a.ldp + p] = o1 + rounded(zz * d1);
      a.x_fm[2 * (size_t)a.ldp + p] = o2 + rounded(zz * d2);
    }
  }
}

// Which samples of the final grid join the output (ray_sampler.py:263-268), decided on the device from the grid size n:
//   eval : torch.linspace(0, n - 1, n_extra).long()            (the reference's formula, fp32, symmetric around the middle)
//   train: a uniformly random n_extra-subset without replacement (the reference: torch.randperm(n)[:n_extra]) = the n_extra smallest
//          of n random keys (drawn on the CPU generator), in key order.  One workgroup.
__global__ __launch_bounds__(1024) void sampler_pick_kernel(const int* __restrict__ n_ptr, const float* __restrict__ keys, int n_extra,
                                                           int* __restrict__ pick) {
  __shared__ __attribute__((aligned(16))) float sk[SMAX];
  const int n = *n_ptr, tid = threadIdx.x;
  if (!keys) {
    const float start = 0.0f, end = (float)(n - 1), step = (end - start) / (float)(n_extra - 1);
    for (int j = tid; j < n_extra; j += 1024) {
      const float v = j < n_extra / 2 ? start + step * (float)j : end - step * (float)(n_extra - 1 - j);
      pick[j] = (int)v;
    }
    return;
  }
  for (int i = tid; i < SMAX; i += 1024) sk[i] = i < n ? keys[i] : INFINITY;      // padding never ranks below a key
  __syncthreads();
  const float4* sk4 = (const float4*)sk;
  const int n4 = (n + 3) >> 2, i = tid;
  if (i >= n) return;
  const float x = sk[i];
  int lt = 0, le = 0;
#pragma unroll 4
  for (int j = 0; j < n4; ++j) {                // (one broadcast ds_read_b128 per four keys)
    const float4 k = sk4[j];
    lt += (k.x < x) + (k.y < x) + (k.z < x) + (k.w < x);
    le += (k.x <= x) + (k.y <= x) + (k.z <= x) + (k.w <= x);
  }
  int rank = lt;
  if (le - lt > 1)                               // equal keys (24-bit uniforms: about 1 call in 80 has a pair): the lower index first
    for (int j = 0; j < i; ++j) rank += sk[j] == x;
  if (rank < n_extra) pick[rank] = i;
}

// a13: sample_pdf (ray_sampler.py:16-59) and the sort of get_z_vals_fine (:97-106), one wavefront per ray.
//   pdf = (weights + 1e-5) / sum;  cdf = [0, cumsum(pdf)] (fp64 accumulation, every knot rounded once: torch-CPU's cumsum);
//   inverse CDF at u with the `denom < 1e-5 -> 1` rule;  then (z_merge given) the sorted union of z_merge and the samples.
struct SamplePdfArgs {
  const float* bins; const float* weights; int nb, R;      // bins [R,nb], weights [R,nb-1]
  const float* u; int u_stride, N;                         // u [N] (stride 0) or [R,N]
  float* samples;                                          // [R,N]
  const float* z_merge; int nz; float* z_out;              // [R,nz] -> z_out [R,nz+N] sorted (null: samples only)
};
__global__ __launch_bounds__(64) void sample_pdf_kernel(SamplePdfArgs a) {
  __shared__ float sb[SMAX], scdf[SMAX], sall[SMAX];
  const int r = blockIdx.x, lane = threadIdx.x, nb = a.nb, m = nb - 1;
  for (int j = lane; j < nb; j += 64) sb[j] = a.bins[(size_t)r * nb + j];
  double sum = 0.0;
  for (int c0 = 0; c0 < m; c0 += 64) {
    const int j = c0 + lane;
    float p = 0.0f;
    if (j < m) { p = a.weights[(size_t)r * m + j] + 1e-5f; scdf[j + 1] = p; }
    sum += wave_sum_d((double)p);
  }
  __syncthreads();
  const float sumf = (float)sum;
  double carry = 0.0;
  for (int c0 = 0; c0 < m; c0 += 64) {
    const int j = c0 + lane;
    const float p = (j < m) ? scdf[j + 1] / sumf : 0.0f;
    const double incl = wave_incl_scan_d((double)p, lane);
    if (j < m) scdf[j + 1] = (float)(carry + incl);
    carry += __shfl(incl, 63);
  }
  if (lane == 0) scdf[0] = 0.0f;
  __syncthreads();
  const int nz = a.z_merge ? a.nz : 0;
  for (int k = lane; k < a.N; k += 64) {
    const float u = a.u[(size_t)r * a.u_stride + k];
    int lo = 0, hi = nb;                          // first index with cdf[idx] > u  (searchsorted, right = True)
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (scdf[mid] > u) hi = mid; else lo = mid + 1; }
    const int below = max(lo - 1, 0), above = min(lo, nb - 1);
    float den = scdf[above] - scdf[below];
    if (den < 1e-5f) den = 1.0f;
    const float t = (u - scdf[below]) / den;
    const float smp = sb[below] + t * (sb[above] - sb[below]);
    a.samples[(size_t)r * a.N + k] = smp;
    sall[nz + k] = smp;
  }
  if (!a.z_merge) return;
  for (int j = lane; j < nz; j += 64) sall[j] = a.z_merge[(size_t)r * nz + j];
  __syncthreads();
  const int tot = nz + a.N;                       // rank of every element of [z_merge | samples] (the samples need not be sorted: eval mode draws u)
  for (int i = lane; i < tot; i += 64) {
    const float x = sall[i];
    int rank = 0;
    for (int j = 0; j < tot; ++j) rank += (sall[j] < x) || (sall[j] == x && j < i);
    a.z_out[(size_t)r * tot + rank] = x;
  }
}

// a2: UniformSampler.get_z_vals (ray_sampler.py:61-95) in one launch.  t = torch.linspace(0, 1, N) is passed in (made once per N by
// the caller), every operation is rounded separately like the reference's chain of elementwise torch ops (no fma contraction):
//   z0 = near (1 - t) + far t;   training: mid = 0.5 (z0[j+1] + z0[j]), upper = [mid | z0[N-1]], lower = [z0[0] | mid],
//   z = lower + (upper - lower) rand.      near / far: per ray (pointer) or one value for all rays.
__global__ void uniform_depths_kernel(const float* __restrict__ near_r, float near_s, const float* __restrict__ far_r, float far_s,
                                      const float* __restrict__ t, const float* __restrict__ rnd, int R, int N, float* __restrict__ z) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R * N) return;
  const int r = i / N, j = i - r * N;
  const float nr = near_r ? near_r[r] : near_s, fr = far_r ? far_r[r] : far_s;
  // (HIP's __fmul_rn / __fadd_rn are plain operators and get contracted; every product goes through rounded())
  auto z0 = [&](int k) { return rounded(nr * rounded(1.0f - t[k])) + rounded(fr * t[k]); };
  float v = rounded(z0(j));
  if (rnd) {
    const float up = j + 1 < N ? rounded(0.5f * rounded(rounded(z0(j + 1)) + v)) : v;
    const float lo = j > 0 ? rounded(0.5f * rounded(v + rounded(z0(j - 1)))) : v;
    v = lo + rounded(rounded(up - lo) * rnd[i]);
  }
  z[i] = v;
}

struct SamplerFinishArgs {
  const float* samples; int N;        // [R,N] final samples
  const float* z; int n;              // [R,n] sampler grid
  const int* pick; int n_extra;       // indices into the grid (shared by all rays), (:263-268)
  float near, far; int R;
  const int* eik_idx;                 // [R] index into the output row (:275-276)
  float* z_vals; float* z_eik;        // [R, N+2+n_extra] sorted, [R]
  int ld_z;                           // row stride of the grid (= n in the host-decided mode)
  // round 6, device-decided rounds: the grid size is *n_final; pick = pick_all[n / n_step - 1] (sampler_init_kernel's pick role), or --
  // pick == null, eval -- torch.linspace(0, n - 1, n_extra).long() evaluated here (the reference's formula, fp32, symmetric around the
  // middle)
  const int* n_final = nullptr; int n_step = 0;
};

__global__ __launch_bounds__(64) void sampler_finish_kernel(SamplerFinishArgs a) {
  __shared__ float v[SMAX];
  const int r = blockIdx.x, lane = threadIdx.x;
  const int M = a.N + 2 + a.n_extra;
  const int* pick = a.pick;
  int n = 0;
  if (a.n_final) {
    n = *a.n_final;
    if (pick) pick += (size_t)(n / a.n_step - 1) * a.n_extra;
  }
  for (int k = lane; k < M; k += 64) {
    float x;
    if (k < a.N) x = a.samples[(size_t)r * a.N + k];
    else if (k == a.N) x = a.near;
    else if (k == a.N + 1) x = a.far;
    else {
      const int j = k - a.N - 2;
      int idx;
      if (pick) idx = pick[j];
      else {
        const float start = 0.0f, end = (float)(n - 1), step = (end - start) / (float)(a.n_extra - 1);
        idx = (int)(j < a.n_extra / 2 ? start + step * (float)j : end - step * (float)(a.n_extra - 1 - j));
      }
      x = a.z[(size_t)r * a.ld_z + idx];
    }
    v[k] = x;
  }
  __syncthreads();
  for (int k = lane; k < M; k += 64) {          // rank sort (M ~ 100): ties broken by position
    const float x = v[k];
    int rank = 0;
    for (int j = 0; j < M; ++j) rank += (v[j] < x) || (v[j] == x && j < k);
    a.z_vals[(size_t)r * M + rank] = x;
    if (rank == a.eik_idx[r]) a.z_eik[r] = x;
  }
}

// ---------------------------------------------------------------------------------------------
// Dataset-side batch assembly ("a GPU path for the __getitem__ sampling", datasets/blender_hawp_dataset.py:186-198,
// scene_hawp_dataset.py:179-190).  The view's images stay on the device; per step the host sends only n draws r_i into the view's
// support pool.  One thread per ray: pixel p = pool[r_i]; uv = (p mod W, p div W); uv_proj = foot point of p; rgb = colour of p;
// label = nearest segment of p; lines2d = that segment (5 floats: x1, y1, x2, y2, score).
// ---------------------------------------------------------------------------------------------
struct GatherBatchArgs {
  const int* pool; const long long* draw; int n, W, npool;
  const float* att; const float* rgb; const int* labels; const float* lines; int nlines;
  float* uv; float* uv_proj; float* rgb_out; float* lines_out; long long* labels_out; long long* pixel_out;
};
__global__ void gather_batch_kernel(GatherBatchArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  long long r = a.draw[i];
  r = r < 0 ? 0 : (r >= a.npool ? a.npool - 1 : r);
  const int p = a.pool[r];
  a.uv[2 * i] = (float)(p % a.W); a.uv[2 * i + 1] = (float)(p / a.W);
  a.uv_proj[2 * i] = a.att[2 * (size_t)p]; a.uv_proj[2 * i + 1] = a.att[2 * (size_t)p + 1];
  for (int c = 0; c < 3; ++c) a.rgb_out[3 * i + c] = a.rgb[3 * (size_t)p + c];
  int lab = a.labels[p];
  lab = lab < 0 ? 0 : (lab >= a.nlines ? a.nlines - 1 : lab);
  for (int c = 0; c < 5; ++c) a.lines_out[5 * i + c] = a.lines[5 * lab + c];
  a.labels_out[i] = lab; a.pixel_out[i] = p;
}

// ---------------------------------------------------------------------------------------------
// Step prefix (neat_copy_batch): up to COPY_BATCH_MAX small copies as one launch; blockIdx.y = the copy, sources in device memory or in
// pinned host memory (read over the bus by the kernel itself).
// ---------------------------------------------------------------------------------------------
constexpr int COPY_BATCH_MAX = 16;
struct CopyBatchArgs { const unsigned* src[COPY_BATCH_MAX]; unsigned* dst[COPY_BATCH_MAX]; long long words[COPY_BATCH_MAX]; int n; };
__global__ __launch_bounds__(256) void copy_batch_kernel(CopyBatchArgs) {
  const CopyBatchArgs* a = (const CopyBatchArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  const int i = blockIdx.y;
  const unsigned* __restrict__ s = a->src[i];
  unsigned* __restrict__ d = a->dst[i];
  const long long w = a->words[i];
  for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < w; k += (long long)gridDim.x * 256) d[k] = s[k];
}

// ---------------------------------------------------------------------------------------------
// Dataset-side attraction field: replacement for the un-vendored `hawp.base._C.encodels`
// (call sites: datasets/blender_hawp_dataset.py:96, scene_hawp_dataset.py:95).  Per pixel: the nearest of the N
// 2-D segments (distance to the segment, projection clamped to its ends); outputs, as the call sites consume them,
//   lmap[0:2] = closest point - pixel, lmap[2:4] = endpoint 1 - pixel, lmap[4:6] = endpoint 2 - pixel  (x, y order)
//   label     = index of that segment (the reference takes argmax over a one-hot [N,H,W] map).
// Ties go to the lower index.  Segments are cached in LDS in chunks; one thread per pixel.
// ---------------------------------------------------------------------------------------------
__global__ void encode_lines_kernel(const float* __restrict__ lines, int N, int H, int W, float* __restrict__ lmap,
                                    int* __restrict__ label, unsigned char* __restrict__ valid) {
  __shared__ float sl[256 * 4];
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const bool ok = idx < H * W;
  const float px = (float)(idx % W), py = (float)(idx / W);
  float best = INFINITY, bx = 0.f, by = 0.f, e1x = 0.f, e1y = 0.f, e2x = 0.f, e2y = 0.f;
  int bi = 0;
  for (int c0 = 0; c0 < N; c0 += 256) {
    const int cn = min(256, N - c0);
    __syncthreads();
    for (int t = threadIdx.x; t < cn * 4; t += blockDim.x) sl[t] = lines[(size_t)c0 * 4 + t];
    __syncthreads();
    if (!ok) continue;
    for (int j = 0; j < cn; ++j) {
      const float x1 = sl[4 * j], y1 = sl[4 * j + 1], x2 = sl[4 * j + 2], y2 = sl[4 * j + 3];
      const float dx = x2 - x1, dy = y2 - y1;
      const float len2 = dx * dx + dy * dy;
      float t = len2 > 0.0f ? ((px - x1) * dx + (py - y1) * dy) / len2 : 0.0f;
      t = fminf(fmaxf(t, 0.0f), 1.0f);
      const float qx = x1 + t * dx, qy = y1 + t * dy;
      const float d2 = (qx - px) * (qx - px) + (qy - py) * (qy - py);
      if (d2 < best) { best = d2; bi = c0 + j; bx = qx - px; by = qy - py; e1x = x1 - px; e1y = y1 - py; e2x = x2 - px; e2y = y2 - py; }
    }
  }
  if (!ok) return;
  const size_t hw = (size_t)H * W;
  lmap[idx] = bx; lmap[hw + idx] = by;
  lmap[2 * hw + idx] = e1x; lmap[3 * hw + idx] = e1y;
  lmap[4 * hw + idx] = e2x; lmap[5 * hw + idx] = e2y;
  label[idx] = bi;
  if (valid) valid[idx] = best < INFINITY ? 1 : 0;      // a nearest segment exists (N > 0 and not all segments non-finite)
}

}  // namespace neat

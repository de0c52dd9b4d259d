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
struct BoundScratch { double e[BOUND_T / 64], s[BOUND_T / 64]; float mx[BOUND_T / 64]; };

template <int PER>
__device__ __forceinline__ float error_bound_t(const float* sdf, const float* dist, const float* dstar, int m, float beta, int tid,
                                               BoundScratch* sc) {
  const int lane = tid & 63, wave = tid >> 6;
  const int j0 = tid * PER;
  // The bisection only compares this bound with eps, so its terms use the hardware exp2 / reciprocal forms (each within ~2 ulp of
  // the library functions, 1 or 2 instructions instead of 15 .. 40): the kernel is VALU-bound -- 11 evaluations x n intervals x
  // 1024 rays -- and the exact expf / expm1f / IEEE-division sequences were 5/6 of its instructions.  The pdf the samples are
  // drawn from (sampler_resample_kernel) keeps the exact functions.
  const float inv_b = 1.0f / beta, inv_q = 1.0f / (4.0f * beta * beta);
  float e[PER], sv[PER];
  double te = 0.0, ts = 0.0;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const bool ok = j0 + k < m;
    const int j = min(j0 + k, m - 1);
    const float d = dist[j], sd = sdf[j];
    const float sg = (sd > 0.0f) ? 0.5f : ((sd < 0.0f) ? -0.5f : 0.0f);
    e[k] = ok ? d * inv_b * (0.5f + sg * (__expf(-fabsf(sd) * inv_b) - 1.0f)) : 0.0f;      // d * laplace_sigma(sd, beta)
    sv[k] = ok ? __expf(-dstar[j] * inv_b) * (d * d) * inv_q : 0.0f;
    te += (double)e[k];
    ts += (double)sv[k];
  }
  const double ie = wave_incl_scan_d(te, lane), is = wave_incl_scan_d(ts, lane);
  if (lane == 63) { sc->e[wave] = ie; sc->s[wave] = is; }
  __syncthreads();
  double re = ie - te, rs = is - ts;                       // exclusive offsets of this thread
#pragma unroll
  for (int w = 0; w < BOUND_T / 64; ++w)
    if (w < wave) { re += sc->e[w]; rs += sc->s[w]; }
  float best = -INFINITY;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    rs += (double)sv[k];
    const float b = (fminf(__expf((float)rs), 1.0e6f) - 1.0f) * __expf(-(float)re);
    if (j0 + k < m) best = fmaxf(best, b);
    re += (double)e[k];
  }
  best = wave_max(best);
  if (lane == 0) sc->mx[wave] = best;
  __syncthreads();
#pragma unroll
  for (int w = 0; w < BOUND_T / 64; ++w) best = fmaxf(best, sc->mx[w]);
  return best;                                             // (the next evaluation's first barrier orders the reuse of *sc)
}

__device__ __forceinline__ float error_bound(const float* sdf, const float* dist, const float* dstar, int m, float beta, int tid, BoundScratch* sc) {
  switch (((m + BOUND_T - 1) / BOUND_T) | 1) {      // n = 128 k samples (the reference's grids): PER = 1 (n <= 256) or 3; SMAX = 1024 -> 5
    case 1: return error_bound_t<1>(sdf, dist, dstar, m, beta, tid, sc);
    case 3: return error_bound_t<3>(sdf, dist, dstar, m, beta, tid, sc);
    default: return error_bound_t<5>(sdf, dist, dstar, m, beta, tid, sc);
  }
}

struct SamplerBoundArgs {
  const float* z; int n, R;                  // [R,n] sorted depths
  const float* sdf_old; const float* sdf_new; const int* order; int n_old;   // merged sdf = gather(cat[old,new], order); order==null: sdf_new is [R,n]
  const float* beta_in; const float* beta0;  // [R], device scalar
  float eps; int iters;
  float* sdf_out; float* beta_out; int* flag;   // merged sdf [R,n], new beta [R], flag |= (beta > beta0)
  const int* gate; int gate_value;              // device-decided rounds (sync-free sampler): run only if *gate == gate_value (null: always)
};

__global__ __launch_bounds__(BOUND_T) void sampler_bound_kernel(SamplerBoundArgs a) {
  __shared__ float ssdf[SMAX], sdist[SMAX], sdstar[SMAX];
  __shared__ BoundScratch sc;
  if (a.gate && *a.gate != a.gate_value) return;
  const int r = blockIdx.x, tid = threadIdx.x, n = a.n;
  const float* z = a.z + (size_t)r * n;
  for (int j = tid; j < n; j += BOUND_T) {
    float v;
    if (a.order) {
      const int o = a.order[(size_t)r * n + j];
      v = o < a.n_old ? a.sdf_old[(size_t)r * a.n_old + o] : a.sdf_new[(size_t)r * (n - a.n_old) + (o - a.n_old)];
    } else {
      v = a.sdf_new[(size_t)r * n + j];
    }
    ssdf[j] = v;
    a.sdf_out[(size_t)r * n + j] = v;
  }
  __syncthreads();
  const int m = n - 1;
  for (int j = tid; j < m; j += BOUND_T) {
    const float d = z[j + 1] - z[j];
    sdist[j] = d;
    sdstar[j] = interval_bound(d, ssdf[j], ssdf[j + 1]);
  }
  __syncthreads();
  const float beta0 = *a.beta0;
  float hi = a.beta_in[r];
  if (error_bound(ssdf, sdist, sdstar, m, beta0, tid, &sc) <= a.eps) hi = beta0;      // (:177-178)
  float lo = beta0;
  for (int it = 0; it < a.iters; ++it) {                                                // bisection (:179-185); every thread holds the same lo / hi
    const float mid = (lo + hi) / 2.0f;
    const float err = error_bound(ssdf, sdist, sdstar, m, mid, tid, &sc);
    if (err <= a.eps) hi = mid;
    if (err > a.eps) lo = mid;
  }
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

__global__ __launch_bounds__(64) void sampler_resample_kernel(SamplerResampleArgs a) {
  __shared__ float sz[SMAX], scdf[SMAX], ssmp[SMAX];
  __shared__ float ssdf[SMAX];
  const int r = blockIdx.x, lane = threadIdx.x, n = a.n, m = n - 1;
  bool refine = a.refine != 0;
  const float* uptr = a.u; int ustride = a.u_stride, N = a.N;
  float* samples = a.samples;
  if (a.cont) {                                   // device-decided round
    if (a.round > 0 && a.cont[a.round - 1] != 1) return;
    refine = (*a.open != 0) && (a.round + 1 < a.max_rounds);
    if (!refine) { uptr = a.u_final; ustride = a.u_final_stride; N = a.N_final; samples = a.samples_final; }
  }
  const float beta = a.beta[r];
  for (int j = lane; j < n; j += 64) { sz[j] = a.z[(size_t)r * n + j]; ssdf[j] = a.sdf[(size_t)r * n + j]; }
  __syncthreads();
  if (a.cont && !refine) {                        // the grid of the final round is what the tail of Algorithm 1 picks its extra samples from
    for (int j = lane; j < n; j += 64) a.z_final[(size_t)r * a.ld_final + j] = sz[j];
    if (r == 0 && lane == 0) { *a.n_final = n; }
  }
  if (a.cont && lane == 0) a.cont[a.round] = refine ? 1 : 2;      // (every ray writes the same value; read by later launches only)
  // pdf over the n-1 intervals -> scdf[1..n-1] (unnormalised), total in `sum`
  double carryE = 0.0, carryS = 0.0, sum = 0.0;
  for (int c0 = 0; c0 < n; c0 += 64) {
    const int j = c0 + lane;
    const bool ok = j < n;
    float e = 0.0f, s = 0.0f;
    if (ok) {
      const float d = (j < m) ? sz[j + 1] - sz[j] : 1e10f;
      e = d * laplace_sigma(ssdf[j], beta);
      if (j < m) s = expf(-interval_bound(d, ssdf[j], ssdf[j + 1]) / beta) * (d * d) / (4.0f * beta * beta);
    }
    const double inclE = wave_incl_scan_d((double)e, lane), inclS = wave_incl_scan_d((double)s, lane);
    double exclE = __shfl_up(inclE, 1);
    if (lane == 0) exclE = 0.0;
    const float T = expf(-(float)(carryE + exclE));
    float pdf = 0.0f;
    if (j < m) {
      if (refine) pdf = (fminf(expf((float)(carryS + inclS)), 1.0e6f) - 1.0f) * T + a.add_tiny;      // (:205-211)
      else pdf = (1.0f - expf(-e)) * T + 1e-5f;                                                  // (:220-222)
      scdf[j + 1] = pdf;
    }
    sum += wave_sum_d((double)pdf);
    carryE += __shfl(inclE, 63);
    carryS += __shfl(inclS, 63);
  }
  __syncthreads();
  // normalise (pdf / sum in fp32, like `pdf / torch.sum(pdf)`), then inclusive cumsum in fp64, each knot rounded once
  // -> cdf[0..n-1] with cdf[0] = 0
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
  // inverse CDF (:237-249): searchsorted(right) + lerp with the 1e-5 denominator guard
  for (int k = lane; k < N; k += 64) {
    const float u = uptr[(size_t)r * ustride + k];
    int lo = 0, hi = n;                          // first index with cdf[idx] > u
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (scdf[mid] > u) hi = mid; else lo = mid + 1; }
    const int below = max(lo - 1, 0), above = min(lo, n - 1);
    float den = scdf[above] - scdf[below];
    if (den < 1e-5f) den = 1.0f;
    const float t = (u - scdf[below]) / den;
    const float smp = sz[below] + t * (sz[above] - sz[below]);
    ssmp[k] = smp;
    samples[(size_t)r * N + k] = smp;
  }
  if (!refine) return;
  __syncthreads();
  // sorted union of z (sorted) and the new samples (sorted: u is increasing): rank by binary search, old first on ties
  const int tot = n + N;
  for (int j = lane; j < n; j += 64) {
    const float v = sz[j];
    int lo = 0, hi = N;                          // #samples < v
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (ssmp[mid] < v) lo = mid + 1; else hi = mid; }
    a.z_merged[(size_t)r * tot + j + lo] = v;
    a.order[(size_t)r * tot + j + lo] = j;
  }
  for (int k = lane; k < N; k += 64) {
    const float v = ssmp[k];
    int lo = 0, hi = n;                          // #z <= v
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (sz[mid] <= v) lo = mid + 1; else hi = mid; }
    a.z_merged[(size_t)r * tot + k + lo] = v;
    a.order[(size_t)r * tot + k + lo] = n + k;
  }
}

// What Algorithm 1 computes before its first round (ray_sampler.py:131-143), one wavefront per ray: beta0 = |beta_param| + beta_min
// (density.py:29-30; written once), the ray's initial beta = sqrt(beta_c * sum_i (z[i+1] - z[i])^2) with beta_c = 1 / (4 log(1 + eps))
// (Lemma 2), and the control words of the device-decided rounds zeroed -- nine elementwise / reduction launches otherwise.
__global__ __launch_bounds__(256) void sampler_init_kernel(const float* __restrict__ z, int R, int n, const float* __restrict__ beta_ptr,
                                                           float beta_min, float beta_c, float* __restrict__ beta0,
                                                           float* __restrict__ beta_out, int* __restrict__ ctl, int nctl) {
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (blockIdx.x == 0) {
    for (int i = threadIdx.x; i < nctl; i += 256) ctl[i] = 0;
    if (threadIdx.x == 0) beta0[0] = fabsf(*beta_ptr) + beta_min;
  }
  if (r >= R) return;
  const float* zr = z + (size_t)r * n;
  float acc = 0.0f;
  for (int j = lane; j + 1 < n; j += 64) { const float g = zr[j + 1] - zr[j]; acc += g * g; }
  acc = wave_sum(acc);
  if (lane == 0) beta_out[r] = sqrtf(beta_c * acc);
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
__device__ __forceinline__ float rounded(float x) { asm volatile("" : "+v"(x)); return x; }      // a product the compiler may not fuse into an fma
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
};

__global__ __launch_bounds__(64) void sampler_finish_kernel(SamplerFinishArgs a) {
  __shared__ float v[SMAX];
  const int r = blockIdx.x, lane = threadIdx.x;
  const int M = a.N + 2 + a.n_extra;
  for (int k = lane; k < M; k += 64) {
    float x;
    if (k < a.N) x = a.samples[(size_t)r * a.N + k];
    else if (k == a.N) x = a.near;
    else if (k == a.N + 1) x = a.far;
    else x = a.z[(size_t)r * a.ld_z + a.pick[k - a.N - 2]];
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

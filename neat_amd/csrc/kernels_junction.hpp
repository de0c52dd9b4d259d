// Device-side rectangular linear-sum assignment: removes the two host round trips per training step the
// reference makes through scipy.optimize.linear_sum_assignment (model/networks/neat_wfr_rend_a.py:473,
// model/networks/loss_wfr.py:108).  scipy (third-party, unpinned in the reference's requirements) implements Crouse's
// shortest-augmenting-path variant of Jonker-Volgenant; this restates that published algorithm for one workgroup,
// keeping its arithmetic (float64 duals over the float32 costs, same evaluation order) and its tie rule (lowest reduced
// cost; on ties an unassigned column wins, among unassigned the last scanned, among assigned the first scanned; the
// scan order is the `remaining` list: initially reversed, removals swap in the last element) so that assignments are
// identical to scipy's, not just equally cheap.
//
// Rows can be masked (row_mask[i] == 0 rows do not take part), which replaces the reference's boolean-mask compaction
// (`cand[cols][good]`, rend_a :478-489) and the host sync that a data-dependent shape costs.  If more rows than
// columns take part the problem is transposed on the device, as scipy does.  Output: pairs sorted by row, padded
// with -1; *n_match = number of pairs (-1: infeasible, i.e. a non-finite cost).
#pragma once
#include <hip/hip_runtime.h>

namespace neat {

constexpr int LSAP_WG = 1024;

struct LsapArgs {
  const float* cost;       // [nr][nc] row-major
  int nr, nc;
  const unsigned char* row_mask;   // [nr] or nullptr
  long long* row_ind;      // [min(nr,nc)]
  long long* col_ind;      // [min(nr,nc)]
  int* n_match;
  double* wsd;             // u[N] v[M] sp[M]
  int* wsi;                // rowlist[nr] path[M] col4row[N] row4col[M] remaining[M] SR[N] SC[M] tmp[M]
  int use_lds, lds_int_off; // work arrays in dynamic LDS instead (doubles first, ints at byte offset lds_int_off)
  const unsigned char* col_mask;   // [nc] or nullptr: columns with 0 do not take part (padded candidate sets)
  int cost_lds_off;                // >= 0: the cost matrix is staged in dynamic LDS at this byte offset (every scan of a row is
                                   // then an LDS read instead of an L2 round trip in a chain of dependent steps), -1: read from memory
};

struct LsapKey { double val; int it; int un; };

__device__ __forceinline__ bool lsap_better(const LsapKey& a, const LsapKey& b) {
  if (a.it < 0) return false;
  if (b.it < 0) return true;
  if (a.val != b.val) return a.val < b.val;
  if (a.un != b.un) return a.un > b.un;
  return a.un ? a.it > b.it : a.it < b.it;
}

__device__ __forceinline__ LsapKey lsap_shfl_xor(const LsapKey& k, int m) {
  LsapKey o;
  o.val = __shfl_xor(k.val, m);
  o.it = __shfl_xor(k.it, m);
  o.un = __shfl_xor(k.un, m);
  return o;
}

// ordered compaction of the indices i in [0,n) with flag(i) != 0 into out[]; returns the count (all threads)
template <class F>
__device__ int lsap_compact(int n, F flag, int* out, int* s_wave, int* s_base) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  if (tid == 0) *s_base = 0;
  __syncthreads();
  for (int b0 = 0; b0 < n; b0 += blockDim.x) {
    const int i = b0 + tid;
    const bool f = i < n && flag(i);
    const unsigned long long bal = __ballot(f);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave[wave] = __popcll(bal);
    __syncthreads();
    int off = *s_base;
    for (int w = 0; w < wave; ++w) off += s_wave[w];
    if (f) out[off + before] = i;
    __syncthreads();
    if (tid == 0) { int t = 0; for (int w = 0; w < nw; ++w) t += s_wave[w]; *s_base += t; }
    __syncthreads();
  }
  const int total = *s_base;
  __syncthreads();          // everyone has read the count before a later call resets it (a fast thread 0 used to race ahead)
  return total;
}

__global__ __launch_bounds__(LSAP_WG) void lsap_kernel(LsapArgs a) {
  // scipy's shortest-augmenting-path algorithm with its tie rules (remaining-list order, "prefer an unassigned column among equal
  // costs"), one workgroup.  Round 5: the phases of a row search were cut from five barriers to two per scan --
  //   * the FIRST scan of a row initialises the row's work arrays (shortest path costs, predecessor, remaining list) as it goes
  //     instead of a pass of its own;
  //   * the dual update touches the rows / columns the search VISITED (a handful: kept as two short lists) and runs, with the
  //     augmentation, inside thread 0's serial section of the scan that found the sink -- no parallel pass, no extra barriers;
  //   * without masks the row / column lists are the identity (no compaction passes).
  // Same arithmetic per element, same results (tests/test_lsap.py: scipy, ties, masks, non-finite costs).
  __shared__ int s_wave[LSAP_WG / 64], s_base;
  __shared__ LsapKey s_key[LSAP_WG / 64];
  __shared__ int s_i, s_nrem, s_fail, s_done, s_nvis;
  __shared__ double s_minval;
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
  const int kmax = min(a.nr, a.nc);
  for (int k = tid; k < kmax; k += nt) { a.row_ind[k] = -1; a.col_ind[k] = -1; }

  // work arrays: in LDS when they fit (every step of the augmenting path is a chain of dependent reads by one thread;
  // at L2 latency that chain is most of the run time for the small problems of a training step), else in `ws`
  extern __shared__ __attribute__((aligned(16))) unsigned char lsap_lds[];
  double* wsd = a.use_lds ? reinterpret_cast<double*>(lsap_lds) : a.wsd;
  int* wsi = a.use_lds ? reinterpret_cast<int*>(lsap_lds + a.lds_int_off) : a.wsi;
  int* rowlist = wsi;
  int* collist = rowlist + a.nr;
  const unsigned char* mask = a.row_mask;
  const unsigned char* cmask = a.col_mask;
  int n_eff, c_eff;
  if (!mask && !cmask) {
    for (int i = tid; i < a.nr; i += nt) rowlist[i] = i;
    for (int j = tid; j < a.nc; j += nt) collist[j] = j;
    n_eff = a.nr; c_eff = a.nc;
  } else {
    n_eff = lsap_compact(a.nr, [&](int i) { return mask ? mask[i] != 0 : true; }, rowlist, s_wave, &s_base);
    c_eff = lsap_compact(a.nc, [&](int j) { return cmask ? cmask[j] != 0 : true; }, collist, s_wave, &s_base);
  }
  const bool T = c_eff < n_eff;
  const int N = T ? c_eff : n_eff, M = T ? n_eff : c_eff;
  if (N == 0) { if (tid == 0) *a.n_match = 0; return; }
  double* u = wsd; double* v = u + N; double* sp = v + M;
  int* path = collist + a.nc; int* col4row = path + M; int* row4col = col4row + N; int* remaining = row4col + M;
  int* vis_row = remaining + M;      // (the SR region: rows scanned by the current search, in order; at most N)
  int* vis_col = vis_row + N;        // (the SC region: columns taken out of the remaining list by it; at most N)
  int* tmp = vis_col + M;
  const float* C = a.cost;
  const int nc0 = a.nc;
  const bool staged = a.use_lds && a.cost_lds_off >= 0;
  const float* Cl = reinterpret_cast<const float*>(lsap_lds + (staged ? a.cost_lds_off : 0));
  if (staged) {
    // the cost matrix into LDS: four independent 16-byte loads in flight per thread (a plain element loop is a chain of dependent
    // L2 round trips: 16 of them for the 8 x 2048 matrix of a training step)
    float* dst = reinterpret_cast<float*>(lsap_lds + a.cost_lds_off);
    const int total = a.nr * a.nc;
    const int total4 = ((reinterpret_cast<size_t>(C) & 15) == 0) ? (total >> 2) : 0;
    const float4* C4 = reinterpret_cast<const float4*>(C);
    float4* d4 = reinterpret_cast<float4*>(dst);
    for (int e = tid; e < total4; e += 4 * nt) {
      float4 x[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) if (e + q * nt < total4) x[q] = C4[e + q * nt];
#pragma unroll
      for (int q = 0; q < 4; ++q) if (e + q * nt < total4) d4[e + q * nt] = x[q];
    }
    for (int e = 4 * total4 + tid; e < total; e += nt) dst[e] = C[e];
  }
  const bool ident = !mask && !cmask;          // identity row / column lists: no index indirection in the scans
  auto cost = [&](int i, int j) -> double {
    const int idx = ident ? (T ? j * nc0 + i : i * nc0 + j) : (T ? rowlist[j] * nc0 + collist[i] : rowlist[i] * nc0 + collist[j]);
    return (double)(staged ? Cl[idx] : C[idx]);
  };
  for (int i = tid; i < N; i += nt) { u[i] = 0.0; col4row[i] = -1; }
  for (int j = tid; j < M; j += nt) { v[j] = 0.0; row4col[j] = -1; }
  if (tid == 0) { s_fail = 0; s_done = 0; s_i = 0; s_nrem = M; s_minval = 0.0; s_nvis = 0; }
  __syncthreads();

  for (int cur = 0; cur < N; ++cur) {
    bool first = true;
    while (true) {
      const int i = s_i, nrem = s_nrem;
      const double minval = s_minval, ui = u[i];
      LsapKey best{INFINITY, -1, 0};
      if (first) {
        // first scan of the row (i = cur, minval = 0, every column remaining in scipy's initial order M-1 .. 0): the pass that set
        // sp = inf / remaining / the visited flags is folded in
        for (int it = tid; it < M; it += nt) {
          const int j = M - 1 - it;
          const double r = minval + cost(i, j) - ui - v[j];
          const bool lower = r < INFINITY;
          const double spj = lower ? r : INFINITY;
          sp[j] = spj;
          if (lower) path[j] = i;
          remaining[it] = j;
          LsapKey k{spj, it, row4col[j] == -1 ? 1 : 0};
          if (spj < INFINITY && lsap_better(k, best)) best = k;
        }
        first = false;
      } else {
        for (int it = tid; it < nrem; it += nt) {
          const int j = remaining[it];
          const double r = minval + cost(i, j) - ui - v[j];
          double spj = sp[j];
          if (r < spj) { path[j] = i; sp[j] = r; spj = r; }
          LsapKey k{spj, it, row4col[j] == -1 ? 1 : 0};
          if (spj < INFINITY && lsap_better(k, best)) best = k;
        }
      }
      for (int m = 32; m >= 1; m >>= 1) {
        LsapKey o = lsap_shfl_xor(best, m);
        if (lsap_better(o, best)) best = o;
      }
      if (lane == 0) s_key[wave] = best;
      __syncthreads();
      if (tid == 0) {
        LsapKey b = s_key[0];
        for (int w = 1; w < nw; ++w) if (lsap_better(s_key[w], b)) b = s_key[w];
        int nvis = s_nvis;
        vis_row[nvis] = i;
        if (b.it < 0) { s_fail = 1; s_done = N; }
        else {
          const double mv = b.val;
          const int j = remaining[b.it];
          vis_col[nvis] = j;
          ++nvis;
          remaining[b.it] = remaining[nrem - 1];
          if (row4col[j] != -1) { s_i = row4col[j]; s_minval = mv; s_nrem = nrem - 1; s_nvis = nvis; }
          else {
            // sink found: dual update over the visited rows / columns (scipy: u[cur] += minVal; u[i] += minVal - sp[col4row[i]] for the
            // other scanned rows; v[j] -= minVal - sp[j] for the scanned columns), then the augmentation along the path
            for (int q = 0; q < nvis; ++q) {
              const int r_ = vis_row[q];
              if (r_ == cur) u[r_] += mv;
              else u[r_] += mv - sp[col4row[r_]];
            }
            for (int q = 0; q < nvis; ++q) { const int c_ = vis_col[q]; v[c_] -= mv - sp[c_]; }
            int jj = j;
            while (true) {
              const int ii = path[jj];
              row4col[jj] = ii;
              const int t = col4row[ii]; col4row[ii] = jj; jj = t;
              if (ii == cur) break;
            }
            s_done = cur + 1; s_i = cur + 1; s_nrem = M; s_minval = 0.0; s_nvis = 0;      // the next row's search starts clean
          }
        }
      }
      __syncthreads();
      if (s_done > cur) break;
    }
    if (s_fail) break;
  }
  if (s_fail) { if (tid == 0) *a.n_match = -1; return; }
  if (!T) {
    for (int i = tid; i < N; i += nt) { a.row_ind[i] = rowlist[i]; a.col_ind[i] = collist[col4row[i]]; }
  } else {      // problem rows are the original columns: emit pairs sorted by original row
    const int cnt = lsap_compact(M, [&](int k) { return row4col[k] != -1; }, tmp, s_wave, &s_base);
    for (int q = tid; q < cnt; q += nt) { const int k = tmp[q]; a.row_ind[q] = rowlist[k]; a.col_ind[q] = collist[row4col[k]]; }
  }
  if (tid == 0) *a.n_match = N;
}

// ---------------------------------------------------------------------------------------------
// Adam over ONE flat parameter buffer (a16: torch.optim.Adam as the reference trainer uses it, volsdf_train.py:177,
// defaults amsgrad=False, weight_decay=0, maximize=False) -- one launch for the 1 219 274 parameters instead of torch's
// multi-tensor kernels over 65 tensors.  Arithmetic in torch's order:
//   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g g ; p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
// ---------------------------------------------------------------------------------------------
#ifndef NEAT_ADAM_PASSES
#define NEAT_ADAM_PASSES 1      // float4 passes per thread: 1 = four times the workgroups, one short chain each (14.8 -> 10.4 us at the 1.2 M parameters; 2: 12.6, 4: 14.8)
#endif
constexpr int ADAM_MAXSEG = 96, ADAM_PASSES = NEAT_ADAM_PASSES;
struct AdamSegs {                 // gradient of segment s covers [off[s], off[s+1]); per-segment bias corrections (torch counts steps per tensor)
  const float* g[ADAM_MAXSEG]; long long off[ADAM_MAXSEG + 1]; float lr_over_bc1[ADAM_MAXSEG], inv_sqrt_bc2[ADAM_MAXSEG]; int nseg;
};

constexpr int ADAM_SEGS_KERNARG_OFFSET = 8;     // adam_flat_kernel(float* p, AdamSegs segs, ...): segs follows the first pointer
// coef (round 6; null: the kernel-argument table's values): device array [2 nseg] = (lr / bias_correction1, 1 / sqrt(bias_correction2))
// per segment -- the step-dependent numbers of a launch that is CAPTURED in the step's HIP graph; the host refreshes the array
// before every replay (same double-precision formulas, neat_amd/optim.py)
__global__ void adam_flat_kernel(float* __restrict__ p, AdamSegs segs, float* __restrict__ m, float* __restrict__ v,
                                 long long n, float beta1, float beta2, float eps, const float* __restrict__ coef) {
  __shared__ long long s_off[ADAM_MAXSEG + 1];
  __shared__ const float* s_g[ADAM_MAXSEG];
  __shared__ float s_a1[ADAM_MAXSEG], s_a2[ADAM_MAXSEG];
  // the segment table is read from the kernel-argument segment through a pointer (plain indexed loads): indexing the
  // by-value struct with a runtime index would copy it to scratch, and a compile-time select chain costs ~700 instructions
  typedef const __attribute__((address_space(4))) char* karg_ptr;
  typedef const __attribute__((address_space(4))) AdamSegs* karg_segs;
  const karg_segs tab = (karg_segs)((karg_ptr)__builtin_amdgcn_kernarg_segment_ptr() + ADAM_SEGS_KERNARG_OFFSET);
  for (int t = threadIdx.x; t <= segs.nseg; t += blockDim.x) {
    s_off[t] = tab->off[t];
    if (t < segs.nseg) { s_g[t] = tab->g[t]; s_a1[t] = coef ? coef[2 * t] : tab->lr_over_bc1[t]; s_a2[t] = coef ? coef[2 * t + 1] : tab->inv_sqrt_bc2[t]; }
  }
  __syncthreads();
  // ADAM_PASSES coalesced float4 passes per thread (the segment table above is built once per 4 * ADAM_PASSES * blockDim
  // elements); per pass: one segment search for four consecutive elements, then the segment only moves forward
  for (int pass = 0; pass < ADAM_PASSES; ++pass) {
    const long long i0 = (((long long)blockIdx.x * ADAM_PASSES + pass) * blockDim.x + threadIdx.x) * 4;
    if (i0 >= n) return;
    int lo = 0, hi = segs.nseg - 1;              // segment of element i0: last s with off[s] <= i0
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_off[mid] <= i0) lo = mid; else hi = mid - 1; }
    const long long iend = i0 + 4 < n ? i0 + 4 : n;
    if (iend <= s_off[lo + 1] && iend == i0 + 4) {                 // the usual case: all four in one segment
      const float* gp = s_g[lo];
      if (!gp) continue;                         // parameter without a gradient this step: untouched, as torch.optim.Adam
      const float a1 = s_a1[lo], a2 = s_a2[lo];
      const float* gq = gp + (i0 - s_off[lo]);
      float4 mm = *reinterpret_cast<const float4*>(m + i0), vv = *reinterpret_cast<const float4*>(v + i0), pp = *reinterpret_cast<const float4*>(p + i0);
      float* me = reinterpret_cast<float*>(&mm); float* ve = reinterpret_cast<float*>(&vv); float* pe = reinterpret_cast<float*>(&pp);
      float ge[4];
      if ((reinterpret_cast<size_t>(gq) & 15) == 0) {
        const float4 g4 = *reinterpret_cast<const float4*>(gq);
        ge[0] = g4.x; ge[1] = g4.y; ge[2] = g4.z; ge[3] = g4.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) ge[e] = gq[e];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float g = ge[e];
        me[e] = beta1 * me[e] + (1.0f - beta1) * g;
        ve[e] = beta2 * ve[e] + (1.0f - beta2) * g * g;
        pe[e] -= a1 * (me[e] / (sqrtf(ve[e]) * a2 + eps));
      }
      *reinterpret_cast<float4*>(m + i0) = mm; *reinterpret_cast<float4*>(v + i0) = vv; *reinterpret_cast<float4*>(p + i0) = pp;
      continue;
    }
    for (long long i = i0; i < iend; ++i) {
      while (i >= s_off[lo + 1]) ++lo;
      const float* gp = s_g[lo];
      if (!gp) continue;
      const float g = gp[i - s_off[lo]];
      const float mm = beta1 * m[i] + (1.0f - beta1) * g;
      const float vv = beta2 * v[i] + (1.0f - beta2) * g * g;
      m[i] = mm; v[i] = vv;
      p[i] -= s_a1[lo] * (mm / (sqrtf(vv) * s_a2[lo] + eps));
    }
  }
}

// ---------------------------------------------------------------------------------------------
// R-sized glue of the junction block and the loss as single launches (the torch formulations cost ~16 and ~30 tiny kernels
// each and the step has five projections and two line losses).
// project2D (model/networks/neat_wfr_rend_a.py:317-326):  cam = K (R x + T);  w = cam_z (+-1e-8 if |cam_z| < 1e-8);
// uv = cam_xy / w.  K: [3,3] row-major, w2c: [3,4] row-major = [R | T].
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void project_point(const float* K, const float* M, const float x[3], float cam[3], float& w) {
  float c[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) c[i] = M[4 * i] * x[0] + M[4 * i + 1] * x[1] + M[4 * i + 2] * x[2] + M[4 * i + 3];
#pragma unroll
  for (int i = 0; i < 3; ++i) cam[i] = K[3 * i] * c[0] + K[3 * i + 1] * c[1] + K[3 * i + 2] * c[2];
  w = cam[2];
  if (fabsf(w) < 1e-8f) w += (w >= 0.0f) ? 1e-8f : -1e-8f;
}

__global__ void project2d_kernel(const float* __restrict__ K, const float* __restrict__ w2c, const float* __restrict__ X, int N,
                                 float* __restrict__ uv) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const float x[3] = {X[3 * i], X[3 * i + 1], X[3 * i + 2]};
  float cam[3], w;
  project_point(K, w2c, x, cam, w);
  uv[2 * i] = cam[0] / w; uv[2 * i + 1] = cam[1] / w;
}

// The same point set through two intrinsics in one launch (round 6): the junction block projects the line end points, the junction
// candidates and the global junctions once with K (pixels) and once with the identity (calibrated coordinates), rend_a :436-441,
// :469-471, :493-496.  Same arithmetic as two project2d_kernel launches.
__global__ void project2d_pair_kernel(const float* __restrict__ K, const float* __restrict__ K2, const float* __restrict__ w2c,
                                      const float* __restrict__ X, int N, float* __restrict__ uv, float* __restrict__ uv2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const float x[3] = {X[3 * i], X[3 * i + 1], X[3 * i + 2]};
  float cam[3], w;
  project_point(K, w2c, x, cam, w);
  uv[2 * i] = cam[0] / w; uv[2 * i + 1] = cam[1] / w;
  project_point(K2, w2c, x, cam, w);
  uv2[2 * i] = cam[0] / w; uv2[2 * i + 1] = cam[1] / w;
}

// d_X = R^T K^T d_cam with d_cam = (d_u / w, d_v / w, -(d_u u + d_v v) / w)
__device__ __forceinline__ void project_bwd_point(const float* K, const float* w2c, const float x[3], float du, float dv, float dX[3]) {
  float cam[3], w;
  project_point(K, w2c, x, cam, w);
  const float dcam[3] = {du / w, dv / w, -(du * (cam[0] / w) + dv * (cam[1] / w)) / w};
  float dc[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) dc[j] = K[j] * dcam[0] + K[3 + j] * dcam[1] + K[6 + j] * dcam[2];
#pragma unroll
  for (int j = 0; j < 3; ++j) dX[j] = w2c[j] * dc[0] + w2c[4 + j] * dc[1] + w2c[8 + j] * dc[2];
}
__global__ void project2d_bwd_kernel(const float* __restrict__ K, const float* __restrict__ w2c, const float* __restrict__ X, int N,
                                     const float* __restrict__ d_uv, float* __restrict__ d_X) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const float x[3] = {X[3 * i], X[3 * i + 1], X[3 * i + 2]};
  float dX[3];
  project_bwd_point(K, w2c, x, d_uv[2 * i], d_uv[2 * i + 1], dX);
#pragma unroll
  for (int j = 0; j < 3; ++j) d_X[3 * i + j] = dX[j];
}
// The identity intrinsics of the calibrated projections (rend_a :441, :496), for the backward passes folded into the loss kernels: the
// products with its 1.0 / 0.0 entries are the ones project2d_bwd_kernel forms with the identity TENSOR, so the results are the same bits.
__device__ __forceinline__ void project_bwd_calib(const float* w2c, const float x[3], float du, float dv, float dX[3]) {
  float I3[9] = {1.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0.0f, 1.0f};
#pragma unroll
  for (int k = 0; k < 9; ++k) asm volatile("" : "+v"(I3[k]));      // opaque: no algebraic shortcut (0 * inf, -0) the tensor form would not take
  project_bwd_point(I3, w2c, x, du, dv, dX);
}

// Endpoint-order-invariant L1 between 2-D segments, gated (model/networks/loss_wfr.py:34-45), one workgroup:
//   target = gt or gt with its endpoints swapped, whichever is closer in L2;  per_line = mean_c |pred - target|
//   loss = sum(per_line w [per_line < thr]) / max(#[per_line < thr], 1)
// out[0] = loss, out[1] = count; per_line [R]; d_pred [R,4] = d loss / d pred.
__global__ __launch_bounds__(1024) void line_loss_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                         const float* __restrict__ weight, int R, float thr, float* __restrict__ out,
                                                         float* __restrict__ per_line, float* __restrict__ d_pred) {
  __shared__ float s_sum[16], s_cnt[16];
  __shared__ float s_inv;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float sum = 0.0f, cnt = 0.0f;
  for (int r = tid; r < R; r += blockDim.x) {
    float p[4], g[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { p[c] = pred[4 * r + c]; g[c] = gt[4 * r + c]; }
    float ds = 0.0f, df = 0.0f;
#pragma unroll
    for (int c = 0; c < 4; ++c) { const float a = p[c] - g[c], b = p[c] - g[c ^ 2]; ds += a * a; df += b * b; }
    const bool straight = ds < df;
    float l = 0.0f;
#pragma unroll
    for (int c = 0; c < 4; ++c) l += fabsf(p[c] - (straight ? g[c] : g[c ^ 2]));
    l *= 0.25f;
    per_line[r] = l;
    if (l < thr) { sum += l * weight[r]; cnt += 1.0f; }
  }
  sum = wave_sum(sum); cnt = wave_sum(cnt);
  if (lane == 0) { s_sum[wave] = sum; s_cnt[wave] = cnt; }
  __syncthreads();
  if (tid == 0) {
    float a = 0.0f, b = 0.0f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { a += s_sum[w]; b += s_cnt[w]; }
    const float den = fmaxf(b, 1.0f);
    out[0] = a / den; out[1] = b;
    s_inv = 1.0f / den;
  }
  __syncthreads();
  const float inv = s_inv;
  for (int r = tid; r < R; r += blockDim.x) {
    float p[4], g[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { p[c] = pred[4 * r + c]; g[c] = gt[4 * r + c]; }
    float ds = 0.0f, df = 0.0f;
#pragma unroll
    for (int c = 0; c < 4; ++c) { const float a = p[c] - g[c], b = p[c] - g[c ^ 2]; ds += a * a; df += b * b; }
    const bool straight = ds < df;
    const float coef = (per_line[r] < thr) ? weight[r] * inv * 0.25f : 0.0f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float d = p[c] - (straight ? g[c] : g[c ^ 2]);
      d_pred[4 * r + c] = coef * (d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f));
    }
  }
}

// inverse of one small (n <= 4) matrix by Gauss-Jordan elimination with partial pivoting, one thread: the pose and
// intrinsics inverses of the junction block / loss (rend_a :440, loss_wfr.py:59) cost a dozen rocSOLVER launches each
template <int NN>
__device__ __forceinline__ void inv_small_body(const float* sA, float* __restrict__ out) {
  float a[NN][2 * NN];                       // every index below is a compile-time constant: the matrix lives in registers
#pragma unroll
  for (int i = 0; i < NN; ++i)
#pragma unroll
    for (int j = 0; j < NN; ++j) { a[i][j] = sA[i * NN + j]; a[i][NN + j] = (i == j) ? 1.0f : 0.0f; }
#pragma unroll
  for (int c = 0; c < NN; ++c) {
    // partial pivoting as compare-and-swap against the rows below: row c ends up with the largest |a[.][c]| (the rows
    // below may end in another order than with one swap, which only permutes later pivot candidates)
#pragma unroll
    for (int r = c + 1; r < NN; ++r) {
      const bool sw = fabsf(a[r][c]) > fabsf(a[c][c]);
#pragma unroll
      for (int j = 0; j < 2 * NN; ++j) { const float x = a[c][j], y = a[r][j]; a[c][j] = sw ? y : x; a[r][j] = sw ? x : y; }
    }
    const float inv = 1.0f / a[c][c];
#pragma unroll
    for (int j = 0; j < 2 * NN; ++j) a[c][j] *= inv;
#pragma unroll
    for (int r = 0; r < NN; ++r) {
      if (r == c) continue;
      const float f = a[r][c];
#pragma unroll
      for (int j = 0; j < 2 * NN; ++j) a[r][j] -= f * a[c][j];
    }
  }
#pragma unroll
  for (int i = 0; i < NN; ++i)
#pragma unroll
    for (int j = 0; j < NN; ++j) out[i * NN + j] = a[i][NN + j];
}

// Both line losses of VolSDFLoss.forward in one launch (loss_wfr.py:52-65): the gated pixel-space term on the (detached) 2-D lines,
// the count of segments it accepts, the K^-1 calibration of the ground-truth end points (:59-63) and the differentiable calibrated
// term whose weights are masked by the first term's gate.  gt5 [R,5] = (x1, y1, x2, y2, weight) as the dataset delivers it.
//   out[0] = l2d (pixel term), out[1] = line loss (calibrated term), out[2] = #{per_line_px < thr};  d_pred_c [R,4] = grad_scale * d out[1] / d pred_c
// (grad_scale = the line term's weight in the total loss: the backward pass then has nothing left to multiply).
// Same arithmetic and summation order as line_loss_kernel / inv_small_kernel / project2d_kernel, which it replaces on this path
// (eleven launches: two slices, compare, ones, cat, inverse, projection, mask product, two line losses, sum).
// w2c / X3 / d_X3 (round 6; null: not wanted): the calibrated predictions are project2d(I, w2c, X3) of the 3-D line end points X3 [R,2,3]
// (rend_a :441); their gradient is carried on to d_X3 = d loss / d X3 here (the arithmetic of project2d_bwd_kernel) -- one launch less.
__device__ __forceinline__ void line_losses_body(const float* __restrict__ pred_u, const float* __restrict__ pred_c,
                                                 const float* __restrict__ gt5, const float* __restrict__ K, int R, float thr,
                                                 float* __restrict__ out, float* __restrict__ d_pred_c, float grad_scale,
                                                 const float* __restrict__ w2c = nullptr, const float* __restrict__ X3 = nullptr,
                                                 float* __restrict__ d_X3 = nullptr) {
  __shared__ float s_acc[4][16];
  __shared__ float s_kinv[9], s_k[9];
  __shared__ float s_inv;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < 9) s_k[tid] = K[tid];
  __syncthreads();
  if (tid == 0) inv_small_body<3>(s_k, s_kinv);
  __syncthreads();
  auto terms = [&](int r, float& lu, float& lc, float& w, float (&dsign)[4]) {
    float pu[4], pc[4], g[4], gc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { pu[c] = pred_u[4 * r + c]; pc[c] = pred_c[4 * r + c]; g[c] = gt5[5 * r + c]; }
    w = gt5[5 * r + 4];
#pragma unroll
    for (int e = 0; e < 2; ++e) {            // K^-1 (x, y, 1), divided by its third component (project2d with w2c = [I | 0])
      const float x[3] = {g[2 * e], g[2 * e + 1], 1.0f};
      float cam[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) cam[i] = s_kinv[3 * i] * x[0] + s_kinv[3 * i + 1] * x[1] + s_kinv[3 * i + 2] * x[2];
      float ww = cam[2];
      if (fabsf(ww) < 1e-8f) ww += (ww >= 0.0f) ? 1e-8f : -1e-8f;
      gc[2 * e] = cam[0] / ww; gc[2 * e + 1] = cam[1] / ww;
    }
    auto one = [&](const float (&p)[4], const float (&t)[4], float* sg) {
      float ds = 0.0f, df = 0.0f;
#pragma unroll
      for (int c = 0; c < 4; ++c) { const float a = p[c] - t[c], b = p[c] - t[c ^ 2]; ds += a * a; df += b * b; }
      const bool straight = ds < df;
      float l = 0.0f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float d = p[c] - (straight ? t[c] : t[c ^ 2]);
        l += fabsf(d);
        if (sg) sg[c] = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
      }
      return l * 0.25f;
    };
    lu = one(pu, g, nullptr);
    lc = one(pc, gc, dsign);
  };
  float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};      // sum / count of the pixel term, sum / count of the calibrated term
  for (int r = tid; r < R; r += blockDim.x) {
    float lu, lc, w, sg[4];
    terms(r, lu, lc, w, sg);
    const bool close = lu < thr;
    if (close) { acc[0] += lu * w; acc[1] += 1.0f; }
    if (lc < thr) { acc[2] += lc * (w * (close ? 1.0f : 0.0f)); acc[3] += 1.0f; }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    acc[k] = wave_sum(acc[k]);
    if (lane == 0) s_acc[k][wave] = acc[k];
  }
  __syncthreads();
  if (tid == 0) {
    float t[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w)
#pragma unroll
      for (int k = 0; k < 4; ++k) t[k] += s_acc[k][w];
    out[0] = t[0] / fmaxf(t[1], 1.0f); out[2] = t[1];
    const float den = fmaxf(t[3], 1.0f);
    out[1] = t[2] / den;
    s_inv = 1.0f / den;
  }
  __syncthreads();
  const float inv = s_inv;
  for (int r = tid; r < R; r += blockDim.x) {
    float lu, lc, w, sg[4];
    terms(r, lu, lc, w, sg);
    const float coef = (lc < thr) ? (w * (lu < thr ? 1.0f : 0.0f)) * inv * 0.25f : 0.0f;
    float dp[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { dp[c] = (coef * sg[c]) * grad_scale; d_pred_c[4 * r + c] = dp[c]; }
    if (d_X3) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const float x[3] = {X3[6 * r + 3 * e], X3[6 * r + 3 * e + 1], X3[6 * r + 3 * e + 2]};
        float dX[3];
        project_bwd_calib(w2c, x, dp[2 * e], dp[2 * e + 1], dX);
#pragma unroll
        for (int j = 0; j < 3; ++j) d_X3[6 * r + 3 * e + j] = dX[j];
      }
    }
  }
}

__global__ __launch_bounds__(1024) void line_losses_kernel(const float* __restrict__ pred_u, const float* __restrict__ pred_c,
                                                           const float* __restrict__ gt5, const float* __restrict__ K, int R, float thr,
                                                           float* __restrict__ out, float* __restrict__ d_pred_c, float grad_scale) {
  line_losses_body(pred_u, pred_c, gt5, K, R, thr, out, d_pred_c, grad_scale);
}

__global__ void inv_small_kernel(const float* __restrict__ A, int n, int lda, float* __restrict__ out) {
  __shared__ float sA[16];
  if (threadIdx.x < n * n) sA[threadIdx.x] = A[(threadIdx.x / n) * lda + threadIdx.x % n];      // one parallel fetch, not 16 dependent ones
  __syncthreads();
  if (threadIdx.x != 0) return;
  if (n == 4) inv_small_body<4>(sA, out);
  else if (n == 3) inv_small_body<3>(sA, out);
  else if (n == 2) inv_small_body<2>(sA, out);
  else out[0] = 1.0f / sA[0];
}

// world-to-camera [R | T] = the first three rows of pose^-1, and the contiguous 3x3 block of the intrinsics: what the junction block's
// projections read (rend_a :424-431).  One launch instead of the inverse plus a strided copy.
__global__ void camera_mats_kernel(const float* __restrict__ pose, const float* __restrict__ K, int kstride, float* __restrict__ w2c,
                                   float* __restrict__ K3) {
  __shared__ float sA[16], sInv[16];
  if (threadIdx.x < 16) sA[threadIdx.x] = pose[threadIdx.x];
  if (threadIdx.x < 9) K3[threadIdx.x] = K[(threadIdx.x / 3) * kstride + threadIdx.x % 3];
  __syncthreads();
  if (threadIdx.x == 0) inv_small_body<4>(sA, sInv);
  __syncthreads();
  if (threadIdx.x < 12) w2c[threadIdx.x] = sInv[threadIdx.x];
}

// Everything a forward derives from the camera alone, in one launch (round 6; before: camera_rays_kernel twice and camera_mats_kernel):
// the rays through `uv` with their origins (rend_a :382-396), the rays through `uv_proj` (:444), [R | T] of pose^-1 and the contiguous
// intrinsics (:424-431).  Same arithmetic as the launches it replaces.
__global__ __launch_bounds__(256) void camera_setup_kernel(const float* __restrict__ uv, const float* __restrict__ uv2, const float* __restrict__ pose,
                                                           const float* __restrict__ K, int kstride, int R, float* __restrict__ dirs,
                                                           float* __restrict__ origins, float* __restrict__ dirs2, float* __restrict__ w2c,
                                                           float* __restrict__ K3) {
  __shared__ float sA[16], sInv[16];
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < R) {
    camera_ray(uv, pose, K, kstride, r, dirs, origins);
    if (uv2) camera_ray(uv2, pose, K, kstride, r, dirs2, nullptr);
  }
  if (blockIdx.x != 0) return;
  if (threadIdx.x < 16) sA[threadIdx.x] = pose[threadIdx.x];
  if (threadIdx.x < 9) K3[threadIdx.x] = K[(threadIdx.x / 3) * kstride + threadIdx.x % 3];
  __syncthreads();
  if (threadIdx.x == 0) inv_small_body<4>(sA, sInv);
  __syncthreads();
  if (threadIdx.x < 12) w2c[threadIdx.x] = sInv[threadIdx.x];
}

// ---------------------------------------------------------------------------------------------
// Global-junction MLP  j3d = W2 relu(W1 relu(W0 z + b0) + b1) + b2  (VolSDFNetwork.ffn on the latents, rend_a :303-313,491):
// J x 256 -> 256 -> 256 -> 3 in fp32.  Three launches replace ~25 (six latency-bound library GEMMs on 64 rows, bias / relu
// / reduction kernels): forward and data-backward are parallel over junction rows (FFN_RB rows per workgroup, thread =
// feature), the weight gradients are parallel over output features (deterministic sums over the rows).
// ---------------------------------------------------------------------------------------------
constexpr int FFN_H = 256, FFN_RB = 8;
constexpr int FFN_FUSED_MIN_ROWS = 4096;  // up to this many rows the layers run as separate many-CU launches (ffn_dense_kernel)

__global__ __launch_bounds__(FFN_H) void ffn_forward_kernel(const float* __restrict__ x, int J, const float* __restrict__ W0,
    const float* __restrict__ b0, const float* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ W2,
    const float* __restrict__ b2, float* __restrict__ h1, float* __restrict__ h2, float* __restrict__ y) {
  __shared__ float xs[FFN_RB][FFN_H], hs[FFN_RB][FFN_H];
  const int n = threadIdx.x, r0 = blockIdx.x * FFN_RB;
  const int nr = min(FFN_RB, J - r0);
  for (int r = 0; r < FFN_RB; ++r) xs[r][n] = r < nr ? x[(size_t)(r0 + r) * FFN_H + n] : 0.0f;
  __syncthreads();
  // thread n owns output feature n and walks its weight row in 32-column chunks: 8 independent float4 loads per chunk
  // (whole cache lines, deep memory parallelism) instead of 64 loads at dependent latency
  auto dense = [&](const float* __restrict__ W, const float* __restrict__ b, float (&in)[FFN_RB][FFN_H], float (&acc)[FFN_RB]) {
#pragma unroll
    for (int r = 0; r < FFN_RB; ++r) acc[r] = b[n];
#pragma unroll 1
    for (int kc = 0; kc < FFN_H; kc += 32) {
      float4 w4[8];
      const float4* wr = reinterpret_cast<const float4*>(W + (size_t)n * FFN_H + kc);
#pragma unroll
      for (int j = 0; j < 8; ++j) w4[j] = wr[j];
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < FFN_RB; ++r)
          acc[r] += w4[j].x * in[r][kc + 4 * j] + w4[j].y * in[r][kc + 4 * j + 1] + w4[j].z * in[r][kc + 4 * j + 2] + w4[j].w * in[r][kc + 4 * j + 3];
    }
  };
  float acc[FFN_RB];
  dense(W0, b0, xs, acc);
  for (int r = 0; r < FFN_RB; ++r) { const float v = fmaxf(acc[r], 0.0f); hs[r][n] = v; if (r < nr) h1[(size_t)(r0 + r) * FFN_H + n] = v; }
  __syncthreads();
  dense(W1, b1, hs, acc);
  __syncthreads();
  for (int r = 0; r < FFN_RB; ++r) { const float v = fmaxf(acc[r], 0.0f); xs[r][n] = v; if (r < nr) h2[(size_t)(r0 + r) * FFN_H + n] = v; }
  __syncthreads();
  if (n < 3 * FFN_RB) {
    const int r = n / 3, c = n % 3;
    float v = b2[c];
    for (int k = 0; k < FFN_H; ++k) v += W2[c * FFN_H + k] * xs[r][k];
    if (r < nr) y[(size_t)(r0 + r) * 3 + c] = v;
  }
}

// d_a2 = (W2^T dy) [h2 > 0] ; d_a1 = (W1^T d_a2) [h1 > 0] ; dx = W0^T d_a1
__global__ __launch_bounds__(FFN_H) void ffn_backward_data_kernel(const float* __restrict__ dy, int J, const float* __restrict__ W0,
    const float* __restrict__ W1, const float* __restrict__ W2, const float* __restrict__ h1, const float* __restrict__ h2,
    float* __restrict__ d_a1, float* __restrict__ d_a2, float* __restrict__ dx) {
  __shared__ __attribute__((aligned(16))) float ds[FFN_RB][FFN_H];
  __shared__ float dys[FFN_RB][3];
  const int k = threadIdx.x, r0 = blockIdx.x * FFN_RB;
  const int nr = min(FFN_RB, J - r0);
  if (k < 3 * FFN_RB) dys[k / 3][k % 3] = (k / 3 < nr) ? dy[(size_t)(r0 + k / 3) * 3 + k % 3] : 0.0f;
  __syncthreads();
  for (int r = 0; r < FFN_RB; ++r) {
    float v = dys[r][0] * W2[k] + dys[r][1] * W2[FFN_H + k] + dys[r][2] * W2[2 * FFN_H + k];
    v = (r < nr && h2[(size_t)(r0 + r) * FFN_H + k] > 0.0f) ? v : 0.0f;
    ds[r][k] = v;
    if (r < nr) d_a2[(size_t)(r0 + r) * FFN_H + k] = v;
  }
  __syncthreads();
  auto back = [&](const float* __restrict__ W, float (&acc)[FFN_RB]) {       // acc[r] = sum_n ds[r][n] W[n][k]
#pragma unroll
    for (int r = 0; r < FFN_RB; ++r) acc[r] = 0.0f;
    for (int n = 0; n < FFN_H; ++n) {
      const float w = W[(size_t)n * FFN_H + k];
#pragma unroll
      for (int r = 0; r < FFN_RB; ++r) acc[r] += ds[r][n] * w;
    }
  };
  float acc[FFN_RB];
  back(W1, acc);
  __syncthreads();
  for (int r = 0; r < FFN_RB; ++r) {
    const float v = (r < nr && h1[(size_t)(r0 + r) * FFN_H + k] > 0.0f) ? acc[r] : 0.0f;
    ds[r][k] = v;
    if (r < nr) d_a1[(size_t)(r0 + r) * FFN_H + k] = v;
  }
  __syncthreads();
  back(W0, acc);
  for (int r = 0; r < nr; ++r) dx[(size_t)(r0 + r) * FFN_H + k] = acc[r];
}

// One dense layer of the junction MLP for few rows, spread over many CUs (the fused kernels above keep 8 rows per
// workgroup and walk the whole 256 KiB matrix per layer on ONE CU: with J = 64 latents that is 8 CUs at L2 latency):
//   y[j][o] = epi( sum_i x[j][i] Wm(o, i) ),  Wm(o, i) = TRANS ? W[i * O + o] : W[o * I + i]
//   epi: + bias[o] (if bias) ; relu (if relu) ; zero where gate[j][o] <= 0 (if gate: the relu mask of a backward step)
// grid (ceil(J / 8), ceil(O / 32)); thread = (output o0 + (tid & 31), input slice tid >> 5 of 8); the 8 slice sums meet in LDS.
template <bool TRANS>
__global__ __launch_bounds__(256) void ffn_dense_kernel(const float* __restrict__ x, int J, int I, int O, const float* __restrict__ W,
    const float* __restrict__ bias, const float* __restrict__ gate, int relu, float* __restrict__ y, float* __restrict__ y2) {
  __shared__ __attribute__((aligned(16))) float xs[FFN_RB][FFN_H];
  __shared__ float red[8][FFN_RB][33];
  const int tid = threadIdx.x, f = tid & 31, ks = tid >> 5;
  const int r0 = blockIdx.x * FFN_RB, o = blockIdx.y * 32 + f;
  const int nr = min(FFN_RB, J - r0);
  for (int idx = tid; idx < FFN_RB * FFN_H; idx += 256) {
    const int r = idx / FFN_H, i = idx % FFN_H;
    xs[r][i] = (r < nr && i < I) ? x[(size_t)(r0 + r) * I + i] : 0.0f;
  }
  const int IS = ((I + 7) / 8 + 3) & ~3;           // inputs per slice (multiple of 4)
  const int i0 = ks * IS, i1 = min(I, i0 + IS);
  // the weights of this thread's slice first (independent loads in flight), then the FMAs
  float w[32];
#pragma unroll
  for (int t = 0; t < 32; ++t) w[t] = 0.0f;
  if (o < O) {
    if (!TRANS && (I & 3) == 0) {
#pragma unroll
      for (int t = 0; t < 32; t += 4)
        if (i0 + t < i1) {
          const float4 v = *reinterpret_cast<const float4*>(W + (size_t)o * I + i0 + t);
          w[t] = v.x; w[t + 1] = v.y; w[t + 2] = v.z; w[t + 3] = v.w;
        }
    } else {
#pragma unroll
      for (int t = 0; t < 32; ++t)
        if (i0 + t < i1) w[t] = TRANS ? W[(size_t)(i0 + t) * O + o] : W[(size_t)o * I + i0 + t];
    }
  }
  __syncthreads();
  float acc[FFN_RB];
#pragma unroll
  for (int r = 0; r < FFN_RB; ++r) acc[r] = 0.0f;
  if (i0 < I) {
#pragma unroll
    for (int t = 0; t < 32; t += 4)
#pragma unroll
      for (int r = 0; r < FFN_RB; ++r) {
        const float4 x4 = *reinterpret_cast<const float4*>(&xs[r][min(i0 + t, FFN_H - 4)]);
        acc[r] += x4.x * w[t]; acc[r] += x4.y * w[t + 1]; acc[r] += x4.z * w[t + 2]; acc[r] += x4.w * w[t + 3];
      }
  }
#pragma unroll
  for (int r = 0; r < FFN_RB; ++r) red[ks][r][f] = acc[r];
  __syncthreads();
  const int r = ks;                                 // thread (row r, output f) finishes one output
  if (r < nr && o < O) {
    float v = bias ? bias[o] : 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) v += red[k][r][f];
    if (relu) v = fmaxf(v, 0.0f);
    if (gate && !(gate[(size_t)(r0 + r) * O + o] > 0.0f)) v = 0.0f;
    y[(size_t)(r0 + r) * O + o] = v;
    if (y2) y2[(size_t)(r0 + r) * O + o] = v;
  }
}

// The two 256 x 256 layers of the junction MLP on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32: exact fp32 products and sums, 64 flop per cycle
// and SIMD instead of the vector ALU's 64 per CU-quarter with its LDS operand reads):  y[j][o] = epi( sum_i x[j][i] Wm(o, i) ), I = O = 256,
// Wm(o, i) = TRANS ? W[i * 256 + o] : W[o * 256 + i];  epi as ffn_dense_kernel.  Workgroup = one 32 (outputs) x 32 (rows) tile, its four
// waves split the 256 inputs (64 each: 32 MFMAs), the four partial tiles meet in LDS and are summed in wave order (deterministic).
// k-order inside a chunk of 8 inputs: MFMA step s takes inputs (8c + s, 8c + 4 + s) -- lane half kh holds inputs 8c + 4 kh .. + 3 of its
// row as one float4, for both operands alike.
template <bool TRANS>
__global__ __launch_bounds__(256) void ffn_mfma_kernel(const float* __restrict__ x, int J, const float* __restrict__ W,
    const float* __restrict__ bias, const float* __restrict__ gate, int relu, float* __restrict__ y) {
  __shared__ float part[4][32][33];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 31, kh = lane >> 5;
  const int j0 = blockIdx.x * 32, o0 = blockIdx.y * 32, k0 = 64 * wave;
  const int jrow = min(j0 + m, J - 1);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  float4 av[8], bv[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int i = k0 + 8 * c + 4 * kh;
    bv[c] = *reinterpret_cast<const float4*>(x + (size_t)jrow * FFN_H + i);
    if (!TRANS) av[c] = *reinterpret_cast<const float4*>(W + (size_t)(o0 + m) * FFN_H + i);
    else av[c] = make_float4(W[(size_t)i * FFN_H + o0 + m], W[(size_t)(i + 1) * FFN_H + o0 + m], W[(size_t)(i + 2) * FFN_H + o0 + m],
                             W[(size_t)(i + 3) * FFN_H + o0 + m]);
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c].x, bv[c].x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c].y, bv[c].y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c].z, bv[c].z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c].w, bv[c].w, acc, 0, 0, 0);
  }
  // accumulator layout: column (row j) = lane & 31, output row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int r = 0; r < 16; ++r) part[wave][(r & 3) + 8 * (r >> 2) + 4 * kh][m] = acc[r];
  __syncthreads();
  const int oo = tid & 31;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int jj = (tid >> 5) + 8 * q, j = j0 + jj, o = o0 + oo;
    if (j >= J) continue;
    float v = bias ? bias[o] : 0.0f;
    v += part[0][oo][jj]; v += part[1][oo][jj]; v += part[2][oo][jj]; v += part[3][oo][jj];
    if (relu) v = fmaxf(v, 0.0f);
    if (gate && !(gate[(size_t)j * FFN_H + o] > 0.0f)) v = 0.0f;
    y[(size_t)j * FFN_H + o] = v;
  }
}

// blockIdx.y: 0: dW0 = d_a1^T x, db0 ; 1: dW1 = d_a2^T h1, db1 ; 2 (rows 0..2): dW2 = dy^T h2, db2.  blockIdx.x = FFN_RN output rows.
// 1024 threads = 16 wavefronts: wavefront g sums ITS sixteenth of the J rows (a lane = four columns of the block's output rows: one 16-byte load
// per row of the input, eight rows in flight), the sixteen partial sums are combined in group order (deterministic).  One thread per
// output over all J rows was a chain of J dependent trips to L2 with every block re-reading the whole input: 87 us at the J = 1024
// latents of the DTU / BlendedMVS confs.
constexpr int FFN_JG = 16, FFN_RN = 4;       // wavefronts (row groups) per block, output rows per block
__global__ __launch_bounds__(64 * FFN_JG) void ffn_backward_weights_kernel(const float* __restrict__ x, const float* __restrict__ h1,
    const float* __restrict__ h2, const float* __restrict__ d_a1, const float* __restrict__ d_a2, const float* __restrict__ dy, int J,
    float* __restrict__ dW0, float* __restrict__ db0, float* __restrict__ dW1, float* __restrict__ db1, float* __restrict__ dW2,
    float* __restrict__ db2) {
  __shared__ __attribute__((aligned(16))) float red[FFN_JG][FFN_RN][FFN_H];
  __shared__ float bred[FFN_JG][FFN_RN];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6, which = blockIdx.y;
  const int n0 = FFN_RN * blockIdx.x;
  const int N = which == 2 ? 3 : FFN_H;
  if (n0 >= N) return;
  const float* d = which == 0 ? d_a1 : (which == 1 ? d_a2 : dy);
  const float* in = which == 0 ? x : (which == 1 ? h1 : h2);
  const int ldd = which == 2 ? 3 : FFN_H;
  const int per = (J + FFN_JG - 1) / FFN_JG;
  const int jb = min(J, grp * per), je = min(J, jb + per);
  float4 a[FFN_RN]; float bs[FFN_RN];
#pragma unroll
  for (int r = 0; r < FFN_RN; ++r) { a[r] = make_float4(0.f, 0.f, 0.f, 0.f); bs[r] = 0.0f; }
  int j = jb;
  for (; j + 4 <= je; j += 4) {               // four rows of the input in flight (the sum order of a group stays j = jb, jb + 1, ...)
    float dv[4][FFN_RN]; float4 xv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int r = 0; r < FFN_RN; ++r) dv[t][r] = n0 + r < N ? d[(size_t)(j + t) * ldd + n0 + r] : 0.0f;
      xv[t] = *reinterpret_cast<const float4*>(in + (size_t)(j + t) * FFN_H + 4 * lane);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < FFN_RN; ++r) {
        a[r].x += dv[t][r] * xv[t].x; a[r].y += dv[t][r] * xv[t].y; a[r].z += dv[t][r] * xv[t].z; a[r].w += dv[t][r] * xv[t].w;
        bs[r] += dv[t][r];
      }
  }
  for (; j < je; ++j) {
    const float4 xv = *reinterpret_cast<const float4*>(in + (size_t)j * FFN_H + 4 * lane);
#pragma unroll
    for (int r = 0; r < FFN_RN; ++r) {
      const float dv = n0 + r < N ? d[(size_t)j * ldd + n0 + r] : 0.0f;
      a[r].x += dv * xv.x; a[r].y += dv * xv.y; a[r].z += dv * xv.z; a[r].w += dv * xv.w;
      bs[r] += dv;
    }
  }
#pragma unroll
  for (int r = 0; r < FFN_RN; ++r) {
    *reinterpret_cast<float4*>(&red[grp][r][4 * lane]) = a[r];
    if (lane == 0) bred[grp][r] = bs[r];
  }
  __syncthreads();
  const int r = threadIdx.x >> 8, k = threadIdx.x & (FFN_H - 1);       // 1024 threads = FFN_RN rows x 256 columns
  if (n0 + r >= N) return;
  float v = 0.0f, b = 0.0f;
#pragma unroll
  for (int g = 0; g < FFN_JG; ++g) { v += red[g][r][k]; b += bred[g][r]; }
  float* dW = which == 0 ? dW0 : (which == 1 ? dW1 : dW2);
  float* db = which == 0 ? db0 : (which == 1 ? db1 : db2);
  dW[(size_t)(n0 + r) * FFN_H + k] = v;
  if (k == 0) db[n0 + r] = b;
}

// The same three weight gradients on the fp32 matrix pipe: dW[o][i] = sum_j d[j][o] in[j][i].  grid (8 input tiles, 8 output tiles, 3 layers;
// the 3-row layer uses output tile 0 only); a workgroup = one 32 x 32 tile of dW, its eight waves split the J rows (contiguous ranges, rows 2t / 2t + 1
// per MFMA), partial tiles and the bias partial sums meet in LDS and are added in wave order (deterministic).
constexpr int FFN_WNW = 8;          // waves of ffn_wgrad_mfma_kernel (each takes 1 / FFN_WNW of the rows)
__global__ __launch_bounds__(64 * FFN_WNW) void ffn_wgrad_mfma_kernel(const float* __restrict__ x, const float* __restrict__ h1, const float* __restrict__ h2,
    const float* __restrict__ d_a1, const float* __restrict__ d_a2, const float* __restrict__ dy, int J, float* __restrict__ dW0,
    float* __restrict__ db0, float* __restrict__ dW1, float* __restrict__ db1, float* __restrict__ dW2, float* __restrict__ db2) {
  __shared__ float part[FFN_WNW][32][33];
  __shared__ float bpart[FFN_WNW][32];
  const int which = blockIdx.z;
  const int N = which == 2 ? 3 : FFN_H;
  const int i0 = blockIdx.x * 32, o0 = blockIdx.y * 32;
  if (o0 >= N) return;
  const float* d = which == 0 ? d_a1 : (which == 1 ? d_a2 : dy);
  const float* in = which == 0 ? x : (which == 1 ? h1 : h2);
  const int ldd = which == 2 ? 3 : FFN_H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 31, kh = lane >> 5;
  const bool okA = o0 + m < N;
  const int per = ((J + FFN_WNW - 1) / FFN_WNW + 1) & ~1;                 // rows per wave (even)
  const int jb = min(J, wave * per), je = min(J, jb + per);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  float bs = 0.0f;
  const float* dp = d + (okA ? o0 + m : 0);
  const float* ip = in + i0 + m;
  int j = jb;
  for (; j + 16 <= je; j += 16) {                        // eight MFMAs per trip, their sixteen operand loads in flight together
    float av[8], bv[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int jj = j + 2 * t + kh;
      av[t] = okA ? dp[(size_t)jj * ldd] : 0.0f;
      bv[t] = ip[(size_t)jj * FFN_H];
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[t], acc, 0, 0, 0);
      bs += av[t];
    }
  }
  for (; j < je; j += 2) {
    const int jj = j + kh;
    const bool okj = jj < je;
    const float a = (okA && okj) ? dp[(size_t)jj * ldd] : 0.0f;
    const float b = okj ? ip[(size_t)jj * FFN_H] : 0.0f;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    bs += a;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) part[wave][(r & 3) + 8 * (r >> 2) + 4 * kh][m] = acc[r];
  bs += __shfl_xor(bs, 32);
  if (kh == 0) bpart[wave][m] = bs;
  __syncthreads();
  float* dW = which == 0 ? dW0 : (which == 1 ? dW1 : dW2);
  float* db = which == 0 ? db0 : (which == 1 ? db1 : db2);
  const int ii = tid & 31;
  for (int oo = tid >> 5; oo < 32; oo += 2 * FFN_WNW) {
    if (o0 + oo >= N) continue;
    float v = 0.0f;
#pragma unroll
    for (int w = 0; w < FFN_WNW; ++w) v += part[w][oo][ii];
    dW[(size_t)(o0 + oo) * FFN_H + i0 + ii] = v;
  }
  if (blockIdx.x == 0 && tid < 32 && o0 + tid < N) {
    float b = 0.0f;
#pragma unroll
    for (int w = 0; w < FFN_WNW; ++w) b += bpart[w][tid];
    db[o0 + tid] = b;
  }
}

// ---------------------------------------------------------------------------------------------
// DBSCAN(eps, min_samples = 2) of n 3-D points and the cluster means, on the device: the junction candidates of the
// DTU / BlendedMVS confs (VolSDFNetwork.cluster_dbscan, rend_a :328-339, sklearn on the host in the reference).  With
// min_samples = 2 every point that has a neighbour within eps is a core point, so the clusters are exactly the
// connected components (of size >= 2) of the eps-graph, and sklearn numbers them by their first point: ascending minimum
// index.  Output is padded: centres [n/2][3] in that order, valid [n/2], *count.
//   1. dbscan_union_kernel: every pair closer than eps (float64 distances of the float32 points, as sklearn computes them) is
//      merged in a lock-free union-find whose links always point from the larger root to the smaller one, so the root of a
//      component is its minimum index whatever the order of the merges;
//   2. dbscan_finish_kernel (one workgroup): roots -> LDS, ordered compaction of the component minima that have a
//      neighbour, means summed in index order (deterministic).
// ---------------------------------------------------------------------------------------------
__global__ void dbscan_init_kernel(int* __restrict__ parent, int* __restrict__ has_nb, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { parent[i] = i; has_nb[i] = 0; }
}

__device__ __forceinline__ int dbscan_find(int* parent, int x) {
  while (true) {
    const int p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (p == x) return x;
    const int gp = __hip_atomic_load(&parent[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (gp != p) atomicMin(&parent[x], gp);      // path halving; parents only ever decrease
    x = p;
  }
}

__global__ void dbscan_union_kernel(const float* __restrict__ pts, int n, double eps2, int* __restrict__ parent, int* __restrict__ has_nb) {
  const int i = blockIdx.x;
  const double xi = pts[3 * i], yi = pts[3 * i + 1], zi = pts[3 * i + 2];
  for (int j = i + 1 + threadIdx.x; j < n; j += blockDim.x) {
    const double dx = xi - (double)pts[3 * j], dy = yi - (double)pts[3 * j + 1], dz = zi - (double)pts[3 * j + 2];
    if (dx * dx + dy * dy + dz * dz > eps2) continue;
    has_nb[i] = 1; has_nb[j] = 1;
    int a = i, b = j;
    while (true) {
      a = dbscan_find(parent, a); b = dbscan_find(parent, b);
      if (a == b) break;
      const int lo = min(a, b), hi = max(a, b);
      if (atomicCAS(&parent[hi], hi, lo) == hi) break;      // hi was still a root: linked under the smaller root
      a = lo; b = hi;                                        // somebody re-rooted hi meanwhile: find again
    }
  }
}

constexpr int DBSCAN_MAXN = 8192;
__global__ __launch_bounds__(1024) void dbscan_finish_kernel(const float* __restrict__ pts, int n, int* __restrict__ parent,
                                                             const int* __restrict__ has_nb, float* __restrict__ centres,
                                                             unsigned char* __restrict__ valid, int* __restrict__ count, int mode) {
  __shared__ int lab[DBSCAN_MAXN];
  __shared__ int s_wave[16], s_base;
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int i = tid; i < n; i += nt) lab[i] = dbscan_find(parent, i);
  const int maxc = n / 2;
  for (int k = tid; k < maxc; k += nt) { valid[k] = 0; centres[3 * k] = 0.f; centres[3 * k + 1] = 0.f; centres[3 * k + 2] = 0.f; }
  int* replist = reinterpret_cast<int*>(centres + 3 * maxc);      // caller provides n/2 ints of space behind the centres
  if (tid == 0) s_base = 0;
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
  for (int b0 = 0; b0 < n; b0 += nt) {          // ordered compaction of the representatives (ascending = sklearn's numbering)
    const int i = b0 + tid;
    const bool f = i < n && lab[i] == i && has_nb[i] != 0;
    const unsigned long long bal = __ballot(f);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave[wave] = __popcll(bal);
    __syncthreads();
    int off = s_base;
    for (int w = 0; w < wave; ++w) off += s_wave[w];
    if (f) replist[off + before] = i;
    __syncthreads();
    if (tid == 0) { int t = 0; for (int w = 0; w < nw; ++w) t += s_wave[w]; s_base += t; }
    __syncthreads();
  }
  const int nclu = s_base;
  extern __shared__ __attribute__((aligned(8))) unsigned char dbscan_dyn[];
  if (mode == 2) {
    // O(n): every member adds its coordinates to its cluster's sums with LDS atomics -- in 2^-32 FIXED POINT, so that the sums do not depend
    // on the order of the additions (deterministic; 2.3e-10 per coordinate, three orders below the float32 mean's own rounding; |x| < 2^19).
    // rank[r] = cluster number of representative r.  (One wavefront per cluster walking all points, below, is O(clusters x n): 133 us at
    // the 4096 line end points of a 2048-ray batch.)
    long long* sx = reinterpret_cast<long long*>(dbscan_dyn);
    long long* sy = sx + maxc; long long* sz = sy + maxc;
    int* cnt = reinterpret_cast<int*>(sz + maxc);
    int* rank = cnt + maxc;
    for (int k = tid; k < nclu; k += nt) { sx[k] = 0; sy[k] = 0; sz[k] = 0; cnt[k] = 0; rank[replist[k]] = k; }
    __syncthreads();
    for (int j = tid; j < n; j += nt) {
      if (has_nb[j] == 0) continue;                  // noise: a component of one point
      const int k = rank[lab[j]];
      atomicAdd(reinterpret_cast<unsigned long long*>(&sx[k]), (unsigned long long)__double2ll_rn((double)pts[3 * j] * 4294967296.0));
      atomicAdd(reinterpret_cast<unsigned long long*>(&sy[k]), (unsigned long long)__double2ll_rn((double)pts[3 * j + 1] * 4294967296.0));
      atomicAdd(reinterpret_cast<unsigned long long*>(&sz[k]), (unsigned long long)__double2ll_rn((double)pts[3 * j + 2] * 4294967296.0));
      atomicAdd(&cnt[k], 1);
    }
    __syncthreads();
    for (int k = tid; k < nclu; k += nt) {
      const double inv = 1.0 / (4294967296.0 * (double)cnt[k]);
      centres[3 * k] = (float)((double)sx[k] * inv); centres[3 * k + 1] = (float)((double)sy[k] * inv); centres[3 * k + 2] = (float)((double)sz[k] * inv);
      valid[k] = 1;
    }
    if (tid == 0) *count = nclu;
    return;
  }
  // the points next to the labels (dynamic LDS, 12 n bytes, mode 1)
  float* dbscan_pts = reinterpret_cast<float*>(dbscan_dyn);
  const float* P = pts;
  if (mode == 1) {
    for (int i = tid; i < 3 * n; i += nt) dbscan_pts[i] = pts[i];
    P = dbscan_pts;
  }
  __syncthreads();
  // mean of a cluster's members: one wavefront per cluster, lane l sums members r + l, r + l + 64, ... in index order, the 64 partial
  // sums are combined by a fixed shuffle tree (deterministic)
  for (int k = wave; k < nclu; k += nw) {
    const int r = replist[k];
    float sx = 0.f, sy = 0.f, sz = 0.f; int cnt = 0;
    for (int j = r + lane; j < n; j += 64) {
      const bool m = lab[j] == r;
      sx += m ? P[3 * j] : 0.0f; sy += m ? P[3 * j + 1] : 0.0f; sz += m ? P[3 * j + 2] : 0.0f; cnt += m ? 1 : 0;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      sx += __shfl_xor(sx, off); sy += __shfl_xor(sy, off); sz += __shfl_xor(sz, off); cnt += __shfl_xor(cnt, off);
    }
    if (lane == 0) {
      const float inv = 1.0f / (float)cnt;
      centres[3 * k] = sx * inv; centres[3 * k + 1] = sy * inv; centres[3 * k + 2] = sz * inv;
      valid[k] = 1;
    }
  }
  if (tid == 0) *count = nclu;
}

// ---------------------------------------------------------------------------------------------
// The tail of VolSDFLoss.forward (model/networks/loss_wfr.py:68-137) in two launches around the junction matching:
//   loss_terms_kernel : rgb L1 (mean), eikonal ((|g| - 1)^2 mean), their cotangents, and the junction pair cost
//                       cdist_1(loc3, glo3) + 0.1 cdist_1(loc2c, glo2c)  [K, J]
//   loss_pairs_kernel : matched-pair means (3-D, calibrated 2-D, pixel 2-D), the count of pairs with cost < 10, and the
//                       cotangents of the global junctions (only matched entries are non-zero)
// One workgroup each (R, E, K, J are a few thousand at most); sums by wave shuffles + LDS, fixed order.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum(float v, float* s_red) {          // all threads get the sum; blockDim.x = 1024
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) s_red[wave] = v;
  __syncthreads();
  float t = 0.0f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += s_red[w];
  return t;
}

struct LossTermsArgs {
  const float* rgb; const float* rgb_gt; int R;
  const float* gtheta; int E;
  const float* loc3; const float* loc2c; int K;
  const float* glo3; const float* glo2c; int J;
  float* scal;                 // [0] rgb loss, [1] eikonal loss
  float* d_rgb; float* d_gtheta; float* pair_cost;
  float eik_grad_scale;        // d_gtheta = eik_grad_scale * d scal[1] / d gtheta (the eikonal weight: gradients of the TOTAL loss, or 1)
};

__device__ __forceinline__ void loss_terms_body(const LossTermsArgs& a) {
  __shared__ float s_red[16];
  const int tid = threadIdx.x, nt = blockDim.x;
  float acc = 0.0f;
  const float inv_rgb = 1.0f / (float)(3 * a.R);
  for (int i = tid; i < 3 * a.R; i += nt) {
    const float d = a.rgb[i] - a.rgb_gt[i];
    acc += fabsf(d);
    a.d_rgb[i] = (d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f)) * inv_rgb;
  }
  const float rgb_loss = block_sum(acc, s_red) * inv_rgb;
  acc = 0.0f;
  const float inv_e = a.E > 0 ? 1.0f / (float)a.E : 0.0f;
  for (int i = tid; i < a.E; i += nt) {
    const float gx = a.gtheta[3 * i], gy = a.gtheta[3 * i + 1], gz = a.gtheta[3 * i + 2];
    const float nrm = sqrtf(gx * gx + gy * gy + gz * gz);
    acc += (nrm - 1.0f) * (nrm - 1.0f);
    const float c = nrm > 0.0f ? (2.0f * (nrm - 1.0f) * inv_e / nrm) * a.eik_grad_scale : 0.0f;        // d/dg |g| = g/|g| (0 at the origin, as torch)
    a.d_gtheta[3 * i] = c * gx; a.d_gtheta[3 * i + 1] = c * gy; a.d_gtheta[3 * i + 2] = c * gz;
  }
  const float eik = block_sum(acc, s_red) * inv_e;
  if (tid == 0) { a.scal[0] = rgb_loss; a.scal[1] = eik; }
  for (int idx = tid; idx < a.K * a.J; idx += nt) {
    const int k = idx / a.J, j = idx % a.J;
    float c3 = 0.0f, c2 = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) c3 += fabsf(a.loc3[3 * k + c] - a.glo3[3 * j + c]);
#pragma unroll
    for (int c = 0; c < 2; ++c) c2 += fabsf(a.loc2c[2 * k + c] - a.glo2c[2 * j + c]);
    a.pair_cost[idx] = c3 + 0.1f * c2;
  }
}

__global__ __launch_bounds__(1024) void loss_terms_kernel(LossTermsArgs a) { loss_terms_body(a); }

// Both independent halves of the loss after the projections in ONE launch (round 6): workgroup 0 = the two line terms
// (line_losses_body), workgroup 1 = rgb + eikonal + the junction pair cost (loss_terms_body).  Same arithmetic as the two launches.
struct LineLossesArgs { const float* pred_u; const float* pred_c; const float* gt5; const float* K; int R; float thr; float* out; float* d_pred_c; float grad_scale;
                        const float* w2c; const float* X3; float* d_X3; };
__global__ __launch_bounds__(1024) void loss_lines_terms_kernel(LineLossesArgs l, LossTermsArgs t) {
  if (blockIdx.x == 0) line_losses_body(l.pred_u, l.pred_c, l.gt5, l.K, l.R, l.thr, l.out, l.d_pred_c, l.grad_scale, l.w2c, l.X3, l.d_X3);
  else loss_terms_body(t);
}

struct LossPairsArgs {
  const long long* ri; const long long* ci; const int* n_match; int Kmax;      // pairs (padded with -1)
  const float* loc3; const float* loc2c; const float* loc2;
  const float* glo3; const float* glo2c; const float* glo2; int J;
  const float* pair_cost;
  float* scal;                 // [2] j3d, [3] j2d (calibrated), [4] j2d (pixels), [5] count of pairs with cost < 10
  float* d_glo3; float* d_glo2c;                                               // [J,3], [J,2], fully written
  const float* line_loss; float w_eik, w_line, w_j3, w_j2;                     // scal[6] = rgb + w_eik eik + w_line line + w_j3 j3d + w_j2 j2d
  int weighted_grads;                                                           // 1: d_glo3 / d_glo2c carry w_j3 / w_j2 (gradients of scal[6])
  float* total;                                                                 // scal[6] once more, in a tensor of its own (or null)
  const float* w2c;            // round 6 (null: not wanted): glo2c = project2d(I, w2c, glo3) (rend_a :496) -- d_glo2c is carried on into d_glo3 here
};

__global__ __launch_bounds__(1024) void loss_pairs_kernel(LossPairsArgs a) {
  __shared__ float s_red[16];
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int i = tid; i < 3 * a.J; i += nt) a.d_glo3[i] = 0.0f;
  for (int i = tid; i < 2 * a.J; i += nt) a.d_glo2c[i] = 0.0f;
  __syncthreads();
  const int n = max(*a.n_match, 0);
  const float inv = 1.0f / (float)max(n, 1);
  const float g3 = a.weighted_grads ? a.w_j3 : 1.0f, g2 = a.weighted_grads ? a.w_j2 : 1.0f;
  float s3 = 0.0f, s2 = 0.0f, spx = 0.0f, cnt = 0.0f;
  for (int q = tid; q < a.Kmax; q += nt) {
    const long long r = a.ri[q], c = a.ci[q];
    if (r < 0 || c < 0 || q >= n) continue;
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      const float d = a.loc3[3 * r + e] - a.glo3[3 * c + e];
      s3 += fabsf(d);
      a.d_glo3[3 * c + e] = (-(d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f)) * inv) * g3;     // each global junction is matched at most once
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float d = a.loc2c[2 * r + e] - a.glo2c[2 * c + e];
      s2 += fabsf(d);
      a.d_glo2c[2 * c + e] = (-(d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f)) * inv) * g2;
      spx += fabsf(a.loc2[2 * r + e] - a.glo2[2 * c + e]);
    }
    if (a.pair_cost[r * a.J + c] < 10.0f) cnt += 1.0f;
  }
  s3 = block_sum(s3, s_red); s2 = block_sum(s2, s_red); spx = block_sum(spx, s_red); cnt = block_sum(cnt, s_red);
  if (a.w2c) {                 // (block_sum's barriers order the writes above): d_glo3 += (d glo2c / d glo3)^T d_glo2c, project2d_bwd_kernel's arithmetic
    for (int j = tid; j < a.J; j += nt) {
      const float x[3] = {a.glo3[3 * j], a.glo3[3 * j + 1], a.glo3[3 * j + 2]};
      float dX[3];
      project_bwd_calib(a.w2c, x, a.d_glo2c[2 * j], a.d_glo2c[2 * j + 1], dX);
#pragma unroll
      for (int e = 0; e < 3; ++e) a.d_glo3[3 * j + e] += dX[e];
    }
  }
  if (tid == 0) {
    a.scal[2] = s3 * inv; a.scal[3] = s2 * inv; a.scal[4] = spx * inv; a.scal[5] = cnt;
    const float tot = a.scal[0] + a.w_eik * a.scal[1] + a.w_line * a.line_loss[0] + a.w_j3 * (s3 * inv) + a.w_j2 * (s2 * inv);
    a.scal[6] = tot;
    if (a.total) a.total[0] = tot;
  }
}

// ---------------------------------------------------------------------------------------------
// More of the junction block (rend_a :441-489) as single launches:
//   l3d_kernel        : t = <x - o, n> / (<d, n> +- 1e-6), l3d = o + t d   (intersection of the projected-pixel ray with the
//                       tangent plane of the rendered surface point), per ray
//   junction_cost_kernel : cost[v][c] = | cand2d[c] - gt2d[v] |_2   (the arithmetic of torch's ((a-b)**2).sum(-1).sqrt())
//   junction_gate_kernel : matched cost per pair, (nan)median gate or the fixed 10 px gate, and the matched candidates
//                       gathered into padded [K, .] arrays + mask (pairs beyond the device-side count: masked, zeros)
// ---------------------------------------------------------------------------------------------
__global__ void l3d_kernel(const float* __restrict__ x, const float* __restrict__ o, const float* __restrict__ d,
                           const float* __restrict__ nrm, int R, float* __restrict__ l3d) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  float den = 0.0f, num = 0.0f;
#pragma unroll
  for (int c = 0; c < 3; ++c) { den += d[3 * r + c] * nrm[3 * r + c]; num += (x[3 * r + c] - o[3 * r + c]) * nrm[3 * r + c]; }
  den += den >= 0.0f ? 1e-6f : -1e-6f;
  const float t = num / den;
#pragma unroll
  for (int c = 0; c < 3; ++c) l3d[3 * r + c] = o[3 * r + c] + d[3 * r + c] * t;
}

__global__ void junction_cost_kernel(const float* __restrict__ cand2d, const float* __restrict__ gt2d, int V, int C, float* __restrict__ cost) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= V * C) return;
  const int v = idx / C, c = idx % C;
  const float dx = cand2d[2 * c] - gt2d[2 * v], dy = cand2d[2 * c + 1] - gt2d[2 * v + 1];
  cost[idx] = __fsqrt_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
}

struct JunctionGateArgs {
  const long long* rows; const long long* cols; int K;         // pairs from neat_lsap (-1 = no pair)
  const float* cost; int C;
  const float* cand3d; const float* cand2d; const float* cand2dc;
  int use_median;
  float* median;                                                // [1] (use_median)
  unsigned char* good;                                          // [K]
  float* j3d; float* j2d; float* j2dc;                          // [K,3], [K,2], [K,2]
};

__global__ __launch_bounds__(1024) void junction_gate_kernel(JunctionGateArgs a) {
  __shared__ float s_m[2048];
  __shared__ int s_n;
  __shared__ float s_med;
  const int tid = threadIdx.x, nt = blockDim.x;
  if (tid == 0) { s_n = 0; s_med = 10.0f; }
  __syncthreads();
  for (int k = tid; k < a.K; k += nt) {
    const long long r = a.rows[k], c = a.cols[k];
    const bool ok = r >= 0 && c >= 0;
    s_m[k] = ok ? a.cost[r * a.C + c] : NAN;
    if (ok) atomicAdd(&s_n, 1);
  }
  __syncthreads();
  const int n = s_n;
  if (a.use_median && n > 0) {               // lower median of the valid matched costs = element of rank (n-1)/2 (torch.median / nanmedian)
    const int want = (n - 1) / 2;
    for (int k = tid; k < a.K; k += nt) {
      const float v = s_m[k];
      if (v != v) continue;
      int rank = 0;
      for (int j = 0; j < a.K; ++j) { const float w = s_m[j]; if (w == w && (w < v || (w == v && j < k))) ++rank; }
      if (rank == want) s_med = v;
    }
  }
  __syncthreads();
  const float thr = a.use_median ? s_med : 10.0f;
  if (tid == 0 && a.use_median) a.median[0] = s_med;
  for (int k = tid; k < a.K; k += nt) {
    const float v = s_m[k];
    const bool ok = v == v;
    a.good[k] = (ok && v < thr) ? 1 : 0;
    const long long c = ok ? a.cols[k] : 0;
#pragma unroll
    for (int e = 0; e < 3; ++e) a.j3d[3 * k + e] = ok ? a.cand3d[3 * c + e] : 0.0f;
#pragma unroll
    for (int e = 0; e < 2; ++e) { a.j2d[2 * k + e] = ok ? a.cand2d[2 * c + e] : 0.0f; a.j2dc[2 * k + e] = ok ? a.cand2dc[2 * c + e] : 0.0f; }
  }
}

}  // namespace neat

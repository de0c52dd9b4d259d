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
  return *s_base;
}

__global__ __launch_bounds__(LSAP_WG) void lsap_kernel(LsapArgs a) {
  __shared__ int s_wave[LSAP_WG / 64], s_base;
  __shared__ LsapKey s_key[LSAP_WG / 64];
  __shared__ int s_i, s_sink, s_nrem, s_fail;
  __shared__ double s_minval;
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
  const int kmax = min(a.nr, a.nc);
  for (int k = tid; k < kmax; k += nt) { a.row_ind[k] = -1; a.col_ind[k] = -1; }

  int* rowlist = a.wsi;
  const unsigned char* mask = a.row_mask;
  const int n_eff = lsap_compact(a.nr, [&](int i) { return mask ? mask[i] != 0 : true; }, rowlist, s_wave, &s_base);
  const bool T = a.nc < n_eff;
  const int N = T ? a.nc : n_eff, M = T ? n_eff : a.nc;
  if (N == 0) { if (tid == 0) *a.n_match = 0; return; }
  double* u = a.wsd; double* v = u + N; double* sp = v + M;
  int* path = rowlist + a.nr; int* col4row = path + M; int* row4col = col4row + N; int* remaining = row4col + M;
  int* SR = remaining + M; int* SC = SR + N; int* tmp = SC + M;
  const float* C = a.cost;
  const int nc0 = a.nc;
  auto cost = [&](int i, int j) -> double {
    return (double)(T ? C[(size_t)rowlist[j] * nc0 + i] : C[(size_t)rowlist[i] * nc0 + j]);
  };
  for (int i = tid; i < N; i += nt) { u[i] = 0.0; col4row[i] = -1; }
  for (int j = tid; j < M; j += nt) { v[j] = 0.0; row4col[j] = -1; }
  if (tid == 0) s_fail = 0;
  __syncthreads();

  for (int cur = 0; cur < N; ++cur) {
    for (int j = tid; j < M; j += nt) { remaining[j] = M - j - 1; SC[j] = 0; sp[j] = INFINITY; }
    for (int i = tid; i < N; i += nt) SR[i] = 0;
    if (tid == 0) { s_i = cur; s_sink = -1; s_nrem = M; s_minval = 0.0; }
    __syncthreads();
    while (true) {
      const int i = s_i, nrem = s_nrem;
      const double minval = s_minval, ui = u[i];
      LsapKey best{INFINITY, -1, 0};
      for (int it = tid; it < nrem; it += nt) {
        const int j = remaining[it];
        const double r = minval + cost(i, j) - ui - v[j];
        double spj = sp[j];
        if (r < spj) { path[j] = i; sp[j] = r; spj = r; }
        LsapKey k{spj, it, row4col[j] == -1 ? 1 : 0};
        if (spj < INFINITY && lsap_better(k, best)) best = k;
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
        SR[i] = 1;
        if (b.it < 0) { s_fail = 1; s_sink = -2; }
        else {
          s_minval = b.val;
          const int j = remaining[b.it];
          if (row4col[j] == -1) s_sink = j; else s_i = row4col[j];
          SC[j] = 1;
          remaining[b.it] = remaining[nrem - 1];
          s_nrem = nrem - 1;
        }
      }
      __syncthreads();
      if (s_sink != -1) break;
    }
    if (s_fail) break;
    const double minval = s_minval;
    for (int i = tid; i < N; i += nt) {
      if (i == cur) u[i] += minval;
      else if (SR[i]) u[i] += minval - sp[col4row[i]];
    }
    for (int j = tid; j < M; j += nt) if (SC[j]) v[j] -= minval - sp[j];
    __syncthreads();
    if (tid == 0) {
      int j = s_sink;
      while (true) {
        const int i = path[j];
        row4col[j] = i;
        const int t = col4row[i]; col4row[i] = j; j = t;
        if (i == cur) break;
      }
    }
    __syncthreads();
  }
  if (s_fail) { if (tid == 0) *a.n_match = -1; return; }
  if (!T) {
    for (int i = tid; i < N; i += nt) { a.row_ind[i] = rowlist[i]; a.col_ind[i] = col4row[i]; }
  } else {      // problem rows are the original columns: emit pairs sorted by original row
    const int cnt = lsap_compact(M, [&](int k) { return row4col[k] != -1; }, tmp, s_wave, &s_base);
    for (int q = tid; q < cnt; q += nt) { const int k = tmp[q]; a.row_ind[q] = rowlist[k]; a.col_ind[q] = row4col[k]; }
  }
  if (tid == 0) *a.n_match = N;
}

// ---------------------------------------------------------------------------------------------
// Adam over ONE flat parameter buffer (a16: torch.optim.Adam as the reference trainer uses it, volsdf_train.py:177,
// defaults amsgrad=False, weight_decay=0, maximize=False) -- one launch for the 1 219 274 parameters instead of torch's
// multi-tensor kernels over 65 tensors.  Arithmetic in torch's order:
//   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g g ; p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
// ---------------------------------------------------------------------------------------------
constexpr int ADAM_MAXSEG = 96;
struct AdamSegs {                 // gradient of segment s covers [off[s], off[s+1]); per-segment bias corrections (torch counts steps per tensor)
  const float* g[ADAM_MAXSEG]; long long off[ADAM_MAXSEG + 1]; float lr_over_bc1[ADAM_MAXSEG], inv_sqrt_bc2[ADAM_MAXSEG]; int nseg;
};

__global__ void adam_flat_kernel(float* __restrict__ p, AdamSegs segs, float* __restrict__ m, float* __restrict__ v,
                                 long long n, float beta1, float beta2, float eps) {
  __shared__ long long s_off[ADAM_MAXSEG + 1];
  __shared__ const float* s_g[ADAM_MAXSEG];
  __shared__ float s_a1[ADAM_MAXSEG], s_a2[ADAM_MAXSEG];
  for (int t = threadIdx.x; t <= segs.nseg; t += blockDim.x) {
    long long o = segs.off[0]; const float* gp = segs.g[0];      // compile-time kernarg indices only (a runtime index = scratch copy)
    float a1 = segs.lr_over_bc1[0], a2 = segs.inv_sqrt_bc2[0];
#pragma unroll
    for (int k = 1; k <= ADAM_MAXSEG; ++k)
      if (k == t) { o = segs.off[k]; if (k < ADAM_MAXSEG) { gp = segs.g[k]; a1 = segs.lr_over_bc1[k]; a2 = segs.inv_sqrt_bc2[k]; } }
    s_off[t] = o;
    if (t < segs.nseg) { s_g[t] = gp; s_a1[t] = a1; s_a2[t] = a2; }
  }
  __syncthreads();
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    int lo = 0, hi = segs.nseg - 1;            // segment of element i: last s with off[s] <= i
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_off[mid] <= i) lo = mid; else hi = mid - 1; }
    const float* gp = s_g[lo];
    if (!gp) continue;                         // parameter without a gradient this step: untouched, as torch.optim.Adam
    const float g = gp[i - s_off[lo]];
    const float mm = beta1 * m[i] + (1.0f - beta1) * g;
    const float vv = beta2 * v[i] + (1.0f - beta2) * g * g;
    m[i] = mm; v[i] = vv;
    p[i] -= s_a1[lo] * (mm / (sqrtf(vv) * s_a2[lo] + eps));
  }
}

}  // namespace neat

// What a persistent layer loop could buy the tangent chain (round 5): traffic-only model of seven tangent layers
// at C2's size.  Per 32-point tile and layer: read the chain variable (16 KB), read two saved operands (h, u: 16 KB each,
// non-temporal), write the next chain variable and m (16 KB each).  Variants:
//   0  seven launches (today's structure), tiles interleaved over 256 workgroups, chain arrays distinct per layer
//   1  ONE launch, each workgroup loops over the layers for its own tiles (same tile order in every layer)
//   2  as 1, tile order reversed on odd layers (the tiles written last are read first: L2 / Infinity Cache hits)
//   3  as 2, two ping-pong chain buffers instead of eight arrays
//   4  as 1 without the chain READ (what a chain variable served from on-chip memory would cost)
//   5  as 1 without chain read and write (the three streams nothing can remove)
//   6  layer PIPELINE inside each XCD: workgroup b = (XCD b & 7, pipeline (b >> 3) & 3, stage b >> 5); a stage keeps ONE layer for
//      the whole launch and hands its tiles to the next stage of its pipeline through the XCD's L2 (progress counter per edge)
//   8  as 0 with two ping-pong chain buffers;  9  as 1 with two ping-pong chain buffers
//   7  as 6 without the chain read and without waiting (the cost of the structure alone)
// hipcc --offload-arch=gfx950 -O3 probe_chain.hip -o probe_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int NL = 8, TILE_U4 = 1024;       // 16 KB tile = 1024 uint4; 512 threads x 2
struct Args {
  const uint4* h[NL]; const uint4* u[NL]; uint4* m[NL];
  uint4* chain[NL + 1];
  int ntiles, l0, l1, reverse, skip_read, skip_write;
};

__device__ __forceinline__ uint4 ldnt(const uint4* p) {
  typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
  const v4u_t v = __builtin_nontemporal_load(reinterpret_cast<const v4u_t*>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void stnt(uint4* p, uint4 v) {
  typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
  const v4u_t w = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(w, reinterpret_cast<v4u_t*>(p));
}

__global__ __launch_bounds__(512) void chain_kernel(Args a) {
  const int T = (a.ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  for (int l = a.l0; l < a.l1; ++l) {
    const uint4* cin = a.chain[l];
    uint4* cout = a.chain[l + 1];
    const bool rev = a.reverse && (l & 1);
    // two tiles per trip: 10 independent 16-byte loads per thread in flight
    for (int t0 = 0; t0 < T; t0 += 2) {
      uint4 x[2][2], hh[2][2], uu[2][2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int tt = t0 + k < T ? t0 + k : T - 1;
        const int tau = rev ? T - 1 - tt : tt;
        const size_t base = ((size_t)blockIdx.x + (size_t)tau * gridDim.x) * TILE_U4 + threadIdx.x;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          x[k][i] = a.skip_read ? make_uint4(1, 2, 3, 4) : cin[base + i * 512];
          hh[k][i] = ldnt(a.h[l] + base + i * 512);
          uu[k][i] = ldnt(a.u[l] + base + i * 512);
        }
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        if (t0 + k >= T) break;
        const int tau = rev ? T - 1 - (t0 + k) : t0 + k;
        const size_t base = ((size_t)blockIdx.x + (size_t)tau * gridDim.x) * TILE_U4 + threadIdx.x;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          uint4 v = x[k][i];
          v.x += hh[k][i].x ^ uu[k][i].y; v.y ^= hh[k][i].z; v.z += uu[k][i].w; v.w ^= hh[k][i].w + uu[k][i].x;
          if (!a.skip_write) cout[base + i * 512] = v;
          uint4 w = v; w.x ^= 0x55u;
          stnt(a.m[l] + base + i * 512, w);
        }
      }
    }
    if (l + 1 < a.l1) {       // the workgroup's own stores must be visible to its own loads of the next layer
      __builtin_amdgcn_s_waitcnt(0);
      __syncthreads();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
  }
}


struct PipeArgs { Args a; unsigned* progress; unsigned* err; int no_wait; };
__global__ __launch_bounds__(512) void pipe_kernel(PipeArgs pa) {
  const Args& a = pa.a;
  const int b = blockIdx.x, xcd = b & 7, pipe = (b >> 3) & 3, stage = b >> 5;       // 8 stages x 4 pipelines x 8 XCDs = 256
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (threadIdx.x == 0 && (int)(xcc & 15) != xcd) atomicAdd(pa.err + 1, 1u);          // placement differs from b % 8 (speed only)
  const int per_xcd = a.ntiles / 8;                  // 520
  const int T = (per_xcd - pipe + 3) / 4;
  unsigned* my_prog = pa.progress + ((xcd * 4 + pipe) * 8 + stage) * 32;              // one 128-byte line per counter
  const unsigned* in_prog = pa.progress + ((xcd * 4 + pipe) * 8 + stage - 1) * 32;
  const int l = stage;
  const uint4* cin = a.chain[l];
  uint4* cout = a.chain[l + 1];
  __shared__ unsigned seen;
  for (int t0 = 0; t0 < T; t0 += 2) {
    const int nt = T - t0 < 2 ? T - t0 : 2;
    if (stage > 0 && !pa.no_wait) {
      if (threadIdx.x == 0) {
        unsigned v = 0; int spins = 0;
        while ((v = __hip_atomic_load(in_prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < (unsigned)(t0 + nt)) {
          __builtin_amdgcn_s_sleep(2);
          if (++spins > (1 << 22)) { atomicAdd(pa.err, 1u); break; }
        }
        seen = v;
      }
      __syncthreads();
    }
    uint4 x[2][2], hh[2][2], uu[2][2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int tt = k < nt ? t0 + k : t0;
      const size_t base = ((size_t)xcd * per_xcd + pipe + 4 * (size_t)tt) * TILE_U4 + threadIdx.x;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        x[k][i] = (a.skip_read || stage == 0) ? make_uint4(1, 2, 3, 4) : ldnt(cin + base + i * 512);
        hh[k][i] = ldnt(a.h[l] + base + i * 512);
        uu[k][i] = ldnt(a.u[l] + base + i * 512);
      }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (k >= nt) break;
      const size_t base = ((size_t)xcd * per_xcd + pipe + 4 * (size_t)(t0 + k)) * TILE_U4 + threadIdx.x;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        uint4 v = x[k][i];
        v.x += hh[k][i].x ^ uu[k][i].y; v.y ^= hh[k][i].z; v.z += uu[k][i].w; v.w ^= hh[k][i].w + uu[k][i].x;
        cout[base + i * 512] = v;
        uint4 w = v; w.x ^= 0x55u;
        stnt(a.m[l] + base + i * 512, w);
      }
    }
    // publish: every wave's stores are in L2 (vmcnt 0), then one lane bumps the counter
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(my_prog, (unsigned)(t0 + nt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

int main() {
  const int ntiles = 133120 / 32;
  const size_t n = (size_t)ntiles * TILE_U4;     // uint4 per array (68 MB)
  std::vector<uint4*> bufs(3 * NL + NL + 1);
  for (auto& b : bufs) { hipMalloc(&b, n * 16); hipMemset(b, 1, n * 16); }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  unsigned* prog; unsigned* err;
  hipMalloc(&prog, 8 * 4 * 8 * 32 * 4); hipMalloc(&err, 8); hipMemset(err, 0, 8);
  for (int variant = 0; variant < 10; ++variant) {
    Args a{};
    for (int l = 0; l < NL; ++l) { a.h[l] = bufs[l]; a.u[l] = bufs[NL + l]; a.m[l] = bufs[2 * NL + l]; }
    for (int l = 0; l <= NL; ++l) a.chain[l] = bufs[3 * NL + ((variant == 3 || variant >= 8) ? (l & 1) : l)];
    a.ntiles = ntiles; a.reverse = variant == 2 || variant == 3; a.skip_read = variant >= 4 && variant < 8; a.skip_write = variant == 5;
    auto go = [&]() {
      if (variant == 0 || variant == 8)
        for (int l = 0; l < NL; ++l) { a.l0 = l; a.l1 = l + 1; hipLaunchKernelGGL(chain_kernel, dim3(256), dim3(512), 0, 0, a); }
      else if (variant == 6 || variant == 7) {
        PipeArgs pa{a, prog, err, variant == 7};
        pa.a.skip_read = variant == 7;
        hipMemsetAsync(prog, 0, 8 * 4 * 8 * 32 * 4, 0);
        hipLaunchKernelGGL(pipe_kernel, dim3(256), dim3(512), 0, 0, pa);
      }
      else { a.l0 = 0; a.l1 = NL; hipLaunchKernelGGL(chain_kernel, dim3(256), dim3(512), 0, 0, a); }
    };
    for (int i = 0; i < 3; ++i) go();
    hipEventRecord(e0, 0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) go();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms / reps * 1e3;
    const double streams = (variant == 4 || variant == 7) ? 4 : (variant == 5 ? 3 : 5);
    printf("variant %d: %7.1f us per chain (%5.1f per layer), %5.0f GB/s over %g streams\n", variant, us, us / NL,
           streams * NL * n * 16.0 / (us * 1e-6) / 1e9, streams);
  }
  unsigned herr[2]; hipMemcpy(herr, err, 8, hipMemcpyDeviceToHost);
  printf("pipeline: %u spin time-outs, %u workgroups not on XCD b %% 8 (summed over %d launches)\n", herr[0], herr[1], 2 * 23);
  return 0;
}

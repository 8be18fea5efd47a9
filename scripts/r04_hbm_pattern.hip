// Diagnostic (r04): what does HBM deliver for phase 1's access pattern when nothing is cached?  256 workgroups, one utterance of
// T rows of 1 KiB each; 8 wavefronts; rows are read ends -> middle (two sides) as the kernels do.  Four 262 MB tensors in
// rotation, so no launch finds its rows in the Infinity Cache (256 MB).  Variants:
//   base       the kernels' pattern (side 0 ascending from row 0, side 1 descending from row T-1, all workgroups in step)
//   fwd        both halves ascending (is the descending stream the problem?)
//   skew       workgroup b starts SKEW(b) rows into its half and wraps (are the workgroups' aligned positions the problem?)
//   stream     a grid-stride read of the same bytes (the box's read rate)
//   build: hipcc -O3 --offload-arch=gfx950 scripts/r04_hbm_pattern.hip -o scratch/r04_hbm_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void pat_kernel(const float4 *__restrict__ x, int T, int skewmul, float *__restrict__ out) {
  const int b = blockIdx.x;
  const int side = threadIdx.x >> 8;
  const int w = (threadIdx.x >> 6) & 3, lane = threadIdx.x & 63;
  const float4 *xb = x + (long)b * T * 64;
  const int tm = T / 2;
  const int n = side == 0 ? tm : T - tm;
  const int skew = (MODE == 2) ? (int)(((long)b * skewmul) % n) : 0;
  float acc = 0.f;
  for (int s = 0; s < n; s += 4 * DEPTH) {
    float4 v[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      int r = s + 4 * d + w;
      r = r < n ? r : n - 1;
      r += skew; r = r >= n ? r - n : r;
      int t;
      if (MODE == 1) t = side == 0 ? r : tm + r;
      else t = side == 0 ? r : T - 1 - r;
      v[d] = xb[(long)t * 64 + lane];
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) acc += v[d].x;
  }
  if (acc == 123.456f) out[0] = acc;
}

// two tensors (the Hessian-vector product's phase 1: logits and vector rows of the same frames); BAR: a workgroup barrier every
// BAR rows per side (0: none), as the kernels' block barrier
template <int DEPTH, int BAR>
__global__ __launch_bounds__(512) void pat2_kernel(const float4 *__restrict__ x, const float4 *__restrict__ y, int T, float *__restrict__ out) {
  const int b = blockIdx.x;
  const int side = threadIdx.x >> 8;
  const int w = (threadIdx.x >> 6) & 3, lane = threadIdx.x & 63;
  const float4 *xb = x + (long)b * T * 64, *yb = y + (long)b * T * 64;
  const int tm = T / 2;
  const int n = tm;  // (T even here)
  float acc = 0.f;
  for (int s = 0; s < n; s += 4 * DEPTH) {
    float4 v[DEPTH], u[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      int r = s + 4 * d + w;
      r = r < n ? r : n - 1;
      const int t = side == 0 ? r : T - 1 - r;
      v[d] = xb[(long)t * 64 + lane];
      u[d] = yb[(long)t * 64 + lane];
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) acc += v[d].x + u[d].x;
    if (BAR) __syncthreads();
  }
  if (acc == 123.456f) out[0] = acc;
}

// the Hessian-vector product's phase 1 as its wavefronts issue it: 10 wavefronts (2 idle: the main chains), blocks of BLK rows per
// side, four loading wavefronts a side taking rows (0,1), (2,3), (4), (5) of every block (NQ = 2, 2, 1, 1) of both tensors, a
// ring of PFD blocks in registers, one workgroup barrier per block
template <int BLK, int PFD, int LDSKB>
__global__ __launch_bounds__(640) void pat3_kernel(const float4 *__restrict__ x, const float4 *__restrict__ y, int T, float *__restrict__ out) {
  __shared__ float lds[LDSKB * 256 + 1];
  const int b = blockIdx.x;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float4 *xb = x + (long)b * T * 64, *yb = y + (long)b * T * 64;
  const int tm = T / 2, nblk = (tm + BLK - 1) / BLK;
  float acc = 0.f;
  if (threadIdx.x == 0) lds[LDSKB * 256] = 0.f;
  const int side = w & 1, slot = (w - 2) >> 1;          // w = 2..9: loaders
  const int p0 = slot == 0 ? 0 : slot == 1 ? 2 : slot == 2 ? 4 : 5, nq = slot < 2 ? 2 : 1;
  auto row = [&](int j, int d) -> int {
    int r = j * BLK + d;
    r = r < tm ? r : tm - 1;
    return side == 0 ? r : T - 1 - r;
  };
  float4 vx[PFD][2], vy[PFD][2];
  if (w >= 2)
    for (int r = 0; r < PFD; ++r)
      for (int q = 0; q < 2; ++q) { vx[r][q] = xb[(long)row(r, p0 + (q < nq ? q : 0)) * 64 + lane]; vy[r][q] = yb[(long)row(r, p0 + (q < nq ? q : 0)) * 64 + lane]; }
  const int ni = (nblk + PFD - 1) / PFD * PFD;
  for (int j0 = 0; j0 < ni; j0 += PFD) {
#pragma unroll
    for (int r = 0; r < PFD; ++r) {
      const int j = j0 + r;
      if (w >= 2) {
#pragma unroll
        for (int q = 0; q < 2; ++q) acc += vx[r][q].x + vy[r][q].x;
#pragma unroll
        for (int q = 0; q < 2; ++q) { vx[r][q] = xb[(long)row(j + PFD, p0 + (q < nq ? q : 0)) * 64 + lane]; vy[r][q] = yb[(long)row(j + PFD, p0 + (q < nq ? q : 0)) * 64 + lane]; }
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_s_barrier();
    }
  }
  if (acc == 123.456f) out[0] = acc + lds[LDSKB * 256];
}

__global__ __launch_bounds__(512) void stream_kernel(const float4 *__restrict__ src, long n, float *__restrict__ out) {
  float acc = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) acc += src[i].x;
  if (acc == 123.456f) out[0] = acc;
}

int main() {
  const int Bn = 256, T = 1000, NBUF = 4;
  const long n4 = (long)Bn * T * 64;
  float4 *A[NBUF];
  float *out;
  for (int i = 0; i < NBUF; ++i) { CK(hipMalloc(&A[i], n4 * 16)); CK(hipMemset(A[i], 0, n4 * 16)); }
  CK(hipMalloc(&out, 64));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto run = [&](const char *name, auto launch) -> int {
    std::vector<float> t;
    for (int i = 0; i < 40; ++i) {
      CK(hipEventRecord(a, st));
      launch(A[i % NBUF]);
      CK(hipEventRecord(b, st));
      CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      if (i >= 8) t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    printf("%-28s median %7.1f us  min %7.1f us  %6.0f GB/s\n", name, t[t.size() / 2] * 1e3, t[0] * 1e3, n4 * 16 / t[t.size() / 2] / 1e6);
    fflush(stdout);
    return 0;
  };
  run("stream grid 2048", [&](float4 *p) { hipLaunchKernelGGL(stream_kernel, dim3(2048), dim3(512), 0, st, p, n4, out); });
  run("stream grid 256", [&](float4 *p) { hipLaunchKernelGGL(stream_kernel, dim3(256), dim3(512), 0, st, p, n4, out); });
  run("base d4", [&](float4 *p) { hipLaunchKernelGGL((pat_kernel<0, 4>), dim3(Bn), dim3(512), 0, st, p, T, 0, out); });
  run("base d8", [&](float4 *p) { hipLaunchKernelGGL((pat_kernel<0, 8>), dim3(Bn), dim3(512), 0, st, p, T, 0, out); });
  run("base d16", [&](float4 *p) { hipLaunchKernelGGL((pat_kernel<0, 16>), dim3(Bn), dim3(512), 0, st, p, T, 0, out); });
  run("fwd d4", [&](float4 *p) { hipLaunchKernelGGL((pat_kernel<1, 4>), dim3(Bn), dim3(512), 0, st, p, T, 0, out); });
  run("fwd d8", [&](float4 *p) { hipLaunchKernelGGL((pat_kernel<1, 8>), dim3(Bn), dim3(512), 0, st, p, T, 0, out); });
  run("skew 37 d4", [&](float4 *p) { hipLaunchKernelGGL((pat_kernel<2, 4>), dim3(Bn), dim3(512), 0, st, p, T, 37, out); });
  run("skew 37 d8", [&](float4 *p) { hipLaunchKernelGGL((pat_kernel<2, 8>), dim3(Bn), dim3(512), 0, st, p, T, 37, out); });
  run("skew 1 d4", [&](float4 *p) { hipLaunchKernelGGL((pat_kernel<2, 4>), dim3(Bn), dim3(512), 0, st, p, T, 1, out); });
  run("skew 2 d4", [&](float4 *p) { hipLaunchKernelGGL((pat_kernel<2, 4>), dim3(Bn), dim3(512), 0, st, p, T, 2, out); });
  run("skew 16 d4", [&](float4 *p) { hipLaunchKernelGGL((pat_kernel<2, 4>), dim3(Bn), dim3(512), 0, st, p, T, 16, out); });
  run("base d4 again", [&](float4 *p) { hipLaunchKernelGGL((pat_kernel<0, 4>), dim3(Bn), dim3(512), 0, st, p, T, 0, out); });
  run("two tensors d2 (4 bufs: 2+2)", [&](float4 *p) { hipLaunchKernelGGL((pat2_kernel<2, 0>), dim3(Bn), dim3(512), 0, st, p, p == A[0] ? A[1] : (p == A[1] ? A[2] : (p == A[2] ? A[3] : A[0])), T, out); });
  run("two tensors d4", [&](float4 *p) { hipLaunchKernelGGL((pat2_kernel<4, 0>), dim3(Bn), dim3(512), 0, st, p, p == A[0] ? A[1] : (p == A[1] ? A[2] : (p == A[2] ? A[3] : A[0])), T, out); });
  run("two tensors d2 barrier", [&](float4 *p) { hipLaunchKernelGGL((pat2_kernel<2, 1>), dim3(Bn), dim3(512), 0, st, p, p == A[0] ? A[1] : (p == A[1] ? A[2] : (p == A[2] ? A[3] : A[0])), T, out); });
  run("two tensors d1 barrier", [&](float4 *p) { hipLaunchKernelGGL((pat2_kernel<1, 1>), dim3(Bn), dim3(512), 0, st, p, p == A[0] ? A[1] : (p == A[1] ? A[2] : (p == A[2] ? A[3] : A[0])), T, out); });
#define NEXT(p) (p == A[0] ? A[1] : (p == A[1] ? A[2] : (p == A[2] ? A[3] : A[0])))
  run("hvp phase 1: blk6 pfd4", [&](float4 *p) { hipLaunchKernelGGL((pat3_kernel<6, 4, 1>), dim3(Bn), dim3(640), 0, st, p, NEXT(p), T, out); });
  run("hvp phase 1: blk6 pfd4 lds150", [&](float4 *p) { hipLaunchKernelGGL((pat3_kernel<6, 4, 150>), dim3(Bn), dim3(640), 0, st, p, NEXT(p), T, out); });
  run("hvp phase 1: blk6 pfd2", [&](float4 *p) { hipLaunchKernelGGL((pat3_kernel<6, 2, 1>), dim3(Bn), dim3(640), 0, st, p, NEXT(p), T, out); });
  run("hvp phase 1: blk12 pfd2", [&](float4 *p) { hipLaunchKernelGGL((pat3_kernel<12, 2, 1>), dim3(Bn), dim3(640), 0, st, p, NEXT(p), T, out); });
  // the same buffer every launch (what the bench's headline protocol sees)
  {
    std::vector<float> t;
    for (int i = 0; i < 40; ++i) {
      CK(hipEventRecord(a, st));
      hipLaunchKernelGGL((pat_kernel<0, 4>), dim3(Bn), dim3(512), 0, st, A[0], T, 0, out);
      CK(hipEventRecord(b, st));
      CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      if (i >= 8) t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    printf("%-28s median %7.1f us  min %7.1f us  %6.0f GB/s\n", "base d4, one buffer", t[t.size() / 2] * 1e3, t[0] * 1e3, n4 * 16 / t[t.size() / 2] / 1e6);
  }
  return 0;
}

// Memory-system probe for the north-star traffic pattern (diagnostic, not part of the product).
//   build: hipcc -O3 --offload-arch=gfx950 scripts/r03_memprobe.hip -o scratch/r03_memprobe
//   run  : scratch/r03_memprobe > gpurun_out/r03_memprobe.json
// Measures on the box it runs on:
//   copy      1 GiB float4 copy, plain / nt stores                              (the box's copy ceiling)
//   reread    a buffer of S MiB read 8x back to back, GB/s of the last 6 passes (Infinity Cache capacity and bandwidth)
//   twopass   the loss+gradient kernel's traffic with NO arithmetic: 256 workgroups, each owning one utterance of T rows
//             of 1 KiB; pass 1 reads the rows ends -> middle (both sides at once), pass 2 re-reads them middle -> ends and
//             writes a 1 KiB gradient row per row read.  Variants: load / store cache policies.  This is the memory floor of
//             ANY kernel with that access pattern on this box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <functional>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
template <int NT> __device__ __forceinline__ float4 ld(const float4 *p) {
  if constexpr (NT) { const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p)); return make_float4(t.x, t.y, t.z, t.w); }
  else return *p;
}
template <int NT> __device__ __forceinline__ void st(float4 *p, float4 v) {
  if constexpr (NT) { v4f t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w; __builtin_nontemporal_store(t, reinterpret_cast<v4f *>(p)); }
  else *p = v;
}

template <int LNT, int SNT>
__global__ __launch_bounds__(512) void copy_kernel(float4 *__restrict__ dst, const float4 *__restrict__ src, long n) {
  long i = (long)blockIdx.x * 512 + threadIdx.x;
  const long stride = (long)gridDim.x * 512;
  for (; i + 3 * stride < n; i += 4 * stride) {
    float4 a = ld<LNT>(src + i), b = ld<LNT>(src + i + stride), c = ld<LNT>(src + i + 2 * stride), d = ld<LNT>(src + i + 3 * stride);
    st<SNT>(dst + i, a); st<SNT>(dst + i + stride, b); st<SNT>(dst + i + 2 * stride, c); st<SNT>(dst + i + 3 * stride, d);
  }
  for (; i < n; i += stride) st<SNT>(dst + i, ld<LNT>(src + i));
}

template <int LNT>
__global__ __launch_bounds__(512) void read_kernel(const float4 *__restrict__ src, long n, float *__restrict__ out) {
  long i = (long)blockIdx.x * 512 + threadIdx.x;
  const long stride = (long)gridDim.x * 512;
  float acc = 0.f;
  for (; i + 3 * stride < n; i += 4 * stride) {
    float4 a = ld<LNT>(src + i), b = ld<LNT>(src + i + stride), c = ld<LNT>(src + i + 2 * stride), d = ld<LNT>(src + i + 3 * stride);
    acc += a.x + b.y + c.z + d.w;
  }
  for (; i < n; i += stride) acc += ld<LNT>(src + i).x;
  if (acc == 123.456f) out[0] = acc;
}

// variant: every workgroup owns one contiguous chunk, UNR loads in flight per lane
template <int UNR, int SNT>
__global__ __launch_bounds__(512) void copy_chunk_kernel(float4 *__restrict__ dst, const float4 *__restrict__ src, long n) {
  const long per = n / gridDim.x;
  const float4 *s = src + per * blockIdx.x;
  float4 *d = dst + per * blockIdx.x;
  long i = threadIdx.x;
  for (; i + (UNR - 1) * 512 < per; i += UNR * 512) {
    float4 v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) v[u] = s[i + u * 512];
#pragma unroll
    for (int u = 0; u < UNR; ++u) st<SNT>(d + i + u * 512, v[u]);
  }
  for (; i < per; i += 512) st<SNT>(d + i, s[i]);
}

template <int SNT>
__global__ __launch_bounds__(512) void fill_kernel(float4 *__restrict__ dst, long n, float v) {
  long i = (long)blockIdx.x * 512 + threadIdx.x;
  const long stride = (long)gridDim.x * 512;
  for (; i < n; i += stride) st<SNT>(dst + i, make_float4(v, v, v, v));
}

// One workgroup of 512 threads per utterance: threads 0..255 are side A, 256..511 side B; a row is 64 float4 = one wavefront's
// access; each side moves 4 rows per step (its 4 wavefronts), DEPTH steps of loads in flight.
//   pass 1: A reads rows 0 .. tm-1 ascending, B reads rows T-1 .. tm descending
//   pass 2: A reads rows tm .. T-1 ascending, B reads rows tm-1 .. 0 descending; every row read is followed by a row written
// PASSES: 1 = pass 1 only, 2 = pass 2 only, 3 = both (one launch, a workgroup barrier in between, as in fused6_kernel)
template <int L1NT, int L2NT, int SNT, int DEPTH>
__global__ __launch_bounds__(512) void twopass_kernel(const float4 *__restrict__ x, float4 *__restrict__ g, int T, int passes, float *__restrict__ out) {
  const int b = blockIdx.x;
  const int side = threadIdx.x >> 8;
  const int w = (threadIdx.x >> 6) & 3, lane = threadIdx.x & 63;
  const float4 *xb = x + (long)b * T * 64;
  float4 *gb = g + (long)b * T * 64;
  const int tm = T / 2;
  float acc = 0.f;
  if (passes & 1) {
    const int n = side == 0 ? tm : T - tm;
    for (int s = 0; s < n; s += 4 * DEPTH) {
      float4 v[DEPTH];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        int r = s + 4 * d + w;
        r = r < n ? r : n - 1;
        const int t = side == 0 ? r : T - 1 - r;
        v[d] = ld<L1NT>(xb + (long)t * 64 + lane);
      }
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) acc += v[d].x;
    }
  }
  __syncthreads();
  if (passes & 2) {
    const int n = side == 0 ? T - tm : tm;
    for (int s = 0; s < n; s += 4 * DEPTH) {
      float4 v[DEPTH];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        int r = s + 4 * d + w;
        r = r < n ? r : n - 1;
        const int t = side == 0 ? tm + r : tm - 1 - r;
        v[d] = ld<L2NT>(xb + (long)t * 64 + lane);
      }
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        int r = s + 4 * d + w;
        r = r < n ? r : n - 1;
        const int t = side == 0 ? tm + r : tm - 1 - r;
        st<SNT>(gb + (long)t * 64 + lane, make_float4(v[d].x + 1.f, v[d].y, v[d].z, v[d].w));
      }
    }
  }
  if (acc == 123.456f) out[0] = acc;
}

static float time_ms(hipStream_t st, int reps, int skip, const std::function<void()> &fn, float *minp = nullptr) {
  std::vector<float> t;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < reps; ++i) {
    CK(hipEventRecord(a, st));
    fn();
    CK(hipEventRecord(b, st));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (i >= skip) t.push_back(ms);
  }
  std::sort(t.begin(), t.end());
  if (minp) *minp = t.front();
  CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
  return t[t.size() / 2];
}

int main(int argc, char **argv) {
  hipStream_t st;
  CK(hipStreamCreate(&st));
  const long GiB = 1L << 30;
  float4 *A, *Bf;
  float *out;
  CK(hipMalloc(&A, GiB)); CK(hipMalloc(&Bf, GiB)); CK(hipMalloc(&out, 64));
  CK(hipMemset(A, 0, GiB)); CK(hipMemset(Bf, 0, GiB));
  printf("{\n");
  // ---- copy ceiling ----
  {
    const long n = GiB / 16;
    float mn;
    float m0 = time_ms(st, 12, 2, [&] { hipLaunchKernelGGL((copy_kernel<0, 0>), dim3(2048), dim3(512), 0, st, Bf, A, n); }, &mn);
    printf(" \"copy_1GiB_plain_GBps\": %.1f, \"copy_1GiB_plain_best_GBps\": %.1f,\n", 2.0 * GiB / m0 / 1e6, 2.0 * GiB / mn / 1e6);
    float m1 = time_ms(st, 12, 2, [&] { hipLaunchKernelGGL((copy_kernel<0, 1>), dim3(2048), dim3(512), 0, st, Bf, A, n); }, &mn);
    printf(" \"copy_1GiB_ntstore_GBps\": %.1f,\n", 2.0 * GiB / m1 / 1e6);
    float m2 = time_ms(st, 12, 2, [&] { hipLaunchKernelGGL((copy_kernel<1, 1>), dim3(2048), dim3(512), 0, st, Bf, A, n); }, &mn);
    printf(" \"copy_1GiB_ntload_ntstore_GBps\": %.1f,\n", 2.0 * GiB / m2 / 1e6);
    float m3 = time_ms(st, 12, 2, [&] { hipLaunchKernelGGL((fill_kernel<0>), dim3(2048), dim3(512), 0, st, Bf, n, 1.f); }, &mn);
    printf(" \"fill_1GiB_plain_GBps\": %.1f,\n", 1.0 * GiB / m3 / 1e6);
    float m4 = time_ms(st, 12, 2, [&] { hipLaunchKernelGGL((fill_kernel<1>), dim3(2048), dim3(512), 0, st, Bf, n, 1.f); }, &mn);
    printf(" \"fill_1GiB_nt_GBps\": %.1f,\n", 1.0 * GiB / m4 / 1e6);
    float m5 = time_ms(st, 12, 2, [&] { hipLaunchKernelGGL((read_kernel<0>), dim3(2048), dim3(512), 0, st, A, n, out); }, &mn);
    printf(" \"read_1GiB_plain_GBps\": %.1f,\n", 1.0 * GiB / m5 / 1e6);
    const int grids[] = {512, 1024, 2048, 4096, 8192, 16384};
    printf(" \"copy_1GiB_chunked_ntstore_GBps\": {");
    for (int g = 0; g < 6; ++g) {
      float a4 = time_ms(st, 8, 2, [&] { hipLaunchKernelGGL((copy_chunk_kernel<4, 1>), dim3(grids[g]), dim3(512), 0, st, Bf, A, n); }, &mn);
      float a8 = time_ms(st, 8, 2, [&] { hipLaunchKernelGGL((copy_chunk_kernel<8, 1>), dim3(grids[g]), dim3(512), 0, st, Bf, A, n); }, &mn);
      printf("%s\"grid%d_unr4\": %.0f, \"grid%d_unr8\": %.0f", g ? ", " : "", grids[g], 2.0 * GiB / a4 / 1e6, grids[g], 2.0 * GiB / a8 / 1e6);
    }
    printf("},\n");
    float m6 = time_ms(st, 8, 2, [&] { CK(hipMemcpyAsync(Bf, A, GiB, hipMemcpyDeviceToDevice, st)); }, &mn);
    printf(" \"hipMemcpyDtoD_1GiB_GBps\": %.1f,\n", 2.0 * GiB / m6 / 1e6);
    float m7 = time_ms(st, 8, 2, [&] { CK(hipMemsetAsync(Bf, 0, GiB, st)); }, &mn);
    printf(" \"hipMemset_1GiB_GBps\": %.1f,\n", 1.0 * GiB / m7 / 1e6);
  }
  // ---- re-read bandwidth against footprint ----
  printf(" \"reread_GBps_by_MiB\": {");
  const int sizes[] = {16, 32, 64, 128, 192, 224, 240, 256, 288, 320, 384, 512, 1024};
  for (int k = 0; k < (int)(sizeof(sizes) / sizeof(int)); ++k) {
    const long bytes = (long)sizes[k] << 20, n = bytes / 16;
    float mn;
    float m = time_ms(st, 10, 3, [&] { hipLaunchKernelGGL((read_kernel<0>), dim3(2048), dim3(512), 0, st, A, n, out); }, &mn);
    printf("%s\"%d\": %.0f", k ? ", " : "", sizes[k], bytes / m / 1e6);
  }
  printf("},\n");
  // ---- the two-pass pattern ----
  const int Bn = 256, T = 1000;
  const double MB1 = (double)Bn * T * 1024;  // one tensor: 262 MB
  auto run2 = [&](const char *name, auto kern, int passes, double bytes) {
    float mn;
    float m = time_ms(st, 24, 4, [&] { hipLaunchKernelGGL(kern, dim3(Bn), dim3(512), 0, st, A, Bf, T, passes, out); }, &mn);
    printf(" \"%s\": {\"median_us\": %.1f, \"min_us\": %.1f, \"GBps\": %.0f},\n", name, m * 1e3, mn * 1e3, bytes / m / 1e6);
  };
  run2("twopass_plain_plain_plain_d4", twopass_kernel<0, 0, 0, 4>, 3, 3 * MB1);
  run2("twopass_plain_plain_ntstore_d4", twopass_kernel<0, 0, 1, 4>, 3, 3 * MB1);
  run2("twopass_plain_nt_ntstore_d4", twopass_kernel<0, 1, 1, 4>, 3, 3 * MB1);
  run2("twopass_nt_nt_ntstore_d4", twopass_kernel<1, 1, 1, 4>, 3, 3 * MB1);
  run2("twopass_nt_plain_ntstore_d4", twopass_kernel<1, 0, 1, 4>, 3, 3 * MB1);
  run2("twopass_plain_plain_ntstore_d8", twopass_kernel<0, 0, 1, 8>, 3, 3 * MB1);
  run2("twopass_plain_plain_ntstore_d2", twopass_kernel<0, 0, 1, 2>, 3, 3 * MB1);
  run2("twopass_plain_plain_ntstore_d1", twopass_kernel<0, 0, 1, 1>, 3, 3 * MB1);
  run2("pass1_only_plain_d4", twopass_kernel<0, 0, 1, 4>, 1, MB1);
  run2("pass2_only_plain_ntstore_d4", twopass_kernel<0, 0, 1, 4>, 2, 2 * MB1);
  run2("pass2_only_plain_plainstore_d4", twopass_kernel<0, 0, 0, 4>, 2, 2 * MB1);
  // pass 1 then pass 2 as two launches (what loss-only + grad_resume do), pass 2 timed alone after a warm pass 1 / after a 512 MiB fill
  {
    std::vector<float> warm, cold;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float4 *F;
    CK(hipMalloc(&F, GiB / 2));
    for (int i = 0; i < 12; ++i) {
      for (int c = 0; c < 2; ++c) {
        hipLaunchKernelGGL((twopass_kernel<0, 0, 1, 4>), dim3(Bn), dim3(512), 0, st, A, Bf, T, 1, out);
        if (c) hipLaunchKernelGGL((fill_kernel<0>), dim3(2048), dim3(512), 0, st, F, GiB / 2 / 16, 2.f);
        CK(hipEventRecord(a, st));
        hipLaunchKernelGGL((twopass_kernel<0, 0, 1, 4>), dim3(Bn), dim3(512), 0, st, A, Bf, T, 2, out);
        CK(hipEventRecord(b, st));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (i >= 2) (c ? cold : warm).push_back(ms);
      }
    }
    std::sort(warm.begin(), warm.end()); std::sort(cold.begin(), cold.end());
    printf(" \"pass2_after_pass1_warm_us\": %.1f, \"pass2_after_512MiB_fill_us\": %.1f,\n", warm[warm.size() / 2] * 1e3, cold[cold.size() / 2] * 1e3);
  }
  printf(" \"note\": \"twopass bytes = 3 x 262.1 MB (two reads of the logits, one write of the gradient); names: <pass-1 loads>_<pass-2 loads>_<stores>_d<loads in flight per lane>\"\n}\n");
  return 0;
}

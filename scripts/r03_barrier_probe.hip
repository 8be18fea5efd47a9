// diagnostic: stamp-to-stamp time across a 12-wavefront barrier (as F6_BARRIER measures it), with and without imbalance
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(768) void k(int n, int spin, long long *out, float *sink) {
  float x = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  long long acc = 0, accw = 0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) {
    const int my = (w == (i % 12)) ? spin : spin / 4;   // a different wavefront is last every iteration
    for (int s = 0; s < my; ++s) x = x * 1.0001f + 1.0f;
    __builtin_amdgcn_s_waitcnt(0xC07F);
    unsigned long long t = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
    unsigned long long u = __builtin_amdgcn_s_memtime();
    if (w == (i % 12)) acc += (long long)(u - t);   // this wavefront was the last to arrive: pure barrier overhead
    accw += (long long)(u - t);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { out[w * 3] = acc; out[w * 3 + 1] = accw; out[w * 3 + 2] = t1 - t0; }
  if (x == 12345.f) sink[0] = x;
}
int main() {
  long long *out; float *sink;
  (void)hipMalloc(&out, 12 * 24); (void)hipMalloc(&sink, 4);
  for (int spin : {0, 200, 1000}) {
    const int n = 1200;
    hipLaunchKernelGGL(k, dim3(256), dim3(768), 0, 0, n, spin, out, sink);
    (void)hipDeviceSynchronize();
    long long h[36]; (void)hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    printf("spin=%d: total %lld cycles / %d iterations = %.0f per iteration\n", spin, h[2], n, (double)h[2] / n);
    for (int w = 0; w < 12; ++w) printf("  wave %2d: stamp-to-stamp when last %.0f cycles; mean over all iterations %.0f\n", w, (double)h[w * 3] / (n / 12), (double)h[w * 3 + 1] / n);
  }
  return 0;
}

// diagnostic: how often ONE wavefront can issue an instruction of each kind on gfx950 (cycles per instruction, s_memtime = shader
// clock), alone on its SIMD and with 2 other wavefronts of the same workgroup on it (12 wavefronts per workgroup as fused6 runs)
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
template <int MODE>
__device__ __forceinline__ void body(float (&a)[8], float s, int &si, float *lds, int lane) {
  if constexpr (MODE == 0) { REP8(asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[0]) : "v"(s));) }                         // dependent FMA chain
  if constexpr (MODE == 1) { asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                                          "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8"
                                          : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(s)); }  // independent FMAs
  if constexpr (MODE == 2) { REP8(asm volatile("s_add_u32 %0, %0, 1" : "+s"(si));) }                                           // SALU
  if constexpr (MODE == 3) { asm volatile("v_fma_f32 %0, %0, %4, %4\n s_add_u32 %8, %8, 1\n v_fma_f32 %1, %1, %4, %4\n s_add_u32 %8, %8, 1\n"
                                          "v_fma_f32 %2, %2, %4, %4\n s_add_u32 %8, %8, 1\n v_fma_f32 %3, %3, %4, %4\n s_add_u32 %8, %8, 1"
                                          : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(s), "v"(a[4]), "v"(a[5]), "v"(a[6]), "s"(si)); }  // 4 VALU + 4 SALU interleaved
  if constexpr (MODE == 4) { REP8(asm volatile("v_exp_f32 %0, %0" : "+v"(a[0]));) }                                            // dependent transcendental
  if constexpr (MODE == 5) { asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                                          : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])); }  // independent transcendentals
  if constexpr (MODE == 6) { REP8(asm volatile("s_nop 1\n v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[0]));) }  // dependent DPP adds (with the hazard nop)
  if constexpr (MODE == 7) { REP8(asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(a[0]) : "v"(0));) }                              // dependent ldexp
  if constexpr (MODE == 8) { float *q = lds + lane; REP8(asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(a[0]) : "v"((int)(size_t)q) : "memory");) }  // LDS round trips
  if constexpr (MODE == 9) { float *q = lds + lane; asm volatile("ds_write_b32 %0, %1\n ds_write_b32 %0, %2 offset:256\n ds_write_b32 %0, %3 offset:512\n ds_write_b32 %0, %4 offset:768\n"
                                          "ds_write_b32 %0, %1 offset:1024\n ds_write_b32 %0, %2 offset:1280\n ds_write_b32 %0, %3 offset:1536\n ds_write_b32 %0, %4 offset:1792\n s_waitcnt lgkmcnt(0)"
                                          :: "v"((int)(size_t)q), "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]) : "memory"); }  // 8 independent LDS writes
  if constexpr (MODE == 10) { asm volatile("v_fma_f32 %0, %0, %4, %4\n ds_write_b32 %5, %4\n v_fma_f32 %1, %1, %4, %4\n ds_write_b32 %5, %4 offset:256\n"
                                           "v_fma_f32 %2, %2, %4, %4\n ds_write_b32 %5, %4 offset:512\n v_fma_f32 %3, %3, %4, %4\n ds_write_b32 %5, %4 offset:768"
                                           : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(s), "v"((int)(size_t)(lds + lane)) : "memory"); }  // 4 VALU + 4 LDS writes interleaved
  if constexpr (MODE == 11) { REP8(asm volatile("v_readlane_b32 %0, %1, 3\n s_nop 3\n v_mov_b32 %1, %0" : "+s"(si), "+v"(a[0]));) }   // VALU -> SGPR -> VALU round trips (2 instructions + nop each)
}
template <int MODE>
__global__ __launch_bounds__(768) void k(int n, int active_mask, long long *out, float *sink, float s) {
  __shared__ float lds[4096];
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  float a[8]; int si = 0;
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
  lds[threadIdx.x] = 0.f;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if ((active_mask >> w) & 1)
    for (int it = 0; it < n; ++it) body<MODE>(a, s, si, lds + 256 * (w & 3), lane);
  __builtin_amdgcn_s_waitcnt(0xC07F);
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  float acc = si; for (int i = 0; i < 8; ++i) acc += a[i];
  if (acc == 12345.f) sink[0] = acc;
}
template <int MODE>
void run(const char *name, int per_iter, long long *out, float *sink) {
  const int n = 4000;
  double r[3]; int masks[3] = {0x001, 0x111, 0xfff};   // wavefront 0 alone; wavefronts 0, 4, 8 = one SIMD full; all twelve
  for (int m = 0; m < 3; ++m) {
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(768), 0, 0, n, masks[m], out, sink, 0.999f); (void)hipDeviceSynchronize();
    long long h; (void)hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
    r[m] = (double)h / ((double)n * per_iter);
  }
  printf("%-58s alone %6.2f   3 on the SIMD %6.2f   12 on the CU %6.2f   cycles per instruction of wavefront 0\n", name, r[0], r[1], r[2]);
}
int main() {
  long long *out; float *sink; (void)hipMalloc(&out, 16); (void)hipMalloc(&sink, 4);
  run<0>("v_fma_f32, dependent chain", 8, out, sink);
  run<1>("v_fma_f32, 8 independent", 8, out, sink);
  run<4>("v_exp_f32, dependent chain", 8, out, sink);
  run<5>("v_exp_f32, 8 independent", 8, out, sink);
  run<6>("s_nop 1 + v_add_f32_dpp row_shr:1, dependent (per pair)", 8, out, sink);
  run<7>("v_ldexp_f32, dependent chain", 8, out, sink);
  run<8>("ds_read_b32 + wait, dependent round trips", 8, out, sink);
  run<9>("8 ds_write_b32 then wait (per write)", 8, out, sink);
  run<10>("4 v_fma + 4 ds_write interleaved (per instruction of 8)", 8, out, sink);
  run<11>("v_readlane -> s_nop 3 -> v_mov round trip (per trip)", 8, out, sink);
}

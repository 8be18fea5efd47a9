// Wave64 reductions of 2 or 4 independent values through ONE register (gfx950 lane-swap instructions).
//   v_permlane32_swap a, b : lanes 32..63 of a <-> lanes 0..31 of b      -> a op b holds two 32-lane partials side by side
//   v_permlane16_swap a, b : odd 16-lane rows of a <-> even rows of b    -> a op b holds four 16-lane partials, one per row
// then one inclusive row scan (4 DPP steps; 5 for two values) finishes all of them at once: 7 VALU operations for two
// values and 10 for four, against 12 and 24 for the per-value reductions of ctc_dpp_batch.h.
// The result of value f sits in lane SwapLanes<N>::lane(f) of the returned register.
#pragma once
#include <hip/hip_runtime.h>

namespace ctc {
namespace fused {

template <int N> struct SwapLanes;
template <> struct SwapLanes<2> { static constexpr int lane(int f) { return f == 0 ? 31 : 63; } };
template <> struct SwapLanes<4> { static constexpr int lane(int f) { return f == 0 ? 15 : f == 1 ? 47 : f == 2 ? 31 : 63; } };

template <bool MAX>
__device__ __forceinline__ float swap_op(float a, float b) {
  if constexpr (MAX) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }  // (no canonicalisation: ctc_common.h vmax_raw)
  else return a + b;
}
template <bool MAX>
__device__ __forceinline__ float fold32(float a, float b) {  // lanes 0..31: a.lo op a.hi, lanes 32..63: b.lo op b.hi
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return swap_op<MAX>(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
template <bool MAX>
__device__ __forceinline__ float fold16(float a, float b) {  // rows: a0 op a1, b0 op b1, a2 op a3, b2 op b3
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return swap_op<MAX>(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// inclusive scan inside every 16-lane row (lane 15 of a row = the row's total); PAIR: rows 1 and 3 then add the totals of
// rows 0 and 2 (lanes 31 and 63 = totals of the two 32-lane halves)
template <bool MAX, bool PAIR>
__device__ __forceinline__ void row_scan(float &v) {
  if constexpr (MAX) {
    asm("s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    if constexpr (PAIR)
      asm("v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
          "s_nop 0"
          : "+v"(v));
  } else {
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    if constexpr (PAIR)
      asm("v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
          "s_nop 0"
          : "+v"(v));
  }
}

template <int N, bool MAX>
__device__ __forceinline__ float swap_reduce(const float (&v)[N]) {
  static_assert(N == 2 || N == 4, "two or four values");
  if constexpr (N == 2) {
    float t = fold32<MAX>(v[0], v[1]);
    row_scan<MAX, true>(t);
    return t;
  } else {
    float t = fold16<MAX>(fold32<MAX>(v[0], v[1]), fold32<MAX>(v[2], v[3]));  // rows: value 0, 2, 1, 3
    row_scan<MAX, false>(t);
    return t;
  }
}

}  // namespace fused
}  // namespace ctc

// Shared device helpers and workspace layout for the gfx950 CTC kernels.
//
// Numeric conventions used by every kernel in csrc/:
//   * all lattice quantities are base-2 logarithms (v_exp_f32 / v_log_f32 are base-2 natively, so a
//     two-argument log-sum-exp is max + log2(1 + exp2(min - max)): 6 VALU ops, 2 of them transcendental);
//   * log(0) is the finite sentinel NEG = -1e30 instead of -inf, so (a - b) never produces NaN and the
//     recursion needs no special-casing (tools.py:57-71 handles (-inf,-inf) with an explicit branch);
//     anything below NEG_THR is reported as -inf at the boundary;
//   * every stored lattice row carries the cumulative renormalisation offset that was subtracted from
//     it, as a (hi, lo) float pair; true value = stored + hi + lo.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Product builds carry no experiment / diagnostic switch: every one of them is cleared here unless the build sets CTC_DIAG
// (scripts/build_*variant.sh do).  (ctc_fused6.hip clears its own parameter macros the same way.)
#ifndef CTC_DIAG
#undef CTC_FUSED_STAMPS
#undef CTC_F5_X
#undef CTC_F5_Y
#undef CTC_DBG_NO_E2
#undef CTC_DBG_NO_G
#undef CTC_HESS_STAMPS
#undef CTC_HESS_DBG_NOSWEEP
#undef CTC_HESS_DBG_NOFILL
#undef CTC_HESS_DBG_NOSTORE
#undef CTC_EXPERIMENT_NO_STORE
#undef CTC_F6_STAMPS
#undef CTC_F6_DEBUG
#undef CTC_F6_DEBUG2
#undef CTC_F6_SYNC
#undef CTC_F6_ONLY
#undef CTC_F6_X
#undef CTC_F6_Y
#undef CTC_F6_PFD
#undef CTC_F6_RN12
#undef CTC_F6_PRIO1
#undef CTC_F6_NH12
#undef CTC_F6_LA
#undef CTC_F6_HPRIO_B
#undef CTC_F6_NS_ONLY
#undef CTC_F6_NO_D7
#undef CTC_F6_GAP_LIVE
#undef CTC_F6_PACKED
#undef CTC_F6_P1SYNC
#endif

namespace ctc {

constexpr float NEG = -1.0e30f;
constexpr float NEG_THR = -1.0e29f;
constexpr float LOG2E = 1.44269504088896340736f;
constexpr double LN2_D = 0.69314718055994530942;
constexpr int WAVE = 64;

__device__ __forceinline__ float fexp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float flog2(float x) { return __builtin_amdgcn_logf(x); }

// v_max_f32 / v_min_f32 / v_max3_f32 exactly as written.  fmaxf() / fminf() quiet signalling NaNs first (IEEE mode), which costs
// one more VALU operation per fresh operand: 5-6 of the ~46 per frame in the softmax statistics of the fused kernels.  With a
// quiet NaN operand these return the other operand, like fmaxf().
__device__ __forceinline__ float vmax_raw(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmin_raw(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmax3_raw(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// base-2 log(2^a + 2^b); operands are finite (sentinel instead of -inf).
__device__ __forceinline__ float lse2(float a, float b) {
  return fmaxf(a, b) + flog2(1.0f + fexp2(-fabsf(a - b)));  // |.| and the negation are free source modifiers
}

// The same on a float64 state (the log-domain roles of the fused tiers since r04: the sweep's log-sum-exp chain is what carries
// their error -- tests/tools/logdomain_error_model.py, DESIGN.md section 2: float32 chain 2e-5 ... 8e-5 of posterior error at T = 1000
// with N(0, 4^2) logits, float64 chain 9e-7 with the SAME float32 emissions; the transcendental part stays float32: its argument
// is a difference, its result lies in [0, 1]).  gfx950 issues v_add_f64 / v_max_f64 at the float32 rate.
__device__ __forceinline__ double lse2(double a, double b) {
  const float d = (float)(a - b);
  return fmax(a, b) + (double)flog2(1.0f + fexp2(-fabsf(d)));
}

// lane i receives x from lane i-1; lane 0 receives `fill` (DPP wave_shr:1, one VALU op, no LDS).
__device__ __forceinline__ float from_prev_lane(float x, float fill) {
  int v = __builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(x), 0x138, 0xf, 0xf, false);
  return __int_as_float(v);
}
// lane i receives x from lane i+1; lane 63 receives `fill` (DPP wave_shl:1).
__device__ __forceinline__ float from_next_lane(float x, float fill) {
  int v = __builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(x), 0x130, 0xf, 0xf, false);
  return __int_as_float(v);
}

// (float64: the two halves travel separately)
__device__ __forceinline__ double from_prev_lane(double x, double fill) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(fill), __double2loint(x), 0x138, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(fill), __double2hiint(x), 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double from_next_lane(double x, double fill) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(fill), __double2loint(x), 0x130, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(fill), __double2hiint(x), 0x130, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// Lanes of ONE wavefront exchanging data through LDS (scatter by one lane, read by another): the hardware executes the
// LDS operations of a wavefront in program order, but the compiler reasons per thread -- without a fence it may forward a
// lane's own earlier store to its load and skip the read (it did, for lanes that sit out a conditional atomic).
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Accessors of workspace rows.  COH = false: plain loads / stores (producer and consumer are different launches).  COH = true: rows
// handed from one workgroup to another INSIDE a launch (ctc_wide.hip): the eight XCDs of the chip have one L2 each, which are not
// coherent with one another during a kernel, so such rows are written through and read past them -- relaxed agent-scope atomics
// compile to global_store / global_load with sc1 set (32 / 64 bits each; wider accesses are split).  Ordering is the caller's:
// a producer waits for its stores (s_waitcnt vmcnt(0)) before it raises a flag, a consumer issues its loads after it has seen the flag.
// (The formally obvious alternative -- plain accesses with agent-scope release / acquire fences -- writes back / invalidates a whole
// L2 per fence: measured 4x slower than the three-kernel pipeline with one fence per row.)
template <bool COH> __device__ __forceinline__ float ld1(const float *p) {
  if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return *p;
}
template <bool COH> __device__ __forceinline__ float2 ld2(const float *p) {
  if constexpr (COH) {
    const unsigned long long u = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float2(__uint_as_float((unsigned)u), __uint_as_float((unsigned)(u >> 32)));
  } else {
    return *reinterpret_cast<const float2 *>(p);
  }
}
template <bool COH> __device__ __forceinline__ float4 ld4(const float *p) {
  if constexpr (COH) {
    const float2 a = ld2<true>(p), b = ld2<true>(p + 2);
    return make_float4(a.x, a.y, b.x, b.y);
  } else {
    return *reinterpret_cast<const float4 *>(p);
  }
}
template <bool COH> __device__ __forceinline__ void st1(float *p, float v) {
  if constexpr (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}
template <bool COH> __device__ __forceinline__ void st2(float *p, float2 v) {
  if constexpr (COH) {
    const unsigned long long u = (unsigned long long)__float_as_uint(v.x) | ((unsigned long long)__float_as_uint(v.y) << 32);
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    *reinterpret_cast<float2 *>(p) = v;
  }
}
template <bool COH> __device__ __forceinline__ void st4(float *p, float4 v) {
  if constexpr (COH) { st2<true>(p, make_float2(v.x, v.y)); st2<true>(p + 2, make_float2(v.z, v.w)); }
  else *reinterpret_cast<float4 *>(p) = v;
}

// ---- wave64 reductions with DPP (result broadcast through an SGPR) ----
// Hand-written: hipcc lowers __builtin_amdgcn_update_dpp reductions to mov + mov_dpp + op per level (18 instructions
// per reduction); here every level is ONE DPP-fused VALU op.  Lanes without a valid DPP source are disabled and keep
// their value.  `s_nop 1` = the 2 wait states a DPP read of a VGPR written by the previous VALU op needs (hipcc pads
// nothing inside asm).  After the row_shr scan lane 15 of each 16-lane row holds the row result; row_bcast:15 / :31
// carry it into lane 63.
#define CTC_WAVE_REDUCE_ASM(OP)                                                      \
  "s_nop 1\n\t" OP " %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"            \
  "s_nop 1\n\t" OP " %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"            \
  "s_nop 1\n\t" OP " %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"            \
  "s_nop 1\n\t" OP " %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"            \
  "s_nop 1\n\t" OP " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"         \
  "s_nop 1\n\t" OP " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"         \
  "s_nop 0"

__device__ __forceinline__ float wave_sum_dpp(float v) {
  asm(CTC_WAVE_REDUCE_ASM("v_add_f32_dpp") : "+v"(v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max_dpp(float v) {
  asm(CTC_WAVE_REDUCE_ASM("v_max_f32_dpp") : "+v"(v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// wave-wide maximum / sum, the same value in every lane.  (Butterflies of __shfl_xor compile to six dependent ds_bpermute
// round trips through the LDS crossbar, ~1 000 cycles of latency per reduction in the one-row-per-wavefront kernels.)
__device__ __forceinline__ float wave_max(float v) { return wave_max_dpp(v); }
__device__ __forceinline__ float wave_sum(float v) { return wave_sum_dpp(v); }

// [0] += v in units of 2^-20 (finite v only), [1] += 1: integer atomics, so the total does not depend on the order.
// A finite loss beyond +-2^42 (a logit of 1e13) enters the sum clamped to that: 2^62 units, so that one such term cannot wrap
// the int64 (include/ctc_amd.h, ctc_amd_loss_grad_sum).
__device__ __forceinline__ void add_loss_fixed(long long *acc, float v, int sign = 1) {
  if ((v - v) == 0.f) {
    v = fminf(fmaxf(v, -4398046511104.0f), 4398046511104.0f);
    atomicAdd(reinterpret_cast<unsigned long long *>(acc), (unsigned long long)((long long)sign * __float2ll_rn(v * 1048576.0f)));
    atomicAdd(reinterpret_cast<unsigned long long *>(acc) + 1, (unsigned long long)(long long)sign);
  }
}

// One call's inputs (device pointers) and shapes.
struct Problem {
  const float *logits;
  const int32_t *labels;
  const int32_t *label_length;
  const int32_t *logit_length;
  int label_stride, blank, B, T, V, U, kind, wrt;
  // producer formats (ctc_amd_loss_grad_ex): element strides of the batch and time axes of logits and gradient (the
  // token axis is contiguous) and their element types (0 = float32, 1 = bfloat16).  Everything else reads `logits`
  // as contiguous float32 [B,T,V].
  long xsb, xst, gsb, gst;
  int xdtype, gdtype;  // 0 = float32, 1 = bfloat16, 2 = float16 (the last: three-kernel pipeline only)
  // packed (ragged) batches, ctc_amd_loss_grad_packed: first row of every utterance in logits and gradient (rows of xst / gst
  // elements; xsb / gsb are then unused and rows beyond logit_length do not exist).  Three-kernel pipeline only.
  const long long *row0 = nullptr;
  // ctc_amd_loss_grad_sum: [0] += sum of the finite losses in units of 2^-20 (integer adds: the same bits whatever the order),
  // [1] += their number; sum_zero (the buffer of the NEXT step, if any) is cleared by this call
  long long *sum_out = nullptr, *sum_zero = nullptr;
  // 1: second half of a two-call loss -> gradient sequence (ctc_amd_grad_resume): the workspace still holds what the
  // loss-only call left (checkpoints, softmax statistics, log P, flags); only the linear-domain fused kernel uses it
  int resume = 0;
  // low 4 bits of (logits pointer | gradient pointer): the 16-byte (float32) / 8-byte (bfloat16) row accesses of the fast
  // paths need aligned bases as well as aligned strides; otherwise the element-wise paths run
  int align_bits = 0;
};

// element offset of row (b, t) of the logits / of the gradient
__device__ __forceinline__ long logits_off(const Problem &p, int b, int t) {
  return (p.row0 ? (long)p.row0[b] * p.xst : (long)b * p.xsb) + (long)t * p.xst;
}
__device__ __forceinline__ long grad_off(const Problem &p, int b, int t) {
  return (p.row0 ? (long)p.row0[b] * p.gst : (long)b * p.gsb) + (long)t * p.gst;
}
// float16 <-> float32 (v_cvt_f32_f16 / v_cvt_f16_f32, round to nearest even)
__device__ __forceinline__ float f16_to_f32(unsigned short h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ unsigned short f32_to_f16(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }
// 16-bit element of type dt (1 = bfloat16, 2 = float16) <-> float32
__device__ __forceinline__ float h16_to_f32(unsigned short h, int dt) { return dt == 2 ? f16_to_f32(h) : __uint_as_float((unsigned)h << 16); }

// bfloat16 <-> float32 (round to nearest even on the way back)
__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
// two at once: gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32, round to nearest even) -- lo in bits 0..15
__device__ __forceinline__ unsigned f32x2_to_bf16x2(float lo, float hi) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef __bf16 b2 __attribute__((ext_vector_type(2)));
  f2 v = {lo, hi};
  b2 w = __builtin_convertvector(v, b2);
  return __builtin_bit_cast(unsigned, w);
}
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
  const unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);  // NaN stays NaN
  return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

// Device workspace layout.  UP = 64*NL lattice slots (label positions) per utterance, NL per lane.
//   emis  [B][T][ERS]      : E[0..UP) = log2 p(label[i] at frame t) (NEG beyond label_length), [UP] = log2 p(blank),
//                            [UP+1] = row max mx, [UP+2] = log2 sum_k exp(x_k - mx)  (both 0 for WRT_LOGPROBS)
//   alpha [B][T+1][SRS]    : classic: pairs (closed, open) of label position l = i+1 at [2i, 2i+1], the l = 0 pair at
//   beta                     [2UP, 2UP+1], offset (hi, lo) at [2UP+2, 2UP+3];  simplified: state l = i+1 at [i], l = 0 at
//                            [UP], offset at [UP+2, UP+3].  A second 16-byte tail [.. +4, +8) carries, in the fused kernel,
//                            the softmax statistics (row max, log2 sum exp) of the frame the reader will process
//   logp  [B] double       : log2 P(label | logits), -inf when infeasible
struct Layout {
  int NL, UP, ERS, SRS;
  // rows per utterance and direction in the alpha / beta regions, checkpoint slots per utterance and direction in the
  // exponent region, and the block length of the compact layout (0: full layout, lattice row t lives at row index t;
  // B > 0: only checkpoint rows exist, lattice time t lives at row index ceil(t / B) -- the fused tiers, which keep one
  // row per block and 8 bytes of softmax statistics per frame in place of the emission rows)
  int rows_b, nslot, ck_blk;
  size_t off_emis, off_alpha, off_beta, off_logp, off_dummy, off_perm, off_kexp, off_flags, off_meet, off_extra, total;
};

inline int nl_for(int U) {
  int nl = 1;
  while (nl * WAVE < U) nl *= 2;
  return nl;
}

// frames per block of the checkpoint + recompute kernels (ctc_fused5.hip / ctc_fused6.hip launch tables)
inline int fused_blk(int NL, int V) { return NL >= 8 ? 3 : (NL == 4 || V > 256) ? 6 : 12; }

inline Layout make_layout(int kind, int B, int T, int U, size_t extra_bytes, int ck_blk = 0) {
  Layout L;
  L.NL = nl_for(U);
  L.UP = L.NL * WAVE;
  L.ERS = L.UP + 4;
  L.SRS = (kind == 0 ? 2 * L.UP : L.UP) + 8;
  L.ck_blk = ck_blk;
  L.rows_b = ck_blk > 0 ? (T + ck_blk - 1) / ck_blk + 3 : T + 1;
  L.nslot = ck_blk > 0 ? L.rows_b : (T + 2) / 3 + 3;
  auto al = [](size_t x) { return (x + 255) & ~size_t(255); };
  size_t o = 0;
  L.off_emis = o;  o = al(o + (ck_blk > 0 ? (size_t)B * T * 8 : (size_t)B * T * L.ERS * 4));
  L.off_alpha = o; o = al(o + (size_t)B * L.rows_b * L.SRS * 4);
  L.off_beta = o;  o = al(o + (size_t)B * L.rows_b * L.SRS * 4);
  L.off_logp = o;  o = al(o + (size_t)B * 8);
  L.off_dummy = o; o = al(o + (size_t)B * 2 * 1024);  // per-wavefront sink for the pacing stores of the fused kernels (diagnostic builds: cycle stamps)
  L.off_perm = o;  o = al(o + (size_t)B * 4);  // longest-first order of the utterances (fused kernel, B > number of CUs)
  // linear-domain fused kernel (ctc_fused6.hip): per-lane exponents of its checkpoint rows ([B][2][nslot][64] int32) and the
  // per-utterance flags that send an utterance to the log-domain roles
  L.off_kexp = o;  o = al(o + (size_t)B * 2 * L.nslot * 64 * 4);
#ifdef CTC_DIAG
  L.off_flags = o; o = al(o + (size_t)B * 4 + (size_t)B * 2048 * 4);  // (+ 8 KB per utterance for diagnostic builds)
#else
  L.off_flags = o; o = al(o + (size_t)B * 4);
#endif
  L.off_meet = o;  o = al(o + (size_t)B * 8);  // per utterance: posterior scale (integer exponent, mantissa factor) from the meeting point
  L.off_extra = o; o = al(o + extra_bytes);
  L.total = o;
  return L;
}

}  // namespace ctc

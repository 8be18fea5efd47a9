// Fused loss + gradient kernel for gfx950 (pipeline v2): ONE launch, one workgroup of two wavefronts per utterance.
//
//   wave A runs alpha forward from frame 0, wave B runs beta backward from frame len-1 (classic_ctc_loss.py:310-462,
//   simplified_ctc_loss.py:291-438).  They meet at tm = len/2:
//     phase 1  A: frames [0, tm)      B: frames [tm, len)    each spills its lattice rows (half of what v1 spills)
//     phase 2  A: frames [tm, len)    B: frames [0, tm)      each reads the OTHER side's rows, forms the posteriors
//                                                           and writes the finished gradient rows
//   log P comes from the meeting point: sum_s alpha[tm,s] beta[tm,s] = P (the invariant the reference tests,
//   tests/test_classic_ctc_loss.py:146-167).
//   Everything per frame is done in-wave, with no inter-wave synchronisation in the steady state:
//     logits row (float4/lane, prefetched 16 frames ahead in registers) -> DPP max / sum reductions (log-softmax,
//     tools.py:27-40) -> gather of the label emissions through an LDS copy of the row (base_loss.py:328-344) ->
//     lattice step in registers -> posterior scatter with ds_add_f32 into an LDS token row (base_loss.py:420-468) ->
//     softmax - posterior written as one 16-byte store per lane (base_loss.py:262-298 + autodiff of tools.py:37-39).
//   HBM traffic per utterance: logits read twice, gradient written once, half of the alpha/beta rows written and read
//   once (vs. v1: 3 passes over [T,V] data plus all alpha/beta rows twice).
//
// Eligibility (else the v1 pipeline of ctc_kernels.hip runs): V a multiple of 256 and <= 1024, U <= 256.
#include "ctc_common.h"

namespace ctc {

namespace fused {

constexpr int PF = 16;   // logits rows of look-ahead (= unrolled block length = renormalisation period)
constexpr int PFS = 8;   // spilled lattice rows of look-ahead in phase 2 (keeps the kernel inside 256 VGPRs)
constexpr int NPACE = 48; // pacing stores after a ring prologue (see Side::pace)

__device__ __forceinline__ int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

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
__device__ __forceinline__ float readlane_f(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// One spilled lattice row as the other side reads it: per lane NL slots of (first[, second]) plus the 16-byte tail.
template <int KIND, int NL>
struct SRow {
  float a[NL];   // classic: closed part / simplified: the state
  float b[NL];   // classic: open part (unused for simplified)
  float4 tail;   // (state outside the slot range, -, off_hi, off_lo)
  float2 stat;   // (row max, log2 sum exp) of the frame the reader processes with this row
};

template <int KIND, int NL>
__device__ __forceinline__ void load_srow(SRow<KIND, NL> &r, const float *__restrict__ row, int lane, int UP) {
  if constexpr (KIND == 0) {
    const float *p = row + 2 * lane * NL;
    if constexpr (NL == 1) {
      float2 v = *reinterpret_cast<const float2 *>(p);
      r.a[0] = v.x; r.b[0] = v.y;
    } else {
#pragma unroll
      for (int q = 0; q < NL / 2; ++q) {
        float4 v = *reinterpret_cast<const float4 *>(p + 4 * q);
        r.a[2 * q] = v.x; r.b[2 * q] = v.y; r.a[2 * q + 1] = v.z; r.b[2 * q + 1] = v.w;
      }
    }
    r.tail = *reinterpret_cast<const float4 *>(row + 2 * UP);
    r.stat = *reinterpret_cast<const float2 *>(row + 2 * UP + 4);
  } else {
    const float *p = row + lane * NL;
    if constexpr (NL == 1) {
      r.a[0] = p[0];
    } else if constexpr (NL == 2) {
      float2 v = *reinterpret_cast<const float2 *>(p);
      r.a[0] = v.x; r.a[1] = v.y;
    } else {
#pragma unroll
      for (int q = 0; q < NL / 4; ++q) {
        float4 v = *reinterpret_cast<const float4 *>(p + 4 * q);
        r.a[4 * q] = v.x; r.a[4 * q + 1] = v.y; r.a[4 * q + 2] = v.z; r.a[4 * q + 3] = v.w;
      }
    }
    r.tail = *reinterpret_cast<const float4 *>(row + UP);
    r.stat = *reinterpret_cast<const float2 *>(row + UP + 4);
  }
}

template <int KIND, int NL>
__device__ __forceinline__ void store_srow(float *__restrict__ row, int lane, int UP, const float (&a)[NL],
                                           const float (&b)[NL], float4 tail, float2 stat) {
  if constexpr (KIND == 0) {
    float *p = row + 2 * lane * NL;
    if constexpr (NL == 1) {
      *reinterpret_cast<float2 *>(p) = make_float2(a[0], b[0]);
    } else {
#pragma unroll
      for (int q = 0; q < NL / 2; ++q)
        *reinterpret_cast<float4 *>(p + 4 * q) = make_float4(a[2 * q], b[2 * q], a[2 * q + 1], b[2 * q + 1]);
    }
    // 32-byte wave-uniform tail in ONE store instruction: even lanes write the first half, odd lanes the second
    const bool odd = lane & 1;
    *reinterpret_cast<float4 *>(row + 2 * UP + (odd ? 4 : 0)) =
        make_float4(odd ? stat.x : tail.x, odd ? stat.y : tail.y, odd ? 0.f : tail.z, odd ? 0.f : tail.w);
  } else {
    float *p = row + lane * NL;
    if constexpr (NL == 1) {
      p[0] = a[0];
    } else if constexpr (NL == 2) {
      *reinterpret_cast<float2 *>(p) = make_float2(a[0], a[1]);
    } else {
#pragma unroll
      for (int q = 0; q < NL / 4; ++q)
        *reinterpret_cast<float4 *>(p + 4 * q) = make_float4(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);
    }
    const bool odd = lane & 1;
    *reinterpret_cast<float4 *>(row + UP + (odd ? 4 : 0)) =
        make_float4(odd ? stat.x : tail.x, odd ? stat.y : tail.y, odd ? 0.f : tail.z, odd ? 0.f : tail.w);
  }
}

// Per-frame emissions in base-2 logs.
template <int NL>
struct Emis {
  float y[NL];
  float bl, mx, l2s;
};

// The whole per-side program.  DIR 0 = side A (alpha, forward), DIR 1 = side B (beta, backward).
// Slot i = lane*NL + j is label position i.  State convention (same as Scan in ctc_kernels.hip):
//   classic    A: c[j] = closed(l=i+1), o[j] = open(l=i+1), cx = closed(l=0)
//              B: c[j] = closed(l=i),   o[j] = open(l=i+1), cx = closed(l=UP)
//   simplified A: c[j] = a(l=i+1), cx = a(l=0);   B: c[j] = b(l=i), cx = b(l=UP)
// Rows are spilled in the layout the OTHER side's slots are aligned with:
//   A -> rows[t] : slot i = (state_c(l=i) [, open(l=i+1)]), tail.x = state_c(l=UP)       (read by B)
//   B -> rows[t] : slot i = (state_c(l=i+1) [, open(l=i+1)]), tail.x = state_c(l=0)      (read by A)
template <int KIND, int NL, int VPL, int DIR, bool LOGITS>
struct Side {
  static constexpr int V = 256 * VPL;
  // lattice state
  float c[NL], o[NL], cx;
  double off;
  bool norep[NL], norep_next[NL];
  int tokoff[NL];  // byte offset of label[i] inside the LDS copy of the logits row (pad slot for i >= label_length)
  float mb[4 * VPL];  // 1.0 at this lane's element that is the blank column, else 0
  // geometry
  int lane, UP, len, ll, blank;
  const float *xbase;   // logits of this utterance
  float *gbase;         // gradient of this utterance
  float *own_rows;      // spill rows this side writes
  const float *oth_rows;  // spill rows the other side writes
  int SRS;
  float *sink;  // global: 1 KB per wavefront, target of the pacing stores of the ring prologues
  float *xs;    // LDS: 2 x (V + 4) floats, gather copies of the logits row
  float *bins;  // LDS: V floats, posterior per token
  float dl;

  __device__ __forceinline__ int frame(int t0, int k) const { return DIR == 0 ? t0 + k : t0 - k; }

  // Pacing store.  hipcc derives the s_waitcnt vmcnt(N) of a software-pipelined loop from the LEAST number of memory
  // operations it can prove to lie between a prefetch and its use, and that minimum comes from the ring prologue where
  // the prefetches would be back to back.  Giving every prologue slot as many memory operations as a steady-state step
  // has makes the derived N as large as in the steady state, so a wait never reaches stores/loads of the last few steps.
  __device__ __forceinline__ void pace(int slot) const {
    volatile float *q = sink + lane * 4;  // volatile: identical stores to one address must not be merged away
    q[0] = (float)slot;
  }

  __device__ __forceinline__ void load_x(float4 (&xr)[VPL], int t) const {
    const float *row = xbase + (long)t * V + lane * 4;
#pragma unroll
    for (int q = 0; q < VPL; ++q) xr[q] = *reinterpret_cast<const float4 *>(row + 256 * q);
  }

  // log-softmax statistics (tools.py:27-40): row max and log2 sum exp by DPP reductions
  __device__ __forceinline__ void stats(const float4 (&xr)[VPL], float &mx, float &l2s) const {
    mx = 0.f; l2s = 0.f;
    if constexpr (LOGITS) {
      float m = fmaxf(fmaxf(xr[0].x, xr[0].y), fmaxf(xr[0].z, xr[0].w));
#pragma unroll
      for (int q = 1; q < VPL; ++q) m = fmaxf(m, fmaxf(fmaxf(xr[q].x, xr[q].y), fmaxf(xr[q].z, xr[q].w)));
      mx = wave_max_dpp(m);
      mx = (mx == -INFINITY) ? 0.f : mx;
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < VPL; ++q)
        s += fexp2((xr[q].x - mx) * LOG2E) + fexp2((xr[q].y - mx) * LOG2E) + fexp2((xr[q].z - mx) * LOG2E) +
             fexp2((xr[q].w - mx) * LOG2E);
      l2s = flog2(wave_sum_dpp(s));
    }
  }

  // emission gather through an LDS copy of the row (base_loss.py:328-344, 365-371), split in two so that the LDS
  // round trip of frame t+1 is in flight while frame t is processed: gather_issue early, gather_finish late.
  struct Raw { float xg[NL]; float xb; };
  __device__ __forceinline__ void gather_issue(const float4 (&xr)[VPL], int parity, Raw &w) const {
    float *buf = xs + parity * (V + 4);
#pragma unroll
    for (int q = 0; q < VPL; ++q) *reinterpret_cast<float4 *>(buf + 256 * q + lane * 4) = xr[q];
    const char *bb = reinterpret_cast<const char *>(buf);
#pragma unroll
    for (int j = 0; j < NL; ++j) w.xg[j] = *reinterpret_cast<const float *>(bb + tokoff[j]);
    w.xb = buf[blank];
  }
  __device__ __forceinline__ void gather_finish(const Raw &w, float mx, float l2s, Emis<NL> &e) const {
#pragma unroll
    for (int j = 0; j < NL; ++j)
      e.y[j] = fmaxf((w.xg[j] - mx) * LOG2E - l2s, NEG);  // v_max returns the non-NaN operand: (-inf) - (-inf) -> sentinel
    e.bl = fmaxf((w.xb - mx) * LOG2E - l2s, NEG);
    e.mx = mx;
    e.l2s = l2s;
  }
  __device__ __forceinline__ void gather(const float4 (&xr)[VPL], int parity, float mx, float l2s, Emis<NL> &e) const {
    Raw w;
    gather_issue(xr, parity, w);
    gather_finish(w, mx, l2s, e);
  }

  __device__ __forceinline__ void emit(const float4 (&xr)[VPL], int parity, Emis<NL> &e) const {
    float mx, l2s;
    stats(xr, mx, l2s);
    gather(xr, parity, mx, l2s, e);
  }

  // one lattice step (identical recursions to Scan::step in ctc_kernels.hip)
  __device__ __forceinline__ void step(const Emis<NL> &e) {
    const float bl = e.bl;
    if constexpr (KIND == 0 && DIR == 0) {
      float m[NL], x[NL];
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        m[j] = lse2(c[j], o[j]);
        x[j] = norep_next[j] ? m[j] : c[j];
      }
      float xin0 = from_prev_lane(x[NL - 1], cx);
#pragma unroll
      for (int j = NL - 1; j >= 0; --j) {
        float xin = (j == 0) ? xin0 : x[j - 1];
        o[j] = e.y[j] + lse2(o[j], xin);
        c[j] = bl + m[j];
      }
      cx += bl;
    } else if constexpr (KIND == 0 && DIR == 1) {
      float h[NL], ee[NL], pn[NL], x[NL];
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        h[j] = bl + c[j];
        ee[j] = e.y[j] + o[j];
        pn[j] = lse2(h[j], ee[j]);
        x[j] = norep[j] ? pn[j] : h[j];
      }
      cx += bl;
      float xinl = from_next_lane(x[0], cx);
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        float xin = (j == NL - 1) ? xinl : x[j + 1];
        o[j] = lse2(xin, ee[j]);
        c[j] = pn[j];
      }
    } else if constexpr (KIND == 1 && DIR == 0) {
      float pin0 = from_prev_lane(c[NL - 1], cx);
#pragma unroll
      for (int j = NL - 1; j >= 0; --j) {
        float pin = (j == 0) ? pin0 : c[j - 1];
        c[j] = lse2(bl + c[j], e.y[j] + pin);
      }
      cx += bl;
    } else {
      float nin = from_next_lane(c[0], cx);
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        float nx = (j == NL - 1) ? nin : c[j + 1];
        c[j] = lse2(bl + c[j], e.y[j] + nx);
      }
      cx += bl;
    }
  }

  __device__ __forceinline__ void renorm() {
    float mx = cx;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      mx = fmaxf(mx, c[j]);
      if constexpr (KIND == 0) mx = fmaxf(mx, o[j]);
    }
    mx = wave_max_dpp(mx);
    mx = (mx > NEG_THR) ? mx : 0.f;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      c[j] -= mx;
      if constexpr (KIND == 0) o[j] -= mx;
    }
    cx -= mx;
    off += (double)mx;
  }

  // spill the current state as lattice row `t` in the layout the other side is aligned with
  __device__ __forceinline__ void spill(int t, float smx, float sl2s) const {
    float cs[NL];
    float tx;
    if constexpr (DIR == 0) {  // slot i <- state_c(l=i): previous slot's c; tail <- state_c(l=UP): last slot's c
#pragma unroll
      for (int j = NL - 1; j > 0; --j) cs[j] = c[j - 1];
      cs[0] = from_prev_lane(c[NL - 1], cx);
      tx = readlane_f(c[NL - 1], 63);
    } else {  // slot i <- state_c(l=i+1): next slot's c; tail <- state_c(l=0): first slot's c
#pragma unroll
      for (int j = 0; j < NL - 1; ++j) cs[j] = c[j + 1];
      cs[NL - 1] = from_next_lane(c[0], cx);
      tx = readlane_f(c[0], 0);
    }
    const float oh = (float)off;
    store_srow<KIND, NL>(own_rows + (long)t * SRS, lane, UP, cs, o, make_float4(tx, 0.f, oh, (float)(off - (double)oh)),
                         make_float2(smx, sl2s));
  }

  // log2 P at the meeting point from this side's state and the other side's row of the same time index
  __device__ __forceinline__ double meet(const SRow<KIND, NL> &r) const {
    float v[2 * NL + 1];
    float m = cx + r.tail.x;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      v[2 * j] = c[j] + r.a[j];
      v[2 * j + 1] = (KIND == 0) ? o[j] + r.b[j] : NEG;
      m = fmaxf(m, fmaxf(v[2 * j], v[2 * j + 1]));
    }
    m = wave_max_dpp(m);
    if (!(m > NEG_THR)) return -INFINITY;
    float s = (lane == 0) ? fexp2(cx + r.tail.x - m) : 0.f;
#pragma unroll
    for (int j = 0; j < 2 * NL; ++j) s += fexp2(v[j] - m);
    s = wave_sum_dpp(s);
    return (double)m + (double)flog2(s) + off + (double)r.tail.z + (double)r.tail.w;
  }

  // posterior scatter + gradient row of frame t.  s1/s2/s0 are base-2 log posteriors of the blank parts, the token parts
  // and the out-of-range blank part (see the table in the kernel body); xr is the logits row, e its statistics.
  __device__ __forceinline__ void grad_row(int t, const float (&s1)[NL], const float (&s2)[NL], float s0,
                                           const float4 (&xr)[VPL], const Emis<NL> &e) const {
#pragma unroll
    for (int q = 0; q < VPL; ++q) *reinterpret_cast<float4 *>(bins + 256 * q + lane * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    float qb = (lane == 0) ? fexp2(s0) : 0.f;
    float qt[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      qb += fexp2(s1[j]);
      qt[j] = fexp2(s2[j]);
    }
    // (no wave barrier needed: LDS ops of one wave execute in order and may-alias accesses keep program order)
    char *bb = reinterpret_cast<char *>(bins);
#pragma unroll
    for (int j = 0; j < NL; ++j) atomicAdd(reinterpret_cast<float *>(bb + tokoff[j]), qt[j]);  // pad slot absorbs i >= ll
    qb = wave_sum_dpp(qb);
    // (no wave barrier needed: LDS ops of one wave execute in order and may-alias accesses keep program order)
    float *g = gbase + (long)t * V + lane * 4;
#pragma unroll
    for (int q = 0; q < VPL; ++q) {
      float4 pq = *reinterpret_cast<const float4 *>(bins + 256 * q + lane * 4);
      pq.x += mb[4 * q] * qb; pq.y += mb[4 * q + 1] * qb; pq.z += mb[4 * q + 2] * qb; pq.w += mb[4 * q + 3] * qb;
      float4 r;
      if constexpr (LOGITS) {
        r.x = dl * (fexp2((xr[q].x - e.mx) * LOG2E - e.l2s) - pq.x);
        r.y = dl * (fexp2((xr[q].y - e.mx) * LOG2E - e.l2s) - pq.y);
        r.z = dl * (fexp2((xr[q].z - e.mx) * LOG2E - e.l2s) - pq.z);
        r.w = dl * (fexp2((xr[q].w - e.mx) * LOG2E - e.l2s) - pq.w);
      } else {
        r.x = -dl * pq.x; r.y = -dl * pq.y; r.z = -dl * pq.z; r.w = -dl * pq.w;
      }
      *reinterpret_cast<float4 *>(g + 256 * q) = r;
    }
    // (no wave barrier needed: LDS ops of one wave execute in order and may-alias accesses keep program order)
  }

  // phase-2 frame: posterior of frame t from this side's state and the other side's row, then the gradient row.
  //   classic    A: after the step, state = alpha[t+1], r = beta[t+1]      B: before the step, state = beta[t+1], r = alpha[t+1]
  //   simplified A: before the step, state = a[t], r = b[t+1]              B: before the step, state = b[t+1], r = a[t]
  // (blank part, token part) per slot:  classic (c + r.a, o + r.b);  simplified A (c + bl + r.a, pin + y + r.a);
  // simplified B (c + bl + r.a, r.a + y + next)   -- regroupings of classic_ctc_loss.py:565-669 / simplified_ctc_loss.py:456-534
  __device__ __forceinline__ void frame2(int t, const float4 (&xr)[VPL], const Emis<NL> &e, const SRow<KIND, NL> &r, double dlogp) {
    float s1[NL], s2[NL], s0;
    if constexpr (KIND == 0 && DIR == 0) step(e);
    const float sc = (float)((double)r.tail.z + (off - dlogp)) + r.tail.w;
    if constexpr (KIND == 0) {
#pragma unroll
      for (int j = 0; j < NL; ++j) { s1[j] = c[j] + r.a[j] + sc; s2[j] = o[j] + r.b[j] + sc; }
      s0 = cx + r.tail.x + sc;
    } else if constexpr (DIR == 0) {
      float pin0 = from_prev_lane(c[NL - 1], cx);
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        float pin = (j == 0) ? pin0 : c[j - 1];
        s1[j] = c[j] + e.bl + r.a[j] + sc;
        s2[j] = pin + e.y[j] + r.a[j] + sc;
      }
      s0 = cx + e.bl + r.tail.x + sc;
    } else {
      float nin = from_next_lane(c[0], cx);
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        float nx = (j == NL - 1) ? nin : c[j + 1];
        s1[j] = c[j] + e.bl + r.a[j] + sc;
        s2[j] = r.a[j] + e.y[j] + nx + sc;
      }
      s0 = cx + e.bl + r.tail.x + sc;
    }
    grad_row(t, s1, s2, s0, xr, e);
    if constexpr (!(KIND == 0 && DIR == 0)) step(e);
  }

  __device__ __forceinline__ void zero_rows(int t_from, int t_to) const {
    for (int t = t_from; t < t_to; ++t) {
      float *g = gbase + (long)t * V + lane * 4;
#pragma unroll
      for (int q = 0; q < VPL; ++q) *reinterpret_cast<float4 *>(g + 256 * q) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
};

template <int KIND, int NL, int VPL, int DIR, bool LOGITS>
__device__ __forceinline__ void run_side(const Problem &p, const Layout &L, float *__restrict__ alpha_ws,
                                         float *__restrict__ beta_ws, double *__restrict__ logp_ws,
                                         float *__restrict__ loss, const float *__restrict__ d_loss,
                                         float *__restrict__ grad, float *__restrict__ sink_ws, float *lds_x,
                                         float *lds_bins) {
  constexpr int V = 256 * VPL;
  using S_t = Side<KIND, NL, VPL, DIR, LOGITS>;
  S_t S;
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x;
  const int T = p.T, UP = L.UP;
  S.lane = lane; S.UP = UP; S.blank = p.blank; S.SRS = L.SRS;
  S.len = clampi(p.logit_length[b], 0, T);
  int ll = p.label_length[b] < 0 ? 0 : p.label_length[b];
  const bool shape_ok = (ll <= p.U);
  if (!shape_ok) ll = 0;
  S.ll = ll;
  S.xbase = p.logits + (long)b * T * V;
  S.gbase = grad + (long)b * T * V;
  S.own_rows = (DIR == 0 ? alpha_ws : beta_ws) + (long)b * (T + 1) * L.SRS;
  S.oth_rows = (DIR == 0 ? beta_ws : alpha_ws) + (long)b * (T + 1) * L.SRS;
  S.sink = sink_ws + ((long)b * 2 + DIR) * 256;
  S.xs = lds_x;
  S.bins = lds_bins;
  S.dl = d_loss ? d_loss[b] : 1.0f;
  S.off = 0.0;
  {
    const int32_t *lab = p.labels + (long)b * p.label_stride;
    auto tok = [&](int i) -> int { return (i >= 0 && i < ll) ? ((i < p.label_stride) ? lab[i] : p.blank) : -1 - (i < 0); };
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      int i = lane * NL + j;
      int tk = tok(i);
      S.norep[j] = (i == 0) || tk != tok(i - 1);
      S.norep_next[j] = tok(i + 1) != tk;
      S.tokoff[j] = 4 * ((tk >= 0 && tk < V && tk != p.blank) ? tk : V);  // V = pad slot
      S.c[j] = NEG;
      S.o[j] = NEG;
    }
#pragma unroll
    for (int q = 0; q < VPL; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) S.mb[4 * q + e] = (256 * q + lane * 4 + e == p.blank) ? 1.f : 0.f;
  }
  // pad slots of the two LDS row copies: "log 0" for label positions beyond label_length
  if (lane == 0) {
    S.xs[V] = -6.0e29f;
    S.xs[V + 4 + V] = -6.0e29f;
  }
  const int len = S.len;
  const int tm = len / 2;

  // ---- initial state ----
  if constexpr (DIR == 0) {
    S.cx = 0.f;
  } else {
    S.cx = (ll == UP) ? 0.f : NEG;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      int i = lane * NL + j;
      if (i == ll) S.c[j] = 0.f;
      if (KIND == 0 && i == ll - 1) S.o[j] = 0.f;
    }
  }
  // Which frame's softmax statistics travel with a spilled row: the frame at which the OTHER side reads that row.
  // Classic A: row t+1 (written after frame t) is read by B at frame t = the current frame; in the three other cases
  // it is the next frame in this side's own order, whose emissions are computed one step ahead (skewed pipeline).
  constexpr bool STAT_CUR = (KIND == 0 && DIR == 0);

  // ================= phase 1 =================
  // A: frames 0 .. tm-1 (row t+1 after frame t);  B: frames len-1 .. tm (row t after frame t)
  {
    const int n1 = (DIR == 0) ? tm : len - tm;
    const int t0 = (DIR == 0) ? 0 : len - 1;
    auto fr = [&](int k) -> int {
      int kk = k < n1 ? k : n1 - 1;
      kk = kk < 0 ? 0 : kk;
      int t = DIR == 0 ? t0 + kk : t0 - kk;
      return t < 0 ? 0 : t;
    };
    Emis<NL> ecur;
    ecur.mx = 0.f; ecur.l2s = 0.f; ecur.bl = NEG;
#pragma unroll
    for (int j = 0; j < NL; ++j) ecur.y[j] = NEG;
    if (n1 > 0) {
      float4 xb[PF][VPL];
#pragma unroll
      for (int d = 0; d < PF / 2; ++d) S.load_x(xb[d], fr(d));  // first half; the second half is loaded at step 0
#pragma unroll
      for (int d = PF / 2; d < PF; ++d) xb[d][0] = xb[0][0];     // (defined values for the compiler; overwritten at step 0)
      S.emit(xb[0], 0, ecur);
      S.spill(DIR == 0 ? 0 : len, ecur.mx, ecur.l2s);  // initial row; read by the other side at this side's first frame
      int k0 = 0;
      for (; k0 + PF <= n1; k0 += PF) {
#pragma unroll
        for (int d = 0; d < PF; ++d) {
          typename S_t::Raw w;
          S.gather_issue(xb[(d + 1) % PF], (d + 1) & 1, w);  // LDS round trip of the next frame starts now
          float nmx, nl2s;
          S.stats(xb[(d + 1) % PF], nmx, nl2s);                // ... and its softmax statistics: independent of the lattice
          S.step(ecur);
          // Ring refill in two batches per block (hipcc sizes a vmcnt wait by the memory operations between a load and
          // the loop's back edge: loads issued early in the body get large counts).  Slots 8..15 (this block's second
          // half) are refilled at step 0... no: see the prologue -- at step 0 the NEXT block's slots are not free yet.
          if (d == PF / 2) {
#pragma unroll
            for (int q = 0; q < PF / 2; ++q) S.load_x(xb[q], fr(k0 + PF + q));            // frames of the next block, first half
          }
          if (d == 0) {
#pragma unroll
            for (int q = PF / 2; q < PF; ++q) S.load_x(xb[q], fr(k0 + q));                 // this block, second half
          }
          if (d == PF - 1) S.renorm();
          const int t = fr(k0 + d);
          S.spill(DIR == 0 ? t + 1 : t, STAT_CUR ? ecur.mx : nmx, STAT_CUR ? ecur.l2s : nl2s);
          S.gather_finish(w, nmx, nl2s, ecur);
          __builtin_amdgcn_sched_barrier(0);  // keep every step's loads/stores in program order (vmcnt counts stay large)
        }
      }
      // tail (< PF frames): rolled loop with direct loads -- one copy of the step body, runs at most once per phase
      for (int k = k0; k < n1; ++k) {
        const int t = fr(k);
        float4 xr[VPL];
        S.load_x(xr, fr(k + 1));
        Emis<NL> enext;
        S.emit(xr, (k + 1) & 1, enext);
        S.step(ecur);
        S.spill(DIR == 0 ? t + 1 : t, STAT_CUR ? ecur.mx : enext.mx, STAT_CUR ? ecur.l2s : enext.l2s);
        ecur = enext;
      }
    } else {
      S.spill(DIR == 0 ? 0 : len, 0.f, 0.f);
    }
  }

  // ================= meeting point =================
  __syncthreads();  // drains the spill stores of both wavefronts (vmcnt(0)) before either reads the other's rows
  double dlogp;
  {
    SRow<KIND, NL> r;
    load_srow<KIND, NL>(r, S.oth_rows + (long)tm * L.SRS, lane, UP);
    dlogp = S.meet(r);
    if (!shape_ok) dlogp = -INFINITY;
  }
  if (DIR == 0 && lane == 0) {
    logp_ws[b] = dlogp;
    loss[b] = (dlogp == -INFINITY) ? INFINITY : (float)(-dlogp * LN2_D);
  }

  // ================= phase 2 =================
  // A: frames tm .. len-1, needs beta[t+1];  B: frames tm-1 .. 0, needs alpha[t+1] (classic) / a[t] (simplified).
  // The softmax statistics of every frame come with the other side's row: no reductions on this pass.
  const int n2 = (DIR == 0) ? len - tm : tm;
  const int t0 = (DIR == 0) ? tm : tm - 1;
  if (dlogp == -INFINITY) {
    // infeasible sample: zero gradient (base_loss.py:283-288)
    if constexpr (DIR == 0) S.zero_rows(tm, T); else S.zero_rows(0, tm);
    return;
  }
  if constexpr (DIR == 0) S.zero_rows(len, T);  // padded frames (base_loss.py:291-296)
  if (n2 > 0) {
    auto fr = [&](int k) -> int { int kk = k < n2 ? k : n2 - 1; return DIR == 0 ? t0 + kk : t0 - kk; };
    auto orow = [&](int t) -> const float * {
      const int idx = (DIR == 0) ? t + 1 : (KIND == 0 ? t + 1 : t);
      return S.oth_rows + (long)idx * L.SRS;
    };
    float4 xb[PF][VPL];
    SRow<KIND, NL> rb[PFS];
#pragma unroll
    for (int d = 0; d < PF / 2; ++d) S.load_x(xb[d], fr(d));
#pragma unroll
    for (int d = PF / 2; d < PF; ++d) xb[d][0] = xb[0][0];
#pragma unroll
    for (int d = 0; d < PFS / 2; ++d) load_srow<KIND, NL>(rb[d], orow(fr(d)), lane, UP);
#pragma unroll
    for (int d = PFS / 2; d < PFS; ++d) rb[d] = rb[0];
    Emis<NL> ecur;
    S.gather(xb[0], 0, rb[0].stat.x, rb[0].stat.y, ecur);
    int k0 = 0;
    for (; k0 + PF <= n2; k0 += PF) {
#pragma unroll
      for (int d = 0; d < PF; ++d) {
        typename S_t::Raw w;
        S.gather_issue(xb[(d + 1) % PF], (d + 1) & 1, w);
        const float nmx = rb[(d + 1) % PFS].stat.x, nl2s = rb[(d + 1) % PFS].stat.y;
        S.frame2(fr(k0 + d), xb[d], ecur, rb[d % PFS], dlogp);
        if (d == PF / 2) {
#pragma unroll
          for (int q = 0; q < PF / 2; ++q) S.load_x(xb[q], fr(k0 + PF + q));
        }
        if (d == 0) {
#pragma unroll
          for (int q = PF / 2; q < PF; ++q) S.load_x(xb[q], fr(k0 + q));
        }
        if (d % (PFS / 2) == 0) {  // lattice rows: ring of PFS, refilled in batches of PFS/2, PFS/2 .. PFS frames ahead
#pragma unroll
          for (int q = 0; q < PFS / 2; ++q) {
            const int slot = (d + PFS / 2 + q) % PFS;
            load_srow<KIND, NL>(rb[slot], orow(fr(k0 + d + PFS / 2 + q)), lane, UP);
          }
        }
        if (d == PF - 1) S.renorm();
        S.gather_finish(w, nmx, nl2s, ecur);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    for (int k = k0; k < n2; ++k) {
      const int t = fr(k);
      float4 xr[VPL];
      SRow<KIND, NL> r;
      S.load_x(xr, t);
      load_srow<KIND, NL>(r, orow(t), lane, UP);
      S.gather(xr, k & 1, r.stat.x, r.stat.y, ecur);
      S.frame2(t, xr, ecur, r, dlogp);
    }
  }
}

template <int KIND, int NL, int VPL, bool LOGITS>
__global__ __launch_bounds__(128) void fused_kernel(Problem p, Layout L, float *__restrict__ alpha_ws,
                                                     float *__restrict__ beta_ws, double *__restrict__ logp_ws,
                                                     float *__restrict__ loss, const float *__restrict__ d_loss,
                                                     float *__restrict__ grad, float *__restrict__ sink_ws) {
  constexpr int V = 256 * VPL;
  __shared__ __attribute__((aligned(16))) float lds_x[2][2 * (V + 4)];
  __shared__ __attribute__((aligned(16))) float lds_bins[2][V + 4];
  const int w = threadIdx.x >> 6;
  if (w == 0)
    run_side<KIND, NL, VPL, 0, LOGITS>(p, L, alpha_ws, beta_ws, logp_ws, loss, d_loss, grad, sink_ws, lds_x[0], lds_bins[0]);
  else
    run_side<KIND, NL, VPL, 1, LOGITS>(p, L, alpha_ws, beta_ws, logp_ws, loss, d_loss, grad, sink_ws, lds_x[1], lds_bins[1]);
}

}  // namespace fused

#ifndef CTC_FUSED_KIND
#error "compile with -DCTC_FUSED_KIND=0 (classic) or 1 (simplified): one translation unit per lattice variant"
#endif


template <int NL, int VPL>
static void launch_v(const Problem &p, const Layout &L, float *a, float *b, double *lp, float *loss, const float *d_loss,
                     float *grad, float *sink, hipStream_t st) {
  hipLaunchKernelGGL((fused::fused_kernel<CTC_FUSED_KIND, NL, VPL, true>), dim3(p.B), dim3(128), 0, st, p, L, a, b, lp,
                     loss, d_loss, grad, sink);
}

template <int NL>
static hipError_t launch_nl(const Problem &p, const Layout &L, float *a, float *b, double *lp, float *loss,
                            const float *d_loss, float *grad, float *sink, hipStream_t st) {
  switch (p.V / 256) {
    case 1: launch_v<NL, 1>(p, L, a, b, lp, loss, d_loss, grad, sink, st); break;
    case 2: launch_v<NL, 2>(p, L, a, b, lp, loss, d_loss, grad, sink, st); break;
    case 4: launch_v<NL, 4>(p, L, a, b, lp, loss, d_loss, grad, sink, st); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

#if CTC_FUSED_KIND == 0
hipError_t run_fused_classic
#else
hipError_t run_fused_simplified
#endif
    (const Problem &p, const Layout &L, char *ws, float *loss, const float *d_loss, float *grad, hipStream_t st) {
  float *alpha = reinterpret_cast<float *>(ws + L.off_alpha);
  float *beta = reinterpret_cast<float *>(ws + L.off_beta);
  double *logp = reinterpret_cast<double *>(ws + L.off_logp);
  float *sink = reinterpret_cast<float *>(ws + L.off_dummy);
  switch (L.NL) {
    case 1: return launch_nl<1>(p, L, alpha, beta, logp, loss, d_loss, grad, sink, st);
    case 2: return launch_nl<2>(p, L, alpha, beta, logp, loss, d_loss, grad, sink, st);
    case 4: return launch_nl<4>(p, L, alpha, beta, logp, loss, d_loss, grad, sink, st);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace ctc

// Device-side building blocks of the three-kernel (log-domain) pipeline, shared by ctc_kernels.hip (the kernels that wrap
// them) and ctc_hvp_fused.hip (which runs them inside its own launch for the utterances its linear-domain chains flag):
//   emit_row   one frame's compact emission row (log-softmax statistics + the gathers)             -> emit_kernel
//   Scan / scan_body   the sequential alpha / beta recursion of one (utterance, direction) by one wavefront  -> scan_kernel
// See ctc_kernels.hip for the algorithms and the reference lines.
#pragma once
#include "ctc_common.h"
#include "ctc_amd.h"

namespace ctc {

__device__ __forceinline__ int v1_clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

// one frame (b, t) by one wavefront (COH: the row is read by other workgroups of the same launch, ctc_common.h; MAXI: label
// positions per lane the caller can meet; DEPTH: 16-byte loads per lane in flight while the row is read)
template <bool COH = false, int MAXI = CTC_AMD_MAX_U / 64, int DEPTH = 8>
__device__ __forceinline__ void emit_row(const Problem &p, const Layout &L, float *__restrict__ emis, int b, int t, int lane) {
  const long row = (long)b * p.T + t;
  const int len = v1_clampi(p.logit_length[b], 0, p.T);
  if (t >= len) return;  // padded frames are never read downstream
  const int ll = p.label_length[b] < 0 ? 0 : p.label_length[b];
  const int V = p.V;
  // element accessor of this frame's row: float32 or bfloat16, any batch/time stride (producer formats)
  const long xoff = logits_off(p, b, t);
  const float *x = p.logits + xoff;                                                   // valid for float32 only
  const unsigned short *xh = reinterpret_cast<const unsigned short *>(p.logits) + xoff;  // valid for bfloat16 / float16 only
  const bool bf = p.xdtype != 0;
  auto xat = [&](int k) -> float { return bf ? h16_to_f32(xh[k], p.xdtype) : x[k]; };

  // the label tokens are requested FIRST, beside the row itself: fetched after the statistics they made the gathers below two
  // dependent round trips (label -> token -> logit) with nothing else of the wavefront in flight.  (The gathers themselves stay
  // behind the row pass, where they hit in L2: issued up front they were 64 scattered misses each.)
  int tokv[MAXI];
#pragma unroll
  for (int n = 0; n < MAXI; ++n) {
    const int i = lane + 64 * n;
    tokv[n] = -1;
    if (i < L.UP && i < ll) tokv[n] = (i < p.label_stride) ? p.labels[(long)b * p.label_stride + i] : p.blank;
  }
  const bool blank_ok = p.blank >= 0 && p.blank < V;

  float mx = -INFINITY, sum = 0.f, log2sum = 0.f;
  if (p.wrt == 0) {
    if (!bf && ((V | xoff) & 3) == 0 && (p.align_bits & 15) == 0) {
      // ONE pass over the row, eight 16-byte loads per lane in flight: every lane keeps a running (maximum, sum of
      // exp(x - maximum)) of its own columns, the lanes are combined once at the end (two passes with one load in flight
      // each read a V = 2048 row at 2.7 TB/s chip-wide)
      float m = -INFINITY, s = 0.f;
      for (int k0 = lane * 4; k0 < V; k0 += 256 * DEPTH) {
        float4 v[DEPTH];
#pragma unroll
        for (int q = 0; q < DEPTH; ++q) {
          const int k = k0 + 256 * q;
          v[q] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
          if (k < V) v[q] = *reinterpret_cast<const float4 *>(x + k);
        }
        float cm = m;
#pragma unroll
        for (int q = 0; q < DEPTH; ++q) cm = fmaxf(fmaxf(cm, fmaxf(v[q].x, v[q].y)), fmaxf(v[q].z, v[q].w));
        const float mr = (cm == -INFINITY) ? 0.f : cm;
        s *= fexp2((m - mr) * LOG2E);  // (m = -inf: s is 0 and stays 0)
#pragma unroll
        for (int q = 0; q < DEPTH; ++q)
          s += (fexp2((v[q].x - mr) * LOG2E) + fexp2((v[q].y - mr) * LOG2E)) + (fexp2((v[q].z - mr) * LOG2E) + fexp2((v[q].w - mr) * LOG2E));
        m = cm;
      }
      mx = wave_max(m);
      const float mref = (mx == -INFINITY) ? 0.f : mx;
      sum = s * fexp2((m - mref) * LOG2E);
      if (!(m > -INFINITY)) sum = 0.f;  // a lane whose columns are all -inf (or that holds none)
    } else {
      for (int k = lane; k < V; k += 64) mx = fmaxf(mx, xat(k));
      mx = wave_max(mx);
      const float mref = (mx == -INFINITY) ? 0.f : mx;
      for (int k = lane; k < V; k += 64) sum += fexp2((xat(k) - mref) * LOG2E);
    }
    sum = wave_sum(sum);
    log2sum = flog2(sum);  // -inf when the whole row is -inf: every emission becomes NEG below
    if (mx == -INFINITY) mx = 0.f;
  } else {
    mx = 0.f;
    log2sum = 0.f;
  }
  // log2 p(token k) = (x[k] - mx) * log2e - log2sum  (one rounding chain, no cancellation for huge logits)
  float *erow = emis + row * (long)L.ERS;
#pragma unroll
  for (int n = 0; n < MAXI; ++n) {
    const int i = lane + 64 * n;
    if (i < L.UP) {
      // (a label equal to the blank id is unsupported input in the reference; every tier treats it as an impossible
      // emission: the sample comes out infeasible, loss +inf, gradient 0)
      const bool ok = tokv[n] >= 0 && tokv[n] < V && tokv[n] != p.blank;
      float e = ok ? fmaxf((xat(ok ? tokv[n] : 0) - mx) * LOG2E - log2sum, NEG) : NEG;
      if (!(e == e)) e = NEG;
      st1<COH>(erow + i, e);
    }
  }
  if (lane == 0) {
    float bl = blank_ok ? fmaxf((xat(p.blank) - mx) * LOG2E - log2sum, NEG) : NEG;
    if (!(bl == bl)) bl = NEG;
    // softmax(x)[k] = exp2((x[k] - mx) * log2e - log2sum); kept as two terms so that huge logits cancel exactly
    st4<COH>(erow + L.UP, make_float4(bl, mx, log2sum, 0.f));
  }
}


template <int NL>
struct ERow {
  float y[NL];
  float bl;
};

template <int NL, bool COH = false>
__device__ __forceinline__ void load_erow(ERow<NL> &r, const float *__restrict__ base, int lane, int UP, int vz) {
  const float *p = base + lane * NL;
  if constexpr (NL == 1) {
    r.y[0] = ld1<COH>(p);
  } else if constexpr (NL == 2) {
    float2 v = ld2<COH>(p);
    r.y[0] = v.x; r.y[1] = v.y;
  } else {
#pragma unroll
    for (int q = 0; q < NL / 4; ++q) {
      float4 v = ld4<COH>(p + 4 * q);
      r.y[4 * q] = v.x; r.y[4 * q + 1] = v.y; r.y[4 * q + 2] = v.z; r.y[4 * q + 3] = v.w;
    }
  }
  // the blank emission is wave-uniform; fetched as a VECTOR load (vz = opaque zero): a scalar load returns out of
  // order, so its use forces lgkmcnt(0), i.e. a wait for the youngest prefetch of the ring instead of the oldest
  r.bl = ld1<COH>(base + UP + vz);
}

// store NL consecutive (a, b) pairs of this lane
template <int NL>
__device__ __forceinline__ void store_pairs(float *__restrict__ row, int lane, const float (&a)[NL], const float (&b)[NL]) {
  float *p = row + 2 * lane * NL;
  if constexpr (NL == 1) {
    *reinterpret_cast<float2 *>(p) = make_float2(a[0], b[0]);
  } else {
#pragma unroll
    for (int q = 0; q < NL / 2; ++q)
      *reinterpret_cast<float4 *>(p + 4 * q) = make_float4(a[2 * q], b[2 * q], a[2 * q + 1], b[2 * q + 1]);
  }
}
template <int NL>
__device__ __forceinline__ void store_singles(float *__restrict__ row, int lane, const float (&a)[NL]) {
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
}

// emission rows kept in flight per wave (= steps of one unrolled block = steps between exact renormalisations of the
// lattice row).  Fewer for long labels, whose rows fill the register file (16 rows of 16 positions per lane spilled
// 1256 VGPRs).
template <int NL>
struct ScanCfg { static constexpr int PF = NL <= 2 ? 16 : (NL == 4 ? 8 : (NL == 8 ? 4 : 2)); };

// One wavefront: blockIdx.x = utterance; DIR 0 = alpha (forward in t), 1 = beta (backward).
// Slot i = lane*NL + j is label position i (token label[i]).
//   classic alpha : c[j] = closed(l=i+1), o[j] = open(l=i+1), cx = closed(l=0)            (classic_ctc_loss.py:415-462)
//   classic beta  : c[j] = closed(l=i),   o[j] = open(l=i+1), cx = closed(l=UP)           (classic_ctc_loss.py:349-377)
//   simplified alpha: c[j] = a(l=i+1), cx = a(l=0); beta: c[j] = b(l=i), cx = b(l=UP)     (simplified_ctc_loss.py:327-438)
// With y[i] = log p(label[i]) the classic transition tables of classic_ctc_loss.py:464-563 reduce to
//   rep[l] = y[l-1],  yo[l] = y[l] unless label[l] == label[l-1]   (for labels free of the blank token).
// The steady-state loop is straight-line code (PF steps unrolled, no branches) so that hipcc emits counted
// s_waitcnt vmcnt(N): each step waits only for the emission row issued PF steps earlier, never for the
// prefetches and row stores still in flight.
// The lattice state is float64 since r04 (rows stay float32: one rounding where a row is written, none accumulated along the sweep):
// the float32 log-sum-exp chain was what this pipeline's gradient error consisted of -- 1.4e-4 / 1.9e-4 at T = 1000 with N(0, 4^2)
// logits, 1.4e-4 at T = 3000 (tests/tools/logdomain_error_model.py; ctc_common.h lse2(double, double)).
template <int KIND, int NL, int DIR>
struct Scan {
  using ST = double;
  ST c[NL], o[NL], cx;
  double off;
  float mpend = 0.f;  // renorm_lagged: the row maximum measured one step ago, not yet subtracted
  bool norep[NL], norep_next[NL];

  __device__ __forceinline__ void step(const ERow<NL> &e) {
    const float bl = e.bl;
    if constexpr (KIND == 0 && DIR == 0) {
      // alpha step (classic_ctc_loss.py:415-451)
      ST m[NL], x[NL];
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        m[j] = lse2(c[j], o[j]);
        x[j] = norep_next[j] ? m[j] : c[j];  // what position l+1 may continue from
      }
      ST xin0 = from_prev_lane(x[NL - 1], cx);
#pragma unroll
      for (int j = NL - 1; j >= 0; --j) {
        ST xin = (j == 0) ? xin0 : x[j - 1];
        o[j] = e.y[j] + lse2(o[j], xin);
        c[j] = bl + m[j];
      }
      cx += bl;
    } else if constexpr (KIND == 0 && DIR == 1) {
      // beta step (classic_ctc_loss.py:349-364)
      ST h[NL], ee[NL], pn[NL], x[NL];
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        h[j] = bl + c[j];
        ee[j] = e.y[j] + o[j];
        pn[j] = lse2(h[j], ee[j]);
        x[j] = norep[j] ? pn[j] : h[j];
      }
      cx += bl;
      ST xinl = from_next_lane(x[0], cx);
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        ST xin = (j == NL - 1) ? xinl : x[j + 1];
        o[j] = lse2(xin, ee[j]);
        c[j] = pn[j];
      }
    } else if constexpr (KIND == 1 && DIR == 0) {
      // simplified alpha step (simplified_ctc_loss.py:393-424)
      ST pin0 = from_prev_lane(c[NL - 1], cx);
#pragma unroll
      for (int j = NL - 1; j >= 0; --j) {
        ST pin = (j == 0) ? pin0 : c[j - 1];
        c[j] = lse2(bl + c[j], e.y[j] + pin);
      }
      cx += bl;
    } else {
      // simplified beta step (simplified_ctc_loss.py:327-343)
      ST nin = from_next_lane(c[0], cx);
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        ST nx = (j == NL - 1) ? nin : c[j + 1];
        c[j] = lse2(bl + c[j], e.y[j] + nx);
      }
      cx += bl;
    }
  }

  // exact renormalisation: subtract the row maximum, remember it in `off` (branch-free)
  __device__ __forceinline__ void renorm() {
    float mx = (float)cx;  // (any common shift will do: the float32 image of the maximum)
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      mx = fmaxf(mx, (float)c[j]);
      if constexpr (KIND == 0) mx = fmaxf(mx, (float)o[j]);
    }
    mx = wave_max(mx);
    mx = (mx > NEG_THR) ? mx : 0.f;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      c[j] -= mx;
      if constexpr (KIND == 0) o[j] -= mx;
    }
    cx -= mx;
    off += (double)mx;
    mpend = 0.f;
  }

  // Per-step renormalisation off the critical path: subtracts the row maximum measured ONE STEP AGO (its wave reduction ran beside
  // this step's log-sum-exps; a state rises by at most log2(3) per step) and measures the present one for the next step.  The state
  // itself does not need it (float64) -- the ROWS do: they are stored in float32, and a row written 15 steps after the last exact
  // renormalisation carries magnitudes of hundreds on sharp logits, i.e. a rounding of 3e-5 per entry, which the tangent sweep of
  // the Hessian-vector product (weights 2^(argument - result) from consecutive rows) accumulates (r04: one redone utterance of
  // ~8 000 at 1.13e-4 of max|Hv|, tests/tools/soak_hvp.py seed 21).
  __device__ __forceinline__ void renorm_lagged() {
    const float m = mpend;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      c[j] -= m;
      if constexpr (KIND == 0) o[j] -= m;
    }
    cx -= m;
    off += (double)m;
    float mx = (float)cx;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      mx = fmaxf(mx, (float)c[j]);
      if constexpr (KIND == 0) mx = fmaxf(mx, (float)o[j]);
    }
    mx = wave_max(mx);
    mpend = (mx > NEG_THR) ? mx : 0.f;
  }

  // row layout: see Layout in ctc_common.h.  The 16-byte tail (l = 0 state + offset) is wave-uniform data
  // written by every lane to the same address, which keeps the store branch-free (ONE: by lane 0 alone -- rows staged in LDS,
  // where 64 writes to one address would queue up, ctc_wide.hip).
  template <bool ONE = false>
  __device__ __forceinline__ void store_row(float *__restrict__ row, int lane, int UP) const {
    const float oh = (float)off;
    const float ol = (float)(off - (double)oh);
    float cf[NL], of[NL];  // (rows are float32)
#pragma unroll
    for (int j = 0; j < NL; ++j) { cf[j] = (float)c[j]; of[j] = (float)o[j]; }
    const float cxf = (float)cx;
    if constexpr (DIR == 0) {
      if constexpr (KIND == 0) {
        store_pairs<NL>(row, lane, cf, of);
        if (!ONE || lane == 0) *reinterpret_cast<float4 *>(row + 2 * UP) = make_float4(cxf, NEG, oh, ol);
      } else {
        store_singles<NL>(row, lane, cf);
        if (!ONE || lane == 0) *reinterpret_cast<float4 *>(row + UP) = make_float4(cxf, 0.f, oh, ol);
      }
    } else {
      float cs[NL];  // state of label position l = i+1 lives in the next slot's c
#pragma unroll
      for (int j = 0; j < NL - 1; ++j) cs[j] = cf[j + 1];
      cs[NL - 1] = from_next_lane(cf[0], cxf);
      const float c00 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(cf[0])));  // state l = 0 (lane 0, slot 0)
      if constexpr (KIND == 0) {
        store_pairs<NL>(row, lane, cs, of);
        if (!ONE || lane == 0) *reinterpret_cast<float4 *>(row + 2 * UP) = make_float4(c00, c00, oh, ol);
      } else {
        store_singles<NL>(row, lane, cs);
        if (!ONE || lane == 0) *reinterpret_cast<float4 *>(row + UP) = make_float4(c00, 0.f, oh, ol);
      }
    }
  }
};

// FINE: the rows are renormalised every step (Scan::renorm_lagged) -- for the tangent sweep of the Hessian-vector product, which
// reads consecutive rows against each other; costs the scan ~10 % and is off for the loss / gradient pipeline.
template <int KIND, int NL, int DIR, bool FINE = false>
__device__ __forceinline__ void scan_body(const Problem &p, const Layout &L, const float *__restrict__ emis,
                                          float *__restrict__ rows_all, double *__restrict__ logp,
                                          float *__restrict__ loss, int b, int lane) {
  const int T = p.T, UP = L.UP;
  const int len = v1_clampi(p.logit_length[b], 0, T);
  int ll = p.label_length[b] < 0 ? 0 : p.label_length[b];
  if (ll > p.U) {  // contract violation: reported as an infeasible sample
    if (DIR == 0 && lane == 0) { logp[b] = -INFINITY; loss[b] = INFINITY; }
    return;
  }
  Scan<KIND, NL, DIR> S;
  S.off = 0.0;
  {
    const int32_t *lab = p.labels + (long)b * p.label_stride;
    auto tok = [&](int i) -> int { return (i >= 0 && i < ll) ? ((i < p.label_stride) ? lab[i] : p.blank) : -1 - (i < 0); };
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      int i = lane * NL + j;
      S.norep[j] = (i == 0) || tok(i) != tok(i - 1);
      S.norep_next[j] = tok(i + 1) != tok(i);
      S.c[j] = NEG;
      S.o[j] = NEG;
    }
  }
  float *rows = rows_all + (long)b * (T + 1) * L.SRS;
  const float *ebase = emis + (long)b * T * L.ERS;

  // ---- initial row ----
  if constexpr (DIR == 0) {
    S.cx = 0.f;  // alpha[0]: only (l=0, closed) is reachable (classic_ctc_loss.py:453-462, simplified_ctc_loss.py:426-438)
  } else {
    // beta[len]: one-hot at l = label_length, both states (classic_ctc_loss.py:366-377, simplified_ctc_loss.py:345-356)
    S.cx = (ll == UP) ? 0.f : NEG;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      int i = lane * NL + j;
      if (i == ll) S.c[j] = 0.f;
      if (KIND == 0 && i == ll - 1) S.o[j] = 0.f;
    }
  }
  S.store_row(rows + (long)(DIR == 0 ? 0 : len) * L.SRS, lane, UP);

  // step k consumes emission row t = k (alpha) / len-1-k (beta) and produces lattice row k+1 / len-1-k
  auto erow_ptr = [&](int k) -> const float * {
    int kk = k < len ? k : len - 1;
    int t = (DIR == 0) ? kk : (len - 1 - kk);
    return ebase + (long)t * L.ERS;
  };
  auto out_row = [&](int k) -> float * { return rows + (long)(DIR == 0 ? k + 1 : len - 1 - k) * L.SRS; };

  if (len > 0) {
    int vz;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vz));
    constexpr int PF = ScanCfg<NL>::PF, RENORM = PF;
    ERow<NL> buf[PF];
#pragma unroll
    for (int d = 0; d < PF; ++d) load_erow<NL>(buf[d], erow_ptr(d), lane, UP, vz);
    int k0 = 0;
    for (; k0 + PF <= len; k0 += PF) {
#pragma unroll
      for (int d = 0; d < PF; ++d) {
        S.step(buf[d]);
        // (the refill stays BEHIND the step that uses the slot: hoisted above it by the scheduler, old and new row were alive together,
        // the slot changed registers every trip and the copies at the loop's edge waited for every load of the trip -- s_waitcnt
        // vmcnt(0) at the head of each unrolled block, r04, after the float64 state had changed the register allocation)
        __builtin_amdgcn_sched_barrier(0);
        load_erow<NL>(buf[d], erow_ptr(k0 + d + PF), lane, UP, vz);  // clamped: re-reads the last row near the end
        if (d == RENORM - 1) S.renorm();
        else if constexpr (FINE) S.renorm_lagged();
#ifndef CTC_EXPERIMENT_NO_STORE
        S.store_row(out_row(k0 + d), lane, UP);
#endif
      }
    }
    // tail: fewer than PF steps left, their rows are already in buf[0 .. len-k0)
#pragma unroll
    for (int d = 0; d < PF; ++d) {
      if (k0 + d < len) {
        S.step(buf[d]);
        if constexpr (FINE) S.renorm_lagged();
        S.store_row(out_row(k0 + d), lane, UP);
      }
    }
  }

  if constexpr (DIR == 0) {
    // loss = -alpha[len, label_length] (classic_ctc_loss.py:152-165, simplified_ctc_loss.py:73-83)
    float mine = NEG;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      int i = lane * NL + j;
      if (i == ll - 1) mine = (float)((KIND == 0) ? lse2(S.c[j], S.o[j]) : S.c[j]);
    }
    float v = (ll == 0) ? (float)S.cx : wave_max(mine);
    if (lane == 0) {
      if (v > NEG_THR) {
        double lp2 = (double)v + S.off;
        logp[b] = lp2;
        loss[b] = (float)(-lp2 * LN2_D);
      } else {
        logp[b] = -INFINITY;
        loss[b] = INFINITY;
      }
    }
  }
}


}  // namespace ctc

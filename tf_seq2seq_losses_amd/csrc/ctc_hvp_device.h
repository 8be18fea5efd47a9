// Device-side building blocks of the log-domain Hessian-vector pipeline, shared by ctc_hvp.hip (the kernels that wrap them)
// and ctc_hvp_fused.hip (which runs them inside its own launch for the utterances its linear-domain chains flag).
// See ctc_hvp.hip for the mathematics.
#pragma once
#include "ctc_fused_common.h"

namespace ctc {

using namespace ctc::fused;

// Extra workspace of the HVP (after Layout::off_extra): tangent emissions [B][T][ERS], tangent lattice rows
// dalpha/dbeta [B][T+1][SRS] (same layouts as the value rows), dlogP [B].
struct HvpLayout {
  size_t off_demis, off_dalpha, off_dbeta, off_dlogp, total;
};
__host__ __device__ static inline HvpLayout make_hvp_layout(const Layout &L, int B, int T) {
  auto al = [](size_t x) { return (x + 255) & ~size_t(255); };
  HvpLayout H;
  size_t o = 0;
  H.off_demis = o;  o = al(o + (size_t)B * T * L.ERS * 4);
  H.off_dalpha = o; o = al(o + (size_t)B * (T + 1) * L.SRS * 4);
  H.off_dbeta = o;  o = al(o + (size_t)B * (T + 1) * L.SRS * 4);
  H.off_dlogp = o;  o = al(o + (size_t)B * 4);
  H.total = o;
  return H;
}
// tangent emissions: dE[i] = u[label[i]] (0 beyond the label), [UP] = u[blank], [UP+1] = s . v (logits mode)
__device__ __forceinline__ void temit_row(const Problem &p, const Layout &L, const float *__restrict__ emis, const float *__restrict__ vec,
                                          float *__restrict__ demis, int b, int t, int lane) {
  const long row = (long)b * p.T + t;
  const int len = clampi(p.logit_length[b], 0, p.T);
  if (t >= len) return;
  const int ll = p.label_length[b] < 0 ? 0 : p.label_length[b];
  const int V = p.V;
  const float *v = vec + row * (long)V;
  float sv = 0.f;
  if (p.wrt == 0) {  // u = J v = v - (softmax . v): direction in log-probability space (tools.py:37-39 differentiated)
    const float *x = p.logits + row * (long)V;
    const float mx = emis[row * (long)L.ERS + L.UP + 1], l2s = emis[row * (long)L.ERS + L.UP + 2];
    if (((V & 3) | (int)((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(v)) & 15)) == 0) {  // 16 bytes per lane
      for (int k = lane * 4; k < V; k += 256) {
        const float4 xv = *reinterpret_cast<const float4 *>(x + k), vv = *reinterpret_cast<const float4 *>(v + k);
        sv += (fexp2((xv.x - mx) * LOG2E - l2s) * vv.x + fexp2((xv.y - mx) * LOG2E - l2s) * vv.y) +
              (fexp2((xv.z - mx) * LOG2E - l2s) * vv.z + fexp2((xv.w - mx) * LOG2E - l2s) * vv.w);
      }
    } else {
      for (int k = lane; k < V; k += 64) sv += fexp2((x[k] - mx) * LOG2E - l2s) * v[k];
    }
    sv = wave_sum(sv);
  }
  float *drow = demis + row * (long)L.ERS;
  for (int i = lane; i < L.UP; i += 64) {
    float d = 0.f;
    if (i < ll) {
      int tok = (i < p.label_stride) ? p.labels[(long)b * p.label_stride + i] : p.blank;
      if (tok >= 0 && tok < V) d = v[tok] - sv;
    }
    drow[i] = d;
  }
  if (lane == 0) {
    drow[L.UP] = v[p.blank] - sv;
    drow[L.UP + 1] = sv;
    drow[L.UP + 2] = 0.f;
    drow[L.UP + 3] = 0.f;
  }
}

// One wavefront per (utterance, direction): tangent sweep.  Slot/state conventions = Scan in ctc_kernels.hip.
template <int KIND, int NL>
__device__ __forceinline__ void tscan_body(const Problem &p, const Layout &L, const float *__restrict__ emis,
                                           const float *__restrict__ demis, const float *__restrict__ alpha,
                                           const float *__restrict__ beta, const double *__restrict__ logp,
                                           float *__restrict__ dalpha, float *__restrict__ dbeta,
                                           float *__restrict__ dlogp, int b, int dir, int lane) {
  const int T = p.T, UP = L.UP;
  const int len = clampi(p.logit_length[b], 0, T);
  int ll = p.label_length[b] < 0 ? 0 : p.label_length[b];
  if (ll > p.U || logp[b] == -INFINITY) {
    if (dir == 0 && lane == 0) dlogp[b] = 0.f;
    return;  // infeasible: hvp_out_kernel writes zeros
  }
  constexpr int PAIR = (KIND == 0) ? 2 : 1;
  const int tailpos = PAIR * UP;
  bool norep[NL], norep_next[NL];
  {
    const int32_t *lab = p.labels + (long)b * p.label_stride;
    auto tok = [&](int i) -> int { return (i >= 0 && i < ll) ? ((i < p.label_stride) ? lab[i] : p.blank) : -1 - (i < 0); };
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      int i = lane * NL + j;
      norep[j] = (i == 0) || tok(i) != tok(i - 1);
      norep_next[j] = tok(i + 1) != tok(i);
    }
  }
  const float *vrows = (dir == 0 ? alpha : beta) + (long)b * (T + 1) * L.SRS;
  float *drows = (dir == 0 ? dalpha : dbeta) + (long)b * (T + 1) * L.SRS;
  const float *erows = emis + (long)b * T * L.ERS;
  const float *derows = demis + (long)b * T * L.ERS;

  struct Row { float c[NL], o[NL], cx; double off; };  // a value row in the chain's NATIVE layout
  struct Em { float y[NL], bl, dy[NL], dbl; };
  // One step's inputs as loaded (conversions need DPP and therefore the data: they are done at use, so that PF steps of
  // loads can be in flight).  Wave-uniform values come through vector loads as well (vz: an opaque zero), which keeps
  // them on vmcnt with the rest instead of serialising on the scalar cache.
  struct Raw { float a[NL], bb[NL], tl, oh, ol, y[NL], bl, dy[NL], dbl; };
  int vz;
  asm volatile("v_mov_b32 %0, 0" : "=v"(vz));
  auto load_row_raw = [&](Raw &w, int trow) {
    const float *q = vrows + (long)trow * L.SRS;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const int i = lane * NL + j;
      if constexpr (KIND == 0) { float2 v = *reinterpret_cast<const float2 *>(q + 2 * i); w.a[j] = v.x; w.bb[j] = v.y; }
      else { w.a[j] = q[i]; w.bb[j] = NEG; }
    }
    w.tl = q[tailpos + vz]; w.oh = q[tailpos + 2 + vz]; w.ol = q[tailpos + 3 + vz];
  };
  auto load_raw = [&](Raw &w, int k) {  // inputs of step k: emissions of the frame it consumes, the value row it produces
    const int kk = k < len ? k : len - 1;
    const int t = dir == 0 ? kk : len - 1 - kk;
    load_row_raw(w, dir == 0 ? t + 1 : t);
    const float *q = erows + (long)t * L.ERS, *dq = derows + (long)t * L.ERS;
#pragma unroll
    for (int j = 0; j < NL; ++j) { w.y[j] = q[lane * NL + j]; w.dy[j] = dq[lane * NL + j]; }
    w.bl = q[UP + vz];
    w.dbl = dq[UP + vz];
  };
  auto to_row = [&](const Raw &w, Row &r) {
    if (dir == 0) {  // alpha rows are stored in native layout
#pragma unroll
      for (int j = 0; j < NL; ++j) { r.c[j] = w.a[j]; r.o[j] = w.bb[j]; }
      r.cx = w.tl;
    } else {  // beta rows: slot i = state l = i+1, tail = l = 0  ->  native: slot i = state_c(l = i), cx = l = UP
#pragma unroll
      for (int j = NL - 1; j > 0; --j) r.c[j] = w.a[j - 1];
      r.c[0] = from_prev_lane(w.a[NL - 1], w.tl);
      r.cx = readlane_f(w.a[NL - 1], 63);
#pragma unroll
      for (int j = 0; j < NL; ++j) r.o[j] = w.bb[j];
    }
    r.off = (double)w.oh + (double)w.ol;
  };
  // tangent state (natural-log units), native layout
  float dc[NL], dob[NL], dcx = 0.f;
#pragma unroll
  for (int j = 0; j < NL; ++j) { dc[j] = 0.f; dob[j] = 0.f; }
  auto store_tangent = [&](int t) {
    float *q = drows + (long)t * L.SRS;
    if (dir == 0) {
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        const int i = lane * NL + j;
        if constexpr (KIND == 0) *reinterpret_cast<float2 *>(q + 2 * i) = make_float2(dc[j], dob[j]);
        else q[i] = dc[j];
      }
      if (lane == 0) q[tailpos] = dcx;
    } else {  // back to the stored beta layout: slot i = l = i+1 (next slot's c), tail = l = 0
      float cs[NL];
#pragma unroll
      for (int j = 0; j < NL - 1; ++j) cs[j] = dc[j + 1];
      cs[NL - 1] = from_next_lane(dc[0], dcx);
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        const int i = lane * NL + j;
        if constexpr (KIND == 0) *reinterpret_cast<float2 *>(q + 2 * i) = make_float2(cs[j], dob[j]);
        else q[i] = cs[j];
      }
      if (lane == 0) q[tailpos] = dc[0];
    }
  };
  // Weight of the first argument of a two-way log-sum-exp whose result `res` is known; the second weight is taken as
  // 1 - w so the pair sums to one exactly: tangents carry a large common component (the running sum of the blank
  // direction) that must pass through unchanged, d = d2 + w (d1 - d2).  Unreachable results carry no weight.
  auto w = [](float arg, float res) -> float { return res > NEG_THR ? fminf(fexp2(arg - res), 1.0f) : 0.f; };
  Row prev;
  {
    Raw w0;
    load_row_raw(w0, dir == 0 ? 0 : len);
    to_row(w0, prev);
  }
  store_tangent(dir == 0 ? 0 : len);  // boundary rows have zero tangent
  // steps of look-ahead (register ring, statically indexed: the loop is unrolled by TPF); shallower for long labels, whose
  // rows fill the register file
  constexpr int TPF = NL <= 2 ? 8 : (NL == 4 ? 4 : (NL == 8 ? 2 : 1));
  Raw ring[TPF];
  if (len > 0) {
    static_for<0, TPF>([&](auto D) { load_raw(ring[decltype(D)::value], decltype(D)::value); });
    int k0 = 0;
    auto step = [&](auto D) __attribute__((always_inline)) {
      constexpr int d = decltype(D)::value;
      const int k = k0 + d;
      {
        const int t = dir == 0 ? k : len - 1 - k;           // frame consumed
        const int tres = dir == 0 ? t + 1 : t;              // row produced
        Row next;
        Em e;
        to_row(ring[d], next);
#pragma unroll
        for (int j = 0; j < NL; ++j) { e.y[j] = ring[d].y[j]; e.dy[j] = ring[d].dy[j]; }
        e.bl = ring[d].bl;
        e.dbl = ring[d].dbl;
        load_raw(ring[d], k + TPF);                         // refill the slot (clamped at the end)
      const float doff = (float)(prev.off - next.off);    // rows carry different renormalisation offsets
      if constexpr (KIND == 0) {
        if (dir == 0) {
          // alpha step (classic_ctc_loss.py:415-451): m = c (+) o, c' = bl + m, o' = y + (o (+) xin)
          float dm[NL], x[NL], dx[NL];
#pragma unroll
          for (int j = 0; j < NL; ++j) {
            const float m = next.c[j] - e.bl - doff;      // m in the units of the previous row
            dm[j] = dob[j] + w(prev.c[j], m) * (dc[j] - dob[j]);
            x[j] = norep_next[j] ? m : prev.c[j];
            dx[j] = norep_next[j] ? dm[j] : dc[j];
          }
          const float xin0 = from_prev_lane(x[NL - 1], prev.cx), dxin0 = from_prev_lane(dx[NL - 1], dcx);
#pragma unroll
          for (int j = NL - 1; j >= 0; --j) {
            const float xin = (j == 0) ? xin0 : x[j - 1], dxin = (j == 0) ? dxin0 : dx[j - 1];
            const float Lr = next.o[j] - e.y[j] - doff;
            dob[j] = e.dy[j] + dxin + w(prev.o[j], Lr) * (dob[j] - dxin);
            dc[j] = e.dbl + dm[j];
          }
          dcx += e.dbl;
        } else {
          // beta step (classic_ctc_loss.py:349-364): c' = h (+) ee, o' = xin (+) ee, h = bl + c, ee = y + o
          float h[NL], ee[NL], dh[NL], dee[NL], x[NL], dx[NL], dpn[NL];
#pragma unroll
          for (int j = 0; j < NL; ++j) {
            h[j] = e.bl + prev.c[j] + doff;               // arguments in the units of the produced row
            ee[j] = e.y[j] + prev.o[j] + doff;
            dh[j] = e.dbl + dc[j];
            dee[j] = e.dy[j] + dob[j];
            dpn[j] = dee[j] + w(h[j], next.c[j]) * (dh[j] - dee[j]);
            x[j] = norep[j] ? next.c[j] : h[j];
            dx[j] = norep[j] ? dpn[j] : dh[j];
          }
          const float cxn = prev.cx + e.bl + doff, dcxn = dcx + e.dbl;
          const float xinl = from_next_lane(x[0], cxn), dxinl = from_next_lane(dx[0], dcxn);
#pragma unroll
          for (int j = 0; j < NL; ++j) {
            const float xin = (j == NL - 1) ? xinl : x[j + 1], dxin = (j == NL - 1) ? dxinl : dx[j + 1];
            dob[j] = dee[j] + w(xin, next.o[j]) * (dxin - dee[j]);
            dc[j] = dpn[j];
          }
          dcx = dcxn;
        }
      } else {
        if (dir == 0) {
          // simplified alpha step: a'(l=i+1) = (bl + a(i+1)) (+) (y_i + a(i))   (simplified_ctc_loss.py:393-424)
          const float pin0 = from_prev_lane(prev.c[NL - 1], prev.cx), dpin0 = from_prev_lane(dc[NL - 1], dcx);
#pragma unroll
          for (int j = NL - 1; j >= 0; --j) {
            const float pin = (j == 0) ? pin0 : prev.c[j - 1], dpin = (j == 0) ? dpin0 : dc[j - 1];
            const float r = next.c[j] - doff;
            const float d2 = e.dy[j] + dpin;
            dc[j] = d2 + w(e.bl + prev.c[j], r) * (e.dbl + dc[j] - d2);
          }
          dcx += e.dbl;
        } else {
          // simplified beta step: b'(l=i) = (bl + b(i)) (+) (y_i + b(i+1))       (simplified_ctc_loss.py:327-343)
          const float nin = from_next_lane(prev.c[0], prev.cx), dnin = from_next_lane(dc[0], dcx);
#pragma unroll
          for (int j = 0; j < NL; ++j) {
            const float nx = (j == NL - 1) ? nin : prev.c[j + 1], dnx = (j == NL - 1) ? dnin : dc[j + 1];
            const float r = next.c[j] - doff;
            const float d2 = e.dy[j] + dnx;
            dc[j] = d2 + w(e.bl + prev.c[j], r) * (e.dbl + dc[j] - d2);
          }
          dcx += e.dbl;
        }
      }
        store_tangent(tres);
        prev = next;
      }
    };
    for (; k0 + TPF <= len; k0 += TPF) static_for<0, TPF>(step);
    static_for<0, TPF>([&](auto D) {
      if (k0 + decltype(D)::value < len) step(D);
    });
  }
  if (dir == 0) {
    // dlogP = tangent of alpha[len, label_length] (classic: of closed (+) open there)
    float mine = 0.f;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const int i = lane * NL + j;
      if (i == ll - 1) {
        if constexpr (KIND == 0) {
          const float r = lse2(prev.c[j], prev.o[j]);
          mine = dob[j] + w(prev.c[j], r) * (dc[j] - dob[j]);
        } else {
          mine = dc[j];
        }
      }
    }
    mine = wave_sum(mine);
    if (lane == 0) dlogp[b] = (ll == 0) ? dcx : mine;
  }
}

// One wavefront per frame: d(posterior) scattered by token, chain rule through log-softmax for logits.
template <int KIND>
__device__ __forceinline__ void hvp_out_row(const Problem &p, const Layout &L, const float *__restrict__ emis,
                                            const float *__restrict__ demis, const float *__restrict__ alpha,
                                            const float *__restrict__ beta, const float *__restrict__ dalpha,
                                            const float *__restrict__ dbeta, const double *__restrict__ logp,
                                            const float *__restrict__ dlogp, const float *__restrict__ vec,
                                            float *__restrict__ out, float *bin, int b, int t, int lane) {
  const long row = (long)b * p.T + t;
  const int V = p.V, UP = L.UP;
  float *o = out + row * (long)V;
  const int len = clampi(p.logit_length[b], 0, p.T);
  const double lp = logp[b];
  int ll = p.label_length[b] < 0 ? 0 : p.label_length[b];
  if (t >= len || lp == -INFINITY || ll > p.U) {
    for (int k = lane; k < V; k += 64) __builtin_nontemporal_store(0.f, o + k);  // the Hessian vanishes there (base_loss.py:240-258)
    return;
  }
  for (int k = lane; k < V; k += 64) bin[k] = 0.f;
  wave_lds_fence();
  constexpr int PAIR = (KIND == 0) ? 2 : 1;
  const int tailpos = PAIR * UP;
  const int32_t *lab = p.labels + (long)b * p.label_stride;
  const long ra_i = ((long)b * (p.T + 1) + (KIND == 0 ? t + 1 : t)) * L.SRS, rb_i = ((long)b * (p.T + 1) + t + 1) * L.SRS;
  const float *ra = alpha + ra_i, *rb = beta + rb_i, *da = dalpha + ra_i, *db = dbeta + rb_i;
  // Posterior q_s = alpha beta / P of a lattice state and its tangent dq_s = q_s (d_s - dlogP), d_s = dalpha_s + dbeta_s (+ the
  // tangents of the emissions between the two rows).  Both are taken relative to the frame's OWN mass (r03; the gradient kernels
  // do the same, ctc_grad_row.h): q_s = e_s / sum_r e_r and dlogP = sum_r q_r d_r -- identities that hold for every frame (every
  // alignment passes through exactly one state per frame), so the row offsets cancel before anything is added, the posterior
  // tangents of a frame sum to zero exactly, and what a 1000-step float32 sweep has accumulated in rounding -- in the values and in
  // the common component of the tangents -- drops out (error against float64 at T = 1000: 2.3-2.9e-4 -> the 1e-5 class).
  // Three passes over the four rows (L1 hits): maximum; mass and mean tangent; scatter.
  (void)dlogp;
  const float *er = emis + row * (long)L.ERS, *der = demis + row * (long)L.ERS;
  const float bl = (KIND == 1) ? er[UP] : 0.f, dbl = (KIND == 1) ? der[UP] : 0.f;
  // (log2 weight, tangent) of the blank part and of the token part of label position i
  auto terms = [&](int i, float &tb, float &db_, float &tt, float &dt_) {
    if constexpr (KIND == 0) {
      const float2 a = *reinterpret_cast<const float2 *>(ra + 2 * i), bb = *reinterpret_cast<const float2 *>(rb + 2 * i);
      const float2 ta = *reinterpret_cast<const float2 *>(da + 2 * i), tbb = *reinterpret_cast<const float2 *>(db + 2 * i);
      tb = a.x + bb.x; db_ = ta.x + tbb.x;
      tt = (i < ll) ? a.y + bb.y : NEG; dt_ = ta.y + tbb.y;
    } else {
      const float ai = ra[i], bi = rb[i];  // state l = i+1 in both rows
      tb = ai + bi + bl; db_ = da[i] + db[i] + dbl;
      const float aprev = (i == 0) ? ra[UP] : ra[i - 1], daprev = (i == 0) ? da[UP] : da[i - 1];
      tt = (i < ll) ? aprev + er[i] + bi : NEG; dt_ = daprev + der[i] + db[i];
    }
  };
  const float t0 = (KIND == 0) ? ra[2 * UP] + rb[2 * UP] : ra[UP] + rb[UP] + bl;          // the l = 0 state
  const float d0 = (KIND == 0) ? da[2 * UP] + db[2 * UP] : da[UP] + db[UP] + dbl;
  float m = t0;
  for (int i = lane; i < UP; i += 64) {
    float tb, db_, tt, dt_;
    terms(i, tb, db_, tt, dt_);
    m = fmaxf(m, fmaxf(tb, tt));
  }
  m = wave_max(m);
  if (!(m > NEG_THR)) {  // no alignment passes through this frame (cannot happen on a feasible sample)
    for (int k = lane; k < V; k += 64) __builtin_nontemporal_store(0.f, o + k);
    return;
  }
  float ssum = 0.f, dsum = 0.f;
  if (lane == 0) { const float e0 = fexp2(t0 - m); ssum = e0; dsum = e0 * d0; }
  for (int i = lane; i < UP; i += 64) {
    float tb, db_, tt, dt_;
    terms(i, tb, db_, tt, dt_);
    const float eb = fexp2(tb - m), et = fexp2(tt - m);  // (NEG - m underflows to 0)
    ssum += eb + et;
    dsum += eb * db_ + ((i < ll) ? et * dt_ : 0.f);
  }
  const float inv = 1.0f / wave_sum(ssum);  // the sum is >= 1: the maximum contributes 2^0
  const float dmean = wave_sum(dsum) * inv;  // d log(mass) = dlogP
  float dblank = (lane == 0) ? fexp2(t0 - m) * inv * (d0 - dmean) : 0.f;
  for (int i = lane; i < UP; i += 64) {
    float tb, db_, tt, dt_;
    terms(i, tb, db_, tt, dt_);
    dblank += fexp2(tb - m) * inv * (db_ - dmean);
    if (i < ll) {
      const float dq = fexp2(tt - m) * inv * (dt_ - dmean);
      const int tok = (i < p.label_stride) ? lab[i] : p.blank;
      if (tok >= 0 && tok < V && tok != p.blank) atomicAdd(&bin[tok], dq);
    }
  }
  dblank = wave_sum(dblank);
  if (lane == 0) bin[p.blank] = dblank;
  wave_lds_fence();
  if (p.wrt == 0) {
    const float *x = p.logits + row * (long)V, *v = vec + row * (long)V;
    const float mx = emis[row * (long)L.ERS + UP + 1], l2s = emis[row * (long)L.ERS + UP + 2];
    const float sv = demis[row * (long)L.ERS + UP + 1];
    if (((V & 3) | (int)((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(o)) & 15)) == 0) {
      typedef float v4f __attribute__((ext_vector_type(4)));
      for (int k = lane * 4; k < V; k += 256) {  // 16 bytes per lane in and out
        const float4 xv = *reinterpret_cast<const float4 *>(x + k), vv = *reinterpret_cast<const float4 *>(v + k);
        const float4 bq = *reinterpret_cast<const float4 *>(bin + k);
        const v4f r = {-bq.x + fexp2((xv.x - mx) * LOG2E - l2s) * (vv.x - sv), -bq.y + fexp2((xv.y - mx) * LOG2E - l2s) * (vv.y - sv),
                       -bq.z + fexp2((xv.z - mx) * LOG2E - l2s) * (vv.z - sv), -bq.w + fexp2((xv.w - mx) * LOG2E - l2s) * (vv.w - sv)};
        __builtin_nontemporal_store(r, reinterpret_cast<v4f *>(o + k));
      }
    } else {
      for (int k = lane; k < V; k += 64) {
        const float s = fexp2((x[k] - mx) * LOG2E - l2s);
        __builtin_nontemporal_store(-bin[k] + s * (v[k] - sv), o + k);  // H_lp u + (diag(s) - s s^T) v
      }
    }
  } else {
    for (int k = lane; k < V; k += 64) __builtin_nontemporal_store(-bin[k], o + k);
  }
}

}  // namespace ctc

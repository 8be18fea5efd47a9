// One row of the wide-vocabulary gradient (V > 1024, float32 rows, 16-byte aligned), shared by the three-kernel pipeline
// (ctc_kernels.hip: grad_wide_kernel) and the one-launch tier (ctc_wide.hip).
#pragma once
#include "ctc_common.h"
#include "ctc_amd.h"
#include "ctc_v1_device.h"

namespace ctc {

// One gradient row by one wavefront, the vocabulary walked in passes of 1024 columns: the posterior of every label position is
// computed ONCE into a per-wavefront table; each pass zeroes a 4 KB bin array, adds the positions whose token falls into it (fixed
// point, integer LDS atomics) and streams its 1024 columns (logits in, gradient out, the next pass's logits already requested).
// `bins` (1024 words) and `qtab` (64 NL words) are this wavefront's LDS.
//
// The posterior of "frame t emits token k" is normalised by the frame's OWN mass sum_s alpha_t[s] beta_t[s] -- which equals P for
// every t (the invariant the reference tests in tests/test_classic_ctc_loss.py:146-167) -- instead of the P the alpha sweep ends
// with: the row offsets cancel (no double-precision sums), the rounding of a 1000-step float32 sweep no longer enters as a common
// factor (gradient error against float64 at T = 1000, V = 4096: 4.2e-5 instead of 1.6e-4), and a row needs nothing from the END of a
// sweep (ctc_wide.hip computes rows while the sweeps are still running).
//   COH: the lattice / emission rows come from other workgroups of the same launch (ctc_common.h ld1 / ld2)
//   wait_rows(len): called once the first logits have been requested, before the lattice rows are read (ctc_wide.hip: waits
//   for the sweeps to have reached this frame)
template <int KIND, int NL, bool COH, class WAIT>
__device__ __forceinline__ void grad_row(const Problem &p, const Layout &L, const float *__restrict__ emis,
                                         const float *__restrict__ alpha, const float *__restrict__ beta,
                                         const float *__restrict__ d_loss, float *__restrict__ grad, int b, int t, int lane,
                                         unsigned *bins, unsigned *qtab, WAIT wait_rows) {
  constexpr int CH = 1024;
  const int V = p.V, UP = L.UP;
  const long row = (long)b * p.T + t;
  float *g = grad + grad_off(p, b, t);
  typedef float v4f __attribute__((ext_vector_type(4)));
  auto gput4 = [&](int k, float4 r) {
    v4f v = {r.x, r.y, r.z, r.w};
    __builtin_nontemporal_store(v, reinterpret_cast<v4f *>(g + k));
  };
  const int len = v1_clampi(p.logit_length[b], 0, p.T);
  const int ll = p.label_length[b] < 0 ? 0 : p.label_length[b];
  if (t >= len && p.row0 != nullptr) return;  // packed batches: rows beyond the length do not exist
  bool zero = t >= len || ll > p.U;  // padded frames, contract violations: exactly zero (base_loss.py:283-298)
  const int32_t *lab = p.labels + (long)b * p.label_stride;
  float *qf = reinterpret_cast<float *>(qtab);
  float inv = 0.f, qblank = 0.f;
  const float *x = p.logits + logits_off(p, b, t);
  const bool wrt_logits = p.wrt == 0;
  // the first 1024 columns of the logits row are requested before anything else: they do not depend on the chains, and the
  // polls and the lattice rows below are three dependent round trips through a saturated memory system
  float4 xv[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int k = lane * 4 + 256 * q;
    xv[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!zero && wrt_logits && k < V) xv[q] = *reinterpret_cast<const float4 *>(x + k);
  }
  if (t < len) wait_rows(len);
  if (!zero) {
    const float *ra = alpha + ((long)b * (p.T + 1) + (KIND == 0 ? t + 1 : t)) * L.SRS;
    const float *rb = beta + ((long)b * (p.T + 1) + t + 1) * L.SRS;
    // log2 of (alpha beta) per lattice state, relative to the two rows' offsets (they cancel in the normalisation);
    // qtab[i] = the token term of label position i, the blank terms stay in registers
    constexpr int MAXI = NL;  // label positions per lane
    float tb[MAXI];
    float t0 = NEG, m = NEG;
    if constexpr (KIND == 0) {
#pragma unroll
      for (int n = 0; n < MAXI; ++n) {
        const int i = lane + 64 * n;
        tb[n] = NEG;
        if (i < UP) {
          const float2 a = ld2<COH>(ra + 2 * i), bb = ld2<COH>(rb + 2 * i);
          tb[n] = a.x + bb.x;
          const float tt = (i < ll) ? a.y + bb.y : NEG;
          qf[i] = tt;
          m = fmaxf(m, fmaxf(tb[n], tt));
        }
      }
      if (lane == 0) t0 = ld1<COH>(ra + 2 * UP) + ld1<COH>(rb + 2 * UP);
    } else {
      const float *er = emis + row * (long)L.ERS;
      const float bl = ld1<COH>(er + UP);
#pragma unroll
      for (int n = 0; n < MAXI; ++n) {
        const int i = lane + 64 * n;
        tb[n] = NEG;
        if (i < UP) {
          const float ai = ld1<COH>(ra + i), bi = ld1<COH>(rb + i);
          tb[n] = ai + bi + bl;
          const float aprev = ld1<COH>(i == 0 ? ra + UP : ra + i - 1);
          const float tt = (i < ll) ? aprev + ld1<COH>(er + i) + bi : NEG;
          qf[i] = tt;
          m = fmaxf(m, fmaxf(tb[n], tt));
        }
      }
      if (lane == 0) t0 = ld1<COH>(ra + UP) + ld1<COH>(rb + UP) + bl;
    }
    m = wave_max(fmaxf(m, t0));
    if (!(m > NEG_THR)) {
      zero = true;  // no alignment passes through this frame: infeasible sample
    } else {
      float s = (lane == 0) ? fexp2(t0 - m) : 0.f;
      qblank = s;
#pragma unroll
      for (int n = 0; n < MAXI; ++n) {
        const int i = lane + 64 * n;
        if (i < UP) {
          const float qb = fexp2(tb[n] - m), qt = fexp2(qf[i] - m);  // (NEG - m underflows to 0)
          qblank += qb;
          s += qb + qt;
          qf[i] = qt;
        }
      }
      s = wave_sum(s);
      qblank = wave_sum(qblank);
      inv = 1.0f / s;  // s >= 1: the maximum contributes 2^0
    }
  }
  if (zero) {
    for (int k = lane * 4; k < V; k += 256) gput4(k, make_float4(0.f, 0.f, 0.f, 0.f));
    return;
  }
  const float fix = inv * 1073741824.0f;  // posteriors in units of 2^-30
  const unsigned qbfix = (unsigned)(fminf(qblank * inv, 1.0f) * 1073741824.0f + 0.5f);
  const float dl = d_loss ? d_loss[b] : 1.0f;
  const float mx = ld1<COH>(emis + row * (long)L.ERS + UP + 1);
  const float l2s = ld1<COH>(emis + row * (long)L.ERS + UP + 2);
  wave_lds_fence();
  for (int c0 = 0; c0 < V; c0 += CH) {
    float4 xn[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // the NEXT pass's logits: four loads in flight under this pass's LDS work and stores
      const int k = c0 + CH + lane * 4 + 256 * q;
      xn[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (wrt_logits && k < V) xn[q] = *reinterpret_cast<const float4 *>(x + k);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<uint4 *>(bins + lane * 4 + 256 * q) = make_uint4(0u, 0u, 0u, 0u);
    wave_lds_fence();
    for (int i = lane; i < ll; i += 64) {
      const int tok = (i < p.label_stride) ? lab[i] : p.blank;
      const unsigned r = (unsigned)(tok - c0);
      if (tok >= 0 && tok < V && tok != p.blank && r < (unsigned)CH) atomicAdd(&bins[r], (unsigned)(qf[i] * fix + 0.5f));
    }
    if (lane == 0 && p.blank >= c0 && p.blank < c0 + CH && p.blank < V) bins[p.blank - c0] = qbfix;
    wave_lds_fence();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = c0 + lane * 4 + 256 * q;
      if (k < V) {
        const uint4 u = *reinterpret_cast<const uint4 *>(bins + lane * 4 + 256 * q);
        const float c = 9.31322574615478515625e-10f;
        float4 r;
        if (wrt_logits) {
          // g_x[k] = d_loss * (softmax(x)[k] - post[k])  (TF autodiff of tools.py:37-39 applied to base_loss.py:150-153)
          r.x = dl * (fexp2((xv[q].x - mx) * LOG2E - l2s) - (float)u.x * c);
          r.y = dl * (fexp2((xv[q].y - mx) * LOG2E - l2s) - (float)u.y * c);
          r.z = dl * (fexp2((xv[q].z - mx) * LOG2E - l2s) - (float)u.z * c);
          r.w = dl * (fexp2((xv[q].w - mx) * LOG2E - l2s) - (float)u.w * c);
        } else {
          r = make_float4(-dl * ((float)u.x * c), -dl * ((float)u.y * c), -dl * ((float)u.z * c), -dl * ((float)u.w * c));  // base_loss.py:262-268
        }
        gput4(k, r);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) xv[q] = xn[q];
    wave_lds_fence();
  }
}

}  // namespace ctc

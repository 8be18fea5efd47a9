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
#include "ctc_fused_common.h"

namespace ctc {

namespace fused {

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
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: keeps everything derived from it in SGPRs
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

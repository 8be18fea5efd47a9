// gfx950 kernel of the dense CTC Hessian hess[B][T][V][T][V] (the O(l^4) path of base_loss.py:186-260).
//
// The reference builds the all-pairs transition tensor gamma[B,T+1,L(,2),T+1,L(,2)] (classic_ctc_loss.py:167-308,
// simplified_ctc_loss.py:85-191; 22.5 GB at B=32,T=200,U=32) and contracts it twice.  Here gamma is never formed.
// With g = -posterior the log-probability-space Hessian is
//     H[t1,k1,t2,k2] = -P(k1 at t1 and k2 at t2 | label) + g[t1,k1] g[t2,k2]            (t1 != t2)
//     H[t,k1,t,k2]   = delta_{k1 k2} g[t,k1] + g[t,k1] g[t,k2]
// One wavefront owns one (b, t1, k1) slab [T][V] and writes ALL of it:
//   * rows t2 > t1: alpha[t1] is pushed through frame t1 with every emission but k1 masked, the restricted vector is
//     propagated forward with the ordinary alpha step and closed with beta[t2+1] exactly like the gradient does;
//   * rows t2 < t1: symmetrically, beta[t1+1] is pulled back through frame t1 with only k1 allowed and propagated
//     backward with the ordinary beta step, closing with alpha (this IS the symmetric half of base_loss.py:223-233,
//     generated directly instead of by transposing 10 GB);
//   * the lattice/emission rows of the next step are prefetched while the current step is processed.
// For logits-space output (what tape.batch_jacobian returns, README.md:58-71) the t1 == t2 blocks get
// + diag(s) - s s^T (s = softmax), because every k-sum of H vanishes and sum_k g = -1.
// The closing / step code is the one of the fused loss+grad kernels (Side::post_step_sc in ctc_fused_common.h).
#include "ctc_fused_common.h"

namespace ctc {

using namespace ctc::fused;

template <int KIND, int NL>
__global__ __launch_bounds__(256) void hess_slab_kernel(Problem p, Layout L, const float *__restrict__ emis,
                                                         const float *__restrict__ alpha, const float *__restrict__ beta,
                                                         const double *__restrict__ logp, const float *__restrict__ g_lp,
                                                         float *__restrict__ hess, int wpb) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: keeps everything derived from it in SGPRs
  const int T = p.T, V = p.V, UP = L.UP;
  const long ntask = (long)p.B * T * V;
  const long task = (long)blockIdx.x * wpb + w;
  if (task >= ntask) return;
  const int k1 = (int)(task % V);
  const int t1 = (int)((task / V) % T);
  const int b = (int)(task / ((long)V * T));
  float *out = hess + task * ((long)T * V);
  const int len = clampi(p.logit_length[b], 0, T);
  int ll = p.label_length[b] < 0 ? 0 : p.label_length[b];
  const double lp = logp[b];
  const int32_t *lab = p.labels + (long)b * p.label_stride;
  auto tok = [&](int i) -> int { return (i >= 0 && i < ll) ? ((i < p.label_stride) ? lab[i] : p.blank) : -1 - (i < 0); };

  auto zero_rows = [&](int t_from, int t_to) {
    if (t_to <= t_from) return;
    float *q = out + (long)t_from * V;
    const long n = (long)(t_to - t_from) * V;
    for (long k = lane; k < n; k += 64) q[k] = 0.f;
  };
  const bool valid = (t1 < len) && (lp != -INFINITY) && (ll <= p.U);
  if (!valid) {  // padded frame or infeasible sample: the whole slab is zero (base_loss.py:240-258)
    zero_rows(0, T);
    return;
  }

  bool sel[NL];
  int tokb[NL];  // byte offset of label[i] in the LDS token row (pad slot for positions beyond the label)
  bool any = (k1 == p.blank);
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    const int i = lane * NL + j;
    const int tk = tok(i);
    sel[j] = (k1 != p.blank) && tk == k1;
    tokb[j] = 4 * ((tk >= 0 && tk < V && tk != p.blank) ? tk : V);
    any = any || sel[j];
  }
  any = __any(any);

  const float *grow = g_lp + (long)b * T * V;
  const float g1 = grow[(long)t1 * V + k1];
  const float *erows = emis + (long)b * T * L.ERS;
  const float *arows = alpha + (long)b * (T + 1) * L.SRS;
  const float *brows = beta + (long)b * (T + 1) * L.SRS;
  constexpr int PAIR = (KIND == 0) ? 2 : 1;
  const int tailpos = PAIR * UP;

  // ---- diagonal block t2 == t1 (base_loss.py:205-221: set_diag with the log-gradient) ----
  {
    float s1 = 0.f, mx = 0.f, l2s = 0.f;
    const float *x = p.logits + ((long)b * T + t1) * V;
    if (p.wrt == 0) {
      mx = erows[(long)t1 * L.ERS + UP + 1];
      l2s = erows[(long)t1 * L.ERS + UP + 2];
      s1 = fexp2((x[k1] - mx) * LOG2E - l2s);
    }
    for (int k2 = lane; k2 < V; k2 += 64) {
      float val = g1 * grow[(long)t1 * V + k2] + (k2 == k1 ? g1 : 0.f);
      if (p.wrt == 0) {
        float s2 = fexp2((x[k2] - mx) * LOG2E - l2s);
        val += (k2 == k1 ? s1 : 0.f) - s1 * s2;
      }
      out[(long)t1 * V + k2] = val;
    }
  }
  zero_rows(len, T);  // columns beyond logit_length (base_loss.py:254-258)
  if (!any) {  // token absent from the label: no joint mass with any other frame, g1 == 0
    zero_rows(0, t1);
    zero_rows(t1 + 1, len);
    return;
  }

  // joint posteriors are scattered by label with fixed-point integer LDS atomics (2^-30 resolution): ds_add_f32 is an
  // order of magnitude slower than ds_add_u32 on gfx950 (see ctc_fused_common.h)
  unsigned *ubin = reinterpret_cast<unsigned *>(lds + (long)w * (V + 4));
  auto tofix = [](float q) -> unsigned { return (unsigned)(fminf(q, 1.0f) * 1073741824.0f + 0.5f); };
  for (int k = lane; k < V + 4; k += 64) ubin[k] = 0u;

  // one prefetched step: emissions of frame t2 and the closing lattice row, already in the layout the sweep is aligned with
  struct Pre {
    float y[NL], bl;
    float a[NL], b2[NL], tx, oh, ol;
  };
  auto load_e = [&](Pre &q, int t) {
    const float *er = erows + (long)t * L.ERS;
#pragma unroll
    for (int j = 0; j < NL; ++j) q.y[j] = er[lane * NL + j];
    q.bl = er[UP];
  };
  // beta row (workspace layout: slot i = state of l = i+1, the l = 0 state at the tail) as the forward sweep needs it
  auto load_fwd_row = [&](Pre &q, int trow) {
    const float *r = brows + (long)trow * L.SRS;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const int i = lane * NL + j;
      if constexpr (KIND == 0) { float2 v = *reinterpret_cast<const float2 *>(r + 2 * i); q.a[j] = v.x; q.b2[j] = v.y; }
      else { q.a[j] = r[i]; q.b2[j] = NEG; }
    }
    q.tx = r[tailpos]; q.oh = r[tailpos + 2]; q.ol = r[tailpos + 3];
  };
  // alpha row shifted into the layout the backward sweep is aligned with: slot i = (state_c(l=i), open(l=i+1)), tail l=UP
  auto load_bwd_row = [&](Pre &q, int trow) {
    const float *r = arows + (long)trow * L.SRS;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const int i = lane * NL + j;
      q.a[j] = (i == 0) ? r[tailpos] : r[PAIR * (i - 1)];
      q.b2[j] = (KIND == 0) ? r[2 * i + 1] : NEG;
    }
    q.tx = r[PAIR * (UP - 1)]; q.oh = r[tailpos + 2]; q.ol = r[tailpos + 3];
  };

  // close one frame: scatter the joint posteriors and write the Hessian row of frame t2
  auto emit_row = [&](int t2, const float (&s1)[NL], const float (&s2)[NL], float s0) {
    float qb = (lane == 0) ? fexp2(s0) : 0.f;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      qb += fexp2(s1[j]);
      atomicAdd(reinterpret_cast<unsigned *>(reinterpret_cast<char *>(ubin) + tokb[j]), tofix(fexp2(s2[j])));
    }
    qb = wave_sum_dpp(qb);
    if (lane == 0) ubin[p.blank] = tofix(qb);
    __builtin_amdgcn_wave_barrier();
    for (int k2 = lane; k2 < V; k2 += 64) {  // base_loss.py:235-237 : -exp(.) + g (x) g
      out[(long)t2 * V + k2] = g1 * grow[(long)t2 * V + k2] - (float)ubin[k2] * 9.31322574615478515625e-10f;
      ubin[k2] = 0u;
    }
    __builtin_amdgcn_wave_barrier();
  };

  auto fill_common = [&](auto &S) {
    S.lane = lane; S.UP = UP; S.blank = p.blank; S.ll = ll;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const int i = lane * NL + j;
      S.norep[j] = (i == 0) || tok(i) != tok(i - 1);
      S.norep_next[j] = tok(i + 1) != tok(i);
    }
  };
  auto masked = [&](const Pre &q, Emis<NL> &e) {  // frame t1 with every emission but k1 forbidden
#pragma unroll
    for (int j = 0; j < NL; ++j) e.y[j] = sel[j] ? q.y[j] : NEG;
    e.bl = (k1 == p.blank) ? q.bl : NEG;
  };

  // =============== rows t2 > t1 : restricted alpha vector pushed forward ===============
  if (t1 + 1 < len) {
    Side<KIND, NL, 1, 0, true> S;
    fill_common(S);
    {
      const float *r = arows + (long)t1 * L.SRS;  // alpha[t1], workspace layout = native layout of an alpha chain
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        const int i = lane * NL + j;
        if constexpr (KIND == 0) { float2 v = *reinterpret_cast<const float2 *>(r + 2 * i); S.c[j] = v.x; S.o[j] = v.y; }
        else { S.c[j] = r[i]; S.o[j] = NEG; }
      }
      S.cx = r[tailpos];
      S.off = (double)r[tailpos + 2] + (double)r[tailpos + 3];
      Pre q1;
      load_e(q1, t1);
      Emis<NL> e1;
      masked(q1, e1);
      S.step(e1);  // = alpha[t1+1] restricted to the states entered by emitting k1 at t1
    }
    Pre cur, nxt;
    load_e(cur, t1 + 1);
    load_fwd_row(cur, t1 + 2);
    for (int t2 = t1 + 1; t2 < len; ++t2) {
      const int tn = (t2 + 1 < len) ? t2 + 1 : t2;
      load_e(nxt, tn);          // prefetch the next step while this one is processed
      load_fwd_row(nxt, tn + 1);
      Emis<NL> e;
      SRow<KIND, NL> r;
#pragma unroll
      for (int j = 0; j < NL; ++j) { e.y[j] = cur.y[j]; r.a[j] = cur.a[j]; r.b[j] = cur.b2[j]; }
      e.bl = cur.bl;
      r.tail = make_float4(cur.tx, 0.f, cur.oh, cur.ol);
      const float sc = (float)((double)cur.oh + (S.off - lp)) + cur.ol;
      float s1[NL], s2[NL], s0;
      S.post_step_sc(e, r, sc, s1, s2, s0);
      emit_row(t2, s1, s2, s0);
      if (((t2 - t1) & 15) == 0) S.renorm();
      cur = nxt;
    }
  }

  // =============== rows t2 < t1 : restricted beta vector pulled backward ===============
  if (t1 > 0) {
    Side<KIND, NL, 1, 1, true> S;
    fill_common(S);
    {
      const float *r = brows + (long)(t1 + 1) * L.SRS;  // beta[t1+1] -> native layout of a beta chain (slot i = l = i)
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        const int i = lane * NL + j;
        S.c[j] = (i == 0) ? r[tailpos] : r[PAIR * (i - 1)];
        S.o[j] = (KIND == 0) ? r[2 * i + 1] : NEG;
      }
      S.cx = r[PAIR * (UP - 1)];
      S.off = (double)r[tailpos + 2] + (double)r[tailpos + 3];
      Pre q1;
      load_e(q1, t1);
      Emis<NL> e1;
      masked(q1, e1);
      S.step(e1);  // = beta-like vector at t1 of the paths that emit k1 at t1
    }
    Pre cur, nxt;
    load_e(cur, t1 - 1);
    load_bwd_row(cur, KIND == 0 ? t1 : t1 - 1);  // classic closes with alpha[t2+1], simplified with a[t2]
    for (int t2 = t1 - 1; t2 >= 0; --t2) {
      const int tn = (t2 > 0) ? t2 - 1 : 0;
      load_e(nxt, tn);
      load_bwd_row(nxt, KIND == 0 ? tn + 1 : tn);
      Emis<NL> e;
      SRow<KIND, NL> r;
#pragma unroll
      for (int j = 0; j < NL; ++j) { e.y[j] = cur.y[j]; r.a[j] = cur.a[j]; r.b[j] = cur.b2[j]; }
      e.bl = cur.bl;
      r.tail = make_float4(cur.tx, 0.f, cur.oh, cur.ol);
      const float sc = (float)((double)cur.oh + (S.off - lp)) + cur.ol;
      float s1[NL], s2[NL], s0;
      S.post_step_sc(e, r, sc, s1, s2, s0);
      emit_row(t2, s1, s2, s0);
      if (((t1 - t2) & 15) == 0) S.renorm();
      cur = nxt;
    }
  }
}

size_t hessian_extra_bytes(int kind, int B, int T, int V, int U) {
  (void)kind; (void)U;
  return (size_t)B * T * V * sizeof(float);  // log-probability-space gradient g = -posterior
}

template <int KIND>
static hipError_t launch_slab(const Problem &p, const Layout &L, const float *emis, const float *alpha, const float *beta,
                              const double *logp, const float *g_lp, float *hess, hipStream_t st) {
  int wpb = 4;
  while (wpb > 1 && (size_t)wpb * (p.V + 4) * 4 > 64 * 1024) wpb >>= 1;
  const size_t shmem = (size_t)wpb * (p.V + 4) * 4;
  const long ntask = (long)p.B * p.T * p.V;
  const long nblk = (ntask + wpb - 1) / wpb;
  if (nblk > 0x7fffffffL) return hipErrorInvalidValue;
  dim3 grid((unsigned)nblk), block(64 * wpb);
  switch (L.NL) {
    case 1: hipLaunchKernelGGL((hess_slab_kernel<KIND, 1>), grid, block, shmem, st, p, L, emis, alpha, beta, logp, g_lp, hess, wpb); break;
    case 2: hipLaunchKernelGGL((hess_slab_kernel<KIND, 2>), grid, block, shmem, st, p, L, emis, alpha, beta, logp, g_lp, hess, wpb); break;
    case 4: hipLaunchKernelGGL((hess_slab_kernel<KIND, 4>), grid, block, shmem, st, p, L, emis, alpha, beta, logp, g_lp, hess, wpb); break;
    case 8: hipLaunchKernelGGL((hess_slab_kernel<KIND, 8>), grid, block, shmem, st, p, L, emis, alpha, beta, logp, g_lp, hess, wpb); break;
    case 16: hipLaunchKernelGGL((hess_slab_kernel<KIND, 16>), grid, block, shmem, st, p, L, emis, alpha, beta, logp, g_lp, hess, wpb); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t run_hessian(const Problem &p, const Layout &L, char *ws, const float *g_lp, float *hess, hipStream_t st) {
  if ((long)p.B * p.T == 0) return hipSuccess;
  const float *emis = reinterpret_cast<const float *>(ws + L.off_emis);
  const float *alpha = reinterpret_cast<const float *>(ws + L.off_alpha);
  const float *beta = reinterpret_cast<const float *>(ws + L.off_beta);
  const double *logp = reinterpret_cast<const double *>(ws + L.off_logp);
  return p.kind == 0 ? launch_slab<0>(p, L, emis, alpha, beta, logp, g_lp, hess, st)
                     : launch_slab<1>(p, L, emis, alpha, beta, logp, g_lp, hess, st);
}

}  // namespace ctc

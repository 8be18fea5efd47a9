// gfx950 kernels of the dense CTC Hessian hess[B][T][V][T][V] (the O(l^4) path of base_loss.py:186-260).
//
// The reference builds the all-pairs transition tensor gamma[B,T+1,L(,2),T+1,L(,2)] (classic_ctc_loss.py:167-308,
// simplified_ctc_loss.py:85-191; 22.5 GB at B=32,T=200,U=32) and contracts it twice.  Here gamma is never
// formed.  With g = -posterior the log-probability-space Hessian is
//     H[t1,k1,t2,k2] = -P(k1 at t1 and k2 at t2 | label) + g[t1,k1] g[t2,k2]            (t1 != t2)
//     H[t,k1,t,k2]   = delta_{k1 k2} g[t,k1] + g[t,k1] g[t,k2]
// and the joint posterior for t2 > t1 is obtained by restricting alpha[t1+1] to the states entered by
// emitting k1, pushing that vector forward with the ordinary alpha step, and closing with beta[t2+1] exactly
// like the gradient does.  One wavefront owns one (b, t1, k1) slab and streams its rows t2 >= t1;
// the block-lower triangle t2 < t1 is the transpose (base_loss.py:223-233) and is written by a tiled
// LDS transpose kernel.  For logits-space output (what tape.batch_jacobian returns, README.md:58-71) the
// t1 == t2 blocks get + diag(s) - s s^T (s = softmax), because every k-sum of H vanishes and sum_k g = -1.
#include "ctc_common.h"

namespace ctc {

__device__ __forceinline__ int clampi2(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

template <int KIND, int NL>
__global__ __launch_bounds__(256) void hess_upper_kernel(Problem p, Layout L, const float *__restrict__ emis,
                                                          const float *__restrict__ alpha, const float *__restrict__ beta,
                                                          const double *__restrict__ logp, const float *__restrict__ g_lp,
                                                          float *__restrict__ hess, int wpb) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  const int T = p.T, V = p.V, UP = L.UP;
  const long ntask = (long)p.B * T * V;
  const long task = (long)blockIdx.x * wpb + w;
  if (task >= ntask) return;
  const int k1 = (int)(task % V);
  const int t1 = (int)((task / V) % T);
  const int b = (int)(task / ((long)V * T));
  float *out = hess + task * ((long)T * V);
  const int len = clampi2(p.logit_length[b], 0, T);
  int ll = p.label_length[b] < 0 ? 0 : p.label_length[b];
  const double lp = logp[b];
  const int32_t *lab = p.labels + (long)b * p.label_stride;
  auto tok = [&](int i) -> int { return (i >= 0 && i < ll) ? ((i < p.label_stride) ? lab[i] : p.blank) : -1 - (i < 0); };

  bool sel[NL], norep_next[NL];
  bool any = (k1 == p.blank);
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    int i = lane * NL + j;
    sel[j] = (k1 != p.blank) && tok(i) == k1;
    norep_next[j] = tok(i + 1) != tok(i);
    any = any || sel[j];
  }
  any = __any(any);
  const bool valid = (t1 < len) && (lp != -INFINITY) && (ll <= p.U);

  auto zero_rows = [&](int t_from) {
    float *q = out + (long)t_from * V;
    const long n = (long)(T - t_from) * V;
    for (long k = lane; k < n; k += 64) q[k] = 0.f;
  };
  if (!valid) {
    zero_rows(t1);
    return;
  }

  const float *grow = g_lp + (long)b * T * V;
  const float g1 = grow[(long)t1 * V + k1];

  // ---- diagonal block t2 == t1 (base_loss.py:205-221: set_diag with the log-gradient) ----
  {
    float s1 = 0.f, mx = 0.f, l2s = 0.f;
    const float *x = p.logits + ((long)b * T + t1) * V;
    if (p.wrt == 0) {
      mx = emis[((long)b * T + t1) * L.ERS + UP + 1];
      l2s = emis[((long)b * T + t1) * L.ERS + UP + 2];
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

  if (!any) {  // token absent from the label: no joint mass with any later frame, g1 == 0
    if (t1 + 1 < T) zero_rows(t1 + 1);
    return;
  }
  float *bin = lds + (long)w * V;
  for (int k = lane; k < V; k += 64) bin[k] = 0.f;
  __builtin_amdgcn_wave_barrier();

  // ---- start vector: alpha[t1+1] restricted to the states entered by emitting k1 at t1 ----
  float c[NL], o[NL], cx;
  double voff;
  {
    const float *ra = alpha + ((long)b * (T + 1) + (KIND == 0 ? t1 + 1 : t1)) * L.SRS;
    const int offpos = (KIND == 0 ? 2 * UP : UP) + 2;
    voff = (double)ra[offpos] + (double)ra[offpos + 1];
    if constexpr (KIND == 0) {
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        int i = lane * NL + j;
        float2 a = *reinterpret_cast<const float2 *>(ra + 2 * i);
        c[j] = (k1 == p.blank) ? a.x : NEG;
        o[j] = sel[j] ? a.y : NEG;
      }
      cx = (k1 == p.blank) ? ra[2 * UP] : NEG;
    } else {
      const float *er = emis + ((long)b * T + t1) * L.ERS;
      const float bl = er[UP];
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        int i = lane * NL + j;
        o[j] = NEG;
        if (k1 == p.blank) {
          c[j] = bl + ra[i];
        } else {
          float aprev = (i == 0) ? ra[UP] : ra[i - 1];
          c[j] = sel[j] ? aprev + er[i] : NEG;
        }
      }
      cx = (k1 == p.blank) ? bl + ra[UP] : NEG;
    }
  }

  // ---- rows t2 > t1 ----
  for (int t2 = t1 + 1; t2 < len; ++t2) {
    const float *er = emis + ((long)b * T + t2) * L.ERS;
    const float *rb = beta + ((long)b * (T + 1) + t2 + 1) * L.SRS;
    const int offpos = (KIND == 0 ? 2 * UP : UP) + 2;
    const double sc = voff + (double)rb[offpos] + (double)rb[offpos + 1] - lp;
    auto post = [&](float a_, float b_) -> float { return fminf(fexp2((float)((double)a_ + (double)b_ + sc)), 1.0f); };  // a posterior never exceeds 1
    auto post3 = [&](float a_, float b_, float c_) -> float { return fminf(fexp2((float)((double)a_ + (double)b_ + (double)c_ + sc)), 1.0f); };
    const float bl = er[UP];
    float y[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) y[j] = er[lane * NL + j];
    float qblank = 0.f;
    if constexpr (KIND == 0) {
      // alpha step on the restricted vector (same recursion as scan_kernel)
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
        o[j] = y[j] + lse2(o[j], xin);
        c[j] = bl + m[j];
      }
      cx += bl;
      // close with beta[t2+1]: the state at t2+1 names the token emitted at t2
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        int i = lane * NL + j;
        float2 bb = *reinterpret_cast<const float2 *>(rb + 2 * i);
        qblank += post(c[j], bb.x);
        if (i < ll) {
          int tk = tok(i);
          if (tk >= 0 && tk < V && tk != p.blank) atomicAdd(&bin[tk], post(o[j], bb.y));
        }
      }
      if (lane == 0) qblank += post(cx, rb[2 * UP]);
    } else {
      // simplified: joint mass first (needs v[t2] and the emissions of t2), then the step
      float pin0 = from_prev_lane(c[NL - 1], cx);
#pragma unroll
      for (int j = NL - 1; j >= 0; --j) {
        int i = lane * NL + j;
        float pin = (j == 0) ? pin0 : c[j - 1];  // v[t2, l = i]
        float bi = rb[i];                          // beta[t2+1, l = i+1]
        qblank += post3(c[j], bi, bl);
        if (i < ll) {
          int tk = tok(i);
          if (tk >= 0 && tk < V && tk != p.blank) atomicAdd(&bin[tk], post3(pin, y[j], bi));
        }
        c[j] = lse2(bl + c[j], y[j] + pin);
      }
      if (lane == 0) qblank += post3(cx, rb[UP], bl);
      cx += bl;
    }
    qblank = wave_sum(qblank);
    if (lane == 0) bin[p.blank] = qblank;
    __builtin_amdgcn_wave_barrier();
    // base_loss.py:235-237 : -exp(.) + g (x) g
    for (int k2 = lane; k2 < V; k2 += 64) {
      out[(long)t2 * V + k2] = g1 * grow[(long)t2 * V + k2] - bin[k2];
      bin[k2] = 0.f;
    }
    __builtin_amdgcn_wave_barrier();
    if (((t2 - t1) & 15) == 0) {
      float mx = cx;
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        mx = fmaxf(mx, c[j]);
        if constexpr (KIND == 0) mx = fmaxf(mx, o[j]);
      }
      mx = wave_max(mx);
      if (mx > NEG_THR) {
#pragma unroll
        for (int j = 0; j < NL; ++j) {
          c[j] -= mx;
          if constexpr (KIND == 0) o[j] -= mx;
        }
        cx -= mx;
        voff += (double)mx;
      }
    }
  }
  if (len < T) zero_rows(len);  // columns beyond logit_length (base_loss.py:254-258)
}

// hess[b][t1][k1][t2][k2] = hess[b][t2][k2][t1][k1] for t1 > t2 (base_loss.py:223-233), 64x64 tiles through LDS.
__global__ __launch_bounds__(256) void hess_mirror_kernel(float *__restrict__ hess, int B, int T, int V) {
  __shared__ float tile[64][65];
  const long N = (long)T * V;
  const int ntile = (int)((N + 63) / 64);
  const int tr = blockIdx.y, tc = blockIdx.x;  // destination tile (row block, col block)
  if (tc > tr) return;
  const int b = blockIdx.z;
  float *M = hess + (long)b * N * N;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const long r0 = (long)tr * 64, c0 = (long)tc * 64;
  // a destination tile holds something to mirror only if its largest t1 exceeds its smallest t2
  if ((r0 + 63 < N ? r0 + 63 : N - 1) / V <= c0 / V) return;
  // source tile: rows c0.., cols r0..
  for (int yy = ty; yy < 64; yy += 4) {
    long sr = c0 + yy, scol = r0 + tx;
    tile[yy][tx] = (sr < N && scol < N) ? M[sr * N + scol] : 0.f;
  }
  __syncthreads();
  for (int yy = ty; yy < 64; yy += 4) {
    long dr = r0 + yy, dc = c0 + tx;
    if (dr < N && dc < N && (dr / V) > (dc / V)) M[dr * N + dc] = tile[tx][yy];
  }
  (void)ntile;
}

size_t hessian_extra_bytes(int kind, int B, int T, int V, int U) {
  (void)kind; (void)U;
  return (size_t)B * T * V * sizeof(float);  // log-probability-space gradient g = -posterior
}

template <int KIND>
static hipError_t launch_upper(const Problem &p, const Layout &L, const float *emis, const float *alpha, const float *beta,
                               const double *logp, const float *g_lp, float *hess, hipStream_t st) {
  int wpb = 4;
  while (wpb > 1 && (size_t)wpb * p.V * 4 > 64 * 1024) wpb >>= 1;
  const size_t shmem = (size_t)wpb * p.V * 4;
  const long ntask = (long)p.B * p.T * p.V;
  const long nblk = (ntask + wpb - 1) / wpb;
  if (nblk > 0x7fffffffL) return hipErrorInvalidValue;
  dim3 grid((unsigned)nblk), block(64 * wpb);
  switch (L.NL) {
    case 1: hipLaunchKernelGGL((hess_upper_kernel<KIND, 1>), grid, block, shmem, st, p, L, emis, alpha, beta, logp, g_lp, hess, wpb); break;
    case 2: hipLaunchKernelGGL((hess_upper_kernel<KIND, 2>), grid, block, shmem, st, p, L, emis, alpha, beta, logp, g_lp, hess, wpb); break;
    case 4: hipLaunchKernelGGL((hess_upper_kernel<KIND, 4>), grid, block, shmem, st, p, L, emis, alpha, beta, logp, g_lp, hess, wpb); break;
    case 8: hipLaunchKernelGGL((hess_upper_kernel<KIND, 8>), grid, block, shmem, st, p, L, emis, alpha, beta, logp, g_lp, hess, wpb); break;
    case 16: hipLaunchKernelGGL((hess_upper_kernel<KIND, 16>), grid, block, shmem, st, p, L, emis, alpha, beta, logp, g_lp, hess, wpb); break;
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
  hipError_t e = p.kind == 0 ? launch_upper<0>(p, L, emis, alpha, beta, logp, g_lp, hess, st)
                             : launch_upper<1>(p, L, emis, alpha, beta, logp, g_lp, hess, st);
  if (e != hipSuccess) return e;
  const long N = (long)p.T * p.V;
  const unsigned nt = (unsigned)((N + 63) / 64);
  if (nt > 65535u || p.B > 65535) return hipErrorInvalidValue;
  hipLaunchKernelGGL(hess_mirror_kernel, dim3(nt, nt, p.B), dim3(256), 0, st, hess, p.B, p.T, p.V);
  return hipGetLastError();
}

}  // namespace ctc

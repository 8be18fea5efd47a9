// gfx950 kernels of the CTC loss + gradient path (pipeline v1: emit -> scan -> grad).
//
//   emit_kernel : one wavefront per (b, t) frame.  Streams the logits row once, computes the softmax
//                 normaliser with wave-shuffle reductions and gathers the U+1 emissions the lattice needs
//                 (base_loss.py:59, 328-344, 365-393; classic_ctc_loss.py:464-563) into a compact row.
//                 HBM-bound, V-independent output.
//   scan_kernel : one wavefront per (utterance, direction).  The strictly sequential alpha / beta recursion
//                 (classic_ctc_loss.py:310-462, simplified_ctc_loss.py:291-438; tools.py:191-277 is the
//                 tf.while_loop it replaces) with the lattice row held in registers, NL label positions per
//                 lane, neighbour exchange by one DPP wave shift, emission rows prefetched PF steps ahead.
//   grad_kernel : one wavefront per (b, t) frame.  Posterior scatter into an LDS token row and the fused
//                 softmax - posterior write (classic_ctc_loss.py:565-669, simplified_ctc_loss.py:456-534,
//                 base_loss.py:262-298, 420-468 and TF's autodiff of tools.py:37-39).
#include "ctc_common.h"
#include "ctc_amd.h"
#include "ctc_swap_reduce.h"
#include "ctc_v1_device.h"
#include "ctc_grad_row.h"

namespace ctc {


__device__ __forceinline__ int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

// ------------------------------------------------------------------------------------------------
// emit
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void emit_kernel(Problem p, Layout L, float *__restrict__ emis) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (row >= (long)p.B * p.T) return;
  emit_row(p, L, emis, (int)(row / p.T), (int)(row % p.T), lane);
}
// Small vocabularies (V <= 512, float32, 16-byte aligned rows): FOUR consecutive frames of one utterance per wavefront.
// One wavefront per row spent most of its time waiting -- a row is one or two 16-byte loads per lane, then two dependent
// gathers (label -> token -> logit) per label position.  Here the rows' loads go out together, the four maxima and the four
// sums are reduced through one register each (ctc_swap_reduce.h), and the label tokens are fetched once for all four rows.
__global__ __launch_bounds__(256) void emit4_kernel(Problem p, Layout L, float *__restrict__ emis) {
  using namespace ctc::fused;
  const int lane = threadIdx.x & 63;
  const int nq = (p.T + 3) / 4;
  const long id = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (id >= (long)p.B * nq) return;
  const int b = (int)(id / nq), t0 = 4 * (int)(id % nq);
  const int len = clampi(p.logit_length[b], 0, p.T);
  if (t0 >= len) return;  // padded frames are never read downstream
  const int ll = p.label_length[b] < 0 ? 0 : p.label_length[b];
  const int V = p.V;
  const float *xb = p.logits + (long)b * p.xsb;
  const float *xr[4];
  bool on[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    on[r] = t0 + r < len;
    xr[r] = xb + (long)(on[r] ? t0 + r : t0) * p.xst;  // (rows past the length re-read row t0: nothing of theirs is written)
  }
  float mx[4], l2s[4];
  if (p.wrt == 0) {
    float4 v[4][2];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int k = lane * 4 + 256 * c;
        v[r][c] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        if (k < V) v[r][c] = *reinterpret_cast<const float4 *>(xr[r] + k);
      }
    float m[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
      m[r] = fmaxf(fmaxf(fmaxf(v[r][0].x, v[r][0].y), fmaxf(v[r][0].z, v[r][0].w)), fmaxf(fmaxf(v[r][1].x, v[r][1].y), fmaxf(v[r][1].z, v[r][1].w)));
    const float mall = swap_reduce<4, true>(m);
    float s[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float mm = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mall), SwapLanes<4>::lane(r)));
      mx[r] = (mm == -INFINITY) ? 0.f : mm;
      s[r] = 0.f;
#pragma unroll
      for (int c = 0; c < 2; ++c)
        s[r] += (fexp2((v[r][c].x - mx[r]) * LOG2E) + fexp2((v[r][c].y - mx[r]) * LOG2E)) +
                (fexp2((v[r][c].z - mx[r]) * LOG2E) + fexp2((v[r][c].w - mx[r]) * LOG2E));
    }
    const float sall = swap_reduce<4, false>(s);
#pragma unroll
    for (int r = 0; r < 4; ++r) l2s[r] = flog2(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(sall), SwapLanes<4>::lane(r))));  // -inf when the whole row is -inf
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) { mx[r] = 0.f; l2s[r] = 0.f; }
  }
  // log2 p(token k) = (x[k] - mx) * log2e - log2sum, gathered per label position (base_loss.py:328-344)
  float *erow = emis + ((long)b * p.T + t0) * (long)L.ERS;
  for (int i = lane; i < L.UP; i += 64) {
    int tok = -1;
    if (i < ll) tok = (i < p.label_stride) ? p.labels[(long)b * p.label_stride + i] : p.blank;
    const bool ok = tok >= 0 && tok < V && tok != p.blank;  // (a label equal to the blank: impossible emission, see emit_kernel)
    float g[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) g[r] = ok ? xr[r][tok] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float e = ok ? fmaxf((g[r] - mx[r]) * LOG2E - l2s[r], NEG) : NEG;
      if (!(e == e)) e = NEG;
      if (on[r]) erow[(long)r * L.ERS + i] = e;
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float bl = NEG;
      if (p.blank >= 0 && p.blank < V) bl = fmaxf((xr[r][p.blank] - mx[r]) * LOG2E - l2s[r], NEG);
      if (!(bl == bl)) bl = NEG;
      if (on[r]) {
        float *e4 = erow + (long)r * L.ERS + L.UP;
        e4[0] = bl; e4[1] = mx[r]; e4[2] = l2s[r]; e4[3] = 0.f;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// scan
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// grad
// ------------------------------------------------------------------------------------------------
// Posterior of "frame t emits token k".  Classic: the lattice state at t+1 names the token emitted at t
// (closed <=> blank, open(l) <=> label[l-1]), so post = sum over states of alpha[t+1] * beta[t+1] / P, which is
// the regrouping of classic_ctc_loss.py:565-669.  Simplified: blank = bl * sum_l a[t,l] b[t+1,l],
// token = sum_{i: label[i]=k} a[t,i] y[t,i] b[t+1,i+1] (simplified_ctc_loss.py:456-534).
template <int KIND>
__global__ __launch_bounds__(256) void grad_kernel(Problem p, Layout L, const float *__restrict__ emis,
                                                    const float *__restrict__ alpha, const float *__restrict__ beta,
                                                    const double *__restrict__ logp, const float *__restrict__ d_loss,
                                                    float *__restrict__ grad, int waves_per_block) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: keeps everything derived from it in SGPRs
  const long row = (long)blockIdx.x * waves_per_block + w;
  if (row >= (long)p.B * p.T) return;
  const int b = (int)(row / p.T), t = (int)(row % p.T);
  const int V = p.V, UP = L.UP;
  // output row in the consumer's format: float32 or bfloat16, any batch/time stride
  const long goff = grad_off(p, b, t);
  float *g = grad + goff;                                                        // valid for float32 only
  unsigned short *gh = reinterpret_cast<unsigned short *>(grad) + goff;           // valid for bfloat16 / float16 only
  const bool gbf = p.gdtype != 0;
  // outputs are written once and not read here: non-temporal stores
  auto gput = [&](int k, float v) {
    if (gbf) __builtin_nontemporal_store(p.gdtype == 2 ? f32_to_f16(v) : f32_to_bf16(v), gh + k);
    else __builtin_nontemporal_store(v, g + k);
  };
  auto gput4 = [&](int k, float4 r) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f v = {r.x, r.y, r.z, r.w};
    __builtin_nontemporal_store(v, reinterpret_cast<v4f *>(g + k));
  };
  const bool gvec = !gbf && ((V | goff) & 3) == 0 && (p.align_bits & 15) == 0;
  const int len = clampi(p.logit_length[b], 0, p.T);
  const double lp = logp[b];
  if (t >= len && p.row0 != nullptr) return;  // packed batches: rows beyond the length do not exist
  if (t >= len || lp == -INFINITY) {
    // padded frames and infeasible samples: exactly zero (base_loss.py:283-298)
    if (gvec) for (int k = lane * 4; k < V; k += 256) gput4(k, make_float4(0.f, 0.f, 0.f, 0.f));
    else for (int k = lane; k < V; k += 64) gput(k, 0.f);
    return;
  }
  int ll = p.label_length[b] < 0 ? 0 : p.label_length[b];
  // posterior scatter with fixed-point integer LDS atomics (2^-30 resolution; ds_add_f32 is far slower on gfx950)
  float *bin = lds + (long)w * V;
  unsigned *ubin = reinterpret_cast<unsigned *>(bin);
  auto tofix = [](float q) -> unsigned { return (unsigned)(fminf(q, 1.0f) * 1073741824.0f + 0.5f); };
  for (int k = lane; k < V; k += 64) ubin[k] = 0u;
  wave_lds_fence();

  const int32_t *lab = p.labels + (long)b * p.label_stride;
  const float *ra = alpha + ((long)b * (p.T + 1) + (KIND == 0 ? t + 1 : t)) * L.SRS;
  const float *rb = beta + ((long)b * (p.T + 1) + t + 1) * L.SRS;
  // Posteriors normalised by the frame's OWN mass sum_s alpha_t[s] beta_t[s] (= P for every t: the invariant of the reference's
  // tests/test_classic_ctc_loss.py:146-167; see ctc_grad_row.h): the row offsets cancel, so the sums stay in float32 even with
  // logits ~1e10 (README.md:74-78), and the rounding of a long sweep does not enter as a common factor.  Three passes over the
  // two lattice rows (L1 hits): maximum, sum, scatter.
  const float *er = emis + row * (long)L.ERS;
  const float bl = (KIND == 1) ? er[UP] : 0.f;
  // log2 (alpha beta) of the blank state and of the token state of label position i (NEG when there is none)
  auto terms = [&](int i, float &tb, float &tt) {
    if constexpr (KIND == 0) {
      const float2 a = *reinterpret_cast<const float2 *>(ra + 2 * i);
      const float2 bb = *reinterpret_cast<const float2 *>(rb + 2 * i);
      tb = a.x + bb.x;
      tt = (i < ll) ? a.y + bb.y : NEG;
    } else {
      const float ai = ra[i], bi = rb[i];  // state l = i+1 in both rows
      tb = ai + bi + bl;
      tt = (i < ll) ? ((i == 0) ? ra[UP] : ra[i - 1]) + er[i] + bi : NEG;  // a[t, l=i] * y[t,i] * b[t+1, l=i+1]
    }
  };
  const float t0 = (KIND == 0) ? ra[2 * UP] + rb[2 * UP] : ra[UP] + rb[UP] + bl;  // the l = 0 state
  float m = t0;
  for (int i = lane; i < UP; i += 64) {
    float tb, tt;
    terms(i, tb, tt);
    m = fmaxf(m, fmaxf(tb, tt));
  }
  m = wave_max(m);
  if (!(m > NEG_THR)) {  // no alignment passes through this frame (cannot happen on a feasible sample)
    if (gvec) for (int k = lane * 4; k < V; k += 256) gput4(k, make_float4(0.f, 0.f, 0.f, 0.f));
    else for (int k = lane; k < V; k += 64) gput(k, 0.f);
    return;
  }
  float ssum = (lane == 0) ? fexp2(t0 - m) : 0.f;
  for (int i = lane; i < UP; i += 64) {
    float tb, tt;
    terms(i, tb, tt);
    ssum += fexp2(tb - m) + fexp2(tt - m);  // (NEG - m underflows to 0)
  }
  const float inv = 1.0f / wave_sum(ssum);  // the sum is >= 1: the maximum contributes 2^0
  float qblank = (lane == 0) ? fexp2(t0 - m) * inv : 0.f;
  for (int i = lane; i < UP; i += 64) {
    float tb, tt;
    terms(i, tb, tt);
    qblank += fexp2(tb - m) * inv;
    if (i < ll) {
      const int tok = (i < p.label_stride) ? lab[i] : p.blank;
      if (tok >= 0 && tok < V && tok != p.blank) atomicAdd(&ubin[tok], tofix(fexp2(tt - m) * inv));
    }
  }
  qblank = wave_sum(qblank);
  if (lane == 0 && p.blank >= 0 && p.blank < V) ubin[p.blank] = tofix(qblank);
  wave_lds_fence();
  for (int k = lane; k < V; k += 64) bin[k] = (float)ubin[k] * 9.31322574615478515625e-10f;  // back to float, in place
  wave_lds_fence();

  const float dl = d_loss ? d_loss[b] : 1.0f;
  if (p.wrt == 0) {
    // g_x[k] = d_loss * (softmax(x)[k] * sum_k' post[k'] - post[k]), sum_k' post = 1 on a valid frame of a feasible
    // sample (TF autodiff of tools.py:37-39 applied to base_loss.py:150-153)
    const long xoff = logits_off(p, b, t);
    const float *x = p.logits + xoff;
    const unsigned short *xh = reinterpret_cast<const unsigned short *>(p.logits) + xoff;
    const bool bf = p.xdtype != 0;
    const float mx = emis[row * (long)L.ERS + UP + 1];
    const float l2s = emis[row * (long)L.ERS + UP + 2];
    if (gvec && !bf && (xoff & 3) == 0) {
      for (int k = lane * 4; k < V; k += 256) {
        float4 v = *reinterpret_cast<const float4 *>(x + k);
        float4 q = *reinterpret_cast<const float4 *>(bin + k);
        float4 r;
        r.x = dl * (fexp2((v.x - mx) * LOG2E - l2s) - q.x);
        r.y = dl * (fexp2((v.y - mx) * LOG2E - l2s) - q.y);
        r.z = dl * (fexp2((v.z - mx) * LOG2E - l2s) - q.z);
        r.w = dl * (fexp2((v.w - mx) * LOG2E - l2s) - q.w);
        gput4(k, r);
      }
    } else {
      for (int k = lane; k < V; k += 64) {
        const float xv = bf ? h16_to_f32(xh[k], p.xdtype) : x[k];
        gput(k, dl * (fexp2((xv - mx) * LOG2E - l2s) - bin[k]));
      }
    }
  } else {
    // gradient w.r.t. log-probabilities: -posterior (base_loss.py:262-268)
    for (int k = lane; k < V; k += 64) gput(k, -dl * bin[k]);
  }
}

// ------------------------------------------------------------------------------------------------
// log posterior (logarithmic_logproba_gradient, base_loss.py:270-298): lg[b,t,k] = log P(frame t emits k | label), computed
// in the LOG domain end to end -- a posterior of e^-150 comes out as -150, where log(-gradient) of the float32 gradient
// gives -inf below e^-87.  Per frame (one wavefront): q_i = alpha + beta + offsets - log P per lattice state (base-2 logs,
// summed in double like grad_kernel), then the segment log-sum-exp by token (tools.py:74-119) in two LDS passes: the
// per-token maximum by an integer atomic max on an order-preserving key, then sum 2^(q_i - max) in fixed point (units of
// 2^-20: up to 1024 states of one token; the maximum itself contributes exactly 1, so the sum is never 0).
// Frames beyond logit_length, infeasible samples and tokens that no state emits: -inf (base_loss.py:283-298).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int order_key(float x) { const int b = __float_as_int(x); return b ^ ((b >> 31) & 0x7fffffff); }
__device__ __forceinline__ float key_value(int k) { return __int_as_float(k ^ ((k >> 31) & 0x7fffffff)); }

template <int KIND>
__global__ __launch_bounds__(256) void logpost_kernel(Problem p, Layout L, const float *__restrict__ emis,
                                                       const float *__restrict__ alpha, const float *__restrict__ beta,
                                                       const double *__restrict__ logp, float *__restrict__ out, int waves_per_block) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long row = (long)blockIdx.x * waves_per_block + w;
  if (row >= (long)p.B * p.T) return;
  const int b = (int)(row / p.T), t = (int)(row % p.T);
  const int V = p.V, UP = L.UP;
  float *o = out + row * (long)V;
  const int len = clampi(p.logit_length[b], 0, p.T);
  const double lp = logp[b];
  int ll = p.label_length[b] < 0 ? 0 : p.label_length[b];
  if (t >= len || lp == -INFINITY || ll > p.U) {
    for (int k = lane; k < V; k += 64) o[k] = -INFINITY;
    return;
  }
  int *kmax = reinterpret_cast<int *>(lds) + (long)w * 2 * V;
  unsigned *ksum = reinterpret_cast<unsigned *>(kmax + V);
  const int KEY_NONE = order_key(-3.0e38f);
  for (int k = lane; k < V; k += 64) { kmax[k] = KEY_NONE; ksum[k] = 0u; }
  wave_lds_fence();
  const int32_t *lab = p.labels + (long)b * p.label_stride;
  const float *ra = alpha + ((long)b * (p.T + 1) + (KIND == 0 ? t + 1 : t)) * L.SRS;
  const float *rb = beta + ((long)b * (p.T + 1) + t + 1) * L.SRS;
  const int offpos = (KIND == 0 ? 2 * UP : UP) + 2;
  const double scale = (double)ra[offpos] + (double)ra[offpos + 1] + (double)rb[offpos] + (double)rb[offpos + 1] - lp;
  // log2 posterior of one state; "impossible" stays at the finite sentinel scale (< NEG_THR)
  auto lq = [&](float a_, float b_) -> float { return (a_ < NEG_THR || b_ < NEG_THR) ? NEG : (float)((double)a_ + (double)b_ + scale); };
  auto lq3 = [&](float a_, float b_, float c_) -> float {
    return (a_ < NEG_THR || b_ < NEG_THR || c_ < NEG_THR) ? NEG : (float)((double)a_ + (double)b_ + (double)c_ + scale);
  };
  constexpr int MAXI = CTC_AMD_MAX_U / 64;  // label positions per lane at most
  float qtok[MAXI], qbl[MAXI];
  float q0 = NEG;  // the boundary state (lane 0)
  const float *er = emis + row * (long)L.ERS;
  const float bl = (KIND == 1) ? er[UP] : 0.f;
#pragma unroll
  for (int n = 0; n < MAXI; ++n) {
    const int i = lane + 64 * n;
    qtok[n] = NEG; qbl[n] = NEG;
    if (i < UP) {
      if constexpr (KIND == 0) {
        const float2 a = *reinterpret_cast<const float2 *>(ra + 2 * i), bb = *reinterpret_cast<const float2 *>(rb + 2 * i);
        qbl[n] = lq(a.x, bb.x);
        if (i < ll) qtok[n] = lq(a.y, bb.y);
      } else {
        const float ai = ra[i], bi = rb[i];
        qbl[n] = lq3(ai, bi, bl);
        if (i < ll) qtok[n] = lq3((i == 0) ? ra[UP] : ra[i - 1], er[i], bi);
      }
    }
  }
  if (lane == 0) q0 = (KIND == 0) ? lq(ra[2 * UP], rb[2 * UP]) : lq3(ra[UP], rb[UP], bl);
  // pass 1: per-token maximum
#pragma unroll
  for (int n = 0; n < MAXI; ++n) {
    const int i = lane + 64 * n;
    if (i < ll && qtok[n] > NEG_THR) {
      const int tok = (i < p.label_stride) ? lab[i] : p.blank;
      if (tok >= 0 && tok < V && tok != p.blank) atomicMax(&kmax[tok], order_key(qtok[n]));
    }
  }
  wave_lds_fence();
  // pass 2: sum of 2^(q - max) per token, fixed point
#pragma unroll
  for (int n = 0; n < MAXI; ++n) {
    const int i = lane + 64 * n;
    if (i < ll && qtok[n] > NEG_THR) {
      const int tok = (i < p.label_stride) ? lab[i] : p.blank;
      if (tok >= 0 && tok < V && tok != p.blank) {
        const float mx = key_value(kmax[tok]);
        atomicAdd(&ksum[tok], (unsigned)(fexp2(qtok[n] - mx) * 1048576.0f + 0.5f));
      }
    }
  }
  // blank: log-sum-exp over every closed state, in registers
  float m = q0;
#pragma unroll
  for (int n = 0; n < MAXI; ++n) m = fmaxf(m, qbl[n]);
  m = wave_max(m);
  float sb = (q0 > NEG_THR) ? fexp2(q0 - m) : 0.f;
#pragma unroll
  for (int n = 0; n < MAXI; ++n) sb += (qbl[n] > NEG_THR) ? fexp2(qbl[n] - m) : 0.f;
  sb = wave_sum(sb);
  const float lblank = (m > NEG_THR && sb > 0.f) ? (m + flog2(sb)) * (float)LN2_D : -INFINITY;
  wave_lds_fence();
  for (int k = lane; k < V; k += 64) {
    const unsigned su = ksum[k];
    float v = (su > 0u) ? (key_value(kmax[k]) + flog2((float)su * 9.5367431640625e-7f)) * (float)LN2_D : -INFINITY;
    if (k == p.blank) v = lblank;
    o[k] = v;
  }
}

hipError_t run_log_posterior(const Problem &p, const Layout &L, char *ws, float *out, hipStream_t st) {
  const long rows = (long)p.B * p.T;
  if (rows == 0) return hipSuccess;
  const float *emis = reinterpret_cast<const float *>(ws + L.off_emis);
  const float *alpha = reinterpret_cast<const float *>(ws + L.off_alpha);
  const float *beta = reinterpret_cast<const float *>(ws + L.off_beta);
  const double *logp = reinterpret_cast<const double *>(ws + L.off_logp);
  int wpb = 4;
  while (wpb > 1 && (size_t)wpb * p.V * 8 > 64 * 1024) wpb >>= 1;
  const dim3 grid((unsigned)((rows + wpb - 1) / wpb)), block(64 * wpb);
  const size_t shmem = (size_t)wpb * p.V * 8;
  if (p.kind == 0) hipLaunchKernelGGL(logpost_kernel<0>, grid, block, shmem, st, p, L, emis, alpha, beta, logp, out, wpb);
  else hipLaunchKernelGGL(logpost_kernel<1>, grid, block, shmem, st, p, L, emis, alpha, beta, logp, out, wpb);
  return hipGetLastError();
}

// Wide vocabularies (V > 1024, float32 rows, 16-byte aligned): one wavefront per frame, the vocabulary walked in passes of 1024
// columns, posteriors normalised by the frame's own mass (ctc_grad_row.h).  LDS per wavefront is 4 KB of bins + 256 NL bytes
// whatever V is (grad_kernel needs 4 V bytes: at V = 8192 that left 4 wavefronts per CU and 2.3 TB/s).
template <int KIND, int NL>
__global__ __launch_bounds__(256) void grad_wide_kernel(Problem p, Layout L, const float *__restrict__ emis,
                                                         const float *__restrict__ alpha, const float *__restrict__ beta,
                                                         const float *__restrict__ d_loss, float *__restrict__ grad) {
  __shared__ __attribute__((aligned(16))) unsigned bins_s[4][1024];
  __shared__ unsigned qtab_s[4][64 * NL];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long row = (long)blockIdx.x * 4 + w;
  if (row >= (long)p.B * p.T) return;
  grad_row<KIND, NL, false>(p, L, emis, alpha, beta, d_loss, grad, (int)(row / p.T), (int)(row % p.T), lane, bins_s[w], qtab_s[w],
                            [](int) {});
}

// ------------------------------------------------------------------------------------------------
// convert: workspace rows -> the reference's alpha/beta tensors (natural log, -inf, padded frames filled in)
// ------------------------------------------------------------------------------------------------
template <int KIND>
__global__ void convert_kernel(Problem p, Layout L, const float *__restrict__ ws_rows, int is_beta, float *__restrict__ out) {
  const int S = (KIND == 0) ? 2 : 1;
  const int Lr = p.U + 1;
  const long n = (long)p.B * (p.T + 1) * Lr * S;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
    int s = (int)(idx % S);
    long r = idx / S;
    int l = (int)(r % Lr); r /= Lr;
    int t = (int)(r % (p.T + 1));
    int b = (int)(r / (p.T + 1));
    const int len = clampi(p.logit_length[b], 0, p.T);
    int ll = p.label_length[b] < 0 ? 0 : p.label_length[b];
    float val;
    if (ll > p.U) {
      val = -INFINITY;
    } else if (is_beta && t >= len) {
      val = (l == ll) ? 0.f : -INFINITY;  // beta on padded frames: blank self-loops keep the one-hot
    } else {
      const int tt = t <= len ? t : len;
      const float *row = ws_rows + ((long)b * (p.T + 1) + tt) * L.SRS;
      const int offpos = (KIND == 0 ? 2 * L.UP : L.UP) + 2;
      const double off = (double)row[offpos] + (double)row[offpos + 1];
      float v;
      if constexpr (KIND == 0) {
        const int pos = (l == 0) ? 2 * L.UP : 2 * (l - 1);
        if (t <= len) {
          v = row[pos + s];
        } else {
          // alpha on padded frames: every state closes (blank with probability 1), open states die
          v = (s == 0) ? lse2(row[pos], row[pos + 1]) : NEG;
        }
      } else {
        const int pos = (l == 0) ? L.UP : (l - 1);
        v = row[pos];
      }
      val = (v > NEG_THR) ? (float)(((double)v + off) * LN2_D) : -INFINITY;
    }
    out[idx] = val;
  }
}

}  // namespace ctc

// ------------------------------------------------------------------------------------------------
// host-side launchers (called by the C ABI in ctc_capi.hip)
// ------------------------------------------------------------------------------------------------
namespace ctc {

// grid (B, ndir): blockIdx.y = 0 runs the alpha sweep, 1 the beta sweep.  The two sweeps are independent
// (beta never reads alpha), so one launch puts both wavefronts of an utterance on the chip at once.
template <int KIND, int NL, bool FINE>
__global__ __launch_bounds__(64) void scan_kernel(Problem p, Layout L, const float *__restrict__ emis,
                                                   float *__restrict__ alpha, float *__restrict__ beta,
                                                   double *__restrict__ logp, float *__restrict__ loss) {
  if (blockIdx.y == 0) scan_body<KIND, NL, 0, FINE>(p, L, emis, alpha, logp, loss, blockIdx.x, threadIdx.x);
  else scan_body<KIND, NL, 1, FINE>(p, L, emis, beta, logp, loss, blockIdx.x, threadIdx.x);
}

// ndir: 1 = alpha only, 2 = both sweeps; + 4 = rows renormalised every step (scan_body FINE: the Hessian-vector product's pipeline)
template <int KIND, int NL>
static void launch_scan_nl(const Problem &p, const Layout &L, const float *emis, float *alpha, float *beta, double *logp,
                           float *loss, int ndir, hipStream_t st) {
  if (ndir & 4) hipLaunchKernelGGL((scan_kernel<KIND, NL, true>), dim3(p.B, ndir & 3), dim3(64), 0, st, p, L, emis, alpha, beta, logp, loss);
  else hipLaunchKernelGGL((scan_kernel<KIND, NL, false>), dim3(p.B, ndir & 3), dim3(64), 0, st, p, L, emis, alpha, beta, logp, loss);
}

template <int KIND>
static hipError_t launch_scan(const Problem &p, const Layout &L, const float *emis, float *alpha, float *beta,
                              double *logp, float *loss, int ndir, hipStream_t st) {
  switch (L.NL) {
    case 1: launch_scan_nl<KIND, 1>(p, L, emis, alpha, beta, logp, loss, ndir, st); break;
    case 2: launch_scan_nl<KIND, 2>(p, L, emis, alpha, beta, logp, loss, ndir, st); break;
    case 4: launch_scan_nl<KIND, 4>(p, L, emis, alpha, beta, logp, loss, ndir, st); break;
    case 8: launch_scan_nl<KIND, 8>(p, L, emis, alpha, beta, logp, loss, ndir, st); break;
    case 16: launch_scan_nl<KIND, 16>(p, L, emis, alpha, beta, logp, loss, ndir, st); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t run_emit_scan(const Problem &p, const Layout &L, char *ws, float *loss, int ndir, hipStream_t st) {
  float *emis = reinterpret_cast<float *>(ws + L.off_emis);
  float *alpha = reinterpret_cast<float *>(ws + L.off_alpha);
  float *beta = reinterpret_cast<float *>(ws + L.off_beta);
  double *logp = reinterpret_cast<double *>(ws + L.off_logp);
  const long rows = (long)p.B * p.T;
  if (rows > 0) {
    const bool four = p.V <= 512 && p.xdtype == 0 && p.row0 == nullptr && (p.align_bits & 15) == 0 && ((p.V | p.xsb | p.xst) & 3) == 0;
    const long waves = (long)p.B * ((p.T + 3) / 4);
    if (four) hipLaunchKernelGGL(emit4_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, p, L, emis);
    else hipLaunchKernelGGL(emit_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, p, L, emis);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  if (p.B == 0) return hipSuccess;
  return p.kind == 0 ? launch_scan<0>(p, L, emis, alpha, beta, logp, loss, ndir, st)
                     : launch_scan<1>(p, L, emis, alpha, beta, logp, loss, ndir, st);
}

hipError_t run_grad(const Problem &p, const Layout &L, char *ws, const float *d_loss, float *grad, hipStream_t st) {
  const long rows = (long)p.B * p.T;
  if (rows == 0) return hipSuccess;
  const float *emis = reinterpret_cast<const float *>(ws + L.off_emis);
  const float *alpha = reinterpret_cast<const float *>(ws + L.off_alpha);
  const float *beta = reinterpret_cast<const float *>(ws + L.off_beta);
  const double *logp = reinterpret_cast<const double *>(ws + L.off_logp);
  const bool wide = p.V > 1024 && p.xdtype == 0 && p.gdtype == 0 && (p.align_bits & 15) == 0 &&
                    ((p.V | p.xsb | p.xst | p.gsb | p.gst) & 3) == 0 && L.UP <= CTC_AMD_MAX_U;
  if (wide) {
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
#define CTC_GW(K, N) hipLaunchKernelGGL((grad_wide_kernel<K, N>), grid, block, 0, st, p, L, emis, alpha, beta, d_loss, grad)
#define CTC_GW_NL(K) switch (L.NL) { case 1: CTC_GW(K, 1); break; case 2: CTC_GW(K, 2); break; case 4: CTC_GW(K, 4); break; \
                                     case 8: CTC_GW(K, 8); break; case 16: CTC_GW(K, 16); break; default: return hipErrorInvalidValue; }
    if (p.kind == 0) { CTC_GW_NL(0) } else { CTC_GW_NL(1) }
#undef CTC_GW_NL
#undef CTC_GW
    return hipGetLastError();
  }
  int wpb = 4;
  while (wpb > 1 && (size_t)wpb * p.V * 4 > 64 * 1024) wpb >>= 1;
  const size_t shmem = (size_t)wpb * p.V * 4;
  dim3 grid((unsigned)((rows + wpb - 1) / wpb)), block(64 * wpb);
  if (p.kind == 0)
    hipLaunchKernelGGL(grad_kernel<0>, grid, block, shmem, st, p, L, emis, alpha, beta, logp, d_loss, grad, wpb);
  else
    hipLaunchKernelGGL(grad_kernel<1>, grid, block, shmem, st, p, L, emis, alpha, beta, logp, d_loss, grad, wpb);
  return hipGetLastError();
}

hipError_t run_convert(const Problem &p, const Layout &L, char *ws, float *alpha_out, float *beta_out, hipStream_t st) {
  const long n = (long)p.B * (p.T + 1) * (p.U + 1) * (p.kind == 0 ? 2 : 1);
  if (n == 0) return hipSuccess;
  const float *alpha = reinterpret_cast<const float *>(ws + L.off_alpha);
  const float *beta = reinterpret_cast<const float *>(ws + L.off_beta);
  unsigned blocks = (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  if (p.kind == 0) {
    hipLaunchKernelGGL(convert_kernel<0>, dim3(blocks), dim3(256), 0, st, p, L, alpha, 0, alpha_out);
    hipLaunchKernelGGL(convert_kernel<0>, dim3(blocks), dim3(256), 0, st, p, L, beta, 1, beta_out);
  } else {
    hipLaunchKernelGGL(convert_kernel<1>, dim3(blocks), dim3(256), 0, st, p, L, alpha, 0, alpha_out);
    hipLaunchKernelGGL(convert_kernel<1>, dim3(blocks), dim3(256), 0, st, p, L, beta, 1, beta_out);
  }
  return hipGetLastError();
}

// perm[rank] = utterance with the rank-th longest logit_length (ties by index): B <= 8192, one thread per utterance; the
// lengths are staged in LDS once per workgroup (a per-thread loop over global memory took ~30 us at B = 512)
static __global__ __launch_bounds__(256) void order_kernel(const int *__restrict__ logit_length, int B, int T,
                                                           int *__restrict__ perm) {
  __shared__ int len_s[8192];
  for (int j = threadIdx.x; j < B; j += 256) len_s[j] = (logit_length[j] < 0 ? 0 : (logit_length[j] > T ? T : logit_length[j]));
  __syncthreads();
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  const int mine = len_s[b];
  int rank = 0;
  for (int j = 0; j < B; ++j) {
    const int other = len_s[j];
    rank += (other > mine) || (other == mine && j < b);
  }
  perm[rank] = b;
}

hipError_t run_order(const Problem &p, const Layout &L, char *ws, hipStream_t st) {
  int *perm = reinterpret_cast<int *>(ws + L.off_perm);
  hipLaunchKernelGGL(order_kernel, dim3((p.B + 255) / 256), dim3(256), 0, st, p.logit_length, p.B, p.T, perm);
  return hipGetLastError();
}

// labels[b][i] for i < min(label_length[b], U, label_stride) must lie in [0, V) and differ from the blank: counts the offenders
static __global__ __launch_bounds__(256) void check_labels_kernel(const int32_t *__restrict__ labels, int label_stride,
                                                                  const int32_t *__restrict__ label_length, int blank, int B, int V,
                                                                  int U, int *__restrict__ bad) {
  const long n = (long)B * label_stride;
  int mine = 0;
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
    const int b = (int)(idx / label_stride), i = (int)(idx % label_stride);
    int ll = label_length[b];
    ll = ll > U ? 0 : ll;  // (such a sample is infeasible by definition; its labels are never read)
    if (i < ll) {
      const int tk = labels[idx];
      mine += (tk < 0 || tk >= V || tk == blank);
    }
  }
  if (mine) atomicAdd(bad, mine);
}
hipError_t run_check_labels(const int32_t *labels, int label_stride, const int32_t *label_length, int blank, int B, int V, int U,
                            int *bad, hipStream_t st) {
  hipError_t e = hipMemsetAsync(bad, 0, sizeof(int), st);
  if (e != hipSuccess) return e;
  const long n = (long)B * label_stride;
  const unsigned blocks = (unsigned)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256);
  hipLaunchKernelGGL(check_labels_kernel, dim3(blocks), dim3(256), 0, st, labels, label_stride, label_length, blank, B, V, U, bad);
  return hipGetLastError();
}

// out[0] = sum of the finite losses, out[1] = their number: what a training loop all-reduces (dist.py), in ONE launch
static __global__ __launch_bounds__(256) void reduce_loss_kernel(const float *__restrict__ loss, int B, float *__restrict__ out) {
  __shared__ float ss[4], sn[4];
  float s = 0.f, n = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) {
    const float v = loss[b];
    const bool fin = (v - v) == 0.f;  // false for +-inf and NaN
    s += fin ? v : 0.f;
    n += fin ? 1.f : 0.f;
  }
  s = wave_sum(s); n = wave_sum(n);
  if ((threadIdx.x & 63) == 0) { ss[threadIdx.x >> 6] = s; sn[threadIdx.x >> 6] = n; }
  __syncthreads();
  if (threadIdx.x == 0) { out[0] = (ss[0] + ss[1]) + (ss[2] + ss[3]); out[1] = (sn[0] + sn[1]) + (sn[2] + sn[3]); }
}
// the same pair as add_loss_fixed accumulates inside fused6_kernel, for the pipelines that do not (one small launch)
static __global__ __launch_bounds__(256) void sum_loss_fixed_kernel(const float *__restrict__ loss, int B, long long *__restrict__ acc,
                                                                     long long *__restrict__ zero_next) {
  for (int b = blockIdx.x * 256 + threadIdx.x; b < B; b += gridDim.x * 256) add_loss_fixed(acc, loss[b]);
  if (zero_next != nullptr && blockIdx.x == 0 && threadIdx.x == 0) { zero_next[0] = 0; zero_next[1] = 0; }
}
hipError_t run_sum_loss_fixed(const float *loss, int B, long long *acc, long long *zero_next, hipStream_t st) {
  hipLaunchKernelGGL(sum_loss_fixed_kernel, dim3(B <= 256 ? 1 : (B + 255) / 256 > 64 ? 64 : (B + 255) / 256), dim3(256), 0, st, loss, B, acc, zero_next);
  return hipGetLastError();
}
hipError_t run_reduce_loss(const float *loss, int B, float *out, hipStream_t st) {
  hipLaunchKernelGGL(reduce_loss_kernel, dim3(1), dim3(256), 0, st, loss, B, out);
  return hipGetLastError();
}

// box probe of bench.py (ctc_amd_probe_copy): a plain streaming copy -- every workgroup owns one contiguous chunk, 16 bytes per
// lane, eight loads in flight, non-temporal stores (the fastest of the copy shapes tried by scripts/r03_memprobe.hip: 5.7 TB/s
// where a grid-stride copy with four loads in flight reached 4.6)
__global__ __launch_bounds__(512) void probe_copy_kernel(float4 *__restrict__ dst, const float4 *__restrict__ src, long n) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  const long per = (n + gridDim.x - 1) / gridDim.x;
  const long lo = per * blockIdx.x, hi = (lo + per < n) ? lo + per : n;
  auto put = [&](long k, float4 v) { v4f t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w; __builtin_nontemporal_store(t, reinterpret_cast<v4f *>(dst + k)); };
  long i = lo + threadIdx.x;
  for (; i + 7 * 512 < hi; i += 8 * 512) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[i + u * 512];
#pragma unroll
    for (int u = 0; u < 8; ++u) put(i + u * 512, v[u]);
  }
  for (; i < hi; i += 512) put(i, src[i]);
}
// stand-in for a latency-bound collective kernel (ctc_amd_probe_spin): one workgroup, some LDS, polls the 100 MHz device clock.
// Every wavefront reaches the exit: the loop ends on elapsed time, with a hard bound on the trip count.
__global__ void probe_spin_kernel(long long ticks, int lds_words) {
  extern __shared__ int spin_lds[];
  if (lds_words > 0) spin_lds[threadIdx.x % lds_words] = threadIdx.x;
  const long long t0 = wall_clock64();
  for (int it = 0; it < (1 << 22); ++it) {
    if (wall_clock64() - t0 >= ticks) break;
    __builtin_amdgcn_s_sleep(8);
  }
  if (lds_words > 0 && spin_lds[0] == -12345) spin_lds[1] = 0;
}
hipError_t run_probe_spin(int threads, int lds_bytes, float us, hipStream_t st) {
  hipLaunchKernelGGL(probe_spin_kernel, dim3(1), dim3(threads), (size_t)lds_bytes, st, (long long)(us * 100.0f), lds_bytes / 4);
  return hipGetLastError();
}
hipError_t run_probe_copy(void *dst, const void *src, size_t bytes, hipStream_t st) {
  if (bytes == 0) return hipSuccess;
  hipLaunchKernelGGL(probe_copy_kernel, dim3(8192), dim3(512), 0, st, static_cast<float4 *>(dst), static_cast<const float4 *>(src), (long)(bytes / 16));
  return hipGetLastError();
}

}  // namespace ctc

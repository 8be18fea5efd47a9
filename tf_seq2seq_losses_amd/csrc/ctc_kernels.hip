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

namespace ctc {


__device__ __forceinline__ int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

// ------------------------------------------------------------------------------------------------
// emit
// ------------------------------------------------------------------------------------------------
// one frame (b, t) by one wavefront
__device__ __forceinline__ void emit_row(const Problem &p, const Layout &L, float *__restrict__ emis, int b, int t, int lane) {
  const long row = (long)b * p.T + t;
  const int len = clampi(p.logit_length[b], 0, p.T);
  if (t >= len) return;  // padded frames are never read downstream
  const int ll = p.label_length[b] < 0 ? 0 : p.label_length[b];
  const int V = p.V;
  // element accessor of this frame's row: float32 or bfloat16, any batch/time stride (producer formats)
  const long xoff = (long)b * p.xsb + (long)t * p.xst;
  const float *x = p.logits + xoff;                                                   // valid for float32 only
  const unsigned short *xh = reinterpret_cast<const unsigned short *>(p.logits) + xoff;  // valid for bfloat16 only
  const bool bf = p.xdtype != 0;
  auto xat = [&](int k) -> float { return bf ? bf16_to_f32(xh[k]) : x[k]; };

  float mx = -INFINITY, sum = 0.f, log2sum = 0.f;
  if (p.wrt == 0) {
    if (!bf && ((V | xoff) & 3) == 0 && (p.align_bits & 15) == 0) {
      // ONE pass over the row, eight 16-byte loads per lane in flight: every lane keeps a running (maximum, sum of
      // exp(x - maximum)) of its own columns, the lanes are combined once at the end (two passes with one load in flight
      // each read a V = 2048 row at 2.7 TB/s chip-wide)
      float m = -INFINITY, s = 0.f;
      for (int k0 = lane * 4; k0 < V; k0 += 256 * 8) {
        float4 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int k = k0 + 256 * q;
          v[q] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
          if (k < V) v[q] = *reinterpret_cast<const float4 *>(x + k);
        }
        float cm = m;
#pragma unroll
        for (int q = 0; q < 8; ++q) cm = fmaxf(fmaxf(cm, fmaxf(v[q].x, v[q].y)), fmaxf(v[q].z, v[q].w));
        const float mr = (cm == -INFINITY) ? 0.f : cm;
        s *= fexp2((m - mr) * LOG2E);  // (m = -inf: s is 0 and stays 0)
#pragma unroll
        for (int q = 0; q < 8; ++q)
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
  for (int i = lane; i < L.UP; i += 64) {
    float e = NEG;
    if (i < ll) {
      int tok = (i < p.label_stride) ? p.labels[(long)b * p.label_stride + i] : p.blank;
      // (a label equal to the blank id is unsupported input in the reference; every tier treats it as an impossible
      // emission: the sample comes out infeasible, loss +inf, gradient 0)
      if (tok >= 0 && tok < V && tok != p.blank) e = fmaxf((xat(tok) - mx) * LOG2E - log2sum, NEG);
      if (!(e == e)) e = NEG;
    }
    erow[i] = e;
  }
  if (lane == 0) {
    float bl = NEG;
    if (p.blank >= 0 && p.blank < V) bl = fmaxf((xat(p.blank) - mx) * LOG2E - log2sum, NEG);
    if (!(bl == bl)) bl = NEG;
    erow[L.UP] = bl;
    // softmax(x)[k] = exp2((x[k] - mx) * log2e - log2sum); kept as two terms so that huge logits cancel exactly
    erow[L.UP + 1] = mx;
    erow[L.UP + 2] = log2sum;
    erow[L.UP + 3] = 0.f;
  }
}

__global__ __launch_bounds__(256) void emit_kernel(Problem p, Layout L, float *__restrict__ emis) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (row >= (long)p.B * p.T) return;
  emit_row(p, L, emis, (int)(row / p.T), (int)(row % p.T), lane);
}
// The frames of SELECTED utterances only (only_if[b] != 0: the utterances a fused kernel flagged for the log-domain pipeline,
// normally none).  A small fixed grid walks the batch -- blockIdx.y strides over utterances, blockIdx.x over the frames of a
// selected one -- so a call that selects nothing costs 2 048 workgroups that read 32 flags each, not B T / 4 empty ones.
__global__ __launch_bounds__(256) void emit_sel_kernel(Problem p, Layout L, float *__restrict__ emis, const int *__restrict__ only_if) {
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int b = blockIdx.y; b < p.B; b += gridDim.y) {
    if (only_if[b] == 0) continue;
    for (int t = blockIdx.x * 4 + w; t < p.T; t += gridDim.x * 4) emit_row(p, L, emis, b, t, lane);
  }
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
template <int NL>
struct ERow {
  float y[NL];
  float bl;
};

template <int NL>
__device__ __forceinline__ void load_erow(ERow<NL> &r, const float *__restrict__ base, int lane, int UP, int vz) {
  const float *p = base + lane * NL;
  if constexpr (NL == 1) {
    r.y[0] = p[0];
  } else if constexpr (NL == 2) {
    float2 v = *reinterpret_cast<const float2 *>(p);
    r.y[0] = v.x; r.y[1] = v.y;
  } else {
#pragma unroll
    for (int q = 0; q < NL / 4; ++q) {
      float4 v = *reinterpret_cast<const float4 *>(p + 4 * q);
      r.y[4 * q] = v.x; r.y[4 * q + 1] = v.y; r.y[4 * q + 2] = v.z; r.y[4 * q + 3] = v.w;
    }
  }
  // the blank emission is wave-uniform; fetched as a VECTOR load (vz = opaque zero): a scalar load returns out of
  // order, so its use forces lgkmcnt(0), i.e. a wait for the youngest prefetch of the ring instead of the oldest
  r.bl = base[UP + vz];
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
template <int KIND, int NL, int DIR>
struct Scan {
  float c[NL], o[NL], cx;
  double off;
  bool norep[NL], norep_next[NL];

  __device__ __forceinline__ void step(const ERow<NL> &e) {
    const float bl = e.bl;
    if constexpr (KIND == 0 && DIR == 0) {
      // alpha step (classic_ctc_loss.py:415-451)
      float m[NL], x[NL];
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        m[j] = lse2(c[j], o[j]);
        x[j] = norep_next[j] ? m[j] : c[j];  // what position l+1 may continue from
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
      // beta step (classic_ctc_loss.py:349-364)
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
      // simplified alpha step (simplified_ctc_loss.py:393-424)
      float pin0 = from_prev_lane(c[NL - 1], cx);
#pragma unroll
      for (int j = NL - 1; j >= 0; --j) {
        float pin = (j == 0) ? pin0 : c[j - 1];
        c[j] = lse2(bl + c[j], e.y[j] + pin);
      }
      cx += bl;
    } else {
      // simplified beta step (simplified_ctc_loss.py:327-343)
      float nin = from_next_lane(c[0], cx);
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        float nx = (j == NL - 1) ? nin : c[j + 1];
        c[j] = lse2(bl + c[j], e.y[j] + nx);
      }
      cx += bl;
    }
  }

  // exact renormalisation: subtract the row maximum, remember it in `off` (branch-free)
  __device__ __forceinline__ void renorm() {
    float mx = cx;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      mx = fmaxf(mx, c[j]);
      if constexpr (KIND == 0) mx = fmaxf(mx, o[j]);
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
  }

  // row layout: see Layout in ctc_common.h.  The 16-byte tail (l = 0 state + offset) is wave-uniform data
  // written by every lane to the same address, which keeps the store branch-free.
  __device__ __forceinline__ void store_row(float *__restrict__ row, int lane, int UP) const {
    const float oh = (float)off;
    const float ol = (float)(off - (double)oh);
    if constexpr (DIR == 0) {
      if constexpr (KIND == 0) {
        store_pairs<NL>(row, lane, c, o);
        *reinterpret_cast<float4 *>(row + 2 * UP) = make_float4(cx, NEG, oh, ol);
      } else {
        store_singles<NL>(row, lane, c);
        *reinterpret_cast<float4 *>(row + UP) = make_float4(cx, 0.f, oh, ol);
      }
    } else {
      float cs[NL];  // state of label position l = i+1 lives in the next slot's c
#pragma unroll
      for (int j = 0; j < NL - 1; ++j) cs[j] = c[j + 1];
      cs[NL - 1] = from_next_lane(c[0], cx);
      const float c00 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(c[0])));  // state l = 0 (lane 0, slot 0)
      if constexpr (KIND == 0) {
        store_pairs<NL>(row, lane, cs, o);
        *reinterpret_cast<float4 *>(row + 2 * UP) = make_float4(c00, c00, oh, ol);
      } else {
        store_singles<NL>(row, lane, cs);
        *reinterpret_cast<float4 *>(row + UP) = make_float4(c00, 0.f, oh, ol);
      }
    }
  }
};

template <int KIND, int NL, int DIR>
__device__ __forceinline__ void scan_body(const Problem &p, const Layout &L, const float *__restrict__ emis,
                                          float *__restrict__ rows_all, double *__restrict__ logp,
                                          float *__restrict__ loss) {
  const int lane = threadIdx.x;
  const int b = blockIdx.x;
  const int T = p.T, UP = L.UP;
  const int len = clampi(p.logit_length[b], 0, T);
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
        load_erow<NL>(buf[d], erow_ptr(k0 + d + PF), lane, UP, vz);  // clamped: re-reads the last row near the end
        if (d == RENORM - 1) S.renorm();
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
      if (i == ll - 1) mine = (KIND == 0) ? lse2(S.c[j], S.o[j]) : S.c[j];
    }
    float v = (ll == 0) ? S.cx : wave_max(mine);
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
  const long goff = (long)b * p.gsb + (long)t * p.gst;
  float *g = grad + goff;                                                        // valid for float32 only
  unsigned short *gh = reinterpret_cast<unsigned short *>(grad) + goff;           // valid for bfloat16 only
  const bool gbf = p.gdtype != 0;
  // outputs are written once and not read here: non-temporal stores
  auto gput = [&](int k, float v) { if (gbf) __builtin_nontemporal_store(f32_to_bf16(v), gh + k); else __builtin_nontemporal_store(v, g + k); };
  auto gput4 = [&](int k, float4 r) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f v = {r.x, r.y, r.z, r.w};
    __builtin_nontemporal_store(v, reinterpret_cast<v4f *>(g + k));
  };
  const bool gvec = !gbf && ((V | goff) & 3) == 0 && (p.align_bits & 15) == 0;
  const int len = clampi(p.logit_length[b], 0, p.T);
  const double lp = logp[b];
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
  const int offpos = (KIND == 0 ? 2 * UP : UP) + 2;
  // alpha~ + beta~ + (offsets - log P) is summed in double: with logits ~1e10 the three addends are each ~1e10 and
  // cancel to O(1) (README.md:74-78 promises sane outputs there); float addition would lose the result entirely.
  const double scale = (double)ra[offpos] + (double)ra[offpos + 1] + (double)rb[offpos] + (double)rb[offpos + 1] - lp;
  auto post = [&](float a_, float b_) -> float { return fminf(fexp2((float)((double)a_ + (double)b_ + scale)), 1.0f); };  // a posterior never exceeds 1
  auto post3 = [&](float a_, float b_, float c_) -> float { return fminf(fexp2((float)((double)a_ + (double)b_ + (double)c_ + scale)), 1.0f); };
  float qblank = 0.f;
  if constexpr (KIND == 0) {
    for (int i = lane; i < UP; i += 64) {
      float2 a = *reinterpret_cast<const float2 *>(ra + 2 * i);
      float2 bb = *reinterpret_cast<const float2 *>(rb + 2 * i);
      qblank += post(a.x, bb.x);
      if (i < ll) {
        float q = post(a.y, bb.y);
        int tok = (i < p.label_stride) ? lab[i] : p.blank;
        if (tok >= 0 && tok < V && tok != p.blank) atomicAdd(&ubin[tok], tofix(q));
      }
    }
    if (lane == 0) qblank += post(ra[2 * UP], rb[2 * UP]);
  } else {
    const float *er = emis + row * (long)L.ERS;
    const float bl = er[UP];
    for (int i = lane; i < UP; i += 64) {
      float ai = ra[i], bi = rb[i];  // state l = i+1 in both rows
      qblank += post3(ai, bi, bl);
      if (i < ll) {
        float aprev = (i == 0) ? ra[UP] : ra[i - 1];  // a[t, l = i]
        float q = post3(aprev, er[i], bi);  // a[t, l=i] * y[t,i] * b[t+1, l=i+1]
        int tok = (i < p.label_stride) ? lab[i] : p.blank;
        if (tok >= 0 && tok < V && tok != p.blank) atomicAdd(&ubin[tok], tofix(q));
      }
    }
    if (lane == 0) qblank += post3(ra[UP], rb[UP], bl);
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
    const long xoff = (long)b * p.xsb + (long)t * p.xst;
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
        const float xv = bf ? bf16_to_f32(xh[k]) : x[k];
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

// Wide vocabularies (V > 1024, float32 rows, 16-byte aligned): the same gradient with the vocabulary walked in passes of
// 1024 columns.  The posterior of every label position is computed ONCE into a per-wavefront table (fixed point); each
// pass zeroes a 4 KB bin array, adds the positions whose token falls into it, and streams its 1024 columns (logits in,
// gradient out).  LDS per wavefront is 8 KB whatever V is (grad_kernel needs 4 V bytes: at V = 8192 that left 4
// wavefronts per CU and 2.3 TB/s), and the bins are touched twice per column instead of four times.
template <int KIND>
__global__ __launch_bounds__(256) void grad_wide_kernel(Problem p, Layout L, const float *__restrict__ emis,
                                                         const float *__restrict__ alpha, const float *__restrict__ beta,
                                                         const double *__restrict__ logp, const float *__restrict__ d_loss,
                                                         float *__restrict__ grad) {
  constexpr int CH = 1024;
  __shared__ __attribute__((aligned(16))) unsigned bins_s[4][CH];
  __shared__ unsigned qtab_s[4][CTC_AMD_MAX_U];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long row = (long)blockIdx.x * 4 + w;
  if (row >= (long)p.B * p.T) return;
  const int b = (int)(row / p.T), t = (int)(row % p.T);
  const int V = p.V, UP = L.UP;
  float *g = grad + (long)b * p.gsb + (long)t * p.gst;
  typedef float v4f __attribute__((ext_vector_type(4)));
  auto gput4 = [&](int k, float4 r) {
    v4f v = {r.x, r.y, r.z, r.w};
    __builtin_nontemporal_store(v, reinterpret_cast<v4f *>(g + k));
  };
  const int len = clampi(p.logit_length[b], 0, p.T);
  const double lp = logp[b];
  if (t >= len || lp == -INFINITY) {  // padded frames and infeasible samples: exactly zero (base_loss.py:283-298)
    for (int k = lane * 4; k < V; k += 256) gput4(k, make_float4(0.f, 0.f, 0.f, 0.f));
    return;
  }
  const int ll = p.label_length[b] < 0 ? 0 : p.label_length[b];
  unsigned *bins = bins_s[w];
  unsigned *qtab = qtab_s[w];
  auto tofix = [](float q) -> unsigned { return (unsigned)(fminf(q, 1.0f) * 1073741824.0f + 0.5f); };
  const int32_t *lab = p.labels + (long)b * p.label_stride;
  const float *ra = alpha + ((long)b * (p.T + 1) + (KIND == 0 ? t + 1 : t)) * L.SRS;
  const float *rb = beta + ((long)b * (p.T + 1) + t + 1) * L.SRS;
  const int offpos = (KIND == 0 ? 2 * UP : UP) + 2;
  const double scale = (double)ra[offpos] + (double)ra[offpos + 1] + (double)rb[offpos] + (double)rb[offpos + 1] - lp;
  auto post = [&](float a_, float b_) -> float { return fminf(fexp2((float)((double)a_ + (double)b_ + scale)), 1.0f); };
  auto post3 = [&](float a_, float b_, float c_) -> float { return fminf(fexp2((float)((double)a_ + (double)b_ + (double)c_ + scale)), 1.0f); };
  // posteriors of the label positions (see grad_kernel for the regrouping), once per row
  float qblank = 0.f;
  if constexpr (KIND == 0) {
    for (int i = lane; i < UP; i += 64) {
      const float2 a = *reinterpret_cast<const float2 *>(ra + 2 * i);
      const float2 bb = *reinterpret_cast<const float2 *>(rb + 2 * i);
      qblank += post(a.x, bb.x);
      qtab[i] = (i < ll) ? tofix(post(a.y, bb.y)) : 0u;
    }
    if (lane == 0) qblank += post(ra[2 * UP], rb[2 * UP]);
  } else {
    const float *er = emis + row * (long)L.ERS;
    const float bl = er[UP];
    for (int i = lane; i < UP; i += 64) {
      const float ai = ra[i], bi = rb[i];
      qblank += post3(ai, bi, bl);
      const float aprev = (i == 0) ? ra[UP] : ra[i - 1];
      qtab[i] = (i < ll) ? tofix(post3(aprev, er[i], bi)) : 0u;
    }
    if (lane == 0) qblank += post3(ra[UP], rb[UP], bl);
  }
  const unsigned qbfix = tofix(wave_sum(qblank));
  const float dl = d_loss ? d_loss[b] : 1.0f;
  const float *x = p.logits + (long)b * p.xsb + (long)t * p.xst;
  const float mx = emis[row * (long)L.ERS + UP + 1];
  const float l2s = emis[row * (long)L.ERS + UP + 2];
  const bool wrt_logits = p.wrt == 0;
  for (int c0 = 0; c0 < V; c0 += CH) {
    float4 xv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // this pass's logits first: four loads in flight under the LDS work
      const int k = c0 + lane * 4 + 256 * q;
      xv[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (wrt_logits && k < V) xv[q] = *reinterpret_cast<const float4 *>(x + k);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<uint4 *>(bins + lane * 4 + 256 * q) = make_uint4(0u, 0u, 0u, 0u);
    wave_lds_fence();  // (LDS operations of one wavefront execute in program order)
    for (int i = lane; i < ll; i += 64) {
      const int tok = (i < p.label_stride) ? lab[i] : p.blank;
      const unsigned r = (unsigned)(tok - c0);
      if (tok >= 0 && tok < V && tok != p.blank && r < (unsigned)CH) atomicAdd(&bins[r], qtab[i]);
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
    wave_lds_fence();
  }
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
template <int KIND, int NL>
__global__ __launch_bounds__(64) void scan_kernel(Problem p, Layout L, const float *__restrict__ emis,
                                                   float *__restrict__ alpha, float *__restrict__ beta,
                                                   double *__restrict__ logp, float *__restrict__ loss, const int *__restrict__ only_if) {
  if (only_if && only_if[blockIdx.x] == 0) return;  // (selected utterances only: see emit_sel_kernel)
  if (blockIdx.y == 0) scan_body<KIND, NL, 0>(p, L, emis, alpha, logp, loss);
  else scan_body<KIND, NL, 1>(p, L, emis, beta, logp, loss);
}

template <int KIND, int NL>
static void launch_scan_nl(const Problem &p, const Layout &L, const float *emis, float *alpha, float *beta, double *logp,
                           float *loss, int ndir, const int *only_if, hipStream_t st) {
  hipLaunchKernelGGL((scan_kernel<KIND, NL>), dim3(p.B, ndir), dim3(64), 0, st, p, L, emis, alpha, beta, logp, loss, only_if);
}

template <int KIND>
static hipError_t launch_scan(const Problem &p, const Layout &L, const float *emis, float *alpha, float *beta,
                              double *logp, float *loss, int ndir, const int *only_if, hipStream_t st) {
  switch (L.NL) {
    case 1: launch_scan_nl<KIND, 1>(p, L, emis, alpha, beta, logp, loss, ndir, only_if, st); break;
    case 2: launch_scan_nl<KIND, 2>(p, L, emis, alpha, beta, logp, loss, ndir, only_if, st); break;
    case 4: launch_scan_nl<KIND, 4>(p, L, emis, alpha, beta, logp, loss, ndir, only_if, st); break;
    case 8: launch_scan_nl<KIND, 8>(p, L, emis, alpha, beta, logp, loss, ndir, only_if, st); break;
    case 16: launch_scan_nl<KIND, 16>(p, L, emis, alpha, beta, logp, loss, ndir, only_if, st); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// grid of the selected-utterance kernels: frames along x, utterances strided along y
dim3 sel_grid(int B, int T) {
  const int gx = (T + 3) / 4 < 256 ? (T + 3) / 4 : 256;
  return dim3(gx < 1 ? 1 : gx, B < 8 ? (B < 1 ? 1 : B) : 8);
}

hipError_t run_emit_scan(const Problem &p, const Layout &L, char *ws, float *loss, int ndir, hipStream_t st, const int *only_if) {
  float *emis = reinterpret_cast<float *>(ws + L.off_emis);
  float *alpha = reinterpret_cast<float *>(ws + L.off_alpha);
  float *beta = reinterpret_cast<float *>(ws + L.off_beta);
  double *logp = reinterpret_cast<double *>(ws + L.off_logp);
  const long rows = (long)p.B * p.T;
  if (rows > 0) {
    const bool four = p.V <= 512 && p.xdtype == 0 && (p.align_bits & 15) == 0 && ((p.V | p.xsb | p.xst) & 3) == 0;
    const long waves = (long)p.B * ((p.T + 3) / 4);
    if (only_if) hipLaunchKernelGGL(emit_sel_kernel, sel_grid(p.B, p.T), dim3(256), 0, st, p, L, emis, only_if);
    else if (four) hipLaunchKernelGGL(emit4_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, p, L, emis);
    else hipLaunchKernelGGL(emit_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, p, L, emis);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  if (p.B == 0) return hipSuccess;
  return p.kind == 0 ? launch_scan<0>(p, L, emis, alpha, beta, logp, loss, ndir, only_if, st)
                     : launch_scan<1>(p, L, emis, alpha, beta, logp, loss, ndir, only_if, st);
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
    if (p.kind == 0) hipLaunchKernelGGL(grad_wide_kernel<0>, grid, block, 0, st, p, L, emis, alpha, beta, logp, d_loss, grad);
    else hipLaunchKernelGGL(grad_wide_kernel<1>, grid, block, 0, st, p, L, emis, alpha, beta, logp, d_loss, grad);
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

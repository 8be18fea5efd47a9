// Fused Hessian-vector product for gfx950: the decomposition of ctc_fused6.hip (one workgroup per utterance; two main chains
// that meet in the middle, two recompute chains, helpers; one checkpoint row per block in HBM, everything else in LDS) with a
// TANGENT riding on every lattice value, in the same linear-domain number format (float32 mantissas, one integer exponent per
// lane shared by the values and their tangents).  No alpha / beta / d alpha / d beta row ever leaves the CU; the five
// launches and ~5.2 GB of row traffic of the log-domain pipeline (ctc_hvp.hip) become one launch that reads logits and
// vector twice and writes the product once.
//
//   out[b,t,k] = sum_{t2,k2} H[b,t,k,t2,k2] v[b,t2,k2]       H = Hessian of the loss w.r.t. LOGITS (README.md:58-71)
//              = s_t[k] (v_t[k] - s_t . v_t) - d/dv posterior_t[k]       (gradient = softmax - posterior)
//
// Replaces gradient_fn.backprop (base_loss.py:157-175).  Tangent-mode recursion (classic_ctc_loss.py:349-364,415-451,
// simplified_ctc_loss.py:327-343,393-424 differentiated):
//   * direction in log-probability space u_t[k] = v_t[k] - s_t . v_t; posteriors do not change when u_t is shifted by a per-
//     frame constant (every path emits exactly one symbol per frame), so the lattice uses the BLANK GAUGE w_t[k] = v_t[k] -
//     v_t[blank]: blank transitions carry no tangent term at all and the common component of the tangents stays small;
//   * value step  o' = y (o + xin), c' = bl m   ->   tangent  do' = w o' + y (do + dxin), dc' = bl dm   (emission tangent dy = y w);
//   * posterior of a state q = alpha beta / P  ->  dq = (d alpha beta + alpha d beta) / P - q dlogP, dlogP = dP / P from the
//     meeting point of the chains;
//   * dq is scattered by token with integer LDS atomics; the fixed-point scale is chosen per frame from the wave-wide sum of
//     |dq| (a bin can never exceed it), so no bound on the tangents is assumed.
// What the number format cannot hold is flagged per utterance exactly as in ctc_fused6.hip (D1..D6); a flagged utterance is
// redone inside the same launch by its own workgroup with the log-domain building blocks of ctc_hvp.hip (emit -> scan ->
// tangent emissions -> tangent scan -> output rows, over the full-row regions of the workspace): a call is one launch whatever
// it meets, and a batch in which every utterance is flagged costs what the five-launch pipeline costs.
//
// Instantiated for logits input, contiguous float32 [B,T,V] with V <= 256 (V % 4 == 0) and U <= 128 (one or two label positions
// per lane): 10 wavefronts, 6-frame blocks, 150 KB of LDS.  Other shapes keep the log-domain pipeline.
#include "ctc_fused_common.h"
#include "ctc_swap_reduce.h"
#include "ctc_hvp_fused.h"
#include "ctc_linear_flags.h"
#include "ctc_v1_device.h"   // emit_row, scan_body: the log-domain building blocks, run in this launch for flagged utterances
#include "ctc_hvp_device.h"  // temit_row, tscan_body, hvp_out_row

#ifndef CTC_FUSED_KIND
#error "compile with -DCTC_FUSED_KIND=0 (classic) or 1 (simplified)"
#endif

namespace ctc {
namespace hvpf {

using namespace ctc::fused;

#ifndef CTC_HVPF_RN
#define CTC_HVPF_RN 3   // frames between renormalisations of a chain (experiment builds: scripts/build_hvp_variant.sh -DCTC_HVPF_RN=6)
#endif
constexpr int BLK = HVPF_BLK, NH = 3, RN = CTC_HVPF_RN, NG = BLK / RN, NW = 4 + 2 * NH;  // 10 wavefronts: 4 chains + 3 helpers a side
constexpr int V = 256;
using linear::DEAD; using linear::GAP; using linear::GAP_WIDE; using linear::DOWN_MAX; using linear::DECAY_MAX; using linear::KK_MAX;
using linear::EMIS_MIN; using linear::MASS_TOL;  // (ctc_linear_flags.h: one copy for this kernel and ctc_fused6.hip)

// packed float32 pairs and the one-instruction inflow for the classic two-positions-per-lane chains (as ctc_fused6.hip, r04)
typedef float f2v __attribute__((ext_vector_type(2)));
#ifndef CTC_HVPF_PACKED
#define CTC_HVPF_PACKED 1
#endif
template <int DIR>
__device__ __forceinline__ void fmac_from_upstream(float &acc, float x, float sc) {
  if constexpr (DIR == 0) asm("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(sc));
  else asm("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(sc));
}

__device__ __forceinline__ void block_barrier_raw() {
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's LDS writes have landed; vmcnt untouched
  __builtin_amdgcn_s_barrier();
}
__device__ __forceinline__ int from_prev_lane_i(int x, int fill) { return __builtin_amdgcn_update_dpp(fill, x, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ int from_next_lane_i(int x, int fill) { return __builtin_amdgcn_update_dpp(fill, x, 0x130, 0xf, 0xf, false); }
__device__ __forceinline__ float ldexp_f(float x, int e) { return __builtin_ldexpf(x, e); }
__device__ __forceinline__ int frexp_e(float x) { return __builtin_amdgcn_frexp_expf(x); }
__device__ __forceinline__ int readlane_i(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }

template <int NL> struct Cfg {
  static constexpr int UP = 64 * NL;
  static constexpr int ES = 2 * UP + 4;   // E row: y[UP], bl at [UP], w[UP] at [UP + 4]
  static constexpr int RS = 4 * UP + 8;   // R row / checkpoint row: per lane (c, o) pairs then (dc, do) pairs; tail (cx, kx, dcx, -)
  static constexpr int LV = (RN + NL - 1) / NL;
};

template <int KIND, int NL>
struct Lds {
  using C = Cfg<NL>;
  float E[2][3][BLK][C::ES];
  float R[2][3][BLK][C::RS];
  int kg[2][3][NG][64];
  float kl[2][3][BLK][64];
  float xcopy[2 * NH + 2][V + 4];  // gather copies: helpers, then the two recompute wavefronts (E stage of phase 1)
  float vcopy[2 * NH + 2][V + 4];
  int bins[2 * NH][V + 4];
  float dump[NW][64];
  double l2s[NW];
  int flag, feasible, lp_int, mode;
  float cf, dlp;
};

struct Geo {
  int len, G, tmb, tm, NB;
  // NI1 / NI2: block iterations (= workgroup barriers) of phase 1 / phase 2, the same for every wavefront: NB + 1 / NB + 3 rounded
  // up to the depth of the register rings of the wavefronts that load rows (4 / 3), so that their unrolled loops need no guard
  // per slot (see estage1)
  int NI1, NI2;
  __device__ __forceinline__ void init(int len_) {
    len = len_; G = (len + BLK - 1) / BLK; tmb = G / 2; tm = tmb * BLK; NB = G - tmb;
    NI1 = (NB + 1 + 3) / 4 * 4; NI2 = (NB + 3 + 2) / 3 * 3;
  }
  __device__ __forceinline__ int nvof(int g) const { int r = len - BLK * g; return r < BLK ? r : BLK; }
  __device__ __forceinline__ int nblocks(int phase, int side) const { return (phase == 1) == (side == 0) ? tmb : G - tmb; }
  __device__ __forceinline__ int absblock(int phase, int side, int j) const {
    if (phase == 1) return side == 0 ? j : G - 1 - j;
    return side == 0 ? tmb + j : tmb - 1 - j;
  }
  __device__ __forceinline__ int frame(int side, int g, int d) const { return side == 0 ? BLK * g + d : BLK * g + nvof(g) - 1 - d; }
  __device__ __forceinline__ int slot(int t) const { return (t + BLK - 1) / BLK; }
};

// a lane's (first, second) pairs: 2 NL floats at p
template <int NL>
__device__ __forceinline__ void ld_pairs(const float *p, float (&a)[NL], float (&b)[NL]) {
  if constexpr (NL == 1) { const float2 t = *reinterpret_cast<const float2 *>(p); a[0] = t.x; b[0] = t.y; }
  else { const float4 t = *reinterpret_cast<const float4 *>(p); a[0] = t.x; b[0] = t.y; a[1] = t.z; b[1] = t.w; }
}
template <int NL>
__device__ __forceinline__ void st_pairs(float *p, const float (&a)[NL], const float (&b)[NL]) {
  if constexpr (NL == 1) *reinterpret_cast<float2 *>(p) = make_float2(a[0], b[0]);
  else *reinterpret_cast<float4 *>(p) = make_float4(a[0], b[0], a[1], b[1]);
}
template <int NL>
__device__ __forceinline__ void ld_slots(const float *p, float (&v)[NL]) {
  if constexpr (NL == 1) v[0] = p[0];
  else { const float2 t = *reinterpret_cast<const float2 *>(p); v[0] = t.x; v[1] = t.y; }
}
template <int NL>
__device__ __forceinline__ void st_slots(float *p, const float (&v)[NL]) {
  if constexpr (NL == 1) p[0] = v[0];
  else *reinterpret_cast<float2 *>(p) = make_float2(v[0], v[1]);
}

// per-frame emissions, linear: y[j] = exp(x[label[i]] - rowmax), bl = exp(x[blank] - rowmax); w[j] = v[label[i]] - v[blank]
template <int NL>
struct Emis {
  float y[NL], w[NL];
  float bl;
};
template <int NL>
__device__ __forceinline__ void read_E(const float *row, int lane, Emis<NL> &e) {
  ld_slots<NL>(row + lane * NL, e.y);
  ld_slots<NL>(row + Cfg<NL>::UP + 4 + lane * NL, e.w);
  e.bl = row[Cfg<NL>::UP];
}
template <int NL>
__device__ __forceinline__ void write_E(float *row, float *dump, int lane, const Emis<NL> &e) {
  st_slots<NL>(row + lane * NL, e.y);
  st_slots<NL>(row + Cfg<NL>::UP + 4 + lane * NL, e.w);
  float *tq = (lane == 0) ? row + Cfg<NL>::UP : dump + lane;
  *tq = e.bl;
}

// one lattice row of the OTHER direction with its tangents, in that direction's native slot order and lane exponents
template <int NL>
struct RRow {
  float c[NL], o[NL], dc[NL], dob[NL];
  float cx, dcx;
  int kx;
};
template <int NL>
__device__ __forceinline__ void read_R(const float *row, int lane, RRow<NL> &r) {
  ld_pairs<NL>(row + 4 * lane * NL, r.c, r.o);
  ld_pairs<NL>(row + 4 * lane * NL + 2 * NL, r.dc, r.dob);
  const float4 t = *reinterpret_cast<const float4 *>(row + 4 * Cfg<NL>::UP);
  r.cx = t.x; r.kx = __float_as_int(t.y); r.dcx = t.z;
}

// ------------------------------------------------------------------------------------------------
// Lattice state of one direction with tangents (slot / state conventions of ctc_fused6.hip Chain<>).
// ------------------------------------------------------------------------------------------------
template <int KIND, int NL, int DIR>
struct Chain {
  float c[NL], o[NL], dc[NL], dob[NL], cx, dcx;
  int k, kx, dk;
  bool norep[NL], norep_next[NL];
  int flag;
  static constexpr bool PACKED = (CTC_HVPF_PACKED != 0) && KIND == 0 && NL == 2;
  float nrf[NL];  // PACKED: 1.0 where the repeat rule lets the diagonal pass (norep_next for A, norep for B)
  float sc = 1.f, scb = 0.f;  // PACKED: 2^dk as a float (0 below 2^-126), and the same on the boundary lane only
  bool boundary = false;
  __device__ __forceinline__ void set_scale() {
    if constexpr (PACKED) {
      sc = (dk < -126) ? 0.f : ldexp_f(1.f, dk < 127 ? dk : 127);
      scb = boundary ? sc : 0.f;
    }
  }
  bool alive = false;
  int age = 0;
  bool relevant = true;

  __device__ __forceinline__ void init_labels(const Problem &p, int b, int lane, int ll) {
    const int32_t *lab = p.labels + (long)b * p.label_stride;
    auto tok = [&](int i) -> int { return (i >= 0 && i < ll) ? ((i < p.label_stride) ? lab[i] : p.blank) : -1 - (i < 0); };
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const int i = lane * NL + j;
      const int tk = tok(i);
      norep[j] = (i == 0) || tk != tok(i - 1);
      norep_next[j] = tok(i + 1) != tk;
      nrf[j] = ((DIR == 0) ? norep_next[j] : norep[j]) ? 1.f : 0.f;
      c[j] = 0.f; o[j] = 0.f; dc[j] = 0.f; dob[j] = 0.f;
    }
    cx = 0.f; dcx = 0.f; k = DEAD; kx = DEAD; dk = 0; flag = 0;
    boundary = lane == (DIR == 0 ? 0 : 63);
    set_scale();
    relevant = lane * NL <= ll;
  }

  __device__ __forceinline__ void start(int lane, int ll) {
    constexpr int UP = Cfg<NL>::UP;
    if constexpr (DIR == 0) {
      cx = 1.f; kx = 0;
    } else {
      if (ll == UP) { cx = 1.f; kx = 0; }
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        const int i = lane * NL + j;
        if (i == ll) { c[j] = 1.f; k = 0; }
        if (KIND == 0 && i == ll - 1) { o[j] = 1.f; k = 0; }
      }
    }
    renorm();
    flag = 0;
  }

  // one lattice step, values and tangents (blank gauge: d bl = 0, d y = y w)
  __device__ __forceinline__ void step(const Emis<NL> &e) {
    const float bl = e.bl;
    if constexpr (PACKED && DIR == 0) {
      // the generic recursion below, two label positions (and value + tangent side by side) per instruction
      const f2v C = {c[0], c[1]}, O = {o[0], o[1]}, DC = {dc[0], dc[1]}, DO = {dob[0], dob[1]};
      const f2v Y = {e.y[0], e.y[1]}, W = {e.w[0], e.w[1]}, NR = {nrf[0], nrf[1]};
      const f2v X = __builtin_elementwise_fma(O, NR, C), DX = __builtin_elementwise_fma(DO, NR, DC);
      const f2v M = C + O, DM = DC + DO;
      float olo = __builtin_fmaf(cx, scb, O.x), dlo = __builtin_fmaf(dcx, scb, DO.x);
      fmac_from_upstream<0>(olo, X.y, sc);
      fmac_from_upstream<0>(dlo, DX.y, sc);
      const f2v OS = {olo, O.y + X.x}, DS = {dlo, DO.y + DX.x};
      const f2v ON = Y * OS;
      const f2v DN = __builtin_elementwise_fma(W, ON, Y * DS);
      const f2v CN = M * bl, DCN = DM * bl;
      o[0] = ON.x; o[1] = ON.y; dob[0] = DN.x; dob[1] = DN.y;
      c[0] = CN.x; c[1] = CN.y; dc[0] = DCN.x; dc[1] = DCN.y;
      cx *= bl; dcx *= bl;
    } else if constexpr (PACKED && DIR == 1) {
      const f2v C = {c[0], c[1]}, O = {o[0], o[1]}, DC = {dc[0], dc[1]}, DO = {dob[0], dob[1]};
      const f2v Y = {e.y[0], e.y[1]}, W = {e.w[0], e.w[1]}, NR = {nrf[0], nrf[1]};
      const f2v H = C * bl, DH = DC * bl;
      const f2v EE = Y * O;
      const f2v DEE = __builtin_elementwise_fma(W, EE, Y * DO);
      const f2v PN = H + EE, DPN = DH + DEE;
      const f2v X = __builtin_elementwise_fma(EE, NR, H), DX = __builtin_elementwise_fma(DEE, NR, DH);
      cx *= bl; dcx *= bl;
      float ohi = __builtin_fmaf(cx, scb, EE.y), dhi = __builtin_fmaf(dcx, scb, DEE.y);
      fmac_from_upstream<1>(ohi, X.x, sc);
      fmac_from_upstream<1>(dhi, DX.x, sc);
      o[0] = EE.x + X.y; o[1] = ohi; dob[0] = DEE.x + DX.y; dob[1] = dhi;
      c[0] = PN.x; c[1] = PN.y; dc[0] = DPN.x; dc[1] = DPN.y;
    } else if constexpr (KIND == 0 && DIR == 0) {
      float m[NL], x[NL], dm[NL], dx[NL];
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        m[j] = c[j] + o[j];
        dm[j] = dc[j] + dob[j];
        x[j] = norep_next[j] ? m[j] : c[j];
        dx[j] = norep_next[j] ? dm[j] : dc[j];
      }
      const float xin0 = ldexp_f(from_prev_lane(x[NL - 1], cx), dk);
      const float dxin0 = ldexp_f(from_prev_lane(dx[NL - 1], dcx), dk);
#pragma unroll
      for (int j = NL - 1; j >= 0; --j) {
        const float xin = (j == 0) ? xin0 : x[j - 1], dxin = (j == 0) ? dxin0 : dx[j - 1];
        const float on = e.y[j] * (o[j] + xin);
        dob[j] = fmaf(e.w[j], on, e.y[j] * (dob[j] + dxin));
        o[j] = on;
        c[j] = bl * m[j];
        dc[j] = bl * dm[j];
      }
      cx *= bl; dcx *= bl;
    } else if constexpr (KIND == 0 && DIR == 1) {
      float h[NL], ee[NL], pn[NL], x[NL], dh[NL], dee[NL], dpn[NL], dx[NL];
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        h[j] = bl * c[j];
        dh[j] = bl * dc[j];
        ee[j] = e.y[j] * o[j];
        dee[j] = fmaf(e.w[j], ee[j], e.y[j] * dob[j]);
        pn[j] = h[j] + ee[j];
        dpn[j] = dh[j] + dee[j];
        x[j] = norep[j] ? pn[j] : h[j];
        dx[j] = norep[j] ? dpn[j] : dh[j];
      }
      cx *= bl; dcx *= bl;
      const float xinl = ldexp_f(from_next_lane(x[0], cx), dk);
      const float dxinl = ldexp_f(from_next_lane(dx[0], dcx), dk);
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        const float xin = (j == NL - 1) ? xinl : x[j + 1], dxin = (j == NL - 1) ? dxinl : dx[j + 1];
        o[j] = xin + ee[j];
        dob[j] = dxin + dee[j];
        c[j] = pn[j];
        dc[j] = dpn[j];
      }
    } else if constexpr (KIND == 1 && DIR == 0) {
      const float pin0 = ldexp_f(from_prev_lane(c[NL - 1], cx), dk);
      const float dpin0 = ldexp_f(from_prev_lane(dc[NL - 1], dcx), dk);
#pragma unroll
      for (int j = NL - 1; j >= 0; --j) {
        const float pin = (j == 0) ? pin0 : c[j - 1], dpin = (j == 0) ? dpin0 : dc[j - 1];
        const float yp = e.y[j] * pin;
        dc[j] = fmaf(bl, dc[j], fmaf(e.w[j], yp, e.y[j] * dpin));
        c[j] = fmaf(bl, c[j], yp);
      }
      cx *= bl; dcx *= bl;
    } else {
      const float nin = ldexp_f(from_next_lane(c[0], cx), dk);
      const float dnin = ldexp_f(from_next_lane(dc[0], dcx), dk);
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        const float nx = (j == NL - 1) ? nin : c[j + 1], dnx = (j == NL - 1) ? dnin : dc[j + 1];
        const float yn = e.y[j] * nx;
        dc[j] = fmaf(bl, dc[j], fmaf(e.w[j], yn, e.y[j] * dnx));
        c[j] = fmaf(bl, c[j], yn);
      }
      cx *= bl; dcx *= bl;
    }
  }

  // per-lane renormalisation, decided by the VALUES; the tangents follow with the same shift
  __device__ __forceinline__ void renorm() {
    constexpr int LV = Cfg<NL>::LV;
    float m = c[0];
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      if constexpr (KIND == 0) m = (j == 0) ? vmax_raw(m, o[0]) : vmax3_raw(m, c[j], o[j]);
      else if (j > 0) m = vmax_raw(m, c[j]);
    }
    const bool live = m > 0.f;
    const int fe = frexp_e(m);
    const int e_own = live ? fe + k : DEAD;
    const bool xlive = cx > 0.f;
    const int ex = xlive ? frexp_e(cx) + kx : DEAD;
    int kn = e_own;
    {
      const int nb = (DIR == 0) ? from_prev_lane_i(kn, ex) : from_next_lane_i(kn, ex);
      kn = imax(kn, nb - (LV == 1 ? GAP_WIDE : GAP));
    }
    {  // every level for every lane, with or without mass (r04, as ctc_fused6.hip: a steep profile of live lanes kept exponents 2^100
       // apart two lanes down after the one level, and the inflow overflowed when the bulk crossed two lanes within a period)
#pragma unroll
      for (int lv = 1; lv < LV; ++lv) {
        const int nb = (DIR == 0) ? from_prev_lane_i(kn, ex) : from_next_lane_i(kn, ex);
        kn = imax(kn, nb - GAP);
      }
    }
    kn = imax(kn, DEAD);
    const int d = k - kn;
    age = (live && alive) ? age + 1 : 0;
    flag |= (live && age >= 3 && d < -DOWN_MAX ? 4 : 0) | (live && fe < -DECAY_MAX ? 8 : 0) | (!live && alive ? 16 : 0);
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      c[j] = ldexp_f(c[j], d); dc[j] = ldexp_f(dc[j], d);
      if constexpr (KIND == 0) { o[j] = ldexp_f(o[j], d); dob[j] = ldexp_f(dob[j], d); }
    }
    k = kn;
    cx = ldexp_f(cx, kx - ex); dcx = ldexp_f(dcx, kx - ex);
    kx = ex;
    dk = ((DIR == 0) ? from_prev_lane_i(k, kx) : from_next_lane_i(k, kx)) - k;
    set_scale();
    alive = live;
  }
  __device__ __forceinline__ int flag_or() const {
    int f = 0;
#pragma unroll
    for (int bit = 4; bit <= 16; bit <<= 1) f |= (__builtin_amdgcn_ballot_w64((flag & bit) != 0) != 0) ? bit : 0;
    return f;
  }
};

// row with tangents in LDS (R row): lane region, tail by lane 0 (the others write `sink`, also LDS: one unconditional store)
template <int KIND, int NL, int DIR>
__device__ __forceinline__ void put_row(float *row, float *sink, int lane, const Chain<KIND, NL, DIR> &S) {
  st_pairs<NL>(row + 4 * lane * NL, S.c, S.o);
  st_pairs<NL>(row + 4 * lane * NL + 2 * NL, S.dc, S.dob);
  float *tq = (lane == 0) ? row + 4 * Cfg<NL>::UP : sink;
  *reinterpret_cast<float4 *>(tq) = make_float4(S.cx, __int_as_float(S.kx), S.dcx, 0.f);
}
// the same row as a checkpoint in HBM
template <int KIND, int NL, int DIR>
__device__ __forceinline__ void spill_row(float *__restrict__ row, int lane, const Chain<KIND, NL, DIR> &S) {
  st_pairs<NL>(row + 4 * lane * NL, S.c, S.o);
  st_pairs<NL>(row + 4 * lane * NL + 2 * NL, S.dc, S.dob);
  if (lane == 0) *reinterpret_cast<float4 *>(row + 4 * Cfg<NL>::UP) = make_float4(S.cx, __int_as_float(S.kx), S.dcx, 0.f);
}
template <int NL>
struct CkRow {
  RRow<NL> r;
  int k;
};
template <int NL>
__device__ __forceinline__ void load_ck(CkRow<NL> &ck, const float *__restrict__ rows, const int *__restrict__ kexp, int slot, int lane) {
  read_R<NL>(rows + (long)slot * Cfg<NL>::RS, lane, ck.r);
  ck.k = kexp[slot * 64 + lane];
}
template <int KIND, int NL, int DIR>
__device__ __forceinline__ void restore(Chain<KIND, NL, DIR> &S, const CkRow<NL> &ck) {
  float m = 0.f;
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    S.c[j] = ck.r.c[j]; S.o[j] = (KIND == 0) ? ck.r.o[j] : 0.f; S.dc[j] = ck.r.dc[j]; S.dob[j] = (KIND == 0) ? ck.r.dob[j] : 0.f;
    m = fmaxf(m, fmaxf(S.c[j], S.o[j]));
  }
  S.cx = ck.r.cx; S.dcx = ck.r.dcx; S.k = ck.k; S.kx = ck.r.kx;
  S.dk = ((DIR == 0) ? from_prev_lane_i(S.k, S.kx) : from_next_lane_i(S.k, S.kx)) - S.k;
  S.set_scale();
  S.alive = m > 0.f;
}

// ------------------------------------------------------------------------------------------------
// logits / vector rows: loads, statistics, gathers, the output row
// ------------------------------------------------------------------------------------------------
template <int KIND, int NL>
struct Rows {
  int tokoff[NL];   // byte offset of label[i] in the LDS copies (pad slot beyond label_length)
  bool valid[NL];
  float mb[4];      // 1.0 at this lane's element that is the blank column
  const float *xbase, *vbase;
  float *obase;
  float *xs, *vs;
  int *bins;
  int lane, blank, Vr;
  bool inrow;       // this lane's four columns exist (V may be smaller than 256)

  __device__ __forceinline__ void init(const Problem &p, int b, int lane_, int ll, const float *vec, float *out) {
    lane = lane_; blank = p.blank; Vr = p.V;
    inrow = lane * 4 < p.V;
    xbase = p.logits + (long)b * p.T * p.V;
    vbase = vec + (long)b * p.T * p.V;
    obase = out + (long)b * p.T * p.V;
    const int32_t *lab = p.labels + (long)b * p.label_stride;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const int i = lane * NL + j;
      const int tk = (i < ll) ? ((i < p.label_stride) ? lab[i] : p.blank) : -1;
      valid[j] = i < ll;
      tokoff[j] = 4 * ((tk >= 0 && tk < p.V && tk != p.blank) ? tk : V);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) mb[e] = (lane * 4 + e == p.blank) ? 1.f : 0.f;
  }
  // rows of frame t: logits (-inf beyond the vocabulary) and vector (0 beyond it); the address is clamped so that the load is
  // unconditional
  // The loads are RAW: mask_xv is applied where the rows are consumed, iterations later.  (r03 masked inside the load: a select
  // on a loaded register makes the compiler wait for the load on the spot -- s_waitcnt vmcnt(0) behind every pair of loads, so
  // nothing was ever in flight ahead of its use and every row cost a full memory round trip: phase 1 took 136 us with its
  // arithmetic switched off, for 524 MB that the same access pattern reads in 89 us -- scripts/r04_hbm_pattern.hip.)
  __device__ __forceinline__ void load_xv(float4 &x, float4 &v, int t) const {
    const int col = inrow ? lane * 4 : 0;
    x = *reinterpret_cast<const float4 *>(xbase + (long)t * Vr + col);
    v = *reinterpret_cast<const float4 *>(vbase + (long)t * Vr + col);
  }
  __device__ __forceinline__ void mask_xv(float4 &x, float4 &v) const {
    if (!inrow) { x = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY); v = make_float4(0.f, 0.f, 0.f, 0.f); }
  }
  __device__ __forceinline__ float4 expo(const float4 &x, float mxl) const {
    return make_float4(fexp2(fmaf(x.x, LOG2E, -mxl)), fexp2(fmaf(x.y, LOG2E, -mxl)), fexp2(fmaf(x.z, LOG2E, -mxl)), fexp2(fmaf(x.w, LOG2E, -mxl)));
  }
  // emission gather through LDS copies of the exponentiated row and of the vector row
  __device__ __forceinline__ void gather(const float4 &ev, const float4 &v, Emis<NL> &e) const {
    *reinterpret_cast<float4 *>(xs + lane * 4) = ev;
    *reinterpret_cast<float4 *>(vs + lane * 4) = v;
    const char *bx = reinterpret_cast<const char *>(xs), *bv = reinterpret_cast<const char *>(vs);
    const float vb = vs[blank];
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      e.y[j] = *reinterpret_cast<const float *>(bx + tokoff[j]);
      e.w[j] = valid[j] ? *reinterpret_cast<const float *>(bv + tokoff[j]) - vb : 0.f;
    }
    e.bl = xs[blank];
  }
  // statistics of one frame: mxl = rowmax log2 e, sum exp, sum exp v
  __device__ __forceinline__ void stats(const float4 &x, const float4 &v, float &mxl, float4 &ev, float &s, float &sv) const {
    float m = vmax_raw(vmax3_raw(x.x, x.y, x.z), x.w);
    m = wave_max_dpp(m);
    m = (m == -INFINITY) ? 0.f : m;
    mxl = m * LOG2E;
    ev = expo(x, mxl);
    float a = (ev.x + ev.y) + (ev.z + ev.w);
    float bq = fmaf(ev.x, v.x, ev.y * v.y) + fmaf(ev.z, v.z, ev.w * v.w);
    float ab[2] = {a, bq};
    const float both = swap_reduce<2, false>(ab);
    s = readlane_f(both, SwapLanes<2>::lane(0));
    sv = readlane_f(both, SwapLanes<2>::lane(1));
  }
  __device__ __forceinline__ void zero_rows(int t_from, int t_to) const {
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f z = {0.f, 0.f, 0.f, 0.f};
    if (inrow)
      for (int t = t_from; t < t_to; ++t) __builtin_nontemporal_store(z, reinterpret_cast<v4f *>(obase + (long)t * Vr + lane * 4));
  }
  // output row of frame t.  dqt[j]: tangent of the token posterior of slot j, dqb: of the blank posterior (this lane's part
  // already summed over the wave), both in units of 2^-30; asum = wave-wide sum of |dqt| (same units): no bin can exceed it.
  // z = the softmax part s_t[k] (v_t[k] - s_t . v_t) of this lane's four columns.
  __device__ __forceinline__ void out_row(int t, float dqb, const float (&dqt)[NL], float asum, const float4 &z) const {
    *reinterpret_cast<int4 *>(bins + lane * 4) = make_int4(0, 0, 0, 0);
    // fixed point: 2^28 units for the wave-wide sum of magnitudes (exact integer adds, any order)
    const int ea = (asum > 0.f) ? frexp_e(asum) : 0;
    const float up = ldexp_f(1.f, 28 - ea), down = ldexp_f(1.f, ea - 58);  // (... and back, including the 2^-30 of the units)
    char *bb = reinterpret_cast<char *>(bins);
#pragma unroll
    for (int j = 0; j < NL; ++j)
      if (valid[j]) atomicAdd(reinterpret_cast<int *>(bb + tokoff[j]), __float2int_rn(dqt[j] * up));
    wave_lds_fence();
    const int4 pu = *reinterpret_cast<const int4 *>(bins + lane * 4);
    const float qb = dqb * 9.31322574615478515625e-10f;
    const float4 dq = make_float4(fmaf((float)pu.x, down, mb[0] * qb), fmaf((float)pu.y, down, mb[1] * qb),
                                  fmaf((float)pu.z, down, mb[2] * qb), fmaf((float)pu.w, down, mb[3] * qb));
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f r = {z.x - dq.x, z.y - dq.y, z.z - dq.z, z.w - dq.w};
    if (inrow) __builtin_nontemporal_store(r, reinterpret_cast<v4f *>(obase + (long)t * Vr + lane * 4));
  }
};

// ------------------------------------------------------------------------------------------------
// E stage of phase 1 (helpers and, before the meeting point, the recompute wavefronts): positions P0 .. P0+NQ-1 of every block of
// side SIDE; one block of look-ahead on the rows.  Records (mxl, 1/sum, s.v) per frame, accumulates log2 sum exp in double,
// tracks the smallest needed emission (D2).
// ------------------------------------------------------------------------------------------------
template <int KIND, int NL, int SIDE, int P0, int NQ>
__device__ __forceinline__ void estage1(const Rows<KIND, NL> &S, Lds<KIND, NL> &lds, const Geo &geo, float4 *__restrict__ stats, float *dump,
                                        int lane, int wave) {
  const int len = geo.len;
  const int nb = geo.nblocks(1, SIDE);
  auto fr = [&](int j, int d) -> int {
    int jj = j < nb ? j : nb - 1;
    jj = jj < 0 ? 0 : jj;
    const int g = geo.absblock(1, SIDE, jj);
    const int nv = geo.nvof(g);
    int dd = d < nv ? d : nv - 1;
    int t = geo.frame(SIDE, g, dd < 0 ? 0 : dd);
    t = t < len ? t : len - 1;
    return t < 0 ? 0 : t;
  };
  // rows are loaded PFD blocks ahead of their use, in a ring of register sets addressed by (block mod PFD) at COMPILE time (the
  // loop is unrolled by PFD): a 6-frame block lasts ~0.7 us, an HBM load under load ~2 us -- with one block of look-ahead every
  // iteration waited for memory (phase 1: 165 us instead of ~70)
#ifndef CTC_HVPF_PFD
#define CTC_HVPF_PFD 4
#endif
  constexpr int PFD = CTC_HVPF_PFD;
  float4 xb[PFD][NQ], vb[PFD][NQ];
  static_for<0, PFD>([&](auto R) {
    constexpr int r = decltype(R)::value;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      xb[r][q] = make_float4(0.f, 0.f, 0.f, 0.f); vb[r][q] = xb[r][q];
      if (nb > 0) S.load_xv(xb[r][q], vb[r][q], fr(r, P0 + q));
    }
  });
  double acc = 0.0;
  float zmin[NL], zb = 1.0f;
#pragma unroll
  for (int j = 0; j < NL; ++j) zmin[j] = 1.0f;
  auto body = [&](auto R, int it) __attribute__((always_inline)) {
    constexpr int r = decltype(R)::value;  // = it mod PFD
    const int j = it;
    if (j < nb) {
      const int g = geo.absblock(1, SIDE, j);
      const int nv = (lds.mode & 16) ? 0 : geo.nvof(g);  // (timing mode 16: the E stage of phase 1 keeps its loads and barriers only)
      float(*E)[Cfg<NL>::ES] = lds.E[SIDE][j % 3];
      float4 xq[NQ], vq[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) { xq[q] = xb[r][q]; vq[q] = vb[r][q]; S.mask_xv(xq[q], vq[q]); }
      float prod = 1.f;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int d = P0 + q;
        if (d < nv) {
          float mxl, s, sv;
          float4 ev;
          S.stats(xq[q], vq[q], mxl, ev, s, sv);
          Emis<NL> e;
          S.gather(ev, vq[q], e);
          write_E<NL>(E[d], dump, lane, e);
#pragma unroll
          for (int jj = 0; jj < NL; ++jj) zmin[jj] = vmin_raw(zmin[jj], e.y[jj]);
          zb = vmin_raw(zb, e.bl);
          prod *= s;
          const float inv = __builtin_amdgcn_rcpf(s);
          if (lane == 0) stats[geo.frame(SIDE, g, d)] = make_float4(mxl, inv, sv * inv, 0.f);
        }
      }
      acc += (double)flog2(prod);
    }
    // Refill the slot, PFD blocks ahead.  (a) AFTER its rows have been used: issued before, old and new rows were alive together,
    // the slot changed registers every iteration and the copy back at the end of the iteration waited for the loads just issued.
    // (b) On EVERY path (the index is clamped, the load is always legal), in a loop without a guard per slot (Geo::NI1): the
    // s_waitcnt in front of a slot's use is a COUNT of younger loads that may stay in flight, fixed at compile time -- with a
    // path on which the younger refills are skipped (the tail: j >= nb) that count is 0, on every iteration.  r03 had both:
    // s_waitcnt vmcnt(0) behind each load, nothing ever in flight across an iteration, 136 us for a phase 1 whose loads take 89.
#ifdef CTC_HVPF_DIAG_NOLOAD
    if (!(lds.mode & 16))
#endif
#pragma unroll
    for (int q = 0; q < NQ; ++q) S.load_xv(xb[r][q], vb[r][q], fr(j + PFD, P0 + q));
    block_barrier_raw();
  };
  static_assert(4 % PFD == 0, "Geo::NI1 is rounded to a multiple of four iterations");
  for (int it0 = 0; it0 < geo.NI1; it0 += PFD) {
    static_for<0, PFD>([&](auto R) { body(R, it0 + decltype(R)::value); });
  }
  bool bad = !(zb >= EMIS_MIN) || !(acc - acc == 0.0);
#pragma unroll
  for (int j = 0; j < NL; ++j) bad = bad || (S.valid[j] && !(zmin[j] >= EMIS_MIN));
  if (nb > 0 && __builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0) atomicOr(&lds.flag, 2);
  if (lane == 0) lds.l2s[wave] = acc;
}

// ------------------------------------------------------------------------------------------------
// main chain
// ------------------------------------------------------------------------------------------------
template <int KIND, int NL, int DIR>
__device__ __forceinline__ void run_main(const Problem &p, float *__restrict__ rows_ws, int *__restrict__ kexp_ws, int nslot,
                                         float *__restrict__ loss, int *__restrict__ flag_ws, Lds<KIND, NL> &lds, const Geo &geo, int b) {
  using C = Cfg<NL>;
  constexpr int UP = C::UP, RS = C::RS;
  Chain<KIND, NL, DIR> S;
  const int lane = threadIdx.x & 63;
  int ll = p.label_length[b] < 0 ? 0 : p.label_length[b];
  const bool shape_ok = (ll <= p.U) && (ll <= UP);
  if (!shape_ok) ll = 0;
  float *own_rows = rows_ws + ((long)b * 2 + DIR) * nslot * RS;
  const float *oth_rows = rows_ws + ((long)b * 2 + (1 - DIR)) * nslot * RS;
  int *own_k = kexp_ws + ((long)b * 2 + DIR) * nslot * 64;
  const int *oth_k = kexp_ws + ((long)b * 2 + (1 - DIR)) * nslot * 64;
  S.init_labels(p, b, lane, ll);
  S.start(lane, ll);
  auto spill = [&](int slot) __attribute__((always_inline)) {
    spill_row<KIND, NL, DIR>(own_rows + (long)slot * RS, lane, S);
    own_k[slot * 64 + lane] = S.k;
  };
  // ================= phase 1 =================
  {
    const int nb = geo.nblocks(1, DIR);
    for (int it = 0; it < geo.NI1; ++it) {
      const int j = it - 1;
      if (j >= 0 && j < nb) {
        const int g = geo.absblock(1, DIR, j);
        const int nv = geo.nvof(g);
        const float(*E)[C::ES] = lds.E[DIR][j % 3];
        if (!(lds.mode & 8)) spill(geo.slot(DIR == 0 ? BLK * g : BLK * g + nv));
        if (nv == BLK) {
          // the emission rows of the whole block go to registers first: the sequential chain never waits for an LDS round trip
          Emis<NL> eb[BLK];
          static_for<0, BLK>([&](auto D) { read_E<NL>(E[decltype(D)::value], lane, eb[decltype(D)::value]); });
          if (!(lds.mode & 8))
          static_for<0, BLK>([&](auto D) {
            constexpr int d = decltype(D)::value;
            S.step(eb[d]);
            if ((d + 1) % RN == 0) S.renorm();
          });
        } else {
          for (int d = 0; d < nv; ++d) {
            Emis<NL> e;
            read_E<NL>(E[d], lane, e);
            S.step(e);
            if ((d + 1) % RN == 0 || d == nv - 1) S.renorm();
          }
        }
      }
      block_barrier_raw();
    }
  }
  spill(geo.slot(geo.tm));
  {
    const int f = S.flag_or();
    if (f != 0 && lane == 0) atomicOr(&lds.flag, f);
  }
  // ================= meeting point: P and dP =================
  __syncthreads();
  if constexpr (DIR == 0) {
    CkRow<NL> ck;
    load_ck<NL>(ck, oth_rows, oth_k, geo.slot(geo.tm), lane);
    const RRow<NL> &r = ck.r;
    const int kn = from_next_lane_i(ck.k, r.kx);
    const float cn = from_next_lane(r.c[0], r.cx), dcn = from_next_lane(r.dc[0], r.dcx);
    float t1 = 0.f, d1 = 0.f;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      if (j < NL - 1) { t1 += S.c[j] * r.c[j + 1]; d1 += fmaf(S.dc[j], r.c[j + 1], S.c[j] * r.dc[j + 1]); }
      if constexpr (KIND == 0) { t1 += S.o[j] * r.o[j]; d1 += fmaf(S.dob[j], r.o[j], S.o[j] * r.dob[j]); }
    }
    const float t2 = S.c[NL - 1] * cn, d2 = fmaf(S.dc[NL - 1], cn, S.c[NL - 1] * dcn);
    const float r0 = readlane_f(r.c[0], 0), dr0 = readlane_f(r.dc[0], 0);
    const float t0 = (lane == 0) ? S.cx * r0 : 0.f, d0 = (lane == 0) ? fmaf(S.dcx, r0, S.cx * dr0) : 0.f;
    const int k0 = readlane_i(ck.k, 0);
    const int e1 = (t1 > 0.f) ? frexp_e(t1) + S.k + ck.k : DEAD;
    const int e2 = (t2 > 0.f) ? frexp_e(t2) + S.k + kn : DEAD;
    const int e0 = (t0 > 0.f) ? frexp_e(t0) + S.kx + k0 : DEAD;
    const int EX = (int)wave_max_dpp((float)imax(e1, imax(e2, e0)));
    float s = 0.f, ds = 0.f;
    if (t1 > 0.f) { s += ldexp_f(t1, S.k + ck.k - EX); ds += ldexp_f(d1, S.k + ck.k - EX); }
    if (t2 > 0.f) { s += ldexp_f(t2, S.k + kn - EX); ds += ldexp_f(d2, S.k + kn - EX); }
    if (t0 > 0.f) { s += ldexp_f(t0, S.kx + k0 - EX); ds += ldexp_f(d0, S.kx + k0 - EX); }
    float sd[2] = {s, ds};
    const float both = swap_reduce<2, false>(sd);
    s = readlane_f(both, SwapLanes<2>::lane(0));
    ds = readlane_f(both, SwapLanes<2>::lane(1));
    const bool okP = shape_ok && EX > DEAD / 2 && s > 0.f && s < 3.0e38f && (ds - ds == 0.f);
    double sl2 = 0.0;
    for (int w = 2; w < NW; ++w) sl2 += lds.l2s[w];
    const int fl = (lds.flag & 3) | (okP ? 0 : 1);
    if (lane == 0) {
      const double dlogp = (double)flog2(s) + (double)EX - sl2;
      if (fl == 0) loss[b] = (float)(-dlogp * LN2_D);  // (flagged utterances: the log-domain pipeline writes theirs)
      const int fe = frexp_e(s);
      lds.lp_int = EX + fe;
      lds.cf = __builtin_amdgcn_rcpf(ldexp_f(s, -fe));
      lds.dlp = ds * __builtin_amdgcn_rcpf(s);  // d log P (natural-log units, blank gauge)
      lds.feasible = (fl == 0) && !(lds.mode & 1);
      lds.flag = fl;
    }
  }
  __syncthreads();
  if (lds.feasible == 0) {
    if (DIR == 0 && lane == 0) flag_ws[b] = lds.mode ? 0 : lds.flag;  // (timing modes: nothing for the fallback to do)
    return;
  }
  const bool idle = (lds.mode & 2) != 0;
  const int lp_int = lds.lp_int;
  const float cf30 = ldexp_f(lds.cf, 30);
  const float dlp = lds.dlp;

  // ================= phase 2 =================
  {
    const int nb = geo.nblocks(2, DIR);
    int kflag = 0;
    for (int it = 0; it < geo.NI2; ++it) {
      const int j = it - 2;
      if (j >= 0 && j < nb && !idle) {
        const int g = geo.absblock(2, DIR, j);
        const int nv = geo.nvof(g);
        const float(*E)[C::ES] = lds.E[DIR][j % 3];
        float(*RR)[RS] = lds.R[DIR][j % 3];
        const int(*KG)[64] = lds.kg[DIR][j % 3];
        float(*KLr)[64] = lds.kl[DIR][j % 3];
        auto grp = [&](int d) -> int {
          const int s = (KIND == 0 && DIR == 1) ? nv - d : nv - 1 - d;
          return (s > 0 ? s - 1 : 0) / RN;
        };
        int q = -1, kR = DEAD, ks = DEAD, k0r = DEAD;
        float KL = 0.f, KS = 0.f, K0 = 0.f;
        auto setK = [&]() __attribute__((always_inline)) {
          const int ka = S.k + kR - lp_int, kb = S.k + ks - lp_int, kc = S.kx + k0r - lp_int;
          kflag |= (imax(ka, imax(kb, kc)) > KK_MAX);  // D5
          KL = ldexp_f(cf30, imin(ka, KK_MAX));
          KS = ldexp_f(cf30, imin(kb, KK_MAX));
          K0 = ldexp_f(cf30, imin(kc, KK_MAX));
        };
        auto one = [&](int d, int qd, bool ren, const Emis<NL> &e, const RRow<NL> &r, int kRq) __attribute__((always_inline)) {
          if (qd != q) {
            q = qd; kR = kRq;
            ks = (DIR == 0) ? from_next_lane_i(kR, r.kx) : from_prev_lane_i(kR, r.kx);
            k0r = readlane_i(kR, DIR == 0 ? 0 : 63);
            setK();
          }
          // the other direction's value one label position over (and its tangent), and its state at this chain's boundary
          const float rs = (DIR == 0) ? from_next_lane(r.c[0], r.cx) : from_prev_lane(r.c[NL - 1], r.cx);
          const float drs = (DIR == 0) ? from_next_lane(r.dc[0], r.dcx) : from_prev_lane(r.dc[NL - 1], r.dcx);
          const float r0 = (DIR == 0) ? readlane_f(r.c[0], 0) : readlane_f(r.c[NL - 1], 63);
          const float dr0 = (DIR == 0) ? readlane_f(r.dc[0], 0) : readlane_f(r.dc[NL - 1], 63);
          // q = a b (raw mantissa product), dq = da b + a db - q dlp
          auto dprod = [&](float a, float da, float bb, float db, float qv) -> float { return fmaf(da, bb, fmaf(a, db, -qv * dlp)); };
          float qal = 0.f, dqal = 0.f, tok[NL], dtok[NL], qsh, dqsh, p0, dp0;
          if constexpr (KIND == 0) {
            if constexpr (DIR == 0) S.step(e);
#pragma unroll
            for (int jj = 0; jj < NL; ++jj) { tok[jj] = S.o[jj] * r.o[jj]; dtok[jj] = dprod(S.o[jj], S.dob[jj], r.o[jj], r.dob[jj], tok[jj]); }
            if constexpr (DIR == 0) {
#pragma unroll
              for (int jj = 0; jj < NL - 1; ++jj) { const float t = S.c[jj] * r.c[jj + 1]; qal += t; dqal += dprod(S.c[jj], S.dc[jj], r.c[jj + 1], r.dc[jj + 1], t); }
              const float t = S.c[NL - 1] * rs;
              qsh = t * KS; dqsh = dprod(S.c[NL - 1], S.dc[NL - 1], rs, drs, t) * KS;
            } else {
#pragma unroll
              for (int jj = 1; jj < NL; ++jj) { const float t = S.c[jj] * r.c[jj - 1]; qal += t; dqal += dprod(S.c[jj], S.dc[jj], r.c[jj - 1], r.dc[jj - 1], t); }
              const float t = S.c[0] * rs;
              qsh = t * KS; dqsh = dprod(S.c[0], S.dc[0], rs, drs, t) * KS;
            }
            p0 = S.cx * r0; dp0 = dprod(S.cx, S.dcx, r0, dr0, p0);
          } else if constexpr (DIR == 0) {
            // simplified: the posterior of a token at frame t is a[t, l=i] y b[t+1, l=i+1]; of the blank a[t, l] bl b[t+1, l]
            const float pin0 = ldexp_f(from_prev_lane(S.c[NL - 1], S.cx), S.dk), dpin0 = ldexp_f(from_prev_lane(S.dc[NL - 1], S.dcx), S.dk);
#pragma unroll
            for (int jj = 0; jj < NL; ++jj) {
              const float pin = (jj == 0) ? pin0 : S.c[jj - 1], dpin = (jj == 0) ? dpin0 : S.dc[jj - 1];
              const float rn = (jj < NL - 1) ? r.c[(jj + 1) % NL] : rs, drn = (jj < NL - 1) ? r.dc[(jj + 1) % NL] : drs;
              const float t = (pin * e.y[jj]) * rn;
              tok[jj] = t;
              dtok[jj] = fmaf(e.y[jj], fmaf(dpin, rn, pin * drn), t * (e.w[jj] - dlp));
              if (jj < NL - 1) { const float u = S.c[jj] * rn; qal += u; dqal += dprod(S.c[jj], S.dc[jj], rn, drn, u); }
            }
            qal *= e.bl; dqal *= e.bl;
            const float u = S.c[NL - 1] * rs;
            qsh = (u * e.bl) * KS; dqsh = (dprod(S.c[NL - 1], S.dc[NL - 1], rs, drs, u) * e.bl) * KS;
            tok[NL - 1] *= KS; dtok[NL - 1] *= KS;
            p0 = S.cx * e.bl * r0; dp0 = dprod(S.cx, S.dcx, r0, dr0, S.cx * r0) * e.bl;
          } else {
            const float nin = ldexp_f(from_next_lane(S.c[0], S.cx), S.dk), dnin = ldexp_f(from_next_lane(S.dc[0], S.dcx), S.dk);
#pragma unroll
            for (int jj = 0; jj < NL; ++jj) {
              const float nx = (jj == NL - 1) ? nin : S.c[(jj + 1) % NL], dnx = (jj == NL - 1) ? dnin : S.dc[(jj + 1) % NL];
              const float rp = (jj > 0) ? r.c[(jj + NL - 1) % NL] : rs, drp = (jj > 0) ? r.dc[(jj + NL - 1) % NL] : drs;
              const float t = (rp * e.y[jj]) * nx;
              tok[jj] = t;
              dtok[jj] = fmaf(e.y[jj], fmaf(drp, nx, rp * dnx), t * (e.w[jj] - dlp));
              if (jj > 0) { const float u = S.c[jj] * rp; qal += u; dqal += dprod(S.c[jj], S.dc[jj], rp, drp, u); }
            }
            qal *= e.bl; dqal *= e.bl;
            const float u = S.c[0] * rs;
            qsh = (u * e.bl) * KS; dqsh = (dprod(S.c[0], S.dc[0], rs, drs, u) * e.bl) * KS;
            tok[0] *= KS; dtok[0] *= KS;
            p0 = S.cx * e.bl * r0; dp0 = dprod(S.cx, S.dcx, r0, dr0, S.cx * r0) * e.bl;
          }
          if (lane == 0) { qsh += p0 * K0; dqsh += dp0 * K0; }
          // S row in place of the R row: [qal, tok.., qsh | dqal, dtok.., dqsh]
          float *srow = RR[d] + 4 * lane * NL;
          if constexpr (NL == 1) *reinterpret_cast<float4 *>(srow) = make_float4(tok[0], qsh, dtok[0], dqsh);
          else {
            *reinterpret_cast<float4 *>(srow) = make_float4(qal, tok[0], tok[1], qsh);
            *reinterpret_cast<float4 *>(srow + 4) = make_float4(dqal, dtok[0], dtok[1], dqsh);
          }
          KLr[d][lane] = KL;
          if constexpr (!(KIND == 0 && DIR == 0)) S.step(e);
          if (ren) { S.renorm(); setK(); }
        };
        if (nv == BLK) {
          // emission rows of the whole block and the exponent groups up front, R rows PR frames ahead of their use
          constexpr int PR = 2;
          Emis<NL> eb[BLK];
          RRow<NL> rb[BLK];
          int kq[NG];
          static_for<0, NG>([&](auto Q) { kq[decltype(Q)::value] = KG[decltype(Q)::value][lane]; });
          static_for<0, PR>([&](auto D) { read_R<NL>(RR[decltype(D)::value], lane, rb[decltype(D)::value]); });
          static_for<0, BLK>([&](auto D) { read_E<NL>(E[decltype(D)::value], lane, eb[decltype(D)::value]); });
          static_for<0, BLK>([&](auto D) {
            constexpr int d = decltype(D)::value;
            if constexpr (d + PR < BLK) read_R<NL>(RR[d + PR], lane, rb[d + PR]);
            constexpr int sst = (KIND == 0 && DIR == 1) ? BLK - d : BLK - 1 - d;
            constexpr int qd = (sst > 0 ? sst - 1 : 0) / RN;
            one(d, qd, (d + 1) % RN == 0, eb[d], rb[d], kq[qd]);
          });
        } else {
          for (int d = 0; d < nv; ++d) {
            Emis<NL> e;
            read_E<NL>(E[d], lane, e);
            RRow<NL> r;
            read_R<NL>(RR[d], lane, r);
            const int qd = grp(d);
            one(d, qd, (d + 1) % RN == 0 || d == nv - 1, e, r, KG[qd][lane]);
          }
        }
      }
      block_barrier_raw();
    }
    if (__builtin_amdgcn_ballot_w64(kflag != 0) != 0 && lane == 0) atomicOr(&lds.flag, 32);  // D5
  }
  __syncthreads();
  if (DIR == 0 && lane == 0) flag_ws[b] = lds.mode ? 0 : lds.flag;
}

// ------------------------------------------------------------------------------------------------
// recompute chain of side SIDE (phase 2): the OTHER direction's recursion inside one block, from that direction's checkpoint
// ------------------------------------------------------------------------------------------------
template <int KIND, int NL, int SIDE>
__device__ __forceinline__ void run_recompute(const Problem &p, const float *__restrict__ rows_ws, const int *__restrict__ kexp_ws, int nslot,
                                              float4 *__restrict__ stats_ws, const float *vec, Lds<KIND, NL> &lds, const Geo &geo, int b) {
  constexpr int RDIR = 1 - SIDE;
  using C = Cfg<NL>;
  constexpr int RS = C::RS;
  const int lane = threadIdx.x & 63;
  int ll = p.label_length[b] < 0 ? 0 : p.label_length[b];
  if (ll > p.U || ll > C::UP) ll = 0;
  const float *ck_rows = rows_ws + ((long)b * 2 + RDIR) * nslot * RS;
  const int *ck_k = kexp_ws + ((long)b * 2 + RDIR) * nslot * 64;
  float *dump = lds.dump[2 + SIDE];
  {  // phase 1: E-stage position 5 of every block of this side
    Rows<KIND, NL> W;
    W.init(p, b, lane, ll, vec, nullptr);
    W.xs = lds.xcopy[2 * NH + SIDE];
    W.vs = lds.vcopy[2 * NH + SIDE];
    if (lane == 0) { W.xs[V] = 0.f; W.vs[V] = 0.f; }
    estage1<KIND, NL, SIDE, BLK - 1, 1>(W, lds, geo, stats_ws + (long)b * p.T, dump, lane, 2 + SIDE);
    __syncthreads();
    __syncthreads();
  }
  if (lds.feasible == 0) return;
  const bool idle = (lds.mode & 2) != 0;
  Chain<KIND, NL, RDIR> S;
  S.init_labels(p, b, lane, ll);
  const int nb = geo.nblocks(2, SIDE);
  auto ck_slot = [&](int j) -> int {
    int jj = j < nb ? j : nb - 1;
    jj = jj < 0 ? 0 : jj;
    const int g = geo.absblock(2, SIDE, jj);
    const int t = (SIDE == 0) ? BLK * g + geo.nvof(g) : BLK * g;
    return geo.slot(t < 0 ? 0 : t);
  };
  CkRow<NL> ck_next;
  load_ck<NL>(ck_next, ck_rows, ck_k, ck_slot(0), lane);
  for (int it = 0; it < geo.NI2; ++it) {
    const int j = it - 1;
    if (j >= 0 && j < nb && !idle) {
      const int g = geo.absblock(2, SIDE, j);
      const int nv = geo.nvof(g);
      const float(*E)[C::ES] = lds.E[SIDE][j % 3];
      float(*RR)[RS] = lds.R[SIDE][j % 3];
      int(*KG)[64] = lds.kg[SIDE][j % 3];
      const CkRow<NL> ck = ck_next;
      load_ck<NL>(ck_next, ck_rows, ck_k, ck_slot(j + 1), lane);
      restore<KIND, NL, RDIR>(S, ck);
      KG[0][lane] = S.k;
      int s = 0;
      auto put = [&](int d) __attribute__((always_inline)) { put_row<KIND, NL, RDIR>(RR[d], dump + (lane & 15) * 4, lane, S); };
      auto stp = [&](int d) __attribute__((always_inline)) { Emis<NL> e; read_E<NL>(E[d], lane, e); S.step(e); };
      auto after = [&](bool more) __attribute__((always_inline)) {
        ++s;
        if (s % RN == 0 && more) { S.renorm(); KG[s / RN][lane] = S.k; }
      };
      Emis<NL> eb[BLK];  // full blocks: the emission rows go to registers before the chain starts
      if (nv == BLK) static_for<0, BLK>([&](auto D) { read_E<NL>(E[decltype(D)::value], lane, eb[decltype(D)::value]); });
      auto stpb = [&](auto D) __attribute__((always_inline)) { S.step(eb[decltype(D)::value]); };
      if constexpr (SIDE == 0) {
        put(nv - 1);
        if (nv == BLK) {
          static_for<0, BLK - 1>([&](auto I) {
            constexpr int d = BLK - 1 - decltype(I)::value;
            stpb(std::integral_constant<int, d>{}); put(d - 1); after(d > 1);
          });
        } else {
          for (int d = nv - 1; d >= 1; --d) { stp(d); put(d - 1); after(d > 1); }
        }
      } else if constexpr (KIND == 0) {
        if (nv == BLK) {
          static_for<0, BLK>([&](auto I) {
            constexpr int i = decltype(I)::value;
            stpb(std::integral_constant<int, BLK - 1 - i>{}); put(BLK - 1 - i); after(i < BLK - 1);
          });
        } else {
          for (int i = 0; i < nv; ++i) { stp(nv - 1 - i); put(nv - 1 - i); after(i < nv - 1); }
        }
      } else {
        put(nv - 1);
        if (nv == BLK) {
          static_for<1, BLK>([&](auto I) {
            constexpr int i = decltype(I)::value;
            stpb(std::integral_constant<int, BLK - i>{}); put(BLK - 1 - i); after(i < BLK - 1);
          });
        } else {
          for (int i = 1; i < nv; ++i) { stp(nv - i); put(nv - 1 - i); after(i < nv - 1); }
        }
      }
    }
    block_barrier_raw();
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// helper wavefront `slot` of side DIR, three per side, two positions of every block each in phase 2 (P2, P2 + 1).  The G stage
// is a long dependent sequence per frame (S row -> three reductions -> bins -> atomics -> bins -> row), so the helpers are bound
// by frames per wavefront, not by issue slots: two helpers a side with three frames each took 2.4 us per block, an uneven 2 / 4
// split (light helper beside the main chain) 2.6.  Phase 1: positions P1 .. P1 + N1 - 1 (2, 2, 1; the recompute wavefront,
// idle as a chain then, takes the sixth).
// ------------------------------------------------------------------------------------------------
template <int KIND, int NL, int DIR, int P1, int N1, int P2, int N2>
__device__ __forceinline__ void run_helper(const Problem &p, float4 *__restrict__ stats_ws, const float *__restrict__ vec, float *__restrict__ out,
                                           Lds<KIND, NL> &lds, const Geo &geo, int slot, int b) {
  using C = Cfg<NL>;
  static_assert(N2 >= 1 && N2 <= 4, "at most four positions per block (one four-value reduction)");
  Rows<KIND, NL> S;
  const int lane = threadIdx.x & 63;
  const int len = geo.len;
  int ll = p.label_length[b] < 0 ? 0 : p.label_length[b];
  if (ll > p.U || ll > C::UP) ll = 0;
  S.init(p, b, lane, ll, vec, out);
  S.xs = lds.xcopy[DIR * NH + slot];
  S.vs = lds.vcopy[DIR * NH + slot];
  S.bins = lds.bins[DIR * NH + slot];
  if (lane == 0) { S.xs[V] = 0.f; S.vs[V] = 0.f; }
  float4 *stats = stats_ws + (long)b * p.T;
  float *dump = lds.dump[4 + DIR * NH + slot];
  const int wave = 4 + DIR * NH + slot;
  // ================= phase 1 =================
  estage1<KIND, NL, DIR, P1, N1>(S, lds, geo, stats, dump, lane, wave);
  __syncthreads();
  __syncthreads();
  if (lds.feasible == 0) return;
  // ================= phase 2: E stage of block it, G stage of block it-3 =================
  const int nb = geo.nblocks(2, DIR);
  if (slot == 0 && DIR == 0) S.zero_rows(len, p.T);  // frames beyond logit_length: the Hessian vanishes there (base_loss.py:240-258)
  auto fr = [&](int j, int d) -> int {
    int jj = j < nb ? j : nb - 1;
    jj = jj < 0 ? 0 : jj;
    const int g = geo.absblock(2, DIR, jj);
    const int nv = geo.nvof(g);
    int dd = d < nv ? d : nv - 1;
    int t = geo.frame(DIR, g, dd < 0 ? 0 : dd);
    t = t < len ? t : len - 1;
    return t < 0 ? 0 : t;
  };
  // Register rings addressed at compile time (the loop is unrolled by 3).  X / Vv / SX: the rows of block it + 3, loaded while
  // block it is worked on (an HBM load under load takes ~2 us, a block ~1 us), with the frame's (rowmax, 1 / sum, s.v / sum).
  // Z: the softmax part of the output row of block it, s_t[k] (v_t[k] - s_t . v_t), computed by its E stage and kept for its G
  // stage three iterations later (out = Z - d posterior) -- r03 read logits and vector rows again there (mostly past the L2: 0.5 GB
  // of the call's 2.0 GB of fabric traffic) and exponentiated again.
  constexpr int PF2 = 3;
  float4 X[PF2][N2], Vv[PF2][N2];
  float SX[PF2][N2], SI[PF2][N2], SV[PF2][N2];
  float4 Z[3][N2];
  auto load_st = [&](int r_, int q, int t) __attribute__((always_inline)) {
    const float4 st = stats[t];
    SX[r_][q] = st.x; SI[r_][q] = st.y; SV[r_][q] = st.z;
  };
  static_for<0, PF2>([&](auto R) {
    constexpr int r = decltype(R)::value;
#pragma unroll
    for (int q = 0; q < N2; ++q) {
      X[r][q] = make_float4(0.f, 0.f, 0.f, 0.f); Vv[r][q] = X[r][q]; SX[r][q] = 0.f; SI[r][q] = 0.f; SV[r][q] = 0.f;
      Z[r][q] = X[r][q];
      if (nb > 0) { S.load_xv(X[r][q], Vv[r][q], fr(r, P2 + q)); load_st(r, q, fr(r, P2 + q)); }
    }
  });
  bool massbad = false;
  const bool idle = (lds.mode & 4) != 0;
  auto body = [&](auto R, int it) __attribute__((always_inline)) {
    constexpr int r = decltype(R)::value;  // = it mod 3
    if (idle) { block_barrier_raw(); return; }
    const int gj = it - 3;
    const bool do_g = gj >= 0 && gj < nb;
    // what this iteration's G stage (block it-3) works on: kept by that block's E stage in this very ring slot
    float4 zg[N2];
#pragma unroll
    for (int q = 0; q < N2; ++q) zg[q] = Z[r][q];
    // ---- E stage (block it) ----
    const int j = it;
    if (j < nb) {
      const int g = geo.absblock(2, DIR, j);
      const int nv = geo.nvof(g);
      float(*E)[C::ES] = lds.E[DIR][j % 3];
      float4 xq[N2], vq[N2];
      float sq[N2], si[N2], sv[N2];
#pragma unroll
      for (int q = 0; q < N2; ++q) { xq[q] = X[r][q]; vq[q] = Vv[r][q]; S.mask_xv(xq[q], vq[q]); sq[q] = SX[r][q]; si[q] = SI[r][q]; sv[q] = SV[r][q]; }
#pragma unroll
      for (int q = 0; q < N2; ++q) {
        const int d = P2 + q;
        if (d < nv) {
          const float4 ev = S.expo(xq[q], sq[q]);
          Emis<NL> e;
          S.gather(ev, vq[q], e);
          write_E<NL>(E[d], dump, lane, e);
          Z[r][q] = make_float4(ev.x * si[q] * (vq[q].x - sv[q]), ev.y * si[q] * (vq[q].y - sv[q]), ev.z * si[q] * (vq[q].z - sv[q]),
                                ev.w * si[q] * (vq[q].w - sv[q]));
        }
      }
    }
    // (refill after use and on every path, as in phase 1)
#pragma unroll
    for (int q = 0; q < N2; ++q) { S.load_xv(X[r][q], Vv[r][q], fr(j + PF2, P2 + q)); load_st(r, q, fr(j + PF2, P2 + q)); }
    // ---- G stage (block it-3) ----
    if (do_g) {
      const int g = geo.absblock(2, DIR, gj);
      const int nv = geo.nvof(g);
      const float(*SR)[C::RS] = lds.R[DIR][gj % 3];
      const float(*KLr)[64] = lds.kl[DIR][gj % 3];
      constexpr int JS = (KIND == 1) ? (DIR == 0 ? NL - 1 : 0) : -1;  // simplified: the slot whose token part is already scaled
      float dqt[N2][NL];
      float qm[4] = {0.f, 0.f, 0.f, 0.f}, db[4] = {0.f, 0.f, 0.f, 0.f}, as[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < N2; ++q) {
        const int d = P2 + q;
        const int dd = d < nv ? d : nv - 1;
        const float *srow = SR[dd] + 4 * lane * NL;
        const float kl = KLr[dd][lane];
        float qal = 0.f, qsh, dqal = 0.f, dqsh, qt[NL];
        if constexpr (NL == 1) {
          const float4 t = *reinterpret_cast<const float4 *>(srow);
          qt[0] = t.x; qsh = t.y; dqt[q][0] = t.z; dqsh = t.w;
        } else {
          const float4 t = *reinterpret_cast<const float4 *>(srow), u = *reinterpret_cast<const float4 *>(srow + 4);
          qal = t.x; qt[0] = t.y; qt[1] = t.z; qsh = t.w;
          dqal = u.x; dqt[q][0] = u.y; dqt[q][1] = u.z; dqsh = u.w;
        }
        float mass = qal * kl + qsh, a = 0.f;
#pragma unroll
        for (int jj = 0; jj < NL; ++jj) {
          if (jj != JS) { qt[jj] *= kl; dqt[q][jj] *= kl; }
          mass += qt[jj];
          a += fabsf(dqt[q][jj]);
        }
        qm[q] = mass; db[q] = dqal * kl + dqsh; as[q] = a;
      }
      const float qall = swap_reduce<4, false>(qm), dball = swap_reduce<4, false>(db), aall = swap_reduce<4, false>(as);
#pragma unroll
      for (int q = 0; q < N2; ++q) {
        const int d = P2 + q;
        if (d < nv) {
          massbad |= !(fabsf(readlane_f(qall, SwapLanes<4>::lane(q)) - 1073741824.0f) < 1073741824.0f * MASS_TOL);  // D6
          S.out_row(geo.frame(DIR, g, d), readlane_f(dball, SwapLanes<4>::lane(q)), dqt[q], readlane_f(aall, SwapLanes<4>::lane(q)), zg[q]);
        }
      }
    }
    block_barrier_raw();
  };
  for (int it0 = 0; it0 < geo.NI2; it0 += 3) {
    static_for<0, 3>([&](auto R) { body(R, it0 + decltype(R)::value); });
  }
  if (massbad && lane == 0) atomicOr(&lds.flag, 64);  // D6
  __syncthreads();
}

// The log-domain pipeline for ONE utterance by the workgroup that flagged it (all ten wavefronts; ctc_hvp.hip for the stages).
template <int KIND, int NL>
__device__ __forceinline__ void redo_log_domain(const Problem &p, const Layout &L, char *ws, const float *__restrict__ vec,
                                                float *__restrict__ loss, float *__restrict__ out, float *lds_f, int w, int b) {
  const int lane = threadIdx.x & 63;
  const HvpLayout H = make_hvp_layout(L, p.B, p.T);
  float *emis = reinterpret_cast<float *>(ws + L.off_emis);
  float *alpha = reinterpret_cast<float *>(ws + L.off_alpha);
  float *beta = reinterpret_cast<float *>(ws + L.off_beta);
  double *logp = reinterpret_cast<double *>(ws + L.off_logp);
  char *ex = ws + L.off_extra;
  float *demis = reinterpret_cast<float *>(ex + H.off_demis);
  float *dalpha = reinterpret_cast<float *>(ex + H.off_dalpha);
  float *dbeta = reinterpret_cast<float *>(ex + H.off_dbeta);
  float *dlogp = reinterpret_cast<float *>(ex + H.off_dlogp);
  // (every __syncthreads below drains the wavefront's stores first: rows written by one wavefront are read by another of the
  // same CU through the shared L1, as at the meeting point of the chains)
  for (int t = w; t < p.T; t += NW) emit_row(p, L, emis, b, t, lane);
  __syncthreads();
  if (w == 0) scan_body<KIND, NL, 0, true>(p, L, emis, alpha, logp, loss, b, lane);   // (rows renormalised every step: the tangent sweep reads them)
  else if (w == 1) scan_body<KIND, NL, 1, true>(p, L, emis, beta, logp, loss, b, lane);
  else for (int t = w - 2; t < p.T; t += NW - 2) temit_row(p, L, emis, vec, demis, b, t, lane);  // (beside the value sweeps)
  __syncthreads();
  if (w < 2) tscan_body<KIND, NL>(p, L, emis, demis, alpha, beta, logp, dalpha, dbeta, dlogp, b, w, lane);
  __syncthreads();
  float *bin = lds_f + w * (V + 4);
  for (int t = w; t < p.T; t += NW) {
    hvp_out_row<KIND>(p, L, emis, demis, alpha, beta, dalpha, dbeta, logp, dlogp, vec, out, bin, b, t, lane);
    wave_lds_fence();  // the bins are reused by the next frame of this wavefront
  }
}

template <int KIND, int NL>
__global__ __launch_bounds__(64 * NW) void hvp_fused_kernel(Problem p, Layout L, char *__restrict__ ws_v1, float *__restrict__ rows_ws, int *__restrict__ kexp_ws, int nslot,
                                                             float4 *__restrict__ stats_ws, float *__restrict__ loss,
                                                             const float *__restrict__ vec, float *__restrict__ out,
                                                             int *__restrict__ flag_ws, int mode) {
  // mode (timing diagnostics through ctc_amd_debug_override("hvp", "diag<mode>"), 0 in every product call): 1 = stop at the meeting
  // point, 2 = the chains keep only their barriers in phase 2, 4 = the helpers do, 8 = the main chains in phase 1, 16 = the E stage
  // of phase 1 keeps its loads and barriers only
  __shared__ __attribute__((aligned(16))) Lds<KIND, NL> lds;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x;
  Geo geo;
  geo.init(clampi(p.logit_length[b], 0, p.T));
  if (threadIdx.x == 0) { lds.flag = 0; lds.feasible = 0; lds.mode = mode; }
  if (threadIdx.x < NW) lds.l2s[threadIdx.x] = 0.0;
  __syncthreads();
  // Wavefronts w, w + 4, w + 8 share a SIMD (0 and 1 hold three, 2 and 3 two): the main chains -- the busiest wavefronts -- sit on
  // the two-wavefront SIMDs with one helper each, the recompute chains with two helpers each.
  // (CTC_HVPF_PROBE_ROLE, diagnostic builds: compile ONE role only, to read its register need off -Rpass-analysis=kernel-resource-usage)
#ifndef CTC_HVPF_PROBE_ROLE
#define CTC_HVPF_PROBE_ROLE -1
#endif
  constexpr int PR = CTC_HVPF_PROBE_ROLE;
  auto is = [&](int role) { return (PR < 0 || PR == role) && w == role; };
  if (is(2)) {
    __builtin_amdgcn_s_setprio(3);
    run_main<KIND, NL, 0>(p, rows_ws, kexp_ws, nslot, loss, flag_ws, lds, geo, b);
  } else if (is(3)) {
    __builtin_amdgcn_s_setprio(3);
    run_main<KIND, NL, 1>(p, rows_ws, kexp_ws, nslot, loss, flag_ws, lds, geo, b);
  } else if (is(0)) {
    __builtin_amdgcn_s_setprio(2);
    run_recompute<KIND, NL, 0>(p, rows_ws, kexp_ws, nslot, stats_ws, vec, lds, geo, b);
  } else if (is(1)) {
    __builtin_amdgcn_s_setprio(2);
    run_recompute<KIND, NL, 1>(p, rows_ws, kexp_ws, nslot, stats_ws, vec, lds, geo, b);
  } else if (is(4)) {
    run_helper<KIND, NL, 0, 0, 2, 0, 2>(p, stats_ws, vec, out, lds, geo, 0, b);
  } else if (is(5)) {
    run_helper<KIND, NL, 1, 0, 2, 0, 2>(p, stats_ws, vec, out, lds, geo, 0, b);
  } else if (is(6)) {
    run_helper<KIND, NL, 0, 2, 2, 2, 2>(p, stats_ws, vec, out, lds, geo, 1, b);
  } else if (is(7)) {
    run_helper<KIND, NL, 1, 2, 2, 2, 2>(p, stats_ws, vec, out, lds, geo, 1, b);
  } else if (is(8)) {
    run_helper<KIND, NL, 0, 4, 1, 4, 2>(p, stats_ws, vec, out, lds, geo, 2, b);
  } else if (is(9)) {
    run_helper<KIND, NL, 1, 4, 1, 4, 2>(p, stats_ws, vec, out, lds, geo, 2, b);
  }
  // utterances the linear domain cannot hold (normally none): redone right here in the log domain, every output row rewritten
  __syncthreads();
  const int fl = lds.mode ? 0 : lds.flag;
  __syncthreads();  // (the LDS is reused from here on)
#ifndef CTC_HVPF_NO_REDO
  if (PR < 0 && fl != 0) redo_log_domain<KIND, NL>(p, L, ws_v1, vec, loss, out, reinterpret_cast<float *>(&lds), w, b);
#endif
}

}  // namespace hvpf

#if CTC_FUSED_KIND == 0
#define CTC_HVPF_ENTRY run_hvp_fused_classic
#else
#define CTC_HVPF_ENTRY run_hvp_fused_simplified
#endif
// ws: the CTC_AMD_WS_HVP workspace (L = its full-row layout: the regions of the log-domain building blocks; the fused kernel's
// own region sits behind them).  flags[b] != 0 afterwards = utterance b was redone in the log domain (diagnostic).
hipError_t CTC_HVPF_ENTRY(const Problem &p, const Layout &L, char *ws, const float *vec, float *loss, float *out, int mode, hipStream_t st) {
  const HvpFusedLayout H = make_hvp_fused_layout(p.B, p.T, p.U);
  char *ws_fused = ws + L.off_extra + make_hvp_layout(L, p.B, p.T).total;
  if (L.NL != (p.U <= 64 ? 1 : 2)) return hipErrorInvalidValue;
  float *rows = reinterpret_cast<float *>(ws_fused + H.off_rows);
  int *kexp = reinterpret_cast<int *>(ws_fused + H.off_kexp);
  float4 *stats = reinterpret_cast<float4 *>(ws_fused + H.off_stats);
  int *flags = reinterpret_cast<int *>(ws_fused + H.off_flags);
  static_assert(sizeof(hvpf::Lds<CTC_FUSED_KIND, 2>) <= 160 * 1024, "LDS budget of one CU");
  static_assert(hvpf::V == HVPF_MAX_V && hvpf::Cfg<2>::UP == HVPF_MAX_U, "limits of ctc_hvp_fused.h");
  const dim3 grid(p.B), block(64 * hvpf::NW);
  if (p.U <= 64)
    hipLaunchKernelGGL((hvpf::hvp_fused_kernel<CTC_FUSED_KIND, 1>), grid, block, 0, st, p, L, ws, rows, kexp, H.nslot, stats, loss, vec, out, flags, mode);
  else
    hipLaunchKernelGGL((hvpf::hvp_fused_kernel<CTC_FUSED_KIND, 2>), grid, block, 0, st, p, L, ws, rows, kexp, H.nslot, stats, loss, vec, out, flags, mode);
  return hipGetLastError();
}

}  // namespace ctc

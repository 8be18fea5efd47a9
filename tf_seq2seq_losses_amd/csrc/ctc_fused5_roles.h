// Roles of the log-domain checkpoint + recompute kernel (see ctc_fused5.hip for the description): main chains, recompute
// chains, helpers, and run_roles = one workgroup's work for one utterance.  Included by ctc_fused5.hip (its own kernel)
// and by ctc_fused6.hip (fallback for flagged utterances inside the same launch).
#pragma once
#include "ctc_fused_common.h"

#ifndef CTC_FUSED_KIND
#error "compile with -DCTC_FUSED_KIND=0 (classic) or 1 (simplified)"
#endif

namespace ctc {
namespace fused5 {

using namespace ctc::fused;

__device__ __forceinline__ void block_barrier() {
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's LDS writes have landed; vmcnt untouched
  __builtin_amdgcn_s_barrier();
}

#ifdef CTC_FUSED_STAMPS
struct Stamps {
  unsigned long long work = 0, wait = 0, t0 = 0, work1 = 0, wait1 = 0;
  __device__ __forceinline__ void begin() { t0 = __builtin_amdgcn_s_memtime(); }
  __device__ __forceinline__ void mid() { unsigned long long t = __builtin_amdgcn_s_memtime(); work += t - t0; t0 = t; }
  __device__ __forceinline__ void end() { unsigned long long t = __builtin_amdgcn_s_memtime(); wait += t - t0; t0 = t; }
  __device__ __forceinline__ void phase1_done() { work1 = work; wait1 = wait; }
  __device__ __forceinline__ void dump(unsigned long long *dst, int lane) {
    if (lane == 0) { dst[0] = work; dst[1] = wait; dst[2] = work1; dst[3] = wait1; }
  }
};
#else
struct Stamps {  // normal builds: nothing
  __device__ __forceinline__ void begin() {}
  __device__ __forceinline__ void mid() {}
  __device__ __forceinline__ void end() {}
  __device__ __forceinline__ void phase1_done() {}
  __device__ __forceinline__ void dump(unsigned long long *, int) {}
};
#endif
#define STAMP(x) x

template <int KIND, int NL, int NH, int BLK, int VPL>
struct Lds {
  static constexpr int V = 256 * VPL, UP = 64 * NL;
  static constexpr int ES = UP + 4;      // E row: y[UP], bl, mx, l2s, -
  static constexpr int RS = 2 * UP + 8;  // R row (recompute chain): the other side's lattice row in its HBM layout;
                                         // S row (main chain, in place): (s1, s2) per slot, s0 at [2 UP]
  static constexpr int NW = 4 + 2 * NH;
  float E[2][3][BLK][ES];   // [side][block % 3]
  float R[2][3][BLK][RS];   // [side][block % 3]
  float xcopy[2 * NH][V + 4];
  float xcopy_r[2][V + 4];  // row copies of the recompute waves (E stage of phase 1)
  float bins[2 * NH][V + 4];
  float dump[NW][64];
  int feasible;
};

// Block geometry shared by every wavefront of the workgroup.
template <int BLK>
struct Geo {
  int len, G, tmb, tm, NB;
  __device__ __forceinline__ void init(int len_) {
    len = len_;
    G = (len + BLK - 1) / BLK;
    tmb = G / 2;
    tm = tmb * BLK;
    NB = G - tmb;  // >= tmb: blocks per side and phase, as iteration bound
  }
  __device__ __forceinline__ int nvof(int g) const { int r = len - BLK * g; return r < BLK ? r : BLK; }
  // side-local block j of (phase, side) -> absolute block; count of blocks
  __device__ __forceinline__ int nblocks(int phase, int side) const { return (phase == 1) == (side == 0) ? tmb : G - tmb; }
  __device__ __forceinline__ int absblock(int phase, int side, int j) const {
    if (phase == 1) return side == 0 ? j : G - 1 - j;
    return side == 0 ? tmb + j : tmb - 1 - j;
  }
  // frame processed at position d of block g by `side` (A ascending, B descending)
  __device__ __forceinline__ int frame(int side, int g, int d) const { return side == 0 ? BLK * g + d : BLK * g + nvof(g) - 1 - d; }
};

// NL consecutive floats (or NL consecutive (a, b) pairs) of this lane in an LDS / HBM row, NL = 1, 2, 4: widest accesses
template <int NL>
__device__ __forceinline__ void ld_slots(const float *p, float (&v)[NL]) {
  if constexpr (NL == 1) v[0] = p[0];
  else if constexpr (NL == 2) { const float2 t = *reinterpret_cast<const float2 *>(p); v[0] = t.x; v[1] = t.y; }
  else {
#pragma unroll
    for (int q = 0; q < NL / 4; ++q) {
      const float4 t = *reinterpret_cast<const float4 *>(p + 4 * q);
      v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
    }
  }
}
template <int NL>
__device__ __forceinline__ void st_slots(float *p, const float (&v)[NL]) {
  if constexpr (NL == 1) p[0] = v[0];
  else if constexpr (NL == 2) *reinterpret_cast<float2 *>(p) = make_float2(v[0], v[1]);
  else {
#pragma unroll
    for (int q = 0; q < NL / 4; ++q) *reinterpret_cast<float4 *>(p + 4 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
  }
}
template <int NL>
__device__ __forceinline__ void ld_pairs(const float *p, float (&a)[NL], float (&b)[NL]) {
  if constexpr (NL == 1) { const float2 t = *reinterpret_cast<const float2 *>(p); a[0] = t.x; b[0] = t.y; }
  else {
#pragma unroll
    for (int q = 0; q < NL / 2; ++q) {
      const float4 t = *reinterpret_cast<const float4 *>(p + 4 * q);
      a[2 * q] = t.x; b[2 * q] = t.y; a[2 * q + 1] = t.z; b[2 * q + 1] = t.w;
    }
  }
}
template <int NL>
__device__ __forceinline__ void st_pairs(float *p, const float (&a)[NL], const float (&b)[NL]) {
  if constexpr (NL == 1) *reinterpret_cast<float2 *>(p) = make_float2(a[0], b[0]);
  else {
#pragma unroll
    for (int q = 0; q < NL / 2; ++q) *reinterpret_cast<float4 *>(p + 4 * q) = make_float4(a[2 * q], b[2 * q], a[2 * q + 1], b[2 * q + 1]);
  }
}

template <int NL, class LDt>
__device__ __forceinline__ void read_E(const float *row, int lane, Emis<NL> &e) {
  ld_slots<NL>(row + lane * NL, e.y);
  float4 tl = *reinterpret_cast<const float4 *>(row + LDt::UP);  // same address in every lane: LDS broadcast
  e.bl = tl.x; e.mx = tl.y; e.l2s = tl.z;
}

// lattice row (HBM layout of Layout::SRS floats) <-> LDS row
template <int KIND, int NL, class LDt>
__device__ __forceinline__ void read_R(const float *row, int lane, SRow<KIND, NL> &r) {
  if constexpr (KIND == 0) {
    ld_pairs<NL>(row + 2 * lane * NL, r.a, r.b);
    r.tail = *reinterpret_cast<const float4 *>(row + 2 * LDt::UP);
  } else {
    ld_slots<NL>(row + lane * NL, r.a);
    r.tail = *reinterpret_cast<const float4 *>(row + LDt::UP);
  }
}
template <int KIND, int NL, class LDt>
__device__ __forceinline__ void write_R(float *row, float *dump, int lane, const float (&a)[NL], const float (&b)[NL], float4 tail) {
  if constexpr (KIND == 0) {
    st_pairs<NL>(row + 2 * lane * NL, a, b);
    float *tq = (lane == 0) ? row + 2 * LDt::UP : dump + (lane & 15) * 4;
    *reinterpret_cast<float4 *>(tq) = tail;
  } else {
    st_slots<NL>(row + lane * NL, a);
    float *tq = (lane == 0) ? row + LDt::UP : dump + (lane & 15) * 4;
    *reinterpret_cast<float4 *>(tq) = tail;
  }
}

// state of a chain in the row layout the other side is aligned with (see Side::spill)
template <int KIND, int NL, int DIR, class S_t>
__device__ __forceinline__ void state_row(const S_t &S, float (&cs)[NL], float4 &tail) {
  float tx;
  if constexpr (DIR == 0) {
#pragma unroll
    for (int j = NL - 1; j > 0; --j) cs[j] = (float)S.c[j - 1];
    cs[0] = from_prev_lane((float)S.c[NL - 1], (float)S.cx);
    tx = readlane_f((float)S.c[NL - 1], 63);
  } else {
#pragma unroll
    for (int j = 0; j < NL - 1; ++j) cs[j] = (float)S.c[j + 1];
    cs[NL - 1] = from_next_lane((float)S.c[0], (float)S.cx);
    tx = readlane_f((float)S.c[0], 0);
  }
  const float oh = (float)S.off;
  tail = make_float4(tx, 0.f, oh, (float)(S.off - (double)oh));
}
// inverse: a chain's native state from one of its own checkpoint rows
template <int KIND, int NL, int DIR, class S_t>
__device__ __forceinline__ void restore_state(S_t &S, const SRow<KIND, NL> &r) {
  if constexpr (DIR == 0) {  // row slot i = state_c(l=i), tail = state_c(l=UP); native slot i = state_c(l=i+1), cx = l=0
#pragma unroll
    for (int j = 0; j < NL - 1; ++j) S.c[j] = r.a[j + 1];
    S.c[NL - 1] = from_next_lane(r.a[0], r.tail.x);
    S.cx = readlane_f(r.a[0], 0);
  } else {  // row slot i = state_c(l=i+1), tail = state_c(l=0); native slot i = state_c(l=i), cx = l=UP
#pragma unroll
    for (int j = NL - 1; j > 0; --j) S.c[j] = r.a[j - 1];
    S.c[0] = from_prev_lane(r.a[NL - 1], r.tail.x);
    S.cx = readlane_f(r.a[NL - 1], 63);
  }
#pragma unroll
  for (int j = 0; j < NL; ++j) S.o[j] = (KIND == 0) ? r.b[j] : NEG;
  S.off = (double)r.tail.z + (double)r.tail.w;
}

template <int KIND, int NL, class S_t>
__device__ __forceinline__ void init_labels(S_t &S, const Problem &p, int b, int lane, int ll) {
  const int32_t *lab = p.labels + (long)b * p.label_stride;
  auto tok = [&](int i) -> int { return (i >= 0 && i < ll) ? ((i < p.label_stride) ? lab[i] : p.blank) : -1 - (i < 0); };
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    int i = lane * NL + j;
    int tk = tok(i);
    S.norep[j] = (i == 0) || tk != tok(i - 1);
    S.norep_next[j] = tok(i + 1) != tk;
    S.tokoff[j] = 4 * ((tk >= 0 && tk < p.V && tk < S_t::V && tk != p.blank) ? tk : S_t::V);
    S.c[j] = NEG;
    S.o[j] = NEG;
  }
#pragma unroll
  for (int e = 0; e < 4 * (S_t::V / 256); ++e) S.mb[e] = (256 * (e / 4) + lane * 4 + (e & 3) == p.blank) ? 1.f : 0.f;
}

// ------------------------------------------------------------------------------------------------
// E stage of phase 1 (logits rows -> log-softmax statistics -> emission rows in LDS), shared by the helpers and by the
// recompute wavefronts, which have no lattice work before the meeting point.  Positions of a 12-frame block of side
// SIDE are dealt out by SIMD: the two helpers that share a SIMD with a main chain take X each, the two on the recompute
// SIMDs Y each, the side's recompute wavefront the remaining R = 12 - 2X - 2Y (SIMD 0/1 carry the chains' 2.5 k VALU
// cycles per block, SIMD 2/3 nothing else in this phase).  The worker also records the statistics of its frames for the
// other side's pass over them in phase 2.  One barrier per block, like every other role.
// ------------------------------------------------------------------------------------------------
// (measured: 1/3/4 for two label positions per lane, 2/3/2 for one -- the chain is half as long there)
#ifndef CTC_F5_X
#define CTC_F5_X (NL == 1 ? 2 : 1)
#endif
#ifndef CTC_F5_Y
#define CTC_F5_Y 3
#endif
template <int BLK, int NH, int NL>
struct P1Split {
  // NH = 4 (12-frame blocks): X / X / Y / Y / R as above.  NH = 2 (6-frame blocks of the 4-positions-per-lane variant):
  // the two helpers and the recompute wavefront take a third each.
  // NH = 1 (3-frame blocks of the 8-positions-per-lane variant): two frames for the helper, one for the recompute wavefront.
  static constexpr int X = NH == 4 ? CTC_F5_X : NH == 2 ? BLK / 3 : 2, Y = NH == 4 ? CTC_F5_Y : NH == 2 ? BLK / 3 : 0;
  static constexpr int R = NH == 4 ? BLK - 2 * X - 2 * Y : NH == 2 ? BLK - X - Y : BLK - X;
  static_assert(NH == 4 || NH == 2 || NH == 1, "helpers per side");
  static_assert(X >= 0 && Y >= 0 && R >= 0 && X <= 6 && Y <= 6 && R <= 6, "phase-1 split: at most 6 frames per worker");
  // worker: 0 .. NH-1 = helpers, NH = recompute wavefront
  static constexpr int count(int worker) {
    if (NH == 4) return worker < 2 ? X : worker < 4 ? Y : R;
    if (NH == 1) return worker == 0 ? X : R;
    return worker == 0 ? X : worker == 1 ? Y : R;
  }
  static constexpr int first(int worker) {
    int f = 0;
    for (int w = 0; w < worker; ++w) f += count(w);
    return f;
  }
};

template <int KIND, int NL, int NH, int BLK, int VPL, int SIDE, int P0, int NQ, class S_t>
__device__ __forceinline__ void estage1(const S_t &S, Lds<KIND, NL, NH, BLK, VPL> &lds, const Geo<BLK> &geo,
                                        float2 *__restrict__ stats, float *dump, int lane, Stamps &st) {
  using LD = Lds<KIND, NL, NH, BLK, VPL>;
  const int len = geo.len;
  const int nb = geo.nblocks(1, SIDE);
  auto write_E = [&](float *row, const Emis<NL> &e) __attribute__((always_inline)) {
    st_slots<NL>(row + lane * NL, e.y);
    float *tq = (lane == 0) ? row + LD::UP : dump + (lane & 15) * 4;  // lanes >= 16 overlap in the sink: harmless
    *reinterpret_cast<float4 *>(tq) = make_float4(e.bl, e.mx, e.l2s, 0.f);
  };
  auto fr = [&](int j, int d) -> int {  // frame at position d of this side's block j, clamped so prefetches stay legal
    int jj = j < nb ? j : nb - 1;
    jj = jj < 0 ? 0 : jj;
    const int g = geo.absblock(1, SIDE, jj);
    const int nv = geo.nvof(g);
    int dd = d < nv ? d : nv - 1;
    int t = geo.frame(SIDE, g, dd < 0 ? 0 : dd);
    t = t < len ? t : len - 1;
    return t < 0 ? 0 : t;
  };
  constexpr int NQA = NQ > 0 ? NQ : 1;
  float4 xb[NQA][VPL];
  static_for<0, NQA>([&](auto Q) {
    constexpr int q = decltype(Q)::value;
    static_for<0, VPL>([&](auto W) { xb[q][decltype(W)::value] = make_float4(0.f, 0.f, 0.f, 0.f); });
    if (NQ > 0 && nb > 0) S.load_x(xb[q], fr(0, P0 + q));
  });
  for (int it = 0; it <= geo.NB; ++it) {
    const int j = it;
    if (NQ > 0 && j < nb) {
      const int g = geo.absblock(1, SIDE, j);
      const int nv = geo.nvof(g);
      float(*E)[LD::ES] = lds.E[SIDE][j % 3];
      float smx = 0.f, sl2 = 0.f;  // lane d keeps the statistics of position d of the block
      if (nv == BLK) {
        if constexpr (NQ > 0) {
          float4 xq[NQA][VPL];
          Emis<NL> e[NQA];
          static_for<0, NQA>([&](auto Q) {
            constexpr int q = decltype(Q)::value;
            static_for<0, VPL>([&](auto W) {
              constexpr int w = decltype(W)::value;
              xq[q][w] = make_float4(xb[q][w].x, xb[q][w].y, xb[q][w].z, xb[q][w].w);
            });
          });
          S.template emit_n<NQA>(xq, e);  // NQ frames in one batch
          static_for<0, NQA>([&](auto Q) {
            constexpr int q = decltype(Q)::value;
            write_E(E[P0 + q], e[q]);
            smx = (lane == P0 + q) ? e[q].mx : smx;
            sl2 = (lane == P0 + q) ? e[q].l2s : sl2;
          });
        }
      } else {
        for (int q = 0; q < NQ; ++q) {
          const int d = P0 + q;
          if (d < nv) {
            float4 xr[VPL];
            S.load_x(xr, geo.frame(SIDE, g, d));
            Emis<NL> e;
            S.emit(xr, 0, e);
            write_E(E[d], e);
            smx = (lane == d) ? e.mx : smx;
            sl2 = (lane == d) ? e.l2s : sl2;
          }
        }
      }
      static_for<0, NQA>([&](auto Q) {
        constexpr int q = decltype(Q)::value;
        S.load_x(xb[q], fr(j + 1, P0 + q));
      });
      if (lane >= P0 && lane < P0 + NQ && lane < nv) stats[geo.frame(SIDE, g, lane)] = make_float2(smx, sl2);
    }
    st.mid();
    block_barrier();
    st.end();
  }
}

// ------------------------------------------------------------------------------------------------
// main chain
// ------------------------------------------------------------------------------------------------
template <int KIND, int NL, int NH, int BLK, int VPL, int DIR>
__device__ __forceinline__ void run_main(const Problem &p, const Layout &L, float *__restrict__ alpha_ws,
                                         float *__restrict__ beta_ws, double *__restrict__ logp_ws,
                                         float *__restrict__ loss, Lds<KIND, NL, NH, BLK, VPL> &lds, const Geo<BLK> &geo,
                                         void *stamp_ws, bool want_grad, int b) {
  using S_t = Side<KIND, NL, VPL, DIR, true, 0, double>;  // (float64 lattice state: ctc_common.h lse2)
  using LD = Lds<KIND, NL, NH, BLK, VPL>;
  S_t S;
  const int lane = threadIdx.x & 63;
  const int T = p.T, UP = L.UP;
  S.lane = lane; S.UP = UP; S.blank = p.blank; S.SRS = L.SRS;
  const int len = geo.len;
  S.len = len;
  int ll = p.label_length[b] < 0 ? 0 : p.label_length[b];
  const bool shape_ok = (ll <= p.U);
  if (!shape_ok) ll = 0;
  S.ll = ll;
  S.own_rows = (DIR == 0 ? alpha_ws : beta_ws) + (long)b * L.rows_b * L.SRS;
  S.oth_rows = (DIR == 0 ? beta_ws : alpha_ws) + (long)b * L.rows_b * L.SRS;
  // row index of lattice time t (a block boundary, the meeting point or `len`): t itself in the full layout, the block
  // slot in the compact one (Layout::ck_blk)
  auto rowidx = [&](int t) -> int { return L.ck_blk > 0 ? (t + BLK - 1) / BLK : t; };
  S.off = 0.0;
  init_labels<KIND, NL>(S, p, b, lane, ll);
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
  float *dump = lds.dump[DIR];
  STAMP(Stamps st; st.begin());

  // ================= phase 1: lattice steps, one checkpoint row per block =================
  {
    const int nb = geo.nblocks(1, DIR);
    for (int it = 0; it <= geo.NB; ++it) {
      const int j = it - 1;
      if (j >= 0 && j < nb) {
        const int g = geo.absblock(1, DIR, j);
        const int nv = geo.nvof(g);
        const float(*E)[LD::ES] = lds.E[DIR][j % 3];
        // checkpoint: the state at the boundary this block starts from (alpha[BLK g] / beta[BLK g + nv])
        S.spill(rowidx(DIR == 0 ? BLK * g : BLK * g + nv), 0.f, 0.f);
        if (nv == BLK) {
#pragma unroll
          for (int d = 0; d < BLK; ++d) {
            Emis<NL> e;
            read_E<NL, LD>(E[d], lane, e);
            S.step(e);
          }
        } else {
          for (int d = 0; d < nv; ++d) {
            Emis<NL> e;
            read_E<NL, LD>(E[d], lane, e);
            S.step(e);
          }
        }
        S.renorm();
      }
      STAMP(st.mid());
      block_barrier();
      STAMP(st.end());
    }
  }
  S.spill(rowidx(geo.tm), 0.f, 0.f);  // alpha[tm] / beta[tm]: the meeting row

  STAMP(st.phase1_done());
  // ================= meeting point =================
  __syncthreads();  // full drain: the checkpoint rows of both chains are in L2 before anybody reads them
  double dlogp;
  {
    SRow<KIND, NL> r;
    load_srow<KIND, NL>(r, S.oth_rows + (long)rowidx(geo.tm) * L.SRS, lane, UP);
    dlogp = S.meet(r);
    if (!shape_ok) dlogp = -INFINITY;
  }
  if (DIR == 0 && lane == 0) {
    logp_ws[b] = dlogp;
    loss[b] = (dlogp == -INFINITY) ? INFINITY : (float)(-dlogp * LN2_D);
    lds.feasible = (dlogp != -INFINITY);
  }
  __syncthreads();
  if (!want_grad) return;  // loss only (grad == NULL): every role leaves here, after the same barriers
  if (dlogp == -INFINITY) dlogp = 0.0;  // infeasible: keep the barrier schedule; the helpers write zeros instead

  // ================= phase 2: everything from LDS =================
  {
    const int nb = geo.nblocks(2, DIR);
    for (int it = 0; it <= geo.NB + 2; ++it) {
      const int j = it - 2;
      if (j >= 0 && j < nb) {
        const int g = geo.absblock(2, DIR, j);
        const int nv = geo.nvof(g);
        const float(*E)[LD::ES] = lds.E[DIR][j % 3];
        float(*RR)[LD::RS] = lds.R[DIR][j % 3];
        // every row of the block carries the offset of the checkpoint it was regenerated from, and this chain
        // renormalises only at block ends: the posterior scale is a per-block constant
        const float4 tl0 = *reinterpret_cast<const float4 *>(RR[0] + (KIND == 0 ? 2 : 1) * LD::UP);
        const float sc = (float)((double)tl0.z + (S.off - dlogp)) + tl0.w;
        auto one = [&](int d) __attribute__((always_inline)) {
          Emis<NL> e;
          read_E<NL, LD>(E[d], lane, e);
          SRow<KIND, NL> r;
          read_R<KIND, NL, LD>(RR[d], lane, r);
          float s1[NL], s2[NL], s0;
          S.post_step_sc(e, r, 0.f, s1, s2, s0);  // exponents WITHOUT the scale: the G stage adds it (off the chain)
          float *row = RR[d];  // S row in place
          st_pairs<NL>(row + 2 * lane * NL, s1, s2);
          float *tq = (lane == 0) ? row + 2 * LD::UP : dump + lane;
          *reinterpret_cast<float2 *>((lane == 0) ? row + 2 * LD::UP : dump + (lane & 31) * 2) = make_float2(s0, sc);
          (void)tq;
        };
        if (nv == BLK) {
#pragma unroll
          for (int d = 0; d < BLK; ++d) one(d);
        } else {
          for (int d = 0; d < nv; ++d) one(d);
        }
        S.renorm();
      }
      STAMP(st.mid());
      block_barrier();
      STAMP(st.end());
    }
  }
  STAMP(st.dump(reinterpret_cast<unsigned long long *>(stamp_ws) + ((long)b * LD::NW + DIR) * 4, lane));
}

// ------------------------------------------------------------------------------------------------
// recompute chain of side SIDE (phase 2): runs the OTHER direction's recursion inside one block, from that direction's
// checkpoint, and leaves the rows its main chain needs in LDS.  R[d] = the row main needs at position d of the block:
//   SIDE A (needs beta[t+1] at frame t = BLK g + d)      : R[nv-1] = checkpoint beta[BLK g + nv]; step frames downward
//   SIDE B classic (needs alpha[t+1] at t = BLK g + nv-1-d): step frames upward from alpha[BLK g], row after each step
//   SIDE B simplified (needs a[t])                        : row before each step
// ------------------------------------------------------------------------------------------------
template <int KIND, int NL, int NH, int BLK, int VPL, int SIDE, int XT>
__device__ __forceinline__ void run_recompute(const Problem &p, const Layout &L, const float *__restrict__ alpha_ws,
                                              const float *__restrict__ beta_ws, float2 *__restrict__ stats_ws,
                                              Lds<KIND, NL, NH, BLK, VPL> &lds, const Geo<BLK> &geo, void *stamp_ws,
                                              bool want_grad, int b) {
  constexpr int RDIR = 1 - SIDE;  // direction of the recursion this wave runs
  using S_t = Side<KIND, NL, VPL, RDIR, true, XT, double>;
  using LD = Lds<KIND, NL, NH, BLK, VPL>;
  S_t S;
  const int lane = threadIdx.x & 63;
  const int T = p.T, UP = L.UP;
  S.lane = lane; S.UP = UP; S.blank = p.blank; S.SRS = L.SRS;
  int ll = p.label_length[b] < 0 ? 0 : p.label_length[b];
  if (ll > p.U) ll = 0;
  S.ll = ll;
  S.off = 0.0;
  S.cx = NEG;
  init_labels<KIND, NL>(S, p, b, lane, ll);
  // checkpoints of the direction this wave runs: written by the OTHER side's main chain in phase 1
  const float *ck_rows = (RDIR == 0 ? alpha_ws : beta_ws) + (long)b * L.rows_b * L.SRS;
  float *dump = lds.dump[2 + SIDE];
  STAMP(Stamps st; st.begin());

  {  // phase 1: nothing to recompute yet -- this wavefront works the E stage of its side (estage1)
    using SP = P1Split<BLK, NH, NL>;
    S.xbase = XT != 2 ? p.logits + (long)b * p.xsb
                      : reinterpret_cast<const float *>(reinterpret_cast<const unsigned short *>(p.logits) + (long)b * p.xsb);
    S.xst = p.xst;
    S.Vr = p.V;
    S.xs = lds.xcopy_r[SIDE];
    if (lane == 0) S.xs[256 * VPL] = -6.0e29f;  // pad slot of the gather copy: "log 0" for label positions beyond label_length
    float2 *stats = stats_ws + (long)b * T;
    estage1<KIND, NL, NH, BLK, VPL, SIDE, SP::first(NH), SP::count(NH)>(S, lds, geo, stats, dump, lane, st);
  }
  STAMP(st.phase1_done());
  __syncthreads();
  __syncthreads();
  if (!want_grad) return;

  const int nb = geo.nblocks(2, SIDE);
  auto ck_index = [&](int j) -> int {
    int jj = j < nb ? j : nb - 1;
    jj = jj < 0 ? 0 : jj;
    const int g = geo.absblock(2, SIDE, jj);
    int idx = (SIDE == 0) ? BLK * g + geo.nvof(g) : BLK * g;  // beta at the upper boundary / alpha at the lower one
    idx = idx < 0 ? 0 : idx;
    return L.ck_blk > 0 ? (idx + BLK - 1) / BLK : idx;  // (row index: see run_main)
  };
  SRow<KIND, NL> ck_next;
  load_srow<KIND, NL>(ck_next, ck_rows + (long)ck_index(0) * L.SRS, lane, UP);
  for (int it = 0; it <= geo.NB + 2; ++it) {
    const int j = it - 1;
    if (j >= 0 && j < nb) {
      const int g = geo.absblock(2, SIDE, j);
      const int nv = geo.nvof(g);
      const float(*E)[LD::ES] = lds.E[SIDE][j % 3];
      float(*RR)[LD::RS] = lds.R[SIDE][j % 3];
      const SRow<KIND, NL> ck = ck_next;
      load_srow<KIND, NL>(ck_next, ck_rows + (long)ck_index(j + 1) * L.SRS, lane, UP);  // next block's checkpoint, a block ahead
      restore_state<KIND, NL, RDIR>(S, ck);
      auto put = [&](int d) __attribute__((always_inline)) {
        float cs[NL], os[NL];
        float4 tail;
        state_row<KIND, NL, RDIR>(S, cs, tail);
#pragma unroll
        for (int j = 0; j < NL; ++j) os[j] = (float)S.o[j];
        write_R<KIND, NL, LD>(RR[d], dump, lane, cs, os, tail);
      };
      auto stp = [&](int d) __attribute__((always_inline)) {
        Emis<NL> e;
        read_E<NL, LD>(E[d], lane, e);
        S.step(e);
      };
      if constexpr (SIDE == 0) {
        // beta recursion downward: R[nv-1] = beta[BLK g + nv] (the checkpoint), then R[d-1] = beta[BLK g + d] after frame d
        write_R<KIND, NL, LD>(RR[nv - 1], dump, lane, ck.a, ck.b, ck.tail);
        if (nv == BLK) {
#pragma unroll
          for (int d = BLK - 1; d >= 1; --d) { stp(d); put(d - 1); }
        } else {
          for (int d = nv - 1; d >= 1; --d) { stp(d); put(d - 1); }
        }
      } else {
        // alpha recursion upward; B's position d holds frame BLK g + nv-1-d (B-side blocks of phase 2 are always full)
        if constexpr (KIND == 0) {
          if (nv == BLK) {
#pragma unroll
            for (int i = 0; i < BLK; ++i) { stp(BLK - 1 - i); put(BLK - 1 - i); }
          } else {
            for (int i = 0; i < nv; ++i) { stp(nv - 1 - i); put(nv - 1 - i); }
          }
        } else {
          write_R<KIND, NL, LD>(RR[nv - 1], dump, lane, ck.a, ck.b, ck.tail);  // a[BLK g]
          if (nv == BLK) {
#pragma unroll
            for (int i = 1; i < BLK; ++i) { stp(BLK - i); put(BLK - 1 - i); }
          } else {
            for (int i = 1; i < nv; ++i) { stp(nv - i); put(nv - 1 - i); }
          }
        }
      }
    }
    STAMP(st.mid());
    block_barrier();
    STAMP(st.end());
  }
  STAMP(st.dump(reinterpret_cast<unsigned long long *>(stamp_ws) + ((long)b * LD::NW + 2 + SIDE) * 4, lane));
}

// ------------------------------------------------------------------------------------------------
// helper wavefront h of NH per side: positions d = h, h + NH, ... of every block (FPH = BLK / NH per block)
// ------------------------------------------------------------------------------------------------
template <int KIND, int NL, int NH, int BLK, int VPL, int DIR, int XT>
__device__ __forceinline__ void run_helper(const Problem &p, const Layout &L, float2 *__restrict__ stats_ws,
                                           const float *__restrict__ d_loss, float *__restrict__ grad,
                                           Lds<KIND, NL, NH, BLK, VPL> &lds, const Geo<BLK> &geo, int h, void *stamp_ws, int b) {
  constexpr int V = 256 * VPL;
  constexpr int FPH = BLK / NH;
  using S_t = Side<KIND, NL, VPL, DIR, true, XT>;
  using LD = Lds<KIND, NL, NH, BLK, VPL>;
  S_t S;
  const int lane = threadIdx.x & 63;
  const int T = p.T;
  S.lane = lane; S.UP = L.UP; S.blank = p.blank;
  const int len = geo.len;
  int ll = p.label_length[b] < 0 ? 0 : p.label_length[b];
  if (ll > p.U) ll = 0;
  S.ll = ll;
  if constexpr (XT != 2) {
    S.xbase = p.logits + (long)b * p.xsb;
    S.gbase = grad + (long)b * p.gsb;
  } else {  // bfloat16 producer/consumer: the pointers are element-typed inside load_x / store_g
    S.xbase = reinterpret_cast<const float *>(reinterpret_cast<const unsigned short *>(p.logits) + (long)b * p.xsb);
    S.gbase = reinterpret_cast<float *>(reinterpret_cast<unsigned short *>(grad) + (long)b * p.gsb);
  }
  S.xst = p.xst;
  S.gst = p.gst;
  S.Vr = p.V;
  S.xs = lds.xcopy[DIR * NH + h];
  S.bins = lds.bins[DIR * NH + h];
  S.dl = d_loss ? d_loss[b] : 1.0f;
  init_labels<KIND, NL>(S, p, b, lane, ll);
  if (lane == 0) S.xs[V] = -6.0e29f;  // pad slot of the gather copy: "log 0" for label positions beyond label_length
  float2 *stats = stats_ws + (long)b * T;
  float *dump = lds.dump[4 + DIR * NH + h];
  STAMP(Stamps st; st.begin());

  auto write_E = [&](float *row, const Emis<NL> &e) __attribute__((always_inline)) {
    st_slots<NL>(row + lane * NL, e.y);
    float *tq = (lane == 0) ? row + LD::UP : dump + (lane & 15) * 4;  // lanes >= 16 overlap in the sink: harmless
    *reinterpret_cast<float4 *>(tq) = make_float4(e.bl, e.mx, e.l2s, 0.f);
  };
  // frame at position d of this side's block j of `phase`, clamped into [0, len) so that prefetches past the end are legal
  auto fr = [&](int phase, int j, int d) -> int {
    const int nb = geo.nblocks(phase, DIR);
    int jj = j < nb ? j : nb - 1;
    jj = jj < 0 ? 0 : jj;
    const int g = geo.absblock(phase, DIR, jj);
    const int nv = geo.nvof(g);
    int dd = d < nv ? d : nv - 1;
    int t = geo.frame(DIR, g, dd < 0 ? 0 : dd);
    t = t < len ? t : len - 1;
    return t < 0 ? 0 : t;
  };

  // ================= phase 1: E stage with statistics (estage1 above) =================
  {
    using SP = P1Split<BLK, NH, NL>;
    if constexpr (NH == 4) {
      switch (h) {
        case 0: estage1<KIND, NL, NH, BLK, VPL, DIR, SP::first(0), SP::count(0)>(S, lds, geo, stats, dump, lane, st); break;
        case 1: estage1<KIND, NL, NH, BLK, VPL, DIR, SP::first(1), SP::count(1)>(S, lds, geo, stats, dump, lane, st); break;
        case 2: estage1<KIND, NL, NH, BLK, VPL, DIR, SP::first(2), SP::count(2)>(S, lds, geo, stats, dump, lane, st); break;
        default: estage1<KIND, NL, NH, BLK, VPL, DIR, SP::first(3), SP::count(3)>(S, lds, geo, stats, dump, lane, st); break;
      }
    } else if constexpr (NH == 2) {
      if (h == 0) estage1<KIND, NL, NH, BLK, VPL, DIR, SP::first(0), SP::count(0)>(S, lds, geo, stats, dump, lane, st);
      else estage1<KIND, NL, NH, BLK, VPL, DIR, SP::first(1), SP::count(1)>(S, lds, geo, stats, dump, lane, st);
    } else {
      estage1<KIND, NL, NH, BLK, VPL, DIR, SP::first(0), SP::count(0)>(S, lds, geo, stats, dump, lane, st);
    }
  }

  STAMP(st.phase1_done());
  // ================= meeting point =================
  __syncthreads();
  __syncthreads();
  if (grad == nullptr) return;  // loss only
  const bool feasible = lds.feasible != 0;

  // ================= phase 2: E stage (statistics from the record), G stage three blocks behind =================
  {
    const int nb = geo.nblocks(2, DIR);
    if (h == 0) {
      if (!feasible) {  // zero gradient for the whole sample (base_loss.py:283-288); barrier schedule unchanged
        if constexpr (DIR == 0) S.zero_rows(geo.tm, T); else S.zero_rows(0, geo.tm);
      } else if (DIR == 0) {
        S.zero_rows(len, T);  // padded frames (base_loss.py:291-296)
      }
    }
    // The logits rows of a block are loaded one block ahead of its E stage and used again by its G stage three blocks
    // later: five blocks are alive at a time.  They sit in a ring of five register sets addressed by (block mod 5) at
    // COMPILE time -- the loop is unrolled by five -- instead of being moved from set to set every block (48 v_mov per
    // block, 7 % of the phase-2 instructions of a helper).  The statistics of a block travel the same way.
    // Wide vocabularies (four row segments per lane) cannot afford five register sets: there the ring holds the block in
    // its E stage and the one being loaded, and the G stage re-reads its rows and statistics (L2 hits) at the top of the
    // iteration.
    constexpr bool RELOAD = VPL >= 4;
    constexpr int RING = RELOAD ? 2 : 5;
    float4 X[RING][FPH][VPL];
    float4 XG[RELOAD ? FPH : 1][VPL];
    float2 SG[RING];
    float2 sgl = make_float2(0.f, 0.f);
    static_for<0, RING>([&](auto R) {
      SG[decltype(R)::value] = make_float2(0.f, 0.f);
      static_for<0, FPH>([&](auto Q) {
        static_for<0, VPL>([&](auto W) { X[decltype(R)::value][decltype(Q)::value][decltype(W)::value] = make_float4(0.f, 0.f, 0.f, 0.f); });
      });
    });
    float2 st_cur = make_float2(0.f, 0.f), st_next = make_float2(0.f, 0.f);
    if (nb > 0) {
      static_for<0, FPH>([&](auto Q) { S.load_x(X[0][decltype(Q)::value], fr(2, 0, h + NH * decltype(Q)::value)); });
      st_cur = stats[fr(2, 0, lane)];
    }
    auto body = [&](auto R, int it) __attribute__((always_inline)) {
      constexpr int r = decltype(R)::value;         // = it mod RING
      constexpr int rn = (r + 1) % RING;            // block it+1 (being loaded)
      constexpr int rg = (r + 2) % RING;            // block it-3 (G stage; five-set ring only)
      if constexpr (RELOAD) {
        static_for<0, FPH>([&](auto Q) { S.load_x(XG[decltype(Q)::value], fr(2, it - 3, h + NH * decltype(Q)::value)); });
        sgl = stats[fr(2, it - 3, lane)];
      }
      // ---- E stage (block it) ----
      const int j = it;
      SG[r] = st_cur;
#ifdef CTC_DBG_NO_E2
      if (false) {
#else
      if (j < nb) {
#endif
        const int g = geo.absblock(2, DIR, j);
        const int nv = geo.nvof(g);
        float(*E)[LD::ES] = lds.E[DIR][j % 3];
        st_next = stats[fr(2, j + 1, lane)];
        if (__builtin_expect(nv == BLK, 1)) {
          static_for<0, FPH>([&](auto Q) {
            constexpr int q = decltype(Q)::value;
            const int d = h + NH * q;
            Emis<NL> e;
            S.gather(X[r][q], 0, readlane_f(st_cur.x, d), readlane_f(st_cur.y, d), e);
            write_E(E[d], e);
          });
        } else {
          for (int d = h; d < nv; d += NH) {
            float4 xr[VPL];
            S.load_x(xr, geo.frame(DIR, g, d));
            float2 sd = stats[geo.frame(DIR, g, d)];
            Emis<NL> e;
            S.gather(xr, 0, sd.x, sd.y, e);
            write_E(E[d], e);
          }
        }
        static_for<0, FPH>([&](auto Q) { S.load_x(X[rn][decltype(Q)::value], fr(2, j + 1, h + NH * decltype(Q)::value)); });
        st_cur = st_next;
      }
      // ---- G stage (block it-3): posterior scatter + gradient rows ----
      const int gj = it - 3;
#ifdef CTC_DBG_NO_G
      if (false) {
#else
      if (feasible && gj >= 0 && gj < nb) {
#endif
        const int g = geo.absblock(2, DIR, gj);
        const int nv = geo.nvof(g);
        const float(*SR)[LD::RS] = lds.R[DIR][gj % 3];
        auto g_frame = [&](int d, const float4(&xr)[VPL], float mx, float l2s) __attribute__((always_inline)) {
          const float *row = SR[d];
          float s1[NL], s2[NL];
          ld_pairs<NL>(row + 2 * lane * NL, s1, s2);
          const float2 t0 = *reinterpret_cast<const float2 *>(row + 2 * LD::UP);  // (s0, posterior scale of the block)
          const float sc30 = t0.y + 30.0f;  // block scale + the 2^30 fixed-point unit of the token row (grad_row30)
#pragma unroll
          for (int jj = 0; jj < NL; ++jj) { s1[jj] += sc30; s2[jj] += sc30; }
          const float s0 = t0.x + sc30;
          Emis<NL> e;
          e.mx = mx; e.l2s = l2s;
          S.grad_row30(geo.frame(DIR, g, d), s1, s2, s0, xr, e);
        };
        if (__builtin_expect(nv == BLK, 1)) {
          static_for<0, FPH>([&](auto Q) {
            constexpr int q = decltype(Q)::value;
            const int d = h + NH * q;
            if constexpr (RELOAD) g_frame(d, XG[q], readlane_f(sgl.x, d), readlane_f(sgl.y, d));
            else g_frame(d, X[rg][q], readlane_f(SG[rg].x, d), readlane_f(SG[rg].y, d));
          });
        } else {
          for (int d = h; d < nv; d += NH) {
            float4 xr[VPL];
            S.load_x(xr, geo.frame(DIR, g, d));
            float2 sd = stats[geo.frame(DIR, g, d)];
            g_frame(d, xr, sd.x, sd.y);
          }
        }
      }
      STAMP(st.mid());
      block_barrier();
      STAMP(st.end());
    };
    for (int it0 = 0; it0 <= geo.NB + 2; it0 += RING) {
      static_for<0, RING>([&](auto R) {
        if (it0 + decltype(R)::value <= geo.NB + 2) body(R, it0 + decltype(R)::value);
      });
    }
  }
  STAMP(st.dump(reinterpret_cast<unsigned long long *>(stamp_ws) + ((long)b * LD::NW + 4 + DIR * NH + h) * 4, lane));
}

// Everything the workgroup does for utterance b (the body of fused5_kernel; also called by fused6_kernel for the utterances
// the linear-domain recursion cannot hold, on the same wavefront roles, LDS reused).
// a wavefront without a role (the caller runs more wavefronts than 4 + 2 NH): the barriers of the roles, nothing else
template <int BLK>
__device__ __forceinline__ void run_idle(const Problem &p, bool want_grad, int b) {
  Geo<BLK> geo;
  geo.init(clampi(p.logit_length[b], 0, p.T));
  for (int it = 0; it <= geo.NB; ++it) block_barrier();
  __syncthreads();
  __syncthreads();
  if (!want_grad) return;
  for (int it = 0; it <= geo.NB + 2; ++it) block_barrier();
}

template <int KIND, int NL, int NH, int BLK, int VPL, int XT>
__device__ __forceinline__ void run_roles(const Problem &p, const Layout &L, float *__restrict__ alpha_ws, float *__restrict__ beta_ws,
                                          double *__restrict__ logp_ws, float2 *__restrict__ stats_ws, float *__restrict__ loss,
                                          const float *__restrict__ d_loss, float *__restrict__ grad, void *stamp_ws,
                                          Lds<KIND, NL, NH, BLK, VPL> &lds, int w, int b) {
  Geo<BLK> geo;  // every wavefront derives the same block schedule: the barrier counts match by construction
  geo.init(clampi(p.logit_length[b], 0, p.T));
  if (w == 0) {
    __builtin_amdgcn_s_setprio(3);  // the sequential chains win issue arbitration against co-resident helpers
    run_main<KIND, NL, NH, BLK, VPL, 0>(p, L, alpha_ws, beta_ws, logp_ws, loss, lds, geo, stamp_ws, grad != nullptr, b);
  } else if (w == 1) {
    __builtin_amdgcn_s_setprio(3);
    run_main<KIND, NL, NH, BLK, VPL, 1>(p, L, alpha_ws, beta_ws, logp_ws, loss, lds, geo, stamp_ws, grad != nullptr, b);
  } else if (w == 2) {
    __builtin_amdgcn_s_setprio(2);
    run_recompute<KIND, NL, NH, BLK, VPL, 0, XT>(p, L, alpha_ws, beta_ws, stats_ws, lds, geo, stamp_ws, grad != nullptr, b);
  } else if (w == 3) {
    __builtin_amdgcn_s_setprio(2);
    run_recompute<KIND, NL, NH, BLK, VPL, 1, XT>(p, L, alpha_ws, beta_ws, stats_ws, lds, geo, stamp_ws, grad != nullptr, b);
  } else if (w < 4 + NH) {
    run_helper<KIND, NL, NH, BLK, VPL, 0, XT>(p, L, stats_ws, d_loss, grad, lds, geo, w - 4, stamp_ws, b);
  } else {
    run_helper<KIND, NL, NH, BLK, VPL, 1, XT>(p, L, stats_ws, d_loss, grad, lds, geo, w - 4 - NH, stamp_ws, b);
  }
}

}  // namespace fused5
}  // namespace ctc

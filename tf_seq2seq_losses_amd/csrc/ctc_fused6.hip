// Fused loss + gradient kernel for gfx950, pipeline v6: the lattice recursion in the LINEAR domain.
//
// Same decomposition as ctc_fused5.hip (one workgroup of 4 + 2*NH wavefronts per utterance: two main chains that meet in
// the middle, two recompute chains, NH helpers a side, one checkpoint row per block in HBM, everything else in LDS, ONE
// raw s_barrier per block), but alpha / beta are carried as float32 MANTISSAS with ONE INTEGER EXPONENT PER LANE (block
// floating point over the NL label positions of a lane) instead of base-2 logarithms:
//
//   * a lattice step is 2 adds + 2 multiplies per label position instead of two log-sum-exp (4 transcendentals): the
//     chains issue ~3x fewer instructions, and the whole kernel moves from VALU-issue-bound to memory-bound;
//   * every operation carries a RELATIVE error of 2^-24 instead of an absolute error of 2^-24 * |log value|: at T = 1000
//     the gradient agrees with a float64 evaluation to ~1e-6 (log-domain float32: 2.6e-4 here, 3e-3 for a scalar port);
//   * emissions are exp(x - rowmax), NOT normalised: log2 sum_k exp(x_k - rowmax) of every frame is accumulated in
//     double by the worker that first touches the frame and enters the loss only; posteriors are scale free;
//   * a row of the lattice spans hundreds of binary orders of magnitude even for N(0,1) logits (alpha peaks at the last
//     label position while beta peaks at the first), which one exponent per ROW cannot hold in float32; one exponent per
//     LANE does: neighbouring label positions differ by a few bits.  Every RN frames each lane renormalises to its own
//     maximum (no cross-lane reduction) and re-aligns the one value it receives from its neighbour lane by the difference
//     of the two exponents (one v_ldexp_f32 per step).
//
// The float32 mantissas flush below 2^-126.  Inputs whose dynamic range inside a lane exceeds what that leaves (logits
// of +-1e10, -inf, hard zeros in the needed emissions -- README.md:74-78) are DETECTED, not approximated: the kernel
// raises a per-utterance flag (conditions D1..D5 below) and the caller re-runs exactly those utterances through the
// log-domain kernel of ctc_fused5.hip, which has no range limit.  For unflagged utterances the mass lost to flushing is
// bounded by 2^-28 of P.
//
//   D1  P == 0, inf or NaN at the meeting point (includes structurally infeasible utterances)
//   D2  an emission needed by the lattice (label token or blank inside label_length) below 2^-120 of the row maximum: an
//       emission that flushes is lost by BOTH chains alike, the one loss the check D6 cannot see
//   D6  (calls with a gradient) the posterior mass of a frame, sum over all lattice states of alpha beta / P, differs from
//       1 by more than MASS_TOL = 3e-5 (ctc_linear_flags.h; each helper's LAST frame of every block is sampled).  Flushing only ever removes mass, and mass that one chain loses at (t1, s1) is still carried
//       by the other chain on the far side of t1, so sum_s alpha_t beta_t stops being the same for all t: the sampled frames are
//       checked against P from the meeting point.  A loss below 3e-5 of P is below the tolerance of the gradient.
//   D5  a lane's alpha and beta exponents exceed log2 P by more than 90 in the posterior: mantissa products that underflow
//       would no longer be negligible (see KK_MAX)
//   D3, D4 (calls WITHOUT a gradient have no phase 2 to check the mass in; they fall back on conservative local signs)
//       D3 a renormalisation scales a lane's own live values down by more than 2^-64; D4 a lane's maximum decays by more
//       than 2^-64 within one renormalisation period, or a live lane goes to zero
//   D7  (calls WITHOUT a gradient) a needed emission below 2^-16 of its row maximum: sharp logits (see EMIS_SOFT)
//
// References: classic_ctc_loss.py:310-462,565-669, simplified_ctc_loss.py:291-438,456-534, base_loss.py:262-298,328-344,
// 420-468, tools.py:27-40.
#include "ctc_fused_common.h"
#include "ctc_swap_reduce.h"
#include "ctc_linear_flags.h"
#include "ctc_fused5_roles.h"  // the log-domain roles: run inside this kernel for the utterances it flags

#ifndef CTC_FUSED_KIND
#error "compile with -DCTC_FUSED_KIND=0 (classic) or 1 (simplified)"
#endif

// (the CTC_F6_* switches below exist in CTC_DIAG builds only: ctc_common.h clears them otherwise)
namespace ctc {
namespace fused6 {

using namespace ctc::fused;

using linear::DEAD; using linear::GAP; using linear::GAP_WIDE; using linear::DOWN_MAX; using linear::DECAY_MAX; using linear::KK_MAX;
using linear::KK_MAX2; using linear::EMIS_MIN; using linear::MASS_TOL;  // (ctc_linear_flags.h: shared with ctc_hvp_fused.hip)
#ifndef CTC_F6_GAP_LIVE
#define CTC_F6_GAP_LIVE 16
#endif
// Gap to which a lane that HOLDS mass is lifted towards its upstream neighbour.  It has to be the adoption gap: r04 tried 80 (a live
// lane's own thin values then survive 2^64 deeper -- tests/tools/linear_model.py shows the mass of tests/golden/soak_case_endloss_u128.npz
// intact with it), but mantissas then reach 2^120 where a steep front crosses thin live lanes, the posterior PRODUCTS of phase 2
// overflow, and between the frames D6 samples that went unnoticed: a gradient 3.0 off, unflagged (tests/tools/flag_stats.py, cell
// sigma 5, V = 3, U = 32, slack 2).  With 16 per level and LV levels a mantissa stays below 2^55 and a product below 2^110.
constexpr int GAP_LIVE = CTC_F6_GAP_LIVE;
// GAP_WIDE (ctc_linear_flags.h):      // ... when ONE level suffices (a lane of 4 or 8 label positions is never crossed within a period):
                                  // neighbouring lanes then differ by 2^100 and more on benign inputs, and lifting a lane to
                                  // 2^-16 of its neighbour pushed its own values towards the float32 underflow (D4)
// D3 / D4: a lane's own values pushed 2^-96 below its exponent (by a larger inflow scale / by decay).  A float32 mantissa
// holds them down to 2^-126, so nothing is lost yet; the margin is for what happens before the next renormalisation.  (64
// was too tight once a lane spans eight label positions: neighbouring lanes then differ by more than 2^64 on benign inputs
// and every loss-only call at U > 256 went to the log domain.)
// DOWN_MAX (ctc_linear_flags.h):      // D3
// DECAY_MAX (ctc_linear_flags.h):     // D4
// KK_MAX (ctc_linear_flags.h):        // Posterior scale 2^KK_MAX at most in ONE factor.  The posterior of a state is (alpha mantissa)(beta
                                  // mantissa) 2^(kA + kB - log2 P); the mantissa PRODUCT underflows below 2^-126, which is harmless
                                  // while the scale stays below 2^90 (the lost term is < 2^-5 units of 2^-30) and fatal beyond --
                                  // sharp logits on a nearly forced alignment get there in the frames just before a renormalisation,
                                  // and so do benign long utterances (T >= 3000: a lane whose two label positions differ by more than
                                  // 2^90 in alpha and by as much the other way in beta while the alignment crosses between them).
                                  // Beyond KK_MAX the scale is applied in TWO factors: the excess 2^(k - KK_MAX) goes onto the chain's
                                  // own operand BEFORE the product (then nothing that matters underflows), the rest after it as before;
                                  // a wave-uniform branch per frame, taken only while some lane of the wavefront needs it.
// KK_MAX2 (ctc_linear_flags.h):      // D5: beyond this even the pre-scaled operand would leave float32
// EMIS_MIN (ctc_linear_flags.h):  // 2^-120 (D2)
// where the forward half of a pair trusts the linear sweeps (see the meeting point): at least BIND_SLACK spare frames, at most
// DWELL_MAX frames per label position, P decaying by at most RATE_MAX_X4 / 4 bits per frame (per lattice kind)
// (decay rate, north-star shape: classic 9.0 bits per frame at N(0, 3^2) -- nothing redone; 9.8 at 3.25^2 -- nothing; 10.5 at 3.5^2 --
// 5 %; 12.1 at 4^2 -- 51 %.  Simplified: 10.7 / 11.6 -- nothing; 12.4 -- 6 %; 13.3 -- 50 %.)
constexpr int BIND_SLACK = 64, DWELL_MAX = 12, RATE_MAX_X4_CLASSIC = 40, RATE_MAX_X4_SIMPLIFIED = 47;
constexpr int DWELL_HARD = 40;      // D10: loss-only calls with more frames per label position than this take the log-domain roles
constexpr int D10_DWELL = 2048;
// D7 (loss-only calls): a needed emission below 2^-16 of its row maximum -- "sharp" logits.  The r03 soak runs found utterances with
// logits N(0, 3^2) on nearly forced alignments (2..15 frames more than labels) whose linear-domain sweeps lose mass that matters later
// WITHOUT tripping D1..D5 (loss off by 1e-4 .. 3e-2 relative): a call with a gradient sees it in the posterior mass (D6) and redoes the
// utterance, a loss-only call has nothing to check against.  Every such case had a needed emission below 2^-18.9; N(0,1) logits stay
// above 2^-13 (the 4.5-sigma tail of 129 000 draws).  So a loss-only call hands sharp utterances to the log domain.
#ifdef CTC_F6_NO_D7  // (diagnostic builds: what do loss-only calls lose without the guard?)
constexpr float EMIS_SOFT = 0.f;
#else
using linear::EMIS_SOFT;                           // 2^-16 (D7)
#endif
// D6: tolerated deviation of a frame's posterior mass from 1.  The gradient of an unflagged utterance is off by about as much, and the
// bar is 1e-4: with a tolerance of 1e-4 the soak runs measured up to 9.0e-5 on unflagged utterances -- no margin (r03).
// MASS_TOL (ctc_linear_flags.h):

#ifdef CTC_F6_STAMPS
// diagnostic build: cycles of work / of waiting at the block barriers, per wavefront and phase (thread-private registers)
struct Stamps {
  unsigned long long work[2] = {0, 0}, wait[2] = {0, 0}, t0 = 0;
  int ph = 0;
};
__device__ Stamps *g_stamps_dummy;
#define F6_ST_ARG , st_
#define F6_STAMP_DECL Stamps st_; st_.t0 = __builtin_amdgcn_s_memtime();
#define F6_STAMP_PHASE2 st_.ph = 1;
#define F6_BARRIER() do { __builtin_amdgcn_s_waitcnt(0xC07F); unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_.work[st_.ph] += t_ - st_.t0; block_barrier_raw(); \
    unsigned long long u_ = __builtin_amdgcn_s_memtime(); st_.wait[st_.ph] += u_ - t_; st_.t0 = u_; } while (0)
#define F6_WAIT(expr) do { unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_.work[st_.ph] += t_ - st_.t0; expr; \
    unsigned long long u_ = __builtin_amdgcn_s_memtime(); st_.wait[st_.ph] += u_ - t_; st_.t0 = u_; } while (0)
#define F6_STAMP_DUMP(wave) do { if ((threadIdx.x & 63) == 0) { unsigned long long *q_ = reinterpret_cast<unsigned long long *>(flag_ws_dbg + p.B) + ((long)b * 16 + (wave)) * 4; \
    q_[0] = st_.work[0]; q_[1] = st_.wait[0]; q_[2] = st_.work[1]; q_[3] = st_.wait[1]; } } while (0)
#else
#define F6_ST_ARG
#define F6_STAMP_DECL
#define F6_STAMP_PHASE2
#define F6_BARRIER() block_barrier_raw()
#define F6_WAIT(expr) do { expr; } while (0)
#define F6_STAMP_DUMP(wave)
#endif
__device__ __forceinline__ void block_barrier_raw() {
#ifdef CTC_F6_SYNC
  __syncthreads();
#else
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's LDS writes have landed; vmcnt untouched
  __builtin_amdgcn_s_barrier();
#endif
}

// ------------------------------------------------------------------------------------------------
// Producer / consumer words in LDS instead of a block barrier (phase 1).  With one s_barrier per block every block lasts as long
// as its slowest wavefront, and the slowest changes from block to block (a late HBM row here, a renormalisation there): r03's
// stamps showed every wavefront WAITING 45 % of phase 1.  Each stage has a dependent chain of ~1 us per block (load -> reduction
// -> exponentials -> reduction -> LDS gather; twelve dependent lattice steps), which the barrier lines up end to end instead
// of letting them overlap.  Now every E-stage worker publishes "my rows of block j are in LDS" in a word of its own, the chain
// publishes "block j is in my registers", and a wavefront only ever waits for what it really needs; the E rows' three slots let
// the workers run up to two blocks ahead of the chain.
//   * LDS operations of one wavefront execute in program order: a progress word written after the rows it announces is
//     visible after them, and the chain's "consumed" word, written after its row reads were issued, lands after they executed.
//   * every wait is bounded: a wavefront that gives up raises D8 (the utterance is redone by the log-domain roles, which use
//     real barriers) and carries on -- nothing can hang.
// ------------------------------------------------------------------------------------------------
#ifndef CTC_F6_P1SYNC
#define CTC_F6_P1SYNC 0
#endif
#ifdef CTC_F6_D9
#define CTC_F6_D9_ON true
#else
#define CTC_F6_D9_ON false
#endif
constexpr int D9_RANGE = 512;                   // flag: a nonzero lattice value left the range a lane's exponent can hold (checked renorm)
constexpr unsigned RANGE_MIN_BITS = 0x0D800000u - 1u;  // bits of 2^-100, minus one: (bits - 1) < this  <=>  0 < value < 2^-100
__device__ __forceinline__ unsigned umin(unsigned a, unsigned b) { return a < b ? a : b; }
constexpr int SYNC_LIMIT = 1 << 16;  // polls (~100 cycles each with the sleep) before a wait gives up
constexpr int D8_SYNC = 256;         // flag: a producer / consumer wait timed out
// (the words are addressed as LDS explicitly: through a generic pointer a volatile access becomes a flat load with `s_waitcnt vmcnt(0)`,
// which would drain every outstanding HBM load of the polling wavefront)
typedef __attribute__((address_space(3))) volatile int lds_vint;
__device__ __forceinline__ lds_vint *as_lds(const int *p) { return (lds_vint *)p; }
// waits until *p >= target (one word, same address in every lane: an LDS broadcast); false on timeout
__device__ __forceinline__ bool wait_word_ge(const int *p, int target) {
  lds_vint *q = as_lds(p);
  for (int n = 0; n < SYNC_LIMIT; ++n) {
    if (__builtin_amdgcn_readfirstlane(*q) >= target) return true;
    __builtin_amdgcn_s_sleep(1);
  }
  return false;
}
// waits until p[0 .. n-1] are all >= target (lane i reads word min(i, n-1))
__device__ __forceinline__ bool wait_words_ge(const int *p, int n, int target, int lane) {
  lds_vint *q = as_lds(p) + (lane < n ? lane : n - 1);
  for (int it = 0; it < SYNC_LIMIT; ++it) {
    if (__builtin_amdgcn_ballot_w64(*q < target) == 0) return true;
    __builtin_amdgcn_s_sleep(1);
  }
  return false;
}
// lane 0 stores v at p, the other lanes into a sink of their own (no exec-mask branch around the store)
__device__ __forceinline__ void publish_word(int *p, float *dump, int lane, int v) {
  const int *q = (lane == 0) ? p : reinterpret_cast<const int *>(dump) + lane;
  *as_lds(q) = v;
}

__device__ __forceinline__ int from_prev_lane_i(int x, int fill) { return __builtin_amdgcn_update_dpp(fill, x, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ int from_next_lane_i(int x, int fill) { return __builtin_amdgcn_update_dpp(fill, x, 0x130, 0xf, 0xf, false); }
__device__ __forceinline__ float ldexp_f(float x, int e) { return __builtin_ldexpf(x, e); }
__device__ __forceinline__ int frexp_e(float x) { return __builtin_amdgcn_frexp_expf(x); }
__device__ __forceinline__ int readlane_i(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
// Packed float32 pairs (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: two label positions per instruction; a wavefront issues one
// vector instruction per ~8 cycles whatever its width, profiles/r03_issue_rate.txt) for the two-positions-per-lane classic chains --
// the roles whose time is their own dependent instruction stream.  NOT for the helpers: their E / G stage arithmetic packed the same
// way (r04: 263 -> 238 instructions per block) made phase 1 two microseconds SLOWER and phase 2 no faster -- a packed operation
// occupies the SIMD for two passes, and the helpers share their SIMDs' pipes with the chains (profiles/r04_kernel_experiments.md).
typedef float f2v __attribute__((ext_vector_type(2)));
#ifndef CTC_F6_PACKED
#define CTC_F6_PACKED 1
#endif
// acc += (x of the upstream neighbour lane) * sc in ONE instruction (v_fmac_f32 with a DPP source; was v_mov_b32_dpp + v_ldexp_f32 +
// v_add_f32).  The lane without an upstream neighbour (0 for wave_shr, 63 for wave_shl) is left unchanged (bound_ctrl off: the
// lane is disabled).  `s_nop 1`: a DPP source written by the preceding VALU instruction needs two wait states, and the compiler
// does not look into inline assembly.
template <int DIR>
__device__ __forceinline__ void fmac_from_upstream(float &acc, float x, float sc) {
  if constexpr (DIR == 0) asm("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(sc));
  else asm("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(sc));
}
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }

// renormalisation period inside a block and the number of lanes the lattice front can cross in one period
template <int BLK, int NL>
struct Cad {
#ifndef CTC_F6_RN12
#define CTC_F6_RN12 6
#endif
  // 12-frame blocks: two label positions per lane renormalise every 6 frames (r03: with the posterior scale in two factors the
  // longer period no longer raises D5 on long utterances; -4 us at the north-star shape), one position per lane every 4
  static constexpr int RN = (BLK % 4 == 0) ? (NL == 2 ? CTC_F6_RN12 : 4) : 3;
  static constexpr int NG = BLK / RN;            // exponent groups of the rows of one block
  static constexpr int LV = (RN + NL - 1) / NL;  // adoption levels: lanes the lattice front can cross in one period
  static constexpr int NSEG = 2 * NG + 1;        // posterior-scale segments of one block (kl_segment)
  static_assert(BLK % RN == 0, "block length must be a multiple of the renormalisation period");
};

// The per-lane posterior scale of the main chain's S rows (run_main) changes when the R rows enter a new exponent group -- before
// the products of that group's first frame -- and after every renormalisation of the main chain; it is written once per such
// SEGMENT, not once per frame (an LDS write per frame on the one wavefront whose instruction count bounds phase 2).  The main
// chain of phase 2 renormalises ren_shift() frames early (after positions RN-2, 2 RN-2, ... where the R rows change group at
// RN-1, 2 RN-1, ...), so that both events open ONE segment and the scales are rebuilt once per period.  kl_segment: the segment
// of position d of a block with nv frames; the main chain counts its own, the helpers look theirs up.
template <int KIND, int DIR>
__device__ __forceinline__ constexpr int ren_shift() { return (KIND == 0 && DIR == 1) ? 0 : 1; }
template <int KIND, int DIR, int RN>
__device__ __forceinline__ constexpr int kl_segment(int d, int nv) {
  int seg = -1, q = -1;
  for (int dd = 0; dd <= d; ++dd) {
    const int s = (KIND == 0 && DIR == 1) ? nv - dd : nv - 1 - dd;
    const int qd = (s > 0 ? s - 1 : 0) / RN;
    bool open = qd != q;
    q = qd;
    if (dd > 0 && (dd + ren_shift<KIND, DIR>()) % RN == 0) open = true;  // the main chain renormalised after position dd - 1
    if (open) ++seg;
  }
  return seg;
}

template <int KIND, int NL, int NH, int BLK, int VPL>
struct Lds {
  static constexpr int V = 256 * VPL, UP = 64 * NL;
  static constexpr int ES = UP + 4;      // E row: y[UP] (exp(x_label - rowmax)), then e_blank
  static constexpr int RS = 2 * UP + 8;  // R row (recompute chain): its NATIVE state -- pairs (c, o) per slot, tail (cx, kx); the main
                                         // chain shifts it into its own slot order when it reads it.  S row (main chain, in place):
                                         // per lane 2 NL floats of posterior parts (see run_main)
  static constexpr int NG = Cad<BLK, NL>::NG;
  static constexpr int NW = 4 + 2 * NH;
  float E[2][3][BLK][ES];   // [side][block % 3]
  float R[2][3][BLK][RS];   // [side][block % 3]
  int kg[2][3][NG][64];     // per-lane exponents of the R rows, one set per renormalisation group
  float kl[2][3][Cad<BLK, NL>::NSEG][64];  // per-lane posterior scale of the S rows' raw parts, one per segment (kl_segment)
  float xcopy[2 * NH][V + 4];
  float xcopy_r[2][V + 4];  // row copies of the recompute waves (E stage of phase 1)
  float bins[2 * NH][V + 4];
  float dump[NW][64];
  double l2s[NW];           // per worker: sum over its phase-1 frames of log2 sum_k exp(x_k - rowmax)
  int p1_prog[2][8];        // phase 1: blocks of side s whose E rows worker w has written (its own word)
  int p1_cons[2];           // phase 1: blocks of side s whose E rows the main chain has read into registers
  int flag;                 // OR of D1..D5 over the wavefronts
  int feasible;             // 1: phase 2 runs
  int lp_int;               // posterior scale: 2^-lp_int * cf = 1 / (P in mantissa units)
  float cf;
  float lossval;            // loss[b] as written at the meeting point and whether it has been added to the running sum there
  int added;                // (ctc_amd_loss_grad_sum; an utterance flagged later takes it back at the end of the kernel)
};

// Block geometry shared by every wavefront of the workgroup (identical to ctc_fused5.hip).
template <int BLK>
struct Geo {
  int len, G, tmb, tm, NB;
  __device__ __forceinline__ void init(int len_) {
    len = len_;
    G = (len + BLK - 1) / BLK;
    tmb = G / 2;
    tm = tmb * BLK;
    NB = G - tmb;
  }
  __device__ __forceinline__ int nvof(int g) const { int r = len - BLK * g; return r < BLK ? r : BLK; }
  __device__ __forceinline__ int nblocks(int phase, int side) const { return (phase == 1) == (side == 0) ? tmb : G - tmb; }
  __device__ __forceinline__ int absblock(int phase, int side, int j) const {
    if (phase == 1) return side == 0 ? j : G - 1 - j;
    return side == 0 ? tmb + j : tmb - 1 - j;
  }
  __device__ __forceinline__ int frame(int side, int g, int d) const { return side == 0 ? BLK * g + d : BLK * g + nvof(g) - 1 - d; }
  // checkpoint slot of lattice time t (multiples of BLK, and `len`): distinct per direction
  __device__ __forceinline__ int slot(int t) const { return (t + BLK - 1) / BLK; }
};

// NL consecutive floats (or pairs) of this lane in an LDS / HBM row
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

// per-frame emissions, linear: y[j] = exp(x[label[i]] - rowmax) (0 beyond label_length), bl = exp(x[blank] - rowmax)
template <int NL>
struct Emis {
  float y[NL];
  float bl;
};
template <int NL, class LDt>
__device__ __forceinline__ void read_E(const float *row, int lane, Emis<NL> &e) {
  ld_slots<NL>(row + lane * NL, e.y);
  e.bl = row[LDt::UP];  // same address in every lane: LDS broadcast
}

// one lattice row of the OTHER direction as that direction's chain holds it (R row): c[j] / o[j] per slot, boundary state cx
// with its exponent kx.  Pair layout for both kinds (simplified: second element unused): the R data of a lane then occupies
// exactly the 2 NL floats its S row entry overwrites in place.  With a packed simplified row, lane L's S entry would
// overlap the R data of lanes 2L and 2L+1 -- and nothing orders one lane's store against ANOTHER lane's earlier load
// (the compiler hoisted a piece of the store above the load; per thread the two never alias).
template <int KIND, int NL>
struct RRow {
  float c[NL], o[NL];
  float cx;
  int kx;
};
template <int KIND, int NL, class LDt>
__device__ __forceinline__ void read_R(const float *row, int lane, RRow<KIND, NL> &r) {
  if constexpr (KIND == 0 && NL == 2 && CTC_F6_PACKED != 0) {  // (c0, c1, o0, o1): both halves are register pairs for the packed products
    const float4 t = *reinterpret_cast<const float4 *>(row + 4 * lane);
    r.c[0] = t.x; r.c[1] = t.y; r.o[0] = t.z; r.o[1] = t.w;
  } else ld_pairs<NL>(row + 2 * lane * NL, r.c, r.o);
  const float2 t = *reinterpret_cast<const float2 *>(row + 2 * LDt::UP);
  r.cx = t.x; r.kx = __float_as_int(t.y);
}
template <int KIND, int NL, class LDt>
__device__ __forceinline__ void write_R(float *row, float *dump, int lane, const float (&c)[NL], const float (&o)[NL], float cx, int kx) {
  if constexpr (KIND == 0 && NL == 2 && CTC_F6_PACKED != 0) *reinterpret_cast<float4 *>(row + 4 * lane) = make_float4(c[0], c[1], o[0], o[1]);
  else st_pairs<NL>(row + 2 * lane * NL, c, o);
  float *tq = (lane == 0) ? row + 2 * LDt::UP : dump + (lane & 31) * 2;  // lanes > 0 write a sink
  *reinterpret_cast<float2 *>(tq) = make_float2(cx, __int_as_float(kx));
}

// ------------------------------------------------------------------------------------------------
// The lattice state of one direction: mantissas + one exponent per lane.  Slot i = lane*NL + j is label position i.
//   classic    A (DIR 0): c[j] = closed(l=i+1), o[j] = open(l=i+1), cx = closed(l=0)
//              B (DIR 1): c[j] = closed(l=i),   o[j] = open(l=i+1), cx = closed(l=UP)
//   simplified A: c[j] = a(l=i+1), cx = a(l=0);   B: c[j] = b(l=i), cx = b(l=UP)
// true value = mantissa * 2^k (lanes) / 2^kx (cx).  dk = (exponent of the upstream neighbour) - k: what the one value a
// lane receives per step has to be shifted by (upstream = previous lane for A, next lane for B; cx for the first / last).
// ------------------------------------------------------------------------------------------------
template <int KIND, int NL, int DIR>
struct Chain {
  float c[NL], o[NL], cx;
  int k, kx, dk;
  bool norep[NL], norep_next[NL];
  int flag;
  static constexpr bool PACKED = (CTC_F6_PACKED != 0) && KIND == 0 && NL == 2;
  float nrf[NL];  // PACKED: 1.0 where the repeat rule lets the diagonal pass (norep_next for A, norep for B), else 0.0
  float sc, scb;  // PACKED: 2^dk as a float (0 below 2^-126: what v_ldexp_f32 would flush), and the same on the boundary lane only

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
      c[j] = 0.f;
      o[j] = 0.f;
    }
    cx = 0.f; k = DEAD; kx = DEAD; dk = 0; flag = 0; sc = 1.f;
    scb = (lane == (DIR == 0 ? 0 : 63)) ? 1.f : 0.f;
    boundary = lane == (DIR == 0 ? 0 : 63);
    relevant = lane * NL <= ll;  // the lane holds a label position that can carry mass (lanes beyond the label stay empty for good)
  }

  // starting state: alpha[0] = delta(closed(l=0)) / beta[len] = delta(closed(l=ll)) + delta(open(l=ll))
  template <int LV>
  __device__ __forceinline__ void start(int lane, int ll, int UP) {
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
    renorm<LV>();
    flag = 0;
  }

  // one lattice step (the recursions of Side::step in ctc_fused_common.h with log-sum-exp -> +, + -> *)
  __device__ __forceinline__ void step(const Emis<NL> &e) {
    const float bl = e.bl;
    if constexpr (PACKED && DIR == 0) {
      // the generic recursion below, two positions per instruction; bit-identical results (x = c + nr o is one rounding like c + o,
      // the inflow x_up 2^dk is exact, its sum with o one rounding -- as v_ldexp_f32 + v_add_f32 gave)
      const f2v C = {c[0], c[1]}, O = {o[0], o[1]}, Y = {e.y[0], e.y[1]}, NR = {nrf[0], nrf[1]};
      const f2v X = __builtin_elementwise_fma(O, NR, C);
      const f2v M = C + O;
      float olo = __builtin_fmaf(cx, scb, O.x);    // lane 0: closed(l = 0) flows in (exponent kx, dk = kx - k there)
      fmac_from_upstream<0>(olo, X.y, sc);
      f2v OS = {olo, O.y + X.x};
      OS = Y * OS;
      const f2v CN = M * bl;
      o[0] = OS.x; o[1] = OS.y; c[0] = CN.x; c[1] = CN.y;
      cx *= bl;
    } else if constexpr (PACKED && DIR == 1) {
      const f2v C = {c[0], c[1]}, O = {o[0], o[1]}, Y = {e.y[0], e.y[1]}, NR = {nrf[0], nrf[1]};
      const f2v H = C * bl;
      const f2v EE = Y * O;
      const f2v PN = H + EE;
      const f2v X = __builtin_elementwise_fma(EE, NR, H);
      cx *= bl;
      float ohi = __builtin_fmaf(cx, scb, EE.y);   // lane 63: closed(l = UP) flows in
      fmac_from_upstream<1>(ohi, X.x, sc);
      o[0] = EE.x + X.y; o[1] = ohi; c[0] = PN.x; c[1] = PN.y;
    } else if constexpr (KIND == 0 && DIR == 0) {
      float m[NL], x[NL];
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        m[j] = c[j] + o[j];
        x[j] = norep_next[j] ? m[j] : c[j];
      }
      const float xin0 = ldexp_f(from_prev_lane(x[NL - 1], cx), dk);
#pragma unroll
      for (int j = NL - 1; j >= 0; --j) {
        const float xin = (j == 0) ? xin0 : x[j - 1];
        o[j] = e.y[j] * (o[j] + xin);
        c[j] = bl * m[j];
      }
      cx *= bl;
    } else if constexpr (KIND == 0 && DIR == 1) {
      float h[NL], ee[NL], pn[NL], x[NL];
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        h[j] = bl * c[j];
        ee[j] = e.y[j] * o[j];
        pn[j] = h[j] + ee[j];
        x[j] = norep[j] ? pn[j] : h[j];
      }
      cx *= bl;
      const float xinl = ldexp_f(from_next_lane(x[0], cx), dk);
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        const float xin = (j == NL - 1) ? xinl : x[j + 1];
        o[j] = xin + ee[j];
        c[j] = pn[j];
      }
    } else if constexpr (KIND == 1 && DIR == 0) {
      const float pin0 = ldexp_f(from_prev_lane(c[NL - 1], cx), dk);
#pragma unroll
      for (int j = NL - 1; j >= 0; --j) {
        const float pin = (j == 0) ? pin0 : c[j - 1];
        c[j] = bl * c[j] + e.y[j] * pin;
      }
      cx *= bl;
    } else {
      const float nin = ldexp_f(from_next_lane(c[0], cx), dk);
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        const float nx = (j == NL - 1) ? nin : c[j + 1];
        c[j] = bl * c[j] + e.y[j] * nx;
      }
      cx *= bl;
    }
  }

  // per-lane renormalisation: k <- exponent of the lane maximum (lanes without mass adopt the upstream exponent - GAP so
  // that what flows in during the next period is representable), cx to its own exponent, dk refreshed
  // `checked` (wave-uniform; loss-only calls, which have no posterior mass to check -- D6 -- against): D9, the exact form of "the format
  // lost something".  With needed emissions > 0 (D2) a lattice value that is nonzero stays nonzero, so (a) a mantissa that WAS
  // nonzero at the previous renormalisation and is zero now has been flushed, and (b) a nonzero mantissa below 2^-RANGE_MIN of its
  // lane's exponent -- before or after this renormalisation's shift -- is about to lose bits to gradual underflow, or cannot take a
  // small inflow any more.  Values only decay between two renormalisations unless something larger flows in, so looking here,
  // every RN frames, misses nothing.  (Until r04 loss-only calls relied on D3 / D4 -- lane maxima only -- and on D7, a sharpness
  // heuristic that sent every utterance of a trained model to the log domain.)
  template <int LV>
  __device__ __forceinline__ void renorm(bool checked = false) {
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
    // one level always (a lane far below its upstream neighbour is lifted to its exponent - GAP); the further levels only
    // serve lanes without mass, which need an exponent before the front reaches them (wave-uniform branch)
    {
      const int nb = (DIR == 0) ? from_prev_lane_i(kn, ex) : from_next_lane_i(kn, ex);
      kn = imax(kn, nb - (live ? imax(GAP_LIVE, LV == 1 ? GAP_WIDE : GAP) : (LV == 1 ? GAP_WIDE : GAP)));
    }
    // ALL levels for every lane, with or without mass (until r04 the further levels ran only while some lane of the wavefront was
    // empty): a STEEP profile of live lanes -- each 2^-100 below its upstream neighbour: sharp logits -- kept, after the one level,
    // exponents 2^100 apart two lanes down (each lane had been lifted against its neighbour's exponent BEFORE that neighbour's own
    // lift), and when the bulk crossed two lanes within a period the inflow arrived scaled by 2^100: mantissas of 2^60 .. 2^127,
    // inf at the meeting point (D1).  19 of 256 N(0, 3^2) utterances at the north-star shape were in that state at the meeting
    // point and 4 overflowed (tests/tools/linear_model.py); with every level applied dk <= GAP holds for every lane.
    {
#pragma unroll
      for (int lv = 1; lv < LV; ++lv) {
        const int nb = (DIR == 0) ? from_prev_lane_i(kn, ex) : from_next_lane_i(kn, ex);
        kn = imax(kn, nb - GAP);
      }
    }
    kn = imax(kn, DEAD);
    const int d = k - kn;
    // D3: own live values crushed by a much larger inflow scale; D4: decayed by more than 2^-DECAY_MAX, or live -> zero
    // (D3 only for a lane that has had mass for a few periods: at the lattice front the first thin paths of a lane are
    // legitimately swamped when the bulk arrives, ~1 in 256 benign utterances)
    age = (live && alive) ? age + 1 : 0;
    flag |= (live && age >= 3 && d < -DOWN_MAX ? 4 : 0) | (live && fe < -DECAY_MAX ? 8 : 0) | (!live && alive ? 16 : 0);
#ifdef CTC_F6_DEBUG
    ++cnt;
    if (live && d < -DOWN_MAX && dbg0 == 0) { dbg0 = cnt; dbg1 = d; dbg2 = fe; }
    mlast = m;
#endif
    if (checked) {
      // min over the lane's NONZERO mantissas, before and after the shift ((bits - 1) as unsigned: zero becomes the largest value),
      // and the pattern of zeros against the previous renormalisation's
      unsigned mn = 0xFFFFFFFFu, zeros = 0u;
      const int dneg = imin(d, 0);
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        mn = umin(mn, __float_as_uint(ldexp_f(c[j], dneg)) - 1u);
        zeros |= (c[j] == 0.f ? 1u : 0u) << (2 * j);
        if constexpr (KIND == 0) {
          mn = umin(mn, __float_as_uint(ldexp_f(o[j], dneg)) - 1u);
          zeros |= (o[j] == 0.f ? 2u : 0u) << (2 * j);
        }
      }
      flag |= ((mn < RANGE_MIN_BITS) || (zeros & ~zprev) != 0u) ? D9_RANGE : 0;
      zprev = zeros;
    }
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      c[j] = ldexp_f(c[j], d);
      if constexpr (KIND == 0) o[j] = ldexp_f(o[j], d);
    }
    k = kn;
    cx = ldexp_f(cx, kx - ex);
    kx = ex;
    dk = ((DIR == 0) ? from_prev_lane_i(k, kx) : from_next_lane_i(k, kx)) - k;
    set_scale();
    alive = live;
  }
  unsigned zprev = 0xFFFFFFFFu;  // D9: which of the lane's mantissas were zero at the previous renormalisation (everything, at the start)
  bool boundary = false;
  // dk as the factor the packed step multiplies by (after every change of dk)
  __device__ __forceinline__ void set_scale() {
    if constexpr (PACKED) {
      sc = (dk < -126) ? 0.f : ldexp_f(1.f, imin(dk, 127));
      scb = boundary ? sc : 0.f;
    }
  }
  bool alive = false;  // the lane had mass at its last renormalisation
  int age = 0;         // consecutive renormalisations with mass
  bool relevant = true;
#ifdef CTC_F6_DEBUG
  int cnt = 0, dbg0 = 0, dbg1 = 0, dbg2 = 0;
  float mlast = 0.f;
#endif
  // number of label positions 1 .. ll-1 that repeat their predecessor (classic: each costs one more frame); wave-uniform
  __device__ __forceinline__ int repeats(int ll, int lane) const {
    int n = 0;
#pragma unroll
    for (int j = 0; j < NL; ++j) n += __builtin_popcountll(__builtin_amdgcn_ballot_w64(!norep[j] && lane * NL + j < ll));
    return n;
  }
  // OR of the lanes' flags (wave-uniform)
  __device__ __forceinline__ int flag_or() const {
    int f = 0;
#pragma unroll
    for (int bit = 4; bit <= 16; bit <<= 1) f |= (__builtin_amdgcn_ballot_w64((flag & bit) != 0) != 0) ? bit : 0;
    f |= (__builtin_amdgcn_ballot_w64((flag & D9_RANGE) != 0) != 0) ? D9_RANGE : 0;
    return f;
  }

};

// checkpoint row in HBM: the chain's NATIVE state.  rows: [slot][SRS] floats (pairs (c, o) for classic, c for simplified,
// tail = (cx, kx)); kexp: [slot][64] per-lane exponents.
template <int KIND, int NL, int DIR>
__device__ __forceinline__ void spill(const Chain<KIND, NL, DIR> &S, float *__restrict__ rows, int *__restrict__ kexp,
                                      int slot, int SRS, int UP, int lane) {
  float *row = rows + (long)slot * SRS;
  if constexpr (KIND == 0) st_pairs<NL>(row + 2 * lane * NL, S.c, S.o);
  else st_slots<NL>(row + lane * NL, S.c);
  if (lane == 0) *reinterpret_cast<float2 *>(row + (KIND == 0 ? 2 : 1) * UP) = make_float2(S.cx, __int_as_float(S.kx));
  kexp[slot * 64 + lane] = S.k;
}
template <int KIND, int NL>
struct CkRow {
  float c[NL], o[NL], cx;
  int k, kx;
};
template <int KIND, int NL>
__device__ __forceinline__ void load_ck(CkRow<KIND, NL> &r, const float *__restrict__ rows, const int *__restrict__ kexp, int slot,
                                        int SRS, int UP, int lane) {
  const float *row = rows + (long)slot * SRS;
  if constexpr (KIND == 0) ld_pairs<NL>(row + 2 * lane * NL, r.c, r.o);
  else {
    ld_slots<NL>(row + lane * NL, r.c);
#pragma unroll
    for (int j = 0; j < NL; ++j) r.o[j] = 0.f;
  }
  const float2 t = *reinterpret_cast<const float2 *>(row + (KIND == 0 ? 2 : 1) * UP);
  r.cx = t.x; r.kx = __float_as_int(t.y);
  r.k = kexp[slot * 64 + lane];
}
template <int KIND, int NL, int DIR>
__device__ __forceinline__ void restore(Chain<KIND, NL, DIR> &S, const CkRow<KIND, NL> &r) {
#pragma unroll
  for (int j = 0; j < NL; ++j) { S.c[j] = r.c[j]; S.o[j] = r.o[j]; }
  S.cx = r.cx; S.k = r.k; S.kx = r.kx;
  S.dk = ((DIR == 0) ? from_prev_lane_i(S.k, S.kx) : from_next_lane_i(S.k, S.kx)) - S.k;
  S.set_scale();
  float m = 0.f;
#pragma unroll
  for (int j = 0; j < NL; ++j) m = fmaxf(m, fmaxf(r.c[j], r.o[j]));
  S.alive = m > 0.f;  // (a lane that only adopted its neighbour's exponent has no mass yet)
}

// ------------------------------------------------------------------------------------------------
// Logits rows: loads / stores / softmax statistics / emission gather.  Wraps Side<> of ctc_fused_common.h for the
// format-specific row accesses (contiguous or strided float32, bfloat16, unaligned rows).
// ------------------------------------------------------------------------------------------------
template <int KIND, int NL, int VPL, int XT>
struct Rows {
  static constexpr int V = 256 * VPL;
  Side<KIND, NL, VPL, 0, true, XT> io;  // load_x / store_g / zero_rows only
  int tokoff[NL];                       // byte offset of label[i] in the LDS copy of the row (pad slot beyond label_length)
  float mb[4 * VPL];                    // 1.0 at this lane's element that is the blank column
  bool valid[NL];                       // slot i < label_length
  int scat[NL];                         // byte offset (from bins) of the slot's posterior bin; slots beyond label_length: a word of their own
  float *xs, *bins;
  int lane, blank;
  float dl;

  __device__ __forceinline__ void init(const Problem &p, int b, int lane_, int ll, const float *d_loss, float *grad) {
    lane = lane_; blank = p.blank;
    io.lane = lane_;
    if constexpr (XT != 2) {
      io.xbase = p.logits + (long)b * p.xsb;
      io.gbase = grad + (long)b * p.gsb;
    } else {
      io.xbase = reinterpret_cast<const float *>(reinterpret_cast<const unsigned short *>(p.logits) + (long)b * p.xsb);
      io.gbase = reinterpret_cast<float *>(reinterpret_cast<unsigned short *>(grad) + (long)b * p.gsb);
    }
    io.xst = p.xst; io.gst = p.gst; io.Vr = p.V;
    dl = d_loss ? d_loss[b] : 1.0f;
    const int32_t *lab = p.labels + (long)b * p.label_stride;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const int i = lane * NL + j;
      const int tk = (i < ll) ? ((i < p.label_stride) ? lab[i] : p.blank) : -1;
      valid[j] = i < ll;
      tokoff[j] = 4 * ((tk >= 0 && tk < p.V && tk < V && tk != p.blank) ? tk : V);
    }
#pragma unroll
    for (int e = 0; e < 4 * VPL; ++e) mb[e] = (256 * (e / 4) + lane * 4 + (e & 3) == p.blank) ? 1.f : 0.f;
  }

  // LDS rows of this wavefront.  The posterior scatter is branch-free: a slot beyond label_length adds into word `lane` of the
  // gather copy, which is dead between two E-stage frames (every frame rewrites it before reading) -- a common pad slot would
  // serialise the adds of a ragged batch, an `if` costs two exec-mask branches per frame.
  __device__ __forceinline__ void set_lds(float *xs_, float *bins_) {
    xs = xs_; bins = bins_;
    const int own = (int)(reinterpret_cast<char *>(xs_) - reinterpret_cast<char *>(bins_)) + 4 * lane;
#pragma unroll
    for (int j = 0; j < NL; ++j) scat[j] = valid[j] ? tokoff[j] : own;
  }

  // exp(x - rowmax) of this lane's elements from the recorded statistic mxl = rowmax * log2(e)
  __device__ __forceinline__ void expo(const float4 (&xr)[VPL], float mxl, float4 (&ev)[VPL]) const {
#pragma unroll
    for (int q = 0; q < VPL; ++q) {
      ev[q] = make_float4(fexp2(fmaf(xr[q].x, LOG2E, -mxl)), fexp2(fmaf(xr[q].y, LOG2E, -mxl)),
                          fexp2(fmaf(xr[q].z, LOG2E, -mxl)), fexp2(fmaf(xr[q].w, LOG2E, -mxl)));
    }
  }
  // emission gather through an LDS copy of the exponentiated row (base_loss.py:328-344, 365-371)
  __device__ __forceinline__ void gather(const float4 (&ev)[VPL], Emis<NL> &e) const {
#pragma unroll
    for (int q = 0; q < VPL; ++q)
      *reinterpret_cast<float4 *>(xs + 256 * q + lane * 4) = make_float4(ev[q].x, ev[q].y, ev[q].z, ev[q].w);
    const char *bb = reinterpret_cast<const char *>(xs);
#pragma unroll
    for (int j = 0; j < NL; ++j) e.y[j] = *reinterpret_cast<const float *>(bb + tokoff[j]);
    e.bl = xs[blank];
  }

  // Q frames at once (phase 1): row maximum and sum of exponentials by batched DPP reductions (ctc_dpp_batch.h), then the
  // gathers.  Outputs per frame: emissions, mxl = rowmax * log2 e, inv = 1 / sum exp, l2s = log2 sum exp.
  // Q frames at once (phase 1): row maximum and sum of exponentials by batched reductions, then the gathers.  Outputs per frame:
  // emissions, mxl = rowmax * log2 e, sm = sum exp (the caller takes ONE reciprocal per block, on the lanes that keep the
  // statistics); and l2s_all = log2 of the PRODUCT of the Q sums (one logarithm for the Q frames: each sum lies in [1, V], so
  // the product of four stays far inside float32) -- transcendentals are quarter rate, 6 of them per frame were 28 % of the
  // E stage's issue time.
  template <int Q>
  __device__ __forceinline__ void emit_n(const float4 (&xr)[Q][VPL], Emis<NL> (&e)[Q], float (&mxl)[Q], float (&sm)[Q], float &l2s_all) const {
    float m[Q];
#pragma unroll
    for (int f = 0; f < Q; ++f) {
      m[f] = vmax_raw(vmax3_raw(xr[f][0].x, xr[f][0].y, xr[f][0].z), xr[f][0].w);
#pragma unroll
      for (int q = 1; q < VPL; ++q) m[f] = vmax3_raw(vmax3_raw(m[f], xr[f][q].x, xr[f][q].y), xr[f][q].z, xr[f][q].w);
    }
    constexpr bool SWAP = (Q >= 1 && Q <= 4);  // (ctc_swap_reduce.h: all Q values through one register; one rides as two, three as four)
    constexpr int QP = SWAP ? (Q == 3 ? 4 : Q == 1 ? 2 : Q) : 2;
    float mall = 0.f;
    if constexpr (SWAP) {
      float mp[QP];
#pragma unroll
      for (int f = 0; f < QP; ++f) mp[f] = m[f < Q ? f : Q - 1];
      mall = swap_reduce<QP, true>(mp);
    } else dpp_max_n<Q>(m);
    float4 ev[Q][VPL];
    float part[Q];
#pragma unroll
    for (int f = 0; f < Q; ++f) {
      float mx;
      if constexpr (SWAP) mx = readlane_f(mall, SwapLanes<QP>::lane(f));
      else mx = readlane_f(m[f], 63);
      mx = (mx == -INFINITY) ? 0.f : mx;
      mxl[f] = mx * LOG2E;
      expo(xr[f], mxl[f], ev[f]);
      part[f] = (ev[f][0].x + ev[f][0].y) + (ev[f][0].z + ev[f][0].w);
#pragma unroll
      for (int q = 1; q < VPL; ++q) part[f] += (ev[f][q].x + ev[f][q].y) + (ev[f][q].z + ev[f][q].w);
    }
    float sall = 0.f;
    if constexpr (SWAP) {
      float pp[QP];
#pragma unroll
      for (int f = 0; f < QP; ++f) pp[f] = f < Q ? part[f] : 0.f;
      sall = swap_reduce<QP, false>(pp);
    } else dpp_sum_n<Q>(part);
    float prod = 1.f;
#pragma unroll
    for (int f = 0; f < Q; ++f) {
      if constexpr (SWAP) sm[f] = readlane_f(sall, SwapLanes<QP>::lane(f));
      else sm[f] = readlane_f(part[f], 63);
      prod = (f == 0) ? sm[0] : prod * sm[f];
    }
    l2s_all = flog2(prod);
#pragma unroll
    for (int f = 0; f < Q; ++f) gather(ev[f], e[f]);  // LDS operations of a wavefront execute in program order: one copy serves all
  }

  // gradient row of frame t from the S row of the main chain: qb = blank posterior of the row, qt[j] = token posterior of slot j,
  // both in units of 2^-30; ev = exp(x - rowmax) of this lane's elements, inv = 1 / sum exp.  Two halves: the LDS half (bins
  // cleared, posteriors added, bins read back -- LDS operations of a wavefront execute in order, so the halves of several frames
  // can be issued back to back through the ONE bin row and their round trips overlap) and the arithmetic + store half.
  __device__ __forceinline__ void scatter(const float (&qt)[NL], uint4 (&pu)[VPL]) const {
#pragma unroll
    for (int q = 0; q < VPL; ++q) *reinterpret_cast<uint4 *>(bins + 256 * q + lane * 4) = make_uint4(0u, 0u, 0u, 0u);  // (same type as the atomics and the read: float stores may be reordered against them)
    char *bb = reinterpret_cast<char *>(bins);
#pragma unroll
    for (int j = 0; j < NL; ++j) atomicAdd(reinterpret_cast<unsigned *>(bb + scat[j]), valid[j] ? (unsigned)(qt[j] + 0.5f) : 0u);
    wave_lds_fence();  // the bins read below were written by other lanes
#pragma unroll
    for (int q = 0; q < VPL; ++q) pu[q] = *reinterpret_cast<const uint4 *>(bins + 256 * q + lane * 4);
    wave_lds_fence();  // (and the next frame's clear stays behind this read)
  }
  __device__ __forceinline__ void grad_out(int t, float qb, const uint4 (&pu)[VPL], const float4 (&ev)[VPL], float inv) const {
    const float c1 = -dl * 9.31322574615478515625e-10f;
    const float c2 = dl * inv;
#pragma unroll
    for (int q = 0; q < VPL; ++q) {
      // (explicit fused multiply-adds: every instantiation -- storage format, batched or per-frame G stage -- rounds alike, which
      // the format tests check bit for bit)
      const float4 pq = make_float4(__builtin_fmaf(mb[4 * q], qb, (float)pu[q].x), __builtin_fmaf(mb[4 * q + 1], qb, (float)pu[q].y),
                                    __builtin_fmaf(mb[4 * q + 2], qb, (float)pu[q].z), __builtin_fmaf(mb[4 * q + 3], qb, (float)pu[q].w));
      const float4 sv = make_float4(c2 * ev[q].x, c2 * ev[q].y, c2 * ev[q].z, c2 * ev[q].w);
      io.store_g(t, q, make_float4(__builtin_fmaf(pq.x, c1, sv.x), __builtin_fmaf(pq.y, c1, sv.y), __builtin_fmaf(pq.z, c1, sv.z), __builtin_fmaf(pq.w, c1, sv.w)));
    }
  }
  __device__ __forceinline__ void grad_row(int t, float qb, const float (&qt)[NL], const float4 (&ev)[VPL], float inv) const {
    uint4 pu[VPL];
    scatter(qt, pu);
    grad_out(t, qb, pu, ev, inv);
  }
};

// ------------------------------------------------------------------------------------------------
// E stage of phase 1, shared by the helpers and the recompute wavefronts (which have no lattice work before the meeting
// point): positions P0 .. P0+NQ-1 of every block of side SIDE.  Records (mxl, inv) per frame for phase 2, accumulates
// log2 sum exp of its frames in double and tracks the smallest needed emission (D2).
// ------------------------------------------------------------------------------------------------
#ifndef CTC_F6_PFD
#define CTC_F6_PFD 2
#endif
#ifndef CTC_F6_ONLY  // experiment: which roles work in phase 2 (1 main, 2 recompute, 4 helper E stage, 8 helper G stage) and in phase 1 (16 main, 32 E stage); others only keep the barriers
#define CTC_F6_ONLY 63
#endif
#ifndef CTC_F6_NH12   // helpers per side of the 12-frame-block instantiations (V <= 256, U <= 128).  6 = sixteen wavefronts, two frames per
#define CTC_F6_NH12 4 // helper and block, was built and measured in r03: 134 against 127 us at B = 64, 161 against 151 at B = 256 (same box) --
#endif                // the main chains lose more to two extra wavefronts on their SIMDs than the helpers gain.  Diagnostic builds only.
#ifndef CTC_F6_PRIO1  // experiment: issue priority of the main chains before the meeting point / of the helpers of side B
#define CTC_F6_PRIO1 3
#endif
#ifndef CTC_F6_HPRIO_B
#define CTC_F6_HPRIO_B 0
#endif
#ifndef CTC_F6_X
#define CTC_F6_X 2
#endif
#ifndef CTC_F6_Y
#define CTC_F6_Y 2
#endif
template <int BLK, int NH, int NL>
struct P1Split {
  // (NH = 1, the 3-frame blocks of the 8-positions-per-lane variant: two frames for the helper, one for the recompute wavefront;
  // NH = 6, sixteen wavefronts: two frames for every helper, none for the recompute wavefronts)
  // (NH = 3 for the six-frame blocks of the four-positions-per-lane variant -- 2, 2, 1 frames for the helpers, 1 for the recompute
  // wavefront -- was built and measured in r04: 227 us against 211 at U = 256: those shapes are bound by their chains, and more
  // helpers only take issue slots from them)
  static constexpr int X = NH == 6 ? BLK / 6 : NH == 4 ? CTC_F6_X : NH == 2 ? BLK / 3 : 2, Y = NH == 6 ? BLK / 6 : NH == 4 ? CTC_F6_Y : NH == 2 ? BLK / 3 : 0;
  static constexpr int R = NH == 6 ? 0 : NH == 4 ? BLK - 2 * X - 2 * Y : NH == 2 ? BLK - X - Y : BLK - X;
  static_assert(NH == 6 || NH == 4 || NH == 2 || NH == 1, "helpers per side");
  static_assert(X >= 0 && Y >= 0 && R >= 0 && X <= 6 && Y <= 6 && R <= 6, "phase-1 split: at most 6 frames per worker");
  static constexpr int count(int worker) {
    if (NH == 6) return worker < 6 ? X : R;
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

template <int KIND, int NL, int NH, int BLK, int VPL, int XT, int SIDE, int P0, int NQ>
__device__ __forceinline__ void estage1(const Rows<KIND, NL, VPL, XT> &S, Lds<KIND, NL, NH, BLK, VPL> &lds, const Geo<BLK> &geo,
                                        float2 *__restrict__ stats, float2 *__restrict__ stats_sink, float *dump, int lane, int wave
#ifdef CTC_F6_STAMPS
                                        , Stamps &st_
#endif
                                        ) {
  using LD = Lds<KIND, NL, NH, BLK, VPL>;
  const int len = geo.len;
  const int nb = geo.nblocks(1, SIDE);
  auto write_E = [&](float *row, const Emis<NL> &e) __attribute__((always_inline)) {
    st_slots<NL>(row + lane * NL, e.y);
    float *tq = (lane == 0) ? row + LD::UP : dump + lane;
    *tq = e.bl;
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
  // logits rows are loaded PFD blocks ahead of their use (a block lasts ~2 us, an HBM load under load about as long: one
  // block of look-ahead left the E stage waiting on memory for half of phase 1), in a ring of register sets addressed by
  // (block mod PFD) at COMPILE time -- the loop is unrolled by PFD
  constexpr int PFD = CTC_F6_PFD;
  float4 xb[PFD][NQA][VPL];
  static_for<0, PFD>([&](auto R) {
    static_for<0, NQA>([&](auto Q) {
      constexpr int q = decltype(Q)::value;
      static_for<0, VPL>([&](auto W) { xb[decltype(R)::value][q][decltype(W)::value] = make_float4(0.f, 0.f, 0.f, 0.f); });
      if (NQ > 0 && nb > 0) S.io.load_x(xb[decltype(R)::value][q], fr(decltype(R)::value, P0 + q));
    });
  });
  double acc = 0.0;
  float zmin[NL], zb = 1.0f;
#pragma unroll
  for (int j = 0; j < NL; ++j) zmin[j] = 1.0f;
  auto track = [&](const Emis<NL> &e) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NL; ++j) zmin[j] = vmin_raw(zmin[j], e.y[j]);
    zb = vmin_raw(zb, e.bl);
  };
  // FAST: steady state -- the block exists and is full, so the body has NO branch around a memory operation.  hipcc derives
  // `s_waitcnt vmcnt(N)` from the fewest memory operations any path can have issued after the load it waits for; with a
  // conditional store or load in the loop that is zero, every use of the ring drained ALL outstanding loads and the
  // look-ahead bought nothing (phase 1 ran at the latency of one HBM round trip per block).
  float2 *sink = stats_sink + (wave & 7) * 32 + (lane & 31);  // statistics of lanes that hold no frame go here
  const int widx = (wave < 4) ? NH : (wave - 4) % NH;         // this worker's progress word (helpers 0 .. NH-1, the recompute wavefront NH)
  bool sync_bad = false;
  auto body = [&](auto R, auto FASTt, int it) __attribute__((always_inline)) {
    constexpr int r = decltype(R)::value;  // = it mod PFD
    constexpr bool FAST = decltype(FASTt)::value;
    const int j = it;
    if ((CTC_F6_ONLY & 32) && (FAST || (NQ > 0 && j < nb))) {
      const int g = geo.absblock(1, SIDE, j);
      const int nv = FAST ? BLK : geo.nvof(g);
      float(*E)[LD::ES] = lds.E[SIDE][j % 3];
      if constexpr (CTC_F6_P1SYNC != 0) {  // slot j % 3 is free once the chain has block j - 3 in its registers
        F6_WAIT(if (j >= 3 && !wait_word_ge(&lds.p1_cons[SIDE], j - 2)) sync_bad = true);
      }
      float smx = 0.f, ssum = 1.f;  // lane d keeps the statistics of position d of the block (its sum; ONE reciprocal below)
      if (FAST || nv == BLK) {
        if constexpr (NQ > 0) {
          float4 xq[NQA][VPL];
          Emis<NL> e[NQA];
          float mxl[NQA], sm[NQA], l2s;
          static_for<0, NQA>([&](auto Q) {
            constexpr int q = decltype(Q)::value;
            static_for<0, VPL>([&](auto W) {
              constexpr int w = decltype(W)::value;
              xq[q][w] = make_float4(xb[r][q][w].x, xb[r][q][w].y, xb[r][q][w].z, xb[r][q][w].w);
            });
          });
          S.template emit_n<NQA>(xq, e, mxl, sm, l2s);
          acc += (double)l2s;
          static_for<0, NQA>([&](auto Q) {
            constexpr int q = decltype(Q)::value;
            write_E(E[P0 + q], e[q]);
            track(e[q]);
            smx = (lane == P0 + q) ? mxl[q] : smx;
            ssum = (lane == P0 + q) ? sm[q] : ssum;
          });
        }
      } else {
        for (int q = 0; q < NQ; ++q) {
          const int d = P0 + q;
          if (d < nv) {
            float4 xr[1][VPL];
            S.io.load_x(xr[0], geo.frame(SIDE, g, d));
            Emis<NL> e[1];
            float mxl[1], sm[1], l2s;
            S.template emit_n<1>(xr, e, mxl, sm, l2s);
            acc += (double)l2s;
            write_E(E[d], e[0]);
            track(e[0]);
            smx = (lane == d) ? mxl[0] : smx;
            ssum = (lane == d) ? sm[0] : ssum;
          }
        }
      }
      static_for<0, NQA>([&](auto Q) {
        constexpr int q = decltype(Q)::value;
        S.io.load_x(xb[r][q], fr(j + PFD, P0 + q));
      });
      const bool mine = lane >= P0 && lane < P0 + NQ && lane < nv;
      float2 *dst = mine ? stats + geo.frame(SIDE, g, mine ? lane : 0) : sink;  // unconditional store: no branch
      *dst = make_float2(smx, __builtin_amdgcn_rcpf(ssum));  // 1 / sum exp of this lane's frame
      if constexpr (CTC_F6_P1SYNC != 0) publish_word(&lds.p1_prog[SIDE][widx], dump, lane, j + 1);  // (after the E rows, in order)
    }
    if constexpr (CTC_F6_P1SYNC == 0) F6_BARRIER();
  };
  // the first PFD blocks through the general body (side B starts with the utterance's last block, the only one that can
  // be partial); then the steady state; then whatever is left of the NB + 1 iterations every wavefront makes
  int it0 = 0;
  static_for<0, PFD>([&](auto R) {
    if (decltype(R)::value <= geo.NB) body(R, std::false_type{}, decltype(R)::value);
  });
  it0 = PFD;
  if constexpr (NQ > 0) {
    // (one trip peeled: the loop is then entered only from a steady-state trip, so the wait counts hipcc derives at its
    // head are those of the steady state, not those of the general bodies before it)
    if (it0 + PFD <= nb) {
      static_for<0, PFD>([&](auto R) { body(R, std::true_type{}, it0 + decltype(R)::value); });
      it0 += PFD;
      for (; it0 + PFD <= nb; it0 += PFD) {
        static_for<0, PFD>([&](auto R) { body(R, std::true_type{}, it0 + decltype(R)::value); });
      }
    }
  }
  for (; it0 <= geo.NB; it0 += PFD) {
    static_for<0, PFD>([&](auto R) {
      if (it0 + decltype(R)::value <= geo.NB) body(R, std::false_type{}, it0 + decltype(R)::value);
    });
  }
  // D2: a needed emission below 2^-100 of its row maximum (also catches NaN: the comparison is false)
  bool bad = !(zb >= EMIS_MIN) || !(acc - acc == 0.0);  // (a NaN or +inf logit makes the row sum, hence acc, non-finite)
#pragma unroll
  for (int j = 0; j < NL; ++j) bad = bad || (S.valid[j] && !(zmin[j] >= EMIS_MIN));
  if (NQ > 0 && nb > 0 && __builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0) atomicOr(&lds.flag, 2);
  bool sharp = !(zb >= EMIS_SOFT);  // D7 (honoured by loss-only calls)
#pragma unroll
  for (int j = 0; j < NL; ++j) sharp = sharp || (S.valid[j] && !(zmin[j] >= EMIS_SOFT));
  if (NQ > 0 && nb > 0 && __builtin_amdgcn_ballot_w64(sharp) != 0 && lane == 0) atomicOr(&lds.flag, 128);
  if (lane == 0) lds.l2s[wave] = acc;
  if (sync_bad && lane == 0) atomicOr(&lds.flag, D8_SYNC);
}

// ------------------------------------------------------------------------------------------------
// main chain
// ------------------------------------------------------------------------------------------------
template <int KIND, int NL, int NH, int BLK, int VPL, int DIR>
__device__ __forceinline__ void run_main(const Problem &p, const Layout &L, float *__restrict__ alpha_ws,
                                         float *__restrict__ beta_ws, int *__restrict__ kexp_ws, double *__restrict__ logp_ws,
                                         float *__restrict__ loss, int *__restrict__ flag_ws, int2 *__restrict__ meet_ws,
                                         Lds<KIND, NL, NH, BLK, VPL> &lds, const Geo<BLK> &geo, bool want_grad, int b) {
  using LD = Lds<KIND, NL, NH, BLK, VPL>;
  using CD = Cad<BLK, NL>;
  constexpr int RN = CD::RN, LV = CD::LV;
  Chain<KIND, NL, DIR> S;
  const int lane = threadIdx.x & 63;
  const int T = p.T, UP = L.UP, SRS = L.SRS;
  const int len = geo.len;
  int ll = p.label_length[b] < 0 ? 0 : p.label_length[b];
  const bool shape_ok = (ll <= p.U);
  if (!shape_ok) ll = 0;
  const int nslot = L.nslot;  // checkpoint slots per direction
  float *own_rows = (DIR == 0 ? alpha_ws : beta_ws) + (long)b * L.rows_b * SRS;
  const float *oth_rows = (DIR == 0 ? beta_ws : alpha_ws) + (long)b * L.rows_b * SRS;
  int *own_k = kexp_ws + ((long)b * 2 + DIR) * nslot * 64;
  const int *oth_k = kexp_ws + ((long)b * 2 + (1 - DIR)) * nslot * 64;
  S.init_labels(p, b, lane, ll);
  S.template start<LV>(lane, ll, UP);
  float *dump = lds.dump[DIR];
  (void)dump;
  int *flag_ws_dbg = flag_ws;
  (void)flag_ws_dbg;
  F6_STAMP_DECL

  if (p.resume == 1) {
    // second call of a loss -> gradient pair: this chain continues from its own row at the meeting point
    CkRow<KIND, NL> r;
    load_ck<KIND, NL>(r, own_rows, own_k, geo.slot(geo.tm), SRS, UP, lane);
    restore<KIND, NL, DIR>(S, r);
  } else {
  // ================= phase 1: lattice steps, one checkpoint row per block =================
  {
    const int nb = geo.nblocks(1, DIR);
    using SPm = P1Split<BLK, NH, NL>;
    constexpr int NWK = NH + (SPm::count(NH) > 0 ? 1 : 0);  // E-stage workers of a side that hold frames
    bool sync_bad = false;
    for (int it = (CTC_F6_P1SYNC != 0 ? 1 : 0); it <= (CTC_F6_P1SYNC != 0 ? nb : geo.NB); ++it) {
      const int j = it - 1;
      if ((CTC_F6_ONLY & 16) && j >= 0 && j < nb) {
        const int g = geo.absblock(1, DIR, j);
        const int nv = geo.nvof(g);
        const float(*E)[LD::ES] = lds.E[DIR][j % 3];
        spill<KIND, NL, DIR>(S, own_rows, own_k, geo.slot(DIR == 0 ? BLK * g : BLK * g + nv), SRS, UP, lane);
        if constexpr (CTC_F6_P1SYNC != 0) {  // every worker of this side has written its rows of block j
          F6_WAIT(if (!wait_words_ge(lds.p1_prog[DIR], NWK, j + 1, lane)) sync_bad = true);
        }
        if (nv == BLK) {
          // the emission rows of the whole block go to registers first: the sequential chain never waits for an LDS round
          // trip (with the read next to its use every frame paid one, ~100 of its ~190 cycles)
          Emis<NL> eb[BLK];
          static_for<0, BLK>([&](auto D) { read_E<NL, LD>(E[decltype(D)::value], lane, eb[decltype(D)::value]); });
          if constexpr (CTC_F6_P1SYNC != 0) publish_word(&lds.p1_cons[DIR], dump, lane, j + 1);  // (behind the reads, in order)
          static_for<0, BLK>([&](auto D) {
            constexpr int d = decltype(D)::value;
            S.step(eb[d]);
            if ((d + 1) % RN == 0) S.template renorm<LV>(CTC_F6_D9_ON && !want_grad);
          });
        } else {
          for (int d = 0; d < nv; ++d) {
            Emis<NL> e;
            read_E<NL, LD>(E[d], lane, e);
            S.step(e);
            if ((d + 1) % RN == 0 || d == nv - 1) S.template renorm<LV>(CTC_F6_D9_ON && !want_grad);
          }
          if constexpr (CTC_F6_P1SYNC != 0) publish_word(&lds.p1_cons[DIR], dump, lane, j + 1);
        }
      }
      if constexpr (CTC_F6_P1SYNC == 0) F6_BARRIER();
    }
    if (sync_bad && lane == 0) atomicOr(&lds.flag, D8_SYNC);
  }
  spill<KIND, NL, DIR>(S, own_rows, own_k, geo.slot(geo.tm), SRS, UP, lane);  // alpha[tm] / beta[tm]: the meeting row
#ifdef CTC_F6_DEBUG
  {
    int *dbg = flag_ws + p.B + ((long)b * 2 + DIR) * 256 + 0;  // diagnostic builds: per lane (renorm count at death, k, last max)
    dbg[lane] = S.dbg0; dbg[64 + lane] = S.dbg1; dbg[128 + lane] = S.dbg2; dbg[192 + lane] = S.k;
  }
#endif
  {
    const int f = S.flag_or();  // D3 / D4 of phase 1
    if (f != 0 && lane == 0) atomicOr(&lds.flag, f);
  }

  // ================= meeting point =================
  __syncthreads();  // full drain: the checkpoint rows of both chains are visible to the workgroup
  if constexpr (DIR == 0) {
    // P = sum over states of alpha[tm] beta[tm]: B's native row shifted into A's slot order (closed parts come from the
    // next slot), every product with its own exponent, reduced with a common exponent EX
    CkRow<KIND, NL> r;
    load_ck<KIND, NL>(r, oth_rows, oth_k, geo.slot(geo.tm), SRS, UP, lane);
    const int kn = from_next_lane_i(r.k, r.kx);                  // exponent of the value shifted in from the next lane
    const float cn = from_next_lane(r.c[0], r.cx);
    // Every product as (mantissa product in [1/4, 1), sum of the two binary exponents): a plain product of two float32 mantissas
    // underflows below 2^-126 although its lane exponents put it at the top of the sum -- with sharp logits the state the best
    // path crosses tm in sits 2^-70 below its lane-mates in BOTH directions, every product of the wavefront flushed, and P came out
    // zero or short (r04: 4 of 256 N(0, 3^2) utterances at the north-star shape raised D1 for this alone).  Once per utterance.
    auto pm = [](float a, float b_) -> float { return __builtin_amdgcn_frexp_mantf(a) * __builtin_amdgcn_frexp_mantf(b_); };
    auto pe = [](float a, float b_) -> int { return (a > 0.f && b_ > 0.f) ? frexp_e(a) + frexp_e(b_) : DEAD; };
    float tmn[2 * NL]; int ten[2 * NL];  // the lane's aligned products (exponent S.k + r.k on top of their own)
    int nt = 0;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      if (j < NL - 1) { tmn[nt] = pm(S.c[j], r.c[j + 1]); ten[nt] = pe(S.c[j], r.c[j + 1]); ++nt; }
      if constexpr (KIND == 0) { tmn[nt] = pm(S.o[j], r.o[j]); ten[nt] = pe(S.o[j], r.o[j]); ++nt; }
    }
    int el = DEAD;
#pragma unroll
    for (int q = 0; q < 2 * NL; ++q) if (q < nt) el = imax(el, ten[q]);
    float t1 = 0.f;
#pragma unroll
    for (int q = 0; q < 2 * NL; ++q) if (q < nt && ten[q] > DEAD) t1 += ldexp_f(tmn[q], imax(ten[q] - el, -200));
    const float t2 = pm(S.c[NL - 1], cn);
    const int x2 = pe(S.c[NL - 1], cn);
    const float r00 = readlane_f(r.c[0], 0);
    const float t0 = (lane == 0) ? pm(S.cx, r00) : 0.f;
    const int x0 = (lane == 0) ? pe(S.cx, r00) : DEAD;
    const int k0 = readlane_i(r.k, 0);
    const int E1 = el + S.k + r.k, E2 = x2 + S.k + kn, E0 = x0 + S.kx + k0;  // (DEAD-based sums stay far below any live exponent)
    const int e1 = (t1 > 0.f && el > DEAD) ? frexp_e(t1) + E1 : DEAD;
    const int e2 = (x2 > DEAD) ? frexp_e(t2) + E2 : DEAD;
    const int e0 = (x0 > DEAD) ? frexp_e(t0) + E0 : DEAD;
    const int EX = (int)wave_max_dpp((float)imax(imax(e1, imax(e2, e0)), DEAD));  // |values| <= 2^24: exact in float32
    float s = 0.f;
    if (e1 > DEAD) s += ldexp_f(t1, imax(E1 - EX, -200));
    if (e2 > DEAD) s += ldexp_f(t2, imax(E2 - EX, -200));
    if (e0 > DEAD) s += ldexp_f(t0, imax(E0 - EX, -200));
    s = wave_sum_dpp(s);
    // D1: P == 0 / inf / NaN (or nothing alive at all)
    const bool okP = shape_ok && EX > DEAD / 2 && s > 0.f && s < 3.0e38f;
    double sl2 = 0.0;
    for (int w = 2; w < LD::NW; ++w) sl2 += lds.l2s[w];
    // D3 / D4 (bits 4, 8, 16) and D7 (128) send a loss-only call to the log-domain kernel; with a gradient the mass check D6 decides.
    // (D9, the exact "a nonzero value left its lane's range" of checked renormalisations, is recorded in CTC_F6_D9 diagnostic builds
    // only: benign utterances flush irrelevant values all the time -- the thin front ahead of the bulk, the tail behind it -- so it
    // flags 60 % of the N(0,1) utterances at T = 1000 and still missed one harmful case in 30 000; tests/tools/flag_stats.py)
    // r04: ... in the FIRST HALF OF A FORWARD / BACKWARD PAIR (ctc_amd_loss_forward: the resume call will check every utterance's
    // posterior mass and redo what fails) only where the sound detector, the mass check D6 of calls with a gradient, finds
    // something to redo -- measured over ~100 000 utterances (tests/tools/flag_stats.py, flag_stats_short_labels.py; U <= 128,
    // V = 3 .. 256, N(0, 1 .. 5^2) logits; profiles/r04_flag_stats_*.log).  The signs D3 / D4 / D7 are heuristics (no local test
    // separates a harmful flush from the thin fronts and tails every utterance sheds all the time -- see D9), and unrestricted they
    // sent EVERY utterance with logits as sharp as a trained model's to the log domain: the public forward + backward path ran at
    // half speed there.  What makes the sweeps lose mass is (a) a BINDING alignment -- N(0, 3^2): 68-100 % of the utterances redone
    // at 0 .. 2 spare frames, 3-20 % at 8, 0-0.4 % at 16, none at 32 .. 512; (b) LONG DWELL -- few labels in many frames force long
    // runs of blanks whatever their probability: at 16 frames per label 0.4-0.8 % redone, at 32 6-45 %, at 170 100 % (also at
    // N(0, 2^2): 40 %); none at <= 12; (c) SHARPNESS itself -- P decaying by more than ~10 (classic) / ~12 (simplified) bits per
    // frame (N(0, 4^2): half of the north-star utterances redone; N(0, 3^2) decays by 9.0 / 10.7, N(0, 3.5^2) by 10.5 / 12.4).  The forward half trusts the linear sweeps
    // only inside all three bounds, with one or two label positions per lane; everything else keeps every sign, like a
    // stand-alone loss-only call (and like r03).
    const int slack = len - ll - (KIND == 0 ? S.repeats(ll, lane) : 0);
    const int decay = -(EX + frexp_e(s));                      // bits by which the unnormalised P has decayed over the utterance
    const bool trusted = p.resume == 2 && NL <= 2 && slack >= BIND_SLACK && len <= DWELL_MAX * (ll + 1) &&
                         4 * decay <= (KIND == 0 ? RATE_MAX_X4_CLASSIC : RATE_MAX_X4_SIMPLIFIED) * len;
    const int soft = trusted ? 0 : (28 | 128);
    // D10 (every loss-only call, two and more label positions per lane): more than DWELL_HARD frames per label position.  The soft
    // signs miss what long dwell does to MILD logits -- N(0, 2^2), 2 labels in 512 frames under a label bound of 128: 17-40 % of
    // the utterances fail the mass check of a call with a gradient, 22 % show no soft sign, and 2 of 256 stand-alone loss-only
    // calls returned a loss more than 1e-4 off (r04, tests/tools/flag_stats_short_labels.py; r03's rules had the same hole).
    // Nothing is redone at 30 frames per label position, 0-1.6 % at 57.
    const int hard = (!want_grad && NL >= 2 && okP && len > DWELL_HARD * (ll + 1)) ? D10_DWELL : 0;
    const int fl = (lds.flag & (want_grad ? (3 | D8_SYNC) : (3 | soft | D8_SYNC))) | (okP ? 0 : 1) | hard;
    if (lane == 0) {
      const double dlogp = (double)flog2(s) + (double)EX - sl2;
      logp_ws[b] = okP ? dlogp : -INFINITY;
      loss[b] = okP ? (float)(-dlogp * LN2_D) : INFINITY;
      lds.lossval = okP ? (float)(-dlogp * LN2_D) : INFINITY;
      // sum(loss) for the training loop (ctc_amd_loss_grad_sum): added HERE, mid-kernel, fire and forget -- at the end of the
      // kernel the two atomics of 256 workgroups finishing together cost 3.6 us
      lds.added = (p.sum_out != nullptr && fl == 0);
      if (lds.added) add_loss_fixed(p.sum_out, lds.lossval);
      const int fe = frexp_e(s);
      lds.lp_int = EX + fe;
      lds.cf = __builtin_amdgcn_rcpf(ldexp_f(s, -fe));   // 1 / mantissa, in (1, 2]
      lds.feasible = (fl == 0);
      lds.flag = fl;
      meet_ws[b] = make_int2(EX + fe, __float_as_int(lds.cf));
    }
  }
  __syncthreads();
  }  // !resume
  const bool go = lds.feasible != 0;
  if (!want_grad || !go) {  // loss only, or flagged (the log-domain kernel redoes this utterance): every role leaves here
    if (DIR == 0 && lane == 0) flag_ws[b] = lds.flag;
    return;
  }
  const int lp_int = lds.lp_int;
  const float cf30 = ldexp_f(lds.cf, 30);
  const bool carrier = lane == (DIR == 0 ? 0 : 63);
  F6_STAMP_PHASE2
  if (CTC_F6_PRIO1 != 3) __builtin_amdgcn_s_setprio(3);

  // ================= phase 2: everything from LDS =================
  {
    const int nb = geo.nblocks(2, DIR);
    int kflag = 0;
    for (int it = 0; it <= geo.NB + 2; ++it) {
      const int j = it - 2;
      if ((CTC_F6_ONLY & 1) && j >= 0 && j < nb) {
        const int g = geo.absblock(2, DIR, j);
        const int nv = geo.nvof(g);
        const float(*E)[LD::ES] = lds.E[DIR][j % 3];
        float(*RR)[LD::RS] = lds.R[DIR][j % 3];
        const int(*KG)[64] = lds.kg[DIR][j % 3];
        // exponent group of the R row at position d: rows are written BEFORE the recompute chain renormalises, s steps
        // after its checkpoint -> group max(s-1, 0) / RN.  s = nv-1-d (A, simplified B) / nv-d (classic B).
        auto grp = [&](int d) -> int {
          const int s = (KIND == 0 && DIR == 1) ? nv - d : nv - 1 - d;
          return (s > 0 ? s - 1 : 0) / RN;
        };
        // The R row holds the other direction's state in ITS slot order and lane exponents (group q); this chain needs it
        // one label position over, so one value per lane comes from the neighbour lane and carries that lane's exponent:
        // products with it ("shifted" parts) are scaled here, with their own scale KS; the others ("aligned") stay raw and
        // the helper multiplies them by the lane's scale KL (lds.kl), off the sequential chain.
        // S row entry per lane (in place of the R row, a region of 2 NL floats):
        //   NL > 2 : [aligned blank part (raw), token parts[NL] (raw; simplified: the shifted slot scaled), shifted blank part (scaled)]
        //   NL = 2 : [token parts[2], aligned blank part, shifted blank part]
        //   NL = 1 : [token part, shifted blank part]
        // the carrier lane's shifted blank part also holds the posterior of the boundary state (scaled with its own K0).
        float(*KLr)[64] = lds.kl[DIR][j % 3];
        int q = -1, kR = DEAD, ks = DEAD, seg = -1;
        float KL = 0.f, KS = 0.f, K0 = 0.f;
        float PL = 1.f, PS = 1.f, P0 = 1.f;  // pre-scale of the chain's operand for the aligned / shifted / boundary products
        bool wide = false;                   // wave-uniform: some lane's scale exceeds 2^KK_MAX in this exponent group
        bool dirty = false;                  // the scales have to be rebuilt before the next products
        auto setK = [&]() __attribute__((always_inline)) {
          // (the boundary state's posterior rides in the shifted part of the CARRIER lane -- 0 for A, 63 for B, the lane that holds
          // the R row's state next to the boundary -- so it needs no cross-lane read; K0 is zero on every other lane)
          const int ka = S.k + kR - lp_int, kb = S.k + ks - lp_int, kc = carrier ? S.kx + kR - lp_int : DEAD;
          const int kmx = imax(ka, imax(kb, kc));
          // D5 proper: only beyond 2^KK_MAX2 (see KK_MAX)
          kflag |= (kmx > KK_MAX2);
          KL = ldexp_f(cf30, imin(ka, KK_MAX));
          KS = ldexp_f(cf30, imin(kb, KK_MAX));
          K0 = carrier ? ldexp_f(cf30, imin(kc, KK_MAX)) : 0.f;
          KLr[++seg][lane] = KL;
          wide = __builtin_amdgcn_ballot_w64(kmx > KK_MAX) != 0;
          if (wide) {
            PL = ldexp_f(1.f, imin(imax(ka - KK_MAX, 0), KK_MAX2 - KK_MAX));
            PS = ldexp_f(1.f, imin(imax(kb - KK_MAX, 0), KK_MAX2 - KK_MAX));
            P0 = ldexp_f(1.f, imin(imax(kc - KK_MAX, 0), KK_MAX2 - KK_MAX));
          }
        };
        auto one = [&](auto FASTt, int d, int qd, bool ren, const Emis<NL> &e, const RRow<KIND, NL> &r, int kRq) __attribute__((always_inline)) {
          constexpr bool FASTF = decltype(FASTt)::value;  // (fast_from below: the scales are opened there)
          if constexpr (!FASTF) {
            if (qd != q) {  // new exponent group of the rows (the boundary exponent r.kx is constant inside a group as well)
              q = qd; kR = kRq;
              ks = (DIR == 0) ? from_next_lane_i(kR, r.kx) : from_prev_lane_i(kR, r.kx);
              dirty = true;
            }
            if (dirty) { setK(); dirty = false; }  // (also: this chain renormalised after the previous frame)
          }
          // the value one label position over: from the next lane for A (needs l = i+1 of a row that holds l = i), from the
          // previous lane for B; and the row's state at this chain's boundary position (l = 0 for A, l = UP for B)
          const float rs = (DIR == 0) ? from_next_lane(r.c[0], r.cx) : from_prev_lane(r.c[NL - 1], r.cx);
          const float r0 = (DIR == 0) ? r.c[0] : r.c[NL - 1];  // (meaningful on the carrier lane only)
          float qal = 0.f, tok[NL], qsh, p0;
          if constexpr (FASTF) {
            // full block of a packed chain, scales within 2^KK_MAX (fast_from below checks that where the scales change): the products
            // of the generic branch below without the two-factor form, the token parts as one packed multiply, no branch
            if constexpr (DIR == 0) S.step(e);
            const f2v SO = {S.o[0], S.o[1]}, RO = {r.o[0], r.o[1]};
            const f2v TK = SO * RO;
            if constexpr (DIR == 0) { qal = S.c[0] * r.c[1]; qsh = (S.c[1] * rs) * KS; }
            else { qal = S.c[1] * r.c[0]; qsh = (S.c[0] * rs) * KS; }
            p0 = S.cx * r0;
            tok[0] = TK.x; tok[1] = TK.y;
          } else if constexpr (KIND == 0) {
            if constexpr (DIR == 0) S.step(e);  // A: posterior of frame t from alpha[t+1], beta[t+1]
            // (A renormalises after the products below; its exponent is still the one the scales were built from.  A mantissa
            // product may underflow -- by then it is below 2^-16 units after scaling -- but never overflows)
            if (__builtin_expect(!wide, 1)) {
#pragma unroll
              for (int jj = 0; jj < NL; ++jj) tok[jj] = S.o[jj] * r.o[jj];
              if constexpr (DIR == 0) {
#pragma unroll
                for (int jj = 0; jj < NL - 1; ++jj) qal += S.c[jj] * r.c[jj + 1];
                qsh = (S.c[NL - 1] * rs) * KS;
              } else {
#pragma unroll
                for (int jj = 1; jj < NL; ++jj) qal += S.c[jj] * r.c[jj - 1];
                qsh = (S.c[0] * rs) * KS;
              }
              p0 = S.cx * r0;
            } else {  // the same products with the excess scale on this chain's operand first (KK_MAX)
#pragma unroll
              for (int jj = 0; jj < NL; ++jj) tok[jj] = (S.o[jj] * PL) * r.o[jj];
              if constexpr (DIR == 0) {
#pragma unroll
                for (int jj = 0; jj < NL - 1; ++jj) qal += (S.c[jj] * PL) * r.c[jj + 1];
                qsh = ((S.c[NL - 1] * PS) * rs) * KS;
              } else {
#pragma unroll
                for (int jj = 1; jj < NL; ++jj) qal += (S.c[jj] * PL) * r.c[jj - 1];
                qsh = ((S.c[0] * PS) * rs) * KS;
              }
              p0 = (S.cx * P0) * r0;
            }
          } else if constexpr (DIR == 0) {
            const float pin0 = ldexp_f(from_prev_lane(S.c[NL - 1], S.cx), S.dk);
            auto parts = [&](auto WIDEt) __attribute__((always_inline)) {
              constexpr bool W = decltype(WIDEt)::value;  // W: excess scale on this chain's operand first (KK_MAX)
#pragma unroll
              for (int jj = 0; jj < NL; ++jj) {
                const float pin = (jj == 0) ? pin0 : S.c[jj - 1];
                const float rn = (jj < NL - 1) ? r.c[(jj + 1) % NL] : rs;  // b(l = i+1)
                const float pf = W ? pin * (jj < NL - 1 ? PL : PS) : pin;
                tok[jj] = (pf * e.y[jj]) * rn;
                if (jj < NL - 1) qal += (W ? S.c[jj] * PL : S.c[jj]) * rn;
              }
              qal *= e.bl;
              qsh = (((W ? S.c[NL - 1] * PS : S.c[NL - 1]) * rs) * e.bl) * KS;
              tok[NL - 1] *= KS;
              p0 = (W ? S.cx * P0 : S.cx) * e.bl * r0;
            };
            if (__builtin_expect(!wide, 1)) parts(std::false_type{}); else parts(std::true_type{});
          } else {
            const float nin = ldexp_f(from_next_lane(S.c[0], S.cx), S.dk);
            auto parts = [&](auto WIDEt) __attribute__((always_inline)) {
              constexpr bool W = decltype(WIDEt)::value;
#pragma unroll
              for (int jj = 0; jj < NL; ++jj) {
                const float nx = (jj == NL - 1) ? nin : S.c[(jj + 1) % NL];
                const float rp = (jj > 0) ? r.c[(jj + NL - 1) % NL] : rs;  // a(l = i)
                if constexpr (W) tok[jj] = ((nx * (jj > 0 ? PL : PS)) * e.y[jj]) * rp;  // (the pre-scaled operand first)
                else tok[jj] = (rp * e.y[jj]) * nx;
                if (jj > 0) qal += (W ? S.c[jj] * PL : S.c[jj]) * rp;
              }
              qal *= e.bl;
              qsh = (((W ? S.c[0] * PS : S.c[0]) * rs) * e.bl) * KS;
              tok[0] *= KS;
              p0 = (W ? S.cx * P0 : S.cx) * e.bl * r0;
            };
            if (__builtin_expect(!wide, 1)) parts(std::false_type{}); else parts(std::true_type{});
          }
          qsh = __builtin_fmaf(p0, K0, qsh);  // the boundary state rides in the carrier lane's scaled part
          float *srow = RR[d] + 2 * lane * NL;
          if constexpr (NL == 1) *reinterpret_cast<float2 *>(srow) = make_float2(tok[0], qsh);
          else if constexpr (NL == 2) *reinterpret_cast<float4 *>(srow) = make_float4(tok[0], tok[1], qal, qsh);  // (token pair first: an aligned register pair)
          else {  // NL + 2 values: [qal, tok[NL], qsh] as 16-byte pieces and one 8-byte tail
            float sv[NL + 2];
            sv[0] = qal; sv[NL + 1] = qsh;
#pragma unroll
            for (int jj = 0; jj < NL; ++jj) sv[1 + jj] = tok[jj];
#pragma unroll
            for (int q4 = 0; q4 < (NL + 2) / 4; ++q4)
              *reinterpret_cast<float4 *>(srow + 4 * q4) = make_float4(sv[4 * q4], sv[4 * q4 + 1], sv[4 * q4 + 2], sv[4 * q4 + 3]);
            *reinterpret_cast<float2 *>(srow + NL) = make_float2(sv[NL], sv[NL + 1]);
          }
          if constexpr (!(KIND == 0 && DIR == 0)) S.step(e);
          if (ren) { S.template renorm<LV>(); dirty = true; }
        };
        // the frames d0 .. nv-1 one by one, everything read where it is used (partial blocks; the tail of a full block of a packed
        // chain from the segment on in which some lane's scale exceeds 2^KK_MAX)
        auto generic_from = [&](int d0) __attribute__((always_inline)) {
          for (int d = d0; d < nv; ++d) {
            Emis<NL> e;
            read_E<NL, LD>(E[d], lane, e);
            RRow<KIND, NL> r;
            read_R<KIND, NL, LD>(RR[d], lane, r);
            const int qd = grp(d);
            one(std::false_type{}, d, qd, (d + 1 + ren_shift<KIND, DIR>()) % RN == 0 || d == nv - 1, e, r, KG[qd][lane]);
          }
        };
        if constexpr (Chain<KIND, NL, DIR>::PACKED) {
          // Full blocks of the packed chains: straight-line code, NO branch per frame (the per-frame test of `wide` merged two
          // versions of every frame: five register copies and five scalar instructions per frame on the wavefront whose
          // instruction count bounds phase 2).  The scales change at positions known at compile time (segment starts); only there
          // is `wide` tested, and if it is set the rest of the block goes through generic_from, which knows the two-factor form.
          int dstart = 0;
          if (nv == BLK) {
            constexpr int PR = 3;
            Emis<NL> eb[BLK];
            RRow<KIND, NL> rb[BLK];
            int kq[CD::NG];
            static_for<0, CD::NG>([&](auto Q) { kq[decltype(Q)::value] = KG[decltype(Q)::value][lane]; });
            static_for<0, PR>([&](auto D) { read_R<KIND, NL, LD>(RR[decltype(D)::value], lane, rb[decltype(D)::value]); });
            static_for<0, BLK>([&](auto D) { read_E<NL, LD>(E[decltype(D)::value], lane, eb[decltype(D)::value]); });
            auto fast_from = [&](auto self, auto Dc) __attribute__((always_inline)) -> int {
              constexpr int d = decltype(Dc)::value;
              if constexpr (d >= BLK) return BLK;
              else {
                constexpr int sA = (KIND == 0 && DIR == 1) ? BLK - d : BLK - 1 - d;
                constexpr int qd = (sA > 0 ? sA - 1 : 0) / RN;
                constexpr int sP = (KIND == 0 && DIR == 1) ? BLK - (d - 1) : BLK - 1 - (d - 1);
                constexpr int qp = (d == 0) ? -1 : (sP > 0 ? sP - 1 : 0) / RN;
                constexpr bool renp = d > 0 && (d + ren_shift<KIND, DIR>()) % RN == 0;  // this chain renormalised after frame d - 1
                if constexpr (d + PR < BLK) read_R<KIND, NL, LD>(RR[d + PR], lane, rb[d + PR]);
                if constexpr (qd != qp || renp) {  // a segment opens here
                  if constexpr (qd != qp) {
                    q = qd; kR = kq[qd];
                    ks = (DIR == 0) ? from_next_lane_i(kR, rb[d].kx) : from_prev_lane_i(kR, rb[d].kx);
                  }
                  setK();
                  if (__builtin_expect(wide, 0)) { --seg; q = -1; return d; }  // (generic_from opens this segment again)
                }
                one(std::true_type{}, d, qd, (d + 1 + ren_shift<KIND, DIR>()) % RN == 0, eb[d], rb[d], kq[qd]);
                return self(self, std::integral_constant<int, d + 1>{});
              }
            };
            dstart = fast_from(fast_from, std::integral_constant<int, 0>{});
          }
          if (__builtin_expect(dstart < nv, 0)) generic_from(dstart);
        } else if (nv == BLK) {
          // emission rows of the whole block and the exponent groups up front, R rows PR frames ahead of their use
          constexpr int PR = 3;
          Emis<NL> eb[BLK];
          RRow<KIND, NL> rb[BLK];
          int kq[CD::NG];
          static_for<0, CD::NG>([&](auto Q) { kq[decltype(Q)::value] = KG[decltype(Q)::value][lane]; });
          static_for<0, PR>([&](auto D) { read_R<KIND, NL, LD>(RR[decltype(D)::value], lane, rb[decltype(D)::value]); });
          static_for<0, BLK>([&](auto D) { read_E<NL, LD>(E[decltype(D)::value], lane, eb[decltype(D)::value]); });
          static_for<0, BLK>([&](auto D) {
            constexpr int d = decltype(D)::value;
            if constexpr (d + PR < BLK) read_R<KIND, NL, LD>(RR[d + PR], lane, rb[d + PR]);
            constexpr int s = (KIND == 0 && DIR == 1) ? BLK - d : BLK - 1 - d;
            constexpr int qd = (s > 0 ? s - 1 : 0) / RN;
            one(std::false_type{}, d, qd, (d + 1 + ren_shift<KIND, DIR>()) % RN == 0, eb[d], rb[d], kq[qd]);
          });
        } else {
          generic_from(0);
        }
      }
      F6_BARRIER();
    }
    if (__builtin_amdgcn_ballot_w64(kflag != 0) != 0 && lane == 0) atomicOr(&lds.flag, 32);  // D5
  }
  __syncthreads();  // every role's phase-2 flags are in
  if (DIR == 0 && lane == 0) flag_ws[b] = lds.flag;
  F6_STAMP_DUMP(DIR);
}

// ------------------------------------------------------------------------------------------------
// recompute chain of side SIDE (phase 2): runs the OTHER direction's recursion inside one block, from that direction's
// checkpoint, and leaves the rows its main chain needs in LDS, in the main chain's slot order, with their exponents.
//   SIDE A (needs beta[t+1] at frame t = BLK g + d)      : R[nv-1] = checkpoint beta[BLK g + nv]; step frames downward
//   SIDE B classic (needs alpha[t+1] at t = BLK g + nv-1-d): step frames upward from alpha[BLK g], row after each step
//   SIDE B simplified (needs a[t])                        : row before each step
// ------------------------------------------------------------------------------------------------
template <int KIND, int NL, int NH, int BLK, int VPL, int SIDE, int XT>
__device__ __forceinline__ void run_recompute(const Problem &p, const Layout &L, const float *__restrict__ alpha_ws,
                                              const float *__restrict__ beta_ws, const int *__restrict__ kexp_ws,
                                              float2 *__restrict__ stats_ws, float2 *__restrict__ sink_ws,
                                              Lds<KIND, NL, NH, BLK, VPL> &lds, const Geo<BLK> &geo, bool want_grad, int b,
                                              int *flag_ws_dbg) {
  constexpr int RDIR = 1 - SIDE;  // direction of the recursion this wave runs
  F6_STAMP_DECL
  using LD = Lds<KIND, NL, NH, BLK, VPL>;
  using CD = Cad<BLK, NL>;
  constexpr int RN = CD::RN, LV = CD::LV;
  const int lane = threadIdx.x & 63;
  const int T = p.T, UP = L.UP, SRS = L.SRS;
  int ll = p.label_length[b] < 0 ? 0 : p.label_length[b];
  if (ll > p.U) ll = 0;
  const int nslot = L.nslot;
  const float *ck_rows = (RDIR == 0 ? alpha_ws : beta_ws) + (long)b * L.rows_b * SRS;
  const int *ck_k = kexp_ws + ((long)b * 2 + RDIR) * nslot * 64;
  float *dump = lds.dump[2 + SIDE];

  if (p.resume != 1) {  // phase 1: nothing to recompute yet -- this wavefront works the E stage of its side
    using SP = P1Split<BLK, NH, NL>;
    Rows<KIND, NL, VPL, XT> W;
    W.init(p, b, lane, ll, nullptr, nullptr);
    W.set_lds(lds.xcopy_r[SIDE], nullptr);
    if (lane == 0) W.xs[256 * VPL] = 0.f;  // pad slot of the gather copy: emission 0 for label positions beyond label_length
    float2 *stats = stats_ws + (long)b * T;
    float2 *sinkp = sink_ws + (long)b * 256;
    estage1<KIND, NL, NH, BLK, VPL, XT, SIDE, SP::first(NH), SP::count(NH)>(W, lds, geo, stats, sinkp, dump, lane, 2 + SIDE F6_ST_ARG);
    __syncthreads();
    __syncthreads();
  }
  if (!want_grad || lds.feasible == 0) return;
  F6_STAMP_PHASE2

  Chain<KIND, NL, RDIR> S;
  S.init_labels(p, b, lane, ll);
  const int nb = geo.nblocks(2, SIDE);
  auto ck_slot = [&](int j) -> int {
    int jj = j < nb ? j : nb - 1;
    jj = jj < 0 ? 0 : jj;
    const int g = geo.absblock(2, SIDE, jj);
    const int t = (SIDE == 0) ? BLK * g + geo.nvof(g) : BLK * g;  // beta at the upper boundary / alpha at the lower one
    return geo.slot(t < 0 ? 0 : t);
  };
  CkRow<KIND, NL> ck_next;
  load_ck<KIND, NL>(ck_next, ck_rows, ck_k, ck_slot(0), SRS, UP, lane);
  for (int it = 0; it <= geo.NB + 2; ++it) {
    const int j = it - 1;
    if ((CTC_F6_ONLY & 2) && j >= 0 && j < nb) {
      const int g = geo.absblock(2, SIDE, j);
      const int nv = geo.nvof(g);
      const float(*E)[LD::ES] = lds.E[SIDE][j % 3];
      float(*RR)[LD::RS] = lds.R[SIDE][j % 3];
      int(*KG)[64] = lds.kg[SIDE][j % 3];
      const CkRow<KIND, NL> ck = ck_next;
      load_ck<KIND, NL>(ck_next, ck_rows, ck_k, ck_slot(j + 1), SRS, UP, lane);  // next block's checkpoint, a block ahead
      restore<KIND, NL, RDIR>(S, ck);
      KG[0][lane] = S.k;
      int s = 0;  // steps since the checkpoint
      auto put = [&](int d) __attribute__((always_inline)) { write_R<KIND, NL, LD>(RR[d], dump, lane, S.c, S.o, S.cx, S.kx); };
      // one step, its row, then (every RN steps, if more rows follow) a renormalisation that opens the next exponent group
      Emis<NL> eb[BLK];  // full blocks: the emission rows go to registers before the chain starts
      if (nv == BLK) static_for<0, BLK>([&](auto D) { read_E<NL, LD>(E[decltype(D)::value], lane, eb[decltype(D)::value]); });
      auto stp = [&](int d) __attribute__((always_inline)) {
        Emis<NL> e;
        read_E<NL, LD>(E[d], lane, e);
        S.step(e);
      };
      auto stpb = [&](auto D) __attribute__((always_inline)) { S.step(eb[decltype(D)::value]); };
      auto after = [&](bool more) __attribute__((always_inline)) {
        ++s;
        if (s % RN == 0 && more) {
          S.template renorm<LV>();
          KG[s / RN][lane] = S.k;
        }
      };
      if constexpr (SIDE == 0) {
        // beta recursion downward: R[nv-1] = beta[BLK g + nv] (the checkpoint), then R[d-1] = beta[BLK g + d] after frame d
        put(nv - 1);
        if (nv == BLK) {
          static_for<0, BLK - 1>([&](auto I) {
            constexpr int d = BLK - 1 - decltype(I)::value;
            stpb(std::integral_constant<int, d>{}); put(d - 1); after(d > 1);
          });
        } else {
          for (int d = nv - 1; d >= 1; --d) { stp(d); put(d - 1); after(d > 1); }
        }
      } else {
        if constexpr (KIND == 0) {
          // alpha recursion upward; B's position d holds frame BLK g + nv-1-d; row after each step
          if (nv == BLK) {
            static_for<0, BLK>([&](auto I) {
              constexpr int i = decltype(I)::value;
              stpb(std::integral_constant<int, BLK - 1 - i>{}); put(BLK - 1 - i); after(i < BLK - 1);
            });
          } else {
            for (int i = 0; i < nv; ++i) { stp(nv - 1 - i); put(nv - 1 - i); after(i < nv - 1); }
          }
        } else {
          put(nv - 1);  // a[BLK g]
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
    }
    F6_BARRIER();
  }
  __syncthreads();
  F6_STAMP_DUMP(2 + SIDE);
}

// ------------------------------------------------------------------------------------------------
// helper wavefront h of NH per side: positions d = h, h + NH, ... of every block (FPH = BLK / NH per block)
// ------------------------------------------------------------------------------------------------
template <int KIND, int NL, int NH, int BLK, int VPL, int DIR, int XT>
__device__ __forceinline__ void run_helper(const Problem &p, const Layout &L, float2 *__restrict__ stats_ws,
                                           float2 *__restrict__ sink_ws, const float *__restrict__ d_loss, float *__restrict__ grad,
                                           Lds<KIND, NL, NH, BLK, VPL> &lds, const Geo<BLK> &geo, int h, int b, int *flag_ws_dbg = nullptr) {
  constexpr int V = 256 * VPL;
  constexpr int FPH = BLK / NH;
  using LD = Lds<KIND, NL, NH, BLK, VPL>;
  Rows<KIND, NL, VPL, XT> S;
  const int lane = threadIdx.x & 63;
  const int T = p.T;
  const int len = geo.len;
  int ll = p.label_length[b] < 0 ? 0 : p.label_length[b];
  if (ll > p.U) ll = 0;
  S.init(p, b, lane, ll, d_loss, grad);
  S.set_lds(lds.xcopy[DIR * NH + h], lds.bins[DIR * NH + h]);
  if (lane == 0) S.xs[V] = 0.f;  // pad slot of the gather copy
  float2 *stats = stats_ws + (long)b * T;
  float2 *sinkp = sink_ws + (long)b * 256;
  float *dump = lds.dump[4 + DIR * NH + h];
  const int wave = 4 + DIR * NH + h;
  F6_STAMP_DECL

  auto write_E = [&](float *row, const Emis<NL> &e) __attribute__((always_inline)) {
    st_slots<NL>(row + lane * NL, e.y);
    float *tq = (lane == 0) ? row + LD::UP : dump + lane;
    *tq = e.bl;
  };
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

  // ================= phase 1: E stage with statistics =================
  if (p.resume != 1) {
    using SP = P1Split<BLK, NH, NL>;
    if constexpr (NH == 6) {
      switch (h) {
        case 0: estage1<KIND, NL, NH, BLK, VPL, XT, DIR, SP::first(0), SP::count(0)>(S, lds, geo, stats, sinkp, dump, lane, wave F6_ST_ARG); break;
        case 1: estage1<KIND, NL, NH, BLK, VPL, XT, DIR, SP::first(1), SP::count(1)>(S, lds, geo, stats, sinkp, dump, lane, wave F6_ST_ARG); break;
        case 2: estage1<KIND, NL, NH, BLK, VPL, XT, DIR, SP::first(2), SP::count(2)>(S, lds, geo, stats, sinkp, dump, lane, wave F6_ST_ARG); break;
        case 3: estage1<KIND, NL, NH, BLK, VPL, XT, DIR, SP::first(3), SP::count(3)>(S, lds, geo, stats, sinkp, dump, lane, wave F6_ST_ARG); break;
        case 4: estage1<KIND, NL, NH, BLK, VPL, XT, DIR, SP::first(4), SP::count(4)>(S, lds, geo, stats, sinkp, dump, lane, wave F6_ST_ARG); break;
        default: estage1<KIND, NL, NH, BLK, VPL, XT, DIR, SP::first(5), SP::count(5)>(S, lds, geo, stats, sinkp, dump, lane, wave F6_ST_ARG); break;
      }
    } else if constexpr (NH == 4) {
      switch (h) {
        case 0: estage1<KIND, NL, NH, BLK, VPL, XT, DIR, SP::first(0), SP::count(0)>(S, lds, geo, stats, sinkp, dump, lane, wave F6_ST_ARG); break;
        case 1: estage1<KIND, NL, NH, BLK, VPL, XT, DIR, SP::first(1), SP::count(1)>(S, lds, geo, stats, sinkp, dump, lane, wave F6_ST_ARG); break;
        case 2: estage1<KIND, NL, NH, BLK, VPL, XT, DIR, SP::first(2), SP::count(2)>(S, lds, geo, stats, sinkp, dump, lane, wave F6_ST_ARG); break;
        default: estage1<KIND, NL, NH, BLK, VPL, XT, DIR, SP::first(3), SP::count(3)>(S, lds, geo, stats, sinkp, dump, lane, wave F6_ST_ARG); break;
      }
    } else if constexpr (NH == 2) {
      if (h == 0) estage1<KIND, NL, NH, BLK, VPL, XT, DIR, SP::first(0), SP::count(0)>(S, lds, geo, stats, sinkp, dump, lane, wave F6_ST_ARG);
      else estage1<KIND, NL, NH, BLK, VPL, XT, DIR, SP::first(1), SP::count(1)>(S, lds, geo, stats, sinkp, dump, lane, wave F6_ST_ARG);
    } else {
      estage1<KIND, NL, NH, BLK, VPL, XT, DIR, SP::first(0), SP::count(0)>(S, lds, geo, stats, sinkp, dump, lane, wave F6_ST_ARG);
    }
    // ================= meeting point =================
    __syncthreads();
    __syncthreads();
  }
  if (grad == nullptr || lds.feasible == 0) return;  // loss only / flagged
  F6_STAMP_PHASE2

  // ================= phase 2: E stage (statistics from the record), G stage three blocks behind =================
  {
    const int nb = geo.nblocks(2, DIR);
    int segq[FPH];  // posterior-scale segment of this helper's positions in a full block
#pragma unroll
    for (int q = 0; q < FPH; ++q) segq[q] = kl_segment<KIND, DIR, Cad<BLK, NL>::RN>(h + NH * q, BLK);
    if (h == 0 && DIR == 0) S.io.zero_rows(len, T);  // padded frames (base_loss.py:291-296)
    // Rows of a block are loaded one block ahead of its E stage, exponentiated there (in place) and used again by its G
    // stage three blocks later: a ring of five register sets addressed by (block mod 5) at COMPILE time.  Wide vocabularies
    // (four row segments per lane) hold two sets; their G stage re-reads and re-exponentiates its rows.
    constexpr bool RELOAD = VPL >= 4;
    // LA = 1: rows of block it+1 requested at the top of iteration it.  LA = 2: the register set the G stage has just consumed
    // (block it-3) takes the rows of block it+2 at once, at the END of iteration it -- nearly two blocks of look-ahead with the same
    // five sets.  Measured in r03 (one process, same box): LA = 2 153 us against 139 at B = 256, 131 against 125 at B = 64; a six-set
    // ring spills (193 against 149).  More look-ahead does not help this kernel; LA = 2 stays a diagnostic switch.
#ifndef CTC_F6_LA
#define CTC_F6_LA 1
#endif
    constexpr int LA = RELOAD ? 1 : CTC_F6_LA;
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
    float2 st_cur = make_float2(0.f, 0.f), st_n1 = make_float2(0.f, 0.f), st_next = make_float2(0.f, 0.f);
    bool massbad = false;
    if (nb > 0) {
      static_for<0, LA>([&](auto A) {
        static_for<0, FPH>([&](auto Q) { S.io.load_x(X[decltype(A)::value][decltype(Q)::value], fr(2, decltype(A)::value, h + NH * decltype(Q)::value)); });
      });
      st_cur = stats[fr(2, 0, lane)];
      if constexpr (LA == 2) st_n1 = stats[fr(2, 1, lane)];
    }
    // FAST: steady state (E stage on a full block, G stage on a full block): no branch around a memory operation, so the
    // `s_waitcnt vmcnt(N)` hipcc derives for the ring leave the look-ahead loads AND the gradient stores of the last
    // blocks in flight (see estage1; with the general body every iteration waited for its own stores to reach memory)
    auto body = [&](auto R, auto FASTt, int it) __attribute__((always_inline)) {
      constexpr bool FAST = decltype(FASTt)::value;
      constexpr int r = decltype(R)::value;         // = it mod RING
      constexpr int rn = (r + 1) % RING;            // block it+1 (LA = 1: being loaded)
      constexpr int rg = (r + 2) % RING;            // block it-3 (G stage; not with RELOAD), then block it+2 (LA = 2)
      if constexpr (RELOAD) {
        static_for<0, FPH>([&](auto Q) { S.io.load_x(XG[decltype(Q)::value], fr(2, it - 3, h + NH * decltype(Q)::value)); });
        sgl = stats[fr(2, it - 3, lane)];
      }
      // ---- E stage (block it) ----
      const int j = it;
      SG[r] = st_cur;
      if ((CTC_F6_ONLY & 4) && (FAST || j < nb)) {
        const int g = geo.absblock(2, DIR, j);
        const int nv = FAST ? BLK : geo.nvof(g);
        float(*E)[LD::ES] = lds.E[DIR][j % 3];
        if constexpr (LA == 1) {
          st_next = stats[fr(2, j + 1, lane)];
          // rows of the next block: their register set was freed by the G stage of the previous iteration, so the loads go out
          // first and have the whole iteration (not the part after this block's E stage) to arrive
          static_for<0, FPH>([&](auto Q) { S.io.load_x(X[rn][decltype(Q)::value], fr(2, j + 1, h + NH * decltype(Q)::value)); });
        }
        if (FAST || __builtin_expect(nv == BLK, 1)) {
          // (three passes over the frames: the gathers of all of them go through the one LDS copy back to back -- in order --
          // and their round trips overlap instead of adding up)
          Emis<NL> e[FPH];
          static_for<0, FPH>([&](auto Q) {
            constexpr int q = decltype(Q)::value;
            float4 ev[VPL];
            S.expo(X[r][q], readlane_f(st_cur.x, h + NH * q), ev);
            static_for<0, VPL>([&](auto W) {
              constexpr int w = decltype(W)::value;
              X[r][q][w] = make_float4(ev[w].x, ev[w].y, ev[w].z, ev[w].w);
            });
          });
          static_for<0, FPH>([&](auto Q) { S.gather(X[r][decltype(Q)::value], e[decltype(Q)::value]); });
          static_for<0, FPH>([&](auto Q) { write_E(E[h + NH * decltype(Q)::value], e[decltype(Q)::value]); });
        } else {
          for (int d = h; d < nv; d += NH) {
            float4 xr[VPL], ev[VPL];
            S.io.load_x(xr, geo.frame(DIR, g, d));
            const float2 sd = stats[geo.frame(DIR, g, d)];
            S.expo(xr, sd.x, ev);
            Emis<NL> e;
            S.gather(ev, e);
            write_E(E[d], e);
          }
        }
        if constexpr (LA == 1) st_cur = st_next;
      }
      // ---- G stage (block it-3): posterior scatter + gradient rows ----
      const int gj = it - 3;
      if ((CTC_F6_ONLY & 8) && (FAST || (gj >= 0 && gj < nb))) {
        const int g = geo.absblock(2, DIR, gj);
        const int nv = FAST ? BLK : geo.nvof(g);
        const float(*SR)[LD::RS] = lds.R[DIR][gj % 3];
        const float(*KLr)[64] = lds.kl[DIR][gj % 3];  // [segment][lane]
        // S row entry -> blank posterior of the lane and token posteriors of its slots, all in units of 2^-30 (see run_main)
        auto read_S = [&](int d, int sg, float &qb, float (&qt)[NL]) __attribute__((always_inline)) {
          const float *srow = SR[d] + 2 * lane * NL;
          const float kl = KLr[sg][lane];
          constexpr int JS = (KIND == 1) ? (DIR == 0 ? NL - 1 : 0) : -1;  // simplified: the slot whose token part is already scaled
          float qal = 0.f, qsh;
          if constexpr (NL == 1) { const float2 t = *reinterpret_cast<const float2 *>(srow); qt[0] = t.x; qsh = t.y; }
          else if constexpr (NL == 2) { const float4 t = *reinterpret_cast<const float4 *>(srow); qt[0] = t.x; qt[1] = t.y; qal = t.z; qsh = t.w; }
          else {
            float sv[NL + 2];
#pragma unroll
            for (int q4 = 0; q4 < (NL + 2) / 4; ++q4) {
              const float4 t = *reinterpret_cast<const float4 *>(srow + 4 * q4);
              sv[4 * q4] = t.x; sv[4 * q4 + 1] = t.y; sv[4 * q4 + 2] = t.z; sv[4 * q4 + 3] = t.w;
            }
            const float2 u = *reinterpret_cast<const float2 *>(srow + NL);
            sv[NL] = u.x; sv[NL + 1] = u.y;
            qal = sv[0]; qsh = sv[NL + 1];
#pragma unroll
            for (int jj = 0; jj < NL; ++jj) qt[jj] = sv[1 + jj];
          }
#pragma unroll
          for (int jj = 0; jj < NL; ++jj) if (jj != JS) qt[jj] *= kl;
          qb = qal * kl + qsh;
        };
        if (FAST || __builtin_expect(nv == BLK, 1)) {
          // qb[FPH]: total posterior mass of this helper's LAST frame of the block (D6).  Mass lost by a chain is missing
          // from every frame between the place of the loss and the end of that chain's range, so one frame per helper
          // and block sees it -- provided it is the helper's frame FARTHEST along the chain: the NH helpers then cover the last
          // NH frames of every block, and a loss anywhere in the block shows in at least the very last one.  (Until r03 the FIRST
          // frame was taken: mass lost in the last frames of a chain's range -- the beta chain reaching frames 0..3 of a nearly forced
          // alignment with sharp logits -- fell between the samples: a gradient 4e-3 off, unflagged; tests/tools/soak_formats.py.)
          float qb[4], qt[FPH][NL];
          static_for<0, FPH>([&](auto Q) {
            constexpr int q = decltype(Q)::value;
            read_S(h + NH * q, segq[q], qb[q], qt[q]);
          });
          qb[FPH] = qb[FPH - 1];
#pragma unroll
          for (int jj = 0; jj < NL; ++jj) qb[FPH] += qt[FPH - 1][jj];
#pragma unroll
          for (int f = FPH + 1; f < 4; ++f) qb[f] = 0.f;
#ifdef CTC_F6_DEBUG2
          if (DIR == 0 && gj == 0 && h == 1 && NL == 2) {  // what helper 1 of side A reads for position d = 1 of the first block
            int *dbg = flag_ws_dbg + p.B + (long)b * 2048 + 1024;
            dbg[lane] = __float_as_int(qb[0]); dbg[64 + lane] = __float_as_int(qt[0][0]); dbg[128 + lane] = __float_as_int(qt[0][NL - 1]);
          }
#endif
          static_assert(FPH == 3 || FPH == 2, "one four-value reduction");
          const float qall = swap_reduce<4, false>(qb);  // blank posteriors of the FPH frames and one total mass
          massbad |= !(fabsf(readlane_f(qall, SwapLanes<4>::lane(FPH)) - 1073741824.0f) < 1073741824.0f * MASS_TOL);
          constexpr bool BATCH = !RELOAD && VPL == 1;  // (register budget: FPH more row sets)
          uint4 PU[BATCH ? FPH : 1][VPL];
          if constexpr (BATCH) static_for<0, FPH>([&](auto Q) { S.scatter(qt[decltype(Q)::value], PU[decltype(Q)::value]); });
          static_for<0, FPH>([&](auto Q) {
            constexpr int q = decltype(Q)::value;
            const int d = h + NH * q;
            const float qbs = readlane_f(qall, SwapLanes<4>::lane(q));
            if constexpr (BATCH) {
              S.grad_out(geo.frame(DIR, g, d), qbs, PU[q], X[rg][q], readlane_f(SG[rg].y, d));
            } else if constexpr (RELOAD) {
              float4 ev[VPL];
              S.expo(XG[q], readlane_f(sgl.x, d), ev);
              S.grad_row(geo.frame(DIR, g, d), qbs, qt[q], ev, readlane_f(sgl.y, d));
            } else {
              S.grad_row(geo.frame(DIR, g, d), qbs, qt[q], X[rg][q], readlane_f(SG[rg].y, d));
            }
          });
        } else {
          for (int d = h; d < nv; d += NH) {
            float4 xr[VPL], ev[VPL];
            S.io.load_x(xr, geo.frame(DIR, g, d));
            const float2 sd = stats[geo.frame(DIR, g, d)];
            S.expo(xr, sd.x, ev);
            float qb, qt[NL];
            read_S(d, kl_segment<KIND, DIR, Cad<BLK, NL>::RN>(d, nv), qb, qt);
            float tot = qb;
#pragma unroll
            for (int jj = 0; jj < NL; ++jj) tot += qt[jj];
            qb = wave_sum_dpp(qb);
            massbad |= !(fabsf(wave_sum_dpp(tot) - 1073741824.0f) < 1073741824.0f * MASS_TOL);
            S.grad_row(geo.frame(DIR, g, d), qb, qt, ev, sd.y);
          }
        }
      }
      if constexpr (LA == 2) {  // (unconditional, clamped frame indices: no branch around a memory operation)
        st_next = stats[fr(2, it + 2, lane)];
        static_for<0, FPH>([&](auto Q) { S.io.load_x(X[rg][decltype(Q)::value], fr(2, it + 2, h + NH * decltype(Q)::value)); });
        st_cur = st_n1; st_n1 = st_next;
      }
      F6_BARRIER();
    };
    // full blocks of this side: all of them, except that side A ends with the utterance's last block, which may be partial
    const int nbf = nb - ((DIR == 0 && nb > 0 && geo.nvof(geo.absblock(2, DIR, nb - 1)) != BLK) ? 1 : 0);
    int it0 = 0;
    // the pipeline fills through the general body; the steady state needs a G stage on a real block (it >= 3) and starts on
    // a multiple of RING (the register sets are addressed by it mod RING at compile time)
    constexpr int START = ((3 + RING - 1) / RING) * RING;
    static_for<0, START>([&](auto I) {
      constexpr int i = decltype(I)::value;
      if (i <= geo.NB + 2) body(std::integral_constant<int, i % RING>{}, std::false_type{}, i);
    });
    it0 = START;
    if (it0 + RING <= nbf) {  // steady state: iterations 3 <= it < nbf (one trip peeled, see estage1)
      static_for<0, RING>([&](auto R) { body(R, std::true_type{}, it0 + decltype(R)::value); });
      it0 += RING;
      for (; it0 + RING <= nbf; it0 += RING) {
        static_for<0, RING>([&](auto R) { body(R, std::true_type{}, it0 + decltype(R)::value); });
      }
    }
    for (; it0 <= geo.NB + 2; it0 += RING) {
      static_for<0, RING>([&](auto R) {
        if (it0 + decltype(R)::value <= geo.NB + 2) body(R, std::false_type{}, it0 + decltype(R)::value);
      });
    }
    if (massbad && lane == 0) atomicOr(&lds.flag, 64);  // D6
  }
  __syncthreads();
  F6_STAMP_DUMP(wave);
}

// Wavefront roles: 0 main A, 1 main B, 2 recompute for A, 3 recompute for B, then NH helpers of A, NH helpers of B.
template <int KIND, int NL, int NH, int BLK, int VPL, int XT>
__global__ __launch_bounds__(64 * (4 + 2 * NH)) void fused6_kernel(Problem p, Layout L, float *__restrict__ alpha_ws,
                                                                    float *__restrict__ beta_ws, int *__restrict__ kexp_ws,
                                                                    double *__restrict__ logp_ws,
                                                                    float2 *__restrict__ stats_ws,
                                                                    float2 *__restrict__ sink_ws,
                                                                    float *__restrict__ loss,
                                                                    const float *__restrict__ d_loss,
                                                                    float *__restrict__ grad, int *__restrict__ flag_ws,
                                                                    int2 *__restrict__ meet_ws, const int *__restrict__ perm) {
  // (the log-domain roles reuse the LDS once the linear-domain ones are done with it)
  __shared__ __attribute__((aligned(16))) union LdsBoth {
    Lds<KIND, NL, NH, BLK, VPL> lin;
    fused5::Lds<KIND, NL, (NH == 6 ? 4 : NH), BLK, VPL> log;  // (the log-domain roles know up to four helpers per side)
  } both;
  constexpr int NH5 = NH == 6 ? 4 : NH;
  Lds<KIND, NL, NH, BLK, VPL> &lds = both.lin;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = perm ? perm[blockIdx.x] : (int)blockIdx.x;
  Geo<BLK> geo;
  geo.init(clampi(p.logit_length[b], 0, p.T));
  if (threadIdx.x == 0) {
    lds.flag = 0; lds.feasible = 0; lds.lossval = INFINITY; lds.added = 0;
    lds.p1_cons[0] = 0; lds.p1_cons[1] = 0;
#pragma unroll
    for (int w_ = 0; w_ < 8; ++w_) { lds.p1_prog[0][w_] = 0; lds.p1_prog[1][w_] = 0; }
    if (p.resume == 1) {  // the loss-only call left the outcome of the meeting point in the workspace
      const int f = flag_ws[b];
      const int2 m = meet_ws[b];
      lds.flag = f; lds.feasible = (f == 0); lds.lp_int = m.x; lds.cf = __int_as_float(m.y);
    }
  }
  if (threadIdx.x < Lds<KIND, NL, NH, BLK, VPL>::NW) lds.l2s[threadIdx.x] = 0.0;
  __syncthreads();
  if (w == 0) {
    __builtin_amdgcn_s_setprio(CTC_F6_PRIO1);
    run_main<KIND, NL, NH, BLK, VPL, 0>(p, L, alpha_ws, beta_ws, kexp_ws, logp_ws, loss, flag_ws, meet_ws, lds, geo, grad != nullptr, b);
  } else if (w == 1) {
    __builtin_amdgcn_s_setprio(CTC_F6_PRIO1);
    run_main<KIND, NL, NH, BLK, VPL, 1>(p, L, alpha_ws, beta_ws, kexp_ws, logp_ws, loss, flag_ws, meet_ws, lds, geo, grad != nullptr, b);
  } else if (w == 2) {
    __builtin_amdgcn_s_setprio(2);
    run_recompute<KIND, NL, NH, BLK, VPL, 0, XT>(p, L, alpha_ws, beta_ws, kexp_ws, stats_ws, sink_ws, lds, geo, grad != nullptr, b, flag_ws);
  } else if (w == 3) {
    __builtin_amdgcn_s_setprio(2);
    run_recompute<KIND, NL, NH, BLK, VPL, 1, XT>(p, L, alpha_ws, beta_ws, kexp_ws, stats_ws, sink_ws, lds, geo, grad != nullptr, b, flag_ws);
  } else {
    // which helper a wavefront is.  Wavefronts go to the SIMDs in the order 0, 2, 1, 3, 0, ... and a SIMD prefers its OLDER wavefronts:
    // with the plain order (side A = wavefronts 4 .. 3+NH, side B after them) the side-B helpers are the youngest wavefront of every
    // SIMD (r04 stamps: 82-85 us of work in phase 2 against 66-70 for side A).  CTC_F6_ROLEMAP (experiment builds): other orders.
#ifndef CTC_F6_ROLEMAP
#define CTC_F6_ROLEMAP 0
#endif
    int hs = (w - 4) / NH, hh = (w - 4) % NH;  // (side, index): the plain order
    if constexpr (CTC_F6_ROLEMAP == 1 && NH == 4) {        // A0 A1 B0 B1 A2 A3 B2 B3: side A beside the main chains, side B beside the recompute chains
      const int q = w - 4; hs = (q >> 1) & 1; hh = (q & 1) + 2 * (q >> 2);
    } else if constexpr (CTC_F6_ROLEMAP == 2) {             // side B first (older)
      hs = 1 - hs;
    } else if constexpr (CTC_F6_ROLEMAP == 3 && NH == 4) {  // A0 B0 A1 B1 A2 B2 A3 B3
      const int q = w - 4; hs = q & 1; hh = q >> 1;
    }
    hs = __builtin_amdgcn_readfirstlane(hs); hh = __builtin_amdgcn_readfirstlane(hh);
    if (hs == 0) {
      run_helper<KIND, NL, NH, BLK, VPL, 0, XT>(p, L, stats_ws, sink_ws, d_loss, grad, lds, geo, hh, b, flag_ws);
    } else {
      if (CTC_F6_HPRIO_B != 0) __builtin_amdgcn_s_setprio(CTC_F6_HPRIO_B);
      run_helper<KIND, NL, NH, BLK, VPL, 1, XT>(p, L, stats_ws, sink_ws, d_loss, grad, lds, geo, hh, b, flag_ws);
    }
  }
  // Utterances the linear domain cannot hold (flags D1..D6, normally none): the same wavefronts redo them in the log
  // domain right here -- same roles, the LDS reused, every output row rewritten -- instead of a second launch that
  // finds nothing to do (4.5 us per call).  In a gradient-resume call the flag is the one the loss-only call left.
  __syncthreads();
  const int fl = lds.flag;
  const float lossval = lds.lossval;
  const int added = lds.added;
  __syncthreads();
  if (fl != 0) {
    __builtin_amdgcn_s_setprio(0);
    if (w < 4 + 2 * NH5) fused5::run_roles<KIND, NL, NH5, BLK, VPL, XT>(p, L, alpha_ws, beta_ws, logp_ws, stats_ws, loss, d_loss, grad, sink_ws, both.log, w, b);
    else fused5::run_idle<BLK>(p, grad != nullptr, b);  // wavefronts the log-domain roles have no work for keep their barriers
  }
  // sum(loss) for the training loop, without a launch of its own (ctc_amd_loss_grad_sum): the thread that wrote loss[b] --
  // lane 0 of main chain A in either domain -- adds it in fixed point; the second call of a loss / gradient pair adds nothing
  if (p.sum_out != nullptr && p.resume != 1 && threadIdx.x == 0 && fl != 0) {  // redone in the log domain: its loss, not the first one
    if (added) add_loss_fixed(p.sum_out, lossval, -1);
    add_loss_fixed(p.sum_out, loss[b]);
  }
  if (p.sum_zero != nullptr && blockIdx.x == 0 && threadIdx.x == 0) { p.sum_zero[0] = 0; p.sum_zero[1] = 0; }
}

}  // namespace fused6

template <int NL, int NH, int BLK, int VPL>
static hipError_t launch6(const Problem &p, const Layout &L, float *a, float *b, int *kexp, double *lp, float2 *stats, float2 *sink, float *loss,
                          const float *d_loss, float *grad, int *flags, int2 *meet, const int *perm, hipStream_t st) {
  static_assert(sizeof(fused6::Lds<CTC_FUSED_KIND, NL, NH, BLK, VPL>) <= 160 * 1024, "LDS budget of one CU");
  const bool al16 = (p.align_bits & 15) == 0;  // 16-byte row accesses need aligned base pointers as well as strides
  const bool plain = al16 && p.xdtype == 0 && p.V == 256 * VPL && p.xst == p.V && p.gst == p.V;
  const dim3 grid(p.B), block(64 * (4 + 2 * NH));
#ifdef CTC_F6_NS_ONLY  // experiment builds (scripts/build_f6_variant.sh): the plain-format instantiation only -- a minute instead of seven
  if (!plain) return hipErrorInvalidValue;
  hipLaunchKernelGGL((fused6::fused6_kernel<CTC_FUSED_KIND, NL, NH, BLK, VPL, 0>), grid, block, 0, st, p, L, a, b, kexp, lp, stats, sink, loss,
                     d_loss, grad, flags, meet, perm);
  return hipGetLastError();
#else
  if (plain)
    hipLaunchKernelGGL((fused6::fused6_kernel<CTC_FUSED_KIND, NL, NH, BLK, VPL, 0>), grid, block, 0, st, p, L, a, b, kexp, lp, stats, sink, loss,
                       d_loss, grad, flags, meet, perm);
  else if (al16 && p.xdtype == 0 && ((p.V | p.xsb | p.xst | p.gsb | p.gst) & 3) == 0)
    hipLaunchKernelGGL((fused6::fused6_kernel<CTC_FUSED_KIND, NL, NH, BLK, VPL, 1>), grid, block, 0, st, p, L, a, b, kexp, lp, stats, sink, loss,
                       d_loss, grad, flags, meet, perm);
  else if (p.xdtype == 0)
    hipLaunchKernelGGL((fused6::fused6_kernel<CTC_FUSED_KIND, NL, NH, BLK, VPL, 3>), grid, block, 0, st, p, L, a, b, kexp, lp, stats, sink, loss,
                       d_loss, grad, flags, meet, perm);
  else
    hipLaunchKernelGGL((fused6::fused6_kernel<CTC_FUSED_KIND, NL, NH, BLK, VPL, 2>), grid, block, 0, st, p, L, a, b, kexp, lp, stats, sink, loss,
                       d_loss, grad, flags, meet, perm);
  return hipGetLastError();
#endif
}

// One translation unit per (lattice kind, label positions per lane): -DCTC_FUSED_KIND=0|1 -DCTC_FUSED6_NL=1|2|4.
// Exported: run_fused6_<kind>_nl<NL>.  Writes flags[b] != 0 for every utterance the caller has to redo in the log domain.
#ifndef CTC_FUSED6_NL
#error "compile with -DCTC_FUSED6_NL=1, 2 or 4"
#endif
#define CTC_F6_CAT2(a, b, c) a##b##c
#define CTC_F6_CAT(a, b, c) CTC_F6_CAT2(a, b, c)
#if CTC_FUSED_KIND == 0
#define CTC_F6_ENTRY CTC_F6_CAT(run_fused6_classic, _nl, CTC_FUSED6_NL)
#else
#define CTC_F6_ENTRY CTC_F6_CAT(run_fused6_simplified, _nl, CTC_FUSED6_NL)
#endif
hipError_t run_order(const Problem &p, const Layout &L, char *ws, hipStream_t st);  // ctc_kernels.hip

hipError_t CTC_F6_ENTRY(const Problem &p, const Layout &L, char *ws, float *loss, const float *d_loss, float *grad,
                        hipStream_t st) {
  float *alpha = reinterpret_cast<float *>(ws + L.off_alpha);
  float *beta = reinterpret_cast<float *>(ws + L.off_beta);
  double *logp = reinterpret_cast<double *>(ws + L.off_logp);
  float2 *stats = reinterpret_cast<float2 *>(ws + L.off_emis);  // the emission region of the v1 pipeline is free here
  float2 *sink = reinterpret_cast<float2 *>(ws + L.off_dummy);  // 2 KB per utterance: target of stores that carry nothing
  int *kexp = reinterpret_cast<int *>(ws + L.off_kexp);
  int *flags = reinterpret_cast<int *>(ws + L.off_flags);
  int2 *meet = reinterpret_cast<int2 *>(ws + L.off_meet);
  if (L.NL != CTC_FUSED6_NL) return hipErrorInvalidValue;
  // more utterances than CUs: longest first (one small kernel; skipped for batches that fit the chip in one go)
  const int *perm = nullptr;
  if (p.B > 256 && p.B <= 8192) {
    hipError_t e = run_order(p, L, ws, st);
    if (e != hipSuccess) return e;
    perm = reinterpret_cast<const int *>(ws + L.off_perm);
  }
#if defined(CTC_F6_NS_ONLY)
  if (p.V > 256) return hipErrorInvalidValue;
#ifndef CTC_F6_NS_NH
#define CTC_F6_NS_NH CTC_F6_NH12
#define CTC_F6_NS_BLK 12
#endif
  return launch6<CTC_FUSED6_NL, CTC_F6_NS_NH, CTC_F6_NS_BLK, 1>(p, L, alpha, beta, kexp, logp, stats, sink, loss, d_loss, grad, flags, meet, perm, st);
#elif CTC_FUSED6_NL == 8
  return p.V <= 256 ? launch6<8, 1, 3, 1>(p, L, alpha, beta, kexp, logp, stats, sink, loss, d_loss, grad, flags, meet, perm, st)
                    : launch6<8, 1, 3, 2>(p, L, alpha, beta, kexp, logp, stats, sink, loss, d_loss, grad, flags, meet, perm, st);
#elif CTC_FUSED6_NL == 4
  return p.V <= 256 ? launch6<4, 2, 6, 1>(p, L, alpha, beta, kexp, logp, stats, sink, loss, d_loss, grad, flags, meet, perm, st)
                    : launch6<4, 2, 6, 2>(p, L, alpha, beta, kexp, logp, stats, sink, loss, d_loss, grad, flags, meet, perm, st);
#else
  return p.V <= 256   ? launch6<CTC_FUSED6_NL, CTC_F6_NH12, 12, 1>(p, L, alpha, beta, kexp, logp, stats, sink, loss, d_loss, grad, flags, meet, perm, st)
         : p.V <= 512 ? launch6<CTC_FUSED6_NL, 2, 6, 2>(p, L, alpha, beta, kexp, logp, stats, sink, loss, d_loss, grad, flags, meet, perm, st)
                      : launch6<CTC_FUSED6_NL, 2, 6, 4>(p, L, alpha, beta, kexp, logp, stats, sink, loss, d_loss, grad, flags, meet, perm, st);
#endif
}

}  // namespace ctc

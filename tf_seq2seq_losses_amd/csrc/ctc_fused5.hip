// Fused loss + gradient kernel for gfx950, pipeline v4: NO lattice spill.  One workgroup of 4 + 2*NH wavefronts per
// utterance (12 at NH = 4, three per SIMD of the CU that owns the utterance):
//
//   wave 0 / 1 : main chain A / B -- the lattice recursion only (alpha forward from frame 0, beta backward from frame
//                len-1, meeting in the middle like ctc_fused.hip / ctc_fused4.hip).  Phase 1 leaves ONE checkpoint row per
//                block in HBM (1/BLK of the rows ctc_fused4.hip spills).  Phase 2 reads everything from LDS.
//   wave 2 / 3 : recompute chain for A / for B.  Phase 2: for the block its main chain will process next it restarts
//                the OTHER side's recursion from that side's checkpoint and regenerates the BLK lattice rows into LDS
//                (R rows): beta rows for A's blocks, alpha rows for B's.  It reuses the emissions the helpers already
//                staged for that block (emissions do not depend on the direction).  Phase 1 (nothing to recompute yet):
//                it works part of its side's E stage (estage1).
//   waves 4..  : NH helpers per side -- logits rows from HBM (prefetched a block ahead in registers), log-softmax
//                statistics by DPP reductions, emission gather through an LDS copy of the row (E stage); posterior scatter
//                with fixed-point ds_add_u32 into an LDS token row and the softmax - posterior store, non-temporal (G
//                stage).  The logits rows of a block stay in the helper's registers from its E stage to its G stage.
//
//   Pipeline per side, iteration `it`:  E(block it) -> recompute(block it-1) -> main(block it-2) -> G(block it-3);
//   E and R/S rows are triple-buffered in LDS; all wavefronts meet at ONE raw s_barrier per iteration and derive the same
//   iteration counts from logit_length, so barrier counts match by construction.
//   Blocks live on an absolute grid of BLK frames; only the block that contains frame len-1 may be partial.
//
//   HBM traffic per utterance: logits read twice (once per phase), gradient written once, T/BLK checkpoint rows and
//   8 bytes of softmax statistics per frame: ~1.55x the algorithmic 2*T*V*4 bytes (ctc_fused4.hip: 2.5x, v1: 4.4x).
//
// References: classic_ctc_loss.py:310-462,565-669, simplified_ctc_loss.py:291-438,456-534, base_loss.py:262-298,328-344,
// 420-468, tools.py:27-40.  Eligibility (fused5_eligible in ctc_capi.hip): logits input, V <= 512 with U <= 512 or V <= 1024 with
// U <= 128 (LDS budget); otherwise ctc_fused.hip / the v1 pipeline run.  The roles live in ctc_fused5_roles.h (shared with
// ctc_fused6.hip, which runs them inside its own launch for the utterances it flags).
// Instantiated per input/output format XT (Side in ctc_fused_common.h): contiguous float32, strided float32, bfloat16,
// and float32 rows that are not 16-byte aligned.  With grad == NULL every role returns at the meeting point (loss only).
#include "ctc_fused5_roles.h"

namespace ctc {
namespace fused5 {

// Wavefront roles: 0 main A, 1 main B, 2 recompute for A, 3 recompute for B, then NH helpers of A, NH helpers of B.
template <int KIND, int NL, int NH, int BLK, int VPL, int XT>
__global__ __launch_bounds__(64 * (4 + 2 * NH)) void fused5_kernel(Problem p, Layout L, float *__restrict__ alpha_ws,
                                                                    float *__restrict__ beta_ws,
                                                                    double *__restrict__ logp_ws,
                                                                    float2 *__restrict__ stats_ws,
                                                                    float *__restrict__ loss,
                                                                    const float *__restrict__ d_loss,
                                                                    float *__restrict__ grad, void *stamp_ws,
                                                                    const int *__restrict__ perm,
                                                                    const int *__restrict__ only_if) {
  __shared__ __attribute__((aligned(16))) Lds<KIND, NL, NH, BLK, VPL> lds;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: keeps everything derived from it in SGPRs
  // utterance of this workgroup: workgroups start in index order, and with more utterances than CUs the longest ones
  // go first (perm from order_kernel; ragged batch of 512: -22 %)
  const int b = perm ? perm[blockIdx.x] : (int)blockIdx.x;
  // fallback launch behind the linear-domain kernel (ctc_fused6.hip): only the utterances it flagged
  if (only_if && only_if[b] == 0) return;
  fused5::run_roles<KIND, NL, NH, BLK, VPL, XT>(p, L, alpha_ws, beta_ws, logp_ws, stats_ws, loss, d_loss, grad, stamp_ws, lds, w, b);
}

}  // namespace fused5

hipError_t run_order(const Problem &p, const Layout &L, char *ws, hipStream_t st);  // ctc_kernels.hip: longest utterances first

template <int NL, int NH, int BLK, int VPL>
static hipError_t launch5(const Problem &p, const Layout &L, float *a, float *b, double *lp, float2 *stats, float *loss,
                          const float *d_loss, float *grad, void *stamp, const int *perm, const int *only_if, hipStream_t st) {
  static_assert(sizeof(fused5::Lds<CTC_FUSED_KIND, NL, NH, BLK, VPL>) <= 160 * 1024, "LDS budget of one CU");
  const bool al16 = (p.align_bits & 15) == 0;  // 16-byte row accesses need aligned base pointers as well as strides
  const bool plain = al16 && p.xdtype == 0 && p.V == 256 * VPL && p.xst == p.V && p.gst == p.V;  // frame stride folded into the addressing
  const dim3 grid(p.B), block(64 * (4 + 2 * NH));
  if (plain)
    hipLaunchKernelGGL((fused5::fused5_kernel<CTC_FUSED_KIND, NL, NH, BLK, VPL, 0>), grid, block, 0, st, p, L, a, b, lp, stats, loss,
                       d_loss, grad, stamp, perm, only_if);
  else if (al16 && p.xdtype == 0 && ((p.V | p.xsb | p.xst | p.gsb | p.gst) & 3) == 0)
    hipLaunchKernelGGL((fused5::fused5_kernel<CTC_FUSED_KIND, NL, NH, BLK, VPL, 1>), grid, block, 0, st, p, L, a, b, lp, stats, loss,
                       d_loss, grad, stamp, perm, only_if);
  else if (p.xdtype == 0)  // vocabulary or strides not a multiple of 4 elements: element-wise row accesses
    hipLaunchKernelGGL((fused5::fused5_kernel<CTC_FUSED_KIND, NL, NH, BLK, VPL, 3>), grid, block, 0, st, p, L, a, b, lp, stats, loss,
                       d_loss, grad, stamp, perm, only_if);
  else
    hipLaunchKernelGGL((fused5::fused5_kernel<CTC_FUSED_KIND, NL, NH, BLK, VPL, 2>), grid, block, 0, st, p, L, a, b, lp, stats, loss,
                       d_loss, grad, stamp, perm, only_if);
  return hipGetLastError();
}

// One translation unit per (lattice kind, label positions per lane): -DCTC_FUSED_KIND=0|1 -DCTC_FUSED5_NL=1|2|4 (the
// instantiations are large; split like this they compile in parallel).  Exported: run_fused5_<kind>_nl<NL>.
#ifndef CTC_FUSED5_NL
#error "compile with -DCTC_FUSED5_NL=1, 2 or 4"
#endif
#define CTC_F5_CAT2(a, b, c) a##b##c
#define CTC_F5_CAT(a, b, c) CTC_F5_CAT2(a, b, c)
#if CTC_FUSED_KIND == 0
#define CTC_F5_ENTRY CTC_F5_CAT(run_fused5_classic, _nl, CTC_FUSED5_NL)
#else
#define CTC_F5_ENTRY CTC_F5_CAT(run_fused5_simplified, _nl, CTC_FUSED5_NL)
#endif
hipError_t CTC_F5_ENTRY(const Problem &p, const Layout &L, char *ws, float *loss, const float *d_loss, float *grad,
                        bool only_flagged, hipStream_t st) {
  float *alpha = reinterpret_cast<float *>(ws + L.off_alpha);
  float *beta = reinterpret_cast<float *>(ws + L.off_beta);
  double *logp = reinterpret_cast<double *>(ws + L.off_logp);
  float2 *stats = reinterpret_cast<float2 *>(ws + L.off_emis);  // the emission region of the v1 pipeline is free here
  void *stamp = ws + L.off_dummy;  // diagnostic builds (-DCTC_FUSED_STAMPS) write per-wavefront cycle counts here
  if (L.NL != CTC_FUSED5_NL) return hipErrorInvalidValue;
  // more utterances than CUs: longest first (one small kernel; skipped for batches that fit the chip in one go)
  int *perm = nullptr;
  // behind ctc_fused6.hip: only the utterances whose flag it set (normally none: the workgroups leave at once)
  const int *only_if = only_flagged ? reinterpret_cast<const int *>(ws + L.off_flags) : nullptr;
#ifndef CTC_FUSED_STAMPS
  if (!only_flagged && p.B > 256 && p.B <= 8192) {
    hipError_t e = run_order(p, L, ws, st);
    if (e != hipSuccess) return e;
    perm = reinterpret_cast<int *>(ws + L.off_perm);
  }
#endif
  // Vocabularies of 257 .. 512 tokens: two 16-byte segments of the logits row per lane (VPL = 2); the five-block ring of
  // logits rows then needs the 256-register budget of the 8-wavefront configuration (6-frame blocks, two helpers a side).
  // 129 .. 256 label positions (four per lane): the LDS rows are twice as long -- the 8-wavefront configuration as well.
  // 513 .. 1024 tokens: four segments per lane; the G stage re-reads its logits rows (no room for the five-block ring).
#if CTC_FUSED5_NL == 8
  // 257 .. 512 label positions (eight per lane): LDS rows of 4 KB leave room for 3-frame blocks, one helper a side (6 wavefronts)
  return p.V <= 256 ? launch5<8, 1, 3, 1>(p, L, alpha, beta, logp, stats, loss, d_loss, grad, stamp, perm, only_if, st)
                    : launch5<8, 1, 3, 2>(p, L, alpha, beta, logp, stats, loss, d_loss, grad, stamp, perm, only_if, st);
#elif CTC_FUSED5_NL == 4
  return p.V <= 256 ? launch5<4, 2, 6, 1>(p, L, alpha, beta, logp, stats, loss, d_loss, grad, stamp, perm, only_if, st)
                    : launch5<4, 2, 6, 2>(p, L, alpha, beta, logp, stats, loss, d_loss, grad, stamp, perm, only_if, st);
#else
  return p.V <= 256   ? launch5<CTC_FUSED5_NL, 4, 12, 1>(p, L, alpha, beta, logp, stats, loss, d_loss, grad, stamp, perm, only_if, st)
         : p.V <= 512 ? launch5<CTC_FUSED5_NL, 2, 6, 2>(p, L, alpha, beta, logp, stats, loss, d_loss, grad, stamp, perm, only_if, st)
                      : launch5<CTC_FUSED5_NL, 2, 6, 4>(p, L, alpha, beta, logp, stats, loss, d_loss, grad, stamp, perm, only_if, st);
#endif
}

}  // namespace ctc

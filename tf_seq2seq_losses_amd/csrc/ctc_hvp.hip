// Hessian-vector product out[b,t,k] = sum_{t2,k2} H[b,t,k,t2,k2] vec[b,t2,k2] WITHOUT materialising the Hessian.
//
// Replaces gradient_fn.backprop (base_loss.py:157-175), which in the reference contracts a materialised
// [B,T,V,T,V] Hessian (and therefore needs O(T^2 V^2) memory and the O(T^2 L^2) gamma tensor).  Here the product is the
// directional derivative of the gradient along u = vec, obtained by tangent-mode (forward-mode) differentiation of the
// alpha/beta recursion -- O(T L) memory and O(T L) work per utterance, so second-order products work at T = 1000.
//
//   every lattice value r = log(2^a1 + 2^a2) has the tangent  dr = d2 + w (d1 - d2),  w = 2^(a1 - r)
//   (the log-sum-exp's softmax weights; the values r are already in the workspace from the emit -> scan pipeline, so
//   a weight costs one exp2 and the tangent sweep needs no logarithm);
//   posterior of lattice state s at frame t: q_s = 2^(alpha_s + beta_s - log2 P),  dq_s = q_s (dalpha_s + dbeta_s - dlogP);
//   H_lp u = -scatter_by_token(dq)   (H_lp = Hessian w.r.t. log-probabilities, base_loss.py:186-260);
//   for logits:  H_x v = H_lp v + s (.) v - s (s . v) per frame (s = softmax; rows/columns of H_lp sum to zero,
//   README.md:58-71 is the batch_jacobian this reproduces when contracted with v).
// Tangents are kept in natural-log units; lattice values stay in base-2 logs like everywhere else.
//
// Pipeline: emit_kernel -> scan_kernel (values, ctc_kernels.hip) -> temit_kernel (tangent emissions) -> tscan_kernel
// (tangent sweeps, one wavefront per utterance x direction, next step prefetched) -> hvp_out_kernel (wave per frame).
#include "ctc_fused_common.h"
#include "ctc_hvp_fused.h"
#include "ctc_hvp_device.h"

namespace ctc {

using namespace ctc::fused;

// offset (inside the extra part of the workspace) of the fused kernel's flag words: diagnostics
size_t hvp_fused_flags_offset(int kind, int B, int T, int U) {
  Layout L = make_layout(kind, B, T, U, 0);
  return make_hvp_layout(L, B, T).total + make_hvp_fused_layout(B, T, U).off_flags;
}
size_t hvp_extra_bytes(int kind, int B, int T, int V, int U) {
  Layout L = make_layout(kind, B, T, U, 0);
  // (the fused kernel's region sits behind the log-domain pipeline's: the latter still serves the utterances the former flags)
  return make_hvp_layout(L, B, T).total + (hvp_fused_shape(0, B, T, V, U) ? make_hvp_fused_layout(B, T, U).total : 0);
}

__global__ __launch_bounds__(256) void temit_kernel(Problem p, Layout L, const float *__restrict__ emis,
                                                     const float *__restrict__ vec, float *__restrict__ demis) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (row >= (long)p.B * p.T) return;
  temit_row(p, L, emis, vec, demis, (int)(row / p.T), (int)(row % p.T), lane);
}
// One wavefront per (utterance, direction): tangent sweep (tscan_body, ctc_hvp_device.h).
template <int KIND, int NL>
__global__ __launch_bounds__(64) void tscan_kernel(Problem p, Layout L, const float *__restrict__ emis,
                                                    const float *__restrict__ demis, const float *__restrict__ alpha,
                                                    const float *__restrict__ beta, const double *__restrict__ logp,
                                                    float *__restrict__ dalpha, float *__restrict__ dbeta,
                                                    float *__restrict__ dlogp) {
  tscan_body<KIND, NL>(p, L, emis, demis, alpha, beta, logp, dalpha, dbeta, dlogp, blockIdx.x, blockIdx.y, threadIdx.x);
}

template <int KIND>
__global__ __launch_bounds__(256) void hvp_out_kernel(Problem p, Layout L, const float *__restrict__ emis,
                                                       const float *__restrict__ demis, const float *__restrict__ alpha,
                                                       const float *__restrict__ beta, const float *__restrict__ dalpha,
                                                       const float *__restrict__ dbeta, const double *__restrict__ logp,
                                                       const float *__restrict__ dlogp, const float *__restrict__ vec,
                                                       float *__restrict__ out, int wpb) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: keeps everything derived from it in SGPRs
  float *bin = lds + (long)w * p.V;
  const long row = (long)blockIdx.x * wpb + w;
  if (row >= (long)p.B * p.T) return;
  hvp_out_row<KIND>(p, L, emis, demis, alpha, beta, dalpha, dbeta, logp, dlogp, vec, out, bin, (int)(row / p.T), (int)(row % p.T), lane);
}

template <int KIND>
static hipError_t launch_hvp(const Problem &p, const Layout &L, const float *emis, float *demis, const float *alpha,
                             const float *beta, float *dalpha, float *dbeta, const double *logp, float *dlogp,
                             const float *vec, float *out, hipStream_t st) {
  dim3 grid(p.B, 2), block(64);
  switch (L.NL) {
    case 1: hipLaunchKernelGGL((tscan_kernel<KIND, 1>), grid, block, 0, st, p, L, emis, demis, alpha, beta, logp, dalpha, dbeta, dlogp); break;
    case 2: hipLaunchKernelGGL((tscan_kernel<KIND, 2>), grid, block, 0, st, p, L, emis, demis, alpha, beta, logp, dalpha, dbeta, dlogp); break;
    case 4: hipLaunchKernelGGL((tscan_kernel<KIND, 4>), grid, block, 0, st, p, L, emis, demis, alpha, beta, logp, dalpha, dbeta, dlogp); break;
    case 8: hipLaunchKernelGGL((tscan_kernel<KIND, 8>), grid, block, 0, st, p, L, emis, demis, alpha, beta, logp, dalpha, dbeta, dlogp); break;
    case 16: hipLaunchKernelGGL((tscan_kernel<KIND, 16>), grid, block, 0, st, p, L, emis, demis, alpha, beta, logp, dalpha, dbeta, dlogp); break;
    default: return hipErrorInvalidValue;
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  const long rows = (long)p.B * p.T;
  int wpb = 4;
  while (wpb > 1 && (size_t)wpb * p.V * 4 > 64 * 1024) wpb >>= 1;
  hipLaunchKernelGGL(hvp_out_kernel<KIND>, dim3((unsigned)((rows + wpb - 1) / wpb)), dim3(64 * wpb), (size_t)wpb * p.V * 4, st, p,
                     L, emis, demis, alpha, beta, dalpha, dbeta, logp, dlogp, vec, out, wpb);
  return hipGetLastError();
}

// values must already be in the workspace (run_emit_scan with both directions)
hipError_t run_hvp(const Problem &p, const Layout &L, char *ws, const float *vec, float *out, hipStream_t st) {
  const long rows = (long)p.B * p.T;
  if (rows == 0) return hipSuccess;
  const HvpLayout H = make_hvp_layout(L, p.B, p.T);
  const float *emis = reinterpret_cast<const float *>(ws + L.off_emis);
  const float *alpha = reinterpret_cast<const float *>(ws + L.off_alpha);
  const float *beta = reinterpret_cast<const float *>(ws + L.off_beta);
  const double *logp = reinterpret_cast<const double *>(ws + L.off_logp);
  char *ex = ws + L.off_extra;
  float *demis = reinterpret_cast<float *>(ex + H.off_demis);
  float *dalpha = reinterpret_cast<float *>(ex + H.off_dalpha);
  float *dbeta = reinterpret_cast<float *>(ex + H.off_dbeta);
  float *dlogp = reinterpret_cast<float *>(ex + H.off_dlogp);
  // (four rows per wavefront, as emit4_kernel does, was tried here and lost: 195 against 172 us at the north-star shape)
  hipLaunchKernelGGL(temit_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, p, L, emis, vec, demis);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return p.kind == 0 ? launch_hvp<0>(p, L, emis, demis, alpha, beta, dalpha, dbeta, logp, dlogp, vec, out, st)
                     : launch_hvp<1>(p, L, emis, demis, alpha, beta, dalpha, dbeta, logp, dlogp, vec, out, st);
}

}  // namespace ctc

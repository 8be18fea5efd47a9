// C ABI of libctc_amd.so (see include/ctc_amd.h for the contract of every entry point).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ctc_amd.h"
#include "ctc_common.h"
#include "ctc_hvp_fused.h"

namespace ctc {
hipError_t run_emit_scan(const Problem &p, const Layout &L, char *ws, float *loss, int ndir, hipStream_t st);
hipError_t run_grad(const Problem &p, const Layout &L, char *ws, const float *d_loss, float *grad, hipStream_t st);
hipError_t run_sum_loss_fixed(const float *loss, int B, long long *acc, long long *zero_next, hipStream_t st);
hipError_t run_log_posterior(const Problem &p, const Layout &L, char *ws, float *out, hipStream_t st);
hipError_t run_convert(const Problem &p, const Layout &L, char *ws, float *alpha_out, float *beta_out, hipStream_t st);
#define CTC_F5_DECL(name) hipError_t name(const Problem &p, const Layout &L, char *ws, float *loss, const float *d_loss, float *grad, bool only_flagged, hipStream_t st)
CTC_F5_DECL(run_fused5_classic_nl1); CTC_F5_DECL(run_fused5_classic_nl2); CTC_F5_DECL(run_fused5_classic_nl4); CTC_F5_DECL(run_fused5_classic_nl8);
CTC_F5_DECL(run_fused5_simplified_nl1); CTC_F5_DECL(run_fused5_simplified_nl2); CTC_F5_DECL(run_fused5_simplified_nl4); CTC_F5_DECL(run_fused5_simplified_nl8);
#undef CTC_F5_DECL
#define CTC_F6_DECL(name) hipError_t name(const Problem &p, const Layout &L, char *ws, float *loss, const float *d_loss, float *grad, hipStream_t st)
CTC_F6_DECL(run_fused6_classic_nl1); CTC_F6_DECL(run_fused6_classic_nl2); CTC_F6_DECL(run_fused6_classic_nl4); CTC_F6_DECL(run_fused6_classic_nl8);
CTC_F6_DECL(run_fused6_simplified_nl1); CTC_F6_DECL(run_fused6_simplified_nl2); CTC_F6_DECL(run_fused6_simplified_nl4); CTC_F6_DECL(run_fused6_simplified_nl8);
#undef CTC_F6_DECL
// shapes the checkpoint + recompute kernel (ctc_fused5.hip) is instantiated for: logits input, V <= 512 (smaller
// vocabularies run with the lanes beyond V masked; V or strides not a multiple of 4: element-wise row accesses), U <= 256
inline bool plain_format(const Problem &p) {  // contiguous float32 [B,T,V] logits and gradient
  return p.xdtype == 0 && p.gdtype == 0 && p.xst == p.V && p.gst == p.V && p.xsb == (long)p.T * p.V && p.gsb == (long)p.T * p.V;
}
inline bool fused5_eligible(const Problem &p, const Layout &L) {
  // producer formats: both tensors float32 or both bfloat16, strides keeping the 16-byte (8-byte) row accesses aligned
  // (bfloat16 needs 8-byte aligned rows: V and the strides multiples of 4; float32 takes any V <= 256 and any stride)
  // (U <= 512: eight label positions per lane, 3-frame blocks)
  return p.wrt == 0 && (p.V <= 512 || (p.V <= 1024 && L.NL <= 2)) && L.NL <= 8 && p.B > 0 && p.T > 0 && p.xdtype == p.gdtype &&
         p.xdtype <= 1 && p.row0 == nullptr &&  // (float16 and packed batches: three-kernel pipeline)
         (p.xdtype == 0 || (((p.V | p.xsb | p.xst | p.gsb | p.gst) & 3) == 0 && (p.align_bits & 7) == 0));
}
inline hipError_t run_fused5(const Problem &p, const Layout &L, char *ws, float *loss, const float *d_loss, float *grad, bool only_flagged, hipStream_t st) {
  switch (L.NL) {  // one translation unit of ctc_fused5.hip per (kind, label positions per lane)
    case 1: return p.kind == 0 ? run_fused5_classic_nl1(p, L, ws, loss, d_loss, grad, only_flagged, st) : run_fused5_simplified_nl1(p, L, ws, loss, d_loss, grad, only_flagged, st);
    case 2: return p.kind == 0 ? run_fused5_classic_nl2(p, L, ws, loss, d_loss, grad, only_flagged, st) : run_fused5_simplified_nl2(p, L, ws, loss, d_loss, grad, only_flagged, st);
    case 4: return p.kind == 0 ? run_fused5_classic_nl4(p, L, ws, loss, d_loss, grad, only_flagged, st) : run_fused5_simplified_nl4(p, L, ws, loss, d_loss, grad, only_flagged, st);
    case 8: return p.kind == 0 ? run_fused5_classic_nl8(p, L, ws, loss, d_loss, grad, only_flagged, st) : run_fused5_simplified_nl8(p, L, ws, loss, d_loss, grad, only_flagged, st);
    default: return hipErrorInvalidValue;
  }
}
inline bool fused6_eligible(const Problem &p, const Layout &L) { return fused5_eligible(p, L); }
// The linear-domain kernel (ctc_fused6.hip) covers the shapes of fused5; it used to be followed by a fused5 launch restricted to the
// utterances it flagged (dynamic range beyond float32 mantissas with per-lane exponents; normally none).
inline hipError_t run_fused6(const Problem &p, const Layout &L, char *ws, float *loss, const float *d_loss, float *grad, hipStream_t st) {
  hipError_t e;
  switch (L.NL) {
    case 1: e = p.kind == 0 ? run_fused6_classic_nl1(p, L, ws, loss, d_loss, grad, st) : run_fused6_simplified_nl1(p, L, ws, loss, d_loss, grad, st); break;
    case 2: e = p.kind == 0 ? run_fused6_classic_nl2(p, L, ws, loss, d_loss, grad, st) : run_fused6_simplified_nl2(p, L, ws, loss, d_loss, grad, st); break;
    case 4: e = p.kind == 0 ? run_fused6_classic_nl4(p, L, ws, loss, d_loss, grad, st) : run_fused6_simplified_nl4(p, L, ws, loss, d_loss, grad, st); break;
    case 8: e = p.kind == 0 ? run_fused6_classic_nl8(p, L, ws, loss, d_loss, grad, st) : run_fused6_simplified_nl8(p, L, ws, loss, d_loss, grad, st); break;
    default: return hipErrorInvalidValue;
  }
  return e;  // (flagged utterances are redone in the log domain inside the same launch: ctc_fused6.hip, end of fused6_kernel)
}
#ifdef CTC_WIDE_EXPERIMENT
// EXPERIMENTAL, parked outside the product tree (experiments/wide/, built by experiments/wide/build_wide_variant.sh): vocabularies beyond the fused tiers with the
// emission, chain and gradient stages beside each other in one persistent launch (experiments/wide/ctc_wide.hip).  Not part of the product library:
// at parity with the three kernels at best, and a soak run found isolated stale rows in large batches.
bool wide_eligible(const Problem &p, const Layout &L);
hipError_t run_wide(const Problem &p, const Layout &L, char *ws, float *loss, const float *d_loss, float *grad, hipStream_t st);
#endif
size_t hessian_extra_bytes(int kind, int B, int T, int V, int U);
hipError_t run_hessian(const Problem &p, const Layout &L, char *ws, const float *grad, float *hess, hipStream_t st);
size_t hvp_extra_bytes(int kind, int B, int T, int V, int U);
size_t hvp_fused_flags_offset(int kind, int B, int T, int U);
// diagnostic overrides (ctc_amd_debug_override): process-wide, written only by tests / benchmarks between calls
int g_force_pipeline = 0;      // 0 = best eligible tier, 1 = v1 (three kernels), 5 = fused5 (log domain) (7 = the parked wide tier, experiments/wide/ builds only)
int g_force_hessian_slab = 0;  // 1 = the general one-slab-per-wavefront Hessian kernel also for short labels
int g_force_hvp_v1 = 0;        // 1 = the log-domain Hessian-vector pipeline also where the fused kernel applies
#ifdef CTC_WIDE_EXPERIMENT
extern int g_wide_diag;        // timing diagnostics of the wide-vocabulary kernel (ctc_wide.hip; results are then meaningless)
#endif
int g_hvp_diag = 0;            // timing diagnostics of the fused kernel (ctc_hvp_fused.hip `mode`; results are then meaningless)
hipError_t run_hvp(const Problem &p, const Layout &L, char *ws, const float *vec, float *out, hipStream_t st);
hipError_t run_hvp_fused_classic(const Problem &p, const Layout &L, char *ws, const float *vec, float *loss, float *out, int mode, hipStream_t st);
hipError_t run_hvp_fused_simplified(const Problem &p, const Layout &L, char *ws, const float *vec, float *loss, float *out, int mode, hipStream_t st);
hipError_t run_reduce_loss(const float *loss, int B, float *out, hipStream_t st);
hipError_t run_probe_copy(void *dst, const void *src, size_t bytes, hipStream_t st);
hipError_t run_probe_spin(int threads, int lds_bytes, float us, hipStream_t st);
hipError_t run_check_labels(const int32_t *labels, int label_stride, const int32_t *label_length, int blank, int B, int V, int U,
                            int *bad, hipStream_t st);
}  // namespace ctc

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

// Limits (include/ctc_amd.h "Limits"): CTC_AMD_EINVAL beyond them
constexpr int MAX_U = CTC_AMD_MAX_U;            // scan_kernel is instantiated for up to 16 label positions per lane
constexpr int MAX_V_GRAD = CTC_AMD_MAX_V;       // one LDS token row per wavefront (64 KB)
constexpr int MAX_V_HESS = CTC_AMD_MAX_V_HESSIAN;  // the Hessian / HVP kernels keep V + 4 floats of LDS per wavefront

int check_common(int kind, int wrt, const float *logits, const int32_t *labels, int label_stride,
                 const int32_t *label_length, const int32_t *logit_length, int blank, int B, int T, int V, int U) {
  if (kind != CTC_AMD_CLASSIC && kind != CTC_AMD_SIMPLIFIED) return fail(CTC_AMD_EINVAL, "kind must be 0 (classic) or 1 (simplified), got %d", kind);
  if (wrt != CTC_AMD_WRT_LOGITS && wrt != CTC_AMD_WRT_LOGPROBS) return fail(CTC_AMD_EINVAL, "wrt must be 0 (logits) or 1 (logprobs), got %d", wrt);
  if (B < 0 || T < 0 || V <= 0 || U < 0 || label_stride < 0) return fail(CTC_AMD_EINVAL, "negative size: B=%d T=%d V=%d U=%d label_stride=%d", B, T, V, U, label_stride);
  if (blank < 0 || blank >= V) return fail(CTC_AMD_EINVAL, "blank_index %d outside [0, %d)", blank, V);
  if (U > MAX_U) return fail(CTC_AMD_EINVAL, "U=%d exceeds the supported maximum %d", U, MAX_U);
  if (B > 0 && (!label_length || !logit_length)) return fail(CTC_AMD_EINVAL, "null length pointer");
  if (B > 0 && T > 0 && !logits) return fail(CTC_AMD_EINVAL, "null logits pointer");
  if (B > 0 && label_stride > 0 && !labels) return fail(CTC_AMD_EINVAL, "null labels pointer");
  if (B > 65535 * 32767) return fail(CTC_AMD_EINVAL, "B too large");
  return CTC_AMD_OK;
}

ctc::Problem make_problem(int kind, int wrt, const float *logits, const int32_t *labels, int label_stride,
                          const int32_t *label_length, const int32_t *logit_length, int blank, int B, int T, int V, int U) {
  ctc::Problem p;
  p.logits = logits; p.labels = labels; p.label_length = label_length; p.logit_length = logit_length;
  p.label_stride = label_stride; p.blank = blank; p.B = B; p.T = T; p.V = V; p.U = U; p.kind = kind; p.wrt = wrt;
  p.xsb = (long)T * V; p.xst = V; p.gsb = (long)T * V; p.gst = V; p.xdtype = 0; p.gdtype = 0;  // contiguous float32
  p.align_bits = (int)(reinterpret_cast<uintptr_t>(logits) & 15);  // (entry points OR in the tensors they write)
  return p;
}

int hip_fail(hipError_t e, const char *where) { return fail(CTC_AMD_EHIP, "%s: %s", where, hipGetErrorString(e)); }

int low_bits(const void *a, const void *b = nullptr, const void *c = nullptr, const void *d = nullptr) {
  return (int)((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) | reinterpret_cast<uintptr_t>(d)) & 15);
}

}  // namespace

extern "C" {

int ctc_amd_abi_version(void) { return CTC_AMD_ABI_VERSION; }

const char *ctc_amd_last_error(void) { return g_err; }

static const char *select_pipeline(const ctc::Problem &p, const ctc::Layout &L, bool want_grad) {
  // pipeline selection by shape eligibility: fused6 (ctc_fused6.hip: linear-domain chains + recompute chains + helpers,
  // followed by a fused5 launch for the utterances it flags) = fused5 (ctc_fused5.hip: the same decomposition in the log
  // domain) > v1 (emit -> scan -> grad; r01's two-wavefront "fused2" tier is gone: it served one corner, V = 1024 with
  // 128 < U <= 256, which the three-kernel pipeline now takes).  A lower tier can be forced through ctc_amd_debug_override
  // (the parity tests run all of them); nothing here reads the environment.
  const int forced = ctc::g_force_pipeline;
  if (forced == 1) return "v1";
  // loss only (grad == NULL): fused5 / fused6 stop at the meeting point of their two chains
  (void)want_grad;
  if (forced == 0 && ctc::fused6_eligible(p, L)) return "fused6";
  if ((forced == 0 || forced == 5) && ctc::fused5_eligible(p, L)) return "fused5";
#ifdef CTC_WIDE_EXPERIMENT
  // "wide" (experiments/wide/ctc_wide.hip: the three stages of v1 beside each other in ONE persistent launch, V > 1024, with a gradient):
  // experimental, diagnostic builds only
  if (forced == 7 && want_grad && ctc::wide_eligible(p, L)) return "wide";
#endif
  return "v1";
}

// The fused tiers keep one checkpoint row per block and 8 bytes of statistics per frame: their layout is the compact one
// (ctc_common.h Layout::ck_blk); every other pipeline needs full lattice rows.
static ctc::Layout layout_for(const ctc::Problem &p, const char *pl) {
  const bool compact = pl[0] == 'f' && (pl[5] == '5' || pl[5] == '6');
  return ctc::make_layout(p.kind, p.B, p.T, p.U, 0, compact ? ctc::fused_blk(ctc::nl_for(p.U), p.V) : 0);
}

int ctc_amd_debug_override(const char *key, const char *value) {
  if (!key || !value) return fail(CTC_AMD_EINVAL, "null key/value");
  if (!strcmp(key, "pipeline")) {
    int f = !strcmp(value, "") ? 0 : !strcmp(value, "v1") ? 1 : !strcmp(value, "fused5") ? 5 : -1;
#ifdef CTC_WIDE_EXPERIMENT
    if (!strcmp(value, "wide")) f = 7;  // (experiments/wide/ only)
#endif
    if (f < 0) return fail(CTC_AMD_EINVAL, "pipeline override must be \"\", \"v1\" or \"fused5\", got \"%s\"", value);
    ctc::g_force_pipeline = f;
    return CTC_AMD_OK;
  }
  if (!strcmp(key, "hessian")) {
    if (strcmp(value, "") && strcmp(value, "slab")) return fail(CTC_AMD_EINVAL, "hessian override must be \"\" or \"slab\", got \"%s\"", value);
    ctc::g_force_hessian_slab = !strcmp(value, "slab");
    return CTC_AMD_OK;
  }
  if (!strcmp(key, "hvp")) {
    int dm = 0;
    if (!strncmp(value, "diag", 4) && sscanf(value + 4, "%d", &dm) == 1 && dm >= 0 && dm <= 31) {  // timing diagnostics (scripts/hvp_time.py)
      ctc::g_hvp_diag = dm;
      ctc::g_force_hvp_v1 = 0;
      return CTC_AMD_OK;
    }
    if (strcmp(value, "") && strcmp(value, "v1")) return fail(CTC_AMD_EINVAL, "hvp override must be \"\" or \"v1\", got \"%s\"", value);
    ctc::g_force_hvp_v1 = !strcmp(value, "v1");
    ctc::g_hvp_diag = 0;
    return CTC_AMD_OK;
  }
#ifdef CTC_WIDE_EXPERIMENT
  if (!strcmp(key, "wide")) {  // timing diagnostics (scripts/wide_time.py): "" or "diag<number 0..511>"
    int v = 0;
    if (strcmp(value, "") && (sscanf(value, "diag%d", &v) != 1 || v < 0 || v > 511)) return fail(CTC_AMD_EINVAL, "wide override must be \"\" or \"diag0\"..\"diag511\", got \"%s\"", value);
    ctc::g_wide_diag = v;
    return CTC_AMD_OK;
  }
#endif
  return fail(CTC_AMD_EINVAL, "unknown override key \"%s\"", key);
}

const char *ctc_amd_pipeline_name(int kind, int wrt, int B, int T, int V, int U, int want_grad) {
  if ((kind != 0 && kind != 1) || B < 0 || T < 0 || V <= 0 || U < 0 || U > MAX_U) return "invalid";
  ctc::Layout L = ctc::make_layout(kind, B, T, U, 0);
  ctc::Problem p = make_problem(kind, wrt, nullptr, nullptr, 0, nullptr, nullptr, 0, B, T, V, U);
  return select_pipeline(p, L, want_grad != 0);
}

int ctc_amd_debug_flags_offset(int kind, int B, int T, int V, int U, size_t *out_offset) {
  if (!out_offset) return fail(CTC_AMD_EINVAL, "out_offset is null");
  if ((kind != 0 && kind != 1) || B < 0 || T < 0 || V <= 0 || U < 0 || U > MAX_U) return fail(CTC_AMD_EINVAL, "bad shape");
  ctc::Problem p = make_problem(kind, CTC_AMD_WRT_LOGITS, nullptr, nullptr, 0, nullptr, nullptr, 0, B, T, V, U);
  const char *pl = select_pipeline(p, ctc::make_layout(kind, B, T, U, 0), true);
  if (strcmp(pl, "fused6")) return fail(CTC_AMD_EINVAL, "these shapes run the %s pipeline, which keeps no flags", pl);
  *out_offset = layout_for(p, pl).off_flags;
  return CTC_AMD_OK;
}

int ctc_amd_debug_hvp_flags_offset(int kind, int B, int T, int V, int U, size_t *out_offset) {
  if (!out_offset) return fail(CTC_AMD_EINVAL, "out_offset is null");
  if ((kind != 0 && kind != 1) || B < 0 || T < 0 || V <= 0 || U < 0 || U > MAX_U) return fail(CTC_AMD_EINVAL, "bad shape");
  if (!ctc::hvp_fused_shape(CTC_AMD_WRT_LOGITS, B, T, V, U)) return fail(CTC_AMD_EINVAL, "these shapes run the log-domain Hessian-vector pipeline, which keeps no flags");
  *out_offset = ctc::make_layout(kind, B, T, U, 0).off_extra + ctc::hvp_fused_flags_offset(kind, B, T, U);
  return CTC_AMD_OK;
}

int ctc_amd_reduce_loss(const float *loss, int B, float *out2, void *stream) {
  if (B < 0 || !out2 || (B > 0 && !loss)) return fail(CTC_AMD_EINVAL, "bad arguments");
  hipError_t e = ctc::run_reduce_loss(loss, B, out2, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return hip_fail(e, "reduce launch");
  return CTC_AMD_OK;
}

int ctc_amd_probe_copy(void *dst, const void *src, size_t bytes, void *stream) {
  if (!dst || !src || (bytes & 15) || low_bits(dst, src) != 0) return fail(CTC_AMD_EINVAL, "probe copy needs 16-byte aligned pointers and size");
  hipError_t e = ctc::run_probe_copy(dst, src, bytes, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return hip_fail(e, "probe copy launch");
  return CTC_AMD_OK;
}

int ctc_amd_probe_spin(int threads, int lds_bytes, float microseconds, void *stream) {
  if (threads < 64 || threads > 1024 || (threads & 63) || lds_bytes < 0 || lds_bytes > 65536 || !(microseconds >= 0.f) || microseconds > 1000.f)
    return fail(CTC_AMD_EINVAL, "probe spin: threads in 64..1024 (multiple of 64), lds_bytes <= 65536, microseconds <= 1000");
  hipError_t e = ctc::run_probe_spin(threads, lds_bytes, microseconds, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return hip_fail(e, "probe spin launch");
  return CTC_AMD_OK;
}

int ctc_amd_check_labels(const int32_t *labels, int label_stride, const int32_t *label_length, int blank_index, int B, int V,
                         int U, void *stream) {
  if (B < 0 || V <= 0 || U < 0 || label_stride < 0) return fail(CTC_AMD_EINVAL, "negative size");
  if (B == 0 || label_stride == 0) return CTC_AMD_OK;
  if (!labels || !label_length) return fail(CTC_AMD_EINVAL, "null pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  int *bad = nullptr;
  hipError_t e = hipMallocAsync(reinterpret_cast<void **>(&bad), sizeof(int), st);
  if (e != hipSuccess) return hip_fail(e, "hipMallocAsync");
  e = ctc::run_check_labels(labels, label_stride, label_length, blank_index, B, V, U, bad, st);
  int host = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&host, bad, sizeof(int), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  hipFreeAsync(bad, st);
  if (e != hipSuccess) return hip_fail(e, "label check");
  if (host != 0) return fail(CTC_AMD_ELABEL, "%d label(s) inside label_length are outside [0, %d) or equal to blank_index %d", host, V, blank_index);
  return CTC_AMD_OK;
}

int ctc_amd_workspace_bytes(int what, int kind, int B, int T, int V, int U, size_t *out_bytes) {
  if (!out_bytes) return fail(CTC_AMD_EINVAL, "out_bytes is null");
  if (kind != 0 && kind != 1) return fail(CTC_AMD_EINVAL, "bad kind %d", kind);
  if (B < 0 || T < 0 || V <= 0 || U < 0 || U > MAX_U) return fail(CTC_AMD_EINVAL, "bad shape B=%d T=%d V=%d U=%d", B, T, V, U);
  size_t extra = 0;
  if (what == CTC_AMD_WS_LOSS_GRAD_LOGITS) {
    // the pipeline a float32 / aligned bfloat16 logits call of this shape selects; its own (smaller) layout when that is a fused tier
    ctc::Layout L0 = ctc::make_layout(kind, B, T, U, 0);
    ctc::Problem p = make_problem(kind, CTC_AMD_WRT_LOGITS, nullptr, nullptr, 0, nullptr, nullptr, 0, B, T, V, U);
    *out_bytes = layout_for(p, select_pipeline(p, L0, true)).total;
    return CTC_AMD_OK;
  }
  if (what == CTC_AMD_WS_HESSIAN) extra = ctc::hessian_extra_bytes(kind, B, T, V, U);
  else if (what == CTC_AMD_WS_HVP) extra = ctc::hvp_extra_bytes(kind, B, T, V, U);
  else if (what != CTC_AMD_WS_LOSS_GRAD && what != CTC_AMD_WS_ALPHA_BETA) return fail(CTC_AMD_EINVAL, "bad workspace selector %d", what);
  *out_bytes = ctc::make_layout(kind, B, T, U, extra).total;
  return CTC_AMD_OK;
}

static int loss_grad_impl(ctc::Problem p, float *loss, void *grad, const float *d_loss, void *workspace,
                          size_t workspace_bytes, void *stream) {
  if (!loss) return fail(CTC_AMD_EINVAL, "null loss pointer");
  if (grad && p.V > MAX_V_GRAD) return fail(CTC_AMD_EINVAL, "V=%d exceeds the supported maximum %d for the gradient", p.V, MAX_V_GRAD);
  hipStream_t st = static_cast<hipStream_t>(stream);
  float *gradf = static_cast<float *>(grad);  // element-typed inside the kernels (Problem::gdtype)
  p.align_bits = low_bits(p.logits, grad);
  const char *pl = select_pipeline(p, ctc::make_layout(p.kind, p.B, p.T, p.U, 0), grad != nullptr);
  const ctc::Layout L = layout_for(p, pl);
  if (!workspace || workspace_bytes < L.total)
    return fail(CTC_AMD_EWORKSPACE, "workspace too small for pipeline %s: %zu < %zu", pl, workspace_bytes, L.total);
  if (pl[0] == 'f') {
    char *wsb = static_cast<char *>(workspace);
    hipError_t ef = (pl[5] == '6') ? ctc::run_fused6(p, L, wsb, loss, d_loss, gradf, st)
                                   : ctc::run_fused5(p, L, wsb, loss, d_loss, gradf, false, st);
    if (ef != hipSuccess) return hip_fail(ef, pl);
    if (p.sum_out && pl[5] != '6') {  // (fused6 adds inside its launch)
      ef = ctc::run_sum_loss_fixed(loss, p.B, p.sum_out, p.sum_zero, st);
      if (ef != hipSuccess) return hip_fail(ef, "loss sum launch");
    }
    return CTC_AMD_OK;
  }
#ifdef CTC_WIDE_EXPERIMENT
  if (pl[0] == 'w') {
    hipError_t ew = ctc::run_wide(p, L, static_cast<char *>(workspace), loss, d_loss, gradf, st);
    if (ew != hipSuccess) return hip_fail(ew, "wide launch");
    if (p.sum_out) {
      ew = ctc::run_sum_loss_fixed(loss, p.B, p.sum_out, p.sum_zero, st);
      if (ew != hipSuccess) return hip_fail(ew, "loss sum launch");
    }
    return CTC_AMD_OK;
  }
#endif
  hipError_t e = ctc::run_emit_scan(p, L, static_cast<char *>(workspace), loss, grad ? 2 : 1, st);
  if (e != hipSuccess) return hip_fail(e, "emit/scan launch");
  if (grad) {
    e = ctc::run_grad(p, L, static_cast<char *>(workspace), d_loss, gradf, st);
    if (e != hipSuccess) return hip_fail(e, "grad launch");
  }
  if (p.sum_out) {
    e = ctc::run_sum_loss_fixed(loss, p.B, p.sum_out, p.sum_zero, st);
    if (e != hipSuccess) return hip_fail(e, "loss sum launch");
  }
  return CTC_AMD_OK;
}

int ctc_amd_loss_grad(int kind, int wrt, const float *logits, const int32_t *labels, int label_stride,
                      const int32_t *label_length, const int32_t *logit_length, int blank_index, int B, int T, int V,
                      int U, float *loss, float *grad, const float *d_loss, void *workspace, size_t workspace_bytes,
                      void *stream) {
  int rc = check_common(kind, wrt, logits, labels, label_stride, label_length, logit_length, blank_index, B, T, V, U);
  if (rc) return rc;
  if (B == 0) return CTC_AMD_OK;
  ctc::Problem p = make_problem(kind, wrt, logits, labels, label_stride, label_length, logit_length, blank_index, B, T, V, U);
  return loss_grad_impl(p, loss, grad, d_loss, workspace, workspace_bytes, stream);
}

int ctc_amd_loss_grad_ex(int kind, int wrt, const void *logits, int logits_dtype, int64_t logits_stride_b,
                         int64_t logits_stride_t, const int32_t *labels, int label_stride, const int32_t *label_length,
                         const int32_t *logit_length, int blank_index, int B, int T, int V, int U, float *loss, void *grad,
                         int grad_dtype, int64_t grad_stride_b, int64_t grad_stride_t, const float *d_loss, void *workspace,
                         size_t workspace_bytes, void *stream) {
  int rc = check_common(kind, wrt, static_cast<const float *>(logits), labels, label_stride, label_length, logit_length,
                        blank_index, B, T, V, U);
  if (rc) return rc;
  if (logits_dtype < CTC_AMD_F32 || logits_dtype > CTC_AMD_F16 || grad_dtype < CTC_AMD_F32 || grad_dtype > CTC_AMD_F16)
    return fail(CTC_AMD_EINVAL, "dtype must be CTC_AMD_F32, CTC_AMD_BF16 or CTC_AMD_F16 (logits %d, grad %d)", logits_dtype, grad_dtype);
  if (B == 0) return CTC_AMD_OK;
  // rows must not overlap: |stride_t| >= V, and the batch stride must step over whole rows in either nesting order
  if (logits_stride_t < V || logits_stride_b < V || (grad && (grad_stride_t < V || grad_stride_b < V)))
    return fail(CTC_AMD_EINVAL, "strides smaller than a row of V=%d elements (logits %lld/%lld, grad %lld/%lld)", V,
                (long long)logits_stride_b, (long long)logits_stride_t, (long long)grad_stride_b, (long long)grad_stride_t);
  ctc::Problem p = make_problem(kind, wrt, static_cast<const float *>(logits), labels, label_stride, label_length, logit_length,
                                blank_index, B, T, V, U);
  p.xsb = logits_stride_b; p.xst = logits_stride_t; p.xdtype = logits_dtype;
  p.gsb = grad_stride_b; p.gst = grad_stride_t; p.gdtype = grad_dtype;
  return loss_grad_impl(p, loss, grad, d_loss, workspace, workspace_bytes, stream);
}

int ctc_amd_loss_grad_packed(int kind, int wrt, const void *logits, int logits_dtype, const int64_t *row_offsets, int64_t row_stride,
                             const int32_t *labels, int label_stride, const int32_t *label_length, const int32_t *logit_length,
                             int blank_index, int B, int T, int V, int U, float *loss, void *grad, int grad_dtype,
                             int64_t grad_row_stride, const float *d_loss, void *workspace, size_t workspace_bytes, void *stream) {
  int rc = check_common(kind, wrt, static_cast<const float *>(logits), labels, label_stride, label_length, logit_length,
                        blank_index, B, T, V, U);
  if (rc) return rc;
  if (logits_dtype < CTC_AMD_F32 || logits_dtype > CTC_AMD_F16 || grad_dtype < CTC_AMD_F32 || grad_dtype > CTC_AMD_F16)
    return fail(CTC_AMD_EINVAL, "dtype must be CTC_AMD_F32, CTC_AMD_BF16 or CTC_AMD_F16 (logits %d, grad %d)", logits_dtype, grad_dtype);
  if (B == 0) return CTC_AMD_OK;
  if (!row_offsets) return fail(CTC_AMD_EINVAL, "null row_offsets pointer");
  if (row_stride < V || (grad && grad_row_stride < V))
    return fail(CTC_AMD_EINVAL, "row strides smaller than a row of V=%d elements (logits %lld, grad %lld)", V, (long long)row_stride,
                (long long)grad_row_stride);
  ctc::Problem p = make_problem(kind, wrt, static_cast<const float *>(logits), labels, label_stride, label_length, logit_length,
                                blank_index, B, T, V, U);
  p.xsb = 0; p.xst = row_stride; p.xdtype = logits_dtype;
  p.gsb = 0; p.gst = grad ? grad_row_stride : V; p.gdtype = grad_dtype;
  p.row0 = reinterpret_cast<const long long *>(row_offsets);
  return loss_grad_impl(p, loss, grad, d_loss, workspace, workspace_bytes, stream);
}

int ctc_amd_loss_grad_sum(int kind, int wrt, const void *logits, int logits_dtype, int64_t logits_stride_b,
                          int64_t logits_stride_t, const int32_t *labels, int label_stride, const int32_t *label_length,
                          const int32_t *logit_length, int blank_index, int B, int T, int V, int U, float *loss, void *grad,
                          int grad_dtype, int64_t grad_stride_b, int64_t grad_stride_t, const float *d_loss, long long *sum2,
                          long long *zero_next, void *workspace, size_t workspace_bytes, void *stream) {
  int rc = check_common(kind, wrt, static_cast<const float *>(logits), labels, label_stride, label_length, logit_length,
                        blank_index, B, T, V, U);
  if (rc) return rc;
  if (!sum2) return fail(CTC_AMD_EINVAL, "null sum2 pointer");
  if (logits_dtype < CTC_AMD_F32 || logits_dtype > CTC_AMD_F16 || grad_dtype < CTC_AMD_F32 || grad_dtype > CTC_AMD_F16)
    return fail(CTC_AMD_EINVAL, "dtype must be CTC_AMD_F32, CTC_AMD_BF16 or CTC_AMD_F16 (logits %d, grad %d)", logits_dtype, grad_dtype);
  if (logits_stride_t < V || logits_stride_b < V || (grad && (grad_stride_t < V || grad_stride_b < V)))
    return fail(CTC_AMD_EINVAL, "strides smaller than a row of V=%d elements", V);
  ctc::Problem p = make_problem(kind, wrt, static_cast<const float *>(logits), labels, label_stride, label_length, logit_length,
                                blank_index, B, T, V, U);
  p.xsb = logits_stride_b; p.xst = logits_stride_t; p.xdtype = logits_dtype;
  p.gsb = grad_stride_b; p.gst = grad_stride_t; p.gdtype = grad_dtype;
  p.sum_out = sum2; p.sum_zero = zero_next;
  if (B == 0) {  // nothing to add; the next step's buffer still has to be cleared
    if (zero_next) {
      hipError_t e = ctc::run_sum_loss_fixed(loss, 0, sum2, zero_next, static_cast<hipStream_t>(stream));
      if (e != hipSuccess) return hip_fail(e, "loss sum launch");
    }
    return CTC_AMD_OK;
  }
  return loss_grad_impl(p, loss, grad, d_loss, workspace, workspace_bytes, stream);
}

int ctc_amd_loss_forward(int kind, int wrt, const void *logits, int logits_dtype, int64_t logits_stride_b,
                         int64_t logits_stride_t, const int32_t *labels, int label_stride, const int32_t *label_length,
                         const int32_t *logit_length, int blank_index, int B, int T, int V, int U, float *loss, void *workspace,
                         size_t workspace_bytes, void *stream) {
  int rc = check_common(kind, wrt, static_cast<const float *>(logits), labels, label_stride, label_length, logit_length,
                        blank_index, B, T, V, U);
  if (rc) return rc;
  if (logits_dtype < CTC_AMD_F32 || logits_dtype > CTC_AMD_F16)
    return fail(CTC_AMD_EINVAL, "dtype must be CTC_AMD_F32, CTC_AMD_BF16 or CTC_AMD_F16 (logits %d)", logits_dtype);
  if (B == 0) return CTC_AMD_OK;
  if (logits_stride_t < V || logits_stride_b < V) return fail(CTC_AMD_EINVAL, "strides smaller than a row of V=%d elements", V);
  ctc::Problem p = make_problem(kind, wrt, static_cast<const float *>(logits), labels, label_stride, label_length, logit_length,
                                blank_index, B, T, V, U);
  p.xsb = logits_stride_b; p.xst = logits_stride_t; p.xdtype = logits_dtype;
  // (no gradient in this half: the pipeline is chosen as for a gradient in the logits' own format, which is what the front ends ask
  // ctc_amd_grad_resume for)
  p.gsb = logits_stride_b; p.gst = logits_stride_t; p.gdtype = logits_dtype;
  // (Problem::resume: 0 = a call of its own, 1 = second half of a pair, 2 = first half of a pair -- the linear-domain kernel then
  // honours its conservative loss-only signs for binding alignments only, ctc_fused6.hip; other pipelines do not look at it)
  p.resume = 2;
  return loss_grad_impl(p, loss, nullptr, nullptr, workspace, workspace_bytes, stream);
}

int ctc_amd_grad_resume(int kind, int wrt, const void *logits, int logits_dtype, int64_t logits_stride_b,
                        int64_t logits_stride_t, const int32_t *labels, int label_stride, const int32_t *label_length,
                        const int32_t *logit_length, int blank_index, int B, int T, int V, int U, float *loss, void *grad,
                        int grad_dtype, int64_t grad_stride_b, int64_t grad_stride_t, const float *d_loss, void *workspace,
                        size_t workspace_bytes, void *stream) {
  int rc = check_common(kind, wrt, static_cast<const float *>(logits), labels, label_stride, label_length, logit_length,
                        blank_index, B, T, V, U);
  if (rc) return rc;
  if (logits_dtype < CTC_AMD_F32 || logits_dtype > CTC_AMD_F16 || grad_dtype < CTC_AMD_F32 || grad_dtype > CTC_AMD_F16)
    return fail(CTC_AMD_EINVAL, "dtype must be CTC_AMD_F32, CTC_AMD_BF16 or CTC_AMD_F16 (logits %d, grad %d)", logits_dtype, grad_dtype);
  if (!grad) return fail(CTC_AMD_EINVAL, "null grad pointer");
  if (B == 0) return CTC_AMD_OK;
  if (logits_stride_t < V || logits_stride_b < V || grad_stride_t < V || grad_stride_b < V)
    return fail(CTC_AMD_EINVAL, "strides smaller than a row of V=%d elements", V);
  ctc::Problem p = make_problem(kind, wrt, static_cast<const float *>(logits), labels, label_stride, label_length, logit_length,
                                blank_index, B, T, V, U);
  p.xsb = logits_stride_b; p.xst = logits_stride_t; p.xdtype = logits_dtype;
  p.gsb = grad_stride_b; p.gst = grad_stride_t; p.gdtype = grad_dtype;
  // only the linear-domain fused kernel keeps what the second half needs; every other pipeline computes loss and gradient anew
  // (eligibility is decided on the same alignment bits the launch will see)
  p.align_bits = low_bits(p.logits, grad);
  ctc::Layout L = ctc::make_layout(p.kind, p.B, p.T, p.U, 0);
  if (!strcmp(select_pipeline(p, L, true), "fused6")) p.resume = 1;
  return loss_grad_impl(p, loss, grad, d_loss, workspace, workspace_bytes, stream);
}

int ctc_amd_alpha_beta(int kind, int wrt, const float *logits, const int32_t *labels, int label_stride,
                       const int32_t *label_length, const int32_t *logit_length, int blank_index, int B, int T, int V,
                       int U, float *loss, float *alpha, float *beta, void *workspace, size_t workspace_bytes,
                       void *stream) {
  int rc = check_common(kind, wrt, logits, labels, label_stride, label_length, logit_length, blank_index, B, T, V, U);
  if (rc) return rc;
  if (B == 0) return CTC_AMD_OK;
  if (!loss || !alpha || !beta) return fail(CTC_AMD_EINVAL, "null output pointer");
  ctc::Layout L = ctc::make_layout(kind, B, T, U, 0);
  if (!workspace || workspace_bytes < L.total) return fail(CTC_AMD_EWORKSPACE, "workspace too small: %zu < %zu", workspace_bytes, L.total);
  ctc::Problem p = make_problem(kind, wrt, logits, labels, label_stride, label_length, logit_length, blank_index, B, T, V, U);
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipError_t e = ctc::run_emit_scan(p, L, static_cast<char *>(workspace), loss, 2, st);
  if (e != hipSuccess) return hip_fail(e, "emit/scan launch");
  e = ctc::run_convert(p, L, static_cast<char *>(workspace), alpha, beta, st);
  if (e != hipSuccess) return hip_fail(e, "convert launch");
  return CTC_AMD_OK;
}

int ctc_amd_log_posterior(int kind, int wrt, const float *logits, const int32_t *labels, int label_stride,
                          const int32_t *label_length, const int32_t *logit_length, int blank_index, int B, int T, int V,
                          int U, float *loss, float *lg, void *workspace, size_t workspace_bytes, void *stream) {
  int rc = check_common(kind, wrt, logits, labels, label_stride, label_length, logit_length, blank_index, B, T, V, U);
  if (rc) return rc;
  if (B == 0) return CTC_AMD_OK;
  if (!loss || (T > 0 && !lg)) return fail(CTC_AMD_EINVAL, "null output pointer");
  if (V > 8192) return fail(CTC_AMD_EINVAL, "V=%d exceeds the supported maximum 8192 of ctc_amd_log_posterior", V);
  ctc::Layout L = ctc::make_layout(kind, B, T, U, 0);
  if (!workspace || workspace_bytes < L.total) return fail(CTC_AMD_EWORKSPACE, "workspace too small: %zu < %zu", workspace_bytes, L.total);
  ctc::Problem p = make_problem(kind, wrt, logits, labels, label_stride, label_length, logit_length, blank_index, B, T, V, U);
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipError_t e = ctc::run_emit_scan(p, L, static_cast<char *>(workspace), loss, 2, st);
  if (e != hipSuccess) return hip_fail(e, "emit/scan launch");
  e = ctc::run_log_posterior(p, L, static_cast<char *>(workspace), lg, st);
  if (e != hipSuccess) return hip_fail(e, "log posterior launch");
  return CTC_AMD_OK;
}

int ctc_amd_hessian(int kind, int wrt, const float *logits, const int32_t *labels, int label_stride,
                    const int32_t *label_length, const int32_t *logit_length, int blank_index, int B, int T, int V,
                    int U, float *loss, float *grad, float *hess, void *workspace, size_t workspace_bytes,
                    void *stream) {
  int rc = check_common(kind, wrt, logits, labels, label_stride, label_length, logit_length, blank_index, B, T, V, U);
  if (rc) return rc;
  if (B == 0) return CTC_AMD_OK;
  if (!loss || !hess) return fail(CTC_AMD_EINVAL, "null output pointer");
  if (V > MAX_V_HESS) return fail(CTC_AMD_EINVAL, "V=%d exceeds the supported maximum %d of the Hessian", V, MAX_V_HESS);
  if (low_bits(logits, grad, hess) != 0) return fail(CTC_AMD_EINVAL, "ctc_amd_hessian needs 16-byte aligned logits / grad / hess pointers");
  ctc::Layout L = ctc::make_layout(kind, B, T, U, ctc::hessian_extra_bytes(kind, B, T, V, U));
  if (!workspace || workspace_bytes < L.total) return fail(CTC_AMD_EWORKSPACE, "workspace too small: %zu < %zu", workspace_bytes, L.total);
  ctc::Problem p = make_problem(kind, wrt, logits, labels, label_stride, label_length, logit_length, blank_index, B, T, V, U);
  hipStream_t st = static_cast<hipStream_t>(stream);
  char *ws = static_cast<char *>(workspace);
  hipError_t e = ctc::run_emit_scan(p, L, ws, loss, 2, st);
  if (e != hipSuccess) return hip_fail(e, "emit/scan launch");
  // the Hessian needs the log-probability-space gradient g = -posterior; it lives in the extra workspace region
  float *g_lp = reinterpret_cast<float *>(ws + L.off_extra);
  ctc::Problem plp = p;
  plp.wrt = CTC_AMD_WRT_LOGPROBS;
  // grad_kernel only reads emis/alpha/beta/logp; with wrt = LOGPROBS it writes -posterior
  e = ctc::run_grad(plp, L, ws, nullptr, g_lp, st);
  if (e != hipSuccess) return hip_fail(e, "posterior launch");
  if (grad) {
    e = ctc::run_grad(p, L, ws, nullptr, grad, st);
    if (e != hipSuccess) return hip_fail(e, "grad launch");
  }
  e = ctc::run_hessian(p, L, ws, g_lp, hess, st);
  if (e != hipSuccess) return hip_fail(e, "hessian launch");
  return CTC_AMD_OK;
}

int ctc_amd_hvp(int kind, int wrt, const float *logits, const int32_t *labels, int label_stride,
                const int32_t *label_length, const int32_t *logit_length, int blank_index, int B, int T, int V, int U,
                const float *vec, float *loss, float *grad, float *out, void *workspace, size_t workspace_bytes,
                void *stream) {
  int rc = check_common(kind, wrt, logits, labels, label_stride, label_length, logit_length, blank_index, B, T, V, U);
  if (rc) return rc;
  if (B == 0) return CTC_AMD_OK;
  if (!loss || !out || (T > 0 && !vec)) return fail(CTC_AMD_EINVAL, "null vec/output pointer");
  if (V > MAX_V_HESS) return fail(CTC_AMD_EINVAL, "V=%d exceeds the supported maximum %d of the Hessian-vector product", V, MAX_V_HESS);
  if (low_bits(logits, vec, grad, out) != 0) return fail(CTC_AMD_EINVAL, "ctc_amd_hvp needs 16-byte aligned logits / vec / grad / out pointers");
  ctc::Layout L = ctc::make_layout(kind, B, T, U, ctc::hvp_extra_bytes(kind, B, T, V, U));
  if (!workspace || workspace_bytes < L.total) return fail(CTC_AMD_EWORKSPACE, "workspace too small: %zu < %zu", workspace_bytes, L.total);
  ctc::Problem p = make_problem(kind, wrt, logits, labels, label_stride, label_length, logit_length, blank_index, B, T, V, U);
  hipStream_t st = static_cast<hipStream_t>(stream);
  char *ws = static_cast<char *>(workspace);
  // The fused linear-domain kernel (ctc_hvp_fused.hip) where its instantiations apply: ONE launch; utterances its number format
  // cannot hold are redone by their own workgroup with the log-domain building blocks, inside the same launch.
  if (!grad && !ctc::g_force_hvp_v1 && ctc::hvp_fused_shape(wrt, B, T, V, U)) {
    hipError_t ef = kind == 0 ? ctc::run_hvp_fused_classic(p, L, ws, vec, loss, out, ctc::g_hvp_diag, st)
                              : ctc::run_hvp_fused_simplified(p, L, ws, vec, loss, out, ctc::g_hvp_diag, st);
    if (ef != hipSuccess) return hip_fail(ef, "fused hvp launch");
    return CTC_AMD_OK;
  }
  hipError_t e = ctc::run_emit_scan(p, L, ws, loss, 2 | 4, st);  // (+ 4: rows renormalised every step, for the tangent sweep)
  if (e != hipSuccess) return hip_fail(e, "emit/scan launch");
  if (grad) {
    e = ctc::run_grad(p, L, ws, nullptr, grad, st);
    if (e != hipSuccess) return hip_fail(e, "grad launch");
  }
  e = ctc::run_hvp(p, L, ws, vec, out, st);
  if (e != hipSuccess) return hip_fail(e, "hvp launch");
  return CTC_AMD_OK;
}

}  // extern "C"

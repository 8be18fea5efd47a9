// Workspace of the fused Hessian-vector product (ctc_hvp_fused.hip), placed behind the log-domain pipeline's regions inside the
// extra part of CTC_AMD_WS_HVP: checkpoint rows with tangents [B][2][nslot][RS], their lane exponents [B][2][nslot][64],
// per-frame statistics [B][T] float4 (rowmax log2 e, 1 / sum exp, softmax . v, -), flags [B] (0 = computed in the linear domain).
#pragma once
#include <stddef.h>

namespace ctc {

constexpr int HVPF_BLK = 6;       // frames per block of the fused HVP kernel
constexpr int HVPF_MAX_U = 128;   // label positions (two per lane)
constexpr int HVPF_MAX_V = 256;   // tokens (one 16-byte row segment per lane)

struct HvpFusedLayout {
  int nslot;
  size_t off_rows, off_kexp, off_stats, off_flags, total;
};

static inline HvpFusedLayout make_hvp_fused_layout(int B, int T, int U) {
  auto al = [](size_t x) { return (x + 255) & ~size_t(255); };
  HvpFusedLayout H;
  const int nl = U <= 64 ? 1 : 2;
  const size_t RS = 4 * 64 * nl + 8;
  H.nslot = (T + HVPF_BLK - 1) / HVPF_BLK + 3;
  size_t o = 0;
  H.off_rows = o;  o = al(o + (size_t)B * 2 * H.nslot * RS * 4);
  H.off_kexp = o;  o = al(o + (size_t)B * 2 * H.nslot * 64 * 4);
  H.off_stats = o; o = al(o + (size_t)B * T * 16);
  H.off_flags = o; o = al(o + (size_t)B * 4);
  H.total = o;
  return H;
}

// shapes the fused kernel is instantiated for (logits input, contiguous float32; pointer alignment is checked at the ABI)
static inline bool hvp_fused_shape(int wrt, int B, int T, int V, int U) {
  return wrt == 0 && B > 0 && T > 0 && V <= HVPF_MAX_V && (V & 3) == 0 && U <= HVPF_MAX_U;
}

}  // namespace ctc

// "wide": loss + gradient for vocabularies the fused tiers do not hold (V > 1024), ONE persistent launch.
//
// The three-kernel pipeline (ctc_kernels.hip) runs emit -> scan -> grad one after the other: at B = 32, T = 1000, V = 4096 the
// strictly sequential alpha / beta sweeps (64 wavefronts on the whole chip, pure latency) are a third of the call, and the logits
// are read twice from HBM.  Here the three stages run BESIDE each other in one grid and hand rows over through flags in global
// memory:
//
//   chain workgroups   (the first 2 B of the grid: one per utterance and direction) run the log-domain recursion of Scan
//                      (ctc_v1_device.h; classic_ctc_loss.py:310-462, simplified_ctc_loss.py:291-438) as soon as the emission rows
//                      exist.  Three wavefronts: a LOADER keeps ~24 emission rows in flight (the memory system is saturated by
//                      the stream workgroups: a load takes microseconds) and stages them in an LDS ring; the CHAIN touches LDS
//                      only -- emission row in, lattice row out; a STORER writes the lattice rows to memory and announces them;
//   stream workgroups  (all others) first compute emission rows (emit_row: softmax statistics + the label gathers, base_loss.py:59,
//                      328-344), four frames per task, in an order that walks every utterance from both ends towards the middle --
//                      the order in which the two chains consume them; then gradient rows (posterior scatter + softmax - posterior,
//                      classic_ctc_loss.py:565-669, simplified_ctc_loss.py:456-534, base_loss.py:262-298), from the middle outwards --
//                      the order in which alpha and beta rows of the same frame become available once the chains have crossed.
//
// So the sweeps hide behind the two HBM-bound passes instead of standing between them, and the second read of the logits happens in
// the reverse order of the first (what the Infinity Cache still holds of the first pass is what the second pass asks for first).
// The posterior of a frame is normalised by the frame's own mass sum_s alpha_t[s] beta_t[s] (= P for every t, the invariant of the
// reference's tests/test_classic_ctc_loss.py:146-167): a gradient row needs no log P from the end of the alpha sweep, and the
// row offsets cancel (no double-precision sums).
//
// Progress / deadlock: every workgroup of the grid is resident (the host sizes the grid from the occupancy query); emission tasks
// wait for nothing; a chain waits only for emission tasks; a gradient task waits only for chains; a stream workgroup finishes all
// its emission tasks before its first gradient task.  Every wait is bounded (WAIT_LIMIT polls): a wave that gives up raises the
// abort word, which ends every other wait at once -- the launch then returns with NaN losses instead of hanging.
// Visibility across the eight XCDs (one L2 each, not coherent with one another inside a kernel): every row that crosses workgroups
// (emission rows, lattice rows) is written through and read past the L2s (sc1 accesses, ctc_common.h ld1 / st1 ...), flags are
// agent-scope atomics; a producer waits for its stores (s_waitcnt vmcnt(0)) before it raises a flag, a consumer issues its loads
// after it has seen the flag.  (First version: plain accesses + agent-scope release / acquire fences, i.e. an L2 write-back or
// invalidate per task: 2.0 ms at B = 32, T = 1000, V = 4096 against 0.47 ms for the three kernels.)
#include "ctc_common.h"
#include "ctc_amd.h"
#include "ctc_v1_device.h"
#include "ctc_grad_row.h"

namespace ctc {
// Timing diagnostics (results are then meaningless) exist in CTC_DIAG builds only: ctc_amd_debug_override("wide",
// "diagN"), N a bit set -- 1: the storer announces rows without waiting for them, 2: lattice rows are dropped, 4: chains alone
// (nobody computes emissions), 8: the loader stages whatever the ring holds, 16: no gradient pass, 32 / 64: an agent-scope
// release fence before every emission-counter bump / every announcement of lattice rows (is visibility the problem?).
#ifdef CTC_DIAG
int g_wide_diag = 0;
#define CTC_WIDE_DIAG(bits) ((diag & (bits)) != 0)
#else
#define CTC_WIDE_DIAG(bits) false
#endif
namespace wide {

constexpr int CHUNK = 64;            // frames per emission counter
constexpr int WAIT_LIMIT = 1 << 21;  // polls (each >= 0.3 us) before a wait gives up

// sync words (int32), zeroed by the host before the launch:
//   [0] abort; [1] row tickets of the stream wavefronts; [16 + b * W ...): per utterance W = NC + 2 words: NC emission counters (valid rows written per 64-frame chunk),
//   then the number of alpha rows / beta rows in memory
struct SyncView {
  int *base;
  int NC;
  __device__ __forceinline__ int *abort_word() const { return base; }
  __device__ __forceinline__ int *cnt(int b) const { return base + 16 + (long)b * (NC + 2); }
  __device__ __forceinline__ int *rows_done(int b, int dir) const { return cnt(b) + NC + dir; }
};
inline size_t sync_bytes(int B, int T) { return (size_t)(16 + (long)B * ((T + CHUNK - 1) / CHUNK + 1 + 2)) * 4; }

__device__ __forceinline__ int ld_relaxed(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// every vector-memory operation of this wavefront issued so far has completed (stores: written through); also a compiler barrier
__device__ __forceinline__ void stores_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// wave-uniform wait until *p >= want (or the launch has been aborted); loads issued after it see what the producer wrote before
__device__ __forceinline__ void wait_ge(const int *p, int want, int *abort_word) {
  bool ok = ld_relaxed(p) >= want;
  if (!ok) {
    for (int it = 0; it < WAIT_LIMIT; ++it) {
      __builtin_amdgcn_s_sleep(8);
      if (ld_relaxed(p) >= want) { ok = true; break; }
      if ((it & 15) == 15 && ld_relaxed(abort_word) != 0) break;
    }
    if (!ok) __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("" ::: "memory");
}

// ---- LDS plumbing of a chain workgroup --------------------------------------------------------------------------------------
// ring sizes (rows): emission ring, lattice-row ring; the loader's register ring (rows in flight); 2..3 load instructions per
// emission row and a 6-bit vmcnt bound the latter
template <int NL> struct Rings {
  static constexpr int ER = 16;
  static constexpr int OR = NL >= 4 ? 8 : 16;
  static constexpr int RL = NL >= 4 ? 16 : 24;
};
__device__ __forceinline__ int lds_ld(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_st(int *p, int v) {
  asm volatile("" ::: "memory");  // (LDS operations of a wavefront execute in program order: the data written before is there first)
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// counters of a chain workgroup (LDS): rows staged by the loader / consumed by the chain, rows written by the chain / taken by
// the storer; dead: a wait of this workgroup gave up; e_ready: steps whose emission rows are in memory (the watcher's news)
struct ChainCtl { int e_staged, e_used, o_written, o_taken, dead, e_ready, pad[2]; };

// wave-uniform wait until *p (LDS) > base; bounded like wait_ge.  LDS operations only: a vector-memory operation inside a branch
// of the loader's loop would turn every counted s_waitcnt vmcnt(N) the compiler derives there into vmcnt(0) (first version: the
// abort word polled here -- one emission row per memory round trip, 0.9 us per step).
__device__ __forceinline__ int lds_wait_gt(const int *p, int base, ChainCtl *ctl) {
  int v = lds_ld(p);
  if (!(v > base)) {
    bool ok = false;
    for (int it = 0; it < WAIT_LIMIT; ++it) {
      __builtin_amdgcn_s_sleep(1);
      v = lds_ld(p);
      if (v > base) { ok = true; break; }
      if ((it & 63) == 63 && lds_ld(&ctl->dead) != 0) break;
    }
    if (!ok) {
      lds_st(&ctl->dead, 1);  // (the chain wavefront reports it to the abort word when it ends)
      v = 0x40000000;         // (every later wait of this wavefront passes at once)
    }
  }
  // compiler barrier: the ring accesses that follow (plain LDS loads / stores) must not move above the counter read -- a relaxed
  // atomic load orders nothing by itself, and s_sleep is no memory operation to the optimiser (r03 soak: rare wrong rows at
  // B = 270, four label positions per lane)
  asm volatile("" ::: "memory");
  return v;
}

// 16 bytes written through (sc1), fire and forget: the storer counts its own vector-memory instructions (vmcnt is in issue order)
__device__ __forceinline__ void st16_sc1(float *p, float4 v) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  const v4f w = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(w) : "memory");
}
template <int N> __device__ __forceinline__ void vmcnt_le() {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// WATCHER: follows the emission counters of utterance b in memory (in the order the chain consumes the frames: alpha t = 0, 1, ...;
// beta t = len-1, len-2, ...) and passes the news on through LDS -- the only wavefront of a chain workgroup that polls memory.
template <int DIR>
__device__ __forceinline__ void chain_watcher(int len, int b, SyncView sv, ChainCtl *ctl) {
  int ready = 0;
  while (ready < len) {
    const int t = DIR == 0 ? ready : len - 1 - ready;
    const int c = t / CHUNK;
    const int lo = c * CHUNK, hi = (lo + CHUNK < len) ? lo + CHUNK : len;
    wait_ge(sv.cnt(b) + c, hi - lo, sv.abort_word());
    ready = DIR == 0 ? hi : len - lo;
    lds_st(&ctl->e_ready, ready);
  }
}

// LOADER: emission rows of utterance b in the order the chain consumes them (alpha: t = 0, 1, ...; beta: t = len-1, len-2, ...),
// RL rows in flight in registers, staged into the LDS ring as the chain frees slots.
template <int NL, int DIR>
__device__ __forceinline__ void chain_loader(const Layout &L, const float *__restrict__ ebase, int len, int lane, float *ering,
                                             ChainCtl *ctl, int diag) {
  constexpr int ER = Rings<NL>::ER, RL = Rings<NL>::RL;
  const int UP = L.UP;
  int ready = 0;  // steps whose emission rows are known to be written (from the watcher, through LDS)
  auto need = [&](int n) {
    if (ready < n) ready = lds_wait_gt(&ctl->e_ready, n - 1, ctl);
  };
  auto erow_ptr = [&](int k) -> const float * {
    const int kk = k < len ? k : len - 1;
    return ebase + (long)(DIR == 0 ? kk : len - 1 - kk) * L.ERS;
  };
  int vz;
  asm volatile("v_mov_b32 %0, 0" : "=v"(vz));
  if (CTC_WIDE_DIAG(8)) {  // (timing diagnostic: whatever the ring holds is declared staged)
    int used = 0;
    for (int k = 0; k < len; ++k) {
      if (k - used >= ER) used = lds_wait_gt(&ctl->e_used, k - ER, ctl);
      lds_st(&ctl->e_staged, k + 1);
    }
    return;
  }
  ERow<NL> buf[RL];
  need(RL < len ? RL : len);
#pragma unroll
  for (int d = 0; d < RL; ++d) load_erow<NL, true>(buf[d], erow_ptr(d), lane, UP, vz);
  int used = 0;
  for (int k0 = 0; k0 < len; k0 += RL) {
#pragma unroll
    for (int d = 0; d < RL; ++d) {
      const int k = k0 + d;
      need(k + RL + 1 < len ? k + RL + 1 : len);
      if (k < len) {
        if (k - used >= ER) used = lds_wait_gt(&ctl->e_used, k - ER, ctl);
        float *slot = ering + (k % ER) * L.ERS;
#pragma unroll
        for (int j = 0; j < NL; ++j) slot[lane * NL + j] = buf[d].y[j];
        if (lane == 0) slot[UP] = buf[d].bl;
        lds_st(&ctl->e_staged, k + 1);
      }
      load_erow<NL, true>(buf[d], erow_ptr(k + RL), lane, UP, vz);  // (clamped: re-reads the last row near the end)
    }
  }
}

// STORER: lattice rows from the LDS ring to memory, four at a time; a group is announced once W younger groups have been issued
// behind it (vmcnt counts in issue order), so the wavefront never drains its stores except at the very end.
template <int KIND, int NL, int DIR>
__device__ __forceinline__ void chain_storer(const Layout &L, float *__restrict__ rows, int len, int b, int lane, SyncView sv,
                                             const float *oring, ChainCtl *ctl, int diag) {
  constexpr int OR = Rings<NL>::OR;
  constexpr int SRS4 = ((KIND == 0 ? 128 * NL : 64 * NL) + 8) / 4;  // 16-byte pieces of a row
  constexpr int RI = (SRS4 + 63) / 64;                               // store instructions per row
  constexpr int GI = 4 * RI + 1;                                     // instructions per group: 4 rows + 1 announcement
  constexpr int W = (63 - 4 * RI) / GI;                              // older groups that may still be in flight
  const int nrows = len + 1;
  int written = 0, g = 0;
  for (int r0 = 0; r0 < nrows; r0 += 4, ++g) {
    const int n = nrows - r0 < 4 ? nrows - r0 : 4;
    if (written < r0 + n) written = lds_wait_gt(&ctl->o_written, r0 + n - 1, ctl);
    float4 v[4][RI];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int q = 0; q < RI; ++q) {
        const int idx = lane + 64 * q;
        v[i][q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < n && idx < SRS4) v[i][q] = *reinterpret_cast<const float4 *>(oring + ((r0 + i) % OR) * L.SRS + 4 * idx);
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the rows are in registers: their slots are free
    lds_st(&ctl->o_taken, r0 + n);
    if (CTC_WIDE_DIAG(2)) continue;  // (timing diagnostic: rows are dropped)
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // (compile-time indices: a runtime row index sent v[][] to scratch memory)
      if (i < n) {
        float *dst = rows + (long)(DIR == 0 ? r0 + i : len - (r0 + i)) * L.SRS;
#pragma unroll
        for (int q = 0; q < RI; ++q) {
          const int idx = lane + 64 * q;
          if (idx < SRS4) st16_sc1(dst + 4 * idx, v[i][q]);  // (never an empty instruction: RI = ceil(SRS4 / 64))
        }
      }
    }
    if (n == 4) {  // (only the last group can be short; the wavefront drains its stores right after it)
      if (CTC_WIDE_DIAG(64)) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // (diagnostic: a full release before every announcement)
      if (CTC_WIDE_DIAG(128)) stores_done();                                      // (diagnostic: drain the stores before every announcement)
      if (!CTC_WIDE_DIAG(1)) vmcnt_le<4 * RI + W * GI>();  // everything older than this group's rows and W whole groups has completed
      // rows of groups 0 .. g-W-1 are in memory: 4 (g - W) rows (an announcement every group keeps the instruction count static)
      const int done_rows = g >= W ? 4 * (g - W) : 0;
      if (lane == 0) __hip_atomic_store(sv.rows_done(b, DIR), done_rows, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  stores_done();
  if (lane == 0) __hip_atomic_store(sv.rows_done(b, DIR), 0x40000000, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// CHAIN: the recursion itself; emission rows from the LDS ring, lattice rows into the LDS ring; no vector-memory operation.
template <int KIND, int NL, int DIR>
__device__ __forceinline__ void chain_main(const Problem &p, const Layout &L, double *__restrict__ logp, float *__restrict__ loss,
                                           int len, int ll, int b, int lane, const float *ering, float *oring, ChainCtl *ctl) {
  constexpr int ER = Rings<NL>::ER, OR = Rings<NL>::OR;
  const int UP = L.UP;
  Scan<KIND, NL, DIR> S;
  S.off = 0.0;
  {
    const int32_t *lab = p.labels + (long)b * p.label_stride;
    auto tok = [&](int i) -> int { return (i >= 0 && i < ll) ? ((i < p.label_stride) ? lab[i] : p.blank) : -1 - (i < 0); };
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const int i = lane * NL + j;
      S.norep[j] = (i == 0) || tok(i) != tok(i - 1);
      S.norep_next[j] = tok(i + 1) != tok(i);
      S.c[j] = NEG;
      S.o[j] = NEG;
    }
  }
  if constexpr (DIR == 0) {
    S.cx = 0.f;  // alpha[0]: only (l=0, closed) is reachable (classic_ctc_loss.py:453-462, simplified_ctc_loss.py:426-438)
  } else {
    // beta[len]: one-hot at l = label_length, both states (classic_ctc_loss.py:366-377, simplified_ctc_loss.py:345-356)
    S.cx = (ll == UP) ? 0.f : NEG;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const int i = lane * NL + j;
      if (i == ll) S.c[j] = 0.f;
      if (KIND == 0 && i == ll - 1) S.o[j] = 0.f;
    }
  }
  int taken = 0, staged = 0;
  int vz;
  asm volatile("v_mov_b32 %0, 0" : "=v"(vz));
  S.template store_row<true>(oring, lane, UP);  // the initial row: row 0 of this sweep
  lds_st(&ctl->o_written, 1);
  // Four steps per trip: one look at the loader's counter, four emission rows out of the ring at once, one look at the storer's
  // counter, then straight-line code (step, row into the ring) -- a trip per step spent more time on its branches and on the
  // LDS round trips than on the recursion (0.29 us per step with nothing else running, against 0.15 for the unrolled sweep of
  // the three-kernel pipeline).  Exact renormalisation every 16 steps as there.
  int k = 0;
  for (; k + 4 <= len; k += 4) {
    if (staged < k + 4) staged = lds_wait_gt(&ctl->e_staged, k + 3, ctl);
    ERow<NL> e[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) load_erow<NL>(e[i], ering + ((k + i) % ER) * L.ERS, lane, UP, vz);
    if (k + 4 - taken >= OR) taken = lds_wait_gt(&ctl->o_taken, k + 4 - OR, ctl);  // slots of rows k+1 .. k+4
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      S.step(e[i]);
      if (i == 3 && (k & 12) == 12) S.renorm();
      S.template store_row<true>(oring + ((k + i + 1) % OR) * L.SRS, lane, UP);
    }
    lds_st(&ctl->e_used, k + 4);
    lds_st(&ctl->o_written, k + 5);
  }
  for (; k < len; ++k) {  // fewer than four steps left
    if (staged < k + 1) staged = lds_wait_gt(&ctl->e_staged, k, ctl);
    ERow<NL> e;
    load_erow<NL>(e, ering + (k % ER) * L.ERS, lane, UP, vz);
    if (k + 1 - taken >= OR) taken = lds_wait_gt(&ctl->o_taken, k + 1 - OR, ctl);
    S.step(e);
    S.template store_row<true>(oring + ((k + 1) % OR) * L.SRS, lane, UP);
    lds_st(&ctl->o_written, k + 2);
  }
  if constexpr (DIR == 0) {
    // loss = -alpha[len, label_length] (classic_ctc_loss.py:152-165, simplified_ctc_loss.py:73-83)
    float mine = NEG;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const int i = lane * NL + j;
      if (i == ll - 1) mine = (KIND == 0) ? lse2(S.c[j], S.o[j]) : S.c[j];
    }
    const float v = (ll == 0) ? S.cx : wave_max(mine);
    if (lane == 0) {
      if (v > NEG_THR) {
        const double lp2 = (double)v + S.off;
        logp[b] = lp2;
        loss[b] = (float)(-lp2 * LN2_D);
      } else {
        logp[b] = -INFINITY;
        loss[b] = INFINITY;
      }
    }
  }
}

// j-th quad (four frames) of an utterance of nq quads in the order "from both ends towards the middle"
__device__ __forceinline__ int outside_in(int j, int nq) { return (j & 1) ? nq - 1 - (j >> 1) : (j >> 1); }
// task id -> (j, b): the workgroups that run at the same time take runs of JG consecutive j of ONE utterance (with the ends
// alternating: JG / 2 neighbouring quads = 2 JG rows at either end) instead of the same j of every utterance -- rows one batch
// stride apart (T V 4 bytes: 16 MB at V = 4096) crowd onto few memory channels
constexpr int JG = 16;
__device__ __forceinline__ void task_of(long id, int B, int &j, int &b) {
  const long grp = id / JG;
  j = (int)(grp / B) * JG + (int)(id % JG);
  b = (int)(grp % B);
}

template <int KIND, int NL>
__global__ __launch_bounds__(256) void wide_kernel(Problem p, Layout L, float *__restrict__ emis, float *__restrict__ alpha,
                                                    float *__restrict__ beta, double *__restrict__ logp, float *__restrict__ loss,
                                                    const float *__restrict__ d_loss, float *__restrict__ grad, int *sync_words,
                                                    int n_chain_wg, int grad_split, int diag) {
  // LDS: stream role = 4 x (1024 bins + 64 NL posteriors); chain role = control words + emission ring + lattice-row ring
  constexpr int STREAM_WORDS = 4 * (1024 + 64 * NL);
  constexpr int SRS_C = (KIND == 0 ? 128 * NL : 64 * NL) + 8, ERS_C = 64 * NL + 4;
  constexpr int CHAIN_WORDS = 8 + Rings<NL>::ER * ERS_C + Rings<NL>::OR * SRS_C;
  __shared__ __attribute__((aligned(16))) unsigned lds_words[STREAM_WORDS > CHAIN_WORDS ? STREAM_WORDS : CHAIN_WORDS];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wg = blockIdx.x;
  SyncView sv{sync_words, (p.T + CHUNK - 1) / CHUNK + 1};

  if (wg < n_chain_wg) {
    // ---- chain role: workgroup = (utterance, direction); wavefront 0 chain, 1 storer, 2 loader, 3 watcher ----
    const int b = wg >> 1, dir = wg & 1;
    ChainCtl *ctl = reinterpret_cast<ChainCtl *>(lds_words);
    float *ering = reinterpret_cast<float *>(lds_words) + 8;
    float *oring = ering + Rings<NL>::ER * ERS_C;
    if (threadIdx.x < 8) lds_words[threadIdx.x] = 0u;
    __syncthreads();
    const int len = v1_clampi(p.logit_length[b], 0, p.T);
    const int ll = p.label_length[b] < 0 ? 0 : p.label_length[b];
    const bool skip = ll > p.U;  // contract violation: reported as an infeasible sample, no row is written (the gradient stage writes zeros)
    float *rows = (dir == 0 ? alpha : beta) + (long)b * (p.T + 1) * L.SRS;
    const float *ebase = emis + (long)b * p.T * L.ERS;
    if (skip) {
      if (w == 0 && lane == 0) {
        if (dir == 0) { logp[b] = -INFINITY; loss[b] = INFINITY; }
        __hip_atomic_store(sv.rows_done(b, dir), 0x40000000, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else if (w == 0) {
      __builtin_amdgcn_s_setprio(3);
      if (dir == 0) chain_main<KIND, NL, 0>(p, L, logp, loss, len, ll, b, lane, ering, oring, ctl);
      else chain_main<KIND, NL, 1>(p, L, logp, loss, len, ll, b, lane, ering, oring, ctl);
      if (lds_ld(&ctl->dead) != 0) __hip_atomic_store(sv.abort_word(), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (dir == 0 && lane == 0 && ld_relaxed(sv.abort_word()) != 0) loss[b] = NAN;  // (a wait gave up: nothing of this launch is valid)
    } else if (w == 1) {
      __builtin_amdgcn_s_setprio(2);
      if (dir == 0) chain_storer<KIND, NL, 0>(L, rows, len, b, lane, sv, oring, ctl, diag);
      else chain_storer<KIND, NL, 1>(L, rows, len, b, lane, sv, oring, ctl, diag);
    } else if (w == 2) {
      __builtin_amdgcn_s_setprio(2);
      if (len > 0) {
        if (dir == 0) chain_loader<NL, 0>(L, ebase, len, lane, ering, ctl, diag);
        else chain_loader<NL, 1>(L, ebase, len, lane, ering, ctl, diag);
      }
    } else {
      if (CTC_WIDE_DIAG(4)) lds_st(&ctl->e_ready, len);  // (timing diagnostic: chains alone, nobody computes emissions)
      else if (dir == 0) chain_watcher<0>(len, b, sv, ctl);
      else chain_watcher<1>(len, b, sv, ctl);
    }
    // the sweep is over: the four wavefronts join the gradient pass (the rings become bins)
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();
  }

  // ---- stream role ----
  if (CTC_WIDE_DIAG(4)) return;
  // Every wavefront is a worker of its own (no barrier): worker i of NW takes the rows i, i + NW, ... of the emission pass -- quads
  // of four consecutive frames, utterances walked from both ends towards the middle of THEIR frames: the order in which the chains
  // consume them -- and then of the gradient pass: the valid quads in the reverse order, middle outwards, the order in which alpha
  // and beta rows of the same frame appear once the chains have crossed; padded quads last.  A wavefront finishes its emission
  // rows before its first gradient row, and an emission row waits for nothing.  (Tickets drawn from one atomic counter would
  // balance the load better, but one address serves ~65 M atomics a second: 3.8 ms at B = 128.)
  const int NQ = (p.T + 3) / 4;
  const int NR = ((NQ + JG - 1) / JG) * JG * p.B * 4;  // rows per pass (j padded to whole runs)
  // emission pass: the stream workgroups' wavefronts; gradient pass: every wavefront of the grid (a chain workgroup joins when its
  // sweep is over -- it cannot take emission rows: its own chain would wait for them)
  const int NWE = (gridDim.x - n_chain_wg) * 4, NWG = gridDim.x * 4;
  const bool is_chain = wg < n_chain_wg;
  unsigned *bins = lds_words + w * 1024;
  unsigned *qtab = lds_words + 4 * 1024 + w * 64 * NL;
  // A row's emission counter is bumped one row later, after the next row has been read: vmcnt counts in issue order, so loads that
  // have returned prove the older stores complete -- a wavefront drains its (written-through, slow) stores once.
  int *pend = nullptr;
  auto flush = [&]() {
    if (pend != nullptr) {
      stores_done();
      if (lane == 0) __hip_atomic_fetch_add(pend, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      pend = nullptr;
    }
  };
  // ids 0 .. NR-1: emission rows, NR .. 2 NR-1: gradient rows.  Three segments: [0, NR) strided over the stream workers;
  // [NR, NR + grad_split) likewise (the chains are still running: the host's estimate of the head start the stream workers
  // have); [NR + grad_split, 2 NR) strided over every wavefront of the grid.
  const long seg2 = (long)NR + grad_split;
  const int ws = (wg - n_chain_wg) * 4 + w, wa = wg * 4 + w;
  long it = is_chain ? seg2 + wa : (long)ws;
  for (; it < 2L * NR;) {
    const int id = (int)it;
    const int rid = id < NR ? id : id - NR;
    if (it < NR) { it += NWE; if (it >= NR) it = (long)NR + ws; if (it >= seg2) it = seg2 + wa; }
    else if (it < seg2) { it += NWE; if (it >= seg2) it = seg2 + wa; }
    else it += NWG;
    // the four wavefronts of a workgroup read the four rows of a quad together (64 KB in one piece); left alone they drift apart
    // (measured: +6 % on the call).  Every wavefront of a stream workgroup walks the same number of emission ids.
    if (id < NR) __syncthreads();
    int j, b;
    task_of(rid >> 2, p.B, j, b);
    if (j >= NQ) continue;
    const int len = v1_clampi(p.logit_length[b], 0, p.T);
    const int nq = (len + 3) / 4;
    if (id < NR) {
      if (j >= nq) continue;
      const int t = 4 * outside_in(j, nq) + (rid & 3);
      if (t >= len) continue;
      emit_row<true, NL, 16>(p, L, emis, b, t, lane);  // (16 KB per wavefront in flight: the pass is latency-bound per wavefront)
      asm volatile("" ::: "memory");
      if (lane == 0 && pend != nullptr) __hip_atomic_fetch_add(pend, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      pend = sv.cnt(b) + t / CHUNK;
    } else {
      if (CTC_WIDE_DIAG(16)) continue;  // (timing diagnostic: no gradient pass)
      flush();
      const int q = j < nq ? outside_in(nq - 1 - j, nq) : j;
      const int t = 4 * q + (rid & 3);
      if (t >= p.T) continue;
      grad_row<KIND, NL, true>(p, L, emis, alpha, beta, d_loss, grad, b, t, lane, bins, qtab, [&](int len) {
        // alpha row t+1 (classic) / t (simplified) and beta row t+1: alpha rows 0..t+1, beta rows t+1..len
        const int lag = CTC_WIDE_DIAG(32) ? 16 : 0;  // (diagnostic: ask for 16 rows more than needed)
        wait_ge(sv.rows_done(b, 0), t + 2 + lag, sv.abort_word());
        wait_ge(sv.rows_done(b, 1), len - t + lag, sv.abort_word());
      });
    }
  }
  flush();
}

template <int KIND, int NL>
static hipError_t launch(const Problem &p, const Layout &L, char *ws, float *loss, const float *d_loss, float *grad, hipStream_t st) {
  static int per_cu = -1, n_cu = 0;  // (written once with the same values by whoever gets here first)
  if (per_cu < 0) {
    int dev = 0, cu = 0, occ = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev);
    if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, wide_kernel<KIND, NL>, 256, 0);
    if (e != hipSuccess) return e;
    if (occ < 1 || cu < 1) return hipErrorLaunchOutOfResources;
    n_cu = cu;
    per_cu = occ > 8 ? 8 : occ;
  }
  const int capacity = per_cu * n_cu;
  // utterances per launch: their chain workgroups (two each) may take at most half of the resident grid
  // ... and at most ONE chain workgroup per CU: with two on a CU (more than n_cu / 2 utterances in a launch) some of the lattice
  // rows they hand over arrive wrong (r03 soak; not understood -- tests/tools/wide_race.py: B = 128 clean, B = 160 not)
  int bmax = capacity >= 4 ? capacity / 4 : 1;
  if (bmax > n_cu / 2) bmax = n_cu / 2 > 0 ? n_cu / 2 : 1;
  for (int b0 = 0; b0 < p.B; b0 += bmax) {
    Problem q = p;
    q.B = p.B - b0 < bmax ? p.B - b0 : bmax;
    q.logits = p.logits + (long)b0 * p.xsb;
    q.labels = p.labels + (long)b0 * p.label_stride;
    q.label_length = p.label_length + b0;
    q.logit_length = p.logit_length + b0;
    float *emis = reinterpret_cast<float *>(ws + L.off_emis) + (long)b0 * p.T * L.ERS;
    float *alpha = reinterpret_cast<float *>(ws + L.off_alpha) + (long)b0 * (p.T + 1) * L.SRS;
    float *beta = reinterpret_cast<float *>(ws + L.off_beta) + (long)b0 * (p.T + 1) * L.SRS;
    double *logp = reinterpret_cast<double *>(ws + L.off_logp) + b0;
    int *sync_words = reinterpret_cast<int *>(ws + L.off_kexp);  // (the exponent region of the fused tiers: unused by full-row layouts)
    hipError_t e = hipMemsetAsync(sync_words, 0, sync_bytes(q.B, q.T), st);
    if (e != hipSuccess) return e;
    const int n_chain = 2 * q.B;
    const long ntask = (long)((q.T + 3) / 4) * q.B;  // (four rows each: a workgroup's four wavefronts)
    long n_stream = capacity - n_chain;
    if (n_stream > ntask) n_stream = ntask;
    if (n_stream < 1) n_stream = 1;
    // Gradient rows the stream workers take alone, before the chain workgroups can join (a model, not a measurement: a sweep
    // step costs ~0.27 us beside the streams, the emission pass reads at ~3.2 TB/s, the gradient pass moves 8 V bytes per row
    // at ~4.5 TB/s; what it decides is only who processes which row)
    const int nq_pad = (((q.T + 3) / 4 + JG - 1) / JG) * JG;
    const long NR = (long)nq_pad * q.B * 4;
    const double t_chain = 0.27e-6 * q.T, t_emit = (double)q.B * q.T * q.V * 4 / 3.2e12;
    const double t_row_all = (double)(n_chain + n_stream) * 4 * 8.0 * q.V / 4.5e12;  // one row per wavefront, all wavefronts busy
    long split = t_chain > t_emit ? (long)((t_chain - t_emit) / t_row_all * (double)(n_stream * 4)) : 0;
    if (split > NR) split = NR;
#ifdef CTC_DIAG
    if (g_wide_diag & 256) split = NR;  // (diagnostic: the chain workgroups take no gradient rows)
#endif
    hipLaunchKernelGGL((wide_kernel<KIND, NL>), dim3((unsigned)(n_chain + n_stream)), dim3(256), 0, st, q, L, emis, alpha, beta, logp,
                       loss + b0, d_loss ? d_loss + b0 : nullptr, grad + (long)b0 * p.gsb, sync_words, n_chain, (int)split,
#ifdef CTC_DIAG
                       g_wide_diag
#else
                       0
#endif
    );
    e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

template <int KIND>
static hipError_t launch_kind(const Problem &p, const Layout &L, char *ws, float *loss, const float *d_loss, float *grad, hipStream_t st) {
  switch (L.NL) {
    case 1: return launch<KIND, 1>(p, L, ws, loss, d_loss, grad, st);
    case 2: return launch<KIND, 2>(p, L, ws, loss, d_loss, grad, st);
    case 4: return launch<KIND, 4>(p, L, ws, loss, d_loss, grad, st);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace wide

// float32 rows, 16-byte aligned, V a multiple of 4 (the accesses of grad_row), labels of up to 256 positions (the LDS rings of
// a chain workgroup hold 8..16 rows), a gradient wanted
bool wide_eligible(const Problem &p, const Layout &L) {
  return p.V > 1024 && L.NL <= 4 && p.xdtype == 0 && p.gdtype == 0 && p.row0 == nullptr && (p.align_bits & 15) == 0 && ((p.V | p.xsb | p.xst | p.gsb | p.gst) & 3) == 0 &&
         L.UP <= CTC_AMD_MAX_U && p.B > 0 && p.T > 0 &&
         wide::sync_bytes(p.B, p.T) <= (size_t)p.B * 2 * L.nslot * 64 * 4;
}

hipError_t run_wide(const Problem &p, const Layout &L, char *ws, float *loss, const float *d_loss, float *grad, hipStream_t st) {
  return p.kind == 0 ? wide::launch_kind<0>(p, L, ws, loss, d_loss, grad, st) : wide::launch_kind<1>(p, L, ws, loss, d_loss, grad, st);
}

}  // namespace ctc

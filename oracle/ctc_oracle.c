/*
 * ctc_oracle.c -- plain-C restatement of the reference's loss + gradient path.  TEST INFRASTRUCTURE ONLY.
 *
 * Same role and rules as oracle/ctc_oracle.py (read its header): only tests/, __graft_entry__.smoke() and
 * the cpu_baseline leg of bench.py may load this; the product never does.  It exists so that (a) parity at
 * the full BASELINE sizes can be checked in seconds and (b) bench.py has a multi-threaded CPU baseline
 * ("kind": "port") with the same algorithmic structure as the reference: log-softmax, batch-parallel,
 * strictly sequential over T, log-space alpha/beta with the 2-argument log-sum-exp of tools.py:57-71,
 * posterior scatter by label, gradient = softmax - posterior.  It is pinned against ctc_oracle.py (which
 * is pinned against the reference's known answers) in tests/test_oracle_c.py.
 *
 * Reference lines followed (paths relative to alexeytochin/tf_seq2seq_losses v0.3.0):
 *   log-softmax                      tools.py:27-40
 *   lse2                             tools.py:57-71
 *   label cleaning / emissions       base_loss.py:328-344, 378-418
 *   classic alpha / beta / loss      classic_ctc_loss.py:415-462, 349-377, 152-165
 *   simplified alpha / beta / loss   simplified_ctc_loss.py:393-438, 327-356, 73-83
 *   combine + scatter                classic_ctc_loss.py:565-669, simplified_ctc_loss.py:456-534, base_loss.py:420-468
 *   gradient, masks                  base_loss.py:262-298 ; chain rule through tools.py:37-39
 *
 * real_t is double (arbiter) or float (the reference's own precision) -- built twice by the Makefile.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef REAL
#define REAL double
#endif
typedef REAL real_t;

#define NEG_INF (-(real_t)INFINITY)

static inline real_t lse2(real_t x, real_t y) { /* tools.py:57-71 */
  if (x < y) return y + (real_t)log1p(exp((double)(x - y)));
  if (x > y) return x + (real_t)log1p(exp((double)(y - x)));
  return x + (real_t)0.69314718055994530942; /* x == y, including (-inf, -inf) -> -inf */
}

/* One sample.  lp: [T][V] log-probabilities with padded frames already overwritten (base_loss.py:378-393).
 * work: 2*(T+1)*L*S reals.  Returns loss; post[T][V] receives the posterior (= -gradient w.r.t. log-probs). */
static real_t sample_classic(const real_t *lp, int T, int V, const int32_t *label, int ll, int U, int blank,
                             real_t *work, real_t *post) {
  const int L = U + 1;
  real_t *alpha = work, *beta = work + (size_t)(T + 1) * L * 2;
#define A(t, l, s) alpha[((size_t)(t) * L + (l)) * 2 + (s)]
#define B(t, l, s) beta[((size_t)(t) * L + (l)) * 2 + (s)]
  /* cleaned label, previous label (cyclic roll), base_loss.py:395-418, 519-525 */
  int *lab = (int *)malloc(sizeof(int) * 2 * L), *prev = lab + L;
  for (int l = 0; l < L; ++l) lab[l] = (l < ll) ? label[l] : blank;
  for (int l = 0; l < L; ++l) prev[l] = lab[(l + L - 1) % L];
  for (int l = 0; l < L; ++l) { A(0, l, 0) = NEG_INF; A(0, l, 1) = NEG_INF; }
  A(0, 0, 0) = 0;
  for (int t = 0; t < T; ++t) {
    const real_t *p = lp + (size_t)t * V;
    const real_t bl = p[blank];
    for (int l = 0; l < L; ++l) {
      /* horizontal: closed <- (closed|open) via blank; open <- open via repeat of prev token */
      real_t rep = (prev[l] == blank) ? NEG_INF : p[prev[l]];
      real_t hc = bl + lse2(A(t, l, 0), A(t, l, 1));
      real_t ho = rep + A(t, l, 1);
      /* diagonal into open(l) from l-1 (cyclic; the wrapped term is -inf because y[U] is masked) */
      int lm = (l + L - 1) % L;
      real_t y = (lm < ll) ? p[lab[lm]] : NEG_INF;
      real_t yo = (lab[lm] != prev[lm]) ? y : NEG_INF;
      real_t d = lse2(y + A(t, lm, 0), yo + A(t, lm, 1));
      A(t + 1, l, 0) = hc;
      A(t + 1, l, 1) = lse2(ho, d);
    }
  }
  for (int l = 0; l < L; ++l) { B(T, l, 0) = B(T, l, 1) = (l == ll) ? 0 : NEG_INF; }
  for (int t = T - 1; t >= 0; --t) {
    const real_t *p = lp + (size_t)t * V;
    const real_t bl = p[blank];
    for (int l = 0; l < L; ++l) {
      int ln = (l + 1) % L;
      real_t rep = (prev[l] == blank) ? NEG_INF : p[prev[l]];
      real_t y = (l < ll) ? p[lab[l]] : NEG_INF;
      real_t yo = (lab[l] != prev[l]) ? y : NEG_INF;
      real_t h = bl + B(t + 1, l, 0);
      B(t, l, 0) = lse2(h, y + B(t + 1, ln, 1));
      B(t, l, 1) = lse2(lse2(h, rep + B(t + 1, l, 1)), yo + B(t + 1, ln, 1));
    }
  }
  const real_t logp = lse2(A(T, ll, 0), A(T, ll, 1));
  const real_t loss = -logp;
  if (post) {
    memset(post, 0, sizeof(real_t) * (size_t)T * V);
    if (isfinite((double)loss)) {
      for (int t = 0; t < T; ++t) {
        const real_t *p = lp + (size_t)t * V;
        real_t *q = post + (size_t)t * V;
        real_t blank_lse = NEG_INF;
        for (int l = 0; l < L; ++l) {
          int ln = (l + 1) % L;
          blank_lse = lse2(blank_lse, lse2(A(t, l, 0), A(t, l, 1)) + B(t + 1, l, 0));
          /* repeat term scattered by prev[l], advance term scattered by label[l] */
          if (prev[l] != blank) q[prev[l]] += (real_t)exp((double)(A(t, l, 1) + p[prev[l]] + B(t + 1, l, 1) + loss));
          if (l < ll && lab[l] != blank) {
            real_t y = p[lab[l]];
            real_t yo = (lab[l] != prev[l]) ? y : NEG_INF;
            q[lab[l]] += (real_t)exp((double)(lse2(A(t, l, 0) + y, A(t, l, 1) + yo) + B(t + 1, ln, 1) + loss));
          }
        }
        q[blank] = (real_t)exp((double)(p[blank] + blank_lse + loss));
      }
    }
  }
  free(lab);
#undef A
#undef B
  return loss;
}

static real_t sample_simplified(const real_t *lp, int T, int V, const int32_t *label, int ll, int U, int blank,
                                real_t *work, real_t *post) {
  const int L = U + 1;
  real_t *alpha = work, *beta = work + (size_t)(T + 1) * L;
#define A(t, l) alpha[(size_t)(t) * L + (l)]
#define B(t, l) beta[(size_t)(t) * L + (l)]
  int *lab = (int *)malloc(sizeof(int) * L);
  for (int l = 0; l < L; ++l) lab[l] = (l < ll) ? label[l] : blank;
  for (int l = 0; l < L; ++l) A(0, l) = NEG_INF;
  A(0, 0) = 0;
  for (int t = 0; t < T; ++t) {
    const real_t *p = lp + (size_t)t * V;
    for (int l = 0; l < L; ++l) {
      int lm = (l + L - 1) % L;
      real_t y = (lm < ll) ? p[lab[lm]] : NEG_INF;
      A(t + 1, l) = lse2(p[blank] + A(t, l), y + A(t, lm));
    }
  }
  for (int l = 0; l < L; ++l) B(T, l) = (l == ll) ? 0 : NEG_INF;
  for (int t = T - 1; t >= 0; --t) {
    const real_t *p = lp + (size_t)t * V;
    for (int l = 0; l < L; ++l) {
      int ln = (l + 1) % L;
      real_t y = (l < ll) ? p[lab[l]] : NEG_INF;
      B(t, l) = lse2(p[blank] + B(t + 1, l), y + B(t + 1, ln));
    }
  }
  const real_t loss = -A(T, ll);
  if (post) {
    memset(post, 0, sizeof(real_t) * (size_t)T * V);
    if (isfinite((double)loss)) {
      for (int t = 0; t < T; ++t) {
        const real_t *p = lp + (size_t)t * V;
        real_t *q = post + (size_t)t * V;
        real_t blank_lse = NEG_INF;
        for (int l = 0; l < L; ++l) {
          blank_lse = lse2(blank_lse, A(t, l) + B(t + 1, l));
          if (l < ll && lab[l] != blank) q[lab[l]] += (real_t)exp((double)(A(t, l) + p[lab[l]] + B(t + 1, (l + 1) % L) + loss));
        }
        q[blank] = (real_t)exp((double)(p[blank] + blank_lse + loss));
      }
    }
  }
  free(lab);
#undef A
#undef B
  return loss;
}

/*
 * Batch entry point.  kind: 0 classic, 1 simplified.  logits [B][T][V] (float32), labels [B][label_stride],
 * loss [B] (double), grad [B][T][V] (double, may be NULL): gradient w.r.t. logits of sum_b loss[b]
 * (softmax * sum(post) - post on valid frames of feasible samples, else 0).
 * U = max(label_length) is computed here like base_loss.py:482-486.  Returns 0.
 */
int ctc_oracle_loss_grad(int kind, const float *logits, const int32_t *labels, int label_stride,
                         const int32_t *label_length, const int32_t *logit_length, int blank, int B, int T, int V,
                         double *loss, double *grad, int n_threads) {
  int U = 0;
  for (int b = 0; b < B; ++b) if (label_length[b] > U) U = label_length[b];
  const int L = U + 1, S = (kind == 0) ? 2 : 1;
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
#pragma omp parallel
  {
    real_t *lp = (real_t *)malloc(sizeof(real_t) * (size_t)(T > 0 ? T : 1) * V);
    real_t *work = (real_t *)malloc(sizeof(real_t) * 2 * (size_t)(T + 1) * L * S);
    real_t *post = grad ? (real_t *)malloc(sizeof(real_t) * (size_t)(T > 0 ? T : 1) * V) : NULL;
    int32_t *lab = (int32_t *)malloc(sizeof(int32_t) * (size_t)L);
#pragma omp for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
      const float *x = logits + (size_t)b * T * V;
      int len = logit_length[b] < 0 ? 0 : (logit_length[b] > T ? T : logit_length[b]);
      int ll = label_length[b] < 0 ? 0 : label_length[b];
      for (int l = 0; l < L; ++l) lab[l] = (l < label_stride) ? labels[(size_t)b * label_stride + l] : blank;
      for (int t = 0; t < T; ++t) {
        real_t *p = lp + (size_t)t * V;
        if (t < len) { /* tools.py:27-40 */
          real_t mx = NEG_INF;
          for (int k = 0; k < V; ++k) if ((real_t)x[(size_t)t * V + k] > mx) mx = (real_t)x[(size_t)t * V + k];
          if (!isfinite((double)mx)) mx = 0;
          real_t s = 0;
          for (int k = 0; k < V; ++k) s += (real_t)exp((double)((real_t)x[(size_t)t * V + k] - mx));
          real_t lse = mx + (real_t)log((double)s);
          for (int k = 0; k < V; ++k) p[k] = (real_t)x[(size_t)t * V + k] - lse;
        } else { /* base_loss.py:378-393 */
          for (int k = 0; k < V; ++k) p[k] = NEG_INF;
          p[blank] = 0;
        }
      }
      real_t l = (kind == 0) ? sample_classic(lp, T, V, lab, ll, U, blank, work, post)
                             : sample_simplified(lp, T, V, lab, ll, U, blank, work, post);
      loss[b] = (double)l;
      if (grad) {
        double *g = grad + (size_t)b * T * V;
        for (int t = 0; t < T; ++t) {
          const real_t *p = lp + (size_t)t * V;
          const real_t *q = post + (size_t)t * V;
          if (t < len && isfinite((double)l)) {
            real_t sum = 0;
            for (int k = 0; k < V; ++k) sum += q[k];
            for (int k = 0; k < V; ++k) g[(size_t)t * V + k] = (double)((real_t)exp((double)p[k]) * sum - q[k]);
          } else {
            for (int k = 0; k < V; ++k) g[(size_t)t * V + k] = 0.0;
          }
        }
      }
    }
    free(lp); free(work); free(post); free(lab);
  }
  return 0;
}

int ctc_oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

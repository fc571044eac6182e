/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * call into this file.  The product (wenet-celoss_amd/) never links or loads it.
 *
 * CPU restatement of the RNN-T (transducer) negative log-likelihood and its
 * gradient with respect to the joiner logits, with the log-softmax fused in.
 *
 * What it restates.  The reference calls a third-party routine for this:
 *   torchaudio.functional.rnnt_loss(logits, targets, logit_lengths,
 *                                   target_lengths, blank=0, reduction=...)
 * at /root/reference/wenet/transducer/transducer.py:142-147 (training,
 * reduction="mean") and :296-301 (rescoring, reduction='none').
 * torchaudio is NOT vendored under /root/reference and is not installed in the
 * build image; the reference pins it only in docs/CI (torchaudio==0.10.0,
 * /root/reference/README.md:64, .github/workflows/unit_test.yml:16-17).
 * This file therefore restates the PUBLISHED algorithm (Graves 2012,
 * "Sequence Transduction with Recurrent Neural Networks", forward-backward
 * over the T x (U+1) lattice, gradient taken through the fused log-softmax --
 * the formulation used by warp-transducer and torchaudio).
 *
 * PARITY STATUS: "parity unpinned" with respect to the reference itself --
 * the reference holds no test, golden vector or fixture for this call and the
 * library that implements it cannot be run here.  The restatement is pinned
 * instead by (tests/test_oracle_rnnt.py):
 *   - the public warp-transducer / torchaudio unit-test vector
 *     (B=1,T=2,U=2,V=5: cost 4.495666, full gradient),
 *   - brute-force enumeration of every alignment path on small lattices,
 *   - a float64 PyTorch autograd of the alpha recursion (independent code).
 *
 * Exported: wr_oracle_rnnt_f64 -- float logits in, all arithmetic in double.
 * This is the checker the HIP path is compared against.  (The threaded float32
 * port timed as the CPU baseline lives in rnnt_baseline.c.)
 *
 * Layout (same as the reference call): logits [B, Tmax, U1max, V] contiguous,
 * targets [B, U1max-1] int32, logit_lengths [B], target_lengths [B].
 * Gradient is written for the whole padded tensor and is exactly zero outside
 * [0,T_b) x [0,U_b].
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline double lae_d(double a, double b) {
    if (a == -INFINITY) return b;
    if (b == -INFINITY) return a;
    double m = a > b ? a : b;
    return m + log1p(exp(-fabs(a - b)));
}

/* ------------------------------------------------------------------ f64 -- */
/* One utterance.  Returns cost; writes grad (may be NULL). */
static double rnnt_one_f64(const float *logits, const int32_t *y, int T, int U,
                           int Tmax, int U1max, int V, int blank, double clamp,
                           float *grad)
{
    const int U1 = U + 1;
    (void)Tmax;
    double *denom = (double *)malloc(sizeof(double) * (size_t)T * U1);
    double *alpha = (double *)malloc(sizeof(double) * (size_t)T * U1);
    double *beta  = (double *)malloc(sizeof(double) * (size_t)T * U1);
#define LG(t, u) (logits + ((size_t)(t) * U1max + (u)) * (size_t)V)
#define IX(t, u) ((size_t)(t) * U1 + (u))
    /* App. A.1: denom(t,u) = logsumexp_v logits[t,u,v] (max-subtracted).
     * Lattice cells are independent here: the pragma only spreads them over the
     * host cores (each cell's sum keeps its sequential order), so that one
     * BASELINE-shape utterance (151 000 cells x 5 000 double exp) checks in seconds. */
#pragma omp parallel for collapse(2) schedule(static)
    for (int t = 0; t < T; ++t)
        for (int u = 0; u < U1; ++u) {
            const float *row = LG(t, u);
            double mx = row[0];
            for (int v = 1; v < V; ++v) if (row[v] > mx) mx = row[v];
            double s = 0.0;
            for (int v = 0; v < V; ++v) s += exp((double)row[v] - mx);
            denom[IX(t, u)] = mx + log(s);
        }
#define SKIP(t, u) ((double)LG(t, u)[blank] - denom[IX(t, u)])
#define EMIT(t, u) ((double)LG(t, u)[y[u]] - denom[IX(t, u)])
    /* alpha recursion */
    for (int t = 0; t < T; ++t)
        for (int u = 0; u < U1; ++u) {
            if (t == 0 && u == 0) { alpha[0] = 0.0; continue; }
            double a = -INFINITY, b = -INFINITY;
            if (t > 0) a = alpha[IX(t - 1, u)] + SKIP(t - 1, u);
            if (u > 0) b = alpha[IX(t, u - 1)] + EMIT(t, u - 1);
            alpha[IX(t, u)] = lae_d(a, b);
        }
    /* beta recursion */
    for (int t = T - 1; t >= 0; --t)
        for (int u = U; u >= 0; --u) {
            if (t == T - 1 && u == U) { beta[IX(t, u)] = SKIP(t, u); continue; }
            double a = -INFINITY, b = -INFINITY;
            if (t < T - 1) a = beta[IX(t + 1, u)] + SKIP(t, u);
            if (u < U) b = beta[IX(t, u + 1)] + EMIT(t, u);
            beta[IX(t, u)] = lae_d(a, b);
        }
    const double cost = -beta[0];
    if (grad) {
        /* Gradient through the fused log-softmax; the case chain and its order
         * follow SURVEY.md App. A.1 (first matching case wins).  Cells are
         * independent (element-wise), hence the pragma. */
#pragma omp parallel for collapse(2) schedule(static)
        for (int t = 0; t < T; ++t)
            for (int u = 0; u < U1; ++u) {
                const float *row = LG(t, u);
                float *grow = grad + ((size_t)t * U1max + u) * (size_t)V;
                const double c = alpha[IX(t, u)] + cost - denom[IX(t, u)];
                const double bt = beta[IX(t, u)];
                for (int v = 0; v < V; ++v) {
                    const double g = (double)row[v] + c;
                    double r;
                    if (v == blank && t == T - 1 && u == U)
                        r = exp(g + bt) - exp(g);
                    else if (v == blank && t < T - 1)
                        r = exp(g + bt) - exp(g + beta[IX(t + 1, u)]);
                    else if (u < U && v == y[u])
                        r = exp(g + bt) - exp(g + beta[IX(t, u + 1)]);
                    else
                        r = exp(g + bt);
                    if (clamp > 0.0) {
                        if (r > clamp) r = clamp;
                        if (r < -clamp) r = -clamp;
                    }
                    grow[v] = (float)r;
                }
            }
    }
#undef SKIP
#undef EMIT
#undef LG
#undef IX
    free(denom); free(alpha); free(beta);
    return cost;
}

/* costs [B] (double); grad [B,Tmax,U1max,V] float or NULL.  Returns 0. */
int wr_oracle_rnnt_f64(const float *logits, const int32_t *targets,
                       const int32_t *logit_lengths, const int32_t *target_lengths,
                       int B, int Tmax, int U1max, int V, int blank, double clamp,
                       double *costs, float *grad)
{
    const size_t per_b = (size_t)Tmax * U1max * (size_t)V;
    if (grad) memset(grad, 0, sizeof(float) * per_b * (size_t)B);
    for (int b = 0; b < B; ++b) {
        costs[b] = rnnt_one_f64(logits + per_b * b, targets + (size_t)b * (U1max - 1),
                                logit_lengths[b], target_lengths[b], Tmax, U1max, V,
                                blank, clamp, grad ? grad + per_b * b : NULL);
    }
    return 0;
}

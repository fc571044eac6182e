/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * call into this file.
 *
 * CPU restatement of the CTC negative log-likelihood with the log-softmax
 * fused in, as the reference computes it in
 *   /root/reference/wenet/transformer/ctc.py:46-64
 *     ys_hat = ctc_lo(hs_pad).transpose(0,1).log_softmax(2)          (:57-60)
 *     loss   = torch.nn.CTCLoss(reduction='sum')(ys_hat, ys_pad, hlens, ys_lens)  (:44,:61)
 *     loss   = loss / B                                               (:63)
 * torch.nn.CTCLoss (ATen ctc_loss, blank=0, zero_infinity=False) is third-party
 * but IS present in the build image (torch 2.10.0 CPU), so this restatement is
 * PINNED: tests/test_oracle_ctc.py checks it against torch.nn.CTCLoss run live
 * and against fixtures produced by importing the reference's own CTC module
 * (tests/golden/make_golden.py -> tests/golden/ctc_ref_*.npz).
 *
 * Algorithm (Graves 2006): extended label sequence l' = (blank, y1, blank, y2,
 * ..., blank) of length S' = 2S+1;
 *   alpha_0(0)=lp(0,blank), alpha_0(1)=lp(0,y1);
 *   alpha_t(s) = lp(t,l'_s) + logsumexp(alpha_{t-1}(s), alpha_{t-1}(s-1),
 *                                       alpha_{t-1}(s-2) if l'_s != blank and l'_s != l'_{s-2})
 *   nll = -logsumexp(alpha_{T-1}(S'-1), alpha_{T-1}(S'-2))
 * and the mirror recursion for beta.  Gradient w.r.t. the pre-softmax
 * activations:  softmax(t,v) - exp(logsumexp_{s: l'_s = v}(alpha_t(s)+beta_t(s)) + nll - lp(t,v)),
 * zero for t >= T_b.
 *
 * Input layout: logits [B, Tmax, V] contiguous float (batch-major; the
 * reference's transpose to (T,B,V) is a layout choice of nn.CTCLoss, not part
 * of the mathematics), targets [B, Smax] int64-or-int32 padded with anything
 * (never read past target_lengths), arithmetic in double.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline double lae(double a, double b) {
    if (a == -INFINITY) return b;
    if (b == -INFINITY) return a;
    double m = a > b ? a : b;
    return m + log1p(exp(-fabs(a - b)));
}

static double ctc_one(const float *logits, const int32_t *y, int T, int S,
                      int Tmax, int V, int blank, float *grad)
{
    const int SP = 2 * S + 1;
    double *lp    = (double *)malloc(sizeof(double) * (size_t)T * V);
    double *alpha = (double *)malloc(sizeof(double) * (size_t)T * SP);
    double *beta  = (double *)malloc(sizeof(double) * (size_t)T * SP);
    (void)Tmax;
    /* frames are independent in the log-softmax: the pragma only spreads them over the host cores */
#pragma omp parallel for schedule(static)
    for (int t = 0; t < T; ++t) {
        const float *row = logits + (size_t)t * V;
        double mx = row[0];
        for (int v = 1; v < V; ++v) if (row[v] > mx) mx = row[v];
        double s = 0.0;
        for (int v = 0; v < V; ++v) s += exp((double)row[v] - mx);
        const double lse = mx + log(s);
        for (int v = 0; v < V; ++v) lp[(size_t)t * V + v] = (double)row[v] - lse;
    }
#define LAB(s) (((s) & 1) ? y[(s) >> 1] : blank)
#define LP(t, s) lp[(size_t)(t) * V + LAB(s)]
#define A(t, s) alpha[(size_t)(t) * SP + (s)]
#define Bt(t, s) beta[(size_t)(t) * SP + (s)]
    double nll;
    if (T == 0) {
        nll = (S == 0) ? 0.0 : INFINITY;
    } else {
        for (int s = 0; s < SP; ++s) A(0, s) = -INFINITY;
        A(0, 0) = LP(0, 0);
        if (SP > 1) A(0, 1) = LP(0, 1);
        for (int t = 1; t < T; ++t)
            for (int s = 0; s < SP; ++s) {
                double a = A(t - 1, s);
                if (s >= 1) a = lae(a, A(t - 1, s - 1));
                if (s >= 2 && LAB(s) != blank && LAB(s) != LAB(s - 2))
                    a = lae(a, A(t - 1, s - 2));
                A(t, s) = (a == -INFINITY) ? -INFINITY : a + LP(t, s);
            }
        double ll = A(T - 1, SP - 1);
        if (SP > 1) ll = lae(ll, A(T - 1, SP - 2));
        nll = -ll;
    }
    if (grad && T > 0) {
        for (int s = 0; s < SP; ++s) Bt(T - 1, s) = -INFINITY;
        Bt(T - 1, SP - 1) = LP(T - 1, SP - 1);
        if (SP > 1) Bt(T - 1, SP - 2) = LP(T - 1, SP - 2);
        for (int t = T - 2; t >= 0; --t)
            for (int s = 0; s < SP; ++s) {
                double a = Bt(t + 1, s);
                if (s + 1 < SP) a = lae(a, Bt(t + 1, s + 1));
                if (s + 2 < SP && LAB(s) != blank && LAB(s) != LAB(s + 2))
                    a = lae(a, Bt(t + 1, s + 2));
                Bt(t, s) = (a == -INFINITY) ? -INFINITY : a + LP(t, s);
            }
        double *occ = (double *)malloc(sizeof(double) * V);
        for (int t = 0; t < T; ++t) {
            for (int v = 0; v < V; ++v) occ[v] = -INFINITY;
            for (int s = 0; s < SP; ++s) {
                const int v = LAB(s);
                occ[v] = lae(occ[v], A(t, s) + Bt(t, s));
            }
            float *grow = grad + (size_t)t * V;
            for (int v = 0; v < V; ++v) {
                const double l = lp[(size_t)t * V + v];
                grow[v] = (float)(exp(l) - exp(occ[v] + nll - l));
            }
        }
        free(occ);
    }
#undef LAB
#undef LP
#undef A
#undef Bt
    free(lp); free(alpha); free(beta);
    return nll;
}

/* nll [B] double (per-utterance, un-reduced); grad [B,Tmax,V] float or NULL
 * (= d nll_b / d logits[b], zero for t >= T_b).  Returns 0. */
int wr_oracle_ctc_f64(const float *logits, const int32_t *targets,
                      const int32_t *input_lengths, const int32_t *target_lengths,
                      int B, int Tmax, int Smax, int V, int blank,
                      double *nll, float *grad)
{
    const size_t per_b = (size_t)Tmax * V;
    if (grad) memset(grad, 0, sizeof(float) * per_b * (size_t)B);
    /* utterances are independent */
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b)
        nll[b] = ctc_one(logits + per_b * b, targets + (size_t)b * Smax,
                         input_lengths[b], target_lengths[b], Tmax, V, blank,
                         grad ? grad + per_b * b : NULL);
    return 0;
}

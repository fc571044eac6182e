/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * call into this file.
 *
 * Threaded float32 port of the RNN-T loss + gradient: the timed CPU baseline
 * ("cpu_baseline.kind": "port") that bench.py reports beside the GPU number.
 * It stands in for torchaudio.functional.rnnt_loss on CPU
 * (/root/reference/wenet/transducer/transducer.py:142-147), which cannot be
 * installed in this image.  Same algorithm as rnnt_oracle.c (SURVEY.md
 * App. A.1) at the reference's working precision, organised the way a CPU
 * wants it: OpenMP over lattice rows for the two streaming passes (row
 * log-sum-exp; gradient rows) and over utterances x direction for the
 * alpha / beta sweeps.  Compiled with -O3 -march=native -ffast-math so the
 * expf loops vectorise through libmvec; it therefore avoids infinities
 * (lattice borders are handled explicitly instead of with -inf).
 *
 * Checked against wr_oracle_rnnt_f64 in tests/test_oracle_rnnt.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

static inline float lae_f(float a, float b) {
    const float m = a > b ? a : b;
    return m + log1pf(expf(-fabsf(a - b)));
}

int wr_oracle_rnnt_f32(const float *logits, const int32_t *targets,
                       const int32_t *logit_lengths, const int32_t *target_lengths,
                       int B, int Tmax, int U1max, int V, int blank, float clamp,
                       float *costs, float *grad, int nthreads)
{
    const size_t cells = (size_t)Tmax * U1max;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
    float *denom = (float *)malloc(sizeof(float) * cells * B);
    float *skip  = (float *)malloc(sizeof(float) * cells * B);
    float *emit  = (float *)malloc(sizeof(float) * cells * B);
    float *alpha = (float *)malloc(sizeof(float) * cells * B);
    float *beta  = (float *)malloc(sizeof(float) * cells * B);

    /* pass 1: row log-sum-exp + gather of blank / label log-probs */
#pragma omp parallel for schedule(static)
    for (long r = 0; r < (long)(cells * B); ++r) {
        const int b = (int)(r / (long)cells);
        const int t = (int)((r % (long)cells) / U1max);
        const int u = (int)(r % U1max);
        const int T = logit_lengths[b], U = target_lengths[b];
        if (t >= T || u > U) continue;
        const float *row = logits + (size_t)r * V;
        float mx = row[0];
        for (int v = 1; v < V; ++v) mx = row[v] > mx ? row[v] : mx;
        float s = 0.f;
        for (int v = 0; v < V; ++v) s += expf(row[v] - mx);
        const float d = mx + logf(s);
        denom[r] = d;
        skip[r] = row[blank] - d;
        emit[r] = (u < U) ? row[targets[(size_t)b * (U1max - 1) + u]] - d : 0.f;
    }

#define IX(t, u) ((size_t)(t) * U1max + (u))
    /* pass 2: alpha / beta lattice sweeps */
#pragma omp parallel for schedule(dynamic, 1)
    for (int j = 0; j < 2 * B; ++j) {
        const int b = j >> 1;
        const int T = logit_lengths[b], U = target_lengths[b];
        const float *sk = skip + cells * b, *em = emit + cells * b;
        if ((j & 1) == 0) {
            float *al = alpha + cells * b;
            al[0] = 0.f;
            for (int u = 1; u <= U; ++u) al[IX(0, u)] = al[IX(0, u - 1)] + em[IX(0, u - 1)];
            for (int t = 1; t < T; ++t) {
                al[IX(t, 0)] = al[IX(t - 1, 0)] + sk[IX(t - 1, 0)];
                for (int u = 1; u <= U; ++u)
                    al[IX(t, u)] = lae_f(al[IX(t - 1, u)] + sk[IX(t - 1, u)],
                                         al[IX(t, u - 1)] + em[IX(t, u - 1)]);
            }
        } else {
            float *be = beta + cells * b;
            be[IX(T - 1, U)] = sk[IX(T - 1, U)];
            for (int u = U - 1; u >= 0; --u) be[IX(T - 1, u)] = be[IX(T - 1, u + 1)] + em[IX(T - 1, u)];
            for (int t = T - 2; t >= 0; --t) {
                be[IX(t, U)] = be[IX(t + 1, U)] + sk[IX(t, U)];
                for (int u = U - 1; u >= 0; --u)
                    be[IX(t, u)] = lae_f(be[IX(t + 1, u)] + sk[IX(t, u)],
                                         be[IX(t, u + 1)] + em[IX(t, u)]);
            }
            costs[b] = -be[0];
        }
    }

    /* pass 3: gradient rows */
    if (grad) {
#pragma omp parallel for schedule(static)
        for (long r = 0; r < (long)(cells * B); ++r) {
            const int b = (int)(r / (long)cells);
            const int t = (int)((r % (long)cells) / U1max);
            const int u = (int)(r % U1max);
            const int T = logit_lengths[b], U = target_lengths[b];
            float *grow = grad + (size_t)r * V;
            if (t >= T || u > U) { memset(grow, 0, sizeof(float) * V); continue; }
            const float *row = logits + (size_t)r * V;
            const float *be = beta + cells * b;
            const float cost = -be[0];
            const float c = alpha[r] + cost - denom[r];
            const float bt = be[IX(t, u)];
            const float cb = c + bt;
            for (int v = 0; v < V; ++v) grow[v] = expf(row[v] + cb);
            /* special entries (App. A.1 case chain, first match wins) */
            int blank_done = 0;
            {
                const float g = row[blank] + c;
                if (t == T - 1 && u == U) { grow[blank] = expf(g + bt) - expf(g); blank_done = 1; }
                else if (t < T - 1) { grow[blank] = expf(g + bt) - expf(g + be[IX(t + 1, u)]); blank_done = 1; }
            }
            if (u < U) {
                const int lab = targets[(size_t)b * (U1max - 1) + u];
                if (!(lab == blank && blank_done)) {
                    const float g = row[lab] + c;
                    grow[lab] = expf(g + bt) - expf(g + be[IX(t, u + 1)]);
                }
            }
            if (clamp > 0.f)
                for (int v = 0; v < V; ++v) {
                    float x = grow[v];
                    grow[v] = x > clamp ? clamp : (x < -clamp ? -clamp : x);
                }
        }
    }
#undef IX
    free(denom); free(skip); free(emit); free(alpha); free(beta);
    return 0;
}

int wr_oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

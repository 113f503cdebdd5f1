/*
 * oracle/zq_kernels_omp.c -- TEST / BENCH INFRASTRUCTURE ONLY (never shipped, never on the product path).
 *
 * The pCMF loop nest of oracle/zq_kernels.c (oriana/models/gap.py:67-80) with OpenMP over the cells: every
 * thread owns a contiguous block of rows, writes its rows of Z_hat_i directly and accumulates the per-gene sums
 * in a private (p, K) buffer; the buffers are added in thread order at the end.
 *
 * This is NOT the reference's behaviour: its numba kernel carries no `parallel` / `prange` (gap.py:67) and runs
 * on one thread, and the per-gene sums here are formed in a different order (thread partials).  bench.py reports
 * it next to the single-thread baseline as "what the host's cores could do", nothing is checked against it
 * except its agreement with the single-thread oracle (tests/test_oracle.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

static float pairwise_sum_f32(const float *a, int64_t n)
{
    if (n < 8) {
        float res = 0.0f;
        for (int64_t i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        float r[8];
        int64_t i;
        for (i = 0; i < 8; i++) r[i] = a[i];
        for (i = 8; i < n - (n % 8); i += 8) {
            r[0] += a[i + 0]; r[1] += a[i + 1]; r[2] += a[i + 2]; r[3] += a[i + 3];
            r[4] += a[i + 4]; r[5] += a[i + 5]; r[6] += a[i + 6]; r[7] += a[i + 7];
        }
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        return pairwise_sum_f32(a, n2) + pairwise_sum_f32(a + n2, n - n2);
    }
}

#define MAXK 4096

int zq_gap_omp(float *Z_hat_i, float *Z_hat_j, const float *log_U_hat, const float *log_V_hat, const float *X,
               int64_t n, int64_t p, int64_t K, int nthreads)
{
    if (K > MAXK || nthreads < 1) return -1;
    if (nthreads > n) nthreads = (int)(n > 0 ? n : 1);
    memset(Z_hat_i, 0, sizeof(float) * (size_t)(n * K));
    memset(Z_hat_j, 0, sizeof(float) * (size_t)(p * K));
    float **part = (float **)calloc((size_t)nthreads, sizeof(float *));
    if (!part) return -2;
    int fail = 0;
#pragma omp parallel num_threads(nthreads)
    {
        const int t = omp_get_thread_num(), nt = omp_get_num_threads();
        float *zj = (float *)calloc((size_t)(p * K), sizeof(float));
        float e[MAXK];
        if (t < nthreads) part[t] = zj;
        if (!zj) {
#pragma omp atomic write
            fail = 1;
        } else {
            const int64_t i0 = n * t / nt, i1 = n * (t + 1) / nt;
            for (int64_t i = i0; i < i1; i++) {
                const float *lu = log_U_hat + i * K;
                for (int64_t j = 0; j < p; j++) {
                    const float x = X[i * p + j];
                    const float *lv = log_V_hat + j * K;
                    for (int64_t k = 0; k < K; k++) e[k] = expf(lu[k] + lv[k]);
                    float den = pairwise_sum_f32(e, K);
                    den = (den > 0) ? den : 1.0f;
                    for (int64_t k = 0; k < K; k++) {
                        const float expectation = (x * e[k]) / den;
                        zj[j * K + k] += expectation;
                        Z_hat_i[i * K + k] += expectation;
                    }
                }
            }
        }
    }
    if (!fail) {
        const int64_t tot = p * K;
#pragma omp parallel for num_threads(nthreads) schedule(static)
        for (int64_t q = 0; q < tot; q++) {
            float acc = 0.0f;
            for (int t = 0; t < nthreads; t++)
                if (part[t]) acc += part[t][q];            /* thread partials in thread order */
            Z_hat_j[q] = acc;
        }
    }
    for (int t = 0; t < nthreads; t++) free(part[t]);
    free(part);
    return fail ? -2 : 0;
}

/*
 * oracle/zq_kernels.c -- TEST INFRASTRUCTURE ONLY (never shipped, never on the product path).
 *
 * Single-thread, plain-C, float32 restatement of the four numba loop nests of the
 * reference (AntoinePassemiers/Oriana).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load the library built from this file.
 *
 *   zq_gap          <- oriana/models/gap.py:67-80          GaP.compute_Z_q_expectations
 *   zq_zigap        <- oriana/models/zigap.py:79-95        ZIGaP.compute_Z_q_expectations
 *   zq_sparse_gap   <- oriana/models/sparse_gap.py:81-97   SparseGaP.compute_Z_q_expectations
 *   zq_sparse_zigap <- oriana/models/sparse_zigap.py:100-116
 *
 * Conventions mirror the reference: outputs first, inputs after; C-contiguous f32
 * 2-D arrays; the callee zero-fills the outputs; no max-subtraction in the softmax;
 * `den = den if den > 0 else 1`.
 *
 * Arithmetic notes (what "the same algorithm" means here):
 *   - e_k = expf(lu_ik + lv_jk): f32 add, then f32 exp (denormals kept).
 *   - den = sum_k e_k follows NumPy's pairwise float32 summation for a contiguous
 *     vector (np.add.reduce): plain left-to-right for K < 8, eight interleaved
 *     partial sums otherwise (blocks of 128).  This is what the reference evaluates
 *     when run un-jitted (how the golden vectors were captured); numba would sum
 *     left-to-right -- the two differ by < 1 f32 ulp of den.
 *   - r = (x * e_k) / den in f32, accumulated sequentially in f32 over j (row sums)
 *     and over i (column sums), exactly in loop order.
 *
 * Compile WITHOUT -ffast-math / -ffp-contract (see oracle/Makefile) so that no
 * multiply-add gets fused.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

/* NumPy pairwise sum for float32 (numpy/core/src/umath/loops_utils.h.src semantics). */
static float np_pairwise_sum_f32(const float *a, int64_t n)
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
        return np_pairwise_sum_f32(a, n2) + np_pairwise_sum_f32(a + n2, n - n2);
    }
}

#define MAXK 4096

/* oriana/models/gap.py:67-80 */
int zq_gap(float *Z_hat_i, float *Z_hat_j,
           const float *log_U_hat, const float *log_V_hat, const float *X,
           int64_t n, int64_t p, int64_t K)
{
    if (K > MAXK) return -1;
    float e[MAXK];
    memset(Z_hat_i, 0, sizeof(float) * (size_t)(n * K));
    memset(Z_hat_j, 0, sizeof(float) * (size_t)(p * K));
    for (int64_t i = 0; i < n; i++) {
        const float *lu = log_U_hat + i * K;
        for (int64_t j = 0; j < p; j++) {
            const float *lv = log_V_hat + j * K;
            for (int64_t k = 0; k < K; k++) e[k] = expf(lu[k] + lv[k]);
            float den = np_pairwise_sum_f32(e, K);
            den = (den > 0) ? den : 1.0f;
            const float x = X[i * p + j];
            for (int64_t k = 0; k < K; k++) {
                float expectation = (x * e[k]) / den;
                Z_hat_j[j * K + k] += expectation;
                Z_hat_i[i * K + k] += expectation;
            }
        }
    }
    return 0;
}

/* oriana/models/zigap.py:79-95.  quirk != 0 reproduces the D_hat[i, k] index of
 * zigap.py:94 (requires K <= p); quirk == 0 uses the evident D_hat[i, j]. */
int zq_zigap(float *DZ_hat_i, float *DZ_hat_j, float *DZ_exp_logsum_hat,
             const float *log_U_hat, const float *log_V_hat, const float *D_hat,
             const float *X, int64_t n, int64_t p, int64_t K, int quirk)
{
    if (K > MAXK || (quirk && K > p)) return -1;
    float e[MAXK], ls[MAXK];
    memset(DZ_hat_i, 0, sizeof(float) * (size_t)(n * K));
    memset(DZ_hat_j, 0, sizeof(float) * (size_t)(p * K));
    memset(DZ_exp_logsum_hat, 0, sizeof(float) * (size_t)(p * K));
    for (int64_t i = 0; i < n; i++) {
        const float *lu = log_U_hat + i * K;
        for (int64_t j = 0; j < p; j++) {
            const float *lv = log_V_hat + j * K;
            for (int64_t k = 0; k < K; k++) { ls[k] = lu[k] + lv[k]; e[k] = expf(ls[k]); }
            float den = np_pairwise_sum_f32(e, K);
            den = (den > 0) ? den : 1.0f;
            const float x = X[i * p + j];
            const float d = D_hat[i * p + j];
            for (int64_t k = 0; k < K; k++) {
                float expectation = (x * e[k]) / den;
                DZ_hat_i[i * K + k] += d * expectation;
                DZ_hat_j[j * K + k] += (quirk ? D_hat[i * p + k] : d) * expectation;
                DZ_exp_logsum_hat[j * K + k] += (d * expectation) * ls[k];
            }
        }
    }
    return 0;
}

/* oriana/models/sparse_gap.py:81-97 */
int zq_sparse_gap(float *SZ_hat_i, float *Z_hat_j, float *Z_exp_logsum_hat,
                  const float *log_U_hat, const float *log_V_hat,
                  const float *S_tilde, const float *S_hat, const float *X,
                  int64_t n, int64_t p, int64_t K)
{
    if (K > MAXK) return -1;
    float e[MAXK], ls[MAXK];
    memset(SZ_hat_i, 0, sizeof(float) * (size_t)(n * K));
    memset(Z_hat_j, 0, sizeof(float) * (size_t)(p * K));
    memset(Z_exp_logsum_hat, 0, sizeof(float) * (size_t)(p * K));
    for (int64_t i = 0; i < n; i++) {
        const float *lu = log_U_hat + i * K;
        for (int64_t j = 0; j < p; j++) {
            const float *lv = log_V_hat + j * K;
            const float *st = S_tilde + j * K;
            const float *sh = S_hat + j * K;
            for (int64_t k = 0; k < K; k++) { ls[k] = lu[k] + lv[k]; e[k] = expf(ls[k]) * st[k]; }
            float den = np_pairwise_sum_f32(e, K);
            den = (den > 0) ? den : 1.0f;
            const float x = X[i * p + j];
            for (int64_t k = 0; k < K; k++) {
                float expectation = (x * e[k]) / den;
                SZ_hat_i[i * K + k] += sh[k] * expectation;
                Z_hat_j[j * K + k] += expectation;
                Z_exp_logsum_hat[j * K + k] += expectation * ls[k];
            }
        }
    }
    return 0;
}

/* oriana/models/sparse_zigap.py:100-116 */
int zq_sparse_zigap(float *DSZ_hat, float *DZ_hat, float *DZ_exp_logsum_hat,
                    const float *log_U_hat, const float *log_V_hat,
                    const float *S_tilde, const float *S_hat, const float *D_hat,
                    const float *X, int64_t n, int64_t p, int64_t K)
{
    if (K > MAXK) return -1;
    float e[MAXK], ls[MAXK];
    memset(DSZ_hat, 0, sizeof(float) * (size_t)(n * K));
    memset(DZ_hat, 0, sizeof(float) * (size_t)(p * K));
    memset(DZ_exp_logsum_hat, 0, sizeof(float) * (size_t)(p * K));
    for (int64_t i = 0; i < n; i++) {
        const float *lu = log_U_hat + i * K;
        for (int64_t j = 0; j < p; j++) {
            const float *lv = log_V_hat + j * K;
            const float *st = S_tilde + j * K;
            const float *sh = S_hat + j * K;
            for (int64_t k = 0; k < K; k++) { ls[k] = lu[k] + lv[k]; e[k] = expf(ls[k]) * st[k]; }
            float den = np_pairwise_sum_f32(e, K);
            den = (den > 0) ? den : 1.0f;
            const float x = X[i * p + j];
            const float d = D_hat[i * p + j];
            for (int64_t k = 0; k < K; k++) {
                float expectation = (x * e[k]) / den;
                DSZ_hat[i * K + k] += (d * sh[k]) * expectation;
                DZ_hat[j * K + k] += d * expectation;
                DZ_exp_logsum_hat[j * K + k] += (d * expectation) * ls[k];
            }
        }
    }
    return 0;
}

/* ---- the same two nests with the zero counts skipped -------------------------------------------------------------------
 * For x == 0 every term of gap.py:78-80 / sparse_gap.py:93-97 is (0 * e_k) / den = +0 (or -0 against a negative log sum):
 * adding it changes no sum.  So zq_gap_nz / zq_sparse_gap_nz return bit for bit what zq_gap / zq_sparse_gap return WHENEVER
 * every expf(lu + lv) is finite (an overflowed term would make 0 * inf = NaN in the full nest) -- tests/test_oracle.py pins
 * that identity on the golden kernel I/O and on random inputs.  They exist so that -m gpu tests can run whole oracle sweeps at
 * sizes where the kernels of the benchmarked configurations engage (n K >= 2^20: 1e8 entries x K exponentials in the full
 * nest, 96 % of them for zero counts). */
int zq_gap_nz(float *Z_hat_i, float *Z_hat_j,
              const float *log_U_hat, const float *log_V_hat, const float *X,
              int64_t n, int64_t p, int64_t K)
{
    if (K > MAXK) return -1;
    float e[MAXK];
    memset(Z_hat_i, 0, sizeof(float) * (size_t)(n * K));
    memset(Z_hat_j, 0, sizeof(float) * (size_t)(p * K));
    for (int64_t i = 0; i < n; i++) {
        const float *lu = log_U_hat + i * K;
        for (int64_t j = 0; j < p; j++) {
            const float x = X[i * p + j];
            if (x == 0.0f) continue;
            const float *lv = log_V_hat + j * K;
            for (int64_t k = 0; k < K; k++) e[k] = expf(lu[k] + lv[k]);
            float den = np_pairwise_sum_f32(e, K);
            den = (den > 0) ? den : 1.0f;
            for (int64_t k = 0; k < K; k++) {
                float expectation = (x * e[k]) / den;
                Z_hat_j[j * K + k] += expectation;
                Z_hat_i[i * K + k] += expectation;
            }
        }
    }
    return 0;
}

int zq_sparse_gap_nz(float *SZ_hat_i, float *Z_hat_j, float *Z_exp_logsum_hat,
                     const float *log_U_hat, const float *log_V_hat,
                     const float *S_tilde, const float *S_hat, const float *X,
                     int64_t n, int64_t p, int64_t K)
{
    if (K > MAXK) return -1;
    float e[MAXK], ls[MAXK];
    memset(SZ_hat_i, 0, sizeof(float) * (size_t)(n * K));
    memset(Z_hat_j, 0, sizeof(float) * (size_t)(p * K));
    memset(Z_exp_logsum_hat, 0, sizeof(float) * (size_t)(p * K));
    for (int64_t i = 0; i < n; i++) {
        const float *lu = log_U_hat + i * K;
        for (int64_t j = 0; j < p; j++) {
            const float x = X[i * p + j];
            if (x == 0.0f) continue;
            const float *lv = log_V_hat + j * K;
            const float *st = S_tilde + j * K;
            const float *sh = S_hat + j * K;
            for (int64_t k = 0; k < K; k++) { ls[k] = lu[k] + lv[k]; e[k] = expf(ls[k]) * st[k]; }
            float den = np_pairwise_sum_f32(e, K);
            den = (den > 0) ? den : 1.0f;
            for (int64_t k = 0; k < K; k++) {
                float expectation = (x * e[k]) / den;
                SZ_hat_i[i * K + k] += sh[k] * expectation;
                Z_hat_j[j * K + k] += expectation;
                Z_exp_logsum_hat[j * K + k] += expectation * ls[k];
            }
        }
    }
    return 0;
}

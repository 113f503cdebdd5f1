// dense.hip -- element-wise pieces of the zero-inflated and sparse models (gfx950).
//
// Bernoulli posterior updates of the dropout node D (oriana/models/zigap.py:130-136,
// sparse_zigap.py:163-169) and of the sparsity node S (sparse_gap.py:134-141,
// sparse_zigap.py:154-161), helper maps on factor matrices, and wide column / row means for the
// M-step (zigap.py:158, sparse_gap.py:165).  All HBM-bound float64 streaming kernels.
#include "common.h"

namespace oriana {

// p_d[i,j] = sigmoid(logit(pi_d[j]) - Lambda[i,j]);  column overrides for pi_d <= 0 / >= 1
// (zigap.py:131-134);  D_hat = float32(p_d) (bernoulli.py:45).  Lambda may alias p_d (in place).
// nzmask (optional): word [(i / 32) * m + j] bit (i % 32) set iff X[i, j] != 0 (oriana_nzmask_f32) -- then
// the override p_d[X != 0] = 1 - 1e-10 (zigap.py:135) is applied here.  colsum (optional): += sum_i p_d[i, j].
__global__ __launch_bounds__(256) void k_dropout_update(double *__restrict__ p_d, float *__restrict__ D_hat,
                                                        const double *Lambda, const double *__restrict__ pi_d,
                                                        const uint32_t *__restrict__ nzmask,
                                                        double *__restrict__ colsum, int64_t rows, int64_t m) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    const double pi = pi_d[j];
    const double lg = logit_f64(pi);
    const int64_t r0 = (int64_t)blockIdx.y * 64;
    const int64_t r1 = (r0 + 64 < rows) ? r0 + 64 : rows;
    uint32_t w0 = 0, w1 = 0;
    if (nzmask) {
        w0 = nzmask[(r0 >> 5) * m + j];
        if (r0 + 32 < rows) w1 = nzmask[((r0 >> 5) + 1) * m + j];
    }
    double cs = 0.0;
    for (int64_t i = r0; i < r1; ++i) {
        const int64_t idx = i * m + j;
        double p = sigmoid_f64(lg - Lambda[idx]);
        if (pi <= 0.0) p = 1e-10;
        if (pi >= 1.0) p = 1.0 - 1e-10;
        const int b = (int)(i - r0);
        if (((b < 32 ? w0 >> b : w1 >> (b - 32)) & 1u)) p = 1.0 - 1e-10;
        p_d[idx] = p;
        D_hat[idx] = (float)p;
        cs += p;
    }
    if (colsum) atomicAdd(&colsum[j], cs);
}

// bit mask of the non-zero entries of a dense (rows, m) f32 matrix: one word per (32 rows, column)
__global__ __launch_bounds__(256) void k_nzmask(uint32_t *__restrict__ mask, const float *__restrict__ D, int64_t rows,
                                                int64_t m) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t w = blockIdx.y;                                   // block of 32 rows
    if (j >= m) return;
    uint32_t bits = 0;
    for (int b = 0; b < 32; ++b) {
        const int64_t i = w * 32 + b;
        if (i < rows && D[i * m + j] != 0.f) bits |= (1u << b);
    }
    mask[w * m + j] = bits;
}

// p_d[X != 0] = 1 - 1e-10 (zigap.py:135): one thread per row-side slot of the tiled layout.
__global__ __launch_bounds__(256) void k_dropout_fix_nz(oriana_counts cm, double *__restrict__ p_d,
                                                        float *__restrict__ D_hat, double one, int64_t ld) {
    const int64_t t = blockIdx.x;
    const int64_t rb = t / cm.ncb, cb = t - rb * cm.ncb;
    const int64_t rbase = cm.roff[t];
    for (int sl = 0; sl < 16; ++sl) {
        const uint32_t s0 = cm.rslice[t * 17 + sl], s1 = cm.rslice[t * 17 + sl + 1];
        for (uint32_t slot = s0 + threadIdx.x; slot < s1; slot += 256) {
            const oriana_rowrec rec = cm.rowrec[rbase + slot];
            if (rec.x == 0.f) continue;
            const int64_t ip = rb * TILE + sl * 16 + (int)(((slot - s0) & 63u) >> 2);
            const int64_t jp = cb * TILE + rec.col;
            const int64_t i = cm.row_perm ? (int64_t)cm.row_perm[ip] : ip;
            const int64_t j = cm.col_perm ? (int64_t)cm.col_perm[jp] : jp;
            if (p_d) p_d[i * ld + j] = one;
            if (D_hat) D_hat[i * ld + j] = (float)one;
        }
    }
}

// out[j] += sum_i A[i,j] for a wide (rows, m) f64 matrix (pi_d = mean(p_d, axis=0), zigap.py:158)
__global__ __launch_bounds__(256) void k_colsum_wide(double *__restrict__ out, const double *__restrict__ A,
                                                     int64_t rows, int64_t m) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    const int64_t r0 = (int64_t)blockIdx.y * 256;
    const int64_t r1 = (r0 + 256 < rows) ? r0 + 256 : rows;
    double s = 0.0;
    for (int64_t i = r0; i < r1; ++i) s += A[i * m + j];
    atomicAdd(&out[j], s);
}

// the same for a float32 matrix (column sums of D_hat while p_d == D_hat exactly, zigap.py:77)
__global__ __launch_bounds__(256) void k_colsum_wide_f32(double *__restrict__ out, const float *__restrict__ A,
                                                     int64_t rows, int64_t m) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    const int64_t r0 = (int64_t)blockIdx.y * 256;
    const int64_t r1 = (r0 + 256 < rows) ? r0 + 256 : rows;
    double s = 0.0;
    for (int64_t i = r0; i < r1; ++i) s += A[i * m + j];
    atomicAdd(&out[j], s);
}

// out[i] = mean_k A[i,k]  (pi_s = mean(p_s, axis=1), sparse_gap.py:165)
__global__ __launch_bounds__(256) void k_rowmean(double *__restrict__ out, const double *__restrict__ A, int64_t r, int K) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= r) return;
    double s = 0.0;
    for (int k = 0; k < K; ++k) s += A[i * K + k];       // NumPy sums a short contiguous row left to right
    out[i] = s / (double)K;
}

// S_q update (sparse_gap.py:134-141):
//   tmp = -Zlog + nan_to_num(c * Vprime_hat);  p_s = nan_to_num(sigmoid(logit(pi_s)[:, None] - tmp))
//   rows with pi_s <= 0 -> 1e-10, pi_s >= 1 -> 1 - 1e-10;  S_hat = float32(p_s)
__global__ __launch_bounds__(256) void k_sparsity_update(double *__restrict__ p_s, float *__restrict__ S_hat,
                                                         const double *__restrict__ pi_s, const float *__restrict__ Zlog,
                                                         const double *__restrict__ c_vec, const double *__restrict__ c_mat,
                                                         const double *__restrict__ Vp, int64_t m, int K) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= m * K) return;
    const int64_t j = idx / K;
    const int k = (int)(idx - j * K);
    const double c = c_mat ? c_mat[idx] : c_vec[k];
    // tmp = -Z (float32) ; tmp += nan_to_num(c * Vprime)  (float32 += float64 rounds to float32)
    float tmp = -Zlog[idx];
    tmp = (float)((double)tmp + nan_to_num(c * Vp[idx]));
    const double pi = pi_s[j];
    double p = nan_to_num(sigmoid_f64(logit_f64(pi) - (double)tmp));
    if (pi <= 0.0) p = 1e-10;
    if (pi >= 1.0) p = 1.0 - 1e-10;
    p_s[idx] = p;
    S_hat[idx] = (float)p;
}

// (p > tau) as float32 (sparse_gap.py:113)
__global__ __launch_bounds__(256) void k_threshold(float *__restrict__ out, const double *__restrict__ p, double tau, int64_t len) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < len) out[i] = (p[i] > tau) ? 1.0f : 0.0f;
}

// Fout[i, k] = Fin[i, k] * mul[src(i), k] (padded (r, Kp) factor times a dense (., K) matrix).
// zero_guard: where Fin == 0 the product is 0 whatever mul holds (mul may be -1e15, inf or NaN).
__global__ __launch_bounds__(256) void k_scale_factor(float *__restrict__ Fout, const float *__restrict__ Fin,
                                                      const float *__restrict__ mul, const int32_t *__restrict__ row_index,
                                                      int64_t r, int K, int Kp, int zero_guard) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= r * Kp) return;
    const int64_t row = idx / Kp;
    const int k = (int)(idx - row * Kp);
    float v = 0.f;
    if (k < K) {
        const float f = Fin[idx];
        const int64_t src = row_index ? (int64_t)row_index[row] : row;
        v = (zero_guard && f == 0.f) ? 0.f : f * mul[src * K + k];
    }
    Fout[idx] = v;
}

// The log sums sum_i r_ijk (lu_ik + lv_jk) (zigap.py:95) are evaluated as FV (sum_i s FU (lu - a_k)) + (lv + a_k) FV sum_i s FU
// with a per-factor centre a_k: split as two plain sums (a = 0) the two terms are each |lu| times larger than their
// sum when the sweeps have drifted along the scale indeterminacy (lu ~ +45, lv ~ -27), and the float32 column sums then
// lose that factor in precision -- the sparsity posterior p_s = sigmoid(logit(pi_s) - (c V' - Zlog)) amplifies it (measured:
// 1.4e-4 on p_s against 2e-6 for the reference's own float32 loop).  a_k = mean of lu_ik over the
// cells that carry weight (FU_ik > 1e-20 of the row's largest factor, finite log), weighted by the cell's own
// responsibility sums Z_i[i, k] when they are at hand (the r-weighted mean of lu): acc = {sum w lu, sum w} per factor.
constexpr int LC_ROWS = 256;
__global__ __launch_bounds__(256) void k_log_center(double *__restrict__ acc, const float *__restrict__ F,
                                                    const float *__restrict__ logF, const float *__restrict__ W,
                                                    const int32_t *__restrict__ row_index, int64_t r, int K, int Kp) {
    // thread = (row group, factor): a factor's LC_ROWS / RG rows of the block are summed in registers (coalesced across
    // the factors), the row groups through LDS, one pair of float64 atomics per factor and block.  (256 rows per block:
    // with 1024 the 500,000 cells of configs[4] were 489 blocks of serial row chains -- 0.33 ms for 384 MB.)
    __shared__ double ssum[256], scnt[256];
    const int Kc = (K <= 64) ? 64 : (K <= 128) ? 128 : 256, RG = 256 / Kc;
    const int k = threadIdx.x % Kc, rg = threadIdx.x / Kc;
    double sum = 0.0, cnt = 0.0;
    if (k < K) {
        const int64_t r0 = (int64_t)blockIdx.x * LC_ROWS;
        const int64_t r1 = (r0 + LC_ROWS < r) ? r0 + LC_ROWS : r;
        for (int64_t row = r0 + rg; row < r1; row += RG) {
            const int64_t src = row_index ? (int64_t)row_index[row] : row;
            const float f = F[row * Kp + k], l = logF[src * K + k];
            if (f > 1e-20f && fabsf(l) < 1e30f) {          // (a rejected row is the constant 1e-30: it never counts)
                const double w = W ? (double)W[src * K + k] : 1.0;
                if (w > 0.0 && w < 1e300) { sum += w * (double)l; cnt += w; }
            }
        }
    }
    ssum[threadIdx.x] = sum;
    scnt[threadIdx.x] = cnt;
    __syncthreads();
    if (rg == 0 && k < K) {
        for (int g = 1; g < RG; ++g) { sum += ssum[g * Kc + k]; cnt += scnt[g * Kc + k]; }
        if (cnt > 0.0) { atomicAdd(&acc[k], sum); atomicAdd(&acc[K + k], cnt); }
    }
}

__device__ __forceinline__ double log_center(const double *__restrict__ acc, int K, int k) {
    if (!acc) return 0.0;
    const double c = acc[K + k];
    return c > 0.0 ? acc[k] / c : 0.0;
}

// Fout[i, k] = Fin[i, k] * (mul[src(i), k] - a_k), 0 where Fin == 0 (the E[log U]-weighted factor of the log sums)
__global__ __launch_bounds__(256) void k_scale_factor_centered(float *__restrict__ Fout, const float *__restrict__ Fin,
                                                               const float *__restrict__ mul, const double *__restrict__ acc,
                                                               const int32_t *__restrict__ row_index, int64_t r, int K, int Kp) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= r * Kp) return;
    const int64_t row = idx / Kp;
    const int k = (int)(idx - row * Kp);
    float v = 0.f;
    if (k < K) {
        const float f = Fin[idx];
        const int64_t src = row_index ? (int64_t)row_index[row] : row;
        if (f != 0.f) v = f * (float)((double)mul[src * K + k] - log_center(acc, K, k));
    }
    Fout[idx] = v;
}

// Zlog[o,k] += FV[j,k] * (C2[j,k] + (lv[o,k] + a_k) * C[j,k])   (o = row_index[j]; combined in float64)
__global__ __launch_bounds__(256) void k_finalize_zlog(float *__restrict__ Zlog, const float *__restrict__ FV,
                                                       const float *__restrict__ C2, const float *__restrict__ C,
                                                       const float *__restrict__ logV, const double *__restrict__ acc,
                                                       const int32_t *__restrict__ row_index, int64_t r, int K, int Kp) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= r * K) return;
    const int64_t row = idx / K;
    const int k = (int)(idx - row * K);
    const int64_t o = (row_index ? (int64_t)row_index[row] : row) * K + k;
    const float f = FV[row * Kp + k];
    float v = 0.f;
    if (f != 0.f)
        v = (float)((double)f * ((double)C2[row * Kp + k] + ((double)logV[o] + log_center(acc, K, k)) * (double)C[row * Kp + k]));
    Zlog[o] += v;
}

// out = A (f64) * B (f32), element-wise (V_hat = S_hat * Vprime_hat, sparse_gap.py:118)
__global__ __launch_bounds__(256) void k_mul_f64_f32(double *__restrict__ out, const double *__restrict__ A,
                                                     const float *__restrict__ B, int64_t len) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < len) out[i] = (double)B[i] * A[i];
}

// dq[i, k] = D[i, k] for k < K: the first K columns of the dense (rows, m) matrix (zigap.py:94)
__global__ __launch_bounds__(256) void k_take_cols(float *__restrict__ out, const float *__restrict__ D, int64_t rows,
                                                   int64_t m, int K) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * K) return;
    const int64_t i = idx / K;
    const int k = (int)(idx - i * K);
    out[idx] = D[i * m + k];
}

}  // namespace oriana

using namespace oriana;

extern "C" int oriana_dropout_update(double *p_d, float *D_hat, const double *Lambda, const double *pi_d,
                                     const uint32_t *nzmask, double *colsum, int64_t rows, int64_t m, void *stream) {
    if (rows < 0 || m < 0) return ORIANA_EINVAL;
    if (rows == 0 || m == 0) return 0;
    if (!p_d || !D_hat || !Lambda || !pi_d) return ORIANA_EINVAL;
    for (int64_t y0 = 0; y0 * 64 < rows; y0 += 65535) {
        const int64_t ny = ((rows + 63) / 64 - y0 < 65535) ? (rows + 63) / 64 - y0 : 65535;
        hipLaunchKernelGGL(k_dropout_update, dim3((unsigned)((m + 255) / 256), (unsigned)ny), dim3(256), 0,
                           (hipStream_t)stream, p_d + y0 * 64 * m, D_hat + y0 * 64 * m, Lambda + y0 * 64 * m, pi_d,
                           nzmask ? nzmask + y0 * 2 * m : nullptr, colsum, rows - y0 * 64, m);
    }
    ORIANA_LAUNCH_CHECK();
    return 0;
}

extern "C" int oriana_nzmask_f32(uint32_t *mask, const float *D, int64_t rows, int64_t m, void *stream) {
    if (rows < 0 || m < 0) return ORIANA_EINVAL;
    if (rows == 0 || m == 0) return 0;
    if (!mask || !D) return ORIANA_EINVAL;
    const int64_t nw = (rows + 31) / 32;
    for (int64_t y0 = 0; y0 < nw; y0 += 65535) {
        const int64_t ny = (nw - y0 < 65535) ? nw - y0 : 65535;
        hipLaunchKernelGGL(k_nzmask, dim3((unsigned)((m + 255) / 256), (unsigned)ny), dim3(256), 0, (hipStream_t)stream,
                           mask + y0 * m, D + y0 * 32 * m, rows - y0 * 32, m);
    }
    ORIANA_LAUNCH_CHECK();
    return 0;
}

extern "C" int oriana_dropout_fix_nz_ld(const oriana_counts *cm, double *p_d, float *D_hat, double value, int64_t ld,
                                        void *stream) {
    if (!cm || (!p_d && !D_hat) || ld < cm->m) return ORIANA_EINVAL;
    const int64_t nt = cm->nrb * cm->ncb;
    if (nt == 0 || cm->rslots == 0) return 0;
    hipLaunchKernelGGL(k_dropout_fix_nz, dim3((unsigned)nt), dim3(256), 0, (hipStream_t)stream, *cm, p_d, D_hat, value, ld);
    ORIANA_LAUNCH_CHECK();
    return 0;
}

extern "C" int oriana_dropout_fix_nz(const oriana_counts *cm, double *p_d, float *D_hat, double value, void *stream) {
    return cm ? oriana_dropout_fix_nz_ld(cm, p_d, D_hat, value, cm->m, stream) : ORIANA_EINVAL;
}

extern "C" int oriana_colsum_wide_f64(double *out, const double *A, int64_t rows, int64_t m, void *stream) {
    if (rows < 0 || m < 0) return ORIANA_EINVAL;
    if (rows == 0 || m == 0) return 0;
    if (!out || !A) return ORIANA_EINVAL;
    for (int64_t y0 = 0; y0 * 256 < rows; y0 += 65535) {
        const int64_t ny = ((rows + 255) / 256 - y0 < 65535) ? (rows + 255) / 256 - y0 : 65535;
        hipLaunchKernelGGL(k_colsum_wide, dim3((unsigned)((m + 255) / 256), (unsigned)ny), dim3(256), 0,
                           (hipStream_t)stream, out, A + y0 * 256 * m, rows - y0 * 256, m);
    }
    ORIANA_LAUNCH_CHECK();
    return 0;
}

extern "C" int oriana_colsum_wide_f32(double *out, const float *A, int64_t rows, int64_t m, void *stream) {
    if (rows < 0 || m < 0) return ORIANA_EINVAL;
    if (rows == 0 || m == 0) return 0;
    if (!out || !A) return ORIANA_EINVAL;
    for (int64_t y0 = 0; y0 * 256 < rows; y0 += 65535) {
        const int64_t ny = ((rows + 255) / 256 - y0 < 65535) ? (rows + 255) / 256 - y0 : 65535;
        hipLaunchKernelGGL(k_colsum_wide_f32, dim3((unsigned)((m + 255) / 256), (unsigned)ny), dim3(256), 0,
                           (hipStream_t)stream, out, A + y0 * 256 * m, rows - y0 * 256, m);
    }
    ORIANA_LAUNCH_CHECK();
    return 0;
}

extern "C" int oriana_rowmean_f64(double *out, const double *A, int64_t r, int64_t K, void *stream) {
    if (r < 0 || K <= 0) return ORIANA_EINVAL;
    if (r == 0) return 0;
    if (!out || !A) return ORIANA_EINVAL;
    hipLaunchKernelGGL(k_rowmean, dim3((unsigned)((r + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, A, r, (int)K);
    ORIANA_LAUNCH_CHECK();
    return 0;
}

extern "C" int oriana_sparsity_update(double *p_s, float *S_hat, const double *pi_s, const float *Zlog,
                                      const double *c_vec, const double *c_mat, const double *Vprime_hat,
                                      int64_t m, int64_t K, void *stream) {
    if (m < 0 || K <= 0) return ORIANA_EINVAL;
    if (m == 0) return 0;
    if (!p_s || !S_hat || !pi_s || !Zlog || !Vprime_hat || (!c_vec && !c_mat)) return ORIANA_EINVAL;
    hipLaunchKernelGGL(k_sparsity_update, dim3((unsigned)((m * K + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       p_s, S_hat, pi_s, Zlog, c_vec, c_mat, Vprime_hat, m, (int)K);
    ORIANA_LAUNCH_CHECK();
    return 0;
}

extern "C" int oriana_threshold_f32(float *out, const double *p, double tau, int64_t len, void *stream) {
    if (len < 0) return ORIANA_EINVAL;
    if (len == 0) return 0;
    if (!out || !p) return ORIANA_EINVAL;
    hipLaunchKernelGGL(k_threshold, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, p, tau, len);
    ORIANA_LAUNCH_CHECK();
    return 0;
}

extern "C" int oriana_scale_factor(float *Fout, const float *Fin, const float *mul, const int32_t *row_index,
                                   int64_t r, int64_t K, int zero_guard, void *stream) {
    const int64_t Kp = oriana_kpad(K);
    if (r < 0 || K <= 0) return ORIANA_EINVAL;
    if (Kp == 0) return ORIANA_EKRANGE;
    if (r == 0) return 0;
    if (!Fout || !Fin || !mul) return ORIANA_EINVAL;
    hipLaunchKernelGGL(k_scale_factor, dim3((unsigned)((r * Kp + 255) / 256)), dim3(256), 0, (hipStream_t)stream, Fout,
                       Fin, mul, row_index, r, (int)K, (int)Kp, zero_guard);
    ORIANA_LAUNCH_CHECK();
    return 0;
}

extern "C" int oriana_log_center(double *acc, const float *F, const float *logF, const float *W, const int32_t *row_index,
                                 int64_t r, int64_t K, void *stream) {
    const int64_t Kp = oriana_kpad(K);
    if (r < 0 || K <= 0) return ORIANA_EINVAL;
    if (Kp == 0) return ORIANA_EKRANGE;
    if (!acc) return ORIANA_EINVAL;
    hipError_t e = hipMemsetAsync(acc, 0, sizeof(double) * 2 * K, (hipStream_t)stream);
    if (e != hipSuccess) return -1000 - (int)e;
    if (r == 0) return 0;
    if (!F || !logF) return ORIANA_EINVAL;
    hipLaunchKernelGGL(k_log_center, dim3((unsigned)((r + LC_ROWS - 1) / LC_ROWS)), dim3(256), 0, (hipStream_t)stream, acc, F, logF,
                       W, row_index, r, (int)K, (int)Kp);
    ORIANA_LAUNCH_CHECK();
    return 0;
}

extern "C" int oriana_scale_factor_centered(float *Fout, const float *Fin, const float *mul, const double *acc,
                                            const int32_t *row_index, int64_t r, int64_t K, void *stream) {
    const int64_t Kp = oriana_kpad(K);
    if (r < 0 || K <= 0) return ORIANA_EINVAL;
    if (Kp == 0) return ORIANA_EKRANGE;
    if (r == 0) return 0;
    if (!Fout || !Fin || !mul) return ORIANA_EINVAL;
    hipLaunchKernelGGL(k_scale_factor_centered, dim3((unsigned)((r * Kp + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       Fout, Fin, mul, acc, row_index, r, (int)K, (int)Kp);
    ORIANA_LAUNCH_CHECK();
    return 0;
}

extern "C" int oriana_finalize_zlog(float *Zlog, const float *FV, const float *C2, const float *C, const float *logV,
                                    const double *acc, const int32_t *row_index, int64_t r, int64_t K, void *stream) {
    const int64_t Kp = oriana_kpad(K);
    if (r < 0 || K <= 0) return ORIANA_EINVAL;
    if (Kp == 0) return ORIANA_EKRANGE;
    if (r == 0) return 0;
    if (!Zlog || !FV || !C2 || !C || !logV) return ORIANA_EINVAL;
    hipLaunchKernelGGL(k_finalize_zlog, dim3((unsigned)((r * K + 255) / 256)), dim3(256), 0, (hipStream_t)stream, Zlog,
                       FV, C2, C, logV, acc, row_index, r, (int)K, (int)Kp);
    ORIANA_LAUNCH_CHECK();
    return 0;
}

extern "C" int oriana_mul_f64_f32(double *out, const double *A, const float *B, int64_t len, void *stream) {
    if (len < 0) return ORIANA_EINVAL;
    if (len == 0) return 0;
    if (!out || !A || !B) return ORIANA_EINVAL;
    hipLaunchKernelGGL(k_mul_f64_f32, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, A, B, len);
    ORIANA_LAUNCH_CHECK();
    return 0;
}

extern "C" int oriana_take_cols_f32(float *out, const float *D, int64_t rows, int64_t m, int64_t K, void *stream) {
    if (rows < 0 || m < 0 || K <= 0 || K > m) return ORIANA_EINVAL;
    if (rows == 0) return 0;
    if (!out || !D) return ORIANA_EINVAL;
    hipLaunchKernelGGL(k_take_cols, dim3((unsigned)((rows * K + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, D,
                       rows, m, (int)K);
    ORIANA_LAUNCH_CHECK();
    return 0;
}

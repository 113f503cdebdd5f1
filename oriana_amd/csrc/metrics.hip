// metrics.hip -- the sparse-side sums of the deviance / Frobenius metrics (reference
// oriana/models/base.py:58-87 with loglikelihood_X, sparse_zigap.py:44-51).  Lambda_ij = <U_hat_i, V_j>
// at the stored (non-zero) entries comes from the responsibility kernels themselves: with
// FU = float32(U_hat), FV = float32(V) the row pass leaves s_ij = x_ij / Lambda_ij in the row-side
// slots; everything that depends on the zero entries is either a closed form of column sums or
// the MODE 1 epilogue of k_dropout_fused (dense_mfma.hip).
#include "common.h"

namespace oriana {

// F[i, k] = float32(E[row_index ? row_index[i] : i, k] * (mul ? mul[same row, k] : 1)), padded to Kp
__global__ __launch_bounds__(256) void k_factor_cast(float *__restrict__ F, const double *__restrict__ E,
                                                     const float *__restrict__ mul,
                                                     const int32_t *__restrict__ row_index, int64_t r, int K, int Kp) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= r * Kp) return;
    const int64_t i = idx / Kp;
    const int k = (int)(idx - i * Kp);
    float v = 0.f;
    if (k < K) {
        const int64_t src = row_index ? (int64_t)row_index[i] : i;
        double e = E[src * K + k];
        if (mul) e *= (double)mul[src * K + k];
        v = (float)e;
    }
    F[idx] = v;
}

__device__ __forceinline__ double block_sum(double v, double *sh) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    double t = 0.0;
    for (int i = 0; i < nw; ++i) t += sh[i];
    return t;
}

// constants of X: colsum[gene] += sum_i x_ij, colnnz[gene] += #{i: x_ij != 0} (caller's gene order),
// out[0] += sum (x log x - x), out[1] += sum x^2, over the stored entries.  One work-group per tile.
__global__ __launch_bounds__(256) void k_count_stats(oriana_counts cm, double *__restrict__ colsum,
                                                     double *__restrict__ colnnz, double *__restrict__ out) {
    __shared__ float cs[TILE];
    __shared__ int cn[TILE];
    __shared__ double sh[4];
    const int64_t t = blockIdx.x;
    const int64_t cb = t % cm.ncb;
    cs[threadIdx.x] = 0.f;
    cn[threadIdx.x] = 0;
    __syncthreads();
    const int64_t rbase = cm.roff[t];
    const uint32_t s_end = cm.rslice[t * 17 + 16];
    double a0 = 0.0, a1 = 0.0;
    for (uint32_t slot = threadIdx.x; slot < s_end; slot += 256) {
        const oriana_rowrec rec = cm.rowrec[rbase + slot];
        if (rec.x == 0.f) continue;
        const double x = (double)rec.x;
        a0 += x * log(x) - x;
        a1 += x * x;
        atomicAdd(&cs[rec.col], rec.x);          // <= 256 counts below 2^24 per column: exact in f32
        atomicAdd(&cn[rec.col], 1);
    }
    __syncthreads();
    const int64_t jp = cb * TILE + threadIdx.x;
    if (jp < cm.m && cn[threadIdx.x] != 0) {
        const int64_t j = cm.col_perm ? (int64_t)cm.col_perm[jp] : jp;
        atomicAdd(&colsum[j], (double)cs[threadIdx.x]);
        atomicAdd(&colnnz[j], (double)cn[threadIdx.x]);
    }
    a0 = block_sum(a0, sh);
    a1 = block_sum(a1, sh);
    if (threadIdx.x == 0) { atomicAdd(&out[0], a0); atomicAdd(&out[1], a1); }
}

// out[0] += sum Lambda, out[1] += sum x log Lambda, out[2] += sum Lambda^2, out[3] += sum x Lambda over the
// stored entries, Lambda = x / s (s from the row pass).  Entries the row pass could not evaluate (NaN
// sentinel: Lambda < 1e-10) are recomputed from the float64 factors.
__global__ __launch_bounds__(256) void k_metric_nnz(oriana_counts cm, const float *__restrict__ s_rs,
                                                    const double *__restrict__ U, const double *__restrict__ V,
                                                    int K, double *__restrict__ out) {
    __shared__ double sh[4];
    const int64_t t = blockIdx.x;
    const int64_t rb = t / cm.ncb, cb = t - rb * cm.ncb;
    const int64_t rbase = cm.roff[t];
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (int sl = 0; sl < 16; ++sl) {
        const uint32_t s0 = cm.rslice[t * 17 + sl], s1 = cm.rslice[t * 17 + sl + 1];
        for (uint32_t slot = s0 + threadIdx.x; slot < s1; slot += 256) {
            const oriana_rowrec rec = cm.rowrec[rbase + slot];
            if (rec.x == 0.f) continue;
            const float s = s_rs[rbase + slot];
            const double x = (double)rec.x;
            double lam;
            if (s == s && s != 0.f) {
                lam = x / (double)s;
            } else {
                const int64_t ip = rb * TILE + sl * 16 + (int)(((slot - s0) & 63u) >> 2);
                const int64_t jp = cb * TILE + rec.col;
                const int64_t i = cm.row_perm ? (int64_t)cm.row_perm[ip] : ip;
                const int64_t j = cm.col_perm ? (int64_t)cm.col_perm[jp] : jp;
                lam = 0.0;
                for (int k = 0; k < K; ++k) lam += U[i * K + k] * V[j * K + k];
            }
            a0 += lam;
            a1 += x * log(lam);
            a2 += lam * lam;
            a3 += x * lam;
        }
    }
    a0 = block_sum(a0, sh);
    a1 = block_sum(a1, sh);
    a2 = block_sum(a2, sh);
    a3 = block_sum(a3, sh);
    if (threadIdx.x == 0) {
        atomicAdd(&out[0], a0); atomicAdd(&out[1], a1); atomicAdd(&out[2], a2); atomicAdd(&out[3], a3);
    }
}

}  // namespace oriana

using namespace oriana;

extern "C" int oriana_factor_cast_f32(float *F, const double *E, const float *mul, const int32_t *row_index,
                                      int64_t r, int64_t K, void *stream) {
    const int64_t Kp = oriana_kpad(K);
    if (r < 0 || Kp == 0) return Kp == 0 ? ORIANA_EKRANGE : ORIANA_EINVAL;
    if (r == 0) return 0;
    if (!F || !E) return ORIANA_EINVAL;
    const int64_t tot = r * Kp;
    hipLaunchKernelGGL(k_factor_cast, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, F, E, mul,
                       row_index, r, (int)K, (int)Kp);
    ORIANA_LAUNCH_CHECK();
    return 0;
}

extern "C" int oriana_count_stats(const oriana_counts *cm, double *colsum, double *colnnz, double *out2,
                                  void *stream) {
    if (!cm || !colsum || !colnnz || !out2) return ORIANA_EINVAL;
    const int64_t nt = cm->nrb * cm->ncb;
    if (nt == 0 || cm->rslots == 0) return 0;
    if (nt > 0x7fffffffLL) return ORIANA_EINVAL;
    hipLaunchKernelGGL(k_count_stats, dim3((unsigned)nt), dim3(256), 0, (hipStream_t)stream, *cm, colsum, colnnz, out2);
    ORIANA_LAUNCH_CHECK();
    return 0;
}

extern "C" int oriana_metric_nnz(const oriana_counts *cm, const float *s_rs, const double *U, const double *V,
                                 int64_t K, double *out4, void *stream) {
    if (!cm || !s_rs || !U || !V || !out4 || K <= 0) return ORIANA_EINVAL;
    const int64_t nt = cm->nrb * cm->ncb;
    if (nt == 0 || cm->rslots == 0) return 0;
    if (nt > 0x7fffffffLL) return ORIANA_EINVAL;
    hipLaunchKernelGGL(k_metric_nnz, dim3((unsigned)nt), dim3(256), 0, (hipStream_t)stream, *cm, s_rs, U, V, (int)K,
                       out4);
    ORIANA_LAUNCH_CHECK();
    return 0;
}

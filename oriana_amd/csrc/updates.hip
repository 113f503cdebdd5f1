// updates.hip -- Gamma shape/rate updates, Gamma expectations and the M-step on gfx950.
//
// Replaces the (n + m) * K NumPy statements of update_variational_parameters /
// update_prior_hyper_parameters (oriana/models/gap.py:96-129 and the three twins) together with
// Gamma.mean / Gamma.meanlog (oriana/nodes/probabilistic/gamma.py:37-61) and the special functions
// of oriana/utils.py.  All of it is float64 element-wise work plus column sums: HBM-bound,
// fused so that each parameter matrix is written once and Z is read once.
#include "common.h"

namespace oriana {

// blockDim = (KT, RY): thread (tx, ty) handles columns tx, tx + KT, ... of rows ty, ty + RY, ...
// inside the block's row strip.  Column sums are carried per thread in f64, reduced over ty in
// LDS and added to the global K-vectors with one f64 atomic per column per block.
constexpr int GU_ROWS_PER_BLOCK = 512;        // upper bound; rows_per_block() picks fewer for short matrices
constexpr int GU_MAXCOLS_PER_THREAD = 4;     // K <= 4 * 128

__global__ __launch_bounds__(256) void k_gamma_update(double *__restrict__ a1, double *__restrict__ a2,
                                                      double *__restrict__ E, float *__restrict__ Elog,
                                                      double *__restrict__ colsum_E, double *__restrict__ colsum_Elog,
                                                      const double *__restrict__ prior1, const double *__restrict__ prior2,
                                                      const float *__restrict__ Z, const float *__restrict__ zmul,
                                                      const double *__restrict__ rate_vec,
                                                      const double *__restrict__ rate_mat,
                                                      const float *__restrict__ rmul, int64_t r, int K, int rpb) {
    __shared__ double red[2][256];
    const int KT = blockDim.x, RY = blockDim.y;
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int64_t r0 = (int64_t)blockIdx.x * rpb;
    const int64_t r1 = (r0 + rpb < r) ? r0 + rpb : r;
    double sE[GU_MAXCOLS_PER_THREAD], sL[GU_MAXCOLS_PER_THREAD];
    #pragma unroll
    for (int c = 0; c < GU_MAXCOLS_PER_THREAD; ++c) { sE[c] = 0.0; sL[c] = 0.0; }

    for (int64_t row = r0 + ty; row < r1; row += RY) {
        #pragma unroll
        for (int c = 0; c < GU_MAXCOLS_PER_THREAD; ++c) {
            const int k = tx + c * KT;
            if (k < K) {
                const int64_t idx = row * K + k;
                double s1, s2;
                if (Z) {
                    double z = (double)Z[idx];
                    if (zmul) z = (double)(zmul[idx] * Z[idx]);          // f32 product, as S_hat * Z_hat_j
                    s1 = clamp_eps(prior1[k] + z);
                    double rt = rate_mat ? rate_mat[idx] : rate_vec[k];
                    if (rmul) rt = (double)rmul[idx] * rt;
                    s2 = clamp_eps(prior2[k] + rt);
                    a1[idx] = s1;
                    a2[idx] = s2;
                } else {
                    s1 = a1[idx];
                    s2 = a2[idx];
                }
                const double e = s1 / s2;                                // gamma.py:37-46
                const float el = gamma_meanlog_f32(s1, s2);              // gamma.py:52-61
                E[idx] = e;
                Elog[idx] = el;
                sE[c] += e;
                sL[c] += (double)el;
            }
        }
    }
    // reduce over ty
    const int flat = ty * KT + tx;
    #pragma unroll
    for (int c = 0; c < GU_MAXCOLS_PER_THREAD; ++c) {
        const int k = tx + c * KT;
        __syncthreads();
        red[0][flat] = sE[c];
        red[1][flat] = sL[c];
        __syncthreads();
        if (ty == 0 && k < K) {
            double tE = 0.0, tL = 0.0;
            for (int y = 0; y < RY; ++y) { tE += red[0][y * KT + tx]; tL += red[1][y * KT + tx]; }
            if (colsum_E) atomicAdd(&colsum_E[k], tE);
            if (colsum_Elog) atomicAdd(&colsum_Elog[k], tL);
        }
    }
}

__global__ __launch_bounds__(256) void k_colsum_f64(double *__restrict__ out, const double *__restrict__ A,
                                                    const float *__restrict__ mul, int64_t r, int K, int rpb) {
    __shared__ double red[256];
    const int KT = blockDim.x, RY = blockDim.y;
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int64_t r0 = (int64_t)blockIdx.x * rpb;
    const int64_t r1 = (r0 + rpb < r) ? r0 + rpb : r;
    for (int c = 0; c < GU_MAXCOLS_PER_THREAD; ++c) {
        const int k = tx + c * KT;
        double s = 0.0;
        if (k < K) {
            for (int64_t row = r0 + ty; row < r1; row += RY) {
                double v = A[row * K + k];
                if (mul) v *= (double)mul[row * K + k];
                s += v;
            }
        }
        __syncthreads();
        red[ty * KT + tx] = s;
        __syncthreads();
        if (ty == 0 && k < K) {
            double t = 0.0;
            for (int y = 0; y < RY; ++y) t += red[y * KT + tx];
            atomicAdd(&out[k], t);
        }
    }
}

// gap.py:117-129: one thread per factor
__global__ void k_mstep_gamma(double *__restrict__ p1, double *__restrict__ p2, const double *__restrict__ colsum_E,
                              const double *__restrict__ colsum_Elog, double count, int K) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    // np.mean over a float32 matrix returns float32 (gap.py:120)
    const float mean_log = (float)(colsum_Elog[k] / count);
    const double y = log(p2[k]) + (double)mean_log;
    const double n1 = clamp_eps(inverse_digamma_f64(y));
    const double mean_e = colsum_E[k] / count;
    const double n2 = clamp_eps(n1 / mean_e);
    p1[k] = n1;
    p2[k] = n2;
}

enum { OP_DIGAMMA = 0, OP_TRIGAMMA, OP_INVDIGAMMA, OP_SIGMOID, OP_LOGIT };
template <int OP>
__global__ void k_map_f64(double *__restrict__ y, const double *__restrict__ x, int64_t len) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    const double v = x[i];
    double o;
    if (OP == OP_DIGAMMA) o = digamma_f64(v);
    else if (OP == OP_TRIGAMMA) o = trigamma_f64(v);
    else if (OP == OP_INVDIGAMMA) o = inverse_digamma_f64(v);
    else if (OP == OP_SIGMOID) o = sigmoid_f64(v);
    else o = logit_f64(v);
    y[i] = o;
}

template <int OP>
static int launch_map(double *y, const double *x, int64_t len, void *stream) {
    if (len < 0) return ORIANA_EINVAL;
    if (len == 0) return 0;
    if (!y || !x) return ORIANA_EINVAL;
    hipLaunchKernelGGL(k_map_f64<OP>, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y, x, len);
    ORIANA_LAUNCH_CHECK();
    return 0;
}

// Rows per block: enough blocks to fill the chip (>= ~8 per CU) even for a short matrix -- at 512 rows
// per block the gene side (30,000 rows) ran on 59 of the 256 CUs and one rank's share of the cell side
// (125,000 rows) on 245, a single 4-wave block each: 0.34 ms per call instead of ~0.1.
static inline int rows_per_block(int64_t r, int ry) {
    int64_t rpb = (r + 2047) / 2048;
    rpb = (rpb + ry - 1) / ry * ry;
    if (rpb < 4 * ry) rpb = 4 * ry;
    if (rpb > GU_ROWS_PER_BLOCK) rpb = GU_ROWS_PER_BLOCK;
    return (int)rpb;
}

static inline void pick_block(int64_t K, dim3 *block) {
    int kt = 1;
    while (kt < K && kt < 128) kt <<= 1;
    *block = dim3(kt, 256 / kt);
}

}  // namespace oriana

using namespace oriana;

extern "C" int oriana_gamma_update(double *a1, double *a2, double *E, float *Elog, double *colsum_E,
                                   double *colsum_Elog, const double *prior1, const double *prior2, const float *Z,
                                   const float *zmul, const double *rate_vec, const double *rate_mat,
                                   const float *rmul, int64_t r, int64_t K, void *stream) {
    if (r < 0 || K <= 0) return ORIANA_EINVAL;
    if (K > 128 * GU_MAXCOLS_PER_THREAD) return ORIANA_EKRANGE;
    if (r == 0) return 0;
    if (!a1 || !a2 || !E || !Elog) return ORIANA_EINVAL;
    if (Z && (!prior1 || !prior2 || (!rate_vec && !rate_mat))) return ORIANA_EINVAL;
    dim3 block;
    pick_block(K, &block);
    const int rpb = rows_per_block(r, (int)block.y);
    const int64_t nblk = (r + rpb - 1) / rpb;
    hipLaunchKernelGGL(k_gamma_update, dim3((unsigned)nblk), block, 0, (hipStream_t)stream, a1, a2, E, Elog, colsum_E,
                       colsum_Elog, prior1, prior2, Z, zmul, rate_vec, rate_mat, rmul, r, (int)K, rpb);
    ORIANA_LAUNCH_CHECK();
    return 0;
}

extern "C" int oriana_colsum_f64(double *out, const double *A, const float *mul, int64_t r, int64_t K, void *stream) {
    if (r < 0 || K <= 0) return ORIANA_EINVAL;
    if (K > 128 * GU_MAXCOLS_PER_THREAD) return ORIANA_EKRANGE;
    if (r == 0) return 0;
    if (!out || !A) return ORIANA_EINVAL;
    dim3 block;
    pick_block(K, &block);
    const int rpb = rows_per_block(r, (int)block.y);
    const int64_t nblk = (r + rpb - 1) / rpb;
    hipLaunchKernelGGL(k_colsum_f64, dim3((unsigned)nblk), block, 0, (hipStream_t)stream, out, A, mul, r, (int)K, rpb);
    ORIANA_LAUNCH_CHECK();
    return 0;
}

extern "C" int oriana_mstep_gamma(double *p1, double *p2, const double *colsum_E, const double *colsum_Elog,
                                  double count, int64_t K, void *stream) {
    if (K <= 0 || !(count > 0)) return ORIANA_EINVAL;
    if (!p1 || !p2 || !colsum_E || !colsum_Elog) return ORIANA_EINVAL;
    hipLaunchKernelGGL(k_mstep_gamma, dim3((unsigned)((K + 63) / 64)), dim3(64), 0, (hipStream_t)stream, p1, p2,
                       colsum_E, colsum_Elog, count, (int)K);
    ORIANA_LAUNCH_CHECK();
    return 0;
}

extern "C" int oriana_digamma_f64(double *y, const double *x, int64_t len, void *s) { return launch_map<OP_DIGAMMA>(y, x, len, s); }
extern "C" int oriana_trigamma_f64(double *y, const double *x, int64_t len, void *s) { return launch_map<OP_TRIGAMMA>(y, x, len, s); }
extern "C" int oriana_inverse_digamma_f64(double *y, const double *x, int64_t len, void *s) { return launch_map<OP_INVDIGAMMA>(y, x, len, s); }
extern "C" int oriana_sigmoid_f64(double *y, const double *x, int64_t len, void *s) { return launch_map<OP_SIGMOID>(y, x, len, s); }
extern "C" int oriana_logit_f64(double *y, const double *x, int64_t len, void *s) { return launch_map<OP_LOGIT>(y, x, len, s); }

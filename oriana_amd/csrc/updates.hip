// updates.hip -- Gamma shape/rate updates, Gamma expectations and the M-step on gfx950.
//
// Replaces the (n + m) * K NumPy statements of update_variational_parameters /
// update_prior_hyper_parameters (oriana/models/gap.py:96-129 and the three twins) together with
// Gamma.mean / Gamma.meanlog (oriana/nodes/probabilistic/gamma.py:37-61) and the special functions
// of oriana/utils.py.  All of it is float64 element-wise work plus column sums: HBM-bound,
// fused so that each parameter matrix is written once and Z is read once.
#include "common.h"
#include <stdlib.h>
#include <string.h>

namespace oriana {

// blockDim = (KT, RY): thread (tx, ty) handles columns tx, tx + KT, ... of rows ty, ty + RY, ...
// inside the block's row strip.  Column sums are carried per thread in f64, reduced over ty in
// LDS and added to the global K-vectors with one f64 atomic per column per block.
constexpr int GU_ROWS_PER_BLOCK = 512;        // upper bound; rows_per_block() picks fewer for short matrices
constexpr int GU_MAXCOLS_PER_THREAD = 4;     // K <= 4 * 128

// FIN: the last step of the responsibility pass (oriana_finalize: Z[o] += F[p] * R[p], o = row_index[p]) is folded in:
// the rows are walked in PACKED order p, Z is completed in place (it stays a full output) and used at once.
// NT: threads per work-group.  Every group ends with 2K float64 atomics on the SAME 2K addresses, which the memory
// side serialises (measured: 26 ns per group; a row of a thread is 0.8 us of dependent loads): a short matrix takes few,
// large groups (1024 threads, 79 of them at 10,000 rows: 17.6 -> ~10 us), a long one many small groups.
template <bool FIN, int NT>
__global__ __launch_bounds__(NT) void k_gamma_update(double *__restrict__ a1, double *__restrict__ a2,
                                                      double *__restrict__ E, float *__restrict__ Elog,
                                                      double *__restrict__ colsum_E, double *__restrict__ colsum_Elog,
                                                      const double *__restrict__ prior1, const double *__restrict__ prior2,
                                                      const float *__restrict__ Z_in, const float *__restrict__ zmul,
                                                      const double *__restrict__ rate_vec,
                                                      const double *__restrict__ rate_mat,
                                                      const float *__restrict__ rmul, int64_t r, int K, int rpb,
                                                      float *__restrict__ Zfin, const float *__restrict__ F,
                                                      const float *__restrict__ Rs, const int32_t *__restrict__ row_index,
                                                      int Kp, int nslab, int64_t slab_row0) {
    const float *Z = FIN ? Zfin : Z_in;
    __shared__ double red[2][NT];
    const int KT = blockDim.x, RY = blockDim.y;
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int64_t r0 = (int64_t)blockIdx.x * rpb;
    const int64_t r1 = (r0 + rpb < r) ? r0 + rpb : r;
    double sE[GU_MAXCOLS_PER_THREAD], sL[GU_MAXCOLS_PER_THREAD];
    #pragma unroll
    for (int c = 0; c < GU_MAXCOLS_PER_THREAD; ++c) { sE[c] = 0.0; sL[c] = 0.0; }

    for (int64_t row = r0 + ty; row < r1; row += RY) {
        const int64_t orow = (FIN && row_index) ? (int64_t)row_index[row] : row;
        #pragma unroll
        for (int c = 0; c < GU_MAXCOLS_PER_THREAD; ++c) {
            const int k = tx + c * KT;
            if (k < K) {
                const int64_t idx = orow * K + k;
                double s1, s2;
                if (FIN) {
                    float rr = Rs[row * Kp + k];
                    const int ns = row >= slab_row0 ? nslab : 1;
                    for (int sl = 1; sl < ns; ++sl) rr += Rs[((int64_t)sl * (r - slab_row0) + row) * Kp + k];
                    const float zf = fmaf(F[row * Kp + k], rr, Zfin[idx]) + 0.0f;                  // k_finalize (accumulate)
                    Zfin[idx] = zf;
                    s1 = clamp_eps(prior1[k] + (double)zf);
                    s2 = clamp_eps(prior2[k] + rate_vec[k]);
                    a1[idx] = s1;
                    a2[idx] = s2;
                } else if (Z) {
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
    // reduce over ty: inside a wave first (KT <= 32: lanes l, l + KT, ... hold the same column), then over the waves
    // through LDS -- at most NT / 64 terms per column for the thread that issues the atomics
    const int flat = ty * KT + tx, lane = flat & 63, wv = flat >> 6, NW = NT / 64;
    const int kw = KT < 64 ? KT : 64;                  // distinct columns inside a wave
    #pragma unroll
    for (int c = 0; c < GU_MAXCOLS_PER_THREAD; ++c) {
        if (c * KT >= K) break;                        // (uniform: no thread holds a column of this round)
        const int k = tx + c * KT;
        double vE = sE[c], vL = sL[c];
        for (int off = kw; off < 64; off <<= 1) { vE += __shfl_xor(vE, off); vL += __shfl_xor(vL, off); }
        __syncthreads();
        if (lane < kw) {
            // column of this lane inside the round: tx for KT <= 64; for KT = 128 a wave covers half of the columns
            red[0][wv * kw + lane] = vE;
            red[1][wv * kw + lane] = vL;
        }
        __syncthreads();
        if (ty == 0 && k < K) {
            double tE = 0.0, tL = 0.0;
            if (KT <= 64) {
                for (int w = 0; w < NW; ++w) { tE += red[0][w * kw + tx]; tL += red[1][w * kw + tx]; }
            } else {
                // KT = 128: rows of threads span two waves; wave 2y + (tx >> 6) holds columns (tx & 63) of row y
                for (int w = (tx >> 6); w < NW; w += 2) { tE += red[0][w * kw + (tx & 63)]; tL += red[1][w * kw + (tx & 63)]; }
            }
            if (colsum_E) atomicAdd(&colsum_E[k], tE);
            if (colsum_Elog) atomicAdd(&colsum_Elog[k], tL);
        }
    }
}

// ------------------------------------------------------------------------------------------
// [r5] The same update, VEC consecutive factors per lane and LPR lanes per row (a power of two >= K / VEC; 64 / LPR rows
// per wave and iteration): every access is a 4 * VEC (float32) or 8 * VEC (float64) byte piece per lane, the pieces of a row
// contiguous -- round 4's kernel gave each lane ONE element per row (512- and 288-byte wave stores at K = 100 that
// straddle the 128-byte lines: 2.3 TB/s for 4.4 GB).  Needs K % VEC == 0 (row starts stay aligned).
//
// PREP outputs (FUn != NULL), the cell-side half of the NEXT sweep's factor preparation: the row's K values of E[log .] are in
// the row's LPR lanes anyway, so the row maximum, F = exp(E[log .] - max) and the per-group partial statistics of the maxima
// (k_row_stats) cost no extra read: FUn (r, Kp) gets the row in the arithmetic of factor_prep_row (padding columns are never
// written: the buffer is zero-filled once), mu_out[r] the maximum (NaN if the row holds a NaN), upart[4 b ..] = {sum, sum of
// squares, count, min} of group b's maxima.  The validity test needs the statistics of ALL rows, i.e. the end of this
// launch: the next sweep's preparation applies it to mu_out and overwrites the rejected rows (k_factor_prep_pair, fix mode).
struct GuVecArgs {
    double *a1, *a2, *E;
    float *Elog;
    double *colsum_E, *colsum_Elog;
    const double *prior1, *prior2;
    const float *Z_in, *zmul;
    const double *rate_vec, *rate_mat;
    const float *rmul;
    int64_t r;
    int K, rpb;
    float *Zfin;
    const float *F, *Rs;
    const int32_t *row_index;
    int Kp, nslab;
    int64_t slab_row0;
    float *FUn, *mu_out, *upart;
    double *a2row;      // [r6] FIN, a2 == NULL: the K rate values every row shares (gap.py:98: alpha2 + sum_j V_hat), written once
};

template <int VEC> struct VecF;
template <> struct VecF<1> { typedef float T; };
template <> struct VecF<2> { typedef float T __attribute__((ext_vector_type(2))); };
template <> struct VecF<4> { typedef float T __attribute__((ext_vector_type(4))); };
typedef double d2 __attribute__((ext_vector_type(2)));

template <int VEC>
__device__ __forceinline__ void ld_f32(float (&o)[VEC], const float *p) {
    typedef typename VecF<VEC>::T V;
    const V v = *reinterpret_cast<const V *>(p);
    if constexpr (VEC == 1) o[0] = v;
    else { _Pragma("unroll") for (int i = 0; i < VEC; ++i) o[i] = v[i]; }
}
template <int VEC>
__device__ __forceinline__ void st_f32(float *p, const float (&o)[VEC]) {
    typedef typename VecF<VEC>::T V;
    V v;
    if constexpr (VEC == 1) v = o[0];
    else { _Pragma("unroll") for (int i = 0; i < VEC; ++i) v[i] = o[i]; }
    *reinterpret_cast<V *>(p) = v;
}
template <int VEC>
__device__ __forceinline__ void ld_f64(double (&o)[VEC], const double *p) {
    if constexpr (VEC == 1) o[0] = *p;
    else { _Pragma("unroll") for (int i = 0; i < VEC; i += 2) { const d2 v = *reinterpret_cast<const d2 *>(p + i); o[i] = v[0]; o[i + 1] = v[1]; } }
}
template <int VEC>
__device__ __forceinline__ void st_f64(double *p, const double (&o)[VEC]) {
    if constexpr (VEC == 1) *p = o[0];
    else { _Pragma("unroll") for (int i = 0; i < VEC; i += 2) { d2 v; v[0] = o[i]; v[1] = o[i + 1]; *reinterpret_cast<d2 *>(p + i) = v; } }
}

// gamma_meanlog_f32 with logf(f32(a2)) given (the pCMF cell / gene side: a2 is one value per factor)
__device__ __forceinline__ float gamma_meanlog_f32_lg(double a1, float lg) {
    const float a1f = (float)a1;
    float psi;
    if (a1f > 0.0f && a1f < INFINITY) psi = (float)digamma_pos_for_f32((double)a1f);
    else psi = (float)digamma_f64((double)a1f);
    return psi - lg;
}

template <bool FIN, int VEC, int LPR>
__global__ __launch_bounds__(256) void k_gamma_update_vec(const GuVecArgs A) {
    constexpr int RPW = 64 / LPR, NW = 4, NCOL = LPR * VEC;
    __shared__ double red[2][NW][NCOL];
    __shared__ float sred[4][NW];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int cg = lane & (LPR - 1), sr = lane / LPR;
    const int k0 = cg * VEC, K = A.K, Kp = A.Kp;
    const bool act = k0 < K;
    const int64_t r = A.r;
    const int64_t r0 = (int64_t)blockIdx.x * A.rpb;
    const int64_t r1 = (r0 + A.rpb < r) ? r0 + A.rpb : r;
    const bool prep = A.FUn != nullptr;
    const bool upd = FIN || A.Z_in != nullptr;
    double p1[VEC], p2[VEC], rv[VEC], s2c[VEC], sE[VEC], sL[VEC];
    float lg2c[VEC];
    #pragma unroll
    for (int v = 0; v < VEC; ++v) {
        p1[v] = (act && upd) ? A.prior1[k0 + v] : 1.0;
        p2[v] = (act && upd) ? A.prior2[k0 + v] : 1.0;
        rv[v] = (act && A.rate_vec) ? A.rate_vec[k0 + v] : 0.0;
        s2c[v] = clamp_eps(p2[v] + rv[v]);
        lg2c[v] = logf((float)s2c[v]);
        sE[v] = 0.0; sL[v] = 0.0;
    }
    float st_sum = 0.f, st_sq = 0.f, st_cnt = 0.f, st_min = INFINITY;
    for (int64_t rb = r0 + w * RPW; rb < r1; rb += NW * RPW) {
        const int64_t row = rb + sr;
        const bool in = act && row < r1;
        float el[VEC];
        #pragma unroll
        for (int v = 0; v < VEC; ++v) el[v] = -INFINITY;
        if (in) {
            const int64_t orow = (FIN && A.row_index) ? (int64_t)A.row_index[row] : row;
            const int64_t idx = orow * K + k0;
            double s1[VEC], s2[VEC], e[VEC];
            if (FIN) {
                float rr[VEC], f[VEC], z[VEC];
                ld_f32<VEC>(rr, A.Rs + row * Kp + k0);
                ld_f32<VEC>(f, A.F + row * Kp + k0);
                ld_f32<VEC>(z, A.Zfin + idx);
                const int ns = row >= A.slab_row0 ? A.nslab : 1;
                for (int sl = 1; sl < ns; ++sl) {
                    float r2[VEC];
                    ld_f32<VEC>(r2, A.Rs + ((int64_t)sl * (r - A.slab_row0) + row) * Kp + k0);
                    #pragma unroll
                    for (int v = 0; v < VEC; ++v) rr[v] += r2[v];
                }
                #pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    z[v] = fmaf(f[v], rr[v], z[v]) + 0.0f;                                  // k_finalize (accumulate)
                    s1[v] = clamp_eps(p1[v] + (double)z[v]);
                    s2[v] = s2c[v];
                }
                st_f32<VEC>(A.Zfin + idx, z);
                st_f64<VEC>(A.a1 + idx, s1);
                if (A.a2) st_f64<VEC>(A.a2 + idx, s2);
                #pragma unroll
                for (int v = 0; v < VEC; ++v)
                    el[v] = gamma_meanlog_f32_lg(s1[v], lg2c[v]);
            } else if (A.Z_in) {
                float z[VEC];
                ld_f32<VEC>(z, A.Z_in + idx);
                if (A.zmul) {
                    float zm[VEC];
                    ld_f32<VEC>(zm, A.zmul + idx);
                    #pragma unroll
                    for (int v = 0; v < VEC; ++v) z[v] = zm[v] * z[v];                        // f32 product, as S_hat * Z_hat_j
                }
                double rt[VEC];
                if (A.rate_mat) ld_f64<VEC>(rt, A.rate_mat + idx);
                else { _Pragma("unroll") for (int v = 0; v < VEC; ++v) rt[v] = rv[v]; }
                if (A.rmul) {
                    float rm[VEC];
                    ld_f32<VEC>(rm, A.rmul + idx);
                    #pragma unroll
                    for (int v = 0; v < VEC; ++v) rt[v] = (double)rm[v] * rt[v];
                }
                #pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    s1[v] = clamp_eps(p1[v] + (double)z[v]);
                    s2[v] = clamp_eps(p2[v] + rt[v]);
                }
                st_f64<VEC>(A.a1 + idx, s1);
                st_f64<VEC>(A.a2 + idx, s2);
                #pragma unroll
                for (int v = 0; v < VEC; ++v) el[v] = gamma_meanlog_f32(s1[v], s2[v]);
            } else {
                ld_f64<VEC>(s1, A.a1 + idx);
                ld_f64<VEC>(s2, A.a2 + idx);
                #pragma unroll
                for (int v = 0; v < VEC; ++v) el[v] = gamma_meanlog_f32(s1[v], s2[v]);
            }
            #pragma unroll
            for (int v = 0; v < VEC; ++v) {
                e[v] = s1[v] / s2[v];                                                       // gamma.py:37-46
                sE[v] += e[v];
                sL[v] += (double)el[v];
            }
            if (A.E) st_f64<VEC>(A.E + idx, e);
            st_f32<VEC>(A.Elog + idx, el);
        }
        if (prep) {
            // (wave-uniform: every lane takes part in the row's reduction; lanes beyond K or beyond the last row hold -inf)
            float mx = -INFINITY;
            bool bad = false;
            #pragma unroll
            for (int v = 0; v < VEC; ++v) { if (el[v] != el[v]) bad = true; mx = fmaxf(mx, el[v]); }
            static_assert(LPR >= 8, "lanes_max / lanes_or reduce over groups of at least 8 lanes");
            mx = lanes_max<LPR>(mx);                           // (DPP / permlane swaps: ten ds_bpermute per row pair before)
            const int badi = (int)lanes_or<LPR>(bad ? 1u : 0u);
            if (in) {
                float fu[VEC];
                #pragma unroll
                for (int v = 0; v < VEC; ++v)
                    fu[v] = (float)exp((double)el[v] - (double)mx);
                st_f32<VEC>(A.FUn + row * Kp + k0, fu);
                if (cg == 0) {
                    A.mu_out[row] = badi ? NAN : mx;
                    if (!badi) {
                        if (fabsf(mx) <= STAT_MAX) { st_sum += mx; st_sq += mx * mx; st_cnt += 1.f; }          // as k_row_stats
                        st_min = fminf(st_min, mx);
                    }
                }
            }
        }
    }
    // column sums: lanes holding the same factors inside the wave, then the four waves through LDS
    #pragma unroll
    for (int v = 0; v < VEC; ++v) {
        double vE = sE[v], vL = sL[v];
        #pragma unroll
        for (int o = LPR; o < 64; o <<= 1) { vE += __shfl_xor(vE, o, 64); vL += __shfl_xor(vL, o, 64); }
        if (sr == 0) { red[0][w][k0 + v] = vE; red[1][w][k0 + v] = vL; }
    }
    if (prep) {
        st_sum = wave_sum(st_sum); st_sq = wave_sum(st_sq); st_cnt = wave_sum(st_cnt); st_min = -wave_max(-st_min);
        if (lane == 0) { sred[0][w] = st_sum; sred[1][w] = st_sq; sred[2][w] = st_cnt; sred[3][w] = st_min; }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += 256) {
        const double tE = ((red[0][0][k] + red[0][1][k]) + red[0][2][k]) + red[0][3][k];
        const double tL = ((red[1][0][k] + red[1][1][k]) + red[1][2][k]) + red[1][3][k];
        if (A.colsum_E) atomicAdd(&A.colsum_E[k], tE);
        if (A.colsum_Elog) atomicAdd(&A.colsum_Elog[k], tL);
    }
    if (FIN && A.a2row && blockIdx.x == 0 && w == 0 && sr == 0 && act) st_f64<VEC>(A.a2row + k0, s2c);
    if (prep && threadIdx.x == 0) {
        float *pp = A.upart + 4 * (size_t)blockIdx.x;
        pp[0] = ((sred[0][0] + sred[0][1]) + sred[0][2]) + sred[0][3];
        pp[1] = ((sred[1][0] + sred[1][1]) + sred[1][2]) + sred[1][3];
        pp[2] = ((sred[2][0] + sred[2][1]) + sred[2][2]) + sred[2][3];
        pp[3] = fminf(fminf(sred[3][0], sred[3][1]), fminf(sred[3][2], sred[3][3]));
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

// both Gamma nodes in one launch (block 0: U side, block 1: V side).  keep_v: the V side's column sums were accumulated
// in colsum_v (scratch of the sweep) and are copied to keep_v (2K values: sums of E, then of Elog), where the next
// sweep's cell-side update reads sum_j V_hat.
__global__ void k_mstep_gamma_pair(double *__restrict__ p1u, double *__restrict__ p2u, const double *__restrict__ cEu,
                                   const double *__restrict__ cLu, double count_u, double *__restrict__ p1v,
                                   double *__restrict__ p2v, const double *__restrict__ cEv,
                                   const double *__restrict__ cLv, double count_v, double *__restrict__ keep_v, int K) {
    const bool vs = blockIdx.x == 1;
    double *p1 = vs ? p1v : p1u, *p2 = vs ? p2v : p2u;
    const double *cE = vs ? cEv : cEu, *cL = vs ? cLv : cLu;
    const double count = vs ? count_v : count_u;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        const double se = cE[k], sl = cL[k];
        const float mean_log = (float)(sl / count);
        const double y = log(p2[k]) + (double)mean_log;
        const double n1 = clamp_eps(inverse_digamma_f64(y));
        const double n2 = clamp_eps(n1 / (se / count));
        p1[k] = n1;
        p2[k] = n2;
        if (vs && keep_v) { keep_v[k] = se; keep_v[K + k] = sl; }
    }
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

static inline void pick_block(int64_t K, dim3 *block, int threads = 256) {
    int kt = 1;
    while (kt < K && kt < 128) kt <<= 1;
    *block = dim3(kt, threads / kt);
}

// short matrices: 1024-thread groups, four rows per thread (see k_gamma_update)
static inline bool gu_large_groups(int64_t r) {
    // (inside a configs[1] sweep: 10,000 rows 19.3 us with 1024-thread groups against 22.4 with 256; 2,000 rows 13.8 against 9.6)
    return r > 4096 && r <= 32768;
}


// ---- dispatch of k_gamma_update_vec -------------------------------------------------------------------------------
static inline bool aligned_to(const void *p, uintptr_t a) { return p == nullptr || ((uintptr_t)p & (a - 1)) == 0; }

// VEC and LPR for this K (false: the element-per-lane kernel serves it)
static inline bool gu_vec_cfg(int64_t K, bool wide_ok, int *vec, int *lpr) {
    int v;
    if (wide_ok && K % 4 == 0 && K <= 256) v = 4;
    else if (wide_ok && K % 2 == 0 && K <= 128) v = 2;
    else if (K <= 64) v = 1;
    else return false;
    int l = 8;
    while (l * v < K) l <<= 1;
    *vec = v; *lpr = l;
    return true;
}

static inline int gu_vec_rpb(int64_t r, int lpr) {
    const int ry = 4 * (64 / lpr);
    return rows_per_block(r, ry);
}

template <bool FIN, int VEC>
static void gu_vec_launch_lpr(const GuVecArgs &a, int lpr, int64_t nblk, hipStream_t s) {
    const dim3 g((unsigned)nblk), b(256);
    switch (lpr) {
    case 8: hipLaunchKernelGGL((k_gamma_update_vec<FIN, VEC, 8>), g, b, 0, s, a); break;
    case 16: hipLaunchKernelGGL((k_gamma_update_vec<FIN, VEC, 16>), g, b, 0, s, a); break;
    case 32: hipLaunchKernelGGL((k_gamma_update_vec<FIN, VEC, 32>), g, b, 0, s, a); break;
    default: hipLaunchKernelGGL((k_gamma_update_vec<FIN, VEC, 64>), g, b, 0, s, a); break;
    }
}

// true: launched (a.rpb is set here); false: no vector configuration for this K / these pointers
template <bool FIN>
static bool gu_vec_launch(GuVecArgs a, hipStream_t s) {
    const bool wide_ok = aligned_to(a.a1, 16) && aligned_to(a.a2, 16) && aligned_to(a.E, 16) && aligned_to(a.Elog, 16) &&
                         aligned_to(a.Z_in, 16) && aligned_to(a.zmul, 16) && aligned_to(a.rate_mat, 16) && aligned_to(a.rmul, 16) &&
                         aligned_to(a.Zfin, 16) && aligned_to(a.F, 16) && aligned_to(a.Rs, 16) && aligned_to(a.FUn, 16) &&
                         aligned_to(a.a2row, 16);
    int vec, lpr;
    if (!gu_vec_cfg(a.K, wide_ok, &vec, &lpr)) return false;
    if (a.FUn) {
        // The caller sized `upart` with oriana_gamma_update_prep_blocks, which assumes the aligned (wide) configuration: a
        // pointer that is not 16-byte aligned would drop to a narrower VEC, a smaller rows-per-block and MORE blocks than
        // upart has room for.  Refuse instead (the entry returns ORIANA_EINVAL).
        int v2, l2;
        if (!gu_vec_cfg(a.K, true, &v2, &l2) || v2 != vec || l2 != lpr) return false;
    }
    a.rpb = gu_vec_rpb(a.r, lpr);
    const int64_t nblk = (a.r + a.rpb - 1) / a.rpb;
    if (vec == 4) gu_vec_launch_lpr<FIN, 4>(a, lpr, nblk, s);
    else if (vec == 2) gu_vec_launch_lpr<FIN, 2>(a, lpr, nblk, s);
    else gu_vec_launch_lpr<FIN, 1>(a, lpr, nblk, s);
    return true;
}

// Short matrices are latency-bound (a sweep of configs[1] is a dozen launches of microseconds): there a thread's serial
// chain is what counts, and one element per lane (k_gamma_update, large groups) is the shorter one -- measured at 2,000 x 20:
// 10.3 us against 15.0 us, at 10,000 x 20: 15.7 against 17.4 (profiles/r05_perf_gamma*.txt).  The vector kernel takes over from
// 2^20 elements on (30,000 x 100: 46 against 60 us), and whenever its PREP outputs are asked for.
static inline bool gu_small(int64_t r, int64_t K) { return r * K < (int64_t)1 << 20; }

static inline bool gu_scalar_forced() {
    static const bool f = [] { const char *e = getenv("ORIANA_GU_KERNEL"); return e && !strcmp(e, "r4"); }();   // A/B runs
    return f;
}

}  // namespace oriana

using namespace oriana;

extern "C" int64_t oriana_gamma_update_prep_blocks(int64_t r, int64_t K) {
    int vec, lpr;
    if (r <= 0 || K <= 0 || gu_scalar_forced() || !gu_vec_cfg(K, true, &vec, &lpr)) return 0;
    const int rpb = gu_vec_rpb(r, lpr);
    return (r + rpb - 1) / rpb;
}

extern "C" int oriana_gamma_update_prep(double *a1, double *a2, double *E, float *Elog, double *colsum_E,
                                        double *colsum_Elog, const double *prior1, const double *prior2, const float *Z,
                                        const float *zmul, const double *rate_vec, const double *rate_mat,
                                        const float *rmul, int64_t r, int64_t K, float *FU_next, float *mu_out, float *upart,
                                        void *stream) {
    if (r < 0 || K <= 0) return ORIANA_EINVAL;
    if (K > 128 * GU_MAXCOLS_PER_THREAD) return ORIANA_EKRANGE;
    if (r == 0) return 0;
    if (!a1 || !a2 || !E || !Elog) return ORIANA_EINVAL;
    if (Z && (!prior1 || !prior2 || (!rate_vec && !rate_mat))) return ORIANA_EINVAL;
    if (FU_next && (!mu_out || !upart || oriana_kpad(K) == 0)) return ORIANA_EINVAL;
    if (!gu_scalar_forced() && (FU_next || !gu_small(r, K))) {
        GuVecArgs a = {a1, a2, E, Elog, colsum_E, colsum_Elog, prior1, prior2, Z, zmul, rate_vec, rate_mat, rmul, r, (int)K, 0,
                       nullptr, nullptr, nullptr, nullptr, (int)oriana_kpad(K), 1, 0, FU_next, mu_out, upart, nullptr};
        if (gu_vec_launch<false>(a, (hipStream_t)stream)) { ORIANA_LAUNCH_CHECK(); return 0; }
    }
    if (FU_next) return ORIANA_EINVAL;                 // (callers ask oriana_gamma_update_prep_blocks first)
    dim3 block;
    const bool big = gu_large_groups(r);
    pick_block(K, &block, big ? 512 : 256);          // (the general form needs more than the 128 VGPRs of a 1024-thread group)
    const int rpb = big ? 4 * (int)block.y : rows_per_block(r, (int)block.y);
    const int64_t nblk = (r + rpb - 1) / rpb;
    if (big)
        hipLaunchKernelGGL((k_gamma_update<false, 512>), dim3((unsigned)nblk), block, 0, (hipStream_t)stream, a1, a2, E, Elog,
                           colsum_E, colsum_Elog, prior1, prior2, Z, zmul, rate_vec, rate_mat, rmul, r, (int)K, rpb,
                           (float *)nullptr, (const float *)nullptr, (const float *)nullptr, (const int32_t *)nullptr, 0, 1, (int64_t)0);
    else
        hipLaunchKernelGGL((k_gamma_update<false, 256>), dim3((unsigned)nblk), block, 0, (hipStream_t)stream, a1, a2, E, Elog,
                           colsum_E, colsum_Elog, prior1, prior2, Z, zmul, rate_vec, rate_mat, rmul, r, (int)K, rpb,
                           (float *)nullptr, (const float *)nullptr, (const float *)nullptr, (const int32_t *)nullptr, 0, 1, (int64_t)0);
    ORIANA_LAUNCH_CHECK();
    return 0;
}

extern "C" int oriana_gamma_update(double *a1, double *a2, double *E, float *Elog, double *colsum_E,
                                   double *colsum_Elog, const double *prior1, const double *prior2, const float *Z,
                                   const float *zmul, const double *rate_vec, const double *rate_mat,
                                   const float *rmul, int64_t r, int64_t K, void *stream) {
    return oriana_gamma_update_prep(a1, a2, E, Elog, colsum_E, colsum_Elog, prior1, prior2, Z, zmul, rate_vec, rate_mat, rmul, r, K,
                                    nullptr, nullptr, nullptr, stream);
}

extern "C" int oriana_gamma_update_finalize_prep(double *a1, double *a2, double *E, float *Elog, double *colsum_E,
                                            double *colsum_Elog, const double *prior1, const double *prior2, float *Z,
                                            const float *F, const float *R, int64_t nslab, int64_t slab_row0, const int32_t *row_index,
                                            const double *rate_vec, int64_t r, int64_t K, float *FU_next, float *mu_out,
                                            float *upart, void *stream) {
    if (r < 0 || K <= 0 || nslab < 1 || nslab > 65535 || slab_row0 < 0) return ORIANA_EINVAL;
    const int64_t Kp = oriana_kpad(K);
    if (K > 128 * GU_MAXCOLS_PER_THREAD || Kp == 0) return ORIANA_EKRANGE;
    if (r == 0) return 0;
    if (!a1 || !a2 || !E || !Elog || !Z || !F || !R || !prior1 || !prior2 || !rate_vec) return ORIANA_EINVAL;
    if (FU_next && (!mu_out || !upart)) return ORIANA_EINVAL;
    if (!gu_scalar_forced() && (FU_next || !gu_small(r, K))) {
        GuVecArgs a = {a1, a2, E, Elog, colsum_E, colsum_Elog, prior1, prior2, nullptr, nullptr, rate_vec, nullptr, nullptr, r, (int)K, 0,
                       Z, F, R, row_index, (int)Kp, (int)nslab, slab_row0, FU_next, mu_out, upart, nullptr};
        if (gu_vec_launch<true>(a, (hipStream_t)stream)) { ORIANA_LAUNCH_CHECK(); return 0; }
    }
    if (FU_next) return ORIANA_EINVAL;
    dim3 block;
    const bool big = gu_large_groups(r);
    pick_block(K, &block, big ? 1024 : 256);
    const int rpb = big ? 4 * (int)block.y : rows_per_block(r, (int)block.y);
    const int64_t nblk = (r + rpb - 1) / rpb;
    if (big)
        hipLaunchKernelGGL((k_gamma_update<true, 1024>), dim3((unsigned)nblk), block, 0, (hipStream_t)stream, a1, a2, E, Elog,
                           colsum_E, colsum_Elog, prior1, prior2, (const float *)nullptr, (const float *)nullptr, rate_vec,
                           (const double *)nullptr, (const float *)nullptr, r, (int)K, rpb, Z, F, R, row_index, (int)Kp, (int)nslab, slab_row0);
    else
        hipLaunchKernelGGL((k_gamma_update<true, 256>), dim3((unsigned)nblk), block, 0, (hipStream_t)stream, a1, a2, E, Elog,
                           colsum_E, colsum_Elog, prior1, prior2, (const float *)nullptr, (const float *)nullptr, rate_vec,
                           (const double *)nullptr, (const float *)nullptr, r, (int)K, rpb, Z, F, R, row_index, (int)Kp, (int)nslab, slab_row0);
    ORIANA_LAUNCH_CHECK();
    return 0;
}

// [r6] pCMF's cell side without the two matrices nobody reads inside a sweep (gap.py:98, 101): a2 is the same K numbers in
// every row (alpha2 + sum_j V_hat) -- they go to a2_row[K] once -- and U_hat = a1 / a2 is not stored (its column sums still
// are summed here).  1.6 of the 4.8 GB the launch moves at configs[3].  The caller (models/gap.py) evaluates both on access.
// ORIANA_EKRANGE: no vector configuration for this K / these pointers -- the caller takes oriana_gamma_update_finalize_prep.
extern "C" int oriana_gamma_update_finalize_lazy(double *a1, double *a2_row, float *Elog, double *colsum_E, double *colsum_Elog,
                                                 const double *prior1, const double *prior2, float *Z, const float *F,
                                                 const float *R, int64_t nslab, int64_t slab_row0, const int32_t *row_index,
                                                 const double *rate_vec, int64_t r, int64_t K, float *FU_next, float *mu_out,
                                                 float *upart, void *stream) {
    if (r < 0 || K <= 0 || nslab < 1 || nslab > 65535 || slab_row0 < 0) return ORIANA_EINVAL;
    const int64_t Kp = oriana_kpad(K);
    if (K > 128 * GU_MAXCOLS_PER_THREAD || Kp == 0) return ORIANA_EKRANGE;
    if (!a1 || !a2_row || !Elog || !Z || !F || !R || !prior1 || !prior2 || !rate_vec) return ORIANA_EINVAL;
    if (FU_next && (!mu_out || !upart)) return ORIANA_EINVAL;
    if (r == 0) return ORIANA_EKRANGE;                 // (nothing would write a2_row)
    if (gu_scalar_forced() || (!FU_next && gu_small(r, K))) return ORIANA_EKRANGE;     // (launch-bound sizes: the scalar kernel)
    GuVecArgs a = {a1, nullptr, nullptr, Elog, colsum_E, colsum_Elog, prior1, prior2, nullptr, nullptr, rate_vec, nullptr, nullptr, r, (int)K, 0,
                   Z, F, R, row_index, (int)Kp, (int)nslab, slab_row0, FU_next, mu_out, upart, a2_row};
    if (!gu_vec_launch<true>(a, (hipStream_t)stream)) return ORIANA_EKRANGE;
    ORIANA_LAUNCH_CHECK();
    return 0;
}

extern "C" int oriana_gamma_update_finalize_from(double *a1, double *a2, double *E, float *Elog, double *colsum_E,
                                            double *colsum_Elog, const double *prior1, const double *prior2, float *Z,
                                            const float *F, const float *R, int64_t nslab, int64_t slab_row0, const int32_t *row_index,
                                            const double *rate_vec, int64_t r, int64_t K, void *stream) {
    return oriana_gamma_update_finalize_prep(a1, a2, E, Elog, colsum_E, colsum_Elog, prior1, prior2, Z, F, R, nslab, slab_row0, row_index,
                                             rate_vec, r, K, nullptr, nullptr, nullptr, stream);
}

extern "C" int oriana_gamma_update_finalize(double *a1, double *a2, double *E, float *Elog, double *colsum_E,
                                            double *colsum_Elog, const double *prior1, const double *prior2, float *Z,
                                            const float *F, const float *R, int64_t nslab, const int32_t *row_index,
                                            const double *rate_vec, int64_t r, int64_t K, void *stream) {
    return oriana_gamma_update_finalize_from(a1, a2, E, Elog, colsum_E, colsum_Elog, prior1, prior2, Z, F, R, nslab, 0, row_index,
                                             rate_vec, r, K, stream);
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

extern "C" int oriana_mstep_gamma_pair(double *p1u, double *p2u, const double *colsum_E_u, const double *colsum_Elog_u,
                                       double count_u, double *p1v, double *p2v, const double *colsum_E_v,
                                       const double *colsum_Elog_v, double count_v, double *keep_v, int64_t K,
                                       void *stream) {
    if (K <= 0 || !(count_u > 0) || !(count_v > 0)) return ORIANA_EINVAL;
    if (!p1u || !p2u || !colsum_E_u || !colsum_Elog_u || !p1v || !p2v || !colsum_E_v || !colsum_Elog_v) return ORIANA_EINVAL;
    hipLaunchKernelGGL(k_mstep_gamma_pair, dim3(2), dim3(K <= 64 ? 64 : 128), 0, (hipStream_t)stream, p1u, p2u, colsum_E_u,
                       colsum_Elog_u, count_u, p1v, p2v, colsum_E_v, colsum_Elog_v, count_v, keep_v, (int)K);
    ORIANA_LAUNCH_CHECK();
    return 0;
}

extern "C" int oriana_digamma_f64(double *y, const double *x, int64_t len, void *s) { return launch_map<OP_DIGAMMA>(y, x, len, s); }
extern "C" int oriana_trigamma_f64(double *y, const double *x, int64_t len, void *s) { return launch_map<OP_TRIGAMMA>(y, x, len, s); }
extern "C" int oriana_inverse_digamma_f64(double *y, const double *x, int64_t len, void *s) { return launch_map<OP_INVDIGAMMA>(y, x, len, s); }
extern "C" int oriana_sigmoid_f64(double *y, const double *x, int64_t len, void *s) { return launch_map<OP_SIGMOID>(y, x, len, s); }
extern "C" int oriana_logit_f64(double *y, const double *x, int64_t len, void *s) { return launch_map<OP_LOGIT>(y, x, len, s); }

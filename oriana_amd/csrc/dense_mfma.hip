// dense_mfma.hip -- the three dense float64 products of the zero-inflated models on the gfx950 matrix
// cores (v_mfma_f64_16x16x4_f64), reading the float32 dropout expectations D_hat in place:
//
//   oriana_dense_times_factor   trans = 0 : out[n, K] += D_hat   . W[m, K]     (zigap.py:116, D_hat V_hat)
//                               trans = 1 : out[m, K] += D_hat^T . W[n, K]     (zigap.py:124, D_hat^T U_hat)
//   oriana_dropout_update_fused p_d = sigmoid(logit(pi_d) - U_hat V_hat^T) with the overrides of
//                               zigap.py:130-136, D_hat = f32(p_d), column sums of p_d -- Lambda never
//                               touches HBM.
//
// Fragment maps of v_mfma_f64_16x16x4_f64 (one f64 per lane for A and B, four for C/D):
//   A[row = lane & 15][k = lane >> 4],  B[k = lane >> 4][col = lane & 15],
//   D reg r: [row = (lane >> 4) + 4 r][col = lane & 15].
#include "common.h"
// (the ablation switches behind the round-1 measurements -- ORIANA_ABL_NOMFMA / _NOSIG / _NOSTORE / _NOMASK -- are archived as
// a patch: tools/experiments/dense_f32_mfma_ablation_switches_r2.diff)

namespace oriana {

typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ d4 mfma_f64(double a, double b, d4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

constexpr int QC = 32;   // reduction-index chunk staged per iteration

// out[p, k] += sum_q A(p, q) W[q, k];  A(p, q) = D[p, q] (TRANS = 0, D is (P, Q) row-major) or
// D[q, p] (TRANS = 1, D is (Q, P) row-major).  W (Q, K) f64.  A work-group of 4 waves owns
// 4 * T * 16 values of p (wave w: tiles w * T ... w * T + T - 1) and one slice of the q range
// (blockIdx.y); NT tiles of 16 cover K.  W chunks go through LDS, D fragments come straight from HBM.
template <int NT, int T, int TRANS>
__global__ __launch_bounds__(256) void k_dense_times_factor(double *__restrict__ out, const float *__restrict__ D,
                                                            const double *__restrict__ W, int64_t P, int64_t Q, int K,
                                                            int64_t q_per_split, int use_atomics) {
    constexpr int WS = NT * 16 + 2;                 // LDS row stride (doubles)
    constexpr int NW = QC * NT * 16 / 256;          // W elements staged per thread and chunk
    extern __shared__ double Wsm[];                 // [2][QC][WS]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int lr = lane & 15;
    const int g = lane >> 4;
    const int64_t p_base = ((int64_t)blockIdx.x * 4 + wave) * (T * 16);
    const int64_t q_begin = (int64_t)blockIdx.y * q_per_split;
    const int64_t q_end = (q_begin + q_per_split < Q) ? q_begin + q_per_split : Q;
    const int64_t ld = TRANS ? P : Q;               // row stride of D
    const bool vec_ok = (!TRANS) && ((Q & 3) == 0);

    d4 acc[T][NT];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[t][nt] = (d4){0.0, 0.0, 0.0, 0.0};

    float dcur[T][8], dnext[T][8];
    double wreg[NW];

    auto load_d = [&](float (&d)[T][8], int64_t q0) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int64_t p = p_base + t * 16 + lr;
            const bool pok = p < P;
            if (!TRANS) {
                // step s uses q = q0 + 8 g + s: two 16-byte loads per lane, 128-byte runs per row
                const int64_t qq = q0 + 8 * g;
                const float *src = D + (pok ? p : 0) * ld + qq;
                if (vec_ok && pok && qq + 8 <= q_end) {
                    const float4 v0 = *reinterpret_cast<const float4 *>(src);
                    const float4 v1 = *reinterpret_cast<const float4 *>(src + 4);
                    d[t][0] = v0.x; d[t][1] = v0.y; d[t][2] = v0.z; d[t][3] = v0.w;
                    d[t][4] = v1.x; d[t][5] = v1.y; d[t][6] = v1.z; d[t][7] = v1.w;
                } else {
#pragma unroll
                    for (int s = 0; s < 8; ++s) d[t][s] = (pok && qq + s < q_end) ? src[s] : 0.f;
                }
            } else {
                // step s uses q = q0 + 4 s + g: 16 consecutive p per row of D
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const int64_t q = q0 + 4 * s + g;
                    d[t][s] = (pok && q < q_end) ? D[q * ld + p] : 0.f;
                }
            }
        }
    };
    auto load_w = [&](int64_t q0) {
#pragma unroll
        for (int e = 0; e < NW; ++e) {
            const int idx = e * 256 + tid;
            const int qq = idx / (NT * 16);
            const int kk = idx % (NT * 16);
            const int64_t q = q0 + qq;
            wreg[e] = (kk < K && q < q_end) ? W[q * K + kk] : 0.0;
        }
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int e = 0; e < NW; ++e) {
            const int idx = e * 256 + tid;
            Wsm[(buf * QC + idx / (NT * 16)) * WS + idx % (NT * 16)] = wreg[e];
        }
    };

    if (q_begin < q_end) {
        load_w(q_begin);
        load_d(dcur, q_begin);
        store_w(0);
        __syncthreads();
        int buf = 0;
        for (int64_t q0 = q_begin; q0 < q_end; q0 += QC) {
            const bool more = q0 + QC < q_end;
            if (more) {
                load_w(q0 + QC);
                load_d(dnext, q0 + QC);
            }
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int wrow = TRANS ? (4 * s + g) : (8 * g + s);
                double b[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) b[nt] = Wsm[(buf * QC + wrow) * WS + nt * 16 + lr];
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const double a = (double)dcur[t][s];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[t][nt] = mfma_f64(a, b[nt], acc[t][nt]);
                }
            }
            if (more) {
                store_w(buf ^ 1);
#pragma unroll
                for (int t = 0; t < T; ++t)
#pragma unroll
                    for (int s = 0; s < 8; ++s) dcur[t][s] = dnext[t][s];
            }
            __syncthreads();
            buf ^= 1;
        }
    }

#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t p = p_base + t * 16 + g + 4 * r;
                const int k = nt * 16 + lr;
                if (p < P && k < K) {
                    if (use_atomics) atomicAdd(&out[p * K + k], acc[t][nt][r]);
                    else out[p * K + k] += acc[t][nt][r];
                }
            }
}

// Fused D_q update.  Work-group = one 256 x 256 block of (cells, genes); the 64-row U_hat sub-block
// lives in LDS (A operand), V_hat fragments come from L2 four reduction steps ahead of their use
// (B operand), the reduction runs over K.  nzmask: oriana_nzmask_f32 layout, two words per lane
// cover the 64 rows of a sub-block.
// MODE 1 (metrics, base.py:58-87 with sparse_zigap.py:44-51): nothing is stored; over the entries with
// X == 0 and round(D_hat) == 1 the kernel sums log(pi_j exp(-Lambda) + 1 - pi_j) and Lambda^2 into
// colsum[0], colsum[1] (one pair of f64 atomics per wave); D_hat is read.
template <int MODE>
__global__ __launch_bounds__(256, 3) void k_dropout_fused(double *__restrict__ p_d, float *__restrict__ D_hat,
                                                       const double *__restrict__ U, const double *__restrict__ V,
                                                       const double *__restrict__ pi_d,
                                                       const uint32_t *__restrict__ nzmask,
                                                       double *__restrict__ colsum, int64_t n, int64_t m, int K,
                                                       int KS, int us) {
    extern __shared__ double Us[];                      // [64][us]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int lr = lane & 15;
    const int g = lane >> 4;
    const int64_t i_blk = (int64_t)blockIdx.y * 256;
    const int64_t j_blk = (int64_t)blockIdx.x * 256;
    const int kcols = KS * 4;

    double cs[4] = {0.0, 0.0, 0.0, 0.0};

    // operands of one (row sub-block, column tile) step that come from global memory; those of the
    // next step are requested before the current step's stores are issued, so that waiting for them
    // does not also wait for the stores (one in-order counter covers both)
    struct TileIn { double bq[4]; double pi; uint32_t w0, w1; };
    auto fetch = [&](TileIn &t, int64_t i0, int ct) {
        const int64_t j = j_blk + (wave * 4 + ct) * 16 + lr;
        const bool ok = (j < m) && (i0 < n);
        const double *vrow = V + (ok ? j : 0) * K;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = 4 * u + g;
            t.bq[u] = (ok && k < K) ? vrow[k] : 0.0;
        }
        t.pi = ok ? pi_d[j] : 0.5;
        t.w0 = 0;
        t.w1 = 0;
        if (nzmask && ok) {
            t.w0 = nzmask[(i0 >> 5) * m + j];
            if (i0 + 32 < n) t.w1 = nzmask[((i0 >> 5) + 1) * m + j];
        }
    };

    TileIn cur;
    fetch(cur, i_blk, 0);
    for (int rsub = 0; rsub < 4; ++rsub) {
        const int64_t i0 = i_blk + rsub * 64;
        if (i0 >= n) break;                             // uniform over the work-group
        __syncthreads();
        for (int rr = tid >> 4; rr < 64; rr += 16) {    // 16 threads per row of the sub-block
            const int64_t i = i0 + rr;
            for (int kk = tid & 15; kk < kcols; kk += 16)
                Us[rr * us + kk] = (i < n && kk < K) ? U[i * K + kk] : 0.0;
        }
        __syncthreads();
#pragma unroll 1
        for (int ct = 0; ct < 4; ++ct) {
            const int64_t j = j_blk + (wave * 4 + ct) * 16 + lr;
            const bool jok = j < m;
            const double *vrow = V + (jok ? j : 0) * K;
            d4 acc[4];
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) acc[rt] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
            for (int ks0 = 0; ks0 < KS; ks0 += 4) {
                double bn[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {                       // next group's fragments, in flight
                    const int k = 4 * (ks0 + 4 + u) + g;
                    bn[u] = (jok && k < K) ? vrow[k] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (ks0 + u < KS) {
#pragma unroll
                        for (int rt = 0; rt < 4; ++rt) {
                            const double a = Us[(rt * 16 + lr) * us + 4 * (ks0 + u) + g];
                            acc[rt] = mfma_f64(a, cur.bq[u], acc[rt]);
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) cur.bq[u] = bn[u];
            }
            const double pi = cur.pi;
            const uint32_t w0 = cur.w0, w1 = cur.w1;
            fetch(cur, (ct == 3) ? i0 + 64 : i0, (ct + 1) & 3);
            if (MODE == 1) {
                if (jok) {
                    int64_t off = (i0 + g) * m + j;
                    const int64_t step = 4 * m;
#pragma unroll
                    for (int rt = 0; rt < 4; ++rt) {
                        __builtin_amdgcn_sched_barrier(0);
                        const uint32_t w = (rt < 2) ? w0 : w1;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int bit = (rt & 1) * 16 + g + 4 * r;
                            const int64_t i = i0 + rt * 16 + g + 4 * r;
                            asm volatile("" : "+v"(off));
                            if (i < n && !((w >> bit) & 1u) && D_hat[off] > 0.5f) {
                                const double lam = acc[rt][r];
                                cs[0] += log(pi * exp(-lam) + (1.0 - pi));
                                cs[1] += lam * lam;
                            }
                            off += step;
                        }
                    }
                }
            } else if (jok) {
                const double lg = logit_f64(pi);
                double csum = 0.0;
                // element (rt, r) sits in row i0 + g + 4 (4 rt + r): one running offset, 4 m per step (kept
                // opaque so that the 16 addresses are not all formed -- and spilled -- ahead of the loop)
                int64_t off = (i0 + g) * m + j;
                const int64_t step = 4 * m;
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) {
                    __builtin_amdgcn_sched_barrier(0);      // one row tile at a time: keeps the f64 exp chains' registers low
                    const uint32_t w = (rt < 2) ? w0 : w1;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int bit = (rt & 1) * 16 + g + 4 * r;
                        const int64_t i = i0 + rt * 16 + g + 4 * r;
                        asm volatile("" : "+v"(off));
                        if (i < n) {
                            double p = sigmoid_f64(lg - acc[rt][r]);
                            if (pi <= 0.0) p = 1e-10;
                            if (pi >= 1.0) p = 1.0 - 1e-10;
                            if ((w >> bit) & 1u) p = 1.0 - 1e-10;
                            {
                                if (p_d) p_d[off] = p;
                                if (D_hat) D_hat[off] = (float)p;
                            }
                            csum += p;
                        }
                        off += step;
                    }
                }
                cs[ct] += csum;
            }
        }
    }
    if (MODE == 1) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            double v = cs[c];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0) atomicAdd(&colsum[c], v);
        }
    } else if (colsum) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            double v = cs[ct];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            const int64_t j = j_blk + (wave * 4 + ct) * 16 + lr;
            if (g == 0 && j < m) atomicAdd(&colsum[j], v);
        }
    }
}

template <int NT, int T, int TRANS>
static int launch_dtf(double *out, const float *D, const double *W, int64_t P, int64_t Q, int K, hipStream_t st) {
    const int64_t pb = (P + 4 * T * 16 - 1) / (4 * T * 16);
    // >= 16 slices of the reduction range (measured best for both orientations at 100k x 20k), more
    // when there are few p blocks; q slices are multiples of the chunk
    int64_t splits = (1024 + pb - 1) / pb;
    if (splits < 16) splits = 16;
    const int64_t max_splits = (Q + 4 * QC - 1) / (4 * QC);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    if (splits > 65535) splits = 65535;
    int64_t qps = (Q + splits - 1) / splits;
    qps = (qps + QC - 1) / QC * QC;
    splits = (Q + qps - 1) / qps;
    const size_t lds = (size_t)2 * QC * (NT * 16 + 2) * sizeof(double);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)k_dense_times_factor<NT, T, TRANS>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return -1000 - (int)e;
    }
    hipLaunchKernelGGL((k_dense_times_factor<NT, T, TRANS>), dim3((unsigned)pb, (unsigned)splits), dim3(256), lds, st,
                       out, D, W, P, Q, K, qps, splits > 1 ? 1 : 0);
    return 0;
}

}  // namespace oriana

using namespace oriana;

extern "C" int oriana_dense_times_factor(double *out, const float *D, const double *W, int64_t n, int64_t m, int64_t K,
                                         int trans, void *stream) {
    if (n < 0 || m < 0 || K < 0 || K > 256) return ORIANA_EINVAL;
    if (n == 0 || m == 0 || K == 0) return 0;
    if (!out || !D || !W) return ORIANA_EINVAL;
    const int64_t P = trans ? m : n, Q = trans ? n : m;
    if ((P + 63) / 64 > 0x7fffffffLL) return ORIANA_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int nt = (int)((K + 15) / 16);
    int rc = 0;
#define ORIANA_DTF(NT_, T_)                                                                      \
    do {                                                                                         \
        if (trans) rc = launch_dtf<NT_, T_, 1>(out, D, W, P, Q, (int)K, st);                      \
        else rc = launch_dtf<NT_, T_, 0>(out, D, W, P, Q, (int)K, st);                            \
    } while (0)
    // NT = ceil(K / 16) tiles of factors, T tiles of p per wave with T * NT <= 16 accumulator tiles
    switch (nt) {
        case 1: ORIANA_DTF(1, 4); break;
        case 2: ORIANA_DTF(2, 4); break;
        case 3: ORIANA_DTF(3, 4); break;
        case 4: ORIANA_DTF(4, 4); break;
        case 5: ORIANA_DTF(5, 3); break;
        case 6: ORIANA_DTF(6, 2); break;
        case 7: ORIANA_DTF(7, 2); break;
        case 8: ORIANA_DTF(8, 2); break;
        default: ORIANA_DTF(16, 1); break;
    }
#undef ORIANA_DTF
    if (rc) return rc;
    ORIANA_LAUNCH_CHECK();
    return 0;
}

template <int MODE>
static int launch_dropout_fused(double *p_d, float *D_hat, const double *U, const double *V, const double *pi_d,
                                const uint32_t *nzmask, double *colsum, int64_t n, int64_t m, int64_t K,
                                void *stream) {
    const int KS = (int)((K + 3) / 4);
    const int us = KS * 4 + 1;
    const size_t lds = (size_t)64 * us * sizeof(double);
    const int64_t ncb = (m + 255) / 256;
    const int64_t slab = 65535LL * 256;                  // row blocks in slabs of 65535 * 256 rows
    if (lds > 64 * 1024)
        ORIANA_HIP_CHECK(hipFuncSetAttribute((const void *)k_dropout_fused<MODE>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int64_t r0 = 0; r0 < n; r0 += slab) {
        const int64_t rows = (n - r0 < slab) ? n - r0 : slab;
        hipLaunchKernelGGL(k_dropout_fused<MODE>, dim3((unsigned)ncb, (unsigned)((rows + 255) / 256)), dim3(256), lds,
                           (hipStream_t)stream, p_d ? p_d + r0 * m : nullptr, D_hat ? D_hat + r0 * m : nullptr,
                           U + r0 * K, V, pi_d, nzmask ? nzmask + (r0 / 32) * m : nullptr, colsum, rows, m, (int)K, KS,
                           us);
    }
    ORIANA_LAUNCH_CHECK();
    return 0;
}

extern "C" int oriana_dropout_update_fused(double *p_d, float *D_hat, const double *U, const double *V,
                                           const double *pi_d, const uint32_t *nzmask, double *colsum, int64_t n,
                                           int64_t m, int64_t K, void *stream) {
    if (n < 0 || m < 0 || K < 0 || K > 256) return ORIANA_EINVAL;
    if (n == 0 || m == 0) return 0;
    if ((!p_d && !D_hat && !colsum) || !pi_d || (K > 0 && (!U || !V))) return ORIANA_EINVAL;
    return launch_dropout_fused<0>(p_d, D_hat, U, V, pi_d, nzmask, colsum, n, m, K, stream);
}

extern "C" int oriana_dropout_metric(double *out2, const float *D_hat, const double *U, const double *V,
                                     const double *pi_d, const uint32_t *nzmask, int64_t n, int64_t m, int64_t K,
                                     void *stream) {
    if (n < 0 || m < 0 || K < 0 || K > 256) return ORIANA_EINVAL;
    if (n == 0 || m == 0) return 0;
    if (!out2 || !D_hat || !pi_d || !nzmask || (K > 0 && (!U || !V))) return ORIANA_EINVAL;
    return launch_dropout_fused<1>(nullptr, const_cast<float *>(D_hat), U, V, pi_d, nzmask, out2, n, m, K, stream);
}

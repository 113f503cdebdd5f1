// dense_f32.hip -- the dense (n x m) work of a zero-inflated SWEEP on the float32 matrix cores
// (v_mfma_f32_32x32x2_f32, 2x the float64 rate), with every long sum carried in float64:
//
//   oriana_dropout_sweep_fused   D_hat = f32(sigmoid(logit(pi_d) - U_hat V_hat^T)) with the overrides of zigap.py:130-136,
//                                the column sums of p_d, AND the product D_hat V_next of the NEXT sweep's cell-side
//                                rates (zigap.py:116) from the tile of D_hat still in registers -- one 4 n m byte write,
//                                no read of D_hat.
//   oriana_dense_t_times_factor_f32   out[m, K] += D_hat^T W[n, K]  (zigap.py:124), D_hat streamed once.
//
// Numerics.  v_mfma_f32_32x32x2_f32 is a chain of single-rounding FMAs in k order, round-to-nearest-even
// (tools/ubench/mfma_f32.hip: bit-identical to a host fmaf() chain): a sum of T positive terms carries an unbiased
// relative error of ~ 1.5e-8 sqrt(T) (measured 1.2e-7 / 3.5e-7 / 9.5e-7 rms at T = 64 / 512 / 4096).  Lambda
// (K <= 128 terms) is used as is; the long sums (over genes / cells) leave the matrix core every 256 terms, are added
// up per work-group in float32 registers (<= 20 partial sums of independent error) and across work-groups in float64
// (atomics), so the rate terms carry ~ 1e-7, an order below the 1e-6 the float32 responsibility passes leave
// (DESIGN section 7).
// The exact float64 kernels of dense_mfma.hip stay: they evaluate p_d itself when a caller asks for it, the metrics,
// and the rate term of a sweep that cannot use the fused product (first sweep, state written from outside).
//
// Fragment map of v_mfma_f32_32x32x2_f32: A[m = lane & 31][k = lane >> 5], B[k = lane >> 5][n = lane & 31],
// D reg v: [m = 8 (v / 4) + 4 (lane >> 5) + v % 4][n = lane & 31].
#include "common.h"

namespace oriana {

typedef float f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f16v mfma32(float a, float b, f16v c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// row of the accumulator register v in lane half h
__device__ __forceinline__ int acc_row(int v, int h) { return 8 * (v >> 2) + 4 * h + (v & 3); }

constexpr int TS = 36;            // row stride of the transpose buffer (floats): 16-byte aligned rows

// LDS carve-up of k_dropout_sweep (floats)
struct SweepLds {
    int us, vt, v2, lg, mk, tb, cs, total;
    int v2s;                      // row stride of the [gene][k] image
    __host__ __device__ SweepLds(int KP2, int NT) {
        v2s = NT * 32 + 8;        // 4 rows apart = 32 banks apart: the two lane halves never collide
        int o = 0;
        us = o; o += 4 * KP2 * 32;           // [wave][k][32 cells]
        vt = o; o += 2 * KP2 * 32;           // [buf][k][32 genes]
        v2 = o; o += 2 * 32 * v2s;           // [buf][32 genes][k]
        lg = o; o += 2 * 32;                 // [buf][32 genes] logit(pi_d) with +-inf for the overrides
        mk = o; o += 2 * 4 * 32;             // [buf][wave][32 genes] non-zero bits of the wave's 32 cells
        tb = o; o += 4 * 32 * TS;            // [wave][32 cells][32 genes] transpose buffer
        cs = o; o += 2 * 4 * 32;             // [parity][wave][32 genes] column partial sums
        total = o;
    }
};

// Work-group: 4 waves = 4 strips of 32 cells, one range of genes walked in tiles of 32.  Per tile and wave:
//   Lambda^T[gene, cell] = V U^T on the matrix core (A = V tile from LDS, B = the wave's U strip from LDS) -- the
//   TRANSPOSED product, so that the accumulator registers, after the sigmoid, are already laid out as the A operand
//   of the second product  DV[cell, k] += D[cell, gene] V_next[gene, k]  (register v of lane half h = gene
//   8 (v / 4) + 4 h + v % 4: the reduction simply visits the genes in that order);
//   D_hat goes to HBM through a 32 x 32 transpose in LDS (128-byte runs per cell row); the same read-back yields
//   the tile's column sums.
template <int NT>
__global__ __launch_bounds__(256, (NT <= 2) ? 2 : 1) void k_dropout_sweep(float *__restrict__ D_hat, const double *__restrict__ U,
                                                       const double *__restrict__ V, const double *__restrict__ pi_d,
                                                       const uint32_t *__restrict__ nzmask, double *__restrict__ colsum,
                                                       const double *__restrict__ Vn, double *__restrict__ DV,
                                                       int64_t n, int64_t m, int K, int KP2, int64_t j_per_split) {
    extern __shared__ float lds[];
    const SweepLds L(KP2, NT);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 31, h = lane >> 5;
    const int64_t i0w = (int64_t)blockIdx.x * 128 + w * 32;
    const int64_t jb = (int64_t)blockIdx.y * j_per_split;
    const int64_t je = (jb + j_per_split < m) ? jb + j_per_split : m;
    const int KS = KP2 >> 1;
    float *Us = lds + L.us + w * KP2 * 32;
    uint32_t *mk = reinterpret_cast<uint32_t *>(lds + L.mk);
    float *T = lds + L.tb + w * 32 * TS;

    // the wave's strip of U_hat, [k][cell]
    for (int e = lane; e < KP2 * 32; e += 64) {
        const int cell = e / KP2, kk = e - cell * KP2;            // consecutive lanes: consecutive k of one cell
        const int64_t i = i0w + cell;
        Us[kk * 32 + cell] = (i < n && kk < K) ? (float)U[i * K + kk] : 0.f;
    }

    // staging of one gene tile, 8 threads per gene: the loads are issued a phase ahead of the LDS stores that consume
    // them (a store waits for its load: issued back to back they would expose the whole memory latency once per tile)
    const int sg = tid >> 3, sk = tid & 7;
    constexpr int NU = NT * 4;                                   // 8 * NU = NT * 32 >= KP2
    double sreg[NU];
    float lgreg = 0.f;
    uint32_t mkreg = 0;
    auto stage_load = [&](const double *__restrict__ src, int64_t j0) {
        const int64_t j = j0 + sg;
        const bool jok = j < je;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int kk = sk + 8 * u;
            sreg[u] = (jok && kk < K) ? src[j * K + kk] : 0.0;
        }
    };
    auto stage_store_vt = [&](int buf) {
        float *vt = lds + L.vt + buf * KP2 * 32;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int kk = sk + 8 * u;
            if (kk < KP2) vt[kk * 32 + sg] = (float)sreg[u];
        }
    };
    auto stage_store_v2 = [&](int buf) {
        float *v2 = lds + L.v2 + buf * 32 * L.v2s;
#pragma unroll
        for (int u = 0; u < NU; ++u) v2[sg * L.v2s + sk + 8 * u] = (float)sreg[u];
    };
    auto meta_load = [&](int64_t j0) {
        if (tid < 32) {
            const int64_t jj = j0 + tid;
            lgreg = 0.f;
            if (jj < je) {
                const double pi = pi_d[jj];
                lgreg = (pi <= 0.0) ? -INFINITY : (pi >= 1.0) ? INFINITY : (float)logit_f64(pi);
            }
        } else if (tid >= 64 && tid < 192) {
            const int ww = (tid - 64) >> 5, g = tid & 31;
            const int64_t jj = j0 + g, ir = (int64_t)blockIdx.x * 128 + ww * 32;
            mkreg = 0;
            if (nzmask && jj < je && ir < n) mkreg = nzmask[(ir >> 5) * m + jj];
        }
    };
    auto meta_store = [&](int buf) {
        if (tid < 32) lds[L.lg + buf * 32 + tid] = lgreg;
        else if (tid >= 64 && tid < 192) mk[(buf * 4 + ((tid - 64) >> 5)) * 32 + (tid & 31)] = mkreg;
    };

    f16v dv[NT], dvs[NT];                                       // matrix-core accumulators; their sums every 256 genes
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int v = 0; v < 16; ++v) { dv[nt][v] = 0.f; dvs[nt][v] = 0.f; }
    }
    const bool rowok = i0w + c < n;
    const bool vec_ok = (m & 3) == 0;

    if (jb < je) {
        stage_load(V, jb);
        meta_load(jb);
        stage_store_vt(0);
        meta_store(0);
        if (Vn) { stage_load(Vn, jb); stage_store_v2(0); }
    }
    __syncthreads();
    int buf = 0, par = 0, since_flush = 0;
    for (int64_t j0 = jb; j0 < je; j0 += 32) {
        const bool more = j0 + 32 < je;
        if (more) { stage_load(V, j0 + 32); meta_load(j0 + 32); }
        // ---- Lambda^T = V U^T
        const float *vt = lds + L.vt + buf * KP2 * 32;
        f16v l0;
#pragma unroll
        for (int v = 0; v < 16; ++v) l0[v] = 0.f;
        for (int s = 0; s < KS; ++s) l0 = mfma32(vt[(2 * s + h) * 32 + c], Us[(2 * s + h) * 32 + c], l0);
        if (more) {
            stage_store_vt(buf ^ 1);
            meta_store(buf ^ 1);
            if (Vn) stage_load(Vn, j0 + 32);
        }
        // ---- sigmoid, overrides
        const float *lgs = lds + L.lg + buf * 32;
        const uint32_t *mks = mk + (buf * 4 + w) * 32;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int g = acc_row(v, h);
            const float lgv = lgs[g];
            const uint32_t word = mks[g];
            if ((v & 3) == 0) __builtin_amdgcn_sched_barrier(0);     // four entries at a time: bounded register use
            const float x = lgv - l0[v];
            float p = __builtin_amdgcn_rcpf(1.0f + __expf(-x));
            if (lgv == -INFINITY) p = 1e-10f;                    // pi_d <= 0                         zigap.py:133
            if ((word >> c) & 1u) p = 1.0f;                      // X != 0: f32(1 - 1e-10) == 1       zigap.py:135
            if (!rowok || j0 + g >= je) p = 0.f;                 // padding never reaches a sum
            l0[v] = p;
            T[c * TS + g] = p;
        }
        __builtin_amdgcn_wave_barrier();
        // ---- D_hat rows out, column sums of the tile
        {
            const int gq = (lane & 7) * 4;
            float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = (lane >> 3) + 8 * q;
                const float4 t = *reinterpret_cast<const float4 *>(T + r * TS + gq);
                csum.x += t.x; csum.y += t.y; csum.z += t.z; csum.w += t.w;
                const int64_t i = i0w + r, j = j0 + gq;
                if (D_hat && i < n) {
                    float *dst = D_hat + i * m + j;
                    if (vec_ok && j + 4 <= je) *reinterpret_cast<float4 *>(dst) = t;
                    else {
                        if (j + 0 < je) dst[0] = t.x;
                        if (j + 1 < je) dst[1] = t.y;
                        if (j + 2 < je) dst[2] = t.z;
                        if (j + 3 < je) dst[3] = t.w;
                    }
                }
            }
            if (colsum) {
#pragma unroll
                for (int o = 8; o < 64; o <<= 1) {
                    csum.x += __shfl_xor(csum.x, o, 64); csum.y += __shfl_xor(csum.y, o, 64);
                    csum.z += __shfl_xor(csum.z, o, 64); csum.w += __shfl_xor(csum.w, o, 64);
                }
                if (lane < 8) *reinterpret_cast<float4 *>(lds + L.cs + (par * 4 + w) * 32 + gq) = csum;
            }
        }
        // ---- DV += D V_next
        if (Vn) {
            const float *v2 = lds + L.v2 + buf * 32 * L.v2s;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int g = acc_row(v, h);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) dv[nt] = mfma32(l0[v], v2[g * L.v2s + nt * 32 + c], dv[nt]);
            }
            if (++since_flush == 8) {                            // 256 genes: leave the matrix core
                since_flush = 0;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int v = 0; v < 16; ++v) { dvs[nt][v] += dv[nt][v]; dv[nt][v] = 0.f; }
            }
            if (more) stage_store_v2(buf ^ 1);
        }
        __syncthreads();
        if (colsum && w == 0 && lane < 32 && j0 + lane < je) {
            const float *cs = lds + L.cs + par * 4 * 32 + lane;
            atomicAdd(&colsum[j0 + lane], (double)cs[0] + (double)cs[32] + (double)cs[64] + (double)cs[96]);
        }
        buf ^= 1;
        par ^= 1;
    }
    if (Vn && DV) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int64_t i = i0w + acc_row(v, h);
                const int k = nt * 32 + c;
                if (i < n && k < K) atomicAdd(&DV[i * K + k], (double)dvs[nt][v] + (double)dv[nt][v]);
            }
    }
}

// out[j, k] += sum_i D[i, j] W[i, k]: a wave owns 32 * GQ genes (lane c: genes jw + GQ c ... + GQ - 1, one vector
// load per row) and NT * 32 factors; the 4 waves of a group share the rows [ib, ie) and the W chunks staged in LDS.
// A = D^T (m index = gene slot), B = W chunk; per row pair GQ * NT matrix instructions.
template <int NT, int GQ>
__global__ __launch_bounds__(256) void k_dt_times_factor_f32(double *__restrict__ out, const float *__restrict__ D,
                                                             const double *__restrict__ W, int64_t n, int64_t m, int K,
                                                             int64_t i_per_split) {
    constexpr int WS = (NT & 1) ? NT * 32 : NT * 32 + 32;       // rows one apart = 32 banks apart
    constexpr int RC = (NT > 2) ? 32 : 64;                       // rows per staged chunk
    __shared__ float Ws[2][RC * WS];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 31, h = lane >> 5;
    const int64_t jw = ((int64_t)blockIdx.x * 4 + w) * (32 * GQ);
    const int64_t ib = (int64_t)blockIdx.y * i_per_split;
    const int64_t ie = (ib + i_per_split < n) ? ib + i_per_split : n;
    const bool vec_ok = (m % GQ) == 0;
    const int64_t j = jw + GQ * c;

    f16v acc[GQ][NT], accs[GQ][NT];                             // matrix-core accumulators; their sums every 256 rows
#pragma unroll
    for (int q = 0; q < GQ; ++q)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int v = 0; v < 16; ++v) { acc[q][nt][v] = 0.f; accs[q][nt][v] = 0.f; }

    // W chunk staging, a quarter (PF row pairs = 16 rows) at a time: 16 threads per row, loads issued ahead of the
    // matrix instructions of a group and stored to LDS after them
    constexpr int PF = 8;                                        // row pairs per group, and in flight
    constexpr int NW = NT * 2;                                   // 16 * NW = NT * 32 factors
    const int sr = tid >> 4, sk = tid & 15;
    double wreg[NW];
    auto stage_load = [&](int64_t i0, int grp) {
        const int64_t i = i0 + grp * 2 * PF + sr;
        const bool iok = i < ie;
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            const int kk = sk + 16 * u;
            wreg[u] = (iok && kk < K) ? W[i * K + kk] : 0.0;
        }
    };
    auto stage_store = [&](int buf, int grp) {
        const int r = grp * 2 * PF + sr;
#pragma unroll
        for (int u = 0; u < NW; ++u) Ws[buf][r * WS + sk + 16 * u] = (float)wreg[u];
    };
    struct Frag { float d[GQ]; };
    auto load_d = [&](Frag &f, int64_t i) {
#pragma unroll
        for (int q = 0; q < GQ; ++q) f.d[q] = 0.f;
        if (i < ie && j < m) {
            const float *src = D + i * m + j;
            if (GQ == 4 && vec_ok && j + 4 <= m) {
                const float4 t = *reinterpret_cast<const float4 *>(src);
                f.d[0] = t.x; f.d[1 % GQ] = t.y; f.d[2 % GQ] = t.z; f.d[3 % GQ] = t.w;
            } else if (GQ == 2 && vec_ok && j + 2 <= m) {
                const float2 t = *reinterpret_cast<const float2 *>(src);
                f.d[0] = t.x; f.d[1 % GQ] = t.y;
            } else {
#pragma unroll
                for (int q = 0; q < GQ; ++q) if (j + q < m) f.d[q] = src[q];
            }
        }
    };

    Frag cur[PF], nxt[PF];
    constexpr int NG = RC / (2 * PF);                            // groups per chunk
    if (ib < ie) {
#pragma unroll
        for (int g = 0; g < NG; ++g) { stage_load(ib, g); stage_store(0, g); }
#pragma unroll
        for (int p = 0; p < PF; ++p) load_d(cur[p], ib + 2 * p + h);
    }
    __syncthreads();
    int buf = 0, chunks = 0;
    for (int64_t i0 = ib; i0 < ie; i0 += RC) {
        const bool more = i0 + RC < ie;
#pragma unroll 1
        for (int g = 0; g < NG; ++g) {
            const int s0 = g * PF;
#pragma unroll
            for (int p = 0; p < PF; ++p) load_d(nxt[p], i0 + 2 * (s0 + PF + p) + h);
            if (more) stage_load(i0 + RC, g);
#pragma unroll
            for (int p = 0; p < PF; ++p) {
                const float *wrow = &Ws[buf][(2 * (s0 + p) + h) * WS + c];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float b = wrow[nt * 32];
#pragma unroll
                    for (int q = 0; q < GQ; ++q) acc[q][nt] = mfma32(cur[p].d[q], b, acc[q][nt]);
                }
            }
            if (more) stage_store(buf ^ 1, g);
#pragma unroll
            for (int p = 0; p < PF; ++p) cur[p] = nxt[p];
        }
        if (++chunks == 256 / RC) {                              // 256 rows: leave the matrix core
            chunks = 0;
#pragma unroll
            for (int q = 0; q < GQ; ++q)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int v = 0; v < 16; ++v) { accs[q][nt][v] += acc[q][nt][v]; acc[q][nt][v] = 0.f; }
        }
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int q = 0; q < GQ; ++q)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int64_t jj = jw + GQ * acc_row(v, h) + q;
                const int k = nt * 32 + c;
                if (jj < m && k < K) atomicAdd(&out[jj * K + k], (double)accs[q][nt][v] + (double)acc[q][nt][v]);
            }
}

template <int NT>
static int launch_sweep(float *D_hat, const double *U, const double *V, const double *pi_d, const uint32_t *nzmask,
                        double *colsum, const double *Vn, double *DV, int64_t n, int64_t m, int K, hipStream_t st) {
    const int KP2 = (K + 1) & ~1;
    const SweepLds L(KP2, NT);
    const size_t lds = (size_t)L.total * sizeof(float);
    const int64_t rb = (n + 127) / 128;
    // gene ranges: multiples of 256 (the float64 hand-over), enough groups to fill the chip a few times over
    int64_t splits = (2048 + rb - 1) / rb;
    const int64_t max_splits = (m + 255) / 256;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    int64_t jps = (m + splits - 1) / splits;
    jps = (jps + 255) / 256 * 256;
    splits = (m + jps - 1) / jps;
    if (splits > 65535 || rb > 0x7fffffffLL) return ORIANA_EINVAL;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)k_dropout_sweep<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return -1000 - (int)e;
    }
    hipLaunchKernelGGL(k_dropout_sweep<NT>, dim3((unsigned)rb, (unsigned)splits), dim3(256), lds, st, D_hat, U, V, pi_d,
                       nzmask, colsum, Vn, DV, n, m, K, KP2, jps);
    return 0;
}

template <int NT, int GQ>
static int launch_dt(double *out, const float *D, const double *W, int64_t n, int64_t m, int K, hipStream_t st) {
    const int64_t jb = (m + 128 * GQ - 1) / (128 * GQ);
    int64_t splits = (1536 + jb - 1) / jb;
    const int64_t max_splits = (n + 511) / 512;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    int64_t ips = (n + splits - 1) / splits;
    ips = (ips + 511) / 512 * 512;
    splits = (n + ips - 1) / ips;
    if (splits > 65535 || jb > 0x7fffffffLL) return ORIANA_EINVAL;
    hipLaunchKernelGGL((k_dt_times_factor_f32<NT, GQ>), dim3((unsigned)jb, (unsigned)splits), dim3(256), 0, st, out, D, W,
                       n, m, K, ips);
    return 0;
}

}  // namespace oriana

using namespace oriana;

extern "C" int oriana_dropout_sweep_fused(float *D_hat, const double *U, const double *V, const double *pi_d,
                                          const uint32_t *nzmask, double *colsum, const double *V_next, double *DV_next,
                                          int64_t n, int64_t m, int64_t K, void *stream) {
    if (n < 0 || m < 0 || K <= 0) return ORIANA_EINVAL;
    if (K > 128) return ORIANA_EKRANGE;
    if (n == 0 || m == 0) return 0;
    if (!D_hat || !U || !V || !pi_d || ((V_next == nullptr) != (DV_next == nullptr))) return ORIANA_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    switch ((int)((K + 31) / 32)) {
        case 1: rc = launch_sweep<1>(D_hat, U, V, pi_d, nzmask, colsum, V_next, DV_next, n, m, (int)K, st); break;
        case 2: rc = launch_sweep<2>(D_hat, U, V, pi_d, nzmask, colsum, V_next, DV_next, n, m, (int)K, st); break;
        case 3: rc = launch_sweep<3>(D_hat, U, V, pi_d, nzmask, colsum, V_next, DV_next, n, m, (int)K, st); break;
        default: rc = launch_sweep<4>(D_hat, U, V, pi_d, nzmask, colsum, V_next, DV_next, n, m, (int)K, st); break;
    }
    if (rc) return rc;
    ORIANA_LAUNCH_CHECK();
    return 0;
}

extern "C" int oriana_dense_t_times_factor_f32(double *out, const float *D, const double *W, int64_t n, int64_t m,
                                               int64_t K, void *stream) {
    if (n < 0 || m < 0 || K <= 0) return ORIANA_EINVAL;
    if (K > 128) return ORIANA_EKRANGE;
    if (n == 0 || m == 0) return 0;
    if (!out || !D || !W) return ORIANA_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    switch ((int)((K + 31) / 32)) {
        case 1: rc = launch_dt<1, 4>(out, D, W, n, m, (int)K, st); break;
        case 2: rc = launch_dt<2, 2>(out, D, W, n, m, (int)K, st); break;
        case 3: rc = launch_dt<3, 1>(out, D, W, n, m, (int)K, st); break;
        default: rc = launch_dt<4, 1>(out, D, W, n, m, (int)K, st); break;
    }
    if (rc) return rc;
    ORIANA_LAUNCH_CHECK();
    return 0;
}

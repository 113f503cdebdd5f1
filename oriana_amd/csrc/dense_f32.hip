// dense_f32.hip -- the dense (n x m) work of a zero-inflated SWEEP in float32 arithmetic on the matrix cores, with
// every long sum carried on in float64.  Two evaluations of the same float32 products (`arithmetic`):
//   ORIANA_MATRIX_BF16X3  three-way bf16 splits of each operand, six cross products on v_mfma_f32_32x32x16_bf16 (K <= 64)
//   ORIANA_MATRIX_F32     v_mfma_f32_32x32x2_f32 (2x the float64 rate; K <= 128)
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
#include <type_traits>
// (the ablation switches behind the measurements of DESIGN_HISTORY.md -- ORIANA_ABL32_NOSTORE / _NOSIG / _NOTRANS -- are
// archived as a patch: tools/experiments/dense_f32_mfma_ablation_switches_r2.diff)

namespace oriana {

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f16v mfma32(float a, float b, f16v c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// Scheduling fence for vector-memory instructions only (everything else may cross): keeps a prefetch where it was
// written -- left alone, the scheduler sinks such loads down to their first use, i.e. turns them into plain loads.
#define ORIANA_VMEM_FENCE() __builtin_amdgcn_sched_barrier(0x38F)

// row of the accumulator register v in lane half h
__device__ __forceinline__ int acc_row(int v, int h) { return 8 * (v >> 2) + 4 * h + (v & 3); }

constexpr int TS = 36;            // row stride of the transpose buffer (floats): 16-byte aligned rows

// LDS carve-up of k_dropout_sweep (floats)
struct SweepLds {
    int us, vt, v2, mt, tb, cs, dm, total;
    int v2s;                      // row stride of the [gene][k] image
    __host__ __device__ SweepLds(int KP2, int NT) {
        v2s = NT * 32 + 8;        // 4 rows apart = 32 banks apart: the two lane halves never collide
        int o = 0;
        us = o; o += 4 * KP2 * 32;           // [wave][k][32 cells]
        vt = o; o += 2 * KP2 * 32;           // [buf][k][32 genes]
        v2 = o; o += 2 * 32 * v2s;           // [buf][32 genes][k]
        mt = o; o += 2 * 4 * 32 * 2;         // [buf][wave][32 genes] {logit(pi_d) (+-inf: overrides), non-zero bits
                                             //  of the wave's 32 cells}
        tb = o; o += 4 * 32 * TS;            // [wave][32 cells][32 genes] transpose buffer
        cs = o; o += 2 * 4 * 32;             // [parity][wave][32 genes] column partial sums
        dm = o; o += 32;                     // where the staging writes of k >= KP2 land (no branch per element)
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
                                                       const double *__restrict__ V, const float *__restrict__ lgit,
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
    float *T = lds + L.tb + w * 32 * TS;

    // the wave's strip of U_hat, [k][cell]
    for (int e = lane; e < KP2 * 32; e += 64) {
        const int cell = e / KP2, kk = e - cell * KP2;            // consecutive lanes: consecutive k of one cell
        const int64_t i = i0w + cell;
        Us[kk * 32 + cell] = (i < n && kk < K) ? (float)U[i * K + kk] : 0.f;
    }

    // Staging of one gene tile, 8 threads per gene.  The vector-memory counter of gfx9 retires loads, stores and
    // atomics in ONE order: a wait for a load also waits for every older store.  So the loads of tile t + 1 are issued
    // at the top of tile t and consumed (LDS stores) right BEFORE tile t's D_hat stores are issued -- what is older
    // than them by then (the stores and atomics of tile t - 1) has had a whole tile to drain.
    const int sg = tid >> 3, sk = tid & 7;
    constexpr int NU = NT * 4;                                   // 8 * NU = NT * 32 >= KP2
    const bool same_v = (Vn == V);
    double sreg[NU], nreg[NU];
    uint32_t mkreg = 0;
    // (addresses are clamped into range and the padding is zeroed when the values are consumed: straight-line
    //  loads, no register is written under a branch)
    bool sok = false;
    auto stage_load = [&](int64_t j0) {
        const int64_t j = j0 + sg;
        sok = j < je;
        const int64_t jc = sok ? j : je - 1;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int kk = sk + 8 * u;
            sreg[u] = V[jc * K + (kk < K ? kk : K - 1)];
        }
        if (Vn && !same_v) {
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int kk = sk + 8 * u;
                nreg[u] = Vn[jc * K + (kk < K ? kk : K - 1)];
            }
        }
    };
    auto stage_store = [&](int buf) {
        float *vt = lds + L.vt + buf * KP2 * 32;
        float *v2 = lds + L.v2 + buf * 32 * L.v2s;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int kk = sk + 8 * u;
            const bool ok = sok && kk < K;
            (kk < KP2 ? vt + kk * 32 : lds + L.dm)[sg] = ok ? (float)sreg[u] : 0.f;
            if (Vn) v2[sg * L.v2s + kk] = ok ? (float)(same_v ? sreg[u] : nreg[u]) : 0.f;
        }
    };
    // per gene and wave: {logit(pi_d), the mask word of the wave's 32 cells}; threads 0..127, one (wave, gene) each
    const int mw = (tid >> 5) & 3;
    const int64_t mrow = ((int64_t)blockIdx.x * 128 + mw * 32) >> 5;
    const bool mrow_ok = nzmask && (int64_t)blockIdx.x * 128 + mw * 32 < n;
    float lgreg = 0.f;
    auto meta_load = [&](int64_t j0) {
        if (tid < 128) {
            const int64_t jj = j0 + (tid & 31);
            const int64_t jc = jj < je ? jj : je - 1;
            lgreg = lgit[jc];
            mkreg = mrow_ok ? nzmask[mrow * m + jc] : 0u;
        }
    };
    auto meta_store = [&](int buf) {
        if (tid < 128) {
            float2 pr;
            pr.x = lgreg;
            pr.y = __uint_as_float(mkreg);
            *reinterpret_cast<float2 *>(lds + L.mt + ((buf * 4 + mw) * 32 + (tid & 31)) * 2) = pr;
        }
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
        stage_load(jb);
        meta_load(jb);
        stage_store(0);
        meta_store(0);
    }
    __syncthreads();
    int buf = 0, par = 0, since_flush = 0;
    // column sums of the tile before (partials of the 4 waves in LDS): atomics issued AFTER the tile's loads
    auto colsum_flush = [&](int64_t jt, int parity) {
        if (colsum && w == 0 && lane < 32 && jt + lane < je) {
            const float *cs = lds + L.cs + parity * 4 * 32 + lane;
            atomicAdd(&colsum[jt + lane], (double)cs[0] + (double)cs[32] + (double)cs[64] + (double)cs[96]);
        }
    };
    for (int64_t j0 = jb; j0 < je; j0 += 32) {
        const bool more = j0 + 32 < je;
        { const int64_t jn = more ? j0 + 32 : j0; stage_load(jn); meta_load(jn); }   // (the last tile again: unused)
        if (j0 > jb) colsum_flush(j0 - 32, par ^ 1);
        ORIANA_VMEM_FENCE();
        // ---- Lambda^T = V U^T: four steps per turn, the operands of the next turn requested before the matrix
        // instructions of this one (an LDS round trip is longer than one 16-pass instruction)
        const float *vt = lds + L.vt + buf * KP2 * 32;
        f16v l0;
#pragma unroll
        for (int v = 0; v < 16; ++v) l0[v] = 0.f;
        {
            const float *pa = vt + h * 32 + c, *pb = Us + h * 32 + c;
            float a[4], b[4], an[4], bn[4];
            int s = 0;
            if (KS >= 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) { a[u] = pa[64 * u]; b[u] = pb[64 * u]; }
                for (; s + 8 <= KS; s += 4) {
                    pa += 256; pb += 256;
#pragma unroll
                    for (int u = 0; u < 4; ++u) { an[u] = pa[64 * u]; bn[u] = pb[64 * u]; }
#pragma unroll
                    for (int u = 0; u < 4; ++u) l0 = mfma32(a[u], b[u], l0);
#pragma unroll
                    for (int u = 0; u < 4; ++u) { a[u] = an[u]; b[u] = bn[u]; }
                }
                pa += 256; pb += 256;
                // the remainder's operands, then the last full turn
                const int rem = KS - s - 4;
#pragma unroll
                for (int u = 0; u < 3; ++u) { const int o = (u < rem) ? 64 * u : 0; an[u] = pa[o]; bn[u] = pb[o]; }
#pragma unroll
                for (int u = 0; u < 4; ++u) l0 = mfma32(a[u], b[u], l0);
#pragma unroll
                for (int u = 0; u < 3; ++u) if (u < rem) l0 = mfma32(an[u], bn[u], l0);
            } else {
                for (; s < KS; ++s) l0 = mfma32(pa[64 * s], pb[64 * s], l0);
            }
        }
        // ---- sigmoid, overrides
        const int jrem = (je - j0 < 32) ? (int)(je - j0) : 32;
        const float2 *mts = reinterpret_cast<const float2 *>(lds + L.mt) + (buf * 4 + w) * 32;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const float2 mt = mts[acc_row(v, h)];
            if ((v & 3) == 0) __builtin_amdgcn_sched_barrier(0);     // four entries at a time: bounded register use
            const float x = mt.x - l0[v];
            float p = __builtin_amdgcn_rcpf(1.0f + __expf(-x));
            if (mt.x == -INFINITY) p = 1e-10f;                   // pi_d <= 0                         zigap.py:133
            if ((__float_as_uint(mt.y) >> c) & 1u) p = 1.0f;     // X != 0: f32(1 - 1e-10) == 1       zigap.py:135
            l0[v] = p;
        }
        if (!(jrem == 32 && i0w + 32 <= n)) {                    // (uniform) padding never reaches a sum
#pragma unroll
            for (int v = 0; v < 16; ++v) if (!rowok || acc_row(v, h) >= jrem) l0[v] = 0.f;
        }
#pragma unroll
        for (int v = 0; v < 16; ++v) T[c * TS + acc_row(v, h)] = l0[v];
        __builtin_amdgcn_wave_barrier();
        stage_store(buf ^ 1);
        meta_store(buf ^ 1);
        ORIANA_VMEM_FENCE();
        // ---- D_hat rows out, column sums of the tile
        {
            const int gq = (lane & 7) * 4;
            const bool full = vec_ok && jrem == 32;              // uniform: whole 16-byte pieces
            f4v csum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = (lane >> 3) + 8 * q;
                const f4v t = *reinterpret_cast<const f4v *>(T + r * TS + gq);
                csum += t;
                const int64_t i = i0w + r;
                if (D_hat && i < n) {
                    float *dst = D_hat + i * m + j0 + gq;
                    if (full) *reinterpret_cast<f4v *>(dst) = t;
                    else {
                        if (gq + 0 < jrem) dst[0] = t.x;
                        if (gq + 1 < jrem) dst[1] = t.y;
                        if (gq + 2 < jrem) dst[2] = t.z;
                        if (gq + 3 < jrem) dst[3] = t.w;
                    }
                }
            }
            if (colsum) {
                csum.x = sum_mod8(csum.x); csum.y = sum_mod8(csum.y); csum.z = sum_mod8(csum.z); csum.w = sum_mod8(csum.w);
                if (lane < 8) *reinterpret_cast<f4v *>(lds + L.cs + (par * 4 + w) * 32 + gq) = csum;
            }
        }
        // ---- DV += D V_next
        if (Vn) {
            const float *v2 = lds + L.v2 + buf * 32 * L.v2s + c;
            float bc[NT], bn[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bc[nt] = v2[acc_row(0, h) * L.v2s + nt * 32];
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int gn = acc_row(v < 15 ? v + 1 : 15, h);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bn[nt] = v2[gn * L.v2s + nt * 32];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) dv[nt] = mfma32(l0[v], bc[nt], dv[nt]);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bc[nt] = bn[nt];
                __builtin_amdgcn_sched_barrier(0);
            }
            if (++since_flush == 8) {                            // 256 genes: leave the matrix core
                since_flush = 0;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int v = 0; v < 16; ++v) { dvs[nt][v] += dv[nt][v]; dv[nt][v] = 0.f; }
            }
        }
        __syncthreads();
        buf ^= 1;
        par ^= 1;
    }
    if (jb < je) colsum_flush(jb + ((je - jb - 1) / 32) * 32, par ^ 1);
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
__global__ __launch_bounds__(256, (NT < 4) ? 2 : 1) void k_dt_times_factor_f32(double *__restrict__ out, const float *__restrict__ D,
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
    // (clamped addresses, padding zeroed when the values are stored: straight-line loads that the compiler leaves where
    //  they are issued -- a load under a branch drags its conversion, and with it the wait, up to the load)
    bool wok = false;
    auto stage_load = [&](int64_t i0, int grp) {
        const int64_t i = i0 + grp * 2 * PF + sr;
        wok = i < ie;
        const double *src = W + (wok ? i : ie - 1) * K;
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            const int kk = sk + 16 * u;
            wreg[u] = src[kk < K ? kk : K - 1];
        }
    };
    auto stage_store = [&](int buf, int grp) {
        const int r = grp * 2 * PF + sr;
#pragma unroll
        for (int u = 0; u < NW; ++u)                             // (a product, not a select: the load stays unconditional)
            Ws[buf][r * WS + sk + 16 * u] = (float)wreg[u] * ((wok && sk + 16 * u < K) ? 1.f : 0.f);
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

    // ring of RING row pairs: the slot a matrix instruction has just consumed is refilled with the pair RING steps ahead
    constexpr int RING = 2 * PF;
    constexpr int NG = RC / (2 * PF);                            // staging groups (PF row pairs) per chunk
    static_assert(RC / 2 % RING == 0, "a chunk is a whole number of ring turns");
    Frag ring[RING];
    if (ib < ie) {
#pragma unroll
        for (int g = 0; g < NG; ++g) { stage_load(ib, g); stage_store(0, g); }
#pragma unroll
        for (int p = 0; p < RING; ++p) load_d(ring[p], ib + 2 * p + h);
    }
    __syncthreads();
    int buf = 0, chunks = 0;
    // interior of the matrix (every row the ring will ask for exists, the wave's genes are whole vectors): one
    // vector load from a running pointer per row pair, nothing else
    typedef float dvec __attribute__((ext_vector_type(GQ == 1 ? 2 : GQ)));   // (GQ == 1 takes the scalar branch)
    const bool wave_fast = vec_ok && jw + 32 * GQ <= m;
    // One step = one row pair: the B operands of the NEXT step are requested, the matrix instructions of this step
    // issued, the ring slot refilled -- and a full scheduling fence: the order below is the schedule (left to itself
    // the scheduler sinks the prefetches down to their uses, i.e. exposes every latency they were written to hide).
    auto chunk = [&](auto fast_tag, int64_t i0, bool more) {
        constexpr bool FAST = decltype(fast_tag)::value;
        const float *dp = D + (i0 + 2 * RING + h) * m + j;       // (dereferenced on the fast path only)
        const int64_t dstep = 2 * m;
        float bc[NT], bn[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bc[nt] = Ws[buf][h * WS + c + nt * 32];
#pragma unroll 1
        for (int t = 0; t < RC / 2 / RING; ++t) {
#pragma unroll
            for (int gg = 0; gg < 2; ++gg) {
                const int g = 2 * t + gg, s0 = g * PF;
                stage_load(more ? i0 + RC : i0, g);              // (unconditional; the last chunk is staged again, unread)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int p = 0; p < PF; ++p) {
                    const int sn = (s0 + p + 1 < RC / 2) ? s0 + p + 1 : RC / 2 - 1;
                    const float *wrow = &Ws[buf][(2 * sn + h) * WS + c];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) bn[nt] = wrow[nt * 32];
                    __builtin_amdgcn_sched_barrier(0);
                    Frag &f = ring[gg * PF + p];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int q = 0; q < GQ; ++q) acc[q][nt] = mfma32(f.d[q], bc[nt], acc[q][nt]);
                    if (FAST) {
                        if (GQ == 1) f.d[0] = *dp;
                        else {
                            const dvec tv = *reinterpret_cast<const dvec *>(dp);
#pragma unroll
                            for (int q = 0; q < GQ; ++q) f.d[q] = tv[q];
                        }
                        dp += dstep;
                    } else {
                        load_d(f, i0 + 2 * (s0 + p + RING) + h);
                    }
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) bc[nt] = bn[nt];
                    __builtin_amdgcn_sched_barrier(0);
                }
                stage_store(buf ^ 1, g);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    for (int64_t i0 = ib; i0 < ie; i0 += RC) {
        const bool more = i0 + RC < ie;
        if (wave_fast && i0 + RC + 2 * RING <= ie) chunk(std::true_type{}, i0, more);
        else chunk(std::false_type{}, i0, more);
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

// ------------------------------------------------------------------------------------------------------------------
// The same sweep kernel with every float32 product evaluated on the bf16 matrix cores: x = hi + mid + lo (three bf16,
// 24 bits, split by truncation so that the remainders are exact) and six of the nine cross products
// (hi hi, hi mid, mid hi, mid mid, hi lo, lo hi; the three dropped ones are below 2^-24 of the product), accumulated in
// float32 by v_mfma_f32_32x32x16_bf16.  tools/ubench/mfma_bf16x3.hip: the error of a sum of <= 512 positive terms is
// that of the float32 FMA chain (2.9e-7 rms at 512), at 2.7 x the rate of v_mfma_f32_32x32x2_f32 after the six-fold
// overhead.  K <= 64.  Fragment map of v_mfma_f32_32x32x16_bf16: A[m = lane & 31][k = 8 (lane >> 5) + e],
// B[k = 8 (lane >> 5) + e][n = lane & 31], e = 0..7 the eight bf16 (four dwords) of the lane; D as for the float32
// 32 x 32 instructions.
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef uint32_t u4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f16v mfma_b16(u4v a, u4v b, f16v c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
}

// two floats -> their (hi, mid, lo) bf16 parts, each pair packed in one dword (x0 in the low half)
__device__ __forceinline__ void split2(float x0, float x1, uint32_t &hi, uint32_t &mid, uint32_t &lo) {
    const uint32_t b0 = __float_as_uint(x0), b1 = __float_as_uint(x1);
    const float r0 = x0 - __uint_as_float(b0 & 0xFFFF0000u), r1 = x1 - __uint_as_float(b1 & 0xFFFF0000u);
    const uint32_t c0 = __float_as_uint(r0), c1 = __float_as_uint(r1);
    const float s0 = r0 - __uint_as_float(c0 & 0xFFFF0000u), s1 = r1 - __uint_as_float(c1 & 0xFFFF0000u);
    hi = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
    mid = __builtin_amdgcn_perm(c1, c0, 0x07060302u);
    lo = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
}

// eight floats -> three operand registers (hi, mid, lo)
__device__ __forceinline__ void split8(const float (&x)[8], u4v (&o)[3]) {
#pragma unroll
    for (int w2 = 0; w2 < 4; ++w2) {
        uint32_t a, b, c2;
        split2(x[2 * w2], x[2 * w2 + 1], a, b, c2);
        o[0][w2] = a; o[1][w2] = b; o[2][w2] = c2;
    }
}

// the six cross products, small terms first
#define ORIANA_MF6(ACC, A, B)                                                                          \
    do {                                                                                               \
        ACC = mfma_b16(A[2], B[0], ACC); ACC = mfma_b16(A[0], B[2], ACC); ACC = mfma_b16(A[1], B[1], ACC); \
        ACC = mfma_b16(A[1], B[0], ACC); ACC = mfma_b16(A[0], B[1], ACC); ACC = mfma_b16(A[0], B[0], ACC); \
    } while (0)

// LDS carve-up of k_dropout_sweep_b16 (bytes): operand images are [..][split][lane] x 16 bytes
struct B16Lds {
    int vt, v2, mt, tb, cs, total;
    __host__ __device__ B16Lds(int KC, int NT) {
        int o = 0;
        vt = o; o += 2 * KC * 3 * 64 * 16;           // [buf][k chunk][split][lane]: A operand of Lambda^T = V U^T
        v2 = o; o += 2 * NT * 2 * 3 * 64 * 16;       // [buf][n tile][instruction][split][lane]: B operand of D V_next
        mt = o; o += 2 * 4 * 32 * 2 * 4;             // [buf][wave][32 genes] {logit(pi_d), mask word}
        tb = o; o += 4 * 32 * TS * 4;                // [wave][32 cells][32 genes] transpose buffer
        cs = o; o += 2 * 4 * 32 * 4;                 // [parity][wave][32 genes] column partial sums
        total = o;
    }
};

// The operand images of a gene tile are the same for every row block: they are built ONCE per sweep, in the layout
// the kernel copies into LDS, by this pre-kernel (one work-group per tile of 32 genes):
//   first image  [k chunk][split][lane = 32 (G & 1) + g]: V[g][8 G .. 8 G + 7], thread (g = tid & 31, G = tid >> 5);
//   second image [n tile][instruction q][split][lane = 32 hh + cc]: V_next[gene(q, hh, e)][32 nt + cc], e = 0..7, the
//                genes in the order the accumulator registers of the first product hold them (acc_row).
template <int NT, int KC>
__global__ __launch_bounds__(256) void k_split_images(u4v *__restrict__ img1, u4v *__restrict__ img2,
                                                      const double *__restrict__ V, const double *__restrict__ Vn,
                                                      int64_t m, int K) {
    const int tid = threadIdx.x;
    const int64_t j0 = (int64_t)blockIdx.x * 32;
    {
        const int g = tid & 31, G = tid >> 5;
        if (G < 2 * KC) {
            const int64_t j = j0 + g;
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int kk = 8 * G + e;
                x[e] = (j < m && kk < K) ? (float)V[j * K + kk] : 0.f;
            }
            u4v o[3];
            split8(x, o);
            u4v *dst = img1 + (int64_t)blockIdx.x * (KC * 3 * 64) + ((G >> 1) * 3) * 64 + (G & 1) * 32 + g;
#pragma unroll
            for (int sp = 0; sp < 3; ++sp) dst[sp * 64] = o[sp];
        }
    }
    if (Vn) {
        const int nt = tid >> 7, q = (tid >> 6) & 1, hh = (tid >> 5) & 1, cc = tid & 31;
        if (nt < NT) {
            const int kk = nt * 32 + cc;
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int64_t j = j0 + acc_row(8 * q + e, hh);
                x[e] = (j < m && kk < K) ? (float)Vn[j * K + kk] : 0.f;
            }
            u4v o[3];
            split8(x, o);
            u4v *dst = img2 + (int64_t)blockIdx.x * (NT * 2 * 3 * 64) + ((nt * 2 + q) * 3) * 64 + hh * 32 + cc;
#pragma unroll
            for (int sp = 0; sp < 3; ++sp) dst[sp * 64] = o[sp];
        }
    }
}

template <int NT, int KC>
__global__ __launch_bounds__(256, 2) void k_dropout_sweep_b16(float *__restrict__ D_hat, const double *__restrict__ U,
                                                              const u4v *__restrict__ img1, const float *__restrict__ lgit,
                                                              const uint32_t *__restrict__ nzmask,
                                                              double *__restrict__ colsum, const u4v *__restrict__ img2,
                                                              double *__restrict__ DV, int64_t n, int64_t m, int K,
                                                              int64_t j_per_split) {
    extern __shared__ float lds[];
    char *ldsb = reinterpret_cast<char *>(lds);
    const B16Lds L(KC, NT);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 31, h = lane >> 5;
    const int64_t i0w = (int64_t)blockIdx.x * 128 + w * 32;
    const int64_t jb = (int64_t)blockIdx.y * j_per_split;
    const int64_t je = (jb + j_per_split < m) ? jb + j_per_split : m;
    float *T = reinterpret_cast<float *>(ldsb + L.tb) + w * 32 * TS;
    u4v *vt_img = reinterpret_cast<u4v *>(ldsb + L.vt);
    u4v *v2_img = reinterpret_cast<u4v *>(ldsb + L.v2);

    // the wave's strip of U_hat as the B operand of the first product: per k chunk, factors 16 kc + 8 h + e of cell c
    u4v ub[KC][3];
    {
        const int64_t i = i0w + c;
        const double *urow = U + (i < n ? i : n - 1) * K;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int kk = 16 * kc + 8 * h + e;
                x[e] = (i < n && kk < K) ? (float)urow[kk < K ? kk : K - 1] : 0.f;
            }
            split8(x, ub[kc]);
        }
    }

    // Staging of one gene tile: straight 16-byte copies of the two operand images k_split_images has built (loads at
    // the top of the previous tile, consumed right before that tile's stores: see k_dropout_sweep).
    constexpr int N1 = KC * 3 * 64, N2 = NT * 2 * 3 * 64;        // 16-byte pieces per image
    constexpr int R1 = (N1 + 255) / 256, R2 = (N2 + 255) / 256;
    const bool has_next = img2 != nullptr;
    u4v sreg[R1], nreg[R2];
    uint32_t mkreg = 0;
    auto stage_load = [&](int64_t j0) {
        const int64_t tile = j0 >> 5;
        const u4v *s1 = img1 + tile * N1;
#pragma unroll
        for (int r = 0; r < R1; ++r) { const int idx = tid + 256 * r; sreg[r] = s1[idx < N1 ? idx : N1 - 1]; }
        if (has_next) {
            const u4v *s2 = img2 + tile * N2;
#pragma unroll
            for (int r = 0; r < R2; ++r) { const int idx = tid + 256 * r; nreg[r] = s2[idx < N2 ? idx : N2 - 1]; }
        }
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int r = 0; r < R1; ++r) { const int idx = tid + 256 * r; if (N1 % 256 == 0 || idx < N1) vt_img[buf * N1 + idx] = sreg[r]; }
        if (has_next) {
#pragma unroll
            for (int r = 0; r < R2; ++r) { const int idx = tid + 256 * r; if (N2 % 256 == 0 || idx < N2) v2_img[buf * N2 + idx] = nreg[r]; }
        }
    };
    const int mw = (tid >> 5) & 3;
    const int64_t mrow = ((int64_t)blockIdx.x * 128 + mw * 32) >> 5;
    const bool mrow_ok = nzmask && (int64_t)blockIdx.x * 128 + mw * 32 < n;
    float lgreg = 0.f;
    auto meta_load = [&](int64_t j0) {
        if (tid < 128) {
            const int64_t jj = j0 + (tid & 31);
            const int64_t jc = jj < je ? jj : je - 1;
            lgreg = lgit[jc];
            mkreg = mrow_ok ? nzmask[mrow * m + jc] : 0u;
        }
    };
    auto meta_store = [&](int buf) {
        if (tid < 128) {
            float2 pr;
            pr.x = lgreg;
            pr.y = __uint_as_float(mkreg);
            *reinterpret_cast<float2 *>(ldsb + L.mt + (((buf * 4 + mw) * 32 + (tid & 31)) * 2) * 4) = pr;
        }
    };

    f16v dv[NT], dvs[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int v = 0; v < 16; ++v) { dv[nt][v] = 0.f; dvs[nt][v] = 0.f; }
    }
    const bool rowok = i0w + c < n;
    const bool vec_ok = (m & 3) == 0;
    float *csb = reinterpret_cast<float *>(ldsb + L.cs);

    if (jb < je) {
        stage_load(jb);
        meta_load(jb);
        stage_store(0);
        meta_store(0);
    }
    __syncthreads();
    int buf = 0, par = 0, since_flush = 0;
    auto colsum_flush = [&](int64_t jt, int parity) {
        if (colsum && w == 0 && lane < 32 && jt + lane < je) {
            const float *cs = csb + parity * 4 * 32 + lane;
            atomicAdd(&colsum[jt + lane], (double)cs[0] + (double)cs[32] + (double)cs[64] + (double)cs[96]);
        }
    };
    for (int64_t j0 = jb; j0 < je; j0 += 32) {
        const bool more = j0 + 32 < je;
        { const int64_t jn = more ? j0 + 32 : j0; stage_load(jn); meta_load(jn); }   // (the last tile again: unused)
        if (j0 > jb) colsum_flush(j0 - 32, par ^ 1);
        ORIANA_VMEM_FENCE();
        // ---- Lambda^T = V U^T
        f16v l0;
#pragma unroll
        for (int v = 0; v < 16; ++v) l0[v] = 0.f;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            const u4v *src = vt_img + buf * N1 + (kc * 3) * 64 + lane;
            u4v a[3];
#pragma unroll
            for (int sp = 0; sp < 3; ++sp) a[sp] = src[sp * 64];
            ORIANA_MF6(l0, a, ub[kc]);
        }
        // ---- sigmoid, overrides
        const int jrem = (je - j0 < 32) ? (int)(je - j0) : 32;
        const float2 *mts = reinterpret_cast<const float2 *>(ldsb + L.mt) + (buf * 4 + w) * 32;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const float2 mt = mts[acc_row(v, h)];
            if ((v & 3) == 0) __builtin_amdgcn_sched_barrier(0);
            const float x = mt.x - l0[v];
            float p = __builtin_amdgcn_rcpf(1.0f + __expf(-x));
            if (mt.x == -INFINITY) p = 1e-10f;                   // pi_d <= 0                         zigap.py:133
            if ((__float_as_uint(mt.y) >> c) & 1u) p = 1.0f;     // X != 0: f32(1 - 1e-10) == 1       zigap.py:135
            l0[v] = p;
        }
        if (!(jrem == 32 && i0w + 32 <= n)) {                    // (uniform) padding never reaches a sum
#pragma unroll
            for (int v = 0; v < 16; ++v) if (!rowok || acc_row(v, h) >= jrem) l0[v] = 0.f;
        }
#pragma unroll
        for (int v = 0; v < 16; ++v) T[c * TS + acc_row(v, h)] = l0[v];
        __builtin_amdgcn_wave_barrier();
        stage_store(buf ^ 1);
        meta_store(buf ^ 1);
        ORIANA_VMEM_FENCE();
        // ---- D_hat rows out, column sums of the tile
        {
            const int gq = (lane & 7) * 4;
            const bool full = vec_ok && jrem == 32;
            f4v csum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = (lane >> 3) + 8 * q;
                const f4v t = *reinterpret_cast<const f4v *>(T + r * TS + gq);
                csum += t;
                const int64_t i = i0w + r;
                if (D_hat && i < n) {
                    float *dst = D_hat + i * m + j0 + gq;
                    if (full) *reinterpret_cast<f4v *>(dst) = t;
                    else {
                        if (gq + 0 < jrem) dst[0] = t.x;
                        if (gq + 1 < jrem) dst[1] = t.y;
                        if (gq + 2 < jrem) dst[2] = t.z;
                        if (gq + 3 < jrem) dst[3] = t.w;
                    }
                }
            }
            if (colsum) {
                csum.x = sum_mod8(csum.x); csum.y = sum_mod8(csum.y); csum.z = sum_mod8(csum.z); csum.w = sum_mod8(csum.w);
                if (lane < 8) *reinterpret_cast<f4v *>(csb + (par * 4 + w) * 32 + gq) = csum;
            }
        }
        // ---- DV += D V_next: the accumulator registers 8 q .. 8 q + 7 are the eight k slots of instruction q
        if (has_next) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float x[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = l0[8 * q + e];
                u4v a[3];
                split8(x, a);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const u4v *src = v2_img + buf * N2 + ((nt * 2 + q) * 3) * 64 + lane;
                    u4v b[3];
#pragma unroll
                    for (int sp = 0; sp < 3; ++sp) b[sp] = src[sp * 64];
                    ORIANA_MF6(dv[nt], a, b);
                }
            }
            if (++since_flush == 8) {                            // 256 genes: leave the matrix core
                since_flush = 0;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int v = 0; v < 16; ++v) { dvs[nt][v] += dv[nt][v]; dv[nt][v] = 0.f; }
            }
        }
        __syncthreads();
        buf ^= 1;
        par ^= 1;
    }
    if (jb < je) colsum_flush(jb + ((je - jb - 1) / 32) * 32, par ^ 1);
    if (has_next && DV) {
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

// Number of ranges to cut the reduction (or gene) axis into, given `blocks` work-groups along the other axis: the
// grid runs in rounds of 512 resident work-groups (2 per CU), and a last round that is nearly empty costs as much as a
// full one -- 1580 groups take 4 rounds at 77 %.  Among 4 to 12 rounds' worth, the count that leaves the last round
// fullest (fewer ranges on ties: each one ends in a set of float64 atomics).
static int64_t pick_splits(int64_t blocks, int64_t max_splits) {
    const int64_t slots = 2 * oriana_device_cus();            // (512 on MI355X)
    int64_t best = 1;
    double best_eff = 0.0;
    for (int64_t sp = 1; sp <= max_splits && sp <= 4096; ++sp) {
        const int64_t groups = blocks * sp;
        if (groups > 12 * slots && sp > 1) break;
        const int64_t rounds = (groups + slots - 1) / slots;
        double eff = (double)groups / (double)(rounds * slots);
        if (groups < 4 * slots) eff *= 0.5 + 0.125 * (double)groups / (double)slots;   // too few to hide the tails
        if (eff > best_eff + 0.02) { best_eff = eff; best = sp; }
    }
    return best;
}

// logit(pi_d) in float32, with the overrides of zigap.py:133-134 encoded as -inf (pi_d <= 0: p_d = 1e-10) and +inf
// (pi_d >= 1: p_d = 1 - 1e-10, 1 in float32)
// [r5] lgs (may be NULL): the same as -logit * log2(e) -- the form dn::k_zi_row takes: its sigmoid is
// 1 / (1 + exp2(fma(Lambda, log2 e, lgs))), one fused multiply-add where the subtraction and the scaling were two instructions
// per entry (floating-point vector work beside matrix instructions is the dear kind on this part, DESIGN_HISTORY.md 10 i)
// [r6] flo (with lgs): the floor dn::k_zi_row ADDS to its sigmoid -- 1e-10 where pi_d <= 0 (there lgs = +inf makes the sigmoid
// +0: the column override of zigap.py:133 as one addition), 0 elsewhere.
__global__ void k_logit_f32(float *__restrict__ lg, float *__restrict__ lgs, float *__restrict__ flo,
                            const double *__restrict__ pi_d, int64_t m) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const double pi = pi_d[j];
    const float v = (pi <= 0.0) ? -INFINITY : (pi >= 1.0) ? INFINITY : (float)logit_f64(pi);
    lg[j] = v;
    if (lgs) { lgs[j] = -v * 1.4426950408889634f; flo[j] = (pi <= 0.0) ? 1e-10f : 0.0f; }
}

template <int NT>
static int launch_sweep(float *D_hat, const double *U, const double *V, const float *pi_d, const uint32_t *nzmask,
                        double *colsum, const double *Vn, double *DV, int64_t n, int64_t m, int K, hipStream_t st) {
    const int KP2 = (K + 1) & ~1;
    const SweepLds L(KP2, NT);
    const size_t lds = (size_t)L.total * sizeof(float);
    const int64_t rb = (n + 127) / 128;
    // gene ranges: whole tiles of 32, as many per row block as fills the chip's work-group slots evenly
    const int64_t splits0 = pick_splits(rb, (m + 255) / 256);
    int64_t jps = (m + splits0 - 1) / splits0;
    jps = (jps + 31) / 32 * 32;
    const int64_t splits = (m + jps - 1) / jps;
    if (splits > 65535 || rb > 0x7fffffffLL) return ORIANA_EINVAL;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)k_dropout_sweep<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return -1000 - (int)e;
    }
    hipLaunchKernelGGL(k_dropout_sweep<NT>, dim3((unsigned)rb, (unsigned)splits), dim3(256), lds, st, D_hat, U, V, pi_d,
                       nzmask, colsum, Vn, DV, n, m, K, KP2, jps);
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// out[j, k] += sum_i D[i, j] W[i, k] with the products on the bf16 matrix cores (three-way splits, six cross products).
// W is split ONCE per call into per-chunk operand images (k_split_rows: 16 rows per chunk, [n tile][split][lane =
// 32 hh + cc]: W[16 chunk + 8 hh + e][32 nt + cc], e = 0..7); D is split on the fly -- it is read exactly once.
// A wave owns 32 genes x NT * 32 factors: A[m = gene c][k = cell 8 h + e] comes from eight 4-byte loads per lane and
// chunk (128-byte row pieces per half-wave), four chunks ahead in a register ring; B from the staged images.
template <int NT>
__global__ __launch_bounds__(256) void k_split_rows(u4v *__restrict__ img, const double *__restrict__ W, int64_t n, int K) {
    const int tid = threadIdx.x;
    const int nt = tid >> 6, hh = (tid >> 5) & 1, cc = tid & 31;
    if (nt >= NT) return;
    const int64_t i0 = (int64_t)blockIdx.x * 16 + 8 * hh;
    const int kk = nt * 32 + cc;
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = (i0 + e < n && kk < K) ? (float)W[(i0 + e) * K + kk] : 0.f;
    u4v o[3];
    split8(x, o);
    u4v *dst = img + (int64_t)blockIdx.x * (NT * 3 * 64) + (nt * 3) * 64 + hh * 32 + cc;
#pragma unroll
    for (int sp = 0; sp < 3; ++sp) dst[sp * 64] = o[sp];
}

template <int NT>
__global__ __launch_bounds__(256, 2) void k_dt_times_factor_b16(double *__restrict__ out, const float *__restrict__ D,
                                                                const u4v *__restrict__ img, int64_t n, int64_t m, int K,
                                                                int64_t i_per_split) {
    constexpr int CH = 2;                                        // chunks (of 16 rows) per staged group and in the ring
                                                                 // (4: > 256 registers, the slow path's index arithmetic)
    constexpr int NP = NT * 3 * 64;                              // 16-byte pieces per chunk image
    constexpr int RS = (CH * NP + 255) / 256;                    // staged pieces per thread and group
    __shared__ u4v Ws[2][CH * NP];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 31, h = lane >> 5;
    const int64_t jw = ((int64_t)blockIdx.x * 4 + w) * 32;
    const int64_t ib = (int64_t)blockIdx.y * i_per_split;       // multiple of 64 (whole groups)
    const int64_t ie = (ib + i_per_split < n) ? ib + i_per_split : n;
    const int64_t j = jw + c;

    f16v acc[NT], accs[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int v = 0; v < 16; ++v) { acc[nt][v] = 0.f; accs[nt][v] = 0.f; }

    u4v wreg[RS];
    auto stage_load = [&](int64_t i0) {                         // (i0 a multiple of 64; images exist for every chunk of n)
        const u4v *src = img + (i0 >> 4) * NP;
        const int64_t avail = (((n + 15) >> 4) - (i0 >> 4)) * NP;           // pieces left in the image array
#pragma unroll
        for (int r = 0; r < RS; ++r) {
            const int idx = tid + 256 * r;
            wreg[r] = src[(idx < CH * NP && idx < avail) ? idx : 0];
        }
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int r = 0; r < RS; ++r) { const int idx = tid + 256 * r; if ((CH * NP) % 256 == 0 || idx < CH * NP) Ws[buf][idx] = wreg[r]; }
    };
    float ring[CH][8];
    auto load_checked = [&](float (&x)[8], int64_t i0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int64_t i = i0 + 8 * h + e;
            x[e] = (i < ie && j < m) ? D[i * m + j] : 0.f;
        }
    };
    const bool wave_fast = jw + 32 <= m;
    if (ib < ie) {
        stage_load(ib);
        stage_store(0);
#pragma unroll
        for (int ch = 0; ch < CH; ++ch) load_checked(ring[ch], ib + 16 * ch);
    }
    __syncthreads();
    int buf = 0, groups = 0;
    // one staged group = CH chunks; FAST: every row the ring will ask for exists and the wave's genes are inside the
    // matrix -- one running pointer, eight plain loads per chunk
    auto group = [&](auto fast_tag, int64_t i0, bool more) {
        constexpr bool FAST = decltype(fast_tag)::value;
        const float *dp = D + (i0 + 16 * CH + 8 * h) * m + j;    // (dereferenced on the fast path only)
        stage_load(more ? i0 + 16 * CH : i0);                    // (unconditional: the last group is staged again, unread)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ch = 0; ch < CH; ++ch) {
            u4v a[3];
            split8(ring[ch], a);
            // the slot just consumed takes the chunk one group ahead
            if (FAST) {
                const float *r = dp;
#pragma unroll
                for (int e = 0; e < 8; ++e) { ring[ch][e] = *r; r += m; }
                dp += 16 * m;
            } else {
                load_checked(ring[ch], i0 + 16 * (ch + CH));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const u4v *src = &Ws[buf][ch * NP + (nt * 3) * 64 + lane];
                u4v b[3];
#pragma unroll
                for (int sp = 0; sp < 3; ++sp) b[sp] = src[sp * 64];
                ORIANA_MF6(acc[nt], a, b);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        stage_store(buf ^ 1);
    };
    for (int64_t i0 = ib; i0 < ie; i0 += 16 * CH) {
        const bool more = i0 + 16 * CH < ie;
        if (wave_fast && i0 + 32 * CH <= ie) group(std::true_type{}, i0, more);
        else group(std::false_type{}, i0, more);
        if (++groups == 16 / CH) {                               // 256 rows: leave the matrix core
            groups = 0;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int v = 0; v < 16; ++v) { accs[nt][v] += acc[nt][v]; acc[nt][v] = 0.f; }
        }
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int64_t jj = jw + acc_row(v, h);
            const int k = nt * 32 + c;
            if (jj < m && k < K) atomicAdd(&out[jj * K + k], (double)accs[nt][v] + (double)acc[nt][v]);
        }
}

template <int NT>
static int launch_dt_b16(double *out, const float *D, const double *W, float *scratch, int64_t n, int64_t m, int K,
                         hipStream_t st) {
    u4v *img = reinterpret_cast<u4v *>(scratch);
    hipLaunchKernelGGL(k_split_rows<NT>, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, st, img, W, n, K);
    const int64_t jb = (m + 127) / 128;
    const int64_t splits0 = pick_splits(jb, (n + 255) / 256);
    int64_t ips = (n + splits0 - 1) / splits0;
    ips = (ips + 63) / 64 * 64;                                  // whole staged groups
    const int64_t splits = (n + ips - 1) / ips;
    if (splits > 65535 || jb > 0x7fffffffLL || (n + 15) / 16 > 0x7fffffffLL) return ORIANA_EINVAL;
    hipLaunchKernelGGL(k_dt_times_factor_b16<NT>, dim3((unsigned)jb, (unsigned)splits), dim3(256), 0, st, out, D,
                       (const u4v *)img, n, m, K, ips);
    return 0;
}

// scratch of oriana_dropout_sweep_fused (floats): logit(pi_d) [m rounded up to 64] | first images | second images
static inline int64_t b16_img_floats(int64_t m, int pieces) { return ((m + 31) / 32) * (int64_t)pieces * 4; }

template <int NT, int KC>
static int launch_sweep_b16(float *D_hat, const double *U, const double *V, const float *lg, const uint32_t *nzmask,
                            double *colsum, const double *Vn, double *DV, float *img_scratch, int64_t n, int64_t m,
                            int K, hipStream_t st) {
    const B16Lds L(KC, NT);
    const size_t lds = (size_t)L.total;
    u4v *img1 = reinterpret_cast<u4v *>(img_scratch);
    u4v *img2 = Vn ? reinterpret_cast<u4v *>(img_scratch + b16_img_floats(m, KC * 3 * 64)) : nullptr;
    hipLaunchKernelGGL((k_split_images<NT, KC>), dim3((unsigned)((m + 31) / 32)), dim3(256), 0, st, img1, img2, V, Vn, m, K);
    const int64_t rb = (n + 127) / 128;
    const int64_t splits0 = pick_splits(rb, (m + 255) / 256);
    int64_t jps = (m + splits0 - 1) / splits0;
    jps = (jps + 31) / 32 * 32;
    const int64_t splits = (m + jps - 1) / jps;
    if (splits > 65535 || rb > 0x7fffffffLL) return ORIANA_EINVAL;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)k_dropout_sweep_b16<NT, KC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return -1000 - (int)e;
    }
    hipLaunchKernelGGL((k_dropout_sweep_b16<NT, KC>), dim3((unsigned)rb, (unsigned)splits), dim3(256), lds, st, D_hat, U,
                       (const u4v *)img1, lg, nzmask, colsum, (const u4v *)img2, DV, n, m, K, jps);
    return 0;
}

template <int NT, int GQ>
static int launch_dt(double *out, const float *D, const double *W, int64_t n, int64_t m, int K, hipStream_t st) {
    const int64_t jb = (m + 128 * GQ - 1) / (128 * GQ);
    const int64_t splits0 = pick_splits(jb, (n + 255) / 256);
    int64_t ips = (n + splits0 - 1) / splits0;
    ips = (ips + 63) / 64 * 64;                                  // whole staged chunks
    const int64_t splits = (n + ips - 1) / ips;
    if (splits > 65535 || jb > 0x7fffffffLL) return ORIANA_EINVAL;
    hipLaunchKernelGGL((k_dt_times_factor_f32<NT, GQ>), dim3((unsigned)jb, (unsigned)splits), dim3(256), 0, st, out, D, W,
                       n, m, K, ips);
    return 0;
}

// 32 < K <= 100 (gene count a multiple of 4): csrc/dense_zi.hip
namespace dn {
bool zi_supported(int64_t m, int64_t K);
bool zi_dt_supported(int64_t m, int64_t K);
int64_t zi_sweep_image_floats(int64_t m);
int64_t zi_dt_image_floats(int64_t n);
int zi_sweep(float *D_hat, const double *U, const double *V, const float *lgit, int64_t mpad, const uint32_t *nztiles,
             double *colsum, const double *Vn, double *DV, float *img_scratch, int64_t n, int64_t m, int K, hipStream_t st);
int64_t zi_tiles_words(int64_t n, int64_t m);
int zi_tiles(uint32_t *out, const uint32_t *nzmask, int64_t n, int64_t m, hipStream_t st);
int zi_dt(double *out, const float *D, const double *W, float *scratch, int64_t n, int64_t m, int K, hipStream_t st);
}  // namespace dn

}  // namespace oriana

using namespace oriana;

extern "C" int64_t oriana_dropout_sweep_scratch_floats(int64_t m, int64_t K) {
    if (m < 0 || K < 0) return 0;
    // logit(pi_d) + the two operand images of the bf16 path at their largest (K <= 64: 4 k chunks, 2 n tiles; or the
    // images of csrc/dense_zi.hip for 64 < K <= 100)
    const int64_t a = b16_img_floats(m, 4 * 3 * 64) + b16_img_floats(m, 2 * 2 * 3 * 64);
    const int64_t b = (K > 32 && K <= 100) ? dn::zi_sweep_image_floats(m) : 0;
    return 3 * ((m + 63) / 64 * 64) + (a > b ? a : b);       // logits, scaled logits + floors (dense_zi.hip), images
}

extern "C" int64_t oriana_nzmask_tiles_words(int64_t n, int64_t m) {
    if (n < 0 || m < 0) return 0;
    return dn::zi_tiles_words(n, m);
}

extern "C" int oriana_nzmask_tiles(uint32_t *tiles, const uint32_t *nzmask, int64_t n, int64_t m, void *stream) {
    if (n < 0 || m < 0) return ORIANA_EINVAL;
    if (n == 0 || m == 0) return 0;
    if (!tiles || !nzmask || ((uintptr_t)tiles & 15) != 0) return ORIANA_EINVAL;
    const int rc = dn::zi_tiles(tiles, nzmask, n, m, (hipStream_t)stream);
    if (rc) return rc;
    ORIANA_LAUNCH_CHECK();
    return 0;
}

extern "C" int oriana_dropout_sweep_fused_tiles(float *D_hat, const double *U, const double *V, const double *pi_d64,
                                                const uint32_t *nzmask, const uint32_t *nztiles, double *colsum,
                                                const double *V_next, double *DV_next, float *scratch, int arithmetic,
                                                int64_t n, int64_t m, int64_t K, void *stream) {
    if (n < 0 || m < 0 || K <= 0) return ORIANA_EINVAL;
    if (K > 128) return ORIANA_EKRANGE;
    if (n == 0 || m == 0) return 0;
    if (!D_hat || !U || !V || !pi_d64 || !scratch || ((V_next == nullptr) != (DV_next == nullptr))) return ORIANA_EINVAL;
    if (((uintptr_t)scratch & 15) != 0) return ORIANA_EINVAL;
    if (arithmetic != ORIANA_MATRIX_F32 && arithmetic != ORIANA_MATRIX_BF16X3) return ORIANA_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int64_t mpad = (m + 63) / 64 * 64;
    hipLaunchKernelGGL(k_logit_f32, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, scratch, scratch + mpad, scratch + 2 * mpad,
                       pi_d64, m);
    const float *pi_d = scratch;
    scratch += mpad;                                   // (the scaled logits and the floors; the operand images follow them)
    int rc = ORIANA_EKRANGE;
    if (arithmetic == ORIANA_MATRIX_BF16X3 && V_next && nztiles && dn::zi_supported(m, K))
        rc = dn::zi_sweep(D_hat, U, V, scratch, mpad, nztiles, colsum, V_next, DV_next, scratch + 2 * mpad, n, m, (int)K, st);
    scratch += mpad;
    if (rc != ORIANA_EKRANGE) {
        // (done, or failed for good; ORIANA_EKRANGE = not this kernel's case, e.g. a D_hat that is not 16-byte aligned)
    } else if (arithmetic == ORIANA_MATRIX_BF16X3 && K <= 64) {
        float *img = scratch + mpad;
        switch ((int)((K + 15) / 16)) {
            case 1: rc = launch_sweep_b16<1, 1>(D_hat, U, V, pi_d, nzmask, colsum, V_next, DV_next, img, n, m, (int)K, st); break;
            case 2: rc = launch_sweep_b16<1, 2>(D_hat, U, V, pi_d, nzmask, colsum, V_next, DV_next, img, n, m, (int)K, st); break;
            case 3: rc = launch_sweep_b16<2, 3>(D_hat, U, V, pi_d, nzmask, colsum, V_next, DV_next, img, n, m, (int)K, st); break;
            default: rc = launch_sweep_b16<2, 4>(D_hat, U, V, pi_d, nzmask, colsum, V_next, DV_next, img, n, m, (int)K, st); break;
        }
    } else {
        switch ((int)((K + 31) / 32)) {
            case 1: rc = launch_sweep<1>(D_hat, U, V, pi_d, nzmask, colsum, V_next, DV_next, n, m, (int)K, st); break;
            case 2: rc = launch_sweep<2>(D_hat, U, V, pi_d, nzmask, colsum, V_next, DV_next, n, m, (int)K, st); break;
            case 3: rc = launch_sweep<3>(D_hat, U, V, pi_d, nzmask, colsum, V_next, DV_next, n, m, (int)K, st); break;
            default: rc = launch_sweep<4>(D_hat, U, V, pi_d, nzmask, colsum, V_next, DV_next, n, m, (int)K, st); break;
        }
    }
    if (rc) return rc;
    ORIANA_LAUNCH_CHECK();
    return 0;
}

// the round-2 entry: without the per-lane flags (oriana_nzmask_tiles) the K = 33 .. 100 kernel of csrc/dense_zi.hip does not apply (K <= 64 takes
// the bf16 kernels of this file, the rest the float32 matrix instruction)
extern "C" int oriana_dropout_sweep_fused(float *D_hat, const double *U, const double *V, const double *pi_d64,
                                          const uint32_t *nzmask, double *colsum, const double *V_next, double *DV_next,
                                          float *scratch, int arithmetic, int64_t n, int64_t m, int64_t K, void *stream) {
    return oriana_dropout_sweep_fused_tiles(D_hat, U, V, pi_d64, nzmask, nullptr, colsum, V_next, DV_next, scratch, arithmetic, n, m,
                                            K, stream);
}

extern "C" int64_t oriana_dense_t_scratch_floats(int64_t n, int64_t K) {
    if (n < 0 || K < 0) return 0;
    const int64_t a = ((n + 15) / 16 + 4) * (int64_t)(2 * 3 * 64) * 4;       // operand images of W at their largest (K <= 64)
    const int64_t b = (K > 32 && K <= 100) ? dn::zi_dt_image_floats(n) : 0;   // csrc/dense_zi.hip
    return a > b ? a : b;
}

extern "C" int oriana_dense_t_times_factor_f32(double *out, const float *D, const double *W, float *scratch, int arithmetic,
                                               int64_t n, int64_t m, int64_t K, void *stream) {
    if (n < 0 || m < 0 || K <= 0) return ORIANA_EINVAL;
    if (K > 128) return ORIANA_EKRANGE;
    if (n == 0 || m == 0) return 0;
    if (!out || !D || !W) return ORIANA_EINVAL;
    if (arithmetic != ORIANA_MATRIX_F32 && arithmetic != ORIANA_MATRIX_BF16X3) return ORIANA_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    int rc = ORIANA_EKRANGE;
    if (arithmetic == ORIANA_MATRIX_BF16X3 && scratch && ((uintptr_t)scratch & 15) == 0 && dn::zi_dt_supported(m, K))
        rc = dn::zi_dt(out, D, W, scratch, n, m, (int)K, st);
    if (rc != ORIANA_EKRANGE) {
    } else if (arithmetic == ORIANA_MATRIX_BF16X3 && K <= 64) {
        if (!scratch || ((uintptr_t)scratch & 15) != 0) return ORIANA_EINVAL;
        if (K <= 32) rc = launch_dt_b16<1>(out, D, W, scratch, n, m, (int)K, st);
        else rc = launch_dt_b16<2>(out, D, W, scratch, n, m, (int)K, st);
    } else {
        switch ((int)((K + 31) / 32)) {
            case 1: rc = launch_dt<1, 4>(out, D, W, n, m, (int)K, st); break;
            case 2: rc = launch_dt<2, 2>(out, D, W, n, m, (int)K, st); break;
            case 3: rc = launch_dt<3, 1>(out, D, W, n, m, (int)K, st); break;
            default: rc = launch_dt<4, 1>(out, D, W, n, m, (int)K, st); break;
        }
    }
    if (rc) return rc;
    ORIANA_LAUNCH_CHECK();
    return 0;
}

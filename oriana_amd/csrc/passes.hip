// passes.hip -- the responsibility pass of CAVI for pCMF-type models on gfx950.
//
// Replaces the four numba loop nests (oriana/models/gap.py:67-80, zigap.py:79-95,
// sparse_gap.py:81-97, sparse_zigap.py:100-116).  With FU = exp(E[log U] - rowshift) and
// FV = exp(E[log V] - rowshift) (oriana_factor_prep), for every non-zero count x_ij
//     den_ij = sum_k FU[i,k] FV[j,k]            s_ij = x_ij / den_ij
//     Z_i[i,k] = FU[i,k] * sum_j s_ij FV[j,k]    (row pass, register accumulators)
//     Z_j[j,k] = FV[j,k] * sum_i s_ij FU[i,k]    (column pass, register accumulators)
// which is r_ijk = x_ij e_k / sum_k e_k, e_k = exp(lu_ik + lv_jk), summed over j and over i, with
// the shifts cancelling in the ratio.  Zero counts contribute nothing (gap.py:78) and are never
// touched: X lives in HBM as 256 x 256 tiles of sliced non-zero records (pack.hip).
//
// Mapping (wave64): a group of G lanes owns one row (row pass) or one column (column pass) for
// the whole kernel and keeps its K-vector and its accumulator in registers, 4*T4 floats per lane
// (Kp = 4*G*T4).  The other side's K-vectors are staged through LDS, 256 rows at a time, and
// read with ds_read_b128.  A wave streams its slice of the tile 64 slots (one 512-byte load) per
// iteration and walks the four records of each quad with DPP broadcasts; the inner loops have no
// data-dependent branch.  No MFMA: the work is a sampled dot product per non-zero plus two scaled
// vector adds over the sparse support of X, not a dense contraction.
#include "common.h"
#include <string.h>
#include <stdlib.h>

// Ablation switches for kernel analysis builds (never defined in the shipped library).
#ifdef ORIANA_ABLATE_NOBARRIER
#define ORIANA_SYNC() do { } while (0)
#else
#define ORIANA_SYNC() __syncthreads()
#endif
// ORIANA_MASK_PAD: the K-vector reads and FMAs of a step run under the lanes' "this slot holds an entry" mask (padding
// slots then cost no LDS bandwidth); analysis switch
#ifdef ORIANA_MASK_PAD
#define ORIANA_PAD_GUARD(cond) if (cond)
#else
#define ORIANA_PAD_GUARD(cond)
#endif
#if defined(ORIANA_ABLATE_NOSSTORE)
#define ORIANA_S_STORE(dst, off, v) do { if ((v) == 12345.678f) (dst)[(off)] = (v); } while (0)   /* no scattered s stores */
#else
#define ORIANA_S_STORE(dst, off, v) (dst)[(off)] = (v)
#endif
#if defined(ORIANA_ABLATE_SAMEROW)
#define ORIANA_LDS_ROW(base, off) (lds)[(off)]                 /* every group reads image row 0: no bank conflicts */
#elif defined(ORIANA_ABLATE_HALFLDS)
#define ORIANA_LDS_ROW(base, off) (((tt) & 1) ? f4{1.f, 1.f, 1.f, 1.f} : (base)[(off)])   /* half the LDS reads */
#else
#define ORIANA_LDS_ROW(base, off) (base)[(off)]
#endif

namespace oriana {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------
// factor preparation
// ------------------------------------------------------------------------------------------
// Centred validity test of the shifted form.  Only the SUMS lu_ik + lv_jk enter the loop nest, and CAVI drifts
// along the scale indeterminacy U c, V / c: in ZI-pCMF at BASELINE configs[2] the row maxima of E[log U] climb
// from 4 to 45 in 25 sweeps while those of E[log V] sink to -27 (scratch note in DESIGN.md) -- a test on |mu_i| and
// |mv_j| separately then sends EVERY entry down the exact slow path (23 -> 400 ms per sweep) although every sum is
// harmless.  With cu, cv the means of the row maxima of the two sides (k_row_stats; rows beyond +-200 are left out),
// a row takes the shifted form iff |mu_i - cu| < Au (resp. |mv_j - cv| < Av), Au + Av chosen so that every sum
// mu_i + mv_j = (mu_i - cu) + (mv_j - cv) + (cu + cv) stays inside (SUM_LO, SUM_HI): there the reference's own
// float32 den = exp(mu_i + mv_j) den' lies in [3e-30, 3e32] and none of its terms that matter is denormal or
// overflowed -- the condition under which the shifted form provably reproduces gap.py:74-78.
constexpr float SUM_LO = -45.0f, SUM_HI = 70.0f;        // (STAT_MAX: common.h)

struct PrepLimits { float c_own, half, dead_max, c_other, half_other; };
__device__ __forceinline__ PrepLimits prep_limits(const float *__restrict__ stats, int side) {
    PrepLimits L = {0.0f, SHIFT_MAX, DEAD_MAX, 0.0f, SHIFT_MAX};
    if (stats) {
        // stats = {sum, sum of squares, count} of the row maxima of E[log U] (0..2) and of E[log V] (3..5)
        const float nu = stats[2], nv = stats[5];
        const float cu = nu > 0.f ? stats[0] / nu : 0.f, cv = nv > 0.f ? stats[3] / nv : 0.f;
        const float su = nu > 0.f ? sqrtf(fmaxf(stats[1] / nu - cu * cu, 0.f)) : 0.f;
        const float sv = nv > 0.f ? sqrtf(fmaxf(stats[4] / nv - cv * cv, 0.f)) : 0.f;
        // (quantised to 1/16: a coarse grid keeps the centres, and with them the path of every row, stable under
        //  small changes of the inputs)
        const float qu = rintf(cu * 16.f) * 0.0625f, qv = rintf(cv * 16.f) * 0.0625f;
        const float G = qu + qv;
        // total half-width W available to the two sides so that every sum stays inside (SUM_LO, SUM_HI); it is
        // shared in proportion to the sides' spreads (ZI-pCMF ends with the cells' shifts within +-1 of each other
        // and the genes' spread over 40 units)
        float W = fminf(SUM_HI - G, G - SUM_LO);
        if (!(W > 0.f)) W = 0.f;                       // hopeless centre: every row takes the exact path
        const float share = rintf(16.f * (su + 1.f) / (su + sv + 2.f)) * 0.0625f;
        const float Au = W * share, Av = W - Au;
        L.c_own = side ? qv : qu;
        L.half = side ? Av : Au;
        L.c_other = side ? qu : qv;
        L.half_other = side ? Au : Av;
        // a fully masked gene row multiplies exp(lu + lv) by 0: harmless as long as no such exponential overflows
        // against an accepted row of the other side (whose logs stay below c_other + A_other)
        L.dead_max = 85.0f - (side ? qu + Au : qv + Av);
    }
    return L;
}

// sum, sum of squares and count of the row maxima (rows with a NaN, no active entry or |max| > STAT_MAX are left out),
// both sides in one launch: blocks [0, nbu) take E[log U], the others E[log V].  Grid-stride over the rows; every
// work-group stores its three partial sums, and the group that finishes LAST adds them up in block order and writes
// the six results -- no float atomics, so the statistics (and with them the choice of path of every row) are the same
// on every run, and no buffer needs clearing between calls (the last group resets the arrival counter).
// scratch: [0..5] results {sum, sumsq, count} x {U, V}; [6] arrival counter; [7] the den threshold of the row kernels (below);
// [8], [9] smallest row maximum of E[log U], E[log V] (over every row with an active, NaN-free entry);
// [STATS_PART0 + 4 b ...] partials {sum, sumsq, count, min} of block b.
//
// [r4] The den threshold.  The row kernels trust s = x / den' of the shifted form when den' >= threshold; below it the entry
// takes the exact slow path.  What has to hold is that the REFERENCE's own float32 den = exp(mu_i + mv_j) den' is a normal
// number with room to spare (>= 3e-30, the bound the constant DEN_MIN = 1e-10 gives with the smallest sum the validity test
// admits, SUM_LO = -45).  The sums of a given pair of factor matrices do not come near SUM_LO in general: every accepted row
// has mu_i >= max(smallest row maximum, c_u - A_u), likewise mv_j, so with sum_lo the sum of the two bounds the threshold
// 3e-30 exp(-sum_lo), clamped to [1e-25, DEN_MIN], serves the same guarantee (1e-25: a flagged row's den <= 256 FILL stays
// below it, and s = x / den' stays far from overflow).  ZI-pCMF at configs[2] drifts along U c, V / c (above): after 25 sweeps
// the cells' dominant factors sit 25-45 units above the rest and a third of the tiles held entries with den' < 1e-10 --
// 2.6 ms of slow path per sweep and growing; with sum_lo = +3.5 there the threshold is 1e-25 and none is left.
constexpr int STATS_MAX_BLOCKS = 1024;                 // per side
constexpr int STATS_PART0 = 16;
__global__ __launch_bounds__(256) void k_row_stats(float *__restrict__ scratch, const float *__restrict__ logU, int64_t n,
                                                   const float *__restrict__ logV, const float *__restrict__ maskV,
                                                   int64_t m, int K, int nbu, int lane_rows, int dyn_den,
                                                   const float *__restrict__ upart, int nupart) {
    // [r5] upart != NULL (then nbu == 0): the partials of side U were left by the cell-side Gamma update that produced
    // E[log U] (k_gamma_update_vec, PREP outputs: nupart groups x {sum, sumsq, count, min}); this launch covers side V only
    __shared__ float bs[4], bq[4], bc[4], bm[4];
    __shared__ bool last;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const bool vside = (int)blockIdx.x >= nbu;
    const float *logF = vside ? logV : logU;
    const float *mask = vside ? maskV : nullptr;
    const int64_t r = vside ? m : n;
    const int64_t b0 = vside ? (int64_t)blockIdx.x - nbu : blockIdx.x, nb = vside ? (int64_t)gridDim.x - nbu : nbu;
    float sum = 0.f, sq = 0.f, cnt = 0.f, mn = INFINITY;
    if (lane_rows) {
        // narrow rows (K <= 32): one LANE per row -- 256 rows per group in flight at once instead of 4
        for (int64_t row = b0 * 256 + threadIdx.x; row < r; row += nb * 256) {
            const float *l = logF + row * K;
            const float *mk = mask ? mask + row * K : nullptr;
            float mx = -INFINITY;
            bool bad = false, any_on = false;
            #pragma unroll 4
            for (int k = 0; k < K; ++k) {
                const float v = l[k];
                const bool on = mk ? (mk[k] != 0.0f) : true;
                if (on) { any_on = true; if (v != v) bad = true; mx = fmaxf(mx, v); }
            }
            if (any_on && !bad && fabsf(mx) <= STAT_MAX) { sum += mx; sq += mx * mx; cnt += 1.f; }
            if (any_on && !bad) mn = fminf(mn, mx);
        }
        sum = wave_sum(sum); sq = wave_sum(sq); cnt = wave_sum(cnt); mn = -wave_max(-mn);
    } else {
        // one wave per row, FOUR rows of a wave in flight (a wave with one 400-byte read outstanding leaves the pass at
        // 1 TB/s: 0.39 ms for the 400 MB of E[log U] at 1M cells); the rows are accumulated in the order of the plain loop
        constexpr int RU = 4;
        const int64_t step = nb * 4;
        for (int64_t row0 = b0 * 4 + w; row0 < r; row0 += step * RU) {
            float mx[RU];
            bool bad[RU], any_on[RU];
            #pragma unroll
            for (int u = 0; u < RU; ++u) { mx[u] = -INFINITY; bad[u] = false; any_on[u] = false; }
            for (int k = lane; k < K; k += 64) {
                float v[RU], mv[RU];
                #pragma unroll
                for (int u = 0; u < RU; ++u) {
                    const int64_t row = row0 + u * step;
                    const bool in = row < r;
                    v[u] = in ? logF[row * K + k] : 0.f;
                    mv[u] = (in && mask) ? mask[row * K + k] : (in ? 1.0f : 0.0f);
                }
                #pragma unroll
                for (int u = 0; u < RU; ++u)
                    if (mv[u] != 0.0f) { any_on[u] = true; if (v[u] != v[u]) bad[u] = true; mx[u] = fmaxf(mx[u], v[u]); }
            }
            #pragma unroll
            for (int u = 0; u < RU; ++u) {
                const float m1 = wave_max(mx[u]);
                const bool b1 = __any(bad[u]), a1 = __any(any_on[u]);
                if (row0 + u * step < r && a1 && !b1 && fabsf(m1) <= STAT_MAX) { sum += m1; sq += m1 * m1; cnt += 1.f; }
                if (row0 + u * step < r && a1 && !b1) mn = fminf(mn, m1);
            }
        }
    }
    if (lane == 0) { bs[w] = sum; bq[w] = sq; bc[w] = cnt; bm[w] = mn; }
    __syncthreads();
    float *part = scratch + STATS_PART0;
    unsigned *arrived = (unsigned *)(scratch + 6);
    if (threadIdx.x == 0) {
        float *pp = part + 4 * (size_t)blockIdx.x;
        __hip_atomic_store(pp + 0, bs[0] + bs[1] + bs[2] + bs[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(pp + 1, bq[0] + bq[1] + bq[2] + bq[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(pp + 2, bc[0] + bc[1] + bc[2] + bc[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(pp + 3, fminf(fminf(bm[0], bm[1]), fminf(bm[2], bm[3])), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned old = __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        last = old + 1 == gridDim.x;
    }
    __syncthreads();
    if (!last) return;
    // the last group: waves 0 / 1 add up the partials of side U / V, each lane its blocks in order, then the lanes in
    // a fixed tree -- a fixed summation order
    if (w < 2) {
        const bool ext = w == 0 && upart != nullptr;
        const float *src = ext ? upart : part;
        const int lo = ext ? 0 : (w ? nbu : 0), hi = ext ? nupart : (w ? (int)gridDim.x : nbu);
        float a = 0.f, q = 0.f, c = 0.f, lo_max = INFINITY;
        for (int b = lo + lane; b < hi; b += 64) {
            a += __hip_atomic_load(src + 4 * (size_t)b + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            q += __hip_atomic_load(src + 4 * (size_t)b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            c += __hip_atomic_load(src + 4 * (size_t)b + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            lo_max = fminf(lo_max, __hip_atomic_load(src + 4 * (size_t)b + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
        a = wave_sum(a); q = wave_sum(q); c = wave_sum(c); lo_max = -wave_max(-lo_max);
        if (lane == 0) { scratch[3 * w + 0] = a; scratch[3 * w + 1] = q; scratch[3 * w + 2] = c; scratch[8 + w] = lo_max; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // the den threshold of the row kernels from the statistics just written (see above)
        const PrepLimits Lu = prep_limits(scratch, 0);
        const float lo_u = fmaxf(scratch[8], Lu.c_own - Lu.half), lo_v = fmaxf(scratch[9], Lu.c_other - Lu.half_other);
        float thr = DEN_MIN;
        const float sum_lo = lo_u + lo_v;
        if (sum_lo == sum_lo && sum_lo > SUM_LO) thr = fminf(DEN_MIN, fmaxf(3e-30f * expf(-fminf(sum_lo, 80.f)), 1e-25f));
        scratch[7] = dyn_den ? thr : DEN_MIN;
        __hip_atomic_store(arrived, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// one wave per row
__device__ __forceinline__ void factor_prep_row(float *__restrict__ F, float *__restrict__ mu_out,
                                                const float *__restrict__ logF, const float *__restrict__ mask,
                                                const int32_t *__restrict__ row_index, int64_t r, int K, int Kp,
                                                const float *__restrict__ stats, int side, int64_t block) {
    const int lane = threadIdx.x & 63;
    const int64_t row = block * 4 + (threadIdx.x >> 6);
    if (row >= r) return;
    const PrepLimits lim = prep_limits(stats, side);
    const int64_t src = row_index ? (int64_t)row_index[row] : row;
    const float *l = logF + src * K;
    const float *mk = mask ? mask + src * K : nullptr;
    float mx = -INFINITY, mx_all = -INFINITY;
    bool bad = false, bad_all = false, any_on = false;
    for (int k = lane; k < K; k += 64) {
        const float v = l[k];
        const bool on = mk ? (mk[k] != 0.0f) : true;
        if (v != v) bad_all = true;
        mx_all = fmaxf(mx_all, v);
        if (on) { any_on = true; if (v != v) bad = true; mx = fmaxf(mx, v); }
    }
    mx = wave_max(mx);
    mx_all = wave_max(mx_all);
    bad = __any(bad);
    bad_all = __any(bad_all);
    any_on = __any(any_on);
    // A row whose mask is entirely off (a gene with no active factor, sparse_gap.py:113) multiplies every
    // exponential by 0: the reference gets den == 0 -> 1 and a contribution of exactly 0 (sparse_gap.py:88-93)
    // provided no exp(lu + lv) overflows to inf (inf * 0 = NaN).  With its logs below dead_max that cannot happen
    // against an ordinary row of the other side: the row is stored as NEGATIVE zeros (a value no other
    // row can hold), which the row pass of the sparse variants recognises (den == 0 and a -0.0 operand) and
    // skips without the slow path; everywhere else -0.0 behaves as 0.
    const bool dead = (mk != nullptr) && !any_on && !bad_all && (mx_all < lim.dead_max);
    // Rows the shifted form cannot represent faithfully get a tiny constant instead: every entry touching
    // them fails the den >= DEN_MIN test (den <= K * FILL) and is evaluated by the exact slow path, and den
    // stays non-zero, i.e. distinguishable from a dead row.
    const bool flagged = !dead && (bad || !(fabsf(mx - lim.c_own) < lim.half));
    for (int k = lane; k < Kp; k += 64) {
        float out = dead ? -0.0f : 0.0f;                // a dead row is NEGATIVE zero in every (padded) column
        if (k < K && !dead) {
            if (flagged) out = FILL;
            else {
                const float mv = mk ? mk[k] : 1.0f;
                if (mv != 0.0f) out = (float)exp((double)l[k] - (double)mx) * mv;
            }
        }
        F[row * Kp + k] = out;
    }
    if (mu_out && lane == 0) mu_out[row] = (flagged || dead) ? NAN : mx;
}

__global__ __launch_bounds__(256) void k_factor_prep(float *__restrict__ F, float *__restrict__ mu_out,
                                                     const float *__restrict__ logF, const float *__restrict__ mask,
                                                     const int32_t *__restrict__ row_index, int64_t r, int K, int Kp) {
    factor_prep_row(F, mu_out, logF, mask, row_index, r, K, Kp, nullptr, 0, blockIdx.x);
}

// both sides in one launch (blocks [0, nbu): FU, the next nbv: FV), limits from `stats`; the blocks after those
// zero-fill the buffers of `clr` (the outputs and scratch a sweep accumulates into: one launch instead of one fill
// kernel per buffer, which is most of a sweep's time on a small matrix)
__global__ __launch_bounds__(256) void k_factor_prep_pair(float *__restrict__ FU, float *__restrict__ FV,
                                                          const float *__restrict__ logU, const float *__restrict__ logV,
                                                          const float *__restrict__ maskV,
                                                          const int32_t *__restrict__ riu, const int32_t *__restrict__ riv,
                                                          int64_t n, int64_t m, int K, int Kp, int nbu, int nbv,
                                                          const float *__restrict__ stats, oriana_clear_list clr,
                                                          const float *__restrict__ mu_u) {
    if ((int)blockIdx.x < nbu) {
        if (mu_u) {
            // [r5] FU was written by the Gamma update that produced E[log U] (row maxima in mu_u, NaN = the row holds a NaN):
            // only the validity test is left, which needs the statistics of all rows -- one LANE per row, a rejected row is
            // overwritten with the constant of factor_prep_row
            const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
            if (row >= n) return;
            const PrepLimits lim = prep_limits(stats, 0);
            const float mx = mu_u[row];
            if (!(fabsf(mx - lim.c_own) < lim.half)) {
                float *f = FU + row * Kp;
                for (int k = 0; k < K; ++k) f[k] = FILL;
            }
            return;
        }
        factor_prep_row(FU, nullptr, logU, nullptr, riu, n, K, Kp, stats, 0, blockIdx.x);
        return;
    }
    if ((int)blockIdx.x < nbu + nbv) { factor_prep_row(FV, nullptr, logV, maskV, riv, m, K, Kp, stats, 1, (int64_t)blockIdx.x - nbu); return; }
    const int64_t cb = (int64_t)blockIdx.x - nbu - nbv, ncl = (int64_t)gridDim.x - nbu - nbv;
    #pragma unroll 1
    for (int e = 0; e < ORIANA_CLEAR_MAX; ++e) {
        uint32_t *p = static_cast<uint32_t *>(clr.ptr[e]);
        const int64_t words = clr.bytes[e] >> 2;
        if (!p || words <= 0) continue;
        if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
            const int64_t quads = words >> 2;
            uint4 *p4 = reinterpret_cast<uint4 *>(p);
            for (int64_t i = cb * 256 + threadIdx.x; i < quads; i += ncl * 256) p4[i] = uint4{0u, 0u, 0u, 0u};
            for (int64_t i = quads * 4 + cb * 256 + threadIdx.x; i < words; i += ncl * 256) p[i] = 0u;
        } else {
            for (int64_t i = cb * 256 + threadIdx.x; i < words; i += ncl * 256) p[i] = 0u;
        }
    }
}

// ------------------------------------------------------------------------------------------
// LDS geometry shared by the tile kernels
// ------------------------------------------------------------------------------------------
constexpr int lds_stride_floats(int KP) { return (KP + 63) / 64 * 64; }     // rows are 256-B aligned
constexpr int LDS_BUDGET = 160 * 1024;
// smallest power-of-two split of the 256 staged rows such that the LDS image fits
constexpr int pick_nsub(int KP) {
    int nsub = 1;
    while ((TILE / nsub) * lds_stride_floats(KP) * 4 > LDS_BUDGET) nsub *= 2;
    return nsub;
}

// Chunk (float4 index / G inside a row) that a lane visits at step t.  Default: the per-quad rotation
// below.  For G = 4, T4 = 6 (K = 100) a rotation cannot keep the four quads of a 16-lane set on distinct
// 64-byte bank quarters (6 chunks over 4 quarters: two quarters hold two chunks each); the table is a
// schedule with the minimum number of colliding steps (2 of 6 instead of 3; exhaustive search).
template <int G, int T4>
__device__ __forceinline__ int chunk_at(int lane, int rot, int t) {
#if !defined(ORIANA_ABLATE_ROT0) && !defined(ORIANA_ABLATE_ROTQ) && !defined(ORIANA_ABLATE_ROTQ7) && !defined(ORIANA_ABLATE_NOSCHED)
    if (G == 4 && T4 == 6) {
        const int c = (lane >> 2) & 3;
        int ch = t ^ (c & 1);                                   // classes 1, 3 swap inside the pairs
        if ((c & 2) && ch >= 2) ch = (ch < 4) ? ch + 2 : ch - 2;   // classes 2, 3 swap the pairs (2,3) <-> (4,5)
        return ch;
    }
#endif
    return (t + rot) % T4;
}

template <int G>
__device__ __forceinline__ int lds_rot(int lane) {
    // ds_read_b128 is serviced in fixed 16-lane sets; quads that are serviced together must start
    // at different 64-byte quarters of the 256-byte bank row.  Measured on MI355X with
    // tools/ubench/lds_pat.hip (random 512-byte rows, 7 chunks): no rotation 16.5, (Q&7)>>1 6.6,
    // this one 6.1, broadcast floor 5.4 cycles per wave-instruction.
#if defined(ORIANA_ABLATE_ROT0)
    return 0;
#elif defined(ORIANA_ABLATE_ROTQ)
    return (lane >> 2) & 3;
#elif defined(ORIANA_ABLATE_ROTQ7)
    return (lane >> 2) % 7;
#endif
    if (G == 4) { const int Q = lane >> 2; return ((Q & 1) << 1) | ((Q >> 1) & 1); }
    if (G == 8) return ((lane >> 3) & 3) >> 1;
    return 0;
}

template <int U> __device__ __forceinline__ uint32_t qb_u32(uint32_t v) { return quad_bcast_u32<U>(v); }
template <int U> __device__ __forceinline__ float qb_f32(float v) { return quad_bcast_f32<U>(v); }

// Staging of `rows` factor rows (global rows j0 .., bounded by jmax) into an LDS image, split in
// two halves so that the global loads are issued BEFORE the barrier that waits for the previous
// image's readers (their latency overlaps the wait) and only the LDS stores come after it.
// With a tail (TAILREP > 1) the last float4 of a row is replicated TAILREP times behind the row:
// the quads of a wave read their tail float from different copies, i.e. from different LDS banks
// (rows are 512 bytes apart, so without this every quad of a ds_read_b32 would hit the same 4 banks).
template <int KP4, int TAILREP, int ROWS>
struct Stage {
    static constexpr int NST = (ROWS * KP4 + 1023) / 1024;                       // float4 per thread
    static constexpr int NTR = (TAILREP > 1) ? (ROWS * (TAILREP - 1) + 1023) / 1024 : 1;
    f4 v[NST];
    f4 t[NTR];

    __device__ __forceinline__ void load_main(const float *__restrict__ F, int64_t j0, int64_t jmax, int tid) {
        // launder the thread index: the per-element index arithmetic must be redone per tile, not
        // hoisted out of the tile loop into a dozen long-lived address registers
        asm volatile("" : "+v"(tid));
        #pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int idx = tid + u * 1024;
            const int jr = idx / KP4, c4 = idx - jr * KP4;
            const int64_t j = j0 + jr;
            // rows past the end are zero-filled: padding slots point at image row 0 and must read finite values
            v[u] = (idx < ROWS * KP4 && j < jmax) ? reinterpret_cast<const f4 *>(F)[j * KP4 + c4] : f4{0.f, 0.f, 0.f, 0.f};
        }
    }

    __device__ __forceinline__ void load_tail(const float *__restrict__ F, int64_t j0, int64_t jmax, int tid) {
        asm volatile("" : "+v"(tid));
        if (TAILREP > 1) {
            #pragma unroll
            for (int u = 0; u < NTR; ++u) {
                const int idx = tid + u * 1024;
                const int64_t j = j0 + idx / (TAILREP - 1);
                t[u] = (idx < ROWS * (TAILREP - 1) && j < jmax) ? reinterpret_cast<const f4 *>(F)[j * KP4 + (KP4 - 1)]
                                                               : f4{0.f, 0.f, 0.f, 0.f};
            }
        }
    }

    __device__ __forceinline__ void load(const float *__restrict__ F, int64_t j0, int64_t jmax, int tid) {
        load_main(F, j0, jmax, tid);
        load_tail(F, j0, jmax, tid);
    }

    template <int STRIDE4>
    __device__ __forceinline__ void store(f4 *img, int tid) const {
        asm volatile("" : "+v"(tid));
        #pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int idx = tid + u * 1024;
            const int jr = idx / KP4, c4 = idx - jr * KP4;
            if (idx < ROWS * KP4) img[jr * STRIDE4 + c4] = v[u];
        }
        if (TAILREP > 1) {
            #pragma unroll
            for (int u = 0; u < NTR; ++u) {
                const int idx = tid + u * 1024;
                const int jr = idx / (TAILREP - 1), rep = idx - jr * (TAILREP - 1) + 1;
                if (idx < ROWS * (TAILREP - 1)) img[jr * STRIDE4 + (KP4 - 1) + rep] = t[u];
            }
        }
    }
};

// copies of the tail that fit behind a row of KP floats inside its 256-byte aligned stride
constexpr int tail_copies(int KP, int TAIL) {
    if (!TAIL) return 1;
    int free4 = (lds_stride_floats(KP) - KP) / 4 + 1;      // float4 slots from the tail to the end of the stride
    return free4 > 8 ? 8 : free4;
}

// End of a column-pass work item.  The lanes first lay their accumulators out in LDS (the image is no longer
// needed) as the [columns][Kp] block they are in memory; then the whole work-group adds the block to C with
// consecutive lanes on consecutive floats: 256 contiguous bytes per wave instruction, the shape global float
// atomics run at full rate (MI355X_MICROARCH.md, global float atomics).  The register layout would give 16 rows x 4
// dwords 16 bytes apart per instruction instead, measured ~5x slower: ~200 us per item of 512 columns, 10 % of the
// column pass at 125,000 cells (tools/perf_col2.py).  `plain` (deterministic debug mode): the block is stored in
// the item's own slab instead, which k_col_reduce then sums in a fixed order.
template <int NTHREADS>
__device__ __forceinline__ void flush_block(const float *ldsf, float *dst, int nfloats, bool plain, int tid) {
    for (int idx = tid; idx < nfloats; idx += NTHREADS) {
        const float v = ldsf[idx];
#ifdef ORIANA_ABL_NOFLUSH
        if (v == 1.2345f) dst[idx] = v;
#else
        if (plain) dst[idx] = v;
        else if (v != 0.f) atomicAdd(dst + idx, v);
#endif
    }
}

template <int G>
struct WaveGeo {
    static constexpr int RW = 64 / G;
    static constexpr int WPS = 16 / RW;
    static constexpr int OWN = 16 * RW;          // rows owned by the workgroup
    static constexpr int SPLIT = TILE / OWN;
};

// ------------------------------------------------------------------------------------------
// row pass:  s = x / <FU_i, FV_j>,   R_i += w s FV_j
//   VAR bit 0: sparse variant (masked factor rows), writes s in row-side slots (s_rs) INSTEAD of forming R (the caller
//   follows with a row product over s_rs -- sparse models with Kp > 64, NMF start, metrics);  bit 1: per-entry weights
//   w_nz / sw_cs;  bit 2: sparse variant with a SECOND image FV2 (= FV * S_hat, sparse_gap.py:95): the dot product
//   runs against FV, the accumulation against FV2 -- the S_hat-weighted row sums come out of this pass and the second
//   row product (oriana_row_spmm over s_rs) disappears (K with both images in LDS: Kp <= 64)
// ------------------------------------------------------------------------------------------
template <int G, int T4, int TAIL, int VAR>
__global__ __launch_bounds__(1024) void k_row_pass(oriana_counts cm, const float *__restrict__ FU,
                                                   const float *__restrict__ FV, const float *__restrict__ w_nz,
                                                   float *__restrict__ R, float *__restrict__ s_cs,
                                                   float *__restrict__ sw_cs, float *__restrict__ s_rs,
                                                   int32_t *__restrict__ tile_flag, const float *__restrict__ FV2,
                                                   const float *__restrict__ den_min_p) {
    const float den_min = den_min_p ? *den_min_p : DEN_MIN;   // (see k_row_stats: the den threshold)
    constexpr bool F2I = (VAR & 4) != 0;
    constexpr bool SPARSE = (VAR & 5) != 0, SROW = (VAR & 1) != 0 && !F2I, HASW = (VAR & 2) != 0;
    constexpr int PD = HASW ? 2 : 3;            // prefetch depth (iterations), bounded by the register budget
    constexpr int KP = 4 * G * T4 + G * TAIL;   // TAIL: one extra float per lane after the float4 chunks
    constexpr int TOFF = 4 * G * T4;            // float offset of the tail inside a row
    constexpr int TREP = (G == 4) ? tail_copies(KP, TAIL) : 1;
    constexpr int KP4 = KP / 4;
    constexpr int STRIDE4 = lds_stride_floats(KP) / 4;
    constexpr int NSUB = pick_nsub(KP);
    constexpr int CT = TILE / NSUB;
    using Geo = WaveGeo<G>;
    extern __shared__ f4 lds[];                 // [CT][STRIDE4]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = tid & (G - 1), ql = lane & 3;
    const int64_t rb = blockIdx.x / Geo::SPLIT;
    const int part = blockIdx.x % Geo::SPLIT;
    const int sl = __builtin_amdgcn_readfirstlane(part * (16 / Geo::SPLIT) + wave / Geo::WPS);   // slice of the tile
    const int h = wave % Geo::WPS;
    const int g = lane / G;                      // row of the wave
    const int rl = sl * 16 + h * Geo::RW + g;    // row inside the 256-row block
    const int64_t row = rb * TILE + rl;
    const int rec_lane = (h * Geo::RW + g) * 4 + ql;   // this lane's slot inside a 64-slot iteration
    const int rot = lds_rot<G>(lane);
    const int toff_lds = TOFF + ((lane >> 2) % TREP) * 4 + q;    // this lane's tail float inside an LDS row

    int choff[T4];                              // float4 offset of the chunk visited at step t
    #pragma unroll
    for (int t = 0; t < T4; ++t) choff[t] = chunk_at<G, T4>(lane, rot, t) * G + q;

    f4 fu[T4], acc[T4];
    float fut = 0.f, acct = 0.f;                // tail element of this lane
    #pragma unroll
    for (int t = 0; t < T4; ++t) { acc[t] = f4{0.f, 0.f, 0.f, 0.f}; fu[t] = f4{0.f, 0.f, 0.f, 0.f}; }
    if (row < cm.n) {
        #pragma unroll
        for (int t = 0; t < T4; ++t) fu[t] = reinterpret_cast<const f4 *>(FU)[row * KP4 + choff[t]];
        if (TAIL) fut = FU[row * KP + TOFF + q];
    }

    // Sparse variants (the only ones with masked factor rows): an ordinary row holds exp(0) = 1 at its largest
    // log; a row that oriana_factor_prep replaced by the FILL constant does not, and its entries must take the
    // slow path even against a dead (fully masked, -0.0) gene row, whose skip is only certified for
    // ordinary rows.
    bool rowfilled = false;
    if (SPARSE) {
        float fm = fut;
        #pragma unroll
        for (int t = 0; t < T4; ++t) fm = fmaxf(fmaxf(fmaxf(fu[t].x, fu[t].y), fmaxf(fu[t].z, fu[t].w)), fm);
        rowfilled = !(group_max<G>(fm) == 1.0f);
    }

    // gridDim.y > 1 (oriana_row_pass_split, short matrices): this group takes the gene tiles [cb0, cb1) of its row
    // block and stores its row sums in slab blockIdx.y of R
    const int64_t cb0 = (int64_t)blockIdx.y * cm.ncb / gridDim.y, cb1 = ((int64_t)blockIdx.y + 1) * cm.ncb / gridDim.y;
    for (int64_t cb = cb0; cb < cb1; ++cb) {
        const int64_t t = rb * cm.ncb + cb;
        const uint32_t s0 = cm.rslice[t * 17 + sl], s1 = cm.rslice[t * 17 + sl + 1];
        const int niter = __builtin_amdgcn_readfirstlane((int)((s1 - s0) >> 6));
        const int64_t rbase = cm.roff[t] + s0 + rec_lane;           // this lane's slot at iteration 0
        const unsigned long long *recp = reinterpret_cast<const unsigned long long *>(cm.rowrec) + rbase;
        float *sdst = s_cs + cm.coff[t];
        float *swdst = HASW ? sw_cs + cm.coff[t] : nullptr;
        const uint32_t dummy = cm.cslice[t * 17 + 16] + lane;       // write-only slot of the tile
        bool bad = false;
        for (int csub = 0; csub < NSUB; ++csub) {
            // record prefetch ring: the next PD iterations are always in flight (global-load latency
            // is several iterations long); the first ones are issued before the factor rows are
            // staged, so their latency hides behind the staging
            unsigned long long rawq[PD];
            float wq[PD];
            #pragma unroll
            for (int d = 0; d < PD; ++d) {
                const int id = (d < niter) ? d : (niter > 0 ? niter - 1 : 0);
                rawq[d] = 0ull; wq[d] = 1.0f;
                if (niter > 0) { rawq[d] = recp[(int64_t)id * 64]; if (HASW) wq[d] = w_nz[rbase + (int64_t)id * 64]; }
            }
            {
                Stage<KP4, TREP, CT> stg;
                stg.load(FV, cb * TILE + csub * CT, cm.m, tid);
                ORIANA_SYNC();                // everybody is done with the previous image
                stg.template store<STRIDE4>(lds, tid);
            }
            if (F2I) {
                Stage<KP4, TREP, CT> stg2;
                stg2.load(FV2, cb * TILE + csub * CT, cm.m, tid);
                stg2.template store<STRIDE4>(lds + CT * STRIDE4, tid);
            }
            ORIANA_SYNC();
            for (int it = 0; it < niter; ++it) {
                uint32_t rx = (uint32_t)rawq[0], rm = (uint32_t)(rawq[0] >> 32);
                const float wcur = wq[0];
                #pragma unroll
                for (int d = 0; d + 1 < PD; ++d) { rawq[d] = rawq[d + 1]; wq[d] = wq[d + 1]; }
                // refill the ring (clamped: past the end it re-reads the last iteration)
                const int nx = (it + PD < niter) ? it + PD : niter - 1;
                rawq[PD - 1] = recp[(int64_t)nx * 64];
                if (HASW) wq[PD - 1] = w_nz[rbase + (int64_t)nx * 64];
                float sbuf = 0.f;
#define ORIANA_ROW_STEP(U)                                                                            \
                {                                                                                     \
                    const uint32_t bm = qb_u32<U>(rm);                                                \
                    const float x = __uint_as_float(qb_u32<U>(rx));                                   \
                    int col = (int)((bm >> 16) & 0xFFu);                                              \
                    bool valid = (x != 0.f);                                                          \
                    if (NSUB > 1) { valid = valid && (col / CT == csub); col &= (CT - 1); }           \
                    const f4 *vrow = lds + col * STRIDE4;                                             \
                    f4 v[T4];                                                                         \
                    _Pragma("unroll") for (int tt = 0; tt < T4; ++tt) v[tt] = ORIANA_LDS_ROW(vrow, choff[tt]);        \
                    float vt = 0.f;                                                                   \
                    if (TAIL) vt = reinterpret_cast<const float *>(vrow)[toff_lds];                   \
                    f2 d01 = {0.f, 0.f}, d23 = {0.f, 0.f};                                            \
                    _Pragma("unroll") for (int tt = 0; tt < T4; ++tt) {                               \
                        d01 = __builtin_elementwise_fma(fu[tt].xy, v[tt].xy, d01);                    \
                        d23 = __builtin_elementwise_fma(fu[tt].zw, v[tt].zw, d23);                    \
                    }                                                                                 \
                    const f2 dd = d01 + d23;                                                          \
                    const float den = group_sum<G>(TAIL ? fmaf(fut, vt, dd.x + dd.y) : dd.x + dd.y);  \
                    const bool ok = den >= den_min;          /* false for 0, tiny and NaN */          \
                    const float s = (ok && valid) ? x * __builtin_amdgcn_rcpf(den) : 0.f;             \
                    const float sw = HASW ? s * qb_f32<U>(wcur) : s;                                  \
                    const f2 ss = {sw, sw};                                                           \
                    if (F2I) {                /* accumulate against the second image */              \
                        const f4 *vrow2 = vrow + CT * STRIDE4;                                        \
                        _Pragma("unroll") for (int tt = 0; tt < T4; ++tt) {                           \
                            const f4 v2 = ORIANA_LDS_ROW(vrow2, choff[tt]);                           \
                            acc[tt].xy = __builtin_elementwise_fma(ss, v2.xy, acc[tt].xy);            \
                            acc[tt].zw = __builtin_elementwise_fma(ss, v2.zw, acc[tt].zw);            \
                        }                                                                             \
                        if (TAIL) acct = fmaf(sw, reinterpret_cast<const float *>(vrow2)[toff_lds], acct); \
                    } else if (!SROW) {       /* (with s_rs the caller only wants s: R is not formed) */ \
                        _Pragma("unroll") for (int tt = 0; tt < T4; ++tt) {                           \
                            acc[tt].xy = __builtin_elementwise_fma(ss, v[tt].xy, acc[tt].xy);         \
                            acc[tt].zw = __builtin_elementwise_fma(ss, v[tt].zw, acc[tt].zw);         \
                        }                                                                             \
                        if (TAIL) acct = fmaf(sw, vt, acct);                                          \
                    }                                                                                 \
                    /* fully masked gene (a -0.0 row): exactly zero contribution, no slow path */     \
                    const bool dead = SPARSE && !rowfilled && den == 0.f &&                           \
                                      __float_as_uint(v[0].x) == 0x80000000u;                         \
                    const bool slow = valid && !ok && !dead; /* NaN = "evaluate me exactly" */       \
                    bad = bad || slow;                                                                \
                    const float sout = slow ? NAN : s;                                                \
                    const uint32_t off = valid ? (bm & 0xFFFFu) : dummy;                              \
                    ORIANA_S_STORE(sdst, off, sout);                                                  \
                    if (HASW) swdst[off] = slow ? NAN : sw;                                           \
                    if (SROW) sbuf = (ql == U) ? sout : sbuf;                                         \
                    /* step fence: one step's K-vector live at a time (keeps the kernel spill-free) */ \
                    _Pragma("unroll") for (int tt = 0; tt < T4; ++tt) asm volatile("" : "+v"(acc[tt]));  \
                    asm volatile("" : "+v"(rm), "+v"(rx));                                            \
                }
                ORIANA_ROW_STEP(0)
                ORIANA_ROW_STEP(1)
                ORIANA_ROW_STEP(2)
                ORIANA_ROW_STEP(3)
#undef ORIANA_ROW_STEP
                if (SROW) {
                    // row-side copy of s (one coalesced store per iteration).  With column sub-tiles
                    // an entry is valid in exactly one of them: later sub-tiles only add their own.
                    if (q < 4) {
                        float *dst = s_rs + rbase + (int64_t)it * 64;
                        if (NSUB == 1 || csub == 0) *dst = sbuf;
                        else if (sbuf != 0.f) *dst = sbuf;
                    }
                }
            }
        }
        if (__any(bad) && lane == 0) tile_flag[t] = 1;
    }
    if (row < cm.n && !SROW) {
        float *Rs = R + (int64_t)blockIdx.y * cm.n * KP;
        #pragma unroll
        for (int t = 0; t < T4; ++t) reinterpret_cast<f4 *>(Rs)[row * KP4 + choff[t]] = acc[t];
        if (TAIL) Rs[row * KP + TOFF + q] = acct;
    }
}

// ------------------------------------------------------------------------------------------
// row SpMM with given s (row-side slots):  R_i = sum_j w s FV_j
// ------------------------------------------------------------------------------------------
template <int G, int T4, int TAIL, bool HASW>
__global__ __launch_bounds__(1024) void k_row_spmm(oriana_counts cm, const float *__restrict__ s_rs,
                                                   const float *__restrict__ w_nz, const float *__restrict__ FV,
                                                   float *__restrict__ R) {
    constexpr int KP = 4 * G * T4 + G * TAIL;
    constexpr int TOFF = 4 * G * T4;
    constexpr int TREP = (G == 4) ? tail_copies(KP, TAIL) : 1;
    constexpr int KP4 = KP / 4;
    constexpr int STRIDE4 = lds_stride_floats(KP) / 4;
    constexpr int NSUB = pick_nsub(KP);
    constexpr int CT = TILE / NSUB;
    using Geo = WaveGeo<G>;
    extern __shared__ f4 lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = tid & (G - 1), ql = lane & 3;
    const int64_t rb = blockIdx.x / Geo::SPLIT;
    const int part = blockIdx.x % Geo::SPLIT;
    const int sl = __builtin_amdgcn_readfirstlane(part * (16 / Geo::SPLIT) + wave / Geo::WPS);
    const int h = wave % Geo::WPS;
    const int g = lane / G;
    const int rl = sl * 16 + h * Geo::RW + g;
    const int64_t row = rb * TILE + rl;
    const int rec_lane = (h * Geo::RW + g) * 4 + ql;
    const int rot = lds_rot<G>(lane);
    const int toff_lds = TOFF + ((lane >> 2) % TREP) * 4 + q;
    int choff[T4];
    #pragma unroll
    for (int t = 0; t < T4; ++t) choff[t] = chunk_at<G, T4>(lane, rot, t) * G + q;
    f4 acc[T4];
    float acct = 0.f;
    #pragma unroll
    for (int t = 0; t < T4; ++t) acc[t] = f4{0.f, 0.f, 0.f, 0.f};

    for (int64_t cb = 0; cb < cm.ncb; ++cb) {
        const int64_t t = rb * cm.ncb + cb;
        const uint32_t s0 = cm.rslice[t * 17 + sl], s1 = cm.rslice[t * 17 + sl + 1];
        const int niter = __builtin_amdgcn_readfirstlane((int)((s1 - s0) >> 6));
        const int64_t rbase = cm.roff[t] + s0 + rec_lane;
        const unsigned long long *recp = reinterpret_cast<const unsigned long long *>(cm.rowrec) + rbase;
        for (int csub = 0; csub < NSUB; ++csub) {
            uint32_t rm = 0; float sv = 0.f;
            if (niter > 0) { rm = (uint32_t)(recp[0] >> 32); sv = s_rs[rbase]; if (HASW) sv *= w_nz[rbase]; }
            Stage<KP4, TREP, CT> stg;
            stg.load(FV, cb * TILE + csub * CT, cm.m, tid);
            ORIANA_SYNC();
            stg.template store<STRIDE4>(lds, tid);
            ORIANA_SYNC();
            for (int it = 0; it < niter; ++it) {
                const uint32_t rmc = rm; const float svc = sv;
                const int nx = (it + 1 < niter) ? it + 1 : it;
                rm = (uint32_t)(recp[(int64_t)nx * 64] >> 32);
                sv = s_rs[rbase + (int64_t)nx * 64];
                if (HASW) sv *= w_nz[rbase + (int64_t)nx * 64];
#define ORIANA_SPMM_STEP(U)                                                                           \
                {                                                                                     \
                    const uint32_t bm = qb_u32<U>(rmc);                                               \
                    float s = qb_f32<U>(svc);                                                         \
                    int col = (int)((bm >> 16) & 0xFFu);                                              \
                    if (NSUB > 1) { if (col / CT != csub) s = 0.f; col &= (CT - 1); }                  \
                    const f4 *vrow = lds + col * STRIDE4;                                             \
                    const f2 ss = {s, s};                                                             \
                    _Pragma("unroll") for (int tt = 0; tt < T4; ++tt) {                               \
                        const f4 v = ORIANA_LDS_ROW(vrow, choff[tt]);                                                 \
                        acc[tt].xy = __builtin_elementwise_fma(ss, v.xy, acc[tt].xy);                 \
                        acc[tt].zw = __builtin_elementwise_fma(ss, v.zw, acc[tt].zw);                 \
                    }                                                                                 \
                    if (TAIL) acct = fmaf(s, reinterpret_cast<const float *>(vrow)[toff_lds], acct);  \
                }
                ORIANA_SPMM_STEP(0)
                ORIANA_SPMM_STEP(1)
                ORIANA_SPMM_STEP(2)
                ORIANA_SPMM_STEP(3)
#undef ORIANA_SPMM_STEP
            }
        }
    }
    if (row < cm.n) {
        #pragma unroll
        for (int t = 0; t < T4; ++t) reinterpret_cast<f4 *>(R)[row * KP4 + choff[t]] = acc[t];
        if (TAIL) R[row * KP + TOFF + q] = acct;
    }
}

// ------------------------------------------------------------------------------------------
// column pass:  C_j += sum_i s_ij G_i      (grid.y = row bands, combined with float atomics)
// ------------------------------------------------------------------------------------------
template <int G, int T4, int TAIL>
__global__ __launch_bounds__(1024) void k_col_pass(oriana_counts cm, const float *__restrict__ s_cs,
                                                   const float *__restrict__ Gm, float *__restrict__ C,
                                                   const int32_t *__restrict__ work, int64_t rb_per_band,
                                                   float *__restrict__ Cpart) {
    constexpr int KP = 4 * G * T4 + G * TAIL;
    constexpr int TOFF = 4 * G * T4;
    constexpr int TREP = (G == 4) ? tail_copies(KP, TAIL) : 1;
    constexpr int KP4 = KP / 4;
    constexpr int STRIDE4 = lds_stride_floats(KP) / 4;
    constexpr int NSUB = pick_nsub(KP);
    constexpr int RT = TILE / NSUB;
    constexpr int CPD = 4;                      // prefetch depth (iterations)
    using Geo = WaveGeo<G>;
    extern __shared__ f4 lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = tid & (G - 1), ql = lane & 3;
    // work item: a column block and a band of row blocks.  With a work list (built at pack time
    // from the tile sizes) every item carries about the same number of slots; without one,
    // grid.y enumerates uniform bands.
    const int64_t item = blockIdx.x / Geo::SPLIT;
    const int part = blockIdx.x % Geo::SPLIT;
    int64_t cb, rb0, rb1;
    if (work) {
        cb = work[item * 3 + 0]; rb0 = work[item * 3 + 1]; rb1 = work[item * 3 + 2];
    } else {
        cb = item;
        rb0 = (int64_t)blockIdx.y * rb_per_band;
        rb1 = (rb0 + rb_per_band < cm.nrb) ? rb0 + rb_per_band : cm.nrb;
    }
    const int sl = __builtin_amdgcn_readfirstlane(part * (16 / Geo::SPLIT) + wave / Geo::WPS);
    const int h = wave % Geo::WPS;
    const int g = lane / G;
    const int ent_lane = (h * Geo::RW + g) * 4 + ql;
    const int rot = lds_rot<G>(lane);
    const int toff_lds = TOFF + ((lane >> 2) % TREP) * 4 + q;
    int choff[T4];
    #pragma unroll
    for (int t = 0; t < T4; ++t) choff[t] = chunk_at<G, T4>(lane, rot, t) * G + q;
    f4 acc[T4];
    float acct = 0.f;
    #pragma unroll
    for (int t = 0; t < T4; ++t) acc[t] = f4{0.f, 0.f, 0.f, 0.f};

#define ORIANA_COL_STEP(U)                                                                            \
                {                                                                                     \
                    float s = qb_f32<U>(svc);                                                         \
                    int r = (int)qb_u32<U>(rvc);                                                      \
                    if (NSUB > 1) { if (r / RT != rsub) s = 0.f; r &= (RT - 1); }                      \
                    const f4 *vrow = lds + r * STRIDE4;                                               \
                    const f2 ss = {s, s};                                                             \
                    _Pragma("unroll") for (int tt = 0; tt < T4; ++tt) {                               \
                        const f4 v = ORIANA_LDS_ROW(vrow, choff[tt]);                                 \
                        acc[tt].xy = __builtin_elementwise_fma(ss, v.xy, acc[tt].xy);                 \
                        acc[tt].zw = __builtin_elementwise_fma(ss, v.zw, acc[tt].zw);                 \
                    }                                                                                 \
                    if (TAIL) acct = fmaf(s, reinterpret_cast<const float *>(vrow)[toff_lds], acct);  \
                }
    if (NSUB == 1) {
        // One image per tile.  Everything the NEXT tile needs from memory -- its factor rows and the first
        // CPD iterations of its (s, row index) stream -- is requested CPD iterations before the end of
        // the current tile, between two branch-free loops (a request inside a loop body would make the
        // compiler drain the memory counter at every iteration), so that the barrier + image rewrite
        // between tiles no longer waits for HBM.
        constexpr int rsub = 0;
        Stage<KP4, TREP, RT> stg;
        float svq[CPD]; uint32_t rvq[CPD];
        int niter = 0;
        int64_t cbase = 0;
        auto tile_geo = [&](int64_t rb, int &ni, int64_t &cbs) {
            const int64_t t = rb * cm.ncb + cb;
            const uint32_t s0 = cm.cslice[t * 17 + sl], s1 = cm.cslice[t * 17 + sl + 1];
            ni = __builtin_amdgcn_readfirstlane((int)((s1 - s0) >> 6));
            cbs = cm.coff[t] + s0 + ent_lane;
        };
        auto ring_fill = [&](float (&sv)[CPD], uint32_t (&rv)[CPD], int ni, int64_t cbs) {
            #pragma unroll
            for (int d = 0; d < CPD; ++d) {
                const int id = (d < ni) ? d : (ni > 0 ? ni - 1 : 0);
                sv[d] = 0.f; rv[d] = 0;
                if (ni > 0) { sv[d] = s_cs[cbs + (int64_t)id * 64]; rv[d] = cm.ridx[cbs + (int64_t)id * 64]; }
            }
        };
        if (rb0 < rb1) {
            tile_geo(rb0, niter, cbase);
            ring_fill(svq, rvq, niter, cbase);
            stg.load_main(Gm, rb0 * TILE, cm.n, tid);
        }
        for (int64_t rb = rb0; rb < rb1; ++rb) {
            stg.load_tail(Gm, rb * TILE, cm.n, tid);               // the (L2-hot) tail replicas: late, few registers
            ORIANA_SYNC();
            stg.template store<STRIDE4>(lds, tid);
            ORIANA_SYNC();
            const int n_main = (niter > CPD) ? niter - CPD : 0;
            for (int it = 0; it < n_main; ++it) {
                const float svc = svq[0]; const uint32_t rvc = rvq[0];
                #pragma unroll
                for (int d = 0; d + 1 < CPD; ++d) { svq[d] = svq[d + 1]; rvq[d] = rvq[d + 1]; }
                svq[CPD - 1] = s_cs[cbase + (int64_t)(it + CPD) * 64];
                rvq[CPD - 1] = cm.ridx[cbase + (int64_t)(it + CPD) * 64];
                ORIANA_COL_STEP(0)
                ORIANA_COL_STEP(1)
                ORIANA_COL_STEP(2)
                ORIANA_COL_STEP(3)
            }
            float svn[CPD]; uint32_t rvn[CPD];
            int niter_n = 0;
            int64_t cbase_n = 0;
            if (rb + 1 < rb1) {
                stg.load_main(Gm, (rb + 1) * TILE, cm.n, tid);
                tile_geo(rb + 1, niter_n, cbase_n);
                ring_fill(svn, rvn, niter_n, cbase_n);
            }
            for (int it = n_main; it < niter; ++it) {
                const float svc = svq[0]; const uint32_t rvc = rvq[0];
                #pragma unroll
                for (int d = 0; d + 1 < CPD; ++d) { svq[d] = svq[d + 1]; rvq[d] = rvq[d + 1]; }
                ORIANA_COL_STEP(0)
                ORIANA_COL_STEP(1)
                ORIANA_COL_STEP(2)
                ORIANA_COL_STEP(3)
            }
            if (rb + 1 < rb1) {
                #pragma unroll
                for (int d = 0; d < CPD; ++d) { svq[d] = svn[d]; rvq[d] = rvn[d]; }
                niter = niter_n;
                cbase = cbase_n;
            }
        }
    } else {
    for (int64_t rb = rb0; rb < rb1; ++rb) {
        const int64_t t = rb * cm.ncb + cb;
        const uint32_t s0 = cm.cslice[t * 17 + sl], s1 = cm.cslice[t * 17 + sl + 1];
        const int niter = __builtin_amdgcn_readfirstlane((int)((s1 - s0) >> 6));
        const int64_t cbase = cm.coff[t] + s0 + ent_lane;
        for (int rsub = 0; rsub < NSUB; ++rsub) {
            // prefetch ring over the next CPD iterations (s and the row index of each slot)
            float svq[CPD]; uint32_t rvq[CPD];
            #pragma unroll
            for (int d = 0; d < CPD; ++d) {
                const int id = (d < niter) ? d : (niter > 0 ? niter - 1 : 0);
                svq[d] = 0.f; rvq[d] = 0;
                if (niter > 0) { svq[d] = s_cs[cbase + (int64_t)id * 64]; rvq[d] = cm.ridx[cbase + (int64_t)id * 64]; }
            }
            Stage<KP4, TREP, RT> stg;
            stg.load(Gm, rb * TILE + rsub * RT, cm.n, tid);
            ORIANA_SYNC();
            stg.template store<STRIDE4>(lds, tid);
            ORIANA_SYNC();
            for (int it = 0; it < niter; ++it) {
                const float svc = svq[0]; const uint32_t rvc = rvq[0];
                #pragma unroll
                for (int d = 0; d + 1 < CPD; ++d) { svq[d] = svq[d + 1]; rvq[d] = rvq[d + 1]; }
                const int nx = (it + CPD < niter) ? it + CPD : niter - 1;
                svq[CPD - 1] = s_cs[cbase + (int64_t)nx * 64];
                rvq[CPD - 1] = cm.ridx[cbase + (int64_t)nx * 64];
                ORIANA_COL_STEP(0)
                ORIANA_COL_STEP(1)
                ORIANA_COL_STEP(2)
                ORIANA_COL_STEP(3)
            }
        }
    }
    }
#undef ORIANA_COL_STEP
    {
        // (everything below is recomputed from the block index: nothing extra stays live across the tile loops)
        const bool plain = Cpart != nullptr;
        float *ldsf = reinterpret_cast<float *>(lds);
        const int64_t item2 = blockIdx.x / Geo::SPLIT;
        const int cl0 = (int)(blockIdx.x % Geo::SPLIT) * Geo::OWN;  // first column of this work-group inside the tile
        const int64_t cb2 = work ? (int64_t)work[item2 * 3] : item2;
        const int cl2 = (tid >> 6) / Geo::WPS * 16 + ((tid >> 6) % Geo::WPS) * Geo::RW + (tid & 63) / G;   // column inside the work-group's range
        ORIANA_SYNC();                                             // every wave is done with the last image
        if (cb2 * TILE + cl0 + cl2 < cm.m) {
            float *row = ldsf + cl2 * KP;
            #pragma unroll
            for (int t = 0; t < T4; ++t) *reinterpret_cast<f4 *>(row + choff[t] * 4) = acc[t];
            if (TAIL) row[TOFF + q] = acct;
        }
        ORIANA_SYNC();
        const int64_t c0 = cb2 * TILE + cl0;
        const int64_t left = cm.m - c0;
        const int ncols = left < Geo::OWN ? (left > 0 ? (int)left : 0) : Geo::OWN;
        float *dst = plain ? Cpart + (item2 * TILE + cl0) * KP : C + c0 * KP;
        flush_block<1024>(ldsf, dst, ncols * KP, plain, tid);
    }
}

// deterministic debug mode: C[col, :] += sum over the work items of the column block, in item order
__global__ __launch_bounds__(256) void k_col_reduce(float *__restrict__ C, const float *__restrict__ Cpart,
                                                    const int32_t *__restrict__ work, int64_t nwork, int64_t m,
                                                    int KP, int width) {
    __shared__ int32_t list[4096];
    __shared__ int nlist;
    const int64_t blk = blockIdx.x;
    const int ncol = width * TILE;
    for (int64_t base = 0; base < nwork; base += 4096) {           // (one pass for any realistic work list)
        if (threadIdx.x == 0) {                                   // one thread: the list keeps the items' order
            int c = 0;
            const int64_t end = (base + 4096 < nwork) ? base + 4096 : nwork;
            for (int64_t it = base; it < end; ++it)
                if (work[it * 3] == (int32_t)blk) list[c++] = (int32_t)it;
            nlist = c;
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < ncol * KP; idx += 256) {
            const int64_t col = blk * ncol + idx / KP;
            if (col >= m) continue;
            float acc = 0.f;
            for (int j = 0; j < nlist; ++j) acc += Cpart[(int64_t)list[j] * ncol * KP + idx];
            C[col * KP + (idx % KP)] += acc;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// finalize:  Z = [Z +] F * R [* mul]     dense (r, K) out from padded (r, Kp) in
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_finalize(float *__restrict__ Z, const float *__restrict__ F,
                                                  const float *__restrict__ R, const float *__restrict__ mul,
                                                  const int32_t *__restrict__ row_index, int64_t r, int K, int Kp,
                                                  int accumulate, int nslab, int64_t slab_row0) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= r * K) return;
    const int64_t row = idx / K;
    const int k = (int)(idx - row * K);
    const int64_t o = (row_index ? (int64_t)row_index[row] : row) * K + k;
    const float f = F[row * Kp + k];
    float rr = R[row * Kp + k];
    const int ns = row >= slab_row0 ? nslab : 1;                                        // (oriana_row_split: full row blocks have slab 0 only)
    for (int sl = 1; sl < ns; ++sl) rr += R[((int64_t)sl * (r - slab_row0) + row) * Kp + k];         // oriana_row_pass_split
    float v = f * rr;
    if (mul) v *= mul[o];
    // (+ 0: a dead factor row is -0.0; the outputs carry +0.  The accumulating form without a multiplier is spelled as
    //  ONE fused multiply-add: k_gamma_update<true> folds this statement in and must round the same way)
    Z[o] = (accumulate ? (mul ? Z[o] + v : fmaf(f, rr, Z[o])) : v) + 0.0f;
}

// ------------------------------------------------------------------------------------------
// fix-up: exact reference arithmetic for the entries flagged with the NaN sentinel
// ------------------------------------------------------------------------------------------
// grid = tiles; block = 256 threads.  [r5] A flagged tile is scanned in windows of 2048 row-side slots; the sentinels of a
// window are queued in LDS and then evaluated by a WAVE each, lanes over the factors: the K exponentials of an entry run in
// parallel and its additions to Z_i / Z_j / Z_log are contiguous K-vectors (one coalesced float atomic per wave and matrix).
// Round 4 gave every sentinel to one THREAD: 2K expf in sequence and, at each k, 64 atomics of a wave to 64 different rows --
// the slowest shape float atomics have on this part (guide: 64 lanes in 64 rows ~ 17 x slower than a contiguous 256 bytes).
// After the reference's default NMF start a ZI-pCMF fit at configs[2] passes through sweeps with 5,600 of 30,889 tiles
// flagged: 4 ms of slow path per sweep in that form.  The arithmetic of an entry is unchanged: expf of the float32 sum, den
// added up LEFT TO RIGHT in float32 (every lane runs the same chain over the wave's LDS copy of the exponentials), the
// den > 0 guard, (x e) / den.
constexpr int FIX_WINDOW = 2048, FIX_KMAX = 256;
__global__ __launch_bounds__(256) void k_fixup(oriana_counts cm, const int32_t *__restrict__ tile_flag,
                                               float *__restrict__ s_cs, float *__restrict__ sw_cs,
                                               float *__restrict__ s_rs, const float *__restrict__ logU,
                                               const float *__restrict__ logV, const float *__restrict__ S_tilde,
                                               const float *__restrict__ S_hat, const float *__restrict__ w_nz,
                                               const float *__restrict__ dq, float *__restrict__ Zi,
                                               float *__restrict__ Zj, float *__restrict__ Zlog, int K, int quirk) {
    const int64_t t = blockIdx.x;
    if (tile_flag[t] == 0) return;
    __shared__ uint32_t queue[FIX_WINDOW];
    __shared__ uint32_t rs[17];
    __shared__ uint32_t qn;
    __shared__ __attribute__((aligned(16))) float ebuf[4][FIX_KMAX];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t rb = t / cm.ncb, cb = t - rb * cm.ncb;
    const int64_t rbase = cm.roff[t], cbase = cm.coff[t];
    if (tid < 17) rs[tid] = cm.rslice[t * 17 + tid];
    __syncthreads();
    const uint32_t total = rs[16];
    const int K4 = (K + 3) & ~3;
    for (uint32_t base = 0; base < total; base += FIX_WINDOW) {
        if (tid == 0) qn = 0;
        __syncthreads();
        #pragma unroll
        for (int u = 0; u < FIX_WINDOW / 256; ++u) {
            const uint32_t slot = base + (uint32_t)u * 256u + (uint32_t)tid;
            if (slot < total) {
                const oriana_rowrec rec = cm.rowrec[rbase + slot];
                if (rec.x != 0.f) {                                  // (0: padding)
                    const float sv = s_cs[cbase + rec.cdst];
                    if (sv != sv) queue[atomicAdd(&qn, 1u)] = slot;  // a sentinel
                }
            }
        }
        __syncthreads();
        const uint32_t nq = qn;
        for (uint32_t q = wave; q < nq; q += 4) {
            const uint32_t slot = queue[q];
            int sl = 0;
            #pragma unroll
            for (int c = 1; c < 16; ++c) sl += (slot >= rs[c]) ? 1 : 0;
            const oriana_rowrec rec = cm.rowrec[rbase + slot];
            const int rl = sl * 16 + (int)((slot & 63u) >> 2);       // (slices start at multiples of 64 slots)
            const int64_t ip = rb * TILE + rl;                       // packed row / column
            const int64_t jp = cb * TILE + rec.col;
            const int64_t i = cm.row_perm ? (int64_t)cm.row_perm[ip] : ip;   // caller's row / gene
            const int64_t j = cm.col_perm ? (int64_t)cm.col_perm[jp] : jp;
            const float x = rec.x;
            const float w = w_nz ? w_nz[rbase + slot] : 1.0f;
            float ls[FIX_KMAX / 64], e[FIX_KMAX / 64];
            #pragma unroll
            for (int r = 0; r < FIX_KMAX / 64; ++r) {
                const int k = lane + 64 * r;
                ls[r] = 0.f; e[r] = 0.f;
                if (k < K) {
                    ls[r] = logU[i * K + k] + logV[j * K + k];
                    e[r] = expf(ls[r]);
                    if (S_tilde) e[r] *= S_tilde[j * K + k];
                }
                if (k < K4) ebuf[wave][k] = e[r];                    // (zeros up to a multiple of 4: den + 0 = den)
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (the wave's own LDS writes, before its lanes read them)
            __builtin_amdgcn_wave_barrier();
            // den = sum_k exp(lu + lv) [* S_tilde], float32, left to right (gap.py:74-76)
            float den = 0.f;
            for (int k = 0; k < K4; k += 4) {
                const f4 v = *reinterpret_cast<const f4 *>(&ebuf[wave][k]);
                den += v.x; den += v.y; den += v.z; den += v.w;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (read before the next entry overwrites the copy)
            __builtin_amdgcn_wave_barrier();
            den = (den > 0.f) ? den : 1.0f;
            #pragma unroll
            for (int r = 0; r < FIX_KMAX / 64; ++r) {
                const int k = lane + 64 * r;
                if (k >= K) continue;
                const float expectation = (x * e[r]) / den;          // gap.py:78
                if (Zi) {
                    float wi = w;
                    if (S_hat) wi = w_nz ? w * S_hat[j * K + k] : S_hat[j * K + k];   // sparse_zigap.py:114 / sparse_gap.py:95
                    const float v = (w_nz || S_hat) ? wi * expectation : expectation;
                    if (v != 0.f) atomicAdd(&Zi[i * K + k], v);
                }
                if (Zj) {
                    float v = expectation;
                    if ((quirk & 1) && dq) v = dq[i * K + k] * expectation;   // zigap.py:94 (D_hat[i, k])
                    else if (w_nz) v = w * expectation;                 // sparse_zigap.py:115
                    // (quirk bit 1: Zj is indexed by the PACKED gene index -- the sharded pCMF sweep exchanges the per-gene
                    //  sums in packed order, engine.zq_gap zj_packed)
                    if (v != 0.f) atomicAdd(&Zj[((quirk & 2) ? jp : j) * K + k], v);
                }
                if (Zlog) {
                    const float v = (w_nz ? w * expectation : expectation) * ls[r];   // zigap.py:95
                    if (v != 0.f) atomicAdd(&Zlog[j * K + k], v);
                }
            }
            if (lane == 0) {
                s_cs[cbase + rec.cdst] = 0.f;
                if (sw_cs) sw_cs[cbase + rec.cdst] = 0.f;
                if (s_rs) s_rs[rbase + slot] = 0.f;
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// K = 85 .. 100 (Kp = 96 or 100): two lanes per row, 32 rows per wave, conflict-free LDS image
// ------------------------------------------------------------------------------------------
// What round 1's kernels lose at the headline K = 100 (profiles/r02_sq_pass_c4.json, tools/ubench/core_pass.hip):
//   * a row of 25 float4 puts chunk groups 4, 5 on the same 64-byte bank quarters as groups 0, 1, so any
//     schedule of the 6 ds_read_b128 of a step costs 8 LDS cycles per 16-lane service set;
//   * with four lanes per row every non-FMA instruction of a step (record broadcast, address arithmetic,
//     reduction, reciprocal, store) serves 16 entries; with two lanes per row it serves 32;
//   * the column pass restages 256 factor rows per 256 columns.
// Here a lane pair owns a row / column (48 + 2 floats per lane), a wave works on two 16-row slices of the
// same packed layout (lanes 0-31: slice 2w, lanes 32-63: slice 2w + 1), and the LDS image stores chunk
// groups 4, 5 TWICE (float4 16..23 again at 24..31 of the 512-byte row) with the tail float4 in a separate
// 4-fold array: at every step the 8 pairs of a 16-lane service set (classes 0..7) read 8 different 32-byte
// bank eighths -- steps 0..7 rotate over pair-chunks 0..7, steps 8..11 read pair-chunks 8..11 from the
// original (classes 0..3) or from the copy (classes 4..7).  The column pass runs 1024 threads over TWO
// adjacent column tiles with one image of the row block.
namespace k100 {

constexpr int T4 = 12;                  // ds_read_b128 per lane and step
constexpr int ROW4 = 32;                // float4 per LDS image row (512 bytes)
constexpr int TREP = 4;                 // copies of the tail float4 (64 bytes per image row)
constexpr int image_bytes(int TAIL) { return TILE * ROW4 * 16 + (TAIL ? TILE * TREP * 16 : 0); }

// ds_read_b128 is serviced in the 16-lane sets {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32): pairs
// {0,1,6,7,10,11,12,13} and {2,3,4,5,8,9,14,15} of a half wave get the classes 0..7 inside their set
__device__ __forceinline__ int pair_class(int lane) { const int p = (lane >> 1) & 15; return (p >> 2) * 2 + (p & 1); }

// float4 index of the chunk a lane visits at step t: inside a factor row in global memory (24 float4 + tail)
// and inside the LDS image row
__device__ __forceinline__ int gchunk(int lane, int t) {
    const int a = pair_class(lane), q = lane & 1;
    const int pc = (t < 8) ? ((a + t) & 7) : 8 + ((a + t) & 3);
    return pc * 2 + q;
}
__device__ __forceinline__ int lchunk(int lane, int t) {
    const int a = pair_class(lane), q = lane & 1;
    const int pc = (t < 8) ? ((a + t) & 7) : 8 + ((a + t) & 3) + ((a >= 4) ? 4 : 0);
    return pc * 2 + q;
}

// broadcast inside a lane pair: lane (U >> 1) of the pair holds the value
template <int U> __device__ __forceinline__ uint32_t pb_u32(uint32_t v) { return (U >> 1) ? dpp_u32<0xF5>(v) : dpp_u32<0xA0>(v); }
template <int U> __device__ __forceinline__ float pb_f32(float v) { return (U >> 1) ? dpp_f32<0xF5>(v) : dpp_f32<0xA0>(v); }

// staging of 256 factor rows (global loads before the barrier, LDS stores after it)
template <int THREADS, int TAIL>
struct Stage {
    static constexpr int KP4 = 24 + TAIL;
    static constexpr int NST = TILE * 24 / THREADS;         // 12 (512 threads) or 6 (1024)
    f4 v[NST];
    f4 t;
    __device__ __forceinline__ void load(const float *__restrict__ F, int64_t j0, int64_t jmax, int tid) {
        asm volatile("" : "+v"(tid));
        #pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int idx = tid + u * THREADS;
            const int jr = idx / 24, c4 = idx - jr * 24;
            const int64_t j = j0 + jr;
            v[u] = (j < jmax) ? reinterpret_cast<const f4 *>(F)[j * KP4 + c4] : f4{0.f, 0.f, 0.f, 0.f};
        }
        if (TAIL) {
            const int64_t j = j0 + tid;
            t = (tid < TILE && j < jmax) ? reinterpret_cast<const f4 *>(F)[j * KP4 + 24] : f4{0.f, 0.f, 0.f, 0.f};
        }
    }
    __device__ __forceinline__ void store(f4 *img, int tid) const {
        asm volatile("" : "+v"(tid));
        #pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int idx = tid + u * THREADS;
            const int jr = idx / 24, c4 = idx - jr * 24;
            img[jr * ROW4 + c4] = v[u];
            if (c4 >= 16) img[jr * ROW4 + c4 + 8] = v[u];
        }
        if (TAIL && tid < TILE) {
            f4 *tl = img + TILE * ROW4 + tid * TREP;
            #pragma unroll
            for (int r = 0; r < TREP; ++r) tl[r] = t;
        }
    }
};

// ---- row pass ------------------------------------------------------------------------------------
// Row-block item of a two-lane row kernel under an oriana_row_split: full row blocks first (one group each), then the
// parts of the split ones
struct RowItem { int64_t rb; int cb0, cb1, slab; };
__device__ __forceinline__ RowItem row_item(const oriana_counts &cm, const oriana_row_split &sp) {
    RowItem it;
    const int b = (int)blockIdx.x;
    if (b < sp.nfull) { it.rb = b; it.cb0 = 0; it.cb1 = (int)cm.ncb; it.slab = 0; return it; }
    const int idx = b - sp.nfull;
    const int blk = idx / sp.parts, part = idx - blk * sp.parts;
    it.rb = sp.nfull + blk; it.slab = part;
    if (sp.edge[0] < 0) {                        // evenly cut ranges (any number of parts)
        it.cb0 = (int)((int64_t)part * cm.ncb / sp.parts); it.cb1 = (int)(((int64_t)part + 1) * cm.ncb / sp.parts);
        return it;
    }
    it.cb0 = sp.edge[0]; it.cb1 = sp.edge[1];
    #pragma unroll
    for (int e = 1; e < 8; ++e)
        if (part == e) { it.cb0 = sp.edge[e]; it.cb1 = sp.edge[e + 1]; }
    return it;
}

template <int TAIL, int VAR>
__global__ __launch_bounds__(512) void k_row_pass_k100(oriana_counts cm, const float *__restrict__ FU,
                                                       const float *__restrict__ FV, const float *__restrict__ w_nz,
                                                       float *__restrict__ R, float *__restrict__ s_cs,
                                                       float *__restrict__ sw_cs, float *__restrict__ s_rs,
                                                       int32_t *__restrict__ tile_flag, oriana_row_split split,
                                                       const float *__restrict__ den_min_p) {
    const float den_min = den_min_p ? *den_min_p : DEN_MIN;   // (see k_row_stats: the den threshold)
    constexpr bool SROW = (VAR & 1) != 0, HASW = (VAR & 2) != 0;
    constexpr int KP = 96 + 4 * TAIL, KP4 = KP / 4;
    constexpr int PD = 3;                       // record prefetch depth (iterations)
    extern __shared__ f4 lds[];
    const float *tails = reinterpret_cast<const float *>(lds + TILE * ROW4);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane & 1, g = (lane >> 1) & 15;
    const int sl = wave * 2 + (lane >> 5);      // slice of this half wave
    const RowItem item = row_item(cm, split);   // row block, gene tiles [cb0, cb1), slab of R
    const int64_t rb = item.rb;
    const int64_t row = rb * TILE + sl * 16 + g;
    const int slot_lane = g * 4 + 2 * q;        // this lane's two records inside a 64-slot iteration
    const int toff = ((lane >> 1) & 3) * 4 + 2 * q;      // float offset inside the 4-fold tail of an image row

    int lidx[T4];
    #pragma unroll
    for (int t = 0; t < T4; ++t) lidx[t] = lchunk(lane, t);

    f4 fu[T4], acc[T4];
    f2 fut = {0.f, 0.f}, acct = {0.f, 0.f};
    #pragma unroll
    for (int t = 0; t < T4; ++t) { acc[t] = f4{0.f, 0.f, 0.f, 0.f}; fu[t] = f4{0.f, 0.f, 0.f, 0.f}; }
    if (row < cm.n) {
        #pragma unroll
        for (int t = 0; t < T4; ++t) fu[t] = reinterpret_cast<const f4 *>(FU)[row * KP4 + gchunk(lane, t)];
        if (TAIL) fut = *reinterpret_cast<const f2 *>(FU + row * KP + 96 + 2 * q);
    }
    bool rowfilled = false;                     // see k_row_pass: rows replaced by the FILL constant (sparse variants)
    if (SROW) {
        float fm = fmaxf(fut.x, fut.y);
        #pragma unroll
        for (int t = 0; t < T4; ++t) fm = fmaxf(fmaxf(fmaxf(fu[t].x, fu[t].y), fmaxf(fu[t].z, fu[t].w)), fm);
        fm = fmaxf(fm, dpp_f32<0xB1>(fm));
        rowfilled = !(fm == 1.0f);
    }

    const int64_t cb0 = item.cb0, cb1 = item.cb1;
    for (int64_t cb = cb0; cb < cb1; ++cb) {
        const int64_t t = rb * cm.ncb + cb;
        const uint32_t s0 = cm.rslice[t * 17 + sl], s1 = cm.rslice[t * 17 + sl + 1];
        const int nit = (int)((s1 - s0) >> 6);                    // iterations of this half wave's slice
        const int niter = max(__builtin_amdgcn_readlane(nit, 0), __builtin_amdgcn_readlane(nit, 32));
        const int64_t rbase = cm.roff[t] + s0 + slot_lane;
        const uint4 *recp = reinterpret_cast<const uint4 *>(reinterpret_cast<const unsigned long long *>(cm.rowrec) + rbase);
        float *sdst = s_cs + cm.coff[t];
        float *swdst = HASW ? sw_cs + cm.coff[t] : nullptr;
        const uint32_t dummy = cm.cslice[t * 17 + 16] + lane;
        bool bad = false;
        // record prefetch ring (two 8-byte records per lane and iteration), clamped to the slice's own length
        uint4 rawq[PD];
        f2 wq[PD];
        #pragma unroll
        for (int d = 0; d < PD; ++d) {
            const int id = (d < nit) ? d : nit - 1;
            rawq[d] = uint4{0u, 0u, 0u, 0u}; wq[d] = f2{1.f, 1.f};
            if (nit > 0) { rawq[d] = recp[(int64_t)id * 32]; if (HASW) wq[d] = *reinterpret_cast<const f2 *>(w_nz + rbase + (int64_t)id * 64); }
        }
#ifdef ORIANA_ABLATE_NOSTAGE
        if (cb == 0)            /* analysis build: the image is staged once (wrong results, timing only) */
#endif
        {
        Stage<512, TAIL> stg;
        stg.load(FV, cb * TILE, cm.m, tid);
        ORIANA_SYNC();
        stg.store(lds, tid);
        ORIANA_SYNC();
        }
#ifdef ORIANA_K100_STAGGER
        // analysis switch: the second wave of every SIMD (waves 4-7) starts a tile ORIANA_K100_STAGGER x 64 cycles late
        if (wave >= 4) __builtin_amdgcn_s_sleep(ORIANA_K100_STAGGER);
#endif
        for (int it = 0; it < niter; ++it) {
            const bool live = it < nit;
            uint4 cur = rawq[0];
            const f2 wcur = wq[0];
            #pragma unroll
            for (int d = 0; d + 1 < PD; ++d) { rawq[d] = rawq[d + 1]; wq[d] = wq[d + 1]; }
            const int nx = (it + PD < nit) ? it + PD : nit - 1;
            if (nit > 0) { rawq[PD - 1] = recp[(int64_t)nx * 32]; if (HASW) wq[PD - 1] = *reinterpret_cast<const f2 *>(w_nz + rbase + (int64_t)nx * 64); }
            if (!live) { cur.x = 0u; cur.z = 0u; }               // past the end of the shorter slice: padding
            f2 sbuf = {0.f, 0.f};
#define ORIANA_ROW_STEP2(U)                                                                           \
            {                                                                                         \
                const uint32_t bm = pb_u32<U>((U & 1) ? cur.w : cur.y);                               \
                const float x = __uint_as_float(pb_u32<U>((U & 1) ? cur.z : cur.x));                  \
                const int col = (int)((bm >> 16) & 0xFFu);                                            \
                const bool valid = (x != 0.f);                                                        \
                const f4 *vrow = lds + col * ROW4;                                                    \
                f4 v[T4];                                                                             \
                _Pragma("unroll") for (int tt = 0; tt < T4; ++tt) v[tt] = vrow[lidx[tt]];             \
                f2 vt = {0.f, 0.f};                                                                   \
                if (TAIL) vt = *reinterpret_cast<const f2 *>(tails + col * (TREP * 4) + toff);        \
                f2 d01 = {0.f, 0.f}, d23 = {0.f, 0.f};                                                \
                _Pragma("unroll") for (int tt = 0; tt < T4; ++tt) {                                   \
                    d01 = __builtin_elementwise_fma(fu[tt].xy, v[tt].xy, d01);                        \
                    d23 = __builtin_elementwise_fma(fu[tt].zw, v[tt].zw, d23);                        \
                }                                                                                     \
                if (TAIL) d01 = __builtin_elementwise_fma(fut, vt, d01);                              \
                const f2 dd = d01 + d23;                                                              \
                float den = dd.x + dd.y;                                                              \
                den += dpp_f32<0xB1>(den);                                                            \
                const bool ok = den >= den_min;          /* false for 0, tiny and NaN */              \
                const float s = (ok && valid) ? x * __builtin_amdgcn_rcpf(den) : 0.f;                 \
                const float sw = HASW ? s * pb_f32<U>((U & 1) ? wcur.y : wcur.x) : s;                 \
                const f2 ss = {sw, sw};                                                               \
                if (!SROW) {                  /* (with s_rs the caller only wants s: R is not formed) */ \
                    _Pragma("unroll") for (int tt = 0; tt < T4; ++tt) {                               \
                        acc[tt].xy = __builtin_elementwise_fma(ss, v[tt].xy, acc[tt].xy);             \
                        acc[tt].zw = __builtin_elementwise_fma(ss, v[tt].zw, acc[tt].zw);             \
                    }                                                                                 \
                    if (TAIL) acct = __builtin_elementwise_fma(ss, vt, acct);                         \
                }                                                                                     \
                const bool dead = SROW && !rowfilled && den == 0.f &&                                 \
                                  __float_as_uint(v[0].x) == 0x80000000u;                             \
                const bool slow = valid && !ok && !dead;                                              \
                bad = bad || slow;                                                                    \
                const float sout = slow ? NAN : s;                                                    \
                const uint32_t off = valid ? (bm & 0xFFFFu) : dummy;                                  \
                ORIANA_S_STORE(sdst, off, sout);                                                      \
                if (HASW) swdst[off] = slow ? NAN : sw;                                               \
                if (SROW && (U >> 1) == q) { if (U & 1) sbuf.y = sout; else sbuf.x = sout; }          \
                _Pragma("unroll") for (int tt = 0; tt < T4; ++tt) asm volatile("" : "+v"(acc[tt]));  \
                asm volatile("" : "+v"(cur.x), "+v"(cur.y), "+v"(cur.z), "+v"(cur.w));                \
            }
            ORIANA_ROW_STEP2(0)
            ORIANA_ROW_STEP2(1)
            ORIANA_ROW_STEP2(2)
            ORIANA_ROW_STEP2(3)
#undef ORIANA_ROW_STEP2
            if (SROW && live) *reinterpret_cast<f2 *>(s_rs + rbase + (int64_t)it * 64) = sbuf;
        }
        if (__any(bad) && lane == 0) tile_flag[t] = 1;
    }
    if (row < cm.n && !SROW) {
        // (slab p >= 1 holds the rows of the split row blocks only: stride (n - 256 nfull) rows)
        float *Rs = R + (int64_t)item.slab * (cm.n - (int64_t)split.nfull * TILE) * KP;
        #pragma unroll
        for (int t = 0; t < T4; ++t) reinterpret_cast<f4 *>(Rs)[row * KP4 + gchunk(lane, t)] = acc[t];
        if (TAIL) *reinterpret_cast<f2 *>(Rs + row * KP + 96 + 2 * q) = acct;
    }
}

// ---- the same image read by FOUR lanes per row (column pass, k_col_pass2 below) ----------------------
// (a 1024-thread group leaves 128 registers per lane: two lanes per column would keep only two of the twelve reads
// of a step in flight -- measured)
__device__ __forceinline__ int quad_class(int lane) { const int Q = lane >> 2; return ((Q & 1) << 1) | ((Q >> 1) & 1); }
__device__ __forceinline__ int gchunk4(int lane, int t) {
    const int a = quad_class(lane), q = lane & 3;
    const int cg = (t < 4) ? ((a + t) & 3) : 4 + ((a & 1) ^ (t & 1));
    return cg * 4 + q;
}
__device__ __forceinline__ int lchunk4(int lane, int t) {
    const int a = quad_class(lane), q = lane & 3;
    const int cg = (t < 4) ? ((a + t) & 3) : 4 + ((a & 1) ^ (t & 1)) + 2 * (a >> 1);
    return cg * 4 + q;
}

}  // namespace k100

// ------------------------------------------------------------------------------------------
// column pass for every Kp <= 116 (four lanes per column): TWO adjacent column tiles per image
// ------------------------------------------------------------------------------------------
// 16 waves = the 16 column slices of a tile; every wave walks its slice of the FIRST tile of the pair and then its
// slice of the SECOND one against the same image of the row block (two accumulator sets): half the staging of
// k_col_pass, and the tile barrier waits for the sum of two slices.  Work items = (pair of column tiles, row-block
// range).  Image: the duplicated-chunk-group layout of namespace k100 for Kp = 96 / 100, k_col_pass's layout
// (rows padded to 256 bytes, rotated chunk order, tail replicated in the padding) otherwise.
template <int T4, int TAIL>
struct ColImage {
    static constexpr bool DUP = (T4 == 6);
    static constexpr int KP = 16 * T4 + 4 * TAIL, KP4 = KP / 4, TOFF = 16 * T4;
    static constexpr int ROW4 = DUP ? k100::ROW4 : lds_stride_floats(KP) / 4;
    static constexpr int TREP = DUP ? k100::TREP : tail_copies(KP, TAIL);
    static constexpr int TBASE = DUP ? TILE * ROW4 * 4 : 0;             // float offset of the tail area
    static constexpr int TSTR = DUP ? k100::TREP * 4 : ROW4 * 4;         // floats between two rows' tails
    static constexpr size_t bytes() { return DUP ? (size_t)k100::image_bytes(TAIL) : (size_t)TILE * ROW4 * 16; }
    __device__ static __forceinline__ int gidx(int lane, int t) {
        return DUP ? k100::gchunk4(lane, t) : chunk_at<4, T4>(lane, lds_rot<4>(lane), t) * 4 + (lane & 3);
    }
    __device__ static __forceinline__ int lidx(int lane, int t) { return DUP ? k100::lchunk4(lane, t) : gidx(lane, t); }
    __device__ static __forceinline__ int toff(int lane) {
        return DUP ? ((lane >> 2) & 3) * 4 + (lane & 3) : TOFF + ((lane >> 2) % TREP) * 4 + (lane & 3);
    }
};

// DUAL: ONE column tile per work item and TWO images (Gm, Gm2) of the row block side by side in LDS: both products
// C += s Gm and C2 += s Gm2 from one walk over the slice's stream (the sparse models' per-gene sums and log sums,
// sparse_gap.py:96-97; Kp <= 64: two images fit).
template <int T4, int TAIL, bool DUAL>
__global__ __launch_bounds__(1024) void k_col_pass2(oriana_counts cm, const float *__restrict__ s_cs,
                                                    const float *__restrict__ Gm, float *__restrict__ C,
                                                    const int32_t *__restrict__ work, int64_t rb_per_band,
                                                    float *__restrict__ Cpart, const float *__restrict__ Gm2,
                                                    float *__restrict__ C2) {
    using Im = ColImage<T4, TAIL>;
    static_assert(!(DUAL && Im::DUP), "two duplicated images do not fit");
    constexpr int KP = Im::KP, ROW4 = Im::ROW4;
    constexpr int IMG4 = TILE * ROW4;                             // float4 per image
    constexpr int CPD = 3;
    extern __shared__ f4 lds[];
    const float *tails = reinterpret_cast<const float *>(lds) + Im::TBASE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int sl = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave = column slice of both tiles
    const int q = lane & 3;
    int64_t c2, rb0, rb1;
    if (work) {
        c2 = work[(int64_t)blockIdx.x * 3 + 0]; rb0 = work[(int64_t)blockIdx.x * 3 + 1]; rb1 = work[(int64_t)blockIdx.x * 3 + 2];
    } else {
        c2 = blockIdx.x;
        rb0 = (int64_t)blockIdx.y * rb_per_band;
        rb1 = (rb0 + rb_per_band < cm.nrb) ? rb0 + rb_per_band : cm.nrb;
    }
    const int64_t cbA = DUAL ? c2 : c2 * 2, cbB = DUAL ? c2 : c2 * 2 + 1;
    const bool hasB = !DUAL && cbB < cm.ncb;
    const int toff = Im::toff(lane);
    int lidx[T4];
    #pragma unroll
    for (int t = 0; t < T4; ++t) lidx[t] = Im::lidx(lane, t);
    f4 accA[T4], accB[T4];
    float actA = 0.f, actB = 0.f;
    #pragma unroll
    for (int t = 0; t < T4; ++t) { accA[t] = f4{0.f, 0.f, 0.f, 0.f}; accB[t] = f4{0.f, 0.f, 0.f, 0.f}; }

    // stream of one slice: (s, row index) per slot; the loads are unconditional -- an index past the slice's end is
    // clamped, and with an empty slice it reads (and discards) slots that still lie inside the tile's region, which
    // ends with 64 dummy slots
    struct Stream { const float *sb; const uint8_t *rb; int nit; float sv[CPD]; uint32_t rv[CPD]; };
    auto open_stream = [&](Stream &st, int64_t t, bool present) {
        // everything but the lane index is wave-uniform: the bases stay in scalar registers
        const uint32_t s0 = cm.cslice[t * 17 + sl];
        const uint32_t s1 = present ? cm.cslice[t * 17 + sl + 1] : s0;
        st.nit = (int)((s1 - s0) >> 6);
        const int64_t base = cm.coff[t] + s0;
        st.sb = s_cs + base;
        st.rb = cm.ridx + base;
        const int last = (st.nit > 0) ? st.nit - 1 : 0;
        #pragma unroll
        for (int d = 0; d < CPD; ++d) {
            const int id = (d < last) ? d : last;
            st.sv[d] = st.sb[id * 64 + lane];
            st.rv[d] = st.rb[id * 64 + lane];
        }
    };
#define ORIANA_COL_STEP4(ACC, ACT, U)                                                                 \
                {                                                                                     \
                    const float s = qb_f32<U>(svc);                                                   \
                    const int r = (int)qb_u32<U>(rvc);                                                \
                    const f4 *vrow = lds + r * ROW4;                                                  \
                    const f2 ss = {s, s};                                                             \
                    ORIANA_PAD_GUARD(s != 0.f) {                                                      \
                    _Pragma("unroll") for (int tt = 0; tt < T4; ++tt) {                               \
                        const f4 v = vrow[lidx[tt]];                                                  \
                        ACC[tt].xy = __builtin_elementwise_fma(ss, v.xy, ACC[tt].xy);                 \
                        ACC[tt].zw = __builtin_elementwise_fma(ss, v.zw, ACC[tt].zw);                 \
                    }                                                                                 \
                    if (TAIL) ACT = fmaf(s, tails[r * Im::TSTR + toff], ACT);                         \
                    }                                                                                 \
                    /* step fence: one step's K-vector live at a time (both accumulator sets stay in registers) */ \
                    _Pragma("unroll") for (int tt = 0; tt < T4; ++tt) asm volatile("" : "+v"(ACC[tt]));  \
                    asm volatile("" : "+v"(svc), "+v"(rvc));                                          \
                }
#define ORIANA_COL_STEP4D(U)                                                                          \
                {                                                                                     \
                    const float s = qb_f32<U>(svc);                                                   \
                    const int r = (int)qb_u32<U>(rvc);                                                \
                    const f4 *vrow = lds + r * ROW4;                                                  \
                    const f2 ss = {s, s};                                                             \
                    _Pragma("unroll") for (int tt = 0; tt < T4; ++tt) {                               \
                        const f4 v = vrow[lidx[tt]];                                                  \
                        accA[tt].xy = __builtin_elementwise_fma(ss, v.xy, accA[tt].xy);               \
                        accA[tt].zw = __builtin_elementwise_fma(ss, v.zw, accA[tt].zw);               \
                        const f4 v2 = vrow[IMG4 + lidx[tt]];                                          \
                        accB[tt].xy = __builtin_elementwise_fma(ss, v2.xy, accB[tt].xy);              \
                        accB[tt].zw = __builtin_elementwise_fma(ss, v2.zw, accB[tt].zw);              \
                    }                                                                                 \
                    if (TAIL) {                                                                       \
                        actA = fmaf(s, tails[r * Im::TSTR + toff], actA);                             \
                        actB = fmaf(s, tails[IMG4 * 4 + r * Im::TSTR + toff], actB);                  \
                    }                                                                                 \
                    _Pragma("unroll") for (int tt = 0; tt < T4; ++tt) asm volatile("" : "+v"(accA[tt]), "+v"(accB[tt]));  \
                    asm volatile("" : "+v"(svc), "+v"(rvc));                                          \
                }
#define ORIANA_COL_RUN4D(ST)                                                                          \
            {                                                                                         \
                const int last = (ST.nit > 0) ? ST.nit - 1 : 0;                                       \
                for (int it = 0; it < ST.nit; ++it) {                                                 \
                    float svc = ST.sv[0]; uint32_t rvc = ST.rv[0];                                    \
                    _Pragma("unroll") for (int d = 0; d + 1 < CPD; ++d) { ST.sv[d] = ST.sv[d + 1]; ST.rv[d] = ST.rv[d + 1]; } \
                    const int nx = (it + CPD < last) ? it + CPD : last;                               \
                    ST.sv[CPD - 1] = ST.sb[nx * 64 + lane];                                           \
                    ST.rv[CPD - 1] = ST.rb[nx * 64 + lane];                                           \
                    ORIANA_COL_STEP4D(0)                                                              \
                    ORIANA_COL_STEP4D(1)                                                              \
                    ORIANA_COL_STEP4D(2)                                                              \
                    ORIANA_COL_STEP4D(3)                                                              \
                }                                                                                     \
            }
#define ORIANA_COL_RUN4(ST, ACC, ACT)                                                                 \
            {                                                                                         \
                const int last = (ST.nit > 0) ? ST.nit - 1 : 0;                                       \
                for (int it = 0; it < ST.nit; ++it) {                                                 \
                    float svc = ST.sv[0]; uint32_t rvc = ST.rv[0];                                    \
                    _Pragma("unroll") for (int d = 0; d + 1 < CPD; ++d) { ST.sv[d] = ST.sv[d + 1]; ST.rv[d] = ST.rv[d + 1]; } \
                    const int nx = (it + CPD < last) ? it + CPD : last;                               \
                    ST.sv[CPD - 1] = ST.sb[nx * 64 + lane];                                           \
                    ST.rv[CPD - 1] = ST.rb[nx * 64 + lane];                                           \
                    ORIANA_COL_STEP4(ACC, ACT, 0)                                                     \
                    ORIANA_COL_STEP4(ACC, ACT, 1)                                                     \
                    ORIANA_COL_STEP4(ACC, ACT, 2)                                                     \
                    ORIANA_COL_STEP4(ACC, ACT, 3)                                                     \
                }                                                                                     \
            }
    Stream stA, stB, stN;
    if (rb0 < rb1) open_stream(stA, rb0 * cm.ncb + cbA, true);
    for (int64_t rb = rb0; rb < rb1; ++rb) {
        if (!DUAL) open_stream(stB, rb * cm.ncb + (hasB ? cbB : cbA), hasB);      // in flight during the first tile's loop
        if (Im::DUP) {
            k100::Stage<1024, TAIL> stg;
            stg.load(Gm, rb * TILE, cm.n, tid);
            ORIANA_SYNC();
            stg.store(lds, tid);
        } else {
            {
                Stage<Im::KP4, Im::TREP, TILE> stg;
                stg.load(Gm, rb * TILE, cm.n, tid);
                ORIANA_SYNC();
                stg.template store<ROW4>(lds, tid);
            }
            if (DUAL) {
                Stage<Im::KP4, Im::TREP, TILE> stg2;
                stg2.load(Gm2, rb * TILE, cm.n, tid);
                stg2.template store<ROW4>(lds + IMG4, tid);
            }
        }
        ORIANA_SYNC();
        if (DUAL) {
            if (rb + 1 < rb1) open_stream(stN, (rb + 1) * cm.ncb + cbA, true);   // the next row block's stream, in flight
            ORIANA_COL_RUN4D(stA)
        } else {
            ORIANA_COL_RUN4(stA, accA, actA)
            // the next row block's first stream is requested before the second tile's loop: its (HBM) latency is
            // hidden behind that loop instead of being paid between two images
            if (rb + 1 < rb1) open_stream(stN, (rb + 1) * cm.ncb + cbA, true);
            ORIANA_COL_RUN4(stB, accB, actB)
        }
        stA = stN;
    }
#undef ORIANA_COL_RUN4D
#undef ORIANA_COL_STEP4D
#undef ORIANA_COL_RUN4
#undef ORIANA_COL_STEP4
    const int cl = sl * 16 + (lane >> 2);
    const bool plain = Cpart != nullptr;
    float *ldsf = reinterpret_cast<float *>(lds);
    #pragma unroll
    for (int h = 0; h < 2; ++h) {
        // one tile at a time through LDS (256 x Kp floats), then a contiguous flush (see flush_block)
        const int64_t c0 = (h ? cbB : cbA) * TILE;
        ORIANA_SYNC();
        if ((h == 0 || hasB || DUAL) && c0 + cl < cm.m) {
            float *row = ldsf + cl * KP;
            #pragma unroll
            for (int t = 0; t < T4; ++t) *reinterpret_cast<f4 *>(row + Im::gidx(lane, t) * 4) = h ? accB[t] : accA[t];
            if (TAIL) row[Im::TOFF + q] = h ? actB : actA;
        }
        ORIANA_SYNC();
        if (h == 0 || hasB || DUAL) {
            const int64_t left = cm.m - c0;
            const int ncols = left < TILE ? (left > 0 ? (int)left : 0) : TILE;
            float *dst = plain ? Cpart + ((int64_t)blockIdx.x * 2 * TILE + h * TILE) * KP : ((DUAL && h) ? C2 : C) + c0 * KP;
            flush_block<1024>(ldsf, dst, ncols * KP, plain, tid);
        }
    }
}


// ==========================================================================================
// [r4] 33 <= Kp <= 64 (BASELINE configs[2] has K = 50, configs[4] K = 64): TWO LANES PER ROW, as namespace k100.
// The four-lane kernels above cost ~24 cycles per non-zero whatever K <= 100 (18 ps per non-zero at K = 50 against 19 at
// K = 100): below Kp = 64 their per-step overhead -- record decode, DPP broadcasts, the lane sum, the reciprocal, the store
// of s -- exceeds their FMAs.  Here a lane holds 32 of the (zero-padded) 64 floats of its row, so a step of a wave covers
// 32 slots with 8 ds_read_b128 + 32 packed FMAs per lane; the image row is exactly the 256 bytes of the LDS bank row, the
// eight lane pairs of a 16-lane service set rotate over its eight 32-byte eighths (k100::pair_class): conflict-free with no
// duplicated chunk.  Kp = 36, 48, 52 run as 64 with zero padding (K = 50: 23 % padded FMAs, still cheaper than four lanes).
// Variants: VAR bit 0 s in row-side slots, bit 1 per-entry weights, bit 2 SECOND image FV2 = FV * S_hat for the
// accumulation (the sparse models' S_hat-weighted row sums, sparse_gap.py:95); column side: two column tiles per image,
// or DUAL (two images, both per-gene sums of the sparse models from one walk, sparse_gap.py:96-97).
// ==========================================================================================
namespace k64 {

constexpr int T4 = 8;                   // ds_read_b128 per lane and step
constexpr int ROW4 = 16;                // float4 per LDS image row (256 bytes)
constexpr int IMG4 = TILE * ROW4;       // float4 per image (64 KB)

__device__ __forceinline__ int chunk(int lane, int t) { return ((k100::pair_class(lane) + t) & 7) * 2 + (lane & 1); }

// staging of 256 factor rows of KP4 float4, zero-padded to 16 (global loads before the barrier, LDS stores after it)
template <int THREADS, int KP4>
struct Stage {
    static constexpr int NST = IMG4 / THREADS;             // 8 (512 threads)
    f4 v[NST];
    __device__ __forceinline__ void load(const float *__restrict__ F, int64_t j0, int64_t jmax, int tid) {
        asm volatile("" : "+v"(tid));
        #pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int idx = tid + u * THREADS;
            const int jr = idx >> 4, c4 = idx & 15;
            const int64_t j = j0 + jr;
            v[u] = (c4 < KP4 && j < jmax) ? reinterpret_cast<const f4 *>(F)[j * KP4 + c4] : f4{0.f, 0.f, 0.f, 0.f};
        }
    }
    __device__ __forceinline__ void store(f4 *img, int tid) const {
        asm volatile("" : "+v"(tid));
        #pragma unroll
        for (int u = 0; u < NST; ++u) img[tid + u * THREADS] = v[u];
    }
};

template <int KP4, int VAR>
__global__ __launch_bounds__(512) void k_row_pass_k64(oriana_counts cm, const float *__restrict__ FU,
                                                      const float *__restrict__ FV, const float *__restrict__ w_nz,
                                                      float *__restrict__ R, float *__restrict__ s_cs,
                                                      float *__restrict__ sw_cs, float *__restrict__ s_rs,
                                                      int32_t *__restrict__ tile_flag, const float *__restrict__ FV2,
                                                      oriana_row_split split, const float *__restrict__ den_min_p) {
    const float den_min = den_min_p ? *den_min_p : DEN_MIN;   // (see k_row_stats: the den threshold)
    constexpr bool F2I = (VAR & 4) != 0;
    constexpr bool SPARSE = (VAR & 5) != 0, SROW = (VAR & 1) != 0 && !F2I, HASW = (VAR & 2) != 0;
    constexpr int KP = 4 * KP4;
    constexpr int PD = 3;                       // record prefetch depth (iterations)
    extern __shared__ f4 lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane & 1, g = (lane >> 1) & 15;
    const int sl = wave * 2 + (lane >> 5);      // slice of this half wave
    const k100::RowItem item = k100::row_item(cm, split);   // row block, gene tiles [cb0, cb1), slab of R
    const int64_t rb = item.rb;
    const int64_t row = rb * TILE + sl * 16 + g;
    const int slot_lane = g * 4 + 2 * q;        // this lane's two records inside a 64-slot iteration

    int lidx[T4];
    #pragma unroll
    for (int t = 0; t < T4; ++t) lidx[t] = chunk(lane, t);

    f4 fu[T4], acc[T4];
    #pragma unroll
    for (int t = 0; t < T4; ++t) { acc[t] = f4{0.f, 0.f, 0.f, 0.f}; fu[t] = f4{0.f, 0.f, 0.f, 0.f}; }
    if (row < cm.n) {
        #pragma unroll
        for (int t = 0; t < T4; ++t)
            if (lidx[t] < KP4) fu[t] = reinterpret_cast<const f4 *>(FU)[row * KP4 + lidx[t]];
    }
    bool rowfilled = false;                     // see k_row_pass: rows replaced by the FILL constant (sparse variants)
    if (SPARSE) {
        float fm = 0.f;
        #pragma unroll
        for (int t = 0; t < T4; ++t) fm = fmaxf(fmaxf(fmaxf(fu[t].x, fu[t].y), fmaxf(fu[t].z, fu[t].w)), fm);
        fm = fmaxf(fm, dpp_f32<0xB1>(fm));
        rowfilled = !(fm == 1.0f);
    }

    const int64_t cb0 = item.cb0, cb1 = item.cb1;
    for (int64_t cb = cb0; cb < cb1; ++cb) {
        const int64_t t = rb * cm.ncb + cb;
        const uint32_t s0 = cm.rslice[t * 17 + sl], s1 = cm.rslice[t * 17 + sl + 1];
        const int nit = (int)((s1 - s0) >> 6);                    // iterations of this half wave's slice
        const int niter = max(__builtin_amdgcn_readlane(nit, 0), __builtin_amdgcn_readlane(nit, 32));
        const int64_t rbase = cm.roff[t] + s0 + slot_lane;
        const uint4 *recp = reinterpret_cast<const uint4 *>(reinterpret_cast<const unsigned long long *>(cm.rowrec) + rbase);
        float *sdst = s_cs + cm.coff[t];
        float *swdst = HASW ? sw_cs + cm.coff[t] : nullptr;
        const uint32_t dummy = cm.cslice[t * 17 + 16] + lane;
        bool bad = false;
        // record prefetch ring (two 8-byte records per lane and iteration), clamped to the slice's own length
        uint4 rawq[PD];
        f2 wq[PD];
        #pragma unroll
        for (int d = 0; d < PD; ++d) {
            const int id = (d < nit) ? d : nit - 1;
            rawq[d] = uint4{0u, 0u, 0u, 0u}; wq[d] = f2{1.f, 1.f};
            if (nit > 0) { rawq[d] = recp[(int64_t)id * 32]; if (HASW) wq[d] = *reinterpret_cast<const f2 *>(w_nz + rbase + (int64_t)id * 64); }
        }
        {
            Stage<512, KP4> stg;
            stg.load(FV, cb * TILE, cm.m, tid);
            ORIANA_SYNC();
            stg.store(lds, tid);
        }
        if (F2I) {
            Stage<512, KP4> stg2;
            stg2.load(FV2, cb * TILE, cm.m, tid);
            stg2.store(lds + IMG4, tid);
        }
        ORIANA_SYNC();
        // the K-vector of a step is read from LDS one step AHEAD (v: this step, vn: the next one): with two waves per SIMD
        // and only 32 packed FMAs per step the read latency would otherwise be exposed at every step
        f4 v[T4];
        {
            const int col0 = (int)((k100::pb_u32<0>(rawq[0].y) >> 16) & 0xFFu);
            #pragma unroll
            for (int tt = 0; tt < T4; ++tt) v[tt] = (lds + col0 * ROW4)[lidx[tt]];
        }
        for (int it = 0; it < niter; ++it) {
            const bool live = it < nit;
            uint4 cur = rawq[0];
            const f2 wcur = wq[0];
            #pragma unroll
            for (int d = 0; d + 1 < PD; ++d) { rawq[d] = rawq[d + 1]; wq[d] = wq[d + 1]; }
            const int nx = (it + PD < nit) ? it + PD : nit - 1;
            if (nit > 0) { rawq[PD - 1] = recp[(int64_t)nx * 32]; if (HASW) wq[PD - 1] = *reinterpret_cast<const f2 *>(w_nz + rbase + (int64_t)nx * 64); }
            if (!live) { cur.x = 0u; cur.z = 0u; }               // past the end of the shorter slice: padding
            f2 sbuf = {0.f, 0.f};
#define ORIANA_ROW_STEP64(U)                                                                          \
            {                                                                                         \
                const uint32_t bm = k100::pb_u32<U>((U & 1) ? cur.w : cur.y);                         \
                const float x = __uint_as_float(k100::pb_u32<U>((U & 1) ? cur.z : cur.x));            \
                const int col = (int)((bm >> 16) & 0xFFu);                                            \
                const bool valid = (x != 0.f);                                                        \
                const f4 *vrow = lds + col * ROW4;                                                    \
                /* the next step's row (step 0 of the next iteration after step 3) */                 \
                const uint32_t bmn = (U == 3) ? k100::pb_u32<0>(rawq[0].y)                            \
                                              : k100::pb_u32<(U + 1) & 3>(((U + 1) & 1) ? cur.w : cur.y); \
                const f4 *vrown = lds + (int)((bmn >> 16) & 0xFFu) * ROW4;                            \
                f4 vn[T4];                                                                            \
                _Pragma("unroll") for (int tt = 0; tt < T4; ++tt) vn[tt] = vrown[lidx[tt]];           \
                f2 d01 = {0.f, 0.f}, d23 = {0.f, 0.f};                                                \
                _Pragma("unroll") for (int tt = 0; tt < T4; ++tt) {                                   \
                    d01 = __builtin_elementwise_fma(fu[tt].xy, v[tt].xy, d01);                        \
                    d23 = __builtin_elementwise_fma(fu[tt].zw, v[tt].zw, d23);                        \
                }                                                                                     \
                const f2 dd = d01 + d23;                                                              \
                float den = dd.x + dd.y;                                                              \
                den += dpp_f32<0xB1>(den);                                                            \
                const bool ok = den >= den_min;          /* false for 0, tiny and NaN */              \
                const float s = (ok && valid) ? x * __builtin_amdgcn_rcpf(den) : 0.f;                 \
                const float sw = HASW ? s * k100::pb_f32<U>((U & 1) ? wcur.y : wcur.x) : s;           \
                const f2 ss = {sw, sw};                                                               \
                if (F2I) {                    /* accumulate against the second image */              \
                    const f4 *vrow2 = vrow + IMG4;                                                    \
                    _Pragma("unroll") for (int tt = 0; tt < T4; ++tt) {                               \
                        const f4 v2 = vrow2[lidx[tt]];                                                \
                        acc[tt].xy = __builtin_elementwise_fma(ss, v2.xy, acc[tt].xy);                \
                        acc[tt].zw = __builtin_elementwise_fma(ss, v2.zw, acc[tt].zw);                \
                    }                                                                                 \
                } else if (!SROW) {           /* (with s_rs the caller only wants s: R is not formed) */ \
                    _Pragma("unroll") for (int tt = 0; tt < T4; ++tt) {                               \
                        acc[tt].xy = __builtin_elementwise_fma(ss, v[tt].xy, acc[tt].xy);             \
                        acc[tt].zw = __builtin_elementwise_fma(ss, v[tt].zw, acc[tt].zw);             \
                    }                                                                                 \
                }                                                                                     \
                /* fully masked gene (a row of -0.0): exactly zero contribution, no slow path.  Steps 0 and 4 read pair  */ \
                /* chunks four apart: one of them lies in the unpadded part of the row (chunks 0..7 < KP4), for both lanes */ \
                const bool neg0 = __float_as_uint(v[0].x) == 0x80000000u || __float_as_uint(v[4].x) == 0x80000000u; \
                const bool dead = SPARSE && !rowfilled && den == 0.f && neg0;                         \
                const bool slow = valid && !ok && !dead;                                              \
                bad = bad || slow;                                                                    \
                const float sout = slow ? NAN : s;                                                    \
                const uint32_t off = valid ? (bm & 0xFFFFu) : dummy;                                  \
                ORIANA_S_STORE(sdst, off, sout);                                                      \
                if (HASW) swdst[off] = slow ? NAN : sw;                                               \
                if (SROW && (U >> 1) == q) { if (U & 1) sbuf.y = sout; else sbuf.x = sout; }          \
                _Pragma("unroll") for (int tt = 0; tt < T4; ++tt) v[tt] = vn[tt];                     \
                _Pragma("unroll") for (int tt = 0; tt < T4; ++tt) asm volatile("" : "+v"(acc[tt]));  \
                asm volatile("" : "+v"(cur.x), "+v"(cur.y), "+v"(cur.z), "+v"(cur.w));                \
            }
            ORIANA_ROW_STEP64(0)
            ORIANA_ROW_STEP64(1)
            ORIANA_ROW_STEP64(2)
            ORIANA_ROW_STEP64(3)
#undef ORIANA_ROW_STEP64
            if (SROW && live) *reinterpret_cast<f2 *>(s_rs + rbase + (int64_t)it * 64) = sbuf;
        }
        if (__any(bad) && lane == 0) tile_flag[t] = 1;
    }
    if (row < cm.n && !SROW) {
        // (slab p >= 1 holds the rows of the split row blocks only: stride (n - 256 nfull) rows)
        float *Rs = R + (int64_t)item.slab * (cm.n - (int64_t)split.nfull * TILE) * KP;
        #pragma unroll
        for (int t = 0; t < T4; ++t)
            if (lidx[t] < KP4) reinterpret_cast<f4 *>(Rs)[row * KP4 + lidx[t]] = acc[t];
    }
}

// ---- column pass: C += s G over the column-side stream, two lanes per column ---------------------------------------
// 1024 threads: waves 0-7 take the FIRST column tile of the pair (two 16-column slices each), waves 8-15 the SECOND one,
// against the same image of the row block; DUAL: both wave sets walk the SAME tile, the first against image Gm into C, the
// second against image Gm2 into C2 (the sparse models' per-gene sums and log sums, sparse_gap.py:96-97).  One accumulator
// set per lane (32 floats) keeps the kernel under 128 registers, i.e. four waves per SIMD: a step is 8 ds_read_b128 and
// 16 packed FMAs per lane, too short to cover the LDS latency with two.  Work items as k_col_pass2.
template <int KP4, bool DUAL>
__global__ __launch_bounds__(1024) void k_col_pass_k64(oriana_counts cm, const float *__restrict__ s_cs,
                                                       const float *__restrict__ Gm, float *__restrict__ C,
                                                       const int32_t *__restrict__ work, int64_t rb_per_band,
                                                       float *__restrict__ Cpart, const float *__restrict__ Gm2,
                                                       float *__restrict__ C2) {
    constexpr int KP = 4 * KP4;
    constexpr int CPD = 3;
    extern __shared__ f4 lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = __builtin_amdgcn_readfirstlane(wave >> 3);          // 0: first tile / image, 1: second
    const int q = lane & 1, g = (lane >> 1) & 15;
    const int sl = (wave & 7) * 2 + (lane >> 5);                        // column slice of this half wave
    int64_t c2, rb0, rb1;
    if (work) {
        c2 = work[(int64_t)blockIdx.x * 3 + 0]; rb0 = work[(int64_t)blockIdx.x * 3 + 1]; rb1 = work[(int64_t)blockIdx.x * 3 + 2];
    } else {
        c2 = blockIdx.x;
        rb0 = (int64_t)blockIdx.y * rb_per_band;
        rb1 = (rb0 + rb_per_band < cm.nrb) ? rb0 + rb_per_band : cm.nrb;
    }
    const int64_t cbA = DUAL ? c2 : c2 * 2, cbB = DUAL ? c2 : c2 * 2 + 1;
    const bool hasB = !DUAL && cbB < cm.ncb;
    const int64_t cb = half ? cbB : cbA;                                 // this wave's column tile
    const bool present = half == 0 || DUAL || hasB;
    const f4 *img = lds + ((DUAL && half) ? IMG4 : 0);
    int lidx[T4];
    #pragma unroll
    for (int t = 0; t < T4; ++t) lidx[t] = chunk(lane, t);
    f4 acc[T4];
    #pragma unroll
    for (int t = 0; t < T4; ++t) acc[t] = f4{0.f, 0.f, 0.f, 0.f};

    // stream of one slice: two (s, row index) slots per lane and iteration; loads are unconditional -- an index past the
    // slice's end is clamped (an empty slice reads, and discards, slots inside the tile's region, which ends with 64 dummy
    // slots), and its values are replaced by s = 0, row 0
    struct Stream { const float *sb; const uint8_t *rbp; int nit, niter; f2 sv[CPD]; uint32_t rv[CPD]; };
    auto open_stream = [&](Stream &st, int64_t t) {
        const int64_t tt = present ? t : (t - cb + cbA);                 // (a missing second tile: harmless reads of the first)
        const uint32_t s0 = cm.cslice[tt * 17 + sl];
        const uint32_t s1 = present ? cm.cslice[tt * 17 + sl + 1] : s0;
        st.nit = (int)((s1 - s0) >> 6);
        st.niter = max(__builtin_amdgcn_readlane(st.nit, 0), __builtin_amdgcn_readlane(st.nit, 32));
        const int64_t base = cm.coff[tt] + s0 + g * 4 + 2 * q;
        st.sb = s_cs + base;
        st.rbp = cm.ridx + base;
        const int last = (st.nit > 0) ? st.nit - 1 : 0;
        #pragma unroll
        for (int d = 0; d < CPD; ++d) {
            const int id = (d < last) ? d : last;
            st.sv[d] = *reinterpret_cast<const f2 *>(st.sb + id * 64);
            st.rv[d] = *reinterpret_cast<const uint16_t *>(st.rbp + id * 64);
        }
    };
#define ORIANA_COL_STEP64(U)                                                                          \
                {                                                                                     \
                    const float s = k100::pb_f32<U>((U & 1) ? svc.y : svc.x);                         \
                    const int r = (int)((k100::pb_u32<U>(rvc) >> ((U & 1) * 8)) & 0xFFu);             \
                    const f4 *vrow = img + r * ROW4;                                                  \
                    const f2 ss = {s, s};                                                             \
                    _Pragma("unroll") for (int tt = 0; tt < T4; ++tt) {                               \
                        const f4 v = vrow[lidx[tt]];                                                  \
                        acc[tt].xy = __builtin_elementwise_fma(ss, v.xy, acc[tt].xy);                 \
                        acc[tt].zw = __builtin_elementwise_fma(ss, v.zw, acc[tt].zw);                 \
                    }                                                                                 \
                    _Pragma("unroll") for (int tt = 0; tt < T4; ++tt) asm volatile("" : "+v"(acc[tt]));  \
                    asm volatile("" : "+v"(svc), "+v"(rvc));                                          \
                }
    Stream stA, stN;
    if (rb0 < rb1) open_stream(stA, rb0 * cm.ncb + cb);
    for (int64_t rb = rb0; rb < rb1; ++rb) {
        {
            Stage<1024, KP4> stg;
            stg.load(Gm, rb * TILE, cm.n, tid);
            ORIANA_SYNC();
            stg.store(lds, tid);
        }
        if (DUAL) {
            Stage<1024, KP4> stg2;
            stg2.load(Gm2, rb * TILE, cm.n, tid);
            stg2.store(lds + IMG4, tid);
        }
        ORIANA_SYNC();
        if (rb + 1 < rb1) open_stream(stN, (rb + 1) * cm.ncb + cb);     // the next row block's stream, in flight
        {
            const int last = (stA.nit > 0) ? stA.nit - 1 : 0;
            for (int it = 0; it < stA.niter; ++it) {
                f2 svc = stA.sv[0]; uint32_t rvc = stA.rv[0];
                if (it >= stA.nit) { svc = f2{0.f, 0.f}; rvc = 0u; }
                #pragma unroll
                for (int d = 0; d + 1 < CPD; ++d) { stA.sv[d] = stA.sv[d + 1]; stA.rv[d] = stA.rv[d + 1]; }
                const int nx = (it + CPD < last) ? it + CPD : last;
                stA.sv[CPD - 1] = *reinterpret_cast<const f2 *>(stA.sb + nx * 64);
                stA.rv[CPD - 1] = *reinterpret_cast<const uint16_t *>(stA.rbp + nx * 64);
                ORIANA_COL_STEP64(0)
                ORIANA_COL_STEP64(1)
                ORIANA_COL_STEP64(2)
                ORIANA_COL_STEP64(3)
            }
        }
        stA = stN;
    }
#undef ORIANA_COL_STEP64
    const int cl = sl * 16 + g;
    const bool plain = Cpart != nullptr;
    float *ldsf = reinterpret_cast<float *>(lds);
    #pragma unroll
    for (int h = 0; h < 2; ++h) {
        // one tile at a time through LDS (256 x Kp floats), then a contiguous flush (see flush_block)
        const int64_t c0 = (h ? cbB : cbA) * TILE;
        ORIANA_SYNC();
        if (half == h && present && c0 + cl < cm.m) {
            float *rowp = ldsf + cl * KP;
            #pragma unroll
            for (int t = 0; t < T4; ++t)
                if (lidx[t] < KP4) *reinterpret_cast<f4 *>(rowp + lidx[t] * 4) = acc[t];
        }
        ORIANA_SYNC();
        if (h == 0 || hasB || DUAL) {
            const int64_t left = cm.m - c0;
            const int ncols = left < TILE ? (left > 0 ? (int)left : 0) : TILE;
            float *dst = plain ? Cpart + ((int64_t)blockIdx.x * 2 * TILE + h * TILE) * KP : ((DUAL && h) ? C2 : C) + c0 * KP;
            flush_block<1024>(ldsf, dst, ncols * KP, plain, tid);
        }
    }
}

}  // namespace k64


// ==========================================================================================
// Narrow factor rows (Kp <= 32, i.e. K <= 32 -- configs[1] has K = 20): ONE LANE PER ROW.
// The kernels above give a matrix row 4 lanes, each holding Kp / 4 factors: a step of a wave covers 16 slots, and for
// K = 20 only 5 of its ~50 instructions are FMAs -- the rest (record decode, DPP broadcasts, the 4-lane sum, the
// reciprocal, the store) is per STEP, whatever K is.  Here a lane owns a whole row of the row block (a whole gene of
// the column tile): it keeps the Kp factors and the Kp accumulators in registers, reads a whole factor row of the
// other side from LDS per slot and needs no cross-lane traffic at all; a step of a wave covers 64 slots.  A wave takes
// four 16-row slices of the sliced layout at once (lanes 16a .. 16a+15 = slice 4w + a), a work-group of 256 threads a
// row block (a column tile).  The LDS image has an ODD row stride in 16-byte units, so that the lanes' reads of
// random rows spread over the banks.
// ==========================================================================================
namespace narrow {

template <int KP>
struct Geo {
    static constexpr int KP4 = KP / 4;
    static constexpr int ST4 = KP4 | 1;                  // image row stride in float4 (odd)
    static constexpr size_t bytes() { return (size_t)TILE * ST4 * sizeof(f4); }
};

// 256 rows x KP4 float4 of F (rows beyond `rows_total` read as 0), as registers of 256 threads
template <int KP>
struct Image {
    static constexpr int KP4 = Geo<KP>::KP4, ST4 = Geo<KP>::ST4;
    f4 v[KP4];
    __device__ __forceinline__ void load(const float *__restrict__ F, int64_t row0, int64_t rows_total, int tid) {
        #pragma unroll
        for (int j = 0; j < KP4; ++j) {
            const int id = tid + j * 256;                // chunk id inside the tile: row = id / KP4
            const int r = id / KP4;
            v[j] = (row0 + r < rows_total) ? reinterpret_cast<const f4 *>(F)[row0 * KP4 + id] : f4{0.f, 0.f, 0.f, 0.f};
        }
    }
    __device__ __forceinline__ void store(f4 *lds, int tid) const {
        #pragma unroll
        for (int j = 0; j < KP4; ++j) {
            const int id = tid + j * 256;
            const int r = id / KP4, c = id - r * KP4;
            lds[r * ST4 + c] = v[j];
        }
    }
};

__device__ __forceinline__ int wave_max_i(int v) {
    for (int o = 32; o > 0; o >>= 1) { const int w = __shfl_xor(v, o, 64); v = w > v ? w : v; }
    return v;
}

// row pass (plain variant: no weights, no row-side copy of s); gridDim.y = gene-tile splits (slabs of R)
template <int KP>
__global__ __launch_bounds__(256) void k_row_pass_narrow(oriana_counts cm, const float *__restrict__ FU,
                                                         const float *__restrict__ FV, float *__restrict__ R,
                                                         float *__restrict__ s_cs, int32_t *__restrict__ tile_flag,
                                                         const float *__restrict__ den_min_p) {
    const float den_min = den_min_p ? *den_min_p : DEN_MIN;   // (see k_row_stats: the den threshold)
    constexpr int KP4 = Geo<KP>::KP4, ST4 = Geo<KP>::ST4;
    constexpr int PD = 3;                                // record prefetch depth (iterations)
    extern __shared__ f4 lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sl = wave * 4 + (lane >> 4), g = lane & 15;
    const int64_t rb = blockIdx.x;
    const int64_t row = rb * TILE + sl * 16 + g;
    f4 fu[KP4], acc[KP4];
    #pragma unroll
    for (int c = 0; c < KP4; ++c) {
        acc[c] = f4{0.f, 0.f, 0.f, 0.f};
        fu[c] = (row < cm.n) ? reinterpret_cast<const f4 *>(FU)[row * KP4 + c] : f4{0.f, 0.f, 0.f, 0.f};
    }
    const int64_t cb0 = (int64_t)blockIdx.y * cm.ncb / gridDim.y, cb1 = ((int64_t)blockIdx.y + 1) * cm.ncb / gridDim.y;
    Image<KP> img;
    if (cb0 < cb1) img.load(FV, cb0 * TILE, cm.m, tid);
    for (int64_t cb = cb0; cb < cb1; ++cb) {
        const int64_t t = rb * cm.ncb + cb;
        const uint32_t s0 = cm.rslice[t * 17 + sl], s1 = cm.rslice[t * 17 + sl + 1];
        const int niter = (int)((s1 - s0) >> 6);                           // of this lane's slice
        const int nwave = __builtin_amdgcn_readfirstlane(wave_max_i(niter));
        // this lane's four records of iteration 0 (slots 4g .. 4g+3 of the slice's 64-slot iterations)
        const uint4 *recp = reinterpret_cast<const uint4 *>(reinterpret_cast<const unsigned long long *>(cm.rowrec) +
                                                            cm.roff[t] + s0 + g * 4);
        float *sdst = s_cs + cm.coff[t];
        const uint32_t dummy = cm.cslice[t * 17 + 16] + lane;              // write-only slot of the tile
        uint4 qa[PD], qb[PD];
        #pragma unroll
        for (int d = 0; d < PD; ++d) {
            qa[d] = uint4{0u, 0u, 0u, 0u}; qb[d] = uint4{0u, 0u, 0u, 0u};
            if (d < niter) { qa[d] = recp[(int64_t)d * 32]; qb[d] = recp[(int64_t)d * 32 + 1]; }
        }
        ORIANA_SYNC();                                                     // everybody is done with the previous image
        img.store(lds, tid);
        ORIANA_SYNC();
        if (cb + 1 < cb1) img.load(FV, (cb + 1) * TILE, cm.m, tid);         // the next image: in flight during the loop
        bool bad = false;
        // The factor row of a slot is read from LDS ONE STEP AHEAD of its use (vn while v is consumed): a short matrix
        // gives a SIMD a single wave, whose time is the chain LDS read -> dot product -> reciprocal -> accumulate.
        f4 v[KP4];
        {
            const f4 *vrow = lds + ((qa[0].y >> 16) & 0xFFu) * ST4;
            #pragma unroll
            for (int c = 0; c < KP4; ++c) v[c] = vrow[c];
        }
        for (int it = 0; it < nwave; ++it) {
            const uint4 ra = qa[0], rbq = qb[0];
            #pragma unroll
            for (int d = 0; d + 1 < PD; ++d) { qa[d] = qa[d + 1]; qb[d] = qb[d + 1]; }
            qa[PD - 1] = uint4{0u, 0u, 0u, 0u}; qb[PD - 1] = uint4{0u, 0u, 0u, 0u};
            if (it + PD < niter) { qa[PD - 1] = recp[(int64_t)(it + PD) * 32]; qb[PD - 1] = recp[(int64_t)(it + PD) * 32 + 1]; }
            // (a lane past the end of its own slice holds zero records: x = 0 = padding, image row 0)
#define ORIANA_NROW_STEP(XB, BM, BMNEXT)                                                              \
            {                                                                                         \
                f4 vn[KP4];                                                                           \
                {                                                                                     \
                    const f4 *nrow = lds + (((BMNEXT) >> 16) & 0xFFu) * ST4;                          \
                    _Pragma("unroll") for (int c = 0; c < KP4; ++c) vn[c] = nrow[c];                  \
                }                                                                                     \
                __builtin_amdgcn_sched_barrier(0);                                                    \
                const float x = __uint_as_float(XB);                                                  \
                const uint32_t bm = (BM);                                                             \
                const bool valid = (x != 0.f);                                                        \
                f2 d01 = {0.f, 0.f}, d23 = {0.f, 0.f};                                                \
                _Pragma("unroll") for (int c = 0; c < KP4; ++c) {                                     \
                    d01 = __builtin_elementwise_fma(fu[c].xy, v[c].xy, d01);                          \
                    d23 = __builtin_elementwise_fma(fu[c].zw, v[c].zw, d23);                          \
                }                                                                                     \
                const f2 dd = d01 + d23;                                                              \
                const float den = dd.x + dd.y;                                                        \
                const bool ok = den >= den_min;              /* false for 0, tiny and NaN */          \
                const float s = (ok && valid) ? x * __builtin_amdgcn_rcpf(den) : 0.f;                 \
                const f2 ss = {s, s};                                                                 \
                _Pragma("unroll") for (int c = 0; c < KP4; ++c) {                                     \
                    acc[c].xy = __builtin_elementwise_fma(ss, v[c].xy, acc[c].xy);                    \
                    acc[c].zw = __builtin_elementwise_fma(ss, v[c].zw, acc[c].zw);                    \
                }                                                                                     \
                const bool slow = valid && !ok;              /* NaN = "evaluate me exactly" */        \
                bad = bad || slow;                                                                    \
                ORIANA_S_STORE(sdst, valid ? (bm & 0xFFFFu) : dummy, slow ? NAN : s);                 \
                _Pragma("unroll") for (int c = 0; c < KP4; ++c) v[c] = vn[c];                         \
            }
            ORIANA_NROW_STEP(ra.x, ra.y, ra.w)
            ORIANA_NROW_STEP(ra.z, ra.w, rbq.y)
            ORIANA_NROW_STEP(rbq.x, rbq.y, rbq.w)
            ORIANA_NROW_STEP(rbq.z, rbq.w, qa[0].y)
#undef ORIANA_NROW_STEP
        }
        if (__any(bad) && lane == 0) tile_flag[t] = 1;
    }
    if (row < cm.n) {
        float *Rs = R + (int64_t)blockIdx.y * cm.n * KP;
        #pragma unroll
        for (int c = 0; c < KP4; ++c) reinterpret_cast<f4 *>(Rs)[row * KP4 + c] = acc[c];
    }
}

// column pass: one column tile per work item (work list of width 1, or grid.y row bands), C += with float atomics
template <int KP>
__global__ __launch_bounds__(256) void k_col_pass_narrow(oriana_counts cm, const float *__restrict__ s_cs,
                                                         const float *__restrict__ Gm, float *__restrict__ C,
                                                         const int32_t *__restrict__ work, int64_t rb_per_band) {
    constexpr int KP4 = Geo<KP>::KP4, ST4 = Geo<KP>::ST4;
    constexpr int PD = 3;
    extern __shared__ f4 lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sl = wave * 4 + (lane >> 4), g = lane & 15;
    int64_t cb, rb0, rb1;
    if (work) {
        cb = work[(int64_t)blockIdx.x * 3 + 0]; rb0 = work[(int64_t)blockIdx.x * 3 + 1]; rb1 = work[(int64_t)blockIdx.x * 3 + 2];
    } else {
        cb = blockIdx.x;
        rb0 = (int64_t)blockIdx.y * rb_per_band;
        rb1 = (rb0 + rb_per_band < cm.nrb) ? rb0 + rb_per_band : cm.nrb;
    }
    f4 acc[KP4];
    #pragma unroll
    for (int c = 0; c < KP4; ++c) acc[c] = f4{0.f, 0.f, 0.f, 0.f};
    Image<KP> img;
    if (rb0 < rb1) img.load(Gm, rb0 * TILE, cm.n, tid);
    for (int64_t rb = rb0; rb < rb1; ++rb) {
        const int64_t t = rb * cm.ncb + cb;
        const uint32_t s0 = cm.cslice[t * 17 + sl], s1 = cm.cslice[t * 17 + sl + 1];
        const int niter = (int)((s1 - s0) >> 6);
        const int nwave = __builtin_amdgcn_readfirstlane(wave_max_i(niter));
        const int64_t cbase = cm.coff[t] + s0 + g * 4;                     // this lane's four slots of iteration 0
        const f4 *sp = reinterpret_cast<const f4 *>(s_cs + cbase);
        const uint32_t *rp = reinterpret_cast<const uint32_t *>(cm.ridx + cbase);
        f4 sq[PD]; uint32_t rq[PD];
        #pragma unroll
        for (int d = 0; d < PD; ++d) {
            sq[d] = f4{0.f, 0.f, 0.f, 0.f}; rq[d] = 0u;
            if (d < niter) { sq[d] = sp[(int64_t)d * 16]; rq[d] = rp[(int64_t)d * 16]; }
        }
        ORIANA_SYNC();
        img.store(lds, tid);
        ORIANA_SYNC();
        if (rb + 1 < rb1) img.load(Gm, (rb + 1) * TILE, cm.n, tid);
        f4 v[KP4];
        {
            const f4 *vrow = lds + (rq[0] & 0xFFu) * ST4;
            #pragma unroll
            for (int c = 0; c < KP4; ++c) v[c] = vrow[c];
        }
        for (int it = 0; it < nwave; ++it) {
            const f4 sv = sq[0]; const uint32_t rv = rq[0];
            #pragma unroll
            for (int d = 0; d + 1 < PD; ++d) { sq[d] = sq[d + 1]; rq[d] = rq[d + 1]; }
            sq[PD - 1] = f4{0.f, 0.f, 0.f, 0.f}; rq[PD - 1] = 0u;
            if (it + PD < niter) { sq[PD - 1] = sp[(int64_t)(it + PD) * 16]; rq[PD - 1] = rp[(int64_t)(it + PD) * 16]; }
            // (the factor row of the next slot is read while this one is accumulated, as in the row pass)
#define ORIANA_NCOL_STEP(S, RNEXT)                                                                    \
            {                                                                                         \
                f4 vn[KP4];                                                                           \
                {                                                                                     \
                    const f4 *nrow = lds + ((RNEXT) & 0xFFu) * ST4;                                   \
                    _Pragma("unroll") for (int c = 0; c < KP4; ++c) vn[c] = nrow[c];                  \
                }                                                                                     \
                __builtin_amdgcn_sched_barrier(0);                                                    \
                const f2 ss = {(S), (S)};                                                             \
                _Pragma("unroll") for (int c = 0; c < KP4; ++c) {                                     \
                    acc[c].xy = __builtin_elementwise_fma(ss, v[c].xy, acc[c].xy);                    \
                    acc[c].zw = __builtin_elementwise_fma(ss, v[c].zw, acc[c].zw);                    \
                }                                                                                     \
                _Pragma("unroll") for (int c = 0; c < KP4; ++c) v[c] = vn[c];                         \
            }
            ORIANA_NCOL_STEP(sv.x, rv >> 8)
            ORIANA_NCOL_STEP(sv.y, rv >> 16)
            ORIANA_NCOL_STEP(sv.z, rv >> 24)
            ORIANA_NCOL_STEP(sv.w, rq[0])
#undef ORIANA_NCOL_STEP
        }
    }
    // through LDS (256 x KP floats: fits the image), then a contiguous flush: a wave's atomics then cover a few
    // cache lines instead of 64 (scattered, one gene per lane, the flush was 85 of the pass's 109 us at 10,000 x 2,000)
    const int cl = sl * 16 + g;
    float *ldsf = reinterpret_cast<float *>(lds);
    ORIANA_SYNC();
    #pragma unroll
    for (int c = 0; c < KP4; ++c) *reinterpret_cast<f4 *>(ldsf + cl * KP + 4 * c) = acc[c];
    ORIANA_SYNC();
    if (rb0 < rb1) {
        const int64_t c0 = cb * TILE, left = cm.m - c0;
        const int ncols = left < TILE ? (left > 0 ? (int)left : 0) : TILE;
        flush_block<256>(ldsf, C + c0 * KP, ncols * KP, false, tid);
    }
}

}  // namespace narrow

// ------------------------------------------------------------------------------------------
// dispatch on K:  Kp = 4 * G * T4
// ------------------------------------------------------------------------------------------
struct KCfg { int G, T4, TAIL; };
// Kp = 16 t (+4): the smallest padded width that holds K.  The tail (one extra float per lane)
// keeps K = 20, 50, 100 ... free of padding work.
static inline bool pick_cfg(int64_t K, KCfg *c) {
    if (K <= 0) return false;
    for (int t = 1; t <= 7; ++t) {
        if (K <= 16 * t) { *c = {4, t, 0}; return true; }
        if (t <= 6 && K <= 16 * t + 4) { *c = {4, t, 1}; return true; }
    }
    if (K <= 128) { *c = {8, 4, 0}; return true; }
    if (K <= 160) { *c = {8, 5, 0}; return true; }
    if (K <= 192) { *c = {8, 6, 0}; return true; }
    if (K <= 224) { *c = {8, 7, 0}; return true; }
    if (K <= 256) { *c = {16, 4, 0}; return true; }
    return false;
}

#define ORIANA_FOR_CFG(cfg, CALL)                                                       \
    do {                                                                                \
        if (cfg.G == 4 && cfg.T4 == 1 && cfg.TAIL == 0) { CALL(4, 1, 0); }              \
        else if (cfg.G == 4 && cfg.T4 == 1 && cfg.TAIL == 1) { CALL(4, 1, 1); }         \
        else if (cfg.G == 4 && cfg.T4 == 2 && cfg.TAIL == 0) { CALL(4, 2, 0); }         \
        else if (cfg.G == 4 && cfg.T4 == 2 && cfg.TAIL == 1) { CALL(4, 2, 1); }         \
        else if (cfg.G == 4 && cfg.T4 == 3 && cfg.TAIL == 0) { CALL(4, 3, 0); }         \
        else if (cfg.G == 4 && cfg.T4 == 3 && cfg.TAIL == 1) { CALL(4, 3, 1); }         \
        else if (cfg.G == 4 && cfg.T4 == 4 && cfg.TAIL == 0) { CALL(4, 4, 0); }         \
        else if (cfg.G == 4 && cfg.T4 == 4 && cfg.TAIL == 1) { CALL(4, 4, 1); }         \
        else if (cfg.G == 4 && cfg.T4 == 5 && cfg.TAIL == 0) { CALL(4, 5, 0); }         \
        else if (cfg.G == 4 && cfg.T4 == 5 && cfg.TAIL == 1) { CALL(4, 5, 1); }         \
        else if (cfg.G == 4 && cfg.T4 == 6 && cfg.TAIL == 0) { CALL(4, 6, 0); }         \
        else if (cfg.G == 4 && cfg.T4 == 6 && cfg.TAIL == 1) { CALL(4, 6, 1); }         \
        else if (cfg.G == 4 && cfg.T4 == 7 && cfg.TAIL == 0) { CALL(4, 7, 0); }         \
        else if (cfg.G == 8 && cfg.T4 == 4) { CALL(8, 4, 0); }                          \
        else if (cfg.G == 8 && cfg.T4 == 5) { CALL(8, 5, 0); }                          \
        else if (cfg.G == 8 && cfg.T4 == 6) { CALL(8, 6, 0); }                          \
        else if (cfg.G == 8 && cfg.T4 == 7) { CALL(8, 7, 0); }                          \
        else if (cfg.G == 16 && cfg.T4 == 4) { CALL(16, 4, 0); }                        \
        else return ORIANA_EKRANGE;                                                     \
    } while (0)

// The K = 85..100 kernels (namespace k100) replace the generic ones for Kp = 96 / 100 unless the environment
// says ORIANA_PASS_IMPL=r1 (A/B measurements, tools/perf1.py).
static bool round1_kernels() {
    static const bool r1 = [] { const char *e = getenv("ORIANA_PASS_IMPL"); return e && e[0] == 'r' && e[1] == '1'; }();
    return r1;
}
// Kp <= 32: one lane per row (namespace narrow) unless ORIANA_PASS_IMPL says r1 or r2 (A/B measurements)
static bool narrow_kernels() {
    static const bool off = [] { const char *e = getenv("ORIANA_PASS_IMPL"); return e && e[0] == 'r' && (e[1] == '1' || e[1] == '2'); }();
    return !off;
}
static bool use_narrow(int G, int T4, int TAIL) { return narrow_kernels() && G == 4 && 4 * T4 + TAIL <= 8; }       // row pass: Kp <= 32
// (column pass: Kp <= 20 -- at Kp = 32 the two-tile kernel measured 26.0 us against 28.3 at 10,000 x 2,000)
static bool use_narrow_col(int G, int T4, int TAIL) { return narrow_kernels() && G == 4 && 4 * T4 + TAIL <= 5; }
static bool use_k100(int G, int T4) { return !round1_kernels() && G == 4 && T4 == 6; }      // row pass, two lanes per row
// 33 <= Kp <= 64: two lanes per row / column (namespace k64) unless ORIANA_PASS_IMPL says r1, r2 or r3 (A/B measurements)
static bool k64_kernels() {
    static const bool off = [] { const char *e = getenv("ORIANA_PASS_IMPL"); return e && e[0] == 'r' && e[1] >= '1' && e[1] <= '3'; }();
    return !off;
}
// ORIANA_DEN_THRESHOLD=fixed: the row kernels keep the constant DEN_MIN (round 3's rule; A/B runs)
static bool den_threshold_dynamic() {
    static const bool fixed = [] { const char *e = getenv("ORIANA_DEN_THRESHOLD"); return e && !strcmp(e, "fixed"); }();
    return !fixed;
}
static constexpr bool k64_cfg(int G, int T4, int TAIL) { return G == 4 && 4 * T4 + TAIL >= 9 && 4 * T4 + TAIL <= 16; }
static bool use_col2(int G, int T4, int TAIL) { return !round1_kernels() && G == 4 && !use_narrow_col(G, T4, TAIL); }   // column pass, two tiles per image

template <typename KernelT>
static int set_lds(KernelT kern, size_t bytes) {
    if (bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return -1000 - (int)e;
    }
    return 0;
}

static inline size_t lds_bytes(int G, int T4, int TAIL) {
    const int KP = 4 * G * T4 + G * TAIL;
    return (size_t)(TILE / pick_nsub(KP)) * lds_stride_floats(KP) * sizeof(float);
}

template <int G, int T4, int TAIL>
static int launch_row_pass(const oriana_counts *cm, const float *FU, const float *FV, const float *w_nz, float *R,
                           float *s_cs, float *sw_cs, float *s_rs, int32_t *tile_flag, hipStream_t s,
                           const float *FV2, const oriana_row_split &sp, const float *den_min = nullptr) {
    const int var = (s_rs ? 1 : 0) | (w_nz ? 2 : 0);
    int rc;
    // two-lane kernels: one group per full row block, then the parts of the split ones (row_item)
    const int64_t items = (int64_t)sp.nfull + (cm->nrb - sp.nfull) * sp.parts;
    if (items > 0x7fffffffLL) return ORIANA_EINVAL;
    if constexpr (k64_cfg(G, T4, TAIL)) {
        if (k64_kernels() && !(FV2 && s_rs)) {
            constexpr int KP4 = 4 * T4 + TAIL;
            const size_t lb6 = (size_t)k64::IMG4 * 16 * (FV2 ? 2 : 1);
            const dim3 grid6((unsigned)items);
#define ORIANA_RP6(V)                                                                                 \
            rc = set_lds(k64::k_row_pass_k64<KP4, V>, lb6);                                           \
            if (rc) return rc;                                                                        \
            hipLaunchKernelGGL((k64::k_row_pass_k64<KP4, V>), grid6, dim3(512), lb6, s, *cm, FU, FV, w_nz, R, s_cs, sw_cs, s_rs, tile_flag, FV2, sp, den_min)
            const int v6 = var | (FV2 ? 4 : 0);
            if (v6 == 0) { ORIANA_RP6(0); }
            else if (v6 == 1) { ORIANA_RP6(1); }
            else if (v6 == 2) { ORIANA_RP6(2); }
            else if (v6 == 3) { ORIANA_RP6(3); }
            else if (v6 == 4) { ORIANA_RP6(4); }
            else { ORIANA_RP6(6); }
#undef ORIANA_RP6
            ORIANA_LAUNCH_CHECK();
            return 0;
        }
    }
    // every other kernel: whole-grid split only, ranges cut evenly by the kernel (gridDim.y)
    const bool whole_grid = sp.nfull == 0 || sp.parts == 1;
    const int gene_splits = sp.parts;
    if (FV2) {
        // two images of 256 factor rows side by side: only where both fit (and the tile needs no column sub-tiles)
        constexpr int KP = 4 * G * T4 + G * TAIL;
        const size_t lb2 = 2 * lds_bytes(G, T4, TAIL);
        if (use_k100(G, T4) || pick_nsub(KP) != 1 || lb2 > (size_t)LDS_BUDGET) return ORIANA_EKRANGE;
        if (!whole_grid) return ORIANA_EINVAL;
        const dim3 grid2((unsigned)(cm->nrb * WaveGeo<G>::SPLIT), (unsigned)gene_splits), block2(1024);
        if (w_nz) {
            rc = set_lds(k_row_pass<G, T4, TAIL, 6>, lb2);
            if (rc) return rc;
            hipLaunchKernelGGL((k_row_pass<G, T4, TAIL, 6>), grid2, block2, lb2, s, *cm, FU, FV, w_nz, R, s_cs, sw_cs, nullptr, tile_flag, FV2, den_min);
        } else {
            rc = set_lds(k_row_pass<G, T4, TAIL, 4>, lb2);
            if (rc) return rc;
            hipLaunchKernelGGL((k_row_pass<G, T4, TAIL, 4>), grid2, block2, lb2, s, *cm, FU, FV, w_nz, R, s_cs, sw_cs, nullptr, tile_flag, FV2, den_min);
        }
        ORIANA_LAUNCH_CHECK();
        return 0;
    }
    if (!use_k100(G, T4) && !whole_grid) return ORIANA_EINVAL;
    if constexpr (G == 4 && 4 * T4 + TAIL <= 8) {
        if (use_narrow(G, T4, TAIL) && var == 0) {
            constexpr int KP = 4 * G * T4 + G * TAIL;
            hipLaunchKernelGGL((narrow::k_row_pass_narrow<KP>), dim3((unsigned)cm->nrb, (unsigned)gene_splits), dim3(256),
                               narrow::Geo<KP>::bytes(), s, *cm, FU, FV, R, s_cs, tile_flag, den_min);
            ORIANA_LAUNCH_CHECK();
            return 0;
        }
    }
    if (use_k100(G, T4)) {
        constexpr int TL = (G == 4 && T4 == 6) ? TAIL : 0;
        const size_t lb2 = k100::image_bytes(TL);
#define ORIANA_RP2(V)                                                                                 \
        rc = set_lds(k100::k_row_pass_k100<TL, V>, lb2);                                              \
        if (rc) return rc;                                                                            \
        hipLaunchKernelGGL((k100::k_row_pass_k100<TL, V>), dim3((unsigned)items), dim3(512), lb2, s, *cm, FU, FV, w_nz, R, s_cs, sw_cs, s_rs, tile_flag, sp, den_min)
        if (var == 0) { ORIANA_RP2(0); }
        else if (var == 1) { ORIANA_RP2(1); }
        else if (var == 2) { ORIANA_RP2(2); }
        else { ORIANA_RP2(3); }
#undef ORIANA_RP2
        ORIANA_LAUNCH_CHECK();
        return 0;
    }
    const dim3 grid((unsigned)(cm->nrb * WaveGeo<G>::SPLIT), (unsigned)gene_splits), block(1024);
    const size_t lb = lds_bytes(G, T4, TAIL);
#define ORIANA_RP(V)                                                                                  \
    rc = set_lds(k_row_pass<G, T4, TAIL, V>, lb);                                                           \
    if (rc) return rc;                                                                                \
    hipLaunchKernelGGL((k_row_pass<G, T4, TAIL, V>), grid, block, lb, s, *cm, FU, FV, w_nz, R, s_cs, sw_cs, s_rs, tile_flag, (const float *)nullptr, den_min)
    if (var == 0) { ORIANA_RP(0); }
    else if (var == 1) { ORIANA_RP(1); }
    else if (var == 2) { ORIANA_RP(2); }
    else { ORIANA_RP(3); }
#undef ORIANA_RP
    ORIANA_LAUNCH_CHECK();
    return 0;
}

template <int G, int T4, int TAIL>
static int launch_row_spmm(const oriana_counts *cm, const float *s_rs, const float *w_nz, const float *FV,
                           float *R, hipStream_t s) {
    const dim3 grid((unsigned)(cm->nrb * WaveGeo<G>::SPLIT)), block(1024);
    const size_t lb = lds_bytes(G, T4, TAIL);
    int rc;
    if (w_nz) {
        rc = set_lds(k_row_spmm<G, T4, TAIL, true>, lb); if (rc) return rc;
        hipLaunchKernelGGL((k_row_spmm<G, T4, TAIL, true>), grid, block, lb, s, *cm, s_rs, w_nz, FV, R);
    } else {
        rc = set_lds(k_row_spmm<G, T4, TAIL, false>, lb); if (rc) return rc;
        hipLaunchKernelGGL((k_row_spmm<G, T4, TAIL, false>), grid, block, lb, s, *cm, s_rs, w_nz, FV, R);
    }
    ORIANA_LAUNCH_CHECK();
    return 0;
}

template <int G, int T4, int TAIL>
static int launch_col_pass(const oriana_counts *cm, const float *s_cs, const float *Gm, float *C,
                           const int32_t *work, int64_t nwork, float *Cpart, hipStream_t s) {
    constexpr int SPLIT = WaveGeo<G>::SPLIT;
    if constexpr (G == 4 && 4 * T4 + TAIL <= 5) {
        if (use_narrow_col(G, T4, TAIL) && !Cpart) {
            // one column tile per work item (oriana_col_block_tiles = 1)
            constexpr int KP = 4 * G * T4 + G * TAIL;
            auto kern = narrow::k_col_pass_narrow<KP>;
            const size_t lbn = narrow::Geo<KP>::bytes();
            if (work) {
                if (nwork <= 0) return 0;
                hipLaunchKernelGGL(kern, dim3((unsigned)nwork), dim3(256), lbn, s, *cm, s_cs, Gm, C, work, (int64_t)0);
            } else {
                int64_t nb = (2048 + cm->ncb - 1) / cm->ncb;
                const int64_t maxb = (cm->nrb + 3) / 4;
                if (nb > maxb) nb = maxb;
                if (nb < 1) nb = 1;
                if (nb > 65535) nb = 65535;
                const int64_t per = (cm->nrb + nb - 1) / nb;
                nb = (cm->nrb + per - 1) / per;
                hipLaunchKernelGGL(kern, dim3((unsigned)cm->ncb, (unsigned)nb), dim3(256), lbn, s, *cm, s_cs, Gm, C,
                                   (const int32_t *)nullptr, per);
            }
            ORIANA_LAUNCH_CHECK();
            return 0;
        }
    }
    if constexpr (k64_cfg(G, T4, TAIL)) {
        if (k64_kernels() && use_col2(G, T4, TAIL)) {
            // work items / grid.x index PAIRS of column tiles, as for k_col_pass2
            constexpr int KP4 = 4 * T4 + TAIL;
            const size_t lb6 = (size_t)k64::IMG4 * 16;
            auto kern6 = k64::k_col_pass_k64<KP4, false>;
            int rc6 = set_lds(kern6, lb6);
            if (rc6) return rc6;
            if (work) {
                if (nwork <= 0) return 0;
                hipLaunchKernelGGL(kern6, dim3((unsigned)nwork), dim3(1024), lb6, s, *cm, s_cs, Gm, C, work, (int64_t)0, Cpart,
                                   (const float *)nullptr, (float *)nullptr);
            } else {
                const int64_t ncp = (cm->ncb + 1) / 2;
                int64_t nb = (1024 + ncp - 1) / ncp;
                const int64_t maxb = (cm->nrb + 7) / 8;
                if (nb > maxb) nb = maxb;
                if (nb < 1) nb = 1;
                if (nb > 65535) nb = 65535;
                const int64_t per = (cm->nrb + nb - 1) / nb;
                nb = (cm->nrb + per - 1) / per;
                hipLaunchKernelGGL(kern6, dim3((unsigned)ncp, (unsigned)nb), dim3(1024), lb6, s, *cm, s_cs, Gm, C,
                                   (const int32_t *)nullptr, per, (float *)nullptr, (const float *)nullptr, (float *)nullptr);
            }
            ORIANA_LAUNCH_CHECK();
            return 0;
        }
    }
    if (use_col2(G, T4, TAIL)) {
        // work items / grid.x index PAIRS of column tiles (oriana_col_block_tiles = 2)
        constexpr int T4c = (G == 4) ? T4 : 1, TLc = (G == 4) ? TAIL : 0;       // (only instantiated for G = 4)
        const size_t lb2 = ColImage<T4c, TLc>::bytes();
        auto kern = k_col_pass2<T4c, TLc, false>;
        int rc2 = set_lds(kern, lb2);
        if (rc2) return rc2;
        if (work) {
            if (nwork <= 0) return 0;
            hipLaunchKernelGGL(kern, dim3((unsigned)nwork), dim3(1024), lb2, s, *cm, s_cs, Gm, C, work, (int64_t)0, Cpart,
                               (const float *)nullptr, (float *)nullptr);
        } else {
            const int64_t ncp = (cm->ncb + 1) / 2;
            int64_t nb = (1024 + ncp - 1) / ncp;
            const int64_t maxb = (cm->nrb + 7) / 8;
            if (nb > maxb) nb = maxb;
            if (nb < 1) nb = 1;
            if (nb > 65535) nb = 65535;
            const int64_t per = (cm->nrb + nb - 1) / nb;
            nb = (cm->nrb + per - 1) / per;
            hipLaunchKernelGGL(kern, dim3((unsigned)ncp, (unsigned)nb), dim3(1024), lb2, s, *cm, s_cs, Gm, C,
                               (const int32_t *)nullptr, per, (float *)nullptr, (const float *)nullptr, (float *)nullptr);
        }
        ORIANA_LAUNCH_CHECK();
        return 0;
    }
    const size_t lbw = lds_bytes(G, T4, TAIL);
    if (work) {
        if (nwork <= 0) return 0;
        int rcw = set_lds(k_col_pass<G, T4, TAIL>, lbw);
        if (rcw) return rcw;
        hipLaunchKernelGGL((k_col_pass<G, T4, TAIL>), dim3((unsigned)(nwork * SPLIT)), dim3(1024), lbw, s, *cm, s_cs, Gm, C,
                           work, (int64_t)0, Cpart);
        ORIANA_LAUNCH_CHECK();
        return 0;
    }
    // enough row bands to fill the chip (>= ~1024 workgroups) without shrinking a band below 8 tiles
    int64_t nb = (1024 + cm->ncb * SPLIT - 1) / (cm->ncb * SPLIT);
    int64_t maxb = (cm->nrb + 7) / 8;
    if (nb > maxb) nb = maxb;
    if (nb < 1) nb = 1;
    if (nb > 65535) nb = 65535;
    const int64_t per = (cm->nrb + nb - 1) / nb;
    nb = (cm->nrb + per - 1) / per;
    const dim3 grid((unsigned)(cm->ncb * SPLIT), (unsigned)nb), block(1024);
    const size_t lb = lds_bytes(G, T4, TAIL);
    int rc = set_lds(k_col_pass<G, T4, TAIL>, lb);
    if (rc) return rc;
    hipLaunchKernelGGL((k_col_pass<G, T4, TAIL>), grid, block, lb, s, *cm, s_cs, Gm, C, (const int32_t *)nullptr, per, (float *)nullptr);
    ORIANA_LAUNCH_CHECK();
    return 0;
}

}  // namespace oriana

using namespace oriana;

extern "C" int64_t oriana_kpad(int64_t K) {
    KCfg c;
    if (!pick_cfg(K, &c)) return 0;
    return 4 * c.G * c.T4 + c.G * c.TAIL;
}

extern "C" const char *oriana_version(void) { return "oriana_hip gfx950 0.4"; }

extern "C" int64_t oriana_col_block_tiles(int64_t K) {
    KCfg c;
    if (!pick_cfg(K, &c)) return 0;
    return use_col2(c.G, c.T4, c.TAIL) ? 2 : 1;
}

static bool counts_ok(const oriana_counts *cm) {
    if (!cm || cm->n < 0 || cm->m < 0) return false;
    if (cm->nrb != (cm->n + TILE - 1) / TILE || cm->ncb != (cm->m + TILE - 1) / TILE) return false;
    if (cm->nrb * cm->ncb > 0 && (!cm->roff || !cm->coff || !cm->rslice || !cm->cslice)) return false;
    if (cm->rslots > 0 && !cm->rowrec) return false;
    if (cm->cslots > 0 && !cm->ridx) return false;
    return true;
}

extern "C" int oriana_factor_prep(float *F, float *mu, const float *logF, const float *mask,
                                  const int32_t *row_index, int64_t r, int64_t K, void *stream) {
    const int64_t Kp = oriana_kpad(K);
    if (r < 0 || K <= 0) return ORIANA_EINVAL;
    if (Kp == 0) return ORIANA_EKRANGE;
    if (r == 0) return 0;
    if (!F || !logF) return ORIANA_EINVAL;
    hipLaunchKernelGGL(k_factor_prep, dim3((unsigned)((r + 3) / 4)), dim3(256), 0, (hipStream_t)stream, F, mu,
                       logF, mask, row_index, r, (int)K, (int)Kp);
    ORIANA_LAUNCH_CHECK();
    return 0;
}

extern "C" int64_t oriana_prep_center_offset(void) { return ((int64_t)sizeof(float) * (STATS_PART0 + 4 * 2 * STATS_MAX_BLOCKS) + 7) / 8 * 8; }
extern "C" int64_t oriana_prep_den_threshold_offset(void) { return 7 * (int64_t)sizeof(float); }
extern "C" int64_t oriana_prep_scratch_bytes(void) { return oriana_prep_center_offset() + 4096; }

static int factor_prep_pair_impl(float *FU, float *FV, const float *logU, const float *logV, const float *maskV,
                                 const int32_t *row_index_u, const int32_t *row_index_v, int64_t n, int64_t m,
                                 int64_t K, float *scratch, const oriana_clear_list *clr, const float *mu_u,
                                 const float *upart, int64_t nupart, void *stream) {
    const int64_t Kp = oriana_kpad(K);
    const bool fused = mu_u != nullptr;
    if (n < 0 || m < 0 || K <= 0) return ORIANA_EINVAL;
    if (Kp == 0) return ORIANA_EKRANGE;
    oriana_clear_list cl;
    memset(&cl, 0, sizeof(cl));
    int64_t clear_bytes = 0;
    if (clr) {
        cl = *clr;
        for (int e = 0; e < ORIANA_CLEAR_MAX; ++e) {
            if (cl.bytes[e] < 0 || (cl.bytes[e] & 3) || (cl.bytes[e] > 0 && (!cl.ptr[e] || ((uintptr_t)cl.ptr[e] & 3)))) return ORIANA_EINVAL;
            clear_bytes += cl.bytes[e];
        }
    }
    if (n == 0 && m == 0 && clear_bytes == 0) return 0;
    if ((n > 0 && (!FU || (!fused && !logU))) || (m > 0 && (!FV || !logV)) || !scratch) return ORIANA_EINVAL;
    if (fused && (!upart || nupart <= 0 || nupart > 0x7fffffffLL || n <= 0)) return ORIANA_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    // (one work-group per 64 rows, at most STATS_MAX_BLOCKS per side: every group ends with an agent-scope
    //  release / acquire pair, which on a small matrix costs more than the rows it covers)
    const int lane_rows = K <= 32 ? 1 : 0;
    const int64_t rows_per_group = lane_rows ? 256 : 64;
    auto capped = [&](int64_t r) { const int64_t b = (r + rows_per_group - 1) / rows_per_group; return (int)(b < STATS_MAX_BLOCKS ? b : STATS_MAX_BLOCKS); };
    const int sbu = fused ? 0 : capped(n);
    const int sbv = (fused && m == 0) ? 1 : capped(m);      // (fused: some group has to combine the cell side's partials)
    if (sbu + sbv > 0)
        hipLaunchKernelGGL(k_row_stats, dim3((unsigned)(sbu + sbv)), dim3(256), 0, s, scratch, logU, n, logV, maskV, m, (int)K, sbu,
                           lane_rows, den_threshold_dynamic() ? 1 : 0, fused ? upart : (const float *)nullptr, (int)nupart);
    const int64_t nbu = fused ? (n + 255) / 256 : (n + 3) / 4, nbv = (m + 3) / 4;
    // zero-fill groups: 16 KB each, at most 4096
    int64_t ncl = (clear_bytes + 16383) / 16384;
    if (ncl > 4096) ncl = 4096;
    if (nbu + nbv + ncl > 0x7fffffffLL) return ORIANA_EINVAL;
    hipLaunchKernelGGL(k_factor_prep_pair, dim3((unsigned)(nbu + nbv + ncl)), dim3(256), 0, s, FU, FV, logU, logV, maskV,
                       row_index_u, row_index_v, n, m, (int)K, (int)Kp, (int)nbu, (int)nbv, (const float *)scratch, cl, mu_u);
    ORIANA_LAUNCH_CHECK();
    return 0;
}

extern "C" int oriana_factor_prep_pair_clear(float *FU, float *FV, const float *logU, const float *logV, const float *maskV,
                                             const int32_t *row_index_u, const int32_t *row_index_v, int64_t n, int64_t m,
                                             int64_t K, float *scratch, const oriana_clear_list *clr, void *stream) {
    return factor_prep_pair_impl(FU, FV, logU, logV, maskV, row_index_u, row_index_v, n, m, K, scratch, clr, nullptr, nullptr, 0, stream);
}

extern "C" int oriana_factor_prep_pair_fused(float *FU, const float *mu_u, const float *upart, int64_t nupart, float *FV,
                                             const float *logV, const float *maskV, const int32_t *row_index_v, int64_t n,
                                             int64_t m, int64_t K, float *scratch, const oriana_clear_list *clr, void *stream) {
    if (!mu_u) return ORIANA_EINVAL;
    return factor_prep_pair_impl(FU, FV, nullptr, logV, maskV, nullptr, row_index_v, n, m, K, scratch, clr, mu_u, upart, nupart, stream);
}

extern "C" int oriana_factor_prep_pair(float *FU, float *FV, const float *logU, const float *logV, const float *maskV,
                                       const int32_t *row_index_u, const int32_t *row_index_v, int64_t n, int64_t m,
                                       int64_t K, float *scratch, void *stream) {
    return oriana_factor_prep_pair_clear(FU, FV, logU, logV, maskV, row_index_u, row_index_v, n, m, K, scratch, nullptr, stream);
}

static oriana_row_split no_split(const oriana_counts *cm) {
    oriana_row_split sp = {};
    sp.nfull = (int32_t)cm->nrb; sp.parts = 1; sp.edge[0] = 0; sp.edge[1] = (int32_t)cm->ncb;
    return sp;
}
static oriana_row_split even_split(const oriana_counts *cm, int64_t gene_splits) {        // oriana_row_pass_split
    oriana_row_split sp = {};
    sp.nfull = 0; sp.parts = (int32_t)gene_splits; sp.edge[0] = -1;
    return sp;
}
static bool split_ok(const oriana_counts *cm, const oriana_row_split &sp) {
    if (sp.nfull < 0 || sp.nfull > cm->nrb || sp.parts < 1 || sp.parts > 65535) return false;
    if (cm->ncb > 0 && sp.parts > cm->ncb) return false;
    if (sp.edge[0] < 0) return true;
    if (sp.parts > 8 || sp.edge[0] != 0 || sp.edge[sp.parts] != cm->ncb) return false;
    for (int e = 0; e < sp.parts; ++e)
        if (sp.edge[e + 1] < sp.edge[e]) return false;
    return true;
}

extern "C" int oriana_row_pass(const oriana_counts *cm, const float *FU, const float *FV, const float *w_nz,
                               float *R, float *s_cs, float *sw_cs, float *s_rs, int32_t *tile_flag, int64_t K,
                               void *stream) {
    if (!counts_ok(cm) || K <= 0) return ORIANA_EINVAL;
    KCfg cfg;
    if (!pick_cfg(K, &cfg)) return ORIANA_EKRANGE;
    if (cm->n == 0) return 0;
    if (!FU || !R || (cm->m > 0 && !FV) || (cm->m > 0 && (!s_cs || !tile_flag))) return ORIANA_EINVAL;
    if ((w_nz != nullptr) != (sw_cs != nullptr)) return ORIANA_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const oriana_row_split sp = no_split(cm);
#define CALL(G, T, L) return launch_row_pass<G, T, L>(cm, FU, FV, w_nz, R, s_cs, sw_cs, s_rs, tile_flag, s, nullptr, sp)
    ORIANA_FOR_CFG(cfg, CALL);
#undef CALL
    return 0;
}

// Gene-tile split of the plain row pass for short matrices: a row block is one work-group (two for K > 116), so a
// matrix of 10,000 cells runs the pass on 40 of the 256 CUs; splitting each row block's gene tiles over several groups
// fills the chip; each group of a row block stores its row sums in its own slab of R, which the consumer adds up
// (atomics on R cost 1.2 us per split at 10,000 x 20: more than the tile a split saves).
extern "C" int oriana_row_pass_plan_cus(const oriana_counts *cm, int64_t K, const double *tile_cost, int64_t cus, oriana_row_split *out) {
    if (cus <= 0) return ORIANA_EINVAL;
    if (!cm || !out || cm->nrb < 0 || cm->nrb > 0x7fffffffLL || cm->ncb > 0x7fffffffLL) return ORIANA_EINVAL;
    *out = no_split(cm);
    KCfg cfg;
    if (!pick_cfg(K, &cfg) || cm->nrb <= 0 || cm->ncb <= 1) return 0;
    const bool two_lane = use_k100(cfg.G, cfg.T4) || (k64_kernels() && k64_cfg(cfg.G, cfg.T4, cfg.TAIL));
    const int64_t groups = cm->nrb * ((two_lane || use_narrow(cfg.G, cfg.T4, cfg.TAIL)) ? 1 : (TILE / (16 * (64 / cfg.G))));
    static const int forced = [] { const char *e = getenv("ORIANA_ROW_SPLITS"); return e ? atoi(e) : 0; }();   // tuning runs
    static const bool rounds_off = [] { const char *e = getenv("ORIANA_ROW_SPLIT_ROUNDS"); return e && !strcmp(e, "off"); }();
    int64_t nfull = 0, parts = 1;
    if (forced > 0) {
        parts = forced;
    } else if (groups < cus) {
        // short matrices: two work-groups per CU at most, evenly sized ranges (measured at 10,000 x 2,000, K = 20:
        // 77 / 42 / 25 / 24 us for 1 / 2 / 4 / 8 groups per row block; 8 is the better sweep)
        parts = 2 * cus / groups;
    } else if (two_lane && !rounds_off) {
        // One 512-thread group per CU (the image and the registers leave room for one): the pass advances in rounds of `cus`
        // (256 on the MI355X the figures are from) row blocks and a partly filled last round costs a whole one (1M x 30k, K = 100: 3840 / 3907 / 4096 row blocks =
        // 33.9 / 35.8 / 36.2 ms; 391 row blocks -- configs[2] -- run as two rounds).  The row blocks of the last round are
        // split into p gene ranges each: ceil(tail * p / 256) / p rounds instead of one, + 1 % per extra range; a finer
        // split has to earn 3 %.
        const int64_t tail = cm->nrb % cus;
        if (tail == 0) return 0;
        double best = 1.0;
        int64_t bp = 1;
        for (int64_t p2 = 2; p2 <= 8 && p2 <= cm->ncb; ++p2) {
            const double c = (double)((tail * p2 + cus - 1) / cus) / (double)p2 + 0.01 * (double)(p2 - 1);
            if (c < 0.97 * best) { best = c; bp = p2; }
        }
        if (bp == 1) return 0;
        nfull = cm->nrb - tail; parts = bp;
    } else {
        return 0;
    }
    if (parts > cm->ncb) parts = cm->ncb;
    if (parts <= 1) return 0;
    {   // whole tiles per range: fewer parts if the tiles do not go round
        const int64_t per = (cm->ncb + parts - 1) / parts;
        parts = (cm->ncb + per - 1) / per;
    }
    out->nfull = (int32_t)nfull; out->parts = (int32_t)parts;
    if (!two_lane || parts > 8) { out->edge[0] = -1; return 0; }       // evenly cut (gridDim.y kernels / many short ranges)
    // equal-cost cut points (genes are packed by decreasing density: the first tiles are the long ones)
    double total = 0.0;
    for (int64_t c = 0; c < cm->ncb; ++c) total += tile_cost ? (tile_cost[c] > 0.0 ? tile_cost[c] : 0.0) : 1.0;
    if (!(total > 0.0)) { out->edge[0] = -1; return 0; }
    out->edge[0] = 0;
    double cum = 0.0;
    int64_t c = 0;
    for (int64_t e = 1; e < parts; ++e) {
        const double want = total * (double)e / (double)parts;
        while (c < cm->ncb) {
            const double w = tile_cost ? (tile_cost[c] > 0.0 ? tile_cost[c] : 0.0) : 1.0;
            if (cum + 0.5 * w > want) break;
            cum += w; ++c;
        }
        // every range keeps at least one tile
        const int64_t lo = out->edge[e - 1] + 1, hi = cm->ncb - (parts - e);
        int64_t edge = c < lo ? lo : (c > hi ? hi : c);
        while (c < edge) { cum += tile_cost ? (tile_cost[c] > 0.0 ? tile_cost[c] : 0.0) : 1.0; ++c; }
        out->edge[e] = (int32_t)edge;
    }
    out->edge[parts] = (int32_t)cm->ncb;
    return 0;
}

// [r5] ... for the device the calling thread has selected (oriana_device_cus: multiProcessorCount, not a literal 256)
extern "C" int oriana_row_pass_plan(const oriana_counts *cm, int64_t K, const double *tile_cost, oriana_row_split *out) {
    return oriana_row_pass_plan_cus(cm, K, tile_cost, oriana_device_cus(), out);
}

// (round 3's interface: the number of gene ranges of a whole-grid split; 1 when the plan splits the last round only)
extern "C" int64_t oriana_row_pass_gene_splits(const oriana_counts *cm, int64_t K) {
    oriana_row_split sp;
    if (oriana_row_pass_plan(cm, K, nullptr, &sp) != 0) return 1;
    return sp.nfull == 0 ? sp.parts : 1;
}

extern "C" int oriana_row_pass_split(const oriana_counts *cm, const float *FU, const float *FV, float *R, float *s_cs,
                                     int32_t *tile_flag, int64_t K, int64_t gene_splits, void *stream) {
    if (!counts_ok(cm) || K <= 0) return ORIANA_EINVAL;
    KCfg cfg;
    if (!pick_cfg(K, &cfg)) return ORIANA_EKRANGE;
    if (cm->n == 0) return 0;
    if (!FU || !R || (cm->m > 0 && !FV) || (cm->m > 0 && (!s_cs || !tile_flag))) return ORIANA_EINVAL;
    if (gene_splits < 1 || gene_splits > 65535 || (cm->ncb > 0 && gene_splits > cm->ncb)) return ORIANA_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const oriana_row_split sp = even_split(cm, gene_splits);
#define CALL(G, T, L) return launch_row_pass<G, T, L>(cm, FU, FV, nullptr, R, s_cs, nullptr, nullptr, tile_flag, s, nullptr, sp)
    ORIANA_FOR_CFG(cfg, CALL);
#undef CALL
    return 0;
}

extern "C" int oriana_row_pass_masked(const oriana_counts *cm, const float *FU, const float *FV, const float *FV2,
                                      const float *w_nz, float *R, float *s_cs, float *sw_cs, int32_t *tile_flag,
                                      int64_t K, void *stream) {
    if (!counts_ok(cm) || K <= 0) return ORIANA_EINVAL;
    KCfg cfg;
    if (!pick_cfg(K, &cfg)) return ORIANA_EKRANGE;
    if (cm->n == 0) return 0;
    if (!FU || !R || (cm->m > 0 && (!FV || !FV2)) || (cm->m > 0 && (!s_cs || !tile_flag))) return ORIANA_EINVAL;
    if ((w_nz != nullptr) != (sw_cs != nullptr)) return ORIANA_EINVAL;
    if (cm->m == 0) return oriana_row_pass(cm, FU, FV, w_nz, R, s_cs, sw_cs, nullptr, tile_flag, K, stream);
    hipStream_t s = (hipStream_t)stream;
    const oriana_row_split sp = no_split(cm);
#define CALL(G, T, L) return launch_row_pass<G, T, L>(cm, FU, FV, w_nz, R, s_cs, sw_cs, nullptr, tile_flag, s, FV2, sp)
    ORIANA_FOR_CFG(cfg, CALL);
#undef CALL
    return 0;
}

// The row pass in full generality: every variant of oriana_row_pass / oriana_row_pass_masked (FV2 may be null) with the gene
// tiles of a row block split over gene_splits work-groups, each storing its row sums in its own slab of R.
extern "C" int oriana_row_pass_general(const oriana_counts *cm, const float *FU, const float *FV, const float *FV2,
                                       const float *w_nz, float *R, float *s_cs, float *sw_cs, float *s_rs,
                                       int32_t *tile_flag, int64_t K, const oriana_row_split *split, const float *den_min,
                                       void *stream) {
    if (!counts_ok(cm) || K <= 0) return ORIANA_EINVAL;
    KCfg cfg;
    if (!pick_cfg(K, &cfg)) return ORIANA_EKRANGE;
    if (cm->n == 0) return 0;
    if (!FU || !R || (cm->m > 0 && !FV) || (cm->m > 0 && (!s_cs || !tile_flag))) return ORIANA_EINVAL;
    if ((w_nz != nullptr) != (sw_cs != nullptr) || (FV2 && s_rs)) return ORIANA_EINVAL;
    const oriana_row_split sp = split ? *split : no_split(cm);
    if (!split_ok(cm, sp)) return ORIANA_EINVAL;
    if (cm->m == 0) FV2 = nullptr;
    hipStream_t s = (hipStream_t)stream;
#define CALL(G, T, L) return launch_row_pass<G, T, L>(cm, FU, FV, w_nz, R, s_cs, sw_cs, s_rs, tile_flag, s, FV2, sp, den_min)
    ORIANA_FOR_CFG(cfg, CALL);
#undef CALL
    return 0;
}

extern "C" int oriana_row_spmm(const oriana_counts *cm, const float *s_rs, const float *w_nz, const float *FV,
                               float *R, int64_t K, void *stream) {
    if (!counts_ok(cm) || K <= 0) return ORIANA_EINVAL;
    KCfg cfg;
    if (!pick_cfg(K, &cfg)) return ORIANA_EKRANGE;
    if (cm->n == 0) return 0;
    if (!R || (cm->m > 0 && !FV) || (cm->rslots > 0 && !s_rs)) return ORIANA_EINVAL;
    hipStream_t s = (hipStream_t)stream;
#define CALL(G, T, L) return launch_row_spmm<G, T, L>(cm, s_rs, w_nz, FV, R, s)
    ORIANA_FOR_CFG(cfg, CALL);
#undef CALL
    return 0;
}

extern "C" int oriana_col_pass(const oriana_counts *cm, const float *s_cs, const float *Gm, float *C, int64_t K,
                               const int32_t *work, int64_t nwork, void *stream) {
    if (!counts_ok(cm) || K <= 0) return ORIANA_EINVAL;
    KCfg cfg;
    if (!pick_cfg(K, &cfg)) return ORIANA_EKRANGE;
    if (cm->n == 0 || cm->m == 0) return 0;
    if (!Gm || !C || !s_cs) return ORIANA_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (nwork < 0 || (work == nullptr && nwork != 0)) return ORIANA_EINVAL;
#define CALL(G, T, L) return launch_col_pass<G, T, L>(cm, s_cs, Gm, C, work, nwork, (float *)nullptr, s)
    ORIANA_FOR_CFG(cfg, CALL);
#undef CALL
    return 0;
}

// two images, one column tile per work item (work list of width 1)
template <int G, int T4, int TAIL>
static int launch_col_pass_dual(const oriana_counts *cm, const float *s_cs, const float *G1, const float *G2, float *C1,
                                float *C2, const int32_t *work, int64_t nwork, hipStream_t s) {
    constexpr int T4c = (G == 4) ? T4 : 1, TLc = (G == 4) ? TAIL : 0;
    using Im = ColImage<T4c, TLc>;
    if constexpr (k64_cfg(G, T4, TAIL)) {
        // measured at configs[4] (500k x 25k, K = 64): 11.3 ms against 10.9 ms for the four-lane kernel below -- with two
        // images per step the walk is bound by the LDS return port either way; ORIANA_COL_DUAL=k64 selects it for A/B runs
        static const bool dual64 = [] { const char *e = getenv("ORIANA_COL_DUAL"); return e && e[0] == 'k'; }();
        if (k64_kernels() && dual64) {
            if (nwork <= 0) return 0;
            constexpr int KP4 = 4 * T4 + TAIL;
            const size_t lb6 = (size_t)k64::IMG4 * 16 * 2;
            auto kern6 = k64::k_col_pass_k64<KP4, true>;
            int rc6 = set_lds(kern6, lb6);
            if (rc6) return rc6;
            hipLaunchKernelGGL(kern6, dim3((unsigned)nwork), dim3(1024), lb6, s, *cm, s_cs, G1, C1, work, (int64_t)0,
                               (float *)nullptr, G2, C2);
            ORIANA_LAUNCH_CHECK();
            return 0;
        }
    }
    if (G != 4 || Im::DUP || 2 * Im::bytes() > (size_t)LDS_BUDGET) return ORIANA_EKRANGE;
    if (nwork <= 0) return 0;
    constexpr bool OK = (G == 4) && !Im::DUP;
    auto kern = k_col_pass2<T4c, OK ? TLc : 0, OK>;
    const size_t lb = 2 * Im::bytes();
    int rc = set_lds(kern, lb);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)nwork), dim3(1024), lb, s, *cm, s_cs, G1, C1, work, (int64_t)0, (float *)nullptr,
                       G2, C2);
    ORIANA_LAUNCH_CHECK();
    return 0;
}

extern "C" int oriana_col_pass_dual(const oriana_counts *cm, const float *s_cs, const float *G1, const float *G2,
                                    float *C1, float *C2, int64_t K, const int32_t *work, int64_t nwork, void *stream) {
    if (!counts_ok(cm) || K <= 0) return ORIANA_EINVAL;
    KCfg cfg;
    if (!pick_cfg(K, &cfg)) return ORIANA_EKRANGE;
    if (cm->n == 0 || cm->m == 0) return 0;
    if (!G1 || !G2 || !C1 || !C2 || !s_cs || !work || nwork < 0) return ORIANA_EINVAL;
    hipStream_t s = (hipStream_t)stream;
#define CALL(G, T, L) return launch_col_pass_dual<G, T, L>(cm, s_cs, G1, G2, C1, C2, work, nwork, s)
    ORIANA_FOR_CFG(cfg, CALL);
#undef CALL
    return 0;
}

static int col_pass_partials(const oriana_counts *cm, const float *s_cs, const float *Gm, float *C, int64_t K,
                             const int32_t *work, int64_t nwork, float *scratch, hipStream_t s) {
    KCfg cfg;
    if (!pick_cfg(K, &cfg)) return ORIANA_EKRANGE;
#define CALL(G, T, L) return launch_col_pass<G, T, L>(cm, s_cs, Gm, C, work, nwork, scratch, s)
    ORIANA_FOR_CFG(cfg, CALL);
#undef CALL
    return 0;
}

extern "C" int64_t oriana_col_pass_det_scratch_bytes(int64_t K, int64_t nwork) {
    const int64_t Kp = oriana_kpad(K), w = oriana_col_block_tiles(K);
    if (Kp == 0 || nwork < 0) return 0;
    return nwork * w * TILE * Kp * (int64_t)sizeof(float);
}

extern "C" int oriana_col_pass_det(const oriana_counts *cm, const float *s_cs, const float *Gm, float *C, int64_t K,
                                   const int32_t *work, int64_t nwork, float *scratch, void *stream) {
    if (!counts_ok(cm) || K <= 0) return ORIANA_EINVAL;
    if (cm->n == 0 || cm->m == 0 || nwork == 0) return 0;
    if (!Gm || !C || !s_cs || !work || nwork < 0 || !scratch) return ORIANA_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    // every (item, column, factor) slot of the scratch is written by exactly one lane of the item that owns it, except
    // the columns past m and the padding lanes of partial column blocks: clear it first
    ORIANA_HIP_CHECK(hipMemsetAsync(scratch, 0, (size_t)oriana_col_pass_det_scratch_bytes(K, nwork), s));
    int rc = col_pass_partials(cm, s_cs, Gm, C, K, work, nwork, scratch, s);
    if (rc) return rc;
    const int64_t w = oriana_col_block_tiles(K);
    const int64_t nblk = (cm->ncb + w - 1) / w;
    hipLaunchKernelGGL(k_col_reduce, dim3((unsigned)nblk), dim3(256), 0, s, C, scratch, work, nwork, cm->m,
                       (int)oriana_kpad(K), (int)w);
    ORIANA_LAUNCH_CHECK();
    return 0;
}

extern "C" int oriana_finalize(float *Z, const float *F, const float *R, const float *mul,
                               const int32_t *row_index, int64_t r, int64_t K, int accumulate, void *stream) {
    const int64_t Kp = oriana_kpad(K);
    if (r < 0 || K <= 0) return ORIANA_EINVAL;
    if (Kp == 0) return ORIANA_EKRANGE;
    if (r == 0) return 0;
    if (!Z || !F || !R) return ORIANA_EINVAL;
    const int64_t tot = r * K;
    hipLaunchKernelGGL(k_finalize, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, Z, F, R,
                       mul, row_index, r, (int)K, (int)Kp, accumulate, 1, (int64_t)0);
    ORIANA_LAUNCH_CHECK();
    return 0;
}

extern "C" int oriana_finalize_slabs_from(float *Z, const float *F, const float *R, int64_t nslab, int64_t slab_row0,
                                          const int32_t *row_index, int64_t r, int64_t K, void *stream) {
    const int64_t Kp = oriana_kpad(K);
    if (r < 0 || K <= 0 || nslab < 1 || nslab > 65535 || slab_row0 < 0) return ORIANA_EINVAL;
    if (Kp == 0) return ORIANA_EKRANGE;
    if (r == 0) return 0;
    if (!Z || !F || !R) return ORIANA_EINVAL;
    const int64_t tot = r * K;
    hipLaunchKernelGGL(k_finalize, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, Z, F, R,
                       (const float *)nullptr, row_index, r, (int)K, (int)Kp, 1, (int)nslab, slab_row0);
    ORIANA_LAUNCH_CHECK();
    return 0;
}

extern "C" int oriana_finalize_slabs(float *Z, const float *F, const float *R, int64_t nslab, const int32_t *row_index,
                                     int64_t r, int64_t K, void *stream) {
    return oriana_finalize_slabs_from(Z, F, R, nslab, 0, row_index, r, K, stream);
}

extern "C" int oriana_fixup(const oriana_counts *cm, const int32_t *tile_flag, float *s_cs, float *sw_cs,
                            float *s_rs, const float *logU, const float *logV, const float *S_tilde,
                            const float *S_hat, const float *w_nz, const float *dq, float *Zi, float *Zj,
                            float *Zlog, int64_t K, int variant, void *stream) {
    if (!counts_ok(cm) || K <= 0) return ORIANA_EINVAL;
    const int64_t nt = cm->nrb * cm->ncb;
    if (nt == 0 || cm->nnz == 0) return 0;
    if (!tile_flag || !s_cs || !logU || !logV) return ORIANA_EINVAL;
    const int quirk = ((variant & 4) ? 1 : 0) | ((variant & 8) ? 2 : 0);       // bit 1: Z_hat_j indexed by the packed gene
    // (K <= number of genes is the caller's to check: cm->m of the sliced part of a hybrid layout counts its own genes only)
    if ((quirk & 1) && !dq) return ORIANA_EQUIRK;
    hipLaunchKernelGGL(k_fixup, dim3((unsigned)nt), dim3(256), 0, (hipStream_t)stream, *cm, tile_flag, s_cs,
                       sw_cs, s_rs, logU, logV, S_tilde, S_hat, w_nz, dq, Zi, Zj, Zlog, (int)K, quirk);
    ORIANA_LAUNCH_CHECK();
    return 0;
}

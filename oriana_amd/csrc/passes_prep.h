// passes_prep.h -- factor preparation of the responsibility pass (row statistics, shifted exponentials, centred validity test)
// Part of the one translation unit csrc/passes.hip (included there, in this order: passes_prep.h, passes_generic.h,
// passes_k100.h, passes_k64.h, passes_narrow.h); DESIGN.md section 0 says which family serves which (model, K).
#pragma once
#include "common.h"

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


}  // namespace oriana

// passes_narrow.h -- Kp <= 32: one lane per row / per gene
// Part of the one translation unit csrc/passes.hip (included there, in this order: passes_prep.h, passes_generic.h,
// passes_k100.h, passes_k64.h, passes_narrow.h); DESIGN.md section 0 says which family serves which (model, K).
#pragma once
#include "common.h"

namespace oriana {

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
        __syncthreads();                                                     // everybody is done with the previous image
        img.store(lds, tid);
        __syncthreads();
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
                sdst[valid ? (bm & 0xFFFFu) : dummy] = slow ? NAN : s;                 \
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
        __syncthreads();
        img.store(lds, tid);
        __syncthreads();
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
    __syncthreads();
    #pragma unroll
    for (int c = 0; c < KP4; ++c) *reinterpret_cast<f4 *>(ldsf + cl * KP + 4 * c) = acc[c];
    __syncthreads();
    if (rb0 < rb1) {
        const int64_t c0 = cb * TILE, left = cm.m - c0;
        const int ncols = left < TILE ? (left > 0 ? (int)left : 0) : TILE;
        flush_block<256>(ldsf, C + c0 * KP, ncols * KP, false, tid);
    }
}

}  // namespace narrow


}  // namespace oriana

// passes_k64.h -- 33 <= Kp <= 64: two lanes per row and per column
// Part of the one translation unit csrc/passes.hip (included there, in this order: passes_prep.h, passes_generic.h,
// passes_k100.h, passes_k64.h, passes_narrow.h); DESIGN.md section 0 says which family serves which (model, K).
#pragma once
#include "common.h"

namespace oriana {

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
            __syncthreads();
            stg.store(lds, tid);
        }
        if (F2I) {
            Stage<512, KP4> stg2;
            stg2.load(FV2, cb * TILE, cm.m, tid);
            stg2.store(lds + IMG4, tid);
        }
        __syncthreads();
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
                sdst[off] = sout;                                                      \
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
            __syncthreads();
            stg.store(lds, tid);
        }
        if (DUAL) {
            Stage<1024, KP4> stg2;
            stg2.load(Gm2, rb * TILE, cm.n, tid);
            stg2.store(lds + IMG4, tid);
        }
        __syncthreads();
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
        __syncthreads();
        if (half == h && present && c0 + cl < cm.m) {
            float *rowp = ldsf + cl * KP;
            #pragma unroll
            for (int t = 0; t < T4; ++t)
                if (lidx[t] < KP4) *reinterpret_cast<f4 *>(rowp + lidx[t] * 4) = acc[t];
        }
        __syncthreads();
        if (h == 0 || hasB || DUAL) {
            const int64_t left = cm.m - c0;
            const int ncols = left < TILE ? (left > 0 ? (int)left : 0) : TILE;
            float *dst = plain ? Cpart + ((int64_t)blockIdx.x * 2 * TILE + h * TILE) * KP : ((DUAL && h) ? C2 : C) + c0 * KP;
            flush_block<1024>(ldsf, dst, ncols * KP, plain, tid);
        }
    }
}

}  // namespace k64



}  // namespace oriana

// passes_k100.h -- K = 85..100 (Kp = 96 / 100): two lanes per row on a conflict-free LDS image; the two-tile column pass for every Kp <= 116
// Part of the one translation unit csrc/passes.hip (included there, in this order: passes_prep.h, passes_generic.h,
// passes_k100.h, passes_k64.h, passes_narrow.h); DESIGN.md section 0 says which family serves which (model, K).
#pragma once
#include "common.h"

namespace oriana {

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
        {
        Stage<512, TAIL> stg;
        stg.load(FV, cb * TILE, cm.m, tid);
        __syncthreads();
        stg.store(lds, tid);
        __syncthreads();
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
                sdst[off] = sout;                                                      \
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
// of a step in flight -- measured, DESIGN_HISTORY.md section 8)
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
                    {                                                      \
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
            __syncthreads();
            stg.store(lds, tid);
        } else {
            {
                Stage<Im::KP4, Im::TREP, TILE> stg;
                stg.load(Gm, rb * TILE, cm.n, tid);
                __syncthreads();
                stg.template store<ROW4>(lds, tid);
            }
            if (DUAL) {
                Stage<Im::KP4, Im::TREP, TILE> stg2;
                stg2.load(Gm2, rb * TILE, cm.n, tid);
                stg2.template store<ROW4>(lds + IMG4, tid);
            }
        }
        __syncthreads();
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
        __syncthreads();
        if ((h == 0 || hasB || DUAL) && c0 + cl < cm.m) {
            float *row = ldsf + cl * KP;
            #pragma unroll
            for (int t = 0; t < T4; ++t) *reinterpret_cast<f4 *>(row + Im::gidx(lane, t) * 4) = h ? accB[t] : accA[t];
            if (TAIL) row[Im::TOFF + q] = h ? actB : actA;
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



}  // namespace oriana

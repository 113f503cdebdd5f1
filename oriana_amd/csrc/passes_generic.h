// passes_generic.h -- the general kernels of the pass: G lanes per row / column, every K <= 256 and every variant; finalize; the exact slow path
// Part of the one translation unit csrc/passes.hip (included there, in this order: passes_prep.h, passes_generic.h,
// passes_k100.h, passes_k64.h, passes_narrow.h); DESIGN.md section 0 says which family serves which (model, K).
#pragma once
#include "common.h"

namespace oriana {

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
    if (G == 4 && T4 == 6) {
        const int c = (lane >> 2) & 3;
        int ch = t ^ (c & 1);                                   // classes 1, 3 swap inside the pairs
        if ((c & 2) && ch >= 2) ch = (ch < 4) ? ch + 2 : ch - 2;   // classes 2, 3 swap the pairs (2,3) <-> (4,5)
        return ch;
    }
    return (t + rot) % T4;
}

template <int G>
__device__ __forceinline__ int lds_rot(int lane) {
    // ds_read_b128 is serviced in fixed 16-lane sets; quads that are serviced together must start
    // at different 64-byte quarters of the 256-byte bank row.  Measured on MI355X with
    // tools/ubench/lds_pat.hip (random 512-byte rows, 7 chunks): no rotation 16.5, (Q&7)>>1 6.6,
    // this one 6.1, broadcast floor 5.4 cycles per wave-instruction.
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
        if (plain) dst[idx] = v;
        else if (v != 0.f) atomicAdd(dst + idx, v);
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
                __syncthreads();                // everybody is done with the previous image
                stg.template store<STRIDE4>(lds, tid);
            }
            if (F2I) {
                Stage<KP4, TREP, CT> stg2;
                stg2.load(FV2, cb * TILE + csub * CT, cm.m, tid);
                stg2.template store<STRIDE4>(lds + CT * STRIDE4, tid);
            }
            __syncthreads();
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
                    _Pragma("unroll") for (int tt = 0; tt < T4; ++tt) v[tt] = vrow[choff[tt]];        \
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
                            const f4 v2 = vrow2[choff[tt]];                           \
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
                    sdst[off] = sout;                                                  \
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
            __syncthreads();
            stg.template store<STRIDE4>(lds, tid);
            __syncthreads();
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
                        const f4 v = vrow[choff[tt]];                                                 \
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
                        const f4 v = vrow[choff[tt]];                                 \
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
            __syncthreads();
            stg.template store<STRIDE4>(lds, tid);
            __syncthreads();
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
            __syncthreads();
            stg.template store<STRIDE4>(lds, tid);
            __syncthreads();
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
        __syncthreads();                                             // every wave is done with the last image
        if (cb2 * TILE + cl0 + cl2 < cm.m) {
            float *row = ldsf + cl2 * KP;
            #pragma unroll
            for (int t = 0; t < T4; ++t) *reinterpret_cast<f4 *>(row + choff[t] * 4) = acc[t];
            if (TAIL) row[TOFF + q] = acct;
        }
        __syncthreads();
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
                                               float *__restrict__ Zj, float *__restrict__ Zlog, int K, int quirk,
                                               int64_t nt, int per) {
    // [r6] One work-group per `per` <= 64 tiles (the host: nt / 2048, so that a start with every tile flagged still spreads
    // over >= 2048 groups): every wave reads the same flags and walks the flagged ones (the mask is wave-uniform and identical
    // in the four waves, so the barriers below are reached by all).  Round 5 launched a group per tile: 0.12 ms at configs[3]
    // for 4e5 groups that read one flag and left.
    __shared__ uint32_t queue[FIX_WINDOW];
    __shared__ uint32_t rs[17];
    __shared__ uint32_t qn;
    __shared__ __attribute__((aligned(16))) float ebuf[4][FIX_KMAX];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t t0 = (int64_t)blockIdx.x * per;
    const bool mine = lane < per && t0 + lane < nt;
    uint64_t pending = __ballot(mine && tile_flag[mine ? t0 + lane : 0] != 0);
    while (pending) {
        const int64_t t = t0 + __builtin_ctzll(pending);
        pending &= pending - 1;
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
        __syncthreads();                     // (rs, queue, qn belong to the next flagged tile from here on)
    }
}


}  // namespace oriana

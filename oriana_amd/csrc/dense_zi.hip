// dense_zi.hip -- the two dense contractions of a ZI sweep for 32 < K <= 100 on the bf16 matrix cores, float32-equivalent
// arithmetic (exact three-way bf16 splits, six cross products, float32 accumulation, no chain beyond 256 terms; the
// tail factors Kp - 16 KC on the float32 matrix instructions): the kernels of dense_pass.hip with the sigmoid in the
// place of s = x / den.
//
//   k_zi_row   D update (zigap.py:130-136, sparse_zigap.py:163-169) fused with the next sweep's D_hat V (zigap.py:116):
//              Lambda^T[gene, cell] = V U^T per 32 x 32 tile, p = sigmoid(logit pi_j - Lambda) + overrides in float32,
//              D_hat out as 128-byte row pieces, sum_i p_d, DV[cell, k] += D_hat V_next.  Same software pipeline as
//              dn::k_dn_row (S(t) beside the matrix instructions of Lambda(t + 1), then the product of tile t).
//   k_zi_col   out[gene, k] += sum_i D_hat[i, gene] W[i, k]  (zigap.py:124): dn::k_dn_col reading D_hat row-major.
//
// Serves 33 <= K <= 100 (zi_cfg / zi_cfg_dt below: measured assignment); dense_f32.hip keeps K <= 32 and the float32 matrix
// instruction for K > 100 or a gene count that is not a multiple of 4 (16-byte row pieces).
#include "dense_tiles.h"
#include <stdlib.h>
#include <string.h>

namespace oriana {
namespace dn {

// Operand images from float64 factors (leading dimension K).  BOTH: gene side -- first image and tail rows from F1 (the V
// of the D update), second image and the accumulator-order tail from F2 (V_next); otherwise the cell side (second image
// + tail from F2 only).
template <int KC, int TAIL, bool BOTH>
__global__ __launch_bounds__(512) void k_zi_images(u4v *__restrict__ img, const double *__restrict__ F1,
                                                   const double *__restrict__ F2, int64_t rows, int K) {
    using C = Cfg<KC, TAIL>;
    constexpr int PIMG = BOTH ? C::PVZ : C::PU;
    // [r6] D update with a tail that fits the last factor tile of the second product (KC = 3: factors 48..51 inside 32..63; KC = 5:
    // 80..83 inside 64..95): the tail factors of V_next ride in that tile's image columns, which were zero -- the second product
    // then needs no float32 4 x 4 x 1 instructions for them (k_zi_row: TIB)
    constexpr int KM2 = (BOTH && TAIL && 32 * C::NT >= C::KM + 4) ? C::KM + 4 : C::KM;
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * 32;
    u4v *dst0 = img + (int64_t)blockIdx.x * PIMG;
    if (BOTH) {
        const int g = tid & 31, G = tid >> 5;
        if (G < 2 * KC) {
            const int64_t r = r0 + g;
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { const int kk = 8 * G + e; x[e] = (r < rows && kk < K) ? (float)F1[r * K + kk] : 0.f; }
            u4v o[3];
            split8(x, o);
            u4v *dst = dst0 + ((G >> 1) * 3) * 64 + (G & 1) * 32 + g;
#pragma unroll
            for (int sp = 0; sp < 3; ++sp) dst[sp * 64] = o[sp];
        }
    }
    {
        const int nt = tid >> 7, q = (tid >> 6) & 1, hh = (tid >> 5) & 1, cc = tid & 31;
        if (nt < C::NT) {
            const int kk = nt * 32 + cc;
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int64_t r = r0 + acc_row(8 * q + e, hh);
                x[e] = (r < rows && kk < KM2 && kk < K) ? (float)F2[r * K + kk] : 0.f;
            }
            u4v o[3];
            split8(x, o);
            u4v *dst = dst0 + (BOTH ? C::P1 : 0) + ((nt * 2 + q) * 3) * 64 + hh * 32 + cc;
#pragma unroll
            for (int sp = 0; sp < 3; ++sp) dst[sp * 64] = o[sp];
        }
    }
    if (TAIL && tid < 64) {
        f4v t = {0.f, 0.f, 0.f, 0.f};
        if (tid < 32) {                              // pieces 0..31: the four tail factors of row tid (first product)
            const int64_t r = r0 + tid;
            const double *F = BOTH ? F1 : F2;
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] = (r < rows && C::KM + e < K) ? (float)F[r * K + C::KM + e] : 0.f;
        } else {                                     // pieces 32..63: [lane half hh][tail factor j][4 q .. 4 q + 3]
            const int idx = tid - 32, hh = idx >> 4, j = (idx >> 2) & 3, q = idx & 3;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int64_t r = r0 + acc_row(4 * q + e, hh);
                t[e] = (r < rows && C::KM + j < K) ? (float)F2[r * K + C::KM + j] : 0.f;
            }
        }
        dst0[(BOTH ? C::P1 : 0) + C::P2 + tid] = __builtin_bit_cast(u4v, t);
    }
}

// ---- D update -------------------------------------------------------------------------------------------------------
// The per-tile side data ride in the padding of the image ring: a gene-side image is PV_RAW pieces (a multiple of 64)
// copied as PV = multiple of 512; the wave whose 64-piece slot is the first padding one copies the 8 waves' non-zero flags
// instead (128 bytes per wave and tile, below), the next one -- from lgit[0 .. 2 mpad) -- the scaled logits of the tile's 32
// genes (lanes 0-7) and their floors (lanes 8-15: 1e-10 where pi_d <= 0, else 0).  They arrive with the image, two tiles
// ahead, and cost no LDS of their own.
//
// [r6] The non-zero flags come PER LANE (oriana_nzmask_tiles): for cell tile ct and gene tile gt, 64 x 16 bits, bit v of
// entry l = (X[32 ct + l % 32, 32 gt + 8 (v / 4) + 4 (l / 32) + v % 4] != 0) -- the 16 values lane l of the wave holds, in
// register order.  One 2-byte LDS read per lane and tile, then v_bfe_i32 (flag -> 0 / ~0) + v_bfi_b32 (select 1.0f) per value:
// round 3 read the mask words of oriana_nzmask_f32 (bit = cell) and spent and + compare + select per value on the lane's bit.
template <int KC, int TAIL>
__device__ __forceinline__ void zi_tile_dma(const u4v *__restrict__ imgV, const uint32_t *__restrict__ nztiles,
                                            const float *__restrict__ lgit, int64_t mpad, u4v *dst, int gt, int64_t ct_blk0,
                                            int ngt, int64_t m, int w, int lane) {
    // Every copy is  wave-uniform base (scalar registers) + 32-bit lane offset , both chosen WITHOUT a branch: the loop
    // body of k_zi_row must stay one basic block (a predicated copy, or a select between per-lane 64-bit pointers, is
    // compiled into control flow / costs the registers the kernel does not have).  Clamped, never predicated: what a
    // clamped lane fetches belongs to padding cells / genes, whose values are neither stored nor summed.
    using C = Cfg<KC, TAIL>;
    constexpr int SLOT0 = C::PV_RAW / 64;
    static_assert(C::PV_RAW % 64 == 0 && C::PVZ / 64 >= SLOT0 + 2, "the image padding holds the flag and logit pieces");
    const int64_t j0 = (int64_t)gt * 32;
    const char *ibase = reinterpret_cast<const char *>(imgV + (int64_t)gt * C::PVZ);
    // (the flag array covers whole work-groups of 8 cell tiles: no clamp; lane l copies piece l % 8 of wave l / 8)
    const char *mbase = reinterpret_cast<const char *>(nztiles + (ct_blk0 * ngt + gt) * 32);
    const char *lbase = reinterpret_cast<const char *>(lgit + j0);
    const int jlim = (int)((m - 4 - j0 < 252) ? m - 4 - j0 : 252);          // last whole 16-byte piece of the row (>= 0)
#pragma unroll
    for (int p = 0; p < C::PVZ / (NW * 64); ++p) {
        const int slot = p * NW + w;                                       // wave-uniform
        const char *base = ibase;
        uint32_t voff = (uint32_t)(slot * 64 + lane) * 16u;                // an image piece (or its padding: harmless)
        if (p * NW + NW - 1 >= SLOT0) {
            const int l8 = lane & 7;
            const uint32_t moff = (uint32_t)(lane >> 3) * (uint32_t)ngt * 128u + (uint32_t)l8 * 16u;   // < 8 ngt x 128 bytes
            const int jl = min(4 * ((lane < 16) ? l8 : lane), jlim);
            const uint32_t loff = ((uint32_t)jl + (((lane >> 3) == 1) ? (uint32_t)mpad : 0u)) * 4u;
            base = (slot == SLOT0) ? mbase : (slot == SLOT0 + 1) ? lbase : ibase;
            voff = (slot == SLOT0) ? moff : (slot == SLOT0 + 1) ? loff : voff;
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + voff),
                                         (__attribute__((address_space(3))) void *)(dst + slot * 64), 16, 0, 0);
    }
}

template <int KC, int TAIL>
constexpr int zi_row_lds_bytes() { return 3 * Cfg<KC, TAIL>::PVZ * 16 + NW * 32 * 32 * 4 + 2 * NW * 32 * 4; }

template <int KC, int TAIL>
__global__ __launch_bounds__(512) void k_zi_row(float *__restrict__ D_hat, const double *__restrict__ U,
                                                const u4v *__restrict__ imgV, const float *__restrict__ lgit, int64_t mpad,
                                                const uint32_t *__restrict__ nztiles, double *__restrict__ colsum,
                                                double *__restrict__ DV, int64_t n, int64_t m, int K, int ngt,
                                                int gt_per_split) {
    using C = Cfg<KC, TAIL>;
    constexpr int NT = C::NT;
    constexpr int MP = C::PV_RAW;                                          // first mask piece inside an image buffer
    // the tail factors of the SECOND product ride in the last factor tile's image (k_zi_images: KM2): no 4 x 4 x 1 instructions
    constexpr bool TIB = TAIL && 32 * NT >= C::KM + 4;
    constexpr bool DMA_TOP = KC <= 4;     // [r6] KC = 4 too: 3.69 against 3.99 ms at K = 50 / 64 on four chunks (profiles/r06_zi_k52_ab.txt)
    extern __shared__ u4v ldsq[];
    u4v *img = ldsq;                                                      // [3][PV]
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    float *T = reinterpret_cast<float *>(ldsq + 3 * C::PVZ) + w * 32 * 32;  // [wave][32 cells][32 genes], 16-byte chunks swizzled
    float *csb = reinterpret_cast<float *>(ldsq + 3 * C::PVZ) + NW * 32 * 32;   // [2][8 waves][32 genes]
    const int64_t ct_blk0 = (int64_t)blockIdx.x * NW;
    const int64_t ct = ct_blk0 + w;                                        // this wave's cell tile
    const int64_t i = ct * 32 + c;
    const int64_t nmrows = (n + 31) / 32;
    const int gt0 = blockIdx.y * gt_per_split;
    const int gt1 = (gt0 + gt_per_split < ngt) ? gt0 + gt_per_split : ngt;
    if (gt0 >= gt1) return;

    // the wave's strip of U_hat as the B operand of the first product: per k chunk, factors 16 kc + 8 h + e of cell c
    // (padding cells: zeros -- their rows are neither stored nor summed)
    u4v ub[KC][3];
    f4v fut = {0.f, 0.f, 0.f, 0.f};
    {
        const double *urow = U + (i < n ? i : 0) * K;
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
        if (TAIL) {
#pragma unroll
            for (int e = 0; e < 4; ++e) fut[e] = (i < n && C::KM + e < K) ? (float)urow[C::KM + e < K ? C::KM + e : K - 1] : 0.f;
        }
    }
    const float futb0 = h ? fut.y : fut.x, futb1 = h ? fut.w : fut.z;

    f16v rs[NT];                         // DV of the strip: [cell acc_row(v, h)][factor 32 nt + c]
    f4v rt = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int v = 0; v < 16; ++v) rs[nt][v] = 0.f;

    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
    // The transpose buffer: row = cell (32 floats = 8 chunks of 16 bytes), chunk j of row r stored at chunk j ^ (r & 7).
    // Writes (ds_write_b128: 8 consecutive lanes per LDS cycle, bank = dword % 32): lanes c .. c + 7 write chunk 2 q + h of
    // rows c .. c + 7 -> eight different physical chunks.  Read-back (ds_read_b128: groups {0-3, 12-15, 20-27}, {4-11, 16-19,
    // 28-31} (+ 32), bank = dword % 64): lane l reads chunk l % 8 of row l / 8 + 8 q; odd rows sit in the upper half of the
    // bank row, and inside a group the lanes of two even (odd) rows take the two halves of their 128 bytes: no conflict
    // either way (round 5: 36-float rows, every read group 2-way: profiles/r05 SQ_LDS_BANK_CONFLICT 10.4 %).
    const int rr = lane >> 3, gq = (lane & 7) * 4;
    const uint32_t Tw0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float *)T + (uint32_t)(c * 128 + ((h ^ (c & 7)) * 16));
    const float *Tr = T + rr * 32 + (((lane & 7) ^ rr) * 4);               // + 8 q rows: row of cell rr + 8 q, genes gq ..
    // D_hat leaves through a buffer resource over the wave's (at most 32) rows: a row beyond the matrix is beyond the
    // resource's size and the hardware drops the store -- no predicate, no branch; a piece beyond the last gene gets an
    // offset that is out of range.  (num_records <= 32 m floats: 32-bit for any m below 3e7)
    const int64_t rows_here = (n - ct * 32 < 0) ? 0 : (n - ct * 32 > 32 ? 32 : n - ct * 32);
    __amdgpu_buffer_rsrc_t drsrc = __builtin_amdgcn_make_buffer_rsrc(D_hat + (rows_here > 0 ? ct * 32 * m : 0), 0,
                                                                     (int)(rows_here * m * 4), 0x00020000);
    const uint32_t dvoff = ((uint32_t)rr * (uint32_t)m + (uint32_t)gq) * 4u;
    const int rleft = (int)rows_here - rr;                                 // row rr + 8 q of the read-back is a cell iff 8 q < rleft

    auto phase_D = [&](const u4v *im) -> f16v {
        f16v l0;
#pragma unroll
        for (int v = 0; v < 16; ++v) l0[v] = 0.f;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            u4v a[3];
#pragma unroll
            for (int sp = 0; sp < 3; ++sp) a[sp] = im[(kc * 3 + sp) * 64 + lane];
            ORIANA_DN_MF6(l0, a, ub[kc]);
        }
        if (TAIL) {
            const float *tl = reinterpret_cast<const float *>(im + C::P1 + C::P2) + c * 4 + h;
            l0 = __builtin_amdgcn_mfma_f32_32x32x2f32(tl[0], futb0, l0, 0, 0, 0);
            l0 = __builtin_amdgcn_mfma_f32_32x32x2f32(tl[2], futb1, l0, 0, 0, 0);
        }
        return l0;
    };
    // sum_i p_d of the previous tile: the eight waves' partial sums (written before the last barrier), one float64
    // atomic per gene
    auto colsum_flush = [&](int gtp, int parity) {
        if (colsum && w == 0 && lane < 32) {
            const int64_t j = (int64_t)gtp * 32 + lane;
            if (j < m) {
                const float *cs = csb + parity * NW * 32 + lane;
                double t = 0.0;
#pragma unroll
                for (int ww = 0; ww < NW; ++ww) t += (double)cs[ww * 32];
                atomicAdd(&colsum[j], t);
            }
        }
    };

    zi_tile_dma<KC, TAIL>(imgV, nztiles, lgit, mpad, img, gt0, ct_blk0, ngt, m, w, lane);
    zi_tile_dma<KC, TAIL>(imgV, nztiles, lgit, mpad, img + C::PVZ, (gt0 + 1 < gt1) ? gt0 + 1 : gt0, ct_blk0, ngt, m, w, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f16v dn = phase_D(img);              // Lambda^T of tile gt0
    int buf = 0, par = 0;
    constexpr int NA = KC * 6 + (TAIL ? 2 : 0);                            // matrix instructions of the first product
    constexpr int NB = NT * 12;                                            // ... of the second one
    for (int gt = gt0; gt < gt1; ++gt) {
        const int bufn = (buf == 2) ? 0 : buf + 1, bufnn = (buf == 0) ? 2 : buf - 1;
        const u4v *im1 = img + bufn * C::PVZ;                               // tile gt + 1: first product
        const u4v *im0 = img + buf * C::PVZ;                                // tile gt: second product, masks, logits
        const int64_t j0 = (int64_t)gt * 32;
        if (gt > gt0) colsum_flush(gt - 1, par ^ 1);
        // The image, masks and logits of tile gt + 2 go out into the buffer tile gt - 1 has left: in the middle of the
        // matrix work for the long loops (item 16 below; measured at 100k x 20k: 5.10 against 5.26 ms at K = 100, 4.38 / 4.46
        // at K = 80), here at the top for KC <= 3 (3.45 against 3.78 ms at K = 48)
        if (DMA_TOP)
            zi_tile_dma<KC, TAIL>(imgV, nztiles, lgit, mpad, img + bufnn * C::PVZ, (gt + 2 < gt1) ? gt + 2 : gt1 - 1, ct_blk0, ngt, m,
                                  w, lane);
        u4v A0[2], A1, A2;
        A2 = im1[2 * 64 + lane]; A0[0] = im1[0 * 64 + lane]; A1 = im1[1 * 64 + lane];
        f16v l0 = dn;                                                      // Lambda^T of tile gt -> p
#pragma unroll
        for (int v = 0; v < 16; ++v) dn[v] = 0.f;
        __builtin_amdgcn_sched_barrier(0);

        // ================= stage A: the matrix instructions of Lambda(gt + 1), one per slot; beside them the sigmoid of
        // tile gt, its way out, the splits of p and the tail products of DV
        f4v tq[4];
        u4v a2[2][3];
        uint32_t sh = 0, sm = 0, sl = 0;
        float tl0 = 0.f, tl2 = 0.f;
        const f4v *tails = reinterpret_cast<const f4v *>(im0 + C::P1 + C::P2);
        f4v t2 = {0.f, 0.f, 0.f, 0.f};
        f4v lg4 = {0.f, 0.f, 0.f, 0.f}, fl4 = {0.f, 0.f, 0.f, 0.f};
        const int nzw = (int)reinterpret_cast<const uint16_t *>(im0 + MP)[w * 64 + lane];      // this lane's 16 non-zero flags
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            if (u < KC * 6) {
                const int kc = u / 6, p = u % 6;
                const u4v aop = (PA[p] == 0) ? A0[kc & 1] : (PA[p] == 1) ? A1 : A2;
                dn = mfma_b16(aop, ub[kc][PB[p]], dn);
                if (kc + 1 < KC) {
                    if (p == 0) A0[(kc + 1) & 1] = im1[((kc + 1) * 3 + 0) * 64 + lane];
                    if (p == 1) A2 = im1[((kc + 1) * 3 + 2) * 64 + lane];
                    if (p == 4) A1 = im1[((kc + 1) * 3 + 1) * 64 + lane];
                } else if (TAIL) {
                    const float *tl = reinterpret_cast<const float *>(im1 + C::P1 + C::P2) + c * 4 + h;
                    if (p == 0) tl0 = tl[0];
                    if (p == 1) tl2 = tl[2];
                }
            } else if (u == KC * 6) {
                dn = __builtin_amdgcn_mfma_f32_32x32x2f32(tl0, futb0, dn, 0, 0, 0);
            } else {
                dn = __builtin_amdgcn_mfma_f32_32x32x2f32(tl2, futb1, dn, 0, 0, 0);
            }
            // ---- vector work of the slot: items 0..15 = p of value v, 16 = read-back of the transposed tile + the next
            // copies, 17 = D_hat rows out + column sums, 18..33 = split (+ DV's tail products) of value vv
            constexpr int NITEM = 34;
#pragma unroll
            for (int it = (u * NITEM) / NA; it < ((u + 1) * NITEM) / NA; ++it) {
                if (it < 16) {
                    const int v = it, q = v >> 2;
                    if ((v & 3) == 0) {              // scaled logits and floors of the genes 8 q + 4 h .. + 3
                        lg4 = __builtin_bit_cast(f4v, im0[MP + 64 + 2 * q + h]);
                        fl4 = __builtin_bit_cast(f4v, im0[MP + 64 + 8 + 2 * q + h]);
                    }
                    // lgs = -logit(pi_d) log2(e) (k_logit_f32): sigmoid(logit - Lambda) = 1 / (1 + exp2(Lambda log2(e) + lgs))
                    const float lgs = lg4[v & 3];
                    float p = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fmaf(l0[v], 1.4426950408889634f, lgs)));
                    // pi_d <= 0: lgs = +inf -> exp2 = inf -> p = +0, and the gene's floor is 1e-10; every other gene: + 0 (zigap.py:133)
                    p += fl4[v & 3];
                    {
                        // X != 0: f32(1 - 1e-10) == 1 (zigap.py:135).  (inline assembly: the compiler turns the same two
                        // operations written in C into shift + compare + select + and + or)
                        uint32_t sel;
                        asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(sel) : "v"(nzw), "n"(v));            // flag v -> 0 or ~0
                        asm("v_bfi_b32 %0, %1, 1.0, %2" : "=v"(p) : "v"(sel), "v"(p));            // (sel & 1.0f) | (~sel & p)
                    }
                    l0[v] = p;
                    if ((v & 3) == 3)
                        *reinterpret_cast<__attribute__((address_space(3))) f4v *>(Tw0 ^ (uint32_t)(q << 5)) =
                            f4v{l0[v - 3], l0[v - 2], l0[v - 1], l0[v]};
                } else if (it == 16) {
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int q = 0; q < 4; ++q) tq[q] = *reinterpret_cast<const f4v *>(Tr + 8 * q * 32);
                    if (!DMA_TOP)
                        zi_tile_dma<KC, TAIL>(imgV, nztiles, lgit, mpad, img + bufnn * C::PVZ, (gt + 2 < gt1) ? gt + 2 : gt1 - 1, ct_blk0, ngt, m,
                                  w, lane);
                } else if (it == 17) {
                    // D_hat rows out
                    const uint32_t vo = (j0 + gq < m) ? dvoff : 0x80000000u;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, tq[q]), drsrc, vo,
                                                               (int)((j0 + (int64_t)8 * q * m) * 4), 0);
                    // sum_i p_d of the wave's 32 cells: lanes with the same lane % 8 hold the same four genes
                    // (padding cells -- only the matrix's last cell tile has any -- enter with weight 0)
                    f4v cs4 = tq[0] * (0 < rleft ? 1.f : 0.f);
                    cs4 = tq[1] * (8 < rleft ? 1.f : 0.f) + cs4;
                    cs4 = tq[2] * (16 < rleft ? 1.f : 0.f) + cs4;
                    cs4 = tq[3] * (24 < rleft ? 1.f : 0.f) + cs4;
                    cs4.x = sum_mod8(cs4.x); cs4.y = sum_mod8(cs4.y); cs4.z = sum_mod8(cs4.z); cs4.w = sum_mod8(cs4.w);
                    *reinterpret_cast<f4v *>(csb + (par * NW + w) * 32 + gq) = cs4;      // (eight lanes, the same value)
                } else {
                    const int vv = it - 18;
                    const float x0 = l0[vv];
                    const uint32_t b0 = __float_as_uint(x0);
                    const float r0 = x0 - __uint_as_float(b0 & 0xFFFF0000u);
                    const uint32_t c0 = __float_as_uint(r0);
                    const float s0 = r0 - __uint_as_float(c0 & 0xFFFF0000u);
                    if (TAIL && !TIB) {
                        // (the B operands of four values at a time: 4 registers in flight instead of 16)
                        if ((vv & 3) == 0) t2 = tails[32 + (h * 4 + (lane & 3)) * 4 + (vv >> 2)];
                        rt = __builtin_amdgcn_mfma_f32_4x4x1f32(x0, t2[vv & 3], rt, 0, 0, 0);
                    }
                    if ((vv & 1) == 0) { sh = b0; sm = c0; sl = __float_as_uint(s0); }
                    else {
                        const int q = vv >> 3, w2 = (vv & 7) >> 1;
                        a2[q][0][w2] = __builtin_amdgcn_perm(b0, sh, 0x07060302u);
                        a2[q][1][w2] = __builtin_amdgcn_perm(c0, sm, 0x07060302u);
                        a2[q][2][w2] = __builtin_amdgcn_perm(__float_as_uint(s0), sl, 0x07060302u);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }

        // ================= stage B: DV += D_hat V_next for tile gt
        u4v B0[2], B1, B2;
        B0[0] = im0[C::P1 + 0 * 64 + lane]; B2 = im0[C::P1 + 2 * 64 + lane]; B1 = im0[C::P1 + 1 * 64 + lane];
        f16v dv;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int g = u / 6, p = u % 6, nt = g >> 1, q = g & 1;
            const u4v bop = (PB[p] == 0) ? B0[g & 1] : (PB[p] == 1) ? B1 : B2;
            if (u % 12 == 0) {
                f16v z;
#pragma unroll
                for (int v = 0; v < 16; ++v) z[v] = 0.f;
                dv = mfma_b16(a2[q][PA[p]], bop, z);
            } else {
                dv = mfma_b16(a2[q][PA[p]], bop, dv);
            }
            if (g + 1 < 2 * NT) {
                const int nt1 = (g + 1) >> 1, q1 = (g + 1) & 1;
                const u4v *src = im0 + C::P1 + ((nt1 * 2 + q1) * 3) * 64 + lane;
                if (p == 0) B0[(g + 1) & 1] = src[0 * 64];
                if (p == 2) B2 = src[2 * 64];
                if (p == 5) B1 = src[1 * 64];
            }
            __builtin_amdgcn_sched_barrier(0);
            if (u % 12 == 11) {
#pragma unroll
                for (int v = 0; v < 16; ++v) rs[nt][v] += dv[v];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        buf = bufn;
        par ^= 1;
    }
    colsum_flush(gt1 - 1, par ^ 1);
    // ---- out: DV[cell, k] += the strip's sums (float64, zeroed by the caller)
    if (DV) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int64_t cell = ct * 32 + acc_row(v, h);
                const int k = nt * 32 + c;
                if (cell < n && k < (TIB ? C::KM + 4 : C::KM) && k < K) atomicAdd(&DV[cell * K + k], (double)rs[nt][v]);
            }
        if (TAIL && !TIB) {
            rt.x += __shfl_xor(rt.x, 32, 64); rt.y += __shfl_xor(rt.y, 32, 64);
            rt.z += __shfl_xor(rt.z, 32, 64); rt.w += __shfl_xor(rt.w, 32, 64);
            if (h == 0) {
                const int k = C::KM + (lane & 3);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int64_t cell = ct * 32 + 4 * (lane >> 2) + e;
                    if (cell < n && k < K) atomicAdd(&DV[cell * K + k], (double)rt[e]);
                }
            }
        }
    }
}

// ---- D_hat^T W ------------------------------------------------------------------------------------------------------
// dn::k_dn_col with the tile of D_hat copied from the row-major matrix: lane l of copy q fetches 16 bytes of cell row
// 8 q + (s >> 1) + 4 (s & 1), s = l / 8 -- the rows the two lane halves read for one register land in LDS rows of
// different parity, i.e. on different banks.
template <int KC, int TAIL>
constexpr int zi_col_lds_bytes() { return 3 * Cfg<KC, TAIL>::PU * 16 + 2 * NW * 256 * 16; }

template <int KC, int TAIL>
__global__ __launch_bounds__(512) void k_zi_col(const float *__restrict__ D, const u4v *__restrict__ imgU,
                                                double *__restrict__ out, int64_t n, int64_t m, int K, int64_t nct,
                                                int ngt, int64_t ct_per_split, int nsplit) {
    using C = Cfg<KC, TAIL>;
    constexpr int NT = C::NT;
    extern __shared__ u4v ldsq[];
    u4v *img = ldsq;                                                      // [3][PU]
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    u4v *sbuf = ldsq + 3 * C::PU + w * 256;                               // [2][8 waves][256 pieces]
    const int split = blockIdx.x % nsplit, grp = blockIdx.x / nsplit;
    const int gt = grp * NW + w;
    const bool active = gt < ngt;
    const int gtc = active ? gt : ngt - 1;
    const int64_t ct0 = (int64_t)split * ct_per_split;
    const int64_t ct1 = (ct0 + ct_per_split < nct) ? ct0 + ct_per_split : nct;
    if (ct0 >= ct1) return;

    f16v cs[NT], dv[NT];
    f4v cta = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int v = 0; v < 16; ++v) { cs[nt][v] = 0.f; dv[nt][v] = 0.f; }

    const int srow = lane >> 3;                                            // LDS row of this lane inside an 8-row copy
    const int drow = (srow >> 1) + 4 * (srow & 1);                         // the cell row (inside the 8) it holds
    // rows / genes beyond the matrix: the copy is CLAMPED to the last row / the last 16-byte piece, never predicated --
    // every wave issues the same number of copies (the waits below count them), and what lands there is a real, finite
    // D_hat value that only ever meets a zero operand (image rows beyond n) or a discarded output (genes beyond m)
    int64_t jpiece = (int64_t)gtc * 32 + 4 * (lane & 7);
    if (jpiece > m - 4) jpiece = m - 4;
    auto issue = [&](int64_t ct, int islot, int sslot) {
        const int64_t cc = ct < ct1 ? ct : ct1 - 1;
        image_dma<C::PU>(imgU + cc * C::PU, img + islot * C::PU, w, lane);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int64_t row = cc * 32 + 8 * q + drow;
            if (row > n - 1) row = n - 1;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(D + row * m + jpiece),
                                             (__attribute__((address_space(3))) void *)(sbuf + sslot * 2048 + q * 64), 16, 0, 0);
        }
    };
    constexpr int NCOPY = C::PU / (NW * 64) + 4;                          // LDS-DMA instructions per wave and tile
    static_assert(NCOPY >= 5 && NCOPY <= 7, "the waits below count the copies of one iteration");
    auto wait_prev = [&]() {             // the previous iteration's copies have landed, this iteration's may be in flight
        if (NCOPY == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        else if (NCOPY == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    };
    issue(ct0, 0, 0);
    issue(ct0 + 1, 1, 1);
    wait_prev();
    __builtin_amdgcn_s_barrier();
    int islot = 0, sslot = 0, since = 0;
    // register v = 4 q + e of lane (g, h) is d[cell acc_row(v, h) = 8 q + 4 h + e][gene g]: LDS row 8 q + 2 e + h
    const float *sread = reinterpret_cast<const float *>(sbuf) + h * 32 + c;
    for (int64_t ct = ct0; ct < ct1; ++ct) {
        const u4v *im = img + islot * C::PU;
        f4v sc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) sc[q][e] = sread[sslot * 2048 * 4 + (8 * q + 2 * e) * 32];
        u4v a2[2][3];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = sc[(8 * q + e) >> 2][(8 * q + e) & 3];
            split8(x, a2[q]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the tile is in registers before its ring slot is refilled)
        issue(ct + 2, (islot == 0) ? 2 : islot - 1, sslot);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                u4v b[3];
#pragma unroll
                for (int sp = 0; sp < 3; ++sp) b[sp] = im[((nt * 2 + q) * 3 + sp) * 64 + lane];
                ORIANA_DN_MF6(dv[nt], a2[q], b);
            }
        }
        if (TAIL) {
            const f4v *t2p = reinterpret_cast<const f4v *>(im + C::P2 + 32) + (h * 4 + (lane & 3)) * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f4v t2 = t2p[q];
#pragma unroll
                for (int e = 0; e < 4; ++e) cta = __builtin_amdgcn_mfma_f32_4x4x1f32(sc[q][e], t2[e], cta, 0, 0, 0);
            }
        }
        if (++since == 8) {              // 256 cells: leave the matrix core
            since = 0;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int v = 0; v < 16; ++v) { cs[nt][v] += dv[nt][v]; dv[nt][v] = 0.f; }
        }
        wait_prev();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        islot = (islot == 2) ? 0 : islot + 1;
        sslot ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!active) return;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int64_t gene = (int64_t)gt * 32 + acc_row(v, h);
            const int k = nt * 32 + c;
            if (gene < m && k < C::KM && k < K) atomicAdd(&out[gene * K + k], (double)cs[nt][v] + (double)dv[nt][v]);
        }
    if (TAIL) {
        cta.x += __shfl_xor(cta.x, 32, 64); cta.y += __shfl_xor(cta.y, 32, 64);
        cta.z += __shfl_xor(cta.z, 32, 64); cta.w += __shfl_xor(cta.w, 32, 64);
        if (h == 0) {
            const int k = C::KM + (lane & 3);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int64_t gene = (int64_t)gt * 32 + 4 * (lane >> 2) + e;
                if (gene < m && k < K) atomicAdd(&out[gene * K + k], (double)cta[e]);
            }
        }
    }
}

template <typename KernelT>
static int zi_set_lds(KernelT kern, size_t bytes) {
    if (bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)bytes);
        if (e != hipSuccess) return -1000 - (int)e;
    }
    return 0;
}

// work-groups of 512 threads, one per CU: the split count that leaves the last round of the chip fullest (fewer on ties)
static int64_t zi_pick_splits(int64_t blocks, int64_t max_splits) {
    const int64_t cus = oriana_device_cus();                 // (256 on MI355X)
    int64_t best = 1;
    double best_eff = 0.0;
    for (int64_t sp = 1; sp <= max_splits && sp <= 64; ++sp) {
        const int64_t groups = blocks * sp;
        if (groups > 16 * cus && sp > 1) break;
        const int64_t rounds = (groups + cus - 1) / cus;
        const double eff = (double)groups / (double)(rounds * cus);
        if (eff > best_eff + 0.03) { best_eff = eff; best = sp; }
    }
    return best;
}

#define ORIANA_ZI_FOR_CFG(KC_, TL_, CALL)                                     \
    do {                                                                      \
        if (KC_ == 6 && TL_ == 1) { CALL(6, 1); }                             \
        else if (KC_ == 6 && TL_ == 0) { CALL(6, 0); }                        \
        else if (KC_ == 5 && TL_ == 1) { CALL(5, 1); }                        \
        else if (KC_ == 5 && TL_ == 0) { CALL(5, 0); }                        \
        else if (KC_ == 4 && TL_ == 1) { CALL(4, 1); }                        \
        else if (KC_ == 4 && TL_ == 0) { CALL(4, 0); }                        \
        else if (KC_ == 3 && TL_ == 1) { CALL(3, 1); }                        \
        else if (KC_ == 3 && TL_ == 0) { CALL(3, 0); }                        \
        else return ORIANA_EKRANGE;                                           \
    } while (0)

// Smallest K served here: 33.  Measured at 100k x 20k against dense_f32.hip's bf16 kernels (tools/perf_zi_per_k.py, round 6's
// build): D update 3.37 against 4.11 ms at K = 48, 3.65 against 4.00 ms at K = 50, 3.72 against 4.11 ms at K = 64; D^T U 1.82
// against 2.00 ms at K = 48 but 2.06 against 1.98 ms at K = 50 WITH the tail factors' 4 x 4 x 1 instructions: below 65 the
// transposed product takes the next whole chunk instead (zi_cfg_dt).
static int64_t zi_min_k() { return 33; }

static bool zi_cfg(int64_t K, int *kc, int *tl) {
    const int64_t Kp = oriana_kpad(K);
    if (K < zi_min_k() || Kp == 0 || Kp > 100 || (Kp % 16 != 0 && Kp % 16 != 4)) return false;
    *kc = (int)(Kp / 16); *tl = (Kp % 16 == 4) ? 1 : 0;
    // Kp = 48 .. 100.  [r6] Kp = 64 (K = 53 .. 64) too: with room for the flag / logit pieces behind the images (Cfg::PVZ) and the
    // copies issued at the top of the tile, k_zi_row<4, 0> takes 3.68 ms where dense_f32.hip's bf16 kernel takes 4.11
    // (K = 49 .. 52 stays on <3, 1> with the tail riding in the last factor tile: 3.65 against 3.72 ms zero-padded to four chunks)
    return *kc >= 3 && *kc <= 6;
}

bool zi_supported(int64_t m, int64_t K) {                                  // the D update
    int kc, tl;
    return (m % 4) == 0 && zi_cfg(K, &kc, &tl);
}

// [r6] D_hat^T W below K = 65: Kp = 36 and 52 (K = 50: configs[2]) take the NEXT whole chunk of 16 with zero-padded factors instead
// of the tail of four -- k_zi_col<4, 0> 1.50 ms against 1.67 ms for k_dt_times_factor_b16 inside the configs[2] sweep (the D update
// itself is faster WITH the tail: 3.49 against 3.94 ms; profiles/r06_zi_k52_ab.txt)
static bool zi_cfg_dt(int64_t K, int *kc, int *tl) {
    const int64_t Kp = oriana_kpad(K);
    if (K < zi_min_k() || Kp == 0 || Kp > 100 || (Kp % 16 != 0 && Kp % 16 != 4)) return false;
    *kc = (int)(Kp / 16); *tl = (Kp % 16 == 4) ? 1 : 0;
    if (*tl && *kc <= 3) { *kc += 1; *tl = 0; }
    return *kc >= 3 && *kc <= 6;
}

bool zi_dt_supported(int64_t m, int64_t K) {                               // D_hat^T W
    int kc, tl;
    return (m % 4) == 0 && zi_cfg_dt(K, &kc, &tl) && (K > 64 || tl == 0);
}

// floats of scratch for the gene-side images of m genes / the cell-side images of n cells (largest configuration)
int64_t zi_sweep_image_floats(int64_t m) { return ((m + 31) / 32) * (int64_t)Cfg<6, 1>::PVZ * 4; }
int64_t zi_dt_image_floats(int64_t n) { return ((n + 31) / 32 + 2) * (int64_t)Cfg<6, 1>::PU * 4; }

// [r6] The non-zero flags of a (cell tile, gene tile) pair in the order k_zi_row's lanes hold their values: 64 x 16 bits
// (zi_tile_dma), for ceil(n / 256) x 8 cell tiles (whole work-groups: tiles beyond the matrix are zero) x ceil(m / 32) gene
// tiles x 128 bytes.  Built once per count matrix (the mask is constant) from the oriana_nzmask_f32 layout.
__global__ void k_nzmask_tiles(uint32_t *__restrict__ out, const uint32_t *__restrict__ nzmask, int64_t nct, int64_t nct8,
                               int64_t m, int ngt) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // one dword each
    if (t >= nct8 * ngt * 32) return;
    const int d = (int)(t & 31);
    const int64_t tile = t >> 5, ct = tile / ngt;
    const int gt = (int)(tile - ct * ngt);
    uint32_t o = 0u;
    if (ct < nct) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {                             // dword d = lanes 2 d, 2 d + 1
            const int l = 2 * d + half, c = l & 31, h = l >> 5;
            uint32_t f = 0u;
            for (int v = 0; v < 16; ++v) {
                const int64_t j = (int64_t)gt * 32 + 8 * (v >> 2) + 4 * h + (v & 3);
                if (j < m) f |= ((nzmask[ct * m + j] >> c) & 1u) << v;
            }
            o |= f << (16 * half);
        }
    }
    out[t] = o;
}

int64_t zi_tiles_words(int64_t n, int64_t m) { return (((n + 31) / 32 + NW - 1) / NW * NW) * ((m + 31) / 32) * 32; }

int zi_tiles(uint32_t *out, const uint32_t *nzmask, int64_t n, int64_t m, hipStream_t st) {
    const int64_t words = zi_tiles_words(n, m);
    if (words == 0) return 0;
    if (words > 0x7fffffffLL * 256) return ORIANA_EINVAL;
    const int64_t nct = (n + 31) / 32;
    hipLaunchKernelGGL(k_nzmask_tiles, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st, out, nzmask, nct,
                       (nct + NW - 1) / NW * NW, m, (int)((m + 31) / 32));
    return 0;
}

// lgit: [0, mpad) the scaled logits -logit(pi_d) log2(e), [mpad, 2 mpad) the floors (k_logit_f32)
int zi_sweep(float *D_hat, const double *U, const double *V, const float *lgit, int64_t mpad, const uint32_t *nztiles,
             double *colsum, const double *Vn, double *DV, float *img_scratch, int64_t n, int64_t m, int K, hipStream_t st) {
    int kc, tl;
    if (!zi_cfg(K, &kc, &tl) || (m % 4) != 0 || m > 16000000 || !Vn || !DV || !nztiles) return ORIANA_EKRANGE;
    const int ngt = (int)((m + 31) / 32);
    const int64_t blocks = ((n + 31) / 32 + NW - 1) / NW;
    int64_t splits = zi_pick_splits(blocks, (ngt + 7) / 8);
    const int per = (int)((ngt + splits - 1) / splits);
    splits = (ngt + per - 1) / per;
    if (blocks > 0x7fffffffLL || splits > 65535) return ORIANA_EINVAL;
    u4v *img = reinterpret_cast<u4v *>(img_scratch);
    if (((reinterpret_cast<uintptr_t>(D_hat) | reinterpret_cast<uintptr_t>(lgit) | (uintptr_t)(mpad * 4) |
          reinterpret_cast<uintptr_t>(nztiles)) & 15) != 0 || (int64_t)ngt * 8 * 128 > 0x7fffffffLL)
        return ORIANA_EKRANGE;                                             // 16-byte pieces: the caller falls back
#define ORIANA_ZI_CALL(KC, TL)                                                                                              \
    do {                                                                                                                    \
        hipLaunchKernelGGL((k_zi_images<KC, TL, true>), dim3((unsigned)ngt), dim3(512), 0, st, img, V, Vn, m, K);           \
        constexpr int lb = zi_row_lds_bytes<KC, TL>();                                                                      \
        const int rc = zi_set_lds(k_zi_row<KC, TL>, lb);                                                                    \
        if (rc) return rc;                                                                                                  \
        hipLaunchKernelGGL((k_zi_row<KC, TL>), dim3((unsigned)blocks, (unsigned)splits), dim3(512), lb, st, D_hat, U,       \
                           (const u4v *)img, lgit, mpad, nztiles, colsum, DV, n, m, K, ngt, per);                           \
    } while (0)
    ORIANA_ZI_FOR_CFG(kc, tl, ORIANA_ZI_CALL);
#undef ORIANA_ZI_CALL
    return 0;
}

int zi_dt(double *out, const float *D, const double *W, float *scratch, int64_t n, int64_t m, int K, hipStream_t st) {
    int kc, tl;
    if (!zi_dt_supported(m, K) || !zi_cfg_dt(K, &kc, &tl)) return ORIANA_EKRANGE;
    if ((reinterpret_cast<uintptr_t>(D) & 15) != 0) return ORIANA_EKRANGE;   // 16-byte row pieces: the caller falls back
    const int ngt = (int)((m + 31) / 32);
    const int64_t nct = (n + 31) / 32;
    const int64_t groups = (ngt + NW - 1) / NW;
    int64_t splits = zi_pick_splits(groups, (nct + 15) / 16);
    const int64_t per = (nct + splits - 1) / splits;
    splits = (nct + per - 1) / per;
    if (splits * groups > 0x7fffffffLL) return ORIANA_EINVAL;
    u4v *img = reinterpret_cast<u4v *>(scratch);
#define ORIANA_ZI_CALL(KC, TL)                                                                                              \
    do {                                                                                                                    \
        hipLaunchKernelGGL((k_zi_images<KC, TL, false>), dim3((unsigned)nct), dim3(512), 0, st, img, (const double *)nullptr, \
                           W, n, K);                                                                                        \
        constexpr int lb = zi_col_lds_bytes<KC, TL>();                                                                      \
        const int rc = zi_set_lds(k_zi_col<KC, TL>, lb);                                                                    \
        if (rc) return rc;                                                                                                  \
        hipLaunchKernelGGL((k_zi_col<KC, TL>), dim3((unsigned)(splits * groups)), dim3(512), lb, st, D, (const u4v *)img,   \
                           out, n, m, K, nct, ngt, per, (int)splits);                                                       \
    } while (0)
    ORIANA_ZI_FOR_CFG(kc, tl, ORIANA_ZI_CALL);
#undef ORIANA_ZI_CALL
    return 0;
}

}  // namespace dn
}  // namespace oriana

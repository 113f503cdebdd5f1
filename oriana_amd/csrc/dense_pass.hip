// dense_pass.hip -- the responsibility pass (gap.py:67-80) over the DENSEST genes, evaluated densely on the bf16
// matrix cores in float32-equivalent arithmetic.
//
// Genes are packed in decreasing order of their non-zero count; the first `gd` packed genes (a multiple of 32, chosen
// by the host from a density threshold) are stored as a dense uint16 count block instead of the sliced non-zero layout,
// and their share of the pass
//     den_ij = sum_k FU[i,k] FV[j,k],   s_ij = x_ij / den_ij,
//     R[i,k] += sum_j s_ij FV[j,k]      (row side, gap.py:79),     C[j,k] += sum_i s_ij FU[i,k]   (gene side, gap.py:80)
// runs as three matrix products.  Every float32 operand is hi + mid + lo (three bf16 = its 24 bits, split by
// truncation, so the decomposition is exact) and six of the nine cross products go through
// v_mfma_f32_32x32x16_bf16 with float32 accumulation -- the evaluation csrc/dense_f32.hip uses for the ZI sweep
// (tools/ubench/mfma_bf16x3.hip: the error of a sum of <= 512 terms is that of the float32 FMA chain).  No sum stays
// on the matrix core beyond one tile: den is a K-term sum, R and C leave the accumulator after every 32 genes / cells
// and are carried on in float32 registers.  The last Kp - 16 KC factors (Kp = 16 KC + 4: K = 100 -> 96 + 4) are
// evaluated on the vector ALU in plain float32, so the matrix instructions carry no padding at the benchmark K.
//
//   k_dn_row  (row side)   work-group = 8 waves x 32 cells, all dense gene tiles in turn.  den^T[gene, cell] = FV FU^T --
//             the TRANSPOSED product, so that after s = x / den the accumulator registers are already the A operand
//             of R += S FV (register v of lane half h is gene 8 (v / 4) + 4 h + v % 4; the reduction visits the
//             genes in that order).  s goes to HBM through a 32 x 32 LDS transpose, in the register order of k_dn_col.
//   k_dn_col  (gene side)  wave = 32 genes, a range of cell tiles: reads s (4 B per entry, 16 bytes per lane and
//             load), C += S^T FU with the FU operand images staged through LDS.
//   entries whose den fails the den >= DEN_MIN test (or that touch a rejected factor row) get the NaN sentinel and
//   are evaluated exactly by k_dn_fixup, as on the sparse side (passes.hip k_fixup).
#include "dense_tiles.h"
// Compile-time ablation switches for the measurements quoted in DESIGN.md (never set in the shipped build; they give
// wrong results): ORIANA_DN_ABL_NODMA / _NOSTORE / _NOX / _NOBAR drop one ingredient of k_dn_row's loop.

namespace oriana {
namespace dn {

#define DN_STAMP(K) do { } while (0)

// One work-group per tile of 32 factor rows (genes or cells); F is a padded (rows, Kp) float32 factor matrix.  F2: the
// matrix the SECOND image (the B operand of the accumulation, and its tail pieces) is taken from -- the sparse models
// accumulate R against FV * S_hat while den runs against the masked FV (sparse_gap.py:88-95); F2 = F otherwise.
template <int KC, int TAIL, bool BOTH>
__global__ __launch_bounds__(512) void k_dn_images(u4v *__restrict__ img, const float *__restrict__ F,
                                                   const float *__restrict__ F2, int64_t rows, int Kp) {
    using C = Cfg<KC, TAIL>;
    constexpr int PIMG = BOTH ? C::PV : C::PU;
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * 32;
    u4v *dst0 = img + (int64_t)blockIdx.x * PIMG;
    if (BOTH) {
        const int g = tid & 31, G = tid >> 5;
        if (G < 2 * KC) {
            const int64_t r = r0 + g;
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = (r < rows) ? F[r * Kp + 8 * G + e] : 0.f;
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
                x[e] = (r < rows && kk < C::KM) ? F2[r * Kp + kk] : 0.f;
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
        if (tid < 32) {                              // pieces 0..31: the four tail factors of row tid
            const int64_t r = r0 + tid;
            if (r < rows) t = *reinterpret_cast<const f4v *>(F + r * Kp + C::KM);
        } else {                                     // pieces 32..63: [lane half hh][tail factor j][4 q .. 4 q + 3]: the
            const int idx = tid - 32, hh = idx >> 4, j = (idx >> 2) & 3, q = idx & 3;   // rows in accumulator order
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int64_t r = r0 + acc_row(4 * q + e, hh);
                t[e] = (r < rows) ? F2[r * Kp + C::KM + j] : 0.f;
            }
        }
        dst0[(BOTH ? C::P1 : 0) + C::P2 + tid] = __builtin_bit_cast(u4v, t);
    }
}

// ---- row side ------------------------------------------------------------------------------------------------------
// Xd: [cell tile][gene tile][1024] uint16 in THIS kernel's register order: value v of lane l = (c, h) is
//     x[cell c][gene acc_row(v, h)], stored at ((v / 8) * 64 + l) * 8 + v % 8.
// S : [cell tile][gene tile][1024] float32 in k_dn_col's register order: value v' of lane l' = (g, h') is
//     s[cell acc_row(v', h')][gene g], stored at ((v' / 4) * 64 + l') * 4 + v' % 4.
//
// Schedule.  A tile has three phases per wave: D (36 matrix instructions: den), S (vector ALU only: s = x / den, the
// LDS transpose, the stores) and R (the bf16 splits of s, 36 matrix instructions: R += S FV).  The phases are
// software-pipelined inside every wave: iteration t runs S(t) -- vector work -- beside the matrix instructions of
// D(t + 1), then R(t); the loop body is one basic block (no branch: the last iteration recomputes D of the last
// tile), so the scheduler is free to interleave the two streams.  The images live in a ring of THREE buffers (tile t
// for R, tile t + 1 for D, tile t + 2 arriving by LDS-DMA).  The four tail factors of den go through
// v_mfma_f32_32x32x2_f32 (exact float32 FMAs).
template <int KC, int TAIL>
__global__ __launch_bounds__(512) void k_dn_row(const uint16_t *__restrict__ Xd, float *__restrict__ S,
                                                const float *__restrict__ FU, const u4v *__restrict__ imgV,
                                                float *__restrict__ R, int32_t *__restrict__ flag, int64_t n, int ngt,
                                                int Kp, int gt_per_split, int atomic_out, int tail_nfull, int tail_parts,
                                                const float *__restrict__ den_min_p) {
    const float den_min = den_min_p ? *den_min_p : DEN_MIN;              // (passes.hip, k_row_stats: the den threshold)
    using C = Cfg<KC, TAIL>;
    constexpr int NT = C::NT;
    extern __shared__ u4v ldsq[];
    u4v *img = ldsq;                                                      // [3][PV]
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);              // (wave-uniform: addresses built from it stay scalar)
    float *T = reinterpret_cast<float *>(ldsq + 3 * C::PV) + w * 32 * TS;  // [wave][32 genes][TS]
    // tail_parts > 1 (struct oriana_row_split, as the two-lane sliced kernels): the 256-cell blocks [0, tail_nfull) are one
    // work-group each; every later block is tail_parts groups over even gene-tile ranges, part p adding into slab p of R
    // (plain read-modify-write: one group per (row, slab)) -- the last, partly filled round of the chip runs shorter groups
    int blk = (int)blockIdx.x, gt0 = blockIdx.y * gt_per_split;
    int gt1 = (gt0 + gt_per_split < ngt) ? gt0 + gt_per_split : ngt;
    if (tail_parts > 1) {
        gt0 = 0; gt1 = ngt;
        if (blk >= tail_nfull) {
            const int idx = blk - tail_nfull, q = idx / tail_parts, part = idx - q * tail_parts;
            blk = tail_nfull + q;
            gt0 = (int)((int64_t)part * ngt / tail_parts); gt1 = (int)(((int64_t)part + 1) * ngt / tail_parts);
            R += (int64_t)part * (n - (int64_t)tail_nfull * 256) * Kp;      // (compact slabs: the rows from the first split block on)
        }
    }
    const int64_t ct = (int64_t)blk * NW + w;                              // this wave's cell tile
    const int64_t i = ct * 32 + c;
    if (gt0 >= gt1) return;

    // the wave's strip of FU as the B operand of the first product: per k chunk, factors 16 kc + 8 h + e of cell c
    u4v ub[KC][3];
    f4v fut = {0.f, 0.f, 0.f, 0.f};
    {
        const float *urow = FU + (i < n ? i : 0) * Kp;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            float x[8];
            const f4v a = *reinterpret_cast<const f4v *>(urow + 16 * kc + 8 * h);
            const f4v b = *reinterpret_cast<const f4v *>(urow + 16 * kc + 8 * h + 4);
            x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
            if (i >= n) {                 // padding cells: FU = 1, so den = sum_k FV > 0 passes the test and s = 0 / den = 0
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = 1.f;
            }
            split8(x, ub[kc]);
        }
        if (TAIL) fut = (i < n) ? *reinterpret_cast<const f4v *>(urow + C::KM) : f4v{1.f, 1.f, 1.f, 1.f};
    }
    // B operands of the two float32 tail instructions: B[k = h][n = c] = FU[c][KM + h], then KM + 2 + h
    const float futb0 = h ? fut.y : fut.x, futb1 = h ? fut.w : fut.z;

    f16v rs[NT];                         // R of the strip: [cell acc_row(v, h)][factor 32 nt + c]
    f4v rt = {0.f, 0.f, 0.f, 0.f};       // tail factors: 4 x 4 blocks (cells x factors), this half's genes
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int v = 0; v < 16; ++v) rs[nt][v] = 0.f;

    const uint16_t *xrow = Xd + (ct * ngt) * 1024;
    float *srow = S + (ct * ngt) * 1024;

    // product order of the six cross terms of a k step, small ones first: (part of A, part of B)
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
    float *Tw = T + 4 * h * TS + c;                                        // + (8 (v / 4) + v % 4) TS: s[gene acc_row(v, h)][cell c]
    const float *Tr = T + c * TS + 4 * h;                                  // + 8 q: row of gene c, cells 8 q + 4 h ..

    // den of one tile, outside the loop (first tile only)
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

    u4v xn[2];
    image_dma<C::PV>(imgV + (int64_t)gt0 * C::PV, img, w, lane);
    {
        const int g1 = (gt0 + 1 < gt1) ? gt0 + 1 : gt0;
        image_dma<C::PV>(imgV + (int64_t)g1 * C::PV, img + C::PV, w, lane);
    }
    xn[0] = reinterpret_cast<const u4v *>(xrow + (int64_t)gt0 * 1024)[lane];
    xn[1] = reinterpret_cast<const u4v *>(xrow + (int64_t)gt0 * 1024)[64 + lane];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f16v dn = phase_D(img);              // den of tile gt0
    int buf = 0;                         // ring position of tile gt
    uint32_t fl = 0;                     // slow-path flags of the wave's tiles (see item 17)
    constexpr int NA = KC * 6 + (TAIL ? 2 : 0);                            // matrix instructions of D
    constexpr int NB = NT * 12;                                            // ... of R
    for (int gt = gt0; gt < gt1; ++gt) {
        const int bufn = (buf == 2) ? 0 : buf + 1, bufnn = (buf == 0) ? 2 : buf - 1;
        const u4v *im1 = img + bufn * C::PV;                               // tile gt + 1: D
        const u4v *im0 = img + buf * C::PV;                                // tile gt: R
        // first operands of D(gt + 1)
        u4v A0[2], A1, A2;
        A2 = im1[2 * 64 + lane]; A0[0] = im1[0 * 64 + lane]; A1 = im1[1 * 64 + lane];
        f16v l0 = dn;                                                      // den of tile gt -> s
#pragma unroll
        for (int v = 0; v < 16; ++v) dn[v] = 0.f;
        __builtin_amdgcn_sched_barrier(0);

        // ================= stage A: the matrix instructions of D(gt + 1), one per slot; beside them S(gt), the splits of
        // s and R's tail products
        bool allok = true;
        f4v tq[4];
        u4v a2[2][3];
        uint32_t sh = 0, sm = 0, sl = 0;                                   // bf16 parts of the even value of a pair
        float tl0 = 0.f, tl2 = 0.f;
        const f4v *tails = reinterpret_cast<const f4v *>(im0 + C::P1 + C::P2);
        f4v t2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            // ---- matrix instruction
            if (u < KC * 6) {
                const int kc = u / 6, p = u % 6;
                const u4v aop = (PA[p] == 0) ? A0[kc & 1] : (PA[p] == 1) ? A1 : A2;
                dn = mfma_b16(aop, ub[kc][PB[p]], dn);
                // operands of the next k chunk, one per slot
                if (kc + 1 < KC) {
                    if (p == 0) A0[(kc + 1) & 1] = im1[((kc + 1) * 3 + 0) * 64 + lane];
                    if (p == 1) A2 = im1[((kc + 1) * 3 + 2) * 64 + lane];
                    if (p == 4) A1 = im1[((kc + 1) * 3 + 1) * 64 + lane];       // (after its last use at p = 3)
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
            // ---- vector work of the slot: items 0..15 = s of value v, 16 = read-back of the transposed tile + the
            // next copies, 17 = stores, 18..33 = split (+ R's tail products) of value vv, 34 = the next tile's counts
            constexpr int NITEM = 35;
#pragma unroll
            for (int it = (u * NITEM) / NA; it < ((u + 1) * NITEM) / NA; ++it) {
                if (it < 16) {
                    const int v = it;
                    const uint32_t wd = xn[v >> 3][(v & 7) >> 1];
                    const uint32_t xi = (v & 1) ? (wd >> 16) : (wd & 0xFFFFu);
                    const float den = l0[v];
                    const bool ok = den >= den_min;                        // false for 0, tiny and NaN
                    allok = allok && ok;
                    // branch-free: a failed test leaves the NaN sentinel in the stored tile (also where x == 0: the
                    // slow path clears it) and 0 in the registers that feed R
                    const float t = (float)xi * (ok ? __builtin_amdgcn_rcpf(den) : NAN);
                    Tw[(8 * (v >> 2) + (v & 3)) * TS] = t;
                    l0[v] = ok ? t : 0.f;
                } else if (it == 16) {
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int q = 0; q < 4; ++q) tq[q] = *reinterpret_cast<const f4v *>(Tr + 8 * q);
                    // the image of tile gt + 2 goes out now (LDS-DMA into the buffer tile gt - 1 has left) -- here, in the
                    // middle of the matrix work, not at the top of the iteration: measured 4-10 % faster (the copies then
                    // do not compete with the first operand reads after the barrier)
                    const int g2 = (gt + 2 < gt1) ? gt + 2 : gt1 - 1;
                    image_dma<C::PV>(imgV + (int64_t)g2 * C::PV, img + bufnn * C::PV, w, lane);
                } else if (it == 17) {
                    float *sblk = srow + (int64_t)gt * 1024;
#pragma unroll
                    for (int q = 0; q < 4; ++q) reinterpret_cast<f4v *>(sblk)[q * 64 + lane] = tq[q];
                    // one flag per (cell tile, gene tile): lane (t mod 64) keeps bit t / 64 of its tile t = gt - gt0 and
                    // the flags leave after the loop (no store, no branch here)
                    fl |= (__any(!allok) && lane == ((gt - gt0) & 63)) ? (1u << ((gt - gt0) >> 6)) : 0u;
                } else if (it < 34) {
                    // split of value vv (three bf16 parts, exact); pairs are packed into the A operand of R
                    const int vv = it - 18;
                    const float x0 = l0[vv];
                    const uint32_t b0 = __float_as_uint(x0);
                    const float r0 = x0 - __uint_as_float(b0 & 0xFFFF0000u);
                    const uint32_t c0 = __float_as_uint(r0);
                    const float s0 = r0 - __uint_as_float(c0 & 0xFFFF0000u);
                    if (TAIL) {
                        // R's tail factors: 16 blocks of 4 cells x 4 factors, one gene per instruction -- A[b][i] = s of cell
                        // 4 (b % 8) + i (this lane's value), B[b][j] = FV[gene][KM + j]: exact float32 FMAs on the matrix pipe
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
                } else {
                    // the counts of tile gt + 1 (this tile's were consumed by the items 0..15)
                    const int g1 = (gt + 1 < gt1) ? gt + 1 : gt1 - 1;
                    xn[0] = reinterpret_cast<const u4v *>(xrow + (int64_t)g1 * 1024)[lane];
                    xn[1] = reinterpret_cast<const u4v *>(xrow + (int64_t)g1 * 1024)[64 + lane];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (u == 15) { DN_STAMP(1); __builtin_amdgcn_sched_barrier(0); }
            if (u == 17) { DN_STAMP(2); __builtin_amdgcn_sched_barrier(0); }
        }
        DN_STAMP(3);

        // ================= stage B: the matrix instructions of R(gt) and the additions of the finished partial sums
        u4v B0[2], B1, B2;
        B0[0] = im0[C::P1 + 0 * 64 + lane]; B2 = im0[C::P1 + 2 * 64 + lane]; B1 = im0[C::P1 + 1 * 64 + lane];
        f16v dv;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int g = u / 6, p = u % 6, nt = g >> 1, q = g & 1;      // group g = (nt, q)
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
                if (p == 5) B1 = src[1 * 64];                                  // (after its last use at p = 4)
            }
            __builtin_amdgcn_sched_barrier(0);
            if (u % 12 == 11) {          // the 32-gene partial sums join the running sums (float32, round to nearest)
#pragma unroll
                for (int v = 0; v < 16; ++v) rs[nt][v] += dv[v];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        DN_STAMP(4);
        // the image copy and the counts of the next tile have landed (the stores of s, older than the counts' loads,
        // were issued 60 slots ago)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        DN_STAMP(5);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        DN_STAMP(6);
        buf = bufn;
    }
    // ---- the flags of the strip's tiles (every entry of flag[ct][gt0 .. gt1) is written)
    for (int j = 0; gt0 + 64 * j < gt1; ++j) {
        const int gt = gt0 + 64 * j + lane;
        if (gt < gt1) flag[ct * ngt + gt] = (int32_t)((fl >> j) & 1u);
    }
    // ---- out: R[cell, k] += the strip's sums
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int64_t cell = ct * 32 + acc_row(v, h);
            const int k = nt * 32 + c;
            if (cell < n && k < C::KM) {
                float *p = R + cell * Kp + k;
                if (atomic_out) atomicAdd(p, rs[nt][v]);
                else *p += rs[nt][v];
            }
        }
    if (TAIL) {
        // lane 4 b + j holds, in register e, the tail sum of cell 4 (b % 8) + e and factor KM + j over its half's genes
        rt.x += __shfl_xor(rt.x, 32, 64); rt.y += __shfl_xor(rt.y, 32, 64);
        rt.z += __shfl_xor(rt.z, 32, 64); rt.w += __shfl_xor(rt.w, 32, 64);
        if (h == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int64_t cell = ct * 32 + 4 * (lane >> 2) + e;
                if (cell < n) {
                    float *p = R + cell * Kp + C::KM + (lane & 3);
                    if (atomic_out) atomicAdd(p, rt[e]);
                    else *p += rt[e];
                }
            }
        }
    }
}

// ---- gene side -----------------------------------------------------------------------------------------------------
// grid.x = nsplit * ngroups, work-group (split, group) = blockIdx.x % nsplit, blockIdx.x / nsplit: the groups of one
// split -- which stage the same cell images -- are dispatched next to each other.
// Every global read is an LDS-DMA copy (the cell images shared by the 8 waves: ring of three; the wave's own tile of s:
// ring of two), issued TWO tiles ahead; the vector-memory counter retires in issue order, so "at most the seven copies
// of this iteration outstanding" means the previous iteration's have landed.  The partial sums stay on the matrix core
// for 8 tiles (256 cells: inside the regime where the bf16 x 3 chain carries the float32 chain's error) and then join
// the running float32 sums; the four tail factors go through v_mfma_f32_4x4x1_16B_f32 (exact float32 FMAs).
constexpr int COL_FLUSH = 8;

template <int KC, int TAIL>
__global__ __launch_bounds__(512) void k_dn_col(const float *__restrict__ S, const u4v *__restrict__ imgU,
                                                float *__restrict__ Cout, int64_t nct, int ngt, int Kp,
                                                int64_t ct_per_split, int nsplit) {
    using C = Cfg<KC, TAIL>;
    constexpr int NT = C::NT;
    extern __shared__ u4v ldsq[];
    u4v *img = ldsq;                                                      // [3][PU]
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    u4v *sbuf = ldsq + 3 * C::PU + w * 256;                               // [2][8 waves][256 pieces]: + 2048 per ring slot
    const int split = blockIdx.x % nsplit, grp = blockIdx.x / nsplit;
    const int gt = grp * NW + w;
    const bool active = gt < ngt;
    const int gtc = active ? gt : ngt - 1;
    const int64_t ct0 = (int64_t)split * ct_per_split;
    const int64_t ct1 = (ct0 + ct_per_split < nct) ? ct0 + ct_per_split : nct;
    if (ct0 >= ct1) return;

    f16v cs[NT], dv[NT];                 // C of the wave's genes: [gene acc_row(v, h)][factor 32 nt + c]; dv: on the matrix core
    f4v cta = {0.f, 0.f, 0.f, 0.f};      // tail factors: 4 x 4 blocks (genes x factors), this half's cells
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int v = 0; v < 16; ++v) { cs[nt][v] = 0.f; dv[nt][v] = 0.f; }

    auto issue = [&](int64_t ct, int islot, int sslot) {
        const int64_t cc = ct < ct1 ? ct : ct1 - 1;
        image_dma<C::PU>(imgU + cc * C::PU, img + islot * C::PU, w, lane);
        const u4v *src = reinterpret_cast<const u4v *>(S + (cc * ngt + gtc) * 1024);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + q * 64 + lane),
                                             (__attribute__((address_space(3))) void *)(sbuf + sslot * 2048 + q * 64), 16, 0, 0);
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
    for (int64_t ct = ct0; ct < ct1; ++ct) {
        const u4v *im = img + islot * C::PU;
        // register v = 4 q' + r of lane (g, h) holds s[cell acc_row(v, h)][gene g]: A operand of C += S^T FU
        f4v sc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) sc[q] = __builtin_bit_cast(f4v, sbuf[sslot * 2048 + q * 64 + lane]);
        u4v a2[2][3];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = sc[(8 * q + e) >> 2][(8 * q + e) & 3];
            split8(x, a2[q]);
        }
        // the tile of s is in registers: its ring slot (and the image slot the work-group left at the last barrier) take
        // the copies of the tile two ahead
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
            // 16 blocks of 4 genes x 4 factors, one cell per instruction: A[b][i] = s of gene 4 (b % 8) + i (this lane's
            // value), B[b][j] = FU[cell][KM + j]
            const f4v *t2p = reinterpret_cast<const f4v *>(im + C::P2 + 32) + (h * 4 + (lane & 3)) * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f4v t2 = t2p[q];
#pragma unroll
                for (int e = 0; e < 4; ++e) cta = __builtin_amdgcn_mfma_f32_4x4x1f32(sc[q][e], t2[e], cta, 0, 0, 0);
            }
        }
        if (++since == COL_FLUSH) {      // 256 cells: leave the matrix core
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the last copies target LDS: they must land before the group ends)
    if (!active) return;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int64_t gene = (int64_t)gt * 32 + acc_row(v, h);
            const int k = nt * 32 + c;
            if (k < C::KM) atomicAdd(Cout + gene * Kp + k, cs[nt][v] + dv[nt][v]);
        }
    if (TAIL) {
        // lane 4 b + j holds, in register e, the tail sum of gene 4 (b % 8) + e and factor KM + j over its half's cells
        cta.x += __shfl_xor(cta.x, 32, 64); cta.y += __shfl_xor(cta.y, 32, 64);
        cta.z += __shfl_xor(cta.z, 32, 64); cta.w += __shfl_xor(cta.w, 32, 64);
        if (h == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                atomicAdd(Cout + ((int64_t)gt * 32 + 4 * (lane >> 2) + e) * Kp + C::KM + (lane & 3), cta[e]);
        }
    }
}

// ---- packing ---------------------------------------------------------------------------------------------------------
// One wave per (cell tile, gene tile): lane (c, h) gathers its 16 counts x[cell c][gene acc_row(v, h)].
template <typename XT>
__global__ __launch_bounds__(64) void k_dn_pack(uint16_t *__restrict__ Xd, const XT *__restrict__ X, int64_t rows,
                                                int64_t ldx, int64_t ct_first, int ngt) {
    const int lane = threadIdx.x, c = lane & 31, h = lane >> 5;
    const int64_t ctl = blockIdx.y, gt = blockIdx.x;
    const int64_t r = ctl * 32 + c;
    uint32_t o[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        uint32_t pr[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int v = 2 * d + e;
            const int64_t col = gt * 32 + acc_row(v, h);
            const float x = (r < rows) ? (float)X[r * ldx + col] : 0.f;
            pr[e] = (uint32_t)x & 0xFFFFu;
        }
        o[d] = pr[0] | (pr[1] << 16);
    }
    u4v *dst = reinterpret_cast<u4v *>(Xd + ((ct_first + ctl) * ngt + gt) * 1024);
    dst[lane] = u4v{o[0], o[1], o[2], o[3]};
    dst[64 + lane] = u4v{o[4], o[5], o[6], o[7]};
}

// ---- slow path -------------------------------------------------------------------------------------------------------
// grid = cell tiles; the work-group gathers the flagged gene tiles of its cell tile, then evaluates the entries that
// carry the NaN sentinel in exact reference arithmetic (gap.py:72-80), adding them to the outputs with float atomics
// like passes.hip k_fixup; the sentinel becomes 0.
__global__ __launch_bounds__(256) void k_dn_fixup(const uint16_t *__restrict__ Xd, float *__restrict__ S,
                                                  const int32_t *__restrict__ flag, const float *__restrict__ logU,
                                                  const float *__restrict__ logV, const int32_t *__restrict__ row_perm,
                                                  const int32_t *__restrict__ col_perm, float *__restrict__ Zi,
                                                  float *__restrict__ Zj, int64_t n, int ngt, int K,
                                                  const float *__restrict__ dq, const float *__restrict__ S_tilde,
                                                  const float *__restrict__ S_hat, float *__restrict__ Zlog, int zj_packed) {
    __shared__ int nhit;
    __shared__ int hits[256];
    const int64_t ct = blockIdx.x;
    if (ct * 32 >= n) return;                           // (tiles of padding cells are never read back)
    for (int g0 = 0; g0 < ngt; g0 += 256) {
        if (threadIdx.x == 0) nhit = 0;
        __syncthreads();
        const int gq = g0 + threadIdx.x;
        if (gq < ngt && flag[ct * ngt + gq] != 0) hits[atomicAdd(&nhit, 1)] = gq;
        __syncthreads();
        const int nh = nhit;
        for (int e = threadIdx.x; e < nh * 1024; e += 256) {
            const int gt = hits[e >> 10];
            const int f = e & 1023;                     // index inside the S block: ((q * 64 + l) * 4 + r)
            float *sp = S + (ct * ngt + gt) * 1024 + f;
            const float s = *sp;
            if (s == s) continue;
            const int r = f & 3, l = (f >> 2) & 63, q = f >> 8;
            const int g = l & 31, hh = l >> 5;
            const int cell = 8 * q + 4 * hh + r;
            // the count: lane (cell, h) of k_dn_row, value v with acc_row(v, h) == g
            const int h = (g >> 2) & 1, v = (g >> 3) * 4 + (g & 3);
            const uint32_t xi = Xd[(ct * ngt + gt) * 1024 + ((v >> 3) * 64 + 32 * h + cell) * 8 + (v & 7)];
            *sp = 0.f;
            const int64_t ip = ct * 32 + cell, jp = (int64_t)gt * 32 + g;
            if (xi == 0u || ip >= n) continue;
            const int64_t i = row_perm ? (int64_t)row_perm[ip] : ip;
            const int64_t j = col_perm ? (int64_t)col_perm[jp] : jp;
            const float *lu = logU + i * K, *lv = logV + j * K;
            const float *st = S_tilde ? S_tilde + j * K : nullptr;
            const float *sh = S_hat ? S_hat + j * K : nullptr;
            const float x = (float)xi;
            float den = 0.f;
            for (int k = 0; k < K; ++k) {
                float ex = expf(lu[k] + lv[k]);
                if (st) ex *= st[k];                                                    // sparse_gap.py:88
                den += ex;
            }
            den = (den > 0.f) ? den : 1.0f;
            for (int k = 0; k < K; ++k) {
                const float ls = lu[k] + lv[k];
                float ex = expf(ls);
                if (st) ex *= st[k];
                const float expectation = (x * ex) / den;                               // gap.py:78
                const float vi = sh ? sh[k] * expectation : expectation;               // sparse_gap.py:95
                if (vi != 0.f) atomicAdd(&Zi[i * K + k], vi);
                const float vj = dq ? dq[i * K + k] * expectation : expectation;        // zigap.py:94 (D_hat[i, k])
                if (vj != 0.f) atomicAdd(&Zj[(zj_packed ? jp : j) * K + k], vj);
                if (Zlog) {
                    const float vl = expectation * ls;                                  // zigap.py:95 / sparse_gap.py:97
                    if (vl != 0.f) atomicAdd(&Zlog[j * K + k], vl);
                }
            }
        }
        __syncthreads();
    }
}

// D_hat[i, j] = value at every non-zero count of the dense block (zigap.py:77, 135): one work-group per (cell tile, gene tile)
__global__ __launch_bounds__(256) void k_dn_fix_nz(const uint16_t *__restrict__ Xd, float *__restrict__ D_hat, int64_t ld,
                                                   const int32_t *__restrict__ row_perm, const int32_t *__restrict__ col_perm,
                                                   float value, int64_t n, int ngt) {
    const int gt = blockIdx.x;
    const int64_t ct = blockIdx.y;
    const uint16_t *blk = Xd + (ct * ngt + gt) * 1024;
    for (int e = threadIdx.x; e < 1024; e += 256) {
        if (blk[e] == 0) continue;
        const int v = ((e >> 9) << 3) | (e & 7), l = (e >> 3) & 63;
        const int64_t ip = ct * 32 + (l & 31), jp = (int64_t)gt * 32 + acc_row(v, l >> 5);
        if (ip >= n) continue;
        const int64_t i = row_perm ? (int64_t)row_perm[ip] : ip;
        const int64_t j = col_perm ? (int64_t)col_perm[jp] : jp;
        D_hat[i * ld + j] = value;
    }
}

// ---- metrics over the dense block (oriana_count_stats + oriana_metric_nnz for the dense genes) -------------------------
// One work-group per (cell tile, gene tile); Lambda = U V^T in float64 at the non-zero counts.
//   colsum[gene] += sum_i x, colnnz[gene] += #{x != 0} (caller's gene order), out2 += {sum (x log x - x), sum x^2},
//   out4 += {sum Lambda, sum x log Lambda, sum Lambda^2, sum x Lambda}   (U, V == NULL: the constants only)
__global__ __launch_bounds__(256) void k_dn_metric(const uint16_t *__restrict__ Xd, const double *__restrict__ U,
                                                   const double *__restrict__ V, const int32_t *__restrict__ row_perm,
                                                   const int32_t *__restrict__ col_perm, double *__restrict__ colsum,
                                                   double *__restrict__ colnnz, double *__restrict__ out2,
                                                   double *__restrict__ out4, int64_t n, int ngt, int K) {
    extern __shared__ double fac[];                // [32 cells][K] then [32 genes][K]
    __shared__ double red[6][4];
    __shared__ float cs[32];
    __shared__ int cn[32];
    const int gt = blockIdx.x;
    const int64_t ct = blockIdx.y;
    const int tid = threadIdx.x;
    if (tid < 32) { cs[tid] = 0.f; cn[tid] = 0; }
    if (U) {
        for (int e = tid; e < 32 * K; e += 256) {
            const int r = e / K, k = e - r * K;
            const int64_t ip = ct * 32 + r, jp = (int64_t)gt * 32 + r;
            const int64_t i = (ip < n) ? (row_perm ? (int64_t)row_perm[ip] : ip) : -1;
            const int64_t j = col_perm ? (int64_t)col_perm[jp] : jp;
            fac[e] = (i >= 0) ? U[i * K + k] : 0.0;
            fac[32 * K + e] = V[j * K + k];
        }
    }
    __syncthreads();
    double a[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    const uint16_t *blk = Xd + (ct * ngt + gt) * 1024;
    for (int e = tid; e < 1024; e += 256) {
        // entry e of the block: ((v / 8) * 64 + l) * 8 + v % 8, lane l = (cell c, half h), gene acc_row(v, h)
        const int v = ((e >> 9) << 3) | (e & 7), l = (e >> 3) & 63;
        const int c = l & 31, h = l >> 5, g = acc_row(v, h);
        const uint32_t xi = blk[e];
        if (xi == 0u) continue;
        const double x = (double)xi;
        a[0] += x * log(x) - x;
        a[1] += x * x;
        atomicAdd(&cs[g], (float)xi);               // <= 32 counts below 2^16 per gene: exact in float32
        atomicAdd(&cn[g], 1);
        if (U) {
            double lam = 0.0;
            const double *u = fac + c * K, *vv = fac + 32 * K + g * K;
            for (int k = 0; k < K; ++k) lam += u[k] * vv[k];
            a[2] += lam; a[3] += x * log(lam); a[4] += lam * lam; a[5] += x * lam;
        }
    }
    for (int q = 0; q < 6; ++q) {
        double t = a[q];
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
        if ((tid & 63) == 0) red[q][tid >> 6] = t;
    }
    __syncthreads();
    if (tid < 6) {
        const double t = red[tid][0] + red[tid][1] + red[tid][2] + red[tid][3];
        if (tid < 2) { if (out2 && t != 0.0) atomicAdd(&out2[tid], t); }
        else if (out4 && t != 0.0) atomicAdd(&out4[tid - 2], t);
    }
    if (tid < 32 && cn[tid] != 0 && colsum) {
        const int64_t jp = (int64_t)gt * 32 + tid;
        const int64_t j = col_perm ? (int64_t)col_perm[jp] : jp;
        atomicAdd(&colsum[j], (double)cs[tid]);
        atomicAdd(&colnnz[j], (double)cn[tid]);
    }
}

template <int KC, int TAIL> constexpr int row_lds_bytes() { return 3 * Cfg<KC, TAIL>::PV * 16 + NW * 32 * TS * 4; }
template <int KC, int TAIL> constexpr int col_lds_bytes() { return 3 * Cfg<KC, TAIL>::PU * 16 + 2 * NW * 256 * 16; }

template <typename Fn>
static int set_lds(Fn fn, int bytes) {
    if (bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return -1000 - (int)e;
    }
    return 0;
}

}  // namespace dn
}  // namespace oriana

using namespace oriana;
using namespace oriana::dn;

// compiled (KC, TAIL) pairs: Kp = 16 KC + 4 TAIL, i.e. every K <= 100
#define ORIANA_DN_FOR_CFG(KC_, TL_, CALL)                                     \
    do {                                                                      \
        if (KC_ == 6 && TL_ == 1) { CALL(6, 1); }                             \
        else if (KC_ == 6 && TL_ == 0) { CALL(6, 0); }                        \
        else if (KC_ == 5 && TL_ == 1) { CALL(5, 1); }                        \
        else if (KC_ == 5 && TL_ == 0) { CALL(5, 0); }                        \
        else if (KC_ == 4 && TL_ == 1) { CALL(4, 1); }                        \
        else if (KC_ == 4 && TL_ == 0) { CALL(4, 0); }                        \
        else if (KC_ == 3 && TL_ == 1) { CALL(3, 1); }                        \
        else if (KC_ == 3 && TL_ == 0) { CALL(3, 0); }                        \
        else if (KC_ == 2 && TL_ == 1) { CALL(2, 1); }                        \
        else if (KC_ == 2 && TL_ == 0) { CALL(2, 0); }                        \
        else if (KC_ == 1 && TL_ == 1) { CALL(1, 1); }                        \
        else if (KC_ == 1 && TL_ == 0) { CALL(1, 0); }                        \
        else return ORIANA_EKRANGE;                                           \
    } while (0)

static bool dn_cfg(int64_t K, int *kc, int *tl, int *kp) {
    const int64_t Kp = oriana_kpad(K);
    if (Kp == 0 || Kp > 100 || (Kp % 16 != 0 && Kp % 16 != 4)) return false;   // (Kp = 112: three image buffers exceed LDS)
    *kc = (int)(Kp / 16); *tl = (Kp % 16 == 4) ? 1 : 0; *kp = (int)Kp;
    return *kc >= 1 && *kc <= 6;
}

extern "C" int oriana_dense_supported(int64_t K) {
    int kc, tl, kp;
    return dn_cfg(K, &kc, &tl, &kp) ? 1 : 0;
}

// 16-byte pieces of one tile's operand image: side 0 = gene side (both images), 1 = cell side
extern "C" int64_t oriana_dense_image_pieces(int64_t K, int side) {
    int kc, tl, kp;
    if (!dn_cfg(K, &kc, &tl, &kp)) return 0;
    int64_t out = 0;
#define ORIANA_DN_CALL(KC, TL) out = side ? Cfg<KC, TL>::PU : Cfg<KC, TL>::PV
    ORIANA_DN_FOR_CFG(kc, tl, ORIANA_DN_CALL);
#undef ORIANA_DN_CALL
    return out;
}

static bool dense_ok(const oriana_dense *d) {
    return d && d->n >= 0 && d->gd >= 0 && d->gd % 32 == 0 && d->nct == (d->n + 255) / 256 * 8 && (d->gd == 0 || d->nct == 0 || d->x);
}

extern "C" int oriana_dense_pack(const void *X, int xdtype, int64_t rows, int64_t gd, int64_t ldx, int64_t ct_first,
                                 uint16_t *xd, void *stream) {
    if (rows < 0 || gd < 0 || gd % 32 != 0 || ct_first < 0) return ORIANA_EINVAL;
    if (rows == 0 || gd == 0) return 0;
    if (!X || !xd) return ORIANA_EINVAL;
    const int ngt = (int)(gd / 32);
    const dim3 grid((unsigned)ngt, (unsigned)((rows + 31) / 32));
    hipStream_t s = (hipStream_t)stream;
    switch (xdtype) {
        case 0: hipLaunchKernelGGL(k_dn_pack<float>, grid, dim3(64), 0, s, xd, (const float *)X, rows, ldx, ct_first, ngt); break;
        case 1: hipLaunchKernelGGL(k_dn_pack<int64_t>, grid, dim3(64), 0, s, xd, (const int64_t *)X, rows, ldx, ct_first, ngt); break;
        case 2: hipLaunchKernelGGL(k_dn_pack<int32_t>, grid, dim3(64), 0, s, xd, (const int32_t *)X, rows, ldx, ct_first, ngt); break;
        case 3: hipLaunchKernelGGL(k_dn_pack<double>, grid, dim3(64), 0, s, xd, (const double *)X, rows, ldx, ct_first, ngt); break;
        default: return ORIANA_EINVAL;
    }
    ORIANA_LAUNCH_CHECK();
    return 0;
}

extern "C" int oriana_dense_images2(void *img, const float *F, const float *F2, int64_t rows, int64_t K, int side,
                                    void *stream) {
    int kc, tl, kp;
    if (rows < 0 || K <= 0) return ORIANA_EINVAL;
    if (!dn_cfg(K, &kc, &tl, &kp)) return ORIANA_EKRANGE;
    if (rows == 0) return 0;
    if (!img || !F) return ORIANA_EINVAL;
    if (!F2) F2 = F;
    const unsigned tiles = (unsigned)((rows + 31) / 32);
    hipStream_t s = (hipStream_t)stream;
#define ORIANA_DN_CALL(KC, TL)                                                                                              \
    do {                                                                                                                    \
        if (side) hipLaunchKernelGGL((k_dn_images<KC, TL, false>), dim3(tiles), dim3(512), 0, s, (u4v *)img, F, F2, rows, kp); \
        else hipLaunchKernelGGL((k_dn_images<KC, TL, true>), dim3(tiles), dim3(512), 0, s, (u4v *)img, F, F2, rows, kp);     \
    } while (0)
    ORIANA_DN_FOR_CFG(kc, tl, ORIANA_DN_CALL);
#undef ORIANA_DN_CALL
    ORIANA_LAUNCH_CHECK();
    return 0;
}

extern "C" int oriana_dense_images(void *img, const float *F, int64_t rows, int64_t K, int side, void *stream) {
    return oriana_dense_images2(img, F, F, rows, K, side, stream);
}

extern "C" int oriana_dense_row_pass_tail(const oriana_dense *d, const float *FU, const void *imgV, float *R, float *S,
                                          int32_t *flag, int64_t K, int64_t gene_splits, int64_t tail_nfull, int64_t tail_parts,
                                          const float *den_min, void *stream) {
    int kc, tl, kp;
    if (!dense_ok(d) || K <= 0 || gene_splits < 1) return ORIANA_EINVAL;
    if (!dn_cfg(K, &kc, &tl, &kp)) return ORIANA_EKRANGE;
    if (d->gd == 0 || d->n == 0) return 0;
    if (!FU || !imgV || !R || !S || !flag) return ORIANA_EINVAL;
    const int ngt = (int)(d->gd / 32);
    const int64_t nblk = d->nct / NW;
    int64_t splits = gene_splits < ngt ? gene_splits : ngt;
    if ((ngt + splits - 1) / splits > 2048) splits = (ngt + 2047) / 2048;      // (the kernel keeps one flag bit per tile in 64 x 32 bits)
    const int per = (int)((ngt + splits - 1) / splits);
    splits = (ngt + per - 1) / per;
    dim3 grid((unsigned)nblk, (unsigned)splits);
    if (tail_parts > 1) {            // the blocks from tail_nfull on in tail_parts even gene ranges (slabs of R), no other split
        if (splits != 1 || tail_nfull < 0 || tail_nfull > nblk || tail_parts > ngt || tail_parts > 65535 || ngt > 2048) return ORIANA_EINVAL;
        const int64_t items = tail_nfull + (nblk - tail_nfull) * tail_parts;
        if (items > 0x7fffffffLL) return ORIANA_EINVAL;
        grid = dim3((unsigned)items, 1u);
    } else {
        tail_nfull = nblk; tail_parts = 1;
    }
    hipStream_t s = (hipStream_t)stream;
#define ORIANA_DN_CALL(KC, TL)                                                                                              \
    do {                                                                                                                    \
        constexpr int lb = row_lds_bytes<KC, TL>();                                                                         \
        const int rc = set_lds(k_dn_row<KC, TL>, lb);                                                                       \
        if (rc) return rc;                                                                                                  \
        hipLaunchKernelGGL((k_dn_row<KC, TL>), grid, dim3(512), lb, s, d->x, S, FU, (const u4v *)imgV, R, flag, d->n, ngt,  \
                           kp, per, splits > 1 ? 1 : 0, (int)tail_nfull, (int)tail_parts, den_min);                         \
    } while (0)
    ORIANA_DN_FOR_CFG(kc, tl, ORIANA_DN_CALL);
#undef ORIANA_DN_CALL
    ORIANA_LAUNCH_CHECK();
    return 0;
}

extern "C" int oriana_dense_row_pass(const oriana_dense *d, const float *FU, const void *imgV, float *R, float *S,
                                     int32_t *flag, int64_t K, int64_t gene_splits, void *stream) {
    return oriana_dense_row_pass_tail(d, FU, imgV, R, S, flag, K, gene_splits, 0, 1, nullptr, stream);
}

extern "C" int oriana_dense_col_pass(const oriana_dense *d, const void *imgU, const float *S, float *C, int64_t K,
                                     int64_t cell_splits, void *stream) {
    int kc, tl, kp;
    if (!dense_ok(d) || K <= 0 || cell_splits < 1) return ORIANA_EINVAL;
    if (!dn_cfg(K, &kc, &tl, &kp)) return ORIANA_EKRANGE;
    if (d->gd == 0 || d->n == 0) return 0;
    if (!imgU || !S || !C) return ORIANA_EINVAL;
    const int ngt = (int)(d->gd / 32);
    const int64_t nct = (d->n + 31) / 32;             // cell tiles that hold rows
    int64_t splits = cell_splits < nct ? cell_splits : nct;
    const int64_t per = (nct + splits - 1) / splits;
    splits = (nct + per - 1) / per;
    const int64_t groups = (ngt + NW - 1) / NW;
    if (splits * groups > 0x7fffffffLL) return ORIANA_EINVAL;
    hipStream_t s = (hipStream_t)stream;
#define ORIANA_DN_CALL(KC, TL)                                                                                              \
    do {                                                                                                                    \
        constexpr int lb = col_lds_bytes<KC, TL>();                                                                         \
        const int rc = set_lds(k_dn_col<KC, TL>, lb);                                                                       \
        if (rc) return rc;                                                                                                  \
        hipLaunchKernelGGL((k_dn_col<KC, TL>), dim3((unsigned)(splits * groups)), dim3(512), lb, s, S, (const u4v *)imgU,   \
                           C, nct, ngt, kp, per, (int)splits);                                                              \
    } while (0)
    ORIANA_DN_FOR_CFG(kc, tl, ORIANA_DN_CALL);
#undef ORIANA_DN_CALL
    ORIANA_LAUNCH_CHECK();
    return 0;
}

extern "C" int oriana_dense_fixup_variant(const oriana_dense *d, const int32_t *flag, float *S, const float *logU,
                                          const float *logV, const int32_t *row_perm, const int32_t *col_perm, float *Zi,
                                          float *Zj, float *Zlog, const float *dq, const float *S_tilde, const float *S_hat,
                                          int64_t K, int zj_packed, void *stream) {
    if (!dense_ok(d) || K <= 0) return ORIANA_EINVAL;
    if (d->gd == 0 || d->n == 0) return 0;
    if (!flag || !S || !logU || !logV || !Zi || !Zj || ((S_tilde == nullptr) != (S_hat == nullptr))) return ORIANA_EINVAL;
    const int ngt = (int)(d->gd / 32);
    hipLaunchKernelGGL(k_dn_fixup, dim3((unsigned)d->nct), dim3(256), 0, (hipStream_t)stream, d->x, S,
                       flag, logU, logV, row_perm, col_perm, Zi, Zj, d->n, ngt, (int)K, dq, S_tilde, S_hat, Zlog, zj_packed);
    ORIANA_LAUNCH_CHECK();
    return 0;
}

extern "C" int oriana_dense_fixup_weighted(const oriana_dense *d, const int32_t *flag, float *S, const float *logU,
                                           const float *logV, const int32_t *row_perm, const int32_t *col_perm, float *Zi,
                                           float *Zj, const float *dq, int64_t K, void *stream) {
    return oriana_dense_fixup_variant(d, flag, S, logU, logV, row_perm, col_perm, Zi, Zj, nullptr, dq, nullptr, nullptr, K, 0, stream);
}

extern "C" int oriana_dense_fixup(const oriana_dense *d, const int32_t *flag, float *S, const float *logU,
                                  const float *logV, const int32_t *row_perm, const int32_t *col_perm, float *Zi,
                                  float *Zj, int64_t K, void *stream) {
    return oriana_dense_fixup_weighted(d, flag, S, logU, logV, row_perm, col_perm, Zi, Zj, nullptr, K, stream);
}

extern "C" int oriana_dense_fix_nz(const oriana_dense *d, float *D_hat, int64_t ld, const int32_t *row_perm,
                                   const int32_t *col_perm, double value, void *stream) {
    if (!dense_ok(d) || ld < d->gd) return ORIANA_EINVAL;
    if (d->gd == 0 || d->n == 0) return 0;
    if (!D_hat) return ORIANA_EINVAL;
    const int ngt = (int)(d->gd / 32);
    hipLaunchKernelGGL(k_dn_fix_nz, dim3((unsigned)ngt, (unsigned)((d->n + 31) / 32)), dim3(256), 0, (hipStream_t)stream, d->x,
                       D_hat, ld, row_perm, col_perm, (float)value, d->n, ngt);
    ORIANA_LAUNCH_CHECK();
    return 0;
}

extern "C" int oriana_dense_metric(const oriana_dense *d, const double *U, const double *V, const int32_t *row_perm,
                                   const int32_t *col_perm, double *colsum, double *colnnz, double *out2, double *out4,
                                   int64_t K, void *stream) {
    if (!dense_ok(d) || K <= 0 || K > 1024) return ORIANA_EINVAL;
    if (d->gd == 0 || d->n == 0) return 0;
    if ((U == nullptr) != (V == nullptr) || (U && !out4) || (colsum && !colnnz)) return ORIANA_EINVAL;
    const int ngt = (int)(d->gd / 32);
    const size_t lb = U ? (size_t)64 * K * sizeof(double) : 0;
    if (lb > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)k_dn_metric, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb);
        if (e != hipSuccess) return -1000 - (int)e;
    }
    hipLaunchKernelGGL(k_dn_metric, dim3((unsigned)ngt, (unsigned)((d->n + 31) / 32)), dim3(256), lb, (hipStream_t)stream, d->x, U,
                       V, row_perm, col_perm, colsum, colnnz, out2, out4, d->n, ngt, (int)K);
    ORIANA_LAUNCH_CHECK();
    return 0;
}

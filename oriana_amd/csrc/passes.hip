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

// (the ablation switches of rounds 1-4 -- no barriers, no scattered s stores, masked padding slots, rotation variants, staging
//  once, staggered waves -- are archived as tools/experiments/passes_ablation_switches_r4.diff)

#include "passes_prep.h"
#include "passes_generic.h"
#include "passes_k100.h"
#include "passes_k64.h"
#include "passes_narrow.h"

namespace oriana {

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

// Which family serves which padded width (one answer per configuration; DESIGN.md section 0 has the table per model):
//   row pass     Kp <= 32, plain variant ......... narrow  (one lane per row)
//                Kp = 36, 48, 52, 64 ............. k64     (two lanes per row; every variant)
//                Kp = 96, 100 ................... k100    (two lanes per row; no second image: two do not fit in LDS)
//                everything else ................ generic (G lanes per row): Kp <= 32 with weights / row-side s / a second
//                                                 image, Kp = 68, 80, 84, 112 .. 256
//   column pass  Kp <= 20 ....................... narrow  (one lane per gene)
//                Kp = 36 .. 64 .................. k64     (two lanes per gene, two column tiles per image)
//                other Kp <= 112 (G = 4) ........ k_col_pass2 (four lanes per gene, two column tiles per image)
//                Kp >= 128 (G = 8, 16) .......... generic
// (rounds 1-4 selected older generations with ORIANA_PASS_IMPL=r1|r2|r3 for A/B runs: gone; the git history has them)
static constexpr bool use_narrow(int G, int T4, int TAIL) { return G == 4 && 4 * T4 + TAIL <= 8; }       // row pass: Kp <= 32
// (column pass: Kp <= 20 -- at Kp = 32 the two-tile kernel measured 26.0 us against 28.3 at 10,000 x 2,000)
static constexpr bool use_narrow_col(int G, int T4, int TAIL) { return G == 4 && 4 * T4 + TAIL <= 5; }
static constexpr bool use_k100(int G, int T4) { return G == 4 && T4 == 6; }
static constexpr bool k64_kernels() { return true; }
// ORIANA_DEN_THRESHOLD=fixed: the row kernels keep the constant DEN_MIN (round 3's rule; A/B runs)
static bool den_threshold_dynamic() {
    static const bool fixed = [] { const char *e = getenv("ORIANA_DEN_THRESHOLD"); return e && !strcmp(e, "fixed"); }();
    return !fixed;
}
static constexpr bool k64_cfg(int G, int T4, int TAIL) { return G == 4 && 4 * T4 + TAIL >= 9 && 4 * T4 + TAIL <= 16; }
static constexpr bool use_col2(int G, int T4, int TAIL) { return G == 4 && !use_narrow_col(G, T4, TAIL); }   // column pass, two tiles per image

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
    int64_t nfull = 0, parts = 1;
    if (groups < cus) {
        // short matrices: two work-groups per CU at most, evenly sized ranges (measured at 10,000 x 2,000, K = 20:
        // 77 / 42 / 25 / 24 us for 1 / 2 / 4 / 8 groups per row block; 8 is the better sweep)
        parts = 2 * cus / groups;
    } else if (two_lane) {
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

// [r6] ANALYSIS entry (tools/parity_report.py): the column pass with float64 accumulators in a fixed
// order -- one thread per gene and group of 8 factors walks the gene's slots through every row block -- and ONE rounding to
// float32 at the end: C += f32(sum_i s_ij G_i).  What a compensated (Kahan / two-float) accumulation of the float32 kernels
// could reach at most; no kernel of a sweep calls it.
__global__ __launch_bounds__(256) void k_col_pass_f64acc(oriana_counts cm, const float *__restrict__ s_cs,
                                                         const float *__restrict__ Gm, float *__restrict__ C, int K, int Kp) {
    const int c = threadIdx.x, sl = c >> 4;
    const int64_t cb = blockIdx.x, col = cb * TILE + c;
    const int k0 = blockIdx.y * 8;
    if (col >= cm.m) return;
    double acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.0;
    for (int64_t rb = 0; rb < cm.nrb; ++rb) {
        const int64_t t = rb * cm.ncb + cb;
        const uint32_t s0 = cm.cslice[t * 17 + sl], s1 = cm.cslice[t * 17 + sl + 1];
        const int ni = (int)((s1 - s0) >> 6);
        const int64_t base = cm.coff[t] + s0 + (c & 15) * 4;
        for (int it = 0; it < ni; ++it)
            for (int r = 0; r < 4; ++r) {
                const float sv = s_cs[base + (int64_t)it * 64 + r];
                if (sv == 0.f) continue;
                const int64_t row = rb * TILE + cm.ridx[base + (int64_t)it * 64 + r];
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (k0 + e < K) acc[e] += (double)sv * (double)Gm[row * Kp + k0 + e];
            }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e)
        if (k0 + e < K) C[col * Kp + k0 + e] += (float)acc[e];
}

extern "C" int oriana_col_pass_f64acc(const oriana_counts *cm, const float *s_cs, const float *Gm, float *C, int64_t K,
                                      void *stream) {
    if (!counts_ok(cm) || K <= 0) return ORIANA_EINVAL;
    const int64_t Kp = oriana_kpad(K);
    if (Kp == 0) return ORIANA_EKRANGE;
    if (cm->n == 0 || cm->m == 0) return 0;
    if (!Gm || !C || !s_cs) return ORIANA_EINVAL;
    hipLaunchKernelGGL(k_col_pass_f64acc, dim3((unsigned)cm->ncb, (unsigned)((K + 7) / 8)), dim3(256), 0, (hipStream_t)stream, *cm,
                       s_cs, Gm, C, (int)K, (int)Kp);
    ORIANA_LAUNCH_CHECK();
    return 0;
}

// two images, one column tile per work item (work list of width 1)
template <int G, int T4, int TAIL>
static int launch_col_pass_dual(const oriana_counts *cm, const float *s_cs, const float *G1, const float *G2, float *C1,
                                float *C2, const int32_t *work, int64_t nwork, hipStream_t s) {
    constexpr int T4c = (G == 4) ? T4 : 1, TLc = (G == 4) ? TAIL : 0;
    using Im = ColImage<T4c, TLc>;
    // (33 <= Kp <= 64: the two-lane dual kernel k64::k_col_pass_k64<KP4, true> measured 11.3 ms against 10.9 ms for this four-lane
    //  one at configs[4] -- with two images per step the walk is bound by the LDS return port either way; it is not dispatched)
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
    const int per = (int)(nt / 2048 < 1 ? 1 : (nt / 2048 > 64 ? 64 : nt / 2048));       // tiles per work-group (k_fixup)
    hipLaunchKernelGGL(k_fixup, dim3((unsigned)((nt + per - 1) / per)), dim3(256), 0, (hipStream_t)stream, *cm, tile_flag, s_cs,
                       sw_cs, s_rs, logU, logV, S_tilde, S_hat, w_nz, dq, Zi, Zj, Zlog, (int)K, quirk, nt, per);
    ORIANA_LAUNCH_CHECK();
    return 0;
}

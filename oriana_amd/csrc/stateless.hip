// stateless.hip -- drop-ins with the reference's exact kernel signatures (outputs first, inputs
// after, dense float32 matrices, callee zero-fills, gap.py:67-80 and twins).  X is repacked on
// every call, as the reference re-casts X on every call (gap.py:94); the model classes keep the
// packed layout resident instead and call the passes directly.
#include "common.h"

namespace oriana {

// exclusive scan of int32 counts into int64 offsets, one workgroup (ntiles is at most a few 1e5)
__global__ __launch_bounds__(1024) void k_scan_tiles(int64_t *__restrict__ off, const int32_t *__restrict__ cnt, int64_t nt) {
    __shared__ int64_t wsum[16];
    __shared__ int64_t carry_s;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int64_t base = 0; base < nt; base += 1024) {
        const int64_t i = base + tid;
        const int64_t v = (i < nt) ? (int64_t)cnt[i] : 0;
        int64_t inc = v;
        #pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int64_t t = __shfl_up(inc, o, 64);
            if (lane >= o) inc += t;
        }
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        int64_t pre = carry_s;
        for (int j = 0; j < w; ++j) pre += wsum[j];
        if (i < nt) off[i] = pre + inc - v;
        __syncthreads();
        if (tid == 1023) carry_s = pre + inc;
        __syncthreads();
    }
    if (tid == 0) off[nt] = carry_s;
}

struct WsLayout {
    int64_t nt, Kp;
    size_t tile_nnz, tile_rslots, tile_cslots, roff, coff, rslice, cslice, tile_flag, totals, rowrec, ridx, s_cs,
        FU, FV, R, C, w_nz, sw_cs, s_rs, F2, G2, C2, dq, prep, total;
};

static inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// rslot_cap / cslot_cap: capacities of the row-side / column-side slot arrays
static WsLayout ws_layout(int64_t n, int64_t m, int64_t K, int64_t rslot_cap, int64_t cslot_cap) {
    WsLayout L;
    const int64_t nrb = (n + TILE - 1) / TILE, ncb = (m + TILE - 1) / TILE;
    L.nt = nrb * ncb;
    L.Kp = oriana_kpad(K);
    size_t o = 0;
    const int64_t nt1 = L.nt > 0 ? L.nt : 1;
    const int64_t rs1 = rslot_cap > 0 ? rslot_cap : 1, cs1 = cslot_cap > 0 ? cslot_cap : 1;
    const int64_t n1 = n > 0 ? n : 1, m1 = m > 0 ? m : 1;
    L.tile_nnz = o;    o = align256(o + sizeof(int32_t) * nt1);
    L.tile_rslots = o; o = align256(o + sizeof(int32_t) * nt1);
    L.tile_cslots = o; o = align256(o + sizeof(int32_t) * nt1);
    L.roff = o;        o = align256(o + sizeof(int64_t) * (nt1 + 1));
    L.coff = o;        o = align256(o + sizeof(int64_t) * (nt1 + 1));
    L.rslice = o;      o = align256(o + sizeof(uint32_t) * nt1 * 17);
    L.cslice = o;      o = align256(o + sizeof(uint32_t) * nt1 * 17);
    L.tile_flag = o;   o = align256(o + sizeof(int32_t) * nt1);
    L.totals = o;      o = align256(o + sizeof(int64_t) * 4);
    L.rowrec = o;      o = align256(o + sizeof(oriana_rowrec) * rs1);
    L.ridx = o;        o = align256(o + cs1);
    L.s_cs = o;        o = align256(o + sizeof(float) * cs1);
    L.FU = o;          o = align256(o + sizeof(float) * n1 * L.Kp);
    L.FV = o;          o = align256(o + sizeof(float) * m1 * L.Kp);
    L.R = o;           o = align256(o + sizeof(float) * n1 * L.Kp);
    L.C = o;           o = align256(o + sizeof(float) * m1 * L.Kp);
    // extras of the ZI / sparse loop nests (always laid out: one workspace size serves the four entries)
    L.w_nz = o;        o = align256(o + sizeof(float) * rs1);
    L.sw_cs = o;       o = align256(o + sizeof(float) * cs1);
    L.s_rs = o;        o = align256(o + sizeof(float) * rs1);
    L.F2 = o;          o = align256(o + sizeof(float) * m1 * L.Kp);
    L.G2 = o;          o = align256(o + sizeof(float) * n1 * L.Kp);
    L.C2 = o;          o = align256(o + sizeof(float) * m1 * L.Kp);
    L.dq = o;          o = align256(o + sizeof(float) * n1 * (K > 0 ? K : 1));
    L.prep = o;        o = align256(o + (size_t)oriana_prep_scratch_bytes());
    L.total = o;
    return L;
}

// Slot capacities that are always sufficient for `nnz_bound` non-zeros: a slice iteration holds 64
// slots and is opened by at least one record, and each tile carries 64 dummy slots.
static inline int64_t slot_bound(int64_t nnz_bound) { return 64 * nnz_bound; }

}  // namespace oriana

using namespace oriana;

extern "C" int64_t oriana_zq_workspace_bytes(int64_t n, int64_t m, int64_t K, int64_t nnz_bound) {
    if (n < 0 || m < 0 || nnz_bound < 0 || oriana_kpad(K) == 0) return 0;
    const int64_t nt = ((n + TILE - 1) / TILE) * ((m + TILE - 1) / TILE);
    // worst case: every record alone in its slice iteration; capped by the dense tile capacity
    int64_t cap = slot_bound(nnz_bound);
    const int64_t dense = nt * 65536;
    if (cap > dense) cap = dense;
    return (int64_t)ws_layout(n, m, K, cap, cap + 64 * nt).total;
}

// The four loop nests on dense inputs.  Zlog / S_tilde / S_hat / D_hat may be NULL (absent in that
// variant); quirk = zigap.py:94 (per-gene sums weighted by D_hat[i, k]).  Mirrors engine.zq.
static int zq_dense(float *Zi, float *Zj, float *Zlog, const float *log_U_hat, const float *log_V_hat,
                    const float *S_tilde, const float *S_hat, const float *D_hat, int quirk, const float *X,
                    int64_t n, int64_t m, int64_t K, void *ws, int64_t ws_bytes, void *stream) {
    if (n < 0 || m < 0 || K <= 0) return ORIANA_EINVAL;
    if (oriana_kpad(K) == 0) return ORIANA_EKRANGE;
    if ((S_tilde == nullptr) != (S_hat == nullptr)) return ORIANA_EINVAL;
    if (quirk && (!D_hat || K > m)) return ORIANA_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (n > 0 && (!Zi || !log_U_hat)) return ORIANA_EINVAL;
    if (m > 0 && (!Zj || !log_V_hat)) return ORIANA_EINVAL;
    // callee zero-fills the outputs (gap.py:69-70, zigap.py:82-84)
    if (n > 0) ORIANA_HIP_CHECK(hipMemsetAsync(Zi, 0, sizeof(float) * n * K, s));
    if (m > 0) ORIANA_HIP_CHECK(hipMemsetAsync(Zj, 0, sizeof(float) * m * K, s));
    if (m > 0 && Zlog) ORIANA_HIP_CHECK(hipMemsetAsync(Zlog, 0, sizeof(float) * m * K, s));
    if (n == 0 || m == 0) return 0;
    if (!X || !ws || ((uintptr_t)ws & 255)) return ORIANA_EINVAL;
    const bool sparse = S_hat != nullptr, weighted = D_hat != nullptr;
    const int64_t nrb = (n + TILE - 1) / TILE, ncb = (m + TILE - 1) / TILE, nt = nrb * ncb;
    // fixed-size part first: counts, offsets, slice tables
    const WsLayout L0 = ws_layout(n, m, K, 0, 0);
    if ((size_t)ws_bytes < L0.total) return ORIANA_EINVAL;
    char *b = (char *)ws;
    int32_t *tile_nnz = (int32_t *)(b + L0.tile_nnz), *tile_rslots = (int32_t *)(b + L0.tile_rslots);
    int32_t *tile_cslots = (int32_t *)(b + L0.tile_cslots), *tile_flag = (int32_t *)(b + L0.tile_flag);
    int64_t *roff = (int64_t *)(b + L0.roff), *coff = (int64_t *)(b + L0.coff);
    uint32_t *rslice = (uint32_t *)(b + L0.rslice), *cslice = (uint32_t *)(b + L0.cslice);

    int rc = oriana_pack_count(X, 0, n, m, m, 0, ncb, tile_nnz, tile_rslots, tile_cslots, rslice, cslice, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, s, roff, tile_rslots, nt);
    hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, s, coff, tile_cslots, nt);
    ORIANA_LAUNCH_CHECK();
    // the one host synchronisation of this entry point: the slot totals size the record arrays
    int64_t tot[2] = {0, 0};
    ORIANA_HIP_CHECK(hipMemcpyAsync(&tot[0], roff + nt, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    ORIANA_HIP_CHECK(hipMemcpyAsync(&tot[1], coff + nt, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    ORIANA_HIP_CHECK(hipStreamSynchronize(s));
    const WsLayout L = ws_layout(n, m, K, tot[0], tot[1]);
    if (L.total > (size_t)ws_bytes) return ORIANA_EINVAL;
    const int64_t rs1 = tot[0] > 0 ? tot[0] : 1, cs1 = tot[1] > 0 ? tot[1] : 1;
    oriana_rowrec *rowrec = (oriana_rowrec *)(b + L.rowrec);
    uint8_t *ridx = (uint8_t *)(b + L.ridx);
    float *s_cs = (float *)(b + L.s_cs), *FU = (float *)(b + L.FU), *FV = (float *)(b + L.FV);
    float *R = (float *)(b + L.R), *C = (float *)(b + L.C);
    float *w_nz = weighted ? (float *)(b + L.w_nz) : nullptr;
    float *sw_cs = weighted ? (float *)(b + L.sw_cs) : nullptr;
    float *s_rs = sparse ? (float *)(b + L.s_rs) : nullptr;
    float *F2 = (float *)(b + L.F2), *G2 = (float *)(b + L.G2), *C2 = (float *)(b + L.C2), *dq = nullptr;
    // padding slots: x == 0 records, row index 0, s == 0
    ORIANA_HIP_CHECK(hipMemsetAsync(rowrec, 0, sizeof(oriana_rowrec) * rs1, s));
    ORIANA_HIP_CHECK(hipMemsetAsync(ridx, 0, cs1, s));
    ORIANA_HIP_CHECK(hipMemsetAsync(s_cs, 0, sizeof(float) * cs1, s));
    if (weighted) {
        ORIANA_HIP_CHECK(hipMemsetAsync(w_nz, 0, sizeof(float) * rs1, s));
        ORIANA_HIP_CHECK(hipMemsetAsync(sw_cs, 0, sizeof(float) * cs1, s));
    }
    if (sparse) ORIANA_HIP_CHECK(hipMemsetAsync(s_rs, 0, sizeof(float) * rs1, s));
    rc = oriana_pack_fill(X, 0, n, m, m, 0, ncb, roff, coff, rslice, cslice, rowrec, ridx, D_hat, m, w_nz, stream);
    if (rc) return rc;
    oriana_counts cm;
    cm.n = n; cm.m = m; cm.nrb = nrb; cm.ncb = ncb; cm.nnz = 1;   /* >0: lets oriana_fixup look at the flags */
    cm.rslots = tot[0]; cm.cslots = tot[1];
    cm.roff = roff; cm.coff = coff; cm.rslice = rslice; cm.cslice = cslice; cm.rowrec = rowrec; cm.ridx = ridx;
    cm.col_perm = nullptr; cm.row_perm = nullptr;
    if (quirk) {
        dq = (float *)(b + L.dq);
        if ((rc = oriana_take_cols_f32(dq, D_hat, n, m, K, stream))) return rc;
    }

    float *prep = (float *)(b + L.prep);
    ORIANA_HIP_CHECK(hipMemsetAsync(prep, 0, 8 * sizeof(float), s));            // the arrival counter
    if ((rc = oriana_factor_prep_pair(FU, FV, log_U_hat, log_V_hat, S_tilde, nullptr, nullptr, n, m, K, prep, stream))) return rc;
    ORIANA_HIP_CHECK(hipMemsetAsync(C, 0, sizeof(float) * m * L.Kp, s));
    ORIANA_HIP_CHECK(hipMemsetAsync(tile_flag, 0, sizeof(int32_t) * nt, s));
    if ((rc = oriana_row_pass(&cm, FU, FV, w_nz, R, s_cs, sw_cs, s_rs, tile_flag, K, stream))) return rc;
    const int variant = (sparse ? 1 : 0) | (weighted ? 2 : 0) | (dq ? 4 : 0);
    if ((rc = oriana_fixup(&cm, tile_flag, s_cs, sw_cs, s_rs, log_U_hat, log_V_hat, S_tilde, S_hat, w_nz, dq, Zi, Zj,
                           Zlog, K, variant, stream))) return rc;
    if (sparse) {
        // S_hat-weighted row sums (sparse_gap.py:95): a second row SpMM with FV * S_hat
        if ((rc = oriana_scale_factor(F2, FV, S_hat, nullptr, m, K, 0, stream))) return rc;
        if ((rc = oriana_row_spmm(&cm, s_rs, w_nz, F2, R, K, stream))) return rc;
    }
    if ((rc = oriana_finalize(Zi, FU, R, nullptr, nullptr, n, K, 1, stream))) return rc;
    // per-gene sums: weighted by D_hat[i, j] (sw), or -- zigap.py:94 -- by D_hat[i, k] on the plain s
    const float *G = FU, *s_for_j = sw_cs ? sw_cs : s_cs;
    if (dq) {
        if ((rc = oriana_scale_factor(G2, FU, dq, nullptr, n, K, 0, stream))) return rc;
        G = G2;
        s_for_j = s_cs;
    }
    if ((rc = oriana_col_pass(&cm, s_for_j, G, C, K, nullptr, 0, stream))) return rc;
    if ((rc = oriana_finalize(Zj, FV, C, nullptr, nullptr, m, K, 1, stream))) return rc;
    if (Zlog) {
        // sum_i r_ijk (lu_ik + lv_jk) = FV (sum_i s FU lu) + FV lv (sum_i s FU), D_hat[i, j]-weighted
        const float *s_log = sw_cs ? sw_cs : s_cs;
        if (dq) {
            ORIANA_HIP_CHECK(hipMemsetAsync(C, 0, sizeof(float) * m * L.Kp, s));
            if ((rc = oriana_col_pass(&cm, s_log, FU, C, K, nullptr, 0, stream))) return rc;
        }
        ORIANA_HIP_CHECK(hipMemsetAsync(C2, 0, sizeof(float) * m * L.Kp, s));
        double *center = (double *)((char *)prep + oriana_prep_center_offset());
        if ((rc = oriana_log_center(center, FU, log_U_hat, Zi, nullptr, n, K, stream))) return rc;
        if ((rc = oriana_scale_factor_centered(G2, FU, log_U_hat, center, nullptr, n, K, stream))) return rc;
        if ((rc = oriana_col_pass(&cm, s_log, G2, C2, K, nullptr, 0, stream))) return rc;
        if ((rc = oriana_finalize_zlog(Zlog, FV, C2, C, log_V_hat, center, nullptr, m, K, stream))) return rc;
    }
    return 0;
}

extern "C" int oriana_zq_gap_f32(float *Z_hat_i, float *Z_hat_j, const float *log_U_hat, const float *log_V_hat,
                                 const float *X, int64_t n, int64_t m, int64_t K, void *ws, int64_t ws_bytes,
                                 void *stream) {
    return zq_dense(Z_hat_i, Z_hat_j, nullptr, log_U_hat, log_V_hat, nullptr, nullptr, nullptr, 0, X, n, m, K, ws,
                    ws_bytes, stream);
}

extern "C" int oriana_zq_zigap_f32(float *DZ_hat_i, float *DZ_hat_j, float *DZ_exp_logsum_hat, const float *log_U_hat,
                                   const float *log_V_hat, const float *D_hat, const float *X, int64_t n, int64_t m,
                                   int64_t K, int reference_quirks, void *ws, int64_t ws_bytes, void *stream) {
    if (!D_hat && n > 0 && m > 0) return ORIANA_EINVAL;
    if (!DZ_exp_logsum_hat && m > 0) return ORIANA_EINVAL;
    return zq_dense(DZ_hat_i, DZ_hat_j, DZ_exp_logsum_hat, log_U_hat, log_V_hat, nullptr, nullptr, D_hat,
                    reference_quirks ? 1 : 0, X, n, m, K, ws, ws_bytes, stream);
}

extern "C" int oriana_zq_sparse_gap_f32(float *SZ_hat_i, float *Z_hat_j, float *Z_exp_logsum_hat,
                                        const float *log_U_hat, const float *log_V_hat, const float *S_tilde,
                                        const float *S_hat, const float *X, int64_t n, int64_t m, int64_t K, void *ws,
                                        int64_t ws_bytes, void *stream) {
    if ((!S_tilde || !S_hat || !Z_exp_logsum_hat) && m > 0) return ORIANA_EINVAL;
    return zq_dense(SZ_hat_i, Z_hat_j, Z_exp_logsum_hat, log_U_hat, log_V_hat, S_tilde, S_hat, nullptr, 0, X, n, m, K,
                    ws, ws_bytes, stream);
}

extern "C" int oriana_zq_sparse_zigap_f32(float *DSZ_hat, float *DZ_hat, float *DZ_exp_logsum_hat,
                                          const float *log_U_hat, const float *log_V_hat, const float *S_tilde,
                                          const float *S_hat, const float *D_hat, const float *X, int64_t n, int64_t m,
                                          int64_t K, void *ws, int64_t ws_bytes, void *stream) {
    if ((!S_tilde || !S_hat || !DZ_exp_logsum_hat) && m > 0) return ORIANA_EINVAL;
    if (!D_hat && n > 0 && m > 0) return ORIANA_EINVAL;
    return zq_dense(DSZ_hat, DZ_hat, DZ_exp_logsum_hat, log_U_hat, log_V_hat, S_tilde, S_hat, D_hat, 0, X, n, m, K, ws,
                    ws_bytes, stream);
}

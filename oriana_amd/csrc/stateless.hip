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
    size_t tile_cnt, tile_off, row_ptr, col_ptr, tile_flag, rowrec, ridx, s_col, FU, FV, R, C, total;
};

static inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

static WsLayout ws_layout(int64_t n, int64_t m, int64_t K, int64_t nnz_bound) {
    WsLayout L;
    const int64_t nrb = (n + TILE - 1) / TILE, ncb = (m + TILE - 1) / TILE;
    L.nt = nrb * ncb;
    L.Kp = oriana_kpad(K);
    size_t o = 0;
    const int64_t nt1 = L.nt > 0 ? L.nt : 1, nz1 = nnz_bound > 0 ? nnz_bound : 1;
    const int64_t n1 = n > 0 ? n : 1, m1 = m > 0 ? m : 1;
    L.tile_cnt = o;  o = align256(o + sizeof(int32_t) * nt1);
    L.tile_off = o;  o = align256(o + sizeof(int64_t) * (nt1 + 1));
    L.row_ptr = o;   o = align256(o + sizeof(uint32_t) * nt1 * (TILE + 1));
    L.col_ptr = o;   o = align256(o + sizeof(uint32_t) * nt1 * (TILE + 1));
    L.tile_flag = o; o = align256(o + sizeof(int32_t) * nt1);
    L.rowrec = o;    o = align256(o + sizeof(oriana_rowrec) * nz1);
    L.ridx = o;      o = align256(o + nz1);
    L.s_col = o;     o = align256(o + sizeof(float) * nz1);
    L.FU = o;        o = align256(o + sizeof(float) * n1 * L.Kp);
    L.FV = o;        o = align256(o + sizeof(float) * m1 * L.Kp);
    L.R = o;         o = align256(o + sizeof(float) * n1 * L.Kp);
    L.C = o;         o = align256(o + sizeof(float) * m1 * L.Kp);
    L.total = o;
    return L;
}

}  // namespace oriana

using namespace oriana;

extern "C" int64_t oriana_zq_workspace_bytes(int64_t n, int64_t m, int64_t K, int64_t nnz_bound) {
    if (n < 0 || m < 0 || nnz_bound < 0 || oriana_kpad(K) == 0) return 0;
    return (int64_t)ws_layout(n, m, K, nnz_bound).total;
}

extern "C" int oriana_zq_gap_f32(float *Z_hat_i, float *Z_hat_j, const float *log_U_hat, const float *log_V_hat,
                                 const float *X, int64_t n, int64_t m, int64_t K, void *ws, int64_t ws_bytes,
                                 void *stream) {
    if (n < 0 || m < 0 || K <= 0) return ORIANA_EINVAL;
    if (oriana_kpad(K) == 0) return ORIANA_EKRANGE;
    hipStream_t s = (hipStream_t)stream;
    if (n > 0 && (!Z_hat_i || !log_U_hat)) return ORIANA_EINVAL;
    if (m > 0 && (!Z_hat_j || !log_V_hat)) return ORIANA_EINVAL;
    // callee zero-fills the outputs (gap.py:69-70)
    if (n > 0) ORIANA_HIP_CHECK(hipMemsetAsync(Z_hat_i, 0, sizeof(float) * n * K, s));
    if (m > 0) ORIANA_HIP_CHECK(hipMemsetAsync(Z_hat_j, 0, sizeof(float) * m * K, s));
    if (n == 0 || m == 0) return 0;
    if (!X || !ws || ((uintptr_t)ws & 255)) return ORIANA_EINVAL;
    // the workspace must at least hold the layout with zero records; its record capacity follows
    const WsLayout L0 = ws_layout(n, m, K, 0);
    if ((size_t)ws_bytes < L0.total) return ORIANA_EINVAL;
    // largest nnz_bound whose layout fits in ws_bytes: 8 + 1 + 4 bytes per record plus alignment
    int64_t cap = ((int64_t)ws_bytes - (int64_t)L0.total) / 13;
    while (cap > 0 && ws_layout(n, m, K, cap).total > (size_t)ws_bytes) cap -= 64;
    if (cap < 0) cap = 0;
    const WsLayout L = ws_layout(n, m, K, cap);
    char *b = (char *)ws;
    int32_t *tile_cnt = (int32_t *)(b + L.tile_cnt);
    int64_t *tile_off = (int64_t *)(b + L.tile_off);
    uint32_t *row_ptr = (uint32_t *)(b + L.row_ptr), *col_ptr = (uint32_t *)(b + L.col_ptr);
    int32_t *tile_flag = (int32_t *)(b + L.tile_flag);
    oriana_rowrec *rowrec = (oriana_rowrec *)(b + L.rowrec);
    uint8_t *ridx = (uint8_t *)(b + L.ridx);
    float *s_col = (float *)(b + L.s_col), *FU = (float *)(b + L.FU), *FV = (float *)(b + L.FV);
    float *R = (float *)(b + L.R), *C = (float *)(b + L.C);
    const int64_t nrb = (n + TILE - 1) / TILE, ncb = (m + TILE - 1) / TILE;

    int rc = oriana_pack_count(X, 0, n, m, m, 0, ncb, tile_cnt, row_ptr, col_ptr, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, s, tile_off, tile_cnt, L.nt);
    ORIANA_LAUNCH_CHECK();
    // the one host synchronisation of this entry point: nnz decides whether the workspace is large enough
    int64_t nnz = 0;
    ORIANA_HIP_CHECK(hipMemcpyAsync(&nnz, tile_off + L.nt, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    ORIANA_HIP_CHECK(hipStreamSynchronize(s));
    if (nnz > cap) return ORIANA_EINVAL;
    rc = oriana_pack_fill(X, 0, n, m, m, 0, ncb, tile_off, row_ptr, col_ptr, rowrec, ridx, nullptr, 0, nullptr, stream);
    if (rc) return rc;
    oriana_counts cm;
    cm.n = n; cm.m = m; cm.nrb = nrb; cm.ncb = ncb; cm.nnz = nnz;
    cm.tile_off = tile_off; cm.row_ptr = row_ptr; cm.col_ptr = col_ptr; cm.rowrec = rowrec; cm.ridx = ridx;

    if ((rc = oriana_factor_prep(FU, nullptr, log_U_hat, nullptr, n, K, stream))) return rc;
    if ((rc = oriana_factor_prep(FV, nullptr, log_V_hat, nullptr, m, K, stream))) return rc;
    ORIANA_HIP_CHECK(hipMemsetAsync(C, 0, sizeof(float) * m * L.Kp, s));
    ORIANA_HIP_CHECK(hipMemsetAsync(tile_flag, 0, sizeof(int32_t) * L.nt, s));
    if ((rc = oriana_row_pass(&cm, FU, FV, nullptr, nullptr, R, s_col, nullptr, nullptr, tile_flag, K, stream))) return rc;
    if ((rc = oriana_fixup(&cm, tile_flag, s_col, nullptr, nullptr, log_U_hat, log_V_hat, nullptr, nullptr, nullptr,
                           nullptr, Z_hat_i, Z_hat_j, nullptr, K, 0, stream))) return rc;
    if ((rc = oriana_col_pass(&cm, s_col, FU, C, K, stream))) return rc;
    if ((rc = oriana_finalize(Z_hat_i, FU, R, nullptr, n, K, 1, stream))) return rc;
    if ((rc = oriana_finalize(Z_hat_j, FV, C, nullptr, m, K, 1, stream))) return rc;
    return 0;
}

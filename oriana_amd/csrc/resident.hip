// resident.hip -- [r5] the planning of the resident layout as C host code, and a RESIDENT HANDLE of the C ABI.
//
// What a sweep's kernels depend on beyond the packed records -- the gene order and the dense-gene set of a hybrid layout,
// the column work list, the row split, the splits of the dense-gene kernels -- was Python (oriana_amd/engine.py) through
// round 4: a host in another language could only call the stateless entries (csrc/stateless.hip), which repack X on every
// call.  Here the same decisions are plain C functions over host arrays (oriana_plan_*: no device needed, tested on the
// CPU against a NumPy restatement), engine.py calls them, and oriana_counts_create_* / oriana_zq_*_resident keep a count
// matrix packed across calls for a non-Python host: the reference calls its loop nest once per step() with the same X
// (oriana/models/gap.py:89-94).
#include "common.h"
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include <new>

#pragma clang fp contract(off)          // the cost model below is checked bit for bit against a NumPy restatement

namespace oriana {

// ---------------------------------------------------------------------------------------------------------------
// kernels of the one-time build
// ---------------------------------------------------------------------------------------------------------------
// per-gene non-zero counts and the number of entries a uint16 block cannot hold (negative, >= 65535, non-integer)
__global__ __launch_bounds__(256) void k_col_stats(const float *__restrict__ X, int64_t rows, int64_t m, int64_t ldx,
                                                   unsigned long long *__restrict__ nnz, unsigned long long *__restrict__ bad) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    const int64_t r0 = (int64_t)blockIdx.y * 256, r1 = (r0 + 256 < rows) ? r0 + 256 : rows;
    unsigned long long cn = 0, cb = 0;
    for (int64_t r = r0; r < r1; ++r) {
        const float v = X[r * ldx + j];
        cn += (v != 0.0f) ? 1u : 0u;
        cb += (v < 0.0f || v >= 65535.0f || v != floorf(v)) ? 1u : 0u;
    }
    if (cn) atomicAdd(&nnz[j], cn);
    if (cb) atomicAdd(&bad[j], cb);
}

// out[r, c] = X[r, perm[c]]  (the packer reads the already column-permuted chunk)
__global__ __launch_bounds__(256) void k_gather_cols(float *__restrict__ out, const float *__restrict__ X,
                                                     const int32_t *__restrict__ perm, int64_t rows, int64_t m, int64_t ldx) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= m) return;
    const int64_t src = perm ? (int64_t)perm[c] : c;
    for (int64_t r = blockIdx.y; r < rows; r += gridDim.y) out[r * m + c] = X[r * ldx + src];     // (gridDim.y <= 65535)
}

// rows [r0, r0 + rows) of a CSR matrix into a zeroed dense chunk: one work-group per row
__global__ __launch_bounds__(256) void k_csr_scatter(float *__restrict__ out, const int64_t *__restrict__ indptr,
                                                     const int32_t *__restrict__ indices, const float *__restrict__ data,
                                                     int64_t e0, int64_t m) {
    const int64_t r = blockIdx.x;
    const int64_t a = indptr[r] - e0, b = indptr[r + 1] - e0;
    for (int64_t e = a + threadIdx.x; e < b; e += 256) atomicAdd(&out[r * m + indices[e]], data[e]);      // (duplicates add up)
}

// longest column-side slice of every tile, in iterations of 64 slots (the cost unit of the column work list)
__global__ __launch_bounds__(256) void k_tile_longest(int32_t *__restrict__ out, const uint32_t *__restrict__ cslice, int64_t nt) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= nt) return;
    uint32_t mx = 0;
    for (int s = 0; s < 16; ++s) {
        const uint32_t len = (cslice[t * 17 + s + 1] - cslice[t * 17 + s]) >> 6;
        mx = len > mx ? len : mx;
    }
    out[t] = (int32_t)mx;
}

// w_nz[slot] = D[cell, gene] at every stored entry (row-side slots; the per-entry weights of zigap.py:93-95): what
// oriana_pack_fill gathers while packing, for a matrix D that changes from call to call
__global__ __launch_bounds__(256) void k_gather_nz(oriana_counts cm, const float *__restrict__ D, int64_t ld,
                                                   float *__restrict__ w_nz) {
    const int64_t t = blockIdx.x;
    const int64_t rb = t / cm.ncb, cb = t - rb * cm.ncb;
    const int64_t rbase = cm.roff[t];
    for (int sl = 0; sl < 16; ++sl) {
        const uint32_t s0 = cm.rslice[t * 17 + sl], s1 = cm.rslice[t * 17 + sl + 1];
        for (uint32_t slot = s0 + threadIdx.x; slot < s1; slot += 256) {
            const oriana_rowrec rec = cm.rowrec[rbase + slot];
            float w = 0.0f;
            if (rec.x != 0.f) {
                const int64_t ip = rb * TILE + sl * 16 + (int)(((slot - s0) & 63u) >> 2);
                const int64_t jp = cb * TILE + rec.col;
                const int64_t i = cm.row_perm ? (int64_t)cm.row_perm[ip] : ip;
                const int64_t j = cm.col_perm ? (int64_t)cm.col_perm[jp] : jp;
                w = D[i * ld + j];
            }
            w_nz[rbase + slot] = w;
        }
    }
}

__global__ __launch_bounds__(1024) void k_scan_i32(int64_t *__restrict__ off, const int32_t *__restrict__ cnt, int64_t nt) {
    // (the scan of stateless.hip: one work-group, nt is at most a few 1e5)
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

}  // namespace oriana

using namespace oriana;

// ---------------------------------------------------------------------------------------------------------------
// planning (host arrays in, host arrays out; no device)
// ---------------------------------------------------------------------------------------------------------------
extern "C" int64_t oriana_device_cus(void) {
    static const long forced = [] { const char *e = getenv("ORIANA_CUS"); return e ? atol(e) : 0L; }();     // tests, tuning runs
    if (forced > 0) return forced;
    static int cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return 256; }
    if (cached[dev] > 0) return cached[dev];
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) { (void)hipGetLastError(); return 256; }
    cached[dev] = v;
    return v;
}

extern "C" int oriana_plan_gene_order(const int64_t *col_nnz, const int64_t *bad, int64_t m, int64_t n_total,
                                      double dense_density, double min_share, int32_t *order, int64_t *gd_out) {
    if (m < 0 || (m > 0 && (!col_nnz || !order)) || !gd_out || m > 0x7fffffffLL) return ORIANA_EINVAL;
    *gd_out = 0;
    std::vector<int32_t> ord((size_t)m);
    for (int64_t j = 0; j < m; ++j) ord[(size_t)j] = (int32_t)j;
    // decreasing non-zero count, ties in the caller's gene order
    std::stable_sort(ord.begin(), ord.end(), [&](int32_t a, int32_t b) { return col_nnz[a] > col_nnz[b]; });
    int64_t gd = 0;
    std::vector<int32_t> cand;
    if (dense_density > 0.0) {
        const double thr = dense_density * (double)(n_total > 1 ? n_total : 1);
        for (int64_t p = 0; p < m; ++p) {
            const int32_t j = ord[(size_t)p];
            if ((double)col_nnz[j] >= thr && (!bad || bad[j] == 0) && col_nnz[j] > 0) cand.push_back(j);
        }
        gd = ((int64_t)cand.size() / 32) * 32;
        if (gd > 0 && min_share > 0.0) {
            double tot = 0.0, got = 0.0;
            for (int64_t j = 0; j < m; ++j) tot += (double)col_nnz[j];
            for (int64_t p = 0; p < gd; ++p) got += (double)col_nnz[cand[(size_t)p]];
            if (!(tot > 0.0) || got < min_share * tot) gd = 0;
        }
    }
    if (gd == 0) {
        for (int64_t p = 0; p < m; ++p) order[p] = ord[(size_t)p];
        return 0;
    }
    std::vector<char> taken((size_t)m, 0);
    for (int64_t p = 0; p < gd; ++p) { order[p] = cand[(size_t)p]; taken[(size_t)cand[(size_t)p]] = 1; }
    int64_t q = gd;
    for (int64_t p = 0; p < m; ++p)
        if (!taken[(size_t)ord[(size_t)p]]) order[q++] = ord[(size_t)p];
    *gd_out = gd;
    return 0;
}

extern "C" int64_t oriana_plan_col_work_capacity(int64_t nrb, int64_t ncb, int64_t width) {
    if (nrb <= 0 || ncb <= 0 || width <= 0) return 0;
    return nrb * ((ncb + width - 1) / width);           // (a column block is cut into at most nrb row ranges)
}

namespace {
struct ColItem { int32_t cb, a, e; };

// one cut of every column block at equal-cost points into about n_items items in all (NumPy's linspace / searchsorted /
// unique, restated)
void col_work_build(const std::vector<std::vector<double>> &cums, int64_t nrb, double total, int64_t n_items, std::vector<ColItem> &out) {
    out.clear();
    double target = total / (double)n_items;
    if (!(target > 1e-9)) target = 1e-9;
    std::vector<int64_t> edges;
    for (size_t cb = 0; cb < cums.size(); ++cb) {
        const std::vector<double> &cum = cums[cb];
        const double last = cum[(size_t)nrb];
        double want = nearbyint(last / target);              // Python's round(): half to even
        if (want < 1.0) want = 1.0;
        if (want > (double)nrb) want = (double)nrb;
        const int64_t nb = (int64_t)want;
        edges.clear();
        edges.push_back(0);
        const double step = last / (double)nb;
        for (int64_t i = 1; i < nb; ++i) {
            const double pt = (double)i * step;
            const int64_t idx = (int64_t)(std::lower_bound(cum.begin(), cum.end(), pt) - cum.begin());
            edges.push_back(idx);
        }
        edges.push_back(nrb);
        std::sort(edges.begin(), edges.end());
        edges.erase(std::unique(edges.begin(), edges.end()), edges.end());
        for (size_t k = 0; k + 1 < edges.size(); ++k)
            if (edges[k + 1] > edges[k]) out.push_back(ColItem{(int32_t)cb, (int32_t)edges[k], (int32_t)edges[k + 1]});
    }
}
}  // namespace

// Work list of the column pass: (column block, first row block, end row block) items of about equal COST, ordered by row band.
// A column block is `width` adjacent column tiles.  Cost of a block's row block = its longest column slice (the work-group
// advances at the pace of its slowest wave; tile_iters [nrb * ncb], iterations of 64 slots) * 1.45 + 3.2 for staging the 256
// factor rows (measured on MI355X, in microseconds).  One 1024-thread group per CU: the pass advances in rounds of `cus` items
// and a partly filled last round costs a whole one, so the list is re-cut until the item count lands just below a multiple
// of `cus` (rounds = 0: the first cut).  target_items = 0: 25-50 tiles per item, between 9 and 36 items per CU.
extern "C" int oriana_plan_col_work(const int32_t *tile_iters, int64_t nrb, int64_t ncb, int64_t width, int64_t cus,
                                    int64_t target_items, int rounds, int sum_price, int32_t *items, int64_t cap,
                                    int64_t *n_items) {
    if (!n_items || nrb < 0 || ncb < 0 || width <= 0 || cus <= 0 || target_items < 0) return ORIANA_EINVAL;
    *n_items = 0;
    if (nrb == 0 || ncb == 0) return 0;
    if (!tile_iters || !items || nrb > 0x7fffffffLL || ncb > 0x7fffffffLL) return ORIANA_EINVAL;
    const int64_t nblk = (ncb + width - 1) / width;
    std::vector<std::vector<double>> cums((size_t)nblk, std::vector<double>((size_t)nrb + 1, 0.0));
    double total = 0.0;
    for (int64_t cb = 0; cb < nblk; ++cb) {
        std::vector<double> &cum = cums[(size_t)cb];
        for (int64_t rb = 0; rb < nrb; ++rb) {
            double nit = 0.0;
            for (int64_t w = 0; w < width; ++w) {
                const int64_t c = cb * width + w;
                const double v = c < ncb ? (double)tile_iters[rb * ncb + c] : 0.0;
                nit = sum_price ? nit + v : (v > nit ? v : nit);
            }
            const double scaled = nit * 1.45;
            const double cost = scaled + 3.2;
            cum[(size_t)rb + 1] = cum[(size_t)rb] + cost;
        }
        total += cum[(size_t)nrb];
    }
    const bool explicit_target = target_items > 0;
    const int64_t nt = nrb * ncb;
    if (!explicit_target) {
        target_items = nt / (50 * width);
        if (target_items < 9 * cus) target_items = 9 * cus;
        if (target_items > 36 * cus) target_items = 36 * cus;
    }
    std::vector<ColItem> cur, trial;
    col_work_build(cums, nrb, total, target_items, cur);
    if (!explicit_target && rounds && (int64_t)cur.size() > cus) {
        const int64_t want = ((int64_t)cur.size() / cus) * cus;
        const int64_t slack = std::max<int64_t>(1, 24 * cus / 256), back = std::max<int64_t>(1, 8 * cus / 256);
        int64_t t = target_items;
        trial = cur;
        for (int it = 0; it < 9; ++it) {
            if (want - slack <= (int64_t)trial.size() && (int64_t)trial.size() <= want) { cur = trial; break; }
            if (it == 8) break;
            const double nxt = nearbyint((double)t * (double)(want - back) / (double)std::max<int64_t>((int64_t)trial.size(), 1));
            t = std::max<int64_t>(cus, (int64_t)nxt);
            col_work_build(cums, nrb, total, t, trial);
        }
    }
    // by row band: the items that run at the same time stage the same factor rows
    std::stable_sort(cur.begin(), cur.end(), [](const ColItem &x, const ColItem &y) {
        const int64_t kx = (int64_t)x.a + x.e, ky = (int64_t)y.a + y.e;
        return kx != ky ? kx < ky : x.cb < y.cb;
    });
    if ((int64_t)cur.size() > cap) return ORIANA_EINVAL;
    for (size_t k = 0; k < cur.size(); ++k) { items[3 * k] = cur[k].cb; items[3 * k + 1] = cur[k].a; items[3 * k + 2] = cur[k].e; }
    *n_items = (int64_t)cur.size();
    return 0;
}

// splits of the dense-gene kernels of a hybrid layout: gene ranges of the row kernel (two work-groups per CU at most) and
// cell ranges of the gene-side kernel (four per CU, a multiple of 8 cell tiles)
extern "C" int oriana_plan_dense_splits(int64_t n, int64_t gd, int64_t cus, int64_t *gene_splits, int64_t *cell_splits) {
    if (n < 0 || gd < 0 || gd % 32 != 0 || cus <= 0 || !gene_splits || !cell_splits) return ORIANA_EINVAL;
    const int64_t ngt = gd / 32, nblk = std::max<int64_t>((n + TILE - 1) / TILE, 1);
    *gene_splits = std::max<int64_t>(1, std::min<int64_t>(ngt, (2 * cus + nblk - 1) / nblk));
    const int64_t groups = std::max<int64_t>((ngt + 7) / 8, 1);
    const int64_t per = ((4 * cus + groups - 1) / groups + 7) / 8 * 8;
    *cell_splits = std::max<int64_t>(1, std::min<int64_t>((n + 31) / 32, per));
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// the resident handle
// ---------------------------------------------------------------------------------------------------------------
struct oriana_resident {
    int64_t n = 0, m = 0, K = 0, Kp = 0, gd = 0, ms = 0, nrb = 0, ncb = 0, nt = 0, nnz = 0, rslots = 0, cslots = 0;
    int64_t cus = 256, bytes = 0;
    std::vector<void *> allocs;
    // layout
    int32_t *col_perm = nullptr;
    int32_t *tile_nnz = nullptr, *tile_rslots = nullptr, *tile_cslots = nullptr, *tile_flag = nullptr;
    int64_t *roff = nullptr, *coff = nullptr;
    uint32_t *rslice = nullptr, *cslice = nullptr;
    oriana_rowrec *rowrec = nullptr;
    uint8_t *ridx = nullptr;
    uint16_t *dense_x = nullptr;
    oriana_counts cm;
    oriana_dense dn;
    // plans
    oriana_row_split split;
    int32_t *col_work = nullptr;
    int64_t n_col_work = 0, dn_gene_splits = 1, dn_cell_splits = 1, nslab = 1;
    // workspace of a call
    float *FU = nullptr, *FV = nullptr, *R = nullptr, *C = nullptr, *s_cs = nullptr, *prep = nullptr;
    float *dn_S = nullptr, *dn_imgV = nullptr, *dn_imgU = nullptr;
    int32_t *dn_flag = nullptr;
    // the ZI / sparse nests (allocated on first use)
    float *w_nz = nullptr, *sw_cs = nullptr, *s_rs = nullptr, *F2 = nullptr, *G2 = nullptr, *C2 = nullptr, *dq = nullptr;
    float *GQ = nullptr;
    // [r6] work list of the dual column pass (one column tile per item) and the caller's declaration about D_hat
    int32_t *col_work1 = nullptr;
    int64_t n_col_work1 = 0;
    int unit_dropout = 0;
};

namespace {

template <typename T>
int dev_alloc(oriana_resident *h, T **p, size_t count, bool zero, hipStream_t s) {
    const size_t bytes = sizeof(T) * (count > 0 ? count : 1);
    void *q = nullptr;
    if (hipMalloc(&q, bytes) != hipSuccess) { (void)hipGetLastError(); return -1000 - (int)hipErrorOutOfMemory; }
    h->allocs.push_back(q);
    h->bytes += (int64_t)bytes;
    if (zero && hipMemsetAsync(q, 0, bytes, s) != hipSuccess) return -1000 - (int)hipGetLastError();
    *p = static_cast<T *>(q);
    return 0;
}

void resident_free(oriana_resident *h) {
    if (!h) return;
    for (void *q : h->allocs) (void)hipFree(q);
    delete h;
}

#define RES_TRY(expr) do { const int _rc = (expr); if (_rc) return _rc; } while (0)

// A source of dense float32 row chunks in the caller's gene order: a dense device matrix (no copy) or CSR host arrays
// (expanded into a device buffer, chunk by chunk)
struct ChunkSource {
    const float *X = nullptr; int64_t ldx = 0;                                   // dense
    const int64_t *indptr = nullptr; const int32_t *indices = nullptr; const float *data = nullptr;   // CSR (host)
    float *buf = nullptr; int64_t *d_indptr = nullptr; int32_t *d_indices = nullptr; float *d_data = nullptr;
    int64_t max_e = 0;
    int64_t m = 0;
    int get(int64_t r0, int64_t rows, const float **out, int64_t *ld, hipStream_t s) {
        if (X) { *out = X + r0 * ldx; *ld = ldx; return 0; }
        const int64_t e0 = indptr[r0], e1 = indptr[r0 + rows];
        if (e1 < e0 || e1 - e0 > max_e) return ORIANA_EINVAL;
        ORIANA_HIP_CHECK(hipMemsetAsync(buf, 0, sizeof(float) * rows * m, s));
        ORIANA_HIP_CHECK(hipMemcpyAsync(d_indptr, indptr + r0, sizeof(int64_t) * (rows + 1), hipMemcpyHostToDevice, s));
        if (e1 > e0) {
            ORIANA_HIP_CHECK(hipMemcpyAsync(d_indices, indices + e0, sizeof(int32_t) * (e1 - e0), hipMemcpyHostToDevice, s));
            ORIANA_HIP_CHECK(hipMemcpyAsync(d_data, data + e0, sizeof(float) * (e1 - e0), hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(k_csr_scatter, dim3((unsigned)rows), dim3(256), 0, s, buf, d_indptr, d_indices, d_data, e0, m);
            ORIANA_LAUNCH_CHECK();
        }
        ORIANA_HIP_CHECK(hipStreamSynchronize(s));          // (the host arrays of the next chunk may be staged by the caller's allocator)
        *out = buf; *ld = m;
        return 0;
    }
};

int resident_build(oriana_resident *h, ChunkSource &src, double dense_density, hipStream_t s) {
    const int64_t n = h->n, m = h->m, K = h->K;
    h->cus = oriana_device_cus();
    // row chunks of about 512 MB, whole row blocks
    int64_t chunk = ((int64_t)(512u << 20) / (m * 4)) / TILE * TILE;
    if (chunk < TILE) chunk = TILE;
    if (chunk > (n + TILE - 1) / TILE * TILE) chunk = (n + TILE - 1) / TILE * TILE;
    if (chunk / TILE > 65535) chunk = (int64_t)65535 * TILE;
    std::vector<void *> scratch;                             // freed at the end of the build
    auto tmp_alloc = [&](void **p, size_t bytes) -> int {
        if (hipMalloc(p, bytes ? bytes : 1) != hipSuccess) { (void)hipGetLastError(); return -1000 - (int)hipErrorOutOfMemory; }
        scratch.push_back(*p);
        return 0;
    };
    struct Cleanup { std::vector<void *> &v; ~Cleanup() { for (void *q : v) (void)hipFree(q); } } cleanup{scratch};
    if (!src.X) {
        int64_t max_e = 0;
        for (int64_t r0 = 0; r0 < n; r0 += chunk) {
            const int64_t r1 = std::min(n, r0 + chunk);
            max_e = std::max(max_e, src.indptr[r1] - src.indptr[r0]);
        }
        src.max_e = max_e;
        RES_TRY(tmp_alloc((void **)&src.buf, sizeof(float) * chunk * m));
        RES_TRY(tmp_alloc((void **)&src.d_indptr, sizeof(int64_t) * (chunk + 1)));
        RES_TRY(tmp_alloc((void **)&src.d_indices, sizeof(int32_t) * std::max<int64_t>(max_e, 1)));
        RES_TRY(tmp_alloc((void **)&src.d_data, sizeof(float) * std::max<int64_t>(max_e, 1)));
    }
    // ---- 1. per-gene statistics -> gene order, dense set
    unsigned long long *d_nnz = nullptr, *d_bad = nullptr;
    RES_TRY(tmp_alloc((void **)&d_nnz, sizeof(unsigned long long) * m));
    RES_TRY(tmp_alloc((void **)&d_bad, sizeof(unsigned long long) * m));
    ORIANA_HIP_CHECK(hipMemsetAsync(d_nnz, 0, sizeof(unsigned long long) * m, s));
    ORIANA_HIP_CHECK(hipMemsetAsync(d_bad, 0, sizeof(unsigned long long) * m, s));
    for (int64_t r0 = 0; r0 < n; r0 += chunk) {
        const int64_t rows = std::min(n, r0 + chunk) - r0;
        const float *Xc; int64_t ld;
        RES_TRY(src.get(r0, rows, &Xc, &ld, s));
        hipLaunchKernelGGL(k_col_stats, dim3((unsigned)((m + 255) / 256), (unsigned)((rows + 255) / 256)), dim3(256), 0, s, Xc, rows, m, ld, d_nnz, d_bad);
        ORIANA_LAUNCH_CHECK();
    }
    std::vector<int64_t> col_nnz((size_t)m), bad((size_t)m);
    ORIANA_HIP_CHECK(hipMemcpyAsync(col_nnz.data(), d_nnz, sizeof(int64_t) * m, hipMemcpyDeviceToHost, s));
    ORIANA_HIP_CHECK(hipMemcpyAsync(bad.data(), d_bad, sizeof(int64_t) * m, hipMemcpyDeviceToHost, s));
    ORIANA_HIP_CHECK(hipStreamSynchronize(s));
    std::vector<int32_t> order((size_t)m);
    int64_t gd = 0;
    const double dd = (dense_density > 0.0 && oriana_dense_supported(K)) ? dense_density : 0.0;
    RES_TRY(oriana_plan_gene_order(col_nnz.data(), bad.data(), m, n, dd, 0.0, order.data(), &gd));
    h->gd = gd; h->ms = m - gd;
    h->nrb = (n + TILE - 1) / TILE; h->ncb = (h->ms + TILE - 1) / TILE; h->nt = h->nrb * h->ncb;
    RES_TRY(dev_alloc(h, &h->col_perm, (size_t)m, false, s));
    ORIANA_HIP_CHECK(hipMemcpyAsync(h->col_perm, order.data(), sizeof(int32_t) * m, hipMemcpyHostToDevice, s));
    // ---- 2. tile tables
    const size_t nt1 = (size_t)std::max<int64_t>(h->nt, 1);
    RES_TRY(dev_alloc(h, &h->tile_nnz, nt1, true, s));
    RES_TRY(dev_alloc(h, &h->tile_rslots, nt1, true, s));
    RES_TRY(dev_alloc(h, &h->tile_cslots, nt1, true, s));
    RES_TRY(dev_alloc(h, &h->tile_flag, nt1, true, s));
    RES_TRY(dev_alloc(h, &h->rslice, nt1 * 17, true, s));
    RES_TRY(dev_alloc(h, &h->cslice, nt1 * 17, true, s));
    RES_TRY(dev_alloc(h, &h->roff, nt1 + 1, true, s));
    RES_TRY(dev_alloc(h, &h->coff, nt1 + 1, true, s));
    float *perm_buf = nullptr;
    RES_TRY(tmp_alloc((void **)&perm_buf, sizeof(float) * chunk * m));
    auto permuted = [&](int64_t r0, int64_t rows) -> int {
        const float *Xc; int64_t ld;
        RES_TRY(src.get(r0, rows, &Xc, &ld, s));
        hipLaunchKernelGGL(k_gather_cols, dim3((unsigned)((m + 255) / 256), (unsigned)std::min<int64_t>(rows, 65535)), dim3(256), 0, s, perm_buf, Xc,
                           h->col_perm, rows, m, ld);
        ORIANA_LAUNCH_CHECK();
        return 0;
    };
    if (h->ms > 0) {
        for (int64_t r0 = 0; r0 < n; r0 += chunk) {
            const int64_t rows = std::min(n, r0 + chunk) - r0;
            RES_TRY(permuted(r0, rows));
            RES_TRY(oriana_pack_count(perm_buf + gd, 0, rows, h->ms, m, r0 / TILE, h->ncb, h->tile_nnz, h->tile_rslots, h->tile_cslots,
                                      h->rslice, h->cslice, s));
        }
        hipLaunchKernelGGL(k_scan_i32, dim3(1), dim3(1024), 0, s, h->roff, h->tile_rslots, h->nt);
        hipLaunchKernelGGL(k_scan_i32, dim3(1), dim3(1024), 0, s, h->coff, h->tile_cslots, h->nt);
        ORIANA_LAUNCH_CHECK();
    }
    int64_t tot[2] = {0, 0};
    std::vector<int32_t> tile_rslots_h((size_t)nt1, 0), tile_nnz_h((size_t)nt1, 0);
    if (h->nt > 0) {
        ORIANA_HIP_CHECK(hipMemcpyAsync(&tot[0], h->roff + h->nt, sizeof(int64_t), hipMemcpyDeviceToHost, s));
        ORIANA_HIP_CHECK(hipMemcpyAsync(&tot[1], h->coff + h->nt, sizeof(int64_t), hipMemcpyDeviceToHost, s));
        ORIANA_HIP_CHECK(hipMemcpyAsync(tile_rslots_h.data(), h->tile_rslots, sizeof(int32_t) * h->nt, hipMemcpyDeviceToHost, s));
        ORIANA_HIP_CHECK(hipMemcpyAsync(tile_nnz_h.data(), h->tile_nnz, sizeof(int32_t) * h->nt, hipMemcpyDeviceToHost, s));
    }
    ORIANA_HIP_CHECK(hipStreamSynchronize(s));
    h->rslots = tot[0]; h->cslots = tot[1];
    int64_t nnz_sparse = 0;
    for (int64_t t = 0; t < h->nt; ++t) nnz_sparse += tile_nnz_h[(size_t)t];
    // ---- 3. records (padding slots: x == 0, row index 0), the dense block
    RES_TRY(dev_alloc(h, &h->rowrec, (size_t)std::max<int64_t>(h->rslots, 1), true, s));
    RES_TRY(dev_alloc(h, &h->ridx, (size_t)std::max<int64_t>(h->cslots, 1), true, s));
    const int64_t nct = h->nrb * 8, ngt = gd / 32;
    if (gd > 0) RES_TRY(dev_alloc(h, &h->dense_x, (size_t)(nct * ngt * 1024), true, s));
    for (int64_t r0 = 0; r0 < n; r0 += chunk) {
        const int64_t rows = std::min(n, r0 + chunk) - r0;
        RES_TRY(permuted(r0, rows));
        if (gd > 0) RES_TRY(oriana_dense_pack(perm_buf, 0, rows, gd, m, r0 / 32, h->dense_x, s));
        if (h->ms > 0)
            RES_TRY(oriana_pack_fill(perm_buf + gd, 0, rows, h->ms, m, r0 / TILE, h->ncb, h->roff, h->coff, h->rslice, h->cslice, h->rowrec,
                                     h->ridx, nullptr, 0, nullptr, s));
    }
    int64_t nnz_dense = 0;
    for (int64_t p = 0; p < gd; ++p) nnz_dense += col_nnz[(size_t)order[(size_t)p]];
    h->nnz = nnz_sparse + nnz_dense;
    oriana_counts &cm = h->cm;
    cm.n = n; cm.m = h->ms; cm.nrb = h->nrb; cm.ncb = h->ncb; cm.nnz = nnz_sparse; cm.rslots = h->rslots; cm.cslots = h->cslots;
    cm.roff = h->roff; cm.coff = h->coff; cm.rslice = h->rslice; cm.cslice = h->cslice; cm.rowrec = h->rowrec; cm.ridx = h->ridx;
    cm.col_perm = h->col_perm + gd; cm.row_perm = nullptr;
    h->dn.n = n; h->dn.gd = gd; h->dn.nct = nct; h->dn.x = h->dense_x;
    // ---- 4. plans: row split (equal-cost gene ranges), column work list, dense-gene splits
    h->split.nfull = (int32_t)h->nrb; h->split.parts = 1; h->split.edge[0] = 0; h->split.edge[1] = (int32_t)h->ncb;
    if (h->nt > 0) {
        std::vector<double> tile_cost((size_t)h->ncb, 0.0);
        for (int64_t c = 0; c < h->ncb; ++c) {
            double sum = 0.0;
            for (int64_t rb = 0; rb < h->nrb; ++rb) sum += (double)tile_rslots_h[(size_t)(rb * h->ncb + c)];
            tile_cost[(size_t)c] = sum / (double)h->nrb / (16.0 * 64.0) + 2.0;
        }
        RES_TRY(oriana_row_pass_plan_cus(&cm, K, tile_cost.data(), h->cus, &h->split));
        int32_t *d_longest = nullptr;
        RES_TRY(tmp_alloc((void **)&d_longest, sizeof(int32_t) * h->nt));
        hipLaunchKernelGGL(k_tile_longest, dim3((unsigned)((h->nt + 255) / 256)), dim3(256), 0, s, d_longest, h->cslice, h->nt);
        ORIANA_LAUNCH_CHECK();
        std::vector<int32_t> longest((size_t)h->nt);
        ORIANA_HIP_CHECK(hipMemcpyAsync(longest.data(), d_longest, sizeof(int32_t) * h->nt, hipMemcpyDeviceToHost, s));
        ORIANA_HIP_CHECK(hipStreamSynchronize(s));
        const int64_t width = std::max<int64_t>(oriana_col_block_tiles(K), 1);
        const int64_t cap = oriana_plan_col_work_capacity(h->nrb, h->ncb, width);
        std::vector<int32_t> items((size_t)cap * 3);
        if (h->cslots > 0) {
            RES_TRY(oriana_plan_col_work(longest.data(), h->nrb, h->ncb, width, h->cus, 0, 1, 0, items.data(), cap, &h->n_col_work));
            RES_TRY(dev_alloc(h, &h->col_work, (size_t)std::max<int64_t>(h->n_col_work * 3, 1), false, s));
            ORIANA_HIP_CHECK(hipMemcpyAsync(h->col_work, items.data(), sizeof(int32_t) * h->n_col_work * 3, hipMemcpyHostToDevice, s));
            ORIANA_HIP_CHECK(hipStreamSynchronize(s));
            // the dual column pass of the sparse nests (two factor images per tile: one column tile per item)
            if (width == 1) { h->col_work1 = h->col_work; h->n_col_work1 = h->n_col_work; }
            else {
                const int64_t cap1 = oriana_plan_col_work_capacity(h->nrb, h->ncb, 1);
                std::vector<int32_t> items1((size_t)cap1 * 3);
                RES_TRY(oriana_plan_col_work(longest.data(), h->nrb, h->ncb, 1, h->cus, 0, 1, 0, items1.data(), cap1, &h->n_col_work1));
                RES_TRY(dev_alloc(h, &h->col_work1, (size_t)std::max<int64_t>(h->n_col_work1 * 3, 1), false, s));
                ORIANA_HIP_CHECK(hipMemcpyAsync(h->col_work1, items1.data(), sizeof(int32_t) * h->n_col_work1 * 3, hipMemcpyHostToDevice, s));
                ORIANA_HIP_CHECK(hipStreamSynchronize(s));
            }
        }
    }
    h->nslab = h->split.parts;
    if (gd > 0) RES_TRY(oriana_plan_dense_splits(n, gd, h->cus, &h->dn_gene_splits, &h->dn_cell_splits));
    // ---- 5. workspace of a call
    const int64_t Kp = h->Kp;
    RES_TRY(dev_alloc(h, &h->FU, (size_t)(std::max<int64_t>(n, 1) * Kp), true, s));
    RES_TRY(dev_alloc(h, &h->FV, (size_t)(std::max<int64_t>(m, 1) * Kp), true, s));
    // (slabs 1.. of a last-round split hold the rows of the split row blocks only)
    RES_TRY(dev_alloc(h, &h->R, (size_t)((std::max<int64_t>(n, 1) + (h->nslab - 1) * (n - (int64_t)h->split.nfull * TILE)) * Kp), true, s));
    RES_TRY(dev_alloc(h, &h->C, (size_t)(std::max<int64_t>(m, 1) * Kp), true, s));
    RES_TRY(dev_alloc(h, &h->s_cs, (size_t)std::max<int64_t>(h->cslots, 1), true, s));      // (padding slots must stay 0)
    RES_TRY(dev_alloc(h, &h->prep, (size_t)(oriana_prep_scratch_bytes() / 4), true, s));
    if (gd > 0) {
        RES_TRY(dev_alloc(h, &h->dn_S, (size_t)(nct * ngt * 1024), true, s));
        RES_TRY(dev_alloc(h, &h->dn_flag, (size_t)std::max<int64_t>(nct * ngt, 1), true, s));
        RES_TRY(dev_alloc(h, &h->dn_imgV, (size_t)(ngt * oriana_dense_image_pieces(K, 0) * 4), false, s));
        RES_TRY(dev_alloc(h, &h->dn_imgU, (size_t)(std::max<int64_t>((n + 31) / 32, 1) * oriana_dense_image_pieces(K, 1) * 4), false, s));
    }
    ORIANA_HIP_CHECK(hipStreamSynchronize(s));
    return 0;
}

int resident_create(oriana_resident **out, ChunkSource &src, int64_t n, int64_t m, int64_t K, double dense_density, void *stream) {
    if (!out) return ORIANA_EINVAL;
    *out = nullptr;
    if (n <= 0 || m <= 0 || K <= 0 || m > 0x7fffffffLL) return ORIANA_EINVAL;
    const int64_t Kp = oriana_kpad(K);
    if (Kp == 0) return ORIANA_EKRANGE;
    oriana_resident *h = new (std::nothrow) oriana_resident();
    if (!h) return ORIANA_EINVAL;
    h->n = n; h->m = m; h->K = K; h->Kp = Kp;
    src.m = m;
    const int rc = resident_build(h, src, dense_density, (hipStream_t)stream);
    if (rc) { resident_free(h); return rc; }
    *out = h;
    return 0;
}

}  // namespace

extern "C" int oriana_counts_create_dense_f32(oriana_resident **out, const float *X, int64_t n, int64_t m, int64_t ldx, int64_t K,
                                              double dense_density, void *stream) {
    if (!X || ldx < m) return ORIANA_EINVAL;
    ChunkSource src;
    src.X = X; src.ldx = ldx;
    return resident_create(out, src, n, m, K, dense_density, stream);
}

extern "C" int oriana_counts_create_csr(oriana_resident **out, const int64_t *indptr, const int32_t *indices, const float *data,
                                        int64_t n, int64_t m, int64_t K, double dense_density, void *stream) {
    if (!indptr || (n > 0 && indptr[n] > indptr[0] && (!indices || !data))) return ORIANA_EINVAL;
    for (int64_t r = 0; r < n; ++r)
        if (indptr[r + 1] < indptr[r]) return ORIANA_EINVAL;          // (a non-monotone indptr would send the scatter kernel out of bounds)
    for (int64_t e = indptr[0]; n > 0 && e < indptr[n]; ++e)
        if (indices[e] < 0 || indices[e] >= m) return ORIANA_EINVAL;
    ChunkSource src;
    src.indptr = indptr; src.indices = indices; src.data = data;
    return resident_create(out, src, n, m, K, dense_density, stream);
}

extern "C" int oriana_counts_destroy(oriana_resident *h) {
    if (!h) return 0;
    (void)hipDeviceSynchronize();
    resident_free(h);
    return 0;
}

extern "C" int oriana_counts_info(const oriana_resident *h, int64_t *info, int64_t len) {
    // {n, m, K, Kp, non-zeros, dense genes, row-side slots, column-side slots, resident bytes, column work items, row-split
    //  parts, first split row block, CUs planned for}
    if (!h || !info || len < 13) return ORIANA_EINVAL;
    const int64_t v[13] = {h->n, h->m, h->K, h->Kp, h->nnz, h->gd, h->rslots, h->cslots, h->bytes, h->n_col_work, h->split.parts,
                           h->split.nfull, h->cus};
    memcpy(info, v, sizeof(v));
    return 0;
}

// The pCMF nest on the resident layout (sliced or hybrid): what engine.zq_gap runs for the model classes.
extern "C" int oriana_zq_gap_resident(oriana_resident *h, float *Z_i, float *Z_j, const float *log_U_hat, const float *log_V_hat,
                                      void *stream) {
    if (!h || !Z_i || !Z_j || !log_U_hat || !log_V_hat) return ORIANA_EINVAL;
    const int64_t n = h->n, m = h->m, K = h->K, Kp = h->Kp, gd = h->gd;
    const float *den_min = reinterpret_cast<const float *>(reinterpret_cast<const char *>(h->prep) + oriana_prep_den_threshold_offset());
    oriana_clear_list cl;
    memset(&cl, 0, sizeof(cl));
    int e = 0;
    cl.ptr[e] = Z_i; cl.bytes[e++] = (int64_t)sizeof(float) * n * K;
    cl.ptr[e] = Z_j; cl.bytes[e++] = (int64_t)sizeof(float) * m * K;
    cl.ptr[e] = h->C; cl.bytes[e++] = (int64_t)sizeof(float) * m * Kp;
    cl.ptr[e] = h->tile_flag; cl.bytes[e++] = (int64_t)sizeof(int32_t) * std::max<int64_t>(h->nt, 1);
    if (h->ms == 0) { cl.ptr[e] = h->R; cl.bytes[e++] = (int64_t)sizeof(float) * (n + (h->nslab - 1) * (n - (int64_t)h->split.nfull * TILE)) * Kp; }
    RES_TRY(oriana_factor_prep_pair_clear(h->FU, h->FV, log_U_hat, log_V_hat, nullptr, nullptr, h->col_perm, n, m, K, h->prep, &cl, stream));
    float *FVs = h->FV + gd * Kp, *Cs = h->C + gd * Kp;
    if (h->ms > 0)
        RES_TRY(oriana_row_pass_general(&h->cm, h->FU, FVs, nullptr, nullptr, h->R, h->s_cs, nullptr, nullptr, h->tile_flag, K, &h->split,
                                        den_min, stream));
    if (gd > 0) {
        RES_TRY(oriana_dense_images(h->dn_imgV, h->FV, gd, K, 0, stream));
        int64_t tail_nfull = 0, tail_parts = 1;
        if (h->split.nfull > 0 && h->split.parts > 1 && h->dn_gene_splits == 1 && h->split.parts <= gd / 32) {
            tail_nfull = h->split.nfull; tail_parts = h->split.parts;
        }
        RES_TRY(oriana_dense_row_pass_tail(&h->dn, h->FU, h->dn_imgV, h->R, h->dn_S, h->dn_flag, K, h->dn_gene_splits, tail_nfull, tail_parts,
                                           den_min, stream));
    }
    if (h->ms > 0)
        RES_TRY(oriana_fixup(&h->cm, h->tile_flag, h->s_cs, nullptr, nullptr, log_U_hat, log_V_hat, nullptr, nullptr, nullptr, nullptr, Z_i, Z_j,
                             nullptr, K, 0, stream));
    if (gd > 0)
        RES_TRY(oriana_dense_fixup_variant(&h->dn, h->dn_flag, h->dn_S, log_U_hat, log_V_hat, nullptr, h->col_perm, Z_i, Z_j, nullptr, nullptr,
                                           nullptr, nullptr, K, 0, stream));
    RES_TRY(oriana_finalize_slabs_from(Z_i, h->FU, h->R, h->nslab, (int64_t)h->split.nfull * TILE, nullptr, n, K, stream));
    if (h->ms > 0) RES_TRY(oriana_col_pass(&h->cm, h->s_cs, h->FU, Cs, K, h->col_work, h->n_col_work, stream));
    if (gd > 0) {
        RES_TRY(oriana_dense_images(h->dn_imgU, h->FU, n, K, 1, stream));
        RES_TRY(oriana_dense_col_pass(&h->dn, h->dn_imgU, h->dn_S, h->C, K, h->dn_cell_splits, stream));
    }
    return oriana_finalize(Z_j, h->FV, h->C, nullptr, h->col_perm, m, K, 1, stream);
}

// [r6] The ZI / sparse nests WITHOUT per-entry weights -- D_hat == 1 at every stored entry, which is what every D_hat the
// reference's own models hand to these nests looks like (zigap.py:135 sets p_d = 1 - 1e-10 at the non-zero counts, bernoulli.py:45
// casts it to float32 = 1) and what the caller declares with oriana_counts_declare_unit_dropout -- as engine.zq runs them for the
// model classes: sliced or hybrid layout; Kp <= 64: the sparse nests' S_hat-weighted row sums out of the fused two-image row
// pass and both per-gene sums out of one dual column pass; above: den-only row pass + second row product + two column passes.
// D_hat itself is read for the D_hat[i, k] weights of zigap.py:94 only (dq).  Zlog may be NULL (the reference's caller never
// reads its third output, zigap.py:105-112): the log sums are then skipped.
static int zq_unit_resident(oriana_resident *h, float *Zi, float *Zj, float *Zlog, const float *log_U_hat, const float *log_V_hat,
                            const float *S_tilde, const float *S_hat, const float *D_hat, int quirk, void *stream) {
    const int64_t n = h->n, m = h->m, K = h->K, Kp = h->Kp, gd = h->gd;
    hipStream_t s = (hipStream_t)stream;
    const bool sparse = S_hat != nullptr, have_sliced = h->ms > 0 || gd == 0;
    const size_t rs1 = (size_t)std::max<int64_t>(h->rslots, 1);
    // (every buffer of a group is tested on its own: a failed allocation must not leave a later call with a NULL sibling)
    if (sparse && !h->F2) RES_TRY(dev_alloc(h, &h->F2, (size_t)(m * Kp), true, s));
    if (Zlog && !h->G2) RES_TRY(dev_alloc(h, &h->G2, (size_t)(n * Kp), true, s));
    if (Zlog && !h->C2) RES_TRY(dev_alloc(h, &h->C2, (size_t)(m * Kp), true, s));
    if (quirk && !h->dq) RES_TRY(dev_alloc(h, &h->dq, (size_t)(n * K), true, s));
    if (quirk && !h->GQ) RES_TRY(dev_alloc(h, &h->GQ, (size_t)(n * Kp), true, s));
    const float *den_min = reinterpret_cast<const float *>(reinterpret_cast<const char *>(h->prep) + oriana_prep_den_threshold_offset());
    const int64_t r_rows = n + (h->nslab - 1) * (n - (int64_t)h->split.nfull * TILE);
    oriana_clear_list cl;
    memset(&cl, 0, sizeof(cl));
    int e = 0;
    cl.ptr[e] = Zi; cl.bytes[e++] = (int64_t)sizeof(float) * n * K;
    cl.ptr[e] = Zj; cl.bytes[e++] = (int64_t)sizeof(float) * m * K;
    cl.ptr[e] = h->C; cl.bytes[e++] = (int64_t)sizeof(float) * m * Kp;
    cl.ptr[e] = h->tile_flag; cl.bytes[e++] = (int64_t)sizeof(int32_t) * std::max<int64_t>(h->nt, 1);
    if (Zlog) { cl.ptr[e] = Zlog; cl.bytes[e++] = (int64_t)sizeof(float) * m * K; cl.ptr[e] = h->C2; cl.bytes[e++] = (int64_t)sizeof(float) * m * Kp; }
    if (!have_sliced) { cl.ptr[e] = h->R; cl.bytes[e++] = (int64_t)sizeof(float) * r_rows * Kp; }
    RES_TRY(oriana_factor_prep_pair_clear(h->FU, h->FV, log_U_hat, log_V_hat, S_tilde, nullptr, h->col_perm, n, m, K, h->prep, &cl, stream));
    float *dq = nullptr;
    if (quirk) { dq = h->dq; RES_TRY(oriana_take_cols_f32(dq, D_hat, n, m, K, stream)); }
    if (sparse) RES_TRY(oriana_scale_factor(h->F2, h->FV, S_hat, h->col_perm, m, K, 0, stream));
    const int64_t goff = gd * Kp;
    const int variant = (sparse ? 1 : 0) | (dq ? 4 : 0);
    int64_t nslab = 1;
    if (have_sliced) {
        bool fused = false;
        if (sparse) {
            const int rc = oriana_row_pass_general(&h->cm, h->FU, h->FV + goff, h->F2 + goff, nullptr, h->R, h->s_cs, nullptr, nullptr,
                                                   h->tile_flag, K, &h->split, den_min, stream);
            if (rc != 0 && rc != ORIANA_EKRANGE) return rc;
            fused = rc == 0;
        }
        float *s_rs = nullptr;
        if (!fused) {
            if (sparse) { if (!h->s_rs) RES_TRY(dev_alloc(h, &h->s_rs, rs1, true, s)); s_rs = h->s_rs; }
            RES_TRY(oriana_row_pass_general(&h->cm, h->FU, h->FV + goff, nullptr, nullptr, h->R, h->s_cs, nullptr, s_rs, h->tile_flag, K,
                                            &h->split, den_min, stream));
        }
        if (fused || !sparse) nslab = h->nslab;              // (the second row product of the unfused sparse form writes one slab)
        RES_TRY(oriana_fixup(&h->cm, h->tile_flag, h->s_cs, nullptr, s_rs, log_U_hat, log_V_hat, S_tilde, S_hat, nullptr, dq, Zi, Zj, Zlog, K,
                             variant, stream));
        if (sparse && !fused) RES_TRY(oriana_row_spmm(&h->cm, s_rs, nullptr, h->F2 + goff, h->R, K, stream));
    }
    if (gd > 0) {
        RES_TRY(oriana_dense_images2(h->dn_imgV, h->FV, sparse ? h->F2 : nullptr, gd, K, 0, stream));
        int64_t tail_nfull = 0, tail_parts = 1;
        if (h->split.nfull > 0 && 1 < h->split.parts && h->split.parts == nslab && h->dn_gene_splits == 1 && h->split.parts <= gd / 32) {
            tail_nfull = h->split.nfull; tail_parts = h->split.parts;
        }
        RES_TRY(oriana_dense_row_pass_tail(&h->dn, h->FU, h->dn_imgV, h->R, h->dn_S, h->dn_flag, K, h->dn_gene_splits, tail_nfull, tail_parts,
                                           den_min, stream));
        RES_TRY(oriana_dense_fixup_variant(&h->dn, h->dn_flag, h->dn_S, log_U_hat, log_V_hat, nullptr, h->col_perm, Zi, Zj, Zlog, dq, S_tilde,
                                           S_hat, K, 0, stream));
    }
    RES_TRY(oriana_finalize_slabs_from(Zi, h->FU, h->R, nslab, (int64_t)h->split.nfull * TILE, nullptr, n, K, stream));
    double *center = reinterpret_cast<double *>(reinterpret_cast<char *>(h->prep) + oriana_prep_center_offset());
    if (Zlog) {
        RES_TRY(oriana_log_center(center, h->FU, log_U_hat, Zi, nullptr, n, K, stream));
        RES_TRY(oriana_scale_factor_centered(h->G2, h->FU, log_U_hat, center, nullptr, n, K, stream));
    }
    // ---- per-gene sums
    auto dense_cols = [&](const float *G, float *Cm) -> int {
        RES_TRY(oriana_dense_images(h->dn_imgU, G, n, K, 1, stream));
        return oriana_dense_col_pass(&h->dn, h->dn_imgU, h->dn_S, Cm, K, h->dn_cell_splits, stream);
    };
    auto dual = [&](bool *done) -> int {          // C += s FU and C2 += s G2 from one walk over the stream, where two images fit
        *done = false;
        if (!h->col_work1) return 0;
        const int rc = oriana_col_pass_dual(&h->cm, h->s_cs, h->FU, h->G2, h->C + goff, h->C2 + goff, K, h->col_work1, h->n_col_work1, stream);
        if (rc != 0 && rc != ORIANA_EKRANGE) return rc;
        *done = rc == 0;
        return 0;
    };
    const float *G = h->FU;
    if (dq) { RES_TRY(oriana_scale_factor(h->GQ, h->FU, dq, nullptr, n, K, 0, stream)); G = h->GQ; }
    bool dual_done = false;
    if (Zlog && !dq && have_sliced) RES_TRY(dual(&dual_done));
    if (!dual_done && have_sliced) RES_TRY(oriana_col_pass(&h->cm, h->s_cs, G, h->C + goff, K, h->col_work, h->n_col_work, stream));
    if (gd > 0) RES_TRY(dense_cols(G, h->C));
    RES_TRY(oriana_finalize(Zj, h->FV, h->C, nullptr, h->col_perm, m, K, 1, stream));
    if (Zlog) {
        if (dq) {                                 // the log sums use the plain column sums (zigap.py:95), Z_j the D_hat[i, k]-weighted ones
            ORIANA_HIP_CHECK(hipMemsetAsync(h->C, 0, sizeof(float) * m * Kp, s));
            if (have_sliced) {
                RES_TRY(dual(&dual_done));
                if (!dual_done) RES_TRY(oriana_col_pass(&h->cm, h->s_cs, h->FU, h->C + goff, K, h->col_work, h->n_col_work, stream));
            }
            if (gd > 0) RES_TRY(dense_cols(h->FU, h->C));
        }
        if (!dual_done && have_sliced) RES_TRY(oriana_col_pass(&h->cm, h->s_cs, h->G2, h->C2 + goff, K, h->col_work, h->n_col_work, stream));
        if (gd > 0) RES_TRY(dense_cols(h->G2, h->C2));
        RES_TRY(oriana_finalize_zlog(Zlog, h->FV, h->C2, h->C, log_V_hat, center, h->col_perm, m, K, stream));
    }
    return 0;
}

// The ZI / sparse nests with a GENERAL D_hat (any weight at the stored entries) on a resident SLICED layout: the kernel sequence
// of the stateless entries (stateless.hip: zq_dense) without the packing; D_hat is gathered at the stored entries on every call.
// The dense-gene kernels of a hybrid layout carry no per-entry weights: a hybrid handle serves these nests under the unit
// declaration (above) only.
static int zq_variant_resident(oriana_resident *h, float *Zi, float *Zj, float *Zlog, const float *log_U_hat, const float *log_V_hat,
                               const float *S_tilde, const float *S_hat, const float *D_hat, int quirk, void *stream) {
    if (!h || !Zi || !Zj || !log_U_hat || !log_V_hat) return ORIANA_EINVAL;
    if ((S_tilde == nullptr) != (S_hat == nullptr)) return ORIANA_EINVAL;
    const int64_t n = h->n, m = h->m, K = h->K, Kp = h->Kp;
    if (quirk && (!D_hat || K > m)) return ORIANA_EQUIRK;
    if (!D_hat || h->unit_dropout)
        return zq_unit_resident(h, Zi, Zj, Zlog, log_U_hat, log_V_hat, S_tilde, S_hat, D_hat, quirk, stream);
    if (h->gd > 0) return ORIANA_EUNIT;
    hipStream_t s = (hipStream_t)stream;
    const bool sparse = S_hat != nullptr, weighted = D_hat != nullptr;
    const size_t rs1 = (size_t)std::max<int64_t>(h->rslots, 1), cs1 = (size_t)std::max<int64_t>(h->cslots, 1);
    if (weighted && !h->w_nz) RES_TRY(dev_alloc(h, &h->w_nz, rs1, true, s));
    if (weighted && !h->sw_cs) RES_TRY(dev_alloc(h, &h->sw_cs, cs1, true, s));
    if (sparse && !h->s_rs) RES_TRY(dev_alloc(h, &h->s_rs, rs1, true, s));
    if (sparse && !h->F2) RES_TRY(dev_alloc(h, &h->F2, (size_t)(m * Kp), true, s));
    if ((Zlog || quirk) && !h->G2) RES_TRY(dev_alloc(h, &h->G2, (size_t)(n * Kp), true, s));
    if (Zlog && !h->C2) RES_TRY(dev_alloc(h, &h->C2, (size_t)(m * Kp), true, s));
    if (quirk && !h->dq) RES_TRY(dev_alloc(h, &h->dq, (size_t)(n * K), true, s));
    const float *den_min = reinterpret_cast<const float *>(reinterpret_cast<const char *>(h->prep) + oriana_prep_den_threshold_offset());
    oriana_clear_list cl;
    memset(&cl, 0, sizeof(cl));
    int e = 0;
    cl.ptr[e] = Zi; cl.bytes[e++] = (int64_t)sizeof(float) * n * K;
    cl.ptr[e] = Zj; cl.bytes[e++] = (int64_t)sizeof(float) * m * K;
    cl.ptr[e] = h->C; cl.bytes[e++] = (int64_t)sizeof(float) * m * Kp;
    cl.ptr[e] = h->tile_flag; cl.bytes[e++] = (int64_t)sizeof(int32_t) * std::max<int64_t>(h->nt, 1);
    if (Zlog) { cl.ptr[e] = Zlog; cl.bytes[e++] = (int64_t)sizeof(float) * m * K; cl.ptr[e] = h->C2; cl.bytes[e++] = (int64_t)sizeof(float) * m * Kp; }
    RES_TRY(oriana_factor_prep_pair_clear(h->FU, h->FV, log_U_hat, log_V_hat, S_tilde, nullptr, h->col_perm, n, m, K, h->prep, &cl, stream));
    float *w_nz = weighted ? h->w_nz : nullptr, *sw_cs = weighted ? h->sw_cs : nullptr, *s_rs = sparse ? h->s_rs : nullptr;
    if (weighted && h->nt > 0) {
        hipLaunchKernelGGL(k_gather_nz, dim3((unsigned)h->nt), dim3(256), 0, s, h->cm, D_hat, m, w_nz);
        ORIANA_LAUNCH_CHECK();
    }
    float *dq = nullptr;
    if (quirk) { dq = h->dq; RES_TRY(oriana_take_cols_f32(dq, D_hat, n, m, K, stream)); }
    // (the whole-grid kernels take no last-round split: the plan's split applies to the plain pass of the two-lane kernels;
    //  with s_rs the pass leaves R alone, so one slab serves)
    oriana_row_split sp = h->split;
    RES_TRY(oriana_row_pass_general(&h->cm, h->FU, h->FV, nullptr, w_nz, h->R, h->s_cs, sw_cs, s_rs, h->tile_flag, K, &sp, den_min, stream));
    const int variant = (sparse ? 1 : 0) | (weighted ? 2 : 0) | (dq ? 4 : 0);
    RES_TRY(oriana_fixup(&h->cm, h->tile_flag, h->s_cs, sw_cs, s_rs, log_U_hat, log_V_hat, S_tilde, S_hat, w_nz, dq, Zi, Zj, Zlog, K, variant, stream));
    int64_t nslab = h->nslab;
    if (sparse) {
        RES_TRY(oriana_scale_factor(h->F2, h->FV, S_hat, h->col_perm, m, K, 0, stream));
        RES_TRY(oriana_row_spmm(&h->cm, s_rs, w_nz, h->F2, h->R, K, stream));
        nslab = 1;
    }
    RES_TRY(oriana_finalize_slabs_from(Zi, h->FU, h->R, nslab, (int64_t)h->split.nfull * TILE, nullptr, n, K, stream));
    const float *G = h->FU, *s_for_j = sw_cs ? sw_cs : h->s_cs;
    if (dq) { RES_TRY(oriana_scale_factor(h->G2, h->FU, dq, nullptr, n, K, 0, stream)); G = h->G2; s_for_j = h->s_cs; }
    RES_TRY(oriana_col_pass(&h->cm, s_for_j, G, h->C, K, h->col_work, h->n_col_work, stream));
    RES_TRY(oriana_finalize(Zj, h->FV, h->C, nullptr, h->col_perm, m, K, 1, stream));
    if (Zlog) {
        const float *s_log = sw_cs ? sw_cs : h->s_cs;
        if (dq) {
            ORIANA_HIP_CHECK(hipMemsetAsync(h->C, 0, sizeof(float) * m * Kp, s));
            RES_TRY(oriana_col_pass(&h->cm, s_log, h->FU, h->C, K, h->col_work, h->n_col_work, stream));
        }
        double *center = reinterpret_cast<double *>(reinterpret_cast<char *>(h->prep) + oriana_prep_center_offset());
        RES_TRY(oriana_log_center(center, h->FU, log_U_hat, Zi, nullptr, n, K, stream));
        RES_TRY(oriana_scale_factor_centered(h->G2, h->FU, log_U_hat, center, nullptr, n, K, stream));
        RES_TRY(oriana_col_pass(&h->cm, s_log, h->G2, h->C2, K, h->col_work, h->n_col_work, stream));
        RES_TRY(oriana_finalize_zlog(Zlog, h->FV, h->C2, h->C, log_V_hat, center, h->col_perm, m, K, stream));
    }
    return 0;
}

extern "C" int oriana_counts_declare_unit_dropout(oriana_resident *h, int on) {
    if (!h) return ORIANA_EINVAL;
    h->unit_dropout = on ? 1 : 0;
    return 0;
}

extern "C" int oriana_zq_zigap_resident(oriana_resident *h, float *DZ_hat_i, float *DZ_hat_j, float *DZ_exp_logsum_hat,
                                        const float *log_U_hat, const float *log_V_hat, const float *D_hat, int reference_quirks,
                                        void *stream) {
    if (!D_hat) return ORIANA_EINVAL;
    return zq_variant_resident(h, DZ_hat_i, DZ_hat_j, DZ_exp_logsum_hat, log_U_hat, log_V_hat, nullptr, nullptr, D_hat,
                               reference_quirks ? 1 : 0, stream);
}

extern "C" int oriana_zq_sparse_gap_resident(oriana_resident *h, float *SZ_hat_i, float *Z_hat_j, float *Z_exp_logsum_hat,
                                             const float *log_U_hat, const float *log_V_hat, const float *S_tilde, const float *S_hat,
                                             void *stream) {
    if (!S_tilde || !S_hat || !Z_exp_logsum_hat) return ORIANA_EINVAL;
    return zq_variant_resident(h, SZ_hat_i, Z_hat_j, Z_exp_logsum_hat, log_U_hat, log_V_hat, S_tilde, S_hat, nullptr, 0, stream);
}

extern "C" int oriana_zq_sparse_zigap_resident(oriana_resident *h, float *DSZ_hat, float *DZ_hat, float *DZ_exp_logsum_hat,
                                               const float *log_U_hat, const float *log_V_hat, const float *S_tilde, const float *S_hat,
                                               const float *D_hat, void *stream) {
    if (!S_tilde || !S_hat || !D_hat || !DZ_exp_logsum_hat) return ORIANA_EINVAL;
    return zq_variant_resident(h, DSZ_hat, DZ_hat, DZ_exp_logsum_hat, log_U_hat, log_V_hat, S_tilde, S_hat, D_hat, 0, stream);
}

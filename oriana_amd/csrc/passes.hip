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
// touched: X lives in HBM as 256 x 256 tiles of non-zero records (pack.hip).
//
// Mapping (wave64): a group of G lanes owns one row (row pass) or one column (column pass) for
// the whole kernel and keeps its K-vector and its accumulator in registers, 4*T4 floats per lane
// (Kp = 4*G*T4).  The other side's K-vectors are staged through LDS, one 256-row tile at a time,
// and read with ds_read_b128.  No MFMA: the work is a sampled dot product per non-zero plus two
// scaled vector adds, not a dense contraction.
#include "common.h"

namespace oriana {

// ------------------------------------------------------------------------------------------
// factor preparation
// ------------------------------------------------------------------------------------------
// one wave per row; K <= 64 * PER lanes-slots
__global__ __launch_bounds__(256) void k_factor_prep(float *__restrict__ F, float *__restrict__ mu_out,
                                                     const float *__restrict__ logF, const float *__restrict__ mask,
                                                     int64_t r, int K, int Kp) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= r) return;
    const float *l = logF + row * K;
    const float *mk = mask ? mask + row * K : nullptr;
    float mx = -INFINITY;
    bool bad = false;
    for (int k = lane; k < K; k += 64) {
        const float v = l[k];
        const bool on = mk ? (mk[k] != 0.0f) : true;
        if (on) { if (v != v) bad = true; mx = fmaxf(mx, v); }
    }
    mx = wave_max(mx);
    bad = __any(bad);
    // rows the shifted form cannot represent faithfully get an all-zero factor row: every entry
    // touching them fails the den >= DEN_MIN test and is evaluated by the exact slow path.
    const bool flagged = bad || !(fabsf(mx) < SHIFT_MAX);
    for (int k = lane; k < Kp; k += 64) {
        float out = 0.0f;
        if (!flagged && k < K) {
            const float mv = mk ? mk[k] : 1.0f;
            if (mv != 0.0f) out = (float)exp((double)l[k] - (double)mx) * mv;
        }
        F[row * Kp + k] = out;
    }
    if (mu_out && lane == 0) mu_out[row] = flagged ? NAN : mx;
}

// ------------------------------------------------------------------------------------------
// LDS geometry shared by the three tile kernels
// ------------------------------------------------------------------------------------------
constexpr int lds_stride_floats(int KP) { return (KP + 63) / 64 * 64; }     // rows are 256-B aligned
constexpr int LDS_BUDGET = 160 * 1024;
// smallest power-of-two split of the 256 staged rows such that `images` LDS images fit
constexpr int pick_nsub(int KP, int images) {
    int nsub = 1;
    while ((TILE / nsub) * lds_stride_floats(KP) * 4 * images > LDS_BUDGET) nsub *= 2;
    return nsub;
}

template <int G>
__device__ __forceinline__ int lds_rot(int lane) {
    // ds_read_b128 services lanes {0-3,12-15,20-27} / {4-11,16-19,28-31} (+32) together; groups
    // that are serviced together start at different 64-byte quarters of the 256-byte bank row.
    if (G == 4) return ((lane >> 2) & 7) >> 1;
    if (G == 8) return ((lane >> 3) & 3) >> 1;
    return 0;
}

__device__ __forceinline__ int quad_count(bool v) {
    uint32_t c = v ? 1u : 0u;
    c += dpp_u32<0xB1>(c);
    c += dpp_u32<0x4E>(c);
    return (int)c;
}

// stage `rows` factor rows starting at global row j0 (bounded by jmax) into an LDS image
template <int KP4, int STRIDE4>
__device__ __forceinline__ void stage_rows(float4 *img, const float *__restrict__ F, int64_t j0, int64_t jmax,
                                           int rows, int tid) {
    for (int idx = tid; idx < rows * KP4; idx += 1024) {
        const int jr = idx / KP4, c4 = idx - jr * KP4;
        const int64_t j = j0 + jr;
        if (j < jmax) img[jr * STRIDE4 + c4] = reinterpret_cast<const float4 *>(F)[j * KP4 + c4];
    }
}

// ------------------------------------------------------------------------------------------
// row pass:  s = x / <FU_i, FVden_j>,   R_i += w s FVacc_j
// ------------------------------------------------------------------------------------------
template <int G, int T4, bool SEPACC, bool HASW>
__global__ __launch_bounds__(1024) void k_row_pass(oriana_counts cm, const float *__restrict__ FU,
                                                   const float *__restrict__ FVden, const float *__restrict__ FVacc,
                                                   const float *__restrict__ w_nz, float *__restrict__ R,
                                                   float *__restrict__ s_col, float *__restrict__ sw_col,
                                                   float *__restrict__ s_row, int32_t *__restrict__ tile_flag) {
    constexpr int KP = 4 * G * T4;
    constexpr int KP4 = KP / 4;                         // float4 per factor row
    constexpr int STRIDE4 = lds_stride_floats(KP) / 4;  // LDS row stride in float4
    constexpr int ROWS = 1024 / G;                      // rows owned by one workgroup
    constexpr int SPLIT = TILE / ROWS;                  // workgroups per 256-row block
    constexpr int NSUB = pick_nsub(KP, SEPACC ? 2 : 1); // column sub-tiles per tile (LDS budget)
    constexpr int CT = TILE / NSUB;
    extern __shared__ float4 lds[];                     // [CT][STRIDE4] (+ [CT][STRIDE4] when SEPACC)
    float4 *ldsA = lds + (SEPACC ? CT * STRIDE4 : 0);

    const int tid = threadIdx.x, lane = tid & 63;
    const int q = tid & (G - 1);
    const int grp = tid / G;
    const int64_t rb = blockIdx.x / SPLIT;
    const int rl = (blockIdx.x % SPLIT) * ROWS + grp;      // row inside the 256-row block
    const int64_t row = rb * TILE + rl;
    const int rot = lds_rot<G>(lane);

    int choff[T4];                              // float4 offset of the chunk visited at step t
    #pragma unroll
    for (int t = 0; t < T4; ++t) choff[t] = ((t + rot) % T4) * G + q;

    float4 fu[T4], acc[T4];
    #pragma unroll
    for (int t = 0; t < T4; ++t) {
        acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        fu[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (row < cm.n) {
        #pragma unroll
        for (int t = 0; t < T4; ++t) fu[t] = reinterpret_cast<const float4 *>(FU)[row * KP4 + choff[t]];
    }
    for (int64_t cb = 0; cb < cm.ncb; ++cb) {
        const int64_t t = rb * cm.ncb + cb;
        const int64_t base = cm.tile_off[t];
        uint32_t p = cm.row_ptr[t * (TILE + 1) + rl];              // cursor over this row's records
        const uint32_t re = cm.row_ptr[t * (TILE + 1) + rl + 1];
        for (int csub = 0; csub < NSUB; ++csub) {
            __syncthreads();                    // everybody is done with the previous image
            stage_rows<KP4, STRIDE4>(lds, FVden, cb * TILE + csub * CT, cm.m, CT, tid);
            if (SEPACC) stage_rows<KP4, STRIDE4>(ldsA, FVacc, cb * TILE + csub * CT, cm.m, CT, tid);
            __syncthreads();
            const uint32_t lim = (uint32_t)(csub + 1) * CT;        // records with col < lim are stageable
            while (p < re) {
                // each quad fetches four consecutive records of its row (all quads of a group
                // fetch the same four), then walks them with quad broadcasts
                const uint32_t mine = p + (lane & 3);
                unsigned long long raw = 0ull;
                float wv = 1.0f;
                if (mine < re) {
                    raw = reinterpret_cast<const unsigned long long *>(cm.rowrec)[base + mine];
                    if (HASW) wv = w_nz[base + mine];
                }
                const uint32_t rx = (uint32_t)raw, rm = (uint32_t)(raw >> 32);
                // records are column-sorted: the ones inside this sub-tile form a prefix
                const int cnt = quad_count(mine < re && (NSUB == 1 || ((rm >> 16) & 0xFFu) < lim));
                float sbuf = 0.f;   // s of the record this lane fetched (for the row-major store)
                #pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (u < cnt) {
                        uint32_t bx = 0, bm = 0; float bw = 1.f;
                        if (u == 0) { bx = quad_bcast_u32<0>(rx); bm = quad_bcast_u32<0>(rm); bw = quad_bcast_f32<0>(wv); }
                        if (u == 1) { bx = quad_bcast_u32<1>(rx); bm = quad_bcast_u32<1>(rm); bw = quad_bcast_f32<1>(wv); }
                        if (u == 2) { bx = quad_bcast_u32<2>(rx); bm = quad_bcast_u32<2>(rm); bw = quad_bcast_f32<2>(wv); }
                        if (u == 3) { bx = quad_bcast_u32<3>(rx); bm = quad_bcast_u32<3>(rm); bw = quad_bcast_f32<3>(wv); }
                        const float x = __uint_as_float(bx);
                        const uint32_t cpos = bm & 0xFFFFu;
                        const int col = (int)((bm >> 16) & 0xFFu) - csub * CT;
                        const float4 *vrow = lds + col * STRIDE4;
                        float4 v[T4];
                        float den = 0.f;
                        #pragma unroll
                        for (int tt = 0; tt < T4; ++tt) {
                            v[tt] = vrow[choff[tt]];
                            den = fmaf(fu[tt].x, v[tt].x, den);
                            den = fmaf(fu[tt].y, v[tt].y, den);
                            den = fmaf(fu[tt].z, v[tt].z, den);
                            den = fmaf(fu[tt].w, v[tt].w, den);
                        }
                        den = group_sum<G>(den);
                        const bool ok = den >= DEN_MIN;          // false for 0, tiny and NaN
                        float s = ok ? x * __builtin_amdgcn_rcpf(den) : 0.f;
                        // one Newton step on the quotient: s <- s + (x - s*den) / den  (keeps s within 1 ulp)
                        if (ok) s = fmaf(fmaf(-s, den, x), __builtin_amdgcn_rcpf(den), s);
                        const float sw = HASW ? s * bw : s;
                        if (SEPACC) {
                            const float4 *arow = ldsA + col * STRIDE4;
                            #pragma unroll
                            for (int tt = 0; tt < T4; ++tt) v[tt] = arow[choff[tt]];
                        }
                        #pragma unroll
                        for (int tt = 0; tt < T4; ++tt) {
                            acc[tt].x = fmaf(sw, v[tt].x, acc[tt].x);
                            acc[tt].y = fmaf(sw, v[tt].y, acc[tt].y);
                            acc[tt].z = fmaf(sw, v[tt].z, acc[tt].z);
                            acc[tt].w = fmaf(sw, v[tt].w, acc[tt].w);
                        }
                        const float sout = ok ? s : NAN;          // NaN = "evaluate me exactly" sentinel
                        if (q == 0) {
                            s_col[base + cpos] = sout;
                            if (sw_col) sw_col[base + cpos] = ok ? sw : NAN;
                        }
                        if ((lane & 3) == u) sbuf = sout;
                        if (!ok && q == 0) tile_flag[t] = 1;
                    }
                }
                if (s_row && q < 4 && (lane & 3) < cnt) s_row[base + mine] = sbuf;
                p += cnt;
                if (cnt < 4) break;             // end of the row or of the sub-tile
            }
        }
    }
    if (row < cm.n) {
        #pragma unroll
        for (int t = 0; t < T4; ++t) reinterpret_cast<float4 *>(R)[row * KP4 + choff[t]] = acc[t];
    }
}

// ------------------------------------------------------------------------------------------
// row SpMM with given s (row-major):  R_i = sum_j w s FV_j      (sparse models: S_hat-weighted sums)
// ------------------------------------------------------------------------------------------
template <int G, int T4, bool HASW>
__global__ __launch_bounds__(1024) void k_row_spmm(oriana_counts cm, const float *__restrict__ s_row,
                                                   const float *__restrict__ w_nz, const float *__restrict__ FV,
                                                   float *__restrict__ R) {
    constexpr int KP = 4 * G * T4;
    constexpr int KP4 = KP / 4;
    constexpr int STRIDE4 = lds_stride_floats(KP) / 4;
    constexpr int ROWS = 1024 / G;
    constexpr int SPLIT = TILE / ROWS;
    constexpr int NSUB = pick_nsub(KP, 1);
    constexpr int CT = TILE / NSUB;
    extern __shared__ float4 lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int q = tid & (G - 1);
    const int grp = tid / G;
    const int64_t rb = blockIdx.x / SPLIT;
    const int rl = (blockIdx.x % SPLIT) * ROWS + grp;
    const int64_t row = rb * TILE + rl;
    const int rot = lds_rot<G>(lane);
    int choff[T4];
    #pragma unroll
    for (int t = 0; t < T4; ++t) choff[t] = ((t + rot) % T4) * G + q;
    float4 acc[T4];
    #pragma unroll
    for (int t = 0; t < T4; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);

    for (int64_t cb = 0; cb < cm.ncb; ++cb) {
        const int64_t t = rb * cm.ncb + cb;
        const int64_t base = cm.tile_off[t];
        uint32_t p = cm.row_ptr[t * (TILE + 1) + rl];
        const uint32_t re = cm.row_ptr[t * (TILE + 1) + rl + 1];
        for (int csub = 0; csub < NSUB; ++csub) {
            __syncthreads();
            stage_rows<KP4, STRIDE4>(lds, FV, cb * TILE + csub * CT, cm.m, CT, tid);
            __syncthreads();
            const uint32_t lim = (uint32_t)(csub + 1) * CT;
            while (p < re) {
                const uint32_t mine = p + (lane & 3);
                uint32_t rm = 0; float sv = 0.f;
                if (mine < re) {
                    rm = (uint32_t)(reinterpret_cast<const unsigned long long *>(cm.rowrec)[base + mine] >> 32);
                    sv = s_row[base + mine];
                    if (HASW) sv *= w_nz[base + mine];
                }
                const int cnt = quad_count(mine < re && (NSUB == 1 || ((rm >> 16) & 0xFFu) < lim));
                #pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (u < cnt) {
                        uint32_t bm = 0; float s = 0.f;
                        if (u == 0) { bm = quad_bcast_u32<0>(rm); s = quad_bcast_f32<0>(sv); }
                        if (u == 1) { bm = quad_bcast_u32<1>(rm); s = quad_bcast_f32<1>(sv); }
                        if (u == 2) { bm = quad_bcast_u32<2>(rm); s = quad_bcast_f32<2>(sv); }
                        if (u == 3) { bm = quad_bcast_u32<3>(rm); s = quad_bcast_f32<3>(sv); }
                        const int col = (int)((bm >> 16) & 0xFFu) - csub * CT;
                        const float4 *vrow = lds + col * STRIDE4;
                        #pragma unroll
                        for (int tt = 0; tt < T4; ++tt) {
                            const float4 v = vrow[choff[tt]];
                            acc[tt].x = fmaf(s, v.x, acc[tt].x);
                            acc[tt].y = fmaf(s, v.y, acc[tt].y);
                            acc[tt].z = fmaf(s, v.z, acc[tt].z);
                            acc[tt].w = fmaf(s, v.w, acc[tt].w);
                        }
                    }
                }
                p += cnt;
                if (cnt < 4) break;
            }
        }
    }
    if (row < cm.n) {
        #pragma unroll
        for (int t = 0; t < T4; ++t) reinterpret_cast<float4 *>(R)[row * KP4 + choff[t]] = acc[t];
    }
}

// ------------------------------------------------------------------------------------------
// column pass:  C_j += sum_i s_ij G_i      (grid.y = row bands, combined with float atomics)
// ------------------------------------------------------------------------------------------
template <int G, int T4>
__global__ __launch_bounds__(1024) void k_col_pass(oriana_counts cm, const float *__restrict__ s_col,
                                                   const float *__restrict__ Gm, float *__restrict__ C,
                                                   int64_t rb_per_band) {
    constexpr int KP = 4 * G * T4;
    constexpr int KP4 = KP / 4;
    constexpr int STRIDE4 = lds_stride_floats(KP) / 4;
    constexpr int COLS = 1024 / G;
    constexpr int SPLIT = TILE / COLS;
    constexpr int NSUB = pick_nsub(KP, 1);
    constexpr int RT = TILE / NSUB;
    extern __shared__ float4 lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int q = tid & (G - 1);
    const int grp = tid / G;
    const int64_t cb = blockIdx.x / SPLIT;
    const int cl = (blockIdx.x % SPLIT) * COLS + grp;
    const int64_t col = cb * TILE + cl;
    const int rot = lds_rot<G>(lane);
    int choff[T4];
    #pragma unroll
    for (int t = 0; t < T4; ++t) choff[t] = ((t + rot) % T4) * G + q;
    float4 acc[T4];
    #pragma unroll
    for (int t = 0; t < T4; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);

    const int64_t rb0 = (int64_t)blockIdx.y * rb_per_band;
    const int64_t rb1 = (rb0 + rb_per_band < cm.nrb) ? rb0 + rb_per_band : cm.nrb;
    for (int64_t rb = rb0; rb < rb1; ++rb) {
        const int64_t t = rb * cm.ncb + cb;
        const int64_t base = cm.tile_off[t];
        uint32_t p = cm.col_ptr[t * (TILE + 1) + cl];
        const uint32_t ce = cm.col_ptr[t * (TILE + 1) + cl + 1];
        for (int rsub = 0; rsub < NSUB; ++rsub) {
            __syncthreads();
            stage_rows<KP4, STRIDE4>(lds, Gm, rb * TILE + rsub * RT, cm.n, RT, tid);
            __syncthreads();
            const uint32_t lim = (uint32_t)(rsub + 1) * RT;
            while (p < ce) {
                const uint32_t mine = p + (lane & 3);
                float sv = 0.f; uint32_t rv = 0;
                if (mine < ce) { sv = s_col[base + mine]; rv = cm.ridx[base + mine]; }
                const int cnt = quad_count(mine < ce && (NSUB == 1 || rv < lim));   // rows are sorted
                #pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (u < cnt) {
                        float s = 0.f; uint32_t r = 0;
                        if (u == 0) { s = quad_bcast_f32<0>(sv); r = quad_bcast_u32<0>(rv); }
                        if (u == 1) { s = quad_bcast_f32<1>(sv); r = quad_bcast_u32<1>(rv); }
                        if (u == 2) { s = quad_bcast_f32<2>(sv); r = quad_bcast_u32<2>(rv); }
                        if (u == 3) { s = quad_bcast_f32<3>(sv); r = quad_bcast_u32<3>(rv); }
                        const float4 *vrow = lds + ((int)r - rsub * RT) * STRIDE4;
                        #pragma unroll
                        for (int tt = 0; tt < T4; ++tt) {
                            const float4 v = vrow[choff[tt]];
                            acc[tt].x = fmaf(s, v.x, acc[tt].x);
                            acc[tt].y = fmaf(s, v.y, acc[tt].y);
                            acc[tt].z = fmaf(s, v.z, acc[tt].z);
                            acc[tt].w = fmaf(s, v.w, acc[tt].w);
                        }
                    }
                }
                p += cnt;
                if (cnt < 4) break;
            }
        }
    }
    if (col < cm.m) {
        float *dst = C + col * KP;
        #pragma unroll
        for (int t = 0; t < T4; ++t) {
            float *d = dst + choff[t] * 4;
            if (acc[t].x != 0.f) atomicAdd(d + 0, acc[t].x);
            if (acc[t].y != 0.f) atomicAdd(d + 1, acc[t].y);
            if (acc[t].z != 0.f) atomicAdd(d + 2, acc[t].z);
            if (acc[t].w != 0.f) atomicAdd(d + 3, acc[t].w);
        }
    }
}

// ------------------------------------------------------------------------------------------
// finalize:  Z = [Z +] F * R [* mul]     dense (r, K) out from padded (r, Kp) in
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_finalize(float *__restrict__ Z, const float *__restrict__ F,
                                                  const float *__restrict__ R, const float *__restrict__ mul,
                                                  int64_t r, int K, int Kp, int accumulate) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= r * K) return;
    const int64_t row = idx / K;
    const int k = (int)(idx - row * K);
    float v = F[row * Kp + k] * R[row * Kp + k];
    if (mul) v *= mul[idx];
    Z[idx] = accumulate ? Z[idx] + v : v;
}

// ------------------------------------------------------------------------------------------
// fix-up: exact reference arithmetic for the entries flagged with the NaN sentinel
// ------------------------------------------------------------------------------------------
// grid = tiles; block = 256 threads, thread r walks row r of the tile.
__global__ __launch_bounds__(256) void k_fixup(oriana_counts cm, const int32_t *__restrict__ tile_flag,
                                               float *__restrict__ s_col, float *__restrict__ sw_col,
                                               float *__restrict__ s_row, const float *__restrict__ logU,
                                               const float *__restrict__ logV, const float *__restrict__ S_tilde,
                                               const float *__restrict__ S_hat, const float *__restrict__ w_nz,
                                               const float *__restrict__ dq, float *__restrict__ Zi,
                                               float *__restrict__ Zj, float *__restrict__ Zlog, int K, int quirk) {
    const int64_t t = blockIdx.x;
    if (tile_flag[t] == 0) return;
    const int64_t rb = t / cm.ncb, cb = t - rb * cm.ncb;
    const int rl = threadIdx.x;
    const int64_t i = rb * TILE + rl;
    if (i >= cm.n) return;
    const int64_t base = cm.tile_off[t];
    const uint32_t rs = cm.row_ptr[t * (TILE + 1) + rl], re = cm.row_ptr[t * (TILE + 1) + rl + 1];
    const float *lu = logU + i * K;
    for (uint32_t p = rs; p < re; ++p) {
        const oriana_rowrec rec = cm.rowrec[base + p];
        const float s = s_col[base + rec.cpos];
        if (s == s) continue;                                   // not a sentinel
        const int64_t j = cb * TILE + rec.col;
        const float *lv = logV + j * K;
        const float *st = S_tilde ? S_tilde + j * K : nullptr;
        const float *sh = S_hat ? S_hat + j * K : nullptr;
        const float x = rec.x;
        const float w = w_nz ? w_nz[base + p] : 1.0f;
        // den = sum_k exp(lu + lv) [* S_tilde], float32, left to right (gap.py:74-76)
        float den = 0.f;
        for (int k = 0; k < K; ++k) {
            float e = expf(lu[k] + lv[k]);
            if (st) e *= st[k];
            den += e;
        }
        den = (den > 0.f) ? den : 1.0f;
        for (int k = 0; k < K; ++k) {
            const float ls = lu[k] + lv[k];
            float e = expf(ls);
            if (st) e *= st[k];
            const float expectation = (x * e) / den;            // gap.py:78
            if (Zi) {
                float wi = w;
                if (sh) wi = w_nz ? w * sh[k] : sh[k];          // sparse_zigap.py:114 / sparse_gap.py:95
                const float v = (w_nz || sh) ? wi * expectation : expectation;
                if (v != 0.f) atomicAdd(&Zi[i * K + k], v);
            }
            if (Zj) {
                float v = expectation;
                if (quirk && dq) v = dq[i * K + k] * expectation;   // zigap.py:94 (D_hat[i, k])
                else if (w_nz) v = w * expectation;                 // sparse_zigap.py:115
                if (v != 0.f) atomicAdd(&Zj[j * K + k], v);
            }
            if (Zlog) {
                const float v = (w_nz ? w * expectation : expectation) * ls;   // zigap.py:95
                if (v != 0.f) atomicAdd(&Zlog[j * K + k], v);
            }
        }
        s_col[base + rec.cpos] = 0.f;
        if (sw_col) sw_col[base + rec.cpos] = 0.f;
        if (s_row) s_row[base + p] = 0.f;
    }
}

// ------------------------------------------------------------------------------------------
// dispatch on K:  Kp = 4 * G * T4
// ------------------------------------------------------------------------------------------
struct KCfg { int G, T4; };
static inline bool pick_cfg(int64_t K, KCfg *c) {
    if (K <= 0) return false;
    if (K <= 16)  { *c = {4, 1}; return true; }
    if (K <= 32)  { *c = {4, 2}; return true; }
    if (K <= 48)  { *c = {4, 3}; return true; }
    if (K <= 64)  { *c = {4, 4}; return true; }
    if (K <= 80)  { *c = {4, 5}; return true; }
    if (K <= 96)  { *c = {4, 6}; return true; }
    if (K <= 112) { *c = {4, 7}; return true; }
    if (K <= 128) { *c = {8, 4}; return true; }
    if (K <= 160) { *c = {8, 5}; return true; }
    if (K <= 192) { *c = {8, 6}; return true; }
    if (K <= 224) { *c = {8, 7}; return true; }
    if (K <= 256) { *c = {16, 4}; return true; }
    return false;
}

#define ORIANA_FOR_CFG(cfg, CALL)                                  \
    do {                                                           \
        if (cfg.G == 4 && cfg.T4 == 1) { CALL(4, 1); }             \
        else if (cfg.G == 4 && cfg.T4 == 2) { CALL(4, 2); }        \
        else if (cfg.G == 4 && cfg.T4 == 3) { CALL(4, 3); }        \
        else if (cfg.G == 4 && cfg.T4 == 4) { CALL(4, 4); }        \
        else if (cfg.G == 4 && cfg.T4 == 5) { CALL(4, 5); }        \
        else if (cfg.G == 4 && cfg.T4 == 6) { CALL(4, 6); }        \
        else if (cfg.G == 4 && cfg.T4 == 7) { CALL(4, 7); }        \
        else if (cfg.G == 8 && cfg.T4 == 4) { CALL(8, 4); }        \
        else if (cfg.G == 8 && cfg.T4 == 5) { CALL(8, 5); }        \
        else if (cfg.G == 8 && cfg.T4 == 6) { CALL(8, 6); }        \
        else if (cfg.G == 8 && cfg.T4 == 7) { CALL(8, 7); }        \
        else if (cfg.G == 16 && cfg.T4 == 4) { CALL(16, 4); }      \
        else return ORIANA_EKRANGE;                                \
    } while (0)

template <typename KernelT>
static int set_lds(KernelT kern, size_t bytes) {
    if (bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return -1000 - (int)e;
    }
    return 0;
}

static inline size_t lds_bytes(int G, int T4, int images) {
    const int KP = 4 * G * T4;
    return (size_t)(TILE / pick_nsub(KP, images)) * lds_stride_floats(KP) * sizeof(float) * images;
}

template <int G, int T4>
static int launch_row_pass(const oriana_counts *cm, const float *FU, const float *FVden, const float *FVacc,
                           const float *w_nz, float *R, float *s_col, float *sw_col, float *s_row,
                           int32_t *tile_flag, hipStream_t s) {
    constexpr int SPLIT = TILE / (1024 / G);
    const dim3 grid((unsigned)(cm->nrb * SPLIT)), block(1024);
    const bool sep = FVacc != nullptr && FVacc != FVden;
    const bool hw = w_nz != nullptr;
    const size_t lb = lds_bytes(G, T4, sep ? 2 : 1);
    int rc;
#define ORIANA_RP(SEP, HW)                                                                            \
    rc = set_lds(k_row_pass<G, T4, SEP, HW>, lb);                                                     \
    if (rc) return rc;                                                                                \
    hipLaunchKernelGGL((k_row_pass<G, T4, SEP, HW>), grid, block, lb, s, *cm, FU, FVden, FVacc, w_nz, \
                       R, s_col, sw_col, s_row, tile_flag)
    if (sep && hw) { ORIANA_RP(true, true); }
    else if (sep) { ORIANA_RP(true, false); }
    else if (hw) { ORIANA_RP(false, true); }
    else { ORIANA_RP(false, false); }
#undef ORIANA_RP
    ORIANA_LAUNCH_CHECK();
    return 0;
}

template <int G, int T4>
static int launch_row_spmm(const oriana_counts *cm, const float *s_row, const float *w_nz, const float *FV,
                           float *R, hipStream_t s) {
    constexpr int SPLIT = TILE / (1024 / G);
    const dim3 grid((unsigned)(cm->nrb * SPLIT)), block(1024);
    const size_t lb = lds_bytes(G, T4, 1);
    int rc;
    if (w_nz) {
        rc = set_lds(k_row_spmm<G, T4, true>, lb); if (rc) return rc;
        hipLaunchKernelGGL((k_row_spmm<G, T4, true>), grid, block, lb, s, *cm, s_row, w_nz, FV, R);
    } else {
        rc = set_lds(k_row_spmm<G, T4, false>, lb); if (rc) return rc;
        hipLaunchKernelGGL((k_row_spmm<G, T4, false>), grid, block, lb, s, *cm, s_row, w_nz, FV, R);
    }
    ORIANA_LAUNCH_CHECK();
    return 0;
}

template <int G, int T4>
static int launch_col_pass(const oriana_counts *cm, const float *s_col, const float *Gm, float *C, hipStream_t s) {
    constexpr int SPLIT = TILE / (1024 / G);
    // enough row bands to fill the chip (>= ~1024 workgroups) without shrinking a band below 8 tiles
    int64_t nb = (1024 + cm->ncb * SPLIT - 1) / (cm->ncb * SPLIT);
    int64_t maxb = (cm->nrb + 7) / 8;
    if (nb > maxb) nb = maxb;
    if (nb < 1) nb = 1;
    if (nb > 65535) nb = 65535;
    const int64_t per = (cm->nrb + nb - 1) / nb;
    nb = (cm->nrb + per - 1) / per;
    const dim3 grid((unsigned)(cm->ncb * SPLIT), (unsigned)nb), block(1024);
    const size_t lb = lds_bytes(G, T4, 1);
    int rc = set_lds(k_col_pass<G, T4>, lb);
    if (rc) return rc;
    hipLaunchKernelGGL((k_col_pass<G, T4>), grid, block, lb, s, *cm, s_col, Gm, C, per);
    ORIANA_LAUNCH_CHECK();
    return 0;
}

}  // namespace oriana

using namespace oriana;

extern "C" int64_t oriana_kpad(int64_t K) {
    KCfg c;
    if (!pick_cfg(K, &c)) return 0;
    return 4 * c.G * c.T4;
}

extern "C" const char *oriana_version(void) { return "oriana_hip gfx950 0.1"; }

static bool counts_ok(const oriana_counts *cm) {
    if (!cm || cm->n < 0 || cm->m < 0) return false;
    if (cm->nrb != (cm->n + TILE - 1) / TILE || cm->ncb != (cm->m + TILE - 1) / TILE) return false;
    if (cm->nrb * cm->ncb > 0 && (!cm->tile_off || !cm->row_ptr || !cm->col_ptr)) return false;
    if (cm->nnz > 0 && (!cm->rowrec || !cm->ridx)) return false;
    return true;
}

extern "C" int oriana_factor_prep(float *F, float *mu, const float *logF, const float *mask, int64_t r,
                                  int64_t K, void *stream) {
    const int64_t Kp = oriana_kpad(K);
    if (r < 0 || K <= 0) return ORIANA_EINVAL;
    if (Kp == 0) return ORIANA_EKRANGE;
    if (r == 0) return 0;
    if (!F || !logF) return ORIANA_EINVAL;
    hipLaunchKernelGGL(k_factor_prep, dim3((unsigned)((r + 3) / 4)), dim3(256), 0, (hipStream_t)stream, F, mu,
                       logF, mask, r, (int)K, (int)Kp);
    ORIANA_LAUNCH_CHECK();
    return 0;
}

extern "C" int oriana_row_pass(const oriana_counts *cm, const float *FU, const float *FVden, const float *FVacc,
                               const float *w_nz, float *R, float *s_col, float *sw_col, float *s_row,
                               int32_t *tile_flag, int64_t K, void *stream) {
    if (!counts_ok(cm) || K <= 0) return ORIANA_EINVAL;
    KCfg cfg;
    if (!pick_cfg(K, &cfg)) return ORIANA_EKRANGE;
    if (cm->n == 0) return 0;
    if (!FU || !R || (cm->m > 0 && !FVden) || (cm->nnz > 0 && (!s_col || !tile_flag))) return ORIANA_EINVAL;
    hipStream_t s = (hipStream_t)stream;
#define CALL(G, T) return launch_row_pass<G, T>(cm, FU, FVden, FVacc, w_nz, R, s_col, sw_col, s_row, tile_flag, s)
    ORIANA_FOR_CFG(cfg, CALL);
#undef CALL
    return 0;
}

extern "C" int oriana_row_spmm(const oriana_counts *cm, const float *s_row, const float *w_nz, const float *FV,
                               float *R, int64_t K, void *stream) {
    if (!counts_ok(cm) || K <= 0) return ORIANA_EINVAL;
    KCfg cfg;
    if (!pick_cfg(K, &cfg)) return ORIANA_EKRANGE;
    if (cm->n == 0) return 0;
    if (!R || (cm->m > 0 && !FV) || (cm->nnz > 0 && !s_row)) return ORIANA_EINVAL;
    hipStream_t s = (hipStream_t)stream;
#define CALL(G, T) return launch_row_spmm<G, T>(cm, s_row, w_nz, FV, R, s)
    ORIANA_FOR_CFG(cfg, CALL);
#undef CALL
    return 0;
}

extern "C" int oriana_col_pass(const oriana_counts *cm, const float *s_col, const float *Gm, float *C, int64_t K,
                               void *stream) {
    if (!counts_ok(cm) || K <= 0) return ORIANA_EINVAL;
    KCfg cfg;
    if (!pick_cfg(K, &cfg)) return ORIANA_EKRANGE;
    if (cm->n == 0 || cm->m == 0) return 0;
    if (!Gm || !C || (cm->nnz > 0 && !s_col)) return ORIANA_EINVAL;
    hipStream_t s = (hipStream_t)stream;
#define CALL(G, T) return launch_col_pass<G, T>(cm, s_col, Gm, C, s)
    ORIANA_FOR_CFG(cfg, CALL);
#undef CALL
    return 0;
}

extern "C" int oriana_finalize(float *Z, const float *F, const float *R, const float *mul, int64_t r, int64_t K,
                               int accumulate, void *stream) {
    const int64_t Kp = oriana_kpad(K);
    if (r < 0 || K <= 0) return ORIANA_EINVAL;
    if (Kp == 0) return ORIANA_EKRANGE;
    if (r == 0) return 0;
    if (!Z || !F || !R) return ORIANA_EINVAL;
    const int64_t tot = r * K;
    hipLaunchKernelGGL(k_finalize, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, Z, F, R,
                       mul, r, (int)K, (int)Kp, accumulate);
    ORIANA_LAUNCH_CHECK();
    return 0;
}

extern "C" int oriana_fixup(const oriana_counts *cm, const int32_t *tile_flag, float *s_col, float *sw_col,
                            float *s_row, const float *logU, const float *logV, const float *S_tilde,
                            const float *S_hat, const float *w_nz, const float *dq, float *Zi, float *Zj,
                            float *Zlog, int64_t K, int variant, void *stream) {
    if (!counts_ok(cm) || K <= 0) return ORIANA_EINVAL;
    const int64_t nt = cm->nrb * cm->ncb;
    if (nt == 0 || cm->nnz == 0) return 0;
    if (!tile_flag || !s_col || !logU || !logV) return ORIANA_EINVAL;
    const int quirk = (variant & 4) ? 1 : 0;
    if (quirk && (!dq || K > cm->m)) return ORIANA_EQUIRK;
    hipLaunchKernelGGL(k_fixup, dim3((unsigned)nt), dim3(256), 0, (hipStream_t)stream, *cm, tile_flag, s_col,
                       sw_col, s_row, logU, logV, S_tilde, S_hat, w_nz, dq, Zi, Zj, Zlog, (int)K, quirk);
    ORIANA_LAUNCH_CHECK();
    return 0;
}

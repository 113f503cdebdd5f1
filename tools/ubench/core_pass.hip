// core_pass.hip -- the inner loop of the responsibility pass in isolation (K = 100), to price design
// variants before they go into csrc/passes.hip:
//   G      lanes per row (4: sixteen rows per wave, 1024-thread work-group; 2: thirty-two rows per wave, 512 threads)
//   LAYOUT 0: round-1 LDS image (512-byte rows, chunk groups 0..5 in place, tail replicated in the row padding)
//          1: chunk groups 4, 5 duplicated in slots 6, 7 (every 16-lane service set hits 16 distinct bank groups
//             at every step) + a separate 4-fold tail array
//   PIPE   1: the next step's LDS reads are issued before the current step's FMAs (two K-vector buffers)
//   MODE   0: row pass step (den, reduction, rcp, accumulate)   1: column pass step (accumulate only)
//   ABL    0: full   1: no LDS reads (VALU only)   2: no FMAs (LDS only)
//   STORE  1: scattered 4-byte store of s per step (row pass)   2: one coalesced store per iteration
//   STAGE  1: barrier + image rewrite from a global factor matrix every TILE_IT iterations
//   CTX    n > 0: every n iterations the lane groups switch to other rows: the accumulator goes back to a global
//          matrix (read-modify-write) and the next row's factor vector / accumulator are loaded (per-tile regrouping)
// Records are streamed from global memory like the real kernel (8 bytes per slot, prefetch ring).
// Prints ns per 16 slots per CU and the in-kernel clock (s_memtime / s_memrealtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

template <int CTRL> __device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL> __device__ __forceinline__ uint32_t dpp_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}

constexpr int TILE_IT = 9;             // iterations between two image rewrites (a 256-column tile at 10 % density)
constexpr int WIN = 9216;              // column-side slots of a tile (window of the scattered stores)

template <int G> struct Cfg {
    static constexpr int T4 = 24 / G;      // ds_read_b128 per lane and step
    static constexpr int TF = 4 / G;       // tail floats per lane
    static constexpr int RW = 64 / G;      // rows per wave
};

template <int G> __device__ __forceinline__ int lane_class(int lane) {
    if (G == 4) { const int Q = lane >> 2; return ((Q & 1) << 1) | ((Q >> 1) & 1); }
    const int p = (lane >> 1) & 15;
    // pairs {0,1,6,7,10,11,12,13} and {2,3,4,5,8,9,14,15} are serviced together: classes 0..7 inside each set
    return (p >> 2) * 2 + (p & 1);        // 0 1 0 1 2 3 2 3 4 5 4 5 6 7 6 7
}

// float4 index inside a factor row in global memory (25 float4, the last one is the tail) and inside the LDS row
template <int G, int LAYOUT> __device__ __forceinline__ void chunk_maps(int lane, int t, int &gidx, int &lidx) {
    const int q = lane & (G - 1), a = lane_class<G>(lane);
    if (G == 4) {
        if (LAYOUT == 0) {
            const int c = (lane >> 2) & 3;
            int ch = t ^ (c & 1);
            if ((c & 2) && ch >= 2) ch = (ch < 4) ? ch + 2 : ch - 2;
            gidx = lidx = ch * 4 + q;
        } else {
            int cg, slot;
            if (t < 4) { cg = (a + t) & 3; slot = cg; }
            else { cg = 4 + ((a & 1) ^ (t & 1)); slot = cg + 2 * (a >> 1); }
            gidx = cg * 4 + q; lidx = slot * 4 + q;
        }
    } else {
        int pc, slot;
        if (t < 8) { pc = (a + t) & 7; slot = pc; }
        else { pc = 8 + ((a + t) & 3); slot = pc + ((a >= 4) ? 4 : 0); }
        gidx = pc * 2 + q; lidx = slot * 2 + q;
    }
}

template <int G, int THREADS, int LAYOUT, int PIPE, int MODE, int ABL, int STORE, int STAGE, int CTX>
__global__ __launch_bounds__(THREADS) void k(float *__restrict__ out, const unsigned long long *__restrict__ rec,
                                             float *__restrict__ sdst, const float *__restrict__ F, int iters,
                                             unsigned long long *__restrict__ clk) {
    using C = Cfg<G>;
    constexpr int T4 = C::T4, TF = C::TF;
    constexpr int NW = THREADS / 64;
    extern __shared__ f4 lds[];                 // [256][32] float4 (+ [256][4] float4 tails for LAYOUT 1)
    float *tails = reinterpret_cast<float *>(lds + 256 * 32);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane & (G - 1);
    // ---- image fill (also the per-tile restaging when STAGE) ----
    auto stage = [&](int tile) {
        const f4 *src = reinterpret_cast<const f4 *>(F) + (size_t)((tile * 256) % 29952) * 25;
        for (int idx = tid; idx < 256 * 25; idx += THREADS) {
            const int jr = idx / 25, c4 = idx - jr * 25;
            const f4 v = src[idx];
            if (c4 < 24) {
                lds[jr * 32 + c4] = v;
                if (LAYOUT == 1 && c4 >= 16) lds[jr * 32 + c4 + 8] = v;
            } else {
                if (LAYOUT == 1) { for (int r = 0; r < 4; ++r) reinterpret_cast<f4 *>(tails)[jr * 4 + r] = v; }
                else { for (int r = 0; r < 8; ++r) lds[jr * 32 + 24 + r] = v; }
            }
        }
    };
    stage(blockIdx.x);
    __syncthreads();
    int gidx[T4], lidx[T4];
    #pragma unroll
    for (int t = 0; t < T4; ++t) chunk_maps<G, LAYOUT>(lane, t, gidx[t], lidx[t]);
    int toff;       // float offset of this lane's tail inside `tails` (LAYOUT 1) or inside the LDS row (LAYOUT 0)
    if (G == 4) toff = (LAYOUT == 1) ? ((lane >> 2) & 3) * 4 + q : 96 + ((lane >> 2) & 7) * 4 + q;
    else toff = ((lane >> 1) & 3) * 4 + 2 * q;

    f4 fu[T4], acc[T4];
    float fut[TF], acct[TF];
    const int myrow = (blockIdx.x * NW + wave) * C::RW + lane / G;
    #pragma unroll
    for (int t = 0; t < T4; ++t) {
        fu[t] = reinterpret_cast<const f4 *>(F)[(size_t)(myrow % 29952) * 25 + gidx[t]];
        acc[t] = f4{0.f, 0.f, 0.f, 0.f};
    }
    #pragma unroll
    for (int u = 0; u < TF; ++u) { fut[u] = F[(size_t)(myrow % 29952) * 100 + 96 + q * TF + u]; acct[u] = 0.f; }

    // record stream of this wave: iteration = 64 slots (G = 4) or 2 x 64 slots (G = 2: one slice per half wave)
    constexpr int RPL = 4 / G;                  // 8-byte records per lane and iteration: 1 or 2
    const unsigned long long *rp = rec + ((size_t)(blockIdx.x * NW + wave) * iters) * (64 * RPL) + lane * RPL;
    constexpr int PD = 3;
    unsigned long long ring[PD][RPL];
    #pragma unroll
    for (int d = 0; d < PD; ++d)
        #pragma unroll
        for (int r = 0; r < RPL; ++r) ring[d][r] = rp[(size_t)d * 64 * RPL + r];

    unsigned long long c0 = 0, r0 = 0;
    if (tid == 0) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }

    f4 vb[2][T4];
    float vtb[2][TF];
    auto issue = [&](int col, f4 (&v)[T4], float (&vt)[TF]) {
        const f4 *vrow = lds + col * 32;
        if (ABL == 1) {
            #pragma unroll
            for (int t = 0; t < T4; ++t) { v[t] = fu[t]; }
            asm volatile("" : "+v"(v[0]));
            #pragma unroll
            for (int u = 0; u < TF; ++u) vt[u] = fut[u];
        } else {
            #pragma unroll
            for (int t = 0; t < T4; ++t) v[t] = vrow[lidx[t]];
            if (LAYOUT == 1) {
                const float *tr = tails + col * 16 + toff;
                if (TF == 1) vt[0] = tr[0];
                else { const f2 t2 = *reinterpret_cast<const f2 *>(tr); vt[0] = t2.x; vt[1] = t2.y; }
            } else {
                vt[0] = reinterpret_cast<const float *>(vrow)[toff];
            }
        }
    };
    auto compute = [&](float x, uint32_t meta, const f4 (&v)[T4], const float (&vt)[TF], float *sbase, float &sbuf, int u) {
        float s;
        if (MODE == 0) {
            f2 d01 = {0.f, 0.f}, d23 = {0.f, 0.f};
            if (ABL != 2) {
                #pragma unroll
                for (int t = 0; t < T4; ++t) {
                    d01 = __builtin_elementwise_fma(fu[t].xy, v[t].xy, d01);
                    d23 = __builtin_elementwise_fma(fu[t].zw, v[t].zw, d23);
                }
            } else {
                #pragma unroll
                for (int t = 0; t < T4; ++t) { d01 += v[t].xy; }
            }
            const f2 dd = d01 + d23;
            float den = dd.x + dd.y;
            #pragma unroll
            for (int uu = 0; uu < TF; ++uu) den = fmaf(fut[uu], vt[uu], den);
            den += dpp_f32<0xB1>(den);                    // lanes 0<->1, 2<->3
            if (G == 4) den += dpp_f32<0x4E>(den);        // pairs
            const bool ok = den >= 1e-10f;
            s = (ok && x != 0.f) ? x * __builtin_amdgcn_rcpf(den) : 0.f;
        } else {
            s = x;
        }
        if (ABL != 2) {
            const f2 ss = {s, s};
            #pragma unroll
            for (int t = 0; t < T4; ++t) {
                acc[t].xy = __builtin_elementwise_fma(ss, v[t].xy, acc[t].xy);
                acc[t].zw = __builtin_elementwise_fma(ss, v[t].zw, acc[t].zw);
            }
            #pragma unroll
            for (int uu = 0; uu < TF; ++uu) acct[uu] = fmaf(s, vt[uu], acct[uu]);
        } else {
            acc[0].x += s;
        }
        if (MODE == 0 && STORE == 1) sbase[meta & 0xFFFFu] = s;
        if (MODE == 0 && STORE == 2) sbuf = ((lane & 3) == u || G == 2) ? s : sbuf;
    };

    float *Rg = sdst + (size_t)256 * 64 * WIN;           // context matrix behind the store windows
    int ctxrow = myrow;
    for (int it = 0; it < iters; ++it) {
        if (CTX > 0 && it > 0 && (it % CTX) == 0) {
            // context switch: store the accumulator of the current rows, load factor vector + accumulator of the next ones
            f4 *rdst = reinterpret_cast<f4 *>(Rg) + (size_t)(ctxrow % 131072) * 25;
            #pragma unroll
            for (int t = 0; t < T4; ++t) rdst[gidx[t]] = acc[t];
            ctxrow = (ctxrow * 17 + 12345 + it) & 0x7FFFFFF;
            ctxrow = (ctxrow / C::RW) * C::RW + (lane / G);            // the wave's rows stay a block of RW consecutive rows
            const f4 *fsrc = reinterpret_cast<const f4 *>(F) + (size_t)(ctxrow % 29952) * 25;
            const f4 *rsrc = reinterpret_cast<const f4 *>(Rg) + (size_t)(ctxrow % 131072) * 25;
            #pragma unroll
            for (int t = 0; t < T4; ++t) { if (MODE == 0) fu[t] = fsrc[gidx[t]]; acc[t] = rsrc[gidx[t]]; }
        }
        if (STAGE && it > 0 && (it % TILE_IT) == 0) {
            __syncthreads();
            stage(blockIdx.x + it);
            __syncthreads();
        }
        unsigned long long cur[RPL];
        #pragma unroll
        for (int r = 0; r < RPL; ++r) cur[r] = ring[0][r];
        #pragma unroll
        for (int d = 0; d + 1 < PD; ++d)
            #pragma unroll
            for (int r = 0; r < RPL; ++r) ring[d][r] = ring[d + 1][r];
        const int nx = (it + PD < iters) ? it + PD : iters - 1;
        #pragma unroll
        for (int r = 0; r < RPL; ++r) ring[PD - 1][r] = rp[(size_t)nx * 64 * RPL + r];
        float *sbase = sdst + ((size_t)blockIdx.x * 64 + ((it / TILE_IT) & 63)) * WIN;
        float sbuf = 0.f;
        // the four records of this row: (x, meta) pairs, broadcast inside the lane group
        float xs[4]; uint32_t ms[4];
        if (G == 4) {
            const uint32_t rx = (uint32_t)cur[0], rm = (uint32_t)(cur[0] >> 32);
            xs[0] = __uint_as_float(dpp_u32<0x00>(rx)); ms[0] = dpp_u32<0x00>(rm);
            xs[1] = __uint_as_float(dpp_u32<0x55>(rx)); ms[1] = dpp_u32<0x55>(rm);
            xs[2] = __uint_as_float(dpp_u32<0xAA>(rx)); ms[2] = dpp_u32<0xAA>(rm);
            xs[3] = __uint_as_float(dpp_u32<0xFF>(rx)); ms[3] = dpp_u32<0xFF>(rm);
        } else {
            // lane q of a pair holds records 2q, 2q+1: quad_perm [0,0,2,2] = 0xA0, [1,1,3,3] = 0xF5
            const uint32_t ax = (uint32_t)cur[0], am = (uint32_t)(cur[0] >> 32), bx = (uint32_t)cur[1], bm = (uint32_t)(cur[1] >> 32);
            xs[0] = __uint_as_float(dpp_u32<0xA0>(ax)); ms[0] = dpp_u32<0xA0>(am);
            xs[1] = __uint_as_float(dpp_u32<0xA0>(bx)); ms[1] = dpp_u32<0xA0>(bm);
            xs[2] = __uint_as_float(dpp_u32<0xF5>(ax)); ms[2] = dpp_u32<0xF5>(am);
            xs[3] = __uint_as_float(dpp_u32<0xF5>(bx)); ms[3] = dpp_u32<0xF5>(bm);
        }
        if (PIPE) {
            issue((int)((ms[0] >> 16) & 0xFFu), vb[0], vtb[0]);
            #pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (u + 1 < 4) issue((int)((ms[u + 1] >> 16) & 0xFFu), vb[(u + 1) & 1], vtb[(u + 1) & 1]);
                compute(xs[u], ms[u], vb[u & 1], vtb[u & 1], sbase, sbuf, u);
            }
        } else {
            #pragma unroll
            for (int u = 0; u < 4; ++u) {
                issue((int)((ms[u] >> 16) & 0xFFu), vb[0], vtb[0]);
                compute(xs[u], ms[u], vb[0], vtb[0], sbase, sbuf, u);
                #pragma unroll
                for (int t = 0; t < T4; ++t) asm volatile("" : "+v"(acc[t]));
            }
        }
        if (MODE == 0 && STORE == 2) sdst[((size_t)(blockIdx.x * NW + wave) * 64 + (it & 63)) * 64 + lane] = sbuf;
    }
    if (tid == 0) {
        clk[blockIdx.x * 2 + 0] = __builtin_amdgcn_s_memtime() - c0;
        clk[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    float r = 0.f;
    #pragma unroll
    for (int t = 0; t < T4; ++t) r += acc[t].x + acc[t].y + acc[t].z + acc[t].w;
    #pragma unroll
    for (int u = 0; u < TF; ++u) r += acct[u];
    out[(size_t)blockIdx.x * THREADS + tid] = r;
}

__global__ void k_init_rec(unsigned long long *rec, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)(i * 2654435761ull) ^ (uint32_t)(i >> 13);
        h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
        const uint32_t col = h & 0xFFu;
        const uint32_t cd = (h >> 8) % WIN;
        const float x = ((h >> 28) < 12) ? 1.0f + (float)((h >> 24) & 7) : 0.0f;     // ~ 3/4 of the slots carry an entry
        rec[i] = ((unsigned long long)((col << 16) | cd) << 32) | __float_as_uint(x);
    }
}
__global__ void k_init_f(float *F, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)(i * 2246822519ull); h ^= h >> 15; h *= 0x85ebca6bu; h ^= h >> 13;
        F[i] = 0.05f + (float)(h & 0xFFFF) / 65536.0f;
    }
}

struct Bufs { float *out; unsigned long long *rec; float *sdst; float *F; unsigned long long *clk; };

template <int G, int THREADS, int LAYOUT, int PIPE, int MODE, int ABL, int STORE, int STAGE, int CTX = 0>
void run(const Bufs &b, int iters, const char *name) {
    auto kern = k<G, THREADS, LAYOUT, PIPE, MODE, ABL, STORE, STAGE, CTX>;
    const size_t lb = 256 * 512 + (LAYOUT == 1 ? 256 * 64 : 0);
    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(THREADS), lb, 0, b.out, b.rec, b.sdst, b.F, iters, b.clk);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms);
    }
    hipError_t e = hipGetLastError();
    std::vector<unsigned long long> c(512);
    (void)hipMemcpy(c.data(), b.clk, 512 * 8, hipMemcpyDeviceToHost);
    std::vector<double> ghz;
    for (int i = 0; i < 256; ++i) if (c[2 * i + 1]) ghz.push_back((double)c[2 * i] / (double)c[2 * i + 1] * 0.1);
    std::sort(ghz.begin(), ghz.end());
    const double clock = ghz.empty() ? 0.0 : ghz[ghz.size() / 2];
    const double slots_per_cu = (double)(THREADS / 64) * iters * 64.0 * (4 / G);     // slots per work-group (= per CU)
    const double ns16 = best * 1e6 / (slots_per_cu / 16.0);
    printf("%-44s %8.3f ms  %7.2f ns / 16 slots / CU  = %6.1f CU-cycles @ %.2f GHz   (%.1f ms per 4.1e9 slots)%s\n", name, best, ns16,
           ns16 * clock, clock, ns16 * (4.1e9 / 16.0) / 256.0 * 1e-6, e == hipSuccess ? "" : "  [HIP ERROR]");
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
}

int main(int argc, char **argv) {
    const int iters = (argc > 1) ? atoi(argv[1]) : 576;
    Bufs b;
    const size_t nrec = (size_t)256 * 16 * iters * 128 + 4096;
    (void)hipMalloc(&b.out, 256 * 1024 * 4);
    (void)hipMalloc(&b.rec, nrec * 8);
    (void)hipMalloc(&b.sdst, ((size_t)256 * 64 + 1) * WIN * 4 + (size_t)131072 * 100 * 4);
    (void)hipMalloc(&b.F, (size_t)30208 * 100 * 4);
    (void)hipMalloc(&b.clk, 512 * 8);
    hipLaunchKernelGGL(k_init_rec, dim3(4096), dim3(256), 0, 0, b.rec, nrec);
    hipLaunchKernelGGL(k_init_f, dim3(1024), dim3(256), 0, 0, b.F, (size_t)30208 * 100);
    (void)hipMemset(b.sdst, 0, ((size_t)256 * 64 + 1) * WIN * 4 + (size_t)131072 * 100 * 4);
    (void)hipDeviceSynchronize();
    //   G  THR  LAY PIPE MODE ABL STORE STAGE CTX
    printf("---- row pass step ----\n");
    run<4, 1024, 0, 0, 0, 0, 1, 1>(b, iters, "row G4 1024 round-1 layout, store, stage");
    run<4, 1024, 0, 0, 0, 0, 1, 1>(b, iters, "row G4 1024 round-1 layout, store, stage (again)");
    run<4, 1024, 1, 0, 0, 0, 1, 1>(b, iters, "row G4 1024 dup, store, stage");
    run<4, 1024, 1, 0, 0, 0, 1, 1, 7>(b, iters, "row G4 1024 dup, store, stage, ctx/7");
    run<4, 1024, 1, 0, 0, 0, 1, 1, 4>(b, iters, "row G4 1024 dup, store, stage, ctx/4");
    run<4, 1024, 1, 0, 0, 0, 0, 0>(b, iters, "row G4 1024 dup, NO store, NO stage");
    run<4, 1024, 1, 0, 0, 1, 0, 0>(b, iters, "row G4 1024 VALU only");
    run<2, 512, 1, 0, 0, 0, 1, 1>(b, iters, "row G2 512 dup, store, stage");
    run<2, 512, 1, 0, 0, 0, 1, 1, 7>(b, iters, "row G2 512 dup, store, stage, ctx/7");
    run<2, 512, 1, 0, 0, 0, 1, 1, 4>(b, iters, "row G2 512 dup, store, stage, ctx/4");
    run<2, 512, 1, 0, 0, 0, 0, 0>(b, iters, "row G2 512 dup, NO store, NO stage");
    run<2, 512, 1, 0, 0, 1, 0, 0>(b, iters, "row G2 512 VALU only");
    run<2, 512, 1, 1, 0, 0, 1, 1>(b, iters, "row G2 512 dup, pipelined, store, stage");
    printf("---- column pass step ----\n");
    run<4, 1024, 0, 0, 1, 0, 0, 1>(b, iters, "col G4 1024 round-1 layout, stage");
    run<4, 1024, 1, 0, 1, 0, 0, 1>(b, iters, "col G4 1024 dup, stage");
    run<4, 1024, 1, 0, 1, 0, 0, 1, 7>(b, iters, "col G4 1024 dup, stage, ctx/7");
    run<4, 1024, 1, 0, 1, 0, 0, 0>(b, iters, "col G4 1024 dup, NO stage");
    run<4, 1024, 1, 0, 1, 1, 0, 0>(b, iters, "col G4 1024 VALU only");
    run<2, 512, 1, 0, 1, 0, 0, 1>(b, iters, "col G2 512 dup, stage");
    run<2, 512, 1, 0, 1, 0, 0, 1, 7>(b, iters, "col G2 512 dup, stage, ctx/7");
    run<2, 1024, 1, 0, 1, 0, 0, 1>(b, iters, "col G2 1024 dup, stage");
    run<2, 1024, 1, 0, 1, 0, 0, 1, 7>(b, iters, "col G2 1024 dup, stage, ctx/7");
    run<2, 512, 1, 0, 1, 0, 0, 0>(b, iters, "col G2 512 dup, NO stage");
    run<2, 512, 1, 0, 1, 1, 0, 0>(b, iters, "col G2 512 VALU only");
    return 0;
}

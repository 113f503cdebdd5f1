// tile_width.hip -- VERDICT r5 item 4: is a narrower gene tile worth a layout change?  The K = 100 row pass (two lanes per
// row, 512 threads, the inner loop of csrc/passes_k100.h as modelled by core_pass.hip) over the REAL per-tile row-length
// distribution of the benchmark matrix (tools/tile_width_dist.py: sliced genes of configs[3] after the hybrid layout's dense
// cut), for gene tiles of 256 / 192 / 128 genes:
//   W = 256, rotating image   the shipped layout: 512-byte image rows, pair-chunk groups rotated per lane class, every
//                             ds_read_b128 address = row base + a per-lane chunk offset (one v_add per read, 12 per step)
//   W = 192 / 128, affine     circular image rows (every rotating group stored twice: 768-byte rows), so the chunk a lane
//                             reads at step t is at  lane base + 32 t  bytes: ONE address per step + immediate offsets
// A work-group owns one 256-row block and walks all its gene tiles: barrier, restage W factor rows, then every wave runs the
// iterations its own two 16-row slices need (the longer of the two), i.e. padding, the barrier wait for the slowest wave and the
// extra tiles of a narrower width are all in the measured time.  One launch = one round of the chip (256 row blocks);
// configs[3] is 15.26 rounds.
//   hipcc --offload-arch=gfx950 -O3 -o tile_width tools/ubench/tile_width.hip && ./tile_width dist_256.bin dist_192.bin dist_128.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int CTRL> __device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL> __device__ __forceinline__ uint32_t dpp_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}

constexpr int WINS = 12288;            // column-side slots of a tile (window of the scattered stores of s)
constexpr int T4 = 12;                 // ds_read_b128 per lane and step (two lanes per row)

__device__ __forceinline__ int lane_class(int lane) {
    const int p = (lane >> 1) & 15;
    return (p >> 2) * 2 + (p & 1);     // 0 1 0 1 2 3 2 3 4 5 4 5 6 7 6 7: classes 0..7 inside each 16-lane service set
}

// ROW4: float4 per image row.  rotating: 32 (512 bytes: pair-chunks 0..7, 8..11 and the copy 12..15 of 8..11); affine: 48
template <int W, bool AFFINE>
__global__ __launch_bounds__(512) void k(float *__restrict__ out, const unsigned long long *__restrict__ rec,
                                         float *__restrict__ sdst, const float *__restrict__ F,
                                         const int32_t *__restrict__ wit /* [nblocks][ntiles][8] */, int nblocks, int ntiles,
                                         int rec_iters) {
    constexpr int ROW4 = AFFINE ? 48 : 32;
    extern __shared__ f4 lds[];                       // [W][ROW4] float4 + [W][4] float4 tails
    float *tails = reinterpret_cast<float *>(lds + W * ROW4);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane & 1, a = lane_class(lane);
    auto stage = [&](int tile) {
        const f4 *src = reinterpret_cast<const f4 *>(F) + (size_t)((tile * W) % 29952) * 25;
        for (int idx = tid; idx < W * 25; idx += 512) {
            const int jr = idx / 25, c4 = idx - jr * 25;
            const f4 v = src[idx];
            if (c4 < 24) {
                if (AFFINE) {
                    // pair-chunk pc = c4 / 2: group A (pc < 8) at positions pc and pc + 8; group B at 16 + (pc - 8) and + 4
                    const int pc = c4 >> 1, h = c4 & 1;
                    const int p0 = pc < 8 ? pc : 16 + (pc - 8);
                    lds[jr * ROW4 + p0 * 2 + h] = v;
                    lds[jr * ROW4 + (p0 + (pc < 8 ? 8 : 4)) * 2 + h] = v;
                } else {
                    lds[jr * ROW4 + c4] = v;
                    if (c4 >= 16) lds[jr * ROW4 + c4 + 8] = v;
                }
            } else {
                for (int r = 0; r < 4; ++r) reinterpret_cast<f4 *>(tails)[jr * 4 + r] = v;
            }
        }
    };
    int gidx[T4], lidx[T4];
#pragma unroll
    for (int t = 0; t < T4; ++t) {
        int pc, slot;
        if (t < 8) { pc = (a + t) & 7; slot = pc; }
        else { pc = 8 + ((a + t) & 3); slot = pc + ((a >= 4) ? 4 : 0); }
        gidx[t] = pc * 2 + q; lidx[t] = slot * 2 + q;
    }
    const int abase = a * 2 + q;                                            // affine: float4 index of step 0 inside the row
    const int bbase = (16 + (a & 3)) * 2 + q;                                // ... of step 8
    const int toff = ((lane >> 1) & 3) * 4 + 2 * q;
    f4 fu[T4], acc[T4];
    float fut[2], acct[2];
    const int myrow = (blockIdx.x * 8 + wave) * 32 + lane / 2;
#pragma unroll
    for (int t = 0; t < T4; ++t) {
        fu[t] = reinterpret_cast<const f4 *>(F)[(size_t)(myrow % 29952) * 25 + gidx[t]];
        acc[t] = f4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) { fut[u] = F[(size_t)(myrow % 29952) * 100 + 96 + q * 2 + u]; acct[u] = 0.f; }
    const unsigned long long *rp0 = rec + ((size_t)(blockIdx.x * 8 + wave) * rec_iters) * 128 + lane * 2;
    const int32_t *mywit = wit + ((size_t)(blockIdx.x % nblocks) * ntiles) * 8 + wave;
    int itpos = 0;
    for (int tile = 0; tile < ntiles; ++tile) {
        __syncthreads();
        stage(tile);
        __syncthreads();
        const int nit = mywit[tile * 8];
        float *sbase = sdst + ((size_t)blockIdx.x * 64 + (tile & 63)) * WINS;
        const unsigned long long *rp = rp0 + (size_t)(itpos % (rec_iters - 4)) * 128;
        itpos += nit;
        constexpr int PD = 3;
        unsigned long long ring[PD][2];
#pragma unroll
        for (int d = 0; d < PD; ++d) { ring[d][0] = rp[(size_t)d * 128]; ring[d][1] = rp[(size_t)d * 128 + 1]; }
        for (int it = 0; it < nit; ++it) {
            unsigned long long cur[2] = {ring[0][0], ring[0][1]};
#pragma unroll
            for (int d = 0; d + 1 < PD; ++d) { ring[d][0] = ring[d + 1][0]; ring[d][1] = ring[d + 1][1]; }
            const int nx = (it + PD < nit) ? it + PD : nit - 1;
            ring[PD - 1][0] = rp[(size_t)nx * 128]; ring[PD - 1][1] = rp[(size_t)nx * 128 + 1];
            const uint32_t ax = (uint32_t)cur[0], am = (uint32_t)(cur[0] >> 32), bx = (uint32_t)cur[1], bm = (uint32_t)(cur[1] >> 32);
            float xs[4]; uint32_t ms[4];
            xs[0] = __uint_as_float(dpp_u32<0xA0>(ax)); ms[0] = dpp_u32<0xA0>(am);
            xs[1] = __uint_as_float(dpp_u32<0xA0>(bx)); ms[1] = dpp_u32<0xA0>(bm);
            xs[2] = __uint_as_float(dpp_u32<0xF5>(ax)); ms[2] = dpp_u32<0xF5>(am);
            xs[3] = __uint_as_float(dpp_u32<0xF5>(bx)); ms[3] = dpp_u32<0xF5>(bm);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int col = (int)((ms[u] >> 16) & 0xFFu);
                if (W < 256) col = (col * W) >> 8;                           // a gene of this tile
                f4 v[T4];
                float vt[2];
                if (AFFINE) {
                    const f4 *va = lds + col * ROW4 + abase, *vb = lds + col * ROW4 + bbase;
#pragma unroll
                    for (int t = 0; t < 8; ++t) v[t] = va[2 * t];            // + 32 t bytes: immediate offsets
#pragma unroll
                    for (int t = 8; t < T4; ++t) v[t] = vb[2 * (t - 8)];
                } else {
                    const f4 *vrow = lds + col * ROW4;
#pragma unroll
                    for (int t = 0; t < T4; ++t) v[t] = vrow[lidx[t]];
                }
                { const f2 t2 = *reinterpret_cast<const f2 *>(tails + col * 16 + toff); vt[0] = t2.x; vt[1] = t2.y; }
                f2 d01 = {0.f, 0.f}, d23 = {0.f, 0.f};
#pragma unroll
                for (int t = 0; t < T4; ++t) {
                    d01 = __builtin_elementwise_fma(fu[t].xy, v[t].xy, d01);
                    d23 = __builtin_elementwise_fma(fu[t].zw, v[t].zw, d23);
                }
                const f2 dd = d01 + d23;
                float den = dd.x + dd.y;
                den = fmaf(fut[0], vt[0], den); den = fmaf(fut[1], vt[1], den);
                den += dpp_f32<0xB1>(den);
                const float x = xs[u];
                const float s = (den >= 1e-10f && x != 0.f) ? x * __builtin_amdgcn_rcpf(den) : 0.f;
                const f2 ss = {s, s};
#pragma unroll
                for (int t = 0; t < T4; ++t) {
                    acc[t].xy = __builtin_elementwise_fma(ss, v[t].xy, acc[t].xy);
                    acc[t].zw = __builtin_elementwise_fma(ss, v[t].zw, acc[t].zw);
                }
                acct[0] = fmaf(s, vt[0], acct[0]); acct[1] = fmaf(s, vt[1], acct[1]);
                sbase[(ms[u] & 0xFFFFu) % WINS] = s;
#pragma unroll
                for (int t = 0; t < T4; ++t) asm volatile("" : "+v"(acc[t]));
            }
        }
    }
    float r = 0.f;
#pragma unroll
    for (int t = 0; t < T4; ++t) r += acc[t].x + acc[t].y + acc[t].z + acc[t].w;
    r += acct[0] + acct[1];
    out[(size_t)blockIdx.x * 512 + tid] = r;
}

__global__ void k_init_rec(unsigned long long *rec, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)(i * 2654435761ull) ^ (uint32_t)(i >> 13);
        h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
        const uint32_t col = h & 0xFFu;
        const uint32_t cd = (h >> 8) % WINS;
        const float x = ((h >> 28) < 11) ? 1.0f + (float)((h >> 24) & 7) : 0.0f;     // ~ 0.69 of the slots carry an entry (slot efficiency)
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

struct Dist { int W = 0, nblocks = 0, ntiles = 0; double nnz = 0, slots = 0, wave_slots = 0, wg_slots = 0; std::vector<int32_t> wit; };

static bool load(const char *path, Dist &d) {
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    int32_t h[4]; double s[4];
    if (fread(h, 4, 4, f) != 4 || fread(s, 8, 4, f) != 4) { fclose(f); return false; }
    d.W = h[0]; d.nblocks = h[1]; d.ntiles = h[2];
    d.nnz = s[0]; d.slots = s[1]; d.wave_slots = s[2]; d.wg_slots = s[3];
    d.wit.resize((size_t)d.nblocks * d.ntiles * 8);
    const bool ok = fread(d.wit.data(), 4, d.wit.size(), f) == d.wit.size();
    fclose(f);
    return ok;
}

template <int W, bool AFFINE>
static void run(const Dist &d, float *out, unsigned long long *rec, float *sdst, float *F, int rec_iters, const char *name) {
    int32_t *wit;
    (void)hipMalloc(&wit, d.wit.size() * 4);
    (void)hipMemcpy(wit, d.wit.data(), d.wit.size() * 4, hipMemcpyHostToDevice);
    auto kern = k<W, AFFINE>;
    const size_t lb = (size_t)W * (AFFINE ? 48 : 32) * 16 + (size_t)W * 64;
    if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb) != hipSuccess) {
        printf("%-34s LDS %zu bytes: not available\n", name, lb);
        (void)hipGetLastError();
        return;
    }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), lb, 0, out, rec, sdst, F, wit, d.nblocks, d.ntiles, rec_iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep) best = std::min(best, ms);
    }
    const hipError_t e = hipGetLastError();
    // slots issued per CU and round: every wave's iterations x 128 slots; the work-group's time follows its slowest wave per tile
    double wave_it = 0, wg_it = 0;
    for (int b = 0; b < d.nblocks; ++b)
        for (int t = 0; t < d.ntiles; ++t) {
            int mx = 0;
            for (int w = 0; w < 8; ++w) { const int v = d.wit[((size_t)b * d.ntiles + t) * 8 + w]; wave_it += v; mx = std::max(mx, v); }
            wg_it += mx;
        }
    wave_it /= d.nblocks; wg_it /= d.nblocks;                              // per row block
    const double ns16 = best * 1e6 / (wave_it * 128.0 / 16.0);            // ns per 16 ISSUED slots per CU
    printf("%-34s LDS %6zu B  tiles %4d  slot eff %.3f (x barrier %.3f)  %8.3f ms per round  %6.2f ns / 16 issued slots / CU"
           "  -> configs[3] row pass of the sliced genes: %6.2f ms%s\n",
           name, lb, d.ntiles, d.nnz / d.slots, d.nnz / d.wg_slots, best, ns16, best * 3907.0 / 256.0, e == hipSuccess ? "" : "  [HIP ERROR]");
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(wit);
}

int main(int argc, char **argv) {
    if (argc < 4) { printf("usage: tile_width dist_256.bin dist_192.bin dist_128.bin\n"); return 1; }
    Dist d256, d192, d128;
    if (!load(argv[1], d256) || !load(argv[2], d192) || !load(argv[3], d128) || d256.W != 256 || d192.W != 192 || d128.W != 128) {
        printf("cannot read the distributions\n"); return 1;
    }
    const int rec_iters = 4096;
    float *out, *sdst, *F; unsigned long long *rec;
    const size_t nrec = (size_t)256 * 8 * rec_iters * 128 + 4096;
    (void)hipMalloc(&out, 256 * 512 * 4);
    (void)hipMalloc(&rec, nrec * 8);
    (void)hipMalloc(&sdst, ((size_t)256 * 64 + 1) * WINS * 4);
    (void)hipMalloc(&F, (size_t)30208 * 100 * 4);
    hipLaunchKernelGGL(k_init_rec, dim3(4096), dim3(256), 0, 0, rec, nrec);
    hipLaunchKernelGGL(k_init_f, dim3(1024), dim3(256), 0, 0, F, (size_t)30208 * 100);
    (void)hipMemset(sdst, 0, ((size_t)256 * 64 + 1) * WINS * 4);
    (void)hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) {
        run<256, false>(d256, out, rec, sdst, F, rec_iters, "W = 256 rotating image (shipped)");
        run<192, true>(d192, out, rec, sdst, F, rec_iters, "W = 192 affine image");
        run<128, true>(d128, out, rec, sdst, F, rec_iters, "W = 128 affine image");
        run<192, false>(d192, out, rec, sdst, F, rec_iters, "W = 192 rotating image");
        run<128, false>(d128, out, rec, sdst, F, rec_iters, "W = 128 rotating image");
    }
    return 0;
}

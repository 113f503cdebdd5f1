// One lane per row: the inner loop of the row pass for Kp <= 64 with a wave of 64 rows (four 16-row slices), no
// cross-lane traffic at all.  The factor image holds 256 rows of 16 float4 (256-byte rows); lane class a (its place in
// the 16-lane ds_read_b128 service set) visits chunk (a + t) & 15 at step t, so the 16 lanes of a set always read 16
// different bank groups whatever their columns.  Measures ns per 16 slots per CU, to set against the shipped
// four-lanes-per-row kernel (C5, K = 64: 29 ns; C3, K = 50: 32 ns per 16 slots per CU incl. staging and stores).
//   MODE 0: full step (dot, rcp, accumulate, scattered store of s)   1: + second image for the accumulation
//   ABL  0: full   1: no LDS reads   2: no FMAs   3: no scattered store of s   4: s through an LDS window, coalesced
//        flush per tile (what a tile-local transposition of the s stream would cost)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int lane_class(int lane) {
    // service sets of ds_read_b128: {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32)
    const int l = lane & 31;
    if (l < 4) return l;
    if (l < 12) return l - 4;
    if (l < 16) return l - 8;
    if (l < 20) return l - 8;
    if (l < 28) return l - 12;
    return l - 16;
}

constexpr int WIN = 9216;

template <int MODE, int ABL>
__global__ __launch_bounds__(512) void k(const u4 *__restrict__ rec, const float *__restrict__ F, float *__restrict__ sdst,
                                         float *__restrict__ out, int iters, int tile_it) {
    extern __shared__ f4 lds[];                                   // [images][256][16]
    const int tid = threadIdx.x, lane = tid & 63;
    const int cls = lane_class(lane);
    int loff[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) loff[t] = (cls + t) & 15;
    f4 fu[16], acc[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) { fu[t] = f4{1.f + t, 0.5f, 0.25f + lane, 0.125f}; acc[t] = f4{0, 0, 0, 0}; }
    const u4 *rp = rec + ((size_t)blockIdx.x * 512 + tid) * 2;
    float *sd = sdst + (size_t)blockIdx.x * WIN;
    u4 q0 = rp[0], q1 = rp[1];
    for (int it = 0; it < iters; ++it) {
        if (it % tile_it == 0) {                                  // restage the image(s)
            __syncthreads();
            if (ABL == 4 && it > 0) {                             // flush the tile's window of s, 16 bytes per lane
                const f4 *w = lds + (MODE ? 2 : 1) * 256 * 16;
                for (int idx = tid; idx < 8192 / 4; idx += 512) reinterpret_cast<f4 *>(sd)[idx] = w[idx];
            }
            for (int idx = tid; idx < (MODE ? 2 : 1) * 256 * 16; idx += 512) lds[idx] = reinterpret_cast<const f4 *>(F)[(idx + it) & 0xFFFF];
            __syncthreads();
        }
        const u4 c0 = q0, c1 = q1;
        const size_t nx = ((size_t)(it + 1) % 64) * 1024 * gridDim.x;
        q0 = rp[nx]; q1 = rp[nx + 1];
#pragma unroll
        for (int U = 0; U < 4; ++U) {
            const uint32_t bm = (U == 0) ? c0.y : (U == 1) ? c0.w : (U == 2) ? c1.y : c1.w;
            const float x = __uint_as_float((U == 0) ? c0.x : (U == 1) ? c0.z : (U == 2) ? c1.x : c1.z);
            const int col = (bm >> 16) & 0xFF;
            const f4 *vrow = lds + col * 16;
            f4 v[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) v[t] = (ABL == 1) ? f4{1.f, 2.f, 3.f, (float)col} : vrow[loff[t]];
            f2 d01 = {0, 0}, d23 = {0, 0};
            if (ABL != 2) {
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    d01 = __builtin_elementwise_fma(fu[t].xy, v[t].xy, d01);
                    d23 = __builtin_elementwise_fma(fu[t].zw, v[t].zw, d23);
                }
            } else { d01 = v[0].xy + v[5].zw; d23 = v[9].xy + v[15].zw; }
            const float den = (d01.x + d01.y) + (d23.x + d23.y);
            const bool ok = den >= 1e-10f;
            const float s = (ok && x != 0.f) ? x * __builtin_amdgcn_rcpf(den) : 0.f;
            const f2 ss = {s, s};
            if (ABL != 2) {
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const f4 w = MODE ? vrow[256 * 16 + loff[t]] : v[t];
                    acc[t].xy = __builtin_elementwise_fma(ss, w.xy, acc[t].xy);
                    acc[t].zw = __builtin_elementwise_fma(ss, w.zw, acc[t].zw);
                }
            } else acc[U].x += s;
            if (ABL == 4) reinterpret_cast<float *>(lds)[(MODE ? 2 : 1) * 256 * 16 * 4 + (bm & 0x1FFF)] = s;
            else if (ABL != 3 || s == 12345.678f) sd[bm & 0x1FFF] = s;
#pragma unroll
            for (int t = 0; t < 16; ++t) asm volatile("" : "+v"(acc[t]));
        }
    }
    float r = 0;
#pragma unroll
    for (int t = 0; t < 16; ++t) r += acc[t].x + acc[t].y + acc[t].z + acc[t].w;
    out[(size_t)blockIdx.x * 512 + tid] = r;
}

template <int MODE, int ABL> void run(const char *name, const u4 *rec, const float *F, float *sd, float *out) {
    const int blocks = 256, iters = 288, tile_it = 9;
    const size_t lds = (MODE ? 2 : 1) * 256 * 16 * 16 + (ABL == 4 ? 8192 * 4 : 0);
    (void)hipFuncSetAttribute((const void *)k<MODE, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, ABL>), dim3(blocks), dim3(512), lds, 0, rec, F, sd, out, iters, tile_it);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double slots_per_cu = 512.0 * 4 * iters;               // one group per CU
    printf("%-44s %.3f ms   %.2f ns / 16 slots / CU\n", name, ms, ms * 1e6 / slots_per_cu * 16);
}

int main() {
    const size_t nrec = (size_t)64 * 1024 * 256 + 4096;
    u4 *rec; float *F, *sd, *out;
    (void)hipMalloc(&rec, nrec * sizeof(u4)); (void)hipMalloc(&F, 0x10000 * 16 + 4096); (void)hipMalloc(&sd, 256 * WIN * 4 + 0x8000);
    (void)hipMalloc(&out, 256 * 512 * 4);
    {   // records: x = 1..5, column = pseudo-random 0..255, store offset < 8192
        u4 *h = (u4 *)malloc(nrec * sizeof(u4));
        uint32_t sdd = 12345;
        for (size_t i = 0; i < nrec; ++i) {
            uint32_t w[4];
            for (int e = 0; e < 2; ++e) {
                sdd = sdd * 1664525u + 1013904223u;
                const float x = 1.f + (sdd >> 29);
                w[2 * e] = *(uint32_t *)&x;
                w[2 * e + 1] = ((sdd >> 8) & 0xFF) << 16 | ((sdd >> 16) & 0x1FFF);
            }
            h[i] = u4{w[0], w[1], w[2], w[3]};
        }
        (void)hipMemcpy(rec, h, nrec * sizeof(u4), hipMemcpyHostToDevice);
        free(h);
        float *hf = (float *)malloc(0x10000 * 16);
        for (int i = 0; i < 0x10000 * 4; ++i) hf[i] = 0.001f * (1 + (i & 1023));
        (void)hipMemcpy(F, hf, 0x10000 * 16, hipMemcpyHostToDevice);
        free(hf);
    }
    run<0, 0>("lane per row, K = 64: full, stage, store", rec, F, sd, out);
    run<0, 1>("  no LDS reads", rec, F, sd, out);
    run<0, 2>("  no FMAs", rec, F, sd, out);
    run<0, 3>("  no scattered store", rec, F, sd, out);
    run<0, 4>("  s through an LDS window, coalesced flush", rec, F, sd, out);
    run<1, 0>("lane per row, K = 64, second image", rec, F, sd, out);
    return 0;
}

// Issue model of the dense-gene kernels: a dependent chain of v_mfma_f32_32x32x16_bf16 with NV independent single-pass
// VALU instructions (v_and / v_sub / v_perm mix, or v_rcp) and ND ds_read_b128 between consecutive matrix instructions,
// one or two waves per SIMD.  Prints cycles per matrix instruction (s_memtime) for each mix.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef uint32_t u4v __attribute__((ext_vector_type(4)));

template <int NV, int ND, int KIND>
__global__ __launch_bounds__(512) void k(float *out, unsigned long long *cyc, int iters) {
    __shared__ u4v lds[1024];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = u4v{(uint32_t)i, 1u, 2u, 3u};
    __syncthreads();
    f16v acc = {0};
    u4v a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a;
    uint32_t v[12];
    for (int e = 0; e < 12; ++e) v[e] = 0x3f800000u + lane * 977u + e;
    float f[4] = {1.5f + lane, 2.5f, 3.5f, 4.5f};
    u4v d = {0, 0, 0, 0};
    u4v ring[4] = {d, d, d, d};
    unsigned long long t0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), acc, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < NV; ++e) {
                if (KIND == 0 || KIND == 3) { uint32_t &x = v[e % 12]; x = (e & 1) ? (x & 0xFFFF0000u) + 0x10000u : __builtin_amdgcn_perm(x, v[(e + 5) % 12], 0x07060302u); }
                else if (KIND == 1) { float &x = f[e % 4]; x = x * 1.0001f + 0.5f; }
                else { float &x = f[e % 4]; x = __builtin_amdgcn_rcpf(x) + 1.0f; }
            }
            if (KIND == 3) {      // prefetched reads: the data of slot u is consumed (as the A operand) four slots later
#pragma unroll
                for (int e = 0; e < ND; ++e) { a ^= ring[(u + 4 * e) & 3]; ring[(u + 4 * e) & 3] = lds[(lane + 64 * (e + u)) & 1023]; }
            } else {
#pragma unroll
                for (int e = 0; e < ND; ++e) { const u4v r = lds[(lane + 64 * (e + u)) & 1023]; d ^= r; }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float s = acc[0] + f[0] + f[1] + f[2] + f[3] + (float)d.x + (float)(ring[0].x ^ ring[1].y ^ ring[2].z ^ ring[3].w);
    for (int e = 0; e < 12; ++e) s += (float)v[e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NV, int ND, int KIND> void run(float *out, unsigned long long *cyc, int threads, const char *what) {
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<NV, ND, KIND>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    (void)hipDeviceSynchronize();
    unsigned long long h[256]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < 256; ++i) m += (double)h[i];
    printf("%-28s NV=%2d ND=%d waves/SIMD=%d : %6.1f cycles per matrix instruction and wave\n", what, NV, ND, threads / 256, m / 256 / (iters * 8.0));
}
int main() {
    float *out; unsigned long long *cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8);
    for (int th : {256, 512}) {
        run<0, 0, 0>(out, cyc, th, "matrix only");
        run<2, 0, 0>(out, cyc, th, "and/perm"); run<4, 0, 0>(out, cyc, th, "and/perm"); run<6, 0, 0>(out, cyc, th, "and/perm");
        run<8, 0, 0>(out, cyc, th, "and/perm"); run<12, 0, 0>(out, cyc, th, "and/perm");
        run<4, 0, 1>(out, cyc, th, "v_fma_f32"); run<8, 0, 1>(out, cyc, th, "v_fma_f32");
        run<2, 0, 2>(out, cyc, th, "v_rcp_f32 + add"); run<4, 0, 2>(out, cyc, th, "v_rcp_f32 + add");
        run<0, 1, 0>(out, cyc, th, "ds_read_b128"); run<0, 2, 0>(out, cyc, th, "ds_read_b128");
        run<6, 1, 0>(out, cyc, th, "and/perm + ds_read_b128"); run<8, 1, 0>(out, cyc, th, "and/perm + ds_read_b128");
        run<0, 1, 3>(out, cyc, th, "prefetched ds_read_b128"); run<4, 1, 3>(out, cyc, th, "and/perm + prefetched read");
        run<6, 1, 3>(out, cyc, th, "and/perm + prefetched read");
    }
    return 0;
}

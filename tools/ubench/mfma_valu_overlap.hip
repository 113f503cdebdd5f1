// Do VALU instructions overlap with matrix instructions on gfx950?  Per SIMD: (a) one wave, MFMAs only; (b) one wave,
// VALU FMAs only; (c) one wave, both interleaved 1 MFMA : NV independent VALU (same wave: the in-order issue lets the
// VALU run under the matrix instruction's 16 passes if the hardware co-executes them); (d) two waves, one of each kind.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE, int NV, int KIND = 0>   // KIND 0: v_pk_fma_f32, 1: v_add_u32 / v_xor, 2: v_exp_f32;  MODE 0: MFMA only, 1: VALU only, 2: interleaved in one wave, 3: wave parity decides (two waves per SIMD)
__global__ __launch_bounds__(512) void k(float *out, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f16v a0 = {0}, a1 = a0;
    f2 vv[8]; for (int e = 0; e < 8; ++e) vv[e] = f2{1.f + lane, 2.f};
    f2 &v0 = vv[0], &v1 = vv[1], &v2 = vv[2], &v3 = vv[3], &v4 = vv[4], &v5 = vv[5], &v6 = vv[6], &v7 = vv[7];
    const f2 m = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
    const float x = 1.0f + lane * 1e-3f, y = 1.0f - lane * 1e-3f;
    const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && wave < 4);
    const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && wave >= 4);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (do_m) { if (u & 1) a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0); else a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0); }
            if (do_v) {
#pragma unroll
                for (int q = 0; q < NV / 8; ++q) {
                    if (KIND == 1) {
                        unsigned *w = reinterpret_cast<unsigned *>(&v0);
#pragma unroll
                        for (int e = 0; e < 8; ++e) { unsigned t = __float_as_uint(vv[e].x); t = (t + 0x9e3779b9u) ^ (t >> 3); vv[e].x = __uint_as_float(t); }
                        (void)w;
                        continue;
                    }
                    if (KIND == 2) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) vv[e].x = __builtin_amdgcn_exp2f(vv[e].x);
                        continue;
                    }
                    v0 = __builtin_elementwise_fma(v0, m, c); v1 = __builtin_elementwise_fma(v1, m, c);
                    v2 = __builtin_elementwise_fma(v2, m, c); v3 = __builtin_elementwise_fma(v3, m, c);
                    v4 = __builtin_elementwise_fma(v4, m, c); v5 = __builtin_elementwise_fma(v5, m, c);
                    v6 = __builtin_elementwise_fma(v6, m, c); v7 = __builtin_elementwise_fma(v7, m, c);
                }
            }
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[1] + v0.x + v1.x + v2.y + v3.x + v4.x + v5.y + v6.x + v7.x;
}
template <int MODE, int NV, int KIND = 0> float run(float *out, int threads) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, NV, KIND>), dim3(256), dim3(threads), 0, 0, out, 4000);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    return ms;
}
int main() {
    float *out; (void)hipMalloc(&out, 256 * 512 * 4);
    printf("per SIMD, 32000 matrix instructions (v_mfma_f32_32x32x2_f32) and / or 32000 x NV v_pk_fma_f32\n");
    printf("one wave,  MFMA only                    %.3f ms\n", run<0, 8>(out, 256));
    printf("one wave,  VALU only, NV = 8            %.3f ms\n", run<1, 8>(out, 256));
    printf("one wave,  VALU only, NV = 16           %.3f ms\n", run<1, 16>(out, 256));
    printf("one wave,  interleaved, NV = 8          %.3f ms\n", run<2, 8>(out, 256));
    printf("one wave,  interleaved, NV = 16         %.3f ms\n", run<2, 16>(out, 256));
    printf("two waves, one MFMA + one VALU (NV = 8)  %.3f ms\n", run<3, 8>(out, 512));
    printf("two waves, one MFMA + one VALU (NV = 16) %.3f ms\n", run<3, 16>(out, 512));
    printf("two waves, both MFMA                    %.3f ms\n", run<0, 8>(out, 512));
    printf("two waves, both interleaved, NV = 8     %.3f ms\n", run<2, 8>(out, 512));
    printf("integer VALU (2 ops per element, 8 elements): one wave alone %.3f ms, interleaved with MFMA %.3f ms, partner wave %.3f ms\n",
           run<1, 8, 1>(out, 256), run<2, 8, 1>(out, 256), run<3, 8, 1>(out, 512));
    printf("v_exp_f32 (8 per MFMA):                        one wave alone %.3f ms, interleaved with MFMA %.3f ms, partner wave %.3f ms\n",
           run<1, 8, 2>(out, 256), run<2, 8, 2>(out, 256), run<3, 8, 2>(out, 512));
    return 0;
}

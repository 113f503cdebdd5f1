// Does a wave64 whose upper (or lower) 32 lanes are inactive issue its VALU / LDS instructions in half the time?
// (gfx950 runs a wave64 VALU instruction as two passes of 32 lanes.)  Full EXEC against half EXEC, packed FMAs and
// ds_read_b128, 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <int MODE>   // 0: VALU full, 1: VALU lower half only, 2: LDS full, 3: LDS lower half only, 4: VALU lanes 0-15 only
__global__ __launch_bounds__(1024) void k(float *out, int iters) {
    __shared__ f4 lds[4096];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4096; i += 1024) lds[i] = f4{1.f, 2.f, 3.f, (float)i};
    __syncthreads();
    f2 a0 = {1.f + lane, 2.f}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;
    const f2 m = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
    f4 acc = {0, 0, 0, 0};
    const bool on = (MODE == 0 || MODE == 2) ? true : (MODE == 4 ? lane < 16 : lane < 32);
    if (on) {
        for (int it = 0; it < iters; ++it) {
            if (MODE == 0 || MODE == 1 || MODE == 4) {
                #pragma unroll
                for (int u = 0; u < 4; ++u) {
                    a0 = __builtin_elementwise_fma(a0, m, c); a1 = __builtin_elementwise_fma(a1, m, c);
                    a2 = __builtin_elementwise_fma(a2, m, c); a3 = __builtin_elementwise_fma(a3, m, c);
                    a4 = __builtin_elementwise_fma(a4, m, c); a5 = __builtin_elementwise_fma(a5, m, c);
                    a6 = __builtin_elementwise_fma(a6, m, c); a7 = __builtin_elementwise_fma(a7, m, c);
                }
            } else {
                #pragma unroll
                for (int u = 0; u < 8; ++u) acc += lds[(lane * 4 + u * 257 + it) & 4095];
            }
        }
    }
    out[blockIdx.x * 1024 + tid] = a0.x + a1.x + a2.y + a3.x + a4.x + a5.y + a6.x + a7.x + acc.x + acc.y + acc.z + acc.w;
}
template <int MODE> void run(float *out, const char *name) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, out, 4000);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%-28s %.3f ms\n", name, ms);
}
int main() {
    float *out; (void)hipMalloc(&out, 256 * 1024 * 4);
    run<0>(out, "pk_fma, all 64 lanes");
    run<1>(out, "pk_fma, lanes 0-31 only");
    run<4>(out, "pk_fma, lanes 0-15 only");
    run<2>(out, "ds_read_b128, all 64 lanes");
    run<3>(out, "ds_read_b128, lanes 0-31");
    return 0;
}

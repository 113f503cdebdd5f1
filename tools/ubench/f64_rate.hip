#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(double *out, int iters) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = threadIdx.x * 2e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_fma(double *out, int iters) {
    double x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3 + i;
    const double a = 1.0000001, b = 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = fma(x[i], a, b);
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_sig(double *out, int iters) {
    double x[4];
    for (int i = 0; i < 4; ++i) x[i] = threadIdx.x * 1e-3 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = 1.0 / (1.0 + exp(-x[i]));
    }
    double s = 0;
    for (int i = 0; i < 4; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <class F>
float timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    double *out; hipMalloc(&out, 8 * 256 * 4096);
    const int blocks = 256 * 8, iters = 20000;
    float ms = timeit([&] { hipLaunchKernelGGL(k_mfma<8>, dim3(blocks), dim3(256), 0, 0, out, iters); });
    printf("mfma f64 16x16x4, 8 acc, 8 waves/SIMD: %.2f ms  %.1f TF/s\n", ms, 2048.0 * 8 * iters * blocks * 4 / ms / 1e9);
    ms = timeit([&] { hipLaunchKernelGGL(k_mfma<8>, dim3(256), dim3(256), 0, 0, out, iters); });
    printf("mfma f64 16x16x4, 8 acc, 1 wave/SIMD:  %.2f ms  %.1f TF/s\n", ms, 2048.0 * 8 * iters * 256 * 4 / ms / 1e9);
    ms = timeit([&] { hipLaunchKernelGGL(k_mfma<2>, dim3(256), dim3(256), 0, 0, out, iters); });
    printf("mfma f64 16x16x4, 2 acc, 1 wave/SIMD:  %.2f ms  %.1f TF/s\n", ms, 2048.0 * 2 * iters * 256 * 4 / ms / 1e9);
    ms = timeit([&] { hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, out, iters); });
    printf("v_fma_f64: %.2f ms  %.1f TF/s\n", ms, 2.0 * 8 * iters * blocks * 256 / ms / 1e9);
    ms = timeit([&] { hipLaunchKernelGGL(k_sig, dim3(blocks), dim3(256), 0, 0, out, 2000); });
    printf("sigmoid f64: %.2f ms  %.2f Gelem/s -> 2e9 elements in %.2f ms\n", ms, 4.0 * 2000 * blocks * 256 / ms / 1e6, 2e9 / (4.0 * 2000 * blocks * 256 / ms));
    return 0;
}

// v_mfma_f32_32x32x2_f32 on gfx950: (1) how does a chain of S accumulating steps round -- compared against the exact
// (float64) sum and against a host chain of single-rounding fmaf() in the same order; (2) sustained rate of
// independent chains (the clock is power-limited under matrix load), next to v_mfma_f64_16x16x4_f64.
// Fragment map assumed (and checked by (1)): A[m = lane & 31][k = lane >> 5], B[k = lane >> 5][n = lane & 31],
// D reg v: [m = 8 (v / 4) + 4 (lane >> 5) + v % 4][n = lane & 31].
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void k_chain(float *C, const float *A, const float *B, int S) {      // A [32][2S], B [2S][32], one wave
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    f16v acc = {0};
    for (int s = 0; s < S; ++s)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * 2 * S + 2 * s + h], B[(2 * s + h) * 32 + r], acc, 0, 0, 0);
    for (int v = 0; v < 16; ++v) C[(8 * (v / 4) + 4 * h + v % 4) * 32 + r] = acc[v];
}

template <int F64>
__global__ __launch_bounds__(256) void k_rate(float *out, int iters) {
    const int lane = threadIdx.x & 63;
    float s = 0.f;
    if (F64) {
        d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        const double x = 1.0 + lane * 1e-3, y = 1.0 - lane * 1e-3;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
            }
        }
        s = (float)(a0[0] + a1[1] + a2[2] + a3[3]);
    } else {
        f16v a0 = {0}, a1 = a0, a2 = a0, a3 = a0;
        const float x = 1.0f + lane * 1e-3f, y = 1.0f - lane * 1e-3f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
            }
        }
        s = a0[0] + a1[1] + a2[2] + a3[3];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// one / two dependent chains per wave (1 or 2 waves per SIMD): does a chain on ONE accumulator keep the matrix pipe full?
template <int CHAINS>
__global__ __launch_bounds__(256) void k_chain_rate(float *out, int iters) {
    const int lane = threadIdx.x & 63;
    f16v a0 = {0}, a1 = a0;
    const float x = 1.0f + lane * 1e-3f, y = 1.0f - lane * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
            if (CHAINS == 2) a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1];
}

int main() {
    float *dA, *dB, *dC, *out;
    const int SMAX = 4096;
    (void)hipMalloc(&dA, 32 * 2 * SMAX * 4); (void)hipMalloc(&dB, 32 * 2 * SMAX * 4); (void)hipMalloc(&dC, 1024 * 4);
    (void)hipMalloc(&out, 2048 * 256 * 4);
    srand(7);
    for (int S : {32, 256, 2048}) {
        std::vector<float> A(32 * 2 * S), B(2 * S * 32), C(1024);
        for (auto &v : A) v = (float)rand() / RAND_MAX;                  // D_hat-like: [0, 1]
        for (auto &v : B) v = 0.01f + 3.f * (float)rand() / RAND_MAX;   // factor-like: positive
        (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, dC, dA, dB, S);
        (void)hipMemcpy(C.data(), dC, 1024 * 4, hipMemcpyDeviceToHost);
        double bias = 0, rms = 0, mx = 0;
        int same_fma = 0, same_pair = 0;
        for (int m = 0; m < 32; ++m)
            for (int n = 0; n < 32; ++n) {
                double ex = 0;
                float chain = 0.f, pair = 0.f;
                for (int k = 0; k < 2 * S; ++k) {
                    ex += (double)A[m * 2 * S + k] * (double)B[k * 32 + n];
                    chain = fmaf(A[m * 2 * S + k], B[k * 32 + n], chain);
                }
                for (int s = 0; s < S; ++s) {       // the two products of a step added exactly, one rounding per step
                    const double p = (double)A[m * 2 * S + 2 * s] * B[(2 * s) * 32 + n] + (double)A[m * 2 * S + 2 * s + 1] * B[(2 * s + 1) * 32 + n];
                    pair = (float)((double)pair + p);
                }
                const double rel = (C[m * 32 + n] - ex) / ex;
                bias += rel; rms += rel * rel; mx = fmax(mx, fabs(rel));
                same_fma += (C[m * 32 + n] == chain);
                same_pair += (C[m * 32 + n] == pair);
            }
        printf("chain of %4d steps (%4d terms): mean signed rel err %+.3e  rms %.3e  max %.3e | == fmaf chain %4d/1024, == one rounding per step %4d/1024\n",
               S, 2 * S, bias / 1024, sqrt(rms / 1024), mx, same_fma, same_pair);
    }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int f64 = 0; f64 < 2; ++f64) {
        float ms = 0;
        const int iters = 20000, blocks = 2048;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            if (f64) hipLaunchKernelGGL(k_rate<1>, dim3(blocks), dim3(256), 0, 0, out, iters);
            else hipLaunchKernelGGL(k_rate<0>, dim3(blocks), dim3(256), 0, 0, out, iters);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        const double flops = (double)blocks * 4 * iters * 16 * (f64 ? 2.0 * 16 * 16 * 4 : 2.0 * 32 * 32 * 2);
        printf("%s: %.2f ms, %.1f TFLOP/s\n", f64 ? "v_mfma_f64_16x16x4_f64 " : "v_mfma_f32_32x32x2_f32", ms, flops / ms * 1e-9);
    }
    for (int chains = 1; chains <= 2; ++chains)
        for (int wg_per_cu = 1; wg_per_cu <= 2; ++wg_per_cu) {
            float ms = 0;
            const int iters = 20000, blocks = 256 * wg_per_cu;
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipEventRecord(e0);
                if (chains == 1) hipLaunchKernelGGL(k_chain_rate<1>, dim3(blocks), dim3(256), 0, 0, out, iters);
                else hipLaunchKernelGGL(k_chain_rate<2>, dim3(blocks), dim3(256), 0, 0, out, iters);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                (void)hipEventElapsedTime(&ms, e0, e1);
            }
            const double n_mfma = (double)iters * 8 * chains;
            printf("%d dependent chain(s), %d wave(s) per SIMD: %.1f ns per matrix instruction and wave, %.1f TFLOP/s\n", chains,
                   wg_per_cu, ms * 1e6 / n_mfma, (double)blocks * 4 * n_mfma * 4096.0 / ms * 1e-9);
        }
    return 0;
}

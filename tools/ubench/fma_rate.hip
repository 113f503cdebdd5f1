#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0+4, x5=x0+5, x6=x0+6, x7=x0+7;
    f2 y0 = {x0, x1}, y1 = {x2, x3}, y2 = {x4, x5}, y3 = {x6, x7};
    f2 av = {a, a}, bv = {b, b};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
            #pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x1) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x2) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x3) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x4) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x5) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x6) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x7) : "v"(a), "v"(b));
            }
        } else {
            #pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y0) : "v"(av), "v"(bv));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y1) : "v"(av), "v"(bv));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y2) : "v"(av), "v"(bv));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y3) : "v"(av), "v"(bv));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y0) : "v"(av), "v"(bv));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y1) : "v"(av), "v"(bv));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y2) : "v"(av), "v"(bv));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y3) : "v"(av), "v"(bv));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0+x1+x2+x3+x4+x5+x6+x7 + y0.x+y0.y+y1.x+y1.y+y2.x+y2.y+y3.x+y3.y;
}
int main() {
    float* out; hipMalloc(&out, 256 * 4096 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wpc = 1; wpc <= 4; wpc *= 2) {       // blocks per CU (4 waves each)
        for (int mode = 0; mode < 2; ++mode) {
            int blocks = 256 * wpc, iters = 20000;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
                else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double instr = (double)blocks * 4 * iters * 64;   // wave-instructions
            double fmas = instr * 64 * (mode ? 2 : 1);
            printf("blocks/CU=%d mode=%s: %.3f ms, %.2f TFLOP/s, %.2f cycles per wave-instr per SIMD (at 2.4GHz)\n", wpc, mode ? "pk_fma" : "fma", ms, 2 * fmas / ms / 1e9,
                   ms * 1e-3 * 2.4e9 * 1024 / instr);
        }
    }
    return 0;
}

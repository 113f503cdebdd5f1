// Do LDS b128 reads overlap with v_pk_fma_f32 work?  Per read: NF packed FMAs on the loaded data.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
template <int NF, bool DOLDS>
__global__ __launch_bounds__(1024) void k(float* out, int iters) {
    extern __shared__ f4 lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 256 * 32; i += 1024) lds[i] = f4{1.f, 2.f, 3.f, (float)i};
    __syncthreads();
    const int q = lane & 3, Q = lane >> 2;
    const int rot = ((Q & 1) << 1) | ((Q >> 1) & 1);
    int choff[8];
    #pragma unroll
    for (int t = 0; t < 8; ++t) choff[t] = ((t + rot) % 8) * 4 + q;
    unsigned seed = (tid >> 2) * 2654435761u + 12345u;
    f4 acc[8];
    #pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = f4{0, 0, 0, 0};
    f4 vconst = {1.f, 0.5f, 0.25f, 2.f};
    for (int it = 0; it < iters; ++it) {
        seed = seed * 1664525u + 1013904223u;
        const int row = (seed >> 24);
        const f4* vrow = lds + row * 32;
        const float sv = __uint_as_float((seed & 0x007FFFFF) | 0x3F000000);
        const f2 ss = {sv, sv};
        #pragma unroll
        for (int t = 0; t < 8; ++t) {
            f4 v = DOLDS ? vrow[choff[t]] : vconst;
            #pragma unroll
            for (int f = 0; f < NF / 2; ++f) {
                acc[t].xy = __builtin_elementwise_fma(ss, v.xy, acc[t].xy);
                acc[t].zw = __builtin_elementwise_fma(ss, v.zw, acc[t].zw);
            }
        }
        if (!DOLDS) asm volatile("" : "+v"(vconst));
    }
    float r = 0;
    #pragma unroll
    for (int t = 0; t < 8; ++t) r += acc[t].x + acc[t].y + acc[t].z + acc[t].w;
    out[blockIdx.x * 1024 + tid] = r;
}
template <int NF, bool DOLDS>
void run(float* out, hipEvent_t e0, hipEvent_t e1, int iters) {
    (void)hipFuncSetAttribute((const void*)k<NF, DOLDS>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<NF, DOLDS>), dim3(256), dim3(1024), 128 * 1024, 0, out, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    printf("NF=%d pk_fma per read, LDS=%d: %.3f ms (%.2f ns per 8-read step per wave-slot)\n", NF, (int)DOLDS, ms, ms * 1e6 / iters / 1.0);
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 1024 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 4000;
    run<2, true>(out, e0, e1, iters);  run<2, false>(out, e0, e1, iters);
    run<4, true>(out, e0, e1, iters);  run<4, false>(out, e0, e1, iters);
    run<0, true>(out, e0, e1, iters);
    return 0;
}

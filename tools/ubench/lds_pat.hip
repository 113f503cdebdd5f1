// LDS read-pattern microbenchmark: quads read 64 contiguous bytes (4 lanes x ds_read_b128) of
// pseudo-random 512-byte rows; NCH chunks per row; chunk order rotated per quad by a pattern.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int rot_of(int pat, int lane) {
    const int Q = lane >> 2;
    switch (pat) {
        case 0: return 0;
        case 1: return (Q & 7) >> 1;
        case 2: return Q & 3;
        case 3: return (Q >> 2) & 3;
        case 4: return ((Q & 3) + (Q >> 2)) & 3;
        case 5: return Q & 7;
        case 6: return (Q >> 1) & 7;
        case 7: return ((Q & 1) << 1) | ((Q >> 1) & 1);
        case 8: return (Q ^ (Q >> 2)) & 3;
        case 9: return ((Q & 3) * 2 + ((Q >> 2) & 1)) & 7;
    }
    return 0;
}
template <int NCH>
__global__ __launch_bounds__(1024) void k(float* out, int iters, int pat, int samerow) {
    extern __shared__ f4 lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 256 * 32; i += 1024) lds[i] = f4{1.f, 2.f, 3.f, (float)i};
    __syncthreads();
    const int q = lane & 3;
    const int rot = rot_of(pat, lane);
    int choff[NCH];
    #pragma unroll
    for (int t = 0; t < NCH; ++t) choff[t] = ((t + rot) % NCH) * 4 + q;
    unsigned seed = (tid >> 2) * 2654435761u + 12345u;
    f4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        seed = seed * 1664525u + 1013904223u;
        const int row = samerow ? 0 : (seed >> 24);
        const f4* vrow = lds + row * 32;
        #pragma unroll
        for (int t = 0; t < NCH; ++t) { f4 v = vrow[choff[t]]; acc += v; }
    }
    out[blockIdx.x * 1024 + tid] = acc.x + acc.y + acc.z + acc.w;
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 1024 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute((const void*)k<7>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    (void)hipFuncSetAttribute((const void*)k<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    const int iters = 4000;
    for (int nch = 7; nch <= 8; ++nch)
        for (int pat = -1; pat < 10; ++pat) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipEventRecord(e0);
                if (nch == 7) hipLaunchKernelGGL(k<7>, dim3(256), dim3(1024), 128 * 1024, 0, out, iters, pat < 0 ? 0 : pat, pat < 0);
                else hipLaunchKernelGGL(k<8>, dim3(256), dim3(1024), 128 * 1024, 0, out, iters, pat < 0 ? 0 : pat, pat < 0);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                (void)hipEventElapsedTime(&ms, e0, e1);
            }
            double reads_per_cu = 16.0 * iters * nch;      // wave-level ds_read_b128 per CU
            printf("NCH=%d pat=%2d: %.3f ms -> %.2f cycles per ds_read_b128 per CU (2.4GHz)\n", nch, pat, ms, ms * 1e-3 * 2.4e9 / reads_per_cu);
        }
    return 0;
}

// LDS gather microbenchmark, row stride variants: quads read 6 x 64 contiguous bytes (4 lanes x
// ds_read_b128) of pseudo-random rows; rows `stride4` float4 apart; with / without the per-quad chunk
// rotation of the pass kernels.  Question: does an unpadded 400-byte stride (K = 100), which spreads the
// rows over 16 bank alignments, make the rotation (and its per-read address arithmetic) unnecessary?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int NCH, int ROT>
__global__ __launch_bounds__(1024) void k(float* out, int iters, int stride4) {
    extern __shared__ f4 lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 256 * stride4; i += 1024) lds[i] = f4{1.f, 2.f, 3.f, (float)i};
    __syncthreads();
    const int q = lane & 3, Q = lane >> 2;
    const int rot = ROT ? (((Q & 1) << 1) | ((Q >> 1) & 1)) : 0;
    int choff[NCH];
    #pragma unroll
    for (int t = 0; t < NCH; ++t) choff[t] = ((t + rot) % NCH) * 4 + q;
    unsigned seed = (tid >> 2) * 2654435761u + 12345u;
    f4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        seed = seed * 1664525u + 1013904223u;
        const int row = (seed >> 24);
        const f4* vrow = lds + row * stride4;
        #pragma unroll
        for (int t = 0; t < NCH; ++t) { f4 v = ROT ? vrow[choff[t]] : vrow[t * 4 + q]; acc += v; }
    }
    out[blockIdx.x * 1024 + tid] = acc.x + acc.y + acc.z + acc.w;
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 1024 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute((const void*)k<6, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    (void)hipFuncSetAttribute((const void*)k<6, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    const int iters = 4000;
    const int strides[] = {32, 25, 26, 27, 28, 29, 31, 33};
    for (int rot = 0; rot < 2; ++rot)
        for (int s : strides) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipEventRecord(e0);
                if (rot) hipLaunchKernelGGL((k<6, 1>), dim3(256), dim3(1024), 140 * 1024, 0, out, iters, s);
                else hipLaunchKernelGGL((k<6, 0>), dim3(256), dim3(1024), 140 * 1024, 0, out, iters, s);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                (void)hipEventElapsedTime(&ms, e0, e1);
            }
            printf("rot=%d stride=%3d B: %.3f ms -> %.2f cycles per ds_read_b128 per CU (2.4 GHz)\n", rot, s * 16, ms,
                   ms * 1e-3 * 2.4e9 / (16.0 * iters * 6));
        }
    return 0;
}

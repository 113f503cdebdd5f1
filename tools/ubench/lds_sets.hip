// Which lanes of a ds_read_b128 are serviced together?  All quads read chunk 0 of image row 0 (one address:
// broadcast) except quads Qa and Qb, which read chunk 0 of two OTHER rows (same banks, different
// addresses).  If Qa and Qb (and the broadcast) share a service set they serialise; otherwise they do not.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(1024) void k(float* out, int iters, int qa, int qb, int nq) {
    extern __shared__ f4 lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 256 * 32; i += 1024) lds[i] = f4{1.f, 2.f, 3.f, (float)i};
    __syncthreads();
    const int q = lane & 3, Q = lane >> 2;
    int row = 0;
    if (nq == 2) { if (Q == qa) row = 1; if (Q == qb) row = 2; }
    else if (nq > 2) { if ((Q % nq) == qa) row = 1 + Q; }       // every nq-th quad reads its own row
    f4 acc = {0, 0, 0, 0};
    // rows are 512 bytes apart: every read below starts at bank 0 whatever the row
    for (int it = 0; it < iters; ++it) {
        const f4* p = lds + ((row * 8 + (it & 7) * 64) & 255) * 32 + q;
        f4 v[8];
        #pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = p[((t * 5) & 7) * 32];
        #pragma unroll
        for (int t = 0; t < 8; ++t) acc += v[t];
    }
    out[blockIdx.x * 1024 + tid] = acc.x + acc.y + acc.z + acc.w;
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 1024 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    const int iters = 2000;
    auto run = [&](int qa, int qb, int nq) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256), dim3(1024), 128 * 1024, 0, out, iters, qa, qb, nq);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        return ms * 1e-3 * 2.4e9 / (16.0 * iters * 8);
    };
    printf("all broadcast: %.2f cycles per ds_read_b128 per CU\n", run(-1, -1, 2));
    for (int qb = 1; qb < 16; ++qb) printf("quads 0 and %2d on other rows: %.2f\n", qb, run(0, qb, 2));
    printf("quads 4 and 5: %.2f   quads 4 and 8: %.2f   quads 5 and 9: %.2f   quads 3 and 4: %.2f\n", run(4, 5, 2), run(4, 8, 2), run(5, 9, 2), run(3, 4, 2));
    for (int nq : {16, 8, 4}) printf("every %2d-th quad on its own row: %.2f\n", nq, run(0, 0, nq));
    return 0;
}

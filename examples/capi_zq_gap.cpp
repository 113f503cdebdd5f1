// capi_zq_gap.cpp -- the drop-in boundary used from plain C++ / HIP, no Python and no PyTorch:
// reads log_U_hat (n, K), log_V_hat (m, K), X (n, m) float32 from a file, calls oriana_zq_gap_f32
// (the C-ABI replacement of GaP.compute_Z_q_expectations, reference oriana/models/gap.py:67-80) and
// writes Z_hat_i (n, K), Z_hat_j (m, K).
//
//   hipcc --offload-arch=gfx950 -O2 -Iinclude examples/capi_zq_gap.cpp -Loriana_amd/csrc -loriana_hip \
//         -Wl,-rpath,$PWD/oriana_amd/csrc -o capi_zq_gap
//   ./capi_zq_gap in.bin out.bin          (in.bin: int64 n, m, K, then the three matrices)
//
// [r5] ./capi_zq_gap in.bin out.bin REPS [DENSE_DENSITY]: the RESIDENT handle instead -- the reference calls the nest once
// per step() with the same X (gap.py:89-94): oriana_counts_create_dense_f32 packs X ONCE (sliced layout, or hybrid with the
// genes expressed in >= DENSE_DENSITY of the cells on the matrix cores), oriana_zq_gap_resident runs REPS times; out.bin gets
// the outputs of the first and of the last call, stdout the time per call.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <vector>
#include "oriana_hip.h"

#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e)); return 2; } } while (0)

int main(int argc, char **argv) {
    if (argc < 3 || argc > 5) { fprintf(stderr, "usage: %s in.bin out.bin [reps [dense_density]]\n", argv[0]); return 1; }
    const int reps = argc > 3 ? atoi(argv[3]) : 0;
    const double dense_density = argc > 4 ? atof(argv[4]) : 0.0;
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    int64_t dims[3];
    if (fread(dims, sizeof(int64_t), 3, f) != 3) return 1;
    const int64_t n = dims[0], m = dims[1], K = dims[2];
    std::vector<float> lu(n * K), lv(m * K), X(n * m), Zi(n * K), Zj(m * K);
    if (fread(lu.data(), 4, lu.size(), f) != lu.size() || fread(lv.data(), 4, lv.size(), f) != lv.size() ||
        fread(X.data(), 4, X.size(), f) != X.size()) return 1;
    fclose(f);

    int64_t nnz = 0;
    for (float v : X) nnz += (v != 0.f);
    float *d_lu, *d_lv, *d_X, *d_Zi, *d_Zj;
    void *ws;
    CHECK(hipMalloc(&d_lu, lu.size() * 4)); CHECK(hipMalloc(&d_lv, lv.size() * 4)); CHECK(hipMalloc(&d_X, X.size() * 4));
    CHECK(hipMalloc(&d_Zi, Zi.size() * 4)); CHECK(hipMalloc(&d_Zj, Zj.size() * 4));
    const int64_t ws_bytes = oriana_zq_workspace_bytes(n, m, K, nnz + 64);
    CHECK(hipMalloc(&ws, (size_t)ws_bytes));                       // hipMalloc is 256-byte aligned
    CHECK(hipMemcpy(d_lu, lu.data(), lu.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_lv, lv.data(), lv.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_X, X.data(), X.size() * 4, hipMemcpyHostToDevice));
    hipStream_t stream;
    CHECK(hipStreamCreate(&stream));
    if (reps > 0) {
        // create once ...
        oriana_resident *h = nullptr;
        int rc = oriana_counts_create_dense_f32(&h, d_X, n, m, m, K, dense_density, stream);
        if (rc) { fprintf(stderr, "oriana_counts_create_dense_f32 failed: %d\n", rc); return 3; }
        CHECK(hipFree(d_X));                                       // the handle owns the packed counts: X itself can go
        int64_t info[13];
        if ((rc = oriana_counts_info(h, info, 13))) return 3;
        f = fopen(argv[2], "wb");
        if (!f) { perror(argv[2]); return 1; }
        // ... call many times (outputs first, then log_U_hat, log_V_hat: the reference's order without X)
        std::chrono::steady_clock::time_point t0;
        for (int r = 0; r < reps; ++r) {
            if (r == 1) { CHECK(hipStreamSynchronize(stream)); t0 = std::chrono::steady_clock::now(); }
            if ((rc = oriana_zq_gap_resident(h, d_Zi, d_Zj, d_lu, d_lv, stream))) { fprintf(stderr, "oriana_zq_gap_resident failed: %d\n", rc); return 3; }
            if (r == 0 || r == reps - 1) {
                CHECK(hipStreamSynchronize(stream));
                CHECK(hipMemcpy(Zi.data(), d_Zi, Zi.size() * 4, hipMemcpyDeviceToHost));
                CHECK(hipMemcpy(Zj.data(), d_Zj, Zj.size() * 4, hipMemcpyDeviceToHost));
                fwrite(Zi.data(), 4, Zi.size(), f);
                fwrite(Zj.data(), 4, Zj.size(), f);
            }
        }
        fclose(f);
        const double ms = reps > 1 ? std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / (reps - 1) : 0.0;
        printf("%s: resident n=%lld m=%lld K=%lld nnz=%lld dense_genes=%lld col_work_items=%lld cus=%lld bytes=%lld calls=%d ms_per_call=%.4f ok\n",
               oriana_version(), (long long)info[0], (long long)info[1], (long long)info[2], (long long)info[4], (long long)info[5],
               (long long)info[9], (long long)info[12], (long long)info[8], reps, ms);
        return oriana_counts_destroy(h);
    }
    // same argument order as the reference: outputs first, then log_U_hat, log_V_hat, X
    const int rc = oriana_zq_gap_f32(d_Zi, d_Zj, d_lu, d_lv, d_X, n, m, K, ws, ws_bytes, stream);
    if (rc) { fprintf(stderr, "oriana_zq_gap_f32 failed: %d\n", rc); return 3; }
    CHECK(hipStreamSynchronize(stream));
    CHECK(hipMemcpy(Zi.data(), d_Zi, Zi.size() * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(Zj.data(), d_Zj, Zj.size() * 4, hipMemcpyDeviceToHost));
    f = fopen(argv[2], "wb");
    if (!f) { perror(argv[2]); return 1; }
    fwrite(Zi.data(), 4, Zi.size(), f);
    fwrite(Zj.data(), 4, Zj.size(), f);
    fclose(f);
    printf("%s: n=%lld m=%lld K=%lld nnz=%lld ok\n", oriana_version(), (long long)n, (long long)m, (long long)K, (long long)nnz);
    return 0;
}

// capi_zq_gap.cpp -- the drop-in boundary used from plain C++ / HIP, no Python and no PyTorch:
// reads log_U_hat (n, K), log_V_hat (m, K), X (n, m) float32 from a file, calls oriana_zq_gap_f32
// (the C-ABI replacement of GaP.compute_Z_q_expectations, reference oriana/models/gap.py:67-80) and
// writes Z_hat_i (n, K), Z_hat_j (m, K).
//
//   hipcc --offload-arch=gfx950 -O2 -Iinclude examples/capi_zq_gap.cpp -Loriana_amd/csrc -loriana_hip \
//         -Wl,-rpath,$PWD/oriana_amd/csrc -o capi_zq_gap
//   ./capi_zq_gap in.bin out.bin          (in.bin: int64 n, m, K, then the three matrices)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include "oriana_hip.h"

#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e)); return 2; } } while (0)

int main(int argc, char **argv) {
    if (argc != 3) { fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]); return 1; }
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

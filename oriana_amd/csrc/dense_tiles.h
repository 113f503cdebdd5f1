// dense_tiles.h -- pieces shared by the matrix-core kernels over dense 32 x 32 tiles (dense_pass.hip: the dense genes of
// pCMF's responsibility pass; dense_zi.hip: the D update and D_hat^T U_hat of the ZI models for 64 < K <= 100): bf16 x 3
// splits, the six-product macro, accumulator geometry, operand image layout, LDS-DMA copies.
#pragma once
#include "common.h"

namespace oriana {
namespace dn {

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef uint32_t u4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f16v mfma_b16(u4v a, u4v b, f16v c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
}

// two floats -> their (hi, mid, lo) bf16 parts, each pair packed in one dword (x0 in the low half)
__device__ __forceinline__ void split2(float x0, float x1, uint32_t &hi, uint32_t &mid, uint32_t &lo) {
    const uint32_t b0 = __float_as_uint(x0), b1 = __float_as_uint(x1);
    const float r0 = x0 - __uint_as_float(b0 & 0xFFFF0000u), r1 = x1 - __uint_as_float(b1 & 0xFFFF0000u);
    const uint32_t c0 = __float_as_uint(r0), c1 = __float_as_uint(r1);
    const float s0 = r0 - __uint_as_float(c0 & 0xFFFF0000u), s1 = r1 - __uint_as_float(c1 & 0xFFFF0000u);
    hi = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
    mid = __builtin_amdgcn_perm(c1, c0, 0x07060302u);
    lo = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
}

__device__ __forceinline__ void split8(const float (&x)[8], u4v (&o)[3]) {
#pragma unroll
    for (int w2 = 0; w2 < 4; ++w2) {
        uint32_t a, b, c2;
        split2(x[2 * w2], x[2 * w2 + 1], a, b, c2);
        o[0][w2] = a; o[1][w2] = b; o[2][w2] = c2;
    }
}

// the six cross products, small terms first
#define ORIANA_DN_MF6(ACC, A, B)                                                                       \
    do {                                                                                               \
        ACC = mfma_b16(A[2], B[0], ACC); ACC = mfma_b16(A[0], B[2], ACC); ACC = mfma_b16(A[1], B[1], ACC); \
        ACC = mfma_b16(A[1], B[0], ACC); ACC = mfma_b16(A[0], B[1], ACC); ACC = mfma_b16(A[0], B[0], ACC); \
    } while (0)

// row of the accumulator register v in lane half h (v_mfma_f32_32x32x*: D reg v = [8 (v / 4) + 4 h + v % 4][lane & 31])
__device__ __forceinline__ int acc_row(int v, int h) { return 8 * (v >> 2) + 4 * h + (v & 3); }

constexpr int TS = 36;            // row stride of the transpose buffer (floats): 16-byte aligned rows
constexpr int NW = 8;             // waves per work-group

// Operand images, in 16-byte pieces [..][split][lane]: a wave-wide copy of 64 consecutive pieces is one LDS-DMA
// instruction (global_load_lds_dwordx4 writes lane x 16 bytes from a wave-uniform base).
//   first image  (A operand of den^T = FV FU^T)   [k chunk][split][lane = 32 G' + g]: F[g][16 kc + 8 G' .. + 7]
//   second image (B operand of the accumulation)  [n tile][instruction q][split][lane = 32 hh + cc]:
//                F[acc_row(8 q + e, hh)][32 nt + cc], e = 0..7 -- the rows in the order the accumulator registers of
//                the first product hold them
//   tail         32 rows x float4 (factors 16 KC .. 16 KC + 3), plain float32; then the same values as
//                [lane half][tail factor][16 rows in accumulator order]: the B operand of the 4 x 4 x 1 float32 instructions
//                that accumulate the tail factors (lane l = 4 b + j of block b supplies B[b][j])
template <int KC, int TAIL>
struct Cfg {
    static constexpr int NT = (KC + 1) / 2;
    static constexpr int P1 = KC * 3 * 64;
    static constexpr int P2 = NT * 2 * 3 * 64;
    static constexpr int PT = TAIL ? 64 : 0;
    static constexpr int PV_RAW = P1 + P2 + PT;                          // gene side: both images + tail
    static constexpr int PU_RAW = P2 + PT;                               // cell side: second image + tail
    static constexpr int PV = (PV_RAW + NW * 64 - 1) / (NW * 64) * (NW * 64);   // every wave copies the same number of pieces
    // [r6] the gene-side copy of dn::k_zi_row: always two 64-piece slots of padding behind the images (the tile's non-zero flags
    // and logits ride there; KC = 4 without tail had none)
    static constexpr int PVZ = (PV_RAW + 2 * 64 + NW * 64 - 1) / (NW * 64) * (NW * 64);
    static constexpr int PU = (PU_RAW + NW * 64 - 1) / (NW * 64) * (NW * 64);
    static constexpr int KM = 16 * KC;                                    // factors on the matrix core
};

// LDS-DMA copy of one image (P pieces, a multiple of 8 x 64) by the 8 waves of a work-group
template <int P>
__device__ __forceinline__ void image_dma(const u4v *__restrict__ src, u4v *dst_lds, int wave, int lane) {
#pragma unroll
    for (int p = 0; p < P / (NW * 64); ++p) {
        const int piece = (p * NW + wave) * 64;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + piece + lane),
                                         (__attribute__((address_space(3))) void *)(dst_lds + piece), 16, 0, 0);
    }
}

}  // namespace dn
}  // namespace oriana

// common.h -- device helpers shared by the gfx950 kernels of the CAVI engine.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <float.h>
#include "../../include/oriana_hip.h"

#define ORIANA_HIP_CHECK(expr)                                     \
    do {                                                           \
        hipError_t _e = (expr);                                    \
        if (_e != hipSuccess) return -1000 - (int)_e;              \
    } while (0)

#define ORIANA_LAUNCH_CHECK()                                      \
    do {                                                           \
        hipError_t _e = hipGetLastError();                         \
        if (_e != hipSuccess) return -1000 - (int)_e;              \
    } while (0)

namespace oriana {

constexpr int TILE = ORIANA_TILE;          // 256 x 256 count tiles
constexpr float DEN_MIN = 1e-10f;          // below this the shifted softmax denominator is not trusted
constexpr float SHIFT_MAX = 22.0f;         // |row shift| above this -> exact slow path for the row
constexpr float FILL = 1e-30f;             // factor value of such rows: den <= 256 * 1e-30 < DEN_MIN, but never 0
constexpr float STAT_MAX = 200.0f;        // row maxima beyond +-this are left out of the statistics of the centred validity test (passes.hip)
constexpr float DEAD_MAX = 40.0f;          // logs of a fully masked row below this: exp(lu + lv) cannot overflow (22 + 40 < 88)

// ---- DPP cross-lane moves (wave64; no LDS traffic) ------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
// broadcast lane U of every quad to the whole quad
template <int U>
__device__ __forceinline__ uint32_t quad_bcast_u32(uint32_t v) { return dpp_u32<U * 0x55>(v); }
template <int U>
__device__ __forceinline__ float quad_bcast_f32(float v) { return dpp_f32<U * 0x55>(v); }

// sum over aligned groups of G lanes (G = 4, 8, 16); every lane of the group gets the total
template <int G>
__device__ __forceinline__ float group_sum(float v) {
    v += dpp_f32<0xB1>(v);            // quad_perm [1,0,3,2]
    v += dpp_f32<0x4E>(v);            // quad_perm [2,3,0,1]
    if (G >= 8)  v += dpp_f32<0x141>(v);   // row_half_mirror: lane i <-> 7 - i
    if (G >= 16) v += dpp_f32<0x140>(v);   // row_mirror: lane i <-> 15 - i
    return v;
}

// max over aligned groups of G lanes (G = 4, 8, 16)
template <int G>
__device__ __forceinline__ float group_max(float v) {
    v = fmaxf(v, dpp_f32<0xB1>(v));
    v = fmaxf(v, dpp_f32<0x4E>(v));
    if (G >= 8)  v = fmaxf(v, dpp_f32<0x141>(v));
    if (G >= 16) v = fmaxf(v, dpp_f32<0x140>(v));
    return v;
}

// [r6] Cross-lane sums without the LDS crossbar.  __shfl_xor compiles to ds_bpermute_b32: an LDS instruction (address VGPR, the
// LDS counter, ~100 cycles of latency in a dependent chain) -- three rounds of four of them per tile were worth 20 % of the ZI D
// update (profiles/r06_zi_row_ablations.txt).  gfx950 does the same moves in the vector ALU:
//   lane ^ 8    row_ror:8 inside each row of 16 lanes, folded into the addition (v_add_f32_dpp)
//   lane ^ 16   v_permlane16_swap_b32: {x, x} -> {rows 0 0 2 2, rows 1 1 3 3}; their sum is x[l] + x[l ^ 16]
//   lane ^ 32   v_permlane32_swap_b32: {x, x} -> {lower half twice, upper half twice}
// (b is made an OPAQUE copy of a -- a one-instruction asm the optimiser cannot see through: with both operands the same SSA value
//  __builtin_amdgcn_permlane16_swap / 32_swap of ROCm 7.2's hipcc returns its first result twice (v_add_f32 v2, v2, v2 after the
//  swap).  The builtin, not inline assembly, issues the swap: the compiler then inserts the wait states of the gfx950 hazards
//  around it itself -- a hand-written swap inside an asm string corrupted the 16-byte stores issued just before it.)
typedef unsigned oriana_u2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void swap_rows16(uint32_t &a, uint32_t &b) {
    const oriana_u2v r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    a = r.x; b = r.y;
}
__device__ __forceinline__ void swap_halves32(uint32_t &a, uint32_t &b) {
    const oriana_u2v r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r.x; b = r.y;
}
// (the s_nop 1 after the move: "VALU write of a swap operand -> v_permlane*_swap reads it" needs two wait states, and the
//  compiler's hazard recogniser does not look inside an asm string)
__device__ __forceinline__ uint32_t opaque_copy(uint32_t a) { uint32_t b; asm("v_mov_b32 %0, %1\n\ts_nop 1" : "=v"(b) : "v"(a)); return b; }
__device__ __forceinline__ float add_xor8(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));
}
__device__ __forceinline__ float add_xor16(float v) {
    uint32_t a = __builtin_bit_cast(uint32_t, v), b = opaque_copy(a);
    swap_rows16(a, b);
    return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
__device__ __forceinline__ float add_xor32(float v) {
    uint32_t a = __builtin_bit_cast(uint32_t, v), b = opaque_copy(a);
    swap_halves32(a, b);
    return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
// sum over the lanes with the same lane % 8 (the eight 8-lane groups of a wave), in every lane
__device__ __forceinline__ float sum_mod8(float v) { return add_xor32(add_xor16(add_xor8(v))); }

// max / bitwise OR over aligned groups of LPR lanes (8, 16, 32, 64), in every lane of the group: DPP inside a row of 16,
// v_permlane16_swap / v_permlane32_swap across rows -- no LDS instruction (k_gamma_update_vec: one reduction per row of cells)
template <int LPR>
__device__ __forceinline__ float lanes_max(float v) {
    v = fmaxf(v, dpp_f32<0xB1>(v));
    v = fmaxf(v, dpp_f32<0x4E>(v));
    v = fmaxf(v, dpp_f32<0x141>(v));
    if (LPR >= 16) v = fmaxf(v, dpp_f32<0x140>(v));
    if (LPR >= 32) {
        uint32_t a = __builtin_bit_cast(uint32_t, v), b = opaque_copy(a);
        swap_rows16(a, b);
        v = fmaxf(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b));
    }
    if (LPR >= 64) {
        uint32_t a = __builtin_bit_cast(uint32_t, v), b = opaque_copy(a);
        swap_halves32(a, b);
        v = fmaxf(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b));
    }
    return v;
}
template <int LPR>
__device__ __forceinline__ uint32_t lanes_or(uint32_t v) {
    v |= dpp_u32<0xB1>(v);
    v |= dpp_u32<0x4E>(v);
    v |= dpp_u32<0x141>(v);
    if (LPR >= 16) v |= dpp_u32<0x140>(v);
    if (LPR >= 32) { uint32_t b = opaque_copy(v); swap_rows16(v, b); v |= b; }
    if (LPR >= 64) { uint32_t b = opaque_copy(v); swap_halves32(v, b); v |= b; }
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ float wave_sum(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- special functions, float64 (scipy.special.digamma / polygamma(1, .) restated) ------------
// digamma for x > 0: upward recurrence to x >= 10, then the asymptotic series.
__device__ __forceinline__ double digamma_pos(double x) {
    double r = 0.0;
    while (x < 10.0) { r -= 1.0 / x; x += 1.0; }
    const double z = 1.0 / (x * x);
    const double y = z * (1.0 / 12.0 - z * (1.0 / 120.0 - z * (1.0 / 252.0 - z * (1.0 / 240.0 -
                     z * (1.0 / 132.0 - z * (691.0 / 32760.0 - z * (1.0 / 12.0)))))));
    return r + log(x) - 0.5 / x - y;
}

__device__ inline double digamma_f64(double x) {
    if (x != x) return x;
    if (x == INFINITY) return x;
    if (x == -INFINITY) return NAN;
    if (x == 0.0) return copysign(INFINITY, -x);
    if (x < 0.0) {
        if (x == floor(x)) return NAN;
        // reflection: psi(x) = psi(1 - x) - pi / tan(pi x)
        double fr = x - floor(x);
        return digamma_pos(1.0 - x) - M_PI / tan(M_PI * fr);
    }
    return digamma_pos(x);
}

__device__ __forceinline__ double trigamma_pos(double x) {
    double r = 0.0;
    while (x < 10.0) { r += 1.0 / (x * x); x += 1.0; }
    const double ix = 1.0 / x;
    const double z = ix * ix;
    // 1/x + 1/(2x^2) + sum B_2k / x^(2k+1)
    const double s = z * ix * (1.0 / 6.0 - z * (1.0 / 30.0 - z * (1.0 / 42.0 - z * (1.0 / 30.0 -
                     z * (5.0 / 66.0 - z * (691.0 / 2730.0 - z * (7.0 / 6.0)))))));
    return r + ix + 0.5 * z + s;
}

__device__ inline double trigamma_f64(double x) {
    if (x != x) return x;
    if (x == INFINITY) return 0.0;
    if (x == -INFINITY) return NAN;
    if (x <= 0.0) {
        if (x == floor(x)) return INFINITY;
        double fr = x - floor(x);
        double sn = sin(M_PI * fr);
        return (M_PI * M_PI) / (sn * sn) - trigamma_pos(1.0 - x);
    }
    return trigamma_pos(x);
}

// oriana/utils.py:39-51
__device__ inline double inverse_digamma_f64(double y) {
    const double psi1 = -0.5772156649015329;   // digamma(1)
    double x = (y >= -2.22) ? exp(y) + 0.5 : -1.0 / (y - psi1);
    // (five Newton steps, utils.py:47-50; a step that no longer moves x ends the loop -- the remaining ones would
    //  reproduce it: each costs a dozen dependent float64 divisions, and the M-step is a single short launch)
    #pragma unroll 1
    for (int it = 0; it < 5; ++it) {
        const double xn = x - (digamma_f64(x) - y) / trigamma_f64(x);
        const bool same = xn == x;
        x = xn;
        if (same) break;
    }
    return x;
}

// np.nan_to_num followed by np.maximum(1e-15, .)  (gap.py:99-100)
__device__ __forceinline__ double nan_to_num(double v) {
    if (v != v) return 0.0;
    if (v == INFINITY) return DBL_MAX;
    if (v == -INFINITY) return -DBL_MAX;
    return v;
}
__device__ __forceinline__ double clamp_eps(double v) { return fmax(1e-15, nan_to_num(v)); }

// oriana/utils.py:9-15
__device__ __forceinline__ double sigmoid_f64(double x) { return 1.0 / (1.0 + exp(-x)); }
__device__ __forceinline__ double logit_f64(double x) {
    if (x == x) x = fmin(fmax(x, 1e-15), 1.0 - 1e-15);   // np.clip keeps NaN
    return log(x / (1.0 - x));
}

// 1 / x for a normal positive double: v_rcp_f64 + two Newton steps (relative error < 1e-15; the compiler's division adds
// the scaling and the fix-up for operands this path never sees: twice the instructions)
__device__ __forceinline__ double rcp_newton(double x) {
    double r = __builtin_amdgcn_rcp(x);
    double e = fma(-x, r, 1.0);
    r = fma(r, e, r);
    e = fma(-x, r, 1.0);
    return fma(r, e, r);
}

// log x for a finite x >= 1: x = 2^e m with m in [sqrt(1/2), sqrt(2)), log m = 2 atanh t, t = (m - 1) / (m + 1), |t| <= 0.1716,
// eight terms of the series (the next one is below 1.2e-14 absolute).  The library's log carries its result in two doubles
// (~230 float64 additions in the update kernel); here the value is rounded to float32 a few operations later.
__device__ __forceinline__ double log_ge1_for_f32(double x) {
    double m = __builtin_amdgcn_frexp_mant(x);                 // [0.5, 1)
    int e = __builtin_amdgcn_frexp_exp(x);
    const bool lo = m < 0.70710678118654752;
    m = lo ? m + m : m;
    e = lo ? e - 1 : e;
    const double t = (m - 1.0) * rcp_newton(m + 1.0);
    const double z = t * t;
    double p = 1.0 / 15.0;
    p = fma(p, z, 1.0 / 13.0); p = fma(p, z, 1.0 / 11.0); p = fma(p, z, 1.0 / 9.0); p = fma(p, z, 1.0 / 7.0);
    p = fma(p, z, 1.0 / 5.0);  p = fma(p, z, 1.0 / 3.0);  p = fma(p, z, 1.0);
    return fma((double)e, 0.69314718055994531, (t + t) * p);
}

// digamma for x > 0 whose result is rounded to float32 afterwards (the bulk E[log .] path).  [r6] Branch-free: below 6 the
// recurrence takes six steps at once, sum_{i < 6} 1 / (x + i) = D'(x) / D(x) with D(x) = x (x + 1) .. (x + 5) (all terms
// positive: no cancellation), one reciprocal instead of up to three data-dependent divisions; then the asymptotic series at
// x + 6 >= 6 (truncation error < 2e-13, far below half a float32 ulp) with one reciprocal for 1 / x and 1 / x^2 and the short
// logarithm above.  Against scipy.special.digamma on 2e6 float32 arguments in (0.01, 20): |error| <= 1.9e-13, one result of
// 2e6 differs after rounding to float32 (by one ulp) -- the same as the loop it replaces (tools/digamma_check.py).
__device__ __forceinline__ double digamma_pos_for_f32(double x) {
    const bool small = x < 6.0;
    const double D = x * (x + 1.0) * (x + 2.0) * (x + 3.0) * (x + 4.0) * (x + 5.0);
    const double Dp = fma(fma(fma(fma(fma(6.0, x, 75.0), x, 340.0), x, 675.0), x, 548.0), x, 120.0);
    const double q = small ? Dp * rcp_newton(D) : 0.0;
    const double xs = small ? x + 6.0 : x;
    const double ix = rcp_newton(xs);
    const double z = ix * ix;
    const double y = z * (1.0 / 12.0 - z * (1.0 / 120.0 - z * (1.0 / 252.0 - z * (1.0 / 240.0 -
                     z * (1.0 / 132.0 - z * (691.0 / 32760.0 - z * (1.0 / 12.0)))))));
    return log_ge1_for_f32(xs) - q - 0.5 * ix - y;
}

// Gamma.meanlog (nodes/probabilistic/gamma.py:52-61): parameters cast to f32 first, SciPy's f32
// digamma is the f64 one rounded to f32, then an f32 log and an f32 subtraction.  Parameters are
// clamped to >= 1e-15 by the callers, so only the positive branch is needed on this path.
__device__ __forceinline__ float gamma_meanlog_f32(double a1, double a2) {
    const float a1f = (float)a1;
    const float a2f = (float)a2;
    float psi;
    if (a1f > 0.0f && a1f < INFINITY) psi = (float)digamma_pos_for_f32((double)a1f);
    else psi = (float)digamma_f64((double)a1f);
    const float lg = logf(a2f);
    return psi - lg;
}

}  // namespace oriana

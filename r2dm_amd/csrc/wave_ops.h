// Cross-lane reductions of one wavefront on the vector ALU (gfx950).
//
// __shfl_xor compiles to ds_bpermute_b32: through the LDS crossbar, ~120 cycles per dependent step, two per double.  A
// ten-step butterfly of doubles in the convolution epilogue cost 2 k cycles per tile that way.  CDNA4 has better tools:
//   v_permlane32_swap / v_permlane16_swap   exchange the upper 32 lanes (the odd 16-lane rows) of one register with the
//                                           lower 32 lanes (the even rows) of another
//   DPP                                      quad permutes, row_half_mirror, row_ror inside a 16-lane row
// A swap of the pair (P, Q) followed by P + Q is one reduce-scatter step (lanes of the lower half / even rows end up with P
// summed over the partner lane, the others with Q) -- or, with P = Q, one all-reduce step.
#pragma once
#include <hip/hip_runtime.h>

namespace r2dm {

// a <- (a in the lanes that keep a, b in the lanes that keep b) + the partner lane's copy of the same value
// rows16 = false: partner = lane ^ 32 (lower 32 lanes keep a);  rows16 = true: partner = lane ^ 16 (even rows keep a)
__device__ __forceinline__ void wave_swap_add(double& a, const double b, bool rows16) {
    const unsigned alo = (unsigned)__double2loint(a), ahi = (unsigned)__double2hiint(a);
    const unsigned blo = (unsigned)__double2loint(b), bhi = (unsigned)__double2hiint(b);
    if (rows16) {
        const auto lo = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
        a = __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
    } else {
        const auto lo = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
        const auto hi = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
        a = __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
    }
}
// DPP controls: 0xB1 quad_perm [1,0,3,2] (lane ^ 1), 0x4E quad_perm [2,3,0,1] (lane ^ 2), 0x141 row_half_mirror (7 - lane
// within 8), 0x128 row_ror:8 (lane ^ 8 within the row)
template <int CTRL>
__device__ __forceinline__ double wave_dpp(const double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ float wave_dpp(const float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
// the value of lane ^ 32
__device__ __forceinline__ float wave_xor32(const float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    // r[0] = (own lower half | the lower half again), r[1] = (the upper half | own upper half)
    return (threadIdx.x & 32) ? __uint_as_float(r[0]) : __uint_as_float(r[1]);
}

// ---- all lanes end up with the wave's total / maximum (fixed order) ----
__device__ __forceinline__ double wave_sum_f64(double v) {
    v += wave_dpp<0xB1>(v);    // pairs, quads, the other quad of the 8 (after the quads are uniform a mirror is as good as ^ 4)
    v += wave_dpp<0x4E>(v);
    v += wave_dpp<0x141>(v);
    v += wave_dpp<0x128>(v);
    wave_swap_add(v, v, true);
    wave_swap_add(v, v, false);
    return v;
}
// the same total with the exchange steps in the order of wave_sum8_scatter below (lane ^ 32, ^ 16, ^ 8, then inside the 8): the order
// gn_finalize_kernel sums a group's 256 per-thread partials in, so that conv_f16x2.hip's folded GroupNorm -- which reduces all eight groups
// at once with the reduce-scatter -- gets the same bits
__device__ __forceinline__ double wave_sum_f64_hi_first(double v) {
    wave_swap_add(v, v, false);
    wave_swap_add(v, v, true);
    v += wave_dpp<0x128>(v);
    v += wave_dpp<0xB1>(v);
    v += wave_dpp<0x4E>(v);
    v += wave_dpp<0x141>(v);
    return v;
}
__device__ __forceinline__ float wave_max_f32(float v) {
    v = fmaxf(v, wave_dpp<0xB1>(v));
    v = fmaxf(v, wave_dpp<0x4E>(v));
    v = fmaxf(v, wave_dpp<0x141>(v));
    v = fmaxf(v, wave_dpp<0x128>(v));
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    const auto t = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(t[0]), __uint_as_float(t[1]));
}

// maximum of non-negative float bit patterns / ints over the wave (NaN patterns sort above infinity and survive, unlike fmaxf)
__device__ __forceinline__ int wave_max_i32(int v) {
    auto mx = [](int a, int b) { return a > b ? a : b; };
    v = mx(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false));
    v = mx(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false));
    v = mx(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false));
    v = mx(v, __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false));
    const auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    v = mx((int)r[0], (int)r[1]);
    const auto t = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    return mx((int)t[0], (int)t[1]);
}

// Eight values at once as a butterfly reduce-scatter (4 + 2 + 1 exchange steps halve the live values, three more finish the
// 8-lane groups): afterwards lane L holds the wave total of v[(L >> 3) & 7] (returned).
__device__ __forceinline__ double wave_sum8_scatter(double (&v)[8], int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) wave_swap_add(v[i], v[i + 4], false);  // lane bit 5 selects v[i] / v[i + 4]
#pragma unroll
    for (int i = 0; i < 2; ++i) wave_swap_add(v[i], v[i + 2], true);   // bit 4
    const bool up = lane & 8;                                           // bit 3
    const double keep = up ? v[1] : v[0], send = up ? v[0] : v[1];
    double t = keep + wave_dpp<0x128>(send);
    t += wave_dpp<0xB1>(t);
    t += wave_dpp<0x4E>(t);
    t += wave_dpp<0x141>(t);
    return t;
}
// the same reduce-scatter on fp32 values (conv_f16x2.hip's tile ends): one instruction per exchange instead of two plus a half-rate add
__device__ __forceinline__ void wave_swap_add(float& a, const float b, bool rows16) {
    if (rows16) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
        a = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    } else {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
        a = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
}
__device__ __forceinline__ float wave_sum8_scatter(float (&v)[8], int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) wave_swap_add(v[i], v[i + 4], false);
#pragma unroll
    for (int i = 0; i < 2; ++i) wave_swap_add(v[i], v[i + 2], true);
    const bool up = lane & 8;
    const float keep = up ? v[1] : v[0], send = up ? v[0] : v[1];
    float t = keep + wave_dpp<0x128>(send);
    t += wave_dpp<0xB1>(t);
    t += wave_dpp<0x4E>(t);
    t += wave_dpp<0x141>(t);
    return t;
}
// ... and gathered: every lane gets all eight totals (v_readlane: wave-uniform results)
__device__ __forceinline__ void wave_sum8(double (&v)[8], int lane) {
    const double t = wave_sum8_scatter(v, lane);
    const int lo = __double2loint(t), hi = __double2hiint(t);
#pragma unroll
    for (int j = 0; j < 8; ++j)
        v[j] = __hiloint2double(__builtin_amdgcn_readlane(hi, 8 * j), __builtin_amdgcn_readlane(lo, 8 * j));
}

}  // namespace r2dm

// The exact 22-bit split of fp32 operands for the fp16 matrix pipe (conv_f16x2.hip, attention.hip):
//   v = h + 2^-11 l,  h = RNE_f16(v),  l = RNE_f16(2^11 (v - h))
// x w = xh wh + 2^-11 (xh wl + xl wh) + O(2^-22): three v_mfma_f32_32x32x16_f16, the xh wh products in one fp32 accumulator,
// the cross products in a second one (scripts/probes/f16x2_probe.hip, profiles/r02_f16x2_probe.txt: more accurate on the
// matrix pipe than three bf16 pieces / six products and than the fp32 MFMA).  Operands must fit the fp16 range (65504).
#pragma once
#include "common.h"

namespace r2dm {

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;

namespace f2 {
constexpr float LSCALE = 2048.f, LINV = 1.f / 2048.f;
}

// Power-of-two scale of a layer's weights for the fp16 packings: max|w| s in [2^9, 2^10).  The split v = h + 2^-11 l has an
// ABSOLUTE floor (l leaves the fp16 normals once |v - h| < 2^-25, i.e. below |v| ~ 1e-4: representation error ~1.5e-11
// absolute), which is 1e-6 RELATIVE for weights of 1e-5 -- a trained zero-initialised convolution
// (/root/reference/models/ops.py:9-11).  Scaled, every weight within 2^-23 of the layer's largest keeps all 22 bits, and
// the inverse (exact) goes into the epilogue.  max_bits = float bits of max|w|; returns s, *inv = 1 / s.
__host__ __device__ __forceinline__ float f16x2_weight_scale(int max_bits, float* inv) {
    int e = (max_bits >> 23) & 0xff;  // biased exponent of the maximum
    if (e == 0 || e == 255) {         // all zero (or subnormal) / not finite: unscaled
        *inv = 1.0f;
        return 1.0f;
    }
    e = e < 20 ? 20 : e > 240 ? 240 : e;
    union { int i; float f; } s, r;
    s.i = (263 - e) << 23;  // 2^(9 - (e - 127))
    r.i = (e - 9) << 23;    // 2^((e - 127) - 9)
    *inv = r.f;
    return s.f;
}

// MODE.FP16_OVFL: f16 conversions of this wave saturate at +-65504 instead of producing inf
__device__ __forceinline__ void f16_saturate_mode() { __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1); }

// packed (h0, h1) and (l0, l1) of two fp32 values (low half = first value)
__device__ __forceinline__ void split_f16x2(float v0, float v1, unsigned& ph, unsigned& pl) {
    using f32x2 = __attribute__((ext_vector_type(2))) float;
    const f16x2 h = __builtin_convertvector(f32x2{v0, v1}, f16x2);  // v_cvt_pk_f16_f32 (RNE)
    // 2^11 (v - h), exact (a power-of-two scaling of an exact difference); written so that the fp16 -> fp32 conversion
    // folds into v_fma_mix_f32: two instructions per value instead of three
    const float r0 = __builtin_fmaf(-(float)h[0], f2::LSCALE, v0 * f2::LSCALE), r1 = __builtin_fmaf(-(float)h[1], f2::LSCALE, v1 * f2::LSCALE);
    const f16x2 l = __builtin_convertvector(f32x2{r0, r1}, f16x2);
    ph = __builtin_bit_cast(unsigned, h);
    pl = __builtin_bit_cast(unsigned, l);
}

// the same split with the residual at its TRUE scale, l = RNE_f16(v - h): what a single accumulator needs (all three products
// of x w = xh wh + xh wl + xl wh + O(2^-22) on one scale).  l leaves the fp16 normals below |v| ~ 2^-3 and is then exact to 2^-25
// absolute: O(1) activations keep ~2^-25 of absolute precision (fp32: 2^-24 relative), weights are pre-scaled per layer
__device__ __forceinline__ void split_f16x2_true(float v0, float v1, unsigned& ph, unsigned& pl) {
    using f32x2 = __attribute__((ext_vector_type(2))) float;
    const f16x2 h = __builtin_convertvector(f32x2{v0, v1}, f16x2);
    const float r0 = v0 - (float)h[0], r1 = v1 - (float)h[1];  // exact
    const f16x2 l = __builtin_convertvector(f32x2{r0, r1}, f16x2);
    ph = __builtin_bit_cast(unsigned, h);
    pl = __builtin_bit_cast(unsigned, l);
}

}  // namespace r2dm

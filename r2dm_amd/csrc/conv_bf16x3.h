// Shared pieces of the split-operand 3x3 convolution kernels (conv_bf16x3.hip, conv_f16x2.hip): tile geometry, the
// exact three-way bf16 split, LDS-DMA issue helpers.
#pragma once
#include "common.h"
#include <type_traits>

namespace r2dm {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

namespace x3 {
constexpr int CO_T = 64, TH = 4, TW = 64, XR = 6, NG = 2, CK = 16;
constexpr int WENT = 3 * 3 * NG * CO_T;  // 16-byte entries per weight stage (one kernel row of one chunk)
constexpr int WBYTES = WENT * 16;        // 18432
constexpr int MR = 2, NR = 2;
}  // namespace x3

// Exact, UNBIASED three-way split of two fp32 values into packed bf16 pairs (low half = first value):
//   v == p1 + p2 + p3 with p1 = RNE_bf16(v), p2 = RNE_bf16(v - p1), p3 = v - p1 - p2 (<= 8 significant bits, exact).
// Round-to-nearest pieces are signed and zero-mean, so the three dropped products (p2*q3 + p3*q2 + p3*q3, <= 2^-24
// relative) carry no systematic sign; a truncation split shrinks every product by ~5e-8 -- a coherent bias that a
// 256-step sampler amplifies (measured).  v_cvt_pk_bf16_f32 is gfx950's hardware RNE conversion.
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ void split3_pk(float v0, float v1, unsigned& p1, unsigned& p2, unsigned& p3) {
    p1 = cvt_pk_bf16(v0, v1);
    const float r0 = v0 - __uint_as_float(p1 << 16), r1 = v1 - __uint_as_float(p1 & 0xffff0000u);
    p2 = cvt_pk_bf16(r0, r1);
    p3 = cvt_pk_bf16(r0 - __uint_as_float(p2 << 16), r1 - __uint_as_float(p2 & 0xffff0000u));
}

namespace x3s {
using namespace x3;
constexpr int XS2 = 67;                  // 66 columns + 1 dump column (never read)
constexpr int XPL2 = NG * XR * XS2;      // entries per plane
constexpr int XBYTES2 = 3 * XPL2 * 16;   // 38592
constexpr int RING = 4;
constexpr int WB0 = 2 * XBYTES2;         // [x buffer 0][x buffer 1][weight ring]
}  // namespace x3s

__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;  // M0 = wave-uniform LDS base of the 1 KiB piece; lane i lands at base + 16 i
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

__device__ __forceinline__ void dma16s(const void* gbase, unsigned voff, unsigned lds_dst) {  // wave-uniform base + lane offset
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(gbase), "s"(lds_dst)
                 : "memory");
}

template <int V>
using ic = std::integral_constant<int, V>;


}  // namespace r2dm

// Operand pre-pass of conv_f16x2 for the deep layers (round 4): GroupNorm-affine + SiLU + the f16 split of an activation, ONCE.
//
// conv_f16x2's staging waves transform every input pixel of a tile -- silu(x a + d), then v = h + 2^-11 l -- once per 64-channel
// OUTPUT tile: eight times for a 512-channel layer (/root/reference/models/efficient_unet.py:72-83,95-110: GroupNorm / AdaGN ->
// SiLU -> ops.Conv2d), and that work, not the matrix pipe, bounds a chunk (in-kernel timelines, LABNOTES.md section 5).  For the
// coarse levels the tensor is small (8-34 MB at batch 8): this kernel applies the transform once, in the staging waves' own
// arithmetic (bit-identical products), and writes the two fp16 planes in the order conv_f16x2's LDS x tile wants them, so that the
// convolution's stagers only issue LDS-DMA:
//
//   xs[b][16-channel chunk][plane h | l][8-channel group][H + 2 rows][W columns][8 channels] fp16     (4 bytes per element)
//
// with one zero row above and below the image (the zero padding of the ACTIVATED tensor, ops.py:32-49: a tile row outside
// [0, H) is an ordinary row of this layout; the ring's wrap-around in W is a lane address).  HBM-bound: reads 4, writes 4 bytes per
// element (the tensor was just written by its producer: L2 / Infinity Cache hits).
#include "common.h"
#include "f16x2.h"

namespace r2dm {

namespace {
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

// One thread = one staging unit of conv_f16x2 (8 channels x 4 consecutive pixels of one row), same expressions in the same order
// as the stagers' `xf` (conv_f16x2.hip): affine, SiLU on v_exp / v_rcp, split_f16x2.
template <int PRO>
__global__ __launch_bounds__(256) void presplit_kernel(const Src x, const float2* __restrict__ aff, unsigned short* __restrict__ xs,
                                                       int Cin, int H, int W) {
    f16_saturate_mode();
    const int nq = W >> 2;                         // quads per row
    const int unit = blockIdx.x * 256 + threadIdx.x;  // (row, quad) of the image, row-major
    const int c8 = blockIdx.y, b = blockIdx.z;     // 8-channel group (global), sample
    if (unit >= H * nq) return;
    const int row = unit / nq, quad = unit - row * nq;
    const long HW = (long)H * W;
    const int ci0 = c8 * 8;
    const float* src = x.plane(b, ci0, HW) + (long)row * W + quad * 4;  // (a group of 8 never straddles the concat seam: launcher)
    f32x4 raw[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) raw[i] = *reinterpret_cast<const f32x4*>(src + i * HW);
    f32x4 ad4[4];
    if (PRO != PRO_NONE) {
        const f32x4* ap = reinterpret_cast<const f32x4*>(aff + (long)b * Cin + ci0);
#pragma unroll
        for (int j = 0; j < 4; ++j) ad4[j] = ap[j];
    }
    unsigned ph[4][4], pl[4][4];  // [pixel][channel pair]
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i2 = 0; i2 < 4; ++i2) {
            float qv0 = raw[2 * i2][e], qv1 = raw[2 * i2 + 1][e];
            if (PRO != PRO_NONE) {
                qv0 = qv0 * ad4[i2][0] + ad4[i2][1];
                qv1 = qv1 * ad4[i2][2] + ad4[i2][3];
            }
            if (PRO == PRO_AFFINE_SILU) {
                float qm0 = qv0 * -1.4426950408889634f, qm1 = qv1 * -1.4426950408889634f;
                qm0 = __builtin_amdgcn_exp2f(qm0);
                qm1 = __builtin_amdgcn_exp2f(qm1);
                qm0 = 1.0f + qm0;
                qm1 = 1.0f + qm1;
                qm0 = __builtin_amdgcn_rcpf(qm0);
                qm1 = __builtin_amdgcn_rcpf(qm1);
                qv0 *= qm0;
                qv1 *= qm1;
            }
            split_f16x2(qv0, qv1, ph[e][i2], pl[e][i2]);
        }
    // [b][chunk][plane][group][H + 2][W][8 ch]: this thread's 4 pixels are 64 contiguous bytes per plane
    const int chunk = c8 >> 1, g = c8 & 1, nch = Cin >> 4;
    const long plane_px = (long)(H + 2) * W;  // 16-byte entries per (plane, group)
    u32x4* base = reinterpret_cast<u32x4*>(xs) + ((long)(b * nch + chunk) * 4 + g) * plane_px + (long)(row + 1) * W + quad * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        base[e] = u32x4{ph[e][0], ph[e][1], ph[e][2], ph[e][3]};
        base[2 * plane_px + e] = u32x4{pl[e][0], pl[e][1], pl[e][2], pl[e][3]};
    }
    // the zero rows above and below the image
    if (row == 0 || row == H - 1) {
        u32x4* z = base + (row == 0 ? -(long)W : (long)W);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            z[e] = u32x4{0u, 0u, 0u, 0u};
            z[2 * plane_px + e] = u32x4{0u, 0u, 0u, 0u};
        }
        if (H == 1) {  // (one image row: it is both the first and the last)
            u32x4* z2 = base + (long)W;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                z2[e] = u32x4{0u, 0u, 0u, 0u};
                z2[2 * plane_px + e] = u32x4{0u, 0u, 0u, 0u};
            }
        }
    }
}
}  // namespace

// floats (4-byte units) of the pre-split tensor of a (B, Cin, H, W) activation
long presplit_floats(int B, int Cin, int H, int W) { return (long)B * (Cin / 16) * 4 * (long)(H + 2) * W * 4; }

bool presplit_supported(const Src& x, int Cin, int H, int W) {
    return Cin % 16 == 0 && W % 4 == 0 && x.c0 % 8 == 0 && x.c0 + x.c1 == Cin;
}

hipError_t launch_presplit(const Src& x, const float2* aff, int prologue, float* xs, int B, int Cin, int H, int W, hipStream_t s) {
    if (!presplit_supported(x, Cin, H, W) || (prologue != PRO_NONE && aff == nullptr)) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((H * (W / 4) + 255) / 256), (unsigned)(Cin / 8), (unsigned)B);
    unsigned short* dst = reinterpret_cast<unsigned short*>(xs);
    switch (prologue) {
        case PRO_NONE: presplit_kernel<PRO_NONE><<<grid, 256, 0, s>>>(x, aff, dst, Cin, H, W); break;
        case PRO_AFFINE: presplit_kernel<PRO_AFFINE><<<grid, 256, 0, s>>>(x, aff, dst, Cin, H, W); break;
        case PRO_AFFINE_SILU: presplit_kernel<PRO_AFFINE_SILU><<<grid, 256, 0, s>>>(x, aff, dst, Cin, H, W); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace r2dm

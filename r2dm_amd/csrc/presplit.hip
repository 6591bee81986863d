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
#include "gn_math.h"
#include "wave_ops.h"

namespace r2dm {

namespace {
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

// One thread = one staging unit of conv_f16x2 (8 channels x 4 consecutive pixels of one row), same expressions in the same order
// as the stagers' `xf` (conv_f16x2.hip): affine, SiLU on v_exp / v_rcp, split_f16x2.  ad4[j] = (a, d) of channels 2 j, 2 j + 1.
__device__ __forceinline__ void presplit_load(const Src& x, f32x4 (&raw)[8], int H, int W, int unit, int c8, int b) {
    const int nq = W >> 2;                         // quads per row
    const int row = unit / nq, quad = unit - row * nq;
    const long HW = (long)H * W;
    const float* src = x.plane(b, c8 * 8, HW) + (long)row * W + quad * 4;  // (a group of 8 never straddles the concat seam: launcher)
#pragma unroll
    for (int i = 0; i < 8; ++i) raw[i] = *reinterpret_cast<const f32x4*>(src + i * HW);
}

template <int PRO>
__device__ __forceinline__ void presplit_unit(const f32x4 (&raw)[8], const f32x4 (&ad4)[4], unsigned short* __restrict__ xs, int Cin, int H, int W, int unit, int c8, int b) {
    const int nq = W >> 2;                         // quads per row
    const int row = unit / nq, quad = unit - row * nq;
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

template <int PRO>
__global__ __launch_bounds__(256) void presplit_kernel(const Src x, const float2* __restrict__ aff, unsigned short* __restrict__ xs,
                                                       int Cin, int H, int W) {
    f16_saturate_mode();
    const int unit = blockIdx.x * 256 + threadIdx.x;  // (row, quad) of the image, row-major
    const int c8 = blockIdx.y, b = blockIdx.z;     // 8-channel group (global), sample
    if (unit >= H * (W >> 2)) return;
    f32x4 raw[8];
    presplit_load(x, raw, H, W, unit, c8, b);
    f32x4 ad4[4] = {};
    if (PRO != PRO_NONE) {
        const f32x4* ap = reinterpret_cast<const f32x4*>(aff + (long)b * Cin + c8 * 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) ad4[j] = ap[j];
    }
    presplit_unit<PRO>(raw, ad4, xs, Cin, H, W, unit, c8, b);
}

// Round 6: the same pass with the GroupNorm in front of the convolution FOLDED IN (as conv_f16x2.hip's staging waves fold it): every block
// reduces the statistics slots of its sample's group itself -- gn_finalize_kernel's reduction, slot assignment, order and arithmetic (norm.hip,
// gn_math.h: the same bits) -- and derives (a, d) of its eight channels; no gn_finalize launch, no (a, d) tensor.  For the launches whose
// output-channel tiles are many and small (u_block4 on 32-channel tiles: eight blocks stage the same x tile, the stagers bound a chunk at 4.5 k
// cycles for 1.7 k of MFMAs).  range (optional): the group's bound (gn_bound) as gn_finalize records it.
template <int PRO>
__global__ __launch_bounds__(256) void presplit_fold_kernel(const Src x, unsigned short* __restrict__ xs, int Cin, int H, int W,
                                                            const double* __restrict__ partial, int stride, int nslots, int cpg, float eps,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ ada, long ada_stride, int* __restrict__ range) {
    f16_saturate_mode();
    const int c8 = blockIdx.y, b = blockIdx.z, G = Cin / cpg, g = (c8 * 8) / cpg;
    const int unit = blockIdx.x * 256 + threadIdx.x;
    const bool live = unit < H * (W >> 2);
    // everything that does not depend on the statistics is requested first: this thread's pixels and the norm's parameters of the block's eight channels
    f32x4 raw[8];
    presplit_load(x, raw, H, W, live ? unit : 0, c8, b);
    float w8[8], sh8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = c8 * 8 + i;
        if (ada) {
            w8[i] = ada[b * ada_stride + c];
            sh8[i] = ada[b * ada_stride + Cin + c];
        } else {
            w8[i] = gamma ? gamma[c] : 1.0f;
            sh8[i] = beta ? beta[c] : 0.0f;
        }
    }
    using d2 = __attribute__((ext_vector_type(2))) double;
    const d2* p2 = reinterpret_cast<const d2*>(partial + ((long)b * G + g) * stride * 2);
    double sum = 0.0, sq = 0.0, emax = 0.0;
    for (int s0 = threadIdx.x; s0 < nslots; s0 += 256 * 8) {  // (gn_finalize_kernel's loop: thread t adds slots t, t + 256, ... in ascending order)
        d2 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int sl = s0 + 256 * i;
            v[i] = sl < nslots ? p2[sl] : d2{0.0, 0.0};
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            sum += v[i][0];
            sq += v[i][1];
            emax = v[i][1] > emax ? v[i][1] : emax;
        }
    }
    const bool bound_here = range && blockIdx.x == 0;  // (one block per (sample, 8 channels) records the bound: block-uniform)
    float gmax = bound_here ? wave_max_f32((float)sqrt(emax) * 1.000001f) : 0.f;
    __shared__ double red[2][4];
    __shared__ float redm[4];
    sum = wave_sum_f64_hi_first(sum);
    sq = wave_sum_f64_hi_first(sq);
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = sum;
        red[1][threadIdx.x >> 6] = sq;
        redm[threadIdx.x >> 6] = gmax;
    }
    __syncthreads();
    sum = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    sq = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    const double n = (double)cpg * (double)H * (double)W;
    const GnMoments mo = gn_moments(sum, sq, n, eps);
    f32x4 ad4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float w0 = ada ? 1.0f + w8[2 * j] : w8[2 * j], w1 = ada ? 1.0f + w8[2 * j + 1] : w8[2 * j + 1];
        const float2 e0 = gn_affine(mo, w0, sh8[2 * j]), e1 = gn_affine(mo, w1, sh8[2 * j + 1]);
        ad4[j] = f32x4{e0.x, e0.y, e1.x, e1.y};
    }
    if (bound_here && threadIdx.x == 0) {
        gmax = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));
        float bound = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float w = ada ? 1.0f + w8[i] : w8[i];
            const float bd = gn_bound(mo, gn_affine(mo, w, sh8[i]), w, sh8[i], gmax, n);
            bound = __float_as_int(bd) > __float_as_int(bound) ? bd : bound;  // (as ints: positive floats order like their bit patterns, a NaN above all)
        }
        atomicMax(range + 1, __float_as_int(bound));
    }
    if (!live) return;
    presplit_unit<PRO>(raw, ad4, xs, Cin, H, W, unit, c8, b);
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

hipError_t launch_presplit_fold(const Src& x, int prologue, float* xs, int B, int Cin, int H, int W, const double* partial, int stride, int nslots, int cpg,
                                float eps, const float* gamma, const float* beta, const float* ada, long ada_stride, int* range, hipStream_t s) {
    if (!presplit_supported(x, Cin, H, W) || !partial || cpg < 8 || cpg % 8 || Cin % cpg || nslots < 1 || nslots > stride || (x.p1 && x.c0 % cpg)) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((H * (W / 4) + 255) / 256), (unsigned)(Cin / 8), (unsigned)B);
    unsigned short* dst = reinterpret_cast<unsigned short*>(xs);
    switch (prologue) {
        case PRO_AFFINE: presplit_fold_kernel<PRO_AFFINE><<<grid, 256, 0, s>>>(x, dst, Cin, H, W, partial, stride, nslots, cpg, eps, gamma, beta, ada, ada_stride, range); break;
        case PRO_AFFINE_SILU: presplit_fold_kernel<PRO_AFFINE_SILU><<<grid, 256, 0, s>>>(x, dst, Cin, H, W, partial, stride, nslots, cpg, eps, gamma, beta, ada, ada_stride, range); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace r2dm

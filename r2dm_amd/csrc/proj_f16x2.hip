// K3f: 1x1 convolution (the attention block's qkv and output projections) on the fp16 matrix pipe with exactly split fp32
// operands -- the arithmetic of conv_f16x2.hip (v = h + 2^-11 l, three v_mfma_f32_32x32x16_f16 per 16 k-values, two fp32
// accumulators), as a plain LDS-tiled GEMM:  Y[b][co][px] = sum_ci W[co][ci] f(X[b][ci][px]) + bias (+ residual) (* scale).
//
// Reference: nn.MultiheadAttention's in_proj / out_proj inside SelfAttentionBlock (/root/reference/models/efficient_unet.py:
// 23-53); f = the folded GroupNorm affine for the qkv projection (PRO_AFFINE), identity for the output projection.
// On the fp32-input MFMA (conv_mfma.hip) these four launches ran at 23-86 % of a 157 TF/s peak: 0.24 ms per step for 2 % of
// the FLOPs; with a third of the matrix-pipe time per product they are bound by their own staging instead.
//
// Block = 256 threads = 4 waves, tile = 64 output channels x 256 pixels (4 rows x 64 columns: the tile geometry of the other
// convolution kernels, so conv_epilogue_wide and the GroupNorm-statistics slot grid apply unchanged); a wave owns 64 channels
// x 64 pixels (MR = NR = 2).  K loop in chunks of 32 input channels: every thread loads 8 channels x 4 pixels, applies the
// affine, splits and writes [pixel][32 ch] fp16 rows (h and l planes) to LDS; the weights come pre-split ([co tile][chunk]
// [plane][co][32 ch], pack_proj_f16x2_kernel).  The next chunk's global loads are in flight during the current chunk's MFMAs;
// two blocks per CU overlap one block's staging with the other's products.
// Range: as conv_f16x2.hip -- MODE.FP16_OVFL, and the engine only routes a projection here whose input is GroupNorm-
// normalised (gn_finalize's bound) or the attention core's output (|o| <= max|v|, tracked by the qkv epilogue).
#include "common.h"
#include "conv_bf16x3.h"
#include "conv_epilogue.h"
#include "f16x2.h"
#include <stdlib.h>

namespace r2dm {

namespace p1 {
using namespace x3;  // CO_T 64, TH 4, TW 64, MR 2, NR 2
constexpr int CKP = 32;                  // input channels per chunk
constexpr int ROWB = CKP * 2 + 16;       // bytes per LDS row (32 fp16 + 16 bytes of padding)
constexpr int PX = TH * TW;              // 256
constexpr int XPL = PX * ROWB;           // bytes per x plane
constexpr int WPL = CO_T * ROWB;         // bytes per weight plane
constexpr int PATCH0 = 2 * XPL + 2 * WPL;
constexpr int LDS_TOTAL = PATCH0 + 4 * 1024;  // 55 296
}  // namespace p1

// NPLK = operand planes in use: 2 = the split arithmetic (parity path), 1 = the h plane alone: one fp16 product per MAC, the
// reduced-precision bulk mode (conv_f16x2.hip); the l planes are then neither written nor read.
template <int PRO, int NPLK, int IOM = 0>
__global__ __launch_bounds__(256, 2) void proj_f16x2_kernel(const ConvParams p) {
    constexpr bool X16 = (IOM & 1) != 0, Y16 = (IOM & 2) != 0;  // (round 6) fp16 storage of the input tensor(s) / of the output (+ residual): the one-plane mode's skip convolutions
    using namespace p1;
    using gcf4 = const f32x4 __attribute__((address_space(1)))*;
    using gcu4 = const u32x4 __attribute__((address_space(1)))*;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f16_saturate_mode();
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.H, W = p.W, HW = H * W;
    const int nTw = W / TW, nTh = H / TH, nCoT = p.Cout / CO_T, nchunks = p.Cin / CKP;
    int L = xcd_remap(blockIdx.x, gridDim.x);
    if (p.reverse) L = (int)gridDim.x - 1 - L;  // (round 6: against the walk of the convolution that read the same tensor last -- Infinity Cache)
    const int cot = L % nCoT;
    L /= nCoT;
    const int tw = L % nTw;
    L /= nTw;
    const int th = L % nTh;
    const int b = L / nTh;

    // staging map: thread -> 4 consecutive pixels (quad q) of 8 channels (group cg) of the chunk
    const int q = tid & 63, cg = tid >> 6;
    const int poff = (th * TH + (q >> 4)) * W + tw * TW + (q & 15) * 4;
    const float* xb0 = p.x.p0 + b * p.x.bs0;
    const float* xb1 = p.x.p1 ? p.x.p1 + b * p.x.bs1 : p.x.p0;
    const int c0 = p.x.p1 ? p.x.c0 : p.Cin;  // (a chunk group of 8 never straddles the seam: launcher)
    const float* affb = PRO != PRO_NONE ? reinterpret_cast<const float*>(p.aff) + (size_t)b * p.Cin * 2 : nullptr;
    const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(p.w) + (size_t)cot * nchunks * (2 * CO_T * CKP * 2);

    f32x4 raw[8], ad4[4];
    u32x4 wv[2];
    auto load_chunk = [&](int c) __attribute__((always_inline)) {
        const int ch = c * CKP + cg * 8;
        if constexpr (X16) {  // (element offsets: load4 addresses halves)
            const float* px = ch < c0 ? p.x.p0 : p.x.p1;
            const long e0 = (ch < c0 ? b * p.x.bs0 + (long)ch * HW : b * p.x.bs1 + (long)(ch - c0) * HW) + poff;
#pragma unroll
            for (int i = 0; i < 8; ++i) raw[i] = load4<true>(px, e0 + (long)i * HW);
        } else {
        const float* pl = ch < c0 ? xb0 + (long)ch * HW : xb1 + (long)(ch - c0) * HW;
#pragma unroll
        for (int i = 0; i < 8; ++i) raw[i] = *(gcf4)(pl + (long)i * HW + poff);
        }
        if (PRO != PRO_NONE) {
#pragma unroll
            for (int j = 0; j < 4; ++j) ad4[j] = *(gcf4)(affb + (size_t)(ch + 2 * j) * 2);
        }
        const unsigned char* ws = wsrc + (size_t)c * (2 * CO_T * CKP * 2);
#pragma unroll
        for (int u = 0; u < NPLK; ++u) wv[u] = *(gcu4)(ws + (size_t)(tid + 256 * u) * 16);
    };
    auto store_chunk = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {  // pixel e of the quad: its 8 channels -> one 16-byte vector per plane
            unsigned ph[4], pl[4];
#pragma unroll
            for (int i2 = 0; i2 < 4; ++i2) {
                float v0 = raw[2 * i2][e], v1 = raw[2 * i2 + 1][e];
                if (PRO != PRO_NONE) {
                    v0 = v0 * ad4[i2][0] + ad4[i2][1];
                    v1 = v1 * ad4[i2][2] + ad4[i2][3];
                }
                split_f16x2(v0, v1, ph[i2], pl[i2]);  // (NPLK == 1: the l half is dead code)
            }
            unsigned char* d = smem + (q * 4 + e) * ROWB + cg * 16;
            *reinterpret_cast<u32x4*>(d) = u32x4{ph[0], ph[1], ph[2], ph[3]};
            if (NPLK == 2) *reinterpret_cast<u32x4*>(d + XPL) = u32x4{pl[0], pl[1], pl[2], pl[3]};
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {  // 16-byte unit: plane, output channel, 8-channel part
            const int un = tid + 256 * u, plane = un >> 8, r = un & 255;
            if (NPLK == 2 || u == 0) *reinterpret_cast<u32x4*>(smem + 2 * XPL + plane * WPL + (r >> 2) * ROWB + (r & 3) * 16) = wv[u];
        }
    };

    f32x16 acc[MR][NR], acl[MR][NR];
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int n = 0; n < NR; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = acl[m][n][r] = 0.f;
    int xrow[NR];
#pragma unroll
    for (int n = 0; n < NR; ++n) {
        const int s = wave * NR + n;  // segment: row s / 2, column block s % 2 of the tile
        xrow[n] = ((s >> 1) * TW + (s & 1) * 32 + l31) * ROWB + hi * 16;
    }
    const int wrow = l31 * ROWB + hi * 16;

    load_chunk(0);
    for (int c = 0; c < nchunks; ++c) {
        __syncthreads();  // the previous chunk's fragments have been read
        store_chunk();
        if (c + 1 < nchunks) load_chunk(c + 1);
        __syncthreads();
#pragma unroll
        for (int st = 0; st < CKP / 16; ++st) {
            f16x8 wh[MR], wl[MR], xh[NR], xl[NR];
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                wh[m] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(smem + 2 * XPL + (m * 32) * ROWB + wrow + st * 32));
                if (NPLK == 2) wl[m] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(smem + 2 * XPL + WPL + (m * 32) * ROWB + wrow + st * 32));
            }
#pragma unroll
            for (int n = 0; n < NR; ++n) {
                xh[n] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(smem + xrow[n] + st * 32));
                if (NPLK == 2) xl[n] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(smem + XPL + xrow[n] + st * 32));
            }
#pragma unroll
            for (int m = 0; m < MR; ++m)
#pragma unroll
                for (int n = 0; n < NR; ++n) {
                    if (NPLK == 2) {
                        acl[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[m], xl[n], acl[m][n], 0, 0, 0);
                        acl[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[m], xh[n], acl[m][n], 0, 0, 0);
                    }
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[m], xh[n], acc[m][n], 0, 0, 0);
                }
        }
    }
    const float wsc = p.wscale ? *p.wscale : 1.0f;  // inverse of the packer's power-of-two weight scale (exact)
    if constexpr (NPLK == 2) {
        conv_epilogue_wide<TH, TW, MR, NR, true>(p, acc, acl, b, th, tw, nTw, cot * CO_T, wave, lane,
                                                   reinterpret_cast<float*>(smem + PATCH0) + wave * 256, f2::LINV, wsc);
    } else {
        f32x16 none[1][1];
        conv_epilogue_wide<TH, TW, MR, NR, false, Y16>(p, acc, none, b, th, tw, nTw, cot * CO_T, wave, lane,
                                                         reinterpret_cast<float*>(smem + PATCH0) + wave * 256, 1.0f, wsc);
    }
}

// ---- the same product for TALL shapes (Cout a multiple of 256: the attention projections, u_block4's skip) --------------------
// Round 3.  With 64-channel output tiles the qkv projection (512 -> 1536 over 8192 tokens: a 1536 x 8192 x 512 GEMM) re-stages --
// loads, normalises, splits -- every pixel 24 times and runs at a fifth of what its MFMAs need (73 us; per-launch table in
// profiles/r03_launch_overhead.txt).  Here a block is 256 output channels x ONE 64-pixel row of the 4 x 64 pixel tile: the four waves
// take 64 channels each over the same pixels, a chunk's x tile is 64 pixels x 32 channels (8 values per thread instead of 32),
// the weights (pre-split, straight copies) 4 x 8 KB.  Same packing, same arithmetic, same epilogue and statistics slots (slot =
// tile x 4 + row: each written once, by the block that owns the row); bit-identical output.  What it bought (j124): the three
// plain-input launches 75 -> 67 us, the two GroupNorm-input ones 98 -> 94 us, -0.3 % on the step -- the transform was NOT what bounds
// these products: either tiling moves ~490 MB from L2 into LDS per qkv launch (a 64 x 256 and a 256 x 64 block tile have the same
// perimeter); the lever left is a larger block tile (8 waves), not taken.
namespace p1 {
constexpr int COB = 4 * CO_T;                 // 256 output channels per block
constexpr int XPLT = TW * ROWB;               // bytes per x plane: 64 pixels
constexpr int WPLT = COB * ROWB;              // bytes per weight plane
constexpr int PATCH0T = 2 * XPLT + 2 * WPLT;  // 51 200
constexpr int LDS_TALL = PATCH0T + 4 * 1024;  // 55 296: two blocks per CU
}  // namespace p1

template <int PRO, int NPLK>
__global__ __launch_bounds__(256, 2) void proj_tall_f16x2_kernel(const ConvParams p) {
    using namespace p1;
    using gcf = const float __attribute__((address_space(1)))*;
    using gcf4 = const f32x4 __attribute__((address_space(1)))*;
    using gcu4 = const u32x4 __attribute__((address_space(1)))*;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f16_saturate_mode();
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.H, W = p.W, HW = H * W;
    const int nTw = W / TW, nTh = H / TH, nCoB = p.Cout / COB, nchunks = p.Cin / CKP;
    int L = xcd_remap(blockIdx.x, gridDim.x);
    if (p.reverse) L = (int)gridDim.x - 1 - L;  // (round 6: against the walk of the convolution that read the same tensor last -- Infinity Cache)
    const int cob = L % nCoB;  // (fastest: the blocks that share a pixel row follow each other)
    L /= nCoB;
    const int row = L % TH;
    L /= TH;
    const int tw = L % nTw;
    L /= nTw;
    const int th = L % nTh;
    const int b = L / nTh;

    // staging map: thread -> one pixel (px) of 8 channels (group cg) of the chunk; 64 consecutive lanes = 64 consecutive pixels
    const int px = tid & 63, cg = tid >> 6;
    const int poff = (th * TH + row) * W + tw * TW + px;
    const float* xb0 = p.x.p0 + b * p.x.bs0;
    const float* xb1 = p.x.p1 ? p.x.p1 + b * p.x.bs1 : p.x.p0;
    const int c0 = p.x.p1 ? p.x.c0 : p.Cin;  // (a chunk group of 8 never straddles the seam: launcher)
    const float* affb = PRO != PRO_NONE ? reinterpret_cast<const float*>(p.aff) + (size_t)b * p.Cin * 2 : nullptr;
    constexpr int WCH = 2 * CO_T * CKP * 2;  // bytes of one (co tile, chunk) in the packed weights: planes h, l
    const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(p.w) + (size_t)(cob * 4) * nchunks * WCH;

    float raw[8];
    f32x4 ad4[4];
    u32x4 wv[4][2];
    auto load_chunk = [&](int c) __attribute__((always_inline)) {
        const int ch = c * CKP + cg * 8;
        const float* pl = ch < c0 ? xb0 + (long)ch * HW : xb1 + (long)(ch - c0) * HW;
#pragma unroll
        for (int i = 0; i < 8; ++i) raw[i] = ((gcf)pl)[(long)i * HW + poff];
        if (PRO != PRO_NONE) {
#pragma unroll
            for (int j = 0; j < 4; ++j) ad4[j] = *(gcf4)(affb + (size_t)(ch + 2 * j) * 2);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {  // co tile 4 cob + j of this chunk: 8 KB = 512 16-byte units (256 per plane)
            const unsigned char* ws = wsrc + ((size_t)j * nchunks + c) * WCH;
#pragma unroll
            for (int u = 0; u < NPLK; ++u) wv[j][u] = *(gcu4)(ws + (size_t)(tid + 256 * u) * 16);
        }
    };
    auto store_chunk = [&]() __attribute__((always_inline)) {
        unsigned ph[4], pl[4];
#pragma unroll
        for (int i2 = 0; i2 < 4; ++i2) {
            float v0 = raw[2 * i2], v1 = raw[2 * i2 + 1];
            if (PRO != PRO_NONE) {
                v0 = v0 * ad4[i2][0] + ad4[i2][1];
                v1 = v1 * ad4[i2][2] + ad4[i2][3];
            }
            split_f16x2(v0, v1, ph[i2], pl[i2]);  // (NPLK == 1: the l half is dead code)
        }
        unsigned char* d = smem + px * ROWB + cg * 16;
        *reinterpret_cast<u32x4*>(d) = u32x4{ph[0], ph[1], ph[2], ph[3]};
        if (NPLK == 2) *reinterpret_cast<u32x4*>(d + XPLT) = u32x4{pl[0], pl[1], pl[2], pl[3]};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int u = 0; u < NPLK; ++u) {  // unit tid of plane u: output channel tid >> 2 of the tile, 8-channel part tid & 3
                *reinterpret_cast<u32x4*>(smem + 2 * XPLT + u * WPLT + (j * CO_T + (tid >> 2)) * ROWB + (tid & 3) * 16) = wv[j][u];
            }
    };

    f32x16 acc[MR][NR], acl[MR][NR];
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int n = 0; n < NR; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = acl[m][n][r] = 0.f;
    const int xrow = l31 * ROWB + hi * 16;                          // + n * 32 pixels
    const int wrow = (wave * CO_T + l31) * ROWB + hi * 16;          // + m * 32 channels

    load_chunk(0);
    for (int c = 0; c < nchunks; ++c) {
        __syncthreads();  // the previous chunk's fragments have been read
        store_chunk();
        if (c + 1 < nchunks) load_chunk(c + 1);
        __syncthreads();
#pragma unroll
        for (int st = 0; st < CKP / 16; ++st) {
            f16x8 wh[MR], wl[MR], xh[NR], xl[NR];
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                wh[m] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(smem + 2 * XPLT + (m * 32) * ROWB + wrow + st * 32));
                if (NPLK == 2) wl[m] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(smem + 2 * XPLT + WPLT + (m * 32) * ROWB + wrow + st * 32));
            }
#pragma unroll
            for (int n = 0; n < NR; ++n) {
                xh[n] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(smem + (n * 32) * ROWB + xrow + st * 32));
                if (NPLK == 2) xl[n] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(smem + XPLT + (n * 32) * ROWB + xrow + st * 32));
            }
#pragma unroll
            for (int m = 0; m < MR; ++m)
#pragma unroll
                for (int n = 0; n < NR; ++n) {
                    if (NPLK == 2) {
                        acl[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[m], xl[n], acl[m][n], 0, 0, 0);
                        acl[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[m], xh[n], acl[m][n], 0, 0, 0);
                    }
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[m], xh[n], acc[m][n], 0, 0, 0);
                }
        }
    }
    const float wsc = p.wscale ? *p.wscale : 1.0f;  // inverse of the packer's power-of-two weight scale (exact)
    // the epilogue of the other convolution kernels: this wave owns channels cob * 256 + wave * 64 .. + 63 of pixel row `row` of the tile
    if constexpr (NPLK == 2) {
        conv_epilogue_wide<TH, TW, MR, NR, true>(p, acc, acl, b, th, tw, nTw, cob * COB + wave * CO_T, row, lane,
                                                   reinterpret_cast<float*>(smem + PATCH0T) + wave * 256, f2::LINV, wsc);
    } else {
        f32x16 none[1][1];
        conv_epilogue_wide<TH, TW, MR, NR, false>(p, acc, none, b, th, tw, nTw, cob * COB + wave * CO_T, row, lane,
                                                    reinterpret_cast<float*>(smem + PATCH0T) + wave * 256, 1.0f, wsc);
    }
}

// ---- weight packing: (Cout, Cin) fp32 -> [co tile][chunk][plane h / l][co 64][32 ch] f16 ----
// range[0] is raised to 1 if a weight does not fit the fp16 range (|w| >= 65504); the packed value saturates.
// wscale: as pack_conv_f16x2_kernel (conv_f16x2.hip) -- the layer's weights are scaled by a power of two, [1] <- its inverse
__global__ void pack_proj_f16x2_kernel(const float* __restrict__ w, unsigned* __restrict__ dst, int Cout, int Cin, long pairs,
                                       int* __restrict__ range, float* __restrict__ wscale) {
    using namespace p1;
    f16_saturate_mode();
    float inv = 1.0f;
    const float ws = wscale ? f16x2_weight_scale(reinterpret_cast<const int*>(wscale)[0], &inv) : 1.0f;
    if (wscale && blockIdx.x == 0 && threadIdx.x == 0) wscale[1] = inv;
    const int nchunks = Cin / CKP;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < pairs; i += (long)gridDim.x * blockDim.x) {
        long r = i;  // one pair of input channels of one output channel
        const int cp = r % (CKP / 2);
        r /= CKP / 2;
        const int col = r % CO_T;
        r /= CO_T;
        const int c = r % nchunks;
        const int cot = r / nchunks;
        const int co = cot * CO_T + col, ci = c * CKP + 2 * cp;
        const float v0 = w[(long)co * Cin + ci] * ws, v1 = w[(long)co * Cin + ci + 1] * ws;
        if ((!(fabsf(v0) < 65504.f) || !(fabsf(v1) < 65504.f)) && range) atomicOr(range, 1);
        unsigned ph, pl;
        split_f16x2(v0, v1, ph, pl);
        const size_t base = ((size_t)(cot * nchunks + c) * 2) * (CO_T * CKP / 2);  // in 4-byte units
        dst[base + (size_t)col * (CKP / 2) + cp] = ph;
        dst[base + (size_t)(CO_T * CKP / 2) + (size_t)col * (CKP / 2) + cp] = pl;
    }
}

bool proj_f16x2_supported(int Cin, int Cout, int taps, int H, int W) {
    return taps == 1 && Cin % p1::CKP == 0 && Cout % p1::CO_T == 0 && H % p1::TH == 0 && W % p1::TW == 0;
}
long proj_f16x2_packed_floats(int Cin, int Cout) { return (long)Cin * Cout; }

hipError_t launch_pack_proj_f16x2(const float* w, float* dst, int Cout, int Cin, int* range_flag, hipStream_t s, float* wscale) {
    if (Cin % p1::CKP || Cout % p1::CO_T) return hipErrorInvalidValue;
    const long pairs = (long)Cout * Cin / 2;
    if (wscale) {
        hipError_t e = launch_weight_absmax(w, (long)Cout * Cin, reinterpret_cast<int*>(wscale), s);
        if (e != hipSuccess) return e;
    }
    const int blocks = (int)((pairs + 255) / 256 < 2048 ? (pairs + 255) / 256 : 2048);
    pack_proj_f16x2_kernel<<<blocks, 256, 0, s>>>(w, reinterpret_cast<unsigned*>(dst), Cout, Cin, pairs, range_flag, wscale);
    return hipGetLastError();
}

template <int PRO, int NPLK>
static hipError_t launch_p1_tall(const ConvParams& p, hipStream_t s) {
    auto kern = proj_tall_f16x2_kernel<PRO, NPLK>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, p1::LDS_TALL);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const unsigned grid = (unsigned)((p.Cout / p1::COB) * p1::TH * (p.W / p1::TW) * (p.H / p1::TH) * p.B);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), p1::LDS_TALL, s, p);
    return hipGetLastError();
}

template <int PRO, int NPLK, int IOM = 0>
static hipError_t launch_p1(const ConvParams& p, hipStream_t s) {
    auto kern = proj_f16x2_kernel<PRO, NPLK, IOM>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, p1::LDS_TOTAL);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const unsigned grid = (unsigned)((p.Cout / p1::CO_T) * (p.W / p1::TW) * (p.H / p1::TH) * p.B);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), p1::LDS_TOTAL, s, p);
    return hipGetLastError();
}

hipError_t launch_proj_f16x2(const ConvParams& p, hipStream_t s) {
    if (!proj_f16x2_supported(p.Cin, p.Cout, p.taps, p.H, p.W)) return hipErrorInvalidValue;
    if (p.x.p1 && p.x.c0 % 8) return hipErrorInvalidValue;  // a thread's 8 channels must not straddle the concat seam
    if (p.prologue == PRO_AFFINE_SILU || (p.prologue != PRO_NONE && p.aff == nullptr)) return hipErrorInvalidValue;
    if (p.stat && p.stat_slots != conv_stat_slots(p.H, p.W)) return hipErrorInvalidValue;
    if (p.x16 || p.y16) {  // fp16 storage (round 6): the skip convolutions of the full-resolution levels in the one-plane mode -- raw fp16 input, fp16 output
        if (!(p.x16 && p.y16) || p.pieces != 1 || p.prologue != PRO_NONE || p.res || p.stat) return hipErrorInvalidValue;
        return launch_p1<PRO_NONE, 1, 3>(p, s);
    }
    static const bool tall_on = getenv("R2DM_PROJ_TALL") == nullptr || atoi(getenv("R2DM_PROJ_TALL")) != 0;  // (0: experiments)
    if (tall_on && p.Cout % p1::COB == 0) {  // 256-channel blocks: every pixel staged Cout / 256 times instead of Cout / 64
        if (p.pieces == 1) return p.prologue == PRO_NONE ? launch_p1_tall<PRO_NONE, 1>(p, s) : launch_p1_tall<PRO_AFFINE, 1>(p, s);
        return p.prologue == PRO_NONE ? launch_p1_tall<PRO_NONE, 2>(p, s) : launch_p1_tall<PRO_AFFINE, 2>(p, s);
    }
    if (p.pieces == 1) return p.prologue == PRO_NONE ? launch_p1<PRO_NONE, 1>(p, s) : launch_p1<PRO_AFFINE, 1>(p, s);
    return p.prologue == PRO_NONE ? launch_p1<PRO_NONE, 2>(p, s) : launch_p1<PRO_AFFINE, 2>(p, s);
}

}  // namespace r2dm

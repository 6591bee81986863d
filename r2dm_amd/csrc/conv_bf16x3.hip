// K2 (fast form): ring-padded 3x3 convolution, fp32 in / fp32 out, on the bf16 matrix cores with every fp32 operand
// split EXACTLY into three bf16 pieces (x = x1 + x2 + x3, 8 mantissa bits each) and six of the nine piece products
// accumulated in fp32:  a*b ~ a1b1 + (a1b2 + a2b1) + (a1b3 + a2b2 + a3b1); the dropped terms are <= 2^-23 |a||b|.
//
// Same contract as conv_mfma.hip (reference ops.Conv2d + ops.Pad, /root/reference/models/ops.py:32-49,149-173, with
// the fused GroupNorm-affine + SiLU prologue and bias / residual / scale / GroupNorm-statistics epilogue of
// /root/reference/models/efficient_unet.py:95-110).  Measured on MI355X (scripts/bf16_split_accuracy.hip): the
// six-product form has the same error against fp64 as the fp32 FMA chain (rms 2.2e-7 vs 2.4e-7 at K=576, 6.7e-7 vs
// 6.9e-7 at K=4608), while v_mfma_f32_32x32x16_bf16 retires 16 k-values in 32 cycles where the fp32-input
// v_mfma_f32_32x32x2_f32 needs 8 x 64: 6 x 32 = 192 cycles instead of 512 per 16 k-values of a 32x32 tile.
//
// GEMM view: M = Cout, N = pixels, K = Cin*9.  One MFMA consumes 16 input channels at one tap; lane (l31, hi) holds
// channels 8hi..8hi+7 of output channel / pixel l31, i.e. one 16-byte LDS entry per operand and plane:
//   x tile   [plane 3][group 2][row 6][col 66][8 ch] bf16   (4x64 output pixels + halo; wrap in W, zero in H)
//   weights  [plane 3][tap-in-row 3][group 2][co 64][8 ch] bf16 per (16-channel chunk, kernel row) stage, pre-split and
//            pre-ordered at load time so that staging is a straight 16-byte copy.
// Block = 4 waves = 64 output channels x (4 x 64) pixels, every wave 64 co x 64 px (2 x 2 MFMA tiles, 24 MFMAs per
// tap).  LDS: one x tile (38 KB) + two weight stages (2 x 18 KB) = 73 KB -> two blocks per CU; while one block
// transforms its next chunk (GroupNorm affine, SiLU, split, pack: VALU) the other one owns the matrix pipe.
#include "common.h"
#include "conv_epilogue.h"

namespace r2dm {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

namespace x3 {
constexpr int CO_T = 64, TH = 4, TW = 64, XR = 6, XS = 66, NG = 2, CK = 16;
constexpr int XPL = NG * XR * XS;        // 16-byte entries per plane of the x tile
constexpr int XBYTES = 3 * XPL * 16;     // 38016
constexpr int WENT = 3 * 3 * NG * CO_T;  // 16-byte entries per weight stage (one kernel row of one chunk)
constexpr int WBYTES = WENT * 16;        // 18432
constexpr int LDS = XBYTES + 2 * WBYTES; // 74880
constexpr int MR = 2, NR = 2;
constexpr int NWT = (WENT + 255) / 256;  // 16-byte weight pieces per thread and stage
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

template <int PRO, bool ACC2>
__global__ __launch_bounds__(256, ACC2 ? 1 : 2) void conv_bf16x3_kernel(const ConvParams p) {
    using namespace x3;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)smem;

    const int tid = threadIdx.x;
    const int lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // = pixel quarter of the tile (1 x 4 waves)

    const int H = p.H, W = p.W;
    const int HW = H * W;
    const int nTw = (W + TW - 1) / TW, nTh = (H + TH - 1) / TH;
    const int nCoT = p.Cout / CO_T;
    int L = xcd_remap(blockIdx.x, gridDim.x);
    const int cot = L % nCoT;
    L /= nCoT;
    const int tw = L % nTw;
    L /= nTw;
    const int th = L % nTh;
    const int b = L / nTh;
    unsigned long long t0 = 0, t1 = 0, t2 = 0;  // perf probe (scripts/conv_phases.py)
    if (p.prof) t0 = __builtin_amdgcn_s_memtime();

    // ---- x staging unit of this thread (constant over the block) ----
    // waves 0-2: one (row, group, 4-column) unit = 8 channels x 4 pixels = 8 float4 loads -> 4 entries x 3 planes;
    // wave 3   : lanes 0-23 one (row, group, side) halo column = 8 scalar loads -> 1 entry x 3 planes.
    const bool quadw = wave < 3;
    int s_row, s_g, s_col, gc;
    if (quadw) {
        s_row = tid >> 5;
        s_g = (tid >> 4) & 1;
        s_col = 1 + (tid & 15) * 4;
        gc = tw * TW + (tid & 15) * 4;
    } else {
        const int u = lane < 24 ? lane : 23;  // surplus lanes redo the last unit (same data, same place)
        s_row = u >> 2;
        s_g = (u >> 1) & 1;
        s_col = (u & 1) ? XS - 1 : 0;
        gc = (u & 1) ? tw * TW + TW : tw * TW - 1;
    }
    if (gc < 0) gc += W;
    while (gc >= W) gc -= W;  // azimuth is periodic; also covers tiles overhanging a narrow image
    const int gr = th * TH + s_row - 1;
    const bool s_ok = gr >= 0 && gr < H;  // rows outside [0,H) are zero padding (of the ACTIVATED tensor)
    const long s_goff = (long)s_g * 8 * HW + (s_ok ? gr * W + gc : 0);
    unsigned char* s_lds = smem + ((s_g * XR + s_row) * XS + s_col) * 16;  // plane 0

    const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(p.w);
    const int nchunks = p.Cin / CK;
    const size_t wstage0 = (size_t)cot * nchunks * 3;
    const float* xb0 = p.x.p0 + b * p.x.bs0;
    const float* xb1 = p.x.p1 ? p.x.p1 + b * p.x.bs1 : p.x.p0;
    const int c0 = p.x.c0;
    const float* affb = PRO != PRO_NONE ? reinterpret_cast<const float*>(p.aff) + ((size_t)b * p.Cin + s_g * 8) * 2 : nullptr;

    f32x4 raw[8];   // waves 0-2: 8 channels x 4 pixels; wave 3 uses component 0 only
    f32x4 ad4[4];   // (a, d) of the unit's 8 channels
    u32x4 wv[NWT];

    auto load_x = [&](int ci0) __attribute__((always_inline)) {
        const float* q = (ci0 >= c0 ? xb1 + (long)(ci0 - c0) * HW : xb0 + (long)ci0 * HW) + s_goff;
        if (quadw) {
#pragma unroll
            for (int i = 0; i < 8; ++i) raw[i] = *reinterpret_cast<const f32x4*>(q + (long)i * HW);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) raw[i][0] = q[(long)i * HW];
        }
        if (PRO != PRO_NONE) {
            const f32x4* ap = reinterpret_cast<const f32x4*>(affb + (size_t)ci0 * 2);
#pragma unroll
            for (int j = 0; j < 4; ++j) ad4[j] = ap[j];
        }
    };
    // transform (affine, SiLU, zero padding), split into three bf16 planes, write the entries of pixel e
    auto store_px = [&](int e, unsigned char* dst) __attribute__((always_inline)) {
        unsigned w1[4], w2[4], w3[4];
#pragma unroll
        for (int i2 = 0; i2 < 4; ++i2) {
            float v[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                float t = raw[2 * i2 + k][e];
                if (PRO != PRO_NONE) {
                    t = t * ad4[i2][2 * k] + ad4[i2][2 * k + 1];
                    if (PRO == PRO_AFFINE_SILU) t = silu_f(t);
                }
                v[k] = s_ok ? t : 0.f;
            }
            split3_pk(v[0], v[1], w1[i2], w2[i2], w3[i2]);
        }
        *reinterpret_cast<u32x4*>(dst) = u32x4{w1[0], w1[1], w1[2], w1[3]};
        *reinterpret_cast<u32x4*>(dst + XPL * 16) = u32x4{w2[0], w2[1], w2[2], w2[3]};
        *reinterpret_cast<u32x4*>(dst + 2 * XPL * 16) = u32x4{w3[0], w3[1], w3[2], w3[3]};
    };
    auto store_x = [&]() __attribute__((always_inline)) {
        if (quadw) {
#pragma unroll
            for (int e = 0; e < 4; ++e) store_px(e, s_lds + e * 16);
        } else {
            store_px(0, s_lds);
        }
    };
    auto load_w = [&](int sigma) __attribute__((always_inline)) {
        const u32x4* w4 = reinterpret_cast<const u32x4*>(wsrc + (wstage0 + sigma) * WBYTES);
#pragma unroll
        for (int i = 0; i < NWT; ++i) {
            const int e = tid + i * 256;  // clamped, never predicated (a predicated load costs a vmcnt wait)
            wv[i] = w4[e < WENT ? e : WENT - 1];
        }
    };
    auto store_w = [&](int buf) __attribute__((always_inline)) {
        u32x4* w4 = reinterpret_cast<u32x4*>(smem + XBYTES + buf * WBYTES);
#pragma unroll
        for (int i = 0; i < NWT; ++i) {
            const int e = tid + i * 256;
            if (WENT % 256 == 0 || e < WENT) w4[e] = wv[i];
        }
    };

    // ---- fragment bases (bytes) ----
    unsigned lds_x[NR];
#pragma unroll
    for (int n = 0; n < NR; ++n) {
        const int s = wave * NR + n;  // segment: row s/2, column block s%2 of the 4x64 tile
        lds_x[n] = lds0 + (unsigned)(((hi * XR + (s >> 1)) * XS + (s & 1) * 32 + l31) * 16);
    }
    const unsigned lds_w0 = lds0 + XBYTES + (unsigned)((hi * CO_T + l31) * 16);

    f32x16 acc[MR][NR];
    f32x16 acc2[ACC2 ? MR : 1][ACC2 ? NR : 1];
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int n = 0; n < NR; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[m][n][r] = 0.f;
                if (ACC2) acc2[m][n][r] = 0.f;
            }

    load_x(0);
    load_w(0);
    store_x();
    store_w(0);
    __syncthreads();
    if (p.prof) t1 = __builtin_amdgcn_s_memtime();

    for (int c = 0; c < nchunks; ++c) {
        const bool more = c + 1 < nchunks;
        if (more) load_x((c + 1) * CK);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int sigma = c * 3 + ky;
            const bool last = !more && ky == 2;
            if (!last) load_w(sigma + 1);
            const unsigned lds_w = lds_w0 + (unsigned)((sigma & 1) * WBYTES);
            // operand fragments are fetched one tap ahead of their use with hand-issued ds_read_b128 and retired with
            // a counted lgkmcnt (LDS operations complete in order; see conv_mfma.hip)
            u32x4 fa[2][3][MR], fb[2][3][NR];
            auto frag = [&](int tx, u32x4 (&a)[3][MR], u32x4 (&bb)[3][NR]) __attribute__((always_inline)) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                    for (int m = 0; m < MR; ++m)
                        asm volatile("ds_read_b128 %0, %1 offset:%2"
                                     : "=v"(a[pl][m])
                                     : "v"(lds_w), "i"(pl * (3 * NG * CO_T * 16) + tx * (NG * CO_T * 16) + m * 512));
#pragma unroll
                    for (int n = 0; n < NR; ++n)
                        asm volatile("ds_read_b128 %0, %1 offset:%2"
                                     : "=v"(bb[pl][n])
                                     : "v"(lds_x[n]), "i"(pl * (XPL * 16) + ky * (XS * 16) + tx * 16));
                }
            };
            frag(0, fa[0], fb[0]);
#pragma unroll
            for (int tx = 0; tx < 3; ++tx) {
                if (tx + 1 < 3) {
                    frag(tx + 1, fa[(tx + 1) & 1], fb[(tx + 1) & 1]);
                    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(3 * (MR + NR)) : "memory");
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
                // smallest products first; the four accumulators alternate so no MFMA waits on its predecessor
                constexpr int PI[6] = {2, 0, 1, 1, 0, 0}, PJ[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                for (int q = 0; q < 6; ++q)
#pragma unroll
                    for (int m = 0; m < MR; ++m)
#pragma unroll
                        for (int n = 0; n < NR; ++n)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                __builtin_bit_cast(bf16x8, fa[tx & 1][PI[q]][m]),
                                __builtin_bit_cast(bf16x8, fb[tx & 1][PJ[q]][n]), acc[m][n], 0, 0, 0);
            }
            if (!last) store_w((sigma + 1) & 1);
            __syncthreads();
        }
        if (more) {
            store_x();  // every wave is past its last read of this chunk's tile
            __syncthreads();
        }
        // v_mfma_f32_32x32x16_bf16 rounds its accumulation toward -inf-ish (measured: the error of +A*B and of -A*B
        // are BOTH negative, ~-7e-11 per instruction at O(1) sums -- a coherent offset that a 256-step sampler
        // amplifies, unlike the zero-mean rounding of the fp32 FMA).  The accumulator therefore changes sign after
        // every chunk and odd chunks use pre-negated weights: acc = (-1)^c * S_c, so consecutive chunks push the
        // offset in opposite directions and it cancels (Cin/16 is even).
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int n = 0; n < NR; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = -acc[m][n][r];
        if (ACC2 && ((c + 1) % 4 == 0)) {  // two-level accumulation: flush every 64 input channels
#pragma unroll
            for (int m = 0; m < MR; ++m)
#pragma unroll
                for (int n = 0; n < NR; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        acc2[m][n][r] += acc[m][n][r];
                        acc[m][n][r] = 0.f;
                    }
        }
    }
    if (p.prof) t2 = __builtin_amdgcn_s_memtime();

    conv_epilogue<4, TH, TW, MR, NR, ACC2>(p, acc, acc2, b, th, tw, nTw, cot * CO_T, wave, lane);

    if (p.prof && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long* o = p.prof + (size_t)blockIdx.x * 8;
        o[0] = t0;
        o[1] = t1;
        o[2] = t2;
        o[3] = __builtin_amdgcn_s_memtime();
        o[4] = o[5] = o[6] = 0;
        o[7] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);
    }
}

// ---- weight packing: OIHW fp32 -> [co tile][chunk][kernel row][plane][tap in row][group][co 64][8 ch] bf16 ----
__global__ void pack_conv_bf16x3_kernel(const float* __restrict__ w, unsigned short* __restrict__ dst, int Cout,
                                        int Cin, long total) {
    using namespace x3;
    const int nchunks = Cin / CK;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long r = i;
        const int ch = r % 8;
        r /= 8;
        const int col = r % CO_T;
        r /= CO_T;
        const int g = r % NG;
        r /= NG;
        const int tx = r % 3;
        r /= 3;
        const int pl = r % 3;
        r /= 3;
        const int ky = r % 3;
        r /= 3;
        const int c = r % nchunks;
        const int cot = r / nchunks;
        const int co = cot * CO_T + col, ci = c * CK + g * 8 + ch;
        unsigned u[3];
        float v = w[((long)co * Cin + ci) * 9 + ky * 3 + tx];
        if (c & 1) v = -v;  // odd chunks are accumulated with flipped sign (see the kernel: rounding-bias cancellation)
        split3_pk(v, v, u[0], u[1], u[2]);
        dst[i] = (unsigned short)(u[pl] & 0xffffu);
    }
}

bool conv_bf16x3_supported(int Cin, int Cout, int taps) {
    return taps == 9 && Cin % (2 * x3::CK) == 0 && Cout % x3::CO_T == 0;  // an even number of 16-channel chunks
}

long conv_bf16x3_packed_floats(int Cin, int Cout) { return (long)Cout * Cin * 9 * 3 / 2; }

hipError_t launch_pack_conv_bf16x3(const float* w, float* dst, int Cout, int Cin, hipStream_t s) {
    const long total = (long)Cout * Cin * 9 * 3;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    pack_conv_bf16x3_kernel<<<blocks, 256, 0, s>>>(w, reinterpret_cast<unsigned short*>(dst), Cout, Cin, total);
    return hipGetLastError();
}

template <int PRO, bool ACC2>
static hipError_t launch_x3(const ConvParams& p, hipStream_t s) {
    auto kern = conv_bf16x3_kernel<PRO, ACC2>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, x3::LDS);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int nTw = (p.W + 63) / 64, nTh = (p.H + 3) / 4, nCoT = p.Cout / x3::CO_T;
    const long nblk = (long)nCoT * nTw * nTh * p.B;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), x3::LDS, s, p);
    return hipGetLastError();
}

hipError_t launch_conv_bf16x3(const ConvParams& p, hipStream_t s) {
    if (!conv_bf16x3_supported(p.Cin, p.Cout, p.taps)) return hipErrorInvalidValue;
    if (p.x.p1 && p.x.c0 % x3::CK) return hipErrorInvalidValue;  // a chunk must not straddle the concat seam
    if (p.prologue != PRO_NONE && p.aff == nullptr) return hipErrorInvalidValue;
    if (p.H * (long)p.W * 16 >= (1L << 31) || p.W % 4) return hipErrorInvalidValue;
    const bool deep = p.Cin > 128;  // long reductions (K = 9*Cin > 1152) get the two-level accumulator
    switch (p.prologue) {
        case PRO_NONE: return deep ? launch_x3<PRO_NONE, true>(p, s) : launch_x3<PRO_NONE, false>(p, s);
        case PRO_AFFINE: return deep ? launch_x3<PRO_AFFINE, true>(p, s) : launch_x3<PRO_AFFINE, false>(p, s);
        case PRO_AFFINE_SILU: return deep ? launch_x3<PRO_AFFINE_SILU, true>(p, s) : launch_x3<PRO_AFFINE_SILU, false>(p, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace r2dm

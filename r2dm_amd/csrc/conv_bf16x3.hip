// K2 (fast form): ring-padded 3x3 convolution, fp32 in / fp32 out, on the bf16 matrix cores with every fp32 operand
// split EXACTLY into three bf16 pieces (x = x1 + x2 + x3, 8 mantissa bits each, round-to-nearest) and six of the nine
// piece products accumulated in fp32:  a*b ~ a1b1 + (a1b2 + a2b1) + (a1b3 + a2b2 + a3b1); the dropped terms are
// <= 2^-23 |a||b| and zero-mean.
//
// Same contract as conv_mfma.hip (reference ops.Conv2d + ops.Pad, /root/reference/models/ops.py:32-49,149-173, with
// the fused GroupNorm-affine + SiLU prologue and bias / residual / scale / GroupNorm-statistics epilogue of
// /root/reference/models/efficient_unet.py:95-110).  Measured on MI355X (scripts/bf16_split_accuracy.hip,
// scripts/conv_accuracy.py): the six-product form has the same error against fp64 as the fp32 FMA chain (rms 2.3e-7
// for both kernels on the U-Net's layer shapes), while v_mfma_f32_32x32x16_bf16 retires 16 k-values in 32 cycles
// where the fp32-input v_mfma_f32_32x32x2_f32 needs 8 x 64: 6 x 32 = 192 cycles instead of 512 per 16 k-values.
// Two properties of the bf16 instruction had to be engineered around (both measured):
//   * its accumulation rounds toward minus infinity (the error of +A*B and of -A*B are BOTH negative, ~-7e-11 per
//     instruction at O(1) sums): a coherent offset that the sampler amplifies.  Every layer therefore accumulates
//     half of its channel blocks with pre-negated weights and opposite sign, so the offsets cancel;
//   * it occupies the matrix pipe for 32 cycles but only ~4 of issue: other instructions are free only if they sit
//     in that shadow, a few per MFMA.  Both kernels are written as "units" = one MFMA + a thin slice of other
//     work, fenced with sched_barrier(0).
//
// GEMM view: M = Cout, N = pixels, K = Cin*9.  One MFMA consumes 16 input channels at one tap; lane (l31, hi) holds
// channels 8hi..8hi+7 of output channel / pixel l31, i.e. one 16-byte LDS entry per operand and plane:
//   x tile   [plane 3][group 2][row 6][col 67][8 ch] bf16   (4x64 output pixels + halo + a dump column; wrap in W,
//            zero in H)
//   weights  [plane 3][tap-in-row 3][group 2][co 64][8 ch] bf16 per (16-channel chunk, kernel row) stage, pre-split,
//            pre-signed and pre-ordered at load time so that staging is a straight LDS-DMA copy.
// Block = 4 waves = 64 output channels x (4 x 64) pixels, every wave 64 co x 64 px (2 x 2 MFMA tiles, 24 MFMAs per
// tap).  Two kernels share this layout:
//   conv_bf16x3_stream_kernel  Cin > 128: one block per CU, everything hidden inside the wave's own MFMA stream;
//   conv_bf16x3_pair_kernel    Cin <= 128: two blocks per CU cover each other's prologue / epilogue / chunk boundary.
#include "common.h"
#include "conv_bf16x3.h"
#include "conv_epilogue.h"
#include <stdlib.h>

namespace r2dm {

// ---- deep variant: one software-pipelined instruction stream per wave, one block per CU ------------------------
// For long reductions (Cin > 128) the two-level accumulator needs 64 more registers than two blocks per CU allow,
// so this variant runs ONE block per CU (up to 512 VGPRs per lane) and hides everything in the shadow of its own
// MFMAs instead of behind a second block.  A bf16 MFMA occupies the matrix pipe for 32 cycles but issues in ~4; an
// in-order wave keeps the pipe full only if at most ~5 other instructions sit between two MFMAs
// (MI355X guide), so the whole kernel is written as 24 "units" per tap = one MFMA + a thin slice of other work,
// fenced by sched_barrier(0) so that hipcc keeps the order:
//   * x tile double-buffered in LDS: the next chunk is transformed (affine, SiLU, split, pack) as 16 two-channel
//     pieces cut into five slices each (taps 1..6, one slice every other unit) and written straight into the other
//     buffer; all four waves run the same code (the halo-column lanes load the aligned quad that holds their column
//     and send the three pixels they do not need to a dump column of the tile);
//   * weights go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4: no registers, no ds_write) into a ring of
//     four (chunk, kernel row) stages, issued three stages ahead, five 1 KiB pieces per wave and stage;
//   * operand fragments of the next tap are read during the first 12 units of a tap, across stage and chunk
//     boundaries;
//   * one s_barrier per stage, after the stage's first tap: it publishes the next stage's weights (every wave
//     first waits for its own DMA pieces with a counted vmcnt) and, in the last stage of a chunk, the next x tile.
template <int PRO, int COT, int NPC>
__global__ __launch_bounds__(256, 1) void conv_bf16x3_stream_kernel(const ConvParams p) {
    using namespace x3s;
    // COT = 64: every wave 64 co x 64 px (2 x 2 MFMA tiles).  COT = 32 (launches that would otherwise leave CUs idle):
    // 32 co x 64 px (1 x 2), half the MFMAs per tap under the same x tile, fragment and transform traffic.
    // NPC = 3: exact three-piece split, six products (fp32-class error).  (NPC = 2 -- two pieces, three products, ~2^-16 relative
    // error -- was round 1's reduced-precision mode; it is no longer instantiated: conv_f16x2.hip does three products exactly.)
    constexpr int CO_T = COT, MR = COT / 32, NPROD = NPC == 3 ? 6 : 3, UNITS = NPROD * MR * NR, NFR = NPC * (MR + NR);
    constexpr int LPU = (12 + UNITS - 1) / UNITS;  // raw-load pieces per unit in tap 7
    static_assert(NPC == 3 || NPC == 2, "pieces");
    static_assert(NFR <= UNITS && 24 % UNITS == 0, "unit schedule");
    constexpr int WBYTES = 3 * 3 * NG * COT * 16, NPIECE = WBYTES / 1024, PPW = (NPIECE + 3) / 4;  // DMA pieces per wave
    constexpr int XSL = 24 / UNITS;  // transform slots per unit (144 slots per chunk over taps 1..6)
    static_assert(COT == 64 || COT == 32, "co tile");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)smem;

    const int tid = threadIdx.x;
    const int lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int H = p.H, W = p.W;
    const int HW = H * W;
    const int nTw = (W + TW - 1) / TW, nTh = (H + TH - 1) / TH;
    const int nCoT = p.Cout / COT;
    int L = xcd_remap(blockIdx.x, gridDim.x);
    const int cot = L % nCoT;
    L /= nCoT;
    const int tw = L % nTw;
    L /= nTw;
    const int th = L % nTh;
    const int b = L / nTh;
    unsigned long long t0 = 0, t1 = 0, t2 = 0;
    if (p.prof) t0 = __builtin_amdgcn_s_memtime();

    // ---- x staging unit of this thread: one aligned quad (8 channels x 4 pixels) of one tile row ----
    // threads 0..191: interior quads, all four pixels stored; threads 192..215: the quad that holds a halo column
    // (left: its last pixel, right: its first), the other three pixels go to the dump column; threads 216..255 repeat
    // unit 215 (same data, same place).
    int s_row, s_g, gc;
    unsigned dsto[4];  // byte offset of pixel e's plane-0 entry within an x buffer
    if (tid < 192) {
        s_row = tid >> 5;
        s_g = (tid >> 4) & 1;
        gc = tw * TW + (tid & 15) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) dsto[e] = (unsigned)(((s_g * XR + s_row) * XS2 + 1 + (tid & 15) * 4 + e) * 16);
    } else {
        const int u = tid - 192 < 24 ? tid - 192 : 23;
        s_row = u >> 2;
        s_g = (u >> 1) & 1;
        const bool right = u & 1;
        gc = right ? tw * TW + TW : tw * TW - 4;
        const unsigned rowb = (unsigned)((s_g * XR + s_row) * XS2);
#pragma unroll
        for (int e = 0; e < 4; ++e) dsto[e] = (rowb + (right ? (e == 0 ? XS2 - 2 : XS2 - 1) : (e == 3 ? 0 : XS2 - 1))) * 16;
    }
    if (gc < 0) gc += W;
    while (gc >= W) gc -= W;  // azimuth is periodic; also covers tiles overhanging a narrow image
    const int gr = th * TH + s_row - 1;
    const bool s_ok = gr >= 0 && gr < H;  // rows outside [0,H) are zero padding (of the ACTIVATED tensor)
    const long s_goff = (long)s_g * 8 * HW + (s_ok ? gr * W + gc : 0);

    const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(p.w);
    const int nchunks = p.Cin / CK, nstages = 3 * nchunks;
    const int bmask = (1 << p.sign_shift) - 1;  // accumulation block = 2^sign_shift chunks (64 channels for the deep layers)
    const size_t wstage0 = (size_t)cot * nchunks * 3;
    const float* xb0 = p.x.p0 + b * p.x.bs0;
    const float* xb1 = p.x.p1 ? p.x.p1 + b * p.x.bs1 : p.x.p0;
    const int c0 = p.x.c0;
    const float* affb = PRO != PRO_NONE ? reinterpret_cast<const float*>(p.aff) + ((size_t)b * p.Cin + s_g * 8) * 2 : nullptr;

    f32x4 raw[8];       // 8 channels x 4 pixels
    f32x4 ad4[4];       // (a, d) of the 8 channels
    // Two transform streams run interleaved (X: pixels 0,1; Y: pixels 2,3), each with one channel pair in flight cut
    // into 14 micro-slices of two INDEPENDENT instructions, one micro-slice per unit: a dependent VALU chain inside
    // one MFMA shadow would stall the wave's in-order issue beyond it.
    // (plain scalars, not a struct array: hipcc parks an indexed struct array in scratch memory)
    float xv0 = 0.f, xv1 = 0.f, xm0 = 0.f, xm1 = 0.f, yv0 = 0.f, yv1 = 0.f, ym0 = 0.f, ym1 = 0.f;
    unsigned xpk[3][4], ypk[3][4];  // the three planes of the pixel being transformed (4 channel pairs each)

    // raw-load pieces (12 per chunk): i < 8 pixel quads, i >= 8 the folded GroupNorm affine
    const float* xq = nullptr;
    const f32x4* aq = nullptr;
    auto load_setup = [&](int ci0) __attribute__((always_inline)) {
        xq = (ci0 >= c0 ? xb1 + (long)(ci0 - c0) * HW : xb0 + (long)ci0 * HW) + s_goff;
        if (PRO != PRO_NONE) aq = reinterpret_cast<const f32x4*>(affb + (size_t)ci0 * 2);
    };
    auto load_piece = [&](int i) __attribute__((always_inline)) {
        if (i < 8)
            raw[i] = *reinterpret_cast<const f32x4*>(xq + (long)i * HW);
        else if (PRO != PRO_NONE)
            ad4[i - 8] = aq[i - 8];  // masked by mask_affine() at the consumer: a select on a just-loaded value makes hipcc
                                     // wait for it -- and for every older load -- on the spot, inside the MFMA stream
    };
    // zero padding of the ACTIVATED tensor (rows outside the image): a = d = 0 gives silu(0) = 0
    auto mask_affine = [&]() __attribute__((always_inline)) {
        if (PRO != PRO_NONE) {
#pragma unroll
            for (int j = 0; j < 4; ++j) ad4[j] = s_ok ? ad4[j] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    // micro-slice `sl` (0..13) of channel pair k = 4*pixel + pair (same arithmetic as silu_f() / split3_pk())
    auto xf = [&](float& qv0, float& qv1, float& qm0, float& qm1, unsigned (&qpk)[3][4], int k, int sl) __attribute__((always_inline)) {
        const int e = k >> 2, i2 = k & 3;
        constexpr bool silu = PRO == PRO_AFFINE_SILU;
        if (sl == 0) {
            qv0 = raw[2 * i2][e];
            qv1 = raw[2 * i2 + 1][e];
            if (PRO != PRO_NONE) {
                qv0 = qv0 * ad4[i2][0] + ad4[i2][1];
                qv1 = qv1 * ad4[i2][2] + ad4[i2][3];
            }
#ifdef R2DM_ACCURATE_SILU  // accuracy ablation (scripts/error_budget.py): libm exp + IEEE division instead of v_exp / v_rcp
        } else if (sl == 1) {
            if (silu) { qv0 = qv0 / (1.0f + expf(-qv0)); qv1 = qv1 / (1.0f + expf(-qv1)); }
        } else if (sl >= 2 && sl <= 5) {
#else
        } else if (sl == 1) {
            if (silu) { qm0 = qv0 * -1.4426950408889634f; qm1 = qv1 * -1.4426950408889634f; }
        } else if (sl == 2) {
            if (silu) { qm0 = __builtin_amdgcn_exp2f(qm0); qm1 = __builtin_amdgcn_exp2f(qm1); }
        } else if (sl == 3) {
            if (silu) { qm0 = 1.0f + qm0; qm1 = 1.0f + qm1; }
        } else if (sl == 4) {
            if (silu) { qm0 = __builtin_amdgcn_rcpf(qm0); qm1 = __builtin_amdgcn_rcpf(qm1); }
        } else if (sl == 5) {
            if (silu) { qv0 *= qm0; qv1 *= qm1; }
#endif
        } else if (sl == 6) {
            if (PRO == PRO_NONE) {
                qv0 = s_ok ? qv0 : 0.f;
                qv1 = s_ok ? qv1 : 0.f;
            }
        } else if (sl == 7) {
            qpk[0][i2] = cvt_pk_bf16(qv0, qv1);
        } else if (sl == 8) {
            qm0 = __uint_as_float(qpk[0][i2] << 16);
            qm1 = __uint_as_float(qpk[0][i2] & 0xffff0000u);
        } else if (sl == 9) {
            qv0 -= qm0;
            qv1 -= qm1;
        } else if (sl == 10) {
            qpk[1][i2] = cvt_pk_bf16(qv0, qv1);
        } else if (NPC == 3 && sl == 11) {
            qm0 = __uint_as_float(qpk[1][i2] << 16);
            qm1 = __uint_as_float(qpk[1][i2] & 0xffff0000u);
        } else if (NPC == 3 && sl == 12) {
            qv0 -= qm0;
            qv1 -= qm1;
        } else if (NPC == 3) {
            qpk[2][i2] = cvt_pk_bf16(qv0, qv1);
        }
    };
    auto xf_write = [&](const unsigned (&qpk)[3][4], unsigned char* buf, int e, int pl) __attribute__((always_inline)) {
        *reinterpret_cast<u32x4*>(buf + dsto[e] + pl * (XPL2 * 16)) = u32x4{qpk[pl][0], qpk[pl][1], qpk[pl][2], qpk[pl][3]};
    };
    // weights of stage s -> ring slot s % RING: NPIECE pieces of 1 KiB; every wave issues PPW (the surplus ones repeat the
    // last piece: same data, same place) so that the vmcnt bookkeeping is the same in all waves
    unsigned dma_v[PPW], dma_l[PPW];  // per-lane byte offset within a stage / LDS base of the piece (ring slot 0)
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        int j = wave + 4 * i;
        j = j < NPIECE ? j : NPIECE - 1;
        dma_v[i] = (unsigned)(j * 1024 + lane * 16);
        dma_l[i] = lds0 + WB0 + j * 1024;
    }
    const unsigned char* wtile = wsrc + wstage0 * WBYTES;
    auto dma_piece = [&](int s, int i) __attribute__((always_inline)) {
        dma16s(wtile + (size_t)s * WBYTES, dma_v[i], dma_l[i] + (unsigned)((s & (RING - 1)) * WBYTES));
    };

    unsigned xcur[NR], xnxt[NR];
#pragma unroll
    for (int n = 0; n < NR; ++n) {
        const int s = wave * NR + n;
        xcur[n] = lds0 + (unsigned)(((hi * XR + (s >> 1)) * XS2 + (s & 1) * 32 + l31) * 16);
        xnxt[n] = xcur[n] + XBYTES2;
    }
    const unsigned lds_w0 = lds0 + WB0 + (unsigned)((hi * CO_T + l31) * 16);

    // Two-level accumulation: every 64 input channels (576 products) acc is flushed into acc2 and restarts from C = 0, so
    // roundoff grows with sqrt(576), not sqrt(9 Cin).  Rounding-bias cancellation (file header) rides on it:
    // odd 64-channel blocks use pre-negated weights and are SUBTRACTED at the flush, so the accumulate-toward-minus-
    // infinity offset of consecutive blocks has opposite sign in acc2 (Cin/64 is even for every deep layer).
    f32x16 acc[MR][NR], acc2[MR][NR];
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int n = 0; n < NR; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = acc2[m][n][r] = 0.f;

    // ---- prologue: weight stages 0..2 in flight, chunk 0 transformed into x buffer 0, chunk 1's pixels requested ----
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int i = 0; i < PPW; ++i) dma_piece(s < nstages ? s : nstages - 1, i);
    load_setup(0);
#pragma unroll
    for (int i = 0; i < 12; ++i) load_piece(i);
    mask_affine();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
#pragma unroll
        for (int sl = 0; sl < 14; ++sl) xf(xv0, xv1, xm0, xm1, xpk, k, sl);
        if ((k & 3) == 3) {
#pragma unroll
            for (int pl = 0; pl < NPC; ++pl) xf_write(xpk, smem, k >> 2, pl);
        }
    }
    load_setup(nchunks > 1 ? CK : 0);
#pragma unroll
    for (int i = 0; i < 12; ++i) load_piece(i);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // (also chunk 1's pixels: once per block)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (p.prof) t1 = __builtin_amdgcn_s_memtime();

    u32x4 fa[2][3][MR], fb[2][3][NR];
    // fragment read r (0..11) of tap (ky, tx): plane r/4, then A m0, A m1, B n0, B n1
    auto frag1 = [&](const unsigned (&xb)[NR], unsigned wb, auto KY, auto TX, auto R, u32x4 (&a)[3][MR], u32x4 (&bb)[3][NR])
                     __attribute__((always_inline)) {
        constexpr int ky = decltype(KY)::value, tx = decltype(TX)::value, r = decltype(R)::value;
        constexpr int pl = r / (MR + NR), w = r % (MR + NR);
        if constexpr (r >= NFR) {
        } else if constexpr (w < MR)
            asm volatile("ds_read_b128 %0, %1 offset:%2"
                         : "=v"(a[pl][w])
                         : "v"(wb), "i"(pl * (3 * NG * CO_T * 16) + tx * (NG * CO_T * 16) + w * 512));
        else
            asm volatile("ds_read_b128 %0, %1 offset:%2"
                         : "=v"(bb[pl][w - MR])
                         : "v"(xb[w - MR]), "i"(pl * (XPL2 * 16) + ky * (XS2 * 16) + tx * 16));
    };
    {
        auto f0 = [&](auto R) __attribute__((always_inline)) { frag1(xcur, lds_w0, ic<0>{}, ic<0>{}, R, fa[0], fb[0]); };
        f0(ic<0>{}); f0(ic<1>{}); f0(ic<2>{}); f0(ic<3>{}); f0(ic<4>{}); f0(ic<5>{});
        f0(ic<6>{}); f0(ic<7>{}); f0(ic<8>{}); f0(ic<9>{}); f0(ic<10>{}); f0(ic<11>{});
    }

    // one tap = 24 units; T = tap within the chunk (ky = T / 3, tx = T % 3), PAR = parity of the chunk
    auto tap = [&](int c, auto TT, auto PAR) __attribute__((always_inline)) {
        constexpr int t = decltype(TT)::value, par = decltype(PAR)::value;
        constexpr int ky = t / 3, tx = t % 3, cur = (par * 9 + t) & 1;
        constexpr int kyn = t < 8 ? (t + 1) / 3 : 0, txn = t < 8 ? (t + 1) % 3 : 0;  // next tap
        const int sigma = 3 * c + ky;
        unsigned char* nbuf = smem + ((c + 1) & 1) * XBYTES2;  // x buffer being filled (chunk c+1)
        if (tx == 1) {
            // B_sigma: everybody is past tap (sigma, 0).  Before it: this wave's pieces of stage sigma+1 have landed
            // (only stage sigma+2's PPW may still be in flight) and its x-tile writes are done.
#ifndef X3S_NO_BARRIER  // (X3S_NO_*: ablation switches for scripts/build_variant.sh -- timing experiments, wrong results)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#else
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
            asm volatile("" ::: "memory");
            if (ky == 0) {
                // hipcc waits for the raw pixels (its own loads) at their first use with a vmcnt that knows nothing
                // about the DMA pieces issued below, i.e. it would wait for those too: take that wait here
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(raw[i]));
                if (PRO != PRO_NONE) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(ad4[j]));
                }
                mask_affine();  // (before this tap's first transform slice)
            }
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this tap's fragments (issued a tap ago)
        }
        const unsigned wbn = lds_w0 + (unsigned)(((t < 8 ? sigma + (kyn != ky ? 1 : 0) : sigma + 1) & (RING - 1)) * WBYTES);
        int sdma = sigma + 3;
        sdma = sdma < nstages ? sdma : nstages - 1;  // past the end: repeat the last stage (harmless, keeps vmcnt uniform)
        if (t == 7) load_setup((c + 2 < nchunks ? c + 2 : nchunks - 1) * CK);
        __builtin_amdgcn_sched_barrier(0);
        constexpr int PI[6] = {NPC == 3 ? 2 : 1, 0, NPC == 3 ? 1 : 0, 1, 0, 0}, PJ[6] = {0, NPC == 3 ? 2 : 1, NPC == 3 ? 1 : 0, 0, 1, 0};
#pragma unroll
        for (int i = 0; i < UNITS; ++i) {
            const int q = i / (MR * NR), m = (i / NR) % MR, n = i % NR;
            f32x16& ac = acc[m][n];
            const bf16x8 fra = __builtin_bit_cast(bf16x8, fa[cur][PI[q]][m]), frb = __builtin_bit_cast(bf16x8, fb[cur][PJ[q]][n]);
            if (t == 0 && i < MR * NR && (c & bmask) == 0) {  // first product of a block: start from zero
                const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                ac = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fra, frb, zero, 0, 0, 0);
            } else {
                ac = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fra, frb, ac, 0, 0, 0);
            }
            // ---- at most a handful of other instructions in the shadow of this MFMA ----
            if (i < NFR) {  // next tap's fragments (the next chunk's first tap reads the other x buffer)
                auto fr = [&](auto R) __attribute__((always_inline)) {
                    if (t < 8)
                        frag1(xcur, wbn, ic<kyn>{}, ic<txn>{}, R, fa[cur ^ 1], fb[cur ^ 1]);
                    else
                        frag1(xnxt, wbn, ic<0>{}, ic<0>{}, R, fa[cur ^ 1], fb[cur ^ 1]);
                };
                if (i == 0) fr(ic<0>{});
                if (i == 1) fr(ic<1>{});
                if (i == 2) fr(ic<2>{});
                if (i == 3) fr(ic<3>{});
                if (i == 4) fr(ic<4>{});
                if (i == 5) fr(ic<5>{});
                if (i == 6) fr(ic<6>{});
                if (i == 7) fr(ic<7>{});
                if (i == 8) fr(ic<8>{});
                if (i == 9) fr(ic<9>{});
                if (i == 10) fr(ic<10>{});
                if (i == 11) fr(ic<11>{});
            }
#ifndef X3S_NO_DMA
            if (tx == 1) {  // PPW pieces spread over the tap, into the ring slot of stage sigma-1
#pragma unroll
                for (int kp = 0; kp < PPW; ++kp)
                    if (i == (kp * UNITS) / PPW) dma_piece(sdma, kp);
            }
#endif
#ifndef X3S_NO_XF
            if (t >= 1 && t <= 6) {  // transform of chunk c+1: 144 slots; stream X pairs 0..7, stream Y pairs 8..15,
#pragma unroll
                for (int sub = 0; sub < XSL; ++sub) {
                    const int slot = ((t - 1) * UNITS + i) * XSL + sub, j = slot / 18, off = slot % 18;  // pair j in slots 18j..18j+17
                    if (off < 14) {
                        xf(xv0, xv1, xm0, xm1, xpk, j, off);
                        xf(yv0, yv1, ym0, ym1, ypk, 8 + j, off);
                    } else if ((j & 3) == 3 && off < 14 + NPC) {
                        xf_write(xpk, nbuf, j >> 2, off - 14);
                        xf_write(ypk, nbuf, (8 + j) >> 2, off - 14);
                    }
                }
            }
#endif
            if (t == 7) {  // raw pixels of the chunk after next: 12 loads over the last units of the tap
#pragma unroll
                for (int jl = 0; jl < LPU; ++jl) {
                    const int idx = i * LPU + jl - (UNITS * LPU - 12);
                    if (idx >= 0) load_piece(idx);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto chunk = [&](int c, auto PAR) __attribute__((always_inline)) {
        tap(c, ic<0>{}, PAR); tap(c, ic<1>{}, PAR); tap(c, ic<2>{}, PAR);
        tap(c, ic<3>{}, PAR); tap(c, ic<4>{}, PAR); tap(c, ic<5>{}, PAR);
        tap(c, ic<6>{}, PAR); tap(c, ic<7>{}, PAR); tap(c, ic<8>{}, PAR);
#ifndef X3S_NO_FLIP
        if ((c & bmask) == bmask) {  // flush (acc restarts from C = 0 in the next chunk: no zeroing pass)
            if (c & (bmask + 1)) {
#pragma unroll
                for (int m = 0; m < MR; ++m)
#pragma unroll
                    for (int n = 0; n < NR; ++n)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc2[m][n][r] -= acc[m][n][r];
            } else {
#pragma unroll
                for (int m = 0; m < MR; ++m)
#pragma unroll
                    for (int n = 0; n < NR; ++n)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc2[m][n][r] += acc[m][n][r];
            }
        }
#endif
#pragma unroll
        for (int n = 0; n < NR; ++n) {
            const unsigned tswap = xcur[n];
            xcur[n] = xnxt[n];
            xnxt[n] = tswap;
        }
    };

    for (int c = 0; c < nchunks; c += 2) {
        chunk(c, ic<0>{});
        chunk(c + 1, ic<1>{});
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // surplus DMA pieces / prefetched fragments must not outlive the block
    if (p.prof) t2 = __builtin_amdgcn_s_memtime();

#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int n = 0; n < NR; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;  // everything was flushed (Cin % 64 == 0)
    // whole tiles (every shape of the network): the wide epilogue through this wave's 1 KiB patch behind the weight ring
#ifndef R2DM_NO_WIDE_EPILOGUE
    if (H % TH == 0 && W % TW == 0)
#else
    if (false)
#endif
        conv_epilogue_wide<TH, TW, MR, NR, true>(p, acc, acc2, b, th, tw, nTw, cot * CO_T, wave, lane,
                                                 reinterpret_cast<float*>(smem + WB0 + RING * WBYTES) + wave * 256);
    else
        conv_epilogue<4, TH, TW, MR, NR, true>(p, acc, acc2, b, th, tw, nTw, cot * CO_T, wave, lane);

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

// ---- shallow variant of the stream kernel: two blocks per CU, single x tile, weight ring of two stages -----------
// Cin <= 128: K is short, so prologue / epilogue / chunk-boundary phases of one block must be covered by a second
// block on the same CU; 75 KB of LDS and 256 registers per block.  The tap stream (MFMA + fragment reads + DMA
// pieces + raw pixel loads interleaved unit by unit) is the one of conv_bf16x3_stream_kernel; the transform of the
// next chunk runs at the chunk boundary because the single x tile is free only then.
template <int PRO, int NPC>
__global__ __launch_bounds__(256, 2) void conv_bf16x3_pair_kernel(const ConvParams p) {
    using namespace x3s;
    constexpr int NPROD = NPC == 3 ? 6 : 3, UNITS = NPROD * MR * NR, NFR = NPC * (MR + NR);  // see conv_bf16x3_stream_kernel
    constexpr int LPU = (12 + UNITS - 1) / UNITS, PPW = 5;
    static_assert(NPC == 3 || NPC == 2, "pieces");
    constexpr int RING2 = 2, WB1 = XBYTES2;  // [x tile][weight ring of two stages]: 75456 bytes, two blocks per CU
    constexpr int NL = PRO != PRO_NONE ? 12 : 8;  // VMEM loads per chunk of raw pixels (+ folded affine)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)smem;

    const int tid = threadIdx.x;
    const int lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

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
    unsigned long long t0 = 0, t1 = 0, t2 = 0;
    if (p.prof) t0 = __builtin_amdgcn_s_memtime();

    // ---- x staging unit of this thread: one aligned quad (8 channels x 4 pixels) of one tile row ----
    // threads 0..191: interior quads, all four pixels stored; threads 192..215: the quad that holds a halo column
    // (left: its last pixel, right: its first), the other three pixels go to the dump column; threads 216..255 repeat
    // unit 215 (same data, same place).
    int s_row, s_g, gc;
    unsigned dsto[4];  // byte offset of pixel e's plane-0 entry within an x buffer
    if (tid < 192) {
        s_row = tid >> 5;
        s_g = (tid >> 4) & 1;
        gc = tw * TW + (tid & 15) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) dsto[e] = (unsigned)(((s_g * XR + s_row) * XS2 + 1 + (tid & 15) * 4 + e) * 16);
    } else {
        const int u = tid - 192 < 24 ? tid - 192 : 23;
        s_row = u >> 2;
        s_g = (u >> 1) & 1;
        const bool right = u & 1;
        gc = right ? tw * TW + TW : tw * TW - 4;
        const unsigned rowb = (unsigned)((s_g * XR + s_row) * XS2);
#pragma unroll
        for (int e = 0; e < 4; ++e) dsto[e] = (rowb + (right ? (e == 0 ? XS2 - 2 : XS2 - 1) : (e == 3 ? 0 : XS2 - 1))) * 16;
    }
    if (gc < 0) gc += W;
    while (gc >= W) gc -= W;  // azimuth is periodic; also covers tiles overhanging a narrow image
    const int gr = th * TH + s_row - 1;
    const bool s_ok = gr >= 0 && gr < H;  // rows outside [0,H) are zero padding (of the ACTIVATED tensor)
    const long s_goff = (long)s_g * 8 * HW + (s_ok ? gr * W + gc : 0);

    const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(p.w);
    const int nchunks = p.Cin / CK, nstages = 3 * nchunks;
    const size_t wstage0 = (size_t)cot * nchunks * 3;
    const float* xb0 = p.x.p0 + b * p.x.bs0;
    const float* xb1 = p.x.p1 ? p.x.p1 + b * p.x.bs1 : p.x.p0;
    const int c0 = p.x.c0;
    const float* affb = PRO != PRO_NONE ? reinterpret_cast<const float*>(p.aff) + ((size_t)b * p.Cin + s_g * 8) * 2 : nullptr;

    f32x4 raw[8];       // 8 channels x 4 pixels
    f32x4 ad4[4];       // (a, d) of the 8 channels
    // Two transform streams run interleaved (X: pixels 0,1; Y: pixels 2,3), each with one channel pair in flight cut
    // into 14 micro-slices of two INDEPENDENT instructions, one micro-slice per unit: a dependent VALU chain inside
    // one MFMA shadow would stall the wave's in-order issue beyond it.
    // (plain scalars, not a struct array: hipcc parks an indexed struct array in scratch memory)
    unsigned xpk[3][4];  // the three planes of the pixel being transformed (4 channel pairs each)

    // raw-load pieces (12 per chunk): i < 8 pixel quads, i >= 8 the folded GroupNorm affine
    const float* xq = nullptr;
    const f32x4* aq = nullptr;
    auto load_setup = [&](int ci0) __attribute__((always_inline)) {
        xq = (ci0 >= c0 ? xb1 + (long)(ci0 - c0) * HW : xb0 + (long)ci0 * HW) + s_goff;
        if (PRO != PRO_NONE) aq = reinterpret_cast<const f32x4*>(affb + (size_t)ci0 * 2);
    };
    auto load_piece = [&](int i) __attribute__((always_inline)) {
        if (i < 8)
            raw[i] = *reinterpret_cast<const f32x4*>(xq + (long)i * HW);
        else if (PRO != PRO_NONE)
            ad4[i - 8] = aq[i - 8];  // masked by mask_affine() at the consumer: a select on a just-loaded value makes hipcc
                                     // wait for it -- and for every older load -- on the spot, inside the MFMA stream
    };
    // zero padding of the ACTIVATED tensor (rows outside the image): a = d = 0 gives silu(0) = 0
    auto mask_affine = [&]() __attribute__((always_inline)) {
        if (PRO != PRO_NONE) {
#pragma unroll
            for (int j = 0; j < 4; ++j) ad4[j] = s_ok ? ad4[j] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    // micro-slice `sl` (0..13) of channel pair k = 4*pixel + pair (same arithmetic as silu_f() / split3_pk())
    auto xf = [&](float& qv0, float& qv1, float& qm0, float& qm1, unsigned (&qpk)[3][4], int k, int sl) __attribute__((always_inline)) {
        const int e = k >> 2, i2 = k & 3;
        constexpr bool silu = PRO == PRO_AFFINE_SILU;
        if (sl == 0) {
            qv0 = raw[2 * i2][e];
            qv1 = raw[2 * i2 + 1][e];
            if (PRO != PRO_NONE) {
                qv0 = qv0 * ad4[i2][0] + ad4[i2][1];
                qv1 = qv1 * ad4[i2][2] + ad4[i2][3];
            }
#ifdef R2DM_ACCURATE_SILU  // accuracy ablation (scripts/error_budget.py): libm exp + IEEE division instead of v_exp / v_rcp
        } else if (sl == 1) {
            if (silu) { qv0 = qv0 / (1.0f + expf(-qv0)); qv1 = qv1 / (1.0f + expf(-qv1)); }
        } else if (sl >= 2 && sl <= 5) {
#else
        } else if (sl == 1) {
            if (silu) { qm0 = qv0 * -1.4426950408889634f; qm1 = qv1 * -1.4426950408889634f; }
        } else if (sl == 2) {
            if (silu) { qm0 = __builtin_amdgcn_exp2f(qm0); qm1 = __builtin_amdgcn_exp2f(qm1); }
        } else if (sl == 3) {
            if (silu) { qm0 = 1.0f + qm0; qm1 = 1.0f + qm1; }
        } else if (sl == 4) {
            if (silu) { qm0 = __builtin_amdgcn_rcpf(qm0); qm1 = __builtin_amdgcn_rcpf(qm1); }
        } else if (sl == 5) {
            if (silu) { qv0 *= qm0; qv1 *= qm1; }
#endif
        } else if (sl == 6) {
            if (PRO == PRO_NONE) {
                qv0 = s_ok ? qv0 : 0.f;
                qv1 = s_ok ? qv1 : 0.f;
            }
        } else if (sl == 7) {
            qpk[0][i2] = cvt_pk_bf16(qv0, qv1);
        } else if (sl == 8) {
            qm0 = __uint_as_float(qpk[0][i2] << 16);
            qm1 = __uint_as_float(qpk[0][i2] & 0xffff0000u);
        } else if (sl == 9) {
            qv0 -= qm0;
            qv1 -= qm1;
        } else if (sl == 10) {
            qpk[1][i2] = cvt_pk_bf16(qv0, qv1);
        } else if (NPC == 3 && sl == 11) {
            qm0 = __uint_as_float(qpk[1][i2] << 16);
            qm1 = __uint_as_float(qpk[1][i2] & 0xffff0000u);
        } else if (NPC == 3 && sl == 12) {
            qv0 -= qm0;
            qv1 -= qm1;
        } else if (NPC == 3) {
            qpk[2][i2] = cvt_pk_bf16(qv0, qv1);
        }
    };
    auto xf_write = [&](const unsigned (&qpk)[3][4], unsigned char* buf, int e, int pl) __attribute__((always_inline)) {
        *reinterpret_cast<u32x4*>(buf + dsto[e] + pl * (XPL2 * 16)) = u32x4{qpk[pl][0], qpk[pl][1], qpk[pl][2], qpk[pl][3]};
    };
    // weights of stage s -> ring slot s % RING: 18 pieces of 1 KiB; every wave issues 5 (the surplus ones repeat piece
    // 17: same data, same place) so that the vmcnt bookkeeping is the same in all waves
    unsigned dma_v[5], dma_l[5];  // per-lane byte offset within a stage / LDS base of the piece (ring slot 0)
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        int j = wave + 4 * i;
        j = j < 18 ? j : 17;
        dma_v[i] = (unsigned)(j * 1024 + lane * 16);
        dma_l[i] = lds0 + WB1 + j * 1024;
    }
    const unsigned char* wtile = wsrc + wstage0 * WBYTES;
    auto dma_piece = [&](int s, int i) __attribute__((always_inline)) {
        dma16s(wtile + (size_t)s * WBYTES, dma_v[i], dma_l[i] + (unsigned)((s & (RING2 - 1)) * WBYTES));
    };

    unsigned xcur[NR];
#pragma unroll
    for (int n = 0; n < NR; ++n) {
        const int s = wave * NR + n;
        xcur[n] = lds0 + (unsigned)(((hi * XR + (s >> 1)) * XS2 + (s & 1) * 32 + l31) * 16);
    }
    const unsigned lds_w0 = lds0 + WB1 + (unsigned)((hi * CO_T + l31) * 16);

    f32x16 acc[MR][NR], accd[1][1];
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int n = 0; n < NR; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    // ---- prologue: weight stages 0 and 1 in flight, chunk 0 transformed into the x tile ----
    // whole tile at once (prologue, chunk boundaries): the four channel pairs of a pixel advance slice by slice together,
    // four independent dependency chains instead of one
    auto transform_all = [&]() __attribute__((always_inline)) {
        float v0[4], v1[4], m0[4], m1[4];
        mask_affine();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int sl = 0; sl < 14; ++sl)
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2) xf(v0[i2], v1[i2], m0[i2], m1[i2], xpk, 4 * e + i2, sl);
#pragma unroll
            for (int pl = 0; pl < NPC; ++pl) xf_write(xpk, smem, e, pl);
        }
    };
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 5; ++i) dma_piece(s < nstages ? s : nstages - 1, i);
    load_setup(0);
#pragma unroll
    for (int i = 0; i < 12; ++i) load_piece(i);
    transform_all();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (p.prof) t1 = __builtin_amdgcn_s_memtime();

    u32x4 fa[2][3][MR], fb[2][3][NR];
    // fragment read r (0..11) of tap (ky, tx): plane r/4, then A m0, A m1, B n0, B n1
    auto frag1 = [&](const unsigned (&xb)[NR], unsigned wb, auto KY, auto TX, auto R, u32x4 (&a)[3][MR], u32x4 (&bb)[3][NR])
                     __attribute__((always_inline)) {
        constexpr int ky = decltype(KY)::value, tx = decltype(TX)::value, r = decltype(R)::value;
        constexpr int pl = r / (MR + NR), w = r % (MR + NR);
        if constexpr (r >= NFR) {
        } else if constexpr (w < MR)
            asm volatile("ds_read_b128 %0, %1 offset:%2"
                         : "=v"(a[pl][w])
                         : "v"(wb), "i"(pl * (3 * NG * CO_T * 16) + tx * (NG * CO_T * 16) + w * 512));
        else
            asm volatile("ds_read_b128 %0, %1 offset:%2"
                         : "=v"(bb[pl][w - MR])
                         : "v"(xb[w - MR]), "i"(pl * (XPL2 * 16) + ky * (XS2 * 16) + tx * 16));
    };
    {
        auto f0 = [&](auto R) __attribute__((always_inline)) { frag1(xcur, lds_w0, ic<0>{}, ic<0>{}, R, fa[0], fb[0]); };
        f0(ic<0>{}); f0(ic<1>{}); f0(ic<2>{}); f0(ic<3>{}); f0(ic<4>{}); f0(ic<5>{});
        f0(ic<6>{}); f0(ic<7>{}); f0(ic<8>{}); f0(ic<9>{}); f0(ic<10>{}); f0(ic<11>{});
    }

    // one tap = 24 units (one MFMA + a thin slice of other work each, see conv_bf16x3_stream_kernel)
    auto tap = [&](int c, auto TT, auto PAR) __attribute__((always_inline)) {
        constexpr int t = decltype(TT)::value, par = decltype(PAR)::value;
        constexpr int ky = t / 3, tx = t % 3, cur = (par * 9 + t) & 1;
        constexpr int kyn = t < 8 ? (t + 1) / 3 : 0, txn = t < 8 ? (t + 1) % 3 : 0;  // next tap
        const int sigma = 3 * c + ky;
        if (tx == 2) {
            // B'_sigma: this wave's pieces of stage sigma+1 have landed (only the raw pixel loads issued during this chunk's
            // first tap may still be in flight) and every wave has its fragments of tap (sigma, 2) in registers: stage
            // sigma+1 may be read, ring slot sigma % 2 may be overwritten.
            if (ky == 0)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (ky == 2) {  // hipcc's own wait for the raw pixels would also wait for the DMA pieces issued below
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(raw[i]));
                if (PRO != PRO_NONE) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(ad4[j]));
                }
            }
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this tap's fragments (issued a tap ago)
        }
        const unsigned wbn = lds_w0 + (unsigned)(((kyn != ky ? sigma + 1 : sigma) & (RING2 - 1)) * WBYTES);
        int sdma = sigma + 2;
        sdma = sdma < nstages ? sdma : nstages - 1;  // past the end: repeat the last stage (harmless, keeps vmcnt uniform)
        if (t == 0) load_setup((c + 1 < nchunks ? c + 1 : nchunks - 1) * CK);
        __builtin_amdgcn_sched_barrier(0);
        constexpr int PI[6] = {NPC == 3 ? 2 : 1, 0, NPC == 3 ? 1 : 0, 1, 0, 0}, PJ[6] = {0, NPC == 3 ? 2 : 1, NPC == 3 ? 1 : 0, 0, 1, 0};
#pragma unroll
        for (int i = 0; i < UNITS; ++i) {
            const int q = i / (MR * NR), m = (i / NR) % MR, n = i % NR;
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[cur][PI[q]][m]),
                                                                __builtin_bit_cast(bf16x8, fb[cur][PJ[q]][n]), acc[m][n], 0, 0, 0);
            if (i < NFR && t < 8) {  // next tap's fragments (the next chunk's first tap waits for the new x tile)
                auto fr = [&](auto R) __attribute__((always_inline)) {
                    frag1(xcur, wbn, ic<kyn>{}, ic<txn>{}, R, fa[cur ^ 1], fb[cur ^ 1]);
                };
                if (i == 0) fr(ic<0>{});
                if (i == 1) fr(ic<1>{});
                if (i == 2) fr(ic<2>{});
                if (i == 3) fr(ic<3>{});
                if (i == 4) fr(ic<4>{});
                if (i == 5) fr(ic<5>{});
                if (i == 6) fr(ic<6>{});
                if (i == 7) fr(ic<7>{});
                if (i == 8) fr(ic<8>{});
                if (i == 9) fr(ic<9>{});
                if (i == 10) fr(ic<10>{});
                if (i == 11) fr(ic<11>{});
            }
            if (tx == 2) {  // PPW pieces spread over the tap, into the ring slot of stage sigma
#pragma unroll
                for (int kp = 0; kp < PPW; ++kp)
                    if (i == (kp * UNITS) / PPW) dma_piece(sdma, kp);
            }
            if (t == 0) {  // raw pixels of the next chunk: 12 loads over the last units of the tap
#pragma unroll
                for (int jl = 0; jl < LPU; ++jl) {
                    const int idx = i * LPU + jl - (UNITS * LPU - 12);
                    if (idx >= 0) load_piece(idx);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto chunk = [&](int c, auto PAR) __attribute__((always_inline)) {
        constexpr int par = decltype(PAR)::value;
        tap(c, ic<0>{}, PAR); tap(c, ic<1>{}, PAR); tap(c, ic<2>{}, PAR);
        tap(c, ic<3>{}, PAR); tap(c, ic<4>{}, PAR); tap(c, ic<5>{}, PAR);
        tap(c, ic<6>{}, PAR); tap(c, ic<7>{}, PAR); tap(c, ic<8>{}, PAR);
        // rounding-bias cancellation (file header): the accumulator changes sign after every second chunk and chunk
        // pairs 1, 3, ... use pre-negated weights: acc = (-1)^(c/2) * S_c (Cin/32 is even)
        if (((c + 1) & ((1 << p.sign_shift) - 1)) == 0) {
#pragma unroll
            for (int m = 0; m < MR; ++m)
#pragma unroll
                for (int n = 0; n < NR; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[m][n][r] = -acc[m][n][r];
        }
        if (c + 1 < nchunks) {
            // chunk boundary: the single x tile is rewritten while the other block of this CU owns the matrix pipe
            __builtin_amdgcn_s_barrier();  // every wave has its last fragments of this chunk in registers
            asm volatile("" ::: "memory");
#ifndef X3P_NO_XF
            transform_all();
#endif
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            constexpr int nb = ((par * 9 + 8) & 1) ^ 1;
            const unsigned wb0 = lds_w0 + (unsigned)(((3 * c + 3) & (RING2 - 1)) * WBYTES);
            auto f0 = [&](auto R) __attribute__((always_inline)) { frag1(xcur, wb0, ic<0>{}, ic<0>{}, R, fa[nb], fb[nb]); };
            f0(ic<0>{}); f0(ic<1>{}); f0(ic<2>{}); f0(ic<3>{}); f0(ic<4>{}); f0(ic<5>{});
            f0(ic<6>{}); f0(ic<7>{}); f0(ic<8>{}); f0(ic<9>{}); f0(ic<10>{}); f0(ic<11>{});
        }
    };

    for (int c = 0; c < nchunks; c += 2) {
        chunk(c, ic<0>{});
        chunk(c + 1, ic<1>{});
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // surplus DMA pieces / prefetched fragments must not outlive the block
    if (p.prof) t2 = __builtin_amdgcn_s_memtime();

#ifndef R2DM_NO_WIDE_EPILOGUE
    if (H % TH == 0 && W % TW == 0)  // whole tiles: wide epilogue, this wave's 1 KiB patch behind the weight ring
#else
    if (false)
#endif
        conv_epilogue_wide<TH, TW, MR, NR, false>(p, acc, accd, b, th, tw, nTw, cot * CO_T, wave, lane,
                                                  reinterpret_cast<float*>(smem + WB1 + RING2 * WBYTES) + wave * 256);
    else
        conv_epilogue<4, TH, TW, MR, NR, false>(p, acc, accd, b, th, tw, nTw, cot * CO_T, wave, lane);

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
                                        int Cin, long total, int sign_shift, int CO_T) {
    using x3::CK;
    using x3::NG;
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
        if ((c >> sign_shift) & 1) v = -v;  // rounding-bias cancellation: chunk pairs (shallow) / 64-channel blocks (deep) of alternating sign
        split3_pk(v, v, u[0], u[1], u[2]);
        dst[i] = (unsigned short)(u[pl] & 0xffffu);
    }
}


// long reductions (K = 9*Cin > 1152): two-level accumulation, stream kernel, block-wise sign pattern
static bool conv_bf16x3_deep(int Cin) { return Cin > 128; }
// chunks per sign block = 2^shift: shallow kernel flips every 2 chunks; stream kernel: 64-channel blocks (32 for Cin = 64)
static int conv_bf16x3_sign_shift(int Cin) { return !conv_bf16x3_deep(Cin) ? 1 : Cin >= 128 ? 2 : 1; }

// 64 output channels per block unless that leaves CUs idle (fewer than 256 blocks) and the deep kernel applies
int conv_bf16x3_co_tile(int Cin, int Cout, long px_batch) {
    const long nblk = (long)(Cout / 64) * ((px_batch + 255) / 256);
    return conv_bf16x3_deep(Cin) && nblk < 256 ? 32 : 64;
}

bool conv_bf16x3_supported(int Cin, int Cout, int taps) {
    // an even number of 32-channel chunk pairs (shallow) / of 64-channel blocks (deep): the sign pattern must balance
    return taps == 9 && Cout % x3::CO_T == 0 && Cin % (32 << conv_bf16x3_sign_shift(Cin)) == 0;
}

long conv_bf16x3_packed_floats(int Cin, int Cout) { return (long)Cout * Cin * 9 * 3 / 2; }

hipError_t launch_pack_conv_bf16x3(const float* w, float* dst, int Cout, int Cin, int co_tile, hipStream_t s) {
    if (co_tile != 64 && co_tile != 32) return hipErrorInvalidValue;
    const long total = (long)Cout * Cin * 9 * 3;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    pack_conv_bf16x3_kernel<<<blocks, 256, 0, s>>>(w, reinterpret_cast<unsigned short*>(dst), Cout, Cin, total,
                                                   conv_bf16x3_sign_shift(Cin), co_tile);
    return hipGetLastError();
}

template <int PRO, int NPC>
static hipError_t launch_x3_pair(const ConvParams& p, hipStream_t s) {
    auto kern = conv_bf16x3_pair_kernel<PRO, NPC>;
    constexpr int lds = x3s::XBYTES2 + 2 * x3::WBYTES + 4096;  // + four 1 KiB epilogue patches: 79552 B, two blocks per CU
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int nTw = (p.W + 63) / 64, nTh = (p.H + 3) / 4, nCoT = p.Cout / x3::CO_T;
    const long nblk = (long)nCoT * nTw * nTh * p.B;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), lds, s, p);
    return hipGetLastError();
}

template <int PRO, int COT, int NPC>
static hipError_t launch_x3_stream(const ConvParams& p, hipStream_t s) {
    auto kern = conv_bf16x3_stream_kernel<PRO, COT, NPC>;
    constexpr int lds = x3s::WB0 + x3s::RING * (3 * 3 * x3::NG * COT * 16) + 4096;  // + four 1 KiB epilogue patches
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int nTw = (p.W + 63) / 64, nTh = (p.H + 3) / 4, nCoT = p.Cout / COT;
    const long nblk = (long)nCoT * nTw * nTh * p.B;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), lds, s, p);
    return hipGetLastError();
}

hipError_t launch_conv_bf16x3(const ConvParams& p, hipStream_t s) {
    if (!conv_bf16x3_supported(p.Cin, p.Cout, p.taps)) return hipErrorInvalidValue;
    if (p.co_tile != 64 && !(p.co_tile == 32 && conv_bf16x3_deep(p.Cin))) return hipErrorInvalidValue;
    if (p.x.p1 && p.x.c0 % x3::CK) return hipErrorInvalidValue;  // a chunk must not straddle the concat seam
    if (p.prologue != PRO_NONE && p.aff == nullptr) return hipErrorInvalidValue;
    if (p.H * (long)p.W * 16 >= (1L << 31) || p.W % 4) return hipErrorInvalidValue;
    const bool deep = conv_bf16x3_deep(p.Cin);
    ConvParams q = p;
    q.sign_shift = conv_bf16x3_sign_shift(p.Cin);
    if (p.pieces != 3) return hipErrorInvalidValue;  // (the two-piece variant of round 1 is superseded by conv_f16x2.hip)
#define X3_DISPATCH(PRO_)                                                                        \
    return !deep ? launch_x3_pair<PRO_, 3>(q, s)                                                 \
           : p.co_tile == 32 ? launch_x3_stream<PRO_, 32, 3>(q, s) : launch_x3_stream<PRO_, 64, 3>(q, s)
    switch (p.prologue) {
        case PRO_NONE: X3_DISPATCH(PRO_NONE);
        case PRO_AFFINE: X3_DISPATCH(PRO_AFFINE);
        case PRO_AFFINE_SILU: X3_DISPATCH(PRO_AFFINE_SILU);
    }
#undef X3_DISPATCH
    return hipErrorInvalidValue;
}

}  // namespace r2dm

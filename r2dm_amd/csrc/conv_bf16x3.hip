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
    // NPC = 3: exact three-piece split, six products (fp32-class error).  NPC = 2 (reduced-precision mode, SURVEY.md
    // section 8 (f).3): two pieces = 16 mantissa bits per operand, three products, error ~2^-16 relative.
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

// ---- shallow variant, second form ("duo"): two tiles per block in counter-phase, persistent ---------------------
// One block = 8 waves = two GROUPS of four; every group owns one 64 co x (4 x 64) pixel tile exactly like a block of
// conv_bf16x3_pair_kernel, and the two groups of a block work on horizontally adjacent tiles of the SAME output
// channel tile, one 16-channel chunk apart:
//
//     phase       2q                    2q+1                  2q+2                 ...
//     leader      MFMA chunk q          other(q -> q+1)       MFMA chunk q+1
//     follower    other(q-1 -> q)       MFMA chunk q          other(q -> q+1)
//
// "other" = everything that is not a matrix instruction: the GroupNorm-affine + SiLU + three-way split of the group's
// next chunk into its (single) x tile, the raw pixel loads of the chunk after that, the sign flip of the rounding-bias
// cancellation and, when a tile is finished, its epilogue (bias / residual / scale / store / fp64 GroupNorm statistics)
// plus the move to the block's next tile -- the block is persistent and walks its share of the launch's tile pairs.
// Two co-resident blocks of the pair kernel overlap these phases only by chance (PMC: matrix pipe 48 % busy); here every
// SIMD holds one wave of each group and the groups alternate by construction, so a SIMD's matrix pipe always has a
// wave in its MFMA stream while the other wave's VALU / LDS / memory work runs beside it.
//   * weights: ONE ring of four (chunk, kernel row) stages serves both groups (the follower reads a stage three
//     segments after the leader).  The follower refills slot sigma % 4 with stage sigma + 4 by LDS-DMA from its MFMA
//     stream right after the barrier that retires its own reads of stage sigma; the leader's stream carries no memory
//     instruction at all.
//   * three block-wide barriers per phase, inside the computing group's stream before the last tap of each kernel row
//     (as in the pair kernel); the other group's work is cut into three parts around them.  They publish the weight
//     stages, retire ring slots, and publish the other group's freshly written x tile to its own four waves.
namespace x3d {
using namespace x3s;
constexpr int RINGD = 4;
constexpr int WBD = 2 * XBYTES2;                     // [x tile of group 0][x tile of group 1][weight ring]
constexpr int LDS_BYTES = WBD + RINGD * x3::WBYTES;  // 150912
constexpr int LDS_TOTAL = LDS_BYTES + 8 * 1024;        // + one 1 KiB transposition patch per wave (epilogue)
}  // namespace x3d

template <int PRO, int NPC>
__global__ __launch_bounds__(512, 2) void conv_bf16x3_duo_kernel(const ConvParams p, const int total_items) {
    using namespace x3d;
    constexpr int NPROD = NPC == 3 ? 6 : 3, UNITS = NPROD * MR * NR, NFR = NPC * (MR + NR), PPW = 5;
    static_assert(NPC == 3 || NPC == 2, "pieces");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)smem;

    const int tid = threadIdx.x;
    const int lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave8 >> 2, wave = wave8 & 3, t4 = tid & 255;
    const bool follower = grp == 1;
#ifdef DUO_PROF  // timeline probe (scripts/duo_timeline.py): wave 0 of each group of block 0 stamps s_memtime at phase events
    int prof_i = 0;
    auto stamp = [&](int code) __attribute__((always_inline)) {
        if (p.prof && blockIdx.x == 0 && wave == 0) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (lane == 0 && prof_i < 1020) p.prof[grp * 1024 + prof_i] = (t << 8) | (unsigned)code;
            ++prof_i;
        }
    };
#else
    auto stamp = [&](int) __attribute__((always_inline)) {};
#endif

    const int H = p.H, W = p.W;
    const int HW = H * W;
    const int nTw = (W + TW - 1) / TW, nTh = (H + TH - 1) / TH;
    const int nTiles = nTw * nTh * p.B;
    const int nCoT = p.Cout / CO_T;
    const int nchunks = p.Cin / CK, nst = 3 * nchunks;
    const int G = gridDim.x;
    const int nIt = (total_items - (int)blockIdx.x + G - 1) / G;  // tile pairs of this block (grid <= total_items)
    const int Q = nIt * nchunks;                                   // chunks each group multiplies
    const int c0 = p.x.c0;

    // item `it` of this block -> output channel tile and this group's pixel tile (an odd tile count: the last pair
    // holds the same tile twice; both groups then compute and store identical values)
    auto decode = [&](int it, int& cot, int& b, int& th, int& tw) __attribute__((always_inline)) {
        const int L = xcd_remap((int)blockIdx.x + it * G, total_items);
        cot = L % nCoT;
        int t = 2 * (L / nCoT) + grp;
        t = t < nTiles ? t : nTiles - 1;
        // (integer division runs on the vector ALU: tell the compiler the results are wave-uniform, or every address
        // derived from them becomes per-lane 64-bit arithmetic)
        cot = __builtin_amdgcn_readfirstlane(cot);
        tw = __builtin_amdgcn_readfirstlane(t % nTw);
        t /= nTw;
        th = __builtin_amdgcn_readfirstlane(t % nTh);
        b = __builtin_amdgcn_readfirstlane(t / nTh);
    };

    // ---- x staging unit of this thread within its group: one aligned quad (8 channels x 4 pixels) of one tile row ----
    // (t4 < 192: interior quads; 192..215: the quad holding a halo column, the three pixels it does not need go to the
    // dump column; 216..255 repeat unit 215) -- see conv_bf16x3_stream_kernel
    int s_row, s_g, s_col;
    unsigned dsto[4];
    if (t4 < 192) {
        s_row = t4 >> 5;
        s_g = (t4 >> 4) & 1;
        s_col = (t4 & 15) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) dsto[e] = (unsigned)(((s_g * XR + s_row) * XS2 + 1 + (t4 & 15) * 4 + e) * 16);
    } else {
        const int u = t4 - 192 < 24 ? t4 - 192 : 23;
        s_row = u >> 2;
        s_g = (u >> 1) & 1;
        const bool right = u & 1;
        s_col = right ? TW : -4;
        const unsigned rowb = (unsigned)((s_g * XR + s_row) * XS2);
#pragma unroll
        for (int e = 0; e < 4; ++e) dsto[e] = (rowb + (right ? (e == 0 ? XS2 - 2 : XS2 - 1) : (e == 3 ? 0 : XS2 - 1))) * 16;
    }
    unsigned char* const xtile = smem + grp * XBYTES2;

    // ---- load side: one chunk ahead of the multiplication, possibly already in the block's next tile ----
    // prep_load() (between two multiplications) fixes the addresses of the next chunk to fetch; its twelve 16-byte loads
    // are issued from the last tap of the MFMA stream, where the second fragment buffer is idle, and land during the
    // first part of between().
    int l_item = 0, l_c = 0;
    const float* l_x0 = nullptr;
    const float* l_x1 = nullptr;
    const float* l_aff = nullptr;
    long l_goff = 0;
    bool l_ok = false;
    auto set_load_item = [&](int it) __attribute__((always_inline)) {
        int cot, b, th, tw;
        decode(it, cot, b, th, tw);
        int gc = tw * TW + s_col;
        if (gc < 0) gc += W;
        while (gc >= W) gc -= W;  // azimuth is periodic; also covers tiles overhanging a narrow image
        const int gr = th * TH + s_row - 1;
        l_ok = gr >= 0 && gr < H;  // rows outside [0,H) are zero padding (of the ACTIVATED tensor)
        l_goff = (long)s_g * 8 * HW + (l_ok ? gr * W + gc : 0);
        l_x0 = p.x.p0 + b * p.x.bs0;
        l_x1 = p.x.p1 ? p.x.p1 + b * p.x.bs1 : p.x.p0;
        if (PRO != PRO_NONE) l_aff = reinterpret_cast<const float*>(p.aff) + ((size_t)b * p.Cin + s_g * 8) * 2;
    };
    f32x4 raw[8];  // 8 channels x 4 pixels of the chunk to be transformed next
    f32x4 ad4[4];  // (a, d) of its 8 channels
    using gcf4 = const f32x4 __attribute__((address_space(1)))*;
    const float __attribute__((address_space(1)))* xq = nullptr;  // this thread's quad in the chunk's first channel
    gcf4 aq = nullptr;                                            // folded GroupNorm affine of the chunk's channels
    bool q_ok = false, raw_ok = false;
    auto prep_load = [&]() __attribute__((always_inline)) {
        const int ci0 = l_c * CK;
        xq = (const float __attribute__((address_space(1)))*)((ci0 >= c0 ? l_x1 + (long)(ci0 - c0) * HW : l_x0 + (long)ci0 * HW) + l_goff);
        if (PRO != PRO_NONE) aq = (gcf4)(l_aff + (size_t)ci0 * 2);
        q_ok = l_ok;
        if (l_c + 1 < nchunks)
            ++l_c;
        else if (l_item + 1 < nIt) {
            l_c = 0;
            set_load_item(++l_item);
        }  // (past the end: the last chunk again)
    };
    auto load_piece = [&](int i) __attribute__((always_inline)) {  // i < 8: pixel quads, i >= 8: the affine
        if (i < 8)
            raw[i] = *(gcf4)(xq + (long)i * HW);
        else if (PRO != PRO_NONE)
            ad4[i - 8] = aq[i - 8];  // (NOT masked here: a select on a just-loaded value makes hipcc wait for it on the spot)
        if (i == 11) raw_ok = q_ok;
    };

    // ---- transform: affine, SiLU, exact three-way split, pack (same arithmetic as conv_bf16x3_pair_kernel) ----
    unsigned xpk[3][4];
    auto xf = [&](float& qv0, float& qv1, float& qm0, float& qm1, int k, int sl) __attribute__((always_inline)) {
        const int e = k >> 2, i2 = k & 3;
        constexpr bool silu = PRO == PRO_AFFINE_SILU;
        if (sl == 0) {
            qv0 = raw[2 * i2][e];
            qv1 = raw[2 * i2 + 1][e];
            if (PRO != PRO_NONE) {
                qv0 = qv0 * ad4[i2][0] + ad4[i2][1];
                qv1 = qv1 * ad4[i2][2] + ad4[i2][3];
            }
#ifdef R2DM_ACCURATE_SILU
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
                qv0 = raw_ok ? qv0 : 0.f;
                qv1 = raw_ok ? qv1 : 0.f;
            }
        } else if (sl == 7) {
            xpk[0][i2] = cvt_pk_bf16(qv0, qv1);
        } else if (sl == 8) {
            qm0 = __uint_as_float(xpk[0][i2] << 16);
            qm1 = __uint_as_float(xpk[0][i2] & 0xffff0000u);
        } else if (sl == 9) {
            qv0 -= qm0;
            qv1 -= qm1;
        } else if (sl == 10) {
            xpk[1][i2] = cvt_pk_bf16(qv0, qv1);
        } else if (NPC == 3 && sl == 11) {
            qm0 = __uint_as_float(xpk[1][i2] << 16);
            qm1 = __uint_as_float(xpk[1][i2] & 0xffff0000u);
        } else if (NPC == 3 && sl == 12) {
            qv0 -= qm0;
            qv1 -= qm1;
        } else if (NPC == 3) {
            xpk[2][i2] = cvt_pk_bf16(qv0, qv1);
        }
    };
    // pixels 2*half, 2*half+1 of the thread's quad: the four channel pairs of a pixel advance slice by slice together
    auto transform_half = [&](int half) __attribute__((always_inline)) {
        float v0[4], v1[4], m0[4], m1[4];
        if (PRO != PRO_NONE && half == 0) {  // zero padding of the ACTIVATED tensor: a = d = 0 gives silu(0) = 0
#pragma unroll
            for (int j = 0; j < 4; ++j) ad4[j] = raw_ok ? ad4[j] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int e = 2 * half; e < 2 * half + 2; ++e) {
#pragma unroll
            for (int sl = 0; sl < 14; ++sl)
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2) xf(v0[i2], v1[i2], m0[i2], m1[i2], 4 * e + i2, sl);
#pragma unroll
            for (int pl = 0; pl < NPC; ++pl)
                *reinterpret_cast<u32x4*>(xtile + dsto[e] + pl * (XPL2 * 16)) = u32x4{xpk[pl][0], xpk[pl][1], xpk[pl][2], xpk[pl][3]};
        }
    };

    // ---- weights: DMA cursor of the follower (stage within the item's co tile; items may change co tile) ----
    const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(p.w);
    // 18 pieces of 1 KiB per stage; wave w issues pieces w, w+4, ..., the surplus ones repeat piece 17 (same data, same
    // place).  Everything but the lane offset is wave-uniform: source and LDS base go through SGPRs.
    const unsigned lane16 = (unsigned)lane * 16;
    auto piece_of = [&](int i) __attribute__((always_inline)) { const int j = wave + 4 * i; return j < 18 ? j : 17; };
    int d_item = 0, d_s = 0;
    const unsigned char* d_base = nullptr;
    auto set_dma_item = [&](int it) __attribute__((always_inline)) {
        int cot, b, th, tw;
        decode(it, cot, b, th, tw);
        d_base = wsrc + (size_t)cot * nst * WBYTES;
    };
    // source of the cursor's stage; advances the cursor (past the end: the last stage again -- harmless, keeps the
    // vmcnt bookkeeping uniform)
    auto dma_next = [&]() __attribute__((always_inline)) -> const unsigned char* {
        const unsigned char* s = d_base + (size_t)d_s * WBYTES;
        if (d_s + 1 < nst)
            ++d_s;
        else if (d_item + 1 < nIt) {
            d_s = 0;
            set_dma_item(++d_item);
        }
        return s;
    };

    // the cursor's stage -> ring slot `slot` (this wave's five pieces)
    auto dma_stage = [&](int slot) __attribute__((always_inline)) {
        const unsigned long long sv = (unsigned long long)dma_next();  // wave-uniform by construction; say so (SGPR operand)
        const unsigned char* src = (const unsigned char*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(sv >> 32)) << 32) |
                                                          (unsigned)__builtin_amdgcn_readfirstlane((int)sv));
#pragma unroll
        for (int i = 0; i < PPW; ++i)
            dma16s(src + piece_of(i) * 1024, lane16, lds0 + WBD + (unsigned)((slot & (RINGD - 1)) * WBYTES + piece_of(i) * 1024));
    };

    unsigned xcur[NR];
#pragma unroll
    for (int n = 0; n < NR; ++n) {
        const int s = wave * NR + n;
        xcur[n] = lds0 + grp * XBYTES2 + (unsigned)(((hi * XR + (s >> 1)) * XS2 + (s & 1) * 32 + l31) * 16);
    }
    const unsigned lds_w0 = lds0 + WBD + (unsigned)((hi * CO_T + l31) * 16);

    f32x16 acc[MR][NR];

    // ---- epilogue of the tile this group has just finished, in pieces that fit between the block's barriers ----
    int e_item = 0, e_cot, e_b, e_th, e_tw;
    decode(0, e_cot, e_b, e_th, e_tw);
    const float sc = p.scale ? *p.scale : 1.0f;
    // Quarter passes (32 output channels x one 32-pixel segment = one accumulator tile).  The MFMA layout gives a lane
    // ONE pixel of 16 channels, i.e. 4-byte global accesses 256 B per instruction -- and a CU retires those at a few
    // bytes per cycle (measured: the epilogue of a tile took 2.5 multiplication phases).  Every 8-channel block is
    // therefore turned through a private 1 KiB LDS patch (4 ds_write_b32 + 1 ds_read_b128, no barrier: LDS operations of
    // one wave execute in order) into "4 consecutive pixels of one channel per lane": residual loads and output stores
    // are 16 bytes per lane, 1 KiB per instruction, a quarter as many.
    double st_s[4], st_q[4];  // statistics of the 32-channel half being finished
    f32x4 rvA[4], rvB[4];  // residual values of two quarters in flight
    using gcf = const float __attribute__((address_space(1)))*;
    using gcf4x = const f32x4 __attribute__((address_space(1)))*;
    using gf4x = f32x4 __attribute__((address_space(1)))*;
    // explicit global address space: a pointer that reaches a load through a phi is otherwise accessed with FLAT
    // instructions, which count on lgkmcnt as well and turn every LDS wait into a wait for HBM
    gf4x yu = nullptr;
    gcf4x ru = nullptr;
    float* const escr = reinterpret_cast<float*>(smem + LDS_BYTES) + wave8 * 256;  // this wave's transposition patch
    const int tq_c = lane >> 3, tq_p = (lane & 7) * 4;  // after the turn: channel within the block, first of 4 pixels
    int e_loff[NR];  // whole tiles only (launcher: H % 4 == 0, W % 64 == 0): no pixel predication
    auto epi_begin = [&]() __attribute__((always_inline)) {
        const int co_u = e_cot * CO_T;  // Cout % 64 == 0: every channel of the tile exists
        yu = (gf4x)(p.y + e_b * p.y_bs + (long)co_u * HW);
        ru = (gcf4x)(p.res + e_b * p.res_bs + (long)co_u * HW);  // (only dereferenced if p.res)
#pragma unroll
        for (int n = 0; n < NR; ++n) {
            const int s = wave * NR + n;  // offsets in units of 4 floats
            e_loff[n] = (tq_c * HW + (e_th * TH + (s >> 1)) * W + e_tw * TW + (s & 1) * 32 + tq_p) >> 2;
        }
    };
    auto epi_zero_stats = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int k8 = 0; k8 < 4; ++k8) st_s[k8] = st_q[k8] = 0.0;
    };
    auto epi_load = [&](auto M, auto N, f32x4 (&rv)[4]) __attribute__((always_inline)) {  // residual values of one quarter
        constexpr int m = decltype(M)::value, n = decltype(N)::value;
        if (p.res) {
#pragma unroll
            for (int k8 = 0; k8 < 4; ++k8) rv[k8] = (ru + (long)(m * 32 + k8 * 8) * (HW >> 2))[e_loff[n]];
        }
    };
    float ebias[MR][4];  // bias of this lane's channel in each 8-channel block (loaded with the first residual values)
    auto epi_bias = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int k8 = 0; k8 < 4; ++k8) ebias[m][k8] = ((gcf)p.bias)[e_cot * CO_T + m * 32 + k8 * 8 + tq_c];
    };
    auto epi_finish = [&](auto M, auto N, f32x4 (&rv)[4]) __attribute__((always_inline)) {
        constexpr int m = decltype(M)::value, n = decltype(N)::value;
        f32x4 t[4];
#pragma unroll
        for (int k8 = 0; k8 < 4; ++k8) {  // turn the four blocks through the patch back to back (in-order LDS: no waits between)
#pragma unroll
            for (int j = 0; j < 4; ++j) escr[(j + 4 * hi) * 32 + l31] = acc[m][n][4 * k8 + j];
            t[k8] = *reinterpret_cast<const f32x4*>(escr + tq_c * 32 + tq_p);
        }
#pragma unroll
        for (int k8 = 0; k8 < 4; ++k8) {
            f32x4 v = t[k8] + ebias[m][k8];
            if (p.res) v = rv[k8] + v;
            if (p.scale) v *= sc;
            (yu + (long)(m * 32 + k8 * 8) * (HW >> 2))[e_loff[n]] = v;
            if (p.stat) {
                // GroupNorm statistics of the output: four pixels are summed in fp32 (3 + 4 roundings of ~6e-8, unbiased
                // and independent from lane to lane: they average out over the >= 10^4 lanes x tiles of a group),
                // everything beyond that in fp64
                const float s4 = (v[0] + v[1]) + (v[2] + v[3]);
                const float q4 = fmaf(v[3], v[3], fmaf(v[2], v[2], fmaf(v[1], v[1], v[0] * v[0])));
                st_s[k8] += (double)s4;
                st_q[k8] += (double)q4;
            }
        }
    };
    auto set_zero = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int n = 0; n < NR; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    };
    auto epi_stats = [&](auto M) __attribute__((always_inline)) {  // both quarters of half M are finished
        constexpr int m = decltype(M)::value;
        if (p.stat) epi_stat_write_bfly8(p, st_s, st_q, e_b, e_th, e_tw, nTw, e_cot * CO_T + m * 32, wave, lane);
    };
    auto epi_end = [&]() __attribute__((always_inline)) {
        if (++e_item < nIt) decode(e_item, e_cot, e_b, e_th, e_tw);
        set_zero();
    };

    set_zero();

    // ---- fragments ----
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
    // all fragments of tap (0, 0) of chunk q (the tile and the stage must be published).  Taps alternate between the
    // two fragment buffers by t & 1: tap 8 and the next chunk's tap 0 both use buffer 0, and nothing is prefetched
    // across the chunk boundary (the x tile is rewritten there), so one instance of the tap stream serves every chunk.
    auto frag_first = [&](int q) __attribute__((always_inline)) {
        const unsigned wb0 = lds_w0 + (unsigned)(((3 * q) & (RINGD - 1)) * WBYTES);
        auto f0 = [&](auto R) __attribute__((always_inline)) { frag1(xcur, wb0, ic<0>{}, ic<0>{}, R, fa[0], fb[0]); };
        f0(ic<0>{}); f0(ic<1>{}); f0(ic<2>{}); f0(ic<3>{}); f0(ic<4>{}); f0(ic<5>{});
        f0(ic<6>{}); f0(ic<7>{}); f0(ic<8>{}); f0(ic<9>{}); f0(ic<10>{}); f0(ic<11>{});
    };

    // ---- one tap of the MFMA stream = 24 units (one MFMA + a thin slice of other work, see conv_bf16x3_stream_kernel) ----
    auto tap = [&](int q, auto TT) __attribute__((always_inline)) {
        constexpr int t = decltype(TT)::value;
        constexpr int ky = t / 3, tx = t % 3, cur = t & 1;
        constexpr int kyn = t < 8 ? (t + 1) / 3 : 0, txn = t < 8 ? (t + 1) % 3 : 0;  // next tap
        const int sigma = 3 * q + ky;
        if (tx == 2) {
            // B'_sigma: every wave of the block has its fragments of tap (sigma, 2) in registers, and the weight stages the
            // leader requested during its last break have landed (it waits for its own LDS-DMA here; nothing else of it is
            // in flight): the next stage may be read, ring slot sigma % 4 may be overwritten once the FOLLOWER has passed
            // this barrier (it is the last reader of stage sigma), the other group's tile writes are done
            if (!follower) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            stamp(2);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            stamp(8);
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this tap's fragments (issued a tap ago)
        }
        const unsigned wbn = lds_w0 + (unsigned)(((kyn != ky ? sigma + 1 : sigma) & (RINGD - 1)) * WBYTES);
        __builtin_amdgcn_sched_barrier(0);
        constexpr int PI[6] = {NPC == 3 ? 2 : 1, 0, NPC == 3 ? 1 : 0, 1, 0, 0}, PJ[6] = {0, NPC == 3 ? 2 : 1, NPC == 3 ? 1 : 0, 0, 1, 0};
#pragma unroll
        for (int i = 0; i < UNITS; ++i) {
            const int qq = i / (MR * NR), m = (i / NR) % MR, n = i % NR;
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[cur][PI[qq]][m]),
                                                                __builtin_bit_cast(bf16x8, fb[cur][PJ[qq]][n]), acc[m][n], 0, 0, 0);
            if (i < NFR && t < 8) {  // next tap's fragments (the next chunk's first tap waits for the new x tile)
                auto fr = [&](auto R) __attribute__((always_inline)) { frag1(xcur, wbn, ic<kyn>{}, ic<txn>{}, R, fa[cur ^ 1], fb[cur ^ 1]); };
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
#ifndef DUO_NO_RAWLOAD  // (DUO_NO_*: ablation switches for scripts/build_variant.sh -- timing experiments, wrong results)
            if (t == 8) {  // the next chunk's twelve pixel / affine loads ride on the last units of the stream, where the second
                           // fragment buffer is idle; unconditional, so that the registers are plainly redefined here
                const int idx = i - (UNITS - 12);
                if (idx >= 0) load_piece(idx);
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // addresses of the group's chunk after next (its loads are issued from the last tap of the coming MFMA stream; issuing
    // them here, a whole phase earlier, keeps 48 more registers live across the stream: spills, measured 12 % slower)
    auto next_raw = [&]() __attribute__((always_inline)) { prep_load(); };

    // ---- everything between two multiplications of a group: three parts around the block's three barriers ----
    // qd: chunk just multiplied, qn = qd + 1 (== Q: none left).  sync = false: the follower's last tile -- no partner left
    // to pace, no barriers.
    auto between = [&](int qd, int qn, bool sync) __attribute__((always_inline)) {
        const int cd = qd % nchunks;
#ifdef DUO_NO_EPI
        const bool item_end = false, more = qn < Q;
#else
        const bool item_end = cd == nchunks - 1, more = qn < Q;
#endif
        stamp(item_end ? 7 : 3);
        // #1 comes first: the partner reaches its first barrier two taps into its stream, about when this group leaves its
        // own (the two streams share the matrix pipe during the overlap), so nothing may sit in front of it
        stamp(9);
        if (sync) __builtin_amdgcn_s_barrier();  // #1
        stamp(4);
        asm volatile("" ::: "memory");
        // rounding-bias cancellation (file header): the accumulator changes sign after every second chunk and chunk
        // pairs 1, 3, ... use pre-negated weights (Cin/32 is even)
        if (((cd + 1) & ((1 << p.sign_shift) - 1)) == 0) {
#pragma unroll
            for (int m = 0; m < MR; ++m)
#pragma unroll
                for (int n = 0; n < NR; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[m][n][r] = -acc[m][n][r];
        }
        stamp(19);
        // Two separate instruction sequences (tile finished / not): hipcc merges its vmcnt bookkeeping at every join, so a
        // conditional block that issues loads in front of an unconditional consumer of OLDER loads makes that consumer wait
        // for everything.  Order within the sequences: the raw pixels die in the transform before the statistics go live.
        if (!item_end) {
#ifndef DUO_NO_XF
            if (more) transform_half(0);
#endif
            stamp(20);
            if (!follower) dma_stage(3 * qd);  // (the follower passed B'_{3 qd} = this group's #1: that slot is free)
            stamp(21);
            stamp(10);
            if (sync) __builtin_amdgcn_s_barrier();  // #2
            stamp(5);
            asm volatile("" ::: "memory");
#ifndef DUO_NO_XF
            if (more) transform_half(1);
#endif
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's part of the new x tile is written
            stamp(22);
            if (!follower) dma_stage(3 * qd + 1);
            stamp(23);
            stamp(11);
            if (sync) __builtin_amdgcn_s_barrier();  // #3: the group's x tile is complete
            stamp(6);
            asm volatile("" ::: "memory");
            frag_first(qn);
            stamp(24);
            if (!follower) dma_stage(3 * qd + 2);
            stamp(25);
        } else {
#ifndef DUO_NO_XF
            if (more) transform_half(0);
#endif
            stamp(30);
            __builtin_amdgcn_sched_barrier(0);  // (fences: keep hipcc from hoisting later loads over earlier steps -- the
            epi_begin();                        //  register budget of this sequence is planned step by step)
            epi_bias();
            epi_load(ic<0>{}, ic<0>{}, rvA);
            epi_load(ic<0>{}, ic<1>{}, rvB);
            stamp(31);
            if (!follower) dma_stage(3 * qd);
            stamp(32);
            stamp(10);
            if (sync) __builtin_amdgcn_s_barrier();  // #2
            stamp(5);
            asm volatile("" ::: "memory");
#ifndef DUO_NO_XF
            if (more) transform_half(1);
#endif
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            stamp(33);
            __builtin_amdgcn_sched_barrier(0);
            epi_zero_stats();
            epi_finish(ic<0>{}, ic<0>{}, rvA);
            stamp(34);
            __builtin_amdgcn_sched_barrier(0);
            epi_load(ic<1>{}, ic<0>{}, rvA);
            __builtin_amdgcn_sched_barrier(0);
            epi_finish(ic<0>{}, ic<1>{}, rvB);
            stamp(35);
            __builtin_amdgcn_sched_barrier(0);
            epi_load(ic<1>{}, ic<1>{}, rvB);
            __builtin_amdgcn_sched_barrier(0);
            epi_stats(ic<0>{});
            stamp(36);
            if (!follower) dma_stage(3 * qd + 1);
            stamp(37);
            stamp(11);
            if (sync) __builtin_amdgcn_s_barrier();  // #3
            stamp(6);
            asm volatile("" ::: "memory");
            epi_zero_stats();
            epi_finish(ic<1>{}, ic<0>{}, rvA);
            stamp(38);
            __builtin_amdgcn_sched_barrier(0);
            epi_finish(ic<1>{}, ic<1>{}, rvB);
            stamp(39);
            __builtin_amdgcn_sched_barrier(0);
            epi_stats(ic<1>{});
            stamp(40);
            epi_end();
            stamp(41);
            __builtin_amdgcn_sched_barrier(0);
            frag_first(qn);  // (also after the last chunk: keeps the fragment registers dead across this function)
            stamp(42);
            if (!follower) dma_stage(3 * qd + 2);
            stamp(43);
        }
        next_raw();
        stamp(44);
    };

    // ---- prologue: ring stages 0..3 in flight (leader), chunk 0 transformed, chunk 1's pixels requested ----
    set_load_item(0);
    if (!follower) {
        set_dma_item(0);
#pragma unroll
        for (int s = 0; s < RINGD; ++s) dma_stage(s);
    }
    prep_load();
#pragma unroll
    for (int i = 0; i < 12; ++i) load_piece(i);
    transform_half(0);
    transform_half(1);
    next_raw();  // chunk 1 (nchunks >= 4)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    if (follower) {  // phase 0: the leader multiplies chunk 0, the follower only keeps the barrier count
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        frag_first(0);
    } else {
        frag_first(0);
    }
    for (int q = 0; q < Q; ++q) {
        stamp(1);
        // The group in its break outranks the multiplying one: the break (VALU + memory issue) is the longer of the two and
        // an MFMA stream needs one issue slot in 32 cycles.  (Measured: the reverse order costs 4-8 %.)
#ifdef DUO_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        tap(q, ic<0>{}); tap(q, ic<1>{}); tap(q, ic<2>{});
        tap(q, ic<3>{}); tap(q, ic<4>{}); tap(q, ic<5>{});
        tap(q, ic<6>{}); tap(q, ic<7>{}); tap(q, ic<8>{});
#ifdef DUO_PRIO
        __builtin_amdgcn_s_setprio(2);
#endif
        between(q, q + 1, !follower || q + 1 < Q);
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

// total_items = (output channel tiles) x (pairs of pixel tiles); one persistent block per CU walks its share
template <int PRO, int NPC>
static hipError_t launch_x3_duo(const ConvParams& p, long total_items, int n_cu, hipStream_t s) {
    auto kern = conv_bf16x3_duo_kernel<PRO, NPC>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, x3d::LDS_TOTAL);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const unsigned grid = (unsigned)(total_items < n_cu ? total_items : n_cu);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), x3d::LDS_TOTAL, s, p, (int)total_items);
    return hipGetLastError();
}

// Cin <= 128: the pair kernel by default.  The duo kernel is an opt-in experiment (R2DM_DUO_MIN=<items>: use it for launches
// with at least that many (co tile, tile pair) items; 1 = every shallow layer): after the fixes that came out of its
// timeline probe went into the pair kernel as well, it is on par with it ("what bounds the shallow
// layers"), so the simpler kernel stays the product path.
static long duo_min_items() {
    static const long v = [] {
        const char* e = getenv("R2DM_DUO_MIN");
        return e ? atol(e) : (1L << 62);
    }();
    return v;
}
static int cu_count() {
    static const int v = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    return v;
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
    if (p.pieces != 3 && p.pieces != 2) return hipErrorInvalidValue;
    const long n_tiles = (long)((p.W + 63) / 64) * ((p.H + 3) / 4) * p.B;
    const long items = (long)(p.Cout / x3::CO_T) * ((n_tiles + 1) / 2);
    const bool whole_tiles = p.H % 4 == 0 && p.W % 64 == 0;  // the duo kernel's epilogue has no pixel predication
    #ifdef DUO_PROF
    const bool duo = !deep && whole_tiles && items >= duo_min_items();
#else
    const bool duo = !deep && whole_tiles && items >= duo_min_items() && p.prof == nullptr;
#endif
#define X3_DISPATCH(PRO_)                                                                                           \
    if (duo) return p.pieces == 3 ? launch_x3_duo<PRO_, 3>(q, items, cu_count(), s) : launch_x3_duo<PRO_, 2>(q, items, cu_count(), s); \
    return !deep ? (p.pieces == 3 ? launch_x3_pair<PRO_, 3>(q, s) : launch_x3_pair<PRO_, 2>(q, s))                  \
           : p.co_tile == 32 ? (p.pieces == 3 ? launch_x3_stream<PRO_, 32, 3>(q, s) : launch_x3_stream<PRO_, 32, 2>(q, s)) \
                             : (p.pieces == 3 ? launch_x3_stream<PRO_, 64, 3>(q, s) : launch_x3_stream<PRO_, 64, 2>(q, s))
    switch (p.prologue) {
        case PRO_NONE: X3_DISPATCH(PRO_NONE);
        case PRO_AFFINE: X3_DISPATCH(PRO_AFFINE);
        case PRO_AFFINE_SILU: X3_DISPATCH(PRO_AFFINE_SILU);
    }
#undef X3_DISPATCH
    return hipErrorInvalidValue;
}

}  // namespace r2dm

// K2 / K3: ring-padded 3x3 and 1x1 convolution as an implicit GEMM on the fp32 matrix cores.
//
// Replaces reference ops.Conv2d + ops.Pad (/root/reference/models/ops.py:32-49,149-173) and, through
// the fused prologue/epilogue, the GroupNorm/AdaGN-apply + SiLU in front of every residual-block
// convolution and the skip-add + 1/sqrt(2) behind it (/root/reference/models/efficient_unet.py:95-110).
//
//   out[b,co,y,x] = s * ( res[b,co,y,x] + bias[co] +
//                         sum_{ci,dy,dx} W[co,ci,dy,dx] * f(in[b,ci,y+dy-1,(x+dx-1) mod W]) )
//   f(v) = v | v*a+d | silu(v*a+d) with (a,d) per (b,ci); rows outside [0,H) contribute 0.
//
// GEMM view: M = Cout, N = pixels, K = Cin*taps.  One MFMA v_mfma_f32_32x32x2_f32 multiplies a
// 32(co) x 2(k) weight fragment with a 2(k) x 32(px) activation fragment; k-pairs are two adjacent
// input channels at one tap.  fp32-in/fp32-accumulate MFMA is bit-equivalent to an fmaf chain, so
// the sampling-parity budget is untouched while the VALU stays free for the fused prologue.
//
// Block = 256 threads = 4 waves, tile = CO_T output channels x (TH x TW) pixels, K walked in chunks
// of CK input channels through a double-buffered LDS stage (one barrier per chunk):
//   * activations: (TH+2) x (TW+2) halo tile per channel (wrap in W, zero in H): the 16-byte-aligned interior as
//     float4 pieces + two halo scalars, global -> registers (issued before the chunk's MFMAs) -> prologue -> LDS
//     (after them); all addressing is precomputed once per block;
//   * weights: the chunk's packed [CK][taps][CO_T] slab is contiguous in HBM: 16-byte loads -> registers
//     -> ds_write_b128, same timing as the activations;
// Accuracy: with ACC2 the accumulators are flushed into a second register set every 64 input
// channels (576 products), so roundoff grows with sqrt(576) not sqrt(K) (K up to 4608).
#include "common.h"
#include "wave_ops.h"
#include <stdlib.h>

namespace r2dm {

template <int TAPS, int CO_T, int TH, int TW, int WCO, int WPX, int CK>
struct ConvCfg {
    static constexpr int HALO = TAPS == 9 ? 1 : 0;
    static constexpr int XR = TH + 2 * HALO;
    // LDS row: [3 pad][left halo][TW interior, 16-byte aligned][right halo][3 pad]  (no halo/pad for 1x1)
    static constexpr int XI = HALO ? 4 : 0;              // index of the first interior column
    static constexpr int XS = TW + (HALO ? 8 : 0);
    static constexpr int XPLANE = XR * XS;
    static constexpr int NX = CK * XPLANE;               // multiple of 4
    static constexpr int NQ = CK * XR * (TW / 4);        // interior 16-byte pieces per chunk
    static constexpr int NQT = (NQ + 255) / 256;
    static constexpr int NH = HALO ? CK * XR * 2 : 0;    // halo scalars per chunk
    static constexpr int NHT = (NH + 255) / 256;
    static constexpr int NW = CK * TAPS * CO_T;
    static constexpr int NW4 = NW / 4;
    static constexpr int NWT = (NW4 + 255) / 256;
    static constexpr int BUF = NX + NW;
    static constexpr int MR = CO_T / WCO / 32;
    static constexpr int SEGW = TW / 32;
    static constexpr int NSEG = TH * SEGW;
    static constexpr int NR = NSEG / WPX;
    static constexpr int NSTEP = (CK / 2) * TAPS;
    static constexpr int FLUSH = 64 / CK;  // chunks per accumulator flush (ACC2)
    static_assert(WCO * WPX == 4, "4 waves per block");
    static_assert(MR >= 1 && NR >= 1 && CK % 2 == 0 && NW % 4 == 0 && 64 % CK == 0 && TW % 4 == 0, "tile shape");
    static_assert(NHT <= 1 && NQT <= 7 && CK <= 16, "staging map packing");
    static constexpr size_t lds_bytes(int cin_pad, bool pro) {
        return (2 * (size_t)BUF + (pro ? 2 * (size_t)((cin_pad + 1) & ~1) : 0)) * sizeof(float);
    }
};

template <int TAPS, int CO_T, int TH, int TW, int WCO, int WPX, int CK, int PRO, bool ACC2, int OCC>
__global__ __launch_bounds__(256, OCC) void conv_mfma_kernel(const ConvParams p) {
    using C = ConvCfg<TAPS, CO_T, TH, TW, WCO, WPX, CK>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float2* affs = reinterpret_cast<float2*>(smem + 2 * C::BUF);

    const int tid = threadIdx.x;
    const int lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_co = wave % WCO, wave_px = wave / WCO;

    const int H = p.H, W = p.W;
    const int HW = H * W;
    const int nTw = (W + TW - 1) / TW, nTh = (H + TH - 1) / TH;
    const int nCoT = (p.Cout + CO_T - 1) / CO_T;
    int L = xcd_remap(blockIdx.x, gridDim.x);
    const int cot = L % nCoT;
    L /= nCoT;
    const int tw = L % nTw;
    L /= nTw;
    const int th = L % nTh;
    const int b = L / nTh;
    unsigned long long t0 = 0, t1 = 0, t2 = 0, ta = 0, tb = 0, s_load = 0, s_mfma = 0, s_bar = 0;  // perf probe
    if (p.prof) t0 = __builtin_amdgcn_s_memtime();

    // ---- per-thread staging map (computed once per block) ----
    // The interior of every staged row is TW contiguous, 16-byte aligned floats in HBM and in LDS: it moves as
    // float4 pieces (global_load_dwordx4 -> prologue -> ds_write_b128); only the two halo columns are scalars.
    // Columns wrap modulo W (azimuth is periodic; also covers tiles overhanging a narrow image), rows outside
    // [0,H) are zero padding.  Piece offsets / validity / LDS positions are per-thread constants, so per chunk the
    // staging costs one load + one write instruction per piece and no address arithmetic (fp32 MFMA and VALU
    // share one execution pipe on gfx950 -- scripts/mfma_valu_overlap.hip -- so every VALU op is paid in full).
    int qoff[C::NQT], qlds[C::NQT];  // global float offset within the chunk (0 if invalid) / LDS float offset
    int hoff = 0, hlds = 0;
    unsigned okbits = 0;             // bit j: piece j valid; bit 8: halo valid
    unsigned clbits = 0;             // 4 bits per piece (+ halo in bits 28..31): channel within the chunk
#pragma unroll
    for (int j = 0; j < C::NQT; ++j) {
        int q = tid + j * 256;
        q = q < C::NQ ? q : C::NQ - 1;  // surplus threads redo the last piece (same data, same place)
        const int cl = q / (C::XR * (TW / 4)), r = (q / (TW / 4)) % C::XR, c4 = q % (TW / 4);
        const int gr = th * TH + r - C::HALO;
        int gc = tw * TW + c4 * 4;
        while (gc >= W) gc -= W;
        const bool ok = gr >= 0 && gr < H;
        qoff[j] = ok ? cl * HW + gr * W + gc : 0;
        qlds[j] = cl * C::XPLANE + r * C::XS + C::XI + c4 * 4;
        okbits |= ok ? (1u << j) : 0u;
        clbits |= (unsigned)cl << (4 * j);
    }
    if (C::NHT) {
        int hq = tid < C::NH ? tid : C::NH - 1;
        const int cl = hq / (C::XR * 2), r = (hq / 2) % C::XR, side = hq & 1;
        const int gr = th * TH + r - C::HALO;
        int gc = side ? tw * TW + TW : tw * TW - 1;
        if (gc < 0) gc += W;
        while (gc >= W) gc -= W;
        const bool ok = gr >= 0 && gr < H;
        hoff = ok ? cl * HW + gr * W + gc : 0;
        hlds = cl * C::XPLANE + r * C::XS + (side ? C::XI + TW : C::XI - 1);
        okbits |= ok ? (1u << 8) : 0u;
        clbits |= (unsigned)cl << 28;
    }

    const float* wsrc = p.w + (size_t)cot * p.CinPad * TAPS * CO_T;
    const float* xb0 = p.x.p0 + b * p.x.bs0;
    const float* xb1 = p.x.p1 ? p.x.p1 + b * p.x.bs1 : p.x.p0;
    const int c0 = p.x.c0;
    f32x4 xq[C::NQT];
    float xh = 0.f;
    f32x4 wv[C::NWT];

    auto stage_load = [&](int ci0) {
        // weights: the chunk's [CK][taps][CO_T] slab is contiguous: 16-byte loads, ds_write_b128 after the MFMAs.
        // (LDS-DMA was tried: with it hipcc must drain vmcnt(0) before the first ds_read of the chunk because the
        // DMA target and the fragment reads live in one LDS array, which serialises HBM latency with the MFMAs.)
        const f32x4* w4 = reinterpret_cast<const f32x4*>(wsrc + (size_t)ci0 * TAPS * CO_T);
#pragma unroll
        for (int i = 0; i < C::NWT; ++i) {
            const int e = tid + i * 256;  // clamped, never predicated: a predicated load becomes an exec-masked
            wv[i] = w4[e < C::NW4 ? e : C::NW4 - 1];  // branch with a vmcnt wait in front of it
        }
        const bool whole = ci0 + CK <= p.Cin && (ci0 + CK <= c0 || ci0 >= c0);
        if (whole) {  // (wave-uniform base of the chunk's first channel) + (precomputed per-lane offset)
            const float* base = ci0 >= c0 ? xb1 + (long)(ci0 - c0) * HW : xb0 + (long)ci0 * HW;
#pragma unroll
            for (int j = 0; j < C::NQT; ++j) xq[j] = *reinterpret_cast<const f32x4*>(base + qoff[j]);
            if (C::NHT) xh = base[hoff];
        } else {  // last partial chunk (Cin % CK != 0) or a chunk straddling the concat seam (in_conv only)
#pragma unroll
            for (int j = 0; j < C::NQT; ++j) {
                const int cl = (clbits >> (4 * j)) & 15, ci = ci0 + cl;
                const bool ok = ((okbits >> j) & 1) && ci < p.Cin;
                const int cc = ci < p.Cin ? ci : p.Cin - 1;  // padded channels: load (and discard) from a plane that exists
                const float* pl = cc < c0 ? xb0 + (long)cc * HW : xb1 + (long)(cc - c0) * HW;
                xq[j] = *reinterpret_cast<const f32x4*>(pl + (ok ? qoff[j] - cl * HW : 0));
            }
            if (C::NHT) {
                const int cl = clbits >> 28, ci = ci0 + cl;
                const bool ok = ((okbits >> 8) & 1) && ci < p.Cin;
                const int cc = ci < p.Cin ? ci : p.Cin - 1;
                const float* pl = cc < c0 ? xb0 + (long)cc * HW : xb1 + (long)(cc - c0) * HW;
                xh = pl[ok ? hoff - cl * HW : 0];
            }
        }
    };
    auto stage_store = [&](float* buf, int ci0) {
        // Three independent batches -- table reads, transforms, LDS writes -- instead of one read/transform/write
        // chain per piece: the table lives in the same LDS array as the staging buffers, so the compiler must
        // keep every table read behind the previous write, and each chain would cost a full LDS latency.
        const bool whole = ci0 + CK <= p.Cin;
        float2 ad[C::NQT + 1];
        if (PRO != PRO_NONE) {
#pragma unroll
            for (int j = 0; j < C::NQT; ++j) ad[j] = affs[ci0 + (int)((clbits >> (4 * j)) & 15)];  // zero-filled to CinPad
            if (C::NHT) ad[C::NQT] = affs[ci0 + (int)(clbits >> 28)];
        }
#pragma unroll
        for (int j = 0; j < C::NQT; ++j) {
            bool ok = (okbits >> j) & 1;
            if (!whole) ok = ok && ci0 + (int)((clbits >> (4 * j)) & 15) < p.Cin;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = xq[j][e];
                if (PRO != PRO_NONE) {
                    v = v * ad[j].x + ad[j].y;
                    if (PRO == PRO_AFFINE_SILU) v = silu_f(v);
                }
                xq[j][e] = ok ? v : 0.f;  // zero padding applies to the *activated* tensor
            }
        }
        if (C::NHT) {
            bool ok = (okbits >> 8) & 1;
            if (!whole) ok = ok && ci0 + (int)(clbits >> 28) < p.Cin;
            float v = xh;
            if (PRO != PRO_NONE) {
                v = v * ad[C::NQT].x + ad[C::NQT].y;
                if (PRO == PRO_AFFINE_SILU) v = silu_f(v);
            }
            xh = ok ? v : 0.f;
        }
#pragma unroll
        for (int j = 0; j < C::NQT; ++j) *reinterpret_cast<f32x4*>(buf + qlds[j]) = xq[j];
        if (C::NHT) buf[hlds] = xh;
        f32x4* w4 = reinterpret_cast<f32x4*>(buf + C::NX);
#pragma unroll
        for (int i = 0; i < C::NWT; ++i) {
            const int e = tid + i * 256;
            if (C::NW4 % 256 == 0 || e < C::NW4) w4[e] = wv[i];
        }
    };

    // ---- fragment bases ----
    int xoff[C::NR];
#pragma unroll
    for (int n = 0; n < C::NR; ++n) {
        const int s = wave_px * C::NR + n;
        xoff[n] = hi * C::XPLANE + (s / C::SEGW) * C::XS + (C::XI - C::HALO) + (s % C::SEGW) * 32 + l31;
    }
    const int woff = C::NX + hi * TAPS * CO_T + wave_co * (CO_T / WCO) + l31;

    f32x16 acc[C::MR][C::NR];
    f32x16 acc2[ACC2 ? C::MR : 1][ACC2 ? C::NR : 1];
#pragma unroll
    for (int m = 0; m < C::MR; ++m)
#pragma unroll
        for (int n = 0; n < C::NR; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[m][n][r] = 0.f;
                if (ACC2) acc2[m][n][r] = 0.f;
            }

    const int nchunks = p.CinPad / CK;
    stage_load(0);
    if (PRO != PRO_NONE) {  // folded GroupNorm affine of this sample (all input channels) -> LDS; its HBM/L2 round
        // trip overlaps the first chunk's loads issued just above
        for (int c = tid; c < p.CinPad; c += 256) affs[c] = c < p.Cin ? p.aff[(size_t)b * p.Cin + c] : make_float2(0.f, 0.f);
        __syncthreads();
    }
    stage_store(smem, 0);
    __syncthreads();
    if (p.prof) t1 = __builtin_amdgcn_s_memtime();

    // Staging tasks of the NEXT chunk (registers -> prologue -> LDS), one per K-step in the second half of the MFMA
    // sequence: their dependent chains (table read -> fma -> exp -> rcp -> ... -> ds_write) then resolve while the
    // 64-cycle MFMAs of the same wave execute, instead of stalling in front of the barrier.
    constexpr int NTASK = C::NQT + C::NHT + C::NWT;
    constexpr int S0 = C::NSTEP / 2;
    constexpr int TPS = (NTASK + (C::NSTEP - S0) - 1) / (C::NSTEP - S0);  // tasks per step
    auto store_task = [&](int t, float* nbuf, int ci1) __attribute__((always_inline)) {
        if (t < C::NQT) {
            const int j = t;
            f32x4 q = xq[j];
            const bool ok = (okbits >> j) & 1;
            if (PRO != PRO_NONE) {
                const float2 ad = affs[ci1 + (int)((clbits >> (4 * j)) & 15)];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = q[e] * ad.x + ad.y;
                    if (PRO == PRO_AFFINE_SILU) v = silu_f(v);
                    q[e] = v;
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) q[e] = ok ? q[e] : 0.f;
            *reinterpret_cast<f32x4*>(nbuf + qlds[j]) = q;
        } else if (t < C::NQT + C::NHT) {
            const bool ok = (okbits >> 8) & 1;
            float v = xh;
            if (PRO != PRO_NONE) {
                const float2 ad = affs[ci1 + (int)(clbits >> 28)];
                v = v * ad.x + ad.y;
                if (PRO == PRO_AFFINE_SILU) v = silu_f(v);
            }
            nbuf[hlds] = ok ? v : 0.f;
        } else if (t < NTASK) {
            const int i = t - C::NQT - C::NHT, e = tid + i * 256;
            if (C::NW4 % 256 == 0 || e < C::NW4) reinterpret_cast<f32x4*>(nbuf + C::NX)[e] = wv[i];
        }
    };

    for (int k = 0; k < nchunks; ++k) {
        const float* buf = smem + (k & 1) * C::BUF;
        float* nbuf = smem + ((k + 1) & 1) * C::BUF;
        const bool more = k + 1 < nchunks;
        const int ci1 = (k + 1) * CK;
        if (p.prof) ta = __builtin_amdgcn_s_memtime();
        if (more) stage_load(ci1);
        if (p.prof) { tb = __builtin_amdgcn_s_memtime(); s_load += tb - ta; }
        const bool inter = more && ci1 + CK <= p.Cin;  // whole next chunk: branch-free per-step staging

        // MFMA operand fragments are read from LDS ONE K-step ahead of their use.  hipcc sinks every ds_read next to
        // its consumer (exposing the LDS latency in front of each group of MFMAs: ~8 % of the loop), so the reads are
        // issued by hand (inline asm, immediate offsets) and retired with a counted lgkmcnt: LDS operations complete
        // in order, so "at most MR+NR outstanding" means the previous step's fragments have landed whatever other
        // LDS traffic (staging writes, table reads) the compiler placed in between.
        float fa[2][C::MR], fb[2][C::NR];
        const unsigned lds_w = (unsigned)(size_t)(__attribute__((address_space(3))) const float*)buf + 4u * (unsigned)woff;
        unsigned lds_x[C::NR];
#pragma unroll
        for (int n = 0; n < C::NR; ++n)
            lds_x[n] = (unsigned)(size_t)(__attribute__((address_space(3))) const float*)buf + 4u * (unsigned)xoff[n];
        auto frag = [&](int s, float* a, float* bb) __attribute__((always_inline)) {
            const int cp = s / TAPS, t = s % TAPS;
            const int dy = TAPS == 9 ? t / 3 : 0, dx = TAPS == 9 ? t % 3 : 0;
#pragma unroll
            for (int m = 0; m < C::MR; ++m)
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(a[m]) : "v"(lds_w), "i"(4 * ((cp * 2 * TAPS + t) * CO_T + m * 32)));
#pragma unroll
            for (int n = 0; n < C::NR; ++n)
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(bb[n]) : "v"(lds_x[n]), "i"(4 * (cp * 2 * C::XPLANE + dy * C::XS + dx)));
        };
        frag(0, fa[0], fb[0]);
        // (one copy of the unrolled MFMA sequence only: a second copy in an if/else makes hipcc spill the accumulators)
#pragma unroll
        for (int s = 0; s < C::NSTEP; ++s) {
            if (s + 1 < C::NSTEP) {
                frag(s + 1, fa[(s + 1) & 1], fb[(s + 1) & 1]);
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(C::MR + C::NR) : "memory");
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the MFMAs below the wait (they do not touch memory)
#pragma unroll
            for (int m = 0; m < C::MR; ++m)
#pragma unroll
                for (int n = 0; n < C::NR; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s & 1][m], fb[s & 1][n], acc[m][n], 0, 0, 0);
            if (s >= S0 && inter) {
#pragma unroll
                for (int j = 0; j < TPS; ++j) store_task((s - S0) * TPS + j, nbuf, ci1);
            }
        }
        if (more && !inter) stage_store(nbuf, ci1);  // partial last chunk (Cin % CK != 0)
        if (ACC2 && ((k + 1) % C::FLUSH == 0)) {
#pragma unroll
            for (int m = 0; m < C::MR; ++m)
#pragma unroll
                for (int n = 0; n < C::NR; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        acc2[m][n][r] += acc[m][n][r];
                        acc[m][n][r] = 0.f;
                    }
        }
        if (p.prof) { ta = __builtin_amdgcn_s_memtime(); s_mfma += ta - tb; }
        __syncthreads();
        if (p.prof) { tb = __builtin_amdgcn_s_memtime(); s_bar += tb - ta; }
    }

    // ---- epilogue: bias, residual, scale; D layout col = lane&31 (pixel), row = (r&3)+8(r>>2)+4hi ----
    // Addresses are (wave-uniform base: SGPR pair) + (per-lane 32-bit offset) so that the 16..64 stores of a lane
    // share one offset VGPR.  All loads of a 32x32 tile (bias, residual) are issued before its first store: the
    // output may alias the residual as far as the compiler knows, and load-after-store would otherwise serialise
    // an L2 round trip per element.
    if (p.prof) t2 = __builtin_amdgcn_s_memtime();
    const float sc = p.scale ? *p.scale : 1.0f;
    const int co_u = cot * CO_T + wave_co * (CO_T / WCO);  // wave-uniform first output channel
    float* yu = p.y + b * p.y_bs + (long)co_u * HW;
    const float* ru = p.res ? p.res + b * p.res_bs + (long)co_u * HW : nullptr;
    const float* bu = p.bias + co_u;
    // EPI_N pixel segments are processed per pass: all their residual loads are in flight together.
    constexpr int EPI_N = (C::MR * C::NR * 16 <= 64) ? C::NR : (C::NR / 2 > 0 ? C::NR / 2 : 1);
    double st_s[C::MR][4], st_q[C::MR][4];  // fused GroupNorm statistics (fp64: E[x^2]-E[x]^2 must not see fp32
                                            // roundoff), per 8-channel block of this wave
    float amax = 0.f;                       // largest |output| of this lane (p.range)
#pragma unroll
    for (int m = 0; m < C::MR; ++m)
#pragma unroll
        for (int k8 = 0; k8 < 4; ++k8) st_s[m][k8] = st_q[m][k8] = 0.0;
    float bv[C::MR][16];
#pragma unroll
    for (int m = 0; m < C::MR; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cu = m * 32 + (r & 3) + 8 * (r >> 2);
            // (a tile may reach past the last output channel -- Cout < tile: the wave-uniform base stays on an existing channel too, not only the lane offset)
            bv[m][r] = (bu + (co_u + cu < p.Cout ? cu : 0))[co_u + cu + 4 * hi < p.Cout ? 4 * hi : 0];
        }
#pragma unroll
    for (int n0 = 0; n0 < C::NR; n0 += EPI_N) {
        int loff[EPI_N];
        bool px_ok[EPI_N];
        float rv[EPI_N][C::MR][16];
#pragma unroll
        for (int j = 0; j < EPI_N; ++j) {
            const int s = wave_px * C::NR + n0 + j;
            const int gr = th * TH + s / C::SEGW;
            const int gc = tw * TW + (s % C::SEGW) * 32 + l31;
            px_ok[j] = gr < H && gc < W;
            loff[j] = (px_ok[j] ? gr * W + gc : 0) + 4 * hi * HW;
            if (ru) {
#pragma unroll
                for (int m = 0; m < C::MR; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int cu = m * 32 + (r & 3) + 8 * (r >> 2);
                        rv[j][m][r] = (ru + (long)(co_u + cu < p.Cout ? cu : 0) * HW)[co_u + cu + 4 * hi < p.Cout ? loff[j] : 0];
                    }
            }
        }
#pragma unroll
        for (int j = 0; j < EPI_N; ++j)
#pragma unroll
            for (int m = 0; m < C::MR; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int cu = m * 32 + (r & 3) + 8 * (r >> 2);
                    float v = acc[m][n0 + j][r];
                    if (ACC2) v += acc2[m][n0 + j][r];
                    v += bv[m][r];
                    if (ru) v = rv[j][m][r] + v;
                    if (p.scale) v *= sc;
                    const bool live = px_ok[j] && co_u + cu + 4 * hi < p.Cout;
                    if (live) (yu + (long)cu * HW)[loff[j]] = v;
                    if (p.range && live) amax = fmaxf(amax, fabsf(v));
                    if (p.stat) {
                        const double vm = live ? (double)v : 0.0;
                        st_s[m][r >> 2] += vm;
                        st_q[m][r >> 2] = fma(vm, vm, st_q[m][r >> 2]);
                    }
                }
    }
    if (p.range) {  // running max |output| (ConvParams::range): the consumer runs on the fp16 matrix pipe
        amax = wave_max_f32(amax);
        const int bits = __float_as_int(amax);  // positive floats order like their bit patterns
        if (lane == 0 && bits > __atomic_load_n(p.range + 1, __ATOMIC_RELAXED)) atomicMax(p.range + 1, bits);
    }
    if (p.stat) {
        // lanes of a wave hold 4 of the 8 channels (by half-wave) x 32 pixels of every 8-channel block: reduce over
        // the wave in fp64, merge the blocks of a group, one slot per (pixel tile, pixel wave) -- fixed order.
        double bs[C::MR * 4], bq[C::MR * 4];
#pragma unroll
        for (int m = 0; m < C::MR; ++m) {
            double v[8];
#pragma unroll
            for (int k8 = 0; k8 < 4; ++k8) {
                v[2 * k8] = st_s[m][k8];
                v[2 * k8 + 1] = st_q[m][k8];
            }
            wave_sum8(v, lane);  // (wave_ops.h: butterfly on permlane swaps / DPP instead of 96 ds_bpermute)
#pragma unroll
            for (int k8 = 0; k8 < 4; ++k8) {
                bs[m * 4 + k8] = v[2 * k8];
                bq[m * 4 + k8] = v[2 * k8 + 1];
            }
        }
        if (lane == 0) {
            const int bpg = p.stat_cpg >> 3;                 // 8-channel blocks per group
            const int slot = (th * nTw + tw) * 4 + wave_px;  // 4 slots per pixel tile (unused ones hold zeros)
#pragma unroll
            for (int g0 = 0; g0 < C::MR * 4; ++g0) {
                if (g0 % bpg) continue;
                double a = 0.0, q = 0.0;
#pragma unroll
                for (int k8 = 0; k8 < C::MR * 4; ++k8)
                    if (k8 >= g0 && k8 < g0 + bpg) {
                        a += bs[k8];
                        q += bq[k8];
                    }
                const int g = p.stat_goff + (co_u + g0 * 8) / p.stat_cpg;
                if (co_u + g0 * 8 < p.Cout) {
                    double* o = p.stat + (((size_t)b * p.stat_G + g) * p.stat_slots + slot) * 2;
                    const int S2 = p.stat_slots & ~1;  // second half of the slots (see conv_epilogue.h): zeros
                    o[0] = a;
                    o[1] = q;
                    o[S2] = 0.0;
                    o[S2 + 1] = 0.0;
                    if (WPX == 2) {  // this variant fills only 2 of the tile's 4 slots
                        o[4] = 0.0;
                        o[5] = 0.0;
                        o[S2 + 4] = 0.0;
                        o[S2 + 5] = 0.0;
                    }

                }
            }
        }
    }
    if (p.prof && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long* o = p.prof + (size_t)blockIdx.x * 8;
        o[0] = t0;
        o[1] = t1;
        o[2] = t2;
        o[3] = __builtin_amdgcn_s_memtime();
        o[4] = s_load;
        o[5] = s_mfma;
        o[6] = s_bar;
        o[7] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);
    }
}

// ---- weight packing: OIHW (or (Cout,Cin) for Linear) -> [nCoT][CinPad][taps][CO_T], zero padded ----
__global__ void pack_conv_kernel(const float* __restrict__ w, float* __restrict__ dst, int Cout, int Cin,
                                 int taps, int co_tile, int cin_pad, long total, int src_cin, int src_off) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int col = i % co_tile;
        long r = i / co_tile;
        const int tap = r % taps;
        r /= taps;
        const int ci = r % cin_pad;
        const int t = r / cin_pad;
        const int co = t * co_tile + col;
        dst[i] = (co < Cout && ci < Cin) ? w[((long)co * src_cin + src_off + ci) * taps + tap] : 0.f;  // input-channel slice
    }
}

hipError_t launch_pack_conv(const float* w, float* dst, int Cout, int Cin, int taps, int co_tile, int cin_pad,
                            hipStream_t s, int algo, int src_cin, int src_off) {
    if (src_cin <= 0) src_cin = Cin;
    if (algo == ALGO_DIRECT) {  // OIHW as is (an input-channel slice: row by row)
        if (src_cin == Cin) return hipMemcpyAsync(dst, w, (size_t)Cout * Cin * taps * sizeof(float), hipMemcpyDeviceToDevice, s);
        return hipMemcpy2DAsync(dst, (size_t)Cin * taps * sizeof(float), w + (size_t)src_off * taps, (size_t)src_cin * taps * sizeof(float),
                                (size_t)Cin * taps * sizeof(float), Cout, hipMemcpyDeviceToDevice, s);
    }
    if (algo == ALGO_BF16X3) return src_cin == Cin ? launch_pack_conv_bf16x3(w, dst, Cout, Cin, co_tile, s) : hipErrorInvalidValue;
    if (algo == ALGO_F16X2) return hipErrorInvalidValue;  // (launch_pack_conv_f16x2: needs the range flag)
    const int nT = (Cout + co_tile - 1) / co_tile;
    const long total = (long)nT * cin_pad * taps * co_tile;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    pack_conv_kernel<<<blocks, 256, 0, s>>>(w, dst, Cout, Cin, taps, co_tile, cin_pad, total, src_cin, src_off);
    return hipGetLastError();
}

// ---- dispatch ----
// tile variants:            CO_T  waves(co x px)  CK   acc regs/lane   blocks/CU target
//   3x3  "A128"             128   2 x 2           4    128             2
//   3x3  "B64"               64   1 x 4           4    64              3      (Cin <= 128)
//   3x3  "B64 deep"          64   1 x 4           8    64 + 64 (ACC2)  2      (Cin > 128: long prefetch distance)
//   3x3  "D32" (out_conv)    32   1 x 4           8    32              3
//   1x1                   32/64/128               16
constexpr int kCK3_128 = 4, kCK3_64 = 4, kCK3_64_DEEP = 8, kCK3_32 = 8, kCK1 = 16;

int conv_pick_algo(int Cin, int Cout, int taps) {
    static const bool force_f32 = [] {
        const char* e = getenv("R2DM_CONV_ALGO");
        return e && e[0] == 'f';
    }();
    if (force_f32) return ALGO_F32;
    // the direct kernel keeps all Cout*Cin*9 weights in LDS (32 KiB budget, launch_conv_direct): larger shapes use the MFMA path
    if (conv_direct_supported(Cout, taps) && (long)Cout * Cin * 36 <= 32768) return ALGO_DIRECT;
    return conv_bf16x3_supported(Cin, Cout, taps) ? ALGO_BF16X3 : ALGO_F32;
}

long conv_packed_floats(int algo, int Cin, int Cout, int taps, int co_tile, int cin_pad) {
    if (algo == ALGO_BF16X3) return conv_bf16x3_packed_floats(Cin, Cout);
    if (algo == ALGO_F16X2) return conv_f16x2_packed_floats(Cin, Cout);
    if (algo == ALGO_P1F16) return proj_f16x2_packed_floats(Cin, Cout);
    if (algo == ALGO_DIRECT) return (long)Cout * Cin * taps;  // OIHW as is
    return (long)((Cout + co_tile - 1) / co_tile) * cin_pad * taps * co_tile;
}

int conv_pick_co_tile(int Cout, int taps, long px_batch) {
    (void)taps;
    if (Cout <= 32) return 32;
    if (Cout % 128 == 0) {
        const long nblk = (long)(Cout / 128) * ((px_batch + 255) / 256);
        if (nblk >= 512) return 128;
    }
    return 64;
}

int conv_stat_slots(int H, int W) { return ((H + 3) / 4) * ((W + 63) / 64) * 4 * 2; }  // two halves: conv_epilogue.h

int conv_cin_pad(int Cin, int taps, int co_tile) {
    const int ck = taps == 9 ? (co_tile == 128 ? kCK3_128 : co_tile == 64 ? kCK3_64_DEEP : kCK3_32) : kCK1;
    return (Cin + ck - 1) / ck * ck;
}

template <int TAPS, int CO_T, int WCO, int WPX, int CK, int PRO, bool ACC2, int OCC>
static hipError_t launch_variant(const ConvParams& p, hipStream_t s) {
    using C = ConvCfg<TAPS, CO_T, 4, 64, WCO, WPX, CK>;
    auto kern = conv_mfma_kernel<TAPS, CO_T, 4, 64, WCO, WPX, CK, PRO, ACC2, OCC>;
    const size_t lds = C::lds_bytes(p.CinPad, PRO != PRO_NONE);
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_lds = lds;
    }
    const int nTw = (p.W + 63) / 64, nTh = (p.H + 3) / 4, nCoT = (p.Cout + CO_T - 1) / CO_T;
    const long nblk = (long)nCoT * nTw * nTh * p.B;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), lds, s, p);
    return hipGetLastError();
}

template <int TAPS, int CO_T, int WCO, int WPX, int CK, bool ACC2, int OCC>
static hipError_t launch_pro(const ConvParams& p, hipStream_t s) {
    switch (p.prologue) {
        case PRO_NONE: return launch_variant<TAPS, CO_T, WCO, WPX, CK, PRO_NONE, ACC2, OCC>(p, s);
        case PRO_AFFINE: return launch_variant<TAPS, CO_T, WCO, WPX, CK, PRO_AFFINE, ACC2, OCC>(p, s);
        case PRO_AFFINE_SILU: return launch_variant<TAPS, CO_T, WCO, WPX, CK, PRO_AFFINE_SILU, ACC2, OCC>(p, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_conv(const ConvParams& p, hipStream_t s) {
    if (p.algo == ALGO_BF16X3) return launch_conv_bf16x3(p, s);
    if (p.algo == ALGO_F16X2) return launch_conv_f16x2(p, s);
    if (p.algo == ALGO_DIRECT) return launch_conv_direct(p, s);
    if (p.algo == ALGO_P1F16) return launch_proj_f16x2(p, s);
    if (p.prologue != PRO_NONE && p.aff == nullptr) return hipErrorInvalidValue;
    if (p.CinPad != conv_cin_pad(p.Cin, p.taps, p.co_tile)) return hipErrorInvalidValue;
    if (p.H * (long)p.W * 16 >= (1L << 31)) return hipErrorInvalidValue;  // 32-bit element offsets within a chunk
    if (p.W % 4) return hipErrorInvalidValue;                              // 16-byte row pieces
    if (p.taps == 9) {
        if (p.co_tile == 128) return launch_pro<9, 128, 2, 2, kCK3_128, false, 2>(p, s);
        if (p.co_tile == 64) {
            // long reductions (K = 9*Cin > 1152) get the two-level accumulator
#ifdef R2DM_ACC2_ALL
            if (true) return launch_pro<9, 64, 1, 4, kCK3_64_DEEP, true, 2>(p, s);
#endif
            // (pieces == 5: per-kernel test hook -- ONE accumulator at any depth, i.e. the plain fmaf chain the split kernels are measured against)
            return p.Cin > 128 && p.pieces != 5 ? launch_pro<9, 64, 1, 4, kCK3_64_DEEP, true, 2>(p, s)
                                                : launch_pro<9, 64, 1, 4, kCK3_64, false, 3>(p, s);
        }
        if (p.co_tile == 32) return launch_pro<9, 32, 1, 4, kCK3_32, false, 3>(p, s);
    } else if (p.taps == 1) {
        if (p.co_tile == 128) return launch_pro<1, 128, 2, 2, kCK1, false, 2>(p, s);
        if (p.co_tile == 64) return launch_pro<1, 64, 1, 4, kCK1, false, 3>(p, s);
        if (p.co_tile == 32) return launch_pro<1, 32, 1, 4, kCK1, false, 3>(p, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace r2dm

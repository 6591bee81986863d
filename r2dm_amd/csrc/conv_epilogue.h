// Epilogue shared by the convolution kernels whose accumulators use the 32x32 MFMA C/D layout
// (col = lane&31 = pixel, row = (r&3) + 8(r>>2) + 4(lane>>5) = output channel; identical for the fp32-input and
// the bf16-input 32x32 instructions on gfx950): bias, residual add, scale, store, and the fused fp64 GroupNorm
// statistics of the output tensor (reference efficient_unet.py:95-110; ops.py:149-173).
#pragma once
#include "common.h"
#include "wave_ops.h"

namespace r2dm {

// Fused GroupNorm statistics, last step: the lanes of a wave hold 4 of the 8 channels (by half-wave) x 32 pixels of every
// 8-channel block; reduce over the wave in fp64, merge the blocks of a group, one slot per (pixel tile, pixel wave) --
// fixed summation order, every slot written exactly once per launch.
template <int WPX, int MR>
__device__ __forceinline__ void epi_stat_write(const ConvParams& p, double (&st_s)[MR][4], double (&st_q)[MR][4], int b,
                                               int th, int tw, int nTw, int co_u, int wave_px, int lane) {
    using gdouble = double __attribute__((address_space(1)))*;  // (global, not FLAT: flat stores also count on lgkmcnt)
    double bs[MR * 4], bq[MR * 4];
#pragma unroll
    for (int m = 0; m < MR; ++m) {
        double v[8];
#pragma unroll
        for (int k8 = 0; k8 < 4; ++k8) {
            v[2 * k8] = st_s[m][k8];
            v[2 * k8 + 1] = st_q[m][k8];
        }
        wave_sum8(v, lane);
#pragma unroll
        for (int k8 = 0; k8 < 4; ++k8) {
            bs[m * 4 + k8] = v[2 * k8];
            bq[m * 4 + k8] = v[2 * k8 + 1];
        }
    }
    if (lane == 0) {
        // Slots per (sample, group): two halves of S = stat_slots / 2, each with one slot per (pixel tile, pixel
        // wave).  A wave whose 32*MR channels contain whole groups writes its sums to half 0 and zeros to half 1; a
        // wave that holds only half of a 64-channel group (32-channel tiles) writes to the half given by its
        // position in the group.
        constexpr int R8 = MR * 4;                       // 8-channel blocks per wave
        const int S = p.stat_slots >> 1;
        const int bpg = p.stat_cpg >> 3;                 // 8-channel blocks per group
        const int slot = (th * nTw + tw) * 4 + wave_px;  // 4 slots per pixel tile (unused ones hold zeros)
        if (bpg <= R8) {
#pragma unroll
            for (int g0 = 0; g0 < R8; ++g0) {
                if (g0 % bpg) continue;
                double a = 0.0, q = 0.0;
#pragma unroll
                for (int k8 = 0; k8 < R8; ++k8)
                    if (k8 >= g0 && k8 < g0 + bpg) {
                        a += bs[k8];
                        q += bq[k8];
                    }
                const int g = p.stat_goff + (co_u + g0 * 8) / p.stat_cpg;
                if (co_u + g0 * 8 < p.Cout) {
                    gdouble o = (gdouble)(p.stat + (((size_t)b * p.stat_G + g) * p.stat_slots + slot) * 2);
                    o[0] = a;
                    o[1] = q;
                    o[2 * S] = 0.0;
                    o[2 * S + 1] = 0.0;
                    if (WPX == 2) {  // this variant fills only 2 of the tile's 4 slots
                        o[4] = 0.0;
                        o[5] = 0.0;
                        o[2 * S + 4] = 0.0;
                        o[2 * S + 5] = 0.0;
                    }

                }
            }
        } else {  // the wave's channels are one half of a group (bpg == 2 * R8)
            double a = 0.0, q = 0.0;
#pragma unroll
            for (int k8 = 0; k8 < R8; ++k8) {
                a += bs[k8];
                q += bq[k8];
            }
            const int g = p.stat_goff + co_u / p.stat_cpg;
            const int half = (co_u % p.stat_cpg) / (R8 * 8);
            gdouble o = (gdouble)(p.stat + (((size_t)b * p.stat_G + g) * p.stat_slots + half * S + slot) * 2);
            o[0] = a;
            o[1] = q;
        }
    }
}

// The same write-out for ONE 32-channel half (MR index m) of a wave that owns 64 output channels, as a butterfly
// reduce-scatter: 8 values (sum, sum of squares of the four 8-channel blocks) over 64 lanes cost 10 exchange+add steps
// instead of 48, and the merge of the blocks of a group continues the butterfly.  Value index v = 2*block + (0: sum,
// 1: squares); after the scatter lane L holds the wave total of v = (L >> 3) & 7.  Groups of 8 / 16 / 32 channels lie
// inside the half (slot half 0, zeros to half 1); a 64-channel group takes the two halves of its wave in slot halves
// 0 and 1, exactly as two 32-channel tiles would.  Fixed summation order, every slot written exactly once per launch.
__device__ __forceinline__ void epi_stat_write_bfly8(const ConvParams& p, double (&st_s)[4], double (&st_q)[4], int b, int th,
                                                     int tw, int nTw, int co_half, int wave_px, int lane) {
    using gdouble = double __attribute__((address_space(1)))*;
    double v[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[2 * j] = st_s[j];
        v[2 * j + 1] = st_q[j];
    }
    double t = wave_sum8_scatter(v, lane);  // lane L: total of value (L >> 3) & 7 (wave_ops.h)
    // lane L: total of block k8 = (L >> 4) & 3, kind = (L >> 3) & 1; partners within a group differ in lane bits 4, 5
    const int bpg = p.stat_cpg >> 3;  // 1, 2, 4 or 8 blocks per group
    if (bpg >= 2) wave_swap_add(t, t, true);    // + the lane 16 further on / back: blocks k8, k8 ^ 1
    if (bpg >= 4) wave_swap_add(t, t, false);   // + the other 32 lanes: blocks k8, k8 ^ 2
    const int k8 = (lane >> 4) & 3, kind = (lane >> 3) & 1;
    const int bin = bpg < 4 ? bpg : 4;  // blocks of a group inside this half
    if ((lane & 7) == 0 && (k8 & (bin - 1)) == 0) {
        const int S = p.stat_slots >> 1;
        const int slot = (th * nTw + tw) * 4 + wave_px;
        // (bpg is 1, 2, 4 or 8 here -- a power of two: shifts, not the ~40 instructions of two integer divisions per half)
        const int sh = 3 + (bpg >= 2) + (bpg >= 4) + (bpg >= 8);  // log2(stat_cpg)
        const int g = p.stat_goff + ((co_half + k8 * 8) >> sh);
        const int half = bpg == 8 ? (co_half & (p.stat_cpg - 1)) >> 5 : 0;
        gdouble o = (gdouble)(p.stat + (((size_t)b * p.stat_G + g) * p.stat_slots + slot) * 2);
        o[2 * S * half + kind] = t;
        if (bpg < 8) o[2 * S + kind] = 0.0;
    }
}

// conv_f16x2.hip's tile ends (round 6): the wave's share of a statistics slot -- 2 quarters x 64 lanes x 4 pixels of an 8-channel block, 512 values -- is summed in
// fp32 (pairwise: 2 + 1 + 6 butterfly levels, ~1e-7 relative, unbiased and independent from slot to slot: a group's >= 128 slots average it down to ~1e-8), fp64 from
// the slot on (gn_finalize / the folded GroupNorm).  One instruction per exchange instead of two plus a half-rate add and sixteen conversions: with the range maximum
// taken from the sums of squares, -0.9 % on the step (profiles/r06_tile_end_diet.txt); -DF2_STATS_F64 restores the fp64 butterfly above.
__device__ __forceinline__ void epi_stat_write_bfly8(const ConvParams& p, float (&st_s)[4], float (&st_q)[4], int b, int th,
                                                     int tw, int nTw, int co_half, int wave_px, int lane) {
    using gdouble = double __attribute__((address_space(1)))*;
    float v[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[2 * j] = st_s[j];
        v[2 * j + 1] = st_q[j];
    }
    float t = wave_sum8_scatter(v, lane);
    const int bpg = p.stat_cpg >> 3;
    if (bpg >= 2) wave_swap_add(t, t, true);
    if (bpg >= 4) wave_swap_add(t, t, false);
    const int k8 = (lane >> 4) & 3, kind = (lane >> 3) & 1;
    const int bin = bpg < 4 ? bpg : 4;
    if ((lane & 7) == 0 && (k8 & (bin - 1)) == 0) {
        const int S = p.stat_slots >> 1;
        const int slot = (th * nTw + tw) * 4 + wave_px;
        const int sh = 3 + (bpg >= 2) + (bpg >= 4) + (bpg >= 8);
        const int g = p.stat_goff + ((co_half + k8 * 8) >> sh);
        const int half = bpg == 8 ? (co_half & (p.stat_cpg - 1)) >> 5 : 0;
        gdouble o = (gdouble)(p.stat + (((size_t)b * p.stat_G + g) * p.stat_slots + slot) * 2);
        o[2 * S * half + kind] = (double)t;
        if (bpg < 8) o[2 * S + kind] = 0.0;
    }
}

// ---- wide epilogue (whole tiles: H % TH == 0, W % TW == 0, Cout % (32 MR) == 0) --------------------------------------
// The MFMA layout gives a lane ONE pixel of 16 channels, i.e. 4-byte global accesses, 256 B per instruction -- and a CU
// retires those at a few bytes per cycle (in-kernel timeline of round 2: the residual loads + stores of one 64 x 256 tile
// took as long as two thirds of its MFMA time).  Every 8-channel block is therefore turned through a private 1 KiB LDS
// patch (4 ds_write_b32 + 1 ds_read_b128; no barrier: the LDS operations of one wave execute in order) into
// "4 consecutive pixels of one channel per lane": residual loads and output stores are 16 bytes per lane, 1 KiB per
// instruction, a quarter as many.  Statistics: the four pixels are summed in fp32 (3 + 4 roundings of ~6e-8, unbiased and
// independent from lane to lane: they average out over the >= 10^4 lanes x tiles of a group), fp64 beyond that, butterfly
// reduce-scatter per 32-channel half (epi_stat_write_bfly8).
// Explicit global address space everywhere: a pointer that reaches a load through a phi is otherwise accessed with FLAT
// instructions, which count on lgkmcnt as well and turn every LDS wait into a wait for HBM.
// conv_f16x2.hip has its own arrangement of the same steps (one quarter at the tile's end, three deferred into the next tile).
// Y16 (round 6): the output -- and the residual -- are stored as fp16 (the one-plane mode's activation storage); statistics and range of the values as stored.
template <int TH, int TW, int MR, int NR, bool ACC2, bool Y16 = false>
__device__ __forceinline__ void conv_epilogue_wide(const ConvParams& p, f32x16 (&acc)[MR][NR],
                                                   f32x16 (&acc2)[ACC2 ? MR : 1][ACC2 ? NR : 1], int b, int th, int tw,
                                                   int nTw, int co_u, int wave_px, int lane, float* patch,
                                                   float acc2_scale = 1.0f,    // result = (acc + acc2_scale * acc2) * out_scale + bias ...
                                                   float out_scale = 1.0f) {   // (ConvParams::wscale: a power of two)
    using gcf = const float __attribute__((address_space(1)))*;
    using gcf4 = const f32x4 __attribute__((address_space(1)))*;
    using gf4 = f32x4 __attribute__((address_space(1)))*;
    constexpr int SEGW = TW / 32;
    const int l31 = lane & 31, hi = lane >> 5;
    const int H = p.H, W = p.W, HW = H * W;
    const int tq_c = lane >> 3, tq_p = (lane & 7) * 4;  // after the turn: channel within the block, first of 4 pixels
    const float sc = p.scale ? *p.scale : 1.0f;
    const gf4 yu = (gf4)(p.y + b * p.y_bs + (long)co_u * HW);
    const gcf4 ru = (gcf4)(p.res + b * p.res_bs + (long)co_u * HW);  // (only dereferenced if p.res)
    int loff[NR];  // in units of 4 floats
#pragma unroll
    for (int n = 0; n < NR; ++n) {
        const int s = wave_px * NR + n;
        loff[n] = (tq_c * HW + (th * TH + s / SEGW) * W + tw * TW + (s % SEGW) * 32 + tq_p) >> 2;
    }
    float bias[MR][4];
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int k8 = 0; k8 < 4; ++k8) bias[m][k8] = ((gcf)p.bias)[co_u + m * 32 + k8 * 8 + tq_c];
    constexpr int NQ = MR * NR;
    // residual values: the quarter being finished and the next one in flight (two buffers)
    f32x4 rv[2][4];
    auto load_q = [&](int m, int n, f32x4 (&r)[4]) __attribute__((always_inline)) {
        if (p.res) {
#pragma unroll
            for (int k8 = 0; k8 < 4; ++k8) {
                if constexpr (Y16) r[k8] = load4<true>(p.res, b * p.res_bs + (long)co_u * HW + 4 * ((long)(m * 32 + k8 * 8) * (HW >> 2) + loff[n]));
                else r[k8] = (ru + (long)(m * 32 + k8 * 8) * (HW >> 2))[loff[n]];
            }
        }
    };
    float amax = 0.f;  // running max |output| (p.range)
    load_q(0, 0, rv[0]);
#pragma unroll
    for (int m = 0; m < MR; ++m) {
        double st_s[4], st_q[4];
#pragma unroll
        for (int k8 = 0; k8 < 4; ++k8) st_s[k8] = st_q[k8] = 0.0;
#pragma unroll
        for (int n = 0; n < NR; ++n) {
            const int q = m * NR + n;
            if (q + 1 < NQ) load_q((q + 1) / NR, (q + 1) % NR, rv[(q + 1) & 1]);
            f32x4 t[4];
#pragma unroll
            for (int k8 = 0; k8 < 4; ++k8) {  // the four blocks through the patch back to back (in-order LDS: no waits between)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = acc[m][n][4 * k8 + j];
                    if (ACC2) v = fmaf(acc2[ACC2 ? m : 0][ACC2 ? n : 0][4 * k8 + j], acc2_scale, v);
                    patch[(j + 4 * hi) * 32 + l31] = v;
                }
                t[k8] = *reinterpret_cast<const f32x4*>(patch + tq_c * 32 + tq_p);
            }
#pragma unroll
            for (int k8 = 0; k8 < 4; ++k8) {
                f32x4 v = t[k8] * out_scale + bias[m][k8];
                if (p.res) v = rv[q & 1][k8] + v;
                if (p.scale) v *= sc;
                if constexpr (Y16) v = store4<true>(p.y, b * p.y_bs + (long)co_u * HW + 4 * ((long)(m * 32 + k8 * 8) * (HW >> 2) + loff[n]), v);
                else (yu + (long)(m * 32 + k8 * 8) * (HW >> 2))[loff[n]] = v;
                if (p.range) amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
                if (p.stat) {
                    const float s4 = (v[0] + v[1]) + (v[2] + v[3]);
                    const float q4 = fmaf(v[3], v[3], fmaf(v[2], v[2], fmaf(v[1], v[1], v[0] * v[0])));
                    st_s[k8] += (double)s4;
                    st_q[k8] += (double)q4;
                }
            }
        }
        if (p.stat) epi_stat_write_bfly8(p, st_s, st_q, b, th, tw, nTw, co_u + m * 32, wave_px, lane);
    }
    if (p.range) {
        amax = wave_max_f32(amax);
        const int bits = __float_as_int(amax);  // positive floats order like their bit patterns
        if (lane == 0 && bits > __atomic_load_n(p.range + 1, __ATOMIC_RELAXED)) atomicMax(p.range + 1, bits);
    }
}

// The wave owns output channels [co_u, co_u + 32*MR) and NR pixel segments (32 consecutive columns of one row each);
// segment index s = wave_px*NR + n -> row s / (TW/32), column block s % (TW/32) of the TH x TW pixel tile.
template <int WPX, int TH, int TW, int MR, int NR, bool ACC2>
__device__ __forceinline__ void conv_epilogue(const ConvParams& p, f32x16 (&acc)[MR][NR],
                                              f32x16 (&acc2)[ACC2 ? MR : 1][ACC2 ? NR : 1], int b, int th, int tw,
                                              int nTw, int co_u, int wave_px, int lane) {
    constexpr int SEGW = TW / 32;
    const int l31 = lane & 31, hi = lane >> 5;
    const int H = p.H, W = p.W, HW = H * W;
    // Addresses are (wave-uniform base: SGPR pair) + (per-lane 32-bit offset) so that the 16..64 stores of a lane
    // share one offset VGPR.  All loads of a pass (bias, residual) are issued before its first store: the output
    // may alias the residual as far as the compiler knows, and load-after-store would otherwise serialise an L2
    // round trip per element.
    const float sc = p.scale ? *p.scale : 1.0f;
    float* yu = p.y + b * p.y_bs + (long)co_u * HW;
    const float* ru = p.res ? p.res + b * p.res_bs + (long)co_u * HW : nullptr;
    const float* bu = p.bias + co_u;
    constexpr int EPI_N = (MR * NR * 16 <= 64) ? NR : (NR / 2 > 0 ? NR / 2 : 1);
    float amax = 0.f;
    double st_s[MR][4], st_q[MR][4];  // fp64: E[x^2]-E[x]^2 must not see fp32 roundoff; per 8-channel block
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int k8 = 0; k8 < 4; ++k8) st_s[m][k8] = st_q[m][k8] = 0.0;
    float bv[MR][16];
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cu = m * 32 + (r & 3) + 8 * (r >> 2);
            // (a tile may reach past the last output channel -- Cout < tile: the wave-uniform base stays on an existing channel too, not only the lane offset)
            bv[m][r] = (bu + (co_u + cu < p.Cout ? cu : 0))[co_u + cu + 4 * hi < p.Cout ? 4 * hi : 0];
        }
#pragma unroll
    for (int n0 = 0; n0 < NR; n0 += EPI_N) {
        int loff[EPI_N];
        bool px_ok[EPI_N];
        float rv[EPI_N][MR][16];
#pragma unroll
        for (int j = 0; j < EPI_N; ++j) {
            const int s = wave_px * NR + n0 + j;
            const int gr = th * TH + s / SEGW;
            const int gc = tw * TW + (s % SEGW) * 32 + l31;
            px_ok[j] = gr < H && gc < W;
            loff[j] = (px_ok[j] ? gr * W + gc : 0) + 4 * hi * HW;
            if (ru) {
#pragma unroll
                for (int m = 0; m < MR; ++m)
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
            for (int m = 0; m < MR; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int cu = m * 32 + (r & 3) + 8 * (r >> 2);
                    float v = acc[m][n0 + j][r];
                    if (ACC2) v += acc2[ACC2 ? m : 0][ACC2 ? n0 + j : 0][r];
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
    if (p.stat) epi_stat_write<WPX, MR>(p, st_s, st_q, b, th, tw, nTw, co_u, wave_px, lane);
    if (p.range) {  // running max |output| (ConvParams::range), as in conv_epilogue_wide
        amax = wave_max_f32(amax);
        const int bits = __float_as_int(amax);
        if (lane == 0 && bits > __atomic_load_n(p.range + 1, __ATOMIC_RELAXED)) atomicMax(p.range + 1, bits);
    }
}

}  // namespace r2dm

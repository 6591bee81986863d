// Epilogue shared by the convolution kernels whose accumulators use the 32x32 MFMA C/D layout
// (col = lane&31 = pixel, row = (r&3) + 8(r>>2) + 4(lane>>5) = output channel; identical for the fp32-input and
// the bf16-input 32x32 instructions on gfx950): bias, residual add, scale, store, and the fused fp64 GroupNorm
// statistics of the output tensor (reference efficient_unet.py:95-110; ops.py:149-173).
#pragma once
#include "common.h"

namespace r2dm {

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
            bv[m][r] = (bu + cu)[co_u + cu + 4 * hi < p.Cout ? 4 * hi : 0];
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
                        rv[j][m][r] = (ru + (long)cu * HW)[co_u + cu + 4 * hi < p.Cout ? loff[j] : 0];
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
                    if (p.stat) {
                        const double vm = live ? (double)v : 0.0;
                        st_s[m][r >> 2] += vm;
                        st_q[m][r >> 2] = fma(vm, vm, st_q[m][r >> 2]);
                    }
                }
    }
    if (p.stat) {
        // lanes of a wave hold 4 of the 8 channels (by half-wave) x 32 pixels of every 8-channel block: reduce over
        // the wave in fp64, merge the blocks of a group, one slot per (pixel tile, pixel wave) -- fixed order.
        double bs[MR * 4], bq[MR * 4];
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int k8 = 0; k8 < 4; ++k8) {
                double a = st_s[m][k8], q = st_q[m][k8];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    a += __shfl_xor(a, o, 64);
                    q += __shfl_xor(q, o, 64);
                }
                bs[m * 4 + k8] = a;
                bq[m * 4 + k8] = q;
            }
        if (lane == 0) {
            // Slots per (sample, group): two halves of S = stat_slots / 2, each with one slot per (pixel tile, pixel
            // wave).  A wave whose 32*MR channels contain whole groups writes its sums to half 0 and zeros to half 1; a
            // wave that holds only half of a 64-channel group (32-channel tiles) writes to the half given by its
            // position in the group.  Every slot is written exactly once per launch: fixed summation order.
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
                        double* o = p.stat + (((size_t)b * p.stat_G + g) * p.stat_slots + slot) * 2;
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
                double* o = p.stat + (((size_t)b * p.stat_G + g) * p.stat_slots + half * S + slot) * 2;
                o[0] = a;
                o[1] = q;
            }
        }
    }
}

}  // namespace r2dm

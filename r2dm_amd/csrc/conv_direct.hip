// K2 (few outputs): ring-padded 3x3 convolution with Cout <= 4 as a direct VALU convolution.
//
// out_conv of the U-Net (64 -> 2 channels at full resolution, reference efficient_unet.py:267,295; ops.Conv2d + Pad,
// ops.py:32-49,149-173) wastes 30 of 32 rows of every MFMA tile; it is 0.6 GFLOP per batch of 8 against 134 MB of input,
// i.e. HBM-bound, so one pass over the input with fp32 FMAs is the right shape: block = 256 threads = 4 x 64 pixels,
// no LDS: a thread owns 4 consecutive pixels of 2 rows, reads its input window straight from global memory (its
// neighbours' lines: L1 / L2 hits) and takes the weights as wave-uniform scalar operands.
#include "common.h"
#include <type_traits>
#include <stdlib.h>
#include "wave_ops.h"

namespace r2dm {

namespace dc {
constexpr int TH = 4, TW = 64, XR = TH + 2, XS = TW + 2, CK = 8, MAXCO = 4;
}

// (Round 2's first version of this file was a 256-thread block staging 8-channel halo tiles through 17 KB of LDS,
// conv_direct_kernel<CO>.  Round 3 found it to be THE kernel behind the "wrong results next to a second process" failure --
// exact alone, a quarter-wave of wrong pixels under CU co-residency with another process' LDS-using kernel, see
// profiles/r03_shared_gpu.txt -- and removed it: the LDS-free kernels below are the only ones.)

// The same convolution without LDS, for image widths that are multiples of 4 (every U-Net geometry): a thread owns 4
// consecutive pixels of one row and, per input channel, loads the three rows it needs straight from global memory -- one
// 16-byte vector plus the left and right neighbour pixel (the neighbours' and the adjacent rows' bytes are the same cache lines
// other threads of the block fetch: L1 / L2 hits) -- with the weights as wave-uniform scalar operands.  No barrier, no
// staging: the LDS version above spent its time in 27 LDS reads per 18 FMAs and reached 1.6 TB/s of its 134 MB input;
// this one reaches 2.4 TB/s (83 -> 57 us at batch 8; the 9 load instructions per thread and channel are its limit -- a DPP wave
// shift for the neighbour pixels was tried and is not worth its select logic).  Block = 4 rows x 256 columns.
// Round 5, tried and reverted: channel batches of 4 double-buffered by hand (inline-assembly loads, counted vmcnt, one batch of 48 loads
// always in flight; 416 registers, one wave per SIMD as before): 52.5 -> 65.7 us at batch 8 (scripts/jobs/j340.sh) -- three 1 KiB-span
// load instructions per (channel, row) keep the CU's address pipeline busy either way; more in flight only queues there.
#ifndef DC_ROWS
#define DC_ROWS 2
#endif
template <int CO, bool X16 = false>  // X16 (round 6): the input tensor is stored as fp16 (the one-plane mode's activation storage)
__global__ __launch_bounds__(256) void conv_direct_rows_kernel(const ConvParams p) {
    using gcf = const float __attribute__((address_space(1)))*;
    using gcf4 = const f32x4 __attribute__((address_space(1)))*;
    using gf4 = f32x4 __attribute__((address_space(1)))*;
    constexpr int R = DC_ROWS;  // output rows per thread: R + 2 input rows serve R output rows
    const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
    const int H = p.H, W = p.W, HW = H * W;
    const int nTw = (W + 255) / 256, nTh = (H + 4 * R - 1) / (4 * R);
    int L = blockIdx.x;
    const int tw = L % nTw;
    L /= nTw;
    const int th = L % nTh;
    const int b = L / nTh;
    const int gr = (th * 4 + ty) * R, gc = tw * 256 + tx * 4;
    if (gr >= H || gc >= W) return;
    const int cl = gc == 0 ? W - 1 : gc - 1, cr = gc + 4 >= W ? gc + 4 - W : gc + 4;  // azimuth is periodic
    bool rok[R + 2];
    int rbase[R + 2];
#pragma unroll
    for (int k = 0; k < R + 2; ++k) {
        const int r = gr - 1 + k;
        rok[k] = r >= 0 && r < H;  // rows outside the image are zero padding
        rbase[k] = (rok[k] ? r : gr) * W;
    }
    float acc[CO][R][4];
#pragma unroll
    for (int o = 0; o < CO; ++o)
#pragma unroll
        for (int i = 0; i < R; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[o][i][j] = 0.f;
    // (fp16 input: the batch offset in halves)
    const float* xb0 = X16 ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(p.x.p0) + b * p.x.bs0) : p.x.p0 + b * p.x.bs0;
    const float* xb1 = !p.x.p1 ? xb0 : X16 ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(p.x.p1) + b * p.x.bs1) : p.x.p1 + b * p.x.bs1;
    const int c0 = p.x.p1 ? p.x.c0 : p.Cin;
    const gcf wg = (gcf)p.w;
#ifndef DC_UNROLL
#define DC_UNROLL 4  // (one wave per SIMD: the loads in flight hide the latency -- 2: 65.6 us, 4: 52.8 us, 8: 64.0 us at batch 8)
#endif
    if constexpr (X16) {
        // fp16 input (the one-plane mode's out_conv), a loop of its own since round 6: through the shared body below the compiler converted every row behind a full wait --
        // ~8 serial round trips per channel, 133 us per launch at batch 8 for HALF the bytes of the fp32 launch (52 us); read in the ISA, profiles/r06_tile_end_diet.txt.
        // Here the raw halves of two channels (2 x (R + 2) rows x 3 loads) are requested before the first conversion: 133 -> 52 us.  Same values into the same FMA order: bit-identical.
        using gcu = const unsigned long long __attribute__((address_space(1)))*;
        using gch = const unsigned short __attribute__((address_space(1)))*;
        const unsigned short* h0 = reinterpret_cast<const unsigned short*>(xb0);
        const unsigned short* h1 = reinterpret_cast<const unsigned short*>(xb1);
#ifndef DC16_BATCH
#define DC16_BATCH 2  // (channels per load batch at batch 8: 1: 53.0 us, 2: 52.3 us, 4: 70.8 us, 8: 68.4 us -- more registers, fewer waves; scripts/jobs/j446.sh)
#endif
        constexpr int NB = DC16_BATCH;
        for (int cb = 0; cb < p.Cin; cb += NB) {  // (Cin % 2 == 0: launcher)
            unsigned long long rv[NB][R + 2];
            unsigned short rl[NB][R + 2], rr[NB][R + 2];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int ci = cb + u;
                const unsigned short* pl = ci < c0 ? h0 + (long)ci * HW : h1 + (long)(ci - c0) * HW;
#pragma unroll
                for (int k = 0; k < R + 2; ++k) {
                    rv[u][k] = *(gcu)(pl + rbase[k] + gc);
                    rl[u][k] = ((gch)pl)[rbase[k] + cl];
                    rr[u][k] = ((gch)pl)[rbase[k] + cr];
                }
            }
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int ci = cb + u;
                float x[R + 2][6];
#pragma unroll
                for (int k = 0; k < R + 2; ++k) {
                    const f32x4 v = f16x4_to_f32(rv[u][k]);
                    const float l = (float)__builtin_bit_cast(_Float16, rl[u][k]), r = (float)__builtin_bit_cast(_Float16, rr[u][k]);
                    x[k][0] = rok[k] ? l : 0.f;
                    x[k][5] = rok[k] ? r : 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) x[k][1 + j] = rok[k] ? v[j] : 0.f;
                }
#pragma unroll
                for (int o = 0; o < CO; ++o) {
                    const gcf wk = wg + ((long)o * p.Cin + ci) * 9;
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        const float w = wk[t];
#pragma unroll
                        for (int i = 0; i < R; ++i)
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc[o][i][j] = fmaf(w, x[i + t / 3][j + t % 3], acc[o][i][j]);
                    }
                }
            }
        }
    } else
#pragma unroll DC_UNROLL
    for (int ci = 0; ci < p.Cin; ++ci) {
        // (the fp32 path is round 5's code to the letter: restructured around a shared loop body it compiled to 72 instead of 122 registers -- fewer loads in
        // flight -- and took 117 instead of 53 us)
        const gcf pl = (gcf)(ci < c0 ? xb0 + (long)ci * HW : xb1 + (long)(ci - c0) * HW);
        const float* plx = ci < c0 ? xb0 : xb1;  // (fp16 input: element offsets -- load4 / load1 address halves)
        const long e0 = (long)(ci < c0 ? ci : ci - c0) * HW;
        float x[R + 2][6];
#pragma unroll
        for (int k = 0; k < R + 2; ++k) {
            const f32x4 v = X16 ? load4<X16>(plx, e0 + rbase[k] + gc) : *(gcf4)(pl + rbase[k] + gc);
            const float l = X16 ? load1<X16>(plx, e0 + rbase[k] + cl) : pl[rbase[k] + cl], r = X16 ? load1<X16>(plx, e0 + rbase[k] + cr) : pl[rbase[k] + cr];
            x[k][0] = rok[k] ? l : 0.f;
            x[k][5] = rok[k] ? r : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) x[k][1 + j] = rok[k] ? v[j] : 0.f;
        }
#pragma unroll
        for (int o = 0; o < CO; ++o) {
            const gcf wk = wg + ((long)o * p.Cin + ci) * 9;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float w = wk[t];
#pragma unroll
                for (int i = 0; i < R; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[o][i][j] = fmaf(w, x[i + t / 3][j + t % 3], acc[o][i][j]);
            }
        }
    }
#pragma unroll
    for (int o = 0; o < CO; ++o) {
        const float bo = ((gcf)p.bias)[o];
#pragma unroll
        for (int i = 0; i < R; ++i)
            if (gr + i < H)
                *(gf4)(p.y + b * p.y_bs + (long)o * HW + (gr + i) * W + gc) = f32x4{acc[o][i][0] + bo, acc[o][i][1] + bo, acc[o][i][2] + bo, acc[o][i][3] + bo};
    }
}

// ---- few INPUT channels (in_conv over the data channels: 2 -> 64 at full resolution) -----------------------------------
// 0.3 GFLOP against 134 MB of output per batch of 8: bound by the output stream, while the fp32-MFMA kernel pads 2 input
// channels to a 16-deep k-step and spends 105 us on it.  Same thread shape as above (4 pixels of one row, wave = 256-column
// row segment, block = 4 rows): the thread keeps its 3 x 6 input window of every channel in registers and walks the output
// channels in blocks of 8 with wave-uniform scalar weights; per channel one 16-byte residual load (the constant coordinate
// map, batch-broadcast, from L2), bias, scale and one 16-byte store.  Fused GroupNorm statistics: four pixels in fp32, fp64
// beyond, one wave_sum per group, written into the slot grid the MFMA epilogues use (conv_stat_slots: 8 slots per 4 x 64
// tile and group -- this wave fills the first one of its row's four tiles and zeroes the other seven).
template <int CI, bool Y16 = false>  // Y16 (round 6): the output is stored as fp16 (RNE); statistics of the values as stored
__global__ __launch_bounds__(256) void conv_few_in_kernel(const ConvParams p) {
    using gcf = const float __attribute__((address_space(1)))*;
    using gcf4 = const f32x4 __attribute__((address_space(1)))*;
    using gf4 = f32x4 __attribute__((address_space(1)))*;
    using gdouble = double __attribute__((address_space(1)))*;
    const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
    const int H = p.H, W = p.W, HW = H * W;
    const int nTw = (W + 255) / 256, nTh = H / 4, nTw64 = (W + 63) / 64;  // (H % 4 == 0: launcher)
    int L = blockIdx.x;
    const int tw = L % nTw;
    L /= nTw;
    const int th = L % nTh;
    const int b = L / nTh;
    const int gr = th * 4 + ty, gc0 = tw * 256 + tx * 4;
    const bool active = gc0 < W;  // (partial last block of a row: the lane takes part in the reductions with zeros)
    const int gc = active ? gc0 : 0;
    const int cl = gc == 0 ? W - 1 : gc - 1, cr = gc + 4 >= W ? gc + 4 - W : gc + 4;  // azimuth is periodic
    float x[CI][3][6];
#pragma unroll
    for (int ci = 0; ci < CI; ++ci) {
        const gcf pl = (gcf)(p.x.p0 + b * p.x.bs0 + (long)ci * HW);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int r = gr - 1 + k;
            const bool ok = r >= 0 && r < H;  // rows outside the image are zero padding
            const int rb = (ok ? r : gr) * W;
            const f32x4 v = *(gcf4)(pl + rb + gc);
            const float l = pl[rb + cl], rr = pl[rb + cr];
            x[ci][k][0] = ok ? l : 0.f;
            x[ci][k][5] = ok ? rr : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) x[ci][k][1 + j] = ok ? v[j] : 0.f;
        }
    }
    // Weights, biases and the scale through the SCALAR cache (constant address space: s_load).  As plain global pointers the compiler
    // had to assume that this kernel's own stores may alias them: it fetched all 8 x 9 CI weights of a block with VECTOR loads (144
    // registers of wave-uniform values: 228 registers, two waves per SIMD) and every bias with a vector load that the store behind it
    // waited for with vmcnt(0) -- 64 serial L2 round trips per thread (round 5: ISA read; 58 us at batch 8 against ~30 us of HBM time).
    using ccf = const float __attribute__((address_space(4)))*;
    const float sc = p.scale ? *(ccf)p.scale : 1.0f;
    const ccf wg = (ccf)p.w, bg = (ccf)p.bias;
    const long pix = (long)gr * W + gc;
    const int bpg = p.stat ? p.stat_cpg >> 3 : 1;  // 8-channel blocks per group
    const int S = p.stat_slots >> 1;
    double gs = 0.0, gq = 0.0;
    // blockIdx.y: this block's share of the output channels (whole groups: launcher) -- twice / four times the waves for the same
    // output stream; the 3 x 6 input windows are re-read per share (L2 hits: the input is 1/32 of the output)
    const int cps = p.Cout / (int)gridDim.y, co_lo = (int)blockIdx.y * cps, co_hi = co_lo + cps;
    // the block's eight residual vectors are requested one block AHEAD, in front of the previous block's stores: vmcnt retires in order,
    // stores included, so a load requested behind eight stores is only usable once those stores are acknowledged
    f32x4 rv[8] = {}, rn[8] = {};
    auto res_fetch = [&](f32x4 (&r)[8], int co0) __attribute__((always_inline)) {
#pragma unroll
        for (int o = 0; o < 8; ++o) r[o] = *(gcf4)(p.res + b * p.res_bs + (long)(co0 + o) * HW + pix);
    };
    if (p.res) res_fetch(rv, co_lo);
    for (int co0 = co_lo; co0 < co_hi; co0 += 8) {
        if (p.res && co0 + 8 < co_hi) res_fetch(rn, co0 + 8);
        float acc[8][4];
#pragma unroll
        for (int o = 0; o < 8; ++o) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[o][j] = 0.f;
            const ccf wk = wg + (long)(co0 + o) * CI * 9;
#pragma unroll
            for (int ci = 0; ci < CI; ++ci)
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const float w = wk[ci * 9 + t];
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[o][j] = fmaf(w, x[ci][t / 3][j + t % 3], acc[o][j]);
                }
        }
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            const float bo = bg[co0 + o];
            f32x4 v = f32x4{acc[o][0] + bo, acc[o][1] + bo, acc[o][2] + bo, acc[o][3] + bo};
            if (p.res) v = rv[o] + v;
            v *= sc;  // (1.0f without p.scale: exact)
            if (active) {
                if constexpr (Y16) v = store4<true>(p.y, b * p.y_bs + (long)(co0 + o) * HW + pix, v);
                else *(gf4)(p.y + b * p.y_bs + (long)(co0 + o) * HW + pix) = v;
                if (p.stat) {
                    gs += (double)((v[0] + v[1]) + (v[2] + v[3]));
                    gq += (double)fmaf(v[3], v[3], fmaf(v[2], v[2], fmaf(v[1], v[1], v[0] * v[0])));
                }
            }
        }
#pragma unroll
        for (int o = 0; o < 8; ++o) rv[o] = rn[o];
        if (p.stat && ((co0 >> 3) + 1) % bpg == 0) {  // the group's last block: wave totals into the slot grid
            const double ts = wave_sum_f64(gs), tq = wave_sum_f64(gq);
            gs = gq = 0.0;
            const int g = p.stat_goff + co0 / p.stat_cpg;
            const int j = tx & 3, hf = (tx >> 2) & 1, tile = 4 * tw + j;
            if (tx < 8 && tile < nTw64) {
                const int slot = (th * nTw64 + tile) * 4 + ty;
                gdouble o = (gdouble)(p.stat + (((size_t)b * p.stat_G + g) * p.stat_slots + slot + (size_t)S * hf) * 2);
                o[0] = tx == 0 ? ts : 0.0;
                o[1] = tx == 0 ? tq : 0.0;
            }
        }
    }
}

// few inputs: Cin <= 4 data channels into Cout % 8 == 0 channels; single-source input, whole 4-row tiles, 16-byte rows
bool conv_few_in_supported(int Cin, int Cout, int taps, int H, int W) {
    return taps == 9 && Cin >= 1 && Cin <= 4 && Cout > dc::MAXCO && Cout % 8 == 0 && H % 4 == 0 && W % 4 == 0;
}

bool conv_direct_supported(int Cout, int taps) { return taps == 9 && Cout >= 1 && Cout <= dc::MAXCO; }

hipError_t launch_conv_direct(const ConvParams& p, hipStream_t s) {
    if (p.Cout > dc::MAXCO) {  // the few-input kernel
        if (!conv_few_in_supported(p.Cin, p.Cout, p.taps, p.H, p.W) || p.prologue != PRO_NONE || p.x.p1 || p.range) return hipErrorInvalidValue;
        if (p.stat && (p.stat_cpg % 8 || p.Cout % p.stat_cpg || p.stat_slots != conv_stat_slots(p.H, p.W))) return hipErrorInvalidValue;
        const unsigned nbx = (unsigned)(((p.W + 255) / 256) * (p.H / 4) * p.B);
        // shares of the output channels (blockIdx.y): whole statistics groups of whole 8-channel blocks, up to four, while the launch has
        // fewer than ~16 waves per CU
        const char* fs = getenv("R2DM_FEW_IN_SPLIT");  // (1 | 2 | 4: experiments and the bit-identity test; read per call)
        const int max_split = fs ? atoi(fs) : 2;
        const int unit = p.stat ? p.stat_cpg : 8;
        unsigned split = 1;
        while ((int)split * 2 <= max_split && p.Cout % ((int)split * 2 * unit) == 0 && nbx * split < 2048) split *= 2;
        const dim3 nb(nbx, split);
        if (p.x16) return hipErrorInvalidValue;  // (the network input is fp32)
        if (p.y16) {
            switch (p.Cin) {
                case 1: conv_few_in_kernel<1, true><<<nb, 256, 0, s>>>(p); break;
                case 2: conv_few_in_kernel<2, true><<<nb, 256, 0, s>>>(p); break;
                case 3: conv_few_in_kernel<3, true><<<nb, 256, 0, s>>>(p); break;
                default: conv_few_in_kernel<4, true><<<nb, 256, 0, s>>>(p); break;
            }
            return hipGetLastError();
        }
        switch (p.Cin) {
            case 1: conv_few_in_kernel<1><<<nb, 256, 0, s>>>(p); break;
            case 2: conv_few_in_kernel<2><<<nb, 256, 0, s>>>(p); break;
            case 3: conv_few_in_kernel<3><<<nb, 256, 0, s>>>(p); break;
            default: conv_few_in_kernel<4><<<nb, 256, 0, s>>>(p); break;
        }
        return hipGetLastError();
    }
    if (!conv_direct_supported(p.Cout, p.taps) || p.prologue != PRO_NONE || p.res || p.scale || p.stat) return hipErrorInvalidValue;
    if (p.W % 4 == 0) {  // (rows of 16-byte vectors: every U-Net geometry)
        const unsigned nb = (unsigned)(((p.W + 255) / 256) * ((p.H + 4 * DC_ROWS - 1) / (4 * DC_ROWS)) * p.B);
        if (p.y16) return hipErrorInvalidValue;  // (the network output is fp32)
        if (p.x16) {
            if (p.Cin % 2) return hipErrorInvalidValue;  // (the fp16-input loop walks the channels two at a time)
            switch (p.Cout) {
                case 1: conv_direct_rows_kernel<1, true><<<nb, 256, 0, s>>>(p); break;
                case 2: conv_direct_rows_kernel<2, true><<<nb, 256, 0, s>>>(p); break;
                case 3: conv_direct_rows_kernel<3, true><<<nb, 256, 0, s>>>(p); break;
                default: conv_direct_rows_kernel<4, true><<<nb, 256, 0, s>>>(p); break;
            }
            return hipGetLastError();
        }
        switch (p.Cout) {
            case 1: conv_direct_rows_kernel<1><<<nb, 256, 0, s>>>(p); break;
            case 2: conv_direct_rows_kernel<2><<<nb, 256, 0, s>>>(p); break;
            case 3: conv_direct_rows_kernel<3><<<nb, 256, 0, s>>>(p); break;
            default: conv_direct_rows_kernel<4><<<nb, 256, 0, s>>>(p); break;
        }
        return hipGetLastError();
    }
    return hipErrorInvalidValue;  // (W % 4 != 0: no kernel -- the U-Net geometries are multiples of 32)
}

}  // namespace r2dm

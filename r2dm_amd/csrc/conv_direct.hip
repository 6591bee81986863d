// K2 (few outputs): ring-padded 3x3 convolution with Cout <= 4 as a direct VALU convolution.
//
// out_conv of the U-Net (64 -> 2 channels at full resolution, reference efficient_unet.py:267,295; ops.Conv2d + Pad,
// ops.py:32-49,149-173) wastes 30 of 32 rows of every MFMA tile; it is 0.6 GFLOP per batch of 8 against 134 MB of input,
// i.e. HBM-bound, so one pass over the input with fp32 FMAs is the right shape: block = 256 threads = 4 x 64 pixels,
// input channels walked in chunks of 8 through an LDS halo tile (wrap in W, zero in H), weights in LDS once per block,
// every thread accumulates its pixel's Cout outputs over (channel, tap) in fp32.
#include "common.h"

namespace r2dm {

namespace dc {
constexpr int TH = 4, TW = 64, XR = TH + 2, XS = TW + 2, CK = 8, MAXCO = 4;
}

template <int CO>
__global__ __launch_bounds__(256) void conv_direct_kernel(const ConvParams p) {
    using namespace dc;
    __shared__ float xs[CK][XR][XS + 1];
    extern __shared__ float wsm[];  // [CO][Cin][9]
    const int tid = threadIdx.x, H = p.H, W = p.W, HW = H * W;
    const int nTw = (W + TW - 1) / TW, nTh = (H + TH - 1) / TH;
    int L = blockIdx.x;
    const int tw = L % nTw;
    L /= nTw;
    const int th = L % nTh;
    const int b = L / nTh;
    for (int i = tid; i < CO * p.Cin * 9; i += 256) wsm[i] = p.w[i];
    const int py = tid >> 6, px = tid & 63;
    float acc[CO];
#pragma unroll
    for (int o = 0; o < CO; ++o) acc[o] = 0.f;
    const float* xb0 = p.x.p0 + b * p.x.bs0;
    const float* xb1 = p.x.p1 ? p.x.p1 + b * p.x.bs1 : p.x.p0;
    const int c0 = p.x.c0;
    // staging map of this thread (constant over the chunks): element e = tid + 256 k of the [CK][XR][XS] halo tile
    constexpr int NE = (CK * XR * XS + 255) / 256;
    int goff[NE];   // offset inside the chunk's first plane (+ cl * HW), -1: zero padding / beyond the tile
    short lrow[NE], lcol[NE], lch[NE];
#pragma unroll
    for (int k = 0; k < NE; ++k) {
        const int e = tid + k * 256;
        const int cl = e / (XR * XS), r = (e / XS) % XR, c = e % XS;
        const int gr = th * TH + r - 1;
        int gc = tw * TW + c - 1;
        if (gc < 0) gc += W;
        while (gc >= W) gc -= W;
        const bool ok = e < CK * XR * XS && gr >= 0 && gr < H;
        goff[k] = ok ? gr * W + gc : -1;
        lch[k] = (short)(e < CK * XR * XS ? cl : -1);
        lrow[k] = (short)r;
        lcol[k] = (short)c;
    }
    for (int ci0 = 0; ci0 < p.Cin; ci0 += CK) {
        __syncthreads();  // previous chunk consumed (first pass: weights written)
        float stage[NE];
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            const int ci = ci0 + (lch[k] < 0 ? 0 : lch[k]);
            const bool ok = goff[k] >= 0 && ci < p.Cin;
            const int cc = ci < p.Cin ? ci : p.Cin - 1;
            const float* pl = cc < c0 ? xb0 + (long)cc * HW : xb1 + (long)(cc - c0) * HW;
            const float v = pl[ok ? goff[k] : 0];
            stage[k] = ok ? v : 0.f;
        }
#pragma unroll
        for (int k = 0; k < NE; ++k)
            if (lch[k] >= 0) xs[lch[k]][lrow[k]][lcol[k]] = stage[k];
        __syncthreads();
        const int nc = p.Cin - ci0 < CK ? p.Cin - ci0 : CK;
        for (int cl = 0; cl < nc; ++cl) {
            float v[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) v[t] = xs[cl][py + t / 3][px + t % 3];
#pragma unroll
            for (int o = 0; o < CO; ++o) {
                const float* wk = wsm + ((long)o * p.Cin + ci0 + cl) * 9;
#pragma unroll
                for (int t = 0; t < 9; ++t) acc[o] = fmaf(wk[t], v[t], acc[o]);
            }
        }
    }
    const int gr = th * TH + py, gc = tw * TW + px;
    if (gr < H && gc < W) {
#pragma unroll
        for (int o = 0; o < CO; ++o) p.y[b * p.y_bs + (long)o * HW + gr * W + gc] = acc[o] + p.bias[o];
    }
}

bool conv_direct_supported(int Cout, int taps) { return taps == 9 && Cout >= 1 && Cout <= dc::MAXCO; }

hipError_t launch_conv_direct(const ConvParams& p, hipStream_t s) {
    if (!conv_direct_supported(p.Cout, p.taps) || p.prologue != PRO_NONE || p.res || p.scale || p.stat) return hipErrorInvalidValue;
    const int nTw = (p.W + dc::TW - 1) / dc::TW, nTh = (p.H + dc::TH - 1) / dc::TH;
    const unsigned nblk = (unsigned)(nTw * nTh * p.B);
    const size_t lds = (size_t)p.Cout * p.Cin * 9 * sizeof(float);
    if (lds > 32768) return hipErrorInvalidValue;
    switch (p.Cout) {
        case 1: conv_direct_kernel<1><<<nblk, 256, lds, s>>>(p); break;
        case 2: conv_direct_kernel<2><<<nblk, 256, lds, s>>>(p); break;
        case 3: conv_direct_kernel<3><<<nblk, 256, lds, s>>>(p); break;
        default: conv_direct_kernel<4><<<nblk, 256, lds, s>>>(p); break;
    }
    return hipGetLastError();
}

}  // namespace r2dm

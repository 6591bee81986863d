// K6 / K7: the [1,3,3,1] FIR resamplers of the U-Net (HBM-bound, one pass, 16-byte accesses).
//
// Reference: ops.Resample (/root/reference/models/ops.py:52-146) as used at
// efficient_unet.py:135 (down=2 after the stage's 3x3 conv) and :171 (up=2 before it).
//   down: y[i,j] = sum_a k_a ( sum_b k_b xp[2i+a-1, 2j+b-1] ),  k = [1,3,3,1]/8
//   up  : per axis  y[2i] = x[i-1]/4 + 3x[i]/4 ,  y[2i+1] = 3x[i]/4 + x[i+1]/4   (zero-insert + [1,3,3,1]/4)
// xp / x[-1] / x[H]: columns wrap (azimuth is periodic), rows outside the image are zero.
// W pass first, then H, like the reference's two depthwise convolutions (ops.py:131-132).
// The FIR window is fixed: the host side refuses checkpoints whose `kernel` buffers differ.
#include "common.h"
#include <stdlib.h>
#include "wave_ops.h"

namespace r2dm {

// one thread -> two horizontally adjacent outputs; needs input columns 4t-1 .. 4t+4 of 4 rows
__global__ __launch_bounds__(256) void fir_down2_kernel(const float* __restrict__ x, long xbs, float* __restrict__ y,
                                                        long ybs, int C, int H, int W) {
    const int Ho = H >> 1, Wo = W >> 1, Wq = Wo >> 1;  // Wq threads per output row
    const long per_plane = (long)Ho * Wq;
    const long total = per_plane * C;
    const int b = blockIdx.y;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = idx / per_plane;
        const long rem = idx % per_plane;
        const int i = rem / Wq, t = rem % Wq;
        const float* xp = x + b * xbs + (long)c * H * W;
        const int cl = 4 * t - 1 < 0 ? W - 1 : 4 * t - 1;
        const int cr = 4 * t + 4 >= W ? 0 : 4 * t + 4;
        float o0 = 0.f, o1 = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int r = 2 * i + a - 1;
            float h0 = 0.f, h1 = 0.f;
            if (r >= 0 && r < H) {
                const float* row = xp + (long)r * W;
                const f32x4 m = *reinterpret_cast<const f32x4*>(row + 4 * t);
                const float l = row[cl], rr = row[cr];
                h0 = 0.125f * l + 0.375f * m[0] + 0.375f * m[1] + 0.125f * m[2];
                h1 = 0.125f * m[1] + 0.375f * m[2] + 0.375f * m[3] + 0.125f * rr;
            }
            const float ka = (a == 0 || a == 3) ? 0.125f : 0.375f;
            o0 += ka * h0;
            o1 += ka * h1;
        }
        float2* out = reinterpret_cast<float2*>(y + b * ybs + (long)c * Ho * Wo + (long)i * Wo + 2 * t);
        *out = make_float2(o0, o1);
    }
}

// one thread -> a 2 x 4 output patch (rows 2i, 2i+1 of the block's pair, columns 4t .. 4t+3): input columns 8t-1 .. 8t+8 of
// 6 rows = 24 load instructions for 8 outputs.  (One output pair per thread was 12 loads for 2 outputs and bound by the CU's
// address pipeline, not by HBM: 3.7 TB/s.)  Needs W % 8 == 0, H % 4 == 0; the generic kernel below takes the rest.
__global__ __launch_bounds__(256) void fir_down2_wide_kernel(const float* __restrict__ x, long xbs, float* __restrict__ y,
                                                             long ybs, int C, int H, int W) {
    const int Ho = H >> 1, Wo = W >> 1, Wq = Wo >> 2, Hq = Ho >> 1;  // Wq threads per pair of output rows
    const long per_plane = (long)Hq * Wq;
    const long total = per_plane * C;
    const int b = blockIdx.y;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = idx / per_plane;
        const long rem = idx % per_plane;
        const int i2 = rem / Wq, t = rem % Wq;
        const float* xp = x + b * xbs + (long)c * H * W;
        const int cl = 8 * t - 1 < 0 ? W - 1 : 8 * t - 1;
        const int cr = 8 * t + 8 >= W ? 0 : 8 * t + 8;
        float h[6][4];  // horizontally filtered rows 4 i2 - 1 .. 4 i2 + 4 at the four output columns
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            const int r = 4 * i2 + a - 1;
            if (r >= 0 && r < H) {
                const float* row = xp + (long)r * W;
                const f32x4 m0 = *reinterpret_cast<const f32x4*>(row + 8 * t), m1 = *reinterpret_cast<const f32x4*>(row + 8 * t + 4);
                const float l = row[cl], rr = row[cr];
                h[a][0] = 0.125f * l + 0.375f * m0[0] + 0.375f * m0[1] + 0.125f * m0[2];
                h[a][1] = 0.125f * m0[1] + 0.375f * m0[2] + 0.375f * m0[3] + 0.125f * m1[0];
                h[a][2] = 0.125f * m0[3] + 0.375f * m1[0] + 0.375f * m1[1] + 0.125f * m1[2];
                h[a][3] = 0.125f * m1[1] + 0.375f * m1[2] + 0.375f * m1[3] + 0.125f * rr;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) h[a][j] = 0.f;
            }
        }
#pragma unroll
        for (int o = 0; o < 2; ++o) {  // output row 2 i2 + o takes input rows (2 o) .. (2 o + 3) of the six; same order of
            f32x4 v;                    // additions as the generic kernel: bit-identical results
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float acc = 0.f;
#pragma unroll
                for (int a = 0; a < 4; ++a) acc += ((a == 0 || a == 3) ? 0.125f : 0.375f) * h[2 * o + a][j];
                v[j] = acc;
            }
            *reinterpret_cast<f32x4*>(y + b * ybs + (long)c * Ho * Wo + (long)(2 * i2 + o) * Wo + 4 * t) = v;
        }
    }
}

// one thread -> input columns 2t, 2t+1 of row i -> a 2x4 output patch
// range (optional): the running maximum of |output| is merged into range[1] as float bits (positive floats order like their
// bit patterns) -- the f16x2 convolution that consumes this tensor needs it below 65504 (engine.hip, r2dm_check_range)
__global__ __launch_bounds__(256) void fir_up2_kernel(const float* __restrict__ x, long xbs, float* __restrict__ y,
                                                      long ybs, int C, int H, int W, int* __restrict__ range) {
    float amax = 0.f;
    const int Wh = W >> 1, Wo = W << 1;
    const long per_plane = (long)H * Wh;
    const long total = per_plane * C;
    const int b = blockIdx.y;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = idx / per_plane;
        const long rem = idx % per_plane;
        const int i = rem / Wh, t = rem % Wh;
        const float* xp = x + b * xbs + (long)c * H * W;
        const int cl = 2 * t - 1 < 0 ? W - 1 : 2 * t - 1;
        const int cr = 2 * t + 2 >= W ? 0 : 2 * t + 2;
        float h[3][4];  // horizontally upsampled rows i-1, i, i+1 at output columns 4t..4t+3
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int r = i + a - 1;
            if (r >= 0 && r < H) {
                const float* row = xp + (long)r * W;
                const float2 m = *reinterpret_cast<const float2*>(row + 2 * t);
                const float l = row[cl], rr = row[cr];
                h[a][0] = l * 0.25f + m.x * 0.75f;
                h[a][1] = m.x * 0.75f + m.y * 0.25f;
                h[a][2] = m.x * 0.25f + m.y * 0.75f;
                h[a][3] = m.y * 0.75f + rr * 0.25f;
            } else {
                h[a][0] = h[a][1] = h[a][2] = h[a][3] = 0.f;
            }
        }
        f32x4 e, o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            e[j] = h[0][j] * 0.25f + h[1][j] * 0.75f;
            o[j] = h[1][j] * 0.75f + h[2][j] * 0.25f;
        }
        float* out = y + b * ybs + (long)c * (4L * H * W) + (long)(2 * i) * Wo + 4 * t;
        *reinterpret_cast<f32x4*>(out) = e;
        *reinterpret_cast<f32x4*>(out + Wo) = o;
        if (range) {
#pragma unroll
            for (int j = 0; j < 4; ++j) amax = fmaxf(amax, fmaxf(fabsf(e[j]), fabsf(o[j])));
        }
    }
    if (range) {
        amax = wave_max_f32(amax);
        if ((threadIdx.x & 63) == 0) {
            const int bits = __float_as_int(amax);
            if (bits > __atomic_load_n(range + 1, __ATOMIC_RELAXED)) atomicMax(range + 1, bits);  // (rarely taken after the first blocks)
        }
    }
}

static int grid_for(long total) {
    long g = (total + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

hipError_t launch_fir_down2(const float* x, long xbs, float* y, long ybs, int B, int C, int H, int W, hipStream_t s) {
    if ((W & 3) || (H & 1)) return hipErrorInvalidValue;
    if (W % 8 == 0 && H % 4 == 0 && getenv("R2DM_FIR_NARROW") == nullptr) {
        const long tot = (long)C * (H / 4) * (W / 8);
        fir_down2_wide_kernel<<<dim3(grid_for(tot), B), 256, 0, s>>>(x, xbs, y, ybs, C, H, W);
        return hipGetLastError();
    }
    const long total = (long)C * (H / 2) * (W / 4);
    fir_down2_kernel<<<dim3(grid_for(total), B), 256, 0, s>>>(x, xbs, y, ybs, C, H, W);
    return hipGetLastError();
}

hipError_t launch_fir_up2(const float* x, long xbs, float* y, long ybs, int B, int C, int H, int W, hipStream_t s, int* range) {
    if (W & 1) return hipErrorInvalidValue;
    const long total = (long)C * H * (W / 2);
    fir_up2_kernel<<<dim3(grid_for(total), B), 256, 0, s>>>(x, xbs, y, ybs, C, H, W, range);
    return hipGetLastError();
}

}  // namespace r2dm

// K1: timestep embedding MLP and all AdaGN projections of one forward pass.
//
// Reference: SinusoidalPositionalEmbedding (/root/reference/models/ops.py:14-29) -> Linear -> SiLU ->
// Linear (/root/reference/models/efficient_unet.py:232-237,275), and per residual block
// AdaGN.proj = SiLU -> Linear(temb, 2C) (/root/reference/models/ops.py:190-200).
// The only consumer of the time embedding is SiLU(temb), so that is what is stored; the 24
// projection matrices are packed row-wise into one [rows][T] matrix and evaluated in one launch.
// The frequency table f_k is precomputed on the host with the reference's own expression.
#include "common.h"
#include "wave_ops.h"

namespace r2dm {

// These vectors condition EVERY AdaGN layer (a common-mode error here is seen 24 times), and the whole kernel
// is a few hundred KFLOP: accumulate in fp64 and use the accurate transcendental paths, so that the embedding
// and the projections are correctly rounded fp32 values of the reference expressions.
__device__ __forceinline__ double wave_sum_d(double v) { return wave_sum_f64(v); }  // (wave_ops.h)

__device__ __forceinline__ float silu_exact(float v) {
    const double d = (double)v;
    return (float)(d / (1.0 + exp(-d)));
}

// Two launches so that the 64 KB + 256 KB weight matrices are streamed by many CUs instead of one block per
// sample (one block per sample took 160 us -- 1 % of a sampling step -- for a few hundred KFLOP):
//   hidden: grid (T/4, B): emb = [sin, cos](t * f_k) recomputed per block (cheap), wave w -> row 4*bx + w of W1
//   output: grid (T/4, B): wave w -> row 4*bx + w of W2
__global__ __launch_bounds__(256) void time_hidden_kernel(EmbedParams p, float* __restrict__ hid) {
    extern __shared__ float emb[];  // [base]
    const int b = blockIdx.y, half = p.base / 2;
    const float t = p.cond[b];
    for (int k = threadIdx.x; k < half; k += 256) {
        const float arg = t * p.freqs[k];  // fp32 product, as the reference forms it (ops.py:24)
        emb[k] = (float)sin((double)arg);
        emb[half + k] = (float)cos((double)arg);
    }
    __syncthreads();
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= p.T) return;
    double acc = 0.0;
    for (int k = lane; k < p.base; k += 64) acc += (double)p.w1[(long)r * p.base + k] * (double)emb[k];
    acc = wave_sum_d(acc);
    if (lane == 0) hid[(long)b * p.T + r] = silu_exact((float)(acc + (double)p.b1[r]));
}

__global__ __launch_bounds__(256) void time_out_kernel(EmbedParams p, const float* __restrict__ hid) {
    const int b = blockIdx.y;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= p.T) return;
    double acc = 0.0;
    for (int k = lane; k < p.T; k += 64) acc += (double)p.w2[(long)r * p.T + k] * (double)hid[(long)b * p.T + k];
    acc = wave_sum_d(acc);
    if (lane == 0) p.act[(long)b * p.T + r] = silu_exact((float)(acc + (double)p.b2[r]));
}

hipError_t launch_time_embedding(const EmbedParams& p, hipStream_t s) {
    if (!p.hidden) return hipErrorInvalidValue;
    const dim3 g((p.T + 3) / 4, p.B);
    time_hidden_kernel<<<g, 256, p.base * sizeof(float), s>>>(p, p.hidden);
    time_out_kernel<<<g, 256, 0, s>>>(p, p.hidden);
    return hipGetLastError();
}

// grid = ceil(rows/4), block = 256: wave w of a block owns row 4*blockIdx.x + w.  Eight samples at a time: the row's weights are
// read once, the eight dot products share one butterfly reduce-scatter (wave_sum8_scatter: 10 exchange steps instead of 8 x 6), and
// lane 8 j writes sample j.  (One sample after the other with a full reduction each was 20 us at batch 8 -- the first thing a
// forward waits for.)
__global__ __launch_bounds__(256) void ada_proj_kernel(const float* __restrict__ act, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ out,
                                                       int B, int T, int rows) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= rows) return;
    const float* wr = w + (long)r * T;
    for (int b0 = 0; b0 < B; b0 += 8) {
        double acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.0;
        if ((T & 255) == 0) {  // (every U-Net: T = 256) 16-byte loads -- the row and the eight activations are 9 requests per lane instead of 36
            for (int k = 4 * lane; k < T; k += 256) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + k);
                f32x4 av[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) av[j] = *reinterpret_cast<const f32x4*>(act + (long)(b0 + j < B ? b0 + j : b0) * T + k);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (b0 + j < B) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[j] += (double)wv[e] * (double)av[j][e];
                    }
            }
        } else {
            for (int k = lane; k < T; k += 64) {
                const double wv = (double)wr[k];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (b0 + j < B) acc[j] += wv * (double)act[(long)(b0 + j) * T + k];
            }
        }
        const double t = wave_sum8_scatter(acc, lane);  // lane L: the total of sample b0 + ((L >> 3) & 7)
        const int j = (lane >> 3) & 7;
        if ((lane & 7) == 0 && b0 + j < B) out[(long)(b0 + j) * rows + r] = (float)(t + (double)bias[r]);
    }
}

hipError_t launch_ada_proj(const float* act, const float* w, const float* bias, float* out, int B, int T, int rows,
                           hipStream_t s) {
    ada_proj_kernel<<<(rows + 3) / 4, 256, 0, s>>>(act, w, bias, out, B, T, rows);
    return hipGetLastError();
}

}  // namespace r2dm

// K1: timestep embedding MLP and all AdaGN projections of one forward pass.
//
// Reference: SinusoidalPositionalEmbedding (/root/reference/models/ops.py:14-29) -> Linear -> SiLU ->
// Linear (/root/reference/models/efficient_unet.py:232-237,275), and per residual block
// AdaGN.proj = SiLU -> Linear(temb, 2C) (/root/reference/models/ops.py:190-200).
// The only consumer of the time embedding is SiLU(temb), so that is what is stored; the 24
// projection matrices are packed row-wise into one [rows][T] matrix and evaluated in one launch.
// The frequency table f_k is precomputed on the host with the reference's own expression.
#include "common.h"

namespace r2dm {

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// grid = B, block = 256.  One wave per output row, lanes stride the reduction dimension.
__global__ __launch_bounds__(256) void time_embedding_kernel(EmbedParams p) {
    extern __shared__ float sm[];  // [base] sinusoid, then [T] hidden
    float* emb = sm;
    float* hid = sm + p.base;
    const int b = blockIdx.x, half = p.base / 2;
    const float t = p.cond[b];
    for (int k = threadIdx.x; k < half; k += 256) {
        const float arg = t * p.freqs[k];
        emb[k] = sinf(arg);
        emb[half + k] = cosf(arg);
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int r = wave; r < p.T; r += 4) {
        float acc = 0.f;
        for (int k = lane; k < p.base; k += 64) acc += p.w1[(long)r * p.base + k] * emb[k];
        acc = wave_sum_f(acc);
        if (lane == 0) hid[r] = silu_f(acc + p.b1[r]);
    }
    __syncthreads();
    for (int r = wave; r < p.T; r += 4) {
        float acc = 0.f;
        for (int k = lane; k < p.T; k += 64) acc += p.w2[(long)r * p.T + k] * hid[k];
        acc = wave_sum_f(acc);
        if (lane == 0) p.act[(long)b * p.T + r] = silu_f(acc + p.b2[r]);
    }
}

hipError_t launch_time_embedding(const EmbedParams& p, hipStream_t s) {
    time_embedding_kernel<<<p.B, 256, (p.base + p.T) * sizeof(float), s>>>(p);
    return hipGetLastError();
}

// grid = ceil(rows/4), block = 256: wave w of a block owns row 4*blockIdx.x + w and loops over samples.
__global__ __launch_bounds__(256) void ada_proj_kernel(const float* __restrict__ act, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ out,
                                                       int B, int T, int rows) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= rows) return;
    const float* wr = w + (long)r * T;
    for (int b = 0; b < B; ++b) {
        float acc = 0.f;
        for (int k = lane; k < T; k += 64) acc += wr[k] * act[(long)b * T + k];
        acc = wave_sum_f(acc);
        if (lane == 0) out[(long)b * rows + r] = acc + bias[r];
    }
}

hipError_t launch_ada_proj(const float* act, const float* w, const float* bias, float* out, int B, int T, int rows,
                           hipStream_t s) {
    ada_proj_kernel<<<(rows + 3) / 4, 256, 0, s>>>(act, w, bias, out, B, T, rows);
    return hipGetLastError();
}

}  // namespace r2dm

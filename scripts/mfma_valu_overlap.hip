// Calibration probe (not product): does VALU work overlap with v_mfma_f32_32x32x2_f32 on gfx950?
//  mode 0: 4 waves/block, each: 72 MFMA + NV independent v_fma per iteration (same wave)
//  mode 1: 8 waves/block: waves 0-3 MFMA only, waves 4-7 VALU only (NV*... per iteration), same SIMDs
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int NV, int MODE>
__global__ __launch_bounds__(512) void probe(const float* in, float* out, int iters) {
    const int tid = threadIdx.x;
    const bool valu_wave = MODE == 1 && tid >= 256;
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float a0 = in[tid], a1 = in[tid + 256], b0 = in[tid + 512], b1 = in[tid + 768];
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = in[tid + j];
    for (int it = 0; it < iters; ++it) {
        if (!valu_wave) {
#pragma unroll
            for (int s = 0; s < 18; ++s) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3], 0, 0, 0);
                if (MODE == 0) {
#pragma unroll
                    for (int j = 0; j < NV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j & 7]) : "v"(a0), "v"(b0));
                }
            }
        } else {
#pragma unroll
            for (int s = 0; s < 18; ++s)
#pragma unroll
                for (int j = 0; j < NV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j & 7]) : "v"(a0), "v"(b0));
        }
    }
    float s = 0;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * 512 + tid] = s;
}

template <int NV, int MODE>
void run(const float* in, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * 4, iters = 64, threads = MODE ? 512 : 256;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<NV, MODE>), dim3(grid), dim3(threads), 0, 0, in, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * 32 * 32 * 2 * 72.0 * iters * 4 * grid;
    printf("mode %d  %2d v_fma per 4 MFMA: %.1f us  MFMA rate %.1f TF/s (%.1f%%)\n", MODE, NV, ms * 1e3, flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100);
}

int main() {
    float *in, *out; hipMalloc(&in, 1 << 20); hipMalloc(&out, 64 << 20); hipMemset(in, 0, 1 << 20);
    run<0, 0>(in, out); run<4, 0>(in, out); run<8, 0>(in, out); run<16, 0>(in, out); run<32, 0>(in, out);
    run<0, 1>(in, out); run<4, 1>(in, out); run<8, 1>(in, out); run<16, 1>(in, out); run<32, 1>(in, out);
    return 0;
}

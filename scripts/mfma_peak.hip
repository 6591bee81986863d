// Calibration probe (not product): sustained fp32 MFMA rate of v_mfma_f32_32x32x2_f32 on this chip for
// (a) a register-only stream, (b) a stream whose operands come from LDS each step, at 1..3 blocks per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int MODE>
__global__ __launch_bounds__(256) void probe(const float* in, float* out, int iters, int lds_pad) {
    extern __shared__ float sm[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 8192; i += 256) sm[i] = in[i];
    __syncthreads();
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float a0 = in[tid], a1 = in[tid + 256], b0 = in[tid + 512], b1 = in[tid + 768];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 18; ++s) {
            if (MODE == 1) {
                a0 = sm[lane + s * 64]; a1 = sm[lane + s * 64 + 32];
                b0 = sm[2048 + lane + s * 66]; b1 = sm[2048 + lane + s * 66 + 32];
            }
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3], 0, 0, 0);
        }
    }
    float s = 0;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * 256 + tid] = s;
}

int main() {
    float *in, *out;
    hipMalloc(&in, 1 << 20); hipMalloc(&out, 64 << 20);
    std::vector<float> h(1 << 18);
    for (auto& v : h) v = (rand() % 2001 - 1000) / 1000.f;
    hipMemcpy(in, h.data(), 1 << 20, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode)
        for (int bpc : {1, 2, 3, 4}) {
            const int lds = 160 * 1024 / bpc - 1024;  // forces bpc blocks per CU
            const int grid = 256 * bpc * 4, iters = 64;  // 4 rounds
            auto k = mode ? probe<1> : probe<0>;
            hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, in, out, iters, 0);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = 2.0 * 32 * 32 * 2 * 72.0 * iters * 4 /*waves*/ * grid;
            printf("mode %s blocks/CU %d grid %d: %.1f us  %.1f TF/s (%.1f%% of 157.3)\n", mode ? "lds " : "regs", bpc, grid, ms * 1e3,
                   flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100);
        }
    return 0;
}

// Calibration probe (not product; hipcc --offload-arch=gfx950 -O3): how much does VALU work of a SECOND wave on the same SIMD
// cost a stream of v_mfma_f32_32x32x16_f16 -- and does it matter whether the accumulators live in ArchVGPRs or AccVGPRs?
//
// conv_f16x2_kernel runs one multiplying wave (12 MFMAs per tap on 8 accumulators of 16 registers) and one staging wave (per
// chunk ~290 VALU instructions, 64 of them quarter-rate transcendentals) on every SIMD; the round-2 timelines show the stager
// at ~2.4x its pure issue time and the multiplier at 1.15x.  An MFMA moves 24 source + 16 destination registers through the
// register file; a VALU instruction 3 + 1.  If the two compete for ArchVGPR bandwidth, AccVGPR accumulators (their own
// file next to the matrix core) should take most of the MFMA's traffic out of the way.
//
//   block = 8 waves: waves 0..3 run ITERS x 12 MFMAs (ACC = 0: builtin, accumulators where the compiler puts them = ArchVGPRs
//   under launch_bounds(512, 2); ACC = 1: inline asm with "a" constraints = AccVGPRs); waves 4..7 run ITERS x NV VALU
//   instructions (KIND 0: v_fma_f32 on 8 independent chains; KIND 1: the stager's mix, 2 transcendentals per 9).
//   One block per CU (grid 256), s_memtime around the loop of wave 0 and wave 4 of block 0.
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int NV, int KIND, int ACC>
__global__ __launch_bounds__(512, 2) void probe(const float* in, float* out, unsigned long long* clk, int iters) {
    extern __shared__ unsigned char pad[];  // (150 KB requested: one block per CU)
    const int tid = threadIdx.x;
    const bool valu_wave = tid >= 256;
    unsigned long long t0 = 0, t1 = 0;
    if (!valu_wave) {
        f32x16 acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        f16x8 a[4], b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 8; ++r) { a[j][r] = (_Float16)in[(tid + j * 8 + r) & 1023]; b[j][r] = (_Float16)in[(tid + 64 + j * 8 + r) & 1023]; }
        if (ACC == 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("" : "+a"(acc[j]));
        }
        __syncthreads();
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 12; ++u) {  // the kernel's order: 4 products into the "cross" accumulators twice, then the 4 "main" ones
                const int j = u < 8 ? 4 + (u & 3) : (u & 3);
                if (ACC == 0) {
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u & 3], b[(u >> 1) & 3], acc[j], 0, 0, 0);
                } else {
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[j]) : "v"(a[u & 3]), "v"(b[(u >> 1) & 3]));
                }
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_nop 15\n\ts_nop 15");
        float s = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[j][r];
        out[blockIdx.x * 512 + tid] = s;
    } else {
        float v[8], m[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { v[j] = in[tid + j]; m[j] = in[tid + 8 + j]; }
        const float c0 = in[tid & 63], c1 = in[(tid & 63) + 64];
        __syncthreads();
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
            if (KIND == 0) {
#pragma unroll
                for (int j = 0; j < NV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j & 7]) : "v"(c0), "v"(c1));
            } else {  // affine, -log2e, exp2, +1, rcp, mul, (split: cvt, mul, fma) ~ 9 instructions per value, 2 quarter rate
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    const int k = j & 7, s = (j >> 3) % 9;
                    if (s == 2) asm volatile("v_exp_f32 %0, %1" : "=v"(m[k]) : "v"(v[k]));
                    else if (s == 4) asm volatile("v_rcp_f32 %0, %1" : "=v"(m[k]) : "v"(m[k]));
                    else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(c0), "v"(m[k]));
                }
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
        float s = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[j] + m[j];
        out[blockIdx.x * 512 + tid] = s;
    }
    if (blockIdx.x == 0 && (tid == 0 || tid == 256)) clk[tid ? 1 : 0] = t1 - t0;
}

template <int NV, int KIND, int ACC>
int run(const float* in, float* out, unsigned long long* clk) {
    const int iters = 400, lds = 150 * 1024;
    auto k = probe<NV, KIND, ACC>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(256), dim3(512), lds, 0, in, out, clk, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    unsigned long long c[2];
    CK(hipMemcpy(c, clk, sizeof(c), hipMemcpyDeviceToHost));
    const double tf = 2.0 * 32 * 32 * 16 * 12.0 * iters * 4 * 256 / (ms * 1e-3) / 1e12;
    printf("acc in %s, VALU kind %d, %3d VALU per 12 MFMA: multiplier %6.1f cycles per 12 MFMA (384 = pipe bound), VALU wave %6.1f cycles per group (%.2f per instruction); kernel %.1f us, %.0f TF/s\n",
           ACC ? "AccVGPR " : "ArchVGPR", KIND, NV, (double)c[0] / iters, (double)c[1] / iters, NV ? (double)c[1] / iters / NV : 0.0, ms * 1e3, tf);
    return 0;
}

int main() {
    float *in, *out;
    unsigned long long* clk;
    CK(hipMalloc(&in, 1 << 20));
    CK(hipMalloc(&out, 64 << 20));
    CK(hipMalloc(&clk, 64));
    CK(hipMemset(in, 0, 1 << 20));
    run<0, 0, 0>(in, out, clk); run<16, 0, 0>(in, out, clk); run<32, 0, 0>(in, out, clk); run<48, 0, 0>(in, out, clk); run<72, 0, 0>(in, out, clk);
    run<0, 0, 1>(in, out, clk); run<16, 0, 1>(in, out, clk); run<32, 0, 1>(in, out, clk); run<48, 0, 1>(in, out, clk); run<72, 0, 1>(in, out, clk);
    run<36, 1, 0>(in, out, clk); run<72, 1, 0>(in, out, clk);
    run<36, 1, 1>(in, out, clk); run<72, 1, 1>(in, out, clk);
    return 0;
}

// GPU-box probe (standalone: hipcc f16x2_probe.hip -o f16x2_probe): accuracy of split schemes on the real matrix pipe.
//   bf16x3 : fp32 = three bf16 pieces, six products, one fp32 accumulator            (the round-1 / round-2a kernels)
//   f16x2  : fp32 = fp16 piece + 2^11-scaled fp16 residual, three products, the a0*b0 products in one accumulator and the
//            two cross products in a second one (combined as hi + 2^-11 lo at the end)
//   f32    : v_mfma_f32_32x32x2_f32
// One wave computes a 32x32 output with K = 576 ... 4608; errors are against an fp64 host reference.  Also reports the
// mean signed error (accumulator truncation bias) and a denormal-heavy input set.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ inline float bf16_round(float v) {  // RNE to bf16, returned as float
    unsigned u = __float_as_uint(v);
    u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
    return __uint_as_float(u);
}
// mode 0: bf16x3, 1: f16x2 hi/lo, 2: f16x2 hi/lo with the sign pattern (blocks of 32 k negated, accumulators flipped), 3: f32 MFMA,
// 4: f16x2 single accumulator (unscaled residuals), 5: single accumulator, the B operand (weights) pre-scaled by 2^13 so that its
// residual is a normal fp16 number, the A residual unscaled (subnormal below |a| ~ 0.25); result scaled back by 2^-13
__global__ void probe(const float* A, const float* B, float* C, int K, int mode) {
    const int lane = threadIdx.x, r32 = lane & 31, hi = lane >> 5;
    f32x16 acc = {0}, lo = {0};
    const float S = 2048.f;
    for (int k0 = 0; k0 < K; k0 += 16) {
        float a[8], b[8];
        for (int i = 0; i < 8; ++i) { a[i] = A[r32 * K + k0 + hi * 8 + i]; b[i] = B[(k0 + hi * 8 + i) * 32 + r32]; }
        const bool neg = (mode == 2) && ((k0 >> 5) & 1);
        if (mode == 0) {
            bf16x8 ap[3], bp[3];
            for (int i = 0; i < 8; ++i) {
                float ra = a[i], rb = b[i];
                for (int p = 0; p < 3; ++p) {
                    float pa = bf16_round(ra), pb = bf16_round(rb);
                    ap[p][i] = (__bf16)pa; bp[p][i] = (__bf16)pb; ra -= pa; rb -= pb;
                }
            }
            const int PI[6] = {2, 0, 1, 1, 0, 0}, PJ[6] = {0, 2, 1, 0, 1, 0};
            for (int q = 0; q < 6; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[PI[q]], bp[PJ[q]], acc, 0, 0, 0);
        } else if (mode == 1 || mode == 2 || mode == 4 || mode == 5) {
            f16x8 a0, a1, b0, b1;
            for (int i = 0; i < 8; ++i) {
                const float bb = (neg ? -b[i] : b[i]) * (mode == 5 ? 8192.f : 1.f);
                a0[i] = (_Float16)a[i]; b0[i] = (_Float16)bb;
                const float sc = (mode == 4 || mode == 5) ? 1.f : S;
                a1[i] = (_Float16)((a[i] - (float)a0[i]) * sc); b1[i] = (_Float16)((bb - (float)b0[i]) * sc);
            }
            if (mode == 4 || mode == 5) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc, 0, 0, 0);
            } else {
                lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, lo, 0, 0, 0);
                lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, lo, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc, 0, 0, 0);
                if (mode == 2 && (k0 & 16)) { for (int r = 0; r < 16; ++r) { acc[r] = -acc[r]; lo[r] = -lo[r]; } }
            }
        } else {
            for (int i = 0; i < 8; ++i) {  // 32x32x2 f32: lane holds one k of two; feed k = hi*8+i pairs sequentially
                // k index pairing: MFMA 32x32x2 takes k = hi (0/1); emulate by 8 calls where lanes with hi=0 carry k0+i, hi=1 carry k0+8+i
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], acc, 0, 0, 0);
            }
        }
    }
    for (int r = 0; r < 16; ++r) {
        const int row = (r / 4) * 8 + hi * 4 + (r % 4), col = r32;
        float v = acc[r];
        if (mode == 1 || mode == 2) v = acc[r] + lo[r] * (1.f / 2048.f);
        if (mode == 5) v *= 1.f / 8192.f;
        C[row * 32 + col] = v;
    }
}

int main() {
    std::mt19937_64 g(1);
    std::normal_distribution<double> nd(0, 1);
    const char* names[6] = {"bf16x3", "f16x2 hi/lo", "f16x2 hi/lo +signs", "f32 mfma", "f16x2 single unscaled", "f16x2 single, weights x 2^13"};
    for (int tiny = 0; tiny < 2; ++tiny)
        for (int K : {576, 1152, 4608}) {
            std::vector<float> A(32 * K), B(K * 32), C(32 * 32);
            for (auto& v : A) { double x = nd(g); v = (float)(x / (1 + std::exp(-x))); if (tiny && (g() % 4 == 0)) v *= 1e-5f; }
            for (auto& v : B) { v = (float)(nd(g) / std::sqrt((double)K)); if (tiny && (g() % 4 == 0)) v *= 1e-4f; }
            std::vector<double> T(32 * 32, 0.0);
            for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { double s = 0; for (int k = 0; k < K; ++k) s += (double)A[i * K + k] * (double)B[k * 32 + j]; T[i * 32 + j] = s; }
            double sc = 0; for (double t : T) sc += t * t; sc = std::sqrt(sc / T.size());
            float *dA, *dB, *dC;
            CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, C.size() * 4));
            CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
            printf("K=%d %s:", K, tiny ? "(quarter of the operands scaled by 1e-5 / 1e-4)" : "");
            for (int mode = 0; mode < 6; ++mode) {
                hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, mode);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
                double e2 = 0, bias = 0;
                for (size_t i = 0; i < T.size(); ++i) { double d = C[i] - T[i]; e2 += d * d; bias += d; }
                printf("  %s rms %.2e bias %+.1e |", names[mode], std::sqrt(e2 / T.size()) / sc, bias / T.size() / sc);
            }
            printf("\n");
            CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
        }
    return 0;
}

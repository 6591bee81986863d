// Calibration probe (not product): accuracy of an fp32 dot product emulated with split-bf16 MFMAs on gfx950.
// C[32x32] = A[32xK] * B[Kx32]; variants: fp32 MFMA chain, bf16 x1, x3 (a1b1+a1b2+a2b1), x6, x9; reference fp64 on host.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u16x8 = __attribute__((ext_vector_type(8))) unsigned short;

__device__ inline void split3(float a, unsigned short& h1, unsigned short& h2, unsigned short& h3) {
    const unsigned u1 = __float_as_uint(a) & 0xffff0000u;
    const float r1 = a - __uint_as_float(u1);
    const unsigned u2 = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(u2);
    h1 = u1 >> 16; h2 = u2 >> 16; h3 = __float_as_uint(r2) >> 16;
}

// mode 0: fp32 mfma; 1: bf16 x1 (truncated); 3: x3; 6: x6; 9: x9
__global__ void dot_kernel(const float* A, const float* B, float* C, int K, int mode) {
    const int lane = threadIdx.x, l31 = lane & 31, hi = lane >> 5;
    f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (mode == 0) {
        for (int k = 0; k < K; k += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[l31 * K + k + hi], B[(k + hi) * 32 + l31], acc, 0, 0, 0);
    } else {
        for (int k = 0; k < K; k += 16) {
            u16x8 a[3], b[3];
            for (int i = 0; i < 8; ++i) {
                unsigned short h1, h2, h3;
                split3(A[l31 * K + k + 8 * hi + i], h1, h2, h3); a[0][i] = h1; a[1][i] = h2; a[2][i] = h3;
                split3(B[(k + 8 * hi + i) * 32 + l31], h1, h2, h3); b[0][i] = h1; b[1][i] = h2; b[2][i] = h3;
            }
            auto mm = [&](int i, int j) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), acc, 0, 0, 0);
            };
            // smallest terms first
            if (mode >= 9) { mm(2, 2); mm(1, 2); mm(2, 1); }
            if (mode >= 6) { mm(0, 2); mm(2, 0); mm(1, 1); }
            if (mode >= 3) { mm(0, 1); mm(1, 0); }
            mm(0, 0);
        }
    }
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + l31] = acc[r];
}

int main() {
    for (int K : {576, 4608}) {
        for (int dist = 0; dist < 2; ++dist) {
            std::vector<float> A(32 * K), B(K * 32);
            srand(1234 + K + dist);
            auto rnd = [] { return (rand() / (float)RAND_MAX) * 2.f - 1.f; };
            auto gauss = [&] { float s = 0; for (int i = 0; i < 12; ++i) s += rand() / (float)RAND_MAX; return s - 6.f; };
            for (auto& v : A) v = (dist ? gauss() : rnd()) / sqrtf((float)K);
            for (auto& v : B) { v = dist ? gauss() : rnd(); if (dist) v = v / (1.f + expf(-v)); }   // SiLU-like activations
            std::vector<double> ref(1024, 0.0), mag(1024, 0.0);
            for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) {
                double s = 0, a = 0; for (int k = 0; k < K; ++k) { s += (double)A[m * K + k] * B[k * 32 + n]; a += fabs((double)A[m * K + k] * B[k * 32 + n]); }
                ref[m * 32 + n] = s; mag[m * 32 + n] = a;
            }
            float *dA, *dB, *dC; hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 4096);
            hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
            printf("K=%d %s:\n", K, dist ? "gauss weights x SiLU(gauss)" : "uniform x uniform");
            for (int mode : {0, 1, 3, 6, 9}) {
                hipLaunchKernelGGL(dot_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, mode);
                std::vector<float> C(1024); hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
                double mx = 0, rms = 0, mxrel = 0;
                for (int i = 0; i < 1024; ++i) { double e = fabs(C[i] - ref[i]); mx = fmax(mx, e); rms += e * e; mxrel = fmax(mxrel, e / mag[i]); }
                printf("   mode %d: max abs err %.3e  rms %.3e  max err/sum|ab| %.3e\n", mode, mx, sqrt(rms / 1024), mxrel);
            }
            // sign test of the accumulation rounding: rerun x6 with A negated; a round-to-nearest datapath gives err(-A) = -err(A)
            {
                std::vector<float> C1(1024), C2(1024), An(A); for (auto& v : An) v = -v;
                hipLaunchKernelGGL(dot_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, 6); hipMemcpy(C1.data(), dC, 4096, hipMemcpyDeviceToHost);
                hipMemcpy(dA, An.data(), An.size() * 4, hipMemcpyHostToDevice);
                hipLaunchKernelGGL(dot_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, 6); hipMemcpy(C2.data(), dC, 4096, hipMemcpyDeviceToHost);
                hipLaunchKernelGGL(dot_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, 0); std::vector<float> C3(1024); hipMemcpy(C3.data(), dC, 4096, hipMemcpyDeviceToHost);
                double m1 = 0, m2 = 0, m3 = 0; for (int i = 0; i < 1024; ++i) { m1 += C1[i] - ref[i]; m2 += C2[i] + ref[i]; m3 += C3[i] + ref[i]; }
                printf("   mean signed error: x6(A) %+.3e   x6(-A) %+.3e   fp32-mfma(-A) %+.3e\n", m1 / 1024, m2 / 1024, m3 / 1024);
            }
            // fp32 sequential fmaf chain on host for reference
            double mx = 0, rms = 0;
            for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) { float s = 0; for (int k = 0; k < K; ++k) s = fmaf(A[m * K + k], B[k * 32 + n], s); double e = fabs(s - ref[m * 32 + n]); mx = fmax(mx, e); rms += e * e; }
            printf("   host fmaf chain: max abs err %.3e  rms %.3e\n", mx, sqrt(rms / 1024));
        }
    }
    return 0;
}

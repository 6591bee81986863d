// K8 (core): multi-head self-attention softmax(Q K^T / sqrt(d)) V on the fp32 matrix cores.
//
// Reference: nn.MultiheadAttention(C, 8 heads, batch_first) inside SelfAttentionBlock
// (/root/reference/models/efficient_unet.py:23-53).  The in/out projections are 1x1 convolutions
// (conv_mfma.hip); this kernel is the flash-style core on the channel-major layout those produce:
//   qkv (B, 3C, N): rows [0,C) = Q, [C,2C) = K, [2C,3C) = V, head h owns rows h*D .. h*D+D-1
//   out (B, C, N)
// One wave owns 32 queries; a block (4 waves) shares each 32-key K/V tile through LDS.
//   S^T = K^T Q   (M = keys, N = queries, K = d): the MFMA result leaves every lane holding 16
//                  scores of ONE query (column = lane&31), so the online softmax is lane-local plus
//                  one cross-half shuffle, and P^T is already the B operand of the next product;
//   O^T = V P^T   (M = d, N = queries, K = keys): key pair (j, j+4) of MFMA r is exactly what the
//                  two half-waves hold in register r -- no data movement between the two GEMMs.
// fp32-in MFMA == fmaf chain, so this is plain fp32 attention numerically.
#include "common.h"

namespace r2dm {

template <int D>
__global__ __launch_bounds__(256) void attention_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                        int C, int N, float scale) {
    constexpr int KS = 32, VS = 33;  // LDS row strides: K rows read along keys, V rows along lanes
    __shared__ float Ks[D * KS];
    __shared__ float Vs[D * VS];

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = tid >> 6;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q0 = (blockIdx.x * 4 + wave) * 32;
    const bool active = q0 < N;

    const float* qp = qkv + ((long)b * 3 * C + (long)h * D) * N;
    const float* kp = qp + (long)C * N;
    const float* vp = kp + (long)C * N;

    float qf[D / 2];
#pragma unroll
    for (int p = 0; p < D / 2; ++p) qf[p] = active ? qp[(long)(2 * p + hi) * N + q0 + l31] : 0.f;

    f32x16 o[D / 32];
#pragma unroll
    for (int t = 0; t < D / 32; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    constexpr int PER = D / 8;  // rows of the K (and V) tile each thread stages: e = tid/32 + 8*i
    float kreg[PER], vreg[PER];
    const int se = tid >> 5, sj = tid & 31;
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            kreg[i] = kp[(long)(se + 8 * i) * N + kt * 32 + sj];
            vreg[i] = vp[(long)(se + 8 * i) * N + kt * 32 + sj];
        }
    };

    const int ntiles = N / 32;
    load_tile(0);
    for (int kt = 0; kt < ntiles; ++kt) {
        if (kt) __syncthreads();
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            Ks[(se + 8 * i) * KS + sj] = kreg[i];
            Vs[(se + 8 * i) * VS + sj] = vreg[i];
        }
        __syncthreads();
        if (kt + 1 < ntiles) load_tile(kt + 1);
        if (!active) continue;

        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int p = 0; p < D / 2; ++p)
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[(2 * p + hi) * KS + l31], qf[p], s, 0, 0, 0);

        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] *= scale;
            mx = fmaxf(mx, s[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __expf(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = __expf(s[r] - m_new);
            psum += s[r];
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int t = 0; t < D / 32; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[t][r] *= alpha;

#pragma unroll
        for (int t = 0; t < D / 32; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = (r & 3) + 8 * (r >> 2) + 4 * hi;  // key held in register r by this half-wave
                o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[(t * 32 + l31) * VS + j], s[r], o[t], 0, 0, 0);
            }
    }
    if (!active) return;
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    float* op = out + ((long)b * C + (long)h * D) * N + q0 + l31;
#pragma unroll
    for (int t = 0; t < D / 32; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int e = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            op[(long)e * N] = o[t][r] * inv;
        }
}

bool attention_supported(int C, int heads, int N) {
    if (heads <= 0 || C % heads) return false;
    const int d = C / heads;
    return (d == 32 || d == 64) && N % 32 == 0 && N > 0;
}

hipError_t launch_attention(const float* qkv, float* out, int B, int C, int heads, int N, hipStream_t s) {
    if (!attention_supported(C, heads, N)) return hipErrorInvalidValue;
    const int d = C / heads;
    const dim3 g((N / 32 + 3) / 4, heads, B);
    const float scale = 1.0f / sqrtf((float)d);
    if (d == 64) attention_kernel<64><<<g, 256, 0, s>>>(qkv, out, C, N, scale);
    else attention_kernel<32><<<g, 256, 0, s>>>(qkv, out, C, N, scale);
    return hipGetLastError();
}

}  // namespace r2dm

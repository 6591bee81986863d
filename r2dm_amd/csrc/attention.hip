// K8 (core): multi-head self-attention softmax(Q K^T / sqrt(d)) V on the fp32 matrix cores.
//
// Reference: nn.MultiheadAttention(C, 8 heads, batch_first) inside SelfAttentionBlock
// (/root/reference/models/efficient_unet.py:23-53).  The in/out projections are 1x1 convolutions
// (conv_mfma.hip); this kernel is the flash-style core on the channel-major layout those produce:
//   qkv (B, 3C, N): rows [0,C) = Q, [C,2C) = K, [2C,3C) = V, head h owns rows h*D .. h*D+D-1
//   out (B, C, N)
// One wave owns 32 queries; a block (4 waves; 8 in the fp16-pipe kernel) shares each 32-key K/V tile through LDS.
//   S^T = K^T Q   (M = keys, N = queries, K = d): the MFMA result leaves every lane holding 16
//                  scores of ONE query (column = lane&31), so the online softmax is lane-local plus
//                  one cross-half shuffle, and P^T is already the B operand of the next product;
//   O^T = V P^T   (M = d, N = queries, K = keys): key pair (j, j+4) of MFMA r is exactly what the
//                  two half-waves hold in register r -- no data movement between the two GEMMs.
// fp32-in MFMA == fmaf chain, so this is plain fp32 attention numerically.
#include "common.h"
#include "wave_ops.h"
#include "conv_bf16x3.h"
#include "f16x2.h"

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
        mx = fmaxf(mx, wave_xor32(mx));
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
    const float l_tot = l_run + wave_xor32(l_run);
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

// ---- the same attention on the fp16 matrix pipe, operands split exactly to 22 bits (f16x2.h) ---------------------------
// Round 2.  Per 32-key tile the fp32-MFMA kernel above issues D/2 + D/2 MFMAs of 64 cycles (4096 cycles at D = 64); here the
// two contractions are 3 x D/16 + 3 x 2 x D/32 = 24 v_mfma_f32_32x32x16_f16 of 32 cycles (768), each with the h*h products in
// one accumulator and the cross products in a second one.  Q (pre-multiplied by 1/sqrt(d)) is split once per wave; the K / V
// tiles are split by the staging threads on their way into LDS (double-buffered: one barrier per tile); P is split in
// registers -- its C-layout registers 8s..8s+7 ARE the B operand of k-step s when V's keys are stored in the matching order.
//   LDS  Kt[buf][plane][key 32][d D (+8 pad)] f16   A operand of S^T = K^T Q : lane (key, half) reads 8 consecutive d
//        Vs[buf][plane][d D][position 32 (+8 pad)]  A operand of O^T = V P^T : position s*16 + half*8 + j <-> key
//                                                    s*16 + (j/4)*8 + half*4 + j%4 (the key register r = 8s + j of that half holds)
// Range: q, k, v must fit fp16 (|v| < 65504): the engine has the producing qkv convolution record max|qkv| in the range flag
// (r2dm_check_range); with r2dm_set_conv_pieces(h, 3) the fp32-MFMA kernel above runs instead.
// NPLK = 1: the h planes alone -- one fp16 product per MAC (Q, K, V and P rounded to fp16, fp32 accumulation and softmax): the
// reduced-precision bulk mode (conv_f16x2.hip).
template <int D, int NPLK>
__global__ __launch_bounds__(512) void attention_f16x2_kernel(const float* __restrict__ qkv, float* __restrict__ out, int C, int N,
                                                              float scale) {
    constexpr int KROW = D * 2 + 16, VROW = 32 * 2 + 16;          // bytes per LDS row (16 bytes of padding: conflict-free b128 reads)
    constexpr int KPL = 32 * KROW, VPL = D * VROW;                 // bytes per plane
    constexpr int KBUF = 2 * KPL, VBUF = 2 * VPL;                  // bytes per buffer (planes h, l)
    constexpr int KS = D / 16, TB = D / 32;                        // k-steps of S^T, 32-row blocks of O^T
    __shared__ __attribute__((aligned(16))) unsigned char Kt[2 * KBUF];
    __shared__ __attribute__((aligned(16))) unsigned char Vs[2 * VBUF];
    f16_saturate_mode();

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = tid >> 6;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q0 = (blockIdx.x * 8 + wave) * 32;  // (eight waves = 256 queries share every staged K / V tile)
    const bool active = q0 < N;

    const float* qp = qkv + ((long)b * 3 * C + (long)h * D) * N;
    const float* kp = qp + (long)C * N;
    const float* vp = kp + (long)C * N;

    // Q: B operand of S^T, lane (query l31, half hi) holds d = 16 s + 8 hi + 0..7 of k-step s; scaled, split once
    // (Round 6, read in the ISA: as `active ? qp[...] * scale : 0` every one of these D / 2 loads sat in its own exec-masked branch with an s_waitcnt vmcnt(0) behind
    // it -- 32 SERIAL round trips before the first key tile.  Now all of them are in flight at once, from an address that is valid in every wave.)
    u32x4 qh[KS], ql[KS];
    {
        const float* qcol = qp + (active ? q0 : 0) + l31;
        float qa[KS][4], qc[KS][4];
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                qa[s][j] = qcol[(long)(s * 16 + hi * 8 + 2 * j) * N];
                qc[s][j] = qcol[(long)(s * 16 + hi * 8 + 2 * j + 1) * N];
            }
        const float qs = active ? scale : 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsigned uh, ul;
                split_f16x2(qa[s][j] * qs, qc[s][j] * qs, uh, ul);
                qh[s][j] = uh;
                ql[s][j] = ul;
            }
    }

    f32x16 o[TB], ol[TB];
#pragma unroll
    for (int t = 0; t < TB; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = ol[t][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // staging roles: threads 0..255 move K -- thread (key = t & 31, d block = t >> 5) 8 d of one key --, threads 256..511 move V --
    // thread (d = t >> 2, key octet = t & 3) 8 consecutive keys of one d.  Threads beyond D / 8 d blocks (D = 32: half of them) idle.
    const int t8 = tid & 255;
    const int k_key = t8 & 31, k_db = t8 >> 5, v_d = t8 >> 2, v_kq = t8 & 3;
    const bool k_on = tid < 256 && k_db < D / 8, v_on = tid >= 256 && v_d < D;
    float kreg[8];
    f32x4 vreg[2];
    auto load_tile = [&](int kt) {
        if (k_on) {
#pragma unroll
            for (int j = 0; j < 8; ++j) kreg[j] = kp[(long)(k_db * 8 + j) * N + kt * 32 + k_key];
        }
        if (v_on) {
            const f32x4* src = reinterpret_cast<const f32x4*>(vp + (long)v_d * N + kt * 32 + v_kq * 8);
            vreg[0] = src[0];
            vreg[1] = src[1];
        }
    };
    auto store_tile = [&](int buf) {
        if (k_on) {
            unsigned ph[4], pl[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) split_f16x2(kreg[2 * j], kreg[2 * j + 1], ph[j], pl[j]);
            unsigned char* dst = Kt + buf * KBUF + k_key * KROW + k_db * 16;
            *reinterpret_cast<u32x4*>(dst) = u32x4{ph[0], ph[1], ph[2], ph[3]};
            if (NPLK == 2) *reinterpret_cast<u32x4*>(dst + KPL) = u32x4{pl[0], pl[1], pl[2], pl[3]};
        }
        if (v_on) {
            // keys v_kq*8 + t: t = 0..3 -> half 0, t = 4..7 -> half 1; position = 16 s + 8 half + 4 (v_kq & 1) + t % 4, s = v_kq >> 1
            unsigned char* dst = Vs + buf * VBUF + v_d * VROW + ((v_kq >> 1) * 16 + (v_kq & 1) * 4) * 2;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                unsigned ph0, pl0, ph1, pl1;
                split_f16x2(vreg[hh][0], vreg[hh][1], ph0, pl0);
                split_f16x2(vreg[hh][2], vreg[hh][3], ph1, pl1);
                using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
                *reinterpret_cast<u32x2*>(dst + hh * 16) = u32x2{ph0, ph1};
                if (NPLK == 2) *reinterpret_cast<u32x2*>(dst + hh * 16 + VPL) = u32x2{pl0, pl1};
            }
        }
    };

    // S^T of a key tile into (s, sl): issued BEHIND the barrier that publishes the tile, i.e. at the end of the previous iteration --
    // its 3 KS MFMAs run while this wave stages the next tile (split + LDS writes: independent vector work), and the softmax
    // that needs them comes after that.  (Round 2's order, stage | S^T | softmax | O^T | barrier, exposed the matrix-pipe and LDS
    // latencies of both contractions in every iteration: ablation timings in profiles/r03_launch_overhead.txt.)
    f32x16 s, sl;
    auto scores = [&](int buf) __attribute__((always_inline)) {
        const unsigned char* kb = Kt + buf * KBUF + l31 * KROW + hi * 16;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = sl[r] = 0.f;
#pragma unroll
        for (int st = 0; st < KS; ++st) {
            const f16x8 kh = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(kb + st * 32));
            const f16x8 qhh = __builtin_bit_cast(f16x8, qh[st]), qll = __builtin_bit_cast(f16x8, ql[st]);
            if (NPLK == 2) {
                const f16x8 kl = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(kb + st * 32 + KPL));
                sl = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qhh, sl, 0, 0, 0);
                sl = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qll, sl, 0, 0, 0);
            }
            s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qhh, s, 0, 0, 0);
        }
    };
    const int ntiles = N / 32;
    load_tile(0);
    store_tile(0);
    if (ntiles > 1) load_tile(1);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = sl[r] = 0.f;
    if (active) scores(0);
    for (int kt = 0; kt < ntiles; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < ntiles) {
            store_tile(buf ^ 1);  // (buffer of tile kt-1: everybody left it at the barrier in the middle of the previous iteration)
            if (kt + 2 < ntiles) load_tile(kt + 2);
        }
        if (active) {
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (NPLK == 2) s[r] = fmaf(sl[r], f2::LINV, s[r]);
                mx = fmaxf(mx, s[r]);
            }
            mx = fmaxf(mx, wave_xor32(mx));
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
            // (the running maximum settles after a few tiles: alpha == 1 in every lane from then on, and multiplying by exactly 1 is
            // the identity -- the 2 x 16 x TB multiplications are skipped wave-uniformly: a seventh of the loop's vector instructions)
            if (__any(alpha != 1.0f)) {
#pragma unroll
                for (int t = 0; t < TB; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        o[t][r] *= alpha;
                        ol[t][r] *= alpha;
                    }
            }
            // P^T: registers 8 st .. 8 st + 7 are this lane's eight keys of k-step st (see the V layout above)
            const unsigned char* vb = Vs + buf * VBUF + l31 * VROW + hi * 16;
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                unsigned ph[4], pl[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) split_f16x2(s[8 * st + 2 * j], s[8 * st + 2 * j + 1], ph[j], pl[j]);
                const f16x8 phh = __builtin_bit_cast(f16x8, u32x4{ph[0], ph[1], ph[2], ph[3]});
                const f16x8 pll = __builtin_bit_cast(f16x8, u32x4{pl[0], pl[1], pl[2], pl[3]});
#pragma unroll
                for (int t = 0; t < TB; ++t) {
                    const f16x8 vh = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(vb + t * 32 * VROW + st * 32));
                    if (NPLK == 2) {
                        const f16x8 vl = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(vb + t * 32 * VROW + st * 32 + VPL));
                        ol[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, phh, ol[t], 0, 0, 0);
                        ol[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pll, ol[t], 0, 0, 0);
                    }
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, phh, o[t], 0, 0, 0);
                }
            }
        }
        __syncthreads();  // tile kt + 1 is in LDS; everybody is done with tile kt
        if (active && kt + 1 < ntiles) scores(buf ^ 1);
    }
    if (!active) return;
    const float l_tot = l_run + wave_xor32(l_run);
    const float inv = 1.0f / l_tot;
    float* op = out + ((long)b * C + (long)h * D) * N + q0 + l31;
#pragma unroll
    for (int t = 0; t < TB; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int e = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            op[(long)e * N] = (NPLK == 2 ? fmaf(ol[t][r], f2::LINV, o[t][r]) : o[t][r]) * inv;
        }
}

// ---- any head size up to 128, any token count: plain fp32 on the vector ALU ------------------------------------------------
// The matrix-pipe kernels above cover what the LiDAR checkpoints use (head size 32 / 64, tokens a multiple of 32).  Other
// geometries the reference accepts (base_channels = 96 -> head size 96; a 16 x 64 image -> 16 tokens, ...) used to be refused;
// they are rare and small, so they get a simple kernel: one thread per query, the keys in order, online softmax, every K / V
// element a wave-uniform (broadcast) load.  fp32 FMA chains = the reference's math path.
template <int DMAX>
__global__ __launch_bounds__(64) void attention_generic_kernel(const float* __restrict__ qkv, float* __restrict__ out, int C, int D, int N,
                                                               float scale) {
    const int n = blockIdx.x * 64 + threadIdx.x, h = blockIdx.y, b = blockIdx.z;
    const bool active = n < N;
    const float* qp = qkv + ((long)b * 3 * C + (long)h * D) * N;
    const float* kp = qp + (long)C * N;
    const float* vp = kp + (long)C * N;
    float q[DMAX], o[DMAX];
#pragma unroll
    for (int e = 0; e < DMAX; ++e) {
        q[e] = (e < D && active) ? qp[(long)e * N + n] * scale : 0.f;
        o[e] = 0.f;
    }
    float m = -INFINITY, l = 0.f;
    for (int j = 0; j < N; ++j) {
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < DMAX; ++e)
            if (e < D) s = fmaf(q[e], kp[(long)e * N + j], s);
        const float m_new = fmaxf(m, s);
        const float a = __expf(m - m_new), p = __expf(s - m_new);
        l = l * a + p;
        m = m_new;
#pragma unroll
        for (int e = 0; e < DMAX; ++e)
            if (e < D) o[e] = fmaf(p, vp[(long)e * N + j], o[e] * a);
    }
    if (!active) return;
    const float inv = 1.0f / l;
    float* op = out + ((long)b * C + (long)h * D) * N + n;
#pragma unroll
    for (int e = 0; e < DMAX; ++e)
        if (e < D) op[(long)e * N] = o[e] * inv;
}

static bool attention_fast_path(int d, int N) { return (d == 32 || d == 64) && N % 32 == 0; }

bool attention_supported(int C, int heads, int N) {
    if (heads <= 0 || C % heads || N <= 0) return false;
    const int d = C / heads;
    return attention_fast_path(d, N) || d <= 128;
}

// planes: 0 = the fp32-MFMA kernel; 2 = the fp16-matrix-pipe kernel with split operands (q, k, v must fit the fp16 range: the
// caller guards it); 1 = the same kernel with one fp16 product per MAC (reduced-precision bulk mode)
hipError_t launch_attention(const float* qkv, float* out, int B, int C, int heads, int N, hipStream_t s, int planes) {
    if (!attention_supported(C, heads, N)) return hipErrorInvalidValue;
    const int d = C / heads;
    const float scale = 1.0f / sqrtf((float)d);
    if (!attention_fast_path(d, N)) {  // (fp32 whatever the mode: at least as accurate as any of them)
        const dim3 gg((N + 63) / 64, heads, B);
        if (d <= 32) attention_generic_kernel<32><<<gg, 64, 0, s>>>(qkv, out, C, d, N, scale);
        else if (d <= 64) attention_generic_kernel<64><<<gg, 64, 0, s>>>(qkv, out, C, d, N, scale);
        else attention_generic_kernel<128><<<gg, 64, 0, s>>>(qkv, out, C, d, N, scale);
        return hipGetLastError();
    }
    const dim3 g((N / 32 + 3) / 4, heads, B), g8((N / 32 + 7) / 8, heads, B);
    if (planes == 2) {
        if (d == 64) attention_f16x2_kernel<64, 2><<<g8, 512, 0, s>>>(qkv, out, C, N, scale);
        else attention_f16x2_kernel<32, 2><<<g8, 512, 0, s>>>(qkv, out, C, N, scale);
    } else if (planes == 1) {
        if (d == 64) attention_f16x2_kernel<64, 1><<<g8, 512, 0, s>>>(qkv, out, C, N, scale);
        else attention_f16x2_kernel<32, 1><<<g8, 512, 0, s>>>(qkv, out, C, N, scale);
    } else {
        if (d == 64) attention_kernel<64><<<g, 256, 0, s>>>(qkv, out, C, N, scale);
        else attention_kernel<32><<<g, 256, 0, s>>>(qkv, out, C, N, scale);
    }
    return hipGetLastError();
}

}  // namespace r2dm

// K4 / K5: GroupNorm statistics and the folded per-(sample, channel) affine.
//
// Reference: nn.GroupNorm(8, C, eps=1e-6) (/root/reference/models/efficient_unet.py:33,72) and AdaGN
// (/root/reference/models/ops.py:176-200).  Both are folded to  y = x * a + d  with
//     GroupNorm : a = rstd * gamma[c]            d = beta[c]     - mean * a
//     AdaGN     : a = rstd * (1 + scale[b,c])    d = shift[b,c]  - mean * a
// (the same scale/bias form ATen's GroupNorm kernels apply).  The apply itself is fused into the
// consuming convolution's load stage (conv_mfma.hip); gn_apply_kernel below exists for unit tests
// and for callers that need the normalised tensor.
//
// Statistics: HBM-bound single pass over the tensor, 16-byte loads, fp64 sum / sum-of-squares
// (fp64 VALU is not a bottleneck on a streaming kernel and makes E[x^2]-E[x]^2 safe for groups of
// up to 2^22 elements), wavefront shuffle reduce, fixed-order cross-block combine => deterministic.
#include "common.h"
#include "gn_math.h"
#include "wave_ops.h"

namespace r2dm {

constexpr int kGnThreads = 256;

__device__ __forceinline__ double wave_sum(double v) { return wave_sum_f64(v); }  // (wave_ops.h)

// grid = (splits, G, B).  Group g of sample b = cpg consecutive planes (all inside one of the two
// sources), i.e. n = cpg*HW contiguous floats; split s reduces elements [s*len, (s+1)*len).
__global__ __launch_bounds__(kGnThreads) void gn_partial_kernel(Src x, int cpg, long hw, int splits,
                                                                double* __restrict__ partial, float* __restrict__ partial_max) {
    const int s = blockIdx.x, g = blockIdx.y, b = blockIdx.z, G = gridDim.y;
    const long n = (long)cpg * hw;
    const float* base = x.plane(b, g * cpg, hw);
    long len = (n + splits - 1) / splits;
    len = (len + 3) & ~3L;
    const long lo = s * len, hi = lo + len < n ? lo + len : n;
    double sum = 0.0, sq = 0.0;
    float amax = 0.f;  // largest |x| of the split (the consumer's observed range bound)
    const bool vec = ((reinterpret_cast<uintptr_t>(base) & 15) == 0) && ((n & 3) == 0);
    if (vec) {
        const f32x4* p4 = reinterpret_cast<const f32x4*>(base);
        for (long i = lo / 4 + threadIdx.x; i < hi / 4; i += kGnThreads) {
            const f32x4 v = p4[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double d = (double)v[j];
                sum += d;
                sq = fma(d, d, sq);
            }
            amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        }
    } else {
        for (long i = lo + threadIdx.x; i < hi; i += kGnThreads) {
            const float v = base[i];
            sum += (double)v;
            sq += (double)v * (double)v;
            amax = fmaxf(amax, fabsf(v));
        }
    }
    __shared__ double red[2][kGnThreads / 64];
    __shared__ float redm[kGnThreads / 64];
    sum = wave_sum(sum);
    sq = wave_sum(sq);
    amax = wave_max_f32(amax);
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = sum;
        red[1][threadIdx.x >> 6] = sq;
        redm[threadIdx.x >> 6] = amax;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, q = 0.0;
        float m = 0.f;
        for (int w = 0; w < kGnThreads / 64; ++w) {
            a += red[0][w];
            q += red[1][w];
            m = fmaxf(m, redm[w]);
        }
        double* o = partial + (((long)b * G + g) * splits + s) * 2;
        o[0] = a;
        o[1] = q;
        if (partial_max) partial_max[((long)b * G + g) * splits + s] = m;
    }
}

// grid = (G, B), 256 threads: combine the splits in a fixed order (thread -> slots, lanes -> wave, waves in index order),
// then emit (a, d) for the group's channels.
__global__ __launch_bounds__(256) void gn_finalize_kernel(const double* __restrict__ partial, int splits, int C,
                                                         int cpg, long hw, float eps,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta,
                                                         const float* __restrict__ ada, long ada_stride,
                                                         float2* __restrict__ aff, float* __restrict__ stats,
                                                         int* __restrict__ range_flag, const float* __restrict__ partial_max) {
    const int g = blockIdx.x, b = blockIdx.y, G = gridDim.x;
    const double* p = partial + ((long)b * G + g) * splits * 2;
    // the largest |x| of the group as its producers observed it (range guard of the fp16 consumers; any order: a maximum)
    float gmax = 0.f;
    if (range_flag && partial_max) {
        const float* pm = partial_max + ((long)b * G + g) * splits;
        for (int s = threadIdx.x; s < splits; s += 256) gmax = fmaxf(gmax, pm[s]);
        gmax = wave_max_f32(gmax);
    }
    // fixed thread -> slot assignment: deterministic.  All of a thread's slots are requested before the first is added (16-byte
    // loads, up to 8 in flight): as a plain loop the <= 8 dependent round trips to L2 were most of this 5 us kernel
    using d2 = __attribute__((ext_vector_type(2))) double;
    const d2* p2 = reinterpret_cast<const d2*>(p);
    double sum = 0.0, sq = 0.0, emax = 0.0;  // emax: the largest slot energy (sum of squares of one slot's elements)
    for (int s0 = threadIdx.x; s0 < splits; s0 += 256 * 8) {
        d2 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int s = s0 + 256 * i;
            v[i] = s < splits ? p2[s] : d2{0.0, 0.0};
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            sum += v[i][0];
            sq += v[i][1];
            emax = v[i][1] > emax ? v[i][1] : emax;
        }
    }
    // No recorded maximum: every element of a slot is bounded by the slot's energy, |x| <= sqrt(sum of squares over the slot)
    // -- a rigorous, data-driven bound that costs the producers nothing (22 ... 45 x the typical maximum for slots of
    // 512 ... 2048 elements, against sqrt(n) = 724 of the worst-case bound over the whole group)
    if (range_flag && !partial_max) gmax = wave_max_f32((float)sqrt(emax) * 1.000001f);
    __shared__ double red[2][4];
    __shared__ float redm[4];
    // (the reduction order from here on is a contract with the consumer-side fold in conv_f16x2.hip: per-thread partials over slots
    // t, t + 256, ... ascending; wave_sum_f64_hi_first; the four waves as (w0 + w1) + (w2 + w3))
    sum = wave_sum_f64_hi_first(sum);
    sq = wave_sum_f64_hi_first(sq);
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = sum;
        red[1][threadIdx.x >> 6] = sq;
        redm[threadIdx.x >> 6] = gmax;
    }
    __syncthreads();
    sum = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    sq = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    gmax = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));
    // (the arithmetic from here on is gn_math.h's: conv_f16x2.hip's consumer-side fold repeats it bit for bit)
    const double n = (double)cpg * (double)hw;
    const GnMoments mo = gn_moments(sum, sq, n, eps);
    const float mean = mo.mean, rstd = mo.rstd;
    if (stats && threadIdx.x == 0) {
        stats[((long)b * G + g) * 2 + 0] = mean;
        stats[((long)b * G + g) * 2 + 1] = rstd;
    }
    for (int i = threadIdx.x; i < cpg; i += 256) {
        const int c = g * cpg + i;
        float w, sh;
        if (ada) {
            w = 1.0f + ada[b * ada_stride + c];
            sh = ada[b * ada_stride + C + c];
        } else {
            w = gamma ? gamma[c] : 1.0f;
            sh = beta ? beta[c] : 0.0f;
        }
        const float2 ad = gn_affine(mo, w, sh);
        aff[(long)b * C + c] = ad;
        // Range guard of the fp16 consumers (gn_math.h: gn_bound): recorded as a running maximum -- positive floats order like their
        // bit patterns, NaN above all -- and compared with the fp16 limit by r2dm_check_range.  SiLU only shrinks the bound.
        if (range_flag) atomicMax(range_flag + 1, __float_as_int(gn_bound(mo, ad, w, sh, gmax, n)));
    }
}

int gn_splits(int B, int groups, long group_elems) {
    // aim at >= ~2048 blocks chip-wide but keep >= 16 KiB per block
    long want = 2048 / ((long)B * groups > 0 ? (long)B * groups : 1);
    long cap = group_elems / 4096;
    long s = want < cap ? want : cap;
    if (s < 1) s = 1;
    if (s > 256) s = 256;
    return (int)s;
}

hipError_t launch_group_norm(const GNParams& p, hipStream_t st) {
    const int C = p.x.channels();
    if (C % p.groups) return hipErrorInvalidValue;
    const int cpg = C / p.groups;
    if (p.x.c1 > 0 && (p.x.c0 % cpg)) return hipErrorInvalidValue;  // a group may not straddle the concat seam
    const long hw = (long)p.H * p.W;
    const int splits = gn_splits(p.B, p.groups, cpg * hw);
    gn_partial_kernel<<<dim3(splits, p.groups, p.B), kGnThreads, 0, st>>>(p.x, cpg, hw, splits, p.partial, p.partial_max);
    gn_finalize_kernel<<<dim3(p.groups, p.B), 256, 0, st>>>(p.partial, splits, C, cpg, hw, p.eps, p.gamma, p.beta,
                                                            p.ada, p.ada_stride, p.aff, p.stats, p.range_flag, p.partial_max);
    return hipGetLastError();
}

hipError_t launch_group_norm_finalize(const GNParams& p, int C, int splits, hipStream_t st) {
    if (C % p.groups) return hipErrorInvalidValue;
    gn_finalize_kernel<<<dim3(p.groups, p.B), 256, 0, st>>>(p.partial, splits, C, C / p.groups, (long)p.H * p.W, p.eps, p.gamma,
                                                            p.beta, p.ada, p.ada_stride, p.aff, p.stats, p.range_flag, p.partial_max);
    return hipGetLastError();
}

// y = f(x*a+d), one (b,c) plane per blockIdx.y -- test / utility path only
__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, const float2* __restrict__ aff,
                                                       float* __restrict__ y, long hw, int silu) {
    const long plane = blockIdx.y;
    const float2 ad = aff[plane];
    const float* xp = x + plane * hw;
    float* yp = y + plane * hw;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < hw; i += (long)gridDim.x * 256) {
        float v = xp[i] * ad.x + ad.y;
        yp[i] = silu ? silu_f(v) : v;
    }
}

hipError_t launch_gn_apply(const float* x, const float2* aff, float* y, int B, int C, long hw, int silu,
                           hipStream_t st) {
    int bx = (int)((hw + 255) / 256);
    if (bx > 64) bx = 64;
    gn_apply_kernel<<<dim3(bx, B * C), 256, 0, st>>>(x, aff, y, hw, silu);
    return hipGetLastError();
}

}  // namespace r2dm

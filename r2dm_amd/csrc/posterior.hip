// K10: fused posterior update  x_t, prediction, noise -> x_s  (one HBM pass: 3 reads + 1 write).
//
// Replays, per element and in the reference's float32 operation order (no FMA contraction, IEEE
// division), the tail of p_step:
//   continuous time  /root/reference/models/diffusion/continuous_time.py:208-229
//   discrete time    /root/reference/models/diffusion/discrete_time.py:140-177
// All schedule scalars arrive in coef[b][8], computed on the host with the reference's torch ops.
//
// also: K12 LiDAR post-processing of finished samples (reference sample_and_save.py:52-57).
#include "common.h"

namespace r2dm {

enum { M_CT_DDPM = 0, M_CT_DDIM = 1, M_DT_DDPM = 2, M_DT_DDIM = 3, M_DT_DDIM_NOISE = 4 };
enum { OBJ_EPS = 0, OBJ_V = 1, OBJ_X0 = 2 };

#pragma clang fp contract(off)
template <int MODE, int OBJ>
__device__ __forceinline__ float posterior_elem(float x, float pr, float z, const float* k, float clip) {
    float x0;
    if (MODE == M_CT_DDPM || MODE == M_CT_DDIM) {
        const float a_t = k[0], s_t = k[1];
        if (OBJ == OBJ_EPS) x0 = (x - s_t * pr) / a_t;
        else if (OBJ == OBJ_V) x0 = a_t * x - s_t * pr;
        else x0 = pr;
    } else {
        if (OBJ == OBJ_X0) x0 = pr;
        else x0 = k[0] * x - k[1] * pr;
    }
    if (clip >= 0.f) x0 = fminf(fmaxf(x0, -clip), clip);
    if (MODE == M_CT_DDPM) {
        const float a_t = k[0], a_s = k[2], c = k[4], sd = k[5];
        const float mean = a_s * (x * (1.0f - c) / a_t + c * x0);
        return mean + sd * z;
    } else if (MODE == M_CT_DDIM) {
        const float a_t = k[0], s_t = k[1], a_s = k[2], c1 = k[6], c2 = k[7];
        const float eps = (x - a_t * x0) / s_t;
        return a_s * x0 + c1 * z + c2 * eps;
    } else if (MODE == M_DT_DDPM) {
        const float mean = k[2] * x0 + k[3] * x;
        return mean + k[4] * z;
    } else {
        const float eps = (x - k[2] * x0) / k[3];
        float xs = k[5] * x0 + k[4] * eps;
        if (MODE == M_DT_DDIM_NOISE) xs = xs + k[6] * z;
        return xs;
    }
}

template <int MODE, int OBJ>
__global__ __launch_bounds__(256) void posterior_kernel(PosteriorParams p) {
    const int b = blockIdx.y;
    float k[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) k[i] = p.coef[b * 8 + i];
    const long n4 = p.per_sample >> 2;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(p.x_t + b * p.per_sample);
    const f32x4* p4 = reinterpret_cast<const f32x4*>(p.pred + b * p.per_sample);
    const f32x4* z4 = p.noise ? reinterpret_cast<const f32x4*>(p.noise + b * p.per_sample) : nullptr;
    f32x4* o4 = reinterpret_cast<f32x4*>(p.x_s + b * p.per_sample);
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 x = x4[i], pr = p4[i];
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        if (z4) z = z4[i];
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = posterior_elem<MODE, OBJ>(x[j], pr[j], z[j], k, p.clip);
        o4[i] = o;
    }
    // tail (per_sample not a multiple of 4)
    for (long i = (n4 << 2) + blockIdx.x * 256L + threadIdx.x; i < p.per_sample; i += (long)gridDim.x * 256) {
        const long e = b * p.per_sample + i;
        p.x_s[e] = posterior_elem<MODE, OBJ>(p.x_t[e], p.pred[e], p.noise ? p.noise[e] : 0.f, k, p.clip);
    }
}

template <int MODE>
static hipError_t launch_obj(const PosteriorParams& p, dim3 g, hipStream_t s) {
    switch (p.objective) {
        case OBJ_EPS: posterior_kernel<MODE, OBJ_EPS><<<g, 256, 0, s>>>(p); break;
        case OBJ_V: posterior_kernel<MODE, OBJ_V><<<g, 256, 0, s>>>(p); break;
        case OBJ_X0: posterior_kernel<MODE, OBJ_X0><<<g, 256, 0, s>>>(p); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_posterior(const PosteriorParams& p, hipStream_t s) {
    if ((p.per_sample & 3) || (reinterpret_cast<uintptr_t>(p.x_t) & 15) || (reinterpret_cast<uintptr_t>(p.pred) & 15) ||
        (reinterpret_cast<uintptr_t>(p.x_s) & 15) || (reinterpret_cast<uintptr_t>(p.noise) & 15))
        return hipErrorInvalidValue;
    if (p.noise == nullptr && p.mode != M_DT_DDIM) return hipErrorInvalidValue;
    long bx = (p.per_sample / 4 + 255) / 256;
    if (bx < 1) bx = 1;
    if (bx > 1024) bx = 1024;
    const dim3 g((unsigned)bx, p.B);
    switch (p.mode) {
        case M_CT_DDPM: return launch_obj<M_CT_DDPM>(p, g, s);
        case M_CT_DDIM: return launch_obj<M_CT_DDIM>(p, g, s);
        case M_DT_DDPM: return launch_obj<M_DT_DDPM>(p, g, s);
        case M_DT_DDIM: return launch_obj<M_DT_DDIM>(p, g, s);
        case M_DT_DDIM_NOISE: return launch_obj<M_DT_DDIM_NOISE>(p, g, s);
    }
    return hipErrorInvalidValue;
}

// ---- LiDAR post-processing ("next" row f.1) ------------------------------------------------
// sample (B,2,H,W) in [-1,1] -> (B,5,H,W): metric depth (decoded by the checkpoint's depth format, masked to
// (min_depth, max_depth)), Cartesian x,y,z along the per-pixel ray angles, reflectance in [0,1].
// /root/reference/utils/lidar.py:49-61,95-120 ; /root/reference/sample_and_save.py:52-57.
// FMT: 0 log_depth  metric = exp2(d log2(max + 1)) - 1 | 1 inverse_depth  metric = (1 / (d + 1e-8)) min | 2 depth  metric = d max
template <int FMT>
__global__ __launch_bounds__(256) void lidar_post_kernel(const float* __restrict__ x, const float* __restrict__ ang,
                                                         float* __restrict__ y, long hw, float min_d, float max_d,
                                                         float log2_range) {
    const int b = blockIdx.y;
    const float* xb = x + (long)b * 2 * hw;
    float* yb = y + (long)b * 5 * hw;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < hw; i += (long)gridDim.x * 256) {
        const float d = (xb[i] + 1.0f) / 2.0f, r = (xb[hw + i] + 1.0f) / 2.0f;
        float metric;
        if (FMT == 0) metric = exp2f(d * log2_range) - 1.0f;
        else if (FMT == 1) metric = __fmul_rn(__frcp_rn(d + 1e-8f), min_d);  // (torch evaluates `scalar / tensor` as tensor.reciprocal() * scalar: two roundings)
        else metric = d * max_d;
        const float m = (metric > min_d && metric < max_d) ? 1.0f : 0.0f;
        metric = metric * m;
        const float m2 = (metric > min_d && metric < max_d) ? 1.0f : 0.0f;
        const float phi = ang[i], theta = ang[hw + i];
        const float cp = cosf(phi);
        yb[i] = metric;
        yb[hw + i] = metric * cp * cosf(theta) * m2;
        yb[2 * hw + i] = metric * cp * sinf(theta) * m2;
        yb[3 * hw + i] = metric * sinf(phi) * m2;
        yb[4 * hw + i] = r;
    }
}

hipError_t launch_lidar_postprocess(const float* x, const float* angles, float* y, int B, int H, int W, float min_d,
                                    float max_d, hipStream_t s, int depth_format) {
    const long hw = (long)H * W;
    long bx = (hw + 255) / 256;
    if (bx > 1024) bx = 1024;
    const dim3 grid((unsigned)bx, B);
    const float l2 = (float)log2((double)max_d + 1.0);
    switch (depth_format) {
        case 0: lidar_post_kernel<0><<<grid, 256, 0, s>>>(x, angles, y, hw, min_d, max_d, l2); break;
        case 1: lidar_post_kernel<1><<<grid, 256, 0, s>>>(x, angles, y, hw, min_d, max_d, l2); break;
        case 2: lidar_post_kernel<2><<<grid, 256, 0, s>>>(x, angles, y, hw, min_d, max_d, l2); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}


// ---- RePaint pieces (reference continuous_time.py:169-190, 287-303), replayed in the reference's op order -------
// blend:  out = mask * (known * alpha + noise * sigma) + (1 - mask) * unknown      (q_step_from_x_0 + mask blend)
// q_step: out = x_s * a_ts + std * noise                                           (forward diffusion s -> t)
// coef[b] = (alpha, sigma) resp. (a_ts, std): host-computed scalars.  mask has mask_c channels (1 = broadcast).
#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void repaint_blend_kernel(const float* __restrict__ known, const float* __restrict__ noise,
                                                            const float* __restrict__ unknown, const float* __restrict__ mask,
                                                            const float* __restrict__ coef, float* __restrict__ out,
                                                            long per_sample, long plane, int mask_c) {
    const int b = blockIdx.y;
    const float alpha = coef[2 * b], sigma = coef[2 * b + 1];
    const long base = b * per_sample;
    for (long i = (blockIdx.x * 256L + threadIdx.x) * 4; i < per_sample; i += (long)gridDim.x * 1024) {
        const f32x4 k = *reinterpret_cast<const f32x4*>(known + base + i), z = *reinterpret_cast<const f32x4*>(noise + base + i);
        const f32x4 u = *reinterpret_cast<const f32x4*>(unknown + base + i);
        const long mi = mask_c == 1 ? b * plane + i % plane : base + i;  // plane % 4 == 0: a quad never straddles planes
        const f32x4 m = *reinterpret_cast<const f32x4*>(mask + mi);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float ks = k[e] * alpha + z[e] * sigma;
            o[e] = m[e] * ks + (1.0f - m[e]) * u[e];
        }
        *reinterpret_cast<f32x4*>(out + base + i) = o;
    }
}

__global__ __launch_bounds__(256) void q_step_kernel(const float* __restrict__ x, const float* __restrict__ noise,
                                                     const float* __restrict__ coef, float* __restrict__ out, long per_sample) {
    const int b = blockIdx.y;
    const float a = coef[2 * b], sd = coef[2 * b + 1];
    const long base = b * per_sample;
    for (long i = (blockIdx.x * 256L + threadIdx.x) * 4; i < per_sample; i += (long)gridDim.x * 1024) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + base + i), z = *reinterpret_cast<const f32x4*>(noise + base + i);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = v[e] * a + sd * z[e];
        *reinterpret_cast<f32x4*>(out + base + i) = o;
    }
}
#pragma clang fp contract(fast)

hipError_t launch_repaint_blend(const float* known, const float* noise, const float* unknown, const float* mask,
                                const float* coef, float* out, int B, long per_sample, int channels, int mask_c,
                                hipStream_t s) {
    if (per_sample % 4 || channels <= 0 || per_sample % channels || (per_sample / channels) % 4) return hipErrorInvalidValue;
    if (mask_c != 1 && mask_c != channels) return hipErrorInvalidValue;
    long bx = (per_sample / 4 + 255) / 256;
    if (bx > 256) bx = 256;
    repaint_blend_kernel<<<dim3((unsigned)bx, B), 256, 0, s>>>(known, noise, unknown, mask, coef, out, per_sample,
                                                               per_sample / channels, mask_c);
    return hipGetLastError();
}

hipError_t launch_q_step(const float* x, const float* noise, const float* coef, float* out, int B, long per_sample,
                         hipStream_t s) {
    if (per_sample % 4) return hipErrorInvalidValue;
    long bx = (per_sample / 4 + 255) / 256;
    if (bx > 256) bx = 256;
    q_step_kernel<<<dim3((unsigned)bx, B), 256, 0, s>>>(x, noise, coef, out, per_sample);
    return hipGetLastError();
}

}  // namespace r2dm

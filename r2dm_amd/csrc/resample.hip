// K6 / K7: the [1,3,3,1] FIR resamplers of the U-Net (HBM-bound, one pass, 16-byte accesses).
//
// Reference: ops.Resample (/root/reference/models/ops.py:52-146) as used at
// efficient_unet.py:135 (down=2 after the stage's 3x3 conv) and :171 (up=2 before it).
//   down: y[i,j] = sum_a k_a ( sum_b k_b xp[2i+a-1, 2j+b-1] ),  k = [1,3,3,1]/8
//   up  : per axis  y[2i] = x[i-1]/4 + 3x[i]/4 ,  y[2i+1] = 3x[i]/4 + x[i+1]/4   (zero-insert + [1,3,3,1]/4)
// xp / x[-1] / x[H]: columns wrap (azimuth is periodic), rows outside the image are zero.
// W pass first, then H, like the reference's two depthwise convolutions (ops.py:131-132).
// The FIR window is fixed: the host side refuses checkpoints whose `kernel` buffers differ.
#include "common.h"
#include <stdlib.h>
#include "wave_ops.h"

namespace r2dm {

// The [1,3,3,1]/8 window as ONE spelled-out chain of fused multiply-adds, shared by the three down-sampling kernels: left to the
// compiler's contraction the same source expression came out differently in different kernels (round 5: the statistics variant and the
// plain wide kernel differed in the last bit of some outputs, 2e-7 rms on the U-Net).
__device__ __forceinline__ float fir4(float a, float b, float c, float d) {
    return __builtin_fmaf(0.125f, d, __builtin_fmaf(0.375f, c, __builtin_fmaf(0.375f, b, 0.125f * a)));
}

// one thread -> two horizontally adjacent outputs; needs input columns 4t-1 .. 4t+4 of 4 rows
__global__ __launch_bounds__(256) void fir_down2_kernel(const float* __restrict__ x, long xbs, float* __restrict__ y,
                                                        long ybs, int C, int H, int W) {
    const int Ho = H >> 1, Wo = W >> 1, Wq = Wo >> 1;  // Wq threads per output row
    const long per_plane = (long)Ho * Wq;
    const long total = per_plane * C;
    const int b = blockIdx.y;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = idx / per_plane;
        const long rem = idx % per_plane;
        const int i = rem / Wq, t = rem % Wq;
        const float* xp = x + b * xbs + (long)c * H * W;
        const int cl = 4 * t - 1 < 0 ? W - 1 : 4 * t - 1;
        const int cr = 4 * t + 4 >= W ? 0 : 4 * t + 4;
        float h0[4], h1[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int r = 2 * i + a - 1;
            h0[a] = h1[a] = 0.f;
            if (r >= 0 && r < H) {
                const float* row = xp + (long)r * W;
                const f32x4 m = *reinterpret_cast<const f32x4*>(row + 4 * t);
                const float l = row[cl], rr = row[cr];
                h0[a] = fir4(l, m[0], m[1], m[2]);
                h1[a] = fir4(m[1], m[2], m[3], rr);
            }
        }
        const float o0 = fir4(h0[0], h0[1], h0[2], h0[3]), o1 = fir4(h1[0], h1[1], h1[2], h1[3]);
        float2* out = reinterpret_cast<float2*>(y + b * ybs + (long)c * Ho * Wo + (long)i * Wo + 2 * t);
        *out = make_float2(o0, o1);
    }
}

// one thread -> a 2 x 4 output patch (rows 2i, 2i+1 of the block's pair, columns 4t .. 4t+3): input columns 8t-1 .. 8t+8 of
// 6 rows = 24 load instructions for 8 outputs.  (One output pair per thread was 12 loads for 2 outputs and bound by the CU's
// address pipeline, not by HBM: 3.7 TB/s.)  Needs W % 8 == 0, H % 4 == 0; the generic kernel above takes the rest.
// xp: the input plane, i2: pair of output rows, t: quad of output columns; v[o]: output row 2 i2 + o, columns 4t .. 4t+3
// Any even width and height (missing #4: the reference's Resample(down=2) takes them, ops.py:91-143; the engine's own geometries always
// have W % 4 == 0): one thread -> one output, the same fir4 chains on the same operands as the kernels below.
__global__ __launch_bounds__(256) void fir_down2_any_kernel(const float* __restrict__ x, long xbs, float* __restrict__ y, long ybs, int C, int H, int W) {
    const int Ho = H >> 1, Wo = W >> 1;
    const long per_plane = (long)Ho * Wo, total = per_plane * C;
    const int b = blockIdx.y;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = idx / per_plane;
        const long rem = idx % per_plane;
        const int i = rem / Wo, j = rem % Wo;
        const float* xp = x + b * xbs + (long)c * H * W;
        const int c0 = 2 * j - 1 < 0 ? W - 1 : 2 * j - 1, c3 = 2 * j + 2 >= W ? 0 : 2 * j + 2;
        float h[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int r = 2 * i + a - 1;
            h[a] = 0.f;
            if (r >= 0 && r < H) {
                const float* row = xp + (long)r * W;
                h[a] = fir4(row[c0], row[2 * j], row[2 * j + 1], row[c3]);
            }
        }
        y[b * ybs + (long)c * per_plane + rem] = fir4(h[0], h[1], h[2], h[3]);
    }
}

// SHFL (round 6): the two halo columns of a row come from the neighbouring lanes' registers (ds_bpermute: no memory instruction) instead of two 4-byte
// loads per row -- 12 of the 24 load instructions of a patch, the ones that cost the CU's address pipeline a whole wave-wide request for 4 bytes per
// lane.  Valid where a wave's lanes hold consecutive column quads of whole rows (Wq = W / 8 divides 64 and lane % Wq == t: launcher); the wrap of the
// azimuth is then a lane rotation inside the row's Wq lanes.  Same operands into the same fir4 chain: bit-identical outputs.
// X16 (round 6): the input is stored as fp16 (the one-plane mode's activation storage at the full-resolution levels); xp + e0: the input plane
template <int SHFL, bool X16 = false>  // SHFL 0: halo columns by loads | 1: lane rotation inside the row's Wq lanes (Wq divides 64) | 2: lanes +-1, a wave's edge lanes load (Wq a multiple of 64)
__device__ __forceinline__ void fir_down2_patch(const float* __restrict__ xp, long e0, int H, int W, int i2, int t, f32x4 (&v)[2], int lane = 0, int Wq = 0) {
    const int cl = 8 * t - 1 < 0 ? W - 1 : 8 * t - 1;
    const int cr = 8 * t + 8 >= W ? 0 : 8 * t + 8;
    // (SHFL) byte addresses of the lanes holding column quads t - 1 and t + 1 of this row (Wq is a power of two here)
    const int lsrc = SHFL == 1 ? (((lane & ~(Wq - 1)) | ((t - 1) & (Wq - 1))) << 2) : (lane - 1) << 2, rsrc = SHFL == 1 ? (((lane & ~(Wq - 1)) | ((t + 1) & (Wq - 1))) << 2) : (lane + 1) << 2;
    float h[6][4];  // horizontally filtered rows 4 i2 - 1 .. 4 i2 + 4 at the four output columns
    if constexpr (SHFL != 0) {
        // Round 6 (read in the ISA): with the loads of a row inside `if (row inside the image)`, each row was an exec-masked block of its own with an s_waitcnt vmcnt(0)
        // behind it -- the lane exchange needs the values --: six serial round trips per patch, and the edge lanes' halo loads six more.  Now every row is loaded from a
        // clamped (valid) row index up front, twelve vectors in flight, the edge lanes fetch their six halo values in one block, and a row outside the image is zeroed
        // where it always was: after the horizontal filter.  Same operands into the same fir4 chains: bit-identical outputs.
        f32x4 m0[6], m1[6];
        bool in[6];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            const int r = 4 * i2 + a - 1;
            in[a] = r >= 0 && r < H;
            const long er = e0 + (long)(r < 0 ? 0 : r >= H ? H - 1 : r) * W;
            m0[a] = load4<X16>(xp, er + 8 * t);
            m1[a] = load4<X16>(xp, er + 8 * t + 4);
        }
        float le[6] = {}, re[6] = {};
        if constexpr (SHFL == 2) {  // (lane % 64 == t % 64: the first and last lane of a wave have their neighbour in another wave -- or across the seam)
            if (lane == 0 || lane == 63) {
#pragma unroll
                for (int a = 0; a < 6; ++a) {
                    const int r = 4 * i2 + a - 1;
                    const long er = e0 + (long)(r < 0 ? 0 : r >= H ? H - 1 : r) * W;
                    const float e = load1<X16>(xp, er + (lane == 0 ? cl : cr));
                    le[a] = e;
                    re[a] = e;
                }
            }
        }
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            // (all lanes take part in the exchange; lanes of another row pair may be outside the image while this one is inside: their values are
            // not read by anybody of this row)
            float l = __int_as_float(__builtin_amdgcn_ds_bpermute(lsrc, __float_as_int(m1[a][3])));
            float rr = __int_as_float(__builtin_amdgcn_ds_bpermute(rsrc, __float_as_int(m0[a][0])));
            if constexpr (SHFL == 2) {
                l = lane == 0 ? le[a] : l;
                rr = lane == 63 ? re[a] : rr;
            }
            h[a][0] = in[a] ? fir4(l, m0[a][0], m0[a][1], m0[a][2]) : 0.f;
            h[a][1] = in[a] ? fir4(m0[a][1], m0[a][2], m0[a][3], m1[a][0]) : 0.f;
            h[a][2] = in[a] ? fir4(m0[a][3], m1[a][0], m1[a][1], m1[a][2]) : 0.f;
            h[a][3] = in[a] ? fir4(m1[a][1], m1[a][2], m1[a][3], rr) : 0.f;
        }
    } else {
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            const int r = 4 * i2 + a - 1;
            const bool in = r >= 0 && r < H;
            if (in) {
                const long er = e0 + (long)r * W;
                const f32x4 m0 = load4<X16>(xp, er + 8 * t), m1 = load4<X16>(xp, er + 8 * t + 4);
                const float l = load1<X16>(xp, er + cl), rr = load1<X16>(xp, er + cr);
                h[a][0] = fir4(l, m0[0], m0[1], m0[2]);
                h[a][1] = fir4(m0[1], m0[2], m0[3], m1[0]);
                h[a][2] = fir4(m0[3], m1[0], m1[1], m1[2]);
                h[a][3] = fir4(m1[1], m1[2], m1[3], rr);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) h[a][j] = 0.f;
            }
        }
    }
#pragma unroll
    for (int o = 0; o < 2; ++o)  // output row 2 i2 + o takes input rows (2 o) .. (2 o + 3) of the six; the generic kernel's chain (fir4):
#pragma unroll                   // bit-identical results
        for (int j = 0; j < 4; ++j) v[o][j] = fir4(h[2 * o][j], h[2 * o + 1][j], h[2 * o + 2][j], h[2 * o + 3][j]);
}

template <int SHFL, bool X16 = false, bool Y16 = false>
__global__ __launch_bounds__(256) void fir_down2_wide_kernel(const float* __restrict__ x, long xbs, float* __restrict__ y,
                                                             long ybs, int C, int H, int W) {
    const int Ho = H >> 1, Wo = W >> 1, Wq = Wo >> 2, Hq = Ho >> 1;  // Wq threads per pair of output rows
    const long per_plane = (long)Hq * Wq;
    const long total = per_plane * C;
    const int b = blockIdx.y;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = idx / per_plane;
        const long rem = idx % per_plane;
        const int i2 = rem / Wq, t = rem % Wq;
        f32x4 v[2];
        fir_down2_patch<SHFL, X16>(x, b * xbs + (long)c * H * W, H, W, i2, t, v, (int)threadIdx.x & 63, Wq);
#pragma unroll
        for (int o = 0; o < 2; ++o) store4<Y16>(y, b * ybs + (long)c * Ho * Wo + (long)(2 * i2 + o) * Wo + 4 * t, v[o]);
    }
}

// The same pass leaving the GroupNorm statistics of its OUTPUT in the slot grid the convolution epilogues fill (conv_epilogue.h:
// [B][G][2 halves of S = slots / 2][sum, sum of squares] in fp64), so that the norm in front of the stage's first residual block
// (reference efficient_unet.py:95-97 behind :135) needs neither the streaming statistics pass nor -- where conv_f16x2.hip folds the
// norm into its staging waves -- a finalize launch (round 5: 3 + 3 launches of a forward).  grid = (wave slots / 4, G, B): wave slot j
// of (sample, group) owns the group's patches [j ipw, (j + 1) ipw) (a group's planes are contiguous: patch index = plane, row pair,
// column quad), lane L the patches L, L + 64, ...; fp64 from the first addition, one wave total per slot: fixed order, every slot
// written exactly once per launch.  Groups of fewer than 64 channels use half 0 and zero half 1 (as the epilogues do); a
// 64-channel group uses all 2 S slots.
template <int SHFL, bool X16 = false, bool Y16 = false>
__global__ __launch_bounds__(256) void fir_down2_stats_kernel(const float* __restrict__ x, long xbs, float* __restrict__ y, long ybs,
                                                              int cpg, int H, int W, int ipw, double* __restrict__ stat, int slots) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = blockIdx.x * 4 + wave, g = blockIdx.y, b = blockIdx.z, G = gridDim.y;
    const int Ho = H >> 1, Wo = W >> 1, Wq = Wo >> 2, Hq = Ho >> 1;
    const int per_plane = Hq * Wq;
    double s = 0.0, q = 0.0;
    for (int k = lane; k < ipw; k += 64) {
        const int idx = j * ipw + k;  // (< cpg * per_plane <= 2^21)
        const int cg = idx / per_plane, rem = idx - cg * per_plane;
        const int c = g * cpg + cg;
        const int i2 = rem / Wq, t = rem - i2 * Wq;
        f32x4 v[2];
        fir_down2_patch<SHFL, X16>(x, b * xbs + (long)c * H * W, H, W, i2, t, v, lane, Wq);
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const f32x4 st = store4<Y16>(y, b * ybs + (long)c * Ho * Wo + (long)(2 * i2 + o) * Wo + 4 * t, v[o]);  // (the statistics are those of the STORED values)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const double d = (double)st[e];
                s += d;
                q = fma(d, d, q);
            }
        }
    }
    s = wave_sum_f64(s);
    q = wave_sum_f64(q);
    if (lane == 0) {
        double* o = stat + (((size_t)b * G + g) * slots + j) * 2;
        o[0] = s;
        o[1] = q;
        if (cpg < 64) {
            o[slots] = 0.0;  // (half 1: slot S + j, S = slots / 2 -> 2 S doubles further on)
            o[slots + 1] = 0.0;
        }
    }
}

// The bilinear weights of the up-sampler as ONE spelled-out chain, shared by its two kernels (round 6; as fir4 for the down-samplers): which kernel a map
// takes depends on its size, and a sample must come out the same bits whatever batch it is part of (tests/test_hip_configs.py).
__device__ __forceinline__ float up1(float far, float near) { return __builtin_fmaf(0.75f, near, 0.25f * far); }  // 3/4 of the nearer sample + 1/4 of the farther one

// one thread -> input columns 2t, 2t+1 of row i -> a 2x4 output patch
// range (optional): the running maximum of |output| is merged into range[1] as float bits (positive floats order like their
// bit patterns) -- the f16x2 convolution that consumes this tensor needs it below 65504 (engine.hip, r2dm_check_range)
template <bool X16 = false, bool Y16 = false>  // (round 6) fp16 storage of the input / the output; the running maximum is that of the STORED values
__global__ __launch_bounds__(256) void fir_up2_kernel(const float* __restrict__ x, long xbs, float* __restrict__ y,
                                                      long ybs, int C, int H, int W, int* __restrict__ range) {
    float amax = 0.f;
    const int Wh = W >> 1, Wo = W << 1;
    const long per_plane = (long)H * Wh;
    const long total = per_plane * C;
    const int b = blockIdx.y;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = idx / per_plane;
        const long rem = idx % per_plane;
        const int i = rem / Wh, t = rem % Wh;
        const long e0 = b * xbs + (long)c * H * W;
        const int cl = 2 * t - 1 < 0 ? W - 1 : 2 * t - 1;
        const int cr = 2 * t + 2 >= W ? 0 : 2 * t + 2;
        float h[3][4];  // horizontally upsampled rows i-1, i, i+1 at output columns 4t..4t+3
#pragma unroll
        for (int a = 0; a < 3; ++a) {  // (round 6: every row loaded from a valid row index, nine loads in flight; a row outside the image is zeroed behind its filter --
            const int r = i + a - 1;   //  inside `if (row inside)` each row was a block of its own with a full wait behind it)
            const bool in = r >= 0 && r < H;
            const long er = e0 + (long)(in ? r : i) * W;
            float2 m;
            if constexpr (X16) m = make_float2(load1<true>(x, er + 2 * t), load1<true>(x, er + 2 * t + 1));
            else m = *reinterpret_cast<const float2*>(x + er + 2 * t);
            const float l = load1<X16>(x, er + cl), rr = load1<X16>(x, er + cr);
            h[a][0] = in ? up1(l, m.x) : 0.f;
            h[a][1] = in ? up1(m.y, m.x) : 0.f;
            h[a][2] = in ? up1(m.x, m.y) : 0.f;
            h[a][3] = in ? up1(rr, m.y) : 0.f;
        }
        f32x4 e, o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            e[j] = up1(h[0][j], h[1][j]);
            o[j] = up1(h[2][j], h[1][j]);
        }
        const long eo = b * ybs + (long)c * (4L * H * W) + (long)(2 * i) * Wo + 4 * t;
        const f32x4 es = store4<Y16>(y, eo, e), os = store4<Y16>(y, eo + Wo, o);
        if (range) {
#pragma unroll
            for (int j = 0; j < 4; ++j) amax = fmaxf(amax, fmaxf(fabsf(es[j]), fabsf(os[j])));
        }
    }
    if (range) {
        amax = wave_max_f32(amax);
        if ((threadIdx.x & 63) == 0) {
            const int bits = __float_as_int(amax);
            if (bits > __atomic_load_n(range + 1, __ATOMIC_RELAXED)) atomicMax(range + 1, bits);  // (rarely taken after the first blocks)
        }
    }
}

// Round 6: one thread -> input columns 4t .. 4t+3 of row i -> a 2 x 8 output patch: three 16-byte loads (rows i-1, i, i+1) and four 16-byte stores, the
// two halo columns of a row from the neighbouring lanes' registers (ds_bpermute) -- the kernel above issues nine loads (8 + 4 + 4 bytes per row) for two
// stores and is bound by the CU's address pipeline (4.2 TB/s).  Lanes at a wave's edge or at the azimuth seam load their halo (one or two lanes of a masked
// instruction).  The same chains (up1) on the same operands: bit-identical to the kernel above.

template <bool X16 = false, bool Y16 = false>
__global__ __launch_bounds__(256) void fir_up2_wide_kernel(const float* __restrict__ x, long xbs, float* __restrict__ y,
                                                           long ybs, int C, int H, int W, int* __restrict__ range) {
    float amax = 0.f;
    const int Wt = W >> 2, Wo = W << 1;
    const long per_plane = (long)H * Wt;
    const long total = per_plane * C;
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = idx / per_plane;
        const long rem = idx % per_plane;
        const int i = rem / Wt, t = rem % Wt;
        const long e0 = b * xbs + (long)c * H * W;
        // the lanes to the left / right hold the neighbouring quads of the same row unless this lane is first / last in its wave or in its row
        // (a partial last wave: its active lanes are the low ones, and total is a multiple of Wt -- a right neighbour in the same row is active)
        const bool lsh = lane > 0 && t > 0, rsh = lane < 63 && t < Wt - 1;
        const int cl = 4 * t - 1 < 0 ? W - 1 : 4 * t - 1;
        const int cr = 4 * t + 4 >= W ? 0 : 4 * t + 4;
        float h[3][8];  // horizontally upsampled rows i-1, i, i+1 at output columns 8t .. 8t+7
        // (round 6, as fir_down2_patch: the three rows' vectors in flight together, the edge lanes' halo values fetched in ONE divergent block -- per row they were
        // two blocks with a full wait each)
        f32x4 mr[3];
        bool inr[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int r = i + a - 1;
            inr[a] = r >= 0 && r < H;  // (rows i-1 / i+1 of lanes in other rows of the wave may differ: every lane takes part in the exchange)
            mr[a] = load4<X16>(x, e0 + (long)(inr[a] ? r : i) * W + 4 * t);
        }
        float le[3] = {}, re[3] = {};
        if (!lsh || !rsh) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const long er = e0 + (long)(inr[a] ? i + a - 1 : i) * W;
                if (!lsh) le[a] = load1<X16>(x, er + cl);
                if (!rsh) re[a] = load1<X16>(x, er + cr);
            }
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const bool in = inr[a];
            const f32x4 m = mr[a];
            float l = __int_as_float(__builtin_amdgcn_ds_bpermute((lane - 1) << 2, __float_as_int(m[3])));
            float rr = __int_as_float(__builtin_amdgcn_ds_bpermute((lane + 1) << 2, __float_as_int(m[0])));
            l = lsh ? l : le[a];
            rr = rsh ? rr : re[a];
            h[a][0] = in ? up1(l, m[0]) : 0.f;
            h[a][1] = in ? up1(m[1], m[0]) : 0.f;
            h[a][2] = in ? up1(m[0], m[1]) : 0.f;
            h[a][3] = in ? up1(m[2], m[1]) : 0.f;
            h[a][4] = in ? up1(m[1], m[2]) : 0.f;
            h[a][5] = in ? up1(m[3], m[2]) : 0.f;
            h[a][6] = in ? up1(m[2], m[3]) : 0.f;
            h[a][7] = in ? up1(rr, m[3]) : 0.f;
        }
        const long eo = b * ybs + (long)c * (4L * H * W) + (long)(2 * i) * Wo + 8 * t;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            f32x4 e, o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                e[j] = up1(h[0][4 * q + j], h[1][4 * q + j]);
                o[j] = up1(h[2][4 * q + j], h[1][4 * q + j]);
            }
            const f32x4 es = store4<Y16>(y, eo + 4 * q, e), os = store4<Y16>(y, eo + Wo + 4 * q, o);
#pragma unroll
            for (int j = 0; j < 4; ++j) amax = fmaxf(amax, fmaxf(fabsf(es[j]), fabsf(os[j])));  // (unconditional: only merged into the flag if `range`)
        }
    }
    if (range) {
        amax = wave_max_f32(amax);
        if ((threadIdx.x & 63) == 0) {
            const int bits = __float_as_int(amax);
            if (bits > __atomic_load_n(range + 1, __ATOMIC_RELAXED)) atomicMax(range + 1, bits);
        }
    }
}

static int grid_for(long total) {
    long g = (total + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

// wave slots a (sample, group) of the statistics variant uses, or 0 where the variant does not apply (the caller then runs the
// streaming statistics pass): a pure function of the geometry -- the engine's dry walk and its real walk must agree
int fir_down2_stat_slots(int C, int G, int H, int W) {
    const char* fe = getenv("R2DM_FIR_STATS");  // (R2DM_FIR_STATS=0: the streaming pass instead -- A/B, tests; read per call like R2DM_GN_FOLD)
    const bool off = fe && atoi(fe) == 0;
    if (off || G < 1 || C % G || (W % 8) || (H % 4)) return 0;
    const int cpg = C / G;
    if (cpg < 8 || cpg > 64 || (cpg & (cpg - 1))) return 0;
    const int slots = conv_stat_slots(H / 2, W / 2);
    const int used = cpg < 64 ? slots / 2 : slots;
    const long items = (long)cpg * (H / 4) * (W / 8);
    if (used < 4 || used % 4 || items % used || (items / used) % 64 || items / used > 64 * 16) return 0;
    return used;
}

// halo columns from the neighbouring lanes (fir_down2_patch<true>): the Wq = W / 8 column quads of a row sit in Wq consecutive lanes of one wave
// (R2DM_FIR_SHFL=0: the loads, as until round 5 -- A/B, tests; read per call)
static int shfl_mode(int W) {
    const char* e = getenv("R2DM_FIR_SHFL");
    const int Wq = W / 8;
    if ((e && atoi(e) == 0) || W % 8 || Wq < 2) return 0;
    return Wq <= 64 && 64 % Wq == 0 ? 1 : Wq % 64 == 0 ? 2 : 0;  // (whole rows inside a wave | whole waves inside a row)
}

hipError_t launch_fir_down2(const float* x, long xbs, float* y, long ybs, int B, int C, int H, int W, hipStream_t s, double* stat, int G, int x16, int y16) {
    if ((W & 1) || (H & 1)) return hipErrorInvalidValue;
    if (y16 && !x16) return hipErrorInvalidValue;  // (fp16 storage, round 6: fp16 -> fp16 between the full-resolution levels, fp16 -> fp32 below them)
    if (x16) {  // the kernels of the engine's geometries only (W % 8 == 0, H % 4 == 0)
        if (W % 8 || H % 4) return hipErrorInvalidValue;
        const int sm = shfl_mode(W);
        if (stat) {
            const int used = fir_down2_stat_slots(C, G, H, W);
            if (!used) return hipErrorInvalidValue;
            const int cpg = C / G;
            const int ipw = (int)((long)cpg * (H / 4) * (W / 8) / used);
            const dim3 grid(used / 4, G, B);
            const int slots = conv_stat_slots(H / 2, W / 2);
#define R2DM_FDS(SM, Y) fir_down2_stats_kernel<SM, true, Y><<<grid, 256, 0, s>>>(x, xbs, y, ybs, cpg, H, W, ipw, stat, slots)
            if (y16) { if (sm == 1) R2DM_FDS(1, true); else if (sm == 2) R2DM_FDS(2, true); else R2DM_FDS(0, true); }
            else { if (sm == 1) R2DM_FDS(1, false); else if (sm == 2) R2DM_FDS(2, false); else R2DM_FDS(0, false); }
#undef R2DM_FDS
            return hipGetLastError();
        }
        const dim3 grid(grid_for((long)C * (H / 4) * (W / 8)), B);
#define R2DM_FDW(SM, Y) fir_down2_wide_kernel<SM, true, Y><<<grid, 256, 0, s>>>(x, xbs, y, ybs, C, H, W)
        if (y16) { if (sm == 1) R2DM_FDW(1, true); else if (sm == 2) R2DM_FDW(2, true); else R2DM_FDW(0, true); }
        else { if (sm == 1) R2DM_FDW(1, false); else if (sm == 2) R2DM_FDW(2, false); else R2DM_FDW(0, false); }
#undef R2DM_FDW
        return hipGetLastError();
    }
    if (W & 3) {  // (never inside the engine: its widths are multiples of 32)
        if (stat) return hipErrorInvalidValue;
        fir_down2_any_kernel<<<dim3(grid_for((long)C * (H / 2) * (W / 2)), B), 256, 0, s>>>(x, xbs, y, ybs, C, H, W);
        return hipGetLastError();
    }
    if (stat) {
        const int used = fir_down2_stat_slots(C, G, H, W);
        if (!used) return hipErrorInvalidValue;
        const int cpg = C / G;
        const int ipw = (int)((long)cpg * (H / 4) * (W / 8) / used);
        // (lane % Wq == t: a wave slot's patches start at a multiple of 64 -- fir_down2_stat_slots -- and 64 % Wq == 0)
        const int sm = shfl_mode(W);
        if (sm == 1) fir_down2_stats_kernel<1><<<dim3(used / 4, G, B), 256, 0, s>>>(x, xbs, y, ybs, cpg, H, W, ipw, stat, conv_stat_slots(H / 2, W / 2));
        else if (sm == 2) fir_down2_stats_kernel<2><<<dim3(used / 4, G, B), 256, 0, s>>>(x, xbs, y, ybs, cpg, H, W, ipw, stat, conv_stat_slots(H / 2, W / 2));
        else fir_down2_stats_kernel<0><<<dim3(used / 4, G, B), 256, 0, s>>>(x, xbs, y, ybs, cpg, H, W, ipw, stat, conv_stat_slots(H / 2, W / 2));
        return hipGetLastError();
    }
    if (W % 8 == 0 && H % 4 == 0 && getenv("R2DM_FIR_NARROW") == nullptr) {
        const long tot = (long)C * (H / 4) * (W / 8);
        // (the grid-stride loop keeps lane % Wq == t: 256 and the stride are multiples of 64, per_plane a multiple of Wq)
        const int sm = shfl_mode(W);
        if (sm == 1) fir_down2_wide_kernel<1><<<dim3(grid_for(tot), B), 256, 0, s>>>(x, xbs, y, ybs, C, H, W);
        else if (sm == 2) fir_down2_wide_kernel<2><<<dim3(grid_for(tot), B), 256, 0, s>>>(x, xbs, y, ybs, C, H, W);
        else fir_down2_wide_kernel<0><<<dim3(grid_for(tot), B), 256, 0, s>>>(x, xbs, y, ybs, C, H, W);
        return hipGetLastError();
    }
    const long total = (long)C * (H / 2) * (W / 4);
    fir_down2_kernel<<<dim3(grid_for(total), B), 256, 0, s>>>(x, xbs, y, ybs, C, H, W);
    return hipGetLastError();
}

hipError_t launch_fir_up2(const float* x, long xbs, float* y, long ybs, int B, int C, int H, int W, hipStream_t s, int* range, int x16, int y16) {
    if (W & 1) return hipErrorInvalidValue;
    if (x16 && !y16) return hipErrorInvalidValue;  // (fp16 storage, round 6: fp32 -> fp16 into the full-resolution levels, fp16 -> fp16 between them)
    const char* e = getenv("R2DM_FIR_UP_WIDE");  // (0: the two-column kernel, as until round 5; 2: the wide kernel at any size -- A/B, tests; read per call)
    if (y16) {
        const bool wide = W % 4 == 0 && !(e && atoi(e) == 0) && ((long)B * C * H * W >= (8L << 20) || (e && atoi(e) == 2));
        const dim3 grid(grid_for(wide ? (long)C * H * (W / 4) : (long)C * H * (W / 2)), B);
        if (wide) {
            if (x16) fir_up2_wide_kernel<true, true><<<grid, 256, 0, s>>>(x, xbs, y, ybs, C, H, W, range);
            else fir_up2_wide_kernel<false, true><<<grid, 256, 0, s>>>(x, xbs, y, ybs, C, H, W, range);
        } else {
            if (x16) fir_up2_kernel<true, true><<<grid, 256, 0, s>>>(x, xbs, y, ybs, C, H, W, range);
            else fir_up2_kernel<false, true><<<grid, 256, 0, s>>>(x, xbs, y, ybs, C, H, W, range);
        }
        return hipGetLastError();
    }
    // (small maps stay on the two-column kernel: with a quarter of the threads the wide one is latency-bound -- 256 channels of 8 x 128 at batch 8: 18 -> 22 us)
    if (W % 4 == 0 && !(e && atoi(e) == 0) && ((long)B * C * H * W >= (8L << 20) || (e && atoi(e) == 2))) {
        fir_up2_wide_kernel<<<dim3(grid_for((long)C * H * (W / 4)), B), 256, 0, s>>>(x, xbs, y, ybs, C, H, W, range);
        return hipGetLastError();
    }
    const long total = (long)C * H * (W / 2);
    fir_up2_kernel<<<dim3(grid_for(total), B), 256, 0, s>>>(x, xbs, y, ybs, C, H, W, range);
    return hipGetLastError();
}

}  // namespace r2dm

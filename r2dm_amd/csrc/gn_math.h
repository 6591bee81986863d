// The scalar arithmetic of a GroupNorm's folded affine, shared by gn_finalize_kernel (norm.hip) and the consumer-side fold of
// conv_f16x2.hip: both must produce the same bits (reference nn.GroupNorm / AdaGN: /root/reference/models/efficient_unet.py:33,72,
// /root/reference/models/ops.py:176-200).
#pragma once
#include "common.h"

namespace r2dm {

struct GnMoments {
    float mean, rstd;
    bool well_conditioned;  // the variance did not cancel: |x_hat| <= sqrt(n - 1) holds for the COMPUTED statistics too
};

// (sum, sum of squares) over n elements -> mean and 1 / sqrt(var + eps), fp64 inside
__device__ __forceinline__ GnMoments gn_moments(double sum, double sq, double n, float eps) {
    const double mean_d = sum / n;
    double var_d = sq / n - mean_d * mean_d;
    var_d = var_d > 0.0 ? var_d : 0.0;
    GnMoments m;
    m.mean = (float)mean_d;
    m.rstd = (float)(1.0 / sqrt(var_d + (double)eps));
    m.well_conditioned = var_d > 1e-6 * mean_d * mean_d;
    return m;
}

// y = x a + d with a = rstd w, d = sh - mean a (one FMA: spelled out so that both users round alike)
__device__ __forceinline__ float2 gn_affine(const GnMoments& m, float w, float sh) {
    const float a = m.rstd * w;
    return make_float2(a, __builtin_fmaf(-m.mean, a, sh));
}

// Range guard of the fp16 consumers: a bound on |a x + d| over the group from the DATA, |a| M + |d| with M >= max|x| (the producers'
// recorded maximum, or the square root of the largest slot energy).  Where that one is looser -- a near-constant group: large
// |mean| / sigma makes |a| M and |d| both huge although they cancel -- the worst case of a normalised value, |x_hat| <= sqrt(n - 1)
// (Samuelson): |w| sqrt(n) + |sh|.  ADVICE round 4: that second bound only holds for the exact mean and variance; when the computed
// variance has cancelled (clamped to 0 or below 1e-6 mean^2) rstd can be far above 1 / sigma, so it is not used there and the data
// bound is taken in the form |w| rstd (M + |mean|) + |sh|, which never cancels.
__device__ __forceinline__ float gn_bound(const GnMoments& m, float2 ad, float w, float sh, float gmax, double n) {
    const float data_bound = fabsf(ad.x) * gmax + fabsf(ad.y);
    if (!m.well_conditioned) {
        const float safe = fabsf(w) * m.rstd * (gmax + fabsf(m.mean)) + fabsf(sh);
        return data_bound > safe ? data_bound : safe;  // (a NaN stays: the comparison is false)
    }
    const float worst_bound = (fabsf(w) * (float)sqrt(n) + fabsf(sh)) * 1.000001f;
    return data_bound > worst_bound ? worst_bound : data_bound;  // (a NaN data bound stays: the comparison is false)
}

}  // namespace r2dm

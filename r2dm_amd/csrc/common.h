// Shared declarations for libr2dm_hip.so (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace r2dm {

constexpr int kWave = 64;  // CDNA wavefront
constexpr float kInvSqrt2 = 0.70710678118654752440f;

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

// fp16 STORAGE of activations (the one-plane mode, round 5 / 6): four consecutive pixels of one channel are 8 bytes.  Pointers stay typed float*; an
// element index e of an fp16 tensor is the byte offset 2 e.  Conversions are RNE (v_cvt_pk_f16_f32) / exact (fp16 -> fp32).
using f16x2v = __attribute__((ext_vector_type(2))) _Float16;
__device__ __forceinline__ f32x4 f16x4_to_f32(unsigned long long r) {
    return f32x4{(float)__builtin_bit_cast(_Float16, (unsigned short)r), (float)__builtin_bit_cast(_Float16, (unsigned short)(r >> 16)),
                 (float)__builtin_bit_cast(_Float16, (unsigned short)(r >> 32)), (float)__builtin_bit_cast(_Float16, (unsigned short)(r >> 48))};
}
__device__ __forceinline__ unsigned long long f32_to_f16x4(const f32x4& v) {
    using f32x2v = __attribute__((ext_vector_type(2))) float;
    return (unsigned long long)__builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{v[0], v[1]}, f16x2v)) |
           ((unsigned long long)__builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{v[2], v[3]}, f16x2v)) << 32);
}
// four consecutive elements starting at element index `e` (a multiple of 4) of a tensor stored as fp32 (X16 = false) or fp16
template <bool X16>
__device__ __forceinline__ f32x4 load4(const float* base, long e) {
    if constexpr (X16) return f16x4_to_f32(*reinterpret_cast<const unsigned long long*>(reinterpret_cast<const unsigned short*>(base) + e));
    else return *reinterpret_cast<const f32x4*>(base + e);
}
template <bool X16>
__device__ __forceinline__ float load1(const float* base, long e) {
    if constexpr (X16) return (float)__builtin_bit_cast(_Float16, reinterpret_cast<const unsigned short*>(base)[e]);
    else return base[e];
}
// stores v (RNE for fp16) and returns the values AS STORED (what a consumer will read: statistics and range maxima are taken of those)
template <bool Y16>
__device__ __forceinline__ f32x4 store4(float* base, long e, const f32x4& v) {
    if constexpr (Y16) {
        const unsigned long long pk = f32_to_f16x4(v);
        *reinterpret_cast<unsigned long long*>(reinterpret_cast<unsigned short*>(base) + e) = pk;
        return f16x4_to_f32(pk);
    } else {
        *reinterpret_cast<f32x4*>(base + e) = v;
        return v;
    }
}

// A (B, C0+C1, H, W) activation that may live in two allocations (channel concat without a copy,
// reference efficient_unet.py:11-12,290-292).  Planes are H*W contiguous floats; `bs` is the batch
// stride in floats (0 = broadcast over the batch, used for the constant coordinate encoding).
struct Src {
    const float* p0;
    const float* p1;
    int c0, c1;
    long bs0, bs1;
    __host__ __device__ int channels() const { return c0 + c1; }
    __device__ __forceinline__ const float* plane(int b, int c, long hw) const {
        return c < c0 ? p0 + b * bs0 + (long)c * hw : p1 + b * bs1 + (long)(c - c0) * hw;
    }
};

__device__ __forceinline__ float silu_f(float v) {
    // x * sigmoid(x) on the transcendental unit: v_exp_f32 + v_rcp_f32 (1 ulp each).  Absolute error
    // <= ~2e-7*|v|, i.e. fp32-roundoff class like the convolution it feeds (parity budget ~1e-6).
#ifdef R2DM_ACCURATE_SILU
    return v / (1.0f + expf(-v));
#else
    return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
#endif
}

// XCD-aware bijective block remap (8 XCDs, block b is dispatched to XCD b % 8): logical ids that
// are neighbours (share input halo / weights) land on the same XCD's L2.
__host__ __device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + idx;
}

// ---- kernel launchers (each in its own .hip) ------------------------------------------

// PRO_PRESPLIT (conv_f16x2 only): the input is the pre-split tensor of presplit.hip -- affine / SiLU / f16 split already applied
enum Prologue { PRO_NONE = 0, PRO_AFFINE = 1, PRO_AFFINE_SILU = 2, PRO_PRESPLIT = 3 };
// ALGO_F32: fp32-input MFMA (conv_mfma.hip).  ALGO_BF16X3: fp32 operands split exactly into three bf16 pieces, six
// bf16 MFMA products per fp32 product, fp32 accumulation (conv_bf16x3.hip) -- same accuracy class, 2.7x fewer
// matrix-pipe cycles; needs 3x3, Cin % 16 == 0, Cout % 64 == 0.
// ALGO_DIRECT: Cout <= 4 (out_conv): HBM-bound, direct fp32 FMA convolution, weights kept OIHW (conv_direct.hip).
// ALGO_F16X2: fp32 operands split exactly to 22 bits into an fp16 piece and a 2^11-scaled fp16 residual, three fp16 MFMA
// products per fp32 product, two fp32 accumulators (conv_f16x2.hip): more accurate than either of the above on the
// matrix pipe (profiles/r02_f16x2_probe.txt) at half the products of ALGO_BF16X3, but operands must fit the fp16 range
// -- the engine uses it for GroupNorm-normalised inputs only, behind a range flag.
// ALGO_P1F16: the 1x1 convolutions around the attention core in the ALGO_F16X2 arithmetic (proj_f16x2.hip)
enum ConvAlgo { ALGO_F32 = 0, ALGO_BF16X3 = 1, ALGO_DIRECT = 2, ALGO_F16X2 = 3, ALGO_P1F16 = 4 };

struct ConvParams {
    Src x;               // input activation
    const float* w;      // packed weights [nCoT][CinPad][taps][CO_T]
    const float* bias;   // [Cout]
    const float2* aff;   // [B][Cin] (a, d): GroupNorm folded to x*a+d, or nullptr
    const float* res;    // residual added before scaling, Cout planes, or nullptr
    long res_bs;
    const float* scale;  // device scalar multiplied after the residual add, or nullptr
    float* y;            // output, Cout planes
    long y_bs;
    int B, H, W, Cin, CinPad, Cout;
    int taps;            // 9 or 1
    int co_tile;         // 32 / 64 / 128 (must match the packing)
    int px_rows = 4;     // ALGO_F16X2: image rows per pixel tile -- 4, or 8 with co_tile 64 (conv_f16x2.hip's one-accumulator "tall" tile; must match the packing)
    int prologue;        // Prologue
    int algo = ALGO_F32; // ConvAlgo (must match the packing of `w`)
    int pieces = 3;      // ALGO_BF16X3: bf16 pieces per fp32 operand (3 = exact split, 6 products; the 2-piece variant of
                         // round 1 is superseded by ALGO_F16X2 and no longer dispatched).  ALGO_F16X2 / ALGO_P1F16: 1 = the h plane
                         // alone, one fp16 product per MAC (the reduced-precision bulk mode); anything else = the split arithmetic
    int sign_shift = 0;  // ALGO_BF16X3 shallow kernel: accumulator sign flips every 2^sign_shift chunks (set by the launcher)
    // optional fused GroupNorm statistics of the OUTPUT (for the GroupNorm that consumes it): per (sample, group)
    // partial (sum, sum of squares) in fp64, one slot per (pixel tile, pixel wave): [B][stat_G][stat_slots][2]
    double* stat = nullptr;
    int stat_G = 0, stat_goff = 0, stat_cpg = 0, stat_slots = 0;
    // (the slots double as the fp16 range guard of the GroupNorm's consumers: every element of a slot is bounded by the
    // square root of the slot's sum of squares -- gn_finalize, norm.hip: data-driven and free for the producer)
    // optional device scalar multiplied into the matrix product before the bias: the inverse of the power-of-two scale the
    // fp16 packers apply to a layer's weights (f16x2.h: max|w| is brought to [2^9, 2^10) so that tiny weights keep 22 bits)
    const float* wscale = nullptr;
    // optional: range[1] takes the running maximum of |output| as float bits -- the engine asks for it when the consumer runs on
    // the fp16 matrix pipe with no GroupNorm in between (a down-sampling convolution; the attention core behind the qkv projection)
    int* range = nullptr;
    // walk the launch's tiles in descending order (conv_f16x2.hip): consecutive layers then alternate direction, so a layer starts on the
    // part of its input that its producer wrote LAST -- the part most likely still in the 256 MB Infinity Cache (level-1 tensors are 134 MB)
    int reverse = 0;
    // GroupNorm folded into its CONSUMER (conv_f16x2.hip, round 5): instead of `aff` the launch gets the producers' statistics slots
    // and the norm's parameters; every block's staging waves reduce the slots of the block's sample in gn_finalize_kernel's order
    // (norm.hip: bit-identical (a, d)) while the first pixels are on their way -- one launch less per GroupNorm.  Only where
    // conv_f16x2_fold_supported says so; `aff` must be nullptr then.
    const double* gn_partial = nullptr;  // [B][8][gn_stride][2] fp64 (sum, sum of squares); slots [0, gn_slots) are read
    int gn_slots = 0, gn_stride = 0, gn_cpg = 0;
    float gn_eps = 0.f;
    const float* gn_gamma = nullptr;     // [Cin] / nullptr
    const float* gn_beta = nullptr;
    const float* gn_ada = nullptr;       // [B][gn_ada_stride] rows = [scale(Cin) | shift(Cin)] (AdaGN) / nullptr
    long gn_ada_stride = 0;
    int* gn_range = nullptr;             // the engine's range flag (the bound gn_finalize would have recorded) / nullptr
    // fp16 STORAGE of activations (conv_f16x2.hip, the one-plane mode: round 5): x16 -- the input tensor(s) hold fp16; y16 -- the output and
    // the residual do.  Pointers stay typed float*, batch strides stay in ELEMENTS.
    int x16 = 0, y16 = 0;
    int stagger = 0;  // experiment (R2DM_F2_STAGGER, conv_f16x2.hip): every other block of an XCD starts so many clock ticks late -- de-phases the blocks' tile ends
    unsigned long long* prof = nullptr;  // optional [nblk][4] s_memtime stamps (perf probe; nullptr in production)
};
int conv_pick_algo(int Cin, int Cout, int taps);  // env R2DM_CONV_ALGO=f32 forces ALGO_F32 everywhere
long conv_packed_floats(int algo, int Cin, int Cout, int taps, int co_tile, int cin_pad);
int conv_pick_co_tile(int Cout, int taps, long pixels_times_batch);
int conv_cin_pad(int Cin, int taps, int co_tile);
hipError_t launch_conv(const ConvParams& p, hipStream_t s);
// src_cin / src_off: pack input channels [src_off, src_off + Cin) of a (Cout, src_cin, k, k) source (0 = whole tensor)
hipError_t launch_pack_conv(const float* w_oihw, float* dst, int Cout, int Cin, int taps, int co_tile,
                            int cin_pad, hipStream_t s, int algo = ALGO_F32, int src_cin = 0, int src_off = 0);
bool conv_bf16x3_supported(int Cin, int Cout, int taps);
bool conv_direct_supported(int Cout, int taps);
bool conv_few_in_supported(int Cin, int Cout, int taps, int H, int W);  // ALGO_DIRECT's second kernel (in_conv over the data channels)
hipError_t launch_conv_direct(const ConvParams& p, hipStream_t s);
long conv_bf16x3_packed_floats(int Cin, int Cout);
int conv_bf16x3_co_tile(int Cin, int Cout, long pixels_times_batch);
hipError_t launch_pack_conv_bf16x3(const float* w_oihw, float* dst, int Cout, int Cin, int co_tile, hipStream_t s);
hipError_t launch_conv_bf16x3(const ConvParams& p, hipStream_t s);
bool conv_f16x2_supported(int Cin, int Cout, int taps, int H, int W, int co_tile = 64, int px_rows = 4);  // tiles 64 x 4 | 128 x 4 | 64 x 8 (conv_f16x2.hip)
long conv_f16x2_packed_floats(int Cin, int Cout);
// One-accumulator tiles where the launch still has a tile per CU at the planned batch -- 128 channels x 4 rows, or 64 x 8 for layers with
// fewer than 128 output channels --, else 64 x 4 (two accumulators).  Returns the channels, *px_rows the rows (nullptr: 4-row tiles only);
// env R2DM_F2_CO_TILE = 64 | 128 | 64x8 forces one of them where the shape allows (experiments, per-kernel tests)
int conv_f16x2_pick_co_tile(int Cin, int Cout, int H, int W, long pixels_times_batch, int* px_rows = nullptr);
// range_flag (device int, may be nullptr): bit 0 is set if a weight does not fit the fp16 range
// wscale (device float[2], may be nullptr = unscaled): [0] scratch (max|w| as float bits), [1] <- the inverse of the power-of-two
// scale applied to the layer's weights (ConvParams::wscale points there)
hipError_t launch_pack_conv_f16x2(const float* w_oihw, float* dst, int Cout, int Cin, int* range_flag, hipStream_t s, float* wscale = nullptr, int co_tile = 64,
                                  int px_rows = 4);
hipError_t launch_weight_absmax(const float* w, long n, int* max_bits, hipStream_t s);  // max_bits <- float bits of max|w| (zeroed first)
hipError_t launch_conv_f16x2(const ConvParams& p, hipStream_t s);
// can this launch (shape, tile, batch: everything but the gn_* fields) take its GroupNorm folded -- 8 groups of 8 .. 64 channels, at
// most 4 x 256 statistics slots per group to read, every block's tiles inside ONE sample
bool conv_f16x2_fold_supported(const ConvParams& p, int groups, int slots);
// the operand pre-pass of conv_f16x2 (presplit.hip): xs <- [b][chunk][plane][group][H + 2][W][8 ch] fp16 of silu(x a + d) (prologue as in
// ConvParams); a convolution with prologue = PRO_PRESPLIT and x.p0 = xs, x.bs0 = presplit_floats(1, Cin, H, W) consumes it
long presplit_floats(int B, int Cin, int H, int W);
bool presplit_supported(const Src& x, int Cin, int H, int W);
hipError_t launch_presplit(const Src& x, const float2* aff, int prologue, float* xs, int B, int Cin, int H, int W, hipStream_t s);
// ... with the GroupNorm folded in (round 6): partial = the producers' statistics slots [B][G][stride][2], of which the first nslots per group count
hipError_t launch_presplit_fold(const Src& x, int prologue, float* xs, int B, int Cin, int H, int W, const double* partial, int stride, int nslots, int cpg,
                                float eps, const float* gamma, const float* beta, const float* ada, long ada_stride, int* range, hipStream_t s);
bool proj_f16x2_supported(int Cin, int Cout, int taps, int H, int W);
long proj_f16x2_packed_floats(int Cin, int Cout);
hipError_t launch_pack_proj_f16x2(const float* w_oi, float* dst, int Cout, int Cin, int* range_flag, hipStream_t s, float* wscale = nullptr);
hipError_t launch_proj_f16x2(const ConvParams& p, hipStream_t s);

struct GNParams {
    Src x;
    int B, H, W, groups;
    float eps;
    const float* gamma;  // [C] or nullptr  (affine GroupNorm)
    const float* beta;
    const float* ada;    // [B][ada_stride] rows = [scale(C) | shift(C)] or nullptr (AdaGN)
    long ada_stride;
    double* partial;     // scratch [B][G][splits][2]
    float2* aff;         // out [B][C]
    float* stats;        // optional out [B][G][2] (mean, rstd) for tests, may be nullptr
    // optional device int[2]: [1] takes the running maximum (as float bits) of a bound |a| M + |d| on the normalised, affine-
    // transformed tensor, M >= max|x| taken from the DATA: the maximum gn_partial recorded (partial_max), or the square root of
    // the largest slot energy (fused statistics) -- the fp16-operand consumers need it below 65504 ([0]: weight flag of the packer)
    int* range_flag = nullptr;
    float* partial_max = nullptr;  // [B][G][splits]: largest |x| of every split (written by gn_partial), or nullptr
};
int gn_splits(int B, int groups, long group_elems);
int conv_stat_slots(int H, int W);  // slots per (sample, group) a convolution's fused statistics occupy
// finalize only: partial [B][G][splits][2] already filled (by a convolution epilogue)
hipError_t launch_group_norm_finalize(const GNParams& p, int C, int splits, hipStream_t s);
hipError_t launch_group_norm(const GNParams& p, hipStream_t s);
hipError_t launch_gn_apply(const float* x, const float2* aff, float* y, int B, int C, long hw, int silu,
                           hipStream_t s);

// stat != nullptr: the GroupNorm statistics of the OUTPUT (G groups) go to the convolution epilogues' slot grid
// ([B][G][conv_stat_slots(H / 2, W / 2)][2] doubles); only where fir_down2_stat_slots(C, G, H, W) != 0
hipError_t launch_fir_down2(const float* x, long xbs, float* y, long ybs, int B, int C, int H, int W,
                            hipStream_t s, double* stat = nullptr, int G = 0, int x16 = 0, int y16 = 0);  // x16 / y16: fp16 storage of the input / output (round 6)
int fir_down2_stat_slots(int C, int G, int H, int W);
hipError_t launch_fir_up2(const float* x, long xbs, float* y, long ybs, int B, int C, int H, int W,
                          hipStream_t s, int* range = nullptr, int x16 = 0, int y16 = 0);  // range[1]: running max |output| as float bits (may be nullptr)

// qkv: (B, 3C, N) channel-major [q | k | v]; out (B, C, N)
// planes: 0 = fp32-input MFMA, 2 = fp16 matrix pipe with split operands, 1 = fp16 matrix pipe, one product (attention.hip)
hipError_t launch_attention(const float* qkv, float* out, int B, int C, int heads, int N, hipStream_t s, int planes = 0);
bool attention_supported(int C, int heads, int N);

struct EmbedParams {
    const float* cond;  // [B]
    const float* freqs; // [base/2] sinusoid frequencies (host-precomputed)
    const float* w1;    // [T][base]
    const float* b1;
    const float* w2;    // [T][T]
    const float* b2;
    float* act;         // out [B][T] = SiLU(time embedding)
    float* hidden;      // scratch [B][T]
    int B, base, T;
};
hipError_t launch_time_embedding(const EmbedParams& p, hipStream_t s);
// out[b][r] = dot(act[b], w[r]) + bias[r]   for all AdaGN projections at once
hipError_t launch_ada_proj(const float* act, const float* w, const float* bias, float* out, int B, int T,
                           int rows, hipStream_t s);

struct PosteriorParams {
    const float* x_t;
    const float* pred;
    const float* noise;  // may be nullptr for deterministic modes
    const float* coef;   // [B][8]
    float* x_s;
    int B;
    long per_sample;
    int mode, objective;
    float clip;  // < 0: no clamping
};
hipError_t launch_posterior(const PosteriorParams& p, hipStream_t s);
// RePaint (reference continuous_time.py:169-190,287-303): forward re-noising and known/unknown blend
hipError_t launch_repaint_blend(const float* known, const float* noise, const float* unknown, const float* mask,
                                const float* coef, float* out, int B, long per_sample, int channels, int mask_c,
                                hipStream_t s);
hipError_t launch_q_step(const float* x, const float* noise, const float* coef, float* out, int B, long per_sample,
                         hipStream_t s);

// (B,2,H,W) in [-1,1] -> (B,5,H,W) [depth, x, y, z, reflectance]  (reference sample_and_save.py:52-57)
hipError_t launch_lidar_postprocess(const float* x, const float* angles, float* y, int B, int H, int W,
                                    float min_depth, float max_depth, hipStream_t s, int depth_format = 0);  // 0 log_depth, 1 inverse_depth, 2 depth

}  // namespace r2dm

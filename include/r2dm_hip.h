/* libr2dm_hip.so -- C ABI of the MI355X (gfx950) R2DM sampling engine.
 *
 * The reference (kazuto1011/r2dm) has no native code and no FFI; its only seam on the sampling path
 * is the Python call  prediction = self.model(x_t, cond)  plus the elementwise posterior update
 * around it.  This header is that seam expressed as a C ABI: what a ctypes / cffi / pybind stub in
 * the reference would bind to run the per-step hot path on hand-written HIP kernels.
 * Every entry point names the reference code it replaces (paths relative to the reference repo).
 *
 * Conventions
 *   - all tensors are fp32, NCHW, contiguous, in DEVICE memory owned by the caller (PyTorch);
 *   - the library never allocates device memory, never synchronises and enqueues everything on the
 *     `stream` it is given (a hipStream_t passed as void*; NULL = default stream);
 *   - a handle is bound to the current device of the creating thread and used from one host thread;
 *   - every function returns 0 on success and a non-zero code on failure; r2dm_last_error() then
 *     returns a thread-local human-readable message.
 */
#ifndef R2DM_HIP_H
#define R2DM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct r2dm_handle r2dm_handle;

/* Geometry of one EfficientUNet instance: the constructor arguments of
 * models/efficient_unet.py:194-211 that shape tensors (ring=True, dropout=0 always). */
typedef struct r2dm_config {
    int32_t in_channels;            /* 2: range + reflectance */
    int32_t out_channels;
    int32_t height, width;          /* resolution, e.g. 64 x 1024 */
    int32_t base_channels;          /* 64 */
    int32_t temb_channels;          /* 4 * base */
    int32_t channel_multiplier[4];  /* (1,2,4,8) */
    int32_t num_residual_blocks[4]; /* (3,3,3,3) */
    int32_t gn_num_groups;          /* 8 */
    float gn_eps;                   /* 1e-6 */
    int32_t attn_num_heads;         /* 8 */
    int32_t coord_channels;         /* channels of the constant coordinate encoding (32 @64x1024; 0 = none) */
    int32_t max_batch;              /* layer tilings are chosen for batches up to this size */
} r2dm_config;

/* One state-dict tensor the engine consumes, in the reference's naming (SURVEY.md appendix A.3,
 * without the "model." prefix), plus two host-precomputed constants:
 *   "__cenc"      (coord_channels, H, W)  FourierFeatures(coords)        models/encoding.py:141-146
 *   "__sin_freqs" (base_channels/2,)      sinusoid frequency table       models/ops.py:22-23          */
typedef struct r2dm_tensor_info {
    const char* key;
    int64_t numel; /* number of fp32 elements the caller must supply */
} r2dm_tensor_info;

const char* r2dm_last_error(void);
const char* r2dm_version(void);

/* -- lifetime ------------------------------------------------------------------------------- */
int r2dm_create(r2dm_handle** out, const r2dm_config* cfg);
void r2dm_destroy(r2dm_handle* h);

/* -- weights: replaces nn.Module.load_state_dict + .to(device) for the denoiser
 *    (utils/inference.py:80-83).  The caller allocates r2dm_blob_bytes() of device memory, binds it,
 *    then feeds every tensor listed by r2dm_tensor_at(); the engine repacks each into its
 *    kernel-friendly layout inside the blob (HIP kernels, asynchronous).  A blob filled on one GPU
 *    can be broadcast (RCCL) and bound on another GPU with an identical config: no per-rank load. */
int64_t r2dm_num_tensors(const r2dm_handle* h);
int r2dm_tensor_at(const r2dm_handle* h, int64_t index, r2dm_tensor_info* out);
size_t r2dm_blob_bytes(const r2dm_handle* h);

/* Fingerprint of the blob's LAYOUT as this handle planned it: every slot's key, offset and packing (algorithm, channel / pixel tile of
 * each convolution's packings -- they depend on max_batch, the device's CU count and experiment switches, not only on the
 * configuration) and the blob size.  A blob filled through one handle may be bound by another only if the two agree (the Python
 * wrapper compares them before it adopts a broadcast blob: a mismatch would run silently wrong convolutions -- ADVICE round 4). */
uint64_t r2dm_blob_layout_hash(const r2dm_handle* h);
int r2dm_bind_blob(r2dm_handle* h, void* dev_blob, size_t bytes);
int r2dm_load_tensor(r2dm_handle* h, int64_t index, const float* dev_src, int64_t numel, void* stream);

/* -- the denoiser: replaces EfficientUNet.forward (models/efficient_unet.py:269-295), called from
 *    p_step at models/diffusion/continuous_time.py:207 and discrete_time.py:139.
 *    x (B,in_channels,H,W), cond (B,) as float, out (B,out_channels,H,W). */
size_t r2dm_workspace_bytes(const r2dm_handle* h, int32_t batch);
int r2dm_unet_forward(r2dm_handle* h, const float* x, const float* cond, float* out, int32_t batch,
                      void* workspace, size_t workspace_bytes, void* stream);

/* -- arithmetic of the convolutions / attention core on the 16-bit matrix pipe (tensors and accumulation are fp32 in every mode):
 *      pieces = 2 (default, parity mode): every fp32 operand v is split to 22 bits, v = h + 2^-11 l with h = RNE_f16(v),
 *                 l = RNE_f16(2^11 (v - h)); a product is xh wh + 2^-11 (xh wl + xl wh) -- 3 MFMA products, the l x l term
 *                 (2^-22 relative) dropped, two fp32 accumulators: conv_f16x2.hip (3x3), proj_f16x2.hip (1x1), attention.hip.
 *                 fp32-class BY MEASUREMENT (error vs fp64 at or below an fp32 FMA chain's), not bit-exact fp32 operands.
 *                 Operands pass through fp16 (|v| < 65504): weights are pre-scaled per layer by a power of two (max|w| into
 *                 [2^9, 2^10); undone exactly in the epilogue); a GroupNorm-normalised operand is bounded from the DATA
 *                 (|a| M + |d| with M >= max|x|: the square root of the largest statistics-slot energy, or the maximum the
 *                 streaming statistics kernel recorded); a raw operand's producer records max|output| (conv epilogues, fir_up2);
 *      pieces = 3 (parity mode, fp32 operand range): three bf16 pieces = exact 24-bit operands, 6 products (conv_bf16x3.hip)
 *                 for the 3x3 convolutions, the fp32-input MFMA for the 1x1 convolutions and the attention core;
 *      pieces = 1 (REDUCED precision, bulk sampling -- the reference's fp16 autocast mode, sample_and_save.py:70,
 *                 utils/option.py:49): the kernels of pieces = 2 with the h piece alone: one fp16 product per MAC.
 *    With pieces = 1 / 2 r2dm_check_range reports a violation of the fp16 operand range.
 *    h == NULL sets the mode of the single-kernel entries r2dm_conv2d_ring / r2dm_attention. */
int r2dm_set_conv_pieces(r2dm_handle* h, int32_t pieces);

/* Waits for `stream` and returns non-zero (r2dm_last_error explains) if, since the last call, a forward of `h` ran an
 * f16x2 convolution on operands that may have left the fp16 range -- its output is then not valid.  The Python wrapper
 * calls this after every stand-alone forward and once at the end of a sampling loop.  The recorded bounds are running maxima: a clean check leaves
 * them (nothing can have tripped unnoticed), a reported trip resets them. */
int r2dm_check_range(r2dm_handle* h, void* stream);

/* Per-site view of the guard (round 6; python -m r2dm_amd.check): a SITE is one guarded producer of a forward in walk order -- a GroupNorm's
 * output bound, or the recorded max|output| of a tensor that the next fp16-operand kernel reads raw (conv_f16x2 records the root of the largest sum of squares of four
 * neighbouring pixels: an upper bound of max|output|, at most twice it).  r2dm_range_sites copies the bounds
 * the LAST r2dm_check_range read (up to `cap`; *n = number of sites of the last forward, site 0 = shared slot) -- call r2dm_check_range
 * first; r2dm_range_site_name labels site k ("d_block1.residual_blocks.0.conv1: GroupNorm output bound ..."; valid until the next forward).
 * A bound >= 65504 is a trip.  Replaces nothing upstream: the reference runs fp32 / autocast and has no operand range to guard. */
int r2dm_range_sites(r2dm_handle* h, float* bounds, int32_t cap, int32_t* n);
const char* r2dm_range_site_name(r2dm_handle* h, int32_t k);

/* Test hook: raises the recorded operand bound to at least `bound` on `stream`, as a convolution input of that size would have
 * (tests/test_hip_range.py simulates a guard trip at a chosen step of a sampling loop with it). */
int r2dm_test_raise_range_bound(r2dm_handle* h, float bound, void* stream);

/* -- measurement aid (bench.py): when enabled, every convolution launch of a forward -- conv_f16x2_kernel (54 3x3 launches in
 *    the default mode; conv_bf16x3_* with pieces = 3), proj_f16x2_kernel (8 1x1 launches; conv_mfma_kernel with pieces = 3) and
 *    the two direct kernels (in_conv, out_conv), 64 launches per forward -- is bracketed by hipEvents recorded on the caller's stream.  r2dm_profile_read waits for them and returns the summed kernel time, the summed
 *    ALGORITHMIC flops (2*B*Cout*Cin*k*k*H*W per launch) and the number of launches, then resets. */
int r2dm_profile_enable(r2dm_handle* h, int32_t on);
int r2dm_profile_read(r2dm_handle* h, double* conv_ms, double* conv_flop, int64_t* launches);
/* same, per kernel class: [0] conv_f16x2_kernel, [1] conv_bf16x3_*, [2] conv_mfma_kernel / conv_direct_kernel */
int r2dm_profile_read_classes(r2dm_handle* h, double* ms3, double* flop3, int64_t* launches3);
/* what a bracketing event pair adds to the kernel between its events, measured on `stream` (medians of 33): the pair around nothing (marker processing)
 * and around an empty kernel (marker processing + one dispatch + that kernel's own microsecond); bench.py prints both next to the per-launch figures,
 * which rocprofv3's kernel durations (profiles/) do not contain. */
int r2dm_profile_event_overhead(r2dm_handle* h, void* stream, double* empty_pair_us, double* empty_kernel_pair_us);

/* -- posterior update: replaces the elementwise tail of p_step
 *    (continuous_time.py:208-229, discrete_time.py:140-177).  coef is (B,8) host-computed scalars,
 *    see r2dm_amd/diffusion.py for the slot meaning per mode.
 *    mode: 0 continuous DDPM, 1 continuous DDIM, 2 discrete DDPM, 3 discrete DDIM (eta=0, noise may
 *    be NULL), 4 discrete DDIM with noise.  objective: 0 eps, 1 v, 2 x_0.  clip < 0 disables clamping. */
int r2dm_posterior_step(const float* x_t, const float* prediction, const float* noise, const float* coef,
                        float* x_s, int32_t batch, int64_t per_sample, int32_t mode, int32_t objective,
                        float clip, void* stream);

/* -- RePaint completion (models/diffusion/continuous_time.py:169-190 q_step_from_x_0 / q_step, :287-303 the
 *    known/unknown blend inside repaint()).  coef is (B,2) host-computed scalars.
 *    blend:  out = mask * (known*alpha + noise*sigma) + (1 - mask) * unknown,  coef = (alpha, sigma) at step s;
 *            mask has `mask_channels` channels (1 = broadcast over the `channels` planes of a sample).
 *    q_step: x_t = x_s * a_ts + std * noise,  coef = (alpha_t/alpha_s, sqrt(sigma_t^2 - a_ts^2 sigma_s^2)). */
int r2dm_repaint_blend(const float* known, const float* noise, const float* unknown, const float* mask,
                       const float* coef, float* out, int32_t batch, int64_t per_sample, int32_t channels,
                       int32_t mask_channels, void* stream);
int r2dm_q_step(const float* x_s, const float* noise, const float* coef, float* x_t, int32_t batch, int64_t per_sample,
                void* stream);

/* -- sample post-processing: LiDARUtility.denormalize/revert_depth/to_xyz + concat
 *    (utils/lidar.py:49-61,98-120; sample_and_save.py:52-57).  x (B,2,H,W) in [-1,1],
 *    ray_angles (2,H,W) [elevation, azimuth] in rad, out (B,5,H,W) = depth,x,y,z,reflectance. */
int r2dm_lidar_postprocess(const float* x, const float* ray_angles, float* out, int32_t batch, int32_t height,
                           int32_t width, float min_depth, float max_depth, void* stream);
/* ... with the checkpoint's depth coding (LiDARUtility.revert_depth, utils/lidar.py:95-112):
 *    depth_format 0 = "log_depth" (the entry above), 1 = "inverse_depth", 2 = "depth". */
int r2dm_lidar_postprocess_fmt(const float* x, const float* ray_angles, float* out, int32_t batch, int32_t height,
                               int32_t width, float min_depth, float max_depth, int32_t depth_format, void* stream);

/* -- single kernels, exported for per-op parity tests against the oracle -------------------- */
/* ops.Conv2d(ring) 3x3 / 1x1 (models/ops.py:149-173) with optional fused GroupNorm-affine(+SiLU)
 * prologue (aff: (B,Cin,2) or NULL; prologue 0 none, 1 affine, 2 affine+SiLU) and optional
 * residual-add + scale epilogue.  w is OIHW; w_packed is caller scratch of r2dm_conv_packed_elems(). */
int64_t r2dm_conv_packed_elems(int32_t cout, int32_t cin, int32_t ksize, int32_t batch, int32_t height,
                               int32_t width);
int r2dm_conv2d_ring(const float* x, const float* w, const float* bias, float* w_packed, const float* aff,
                     int32_t prologue, const float* residual, const float* scale, float* y, int32_t batch,
                     int32_t cin, int32_t cout, int32_t height, int32_t width, int32_t ksize, void* stream);
/* nn.GroupNorm / AdaGN statistics folded to y = x*a + d (models/efficient_unet.py:72; ops.py:176-200).
 * gamma/beta (C,) or NULL; ada (B,2C) [scale|shift] or NULL; partial: scratch of
 * r2dm_group_norm_scratch_bytes(); aff out (B,C,2); stats out (B,G,2) mean/rstd or NULL. */
size_t r2dm_group_norm_scratch_bytes(int32_t batch, int32_t groups);
int r2dm_group_norm_affine(const float* x, const float* gamma, const float* beta, const float* ada, void* scratch,
                           float* aff, float* stats, int32_t batch, int32_t channels, int32_t height,
                           int32_t width, int32_t groups, float eps, void* stream);
int r2dm_affine_act(const float* x, const float* aff, float* y, int32_t batch, int32_t channels, int64_t hw,
                    int32_t silu, void* stream);
/* ops.Resample(down=2) / (up=2) (models/ops.py:52-146) */
int r2dm_fir_down2(const float* x, float* y, int32_t batch, int32_t channels, int32_t height, int32_t width,
                   void* stream);
int r2dm_fir_up2(const float* x, float* y, int32_t batch, int32_t channels, int32_t height, int32_t width,
                 void* stream);
/* ops.Resample(down=2) that also leaves the GroupNorm statistics of its OUTPUT (nn.GroupNorm(groups, C) behind it:
 * models/efficient_unet.py:95-97 behind :135) in the convolution epilogues' slot grid -- stat: (B, groups, slots, 2) doubles
 * [sum, sum of squares], slots = r2dm_fir_down2_stat_slots() (0: geometry not supported, call r2dm_fir_down2); every slot is
 * written, the sums of a (sample, group) over ALL its slots are the group's moments.  Kernel-level test hook of the engine's path. */
int32_t r2dm_fir_down2_stat_slots(int32_t channels, int32_t groups, int32_t height, int32_t width);
int r2dm_fir_down2_stats(const float* x, float* y, double* stat, int32_t batch, int32_t channels, int32_t groups,
                         int32_t height, int32_t width, void* stream);
/* attention core of nn.MultiheadAttention on channel-major qkv (B,3C,N) -> (B,C,N)
 * (models/efficient_unet.py:34-38,46) */
int r2dm_attention(const float* qkv, float* out, int32_t batch, int32_t channels, int32_t heads, int32_t tokens,
                   void* stream);
/* time embedding: cond (B,) -> SiLU(time_embedding(cond)) (B,T)  (models/efficient_unet.py:232-237);
 * hidden: (B,T) scratch */
int r2dm_time_embedding(const float* cond, const float* freqs, const float* w1, const float* b1, const float* w2,
                        const float* b2, float* act, float* hidden, int32_t batch, int32_t base, int32_t temb,
                        void* stream);

#ifdef __cplusplus
}
#endif
#endif /* R2DM_HIP_H */

/* mipnerf_hip.h -- C ABI of the MI355X-native Mip-NeRF volume-rendering hot path.
 *
 * Shared library: mipnerf_pl_amd/csrc/libmipnerf_hip.so (hipcc --offload-arch=gfx950).
 * The reference (hjxwhy/mipnerf_pl) has no FFI of its own -- its boundary is the Python
 * class contract `MipNerf.forward(rays, randomized, white_bkgd)` (models/mip_nerf.py:172-248)
 * built from the free functions of models/mip.py.  Each entry point below replaces one of
 * those functions (cited per prototype); `mipnerf_forward` replaces the whole level loop.
 * INTEGRATION.md shows the ctypes binding a reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc / torch.cuda tensor .data_ptr()) unless
 *     the parameter name ends in `_host`; float32, row-major, contiguous;
 *   - `stream` is a hipStream_t passed as void* (0 = default stream); all work is
 *     enqueued asynchronously on it, no entry point synchronises or allocates, except
 *     mipnerf_create / mipnerf_destroy (hipMalloc / hipFree of the packed-weight buffers);
 *   - return value: 0 = MIPNERF_OK, otherwise an error code; mipnerf_last_error() returns
 *     a thread-local message.  The Python host turns codes into RuntimeError /
 *     NotImplementedError (the reference raises NotImplementedError for unsupported
 *     enum values: mip_nerf.py:50,70,165,170, mip.py:98);
 *   - inputs are never written; outputs never alias inputs (mip.py:184 mutates its
 *     argument in place -- this library does not).
 */
#ifndef MIPNERF_HIP_H
#define MIPNERF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIPNERF_ABI_VERSION 6

enum {
    MIPNERF_OK = 0,
    MIPNERF_E_INVALID = 1,      /* bad argument (null pointer, size <= 0, N too large ...) */
    MIPNERF_E_UNSUPPORTED = 2,  /* NotImplementedError in the reference, or an MLP shape the
                                   compiled kernels do not cover */
    MIPNERF_E_HIP = 3,          /* a HIP runtime call failed (message has hipGetErrorString) */
    MIPNERF_E_WORKSPACE = 4     /* workspace too small */
};

/* compute precision of the MLP (the rest of the path is always fp32) */
enum {
    MIPNERF_PREC_FP32 = 0, /* v_mfma_f32_32x32x2_f32, exact fp32 products (parity mode)   */
    MIPNERF_PREC_BF16 = 1, /* v_mfma_f32_32x32x16_bf16, bf16 operands / fp32 accumulate    */
    MIPNERF_OUT_BF16_FRAGMENTS = 2 /* out_dtype of mipnerf_cast_ipe_360 only: bf16 in the MFMA B-operand fragment layout  */
                                   /* [wave tile of 32 samples][k-step][64 lanes][8], whole 256-sample tiles (the buffer    */
                                   /* holds ceil(M / 256) * 256 rows) -- what the two-kernel MLP form reads fastest         */
};

/* flags */
enum {
    MIPNERF_FLAG_WHITE_BKGD = 1, /* volumetric_rendering(white_bkgd=True), mip.py:399-400  */
    MIPNERF_FLAG_DISPARITY = 2   /* sample linearly in disparity, mip.py:149-150           */
};

/* Mirrors the keyword arguments of MipNerf.__init__ (models/mip_nerf.py:117-141) that
 * change the arithmetic.  ray_shape is always 'cone' ('cylinder' raises in the reference). */
typedef struct mipnerf_config {
    int32_t num_samples;         /* nerf.num_samples, N (<= MIPNERF_MAX_SAMPLES)           */
    int32_t num_levels;          /* nerf.num_levels (1 or 2)                                */
    int32_t min_deg_point;       /* 0                                                       */
    int32_t max_deg_point;       /* 16                                                      */
    int32_t deg_view;            /* 4                                                       */
    int32_t use_viewdirs;        /* 1                                                       */
    int32_t disparity;           /* 0                                                       */
    int32_t disable_integration; /* 0 ; 1 => covariances zeroed before the IPE (PE)         */
    int32_t net_depth;           /* 8                                                       */
    int32_t net_width;           /* 256                                                     */
    int32_t net_depth_condition; /* 1                                                       */
    int32_t net_width_condition; /* 128                                                     */
    int32_t skip_index;          /* 4                                                       */
    int32_t num_rgb_channels;    /* 3                                                       */
    int32_t num_density_channels;/* 1                                                       */
    float resample_padding;      /* 0.01                                                    */
    float density_bias;          /* -1                                                      */
    float rgb_padding;           /* 0.001                                                   */
    float density_noise;         /* 0 ; std-dev of the noise added to raw density when a    */
                                 /* density_randn tensor is given (mip_nerf.py:232-233)     */
    int32_t unbounded;           /* 0 ; 1 => the unbounded-scene (mip-NeRF 360) path: fence posts uniform in inverse depth,  */
                                 /* contracted full-covariance Gaussians, off-axis IPE with 42 features per degree (what     */
                                 /* models/mip.py:106-124, 292-319, 424-447 aim at).  fp32 or bf16 (the 672-wide encoding    */
                                 /* runs as k_pre_gemm + a trunk kernel, csrc/gen_pre_gemm.py): mipnerf_forward,             */
                                 /* mipnerf_mlp_forward and the per-stage training entry points (mipnerf_mlp_forward_train / */
                                 /* _dgrad / _backward) and, in bf16, mipnerf_train_step                                     */
} mipnerf_config;

#define MIPNERF_MAX_SAMPLES 1024
#define MIPNERF_NUM_PARAM_TENSORS 24 /* 2 x (8 trunk + density + extra + 1 view + color)   */

/* The 7 fields of the reference `Rays` namedtuple (datasets/datasets.py:13-16), SoA. */
typedef struct mipnerf_rays {
    const float* origins;    /* [B,3] */
    const float* directions; /* [B,3] un-normalised */
    const float* viewdirs;   /* [B,3] unit          */
    const float* radii;      /* [B,1] */
    const float* lossmult;   /* [B,1] (unused by forward) */
    const float* near;       /* [B,1] */
    const float* far;        /* [B,1] */
} mipnerf_rays;

/* One level of the list returned by MipNerf.forward (mip_nerf.py:246). */
typedef struct mipnerf_level_out {
    float* comp_rgb;  /* [B,3]   */
    float* distance;  /* [B]     */
    float* acc;       /* [B]     */
    float* weights;   /* [B,N]   */
    float* t_samples; /* [B,N+1] */
} mipnerf_level_out;

/* Output of mipnerf_generate_rays: the same 7 fields, writable. */
typedef struct mipnerf_rays_out {
    float* origins; float* directions; float* viewdirs; float* radii; float* lossmult; float* near; float* far;
} mipnerf_rays_out;

/* One camera for mipnerf_generate_rays: 32 floats =
 * c2w[3][4] | pix2cam[3][3] | width, height, near, far, lossmult, mode, focal | 4 pad.
 * mode 0: Blender pinhole formula with `focal` (datasets.py:226-228); mode 1: pix2cam matrix (datasets.py:125-131). */
#define MIPNERF_CAMERA_FLOATS 32

typedef struct mipnerf_ctx mipnerf_ctx;

const char* mipnerf_last_error(void);
int mipnerf_abi_version(void);

/* ---- context: configuration + packed weights ------------------------------------------ */
/* MipNerf.__init__ (mip_nerf.py:117-170).  Fails with MIPNERF_E_UNSUPPORTED when the MLP
 * shape is not the one the MFMA kernels were generated for (see mipnerf_compiled_arch). */
int mipnerf_create(const mipnerf_config* cfg, mipnerf_ctx** out);
int mipnerf_destroy(mipnerf_ctx* ctx);
/* Writes the MLP shape the library was compiled for into *cfg (other fields defaulted). */
int mipnerf_compiled_arch(mipnerf_config* cfg);
/* The library carries tables (+ a bf16 inference kernel) for a fixed list of MLP shapes ("variants", csrc/gen_mlp_bf16.py
 * VARIANTS); variant 0 is the shipped shape; bf16 TRAINING kernels + tables are generated per variant as well
 * (*has_bf16_training; a variant without them would train in fp32).
 * mipnerf_create picks the variant that matches cfg or fails with MIPNERF_E_UNSUPPORTED listing them. */
int mipnerf_num_variants(void);
int mipnerf_variant_arch(int variant, mipnerf_config* cfg, int* has_bf16_training);

/* Number of parameter tensors of the context's architecture = 2 x (net_depth + 3 + net_depth_condition) in the reference
 * MLP's state_dict order (MIPNERF_NUM_PARAM_TENSORS = 24 for the shipped 8-layer shape). */
int mipnerf_num_param_tensors(const mipnerf_ctx* ctx);
/* (Re)pack the fp32 master parameters into the MFMA operand streams (bf16 fragment stream,
 * fp32 fragment stream, bias tables).  `params` is a HOST array of
 * mipnerf_num_param_tensors(ctx) device pointers (24 for the shipped shape) in state_dict order of the reference MLP
 * (mip_nerf.py:19-73): layers.{0..7}.0.{weight,bias}, density_layer.{weight,bias},
 * extra_layer.{weight,bias}, view_layers.0.0.{weight,bias}, color_layer.{weight,bias}.
 * Call after every optimizer step / load_state_dict. */
int mipnerf_set_params(mipnerf_ctx* ctx, const float* const* params_host, void* stream);

/* ---- the whole hot path: MipNerf.forward (mip_nerf.py:172-248) ------------------------ */
size_t mipnerf_workspace_bytes(const mipnerf_ctx* ctx, int64_t num_rays);
/* t_rand [B,N+1] / u_rand [B,N+1]: uniform [0,1) noise replacing torch.rand (mip.py:159)
 * and uniform_ (mip.py:201); both NULL <=> randomized=False.  density_randn (NULL = none):
 * standard-normal draws [num_levels, B, N] replacing torch.randn of mip_nerf.py:232-233; the raw
 * density of level l becomes raw + cfg.density_noise * density_randn[l] before the softplus.
 * out[level], level < num_levels. */
int mipnerf_forward(mipnerf_ctx* ctx, int64_t num_rays, const mipnerf_rays* rays,
                    const float* t_rand, const float* u_rand, const float* density_randn,
                    uint32_t flags, int precision,
                    void* workspace, size_t workspace_bytes, const mipnerf_level_out* out,
                    void* stream);

/* ---- per-stage entry points (each parity-tested alone) --------------------------------- */
/* sample_along_rays (mip.py:127-165), t part: t_samples [B,N+1]. */
int mipnerf_sample_along_rays(int64_t num_rays, int32_t num_samples, const float* near,
                              const float* far, const float* t_rand, int32_t disparity,
                              float* t_samples, void* stream);
/* cast_rays (mip.py:81-103) + conical_frustum_to_gaussian (50-78) + lift_gaussian (22-36):
 * means, covs [B,N,3] (either may be NULL). */
int mipnerf_cast_rays(int64_t num_rays, int32_t num_samples, const float* t_samples,
                      const float* origins, const float* directions, const float* radii,
                      float* means, float* covs, void* stream);
/* cast_rays + integrated_pos_enc (mip.py:322-350) fused: enc [B*N, 6*(max_deg-min_deg)];
 * out_dtype MIPNERF_PREC_FP32 (float) or MIPNERF_PREC_BF16 (bfloat16 RNE). */
int mipnerf_cast_ipe(int64_t num_rays, int32_t num_samples, int32_t min_deg, int32_t max_deg,
                     int32_t disable_integration, const float* t_samples, const float* origins,
                     const float* directions, const float* radii, void* enc, int out_dtype,
                     void* stream);
/* integrated_pos_enc (mip.py:322-350, diagonal) on given means / covs [M,3]. */
int mipnerf_integrated_pos_enc(int64_t num_points, int32_t min_deg, int32_t max_deg,
                               const float* means, const float* covs, void* enc, int out_dtype,
                               void* stream);
/* pos_enc(viewdirs, 0, deg_view, append_identity=True) (mip.py:353-363): [B, 3+6*deg_view]
 * written with row stride `ld` elements (ld >= 3+6*deg; pad columns zeroed). */
int mipnerf_pos_enc(int64_t num_rays, int32_t deg_view, const float* viewdirs, void* out,
                    int32_t ld, int out_dtype, void* stream);
/* MLP.forward (mip_nerf.py:75-111) + activations (mip_nerf.py:236-238):
 * enc [M,xyz_dim] row-major (dtype = precision; xyz_dim = 96, or 672 for the unbounded-scene variant), viewenc [B,32]
 * (dtype = precision, ld 32), sample m belongs to ray m / num_samples.  rgb_sigma [M,4] = (r,g,b,sigma) after
 * sigmoid/padding and softplus(raw+bias); raw [M,4] = (raw_rgb, raw_density) or NULL.
 * bf16 on the unbounded-scene variant runs two kernels with 1.5 KiB of scratch per sample between them; this entry point has no
 * workspace argument, so the context keeps that buffer and GROWS it with hipMalloc when num_points exceeds every earlier call
 * (synchronises the stream, not capturable into a hipGraph at that moment; one stream at a time per context for this entry point on
 * that variant) -- mipnerf_forward uses the caller's workspace and has neither restriction. */
int mipnerf_mlp_forward(mipnerf_ctx* ctx, int64_t num_points, int32_t num_samples,
                        const void* enc, const void* viewenc, int precision, float* rgb_sigma,
                        float* raw, void* stream);
/* volumetric_rendering (mip.py:366-401). */
int mipnerf_volumetric_rendering(int64_t num_rays, int32_t num_samples, const float* rgb_sigma,
                                 const float* t_samples, const float* directions,
                                 int32_t white_bkgd, float* comp_rgb, float* distance, float* acc,
                                 float* weights, void* stream);
/* resample_along_rays (mip.py:232-280) t part: blur-pool + padding +
 * sorted_piecewise_constant_pdf (mip.py:168-229) with num_samples+1 draws. */
int mipnerf_resample_along_rays(int64_t num_rays, int32_t num_samples, const float* t_samples,
                                const float* weights, const float* u_rand, float resample_padding,
                                float* t_new, void* stream);
/* sorted_piecewise_constant_pdf alone (mip.py:168-229): bins [B,N+1], weights [B,N]
 * (not mutated), num_draws samples out [B,num_draws]. */
int mipnerf_sorted_piecewise_constant_pdf(int64_t num_rays, int32_t num_bins, const float* bins,
                                          const float* weights, int32_t num_draws,
                                          const float* u_rand, float* samples, void* stream);

/* ---- device-side ray generation (Blender._generate_rays datasets.py:214-263, Multicam._generate_rays :116-168):
 * ray i = pixel pix_idx[i] (row-major y*W + x; NULL = pixel i) of camera cam_idx[i] (NULL = camera 0) of the
 * `cameras` table [ncam][MIPNERF_CAMERA_FLOATS] (device memory).  Radii follow the reference: distance to the
 * neighbouring pixel along image rows, last row repeated, times 2/sqrt(12). */
int mipnerf_generate_rays(int64_t num_rays, const float* cameras, const int32_t* cam_idx,
                          const int32_t* pix_idx, const mipnerf_rays_out* out, void* stream);
/* The same with a camera table of DOUBLES and float64 arithmetic, results rounded to float32 once at the end: `RenderGen`
 * (render_video.py:29-112) forms its rays in float64 numpy from the float64 poses of create_spheric_poses and casts with .float()
 * (render_video.py:131); the radii are the norm of a DIFFERENCE of neighbouring directions (1e-3 of their size), which float32
 * arithmetic gets to 1e-4 relative only. */
int mipnerf_generate_rays_f64(int64_t num_rays, const double* cameras, const int32_t* cam_idx,
                              const int32_t* pix_idx, const mipnerf_rays_out* out, void* stream);

/* ---- unbounded scenes (mip-NeRF 360) --------------------------------------------------------------------------------
 * Correct versions of what the reference's dead code aims at (models/mip.py:106-124 sample_along_rays_360, :38-47 full
 * lift_gaussian, :424-447 contract / parameterization, :292-319 integrated_pos_enc_360); they follow Barron et al.,
 * "Mip-NeRF 360" (CVPR 2022), see csrc/raymath360.hpp.  Parity: against oracle/mipnerf360_oracle.py ("parity unpinned":
 * the reference code is broken and has no outputs to pin against).
 * sample_along_rays_360: fence posts uniform in normalised inverse depth, t = 1 / (s / far + (1 - s) / near); t_rand [B,N+1]
 *   (NULL = deterministic) jitters between midpoints in inverse-depth space.  Outputs t_inv, t_samples [B,N+1].
 * cast_ipe_360: t [B,N+1] -> conical-frustum Gaussians with FULL covariance -> scene contraction of mean and covariance
 *   (contracted != 0) -> off-axis IPE on 21 basis directions and frequencies 2^l, l in [min_deg, max_deg):
 *   enc [B*N, 2*21*(max_deg-min_deg)] (fp32 or bf16; feature = half*21L + l*21 + basis; or MIPNERF_OUT_BF16_FRAGMENTS).  means [B*N,3] / covs [B*N,3,3]
 *   (both or neither; may be the only outputs, enc = NULL) receive the (contracted) Gaussians. */
int mipnerf_sample_along_rays_360(int64_t num_rays, int32_t num_samples, const float* near, const float* far,
                                  const float* t_rand, float* t_inv, float* t_samples, void* stream);
int mipnerf_cast_ipe_360(int64_t num_rays, int32_t num_samples, int32_t min_deg, int32_t max_deg, int32_t contracted,
                         const float* t_samples, const float* origins, const float* directions, const float* radii,
                         void* enc, int out_dtype, float* means, float* covs, void* stream);
/* The same on GIVEN Gaussians, means [M,3] / covs [M,3,3]: contraction of mean and covariance (contracted != 0; `contract`,
 * `parameterization`, mip.py:424-447) into means_out / covs_out (covs_out may be NULL) and / or the off-axis encoding
 * enc [M, 2*21*(max_deg-min_deg)] (`integrated_pos_enc_360`, mip.py:292-319) of the (contracted) Gaussians. */
int mipnerf_gauss_360(int64_t num_points, int32_t min_deg, int32_t max_deg, int32_t contracted, const float* means,
                      const float* covs, void* enc, int out_dtype, float* means_out, float* covs_out, void* stream);

/* ---- evaluation metrics: eval_errors (utils/metrics.py:191-197) on one frame: pred, gt [H,W,3] fp32 in [0,1];
 * out[0] = PSNR (metrics.py:182-188), out[1] = mean SSIM, 11x11 Gaussian window sigma 1.5, zero padding
 * (metrics.py:44-126).  workspace: mipnerf_eval_workspace_floats(H, W) floats. */
int64_t mipnerf_eval_workspace_floats(int32_t height, int32_t width);
int mipnerf_eval_errors(int32_t height, int32_t width, const float* pred, const float* gt,
                        float* workspace, float* out_psnr_ssim, void* stream);

/* ---- training side ---------------------------------------------------------------------- */
/* activations (mip_nerf.py:236-238): raw [M,4] = (raw_rgb, raw_density) -> rgb_sigma [M,4];
 * density_randn [M] (NULL = none): raw_density + density_noise * density_randn first (mip_nerf.py:232-233). */
int mipnerf_activate(int64_t num_points, const float* raw, float rgb_padding, float density_bias,
                     const float* density_randn, float density_noise, float* rgb_sigma, void* stream);
/* backward of volumetric_rendering (mip.py:366-401) fused with the activation derivatives:
 * upstream g_rgb [B,3], g_dist [B], g_acc [B], g_w [B,N] (any may be NULL = zero) ->
 * d_raw [B*N,4] = dL/d(raw_rgb, raw_density).  rgb_sigma is the ACTIVATED forward tensor. */
int mipnerf_volumetric_rendering_bwd(int64_t num_rays, int32_t num_samples, const float* rgb_sigma,
                                     const float* t_samples, const float* directions,
                                     int32_t white_bkgd, const float* g_rgb, const float* g_dist,
                                     const float* g_acc, const float* g_w, float rgb_padding,
                                     float* d_raw, void* stream);
/* distloss (mip.py:8-20) without the [B,N,N] temporaries: ray_loss [B] (the reference value is
 * its mean over rays); if g_ray [B] is given also d_w [B,N] = g_ray[b] * d ray_loss[b] / d w. */
int mipnerf_distloss(int64_t num_rays, int32_t num_samples, const float* weights,
                     const float* t_samples, float* ray_loss, const float* g_ray, float* d_w,
                     void* stream);

/* ---- gradient through the resampler: MipNerf(stop_resample_grad=False) (mip.py:265-279, mip_nerf.py:204-214) ----
 * With the flag off the fine level's fence posts stay in the autograd graph; these are the extra backward pieces
 * (fp32 parity mode): compositing and distloss also return dL/dt_samples [B, N+1]; the MLP backward returns the gradient
 * w.r.t. its input encoding; cast_rays + integrated_pos_enc and the PDF resampler get their backward. */
int mipnerf_volumetric_rendering_bwd_t(int64_t num_rays, int32_t num_samples, const float* rgb_sigma,
                                       const float* t_samples, const float* directions, int32_t white_bkgd,
                                       const float* g_rgb, const float* g_dist, const float* g_acc, const float* g_w,
                                       float rgb_padding, float* d_raw, float* d_t, void* stream);
int mipnerf_distloss_bwd(int64_t num_rays, int32_t num_samples, const float* weights, const float* t_samples,
                         const float* g_ray, float* d_w, float* d_t, void* stream);
/* d_t [B, N+1] += dL/dt from d_enc [B*N, 6*(max_deg-min_deg)] fp32 (d_t zeroed or holding earlier contributions). */
int mipnerf_cast_ipe_bwd(int64_t num_rays, int32_t num_samples, int32_t min_deg, int32_t max_deg,
                         int32_t disable_integration, const float* t_samples, const float* origins,
                         const float* directions, const float* radii, const float* d_enc, float* d_t, void* stream);
/* d_weights [B, N] of mipnerf_resample_along_rays for d_t_new [B, N+1]; same u_rand (or NULL) as the forward. */
int mipnerf_resample_along_rays_bwd(int64_t num_rays, int32_t num_samples, const float* t_samples, const float* weights,
                                    const float* u_rand, float resample_padding, const float* d_t_new,
                                    float* d_weights, void* stream);

/* ---- native MLP training step (bf16 MFMA kernels; what torch autograd does to mip_nerf.py:75-111) ----
 * mipnerf_mlp_forward_train = mipnerf_mlp_forward (bf16) that also saves, per 32-sample wave tile, the
 * transposed activations of every layer input (`act`) and the ReLU bit masks (`masks`).
 * mipnerf_mlp_backward: d_raw [M,4] = dL/d(raw_rgb, raw_density) -> grad_flat [612,740] fp32, the gradients
 * of the 24 parameter tensors concatenated in state_dict order (accumulate = 0: overwritten; 1: added to).  `delta` and
 * `partials` are scratch.  Buffer sizes for M samples come from mipnerf_mlp_train_sizes. */
int mipnerf_mlp_train_sizes(const mipnerf_ctx* ctx, int64_t num_points, size_t* act_bytes,
                            size_t* mask_bytes, size_t* delta_bytes, size_t* partial_bytes);
/* `enc`: bf16, row-major [num_points, xyz_dim].  LIFETIME (two-kernel variants, i.e. contexts with unbounded = 1): their weight-gradient
 * jobs read the ENCODING ITSELF -- `act` ends in a record of the `enc` pointer -- so `enc` must stay allocated and unmodified until the
 * mipnerf_mlp_backward / mipnerf_mlp_wgrad call that consumes this `act` has been issued on the same stream; the standard shapes transpose
 * their 96 features into `act` and do not look at `enc` again. */
int mipnerf_mlp_forward_train(mipnerf_ctx* ctx, int64_t num_points, int32_t num_samples,
                              const void* enc, const void* viewenc, float* rgb_sigma, float* raw,
                              void* act, void* masks, void* stream);
/* The same with `enc` in the MFMA-fragment layout mipnerf_cast_ipe_360 writes for MIPNERF_OUT_BF16_FRAGMENTS (whole 256-sample tiles;
 * k_pre_gemm and the weight-gradient jobs read it lane-linearly).  Two-kernel variants only: MIPNERF_E_UNSUPPORTED for every other
 * context (a fragment buffer must never be read as rows).  Same lifetime rule for `enc`.  (ABI 6: replaces the per-context option 6.) */
int mipnerf_mlp_forward_train_fragments(mipnerf_ctx* ctx, int64_t num_points, int32_t num_samples,
                                        const void* enc, const void* viewenc, float* rgb_sigma, float* raw,
                                        void* act, void* masks, void* stream);
int mipnerf_mlp_backward(mipnerf_ctx* ctx, int64_t num_points, const float* d_raw, const void* act,
                         const void* masks, void* delta, float* partials, float* grad_flat,
                         int32_t accumulate, void* stream);
/* The two halves of mipnerf_mlp_backward, separately testable / timeable: delta chain (writes `delta`), and
 * weight gradients (reads act + delta; grad_flat may be NULL = leave the fp32 partials unreduced). */
int mipnerf_mlp_dgrad(mipnerf_ctx* ctx, int64_t num_points, const float* d_raw, const void* masks,
                      void* delta, void* stream);
int mipnerf_mlp_wgrad(mipnerf_ctx* ctx, int64_t num_points, const void* act, const void* delta,
                      float* partials, float* grad_flat, int32_t accumulate, void* stream);
/* torch.optim.Adam.step() of nerf_system.py:71-72 (betas, eps as given; no weight decay / amsgrad) over ONE flat
 * buffer of n parameters: param, grad, exp_avg, exp_avg_sq [n] fp32; `step` = 1-based step count.  The hyper-parameters are the
 * doubles torch holds (ABI 4; floats before): bias corrections, lr / (1 - beta1^t) and 1 - beta are formed in double. */
int mipnerf_adam_step(int64_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                      double lr, double beta1, double beta2, double eps, int32_t step, void* stream);
/* Tuning: workgroups per weight-gradient job (HOST array, one entry per job of the context's architecture: 12 for the shipped MLP;
 * 0 skips a job, for timing only).  NULL restores the default (workgroups ~ the 2-KiB blocks a job moves per stage + 4, all CUs
 * handed out).  Changes partial_bytes of mipnerf_mlp_train_sizes; synchronises. */
int mipnerf_set_wgrad_splits(mipnerf_ctx* ctx, const int32_t* splits_host);

/* ---- parity-mode (fp32) MLP training: the fused fp32 forward also writes every layer output into `save`
 * (mipnerf_mlp_train_f32_bytes), the backward is dgrad / wgrad GEMMs on v_mfma_f32_32x32x2_f32 (exact fp32
 * products, fp32 accumulation).  enc [M,96] and viewenc [B,32] are fp32.  Correctness-first. */
size_t mipnerf_mlp_train_f32_bytes(const mipnerf_ctx* ctx, int64_t num_points, size_t* save_bytes,
                                   size_t* workspace_bytes);
int mipnerf_mlp_forward_train_f32(mipnerf_ctx* ctx, int64_t num_points, int32_t num_samples, const float* enc,
                                  const float* viewenc, float* rgb_sigma, float* raw, float* save,
                                  void* stream);
int mipnerf_mlp_backward_f32(mipnerf_ctx* ctx, int64_t num_points, int32_t num_samples, const float* d_raw,
                             const float* enc, const float* viewenc, const float* save, void* workspace,
                             float* grad_flat, int32_t accumulate, void* stream);
/* The same plus d_enc [num_points, xyz_dim] = dL/d(encoding) (needed only with stop_resample_grad=False). */
int mipnerf_mlp_backward_f32_enc(mipnerf_ctx* ctx, int64_t num_points, int32_t num_samples, const float* d_raw,
                                 const float* enc, const float* viewenc, const float* save, void* workspace,
                                 float* grad_flat, int32_t accumulate, float* d_enc, void* stream);

/* ---- optimiser step with the LR schedule on the device (graph-capturable: no per-step host scalars) ------------------
 * torch.optim.Adam(lr) + MipLRDecay of the reference (nerf_system.py:70-76, utils/lr_schedule.py:51-59).
 * *step_count (device int64, starts at 0) is incremented to t; the step runs with lr = MipLRDecay(last_epoch = t - 1)
 * (constant_lr > 0 overrides the schedule), bias corrections of step t, and grad * grad_scale (1 / world_size after a SUM
 * all-reduce).  hyper_out (device float[4]) receives lr, lr/(1-beta1^t), sqrt(1-beta2^t), grad_scale (the `lr` that
 * nerf_system.py:117 logs). */
typedef struct mipnerf_lr_schedule {
    double lr_init, lr_final, lr_delay_mult, constant_lr;
    int64_t max_steps, lr_delay_steps;
    double beta1, beta2, eps;
    float grad_scale;
    int32_t reserved;
} mipnerf_lr_schedule;
int mipnerf_adam_step_scheduled(int64_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                                const mipnerf_lr_schedule* schedule, int64_t* step_count, float* hyper_out,
                                void* stream);

/* ---- the whole training step (bf16): MipNeRFSystem.training_step (nerf_system.py:95-111) + loss.backward() --------
 * forward of all levels with saved activations, loss = cm (mse_c + dm dl_c) + mse_f + dm dl_f (cm = loss.coarse_loss_mult,
 * dm = 0.01, mse masked by rays.lossmult unless disable_multiscale_loss), backward of compositing / activations / MLP.
 * grad_flat [612,740] = d loss / d parameters in state_dict order (accumulate = 0 overwrites).  out_scalars [6] =
 * loss, mse_coarse, mse_fine, distloss_coarse, distloss_fine, psnr_fine.  `out` (may be NULL, or hold NULL fields)
 * receives copies of what MipNerf.forward returns.  No autograd graph, no allocation, one stream: graph-capturable.
 * Every variant with bf16 training kernels, including (round 5) the unbounded-scene model's two-kernel form: inverse-depth
 * fence posts, encodings as fragments, one weight-gradient launch per level; grad_flat then has that variant's numel. */
size_t mipnerf_train_workspace_bytes(const mipnerf_ctx* ctx, int64_t num_rays);
int mipnerf_train_step(mipnerf_ctx* ctx, int64_t num_rays, const mipnerf_rays* rays, const float* gt_rgb,
                       const float* t_rand, const float* u_rand, const float* density_randn, uint32_t flags,
                       float coarse_loss_mult,
                       float distloss_mult, int32_t disable_multiscale_loss, void* workspace,
                       size_t workspace_bytes, float* grad_flat, int32_t accumulate, float* out_scalars,
                       const mipnerf_level_out* out, void* stream);

/* ---- instrumentation ------------------------------------------------------------------ */
/* Times `iters` launches of the bf16 MLP kernel with hipEvents on `stream`; returns the
 * average milliseconds per launch in *ms (used by bench.py for roofline.achieved). */
int mipnerf_time_mlp(mipnerf_ctx* ctx, int64_t num_points, int32_t num_samples, const void* enc,
                     const void* viewenc, int precision, float* rgb_sigma, int iters, float* ms,
                     void* stream);
/* Hardware self-test of the MFMA fragment layouts and the LDS-DMA path the kernels rely
 * on; returns 0 when the device behaves as the kernels assume (message in
 * mipnerf_last_error() either way). */
int mipnerf_selftest(void* stream);
/* Tuning / debug knobs.  option 0: bf16 MLP weight staging (1 = global_load_lds ring
 * [default], 0 = register-staged ring, same schedule); option 1: persistent grid size of the
 * bf16 MLP kernel (default = number of CUs); option 2: 1 = record a HIP event pair around
 * every MLP launch issued by mipnerf_forward, 2 = around every weight-gradient launch (read with
 * mipnerf_mlp_launch_stats); option 3: 1 [default] = the bf16
 * MLP kernel of mipnerf_forward computes the integrated positional encoding itself (no [M,96] buffer, no k_cast_ipe
 * launch), 0 = separate mipnerf_cast_ipe + encoding buffer (same bits); option 4: 1 [default] = mipnerf_forward runs pos_enc + the
 * coarse fence posts as ONE launch and the coarse level's compositing + the fine level's resampling as ONE launch (N <= 128 or 192 < N <= 256; the
 * weights go from registers to the sampler's LDS row), 0 = one launch per stage (same bits); option 5: 1 [default] = fp32 inference (mipnerf_mlp_forward,
 * mipnerf_forward) runs the register-resident kernel k_mlp_f32r where one was generated for the architecture (widths <= 256), 0 = the LDS-resident
 * k_mlp_f32 (same function, another summation order: results agree to fp32 rounding); option 6 (ABI 6, round 6): 1 [default] = the bf16 forward of an
 * unbounded = 1 context (mipnerf_forward) runs as ONE MLP kernel per level -- layer 0 and the skip layer as k-step-major ops of the trunk kernel, the 672-wide encoding
 * streamed through a wave-private LDS ring --, 0 = k_pre_gemm + trunk kernel with their 1.5-KiB-per-sample hand-off through HBM (same bits). */
int mipnerf_set_option(mipnerf_ctx* ctx, int option, int value);
/* Sum of the elapsed times (ms) and the number of MLP launches recorded since the last call
 * (option 2); synchronises on the recorded events. */
int mipnerf_mlp_launch_stats(mipnerf_ctx* ctx, double* total_ms, int64_t* launches);
/* Host-only exports of the static plan tables (no GPU needed), used by the CPU tests to
 * prove the C++ plan expansion equals mipnerf_pl_amd/mlp_plan.py.  which: 0 = bf16 stream
 * pack table, 1 = bias table, 2 = fp32 stream pack table (flat parameter indices, -1 = 0),
 * 3 = dgrad (W^T) stream pack table, 4 = wgrad partial -> parameter index table, 5 = wgrad job table.
 * Return the element count; copy only when cap is large enough. */
int64_t mipnerf_debug_table_variant(int variant, int which, int32_t* out_host, int64_t cap);   /* which: 0 bf16 pack, 1 bias, 2 fp32 pack */
int64_t mipnerf_debug_table(int which, int32_t* out_host, int64_t cap);
int64_t mipnerf_debug_f32net(int32_t* out_host, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* MIPNERF_HIP_H */

// Internal launcher declarations shared by the .hip translation units and capi.hip.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mip {

struct RayInputs {          // fused path of the bf16 MLP kernels: they compute the integrated positional encoding themselves
    const float* t;         // [B, N+1] sample boundaries of this level
    const float* origins;
    const float* dirs;
    const float* radii;
    int min_deg;
    int disable_integration;
};

// ---- kernels_ray.hip ------------------------------------------------------------------------
hipError_t launch_sample_along_rays(int64_t B, int N, const float* nearp, const float* farp,
                                    const float* t_rand, int disparity, float* t_out, hipStream_t st);
hipError_t launch_cast_rays(int64_t B, int N, const float* t, const float* origins, const float* dirs,
                            const float* radii, float* means, float* covs, hipStream_t st);
hipError_t launch_cast_ipe(int64_t B, int N, int min_deg, int max_deg, int disable_integration,
                           const float* t, const float* origins, const float* dirs, const float* radii,
                           void* enc, bool bf16, hipStream_t st);
hipError_t launch_integrated_pos_enc(int64_t M, int min_deg, int max_deg, const float* means, const float* covs,
                                     void* enc, bool bf16, hipStream_t st);
hipError_t launch_pos_enc(int64_t B, int deg, const float* viewdirs, void* out, int ld, bool bf16, hipStream_t st);
hipError_t launch_composite_train(int64_t B, int N, const float* rgb_sigma, const float* t, const float* dirs, int white_bkgd,
                                  float* comp_rgb, float* distance, float* acc, float* weights, float* ray_loss, float g_const,
                                  float* d_w, const float* u_rand, float padding, float* t_new, hipStream_t st,
                                  const float* bins = nullptr);   // bins: what the sampler inverts over (nullptr = t; unbounded scenes: the inverse depths)
hipError_t launch_composite_resample(int64_t B, int N, const float* rgb_sigma, const float* t, const float* dirs, int white_bkgd,
                                     float* comp_rgb, float* distance, float* acc, float* weights, const float* bins,
                                     const float* u_rand, float padding, float* t_new, hipStream_t st);   // hipErrorInvalidValue: N > 1024
hipError_t launch_ray_prologue(int64_t B, int deg, const float* viewdirs, void* venc, int ld, bool venc_bf16, int N, const float* nearp,
                               const float* farp, const float* t_rand, int disparity, float* t_out, hipStream_t st);
hipError_t launch_volumetric_rendering(int64_t B, int N, const float* rgb_sigma, const float* t, const float* dirs,
                                       int white_bkgd, float* comp_rgb, float* distance, float* acc,
                                       float* weights, hipStream_t st);
hipError_t launch_piecewise_constant_pdf(int64_t B, int N, const float* bins, const float* weights, int n_draws,
                                         const float* u_rand, bool blur, float padding, float* out,
                                         hipStream_t st);

hipError_t launch_generate_rays(int64_t n, const float* cams, const int32_t* cam_idx, const int32_t* pix_idx,
                                float* origins, float* directions, float* viewdirs, float* radii, float* lossmult,
                                float* nearp, float* farp, hipStream_t st);
hipError_t launch_generate_rays_f64(int64_t n, const double* cams, const int32_t* cam_idx, const int32_t* pix_idx,
                                    float* origins, float* directions, float* viewdirs, float* radii, float* lossmult,
                                    float* nearp, float* farp, hipStream_t st);

// ---- kernels_360.hip (unbounded scenes: s-space sampling, contraction, off-axis IPE; raymath360.hpp) ----
hipError_t launch_reciprocal(int64_t n, const float* x, float* y, hipStream_t st);
hipError_t launch_sample_along_rays_360(int64_t B, int N, const float* nearp, const float* farp, const float* t_rand,
                                        float* t_inv, float* t, hipStream_t st);
hipError_t launch_cast_ipe_360(int64_t B, int N, int min_deg, int max_deg, int contracted, const float* t, const float* origins,
                               const float* dirs, const float* radii, void* enc, bool bf16, float* means, float* covs,
                               hipStream_t st, bool frag = false);   // frag: the MFMA B-operand fragment layout k_pre_gemm reads (bf16 only)

hipError_t launch_gauss_360(int64_t M, int min_deg, int max_deg, int contracted, const float* means, const float* covs, void* enc,
                            bool bf16, float* means_out, float* covs_out, hipStream_t st);

// ---- kernels_train.hip ------------------------------------------------------------------------
// dnoise (nullable): standard-normal draws [M]; the density pre-activation becomes raw + dnoise_scale * dnoise (mip_nerf.py:232-233)
hipError_t launch_activate(int64_t M, const float* raw, float rgb_padding, float density_bias, const float* dnoise,
                           float dnoise_scale, float* out, hipStream_t st);
hipError_t launch_volumetric_rendering_bwd(int64_t B, int N, const float* rgb_sigma, const float* t, const float* dirs,
                                           int white_bkgd, const float* g_rgb, const float* g_dist, const float* g_acc,
                                           const float* g_w, float rgb_padding, float* d_raw, hipStream_t st,
                                           float* d_t = nullptr);      // d_t [B, N+1]: gradient w.r.t. t_samples (stop_resample_grad=False)
hipError_t launch_distloss(int64_t B, int N, const float* weights, const float* t, float* ray_loss, const float* g_ray,
                           float g_const, float* d_w, hipStream_t st, float* d_t = nullptr);
hipError_t launch_loss_fused(int64_t B, int nlevels, const float* rgb0, const float* rgb1, const float* gt, const float* lossmult,
                             const float* ray_loss0, const float* ray_loss1, float coarse_mult, float dist_mult, float* g_rgb0,
                             float* g_rgb1, float* out, hipStream_t st);

hipError_t launch_adam_flat(int64_t n, float* p, const float* g, float* m, float* v, double lr, double beta1, double beta2,
                            double eps, int step, hipStream_t st);

struct LrSchedule {          // MipLRDecay (utils/lr_schedule.py:5-59) + Adam constants, evaluated on the device
    double lr_init, lr_final, lr_delay_mult, constant_lr;   // constant_lr > 0: no schedule
    int64_t max_steps, lr_delay_steps;
    double beta1, beta2, eps;
    float grad_scale;
    int pad;
};
hipError_t launch_adam_scheduled(int64_t n, float* p, const float* g, float* m, float* v, const LrSchedule& sc,
                                 int64_t* step_count, float* hyper, hipStream_t st);

// ---- mlp_bf16_gen.hip (generated) -------------------------------------------------------------
int mlp_bf16_lds_bytes();
// rays != nullptr: enc is ignored (may be null) and the encoding is computed in the kernel from `rays`
hipError_t launch_mlp_bf16(const void* stream_w, const float* bias_tab, const void* enc, const void* viewenc,
                           float* rgb_sigma, float* raw_out, int64_t M, int num_samples, float density_bias,
                           float rgb_padding, int grid_limit, bool dma, const RayInputs* rays, const float* dnoise,
                           float dnoise_scale, hipStream_t st);

// architecture variants (gen_mlp_bf16.VARIANTS): declared and tabulated in the generated mlp_variants_gen.hpp /
// mlp_train_variants_gen.hpp

// ---- mlp_bf16_trainfwd_gen.hip / mlp_bf16_dgrad_gen.hip (generated by gen_mlp_train.py) -------
int mlp_trainfwd_lds_bytes();
hipError_t launch_mlp_bf16_trainfwd(const void* stream_w, const float* bias_tab, const void* enc, const void* viewenc,
                                    float* rgb_sigma, float* raw_out, void* HT, void* masks, int64_t M, int num_samples,
                                    float density_bias, float rgb_padding, int grid_limit, const RayInputs* rays,
                                    const float* dnoise, float dnoise_scale, hipStream_t st);
int mlp_dgrad_lds_bytes();
hipError_t launch_mlp_bf16_dgrad(const void* stream_wT, const float* d_raw, const void* masks, void* GT, int64_t M,
                                 int grid_limit, hipStream_t st);

// ---- kernels_resample_grad.hip: the gradient through the resampler (stop_resample_grad=False, mip.py:265-279) -------------
// d_t[b, i] += dL/dt from dL/denc [M, 6 * ndeg] fp32 through integrated_pos_enc, lift_gaussian and conical_frustum_to_gaussian
// (d_t must be zero-initialised or hold earlier contributions; two commutative atomic adds per element)
hipError_t launch_cast_ipe_bwd(int64_t B, int N, int min_deg, int max_deg, int disable_integration, const float* t,
                               const float* origins, const float* dirs, const float* radii, const float* d_enc, float* d_t,
                               hipStream_t st);
// d_weights [B, N] = dL/dweights of resample_along_rays' t part (blur pool + padding + sorted_piecewise_constant_pdf) for
// dL/dt_new [B, N+1]; the same draws (u_rand or the deterministic linspace) as the forward
hipError_t launch_resample_bwd(int64_t B, int N, const float* bins, const float* weights, const float* u_rand, float padding,
                               const float* d_t_new, float* d_weights, hipStream_t st);

// ---- kernels_wgrad.hip -----------------------------------------------------------------------
constexpr int kWgradJobFloats = 8 * 9 * 64 * 16;     // fp32 partials per (job, split): [wave][slot][lane][reg]
struct WgradJob {                                     // one row of mlp_train_plan.TrainPlan.job_table()
    int nA, nB, bias, b_src;                          // b_src 1: the B blocks are 32-feature column blocks of the row-major ENCODING (pre-GEMM plans)
    int a_blk[8];
    int b_blk[8];
};
struct WgradEnc {                                     // the encoding the b_src = 1 jobs contract against (null / 0 when the plan has none)
    const void* enc;                                  // bf16 [M, row_elems] row-major -- or (row_bytes == 0) the MFMA B-operand FRAGMENTS k_cast_ipe_360_tile
                                                      // writes for k_pre_gemm: [wave tile][frag_ksteps][64 lanes][8 bf16], whole 256-sample tiles
    int64_t M;                                        // rows (samples); wave tiles reach past it, rows are clamped
    int row_bytes;                                    // 2 * xyz_dim; 0 = fragment layout
    int frag_ksteps;                                  // fragment layout: k-steps per wave tile (xyz_dim / 16)
};
int mlp_wgrad_lds_bytes();
hipError_t launch_transpose_sq(int n, const float* in, float* out, hipStream_t st);
hipError_t launch_mlp_wgrad(const void* HT, const void* GT, const WgradJob* jobs, const void* wg_tab, int num_wgs,
                            int64_t n_wt, int NH, int NG, float* partials, hipStream_t st, const WgradEnc* enc_record = nullptr);
// writes the WgradEnc record behind an act buffer's T-blocks (device memory: the forward may be part of a captured graph)
hipError_t launch_wgrad_record_enc(void* record, const void* enc, int64_t M, int row_bytes, hipStream_t st, int frag_ksteps = 0);
struct WgradPost {                                    // chain-rule step that replaces the bottleneck T-blocks (W == 0: none)
    int W, Wc, ldv;                                   // net_width, net_width_condition, in_features of the view layer
    int off_extra_w, off_extra_b, off_view_w, off_view_b;   // flat gradient offsets
    const float* extra_wT;                            // W_extra^T, fp32 copy made by mipnerf_set_params
    const float* extra_w;                             // fp32 master parameters (device)
    const float* extra_b;
    const float* view_w;
};
hipError_t launch_wgrad_reduce(const float* partials, const int32_t* otab, const void* job_slots, int njobs,
                               float* grad_flat, float* scratch, int nparams, const WgradPost& post, bool accumulate,
                               hipStream_t st);

// ---- kernels_mlp_f32.hip ----------------------------------------------------------------------
constexpr int kF32MaxLayers = 16;
struct F32Layer {
    int x_in0, kb0;  // K segment 0: first LDS column, number of 16-wide k blocks (the activation buffer, or the encoding for layer 0)
    int x_in1, kb1;  // K segment 1 (encoding of the skip concat / view features), kb1 = 0 when absent
    int x_out;       // first LDS column of the output buffer
    int ntiles;      // out tiles (32 rows each)
    int first_tile;  // global tile index (bias table row)
    int relu;
    int kind;        // 0: write tiles to X[:, x_out + 32 t]; 1: head (last tile = density); 2: colour
    int chunk0;      // first 2-KiB chunk of this layer in the fp32 stream
    int stage_view;  // 1: load the view encoding into the encoding columns during this layer
    int pad;         // bit 0 / 1: K segment 0 / 1 = the sample encoding streamed from global memory (F32Net.pad = 1: wide encodings)
};
struct F32Net {
    int nlayers;
    int width;       // net_width
    int xyz_dim;
    int ldx;         // LDS row stride in floats: 2 * width + max(xyz_dim, 32) + 4 (2 * width + 64 + 4 when the encoding is streamed)
    int enc_col;     // 2 * width
    int dens_col;    // first spare column behind the encoding
    int num_rgb;
    int pad;         // 1: the sample encoding is NOT staged in LDS (it would not fit a 64-sample tile); layers flag the segments that stream it
    const float* dens_w;   // fp32 master parameters of the two thin heads (device pointers, set per launch):
    const float* dens_b;   //   density_layer.weight [1, W] / .bias, color_layer.weight [num_rgb, K] / .bias
    const float* col_w;
    const float* col_b;
    F32Layer layers[kF32MaxLayers];
};
int mlp_f32_tile_samples(int ldx);      // 64, 32, or 0 (the LDS row does not fit)
hipError_t launch_mlp_f32(const F32Net& net, const float* stream_w, const float* bias_tab, const float* enc,
                          const float* viewenc, float* rgb_sigma, float* raw_out, int64_t M, int num_samples,
                          float density_bias, float rgb_padding, float* save, unsigned long long* save_bits, const float* dnoise,
                          float dnoise_scale, hipStream_t st);
// words (uint64) of one slot of the ReLU sign-bit side table of `save` (k_mlp_f32 writes it, the dgrad epilogue reads it)
__host__ __device__ inline int64_t f32_bits_slot_words(int64_t M, int W) { return ((M * (W / 4) + 63) / 64) * 4; }

// ---- kernels_gemm_f32.hip (parity-mode backward) -------------------------------------------------
hipError_t launch_gemm_f32(bool trans_a, int M, int N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb,
                           int b_row_div, bool b_ones, float* C, int64_t ldc, bool accumulate, int splits, float* partial,
                           hipStream_t st);
bool gemm_f32_big_ok(int M, int N, int64_t K, const float* A, int64_t lda);
hipError_t launch_gemm_f32_big(bool trans_a, int M, int N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb,
                               float* C, int64_t ldc, bool accumulate, int splits, float* partial, const float* relu_x,
                               const float* r1_col, int64_t r1_ld, const float* r1_row, float* bias_out, hipStream_t st,
                               const unsigned long long* relu_bits = nullptr);
hipError_t launch_thin_wgrad(int64_t S, int C, int R, const float* Xc, int64_t ldxc, const float* Yr, int64_t ldyr, int rowdiv,
                             float* out, int64_t ldo_c, int64_t ldo_r, float* out_bias, bool accumulate, float* partial,
                             hipStream_t st);
hipError_t launch_relu_mask(int64_t n, const float* x, float* g, hipStream_t st);

// ---- kernels_pack.hip ---------------------------------------------------------------------------
constexpr int kMaxParamTensors = 32;
struct ParamPtrs {
    const float* p[kMaxParamTensors];
};
// table entry: -1 => 0, else (tensor << 20) | element offset
hipError_t launch_pack(const int32_t* table, int64_t n, const ParamPtrs& ptrs, void* out, bool bf16, hipStream_t st);
constexpr int kMaxPackSegments = 16;
struct PackSegments {              // several (table -> stream) packs as one launch
    int n;
    int64_t start[kMaxPackSegments + 1];
    const int32_t* table[kMaxPackSegments];
    void* out[kMaxPackSegments];
    int bf16[kMaxPackSegments];
};
hipError_t launch_pack_multi(const PackSegments& sg, const ParamPtrs& ptrs, hipStream_t st);

// ---- kernels_eval.hip ---------------------------------------------------------------------------
int64_t eval_errors_partial_floats(int H, int W);
hipError_t launch_eval_errors(int H, int W, const float* pred, const float* gt, float* partial, float* out, hipStream_t st);

// ---- selftest.hip -------------------------------------------------------------------------------
// returns 0 if the MFMA fragment layouts and the LDS-DMA path behave as the kernels assume;
// otherwise a bit mask (1: bf16 32x32x16 layout, 2: f32 32x32x2 layout, 4: global_load_lds)
int run_selftest(hipStream_t st, char* msg, int msg_cap);

// ---- kernels_diag.hip (diagnostics bench.py reports next to the headline; they allocate and synchronise) ---------------
int run_mfma_ceiling(int lds_reads_per_mfma, int waves_per_simd, int random_operands, double seconds, double* tflops,
                     double* ms_per_launch, double* clock_ghz, hipStream_t st, char* msg, int msg_cap);
int run_handoff_probe(int same_xcd, int flavour, int tiles, int ring, int tile_bytes, int mfma_per_wave, int reps, double* out,
                      hipStream_t st, char* msg, int msg_cap);

}  // namespace mip

// Internal launcher declarations shared by the .hip translation units and capi.hip.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mip {

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
hipError_t launch_volumetric_rendering(int64_t B, int N, const float* rgb_sigma, const float* t, const float* dirs,
                                       int white_bkgd, float* comp_rgb, float* distance, float* acc,
                                       float* weights, hipStream_t st);
hipError_t launch_piecewise_constant_pdf(int64_t B, int N, const float* bins, const float* weights, int n_draws,
                                         const float* u_rand, bool blur, float padding, float* out,
                                         hipStream_t st);

// ---- kernels_train.hip ------------------------------------------------------------------------
hipError_t launch_activate(int64_t M, const float* raw, float rgb_padding, float density_bias, float* out, hipStream_t st);
hipError_t launch_volumetric_rendering_bwd(int64_t B, int N, const float* rgb_sigma, const float* t, const float* dirs,
                                           int white_bkgd, const float* g_rgb, const float* g_dist, const float* g_acc,
                                           const float* g_w, float rgb_padding, float* d_raw, hipStream_t st);
hipError_t launch_distloss(int64_t B, int N, const float* weights, const float* t, float* ray_loss, const float* g_ray,
                           float* d_w, hipStream_t st);

// ---- mlp_bf16_gen.hip (generated) -------------------------------------------------------------
int mlp_bf16_lds_bytes();
hipError_t launch_mlp_bf16(const void* stream_w, const float* bias_tab, const void* enc, const void* viewenc,
                           float* rgb_sigma, float* raw_out, int64_t M, int num_samples, float density_bias,
                           float rgb_padding, int grid_limit, bool dma, hipStream_t st);

// ---- kernels_mlp_f32.hip ----------------------------------------------------------------------
constexpr int kF32MaxLayers = 16;
struct F32Layer {
    int x_in;        // first LDS column of the layer input
    int kb;          // number of 16-wide k blocks
    int ntiles;      // out tiles (32 rows each)
    int first_tile;  // global tile index (bias table row)
    int relu;
    int kind;        // 0: write tiles to X[:, 0:32*ntiles]; 1: head (last tile = density); 2: colour
    int chunk0;      // first 2-KiB chunk of this layer in the fp32 stream
    int pad;
};
struct F32Net {
    int nlayers;
    int width;       // net_width (column where the encoding / view features live)
    int xyz_dim;
    int ldx;         // LDS row stride in floats
    F32Layer layers[kF32MaxLayers];
};
hipError_t launch_mlp_f32(const F32Net& net, const float* stream_w, const float* bias_tab, const float* enc,
                          const float* viewenc, float* rgb_sigma, float* raw_out, int64_t M, int num_samples,
                          float density_bias, float rgb_padding, hipStream_t st);

// ---- kernels_pack.hip ---------------------------------------------------------------------------
constexpr int kMaxParamTensors = 32;
struct ParamPtrs {
    const float* p[kMaxParamTensors];
};
// table entry: -1 => 0, else (tensor << 20) | element offset
hipError_t launch_pack(const int32_t* table, int64_t n, const ParamPtrs& ptrs, void* out, bool bf16, hipStream_t st);

// ---- selftest.hip -------------------------------------------------------------------------------
// returns 0 if the MFMA fragment layouts and the LDS-DMA path behave as the kernels assume;
// otherwise a bit mask (1: bf16 32x32x16 layout, 2: f32 32x32x2 layout, 4: global_load_lds)
int run_selftest(hipStream_t st, char* msg, int msg_cap);

}  // namespace mip

// Ray-side kernels of the Mip-NeRF hot path for gfx950 (wave64):
//   sample_along_rays (t part), cast_rays, cast_rays+integrated_pos_enc, pos_enc,
//   volumetric_rendering (one wavefront per ray, wave scan), resample_along_rays /
//   sorted_piecewise_constant_pdf (one wavefront per ray, CDF in LDS).
// All of them are HBM-/VALU-bound elementwise or per-ray-scan work: coalesced SoA reads,
// 16-byte stores, scans with DPP/shuffle wave primitives.  Compiled with -ffp-contract=off
// (see raymath.hpp).
#include <hip/hip_runtime.h>

#include "kernels.hpp"
#include "raymath.hpp"
#include "raywave.hpp"

namespace mip {

// ------------------------------------------------------------------------------------------
// sample_along_rays, t part (models/mip.py:143-163)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float level0_t(float nearv, float farv, int n_samples, int i, bool disparity) {
    const float lin = torch_linspace_at(0.0f, 1.0f, n_samples + 1, i);
    if (disparity) return 1.0f / (1.0f / nearv * (1.0f - lin) + 1.0f / farv * lin);
    return nearv + (farv - nearv) * lin;
}

__global__ void __launch_bounds__(256)
k_sample_along_rays(int64_t B, int N, const float* __restrict__ nearp, const float* __restrict__ farp,
                    const float* __restrict__ t_rand, int disparity, float* __restrict__ t_out) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = B * (int64_t)(N + 1);
    if (idx >= total) return;
    const int64_t b = idx / (N + 1);
    const int i = (int)(idx - b * (N + 1));
    const float nv = nearp[b], fv = farp[b];
    float t = level0_t(nv, fv, N, i, disparity);
    if (t_rand != nullptr) {
        // mids = .5*(t[1:]+t[:-1]); upper=[mids, t[-1]]; lower=[t[0], mids]  (mip.py:156-160)
        const float tm = (i > 0) ? level0_t(nv, fv, N, i - 1, disparity) : t;
        const float tp = (i < N) ? level0_t(nv, fv, N, i + 1, disparity) : t;
        const float lower = (i > 0) ? 0.5f * (t + tm) : t;
        const float upper = (i < N) ? 0.5f * (tp + t) : t;
        t = lower + (upper - lower) * t_rand[idx];
    }
    t_out[idx] = t;
}

// ------------------------------------------------------------------------------------------
// cast_rays -> means / covs (models/mip.py:81-103), one thread per sample
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_cast_rays(int64_t B, int N, const float* __restrict__ t, const float* __restrict__ origins,
            const float* __restrict__ dirs, const float* __restrict__ radii,
            float* __restrict__ means, float* __restrict__ covs) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= B * (int64_t)N) return;
    const int64_t b = s / N;
    const int i = (int)(s - b * N);
    const float d[3] = {dirs[b * 3], dirs[b * 3 + 1], dirs[b * 3 + 2]};
    const float o[3] = {origins[b * 3], origins[b * 3 + 1], origins[b * 3 + 2]};
    const float t0 = t[b * (N + 1) + i], t1 = t[b * (N + 1) + i + 1];
    const Gauss3 g = conical_frustum_to_gaussian(t0, t1, d, o, radii[b]);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (means) means[s * 3 + a] = g.mean[a];
        if (covs) covs[s * 3 + a] = g.cov[a];
    }
}

// ------------------------------------------------------------------------------------------
// cast_rays + integrated_pos_enc fused (models/mip.py:81-103, 322-350).
// Two threads per sample: thread q handles degrees [q*L/2, (q+1)*L/2) of BOTH halves
// (sin | "cos"), so each exp(-0.5*var) is computed once.  Output row = 6L features,
// written as 16-byte vectors.  L = max_deg - min_deg must be even (16 at the shipped config).
// ------------------------------------------------------------------------------------------
template <typename OutT> struct Pack;
template <> struct Pack<float> {
    static constexpr int kPer16 = 4;
    __device__ static void store(float* dst, const float* v, int n) {  // n multiple of 4
        for (int i = 0; i < n; i += 4)
            *reinterpret_cast<float4*>(dst + i) = make_float4(v[i], v[i + 1], v[i + 2], v[i + 3]);
    }
};
template <> struct Pack<__bf16> {
    static constexpr int kPer16 = 8;
    __device__ static void store(__bf16* dst, const float* v, int n) {  // n multiple of 8
        typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
        for (int i = 0; i < n; i += 8) {
            bf16x8 p;
#pragma unroll
            for (int j = 0; j < 8; ++j) p[j] = (__bf16)v[i + j];   // RNE
            *reinterpret_cast<bf16x8*>(dst + i) = p;
        }
    }
};

// Accuracy policy per output type: float rows feed the exact-fp32 MLP (parity mode) and use the accurate
// libm sin/exp (<= 1-2 ulp, like torch); bf16 rows are rounded to 8 bits anyway and use the fast pair.
template <typename OutT> struct IpeMath;
template <> struct IpeMath<float> {
    __device__ static float sin(float x) { return sin_accurate(x); }
    __device__ static float exp(float x) { return exp_accurate(x); }
};
template <> struct IpeMath<__bf16> {
    __device__ static float sin(float x) { return sin_fast(x); }
    __device__ static float exp(float x) { return exp_fast(x); }
};

// Thread q in {0,1} of a sample writes degrees [q*L/2, (q+1)*L/2) of both halves (sin | "cos").
template <typename OutT, int L>
__device__ __forceinline__ void ipe_write(const Gauss3& g, int q, int min_deg, OutT* row) {
    constexpr int H = 3 * L / 2;   // features per thread per half
    float fs[H], fc[H];
#pragma unroll
    for (int ll = 0; ll < L / 2; ++ll) {
        const int l = q * (L / 2) + ll;
        const float scale = (float)(1u << (l + min_deg));
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float y = g.mean[a] * scale;
            const float yv = g.cov[a] * (scale * scale);
            const float damp = IpeMath<OutT>::exp(-0.5f * yv);
            fs[ll * 3 + a] = damp * IpeMath<OutT>::sin(y);
            fc[ll * 3 + a] = damp * IpeMath<OutT>::sin(y + kHalfPiF);
        }
    }
    Pack<OutT>::store(row + q * H, fs, H);
    Pack<OutT>::store(row + 3 * L + q * H, fc, H);
}

template <typename OutT, int L>
__global__ void __launch_bounds__(256)
k_cast_ipe(int64_t B, int N, int min_deg, int disable_integration, const float* __restrict__ t,
           const float* __restrict__ origins, const float* __restrict__ dirs,
           const float* __restrict__ radii, OutT* __restrict__ enc) {
    static_assert(L % 2 == 0 && (3 * L / 2) % 8 == 0, "vector stores need 3L/2 % 8 == 0");
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t s = gid >> 1;
    const int q = (int)(gid & 1);
    if (s >= B * (int64_t)N) return;
    const int64_t b = s / N;
    const int i = (int)(s - b * N);
    const float d[3] = {dirs[b * 3], dirs[b * 3 + 1], dirs[b * 3 + 2]};
    const float o[3] = {origins[b * 3], origins[b * 3 + 1], origins[b * 3 + 2]};
    const float t0 = t[b * (N + 1) + i], t1 = t[b * (N + 1) + i + 1];
    Gauss3 g = conical_frustum_to_gaussian(t0, t1, d, o, radii[b]);
    if (disable_integration) g.cov[0] = g.cov[1] = g.cov[2] = 0.0f;   // mip_nerf.py:210-211
    ipe_write<OutT, L>(g, q, min_deg, enc + s * (int64_t)(6 * L));
}

// integrated_pos_enc on given means / diagonal covariances (models/mip.py:322-350)
template <typename OutT, int L>
__global__ void __launch_bounds__(256)
k_integrated_pos_enc(int64_t M, int min_deg, const float* __restrict__ means, const float* __restrict__ covs,
                     OutT* __restrict__ enc) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t s = gid >> 1;
    const int q = (int)(gid & 1);
    if (s >= M) return;
    Gauss3 g;
#pragma unroll
    for (int a = 0; a < 3; ++a) { g.mean[a] = means[s * 3 + a]; g.cov[a] = covs[s * 3 + a]; }
    ipe_write<OutT, L>(g, q, min_deg, enc + s * (int64_t)(6 * L));
}

// ------------------------------------------------------------------------------------------
// pos_enc(viewdirs) (models/mip.py:353-363), row stride ld, pad columns zeroed
// ------------------------------------------------------------------------------------------
template <typename OutT>
__global__ void __launch_bounds__(256)
k_pos_enc(int64_t B, int deg, const float* __restrict__ viewdirs, OutT* __restrict__ out, int ld) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= B * (int64_t)ld) return;
    const int64_t b = gid / ld;
    const int c = (int)(gid - b * ld);
    const float v[3] = {viewdirs[b * 3], viewdirs[b * 3 + 1], viewdirs[b * 3 + 2]};
    const float f = (c < 3 + 6 * deg) ? view_feature(v, c, deg) : 0.0f;
    out[gid] = (OutT)f;
}

// ------------------------------------------------------------------------------------------
// volumetric_rendering (models/mip.py:366-401).  One wavefront per ray, 4 rays per block.
// Lane l owns the K = ceil(N/64) consecutive samples [l*K, l*K+K): a local running sum plus
// one wave-wide exclusive scan gives the exclusive cumsum of sigma*delta.
// ------------------------------------------------------------------------------------------
template <int K>
__global__ void __launch_bounds__(256)
k_volumetric_rendering(int64_t B, int N, const float4* __restrict__ rgb_sigma,
                       const float* __restrict__ t, const float* __restrict__ dirs, int white_bkgd,
                       float* __restrict__ comp_rgb, float* __restrict__ distance,
                       float* __restrict__ acc_out, float* __restrict__ weights) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= B) return;   // whole wave exits together (b is wave-uniform)
    const float dx = dirs[b * 3], dy = dirs[b * 3 + 1], dz = dirs[b * 3 + 2];
    const float dn = sqrtf(dx * dx + dy * dy + dz * dz);   // torch.linalg.norm
    float w[K];
    composite_ray<K>(true, lane, N, rgb_sigma + b * (int64_t)N, t + b * (int64_t)(N + 1), dn, white_bkgd, comp_rgb + b * 3, distance + b,
                     acc_out + b, weights + b * (int64_t)N, w);
}

// (the sampler's per-ray body pdf_ray and its constants live in raywave.hpp)
template <int K, bool BLUR>
__global__ void __launch_bounds__(64 * kRaysPerBlock)
k_piecewise_constant_pdf(int64_t B, int N, const float* __restrict__ bins, const float* __restrict__ weights,
                         int n_draws, const float* __restrict__ u_rand, float padding,
                         float u_step, float u_jitter, float* __restrict__ out) {
    __shared__ float s_w[kRaysPerBlock][PdfRow<K>::kBins + 2];
    __shared__ float s_cdf[kRaysPerBlock][PdfRow<K>::kBins + 2];
    __shared__ float s_bins[kRaysPerBlock][PdfRow<K>::kBins + 2];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int64_t b = (int64_t)blockIdx.x * kRaysPerBlock + wv;
    const bool active = b < B;
    const int64_t bb = active ? b : B - 1;    // inactive waves shadow the last ray (no stores)
    const float* wb = weights + bb * (int64_t)N;
    const float* binb = bins + bb * (int64_t)(N + 1);
    const int i0 = lane * K;
    // stage weights and bins
#pragma unroll
    for (int k = 0; k < K; ++k)
        if (i0 + k < N) s_w[wv][i0 + k] = wb[i0 + k];
    for (int j = lane; j <= N; j += 64) s_bins[wv][j] = binb[j];
    __syncthreads();
    pdf_ray<K, BLUR>(lane, N, s_w[wv], s_cdf[wv], s_bins[wv], n_draws, u_rand ? u_rand + bb * (int64_t)n_draws : nullptr, padding, u_step,
                     u_jitter, active ? out + b * (int64_t)n_draws : nullptr);
}

// volumetric_rendering of one level FOLLOWED BY resample_along_rays' t part for the next level in ONE launch (the coarse level of
// every forward): the ray's weights go from the compositing registers straight into the sampler's LDS row -- one launch, one
// kernel boundary and one [B, N] read less per forward than k_volumetric_rendering + k_piecewise_constant_pdf<., true>.
template <int K>
__global__ void __launch_bounds__(64 * kRaysPerBlock)
k_composite_resample(int64_t B, int N, const float4* __restrict__ rgb_sigma, const float* __restrict__ t,
                     const float* __restrict__ dirs, int white_bkgd, float* __restrict__ comp_rgb, float* __restrict__ distance,
                     float* __restrict__ acc_out, float* __restrict__ weights, const float* __restrict__ bins,
                     const float* __restrict__ u_rand, float padding, float u_step, float u_jitter, float* __restrict__ t_new) {
    __shared__ float s_w[kRaysPerBlock][PdfRow<K>::kBins + 2];
    __shared__ float s_cdf[kRaysPerBlock][PdfRow<K>::kBins + 2];
    __shared__ float s_bins[kRaysPerBlock][PdfRow<K>::kBins + 2];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int64_t b = (int64_t)blockIdx.x * kRaysPerBlock + wv;
    const bool active = b < B;
    const int64_t bb = active ? b : B - 1;
    const float dx = dirs[bb * 3], dy = dirs[bb * 3 + 1], dz = dirs[bb * 3 + 2];
    const float dn = sqrtf(dx * dx + dy * dy + dz * dz);
    // the sampler's bins: this level's fence posts, or (unbounded scenes) their inverse depths -- staged FIRST, so that their load is in
    // flight beside the compositing's loads instead of starting a second memory round trip behind its scans
    const float* binb = bins + bb * (int64_t)(N + 1);
    for (int j = lane; j <= N; j += 64) s_bins[wv][j] = binb[j];
    float w[K];
    composite_ray<K>(active, lane, N, rgb_sigma + bb * (int64_t)N, t + bb * (int64_t)(N + 1), dn, white_bkgd, comp_rgb + bb * 3,
                     distance + bb, acc_out + bb, weights + bb * (int64_t)N, w);
    const int i0 = lane * K;
#pragma unroll
    for (int k = 0; k < K; ++k)
        if (i0 + k < N) s_w[wv][i0 + k] = w[k];
    __syncthreads();
    const int n_draws = N + 1;
    pdf_ray<K, true>(lane, N, s_w[wv], s_cdf[wv], s_bins[wv], n_draws, u_rand ? u_rand + bb * (int64_t)n_draws : nullptr, padding, u_step,
                     u_jitter, active ? t_new + b * (int64_t)n_draws : nullptr);
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
static inline unsigned grid_for(int64_t n, int block) { return (unsigned)((n + block - 1) / block); }

hipError_t launch_sample_along_rays(int64_t B, int N, const float* nearp, const float* farp,
                                    const float* t_rand, int disparity, float* t_out, hipStream_t st) {
    const int64_t total = B * (int64_t)(N + 1);
    hipLaunchKernelGGL(k_sample_along_rays, dim3(grid_for(total, 256)), dim3(256), 0, st, B, N, nearp, farp,
                       t_rand, disparity, t_out);
    return hipGetLastError();
}

hipError_t launch_cast_rays(int64_t B, int N, const float* t, const float* origins, const float* dirs,
                            const float* radii, float* means, float* covs, hipStream_t st) {
    hipLaunchKernelGGL(k_cast_rays, dim3(grid_for(B * (int64_t)N, 256)), dim3(256), 0, st, B, N, t, origins, dirs,
                       radii, means, covs);
    return hipGetLastError();
}

hipError_t launch_cast_ipe(int64_t B, int N, int min_deg, int max_deg, int disable_integration,
                           const float* t, const float* origins, const float* dirs, const float* radii,
                           void* enc, bool bf16, hipStream_t st) {
    if (max_deg - min_deg != 16) return hipErrorInvalidValue;   // generated for L = 16
    const int64_t threads = 2 * B * (int64_t)N;
    if (bf16)
        hipLaunchKernelGGL((k_cast_ipe<__bf16, 16>), dim3(grid_for(threads, 256)), dim3(256), 0, st, B, N, min_deg,
                           disable_integration, t, origins, dirs, radii, (__bf16*)enc);
    else
        hipLaunchKernelGGL((k_cast_ipe<float, 16>), dim3(grid_for(threads, 256)), dim3(256), 0, st, B, N, min_deg,
                           disable_integration, t, origins, dirs, radii, (float*)enc);
    return hipGetLastError();
}

hipError_t launch_integrated_pos_enc(int64_t M, int min_deg, int max_deg, const float* means, const float* covs,
                                     void* enc, bool bf16, hipStream_t st) {
    if (max_deg - min_deg != 16) return hipErrorInvalidValue;
    const int64_t threads = 2 * M;
    if (bf16)
        hipLaunchKernelGGL((k_integrated_pos_enc<__bf16, 16>), dim3(grid_for(threads, 256)), dim3(256), 0, st, M, min_deg,
                           means, covs, (__bf16*)enc);
    else
        hipLaunchKernelGGL((k_integrated_pos_enc<float, 16>), dim3(grid_for(threads, 256)), dim3(256), 0, st, M, min_deg,
                           means, covs, (float*)enc);
    return hipGetLastError();
}

hipError_t launch_pos_enc(int64_t B, int deg, const float* viewdirs, void* out, int ld, bool bf16, hipStream_t st) {
    const int64_t total = B * (int64_t)ld;
    if (bf16)
        hipLaunchKernelGGL((k_pos_enc<__bf16>), dim3(grid_for(total, 256)), dim3(256), 0, st, B, deg, viewdirs,
                           (__bf16*)out, ld);
    else
        hipLaunchKernelGGL((k_pos_enc<float>), dim3(grid_for(total, 256)), dim3(256), 0, st, B, deg, viewdirs,
                           (float*)out, ld);
    return hipGetLastError();
}

hipError_t launch_volumetric_rendering(int64_t B, int N, const float* rgb_sigma, const float* t, const float* dirs,
                                       int white_bkgd, float* comp_rgb, float* distance, float* acc,
                                       float* weights, hipStream_t st) {
    const dim3 grid(grid_for(B, 4)), block(256);
    const float4* c = reinterpret_cast<const float4*>(rgb_sigma);
    const int K = (N + 63) / 64;
#define MIP_VR(KK)                                                                                        \
    hipLaunchKernelGGL((k_volumetric_rendering<KK>), grid, block, 0, st, B, N, c, t, dirs, white_bkgd,  \
                       comp_rgb, distance, acc, weights)
    switch (K) {
        // the K buckets EVERY per-ray kernel uses (1, 2, 4, 8, 16 samples per lane): a ray's sums associate by K, so one set of buckets lets
        // the fused launches (k_composite_resample, k_composite_train) reproduce the per-stage kernels bit for bit at every N <= 1024
        case 1: MIP_VR(1); break;
        case 2: MIP_VR(2); break;
        case 3: case 4: MIP_VR(4); break;
        case 5: case 6: case 7: case 8: MIP_VR(8); break;
        case 9: case 10: case 11: case 12: case 13: case 14: case 15: case 16: MIP_VR(16); break;
        default: return hipErrorInvalidValue;
    }
#undef MIP_VR
    return hipGetLastError();
}

hipError_t launch_piecewise_constant_pdf(int64_t B, int N, const float* bins, const float* weights, int n_draws,
                                         const float* u_rand, bool blur, float padding, float* out,
                                         hipStream_t st) {
    if (N > kPdfMaxBins || N < 1) return hipErrorInvalidValue;
    const dim3 grid(grid_for(B, kRaysPerBlock)), block(64 * kRaysPerBlock);
    // s = 1/num_samples ; jitter range (s - eps32) evaluated in double like the Python reference
    const double s = 1.0 / (double)n_draws;
    const float u_step = (float)s;
    const float u_jitter = (float)(s - (double)1.1920928955078125e-07f);
    const int K = (N + 63) / 64;
#define MIP_PDF(KK)                                                                                          \
    do {                                                                                                     \
        if (blur)                                                                                            \
            hipLaunchKernelGGL((k_piecewise_constant_pdf<KK, true>), grid, block, 0, st, B, N, bins, weights, \
                               n_draws, u_rand, padding, u_step, u_jitter, out);                            \
        else                                                                                                 \
            hipLaunchKernelGGL((k_piecewise_constant_pdf<KK, false>), grid, block, 0, st, B, N, bins, weights, \
                               n_draws, u_rand, padding, u_step, u_jitter, out);                            \
    } while (0)
    switch (K) {
        case 1: MIP_PDF(1); break;
        case 2: MIP_PDF(2); break;
        case 3: case 4: MIP_PDF(4); break;
        case 5: case 6: case 7: case 8: MIP_PDF(8); break;
        case 9: case 10: case 11: case 12: case 13: case 14: case 15: case 16: MIP_PDF(16); break;
        default: return hipErrorInvalidValue;
    }
#undef MIP_PDF
    return hipGetLastError();
}

hipError_t launch_composite_resample(int64_t B, int N, const float* rgb_sigma, const float* t, const float* dirs, int white_bkgd,
                                     float* comp_rgb, float* distance, float* acc, float* weights, const float* bins,
                                     const float* u_rand, float padding, float* t_new, hipStream_t st) {
    if (N > kPdfMaxBins || N < 1) return hipErrorInvalidValue;
    const dim3 grid(grid_for(B, kRaysPerBlock)), block(64 * kRaysPerBlock);
    const float4* c = reinterpret_cast<const float4*>(rgb_sigma);
    const int n_draws = N + 1;
    const double s = 1.0 / (double)n_draws;           // as launch_piecewise_constant_pdf
    const float u_step = (float)s;
    const float u_jitter = (float)(s - (double)1.1920928955078125e-07f);
    const int K = (N + 63) / 64;
#define MIP_CR(KK)                                                                                                       \
    hipLaunchKernelGGL((k_composite_resample<KK>), grid, block, 0, st, B, N, c, t, dirs, white_bkgd, comp_rgb, distance, acc, \
                       weights, bins, u_rand, padding, u_step, u_jitter, t_new)
    switch (K) {      // the same K buckets as the stand-alone kernels use, so both routes give the same bits
        case 1: MIP_CR(1); break;
        case 2: MIP_CR(2); break;
        case 3: case 4: MIP_CR(4); break;
        case 5: case 6: case 7: case 8: MIP_CR(8); break;
        case 9: case 10: case 11: case 12: case 13: case 14: case 15: case 16: MIP_CR(16); break;
        default: return hipErrorInvalidValue;
    }
#undef MIP_CR
    return hipGetLastError();
}

// pos_enc(viewdirs) and the coarse level's fence posts (sample_along_rays, t part) in ONE launch: both only read per-ray inputs
__global__ void __launch_bounds__(256)
k_ray_prologue(int64_t B, int deg, const float* __restrict__ viewdirs, void* __restrict__ venc, int ld, int venc_bf16, int N,
               const float* __restrict__ nearp, const float* __restrict__ farp, const float* __restrict__ t_rand, int disparity,
               float* __restrict__ t_out) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid < B * (int64_t)ld) {                   // k_pos_enc
        const int64_t b = gid / ld;
        const int c = (int)(gid - b * ld);
        const float v[3] = {viewdirs[b * 3], viewdirs[b * 3 + 1], viewdirs[b * 3 + 2]};
        const float f = (c < 3 + 6 * deg) ? view_feature(v, c, deg) : 0.0f;
        if (venc_bf16) reinterpret_cast<__bf16*>(venc)[gid] = (__bf16)f;
        else reinterpret_cast<float*>(venc)[gid] = f;
    }
    const int64_t total = B * (int64_t)(N + 1);
    if (gid < total) {                             // k_sample_along_rays
        const int64_t b = gid / (N + 1);
        const int i = (int)(gid - b * (N + 1));
        const float nv = nearp[b], fv = farp[b];
        float t = level0_t(nv, fv, N, i, disparity);
        if (t_rand != nullptr) {
            const float tm = (i > 0) ? level0_t(nv, fv, N, i - 1, disparity) : t;
            const float tp = (i < N) ? level0_t(nv, fv, N, i + 1, disparity) : t;
            const float lower = (i > 0) ? 0.5f * (t + tm) : t;
            const float upper = (i < N) ? 0.5f * (tp + t) : t;
            t = lower + (upper - lower) * t_rand[gid];
        }
        t_out[gid] = t;
    }
}

hipError_t launch_ray_prologue(int64_t B, int deg, const float* viewdirs, void* venc, int ld, bool venc_bf16, int N, const float* nearp,
                               const float* farp, const float* t_rand, int disparity, float* t_out, hipStream_t st) {
    const int64_t n = B * (int64_t)(ld > N + 1 ? ld : N + 1);
    hipLaunchKernelGGL(k_ray_prologue, dim3(grid_for(n, 256)), dim3(256), 0, st, B, deg, viewdirs, venc, ld, venc_bf16 ? 1 : 0, N, nearp,
                       farp, t_rand, disparity, t_out);
    return hipGetLastError();
}

}  // namespace mip

// ---- device-side ray generation ----------------------------------------------------------------
// datasets/datasets.py:214-263 (Blender._generate_rays) and :116-168 (Multicam._generate_rays) computed per pixel
// on the device instead of materialising 52 B/ray for every pixel of every image on the host.
// camera record (32 floats): c2w[3][4] | pix2cam[3][3] | W, H, near, far, lossmult, mode, focal | pad
//   mode 0 (Blender): camera_dir = ((x - W/2 + .5)/focal, -(y - H/2 + .5)/focal, -1)        datasets.py:226-228
//   mode 1 (Multicam): camera_dir = pix2cam @ (x + .5, y + .5, 1)                             datasets.py:125-131
// direction = c2w[:3,:3] @ camera_dir; origin = c2w[:3,3]; viewdir = direction / |direction|;
// radius = |direction(y, x) - direction(y+1, x)| * 2/sqrt(12), last row repeats the previous one (datasets.py:246-253:
// the neighbour is taken along axis 0 = image rows).
namespace mip {
namespace {
// T = float: the arithmetic of Blender / Multicam._generate_rays (float32 numpy); T = double: RenderGen (render_video.py:29-112 runs on the
// float64 poses of create_spheric_poses, so its directions and above all its radii -- the norm of a DIFFERENCE of neighbouring directions,
// 1e-3 of their size -- carry float64 accuracy before the .float() of render_video.py:131)
template <typename T>
__device__ __forceinline__ void pixel_direction(const T* __restrict__ cam, T x, T y, T d[3]) {
    T c0, c1, c2;
    if (cam[26] == T(0)) {
        const T W = cam[21], H = cam[22], f = cam[27];
        c0 = (x - W * T(0.5) + T(0.5)) / f;
        c1 = -(y - H * T(0.5) + T(0.5)) / f;
        c2 = T(-1);
    } else {
        const T px = x + T(0.5), py = y + T(0.5);
        c0 = cam[12] * px + cam[13] * py + cam[14];
        c1 = cam[15] * px + cam[16] * py + cam[17];
        c2 = cam[18] * px + cam[19] * py + cam[20];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) d[i] = cam[4 * i] * c0 + cam[4 * i + 1] * c1 + cam[4 * i + 2] * c2;
}
}  // namespace

template <typename T>
__global__ void __launch_bounds__(256)
k_generate_rays(int64_t n, const T* __restrict__ cams, const int32_t* __restrict__ cam_idx,
                const int32_t* __restrict__ pix_idx, float* __restrict__ origins, float* __restrict__ directions,
                float* __restrict__ viewdirs, float* __restrict__ radii, float* __restrict__ lossmult,
                float* __restrict__ nearp, float* __restrict__ farp) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const T* cam = cams + (size_t)(cam_idx ? cam_idx[i] : 0) * 32;
    const int W = (int)cam[21], H = (int)cam[22];
    const int p = pix_idx ? pix_idx[i] : (int)i;
    const int yy = p / W, xx = p - yy * W;
    T d[3], dn[3];
    pixel_direction<T>(cam, (T)xx, (T)yy, d);
    const int yn = yy + 1 < H ? yy + 1 : yy - 1;
    pixel_direction<T>(cam, (T)xx, (T)yn, dn);
    const T e0 = d[0] - dn[0], e1 = d[1] - dn[1], e2 = d[2] - dn[2];
    const T dx = sqrt(e0 * e0 + e1 * e1 + e2 * e2);
    const T nrm = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        origins[i * 3 + k] = (float)cam[4 * k + 3];
        directions[i * 3 + k] = (float)d[k];
        viewdirs[i * 3 + k] = (float)(d[k] / nrm);
    }
    radii[i] = (float)(dx * T(2) / (T)(sizeof(T) == 4 ? 3.4641016151377544f : 3.4641016151377544));      // np.sqrt(12) in the table's precision
    lossmult[i] = (float)cam[25];
    nearp[i] = (float)cam[23];
    farp[i] = (float)cam[24];
}

hipError_t launch_generate_rays(int64_t n, const float* cams, const int32_t* cam_idx, const int32_t* pix_idx,
                                float* origins, float* directions, float* viewdirs, float* radii, float* lossmult,
                                float* nearp, float* farp, hipStream_t st) {
    hipLaunchKernelGGL(k_generate_rays<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, cams, cam_idx, pix_idx,
                       origins, directions, viewdirs, radii, lossmult, nearp, farp);
    return hipGetLastError();
}

hipError_t launch_generate_rays_f64(int64_t n, const double* cams, const int32_t* cam_idx, const int32_t* pix_idx,
                                    float* origins, float* directions, float* viewdirs, float* radii, float* lossmult,
                                    float* nearp, float* farp, hipStream_t st) {
    hipLaunchKernelGGL(k_generate_rays<double>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, cams, cam_idx, pix_idx,
                       origins, directions, viewdirs, radii, lossmult, nearp, farp);
    return hipGetLastError();
}
}  // namespace mip

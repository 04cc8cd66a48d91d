// Training-side ray kernels for gfx950: activations of the raw MLP outputs, backward of
// volumetric_rendering (models/mip.py:366-401) fused with the activation derivatives
// (models/mip_nerf.py:236-238), and the distortion loss (models/mip.py:8-20) forward/backward in O(N)
// per ray (the reference builds two [B,N,N] tensors).  One wavefront per ray, scans in fp64.
#include <hip/hip_runtime.h>
#include <math.h>

#include "kernels.hpp"
#include "raymath.hpp"
#include "raywave.hpp"

namespace mip {

__device__ __forceinline__ double wsum64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// exclusive prefix (forward) / exclusive suffix (reverse) sums over the 64 lanes
__device__ __forceinline__ double wexcl_prefix64(double v, int lane) {
    double inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double n = __shfl_up(inc, o, 64);
        if (lane >= o) inc += n;
    }
    const double ex = __shfl_up(inc, 1, 64);
    return lane == 0 ? 0.0 : ex;
}
__device__ __forceinline__ double wexcl_suffix64(double v, int lane) {
    double inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double n = __shfl_down(inc, o, 64);
        if (lane + o < 64) inc += n;
    }
    const double ex = __shfl_down(inc, 1, 64);
    return lane == 63 ? 0.0 : ex;
}

// raw [M,4] = (raw_rgb, raw_density) -> (sigmoid*(1+2p)-p, softplus(raw+bias))  (mip_nerf.py:236-238)
__global__ void __launch_bounds__(256)
k_activate(int64_t M, const float4* __restrict__ raw, float rgb_padding, float density_bias, const float* __restrict__ dnoise,
           float dnoise_scale, float4* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const float4 r = raw[i];
    const float nd = dnoise ? r.w + dnoise_scale * dnoise[i] : r.w;      // mip_nerf.py:232-233
    out[i] = make_float4(rgb_activation(r.x, rgb_padding), rgb_activation(r.y, rgb_padding),
                         rgb_activation(r.z, rgb_padding), density_activation(nd, density_bias));
}

// Backward of volumetric_rendering + activations.
//   w_i = (1 - e^{-x_i}) T_i,  x_i = sigma_i * delta_i,  T_i = exp(-sum_{j<i} x_j)
//   G_i  = dL/dw_i = sum_c g_rgb_c (c_ic - [white]) + g_acc + g_dist * tmid_i + g_w_i
//   dL/dx_i = G_i T_{i+1} - sum_{k>i} G_k w_k ,  dL/dsigma_i = delta_i dL/dx_i ,  dL/dc_i = w_i g_rgb
//   d raw_rgb = dL/dc * (1+2p) s (1-s),  s = (c + p)/(1+2p) ;  d raw_density = dL/dsigma * (1 - e^{-sigma})
template <int K>
__global__ void __launch_bounds__(256)
k_volumetric_rendering_bwd(int64_t B, int N, const float4* __restrict__ rgb_sigma, const float* __restrict__ t,
                           const float* __restrict__ dirs, int white_bkgd, const float* __restrict__ g_rgb,
                           const float* __restrict__ g_dist, const float* __restrict__ g_acc,
                           const float* __restrict__ g_w, float rgb_padding, float4* __restrict__ d_raw,
                           float* __restrict__ d_t) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= B) return;
    const float dx = dirs[b * 3], dy = dirs[b * 3 + 1], dz = dirs[b * 3 + 2];
    const float dn = sqrtf(dx * dx + dy * dy + dz * dz);
    const float* tb = t + b * (int64_t)(N + 1);
    const float4* cb = rgb_sigma + b * (int64_t)N;
    const int i0 = lane * K;
    const float gr = g_rgb ? g_rgb[b * 3] : 0.f, gg = g_rgb ? g_rgb[b * 3 + 1] : 0.f, gb = g_rgb ? g_rgb[b * 3 + 2] : 0.f;
    const float ga = g_acc ? g_acc[b] : 0.f;
    const float bg = white_bkgd ? (gr + gg + gb) : 0.f;

    float tv[K + 1];
#pragma unroll
    for (int k = 0; k <= K; ++k) tv[k] = (i0 + k <= N) ? tb[i0 + k] : 0.0f;
    float4 c[K];
    float xx[K], delta[K];
    double pre[K];
    double run = 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const bool ok = i0 + k < N;
        c[k] = ok ? cb[i0 + k] : make_float4(0.f, 0.f, 0.f, 0.f);
        delta[k] = (tv[k + 1] - tv[k]) * dn;
        xx[k] = ok ? c[k].w * delta[k] : 0.0f;
        pre[k] = run;
        run += (double)xx[k];
    }
    const double off = wexcl_prefix64(run, lane);
    // distance = clamp(nan_to_num(sum w tmid), t0, tN): gradient flows only strictly inside the clamp
    float gd = 0.f;
    float w[K], T1[K], G[K];
    double sd = 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const bool ok = i0 + k < N;
        const float ex = expf(-xx[k]);
        const float trans = expf(-(float)(off + pre[k]));
        w[k] = ok ? (1.0f - ex) * trans : 0.0f;
        T1[k] = trans * ex;                     // T_{i+1}
        sd += (double)(w[k] * (0.5f * (tv[k] + tv[k + 1])));
    }
    float gd_lo = 0.f, gd_hi = 0.f;     // clamp(distance, t_0, t_N): outside the bounds the gradient goes to the bound (d_t only)
    if (g_dist) {
        const float dsum = (float)wsum64(sd);
        const float tn = tb[0], tf = tb[N];
        gd = (dsum == dsum && dsum >= tn && dsum <= tf) ? g_dist[b] : 0.f;
        if (dsum == dsum && dsum < tn) gd_lo = g_dist[b];
        if (dsum == dsum && dsum > tf) gd_hi = g_dist[b];
    }
    double loc = 0.0;       // sum over this lane's samples of G_k w_k
    double suf[K];          // suffix within the lane (exclusive)
#pragma unroll
    for (int k = K - 1; k >= 0; --k) {
        const bool ok = i0 + k < N;
        const float gw = (g_w && ok) ? g_w[b * (int64_t)N + i0 + k] : 0.f;
        G[k] = gr * c[k].x + gg * c[k].y + gb * c[k].z - bg + ga + gd * (0.5f * (tv[k] + tv[k + 1])) + gw;
        suf[k] = loc;
        loc += (double)(G[k] * w[k]);
    }
    const double soff = wexcl_suffix64(loc, lane);
    const float p = rgb_padding, sc = 1.0f + 2.0f * rgb_padding;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (i0 + k < N) {
            const float dxk = G[k] * T1[k] - (float)(soff + suf[k]);
            const float dsig = delta[k] * dxk;
            const float s0 = (c[k].x + p) / sc, s1 = (c[k].y + p) / sc, s2 = (c[k].z + p) / sc;
            float4 o;
            o.x = w[k] * gr * sc * s0 * (1.0f - s0);
            o.y = w[k] * gg * sc * s1 * (1.0f - s1);
            o.z = w[k] * gb * sc * s2 * (1.0f - s2);
            o.w = dsig * (1.0f - expf(-c[k].w));          // softplus'(x) = sigmoid(x) = 1 - e^{-softplus(x)}
            d_raw[b * (int64_t)N + i0 + k] = o;
        }
    }
    if (d_t) {
        // t enters through delta_i = (t_{i+1} - t_i) |d| (x_i = sigma_i delta_i) and through tmid_i in `distance`:
        //   dL/ddelta_i = sigma_i dL/dx_i ;  d_t[i] = |d| (a_{i-1} - a_i) + (h_{i-1} + h_i),  a = dL/ddelta / |d| * |d|, h = g_dist w / 2
        float a[K], h[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const bool ok = i0 + k < N;
            const float dxk = ok ? G[k] * T1[k] - (float)(soff + suf[k]) : 0.f;
            a[k] = ok ? c[k].w * dxk * dn : 0.f;
            h[k] = ok ? 0.5f * gd * w[k] : 0.f;
        }
        float pa = __shfl_up(a[K - 1], 1, 64), ph = __shfl_up(h[K - 1], 1, 64);
        if (lane == 0) { pa = 0.f; ph = 0.f; }
#pragma unroll
        for (int k = 0; k <= K; ++k) {
            const int i = i0 + k;
            const float ak = k < K ? a[k < K ? k : 0] : 0.f, hk = k < K ? h[k < K ? k : 0] : 0.f;
            // index i0 + K belongs to the next lane unless it is the last fence post of the ray
            if (i <= N && (k < K || i == N)) {
                float v = (pa - ak) + (ph + hk);
                if (i == 0) v += gd_lo;
                if (i == N) v += gd_hi;
                d_t[b * (int64_t)(N + 1) + i] = v;
            }
            if (k < K) { pa = a[k]; ph = h[k]; }
        }
    }
}

// distloss (models/mip.py:8-20) per ray, O(N): t is sorted, so |m_i - m_j| = m_max - m_min and
//   sum_ij w_i w_j |m_i - m_j| = 2 sum_i w_i (m_i P_i - Q_i),  P_i = sum_{j<i} w_j,  Q_i = sum_{j<i} w_j m_j
// ray_loss[b] = (1/3) sum_i interval_i w_i^2 + that double sum   (the reference then takes the batch mean)
// If g_ray != nullptr also writes d_w[b,i] = g_ray[b] * ( (2/3) interval_i w_i + 2 sum_j w_j |m_i - m_j| ).
template <int K>
__global__ void __launch_bounds__(256)
k_distloss(int64_t B, int N, const float* __restrict__ weights, const float* __restrict__ t,
           float* __restrict__ ray_loss, const float* __restrict__ g_ray, float g_const, float* __restrict__ d_w,
           float* __restrict__ d_t) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= B) return;
    const float* wb = weights + b * (int64_t)N;
    const int i0 = lane * K;
    float w[K];
#pragma unroll
    for (int k = 0; k < K; ++k) w[k] = i0 + k < N ? wb[i0 + k] : 0.f;
    distloss_ray<K>(lane, N, w, t + b * (int64_t)(N + 1), ray_loss ? ray_loss + b : nullptr, g_ray ? g_ray[b] : g_const,
                    d_w ? d_w + b * (int64_t)N : nullptr, d_t ? d_t + b * (int64_t)(N + 1) : nullptr);
}

// The ray-side tail of one level of the TRAINING forward in one launch: volumetric_rendering (weights stay in registers), the
// distortion loss forward + backward on those weights (d loss / d ray_loss is a constant), and -- for every level but the last --
// the next level's fence posts from the blurred weights (RESAMPLE).  Same per-ray device functions as k_volumetric_rendering,
// k_distloss and k_piecewise_constant_pdf<K, true>: same bits.  mipnerf_train_step ran those as three launches per level.
template <int K, bool RESAMPLE>
__global__ void __launch_bounds__(64 * kRaysPerBlock)
k_composite_train(int64_t B, int N, const float4* __restrict__ rgb_sigma, const float* __restrict__ t, const float* __restrict__ dirs,
                  int white_bkgd, float* __restrict__ comp_rgb, float* __restrict__ distance, float* __restrict__ acc_out,
                  float* __restrict__ weights, float* __restrict__ ray_loss, float g_const, float* __restrict__ d_w,
                  const float* __restrict__ u_rand, float padding, float u_step, float u_jitter, float* __restrict__ t_new,
                  const float* __restrict__ bins) {
    __shared__ float s_w[RESAMPLE ? kRaysPerBlock : 1][RESAMPLE ? PdfRow<K>::kBins + 2 : 1];
    __shared__ float s_cdf[RESAMPLE ? kRaysPerBlock : 1][RESAMPLE ? PdfRow<K>::kBins + 2 : 1];
    __shared__ float s_bins[RESAMPLE ? kRaysPerBlock : 1][RESAMPLE ? PdfRow<K>::kBins + 2 : 1];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int64_t b = (int64_t)blockIdx.x * kRaysPerBlock + wv;
    const bool active = b < B;
    if (!RESAMPLE && !active) return;          // no block barriers on this path
    const int64_t bb = active ? b : B - 1;     // RESAMPLE: inactive waves shadow the last ray (no stores)
    const float dx = dirs[bb * 3], dy = dirs[bb * 3 + 1], dz = dirs[bb * 3 + 2];
    const float dn = sqrtf(dx * dx + dy * dy + dz * dz);
    const float* tb = t + bb * (int64_t)(N + 1);
    float w[K];
    composite_ray<K>(active, lane, N, rgb_sigma + bb * (int64_t)N, tb, dn, white_bkgd, comp_rgb + bb * 3, distance + bb, acc_out + bb,
                     weights + bb * (int64_t)N, w);
    distloss_ray<K>(lane, N, w, tb, active ? ray_loss + bb : nullptr, g_const, active ? d_w + bb * (int64_t)N : nullptr, nullptr);
    if (RESAMPLE) {
        const int i0 = lane * K;
#pragma unroll
        for (int k = 0; k < K; ++k)
            if (i0 + k < N) s_w[wv][i0 + k] = w[k];
        // the sampler's bins: this level's fence posts, or (unbounded scenes) their inverse depths
        const float* binb = bins ? bins + bb * (int64_t)(N + 1) : tb;
        for (int j = lane; j <= N; j += 64) s_bins[wv][j] = binb[j];
        __syncthreads();
        const int n_draws = N + 1;
        pdf_ray<K, true>(lane, N, s_w[wv], s_cdf[wv], s_bins[wv], n_draws, u_rand ? u_rand + bb * (int64_t)n_draws : nullptr, padding, u_step,
                         u_jitter, active ? t_new + b * (int64_t)n_draws : nullptr);
    }
}

static inline unsigned gridf(int64_t n, int block) { return (unsigned)((n + block - 1) / block); }

hipError_t launch_composite_train(int64_t B, int N, const float* rgb_sigma, const float* t, const float* dirs, int white_bkgd,
                                  float* comp_rgb, float* distance, float* acc, float* weights, float* ray_loss, float g_const,
                                  float* d_w, const float* u_rand, float padding, float* t_new, hipStream_t st, const float* bins) {
    if (N > kPdfMaxBins || N < 1) return hipErrorInvalidValue;
    const dim3 grid(gridf(B, kRaysPerBlock)), block(64 * kRaysPerBlock);
    const float4* c = reinterpret_cast<const float4*>(rgb_sigma);
    const int n_draws = N + 1;
    const double s = 1.0 / (double)n_draws;           // as launch_piecewise_constant_pdf
    const float u_step = (float)s;
    const float u_jitter = (float)(s - (double)1.1920928955078125e-07f);
    const int K = (N + 63) / 64;
#define MIP_CT(KK)                                                                                                                   \
    do {                                                                                                                             \
        if (t_new) hipLaunchKernelGGL((k_composite_train<KK, true>), grid, block, 0, st, B, N, c, t, dirs, white_bkgd, comp_rgb, distance, \
                                      acc, weights, ray_loss, g_const, d_w, u_rand, padding, u_step, u_jitter, t_new, bins);              \
        else hipLaunchKernelGGL((k_composite_train<KK, false>), grid, block, 0, st, B, N, c, t, dirs, white_bkgd, comp_rgb, distance, acc, \
                                weights, ray_loss, g_const, d_w, u_rand, padding, u_step, u_jitter, t_new, bins);                         \
    } while (0)
    switch (K) {      // the K buckets every stand-alone per-ray kernel uses (1, 2, 4, 8, 16), so the fused route gives the same bits at every N <= 1024
        case 1: MIP_CT(1); break;
        case 2: MIP_CT(2); break;
        case 3: case 4: MIP_CT(4); break;
        case 5: case 6: case 7: case 8: MIP_CT(8); break;
        case 9: case 10: case 11: case 12: case 13: case 14: case 15: case 16: MIP_CT(16); break;
        default: return hipErrorInvalidValue;
    }
#undef MIP_CT
    return hipGetLastError();
}

hipError_t launch_activate(int64_t M, const float* raw, float rgb_padding, float density_bias, const float* dnoise,
                           float dnoise_scale, float* out, hipStream_t st) {
    hipLaunchKernelGGL(k_activate, dim3(gridf(M, 256)), dim3(256), 0, st, M, (const float4*)raw, rgb_padding,
                       density_bias, dnoise, dnoise_scale, (float4*)out);
    return hipGetLastError();
}

hipError_t launch_volumetric_rendering_bwd(int64_t B, int N, const float* rgb_sigma, const float* t, const float* dirs,
                                           int white_bkgd, const float* g_rgb, const float* g_dist, const float* g_acc,
                                           const float* g_w, float rgb_padding, float* d_raw, hipStream_t st, float* d_t) {
    const dim3 grid(gridf(B, 4)), block(256);
    const int K = (N + 63) / 64;
#define MIP_VB(KK)                                                                                                  \
    hipLaunchKernelGGL((k_volumetric_rendering_bwd<KK>), grid, block, 0, st, B, N, (const float4*)rgb_sigma, t, dirs, \
                       white_bkgd, g_rgb, g_dist, g_acc, g_w, rgb_padding, (float4*)d_raw, d_t)
    switch (K) {
        case 1: MIP_VB(1); break;
        case 2: MIP_VB(2); break;
        case 3: case 4: MIP_VB(4); break;
        case 5: case 6: case 7: case 8: MIP_VB(8); break;
        case 9: case 10: case 11: case 12: case 13: case 14: case 15: case 16: MIP_VB(16); break;
        default: return hipErrorInvalidValue;
    }
#undef MIP_VB
    return hipGetLastError();
}

hipError_t launch_distloss(int64_t B, int N, const float* weights, const float* t, float* ray_loss, const float* g_ray,
                           float g_const, float* d_w, hipStream_t st, float* d_t) {
    const dim3 grid(gridf(B, 4)), block(256);
    const int K = (N + 63) / 64;
#define MIP_DL(KK) hipLaunchKernelGGL((k_distloss<KK>), grid, block, 0, st, B, N, weights, t, ray_loss, g_ray, g_const, d_w, d_t)
    switch (K) {
        case 1: MIP_DL(1); break;
        case 2: MIP_DL(2); break;
        case 3: case 4: MIP_DL(4); break;
        case 5: case 6: case 7: case 8: MIP_DL(8); break;
        case 9: case 10: case 11: case 12: case 13: case 14: case 15: case 16: MIP_DL(16); break;
        default: return hipErrorInvalidValue;
    }
#undef MIP_DL
    return hipGetLastError();
}


// ---- loss of nerf_system.py:99-111 and its gradient w.r.t. the rendered colours, one launch --------------------------
//   mse_l = sum_b mask_b sum_c (rgb_l[b,c] - gt[b,c])^2 / sum_b mask_b       (mask = lossmult, or ones)
//   loss  = cm (mse_c + dm dl_c) + mse_f + dm dl_f,   dl_l = mean_b ray_loss_l[b]
//   d loss / d rgb_l[b,c] = k_l 2 mask_b (rgb_l - gt) / sum mask,  k_c = cm, k_f = 1
// One workgroup (the inputs are B x 3 floats per level): fp64 tree reductions, deterministic.
// out[0..5] = loss, mse_c, mse_f, dl_c, dl_f, psnr_f (= -10 log10 mean((rgb_f - gt)^2), nerf_system.py:113).
__global__ void __launch_bounds__(1024)
k_loss_fused(int64_t B, int nlevels, const float* __restrict__ rgb0, const float* __restrict__ rgb1,
             const float* __restrict__ gt, const float* __restrict__ lossmult, const float* __restrict__ ray_loss0,
             const float* __restrict__ ray_loss1, float coarse_mult, float dist_mult, float* __restrict__ g_rgb0,
             float* __restrict__ g_rgb1, float* __restrict__ out) {
    __shared__ double red[6][16];
    __shared__ double tot[6];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    double acc[6] = {0, 0, 0, 0, 0, 0};     // sum mask, sum mask se0, sum mask se1, sum dl0, sum dl1, sum se_fine (unmasked)
    const float* rgbf = nlevels > 1 ? rgb1 : rgb0;
#pragma unroll 4
    for (int64_t b = tid; b < B; b += blockDim.x) {
        const float m = lossmult ? lossmult[b] : 1.0f;
        float se0 = 0.f, se1 = 0.f, sef = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float g = gt[b * 3 + c];
            const float d0 = rgb0[b * 3 + c] - g;
            se0 += d0 * d0;
            if (nlevels > 1) { const float d1 = rgb1[b * 3 + c] - g; se1 += d1 * d1; }
            const float df = rgbf[b * 3 + c] - g;
            sef += df * df;
        }
        acc[0] += m; acc[1] += (double)m * se0; acc[2] += (double)m * se1;
        acc[3] += ray_loss0 ? ray_loss0[b] : 0.f; acc[4] += (nlevels > 1 && ray_loss1) ? ray_loss1[b] : 0.f;
        acc[5] += sef;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        double v = acc[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) red[k][w] = v;
    }
    __syncthreads();
    if (tid < 6) {
        double v = 0;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) v += red[tid][i];
        tot[tid] = v;
    }
    __syncthreads();
    const double sm = tot[0];
    if (tid == 0) {
        const double mse0 = tot[1] / sm, mse1 = tot[2] / sm, dl0 = tot[3] / (double)B, dl1 = tot[4] / (double)B;
        double loss;
        if (nlevels > 1) loss = coarse_mult * (mse0 + dist_mult * dl0) + mse1 + dist_mult * dl1;
        else loss = mse0 + dist_mult * dl0;                     // single level: it is the fine one
        out[0] = (float)loss; out[1] = (float)mse0; out[2] = (float)(nlevels > 1 ? mse1 : mse0);
        out[3] = (float)dl0; out[4] = (float)dl1;
        out[5] = (float)(-10.0 * log10(tot[5] / (3.0 * (double)B)));
    }
    const float k0 = nlevels > 1 ? coarse_mult : 1.0f;
    const float inv = (float)(2.0 / sm);
    for (int64_t i = tid; i < B * 3; i += blockDim.x) {
        const int64_t b = i / 3;
        const float m = lossmult ? lossmult[b] : 1.0f;
        const float g = gt[i];
        g_rgb0[i] = k0 * inv * m * (rgb0[i] - g);
        if (nlevels > 1) g_rgb1[i] = inv * m * (rgb1[i] - g);
    }
}

hipError_t launch_loss_fused(int64_t B, int nlevels, const float* rgb0, const float* rgb1, const float* gt, const float* lossmult,
                             const float* ray_loss0, const float* ray_loss1, float coarse_mult, float dist_mult, float* g_rgb0,
                             float* g_rgb1, float* out, hipStream_t st) {
    hipLaunchKernelGGL(k_loss_fused, dim3(1), dim3(1024), 0, st, B, nlevels, rgb0, rgb1, gt, lossmult, ray_loss0, ray_loss1,
                       coarse_mult, dist_mult, g_rgb0, g_rgb1, out);
    return hipGetLastError();
}

// ---- fused Adam over one flat parameter buffer -----------------------------------------------------
// torch.optim.Adam(params, lr) as the reference configures it (nerf_system.py:71-72: betas (0.9, 0.999), eps 1e-8, no
// weight decay, no amsgrad), same operation order as torch's single-tensor implementation:
//   m = lerp(m, g, 1-b1); v = b2 v + (1-b2) g^2; p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
__global__ void __launch_bounds__(256)
k_adam_flat(int64_t n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
            float step_size, float w1, float beta2, float w2, float eps, float bc2_sqrt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    const float mi = m[i] + (gi - m[i]) * w1;
    const float vi = v[i] * beta2 + w2 * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mi / denom);
}

// ---- device-side MipLRDecay + Adam hyper-parameters (graph-capturable optimiser step) ---------------------------------------
// One thread: t = ++(*step_count) is the 1-based Adam step; the learning rate is the reference scheduler's value at
// last_epoch = t - 1 (utils/lr_schedule.py:51-59: the scheduler is stepped AFTER the optimiser, so step t runs with
// get_lr(t - 1)); fp64 like numpy.  hyper[0..3] = lr, 1 - beta1^t, sqrt(1 - beta2^t), grad_scale.
__global__ void k_lr_schedule(LrSchedule sc, int64_t* __restrict__ step_count, float* __restrict__ hyper) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int64_t t = *step_count + 1;
    *step_count = t;
    const double epoch = (double)(t - 1);
    double delay_rate = 1.0;
    if (sc.lr_delay_steps > 0) {
        double x = epoch / (double)sc.lr_delay_steps;
        x = x < 0.0 ? 0.0 : (x > 1.0 ? 1.0 : x);
        delay_rate = sc.lr_delay_mult + (1.0 - sc.lr_delay_mult) * sin(0.5 * 3.141592653589793 * x);
    }
    double tt = epoch / (double)sc.max_steps;
    tt = tt < 0.0 ? 0.0 : (tt > 1.0 ? 1.0 : tt);
    const double log_lerp = exp(log(sc.lr_init) * (1.0 - tt) + log(sc.lr_final) * tt);
    const double lr = sc.constant_lr > 0.0 ? sc.constant_lr : delay_rate * log_lerp;
    // like torch's Adam: step_size = lr / bias_correction1 and sqrt(bias_correction2) are formed in double, rounded once
    hyper[0] = (float)lr;
    hyper[1] = (float)(lr / (1.0 - pow(sc.beta1, (double)t)));
    hyper[2] = (float)sqrt(1.0 - pow(sc.beta2, (double)t));
    hyper[3] = sc.grad_scale;
}

// k_adam_flat with lr / bias corrections read from device memory and the gradient pre-scaled (1 / world_size of the
// data-parallel SUM all-reduce: the mean is taken here instead of in a separate div_ kernel)
__global__ void __launch_bounds__(256)
k_adam_flat_dev(int64_t n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                const float* __restrict__ hyper, float w1, float beta2, float w2, float eps) {
    // w1 = float(1 - beta1), w2 = float(1 - beta2), rounded from the double differences like torch's scalar arguments
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float step_size = hyper[1], bc2_sqrt = hyper[2], gs = hyper[3];
    const float gi = gs == 1.0f ? g[i] : g[i] * gs;
    const float mi = m[i] + (gi - m[i]) * w1;               // exp_avg.lerp_(grad, 1 - beta1)
    const float vi = v[i] * beta2 + w2 * gi * gi;           // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mi / denom);                 // param.addcdiv_(exp_avg, denom, value=-step_size)
}

hipError_t launch_adam_scheduled(int64_t n, float* p, const float* g, float* m, float* v, const LrSchedule& sc,
                                 int64_t* step_count, float* hyper, hipStream_t st) {
    hipLaunchKernelGGL(k_lr_schedule, dim3(1), dim3(64), 0, st, sc, step_count, hyper);
    hipLaunchKernelGGL(k_adam_flat_dev, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, p, g, m, v, hyper,
                       (float)(1.0 - sc.beta1), (float)sc.beta2, (float)(1.0 - sc.beta2), (float)sc.eps);
    return hipGetLastError();
}

hipError_t launch_adam_flat(int64_t n, float* p, const float* g, float* m, float* v, double lr, double beta1, double beta2,
                            double eps, int step, hipStream_t st) {
    // the hyper-parameters arrive as the python doubles torch.optim.Adam holds; every derived scalar is formed in double and
    // rounded once, like torch's scalar arguments (same as the scheduled path)
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    hipLaunchKernelGGL(k_adam_flat, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, p, g, m, v, (float)(lr / bc1),
                       (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (float)sqrt(bc2));
    return hipGetLastError();
}
}  // namespace mip

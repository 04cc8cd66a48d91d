// Gradient through the resampler: MipNerf(stop_resample_grad=False) (models/mip.py:265-279, mip_nerf.py:204-214).
// With the flag off, the fine level's fence posts t' = sorted_piecewise_constant_pdf(t, blurpool(w) + padding) stay in the
// autograd graph, so the fine level's loss reaches the COARSE level's weights through
//   loss -> (a) the fine compositing (delta_i = (t'_{i+1} - t'_i)|d|) and distloss(w', t')     kernels_train.hip (d_t outputs)
//        -> (b) the fine MLP's input encoding: d_enc = delta_0 W_0 + delta_skip W_skip[:, W:]      capi.hip (two dgrad GEMMs)
//             -> integrated_pos_enc -> lift_gaussian -> conical_frustum_to_gaussian -> t'           k_cast_ipe_bwd (here)
//        -> t' -> the CDF of the coarse weights -> blur pool -> w                                    k_resample_bwd (here)
// fp32 parity mode only (the bf16 dgrad kernel stops at layer 1 by design: SURVEY 8d counts no dgrad into the encoding).
#include <hip/hip_runtime.h>

#include "kernels.hpp"
#include "raymath.hpp"

namespace mip {
namespace {

__device__ __forceinline__ double wsum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wexcl_prefix_d(double v, int lane) {
    double inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double n = __shfl_up(inc, o, 64);
        if (lane >= o) inc += n;
    }
    return inc - v;
}

// One thread per sample.  Forward (raymath.hpp): mu, hw -> (t_mean, t_var, r_var) -> mean_a = o_a + d_a t_mean,
// cov_a = t_var d_a^2 + r_var (1 - d_a^2 / (|d|^2 + 1e-10)) -> feature(half, l, a) = exp(-cov_a 4^l / 2) sin(mean_a 2^l [+ pi/2]).
__global__ void __launch_bounds__(256)
k_cast_ipe_bwd(int64_t B, int N, int min_deg, int ndeg, int disable_integration, const float* __restrict__ t,
               const float* __restrict__ origins, const float* __restrict__ dirs, const float* __restrict__ radii,
               const float* __restrict__ d_enc, float* __restrict__ d_t) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= B * (int64_t)N) return;
    const int64_t b = s / N;
    const int i = (int)(s - b * N);
    const float t0 = t[b * (int64_t)(N + 1) + i], t1 = t[b * (int64_t)(N + 1) + i + 1];
    const float d[3] = {dirs[b * 3], dirs[b * 3 + 1], dirs[b * 3 + 2]};
    const float o[3] = {origins[b * 3], origins[b * 3 + 1], origins[b * 3 + 2]};
    const float radius = radii[b];
    const Gauss3 g = conical_frustum_to_gaussian(t0, t1, d, o, radius);      // the forward's own fp32 values
    const float* ge = d_enc + s * (int64_t)(6 * ndeg);
    double dmean[3] = {0, 0, 0}, dcov[3] = {0, 0, 0};
    for (int l = 0; l < ndeg; ++l) {
        const float scale = (float)(1u << (l + min_deg));
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float y = g.mean[a] * scale;
            const float yv = disable_integration ? 0.0f : g.cov[a] * (scale * scale);
            const double e = exp(-0.5 * (double)yv);
            const double gs = ge[l * 3 + a], gc = ge[3 * ndeg + l * 3 + a];        // sin half, "cos" half
            const double ys = y, yc = y + kHalfPiF;
            const double ss = sin(ys), cs = cos(ys), sc = sin(yc), cc = cos(yc);
            dmean[a] += (double)scale * e * (gs * cs + gc * cc);
            dcov[a] += -0.5 * (double)scale * (double)scale * e * (gs * ss + gc * sc);
        }
    }
    // lift_gaussian (diagonal): mean_a = o_a + d_a t_mean ; cov_a = t_var d_a^2 + r_var (1 - d_a^2 / dn)
    const double dd[3] = {(double)d[0] * d[0], (double)d[1] * d[1], (double)d[2] * d[2]};
    const double dn = dd[0] + dd[1] + dd[2] + 1e-10;
    double g_tmean = 0, g_tvar = 0, g_rvar = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        g_tmean += dmean[a] * d[a];
        if (!disable_integration) {
            g_tvar += dcov[a] * dd[a];
            g_rvar += dcov[a] * (1.0 - dd[a] / dn);
        }
    }
    // conical_frustum_to_gaussian (stable form): derivatives w.r.t. mu = (t0 + t1)/2 and hw = (t1 - t0)/2
    const double mu = 0.5 * ((double)t0 + t1), hw = 0.5 * ((double)t1 - t0);
    const double mu2 = mu * mu, hw2 = hw * hw, hw3 = hw2 * hw, hw4 = hw2 * hw2, hw5 = hw4 * hw;
    const double D = 3.0 * mu2 + hw2, D2 = D * D, D3 = D2 * D;
    const double tm_mu = 1.0 + 2.0 * hw2 / D - 12.0 * mu2 * hw2 / D2;
    const double tm_hw = 4.0 * mu * hw / D - 4.0 * mu * hw3 / D2;
    const double Nn = hw4 * (12.0 * mu2 - hw2);
    const double tv_mu = -(4.0 / 15.0) * (24.0 * mu * hw4 / D2 - 12.0 * mu * Nn / D3);
    const double tv_hw = 2.0 * hw / 3.0 - (4.0 / 15.0) * ((48.0 * mu2 * hw3 - 6.0 * hw5) / D2 - 4.0 * hw * Nn / D3);
    const double r2 = (double)radius * radius;
    const double rv_mu = r2 * (0.5 * mu + (8.0 / 5.0) * mu * hw4 / D2);
    const double rv_hw = r2 * (5.0 * hw / 6.0 - (4.0 / 15.0) * (4.0 * hw3 / D - 2.0 * hw5 / D2));
    const double g_mu = g_tmean * tm_mu + g_tvar * tv_mu + g_rvar * rv_mu;
    const double g_hw = g_tmean * tm_hw + g_tvar * tv_hw + g_rvar * rv_hw;
    // each fence post collects exactly two contributions (its two intervals) + whatever the buffer held: commutative
    atomicAdd(d_t + b * (int64_t)(N + 1) + i, (float)(0.5 * g_mu - 0.5 * g_hw));
    atomicAdd(d_t + b * (int64_t)(N + 1) + i + 1, (float)(0.5 * g_mu + 0.5 * g_hw));
}

// One wavefront per ray; mirrors k_piecewise_constant_pdf<K, BLUR = true> (kernels_ray.hip) up to the CDF, then walks the
// reference's autograd backwards: t'_j = b0 + (u_j - c0)/(c1 - c0) (b1 - b0) -> cdf -> cumsum -> pdf = v / sum v -> v = blur + pad ->
// blur = (max(w_{i-1}, w_i) + max(w_i, w_{i+1})) / 2 with torch.maximum's tie rule (half each).
constexpr int kMaxBins = 1024;     // N <= 1024 (MIPNERF_MAX_SAMPLES); LDS rows: 512 entries for K <= 8, 1024 for the K = 16 bucket
template <int K> struct GradRow { static constexpr int kBins = K <= 8 ? 512 : kMaxBins; };
template <int K>
__global__ void __launch_bounds__(64)
k_resample_bwd(int64_t B, int N, const float* __restrict__ bins, const float* __restrict__ weights,
               const float* __restrict__ u_rand, float padding, float u_step, float u_jitter,
               const float* __restrict__ d_t_new, float* __restrict__ d_weights) {
    __shared__ float s_w[GradRow<K>::kBins + 2];
    __shared__ float s_cdf[GradRow<K>::kBins + 2];
    __shared__ float s_bins[GradRow<K>::kBins + 2];
    __shared__ float s_g0[GradRow<K>::kBins + 2];     // gradient w.r.t. cdf[i] from the draws whose lower entry is i
    __shared__ float s_g1[GradRow<K>::kBins + 2];     // ... w.r.t. cdf[i + 1] from the same draws
    __shared__ float s_dv[GradRow<K>::kBins + 2];
    const int lane = threadIdx.x;
    const int64_t b = blockIdx.x;
    const float* wb = weights + b * (int64_t)N;
    const float* binb = bins + b * (int64_t)(N + 1);
    const int i0 = lane * K;
    const int nd = N + 1;                    // draws
#pragma unroll
    for (int k = 0; k < K; ++k)
        if (i0 + k < N) s_w[i0 + k] = wb[i0 + k];
    for (int j = lane; j <= N; j += 64) s_bins[j] = binb[j];
    __syncthreads();
    float v[K];
    double run = 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int i = i0 + k;
        float x = 0.0f;
        if (i < N) {
            const float wc = s_w[i], wl = s_w[i > 0 ? i - 1 : 0], wr = s_w[i < N - 1 ? i + 1 : N - 1];
            x = 0.5f * (fmaxf(wl, wc) + fmaxf(wc, wr)) + padding;
        }
        v[k] = x;
        run += (double)x;
    }
    float wsum = (float)wsum_d(run);
    const float pad = fmaxf(0.0f, 1e-5f - wsum);
    const float padn = pad / (float)N;
    wsum += pad;
    float pdf[K];
    double pre[K];
    run = 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        pdf[k] = (i0 + k < N) ? (v[k] + padn) / wsum : 0.0f;
        pre[k] = run;
        run += (double)pdf[k];
    }
    const double off = wexcl_prefix_d(run, lane);
    float cs[K];                             // cumsum before the min(1, .)
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int i = i0 + k;
        cs[k] = (float)(off + pre[k]);
        if (i < N) s_cdf[i] = (i == 0) ? 0.0f : fminf(1.0f, cs[k]);
    }
    if (lane == 0) s_cdf[N] = 1.0f;
    __syncthreads();

    const float umax = 1.0f - 1.1920928955078125e-07f;
    auto draw = [&](int j) -> float {
        if (u_rand != nullptr) return fminf((float)j * u_step + u_rand[b * (int64_t)nd + j] * u_jitter, umax);
        return torch_linspace_at(0.0f, umax, nd, j);
    };
    // interval i = [cdf[i], cdf[i+1]) receives the draws j with searchsorted(cdf, u_j, right) - 1 == i; u is non-decreasing in
    // j, so they are the contiguous range [first j with u_j >= cdf[i] ..., first j with u_j >= cdf[i+1]) -- found by bisection,
    // summed in j order: deterministic, no atomics.  (searchsorted right: entries <= u count, so u == cdf[i+1] belongs to i+1.)
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int i = i0 + k;
        if (i >= N) continue;
        auto first_ge = [&](float c) {        // smallest j in [0, nd] with u_j >= c
            int lo = 0, hi = nd;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (draw(mid) >= c) hi = mid; else lo = mid + 1;
            }
            return lo;
        };
        // interval membership must replicate the forward's `cdf[mid] <= u` walk exactly, duplicates in the cdf included: draw j
        // belongs to the LAST index i with cdf[i] <= u_j.  If cdf[i+1] == cdf[i] the interval i is empty.
        const float c0 = s_cdf[i], c1 = s_cdf[i + 1];
        const int jb = (i == 0) ? 0 : first_ge(c0);
        const int je = (i + 1 == N) ? nd : first_ge(c1);
        float g0 = 0.0f, g1 = 0.0f;
        const float den_raw = c1 - c0;
        const float db = s_bins[i + 1] - s_bins[i];
        for (int j = jb; j < je; ++j) {
            const float u = draw(j);
            const float dtt = d_t_new[b * (int64_t)nd + j] * db;
            if (den_raw < 1e-5f) {            // denom replaced by the constant 1 (mip.py:226): t = u - c0
                g0 += -dtt;
            } else {
                const float inv2 = 1.0f / (den_raw * den_raw);
                g0 += dtt * (u - c1) * inv2;
                g1 += -dtt * (u - c0) * inv2;
            }
        }
        s_g0[i] = g0;
        s_g1[i] = g1;
    }
    __syncthreads();
    // d cdf[i] = g0[i] + g1[i-1] for i = 1..N-1 (cdf[0] = 0 and cdf[N] = 1 are constants); through min(1, cs): torch.minimum
    // passes the gradient where cs < 1, half of it at a tie, nothing above
    double dcs[K];
    run = 0.0;
    double sufpre[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int i = i0 + k;
        double gci = 0.0;
        if (i >= 1 && i < N) {
            gci = (double)s_g0[i] + (double)s_g1[i - 1];
            gci *= cs[k] < 1.0f ? 1.0 : (cs[k] == 1.0f ? 0.5 : 0.0);
        }
        dcs[k] = gci;
    }
    // cs_i = sum_{m < i} pdf_m  ->  d pdf_m = sum_{i = m+1}^{N-1} dcs_i  (exclusive suffix sum)
    double lanesum = 0.0;
#pragma unroll
    for (int k = K - 1; k >= 0; --k) {
        sufpre[k] = lanesum;
        lanesum += dcs[k];
    }
    const double tot = wsum_d(lanesum);
    const double before = wexcl_prefix_d(lanesum, lane);          // sum over lower lanes
    const double after = tot - before - lanesum;                   // sum over higher lanes
    double dpdf[K];
    double dot = 0.0, sumd = 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        dpdf[k] = (i0 + k < N) ? after + sufpre[k] : 0.0;
        dot += dpdf[k] * (double)pdf[k];
        sumd += dpdf[k];
    }
    dot = wsum_d(dot);
    sumd = wsum_d(sumd);
    // pdf_i = v_i / S.  pad == 0: S = sum v.  pad > 0 (sum v < 1e-5): v_i += (1e-5 - sum v) / N and S = 1e-5 (a constant)
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int i = i0 + k;
        if (i < N) s_dv[i] = (float)(pad > 0.0f ? (dpdf[k] - sumd / (double)N) / (double)wsum : (dpdf[k] - dot) / (double)wsum);
    }
    __syncthreads();
    // blur pool backward (gather form): w_c wins / ties / loses each of the maxima it takes part in
    auto share = [](float mine, float other) { return mine > other ? 1.0f : (mine == other ? 0.5f : 0.0f); };
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int c = i0 + k;
        if (c >= N) continue;
        const float wc = s_w[c];
        const float wl = s_w[c > 0 ? c - 1 : 0], wr = s_w[c < N - 1 ? c + 1 : N - 1];
        float g = 0.5f * s_dv[c] * (share(wc, wl) + share(wc, wr));                  // as the centre of v_c
        if (c == 0) g += 0.5f * s_dv[0] * share(wl, wc);                              // ... and as its own clamped left neighbour
        if (c == N - 1) g += 0.5f * s_dv[N - 1] * share(wr, wc);                      // ... / right neighbour
        if (c + 1 < N) g += 0.5f * s_dv[c + 1] * share(wc, s_w[c + 1]);              // as the left neighbour of v_{c+1}
        if (c >= 1) g += 0.5f * s_dv[c - 1] * share(wc, s_w[c - 1]);                  // as the right neighbour of v_{c-1}
        d_weights[b * (int64_t)N + c] = g;
    }
}
}  // namespace

hipError_t launch_cast_ipe_bwd(int64_t B, int N, int min_deg, int max_deg, int disable_integration, const float* t,
                               const float* origins, const float* dirs, const float* radii, const float* d_enc, float* d_t,
                               hipStream_t st) {
    const int64_t M = B * (int64_t)N;
    hipLaunchKernelGGL(k_cast_ipe_bwd, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, B, N, min_deg, max_deg - min_deg,
                       disable_integration, t, origins, dirs, radii, d_enc, d_t);
    return hipGetLastError();
}

hipError_t launch_resample_bwd(int64_t B, int N, const float* bins, const float* weights, const float* u_rand, float padding,
                               const float* d_t_new, float* d_weights, hipStream_t st) {
    if (N > kMaxBins || N < 1) return hipErrorInvalidValue;
    const int n_draws = N + 1;
    const double s = 1.0 / (double)n_draws;
    const float u_step = (float)s, u_jitter = (float)(s - (double)1.1920928955078125e-07f);
    const int K = (N + 63) / 64;
#define MIP_RB(KK) hipLaunchKernelGGL((k_resample_bwd<KK>), dim3((unsigned)B), dim3(64), 0, st, B, N, bins, weights, u_rand, padding, \
                                      u_step, u_jitter, d_t_new, d_weights)
    switch (K) {
        case 1: MIP_RB(1); break;
        case 2: MIP_RB(2); break;
        case 3: case 4: MIP_RB(4); break;
        case 5: case 6: case 7: case 8: MIP_RB(8); break;
        case 9: case 10: case 11: case 12: case 13: case 14: case 15: case 16: MIP_RB(16); break;
        default: return hipErrorInvalidValue;
    }
#undef MIP_RB
    return hipGetLastError();
}

}  // namespace mip

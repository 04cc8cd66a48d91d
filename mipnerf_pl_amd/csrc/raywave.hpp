// Per-ray device functions shared by the ray-side kernels (kernels_ray.hip, kernels_train.hip): wave64 scan / reduction helpers,
// volumetric_rendering of one ray (models/mip.py:366-401), the inverse-CDF sampler of one ray (mip.py:168-229, 252-257) and the
// distortion loss of one ray (mip.py:8-20).  The stand-alone kernels and the fused ones (k_composite_resample,
// k_composite_train) call the SAME functions, so every route gives the same bits.  Compile with -ffp-contract=off (raymath.hpp).
#pragma once

#include <hip/hip_runtime.h>

#include "raymath.hpp"

namespace mip {

// ------------------------------------------------------------------------------------------
// wave64 helpers
// ------------------------------------------------------------------------------------------
// Round 5: cross-lane movement by DPP (v_mov_b32_dpp: row_shr / row_bcast, a few cycles each) instead of __shfl (ds_bpermute_b32
// through the LDS crossbar, ~100 cycles of latency per step, two per double).  Every scan / reduction of the ray-side kernels
// sits on a per-ray dependent chain (compositing -> CDF -> search), so the latency is what they cost.  The DPP forms associate the
// sums differently from the shuffle trees (row-local prefix, then row totals); the operands are fp32 values widened to double, so
// the double sums are exact except where magnitudes differ by > 2^29 and the fp32 roundings of the results agree with the shuffle
// forms except in such corner cases (every golden / fused-vs-stage test runs on these).  MIP_WAVE_DPP=0 restores the shuffles.
#ifndef MIP_WAVE_DPP
#define MIP_WAVE_DPP 1
#endif
// wave_shr:1 (0x138) and row_bcast:15 / :31 (0x142 / 0x143) exist on the GCN / CDNA family only: a device pass for any other
// architecture falls back to the shuffle forms instead of failing to assemble (this tree builds gfx950 only; the guard costs nothing)
#if MIP_WAVE_DPP && defined(__HIP_DEVICE_COMPILE__) && !defined(__GFX9__)
#undef MIP_WAVE_DPP
#define MIP_WAVE_DPP 0
#endif

#if MIP_WAVE_DPP
// lane i <- lane (i - n) within its row of 16 (row_shr:n), lanes without a source get `ident`
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f32(float ident, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(ident), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {      // identity 0.0
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
constexpr int kDppRowShr1 = 0x111, kDppRowShr2 = 0x112, kDppRowShr4 = 0x114, kDppRowShr8 = 0x118, kDppRowBcast15 = 0x142,
              kDppRowBcast31 = 0x143, kDppWaveShr1 = 0x138;

// inclusive prefix sum over the 64 lanes (lane 63 holds the total)
__device__ __forceinline__ float wave_incl_scan_dpp(float v) {
    v += dpp_f32<kDppRowShr1, 0xf>(0.0f, v);
    v += dpp_f32<kDppRowShr2, 0xf>(0.0f, v);
    v += dpp_f32<kDppRowShr4, 0xf>(0.0f, v);
    v += dpp_f32<kDppRowShr8, 0xf>(0.0f, v);
    v += dpp_f32<kDppRowBcast15, 0xa>(0.0f, v);      // lane 15 of rows 0 / 2 -> every lane of rows 1 / 3
    v += dpp_f32<kDppRowBcast31, 0xc>(0.0f, v);      // lane 31 -> every lane of rows 2, 3
    return v;
}
__device__ __forceinline__ double wave_incl_scan_dpp(double v) {
    v += dpp_f64<kDppRowShr1, 0xf>(v);
    v += dpp_f64<kDppRowShr2, 0xf>(v);
    v += dpp_f64<kDppRowShr4, 0xf>(v);
    v += dpp_f64<kDppRowShr8, 0xf>(v);
    v += dpp_f64<kDppRowBcast15, 0xa>(v);
    v += dpp_f64<kDppRowBcast31, 0xc>(v);
    return v;
}
__device__ __forceinline__ float readlane63(float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63)); }
__device__ __forceinline__ double readlane63(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

__device__ __forceinline__ float wave_sum(float v) { return readlane63(wave_incl_scan_dpp(v)); }

// exclusive prefix sum across the 64 lanes; *total = sum over all lanes
__device__ __forceinline__ float wave_excl_scan(float v, int lane, float* total) {
    const float inc = wave_incl_scan_dpp(v);
    *total = readlane63(inc);
    return dpp_f32<kDppWaveShr1, 0xf>(0.0f, inc);     // lane i <- lane i - 1 across the whole wave, lane 0 <- 0
}

// double-precision variants: torch's CPU cumsum accumulates float32 in double
// (at::acc_type<float,false>), and the inverse-CDF / transmittance are sensitive to the prefix sums
// (a 1e-7 error of the CDF moves a resampled t by 1e-5 where the pdf is ~5e-4), so the scans run in
// fp64 -- a few dozen DP adds per ray on a chip with full-rate fp64.
__device__ __forceinline__ double wave_sum_f64(double v) { return readlane63(wave_incl_scan_dpp(v)); }
__device__ __forceinline__ double wave_excl_scan_f64(double v, int lane) {
    return dpp_f64<kDppWaveShr1, 0xf>(wave_incl_scan_dpp(v));
}
#else
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// exclusive prefix sum across the 64 lanes; *total = sum over all lanes
__device__ __forceinline__ float wave_excl_scan(float v, int lane, float* total) {
    float inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float n = __shfl_up(inc, o, 64);
        if (lane >= o) inc += n;
    }
    *total = __shfl(inc, 63, 64);
    const float ex = __shfl_up(inc, 1, 64);
    return lane == 0 ? 0.0f : ex;
}

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_excl_scan_f64(double v, int lane) {
    double inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double n = __shfl_up(inc, o, 64);
        if (lane >= o) inc += n;
    }
    const double ex = __shfl_up(inc, 1, 64);
    return lane == 0 ? 0.0 : ex;
}
#endif


// The per-ray bodies are device functions shared by the stand-alone kernels and by the fused k_composite_resample (one launch
// for "composite level 0, then draw the fine level's fence posts from its weights": same arithmetic, same order, same bits).
template <int K>
__device__ __forceinline__ void composite_ray(bool active, int lane, int N, const float4* __restrict__ cb, const float* __restrict__ tb,
                                              float dn, int white_bkgd, float* __restrict__ comp_rgb_b, float* __restrict__ distance_b,
                                              float* __restrict__ acc_b, float* __restrict__ weights_b, float (&w_out)[K]) {
    const int i0 = lane * K;
    float tv[K + 1];
#pragma unroll
    for (int k = 0; k <= K; ++k) tv[k] = (i0 + k <= N) ? tb[i0 + k] : 0.0f;
    float4 c[K];
    float dd[K];
    double pre[K];
    double run = 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const bool ok = i0 + k < N;
        c[k] = ok ? cb[i0 + k] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float delta = (tv[k + 1] - tv[k]) * dn;
        dd[k] = ok ? c[k].w * delta : 0.0f;   // density_delta
        pre[k] = run;
        run += (double)dd[k];
    }
    const double off = wave_excl_scan_f64(run, lane);

    float sr = 0.f, sg = 0.f, sb = 0.f, sa = 0.f, sd = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const bool ok = i0 + k < N;
        const float alpha = 1.0f - expf(-dd[k]);
        const float trans = expf(-(float)(off + pre[k]));   // exclusive cumsum rounded to fp32 like torch
        const float w = ok ? alpha * trans : 0.0f;
        w_out[k] = w;
        if (ok && active && weights_b) weights_b[i0 + k] = w;
        sr += w * c[k].x;
        sg += w * c[k].y;
        sb += w * c[k].z;
        sa += w;
        sd += w * (0.5f * (tv[k] + tv[k + 1]));
    }
    sr = wave_sum(sr); sg = wave_sum(sg); sb = wave_sum(sb); sa = wave_sum(sa); sd = wave_sum(sd);
    if (lane == 0 && active) {
        const float tnear = tb[0], tfar = tb[N];
        float dist = nan_to_num(sd);
        dist = fminf(fmaxf(dist, tnear), tfar);   // torch.clamp(x, min, max) = min(max(x,min),max)
        if (white_bkgd) {
            const float bg = 1.0f - sa;
            sr += bg; sg += bg; sb += bg;
        }
        comp_rgb_b[0] = sr; comp_rgb_b[1] = sg; comp_rgb_b[2] = sb;
        *distance_b = dist;
        *acc_b = sa;
    }
}


// ------------------------------------------------------------------------------------------
// resample_along_rays t part (models/mip.py:232-280) and sorted_piecewise_constant_pdf
// (models/mip.py:168-229).  One wavefront per ray; the ray's CDF and bins live in LDS; each
// lane inverts the CDF for its draws with a binary search (torch.searchsorted right=True).
//   BLUR = true : weights are blur-pooled and `padding` added first (resample path)
//   BLUR = false: weights used as given
// ------------------------------------------------------------------------------------------
constexpr int kPdfMaxBins = 1024;     // N <= 1024 = 64 lanes x 16 samples (MIPNERF_MAX_SAMPLES)
constexpr int kRaysPerBlock = 4;
// LDS row length of a kernel instantiated for K samples per lane: the K <= 8 buckets keep their 512-entry rows (24 KiB per block: six blocks per CU);
// only the K = 16 bucket (512 < N <= 1024) pays for 1024-entry rows
template <int K> struct PdfRow { static constexpr int kBins = K <= 8 ? 512 : kPdfMaxBins; };

// s_w / s_bins hold the ray's weights [N] and bins [N+1] (staged by the caller, block barrier done); every wave of the block
// must call this (it contains block barriers); out_row = nullptr: no stores (a wave shadowing the last ray)
template <int K, bool BLUR>
__device__ __forceinline__ void pdf_ray(int lane, int N, const float* __restrict__ s_w, float* __restrict__ s_cdf,
                                        const float* __restrict__ s_bins, int n_draws, const float* __restrict__ u_row,
                                        float padding, float u_step, float u_jitter, float* __restrict__ out_row) {
    const int i0 = lane * K;
    float w[K];
    double run = 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int i = i0 + k;
        float v = 0.0f;
        if (i < N) {
            if (BLUR) {
                // weights_pad = [w0, w, w_{N-1}]; max of neighbours; mean of neighbours (mip.py:252-254)
                const float wc = s_w[i];
                const float wl = s_w[i > 0 ? i - 1 : 0];
                const float wr = s_w[i < N - 1 ? i + 1 : N - 1];
                v = 0.5f * (fmaxf(wl, wc) + fmaxf(wc, wr)) + padding;
            } else {
                v = s_w[i];
            }
        }
        w[k] = v;
        run += (double)v;
    }
    // eps padding so the sum is >= 1e-5 (mip.py:181-185)
    float wsum = (float)wave_sum_f64(run);
    const float pad = fmaxf(0.0f, 1e-5f - wsum);
    const float padn = pad / (float)N;
    wsum += pad;
    double pre[K];
    run = 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int i = i0 + k;
        const float pdf = (i < N) ? (w[k] + padn) / wsum : 0.0f;
        pre[k] = run;
        run += (double)pdf;
    }
    const double off = wave_excl_scan_f64(run, lane);
    // cdf = [0, min(1, cumsum(pdf[:-1])), 1]  (mip.py:190-195): cdf[i] = min(1, sum_{j<i} pdf_j)
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int i = i0 + k;
        if (i < N) s_cdf[i] = (i == 0) ? 0.0f : fminf(1.0f, (float)(off + pre[k]));
    }
    if (lane == 0) s_cdf[N] = 1.0f;
    __syncthreads();

    const float eps32 = 1.1920928955078125e-07f;
    const float umax = 1.0f - eps32;
    // Round 5: a lane's draws (j = lane, lane + 64, ...) are searched TOGETHER, T = K + 1 at a time -- the resampling path has N + 1 draws, i.e.
    // K full trips plus one draw, and each binary search is a chain of ~log2(N) dependent LDS reads: one after the other they cost (K + 1)
    // chains, interleaved one.  Same comparisons per draw, fixed step count with predicated updates: the same bits as the sequential loop.
    constexpr int T = K + 1;
    const int steps = 32 - __builtin_clz((unsigned)(N + 1));          // a range of N + 1 entries is empty after at most that many halvings
    for (int j0 = lane; j0 < n_draws; j0 += 64 * T) {
        float u[T];
        int lo[T], hi[T];
#pragma unroll
        for (int q = 0; q < T; ++q) {
            const int j = j0 + 64 * q;
            const int jc = j < n_draws ? j : n_draws - 1;             // a lane without a draw in this trip shadows the last one (not stored)
            if (u_row != nullptr) {
                // u = arange*s + U[0, s-eps), clipped to 1-eps (mip.py:198-204)
                u[q] = fminf((float)jc * u_step + u_row[jc] * u_jitter, umax);
            } else {
                u[q] = torch_linspace_at(0.0f, umax, n_draws, jc);   // mip.py:207
            }
            lo[q] = 0;
            hi[q] = N + 1;
        }
        // searchsorted(cdf, u, right=True): number of entries <= u, over cdf[0..N]
        for (int it = 0; it < steps; ++it) {
#pragma unroll
            for (int q = 0; q < T; ++q) {
                // branch-free (selects): the T reads of a step are issued together; a finished search (lo == hi) re-reads an entry and keeps its range
                const int mid = (lo[q] + hi[q]) >> 1;
                const float cm = s_cdf[min(mid, N)];
                const bool go = lo[q] < hi[q], le = cm <= u[q];
                lo[q] = (go && le) ? mid + 1 : lo[q];
                hi[q] = (go && !le) ? mid : hi[q];
            }
        }
        float c0[T], c1[T], b0[T], b1[T];
#pragma unroll
        for (int q = 0; q < T; ++q) {                  // the 4 T interval reads of the trip together (pinned: the compiler would sink
            const int below = max(0, lo[q] - 1);       // them into the store's branch, one dependent LDS round trip per draw)
            const int above = min(N, lo[q]);
            c0[q] = s_cdf[below]; c1[q] = s_cdf[above];
            b0[q] = s_bins[below]; b1[q] = s_bins[above];
        }
#pragma unroll
        for (int q = 0; q < T; ++q) asm volatile("" : "+v"(c0[q]), "+v"(c1[q]), "+v"(b0[q]), "+v"(b1[q]));
#pragma unroll
        for (int q = 0; q < T; ++q) {
            const int j = j0 + 64 * q;
            float denom = c1[q] - c0[q];
            denom = (denom < 1e-5f) ? 1.0f : denom;
            const float tt = (u[q] - c0[q]) / denom;
            if (out_row && j < n_draws) out_row[j] = b0[q] + tt * (b1[q] - b0[q]);
        }
    }
}


// distloss (models/mip.py:8-20) of one ray in O(N), forward and (optionally) backward: one wavefront, lane owns K consecutive
// samples; the weights come in REGISTERS (from global memory in k_distloss, straight from the compositing in the fused
// training kernel).  ray_loss_b / d_w_b / d_t_b: this ray's outputs or nullptr; g = the upstream gradient of ray_loss.
template <int K>
__device__ __forceinline__ void distloss_ray(int lane, int N, const float (&w_in)[K], const float* __restrict__ tb,
                                             float* __restrict__ ray_loss_b, float g, float* __restrict__ d_w_b,
                                             float* __restrict__ d_t_b) {
    const int i0 = lane * K;
    float w[K], m[K], iv[K];      // w: a copy, so that the caller's registers are not modified
    double pP[K], pQ[K];
    double rP = 0.0, rQ = 0.0, uni = 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const bool ok = i0 + k < N;
        const float t0 = ok ? tb[i0 + k] : 0.f, t1 = ok ? tb[i0 + k + 1] : 0.f;
        w[k] = ok ? w_in[k] : 0.f;
        m[k] = (t1 + t0) * 0.5f;
        iv[k] = t1 - t0;
        pP[k] = rP; pQ[k] = rQ;
        rP += (double)w[k];
        rQ += (double)w[k] * (double)m[k];
        uni += (double)(iv[k] * w[k] * w[k]);
    }
    const double oP = wave_excl_scan_f64(rP, lane), oQ = wave_excl_scan_f64(rQ, lane);
    const double totP = wave_sum_f64(rP), totQ = wave_sum_f64(rQ);
    double bi = 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k) bi += (double)w[k] * ((double)m[k] * (oP + pP[k]) - (oQ + pQ[k]));
    const double tot = wave_sum_f64(uni) / 3.0 + 2.0 * wave_sum_f64(bi);
    if (lane == 0 && ray_loss_b) *ray_loss_b = (float)tot;
    if (d_w_b) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (i0 + k < N) {
                const double P = oP + pP[k], Q = oQ + pQ[k];
                const double wi = w[k], mi = m[k];
                // sum_j w_j |m_i - m_j| = m_i P - Q + (Qtot - Q - w_i m_i) - m_i (Ptot - P - w_i)
                const double sj = mi * P - Q + (totQ - Q - wi * mi) - mi * (totP - P - wi);
                d_w_b[i0 + k] = g * (float)((2.0 / 3.0) * iv[k] * wi + 2.0 * sj);
            }
        }
    }
    if (d_t_b) {
        // interval_i = t_{i+1} - t_i, m_i = (t_i + t_{i+1}) / 2:  dL/dinterval_i = w_i^2 / 3,
        // dL/dm_i = 2 w_i (sum_{j<i} w_j - sum_{j>i} w_j)  (t sorted)  ->  d_t[i] = (A_{i-1} - A_i) + (C_{i-1} + C_i) / 2
        float A[K], C[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const bool ok = i0 + k < N;
            const double P = oP + pP[k], wi = w[k];
            A[k] = ok ? g * (float)(wi * wi / 3.0) : 0.f;
            C[k] = ok ? g * (float)(2.0 * wi * (2.0 * P + wi - totP)) : 0.f;
        }
        float pa = __shfl_up(A[K - 1], 1, 64), pc = __shfl_up(C[K - 1], 1, 64);
        if (lane == 0) { pa = 0.f; pc = 0.f; }
#pragma unroll
        for (int k = 0; k <= K; ++k) {
            const int i = i0 + k;
            const float ak = k < K ? A[k < K ? k : 0] : 0.f, ck = k < K ? C[k < K ? k : 0] : 0.f;
            if (i <= N && (k < K || i == N)) d_t_b[i] = (pa - ak) + 0.5f * (pc + ck);
            if (k < K) { pa = A[k]; pc = C[k]; }
        }
    }
}

}  // namespace mip

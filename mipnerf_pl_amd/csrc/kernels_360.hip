// Unbounded-scene (mip-NeRF 360) ray path for gfx950: s-space sampling and the fused
// frustum -> full-covariance Gaussian -> contraction -> off-axis integrated positional encoding kernel
// (see raymath360.hpp for the math and why it follows the paper rather than the reference's dead code,
// models/mip.py:106-124, 292-319, 424-447).  HBM-bound elementwise work: one thread per (sample, basis direction),
// a wave writes runs of 21 consecutive features per frequency.
#include <hip/hip_runtime.h>

#include "kernels.hpp"
#include "raymath360.hpp"

namespace mip {

// fence posts uniform in normalised inverse depth (paper eq. (11)-(13), g(t) = 1/t); t_rand: stratified jitter
// between the midpoints in inverse-depth space (models/mip.py:113-118)
__global__ void __launch_bounds__(256)
k_sample_along_rays_360(int64_t B, int N, const float* __restrict__ nearp, const float* __restrict__ farp,
                        const float* __restrict__ t_rand, float* __restrict__ t_inv_out, float* __restrict__ t_out) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= B * (int64_t)(N + 1)) return;
    const int64_t b = gid / (N + 1);
    const int i = (int)(gid - b * (N + 1));
    const float ni = 1.0f / nearp[b], fi = 1.0f / farp[b];
    auto post = [&](int k) {
        const float s = torch_linspace_at(0.0f, 1.0f, N + 1, k);
        return fi * s + (1.0f - s) * ni;
    };
    float ti = post(i);
    if (t_rand) {
        const float lo = i == 0 ? ti : 0.5f * (ti + post(i - 1));
        const float up = i == N ? ti : 0.5f * (post(i + 1) + ti);
        ti = lo + (up - lo) * t_rand[gid];
    }
    t_inv_out[gid] = ti;
    t_out[gid] = 1.0f / ti;
}

// Where feature f of sample s goes.  Row-major [M, F], or (frag != 0, bf16 only) the MFMA B-operand fragments k_pre_gemm reads
// (gen_pre_gemm.py): [wave tile of 32 samples][k-step of 16 features][lane (f / 8 % 2, s % 32)][f % 8] -- one lane-linear 1-KiB
// fragment per wave tile and k-step.
__device__ __forceinline__ int64_t enc360_index(int64_t s, int f, int F, int frag) {
    if (!frag) return s * F + f;
    return ((s >> 5) * (F >> 4) + (f >> 4)) * 512 + ((f >> 3) & 1) * 256 + (s & 31) * 8 + (f & 7);
}

// Accuracy policy per output type, as in kernels_ray.hip: float rows feed the exact-fp32 MLP (parity mode) and use the accurate libm
// sin / exp; bf16 rows are rounded to 8 bits anyway and use the fast pair -- in BOTH layouts, so the fragment route of mipnerf_forward
// and the row-major route of the per-stage API hand the MLP the same bits.
template <typename OutT> struct Ipe360Math;
template <> struct Ipe360Math<float> {
    // sin(x) and "cos" = sin(fl32(x + fl32(pi/2))), the reference's form (mip.py:349-350)
    __device__ static void sincos(float x, float& s, float& c) { s = sin_accurate(x); c = sin_accurate(x + kHalfPiF); }
    __device__ static float damp(float vs) { return exp_accurate(-0.5f * vs); }       // exp(-0.5 * 4^l var), the reference's expression
};
template <> struct Ipe360Math<__bf16> {
    // ONE range reduction for the pair, in two-float fp32 arithmetic (x is up to 2^16 rad at the top degree; the fp64 form of sin_fast costs
    // five half-rate instructions per sine): 1 / 2 pi = hi + lo, p = fl(x hi), e = x hi - p exactly (fma), t = fl(x lo + e), and
    // frac(x / 2 pi) = (p - rint(p)) + t -- the subtraction is exact; measured against fp64 on 2e5 arguments up to 2^17: 2.6e-8 turns.  Then the
    // hardware sine of r and of r + 1/4 turn.  The "cos" here is the true cosine of x; the reference's fl32(x + pi/2) differs from it by
    // at most half an fp32 ulp of x (4e-3 rad at 2^16 rad: one bf16 ulp of a feature, and only where the damping has not removed it).
    __device__ static void sincos(float x, float& s, float& c) {
        constexpr float kHi = 0.15915494f, kLo = 6.4206382e-09f;
        const float p = x * kHi;
        const float e = __builtin_fmaf(x, kHi, -p);
        const float t = __builtin_fmaf(x, kLo, e);
        const float rf = (p - __builtin_rintf(p)) + t;
        s = __builtin_amdgcn_sinf(rf);
        c = __builtin_amdgcn_sinf(rf + 0.25f);
    }
    __device__ static float damp(float vs) { return __builtin_amdgcn_exp2f(vs * -0.72134752f); }      // exp(-0.5 vs) = 2^(-0.5 log2(e) vs): one multiply
};
// off-axis IPE feature pair (l, basis j) from the projection (y, var): exp(-0.5 * 4^l var) * (sin(2^l y), sin(2^l y + pi/2));
// the expressions of raymath360.hpp:ipe360_feature
template <typename OutT>
__device__ __forceinline__ void ipe360_pair(float y, float var, int l, int min_deg, OutT& fs, OutT& fc) {
    // 2^e and 4^e assembled from the exponent (the same values as (float)(1u << e) and its square, e <= 31, without a conversion per feature;
    // with a wave-uniform l they stay in scalar registers)
    const int e = l + min_deg;
    const float scale = __builtin_bit_cast(float, (e + 127) << 23);
    const float ys = y * scale;
    const float vs = var * __builtin_bit_cast(float, (2 * e + 127) << 23);
    const float damp = Ipe360Math<OutT>::damp(vs);
    float sn, cs;
    Ipe360Math<OutT>::sincos(ys, sn, cs);
    fs = (OutT)(damp * sn);
    fc = (OutT)(damp * cs);
}

// The encoding of a TILE of 64 samples per workgroup (what mipnerf_forward and the per-stage entry point use when only the encoding is asked for).
// The Gaussian of a sample is formed ONCE (k_cast_ipe_360 below recomputes it in each of its 21 threads) by one full wave -- which wave rotates
// with the workgroup index, so that the serial part does not always land on the same SIMD --, its 21 projections go through LDS, and every thread
// then writes whole vectors of eight consecutive features, pairing the "sin" vector of eight (degree, direction) features with its "cos" vector
// 21 L features later (same damping factor).  Needs 21 * L to be a multiple of 8.  Same per-feature expressions as k_cast_ipe_360: same bits.
//   FRAG (bf16 only): the MFMA B-operand fragments of enc360_index -- lane n of a vector = sample n, so a wave's stores are lane-linear; the
//     buffer covers whole 256-sample tiles (an even number of wave tiles), samples past the end repeat the last one;
//   rows: row-major [M, F]: consecutive threads write consecutive 8-feature runs of one sample's row.
constexpr int kFragSamples = 64;
// LDS row of a sample's projections = 21 + PAD floats.  PAD = 7 is what the wrap-around needs; with it a wave's 64 lanes (one sample each)
// read rows 28 floats apart = 16 distinct banks: 4-way conflicts, 66 % of the kernel's LDS-active cycles (profiles/r05_lds_conflicts.txt).
// An odd row length (PAD = 8) makes the 64 rows start in 64 different banks -- measured, no difference: 7.78-7.81 vs 7.79-7.82 ms per unbounded
// bf16 forward; the kernel is bound by its 5.1 TB/s of stores, the conflicts hide behind them.  Knob kept (build.py: MLP_IPE360_ROW_PAD).
#ifndef MIP_IPE360_ROW_PAD
#define MIP_IPE360_ROW_PAD 7
#endif
template <typename OutT, bool FRAG>
__global__ void __launch_bounds__(256)
k_cast_ipe_360_tile(int64_t B, int N, int min_deg, int L, int contracted, const float* __restrict__ t, const float* __restrict__ origins,
                    const float* __restrict__ dirs, const float* __restrict__ radii, OutT* __restrict__ enc) {
    typedef OutT vec8 __attribute__((ext_vector_type(8)));
    // projections of a sample: its 21 directions followed by the first 7 again, so that the eight consecutive features of a vector
    // (directions j0 .. j0 + 7, wrapping into the next degree) are eight consecutive LDS words -- one address per vector, literal offsets
    constexpr int kRow = kBasis360N + MIP_IPE360_ROW_PAD;      // (>= 7 columns of wrap-around; see MIP_IPE360_ROW_PAD above)
    __shared__ GaussFull sg[kFragSamples];
    __shared__ float sy[kFragSamples][kRow], sv[kFragSamples][kRow];
    const int tid = threadIdx.x;
    const int64_t s0 = (int64_t)blockIdx.x * kFragSamples, M = B * (int64_t)N;
    if ((tid >> 6) == (int)(blockIdx.x & 3)) {
        const int m = tid & 63;
        const int64_t s = s0 + m, sc = s < M ? s : M - 1;
        const int64_t b = sc / N;
        const int i = (int)(sc - b * N);
        const float d[3] = {dirs[b * 3], dirs[b * 3 + 1], dirs[b * 3 + 2]};
        const float o[3] = {origins[b * 3], origins[b * 3 + 1], origins[b * 3 + 2]};
        sg[m] = conical_frustum_to_gaussian_full(t[b * (N + 1) + i], t[b * (N + 1) + i + 1], d, o, radii[b], contracted != 0);
    }
    __syncthreads();
    for (int idx = tid; idx < kFragSamples * kRow; idx += 256) {
        const int m = idx / kRow, jj = idx - m * kRow;
        float y, var;
        project_360(sg[m], jj < kBasis360N ? jj : jj - kBasis360N, y, var);
        sy[m][jj] = y;
        sv[m][jj] = var;
    }
    __syncthreads();
    const int nq = kBasis360N * L / 8;          // vectors per half (42 for 16 degrees); fragment layout: k-steps per sample = nq
    for (int w = tid; w < kFragSamples * nq; w += 256) {
        // FRAG: consecutive threads = consecutive samples of one vector index; rows: consecutive vector indices of one sample
        // (FRAG: q is the same for the 64 threads of a wave -- readfirstlane tells the compiler, so the (degree, direction) stepping, the
        // frequency scales and the LDS column offsets below become scalar work)
        const int m = FRAG ? (w & 63) : w / nq, q = FRAG ? __builtin_amdgcn_readfirstlane(w >> 6) : w - (w / nq) * nq;
        vec8 fs, fc;
        const int l0 = (q * 8) / kBasis360N, j0 = q * 8 - l0 * kBasis360N;       // (degree, direction) of the vector's first feature
        const float* py = &sy[m][j0];
        const float* pv = &sv[m][j0];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            OutT a, c;
            ipe360_pair<OutT>(py[i], pv[i], l0 + (j0 + i >= kBasis360N ? 1 : 0), min_deg, a, c);      // past direction 20: the next degree
            fs[i] = a;
            fc[i] = c;
        }
        if (FRAG) {
            // feature f = 8 q + i of sample m: wave tile m / 32, k-step q / 2, lane half q % 2, lane m % 32; the "cos" half starts nq vectors later
            vec8* o = reinterpret_cast<vec8*>(enc) + (s0 >> 5) * (int64_t)(nq * 64) + (m >> 5) * (nq * 64) + (m & 31);
            o[(q >> 1) * 64 + (q & 1) * 32] = fs;
            o[((q + nq) >> 1) * 64 + ((q + nq) & 1) * 32] = fc;
        } else if (s0 + m < M) {
            vec8* row = reinterpret_cast<vec8*>(enc + (s0 + m) * (int64_t)(16 * nq));
            row[q] = fs;
            row[q + nq] = fc;
        }
    }
}

template <typename OutT>
__global__ void __launch_bounds__(256)
k_cast_ipe_360(int64_t B, int N, int min_deg, int L, int contracted, const float* __restrict__ t,
               const float* __restrict__ origins, const float* __restrict__ dirs, const float* __restrict__ radii,
               OutT* __restrict__ enc, float* __restrict__ means_out, float* __restrict__ covs_out, int frag) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t s = gid / kBasis360N;
    const int j = (int)(gid - s * kBasis360N);
    if (s >= B * (int64_t)N) return;
    const int64_t b = s / N;
    const int i = (int)(s - b * N);
    const float d[3] = {dirs[b * 3], dirs[b * 3 + 1], dirs[b * 3 + 2]};
    const float o[3] = {origins[b * 3], origins[b * 3 + 1], origins[b * 3 + 2]};
    const GaussFull g = conical_frustum_to_gaussian_full(t[b * (N + 1) + i], t[b * (N + 1) + i + 1], d, o, radii[b], contracted != 0);
    if (j == 0 && means_out) {
#pragma unroll
        for (int a = 0; a < 3; ++a) means_out[s * 3 + a] = g.mean[a];
        const float full[9] = {g.cov[0], g.cov[1], g.cov[2], g.cov[1], g.cov[3], g.cov[4], g.cov[2], g.cov[4], g.cov[5]};
#pragma unroll
        for (int a = 0; a < 9; ++a) covs_out[s * 9 + a] = full[a];
    }
    if (!enc) return;
    float y, var;
    project_360(g, j, y, var);
    const int F = 2 * kBasis360N * L;
    for (int l = 0; l < L; ++l) {
        OutT fs, fc;
        ipe360_pair<OutT>(y, var, l, min_deg, fs, fc);
        enc[enc360_index(s, l * kBasis360N + j, F, frag)] = fs;
        enc[enc360_index(s, (L + l) * kBasis360N + j, F, frag)] = fc;
    }
}

// The same on GIVEN Gaussians (the reference's free functions take (means, covs) tensors): optional contraction of mean and
// covariance (`parameterization`, mip.py:431-447 -- generic J Sigma J^T, see raymath360.hpp for its conditioning), optional
// off-axis encoding (`integrated_pos_enc_360`, mip.py:292-319).  means [M,3], covs [M,3,3] (symmetric; the upper triangle is read).
template <typename OutT>
__global__ void __launch_bounds__(256)
k_gauss_360(int64_t M, int min_deg, int L, int contracted, const float* __restrict__ means, const float* __restrict__ covs,
            OutT* __restrict__ enc, float* __restrict__ means_out, float* __restrict__ covs_out) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t s = gid / kBasis360N;
    const int j = (int)(gid - s * kBasis360N);
    if (s >= M) return;
    GaussFull g;
#pragma unroll
    for (int a = 0; a < 3; ++a) g.mean[a] = means[s * 3 + a];
    const float* c = covs + s * 9;
    g.cov[0] = c[0]; g.cov[1] = c[1]; g.cov[2] = c[2]; g.cov[3] = c[4]; g.cov[4] = c[5]; g.cov[5] = c[8];
    if (contracted) contract_gaussian(g);
    if (j == 0 && means_out) {
#pragma unroll
        for (int a = 0; a < 3; ++a) means_out[s * 3 + a] = g.mean[a];
        if (covs_out) {
            const float full[9] = {g.cov[0], g.cov[1], g.cov[2], g.cov[1], g.cov[3], g.cov[4], g.cov[2], g.cov[4], g.cov[5]};
#pragma unroll
            for (int a = 0; a < 9; ++a) covs_out[s * 9 + a] = full[a];
        }
    }
    if (!enc) return;
    float y, var;
    project_360(g, j, y, var);
    OutT* row = enc + s * (int64_t)(2 * kBasis360N * L);
    for (int l = 0; l < L; ++l)
        ipe360_pair<OutT>(y, var, l, min_deg, row[l * kBasis360N + j], row[(L + l) * kBasis360N + j]);
}

// t = 1 / t_inv for the resampled inverse-depth fence posts of the fine level
__global__ void __launch_bounds__(256) k_reciprocal(int64_t n, const float* __restrict__ x, float* __restrict__ y) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = 1.0f / x[i];
}
hipError_t launch_reciprocal(int64_t n, const float* x, float* y, hipStream_t st) {
    hipLaunchKernelGGL(k_reciprocal, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, x, y);
    return hipGetLastError();
}

hipError_t launch_gauss_360(int64_t M, int min_deg, int max_deg, int contracted, const float* means, const float* covs, void* enc,
                            bool bf16, float* means_out, float* covs_out, hipStream_t st) {
    const int64_t n = M * kBasis360N;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    const int L = max_deg - min_deg;
    if (bf16)
        hipLaunchKernelGGL((k_gauss_360<__bf16>), grid, block, 0, st, M, min_deg, L, contracted, means, covs, (__bf16*)enc, means_out, covs_out);
    else
        hipLaunchKernelGGL((k_gauss_360<float>), grid, block, 0, st, M, min_deg, L, contracted, means, covs, (float*)enc, means_out, covs_out);
    return hipGetLastError();
}

hipError_t launch_sample_along_rays_360(int64_t B, int N, const float* nearp, const float* farp, const float* t_rand,
                                        float* t_inv, float* t, hipStream_t st) {
    const int64_t n = B * (int64_t)(N + 1);
    hipLaunchKernelGGL(k_sample_along_rays_360, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, B, N, nearp, farp, t_rand,
                       t_inv, t);
    return hipGetLastError();
}

// frag: bf16 only, (2 * 21 * L) % 16 == 0 -- the fragment layout of enc360_index; the kernel stores whole 256-sample tiles of the MLP kernels (64-sample workgroups, samples past the end repeat the last one), so the buffer must cover ceil(M / 256) * 256 rows
hipError_t launch_cast_ipe_360(int64_t B, int N, int min_deg, int max_deg, int contracted, const float* t, const float* origins,
                               const float* dirs, const float* radii, void* enc, bool bf16, float* means, float* covs,
                               hipStream_t st, bool frag) {
    const int64_t n = B * (int64_t)N * kBasis360N;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    const int L = max_deg - min_deg;
    if (frag && (!bf16 || (2 * kBasis360N * L) % 16 != 0)) return hipErrorInvalidValue;
    if (enc && !means && (kBasis360N * L) % 8 == 0) {         // only the encoding: the tiled kernel (one Gaussian per sample, vector stores)
        // fragments: whole 256-sample workgroup tiles of the MLP kernels (samples past the end repeat the last one: finite values that the
        // weight-gradient kernel of the training step multiplies by zero deltas)
        const int64_t wgs = frag ? ((B * (int64_t)N + 255) / 256) * (256 / kFragSamples) : (B * (int64_t)N + kFragSamples - 1) / kFragSamples;
        if (wgs > 0x7fffffff) return hipErrorInvalidValue;
        const dim3 g((unsigned)wgs);
        if (frag) hipLaunchKernelGGL((k_cast_ipe_360_tile<__bf16, true>), g, block, 0, st, B, N, min_deg, L, contracted, t, origins, dirs, radii, (__bf16*)enc);
        else if (bf16) hipLaunchKernelGGL((k_cast_ipe_360_tile<__bf16, false>), g, block, 0, st, B, N, min_deg, L, contracted, t, origins, dirs, radii, (__bf16*)enc);
        else hipLaunchKernelGGL((k_cast_ipe_360_tile<float, false>), g, block, 0, st, B, N, min_deg, L, contracted, t, origins, dirs, radii, (float*)enc);
        return hipGetLastError();
    }
    if (bf16)
        hipLaunchKernelGGL((k_cast_ipe_360<__bf16>), grid, block, 0, st, B, N, min_deg, L, contracted, t, origins, dirs, radii,
                           (__bf16*)enc, means, covs, frag ? 1 : 0);
    else
        hipLaunchKernelGGL((k_cast_ipe_360<float>), grid, block, 0, st, B, N, min_deg, L, contracted, t, origins, dirs, radii,
                           (float*)enc, means, covs, 0);
    return hipGetLastError();
}

}  // namespace mip

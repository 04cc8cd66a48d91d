// Per-sample ray math of the Mip-NeRF hot path: conical frustum -> Gaussian, integrated
// positional encoding, view-direction encoding, activations.
//
// Every function is MIP_HD (host + device) so that the exact same source is (a) inlined into
// the gfx950 kernels and (b) compiled with g++ by tests/hostmath to be checked against the
// oracle on a machine without a GPU.  The operation ORDER follows the reference
// expression-by-expression (cited per function) and this header must be compiled with
// -ffp-contract=off: a fused multiply-add in `mean = d*t_mean + o` changes the last bit of
// the mean, and 2^l * mean for l up to 15 turns that bit into up to 1e-3 rad of phase
// ("sin at large arguments").
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define MIP_HD __host__ __device__ __forceinline__
#else
#define MIP_HD inline
#endif

namespace mip {

constexpr int kMaxDeg = 16;           // nerf.max_deg_point of the shipped config
constexpr float kHalfPiF = 1.57079637050628662109375f;  // float32(0.5 * float32(pi)), mip.py:350

struct Gauss3 {
    float mean[3];
    float cov[3];
};

// models/mip.py:50-78 (stable branch) + 22-36 (diagonal lift) + 101-102 (means += origins).
// d = rays.directions (NOT normalised), o = rays.origins, radius = rays.radii.
MIP_HD Gauss3 conical_frustum_to_gaussian(float t0, float t1, const float d[3], const float o[3],
                                          float radius) {
    const float mu = (t0 + t1) / 2.0f;
    const float hw = (t1 - t0) / 2.0f;
    const float mu2 = mu * mu;
    const float hw2 = hw * hw;
    const float hw4 = hw2 * hw2;            // reference: hw ** 4 (pow, <= 1 ulp from this)
    const float den = 3.0f * mu2 + hw2;
    const float t_mean = mu + (2.0f * mu * hw2) / den;
    const float t_var = hw2 / 3.0f - (float)(4.0 / 15.0) * ((hw4 * (12.0f * mu2 - hw2)) / (den * den));
    const float r_var = (radius * radius) *
                        (mu2 / 4.0f + (float)(5.0 / 12.0) * hw2 - (float)(4.0 / 15.0) * hw4 / den);
    const float dd0 = d[0] * d[0], dd1 = d[1] * d[1], dd2 = d[2] * d[2];
    const float dn = (dd0 + dd1 + dd2) + 1e-10f;   // torch.sum(d**2) + 1e-10
    Gauss3 g;
    const float dd[3] = {dd0, dd1, dd2};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        g.mean[a] = d[a] * t_mean + o[a];
        const float null_outer = 1.0f - dd[a] / dn;
        g.cov[a] = t_var * dd[a] + r_var * null_outer;
    }
    return g;
}

// Accurate sin for |x| up to ~1e6: the reference calls torch.sin on fp32 (<= 1 ulp).
MIP_HD float sin_accurate(float x) { return sinf(x); }
MIP_HD float exp_accurate(float x) { return expf(x); }

#if defined(__HIPCC__)
// Fast variants for the bf16 pipeline (features are rounded to 8 mantissa bits right after):
// exact argument reduction in fp64 -- x/(2 pi) is formed with a 53-bit product, so even at |x| = 3e5 rad
// the reduced phase is good to 1e-11 turns -- followed by the hardware v_sin_f32 (input in turns) and
// v_exp_f32.  ~12 issue slots instead of ~150 for the accurate libm pair.
#ifndef MIP_SIN_FAST_TWOFLOAT
#define MIP_SIN_FAST_TWOFLOAT 0
#endif
__device__ __forceinline__ float sin_fast(float x) {
#if MIP_SIN_FAST_TWOFLOAT
    // the same reduction in two-float fp32 arithmetic (six full-rate instructions instead of five half-/quarter-rate fp64 ones): 1 / 2 pi = hi + lo,
    // p = fl(x hi), e = x hi - p exactly (fma), t = fl(x lo + e), frac(x / 2 pi) = (p - rint(p)) + t with an exact subtraction; 2.6e-8 turns
    // against the fp64 form on 2e5 arguments up to 2^17 rad (build knob MLP_SIN_TWOFLOAT, see build.py)
    constexpr float kHi = 0.15915494f, kLo = 6.4206382e-09f;
    const float p = x * kHi;
    const float e = __builtin_fmaf(x, kHi, -p);
    const float t = __builtin_fmaf(x, kLo, e);
    return __builtin_amdgcn_sinf((p - __builtin_rintf(p)) + t);
#else
    double r = (double)x * 0.15915494309189535;   // 1 / (2 pi)
    r -= rint(r);                                 // [-0.5, 0.5] turns
    return __builtin_amdgcn_sinf((float)r);
#endif
}
__device__ __forceinline__ float exp_fast(float x) { return __expf(x); }
#endif

// models/mip.py:322-350 + 283-289: feature (half, l, axis) of the integrated positional
// encoding, index = half*3L + l*3 + axis; "cos" is sin(fl32(y + fl32(pi/2))) as the reference.
MIP_HD float ipe_feature(const Gauss3& g, int half, int l, int axis, int min_deg) {
    const float scale = (float)(1u << (l + min_deg));            // 2^l exact
    const float y = g.mean[axis] * scale;                        // exact scaling
    const float yv = g.cov[axis] * (scale * scale);
    const float x = half ? (y + kHalfPiF) : y;
    return exp_accurate(-0.5f * yv) * sin_accurate(x);
}

// models/mip.py:353-363: pos_enc(viewdirs, 0, deg, append_identity=True)
// -> [x(3) | sin(2^l x) (3*deg) | sin(2^l x + pi/2) (3*deg)]
MIP_HD float view_feature(const float v[3], int idx, int deg) {
    if (idx < 3) return v[idx];
    int k = idx - 3;
    const int half = k >= 3 * deg;
    if (half) k -= 3 * deg;
    const int l = k / 3, a = k % 3;
    const float xb = v[a] * (float)(1u << l);
    return sin_accurate(half ? (xb + kHalfPiF) : xb);
}

// models/mip_nerf.py:236-238
MIP_HD float rgb_activation(float raw, float rgb_padding) {
    const float s = 1.0f / (1.0f + expf(-raw));
    return s * (1.0f + 2.0f * rgb_padding) - rgb_padding;
}
// torch.nn.Softplus(beta=1, threshold=20) applied to raw + density_bias
MIP_HD float density_activation(float raw, float density_bias) {
    const float x = raw + density_bias;
    return x > 20.0f ? x : log1pf(expf(x));
}

// torch.linspace(start, end, steps)[i] in fp32 (ATen: start + step*i below the midpoint,
// end - step*(steps-1-i) above).  ATen's kernels contract the multiply-add (FMA on the CPU
// build, -fmad on the CUDA build) -- verified against torch.linspace bit-for-bit in
// tests/test_hostmath_cpu.py -- so this is the one place where a fused multiply-add IS the
// reference rounding.
MIP_HD float torch_linspace_at(float start, float end, int steps, int i) {
    if (steps == 1) return start;
    const float step = (end - start) / (float)(steps - 1);
    return (i < steps / 2) ? fmaf(step, (float)i, start) : fmaf(-step, (float)(steps - 1 - i), end);
}

MIP_HD float nan_to_num(float x) {
    if (x != x) return 0.0f;
    if (x > 3.4028234663852886e38f) return 3.4028234663852886e38f;
    if (x < -3.4028234663852886e38f) return -3.4028234663852886e38f;
    return x;
}

}  // namespace mip

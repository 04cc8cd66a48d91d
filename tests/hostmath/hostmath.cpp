// Test-only: compiles mipnerf_pl_amd/csrc/raymath.hpp (the exact source inlined into the gfx950
// kernels) with g++ so the per-sample formulas can be checked against the oracle without a GPU.
// Build: g++ -O2 -ffp-contract=off -shared -fPIC hostmath.cpp -o _hostmath.so
#include "../../mipnerf_pl_amd/csrc/raymath.hpp"
#include "../../mipnerf_pl_amd/csrc/raymath360.hpp"

extern "C" {
void hm_cast(int n, const float* t0, const float* t1, const float* d, const float* o, const float* radius,
             float* means, float* covs) {
    for (int i = 0; i < n; ++i) {
        mip::Gauss3 g = mip::conical_frustum_to_gaussian(t0[i], t1[i], d + 3 * i, o + 3 * i, radius[i]);
        for (int a = 0; a < 3; ++a) { means[3 * i + a] = g.mean[a]; covs[3 * i + a] = g.cov[a]; }
    }
}
void hm_ipe(int n, const float* means, const float* covs, int min_deg, int L, float* enc) {
    for (int i = 0; i < n; ++i) {
        mip::Gauss3 g;
        for (int a = 0; a < 3; ++a) { g.mean[a] = means[3 * i + a]; g.cov[a] = covs[3 * i + a]; }
        for (int h = 0; h < 2; ++h)
            for (int l = 0; l < L; ++l)
                for (int a = 0; a < 3; ++a) enc[i * 6 * L + h * 3 * L + l * 3 + a] = mip::ipe_feature(g, h, l, a, min_deg);
    }
}
void hm_view(int n, const float* v, int deg, float* out) {
    const int w = 3 + 6 * deg;
    for (int i = 0; i < n; ++i)
        for (int c = 0; c < w; ++c) out[i * w + c] = mip::view_feature(v + 3 * i, c, deg);
}
void hm_act(int n, const float* raw_rgb, const float* raw_density, float rgb_padding, float density_bias,
            float* rgb, float* density) {
    for (int i = 0; i < n; ++i) {
        rgb[i] = mip::rgb_activation(raw_rgb[i], rgb_padding);
        density[i] = mip::density_activation(raw_density[i], density_bias);
    }
}
void hm_linspace(float start, float end, int steps, float* out) {
    for (int i = 0; i < steps; ++i) out[i] = mip::torch_linspace_at(start, end, steps, i);
}
// mip-NeRF 360 path: frustum -> full-covariance Gaussian [-> contraction] -> off-axis IPE (raymath360.hpp)
void hm_cast_ipe_360(int n, const float* t0, const float* t1, const float* d, const float* o, const float* radius, int contracted,
                     int min_deg, int L, float* means, float* covs, float* enc) {
    for (int i = 0; i < n; ++i) {
        mip::GaussFull g = mip::conical_frustum_to_gaussian_full(t0[i], t1[i], d + 3 * i, o + 3 * i, radius[i], contracted == 1);
        if (contracted == 2) mip::contract_gaussian(g);      // generic triple product (any covariance)
        for (int a = 0; a < 3; ++a) means[3 * i + a] = g.mean[a];
        const float full[9] = {g.cov[0], g.cov[1], g.cov[2], g.cov[1], g.cov[3], g.cov[4], g.cov[2], g.cov[4], g.cov[5]};
        for (int a = 0; a < 9; ++a) covs[9 * i + a] = full[a];
        for (int j = 0; j < mip::kBasis360N; ++j) {
            float y, var;
            mip::project_360(g, j, y, var);
            for (int l = 0; l < L; ++l) {
                enc[(size_t)i * 2 * 21 * L + l * 21 + j] = mip::ipe360_feature(y, var, 0, l, min_deg);
                enc[(size_t)i * 2 * 21 * L + (L + l) * 21 + j] = mip::ipe360_feature(y, var, 1, l, min_deg);
            }
        }
    }
}
}

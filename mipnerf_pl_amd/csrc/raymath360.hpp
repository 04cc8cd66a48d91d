// Per-sample math of the unbounded-scene (mip-NeRF 360) ray path: full-covariance conical-frustum Gaussian,
// scene contraction applied to a Gaussian, off-axis integrated positional encoding.  Host + device (MIP_HD) like
// raymath.hpp, so tests/hostmath can check it against oracle/mipnerf360_oracle.py without a GPU.
//
// The reference's functions for this path are dead and wrong (models/mip.py:38-47 uses t_var for the perpendicular
// term, :445 replaces the covariance by the Jacobian, :316-319 has no frequency scales, and nothing calls them), so this
// follows the paper they aim at -- Barron et al., "Mip-NeRF 360", CVPR 2022 -- equation numbers cited per function.
#pragma once

#include "raymath.hpp"

namespace mip {

struct GaussFull {
    float mean[3];
    float cov[6];     // symmetric 3x3: xx, xy, xz, yy, yz, zz
};

constexpr int kBasis360N = 21;
// the 21 non-antipodal vertices of a twice-tessellated icosahedron (same table as models/mip.py:293-313)
#if defined(__HIPCC__)
__device__ __constant__
#endif
static const float kBasis360[kBasis360N][3] = {
    {0.8506508f, 0.f, 0.5257311f}, {0.809017f, 0.5f, 0.309017f}, {0.5257311f, 0.8506508f, 0.f}, {1.f, 0.f, 0.f},
    {0.809017f, 0.5f, -0.309017f}, {0.8506508f, 0.f, -0.5257311f}, {0.309017f, 0.809017f, -0.5f},
    {0.f, 0.5257311f, -0.8506508f}, {0.5f, 0.309017f, -0.809017f}, {0.f, 1.f, 0.f}, {-0.5257311f, 0.8506508f, 0.f},
    {-0.309017f, 0.809017f, -0.5f}, {0.f, 0.5257311f, 0.8506508f}, {-0.309017f, 0.809017f, 0.5f},
    {0.309017f, 0.809017f, 0.5f}, {0.5f, 0.309017f, 0.809017f}, {0.5f, -0.309017f, 0.809017f}, {0.f, 0.f, 1.f},
    {-0.5f, 0.309017f, 0.809017f}, {-0.809017f, 0.5f, 0.309017f}, {-0.809017f, 0.5f, -0.309017f}};

// Jacobian of contract() at x, |x| > 1 (symmetric): J = a (I - u u^T) + b u u^T, u = x/|x|, a = (2|x| - 1)/|x|^2, b = 1/|x|^2
struct ContractJ {
    float u[3], a, b, mean_scale;
};
MIP_HD ContractJ contract_jacobian(const float m[3], float n2) {
    ContractJ c;
    const float n = sqrtf(n2);
    c.u[0] = m[0] / n; c.u[1] = m[1] / n; c.u[2] = m[2] / n;
    c.a = (2.0f * n - 1.0f) / n2;
    c.b = 1.0f / n2;
    c.mean_scale = (2.0f - 1.0f / n) / n;
    return c;
}

// mip-NeRF eq. (7) (stable form, as raymath.hpp) + eq. (8) with the FULL covariance
//   mean = d t_mean + o,   cov = t_var d d^T + r_var (I - d d^T / |d|^2),
// and, when `contracted`, the paper's eq. (9)/(10): mean' = contract(mean), cov' = J cov J^T.  The structure of cov is
// used instead of a generic 3x3 triple product: with v = J d,
//   cov' = t_var v v^T + r_var (J J^T - v v^T / |d|^2),   J J^T = a^2 (I - u u^T) + b^2 u u^T,
// every term a well-scaled outer product (a generic J C J^T in fp32 loses ~1e-4 of the largest entry at |x| ~ 1e3,
// where J's radial and tangential scales differ by |x| and C's by (t_var / r_var)).
MIP_HD GaussFull conical_frustum_to_gaussian_full(float t0, float t1, const float d[3], const float o[3], float radius,
                                                  bool contracted = false) {
    const float mu = (t0 + t1) / 2.0f;
    const float hw = (t1 - t0) / 2.0f;
    const float mu2 = mu * mu, hw2 = hw * hw, hw4 = hw2 * hw2;
    const float den = 3.0f * mu2 + hw2;
    const float t_mean = mu + (2.0f * mu * hw2) / den;
    const float t_var = hw2 / 3.0f - (float)(4.0 / 15.0) * ((hw4 * (12.0f * mu2 - hw2)) / (den * den));
    const float r_var = (radius * radius) * (mu2 / 4.0f + (float)(5.0 / 12.0) * hw2 - (float)(4.0 / 15.0) * hw4 / den);
    const float dn = (d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) + 1e-10f;
    GaussFull g;
#pragma unroll
    for (int a = 0; a < 3; ++a) g.mean[a] = d[a] * t_mean + o[a];
    const float n2 = (g.mean[0] * g.mean[0] + g.mean[1] * g.mean[1]) + g.mean[2] * g.mean[2];
    if (contracted && n2 > 1.0f) {
        const ContractJ c = contract_jacobian(g.mean, n2);
        const float ud = (c.u[0] * d[0] + c.u[1] * d[1]) + c.u[2] * d[2];
        float v[3];                                   // v = J d = a d + (b - a) (u . d) u
#pragma unroll
        for (int a = 0; a < 3; ++a) v[a] = c.a * d[a] + ((c.b - c.a) * ud) * c.u[a];
        const float a2 = c.a * c.a, b2a2 = c.b * c.b - a2;
        int k = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = i; j < 3; ++j) {
                const float vv = v[i] * v[j];
                const float jj = (i == j ? a2 : 0.0f) + b2a2 * (c.u[i] * c.u[j]);       // (J J^T)_ij
                g.cov[k++] = t_var * vv + r_var * (jj - vv / dn);
            }
#pragma unroll
        for (int a = 0; a < 3; ++a) g.mean[a] *= c.mean_scale;
        return g;
    }
    int k = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = a; b < 3; ++b) {
            const float dd = d[a] * d[b];
            const float null_outer = (a == b ? 1.0f : 0.0f) - dd / dn;
            g.cov[k++] = t_var * dd + r_var * null_outer;
        }
    return g;
}

// contract() of an ARBITRARY Gaussian (paper eq. (9)/(10)): mean' = contract(mean), cov' = J cov J^T (generic triple
// product; conditioned like |x| -- prefer the fused form above when the Gaussian comes from a conical frustum).
MIP_HD void contract_gaussian(GaussFull& g) {
    const float n2 = (g.mean[0] * g.mean[0] + g.mean[1] * g.mean[1]) + g.mean[2] * g.mean[2];
    if (!(n2 > 1.0f)) return;
    const ContractJ c = contract_jacobian(g.mean, n2);
    float J[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float uu = c.u[i] * c.u[j];
            J[i][j] = c.a * ((i == j ? 1.0f : 0.0f) - uu) + c.b * uu;
        }
    const float C[3][3] = {{g.cov[0], g.cov[1], g.cov[2]}, {g.cov[1], g.cov[3], g.cov[4]}, {g.cov[2], g.cov[4], g.cov[5]}};
    float T[3][3];      // J C
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) T[i][j] = (J[i][0] * C[0][j] + J[i][1] * C[1][j]) + J[i][2] * C[2][j];
    int k = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = i; j < 3; ++j) g.cov[k++] = (T[i][0] * J[j][0] + T[i][1] * J[j][1]) + T[i][2] * J[j][2];   // (J C J^T)_ij
#pragma unroll
    for (int a = 0; a < 3; ++a) g.mean[a] *= c.mean_scale;
}

// projection on basis direction j: y = p . mean, var = p^T cov p
MIP_HD void project_360(const GaussFull& g, int j, float& y, float& var) {
    const float px = kBasis360[j][0], py = kBasis360[j][1], pz = kBasis360[j][2];
    y = (g.mean[0] * px + g.mean[1] * py) + g.mean[2] * pz;
    const float cx = (g.cov[0] * px + g.cov[1] * py) + g.cov[2] * pz;
    const float cy = (g.cov[1] * px + g.cov[3] * py) + g.cov[4] * pz;
    const float cz = (g.cov[2] * px + g.cov[4] * py) + g.cov[5] * pz;
    var = (cx * px + cy * py) + cz * pz;
}

// off-axis IPE feature (half, l, basis j): exp(-0.5 * 4^l var) * sin(2^l y [+ pi/2]); index = half*21L + l*21 + j
MIP_HD float ipe360_feature(float y, float var, int half, int l, int min_deg) {
    const float scale = (float)(1u << (l + min_deg));
    const float ys = y * scale;
    const float vs = var * (scale * scale);
    return exp_accurate(-0.5f * vs) * sin_accurate(half ? (ys + kHalfPiF) : ys);
}

}  // namespace mip

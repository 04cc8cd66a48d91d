"""First-principles pins of `oracle/mipnerf360_oracle.py`.

The reference's unbounded-scene functions (models/mip.py:38-47 full-covariance lift, :292-319 off-axis encoding, :424-447
contract / parameterization) are dead and wrong upstream, so nothing reference-held can pin the oracle's covariances, its
linearised contraction or its off-axis integrated positional encoding -- and the kernels and the oracle were written by the same
hand from the same paper.  These tests pin the oracle INDEPENDENTLY of that hand, against the definitions the formulas come from:

  1. the 21 basis directions            == the vertices of a twice-tessellated icosahedron, built here from the golden ratio
  2. the frustum -> Gaussian lift       == mean / full covariance of the UNIFORM distribution over the conical frustum (closed-form
                                           float64 integrals of t^k and a Monte-Carlo over the volume for the 3 x 3 structure)
  3. the contraction's Jacobian         == central finite differences of contract() in float64
  4. contract_gaussian (linearisation)  == sample mean / covariance of contract(x), x ~ N(mu, Sigma), INSIDE the validity radius
                                           found below (kappa = sqrt(lambda_max(Sigma)) / |mu| <= 0.03: mean within 0.6 kappa^2, covariance
                                           within Monte-Carlo noise); outside it the error must grow like kappa^2 (that IS the linearisation)
  5. integrated_pos_enc_360             == E[sin / cos(2^l p^T x)], x ~ N(mu, Sigma): Gauss-Hermite quadrature of the 1-D marginal (tight)
                                           and a 3-D Monte-Carlo (exact for Gaussians up to sampling noise)
  6. frustum -> contract -> encoding    == Monte-Carlo E[sin / cos(2^l P^T contract(x))] over x drawn from the frustum's Gaussian

Sample counts: PIN360_SAMPLES (default 2e5 per Gaussian so the CPU suite stays fast); `python tests/test_oracle360_first_principles.py
--full` runs ~50 Gaussians x 2e6 samples and appends the residuals to profiles/r04_parity.jsonl.
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import mipnerf360_oracle as o360  # noqa: E402

S_DEFAULT = int(float(os.environ.get("PIN360_SAMPLES", "2e5")))
RESIDUALS = {}


def _rec(tag, **vals):
    RESIDUALS[tag] = {k: float(v) for k, v in vals.items()}


def contract64(x):
    """Paper eq. (10) in float64, written from the definition (NOT imported from the oracle)."""
    n = np.linalg.norm(x, axis=-1, keepdims=True)
    return np.where(n > 1.0, (2.0 - 1.0 / np.maximum(n, 1e-300)) * x / np.maximum(n, 1e-300), x)


def _random_cov(rng, scale):
    a = rng.standard_normal((3, 3))
    c = a @ a.T + 0.05 * np.eye(3)
    return c * (scale ** 2 / np.linalg.eigvalsh(c).max())


# ---- 1. basis ---------------------------------------------------------------------------------------------------------
def _icosahedron():
    phi = (1.0 + np.sqrt(5.0)) / 2.0
    v = []
    for a in (-1.0, 1.0):
        for b in (-phi, phi):
            v += [(0.0, a, b), (a, b, 0.0), (b, 0.0, a)]
    v = np.asarray(v)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    d = np.linalg.norm(v[:, None] - v[None], axis=-1)
    edge = d[d > 1e-9].min()
    faces = [(i, j, k) for i in range(12) for j in range(i + 1, 12) for k in range(j + 1, 12)
             if abs(d[i, j] - edge) < 1e-9 and abs(d[j, k] - edge) < 1e-9 and abs(d[i, k] - edge) < 1e-9]
    assert len(faces) == 20
    return v, faces


def test_basis_is_the_twice_tessellated_icosahedron():
    """mip.py:293-313 lists 21 directions; the paper says: vertices of a twice-tessellated icosahedron, antipodal copies removed."""
    v, faces = _icosahedron()
    pts = []
    for (i, j, k) in faces:                       # tessellation factor 2: vertices + edge midpoints of every face
        for (a, b, c) in ((2, 0, 0), (0, 2, 0), (0, 0, 2), (1, 1, 0), (1, 0, 1), (0, 1, 1)):
            p = (a * v[i] + b * v[j] + c * v[k]) / 2.0
            pts.append(p / np.linalg.norm(p))
    pts = np.asarray(pts)
    uniq = []
    for p in pts:                                  # drop duplicates and antipodes
        if not any(min(np.linalg.norm(p - q), np.linalg.norm(p + q)) < 1e-6 for q in uniq):
            uniq.append(p)
    uniq = np.asarray(uniq)
    assert uniq.shape == (21, 3)
    basis = o360.BASIS_360.T.astype(np.float64)
    assert basis.shape == (21, 3)
    assert np.max(np.abs(np.linalg.norm(basis, axis=1) - 1.0)) < 2e-7
    # every listed direction is one of the 21 (up to sign) and they are all different
    match = [int(np.argmin([min(np.linalg.norm(b - q), np.linalg.norm(b + q)) for q in uniq])) for b in basis]
    err = max(min(np.linalg.norm(b - uniq[m]), np.linalg.norm(b + uniq[m])) for b, m in zip(basis, match))
    assert sorted(match) == list(range(21)) and err < 2e-7, (match, err)
    _rec("pin360 basis", max_dir_err=err)


# ---- 2. frustum -> Gaussian -------------------------------------------------------------------------------------------
def _frustum_cases(rng, n):
    t0 = np.concatenate([rng.uniform(0.05, 6.0, n // 2), 10 ** rng.uniform(-1, 2.3, n - n // 2)])
    rel = 10 ** rng.uniform(-3, 0.3, n)           # thin slabs (t1/t0 - 1 = 1e-3) to 3x ranges
    t1 = t0 * (1.0 + rel)
    rad = 10 ** rng.uniform(-4, -1.5, n)
    d = rng.standard_normal((n, 3))
    d *= rng.uniform(0.8, 1.3, (n, 1)) / np.linalg.norm(d, axis=1, keepdims=True)       # un-normalised like the datasets'
    return t0, t1, rad, d


def test_frustum_moments_equal_the_integrals_over_the_cone():
    """mip-NeRF eq. (7): the moments of the uniform distribution over {t in [t0,t1], |x_perp| <= r t}: density of t ~ t^2,
    E[t^k] = 3 (t1^(k+3) - t0^(k+3)) / ((k+3)(t1^3 - t0^3)); a disc of radius R has per-axis variance R^2 / 4."""
    rng = np.random.default_rng(1)
    t0, t1, rad, d = _frustum_cases(rng, 400)
    tm, tv, rv = o360.conical_frustum_moments(t0, t1, rad)
    vol = t1 ** 3 - t0 ** 3
    e1 = 0.75 * (t1 ** 4 - t0 ** 4) / vol
    e2 = 0.6 * (t1 ** 5 - t0 ** 5) / vol
    # thin slabs: the variance is hw^2/3 ~ 1e-7 t^2 and the fp32 inputs t0, t1 themselves are rounded (6e-8 relative): compare in the
    # fp32 inputs the oracle actually saw
    t0f, t1f = t0.astype(np.float32).astype(np.float64), t1.astype(np.float32).astype(np.float64)
    vol = t1f ** 3 - t0f ** 3
    e1 = 0.75 * (t1f ** 4 - t0f ** 4) / vol
    e2 = 0.6 * (t1f ** 5 - t0f ** 5) / vol
    # E[t^2] - E[t]^2 cancels catastrophically for thin slabs even in float64 -> the stable float64 form for the variance
    mu, hw = (t0f + t1f) / 2, (t1f - t0f) / 2
    var_t = hw ** 2 / 3 - (4 / 15) * hw ** 4 * (12 * mu ** 2 - hw ** 2) / (3 * mu ** 2 + hw ** 2) ** 2
    wide = (t1f / t0f) > 1.2
    assert np.max(np.abs((e2 - e1 ** 2)[wide] / var_t[wide] - 1)) < 1e-9          # the stable form IS the integral where both are accurate
    radf = rad.astype(np.float32).astype(np.float64)
    err_m = np.max(np.abs(tm / e1 - 1))
    err_v = np.max(np.abs(tv / var_t - 1))
    err_r = np.max(np.abs(rv / (radf ** 2 * e2 / 4) - 1))
    _rec("pin360 frustum moments", t_mean_rel=err_m, t_var_rel=err_v, r_var_rel=err_r)
    assert err_m < 5e-7 and err_r < 1e-6, (err_m, err_r)
    assert err_v < 2e-5, err_v           # fp32 evaluation of hw^2/3 - (4/15)(...): a few ulps of each term


def test_full_covariance_lift_vs_monte_carlo_over_the_frustum_volume():
    """mip-NeRF eq. (8) (what mip.py:38-47 gets wrong: it uses t_var for the perpendicular term): points drawn UNIFORMLY in the
    frustum's volume, sample mean / covariance against lift_gaussian_full and cast_rays_360(contracted=False)."""
    rng = np.random.default_rng(2)
    S = max(S_DEFAULT, 200000)
    worst_m, worst_c = 0.0, 0.0
    for case in range(6):
        t0 = float(rng.uniform(0.3, 5.0))
        t1 = t0 * float(rng.uniform(1.05, 2.5))
        rad = float(10 ** rng.uniform(-2.5, -0.7))                   # fat cones so that the perpendicular term is visible in the MC
        d = rng.standard_normal(3)
        d *= rng.uniform(0.8, 1.3) / np.linalg.norm(d)
        origin = rng.uniform(-1, 1, 3)
        # inverse-CDF sampling of t (density ~ t^2) and of the disc radius (density ~ rho)
        u = rng.uniform(size=S)
        t = (t0 ** 3 + u * (t1 ** 3 - t0 ** 3)) ** (1.0 / 3.0)
        rho = np.sqrt(rng.uniform(size=S))
        ang = rng.uniform(0, 2 * np.pi, S)
        e1 = np.cross(d, [0.3, -0.5, 0.8])
        e1 /= np.linalg.norm(e1)
        e2 = np.cross(d / np.linalg.norm(d), e1)
        x = origin + t[:, None] * d + (rad * t * rho)[:, None] * (np.cos(ang)[:, None] * e1 + np.sin(ang)[:, None] * e2)
        m_mc, c_mc = x.mean(0), np.cov(x.T)
        tm, tv, rv = o360.conical_frustum_moments(np.array([[t0]]), np.array([[t1]]), np.array([[rad]]))
        mean, cov = o360.lift_gaussian_full(d[None].astype(np.float32), tm, tv, rv)
        mean = mean[0, 0] + origin
        m2, c2 = o360.cast_rays_360(np.array([[t0, t1]], np.float32), origin[None], d[None], np.array([[rad]]), False)
        assert np.max(np.abs(m2[0, 0] - mean)) < 1e-5 * max(1.0, np.abs(mean).max())
        assert np.linalg.norm(c2[0, 0] - cov[0, 0]) < 1e-4 * np.linalg.norm(cov[0, 0])
        sd = np.sqrt(np.trace(c_mc))
        em = np.linalg.norm(m_mc - mean) / sd                         # in units of the distribution's own spread
        ec = np.linalg.norm(c_mc - cov[0, 0]) / np.linalg.norm(c_mc)
        worst_m, worst_c = max(worst_m, em), max(worst_c, ec)
        # the perpendicular variance alone (the term upstream gets wrong): project out d
        P = np.eye(3) - np.outer(d, d) / (d @ d)
        perp_mc = np.trace(P @ c_mc @ P) / 2
        assert abs(perp_mc / float(rv[0, 0]) - 1) < 6.0 / np.sqrt(S) + 1e-5, (perp_mc, float(rv[0, 0]))
    _rec("pin360 lift vs frustum MC", mean_err_in_sigmas=worst_m, cov_rel_fro=worst_c, samples=S)
    assert worst_m < 5.0 / np.sqrt(S) and worst_c < 8.0 / np.sqrt(S), (worst_m, worst_c)


# ---- 3. Jacobian of the contraction -------------------------------------------------------------------------------------
def test_contraction_jacobian_by_finite_differences():
    rng = np.random.default_rng(3)
    worst = 0.0
    for r in (1.05, 1.5, 3.0, 10.0, 30.0, 300.0, 0.5):
        for _ in range(4):
            u = rng.standard_normal(3)
            mu = r * u / np.linalg.norm(u)
            h = 1e-5 * r
            J = np.stack([(contract64(mu + h * e) - contract64(mu - h * e)) / (2 * h) for e in np.eye(3)], axis=1)
            Sig = _random_cov(rng, 0.01 * r)
            want = J @ Sig @ J.T
            _, got = o360.contract_gaussian(mu.astype(np.float32)[None], Sig.astype(np.float32)[None])
            mu32 = mu.astype(np.float32).astype(np.float64)            # the oracle linearises at the fp32 mean it was given
            J32 = np.stack([(contract64(mu32 + h * e) - contract64(mu32 - h * e)) / (2 * h) for e in np.eye(3)], axis=1)
            want32 = J32 @ Sig.astype(np.float32).astype(np.float64) @ J32.T
            err = np.linalg.norm(got[0] - want32) / np.linalg.norm(want32)
            worst = max(worst, err)
            assert np.linalg.norm(want - want32) / np.linalg.norm(want) < 1e-5
    _rec("pin360 contraction Jacobian vs finite differences", cov_rel_fro=worst)
    assert worst < 2e-6, worst


# ---- 4. linearised contraction vs the true push-forward ------------------------------------------------------------------
KAPPAS_VALID = (0.003, 0.01, 0.03)
KAPPAS_BEYOND = (0.1, 0.3)


def _pushforward_errors(rng, mu_norm, kappa, S):
    u = rng.standard_normal(3)
    mu = mu_norm * u / np.linalg.norm(u)
    Sig = _random_cov(rng, kappa * mu_norm)
    x = rng.multivariate_normal(mu, Sig, size=S)
    z = contract64(x)
    m, C = o360.contract_gaussian(mu.astype(np.float32)[None], Sig.astype(np.float32)[None])
    em = np.linalg.norm(z.mean(0) - m[0]) / np.linalg.norm(m[0])
    ec = np.linalg.norm(np.cov(z.T) - C[0]) / np.linalg.norm(C[0])
    return em, ec


def test_contract_gaussian_vs_monte_carlo_pushforward(S=None, norms=(1.3, 2.0, 4.0, 12.0, 30.0)):
    """Paper eq. (9) linearises the contraction at the mean.  Validity radius found: kappa = sqrt(lambda_max(Sigma)) / |mu| <= 0.03
    keeps the mean within 0.6 kappa^2 (5e-4) of the true push-forward's and the covariance within sampling noise (1 % at kappa = 0.1);
    beyond it both errors grow like kappa^2 -- the signature of a first-order expansion, which is what the paper (and so the oracle
    and the kernels) define as the model."""
    S = S or S_DEFAULT
    rng = np.random.default_rng(4)
    noise = 1.0 / np.sqrt(S)
    rec = {}
    for r in norms:
        errs = {k: _pushforward_errors(rng, r, k, S) for k in KAPPAS_VALID + KAPPAS_BEYOND}
        for k in KAPPAS_VALID:
            em, ec = errs[k]
            # the Monte-Carlo mean itself is known to kappa / sqrt(S) relative to |mu| (~|z| here)
            assert em <= 0.8 * k * k + 6.0 * k * noise, (r, k, em)
            assert ec <= 4.0 * k * k + 8.0 * noise, (r, k, ec)
        (em1, ec1), (em3, ec3) = errs[0.1], errs[0.3]
        assert 0.2 * 0.01 < em1 < 0.8 * 0.01 and 0.2 * 0.09 < em3 < 0.9 * 0.09, (r, em1, em3)       # ~0.5 kappa^2
        assert ec3 > 2.5 * ec1 > 0, (r, ec1, ec3)        # (near the unit sphere part of a wide Gaussian falls inside, where J = I)
        for k, (em, ec) in errs.items():
            rec[f"mean_rel_r{r}_k{k}"] = em
            rec[f"cov_rel_r{r}_k{k}"] = ec
    # inside the unit ball the contraction is the identity: nothing to linearise
    em, ec = _pushforward_errors(rng, 0.2, 0.1, S)
    assert em < 6 * 0.1 * noise + 1e-6 and ec < 8 * noise, (em, ec)
    _rec("pin360 linearised contraction vs MC push-forward", samples=S, **rec)


# ---- 5. the off-axis IPE is exact for Gaussians -----------------------------------------------------------------------------
def _gauss_hermite_expectation(y_mean, y_var, scale, n=96):
    """E[sin(scale * y)], E[cos(scale * y)] for y ~ N(y_mean, y_var) by Gauss-Hermite quadrature (float64)."""
    xs, ws = np.polynomial.hermite_e.hermegauss(n)
    ws = ws / ws.sum()
    y = y_mean[..., None] + np.sqrt(y_var)[..., None] * xs
    return (np.sin(scale * y) * ws).sum(-1), (np.cos(scale * y) * ws).sum(-1)


def _ipe_gaussians(rng, n):
    """Gaussians spanning |mu| in [0.2, 30] uncontracted, covariances from 1e-4 to O(1) of |mu|."""
    out = []
    for i in range(n):
        r = 0.2 * (150.0 ** (i / max(n - 1, 1)))
        u = rng.standard_normal(3)
        out.append((r * u / np.linalg.norm(u), _random_cov(rng, r * 10 ** rng.uniform(-4, -0.3))))
    return out


def test_off_axis_ipe_equals_the_gaussian_expectation(S=None, n_gauss=12, degs=(0, 8)):
    S = S or S_DEFAULT
    rng = np.random.default_rng(5)
    L = degs[1] - degs[0]
    P = o360.BASIS_360.astype(np.float64)
    worst_gh, worst_mc = 0.0, 0.0
    for mu, Sig in _ipe_gaussians(rng, n_gauss):
        enc = o360.integrated_pos_enc_360((mu.astype(np.float32)[None], Sig.astype(np.float32)[None]), degs[0], degs[1])[0]
        enc = enc.reshape(2, L, 21).astype(np.float64)                       # [sin | cos], degree-major, then basis
        mu32, Sig32 = mu.astype(np.float32).astype(np.float64), Sig.astype(np.float32).astype(np.float64)
        ym = mu32 @ P
        yv = np.einsum("ik,ij,jk->k", P, Sig32, P)
        x = rng.multivariate_normal(mu32, Sig32, size=S) @ P                  # [S, 21]
        for li, l in enumerate(range(*degs)):
            # 128 nodes resolve sin(a y) over the Gaussian's bulk while a sigma <= 5 rad; beyond, |E| <= exp(-12.5) = 3.7e-6
            s_gh, c_gh = _gauss_hermite_expectation(ym, yv, 2.0 ** l, n=128)
            damped = 4.0 ** l * yv > 25.0
            s_gh, c_gh = np.where(damped, 0.0, s_gh), np.where(damped, 0.0, c_gh)
            # fp32 arithmetic of the oracle: the phase 2^l y carries |2^l y| * 6e-8 of rounding
            tol = 4e-6 + 4e-7 * 2.0 ** l * np.abs(ym).max()
            e = max(np.abs(enc[0, li] - s_gh).max(), np.abs(enc[1, li] - c_gh).max())
            worst_gh = max(worst_gh, e / tol)
            assert e <= tol, (l, e, tol)
            e_mc = max(np.abs(np.sin(2.0 ** l * x).mean(0) - enc[0, li]).max(), np.abs(np.cos(2.0 ** l * x).mean(0) - enc[1, li]).max())
            worst_mc = max(worst_mc, e_mc * np.sqrt(S))
            assert e_mc <= 4.5 / np.sqrt(S) + tol, (l, e_mc)                 # std of sin <= 0.71; 42 x L x n comparisons
    _rec("pin360 off-axis IPE vs Gaussian expectation", worst_quadrature_err_over_tol=worst_gh, worst_mc_err_in_sqrtS_units=worst_mc,
         samples=S, gaussians=n_gauss)


# ---- 6. frustum -> contraction -> encoding, end to end -----------------------------------------------------------------------
def test_contracted_frustum_encoding_vs_monte_carlo(S=None, n_rays=4, degs=(0, 7)):
    """The whole chain the kernels implement (k_cast_ipe_360): frustum -> full-covariance Gaussian -> contraction -> off-axis IPE,
    against E[sin / cos(2^l P^T contract(x))] with x drawn from the frustum's Gaussian and contracted POINTWISE.  The only modelling
    step between the two is the paper's linearisation, whose error is bounded through the validity radius of test 4."""
    S = S or S_DEFAULT
    rng = np.random.default_rng(6)
    L = degs[1] - degs[0]
    P = o360.BASIS_360.astype(np.float64)
    origins = rng.uniform(-0.5, 0.5, (n_rays, 3))
    d = rng.standard_normal((n_rays, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    radii = 10 ** rng.uniform(-3.3, -2.3, (n_rays, 1))
    # fence posts uniform in inverse depth like sample_along_rays_360 (near 0.5, far 30): four frustums per ray, from inside the unit
    # ball to |x| ~ 30
    s = np.array([0.0, 0.02, 0.3, 0.31, 0.9, 0.91, 0.997, 1.0])
    t = 1.0 / (s / 30.0 + (1 - s) / 0.5)
    t = np.broadcast_to(t, (n_rays, t.size)).astype(np.float32)
    mean_u, cov_u = o360.cast_rays_360(t, origins, d, radii, False)
    mean_c, cov_c = o360.cast_rays_360(t, origins, d, radii, True)
    enc = o360.integrated_pos_enc_360((mean_c, cov_c), degs[0], degs[1]).astype(np.float64)
    worst, worst_k = 0.0, 0.0
    for b in range(n_rays):
        for j in (0, 2, 4, 6):
            mu, Sig = mean_u[b, j].astype(np.float64), cov_u[b, j].astype(np.float64)
            kappa = np.sqrt(np.linalg.eigvalsh(Sig).max()) / np.linalg.norm(mu)
            x = contract64(rng.multivariate_normal(mu, Sig, size=S)) @ P
            e = enc[b, j].reshape(2, L, 21)
            for li, l in enumerate(range(*degs)):
                e_mc = max(np.abs(np.sin(2.0 ** l * x).mean(0) - e[0, li]).max(), np.abs(np.cos(2.0 ** l * x).mean(0) - e[1, li]).max())
                # linearisation: the mean moves by <= 0.6 kappa^2 |z| (|z| < 2), seen through a feature of slope 2^l
                bound = 4.5 / np.sqrt(S) + 1e-5 + 2.0 ** l * 0.6 * kappa ** 2 * 2.0 + 4.0 * kappa ** 2
                worst = max(worst, e_mc / bound)
                assert e_mc <= bound, (b, j, l, e_mc, bound, kappa)
            worst_k = max(worst_k, kappa)
    _rec("pin360 frustum -> contract -> IPE vs pointwise MC", worst_err_over_bound=worst, largest_kappa=worst_k, samples=S)


if __name__ == "__main__":
    full = "--full" in sys.argv
    S = int(2e6) if full else S_DEFAULT
    S_DEFAULT = S
    test_basis_is_the_twice_tessellated_icosahedron()
    test_frustum_moments_equal_the_integrals_over_the_cone()
    test_full_covariance_lift_vs_monte_carlo_over_the_frustum_volume()
    test_contraction_jacobian_by_finite_differences()
    test_contract_gaussian_vs_monte_carlo_pushforward(S=S, norms=(1.1, 1.3, 2.0, 4.0, 8.0, 12.0, 20.0, 30.0) if full else (1.3, 4.0, 30.0))
    test_off_axis_ipe_equals_the_gaussian_expectation(S=S, n_gauss=50 if full else 12)
    test_contracted_frustum_encoding_vs_monte_carlo(S=S, n_rays=12 if full else 4)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r04_parity.jsonl")
    with open(out, "a") as f:
        for tag, vals in RESIDUALS.items():
            f.write(json.dumps(dict(tag=tag + (" (full)" if full else ""), **vals)) + "\n")
    print(json.dumps(RESIDUALS, indent=1))

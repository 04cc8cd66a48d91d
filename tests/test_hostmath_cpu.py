"""CPU: the per-sample device math (mipnerf_pl_amd/csrc/raymath.hpp -- the very source inlined into
the gfx950 kernels) compiled with g++ and compared with the oracle / golden vectors."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import mipnerf_oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "hostmath", "hostmath.cpp")
SO = os.path.join(HERE, "hostmath", "_hostmath.so")


@pytest.fixture(scope="module")
def hm():
    hdr = os.path.join(HERE, "..", "mipnerf_pl_amd", "csrc", "raymath.hpp")
    hdr360 = os.path.join(HERE, "..", "mipnerf_pl_amd", "csrc", "raymath360.hpp")
    if (not os.path.exists(SO)) or os.path.getmtime(SO) < max(os.path.getmtime(SRC), os.path.getmtime(hdr), os.path.getmtime(hdr360)):
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", SRC, "-o", SO])
    return C.CDLL(SO)


def p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_frustum_and_ipe_match_golden(hm, golden_dir):
    g = dict(np.load(os.path.join(golden_dir, "stages_16x64_trained.npz")))
    t = g["t1"]                              # resampled (non-uniform) fence posts
    B, N1 = t.shape
    N = N1 - 1
    t0 = np.ascontiguousarray(t[:, :-1]).ravel()
    t1 = np.ascontiguousarray(t[:, 1:]).ravel()
    d = np.repeat(g["rays_directions"], N, axis=0).astype(np.float32)
    o = np.repeat(g["rays_origins"], N, axis=0).astype(np.float32)
    r = np.repeat(g["rays_radii"][:, 0], N).astype(np.float32)
    means = np.empty((B * N, 3), np.float32)
    covs = np.empty((B * N, 3), np.float32)
    hm.hm_cast(B * N, p(t0), p(t1), p(d), p(o), p(r), p(means), p(covs))
    # same operation order as torch => means bit-exact, covs within an ulp (pow vs mul)
    np.testing.assert_array_equal(means.reshape(B, N, 3), g["means1"])
    np.testing.assert_allclose(covs.reshape(B, N, 3), g["covs1"], rtol=3e-6, atol=1e-14)
    enc = np.empty((B * N, 96), np.float32)
    m = np.ascontiguousarray(g["means1"].reshape(-1, 3))
    c = np.ascontiguousarray(g["covs1"].reshape(-1, 3))
    hm.hm_ipe(B * N, p(m), p(c), 0, 16, p(enc))
    np.testing.assert_allclose(enc.reshape(B, N, 96), g["enc1"], rtol=0, atol=2e-6)


def test_view_encoding_and_activations(hm, golden_dir):
    g = dict(np.load(os.path.join(golden_dir, "stages_16x64_trained.npz")))
    v = np.ascontiguousarray(g["rays_viewdirs"])
    out = np.empty((v.shape[0], 27), np.float32)
    hm.hm_view(v.shape[0], p(v), 4, p(out))
    np.testing.assert_allclose(out, g["viewdirs_enc"], rtol=0, atol=1e-6)
    rr = np.ascontiguousarray(g["raw_rgb0"].ravel())
    n = rr.size
    dd = np.ascontiguousarray(np.resize(g["raw_density0"].ravel(), n))
    rgb = np.empty(n, np.float32)
    den = np.empty(n, np.float32)
    hm.hm_act.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    hm.hm_act(n, p(rr), p(dd), 0.001, -1.0, p(rgb), p(den))
    np.testing.assert_allclose(rgb, g["rgb0"].ravel(), rtol=0, atol=2e-7)
    np.testing.assert_allclose(den, orc.softplus(dd - np.float32(1)), rtol=2e-6, atol=1e-9)


@pytest.mark.parametrize("steps", [2, 3, 65, 129, 130, 257])
def test_linspace_matches_torch(hm, steps):
    import torch
    eps = float(np.finfo(np.float32).eps)
    hm.hm_linspace.argtypes = [C.c_float, C.c_float, C.c_int, C.c_void_p]
    for (a, b) in ((0.0, 1.0), (0.0, 1.0 - eps), (2.0, 6.0)):
        out = np.empty(steps, np.float32)
        hm.hm_linspace(a, b, steps, p(out))
        ref = torch.linspace(a, b, steps).numpy()
        np.testing.assert_array_equal(out, ref)      # bit-exact with torch
        np.testing.assert_array_equal(orc.torch_linspace(a, b, steps), out)


def _rays_360(B, seed):
    """Unbounded-scene rays: cameras near the origin looking outwards, near 0.2, far up to 1e3 (most samples beyond the unit ball)."""
    rng = np.random.default_rng(seed)
    o = rng.uniform(-0.5, 0.5, (B, 3)).astype(np.float32)
    d = rng.standard_normal((B, 3)).astype(np.float32)
    d *= rng.uniform(0.8, 1.2, (B, 1)).astype(np.float32) / np.linalg.norm(d, axis=-1, keepdims=True)
    r = rng.uniform(5e-4, 4e-3, (B, 1)).astype(np.float32)
    near = np.full((B, 1), 0.2, np.float32)
    far = rng.uniform(30.0, 1000.0, (B, 1)).astype(np.float32)
    return o, d, r, near, far


@pytest.mark.parametrize("contracted", [1, 0])
def test_mipnerf360_math_matches_oracle(hm, contracted):
    """raymath360.hpp (full-covariance frustum Gaussian, contraction of mean AND covariance, off-axis IPE) compiled with
    g++ against oracle/mipnerf360_oracle.py (paper equations; parity unpinned -- the reference's code for this is dead)."""
    from oracle import mipnerf360_oracle as o360
    B, N, L = 24, 32, 6
    o, d, r, near, far = _rays_360(B, 5)
    rng = np.random.default_rng(6)
    t_inv, t, (means, covs) = o360.sample_along_rays_360(o, d, r, N, near, far, True, t_rand=rng.uniform(0, 1, (B, N + 1)).astype(np.float32))
    assert np.all(np.diff(t, axis=-1) > 0) and np.all(np.diff(t_inv, axis=-1) < 0)
    want_m, want_c = o360.cast_rays_360(t, o, d, r, bool(contracted))
    if contracted:
        assert (np.linalg.norm(means, axis=-1) > 1).mean() > 0.15 and np.linalg.norm(want_m, axis=-1).max() < 2.0
    want_e = o360.integrated_pos_enc_360((want_m, want_c), 0, L)

    def run(mode):
        gm = np.empty((B * N, 3), np.float32)
        gc = np.empty((B * N, 9), np.float32)
        ge = np.empty((B * N, 2 * 21 * L), np.float32)
        hm.hm_cast_ipe_360(B * N, p(np.ascontiguousarray(t[:, :-1]).ravel()), p(np.ascontiguousarray(t[:, 1:]).ravel()),
                           p(np.repeat(d, N, axis=0)), p(np.repeat(o, N, axis=0)), p(np.repeat(r[:, 0], N)), mode, 0, L,
                           p(gm), p(gc), p(ge))
        return gm.reshape(B, N, 3), gc.reshape(B, N, 3, 3), ge.reshape(B, N, -1)
    gm, gc, ge = run(contracted)
    np.testing.assert_allclose(gm, want_m, rtol=2e-6, atol=2e-6)
    # covariance entries cancel heavily (|cov_ij| << trace for a thin frustum): compare relative to the largest entry.
    # fp32 limit of the fused (structured) form at |x| up to 1e3: 5e-5; the generic J C J^T on a rounded C: 4e-4
    scale = np.abs(want_c).max(axis=(-1, -2), keepdims=True)
    assert (np.abs(gc - want_c) / scale).max() <= (1e-4 if contracted else 2e-6)
    # features: a phase error of eps * 2^l * |y| radians is unavoidable in fp32 (|y| <= 2 after contraction; without it
    # |y| reaches 1e3 and the top frequencies are noise in ANY fp32 evaluation)
    if contracted:
        np.testing.assert_allclose(ge, want_e, rtol=0, atol=1e-4)
        gm2, gc2, _ = run(2)                                    # generic contract_gaussian(): same mean, looser covariance
        np.testing.assert_allclose(gm2, want_m, rtol=2e-6, atol=2e-6)
        assert (np.abs(gc2 - want_c) / scale).max() <= 1e-3
        inside = np.linalg.norm(means, axis=-1) <= 1            # known answer: inside the unit ball nothing changes
        m0, c0 = o360.cast_rays_360(t, o, d, r, False)
        assert inside.any() and np.array_equal(want_m[inside], m0[inside]) and np.array_equal(want_c[inside], c0[inside])
    else:
        lo = slice(0, 21 * 2)                                   # the two lowest frequencies of the sin half
        np.testing.assert_allclose(ge[..., lo], want_e[..., lo], rtol=0, atol=1e-3)

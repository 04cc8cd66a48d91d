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
    if (not os.path.exists(SO)) or os.path.getmtime(SO) < max(os.path.getmtime(SRC), os.path.getmtime(hdr)):
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

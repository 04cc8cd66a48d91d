"""CPU (no GPU): the C-ABI shared library loads, exports every symbol include/mipnerf_hip.h
declares, and its host-side plan expansion equals mipnerf_pl_amd/mlp_plan.py.  No compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from mipnerf_pl_amd import _lib as L
from mipnerf_pl_amd.mlp_plan import Plan, emulate_wave
from oracle import mipnerf_oracle as orc

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(L.LIB_PATH):
        from mipnerf_pl_amd import build
        build.build(verbose=False)
    return L.lib()


def header_functions():
    src = open(os.path.join(REPO, "include", "mipnerf_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mipnerf_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(lib):
    names = header_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mipnerf_hip.h but not exported"
    # and the Python binding covers exactly the header
    assert sorted(L.SIGNATURES) == names


def test_abi_version_and_compiled_arch(lib):
    assert lib.mipnerf_abi_version() == 2
    cfg = L.Config()
    assert lib.mipnerf_compiled_arch(C.byref(cfg)) == 0
    assert (cfg.net_depth, cfg.net_width, cfg.net_depth_condition, cfg.net_width_condition, cfg.skip_index) == (8, 256, 1, 128, 4)
    assert (cfg.max_deg_point, cfg.deg_view) == (16, 4)


def test_error_reporting_without_gpu(lib):
    # argument validation happens before any HIP call
    assert lib.mipnerf_sample_along_rays(0, 64, None, None, None, 0, None, None) == L.E_INVALID
    assert b"sample_along_rays" in lib.mipnerf_last_error()
    with pytest.raises(ValueError):
        L.check(L.E_INVALID, "x")
    with pytest.raises(NotImplementedError):
        L.check(L.E_UNSUPPORTED, "x")


@pytest.mark.parametrize("which,ref", [(0, "pack_table"), (1, "bias_table"), (2, "pack_table_f32")])
def test_host_plan_tables_match_python_plan(lib, which, ref):
    plan = Plan.build()
    want = getattr(plan, ref)().astype(np.int32).ravel()
    n = lib.mipnerf_debug_table(which, None, 0)
    assert n == want.size
    got = np.empty(n, np.int32)
    assert lib.mipnerf_debug_table(which, got.ctypes.data, n) == n
    np.testing.assert_array_equal(got, want)


def test_f32_layer_descriptors(lib):
    plan = Plan.build()
    n = lib.mipnerf_debug_f32net(None, 0)
    buf = np.empty((n, 8), np.int32)
    lib.mipnerf_debug_f32net(buf.ctypes.data, buf.size)
    layers = plan.f32_layers()
    assert n == len(layers)
    chunk0 = 0
    for row, Ly in zip(buf, layers):
        assert row[0] == Ly["x_in"] and row[1] == Ly["kb"] and row[2] == len(Ly["tiles"])
        assert row[3] == Ly["first_tile"] and row[4] == int(Ly["relu"]) and row[6] == chunk0
        chunk0 += len(Ly["tiles"]) * Ly["kb"]
        assert row[7] == 356
    assert [r[5] for r in buf] == [0] * 8 + [1, 0, 2]


def test_register_dataflow_emulation_matches_oracle():
    """The k-permutation / D-layout repacking the bf16 kernel relies on, emulated in numpy in fp32
    (no bf16 rounding), reproduces the oracle MLP to fp32 round-off."""
    plan = Plan.build()
    params = orc.make_params(seed=11, density_gain=10.0)
    names = [n for n, _ in plan.arch.param_shapes()]
    assert names == list(params.keys())
    flat = np.concatenate([params[n].ravel() for n in names])
    rng = np.random.default_rng(0)
    enc = rng.uniform(-1, 1, (32, 96)).astype(np.float32)
    v27 = rng.uniform(-1, 1, (32, 27)).astype(np.float32)
    view = np.zeros((32, 32), np.float32)
    view[:, :27] = v27
    rgb, dens = emulate_wave(plan, flat, enc, view)
    rr, dd = orc.mlp_forward(params, enc[:, None, :], v27)
    np.testing.assert_allclose(rgb, rr[:, 0], atol=5e-6)
    np.testing.assert_allclose(dens, dd[:, 0, 0], atol=2e-5)

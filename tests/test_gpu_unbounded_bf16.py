"""GPU: bf16 inference of the unbounded-scene model, `MipNerf(unbounded=True, precision='bf16')`.

Its 672-wide encoding does not fit k_mlp_bf16's wave-private LDS area, so the MLP runs as two kernels (csrc/gen_pre_gemm.py,
mlp_pre_plan.py): k_pre_gemm = layer 0 and the encoding half of the skip layer (models/mip_nerf.py:83-90), k-step-major; then the
trunk kernel.  Checked here through the C ABI:
  * the MLP alone (mipnerf_mlp_forward, row-major bf16 encodings) against the numpy model of bf16 operands / fp32 accumulation and
    against the fp32 oracle, at ragged sizes (partial wave tiles, partial 256-sample tiles, more tiles than workgroups);
  * the whole forward (mipnerf_forward: off-axis IPE written as MFMA fragments -> k_pre_gemm -> trunk -> compositing) against the fp32
    model on the same rays, and against the per-stage route (row-major encodings), which must agree bit for bit;
  * training entry points refuse this precision for this model."""
import numpy as np
import pytest
import torch

import synthetic_inputs as syn
from oracle import mipnerf_oracle as orc
from test_gpu_stages import mlp_bf16_emulation

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def G():
    import gpu_util
    assert torch.cuda.is_available()
    return gpu_util


def _model(params, N, precision):
    from mipnerf_pl_amd import MipNerf
    m = MipNerf(num_samples=N, unbounded=True, precision=precision)
    m.load_state_dict({"mlp." + k: torch.from_numpy(v.copy()) for k, v in params.items()}, strict=True)
    return m.to(DEV)


@pytest.mark.parametrize("shape", [(1, 1), (1, 33), (3, 43), (2, 128), (5, 77), (9, 256), (300, 64), (1030, 70)])
def test_mlp_two_kernel_form_vs_emulation_and_oracle(G, shape):
    B, N = shape
    params = syn.make_params(seed=31, density_gain=8.0, xyz_dim=672)
    model = _model(params, max(N, 1), "bf16")
    rng = np.random.default_rng(B * 1000 + N)
    enc = rng.uniform(-1, 1, (B, N, 672)).astype(np.float32)
    v27 = rng.uniform(-1, 1, (B, 27)).astype(np.float32)
    with torch.no_grad():
        rgb, den = model.mlp(torch.from_numpy(enc).to(DEV), torch.from_numpy(v27).to(DEV))
        rgb2, den2 = model.mlp(torch.from_numpy(enc).to(DEV), torch.from_numpy(v27).to(DEV))
    assert torch.equal(rgb, rgb2) and torch.equal(den, den2)                      # deterministic
    assert bool(torch.isfinite(rgb).all()) and bool(torch.isfinite(den).all())
    er_, ed_ = mlp_bf16_emulation(params, enc, v27)
    rr, dd = orc.mlp_forward(params, enc, v27)
    e = dict(vs_emulation_rgb=G.maxdiff(rgb, er_), vs_emulation_density=G.maxdiff(den, ed_), vs_fp32_rgb=G.maxdiff(rgb, rr),
             vs_fp32_density=G.maxdiff(den, dd), max_abs_density=float(np.abs(dd).max()))
    G.record(f"unbounded bf16 mlp {B}x{N}", **e)
    # the tolerances of the standard model's bf16 test (test_gpu_stages.test_mlp_bf16): vs the bf16 emulation only the accumulation
    # order differs (rare one-ulp flips of a bf16 activation); vs fp32 the bf16 rounding of 10 chained layers
    assert e["vs_emulation_rgb"] <= 6e-3 and e["vs_emulation_density"] <= 0.15
    assert e["vs_fp32_rgb"] <= 2e-2 and e["vs_fp32_density"] <= 0.4


def test_mlp_does_not_depend_on_the_grid(G):
    """1 ... 256 persistent workgroups share the tiles (both kernels): bit-identical outputs"""
    params = syn.make_params(seed=32, density_gain=8.0, xyz_dim=672)
    model = _model(params, 64, "bf16")
    rng = np.random.default_rng(3)
    enc = torch.from_numpy(rng.uniform(-1, 1, (70, 64, 672)).astype(np.float32)).to(DEV)      # 4480 samples = 17.5 tiles
    v = torch.from_numpy(rng.uniform(-1, 1, (70, 27)).astype(np.float32)).to(DEV)
    ctx = model.mlp.native(enc.device)
    outs = []
    with torch.no_grad():
        for grid in (256, 5, 1, 256):
            ctx.set_option(1, grid)
            outs.append(torch.cat(model.mlp(enc, v), -1).clone())
    ctx.set_option(1, 256)
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


@pytest.mark.parametrize("randomized", [False, True])
@pytest.mark.parametrize("B", [1, 5, 300])
def test_forward_bf16_vs_fp32_and_vs_per_stage_route(G, randomized, B):
    from mipnerf_pl_amd import _lib as L
    from mipnerf_pl_amd import ops
    N = 64                    # 64 / 320 / 19,200 samples: a quarter of a tile, 1.25 tiles, 75 tiles of 256
    rays = syn.synthetic_rays(B, seed=71, unbounded=True)
    params = syn.make_params(seed=17, density_gain=40.0, xyz_dim=672)
    rng = np.random.default_rng(6)
    T = lambda a: None if a is None else torch.from_numpy(a).to(DEV)     # noqa: E731
    tr = T(rng.uniform(0, 1, (B, N + 1)).astype(np.float32)) if randomized else None
    ur = T(rng.uniform(0, 1, (B, N + 1)).astype(np.float32)) if randomized else None
    m16, m32 = _model(params, N, "bf16"), _model(params, N, "fp32")
    R = G.to_dev(rays)
    with torch.no_grad():
        got = m16(R, randomized, True, t_rand=tr, u_rand=ur)
        ref = m32(R, randomized, True, t_rand=tr, u_rand=ur)
    errs = {}
    for lvl in range(2):
        for nm, a, b in zip(G.NAMES, got[lvl], ref[lvl]):
            errs[f"l{lvl}_{nm}"] = G.maxdiff(a, b)
    mse = float(torch.mean((got[1][0] - ref[1][0]) ** 2))
    errs["psnr_fine_rgb_db"] = float(-10 * np.log10(max(mse, 1e-20)))
    G.record(f"unbounded bf16 forward vs fp32 B={B} randomized={randomized}", **errs)
    # the coarse level sees the same fence posts: colours within the bf16 tolerance of the standard model's full-size test (3e-2 max,
    # PSNR >= 55 dB); the fine level's fence posts move with the coarse weights, so it is held by PSNR only
    assert errs["l0_rgb"] <= 3e-2 and errs["l0_acc"] <= 3e-2
    assert errs["psnr_fine_rgb_db"] >= 50.0                  # measured 55-66 dB
    for lvl in range(2):
        assert all(bool(torch.isfinite(t).all()) for t in got[lvl])
    # per-stage route of the coarse level: row-major bf16 encodings -> mipnerf_mlp_forward -> compositing; the forward call wrote the same
    # encodings as fragments, so every output bit must agree
    t0 = got[0][4]
    with torch.no_grad():
        enc = ops.cast_ipe_360(t0, R.origins, R.directions, R.radii, m16.min_deg_point, m16.max_deg_point, precision=L.PREC_BF16)
        venc = ops.pos_enc(R.viewdirs, 0, m16.deg_view)
        _, _, act = m16.mlp(enc, venc, return_activated=True)
        comp = ops.volumetric_rendering_packed(act, t0, R.directions, True)
    assert torch.equal(comp[0], got[0][0]) and torch.equal(comp[3], got[0][3])


def test_train_in_fp32_render_in_bf16(G):
    """set_precision switches the arithmetic of a live model: an fp32 model switched to bf16 renders exactly what a bf16 model with the same
    weights renders (the context packs the streams of both precisions), and switches back"""
    params = syn.make_params(seed=17, density_gain=40.0, xyz_dim=672)
    rays = G.to_dev(syn.synthetic_rays(50, seed=9, unbounded=True))
    m16, m32 = _model(params, 64, "bf16"), _model(params, 64, "fp32")
    with torch.no_grad():
        a = m16(rays, False, True)
        b32 = m32(rays, False, True)
        b = m32.set_precision("bf16")(rays, False, True)
        c32 = m32.set_precision("fp32")(rays, False, True)
    for lvl in range(2):
        for x, y in zip(a[lvl], b[lvl]):
            assert torch.equal(x, y)
        for x, y in zip(b32[lvl], c32[lvl]):
            assert torch.equal(x, y)
    assert not torch.equal(a[1][0], b32[1][0])
    with pytest.raises(ValueError):
        m32.set_precision("fp16")


def test_training_entry_points_refuse_bf16_for_this_model(G):
    params = syn.make_params(seed=17, density_gain=40.0, xyz_dim=672)
    m16 = _model(params, 64, "bf16")
    rays = G.to_dev(syn.synthetic_rays(8, seed=2, unbounded=True))
    with pytest.raises(NotImplementedError, match="inference only"):
        m16(rays, True, True)                              # parameters require grad: the autograd route
    with pytest.raises(NotImplementedError):
        m16.train_step_native(rays, torch.zeros(8, 3, device=DEV), True, True)

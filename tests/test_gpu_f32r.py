"""GPU: the register-resident fp32 MLP kernel k_mlp_f32r (csrc/gen_mlp_f32r.py) against the oracle and against the LDS-resident
k_mlp_f32 of rounds 1-3 (mipnerf_set_option(ctx, 5, 0)), which every fp32 golden was first met with.

* ragged / tiny sample counts: a launch of 1, 31, 33, 127, 129, ... samples (partial 128-sample tiles, partial 32-sample wave tiles,
  fewer tiles than workgroups, many tiles per workgroup) must equal the oracle's MLP (models/mip_nerf.py:75-111) and the other kernel;
* every architecture variant the kernel is generated for, incl. the 672-wide unbounded-scene encoding (streamed natural blocks);
* the outputs do not depend on how many persistent workgroups share the tiles.
"""
import numpy as np
import pytest
import torch

import synthetic_inputs as syn
from oracle import mipnerf_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def G():
    import gpu_util
    assert torch.cuda.is_available()
    return gpu_util


def _run(model, enc, venc, resident):
    """mipnerf_mlp_forward in fp32 through the chosen kernel -> (raw [B,N,4], activated [B,N,4])"""
    from mipnerf_pl_amd import _lib as L
    from mipnerf_pl_amd import ops
    B, N, _ = enc.shape
    ctx = model.mlp.native(enc.device)
    ctx.set_option(5, 1 if resident else 0)
    v32 = torch.zeros(B, 32, device=enc.device)
    if venc is not None:
        v32[:, :venc.shape[-1]] = venc
    act = torch.empty(B, N, 4, device=enc.device)
    raw = torch.empty_like(act)
    L.check(L.lib().mipnerf_mlp_forward(ctx.handle, B * N, N, enc.contiguous().data_ptr(), v32.data_ptr(), L.PREC_FP32, act.data_ptr(),
                                        raw.data_ptr(), ops._stream()), "mlp_forward")
    ctx.set_option(5, 1)
    return raw, act


@pytest.mark.parametrize("shape", [(1, 1), (1, 31), (1, 33), (3, 43), (1, 127), (1, 129), (2, 64), (5, 77), (257, 3), (300, 128)])
def test_ragged_sample_counts_equal_oracle_and_lds_kernel(G, shape):
    B, N = shape
    params = syn.make_params(seed=21, density_gain=8.0)
    model = G.make_model(params, max(N, 1), "fp32")
    rng = np.random.default_rng(B * 1000 + N)
    enc = rng.uniform(-1, 1, (B, N, 96)).astype(np.float32)
    v27 = rng.uniform(-1, 1, (B, 27)).astype(np.float32)
    e, v = torch.from_numpy(enc).to(DEV), torch.from_numpy(v27).to(DEV)
    with torch.no_grad():
        raw_r, act_r = _run(model, e, v, True)
        raw_l, act_l = _run(model, e, v, False)
    rr, dd = orc.mlp_forward(params, enc, v27)
    want = np.concatenate([rr, dd], -1)
    err_r = float(np.max(np.abs(raw_r.cpu().numpy() - want)))
    err_l = float(np.max(np.abs(raw_l.cpu().numpy() - want)))
    G.record(f"f32r ragged {B}x{N}", resident_vs_oracle=err_r, lds_vs_oracle=err_l,
             resident_vs_lds=G.maxdiff(raw_r, raw_l), act=G.maxdiff(act_r, act_l))
    assert err_r <= 2e-5 and err_l <= 2e-5, (err_r, err_l)            # |raw| up to ~10 with density_gain 8: a few ulps of the sums
    assert G.maxdiff(act_r[..., :3], act_l[..., :3]) <= 2e-6
    assert bool(torch.isfinite(act_r).all())


VARIANTS = {"w128": dict(mlp_net_width=128, mlp_net_width_condition=128),
            "noview": dict(mlp_net_width_condition=256, use_viewdirs=False),
            "d6s3": dict(mlp_net_depth=6, mlp_skip_index=3),
            "dc2": dict(mlp_net_depth_condition=2)}


@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_every_generated_variant_equals_oracle_and_lds_kernel(G, name):
    kw = VARIANTS[name]
    arch = dict(net_width=kw.get("mlp_net_width", 256), net_width_condition=kw.get("mlp_net_width_condition", 128),
                net_depth=kw.get("mlp_net_depth", 8), skip_index=kw.get("mlp_skip_index", 4),
                net_depth_condition=kw.get("mlp_net_depth_condition", 1))
    params = syn.make_params(seed=22, density_gain=8.0, **arch)
    model = G.make_model(params, 64, "fp32", **kw)
    rng = np.random.default_rng(5)
    B, N = 7, 53
    enc = rng.uniform(-1, 1, (B, N, 96)).astype(np.float32)
    v27 = rng.uniform(-1, 1, (B, 27)).astype(np.float32)
    use_view = kw.get("use_viewdirs", True)
    e, v = torch.from_numpy(enc).to(DEV), torch.from_numpy(v27).to(DEV)
    with torch.no_grad():
        raw_r, _ = _run(model, e, v if use_view else None, True)
        raw_l, _ = _run(model, e, v if use_view else None, False)
    rr, dd = orc.mlp_forward(params, enc, v27 if use_view else None, skip_index=arch["skip_index"], net_depth=arch["net_depth"],
                             net_depth_condition=arch["net_depth_condition"])
    want = np.concatenate([rr, dd], -1)
    err_r = float(np.max(np.abs(raw_r.cpu().numpy() - want)))
    G.record(f"f32r variant {name}", resident_vs_oracle=err_r, resident_vs_lds=G.maxdiff(raw_r, raw_l))
    assert err_r <= 2e-5 and G.maxdiff(raw_r, raw_l) <= 2e-5


def test_unbounded_variant_streams_its_natural_blocks(G):
    """672 encoding features = 21 natural blocks per reader, cycled through the four wave-private slots (43 fetches per tile)"""
    params = syn.make_params(seed=23, density_gain=8.0, xyz_dim=672)
    from mipnerf_pl_amd import MipNerf
    model = MipNerf(num_samples=64, precision="fp32", unbounded=True)
    model.load_state_dict({"mlp." + k: torch.from_numpy(v.copy()) for k, v in params.items()})
    model = model.to(DEV)
    rng = np.random.default_rng(6)
    for (B, N) in ((2, 37), (9, 128)):
        enc = rng.uniform(-1, 1, (B, N, 672)).astype(np.float32)
        v27 = rng.uniform(-1, 1, (B, 27)).astype(np.float32)
        e, v = torch.from_numpy(enc).to(DEV), torch.from_numpy(v27).to(DEV)
        with torch.no_grad():
            raw_r, _ = _run(model, e, v, True)
            raw_l, _ = _run(model, e, v, False)
        rr, dd = orc.mlp_forward(params, enc, v27)
        want = np.concatenate([rr, dd], -1)
        err = float(np.max(np.abs(raw_r.cpu().numpy() - want)))
        G.record(f"f32r unbounded {B}x{N}", resident_vs_oracle=err, resident_vs_lds=G.maxdiff(raw_r, raw_l))
        assert err <= 3e-5 and G.maxdiff(raw_r, raw_l) <= 3e-5


def test_forward_is_deterministic_and_independent_of_the_grid(G):
    """the same samples through 1 ... 256 persistent workgroups (tiles per workgroup 1 ... many): bit-identical outputs"""
    params = syn.make_params(seed=24, density_gain=8.0)
    model = G.make_model(params, 64, "fp32")
    rng = np.random.default_rng(7)
    B, N = 40, 64            # 2560 samples = 20 tiles
    enc = torch.from_numpy(rng.uniform(-1, 1, (B, N, 96)).astype(np.float32)).to(DEV)
    v = torch.from_numpy(rng.uniform(-1, 1, (B, 27)).astype(np.float32)).to(DEV)
    ctx = model.mlp.native(enc.device)
    outs = []
    with torch.no_grad():
        for grid in (256, 7, 1, 256):
            ctx.set_option(1, grid)
            outs.append(_run(model, enc, v, True)[0].clone())
    ctx.set_option(1, 256)
    for o in outs[1:]:
        assert torch.equal(o, outs[0])

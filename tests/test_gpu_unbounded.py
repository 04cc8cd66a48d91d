"""SURVEY 8(f)-4 as a MODEL: `MipNerf(unbounded=True)` -- fence posts uniform in inverse depth, fine-level
resampling over the inverse-depth fence posts, contracted full-covariance Gaussians, off-axis IPE (42 features per degree: the
MLP's first layer / skip concat are 672 wide for 16 degrees), the reference MLP, compositing -- end to end in fp32 against
oracle/mipnerf360_oracle.mipnerf360_forward (+ the existing oracle MLP / sampler / compositing).

What pins the oracle (scripts/make_golden.py --only-360, golden pin360_24x64): the parts of the reference's dead 360 code that
are RIGHT -- `contract` (mip.py:424-428, |x| > 1) and the inverse-depth fence posts + Gaussian means of
`sample_along_rays_360` (mip.py:106-124), deterministic and randomized.  Its full covariances (mip.py:38-47), `parameterization`
(mip.py:431-447) and `integrated_pos_enc_360` (mip.py:292-319: no frequency scales) are wrong upstream: those stages follow the
paper and stay "parity unpinned" (oracle header)."""
import numpy as np
import pytest
import torch

import synthetic_inputs as syn
from oracle import mipnerf360_oracle as o360
from oracle import mipnerf_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ARCH = dict(xyz_dim=672)


@pytest.fixture(scope="module")
def G():
    import gpu_util
    assert torch.cuda.is_available()
    return gpu_util


def _model(params, N, **kw):
    from mipnerf_pl_amd import MipNerf
    m = MipNerf(num_samples=N, unbounded=True, **kw)
    assert m.mlp.arch["xyz_dim"] == 672 and tuple(m.mlp.layers[0][0].weight.shape) == (256, 672)
    assert tuple(m.mlp.layers[5][0].weight.shape) == (256, 256 + 672)
    missing, unexpected = m.load_state_dict({"mlp." + k: torch.from_numpy(v.copy()) for k, v in params.items()}, strict=True)
    assert not missing and not unexpected
    return m.to(DEV)


def test_kernels_match_the_reference_where_the_reference_is_right(G):
    """contract() and the fence posts / means of sample_along_rays_360 straight against the reference's outputs."""
    from mipnerf_pl_amd import ops
    g = G.load_golden("pin360_24x64")
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)     # noqa: E731
    x = g["contract_x"]
    outside = np.linalg.norm(x, axis=-1) > 1
    y = ops.contract(T(x)).cpu().numpy()
    assert np.abs(y[outside] - g["contract_y"][outside]).max() <= 5e-7
    assert np.array_equal(y[~outside], x[~outside])                      # inside the unit ball: identity (mip.py:441 masks it)
    R = G.rays_of(g)
    N = int(g["num_samples"])
    for tag, randomized, tr in (("det", False, None), ("rand", True, T(g["t_rand"]))):
        t_inv, (means, covs) = ops.sample_along_rays_360(T(R.origins), T(R.directions), T(R.radii), N, T(R.near), T(R.far), randomized,
                                                         False, "cone", t_rand=tr)
        e_t = G.maxdiff(t_inv, g[f"{tag}_t_inv"])
        e_m = G.maxdiff(means, g[f"{tag}_means"]) / float(np.abs(g[f"{tag}_means"]).max())
        G.record(f"pin360 {tag}", t_inv_abs=e_t, means_rel=e_m)
        assert e_t <= 2.4e-7 * float(np.abs(g[f"{tag}_t_inv"]).max()) and e_m <= 1e-6


@pytest.mark.parametrize("randomized", [False, True])
@pytest.mark.parametrize("white,shape", [(True, (40, 64)), (False, (40, 64)), (True, (3, 600)), (False, (7, 45))])     # 600: the K = 16 bucket (round 5)
def test_unbounded_model_forward_vs_oracle(G, randomized, white, shape):
    B, N = shape
    rays = syn.synthetic_rays(B, seed=61, unbounded=True)
    params = syn.make_params(seed=17, density_gain=40.0, **ARCH)
    rng = np.random.default_rng(5)
    tr = rng.uniform(0, 1, (B, N + 1)).astype(np.float32) if randomized else None
    ur = rng.uniform(0, 1, (B, N + 1)).astype(np.float32) if randomized else None
    want, stages = o360.mipnerf360_forward(params, rays, randomized, white, num_samples=N, t_rand=tr, u_rand=ur, return_stages=True)
    model = _model(params, N)
    T = lambda a: None if a is None else torch.from_numpy(a).to(DEV)     # noqa: E731
    with torch.no_grad():
        got = model(G.to_dev(rays), randomized, white, t_rand=T(tr), u_rand=T(ur))
    errs = {}
    for lvl in range(2):
        for nm, a, b in zip(G.NAMES, got[lvl], want[lvl]):
            errs[f"l{lvl}_{nm}"] = G.maxdiff(a, b)
    G.record(f"unbounded forward B={B} N={N} randomized={randomized} white={white}", **errs)
    # level 0: nothing upstream but the fence posts (bit-exact) and the encoding (1e-5): fp32 MLP accuracy
    assert errs["l0_t_samples"] <= 1e-6 * float(np.abs(want[0][4]).max())
    assert errs["l0_rgb"] <= 5e-5 and errs["l0_acc"] <= 5e-5 and errs["l0_weights"] <= 5e-5
    # level 1: the resampled inverse depths move by an ulp (t = 1 / t_inv amplifies it by t^2 at the far end)
    far = float(np.abs(want[1][4]).max())
    assert errs["l1_t_samples"] <= 5e-5 * far                  # measured 2.4e-5 * far = 1e-6 of the inverse depth
    assert errs["l1_rgb"] <= 2e-4 and errs["l1_acc"] <= 2e-4
    assert errs["l1_distance"] <= 2e-4 * far
    # (bf16 inference of this architecture: tests/test_gpu_unbounded_bf16.py; bf16 TRAINING is refused loudly there)


def _torch_volumetric_rendering(rgb, density, t_samples, dirs, white_bkgd):
    """torch restatement of models/mip.py:366-401 (differentiable; same as the one in test_gpu_train.py)."""
    t_mids = 0.5 * (t_samples[..., :-1] + t_samples[..., 1:])
    delta = (t_samples[..., 1:] - t_samples[..., :-1]) * torch.linalg.norm(dirs[:, None, :], dim=-1)
    dd = density[..., 0] * delta
    alpha = 1 - torch.exp(-dd)
    trans = torch.exp(-torch.cat([torch.zeros_like(dd[..., :1]), torch.cumsum(dd[..., :-1], -1)], -1))
    w = alpha * trans
    comp = (w[..., None] * rgb).sum(-2)
    acc = w.sum(-1)
    dist = torch.clamp(torch.nan_to_num((w * t_mids).sum(-1)), t_samples[:, 0], t_samples[:, -1])
    if white_bkgd:
        comp = comp + (1. - acc[..., None])
    return comp, dist, acc, w


def test_unbounded_model_trains_in_fp32(G):
    """One training step through autograd: loss + gradients of all 24 tensors against torch autograd over a plain-PyTorch
    restatement of MLP + activations + compositing + the loss of nerf_system.py:99-111 fed with the native encodings / fence
    posts (the sampler carries no gradient: stop_resample_grad), then a few optimiser steps must lower the loss."""
    from mipnerf_pl_amd import ops
    from mipnerf_pl_amd.autograd import distloss
    from mipnerf_pl_amd.system import DEFAULT_HPARAMS, MipNeRFSystem
    B, N = 48, 32
    rays_np = syn.synthetic_rays(B, seed=62, unbounded=True, multiscale=True)
    params = syn.make_params(seed=18, density_gain=20.0, **ARCH)
    hp = dict(DEFAULT_HPARAMS)
    hp.update({"nerf.num_samples": N, "nerf.unbounded": True, "train.randomized": False, "optimizer.lr_init": 1e-3,
               "optimizer.lr_delay_steps": 0})
    system = MipNeRFSystem(hp, precision="fp32")
    system.load_state_dict({"mip_nerf.mlp." + k: torch.from_numpy(v.copy()) for k, v in params.items()})
    system = system.to(DEV)
    rays = G.to_dev(rays_np)
    gt = torch.rand(B, 3, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
    loss = system.training_step((rays, gt), 0)
    loss.backward()
    mine = {k: p.grad.detach().clone() for k, p in system.mip_nerf.mlp.named_parameters()}
    # ---- torch restatement on the same encodings
    model = system.mip_nerf
    with torch.no_grad():
        ret = model(rays, False, True)
    venc = ops.pos_enc(rays.viewdirs, 0, 4, True, precision=0, ld=32)
    system.zero_grad(set_to_none=True)
    losses, dls = [], []
    for lvl in range(2):
        t = ret[lvl][4]
        enc = ops.cast_ipe_360(t, rays.origins, rays.directions, rays.radii, 0, 16, contracted=True)
        raw = G.mlp_torch(model.mlp, enc, venc[:, :27], torch.float32)
        rgb = torch.sigmoid(raw[..., :3]) * (1 + 2 * 0.001) - 0.001
        sigma = torch.nn.functional.softplus(raw[..., 3:] - 1.0)
        comp, _, _, w = _torch_volumetric_rendering(rgb, sigma, t, rays.directions, True)
        mask = rays.lossmult
        losses.append((mask * (comp - gt) ** 2).sum() / mask.sum())
        dls.append(distloss(w, t))
    ref_loss = 0.1 * (losses[0] + 0.01 * dls[0]) + losses[1] + 0.01 * dls[1]
    ref_loss.backward()
    assert abs(float(ref_loss) - float(loss)) <= 2e-5 * max(1.0, abs(float(ref_loss)))
    worst = 0.0
    for k, p in system.mip_nerf.mlp.named_parameters():
        a, b = mine[k].double().reshape(-1), p.grad.double().reshape(-1)
        rel = float((a - b).norm() / b.norm().clamp_min(1e-30))
        worst = max(worst, rel)
        assert rel <= 2e-3, (k, rel)
    G.record("unbounded training step fp32", loss=float(loss), worst_grad_rel_l2=worst)
    system.zero_grad(set_to_none=True)
    (opt,), (sch,) = system.configure_optimizers()
    first = None
    for it in range(8):
        opt.zero_grad()
        l_ = system.training_step((rays, gt), it)
        l_.backward()
        opt.step()
        sch["scheduler"].step()
        first = float(l_) if first is None else first
    assert float(l_) < first
    # rendering through the system hook works on the unbounded model too (chunked, ragged tail)
    system.val_chunk_size = 20
    img_rays = type(rays)(*[x.reshape(1, 6, 8, -1) for x in rays])
    c_rgb, f_rgb, _ = system.render_image((img_rays, torch.zeros(1, 6, 8, 3, device=DEV)))
    assert f_rgb.shape == (1, 6, 8, 3) and bool(torch.isfinite(f_rgb).all())
    # ... and replayed from one captured hipGraph (model.GraphedFrame) it gives the same frame
    system.enable_hip_graph(True)
    c2, f2, _ = system.render_image((img_rays, torch.zeros(1, 6, 8, 3, device=DEV)))
    assert torch.equal(f2, f_rgb) and torch.equal(c2, c_rgb)
    # trained in fp32, rendered in bf16 (round 4: k_pre_gemm + trunk kernel): the live model switched over; the captured frame notices the
    # switch, re-captures, and equals the eager bf16 frame; close to the fp32 frame
    system.mip_nerf.set_precision("bf16")
    c3, f3, _ = system.render_image((img_rays, torch.zeros(1, 6, 8, 3, device=DEV)))
    system.enable_hip_graph(False)
    c4, f4, _ = system.render_image((img_rays, torch.zeros(1, 6, 8, 3, device=DEV)))
    assert torch.equal(f3, f4) and torch.equal(c3, c4) and not torch.equal(f3, f_rgb)
    assert float((f3 - f_rgb).abs().max()) <= 3e-2
    system.mip_nerf.set_precision("fp32")


@pytest.mark.parametrize("shape", [(1, 1), (5, 13), (40, 64)])
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_tiled_encoding_kernel_equals_the_per_direction_kernel(G, shape, precision):
    """mipnerf_cast_ipe_360 asked for the encoding only runs the tiled kernel (one Gaussian per sample, vector stores; round 4); asked for the
    Gaussians as well it runs the per-(sample, direction) kernel of round 3.  Same per-feature expressions: the same bits, in both output types,
    at ragged sample counts (the row-major buffer ends exactly at the last sample: nothing may be written behind it)."""
    import ctypes as C
    from mipnerf_pl_amd import _lib as L
    from mipnerf_pl_amd import ops
    B, N = shape
    rays = G.to_dev(syn.synthetic_rays(B, seed=B + N, unbounded=True))
    t_inv, t = ops.sample_t_360(N, rays.near, rays.far, False)
    prec = L.PREC_BF16 if precision == "bf16" else L.PREC_FP32
    dt = torch.bfloat16 if precision == "bf16" else torch.float32
    guard = 64
    a = torch.full((B * N * 672 + guard,), 7.0, device=DEV, dtype=dt)          # canary behind the last row
    b = torch.empty(B * N * 672, device=DEV, dtype=dt)
    means = torch.empty(B, N, 3, device=DEV)
    covs = torch.empty(B, N, 3, 3, device=DEV)
    o, d, r = rays.origins.contiguous(), rays.directions.contiguous(), rays.radii.contiguous()
    P = lambda x: C.c_void_p(x.data_ptr())                                       # noqa: E731
    L.check(L.lib().mipnerf_cast_ipe_360(B, N, 0, 16, 1, P(t), P(o), P(d), P(r), P(a), prec, None, None, ops._stream()), "tiled")
    L.check(L.lib().mipnerf_cast_ipe_360(B, N, 0, 16, 1, P(t), P(o), P(d), P(r), P(b), prec, P(means), P(covs), ops._stream()), "per-direction")
    torch.cuda.synchronize()
    assert torch.equal(a[:B * N * 672], b)
    assert bool((a[B * N * 672:] == 7.0).all())
    assert bool(torch.isfinite(b.float()).all())

"""MipNerf(stop_resample_grad=False) (mip.py:265-279): the gradient that reaches the coarse weights through the resampled
fence posts.  Every native backward piece against torch autograd over a plain-torch restatement of the same formulas (test
reference only), then the whole training step against the reference's own gradients (tests/golden/var_resamplegrad_48x64.npz)."""
import math

import numpy as np
import pytest
import torch

from oracle import mipnerf_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
F32_EPS = float(np.finfo(np.float32).eps)


def rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


# ---- plain-torch restatements (differentiable) -------------------------------------------------------------------------------
def t_distloss_rays(w, t):
    iv = t[:, 1:] - t[:, :-1]
    m = 0.5 * (t[:, 1:] + t[:, :-1])
    uni = (iv * w * w).sum(-1) / 3.0
    bi = (w[:, :, None] * w[:, None, :] * (m[:, :, None] - m[:, None, :]).abs()).sum((-1, -2))
    return uni + bi


def t_render(rgb, sigma, t, dirs, white):
    t_mids = 0.5 * (t[:, :-1] + t[:, 1:])
    delta = (t[:, 1:] - t[:, :-1]) * dirs.norm(dim=-1, keepdim=True)
    x = sigma * delta
    alpha = 1 - torch.exp(-x)
    trans = torch.exp(-torch.cat([torch.zeros_like(x[:, :1]), torch.cumsum(x[:, :-1], -1)], -1))
    w = alpha * trans
    comp = (w[..., None] * rgb).sum(1)
    acc = w.sum(-1)
    dist = (w * t_mids).sum(-1)
    dist = torch.maximum(torch.minimum(torch.nan_to_num(dist, nan=float("inf")), t[:, -1]), t[:, 0])
    if white:
        comp = comp + (1.0 - acc[:, None])
    return comp, dist, acc, w


def t_cast_ipe(t, o, d, r, min_deg, max_deg, disable_integration=False):
    t0, t1 = t[:, :-1], t[:, 1:]
    mu, hw = (t0 + t1) / 2, (t1 - t0) / 2
    den = 3 * mu ** 2 + hw ** 2
    t_mean = mu + 2 * mu * hw ** 2 / den
    t_var = hw ** 2 / 3 - (4 / 15) * (hw ** 4 * (12 * mu ** 2 - hw ** 2)) / den ** 2
    r_var = r ** 2 * (mu ** 2 / 4 + (5 / 12) * hw ** 2 - (4 / 15) * hw ** 4 / den)
    mean = d[:, None, :] * t_mean[..., None] + o[:, None, :]
    d2 = d ** 2
    null = 1 - d2 / (d2.sum(-1, keepdim=True) + 1e-10)
    cov = t_var[..., None] * d2[:, None, :] + r_var[..., None] * null[:, None, :]
    if disable_integration:
        cov = torch.zeros_like(cov)
    scales = torch.tensor([2.0 ** i for i in range(min_deg, max_deg)], device=t.device, dtype=t.dtype)
    y = (mean[..., None, :] * scales[:, None]).flatten(-2)
    yv = (cov[..., None, :] * scales[:, None] ** 2).flatten(-2)
    x = torch.cat([y, y + 0.5 * math.pi], -1)
    return torch.exp(-0.5 * torch.cat([yv, yv], -1)) * torch.sin(x)


def t_resample(bins, weights, padding, u):
    """mip.py:250-271 + 168-229 in float32 (the index of every draw has to be the forward kernel's)."""
    wp = torch.cat([weights[:, :1], weights, weights[:, -1:]], -1)
    mx = torch.maximum(wp[:, :-1], wp[:, 1:])
    w = 0.5 * (mx[:, :-1] + mx[:, 1:]) + padding
    s = w.sum(-1, keepdim=True)
    pad = torch.clamp(1e-5 - s, min=0)
    w = w + pad / w.shape[-1]
    s = s + pad
    pdf = w / s
    cdf = torch.clamp(torch.cumsum(pdf[:, :-1], -1), max=1.0)
    cdf = torch.cat([torch.zeros_like(cdf[:, :1]), cdf, torch.ones_like(cdf[:, :1])], -1)
    inds = torch.searchsorted(cdf.detach().contiguous(), u.contiguous(), right=True)
    below, above = (inds - 1).clamp_min(0), inds.clamp_max(cdf.shape[-1] - 1)
    c0, c1 = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
    b0, b1 = torch.gather(bins, 1, below), torch.gather(bins, 1, above)
    den = c1 - c0
    den = torch.where(den < 1e-5, torch.ones_like(den), den)
    return b0 + (u - c0) / den * (b1 - b0)


def draws(B, N, randomized, seed):
    n = N + 1
    if not randomized:
        return None, torch.linspace(0.0, 1.0 - F32_EPS, n, device=DEV)[None].expand(B, n).contiguous()
    g = torch.Generator(device=DEV).manual_seed(seed)
    r = torch.rand(B, n, device=DEV, generator=g)
    s = 1.0 / n
    u = torch.arange(n, device=DEV, dtype=torch.float32)[None] * np.float32(s) + r * np.float32(s - F32_EPS)
    return r, torch.clamp(u, max=1.0 - F32_EPS)


def some_rays(B, seed):
    rays = orc.synthetic_rays(B, seed=seed, multiscale=True)
    return [torch.from_numpy(np.asarray(a)).to(DEV) for a in rays]


def sorted_t(B, N, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return (2.0 + 4.0 * torch.sort(torch.rand(B, N + 1, device=DEV, generator=g), -1).values).contiguous()


# ---- pieces ----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,N", [(5, 7), (3, 64), (4, 130), (2, 257), (2, 600), (1, 1024)])       # 600 / 1024: the K = 16 bucket (round 5)
def test_distloss_and_compositing_gradient_wrt_t(B, N):
    from mipnerf_pl_amd.autograd import distloss, render_from_raw
    g = torch.Generator(device=DEV).manual_seed(B * 10 + N)
    t = sorted_t(B, N, 1).requires_grad_(True)
    w = torch.rand(B, N, device=DEV, generator=g).requires_grad_(True)
    distloss(w, t).backward()
    td, wd = t.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
    t_distloss_rays(wd, td).mean().backward()
    assert rel(t.grad, td.grad) <= 2e-5 and rel(w.grad, wd.grad) <= 2e-5
    # compositing: raw -> activations -> weights / rgb / distance, all four outputs feeding the loss
    raw = torch.randn(B, N, 4, device=DEV, generator=g).requires_grad_(True)
    dirs = torch.randn(B, 3, device=DEV, generator=g)
    coef = [torch.randn(B, 3, device=DEV, generator=g), torch.randn(B, device=DEV, generator=g),
            torch.randn(B, device=DEV, generator=g), torch.randn(B, N, device=DEV, generator=g)]
    for white in (True, False):
        t2 = t.detach().clone().requires_grad_(True)
        raw.grad = None
        out = render_from_raw(raw, t2, dirs, white, 0.001, -1.0)
        sum((o * c).sum() for o, c in zip(out, coef)).backward()
        rd, tdd = raw.detach().double().requires_grad_(True), t2.detach().double().requires_grad_(True)
        rgb = torch.sigmoid(rd[..., :3]) * 1.002 - 0.001
        sig = torch.nn.functional.softplus(rd[..., 3] - 1.0)
        ref = t_render(rgb, sig, tdd, dirs.double(), white)
        sum((o * c.double()).sum() for o, c in zip(ref, coef)).backward()
        assert rel(t2.grad, tdd.grad) <= 5e-5, (white, rel(t2.grad, tdd.grad))
        assert rel(raw.grad, rd.grad) <= 5e-5


@pytest.mark.parametrize("B,N,noint", [(6, 16, False), (3, 64, False), (4, 33, True)])
def test_cast_ipe_backward(B, N, noint):
    from mipnerf_pl_amd.autograd import _CastIPE
    o, d, _, r = some_rays(B, 4)[:4]
    t = sorted_t(B, N, 2).requires_grad_(True)
    g = torch.Generator(device=DEV).manual_seed(5)
    # low degrees only carry a usable signal at these scales; weight the features like a trained first layer would (decaying)
    coef = torch.randn(B, N, 96, device=DEV, generator=g) * (0.5 ** torch.arange(16, device=DEV).repeat_interleave(3).repeat(2))
    enc = _CastIPE.apply(t, o, d, r, 0, 16, noint)
    (enc * coef).sum().backward()
    td = t.detach().double().requires_grad_(True)
    ref = t_cast_ipe(td, o.double(), d.double(), r.double(), 0, 16, noint)
    # same features: fp32 phases up to 2^15 |x| rad vs float64 (without the integration nothing damps the top degrees: 1e-2 rad)
    assert float((enc.detach().double() - ref.detach()).abs().max()) <= (5e-2 if noint else 5e-3)
    (ref * coef.double()).sum().backward()
    assert rel(t.grad, td.grad) <= (3e-2 if noint else 2e-3), rel(t.grad, td.grad)   # the fp32 Gaussian feeds 2^15-rad phases


@pytest.mark.parametrize("B,N,randomized,padding", [(7, 9, False, 0.01), (5, 64, True, 0.01), (4, 128, False, 0.01),
                                                     (3, 200, True, 0.05), (3, 16, False, 0.0), (2, 800, True, 0.01)])
def test_resample_backward(B, N, randomized, padding):
    from mipnerf_pl_amd.autograd import _ResampleT
    g = torch.Generator(device=DEV).manual_seed(N)
    bins = sorted_t(B, N, 3)
    w = torch.rand(B, N, device=DEV, generator=g) ** 3
    w[:, N // 3] += 2.0                                     # a spike: many draws in one bin, plateaus in the blur pool
    if padding == 0.0:
        w[0] = 0.0                                          # all-zero ray: the 1e-5 padding branch (mip.py:181-185)
        w[1, : N // 2] = 0.0
    w.requires_grad_(True)
    r, u = draws(B, N, randomized, 11)
    coef = torch.randn(B, N + 1, device=DEV, generator=g)
    t_new = _ResampleT.apply(bins, w, padding, r)
    (t_new * coef).sum().backward()
    w2 = w.detach().clone().requires_grad_(True)
    ref = t_resample(bins, w2, padding, u)
    assert float((t_new.detach() - ref.detach()).abs().max()) <= 2e-5
    (ref * coef).sum().backward()
    assert rel(w.grad, w2.grad) <= 2e-3, rel(w.grad, w2.grad)       # fp32 autograd of the same fp32 forward


def test_mlp_input_gradient():
    import gpu_util as G
    from mipnerf_pl_amd.autograd import mlp_native_f32
    params = orc.make_params(seed=2, density_gain=10.0)
    model = G.make_model(params, 16, "fp32")
    g = torch.Generator(device=DEV).manual_seed(1)
    B, N = 9, 16
    enc = (torch.randn(B, N, 96, device=DEV, generator=g) * 0.5).requires_grad_(True)
    venc = torch.zeros(B, 32, device=DEV)
    venc[:, :27] = torch.randn(B, 27, device=DEV, generator=g)
    coef = torch.randn(B, N, 4, device=DEV, generator=g)
    (mlp_native_f32(model.mlp, enc, venc) * coef).sum().backward()
    got = {k: p.grad.clone() for k, p in model.mlp.named_parameters()}
    e2 = enc.detach().double().requires_grad_(True)
    model.zero_grad()
    (G.mlp_torch(model.mlp, e2, venc[:, :27], torch.float64).double() * coef.double()).sum().backward()
    assert rel(enc.grad, e2.grad) <= 1e-4, rel(enc.grad, e2.grad)
    for k, p in model.mlp.named_parameters():
        assert rel(got[k], p.grad) <= 1e-3, k


# ---- the whole step against the reference ----------------------------------------------------------------------------------------
def test_training_step_matches_reference_gradients():
    import gpu_util as G
    from mipnerf_pl_amd.system import DEFAULT_HPARAMS, MipNeRFSystem
    g = G.load_golden("var_resamplegrad_48x64")
    params = orc.make_params(seed=int(g["param_seed"]), density_gain=float(g["density_gain"]))
    hp = dict(DEFAULT_HPARAMS)
    hp.update({"nerf.num_samples": int(g["num_samples"]), "train.randomized": False, "nerf.stop_resample_grad": False})
    rays, gt = G.to_dev(G.rays_of(g)), torch.from_numpy(g["gt"]).to(DEV)
    system = MipNeRFSystem(hp, precision="fp32")
    system.load_state_dict({"mip_nerf.mlp." + k: torch.from_numpy(v.copy()) for k, v in params.items()})
    system = system.to(DEV)
    loss = system.training_step((rays, gt), 0)
    loss.backward()
    assert abs(float(loss.detach()) - float(g["loss"])) <= 2e-5 * max(1.0, float(g["loss"]))
    worst = 0.0
    for k, p in system.mip_nerf.mlp.named_parameters():
        grad = p.grad.detach().cpu().numpy().ravel()
        l2 = float(g["g_l2_" + k])
        stride = max(1, grad.size // 64)
        smp = grad[::stride][:64]
        scale = max(float(np.abs(g["g_smp_" + k]).max()), l2 / np.sqrt(grad.size), 1e-12)
        err = float(np.max(np.abs(smp - g["g_smp_" + k]))) / scale
        worst = max(worst, err)
        assert abs(float(np.sqrt((grad.astype(np.float64) ** 2).sum())) - l2) <= 5e-3 * max(l2, 1e-9), (k, l2)
        assert err <= 5e-3, (k, err)
    G.record("stop_resample_grad_false_vs_reference", worst_grad_rel=worst, loss=float(loss.detach()))
    # the flag matters: with the stop-gradient resampler the trunk / density gradients are different by ~100 % (not a no-op test)
    hp["nerf.stop_resample_grad"] = True
    s2 = MipNeRFSystem(hp, precision="fp32")
    s2.load_state_dict(system.state_dict())
    s2 = s2.to(DEV)
    s2.training_step((rays, gt), 0).backward()
    a = system.mip_nerf.mlp.density_layer.weight.grad
    b = s2.mip_nerf.mlp.density_layer.weight.grad
    assert float((a - b).norm() / a.norm()) > 0.3
    # bf16 precision and the one-call native step say what they do not do
    s3 = MipNeRFSystem(hp | {"nerf.stop_resample_grad": False}, precision="bf16").to(DEV)
    with pytest.raises(NotImplementedError, match="fp32"):
        s3.training_step((rays, gt), 0)
    with pytest.raises(NotImplementedError, match="autograd"):
        s3.training_step_native((rays, gt), 0)


def test_ops_resample_along_rays_stop_grad_false():
    """The free function of mip.py:232-280 with stop_grad=False: new_t_vals carries the gradient to the weights."""
    from mipnerf_pl_amd import ops
    B, N = 6, 40
    o, d, _, r = some_rays(B, 9)[:4]
    bins = sorted_t(B, N, 8)
    w = torch.rand(B, N, device=DEV, generator=torch.Generator(device=DEV).manual_seed(2)).requires_grad_(True)
    t_new, (means, covs) = ops.resample_along_rays(o, d, r, bins, w, False, "cone", False, 0.01)
    assert t_new.requires_grad and means.shape == (B, N, 3) and covs.shape == (B, N, 3)
    coef = torch.randn(B, N + 1, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
    (t_new * coef).sum().backward()
    w2 = w.detach().clone().requires_grad_(True)
    (t_resample(bins, w2, 0.01, draws(B, N, False, 0)[1]) * coef).sum().backward()
    assert rel(w.grad, w2.grad) <= 2e-3
    t_sg, _ = ops.resample_along_rays(o, d, r, bins, w, False, "cone", True, 0.01)      # the shipped mode: no graph
    assert not t_sg.requires_grad and torch.equal(t_sg, t_new.detach())
    # the Gaussians of the stop_grad=False branch must not silently drop the gradient the reference keeps
    # (mip.py:265-280): differentiating them outside MipNerf raises
    assert means.requires_grad and covs.requires_grad
    with pytest.raises(NotImplementedError, match="cast_ipe"):
        (means.sum() + covs.sum()).backward()

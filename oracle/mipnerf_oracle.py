"""CPU oracle for the Mip-NeRF volume-rendering hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a plain numpy (fp32) restatement of the reference's algorithm
(hjxwhy/mipnerf_pl, models/mip.py + models/mip_nerf.py + the loss of
models/nerf_system.py).  It exists only so that tests/, __graft_entry__.smoke()
and bench.py's `cpu_baseline` leg have something to check / time the HIP path
against.  Nothing in `mipnerf_pl_amd/` (the product) imports it.

Parity status: PINNED.  The reference holds no tests or golden vectors of its
own, so the pin is against outputs of the reference itself:
`scripts/make_golden.py` imports the unmodified reference from /root/reference
on CPU, feeds it seeded inputs and stores inputs+outputs under tests/golden/;
`tests/test_oracle_golden.py` checks every function below against those files.

Every function cites the reference file:line it follows.  All arithmetic is
float32 like the reference (numpy keeps float32 through python-scalar ops).
"""
from __future__ import annotations

import collections
import os
import sys

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
# seeded inputs shared with bench.py and scripts/make_golden.py (synthetic_inputs.py at the repo root);
# `Rays` is the record of datasets/datasets.py:13-16
from synthetic_inputs import Rays, make_params, param_shapes, synthetic_rays  # noqa: E402,F401

F32 = np.float32

EPS32 = np.finfo(np.float32).eps  # torch.finfo(torch.float32).eps


def _f32(x):
    return np.asarray(x, dtype=F32)


def torch_linspace(start, end, steps):
    """torch.linspace(start, end, steps) for float32 (ATen RangeFactories: value =
    start + step*i for i < steps/2, else end - step*(steps-1-i), with the multiply-add FUSED,
    as ATen's CPU (FMA) and CUDA (-fmad) builds do; checked bit-for-bit against torch in
    tests/test_hostmath_cpu.py).  The fused op is emulated in float64 (exact product)."""
    start = F32(start)
    end = F32(end)
    if steps == 1:
        return np.array([start], dtype=F32)
    step = F32((end - start) / F32(steps - 1))
    i = np.arange(steps)
    half = steps // 2
    lo = (np.float64(start) + np.float64(step) * i).astype(F32)
    hi = (np.float64(end) - np.float64(step) * (steps - 1 - i)).astype(F32)
    return np.where(i < half, lo, hi).astype(F32)


# --------------------------------------------------------------------------- mip.py
def lift_gaussian(directions, t_mean, t_var, r_var):
    """models/mip.py:22-36 (diagonal branch)."""
    directions = _f32(directions)
    mean = directions[..., None, :] * t_mean[..., None]
    d_norm_denominator = np.sum(directions ** 2, axis=-1, keepdims=True, dtype=F32) + F32(1e-10)
    d_outer_diag = directions ** 2
    null_outer_diag = F32(1) - d_outer_diag / d_norm_denominator
    t_cov_diag = t_var[..., None] * d_outer_diag[..., None, :]
    xy_cov_diag = r_var[..., None] * null_outer_diag[..., None, :]
    cov_diag = t_cov_diag + xy_cov_diag
    return mean.astype(F32), cov_diag.astype(F32)


def conical_frustum_to_gaussian(directions, t0, t1, base_radius):
    """models/mip.py:50-78 (stable branch 65-72, diagonal)."""
    t0 = _f32(t0)
    t1 = _f32(t1)
    base_radius = _f32(base_radius)
    mu = (t0 + t1) / F32(2)
    hw = (t1 - t0) / F32(2)
    t_mean = mu + (F32(2) * mu * hw ** 2) / (F32(3) * mu ** 2 + hw ** 2)
    t_var = (hw ** 2) / F32(3) - F32(4 / 15) * ((hw ** 4 * (F32(12) * mu ** 2 - hw ** 2)) /
                                               (F32(3) * mu ** 2 + hw ** 2) ** 2)
    r_var = base_radius ** 2 * ((mu ** 2) / F32(4) + F32(5 / 12) * hw ** 2 - F32(4 / 15) *
                                (hw ** 4) / (F32(3) * mu ** 2 + hw ** 2))
    return lift_gaussian(directions, t_mean.astype(F32), t_var.astype(F32), r_var.astype(F32))


def cast_rays(t_samples, origins, directions, radii, ray_shape="cone"):
    """models/mip.py:81-103."""
    if ray_shape != "cone":
        raise NotImplementedError  # mip.py:97-98
    t0 = t_samples[..., :-1]
    t1 = t_samples[..., 1:]
    means, covs = conical_frustum_to_gaussian(directions, t0, t1, radii)
    means = means + _f32(origins)[..., None, :]
    return means.astype(F32), covs


def sample_along_rays(origins, directions, radii, num_samples, near, far, randomized,
                      disparity, ray_shape="cone", t_rand=None):
    """models/mip.py:127-165.  `t_rand` ([B, N+1] uniform [0,1)) replaces the
    reference's torch.rand draw (mip.py:159) so the stratified path is testable."""
    batch_size = origins.shape[0]
    near = _f32(near)
    far = _f32(far)
    t = torch_linspace(0., 1., num_samples + 1)
    if disparity:
        t_samples = F32(1.) / (F32(1.) / near * (F32(1.) - t) + F32(1.) / far * t)
    else:
        t_samples = near + (far - near) * t
    t_samples = t_samples.astype(F32)
    if randomized:
        mids = F32(0.5) * (t_samples[..., 1:] + t_samples[..., :-1])
        upper = np.concatenate([mids, t_samples[..., -1:]], -1)
        lower = np.concatenate([t_samples[..., :1], mids], -1)
        assert t_rand is not None and t_rand.shape == (batch_size, num_samples + 1)
        t_samples = (lower + (upper - lower) * _f32(t_rand)).astype(F32)
    else:
        t_samples = np.broadcast_to(t_samples, (batch_size, num_samples + 1)).astype(F32)
    means, covs = cast_rays(t_samples, origins, directions, radii, ray_shape)
    return t_samples, (means, covs)


def sorted_piecewise_constant_pdf(bins, weights, num_samples, randomized, u_rand=None):
    """models/mip.py:168-229.  Does NOT mutate `weights` (the reference does, in
    place, mip.py:184; its caller always passes a fresh tensor).  `u_rand`
    ([B, num_samples] in [0,1)) stands in for the uniform_ draw of mip.py:201:
    jitter = u_rand * (s - eps32)."""
    bins = _f32(bins)
    weights = _f32(weights).copy()
    eps = F32(1e-5)
    weight_sum = np.sum(weights, axis=-1, keepdims=True, dtype=F32)
    padding = np.maximum(F32(0), eps - weight_sum)
    weights = weights + padding / F32(weights.shape[-1])
    weight_sum = weight_sum + padding

    pdf = (weights / weight_sum).astype(F32)
    # torch CPU cumsum accumulates float in double and rounds each prefix (ATen
    # cumsum_cpu_kernel uses at::acc_type<float,false> = double).
    cdf = np.cumsum(pdf[..., :-1].astype(np.float64), axis=-1).astype(F32)
    cdf = np.minimum(F32(1), cdf)
    shp = list(cdf.shape[:-1]) + [1]
    cdf = np.concatenate([np.zeros(shp, F32), cdf, np.ones(shp, F32)], axis=-1)

    if randomized:
        s = 1 / num_samples                       # python float (double), mip.py:199
        u = (np.arange(num_samples).astype(F32) * F32(s))[None, :]
        assert u_rand is not None
        u = u + _f32(u_rand) * F32(s - float(EPS32))   # uniform_(to=s-eps): x * float32(to)
        u = np.minimum(u, F32(1. - EPS32)).astype(F32)
    else:
        u = torch_linspace(0., 1. - EPS32, num_samples)
        u = np.broadcast_to(u, list(cdf.shape[:-1]) + [num_samples]).astype(F32)

    # torch.searchsorted(cdf, u, right=True): first index with cdf[idx] > u
    inds = np.empty(u.shape, dtype=np.int64)
    for b in range(u.shape[0]):
        inds[b] = np.searchsorted(cdf[b], u[b], side="right")
    below = np.maximum(0, inds - 1)
    above = np.minimum(cdf.shape[-1] - 1, inds)
    cdf_g0 = np.take_along_axis(cdf, below, axis=-1)
    cdf_g1 = np.take_along_axis(cdf, above, axis=-1)
    bins_g0 = np.take_along_axis(bins, below, axis=-1)
    bins_g1 = np.take_along_axis(bins, above, axis=-1)
    denom = cdf_g1 - cdf_g0
    denom = np.where(denom < F32(1e-5), F32(1), denom)
    t = (u - cdf_g0) / denom
    samples = bins_g0 + t * (bins_g1 - bins_g0)
    return samples.astype(F32)


def resample_along_rays(origins, directions, radii, t_samples, weights, randomized,
                        ray_shape="cone", stop_grad=True, resample_padding=0.01, u_rand=None):
    """models/mip.py:232-280."""
    weights = _f32(weights)
    weights_pad = np.concatenate([weights[..., :1], weights, weights[..., -1:]], axis=-1)
    weights_max = np.maximum(weights_pad[..., :-1], weights_pad[..., 1:])
    weights_blur = F32(0.5) * (weights_max[..., :-1] + weights_max[..., 1:])
    w = (weights_blur + F32(resample_padding)).astype(F32)
    new_t_vals = sorted_piecewise_constant_pdf(t_samples, w, t_samples.shape[-1], randomized,
                                               u_rand=u_rand)
    means, covs = cast_rays(new_t_vals, origins, directions, radii, ray_shape)
    return new_t_vals, (means, covs)


def expected_sin_mean(x, x_var):
    """models/mip.py:283-289, element [0] only (the hot path discards y_var, mip.py:350)."""
    return (np.exp(F32(-0.5) * x_var) * np.sin(x)).astype(F32)


def integrated_pos_enc(means_covs, min_deg, max_deg):
    """models/mip.py:322-350 (diagonal).  Feature index = half*3L + l*3 + axis."""
    means, covs_diag = means_covs
    scales = np.array([2 ** i for i in range(min_deg, max_deg)], dtype=F32)
    shp = list(means.shape[:-1]) + [-1]
    y = (means[..., None, :] * scales[:, None]).reshape(shp).astype(F32)
    y_var = (covs_diag[..., None, :] * scales[:, None] ** 2).reshape(shp).astype(F32)
    half_pi = F32(0.5) * F32(np.pi)  # 0.5 * torch.tensor(np.pi) -> float32
    x = np.concatenate([y, (y + half_pi).astype(F32)], axis=-1)
    xv = np.concatenate([y_var, y_var], axis=-1)
    return expected_sin_mean(x, xv)


def pos_enc(x, min_deg, max_deg, append_identity=True):
    """models/mip.py:353-363."""
    x = _f32(x)
    scales = np.array([2 ** i for i in range(min_deg, max_deg)], dtype=F32)
    xb = (x[..., None, :] * scales[:, None]).reshape(list(x.shape[:-1]) + [-1]).astype(F32)
    half_pi = F32(0.5) * F32(np.pi)
    four_feat = np.sin(np.concatenate([xb, (xb + half_pi).astype(F32)], axis=-1)).astype(F32)
    if append_identity:
        return np.concatenate([x, four_feat], axis=-1)
    return four_feat


def volumetric_rendering(rgb, density, t_samples, dirs, white_bkgd):
    """models/mip.py:366-401."""
    t_samples = _f32(t_samples)
    t_mids = F32(0.5) * (t_samples[..., :-1] + t_samples[..., 1:])
    t_interval = t_samples[..., 1:] - t_samples[..., :-1]
    dnorm = np.sqrt(np.sum(_f32(dirs) ** 2, axis=-1, dtype=F32)).astype(F32)[..., None]
    delta = t_interval * dnorm
    density_delta = (density[..., 0] * delta).astype(F32)
    alpha = F32(1) - np.exp(-density_delta)
    csum = np.cumsum(density_delta[..., :-1].astype(np.float64), axis=-1).astype(F32)
    trans = np.exp(-np.concatenate([np.zeros_like(density_delta[..., :1]), csum], axis=-1))
    weights = (alpha * trans).astype(F32)
    comp_rgb = (weights[..., None] * rgb).sum(axis=-2, dtype=F32)
    acc = weights.sum(axis=-1, dtype=F32)
    distance = (weights * t_mids).sum(axis=-1, dtype=F32)
    distance = np.clip(np.nan_to_num(distance), t_samples[:, 0], t_samples[:, -1])
    if white_bkgd:
        comp_rgb = comp_rgb + (F32(1.) - acc[..., None])
    return comp_rgb.astype(F32), distance.astype(F32), acc.astype(F32), weights


def distloss(weight, samples):
    """models/mip.py:8-20 (O(N^2) form, as the reference)."""
    weight = _f32(weight)
    samples = _f32(samples)
    interval = samples[:, 1:] - samples[:, :-1]
    mid_points = (samples[:, 1:] + samples[:, :-1]) * F32(0.5)
    loss_uni = F32(1 / 3) * (interval * weight ** 2).sum(-1, dtype=F32).mean(dtype=F32)
    ww = weight[..., :, None] * weight[..., None, :]
    mm = np.abs(mid_points[..., :, None] - mid_points[..., None, :])
    loss_bi = (ww * mm).sum((-1, -2), dtype=F32).mean(dtype=F32)
    return F32(loss_uni + loss_bi)


# ----------------------------------------------------------------------- mip_nerf.py
def mlp_forward(params, x, view_direction=None, skip_index=4, net_depth=8, net_depth_condition=1):
    """models/mip_nerf.py:75-111."""
    inputs = x
    for i in range(net_depth):
        x = x @ params[f"layers.{i}.0.weight"].T + params[f"layers.{i}.0.bias"]
        x = np.maximum(x, F32(0))
        if i % skip_index == 0 and i > 0:
            x = np.concatenate([x, inputs], axis=-1)
    raw_density = x @ params["density_layer.weight"].T + params["density_layer.bias"]
    if view_direction is not None:
        bottleneck = x @ params["extra_layer.weight"].T + params["extra_layer.bias"]
        vd = np.broadcast_to(view_direction[:, None, :],
                             (x.shape[0], x.shape[1], view_direction.shape[-1]))
        x = np.concatenate([bottleneck, vd], axis=-1)
        for i in range(net_depth_condition):
            x = x @ params[f"view_layers.{i}.0.weight"].T + params[f"view_layers.{i}.0.bias"]
            x = np.maximum(x, F32(0))
    raw_rgb = x @ params["color_layer.weight"].T + params["color_layer.bias"]
    return raw_rgb.astype(F32), raw_density.astype(F32)


def mlp_backward(params, x, view_direction, d_raw_rgb, d_raw_density, skip_index=4, net_depth=8,
                 net_depth_condition=1):
    """Gradients of MLP.forward (models/mip_nerf.py:75-111) w.r.t. its 24 parameter tensors for
    the upstream gradients d_raw_rgb [B,N,3], d_raw_density [B,N,1]: what torch autograd derives
    from those lines (pinned against the reference's autograd by tests/golden/mlp_bwd_*.npz).
    Returns an OrderedDict with the keys / shapes of `params`."""
    B, N = x.shape[0], x.shape[1]
    inputs = x.reshape(B * N, -1).astype(F32)
    # forward, keeping the input of every Linear (mip_nerf.py:88-110)
    acts = []
    h = inputs
    for i in range(net_depth):
        acts.append(h)
        h = np.maximum(h @ params[f"layers.{i}.0.weight"].T + params[f"layers.{i}.0.bias"], F32(0))
        if i % skip_index == 0 and i > 0:
            h = np.concatenate([h, inputs], axis=-1)
    trunk = h
    g = collections.OrderedDict((k, None) for k in params)
    d_rgb = d_raw_rgb.reshape(B * N, -1).astype(F32)
    d_den = d_raw_density.reshape(B * N, -1).astype(F32)
    g["density_layer.weight"] = d_den.T @ trunk
    g["density_layer.bias"] = d_den.sum(0)
    if view_direction is None:
        # MLP.forward(x, None) (mip_nerf.py:99-110): the colour head reads the trunk output; extra_layer / view_layers are
        # unused (autograd leaves their .grad None: zeros here)
        g["color_layer.weight"] = d_rgb.T @ trunk
        g["color_layer.bias"] = d_rgb.sum(0)
        for k in params:
            if k.startswith(("extra_layer", "view_layers")):
                g[k] = np.zeros_like(params[k])
        d_bott_term = d_rgb @ params["color_layer.weight"]
    else:
        bottleneck = trunk @ params["extra_layer.weight"].T + params["extra_layer.bias"]
        vd = np.repeat(view_direction.astype(F32), N, axis=0)
        hv = np.concatenate([bottleneck, vd], axis=-1)
        vacts = []
        for i in range(net_depth_condition):
            vacts.append(hv)
            hv = np.maximum(hv @ params[f"view_layers.{i}.0.weight"].T + params[f"view_layers.{i}.0.bias"], F32(0))
        g["color_layer.weight"] = d_rgb.T @ hv
        g["color_layer.bias"] = d_rgb.sum(0)
        d = d_rgb @ params["color_layer.weight"]
        for i in reversed(range(net_depth_condition)):
            d = d * (hv > 0)                                   # ReLU of view layer i (hv is its output)
            g[f"view_layers.{i}.0.weight"] = d.T @ vacts[i]
            g[f"view_layers.{i}.0.bias"] = d.sum(0)
            d = d @ params[f"view_layers.{i}.0.weight"]
            hv = vacts[i]
        d_bott = d[:, :bottleneck.shape[1]]                    # the view features get no gradient
        g["extra_layer.weight"] = d_bott.T @ trunk
        g["extra_layer.bias"] = d_bott.sum(0)
        d_bott_term = d_bott @ params["extra_layer.weight"]
    d = d_bott_term + d_den @ params["density_layer.weight"]
    out = trunk
    width = params["layers.0.0.weight"].shape[0]
    for i in reversed(range(net_depth)):
        d = d[:, :width] * (out[:, :width] > 0)            # drop the skip-concat columns, ReLU of layer i
        g[f"layers.{i}.0.weight"] = d.T @ acts[i]
        g[f"layers.{i}.0.bias"] = d.sum(0)
        if i > 0:
            d = d @ params[f"layers.{i}.0.weight"]
            out = acts[i]
    return collections.OrderedDict((k, v.astype(F32)) for k, v in g.items())


def softplus(x):
    """torch.nn.Softplus(beta=1, threshold=20)."""
    x = _f32(x)
    return np.where(x > F32(20), x, np.log1p(np.exp(np.minimum(x, F32(20))))).astype(F32)


def sigmoid(x):
    x = _f32(x)
    return (F32(1) / (F32(1) + np.exp(-x))).astype(F32)


def mipnerf_forward(params, rays, randomized, white_bkgd, num_samples=128, num_levels=2,
                    resample_padding=0.01, stop_resample_grad=True, use_viewdirs=True,
                    disparity=False, ray_shape="cone", min_deg_point=0, max_deg_point=16,
                    deg_view=4, density_noise=0., density_bias=-1., rgb_padding=0.001,
                    disable_integration=False, skip_index=4, net_depth=8,
                    net_depth_condition=1, t_rand=None, u_rand=None, return_stages=False,
                    density_randn=None):
    """models/mip_nerf.py:172-248.  Returns the list of per-level 5-tuples
    (comp_rgb, distance, acc, weights, t_samples)."""
    ret = []
    stages = []
    t_samples, weights = None, None
    for i_level in range(num_levels):
        if i_level == 0:
            t_samples, means_covs = sample_along_rays(
                rays.origins, rays.directions, rays.radii, num_samples, rays.near, rays.far,
                randomized, disparity, ray_shape, t_rand=t_rand)
        else:
            t_samples, means_covs = resample_along_rays(
                rays.origins, rays.directions, rays.radii, t_samples, weights, randomized,
                ray_shape, stop_resample_grad, resample_padding=resample_padding, u_rand=u_rand)
        if disable_integration:
            means_covs = (means_covs[0], np.zeros_like(means_covs[1]))
        samples_enc = integrated_pos_enc(means_covs, min_deg_point, max_deg_point)
        if use_viewdirs:
            viewdirs_enc = pos_enc(rays.viewdirs, 0, deg_view, True)
            raw_rgb, raw_density = mlp_forward(params, samples_enc, viewdirs_enc, skip_index,
                                               net_depth, net_depth_condition)
        else:
            raw_rgb, raw_density = mlp_forward(params, samples_enc, None, skip_index,
                                               net_depth, net_depth_condition)
        # mip_nerf.py:232-233: raw_density += density_noise * randn (only when randomized and density_noise > 0);
        # density_randn [num_levels, B, N] carries the standard-normal draws (the reference draws them with torch.randn)
        if randomized and density_noise > 0:
            if density_randn is None:
                raise ValueError("density_noise > 0 needs the standard-normal draws `density_randn`")
            z = _f32(density_randn[i_level]).reshape(raw_density.shape)
            raw_density = (raw_density + F32(density_noise) * z).astype(F32)
        rgb = sigmoid(raw_rgb)
        rgb = (rgb * F32(1 + 2 * rgb_padding) - F32(rgb_padding)).astype(F32)
        density = softplus(raw_density + F32(density_bias))
        comp_rgb, distance, acc, weights = volumetric_rendering(
            rgb, density, t_samples, rays.directions, white_bkgd)
        ret.append((comp_rgb, distance, acc, weights, t_samples))
        if return_stages:
            stages.append(dict(t_samples=t_samples, means=means_covs[0], covs=means_covs[1],
                               samples_enc=samples_enc, raw_rgb=raw_rgb, raw_density=raw_density,
                               rgb=rgb, density=density))
    if return_stages:
        return ret, stages
    return ret


# --------------------------------------------------------------------- nerf_system.py
def training_loss(ret, rays, rgbs, coarse_loss_mult=0.1, disable_multiscale_loss=False):
    """models/nerf_system.py:99-111."""
    mask = _f32(rays.lossmult)
    if disable_multiscale_loss:
        mask = np.ones_like(mask)
    losses, distlosses = [], []
    for (rgb, _, _, weights, t_samples) in ret:
        losses.append(F32((mask * (rgb - rgbs[..., :3]) ** 2).sum(dtype=F32) / mask.sum(dtype=F32)))
        distlosses.append(distloss(weights, t_samples))
    mse_coarse, mse_fine = losses
    return F32(F32(coarse_loss_mult) * (mse_coarse + F32(0.01) * distlosses[0])
               + mse_fine + F32(0.01) * distlosses[-1])


def calc_psnr(x, y):
    """utils/metrics.py:182-188."""
    return F32(-10.0) * np.log10(np.mean((_f32(x) - _f32(y)) ** 2, dtype=F32))


# ------------------------------------------------------------------ synthetic inputs
def eval_errors(pred, gt):
    """utils/metrics.py:191-197: (psnr, mean ssim) of [H,W,3] images; SSIM = metrics.py:44-126 with the 11x11 Gaussian
    window of metrics.py:10-41 (sigma 1.5), zero padding 5, C1 = 0.01^2, C2 = 0.03^2."""
    pred, gt = np.asarray(pred, np.float64), np.asarray(gt, np.float64)
    psnr = -10.0 * np.log10(np.mean((pred - gt) ** 2))
    g = np.exp(-((np.arange(11) - 5) ** 2) / (2 * 1.5 ** 2)).astype(F32)
    g = (g / g.sum()).astype(np.float64)
    win = np.outer(g, g)

    def filt(img):
        H, W, _ = img.shape
        p = np.pad(img, ((5, 5), (5, 5), (0, 0)))
        out = np.zeros_like(img)
        for dy in range(11):
            for dx in range(11):
                out += win[dy, dx] * p[dy:dy + H, dx:dx + W]
        return out
    mu1, mu2 = filt(pred), filt(gt)
    s1, s2, s12 = filt(pred * pred) - mu1 ** 2, filt(gt * gt) - mu2 ** 2, filt(pred * gt) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    ssim_map = ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 ** 2 + mu2 ** 2 + C1) * (s1 + s2 + C2))
    return F32(psnr), F32(ssim_map.mean())


def generate_rays_blender(c2w, width, height, focal, near, far):
    """datasets/datasets.py:214-263 (Blender._generate_rays) for ONE camera: Rays of [H, W, k] float32."""
    x, y = np.meshgrid(np.arange(width, dtype=F32), np.arange(height, dtype=F32), indexing="xy")
    focal = F32(focal)
    camera_dirs = np.stack([(x - F32(width * 0.5) + F32(0.5)) / focal,
                            -(y - F32(height * 0.5) + F32(0.5)) / focal, -np.ones_like(x)], axis=-1).astype(F32)
    return _finish_rays(camera_dirs, np.asarray(c2w, F32), F32(1), near, far)


def generate_rays_multicam(c2w, pix2cam, width, height, near, far, lossmult):
    """datasets/datasets.py:116-168 (Multicam._generate_rays) for ONE camera."""
    x, y = np.meshgrid(np.arange(width, dtype=F32) + F32(.5), np.arange(height, dtype=F32) + F32(.5), indexing="xy")
    pixel_dirs = np.stack([x, y, np.ones_like(x)], axis=-1)
    camera_dirs = (pixel_dirs @ np.asarray(pix2cam, F32)[:3, :3].T).astype(F32)
    return _finish_rays(camera_dirs, np.asarray(c2w, F32), lossmult, near, far)


def _finish_rays(camera_dirs, c2w, lossmult, near, far):
    directions = (camera_dirs @ c2w[:3, :3].T).astype(F32)
    origins = np.broadcast_to(c2w[:3, -1], directions.shape).astype(F32)
    viewdirs = (directions / np.linalg.norm(directions, axis=-1, keepdims=True)).astype(F32)
    dx = np.sqrt(np.sum((directions[:-1] - directions[1:]) ** 2, -1))        # neighbour along axis 0 (rows)
    dx = np.concatenate([dx, dx[-2:-1]], 0)
    radii = (dx[..., None] * F32(2) / np.sqrt(F32(12))).astype(F32)
    ones = np.ones_like(origins[..., :1])
    return Rays(origins, directions, viewdirs, radii, (F32(lossmult) * ones).astype(F32),
                (F32(near) * ones).astype(F32), (F32(far) * ones).astype(F32))



"""Seeded synthetic inputs of the Mip-NeRF hot path: lego-like rays and random-init /
"trained-like" MLP parameters.  numpy-only and deterministic across machines, so bench.py, the tests, the oracle and
scripts/make_golden.py (which feeds the same arrays to the reference) all see bit-identical inputs.
Neutral helper: not part of the product package and not part of the oracle."""
import collections

import numpy as np

F32 = np.float32

# datasets/datasets.py:13-16
Rays = collections.namedtuple(
    "Rays", ("origins", "directions", "viewdirs", "radii", "lossmult", "near", "far"))


def param_shapes(net_depth=8, net_width=256, net_depth_condition=1, net_width_condition=128,
                 skip_index=4, num_rgb=3, num_density=1, xyz_dim=96, view_dim=27):
    """state_dict keys/shapes of the reference MLP (models/mip_nerf.py:19-73), in
    registration order, without the `mlp.` prefix."""
    shapes = collections.OrderedDict()
    for i in range(net_depth):
        if i == 0:
            din = xyz_dim
        elif (i - 1) % skip_index == 0 and i > 1:
            din = net_width + xyz_dim
        else:
            din = net_width
        shapes[f"layers.{i}.0.weight"] = (net_width, din)
        shapes[f"layers.{i}.0.bias"] = (net_width,)
    shapes["density_layer.weight"] = (num_density, net_width)
    shapes["density_layer.bias"] = (num_density,)
    shapes["extra_layer.weight"] = (net_width, net_width)
    shapes["extra_layer.bias"] = (net_width,)
    for i in range(net_depth_condition):
        din = net_width + view_dim if i == 0 else net_width_condition
        shapes[f"view_layers.{i}.0.weight"] = (net_width_condition, din)
        shapes[f"view_layers.{i}.0.bias"] = (net_width_condition,)
    shapes["color_layer.weight"] = (num_rgb, net_width_condition)
    shapes["color_layer.bias"] = (num_rgb,)
    return shapes


def make_params(seed=0, density_gain=1.0, **arch):
    """Deterministic synthetic parameters (numpy Generator, stable across versions).
    Xavier-uniform-like weights, small non-zero biases.  `density_gain` > 1 scales the
    density head so sigma spans 0..tens ("trained-like")."""
    rng = np.random.default_rng(seed)
    params = collections.OrderedDict()
    for k, shp in param_shapes(**arch).items():
        if k.endswith("weight"):
            fan_out, fan_in = shp
            bound = np.sqrt(6.0 / (fan_in + fan_out))
            w = rng.uniform(-bound, bound, size=shp).astype(F32)
            if k == "density_layer.weight":
                w = (w * F32(density_gain)).astype(F32)
            params[k] = w
        else:
            params[k] = rng.uniform(-0.1, 0.1, size=shp).astype(F32)
    return params


def synthetic_rays(batch, seed=0, multiscale=False, unbounded=False):
    """Lego-like synthetic rays, SURVEY.md section 8(d): camera on a radius-4 sphere
    looking at the origin, focal 1111.11, pixel offsets U[-400,400], radii 5.2e-4,
    near 2 / far 6.  numpy Generator so the GPU box reproduces them bit-exactly."""
    rng = np.random.default_rng(seed)
    focal = 0.5 * 800 / np.tan(0.5 * 0.6911112)
    c = rng.standard_normal((batch, 3))
    c /= np.linalg.norm(c, axis=-1, keepdims=True)
    origins = 4.0 * c
    fwd = -c
    up = np.array([0.0, 0.0, 1.0])
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right, axis=-1, keepdims=True) + 1e-12
    upv = np.cross(right, fwd)
    px = rng.uniform(-400, 400, size=(batch, 1)) / focal
    py = rng.uniform(-400, 400, size=(batch, 1)) / focal
    directions = fwd + px * right + py * upv
    viewdirs = directions / np.linalg.norm(directions, axis=-1, keepdims=True)
    radii = np.full((batch, 1), 5.2e-4)
    lossmult = np.ones((batch, 1))
    near = np.full((batch, 1), 2.0)
    far = np.full((batch, 1), 6.0)
    if multiscale:
        j = rng.integers(0, 4, size=(batch, 1))
        radii = radii * (2.0 ** j)
        lossmult = 4.0 ** j
    if unbounded:
        near = rng.uniform(0.5, 1.5, size=(batch, 1))
        far = rng.uniform(4.0, 20.0, size=(batch, 1))
    return Rays(*[np.ascontiguousarray(a, dtype=F32) for a in
                  (origins, directions, viewdirs, radii, lossmult, near, far)])


def traj_target(rays):
    """Procedural ground-truth colour of a ray (smooth in origin / view direction; learnable in a few hundred steps)."""
    v = np.asarray(rays.viewdirs, np.float64)
    c = np.stack([0.5 + 0.4 * np.sin(2.3 * v[:, 1] + 0.5), 0.5 + 0.4 * np.cos(1.9 * v[:, 2] - 1.1 * v[:, 0]),
                  0.5 + 0.4 * np.sin(2.9 * v[:, 0] * v[:, 1] + 1.0)], -1)
    return c.astype(np.float32)

"""Tiny synthetic datasets in the reference's three on-disk formats (TEST INFRASTRUCTURE).

Written deterministically from a seed (PNG is lossless, JSON floats round-trip), so `scripts/make_golden.py --only-datasets`
(which runs the reference's dataset classes on them in the build container) and the tests (which run this repo's classes on
them, also on the GPU box) see byte-identical pixel values and poses without committing image files."""
import json
import os
import struct

import numpy as np


def _look_at_pose(rng, radius=4.0):
    """camera-to-world [4,4] on a sphere looking at the origin, OpenGL axes (x right, y up, z back)."""
    pos = rng.normal(size=3)
    pos[2] = abs(pos[2]) + 0.3
    pos = radius * pos / np.linalg.norm(pos)
    z = pos / np.linalg.norm(pos)
    x = np.cross([0.0, 0.0, 1.0], z)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = x, y, z, pos
    return m


def _png(path, arr):
    from PIL import Image
    os.makedirs(os.path.dirname(path), exist_ok=True)
    Image.fromarray(arr).save(path)


def write_blender(root, seed=0, counts=(("train", 3), ("val", 2), ("test", 2)), w=12, h=10):
    rng = np.random.RandomState(seed)
    for split, n in counts:
        frames = []
        for i in range(n):
            rel = f"./{split}/r_{i}"
            _png(os.path.join(root, rel + ".png"), rng.randint(0, 256, size=(h, w, 4)).astype(np.uint8))
            frames.append({"file_path": rel, "transform_matrix": _look_at_pose(rng).tolist()})
        with open(os.path.join(root, f"transforms_{split}.json"), "w") as f:
            json.dump({"camera_angle_x": 0.6911112070083618, "frames": frames}, f)
    return root


def write_multicam(root, seed=1, counts=(("train", 2), ("val", 1), ("test", 1)), w=16, h=12, scales=3):
    """metadata.json as convert_blender_data.py:84-117 writes it: every image at `scales` resolutions, lossmult = 4^j."""
    rng = np.random.RandomState(seed)
    meta = {}
    for split, n in counts:
        m = {k: [] for k in ("file_path", "cam2world", "width", "height", "focal", "label", "near", "far", "lossmult", "pix2cam")}
        for i in range(n):
            c2w = _look_at_pose(rng)
            focal0 = 0.5 * w / np.tan(0.5 * 0.69)
            for j in range(scales):
                wj, hj, fj = w // 2 ** j, h // 2 ** j, focal0 / 2 ** j
                rel = f"{split}/r_{i}_d{j}.png"
                _png(os.path.join(root, rel), rng.randint(0, 256, size=(hj, wj, 4)).astype(np.uint8))
                m["file_path"].append(rel)
                m["cam2world"].append(c2w.tolist())
                m["width"].append(wj)
                m["height"].append(hj)
                m["focal"].append(fj)
                m["label"].append(j)
                m["near"].append(2.0)
                m["far"].append(6.0)
                m["lossmult"].append(4.0 ** j)
                m["pix2cam"].append([[1.0 / fj, 0.0, -0.5 * wj / fj], [0.0, -1.0 / fj, 0.5 * hj / fj], [0.0, 0.0, -1.0]])
        meta[split] = m
    os.makedirs(root, exist_ok=True)
    with open(os.path.join(root, "metadata.json"), "w") as f:
        json.dump(meta, f)
    return root


def write_llff(root, seed=2, n=10, w=14, h=9, factor=4):
    """LLFF / mip-NeRF-360 layout: images_<factor>/*.png, poses_bounds.npy [n, 17], sparse/0/cameras.bin (one PINHOLE camera)."""
    rng = np.random.RandomState(seed)
    rows = []
    for i in range(n):
        _png(os.path.join(root, f"images_{factor}", f"img_{i:03d}.png"), rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8))
        c2w = _look_at_pose(rng, radius=3.0 + rng.rand())
        llff = np.concatenate([-c2w[:3, 1:2], c2w[:3, 0:1], c2w[:3, 2:3], c2w[:3, 3:4],      # (down, right, back) | t | hwf
                               np.array([[h * factor], [w * factor], [500.0]])], axis=1)
        near = 0.5 + rng.rand()
        rows.append(np.concatenate([llff.reshape(-1), [near, near + 4.0 + 10.0 * rng.rand()]]))
    os.makedirs(root, exist_ok=True)
    np.save(os.path.join(root, "poses_bounds.npy"), np.stack(rows))
    os.makedirs(os.path.join(root, "sparse", "0"), exist_ok=True)
    with open(os.path.join(root, "sparse", "0", "cameras.bin"), "wb") as f:
        f.write(struct.pack("<Q", 1))
        f.write(struct.pack("<iiQQ", 1, 1, w * factor, h * factor))
        f.write(struct.pack("<dddd", 500.0, 510.0, 0.5 * w * factor - 0.7, 0.5 * h * factor + 0.4))
    return root


# ---- a LEARNABLE procedural scene in the multi-scale Blender format (round 3: quality stand-in at realistic scale) ----------
SCENE_BLOBS = (   # centre (xyz), radius (xyz), peak density, colour
    ((0.00, 0.00, 0.05), (0.55, 0.40, 0.30), 38.0, (0.85, 0.25, 0.20)),
    ((0.45, -0.35, 0.30), (0.25, 0.25, 0.35), 30.0, (0.15, 0.60, 0.85)),
    ((-0.50, 0.30, -0.10), (0.30, 0.22, 0.22), 34.0, (0.20, 0.80, 0.30)),
    ((0.10, 0.55, 0.45), (0.18, 0.30, 0.18), 26.0, (0.95, 0.85, 0.15)),
    ((-0.25, -0.55, 0.50), (0.22, 0.18, 0.28), 28.0, (0.70, 0.30, 0.90)),
)


def _scene_density_colour(p):
    """p [..., 3] float64 -> density [...], colour [..., 3]: anisotropic Gaussian blobs, colour = density-weighted mix + a smooth
    position-dependent tint (so the network has view-independent but spatially varying colour to fit)."""
    sig = np.zeros(p.shape[:-1])
    col = np.zeros(p.shape)
    for c, r, a, rgb in SCENE_BLOBS:
        q = (p - np.asarray(c)) / np.asarray(r)
        g = a * np.exp(-0.5 * np.sum(q * q, -1))
        sig += g
        col += g[..., None] * np.asarray(rgb)
    col = col / np.maximum(sig[..., None], 1e-12)
    tint = 0.12 * np.stack([np.sin(3.1 * p[..., 0]), np.sin(2.7 * p[..., 1] + 0.5), np.sin(3.7 * p[..., 2] + 1.0)], -1)
    return sig, np.clip(col + tint, 0.0, 1.0)


def _render_scene(origins, directions, near, far, steps=192):
    """Quadrature of the volume-rendering integral along un-normalised rays o + t d, t in [near, far] (float64):
    returns straight colour [...,3] and alpha [...]."""
    o = np.asarray(origins, np.float64)[..., None, :]
    d = np.asarray(directions, np.float64)[..., None, :]
    t = np.linspace(near, far, steps + 1)
    tm = 0.5 * (t[1:] + t[:-1])
    dt = (t[1:] - t[:-1]) * np.linalg.norm(d, axis=-1)            # [..., steps]
    sig, col = _scene_density_colour(o + tm[:, None] * d)
    tau = sig * dt
    trans = np.exp(-(np.cumsum(tau, -1) - tau))
    w = (1.0 - np.exp(-tau)) * trans
    alpha = w.sum(-1)
    rgb = (w[..., None] * col).sum(-2) / np.maximum(alpha[..., None], 1e-9)
    return np.clip(rgb, 0.0, 1.0), np.clip(alpha, 0.0, 1.0)


def write_multicam_scene(root, seed=7, counts=(("train", 24), ("val", 2), ("test", 4)), base=64, scales=4):
    """Multi-scale Blender-format dataset (metadata.json as convert_blender_data.py:84-117 writes it) of the analytic scene above:
    every view rendered at base x base through the SAME pixel -> ray rule as Multicam._generate_rays (datasets.py:116-131), then
    box-averaged to base/2, base/4, ... like the converter's area down-sampling; RGBA PNGs (straight colour + alpha)."""
    from oracle import mipnerf_oracle as orc      # test infrastructure on both sides
    rng = np.random.RandomState(seed)
    meta = {}
    for split, n in counts:
        m = {k: [] for k in ("file_path", "cam2world", "width", "height", "focal", "label", "near", "far", "lossmult", "pix2cam")}
        for i in range(n):
            c2w = _look_at_pose(rng)
            focal0 = 0.5 * base / np.tan(0.5 * 0.69)
            p2c0 = [[1.0 / focal0, 0.0, -0.5 * base / focal0], [0.0, -1.0 / focal0, 0.5 * base / focal0], [0.0, 0.0, -1.0]]
            r = orc.generate_rays_multicam(c2w[:3], np.asarray(p2c0), base, base, 2.0, 6.0, 1.0)
            rgb, alpha = _render_scene(r.origins, r.directions, 2.0, 6.0)
            img = np.concatenate([rgb * alpha[..., None], alpha[..., None]], -1)      # premultiplied for the box filter
            for j in range(scales):
                wj, fj = base // 2 ** j, focal0 / 2 ** j
                if j:
                    img = 0.25 * (img[0::2, 0::2] + img[1::2, 0::2] + img[0::2, 1::2] + img[1::2, 1::2])
                a = img[..., 3:]
                straight = np.where(a > 1e-6, img[..., :3] / np.maximum(a, 1e-6), 1.0)
                rel = f"{split}/r_{i}_d{j}.png"
                _png(os.path.join(root, rel), np.round(255.0 * np.clip(np.concatenate([straight, a], -1), 0, 1)).astype(np.uint8))
                m["file_path"].append(rel)
                m["cam2world"].append(c2w.tolist())
                m["width"].append(wj)
                m["height"].append(wj)
                m["focal"].append(fj)
                m["label"].append(j)
                m["near"].append(2.0)
                m["far"].append(6.0)
                m["lossmult"].append(4.0 ** j)
                m["pix2cam"].append([[1.0 / fj, 0.0, -0.5 * wj / fj], [0.0, -1.0 / fj, 0.5 * wj / fj], [0.0, 0.0, -1.0]])
        meta[split] = m
    os.makedirs(root, exist_ok=True)
    with open(os.path.join(root, "metadata.json"), "w") as f:
        json.dump(meta, f)
    return root


QUALITY = dict(batch=1024, num_samples=128, steps=500, lr_init=2e-3, lr_final=2e-5, max_steps=500, lr_delay_steps=50,
               lr_delay_mult=0.01, id_seed=20240, param_seed=3)


def quality_batch_ids(n_pixels, steps, batch, seed):
    """The SAME pixel ids for the reference's loop and the native loop: step k draws `batch` ids without replacement inside an
    epoch-wise permutation (what a shuffled DataLoader over the flattened rays does), from numpy's seeded generator."""
    rng = np.random.default_rng(seed)
    ids, perm, pos = [], rng.permutation(n_pixels), 0
    for _ in range(steps):
        if pos + batch > n_pixels:
            perm, pos = rng.permutation(n_pixels), 0
        ids.append(perm[pos:pos + batch].copy())
        pos += batch
    return np.stack(ids)


# ---- an UNBOUNDED procedural scene (round 5: a trained field for the parity tests of `MipNerf(unbounded=True)`) ---------------
# The blobs above around the origin + a far "sky" shell of radius 8 whose density is switched on and off smoothly with the
# direction, so a ray through the scene ends on a blob (opaque, near), on the shell (opaque or soft, far: contracted space),
# or leaves through a hole of the shell (empty).  Cameras sit INSIDE the shell and look at the origin, LLFF-style per-view bounds.
SCENE360 = dict(shell_radius=8.0, shell_sigma=0.5, shell_density=6.0, views=28, res=44, fov=1.6, radius=(2.6, 3.4), near=(0.6, 1.1), far=(16.0, 22.0), seed=77)


def _scene360_density_colour(p):
    sig, col = _scene_density_colour(p)
    r = np.linalg.norm(p, axis=-1)
    u = p / np.maximum(r[..., None], 1e-9)
    window = 0.5 + 0.5 * np.tanh(10.0 * (np.sin(3.0 * np.arctan2(u[..., 1], u[..., 0])) * np.cos(2.5 * u[..., 2]) - 0.1))     # ~35 % of the sky is a hole
    s_shell = SCENE360["shell_density"] * window * np.exp(-0.5 * ((r - SCENE360["shell_radius"]) / SCENE360["shell_sigma"]) ** 2)
    c_shell = np.stack([0.55 + 0.35 * np.sin(2.0 * u[..., 0] + 4.0 * u[..., 1]), 0.50 + 0.35 * np.cos(3.0 * u[..., 2] - u[..., 0]),
                        0.60 + 0.30 * np.sin(5.0 * u[..., 1] * u[..., 2] + 0.7)], -1)
    tot = sig + s_shell
    colour = (sig[..., None] * col + s_shell[..., None] * c_shell) / np.maximum(tot[..., None], 1e-12)
    return tot, np.clip(colour, 0.0, 1.0)


def render_scene360(origins, directions, near, far, steps=1024, white_bkgd=True):
    """float64 quadrature of the volume-rendering integral, fence posts uniform in INVERSE depth between per-ray near and far
    (fine near the camera, coarse far away, like the model's own parameterisation); returns the composited colour [...,3]."""
    o = np.asarray(origins, np.float64)[..., None, :]
    d = np.asarray(directions, np.float64)[..., None, :]
    s = np.linspace(0.0, 1.0, steps + 1)
    t = 1.0 / (s / np.asarray(far, np.float64) + (1.0 - s) / np.asarray(near, np.float64))       # [..., steps + 1]
    tm = 0.5 * (t[..., 1:] + t[..., :-1])
    dt = (t[..., 1:] - t[..., :-1]) * np.linalg.norm(d, axis=-1)
    sig, col = _scene360_density_colour(o + tm[..., None] * d)
    tau = sig * dt
    trans = np.exp(-(np.cumsum(tau, -1) - tau))
    w = (1.0 - np.exp(-tau)) * trans
    rgb = (w[..., None] * col).sum(-2)
    if white_bkgd:
        rgb = rgb + (1.0 - w.sum(-1))[..., None]
    return np.clip(rgb, 0.0, 1.0)


def scene360_rays():
    """Every training ray of the unbounded scene + its ground-truth colour: (Rays of [n, k] float32 arrays, rgb [n, 3] float32).
    Pixel -> ray rule of Multicam._generate_rays (datasets.py:116-131) through the oracle's restatement; per-view near / far."""
    from oracle import mipnerf_oracle as orc
    S = SCENE360
    rng = np.random.RandomState(S["seed"])
    rays, rgbs = [], []
    res = S["res"]
    focal = 0.5 * res / np.tan(0.5 * S["fov"])       # wide: ~24 % of the rays leave through a hole of the shell, ~47 % end opaque
    p2c = np.asarray([[1.0 / focal, 0.0, -0.5 * res / focal], [0.0, -1.0 / focal, 0.5 * res / focal], [0.0, 0.0, -1.0]])
    for _ in range(S["views"]):
        c2w = _look_at_pose(rng, radius=rng.uniform(*S["radius"]))
        near, far = rng.uniform(*S["near"]), rng.uniform(*S["far"])
        r = orc.generate_rays_multicam(c2w[:3], p2c, res, res, near, far, 1.0)
        rays.append([np.asarray(a, np.float32).reshape(res * res, -1) for a in r])
        rgbs.append(render_scene360(r.origins, r.directions, r.near, r.far).reshape(res * res, 3))
    R = type(r)(*[np.concatenate([v[i] for v in rays], 0) for i in range(len(rays[0]))])
    return R, np.concatenate(rgbs, 0).astype(np.float32)

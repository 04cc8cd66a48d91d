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

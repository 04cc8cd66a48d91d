"""Device-resident datasets: the reference's on-disk formats feeding the device-side ray generator.

The reference (datasets/datasets.py) expands every pixel of every image into a 52-byte `Rays` row on the host at start-up
(64 M rays = 3.3 GB for lego) and pays a host gather + H2D copy per batch.  Here a dataset is: the images as one flat
`[P, 3]` fp32 device tensor, a camera table `[n_images, 32]` (`ops.camera_record`), and pixel offsets; a batch is drawn as
pixel ids ON the device and its rays come out of `k_generate_rays` (`ops.generate_rays`) -- nothing per-ray ever lives on
the host.  Same class names, constructor arguments, file formats, `__len__` / `__getitem__` meaning and `dataset_dict`
keys as the reference, so `MipNeRFSystem.setup` / `eval.py` read the same directories:

    Blender      `transforms_{split}.json` + RGBA PNGs, white-background compositing        datasets.py:171-263
    Multicam     `metadata.json` of convert_blender_data.py (pix2cam / cam2world / lossmult)  datasets.py:84-168
    RealData360  LLFF `poses_bounds.npy` + `images[_f]/` + COLMAP `sparse/0/cameras.bin`    datasets.py:266-474
                 (upstream never registers it; here it is `dataset_dict['llff']`)
    RenderGen    the spherical render path of render_video.py:19-118 as a camera table

File parsing and pose algebra are host numpy (they run once); the per-ray arithmetic is the HIP kernel's.
"""
import json
import os
import struct

import numpy as np
import torch

from . import ops
from .rays import Rays, Rays_keys


# ---------------------------------------------------------------------------------------------------------------------
# host side: files -> (images, camera records)
# ---------------------------------------------------------------------------------------------------------------------
def _read_image(fname):
    from PIL import Image
    with open(fname, "rb") as f:
        return np.array(Image.open(f), dtype=np.float32) / 255.0


def _composite(image, white_bkgd):
    """datasets.py:108-110 / 201-203: RGBA over white when `white_bkgd`, then drop alpha."""
    if white_bkgd:
        image = image[..., :3] * image[..., -1:] + (1.0 - image[..., -1:])
    return np.ascontiguousarray(image[..., :3], dtype=np.float32)


def _halve(image):
    """`cv2.resize(image, (w//2, h//2), interpolation=cv2.INTER_AREA)` of datasets.py:193-196 for an exact factor 2: the mean of
    each 2x2 block (what INTER_AREA computes when the scale is integral)."""
    h2, w2 = image.shape[0] // 2, image.shape[1] // 2
    v = image[:h2 * 2, :w2 * 2].reshape(h2, 2, w2, 2, -1)
    return v.mean(axis=(1, 3), dtype=np.float32)


def load_blender(data_dir, split, white_bkgd=True, factor=0):
    """datasets.py:183-212.  Returns (images, records): `transform_matrix` is camera-to-world, focal from `camera_angle_x`."""
    with open(os.path.join(data_dir, f"transforms_{split}.json")) as fp:
        meta = json.load(fp)
    images, c2ws = [], []
    for frame in meta["frames"]:
        image = _read_image(os.path.join(data_dir, frame["file_path"] + ".png"))
        if factor == 2:
            image = _halve(image)
        elif factor > 0:
            raise ValueError(f"Blender dataset only supports factor=0 or 2, {factor} set.")
        images.append(_composite(image, white_bkgd))
        c2ws.append(np.array(frame["transform_matrix"], dtype=np.float32))
    h, w = images[0].shape[:2]
    focal = 0.5 * w / np.tan(0.5 * float(meta["camera_angle_x"]))
    records = [ops.camera_record(c2w, w, h, 2.0, 6.0, focal=focal) for c2w in c2ws]
    return images, records, dict(h=h, w=w, focal=focal, camtoworlds=c2ws)


def load_multicam(data_dir, split, white_bkgd=True):
    """datasets.py:98-113 + the per-image attributes `_generate_rays` broadcasts (:116-168)."""
    with open(os.path.join(data_dir, "metadata.json")) as fp:
        meta = json.load(fp)[split]
    meta = {k: np.array(meta[k]) for k in meta}
    images = [_composite(_read_image(os.path.join(data_dir, rel)), white_bkgd) for rel in meta["file_path"]]
    records = []
    for i in range(len(images)):
        records.append(ops.camera_record(meta["cam2world"][i].astype(np.float32), float(meta["width"][i]), float(meta["height"][i]),
                                         float(meta["near"][i]), float(meta["far"][i]),
                                         pix2cam=meta["pix2cam"][i].astype(np.float32), lossmult=float(meta["lossmult"][i])))
    return images, records, dict(meta=meta)


def _unit(v):
    return v / np.linalg.norm(v)


def _look_at(z, up, pos):
    """[x | y | z | pos] with z along `z`, x = up x z, y = z x x (datasets.py:432-439)."""
    z = _unit(z)
    x = _unit(np.cross(up, z))
    y = _unit(np.cross(z, x))
    return np.stack([x, y, z, pos], axis=1)


def _to44(p34):
    bottom = np.broadcast_to(np.array([0.0, 0.0, 0.0, 1.0], dtype=p34.dtype), p34.shape[:-2] + (1, 4))
    return np.concatenate([p34, bottom], axis=-2)


def recenter_poses(poses):
    """datasets.py:379-390: express every pose in the frame of the average pose (mean position, summed z and y axes)."""
    avg = _look_at(poses[:, :3, 2].sum(0), poses[:, :3, 1].sum(0), poses[:, :3, 3].mean(0))
    out = poses.copy()
    out[:, :3, :4] = (np.linalg.inv(_to44(avg.astype(np.float64))) @ _to44(poses[:, :3, :4].astype(np.float64)))[:, :3, :4]
    return out


def spherify_poses(poses):
    """datasets.py:445-474: move the origin to the point closest to all optical axes and make `up` = mean offset of the camera
    centres from it (the third axis of the new frame), with the fixed helper vector (.1, .2, .3) fixing the in-plane rotation."""
    d = poses[:, :3, 2:3]
    o = poses[:, :3, 3:4]
    a = np.eye(3) - d * np.transpose(d, [0, 2, 1])                   # projector orthogonal to each axis
    b = -a @ o
    center = np.squeeze(-np.linalg.inv((np.transpose(a, [0, 2, 1]) @ a).mean(0)) @ b.mean(0))
    v0 = _unit((poses[:, :3, 3] - center).mean(0))
    v1 = _unit(np.cross([0.1, 0.2, 0.3], v0))
    v2 = _unit(np.cross(v0, v1))
    frame = np.stack([v1, v2, v0, center], axis=1)
    reset = np.linalg.inv(_to44(frame[None])) @ _to44(poses[:, :3, :4])
    hwf = np.broadcast_to(poses[0, :3, -1:], reset[:, :3, -1:].shape)
    return np.concatenate([reset[:, :3, :4], hwf], axis=-1)


def read_colmap_pinhole(fname):
    """First camera of a COLMAP `cameras.bin` as K = [[fx,0,cx],[0,fy,cy],[0,0,1]] (datasets.py:392-413 reads exactly this much:
    count, (camera_id, model_id, width, height), four doubles)."""
    with open(fname, "rb") as f:
        struct.unpack("<Q", f.read(8))
        struct.unpack("<iiQQ", f.read(24))
        fx, fy, cx, cy = struct.unpack("<dddd", f.read(32))
    return np.array([[fx, 0.0, cx], [0.0, fy, cy], [0.0, 0.0, 1.0]])


def load_realdata360(data_dir, split, white_bkgd=True, factor=0):
    """datasets.py:278-345.  `factor` must be > 0 (upstream divides by it at :312 and :335; with the shipped default 0 it
    produces infinities -- raised here instead).  Every 8th image is the test split."""
    if factor <= 0:
        raise ValueError("RealData360 needs factor > 0 (images_<factor>/; datasets.py:312 divides by it)")
    imgdir = os.path.join(data_dir, f"images_{factor}")
    if not os.path.exists(imgdir):
        raise ValueError(f"Image folder {imgdir} does not exist.")
    files = [os.path.join(imgdir, f) for f in sorted(os.listdir(imgdir)) if f.endswith(("JPG", "jpg", "png"))]
    images = np.stack([_read_image(f) for f in files], axis=0)
    arr = np.load(os.path.join(data_dir, "poses_bounds.npy"))
    if arr.shape[0] != images.shape[0]:
        raise RuntimeError(f"Mismatch between imgs {images.shape[0]} and poses {arr.shape[0]}")
    poses = arr[:, :-2].reshape(-1, 3, 5).copy()                     # [n, 3, 5] = [R | t | (h, w, f)] in LLFF axis order
    bds = arr[:, -2:].astype(np.float32)
    poses[:, 0, 4], poses[:, 1, 4] = images.shape[1], images.shape[2]
    poses[:, 2, 4] /= factor
    poses = np.concatenate([poses[:, :, 1:2], -poses[:, :, 0:1], poses[:, :, 2:]], axis=2).astype(np.float32)   # (down,right,back) -> (right,up,back)
    poses = spherify_poses(recenter_poses(poses))
    idx = np.arange(images.shape[0])
    test = idx[::8]
    sel = np.array([i for i in idx if i not in test]) if split == "train" else test
    K = read_colmap_pinhole(os.path.join(data_dir, "sparse", "0", "cameras.bin"))
    K[:2, :] /= factor
    K_inv = np.linalg.inv(K)
    K_inv[1:, :] *= -1
    h, w = images.shape[1:3]
    records = [ops.camera_record(poses[i, :3, :4], w, h, float(bds[i, 0]), float(bds[i, 1]), pix2cam=K_inv) for i in sel]
    imgs = [np.ascontiguousarray(images[i, ..., :3], dtype=np.float32) for i in sel]
    return imgs, records, dict(h=h, w=w, K=K, K_inv=K_inv, bds=bds[sel], camtoworlds=poses[sel][:, :3, :4], focal=poses[0, -1, -1])


def create_spheric_poses(radius, n_poses=120):
    """utils/vis.py:159-198: `n_poses` camera-to-world matrices on a circle around the z axis, looking 36 degrees down."""
    phi = -np.pi / 5
    trans = np.eye(4)
    trans[2, 3] = radius
    rot_phi = np.array([[1, 0, 0, 0], [0, np.cos(phi), -np.sin(phi), 0], [0, np.sin(phi), np.cos(phi), 0], [0, 0, 0, 1]])
    swap = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]])
    out = []
    for th in np.linspace(0, 2 * np.pi, n_poses + 1)[:-1]:
        rot_th = np.array([[np.cos(th), 0, -np.sin(th), 0], [0, 1, 0, 0], [np.sin(th), 0, np.cos(th), 0], [0, 0, 0, 1]])
        out.append((swap @ (rot_th @ rot_phi @ trans))[:3])
    return np.stack(out, 0)


# ---------------------------------------------------------------------------------------------------------------------
# device side
# ---------------------------------------------------------------------------------------------------------------------
def _default_device():
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None


class BaseDataset(torch.utils.data.Dataset):
    """datasets.py:25-81 with the rays left implicit.  `split == 'train'`: `len` = number of pixels of all images
    (batch_type 'all_images'), item i = (ray i, pixel i); otherwise `len` = number of images and an item is a whole image
    `(Rays [H,W,k], image [H,W,3])` -- like upstream, the val split ignores the index and walks the images in order."""

    def __init__(self, data_dir, split, white_bkgd=True, batch_type="all_images", factor=0, device=None):
        super().__init__()
        self.near, self.far = 2, 6
        self.split, self.data_dir, self.white_bkgd, self.batch_type, self.factor = split, data_dir, white_bkgd, batch_type, factor
        self.it = -1
        if split == "train":
            assert batch_type == "all_images", "The batch_type can only be all_images with flatten"
        else:
            assert batch_type == "single_image", "The batch_type can only be single_image without flatten"
        self.images, records, info = self._load()
        for k, v in info.items():
            setattr(self, k, v)
        self.n_examples = len(self.images)
        self.cameras = torch.stack(records)                                    # [n, 32] host copy of the camera table
        self.sizes = [(im.shape[0], im.shape[1]) for im in self.images]
        self.offsets = np.concatenate([[0], np.cumsum([h * w for h, w in self.sizes])]).astype(np.int64)
        self.device = None
        self._dev = None
        dev = device if device is not None else _default_device()
        if dev is not None:
            self.to(dev)

    def _load(self):
        raise ValueError("Implement in different dataset.")

    # -- device residency
    def to(self, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("datasets: rays are generated by the HIP kernel; need a HIP device")
        flat = np.concatenate([im.reshape(-1, 3) for im in self.images], axis=0)
        self._dev = dict(cameras=self.cameras.to(device), pixels=torch.from_numpy(flat).to(device),
                         offsets=torch.from_numpy(self.offsets).to(device))
        self.device = device
        return self

    def _need_device(self):
        if self._dev is None:
            raise RuntimeError("datasets: no HIP device (construct with device=... or call .to(device)); rays are computed by "
                               "k_generate_rays, there is no host fallback")
        return self._dev

    @property
    def num_pixels(self):
        return int(self.offsets[-1])

    def rays_at(self, ids):
        """(Rays [n,k], pixels [n,3]) of global pixel ids `ids` (int64 device tensor over the concatenation of all images)."""
        d = self._need_device()
        ids = ids.to(d["offsets"].device, torch.int64).reshape(-1).contiguous()
        cam = torch.searchsorted(d["offsets"], ids, right=True) - 1
        pix = ids - d["offsets"][cam]
        rays = ops.generate_rays(d["cameras"], cam_idx=cam, pix_idx=pix)
        return rays, d["pixels"][ids]

    def image_rays(self, i):
        """(Rays [H,W,k], image [H,W,3]) of image i."""
        d = self._need_device()
        h, w = self.sizes[i]
        cam = torch.full((h * w,), i, dtype=torch.int32, device=self.device)
        rays = ops.generate_rays(d["cameras"], cam_idx=cam, pix_idx=torch.arange(h * w, dtype=torch.int32, device=self.device))
        lo, hi = int(self.offsets[i]), int(self.offsets[i + 1])
        return Rays(*[t.reshape(h, w, -1) for t in rays]), d["pixels"][lo:hi].reshape(h, w, 3)

    def sample(self, batch_size, generator=None):
        """One training batch drawn uniformly over all pixels, entirely on the device."""
        d = self._need_device()
        ids = torch.randint(0, self.num_pixels, (batch_size,), device=self.device, generator=generator)
        return self.rays_at(ids)

    # -- torch Dataset protocol (datasets.py:72-81)
    def __len__(self):
        return self.num_pixels if self.split == "train" else self.n_examples

    def __getitem__(self, index):
        if self.split == "train":
            scalar = not torch.is_tensor(index) and np.ndim(index) == 0
            ids = torch.as_tensor(index, dtype=torch.int64).reshape(-1)
            rays, pix = self.rays_at(ids.to(self.device) if self.device is not None else ids)
            return (Rays(*[t[0] for t in rays]), pix[0]) if scalar else (rays, pix)
        if self.split == "val":
            index = (self.it + 1) % self.n_examples
            self.it += 1
        return self.image_rays(int(index))


class Blender(BaseDataset):
    """Blender Dataset (datasets.py:171-263)."""

    def __init__(self, data_dir, split="train", white_bkgd=True, batch_type="all_images", factor=0, device=None):
        super().__init__(data_dir, split, white_bkgd, batch_type, factor, device)

    def _load(self):
        return load_blender(self.data_dir, self.split, self.white_bkgd, self.factor)


class Multicam(BaseDataset):
    """Multicam (multi-scale Blender) Dataset (datasets.py:84-168)."""

    def __init__(self, data_dir, split="train", white_bkgd=True, batch_type="all_images", device=None):
        super().__init__(data_dir, split, white_bkgd, batch_type, 0, device)

    def _load(self):
        return load_multicam(self.data_dir, self.split, self.white_bkgd)


class RealData360(BaseDataset):
    """RealData360 / LLFF Dataset (datasets.py:266-474); per-image near / far from `poses_bounds.npy`."""

    def __init__(self, data_dir, split="train", white_bkgd=True, batch_type="all_images", factor=4, device=None):
        super().__init__(data_dir, split, white_bkgd, batch_type, factor, device)

    def _load(self):
        return load_realdata360(self.data_dir, self.split, self.white_bkgd, self.factor)


class RenderGen(torch.utils.data.Dataset):
    """render_video.py:19-118: `scales` pyramids of the 120-pose spherical path; item i = Rays [H_i, W_i, k] of camera i."""

    def __init__(self, base_focal, base_size, scales=4, device=None):
        super().__init__()
        self.near, self.far = 2, 6
        c2w = create_spheric_poses(4)
        records, self.sizes = [], []
        for i in range(scales):
            w, h, f = base_size[0] / 2 ** i, base_size[1] / 2 ** i, base_focal / 2 ** i
            pix2cam = np.array([[1.0 / f, 0.0, -0.5 * w / f], [0.0, -1.0 / f, 0.5 * h / f], [0.0, 0.0, -1.0]])
            for m in c2w:
                # float64 table -> float64 ray arithmetic on the device, as the reference's numpy (its poses and pix2cam are float64)
                records.append(ops.camera_record(np.asarray(m, np.float64), w, h, self.near, self.far, pix2cam=pix2cam, dtype=torch.float64))
                self.sizes.append((int(h), int(w)))
        self.n_sample = len(records)
        self.cameras = torch.stack(records)
        dev = device if device is not None else _default_device()
        self.device = torch.device(dev) if dev is not None else None
        self._cams_dev = self.cameras.to(self.device) if self.device is not None else None

    def __len__(self):
        return self.n_sample

    def __getitem__(self, index):
        if self._cams_dev is None:
            raise RuntimeError("RenderGen: no HIP device; rays are computed by k_generate_rays")
        h, w = self.sizes[index]
        cam = torch.full((h * w,), int(index), dtype=torch.int32, device=self.device)
        rays = ops.generate_rays(self._cams_dev, cam_idx=cam, pix_idx=torch.arange(h * w, dtype=torch.int32, device=self.device))
        return Rays(*[t.reshape(h, w, -1) for t in rays])


class RayLoader:
    """What `DataLoader(dataset, shuffle, batch_size)` yields in nerf_system.py:78-93, without the host: a fresh device
    permutation of the pixel ids per epoch (train) or one image per item with the leading batch dimension of 1 (val / test).
    With world_size > 1 (default: the initialised torch.distributed group) every rank draws the SAME permutation (same seed,
    same epoch), PADS it to a multiple of world_size by wrapping around to its own start, and keeps every world_size-th id
    starting at its rank -- what Lightning's DistributedSampler (drop_last=False) does for the reference under
    train.py:56-60 -- so every rank sees the same number of rays and of batches per epoch (a rank with one batch more would
    wait forever in the gradient all-reduce) and the global batch is batch_size x world_size rays, disjoint except for the
    < world_size wrapped ids of the last batch."""

    def __init__(self, dataset, batch_size=1, shuffle=False, drop_last=False, seed=0, rank=None, world_size=None):
        self.dataset, self.batch_size, self.shuffle, self.drop_last = dataset, int(batch_size), shuffle, drop_last
        self.seed, self.epoch = int(seed), 0
        if world_size is None:
            import torch.distributed as dist
            on = dist.is_available() and dist.is_initialized()
            world_size, rank = (dist.get_world_size(), dist.get_rank()) if on else (1, 0)
        self.rank, self.world_size = int(rank or 0), int(world_size)

    def _local_count(self):
        n = len(self.dataset)
        if self.dataset.split != "train":
            return n
        return (n + self.world_size - 1) // self.world_size          # identical on every rank (padded permutation)

    def __len__(self):
        n = self._local_count()
        if self.dataset.split != "train":
            return n
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def __iter__(self):
        ds = self.dataset
        if ds.split != "train":
            for i in range(len(ds)):
                rays, image = ds[i]
                yield Rays(*[t[None] for t in rays]), image[None]
            return
        n = len(ds)
        if self.shuffle:
            g = torch.Generator(device=ds.device)
            g.manual_seed(self.seed + self.epoch)
            order = torch.randperm(n, device=ds.device, generator=g)
        else:
            order = torch.arange(n, device=ds.device)
        self.epoch += 1
        pad = self._local_count() * self.world_size - n
        if pad:      # DistributedSampler's rule: wrap around, repeating the list when the pad exceeds it (n < world_size - 1)
            order = order.repeat(pad // max(n, 1) + 2)[:n + pad]
        order = order[self.rank::self.world_size]
        for b in range(len(self)):
            yield ds.rays_at(order[b * self.batch_size:(b + 1) * self.batch_size])


dataset_dict = {
    "blender": Blender,             # datasets/__init__.py:2-4
    "multi_blender": Multicam,
    "llff": RealData360,            # registered here, dead upstream
    "realdata360": RealData360,
}

__all__ = ["Blender", "Multicam", "RealData360", "RenderGen", "RayLoader", "dataset_dict", "Rays", "Rays_keys",
           "create_spheric_poses", "load_blender", "load_multicam", "load_realdata360"]

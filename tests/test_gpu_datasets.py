"""mipnerf_pl_amd.datasets on the device: every ray of every image of the three on-disk formats (synthetic fixtures) equals what
the reference's own dataset classes produced for the same files (tests/golden/datasets_tiny.npz), the loader walks every pixel
exactly once per epoch, and MipNeRFSystem.setup / dataloaders / training_step / validation render run off a dataset directory."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import dataset_fixture as fx  # noqa: E402
from gpu_util import record  # noqa: E402

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(HERE, "golden", "datasets_tiny.npz"))
DEV = "cuda:0"


@pytest.fixture(scope="module")
def roots(tmp_path_factory):
    t = tmp_path_factory.mktemp("ds")
    return dict(blender=fx.write_blender(str(t / "blender")), multicam=fx.write_multicam(str(t / "multicam")),
                llff=fx.write_llff(str(t / "llff")))


def cmp_rays(rays, tag, tol, what):
    worst = 0.0
    for k in rays._fields:
        got = getattr(rays, k).double().cpu().numpy()
        ref = G[f"{tag}_{k}"].reshape(got.shape)
        err = float(np.max(np.abs(got - ref) / (1.0 + np.abs(ref))))
        worst = max(worst, err)
        assert err <= tol, f"{what} {tag} {k}: {err}"
    record(f"{what}/{tag}", max_rel_err=worst, tol=tol)


@pytest.mark.parametrize("name,cls_name,tag,kw,tol", [
    ("blender", "Blender", "blender_train", dict(), 3e-6),
    ("blender", "Blender", "blender_train_black", dict(white_bkgd=False), 3e-6),
    ("multicam", "Multicam", "multicam_train", dict(), 3e-6),
    ("llff", "RealData360", "llff_train", dict(factor=4), 1e-5),
])
def test_train_split_every_ray(roots, name, cls_name, tag, kw, tol):
    from mipnerf_pl_amd import datasets as D
    ds = getattr(D, cls_name)(roots[name], "train", batch_type="all_images", device=DEV, **kw)
    n = len(ds)
    rays, pix = ds[torch.arange(n)]
    assert rays.origins.is_cuda and rays.origins.shape == (n, 3) and pix.shape == (n, 3)
    cmp_rays(rays, tag, tol, "dataset_train_rays")
    np.testing.assert_allclose(pix.cpu().numpy(), G[tag + "_images"], rtol=0, atol=1e-7)
    r1, p1 = ds[n - 1]                                              # scalar index: one row, like the reference's __getitem__
    assert r1.origins.shape == (3,) and torch.equal(p1, pix[-1]) and torch.equal(r1.radii, rays.radii[-1])
    # one epoch of the device loader = a permutation of the pixels
    seen = torch.zeros(n, dtype=torch.int32, device=DEV)
    loader = D.RayLoader(ds, batch_size=37, shuffle=True, seed=3)
    assert len(loader) == (n + 36) // 37
    inner = ds.rays_at

    def spy(ids):
        seen[ids] += 1
        return inner(ids)
    ds.rays_at = spy
    for rb, pb in loader:
        assert rb.origins.shape[0] == pb.shape[0] <= 37
    ds.rays_at = inner
    assert int(seen.min()) == 1 and int(seen.max()) == 1
    # two ranks of a data-parallel job: the same permutation, disjoint halves (DistributedSampler semantics)
    seen.zero_()
    ds.rays_at = spy
    for r in range(2):
        ld = D.RayLoader(ds, batch_size=37, shuffle=True, seed=3, rank=r, world_size=2)
        assert len(ld) == (((n + 1) // 2) + 36) // 37              # identical on both ranks (padded permutation)
        for rb, pb in ld:
            pass
    ds.rays_at = inner
    assert int(seen.min()) == 1 and int(seen.sum()) == 2 * ((n + 1) // 2) and int(seen.max()) <= 2
    rs, ps = ds.sample(64)
    assert rs.viewdirs.shape == (64, 3) and ps.shape == (64, 3) and bool(torch.isfinite(rs.radii).all())


@pytest.mark.parametrize("name,cls_name,split,tag,kw,tol", [
    ("blender", "Blender", "val", "blender_val", dict(), 3e-6),
    ("multicam", "Multicam", "test", "multicam_test", dict(), 3e-6),
    ("llff", "RealData360", "test", "llff_test", dict(factor=4), 1e-5),
])
def test_image_splits(roots, name, cls_name, split, tag, kw, tol):
    from mipnerf_pl_amd import datasets as D
    ds = getattr(D, cls_name)(roots[name], split, batch_type="single_image", device=DEV, **kw)
    assert len(ds) == int(G[tag + "_n"])
    for i in range(len(ds)):
        rays, img = ds[0 if split == "val" else i]       # the val split ignores the index and walks the images (datasets.py:76-78)
        h, w = ds.sizes[i]
        assert rays.origins.shape == (h, w, 3) and img.shape == (h, w, 3)
        cmp_rays(rays, f"{tag}{i}", tol, "dataset_image_rays")
        np.testing.assert_allclose(img.cpu().numpy(), G[f"{tag}_image{i}"], rtol=0, atol=1e-7)
    items = list(D.RayLoader(ds, batch_size=1))
    assert len(items) == len(ds) and items[0][0].origins.shape[0] == 1 and items[0][1].dim() == 4


def test_render_path_rays():
    from mipnerf_pl_amd import datasets as D
    rg = D.RenderGen(float(G["render_focal"]), [24, 20], 2, device=DEV)
    assert len(rg) == 240
    for i in (0, 7, 119, 120, 239):
        cmp_rays(rg[i], f"render{i}", 4e-6, "render_path_rays")


def test_system_runs_off_a_dataset_directory(roots):
    """setup -> train_dataloader -> training_step -> val_dataloader -> validation render, as Lightning drives them
    (nerf_system.py:56-149), with the rays coming from the device-side generator."""
    from mipnerf_pl_amd.system import DEFAULT_HPARAMS, MipNeRFSystem
    hp = dict(DEFAULT_HPARAMS)
    hp.update({"dataset_name": "blender", "data_path": roots["blender"], "train.batch_size": 48, "train.batch_type": "all_images",
               "val.batch_type": "single_image", "val.chunk_size": 64, "nerf.num_samples": 32})
    torch.manual_seed(0)
    system = MipNeRFSystem(hp).to(DEV)
    system.setup("fit")
    assert system.train_dataset.device.type == "cuda" and len(system.train_dataset) == 3 * 12 * 10
    (opt,), _ = system.configure_optimizers()
    it = iter(system.train_dataloader())
    losses = []
    for step in range(3):
        batch = next(it)
        loss = system.training_step(batch, step)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses))
    vb = next(iter(system.val_dataloader()))
    out = system.validation_step(vb, 0)
    assert np.isfinite(float(out["val/psnr"])) and np.isfinite(float(out["val/loss"]))
    record("system_off_dataset_dir", loss_first=losses[0], loss_last=losses[-1], val_psnr=float(out["val/psnr"]))


def test_unbounded_system_runs_off_an_llff_directory(roots):
    """SURVEY 8(f)-4 end to end: the registered `RealData360` dataset (LLFF / mip-NeRF-360 files, per-image near / far from
    poses_bounds.npy) feeding `MipNerf(unbounded=True)` through the system hooks -- setup, device-side batches, fp32 training
    steps through autograd, a validation render."""
    from mipnerf_pl_amd.system import DEFAULT_HPARAMS, MipNeRFSystem
    hp = dict(DEFAULT_HPARAMS)
    hp.update({"dataset_name": "llff", "data_path": roots["llff"], "train.batch_size": 40, "train.batch_type": "all_images",
               "val.batch_type": "single_image", "val.chunk_size": 64, "nerf.num_samples": 32, "nerf.unbounded": True,
               "optimizer.lr_init": 1e-3, "optimizer.lr_delay_steps": 0})
    torch.manual_seed(0)
    system = MipNeRFSystem(hp, precision="fp32").to(DEV)
    assert system.mip_nerf.unbounded and tuple(system.mip_nerf.mlp.layers[0][0].weight.shape) == (256, 672)
    system.setup("fit")
    (opt,), _ = system.configure_optimizers()
    it = iter(system.train_dataloader())
    batch = next(it)
    losses = []
    for step in range(6):                       # the same batch: the loss must fall
        loss = system.training_step(batch, step)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    vb = next(iter(system.val_dataloader()))
    out = system.validation_step(vb, 0)
    assert np.isfinite(float(out["val/psnr"]))
    record("unbounded_system_off_llff_dir", loss_first=losses[0], loss_last=losses[-1], val_psnr=float(out["val/psnr"]))

"""Host side of mipnerf_pl_amd.datasets on the synthetic datasets of tests/dataset_fixture.py, against goldens produced by the
reference's own Blender / Multicam / RealData360 / create_spheric_poses (scripts/make_golden.py --only-datasets):
file parsing, compositing, pose algebra, camera records.  The rays themselves are the HIP kernel's (tests/test_gpu_datasets.py);
here the camera records are pushed through the oracle's ray generation to show they describe the reference's cameras."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import dataset_fixture as fx  # noqa: E402
from mipnerf_pl_amd import datasets as D  # noqa: E402
from oracle import mipnerf_oracle as orc  # noqa: E402

G = np.load(os.path.join(HERE, "golden", "datasets_tiny.npz"))
FIELDS = D.Rays_keys


@pytest.fixture(scope="module")
def roots(tmp_path_factory):
    t = tmp_path_factory.mktemp("ds")
    return dict(blender=fx.write_blender(str(t / "blender")), multicam=fx.write_multicam(str(t / "multicam")),
                llff=fx.write_llff(str(t / "llff")))


def oracle_rays(rec):
    rec = rec.numpy()
    c2w = rec[:12].reshape(3, 4)
    w, h, near, far, lossmult = int(rec[21]), int(rec[22]), rec[23], rec[24], rec[25]
    if rec[26] == 0:
        return orc.generate_rays_blender(c2w, w, h, rec[27], near, far)
    return orc.generate_rays_multicam(c2w, rec[12:21].reshape(3, 3), w, h, near, far, lossmult)


def check_train(ds, tag, rtol):
    flat = np.concatenate([im.reshape(-1, 3) for im in ds.images])
    np.testing.assert_allclose(flat, G[tag + "_images"], rtol=0, atol=1e-7)
    assert len(ds) == flat.shape[0] == ds.num_pixels
    rays = [oracle_rays(r) for r in ds.cameras]
    for k in FIELDS:
        got = np.concatenate([getattr(r, k).reshape(-1, getattr(r, k).shape[-1]) for r in rays])
        np.testing.assert_allclose(got, G[f"{tag}_{k}"], rtol=rtol, atol=rtol, err_msg=f"{tag} {k}")


def check_images(ds, tag, rtol):
    assert len(ds) == int(G[tag + "_n"]) == ds.n_examples
    for i in range(len(ds)):
        np.testing.assert_allclose(ds.images[i], G[f"{tag}_image{i}"], rtol=0, atol=1e-7)
        r = oracle_rays(ds.cameras[i])
        for k in FIELDS:
            np.testing.assert_allclose(getattr(r, k), G[f"{tag}{i}_{k}"], rtol=rtol, atol=rtol, err_msg=f"{tag}{i} {k}")


def test_blender_files(roots):
    check_train(D.Blender(roots["blender"], "train", True, "all_images", device=None), "blender_train", 2e-6)
    check_train(D.Blender(roots["blender"], "train", False, "all_images"), "blender_train_black", 2e-6)
    check_images(D.Blender(roots["blender"], "val", True, "single_image"), "blender_val", 2e-6)
    with pytest.raises(AssertionError):
        D.Blender(roots["blender"], "val", True, "all_images")
    with pytest.raises(ValueError):
        D.Blender(roots["blender"], "train", True, "all_images", factor=3)
    half = D.Blender(roots["blender"], "train", True, "all_images", factor=2)       # 2x2 box mean before compositing
    assert half.sizes[0] == (5, 6) and abs(half.focal - 0.5 * 6 / np.tan(0.5 * 0.6911112070083618)) < 1e-9


def test_multicam_files(roots):
    ds = D.Multicam(roots["multicam"], "train", True, "all_images")
    assert len(set(ds.sizes)) == 3 and ds.cameras[1, 25] == 4.0            # three scales, lossmult 4^j
    check_train(ds, "multicam_train", 2e-6)
    check_images(D.Multicam(roots["multicam"], "test", True, "single_image"), "multicam_test", 2e-6)


def test_realdata360_files(roots):
    tr = D.RealData360(roots["llff"], "train", True, "all_images", factor=4)
    te = D.RealData360(roots["llff"], "test", True, "single_image", factor=4)
    assert tr.n_examples == 8 and te.n_examples == 2                       # every 8th image is held out (datasets.py:326-333)
    np.testing.assert_allclose(tr.camtoworlds, G["llff_train_c2w"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(te.camtoworlds, G["llff_test_c2w"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(tr.K_inv, G["llff_K_inv"], rtol=1e-12)
    check_train(tr, "llff_train", 1e-5)          # upstream computes these rays in float64; the records are float32
    check_images(te, "llff_test", 1e-5)
    with pytest.raises(ValueError):
        D.RealData360(roots["llff"], "train", True, "all_images", factor=0)   # upstream: division by zero -> inf rays
    assert D.dataset_dict["llff"] is D.RealData360 and D.dataset_dict["blender"] is D.Blender \
        and D.dataset_dict["multi_blender"] is D.Multicam


def test_render_path():
    np.testing.assert_allclose(D.create_spheric_poses(4), G["render_poses"], rtol=1e-12, atol=1e-12)
    rg = D.RenderGen(float(G["render_focal"]), [24, 20], 2, device=None)
    assert len(rg) == int(G["render_n"]) == 240
    for i in (0, 7, 119, 120, 239):
        r = oracle_rays(rg.cameras[i])
        for k in FIELDS:
            np.testing.assert_allclose(getattr(r, k), G[f"render{i}_{k}"], rtol=3e-6, atol=3e-6, err_msg=f"render{i} {k}")


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-device error path")
def test_rays_need_the_device(roots):
    ds = D.Blender(roots["blender"], "train", True, "all_images")
    with pytest.raises(RuntimeError, match="no HIP device"):
        ds[0]
    with pytest.raises(RuntimeError, match="HIP device"):
        ds.to("cpu")

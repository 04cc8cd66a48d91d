"""CPU: the numpy oracle against the golden vectors generated from the unmodified
reference (scripts/make_golden.py).  This is the pin that makes the oracle trustworthy."""
import os

import numpy as np
import pytest

from oracle import mipnerf_oracle as orc

FWD_CASES = [
    "fwd_c1_256x64_xavier", "fwd_c1_256x64_trained", "fwd_ragged_100x128_trained",
    "fwd_unbounded_24x256_trained", "fwd_disparity_32x64_trained",
]
NAMES = ("rgb", "distance", "acc", "weights", "t_samples")
# fp32 tolerance of the oracle vs the reference: both are fp32 on CPU, differences come from
# BLAS summation order and libm (sleef vs numpy) ulps, amplified by exp() in compositing.
TOL = 5e-5


def load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name + ".npz")))


def rays_of(g):
    return orc.Rays(*[g["rays_" + k] for k in orc.Rays._fields])


@pytest.mark.parametrize("name", FWD_CASES)
@pytest.mark.parametrize("wb", [True, False])
def test_forward_matches_reference(golden_dir, name, wb):
    g = load(golden_dir, name)
    params = orc.make_params(seed=int(g["param_seed"]), density_gain=float(g["density_gain"]))
    ret = orc.mipnerf_forward(params, rays_of(g), False, wb, num_samples=int(g["num_samples"]),
                              disparity=bool(g["disparity"]))
    for lvl in range(2):
        for nm, val in zip(NAMES, ret[lvl]):
            ref = g[f"wb{int(wb)}_l{lvl}_{nm}"]
            assert val.shape == ref.shape and val.dtype == np.float32
            np.testing.assert_allclose(val, ref, rtol=0, atol=TOL, err_msg=f"{name} l{lvl} {nm}")


def test_randomized_matches_reference(golden_dir):
    g = load(golden_dir, "fwd_randomized_64x128_trained")
    params = orc.make_params(seed=int(g["param_seed"]), density_gain=float(g["density_gain"]))
    ret = orc.mipnerf_forward(params, rays_of(g), True, True, num_samples=int(g["num_samples"]),
                              t_rand=g["t_rand"], u_rand=g["u_rand"])
    for lvl in range(2):
        for nm, val in zip(NAMES, ret[lvl]):
            np.testing.assert_allclose(val, g[f"wb1_l{lvl}_{nm}"], rtol=0, atol=TOL)


def test_stage_functions_match_reference(golden_dir):
    g = load(golden_dir, "stages_16x64_trained")
    rays = rays_of(g)
    N = int(g["num_samples"])
    params = orc.make_params(seed=int(g["param_seed"]), density_gain=float(g["density_gain"]))
    t0, (m0, c0) = orc.sample_along_rays(rays.origins, rays.directions, rays.radii, N, rays.near,
                                         rays.far, False, False)
    np.testing.assert_array_equal(t0, g["t0"])
    np.testing.assert_allclose(m0, g["means0"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(c0, g["covs0"], rtol=2e-5, atol=1e-12)
    enc0 = orc.integrated_pos_enc((g["means0"], g["covs0"]), 0, 16)
    np.testing.assert_allclose(enc0, g["enc0"], rtol=0, atol=2e-6)
    venc = orc.pos_enc(rays.viewdirs, 0, 4, True)
    np.testing.assert_allclose(venc, g["viewdirs_enc"], rtol=0, atol=1e-6)
    raw_rgb, raw_density = orc.mlp_forward(params, g["enc0"], g["viewdirs_enc"])
    np.testing.assert_allclose(raw_rgb, g["raw_rgb0"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(raw_density, g["raw_density0"], rtol=2e-5, atol=2e-4)
    np.testing.assert_allclose(orc.softplus(g["raw_density0"] - np.float32(1)), g["density0"],
                               rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(orc.sigmoid(g["raw_rgb0"]) * np.float32(1.002) - np.float32(0.001),
                               g["rgb0"], rtol=0, atol=1e-6)
    comp = orc.volumetric_rendering(g["rgb0"], g["density0"], g["t0"], rays.directions, True)
    for val, key in zip(comp, ("comp_rgb0", "distance0", "acc0", "weights0")):
        np.testing.assert_allclose(val, g[key], rtol=0, atol=5e-6)
    t1, (m1, c1) = orc.resample_along_rays(rays.origins, rays.directions, rays.radii, g["t0"],
                                           g["weights0"], False)
    np.testing.assert_allclose(t1, g["t1"], rtol=0, atol=5e-6)
    np.testing.assert_allclose(orc.distloss(g["weights0"], g["t0"]), g["distloss0"], rtol=1e-5)
    pdf_t = orc.sorted_piecewise_constant_pdf(g["pdf_bins"], g["pdf_w"], N + 1, False)
    np.testing.assert_allclose(pdf_t, g["pdf_t"], rtol=0, atol=5e-6)


def test_training_loss_matches_reference(golden_dir):
    g = load(golden_dir, "train_64x64_trained")
    ret = [tuple(g[f"wb1_l{l}_{nm}"] for nm in NAMES) for l in range(2)]
    loss = orc.training_loss(ret, rays_of(g), g["gt"])
    np.testing.assert_allclose(loss, g["loss"], rtol=1e-5)


# ---- known-answer checks derived from the reference code ----------
def test_known_answers():
    B, N = 3, 64
    rays = orc.synthetic_rays(B, seed=9)
    t = np.broadcast_to(orc.torch_linspace(2, 6, N + 1), (B, N + 1)).astype(np.float32)
    rgb = np.full((B, N, 3), 0.25, np.float32)
    zero = np.zeros((B, N, 1), np.float32)
    c, d, a, w = orc.volumetric_rendering(rgb, zero, t, rays.directions, True)
    assert np.all(a == 0) and np.all(c == 1) and np.all(d == t[:, 0])     # zero density, white
    c, d, a, w = orc.volumetric_rendering(rgb, zero, t, rays.directions, False)
    assert np.all(c == 0)
    dens = zero.copy()
    dens[:, 10, 0] = 1e6                                                   # opaque bin 10
    c, d, a, w = orc.volumetric_rendering(rgb, dens, t, rays.directions, False)
    assert np.allclose(w[:, 10], 1) and np.allclose(np.delete(w, 10, axis=1), 0)
    np.testing.assert_allclose(d, 0.5 * (t[:, 10] + t[:, 11]), rtol=1e-6)
    # uniform weights => resampled t is the linear map of u
    tt = orc.sorted_piecewise_constant_pdf(t, np.ones((B, N), np.float32), N + 1, False)
    np.testing.assert_allclose(tt, t, atol=2e-6)
    # IPE with zero covariance == plain PE ordering [sin l0(xyz) .. l15 | cos ..]
    m = np.array([[[0.1, -0.2, 0.3]]], np.float32)
    e = orc.integrated_pos_enc((m, np.zeros_like(m)), 0, 16)
    assert e.shape == (1, 1, 96)
    np.testing.assert_allclose(e[0, 0, :3], np.sin(m[0, 0]), atol=1e-7)
    np.testing.assert_allclose(e[0, 0, 3:6], np.sin(2 * m[0, 0]), atol=1e-7)
    np.testing.assert_allclose(e[0, 0, 48:51], np.cos(m[0, 0]), atol=1e-6)
    # MLP with zero weights => rgb = sigmoid(b)*1.002-0.001
    p = orc.make_params(seed=0)
    for k in p:
        if k.endswith("weight"):
            p[k][...] = 0
    rr, dd = orc.mlp_forward(p, np.zeros((1, 2, 96), np.float32), np.zeros((1, 27), np.float32))
    np.testing.assert_allclose(rr[0, 0], p["color_layer.bias"], atol=1e-7)
    np.testing.assert_allclose(dd[0, 0], p["density_layer.bias"], atol=1e-7)


def grad_check(g, grads, rtol_l2, rtol_smp):
    """Compare full gradient tensors with the (l2, strided samples) summary stored in a golden file."""
    worst = 0.0
    for k, gr in grads.items():
        flat = np.asarray(gr, dtype=np.float32).ravel()
        l2 = float(np.sqrt((flat.astype(np.float64) ** 2).sum()))
        ref_l2 = float(g["g_l2_" + k])
        smp_ref = g["g_smp_" + k]
        stride = max(1, flat.size // smp_ref.size)
        smp = flat[::stride][:smp_ref.size]
        rel = abs(l2 - ref_l2) / max(ref_l2, 1e-20)
        es = float(np.max(np.abs(smp - smp_ref))) / max(float(np.max(np.abs(smp_ref))), 1e-20)
        assert rel <= rtol_l2 and es <= rtol_smp, (k, rel, es)
        worst = max(worst, rel, es)
    return worst


def test_mlp_backward_matches_reference_autograd(golden_dir):
    g = load(golden_dir, "mlp_bwd_8x32_trained")
    params = orc.make_params(seed=int(g["param_seed"]), density_gain=float(g["density_gain"]))
    rr, dd = orc.mlp_forward(params, g["enc"], g["venc"])
    np.testing.assert_allclose(rr, g["raw_rgb"], atol=2e-5)
    np.testing.assert_allclose(dd, g["raw_density"], atol=2e-4)
    grads = orc.mlp_backward(params, g["enc"], g["venc"], g["d_rgb"], g["d_den"])
    assert list(grads) == list(params)
    grad_check(g, grads, 1e-5, 1e-5)


def test_ray_generation_matches_reference(golden_dir):
    """datasets.py:214-263 (Blender) and :116-168 (Multicam) ray generation, oracle vs the reference's own output."""
    g = load(golden_dir, "raygen_20x14")
    W, H, focal = int(g["width"]), int(g["height"]), float(g["focal"])
    for i in range(2):
        o = orc.generate_rays_blender(g["blender_c2w"][i], W, H, focal, float(g["near"]), float(g["far"]))
        for k in orc.Rays._fields:
            np.testing.assert_allclose(getattr(o, k), g["blender_" + k][i], atol=2e-6, err_msg=f"blender {k}")
        w, h = W // 2 ** i, H // 2 ** i
        o = orc.generate_rays_multicam(g["multicam_c2w"][i], g["multicam_pix2cam"][i], w, h, 2.0, 6.0, 4.0 ** i)
        for k in orc.Rays._fields:
            np.testing.assert_allclose(getattr(o, k), g[f"multicam{i}_" + k], atol=2e-6, err_msg=f"multicam {k}")


def test_eval_errors_matches_reference(golden_dir):
    g = load(golden_dir, "metrics_45x70")
    psnr, ssim = orc.eval_errors(g["pred"], g["gt"])
    assert abs(float(psnr) - float(g["psnr"])) <= 1e-4 and abs(float(ssim) - float(g["ssim"])) <= 1e-5


@pytest.mark.parametrize("name", ["full_c2_4096x128", "full_c4_8192x256"])
def test_fullsize_golden_inputs_regenerate(golden_dir, name):
    """The full-size goldens store only the reference's outputs; their inputs must regenerate bit-for-bit from the seeds."""
    import hashlib
    import os
    import synthetic_inputs as syn
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    rays = syn.synthetic_rays(int(g["batch"]), seed=int(g["ray_seed"]), unbounded=bool(g["unbounded"]))
    params = syn.make_params(seed=int(g["param_seed"]), density_gain=float(g["density_gain"]))
    h = hashlib.sha256()
    for a in rays:
        h.update(np.ascontiguousarray(a).tobytes())
    for k in sorted(params):
        h.update(np.ascontiguousarray(params[k]).tobytes())
    assert h.hexdigest() == str(g["input_sha256"])
    assert g["l1_rgb"].shape == (int(g["batch"]), 3) and np.isfinite(g["l1_rgb"]).all()


def test_oracle_on_fullsize_golden_sample(golden_dir):
    """Oracle vs the reference's full-size configs[1] outputs on a 96-ray sample (rays are independent)."""
    import os
    import synthetic_inputs as syn
    g = np.load(os.path.join(golden_dir, "full_c2_4096x128.npz"))
    rays = syn.synthetic_rays(int(g["batch"]), seed=int(g["ray_seed"]))
    params = syn.make_params(seed=int(g["param_seed"]), density_gain=float(g["density_gain"]))
    idx = np.arange(0, 4096, 43)[:96]
    sub = orc.Rays(*[a[idx] for a in rays])
    ret = orc.mipnerf_forward(params, sub, False, True, num_samples=int(g["num_samples"]))
    for lvl in range(2):
        assert np.max(np.abs(ret[lvl][0] - g[f"l{lvl}_rgb"][idx])) <= 2e-4
        assert np.max(np.abs(ret[lvl][1] - g[f"l{lvl}_distance"][idx])) <= 5e-4
        assert np.max(np.abs(ret[lvl][2] - g[f"l{lvl}_acc"][idx])) <= 2e-4


@pytest.mark.parametrize("name,kw,tol1", [
    ("ctor_levels1_40x64", dict(num_levels=1), 2e-4),
    ("ctor_scalars_40x64", dict(min_deg_point=1, max_deg_point=17, resample_padding=0.05, density_bias=-0.5, rgb_padding=0.01), 2e-4),
    ("ctor_noint_40x64", dict(disable_integration=True), 1e-2),      # level 1 is ill-conditioned without the integration
    ("var_w128_48x64", dict(), 2e-4), ("var_noview_48x64", dict(use_viewdirs=False), 2e-4),
    ("var_w200c72_48x64", dict(), 2e-4), ("var_w100c40_48x64", dict(), 2e-4)])
def test_oracle_constructor_variants(golden_dir, name, kw, tol1):
    """Oracle vs the reference's outputs for constructor arguments off their defaults (round-2 goldens)."""
    import os
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    arch = {}
    if "net_width" in g:
        arch = dict(net_width=int(g["net_width"]), net_width_condition=int(g["net_width_condition"]))
    params = orc.make_params(seed=int(g["param_seed"]), density_gain=float(g["density_gain"]), **arch)
    rays = orc.Rays(*[g["rays_" + k] for k in orc.Rays._fields])
    ret = orc.mipnerf_forward(params, rays, False, True, num_samples=int(g["num_samples"]), **kw)
    for lvl in range(len(ret)):
        for nm, val in zip(("rgb", "distance", "acc", "weights", "t_samples"), ret[lvl]):
            d = float(np.max(np.abs(val - g[f"wb1_l{lvl}_{nm}"])))
            assert d <= (2e-4 if lvl == 0 else tol1), (name, lvl, nm, d)


def test_360_oracle_blocks_pinned_by_the_reference_where_it_is_right():
    """`contract` (mip.py:424-428, |x| > 1) and the inverse-depth fence posts + Gaussian means of
    `sample_along_rays_360` (mip.py:106-124) of the reference ARE correct and pin the corresponding oracle blocks
    (scripts/make_golden.py --only-360); full covariances / parameterization / integrated_pos_enc_360 are wrong upstream and
    cannot (the oracle follows the paper there: "parity unpinned")."""
    from oracle import mipnerf360_oracle as o360
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pin360_24x64.npz"))
    x = g["contract_x"]
    outside = np.linalg.norm(x, axis=-1) > 1
    assert outside.sum() > 200 and (~outside).sum() > 0
    assert np.abs(o360.contract(x)[outside] - g["contract_y"][outside]).max() <= 5e-7
    assert np.array_equal(o360.contract(x)[~outside], x[~outside])
    rays = orc.Rays(*[g["rays_" + k] for k in orc.Rays._fields])
    N = int(g["num_samples"])
    for tag, randomized, tr in (("det", False, None), ("rand", True, g["t_rand"])):
        t_inv, t, (m, c) = o360.sample_along_rays_360(rays.origins, rays.directions, rays.radii, N, rays.near, rays.far, randomized, t_rand=tr)
        assert np.array_equal(t_inv, g[f"{tag}_t_inv"])                              # bit-exact fence posts
        assert np.abs(m - g[f"{tag}_means"]).max() <= 2e-7 * np.abs(g[f"{tag}_means"]).max()
        assert np.array_equal(t, (np.float32(1) / t_inv).astype(np.float32))


def test_oracle_renders_chunks_of_the_reference_frame(golden_dir):
    """Round 6: tests/golden/frame_c5_800x800.npz is ONE whole 800 x 800 RenderGen pose rendered by the unmodified reference (BASELINE
    configs[4]; scripts/make_golden.py --only-frame).  The oracle -- rays restated from the stored pose (RenderGen's pixel -> ray rule is
    Multicam's, render_video.py:60-80), forward on the trained field -- reproduces two of its 79 chunks on CPU: a slice of the middle of
    the image and the ragged 1024-ray tail."""
    import hashlib
    g = load(golden_dir, "frame_c5_800x800")
    f = load(golden_dir, str(g["field"]))
    params = {k[2:]: f[k] for k in f if k.startswith("p_")}
    h = hashlib.sha256()
    for k in sorted(params):
        h.update(np.ascontiguousarray(params[k]).tobytes())
    assert h.hexdigest() == str(g["field_sha256"])
    size, focal = int(g["cfg_size"]), float(g["focal"])
    assert g["fine_rgb"].shape == (1, size, size, 3) and g["distance"].shape == (size, size) and size * size % int(g["cfg_chunk"]) == 1024
    assert np.array_equal(g["val_mask"], np.ones((1, size, size, 1), np.float32))                    # lossmult (mip.py:407)
    p2c = np.array([[1.0 / focal, 0.0, -0.5 * size / focal], [0.0, -1.0 / focal, 0.5 * size / focal], [0.0, 0.0, -1.0]])
    R = orc.generate_rays_multicam(g["pose"][:3], p2c, size, size, 2.0, 6.0, 1.0)
    flat = orc.Rays(*[a.reshape(size * size, -1) for a in R])
    for lo, n in ((size * size - 1024, 1024), (400 * size + 100, 600)):
        part = orc.Rays(*[a[lo:lo + n] for a in flat])
        ret = orc.mipnerf_forward(params, part, False, True, num_samples=int(g["cfg_num_samples"]))
        for name, lvl, k in (("coarse_rgb", 0, 0), ("fine_rgb", 1, 0), ("distance", 1, 1), ("acc", 1, 2)):
            want = g[name].reshape(size * size, -1)[lo:lo + n]
            got = np.asarray(ret[lvl][k]).reshape(n, -1)
            np.testing.assert_allclose(got, want, rtol=0, atol=2e-4, err_msg=f"{name} rays {lo}..{lo + n}")
    gt = g["gt_u8"].astype(np.float32) / 255.0
    assert abs(float(-10.0 * np.log10(np.mean((g["fine_rgb"][0].astype(np.float64) - gt) ** 2))) - float(g["psnr_fine"])) < 1e-6

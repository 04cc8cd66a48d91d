"""GPU parity, stage by stage: every C-ABI entry point against the oracle on identical inputs."""
import numpy as np
import pytest
import torch

from oracle import mipnerf_oracle as orc
from mipnerf_pl_amd.mlp_plan import bf16_round

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import gpu_util
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    return gpu_util


@pytest.fixture(scope="module")
def stage(G):
    g = G.load_golden("stages_16x64_trained")
    return g, G.rays_of(g), G.to_dev(G.rays_of(g))


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def test_hardware_selftest(G):
    from mipnerf_pl_amd import ops
    print(ops.selftest())


def test_single_hip_runtime_and_torch_streams(G):
    """The C-ABI library must run on the SAME HIP runtime instance as torch (one libamdhip64 mapped), so
    that torch streams are valid stream handles for it and torch.cuda.synchronize() covers its kernels."""
    from mipnerf_pl_amd import ops
    mapped = {l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l}
    assert len(mapped) == 1, mapped
    rays = orc.synthetic_rays(64, seed=1)
    R = G.to_dev(rays)
    to, _ = orc.sample_along_rays(rays.origins, rays.directions, rays.radii, 128, rays.near, rays.far, False, False)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):          # non-default stream: handle is passed through the C ABI
        t = ops.sample_t(128, R.near, R.far, False, False)
    side.synchronize()
    assert G.maxdiff(t, to) == 0.0


@pytest.mark.parametrize("N", [64, 100, 128, 256])
@pytest.mark.parametrize("disparity", [False, True])
@pytest.mark.parametrize("randomized", [False, True])
def test_sample_along_rays(G, N, disparity, randomized):
    from mipnerf_pl_amd import ops
    rays = orc.synthetic_rays(37, seed=N, unbounded=True)
    R = G.to_dev(rays)
    t_rand = np.random.default_rng(1).random((37, N + 1), dtype=np.float32) if randomized else None
    t, (m, c) = ops.sample_along_rays(R.origins, R.directions, R.radii, N, R.near, R.far, randomized, disparity, "cone",
                                      t_rand=T(t_rand) if randomized else None)
    to, (mo, co) = orc.sample_along_rays(rays.origins, rays.directions, rays.radii, N, rays.near, rays.far,
                                         randomized, disparity, t_rand=t_rand)
    G.record(f"sample_along_rays N={N} disp={disparity} rand={randomized}", t=G.maxdiff(t, to), means=G.maxdiff(m, mo))
    # same op order, IEEE ops: t bit-exact except 1/x in the disparity form (1 ulp)
    assert G.maxdiff(t, to) <= (0 if not disparity else 4e-6)
    assert G.maxdiff(m, mo) <= (0 if not disparity else 2e-5)
    np.testing.assert_allclose(c.cpu().numpy(), co, rtol=2e-5, atol=1e-12)


def test_cast_rays_and_ipe(G, stage):
    from mipnerf_pl_amd import ops, _lib as L
    g, rays, R = stage
    t1 = T(g["t1"])
    m, c = ops.cast_rays(t1, R.origins, R.directions, R.radii)
    assert G.maxdiff(m, g["means1"]) == 0.0          # bit-exact (no FMA contraction, same order)
    np.testing.assert_allclose(c.cpu().numpy(), g["covs1"], rtol=3e-6, atol=1e-14)
    enc = ops.integrated_pos_enc((T(g["means1"]), T(g["covs1"])), 0, 16)
    e1 = G.maxdiff(enc, g["enc1"])
    fused = ops.cast_ipe(t1, R.origins, R.directions, R.radii, 0, 16)
    e2 = G.maxdiff(fused, g["enc1"])
    encb = ops.cast_ipe(t1, R.origins, R.directions, R.radii, 0, 16, precision=L.PREC_BF16)
    e3 = G.maxdiff(encb.float(), bf16_round(g["enc1"]))
    G.record("ipe", separate=e1, fused=e2, bf16_vs_rounded=e3)
    assert e1 <= 2e-6 and e2 <= 2e-6      # sin/exp within a few ulp of torch's
    assert e3 <= 2 ** -8                  # at most one bf16 ulp (|enc| <= 1) where fp32 differs at a rounding tie
    # disable_integration: covariance zeroed -> plain positional encoding
    pe = ops.cast_ipe(t1, R.origins, R.directions, R.radii, 0, 16, disable_integration=True)
    peo = orc.integrated_pos_enc((g["means1"], np.zeros_like(g["covs1"])), 0, 16)
    assert G.maxdiff(pe, peo) <= 2e-6


def test_pos_enc(G, stage):
    from mipnerf_pl_amd import ops
    g, rays, R = stage
    v = ops.pos_enc(R.viewdirs, 0, 4, True)
    assert v.shape == (16, 27)
    assert G.maxdiff(v, g["viewdirs_enc"]) <= 1e-6
    v32 = ops.pos_enc(R.viewdirs, 0, 4, True, ld=32)
    assert float(v32[:, 27:].abs().max()) == 0.0


def test_mlp_fp32(G, stage):
    from mipnerf_pl_amd import _lib as L
    g, rays, R = stage
    params = orc.make_params(seed=int(g["param_seed"]), density_gain=float(g["density_gain"]))
    model = G.make_model(params, 64, "fp32")
    with torch.no_grad():
        raw_rgb, raw_density, act = model.mlp(T(g["enc0"]), T(g["viewdirs_enc"]), precision=L.PREC_FP32,
                                              return_activated=True)
    er, ed = G.maxdiff(raw_rgb, g["raw_rgb0"]), G.maxdiff(raw_density, g["raw_density0"])
    ea = G.maxdiff(act[..., :3], g["rgb0"])
    es = float(np.max(np.abs(act[..., 3:].cpu().numpy() - g["density0"]) / (1 + np.abs(g["density0"]))))
    G.record("mlp_fp32", raw_rgb=er, raw_density=ed, rgb=ea, density_rel=es)
    assert er <= 2e-5 and ed <= 2e-4 and ea <= 5e-6 and es <= 2e-5


def mlp_bf16_emulation(params, x, v):
    """numpy model of the bf16 kernel: bf16 weights/activations, fp32 accumulate, fp32 bias/ReLU."""
    r = bf16_round
    P = {k: (r(w) if k.endswith("weight") else w) for k, w in params.items()}
    inputs = r(x)
    x = inputs
    for i in range(8):
        x = r(np.maximum(x @ P[f"layers.{i}.0.weight"].T + P[f"layers.{i}.0.bias"], 0))
        if i % 4 == 0 and i > 0:
            x = np.concatenate([x, inputs], -1)
    dens = x @ P["density_layer.weight"].T + P["density_layer.bias"]
    b = r(x @ P["extra_layer.weight"].T + P["extra_layer.bias"])
    vd = np.broadcast_to(r(v)[:, None, :], (x.shape[0], x.shape[1], v.shape[-1]))
    x = r(np.maximum(np.concatenate([b, vd], -1) @ P["view_layers.0.0.weight"].T + P["view_layers.0.0.bias"], 0))
    return (x @ P["color_layer.weight"].T + P["color_layer.bias"]).astype(np.float32), dens.astype(np.float32)


def test_mlp_bf16(G, stage):
    from mipnerf_pl_amd import _lib as L
    g, rays, R = stage
    params = orc.make_params(seed=int(g["param_seed"]), density_gain=float(g["density_gain"]))
    model = G.make_model(params, 64, "bf16")
    enc, venc = T(g["enc0"]), T(g["viewdirs_enc"])
    with torch.no_grad():
        rgb_dma, den_dma = [x.clone() for x in model.mlp(enc, venc, precision=L.PREC_BF16)]
        ctx = model.mlp.native(enc.device)
        ctx.set_option(0, 0)          # register-staged ring instead of LDS DMA: must be bit-identical
        rgb_reg, den_reg = [x.clone() for x in model.mlp(enc, venc, precision=L.PREC_BF16)]
        ctx.set_option(0, 1)
    assert torch.equal(rgb_dma, rgb_reg) and torch.equal(den_dma, den_reg)
    er_, ed_ = mlp_bf16_emulation(params, g["enc0"], g["viewdirs_enc"])
    e_emu_r, e_emu_d = G.maxdiff(rgb_dma, er_), G.maxdiff(den_dma, ed_)
    e_f32_r, e_f32_d = G.maxdiff(rgb_dma, g["raw_rgb0"]), G.maxdiff(den_dma, g["raw_density0"])
    G.record("mlp_bf16", vs_emulation_rgb=e_emu_r, vs_emulation_density=e_emu_d, vs_fp32_rgb=e_f32_r,
             vs_fp32_density=e_f32_d)
    # vs the bf16 emulation only accumulation order differs (occasional 1-bf16-ulp flips of an activation)
    assert e_emu_r <= 6e-3 and e_emu_d <= 0.15
    # vs the fp32 reference: bf16 rounding of 10 chained layers (|raw_density| up to ~23 here)
    assert e_f32_r <= 2e-2 and e_f32_d <= 0.4


@pytest.mark.parametrize("N", [64, 100, 128, 256, 300, 513, 700, 1024])       # 513+: the K = 16 bucket (round 5: N <= 1024)
@pytest.mark.parametrize("white", [True, False])
def test_volumetric_rendering(G, N, white):
    from mipnerf_pl_amd import ops
    B = 33
    rng = np.random.default_rng(N)
    rays = orc.synthetic_rays(B, seed=3)
    t = np.sort(rng.uniform(2, 6, (B, N + 1)).astype(np.float32), axis=-1)
    rgb = rng.uniform(0, 1, (B, N, 3)).astype(np.float32)
    dens = (rng.uniform(0, 1, (B, N, 1)) ** 8 * 60).astype(np.float32)
    dens[0] = 0            # empty ray
    dens[1] = 0
    dens[1, N // 2] = 1e6  # single opaque bin
    out = ops.volumetric_rendering(T(rgb), T(dens), T(t), T(rays.directions), white)
    ref = orc.volumetric_rendering(rgb, dens, t, rays.directions, white)
    errs = [G.maxdiff(a, b) for a, b in zip(out, ref)]
    G.record(f"volumetric_rendering N={N} white={white}", rgb=errs[0], distance=errs[1], acc=errs[2], weights=errs[3])
    assert errs[0] <= 1e-5 and errs[1] <= 5e-5 and errs[2] <= 1e-5 and errs[3] <= 1e-5
    # known answers: zero density => acc 0, rgb = background, distance = near
    assert float(out[2][0]) == 0.0 and float(out[1][0]) == float(t[0, 0])
    assert torch.all(out[0][0] == (1.0 if white else 0.0))
    assert abs(float(out[3][1, N // 2]) - 1.0) < 1e-6


@pytest.mark.parametrize("N", [64, 100, 128, 256, 513, 1024])
@pytest.mark.parametrize("randomized", [False, True])
def test_resample_along_rays(G, N, randomized):
    from mipnerf_pl_amd import ops
    B = 29
    rng = np.random.default_rng(N + 7)
    rays = orc.synthetic_rays(B, seed=5)
    t = np.sort(rng.uniform(2, 6, (B, N + 1)).astype(np.float32), axis=-1)
    w = (rng.uniform(0, 1, (B, N)) ** 6).astype(np.float32)
    w[0] = 0
    w[1] = 0
    w[1, 7] = 1.0
    u = rng.random((B, N + 1), dtype=np.float32) if randomized else None
    R = G.to_dev(rays)
    tn, (m, c) = ops.resample_along_rays(R.origins, R.directions, R.radii, T(t), T(w), randomized, "cone", True, 0.01,
                                         u_rand=T(u) if randomized else None)
    to, _ = orc.resample_along_rays(rays.origins, rays.directions, rays.radii, t, w, randomized, u_rand=u)
    e = G.maxdiff(tn, to)
    G.record(f"resample N={N} rand={randomized}", t=e)
    # Adversarial input: weights ~ U^6 give bins whose pdf is ~5e-4 and (random fence posts) up to 0.2 wide, where the
    # reference itself is ill-conditioned: a 1e-7 relative change of weight_sum (fp32 summation order) moves t by
    # 1e-7 / 5e-4 * 0.2 = 4e-5.  Prefix sums run in fp64 like torch's CPU cumsum.  Realistic weights: see
    # test_resample_matches_reference_stage (5e-6).
    assert e <= 1e-4
    tn_np = tn.cpu().numpy()
    assert np.all(np.diff(tn_np, axis=-1) >= 0)                       # sorted, no explicit sort needed
    assert np.all(tn_np >= t[:, :1] - 1e-6) and np.all(tn_np <= t[:, -1:] + 1e-6)


def test_resample_matches_reference_stage(G, stage):
    from mipnerf_pl_amd import ops
    g, rays, R = stage
    t1 = ops.resample_t(T(g["t0"]), T(g["weights0"]), False, 0.01)
    e = G.maxdiff(t1, g["t1"])
    G.record("resample_stage_golden", t=e)
    assert e <= 5e-6


def test_sorted_piecewise_constant_pdf_edge_cases(G, stage):
    from mipnerf_pl_amd import ops
    g, _, _ = stage
    w = T(g["pdf_w"])
    w0 = w.clone()
    out = ops.sorted_piecewise_constant_pdf(T(g["pdf_bins"]), w, g["pdf_t"].shape[-1], False)
    assert torch.equal(w, w0), "weights must not be mutated (the reference mutates its argument, mip.py:184)"
    e = G.maxdiff(out, g["pdf_t"])
    G.record("pdf_edge_cases", t=e)
    assert e <= 5e-6


def test_generate_rays_matches_reference(G):
    """Device-side ray generation vs the reference's Blender / Multicam `_generate_rays` goldens:
    whole images, and a random (camera, pixel) gather as a training batch would draw it."""
    from mipnerf_pl_amd import ops
    g = G.load_golden("raygen_20x14")
    W, H, focal = int(g["width"]), int(g["height"]), float(g["focal"])
    cams = [ops.camera_record(g["blender_c2w"][i], W, H, 2.0, 6.0, focal=focal) for i in range(2)]
    cams += [ops.camera_record(g["multicam_c2w"][i], W // 2 ** i, H // 2 ** i, 2.0, 6.0, pix2cam=g["multicam_pix2cam"][i],
                               lossmult=4.0 ** i) for i in range(2)]
    table = torch.stack(cams).to(G.DEV)
    worst = 0.0
    for ci, (prefix, idx) in enumerate([("blender_", 0), ("blender_", 1), ("multicam0_", None), ("multicam1_", None)]):
        w, h = int(cams[ci][21]), int(cams[ci][22])
        cidx = torch.full((w * h,), ci, dtype=torch.int32, device=G.DEV)
        rays = ops.generate_rays(table, num_rays=w * h, cam_idx=cidx)
        for k in orc.Rays._fields:
            ref = g[prefix + k] if idx is None else g[prefix + k][idx]
            e = G.maxdiff(getattr(rays, k).reshape(h, w, -1), ref.astype(np.float32))
            worst = max(worst, e)
            assert e <= 3e-6, (prefix, k, e)
    # random gather over cameras 0/1
    rng = np.random.default_rng(0)
    n = 777
    ci = rng.integers(0, 2, n)
    pi = rng.integers(0, W * H, n)
    rays = ops.generate_rays(table, cam_idx=torch.from_numpy(ci).to(G.DEV), pix_idx=torch.from_numpy(pi).to(G.DEV))
    for k in orc.Rays._fields:
        ref = g["blender_" + k].reshape(2, W * H, -1)[ci, pi].astype(np.float32)
        assert G.maxdiff(getattr(rays, k), ref) <= 3e-6, k
    G.record("generate_rays", worst=worst)


def test_eval_errors_matches_reference(G):
    """fused PSNR + 11x11-Gaussian SSIM kernel vs the reference's eval_errors (golden) and the oracle
    on an 800x800 frame (size-independent check: identical images -> ssim 1, psnr = +inf)."""
    from mipnerf_pl_amd import ops
    g = G.load_golden("metrics_45x70")
    pred, gt = torch.from_numpy(g["pred"]).to(G.DEV), torch.from_numpy(g["gt"]).to(G.DEV)
    psnr, ssim = ops.eval_errors(pred[None], gt[None])
    e1, e2 = abs(float(psnr) - float(g["psnr"])), abs(float(ssim) - float(g["ssim"]))
    G.record("eval_errors", psnr_abs=e1, ssim_abs=e2)
    assert e1 <= 1e-4 and e2 <= 1e-5
    big = torch.rand(800, 800, 3, device=G.DEV)
    p2, s2 = ops.eval_errors(big, big)
    assert abs(float(s2) - 1.0) <= 1e-6 and float(p2) == float("inf")
    noisy = (big + 0.05 * torch.randn_like(big)).clamp(0, 1)
    p3, s3 = ops.eval_errors(noisy, big)
    op, os_ = orc.eval_errors(noisy.cpu().numpy(), big.cpu().numpy())
    assert abs(float(p3) - float(op)) <= 1e-3 and abs(float(s3) - float(os_)) <= 2e-5


@pytest.mark.parametrize("contracted", [True, False])
@pytest.mark.parametrize("randomized", [False, True])
def test_mipnerf360_path(G, contracted, randomized):
    """the unbounded-scene path (s-space sampling, full-covariance frustum Gaussians, scene contraction of
    mean and covariance, off-axis IPE) on the GPU against oracle/mipnerf360_oracle.py.  PARITY UNPINNED: the oracle
    restates Barron et al. 2022 (the reference's own code for this path is dead and wrong, see the oracle header);
    stated tolerances: t 1e-6 relative, means 2e-6, covariances 3e-4 of the largest entry (fp32 limit at |x| ~ 1e3: measured 1.5e-4),
    features 1e-4 when contracted (|y| <= 2), and the structure properties below."""
    from mipnerf_pl_amd import ops, _lib as L
    from oracle import mipnerf360_oracle as o360
    DEV = G.DEV
    rng = np.random.default_rng(31)
    B, N, Lf = 96, 64, 6
    o = rng.uniform(-0.5, 0.5, (B, 3)).astype(np.float32)
    d = rng.standard_normal((B, 3)).astype(np.float32)
    d *= rng.uniform(0.8, 1.2, (B, 1)).astype(np.float32) / np.linalg.norm(d, axis=-1, keepdims=True)
    r = rng.uniform(5e-4, 4e-3, (B, 1)).astype(np.float32)
    near = np.full((B, 1), 0.2, np.float32)
    far = rng.uniform(30.0, 1000.0, (B, 1)).astype(np.float32)
    tr = rng.uniform(0, 1, (B, N + 1)).astype(np.float32) if randomized else None
    want_tinv, want_t, (m0, c0) = o360.sample_along_rays_360(o, d, r, N, near, far, randomized, t_rand=tr)
    T = lambda a: torch.from_numpy(a).to(DEV)     # noqa: E731
    t_inv, (gm0, gc0) = ops.sample_along_rays_360(T(o), T(d), T(r), N, T(near), T(far), randomized, False, "cone",
                                                  t_rand=None if tr is None else T(tr))
    np.testing.assert_allclose(t_inv.cpu().numpy(), want_tinv, rtol=1e-6, atol=0)
    np.testing.assert_allclose(gm0.cpu().numpy(), m0, rtol=2e-5, atol=2e-5)
    sc0 = np.abs(c0).max(axis=(-1, -2), keepdims=True)
    assert (np.abs(gc0.cpu().numpy() - c0) / sc0).max() <= 2e-5
    # from here on the kernels get the ORACLE's t, so that rounding of 1/t_inv (amplified by 2^l) does not enter
    t = T(want_t)
    want_m, want_c = o360.cast_rays_360(want_t, o, d, r, contracted)
    gm, gc = ops.cast_rays_360(t, T(o), T(d), T(r), contracted=contracted)
    np.testing.assert_allclose(gm.cpu().numpy(), want_m, rtol=2e-6, atol=2e-6)
    sc = np.abs(want_c).max(axis=(-1, -2), keepdims=True)
    cerr = float((np.abs(gc.cpu().numpy() - want_c) / sc).max())
    enc = ops.cast_ipe_360(t, T(o), T(d), T(r), 0, Lf, contracted=contracted).cpu().numpy()
    want_e = o360.integrated_pos_enc_360((want_m, want_c), 0, Lf)
    assert enc.shape == (B, N, 42 * Lf) and np.isfinite(enc).all() and np.abs(enc).max() <= 1.0 + 1e-6
    eerr = float(np.abs(enc - want_e).max()) if contracted else float(np.abs(enc[..., :42] - want_e[..., :42]).max())
    G.record(f"mipnerf360 contracted={contracted} randomized={randomized}", cov_rel=cerr, enc_abs=eerr)
    assert cerr <= (3e-4 if contracted else 2e-6)
    assert eerr <= (1e-4 if contracted else 1e-3)
    # bf16 features: one bf16 ulp of the fp32 kernel's output
    e16 = ops.cast_ipe_360(t, T(o), T(d), T(r), 0, Lf, contracted=contracted, precision=L.PREC_BF16).float().cpu().numpy()
    assert np.abs(e16 - enc).max() <= 2 ** -8
    # structure: contracted means live in the ball of radius 2, covariances stay symmetric PSD
    if contracted:
        assert np.linalg.norm(gm.cpu().numpy(), axis=-1).max() < 2.0
    cg = gc.cpu().numpy().astype(np.float64)
    assert np.abs(cg - np.swapaxes(cg, -1, -2)).max() == 0.0
    assert np.linalg.eigvalsh(cg).min() >= -1e-6 * np.abs(cg).max()


def test_mipnerf360_free_functions(G):
    """The reference's free functions for unbounded scenes, by name (mip.py:424-447 contract / parameterization, :292-319
    integrated_pos_enc_360), on GIVEN Gaussians: against the oracle (parity unpinned, paper equations).  The generic
    J Sigma J^T of `parameterization` is conditioned like |x| (raymath360.hpp): covariances within 1e-3 of the largest entry."""
    from mipnerf_pl_amd import ops, _lib as L
    from oracle import mipnerf360_oracle as o360
    DEV = G.DEV
    rng = np.random.default_rng(41)
    M = 4096
    x = (rng.standard_normal((M, 3)) * rng.choice([0.3, 2.0, 30.0, 400.0], size=(M, 1))).astype(np.float32)
    A = rng.standard_normal((M, 3, 3)).astype(np.float32) * rng.choice([1e-2, 1.0], size=(M, 1, 1)).astype(np.float32)
    cov = (A @ np.swapaxes(A, -1, -2)).astype(np.float32)
    T = lambda a: torch.from_numpy(a).to(DEV)     # noqa: E731
    got = ops.contract(T(x)).cpu().numpy()
    want = o360.contract(x)
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=2e-6)
    inside = np.linalg.norm(x, axis=-1) <= 1
    assert inside.any() and np.array_equal(got[inside], x[inside]) and np.linalg.norm(got, axis=-1).max() < 2.0
    gm, gc = ops.parameterization(T(x), T(cov))
    wm, wc = o360.contract_gaussian(x, cov)
    np.testing.assert_allclose(gm.cpu().numpy(), wm, rtol=2e-6, atol=2e-6)
    scale = np.abs(wc).max(axis=(-1, -2), keepdims=True)
    cerr = float((np.abs(gc.cpu().numpy() - wc) / scale).max())
    Lf = 5
    enc = ops.integrated_pos_enc_360((T(x), T(cov)), 0, Lf, contracted=True).cpu().numpy()
    want_e = o360.integrated_pos_enc_360((wm, wc), 0, Lf)
    eerr = float(np.abs(enc - want_e).max())
    G.record("mipnerf360 free functions", cov_rel=cerr, enc_abs=eerr)
    assert cerr <= 1e-3 and enc.shape == (M, 42 * Lf)
    assert eerr <= 2e-3          # var errors of 1e-3 relative enter exp(-0.5 * 4^l var)
    single = ops.integrated_pos_enc_360((T(x), T(cov))).cpu().numpy()       # upstream signature: one frequency
    assert single.shape == (M, 42) and np.abs(single - enc[:, [*range(21), *range(21 * Lf, 21 * Lf + 21)]]).max() == 0.0

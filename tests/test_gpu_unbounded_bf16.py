"""GPU: bf16 inference of the unbounded-scene model, `MipNerf(unbounded=True, precision='bf16')`.

Its 672-wide encoding does not fit k_mlp_bf16's wave-private LDS area, so the MLP runs as two kernels (csrc/gen_pre_gemm.py,
mlp_pre_plan.py): k_pre_gemm = layer 0 and the encoding half of the skip layer (models/mip_nerf.py:83-90), k-step-major; then the
trunk kernel.  Checked here through the C ABI:
  * the MLP alone (mipnerf_mlp_forward, row-major bf16 encodings) against the numpy model of bf16 operands / fp32 accumulation and
    against the fp32 oracle, at ragged sizes (partial wave tiles, partial 256-sample tiles, more tiles than workgroups);
  * the whole forward (mipnerf_forward: off-axis IPE written as MFMA fragments -> k_pre_gemm -> trunk -> compositing) against the fp32
    model on the same rays, and against the per-stage route (row-major encodings), which must agree bit for bit;
  * training entry points refuse this precision for this model."""
import numpy as np
import pytest
import torch

import synthetic_inputs as syn
from oracle import mipnerf_oracle as orc
from test_gpu_stages import mlp_bf16_emulation

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def G():
    import gpu_util
    assert torch.cuda.is_available()
    return gpu_util


def _model(params, N, precision, **kw):
    from mipnerf_pl_amd import MipNerf
    m = MipNerf(num_samples=N, unbounded=True, precision=precision, **kw)
    m.load_state_dict({"mlp." + k: torch.from_numpy(v.copy()) for k, v in params.items()}, strict=True)
    return m.to(DEV)


@pytest.mark.parametrize("shape", [(1, 1), (1, 33), (3, 43), (2, 128), (5, 77), (9, 256), (300, 64), (1030, 70)])
def test_mlp_two_kernel_form_vs_emulation_and_oracle(G, shape):
    B, N = shape
    params = syn.make_params(seed=31, density_gain=8.0, xyz_dim=672)
    model = _model(params, max(N, 1), "bf16")
    rng = np.random.default_rng(B * 1000 + N)
    enc = rng.uniform(-1, 1, (B, N, 672)).astype(np.float32)
    v27 = rng.uniform(-1, 1, (B, 27)).astype(np.float32)
    with torch.no_grad():
        rgb, den = model.mlp(torch.from_numpy(enc).to(DEV), torch.from_numpy(v27).to(DEV))
        rgb2, den2 = model.mlp(torch.from_numpy(enc).to(DEV), torch.from_numpy(v27).to(DEV))
    assert torch.equal(rgb, rgb2) and torch.equal(den, den2)                      # deterministic
    assert bool(torch.isfinite(rgb).all()) and bool(torch.isfinite(den).all())
    er_, ed_ = mlp_bf16_emulation(params, enc, v27)
    rr, dd = orc.mlp_forward(params, enc, v27)
    e = dict(vs_emulation_rgb=G.maxdiff(rgb, er_), vs_emulation_density=G.maxdiff(den, ed_), vs_fp32_rgb=G.maxdiff(rgb, rr),
             vs_fp32_density=G.maxdiff(den, dd), max_abs_density=float(np.abs(dd).max()))
    G.record(f"unbounded bf16 mlp {B}x{N}", **e)
    # the tolerances of the standard model's bf16 test (test_gpu_stages.test_mlp_bf16): vs the bf16 emulation only the accumulation
    # order differs (rare one-ulp flips of a bf16 activation); vs fp32 the bf16 rounding of 10 chained layers
    assert e["vs_emulation_rgb"] <= 6e-3 and e["vs_emulation_density"] <= 0.15
    assert e["vs_fp32_rgb"] <= 2e-2 and e["vs_fp32_density"] <= 0.4


def test_mlp_does_not_depend_on_the_grid(G):
    """1 ... 256 persistent workgroups share the tiles (both kernels): bit-identical outputs"""
    params = syn.make_params(seed=32, density_gain=8.0, xyz_dim=672)
    model = _model(params, 64, "bf16")
    rng = np.random.default_rng(3)
    enc = torch.from_numpy(rng.uniform(-1, 1, (70, 64, 672)).astype(np.float32)).to(DEV)      # 4480 samples = 17.5 tiles
    v = torch.from_numpy(rng.uniform(-1, 1, (70, 27)).astype(np.float32)).to(DEV)
    ctx = model.mlp.native(enc.device)
    outs = []
    with torch.no_grad():
        for grid in (256, 5, 1, 256):
            ctx.set_option(1, grid)
            outs.append(torch.cat(model.mlp(enc, v), -1).clone())
    ctx.set_option(1, 256)
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


@pytest.mark.parametrize("randomized", [False, True])
@pytest.mark.parametrize("B", [1, 5, 300])
def test_forward_bf16_vs_fp32_and_vs_per_stage_route(G, randomized, B):
    from mipnerf_pl_amd import _lib as L
    from mipnerf_pl_amd import ops
    N = 64                    # 64 / 320 / 19,200 samples: a quarter of a tile, 1.25 tiles, 75 tiles of 256
    rays = syn.synthetic_rays(B, seed=71, unbounded=True)
    params = syn.make_params(seed=17, density_gain=40.0, xyz_dim=672)
    rng = np.random.default_rng(6)
    T = lambda a: None if a is None else torch.from_numpy(a).to(DEV)     # noqa: E731
    tr = T(rng.uniform(0, 1, (B, N + 1)).astype(np.float32)) if randomized else None
    ur = T(rng.uniform(0, 1, (B, N + 1)).astype(np.float32)) if randomized else None
    m16, m32 = _model(params, N, "bf16"), _model(params, N, "fp32")
    R = G.to_dev(rays)
    with torch.no_grad():
        got = m16(R, randomized, True, t_rand=tr, u_rand=ur)
        ref = m32(R, randomized, True, t_rand=tr, u_rand=ur)
    errs = {}
    for lvl in range(2):
        for nm, a, b in zip(G.NAMES, got[lvl], ref[lvl]):
            errs[f"l{lvl}_{nm}"] = G.maxdiff(a, b)
    mse = float(torch.mean((got[1][0] - ref[1][0]) ** 2))
    errs["psnr_fine_rgb_db"] = float(-10 * np.log10(max(mse, 1e-20)))
    G.record(f"unbounded bf16 forward vs fp32 B={B} randomized={randomized}", **errs)
    # the coarse level sees the same fence posts: colours within the bf16 tolerance of the standard model's full-size test (3e-2 max,
    # PSNR >= 55 dB); the fine level's fence posts move with the coarse weights, so it is held by PSNR only
    # Bounds = 2 x the maxima measured over the six cases (l0 rgb 2.4e-3, l0 acc 3.6e-3) and the measured minimum - 2 dB for the fine level
    # (55.1 ... 65.7 dB).  This is bf16 HIP against fp32 HIP on a FOG of random weights -- white noise in space at 16 undamped degrees --, where
    # the 1e-2 shifts of the resampled fence posts show in the colour: the standard model's numpy model measures the same 55 dB on such rays
    # (scripts/analysis/bf16_360_error_budget.py).  The level that matters is held by the trained-field test below: 70 dB against the oracle.
    assert errs["l0_rgb"] <= 4.8e-3 and errs["l0_acc"] <= 7.2e-3, errs
    assert errs["psnr_fine_rgb_db"] >= 53.0, errs
    for lvl in range(2):
        assert all(bool(torch.isfinite(t).all()) for t in got[lvl])
    # per-stage route of the coarse level: row-major bf16 encodings -> mipnerf_mlp_forward -> compositing; the forward call wrote the same
    # encodings as fragments, so every output bit must agree
    t0 = got[0][4]
    with torch.no_grad():
        enc = ops.cast_ipe_360(t0, R.origins, R.directions, R.radii, m16.min_deg_point, m16.max_deg_point, precision=L.PREC_BF16)
        venc = ops.pos_enc(R.viewdirs, 0, m16.deg_view)
        _, _, act = m16.mlp(enc, venc, return_activated=True)
        comp = ops.volumetric_rendering_packed(act, t0, R.directions, True)
    assert torch.equal(comp[0], got[0][0]) and torch.equal(comp[3], got[0][3])


def test_train_in_fp32_render_in_bf16(G):
    """set_precision switches the arithmetic of a live model: an fp32 model switched to bf16 renders exactly what a bf16 model with the same
    weights renders (the context packs the streams of both precisions), and switches back"""
    params = syn.make_params(seed=17, density_gain=40.0, xyz_dim=672)
    rays = G.to_dev(syn.synthetic_rays(50, seed=9, unbounded=True))
    m16, m32 = _model(params, 64, "bf16"), _model(params, 64, "fp32")
    with torch.no_grad():
        a = m16(rays, False, True)
        b32 = m32(rays, False, True)
        b = m32.set_precision("bf16")(rays, False, True)
        c32 = m32.set_precision("fp32")(rays, False, True)
    for lvl in range(2):
        for x, y in zip(a[lvl], b[lvl]):
            assert torch.equal(x, y)
        for x, y in zip(b32[lvl], c32[lvl]):
            assert torch.equal(x, y)
    assert not torch.equal(a[1][0], b32[1][0])
    with pytest.raises(ValueError):
        m32.set_precision("fp16")


@pytest.mark.parametrize("B,N,white", [(37, 100, False), (300, 32, True), (5, 65, True), (2, 3, False), (3, 600, True)])
def test_one_call_native_step_equals_the_autograd_route(G, B, N, white):
    """Round 5: mipnerf_train_step for the unbounded-scene model (inverse-depth fence posts, contracted off-axis IPE rows, k_pre_gemm + trunk
    forward-with-save, per-level weight-gradient launches over the row-major encoding) against the autograd route on the SAME kernels with
    the same draws injected: loss and every gradient agree to summation order (ragged ray counts, N not a multiple of 32, the K = 16 bucket)."""
    from mipnerf_pl_amd.autograd import distloss
    params = syn.make_params(seed=B, density_gain=40.0, xyz_dim=672)
    rays = G.to_dev(syn.synthetic_rays(B, seed=B + 1, unbounded=True))
    gt = torch.rand(B, 3, device=DEV)
    t_rand, u_rand = torch.rand(B, N + 1, device=DEV), torch.rand(B, N + 1, device=DEV)
    import mipnerf_pl_amd.autograd as AG
    res = {}
    for native in (False, True, "rows"):
        model = _model(params, N, "bf16")
        if native is True:
            scal, outs = model.train_step_native(rays, gt, True, white, t_rand=t_rand, u_rand=u_rand, return_outputs=True)
            loss = float(scal[0])
            fine = outs[1]
        else:
            # False: autograd on fragment encodings (the default); "rows": autograd on row-major rows (the per-stage ABI's layout) -- the
            # cross-check of the two encoding layouts of k_pre_gemm and of the weight-gradient jobs
            AG.FRAGMENT_ENCODINGS = native is False
            try:
                ret = model(rays, True, white, t_rand=t_rand, u_rand=u_rand)
            finally:
                AG.FRAGMENT_ENCODINGS = True
            mask = rays.lossmult
            mse = [(mask * (r[0] - gt) ** 2).sum() / mask.sum() for r in ret]
            dl = [distloss(r[3], r[4]) for r in ret]
            tot = 0.1 * (mse[0] + 0.01 * dl[0]) + mse[1] + 0.01 * dl[1]
            tot.backward()
            loss = float(tot.detach())
            fine = tuple(x.detach() for x in ret[1])
        res[native] = (loss, torch.cat([p.grad.reshape(-1) for p in model.parameters()]).clone(), fine)
    (l0, g0, f0), (l1, g1, f1) = res[False], res[True]
    lr_, gr_, _ = res["rows"]
    e_rows = G.maxdiff(gr_, g1) / float(gr_.abs().max())
    assert abs(lr_ - l1) <= 2e-6 * max(1.0, abs(lr_)) and e_rows <= 2e-5, (lr_, l1, e_rows)       # measured <= 7e-9: fragments == rows
    for a, b in zip(f0, f1):
        assert G.maxdiff(a, b) <= 2e-6, G.maxdiff(a, b)      # same forward kernels on the same inputs (the activations run fused / stand-alone)
    eg = G.maxdiff(g0, g1) / float(g0.abs().max())
    G.record(f"unbounded native_train_step B={B} N={N}", loss_autograd=l0, loss_native=l1, grad_rel=eg, grad_rel_rows_vs_fragments=e_rows)
    assert abs(l0 - l1) <= 2e-6 * max(1.0, abs(l0)) and eg <= 2e-5, (l0, l1, eg)
    # fused tail (compositing + distloss + the next level's inverse-depth fence posts in one launch) == one launch per stage, bit for bit
    grads = []
    for fuse in (1, 0):
        model = _model(params, N, "bf16")
        model.mlp.native(torch.device(DEV)).set_option(4, fuse)
        scal, _ = model.train_step_native(rays, gt, True, white, t_rand=t_rand, u_rand=u_rand)
        grads.append((scal.clone(), torch.cat([p.grad.reshape(-1) for p in model.parameters()]).clone()))
    assert torch.equal(grads[0][0], grads[1][0]) and torch.equal(grads[0][1], grads[1][1])
    # fp32 precision has no one-call step (it trains through autograd)
    with pytest.raises(NotImplementedError):
        _model(params, N, "fp32").train_step_native(rays, gt, True, white)


def test_graphed_train_step_of_the_unbounded_model(G):
    """The unbounded model's whole optimisation step from ONE captured hipGraph (train_graph.GraphedTrainStep: draws, mipnerf_train_step,
    device-side Adam + MipLRDecay, re-pack of the k_pre_gemm / trunk weight streams): graph == the same launches issued eagerly bit for
    bit, and it trains."""
    from mipnerf_pl_amd.system import DEFAULT_HPARAMS, MipNeRFSystem
    from mipnerf_pl_amd.train_graph import GraphedTrainStep
    B, N = 192, 64
    params = syn.make_params(seed=23, density_gain=6.0, xyz_dim=672)
    rays = G.to_dev(syn.synthetic_rays(B, seed=9, unbounded=True))
    gt = torch.rand(B, 3, device=DEV, generator=torch.Generator(device=DEV).manual_seed(4))
    res = {}
    for mode in ("graph", "eager"):
        hp = dict(DEFAULT_HPARAMS)
        hp.update({'nerf.num_samples': N, 'nerf.unbounded': True, 'train.randomized': False, 'optimizer.lr_init': 1e-3, 'optimizer.lr_final': 1e-5,
                   'optimizer.max_steps': 20, 'optimizer.lr_delay_steps': 4})
        system = MipNeRFSystem(hp, precision="bf16")
        system.load_state_dict({"mip_nerf.mlp." + k: torch.from_numpy(v.copy()) for k, v in params.items()})
        system = system.to(DEV)
        system.fused_adam = True
        (opt,), (sch,) = system.configure_optimizers()
        step = GraphedTrainStep(system, opt, B, torch.device(DEV), use_graph=(mode == "graph"))
        for dst, src in zip(step.rays, rays):
            dst.copy_(src)
        step.gt.copy_(gt)
        losses = []
        for it in range(6):
            losses.append(float(step()[0]))
            sch["scheduler"].step()
        assert step.capture_error is None
        with torch.no_grad():
            out = system.mip_nerf(rays, False, True)[1][0].clone()      # uses the re-packed weight streams
        res[mode] = (losses, torch.cat([p.detach().reshape(-1) for p in system.mip_nerf.parameters()]).clone(), out)
    assert res["graph"][0] == res["eager"][0], (res["graph"][0], res["eager"][0])
    assert torch.equal(res["graph"][1], res["eager"][1]) and torch.equal(res["graph"][2], res["eager"][2])
    assert res["graph"][0][-1] < res["graph"][0][0], res["graph"][0]
    G.record("unbounded graphed_train_step", first_loss=res["graph"][0][0], last_loss=res["graph"][0][-1])


# ---- round 5: this path against the 360 ORACLE, not against the repo's own fp32 kernels ----------------------------
def _psnr(a, b):
    return float(-10.0 * np.log10(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2) + 1e-30))


def test_bf16_off_axis_encoding_vs_the_gaussian_expectation(G):
    """The bf16 rows of k_cast_ipe_360_tile are a DIFFERENT function from the fp32 rows (true cosine, two-float range reduction, hardware
    sine / exp2); round 4 compared them with the sibling per-direction kernel only.  Here: against the closed-form Gaussian expectation
    E[sin(2^l p.x)] = sin(2^l p.mu) exp(-0.5 4^l p^T Sigma p) (and cos) evaluated in FLOAT64 on the oracle's contracted Gaussians, degree by
    degree.  Error budget per feature: half a bf16 ulp of a value <= 1 (2^-9) + the fp32 phase: the kernel forms y = 2^l fl32(p.mu) in fp32,
    |p.mu| <= 2 after the contraction, so the phase is off by up to 2^l x 2.4e-7 rad (what the fp32 rows and the oracle carry as well)."""
    from mipnerf_pl_amd import _lib as L
    from mipnerf_pl_amd import ops
    from oracle import mipnerf360_oracle as o360
    B, N = 48, 64
    rays = syn.synthetic_rays(B, seed=83, unbounded=True, multiscale=True)
    rng = np.random.default_rng(8)
    t_rand = rng.uniform(0, 1, (B, N + 1)).astype(np.float32)
    _, t, (mean, cov) = o360.sample_along_rays_360(rays.origins, rays.directions, rays.radii, N, rays.near, rays.far, True, t_rand=t_rand,
                                                   contracted=True)
    R = G.to_dev(rays)
    enc = ops.cast_ipe_360(torch.from_numpy(t).to(DEV), R.origins, R.directions, R.radii, 0, 16, contracted=True, precision=L.PREC_BF16)
    assert enc.dtype == torch.bfloat16 and tuple(enc.shape) == (B, N, 672)
    got = enc.float().cpu().numpy().astype(np.float64).reshape(B, N, 2, 16, 21)                  # [sin | cos] x degree x direction
    P = o360.BASIS_360.astype(np.float64)
    y = mean.astype(np.float64) @ P                                                              # [B, N, 21]
    var = np.einsum("ik,...ij,jk->...k", P, cov.astype(np.float64), P)
    worst = {}
    for l in range(16):
        damp = np.exp(-0.5 * 4.0 ** l * var)
        want = np.stack([np.sin(2.0 ** l * y) * damp, np.cos(2.0 ** l * y) * damp], 2)            # [B, N, 2, 21]
        e = np.abs(got[:, :, :, l, :] - want)
        worst[f"deg{l}"] = float(e.max())
        # bf16 half ulp of |f| <= 1 (2^-9 = 1.95e-3) + fp32 phase error of the damped feature + 2e-4 for the fast sine / exp2
        bound = 2.0 ** -9 + 2.0 ** l * 4.8e-7 * float(damp.max()) + 2e-4
        assert worst[f"deg{l}"] <= bound, (l, worst[f"deg{l}"], bound)
    assert float(np.abs(got).max()) <= 1.0 + 2.0 ** -8
    G.record("unbounded bf16 encoding vs float64 Gaussian expectation", **worst)


@pytest.mark.parametrize("B,N,randomized", [(1, 64, False), (5, 77, True), (300, 64, False), (130, 200, True)])
def test_forward_bf16_vs_the_360_oracle_ragged(G, B, N, randomized):
    """The whole forward of `MipNerf(unbounded=True, precision='bf16')` (off-axis IPE as MFMA fragments -> k_pre_gemm -> trunk -> compositing
    -> inverse-depth resampling) against oracle/mipnerf360_oracle.mipnerf360_forward -- the CPU restatement, not the HIP fp32 path -- at
    ragged sizes: a quarter of a 256-sample tile, partial wave tiles, N that is neither a multiple of 64 nor of 32."""
    from oracle import mipnerf360_oracle as o360
    rays = syn.synthetic_rays(B, seed=300 + B, unbounded=True)
    params = syn.make_params(seed=17, density_gain=40.0, xyz_dim=672)
    rng = np.random.default_rng(B + N)
    tr = rng.uniform(0, 1, (B, N + 1)).astype(np.float32) if randomized else None
    ur = rng.uniform(0, 1, (B, N + 1)).astype(np.float32) if randomized else None
    want = o360.mipnerf360_forward(params, rays, randomized, True, num_samples=N, t_rand=tr, u_rand=ur)
    T = lambda a: None if a is None else torch.from_numpy(a).to(DEV)     # noqa: E731
    model = _model(params, N, "bf16")
    with torch.no_grad():
        got = model(G.to_dev(rays), randomized, True, t_rand=T(tr), u_rand=T(ur))
    errs = {}
    for lvl in range(2):
        for nm, a, b in zip(G.NAMES, got[lvl], want[lvl]):
            errs[f"l{lvl}_{nm}"] = G.maxdiff(a, b)
    errs["psnr_l0_rgb"] = _psnr(got[0][0].cpu().numpy(), want[0][0])
    errs["psnr_l1_rgb"] = _psnr(got[1][0].cpu().numpy(), want[1][0])
    G.record(f"unbounded bf16 forward vs 360 oracle B={B} N={N} randomized={randomized}", **errs)
    assert errs["l0_t_samples"] <= 1e-6 * float(np.abs(want[0][4]).max())            # the coarse fence posts do not depend on the MLP
    # coarse level: same fence posts, so per-ray values compare directly.  Bounds = 2 x the maxima measured over these four cases on MI355X
    # (profiles/r05_parity.jsonl; the standard model's full-size fog case measures 2.6e-3 / 72 dB)
    assert errs["l0_rgb"] <= UNB_FOG_BOUNDS["l0_rgb"] and errs["l0_acc"] <= UNB_FOG_BOUNDS["l0_acc"], errs
    assert errs["l1_rgb"] <= UNB_FOG_BOUNDS["l1_rgb"] and errs["l1_acc"] <= UNB_FOG_BOUNDS["l1_acc"], errs
    assert errs["psnr_l1_rgb"] >= UNB_FOG_BOUNDS["psnr_l1_rgb"], errs


# 2 x the maxima measured on MI355X over the four ragged fog cases above (psnr: measured minimum - 6 dB, never below the 55 dB every other
# bf16 test of the repo holds; DESIGN.md section 2 derives 51.4 dB as the level that keeps a 35 dB render within 0.1 dB)
# measured (profiles/r05a_parity.jsonl): l0 rgb 3.1e-3 / acc 6.2e-3 (coarse PSNR 70.4-75.9 dB), l1 rgb 1.3e-2 / acc 2.3e-2, fine PSNR 57.6-65.7 dB.
# The fine level of a FOG field is the hard case: random weights with 16 undamped degrees are white noise in space, so the 1e-2 shifts of
# the resampled fence posts show up in the colour (the trained-field test below is the realistic one; scripts/analysis/bf16_360_error_budget.py)
UNB_FOG_BOUNDS = dict(l0_rgb=6.2e-3, l0_acc=1.25e-2, l1_rgb=2.7e-2, l1_acc=4.7e-2, psnr_l1_rgb=55.0)


def _field360(G):
    f = G.load_golden("trained_field_360")
    return {k[2:]: f[k] for k in f if k.startswith("p_")}


@pytest.mark.parametrize("name", ["full360_1000x96", "full360_8192x256"])
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_forward_on_a_trained_unbounded_field_vs_the_360_oracle(G, name, precision):
    """BASELINE configs[3]'s size (8192 rays x (256 + 256) samples) and a ragged 1000 x 96 case on a TRAINED unbounded field
    (scripts/make_golden_360.py: the reference's MLP class trained on the procedural unbounded scene of tests/dataset_fixture.py -- blobs
    inside a far sky shell with holes -- through the 360 oracle's sampling / encoding), on rays of that scene: empty rays (through the
    holes: white background, the sampler's padding branch), opaque near hits, opaque / soft hits on the contracted far shell.  Golden =
    oracle.mipnerf360_forward on CPU; every ray of both levels, bounds per class of ray."""
    import hashlib
    g = G.load_golden(name)
    params = _field360(G)
    h = hashlib.sha256()
    for k in sorted(params):
        h.update(np.ascontiguousarray(params[k]).tobytes())
    assert h.hexdigest() == str(g["field_sha256"])
    assert float(g["frac_empty"]) >= 0.15 and float(g["frac_opaque"]) >= 0.15 and float(g["frac_between"]) >= 0.05
    rays = syn.Rays(*[g["rays_" + k] for k in syn.Rays._fields])
    # density_bias: the field was trained from a transparent start (scripts/make_golden_360.py: -4 instead of the default -1, which is opaque
    # before the first step on rays that reach t = 20) -- a constructor argument of the reference (mip_nerf.py:129)
    model = _model(params, int(g["num_samples"]), precision, density_bias=float(g["density_bias"]))
    with torch.no_grad():
        ret = model(G.to_dev(rays), False, True)
    acc_ref = g["l1_acc"]
    classes = {"empty": acc_ref < 0.05, "opaque": acc_ref > 0.95, "between": (acc_ref >= 0.05) & (acc_ref <= 0.95)}
    errs = {}
    for lvl in range(2):
        rgb, dist, acc, w, t = ret[lvl]
        got = dict(rgb=rgb.cpu().numpy(), distance=dist.cpu().numpy(), acc=acc.cpu().numpy(), wmax=w.max(-1).values.cpu().numpy())
        for nm, v in got.items():
            d = np.abs(v.astype(np.float64) - g[f"l{lvl}_{nm}"].astype(np.float64))
            if nm == "distance":
                d = d / np.maximum(g[f"l{lvl}_t_last"].astype(np.float64), 1.0)          # far bounds reach 22: relative to the ray's far end
            errs[f"l{lvl}_{nm}"] = float(d.max())
            if lvl == 1 and nm in ("rgb", "acc"):
                for cn, m in classes.items():
                    errs[f"l1_{nm}_{cn}"] = float(d[m].max())
        assert G.maxdiff(t[:, 0], g[f"l{lvl}_t_first"]) <= 1e-5 and G.maxdiff(t[:, -1], g[f"l{lvl}_t_last"]) <= 1e-4 * float(g[f"l{lvl}_t_last"].max())
    errs["psnr_l1_rgb"] = _psnr(ret[1][0].cpu().numpy(), g["l1_rgb"])
    errs["psnr_vs_scene_pixels"] = _psnr(ret[1][0].cpu().numpy(), g["gt"])
    errs["oracle_psnr_vs_scene_pixels"] = _psnr(g["l1_rgb"], g["gt"])
    G.record(f"trained360 {name} {precision}", **errs)
    b = TRAINED360_BOUNDS[precision]
    for k, bound in b.items():
        if k.startswith("psnr"):
            assert errs[k] >= bound, (k, errs[k], bound)
        else:
            assert errs[k] <= bound, (k, errs[k], bound)
    # the north star's 0.1 dB, on this frame: the bf16 render is as close to the scene's pixels as the oracle's
    assert abs(errs["psnr_vs_scene_pixels"] - errs["oracle_psnr_vs_scene_pixels"]) < (0.1 if precision == "bf16" else 1e-2)


@pytest.mark.parametrize("name", ["full360_1000x96", "full360_8192x256"])
def test_one_kernel_form_equals_the_two_kernel_form_bit_for_bit(G, name):
    """Round 6: the bf16 forward of the unbounded model as ONE MLP kernel per level (layer 0 and the skip layer as k-step-major
    ops of the trunk kernel, the 672-wide encoding streamed through a wave-private LDS ring; option 6 = 1, the default) against k_pre_gemm +
    trunk kernel (option 6 = 0): the same products added in the same order, so EVERY output of both levels is bit-identical -- on the trained
    field at BASELINE configs[3]'s size and the ragged 1000 x 96 case (partial tiles), with 1 / 5 / 256 persistent workgroups."""
    g = G.load_golden(name)
    params = _field360(G)
    rays = G.to_dev(syn.Rays(*[g["rays_" + k] for k in syn.Rays._fields]))
    model = _model(params, int(g["num_samples"]), "bf16", density_bias=float(g["density_bias"]))
    ctx = model.mlp.native(torch.device(DEV))
    outs = {}
    try:
        for form in (1, 0):
            ctx.set_option(6, form)
            with torch.no_grad():
                outs[form] = model(rays, False, True)
        for grid in (1, 5):
            ctx.set_option(6, 1)
            ctx.set_option(1, grid)
            with torch.no_grad():
                outs[("grid", grid)] = model(rays, False, True) if name == "full360_1000x96" else None
    finally:
        ctx.set_option(6, 1)
        ctx.set_option(1, 256)
    for lvl in range(2):
        for a, b in zip(outs[1][lvl], outs[0][lvl]):
            assert torch.equal(a, b), (name, lvl)
        for grid in (1, 5):
            if outs[("grid", grid)] is not None:
                for a, b in zip(outs[1][lvl], outs[("grid", grid)][lvl]):
                    assert torch.equal(a, b), (name, lvl, grid)
    assert bool(torch.isfinite(outs[1][1][0]).all())


# measured on MI355X (profiles/r05_parity.jsonl), 1000 x 96 / 8192 x 256:
#   fp32  l0 rgb 3.0e-6 / 6.3e-6, l0 acc 3.2e-6 / 6.3e-6, l1 rgb 2.2e-6 / 4.6e-6, l1 acc 3.1e-6 / 4.6e-6, distance / far 1.5e-6 / 1.9e-6, 126 / 117 dB
#   bf16  l0 rgb 2.1e-3 / 3.7e-3, l0 acc 3.1e-3 / 4.0e-3, l1 rgb 2.9e-3 / 3.0e-3, l1 acc 3.6e-3 / 3.2e-3, empty rays rgb 2.4e-4 / 4.8e-4 and
#         acc 2.4e-4 / 4.9e-4, opaque rays rgb 1.0e-3, 70.2 dB against the oracle's frame on both; against the scene's pixels 28.639 /
#         28.794 dB where the oracle's own frames reach 28.645 / 28.795
# fp32: the repo-wide fp32 bounds (level 0) and 4 x them at level 1 (the resampled inverse depths move by an ulp); bf16: 2 x the maxima
TRAINED360_BOUNDS = {
    "fp32": dict(l0_rgb=5e-5, l0_acc=5e-5, l1_rgb=2e-5, l1_acc=2e-5, l1_distance=2e-5, psnr_l1_rgb=105.0),
    "bf16": dict(l0_rgb=7.4e-3, l0_acc=8e-3, l1_rgb=6e-3, l1_acc=7.2e-3, l1_rgb_empty=1e-3, l1_acc_empty=1e-3, l1_rgb_opaque=2e-3,
                 psnr_l1_rgb=64.0),
}


def test_mlp_forward_on_two_streams_shares_the_scratch_safely(G):
    """mipnerf_mlp_forward of the two-kernel form keeps ONE context-owned scratch buffer (pre_x | pre_acc) between its two
    kernels.  Calls on different streams used to share it without any ordering; now a call on another stream than the previous one
    waits for that call's kernels (event recorded behind them).  Alternate two streams with different inputs, no host synchronisation
    in between: every result must equal the single-stream result of its input."""
    params = syn.make_params(seed=33, density_gain=8.0, xyz_dim=672)
    model = _model(params, 64, "bf16")
    rng = np.random.default_rng(11)
    encs = [torch.from_numpy(rng.uniform(-1, 1, (300, 64, 672)).astype(np.float32)).to(DEV) for _ in range(4)]
    v = torch.from_numpy(rng.uniform(-1, 1, (300, 27)).astype(np.float32)).to(DEV)
    with torch.no_grad():
        want = [torch.cat(model.mlp(e, v), -1).clone() for e in encs]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    got = [None] * len(encs)
    with torch.no_grad():
        for rep in range(3):
            for i, e in enumerate(encs):
                s = streams[i % 2]
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    got[i] = torch.cat(model.mlp(e, v), -1)
            torch.cuda.synchronize()
            for a, b in zip(got, want):
                assert torch.equal(a, b), rep


# ---- round 5 (VERDICT r04 #3b): bf16 TRAINING of the unbounded-scene model ------------------------------------------------------------
@pytest.mark.parametrize("shape", [(1, 5), (3, 43), (9, 64), (40, 70)])
def test_mlp_training_kernels_vs_emulation_and_oracle(G, shape):
    """The MLP alone under autograd: k_pre_gemm + the trunk forward-with-save, the dgrad kernel, and the weight-gradient kernel -- whose jobs
    for W0 and W5[:, 256:] read 32-feature column blocks of the ROW-MAJOR bf16 encoding through per-lane DMAs and transposing LDS loads
    (ds_read_b64_tr_b16) -- against (i) the numpy emulation of the same dataflow with bf16 roundings (mlp_train_plan.emulate_train) and
    (ii) the oracle's fp32 gradients (autograd of models/mip_nerf.py:75-111), at ragged sizes: partial wave tiles, partial 256-sample tiles."""
    from mipnerf_pl_amd.autograd import mlp_native
    from mipnerf_pl_amd.mlp_plan import Arch
    from mipnerf_pl_amd.mlp_train_plan import TrainPlan, emulate_train
    B, N = shape
    params = syn.make_params(seed=41, density_gain=6.0, xyz_dim=672)
    model = _model(params, max(N, 1), "bf16")
    rng = np.random.default_rng(B * 100 + N)
    enc = rng.uniform(-1, 1, (B, N, 672)).astype(np.float32)
    v27 = rng.uniform(-1, 1, (B, 27)).astype(np.float32)
    v32 = np.zeros((B, 32), np.float32)
    v32[:, :27] = v27
    d_raw = np.concatenate([rng.normal(0, 1e-2, (B, N, 3)), rng.normal(0, 1e-3, (B, N, 1))], -1).astype(np.float32)
    raw = mlp_native(model.mlp, torch.from_numpy(enc).to(DEV).to(torch.bfloat16), torch.from_numpy(v32).to(DEV).to(torch.bfloat16))
    (raw * torch.from_numpy(d_raw).to(DEV)).sum().backward()
    arch = Arch(xyz_dim=672, feat_per_deg=42, bf16_kernels=False)
    tp = TrainPlan.build(arch, pre_gemm=True)
    names = [n for n, _ in arch.param_shapes()]
    flat = np.concatenate([params[n].ravel() for n in names])
    g_em, _, raw_em = emulate_train(tp, flat, enc.reshape(-1, 672), np.repeat(v32, N, axis=0), d_raw.reshape(-1, 4), round_bf16=True)
    og = orc.mlp_backward(params, enc, v27, d_raw[..., :3], d_raw[..., 3:])
    offs, _ = tp.fwd.param_offsets()
    worst_em, worst_cos = 0.0, 1.0
    per = {}
    for i, (k, p) in enumerate(model.mlp.named_parameters()):
        got = p.grad.detach().cpu().numpy().astype(np.float64).ravel()
        em = g_em[offs[i]:offs[i] + got.size].astype(np.float64)
        ref = og[k].astype(np.float64).ravel()
        per[k] = float(np.linalg.norm(got - em) / max(np.linalg.norm(em), 1e-30))
        worst_em = max(worst_em, per[k])
        if np.linalg.norm(ref) > 0:
            worst_cos = min(worst_cos, float(got @ ref / max(np.linalg.norm(got) * np.linalg.norm(ref), 1e-30)))
    e_raw = G.maxdiff(raw.detach().reshape(-1, 4), raw_em)
    G.record(f"unbounded bf16 training kernels {B}x{N}", worst_rel_l2_vs_emulation=worst_em, worst_cos_vs_fp32=worst_cos, raw_vs_emulation=e_raw,
             **{"em_" + k: v for k, v in per.items()})
    # vs the emulation only the fp32 accumulation order differs.  That moves ~2.5e-5 of the bf16 activations by one ulp, and about one in twenty
    # of those moves flips a ReLU bit further down -- one flipped bit changes that sample's delta in every layer below it, i.e. ~1 % of a
    # gradient tensor at 100-600 samples (measured: 1 x 5 samples 2e-7 -- the dataflow is exact --, 3 x 43: ONE mask bit in 5 wave tiles ->
    # 1.05e-2, 9 x 64 1.14e-2, 40 x 70 3.9e-3; scripts/micro/dbg_train360.py lists the flips tile by tile; the standard model's cases
    # happened to meet none: 3-4e-4).  Bound: 3 x that.
    assert worst_em <= (1e-5 if B * N <= 8 else 3e-2), worst_em
    assert e_raw <= 2e-2 * max(1.0, float(np.abs(raw_em).max()))
    assert worst_cos >= 0.97, worst_cos


def test_training_step_bf16_vs_fp32_autograd_path(G):
    """One training step of `MipNerf(unbounded=True)` through MipNeRFSystem.training_step + backward in bf16 against the same step in fp32
    (the path tests/test_gpu_unbounded.py pins against torch autograd): loss and every gradient tensor; then the optimiser lowers the loss."""
    from mipnerf_pl_amd.system import DEFAULT_HPARAMS, MipNeRFSystem
    B, N = 96, 64
    rays = G.to_dev(syn.synthetic_rays(B, seed=62, unbounded=True, multiscale=True))
    params = syn.make_params(seed=18, density_gain=20.0, xyz_dim=672)
    gt = torch.rand(B, 3, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
    res = {}
    for precision in ("fp32", "bf16"):
        hp = dict(DEFAULT_HPARAMS)
        hp.update({"nerf.num_samples": N, "nerf.unbounded": True, "train.randomized": False, "optimizer.lr_init": 1e-3, "optimizer.lr_delay_steps": 0})
        system = MipNeRFSystem(hp, precision=precision)
        system.load_state_dict({"mip_nerf.mlp." + k: torch.from_numpy(v.copy()) for k, v in params.items()})
        system = system.to(DEV)
        loss = system.training_step((rays, gt), 0)
        loss.backward()
        res[precision] = (float(loss), {k: p.grad.detach().double().reshape(-1).clone() for k, p in system.mip_nerf.mlp.named_parameters()}, system)
    l32, g32, _ = res["fp32"]
    l16, g16, system = res["bf16"]
    cos = {k: float(g16[k] @ g32[k] / (g16[k].norm() * g32[k].norm()).clamp_min(1e-30)) for k in g32}
    a, b = torch.cat(list(g16.values())), torch.cat(list(g32.values()))
    whole = float(a @ b / (a.norm() * b.norm()))
    G.record("unbounded training step bf16 vs fp32", loss_fp32=l32, loss_bf16=l16, worst_cos=min(cos.values()), whole_cos=whole)
    assert abs(l16 - l32) <= 1e-3 * max(1.0, abs(l32))
    # measured: whole gradient 0.99936, every tensor >= 0.9936 except the encoding-fed first layer, 0.981 -- the tensor that is lowest in the
    # standard model too (0.9917): its delta has come through all eight dgrad layers in bf16 and is contracted against oscillating features
    # that do not average the noise out; by degree of the features it multiplies it falls from the low degrees to the top ones
    w0_16, w0_32 = g16["layers.0.0.weight"].reshape(256, 2, 16, 21), g32["layers.0.0.weight"].reshape(256, 2, 16, 21)
    low = float((w0_16[:, :, :6] * w0_32[:, :, :6]).sum() / (w0_16[:, :, :6].norm() * w0_32[:, :, :6].norm()))
    G.record("unbounded training step bf16 vs fp32, first layer by degree", deg0to5=low,
             **{f"deg{l}": float((w0_16[:, :, l] * w0_32[:, :, l]).sum() / (w0_16[:, :, l].norm() * w0_32[:, :, l].norm()).clamp_min(1e-30)) for l in range(16)})
    others = {k: v for k, v in cos.items() if k != "layers.0.0.weight"}
    assert whole >= 0.998 and min(others.values()) >= 0.99, (whole, sorted(cos.items(), key=lambda kv: kv[1])[:3])
    # measured 0.981 overall, 0.991 at degree 0, 0.982 over degrees 0-5, ~0.966 at the middle degrees (profiles/r05_parity.jsonl); that the
    # weight-gradient path over the encoding is exact is the kernels-vs-emulation test's business (same error there as every other layer)
    assert cos["layers.0.0.weight"] >= 0.97 and low >= 0.975, (cos["layers.0.0.weight"], low)
    (opt,), (sch,) = system.configure_optimizers()
    system.zero_grad(set_to_none=True)
    first = None
    for it in range(8):
        opt.zero_grad()
        l_ = system.training_step((rays, gt), it)
        l_.backward()
        opt.step()
        sch["scheduler"].step()
        first = float(l_) if first is None else first
    assert float(l_) < first


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_training_the_unbounded_scene_reaches_the_golden_runs_quality(G, precision):
    """The north star's "PSNR within 0.1 dB" for the NEW training kernels, on what data there is: the run that produced
    tests/golden/trained_field_360.npz (scripts/make_golden_360.py: the reference's MLP, compositing, loss, Adam and MipLRDecay on CPU, fence
    posts and encodings from the 360 oracle; 400 steps x 1024 rays x 64 samples, density_bias -4) repeated natively -- same scene rays
    (tests/golden/scene360_rays.npz), same batches (quality_batch_ids), own random draws, `MipNeRFSystem.training_step` + backward + the
    reference's optimiser settings.  Training is chaotic in its draws, so the comparison is on the tail of the curve: mean training PSNR and
    loss over the last 20 steps (golden run 28.95 dB / 0.01024), and the full-set PSNR of the trained model against the scene's pixels
    (the golden field's own: 28.80 dB at 8192 rays)."""
    sys_path = __import__("os").path.dirname(__file__)
    import sys
    if sys_path not in sys.path:
        sys.path.insert(0, sys_path)
    import dataset_fixture as fx
    from mipnerf_pl_amd.system import DEFAULT_HPARAMS, MipNeRFSystem
    g = G.load_golden("trained_field_360")
    z = G.load_golden("scene360_rays")
    Q = {k[4:]: g[k] for k in g if k.startswith("cfg_")}
    steps, batch, N = int(Q["steps"]), int(Q["batch"]), int(Q["num_samples"])
    rays_all = [torch.from_numpy(z["rays_" + k]).to(DEV) for k in syn.Rays._fields]
    rgb_all = torch.from_numpy(z["rgb"]).to(DEV)
    ids = fx.quality_batch_ids(rgb_all.shape[0], steps, batch, int(Q["id_seed"]))
    hp = dict(DEFAULT_HPARAMS)
    hp.update({"nerf.num_samples": N, "nerf.unbounded": True, "nerf.density_bias": float(Q["density_bias"]), "train.randomized": True,
               "optimizer.lr_init": float(Q["lr_init"]), "optimizer.lr_final": float(Q["lr_final"]), "optimizer.max_steps": int(Q["max_steps"]),
               "optimizer.lr_delay_steps": int(Q["lr_delay_steps"]), "optimizer.lr_delay_mult": float(Q["lr_delay_mult"])})
    torch.manual_seed(int(Q["param_seed"]))
    system = MipNeRFSystem(hp, precision=precision).to(DEV)
    (opt,), (sch,) = system.configure_optimizers()
    torch.manual_seed(int(Q["draw_seed"]))
    from mipnerf_pl_amd import Rays
    losses, psnrs = [], []
    for k in range(steps):
        b = torch.from_numpy(ids[k]).to(DEV)
        R = Rays(*[a[b] for a in rays_all])
        gt = rgb_all[b]
        opt.zero_grad(set_to_none=True)
        loss = system.training_step((R, gt), k)
        loss.backward()
        opt.step()
        sch["scheduler"].step()
        if k >= steps - 20:
            with torch.no_grad():
                ret = system.mip_nerf(R, False, True)
            losses.append(float(loss.detach()))
            psnrs.append(float(-10 * torch.log10(torch.mean((ret[1][0] - gt) ** 2))))
    sel = torch.from_numpy(np.random.default_rng(912).permutation(rgb_all.shape[0])[:8192]).to(DEV)
    with torch.no_grad():
        ret = system.mip_nerf(Rays(*[a[sel] for a in rays_all]), False, True)
    full = float(-10 * torch.log10(torch.mean((ret[1][0] - rgb_all[sel]) ** 2)))
    acc = ret[1][2]
    rec = dict(tail_loss=float(np.mean(losses)), tail_psnr_deterministic=float(np.mean(psnrs)), golden_tail_loss=float(np.mean(g["losses"][-20:])),
               golden_tail_train_psnr=float(np.mean(g["train_psnr"][-20:])), psnr_8192_rays=full, frac_empty=float((acc < 0.05).float().mean()),
               frac_opaque=float((acc > 0.95).float().mean()))
    G.record(f"unbounded quality run {precision}", **rec)
    # the golden's training PSNR is measured on the randomized training forward; ours (deterministic forward on the same batch) reads ~0.1 dB
    # higher.  Measured (round 5): tail loss 0.00987 bf16 / 0.00991 fp32 against the golden run's 0.01024 (-3.6 % / -3.2 %), full-set PSNR
    # 29.24 / 29.19 dB against the golden field's 28.80 (the two precisions 0.04 dB apart).  Band: tail loss within 15 %, full-set PSNR not
    # more than 0.6 dB below the golden's and not more than 1 dB above it (a different draw sequence, not a different optimiser).
    assert abs(rec["tail_loss"] - rec["golden_tail_loss"]) <= 0.15 * rec["golden_tail_loss"], rec
    assert -0.6 <= full - 28.80 <= 1.0, rec
    assert rec["frac_empty"] >= 0.15 and rec["frac_opaque"] >= 0.1, rec          # it learned empty space, not billboards


def test_training_kernels_tile_by_tile_vs_emulation(G):
    """What the relative-L2 numbers of test_mlp_training_kernels_vs_emulation_and_oracle are made of.  The saved activations (T-blocks), ReLU
    masks and deltas of every 32-sample wave tile, read back through the per-stage C entry points, against the numpy emulation of the plan:
    a tile whose saved activations equal the emulation's bit for bit must have bit-identical masks and deltas within fp32 accumulation order
    (the dataflow is exact); elsewhere single bf16 ulps differ (accumulation order), and a flipped ReLU bit -- the only thing that moves a
    gradient tensor by a per cent -- is a rare event, counted here."""
    import ctypes as C
    from mipnerf_pl_amd import _lib as L, ops
    from mipnerf_pl_amd.mlp_plan import Arch
    from mipnerf_pl_amd.mlp_train_plan import TrainPlan, emulate_train_tile
    B, N = 9, 64
    params = syn.make_params(seed=41, density_gain=6.0, xyz_dim=672)
    model = _model(params, N, "bf16")
    rng = np.random.default_rng(B * 100 + N)
    enc = rng.uniform(-1, 1, (B, N, 672)).astype(np.float32)
    v32 = np.zeros((B, 32), np.float32)
    v32[:, :27] = rng.uniform(-1, 1, (B, 27)).astype(np.float32)
    d_raw = np.concatenate([rng.normal(0, 1e-2, (B, N, 3)), rng.normal(0, 1e-3, (B, N, 1))], -1).astype(np.float32)
    e16 = torch.from_numpy(enc).to(DEV).to(torch.bfloat16).contiguous()
    v16 = torch.from_numpy(v32).to(DEV).to(torch.bfloat16).contiguous()
    nctx = model.mlp.native(torch.device(DEV))
    M = B * N
    sz = nctx.train_sizes(M)
    act = torch.zeros(sz[0], dtype=torch.uint8, device=DEV)
    masks = torch.zeros(sz[1], dtype=torch.uint8, device=DEV)
    delta = torch.zeros(sz[2], dtype=torch.uint8, device=DEV)
    raw = torch.empty(B, N, 4, device=DEV)
    rs = torch.empty_like(raw)
    L.check(L.lib().mipnerf_mlp_forward_train(nctx.handle, M, N, e16.data_ptr(), v16.data_ptr(), rs.data_ptr(), raw.data_ptr(), act.data_ptr(),
                                              masks.data_ptr(), ops._stream()), "mlp_forward_train")
    dr = torch.from_numpy(d_raw).to(DEV).contiguous()
    L.check(L.lib().mipnerf_mlp_dgrad(nctx.handle, M, dr.data_ptr(), masks.data_ptr(), delta.data_ptr(), ops._stream()), "mlp_dgrad")
    torch.cuda.synchronize()
    arch = Arch(xyz_dim=672, feat_per_deg=42, bf16_kernels=False)
    tp = TrainPlan.build(arch, pre_gemm=True)
    names = [n for n, _ in arch.param_shapes()]
    flat = np.concatenate([params[n].ravel() for n in names])
    n256 = (M + 255) // 256
    HT_g = act[:n256 * 8 * tp.NH * 2048].view(torch.bfloat16).float().cpu().numpy().reshape(-1, tp.NH, 2, 64, 8)
    GT_g = delta.view(torch.bfloat16).float().cpu().numpy().reshape(-1, tp.NG, 2, 64, 8)
    MK_g = masks.cpu().numpy().view(np.uint32).reshape(-1, tp.NMASK, 64, 4)
    encf = e16.float().cpu().numpy().reshape(-1, 672)
    viewf = np.repeat(v16.float().cpu().numpy(), N, axis=0)
    n_wt = (M + 31) // 32
    exact_tiles, flips, clean_delta, worst_act = 0, 0, 0, 0.0
    for t in range(n_wt):
        idx = np.minimum(np.arange(t * 32, t * 32 + 32), M - 1)
        valid = np.arange(t * 32, t * 32 + 32) < M
        HT, GT, _, _, MK = emulate_train_tile(tp, flat, encf[idx], viewf[idx], d_raw.reshape(-1, 4)[idx], valid, True, return_masks=True)
        nflip = int(np.unpackbits((MK_g[t] ^ MK).view(np.uint8)).sum())
        flips += nflip
        same_act = bool(np.array_equal(HT_g[t], HT))
        worst_delta = max(float(np.abs(GT_g[t, b] - GT[b]).max() / max(np.abs(GT[b]).max(), 1e-30)) for b in range(tp.NG))
        if same_act:
            exact_tiles += 1
            assert nflip == 0 and worst_delta <= 2.0 ** -7, (t, nflip, worst_delta)       # same activations: same masks, deltas within a bf16 ulp
        if nflip == 0 and worst_delta <= 2.0 ** -6:
            clean_delta += 1
        # a saved activation that differs does so by bf16 ulps OF ITS LAYER'S SCALE (a post-ReLU value next to zero may be 0 on one side, so the
        # measure is the difference over the block's largest value, not over the value itself)
        for b in range(tp.NH):
            worst_act = max(worst_act, float(np.abs(HT_g[t, b] - HT[b]).max() / max(float(np.abs(HT[b]).max()), 2.0 ** -20)))
    G.record("unbounded bf16 training kernels, tile by tile", tiles=n_wt, tiles_with_identical_activations=exact_tiles, relu_bit_flips=flips,
             tiles_with_clean_deltas=clean_delta, worst_activation_diff_over_block_max=worst_act)
    # measured (round 5, B=9 N=64: 18 wave tiles): 3 tiles bit-identical, 10 ReLU bits of 18 x 32 x 2,176 flipped, 13 tiles with clean deltas,
    # worst saved-activation difference 0.00595 of its block's largest value (1.5 bf16 ulps); bounds = 2 x measured
    assert worst_act <= 0.012, worst_act
    assert exact_tiles >= 1 and flips <= 20 and clean_delta >= n_wt // 2, (n_wt, exact_tiles, flips, clean_delta)

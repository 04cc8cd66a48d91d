"""GPU parity of the training-side kernels and of one training step against the reference's autograd
(golden loss / gradient checksums from scripts/make_golden.py)."""
import numpy as np
import pytest
import torch

from oracle import mipnerf_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def G():
    import gpu_util
    assert torch.cuda.is_available()
    return gpu_util


def torch_volumetric_rendering(rgb, density, t_samples, dirs, white_bkgd):
    """fp64 torch restatement of models/mip.py:366-401 (test-local reference for the backward kernel)."""
    t_mids = 0.5 * (t_samples[..., :-1] + t_samples[..., 1:])
    delta = (t_samples[..., 1:] - t_samples[..., :-1]) * torch.linalg.norm(dirs[:, None, :], dim=-1)
    dd = density[..., 0] * delta
    alpha = 1 - torch.exp(-dd)
    trans = torch.exp(-torch.cat([torch.zeros_like(dd[..., :1]), torch.cumsum(dd[..., :-1], -1)], -1))
    w = alpha * trans
    comp = (w[..., None] * rgb).sum(-2)
    acc = w.sum(-1)
    dist = torch.clamp(torch.nan_to_num((w * t_mids).sum(-1)), t_samples[:, 0], t_samples[:, -1])
    if white_bkgd:
        comp = comp + (1. - acc[..., None])
    return comp, dist, acc, w


@pytest.mark.parametrize("N", [64, 100, 128, 256])
@pytest.mark.parametrize("white", [True, False])
def test_render_backward_matches_autograd(G, N, white):
    from mipnerf_pl_amd.autograd import render_from_raw
    B = 21
    rng = np.random.default_rng(N)
    rays = orc.synthetic_rays(B, seed=2)
    t = torch.from_numpy(np.sort(rng.uniform(2, 6, (B, N + 1)), axis=-1).astype(np.float32))
    raw = torch.from_numpy(rng.normal(0, 2.5, (B, N, 4)).astype(np.float32))
    dirs = torch.from_numpy(rays.directions)
    gr, gd, ga, gw = [torch.from_numpy(rng.normal(0, 1, s).astype(np.float32)) for s in ((B, 3), (B,), (B,), (B, N))]
    # reference: fp64 torch autograd on CPU
    r64 = raw.double().requires_grad_(True)
    rgb = torch.sigmoid(r64[..., :3]) * 1.002 - 0.001
    den = torch.nn.functional.softplus(r64[..., 3:] - 1.0)
    out = torch_volumetric_rendering(rgb, den, t.double(), dirs.double(), white)
    (out[0] * gr).sum().add((out[1] * gd).sum()).add((out[2] * ga).sum()).add((out[3] * gw).sum()).backward()
    # HIP
    rg = raw.to(DEV).requires_grad_(True)
    o = render_from_raw(rg, t.to(DEV), dirs.to(DEV), white)
    ((o[0] * gr.to(DEV)).sum() + (o[1] * gd.to(DEV)).sum() + (o[2] * ga.to(DEV)).sum() + (o[3] * gw.to(DEV)).sum()).backward()
    for a, b in zip(o, out):
        assert G.maxdiff(a, b.float()) <= 2e-5
    e = G.maxdiff(rg.grad, r64.grad.float())
    scale = float(r64.grad.abs().max())
    G.record(f"render_bwd N={N} white={white}", err=e, scale=scale)
    assert e <= 2e-5 * max(1.0, scale)


@pytest.mark.parametrize("N", [64, 100, 128, 256])
def test_distloss_forward_backward(G, N):
    from mipnerf_pl_amd.autograd import distloss
    B = 19
    rng = np.random.default_rng(N + 1)
    t = torch.from_numpy(np.sort(rng.uniform(2, 6, (B, N + 1)), axis=-1).astype(np.float32))
    w = torch.from_numpy((rng.uniform(0, 1, (B, N)) ** 3).astype(np.float32))
    w64 = w.double().requires_grad_(True)
    s = t.double()
    interval = s[:, 1:] - s[:, :-1]
    mid = (s[:, 1:] + s[:, :-1]) * 0.5
    ref = (1 / 3) * (interval * w64.pow(2)).sum(-1).mean() + \
        ((w64[..., None] * w64[..., None, :]) * (mid[..., None] - mid[..., None, :]).abs()).sum((-1, -2)).mean()
    ref.backward()
    wg = w.to(DEV).requires_grad_(True)
    val = distloss(wg, t.to(DEV))
    val.backward()
    ev = abs(float(val) - float(ref)) / abs(float(ref))
    eg = G.maxdiff(wg.grad, w64.grad.float())
    G.record(f"distloss N={N}", rel_value=ev, grad=eg, grad_scale=float(w64.grad.abs().max()))
    assert ev <= 2e-6 and eg <= 2e-6 * max(1.0, float(w64.grad.abs().max()))
    np.testing.assert_allclose(float(val), float(orc.distloss(w.numpy(), t.numpy())), rtol=2e-5)


def test_training_step_matches_reference_gradients(G):
    """loss and d loss / d every parameter tensor vs the reference's autograd (fp32 mode)."""
    from mipnerf_pl_amd.system import DEFAULT_HPARAMS, MipNeRFSystem
    g = G.load_golden("train_64x64_trained")
    hp = dict(DEFAULT_HPARAMS)
    hp.update({'nerf.num_samples': int(g["num_samples"]), 'train.randomized': False})
    system = MipNeRFSystem(hp, precision="fp32")
    params = orc.make_params(seed=int(g["param_seed"]), density_gain=float(g["density_gain"]))
    system.load_state_dict({"mip_nerf.mlp." + k: torch.from_numpy(v.copy()) for k, v in params.items()})
    system = system.to(DEV)
    rays = G.to_dev(G.rays_of(g))
    rgbs = torch.from_numpy(g["gt"]).to(DEV)
    loss = system.training_step((rays, rgbs), 0)
    loss.backward()
    el = abs(float(loss) - float(g["loss"]))
    worst = 0.0
    for k, p in system.mip_nerf.mlp.named_parameters():
        gr = p.grad.detach().cpu().numpy().ravel()
        l2 = float(np.sqrt((gr.astype(np.float64) ** 2).sum()))
        ref_l2 = float(g["g_l2_" + k])
        stride = max(1, gr.size // 64)
        smp = gr[::stride][:64]
        rel = abs(l2 - ref_l2) / max(ref_l2, 1e-12)
        es = float(np.max(np.abs(smp - g["g_smp_" + k]))) / max(float(np.max(np.abs(g["g_smp_" + k]))), 1e-12)
        worst = max(worst, rel, es)
        assert rel <= 2e-3 and es <= 5e-3, (k, rel, es)
    G.record("training_step_fp32", loss_abs=el, worst_grad_rel=worst)
    assert el <= 2e-5


def test_training_step_bf16_close_to_fp32_and_optimizer_runs(G):
    from mipnerf_pl_amd.system import DEFAULT_HPARAMS, MipNeRFSystem
    g = G.load_golden("train_64x64_trained")
    params = orc.make_params(seed=int(g["param_seed"]), density_gain=float(g["density_gain"]))
    grads = {}
    for prec in ("fp32", "bf16"):
        hp = dict(DEFAULT_HPARAMS)
        hp.update({'nerf.num_samples': 64, 'train.randomized': False})
        system = MipNeRFSystem(hp, precision=prec)
        system.load_state_dict({"mip_nerf.mlp." + k: torch.from_numpy(v.copy()) for k, v in params.items()})
        system = system.to(DEV)
        (opt,), (sch,) = system.configure_optimizers()
        loss = system.training_step((G.to_dev(G.rays_of(g)), torch.from_numpy(g["gt"]).to(DEV)), 0)
        loss.backward()
        grads[prec] = torch.cat([p.grad.reshape(-1) for p in system.mip_nerf.parameters()]).clone()
        before = system.mip_nerf.mlp.layers[3][0].weight.detach().clone()
        opt.step()
        sch["scheduler"].step()
        assert not torch.equal(before, system.mip_nerf.mlp.layers[3][0].weight)
        # the no-grad (native MFMA) forward must see the updated weights (re-pack on version change)
        with torch.no_grad():
            a = system(G.to_dev(G.rays_of(g)), False, True)[1][0].clone()
        opt.step()
        with torch.no_grad():
            b = system(G.to_dev(G.rays_of(g)), False, True)[1][0]
        assert not torch.equal(a, b)
    cos = float(torch.nn.functional.cosine_similarity(grads["fp32"], grads["bf16"], dim=0))
    G.record("training_grad_bf16_vs_fp32", cosine=cos)
    assert cos > 0.99


def test_render_image_chunked_equals_unchunked(G):
    from mipnerf_pl_amd import Rays
    from mipnerf_pl_amd.system import DEFAULT_HPARAMS, MipNeRFSystem
    H, W = 20, 33     # 660 rays; chunk 256 -> ragged tail 148
    rays_np = orc.synthetic_rays(H * W, seed=8)
    hp = dict(DEFAULT_HPARAMS)
    hp.update({'nerf.num_samples': 64, 'val.chunk_size': 256})
    system = MipNeRFSystem(hp, precision="bf16")
    params = orc.make_params(seed=8, density_gain=40.0)
    system.load_state_dict({"mip_nerf.mlp." + k: torch.from_numpy(v.copy()) for k, v in params.items()})
    system = system.to(DEV)
    R = G.to_dev(rays_np)
    img_rays = Rays(*[x.reshape(1, H, W, -1) for x in R])
    rgbs = torch.zeros(1, H, W, 3, device=DEV)
    coarse, fine, mask = system.render_image((img_rays, rgbs))
    assert coarse.shape == (1, H, W, 3) and fine.shape == (1, H, W, 3) and mask.shape == (1, H, W, 1)
    with torch.no_grad():
        ret = system(R, False, True)
    assert torch.equal(fine.reshape(-1, 3), ret[1][0]) and torch.equal(coarse.reshape(-1, 3), ret[0][0])
    out = system.validation_step((img_rays, rgbs), 0)
    assert torch.isfinite(out['val/psnr'])


# ---- native bf16 MLP training kernels (forward-with-save, dgrad, wgrad) ---------------------------------------
def _mlp_case(B, N, seed, **arch):
    rng = np.random.default_rng(seed)
    params = orc.make_params(seed=seed, density_gain=40.0, **arch)
    enc = (rng.uniform(-1, 1, (B, N, 96)) * rng.uniform(0, 1, (1, 1, 96)) ** 2).astype(np.float32)
    vdir = rng.normal(0, 1, (B, 3)).astype(np.float32)
    vdir /= np.linalg.norm(vdir, axis=-1, keepdims=True)
    venc = orc.pos_enc(vdir, 0, 4, True).astype(np.float32)
    d_raw = np.concatenate([rng.normal(0, 1e-2, (B, N, 3)), rng.normal(0, 1e-3, (B, N, 1))], -1).astype(np.float32)
    return params, enc, venc, d_raw


def _run_native_mlp(G, params, enc, venc, d_raw, **model_kw):
    from mipnerf_pl_amd.autograd import mlp_native
    B, N = enc.shape[:2]
    model = G.make_model(params, N, "bf16", **model_kw)
    v32 = np.zeros((B, 32), np.float32)
    v32[:, :27] = venc
    e = torch.from_numpy(enc).to(DEV).to(torch.bfloat16)
    v = torch.from_numpy(v32).to(DEV).to(torch.bfloat16)
    raw = mlp_native(model.mlp, e, v)
    (raw * torch.from_numpy(d_raw).to(DEV)).sum().backward()
    grads = {k: (p.grad.detach().cpu().numpy() if p.grad is not None else None) for k, p in model.mlp.named_parameters()}
    return raw.detach().cpu().numpy(), grads, e.float().cpu().numpy(), v.float().cpu().numpy()


@pytest.mark.parametrize("B,N", [(8, 32), (5, 24), (3, 100)])
def test_native_mlp_backward_equals_bf16_emulation(G, B, N):
    """The three training kernels against the numpy emulation of the SAME dataflow with bf16 operand rounding
    (fp32 accumulation order is the only difference) and against the fp32 oracle gradients."""
    from mipnerf_pl_amd.mlp_train_plan import TrainPlan, emulate_train
    params, enc, venc, d_raw = _mlp_case(B, N, seed=B * 100 + N)
    raw, grads, enc_bf, v_bf = _run_native_mlp(G, params, enc, venc, d_raw)
    tp = TrainPlan.build()
    flatp = np.concatenate([v.ravel() for v in params.values()])
    S = B * N
    flat, seen, raw_em = emulate_train(tp, flatp, enc_bf.reshape(S, 96), np.repeat(v_bf, N, axis=0),
                                       d_raw.reshape(S, 4), round_bf16=True)
    e_raw = G.maxdiff(raw.reshape(S, 4), raw_em)
    og = orc.mlp_backward(params, enc, venc, d_raw[..., :3], d_raw[..., 3:])
    off, worst_em, worst_or = 0, 0.0, 0.0
    for k, v in og.items():
        g = grads[k].ravel().astype(np.float64)
        em = flat[off:off + v.size].astype(np.float64)
        off += v.size
        rel_em = np.linalg.norm(g - em) / max(np.linalg.norm(em), 1e-30)
        rel_or = np.linalg.norm(g - v.ravel()) / max(np.linalg.norm(v.ravel()), 1e-30)
        worst_em, worst_or = max(worst_em, rel_em), max(worst_or, rel_or)
    G.record(f"native_mlp_bwd B={B} N={N}", raw_vs_emul=e_raw, grad_rel_l2_vs_emul=worst_em, grad_rel_l2_vs_fp32=worst_or)
    # same dataflow: only the fp32 accumulation order (and rare bf16 round-to-even tie flips) differ
    assert worst_em <= 1e-2, worst_em
    # bf16 operands (8-bit mantissa) for activations AND deltas through up to 10 chained layers: measured 0.11-0.13
    # relative L2 on layers.0.weight (the deepest gradient), < 0.05 on the heads
    assert worst_or <= 0.2, worst_or       # measured 0.11-0.16 (white-noise upstream gradients: pure cancellation, the adversarial case)
    assert e_raw <= 2e-2 * max(1.0, float(np.abs(raw_em).max()))



@pytest.mark.parametrize("B,N", [(8, 32), (3, 100)])
def test_native_mlp_backward_variant_w128(G, B, N):
    """The training kernels generated for the 128-wide variant (gen_mlp_train.train_variants) against the numpy emulation
    of the same dataflow with bf16 operand rounding and against the fp32 oracle gradients of that architecture."""
    from mipnerf_pl_amd.mlp_plan import Arch
    from mipnerf_pl_amd.mlp_train_plan import TrainPlan, emulate_train
    arch = dict(net_width=128, net_width_condition=128)
    params, enc, venc, d_raw = _mlp_case(B, N, seed=B * 100 + N + 7, **arch)
    raw, grads, enc_bf, v_bf = _run_native_mlp(G, params, enc, venc, d_raw, mlp_net_width=128, mlp_net_width_condition=128)
    tp = TrainPlan.build(Arch(**arch))
    flatp = np.concatenate([v.ravel() for v in params.values()])
    S = B * N
    flat, seen, raw_em = emulate_train(tp, flatp, enc_bf.reshape(S, 96), np.repeat(v_bf, N, axis=0), d_raw.reshape(S, 4),
                                       round_bf16=True)
    e_raw = G.maxdiff(raw.reshape(S, 4), raw_em)
    og = orc.mlp_backward(params, enc, venc, d_raw[..., :3], d_raw[..., 3:])
    off, worst_em, worst_or = 0, 0.0, 0.0
    for k, v in og.items():
        g = grads[k].ravel().astype(np.float64)
        em = flat[off:off + v.size].astype(np.float64)
        off += v.size
        worst_em = max(worst_em, np.linalg.norm(g - em) / max(np.linalg.norm(em), 1e-30))
        worst_or = max(worst_or, np.linalg.norm(g - v.ravel()) / max(np.linalg.norm(v.ravel()), 1e-30))
    G.record(f"native_mlp_bwd_w128 B={B} N={N}", raw_vs_emul=e_raw, grad_rel_l2_vs_emul=worst_em, grad_rel_l2_vs_fp32=worst_or)
    assert worst_em <= 1e-2, worst_em
    assert worst_or <= 0.2, worst_or
    assert e_raw <= 2e-2 * max(1.0, float(np.abs(raw_em).max()))


def test_native_mlp_backward_against_reference_golden(G):
    """bf16 kernels vs the reference's own autograd gradients (golden, fp32): direction and norm."""
    g = G.load_golden("mlp_bwd_8x32_trained")
    params = orc.make_params(seed=int(g["param_seed"]), density_gain=float(g["density_gain"]))
    d_raw = np.concatenate([g["d_rgb"], g["d_den"]], -1)
    raw, grads, _, _ = _run_native_mlp(G, params, g["enc"], g["venc"], d_raw)
    assert G.maxdiff(raw[..., :3], g["raw_rgb"]) <= 3e-2 and G.maxdiff(raw[..., 3:], g["raw_density"]) <= 0.5
    worst = 0.0
    for k, gr in grads.items():
        flat = gr.ravel()
        ref = g["g_smp_" + k]
        stride = max(1, flat.size // ref.size)
        smp = flat[::stride][:ref.size].astype(np.float64)
        l2 = float(np.sqrt((flat.astype(np.float64) ** 2).sum()))
        rel_l2 = abs(l2 - float(g["g_l2_" + k])) / float(g["g_l2_" + k])
        cos = float((smp * ref).sum() / max(np.linalg.norm(smp) * np.linalg.norm(ref), 1e-30))
        worst = max(worst, rel_l2, 1 - cos)
        G.record("native_mlp_bwd_vs_reference_golden " + k, rel_l2_norm=rel_l2, cosine=cos)
        assert rel_l2 <= 5e-2 and cos >= 0.98, (k, rel_l2, cos)      # measured: cos 0.992 on layers.0.weight


def test_native_mlp_full_size_linearity_and_split_invariance(G):
    """BASELINE configs[1] size (4096 x 128 samples): size-independent properties of the backward --
    linear in d_raw, and the sum of the gradients of two ray halves equals the gradient of the whole batch."""
    B, N = 4096, 128
    params, enc, venc, d_raw = _mlp_case(64, N, seed=77)
    reps = B // 64
    enc, venc, d_raw = np.tile(enc, (reps, 1, 1)), np.tile(venc, (reps, 1)), np.tile(d_raw, (reps, 1, 1))
    d_raw = d_raw * np.random.default_rng(5).uniform(0.5, 1.5, (B, 1, 1)).astype(np.float32)
    _, g_all, _, _ = _run_native_mlp(G, params, enc, venc, d_raw)
    _, g_a, _, _ = _run_native_mlp(G, params, enc[:B // 2], venc[:B // 2], d_raw[:B // 2])
    _, g_b, _, _ = _run_native_mlp(G, params, enc[B // 2:], venc[B // 2:], d_raw[B // 2:])
    _, g_2x, _, _ = _run_native_mlp(G, params, enc, venc, 2.0 * d_raw)
    worst = 0.0
    for k in g_all:
        scale = float(np.abs(g_all[k]).max()) + 1e-30
        e_split = float(np.abs(g_a[k] + g_b[k] - g_all[k]).max()) / scale
        e_lin = float(np.abs(g_2x[k] - 2.0 * g_all[k]).max()) / scale
        worst = max(worst, e_split, e_lin)
        assert np.isfinite(g_all[k]).all() and e_split <= 1e-3 and e_lin <= 1e-6, (k, e_split, e_lin)
    G.record("native_mlp_bwd_full_size", worst=worst)


def test_render_image_hip_graph_equals_eager(G):
    """BASELINE configs[4] plumbing: the chunk loop of render_image replayed from a captured hipGraph returns
    bit-identical images (incl. the ragged last chunk and a parameter update between frames)."""
    from mipnerf_pl_amd import Rays
    from mipnerf_pl_amd.system import DEFAULT_HPARAMS, MipNeRFSystem
    H, W = 24, 37     # 888 rays; chunk 256 -> 3 full chunks + ragged tail of 120
    rays_np = orc.synthetic_rays(H * W, seed=9)
    hp = dict(DEFAULT_HPARAMS)
    hp.update({'nerf.num_samples': 64, 'val.chunk_size': 256})
    system = MipNeRFSystem(hp, precision="bf16")
    params = orc.make_params(seed=9, density_gain=40.0)
    system.load_state_dict({"mip_nerf.mlp." + k: torch.from_numpy(v.copy()) for k, v in params.items()})
    system = system.to(DEV)
    R = G.to_dev(rays_np)
    img_rays = Rays(*[x.reshape(1, H, W, -1) for x in R])
    rgbs = torch.zeros(1, H, W, 3, device=DEV)
    eager = system.render_image((img_rays, rgbs), return_distance=True)
    system.enable_hip_graph(True)
    for _ in range(2):      # second frame replays without re-capturing
        graphed = system.render_image((img_rays, rgbs), return_distance=True)
        for a, b in zip(eager, graphed):
            assert torch.equal(a, b)
    with torch.no_grad():
        system.mip_nerf.mlp.color_layer.bias.add_(0.25)      # optimizer-step stand-in: streams must be re-packed
    g2 = system.render_image((img_rays, rgbs))
    system.enable_hip_graph(False)
    e2 = system.render_image((img_rays, rgbs))
    assert torch.equal(g2[1], e2[1]) and not torch.equal(g2[1], eager[1])


def _train_system(G, g, fused):
    from mipnerf_pl_amd.system import DEFAULT_HPARAMS, MipNeRFSystem
    hp = dict(DEFAULT_HPARAMS)
    hp.update({'nerf.num_samples': 64, 'train.randomized': False, 'optimizer.lr_delay_steps': 3, 'optimizer.max_steps': 10})
    system = MipNeRFSystem(hp, precision="bf16")
    params = orc.make_params(seed=int(g["param_seed"]), density_gain=float(g["density_gain"]))
    system.load_state_dict({"mip_nerf.mlp." + k: torch.from_numpy(v.copy()) for k, v in params.items()})
    system = system.to(DEV)
    system.fused_adam = fused
    (opt,), (sch,) = system.configure_optimizers()
    return system, opt, sch["scheduler"]


def test_flat_adam_equals_torch_adam(G):
    """flat gradient buffer (written by the wgrad reduction, both levels accumulated in place) + one fused
    Adam kernel + MipLRDecay == torch.optim.Adam on per-tensor autograd gradients, step for step."""
    g = G.load_golden("train_64x64_trained")
    rays, gt = G.to_dev(G.rays_of(g)), torch.from_numpy(g["gt"]).to(DEV)
    runs = {}
    for fused in (False, True):
        system, opt, sch = _train_system(G, g, fused)
        assert system.mip_nerf.mlp.is_flat() == fused
        losses, grads = [], None
        for it in range(4):
            opt.zero_grad()
            loss = system.training_step((rays, gt), it)
            loss.backward()
            if it == 0:
                grads = torch.cat([p.grad.reshape(-1) for p in system.mip_nerf.parameters()]).clone()
            opt.step()
            sch.step()
            if it == 0:
                after1 = torch.cat([p.detach().reshape(-1) for p in system.mip_nerf.parameters()]).clone()
            losses.append(float(loss))
        runs[fused] = (losses, grads, torch.cat([p.detach().reshape(-1) for p in system.mip_nerf.parameters()]).clone(),
                       list(system.mip_nerf.state_dict().keys()), after1)
    (l0, g0, p0, k0, a0), (l1, g1, p1, k1, a1) = runs[False], runs[True]
    assert k0 == k1
    # the optimiser arithmetic itself, pinned BEFORE the bf16 flip cascade described below can start -- after ONE step from
    # identical parameters and (asserted next) identical gradients the two parameter sets differ by fp32 rounding of the update only
    e1 = G.maxdiff(a0, a1)
    G.record("flat_adam_vs_torch_adam after one step", param_abs=e1, param_mean_abs=float((a0 - a1).abs().mean()))
    # measured on MI355X (profiles/r06_parity.jsonl): max 7.5e-9 (one ulp of a parameter of magnitude 0.06-0.12), mean 1.1e-13 -- all but a few
    # hundred of the 612,740 parameters agree bit for bit; bounds = 4 x those
    assert e1 <= 3e-8 and float((a0 - a1).abs().mean()) <= 5e-13, (e1, float((a0 - a1).abs().mean()))
    eg = G.maxdiff(g0, g1) / float(g0.abs().max())
    ep = G.maxdiff(p0, p1)
    emean = float((p0 - p1).abs().mean())
    G.record("flat_adam_vs_torch_adam", grad_rel=eg, param_abs=ep, param_mean_abs=emean, loss0=l0[-1], loss1=l1[-1])
    assert eg <= 1e-6                     # same kernels; only (a + b) association of the two levels could differ
    # The two optimisers round differently by at most an fp32 ulp per step; once one master weight sits within that ulp of
    # a bf16 rounding boundary its packed value flips, the loss moves by ~1e-6 and Adam's g / sqrt(v) turns sign changes
    # of near-zero gradients into +-lr (observed: 3.5e-4 on a handful of parameters after 4 steps, loss equal to 2e-5).
    # Bound = a few learning rates; the per-step arithmetic itself is pinned by test_device_lr_schedule_equals_host_miplrdecay.
    assert ep <= 4 * 1.73e-4 and abs(l0[-1] - l1[-1]) <= 1e-4 * max(1.0, abs(l0[-1]))
    # mean over the 612,740 parameters: some thousand such flips of a fraction of lr.  The figure is chaotic in the ulps of the step's
    # reductions: 2e-7 with the shuffle-tree wave sums of rounds 1-4, 1.26e-6 with the DPP sums of round 5 (same gradients at step 0:
    # grad_rel == 0 above, same loss to 2e-5, every golden / trajectory / quality test unchanged) -- bounded at 2.5 x the larger one
    assert emean <= 3e-6, emean
    assert l0[-1] != l0[0]                # the weights did move (re-pack after the raw-kernel update is effective)


def test_adam_single_step_equals_torch_adam_any_betas(G):
    """mipnerf_adam_step takes the python doubles torch.optim.Adam holds (no float round trip, no special-cased
    0.9 / 0.999): ONE step on an identical gradient equals torch's single-tensor Adam to an fp32 ulp, also for other betas
    and at a late step count (bias corrections formed in double)."""
    from mipnerf_pl_amd import _lib as L
    from mipnerf_pl_amd import ops
    gen = torch.Generator(device=DEV).manual_seed(5)
    n = 100_003
    for betas, lr, eps, steps_before in (((0.9, 0.999), 5e-4, 1e-8, 0), ((0.8, 0.95), 3e-4, 1e-6, 0), ((0.85, 0.9875), 1.7e-3, 1e-8, 37)):
        p0 = torch.randn(n, device=DEV, generator=gen) * 0.1
        ref = torch.nn.Parameter(p0.clone())
        opt = torch.optim.Adam([ref], lr=lr, betas=betas, eps=eps, foreach=False, fused=False)
        mine, m, v = p0.clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
        for t in range(1, steps_before + 2):
            g = torch.randn(n, device=DEV, generator=gen) * 1e-2
            ref.grad = g.clone()
            opt.step()
            L.check(L.lib().mipnerf_adam_step(n, mine.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), lr, betas[0], betas[1], eps, t,
                                              ops._stream()), "adam_step")
        torch.cuda.synchronize()
        d = (mine - ref.detach()).abs()
        G.record(f"adam_single_step betas={betas} steps={steps_before + 1}", max_abs=float(d.max()))
        # one fp32 ulp of the parameter per step at most (the division order inside addcdiv is torch's)
        assert float(d.max()) <= (steps_before + 1) * 1.2e-7 * float(ref.detach().abs().max()) + 1e-9, float(d.max())
        # moments: one rounding of (g - m) * w apart (torch's lerp kernel may contract into an fma): an ulp of the LARGER operand
        assert torch.allclose(m, opt.state[ref]["exp_avg"], rtol=3e-7, atol=4e-9), float((m - opt.state[ref]["exp_avg"]).abs().max())
        assert torch.allclose(v, opt.state[ref]["exp_avg_sq"], rtol=3e-7, atol=1e-11), float((v - opt.state[ref]["exp_avg_sq"]).abs().max())


def test_end_to_end_training_bf16_tracks_fp32(G):
    """Stand-in for "PSNR within 0.1 dB of the reference" without Blender data (scripts/train_synthetic.py): a student
    trained on device-generated rays of a procedural scene, native bf16 kernels + FlatAdam vs the fp32 parity mode on
    identical batches.  Both must learn (validation PSNR rises by > 4 dB) and stay close: the curves cross each other
    by +-0.5 dB while the loss is still falling fast (160 steps here), and agree to 0.016 dB once it flattens (400
    steps, profiles/r01h_train_synthetic.json: 31.459 vs 31.474 dB)."""
    import subprocess
    import sys
    import os
    import json
    out = subprocess.run([sys.executable, os.path.join(G.REPO, "scripts", "train_synthetic.py"), "--steps", "160", "--rays", "2048",
                          "--samples", "32", "--every", "40", "--compare-fp32"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads(out.stdout.strip().splitlines()[-1])
    b, f = res["bf16_native"], res["fp32_parity_mode"]
    G.record("train_synthetic_160", bf16_psnr0=b[0][2], bf16_psnr=b[-1][2], fp32_psnr=f[-1][2])
    assert b[-1][2] > b[0][2] + 4.0 and f[-1][2] > f[0][2] + 4.0
    assert abs(b[-1][2] - f[-1][2]) <= 1.0


@pytest.mark.parametrize("flat", [True, False])
def test_native_train_step_equals_autograd_path(G, flat):
    """mipnerf_train_step (forward + loss + backward in one native call, no autograd graph) against
    training_step + loss.backward() through the per-stage custom autograd Functions: same kernels underneath, so loss and every
    gradient must agree to fp32 round-off (the loss reduction runs in fp64 instead of torch's fp32).  Round 6: the DEFAULT
    training_step (one autograd node whose forward is that native call, MipNerf.loss_native) + loss.backward() must give the
    native call's loss and gradient BIT FOR BIT, and twice that after a second backward (AccumulateGrad)."""
    from mipnerf_pl_amd.system import DEFAULT_HPARAMS, MipNeRFSystem
    g = G.load_golden("train_64x64_trained")
    rays, gt = G.to_dev(G.rays_of(g)), torch.from_numpy(g["gt"]).to(DEV)
    params = orc.make_params(seed=int(g["param_seed"]), density_gain=float(g["density_gain"]))
    res = {}
    for route in ("per_stage", "native_call", "routed"):
        hp = dict(DEFAULT_HPARAMS)
        hp.update({'nerf.num_samples': 64, 'train.randomized': False})
        system = MipNeRFSystem(hp, precision="bf16")
        system.load_state_dict({"mip_nerf.mlp." + k: torch.from_numpy(v.copy()) for k, v in params.items()})
        system = system.to(DEV)
        if flat:
            system.mip_nerf.mlp.flatten_parameters()
        if route == "native_call":
            loss = system.training_step_native((rays, gt), 0)
            loss2 = system.training_step_native((rays, gt), 1)          # second call accumulates like a second backward()
        else:
            system.native_training_step = route == "routed"
            assert system._native_step_route(rays) == (route == "routed")
            loss = system.training_step((rays, gt), 0)
            assert loss.requires_grad and loss.shape == ()
            loss.backward()
            if route == "routed":
                g1 = torch.cat([p.grad.reshape(-1) for p in system.mip_nerf.parameters()]).clone()
                loss2 = system.training_step((rays, gt), 1)
                loss2.backward()
        grads = torch.cat([p.grad.reshape(-1) for p in system.mip_nerf.parameters()]).clone()
        res[route] = (float(loss), grads, float(system.logged['train/psnr']))
        if route != "per_stage":
            assert abs(float(loss2) - float(loss)) < 1e-7
            assert G.maxdiff(grads, 2.0 * res["per_stage"][1]) <= 2e-5 * float(res["per_stage"][1].abs().max())
            res[route] = (float(loss), grads / 2.0 if route == "native_call" else g1, res[route][2])
    (l0, g0, p0), (l1, g1, p1), (l2, g2, p2) = res["per_stage"], res["native_call"], res["routed"]
    eg = G.maxdiff(g0, g1) / float(g0.abs().max())
    G.record(f"native_train_step flat={flat}", loss_autograd=l0, loss_native=l1, grad_rel=eg, psnr_autograd=p0, psnr_native=p1,
             routed_vs_native_call=G.maxdiff(g2, g1))
    assert abs(l0 - l1) <= 2e-6 * max(1.0, abs(l0)) and abs(p0 - p1) <= 1e-3 and eg <= 2e-5
    assert abs(l1 - float(g["loss"])) <= 5e-3      # and the bf16 loss is the reference's loss (fp32) to bf16 accuracy
    # the routed training_step IS the native call: same loss bits, same psnr bits, and (first backward: gradient x 1.0) the same gradient
    # bits up to the halving of the accumulated native gradient above (exact in binary floating point unless a value is denormal)
    assert l2 == l1 and p2 == p1 and G.maxdiff(g2, g1) <= 1e-12 * float(g1.abs().max()) + 1e-30


def test_routed_training_step_scales_with_the_incoming_gradient_and_feeds_hooks(G):
    """MipNerf.loss_native is an ordinary autograd node: (3 * loss).backward() gives 3 x the gradient, torch.autograd.grad returns the
    per-parameter gradients, and tensor hooks on the parameters (what DistributedDataParallel's reducer registers) fire once each."""
    from mipnerf_pl_amd.system import DEFAULT_HPARAMS, MipNeRFSystem
    g = G.load_golden("train_64x64_trained")
    rays, gt = G.to_dev(G.rays_of(g)), torch.from_numpy(g["gt"]).to(DEV)
    params = orc.make_params(seed=int(g["param_seed"]), density_gain=float(g["density_gain"]))
    hp = dict(DEFAULT_HPARAMS)
    hp.update({'nerf.num_samples': 64, 'train.randomized': False})
    system = MipNeRFSystem(hp, precision="bf16")
    system.load_state_dict({"mip_nerf.mlp." + k: torch.from_numpy(v.copy()) for k, v in params.items()})
    system = system.to(DEV)
    ps = list(system.mip_nerf.parameters())
    loss = system.training_step((rays, gt), 0)
    base = torch.autograd.grad(loss, ps)
    assert all(b is not None and b.shape == p.shape for b, p in zip(base, ps)) and all(p.grad is None for p in ps)
    fired = []
    handles = [p.register_hook(lambda gr, i=i: fired.append(i)) for i, p in enumerate(ps)]
    (3.0 * system.training_step((rays, gt), 1)).backward()
    assert sorted(fired) == list(range(len(ps)))
    for p, b in zip(ps, base):
        assert G.maxdiff(p.grad, 3.0 * b) <= 1e-6 * float(b.abs().max()) + 1e-30
    for h in handles:
        h.remove()
    # evaluation / no_grad and frozen parameters keep the ordinary routes
    with torch.no_grad():
        assert not system._native_step_route(rays)
    ps[0].requires_grad_(False)
    assert not system._native_step_route(rays)


def test_mlp_module_forward_is_differentiable(G):
    """The standalone MLP module (models/mip_nerf.py:14-111 contract) must produce gradients when called under
    autograd, like the reference's nn.Module."""
    params, enc, venc, d_raw = _mlp_case(4, 32, seed=3)
    model = G.make_model(params, 32, "bf16")
    rgb, den = model.mlp(torch.from_numpy(enc).to(DEV), torch.from_numpy(venc).to(DEV))
    assert rgb.shape == (4, 32, 3) and den.shape == (4, 32, 1) and rgb.requires_grad
    (rgb.sum() + den.sum()).backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.mlp.parameters())
    with torch.no_grad():
        rgb2, den2 = model.mlp(torch.from_numpy(enc).to(DEV), torch.from_numpy(venc).to(DEV))
    assert G.maxdiff(rgb, rgb2) <= 1e-6 and not rgb2.requires_grad     # same forward numbers with or without the graph


def test_fp32_native_mlp_backward_matches_reference_golden(G):
    """Parity mode: fused fp32 forward-with-save + fp32-MFMA GEMM backward against the reference's autograd gradients
    (golden, full fp32 tolerance) and against a plain-PyTorch fp32 restatement of the same op."""
    from mipnerf_pl_amd.autograd import mlp_native_f32
    from gpu_util import mlp_torch
    g = G.load_golden("mlp_bwd_8x32_trained")
    params = orc.make_params(seed=int(g["param_seed"]), density_gain=float(g["density_gain"]))
    d_raw = torch.from_numpy(np.concatenate([g["d_rgb"], g["d_den"]], -1)).to(DEV)
    enc = torch.from_numpy(g["enc"]).to(DEV)
    v32 = torch.zeros(8, 32, device=DEV)
    v32[:, :27] = torch.from_numpy(g["venc"]).to(DEV)
    res = {}
    for name, fn in (("native", lambda m: mlp_native_f32(m.mlp, enc, v32)),
                     ("torch", lambda m: mlp_torch(m.mlp, enc, v32[:, :27], torch.float32))):
        model = G.make_model(params, 32, "fp32")
        raw = fn(model)
        (raw * d_raw).sum().backward()
        res[name] = (raw.detach(), {k: p.grad.detach().cpu().numpy() for k, p in model.mlp.named_parameters()})
    raw = res["native"][0].cpu().numpy()
    assert G.maxdiff(raw[..., :3], g["raw_rgb"]) <= 2e-5 and G.maxdiff(raw[..., 3:], g["raw_density"]) <= 2e-4
    worst = 0.0
    for k, gr in res["native"][1].items():
        flat, ref = gr.ravel(), g["g_smp_" + k]
        stride = max(1, flat.size // ref.size)
        smp = flat[::stride][:ref.size]
        l2 = float(np.sqrt((flat.astype(np.float64) ** 2).sum()))
        rel_l2 = abs(l2 - float(g["g_l2_" + k])) / float(g["g_l2_" + k])
        es = float(np.max(np.abs(smp - ref))) / max(float(np.max(np.abs(ref))), 1e-20)
        et = float(np.max(np.abs(gr - res["torch"][1][k]))) / max(float(np.max(np.abs(res["torch"][1][k]))), 1e-20)
        worst = max(worst, rel_l2, es, et)
        assert rel_l2 <= 1e-4 and es <= 1e-4 and et <= 1e-4, (k, rel_l2, es, et)
    G.record("fp32_native_mlp_bwd_vs_reference_golden", worst=worst)


@pytest.mark.parametrize("B,N,white", [(37, 100, False), (300, 32, True), (1, 1, True), (2, 3, False), (5, 65, True), (1, 512, False), (2, 1000, True)])
def test_native_train_step_ragged_shapes_randomized(G, B, N, white):
    """mipnerf_train_step vs the autograd path on ragged sizes (rays not a multiple of the 256-sample tile, N not a
    multiple of 32) with the SAME stratified / resampling draws injected into both."""
    from mipnerf_pl_amd import MipNerf
    from mipnerf_pl_amd.autograd import distloss
    rays = G.to_dev(orc.synthetic_rays(B, seed=B, multiscale=True))
    params = orc.make_params(seed=B, density_gain=40.0)
    gt = torch.rand(B, 3, device=DEV)
    t_rand, u_rand = torch.rand(B, N + 1, device=DEV), torch.rand(B, N + 1, device=DEV)
    res = {}
    for native in (False, True):
        model = G.make_model(params, N, "bf16")
        if native:
            scal, _ = model.train_step_native(rays, gt, True, white, t_rand=t_rand, u_rand=u_rand)
            loss = float(scal[0])
        else:
            ret = model(rays, True, white, t_rand=t_rand, u_rand=u_rand)
            mask = rays.lossmult
            mse = [(mask * (r[0] - gt) ** 2).sum() / mask.sum() for r in ret]
            dl = [distloss(r[3], r[4]) for r in ret]
            tot = 0.1 * (mse[0] + 0.01 * dl[0]) + mse[1] + 0.01 * dl[1]
            tot.backward()
            loss = float(tot.detach())
        res[native] = (loss, torch.cat([p.grad.reshape(-1) for p in model.parameters()]).clone())
    (l0, g0), (l1, g1) = res[False], res[True]
    eg = G.maxdiff(g0, g1) / float(g0.abs().max())
    G.record(f"native_train_step_ragged B={B} N={N}", loss_autograd=l0, loss_native=l1, grad_rel=eg)
    assert abs(l0 - l1) <= 2e-6 * max(1.0, abs(l0)) and eg <= 2e-5


def test_device_lr_schedule_equals_host_miplrdecay(G):
    """FlatAdam(schedule=...) evaluates MipLRDecay (utils/lr_schedule.py:51-59) and Adam's bias corrections
    on the device.  Against torch.optim.Adam + the host MipLRDecay on the same gradients: parameters agree to fp32
    round-off, and the learning rate the device used equals the host schedule at every step."""
    from mipnerf_pl_amd import MipNerf
    from mipnerf_pl_amd.lr_schedule import MipLRDecay, mip_lr
    from mipnerf_pl_amd.optim import FlatAdam
    sched = dict(lr_init=2e-3, lr_final=1e-4, max_steps=40, lr_delay_steps=10, lr_delay_mult=0.01)
    torch.manual_seed(0)
    m1, m2 = MipNerf(num_samples=8).to(DEV), MipNerf(num_samples=8).to(DEV)
    m2.load_state_dict(m1.state_dict())
    o1 = FlatAdam(m1.mlp, lr=sched["lr_init"], schedule=sched)
    o2 = torch.optim.Adam(m2.parameters(), lr=sched["lr_init"])
    s2 = MipLRDecay(o2, **sched)
    n = m1.mlp._flat_grad.numel()
    for step in range(45):
        g = torch.randn(n, device=DEV) * (0.1 if step % 3 else 1e-4)
        m1.mlp._flat_grad.copy_(g)
        m1.mlp._flat_grad_valid = True
        off = 0
        for p in m2.parameters():
            p.grad = g[off:off + p.numel()].view_as(p).clone()
            off += p.numel()
        want_lr = o2.param_groups[0]["lr"]
        o1.step()
        o2.step()
        s2.step()
        assert abs(o1.last_lr() - want_lr) <= 2e-7 * want_lr, (step, o1.last_lr(), want_lr)
        assert abs(want_lr - mip_lr(step, **sched)) <= 1e-18
    assert o1.steps == 45 and int(o1._dev_step.item()) == 45
    a = torch.cat([p.detach().reshape(-1) for p in m1.parameters()])
    b = torch.cat([p.detach().reshape(-1) for p in m2.parameters()])
    G.record("device_lr_adam", max_abs=G.maxdiff(a, b))
    assert G.maxdiff(a, b) <= 5e-6


def test_graphed_train_step_equals_eager(G):
    """The whole optimisation step (forward + loss + backward + scheduled Adam + weight re-pack) replayed from one
    captured hipGraph must equal the same launches issued eagerly, bit for bit, and the plain hook loop
    (training_step_native + FlatAdam.step + scheduler.step) to round-off."""
    from mipnerf_pl_amd.system import DEFAULT_HPARAMS, MipNeRFSystem
    from mipnerf_pl_amd.train_graph import GraphedTrainStep
    g = G.load_golden("train_64x64_trained")
    rays, gt = G.to_dev(G.rays_of(g)), torch.from_numpy(g["gt"]).to(DEV)
    params = orc.make_params(seed=int(g["param_seed"]), density_gain=float(g["density_gain"]))
    B = rays.origins.shape[0]

    def make():
        hp = dict(DEFAULT_HPARAMS)
        hp.update({'nerf.num_samples': 64, 'train.randomized': False, 'optimizer.lr_init': 1e-3, 'optimizer.lr_final': 1e-5,
                   'optimizer.max_steps': 20, 'optimizer.lr_delay_steps': 4})
        system = MipNeRFSystem(hp, precision="bf16")
        system.load_state_dict({"mip_nerf.mlp." + k: torch.from_numpy(v.copy()) for k, v in params.items()})
        system = system.to(DEV)
        system.fused_adam = True
        (opt,), (sch,) = system.configure_optimizers()
        return system, opt, sch["scheduler"]
    res = {}
    for mode in ("graph", "eager", "hooks"):
        system, opt, sch = make()
        losses = []
        if mode == "hooks":
            for it in range(5):
                opt.zero_grad()
                losses.append(float(system.training_step_native((rays, gt), it)))
                opt.step()
                sch.step()
        else:
            step = GraphedTrainStep(system, opt, B, torch.device(DEV), use_graph=(mode == "graph"))
            for dst, src in zip(step.rays, rays):
                dst.copy_(src)
            step.gt.copy_(gt)
            for it in range(5):
                losses.append(float(step()[0]))
                sch.step()
            assert opt.steps == 5 and int(opt._dev_step.item()) == 5
        with torch.no_grad():
            out = system.mip_nerf(rays, False, True)[1][0].clone()      # uses the re-packed weight streams
        res[mode] = (losses, torch.cat([p.detach().reshape(-1) for p in system.mip_nerf.parameters()]).clone(), out)
    assert res["graph"][0] == res["eager"][0], (res["graph"][0], res["eager"][0])
    assert torch.equal(res["graph"][1], res["eager"][1]) and torch.equal(res["graph"][2], res["eager"][2])
    assert res["graph"][0][-1] < res["graph"][0][0]           # it trains
    for a, b in zip(res["graph"][0], res["hooks"][0]):
        assert abs(a - b) <= 1e-6 * max(1.0, abs(b))
    G.record("graphed_train_step", hooks_vs_graph_params=G.maxdiff(res["graph"][1], res["hooks"][1]))
    assert G.maxdiff(res["graph"][1], res["hooks"][1]) <= 1e-6


def test_native_mlp_backward_variant_noview(G):
    """The training kernels generated for use_viewdirs=False (MLP.forward(x, None), mip_nerf.py:99-110): native gradients vs the
    numpy emulation of the same dataflow (bf16 operands) and vs the fp32 oracle; extra_layer / view_layers get exact zeros."""
    from mipnerf_pl_amd.mlp_plan import Arch
    from mipnerf_pl_amd.mlp_train_plan import TrainPlan, emulate_train
    B, N = 5, 40
    params, enc, venc, d_raw = _mlp_case(B, N, seed=91, net_width_condition=256)
    raw, grads, enc_bf, v_bf = _run_native_mlp(G, params, enc, venc, d_raw, mlp_net_width_condition=256, use_viewdirs=False)
    tp = TrainPlan.build(Arch(net_width_condition=256, use_viewdirs=False))
    flatp = np.concatenate([v.ravel() for v in params.values()])
    S = B * N
    flat, seen, raw_em = emulate_train(tp, flatp, enc_bf.reshape(S, 96), np.zeros((S, 32), np.float32), d_raw.reshape(S, 4),
                                       round_bf16=True)
    e_raw = G.maxdiff(raw.reshape(S, 4), raw_em)
    og = orc.mlp_backward(params, enc, None, d_raw[..., :3], d_raw[..., 3:])
    off, worst_em, worst_or = 0, 0.0, 0.0
    for k, v in og.items():
        g = (grads[k] if grads[k] is not None else np.zeros_like(v)).ravel().astype(np.float64)
        em = flat[off:off + v.size].astype(np.float64)
        off += v.size
        if not np.any(v):
            assert not np.any(g), k
            continue
        worst_em = max(worst_em, np.linalg.norm(g - em) / max(np.linalg.norm(em), 1e-30))
        worst_or = max(worst_or, np.linalg.norm(g - v.ravel()) / max(np.linalg.norm(v.ravel()), 1e-30))
    G.record("native_mlp_bwd_noview", raw_vs_emul=e_raw, grad_rel_l2_vs_emul=worst_em, grad_rel_l2_vs_fp32=worst_or)
    assert worst_em <= 1e-2 and worst_or <= 0.2, (worst_em, worst_or)
    assert e_raw <= 2e-2 * max(1.0, float(np.abs(raw_em).max()))


def test_training_step_other_boundary_settings_vs_reference(G):
    """The settings of the boundary the main training golden leaves at their defaults (nerf_system.py:17-21, 95-111;
    mip_nerf.py:186-214): black background, disparity sampling, `loss.disable_multiscale_loss`, randomized=True with the
    reference's two draws replayed -- loss and every gradient vs the reference's autograd in fp32 mode; the bf16 autograd path and
    the one-call native step against that."""
    from mipnerf_pl_amd.system import DEFAULT_HPARAMS, MipNeRFSystem
    g = G.load_golden("train_options_40x64")
    params = orc.make_params(seed=int(g["param_seed"]), density_gain=float(g["density_gain"]))
    hp = dict(DEFAULT_HPARAMS)
    hp.update({'nerf.num_samples': int(g["num_samples"]), 'nerf.disparity': True, 'train.white_bkgd': False,
               'loss.disable_multiscale_loss': True})
    rays, gt = G.to_dev(G.rays_of(g)), torch.from_numpy(g["gt"]).to(DEV)
    t_rand, u_rand = torch.from_numpy(g["t_rand"]).to(DEV), torch.from_numpy(g["u_rand"]).to(DEV)
    grads = {}
    for precision in ("fp32", "bf16"):
        system = MipNeRFSystem(hp, precision=precision)
        system.load_state_dict({"mip_nerf.mlp." + k: torch.from_numpy(v.copy()) for k, v in params.items()})
        system = system.to(DEV)
        ret = system.mip_nerf(rays, True, False, t_rand=t_rand, u_rand=u_rand)
        if precision == "fp32":
            for lvl in range(2):
                for nm, val in zip(G.NAMES, ret[lvl]):
                    assert G.maxdiff(val, g[f"wb0_l{lvl}_{nm}"]) <= G.TOL_FP32[nm], (lvl, nm)
        loss = system.compute_loss(ret, rays, gt)[0]
        loss.backward()
        grads[precision] = (float(loss.detach()), {k: p.grad.detach().clone() for k, p in system.mip_nerf.mlp.named_parameters()})
    l32, g32 = grads["fp32"]
    assert abs(l32 - float(g["loss"])) <= 2e-5 * max(1.0, float(g["loss"]))
    worst = 0.0
    for k, gr in g32.items():
        a = gr.cpu().numpy().ravel()
        l2 = float(g["g_l2_" + k])
        stride = max(1, a.size // 64)
        es = float(np.max(np.abs(a[::stride][:64] - g["g_smp_" + k]))) / max(float(np.abs(g["g_smp_" + k]).max()), l2 / np.sqrt(a.size), 1e-12)
        worst = max(worst, es, abs(float(np.sqrt((a.astype(np.float64) ** 2).sum())) - l2) / max(l2, 1e-12))
        assert es <= 5e-3 and abs(float(np.sqrt((a.astype(np.float64) ** 2).sum())) - l2) <= 2e-3 * max(l2, 1e-12), (k, es)
    lb, gb = grads["bf16"]
    assert abs(lb - l32) <= 2e-2 * max(1.0, abs(l32))
    cos_worst = min(float((gb[k].double() * g32[k].double()).sum() / (gb[k].double().norm() * g32[k].double().norm()).clamp_min(1e-30))
                    for k in g32)
    assert cos_worst >= 0.97, cos_worst
    # the one-call native step with the same draws
    nsys = MipNeRFSystem(hp, precision="bf16")
    nsys.load_state_dict({"mip_nerf.mlp." + k: torch.from_numpy(v.copy()) for k, v in params.items()})
    nsys = nsys.to(DEV)
    sc, _ = nsys.mip_nerf.train_step_native(rays, gt, True, False, disable_multiscale_loss=True, t_rand=t_rand, u_rand=u_rand)
    assert abs(float(sc[0]) - lb) <= 1e-4 * max(1.0, abs(lb))
    for k, p in nsys.mip_nerf.mlp.named_parameters():
        assert G.maxdiff(p.grad, gb[k]) <= 2e-3 * max(1e-6, float(gb[k].abs().max())), k
    G.record("training_step_other_settings", worst_grad_rel_fp32=worst, loss_fp32=l32, loss_bf16=lb, worst_cos_bf16=cos_worst)


@pytest.mark.parametrize("B,N,randomized", [(37, 32, True), (50, 64, False), (21, 100, True), (64, 128, True), (11, 160, True), (9, 256, False), (5, 300, True),
                                            (3, 512, False), (2, 700, True)])      # round 5: 160 / 300 / 512 / 700 run FUSED too (K buckets 4, 8, 16)
def test_native_train_step_fused_tail_equals_per_stage_kernels(G, B, N, randomized):
    """Round 3: mipnerf_train_step runs pos_enc + the coarse fence posts as one launch and, per level, compositing + distloss
    (+ the next level's fence posts) as one launch (k_composite_train) instead of three; option 4 = 0 restores one launch per
    stage.  Same per-ray device functions: loss, every gradient and the returned outputs must be bit-identical."""
    from mipnerf_pl_amd import MipNerf
    rays_np = orc.synthetic_rays(B, seed=B + N, multiscale=True)
    params = orc.make_params(seed=3, density_gain=30.0)
    rays = G.to_dev(rays_np)
    gt = torch.rand(B, 3, device=DEV, generator=torch.Generator(device=DEV).manual_seed(N))
    gen = torch.Generator(device=DEV).manual_seed(B)
    t_rand = torch.rand(B, N + 1, device=DEV, generator=gen) if randomized else None
    u_rand = torch.rand(B, N + 1, device=DEV, generator=gen) if randomized else None
    res = []
    for fuse in (1, 0):
        model = G.make_model(params, N, "bf16")
        model.mlp.native(torch.device(DEV)).set_option(4, fuse)
        sc, outs = model.train_step_native(rays, gt, randomized, True, t_rand=t_rand, u_rand=u_rand, return_outputs=True)
        grads = torch.cat([p.grad.reshape(-1) for p in model.mlp.parameters()]).clone()
        res.append((sc.clone(), grads, [tuple(x.clone() for x in lv) for lv in outs]))
    (s1, g1, o1), (s0, g0, o0) = res
    assert torch.equal(s1, s0), (s1, s0)
    assert torch.equal(g1, g0)
    for lvl in range(2):
        for a, b in zip(o1[lvl], o0[lvl]):
            assert torch.equal(a, b), lvl

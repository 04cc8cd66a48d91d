"""Helpers shared by the -m gpu tests (the parity tests proper)."""
import json
import os

import numpy as np
import torch

from mipnerf_pl_amd import Rays
from oracle import mipnerf_oracle as orc

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, "gpurun_out")
DEV = "cuda:0"

# ---- stated tolerances ---------------------------------------------------------------------------
# fp32 mode (exact-fp32 MFMA): differences from the fp32 reference come from summation order
# (GEMM k-order, wave scans vs sequential cumsum) and libm ulps; 1-ulp changes of the resampled t move
# high-frequency IPE features by up to 3e-4, which reaches the outputs attenuated (measured with the
# oracle: rgb 7e-7, weights 7e-6).  Tolerance on every final output of MipNerf.forward:
TOL_FP32 = dict(rgb=5e-5, distance=2e-4, acc=5e-5, weights=5e-5, t_samples=2e-5)
# bf16 mode (bf16 operands, fp32 accumulate, 10 chained layers).  Level 0 measured on MI355X equals the
# numpy bf16 emulation (rgb 7e-4, distance 5e-3, weights 4e-3 on the C1 case; up to rgb 4e-3, acc 6e-3 on
# other seeds).  At level 1 the fine samples are drawn from the bf16 coarse weights, so t_samples move by up to
# ~0.04 and per-bin weights are no longer comparable bin-by-bin (up to 0.07); the rendered values stay close
# (rgb <= 1.1e-2, acc <= 2.1e-2, distance <= 8e-2) and the PSNR of bf16 vs reference renders is ~75 dB, which
# is the acceptance criterion (> 51.4 dB keeps a 35 dB render within 0.1 dB).
# round 2: tightened to ~1.5x the largest value measured over every bf16 case incl. the full-size goldens
# (rgb 1.1e-2, distance 7.9e-2, acc 2.1e-2, weights 7.1e-2, t_samples 4.0e-2)
TOL_BF16 = dict(rgb=2e-2, distance=0.12, acc=3e-2, weights=0.1, t_samples=6e-2)
NAMES = ("rgb", "distance", "acc", "weights", "t_samples")

# round 5: the catch-all TOL_BF16 above is 1.5x the worst case over ALL goldens, i.e. 4-20x what most single cases
# measure -- a regression that triples a case's error passed.  Per case: bound = 2 x the maximum MEASURED on MI355X for that case
# (max over both levels; profiles/r04z_parity.jsonl, re-measured in profiles/r05_parity.jsonl), floored where the measured value is
# ~0 (a floor of a few bf16 ulps of the quantity: errors that small are accumulation-order noise, not a level).  bf16 results are
# deterministic for a given build; a change of a kernel's summation order moves them by far less than 2x.
BF16_MEASURED = {
    "forward fwd_c1_256x64_xavier": dict(rgb=5.6e-4, distance=7.5e-4, acc=2.6e-4, weights=6.0e-5, t_samples=5.7e-4),
    "forward fwd_c1_256x64_trained": dict(rgb=1.8e-3, distance=2.1e-2, acc=9.7e-5, weights=7.1e-2, t_samples=1.9e-2),
    "forward fwd_ragged_100x128_trained": dict(rgb=1.9e-3, distance=2.5e-2, acc=8.6e-4, weights=3.1e-2, t_samples=3.0e-2),
    "forward fwd_unbounded_24x256_trained": dict(rgb=1.5e-3, distance=2.4e-2, acc=4.2e-7, weights=2.4e-2, t_samples=2.2e-2),
    "forward fwd_disparity_32x64_trained": dict(rgb=8.6e-3, distance=2.7e-2, acc=1.7e-2, weights=3.1e-2, t_samples=4.0e-2),
    "forward randomized": dict(rgb=1.1e-2, distance=7.9e-2, acc=2.1e-2, weights=1.3e-2, t_samples=1.8e-2),
    "full_size B=4096": dict(rgb=9.1e-3, distance=3.7e-2, acc=1.9e-2, weights=1.9e-2, t_samples=1.1e-2),
    "density_noise": dict(rgb=5.0e-4, distance=2.0e-3, acc=8.2e-4, weights=2.3e-3, t_samples=5.5e-3),
    "variant var_d6s3_48x64": dict(rgb=4.9e-3, distance=5.4e-2, acc=8.6e-3, weights=4.0e-2, t_samples=3.3e-2),
    "variant var_dc2_48x64": dict(rgb=6.4e-3, distance=4.1e-2, acc=1.2e-2, weights=2.7e-2, t_samples=3.6e-2),
    "variant var_noview_48x64": dict(rgb=7.7e-4, distance=9.2e-3, acc=4.2e-4, weights=7.7e-3, t_samples=1.2e-2),
    "variant var_w512_24x64": dict(rgb=3.0e-3, distance=1.1e-2, acc=4.5e-3, weights=5.3e-3, t_samples=9.8e-3),     # round 5: 60.1 dB
    "variant var_w100c40_48x64": dict(rgb=2.2e-3, distance=2.0e-2, acc=1.9e-3, weights=2.6e-2, t_samples=2.5e-2),
    "variant var_w128_48x64": dict(rgb=2.1e-3, distance=0.0, acc=3.5e-3, weights=2.8e-3, t_samples=1.1e-2),
    "variant var_w200c72_48x64": dict(rgb=7.4e-4, distance=1.2e-2, acc=5.5e-5, weights=8.0e-3, t_samples=7.6e-3),
    "ctor ctor_levels1_40x64": dict(rgb=9.3e-4, distance=0.0, acc=1.6e-3, weights=9.6e-4, t_samples=0.0),
    "ctor ctor_scalars_40x64": dict(rgb=7.6e-4, distance=2.6e-3, acc=1.3e-5, weights=3.2e-3, t_samples=2.0e-3),
    # level 0 of ctor_noint (level 1 is only required to be finite, see test_constructor_scalars_forward)
    "ctor ctor_noint_40x64 level0": dict(rgb=1.1e-3, distance=0.0, acc=1.5e-3, weights=1.0e-3, t_samples=0.0),
}
# PSNR of the bf16 fine-level render against the reference's render, measured (dB); asserted at measured - 6 dB = twice the error
BF16_PSNR_MEASURED = {"fwd_c1_256x64_xavier": 74.8, "fwd_c1_256x64_trained": 69.1, "fwd_ragged_100x128_trained": 68.2,
                      "fwd_unbounded_24x256_trained": 67.9, "fwd_disparity_32x64_trained": 56.6}
BF16_FLOOR = dict(rgb=4e-4, distance=2e-3, acc=4e-4, weights=2e-3, t_samples=2e-3)


def bf16_tol(case):
    """2 x the measured maxima of `case` (see BF16_MEASURED); the catch-all TOL_BF16 for a case without a record"""
    m = BF16_MEASURED.get(case)
    if m is None:
        return dict(TOL_BF16)
    return {k: min(TOL_BF16[k], max(2.0 * m[k], BF16_FLOOR[k])) for k in NAMES}


def tol_for(precision, case):
    return dict(TOL_FP32) if precision == "fp32" else bf16_tol(case)


def to_dev(rays_np):
    return Rays(*[torch.from_numpy(np.ascontiguousarray(a)).to(DEV) for a in rays_np])


def rays_of(g):
    return orc.Rays(*[g["rays_" + k] for k in orc.Rays._fields])


def load_golden(name):
    return dict(np.load(os.path.join(REPO, "tests", "golden", name + ".npz")))


def make_model(params, num_samples, precision, **kw):
    from mipnerf_pl_amd import MipNerf
    m = MipNerf(num_samples=num_samples, precision=precision, **kw)
    sd = {"mlp." + k: torch.from_numpy(v.copy()) for k, v in params.items()}
    missing, unexpected = m.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    return m.to(DEV)


def record(tag, **vals):
    """Append one JSON line of measured errors to gpurun_out/parity.jsonl (merged back by gpurun)."""
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "parity.jsonl"), "a") as f:
        f.write(json.dumps(dict(tag=tag, **{k: float(v) for k, v in vals.items()})) + "\n")


def maxdiff(a, b):
    a = a.detach().float().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().float().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64)))) if a.size else 0.0


import torch.nn.functional as F  # noqa: E402


def mlp_torch(mlp, samples_enc, viewdirs_enc, dtype):
    """models/mip_nerf.py:75-111 through torch ops (library GEMMs): plain-PyTorch fp32 reference of the MLP for the tests."""
    def lin(layer, x):
        return F.linear(x, layer.weight.to(dtype), layer.bias.to(dtype))
    num_samples = samples_enc.shape[1]
    inputs = samples_enc.to(dtype)
    x = inputs
    for i, layer in enumerate(mlp.layers):
        x = torch.relu(lin(layer[0], x))
        if i % mlp.skip_index == 0 and i > 0:
            x = torch.cat([x, inputs], dim=-1)
    raw_density = lin(mlp.density_layer, x)
    bottleneck = lin(mlp.extra_layer, x)
    vd = viewdirs_enc.to(dtype)[:, None, :].expand(-1, num_samples, -1)
    x = torch.cat([bottleneck, vd], dim=-1)
    for layer in mlp.view_layers:
        x = torch.relu(lin(layer[0], x))
    raw_rgb = lin(mlp.color_layer, x)
    return torch.cat([raw_rgb, raw_density], dim=-1).float()



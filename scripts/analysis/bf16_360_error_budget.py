"""Where does the bf16 error of the unbounded-scene model come from?    CPU only (numpy oracle + the numpy
model of the bf16 kernels: bf16 operands, fp32 accumulation, fp32 bias / ReLU).

The bench's unbounded bf16 sub-record reports 53.4 dB (fine-level rgb of the bf16 forward against the fp32 forward, fog weights, 8192 x 256)
where the standard model measures 72 dB.  This script runs the 360 oracle's forward with single sources of bf16 rounding switched on:

    enc      only the 672 encoding features rounded to bf16 (fp32 MLP)
    layer0   bf16 encoding + bf16 W0 / W5[:, 256:] (the two contractions over the encoding), rest fp32
    trunk    fp32 encoding contractions, every other layer in bf16
    all      the whole bf16 model (what the kernels compute, up to accumulation order)

and, for comparison, the standard model (96-wide diagonal encoding) with everything in bf16, on the same kind of field.
Usage: python scripts/analysis/bf16_360_error_budget.py [rays] [samples] [field: fog | trained]"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import synthetic_inputs as syn  # noqa: E402
from oracle import mipnerf360_oracle as o360  # noqa: E402
from oracle import mipnerf_oracle as orc  # noqa: E402

F32 = np.float32


def bf16(x):
    u = np.ascontiguousarray(x, F32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(F32).reshape(np.shape(x))


def mlp(params, enc, venc, mode):
    """orc.mlp_forward with selectable bf16 rounding; mode in {"fp32", "enc", "layer0", "trunk", "all"}"""
    r = bf16
    ident = lambda a: a                                                                       # noqa: E731
    r_enc = r if mode in ("enc", "layer0", "all") else ident
    r_w_enc = r if mode in ("layer0", "all") else ident        # weights that multiply the encoding
    r_trunk = r if mode in ("trunk", "all") else ident         # weights + activations of every other layer
    e = r_enc(enc)
    W0 = params["layers.0.0.weight"]
    x = np.maximum(e @ r_w_enc(W0).T + params["layers.0.0.bias"], 0)
    x = r_trunk(x) if mode != "layer0" else (r(x) if False else x)
    for i in range(1, 8):
        W = params[f"layers.{i}.0.weight"]
        if i == 5:
            z = r_trunk(x) @ r_trunk(W[:, :256]).T + e @ r_w_enc(W[:, 256:]).T + params[f"layers.{i}.0.bias"]
        else:
            z = r_trunk(x) @ r_trunk(W).T + params[f"layers.{i}.0.bias"]
        x = r_trunk(np.maximum(z, 0))
    dens = x @ r_trunk(params["density_layer.weight"]).T + params["density_layer.bias"]
    b = r_trunk(x @ r_trunk(params["extra_layer.weight"]).T + params["extra_layer.bias"])
    vd = np.broadcast_to(r_trunk(venc)[:, None, :], (x.shape[0], x.shape[1], venc.shape[-1]))
    h = r_trunk(np.maximum(np.concatenate([b, vd], -1) @ r_trunk(params["view_layers.0.0.weight"]).T + params["view_layers.0.0.bias"], 0))
    rgb = h @ r_trunk(params["color_layer.weight"]).T + params["color_layer.bias"]
    return rgb.astype(F32), dens.astype(F32)


def forward(params, rays, N, mode, unbounded):
    venc = orc.pos_enc(rays.viewdirs, 0, 4, True)
    ret = []
    t_inv = w = t = None
    for lvl in range(2):
        if unbounded:
            if lvl == 0:
                t_inv, t, mc = o360.sample_along_rays_360(rays.origins, rays.directions, rays.radii, N, rays.near, rays.far, False, contracted=True)
            else:
                wp = np.concatenate([w[:, :1], w, w[:, -1:]], -1)
                wmax = np.maximum(wp[:, :-1], wp[:, 1:])
                wblur = (F32(0.5) * (wmax[:, :-1] + wmax[:, 1:])).astype(F32) + F32(0.01)
                t_inv = orc.sorted_piecewise_constant_pdf(t_inv, wblur, t_inv.shape[-1], False)
                t = (F32(1) / t_inv).astype(F32)
                mc = o360.cast_rays_360(t, rays.origins, rays.directions, rays.radii, True)
            enc = o360.integrated_pos_enc_360(mc, 0, 16)
        else:
            if lvl == 0:
                t, mc = orc.sample_along_rays(rays.origins, rays.directions, rays.radii, N, rays.near, rays.far, False, False)
            else:
                t, mc = orc.resample_along_rays(rays.origins, rays.directions, rays.radii, t, w, False, resample_padding=0.01)
            enc = orc.integrated_pos_enc(mc, 0, 16)
        raw_rgb, raw_d = mlp(params, enc, venc, mode)
        rgb = (orc.sigmoid(raw_rgb) * F32(1.002) - F32(0.001)).astype(F32)
        comp, dist, acc, w = orc.volumetric_rendering(rgb, orc.softplus(raw_d - F32(1)), t, rays.directions, True)
        ret.append((comp, acc, w))
    return ret


def psnr(a, b):
    return float(-10 * np.log10(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2) + 1e-30))


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    field = sys.argv[3] if len(sys.argv) > 3 else "fog"
    for unbounded in (True, False):
        rays = syn.synthetic_rays(B, seed=71, unbounded=True)          # per-ray near / far for both (BASELINE configs[3] style)
        if field == "fog":
            params = syn.make_params(seed=17, density_gain=40.0, **(dict(xyz_dim=672) if unbounded else {}))
        else:
            f = np.load(os.path.join(REPO, "tests", "golden", "trained_field_360.npz" if unbounded else "trained_field.npz"))
            params = {k[2:]: f[k] for k in f.files if k.startswith("p_")}
            if unbounded:
                sys.path.insert(0, os.path.join(REPO, "tests"))
                z = np.load(os.environ.get("SCENE360_CACHE", "/tmp/scene360_rays.npz"))
                ids = np.random.default_rng(5).permutation(z["rgb"].shape[0])[:B]
                rays = syn.Rays(*[z["rays_" + k][ids] for k in syn.Rays._fields])
        ref = forward(params, rays, N, "fp32", unbounded)
        for mode in (("enc", "layer0", "trunk", "all") if unbounded else ("all",)):
            got = forward(params, rays, N, mode, unbounded)
            print(f"{'unbounded' if unbounded else 'standard '} {field} {B}x{N} mode={mode:7s} coarse rgb PSNR {psnr(got[0][0], ref[0][0]):6.2f} dB  "
                  f"max {np.abs(got[0][0] - ref[0][0]).max():.2e}   fine rgb PSNR {psnr(got[1][0], ref[1][0]):6.2f} dB  max {np.abs(got[1][0] - ref[1][0]).max():.2e}  "
                  f"acc max {np.abs(got[1][1] - ref[1][1]).max():.2e}", flush=True)

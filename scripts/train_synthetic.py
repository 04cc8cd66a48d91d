#!/usr/bin/env python3
"""End-to-end training on a procedurally generated scene (no dataset on the box): a fixed "teacher" Mip-NeRF renders
the ground-truth colours of rays generated on the device from random Blender-style cameras (datasets.py:214-263);
a freshly initialised student is trained on them with the reference's loss / optimizer / LR schedule
(nerf_system.py:95-121, 70-76).  Reports the PSNR curve of the native bf16 path (forward-with-save, dgrad, wgrad,
FlatAdam) and, with --compare-fp32, of the fp32 parity mode on the same batches -- the stand-in for "PSNR within
0.1 dB of the reference" that can run without Blender data.

    python scripts/train_synthetic.py [--steps 400] [--rays 4096] [--samples 64] [--compare-fp32]
"""
import argparse
import json
import math
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mipnerf_pl_amd import ops  # noqa: E402
from mipnerf_pl_amd.system import DEFAULT_HPARAMS, MipNeRFSystem  # noqa: E402


def cameras(n, W, H, seed):
    rng = np.random.default_rng(seed)
    focal = .5 * W / math.tan(.5 * 0.6911112070083618)
    recs = []
    for _ in range(n):
        z = rng.normal(size=3)
        z /= np.linalg.norm(z)                       # camera looks along -z towards the origin from radius 4
        up = np.array([0, 0, 1.0])
        x = np.cross(up, z)
        x /= np.linalg.norm(x) + 1e-9
        y = np.cross(z, x)
        c2w = np.eye(4, dtype=np.float32)
        c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = x, y, z, 4.0 * z
        recs.append(ops.camera_record(c2w, W, H, 2.0, 6.0, focal=focal))
    return torch.stack(recs)


def make_system(precision, num_samples, seed, fused):
    hp = dict(DEFAULT_HPARAMS)
    hp.update({"nerf.num_samples": num_samples, "optimizer.lr_delay_steps": 50, "optimizer.max_steps": 2000})
    torch.manual_seed(seed)
    system = MipNeRFSystem(hp, precision=precision).cuda()
    system.fused_adam = fused
    return system


def run(args, precision, batches, gts, val):
    system = make_system(precision, args.samples, seed=1, fused=(precision == "bf16"))
    (opt,), (sch,) = system.configure_optimizers()
    curve = []
    for it in range(args.steps):
        opt.zero_grad()
        loss = system.training_step((batches[it % len(batches)], gts[it % len(batches)]), it)
        loss.backward()
        opt.step()
        sch["scheduler"].step()
        if it % args.every == 0 or it == args.steps - 1:
            with torch.no_grad():
                ret = system(val[0], False, True)
                mse = torch.mean((ret[1][0] - val[1]) ** 2)
            curve.append((it, float(loss), float(-10 * torch.log10(mse))))
    return curve


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--samples", type=int, default=64)
    ap.add_argument("--every", type=int, default=50)
    ap.add_argument("--compare-fp32", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    W = H = 64
    cams = cameras(24, W, H, seed=3).to(dev)
    # teacher: trained-like weights (density head scaled so that there is opaque structure to learn)
    teacher = make_system("fp32", args.samples, seed=7, fused=False)
    with torch.no_grad():
        teacher.mip_nerf.mlp.density_layer.weight.mul_(30.0)
    g = torch.Generator(device="cpu").manual_seed(0)
    batches, gts = [], []
    for _ in range(16):
        ci = torch.randint(0, cams.shape[0] - 2, (args.rays,), generator=g).to(dev)
        pi = torch.randint(0, W * H, (args.rays,), generator=g).to(dev)
        rays = ops.generate_rays(cams, cam_idx=ci, pix_idx=pi)
        with torch.no_grad():
            gt = teacher(rays, False, True)[1][0].clone()
        batches.append(rays)
        gts.append(gt)
    vrays = ops.generate_rays(cams, num_rays=W * H, cam_idx=torch.full((W * H,), cams.shape[0] - 1, dtype=torch.int32, device=dev))
    with torch.no_grad():
        val = (vrays, teacher(vrays, False, True)[1][0].clone())
    out = {"config": vars(args)}
    out["bf16_native"] = run(args, "bf16", batches, gts, val)
    if args.compare_fp32:
        out["fp32_parity_mode"] = run(args, "fp32", batches, gts, val)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

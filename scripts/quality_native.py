#!/usr/bin/env python3
"""Native side of the quality stand-in (tests/test_gpu_quality.py) for several device-RNG seeds: test PSNR per scale after the
500-step loop on the procedural multi-scale scene.   usage: quality_native.py [--seeds 3] [--precision bf16]"""
import argparse
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import dataset_fixture as fx  # noqa: E402
import test_gpu_quality as tq  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seeds", type=int, default=3)
ap.add_argument("--precision", default="bf16")
a = ap.parse_args()
root = fx.write_multicam_scene(os.path.join(tempfile.mkdtemp(), "scene"))
from mipnerf_pl_amd import datasets as D  # noqa: E402
n_pix = len(D.Multicam(root, "train", True, "all_images", device="cuda:0"))
Q = dict(fx.QUALITY)
ids = fx.quality_batch_ids(n_pix, Q["steps"], Q["batch"], Q["id_seed"])
orig = torch.manual_seed
for s in range(a.seeds):
    # _train_and_eval seeds the device RNG with 1234 after the init: shift it per run
    calls = {"n": 0}

    def seeded(v, _s=s):
        calls["n"] += 1
        return orig(v if calls["n"] == 1 else v + 1000 * _s)
    torch.manual_seed = seeded
    cms = torch.cuda.manual_seed
    torch.cuda.manual_seed = lambda v, _s=s: cms(v + 1000 * _s)
    losses, psnrs, per_scale = tq._train_and_eval(root, a.precision, Q, Q["steps"], ids)
    torch.manual_seed, torch.cuda.manual_seed = orig, cms
    print(json.dumps({"seed": s, "precision": a.precision, "test_psnr_per_scale": [round(x, 3) for x in per_scale], "mean": round(float(np.mean(per_scale)), 3),
                      "tail_loss": float(losses[-50:].mean()), "tail_train_psnr": float(psnrs[-50:].mean())}), flush=True)

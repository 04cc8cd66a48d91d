#!/usr/bin/env python3
"""A LONG training run of the native bf16 path (one-graph step) against the fp32 parity mode (autograd; pinned to the reference's gradients at 1e-3)
on the procedural multi-scale scene of tests/test_gpu_quality.py -- the reference itself trains at 16 s per step on the host, so beyond the
500-step golden runs the fp32 mode stands in for it.  Same pixel ids per step, same initialisation, `--seeds` draw seeds per precision; prints
one JSON line per run (test PSNR per scale after the last step) and a summary line.
usage: quality_long.py [--steps 8000] [--seeds 2] [--precisions bf16 fp32]"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import dataset_fixture as fx  # noqa: E402
import test_gpu_quality as tq  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=8000)
    ap.add_argument("--seeds", type=int, default=2)
    ap.add_argument("--precisions", nargs="+", default=["bf16", "fp32"])
    a = ap.parse_args()
    root = fx.write_multicam_scene(os.path.join(tempfile.mkdtemp(), "scene"))
    from mipnerf_pl_amd import datasets as D
    n_pix = len(D.Multicam(root, "train", True, "all_images", device="cuda:0"))
    Q = dict(fx.QUALITY)
    Q.update(steps=a.steps, max_steps=a.steps, lr_delay_steps=max(50, a.steps // 20))
    ids = fx.quality_batch_ids(n_pix, a.steps, Q["batch"], Q["id_seed"])
    res = {}
    for precision in a.precisions:
        for s in range(a.seeds):
            t0 = time.time()
            losses, psnrs, per_scale = tq._train_and_eval(root, precision, Q, a.steps, ids, draw_seed=1234 + s)
            rec = {"precision": precision, "draw_seed": 1234 + s, "steps": a.steps, "test_psnr_per_scale": [round(x, 3) for x in per_scale],
                   "mean": round(float(np.mean(per_scale)), 3), "tail_loss": float(losses[-100:].mean()), "tail_train_psnr": float(psnrs[-100:].mean()),
                   "seconds": round(time.time() - t0, 1)}
            res.setdefault(precision, []).append(rec)
            print(json.dumps(rec), flush=True)
    summ = {p: round(float(np.mean([r["mean"] for r in v])), 3) for p, v in res.items()}
    if "bf16" in summ and "fp32" in summ:
        summ["bf16_minus_fp32_db"] = round(summ["bf16"] - summ["fp32"], 3)
    print(json.dumps({"summary": summ, "steps": a.steps, "rays_per_step": Q["batch"], "samples": Q["num_samples"]}))


if __name__ == "__main__":
    main()

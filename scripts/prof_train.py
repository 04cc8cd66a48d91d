#!/usr/bin/env python3
"""Times the three native MLP training kernels on synthetic data of BASELINE configs[1] size with HIP events
(torch.cuda.Event on the launch stream; one HIP runtime), optionally with custom wgrad splits.

    python scripts/prof_train.py [--rays 4096] [--samples 128] [--iters 10] [--experiments]
"""
import argparse
import ctypes as C
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mipnerf_pl_amd import MipNerf, _lib as L  # noqa: E402
from mipnerf_pl_amd.mlp_train_plan import TrainPlan  # noqa: E402


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--samples", type=int, default=128)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--experiments", action="store_true")
    ap.add_argument("--job-ints", type=int, default=0, help="experiments with MIPNERF_LIB=<another build>: int32 per row of ITS job table")
    ap.add_argument("--wgrad-only", action="store_true", help="time only the wgrad kernel (for rocprofv3 --pmc passes)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    B, N = args.rays, args.samples
    M = B * N
    torch.manual_seed(0)
    model = MipNerf(num_samples=N, precision="bf16").to(dev)
    with torch.no_grad():
        model.mlp.density_layer.weight.mul_(40.0)
    nctx = model.mlp.native(dev)
    h = nctx.handle
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    enc = (torch.rand(B, N, 96, device=dev) * 2 - 1).to(torch.bfloat16)
    venc = torch.zeros(B, 32, device=dev)
    venc[:, :27] = torch.randn(B, 27, device=dev)
    venc = venc.to(torch.bfloat16)
    d_raw = torch.randn(B, N, 4, device=dev) * 1e-3
    sz = nctx.train_sizes(M)
    act = torch.empty(sz[0], dtype=torch.uint8, device=dev)
    masks = torch.empty(sz[1], dtype=torch.uint8, device=dev)
    delta = torch.empty(sz[2], dtype=torch.uint8, device=dev)
    raw = torch.empty(B, N, 4, device=dev)
    rgbs = torch.empty_like(raw)
    grad = torch.empty(612740, device=dev)

    def partials():
        return torch.empty(nctx.train_sizes(M)[3], dtype=torch.uint8, device=dev)
    part = partials()

    def fwd():
        L.check(lib.mipnerf_mlp_forward_train(h, M, N, enc.data_ptr(), venc.data_ptr(), rgbs.data_ptr(), raw.data_ptr(),
                                              act.data_ptr(), masks.data_ptr(), st))

    def dgrad():
        L.check(lib.mipnerf_mlp_dgrad(h, M, d_raw.data_ptr(), masks.data_ptr(), delta.data_ptr(), st))

    def wgrad():
        L.check(lib.mipnerf_mlp_wgrad(h, M, act.data_ptr(), delta.data_ptr(), part.data_ptr(), grad.data_ptr(), 0, st))

    def wgrad_noreduce():
        L.check(lib.mipnerf_mlp_wgrad(h, M, act.data_ptr(), delta.data_ptr(), part.data_ptr(), None, 0, st))

    def infer():
        L.check(lib.mipnerf_mlp_forward(h, M, N, enc.data_ptr(), venc.data_ptr(), L.PREC_BF16, rgbs.data_ptr(), None, st))
    fwd()
    dgrad()
    torch.cuda.synchronize()
    if args.wgrad_only:
        for _ in range(args.iters):
            wgrad_noreduce()
        torch.cuda.synchronize()
        return
    tp = TrainPlan.build()
    gb = lambda b: b * (M / 32) / 1e9
    res = dict(M=M, infer_ms=timed(infer, args.iters), trainfwd_ms=timed(fwd, args.iters), dgrad_ms=timed(dgrad, args.iters),
               wgrad_ms=timed(wgrad, args.iters), wgrad_noreduce_ms=timed(wgrad_noreduce, args.iters),
               trainfwd_GB=gb(tp.NH * 2048 + tp.NMASK * 1024 + 32 * (192 + 16 + 16)), dgrad_GB=gb(tp.NG * 2048 + tp.NMASK * 1024 + 512),
               wgrad_GB=gb(sum(len(j.a_blocks) + len(j.b_blocks) for j in tp.jobs) * 2048))
    for k in ("trainfwd", "dgrad", "wgrad"):
        res[k + "_TBps"] = res[k + "_GB"] / res[k + "_ms"]
    print(json.dumps(res), flush=True)
    if not args.experiments:
        return
    names = [j.name for j in tp.jobs]
    nj = len(names)
    if args.job_ints:       # A/B against a library built from another commit: take the job list from ITS table
        n = int(lib.mipnerf_debug_table_variant(0, 5, None, 0))
        buf = (C.c_int32 * n)()
        lib.mipnerf_debug_table_variant(0, 5, buf, n)
        jt = np.array(buf[:]).reshape(-1, args.job_ints)
        nj = jt.shape[0]
        names = [f"job{j}" for j in range(nj)]

        class _J:
            pass
        tp.jobs = []
        for r in jt:
            j = _J()
            j.a_blocks, j.b_blocks = list(range(int(r[0]))), list(range(int(r[1])))
            tp.jobs.append(j)

    def set_splits(sp):
        nonlocal part
        arr = (C.c_int32 * nj)(*sp)
        L.check(lib.mipnerf_set_wgrad_splits(h, arr))
        part = partials()
    # (a) every job alone on 16 workgroups: time per stage (wave tile) of one workgroup
    for j in range(nj):
        sp = [0] * nj
        sp[j] = 16
        set_splits(sp)
        t = timed(wgrad_noreduce, 5)
        stages = (M // 32) / 16
        print(json.dumps(dict(exp="job_alone_16wg", job=names[j], ms=t, us_per_stage=t * 1e3 / stages,
                              blocks=len(tp.jobs[j].a_blocks) + len(tp.jobs[j].b_blocks))), flush=True)
    # (b) whole job list under different split policies
    cost = np.array([len(j.a_blocks) + len(j.b_blocks) for j in tp.jobs], float)
    for total in (248, 256, 384, 496, 512, 744):
        sp = np.maximum(1, np.floor(cost / cost.sum() * total)).astype(int)
        set_splits(list(sp))
        print(json.dumps(dict(exp="bytes_proportional", total=int(sp.sum()), ms=timed(wgrad_noreduce, 5))), flush=True)
    for each in (18, 36):
        set_splits([each] * nj)
        print(json.dumps(dict(exp="equal_splits", total=each * nj, ms=timed(wgrad_noreduce, 5))), flush=True)
    # (b2) round 3 (merged skip-layer job: 19 blocks): workgroups per job ~ blocks + c0, c0 = the fixed cost of a stage in block units
    for total in (256,):
        for c0 in (0, 4, 8, 12, 16, 24, 32, 64, 1e6):
            w = cost + c0
            sp = np.maximum(1, np.floor(w / w.sum() * total)).astype(int)
            order = np.argsort(-(w / w.sum() * total - sp))            # hand the remainder to the largest fractional parts
            for k in range(int(total - sp.sum())):
                sp[order[k % nj]] += 1
            set_splits(list(int(x) for x in sp))
            print(json.dumps(dict(exp="blocks_plus_c0", c0=c0, total=int(sp.sum()), splits=[int(x) for x in sp], ms=timed(wgrad_noreduce, 8))), flush=True)
    # (c) main jobs only / small jobs only at default proportions
    sp = np.maximum(1, np.floor(cost / cost.sum() * 256)).astype(int)
    big = [int(s) if cost[i] == 16 else 0 for i, s in enumerate(sp)]
    small = [int(s) if cost[i] != 16 else 0 for i, s in enumerate(sp)]
    set_splits(big)
    print(json.dumps(dict(exp="main_jobs_only", wgs=sum(big), ms=timed(wgrad_noreduce, 5))), flush=True)
    set_splits(small)
    print(json.dumps(dict(exp="small_jobs_only", wgs=sum(small), ms=timed(wgrad_noreduce, 5))), flush=True)
    L.check(lib.mipnerf_set_wgrad_splits(h, None))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Benchmark of the Mip-NeRF hot path on MI355X (contract: see the task statement).

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one MipNerf.forward (coarse + fine level) over one batch of synthetic lego-like rays
resident in HBM: BASELINE.json configs[1] = 4096 rays x (128 + 128) samples, bf16 MLP.
metric = ray-samples/sec (whole job, all ranks), ray-samples per step = B x N x num_levels.
Ranks shard rays (independent, no data-path collective): weak scaling.
Also reported: the roofline of the dominant kernel (the bf16 MFMA MLP), timed live with HIP events
through mipnerf_time_mlp, and the CPU baseline = the numpy oracle on a bounded sample (rank 0, N=1).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

FLOP_PER_SAMPLE = 1_220_608          # 2 x 610,304 MAC of the MLP per ray-sample
PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}   # MI355X_MICROARCH.md dense MFMA peaks


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--samples", type=int, default=128)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", default="inference", choices=["inference", "train", "render"],
                    help="inference (headline): MipNerf.forward; train: forward + loss + backward + grad all-reduce + Adam; "
                         "render: BASELINE configs[4], one 800x800 frame (640k rays) in 8192-ray chunks replayed from a "
                         "captured hipGraph, rays split over the ranks, rgb gathered")
    ap.add_argument("--autograd", action="store_true", help="train mode: training_step + loss.backward() through the custom "
                    "autograd Functions (what a Lightning loop does) instead of the single native mipnerf_train_step call")
    ap.add_argument("--torch-adam", action="store_true", help="train mode: torch.optim.Adam on per-tensor gradients "
                    "(what the reference configures) instead of the fused flat Adam kernel")
    ap.add_argument("--no-graph", action="store_true", help="render mode: eager chunk loop instead of the hipGraph")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from mipnerf_pl_amd import MipNerf, Rays, _lib as L
    from oracle import mipnerf_oracle as orc       # cpu_baseline leg + synthetic inputs only

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"[bench] WORLD_SIZE={world} != --gpus {args.gpus}; using WORLD_SIZE", file=sys.stderr)
    if os.environ.get("MIPNERF_BENCH_SHARE_GPU") == "1":     # plumbing test on a 1-GPU box: every rank on cuda:0
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" is RCCL on ROCm; the override exists only for the shared-GPU plumbing test (RCCL refuses two ranks per GPU)
        dist.init_process_group(os.environ.get("MIPNERF_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)

    B, N = args.rays, args.samples
    rays_np = orc.synthetic_rays(B, seed=100 + rank)
    params = orc.make_params(seed=0, density_gain=40.0)
    model = MipNerf(num_samples=N, precision=args.precision)
    model.load_state_dict({"mlp." + k: torch.from_numpy(v.copy()) for k, v in params.items()})
    model = model.to(dev)
    R = Rays(*[torch.from_numpy(a).to(dev) for a in rays_np])

    if args.mode == "train":
        from mipnerf_pl_amd.parallel import FlatGradAllReduce
        from mipnerf_pl_amd.system import DEFAULT_HPARAMS, MipNeRFSystem
        hp = dict(DEFAULT_HPARAMS)
        hp.update({"nerf.num_samples": N})
        system = MipNeRFSystem(hp, precision=args.precision)
        system.mip_nerf.load_state_dict(model.state_dict())
        system = system.to(dev)
        model = system.mip_nerf
        system.fused_adam = not args.torch_adam         # FlatAdam: flat parameter / gradient buffers, one Adam kernel
        (opt,), (sch,) = system.configure_optimizers()
        reduce_grads = FlatGradAllReduce(list(model.parameters()), mlp=model.mlp)
        gt = torch.rand(B, 3, device=dev)

        def step():
            opt.zero_grad(set_to_none=False)              # FlatAdam: no kernel, the next backward overwrites
            if args.autograd:
                loss = system.training_step((R, gt), 0)      # randomized=True, nerf_system.py:95-121
                loss.backward()
            else:
                loss = system.training_step_native((R, gt), 0)     # the same, one native call (mipnerf_train_step)
            reduce_grads()                                # one flat all-reduce over RCCL (no-op at world 1)
            opt.step()
            sch["scheduler"].step()
            return [(loss.detach().reshape(1),)]
    elif args.mode == "render":
        from mipnerf_pl_amd.parallel import gather_rendered, shard_bounds
        from mipnerf_pl_amd.system import DEFAULT_HPARAMS, MipNeRFSystem
        Himg = Wimg = 800
        lo, hi = shard_bounds(Himg * Wimg, rank, world)          # contiguous ray shard of this rank (strong scaling)
        nloc = hi - lo
        frame_np = orc.synthetic_rays(8192, seed=7)               # tiled: 640k distinct draws would dominate start-up
        reps = (nloc + 8191) // 8192
        FR = Rays(*[torch.from_numpy(np.tile(a, (reps, 1))[:nloc]).to(dev) for a in frame_np])
        hp = dict(DEFAULT_HPARAMS)
        hp.update({"nerf.num_samples": N, "val.chunk_size": 8192})
        system = MipNeRFSystem(hp, precision=args.precision)
        system.mip_nerf.load_state_dict(model.state_dict())
        system = system.to(dev)
        system.enable_hip_graph(not args.no_graph)
        model = system.mip_nerf
        img_rays = Rays(*[x.reshape(1, 1, nloc, -1) for x in FR])
        dummy = torch.zeros(1, 1, nloc, 3, device=dev)

        def step():
            _, fine, _ = system.render_image((img_rays, dummy))
            full = gather_rendered(fine.reshape(nloc, 3), Himg * Wimg)
            return [(full,)]
        B = Himg * Wimg // world      # for the samples-per-step accounting below (whole frame / world per rank)
    else:
        def step():
            with torch.no_grad():
                return model(R, False, True)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ctx = model.mlp.native(dev)
    if args.mode == "inference":
        ctx.set_option(2, 1)      # HIP events around every MLP launch of the timed region (launch stream)
    elif args.mode == "train" and args.precision == "bf16":
        ctx.set_option(2, 2)      # ... around every weight-gradient launch (the dominant, HBM-bound kernel of the step)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert bool(torch.isfinite(out[-1][0]).all())

    samples_per_step = B * N * model.num_levels
    value = samples_per_step * world * args.steps / dt

    # ---- roofline of the dominant kernel (bf16 / fp32 MFMA MLP): average duration of the MLP launches made
    # INSIDE the timed region, from HIP events recorded on the launch stream by the library ----
    roofline = None
    cpu_baseline = None
    import ctypes as C
    tot_ms, nl = C.c_double(), C.c_int64()
    L.check(L.lib().mipnerf_mlp_launch_stats(ctx.handle, C.byref(tot_ms), C.byref(nl)), "mlp_launch_stats")
    ctx.set_option(2, 0)
    if rank == 0 and nl.value > 0 and args.mode == "train":
        # dominant kernel of the training step: k_mlp_wgrad, HBM-bound; algorithmic bytes = the T-blocks it reads
        # (157 x 2 KiB per 32-sample wave tile, mlp_train_plan.py) + the fp32 partials it writes
        from mipnerf_pl_amd.mlp_train_plan import JOB_FLOATS, TrainPlan
        tp = TrainPlan.build()
        M = B * N
        blocks = sum(len(j.a_blocks) + len(j.b_blocks) for j in tp.jobs)
        nbytes = ((M + 255) // 256) * 8 * blocks * 2048 + (256 // len(tp.jobs)) * len(tp.jobs) * JOB_FLOATS * 4
        launch_ms = tot_ms.value / nl.value
        gbs = nbytes / (launch_ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": "k_mlp_wgrad", "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s",
                    "frac": round(gbs / 8000.0, 4), "traffic": 5321000000 if M == 524288 else None,
                    "launch_ms": round(launch_ms, 4), "launches_timed": int(nl.value), "samples_per_launch": M,
                    "bytes_per_sample": round(nbytes / M, 1)}
    elif rank == 0 and nl.value > 0:
        prec = model.precision
        M = B * N
        launch_ms = tot_ms.value / max(1, nl.value)
        tflops = FLOP_PER_SAMPLE * M / (launch_ms * 1e-3) / 1e12
        peak = PEAK_TFLOPS[args.precision]
        traffic = None
        pmc = os.path.join(REPO, "profiles", "mlp_pmc.json")     # HBM bytes/launch from rocprofv3 --pmc passes
        if os.path.exists(pmc):
            pj = json.load(open(pmc))
            if pj.get("precision") == args.precision and pj.get("samples_per_launch") == M:
                traffic = pj.get("hbm_bytes_per_launch")
        roofline = {"bound": "mfma", "kernel": "k_mlp_bf16" if prec == L.PREC_BF16 else "k_mlp_f32",
                    "achieved": round(tflops, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(tflops / peak, 4),
                    "traffic": traffic, "launch_ms": round(launch_ms, 4), "launches_timed": int(nl.value),
                    "samples_per_launch": M, "flop_per_sample": FLOP_PER_SAMPLE}
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and args.mode == "inference":
            # bounded sample of the same workload: 256 rays x N x 2 levels through the numpy oracle
            nb = 256
            sub = orc.Rays(*[a[:nb] for a in rays_np])
            orc.mipnerf_forward(params, orc.Rays(*[a[:32] for a in rays_np]), False, True, num_samples=N)  # warm BLAS
            reps, tcpu = 0, 0.0
            while tcpu < 10.0 and reps < 20:
                c0 = time.perf_counter()
                orc.mipnerf_forward(params, sub, False, True, num_samples=N)
                tcpu += time.perf_counter() - c0
                reps += 1
            cpu_baseline = {"value": round(nb * N * 2 * reps / tcpu, 1), "unit": "ray-samples/s",
                            "cores": os.cpu_count(), "kind": "port",
                            "sample": f"{reps} x oracle.mipnerf_forward on {nb} rays x {N} samples x 2 levels (numpy fp32, "
                                      f"BLAS threads = all cores)"}

    if rank == 0:
        line = {
            "metric": "ray-samples/sec", "value": round(value, 1), "unit": "ray-samples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong" if args.mode == "render" else "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": (f"BASELINE.json configs[1]: MipNerf.forward inference, {B} rays x ({N} coarse + {N} fine) "
                                    f"samples per GPU, 8x256 MLP, random-init trained-like weights") if args.mode == "inference"
                       else (f"BASELINE.json configs[4]: one 800x800 frame = 640,000 rays x ({N}+{N}) samples in 8192-ray chunks, "
                             f"{'eager chunk loop' if args.no_graph else 'chunk forward replayed from a captured hipGraph'}, rays split over "
                             f"{world} rank(s), rgb all-gathered") if args.mode == "render"
                       else (f"training step (forward randomized + loss incl. distloss + backward + grad all-reduce + Adam), "
                             f"{B} rays x ({N}+{N}) samples per GPU; MLP forward-with-save / dgrad / wgrad = native bf16 MFMA kernels"),
                       "mode": args.mode,
                       "rays_per_gpu": B, "samples_per_level": N, "levels": model.num_levels,
                       "parallelism": f"ray-split x{world} (no data-path collective)"},
            "per_gpu": round(value / world, 1),
            "roofline": roofline, "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Benchmark of the Mip-NeRF hot path on MI355X (contract: see the task statement).

    python bench.py --gpus 1 --steps 50 --warmup 5
    python bench.py --gpus 8                       # re-executes itself under torch.distributed.run, one rank per GPU (RCCL)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Headline (top level of the JSON line): a "step" is one MipNerf.forward (coarse + fine level) over one batch of synthetic
lego-like rays resident in HBM: BASELINE.json configs[1] = 4096 rays x (128 + 128) samples, bf16 MLP.
metric = ray-samples/sec (whole job, all ranks), ray-samples per step = B x N x num_levels.  Ranks shard rays
(independent, no data-path collective): weak scaling.  The default run (`--mode all`) also carries two sub-records,
each with its own timed region (same barrier + synchronize bracketing, max over ranks) and its own roofline:

  "train":  one training step = randomized forward + loss (incl. distloss) + backward + gradient all-reduce (RCCL, one
            flat 2.45 MB buffer) + Adam + LR schedule, 4096 rays x (128+128) per GPU (weak scaling);
  "render": BASELINE.json configs[4], one 800x800 frame (640k rays) in 8192-ray chunks, rays split over the ranks
            (strong scaling), rgb all-gathered.

Also reported: the roofline of the dominant kernel (bf16 MFMA MLP), timed live with HIP events on the launch stream,
a sustained figure (>= 2 s of steps after the timed region: the chip's clock settles on its power budget), and the CPU
baseline = the reference's own MipNerf.forward (staged under oracle/_ref by oracle/build_ref.py) on the host cores,
rank 0, N=1 only (the numpy port when nothing is staged).
"""
import argparse
import json
import math
import os
import socket
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

FLOP_PER_SAMPLE = 1_220_608          # 2 x 610,304 MAC of the MLP per ray-sample (forward)
FLOP_PER_SAMPLE_TRAIN = 3_556_608    # forward + dgrad + wgrad
CURRENT_ROUND = 6          # profiles/mlp_pmc.json must carry this round's PMC passes (VERDICT r03 hygiene): bump it and re-run scripts/pmc_traffic.sh every round
PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}   # MI355X_MICROARCH.md dense MFMA peaks


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--samples", type=int, default=128)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-lightning-route", action="store_true", help="train: skip the lightning_route sub-record (PMC passes: its per-level "
                    "launches would mix into the per-kernel means)")
    ap.add_argument("--mode", default="all", choices=["all", "inference", "train", "render"],
                    help="all (default): headline inference + train and render sub-records; inference / train / render: "
                         "only that workload, reported at the top level")
    ap.add_argument("--autograd", action="store_true", help="train: training_step + loss.backward() through the custom "
                    "autograd Functions (what a Lightning loop does) instead of the single native mipnerf_train_step call")
    ap.add_argument("--torch-adam", action="store_true", help="train: torch.optim.Adam on per-tensor gradients "
                    "(what the reference configures) instead of the fused flat Adam kernel")
    ap.add_argument("--render-chunk", type=int, default=8192, help="render: rays per chunk of the frame (val.chunk_size)")
    ap.add_argument("--no-graph", action="store_true", help="render / train: eager launches instead of the captured hipGraph")
    ap.add_argument("--preheat-seconds", type=float, default=3.0, help="untimed steps for this long BEFORE the W warm-up steps of every "
                    "timed region: the package reaches its steady clock / power state (the first tens of steps after an idle period "
                    "run 3-6 %% slower than the sustained rate); reported in config.preheat_seconds, 0 = off")
    ap.add_argument("--ceiling-seconds", type=float, default=1.2, help="all / inference at N=1: seconds per variant of the in-process MFMA "
                    "ceiling measurement (0 = skip)")
    ap.add_argument("--no-fp32", action="store_true", help="all: skip the fp32 configs[3] sub-record")
    ap.add_argument("--sustain-seconds", type=float, default=2.0, help="inference: extra steps after the timed region")
    ap.add_argument("--host-rays", action="store_true", help="inference: the 52 B/ray batch comes from pinned host memory every step "
                    "(what the reference's DataLoader hands over): the PCIe-inclusive rate quoted in DESIGN.md, never `value` of the default line")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: become N ranks, one per GPU, under
    torch.distributed.run on this node (what the reference's train.py does with devices=num_gpus + DDPPlugin,
    train.py:56-60)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)


class Env:
    pass


def setup(args):
    import torch
    import torch.distributed as dist
    e = Env()
    e.rank = int(os.environ.get("RANK", "0"))
    e.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    e.world = int(os.environ.get("WORLD_SIZE", "1"))
    if e.world != args.gpus:
        raise SystemExit(f"[bench] WORLD_SIZE={e.world} but --gpus {args.gpus}: launch with --nproc-per-node {args.gpus} "
                         f"(or run `python bench.py --gpus {args.gpus}` without a launcher: it starts the ranks itself)")
    e.share = os.environ.get("MIPNERF_BENCH_SHARE_GPU") == "1"     # plumbing test on a 1-GPU box: every rank on cuda:0
    if not e.share and torch.cuda.device_count() < e.world:
        raise SystemExit(f"[bench] --gpus {args.gpus} needs {e.world} visible GPUs, found {torch.cuda.device_count()}")
    if e.share:
        e.local_rank = 0
    torch.cuda.set_device(e.local_rank)
    e.dev = torch.device("cuda", e.local_rank)
    if e.world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" is RCCL on ROCm; the override exists only for the shared-GPU plumbing test (RCCL refuses two ranks per GPU)
        dist.init_process_group(os.environ.get("MIPNERF_BENCH_BACKEND", "nccl"), rank=e.rank, world_size=e.world)
        assert dist.get_world_size() == args.gpus
    elif os.environ.get("MIPNERF_FORCE_COLLECTIVE_PATH") == "1":
        # a 1-rank RCCL communicator: the training record then runs graph A -> all_reduce -> graph B like every world > 1 run
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            s_ = socket.socket()
            s_.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(s_.getsockname()[1])
            s_.close()
        dist.init_process_group(os.environ.get("MIPNERF_BENCH_BACKEND", "nccl"), rank=0, world_size=1)
    return e


def rank_report(e):
    """Who ran: per rank its device, so that a multi-GPU record explains itself -- "RCCL saw N ranks on N distinct GPUs".
    Collective when world > 1 (all ranks call it)."""
    import torch
    import torch.distributed as dist
    p = torch.cuda.get_device_properties(e.dev)
    me = {"rank": e.rank, "local_rank": e.local_rank, "device": str(e.dev), "name": torch.cuda.get_device_name(e.dev),
          "gcn_arch": getattr(p, "gcnArchName", None), "compute_units": p.multi_processor_count, "hbm_gib": round(p.total_memory / 2 ** 30, 1),
          "pci_bus_id": getattr(p, "pci_bus_id", None), "uuid": str(getattr(p, "uuid", "")) or None, "pid": os.getpid()}
    ranks = [me]
    if e.world > 1:
        ranks = [None] * e.world
        dist.all_gather_object(ranks, me)
    try:
        ver = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:  # noqa: BLE001
        ver = None
    ids = [(r["uuid"] or r["pci_bus_id"] or r["device"]) for r in ranks]
    return {"world_size": e.world, "backend": (dist.get_backend() if dist.is_initialized() else None),
            "rccl_version": ver, "distinct_devices": len(set(ids)), "shared_gpu_plumbing_test": bool(e.share), "ranks": ranks}


PREHEAT = {"seconds": 0.0}


def preheat(step, e=None):
    """Untimed steps for about --preheat-seconds (steady clock / power state of the package).  The step COUNT is agreed between the
    ranks (a step may contain a collective: every rank must run the same number): 8 probe steps, the slowest rank's time decides."""
    import torch
    if PREHEAT["seconds"] <= 0:
        return
    t0 = time.perf_counter()
    for _ in range(8):
        step()
    torch.cuda.synchronize()
    per = max((time.perf_counter() - t0) / 8, 1e-6)
    if e is not None and e.world > 1:
        import torch.distributed as dist
        tt = torch.tensor([per], device=e.dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        per = float(tt.item())
    n = int(min(max(PREHEAT["seconds"] / per - 8, 0), 20000))
    for i in range(n):
        step()
        if i % 64 == 63:
            torch.cuda.synchronize()
    torch.cuda.synchronize()


def timed(e, step, warmup, steps, return_local=False):
    """W untimed steps, then exactly K steps between barrier + synchronize on both sides; max over ranks (seconds).
    return_local: also this rank's own elapsed time, stopped after its OWN synchronize, before the closing barrier."""
    import torch
    import torch.distributed as dist

    def barrier():
        if e.world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    out = None
    for _ in range(warmup):
        out = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    if return_local:
        torch.cuda.synchronize()
    dt_local = time.perf_counter() - t0
    barrier()
    dt = time.perf_counter() - t0
    if e.world > 1:
        tt = torch.tensor([dt], device=e.dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if return_local:
        return dt, out, dt_local
    return dt, out


def launch_stats(ctx):
    import ctypes as C
    from mipnerf_pl_amd import _lib as L
    tot_ms, nl = C.c_double(), C.c_int64()
    L.check(L.lib().mipnerf_mlp_launch_stats(ctx.handle, C.byref(tot_ms), C.byref(nl)), "mlp_launch_stats")
    return tot_ms.value, int(nl.value)


def make_model(args, e, N):
    import torch
    import synthetic_inputs as syn
    from mipnerf_pl_amd import MipNerf
    params = syn.make_params(seed=0, density_gain=40.0)
    model = MipNerf(num_samples=N, precision=args.precision)
    model.load_state_dict({"mlp." + k: torch.from_numpy(v.copy()) for k, v in params.items()})
    return model.to(e.dev), params


# ---------------------------------------------------------------------------------------------------------------------
def run_inference(args, e):
    import torch
    import synthetic_inputs as syn
    from mipnerf_pl_amd import Rays, _lib as L
    B, N = args.rays, args.samples
    rays_np = syn.synthetic_rays(B, seed=100 + e.rank)
    model, params = make_model(args, e, N)
    R = Rays(*[torch.from_numpy(a).to(e.dev) for a in rays_np])
    H = Rays(*[torch.from_numpy(a).pin_memory() for a in rays_np]) if args.host_rays else None

    def step_eager():
        with torch.no_grad():
            if H is not None:      # host batch -> device inside the timed region (async copies on the launch stream)
                return model(Rays(*[t.to(e.dev, non_blocking=True) for t in H]), False, True)
            return model(R, False, True)
    # round 5: the step is ONE captured hipGraph (model.GraphedForward) over the batch resident in HBM -- same launches, same bits,
    # no Python between the kernels; --no-graph / --host-rays keep the eager launches
    gf = None
    if H is None and not args.no_graph:
        from mipnerf_pl_amd.model import GraphedForward
        gf = GraphedForward(model, B, True, e.dev)
        for dst, src in zip(gf.static_in, R):
            dst.copy_(src)
        ref_out = step_eager()
        got_out = gf.replay()
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for la, lb in zip(got_out, ref_out) for a, b in zip(la, lb)), "graph replay != eager forward"

    def step():
        return gf.replay() if gf is not None else step_eager()
    step()
    ctx = model.mlp.native(e.dev)
    preheat(step, e)
    for _ in range(args.warmup):
        step()
    dt, out = timed(e, step, 0, args.steps)
    # launch duration of the MLP kernel: HIP events around every MLP launch (on the launch stream).  Events cannot be read per replay
    # inside a graph, so when the timed steps were graph replays the SAME K steps run once more as eager launches right behind them,
    # instrumented (its own barrier-to-barrier time is reported as `eager_ms_per_step`)
    ctx.set_option(2, 1)
    if gf is not None and gf.graph:
        dt_eager, _ = timed(e, step_eager, 0, args.steps)
    else:
        dt_eager = None
        ctx.set_option(2, 0)
        ctx.set_option(2, 1)
        dt, out = timed(e, step, 0, args.steps)       # eager steps: instrument the timed region itself, as in rounds 1-4
    tot_ms, nl = launch_stats(ctx)
    ctx.set_option(2, 0)
    assert bool(torch.isfinite(out[-1][0]).all())
    M = B * N
    samples_per_step = B * N * model.num_levels
    value = samples_per_step * e.world * args.steps / dt
    peak = PEAK_TFLOPS[args.precision]
    kname = "k_mlp_bf16" if model.precision == L.PREC_BF16 else "k_mlp_f32r (register-resident, generated; round 3: the LDS-resident k_mlp_f32)"
    roofline = None
    if nl > 0:
        launch_ms = tot_ms / nl
        tflops = FLOP_PER_SAMPLE * M / (launch_ms * 1e-3) / 1e12
        traffic, tsrc = pmc_traffic("inference", args.precision, M)
        roofline = {"bound": "mfma", "kernel": kname, "achieved": round(tflops, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(tflops / peak, 4), "traffic": traffic, "traffic_source": tsrc,
                    "launch_ms": round(launch_ms, 4), "launches_timed": nl, "samples_per_launch": M,
                    "flop_per_sample": FLOP_PER_SAMPLE,
                    "launch_timing": ("HIP events around every MLP launch of the same K steps run as eager launches right behind the timed graph "
                                      "replays" if dt_eager is not None else "HIP events around every MLP launch of the timed region")}
    # sustained: keep stepping for >= sustain-seconds; the package settles on its power budget (DVFS)
    sustained = None
    if args.sustain_seconds > 0:
        per = max(1, int(0.25 / max(dt / args.steps, 1e-5)))
        t_end = time.perf_counter() + args.sustain_seconds
        while time.perf_counter() < t_end:          # un-instrumented heat-up
            for _ in range(per):
                step()
            torch.cuda.synchronize()
        ks = max(args.steps, 20)
        if gf is not None and gf.graph:               # graph replays carry no events: time them, then the same steps eagerly with events
            dts, _ = timed(e, step, 0, ks)
            ctx.set_option(2, 1)
            timed(e, step_eager, 0, ks)
        else:
            ctx.set_option(2, 1)
            dts, _ = timed(e, step, 0, ks)
        tot2, nl2 = launch_stats(ctx)
        sustained = {"after_seconds": args.sustain_seconds, "steps": ks, "ms_per_step": round(dts / ks * 1e3, 4),
                     "value": round(samples_per_step * e.world * ks / dts, 1)}
        if nl2 > 0:
            tf2 = FLOP_PER_SAMPLE * M / (tot2 / nl2 * 1e-3) / 1e12
            sustained.update({"launch_ms": round(tot2 / nl2, 4), "achieved": round(tf2, 2), "frac": round(tf2 / peak, 4)})
    ctx.set_option(2, 0)
    rec = {"value": round(value, 1), "ms_per_step": round(dt / args.steps * 1e3, 4), "steps": args.steps, "warmup": args.warmup,
           "scaling": "weak", "roofline": roofline, "sustained": sustained,
           "config": {"workload": (f"BASELINE.json configs[1]: MipNerf.forward inference, {B} rays x ({N} coarse + {N} fine) "
                                   f"samples per GPU, 8x256 MLP, random-init trained-like weights"),
                      "mode": "inference", "preheat_seconds": PREHEAT["seconds"], "rays_per_gpu": B, "samples_per_level": N, "levels": model.num_levels,
                      "hip_graph": bool(gf is not None and gf.graph), "hip_graph_capture_error": gf.capture_error if gf is not None else None,
                      "eager_ms_per_step": None if dt_eager is None else round(dt_eager / args.steps * 1e3, 4),
                      "inputs": "pinned host memory, copied every step (PCIe-inclusive)" if args.host_rays else "resident in HBM",
                      "parallelism": f"ray-split x{e.world} (no data-path collective)"}}
    return rec, (rays_np, params)


def run_trained_field(args, e):
    """The headline kernel on a TRAINED field instead of the saturated fog: the work of the forward is
    data-independent but the clock the package grants is not, so the quoted `frac` is shown on the weights of tests/golden/trained_field.npz
    (the reference trained on the procedural multi-scale scene) with 4096 rays OF THAT SCENE (a third empty, nearly half opaque, a fifth
    soft) -- the inputs of the golden fulltrained_c2_4096x128, whose reference outputs also give a live parity figure."""
    import numpy as np
    import torch
    from mipnerf_pl_amd import MipNerf, Rays, _lib as L
    gdir = os.path.join(REPO, "tests", "golden")
    g = np.load(os.path.join(gdir, "fulltrained_c2_4096x128.npz"))
    f = np.load(os.path.join(gdir, str(g["field"]) + ".npz"))
    params = {k[2:]: f[k] for k in f.files if k.startswith("p_")}
    N, B = int(g["num_samples"]), int(g["batch"])
    model = MipNerf(num_samples=N, precision=args.precision)
    model.load_state_dict({"mlp." + k: torch.from_numpy(v.copy()) for k, v in params.items()})
    model = model.to(e.dev)
    R = Rays(*[torch.from_numpy(np.ascontiguousarray(g["rays_" + k])).to(e.dev) for k in Rays._fields])

    def step_eager():
        with torch.no_grad():
            return model(R, False, True)
    gf = None
    if not args.no_graph:
        from mipnerf_pl_amd.model import GraphedForward
        gf = GraphedForward(model, B, True, e.dev)
        for dst, src in zip(gf.static_in, R):
            dst.copy_(src)

    def step():
        return gf.replay() if gf is not None else step_eager()
    out = step()
    ctx = model.mlp.native(e.dev)
    preheat(step, e)
    for _ in range(args.warmup):
        step()
    dt, out = timed(e, step, 0, args.steps)
    ctx.set_option(2, 1)
    timed(e, step_eager, 0, args.steps)
    tot_ms, nl = launch_stats(ctx)
    ctx.set_option(2, 0)
    rgb = out[-1][0].cpu().numpy().astype(np.float64)
    psnr = float(-10.0 * np.log10(np.mean((rgb - g["l1_rgb"].astype(np.float64)) ** 2) + 1e-30))
    acc = g["l1_acc"]
    M = B * N
    peak = PEAK_TFLOPS[args.precision]
    launch_ms = tot_ms / max(nl, 1)
    tflops = FLOP_PER_SAMPLE * M / (launch_ms * 1e-3) / 1e12
    return {"value": round(B * N * model.num_levels * e.world * args.steps / dt, 1), "ms_per_step": round(dt / args.steps * 1e3, 4), "steps": args.steps,
            "warmup": args.warmup, "scaling": "weak",
            "roofline": {"bound": "mfma", "kernel": "k_mlp_bf16" if model.precision == L.PREC_BF16 else "k_mlp_f32r", "achieved": round(tflops, 2),
                         "peak": peak, "unit": "TFLOP/s", "frac": round(tflops / peak, 4), "traffic": None, "launch_ms": round(launch_ms, 4),
                         "launches_timed": nl, "samples_per_launch": M},
            "parity": {"psnr_fine_rgb_vs_reference_db": round(psnr, 2), "max_abs_fine_rgb": float(np.abs(rgb - g["l1_rgb"]).max()),
                       "golden": "tests/golden/fulltrained_c2_4096x128.npz (the unmodified reference's outputs on these rays)"},
            "config": {"workload": (f"the headline forward on a TRAINED field: {B} rays of the procedural multi-scale scene x ({N} + {N}) samples, weights of "
                                    f"tests/golden/trained_field.npz; rays {float((acc < 0.05).mean()):.2f} empty / {float((acc > 0.95).mean()):.2f} opaque"),
                       "mode": "trained_field", "hip_graph": bool(gf is not None and gf.graph), "rays_per_gpu": B, "samples_per_level": N}}


def run_train(args, e):
    import torch
    import synthetic_inputs as syn
    from mipnerf_pl_amd import Rays
    from mipnerf_pl_amd.parallel import FlatGradAllReduce
    from mipnerf_pl_amd.system import DEFAULT_HPARAMS, MipNeRFSystem
    B, N = args.rays, args.samples
    rays_np = syn.synthetic_rays(B, seed=100 + e.rank)
    R = Rays(*[torch.from_numpy(a).to(e.dev) for a in rays_np])
    model0, _ = make_model(args, e, N)
    hp = dict(DEFAULT_HPARAMS)
    hp.update({"nerf.num_samples": N})
    system = MipNeRFSystem(hp, precision=args.precision)
    system.mip_nerf.load_state_dict(model0.state_dict())
    system = system.to(e.dev)
    model = system.mip_nerf
    # FlatAdam: flat parameter / gradient buffers, one Adam kernel (bf16 native path); fp32 parity mode trains through
    # autograd with the optimiser the reference configures (torch.optim.Adam)
    system.fused_adam = (not args.torch_adam) and args.precision == "bf16"
    (opt,), (sch,) = system.configure_optimizers()
    reduce_grads = FlatGradAllReduce(list(model.parameters()), mlp=model.mlp)
    gt = torch.rand(B, 3, device=e.dev)
    native = not args.autograd and args.precision == "bf16"
    graphed = native and system.fused_adam
    if graphed:
        # the whole step (draws + forward + loss + backward [+ all-reduce] + scheduled Adam + weight re-pack) replayed from
        # captured hipGraph(s); MipLRDecay runs on the device, the scheduler object only mirrors the epoch on the host
        from mipnerf_pl_amd.train_graph import GraphedTrainStep
        gstep = GraphedTrainStep(system, opt, B, e.dev, use_graph=not args.no_graph)
        gstep.time_allreduce = gstep.collective
        for dst, src in zip(gstep.rays, R):
            dst.copy_(src)
        gstep.gt.copy_(gt)

        def step():
            sc = gstep()
            sch["scheduler"].step()
            return [(sc[:1],)]
    else:
        def step():
            opt.zero_grad(set_to_none=False)              # FlatAdam: no kernel, the next backward overwrites
            if native:
                loss = system.training_step_native((R, gt), 0)     # forward + loss + backward, one native call
            else:
                loss = system.training_step((R, gt), 0)            # randomized=True, nerf_system.py:95-121
                loss.backward()
            reduce_grads()                                # one flat all-reduce over RCCL (no-op at world 1)
            opt.step()
            sch["scheduler"].step()
            return [(loss.detach().reshape(1),)]
    step()
    preheat(step, e)
    if graphed:
        gstep.allreduce_stats()                              # drop the events of the first (capturing) step
    dt, out, dt_local = timed(e, step, args.warmup, args.steps, return_local=True)
    assert bool(torch.isfinite(out[-1][0]).all())
    samples_per_step = B * N * model.num_levels
    value = samples_per_step * e.world * args.steps / dt
    ms = dt / args.steps * 1e3
    # per-rank step time (each rank's own clock between the same two barriers) and the gradient all-reduce as the launch stream
    # sees it (HIP events around all_reduce + wait): what a first multi-GPU run needs to explain itself
    per_rank = {"ms_per_step_min": round(dt_local / args.steps * 1e3, 4), "ms_per_step_max": round(dt_local / args.steps * 1e3, 4)}
    ar_ms, ar_n = (gstep.allreduce_stats() if graphed else (None, 0))
    if e.world > 1:
        import torch.distributed as dist
        lo = torch.tensor([dt_local], device=e.dev, dtype=torch.float64)
        hi = lo.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        per_rank = {"ms_per_step_min": round(float(lo) / args.steps * 1e3, 4), "ms_per_step_max": round(float(hi) / args.steps * 1e3, 4)}
        if ar_ms is not None:
            am = torch.tensor([ar_ms], device=e.dev, dtype=torch.float64)
            dist.all_reduce(am, op=dist.ReduceOp.MAX)
            ar_ms = float(am)
    per_rank["allreduce_ms"] = None if ar_ms is None else round(ar_ms, 4)
    per_rank["allreduce_timed"] = ar_n
    peak = PEAK_TFLOPS[args.precision]
    tflops = FLOP_PER_SAMPLE_TRAIN * samples_per_step / (ms * 1e-3) / 1e12
    traffic, tsrc = pmc_traffic("train", args.precision) if (native and (B, N) == (4096, 128)) else (None, None)
    roofline = {"bound": "mfma", "kernel": "whole step (forward-with-save + dgrad + wgrad MFMA kernels and everything around them)",
                "achieved": round(tflops, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(tflops / peak, 4), "traffic": traffic,
                "traffic_source": tsrc,
                "flop_per_sample": FLOP_PER_SAMPLE_TRAIN, "samples_per_step": samples_per_step}
    if traffic:
        # what actually bounds the step: the saved activations / deltas, written once and read once
        roofline["hbm_view"] = {"bytes_per_step": traffic, "achieved_TBps": round(traffic / (ms * 1e-3) / 1e12, 3), "peak_TBps": 8.0,
                                "frac_of_peak": round(traffic / (ms * 1e-3) / 8e12, 4),
                                "note": "store-everything backprop: 3.73 TFLOP / 20.9 GB = 178 FLOP/B, below the 312 FLOP/B ridge"}
    lightning = None
    if e.world == 1 and args.precision == "bf16" and not args.autograd and not args.no_lightning_route:
        try:
            lightning = run_lightning_route(args, e, model0.state_dict(), R, gt, N)
            lightning["vs_graphed_step"] = round(lightning["ms_per_step"] / ms, 4)
        except Exception as ex:      # noqa: BLE001  (a sub-record must not take the bench line down)
            lightning = {"error": f"{type(ex).__name__}: {ex}"}
    rec = {"value": round(value, 1), "ms_per_step": round(ms, 4), "steps": args.steps, "warmup": args.warmup, "scaling": "weak",
           "roofline": roofline, "ranks": per_rank, "lightning_route": lightning,
           "config": {"workload": (f"training step (randomized forward + loss incl. distloss + backward + one flat gradient all-reduce + "
                                   f"Adam + MipLRDecay), {B} rays x ({N}+{N}) samples per GPU"),
                      "mode": "train", "preheat_seconds": PREHEAT["seconds"], "rays_per_gpu": B, "global_batch_rays": B * e.world, "samples_per_level": N, "levels": model.num_levels,
                      "native_step": native, "fused_adam": bool(system.fused_adam),
                      "hip_graph": bool(graphed and gstep.use_graph),          # what actually ran (a failed capture falls back)
                      "hip_graph_requested": bool(graphed and not args.no_graph),
                      "hip_graph_capture_error": gstep.capture_error if graphed else None,
                      "collective_path": bool(graphed and gstep.collective),
                      "lr_schedule": "device" if graphed else "host",
                      "parallelism": f"data-parallel x{e.world}, one {4 * sum(p.numel() for p in model.parameters())} B all-reduce per step"}}
    return rec


def run_lightning_route(args, e, state_dict, R, gt, N):
    """The step an UNMODIFIED train.py executes (nerf_system.py:70-121 under Lightning's automatic optimisation, train.py:48-64): the system's
    own `configure_optimizers()` (torch.optim.Adam over the 24 parameter tensors + host-side MipLRDecay), then per batch
    optimizer.zero_grad() -> training_step -> loss.backward() -> optimizer.step() -> scheduler.step(), eager, no opt-ins.  Timed twice: with
    training_step routed onto the one-call native step behind a single autograd node (the default since round 6) and with the per-stage
    autograd Functions of rounds 1-5 (`native_training_step = False`)."""
    import torch
    from mipnerf_pl_amd.system import DEFAULT_HPARAMS, MipNeRFSystem
    out = {}
    for key, native in (("ms_per_step", True), ("per_stage_autograd_ms_per_step", False)):
        hp = dict(DEFAULT_HPARAMS)
        hp.update({"nerf.num_samples": N})
        system = MipNeRFSystem(hp, precision=args.precision)
        system.mip_nerf.load_state_dict(state_dict)
        system = system.to(e.dev)
        system.native_training_step = native
        (opt,), (sch,) = system.configure_optimizers()
        assert type(opt) is torch.optim.Adam

        def step():
            opt.zero_grad()
            loss = system.training_step((R, gt), 0)
            loss.backward()
            opt.step()
            sch["scheduler"].step()
            return [(loss.detach().reshape(1),)]
        step()
        preheat(step, e)
        dt, o = timed(e, step, args.warmup, args.steps)
        assert bool(torch.isfinite(o[-1][0]).all())
        out[key] = round(dt / args.steps * 1e3, 4)
        if native:
            out["routed_onto_native_step"] = bool(system._native_step_route(R))
        del system, opt, sch
    B = R.origins.shape[0]
    out["value"] = round(B * N * 2 / (out["ms_per_step"] * 1e-3), 1)
    out["workload"] = (f"MipNeRFSystem.training_step + loss.backward() + torch.optim.Adam.step() + MipLRDecay.step(), eager, {B} rays x ({N}+{N}) samples: "
                       "what Lightning's automatic optimisation runs for an unmodified train.py")
    return out


def pmc_traffic(kind, precision, samples_per_launch=None, path=None):
    """HBM bytes from profiles/mlp_pmc.json (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, scripts/pmc_traffic.sh) for the bench
    line's roofline.traffic -> (bytes or None, source string or None).  A file written in another round is REFUSED (VERDICT r03 hygiene):
    the number must come from this round's kernels.  kind: "inference" (bytes per MLP launch) | "train" (bytes per step)."""
    path = path or os.path.join(REPO, "profiles", "mlp_pmc.json")
    if not os.path.exists(path):
        return None, None
    with open(path) as f:
        pj = json.load(f)
    if pj.get("round") != CURRENT_ROUND:
        return None, f"profiles/mlp_pmc.json is from round {pj.get('round')}, this is round {CURRENT_ROUND}: refused (re-run scripts/pmc_traffic.sh)"
    src = f"profiles/mlp_pmc.json (rocprofv3 --pmc passes of round {CURRENT_ROUND}, not this run)"
    if kind == "train":
        return (pj.get("train_hbm_bytes_per_step"), src) if precision == "bf16" else (None, None)
    if pj.get("samples_per_launch") != samples_per_launch:
        return None, None
    if precision == "bf16":
        return pj.get("hbm_bytes_per_launch"), src
    if "k_mlp_f32r" in pj.get("kernels", {}):
        return pj["kernels"]["k_mlp_f32r"]["hbm_bytes_per_launch"], src
    return None, None


def unbounded_bf16_traffic(samples_per_launch, path=None):
    """HBM bytes per launch of the three kernels of the unbounded model's bf16 forward, from this round's separate --pmc passes
    (profiles/mlp_pmc.json: "unbounded_bf16", scripts/pmc_unbounded_bf16.sh); None when the file is another round's or another size's"""
    path = path or os.path.join(REPO, "profiles", "mlp_pmc.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        pj = json.load(f)
    u = pj.get("unbounded_bf16")
    if pj.get("round") != CURRENT_ROUND or not u or u.get("samples_per_launch") != samples_per_launch:
        return None
    out = {k: v["hbm_bytes_per_launch"] for k, v in u["kernels"].items()}
    out["source"] = f"profiles/mlp_pmc.json (rocprofv3 --pmc passes of round {CURRENT_ROUND}, not this run)"
    return out


def measure_one_rank_rccl_allreduce(numel, dev):
    """N = 1 only: what ONE gradient all-reduce costs on the launch stream before any byte crosses a link -- a 1-rank RCCL communicator
    on this GPU (HIP events around all_reduce + wait, like GraphedTrainStep's collective form).  Runs in a CHILD process with a hard
    timeout: a communicator that cannot be created (or hangs) on some box must not take the bench line with it.  Returns the median ms,
    or {"error": ...}."""
    import subprocess
    code = r"""
import json, os, socket, sys
import torch, torch.distributed as dist
with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port); os.environ["NCCL_DEBUG"] = "WARN"
torch.cuda.set_device(int(sys.argv[2]))
dist.init_process_group("nccl", rank=0, world_size=1)
buf = torch.zeros(int(sys.argv[1]), device="cuda")
dist.all_reduce(buf); torch.cuda.synchronize()
ms = []
for _ in range(40):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); w = dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True); w.wait(); b.record(); b.synchronize()
    ms.append(a.elapsed_time(b))
dist.destroy_process_group()
ms.sort()
open(sys.argv[3], "w").write(json.dumps({"ms": ms[len(ms) // 2]}))      # (stdout belongs to RCCL's banner / warnings)
"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    import tempfile
    res = os.path.join(tempfile.gettempdir(), f"mipnerf_rccl1_{os.getpid()}.json")
    try:
        out = subprocess.run([sys.executable, "-c", code, str(int(numel)), str(dev.index or 0), res], capture_output=True, text=True, timeout=120, env=env)
        if os.path.exists(res):
            with open(res) as f:
                return float(json.load(f)["ms"])
        return {"error": f"rc {out.returncode}: " + (out.stderr or out.stdout)[-300:]}
    except Exception as ex:  # noqa: BLE001
        return {"error": f"{type(ex).__name__}: {ex}"}
    finally:
        if os.path.exists(res):
            os.remove(res)


def scale_model(line, ar1_ms, grad_bytes):
    """A STATED expectation for the driver's 1 / 2 / 4 / 8-GPU curve (no multi-GPU box was ever available to the builder: nothing here is a
    measurement beyond the 1-GPU numbers it starts from).  Inference and training are weak scaling (fixed per-GPU batch), the frame is
    strong scaling.  Training adds ONE 2.45-MB SUM all-reduce between backward and Adam that nothing overlaps:
        t_ar(N) = t_launch + 2 (N - 1) / N x bytes / link_bw + 2 (N - 1) x t_hop          (ring over point-to-point xGMI links)
    with link_bw = 153 GB/s per link (guide: 7 links x ~153 GB/s per GPU), t_hop = 3 us per ring step (ASSUMED, unmeasured), t_launch =
    the 1-rank RCCL all-reduce measured in this process (or 0.03 ms assumed)."""
    link_bw, t_hop = 153e9, 3e-6
    t_launch = ar1_ms * 1e-3 if isinstance(ar1_ms, float) else 30e-6
    out = {"measured": False,       # a prediction from 1-GPU numbers + stated assumptions; the driver's SCALE run is the measurement
           "measured_inputs": {"train_ms_per_step_1gpu": line["train"]["ms_per_step"], "render_ms_per_frame_1gpu": line["render"]["ms_per_step"],
                               "inference_ms_per_step_1gpu": line["ms_per_step"],
                               "allreduce_1rank_rccl_ms": ar1_ms if isinstance(ar1_ms, (float, type(None))) else None,
                               "allreduce_1rank_error": ar1_ms.get("error") if isinstance(ar1_ms, dict) else None},
           "assumptions": {"xgmi_link_GBps": 153, "ring_hop_latency_us": 3.0, "gradient_bytes": grad_bytes,
                           "allreduce_overlap": "none (between graph A and graph B)", "render_gather_bytes_per_ray": 12,
                           "per_gpu_clock_spread": "+-4 % between boxes observed in rounds 2-4: the slowest rank sets the step"},
           "predicted": {}}
    t1, f1 = line["train"]["ms_per_step"], line["render"]["ms_per_step"]
    rays = 640000
    for n in (1, 2, 4, 8):
        t_ar = 0.0 if n == 1 else (t_launch + 2 * (n - 1) / n * grad_bytes / link_bw + 2 * (n - 1) * t_hop) * 1e3
        gather = 0.0 if n == 1 else (t_launch + (n - 1) / n * rays * 12 / link_bw + (n - 1) * t_hop) * 1e3
        out["predicted"][str(n)] = {
            "train_ms_per_step": round(t1 + t_ar, 4), "train_allreduce_ms": round(t_ar, 4), "train_weak_efficiency": round(t1 / (t1 + t_ar), 4),
            "render_ms_per_frame": round(f1 / n + gather, 3), "render_strong_efficiency": round(f1 / (n * (f1 / n + gather)), 4),
            "inference_weak_efficiency": 1.0}
    return out


def run_render(args, e):
    import numpy as np
    import torch
    import synthetic_inputs as syn
    from mipnerf_pl_amd import Rays
    from mipnerf_pl_amd.parallel import gather_rendered, shard_bounds
    from mipnerf_pl_amd.system import DEFAULT_HPARAMS, MipNeRFSystem
    N = args.samples
    Himg = Wimg = 800
    lo, hi = shard_bounds(Himg * Wimg, e.rank, e.world)          # contiguous ray shard of this rank (strong scaling)
    nloc = hi - lo
    frame_np = syn.synthetic_rays(8192, seed=7)               # tiled: 640k distinct draws would dominate start-up
    reps = (nloc + 8191) // 8192
    FR = Rays(*[torch.from_numpy(np.tile(a, (reps, 1))[:nloc]).to(e.dev) for a in frame_np])
    model0, _ = make_model(args, e, N)
    hp = dict(DEFAULT_HPARAMS)
    hp.update({"nerf.num_samples": N, "val.chunk_size": int(args.render_chunk)})
    system = MipNeRFSystem(hp, precision=args.precision)
    system.mip_nerf.load_state_dict(model0.state_dict())
    system = system.to(e.dev)
    system.enable_hip_graph(not args.no_graph)
    img_rays = Rays(*[x.reshape(1, 1, nloc, -1) for x in FR])
    dummy = torch.zeros(1, 1, nloc, 3, device=e.dev)

    def step():
        _, fine, _ = system.render_image((img_rays, dummy))
        full = gather_rendered(fine.reshape(nloc, 3), Himg * Wimg)
        return [(full,)]
    frames = args.steps if args.mode == "render" else max(2, args.steps // 10)
    warm = args.warmup if args.mode == "render" else 1
    step()
    dt, out = timed(e, step, warm, frames)           # (a frame is 140 ms of back-to-back launches: no pre-heat needed)
    assert bool(torch.isfinite(out[-1][0]).all())
    samples_per_frame = Himg * Wimg * N * system.mip_nerf.num_levels
    ms = dt / frames * 1e3
    peak = PEAK_TFLOPS[args.precision]
    tflops = FLOP_PER_SAMPLE * samples_per_frame / (ms * 1e-3) / 1e12 / e.world
    rec = {"value": round(samples_per_frame * frames / dt, 1), "ms_per_step": round(ms, 4), "steps": frames, "warmup": warm,
           "scaling": "strong",
           "roofline": {"bound": "mfma", "kernel": "whole frame (k_mlp_bf16 >= 95 % of it)", "achieved": round(tflops, 2), "peak": peak,
                        "unit": "TFLOP/s per GPU", "frac": round(tflops / peak, 4), "traffic": None},
           "config": {"workload": (f"BASELINE.json configs[4]: one 800x800 frame = 640,000 rays x ({N}+{N}) samples in {int(args.render_chunk)}-ray chunks, "
                                   f"{'eager chunk loop' if args.no_graph else 'chunk forward replayed from a captured hipGraph'}, rays split "
                                   f"over {e.world} rank(s), rgb all-gathered"),
                      "mode": "render", "samples_per_level": N, "frame_rays": Himg * Wimg,
                      "rays_per_gpu_max": shard_bounds(Himg * Wimg, 0, e.world)[1], "parallelism": f"ray-split x{e.world}, all_gather of 12 B/ray"}}
    if e.world == 1 and N == 128:
        try:
            rec["reference_frame"] = render_reference_frame(args, e, frames)
        except Exception as ex:      # noqa: BLE001  (a sub-record must not take the bench line down)
            rec["reference_frame"] = {"error": f"{type(ex).__name__}: {ex}"}
    return rec


def render_reference_frame(args, e, frames):
    """configs[4] with a LIVE parity figure: the pose of tests/golden/frame_c5_800x800.npz -- one whole 800 x 800 `RenderGen`
    frame rendered by the unmodified reference on the trained field (scripts/make_golden.py --only-frame) -- through the product route:
    datasets.RenderGen (rays generated on the device) -> MipNeRFSystem.render_image with the frame's captured hipGraph; timed, then compared
    with the reference's pixels."""
    import numpy as np
    import torch
    from mipnerf_pl_amd import Rays
    from mipnerf_pl_amd.datasets import RenderGen
    from mipnerf_pl_amd.system import DEFAULT_HPARAMS, MipNeRFSystem
    gdir = os.path.join(REPO, "tests", "golden")
    g = np.load(os.path.join(gdir, "frame_c5_800x800.npz"))
    f = np.load(os.path.join(gdir, str(g["field"]) + ".npz"))
    size = int(g["cfg_size"])
    hp = dict(DEFAULT_HPARAMS)
    hp.update({"nerf.num_samples": int(g["cfg_num_samples"]), "val.chunk_size": int(g["cfg_chunk"])})
    system = MipNeRFSystem(hp, precision=args.precision)
    system.load_state_dict({"mip_nerf.mlp." + k[2:]: torch.from_numpy(f[k].copy()) for k in f.files if k.startswith("p_")})
    system = system.to(e.dev).eval()
    system.enable_hip_graph(not args.no_graph)
    ds = RenderGen(float(g["focal"]), [size, size], scales=1, device=e.dev)
    rgbs = torch.empty(1, size, size, 3, device=e.dev)

    def step():
        rays = ds[int(g["cfg_pose"])]                          # rays of the pose from the camera table, every frame (k_generate_rays)
        return system.render_image((Rays(*[x[None] for x in rays]), rgbs))
    step()
    dt, out = timed(e, step, 1, frames)
    fine = out[1][0].cpu().numpy().astype(np.float64)
    gt = g["gt_u8"].astype(np.float64) / 255.0
    psnr = lambda a, b: float(-10.0 * np.log10(np.mean((a - b) ** 2) + 1e-30))
    return {"ms_per_step": round(dt / frames * 1e3, 3), "steps": frames, "hip_graph": bool(getattr(system._graphed, "graph", None)) if not args.no_graph else False,
            "psnr_vs_reference_frame_db": round(psnr(fine, g["fine_rgb"][0].astype(np.float64)), 2),
            "max_abs_fine_rgb": float(np.abs(fine - g["fine_rgb"][0]).max()),
            "psnr_vs_scene_db": round(psnr(fine, gt), 3), "reference_psnr_vs_scene_db": round(float(g["psnr_fine"]), 3),
            "workload": (f"pose {int(g['cfg_pose'])} of the reference's spheric render path at {size} x {size}, trained field, rays generated on the device, "
                         f"{int(g['cfg_chunk'])}-ray chunks (79, the last one 1024 rays); golden = the unmodified reference's frame, "
                         "tests/golden/frame_c5_800x800.npz")}


def run_ceiling(args, e):
    """What an MFMA stream sustains on THIS chip, in THIS process (VERDICT r02 #2a): back-to-back v_mfma_f32_32x32x16_bf16 on
    MLP-like random operands, 2 waves per SIMD, with the operands in registers and with one LDS weight-fragment read per MFMA
    (what k_mlp_bf16 does); each variant runs >= 1 s (first half heat-up).  The datasheet's 2.5 PFLOP/s assumes 2.4 GHz."""
    import ctypes as C
    import torch
    from mipnerf_pl_amd import _lib as L
    out = {"unit": "TFLOP/s", "operands": "weights U(-0.1,0.1), activations relu(N(0,1)), bf16", "waves_per_simd": 2,
           "seconds_per_variant": args.ceiling_seconds, "peak": PEAK_TFLOPS["bf16"], "variants": []}
    st = torch.cuda.current_stream().cuda_stream
    D = L.diag_lib()          # measurement tooling lives in its own library (include/mipnerf_diag.h), not in the drop-in .so
    if D is None:
        return {"absent": "libmipnerf_diag.so was not built (python -m mipnerf_pl_amd.build)"}
    for lds in (0, 1, 2, 3):
        r = (C.c_double * 3)()
        L.diag_check(D.mipnerf_mfma_ceiling(lds, 2, 1, float(args.ceiling_seconds), r, st), "mfma_ceiling")
        out["variants"].append({"lds_weight_reads_per_mfma": min(lds, 1), "weight_dma_l2_to_lds": lds >= 2,
                                "saved_activation_stores": "4608 B/sample non-temporal (k_mlp_bf16_trainfwd's stream)" if lds == 3 else None,
                                "tflops": round(r[0], 1),
                                "frac_of_peak": round(r[0] / PEAK_TFLOPS["bf16"], 4),
                                "ms_per_launch": round(r[1], 4), "effective_clock_ghz": round(r[2], 3)})
    out["register_fed"] = out["variants"][0]["frac_of_peak"]
    out["lds_fed"] = out["variants"][1]["frac_of_peak"]
    out["lds_and_dma_fed"] = out["variants"][2]["frac_of_peak"]     # + the kernel's L2 -> LDS weight stream (global_load_lds) at its rate
    # + the training forward's saved-activation stream (one 1-KiB non-temporal store per wave per 9.4 MFMAs, 3.7 GB per launch): what the
    # matrix pipe sustains while the producers of the training step write their T-blocks; the stores run at store_TBps
    out["lds_dma_and_store_fed"] = out["variants"][3]["frac_of_peak"]
    out["store_TBps_in_that_variant"] = round(256 * 8 * 256 * 7 * 1024 / (out["variants"][3]["ms_per_launch"] * 1e-3) / 1e12, 3)
    # the fp32 matrix instruction of the parity mode / configs[3] (v_mfma_f32_32x32x2_f32), register-fed
    r = (C.c_double * 3)()
    L.diag_check(D.mipnerf_mfma_ceiling(10, 2, 1, float(args.ceiling_seconds), r, st), "mfma_ceiling fp32")
    out["fp32_register_fed"] = {"tflops": round(r[0], 1), "frac_of_peak": round(r[0] / PEAK_TFLOPS["fp32"], 4), "effective_clock_ghz": round(r[2], 3)}
    return out


def run_fp32_c4(args, e):
    """BASELINE.json configs[3]-shaped forward in fp32 parity mode: 8192 rays x (256 + 256) samples, per-ray near / far, exact-fp32
    MFMA (v_mfma_f32_32x32x2_f32) -- so the driver's record carries the fp32 kernel's roofline too (VERDICT r02 #2b)."""
    import torch
    import synthetic_inputs as syn
    from mipnerf_pl_amd import MipNerf, Rays
    B, N = 8192, 256
    rays_np = syn.synthetic_rays(B, seed=100 + e.rank, unbounded=True)
    params = syn.make_params(seed=0, density_gain=40.0)
    model = MipNerf(num_samples=N, precision="fp32")
    model.load_state_dict({"mlp." + k: torch.from_numpy(v.copy()) for k, v in params.items()})
    model = model.to(e.dev)
    R = Rays(*[torch.from_numpy(a).to(e.dev) for a in rays_np])

    def step():
        with torch.no_grad():
            return model(R, False, True)
    step()
    ctx = model.mlp.native(e.dev)
    steps, warm = max(3, args.steps // 10), 1            # (42 ms per step: no pre-heat needed)
    for _ in range(warm):
        step()
    ctx.set_option(2, 1)
    dt, out = timed(e, step, 0, steps)
    tot_ms, nl = launch_stats(ctx)
    ctx.set_option(2, 0)
    assert bool(torch.isfinite(out[-1][0]).all())
    M = B * N
    launch_ms = tot_ms / max(nl, 1)
    tflops = FLOP_PER_SAMPLE * M / (launch_ms * 1e-3) / 1e12
    peak = PEAK_TFLOPS["fp32"]
    # the same batch through the unbounded-scene model (configs[3] says "360 unbounded"): inverse-depth sampling, contracted
    # full-covariance Gaussians, 672-wide first layer / skip concat -> 2 x 576 x 256 more MACs per sample
    unb = None
    try:
        flop_u = FLOP_PER_SAMPLE + 2 * 2 * (672 - 96) * 256
        um = MipNerf(num_samples=N, precision="fp32", unbounded=True)
        um.load_state_dict({"mlp." + k: torch.from_numpy(v.copy()) for k, v in syn.make_params(seed=0, density_gain=40.0, xyz_dim=672).items()})
        um = um.to(e.dev)

        def ustep():
            with torch.no_grad():
                return um(R, False, True)
        ustep()
        uctx = um.mlp.native(e.dev)
        ustep()
        uctx.set_option(2, 1)
        udt, uout = timed(e, ustep, 0, steps)
        utot, unl = launch_stats(uctx)
        uctx.set_option(2, 0)
        ulaunch = utot / max(unl, 1)
        utf = flop_u * M / (ulaunch * 1e-3) / 1e12
        unb = {"ms_per_step": round(udt / steps * 1e3, 4), "value": round(B * N * 2 * e.world * steps / udt, 1), "launch_ms": round(ulaunch, 4),
               "flop_per_sample": flop_u, "achieved": round(utf, 2), "frac": round(utf / peak, 4), "finite": bool(torch.isfinite(uout[-1][0]).all()),
               "model": "MipNerf(unbounded=True): 672-wide encoding, fp32 precision"}
        # ... and in bf16 (round 4: k_pre_gemm + trunk kernel, csrc/gen_pre_gemm.py): same rays, same weights, agreement with the fp32 frame
        try:
            bm = MipNerf(num_samples=N, precision="bf16", unbounded=True)
            bm.load_state_dict(um.state_dict())
            bm = bm.to(e.dev)

            def bstep():
                with torch.no_grad():
                    return bm(R, False, True)
            bstep()
            bctx = bm.mlp.native(e.dev)
            bstep()
            bctx.set_option(2, 1)
            bsteps = max(5, args.steps // 4)
            bdt, bout = timed(e, bstep, 0, bsteps)
            btot, bnl = launch_stats(bctx)
            bctx.set_option(2, 0)
            blaunch = btot / max(bnl, 1)
            btf = flop_u * M / (blaunch * 1e-3) / 1e12
            mse = float(torch.mean((bout[-1][0] - uout[-1][0]) ** 2))
            unb["bf16"] = {"ms_per_step": round(bdt / bsteps * 1e3, 4), "value": round(B * N * 2 * e.world * bsteps / bdt, 1), "steps": bsteps,
                           "mlp_ms_per_level": round(blaunch, 4), "achieved": round(btf, 2), "peak": PEAK_TFLOPS["bf16"],
                           "frac": round(btf / PEAK_TFLOPS["bf16"], 4), "kernels": "the one-kernel form (round 6): layer 0 and the skip layer as k-step-major ops of the MLP kernel, timed per level",
                           "psnr_vs_fp32_frame_db": round(float(-10 * math.log10(max(mse, 1e-20))), 2),
                           "finite": bool(torch.isfinite(bout[-1][0]).all()), "traffic": unbounded_bf16_traffic(M)}
        except Exception as ex:  # noqa: BLE001
            unb["bf16"] = {"error": f"{type(ex).__name__}: {ex}"}
        # ... and TRAINING this model (round 5): one optimisation step through autograd (training_step + backward + torch Adam, what the
        # reference's Lightning loop does) on 4096 rays x (128 + 128) samples, bf16 kernels against the fp32 path
        try:
            unb["train"] = run_train_unbounded(args, e)
        except Exception as ex:  # noqa: BLE001
            import traceback
            traceback.print_exc()          # (with world > 1 the other ranks are inside this sub-record's collectives: say why this one left)
            unb["train"] = {"error": f"{type(ex).__name__}: {ex}"}
    except Exception as ex:  # noqa: BLE001  (an extra must not take the record down)
        unb = {"error": f"{type(ex).__name__}: {ex}"}
    return {"unbounded": unb, "value": round(B * N * 2 * e.world * steps / dt, 1), "ms_per_step": round(dt / steps * 1e3, 4), "steps": steps, "warmup": warm,
            "scaling": "weak", "dtype": "fp32",
            "roofline": {"bound": "mfma", "kernel": "k_mlp_f32r (register-resident, generated by csrc/gen_mlp_f32r.py; round 3: the LDS-resident k_mlp_f32)",
                         "achieved": round(tflops, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(tflops / peak, 4), "traffic": None, "launch_ms": round(launch_ms, 4), "launches_timed": nl,
                         "samples_per_launch": M, "flop_per_sample": FLOP_PER_SAMPLE},
            "config": {"workload": (f"BASELINE.json configs[3] shape: MipNerf.forward inference, {B} rays x ({N} coarse + {N} fine) samples per GPU, "
                                    f"per-ray near/far, fp32 parity mode"), "mode": "fp32", "rays_per_gpu": B, "samples_per_level": N,
                       "parallelism": f"ray-split x{e.world} (no data-path collective)"}}


def run_variants(args, e):
    """The other generated MLP shapes that got bf16 kernels in round 5 (two view layers; the 512-wide trunk at one wave per SIMD): MLP.forward =
    one kernel launch, 524,288 samples, bf16 encodings resident, against the same shape's fp32 kernel; FLOP = 2 x the torch weights' MACs."""
    import torch
    from mipnerf_pl_amd import MipNerf
    B, N = 4096, 128
    out = {"workload": f"MLP.forward (one kernel) of {B * N} samples per architecture variant, random-init weights, bf16 vs fp32"}
    torch.manual_seed(0)
    v = torch.rand(B, 27, device=e.dev) * 2 - 1
    for name, kw in (("two_view_layers", dict(mlp_net_depth_condition=2)), ("w512_c256", dict(mlp_net_width=512, mlp_net_width_condition=256))):
        rec = {}
        for prec, dt_ in (("bf16", torch.bfloat16), ("fp32", torch.float32)):
            m = MipNerf(num_samples=N, precision=prec, **kw).to(e.dev)
            macs = sum(p.numel() for n_, p in m.mlp.named_parameters() if n_.endswith("weight"))
            x = (torch.rand(B, N, 96, device=e.dev) * 2 - 1).to(dt_)
            with torch.no_grad():
                m.mlp(x, v)
                steps = 5
                dt, _ = timed(e, lambda: m.mlp(x, v), 1, steps)
            ms = dt / steps * 1e3
            tf = 2 * macs * B * N / (ms * 1e-3) / 1e12
            rec[prec] = {"ms_per_launch": round(ms, 4), "achieved_tflops": round(tf, 1), "frac_of_peak": round(tf / PEAK_TFLOPS[prec], 4)}
            del m, x
        rec["speedup_bf16_over_fp32"] = round(rec["fp32"]["ms_per_launch"] / rec["bf16"]["ms_per_launch"], 2)
        out[name] = rec
    # (the record's `value`: samples per second through the 512-wide trunk's bf16 kernel on this rank x ranks)
    out["value"] = round(B * N * e.world / (out["w512_c256"]["bf16"]["ms_per_launch"] * 1e-3), 1)
    torch.cuda.empty_cache()
    return out


def run_train_unbounded(args, e, which=("fp32", "bf16", "bf16_graph")):
    import torch
    import synthetic_inputs as syn
    from mipnerf_pl_amd import Rays
    from mipnerf_pl_amd.system import DEFAULT_HPARAMS, MipNeRFSystem
    B, N = 4096, 128
    R = Rays(*[torch.from_numpy(a).to(e.dev) for a in syn.synthetic_rays(B, seed=100 + e.rank, unbounded=True)])
    gt = torch.rand(B, 3, device=e.dev)
    params = syn.make_params(seed=0, density_gain=40.0, xyz_dim=672)
    out = {"workload": f"MipNerf(unbounded=True) training step through autograd, {B} rays x ({N} + {N}) samples, randomized, torch Adam",
           "flop_per_sample": 3 * 1810432 - 2 * 2 * 672 * 256}          # forward + wgrad + dgrad (no dgrad into the encoding: 2 x 672 x 256 MACs less)
    for precision, steps in (("fp32", 3), ("bf16", 10)):
        if precision not in which:
            continue
        hp = dict(DEFAULT_HPARAMS)
        hp.update({"nerf.num_samples": N, "nerf.unbounded": True})
        system = MipNeRFSystem(hp, precision=precision)
        system.load_state_dict({"mip_nerf.mlp." + k: torch.from_numpy(v.copy()) for k, v in params.items()})
        system = system.to(e.dev)
        (opt,), (sch,) = system.configure_optimizers()

        def step():
            opt.zero_grad(set_to_none=True)
            loss = system.training_step((R, gt), 0)
            loss.backward()
            opt.step()
            sch["scheduler"].step()
            return loss.detach()          # (not the graph: its node keeps the model, and with it the step's 12-GiB workspace, alive past `del system`)
        step()
        step()
        dt, loss = timed(e, step, 0, steps)
        ms = dt / steps * 1e3
        tf = out["flop_per_sample"] * B * N * 2 / (ms * 1e-3) / 1e12
        out[precision] = {"ms_per_step": round(ms, 3), "steps": steps, "value": round(B * N * 2 * e.world / (ms * 1e-3), 1), "achieved_tflops": round(tf, 1),
                          "frac_of_peak": round(tf / PEAK_TFLOPS[precision], 4), "loss_finite": bool(torch.isfinite(loss))}
        del system, opt, loss
        torch.cuda.empty_cache()
    if "fp32" in out and "bf16" in out:
        out["speedup_bf16_over_fp32"] = round(out["fp32"]["ms_per_step"] / out["bf16"]["ms_per_step"], 2)
    if "bf16_graph" not in which:
        return out
    # round 5: the same step as ONE captured hipGraph (mipnerf_train_step's unbounded branch + device-side Adam / MipLRDecay + weight re-pack)
    from mipnerf_pl_amd.train_graph import GraphedTrainStep
    hp = dict(DEFAULT_HPARAMS)
    hp.update({"nerf.num_samples": N, "nerf.unbounded": True})
    system = MipNeRFSystem(hp, precision="bf16")
    system.load_state_dict({"mip_nerf.mlp." + k: torch.from_numpy(v.copy()) for k, v in params.items()})
    system = system.to(e.dev)
    system.fused_adam = True
    (opt,), (sch,) = system.configure_optimizers()
    g = GraphedTrainStep(system, opt, B, e.dev, use_graph=True)
    for dst, src in zip(g.rays, R):
        dst.copy_(src)
    g.gt.copy_(gt)
    g()
    g()
    steps = 10
    dt, sc = timed(e, g, 0, steps)
    ms = dt / steps * 1e3
    tf = out["flop_per_sample"] * B * N * 2 / (ms * 1e-3) / 1e12
    out["bf16_graph"] = {"ms_per_step": round(ms, 3), "steps": steps, "value": round(B * N * 2 * e.world / (ms * 1e-3), 1), "achieved_tflops": round(tf, 1),
                         "frac_of_peak": round(tf / PEAK_TFLOPS["bf16"], 4), "loss_finite": bool(torch.isfinite(sc[0])), "hip_graph": g._graphs is not None,
                         "hip_graph_capture_error": g.capture_error,
                         "workload": "the same step through mipnerf_train_step (one call) + device-side Adam, one captured hipGraph"}
    if "fp32" in out:
        out["speedup_bf16_graph_over_fp32"] = round(out["fp32"]["ms_per_step"] / ms, 2)
    del system, opt, g
    torch.cuda.empty_cache()
    return out


def cpu_baseline(args, rays_np, params):
    """The reference's own CPU path on this host (rank 0, N=1): MipNerf.forward of the staged reference (oracle/_ref) under
    no_grad, fp32, all host cores, on the FULL headline batch; falls back to the numpy port on a 256-ray sample."""
    import numpy as np
    import torch
    B, N = args.rays, args.samples
    cpu = "?"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                cpu = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    from oracle import ref as oref
    r = oref.load()
    if r is not None:
        model = r.MipNerf(num_samples=N)
        model.load_state_dict({"mlp." + k: torch.from_numpy(v.copy()) for k, v in params.items()}, strict=True)
        model.eval()
        RR = r.Rays(*[torch.from_numpy(np.asarray(a)) for a in rays_np])
        ts = []
        with torch.no_grad():
            # torch's CPU ops stop scaling long before 256 hardware threads (measured on the 2x64-core EPYC host: 32 threads
            # 2.2e5 ray-samples/s, 256 threads 1.2e4): give the reference the thread count that is FASTEST for it here
            # the count is chosen on the FULL batch (a 512-ray probe picked 8 threads on one box where 16-32 are 20 % faster on
            # 4096 rays): one forward per candidate, ascending, stop at the first that is slower
            torch.set_num_threads(min(16, os.cpu_count()))
            model(RR, False, True)                                   # warm-up on the full batch (pages, allocator)
            best, best_t, probe = None, None, {}
            for th in sorted({t for t in (8, 16, 32, 64, os.cpu_count()) if t <= os.cpu_count()}):
                torch.set_num_threads(th)
                c0 = time.perf_counter()
                model(RR, False, True)
                dt_ = time.perf_counter() - c0
                probe[th] = round(dt_, 3)
                if best_t is None or dt_ < best_t:
                    best, best_t = th, dt_
                elif dt_ > 1.05 * best_t:
                    break
            torch.set_num_threads(best)
            ts.append(best_t)
            for _ in range(2):
                c0 = time.perf_counter()
                model(RR, False, True)
                ts.append(time.perf_counter() - c0)
        med = sorted(ts)[1]
        return {"value": round(B * N * 2 / med, 1), "unit": "ray-samples/s", "cores": torch.get_num_threads(), "host_cores": os.cpu_count(),
                "kind": "reference", "cpu": cpu, "seconds_per_forward": round(med, 3),
                "sample": f"median of 3 x the reference's MipNerf.forward (oracle/_ref, torch {torch.__version__} CPU, fp32, no_grad, "
                          f"{torch.get_num_threads()} threads = the fastest on the full batch, seconds per forward by thread count: {probe}) on the full "
                          f"{B} rays x {N} samples x 2 levels batch"}
    from oracle import mipnerf_oracle as orc
    nb = 256
    sub = orc.Rays(*[a[:nb] for a in rays_np])
    orc.mipnerf_forward(params, orc.Rays(*[a[:32] for a in rays_np]), False, True, num_samples=N)  # warm BLAS
    reps, tcpu = 0, 0.0
    while tcpu < 10.0 and reps < 20:
        c0 = time.perf_counter()
        orc.mipnerf_forward(params, sub, False, True, num_samples=N)
        tcpu += time.perf_counter() - c0
        reps += 1
    return {"value": round(nb * N * 2 * reps / tcpu, 1), "unit": "ray-samples/s", "cores": os.cpu_count(), "kind": "port", "cpu": cpu,
            "sample": f"oracle/_ref not staged: {reps} x oracle.mipnerf_forward on {nb} rays x {N} samples x 2 levels (numpy fp32, "
                      f"BLAS threads = all cores)"}


def cpu_baseline_train(args, rays_np, params, threads):
    """The reference's own training step on this host for the train sub-record ("also time fwd+bwd with the
    nerf_system loss"): staged reference MipNerf.forward(randomized=True) + the loss of nerf_system.py:99-111 (restated: the
    module needs Lightning) + backward + torch.optim.Adam, one step on the full batch (about 20 s of CPU at 4096 x 128)."""
    import numpy as np
    import torch
    from oracle import ref as oref
    r = oref.load()
    if r is None:
        return None
    N, nb = args.samples, int(args.rays)
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    model = r.MipNerf(num_samples=N)
    model.load_state_dict({"mlp." + k: torch.from_numpy(v.copy()) for k, v in params.items()}, strict=True)
    opt = torch.optim.Adam(model.parameters(), lr=5e-4)

    def batch(n):
        return r.Rays(*[torch.from_numpy(np.asarray(a)[:n].copy()) for a in rays_np]), torch.rand(n, 3)

    def step(RR, gt):
        ret = model(RR, True, True)
        mask = RR.lossmult
        terms = [(mask * (rgb - gt) ** 2).sum() / mask.sum() + 0.01 * r.mip.distloss(w, t) for rgb, _, _, w, t in ret]
        loss = 0.1 * terms[0] + terms[1]
        opt.zero_grad()
        loss.backward()
        opt.step()
    step(*batch(min(256, nb)))          # warm-up (allocator, thread pool) on a small slice
    full = batch(nb)
    c0 = time.perf_counter()
    step(*full)                         # ONE step on the FULL batch (no extrapolation from a 1024-ray sample)
    sec = time.perf_counter() - c0
    return {"value": round(nb * N * 2 / sec, 1), "unit": "ray-samples/s", "cores": threads, "host_cores": os.cpu_count(), "kind": "reference",
            "seconds_per_step": round(sec, 3),
            "sample": f"one training step of the reference (oracle/_ref MipNerf.forward randomized + nerf_system.py:99-111 loss + "
                      f"backward + torch Adam, fp32, {threads} threads) on the FULL batch: {nb} rays x {N} samples x 2 levels, after a "
                      f"256-ray warm-up step"}


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    # a rank stuck in a collective says WHERE: every thread's Python stack on stderr after N seconds, then exit.  On by default for
    # multi-rank runs (1800 s: a healthy --gpus 8 run takes a few minutes), so that a hang in somebody else's unattended SCALE run leaves
    # a diagnosis instead of a timeout; MIPNERF_BENCH_WATCHDOG_SECONDS=0 turns it off, any other value sets it (also for one rank)
    wd = os.environ.get("MIPNERF_BENCH_WATCHDOG_SECONDS", "1800" if int(os.environ.get("WORLD_SIZE", "1")) > 1 else "0")
    if float(wd) > 0:
        import faulthandler
        faulthandler.dump_traceback_later(float(wd), exit=True)
    e = setup(args)
    PREHEAT["seconds"] = max(0.0, args.preheat_seconds)
    import torch.distributed as dist
    recs, inputs = {}, None
    if args.mode in ("all", "inference"):
        recs["inference"], inputs = run_inference(args, e)
    def sub(name, fn):
        # a sub-record must not take the headline down with it (mode all); a single-mode run fails loudly
        if args.mode != "all":
            return fn(args, e)
        try:
            return fn(args, e)
        except Exception as ex:  # noqa: BLE001
            import traceback
            traceback.print_exc()
            return {"value": 0.0, "ms_per_step": None, "steps": 0, "warmup": 0, "scaling": None, "roofline": None,
                    "config": {"mode": name}, "error": f"{type(ex).__name__}: {ex}"}
    if args.mode in ("all", "train"):
        recs["train"] = sub("train", run_train)
    if args.mode in ("all", "render"):
        recs["render"] = sub("render", run_render)
    if args.mode == "all" and not args.no_fp32 and args.precision == "bf16":
        recs["fp32"] = sub("fp32", run_fp32_c4)
    if args.mode == "all" and args.rays == 4096 and args.samples == 128:      # the default (driver) invocation
        recs["trained_field"] = sub("trained_field", run_trained_field)
        if not args.no_fp32 and args.precision == "bf16":
            recs["variants"] = sub("variants", run_variants)
    ceiling = None
    if args.mode in ("all", "inference") and args.precision == "bf16" and e.world == 1 and args.ceiling_seconds > 0:
        try:
            ceiling = run_ceiling(args, e)
        except Exception as ex:  # noqa: BLE001  (a diagnostic must not take the line down)
            ceiling = {"error": f"{type(ex).__name__}: {ex}"}
    head_mode = "inference" if "inference" in recs else args.mode
    head = recs.pop(head_mode)
    who = rank_report(e)
    cpu = None
    if e.rank == 0 and e.world == 1 and not args.no_cpu_baseline and inputs is not None:
        cpu = cpu_baseline(args, *inputs)
        if "train" in recs and recs["train"].get("value") and cpu.get("kind") == "reference":
            try:
                recs["train"]["cpu_baseline"] = cpu_baseline_train(args, *inputs, cpu["cores"])
            except Exception as ex:  # noqa: BLE001  (a baseline must not take the line down)
                recs["train"]["cpu_baseline"] = {"error": f"{type(ex).__name__}: {ex}"}
    if e.rank == 0:
        line = {"metric": "ray-samples/sec", "value": head["value"], "unit": "ray-samples/s", "n_gpus": e.world,
                "steps": head["steps"], "warmup": head["warmup"], "ms_per_step": head["ms_per_step"], "higher_is_better": True,
                "scaling": head["scaling"], "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
                "config": head["config"], "per_gpu": round(head["value"] / e.world, 1), "roofline": head["roofline"],
                "cpu_baseline": cpu}
        if head.get("sustained") is not None:
            line["sustained"] = head["sustained"]
        tr = recs.get("train") or (head if head_mode == "train" else {})
        if tr.get("ranks"):                      # the gradient all-reduce as the launch stream timed it, next to the ranks that took part
            who["train_allreduce_ms"] = tr["ranks"].get("allreduce_ms")
            who["train_ms_per_step_min_max"] = [tr["ranks"].get("ms_per_step_min"), tr["ranks"].get("ms_per_step_max")]
        line["ranks"] = who
        if ceiling is not None:
            line["ceiling"] = ceiling
            if line.get("roofline") and "lds_fed" in ceiling and ceiling["lds_fed"] > 0:
                line["roofline"]["frac_of_measured_lds_fed_ceiling"] = round(line["roofline"]["frac"] / ceiling["lds_fed"], 4)
                if "fp32" in recs and recs["fp32"].get("roofline") and ceiling.get("fp32_register_fed", {}).get("frac_of_peak", 0) > 0:
                    recs["fp32"]["roofline"]["frac_of_measured_register_fed_ceiling"] = round(
                        recs["fp32"]["roofline"]["frac"] / ceiling["fp32_register_fed"]["frac_of_peak"], 4)
                if ceiling.get("lds_and_dma_fed", 0) > 0:
                    line["roofline"]["frac_of_measured_lds_and_dma_fed_ceiling"] = round(line["roofline"]["frac"] / ceiling["lds_and_dma_fed"], 4)
        for k, r in recs.items():
            r.update({"metric": "ray-samples/sec", "unit": "ray-samples/s", "n_gpus": e.world, "per_gpu": round(r["value"] / e.world, 1)})
            line[k] = r
        if e.world == 1 and args.mode == "all" and line.get("train", {}).get("ms_per_step") and line.get("render", {}).get("ms_per_step"):
            # what the builder EXPECTS of the 1 / 2 / 4 / 8 curve, stated before anybody measures it
            try:
                line["scale_model"] = scale_model(line, measure_one_rank_rccl_allreduce(612740, e.dev), 4 * 612740)
            except Exception as ex:  # noqa: BLE001
                line["scale_model"] = {"error": f"{type(ex).__name__}: {ex}"}
        # the ONE JSON line must be the last thing on stdout: RCCL prints its version banner through C stdio, which is flushed at exit
        # (i.e. AFTER a Python print) unless it is flushed here first
        try:
            import ctypes
            sys.stdout.flush()
            ctypes.CDLL(None).fflush(None)
        except Exception:  # noqa: BLE001
            pass
        print(json.dumps(line), flush=True)
    if e.world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

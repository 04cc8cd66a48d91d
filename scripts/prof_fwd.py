#!/usr/bin/env python3
"""The headline forward (MipNerf.forward, fused-IPE bf16 MLP kernel) a few times: target for rocprofv3 --pmc passes and for A/B
timing of kernel builds (MIPNERF_LIB=<other .so>).   usage: prof_fwd.py [--iters 20] [--rays 4096] [--samples 128]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import synthetic_inputs as syn  # noqa: E402
from mipnerf_pl_amd import MipNerf, Rays  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--rays", type=int, default=4096)
ap.add_argument("--samples", type=int, default=128)
ap.add_argument("--heat", type=float, default=0.0, help="seconds of untimed forwards first")
ap.add_argument("--graph", action="store_true", help="replay the forward from one captured hipGraph (model.GraphedForward)")
ap.add_argument("--grid", type=int, default=0, help="persistent workgroups of the MLP kernel (option 1; default = CUs)")
a = ap.parse_args()
dev = torch.device("cuda:0")
params = syn.make_params(seed=0, density_gain=40.0)
m = MipNerf(num_samples=a.samples, precision="bf16")
m.load_state_dict({"mlp." + k: torch.from_numpy(v.copy()) for k, v in params.items()})
m = m.to(dev)
R = Rays(*[torch.from_numpy(x).to(dev) for x in syn.synthetic_rays(a.rays, seed=100)])
if a.grid:
    m.mlp.native(dev).set_option(1, a.grid)
with torch.no_grad():
    m(R, False, True)
    t_end = time.perf_counter() + a.heat
    while time.perf_counter() < t_end:
        for _ in range(8):
            m(R, False, True)
        torch.cuda.synchronize()
    ctx = m.mlp.native(dev)
    ctx.set_option(2, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        out = m(R, False, True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.iters
    if a.graph:          # the same steps as graph replays (the MLP launch time above comes from the eager pass: events do not live in graphs)
        from mipnerf_pl_amd.model import GraphedForward
        ctx.set_option(2, 0)
        gf = GraphedForward(m, a.rays, True)
        gf(R)
        for _ in range(50):
            gf.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            out = gf.replay()
        torch.cuda.synchronize()
        dtg = (time.perf_counter() - t0) / a.iters
        print(f"graph replay: {dtg * 1e3:.4f} ms/step (eager {dt * 1e3:.4f}); captured: {bool(gf.graph)}")
import ctypes as C  # noqa: E402
from mipnerf_pl_amd import _lib as L  # noqa: E402
tot, nl = C.c_double(), C.c_int64()
L.check(L.lib().mipnerf_mlp_launch_stats(ctx.handle, C.byref(tot), C.byref(nl)))
lm = tot.value / max(nl.value, 1)
print(f"forward {a.rays}x{a.samples}: {dt * 1e3:.4f} ms/step, MLP launch {lm:.4f} ms = {1220608 * a.rays * a.samples / (lm * 1e-3) / 1e12:.1f} TFLOP/s "
      f"({os.path.basename(os.environ.get('MIPNERF_LIB', 'libmipnerf_hip.so'))}, grid {a.grid or 'CUs'}) checksum {float(out[1][0].double().sum()):.9f}")

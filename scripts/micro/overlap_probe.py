"""round 6 probe: can the weight-gradient kernel of one level (HBM-read-bound) run BESIDE the dgrad kernel of the other level (MFMA + store-bound)?
Timing only (garbage operands): 524,288 samples per level as in the training step of BASELINE configs[1].
  A  dgrad then wgrad on one stream (what mipnerf_train_step does per level)
  B  dgrad on stream 1 with `grid` persistent workgroups beside wgrad on stream 2
usage: overlap_probe.py [iterations]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import synthetic_inputs as syn  # noqa: E402
from mipnerf_pl_amd import MipNerf, _lib as L  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
m = MipNerf(num_samples=128, precision="bf16")
m.load_state_dict({"mlp." + k: torch.from_numpy(v.copy()) for k, v in syn.make_params(seed=0, density_gain=40.0).items()})
m = m.to(dev)
ctx = m.mlp.native(dev)
M = 4096 * 128
sz = ctx.train_sizes(M)
gen = torch.Generator(device=dev).manual_seed(0)
act = torch.randint(0, 255, (sz[0],), dtype=torch.uint8, device=dev, generator=gen)
masks = torch.randint(0, 255, (sz[1],), dtype=torch.uint8, device=dev, generator=gen)
delta = [torch.randint(0, 255, (sz[2],), dtype=torch.uint8, device=dev, generator=gen) for _ in range(2)]
partials = [torch.empty(sz[3], dtype=torch.uint8, device=dev) for _ in range(2)]
d_raw = torch.randn(M, 4, device=dev, generator=gen) * 1e-3
lib = L.lib()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def dgrad(st, which):
    L.check(lib.mipnerf_mlp_dgrad(ctx.handle, M, d_raw.data_ptr(), masks.data_ptr(), delta[which].data_ptr(), st.cuda_stream), "dgrad")


def wgrad(st, which):
    L.check(lib.mipnerf_mlp_wgrad(ctx.handle, M, act.data_ptr(), delta[which].data_ptr(), partials[which].data_ptr(), None, 0, st.cuda_stream), "wgrad")


def timed(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


cur = torch.cuda.current_stream()
wgrad(cur, 1)        # delta[1] valid-ish
for rnd in range(2):
    ctx.set_option(1, 256)
    print(f"dgrad alone {timed(lambda: dgrad(cur, 0)):.3f} ms, wgrad alone {timed(lambda: wgrad(cur, 1)):.3f} ms, "
          f"A sequential {timed(lambda: (dgrad(cur, 0), wgrad(cur, 1))):.3f} ms", flush=True)
    for grid in (256, 192, 128, 64):
        def both():
            s1.wait_stream(cur)
            s2.wait_stream(cur)
            ctx.set_option(1, grid)
            dgrad(s1, 0)
            wgrad(s2, 1)
            cur.wait_stream(s1)
            cur.wait_stream(s2)
        print(f"  B dgrad (grid {grid}) beside wgrad: {timed(both):.3f} ms", flush=True)
        def both2():
            s1.wait_stream(cur)
            s2.wait_stream(cur)
            ctx.set_option(1, grid)
            wgrad(s2, 1)
            dgrad(s1, 0)
            cur.wait_stream(s1)
            cur.wait_stream(s2)
        print(f"  B' wgrad first, dgrad (grid {grid}) beside: {timed(both2):.3f} ms", flush=True)
ctx.set_option(1, 256)

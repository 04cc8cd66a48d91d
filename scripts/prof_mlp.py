#!/usr/bin/env python3
"""Run only the MLP kernel (BASELINE configs[1] size) a few times: target for rocprofv3 --pmc passes
and quick A/B timing.  usage: prof_mlp.py [--iters 10] [--precision bf16] [--rays 4096] [--samples 128]"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from mipnerf_pl_amd import MipNerf, _lib as L  # noqa: E402
import synthetic_inputs as orc  # noqa: E402  (synthetic weights)

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--precision", default="bf16")
ap.add_argument("--rays", type=int, default=4096)
ap.add_argument("--samples", type=int, default=128)
ap.add_argument("--grid", type=int, default=0)
ap.add_argument("--zero", action="store_true", help="all-zero weights and inputs (DVFS / power experiment)")
a = ap.parse_args()
dev = torch.device("cuda:0")
params = orc.make_params(seed=0, density_gain=40.0)
if a.zero:
    params = {k: v * 0 for k, v in params.items()}
m = MipNerf(num_samples=a.samples, precision=a.precision)
m.load_state_dict({"mlp." + k: torch.from_numpy(v.copy()) for k, v in params.items()})
m = m.to(dev)
B, N = a.rays, a.samples
M = B * N
dt = torch.bfloat16 if m.precision == L.PREC_BF16 else torch.float32
enc = (torch.rand(M, 96, device=dev) * 2 - 1).to(dt)
venc = torch.zeros(B, 32, device=dev, dtype=dt)
venc[:, :27] = (torch.rand(B, 27, device=dev) * 2 - 1).to(dt)
if a.zero:
    enc.zero_()
    venc.zero_()
out = torch.empty(M, 4, device=dev)
ctx = m.mlp.native(dev)
if a.grid:
    ctx.set_option(1, a.grid)
ms = C.c_float()
L.check(L.lib().mipnerf_time_mlp(ctx.handle, M, N, enc.data_ptr(), venc.data_ptr(), m.precision, out.data_ptr(),
                                 a.iters, C.byref(ms), torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
tf = 1220608 * M / (ms.value * 1e-3) / 1e12
print(f"mlp {a.precision} M={M} grid={a.grid or 'CUs'}: {ms.value:.4f} ms/launch  {tf:.1f} TFLOP/s algorithmic")

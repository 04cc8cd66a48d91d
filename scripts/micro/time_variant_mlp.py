"""Time MLP.forward (mipnerf_mlp_forward: one kernel) of an architecture variant in bf16 and fp32 on the same inputs, and compare the two outputs.
usage: time_variant_mlp.py [w512|dc2|default] [rays] [samples per ray]; prints ms per launch, TFLOP/s (2 x MACs of the torch weights) and the
bf16-vs-fp32 differences.  Used for the round-5 record of the 512-wide trunk's bf16 kernel (profiles/README.md)."""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mipnerf_pl_amd import MipNerf  # noqa: E402

KW = {"w512": dict(mlp_net_width=512, mlp_net_width_condition=256), "dc2": dict(mlp_net_depth_condition=2), "default": {}}
name = sys.argv[1] if len(sys.argv) > 1 else "w512"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
N = int(sys.argv[3]) if len(sys.argv) > 3 else 128
torch.manual_seed(0)
x = torch.rand(B, N, 96, device="cuda") * 2 - 1
v = torch.rand(B, 27, device="cuda") * 2 - 1
outs = {}
for prec in ("bf16", "fp32"):
    torch.manual_seed(1)
    m = MipNerf(num_samples=N, precision=prec, **KW[name]).cuda()
    macs = sum(p.numel() for n_, p in m.mlp.named_parameters() if n_.endswith("weight"))
    with torch.no_grad():
        for _ in range(2):
            out = m.mlp(x, v)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 10
        e0.record()
        for _ in range(iters):
            out = m.mlp(x, v)
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    outs[prec] = out
    print(f"{name} {prec}: {ms:.4f} ms per MLP.forward of {B * N} samples (incl. the input cast), {2 * macs * B * N / ms / 1e9:.1f} TFLOP/s", flush=True)
d_rgb = float((outs["bf16"][0] - outs["fp32"][0]).abs().max())
d_den = float((outs["bf16"][1] - outs["fp32"][1]).abs().max())
print(f"{name}: bf16 vs fp32 raw rgb max diff {d_rgb:.3e}, raw density max diff {d_den:.3e} (max |density| {float(outs['fp32'][1].abs().max()):.3f})")

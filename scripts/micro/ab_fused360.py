"""round 6: the unbounded model's bf16 forward at BASELINE configs[3] shape, one-kernel form (option 6 = 1) against k_pre_gemm + trunk (0),
alternating in one process; usage: ab_fused360.py [rounds] [iterations]"""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import synthetic_inputs as syn  # noqa: E402
from mipnerf_pl_amd import MipNerf, Rays  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
B, N = 8192, 256
um = MipNerf(num_samples=N, precision="bf16", unbounded=True)
um.load_state_dict({"mlp." + k: torch.from_numpy(v.copy()) for k, v in syn.make_params(seed=0, density_gain=40.0, xyz_dim=672).items()})
um = um.cuda()
R = Rays(*[torch.from_numpy(a).cuda() for a in syn.synthetic_rays(B, seed=100, unbounded=True)])
ctx = um.mlp.native(torch.device("cuda:0"))
outs = {}
with torch.no_grad():
    for r in range(rounds):
        for form in (1, 0):
            ctx.set_option(6, form)
            um(R, False, True)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                out = um(R, False, True)
            e1.record()
            torch.cuda.synchronize()
            outs[form] = out
            print(f"round {r} {'one kernel ' if form else 'two kernels'}: {e0.elapsed_time(e1) / iters:.3f} ms per forward", flush=True)
print("bit-identical:", all(torch.equal(a, b) for lvl in range(2) for a, b in zip(outs[1][lvl], outs[0][lvl])))

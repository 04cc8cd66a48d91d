"""forward of the unbounded-scene model at BASELINE configs[3] shape (8192 rays x (256 + 256) samples), for rocprofv3 --kernel-trace --stats.
usage: prof_unbounded.py [fp32|bf16] [iterations] [bf16 form: 1 = one MLP kernel per level (default), 0 = k_pre_gemm + trunk]; prints the wall time
per forward (torch events)"""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import synthetic_inputs as syn  # noqa: E402
from mipnerf_pl_amd import MipNerf, Rays  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 4
B, N = 8192, 256
rays_np = syn.synthetic_rays(B, seed=100, unbounded=True)
um = MipNerf(num_samples=N, precision=prec, unbounded=True)
um.load_state_dict({"mlp." + k: torch.from_numpy(v.copy()) for k, v in syn.make_params(seed=0, density_gain=40.0, xyz_dim=672).items()})
um = um.cuda()
R = Rays(*[torch.from_numpy(a).cuda() for a in rays_np])
if len(sys.argv) > 3 and prec == "bf16":
    um.mlp.native(torch.device("cuda:0")).set_option(6, int(sys.argv[3]))
with torch.no_grad():
    um(R, False, True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        out = um(R, False, True)
    e1.record()
torch.cuda.synchronize()
print(f"unbounded {prec}: {e0.elapsed_time(e1) / iters:.3f} ms per forward, finite={bool(torch.isfinite(out[-1][0]).all())}")

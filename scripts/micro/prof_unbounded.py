import torch, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import synthetic_inputs as syn
from mipnerf_pl_amd import MipNerf, Rays
B, N = 8192, 256
rays_np = syn.synthetic_rays(B, seed=100, unbounded=True)
um = MipNerf(num_samples=N, precision="fp32", unbounded=True)
um.load_state_dict({"mlp." + k: torch.from_numpy(v.copy()) for k, v in syn.make_params(seed=0, density_gain=40.0, xyz_dim=672).items()})
um = um.cuda()
R = Rays(*[torch.from_numpy(a).cuda() for a in rays_np])
with torch.no_grad():
    for _ in range(4):
        um(R, False, True)
torch.cuda.synchronize()

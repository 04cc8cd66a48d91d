import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from mipnerf_pl_amd import MipNerf, Rays
from oracle import mipnerf_oracle as orc
dev = torch.device("cuda", 0)
B, N = 4096, 128
params = orc.make_params(seed=0, density_gain=40.0)
model = MipNerf(num_samples=N, precision="bf16")
model.load_state_dict({"mlp." + k: torch.from_numpy(v.copy()) for k, v in params.items()})
model = model.to(dev)
R = Rays(*[torch.from_numpy(a).to(dev) for a in orc.synthetic_rays(B, seed=100)])
ctx = model.mlp.native(dev)
def run(fused, steps=40):
    ctx.set_option(3, fused)
    with torch.no_grad():
        for _ in range(5): model(R, False, True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps): model(R, False, True)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
for rnd in range(4):
    print(f"round {rnd}: fused {run(1):.4f} ms/step   separate {run(0):.4f} ms/step", flush=True)

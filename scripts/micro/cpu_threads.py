"""How many threads should the reference's CPU forward get on this host?  (bench.py cpu_baseline)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import synthetic_inputs as syn
from oracle import ref as oref
r = oref.load()
B, N = 2048, 128
rays = syn.synthetic_rays(B, seed=100)
params = syn.make_params(seed=0, density_gain=40.0)
model = r.MipNerf(num_samples=N)
model.load_state_dict({"mlp." + k: torch.from_numpy(v.copy()) for k, v in params.items()})
RR = r.Rays(*[torch.from_numpy(a) for a in rays])
for th in (8, 16, 32, 64, 128, 256):
    torch.set_num_threads(th)
    with torch.no_grad():
        model(r.Rays(*[x[:256] for x in RR]), False, True)
        t0 = time.perf_counter(); model(RR, False, True); dt = time.perf_counter() - t0
    print(f"threads {th}: {dt:.2f} s  {B*N*2/dt:.3e} ray-samples/s", flush=True)

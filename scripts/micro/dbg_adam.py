"""Debug: FlatAdam (device schedule) vs torch.optim.Adam + MipLRDecay through the autograd training path, per step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
import gpu_util as G
from oracle import mipnerf_oracle as orc
from mipnerf_pl_amd.system import DEFAULT_HPARAMS, MipNeRFSystem
g = G.load_golden("train_64x64_trained")
rays, gt = G.to_dev(G.rays_of(g)), torch.from_numpy(g["gt"]).to(G.DEV)
def make(fused, device_lr=True):
    hp = dict(DEFAULT_HPARAMS)
    hp.update({'nerf.num_samples': 64, 'train.randomized': False, 'optimizer.lr_delay_steps': 3, 'optimizer.max_steps': 10})
    system = MipNeRFSystem(hp, precision="bf16")
    params = orc.make_params(seed=int(g["param_seed"]), density_gain=float(g["density_gain"]))
    system.load_state_dict({"mip_nerf.mlp." + k: torch.from_numpy(v.copy()) for k, v in params.items()})
    system = system.to(G.DEV)
    system.fused_adam = fused
    system.device_lr_schedule = device_lr
    (opt,), (sch,) = system.configure_optimizers()
    return system, opt, sch["scheduler"]
runs = {}
for tag, fused, dl in (("torch", False, True), ("flat_dev", True, True), ("flat_host", True, False)):
    system, opt, sch = make(fused, dl)
    hist = []
    for it in range(4):
        opt.zero_grad()
        lr_before = opt.param_groups[0]["lr"]
        loss = system.training_step((rays, gt), it)
        loss.backward()
        gr = torch.cat([p.grad.reshape(-1) for p in system.mip_nerf.parameters()]).clone()
        opt.step()
        sch.step()
        pr = torch.cat([p.detach().reshape(-1) for p in system.mip_nerf.parameters()]).clone()
        used = opt.last_lr() if hasattr(opt, "last_lr") and fused and dl else lr_before
        hist.append((float(loss.detach()), gr, pr, lr_before, used))
    runs[tag] = hist
for it in range(4):
    a = runs["torch"][it]
    for tag in ("flat_dev", "flat_host"):
        b = runs[tag][it]
        d = (a[2] - b[2]).abs()
        i = int(d.argmax())
        print(f"step {it} {tag}: loss {a[0]:.8f} vs {b[0]:.8f}  grad maxdiff {float((a[1]-b[1]).abs().max()):.3e}  param maxdiff {float(d.max()):.3e} at {i} "
              f"(grad there {float(a[1][i]):.3e} / {float(b[1][i]):.3e})  lr host {a[3]:.6e} vs {tag} {b[4]:.6e}")

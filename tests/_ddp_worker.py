"""Worker of tests/test_gpu_ddp.py (one process per rank, started by torch.distributed.run)."""
import json, os, sys, torch
sys.path.insert(0, os.getcwd())
import torch.distributed as dist
from torch.nn.parallel import DistributedDataParallel as DDP
import synthetic_inputs as syn
from mipnerf_pl_amd import Rays
from mipnerf_pl_amd.system import DEFAULT_HPARAMS, MipNeRFSystem
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("gloo", rank=rank, world_size=world)
B, N = 256, 64
hp = dict(DEFAULT_HPARAMS); hp.update({"nerf.num_samples": N, "train.randomized": False, "optimizer.lr_init": 1e-3, "optimizer.lr_delay_steps": 0})
params = syn.make_params(seed=0, density_gain=40.0)
def fresh():
    s = MipNeRFSystem(hp, precision="bf16")
    s.load_state_dict({"mip_nerf.mlp." + k: torch.from_numpy(v.copy()) for k, v in params.items()})
    return s.to(dev)
R = Rays(*[torch.from_numpy(a).to(dev) for a in syn.synthetic_rays(B, seed=10 + rank, multiscale=True)])      # this rank's shard
gt = torch.rand(B, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(20 + rank))
# local gradient, no DDP
ref = fresh()
assert ref._native_step_route(R)
ref.training_step((R, gt), 0).backward()
local = torch.cat([p.grad.reshape(-1) for p in ref.mip_nerf.parameters()])
both = [torch.empty_like(local) for _ in range(world)]
dist.all_gather(both, local)
want = sum(both) / world

class Wrapped(torch.nn.Module):          # what Lightning's LightningDistributedModule does: forward -> training_step
    def __init__(self, system):
        super().__init__()
        self.module = system
    def forward(self, batch, batch_idx):
        return self.module.training_step(batch, batch_idx)

system = fresh()
ddp = DDP(Wrapped(system), device_ids=[0], find_unused_parameters=False)
opt = torch.optim.Adam(system.mip_nerf.parameters(), lr=1e-3)
opt.zero_grad()
loss = ddp((R, gt), 0)
loss.backward()
got = torch.cat([p.grad.reshape(-1) for p in system.mip_nerf.parameters()])
err = float((got - want).abs().max() / want.abs().max())
opt.step()
for _ in range(2):                       # two more steps: the replicas stay identical
    opt.zero_grad(); ddp((R, gt), 0).backward(); opt.step()
p = torch.cat([q.detach().reshape(-1) for q in system.mip_nerf.parameters()])
ps = [torch.empty_like(p) for _ in range(world)]
dist.all_gather(ps, p)
print(json.dumps({"rank": rank, "grad_rel_err": err, "replicas_equal": bool(torch.equal(ps[0], ps[1])), "loss": float(loss),
                  "moved": bool((p - torch.cat([q.detach().reshape(-1) for q in ref.mip_nerf.parameters()])).abs().max() > 0)}))
dist.destroy_process_group()

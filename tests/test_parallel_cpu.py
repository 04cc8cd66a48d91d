"""CPU, world_size 2 over gloo: the N>1 pieces of the hot path (gradient all-reduce in one flat buffer,
ray sharding + gather for rendering).  No kernels run here."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mipnerf_pl_amd import Rays
from mipnerf_pl_amd.parallel import FlatGradAllReduce, gather_rendered, shard_bounds, shard_rays


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        params = [torch.nn.Parameter(torch.zeros(4, 3)), torch.nn.Parameter(torch.zeros(5)),
                  torch.nn.Parameter(torch.zeros(2, 2))]
        for i, p in enumerate(params[:2]):
            p.grad = torch.full_like(p, float(rank + 1) * (i + 1))      # rank-dependent gradients
        # params[2].grad stays None on purpose (an unused parameter)
        FlatGradAllReduce(params)()
        mean01 = (1 + 2) / 2.0
        ok = all(torch.allclose(p.grad, torch.full_like(p, mean01 * (i + 1))) for i, p in enumerate(params[:2]))
        ok = ok and torch.count_nonzero(params[2].grad) == 0
        # flat mode (MLP.flatten_parameters): the gradient buffer the wgrad reduction writes is all-reduced IN PLACE
        from mipnerf_pl_amd import MipNerf
        model = MipNerf(num_samples=8)
        mlp = model.mlp.flatten_parameters()
        assert mlp.is_flat() and mlp.grads_are_flat() and list(model.state_dict().keys())[0] == "mlp.layers.0.0.weight"
        mlp._flat_grad.copy_(torch.arange(mlp._flat_grad.numel(), dtype=torch.float32) * (rank + 1))
        ptr = mlp._flat_grad.data_ptr()
        FlatGradAllReduce(list(model.parameters()), mlp=mlp)()
        want = torch.arange(mlp._flat_grad.numel(), dtype=torch.float32) * 1.5
        ok = ok and mlp._flat_grad.data_ptr() == ptr and torch.allclose(mlp._flat_grad, want)
        ok = ok and torch.allclose(model.mlp.color_layer.bias.grad, want[-3:])     # .grad are views of the reduced buffer
        # asynchronous form (what GraphedTrainStep / bench use): SUM only, the mean is folded into the Adam kernel
        mlp._flat_grad.copy_(torch.arange(mlp._flat_grad.numel(), dtype=torch.float32) * (rank + 1))
        work = FlatGradAllReduce(list(model.parameters()), mlp=mlp).start()
        ok = ok and work is not None
        work.wait()
        ok = ok and mlp._flat_grad.data_ptr() == ptr and torch.allclose(mlp._flat_grad, want * 2)
        # rendering: shard 11 rays, "render" = 2*origin, gather
        n = 11
        rays = Rays(*[torch.arange(n * k, dtype=torch.float32).reshape(n, k) for k in (3, 3, 3, 1, 1, 1, 1)])
        local = shard_rays(rays, rank, world)
        lo, hi = shard_bounds(n, rank, world)
        ok = ok and local.origins.shape[0] == hi - lo
        full = gather_rendered(local.origins * 2, n)
        ok = ok and torch.equal(full, rays.origins * 2)
        out[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_flat_grad_allreduce_and_ray_shards_world2():
    world = 2
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
        assert dict(out) == {0: True, 1: True}


def _worker8(rank, world, port, out):
    """world 8 (what the driver's SCALE run goes to): the flat all-reduce with 1 / 8, and the frame split / gather at 640,000 rays (80,000 per
    rank), at a ragged count (8 does not divide 100,003) and with EMPTY shards (5 rays over 8 ranks)"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p = torch.nn.Parameter(torch.zeros(1000))
        p.grad = torch.full_like(p, float(rank + 1))
        FlatGradAllReduce([p])()
        ok = torch.allclose(p.grad, torch.full_like(p, 4.5))                 # mean of 1..8
        for n in (640000, 100003, 5):
            origins = torch.arange(n, dtype=torch.float32)[:, None] * torch.ones(1, 3)
            rays = Rays(origins, origins + 1, origins + 2, *[torch.ones(n, 1) for _ in range(4)])
            local = shard_rays(rays, rank, world)
            lo, hi = shard_bounds(n, rank, world)
            ok = ok and local.origins.shape[0] == hi - lo and (hi - lo in (n // world, n // world + 1))
            rendered = local.origins * 2 + 1                                 # a shard's "render" (empty for ranks >= 5 at n = 5)
            full = gather_rendered(rendered, n)
            ok = ok and full.shape == (n, 3) and torch.equal(full, origins * 2 + 1)
        out[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_flat_grad_allreduce_and_frame_gather_world8():
    world = 8
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker8, args=(world, port, out), nprocs=world, join=True)
        assert dict(out) == {r: True for r in range(world)}


@pytest.mark.parametrize("n,world", [(640000, 8), (10, 3), (7, 8), (4096, 4)])
def test_shard_bounds_partition(n, world):
    b = [shard_bounds(n, r, world) for r in range(world)]
    assert b[0][0] == 0 and b[-1][1] == n
    assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
    sizes = [hi - lo for lo, hi in b]
    assert max(sizes) - min(sizes) <= 1


def test_rayloader_equal_batches_per_rank_when_pixels_do_not_divide():
    """n % world != 0 must not give one rank an extra ray / batch (that rank would block in the gradient
    all-reduce): the permutation is padded by wrapping, like DistributedSampler(drop_last=False)."""
    from mipnerf_pl_amd.datasets import RayLoader

    class Stub:
        split, device = "train", torch.device("cpu")

        def __init__(self, n):
            self.n = n

        def __len__(self):
            return self.n

        def rays_at(self, ids):
            return ids

    # (1, 4, 1), (2, 8, 1): n < world - 1 -- the pad exceeds the list, DistributedSampler repeats it
    for n, world, bs in ((10, 3, 2), (101, 4, 5), (64, 8, 8), (7, 8, 1), (1000, 7, 13), (1, 4, 1), (2, 8, 1)):
        lens, seen = set(), torch.zeros(n, dtype=torch.int64)
        counts = []
        for r in range(world):
            ld = RayLoader(Stub(n), batch_size=bs, shuffle=True, seed=5, rank=r, world_size=world)
            batches = list(ld)
            assert len(batches) == len(ld)
            lens.add(len(ld))
            counts.append(sum(b.numel() for b in batches))
            for b in batches:
                seen[b] += 1
        assert len(lens) == 1, (n, world, lens)                      # same number of batches on every rank
        assert len(set(counts)) == 1 and counts[0] == -(-n // world)   # same number of rays on every rank
        assert int(seen.min()) >= 1 and int(seen.sum()) == world * (-(-n // world)) and int(seen.max()) <= max(2, -(-world // n))


def test_device_lr_scheduler_is_a_torch_lrscheduler():
    """what configure_optimizers returns with fused_adam must pass Lightning's scheduler validation."""
    from mipnerf_pl_amd.lr_schedule import DeviceMipLRDecay, mip_lr
    p = torch.nn.Parameter(torch.zeros(3))
    opt = torch.optim.Adam([p], lr=1.0)
    args = dict(lr_init=2e-3, lr_final=1e-4, max_steps=300, lr_delay_steps=30, lr_delay_mult=0.01)
    import warnings
    # without GraphedTrainStep driving the optimiser torch's scheduler-before-optimiser warning stays armed
    with pytest.warns(UserWarning, match="before `optimizer.step"):
        DeviceMipLRDecay(torch.optim.Adam([torch.nn.Parameter(torch.zeros(3))], lr=1.0), **args).step()
    opt._graph_driven = True             # what GraphedTrainStep.__init__ sets on its FlatAdam
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        s = DeviceMipLRDecay(opt, **args)
        assert isinstance(s, torch.optim.lr_scheduler.LRScheduler)
        assert s.last_epoch == 0 and opt.param_groups[0]["lr"] == mip_lr(0, *s.args)
        for _ in range(5):
            s.step()                       # no optimizer.step() in between (graph replay): must not warn
    assert s.last_epoch == 5 and abs(s.get_last_lr()[0] - mip_lr(5, *s.args)) < 1e-18
    s2 = DeviceMipLRDecay(torch.optim.Adam([p], lr=1.0), **args)
    s2.load_state_dict(s.state_dict())
    assert s2.last_epoch == 5
    try:
        from pytorch_lightning.core.optimizer import _validate_scheduler_api
        from pytorch_lightning.utilities.types import LRSchedulerConfig
    except Exception:      # noqa: BLE001  (not installed in this image)
        return
    _validate_scheduler_api([LRSchedulerConfig(scheduler=s, interval="step")], torch.nn.Module())

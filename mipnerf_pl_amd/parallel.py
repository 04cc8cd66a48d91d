"""Multi-GPU pieces of the hot path, one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

* training: the only collective is the gradient all-reduce of the 612,740 fp32 parameters (2.45 MB).
  It is latency-bound, so all gradients are packed into ONE flat buffer and reduced with ONE
  all_reduce (SUM, then / world) instead of 24 small ones.
* rendering: rays are independent -> contiguous ray shards per rank, gather of the 20 B/ray outputs.
"""
from __future__ import annotations

from typing import List, Sequence

import torch
import torch.distributed as dist


def shard_bounds(n: int, rank: int, world: int):
    """Contiguous, balanced [lo, hi) split of n rays over `world` ranks (first n % world ranks get one more)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_rays(rays, rank: int, world: int):
    lo, hi = shard_bounds(rays[0].shape[0], rank, world)
    return type(rays)(*[x[lo:hi].contiguous() for x in rays])


class FlatGradAllReduce:
    """One-buffer gradient all-reduce (mean) for a fixed parameter list."""

    def __init__(self, params: Sequence[torch.nn.Parameter], mlp=None):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.numel = sum(p.numel() for p in self.params)
        self.flat = None
        self.mlp = mlp      # optional: an MLP in flat mode (MLP.flatten_parameters) -> its gradient buffer is reduced in place

    def start(self, group=None):
        """Flat mode only: issue the ONE all-reduce (SUM) of the gradient buffer asynchronously and return the work handle
        (None at world 1).  c10d orders it behind the kernels already queued on the current stream and runs it on its own
        stream; the caller overlaps whatever does not need the gradient (next batch's rays / random draws), then calls
        `handle.wait()` and lets the optimiser take the mean (`FlatAdam.grad_scale = 1 / world`) -- no div_ kernel."""
        if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
            return None
        if self.mlp is None or not self.mlp.grads_are_flat():
            raise RuntimeError("FlatGradAllReduce.start needs an MLP in flat mode (MLP.flatten_parameters)")
        return dist.all_reduce(self.mlp._flat_grad, op=dist.ReduceOp.SUM, group=group, async_op=True)

    def __call__(self, group=None, mean: bool = True) -> None:
        if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
            return
        if self.mlp is not None and self.mlp.grads_are_flat():
            g = self.mlp._flat_grad
            dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group)
            if mean:
                g.div_(dist.get_world_size(group))
            return
        p0 = self.params[0]
        if self.flat is None or self.flat.device != p0.device:
            self.flat = torch.empty(self.numel, dtype=torch.float32, device=p0.device)
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                self.flat[off:off + n].zero_()
            else:
                self.flat[off:off + n].copy_(p.grad.reshape(-1))
            off += n
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        self.flat.div_(dist.get_world_size(group))
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                p.grad = self.flat[off:off + n].view_as(p).clone()
            else:
                p.grad.copy_(self.flat[off:off + n].view_as(p))
            off += n


def gather_rendered(local: torch.Tensor, n_total: int, group=None, force_collective: bool = False) -> torch.Tensor:
    """all_gather of per-ray outputs produced from shard_rays() shards -> [n_total, ...] on every rank.
    force_collective: go through the communicator at world 1 too (a 1-rank RCCL all_gather is legal: how a single-GPU box
    exercises the multi-GPU rendering path end to end)."""
    if not dist.is_available() or not dist.is_initialized():
        return local
    if dist.get_world_size(group) == 1 and not force_collective:
        return local
    world = dist.get_world_size(group)
    sizes = [shard_bounds(n_total, r, world) for r in range(world)]
    maxn = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((maxn,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    outs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad, group=group)
    return torch.cat([o[:hi - lo] for o, (lo, hi) in zip(outs, sizes)], dim=0)

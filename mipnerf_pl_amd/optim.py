"""Flat-buffer optimizer for the MLP: `torch.optim.Adam(self.mip_nerf.parameters(), lr)` of the
reference (nerf_system.py:71-72) as ONE HIP kernel over the 612,740 parameters, sharing the flat gradient buffer with
the weight-gradient reduction (written there directly) and with the RCCL all-reduce (reduced in place).

`FlatAdam` is a torch.optim.Optimizer with one param group, so `MipLRDecay` and any other LR scheduler drive
`param_groups[0]['lr']` as usual; the parameters stay ordinary leaf nn.Parameters (views of one flat storage)."""
from __future__ import annotations

import torch

from . import _lib as L
from . import ops


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, mlp, lr: float = 5e-4, betas=(0.9, 0.999), eps: float = 1e-8):
        if not mlp.is_flat():
            mlp.flatten_parameters()
        self.mlp = mlp
        super().__init__(mlp.ordered_params(), dict(lr=lr, betas=betas, eps=eps))
        self.exp_avg = torch.zeros_like(mlp._flat_param)
        self.exp_avg_sq = torch.zeros_like(mlp._flat_param)
        self.steps = 0

    def zero_grad(self, set_to_none: bool = True):
        """No kernel: the next backward overwrites the flat gradient (accumulate = 0) instead of adding to it."""
        self.mlp._flat_grad_valid = False

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        mlp = self.mlp
        if not mlp.is_flat():
            raise RuntimeError("FlatAdam: the MLP parameters are no longer views of the flat buffer "
                               "(module was moved / re-created); call mlp.flatten_parameters() and rebuild the optimizer")
        mlp.gather_foreign_grads()
        if not mlp._flat_grad_valid:
            return loss
        g = self.param_groups[0]
        self.steps += 1
        flat = mlp._flat_param
        L.check(L.lib().mipnerf_adam_step(flat.numel(), flat.data_ptr(), mlp._flat_grad.data_ptr(), self.exp_avg.data_ptr(),
                                          self.exp_avg_sq.data_ptr(), float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]),
                                          float(g["eps"]), self.steps, ops._stream()), "adam_step")
        mlp.invalidate_packed()
        return loss

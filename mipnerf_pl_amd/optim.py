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
        # The flat moments and the step count live in `self.state` (keyed on the first parameter), so that
        # Optimizer.state_dict() / load_state_dict() -- what a Lightning checkpoint stores -- save and restore them
        # like torch.optim.Adam's per-parameter state.
        self.state[self._key()] = {"step": torch.zeros((), dtype=torch.int64), "exp_avg": torch.zeros_like(mlp._flat_param),
                                   "exp_avg_sq": torch.zeros_like(mlp._flat_param)}

    def _key(self):
        return self.param_groups[0]["params"][0]

    @property
    def exp_avg(self):
        return self.state[self._key()]["exp_avg"]

    @property
    def exp_avg_sq(self):
        return self.state[self._key()]["exp_avg_sq"]

    @property
    def steps(self) -> int:
        return int(self.state[self._key()]["step"])

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        st = self.state[self._key()]
        flat = self.mlp._flat_param
        for k in ("exp_avg", "exp_avg_sq"):     # Optimizer.load_state_dict casts to the PARAMETER's shape-agnostic dtype/device only
            st[k] = st[k].to(device=flat.device, dtype=torch.float32).reshape(-1).clone()
            if st[k].numel() != flat.numel():
                raise ValueError(f"FlatAdam.load_state_dict: {k} has {st[k].numel()} elements, expected {flat.numel()}")
        st["step"] = torch.as_tensor(int(st["step"]), dtype=torch.int64)

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
        st = self.state[self._key()]
        st["step"] += 1
        flat = mlp._flat_param
        L.check(L.lib().mipnerf_adam_step(flat.numel(), flat.data_ptr(), mlp._flat_grad.data_ptr(), st["exp_avg"].data_ptr(),
                                          st["exp_avg_sq"].data_ptr(), float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]),
                                          float(g["eps"]), int(st["step"]), ops._stream()), "adam_step")
        mlp.invalidate_packed()
        return loss

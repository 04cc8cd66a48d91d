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
    """`schedule` (optional dict: lr_init, lr_final, max_steps, lr_delay_steps, lr_delay_mult) moves MipLRDecay onto the
    device: every step is `mipnerf_adam_step_scheduled` -- the step counter lives in device memory, the learning rate and
    the bias corrections are computed by a one-thread kernel, so the whole optimiser step takes no per-step host scalar
    and can be captured in a hipGraph (train_graph.GraphedTrainStep).  `grad_scale` multiplies the gradient inside the
    Adam kernel (1 / world_size after a SUM all-reduce)."""

    def __init__(self, mlp, lr: float = 5e-4, betas=(0.9, 0.999), eps: float = 1e-8, schedule=None):
        if not mlp.is_flat():
            mlp.flatten_parameters()
        self.mlp = mlp
        self.schedule = dict(schedule) if schedule else None
        self.grad_scale = 1.0
        self._dev_step = None       # device int64 step counter + float[4] hyper-parameters of the last step
        self._hyper = None
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
        if self._dev_step is not None:
            self._dev_step.fill_(int(st["step"]))

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
        if self.schedule is not None or self.grad_scale != 1.0:
            self.launch_scheduled()
        else:
            st["step"] += 1
            flat = mlp._flat_param
            L.check(L.lib().mipnerf_adam_step(flat.numel(), flat.data_ptr(), mlp._flat_grad.data_ptr(), st["exp_avg"].data_ptr(),
                                              st["exp_avg_sq"].data_ptr(), float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]),
                                              float(g["eps"]), int(st["step"]), ops._stream()), "adam_step")
        mlp.invalidate_packed()
        return loss

    # -- device-side schedule ---------------------------------------------------------------------------------------
    def _schedule_struct(self):
        g = self.param_groups[0]
        sc = self.schedule
        if sc is None:      # constant lr taken from the param group (an external scheduler may change it between steps)
            return L.LrSchedule(1.0, 1.0, 1.0, float(g["lr"]), 1, 0, float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
                                float(self.grad_scale), 0)
        return L.LrSchedule(float(sc["lr_init"]), float(sc["lr_final"]), float(sc["lr_delay_mult"]), 0.0, int(sc["max_steps"]),
                            int(sc["lr_delay_steps"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
                            float(self.grad_scale), 0)

    def launch_scheduled(self, count_on_host: bool = True):
        """One optimiser step with the step counter / learning rate on the device (capturable).  `count_on_host=False`
        leaves the host mirror of the step count alone (GraphedTrainStep bumps it per replay)."""
        import ctypes as C
        mlp = self.mlp
        st = self.state[self._key()]
        flat = mlp._flat_param
        if self._dev_step is None or self._dev_step.device != flat.device:
            self._dev_step = torch.full((1,), int(st["step"]), dtype=torch.int64, device=flat.device)
            self._hyper = torch.zeros(4, dtype=torch.float32, device=flat.device)
        sc = self._schedule_struct()
        L.check(L.lib().mipnerf_adam_step_scheduled(flat.numel(), flat.data_ptr(), mlp._flat_grad.data_ptr(),
                                                    st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), C.byref(sc),
                                                    self._dev_step.data_ptr(), self._hyper.data_ptr(), ops._stream()),
                "adam_step_scheduled")
        if count_on_host:
            st["step"] += 1

    def last_lr(self) -> float:
        """Learning rate the last device-scheduled step used (what nerf_system.py:117 logs); synchronises."""
        if self._hyper is None:
            return float(self.param_groups[0]["lr"])
        return float(self._hyper[0].item())

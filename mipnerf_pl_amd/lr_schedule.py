"""MipLRDecay: log-linear lr_init -> lr_final with a sin warm-up (reference: utils/lr_schedule.py:5-59).
Host scalar math; kept because MipNeRFSystem.configure_optimizers (nerf_system.py:70-76) returns it."""
import numpy as np
import torch


class MipLRDecay(torch.optim.lr_scheduler._LRScheduler):
    def __init__(self, optimizer, lr_init: float, lr_final: float, max_steps: int, lr_delay_steps: int,
                 lr_delay_mult: float):
        self.lr_init = lr_init
        self.lr_final = lr_final
        self.max_steps = max_steps
        self.lr_delay_steps = lr_delay_steps
        self.lr_delay_mult = lr_delay_mult
        super().__init__(optimizer)

    def get_lr(self):
        step = self.last_epoch
        if self.lr_delay_steps > 0:
            delay_rate = self.lr_delay_mult + (1 - self.lr_delay_mult) * np.sin(
                0.5 * np.pi * np.clip(step / self.lr_delay_steps, 0, 1))
        else:
            delay_rate = 1.
        t = np.clip(step / self.max_steps, 0, 1)
        log_lerp = np.exp(np.log(self.lr_init) * (1 - t) + np.log(self.lr_final) * t)
        return [delay_rate * log_lerp for _ in self.optimizer.param_groups]

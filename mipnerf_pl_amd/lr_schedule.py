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


def mip_lr(step: int, lr_init: float, lr_final: float, max_steps: int, lr_delay_steps: int, lr_delay_mult: float) -> float:
    """The schedule as a plain function of the scheduler epoch (same expression as MipLRDecay.get_lr)."""
    if lr_delay_steps > 0:
        delay_rate = lr_delay_mult + (1 - lr_delay_mult) * np.sin(0.5 * np.pi * np.clip(step / lr_delay_steps, 0, 1))
    else:
        delay_rate = 1.
    t = np.clip(step / max_steps, 0, 1)
    return float(delay_rate * np.exp(np.log(lr_init) * (1 - t) + np.log(lr_final) * t))


class DeviceMipLRDecay:
    """Scheduler object returned by MipNeRFSystem.configure_optimizers when the optimiser evaluates MipLRDecay on the
    device (FlatAdam(schedule=...)): `step()` only advances the host mirror of the epoch (no kernel, no write to
    param_groups that the device would have to read back), `get_last_lr()` evaluates the same formula on the host."""

    def __init__(self, optimizer, lr_init: float, lr_final: float, max_steps: int, lr_delay_steps: int, lr_delay_mult: float):
        self.optimizer = optimizer
        self.args = (lr_init, lr_final, max_steps, lr_delay_steps, lr_delay_mult)
        self.last_epoch = 0
        optimizer.param_groups[0]["lr"] = mip_lr(0, *self.args)

    def step(self):
        self.last_epoch += 1
        self.optimizer.param_groups[0]["lr"] = mip_lr(self.last_epoch, *self.args)      # informational (logging)

    def get_last_lr(self):
        return [mip_lr(self.last_epoch, *self.args)]

    def state_dict(self):
        return {"last_epoch": self.last_epoch}

    def load_state_dict(self, sd):
        self.last_epoch = int(sd["last_epoch"])

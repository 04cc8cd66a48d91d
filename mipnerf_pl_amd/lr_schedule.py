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


class DeviceMipLRDecay(MipLRDecay):
    """Scheduler object returned by MipNeRFSystem.configure_optimizers when the optimiser evaluates MipLRDecay on the
    device (FlatAdam(schedule=...)).  It IS a torch LRScheduler (Lightning's `_validate_scheduler_api` accepts only those
    unless `lr_scheduler_step` is overridden), i.e. MipLRDecay itself: `step()` advances the epoch and writes the host value
    of the same formula into `param_groups[0]['lr']` -- informational (logging, checkpoints): the device-scheduled Adam
    kernel never reads it, it derives the rate from its own device-side step counter."""

    def __init__(self, optimizer, lr_init: float, lr_final: float, max_steps: int, lr_delay_steps: int, lr_delay_mult: float):
        self.args = (lr_init, lr_final, max_steps, lr_delay_steps, lr_delay_mult)
        super().__init__(optimizer, lr_init, lr_final, max_steps, lr_delay_steps, lr_delay_mult)

    def step(self, epoch=None):
        # the optimiser step may have been a graph replay (GraphedTrainStep) that never went through optimizer.step():
        # torch's "lr_scheduler.step() before optimizer.step()" warning does not apply
        if getattr(self.optimizer, "_graph_driven", False):       # set by GraphedTrainStep; otherwise torch's warning stays armed
            self.optimizer._opt_called = True
        return super().step() if epoch is None else super().step(epoch)

"""Learning-rate rules as pure functions (used by the callbacks and directly by scripts).

Linear scaling ``lr × size`` (``imagenet_pytorch_horovod.py:333``), Goyal et al. gradual
warm-up over 5 epochs and the ×1/×0.1/×0.01/×0.001 staircase at epochs 30/60/80
(``imagenet_keras_horovod.py:208-224``).  ``poly_lr`` is the polynomial decay customarily
paired with LARS for large-batch runs (BASELINE.json config 4).
"""
from __future__ import annotations

from ..callbacks import warmup_multiplier

__all__ = ["scaled_lr", "staircase_multiplier", "goyal_lr", "poly_lr", "LRSchedule"]


def scaled_lr(base_lr: float, size: int) -> float:
    return base_lr * size


def staircase_multiplier(epoch, boundaries=(30, 60, 80), factor=0.1) -> float:
    m = 1.0
    for b in boundaries:
        if epoch >= b:
            m *= factor
    return m


def goyal_lr(epoch_float, base_lr, size, warmup_epochs=5, boundaries=(30, 60, 80), factor=0.1):
    """LR at fractional ``epoch_float`` for a job of ``size`` workers."""
    target = scaled_lr(base_lr, size)
    if epoch_float < warmup_epochs:
        return target * warmup_multiplier(epoch_float, warmup_epochs, size)
    return target * staircase_multiplier(int(epoch_float), boundaries, factor)


def poly_lr(step, total_steps, peak_lr, warmup_steps=0, power=2.0, end_lr=0.0):
    if warmup_steps and step < warmup_steps:
        return peak_lr * (step + 1) / warmup_steps
    span = max(total_steps - warmup_steps, 1)
    frac = min(max(step - warmup_steps, 0) / span, 1.0)
    return (peak_lr - end_lr) * (1.0 - frac) ** power + end_lr


class LRSchedule:
    """Stateful wrapper: ``sched.apply(optimizer, epoch, batch)`` sets every group's LR."""

    def __init__(self, base_lr, size, steps_per_epoch, warmup_epochs=5,
                 boundaries=(30, 60, 80), factor=0.1):
        self.base_lr, self.size, self.steps_per_epoch = base_lr, size, max(steps_per_epoch, 1)
        self.warmup_epochs, self.boundaries, self.factor = warmup_epochs, boundaries, factor
        self.last = None

    def lr_at(self, epoch, batch=0):
        return goyal_lr(epoch + batch / self.steps_per_epoch, self.base_lr, self.size,
                        self.warmup_epochs, self.boundaries, self.factor)

    def apply(self, optimizer, epoch, batch=0):
        self.last = self.lr_at(epoch, batch)
        for g in optimizer.param_groups:
            g["lr"] = self.last
        return self.last

    def state_dict(self):
        return {"last": self.last}

    def load_state_dict(self, state):
        self.last = state.get("last")

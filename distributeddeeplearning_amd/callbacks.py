"""Training callbacks: the PyTorch-side equivalents of ``hvd.callbacks.*``.

Parity targets (``HorovodKeras/src/imagenet_keras_horovod.py:194-227``):

* ``BroadcastGlobalVariablesCallback(root)``  — ``:202`` (and TF ``BroadcastGlobalVariablesHook``)
* ``MetricAverageCallback()``                 — ``:207``
* ``LearningRateWarmupCallback(warmup_epochs=5)`` — ``:211-213``; Goyal et al. gradual warm-up,
  per-batch granularity, ``lr' = lr · (1/size · (epoch·(size−1)/warmup + 1))`` [EXT formula]
* ``LearningRateScheduleCallback(start_epoch, end_epoch, multiplier)`` — ``:215-224``; piecewise
  multipliers on the *initial* LR, with momentum correction when the LR changes [EXT]
* ``LoggerCallback`` — ``:230-244`` per-epoch duration/metrics logging
* rank-0 ``ModelCheckpoint("checkpoint-{epoch}")`` — ``:310-319`` → ``CheckpointCallback``

They are framework-free: a training loop calls ``CallbackList`` hooks (``on_train_begin``,
``on_epoch_begin``, ``on_batch_begin``, ``on_batch_end``, ``on_epoch_end``) and the callbacks
mutate the optimizer's ``param_groups`` / a ``logs`` dict.
"""
from __future__ import annotations

from .parallel import backend as _b
from .parallel import functions as _f
from .utils.timer import Timer

__all__ = ["Callback", "CallbackList", "BroadcastGlobalVariablesCallback",
           "MetricAverageCallback", "LearningRateScheduleCallback",
           "LearningRateWarmupCallback", "LoggerCallback", "CheckpointCallback",
           "warmup_multiplier", "default_imagenet_schedule"]


class Callback:
    model = None
    optimizer = None
    steps_per_epoch = None

    def bind(self, model=None, optimizer=None, steps_per_epoch=None):
        self.model, self.optimizer, self.steps_per_epoch = model, optimizer, steps_per_epoch
        return self

    def on_train_begin(self, logs=None):
        pass

    def on_epoch_begin(self, epoch, logs=None):
        pass

    def on_batch_begin(self, batch, logs=None):
        pass

    def on_batch_end(self, batch, logs=None):
        pass

    def on_epoch_end(self, epoch, logs=None):
        pass

    def on_train_end(self, logs=None):
        pass


class CallbackList(Callback):
    def __init__(self, callbacks=(), model=None, optimizer=None, steps_per_epoch=None):
        self.callbacks = list(callbacks)
        self.bind(model, optimizer, steps_per_epoch)

    def bind(self, model=None, optimizer=None, steps_per_epoch=None):
        super().bind(model, optimizer, steps_per_epoch)
        for cb in self.callbacks:
            cb.bind(model, optimizer, steps_per_epoch)
        return self

    def on_train_begin(self, logs=None):
        for cb in self.callbacks:
            cb.on_train_begin(logs)

    def on_epoch_begin(self, epoch, logs=None):
        for cb in self.callbacks:
            cb.on_epoch_begin(epoch, logs)

    def on_batch_begin(self, batch, logs=None):
        for cb in self.callbacks:
            cb.on_batch_begin(batch, logs)

    def on_batch_end(self, batch, logs=None):
        for cb in self.callbacks:
            cb.on_batch_end(batch, logs)

    def on_epoch_end(self, epoch, logs=None):
        for cb in self.callbacks:
            cb.on_epoch_end(epoch, logs)

    def on_train_end(self, logs=None):
        for cb in self.callbacks:
            cb.on_train_end(logs)


class BroadcastGlobalVariablesCallback(Callback):
    """Equalise model state and optimizer state from ``root_rank`` at train begin."""

    def __init__(self, root_rank=0):
        self.root_rank = root_rank
        self.done = False

    def on_train_begin(self, logs=None):
        if self.done:
            return
        if self.model is not None:
            _f.broadcast_parameters(self.model.state_dict(), self.root_rank)
        if self.optimizer is not None:
            _f.broadcast_optimizer_state(self.optimizer, self.root_rank)
        self.done = True


class MetricAverageCallback(Callback):
    """Replace every numeric entry of ``logs`` by its average over ranks at epoch end."""

    def on_epoch_end(self, epoch, logs=None):
        if not logs or _b.size() == 1:
            return
        keys = sorted(k for k, v in logs.items() if isinstance(v, (int, float)))
        if not keys:
            return
        averaged = _f.metric_average([logs[k] for k in keys], name="metric_average")
        for k, v in zip(keys, averaged):
            logs[k] = v


def warmup_multiplier(epoch_float, warmup_epochs, size):
    """Goyal et al. gradual warm-up factor on the *scaled* LR: ramps 1/size → 1."""
    if warmup_epochs <= 0 or epoch_float >= warmup_epochs:
        return 1.0
    return 1.0 / size * (epoch_float * (size - 1) / warmup_epochs + 1.0)


class LearningRateScheduleCallback(Callback):
    """LR = initial_lr × multiplier for epochs in [start_epoch, end_epoch).

    ``multiplier`` may be a constant or a function of the (fractional, when
    ``staircase=False``) epoch.  ``momentum_correction`` applies Goyal et al.'s fix: when the
    LR changes, the momentum buffer contribution is rescaled by new_lr/old_lr for that step.
    """

    def __init__(self, multiplier, start_epoch=0, end_epoch=None, staircase=True,
                 momentum_correction=True, steps_per_epoch=None, initial_lr=None):
        self.start_epoch = start_epoch
        self.end_epoch = end_epoch
        self.staircase = staircase
        self.momentum_correction = momentum_correction
        self.steps_per_epoch = steps_per_epoch
        self.initial_lr = initial_lr
        self.current_epoch = 0
        self._restore_momentum = None
        self.multiplier = multiplier if callable(multiplier) else (lambda epoch: multiplier)

    def bind(self, model=None, optimizer=None, steps_per_epoch=None):
        keep = self.steps_per_epoch
        super().bind(model, optimizer, steps_per_epoch)
        if steps_per_epoch is None:
            self.steps_per_epoch = keep
        return self

    def _initial_lrs(self):
        if self.initial_lr is not None:
            return [self.initial_lr] * len(self.optimizer.param_groups)
        lrs = []
        for g in self.optimizer.param_groups:
            g.setdefault("initial_lr", g["lr"])
            lrs.append(g["initial_lr"])
        return lrs

    def on_train_begin(self, logs=None):
        self._initial_lrs()

    def _in_range(self, epoch):
        return epoch >= self.start_epoch and (self.end_epoch is None or epoch < self.end_epoch)

    def _adjust(self, epoch_value):
        for g, base in zip(self.optimizer.param_groups, self._initial_lrs()):
            old = g["lr"]
            new = base * self.multiplier(epoch_value)
            g["lr"] = new
            if self.momentum_correction and "momentum" in g and old > 0 and new != old:
                if self._restore_momentum is None:
                    self._restore_momentum = []
                self._restore_momentum.append((g, g["momentum"]))
                g["momentum"] = g["momentum"] * new / old

    def on_epoch_begin(self, epoch, logs=None):
        self.current_epoch = epoch
        if self.staircase and self._in_range(epoch):
            self._adjust(epoch)

    def on_batch_begin(self, batch, logs=None):
        if self.staircase or not self._in_range(self.current_epoch):
            return
        if not self.steps_per_epoch:
            raise ValueError("steps_per_epoch is required for per-batch LR adjustment")
        self._adjust(self.current_epoch + float(batch) / self.steps_per_epoch)

    def on_batch_end(self, batch, logs=None):
        if self._restore_momentum:
            for g, m in self._restore_momentum:
                g["momentum"] = m
            self._restore_momentum = None

    def on_epoch_end(self, epoch, logs=None):
        if logs is not None and self.optimizer is not None:
            logs["lr"] = self.optimizer.param_groups[0]["lr"]


class LearningRateWarmupCallback(LearningRateScheduleCallback):
    """Gradual warm-up from lr/size to lr over ``warmup_epochs`` (per batch)."""

    def __init__(self, warmup_epochs=5, momentum_correction=True, steps_per_epoch=None,
                 verbose=0, initial_lr=None, size=None):
        self.warmup_epochs = warmup_epochs
        self.verbose = verbose
        self._size = size

        def multiplier(epoch):
            # "+ 1/steps" so that the last warm-up batch lands exactly on the target LR
            world = self._size if self._size is not None else _b.size()
            spe = self.steps_per_epoch or 1
            return warmup_multiplier(epoch + 1.0 / spe, warmup_epochs, world)

        super().__init__(multiplier, start_epoch=0, end_epoch=warmup_epochs, staircase=False,
                         momentum_correction=momentum_correction,
                         steps_per_epoch=steps_per_epoch, initial_lr=initial_lr)

    def on_epoch_end(self, epoch, logs=None):
        super().on_epoch_end(epoch, logs)
        if epoch == self.warmup_epochs - 1 and self.verbose:
            print("Epoch %d: finished gradual learning rate warmup to %g."
                  % (epoch + 1, self.optimizer.param_groups[0]["lr"]))


def default_imagenet_schedule(warmup_epochs=5, verbose=0):
    """The reference's Keras schedule (``imagenet_keras_horovod.py:208-224``): warm-up for 5
    epochs, then ×1 until 30, ×0.1 until 60, ×0.01 until 80, ×0.001 after."""
    return [
        LearningRateWarmupCallback(warmup_epochs=warmup_epochs, verbose=verbose),
        LearningRateScheduleCallback(1.0, start_epoch=warmup_epochs, end_epoch=30),
        LearningRateScheduleCallback(1e-1, start_epoch=30, end_epoch=60),
        LearningRateScheduleCallback(1e-2, start_epoch=60, end_epoch=80),
        LearningRateScheduleCallback(1e-3, start_epoch=80),
    ]


class LoggerCallback(Callback):
    """Logs epoch duration and metrics (reference ``LoggerCallback`` :230-244)."""

    def __init__(self, logger, data_length=None):
        self.logger = logger
        self.data_length = data_length
        self._timer = Timer()

    def on_epoch_begin(self, epoch, logs=None):
        self.logger.set_epoch(epoch)
        self._timer.start()

    def on_epoch_end(self, epoch, logs=None):
        self._timer.stop()
        duration = self._timer.elapsed
        self.logger.info("Epoch duration: {:.3f}".format(duration))
        if self.data_length:
            self.logger.info("Images/sec: {:.3f}".format(self.data_length / duration))
        for k, v in sorted((logs or {}).items()):
            self.logger.info("{}: {}".format(k, v))


class CheckpointCallback(Callback):
    """Rank-0-only checkpoint at every epoch end (other ranks write nothing)."""

    def __init__(self, directory, scheduler=None, keep_last=None):
        self.directory = directory
        self.scheduler = scheduler
        self.keep_last = keep_last

    def on_epoch_end(self, epoch, logs=None):
        from .utils import checkpoint

        if self.directory and _b.rank() == 0:
            checkpoint.save(self.directory, epoch + 1, self.model, self.optimizer,
                            extra={"logs": dict(logs or {})}, keep_last=self.keep_last)

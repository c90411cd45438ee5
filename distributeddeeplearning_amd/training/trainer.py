"""``Trainer``: the ``model.fit_generator`` / ``evaluate_generator`` flow of the reference's
Keras flavour (``HorovodKeras/src/imagenet_keras_horovod.py:299-353``) for PyTorch models.

    trainer = Trainer(model, optimizer, callbacks=[...], metrics=("accuracy", "top_k"))
    trainer.fit(loader, epochs=90, initial_epoch=resume_from_epoch, steps_per_epoch=n)
    scores = trainer.evaluate(val_loader)            # rank-averaged [loss, acc, top-5]

Callbacks are the PyTorch-side ``hvd.callbacks`` (``distributeddeeplearning_amd.callbacks``):
broadcast-at-start, metric averaging, LR warm-up and step schedule, logging, rank-0
checkpoints — the list ``_get_hooks`` builds at ``:194-227``.
"""
from __future__ import annotations

import torch

from .. import torch as hvd
from ..callbacks import CallbackList
from ..ops import functional as ops
from ..utils import trace
from .loop import to_device_batch

__all__ = ["Trainer"]


class Trainer:
    def __init__(self, model, optimizer, loss_fn=None, callbacks=(), device=None, dtype=None,
                 top_k=5):
        self.model = model
        self.optimizer = optimizer
        self.loss_fn = loss_fn or ops.softmax_cross_entropy
        self.callbacks = list(callbacks)
        p = next(model.parameters())
        self.device = torch.device(device) if device is not None else p.device
        conv_dtypes = [q.dtype for q in model.parameters() if q.dim() == 4]
        self.dtype = dtype or (conv_dtypes[0] if conv_dtypes else p.dtype)
        self.top_k = top_k
        self.history = []

    def _labels(self, target):
        # Keras iterators deliver one-hot labels (data_generator.py:45-52)
        return target.argmax(dim=1) if target.dim() == 2 else target

    def fit(self, loader, epochs=1, initial_epoch=0, steps_per_epoch=None, verbose=0,
            validation_loader=None, validation_steps=None):
        if steps_per_epoch is None:
            steps_per_epoch = len(loader)
        cbs = CallbackList(self.callbacks, self.model, self.optimizer, steps_per_epoch)
        cbs.on_train_begin({})
        for epoch in range(initial_epoch, epochs):
            self.model.train()
            if hasattr(loader, "set_epoch"):
                loader.set_epoch(epoch)
            sampler = getattr(loader, "sampler", None)
            if hasattr(sampler, "set_epoch"):
                sampler.set_epoch(epoch)
            logs = {}
            cbs.on_epoch_begin(epoch, logs)
            loss_sum = torch.zeros((), dtype=torch.float64, device=self.device)
            hits = torch.zeros(2, dtype=torch.float64, device=self.device)
            seen = 0
            for step, (data, target) in enumerate(loader):
                if step >= steps_per_epoch:
                    break
                cbs.on_batch_begin(step, logs)
                data, target = to_device_batch(data, target, self.device, self.dtype)
                target = self._labels(target)
                self.optimizer.zero_grad()
                with trace.range("fwd"):
                    out = self.model(data)
                    loss = self.loss_fn(out, target)
                with trace.range("bwd"):
                    loss.backward()
                with trace.range("step"):
                    self.optimizer.step()
                with torch.no_grad():
                    n = len(data)
                    loss_sum += loss.detach().double() * n
                    hits += ops.topk_hits(out, target, (1, self.top_k))
                    seen += n
                cbs.on_batch_end(step, logs)
            seen = max(seen, 1)
            logs.update({"loss": float(loss_sum) / seen, "acc": float(hits[0]) / seen,
                         "top_k_categorical_accuracy": float(hits[1]) / seen})
            if validation_loader is not None:
                val = self.evaluate(validation_loader, steps=validation_steps, average=False)
                logs.update({"val_loss": val[0], "val_acc": val[1],
                             "val_top_k_categorical_accuracy": val[2]})
            cbs.on_epoch_end(epoch, logs)
            self.history.append(dict(logs, epoch=epoch))
            if verbose:
                print("Epoch %d: %s" % (epoch + 1, logs))
        cbs.on_train_end({})
        return self.history

    @torch.no_grad()
    def evaluate(self, loader, steps=None, average=True):
        """[loss, accuracy, top-k accuracy]; ``average=True`` allreduce-averages over ranks
        like ``hvd.allreduce(model.evaluate_generator(...))`` (:348-350)."""
        was_training = self.model.training
        self.model.eval()
        loss_sum = torch.zeros((), dtype=torch.float64, device=self.device)
        hits = torch.zeros(2, dtype=torch.float64, device=self.device)
        seen = 0
        for step, (data, target) in enumerate(loader):
            if steps is not None and step >= steps:
                break
            data, target = to_device_batch(data, target, self.device, self.dtype)
            target = self._labels(target)
            out = self.model(data)
            n = len(data)
            loss_sum += self.loss_fn(out, target).double() * n
            hits += ops.topk_hits(out, target, (1, self.top_k))
            seen += n
        if was_training:
            self.model.train()
        seen = max(seen, 1)
        scores = [float(loss_sum) / seen, float(hits[0]) / seen, float(hits[1]) / seen]
        if average and hvd.is_initialized() and hvd.size() > 1:
            scores = hvd.allreduce(scores)
        return scores

    @torch.no_grad()
    def predict(self, loader, steps=None):
        self.model.eval()
        outs = []
        for step, batch in enumerate(loader):
            if steps is not None and step >= steps:
                break
            data = batch[0] if isinstance(batch, (tuple, list)) else batch
            data, _ = to_device_batch(data, torch.zeros(0), self.device, self.dtype)
            outs.append(self.model(data).float().cpu())
        return torch.cat(outs) if outs else torch.empty(0)

"""``Estimator``: the ``tf.estimator`` flow of the reference's TensorFlow flavour
(``HorovodTF/src/imagenet_estimator_tf_horovod.py``) for PyTorch.

The user supplies ``model_fn(features, labels, mode, params) -> EstimatorSpec`` (``:170-220``)
and input functions returning iterables of ``(features, labels)`` (``:235-345``); the
estimator owns the step loop, hooks (``BroadcastGlobalVariablesHook(0)`` ``:377-384``),
rank-0-only ``model_dir`` checkpoints (``:364-374``) and the TRAIN / EVAL / PREDICT modes.

The network, optimizer and any other state live in a ``state`` dict that ``model_fn`` creates
on the first call (``params["state"]``), which plays the role of TF's variable collections.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Callable, Optional

import torch

from .. import torch as hvd
from ..utils import checkpoint

__all__ = ["ModeKeys", "EstimatorSpec", "RunConfig", "Estimator", "BroadcastGlobalVariablesHook"]


class ModeKeys:
    TRAIN = "train"
    EVAL = "eval"
    PREDICT = "infer"


@dataclass
class EstimatorSpec:
    mode: str
    predictions: Any = None
    loss: Optional[torch.Tensor] = None
    train_op: Optional[Callable[[], None]] = None  # backward + optimizer step
    eval_metric_ops: dict = field(default_factory=dict)


@dataclass
class RunConfig:
    """Reference ``_get_runconfig`` (:348-361): pin the GPU by local rank; periodic checkpoints
    disabled when distributed unless requested."""
    model_dir: Optional[str] = None
    save_checkpoints_steps: Optional[int] = None
    log_step_count_steps: int = 100
    device: Optional[str] = None


class BroadcastGlobalVariablesHook:
    """Broadcast model + optimizer state from ``root_rank`` once the state exists."""

    def __init__(self, root_rank=0):
        self.root_rank = root_rank
        self.done = False

    def after_create_session(self, state):
        if self.done:
            return
        model, opt = state.get("model"), state.get("optimizer")
        if model is not None:
            hvd.broadcast_parameters(model.state_dict(), self.root_rank)
        if opt is not None:
            hvd.broadcast_optimizer_state(opt, self.root_rank)
        self.done = True

    def before_run(self, step):
        pass

    def after_run(self, step, loss):
        pass


class Estimator:
    def __init__(self, model_fn, model_dir=None, params=None, config=None):
        self.model_fn = model_fn
        self.config = config or RunConfig(model_dir=model_dir)
        self.model_dir = model_dir if model_dir is not None else self.config.model_dir
        self.params = dict(params or {})
        self.params.setdefault("state", {})
        self.global_step = 0

    @property
    def state(self):
        return self.params["state"]

    def _maybe_restore(self):
        st = self.state
        if self.model_dir and st.get("model") is not None and not st.get("_restored"):
            self.global_step = st.get("global_step", 0)
            epoch = checkpoint.resume(self.model_dir, st["model"], st.get("optimizer"))
            st["_restored"] = True
            if epoch:
                self.global_step = epoch

    def train(self, input_fn, steps=None, hooks=(), max_steps=None):
        hooks = list(hooks)
        created = False
        done = 0
        for features, labels in input_fn():
            if steps is not None and done >= steps:
                break
            if max_steps is not None and self.global_step >= max_steps:
                break
            spec = self.model_fn(features, labels, ModeKeys.TRAIN, self.params)
            if not created:
                self._maybe_restore()
                for h in hooks:
                    h.after_create_session(self.state)
                created = True
                # restore / broadcast rewrote the variables in place: rebuild the step on them
                spec = self.model_fn(features, labels, ModeKeys.TRAIN, self.params)
            for h in hooks:
                h.before_run(self.global_step)
            spec.train_op()
            self.global_step += 1
            done += 1
            for h in hooks:
                h.after_run(self.global_step, spec.loss)
            n = self.config.save_checkpoints_steps
            if n and self.global_step % n == 0:
                self._save()
        self._save()
        return self

    def _save(self):
        st = self.state
        if self.model_dir and st.get("model") is not None and (not hvd.is_initialized()
                                                               or hvd.rank() == 0):
            checkpoint.save(self.model_dir, self.global_step, st["model"], st.get("optimizer"))

    @torch.no_grad()
    def evaluate(self, input_fn, steps=None):
        totals, count = {}, 0
        for i, (features, labels) in enumerate(input_fn()):
            if steps is not None and i >= steps:
                break
            spec = self.model_fn(features, labels, ModeKeys.EVAL, self.params)
            n = len(labels)
            metrics = dict(spec.eval_metric_ops)
            if spec.loss is not None:
                metrics.setdefault("loss", spec.loss)
            for k, v in metrics.items():
                totals[k] = totals.get(k, 0.0) + float(v) * n
            count += n
        out = {k: v / max(count, 1) for k, v in totals.items()}
        out["global_step"] = self.global_step
        return out

    @torch.no_grad()
    def predict(self, input_fn):
        for batch in input_fn():
            features = batch[0] if isinstance(batch, (tuple, list)) else batch
            spec = self.model_fn(features, None, ModeKeys.PREDICT, self.params)
            yield spec.predictions

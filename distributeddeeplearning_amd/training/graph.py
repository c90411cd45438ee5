"""Whole-step HIP graph: zero_grad + forward + loss + backward + optimizer step in ONE launch.

The reference's hot loop (``HorovodPytorch/src/imagenet_pytorch_horovod.py:210-218``) issues
every device operation from Python, every iteration; so does this framework's eager path
(~700 kernel launches per ResNet-50 step, ``host_issue_ms_per_step`` in ``bench.py``).  With
static shapes the launch sequence never changes, so :class:`GraphedTrainStep` records it once
into a ``hipGraph`` (through ``torch.cuda.CUDAGraph``) and afterwards replays it with a single
``hipGraphLaunch`` per step — the host is out of the loop and a slow or contended CPU (8 ranks
on one node) no longer shows up in step time.

What makes the framework's step capturable:

* every hand-written kernel is launched on ``torch.cuda.current_stream()`` (the capture stream).
  The weight-gradient side stream forks from / joins it with events, which stream capture does
  follow - but ROCm 7.2 replays a graph with ~50 cross-stream edges at ~25 us per node (29 ms
  for the 19 ms step, 13 ms even at batch 16), while the same step captured on one stream
  replays at the speed of eager single-stream issue.  ``single_stream=True`` (default) therefore
  records the weight gradients on the capture stream (same kernels, same order of every sum);
* scratch memory comes from torch's allocator (private pool during capture), gradients are
  static views into the arenas (``parallel/bucket.py``), ``zero_grad`` is a memset node;
* nothing in the step reads a value back to the host; the loss stays on the device;
* the learning rate is a by-value kernel argument times a device-side scalar
  (``optim/fused.py::freeze_lr``), so LR schedules keep working: :meth:`GraphedTrainStep.__call__`
  rewrites that scalar outside the graph when ``param_groups[0]["lr"]`` moved.  Momentum, weight
  decay and the other hyper-parameters are by-value arguments: a step on which they differ from
  the captured ones (momentum correction) runs eagerly;

The first ``warmup`` calls run eagerly (lazy initialisation: kernel library, side stream,
transposed-weight tables, MIOpen-free but allocator warm-up), the next call captures and
replays, later calls copy the batch into the static input buffers and replay.  Eager and
graphed steps execute the same kernels in the same order, so results are bit-identical
(``tests/test_gpu_kernels.py::test_graphed_train_step_matches_eager``).

Gradient exchange (world > 1) inside a captured step relies on the communication library being
capture-safe; it is therefore opt-in (``allow_collectives=True``) and any capture failure falls
back to eager execution with a warning instead of aborting the job.
"""
from __future__ import annotations

import warnings

import torch

__all__ = ["GraphedTrainStep"]


class GraphedTrainStep:
    def __init__(self, model, optimizer, loss_fn, warmup: int = 3, allow_collectives: bool = False,
                 enabled: bool = True, single_stream: bool = True):
        """``loss_fn(model_output, labels) -> scalar tensor``; ``optimizer`` is a FusedSGD /
        FusedLARS (optionally wrapped by ``hvd.DistributedOptimizer``) or any optimizer whose
        ``step`` is capture-safe."""
        self.model, self.optimizer, self.loss_fn = model, optimizer, loss_fn
        self.warmup = max(int(warmup), 1)
        self.allow_collectives = allow_collectives
        self.enabled = enabled
        self.single_stream = single_stream
        self.calls = 0
        self.graph = None
        self.failed = None          # the exception that made capture fall back to eager
        self._static_in = None
        self._static_loss = None
        self._pool = None
        self._captured_hyper = None

    def _hyper(self):
        """Everything but the learning rate is a by-value kernel argument inside the graph."""
        return [tuple((k, v) for k, v in sorted(g.items()) if k not in ("params", "lr") and
                      isinstance(v, (int, float, bool)))
                for g in self.optimizer.param_groups]

    # ---------------------------------------------------------------------------------------
    def _eager(self, images, labels):
        self.optimizer.zero_grad()
        loss = self.loss_fn(self.model(images), labels)
        loss.backward()
        self.optimizer.step()
        # detached: a caller holding the loss must not keep this step's autograd graph (and its
        # AccumulateGrad nodes, which remember the stream they first ran on) alive into the
        # capture — autograd would then synchronise the capture stream with the default stream
        return loss.detach()

    def _capturable(self, images):
        if not self.enabled or self.failed is not None or not images.is_cuda:
            return False
        if not self.allow_collectives:
            from ..parallel import backend as _b

            if _b.is_initialized() and (_b.size() > 1 or _b.comm() is not None):
                return False
        return True

    def _capture(self, images, labels):
        opt = self.optimizer
        if hasattr(opt, "freeze_lr"):
            opt.freeze_lr()
        self._static_in = (images.clone(), labels.clone())
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        from ..ops import hip_ops as _hip_ops

        side_was_on = _hip_ops._SIDE["on"]
        if self.single_stream:
            _hip_ops.set_wgrad_side_stream(False)
        try:
            with torch.cuda.graph(g):
                self._static_loss = self._eager(*self._static_in)
        except Exception as e:  # capture refused something: stay eager, say why
            _hip_ops.set_wgrad_side_stream(side_was_on)
            self.failed = e
            self._static_in = self._static_loss = None
            torch.cuda.synchronize()
            warnings.warn("HIP graph capture of the training step failed (%s: %s); continuing "
                          "eagerly" % (type(e).__name__, e))
            return False
        _hip_ops.set_wgrad_side_stream(side_was_on)
        self.graph = g
        self._captured_hyper = self._hyper()
        return True

    def __call__(self, images, labels):
        """One training step; returns the (device) loss tensor.  The returned tensor is static
        once the graph is live: read or clone it before the next call."""
        self.calls += 1
        if self.graph is None:
            if self.calls <= self.warmup or not self._capturable(images):
                return self._eager(images, labels)
            if not self._capture(images, labels):
                return self._eager(images, labels)
            # capture records, it does not execute: fall through and replay for this batch
        si, sl = self._static_in
        if hasattr(self.optimizer, "sync_lr"):
            self.optimizer.sync_lr()   # the frozen learning rate lives in a device scalar from now on
        if images.shape != si.shape or labels.shape != sl.shape or images.dtype != si.dtype:
            # e.g. the short last batch of an epoch: the graph is for one shape only
            return self._eager(images, labels)
        if self._hyper() != self._captured_hyper:
            # momentum / weight decay were captured by value: a callback that moves them (the
            # one-step momentum correction of the LR schedule callbacks) gets an eager step
            return self._eager(images, labels)
        if images.data_ptr() != si.data_ptr():
            si.copy_(images, non_blocking=True)
        if labels.data_ptr() != sl.data_ptr():
            sl.copy_(labels, non_blocking=True)
        self.graph.replay()
        return self._static_loss

    @property
    def graphed(self) -> bool:
        return self.graph is not None

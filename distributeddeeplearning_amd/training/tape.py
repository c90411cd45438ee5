"""Whole-step launch tape: the training step is issued from Python ONCE, afterwards from native code.

The reference's hot loop (``HorovodPytorch/src/imagenet_pytorch_horovod.py:210-218``) runs every
device operation through the framework's Python front end, every iteration, and so does this
framework's eager path: ~600 launches per ResNet-50 step, 11 of 18 ms in host time on an idle
machine - 28 of 44 ms for ResNet-152 - and BASELINE config 5 puts eight such processes on one host
(``profiles/r2_notes.md`` section 13).  With static shapes the launch sequence never changes.
:class:`TapedTrainStep` runs a few eager steps, then records one step into the kernel library's
launch tape (``csrc/kernels/tape.h``: kernel pointer + geometry + stream + arguments by value for
every launch, the stream-ordering calls and memsets in between) and from then on replays it with
one native call per step: batch draw, zero_grad, forward, loss, backward on both streams,
optimizer update - the same kernels with the same arguments on the same streams in the same order
(``tests/test_gpu_tape.py`` compares replayed against eager training bit for bit).

Why not a hipGraph: ROCm 7.2 replays a graph with ~50 cross-stream edges at ~25 us per node (29 ms
for the 18 ms step; ``training/graph.py``), and a one-stream graph loses the weight-gradient
stream's overlap.  The tape re-issues ordinary stream launches, so it keeps both streams.

What makes a step recordable, and what guards it:

* addresses must repeat: the recorded step allocates from a private ``torch.cuda.MemPool`` that is
  kept for the life of the tape (the allocator's usual stream-ordered reuse inside the step is
  reproduced exactly, because the same launches run in the same order);
* everything the step does on the device must be an in-tree launch.  A ``TorchDispatchMode`` watches
  the recorded step: any framework operator that would run a kernel (anything but allocation and
  view bookkeeping) makes the recording invalid - the step then stays eager and says which
  operators were seen.  The root gradient is passed explicitly (``loss.backward(ones)``): autograd's
  own ``ones_like`` would be such an operator;
* per-step host values must live on the device: the learning rate (``optim.fused.freeze_lr``), the
  loader's batch counter, the BatchNorm step counters.  Momentum / weight decay are by-value kernel
  arguments: a step on which they differ from the recorded ones runs eagerly;
* hand-offs to the communication engine are *live actions* (``ops/tape.py::live``): they cut the
  tape into segments and run again, in place, at every replay.
"""
from __future__ import annotations

import traceback
import warnings

import torch
from torch.utils._python_dispatch import TorchDispatchMode

from ..ops import tape as _tape

__all__ = ["TapedTrainStep", "KernelOpWatch"]

# operators that only allocate or re-describe memory: they launch nothing
_BOOKKEEPING = {
    "empty.memory_format", "empty_strided.default", "empty_like.default", "new_empty.default",
    "new_empty_strided.default", "view.default", "_unsafe_view.default", "_reshape_alias.default",
    "as_strided.default", "slice.Tensor", "select.int", "permute.default", "transpose.int", "t.default",
    "detach.default", "alias.default", "expand.default", "unsqueeze.default", "squeeze.default",
    "squeeze.dim", "squeeze.dims", "unbind.int", "split.Tensor", "split_with_sizes.default",
    "view_as_real.default", "unfold.default", "lift_fresh.default", "is_same_size.default",
    "sym_size.int", "sym_stride.int", "sym_numel.default", "sym_storage_offset.default", "dim.default",
    "size.default", "stride.default", "numel.default", "is_contiguous.default", "_local_scalar_dense.default",
    "record_stream.default", "is_pinned.default", "prim.device.default", "prim.dtype.default", "prim.layout.default",
}


class KernelOpWatch(TorchDispatchMode):
    """Collects framework operators that touch device tensors and are not pure bookkeeping."""

    def __init__(self):
        super().__init__()
        self.seen = {}
        self.where = {}     # operator -> innermost frames of its first occurrence (who to blame)

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if name.startswith("aten."):
            name = name[5:]
        if name not in _BOOKKEEPING:
            touched = any(isinstance(a, torch.Tensor) and a.is_cuda for a in _flat(args, kwargs, out))
            if touched:
                self.seen[name] = self.seen.get(name, 0) + 1
                if name not in self.where:
                    frames = traceback.extract_stack()[:-1]
                    self.where[name] = ["%s:%d %s" % (f.filename.rsplit("/", 1)[-1], f.lineno, f.name)
                                        for f in frames[-4:]] or ["<autograd engine, no Python frame>"]
        return out


def _flat(*objs):
    for o in objs:
        if isinstance(o, (list, tuple)):
            yield from _flat(*o)
        elif isinstance(o, dict):
            yield from _flat(*o.values())
        else:
            yield o


class TapedTrainStep:
    def __init__(self, model, optimizer, loss_fn, loader=None, warmup: int = 3, enabled: bool = True,
                 allow_collectives: bool = False, strict: bool = True):
        """``loss_fn(model_output, labels) -> scalar tensor``.  ``loader`` (optional): an object whose
        ``batch()`` draws the next batch with in-tree kernels (``data.synthetic.DeviceSyntheticLoader``);
        the draw then is part of the recorded step and ``step()`` takes no arguments.  Without a loader
        the caller passes ``(images, labels)`` and they are copied into static input buffers.
        ``strict``: refuse the tape when the recorded step contained framework operators (default);
        ``allow_collectives``: record although a communicator exists (its hand-offs run as live actions)."""
        self.model, self.optimizer, self.loss_fn, self.loader = model, optimizer, loss_fn, loader
        self.warmup = max(int(warmup), 1)
        self.enabled, self.allow_collectives, self.strict = enabled, allow_collectives, strict
        self.calls = 0
        self.tape = None
        self.actions = []
        self.segments = 0
        self.launches = 0
        self.failed = None
        self.foreign_ops = None
        self.foreign_where = None
        self._pool = None
        self._ones = None
        self._static_in = None
        self._static_loss = None
        self._captured_hyper = None
        self._keep = None

    def __del__(self):
        try:
            _tape.free(self.tape)
        except Exception:  # interpreter teardown
            pass

    # ---------------------------------------------------------------------------------------
    def _hyper(self):
        return [tuple((k, v) for k, v in sorted(g.items()) if k not in ("params", "lr") and
                      isinstance(v, (int, float, bool)))
                for g in self.optimizer.param_groups]

    def _body(self, images, labels):
        self.optimizer.zero_grad()
        loss = self.loss_fn(self.model(images), labels)
        if self._ones is None or self._ones.device != loss.device or self._ones.dtype != loss.dtype:
            self._ones = torch.ones((), dtype=loss.dtype, device=loss.device)
        loss.backward(self._ones)     # explicit root gradient: no framework fill inside the step
        self.optimizer.step()
        return loss.detach()

    def _eager(self, images, labels):
        if images is None:
            images, labels = self.loader.batch()
        return self._body(images, labels)

    def _recordable(self, images):
        if not self.enabled or self.failed is not None:
            return False
        from ..ops import kernels

        if not kernels.enabled():
            return False
        dev_ok = (images is None and self.loader is not None) or (images is not None and images.is_cuda)
        if not dev_ok:
            return False
        if not self.allow_collectives:
            from ..parallel import backend as _b

            if _b.is_initialized() and (_b.size() > 1 or _b.comm() is not None):
                return False
        return True

    def _record(self, images, labels):
        opt = self.optimizer
        if hasattr(opt, "freeze_lr"):
            opt.freeze_lr()
        dev = torch.cuda.current_device()
        if images is not None:
            self._static_in = (images.clone(), labels.clone())
        torch.cuda.synchronize()
        pool = torch.cuda.MemPool()
        watch = KernelOpWatch()
        rec = _tape.Recording()
        torch._C._cuda_beginAllocateToPool(dev, pool.id)   # every thread: backward runs on autograd's
        try:
            with rec, watch:
                if self._static_in is not None:
                    loss = self._body(*self._static_in)
                    keep = None
                else:
                    batch = self.loader.batch()
                    loss = self._body(*batch)
                    keep = batch
        except Exception as e:
            self.failed = e
            self._static_in = None
            warnings.warn("recording the training step failed (%s: %s); continuing eagerly" % (type(e).__name__, e))
            raise
        finally:
            torch._C._cuda_endAllocateToPool(dev, pool.id)
            torch._C._cuda_releasePool(dev, pool.id)
        self.foreign_ops = dict(watch.seen)
        self.foreign_where = dict(watch.where)
        if watch.seen and self.strict:
            _tape.free(rec.tape)
            self.failed = _tape.TapeError("framework operators inside the step: %s; first seen at %s"
                                          % (sorted(watch.seen.items()), watch.where))
            self._static_in = None
            warnings.warn("the training step is not recordable - it ran framework operators that a replay would "
                          "skip: %s; continuing eagerly" % sorted(watch.seen.items()))
            return loss
        self.tape, self.actions, self.segments, self.launches = rec.tape, rec.actions, rec.segments, rec.calls
        self._pool = pool             # the recorded addresses live in this pool: kept as long as the tape
        self._keep = keep
        self._static_loss = loss
        self._captured_hyper = self._hyper()
        return loss

    def __call__(self, images=None, labels=None):
        """One training step; returns the (device) loss tensor - static once the tape is live: read or
        clone it before the next call."""
        self.calls += 1
        if self.tape is None:
            if self.calls <= self.warmup or not self._recordable(images):
                return self._eager(images, labels)
            return self._record(images, labels)          # the recorded step is a real step
        opt = self.optimizer
        if hasattr(opt, "sync_lr"):
            opt.sync_lr()
        if self._hyper() != self._captured_hyper:
            return self._eager(images, labels)            # by-value hyper-parameters moved (momentum correction)
        if self._static_in is not None:
            si, sl = self._static_in
            if images is None or images.shape != si.shape or labels.shape != sl.shape or images.dtype != si.dtype:
                return self._eager(images, labels)        # e.g. the short last batch of an epoch
            if images.data_ptr() != si.data_ptr():
                si.copy_(images, non_blocking=True)
            if labels.data_ptr() != sl.data_ptr():
                sl.copy_(labels, non_blocking=True)
        elif images is not None:
            return self._eager(images, labels)
        _tape.replay(self.tape, self.actions, self.segments)
        done = getattr(opt, "after_replayed_step", None)
        if done is not None:
            done()
        return self._static_loss

    @property
    def taped(self) -> bool:
        return self.tape is not None

"""Tensor collectives with Horovod's handle-based asynchronous surface.

Parity target (SURVEY §2.3 and its "not called but needed" list): ``allreduce``,
``allreduce_``, ``allreduce_async``, ``allreduce_async_``, ``allgather``, ``allgather_async``,
``broadcast``, ``broadcast_``, ``broadcast_async``, ``broadcast_async_``, ``synchronize``,
``poll``; call sites ``imagenet_keras_horovod.py:289-291`` (scalar broadcast) and ``:348-350``
(metric allreduce).

Design notes (MI355X): on the RCCL backend an asynchronous collective is enqueued on RCCL's
own HIP stream, ordered after the producer stream by an event; ``synchronize`` makes the
*current stream* wait on the collective's completion event and does not block the host, so
the Python thread keeps launching kernels ("avoid hipStreamSynchronize in
step()").  Averaging is fused into the reduction (``ncclAvg``) on RCCL; the gloo/CPU path sums
and then scales.  Collectives carry an optional name; in debug mode (``HVD_DEBUG=1``) name,
shape and dtype are cross-checked between ranks before the data moves, which is the
consistency check Horovod's coordinator did every cycle [EXT, SURVEY §2.2 N1].
"""
from __future__ import annotations

import hashlib
import itertools
import threading
import time

import torch

from . import backend as _b
from .compression import Compression
from . import timeline as _tl

__all__ = ["allreduce", "allreduce_", "allreduce_async", "allreduce_async_", "allgather",
           "allgather_async", "broadcast", "broadcast_", "broadcast_async", "broadcast_async_",
           "synchronize", "poll", "broadcast_object", "allgather_object", "Sum", "Average",
           "Min", "Max", "HandleError"]

Average = "average"
Sum = "sum"
Min = "min"
Max = "max"


class HandleError(RuntimeError):
    pass


class _Pending:
    __slots__ = ("work", "result", "finish", "name", "t0")

    def __init__(self, work, result, finish, name):
        self.work = work
        self.result = result
        self.finish = finish
        self.name = name
        self.t0 = time.monotonic()


_handles: dict[int, _Pending] = {}
_handle_ids = itertools.count(1)
_handles_lock = threading.Lock()
_auto_names = itertools.count()


def _register(work, result, finish=None, name=None) -> int:
    h = next(_handle_ids)
    with _handles_lock:
        _handles[h] = _Pending(work, result, finish, name)
    return h


def _resolve_op(average, op):
    if op is None:
        return Average if (average is None or average) else Sum
    if average is not None:
        raise ValueError("pass either average= or op=, not both")
    if op not in (Average, Sum, Min, Max):
        raise ValueError("unsupported reduction op %r" % (op,))
    return op


def _check_consistency(kind, tensor, name):
    """Debug-mode guard: all ranks must submit the same (kind, name, shape, dtype)."""
    if not _b.engine_config().debug or _b.trivial() or _b.comm().checks_consistency:
        return
    sig = "%s|%s|%s|%s" % (kind, name, tuple(tensor.shape), tensor.dtype)
    digest = int.from_bytes(hashlib.sha1(sig.encode()).digest()[:6], "big")
    probe = torch.tensor([digest, -digest], dtype=torch.int64, device=_b.device())
    _b.comm().all_reduce(probe, Max, name="consistency").wait()
    hi, neg_lo = probe.tolist()
    if hi != digest or -neg_lo != digest:
        raise HandleError("collective mismatch across ranks for %s (this rank: %s)" % (name, sig))


def _staged(tensor):
    """Tensor as the active backend can move it: gloo cannot touch device memory, so a GPU
    tensor under the gloo backend is staged through the host (multi-rank-on-one-GPU
    tests)."""
    host_only = _b.backend_name() == "gloo"
    if host_only and tensor.is_cuda:
        return tensor.detach().cpu(), True
    return tensor, False


def _launch_allreduce(buf, op, name=None):
    """Enqueue the reduction of ``buf`` in place; returns (work, needs_scale)."""
    c = _b.comm()
    if op == Average:
        if c.supports_avg and buf.is_floating_point():
            return c.all_reduce(buf, Average, name), False
        return c.all_reduce(buf, Sum, name), True
    return c.all_reduce(buf, op, name), False


def _scale_inplace(t, world):
    if t.is_floating_point():
        t.div_(world)
    else:
        t.copy_(torch.div(t, world, rounding_mode="floor"))


def allreduce_async_(tensor, average=None, name=None, op=None, compression=Compression.none):
    """In-place asynchronous allreduce; returns a handle for ``synchronize``/``poll``."""
    op = _resolve_op(average, op)
    name = name or "allreduce.noname.%d" % next(_auto_names)
    world = _b.size()
    if _b.trivial():
        return _register(None, tensor, None, name)
    _check_consistency("allreduce", tensor, name)
    _tl.begin(name, "ALLREDUCE")
    wire, ctx = compression.compress(tensor)
    buf, staged = _staged(wire)
    if not buf.is_contiguous():
        buf = buf.contiguous()
        staged = True
    work, needs_scale = _launch_allreduce(buf, op, name)

    def finish():
        if needs_scale:
            _scale_inplace(buf, world)
        out = compression.decompress(buf, ctx)
        if out.data_ptr() != tensor.data_ptr() or staged:
            tensor.copy_(out.to(tensor.device) if out.device != tensor.device else out)
        _tl.end(name)
        return tensor

    return _register(work, tensor, finish, name)


def allreduce_async(tensor, average=None, name=None, op=None, compression=Compression.none):
    return allreduce_async_(tensor.detach().clone(), average, name, op, compression)


def allreduce_(tensor, average=None, name=None, op=None, compression=Compression.none):
    return synchronize(allreduce_async_(tensor, average, name, op, compression))


def allreduce(tensor, average=None, name=None, op=None, compression=Compression.none):
    """Out-of-place allreduce; differentiable (the gradient of an average is an average)."""
    if isinstance(tensor, torch.Tensor) and tensor.requires_grad and torch.is_grad_enabled():
        return _AllreduceFn.apply(tensor, _resolve_op(average, op), name)
    if not isinstance(tensor, torch.Tensor):
        # Keras call site passes a python list of metrics (imagenet_keras_horovod.py:348-350).
        t = torch.as_tensor(tensor, dtype=torch.float64)
        out = synchronize(allreduce_async(t, average, name, op))
        return out.tolist()
    return synchronize(allreduce_async(tensor, average, name, op, compression))


class _AllreduceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tensor, op, name):
        ctx.op = op
        return synchronize(allreduce_async(tensor, None, name, op))

    @staticmethod
    def backward(ctx, grad):
        return synchronize(allreduce_async(grad.contiguous(), None, None, ctx.op)), None, None


def allgather_async(tensor, name=None):
    """Concatenate along dim 0 across ranks; first dimensions may differ between ranks."""
    name = name or "allgather.noname.%d" % next(_auto_names)
    world = _b.size()
    if _b.trivial():
        return _register(None, tensor.detach().clone(), None, name)
    if tensor.dim() == 0:
        raise ValueError("allgather needs at least one dimension")
    _check_consistency("allgather", tensor[:0], name)
    _tl.begin(name, "ALLGATHER")
    src, _ = _staged(tensor.detach().contiguous())
    dev = src.device
    first = torch.tensor([src.shape[0]], dtype=torch.int64, device=dev)
    counts_t = torch.zeros((world, 1), dtype=torch.int64, device=dev)
    _b.comm().all_gather_into(counts_t, first, name=name + ".counts").wait()
    counts = [int(c) for c in counts_t.view(-1).tolist()]
    longest = max(counts)
    padded = src
    if src.shape[0] != longest:
        padded = src.new_zeros((longest,) + tuple(src.shape[1:]))
        padded[: src.shape[0]].copy_(src)
    gathered = torch.empty((world,) + tuple(padded.shape), dtype=padded.dtype, device=dev)
    parts = list(gathered.unbind(0))
    work = _b.comm().all_gather_into(gathered, padded.contiguous(), name=name)
    holder = {}

    def finish():
        out = torch.cat([p[:n] for p, n in zip(parts, counts)], dim=0)
        holder["out"] = out.to(tensor.device)
        _tl.end(name)
        return holder["out"]

    return _register(work, None, finish, name)


def allgather(tensor, name=None):
    return synchronize(allgather_async(tensor, name))


def broadcast_async_(tensor, root_rank, name=None):
    name = name or "broadcast.noname.%d" % next(_auto_names)
    if _b.trivial():
        return _register(None, tensor, None, name)
    if not 0 <= root_rank < _b.size():
        raise ValueError("root_rank %d out of range" % root_rank)
    _check_consistency("broadcast", tensor, name)
    _tl.begin(name, "BROADCAST")
    buf, staged = _staged(tensor)
    if not buf.is_contiguous():
        buf, staged = buf.contiguous(), True
    work = _b.comm().broadcast(buf, root_rank, name)

    def finish():
        if staged:
            tensor.copy_(buf.to(tensor.device))
        _tl.end(name)
        return tensor

    return _register(work, tensor, finish, name)


def broadcast_async(tensor, root_rank, name=None):
    return broadcast_async_(tensor.detach().clone(), root_rank, name)


def broadcast_(tensor, root_rank, name=None):
    return synchronize(broadcast_async_(tensor, root_rank, name))


def broadcast(tensor, root_rank, name=None):
    """Broadcast a tensor, or any picklable python value (the Keras flavour broadcasts a
    plain int ``resume_from_epoch``, imagenet_keras_horovod.py:289-291)."""
    if not isinstance(tensor, torch.Tensor):
        return broadcast_object(tensor, root_rank, name)
    return synchronize(broadcast_async(tensor, root_rank, name))


def synchronize(handle):
    """Wait for an asynchronous collective and return its result tensor.

    On RCCL this orders the current HIP stream after the collective (no host block)."""
    with _handles_lock:
        pending = _handles.pop(handle, None)
    if pending is None:
        raise HandleError("unknown or already synchronised handle %r" % (handle,))
    if pending.work is not None:
        if not pending.work.wait(_b.engine_config().collective_timeout_s):
            raise HandleError("collective %s timed out" % (pending.name,))
    if pending.finish is not None:
        return pending.finish()
    return pending.result


def poll(handle) -> bool:
    """True once the collective behind ``handle`` has completed (non-blocking)."""
    with _handles_lock:
        pending = _handles.get(handle)
    if pending is None:
        raise HandleError("unknown or already synchronised handle %r" % (handle,))
    return True if pending.work is None else bool(pending.work.is_completed())


def broadcast_object(obj, root_rank=0, name=None):
    """Broadcast an arbitrary picklable object through the control plane."""
    if _b.trivial():
        return obj
    return _b.comm().broadcast_object(obj, root_rank, _b.rank())


def allgather_object(obj, name=None):
    if _b.trivial():
        return [obj]
    return _b.comm().all_gather_object(obj, _b.size())

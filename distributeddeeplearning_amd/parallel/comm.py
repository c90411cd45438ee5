"""Communication back-ends behind the Horovod-style API.

Two interchangeable implementations of one small interface (``all_reduce``, ``broadcast``,
``all_gather_into``, ``barrier``, object broadcast/gather; every tensor op returns a ``Work``):

* ``TorchComm``  — ``torch.distributed`` process group: RCCL over xGMI on GPUs (backend "nccl"
  IS RCCL on ROCm), gloo on CPU.  Default.
* ``NativeComm`` — the in-tree C++ engine (``csrc/engine``; ``HVD_ENGINE=native``): TCP
  rendezvous store instead of MPI/c10d, a background communication thread with a submit queue
  and integer handles, header-checked ring collectives over TCP for host tensors, RCCL called
  directly on a dedicated high-priority HIP stream with event hand-offs for device tensors,
  engine-side timeline / stall warnings / collective timeout.
"""
from __future__ import annotations

import ctypes
import datetime
import itertools
import os
import pickle

import torch
import torch.distributed as dist

from ..ops import native as _native

__all__ = ["TorchComm", "NativeComm", "Work", "CommError"]


class CommError(RuntimeError):
    pass


class Work:
    """Completion handle of one collective."""

    def wait(self, timeout_s=None) -> bool:  # pragma: no cover - interface
        raise NotImplementedError

    def is_completed(self) -> bool:  # pragma: no cover - interface
        raise NotImplementedError


class _DoneWork(Work):
    def wait(self, timeout_s=None):
        return True

    def is_completed(self):
        return True


# --------------------------------------------------------------------------------------------
class _TorchWork(Work):
    def __init__(self, work, on_gloo):
        self.work, self.on_gloo = work, on_gloo

    def wait(self, timeout_s=None):
        if self.on_gloo and timeout_s is not None:
            return bool(self.work.wait(timeout=datetime.timedelta(seconds=timeout_s)))
        self.work.wait()  # nccl: stream-ordered, the host does not block
        return True

    def is_completed(self):
        return bool(self.work.is_completed())


class TorchComm:
    name = "torch"

    def __init__(self, backend, group):
        self.backend = backend  # "rccl" | "gloo"
        self.group = group

    @property
    def supports_avg(self):
        return self.backend == "rccl"

    checks_consistency = False

    def all_reduce(self, tensor, op, name=None):
        red = {"sum": dist.ReduceOp.SUM, "average": dist.ReduceOp.AVG, "min": dist.ReduceOp.MIN,
               "max": dist.ReduceOp.MAX}[op]
        return _TorchWork(dist.all_reduce(tensor, op=red, group=self.group, async_op=True),
                          self.backend == "gloo")

    def broadcast(self, tensor, root, name=None):
        return _TorchWork(dist.broadcast(tensor, src=root, group=self.group, async_op=True),
                          self.backend == "gloo")

    def all_gather_into(self, out, tensor, name=None):
        """``out`` is [world, *tensor.shape]."""
        parts = list(out.unbind(0))
        return _TorchWork(dist.all_gather(parts, tensor, group=self.group, async_op=True),
                          self.backend == "gloo")

    def barrier(self):
        if self.backend == "rccl":
            dist.barrier(group=self.group, device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier(group=self.group)

    def broadcast_object(self, obj, root, rank):
        box = [obj if rank == root else None]
        kwargs = {}
        if self.backend == "rccl":
            kwargs["device"] = torch.device("cuda", torch.cuda.current_device())
        dist.broadcast_object_list(box, src=root, group=self.group, **kwargs)
        return box[0]

    def all_gather_object(self, obj, world):
        out = [None] * world
        dist.all_gather_object(out, obj, group=self.group)
        return out

    def shutdown(self):
        pass


# --------------------------------------------------------------------------------------------
_DTYPES = {torch.float32: 0, torch.float64: 1, torch.int64: 2, torch.bfloat16: 3, torch.int32: 4,
           torch.uint8: 5, torch.float16: 6}
_OPS = {"sum": 0, "average": 1, "min": 2, "max": 3}


class _HostWork(Work):
    def __init__(self, comm, handle, keep):
        self.comm, self.handle, self.keep = comm, handle, keep
        self.done = False

    def wait(self, timeout_s=None):
        if self.done:
            return True
        rc = self.comm.lib.ddl_eng_wait(ctypes.c_longlong(self.handle),
                                        ctypes.c_double(timeout_s or self.comm.timeout_s))
        if rc == 1:
            return False
        if rc != 0:
            raise CommError("collective failed: %s" % self.comm.last_error())
        self.done = True
        self.keep = None
        return True

    def is_completed(self):
        return self.done or self.comm.lib.ddl_eng_poll(ctypes.c_longlong(self.handle)) >= 2


class _DeviceWork(Work):
    def __init__(self, comm, handle, keep):
        self.comm, self.handle, self.keep = comm, handle, keep
        self.done = False

    def wait(self, timeout_s=None):
        if not self.done:
            stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            if self.comm.lib.ddl_rccl_wait_stream(ctypes.c_longlong(self.handle), stream) != 0:
                raise CommError("unknown RCCL handle")
            self.done = True
        return True

    def is_completed(self):
        return self.done or self.comm.lib.ddl_rccl_poll(ctypes.c_longlong(self.handle)) == 1


class NativeComm:
    name = "native"
    checks_consistency = True  # every engine message carries a verified (op, dtype, count, name) header

    def __init__(self, rank, size, local_rank, master_addr, master_port, timeout_s=300.0,
                 stall_warning_s=60.0, use_rccl=False, timeline=None):
        lib = _native.engine_lib()
        if lib is None:
            raise CommError("native engine library is not built (run `python build.py`)")
        self.lib = lib
        lib.ddl_eng_last_error.restype = ctypes.c_char_p
        lib.ddl_rccl_last_error.restype = ctypes.c_char_p
        lib.ddl_eng_store_data.restype = ctypes.c_void_p
        for fn in ("ddl_eng_allreduce_async", "ddl_eng_broadcast_async", "ddl_eng_allgather_async",
                   "ddl_eng_barrier_async", "ddl_eng_store_get", "ddl_eng_store_add",
                   "ddl_rccl_allreduce", "ddl_rccl_broadcast", "ddl_rccl_allgather"):
            getattr(lib, fn).restype = ctypes.c_longlong
        self.rank, self.size, self.timeout_s = rank, size, float(timeout_s)
        # the engine's control port sits next to the launcher's MASTER_PORT
        port = int(master_port) + 1
        rc = lib.ddl_eng_init(rank, size, master_addr.encode(), port, ctypes.c_double(timeout_s),
                              ctypes.c_double(stall_warning_s))
        if rc != 0:
            raise CommError("engine init failed: %s" % self.last_error())
        self.rccl = False
        if use_rccl:
            dev = torch.cuda.current_device()
            if lib.ddl_rccl_init(rank, size, dev, int(timeout_s * 1000)) != 0:
                raise CommError("RCCL init failed: %s" % lib.ddl_rccl_last_error().decode())
            self.rccl = True
        if timeline:
            lib.ddl_eng_timeline_start(("%s.engine.%d.json" % (timeline, rank)).encode())
        self._seq = itertools.count()

    backend = "native"

    @property
    def supports_avg(self):
        return True

    def last_error(self):
        return (self.lib.ddl_eng_last_error() or b"").decode()

    def _check(self, t):
        if t.dtype not in _DTYPES:
            raise CommError("dtype %s is not supported by the native engine" % t.dtype)
        if not t.is_contiguous():
            raise CommError("native engine needs contiguous tensors")

    def _name(self, name):
        return (name or "noname").encode()

    def all_reduce(self, tensor, op, name=None):
        self._check(tensor)
        if self.size == 1:
            return _DoneWork()
        n, dt, p = tensor.numel(), _DTYPES[tensor.dtype], ctypes.c_void_p(tensor.data_ptr())
        if tensor.is_cuda:
            if not self.rccl:
                raise CommError("device tensor given but the engine was started without RCCL")
            s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            h = self.lib.ddl_rccl_allreduce(p, ctypes.c_longlong(n), dt, _OPS[op], s, self._name(name))
            if h < 0:
                raise CommError(self.lib.ddl_rccl_last_error().decode())
            return _DeviceWork(self, h, tensor)
        h = self.lib.ddl_eng_allreduce_async(p, ctypes.c_longlong(n), dt, _OPS[op], self._name(name))
        return _HostWork(self, h, tensor)

    def broadcast(self, tensor, root, name=None):
        self._check(tensor)
        if self.size == 1:
            return _DoneWork()
        n, dt, p = tensor.numel(), _DTYPES[tensor.dtype], ctypes.c_void_p(tensor.data_ptr())
        if tensor.is_cuda:
            if not self.rccl:
                raise CommError("device tensor given but the engine was started without RCCL")
            s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            h = self.lib.ddl_rccl_broadcast(p, ctypes.c_longlong(n), dt, root, s, self._name(name))
            if h < 0:
                raise CommError(self.lib.ddl_rccl_last_error().decode())
            return _DeviceWork(self, h, tensor)
        h = self.lib.ddl_eng_broadcast_async(p, ctypes.c_longlong(n), dt, root, self._name(name))
        return _HostWork(self, h, tensor)

    def all_gather_into(self, out, tensor, name=None):
        self._check(tensor)
        self._check(out)
        n, dt = tensor.numel(), _DTYPES[tensor.dtype]
        pi, po = ctypes.c_void_p(tensor.data_ptr()), ctypes.c_void_p(out.data_ptr())
        if tensor.is_cuda:
            if not self.rccl:
                raise CommError("device tensor given but the engine was started without RCCL")
            s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            h = self.lib.ddl_rccl_allgather(pi, po, ctypes.c_longlong(n), dt, s, self._name(name))
            if h < 0:
                raise CommError(self.lib.ddl_rccl_last_error().decode())
            return _DeviceWork(self, h, (tensor, out))
        h = self.lib.ddl_eng_allgather_async(pi, po, ctypes.c_longlong(n), dt, self._name(name))
        return _HostWork(self, h, (tensor, out))

    def barrier(self):
        if self.size == 1:
            return
        h = self.lib.ddl_eng_barrier_async(b"barrier")
        if not _HostWork(self, h, None).wait(self.timeout_s):
            raise CommError("barrier timed out")
        if self.rccl:
            torch.cuda.current_stream().synchronize()

    # -- objects through the key-value store --------------------------------------------------
    def _put(self, key, obj):
        blob = pickle.dumps(obj)
        if self.lib.ddl_eng_store_set(key.encode(), blob, ctypes.c_longlong(len(blob))) != 0:
            raise CommError("store set failed")

    def _get(self, key):
        n = self.lib.ddl_eng_store_get(key.encode(), int(self.timeout_s * 1000))
        if n < 0:
            raise CommError("store get(%s) timed out" % key)
        return pickle.loads(ctypes.string_at(self.lib.ddl_eng_store_data(), n))

    def broadcast_object(self, obj, root, rank):
        if self.size == 1:
            return obj
        key = "obj/%d" % next(self._seq)
        if rank == root:
            self._put(key, obj)
            return obj
        return self._get(key)

    def all_gather_object(self, obj, world):
        if self.size == 1:
            return [obj]
        seq = next(self._seq)
        self._put("gobj/%d/%d" % (seq, self.rank), obj)
        return [self._get("gobj/%d/%d" % (seq, r)) for r in range(world)]

    def shutdown(self):
        if self.rccl:
            self.lib.ddl_rccl_shutdown()
        self.lib.ddl_eng_shutdown()

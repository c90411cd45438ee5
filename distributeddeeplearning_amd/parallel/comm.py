"""Communication back-ends behind the Horovod-style API.

Two interchangeable implementations of one small interface (``all_reduce``, ``broadcast``,
``all_gather_into``, ``barrier``, object broadcast/gather; every tensor op returns a ``Work``):

* ``TorchComm``  — ``torch.distributed`` process group: gloo on CPU (the default there), RCCL
  through c10d on GPUs only when ``HVD_ENGINE=torch`` asks for it.
* ``NativeComm`` — the in-tree C++ engine (``csrc/engine``), the DEFAULT data plane on GPUs: TCP
  rendezvous store instead of MPI/c10d, a background communication thread with a submit queue
  and integer handles, header-checked ring collectives over TCP for host tensors, RCCL called
  directly on a dedicated high-priority HIP stream with event hand-offs for device tensors,
  engine-side timeline / stall warnings / collective timeout.
"""
from __future__ import annotations

import ctypes
import datetime
import hashlib
import hmac
import itertools
import os
import io
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
_ALGOS = {"rccl": 0, "twoshot": 1, "oneshot": 2}
# HVD_ALLREDUCE_ALGO=auto picks the all-reduce by bucket size from a MEASURED table (written by
# ``benchmarks/allreduce_bw.py --algo all --table <path>`` on the node the job runs on; path in
# HVD_ALLREDUCE_TABLE, default profiles/allreduce_table_n<world>.json).  There are no built-in
# thresholds: without the table ``auto`` refuses to start and says how to produce it.


def default_table_path(world):
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    return os.environ.get("HVD_ALLREDUCE_TABLE") or os.path.join(root, "profiles", "allreduce_table_n%d.json" % world)


def load_algo_table(world, path=None):
    """[(max_bytes, algo), ...] ascending, the last entry open-ended; raises CommError when the table
    is missing, was measured for another world size, or names an unknown algorithm."""
    import json

    path = path or default_table_path(world)
    if not os.path.isfile(path):
        raise CommError(
            "HVD_ALLREDUCE_ALGO=auto needs the measured table %s (not found).  Produce it on this node with\n"
            "  python -m torch.distributed.run --nproc-per-node %d benchmarks/allreduce_bw.py --algo all --table %s\n"
            "or choose HVD_ALLREDUCE_ALGO=rccl|twoshot|oneshot explicitly." % (path, world, path))
    with open(path) as fh:
        doc = json.load(fh)
    if int(doc.get("world", -1)) != int(world):
        raise CommError("all-reduce table %s was measured with %s ranks, this job has %d" % (path, doc.get("world"), world))
    rows = []
    for ent in doc.get("thresholds", []):
        algo, mx = ent.get("algo"), ent.get("max_bytes")
        if algo not in _ALGOS:
            raise CommError("all-reduce table %s names unknown algorithm %r" % (path, algo))
        rows.append((float("inf") if mx is None else int(mx), algo))
    rows.sort(key=lambda r: r[0])
    if not rows or rows[-1][0] != float("inf"):
        raise CommError("all-reduce table %s has no open-ended last entry" % path)
    return rows


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


class _PlainUnpickler(pickle.Unpickler):
    """Unpickler for control-plane objects that arrive WITHOUT the job token's authentication: builds
    containers, numbers, strings, bytes and torch dtypes / sizes only, so a blob written by a foreign
    process cannot name a callable (what ``hvd.broadcast_object`` / ``allgather_object`` carry in
    this framework - resume epochs, metric lists, the optimizer-state skeleton - is plain data)."""

    _BUILTINS = {"dict", "list", "tuple", "set", "frozenset", "str", "int", "float", "bool", "bytes",
                 "bytearray", "complex", "slice", "range"}

    def find_class(self, module, name):
        if module == "builtins" and name in self._BUILTINS:
            return getattr(__import__("builtins"), name)
        if module == "collections" and name == "OrderedDict":
            import collections

            return collections.OrderedDict
        if module == "torch" and (name == "Size" or isinstance(getattr(torch, name, None), torch.dtype)):
            return getattr(torch, name)
        raise CommError("control-plane object refused: it references %s.%s and the job has no DDL_JOB_TOKEN "
                        "to authenticate it (launch with ddlrun, export DDL_JOB_TOKEN on every rank, or "
                        "set DDL_TRUST_CONTROL_PLANE=1)" % (module, name))


class _DeviceWork(Work):
    """A device collective launched by the engine's communication thread.  ``wait`` orders the
    current HIP stream after it (the host blocks only until the comm thread has issued the
    launch, microseconds); a collective that the watchdog declared dead (``HVD_COLLECTIVE_
    TIMEOUT_S``) raises here."""

    def __init__(self, comm, handle, keep):
        self.comm, self.handle, self.keep = comm, handle, keep
        self.done = False

    def wait(self, timeout_s=None):
        if not self.done:
            stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            rc = self.comm.lib.ddl_eng_device_wait(ctypes.c_longlong(self.handle), stream,
                                                   ctypes.c_double(timeout_s or self.comm.timeout_s))
            if rc == 1:
                return False
            if rc != 0:
                raise CommError("device collective failed: %s" % self.comm.last_error())
            self.done = True
        return True

    def wait_host(self, timeout_s=None):
        """Block the HOST until the collective has completed on the device."""
        if not self.done:
            rc = self.comm.lib.ddl_eng_device_wait_host(ctypes.c_longlong(self.handle),
                                                        ctypes.c_double(timeout_s or self.comm.timeout_s))
            if rc == 1:
                return False
            if rc != 0:
                raise CommError("device collective failed: %s" % self.comm.last_error())
            self.done = True
        return True

    def is_completed(self):
        return self.done or self.comm.lib.ddl_eng_poll(ctypes.c_longlong(self.handle)) >= 2


def engine_port(rank, master_addr, master_port, timeout_s=300.0, lib=None):
    """Control port of the native engine's TCP store.

    ``DDL_ENGINE_PORT`` wins.  Under ``torch.distributed.run`` the agent already hosts a store on
    MASTER_PORT: rank 0 lets the ENGINE bind any free port (``ddl_eng_listen(0)``: the socket that
    is probed is the socket that serves, nobody can take the port in between) and publishes the
    port it got there, under a key that carries the run id AND the restart count - after an
    elastic worker restart the other ranks must not read the previous attempt's (dead) port.
    Otherwise - our own launcher, ``mpirun`` - the engine listens on MASTER_PORT + 1."""
    explicit = os.environ.get("DDL_ENGINE_PORT")
    if explicit:
        return int(explicit)
    if os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "").lower() == "true":
        try:
            store = dist.TCPStore(master_addr, int(master_port), is_master=False,
                                  timeout=datetime.timedelta(seconds=timeout_s))
            key = "ddl/engine_port/%s/%s" % (os.environ.get("TORCHELASTIC_RUN_ID", "0"),
                                             os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"))
            if rank == 0:
                loopback = master_addr in ("127.0.0.1", "localhost")
                port = -1
                if lib is not None and hasattr(lib, "ddl_eng_listen"):
                    port = int(lib.ddl_eng_listen(0, 1 if loopback else 0))
                if port <= 0:   # older engine library: probe (racy, kept as a last resort)
                    import socket

                    with socket.socket() as sk:
                        sk.bind(("127.0.0.1" if loopback else "", 0))
                        port = sk.getsockname()[1]
                store.set(key, str(port))
                return port
            return int(store.get(key).decode())
        except Exception:  # agent store unreachable: fall back to the fixed rule
            pass
    return int(master_port) + 1


class NativeComm:
    name = "native"
    checks_consistency = True  # every engine message carries a verified (op, dtype, count, name) header

    def __init__(self, rank, size, local_rank, master_addr, master_port, timeout_s=300.0,
                 stall_warning_s=60.0, use_rccl=False, timeline=None, allreduce_algo="rccl"):
        lib = _native.engine_lib()
        if lib is None:
            raise CommError("native engine library is not built (run `python build.py`)")
        self.lib = lib
        lib.ddl_eng_last_error.restype = ctypes.c_char_p
        lib.ddl_rccl_last_error.restype = ctypes.c_char_p
        lib.ddl_eng_store_data.restype = ctypes.c_void_p
        lib.ddl_rccl_stream.restype = ctypes.c_void_p
        for fn in ("ddl_eng_allreduce_async", "ddl_eng_broadcast_async", "ddl_eng_allgather_async",
                   "ddl_eng_barrier_async", "ddl_eng_store_get", "ddl_eng_store_add",
                   "ddl_eng_device_collective", "ddl_eng_stats", "ddl_rccl_launch_count",
                   "ddl_xgmi_launch_count"):
            getattr(lib, fn).restype = ctypes.c_longlong
        lib.ddl_eng_device_collective.argtypes = [
            ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int,
            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_char_p]
        lib.ddl_eng_device_wait.argtypes = [ctypes.c_longlong, ctypes.c_void_p, ctypes.c_double]
        lib.ddl_eng_device_wait_host.argtypes = [ctypes.c_longlong, ctypes.c_double]
        self.rank, self.size, self.timeout_s = rank, size, float(timeout_s)
        port = engine_port(rank, master_addr, master_port, timeout_s, lib)
        rc = lib.ddl_eng_init(rank, size, master_addr.encode(), port, ctypes.c_double(timeout_s),
                              ctypes.c_double(stall_warning_s))
        if rc != 0:
            raise CommError("engine init failed: %s" % self.last_error())
        self.rccl = False
        self.has_rccl_comm = False
        self.algo = "rccl"
        self.force_algo = None     # "rccl" | "twoshot" | "oneshot": overrides the policy (benchmarks, self-test)
        self.algo_table = None     # HVD_ALLREDUCE_ALGO=auto: [(max_bytes, algo)] from the measured table
        self._regions = []
        # broadcast_object blobs are authenticated with a per-job secret when the launcher
        # provides one (DDL_JOB_TOKEN): nothing unauthenticated is ever unpickled
        self._token = (os.environ.get("DDL_JOB_TOKEN") or "").encode()
        if use_rccl:
            dev = torch.cuda.current_device()
            if lib.ddl_rccl_init(rank, size, dev, int(timeout_s * 1000)) != 0:
                raise CommError("RCCL init failed: %s" % lib.ddl_rccl_last_error().decode())
            self.rccl = True
            # DDL_RCCL_SKIP_COMM=1 (several ranks on ONE GPU, which RCCL refuses): the device plane
            # has no communicator; whatever the xGMI kernel cannot serve is staged through the host
            self.has_rccl_comm = bool(lib.ddl_rccl_has_comm())
            self._init_algo(allreduce_algo)
        if timeline:
            lib.ddl_eng_timeline_start(("%s.engine.%d.json" % (timeline, rank)).encode())
        self._seq = itertools.count()

    backend = "native"

    def _init_algo(self, algo):
        algo = (algo or "rccl").lower()
        if algo not in ("rccl", "twoshot", "oneshot", "auto", "manual"):
            raise CommError("unknown HVD_ALLREDUCE_ALGO %r (expected rccl|twoshot|oneshot|auto|manual)" % algo)
        if algo == "rccl" or self.size == 1:
            return
        if algo == "auto":
            self.algo_table = load_algo_table(self.size)   # refuses (CommError) until measured
        if self.size > 8:
            raise CommError("HVD_ALLREDUCE_ALGO=%s spans one node of at most 8 GPUs" % algo)
        from ..ops import kernels as _k

        fn = ctypes.cast(_k.lib().ddl_xgmi_allreduce, ctypes.c_void_p)
        blocks = int(os.environ.get("HVD_XGMI_BLOCKS", "16"))
        # Spin budget of the kernel's flag waits.  One poll (sc0 sc1 load + s_sleep 8 + loop) takes
        # ~0.5-1 us, so the 20 M-poll default gave up after a few tens of seconds - far below the
        # collective timeout every other path honours (a rank that checkpoints, validates or stalls in
        # its loader is legitimately late).  The budget now follows HVD_COLLECTIVE_TIMEOUT_S at a
        # conservative 0.4 us per poll (the kernel can only be MORE patient than the watchdog, which
        # then is the one that declares the collective dead); HVD_XGMI_SPIN_LIMIT overrides.
        spins = int(os.environ.get("HVD_XGMI_SPIN_LIMIT", "0")) or max(int(self.timeout_s * 2.5e6), 20000000)
        if self.lib.ddl_xgmi_init(fn, blocks, ctypes.c_longlong(spins)) != 0:
            raise CommError("xGMI plane init failed: %s" % self.lib.ddl_rccl_last_error().decode())
        self.algo = algo

    def register_region(self, tensor):
        """Collective: make ``tensor``'s memory (a gradient arena) reachable by the direct xGMI
        all-reduce on every rank.  No-op unless HVD_ALLREDUCE_ALGO selects that kernel."""
        if self.algo == "rccl" or not tensor.is_cuda:
            return -1
        nbytes = tensor.numel() * tensor.element_size()
        rid = self.lib.ddl_xgmi_register(ctypes.c_void_p(tensor.data_ptr()), ctypes.c_longlong(nbytes))
        if rid < 0:
            raise CommError("xGMI region registration failed: %s" % self.lib.ddl_rccl_last_error().decode())
        self._regions.append(tensor)  # keep the storage alive as long as peers map it
        return rid

    def _pick_algo(self, tensor, op):
        if self.algo == "rccl" or op not in ("sum", "average") or \
                tensor.dtype not in (torch.float32, torch.bfloat16):
            return 0
        if self.force_algo is not None:
            return _ALGOS[self.force_algo]
        if self.algo == "manual":
            return 0
        if self.algo != "auto":
            return _ALGOS[self.algo]
        nbytes = tensor.numel() * tensor.element_size()
        for max_bytes, algo in self.algo_table:
            if nbytes <= max_bytes:
                return _ALGOS[algo]
        return 0

    def enabled_algos(self):
        """Algorithms a gradient bucket can take in this job (the init self-test covers each)."""
        if self.algo == "rccl" or self.size == 1:
            return ["rccl"]
        if self.algo in ("auto", "manual"):
            return ["rccl", "twoshot", "oneshot"] if self.has_rccl_comm else ["twoshot", "oneshot"]
        return [self.algo]

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

    def _device(self, kind, tensor, out, n, dt, redop, root, algo, name, keep):
        if not self.rccl:
            raise CommError("device tensor given but the engine was started without RCCL")
        if not self.has_rccl_comm and not (kind == 0 and algo != 0 and self._registered(tensor)):
            return self._staged(kind, tensor, out, n, dt, redop, root, name)
        h = self.lib.ddl_eng_device_collective(kind, tensor.data_ptr(), None if out is None else out.data_ptr(),
                                               n, dt, redop, root, algo,
                                               torch.cuda.current_stream().cuda_stream, self._name(name))
        if h < 0:
            raise CommError(self.last_error())
        return _DeviceWork(self, h, keep)

    def _registered(self, t):
        a, b = t.data_ptr(), t.data_ptr() + t.numel() * t.element_size()
        return any(r.data_ptr() <= a and b <= r.data_ptr() + r.numel() * r.element_size()
                   for r in self._regions) and t.numel() % (8 * self.size) == 0

    def _staged(self, kind, tensor, out, n, dt, redop, root, name):
        """Device tensor through the host TCP ring (communicator-less test mode only)."""
        host = tensor.detach().cpu()
        if kind == 0:
            h = self.lib.ddl_eng_allreduce_async(ctypes.c_void_p(host.data_ptr()), ctypes.c_longlong(n), dt,
                                                 redop, self._name(name))
        elif kind == 1:
            h = self.lib.ddl_eng_broadcast_async(ctypes.c_void_p(host.data_ptr()), ctypes.c_longlong(n), dt,
                                                 root, self._name(name))
        else:
            hout = torch.empty(out.shape, dtype=out.dtype)
            h = self.lib.ddl_eng_allgather_async(ctypes.c_void_p(host.data_ptr()), ctypes.c_void_p(hout.data_ptr()),
                                                 ctypes.c_longlong(n), dt, self._name(name))
        if not _HostWork(self, h, host).wait(self.timeout_s):
            raise CommError("staged collective timed out")
        if kind == 2:
            out.copy_(hout)
        else:
            tensor.copy_(host)
        return _DoneWork()

    def all_reduce(self, tensor, op, name=None):
        self._check(tensor)
        n, dt = tensor.numel(), _DTYPES[tensor.dtype]
        if tensor.is_cuda:
            return self._device(0, tensor, None, n, dt, _OPS[op], 0, self._pick_algo(tensor, op), name, tensor)
        h = self.lib.ddl_eng_allreduce_async(ctypes.c_void_p(tensor.data_ptr()), ctypes.c_longlong(n), dt,
                                             _OPS[op], self._name(name))
        return _HostWork(self, h, tensor)

    def broadcast(self, tensor, root, name=None):
        self._check(tensor)
        n, dt = tensor.numel(), _DTYPES[tensor.dtype]
        if tensor.is_cuda:
            return self._device(1, tensor, None, n, dt, 0, root, 0, name, tensor)
        h = self.lib.ddl_eng_broadcast_async(ctypes.c_void_p(tensor.data_ptr()), ctypes.c_longlong(n), dt,
                                             root, self._name(name))
        return _HostWork(self, h, tensor)

    def all_gather_into(self, out, tensor, name=None):
        self._check(tensor)
        self._check(out)
        n, dt = tensor.numel(), _DTYPES[tensor.dtype]
        if tensor.is_cuda:
            return self._device(2, tensor, out, n, dt, 0, 0, 0, name, (tensor, out))
        h = self.lib.ddl_eng_allgather_async(ctypes.c_void_p(tensor.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                             ctypes.c_longlong(n), dt, self._name(name))
        return _HostWork(self, h, (tensor, out))

    def barrier(self):
        h = self.lib.ddl_eng_barrier_async(b"barrier")
        if not _HostWork(self, h, None).wait(self.timeout_s):
            raise CommError("barrier timed out")
        if self.rccl:
            torch.cuda.current_stream().synchronize()

    def failed(self):
        return bool(self.lib.ddl_eng_failed())

    def stats(self):
        """{collective name: {"count", "last_ms", "mean_ms", "algo", "bytes"}}: DEVICE time of every
        named device collective, measured by the engine from HIP events."""
        cap = 1 << 16
        buf = ctypes.create_string_buffer(cap)
        n = self.lib.ddl_eng_stats(buf, ctypes.c_longlong(cap))
        if n >= cap:   # the engine reports the required length: grow once and ask again
            cap = int(n) + 4096
            buf = ctypes.create_string_buffer(cap)
            n = self.lib.ddl_eng_stats(buf, ctypes.c_longlong(cap))
        out = {}
        if n <= 0:
            return out
        algos = {v: k for k, v in _ALGOS.items()}
        for line in buf.value.decode().splitlines():
            parts = line.rsplit(" ", 5)
            if len(parts) == 6:
                out[parts[0]] = {"count": int(parts[1]), "last_ms": float(parts[2]), "mean_ms": float(parts[3]),
                                 "algo": algos.get(int(parts[4]), "?"), "bytes": int(parts[5])}
        return out

    def launch_counts(self):
        return {"rccl": int(self.lib.ddl_rccl_launch_count()), "xgmi": int(self.lib.ddl_xgmi_launch_count())}

    # -- objects through the key-value store --------------------------------------------------
    def _seal(self, blob):
        if not self._token:
            return b"\x00" + blob
        return b"\x01" + hmac.new(self._token, blob, hashlib.sha256).digest() + blob

    def _unseal(self, data):
        if data[:1] == b"\x01":
            mac, blob = data[1:33], data[33:]
            if not self._token or not hmac.compare_digest(mac, hmac.new(self._token, blob, hashlib.sha256).digest()):
                raise CommError("control-plane object failed authentication (DDL_JOB_TOKEN mismatch)")
            return blob
        if self._token:
            raise CommError("unauthenticated control-plane object refused (this job runs with DDL_JOB_TOKEN)")
        return data[1:]

    def _put(self, key, obj):
        blob = self._seal(pickle.dumps(obj))
        if self.lib.ddl_eng_store_set(key.encode(), blob, ctypes.c_longlong(len(blob))) != 0:
            raise CommError("store set failed")

    def _get(self, key):
        n = self.lib.ddl_eng_store_get(key.encode(), int(self.timeout_s * 1000))
        if n < 0:
            raise CommError("store get(%s) timed out" % key)
        data = ctypes.string_at(self.lib.ddl_eng_store_data(), n)
        blob = self._unseal(data)
        if data[:1] == b"\x01" or os.environ.get("DDL_TRUST_CONTROL_PLANE") == "1":
            return pickle.loads(blob)          # authenticated with the job token (or explicitly trusted)
        # no job token (e.g. launched by torchrun): only plain data may come out of the store
        return _PlainUnpickler(io.BytesIO(blob)).load()

    def _ack(self, key, readers, keys):
        """The last of ``readers`` readers deletes ``keys`` (the store does not grow with the job)."""
        if readers <= 0:
            return
        if self.lib.ddl_eng_store_add((key + "/acks").encode(), ctypes.c_longlong(1)) == readers:
            for k in list(keys) + [key + "/acks"]:
                self.lib.ddl_eng_store_del(k.encode())

    def broadcast_object(self, obj, root, rank):
        if self.size == 1:
            return obj
        key = "obj/%d" % next(self._seq)
        if rank == root:
            self._put(key, obj)
            return obj
        out = self._get(key)
        self._ack(key, self.size - 1, [key])
        return out

    def all_gather_object(self, obj, world):
        if self.size == 1:
            return [obj]
        seq = next(self._seq)
        keys = ["gobj/%d/%d" % (seq, r) for r in range(world)]
        self._put(keys[self.rank], obj)
        out = [self._get(k) for k in keys]
        self._ack("gobj/%d" % seq, world, keys)
        return out

    def shutdown(self):
        if self.rccl:
            self.lib.ddl_rccl_shutdown()
        self.lib.ddl_eng_shutdown()
        self._regions = []

"""Process-group bootstrap: the ``hvd.init/rank/size/local_rank`` half of the Horovod contract.

Parity target: ``hvd.init()``, ``hvd.rank()``, ``hvd.local_rank()``, ``hvd.size()`` as used at
``HorovodPytorch/src/imagenet_pytorch_horovod.py:62,196,250,261,272,276,331``.

MI355X design: one process per GPU, rendezvous from the environment (``RANK``/``WORLD_SIZE``/
``LOCAL_RANK``/``MASTER_ADDR``/``MASTER_PORT`` as exported by ``torch.distributed.run`` or by
our own launcher; OpenMPI's ``OMPI_COMM_WORLD_*`` names are honoured too so an ``mpirun`` line
from the reference notebooks still works).  On GPUs the data plane is the in-tree C++ engine
(``csrc/engine``: RCCL over xGMI called directly from its own communication thread, optionally
the direct xGMI kernel); ``HVD_ENGINE=torch`` selects ``torch.distributed``'s process group
instead (backend ``"nccl"`` IS RCCL on ROCm).  The CPU/gloo backend serves the GPU-less plumbing
configuration (BASELINE.json config 1) and the tests.  No MPI anywhere.

Every communicator proves itself before ``init()`` returns (``parallel/selftest.py``): a checked
all-reduce per enabled algorithm and wire dtype, a broadcast and an all-gather, under a timeout.
A failure ends the job with the engine's diagnosis unless ``HVD_ENGINE_FALLBACK=torch`` lets all
ranks agree on the c10d process group - loudly (stderr, ``engine_name()``).
"""
from __future__ import annotations

import datetime
import os
import threading

import torch
import torch.distributed as dist

from ..utils.config import EngineConfig

__all__ = ["init", "shutdown", "is_initialized", "rank", "size", "local_rank", "local_size",
           "cross_rank", "cross_size", "backend_name", "device", "group", "barrier",
           "engine_config", "NotInitializedError", "supports_avg", "comm", "engine_name",
           "engine_fallback_reason", "selftest_report"]


class NotInitializedError(ValueError):
    """Raised by rank()/size() before init(); mirrors Horovod raising ValueError there, which
    the reference relies on in ``_get_rank`` (:59-66)."""


class _State:
    def __init__(self):
        self.initialized = False
        self.rank = 0
        self.size = 1
        self.local_rank = 0
        self.local_size = 1
        self.backend = "single"
        self.group = None
        self.comm = None
        self.engine = "torch"
        self.engine_fallback = None   # why the native engine was abandoned (HVD_ENGINE_FALLBACK=torch)
        self.selftest = None          # what the init self-test covered
        self.owns_default_group = False
        self.cpu_affinity = None  # CPU set chosen by utils.affinity (None = left alone)
        self.config = EngineConfig()
        self.lock = threading.Lock()


_S = _State()


def _env_first(*names, default=None):
    for n in names:
        v = os.environ.get(n)
        if v not in (None, ""):
            return v
    return default


def _comm_device(want):
    return torch.device("cuda", torch.cuda.current_device()) if want == "rccl" else torch.device("cpu")


def _start_native(rk, world, want, cfg, tmo):
    """Bring the in-tree engine up and prove it.  Returns None on success, else the exception -
    after every rank has learned through the engine's TCP store that SOME rank failed, so that a
    fallback (or the exit) is the same decision everywhere."""
    from .comm import CommError, NativeComm

    problem = None
    comm = None
    try:
        comm = NativeComm(rk, world, _S.local_rank,
                          os.environ.get("MASTER_ADDR", "127.0.0.1"),
                          int(os.environ.get("MASTER_PORT", "29500")),
                          tmo, cfg.stall_warning_s,
                          use_rccl=(want == "rccl"), timeline=cfg.timeline,
                          allreduce_algo=cfg.allreduce_algo)
        if cfg.init_selftest:
            from .selftest import run_init_selftest

            _S.selftest = run_init_selftest(comm, rk, world, _comm_device(want))
    except CommError as e:
        problem = e
    # agreement: a rank whose own test passed must not run ahead of one that is about to leave
    if comm is not None and world > 1 and getattr(comm, "lib", None) is not None:
        try:
            verdicts = comm.all_gather_object("ok" if problem is None else "failed: %s" % problem, world)
            if problem is None:
                bad = [(r, v) for r, v in enumerate(verdicts) if v != "ok"]
                if bad:
                    problem = CommError("rank %d reported: %s" % bad[0])
        except CommError as e:  # the control plane itself is gone
            problem = problem or e
    if problem is None:
        _S.comm = comm
        _S.backend = want
        return None
    if comm is not None:
        try:
            comm.shutdown()
        except Exception:  # pragma: no cover - teardown best effort
            pass
    return problem


def _start_torch(rk, world, want, cfg, tmo_s):
    if dist.is_initialized():
        _S.group = dist.group.WORLD
        _S.owns_default_group = False
    else:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ["RANK"] = str(rk)
        os.environ["WORLD_SIZE"] = str(world)
        tmo = datetime.timedelta(seconds=tmo_s)
        kwargs = {}
        if want == "rccl":
            kwargs["device_id"] = torch.device("cuda", torch.cuda.current_device())
            # Collectives on a HIGH-priority stream: besides pre-empting compute at
            # workgroup dispatch, a priority class of its own gives the stream a hardware
            # queue of its own.  A normal-priority pool stream shared queue 4 with the
            # compute stream (rocprofv3 trace, profiles/r1_notes.md): the collective's
            # wait for the weight-gradient stream then sat in front of the compute
            # stream's packets and stalled it 100-150 us per bucket.
            if cfg.high_priority_comm and hasattr(dist, "ProcessGroupNCCL"):
                try:
                    kwargs["pg_options"] = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
                except Exception:  # pragma: no cover - older torch
                    pass
        dist.init_process_group(backend="nccl" if want == "rccl" else "gloo",
                                init_method="env://", rank=rk, world_size=world,
                                timeout=tmo, **kwargs)
        _S.group = dist.group.WORLD
        _S.owns_default_group = True
    _S.backend = want
    from .comm import TorchComm

    _S.comm = TorchComm(want, _S.group)


def init(backend: str | None = None, timeout_s: float | None = None) -> None:
    """Idempotent global initialisation.

    With ``WORLD_SIZE`` unset (or 1) this is a single-process job: no process group is created
    and every collective degenerates to a local no-op, which is the reference's
    ``DISTRIBUTED=False`` behaviour with the same code path.
    """
    with _S.lock:
        if _S.initialized:
            return
        cfg = EngineConfig.from_env()
        _S.config = cfg
        world = int(_env_first("WORLD_SIZE", "OMPI_COMM_WORLD_SIZE", "PMI_SIZE", default="1"))
        rk = int(_env_first("RANK", "OMPI_COMM_WORLD_RANK", "PMI_RANK", default="0"))
        lrk = _env_first("LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_RANK", "MPI_LOCALRANKID")
        lsz = _env_first("LOCAL_WORLD_SIZE", "OMPI_COMM_WORLD_LOCAL_SIZE", "MPI_LOCALNRANKS")
        want = (backend or cfg.backend or "auto").lower()
        n_dev = torch.cuda.device_count()
        if want in ("auto", ""):
            want = "rccl" if n_dev > 0 else "gloo"
        if want in ("nccl", "rccl"):
            want = "rccl"
            if n_dev == 0:
                raise RuntimeError("HVD_BACKEND=rccl requested but no GPU is visible")
        elif want != "gloo":
            raise ValueError("unknown backend %r (expected rccl|gloo|auto)" % want)

        if lrk is None:
            lrk = rk % n_dev if n_dev > 0 else rk
        _S.local_rank = int(lrk)
        _S.local_size = int(lsz) if lsz is not None else (min(world, n_dev) if n_dev else world)
        _S.rank, _S.size = rk, world
        if world > 1:
            # events between the two streams of a device keep their system-scope fences when several ranks run
            # (csrc/kernels/elementwise.hip::ddl_event_create reads WORLD_SIZE itself; mpirun-style launchers export other names)
            os.environ.setdefault("DDL_EVENT_FLAGS", "0")

        if want == "rccl":
            # Pin the device before the communicator exists: RCCL binds to the current device.
            torch.cuda.set_device(_S.local_rank % n_dev)
            if world > 1:  # host threads next to the GPU they feed (best effort, HVD_NUMA_BIND=0 disables)
                from ..utils.affinity import bind_to_gpu_numa

                _S.cpu_affinity = bind_to_gpu_numa(_S.local_rank % n_dev)

        engine = (cfg.engine or "auto").lower()
        if engine not in ("auto", "torch", "native"):
            raise ValueError("unknown HVD_ENGINE %r (expected auto|torch|native)" % engine)
        if engine == "auto":
            # GPUs: the in-tree engine (own comm thread, RCCL called directly, watchdog) IS the data
            # plane; the c10d process group stays available as HVD_ENGINE=torch.  CPU/gloo jobs keep
            # torch.distributed's gloo (the native engine's TCP ring is opt-in there).
            from ..ops import native as _native

            engine = "native" if (want == "rccl" and _native.engine_lib() is not None) else "torch"
        _S.engine = engine
        if cfg.cycle_time_ms:
            import warnings

            warnings.warn("HVD_CYCLE_TIME / HOROVOD_CYCLE_TIME=%g ms has no effect: gradient buckets are "
                          "planned statically and launched the moment their last gradient arrives "
                          "(there is no negotiation cycle to pace)" % cfg.cycle_time_ms)
        force = bool(cfg.force_collectives)
        tmo = timeout_s or cfg.collective_timeout_s
        if (world > 1 or force) and engine == "native":
            problem = _start_native(rk, world, want, cfg, tmo)
            if problem is not None:
                if cfg.engine_fallback != "torch":
                    raise problem
                import sys

                print("[ddl rank %d] native engine abandoned, continuing on torch.distributed (HVD_ENGINE_FALLBACK=torch): %s"
                      % (rk, problem), file=sys.stderr, flush=True)
                _S.engine_fallback = str(problem).splitlines()[0][:300]
                _S.engine = engine = "torch"
        if (world > 1 or force) and engine == "torch":
            _start_torch(rk, world, want, cfg, tmo)
            if cfg.init_selftest:
                from .selftest import run_init_selftest

                _S.selftest = run_init_selftest(_S.comm, rk, world, _comm_device(want))
        if _S.comm is None:
            _S.backend = "single"
        _S.initialized = True


def shutdown() -> None:
    with _S.lock:
        if not _S.initialized:
            return
        if _S.comm is not None and _S.engine == "native":
            try:
                _S.comm.shutdown()
            except Exception:  # pragma: no cover - teardown best effort
                pass
        if _S.comm is not None and _S.owns_default_group and dist.is_initialized():
            try:
                dist.destroy_process_group()
            except Exception:  # pragma: no cover - teardown best effort
                pass
        _S.__init__()


def is_initialized() -> bool:
    return _S.initialized


def _require():
    if not _S.initialized:
        raise NotInitializedError("distributed layer has not been initialised; call init() first")


def rank() -> int:
    _require()
    return _S.rank


def size() -> int:
    _require()
    return _S.size


def local_rank() -> int:
    _require()
    return _S.local_rank


def local_size() -> int:
    _require()
    return _S.local_size


def cross_rank() -> int:
    _require()
    return _S.rank // max(_S.local_size, 1)


def cross_size() -> int:
    _require()
    return max(_S.size // max(_S.local_size, 1), 1)


def backend_name() -> str:
    """'rccl', 'gloo' or 'single'."""
    _require()
    return _S.backend


def supports_avg() -> bool:
    """RCCL fuses the 1/size scaling into the reduction (ncclAvg) and so does the native engine;
    gloo does not."""
    return _S.comm.supports_avg if _S.comm is not None else _S.backend == "rccl"


def comm():
    """The active communication back-end (``parallel.comm``); None for single-process jobs."""
    return _S.comm


def trivial() -> bool:
    """True when collectives are local no-ops: one rank and no communicator (a one-rank job run
    with ``HVD_FORCE_COLLECTIVES=1`` does have one and exercises the real code path)."""
    return _S.size == 1 and _S.comm is None


def engine_name() -> str:
    return _S.engine


def engine_fallback_reason():
    """None, or why HVD_ENGINE_FALLBACK=torch had to replace the native engine in this job."""
    return _S.engine_fallback


def selftest_report():
    """What the init self-test covered ({"algos", "dtypes", "numel"}); None when it did not run."""
    return _S.selftest


def device() -> torch.device:
    """The device collectives of this rank run on."""
    _require()
    if _S.backend == "rccl" or (_S.backend == "single" and torch.cuda.is_available()):
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def group():
    _require()
    return _S.group


def engine_config() -> EngineConfig:
    return _S.config


def barrier() -> None:
    _require()
    if _S.comm is not None:
        _S.comm.barrier()

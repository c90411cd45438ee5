"""Horovod-compatible facade: ``import distributeddeeplearning_amd.torch as hvd``.

Every ``hvd.*`` symbol the reference calls plus the rest of the Horovod PyTorch
surface named there (``shutdown``, ``local_size``, in-place/async variants, ``allgather``,
``synchronize``, ``poll``, ``broadcast_optimizer_state``, ``Compression``,
``backward_passes_per_step``).  The Keras/TF-only symbols (``callbacks.*``,
``BroadcastGlobalVariablesHook``) are exposed in PyTorch-side form.
"""
from ..parallel.backend import (init, shutdown, is_initialized, rank, size, local_rank,
                                local_size, cross_rank, cross_size, barrier, comm, engine_name,
                                engine_fallback_reason, selftest_report,
                                backend_name, NotInitializedError)
from ..parallel.collectives import (allreduce, allreduce_, allreduce_async, allreduce_async_,
                                    allgather, allgather_async, broadcast, broadcast_,
                                    broadcast_async, broadcast_async_, synchronize, poll,
                                    broadcast_object, allgather_object, Sum, Average, Min, Max)
from ..parallel.functions import (broadcast_parameters, broadcast_optimizer_state,
                                  metric_average)
from ..parallel.optimizer import DistributedOptimizer
from ..parallel.compression import Compression
from .. import callbacks
from ..callbacks import BroadcastGlobalVariablesCallback as BroadcastGlobalVariablesHook


def mpi_threads_supported():
    """There is no MPI in this stack; kept for source compatibility."""
    return False


def nccl_built():
    """RCCL is the GPU data plane (``torch.distributed`` backend "nccl" on ROCm)."""
    import torch.distributed as dist

    return dist.is_nccl_available()


def gloo_built():
    import torch.distributed as dist

    return dist.is_gloo_available()


__all__ = [
    "init", "shutdown", "is_initialized", "rank", "size", "local_rank", "local_size",
    "cross_rank", "cross_size", "barrier", "allreduce", "allreduce_", "allreduce_async",
    "allreduce_async_", "allgather", "allgather_async", "broadcast", "broadcast_",
    "broadcast_async", "broadcast_async_", "synchronize", "poll", "broadcast_object",
    "allgather_object", "broadcast_parameters", "broadcast_optimizer_state", "metric_average",
    "DistributedOptimizer", "Compression", "callbacks", "BroadcastGlobalVariablesHook",
    "Sum", "Average", "Min", "Max", "mpi_threads_supported", "nccl_built", "gloo_built",
    "NotInitializedError", "comm", "engine_name", "backend_name",
    "engine_fallback_reason", "selftest_report",
]

"""Direct xGMI all-reduce over peer-mapped buckets (kernel K18).

One process per GPU.  At construction every rank shares its bucket storage and a small flag
page with its node-local peers through HIP IPC (``torch``'s CUDA-IPC storage sharing, dmabuf
based on this stack: ``HSA_ENABLE_IPC_MODE_LEGACY=0``), exchanged over the control plane.  An
all-reduce is then ONE small kernel per rank that reads/writes the peers' memory directly:
two-shot (reduce-scatter + all-gather by push, all seven links busy) for bandwidth-bound
buckets, one-shot for latency-bound ones.

Opt-in: ``HVD_ALLREDUCE_ALGO=twoshot|oneshot|auto`` (default ``rccl``).  The kernel's protocol
is unit-tested on one device with simulated ranks; multi-GPU peer visibility could not be
validated in this round (single-GPU test boxes), hence not the default.
"""
from __future__ import annotations

import torch

from . import backend as _b
from . import collectives as _c

__all__ = ["XgmiAllreduce", "ONE_SHOT_MAX_BYTES"]

ONE_SHOT_MAX_BYTES = 512 * 1024
_FLAG_WORDS = 32


def _share(t):
    st = t.untyped_storage()
    return st._share_cuda_()


def _open(handle):
    return torch.UntypedStorage._new_shared_cuda(*handle)


class XgmiAllreduce:
    """Peer-mapped view of one flat bucket tensor across the ranks of a node."""

    def __init__(self, bucket: torch.Tensor, blocks=16):
        if not bucket.is_cuda or not bucket.is_contiguous():
            raise ValueError("XgmiAllreduce needs a contiguous device tensor")
        self.rank, self.size = _b.rank(), _b.size()
        if self.size > 8 or _b.local_size() != self.size:
            raise ValueError("the direct xGMI all-reduce spans one node of at most 8 GPUs")
        if bucket.numel() % (8 * self.size):
            raise ValueError("bucket length must be a multiple of 8 x world size")
        self.bucket = bucket
        self.blocks = blocks
        self.flags = torch.zeros(_FLAG_WORDS, dtype=torch.int32, device=bucket.device)
        self.timeout_word = torch.zeros(1, dtype=torch.int32, device=bucket.device)
        self.scratch = None
        self.epoch = 0
        off = bucket.storage_offset() * bucket.element_size()
        infos = _c.allgather_object({"data": _share(bucket), "flags": _share(self.flags),
                                     "offset": off, "rank": self.rank})
        self._keep = []
        self.buf_ptrs, self.flag_ptrs = [0] * self.size, [0] * self.size
        for info in infos:
            r = info["rank"]
            if r == self.rank:
                self.buf_ptrs[r] = bucket.data_ptr()
                self.flag_ptrs[r] = self.flags.data_ptr()
                continue
            data, fl = _open(info["data"]), _open(info["flags"])
            self._keep += [data, fl]
            self.buf_ptrs[r] = data.data_ptr() + info["offset"]
            self.flag_ptrs[r] = fl.data_ptr()
        _b.barrier()

    def allreduce_(self, average=True, algo="auto"):
        """In-place all-reduce of the bucket on the current stream."""
        from ..ops import kernels

        nbytes = self.bucket.numel() * self.bucket.element_size()
        one_shot = algo == "oneshot" or (algo == "auto" and nbytes <= ONE_SHOT_MAX_BYTES)
        out = None
        if one_shot:
            if self.scratch is None:
                self.scratch = torch.empty_like(self.bucket)
            out = self.scratch
        self.epoch += 1
        outs = None
        if one_shot:
            outs = [0] * self.size
            outs[self.rank] = out.data_ptr()
        kernels.xgmi_allreduce(self.buf_ptrs, self.flag_ptrs, outs, self.bucket.numel(), self.rank,
                               self.size, self.bucket.dtype == torch.bfloat16, one_shot,
                               self.epoch, 1.0 / self.size if average else 1.0,
                               self.timeout_word, self.blocks)
        if one_shot:
            self.bucket.copy_(out)
        return self.bucket

    def check_timeout(self):
        """Host-side check (synchronises): raises if any spin in the kernel expired."""
        if int(self.timeout_word.item()) != 0:
            raise RuntimeError("xGMI all-reduce timed out waiting for a peer")

"""Checked collectives: known payloads, closed-form results, every word compared.

The reference trusts NCCL blindly and finds a broken transport as a diverging loss hours later
(its only instrument is ``NCCL_DEBUG=INFO`` in the job command,
``HorovodPytorch/01_TrainPyTorchModel.ipynb:107``).  This framework ships two data planes of its
own (RCCL called from the in-tree engine, the direct xGMI kernel), so every job proves them
before it trains:

* :func:`run_init_selftest` - called by ``hvd.init()`` when there is a communicator: one checked
  all-reduce per enabled algorithm and wire dtype, under a timeout, plus a checked broadcast and
  all-gather.  A wrong word or a timeout raises :class:`CommError` with the engine's diagnosis -
  the job exits non-zero instead of hanging or training on garbage;
* :func:`payload` / :func:`expected_average` - the rank- and index-dependent pattern and its
  closed form, shared with ``benchmarks/allreduce_bw.py`` and the tests.  Values are small
  integers (|v| <= 8), so sums over up to 8 ranks and their averages over power-of-two worlds are
  exact in bf16 as well as fp32: the comparison needs no tolerance there.
"""
from __future__ import annotations

import os

import torch

__all__ = ["payload", "expected_average", "expected_sum", "compare", "checked_allreduce",
           "run_init_selftest"]


def payload(numel, rank, dtype, device):
    """v_r[i] = ((7 i + 13 r) mod 17) - 8: differs between ranks AND between neighbouring words, so a
    collective that skips a rank, a shard, a chunk or a 16-byte lane cannot produce the closed form."""
    idx = torch.arange(numel, device=device, dtype=torch.int64)
    return (((idx * 7 + rank * 13) % 17) - 8).to(dtype)


def expected_sum(numel, world, device):
    idx = torch.arange(numel, device=device, dtype=torch.int64)
    acc = torch.zeros(numel, device=device, dtype=torch.float64)
    for r in range(world):
        acc += (((idx * 7 + r * 13) % 17) - 8).to(torch.float64)
    return acc


def expected_average(numel, world, device):
    return expected_sum(numel, world, device) / world


def compare(result, expect64, world, what):
    """Full-buffer comparison; returns None or a one-line diagnosis naming the first wrong word."""
    got = result.detach().to(torch.float64)
    exact = (world & (world - 1)) == 0
    tol = 0.0 if exact else (2.0 ** -7 if result.dtype == torch.bfloat16 else 1e-5) * 8.0
    bad = (got - expect64).abs() > tol
    bad |= ~torch.isfinite(got)
    nbad = int(bad.sum().item())
    if nbad == 0:
        return None
    first = int(torch.nonzero(bad)[0].item())
    return ("%s: %d of %d words wrong; first at index %d: expected %.6g, got %.6g"
            % (what, nbad, result.numel(), first, float(expect64[first]), float(got[first])))


def checked_allreduce(comm, rank, world, numel, dtype, device, algo=None, region=None, timeout_s=60.0,
                      name="selftest"):
    """One averaged all-reduce of the pattern through ``comm`` (optionally forcing ``algo`` on the
    native engine; ``region`` = a registered tensor to run inside).  Returns a diagnosis or None."""
    from .comm import CommError

    buf = region[:numel] if region is not None else torch.empty(numel, dtype=dtype, device=device)
    buf.copy_(payload(numel, rank, dtype, device))
    prev = getattr(comm, "force_algo", None)
    if algo is not None and hasattr(comm, "force_algo"):
        comm.force_algo = algo
    try:
        avg = comm.supports_avg
        work = comm.all_reduce(buf, "average" if avg else "sum", name)
        waiter = getattr(work, "wait_host", None) or work.wait
        if not waiter(timeout_s):
            raise CommError("%s (algo %s, %s, %d elements) did not complete within %.0f s"
                            % (name, algo or "default", dtype, numel, timeout_s))
    finally:
        if hasattr(comm, "force_algo"):
            comm.force_algo = prev
    if buf.is_cuda:
        torch.cuda.current_stream().synchronize()
    expect = expected_average(numel, world, device) if avg else expected_sum(numel, world, device)
    return compare(buf, expect, world, "%s all-reduce via %s, %s" % (name, algo or getattr(comm, "name", "?"), dtype))


def run_init_selftest(comm, rank, world, device, timeout_s=None):
    """Every data path a gradient bucket (and the start-up broadcast) can take in this job, checked
    once.  Collective: every rank calls it at the same point of ``init()``."""
    from .comm import CommError

    timeout_s = float(timeout_s or os.environ.get("HVD_SELFTEST_TIMEOUT_S", "60"))
    on_gpu = device.type == "cuda"
    dtypes = [torch.float32, torch.bfloat16] if on_gpu else [torch.float32]
    algos = comm.enabled_algos() if hasattr(comm, "enabled_algos") else [None]
    numel = 64 * 8 * max(world, 1) * 4          # satisfies the xGMI kernel's 8*world granularity
    problems = []
    region = None
    if on_gpu and any(a in ("twoshot", "oneshot") for a in algos):
        # the direct kernel runs in peer-mapped memory only: a small scratch region of its own
        region = torch.zeros(numel * 2, dtype=torch.float32, device=device)   # sized for either dtype
        comm.register_region(region)
    for algo in algos:
        for dt in dtypes:
            reg = None
            if algo in ("twoshot", "oneshot"):
                reg = region.view(dt)
            msg = checked_allreduce(comm, rank, world, numel, dt, device, algo=algo if algo else None,
                                    region=reg, timeout_s=timeout_s,
                                    name="selftest.%s.%s" % (algo or "default", str(dt).replace("torch.", "")))
            if msg:
                problems.append(msg)
    # broadcast (broadcast_parameters) and all-gather (replica check, allgather API)
    root = world - 1
    b = payload(numel, rank, torch.float32, device)
    w = comm.broadcast(b, root, "selftest.broadcast")
    if not (getattr(w, "wait_host", None) or w.wait)(timeout_s):
        raise CommError("selftest.broadcast did not complete within %.0f s" % timeout_s)
    if b.is_cuda:
        torch.cuda.current_stream().synchronize()
    if not torch.equal(b, payload(numel, root, torch.float32, device)):
        problems.append("broadcast from rank %d delivered wrong data on rank %d" % (root, rank))
    mine = payload(256, rank, torch.float32, device).to(torch.int64)
    out = torch.zeros((world, 256), dtype=torch.int64, device=device)
    w = comm.all_gather_into(out, mine, "selftest.allgather")
    if not (getattr(w, "wait_host", None) or w.wait)(timeout_s):
        raise CommError("selftest.allgather did not complete within %.0f s" % timeout_s)
    if out.is_cuda:
        torch.cuda.current_stream().synchronize()
    for r in range(world):
        if not torch.equal(out[r], payload(256, r, torch.float32, device).to(torch.int64)):
            problems.append("all-gather slot %d is wrong on rank %d" % (r, rank))
            break
    if hasattr(comm, "failed") and comm.failed():
        problems.append("the engine reported a failed collective: %s" % comm.last_error())
    if problems:
        raise CommError("communication self-test failed on rank %d of %d (engine %s):\n  %s"
                        % (rank, world, getattr(comm, "name", "?"), "\n  ".join(problems)))
    return {"algos": [a or "default" for a in algos], "dtypes": [str(d) for d in dtypes], "numel": numel}

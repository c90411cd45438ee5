#!/usr/bin/env python
"""All-reduce bandwidth by message size — the measurement behind the bucket-size choice
(build-plan step 9).  Launch one rank per GPU:

    python -m torch.distributed.run --nproc-per-node 8 benchmarks/allreduce_bw.py [--algo rccl|twoshot|oneshot]

Reports time, algorithm bandwidth (S / t) and bus bandwidth (2 (N-1)/N S / t) per size for the
framework's gradient path (bf16 buckets, fused averaging), on RCCL or on the direct xGMI kernel.
On CPU-only machines it runs over gloo (plumbing check).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import distributeddeeplearning_amd.torch as hvd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--algo", default="rccl", choices=["rccl", "twoshot", "oneshot"])
    ap.add_argument("--sizes-mib", default="0.25,1,4,8,16,32,64")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--dtype", default="bf16")
    a = ap.parse_args()
    os.environ.setdefault("HVD_ALLREDUCE_ALGO", a.algo)   # twoshot / oneshot: the direct xGMI kernel of the native engine
    hvd.init()
    cuda = torch.cuda.is_available()
    dev = torch.device("cuda", torch.cuda.current_device()) if cuda else torch.device("cpu")
    dt = torch.bfloat16 if (a.dtype == "bf16" and cuda) else torch.float32
    n = hvd.size()
    rows = []
    for mib in [float(x) for x in a.sizes_mib.split(",")]:
        numel = int(mib * (1 << 20)) // torch.empty((), dtype=dt).element_size()
        numel = max(numel // (8 * 8 * n) * (8 * 8 * n), 8 * 8 * n)
        buf = torch.ones(numel, dtype=dt, device=dev)
        comm = hvd.comm()
        if a.algo != "rccl" and comm is not None and hasattr(comm, "register_region"):
            comm.register_region(buf)   # collective: peer-map this buffer on every rank

        def once():
            hvd.synchronize(hvd.allreduce_async_(buf, average=True, name="bw.%d" % numel))

        for _ in range(3):
            once()
        if cuda:
            torch.cuda.synchronize()
        hvd.barrier()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            once()
        if cuda:
            torch.cuda.synchronize()
        dt_s = (time.perf_counter() - t0) / a.iters
        nbytes = numel * buf.element_size()
        row = {"bytes": nbytes, "us": dt_s * 1e6, "algbw_GBps": nbytes / dt_s / 1e9,
               "busbw_GBps": 2 * (n - 1) / max(n, 1) * nbytes / dt_s / 1e9, "ok": bool(abs(float(buf[0]) - 1.0) < 1e-2)}
        rows.append(row)
        if hvd.rank() == 0:
            print("%10d B  %10.1f us  alg %8.2f GB/s  bus %8.2f GB/s  %s" % (
                nbytes, row["us"], row["algbw_GBps"], row["busbw_GBps"], "ok" if row["ok"] else "WRONG"))
        if comm is not None and hasattr(comm, "failed") and comm.failed():
            raise RuntimeError("the engine reported a failed collective: %s" % comm.last_error())
    if hvd.rank() == 0:
        extra = {}
        if comm is not None and hasattr(comm, "launch_counts"):
            extra = {"launches": comm.launch_counts(), "device_ms": {k: v["mean_ms"] for k, v in comm.stats().items()}}
        print(json.dumps({"algo": a.algo, "world": n, "engine": hvd.engine_name(), "rows": rows, **extra}))
    hvd.shutdown()


if __name__ == "__main__":
    main()

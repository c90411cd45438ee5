#!/usr/bin/env python
"""All-reduce correctness and bandwidth by message size, algorithm and wire dtype — the
measurement behind the bucket plan and behind ``HVD_ALLREDUCE_ALGO=auto`` (SURVEY §5.8,
build-plan step 9).  It stands in for the NCCL tuning the reference does by environment
(``HorovodPytorch/01_TrainPyTorchModel.ipynb:107``, ``HorovodPytorch/Docker/Dockerfile:67-68``).
Launch one rank per GPU:

    python -m torch.distributed.run --nproc-per-node 8 benchmarks/allreduce_bw.py \
        --algo all --dtype both --json gpurun_out/allreduce_bw_n8.json --table profiles/allreduce_table_n8.json

For every (size, algorithm, dtype) it first runs ONE CHECKED all-reduce: rank- and index-dependent
payload (``parallel/selftest.py::payload``), the whole result buffer compared against the closed
form — a collective that reduces nothing, skips a rank or drops a shard reports WRONG and makes
the script exit non-zero.  Then it times ``--iters`` calls and reports host time per call, device
time (the engine's HIP events), algorithm bandwidth S / t and bus bandwidth 2 (N-1)/N S / t.

``--table`` writes the size thresholds ``HVD_ALLREDUCE_ALGO=auto`` reads: per size the fastest
CORRECT algorithm on the gradient wire dtype (bf16), boundaries at the geometric mean of
neighbouring sizes.  On CPU-only machines the script runs over gloo (plumbing check, fp32).
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import distributeddeeplearning_amd.torch as hvd  # noqa: E402
from distributeddeeplearning_amd.parallel import selftest  # noqa: E402


def thresholds_from(rows, dtype="bf16"):
    """[{max_bytes, algo}] ascending, last open-ended: the fastest correct algorithm per size."""
    best = {}
    for r in rows:
        if r["dtype"] != dtype or not r["ok"]:
            continue
        cur = best.get(r["bytes"])
        if cur is None or r["us"] < cur["us"]:
            best[r["bytes"]] = r
    sizes = sorted(best)
    out = []
    for i, b in enumerate(sizes):
        algo = best[b]["algo"]
        if out and out[-1]["algo"] == algo:
            out[-1]["_last"] = b
            continue
        if out:
            out[-1]["max_bytes"] = int(math.sqrt(out[-1]["_last"] * b))
        out.append({"algo": algo, "max_bytes": None, "_last": b})
    for e in out:
        e.pop("_last", None)
    return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--algo", default="rccl", choices=["rccl", "twoshot", "oneshot", "all"])
    ap.add_argument("--sizes-mib", default="0.0625,0.25,1,4,8,16,32,64")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32", "both"])
    ap.add_argument("--json", default="", help="write every row here (rank 0)")
    ap.add_argument("--table", default="", help="write the auto-selection thresholds here (rank 0)")
    a = ap.parse_args(argv)
    cuda = torch.cuda.is_available()
    algos = ["rccl", "twoshot", "oneshot"] if a.algo == "all" else [a.algo]
    if not cuda:
        algos = ["rccl"]     # label of "whatever the CPU engine does"
    if any(x != "rccl" for x in algos):
        # plane initialised, policy = RCCL unless comm.force_algo says otherwise
        os.environ["HVD_ALLREDUCE_ALGO"] = "manual"
    hvd.init()
    dev = torch.device("cuda", torch.cuda.current_device()) if cuda else torch.device("cpu")
    n, rank = hvd.size(), hvd.rank()
    comm = hvd.comm()
    if comm is None:
        raise SystemExit("allreduce_bw needs a communicator: launch with >= 2 ranks or HVD_FORCE_COLLECTIVES=1")
    native = hasattr(comm, "force_algo")
    if not native:
        algos = ["rccl"]
    if native and n == 1:
        algos = ["rccl"]     # the xGMI plane is not set up for a single rank
    dts = {"bf16": [torch.bfloat16], "fp32": [torch.float32], "both": [torch.bfloat16, torch.float32]}[a.dtype]
    if not cuda:
        dts = [torch.float32]
    rows, wrong = [], 0
    for mib in [float(x) for x in a.sizes_mib.split(",")]:
        for dt in dts:
            es = torch.empty((), dtype=dt).element_size()
            gran = 8 * 8 * n
            numel = max(int(mib * (1 << 20)) // es // gran * gran, gran)
            buf = torch.zeros(numel, dtype=dt, device=dev)
            if native and any(x != "rccl" for x in algos):
                comm.register_region(buf)   # collective: peer-map this buffer on every rank
            for algo in algos:
                if algo == "oneshot" and numel * es > 8 << 20:
                    continue                # latency algorithm: every rank reads every buffer whole
                name = "bw.%s.%s.%d" % (algo, str(dt).replace("torch.", ""), numel)
                msg = selftest.checked_allreduce(comm, rank, n, numel, dt, dev, algo=algo if native else None,
                                                 region=buf if native else None, name=name)
                ok = msg is None
                if not ok:
                    wrong += 1
                    print("[rank %d] WRONG %s" % (rank, msg), file=sys.stderr, flush=True)
                if native:
                    comm.force_algo = algo

                def once():
                    hvd.synchronize(hvd.allreduce_async_(buf, average=True, name=name))

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
                if native:
                    comm.force_algo = None
                nbytes = numel * es
                dev_ms = None
                if native:
                    st = comm.stats().get(name)
                    dev_ms = st["mean_ms"] if st else None
                row = {"algo": algo, "dtype": "bf16" if dt == torch.bfloat16 else "fp32", "bytes": nbytes,
                       "us": dt_s * 1e6, "device_us": None if dev_ms is None else dev_ms * 1e3,
                       "algbw_GBps": nbytes / dt_s / 1e9,
                       "busbw_GBps": 2 * (n - 1) / max(n, 1) * nbytes / dt_s / 1e9, "ok": ok}
                rows.append(row)
                if rank == 0:
                    print("%-8s %-4s %10d B  %10.1f us  dev %s us  alg %8.2f GB/s  bus %8.2f GB/s  %s" % (
                        algo, row["dtype"], nbytes, row["us"],
                        "%8.1f" % row["device_us"] if row["device_us"] is not None else "       -",
                        row["algbw_GBps"], row["busbw_GBps"], "ok" if ok else "WRONG"), flush=True)
                if native and comm.failed():
                    raise RuntimeError("the engine reported a failed collective: %s" % comm.last_error())
    # every rank's verdict counts: a wrong word on any rank fails the run everywhere
    total_wrong = sum(hvd.allgather_object(wrong))
    if rank == 0:
        doc = {"world": n, "engine": hvd.engine_name(), "rows": rows, "wrong": total_wrong,
               "launches": comm.launch_counts() if native else None}
        if a.json:
            os.makedirs(os.path.dirname(os.path.abspath(a.json)), exist_ok=True)
            with open(a.json, "w") as fh:
                json.dump(doc, fh, indent=1)
        if a.table and total_wrong == 0:
            wire = "bf16" if cuda else "fp32"
            table = {"world": n, "engine": hvd.engine_name(), "dtype": wire, "thresholds": thresholds_from(rows, wire),
                     "source": "benchmarks/allreduce_bw.py %s" % " ".join(sys.argv[1:])}
            os.makedirs(os.path.dirname(os.path.abspath(a.table)), exist_ok=True)
            with open(a.table, "w") as fh:
                json.dump(table, fh, indent=1)
            print("thresholds: %s" % table["thresholds"])
        print(json.dumps({k: v for k, v in doc.items() if k != "rows"} | {"n_rows": len(rows)}))
    hvd.shutdown()
    if total_wrong:
        raise SystemExit("allreduce_bw: %d checked all-reduce(s) delivered wrong data" % total_wrong)


if __name__ == "__main__":
    main()

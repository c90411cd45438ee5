#!/usr/bin/env python
"""Weak-scaling sweep: runs ``bench.py`` at 1/2/4/8 GPUs of this node and records, per GPU count,
images/sec, ms/step, which engine and all-reduce algorithm carried the gradients, the exposed
communication time and the device time of every bucket (BASELINE.json target: >= 90 % scaling
efficiency from 1 to 8 MI355X).  Self-skipping: GPU counts above ``torch.cuda.device_count()``
are not attempted.  The reference's equivalent is reading ``Total images/sec`` off the job logs
at different ``-np`` (``HorovodPytorch/src/imagenet_pytorch_horovod.py:242-255``).

    python benchmarks/scaling.py [--gpus 1,2,4,8] [--steps 30] [--warmup 8] [--model resnet50]
                                 [--ab] [--busbw] [--json profiles/scaling.json]

``--ab``     at every N > 1 additionally runs the alternatives back to back: the c10d engine
             (``HVD_ENGINE=torch``) against the native one, and the direct xGMI kernel
             (``HVD_ALLREDUCE_ALGO=twoshot``) against RCCL - one JSON row each.
``--busbw``  at every N > 1 runs ``benchmarks/allreduce_bw.py --algo all --dtype both`` (every
             result word checked) and stores its bandwidth table and the thresholds file that
             ``HVD_ALLREDUCE_ALGO=auto`` reads (``profiles/allreduce_table_n<N>.json``).
A failed variant is recorded with its error tail instead of ending the sweep: the first run on
a real node must yield everything it can.
"""
import argparse
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(n, script, args, env_extra=None, timeout=1800):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(env_extra or {})
    if n == 1:
        cmd = [sys.executable, script] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script] + args
    try:
        out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, env=env, timeout=timeout)
        return out.returncode, out.stdout, out.stderr
    except subprocess.TimeoutExpired as e:
        return 124, (e.stdout or b"").decode(errors="replace") if isinstance(e.stdout, bytes) else (e.stdout or ""), \
            "timed out after %d s" % timeout


def run_bench(n, steps, warmup, extra, env_extra=None):
    rc, so, se = _launch(n, os.path.join(ROOT, "bench.py"),
                         ["--gpus", str(n), "--steps", str(steps), "--warmup", str(warmup)] + extra, env_extra)
    for line in reversed(so.splitlines()):
        if line.startswith("{"):
            return json.loads(line), None
    return None, "rc %d\n%s\n%s" % (rc, so[-1500:], se[-2500:])


def row_from(r, n, variant, base):
    comm = r.get("comm") or {}
    eff = None
    if base:
        eff = r["value"] / (base * r["n_gpus"])
    return {"n_gpus": n, "variant": variant, "images_per_sec": r["value"], "ms_per_step": r["ms_per_step"],
            "host_issue_ms_per_step": r.get("host_issue_ms_per_step"), "efficiency": eff,
            "engine": r.get("engine"), "engine_fallback": r.get("engine_fallback"),
            "allreduce_algo": r.get("allreduce_algo"),
            "comm_exposed_ms": comm.get("comm_exposed_ms_mean", comm.get("comm_exposed_ms")),
            "bucket_ms": comm.get("bucket_ms"), "final_loss": r.get("final_loss")}


def main(argv=None):
    import torch

    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", default="1,2,4,8")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--model", default="resnet50")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--optimizer", default="sgd")
    ap.add_argument("--ab", action="store_true", help="native vs torch engine, rccl vs twoshot, back to back")
    ap.add_argument("--busbw", action="store_true", help="checked all-reduce bandwidth table per N")
    ap.add_argument("--json", default="")
    a = ap.parse_args(argv)
    have = torch.cuda.device_count()
    extra = ["--model", a.model, "--batch", str(a.batch), "--optimizer", a.optimizer]
    rows, busbw, base = [], {}, None
    for n in [int(x) for x in a.gpus.split(",")]:
        if n > have:
            print("skip %d GPUs (node has %d)" % (n, have))
            continue
        variants = [("default", {})]
        if a.ab and n > 1:
            variants += [("engine=torch", {"HVD_ENGINE": "torch"}),
                         ("algo=twoshot", {"HVD_ALLREDUCE_ALGO": "twoshot"})]
        for label, env in variants:
            r, err = run_bench(n, a.steps, a.warmup, extra, env)
            if r is None:
                rows.append({"n_gpus": n, "variant": label, "error": err})
                print("%d GPU(s) [%s]: FAILED\n%s" % (n, label, err))
                continue
            if base is None and label == "default":
                base = r["value"] / r["n_gpus"]
            row = row_from(r, n, label, base)
            rows.append(row)
            print("%d GPU(s) [%-12s]: %10.1f img/s  %8.2f ms/step  eff %s  engine %s  algo %s  comm exposed %s ms"
                  % (n, label, row["images_per_sec"], row["ms_per_step"],
                     "%.1f %%" % (row["efficiency"] * 100) if row["efficiency"] else "-", row["engine"],
                     row["allreduce_algo"], row["comm_exposed_ms"]))
        if a.busbw and n > 1:
            table = os.path.join(ROOT, "profiles", "allreduce_table_n%d.json" % n)
            full = os.path.join(ROOT, "profiles", "allreduce_bw_n%d.json" % n)
            rc, so, se = _launch(n, os.path.join(ROOT, "benchmarks", "allreduce_bw.py"),
                                 ["--algo", "all", "--dtype", "both", "--json", full, "--table", table])
            print(so[-3000:])
            busbw[str(n)] = {"rc": rc, "table": table if rc == 0 else None,
                             "error": None if rc == 0 else (so[-1000:] + se[-2000:])}
            if os.path.isfile(full):
                with open(full) as fh:
                    busbw[str(n)]["rows"] = json.load(fh).get("rows")
    if a.json:
        with open(a.json, "w") as fh:
            json.dump({"rows": rows, "busbw": busbw, "model": a.model, "per_gpu_batch": a.batch}, fh, indent=1)
    return rows


if __name__ == "__main__":
    main()

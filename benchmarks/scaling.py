#!/usr/bin/env python
"""Weak-scaling sweep: runs ``bench.py`` at 1/2/4/8 GPUs of this node and prints images/sec and
scaling efficiency (BASELINE.json target: >= 90 % from 1 to 8 MI355X).  Self-skipping: GPU
counts above ``torch.cuda.device_count()`` are not attempted.

    python benchmarks/scaling.py [--gpus 1,2,4,8] [--steps 30] [--warmup 8] [--model resnet50]
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


def run(n, steps, warmup, extra):
    bench = os.path.join(ROOT, "bench.py")
    args = ["--gpus", str(n), "--steps", str(steps), "--warmup", str(warmup)] + extra
    if n == 1:
        cmd = [sys.executable, bench] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), bench] + args
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True)
    for line in reversed(out.stdout.splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise RuntimeError("bench at %d GPUs produced no result:\n%s\n%s" % (n, out.stdout[-2000:], out.stderr[-2000:]))


def main():
    import torch

    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", default="1,2,4,8")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--model", default="resnet50")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--optimizer", default="sgd")
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    have = torch.cuda.device_count()
    rows, base = [], None
    for n in [int(x) for x in a.gpus.split(",")]:
        if n > have:
            print("skip %d GPUs (node has %d)" % (n, have))
            continue
        r = run(n, a.steps, a.warmup, ["--model", a.model, "--batch", str(a.batch), "--optimizer", a.optimizer])
        base = base or r["value"] / r["n_gpus"]
        eff = r["value"] / (base * r["n_gpus"])
        rows.append({"n_gpus": n, "images_per_sec": r["value"], "ms_per_step": r["ms_per_step"], "efficiency": eff})
        print("%d GPU(s): %10.1f img/s  %8.2f ms/step  efficiency %.1f %%" % (n, r["value"], r["ms_per_step"], eff * 100))
    if a.json:
        json.dump(rows, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()

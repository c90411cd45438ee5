#!/usr/bin/env python
"""What a late peer costs the local backward pass.

The direct xGMI all-reduce kernel (csrc/kernels/xgmi_allreduce.hip) is RESIDENT while it waits for
its peers' flags: a few workgroups that poll.  Next to them run the backward convolutions.  This
benchmark measures, in the two-processes-on-one-GPU rig (the only multi-rank hardware the boxes
have), how much a fixed block of convolution launches on rank 0 slows down while its all-reduce of a
16 MB bucket spins for a peer that arrives D ms late - per stream priority of the communication
stream and per grid size of the kernel.  Rank 1 does NO compute (it only sleeps and joins), so the
device is rank 0's alone and any slowdown is the spinning kernel's.

    python benchmarks/late_peer.py                 # driver: launches the 2-rank runs, prints a table
    python benchmarks/late_peer.py --json out.json

Reference behaviour this protects: the reference's gradient exchange overlaps with the rest of
backward on a side stream (``HorovodPytorch/src/imagenet_pytorch_horovod.py:217-218`` through Horovod's
background thread); a collective that steals the compute units it overlaps with defeats that.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker():
    import torch
    import torch.nn.functional as F  # noqa: F401

    import distributeddeeplearning_amd.torch as hvd
    from distributeddeeplearning_amd.ops import hip_ops as hip

    hvd.init()
    r = hvd.rank()
    comm = hvd.comm()
    assert hvd.engine_name() == "native" and comm.algo != "rccl", (hvd.engine_name(), comm.algo)
    BF, CL = torch.bfloat16, torch.channels_last
    arena = torch.zeros(8 << 20, dtype=BF, device="cuda")       # one 16 MB gradient bucket
    comm.register_region(arena)
    delays = [float(v) for v in os.environ.get("LATE_MS", "0,5").split(",")]
    rounds = int(os.environ.get("LATE_ROUNDS", "12"))

    if r == 0:
        g = torch.Generator(device="cuda").manual_seed(1)
        # a block of backward-like work: matrix-bound 3x3 (c4) and bandwidth-bound 1x1 (a4) launches
        xc = torch.randn((256, 256, 14, 14), generator=g, device="cuda").to(BF).contiguous(memory_format=CL)
        wc = (torch.randn((256, 256, 3, 3), generator=g, device="cuda") * 0.02).to(BF).contiguous(memory_format=CL)
        xa = torch.randn((256, 64, 56, 56), generator=g, device="cuda").to(BF).contiguous(memory_format=CL)
        wa = (torch.randn((256, 64, 1, 1), generator=g, device="cuda") * 0.1).to(BF).contiguous(memory_format=CL)

        def work():
            with torch.no_grad():
                for _ in range(4):
                    for _ in range(8):
                        hip.conv2d_stats(xc, wc, 1, 1, True)
                    for _ in range(4):
                        hip.conv2d_stats(xa, wa, 1, 0, True)

        for _ in range(3):
            work()
        torch.cuda.synchronize()

    def timed_round(delay_ms, with_collective, tag):
        hvd.barrier()
        torch.cuda.synchronize()
        h = None
        if r == 0:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if with_collective:
                h = hvd.allreduce_async_(arena, op=hvd.Average, name=tag)
            e0.record()
            work()
            e1.record()
            if h is not None:
                hvd.synchronize(h)
            torch.cuda.synchronize()
            return e0.elapsed_time(e1)
        if with_collective:
            time.sleep(delay_ms / 1e3)
            h = hvd.allreduce_async_(arena, op=hvd.Average, name=tag)
            hvd.synchronize(h)
        torch.cuda.synchronize()
        return 0.0

    res = {}
    base = [timed_round(0.0, False, "none") for _ in range(rounds)]
    res["no_collective_ms"] = sorted(base)[len(base) // 2]
    for d in delays:
        ts = [timed_round(d, True, "late.%g" % d) for _ in range(rounds)]
        res["late_%g_ms" % d] = sorted(ts)[len(ts) // 2]
    if r == 0:
        print("RESULT " + json.dumps(res, sort_keys=True), flush=True)
    hvd.shutdown()


def driver(args):
    from distributeddeeplearning_amd.launch import run as ddlrun
    import io

    table = []
    for prio in args.priorities.split(","):
        for blocks in args.blocks.split(","):
            env = {"OMP_NUM_THREADS": "2", "DDL_RCCL_SKIP_COMM": "1", "HVD_ALLREDUCE_ALGO": "twoshot",
                   "HSA_ENABLE_IPC_MODE_LEGACY": "0", "HVD_COLLECTIVE_TIMEOUT_S": "60",
                   "DDL_COMM_STREAM_PRIORITY": prio, "HVD_XGMI_BLOCKS": blocks, "LATE_MS": args.late_ms,
                   "LATE_ROUNDS": str(args.rounds), "LATE_PEER_WORKER": "1"}
            sink = io.StringIO()
            rc = ddlrun.launch([sys.executable, "-u", os.path.abspath(__file__)], 2, hosts=[("localhost", 2)],
                               exports=env, timeout=180, sink=sink)
            out = sink.getvalue()
            rec = None
            for line in out.splitlines():
                if "RESULT " in line:
                    rec = json.loads(line.split("RESULT ", 1)[1])
            if rc != 0 or rec is None:
                print("priority %s blocks %s: FAILED rc=%s\n%s" % (prio, blocks, rc, out[-1500:]), flush=True)
                return 1
            rec.update(priority=prio, blocks=int(blocks))
            table.append(rec)
            print(json.dumps(rec, sort_keys=True), flush=True)
    print("\ncomm stream | blocks | work alone ms | + all-reduce, peer on time | + peer late by %s ms" % args.late_ms.split(",")[-1])
    for rec in table:
        keys = sorted(k for k in rec if k.startswith("late_"))
        b = rec["no_collective_ms"]
        print("%-11s | %6d | %13.3f | %s" % (rec["priority"], rec["blocks"], b,
                                           " | ".join("%.3f (%+.1f %%)" % (rec[k], 100.0 * (rec[k] - b) / b) for k in keys)))
    if args.json:
        with open(args.json, "w") as fh:
            json.dump(table, fh, indent=1)
    return 0


if __name__ == "__main__":
    if os.environ.get("LATE_PEER_WORKER") == "1":
        worker()
    else:
        ap = argparse.ArgumentParser()
        ap.add_argument("--priorities", default="high,normal,low")
        ap.add_argument("--blocks", default="16,64")
        ap.add_argument("--late-ms", default="0,5")
        ap.add_argument("--rounds", type=int, default=12)
        ap.add_argument("--json", default="")
        sys.exit(driver(ap.parse_args()))

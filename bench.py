#!/usr/bin/env python
"""Headline benchmark: ResNet-50 synthetic 224×224 training throughput (images/sec, whole job).

Contract (see the task statement): ``python bench.py --gpus N --steps K --warmup W``; for N>1
the driver launches one rank per GPU with ``torch.distributed.run`` and this script reads
RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* from the environment.  W untimed warm-up steps, then
exactly K timed steps bracketed by barrier + ``torch.cuda.synchronize()``; the time is the MAX
over ranks and rank 0 prints ONE JSON line.

Metric and configuration follow BASELINE.json: ResNet-50, bf16, synthetic data, batch 256 per
GPU (weak scaling), SGD momentum 0.9 with lr 0.001×N as in the reference
(``HorovodPytorch/src/imagenet_pytorch_horovod.py:333``), random-init weights.  Every timed
step does the full work of the reference's hot loop (``:210-218``): batch fetch, zero_grad,
forward, loss, backward with overlapped gradient all-reduce, optimizer step.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# The benchmark must yield a number on the first multi-GPU node it ever sees: if the native engine
# fails to start or fails its checked self-test there, every rank agrees to continue on
# torch.distributed's RCCL process group - reported on stderr and in the JSON line ("engine",
# "engine_fallback").  Training scripts keep the strict default (exit non-zero).
os.environ.setdefault("HVD_ENGINE_FALLBACK", "torch")

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import distributeddeeplearning_amd.torch as hvd  # noqa: E402
from distributeddeeplearning_amd.data.synthetic import DeviceSyntheticLoader  # noqa: E402
from distributeddeeplearning_amd.models import build_model  # noqa: E402
from distributeddeeplearning_amd.ops import functional as ops  # noqa: E402
from distributeddeeplearning_amd.optim import FusedLARS, FusedSGD  # noqa: E402

# Stock PyTorch-ROCm (MIOpen, bf16 autocast, channels_last) ResNet-50 batch 256 on one MI355X,
# measured with benchmarks/stock_pytorch.py — see BASELINE.md.  The reference itself
# publishes no number.
STOCK_PYTORCH_IMG_PER_S_1GPU = None


def _read_baseline():
    path = os.path.join(ROOT, "BASELINE_MEASURED.json")
    try:
        with open(path) as fh:
            return float(json.load(fh)["stock_pytorch_resnet50_bf16_b256_img_per_s_1gpu"])
    except Exception:
        return STOCK_PYTORCH_IMG_PER_S_1GPU


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (BASELINE config: 256)")
    ap.add_argument("--model", default="resnet50")
    ap.add_argument("--optimizer", default="sgd", choices=["sgd", "lars"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--device", default=None, help="cuda|cpu (default: cuda when available)")
    ap.add_argument("--image", type=int, default=224)
    ap.add_argument("--stall-watch", type=float, default=0.0, metavar="SECONDS",
                    help="diagnostic: dump every thread's stack to stderr when one step takes longer than this")
    ap.add_argument("--tape", default="auto", choices=["auto", "on", "off"],
                    help="replay the step from the native launch tape after the first warm-up steps "
                         "(training/tape.py: same kernels, arguments, streams and order as the eager step; "
                         "auto = one rank, --warmup >= 3)")
    ap.add_argument("--graph", action="store_true",
                    help="replay the whole step as one HIP graph after the first warm-up steps "
                         "(training/graph.py); needs --warmup >= 2")
    return ap.parse_args(argv)


_T_START = time.perf_counter()


def main(argv=None):
    args = parse_args(argv)
    # DDL_BENCH_WATCHDOG=<seconds>: every thread's Python stack goes to stderr if the run is still alive after that
    # long (repeated): a hang in start-up, in a collective or in a step names itself instead of just timing out
    # (several ranks: on by default - the first run on a multi-GPU node has to explain itself if it stalls)
    wd = float(os.environ.get("DDL_BENCH_WATCHDOG", "180" if int(os.environ.get("WORLD_SIZE", "1")) > 1 else "0") or 0)
    if wd > 0:
        import faulthandler

        faulthandler.dump_traceback_later(wd, repeat=True, exit=False)
    use_cuda = (args.device or ("cuda" if torch.cuda.is_available() else "cpu")) == "cuda"
    hvd.init()
    world, rank = hvd.size(), hvd.rank()
    if args.gpus != world and rank == 0:
        print("note: --gpus %d but WORLD_SIZE is %d; using WORLD_SIZE" % (args.gpus, world),
              file=sys.stderr)
    device = torch.device("cuda", torch.cuda.current_device()) if use_cuda else torch.device("cpu")
    torch.manual_seed(42)
    dtype = torch.bfloat16 if (args.dtype == "bf16" and use_cuda) else torch.float32

    model = build_model(args.model, num_classes=1000)
    if dtype == torch.bfloat16:
        model.to_compute(device, dtype)
    else:
        model.to(device)
    model.train()
    hvd.broadcast_parameters(model.state_dict(), root_rank=0)

    lr = 0.001 * world
    if args.optimizer == "lars":
        opt = FusedLARS(model.named_parameters(), lr=lr, momentum=0.9, weight_decay=5e-5)
    else:
        opt = FusedSGD(model.named_parameters(), lr=lr, momentum=0.9)
    opt = hvd.DistributedOptimizer(opt, named_parameters=model.named_parameters())
    # one process: each gradient bucket is updated as soon as it is complete, next to the rest of backward
    # (nothing here touches gradients between backward() and step()); DDL_EARLY_UPDATE=0 = all updates in step()
    early_update = os.environ.get("DDL_EARLY_UPDATE", "1") != "0" and opt.enable_early_update()

    loader = DeviceSyntheticLoader(args.batch, length=args.batch * world * (args.steps + args.warmup),
                                   height=args.image, width=args.image, device=device,
                                   dtype=dtype, rank=rank, world=world)

    graphed = None
    if args.graph and use_cuda and args.warmup >= 2:
        from distributeddeeplearning_amd.training.graph import GraphedTrainStep

        graphed = GraphedTrainStep(model, opt, ops.softmax_cross_entropy,
                                   warmup=min(3, args.warmup - 1),
                                   allow_collectives=os.environ.get("DDL_GRAPH_COLLECTIVES", "0") == "1")

    taped = None
    want_tape = args.tape == "on" or (args.tape == "auto" and world == 1 and hvd.comm() is None)
    if want_tape and use_cuda and graphed is None and args.warmup >= 3 and ops.backend() != "aten":
        from distributeddeeplearning_amd.training.tape import TapedTrainStep

        # eager steps first (lazy initialisation), the recording (a real step, slow: it allocates its
        # private pool) inside the warm-up, replays from then on
        taped = TapedTrainStep(model, opt, ops.softmax_cross_entropy, loader=loader,
                               warmup=min(3, args.warmup - 2),
                               allow_collectives=os.environ.get("DDL_TAPE_COLLECTIVES", "0") == "1")

    def one_step():
        if taped is not None:
            return taped()
        images, labels = loader.batch()
        if graphed is not None:
            return graphed(images, labels)
        opt.zero_grad()
        loss = ops.softmax_cross_entropy(model(images), labels)
        loss.backward()
        opt.step()
        return loss

    def sync():
        hvd.barrier()
        if use_cuda:
            torch.cuda.synchronize()

    if args.stall_watch > 0:
        import faulthandler

        plain_step = one_step

        counter = {"step": 0}

        def one_step():  # noqa: F811 - diagnostic wrapper
            faulthandler.dump_traceback_later(args.stall_watch, exit=False)
            ms0 = torch.cuda.memory_stats() if use_cuda else {}
            t_s = time.perf_counter()
            out = plain_step()
            faulthandler.cancel_dump_traceback_later()
            dt = time.perf_counter() - t_s
            counter["step"] += 1
            if dt > args.stall_watch:
                ms1 = torch.cuda.memory_stats() if use_cuda else {}
                keys = ("segment.all.allocated", "segment.all.freed", "num_alloc_retries", "num_device_alloc",
                        "num_device_free", "num_sync_all_streams")
                delta = {k: ms1.get(k, 0) - ms0.get(k, 0) for k in keys}
                print("stall-watch: step %d took %.3f s on the host (%.2f s after start); allocator deltas %s; "
                      "reserved %.1f GB" % (counter["step"], dt, time.perf_counter() - _T_START, delta,
                                            ms1.get("reserved_bytes.all.current", 0) / 2 ** 30),
                      file=sys.stderr, flush=True)
            return out

    loss = None
    for _ in range(args.warmup):
        loss = one_step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = one_step()
    host_elapsed = time.perf_counter() - t0  # host-side issue time (the GPU may still be working)
    sync()
    elapsed = time.perf_counter() - t0

    # host cost of issuing ONE step into an empty queue (the loop above runs ahead of the GPU until the
    # hardware queue pushes back, so its host time says little once the step is GPU-bound)
    single = []
    for _ in range(3 if use_cuda else 0):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        loss = one_step()
        single.append(time.perf_counter() - t1)
    sync()

    t = torch.tensor([elapsed], dtype=torch.float64, device=device if use_cuda else "cpu")
    if world > 1:
        t = hvd.allreduce(t, op=hvd.Max, name="bench.elapsed")
    elapsed = float(t.item())
    final_loss = float(loss.float().item()) if loss is not None else float("nan")

    if rank == 0:
        global_batch = args.batch * world
        value = global_batch * args.steps / elapsed
        base = _read_baseline()
        # the stock number is a 1-GPU figure; compare per-GPU throughput against it
        # (measured for ResNet-50, batch 256, bf16 only: other configurations report null)
        flagship = args.model == "resnet50" and args.batch == 256 and args.image == 224 and dtype == torch.bfloat16
        vs = (value / world) / base if (base and flagship) else None
        pretty = {"resnet%d" % d: "ResNet-%d" % d for d in (18, 34, 50, 101, 152, 200)}.get(args.model, args.model)
        print(json.dumps({
            "metric": "images/sec (whole node) %s synthetic %dx%d training" % (pretty, args.image, args.image),
            "value": round(value, 2),
            "unit": "images/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "host_issue_ms_per_step": round(host_elapsed / args.steps * 1e3, 3),
            "host_issue_ms_single_step": round(min(single) * 1e3, 3) if single else None,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None if vs is None else round(vs, 4),
            "baseline": "stock PyTorch-ROCm ResNet-50 bf16 channels_last batch 256 on one MI355X "
                        "(BASELINE_MEASURED.json); ratio of per-GPU throughput",
            "dtype": "bf16" if dtype == torch.bfloat16 else "fp32",
            "data": "synthetic (on-GPU Philox pool, random-init weights)",
            "final_loss": round(final_loss, 4),
            "ops_backend": "hip" if (use_cuda and ops.backend() != "aten") else "aten",
            "hip_graph": bool(graphed is not None and graphed.graphed),
            # launch tape: the timed steps were replayed from native code (launches per step) or issued from Python
            "launch_tape": bool(taped is not None and taped.taped),
            "early_update": bool(early_update),
            "tape_launches": taped.launches if (taped is not None and taped.taped) else None,
            "tape_refused": None if (taped is None or taped.failed is None) else str(taped.failed)[:200],
            "reserved_gb": round(torch.cuda.memory_reserved() / 2 ** 30, 1) if use_cuda else 0.0,
            # gradient exchange: which engine carried it, how long the compute stream sat waiting for
            # it per step (HIP events) and the device time of each bucket's collective
            "engine": hvd.engine_name() if hvd.comm() is not None else "none",
            "engine_fallback": hvd.engine_fallback_reason(),
            "selftest": hvd.selftest_report(),
            "allreduce_algo": getattr(hvd.comm(), "algo", "rccl") if hvd.comm() is not None else "none",
            "comm": opt.comm_metrics() if hasattr(opt, "comm_metrics") else None,
            "config": {"model": args.model, "global_batch": global_batch,
                       "per_gpu_batch": args.batch, "image": args.image,
                       "optimizer": args.optimizer, "parallelism": "dp%d" % world},
        }))
    hvd.shutdown()


if __name__ == "__main__":
    main()

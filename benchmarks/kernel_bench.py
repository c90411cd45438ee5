#!/usr/bin/env python
"""Per-kernel microbenchmark: the 23 unique ResNet-50 conv shapes x forward /
backward-data / backward-weights on the hand-written HIP kernels, plus the BatchNorm kernels.

    python benchmarks/kernel_bench.py [--batch 256] [--iters 10] [--aten] [--only a3,c4]

Prints achieved TFLOP/s (2*M*N*K) and effective HBM GB/s (compulsory bytes) per kernel; with
``--aten`` also times the ATen/MIOpen op of the same shape for comparison.  Results feed
``profiles/`` and the per-shape tile table in ``csrc/kernels/conv_igemm.hip``.
"""
import argparse
import json
import math
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = [  # id, k, cin, cout, stride, hin, count
    ("a1", 1, 64, 64, 1, 56, 1), ("a2", 1, 256, 64, 1, 56, 2), ("a3", 3, 64, 64, 1, 56, 3),
    ("a4", 1, 64, 256, 1, 56, 4), ("b1", 1, 256, 128, 1, 56, 1), ("b2", 3, 128, 128, 2, 56, 1),
    ("b3", 1, 512, 128, 1, 28, 3), ("b4", 3, 128, 128, 1, 28, 3), ("b5", 1, 128, 512, 1, 28, 4),
    ("b6", 1, 256, 512, 2, 56, 1), ("c1", 1, 512, 256, 1, 28, 1), ("c2", 3, 256, 256, 2, 28, 1),
    ("c3", 1, 1024, 256, 1, 14, 5), ("c4", 3, 256, 256, 1, 14, 5), ("c5", 1, 256, 1024, 1, 14, 6),
    ("c6", 1, 512, 1024, 2, 28, 1), ("d1", 1, 1024, 512, 1, 14, 1), ("d2", 3, 512, 512, 2, 14, 1),
    ("d3", 1, 2048, 512, 1, 7, 2), ("d4", 3, 512, 512, 1, 7, 2), ("d5", 1, 512, 2048, 1, 7, 3),
    ("d6", 1, 1024, 2048, 2, 14, 1),
]


def timeit(fn, iters):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters  # ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--aten", action="store_true")
    ap.add_argument("--blaslt", action="store_true",
                    help="1x1 stride-1 shapes: also time the plain library GEMM (torch.matmul -> hipBLASLt)")
    ap.add_argument("--epi4", action="store_true",
                    help="also time backward-data of the 1x1 layers with the chain epilogue of the training "
                         "step (shortcut addend + ReLU bit mask + BatchNorm partials) and report its traffic rate")
    ap.add_argument("--only", default="")
    ap.add_argument("--json", default="")
    args = ap.parse_args()
    from distributeddeeplearning_amd.ops import hip_ops, kernels

    kernels.require()
    dev, bf, cl = "cuda", torch.bfloat16, torch.channels_last
    B = args.batch
    only = set(filter(None, args.only.split(",")))
    rows, tot = [], {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0, "aten_fwd": 0.0, "aten_bwd": 0.0}
    print("%-4s %-28s %9s %9s %9s | TF/s %6s %6s %6s | GB/s %6s %6s %6s" % (
        "id", "shape", "fwd us", "dgrad us", "wgrad us", "fwd", "dgrad", "wgrad", "fwd", "dgrad",
        "wgrad"))
    for sid, k, cin, cout, stride, hin, count in SHAPES:
        if only and sid not in only:
            continue
        pad = k // 2
        x = torch.randn((B, cin, hin, hin), device=dev).to(bf).contiguous(memory_format=cl)
        w = (torch.randn((cout, cin, k, k), device=dev) / math.sqrt(cin * k * k)).to(bf)
        w = w.contiguous(memory_format=cl)
        hout = (hin + 2 * pad - k) // stride + 1
        M = B * hout * hout
        flops = 2.0 * M * cout * cin * k * k
        y = torch.empty((B, cout, hout, hout), dtype=bf, device=dev, memory_format=cl)
        stats = torch.empty((kernels.conv_tiles_m(M), 2, cout), dtype=torch.float32, device=dev)
        dy = torch.randn((B, cout, hout, hout), device=dev).to(bf).contiguous(memory_format=cl)
        dx = torch.empty_like(x)
        hip_ops.reset_caches()
        wt = hip_ops._WT.get(w)
        geom_x, geom_dy = hip_ops._geom(x), hip_ops._geom(dy)

        def fwd():
            kernels.conv_igemm(x, geom_x, w, y, cout, k, k, stride, pad, pad, 1, hout, hout,
                               stats=stats)

        def dgrad():
            if k == 1 and stride == 2:  # hot path: compact GEMM, scattered by the next kernel
                hip_ops._conv_dgrad_1x1s2_compact(dy, w)
            else:
                kernels.conv_igemm(dy, geom_dy, wt, dx, cin, k, k, 1, k - 1 - pad, k - 1 - pad,
                                   stride, hin, hin)

        dw = torch.empty_like(w)

        def wgrad():
            hip_ops._wgrad_into(dy, x, geom_x, dw, cout, k, k, stride, pad, pad, hout, hout)

        tf_, td, tw = timeit(fwd, args.iters), timeit(dgrad, args.iters), timeit(wgrad, args.iters)
        bx, by, bw = x.numel() * 2, y.numel() * 2, w.numel() * 2
        gb = lambda b, t: b / t / 1e6  # noqa: E731
        row = {"id": sid, "k": k, "cin": cin, "cout": cout, "stride": stride, "hin": hin,
               "count": count, "fwd_us": tf_ * 1e3, "dgrad_us": td * 1e3, "wgrad_us": tw * 1e3,
               "fwd_tflops": flops / tf_ / 1e9, "dgrad_tflops": flops / td / 1e9,
               "wgrad_tflops": flops / tw / 1e9}
        line = "%-4s %-28s %9.1f %9.1f %9.1f | TF/s %6.0f %6.0f %6.0f | GB/s %6.0f %6.0f %6.0f" % (
            sid, "%dx%d %d->%d s%d @%d" % (k, k, cin, cout, stride, hin), tf_ * 1e3, td * 1e3,
            tw * 1e3, row["fwd_tflops"], row["dgrad_tflops"], row["wgrad_tflops"],
            gb(bx + by + bw, tf_), gb(bx + by + bw, td), gb(bx + by + bw, tw))
        tot["fwd"] += tf_ * count
        tot["dgrad"] += td * count
        tot["wgrad"] += tw * count
        if args.epi4 and k == 1 and stride == 1 and cin >= 4 * cout:
            # conv1 of a bottleneck block: dx [M, cin] = dy W + g_short, masked by the block input's
            # ReLU bits, plus sum g / sum g*y3 of the previous block's last BatchNorm
            addend = torch.randn_like(x)
            y3 = torch.randn_like(x)
            bits = torch.randint(0, 256, (x.numel() // 8,), dtype=torch.uint8, device=dev)
            zsrc = torch.relu(torch.randn_like(x))

            def epi4(mask_bits):
                return lambda: hip_ops._conv_dgrad(dy, w, x.shape, 1, 0, addend=addend, mask_src=zsrc,
                                                   bn_y=y3, mask_bits=mask_bits)

            t4, t4z = timeit(epi4(bits), args.iters), timeit(epi4(None), args.iters)
            moved = dy.numel() * 2 + 3 * x.numel() * 2 + bits.numel()
            line += "\n     EPI-4 (bits) %7.1f us = %5.2f TB/s of %4.0f MB   (mask from the bf16 activation: %7.1f us)" % (
                t4 * 1e3, moved / t4 / 1e9, moved / 1e6, t4z * 1e3)
            row["epi4_us"], row["epi4_zmask_us"] = t4 * 1e3, t4z * 1e3
        if args.aten:
            xr = x.detach().requires_grad_(True)
            wr = w.detach().requires_grad_(True)

            def afwd():
                return F.conv2d(xr, wr, None, stride, pad)

            ta = timeit(afwd, args.iters)
            yr = afwd()

            def abwd():
                torch.autograd.grad(yr, (xr, wr), dy, retain_graph=True)

            tb = timeit(abwd, args.iters)
            row["aten_fwd_us"], row["aten_bwd_us"] = ta * 1e3, tb * 1e3
            tot["aten_fwd"] += ta * count
            tot["aten_bwd"] += tb * count
            line += " | aten fwd %8.1f bwd %8.1f" % (ta * 1e3, tb * 1e3)
        if args.blaslt and k == 1 and stride == 1:
            x2 = x.permute(0, 2, 3, 1).reshape(M, cin)
            w2 = w.reshape(cout, cin)
            o2 = torch.empty((M, cout), dtype=bf, device=dev)

            def lt():
                torch.matmul(x2, w2.t(), out=o2)

            tl = timeit(lt, args.iters)
            row["blaslt_us"] = tl * 1e3
            tot["blaslt"] = tot.get("blaslt", 0.0) + tl * count
            tot["fwd_1x1s1"] = tot.get("fwd_1x1s1", 0.0) + tf_ * count
            line += " | hipBLASLt gemm %7.1f us %5.0f TF/s" % (tl * 1e3, flops / tl / 1e9)
            # the weight gradient of a 1x1 conv as a plain library GEMM: dW[K, C] = dY^T[K, M] . X[M, C]
            dy2 = dy.permute(0, 2, 3, 1).reshape(M, cout)
            ow = torch.empty((cout, cin), dtype=bf, device=dev)

            def ltw():
                torch.matmul(dy2.t(), x2, out=ow)

            tlw = timeit(ltw, args.iters)
            row["blaslt_wgrad_us"] = tlw * 1e3
            tot["blaslt_wgrad"] = tot.get("blaslt_wgrad", 0.0) + tlw * count
            tot["wgrad_1x1s1"] = tot.get("wgrad_1x1s1", 0.0) + tw * count
            line += " | wgrad gemm %7.1f us" % (tlw * 1e3)
        print(line, flush=True)
        rows.append(row)
        del x, w, y, dy, dx, stats
    print("per-step totals (count-weighted, ms):", {k: round(v, 3) for k, v in tot.items()})
    if args.json:
        with open(args.json, "w") as fh:
            json.dump({"batch": B, "rows": rows, "totals_ms": tot}, fh, indent=1)


if __name__ == "__main__":
    main()

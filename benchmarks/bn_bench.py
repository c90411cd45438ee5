#!/usr/bin/env python
"""Bandwidth of the BatchNorm streaming kernels in isolation (GB/s of compulsory traffic) on the
activation shapes of ResNet-50 at batch 256; compare with benchmarks/native/stream_bench."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributeddeeplearning_amd.ops import kernels as K  # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    K.require()
    dev, bf = "cuda", torch.bfloat16
    print("%-22s %10s %10s %10s %10s %10s" % ("[M, C]", "MB", "fwd GB/s", "fwd+res", "bwd m0", "bwd m2"))
    for M, C in [(802816, 256), (802816, 64), (200704, 512), (200704, 128), (50176, 1024),
                 (50176, 256), (12544, 2048), (12544, 512)]:
        y = torch.randn(M, C, device=dev).to(bf)
        r = torch.randn(M, C, device=dev).to(bf)
        z, dy = torch.empty_like(y), torch.empty_like(y)
        sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
        fco = torch.stack([sc, sh]).contiguous()
        coef = torch.rand(3, C, device=dev)
        nb = y.numel() * 2
        t1 = timeit(lambda: K.bn_apply_fwd(y, sc, sh, None, z, M, C, True))
        t2 = timeit(lambda: K.bn_apply_fwd(y, sc, sh, r, z, M, C, True))
        t3 = timeit(lambda: K.bn_bwd_apply(r, y, None, None, coef, dy, None, M, C, 0))
        t4 = timeit(lambda: K.bn_bwd_apply(r, y, None, fco, coef, dy, None, M, C, 2))
        gb = lambda n, t: n * nb / t / 1e6  # noqa: E731
        print("%-22s %10.0f %10.0f %10.0f %10.0f %10.0f" % ("[%d, %d]" % (M, C), nb / 1e6, gb(2, t1),
                                                            gb(3, t2), gb(3, t3), gb(3, t4)))


if __name__ == "__main__":
    main()

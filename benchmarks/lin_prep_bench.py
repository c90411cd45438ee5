#!/usr/bin/env python
"""Weight preparation of the BatchNorm-by-linearity path: one launch (bn_lin_prep_q) against two
(bn_lin_prep + a GEMM of the conv kernel), per ResNet-50 stage."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributeddeeplearning_amd.ops import kernels as K  # noqa: E402

K.require()
DEV, BF = torch.device("cuda", 0), torch.bfloat16


def timeit(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for C, C2 in [(256, 64), (512, 128), (1024, 256)]:
    wt = (torch.randn((C2, C), device=DEV) / math.sqrt(C2)).to(BF)
    coef = torch.randn((3, C), device=DEV)
    b1, a2 = torch.empty((C2, C), dtype=BF, device=DEV), torch.empty((C2, C), dtype=BF, device=DEV)
    q = torch.empty((C2, C2), dtype=BF, device=DEV)
    bias = torch.empty(C2, dtype=torch.float32, device=DEV)

    def one():
        K.bn_lin_prep_q(wt, C, coef.data_ptr(), C, C2, b1, q, bias)

    def two():
        K.bn_lin_prep(wt, C, coef.data_ptr(), C, C2, b1, a2, bias)
        K.conv_igemm(a2, (C2, 1, 1, C, C, C, C, a2.numel()), wt, q, C2, 1, 1, 1, 0, 0, 1, 1, 1)

    print("C %4d C2 %3d: one launch %6.1f us, two launches %6.1f us (back-to-back issue)" % (C, C2, timeit(one), timeit(two)), flush=True)

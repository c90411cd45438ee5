#!/usr/bin/env python
"""Row-streaming 3x3 kernel: the same launch repeated under load must give the same bits (race screen)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributeddeeplearning_amd.ops import kernels as K  # noqa: E402

K.require()
DEV, BF, CL = torch.device("cuda", 0), torch.bfloat16, torch.channels_last
B, H = 256, 56
x = torch.randn((B, 64, H, 56), device=DEV).to(BF).contiguous(memory_format=CL)
w = (torch.randn((64, 64, 3, 3), device=DEV) / 24.0).to(BF).contiguous(memory_format=CL)
yp = torch.randn((B, 64, H, 56), device=DEV).to(BF).contiguous(memory_format=CL)
fco = torch.stack([torch.rand(64, device=DEV) + 0.5, torch.randn(64, device=DEV) * 0.3]).contiguous()
big = torch.empty(256 << 20, dtype=torch.uint8, device=DEV)
side = torch.cuda.Stream()
for bwd in (False, True):
    ref_y = ref_s = None
    bad_y = bad_s = 0
    for it in range(40):
        y = torch.full_like(x, float("nan"))
        st = torch.full((B, 2, 64), float("nan"), device=DEV)
        if it % 2:   # memory traffic next to it on another stream
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(4):
                    big.add_(1)
        if bwd:
            K.conv_rows3x3(x, w, y, stats=st, ep_y=yp, ep_fcoef=fco)
        else:
            K.conv_rows3x3(x, w, y, stats=st)
        torch.cuda.synchronize()
        if ref_y is None:
            ref_y, ref_s = y.clone(), st.clone()
            assert not torch.isnan(y.float()).any() and not torch.isnan(st).any()
        else:
            ny = int((y.view(torch.int16) != ref_y.view(torch.int16)).sum())
            ns = int((st != ref_s).sum())
            bad_y += ny > 0
            bad_s += ns > 0
            if ny or ns:
                d = (y.float() - ref_y.float()).abs()
                idx = torch.nonzero(d.permute(0, 2, 3, 1) > 0)[:4].tolist() if ny else []
                print("  run %d: %d output elements and %d sums differ; first (b, h, w, c): %s  max |d| %.4g" % (it, ny, ns, idx, float(d.max())))
    print("%s: %d of 39 repeats differ in the output, %d in the sums" % ("backward-data" if bwd else "forward", bad_y, bad_s), flush=True)

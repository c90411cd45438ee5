"""Phase stamps of the row-streaming weight-gradient kernel (probe library built with -DDDL_ROWS_PROBE)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributeddeeplearning_amd.ops import hip_ops, kernels as K

K.require()
dev, bf, cl = "cuda", torch.bfloat16, torch.channels_last
for (C, Kc, H) in ((64, 64, 56), (128, 128, 28)):
    B = 256
    x = torch.randn((B, C, H, H), device=dev).to(bf).contiguous(memory_format=cl)
    dy = torch.randn((B, Kc, H, H), device=dev).to(bf).contiguous(memory_format=cl)
    M = B * H * H
    splits = K.conv_wgrad_splits(M, Kc, C, 3, 3, H, H, 1, B)
    slab = Kc * 9 * C
    part = torch.zeros((splits, slab), dtype=torch.float32, device=dev)
    for _ in range(3):
        K.conv_wgrad(dy, x, hip_ops._geom(x), part, Kc, 3, 3, 1, 1, 1, H, H, splits)
    torch.cuda.synchronize()
    d = part[:, :64].view(splits, 8, 8)[:, :, :5].cpu()   # [split][wave][fetch, compute, wait, barrier, T]
    m = d.mean(0)
    print("C=%d K=%d H=%d: %d splits, steps per workgroup %.0f" % (C, Kc, H, splits, float(m[0, 4])))
    for w in range(8):
        T = float(m[w, 4])
        print("   wave %d per step: fetch %.0f  compute %.0f  dma wait %.0f  barrier %.0f  (total %.0f cycles)" %
              (w, m[w, 0] / T, m[w, 1] / T, m[w, 2] / T, m[w, 3] / T, float(m[w, :4].sum()) / T))

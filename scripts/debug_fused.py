"""GPU debug: fused BN-reduce epilogue at benchmark-sized shapes."""
import sys, os, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributeddeeplearning_amd.ops import hip_ops as hip, kernels as K
K.require()
dev, BF, CL = "cuda", torch.bfloat16, torch.channels_last
for (B, Cin, H, Cout, k, stride) in [(256, 64, 56, 64, 3, 1), (256, 64, 56, 256, 1, 1), (256, 128, 56, 128, 3, 2), (256, 512, 7, 512, 3, 1)]:
    pad = k // 2
    Ho = (H + 2 * pad - k) // stride + 1
    w = (torch.randn((Cout, Cin, k, k), device=dev) / math.sqrt(Cin * k * k)).to(BF).contiguous(memory_format=CL)
    dy = torch.randn((B, Cout, Ho, Ho), device=dev).to(BF).contiguous(memory_format=CL)
    yprev = torch.randn((B, Cin, H, H), device=dev).to(BF).contiguous(memory_format=CL)
    coeffs = torch.stack([torch.zeros(Cin, device=dev), torch.ones(Cin, device=dev), torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.3])
    gamma = torch.ones(Cin, device=dev)
    g, partial = hip._conv_dgrad_bn(dy, w, (B, Cin, H, H), stride, pad, yprev, coeffs[2:4])
    torch.cuda.synchronize()
    print("dgrad ok", partial.shape, flush=True)
    out = hip._bn_bwd(g, yprev, None, gamma, coeffs, True, False, partial=partial)
    torch.cuda.synchronize()
    s = partial.sum(0)
    gf = g.float()
    print("bn ok", float((s[0] - gf.sum(dim=(0, 2, 3))).abs().max()), float(out[1].abs().max()), flush=True)

"""Fused stem (one node) against the composition of separate ops: relative errors of every gradient,
and against an fp32 ATen reference of the same computation."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from distributeddeeplearning_amd.ops import hip_ops as hip, kernels as K

K.require()
DEV, BF, CL = "cuda", torch.bfloat16, torch.channels_last


def rel(a, b):
    a, b = a.float(), b.float()
    return float((a - b).norm() / b.norm().clamp_min(1e-12))


for (B, H, W) in [(16, 224, 224), (5, 64, 96), (3, 70, 90)]:
    torch.manual_seed(3)
    hip.reset_caches()
    x = torch.rand((B, 3, H, W), device=DEV).to(BF).contiguous(memory_format=CL)
    w0 = (torch.randn((64, 3, 7, 7), device=DEV) / math.sqrt(147)).to(BF).contiguous(memory_format=CL)
    g0, b0 = torch.rand(64, device=DEV) + 0.5, torch.randn(64, device=DEV) * 0.2
    Hc, Wc = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    Hp, Wp = (Hc - 1) // 2 + 1, (Wc - 1) // 2 + 1
    gout = torch.randn((B, 64, Hp, Wp), device=DEV).to(BF).contiguous(memory_format=CL)

    def run(mode):
        w = w0.clone().requires_grad_(True)
        gamma, beta = g0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
        rm, rv = torch.zeros(64, device=DEV), torch.ones(64, device=DEV)
        if mode == "fused":
            out = hip.stem(x, w, gamma, beta, rm, rv, 0.1, 1e-5)
        elif mode == "composed":
            y, st = hip.conv2d_stats(x, w, 2, 3, True)
            z = hip.batchnorm_act(y, gamma, beta, rm, rv, True, 0.1, 1e-5, True, None, stats=st)
            out = hip.maxpool3x3s2(z)
        else:
            wf = w0.float().requires_grad_(True)
            w = wf
            y = F.conv2d(x.float(), wf, None, 2, 3)
            z = F.relu(F.batch_norm(y, None, None, gamma, beta, True, 0.1, 1e-5))
            out = F.max_pool2d(z, 3, 2, 1)
            out.backward(gout.float())
            return out.detach(), w.grad, gamma.grad, beta.grad
        out.backward(gout)
        return out.detach(), w.grad, gamma.grad, beta.grad

    f, c, r = run("fused"), run("composed"), run("fp32")
    print("shape", (B, H, W), "out equal:", torch.equal(f[0], c[0]))
    for i, n in ((1, "dw"), (2, "dgamma"), (3, "dbeta")):
        print("  %-7s fused vs composed %.2e   fused vs fp32 %.2e   composed vs fp32 %.2e" % (n, rel(f[i], c[i]), rel(f[i], r[i]), rel(c[i], r[i])))

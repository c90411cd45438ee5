"""Per-parameter gradient error of a batch of replicated images against the batch of 8 (debug aid)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributeddeeplearning_amd.models import resnet50
from distributeddeeplearning_amd.ops import functional as ops, hip_ops
from distributeddeeplearning_amd.optim import FusedSGD
DEV, BF, CL = "cuda", torch.bfloat16, torch.channels_last
g = torch.Generator(device="cpu").manual_seed(21)
x8 = torch.rand((8, 3, 224, 224), generator=g).to(DEV).to(BF).contiguous(memory_format=CL)
y8 = torch.randint(0, 1000, (8,), generator=g).to(DEV)
ATEN = os.environ.get("DBG_ATEN") == "1"
def run(rep):
    hip_ops.reset_caches(); torch.manual_seed(0)
    if ATEN:
        ops.set_backend("aten")
        model = resnet50(num_classes=1000).to(DEV).train()
        opt = torch.optim.SGD(model.parameters(), lr=0.0)
        x = x8.float().repeat(rep, 1, 1, 1); y = y8.repeat(rep)
    else:
        model = resnet50(num_classes=1000).to_compute(torch.device(DEV), BF).train()
        opt = FusedSGD(model.named_parameters(), lr=0.0, momentum=0.0)
        x = x8.repeat(rep, 1, 1, 1).contiguous(memory_format=CL); y = y8.repeat(rep)
    opt.zero_grad(); loss = ops.softmax_cross_entropy(model(x), y); loss.backward(); torch.cuda.synchronize()
    return float(loss.detach()), {n: p.grad.float().clone() for n, p in model.named_parameters()}
l1, g1 = run(1)
for rep in [int(a) for a in sys.argv[1:]]:
    torch.cuda.empty_cache()
    l, gg = run(rep)
    rows = []
    for n in g1:
        a, b = gg[n], g1[n]
        rows.append((float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp_min(1e-20)), n, float(b.abs().max()),
                     float((a * b).sum() / (a.norm() * b.norm()).clamp_min(1e-30)), float((a * b).sum() / (b * b).sum().clamp_min(1e-30))))
    rows.sort(reverse=True)
    print("rep", rep, "loss", l, "vs", l1)
    for r in rows[:8] + [q for q in rows if "conv" in q[1]][:4]: print("   %.4f  %-32s absmax %.3e cos %.4f proj %.4f" % r)
    print("   median err %.4f" % sorted(r[0] for r in rows)[len(rows)//2], flush=True)

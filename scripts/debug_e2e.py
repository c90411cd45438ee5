import sys, torch
sys.path.insert(0, '.')
from distributeddeeplearning_amd.models import resnet50
from distributeddeeplearning_amd.ops import functional as ops
DEV='cuda'; BF=torch.bfloat16
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
S = int(sys.argv[2]) if len(sys.argv) > 2 else 64
model = resnet50(num_classes=1000).to_compute(torch.device(DEV), BF).train()
x = torch.rand((B,3,S,S), device=DEV).to(BF).contiguous(memory_format=torch.channels_last)
t = torch.randint(0,1000,(B,),device=DEV)
state = {k: v.clone() for k, v in model.state_dict().items()}
acts = {}
def mk(name, store):
    def hook(m, i, o): store[name] = o.detach().float().clone()
    return hook
def run(backend):
    store = {}
    hs = []
    for name, m in model.named_modules():
        if name.count('.') == 1 and name.startswith('layer'):
            hs.append(m.register_forward_hook(mk(name, store)))
    ops.set_backend(backend)
    model.load_state_dict(state)
    model.zero_grad(set_to_none=True)
    from distributeddeeplearning_amd.models.resnet import _cba
    s = _cba(model.conv1, model.bn1, x); store['stem'] = s.detach().float().clone()
    logits = model(x)
    loss = ops.softmax_cross_entropy(logits, t)
    loss.backward()
    for h in hs: h.remove()
    store['logits'] = logits.detach().float()
    grads = {n: p.grad.detach().float().clone() for n, p in model.named_parameters()}
    return store, float(loss), grads
def rr(a, b): return float((a-b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp_min(1e-12))
sh, lh, gh = run('auto')
sa, la, ga = run('aten')
sa2, la2, ga2 = run('aten')
print('loss hip %.4f aten %.4f aten-again %.4f' % (lh, la, la2))
for k in sh:
    print('%-12s hip-vs-aten %.4f  aten-vs-aten %.4f |x| %.3f' % (k, rr(sh[k], sa[k]), rr(sa2[k], sa[k]), float(sa[k].abs().mean())))
for n in list(gh)[:6] + ['layer2.0.conv2.weight','layer3.0.conv1.weight','layer4.2.conv3.weight','fc.weight','fc.bias']:
    print('%-28s grad rel %.4f  aten-aten %.4f' % (n, rr(gh[n], ga[n]), rr(ga2[n], ga[n])))

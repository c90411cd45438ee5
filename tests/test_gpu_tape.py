"""The launch tape (csrc/kernels/tape.h, training/tape.py): a recorded training step replayed from
native code must be the eager step bit for bit - same kernels, same arguments, same streams, same
order - and a step that is NOT recordable must be refused loudly, never replayed wrongly."""
import json
import os
import subprocess
import sys
import warnings

import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
DEV = "cuda"
BF = torch.bfloat16


def _setup(ctor, classes, batch, res, lr=0.05, seed=3, opt="sgd", **kw):
    from distributeddeeplearning_amd.data.synthetic import DeviceSyntheticLoader
    from distributeddeeplearning_amd.ops import hip_ops
    from distributeddeeplearning_amd.optim import FusedLARS, FusedSGD

    hip_ops.reset_caches()
    torch.manual_seed(seed)
    model = ctor(num_classes=classes, **kw).to_compute(torch.device(DEV), BF).train()
    if opt == "sgd":
        o = FusedSGD(model.named_parameters(), lr=lr, momentum=0.9, weight_decay=5e-5)
    else:
        o = FusedLARS(model.named_parameters(), lr=lr, momentum=0.9, weight_decay=5e-5, trust_coefficient=0.02)
    loader = DeviceSyntheticLoader(batch, length=batch * 64, pool_size=32, height=res, width=res, n_classes=classes,
                                   device=DEV, dtype=BF, seed=11)
    return model, o, loader


def _state(model, o):
    torch.cuda.synchronize()
    out = {"master." + str(k): v.clone() for k, v in o.master.items()}
    out.update({"mom." + str(k): v.clone() for k, v in o.momentum_buf.items()})
    out.update({"buf." + k: v.clone() for k, v in model.named_buffers()})
    return out


def _run(ctor, classes, batch, res, steps, taped, lr_at=None, **kw):
    from distributeddeeplearning_amd.ops import functional as ops
    from distributeddeeplearning_amd.training.tape import TapedTrainStep

    model, o, loader = _setup(ctor, classes, batch, res, **kw)
    step = TapedTrainStep(model, o, ops.softmax_cross_entropy, loader=loader, warmup=2, enabled=taped)
    losses = []
    for s in range(steps):
        if lr_at and s in lr_at:
            o.param_groups[0]["lr"] = lr_at[s]
        losses.append(step().float().clone())
    torch.cuda.synchronize()
    return step, [float(v) for v in losses], _state(model, o)


def _same(a, b):
    assert set(a) == set(b)
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_taped_resnet18_steps_equal_eager_bit_for_bit():
    from distributeddeeplearning_amd.models import resnet18

    s_e, l_e, st_e = _run(resnet18, 32, 8, 64, 9, taped=False)
    s_t, l_t, st_t = _run(resnet18, 32, 8, 64, 9, taped=True)
    assert not s_e.taped and s_t.taped, s_t.failed
    assert s_t.foreign_ops == {} and s_t.segments == 1 and s_t.launches > 100
    assert l_t == l_e, (l_t, l_e)
    _same(st_t, st_e)
    assert int(st_t["buf.bn1.num_batches_tracked"]) == 9          # the counters advanced inside the replays
    assert len(set(l_t)) == len(l_t)                                # every step drew a different batch


def test_taped_resnet50_with_learning_rate_schedule_equals_eager():
    """Bottlenecks, projection shortcuts on the second stream, BatchNorm by linearity, and a learning
    rate that moves while the tape is live (device-side scalar, optim/fused.py::freeze_lr)."""
    from distributeddeeplearning_amd.models import resnet50

    sched = {4: 0.02, 6: 0.002}
    s_e, l_e, st_e = _run(resnet50, 64, 8, 96, 8, taped=False, lr_at=sched, zero_init_residual=True)
    s_t, l_t, st_t = _run(resnet50, 64, 8, 96, 8, taped=True, lr_at=sched, zero_init_residual=True)
    assert s_t.taped, s_t.failed
    assert l_t == l_e, (l_t, l_e)
    _same(st_t, st_e)


def test_taped_lars_step_has_no_framework_operators():
    """FusedLARS: segment norms, trust ratios and the update are in-tree launches, so the large-batch recipe
    (BASELINE config 4) replays from the tape as well.  The norms are summed with fp32 atomics (order not
    fixed), so replayed and eager runs agree to rounding, not bit for bit."""
    from distributeddeeplearning_amd.models import resnet18

    s_e, l_e, _ = _run(resnet18, 32, 8, 64, 8, taped=False, opt="lars")
    s_t, l_t, _ = _run(resnet18, 32, 8, 64, 8, taped=True, opt="lars")
    assert s_t.taped and s_t.foreign_ops == {}, s_t.failed
    assert max(abs(a - b) / max(abs(b), 1e-3) for a, b in zip(l_t, l_e)) < 2e-2, (l_t, l_e)
    assert l_t[-1] < l_t[0]


def test_a_step_with_framework_operators_is_refused_not_replayed():
    """A loss function that multiplies by a tensor runs an ATen kernel in forward and one in backward
    (on autograd's thread): a replay would skip both.  The watch must see them, the step must stay
    eager and training must go on unharmed."""
    from distributeddeeplearning_amd.models import resnet18
    from distributeddeeplearning_amd.ops import functional as ops
    from distributeddeeplearning_amd.training.tape import TapedTrainStep

    def loss_fn(out, y):
        return ops.softmax_cross_entropy(out, y) * torch.ones((), device=out.device)

    model, o, loader = _setup(resnet18, 32, 8, 64)
    step = TapedTrainStep(model, o, loss_fn, loader=loader, warmup=2)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        losses = [float(step()) for _ in range(6)]
    assert not step.taped and step.failed is not None
    assert any("mul" in k for k in step.foreign_ops), step.foreign_ops
    assert any("not recordable" in str(x.message) for x in w)
    ref_model, ref_o, ref_loader = _setup(resnet18, 32, 8, 64)
    ref = []
    for _ in range(6):
        x, y = ref_loader.batch()
        ref_o.zero_grad()
        l = loss_fn(ref_model(x), y)
        l.backward()
        ref_o.step()
        ref.append(float(l))
    assert losses == ref


def test_taped_step_with_static_inputs():
    """Caller-provided batches (the real-data path): copied into the recorded input buffers."""
    from distributeddeeplearning_amd.models import resnet18
    from distributeddeeplearning_amd.ops import functional as ops
    from distributeddeeplearning_amd.training.tape import TapedTrainStep

    g = torch.Generator().manual_seed(5)
    xs = [torch.rand(8, 3, 64, 64, generator=g).to(DEV).to(BF).contiguous(memory_format=torch.channels_last) for _ in range(7)]
    ys = [torch.randint(0, 32, (8,), generator=g).to(DEV) for _ in range(7)]
    out = []
    for taped in (False, True):
        model, o, _ = _setup(resnet18, 32, 8, 64)
        step = TapedTrainStep(model, o, ops.softmax_cross_entropy, warmup=2, enabled=taped)
        losses = [float(step(xs[i], ys[i])) for i in range(7)]
        assert step.taped == taped, step.failed
        out.append((losses, _state(model, o)))
    assert out[0][0] == out[1][0]
    _same(out[0][1], out[1][1])


def test_taped_step_with_forced_collectives_runs_the_exchange_live():
    """One rank, HVD_FORCE_COLLECTIVES=1: the bucket hand-offs to the engine are live actions between
    tape segments - every replayed step launches every bucket's all-reduce again."""
    code = r'''
import os, json, torch
os.environ["HVD_FORCE_COLLECTIVES"] = "1"
os.environ["MASTER_PORT"] = str(35500 + os.getpid() % 2000)
import distributeddeeplearning_amd.torch as hvd
from distributeddeeplearning_amd.models import resnet50
from distributeddeeplearning_amd.ops import functional as ops
from distributeddeeplearning_amd.optim import FusedSGD
from distributeddeeplearning_amd.data.synthetic import DeviceSyntheticLoader
from distributeddeeplearning_amd.training.tape import TapedTrainStep
hvd.init()
c = hvd.comm()
res = {}
for taped in (False, True):
    from distributeddeeplearning_amd.ops import hip_ops
    hip_ops.reset_caches()
    torch.manual_seed(3)
    m = resnet50(num_classes=64, zero_init_residual=True).to_compute(torch.device("cuda", 0), torch.bfloat16).train()
    opt = hvd.DistributedOptimizer(FusedSGD(m.named_parameters(), lr=0.02, momentum=0.9), named_parameters=m.named_parameters())
    loader = DeviceSyntheticLoader(8, length=8 * 64, pool_size=32, height=96, width=96, n_classes=64, device="cuda", dtype=torch.bfloat16, seed=11)
    step = TapedTrainStep(m, opt, ops.softmax_cross_entropy, loader=loader, warmup=2, enabled=taped, allow_collectives=True)
    before = c.launch_counts()["rccl"]
    losses = [float(step()) for _ in range(7)]
    torch.cuda.synchronize()
    nb = sum(len(p.buckets) for p in opt.arena.plans.values())
    res[str(taped)] = {"losses": losses, "launched": c.launch_counts()["rccl"] - before, "buckets": nb, "taped": step.taped,
                       "segments": step.segments, "actions": len(step.actions), "failed": str(step.failed),
                       "sum": float(sum(v.double().sum() for v in opt.master.values())), "metrics": opt.comm_metrics()}
    del opt, m, step
print("RESULT " + json.dumps(res))
hvd.shutdown()
'''
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=420)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    rec = json.loads(out.stdout.split("RESULT ", 1)[1].splitlines()[0])
    e, t = rec["False"], rec["True"]
    print(t)
    assert t["taped"] and not e["taped"], t["failed"]
    assert e["launched"] == e["buckets"] * 7 and t["launched"] == t["buckets"] * 7
    assert t["segments"] > t["buckets"] and t["actions"] >= 2 * t["buckets"]
    assert t["losses"] == e["losses"] and t["sum"] == e["sum"]
    assert t["metrics"]["comm_exposed_ms"] is not None


def test_early_bucket_updates_are_bit_identical_to_the_plain_step():
    """One process: `DistributedOptimizer.enable_early_update()` updates every bucket as soon as its gradients
    are complete (on the weight-gradient stream, during backward).  Same parameters, momenta and losses, bit for
    bit, as updating everything in step() - eagerly and from the launch tape."""
    import distributeddeeplearning_amd.torch as hvd
    from distributeddeeplearning_amd.models import build_model
    from distributeddeeplearning_amd.ops import functional as ops
    from distributeddeeplearning_amd.optim import FusedSGD

    dev = torch.device("cuda", 0)

    def run(early, steps=4):
        torch.manual_seed(7)
        model = build_model("resnet50", num_classes=1000).to_compute(dev, torch.bfloat16).train()
        base = FusedSGD(model.named_parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4)
        opt = hvd.DistributedOptimizer(base, named_parameters=model.named_parameters())
        assert opt.enable_early_update(early) == early
        g = torch.Generator(device=dev).manual_seed(3)
        x = torch.randn((16, 3, 224, 224), generator=g, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        y = torch.randint(0, 1000, (16,), generator=g, device=dev)
        losses = []
        for _ in range(steps):
            opt.zero_grad()
            loss = ops.softmax_cross_entropy(model(x), y)
            loss.backward()
            opt.step()
            losses.append(float(loss))
        torch.cuda.synchronize()
        arena = opt.arena
        return losses, {str(dt): arena.params[dt].clone() for dt in arena.params}, \
            {str(dt): opt.momentum_buf[dt].clone() for dt in opt.momentum_buf}

    l0, p0, m0 = run(False)
    l1, p1, m1 = run(True)
    assert l0 == l1, (l0, l1)
    for k in p0:
        assert torch.equal(p0[k], p1[k]), k
    for k in m0:
        assert torch.equal(m0[k], m1[k]), k

"""Bucket planner, gradient arenas, fused optimizers, LR schedules, callbacks (CPU)."""
import math

import numpy as np
import pytest
import torch

from distributeddeeplearning_amd import callbacks as cb
from distributeddeeplearning_amd.models import resnet50
from distributeddeeplearning_amd.optim import (FusedLARS, FusedSGD, LRSchedule, goyal_lr, poly_lr,
                                               scaled_lr, staircase_multiplier)
from distributeddeeplearning_amd.parallel.bucket import (BUCKET_ALIGN_BYTES, GradArena,
                                                         SEG_ALIGN_BYTES, plan_arenas)
from distributeddeeplearning_amd.parallel.compression import Compression


def test_plan_reverse_order_caps_and_alignment():
    entries = [("a", (1000,), torch.float32), ("b", (3000,), torch.float32),
               ("c", (10,), torch.bfloat16), ("d", (5000,), torch.float32)]
    plans = plan_arenas(entries, first_bucket_bytes=4096 * 4, bucket_cap_bytes=4096 * 4)
    p32 = plans[torch.float32]
    assert [s.name for s in p32.segments] == ["d", "b", "a"]  # reverse registration order
    assert [len(b.segments) for b in p32.buckets] == [1, 2]   # d alone (over the cap), then b+a
    for s in p32.segments:
        assert (s.offset * 4) % SEG_ALIGN_BYTES == 0
    for b in p32.buckets:
        assert (b.end * 4) % BUCKET_ALIGN_BYTES == 0
    assert torch.bfloat16 in plans and plans[torch.bfloat16].segments[0].name == "c"


def test_resnet50_bucket_plan_is_small_and_ordered():
    m = resnet50()
    arena = GradArena(list(m.named_parameters()), 8 << 20, 32 << 20)
    buckets = arena.all_buckets()
    assert 3 <= len(buckets) <= 8  # 4-6 buckets for ResNet-50
    first = buckets[0][1].segments[0].name
    assert first.startswith("fc.")  # backward produces fc first
    assert buckets[-1][1].segments[-1].name == "conv1.weight"
    for _, p in m.named_parameters():
        assert p.grad is not None and p.grad.shape == p.shape
    total = sum(b.numel for _, b in buckets)
    assert total >= 25557032


def test_grad_views_survive_backward_and_zero():
    lin = torch.nn.Linear(8, 4)
    arena = GradArena(list(lin.named_parameters()), 1 << 20, 1 << 20)
    ptrs = {n: p.grad.data_ptr() for n, p in lin.named_parameters()}
    lin(torch.randn(3, 8)).sum().backward()
    assert all(p.grad.data_ptr() == ptrs[n] for n, p in lin.named_parameters())
    assert float(arena.grads[torch.float32].abs().sum()) > 0
    arena.zero_()
    assert float(lin.weight.grad.abs().sum()) == 0.0
    lin.weight.grad = None
    assert arena.ensure_view(lin.weight).data_ptr() == ptrs["weight"]


def test_duplicate_names_rejected():
    p = torch.nn.Parameter(torch.zeros(3))
    q = torch.nn.Parameter(torch.zeros(3))
    with pytest.raises(ValueError):
        GradArena([("w", p), ("w", q)], 1024, 1024)


@pytest.mark.parametrize("nesterov,wd", [(False, 0.0), (True, 1e-2), (False, 5e-5)])
def test_fused_sgd_equals_torch_sgd(nesterov, wd):
    torch.manual_seed(0)
    a, b = torch.nn.Linear(16, 8), torch.nn.Linear(16, 8)
    b.load_state_dict(a.state_dict())
    fo = FusedSGD(a.named_parameters(), lr=0.1, momentum=0.9, weight_decay=wd, nesterov=nesterov,
                  weight_decay_filter=lambda n, p: True)
    to = torch.optim.SGD(b.parameters(), lr=0.1, momentum=0.9, weight_decay=wd, nesterov=nesterov)
    for _ in range(4):
        x = torch.randn(5, 16)
        fo.zero_grad(); to.zero_grad()
        a(x).pow(2).sum().backward(); b(x).pow(2).sum().backward()
        fo.step(); to.step()
    assert torch.allclose(a.weight, b.weight, atol=1e-6) and torch.allclose(a.bias, b.bias, atol=1e-6)


def test_fused_sgd_default_filter_skips_1d():
    lin = torch.nn.Linear(4, 4)
    opt = FusedSGD(lin.named_parameters(), lr=1.0, momentum=0.0, weight_decay=0.5)
    w0, b0 = lin.weight.detach().clone(), lin.bias.detach().clone()
    opt.zero_grad()
    opt.step()  # zero gradients: only weight decay acts, and only on the 2-D kernel
    assert torch.allclose(lin.weight, w0 * 0.5) and torch.equal(lin.bias, b0)


def test_lars_matches_numpy_reference():
    torch.manual_seed(1)
    lin = torch.nn.Linear(6, 3)
    w0, b0 = lin.weight.detach().numpy().copy(), lin.bias.detach().numpy().copy()
    opt = FusedLARS(lin.named_parameters(), lr=0.5, momentum=0.9, weight_decay=1e-2,
                    trust_coefficient=0.001)
    x = torch.randn(4, 6)
    opt.zero_grad()
    lin(x).pow(2).sum().backward()
    gw, gb = lin.weight.grad.numpy().copy(), lin.bias.grad.numpy().copy()
    opt.step()
    wn, gn = np.linalg.norm(w0), np.linalg.norm(gw)
    ratio = 0.001 * wn / (gn + 1e-2 * wn + 1e-8)
    v = 0.5 * ratio * (gw + 1e-2 * w0)
    assert np.allclose(lin.weight.detach().numpy(), w0 - v, atol=1e-6)
    assert np.allclose(lin.bias.detach().numpy(), b0 - 0.5 * gb, atol=1e-6)  # 1-D: plain SGD


def test_optimizer_state_dict_roundtrip():
    lin = torch.nn.Linear(4, 2)
    opt = FusedSGD(lin.named_parameters(), lr=0.1, momentum=0.9)
    opt.zero_grad(); lin(torch.randn(3, 4)).sum().backward(); opt.step()
    state = opt.state_dict()
    lin2 = torch.nn.Linear(4, 2)
    opt2 = FusedSGD(lin2.named_parameters(), lr=0.3, momentum=0.9)
    opt2.load_state_dict(state)
    assert opt2.param_groups[0]["lr"] == 0.1
    for dt in opt.momentum_buf:
        assert torch.equal(opt.momentum_buf[dt], opt2.momentum_buf[dt])


def test_lr_math():
    assert scaled_lr(0.001, 8) == 0.008
    assert staircase_multiplier(29) == 1.0 and staircase_multiplier(30) == pytest.approx(0.1)
    assert staircase_multiplier(85) == pytest.approx(1e-3)
    # Goyal warm-up: lr/size at epoch 0 -> lr at warmup end
    assert goyal_lr(0.0, 0.1, 8, warmup_epochs=5) < goyal_lr(5.0, 0.1, 8, warmup_epochs=5)
    assert goyal_lr(45.0, 0.1, 8) == pytest.approx(goyal_lr(10.0, 0.1, 8) * 0.1)
    assert cb.warmup_multiplier(0.0, 5, 8) == pytest.approx(1 / 8)
    assert cb.warmup_multiplier(5.0, 5, 8) == 1.0
    assert cb.warmup_multiplier(2.5, 5, 8) == pytest.approx(1 / 8 * (2.5 * 7 / 5 + 1))
    assert poly_lr(0, 100, 1.0, warmup_steps=10) < poly_lr(10, 100, 1.0, warmup_steps=10)
    assert poly_lr(100, 100, 1.0) == pytest.approx(0.0)


def test_warmup_and_schedule_callbacks_drive_optimizer():
    lin = torch.nn.Linear(2, 2)
    opt = torch.optim.SGD(lin.parameters(), lr=0.8, momentum=0.9)
    cbs = cb.CallbackList([cb.LearningRateWarmupCallback(warmup_epochs=2, size=8),
                           cb.LearningRateScheduleCallback(1.0, start_epoch=2, end_epoch=4),
                           cb.LearningRateScheduleCallback(0.1, start_epoch=4)],
                          lin, opt, steps_per_epoch=4)
    cbs.on_train_begin({})
    seen = []
    for epoch in range(6):
        cbs.on_epoch_begin(epoch, {})
        for b in range(4):
            cbs.on_batch_begin(b, {})
            seen.append(opt.param_groups[0]["lr"])
            cbs.on_batch_end(b, {})
            assert opt.param_groups[0]["momentum"] == 0.9  # momentum correction is per step
        cbs.on_epoch_end(epoch, {})
    assert seen[0] == pytest.approx(0.8 / 8 * (0.25 * 7 / 2 + 1))
    assert seen[7] == pytest.approx(0.8)            # last warm-up batch lands on the target
    assert seen[8] == pytest.approx(0.8) and seen[16] == pytest.approx(0.08)
    assert all(b >= a - 1e-12 for a, b in zip(seen[:8], seen[1:8]))


def test_default_imagenet_schedule_shape():
    sched = cb.default_imagenet_schedule()
    assert len(sched) == 5 and isinstance(sched[0], cb.LearningRateWarmupCallback)


def test_compression_roundtrip():
    t = torch.randn(100)
    for comp in (Compression.none, Compression.fp16, Compression.bf16):
        wire, ctx = comp.compress(t)
        back = comp.decompress(wire, ctx)
        assert back.dtype == t.dtype and torch.allclose(back, t, atol=2e-2)
    w, _ = Compression.fp16.compress(t)
    assert w.dtype == torch.float16


def test_tail_bucket_keeps_the_last_exchange_small():
    """The first layers' gradients arrive last; with ``tail_bucket_bytes`` they get a small bucket
    of their own so that the large bucket before it does not wait for the end of backward."""
    from distributeddeeplearning_amd.parallel.bucket import plan_arenas

    m = resnet50()
    ents = [(n, tuple(p.shape), torch.float32) for n, p in m.named_parameters() if p.dim() > 1]
    base = plan_arenas(ents, 8 << 20, 32 << 20)[torch.float32]
    tail = plan_arenas(ents, 8 << 20, 32 << 20, tail_bucket_bytes=256 << 10)[torch.float32]
    assert len(tail.buckets) == len(base.buckets) + 1
    last = tail.buckets[-1]
    assert last.segments[-1].name == "conv1.weight"
    assert (last.end - last.start) * 4 <= (256 << 10) + 4096
    assert [s.name for b in tail.buckets for s in b.segments] == [s.name for b in base.buckets for s in b.segments]
    # every segment still inside its bucket, buckets contiguous and aligned
    for b in tail.buckets:
        assert all(b.start <= s.offset and s.offset + s.numel <= b.end for s in b.segments)
    assert all(a.end == b.start for a, b in zip(tail.buckets, tail.buckets[1:]))


def test_direct_gradients_are_counted_once_per_backward():
    """A fused block writes a gradient into the arena itself, reports it through the ops layer's
    listener and returns None to autograd - and autograd STILL runs the parameter's
    post-accumulate hook (torch >= 2.x).  Counting both launched every bucket after half of its
    gradients had arrived (found on the GPU by test_bucket_is_complete_when_its_collective_runs)."""
    import distributeddeeplearning_amd.torch as hvd
    from distributeddeeplearning_amd.ops import hip_ops

    hvd.init()
    w1 = torch.nn.Parameter(torch.ones(4))
    w2 = torch.nn.Parameter(torch.ones(4))

    class Direct(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w):
            ctx.w = w
            return x * w.detach()

        @staticmethod
        def backward(ctx, g):
            ctx.w.grad.add_(g)          # "kernel" writes the arena view
            hip_ops._notify(ctx.w)      # ... and reports readiness
            return g * ctx.w.detach(), None

    opt = hvd.DistributedOptimizer(torch.optim.SGD([w1, w2], lr=0.1), named_parameters=[("w1", w1), ("w2", w2)])
    assert len(opt._ddl_buckets) == 1
    seen = {}
    x = torch.ones(4, requires_grad=True)
    h = Direct.apply(x, w1)
    def between(g):
        seen["launched_after_w2_only"] = opt._ddl_buckets[0].launched

    h.register_hook(between)
    opt.zero_grad()
    Direct.apply(h, w2).sum().backward()
    assert seen["launched_after_w2_only"] is False
    assert opt._ddl_buckets[0].launched and opt._ddl_buckets[0].pending == 0
    opt.step()
    assert torch.allclose(w1.detach(), torch.full((4,), 0.9)) and torch.allclose(w2.detach(), torch.full((4,), 0.9))


def test_freeze_lr_at_zero_keeps_following_the_schedule():
    """A warm-up starting at lr = 0: the frozen by-value factor must not be 0."""
    from distributeddeeplearning_amd.optim import FusedSGD

    m = torch.nn.Linear(4, 3)
    opt = FusedSGD(m.named_parameters(), lr=0.0, momentum=0.9)
    scale = opt.freeze_lr()
    assert float(scale) == 0.0 and opt._kernel_lr(0.0)[0] == 1.0
    opt.param_groups[0]["lr"] = 0.25
    opt.sync_lr()
    assert float(opt.lr_scale) == 0.25
    frozen, dev = opt._kernel_lr(0.25)
    assert frozen * float(dev) == 0.25


def test_graph_step_runs_eagerly_when_momentum_is_corrected():
    from distributeddeeplearning_amd.training.graph import GraphedTrainStep
    from distributeddeeplearning_amd.optim import FusedSGD

    m = torch.nn.Linear(4, 3)
    opt = FusedSGD(m.named_parameters(), lr=0.1, momentum=0.9)
    gs = GraphedTrainStep(m, opt, lambda o, y: o.sum())
    before = gs._hyper()
    opt.param_groups[0]["momentum"] = 0.45   # what LearningRateScheduleCallback does for one step
    assert gs._hyper() != before
    opt.param_groups[0]["momentum"] = 0.9
    opt.param_groups[0]["lr"] = 0.5           # the learning rate alone does not force eager steps
    assert gs._hyper() == before


def test_comm_dtype_fp32_keeps_gradient_arenas_in_fp32(monkeypatch):
    """HVD_COMM_DTYPE=fp32: bf16 parameters, fp32 gradient arena (= all-reduce buffer and the
    accumulator of backward_passes_per_step), fp32 masters; the update equals fp32 SGD."""
    from distributeddeeplearning_amd.optim import FusedSGD

    monkeypatch.setenv("HVD_COMM_DTYPE", "fp32")
    torch.manual_seed(3)
    lin = torch.nn.Linear(16, 8)
    lin.weight.data = lin.weight.data.to(torch.bfloat16)
    opt = FusedSGD(lin.named_parameters(), lr=0.1, momentum=0.9)
    assert opt.arena.grads[torch.bfloat16].dtype == torch.float32
    assert opt.arena.params[torch.bfloat16].dtype == torch.bfloat16
    w0 = lin.weight.detach().float().clone()
    g = torch.randn(8, 16) * 1e-3          # below bf16 resolution of the weights: must not be lost
    opt.zero_grad()
    opt.arena.ensure_view(lin.weight).copy_(g)
    assert lin.weight.grad.dtype == torch.float32
    opt.step()
    seg = opt.arena.plans[torch.bfloat16].segments[0]
    master = opt.master[torch.bfloat16][seg.offset: seg.offset + seg.numel].view(8, 16)
    assert torch.allclose(master, w0 - 0.1 * g, atol=1e-7)


def test_bucket_plan_resnet152_tensors_reverse_order_caps_and_tail():
    """BASELINE config 5's model: 467 gradient tensors (155 conv + 310 BN + 2 fc),
    planned in reverse registration order, every bucket within the byte caps, the tail bucket (the
    stem, ready only when backward ends) at most 256 KiB."""
    from distributeddeeplearning_amd.models import resnet152
    from distributeddeeplearning_amd.parallel.bucket import GradArena, arena_options
    from distributeddeeplearning_amd.utils.config import EngineConfig

    model = resnet152(num_classes=1000)
    for m in model.modules():   # the compute layout: bf16 conv / fc weights, fp32 BatchNorm and bias
        w = getattr(m, "weight", None)
        if w is not None and w.dim() > 1:
            w.data = w.data.to(torch.bfloat16)
    named = list(model.named_parameters())
    assert len(named) == 467 and sum(p.numel() for _, p in named) == 60192808
    cfg = EngineConfig()
    arena = GradArena(named, **arena_options(cfg))
    order = {n: i for i, (n, _) in enumerate(named)}
    for dt, plan in arena.plans.items():
        es = torch.empty((), dtype=dt).element_size()
        idx = [order[s.name] for s in plan.segments]
        assert idx == sorted(idx, reverse=True)                 # reverse registration = backward order
        assert [s.name for b in plan.buckets for s in b.segments] == [s.name for s in plan.segments]
        biggest_tensor = max(s.numel for s in plan.segments) * es
        for k, b in enumerate(plan.buckets):
            nbytes = (b.end - b.start) * es
            cap = cfg.first_bucket_bytes if k == 0 else cfg.fusion_threshold
            assert nbytes <= max(cap, biggest_tensor), (dt, k, nbytes)
            assert b.start % 8 == 0                             # 16-byte aligned for either dtype
        tail = plan.buckets[-1]
        if plan.numel * es > 4 * cfg.tail_bucket_bytes:
            # the stem's gradients (ready only when backward ends) travel in a bucket of their own
            assert (tail.end - tail.start) * es <= cfg.tail_bucket_bytes, (dt, (tail.end - tail.start) * es)
            assert tail.segments[-1].name == "conv1.weight"
        else:
            # an arena below 1 MiB (the fp32 BatchNorm / bias arena: 610 KB here) is ONE latency-bound
            # bucket: splitting it would add a collective launch (~65 us, profiles/r1_notes.md) to hide ~2 us
            assert len(plan.buckets) == 1
        assert plan.buckets[0].segments[0].name.startswith("fc.")
    nb = {str(dt): len(plan.buckets) for dt, plan in arena.plans.items()}
    assert nb["torch.bfloat16"] >= 5 and nb["torch.float32"] == 1, nb

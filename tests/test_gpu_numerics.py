"""Correctness evidence at PRODUCTION shapes:

* every convolution shape class of ResNet-50 (22 body shapes + the stem) at batch 256 -
  the tile rule, split-K plan and parity classes the flagship bench actually runs - forward,
  backward-data and backward-weights against fp32 ``F.conv2d``;
* loss trajectories: HIP bf16 training against the ATen fp32 oracle on a fixed image set;
* the bf16-parameter / fp32-master / bf16-gradient path of FusedSGD and FusedLARS against plain
  PyTorch fp32 implementations over 10 steps;
* the one-code-path rule: a GPU tensor outside the HIP kernels' envelope raises.
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"
BF = torch.bfloat16
CL = torch.channels_last


def _hip():
    from distributeddeeplearning_amd.ops import hip_ops, kernels

    kernels.require()
    return hip_ops


def relerr(a, ref):
    a, ref = a.float(), ref.float()
    return float((a - ref).abs().max() / ref.abs().max().clamp_min(1e-6))


def rms_relerr(a, ref):
    a, ref = a.float(), ref.float()
    return float((a - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt().clamp_min(1e-12))


# (name, Cin, H, Cout, k, stride) of SURVEY 2.7; padding = k // 2
RESNET50_SHAPES = [
    ("a1", 64, 56, 64, 1, 1), ("a2", 256, 56, 64, 1, 1), ("a3", 64, 56, 64, 3, 1), ("a4", 64, 56, 256, 1, 1),
    ("b1", 256, 56, 128, 1, 1), ("b2", 128, 56, 128, 3, 2), ("b3", 512, 28, 128, 1, 1), ("b4", 128, 28, 128, 3, 1),
    ("b5", 128, 28, 512, 1, 1), ("b6", 256, 56, 512, 1, 2),
    ("c1", 512, 28, 256, 1, 1), ("c2", 256, 28, 256, 3, 2), ("c3", 1024, 14, 256, 1, 1), ("c4", 256, 14, 256, 3, 1),
    ("c5", 256, 14, 1024, 1, 1), ("c6", 512, 28, 1024, 1, 2),
    ("d1", 1024, 14, 512, 1, 1), ("d2", 512, 14, 512, 3, 2), ("d3", 2048, 7, 512, 1, 1), ("d4", 512, 7, 512, 3, 1),
    ("d5", 512, 7, 2048, 1, 1), ("d6", 1024, 14, 2048, 1, 2),
]


@pytest.mark.parametrize("shape", RESNET50_SHAPES, ids=[s[0] for s in RESNET50_SHAPES])
def test_conv_resnet50_shape_at_batch_256(shape):
    """The HIP kernels at B = 256.  Forward and backward-data are compared on sampled images (a
    convolution is independent per image), the weight gradient - a reduction over ALL 256 images -
    in full against fp32 autograd accumulated over chunks."""
    hip = _hip()
    _, Cin, H, Cout, k, stride = shape
    B, pad = 256, k // 2
    g = torch.Generator(device=DEV).manual_seed(hash(shape[0]) % 1000)
    x = torch.randn((B, Cin, H, H), generator=g, device=DEV).to(BF).contiguous(memory_format=CL).requires_grad_(True)
    w = (torch.randn((Cout, Cin, k, k), generator=g, device=DEV) / math.sqrt(Cin * k * k)).to(BF)
    w = w.contiguous(memory_format=CL).requires_grad_(True)
    y, stats = hip.conv2d_stats(x, w, stride, pad, True)
    gy = torch.randn(y.shape, generator=g, device=DEV).to(BF).contiguous(memory_format=CL)
    y.backward(gy)
    torch.cuda.synchronize()
    # fused BatchNorm statistics of the epilogue = column sums of the bf16 output
    y2 = y.detach().float().permute(0, 2, 3, 1).reshape(-1, Cout)
    assert torch.allclose(stats[:, 0, :].sum(0), y2.sum(0), rtol=2e-3, atol=2e-2 * float(y2.abs().max()))
    assert torch.allclose(stats[:, 1, :].sum(0), (y2 * y2).sum(0), rtol=2e-3, atol=1e-2)
    del y2
    wr = w.detach().float().contiguous().requires_grad_(True)
    sample = [0, 1, 100, 127, 128, 255]
    xs = x.detach()[sample].float().contiguous().requires_grad_(True)
    ys = F.conv2d(xs, wr, None, stride, pad)
    assert relerr(y.detach()[sample], ys) < 1.5e-2 and rms_relerr(y.detach()[sample], ys) < 4e-3
    (dxs,) = torch.autograd.grad(ys, xs, gy[sample].float())
    assert relerr(x.grad[sample], dxs) < 1.5e-2 and rms_relerr(x.grad[sample], dxs) < 4e-3
    # full weight gradient, fp32, 32 images at a time
    dw = torch.zeros_like(wr)
    for i in range(0, B, 32):
        xc = x.detach()[i:i + 32].float().contiguous()
        dw += torch.autograd.grad(F.conv2d(xc, wr, None, stride, pad), wr, gy[i:i + 32].float())[0]
    assert relerr(w.grad, dw) < 1.5e-2 and rms_relerr(w.grad, dw) < 4e-3, (relerr(w.grad, dw), rms_relerr(w.grad, dw))


def test_stem_conv_at_batch_256():
    hip = _hip()
    B = 256
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.rand((B, 3, 224, 224), generator=g, device=DEV).to(BF).contiguous(memory_format=CL)
    w = (torch.randn((64, 3, 7, 7), generator=g, device=DEV) / math.sqrt(147)).to(BF)
    w = w.contiguous(memory_format=CL).requires_grad_(True)
    y, _ = hip.conv2d_stats(x, w, 2, 3, True)
    gy = torch.randn(y.shape, generator=g, device=DEV).to(BF).contiguous(memory_format=CL)
    y.backward(gy)
    wr = w.detach().float().contiguous().requires_grad_(True)
    sample = [0, 77, 255]
    ys = F.conv2d(x[sample].float().contiguous(), wr, None, 2, 3)
    assert relerr(y.detach()[sample], ys) < 1.5e-2
    dw = torch.zeros_like(wr)
    for i in range(0, B, 32):
        dw += torch.autograd.grad(F.conv2d(x[i:i + 32].float().contiguous(), wr, None, 2, 3), wr, gy[i:i + 32].float())[0]
    assert relerr(w.grad, dw) < 1.5e-2 and rms_relerr(w.grad, dw) < 4e-3


# ---- loss trajectories --------------------------------------------------------------------------
def _trajectory(ctor, steps, images, res, classes, hip_path, lr=0.02, batch=32, **ctor_kw):
    from distributeddeeplearning_amd.ops import functional as ops
    from distributeddeeplearning_amd.optim import FusedSGD

    g = torch.Generator(device="cpu").manual_seed(11)
    X = torch.rand((images, 3, res, res), generator=g)
    Y = torch.randint(0, classes, (images,), generator=g)
    torch.manual_seed(0)
    model = ctor(num_classes=classes, **ctor_kw)
    if hip_path:
        _hip().reset_caches()
        model = model.to_compute(torch.device(DEV), BF).train()
        opt = FusedSGD(model.named_parameters(), lr=lr, momentum=0.9, weight_decay=5e-5)
        X = X.to(DEV).to(BF).contiguous(memory_format=CL)
    else:
        model = model.to(DEV).train()
        opt = torch.optim.SGD([{"params": [p for p in model.parameters() if p.dim() > 1], "weight_decay": 5e-5},
                               {"params": [p for p in model.parameters() if p.dim() <= 1], "weight_decay": 0.0}],
                              lr=lr, momentum=0.9)
        X = X.to(DEV)
    Y = Y.to(DEV)
    prev = ops.backend()
    ops.set_backend("auto" if hip_path else "aten")
    losses = []
    try:
        for s in range(steps):
            i = (s * batch) % images
            opt.zero_grad()
            loss = ops.softmax_cross_entropy(model(X[i:i + batch]), Y[i:i + batch])
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
    finally:
        ops.set_backend(prev)
    return losses


def _compare_trajectories(hip, ref, band, final_tol):
    assert all(math.isfinite(v) for v in hip), hip
    # per-step band: relative above a loss of 0.5, absolute (band * 0.5) below it
    worst = max(abs(a - b) / max(abs(b), 0.5) for a, b in zip(hip, ref))
    assert worst < band, (worst, hip, ref)
    tail_h, tail_r = sum(hip[-4:]) / 4, sum(ref[-4:]) / 4
    assert abs(tail_h - tail_r) / max(tail_r, 0.05) < final_tol, (tail_h, tail_r)
    assert tail_r < 0.8 * ref[0] and tail_h < 0.8 * hip[0], (hip, ref)   # both runs learn


def test_loss_trajectory_resnet18_hip_bf16_vs_aten_fp32():
    """40 SGD steps over a fixed 64-image set (2 batches of 32, 20 epochs): every per-step loss of
    the HIP bf16 path within 15 % of the fp32 ATen run, the mean of the last four within 5 %."""
    from distributeddeeplearning_amd.models import resnet18

    ref = _trajectory(resnet18, 40, 64, 64, 16, hip_path=False)
    hip = _trajectory(resnet18, 40, 64, 64, 16, hip_path=True)
    print("resnet18 fp32:", [round(v, 3) for v in ref])
    print("resnet18 hip :", [round(v, 3) for v in hip])
    _compare_trajectories(hip, ref, band=0.15, final_tol=0.05)


def test_loss_trajectory_resnet50_20_steps_hip_bf16_vs_aten_fp32():
    from distributeddeeplearning_amd.models import resnet50

    # zero-initialised last BatchNorm of every block (the usual large-batch recipe): without it
    # 20 steps of momentum SGD on 64 random-label images do not converge in fp32 either, and two
    # diverging runs cannot be compared
    ref = _trajectory(resnet50, 20, 64, 96, 16, hip_path=False, lr=0.02, zero_init_residual=True)
    hip = _trajectory(resnet50, 20, 64, 96, 16, hip_path=True, lr=0.02, zero_init_residual=True)
    print("resnet50 fp32:", [round(v, 3) for v in ref])
    print("resnet50 hip :", [round(v, 3) for v in hip])
    _compare_trajectories(hip, ref, band=0.20, final_tol=0.05)


# ---- the large-batch recipe end to end: LARS + gradual warm-up + momentum correction -----------------
class _RefLARS(torch.optim.Optimizer):
    """Independent fp32 implementation of the rule in optim/fused.py's docstring (You et al. 2017):
    tensors with ndim > 1 are layer-wise adaptive with weight decay, 1-D tensors take plain momentum SGD."""

    def __init__(self, params, lr, momentum, weight_decay, eta, eps=1e-8):
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay, eta=eta, eps=eps))

    @torch.no_grad()
    def step(self):
        for g in self.param_groups:
            lr, mu, wd, eta, eps = g["lr"], g["momentum"], g["weight_decay"], g["eta"], g["eps"]
            for p in g["params"]:
                v = self.state[p].setdefault("v", torch.zeros_like(p))
                if p.dim() > 1:
                    wn, gn = p.norm(), p.grad.norm()
                    lam = eta * wn / (gn + wd * wn + eps) if (wn > 0 and gn > 0) else 1.0
                    v.mul_(mu).add_((p.grad + wd * p) * (lr * lam))
                    p.sub_(v)
                else:
                    v.mul_(mu).add_(p.grad)
                    p.sub_(v, alpha=lr)


def _lars_warmup_trajectory(hip_path, steps=40, lr=0.03, eta=0.01, warm=2, spe=10, size=8):
    from distributeddeeplearning_amd.callbacks import LearningRateWarmupCallback
    from distributeddeeplearning_amd.models import resnet18
    from distributeddeeplearning_amd.ops import functional as ops
    from distributeddeeplearning_amd.optim import FusedLARS

    g = torch.Generator(device="cpu").manual_seed(11)
    X = torch.rand((64, 3, 64, 64), generator=g)
    Y = torch.randint(0, 16, (64,), generator=g).to(DEV)
    torch.manual_seed(0)
    model = resnet18(num_classes=16)
    if hip_path:
        _hip().reset_caches()
        model = model.to_compute(torch.device(DEV), BF).train()
        opt = FusedLARS(model.named_parameters(), lr=lr, momentum=0.9, weight_decay=5e-5, trust_coefficient=eta)
        X = X.to(DEV).to(BF).contiguous(memory_format=CL)
    else:
        model = model.to(DEV).train()
        opt = _RefLARS(model.parameters(), lr=lr, momentum=0.9, weight_decay=5e-5, eta=eta)
        X = X.to(DEV)
    # the reference's schedule head (imagenet_keras_horovod.py:211-213): lr/size -> lr over `warm`
    # epochs, per batch, with Goyal et al.'s momentum correction at every change of the rate
    cb = LearningRateWarmupCallback(warmup_epochs=warm, steps_per_epoch=spe, size=size)
    cb.bind(optimizer=opt, steps_per_epoch=spe)
    cb.on_train_begin()
    prev = ops.backend()
    ops.set_backend("auto" if hip_path else "aten")
    losses, lrs, mus = [], [], []
    try:
        for s in range(steps):
            e, b = divmod(s, spe)
            if b == 0:
                cb.on_epoch_begin(e)
            cb.on_batch_begin(b)
            lrs.append(opt.param_groups[0]["lr"])
            mus.append(opt.param_groups[0]["momentum"])
            i = (s * 32) % 64
            opt.zero_grad()
            loss = ops.softmax_cross_entropy(model(X[i:i + 32]), Y[i:i + 32])
            loss.backward()
            opt.step()
            cb.on_batch_end(b)
            losses.append(float(loss.detach()))
    finally:
        ops.set_backend(prev)
    return losses, lrs, mus


def test_lars_with_warmup_and_momentum_correction_trajectory_hip_bf16_vs_fp32():
    """BASELINE config 4's recipe end to end: 40 ResNet-18 steps with HIP-bf16
    FusedLARS + LearningRateWarmupCallback (8 simulated ranks: 0.03/8 -> 0.03 over 20 steps, momentum
    corrected at every change) against an fp32 PyTorch implementation of the same rule under the same
    schedule.  fp32 reference on CPU (tuned there): 2.801 2.841 2.766 ... 0.192 0.182 0.154."""
    ref, lr_r, mu_r = _lars_warmup_trajectory(False)
    hip, lr_h, mu_h = _lars_warmup_trajectory(True)
    print("lars fp32:", [round(v, 3) for v in ref])
    print("lars hip :", [round(v, 3) for v in hip])
    assert lr_r == lr_h and mu_r == mu_h                       # one schedule drove both optimizers
    assert abs(lr_h[0] - 0.03 / 8 * (1 + 0.1 * 7 / 2)) < 1e-9 and lr_h[19] == pytest.approx(0.03) and lr_h[-1] == pytest.approx(0.03)
    assert all(m > 0.9 for m in mu_h[1:20]) and all(m == 0.9 for m in mu_h[20:])   # corrected while the rate moves
    # measured on MI355X: every step within 4.3 % of the fp32 run, the mean of the last four within 5.0 %
    # (0.1975 against 0.1881: the bf16 run trails the fp32 one by about a third of a step from the second
    # update on, like the SGD trajectories above, and at a loss of 0.2 that is 5 %)
    _compare_trajectories(hip, ref, band=0.15, final_tol=0.08)


# ---- optimizers: bf16 parameters, fp32 masters, bf16 gradients ------------------------------------
def _mixed_params(seed=4):
    g = torch.Generator(device=DEV).manual_seed(seed)
    shapes = [("conv1.weight", (64, 3, 7, 7)), ("bn1.weight", (64,)), ("bn1.bias", (64,)),
              ("layer.conv.weight", (128, 64, 3, 3)), ("layer.bn.weight", (128,)), ("fc.weight", (100, 128)),
              ("fc.bias", (100,))]
    named = []
    for n, s in shapes:
        dt = BF if len(s) > 1 else torch.float32
        p = (torch.randn(s, generator=g, device=DEV) * 0.1).to(dt)
        named.append((n, torch.nn.Parameter(p)))
    return named


def _grads(named, step):
    g = torch.Generator(device=DEV).manual_seed(100 + step)
    return [(torch.randn(p.shape, generator=g, device=DEV) * 0.05).to(p.dtype) for _, p in named]


def test_fused_sgd_bf16_params_fp32_master_matches_torch_sgd_over_10_steps():
    from distributeddeeplearning_amd.optim import FusedSGD

    named = _mixed_params()
    ref = [p.detach().float().clone().requires_grad_(True) for _, p in named]
    opt = FusedSGD(named, lr=0.05, momentum=0.9, weight_decay=1e-3)
    opt_r = torch.optim.SGD([{"params": [r for r in ref if r.dim() > 1], "weight_decay": 1e-3},
                             {"params": [r for r in ref if r.dim() <= 1], "weight_decay": 0.0}], lr=0.05, momentum=0.9)
    for step in range(10):
        opt.zero_grad()
        for (_, p), gr, r in zip(named, _grads(named, step), ref):
            opt.arena.ensure_view(p).copy_(gr)
            r.grad = gr.float()
        opt.step()
        opt_r.step()
    torch.cuda.synchronize()
    for (n, p), r in zip(named, ref):
        seg = p.detach().float()
        if p.dtype == BF:   # the bf16 copy is the fp32 master rounded once: at most 1 ulp from the rounded reference
            assert torch.allclose(seg, r.detach().to(BF).float(), rtol=2 ** -7, atol=1e-6), n
        else:
            assert torch.allclose(seg, r.detach(), rtol=1e-5, atol=1e-6), n
    # and the fp32 masters themselves track torch.optim.SGD to fp32 accuracy
    for s in opt.arena.plans[BF].segments:
        m = opt.master[BF][s.offset:s.offset + s.numel]
        r = dict(zip([n for n, _ in named], ref))[s.name].detach().reshape(-1)
        assert torch.allclose(m, r, rtol=2e-5, atol=1e-6), s.name


def test_fused_lars_bf16_params_matches_plain_pytorch_lars_over_10_steps():
    """LARS on the GPU kernels (segmented norms + fused update, bf16 parameters with fp32 masters)
    against an independent fp32 implementation of the rule in the module docstring."""
    from distributeddeeplearning_amd.optim import FusedLARS

    named = _mixed_params(seed=9)
    lr, mu, wd, eta, eps = 0.5, 0.9, 1e-4, 0.001, 1e-8
    opt = FusedLARS(named, lr=lr, momentum=mu, weight_decay=wd, trust_coefficient=eta, eps=eps)
    W = [p.detach().float().clone() for _, p in named]
    V = [torch.zeros_like(w) for w in W]
    for step in range(10):
        grads = _grads(named, step)
        opt.zero_grad()
        for (_, p), gr in zip(named, grads):
            opt.arena.ensure_view(p).copy_(gr)
        opt.step()
        for i, gr in enumerate(grads):
            g32, w = gr.float(), W[i]
            if w.dim() > 1:
                wn, gn = w.norm(), g32.norm()
                lam = eta * wn / (gn + wd * wn + eps) if (wn > 0 and gn > 0) else 1.0
                V[i] = mu * V[i] + lr * lam * (g32 + wd * w)
                W[i] = w - V[i]
            else:
                V[i] = mu * V[i] + g32
                W[i] = w - lr * V[i]
    torch.cuda.synchronize()
    for (n, p), w in zip(named, W):
        if p.dtype == BF:
            assert torch.allclose(p.detach().float(), w.to(BF).float(), rtol=2 ** -7, atol=1e-6), n
        else:
            assert torch.allclose(p.detach(), w, rtol=1e-5, atol=1e-6), n
    for s in opt.arena.plans[BF].segments:
        m = opt.master[BF][s.offset:s.offset + s.numel]
        w = dict(zip([n for n, _ in named], W))[s.name].reshape(-1)
        assert torch.allclose(m, w, rtol=5e-5, atol=1e-6), s.name


# ---- one code path on the GPU -------------------------------------------------------------------------
def test_gpu_tensors_outside_the_kernel_envelope_raise():
    from distributeddeeplearning_amd.ops import functional as ops

    _hip()
    x32 = torch.randn(2, 8, 8, 8, device=DEV)
    w32 = torch.randn(8, 8, 3, 3, device=DEV)
    with pytest.raises(ops.NoHipKernel):
        ops.conv2d(x32, w32, 1, 1)                      # fp32 on the GPU: no silent MIOpen
    xb = x32.to(BF).contiguous(memory_format=CL)
    with pytest.raises(ops.NoHipKernel):
        ops.conv2d(xb, torch.randn(8, 8, 5, 5, device=DEV).to(BF), 1, 2)   # 5x5: outside the envelope
    with pytest.raises(ops.NoHipKernel):
        ops.batchnorm_act(x32, torch.ones(8, device=DEV), torch.zeros(8, device=DEV), torch.zeros(8, device=DEV),
                          torch.ones(8, device=DEV), True)
    prev = ops.backend()
    ops.set_backend("aten")                                # the oracle is an explicit request
    try:
        assert ops.conv2d(x32, w32, 1, 1).shape == (2, 8, 8, 8)
    finally:
        ops.set_backend(prev)


def test_topk_hits_runs_the_label_rank_kernel_and_matches_topk():
    from distributeddeeplearning_amd.ops import functional as ops

    _hip()
    g = torch.Generator(device=DEV).manual_seed(2)
    logits = torch.randn((300, 1000), generator=g, device=DEV).to(BF)
    target = torch.randint(0, 1000, (300,), generator=g, device=DEV)
    hits = ops.topk_hits(logits, target, (1, 5))
    assert hits.is_cuda and hits.dtype == torch.float64
    _, pred = logits.float().topk(5, dim=1)
    hit = pred.eq(target.view(-1, 1))
    # bf16 logits tie often: the kernel counts strictly-greater logits, topk breaks ties by index
    assert abs(float(hits[0]) - float(hit[:, :1].any(dim=1).sum())) <= 2
    assert abs(float(hits[1]) - float(hit.any(dim=1).sum())) <= 3
    assert ops.topk_correct(logits, target, (1, 5)) == [int(hits[0]), int(hits[1])]


def test_validate_loop_uses_device_side_topk():
    """training.loop.validate on the HIP path: top-1/top-5 through K21, rank-averaged dict."""
    from distributeddeeplearning_amd.models import resnet18
    from distributeddeeplearning_amd.ops import functional as ops
    from distributeddeeplearning_amd.training.loop import validate
    from distributeddeeplearning_amd.utils.logging import get_logger

    torch.manual_seed(1)
    model = resnet18(num_classes=16).to_compute(torch.device(DEV), BF)
    data = [(torch.rand(8, 3, 64, 64), torch.randint(0, 16, (8,))) for _ in range(3)]
    out = validate(data, model, ops.softmax_cross_entropy, get_logger("test"), torch.device(DEV), BF)
    assert out["samples"] == 24 and 0.0 <= out["top1"] <= out["top5"] <= 1.0 and math.isfinite(out["loss"])


# ---- tensors beyond 2 GiB -----------------------------------------------------
@pytest.mark.parametrize("case", [(1400, 256, 56, 64, 1), (5400, 64, 56, 64, 3)], ids=["1x1_2.25GB", "3x3_2.17GB"])
def test_conv_kernels_on_tensors_beyond_2gib(case):
    """Input (and input gradient) of 2.2 GB: the buffer descriptors are re-based per tile / per
    split, offsets inside stay 32-bit.  Sampled images against fp32, the weight gradient in full."""
    hip = _hip()
    B, Cin, H, Cout, k = case
    pad = k // 2
    g = torch.Generator(device=DEV).manual_seed(B)
    x = torch.empty((B, Cin, H, H), dtype=BF, device=DEV).contiguous(memory_format=CL)
    for i in range(0, B, 200):      # filled in slices: no 9 GB fp32 temporary
        x[i:i + 200] = torch.randn((min(200, B - i), Cin, H, H), generator=g, device=DEV).to(BF)
    assert x.numel() * 2 > (1 << 31)
    x.requires_grad_(True)
    w = (torch.randn((Cout, Cin, k, k), generator=g, device=DEV) / math.sqrt(Cin * k * k)).to(BF)
    w = w.contiguous(memory_format=CL).requires_grad_(True)
    y, stats = hip.conv2d_stats(x, w, 1, pad, True)
    gy = torch.empty_like(y)
    for i in range(0, B, 200):
        gy[i:i + 200] = torch.randn((min(200, B - i), Cout, H, H), generator=g, device=DEV).to(BF)
    y.backward(gy)
    torch.cuda.synchronize()
    wr = w.detach().float().contiguous().requires_grad_(True)
    sample = [0, 1, B // 2, B - 2, B - 1]          # the last images live beyond the 2 GiB mark
    xs = x.detach()[sample].float().contiguous().requires_grad_(True)
    ys = F.conv2d(xs, wr, None, 1, pad)
    assert relerr(y.detach()[sample], ys) < 1.5e-2 and rms_relerr(y.detach()[sample], ys) < 4e-3
    (dxs,) = torch.autograd.grad(ys, xs, gy[sample].float())
    assert relerr(x.grad[sample], dxs) < 1.5e-2 and rms_relerr(x.grad[sample], dxs) < 4e-3
    dw = torch.zeros_like(wr)
    for i in range(0, B, 100):
        xc = x.detach()[i:i + 100].float().contiguous()
        dw += torch.autograd.grad(F.conv2d(xc, wr, None, 1, pad), wr, gy[i:i + 100].float())[0]
    assert relerr(w.grad, dw) < 1.5e-2 and rms_relerr(w.grad, dw) < 4e-3, (relerr(w.grad, dw), rms_relerr(w.grad, dw))


def test_resnet50_batch_1408_of_replicated_images_equals_batch_128():
    """Every kernel of the training step on > 2 GiB tensors (the stem output of 1408 images is
    2.26 GB, every layer-1 tensor 2.26 GB): a batch made of 176 copies of the same 8 images has the
    batch statistics, the mean loss and the mean gradients of a batch of 16 copies, so any 32-bit
    offset that wraps shows up as a difference between the two runs.  (The reference run is itself
    replicated because a freshly initialised 50-layer net amplifies rounding differences: 8 images
    against 2 copies of them already differ by 2 % in fp32 ATen gradients, 16 against 176 copies do
    not - scripts/debug_big_batch.py.)"""
    from distributeddeeplearning_amd.models import resnet50
    from distributeddeeplearning_amd.ops import functional as ops
    from distributeddeeplearning_amd.optim import FusedSGD

    g = torch.Generator(device="cpu").manual_seed(21)
    x8 = torch.rand((8, 3, 224, 224), generator=g).to(DEV).to(BF).contiguous(memory_format=CL)
    y8 = torch.randint(0, 1000, (8,), generator=g).to(DEV)

    def run(rep):
        _hip().reset_caches()
        torch.manual_seed(0)
        model = resnet50(num_classes=1000, zero_init_residual=True).to_compute(torch.device(DEV), BF).train()
        opt = FusedSGD(model.named_parameters(), lr=0.0, momentum=0.0)
        x = x8.repeat(rep, 1, 1, 1).contiguous(memory_format=CL)
        y = y8.repeat(rep)
        opt.zero_grad()
        loss = ops.softmax_cross_entropy(model(x), y)
        loss.backward()
        torch.cuda.synchronize()
        grads = {n: p.grad.float().clone() for n, p in model.named_parameters()}
        rm = model.bn1.running_mean.detach().clone()
        return float(loss.detach()), grads, rm

    lref, gref, rmref = run(16)
    torch.cuda.empty_cache()
    lbig, gbig, rmbig = run(176)
    assert abs(lbig - lref) < 1e-4 * abs(lref), (lbig, lref)
    assert torch.allclose(rmbig, rmref, rtol=1e-3, atol=1e-5)
    # a wrapped offset drops or duplicates whole images (errors of order 1); rounding differences of
    # the two tilings stay at the per-cent level even in the deepest (stem) gradients
    worst = max((rms_relerr(gbig[n], gref[n]), n) for n in gref if float(gref[n].abs().max()) > 0)
    assert worst[0] < 0.15, worst
    cos = min((float((gbig[n] * gref[n]).sum() / (gbig[n].norm() * gref[n].norm()).clamp_min(1e-30)), n)
              for n in gref if float(gref[n].abs().max()) > 0)
    assert cos[0] > 0.99, cos


def test_training_with_fp32_gradient_arenas_matches_bf16_arenas(monkeypatch):
    """HVD_COMM_DTYPE=fp32 on the HIP path: the weight-gradient reduction writes fp32 into the arena
    (the all-reduce buffer), the fused SGD reads fp32 gradients for bf16 parameters."""
    from distributeddeeplearning_amd.models import resnet18

    ref = _trajectory(resnet18, 8, 64, 64, 16, hip_path=True)
    monkeypatch.setenv("HVD_COMM_DTYPE", "fp32")
    got = _trajectory(resnet18, 8, 64, 64, 16, hip_path=True)
    assert all(math.isfinite(v) for v in got), got
    assert max(abs(a - b) / max(abs(b), 0.5) for a, b in zip(got, ref)) < 0.05, (got, ref)
    assert got[-1] < 0.8 * got[0]


def test_two_layer1_blocks_beyond_2gib_per_layer_against_fp32():
    """The chained, linearity-taking path at the size that crosses 2 GiB: the
    first two bottlenecks of layer 1 (projection + identity, 56x56) at batch 1400 - every wide
    tensor is 2.25 GB - against the same blocks in fp32 ATen, parameter by parameter and for the
    input gradient; error budget = twice what the stock bf16 ATen path loses against that
    reference (floor 2 %).  The whole-net 1408-image test above then is not the only guard of the
    > 2 GiB offsets in the fused epilogues (mask bits, BatchNorm partials, two-segment GEMM).

    The big batch is 175 copies of 8 images (input and upstream gradient alike) and the fp32 / stock
    bf16 references run on 2 copies: replicated batches have the same batch statistics and the same
    per-image input gradients, and parameter gradients that scale with the number of copies - so the
    references stay small (MIOpen on 4.5 GB fp32 tensors takes minutes) while a wrapped 32-bit offset
    anywhere in the 2.25 GB tensors still shows as an error of order one."""
    from distributeddeeplearning_amd.models.resnet import Bottleneck, _block_spec
    from distributeddeeplearning_amd.ops import functional as ops

    hip = _hip()
    hip.reset_caches()
    REP_BIG, REP_REF = 175, 2

    def make(dtype):
        torch.manual_seed(5)
        ms = [Bottleneck(64, 64, 1, True, False), Bottleneck(256, 64, 1, False, False)]
        for m in ms:
            m.to(DEV).train()
            for mod in m.modules():
                if hasattr(mod, "kernel"):
                    wq = mod.weight.data.to(BF)
                    mod.weight.data = wq.to(dtype).contiguous(memory_format=CL)
        return ms

    g = torch.Generator(device=DEV).manual_seed(77)
    x8 = torch.relu(torch.randn((8, 64, 56, 56), generator=g, device=DEV)).to(BF).contiguous(memory_format=CL)
    g8 = torch.randn((8, 256, 56, 56), generator=g, device=DEV).to(BF).contiguous(memory_format=CL)

    def run(backend, dtype, chained, rep):
        ms = make(dtype)
        x = x8.repeat(rep, 1, 1, 1).contiguous(memory_format=CL).to(dtype).requires_grad_(True)
        gout = g8.repeat(rep, 1, 1, 1).contiguous(memory_format=CL).to(dtype)
        ops.set_backend(backend)
        try:
            if chained:
                out = ops.residual_chain(x, [_block_spec(m.pairs(), m.downsample) for m in ms], True, 0.1, 1e-5)
            else:
                out = x
                for m in ms:
                    out = m(out)
            assert rep < 100 or out.numel() * 2 > (1 << 31)
            out.backward(gout)
            torch.cuda.synchronize()
        finally:
            ops.set_backend("auto")
        # per-image input gradients do not depend on the number of copies; parameter gradients are sums
        # over the batch: normalise to one copy
        res = (x.grad[-8:].float().clone(), x.grad[:8].float().clone(),
               [p.grad.float() / rep for m in ms for p in m.parameters()],
               [n for m in ms for n, _ in m.named_parameters()])
        del out, x, gout, ms
        torch.cuda.empty_cache()
        return res

    dxl_r, dxf_r, g_r, names = run("aten", torch.float32, False, REP_REF)
    dxl_a, dxf_a, g_a, _ = run("aten", BF, False, REP_REF)
    dxl_h, dxf_h, g_h, _ = run("auto", BF, True, REP_BIG)

    def ok(h, a, r, what):
        eh, ea = rms_relerr(h, r), rms_relerr(a, r)
        assert eh < max(2.0 * ea, 2e-2), "%s: hip %.4f vs stock bf16 %.4f" % (what, eh, ea)

    ok(dxl_h, dxl_a, dxl_r, "dx (last 8 images: beyond the 2 GiB mark)")
    ok(dxf_h, dxf_a, dxf_r, "dx (first 8 images)")
    for n, h, a, r in zip(names, g_h, g_a, g_r):
        ok(h, a, r, n)

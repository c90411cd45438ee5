"""GPU numerics tests: every HIP kernel against a plain PyTorch fp32 reference of the same op.

Run on an MI355X with ``python -m pytest tests -m gpu``.  Strategy follows SURVEY §4: all
ResNet-50 conv shape classes x forward / backward-data / backward-weights, BatchNorm
forward/backward, pooling, dense head, loss, the fused optimizers and the synthetic generator.
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


def rand_nhwc(B, C, H, W, scale=1.0, seed=0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    x = (torch.randn((B, C, H, W), generator=g, device=DEV) * scale).to(BF)
    return x.contiguous(memory_format=CL)


CONV_CASES = [
    # (B, Cin, H, W, Cout, k, stride, pad)   -- shape classes of SURVEY §2.7
    (4, 64, 56, 56, 64, 1, 1, 0),     # a1
    (4, 256, 56, 56, 64, 1, 1, 0),    # a2
    (4, 64, 56, 56, 64, 3, 1, 1),     # a3
    (4, 64, 56, 56, 256, 1, 1, 0),    # a4
    (3, 128, 56, 56, 128, 3, 2, 1),   # b2  (odd batch -> ragged M tile)
    (4, 256, 56, 56, 512, 1, 2, 0),   # b6  1x1 stride 2
    (5, 256, 14, 14, 256, 3, 1, 1),   # c4
    (4, 1024, 14, 14, 256, 1, 1, 0),  # c3
    (6, 512, 7, 7, 512, 3, 1, 1),     # d4
    (6, 512, 7, 7, 2048, 1, 1, 0),    # d5
    (2, 512, 14, 14, 512, 3, 2, 1),   # d2
    (2, 32, 9, 11, 96, 3, 1, 1),      # BK=32 path, odd sizes, Cout not a tile multiple
    (2, 64, 15, 15, 64, 3, 2, 1),     # stride 2 with odd extent: dilated (non-parity) dgrad path
    (3, 64, 10, 12, 128, 1, 2, 0),    # 1x1 stride 2, parity dgrad with empty tap classes
    (65, 128, 28, 27, 256, 3, 1, 1),  # 256x128 block tile (N >= 256, R*S*C >= 1024, >= 384 tiles), ragged last tile
    (96, 1024, 23, 23, 256, 1, 1, 0), # 256x128 tile, 1x1 (fwd); its dgrad (N = 1024, C = 256) stays on 128x128
    (33, 64, 56, 56, 256, 1, 1, 0),   # weight gradient over > 100 000 rows: interleaved reduction ranges, ragged last group
    (33, 256, 56, 56, 64, 1, 1, 0),   # the same with the wide side as the input
    (64, 256, 14, 14, 1024, 1, 1, 0), # 256x256 / 8-wave plain-GEMM kernel (gemm256.hip): 196 whole tiles, short reduction
    (65, 1024, 14, 14, 1024, 1, 1, 0),  # ... long reduction, ragged last tile (68 rows in its last 128-row statistics row)
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_bwd(case):
    hip = _hip()
    B, Cin, H, W, Cout, k, stride, pad = case
    x = rand_nhwc(B, Cin, H, W, seed=1).requires_grad_(True)
    w = (torch.randn((Cout, Cin, k, k), device=DEV) / math.sqrt(Cin * k * k)).to(BF)
    w = w.contiguous(memory_format=CL).requires_grad_(True)
    y, stats = hip.conv2d_stats(x, w, stride, pad, True)
    xr = x.detach().float().requires_grad_(True)
    wr = w.detach().float().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, stride, pad)
    assert y.shape == yr.shape
    assert relerr(y, yr) < 1.5e-2 and rms_relerr(y, yr) < 4e-3
    # fused statistics = column sums of the bf16 output
    y2 = y.detach().float().permute(0, 2, 3, 1).reshape(-1, Cout)
    s1 = stats[:, 0, :].sum(0)
    s2 = stats[:, 1, :].sum(0)
    assert torch.allclose(s1, y2.sum(0), rtol=1e-3, atol=1e-2 * float(y2.abs().max()))
    assert torch.allclose(s2, (y2 * y2).sum(0), rtol=1e-3, atol=1e-3)
    g = rand_nhwc(*y.shape, seed=2)
    y.backward(g)
    yr.backward(g.float())
    assert relerr(x.grad, xr.grad) < 1.5e-2 and rms_relerr(x.grad, xr.grad) < 4e-3
    assert relerr(w.grad, wr.grad) < 1.5e-2 and rms_relerr(w.grad, wr.grad) < 4e-3


def test_stem_conv():
    hip = _hip()
    B, H, W = 3, 64, 96
    x = torch.rand((B, 3, H, W), device=DEV).to(BF).contiguous(memory_format=CL)
    w = (torch.randn((64, 3, 7, 7), device=DEV) / math.sqrt(147)).to(BF)
    w = w.contiguous(memory_format=CL).requires_grad_(True)
    y, stats = hip.conv2d_stats(x, w, 2, 3, True)
    wr = w.detach().float().requires_grad_(True)
    yr = F.conv2d(x.float(), wr, None, 2, 3)
    assert y.shape == yr.shape
    assert relerr(y, yr) < 1.5e-2
    g = rand_nhwc(*y.shape, seed=3)
    y.backward(g)
    yr.backward(g.float())
    assert relerr(w.grad, wr.grad) < 1.5e-2


@pytest.mark.parametrize("relu,res", [(True, False), (True, True), (False, False), (False, True)])
@pytest.mark.parametrize("shape", [(4, 64, 28, 28), (3, 256, 7, 9), (2, 2048, 7, 7)])
def test_batchnorm_act(shape, relu, res):
    hip = _hip()
    B, C, H, W = shape
    y = (rand_nhwc(B, C, H, W, scale=2.0, seed=4).float() + 0.5).to(BF).contiguous(memory_format=CL)
    y.requires_grad_(True)
    gamma = (torch.rand(C, device=DEV) + 0.5).requires_grad_(True)
    beta = (torch.randn(C, device=DEV) * 0.1).requires_grad_(True)
    r = rand_nhwc(B, C, H, W, seed=5).requires_grad_(True) if res else None
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    z = hip.batchnorm_act(y, gamma, beta, rm, rv, True, 0.1, 1e-5, relu, r)
    yr = y.detach().float().requires_grad_(True)
    gr, br = gamma.detach().clone().requires_grad_(True), beta.detach().clone().requires_grad_(True)
    rr = r.detach().float().requires_grad_(True) if res else None
    rm2, rv2 = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    zr = F.batch_norm(yr, rm2, rv2, gr, br, True, 0.1, 1e-5)
    if res:
        zr = zr + rr
    if relu:
        zr = F.relu(zr)
    assert relerr(z, zr) < 1.5e-2
    assert torch.allclose(rm, rm2, rtol=1e-3, atol=1e-4)
    assert torch.allclose(rv, rv2, rtol=1e-3, atol=1e-4)
    g = rand_nhwc(B, C, H, W, seed=6)
    # use the kernel's own activation pattern for the reference too (bf16 rounding can flip
    # elements that sit exactly at the ReLU threshold)
    z.backward(g)
    zr.backward(g.float())
    assert rms_relerr(y.grad, yr.grad) < 1e-2
    assert relerr(gamma.grad, gr.grad) < 1e-2
    assert relerr(beta.grad, br.grad) < 1e-2
    if res:
        assert rms_relerr(r.grad, rr.grad) < 1e-2


@pytest.mark.parametrize("T,C", [(37, 64), (512, 256), (513, 128), (1568, 128), (1568, 520), (2048, 64), (2049, 64),
                                 (6272, 64)])
def test_bn_fwd_finalize_table_lengths(T, C):
    """Both routes of the statistics finalize (one launch up to 512 tiles, fold + finalize above)
    against an fp64 reduction of the same partial table."""
    from distributeddeeplearning_amd.ops import kernels as K

    g = torch.Generator(device="cpu").manual_seed(T * 131 + C)
    rows = 128
    M = (T - 1) * rows + 77
    s1 = torch.randn((T, C), generator=g) * 4.0 + 64.0            # per-tile sums with a common offset
    s2 = s1 * s1 / rows + torch.rand((T, C), generator=g) * 50.0   # per-tile sums of squares (>= s1^2 / rows)
    partial = torch.stack([s1, s2], dim=1).contiguous().to(DEV)
    gamma = (torch.rand(C, generator=g) + 0.5).to(DEV)
    beta = torch.randn(C, generator=g).to(DEV)
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    mean, inv, sc, sh = (torch.empty(C, device=DEV) for _ in range(4))
    K.bn_fwd_finalize(partial, T, C, M, rows, gamma, beta, rm, rv, 0.1, 1e-5, mean, inv, sc, sh)
    torch.cuda.synchronize()
    p64 = partial.double().cpu()
    S1, S2 = p64[:, 0].sum(0), p64[:, 1].sum(0)
    mu = S1 / M
    var = (S2 / M - mu * mu).clamp_min(0)
    inv_ref = 1.0 / torch.sqrt(var + 1e-5)
    assert torch.allclose(mean.cpu().double(), mu, rtol=1e-6, atol=1e-6)
    assert torch.allclose(inv.cpu().double(), inv_ref, rtol=2e-5, atol=0)
    assert torch.allclose(sc.cpu().double(), gamma.cpu().double() * inv_ref, rtol=2e-5, atol=0)
    assert torch.allclose(sh.cpu().double(), beta.cpu().double() - mu * gamma.cpu().double() * inv_ref, rtol=2e-5, atol=1e-5)
    assert torch.allclose(rm.cpu().double(), 0.1 * mu, rtol=1e-5, atol=1e-6)
    assert torch.allclose(rv.cpu().double(), 0.9 + 0.1 * var * M / (M - 1), rtol=2e-5, atol=1e-6)


def test_bn_apply_writes_relu_bitmask():
    """``bits`` of the forward BatchNorm apply: bit e of byte i = (z[8i + e] > 0)."""
    from distributeddeeplearning_amd.ops import kernels as K
    B, C, H, W = 5, 72, 9, 7
    y = rand_nhwc(B, C, H, W, seed=51)
    res = rand_nhwc(B, C, H, W, seed=52)
    scale = torch.rand(C, device=DEV) + 0.5
    shift = torch.randn(C, device=DEV) * 0.2
    z = torch.empty_like(y)
    bits = torch.zeros(B * H * W * C // 8, dtype=torch.uint8, device=DEV)
    K.bn_apply_fwd(y, scale, shift, res, z, B * H * W, C, True, bits=bits)
    z_ref = torch.empty_like(y)
    K.bn_apply_fwd(y, scale, shift, res, z_ref, B * H * W, C, True)
    assert torch.equal(z, z_ref)
    zb = (z.permute(0, 2, 3, 1).reshape(-1, 8) > 0).to(torch.int32)
    want = (zb << torch.arange(8, device=DEV, dtype=torch.int32)).sum(1).to(torch.uint8)
    assert torch.equal(bits, want)
    assert 0.2 < float((z > 0).float().mean()) < 0.8


def test_batchnorm_eval():
    hip = _hip()
    B, C, H, W = 2, 128, 5, 5
    y = rand_nhwc(B, C, H, W, seed=7)
    gamma, beta = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV)
    rm, rv = torch.randn(C, device=DEV), torch.rand(C, device=DEV) + 0.5
    z = hip.batchnorm_act(y, gamma, beta, rm, rv, False, 0.1, 1e-5, True, None)
    zr = F.relu(F.batch_norm(y.float(), rm, rv, gamma, beta, False, 0.1, 1e-5))
    assert relerr(z, zr) < 1.5e-2


def test_maxpool():
    hip = _hip()
    for (B, C, H, W) in [(2, 64, 112, 112), (3, 64, 17, 23)]:
        x = F.relu(rand_nhwc(B, C, H, W, seed=8)).contiguous(memory_format=CL).requires_grad_(True)
        out = hip.maxpool3x3s2(x)
        xr = x.detach().float().requires_grad_(True)
        outr = F.max_pool2d(xr, 3, 2, 1)
        assert torch.equal(out.float(), outr)
        g = rand_nhwc(*out.shape, seed=9)
        out.backward(g)
        outr.backward(g.float())
        assert relerr(x.grad, xr.grad) < 1e-2


def test_global_avgpool():
    hip = _hip()
    x = rand_nhwc(5, 2048, 7, 7, seed=10).requires_grad_(True)
    out = hip.global_avgpool(x)
    xr = x.detach().float().requires_grad_(True)
    outr = xr.mean(dim=(2, 3))
    assert relerr(out, outr) < 1e-2
    g = torch.randn_like(outr).to(BF)
    out.backward(g)
    outr.backward(g.float())
    assert relerr(x.grad, xr.grad) < 1e-2


@pytest.mark.parametrize("B,Cin,N", [(32, 2048, 1000), (7, 512, 1000), (256, 2048, 1024)])
def test_linear(B, Cin, N):
    hip = _hip()
    x = (torch.randn(B, Cin, device=DEV)).to(BF).requires_grad_(True)
    w = (torch.randn(N, Cin, device=DEV) * 0.02).to(BF).requires_grad_(True)
    b = torch.randn(N, device=DEV).requires_grad_(True)
    y = hip.linear(x, w, b)
    xr, wr, br = (t.detach().float().requires_grad_(True) for t in (x, w, b))
    yr = F.linear(xr, wr, br)
    assert relerr(y, yr) < 1.5e-2
    g = torch.randn_like(yr).to(BF)
    y.backward(g)
    yr.backward(g.float())
    assert relerr(x.grad, xr.grad) < 1.5e-2
    assert relerr(w.grad, wr.grad) < 1.5e-2
    assert relerr(b.grad, br.grad) < 1e-2


def test_softmax_cross_entropy():
    hip = _hip()
    B, N = 37, 1000
    logits = (torch.randn(B, N, device=DEV) * 3).to(BF).requires_grad_(True)
    target = torch.randint(0, N, (B,), device=DEV)
    loss = hip.softmax_cross_entropy(logits, target)
    lr = logits.detach().float().requires_grad_(True)
    lossr = F.cross_entropy(lr, target)
    assert abs(float(loss) - float(lossr)) < 2e-3 * max(1.0, abs(float(lossr)))
    loss.backward()
    lossr.backward()
    assert relerr(logits.grad, lr.grad) < 1e-2


def test_weight_cache_tracks_updates():
    hip = _hip()
    hip.reset_caches()
    x = rand_nhwc(2, 64, 8, 8, seed=11).requires_grad_(True)
    w = (torch.randn((64, 64, 3, 3), device=DEV) * 0.05).to(BF).contiguous(memory_format=CL)
    w.requires_grad_(True)
    for it in range(2):
        x.grad = None
        y = hip.conv2d(x, w, 1, 1)
        y.backward(torch.ones_like(y))
        xr = x.detach().float().requires_grad_(True)
        F.conv2d(xr, w.detach().float(), None, 1, 1).backward(torch.ones_like(y).float())
        assert relerr(x.grad, xr.grad) < 1.5e-2
        with torch.no_grad():
            w.mul_(-0.5)  # in-place update must invalidate the transposed copy


def test_bottleneck_block_matches_fp32_reference():
    """One projection bottleneck (4 convs, 4 BNs, residual) forward + backward on the HIP kernels
    against the same block evaluated by ATen in fp32; the error budget is the error of the
    stock bf16 ATen/MIOpen path against that same fp32 reference (x2).  (Whole-network
    comparisons are not meaningful: a random-init ResNet-50 in training mode amplifies bf16
    rounding noise by ~1.3x per block and even two ATen runs of it disagree by >50 % in the
    early-layer gradients — measured with scripts/debug_e2e.py.)"""
    from distributeddeeplearning_amd.models.resnet import Bottleneck
    from distributeddeeplearning_amd.ops import functional as ops

    _hip().reset_caches()
    torch.manual_seed(0)

    def make(dtype):
        torch.manual_seed(0)
        m = Bottleneck(256, 128, 2, True, False).to(DEV).train()
        for mod in m.modules():
            if hasattr(mod, "kernel"):
                wq = mod.weight.data.to(BF)  # every variant sees the same bf16-rounded weights
                mod.weight.data = wq.to(dtype).contiguous(memory_format=CL)
        return m

    x = rand_nhwc(8, 256, 28, 28, seed=12)
    g = rand_nhwc(8, 512, 14, 14, seed=13)

    def run(m, backend, dtype):
        ops.set_backend(backend)
        try:
            xi = x.detach().to(dtype).requires_grad_(True)
            out = m(xi)
            out.backward(g.to(dtype))
        finally:
            ops.set_backend("auto")
        grads = {n: p.grad.detach().float() for n, p in m.named_parameters()}
        return out.detach().float(), xi.grad.detach().float(), grads

    o_ref, dx_ref, g_ref = run(make(torch.float32), "aten", torch.float32)
    o_at, dx_at, g_at = run(make(BF), "aten", BF)
    o_hip, dx_hip, g_hip = run(make(BF), "auto", BF)

    def ok(h, a, r, what):
        eh, ea = rms_relerr(h, r), rms_relerr(a, r)
        assert eh < max(2.0 * ea, 2e-2), "%s: hip %.4f vs stock bf16 %.4f" % (what, eh, ea)

    ok(o_hip, o_at, o_ref, "output")
    ok(dx_hip, dx_at, dx_ref, "dx")
    for n in g_ref:
        ok(g_hip[n], g_at[n], g_ref[n], n)


@pytest.mark.parametrize("case", [(4, 128, 28, 28, 128, 3, 1, 1), (3, 128, 28, 28, 128, 3, 2, 1),
                                  (5, 64, 14, 14, 256, 1, 1, 0), (2, 64, 15, 15, 64, 3, 2, 1),
                                  (65, 256, 28, 27, 128, 3, 1, 1)])  # last: 256x128 tile dgrad
def test_dgrad_fused_bn_reduce_and_relu_mask_epilogues(case):
    """EPI 2 (masked gradient + BatchNorm-backward partial sums) and EPI 3 (ReLU pre-mask) of the
    backward-data kernel against the unfused kernel followed by plain tensor math."""
    hip = _hip()
    hip.reset_caches()
    B, Cin, H, W, Cout, k, stride, pad = case
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    w = (torch.randn((Cout, Cin, k, k), device=DEV) / math.sqrt(Cin * k * k)).to(BF)
    w = w.contiguous(memory_format=CL)
    dy = rand_nhwc(B, Cout, Ho, Wo, seed=31)
    yprev = rand_nhwc(B, Cin, H, W, seed=32)
    fco = torch.stack([torch.rand(Cin, device=DEV) + 0.5, torch.randn(Cin, device=DEV) * 0.3])
    dx = hip._conv_dgrad(dy, w, (B, Cin, H, W), stride, pad)
    g, partial = hip._conv_dgrad_bn(dy, w, (B, Cin, H, W), stride, pad, yprev, fco)
    on = torch.addcmul(fco[1].view(1, -1, 1, 1), yprev.float(), fco[0].view(1, -1, 1, 1)) > 0
    g_ref = torch.where(on, dx.float(), torch.zeros((), device=DEV))
    # the mask is recomputed with one fused multiply-add in both places: only elements whose
    # pre-activation rounds to exactly zero may differ
    mism = (g.float() != g_ref).float().mean()
    assert float(mism) < 1e-4
    s = partial.sum(0)
    gf = g.float()
    s1 = gf.sum(dim=(0, 2, 3))
    s2 = (gf * yprev.float()).sum(dim=(0, 2, 3))
    assert torch.allclose(s[0], s1, rtol=1e-3, atol=1e-3 * float(s1.abs().max()))
    assert torch.allclose(s[1], s2, rtol=1e-3, atol=1e-3 * float(s2.abs().max()))
    z = torch.relu(rand_nhwc(B, Cin, H, W, seed=33))
    add = rand_nhwc(B, Cin, H, W, seed=34)
    for addend in (None, add):
        if k == 1 and stride == 2 and addend is None:
            continue
        base = hip._conv_dgrad(dy, w, (B, Cin, H, W), stride, pad, addend=addend)
        masked = hip._conv_dgrad(dy, w, (B, Cin, H, W), stride, pad, addend=addend, mask_src=z)
        assert torch.equal(masked, torch.where(z > 0, base, torch.zeros_like(base)))
        both, part = hip._conv_dgrad(dy, w, (B, Cin, H, W), stride, pad, addend=addend, mask_src=z,
                                     bn_y=yprev)
        assert torch.equal(both, masked)
        t1 = masked.float().sum(dim=(0, 2, 3))
        t2 = (masked.float() * yprev.float()).sum(dim=(0, 2, 3))
        ps = part.sum(0)
        assert torch.allclose(ps[0], t1, rtol=1e-3, atol=1e-3 * float(t1.abs().max()))
        assert torch.allclose(ps[1], t2, rtol=1e-3, atol=1e-3 * float(t2.abs().max()))
        # the same mask as one bit per element (as the forward BatchNorm apply writes it)
        zb = (z.permute(0, 2, 3, 1).reshape(-1, 8) > 0).to(torch.int32)
        bits = (zb << torch.arange(8, device=DEV, dtype=torch.int32)).sum(1).to(torch.uint8)
        viabits = hip._conv_dgrad(dy, w, (B, Cin, H, W), stride, pad, addend=addend, mask_src=z,
                                  mask_bits=bits)
        assert torch.equal(viabits, masked)
        # a second BatchNorm fed by the same gradient (projection shortcut): its own table
        y2 = rand_nhwc(B, Cin, H, W, seed=35)
        both2, part_a, part_b = hip._conv_dgrad(dy, w, (B, Cin, H, W), stride, pad, addend=addend,
                                                mask_src=z, bn_y=yprev, bn_y2=y2)
        assert torch.equal(both2, masked) and torch.equal(part_a, part)
        t3 = (masked.float() * y2.float()).sum(dim=(0, 2, 3))
        pb = part_b.sum(0)
        assert torch.equal(part_b[:, 0], part[:, 0])
        assert torch.allclose(pb[1], t3, rtol=1e-3, atol=1e-3 * float(t3.abs().max()))


def test_residual_chain_matches_unchained_blocks():
    """``ops.residual_chain`` (ReLU-masked gradients handed between fused block nodes, BatchNorm
    reduce in the dgrad epilogue) against the same blocks run one by one without either fusion."""
    from distributeddeeplearning_amd.models.resnet import Bottleneck, _block_spec
    from distributeddeeplearning_amd.ops import functional as ops

    hip = _hip()
    hip.reset_caches()

    def make():
        torch.manual_seed(7)
        ms = [Bottleneck(128, 64, 2, True, False), Bottleneck(256, 64, 1, False, False),
              Bottleneck(256, 64, 1, False, False)]
        for m in ms:
            m.to(DEV).train()
            for mod in m.modules():
                if hasattr(mod, "kernel"):
                    mod.weight.data = mod.weight.data.to(BF).contiguous(memory_format=CL)
        return ms

    x0 = torch.relu(rand_nhwc(6, 128, 28, 28, seed=41))
    gout = rand_nhwc(6, 256, 14, 14, seed=42)

    def run(chained):
        ms = make()
        x = x0.detach().clone().requires_grad_(True)
        hip._FUSE_BN_REDUCE["on"] = chained
        try:
            if chained:
                specs = [_block_spec(m.pairs(), m.downsample) for m in ms]
                out = ops.residual_chain(x, specs, True, 0.1, 1e-5)
            else:
                out = x
                for m in ms:
                    out = m(out)
            out.backward(gout)
        finally:
            hip._FUSE_BN_REDUCE["on"] = True
        return out.detach(), x.grad, [p.grad for m in ms for p in m.parameters()]

    o_a, dx_a, g_a = run(False)
    o_b, dx_b, g_b = run(True)
    assert torch.equal(o_a, o_b)
    # the bit-packed ReLU mask is the same mask: switching it off changes nothing at all
    hip._RELU_BITS["on"] = False
    try:
        o_c, dx_c, g_c = run(True)
    finally:
        hip._RELU_BITS["on"] = True
    assert torch.equal(o_c, o_b) and torch.equal(dx_c, dx_b)
    assert all(torch.equal(a, b) for a, b in zip(g_c, g_b))
    # the projection BatchNorm's partials from the epilogue against the standalone reduce pass:
    # same sums in another order
    hip._FUSE_DS_REDUCE["on"] = False
    try:
        o_d, dx_d, g_d = run(True)
    finally:
        hip._FUSE_DS_REDUCE["on"] = True
    # (with the fused partials the projection's BatchNorm also goes by linearity, without them its dy is
    # materialised: same math, bf16 roundings in other places)
    assert rms_relerr(dx_d, dx_b) < 1e-2
    for a, b in zip(g_d, g_b):
        assert rms_relerr(a, b) < 1e-2
    # identical math up to the summation order of the BatchNorm partials (bf16 re-rounding of
    # dy then propagates): compare in rms
    assert rms_relerr(dx_b, dx_a) < 1e-2
    for a, b in zip(g_a, g_b):
        assert rms_relerr(b, a) < 1e-2


def test_residual_block_direct_grads_match_autograd_accumulation():
    """Fused block writing into pre-existing ``param.grad`` arena views (direct mode) must give
    the same gradients as returning them to autograd, and must notify the listeners once per
    parameter."""
    from distributeddeeplearning_amd.models.resnet import Bottleneck
    from distributeddeeplearning_amd.parallel.bucket import GradArena

    hip = _hip()
    hip.reset_caches()

    def make():
        torch.manual_seed(3)
        m = Bottleneck(128, 64, 1, True, False).to(DEV).train()
        for mod in m.modules():
            if hasattr(mod, "kernel"):
                mod.weight.data = mod.weight.data.to(BF).contiguous(memory_format=CL)
        return m

    x = rand_nhwc(4, 128, 14, 14, seed=21)
    g = rand_nhwc(4, 256, 14, 14, seed=22)
    ref = make()
    hip.set_direct_grads(False)
    ref(x).backward(g)
    m = make()
    arena = GradArena(list(m.named_parameters()), 1 << 20, 1 << 20)
    seen = []
    listener = lambda p: seen.append(id(p))  # noqa: E731
    hip.add_grad_listener(listener)
    hip.set_direct_grads(True)
    try:
        m(x).backward(g)
    finally:
        hip.set_direct_grads(False)
        hip.remove_grad_listener(listener)
    assert sorted(seen) == sorted(id(p) for p in m.parameters())
    for (n, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
        assert p.grad.data_ptr() == arena.views[id(p)].data_ptr(), n
        assert torch.allclose(p.grad.float(), q.grad.float(), rtol=2e-2, atol=1e-3), n


def test_resnet50_trains_on_hip_kernels():
    """End-to-end: ResNet-50 on the HIP path memorises one small batch (loss must fall)."""
    from distributeddeeplearning_amd.models import resnet50
    from distributeddeeplearning_amd.ops import functional as ops
    from distributeddeeplearning_amd.optim import FusedSGD

    _hip().reset_caches()
    torch.manual_seed(0)
    model = resnet50(num_classes=1000, zero_init_residual=True).to_compute(torch.device(DEV), BF).train()
    opt = FusedSGD(model.named_parameters(), lr=0.05, momentum=0.9)
    x = torch.rand((16, 3, 64, 64), device=DEV).to(BF).contiguous(memory_format=CL)
    t = torch.randint(0, 1000, (16,), device=DEV)
    losses = []
    for _ in range(12):
        opt.zero_grad()
        loss = ops.softmax_cross_entropy(model(x), t)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(math.isfinite(l) for l in losses), losses
    assert losses[-1] < 0.5 * losses[0], losses


def test_graphed_train_step_matches_eager():
    """training/graph.py: the whole step (zero_grad, forward, loss, backward with the side-stream
    weight gradients, fused SGD) captured into one HIP graph and replayed gives the same losses
    and weights as eager execution, including a learning-rate change after capture (device-side
    lr scale) and fresh data every step."""
    from distributeddeeplearning_amd.models import resnet50
    from distributeddeeplearning_amd.ops import functional as ops
    from distributeddeeplearning_amd.optim import FusedSGD
    from distributeddeeplearning_amd.training.graph import GraphedTrainStep

    g = torch.Generator(device="cpu").manual_seed(5)
    xs = [torch.rand((16, 3, 64, 64), generator=g).to(DEV).to(BF).contiguous(memory_format=CL) for _ in range(8)]
    ts = [torch.randint(0, 1000, (16,), generator=g).to(DEV) for _ in range(8)]

    def run(graph):
        _hip().reset_caches()
        torch.manual_seed(0)
        model = resnet50(num_classes=1000).to_compute(torch.device(DEV), BF).train()
        # powers of two: lr_by_value * device_scale is then exactly the eager path's lr
        opt = FusedSGD(model.named_parameters(), lr=2.0 ** -6, momentum=0.9, weight_decay=1e-4)
        step = GraphedTrainStep(model, opt, ops.softmax_cross_entropy, warmup=2, enabled=graph)
        losses = []
        for i, (x, t) in enumerate(zip(xs, ts)):
            if i == 5:
                opt.param_groups[0]["lr"] = 2.0 ** -8
            losses.append(float(step(x, t).detach().float()))
        # a short last batch after capture runs eagerly (the graph is for one shape), then replay again
        losses.append(float(step(xs[0][:8], ts[0][:8]).detach().float()))
        losses.append(float(step(xs[1], ts[1]).detach().float()))
        torch.cuda.synchronize()
        assert step.graphed == graph, step.failed
        bn = model.bn1.running_mean.detach().clone()
        return losses, opt.master[BF].detach().clone(), bn, int(model.bn1.num_batches_tracked)

    le, we, be, ne = run(False)
    lg, wg, bg, ng = run(True)
    assert all(math.isfinite(v) for v in lg), lg
    assert le == lg, (le, lg)
    assert torch.equal(we, wg)
    assert torch.equal(be, bg) and ne == ng == 10


def test_fused_sgd_matches_torch():
    from distributeddeeplearning_amd.optim import FusedSGD

    torch.manual_seed(1)
    lin = torch.nn.Linear(64, 32).to(DEV)
    ref = torch.nn.Linear(64, 32).to(DEV)
    ref.load_state_dict(lin.state_dict())
    opt = FusedSGD(lin.named_parameters(), lr=0.1, momentum=0.9, weight_decay=1e-3,
                   weight_decay_filter=lambda n, p: True)
    opt_r = torch.optim.SGD(ref.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-3)
    for _ in range(3):
        x = torch.randn(16, 64, device=DEV)
        opt.zero_grad()
        opt_r.zero_grad()
        lin(x).pow(2).mean().backward()
        ref(x).pow(2).mean().backward()
        opt.step()
        opt_r.step()
    assert torch.allclose(lin.weight, ref.weight, rtol=1e-5, atol=1e-6)
    assert torch.allclose(lin.bias, ref.bias, rtol=1e-5, atol=1e-6)


def test_philox_statistics():
    from distributeddeeplearning_amd.ops import kernels

    kernels.require()
    t = torch.empty(1 << 20, dtype=BF, device=DEV)
    kernels.philox_uniform_(t, 42)
    f = t.float()
    assert 0.0 <= float(f.min()) and float(f.max()) < 1.0
    assert abs(float(f.mean()) - 0.5) < 5e-3
    assert abs(float(f.var()) - 1.0 / 12) < 5e-3
    lab = torch.empty(1 << 16, dtype=torch.int64, device=DEV)
    kernels.philox_labels_(lab, 1000, 7)
    assert int(lab.min()) >= 0 and int(lab.max()) < 1000


@pytest.mark.parametrize("one_shot", [False, True])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("n", [2, 8])
def test_xgmi_allreduce_protocol_simulated_ranks(n, dtype, one_shot):
    """K18 on one device: ``n`` simulated ranks (n buffers, n flag pages, one launch whose
    blockIdx.y plays the rank, so all ranks are co-resident) run the two-shot / one-shot
    protocol; uneven input magnitudes, every word checked; three epochs back to back exercise
    flag and ticket reuse."""
    from distributeddeeplearning_amd.ops import kernels

    kernels.require()
    count = 8 * n * 1237
    bufs = [torch.empty(count, dtype=dtype, device=DEV) for _ in range(n)]
    flags = [torch.zeros(32, dtype=torch.int32, device=DEV) for _ in range(n)]
    outs = [torch.empty(count, dtype=dtype, device=DEV) for _ in range(n)]
    tmo = torch.zeros(1, dtype=torch.int32, device=DEV)
    for epoch in (1, 2, 3):
        src = [(torch.randn(count, device=DEV) * (r + 1) * epoch).to(dtype) for r in range(n)]
        for r in range(n):
            bufs[r].copy_(src[r])
        expect = sum(t.float() for t in src) / n
        kernels.xgmi_allreduce([b.data_ptr() for b in bufs], [f.data_ptr() for f in flags],
                               [o.data_ptr() for o in outs] if one_shot else None, count, 0, n,
                               dtype == torch.bfloat16, one_shot, epoch, 1.0 / n, tmo, blocks=8,
                               spin_limit=2_000_000, simulate=True)
        torch.cuda.synchronize()
        assert int(tmo.item()) == 0, "a simulated rank timed out"
        tol = 2e-2 if dtype == torch.bfloat16 else 1e-5
        for r in range(n):
            got = (outs[r] if one_shot else bufs[r]).float()
            assert torch.allclose(got, expect, rtol=tol, atol=tol * float(expect.abs().max())), (epoch, r)


def test_pack_scale_cast_roundtrip():
    from distributeddeeplearning_amd.ops import kernels

    kernels.require()
    ts = [torch.randn(n, device=DEV) for n in (5, 1000, 77)]
    offs = [0, 8, 1016]
    flat = torch.zeros(1100, dtype=BF, device=DEV)
    keep = kernels.pack_scale_cast(ts, flat, offs, scale=0.5)
    torch.cuda.synchronize()
    for t, o in zip(ts, offs):
        assert torch.allclose(flat[o:o + t.numel()].float(), t * 0.5, rtol=1e-2, atol=1e-2)
    back = [torch.empty_like(t) for t in ts]
    keep2 = kernels.pack_scale_cast(back, flat, offs, scale=2.0, unpack=True)
    torch.cuda.synchronize()
    for t, b in zip(ts, back):
        assert torch.allclose(b, t, rtol=1e-2, atol=2e-2)
    del keep, keep2


def test_fused_lars_matches_cpu_reference_path():
    """K15 kernels (segmented norms + LARS update) against the optimizer's plain-PyTorch path."""
    from distributeddeeplearning_amd.optim import FusedLARS

    def run(device):
        torch.manual_seed(5)
        net = torch.nn.Sequential(torch.nn.Linear(64, 48), torch.nn.ReLU(), torch.nn.Linear(48, 16))
        net.to(device)
        opt = FusedLARS(net.named_parameters(), lr=0.2, momentum=0.9, weight_decay=1e-3)
        g = torch.Generator().manual_seed(6)
        for _ in range(3):
            x = torch.randn(32, 64, generator=g).to(device)
            opt.zero_grad()
            net(x).pow(2).mean().backward()
            opt.step()
        return [p.detach().cpu() for p in net.parameters()]

    for a, b in zip(run(DEV), run("cpu")):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-5)


def test_nchw_to_nhwc_normalize():
    from distributeddeeplearning_amd.ops import kernels

    kernels.require()
    x = torch.rand(3, 3, 17, 19, device=DEV)
    mean, sd = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    out = kernels.nchw_to_nhwc_normalize(x, mean, sd)
    ref = (x - torch.tensor(mean, device=DEV).view(1, 3, 1, 1)) / torch.tensor(sd, device=DEV).view(1, 3, 1, 1)
    assert out.shape == x.shape and out.stride(1) == 1
    assert torch.allclose(out.float(), ref, atol=2e-2)
    u8 = (x * 255).to(torch.uint8)
    out8 = kernels.nchw_to_nhwc_normalize(u8, mean, sd, in_scale=1.0 / 255)
    assert torch.allclose(out8.float(), ref, atol=3e-2)


def test_bench_step_with_forced_rccl_collectives():
    """The whole distributed step (arena views, direct gradient writes, bucket launch from the
    listeners, RCCL all-reduce on its stream, event-ordered fused update) with a one-rank RCCL
    communicator (``HVD_FORCE_COLLECTIVES=1``): what the driver's N>1 runs execute, minus peers."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HVD_FORCE_COLLECTIVES="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533",
               HVD_DEBUG="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup",
                          "2", "--batch", "32", "--image", "96"], env=env, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec["n_gpus"] == 1 and rec["ops_backend"] == "hip"
    assert math.isfinite(rec["final_loss"]) and rec["final_loss"] < 12.0


def test_fused_stem_matches_composition():
    """StemFn (conv 7x7/2 -> BN -> ReLU -> maxpool in one node, activation never written) against
    the same computation composed from the separate HIP ops."""
    hip = _hip()
    for (B, H, W) in [(5, 64, 96), (3, 70, 90), (2, 66, 58)]:   # conv output 32x48, 35x45 (odd: partial 2x2 patches), 33x29
        _fused_stem_case(hip, B, H, W)


def _fused_stem_case(hip, B, H, W):
    hip.reset_caches()
    x = torch.rand((B, 3, H, W), device=DEV).to(BF).contiguous(memory_format=CL)
    w0 = (torch.randn((64, 3, 7, 7), device=DEV) / math.sqrt(147)).to(BF).contiguous(memory_format=CL)
    g0 = torch.rand(64, device=DEV) + 0.5
    b0 = torch.randn(64, device=DEV) * 0.2
    Hc, Wc = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    Hp, Wp = (Hc + 2 - 3) // 2 + 1, (Wc + 2 - 3) // 2 + 1
    gout = rand_nhwc(B, 64, Hp, Wp, seed=51)

    def run(fused):
        w = w0.clone().requires_grad_(True)
        gamma, beta = g0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
        rm, rv = torch.zeros(64, device=DEV), torch.ones(64, device=DEV)
        if fused:
            out = hip.stem(x, w, gamma, beta, rm, rv, 0.1, 1e-5)
        else:
            y, st = hip.conv2d_stats(x, w, 2, 3, True)
            z = hip.batchnorm_act(y, gamma, beta, rm, rv, True, 0.1, 1e-5, True, None, stats=st)
            out = hip.maxpool3x3s2(z)
        out.backward(gout)
        return out.detach(), w.grad, gamma.grad, beta.grad, rm, rv

    a, b = run(False), run(True)
    assert a[0].shape == (B, 64, Hp, Wp)
    assert torch.equal(a[0], b[0])
    for i in (1, 2, 3):
        assert rms_relerr(b[i], a[i]) < 1e-2, i
    assert torch.allclose(a[4], b[4]) and torch.allclose(a[5], b[5])


def test_wgrad_side_stream_gives_identical_gradients():
    """Weight gradients computed on the second HIP stream (direct-gradient mode) must equal the
    single-stream result bit for bit, with the streams joined by the end of ``backward()``."""
    from distributeddeeplearning_amd.models import resnet18
    from distributeddeeplearning_amd.ops import functional as ops
    from distributeddeeplearning_amd.parallel.bucket import GradArena

    hip = _hip()
    x = torch.rand((8, 3, 64, 64), device=DEV).to(BF).contiguous(memory_format=CL)
    t = torch.randint(0, 10, (8,), device=DEV)

    def run(side):
        hip.reset_caches()
        torch.manual_seed(11)
        m = resnet18(num_classes=16).to_compute(torch.device(DEV), BF).train()
        arena = GradArena(list(m.named_parameters()), 1 << 20, 1 << 20)
        hip.set_wgrad_side_stream(side)
        hip.set_direct_grads(True)
        try:
            for _ in range(2):  # second pass accumulates on top of the first
                ops.softmax_cross_entropy(m(x), t).backward()
        finally:
            hip.set_direct_grads(False)
            hip.set_wgrad_side_stream(True)
        # no explicit synchronisation here: reading on the current stream must already be safe
        return {n: p.grad.detach().float().clone() for n, p in m.named_parameters()}, arena

    (a, _), (b, _) = run(False), run(True)
    assert a.keys() == b.keys()
    for n in a:
        assert torch.equal(a[n], b[n]), n


def test_topk_label_rank_kernel():
    """K21: top-1 / top-5 counts from the label-rank kernel against torch.topk (ties included)."""
    from distributeddeeplearning_amd.ops import functional as ops

    _hip()
    torch.manual_seed(4)
    logits = (torch.randn(301, 1000, device=DEV) * 2).to(BF)
    logits[:, 100:110] = logits[:, 100:101]  # runs of exactly equal values
    target = torch.randint(0, 1000, (301,), device=DEV)
    target[:40] = logits[:40].float().argmax(dim=1)
    target[40:80] = torch.randint(100, 110, (40,), device=DEV)
    got = ops.topk_correct(logits, target, ks=(1, 5, 10))
    _, pred = logits.float().topk(10, dim=1)
    hit = pred.eq(target.view(-1, 1))
    want = [int(hit[:, :k].any(dim=1).sum()) for k in (1, 5, 10)]
    # torch.topk's order among exactly equal values is not specified; the kernel breaks ties by
    # index, so allow the tied rows to differ
    tied = int(((target >= 100) & (target < 110)).sum())
    assert got[0] >= 40
    for g, w in zip(got, want):
        assert abs(g - w) <= tied, (got, want)


def test_forced_rccl_collectives_api_surface():
    """Every hvd.* collective on a real one-rank RCCL communicator (``HVD_FORCE_COLLECTIVES=1``):
    broadcast_parameters / optimizer state, allreduce (average, max), allgather, broadcast,
    object broadcast, barrier — the calls the N>1 bench makes before and after its timed loop."""
    import os
    import subprocess
    import sys

    code = r"""
import os, torch
import distributeddeeplearning_amd.torch as hvd
from distributeddeeplearning_amd.models import resnet18
from distributeddeeplearning_amd.optim import FusedSGD
hvd.init()
assert hvd.size() == 1 and hvd.rank() == 0
dev = torch.device('cuda', torch.cuda.current_device())
m = resnet18(num_classes=8).to_compute(dev, torch.bfloat16)
before = {k: v.clone() for k, v in m.state_dict().items()}
hvd.broadcast_parameters(m.state_dict(), root_rank=0)
for k, v in m.state_dict().items():
    assert torch.equal(v, before[k]), k
opt = FusedSGD(m.named_parameters(), lr=0.1, momentum=0.9)
hvd.broadcast_optimizer_state(opt, root_rank=0)
t = torch.arange(10, dtype=torch.float32, device=dev)
assert torch.equal(hvd.allreduce(t, name='a'), t)
assert torch.equal(hvd.allreduce(t.double(), op=hvd.Max, name='b'), t.double())
assert torch.equal(hvd.allgather(t.view(5, 2), name='c'), t.view(5, 2))
assert torch.equal(hvd.broadcast(t.to(torch.bfloat16), 0, name='d'), t.to(torch.bfloat16))
h = hvd.allreduce_async_(t.clone(), name='e'); hvd.synchronize(h)
assert hvd.broadcast_object({'epoch': 3}, 0) == {'epoch': 3}
hvd.barrier()
hvd.shutdown()
print('OK')
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HVD_FORCE_COLLECTIVES="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29534",
               HVD_DEBUG="1", PYTHONPATH=root)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True,
                         timeout=300, cwd=root)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stderr[-3000:]


def test_device_prefetcher_uploads_and_converts_on_a_copy_stream():
    """data/prefetch.py on the GPU: float and uint8 NCHW host batches arrive as channels_last bf16
    (kernel K20 on the copy stream), in order, and the consumer needs no host sync."""
    from distributeddeeplearning_amd.data import DevicePrefetcher
    from distributeddeeplearning_amd.utils.config import RGB_MEAN, RGB_SD

    g = torch.Generator().manual_seed(3)
    f32 = [(torch.randn(6, 3, 32, 24, generator=g), torch.arange(6) + 10 * i) for i in range(5)]
    got = []
    for data, target in DevicePrefetcher(f32, DEV, dtype=BF, depth=2):
        assert data.is_cuda and data.dtype == BF and data.is_contiguous(memory_format=CL)
        got.append(((data.float() * 2).cpu(), target.cpu()))  # consume on the current stream
    assert len(got) == 5
    for (a, b), (c, d) in zip(f32, got):
        assert torch.equal(b, d)
        assert torch.allclose(c / 2, a.to(BF).float())
    u8 = [(torch.randint(0, 256, (4, 3, 16, 16), dtype=torch.uint8, generator=g), torch.arange(4))
          for _ in range(3)]
    mean = torch.tensor(RGB_MEAN).view(1, 3, 1, 1)
    sd = torch.tensor(RGB_SD).view(1, 3, 1, 1)
    for (a, _), (c, _) in zip(u8, DevicePrefetcher(u8, DEV, dtype=BF, depth=3)):
        want = (a.float() / 255.0 - mean) / sd
        assert rms_relerr(c.float().cpu(), want) < 5e-3
    # fp32 compute dtype: no kernel, same values
    for (a, _), (c, _) in zip(u8, DevicePrefetcher(u8, DEV, dtype=torch.float32)):
        assert torch.allclose(c.cpu(), (a.float() / 255.0 - mean) / sd, atol=1e-5)


@pytest.mark.parametrize("shape", [(1000, 256, 64, 64), (6272, 1024, 256, 256), (300, 2048, 512, 512)])
def test_conv_igemm_two_segment_gemm_with_bias(shape):
    """conv_igemm DUAL: y = x w^T + x2 w2^T + bias in one launch (the input gradient of the
    BatchNorm-by-linearity path), every tile configuration the bottlenecks use."""
    from distributeddeeplearning_amd.ops import kernels as K

    _hip()
    M, C, C2, N = shape
    g = torch.Generator(device=DEV).manual_seed(M)
    x = torch.randn((M, C), generator=g, device=DEV).to(BF)
    x2 = torch.randn((M, C2), generator=g, device=DEV).to(BF)
    w = (torch.randn((N, C), generator=g, device=DEV) / math.sqrt(C)).to(BF)
    w2 = (torch.randn((N, C2), generator=g, device=DEV) / math.sqrt(C2)).to(BF)
    bias = torch.randn(N, generator=g, device=DEV)
    y = torch.empty((M, N), dtype=BF, device=DEV)
    K.conv_igemm(x, (M, 1, 1, C, C, C, C, x.numel()), w, y, N, 1, 1, 1, 0, 0, 1, 1, 1,
                 x2=x2, w2=w2, c2=C2, bias=bias)
    ref = x.float() @ w.float().t() + x2.float() @ w2.float().t() + bias
    assert relerr(y, ref) < 1.5e-2 and rms_relerr(y, ref) < 4e-3


def test_bn3_by_linearity_equals_materialised_gradient():
    """The same chain of three bottlenecks with DDL_BN3_LINEAR on and off: input gradient, every
    weight gradient and every BatchNorm gradient agree to bf16 accuracy (the first two blocks take
    the linear path, the last one - unmasked gradient - never does)."""
    from distributeddeeplearning_amd.models.resnet import Bottleneck, _block_spec
    from distributeddeeplearning_amd.ops import functional as ops

    hip = _hip()

    def make():
        torch.manual_seed(7)
        ms = [Bottleneck(128, 64, 2, True, False), Bottleneck(256, 64, 1, False, False),
              Bottleneck(256, 64, 1, False, False)]
        for m in ms:
            m.to(DEV).train()
            for mod in m.modules():
                if hasattr(mod, "kernel"):
                    mod.weight.data = mod.weight.data.to(BF).contiguous(memory_format=CL)
        return ms

    x0 = torch.relu(rand_nhwc(8, 128, 28, 28, seed=51))
    gout = rand_nhwc(8, 256, 14, 14, seed=52)
    used = {"n": 0}
    real = hip._bn3_linear_dgrad

    def counting(*a, **k):
        used["n"] += 1
        return real(*a, **k)

    def run(linear):
        hip.reset_caches()
        ms = make()
        x = x0.detach().clone().requires_grad_(True)
        hip._BN3_LINEAR["on"] = linear
        hip._bn3_linear_dgrad = counting
        saved = hip._BN3_LINEAR["max_c2"], hip._DS_LINEAR["max_cin"]
        hip._BN3_LINEAR["max_c2"] = hip._DS_LINEAR["max_cin"] = 4096    # every eligible BatchNorm, whatever its width
        try:
            specs = [_block_spec(m.pairs(), m.downsample) for m in ms]
            out = ops.residual_chain(x, specs, True, 0.1, 1e-5)
            out.backward(gout)
            torch.cuda.synchronize()
        finally:
            hip._BN3_LINEAR["on"] = True
            hip._bn3_linear_dgrad = real
            hip._BN3_LINEAR["max_c2"], hip._DS_LINEAR["max_cin"] = saved
        names = [n for m in ms for n, _ in m.named_parameters()]
        return x.grad, names, [p.grad.float() for m in ms for p in m.parameters()]

    dx_a, names, g_a = run(True)
    assert used["n"] == 3, used     # two main-path BatchNorms + the first block's projection shortcut
    dx_b, _, g_b = run(False)
    assert used["n"] == 3
    assert rms_relerr(dx_a, dx_b) < 1e-2
    for n, a, b in zip(names, g_a, g_b):
        assert rms_relerr(a, b) < 1.5e-2, (n, rms_relerr(a, b))


@pytest.mark.parametrize("beta2", [0.0, 2.0])
def test_linearity_path_against_fp32_reference(beta2):
    """(``beta2``: bias of every block's second BatchNorm - with 2.0 the narrow activation the Gram
    matrix is built from has a mean of about twice its standard deviation, the case the centred
    Gram matrix exists for.)  A chain of three bottlenecks (projection + two identity blocks) with EVERY eligible BatchNorm
    backward taken by linearity, against the same blocks in fp32 ATen; error budget = twice the
    error of the stock bf16 ATen path against that reference, as for the single block."""
    from distributeddeeplearning_amd.models.resnet import Bottleneck, _block_spec
    from distributeddeeplearning_amd.ops import functional as ops

    hip = _hip()
    hip.reset_caches()

    def make(dtype):
        torch.manual_seed(3)
        ms = [Bottleneck(128, 64, 2, True, False), Bottleneck(256, 64, 1, False, False),
              Bottleneck(256, 64, 1, False, False)]
        for m in ms:
            m.to(DEV).train()
            for mod in m.modules():
                if hasattr(mod, "kernel"):
                    wq = mod.weight.data.to(BF)
                    mod.weight.data = wq.to(dtype).contiguous(memory_format=CL)
            m.bn2.bias.data.fill_(beta2)
        return ms

    x0 = torch.relu(rand_nhwc(8, 128, 28, 28, seed=61))
    gout = rand_nhwc(8, 256, 14, 14, seed=62)

    def run(backend, dtype, chained):
        ms = make(dtype)
        x = x0.detach().to(dtype).requires_grad_(True)
        ops.set_backend(backend)
        saved = hip._BN3_LINEAR["max_c2"], hip._DS_LINEAR["max_cin"]
        hip._BN3_LINEAR["max_c2"] = hip._DS_LINEAR["max_cin"] = 4096
        try:
            if chained:
                out = ops.residual_chain(x, [_block_spec(m.pairs(), m.downsample) for m in ms], True, 0.1, 1e-5)
            else:
                out = x
                for m in ms:
                    out = m(out)
            out.backward(gout.to(dtype))
            torch.cuda.synchronize()
        finally:
            ops.set_backend("auto")
            hip._BN3_LINEAR["max_c2"], hip._DS_LINEAR["max_cin"] = saved
        return x.grad.float(), [p.grad.float() for m in ms for p in m.parameters()], \
            [n for m in ms for n, _ in m.named_parameters()]

    dx_r, g_r, names = run("aten", torch.float32, False)
    dx_a, g_a, _ = run("aten", BF, False)
    dx_h, g_h, _ = run("auto", BF, True)

    def ok(h, a, r, what):
        eh, ea = rms_relerr(h, r), rms_relerr(a, r)
        assert eh < max(2.0 * ea, 2e-2), "%s: hip (linearity) %.4f vs stock bf16 %.4f" % (what, eh, ea)

    ok(dx_h, dx_a, dx_r, "dx")
    for n, h, a, r in zip(names, g_h, g_a, g_r):
        ok(h, a, r, n)


def test_training_is_bit_reproducible_and_independent_of_the_side_stream():
    """Four ResNet-50 steps at a size where the weight-gradient stream runs well behind the main
    stream: two runs with the side stream and one without give identical losses and weights.  (A
    tensor the side stream still reads after its autograd node returned - found this way: the saved
    BatchNorm mean of the linearity path - shows up as run-to-run noise.)"""
    from distributeddeeplearning_amd.models import resnet50
    from distributeddeeplearning_amd.ops import functional as ops
    from distributeddeeplearning_amd.optim import FusedSGD

    hip = _hip()
    g = torch.Generator(device="cpu").manual_seed(9)
    x = torch.rand((64, 3, 160, 160), generator=g).to(DEV).to(BF).contiguous(memory_format=CL)
    y = torch.randint(0, 1000, (64,), generator=g).to(DEV)

    def run(side):
        hip.reset_caches()
        hip.set_wgrad_side_stream(side)
        try:
            torch.manual_seed(0)
            model = resnet50(num_classes=1000).to_compute(torch.device(DEV), BF).train()
            opt = FusedSGD(model.named_parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4)
            losses = []
            for _ in range(4):
                opt.zero_grad()
                loss = ops.softmax_cross_entropy(model(x), y)
                loss.backward()
                opt.step()
                losses.append(float(loss.detach()))
            torch.cuda.synchronize()
            return losses, opt.master[BF].clone()
        finally:
            hip.set_wgrad_side_stream(True)

    la, wa = run(True)
    lb, wb = run(True)
    lc, wc = run(False)
    assert la == lb == lc, (la, lb, lc)
    assert torch.equal(wa, wb) and torch.equal(wa, wc)


@pytest.mark.parametrize("shape", [(3, 64, 8, 8), (2, 256, 7, 9), (4, 8, 1, 5)])
def test_subsample2_is_the_strided_slice(shape):
    """ddl_subsample2 (input of the stride-2 projections, csrc/kernels/step_aux.hip) against slicing."""
    from distributeddeeplearning_amd.ops import kernels as K

    x = torch.randn(shape, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    out = K.subsample2(x)
    ref = x[:, :, ::2, ::2]
    assert out.shape == ref.shape and out.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(out, ref)


@pytest.mark.parametrize("C,C2", [(256, 64), (512, 128), (1024, 256), (96, 64)])
def test_bn_lin_prep_q_matches_the_two_launch_form(C, C2):
    """One-launch weight preparation of the BatchNorm-by-linearity path (b1, bias, Q) against fp32 math
    and against the two launches it replaces (scaling kernel + a GEMM of the conv kernel)."""
    from distributeddeeplearning_amd.ops import kernels as K

    K.require()
    g = torch.Generator(device=DEV).manual_seed(C + C2)
    wt = (torch.randn((C2, C), generator=g, device=DEV) / math.sqrt(C2)).to(BF)
    coef = torch.randn((3, C), generator=g, device=DEV)
    b1 = torch.empty((C2, C), dtype=BF, device=DEV)
    q = torch.empty((C2, C2), dtype=BF, device=DEV)
    bias = torch.empty(C2, dtype=torch.float32, device=DEV)
    K.bn_lin_prep_q(wt, C, coef.data_ptr(), C, C2, b1, q, bias)
    w = wt.float()
    assert torch.equal(b1, (w * coef[0]).to(BF))
    assert torch.allclose(bias, (w * coef[2]).sum(1), rtol=1e-4, atol=1e-4)
    a2 = (w * coef[1]).to(BF).float()
    qr = a2 @ w.t()
    assert relerr(q, qr) < 6e-3
    if C % 64 == 0:   # the conv kernel's GEMM wants 64-channel multiples here
        b1o, a2o, qo = torch.empty_like(b1), torch.empty_like(b1), torch.empty_like(q)
        biaso = torch.empty_like(bias)
        K.bn_lin_prep(wt, C, coef.data_ptr(), C, C2, b1o, a2o, biaso)
        K.conv_igemm(a2o, (C2, 1, 1, C, C, C, C, a2o.numel()), wt, qo, C2, 1, 1, 1, 0, 0, 1, 1, 1)
        assert torch.equal(b1, b1o)
        assert torch.allclose(bias, biaso, rtol=1e-5, atol=1e-5)
        assert relerr(q, qo) < 6e-3


@pytest.mark.parametrize("B,H", [(3, 56), (2, 6)])
def test_conv_rows3x3_forward_and_masked_backward(B, H):
    """Row-streaming 3x3 kernel (64 -> 64 channels, 56-pixel rows): forward with per-image statistics, and the
    backward-data form that masks with relu(bn(y_prev)) and keeps the BatchNorm-backward sums, against fp32."""
    from distributeddeeplearning_amd.ops import hip_ops as hip, kernels as K

    K.require()
    if not K.conv_rows3x3_ok(64, 64, 3, 3, 1, 1, H, 56):
        pytest.skip("row-streaming kernel switched off")
    x = rand_nhwc(B, 64, H, 56, seed=11)
    w = (torch.randn((64, 64, 3, 3), device=DEV) / 24.0).to(BF).contiguous(memory_format=CL)
    y = torch.empty_like(x)
    stats = torch.full((B, 2, 64), float("nan"), device=DEV)
    K.conv_rows3x3(x, w, y, stats=stats)
    yr = F.conv2d(x.float(), w.float(), None, 1, 1)
    assert relerr(y, yr) < 1.5e-2 and rms_relerr(y, yr) < 4e-3
    yf = y.float()
    assert torch.allclose(stats[:, 0], yf.sum((2, 3)), rtol=1e-3, atol=1e-2 * float(yf.abs().max()))
    assert torch.allclose(stats[:, 1], (yf * yf).sum((2, 3)), rtol=1e-3, atol=1e-3)
    # backward-data: dy -> masked gradient of relu(bn(y_prev)); the filter operand is the flipped transpose
    dy = rand_nhwc(B, 64, H, 56, seed=12)
    y_prev = rand_nhwc(B, 64, H, 56, seed=13)
    fco = torch.stack([torch.rand(64, device=DEV) + 0.5, torch.randn(64, device=DEV) * 0.3]).contiguous()
    g, part = hip._conv_dgrad_bn(dy, w, x.shape, 1, 1, y_prev, fco.data_ptr())
    assert part.shape == (B, 2, 64)
    dxr = torch.nn.grad.conv2d_input(x.shape, w.float(), dy.float(), 1, 1)
    on = (y_prev.float() * fco[0].view(1, -1, 1, 1) + fco[1].view(1, -1, 1, 1)) > 0
    # (elements within rounding distance of the ReLU threshold may flip: compare where the decision is clear)
    clear = (y_prev.float() * fco[0].view(1, -1, 1, 1) + fco[1].view(1, -1, 1, 1)).abs() > 1e-3
    gr = dxr * on
    assert relerr(g.float() * clear, gr * clear) < 1.5e-2
    gf = g.float()
    assert torch.allclose(part[:, 0], gf.sum((2, 3)), rtol=1e-3, atol=1e-2 * float(gf.abs().max()))
    assert torch.allclose(part[:, 1], (gf * y_prev.float()).sum((2, 3)), rtol=1e-3, atol=1e-2 * float(gf.abs().max()))
    hip.reset_caches()


@pytest.mark.parametrize("B,C2,hw", [(6, 64, 56), (5, 128, 28), (24, 256, 14)])
def test_gram_weight_gradient_emits_column_sums(B, C2, hw):
    """The Gram launch of the BatchNorm-by-linearity path (z2^T z2 as a plain-GEMM weight gradient) sums the columns
    of z2 on the side: slabs and column sums against fp32."""
    from distributeddeeplearning_amd.ops import hip_ops as hip, kernels as K

    K.require()
    M = B * hw * hw
    if not K.conv_wgrad_colsum_ok(M, C2, C2):
        pytest.skip("column sums switched off or no instantiation for this tile")
    z2 = (rand_nhwc(B, C2, hw, hw, seed=21).float().relu() + 0.25).to(BF).contiguous(memory_format=CL)
    s2 = K.conv_wgrad_splits(M, C2, C2, 1, 1)
    slabs = torch.empty((s2, C2, C2), dtype=torch.float32, device=DEV)
    cs = torch.full((s2, 2, C2), float("nan"), device=DEV)
    K.conv_wgrad(z2, z2, hip._geom(z2), slabs, C2, 1, 1, 1, 0, 0, hw, hw, s2, colsum=cs)
    zf = z2.float().permute(0, 2, 3, 1).reshape(M, C2)
    assert torch.allclose(cs[:, 0].sum(0), zf.sum(0), rtol=1e-4)
    assert torch.equal(cs[:, 0], cs[:, 1])
    assert relerr(slabs.sum(0), zf.t() @ zf) < 1e-3


def test_conv_rows3x3_applies_the_batchnorm_in_front_of_it():
    """Row-streaming 3x3 kernel with the input transform: x is the raw output of the previous conv, the kernel applies
    scale / shift / ReLU to the staged rows, convolves them and writes them out - against the separate apply pass."""
    from distributeddeeplearning_amd.ops import kernels as K

    K.require()
    B, H = 3, 56
    if not K.conv_rows3x3_ok(64, 64, 3, 3, 1, 1, H, 56):
        pytest.skip("row-streaming kernel switched off")
    y0 = rand_nhwc(B, 64, H, 56, seed=31)
    w = (torch.randn((64, 64, 3, 3), device=DEV) / 24.0).to(BF).contiguous(memory_format=CL)
    fco = torch.stack([torch.rand(64, device=DEV) + 0.5, torch.randn(64, device=DEV) * 0.3]).contiguous()
    # reference: the library's own apply pass, then the plain row-streaming conv
    z_ref = torch.empty_like(y0)
    K.bn_apply_fwd(y0, fco[0], fco[1], None, z_ref, B * H * 56, 64, True)
    y_ref = torch.empty_like(y0)
    st_ref = torch.empty((B, 2, 64), device=DEV)
    K.conv_rows3x3(z_ref, w, y_ref, stats=st_ref)
    z = torch.full_like(y0, float("nan"))
    y = torch.full_like(y0, float("nan"))
    st = torch.full((B, 2, 64), float("nan"), device=DEV)
    K.conv_rows3x3(y0, w, y, stats=st, in_fcoef=fco, z_out=z)
    assert torch.equal(z, z_ref)          # same rounding as the apply kernel: the backward mask is re-derived from it
    assert torch.equal(y, y_ref)
    assert torch.equal(st, st_ref)
    # ... and without writing the activated tensor at all
    y2 = torch.full_like(y0, float("nan"))
    K.conv_rows3x3(y0, w, y2, stats=st, in_fcoef=fco)
    assert torch.equal(y2, y_ref)
    # the row-streaming weight gradient applies the same transform to the rows it stages
    from distributeddeeplearning_amd.ops import hip_ops as hip

    if K.conv_wgrad_rows_ok(64, 64, 3, 3, H, 56, 1, B):
        dy = rand_nhwc(B, 64, H, 56, seed=32)
        splits = K.conv_wgrad_splits(B * H * 56, 64, 64, 3, 3, H, 56, 1, B)
        s_ref = torch.empty((splits, 64 * 9 * 64), device=DEV)
        s_xf = torch.full_like(s_ref, float("nan"))
        K.conv_wgrad(dy, z_ref, hip._geom(z_ref), s_ref, 64, 3, 3, 1, 1, 1, H, 56, splits)
        K.conv_wgrad(dy, y0, hip._geom(y0), s_xf, 64, 3, 3, 1, 1, 1, H, 56, splits, x_fcoef=fco)
        assert torch.equal(s_xf, s_ref)


def test_row_streaming_kernels_form_the_batchnorm_backward_gradient_themselves():
    """dy = ca*g + cb*y + cc (a BatchNorm's backward apply) formed inside the row-streaming backward-data and
    weight-gradient kernels on the rows they stage, against the separate apply pass followed by the plain kernels:
    bit-identical (same operation order, same bf16 rounding of dy)."""
    from distributeddeeplearning_amd.ops import hip_ops as hip
    from distributeddeeplearning_amd.ops import kernels as K

    K.require()
    B, H = 3, 56
    if not (K.conv_rows3x3_ok(64, 64, 3, 3, 1, 1, H, 56) and K.conv_wgrad_rows_ok(64, 64, 3, 3, H, 56, 1, B)):
        pytest.skip("row-streaming kernels switched off")
    M = B * H * 56
    g2 = rand_nhwc(B, 64, H, 56, seed=41)
    y2 = rand_nhwc(B, 64, H, 56, seed=42)
    y_prev = rand_nhwc(B, 64, H, 56, seed=43)
    x_raw = rand_nhwc(B, 64, H, 56, seed=44)
    wt = (torch.randn((64, 64, 3, 3), device=DEV) / 24.0).to(BF).contiguous(memory_format=CL)
    coef = torch.stack([torch.rand(64, device=DEV) + 0.5, torch.randn(64, device=DEV) * 0.2,
                        torch.randn(64, device=DEV) * 0.1]).contiguous()
    fco = torch.stack([torch.rand(64, device=DEV) + 0.5, torch.randn(64, device=DEV) * 0.3]).contiguous()
    dy = torch.empty_like(g2)
    K.bn_bwd_apply(g2, y2, None, None, coef.data_ptr(), dy, None, M, 64, 0)
    # backward-data: masked gradient and the partial sums of the BatchNorm in front
    g_ref = torch.empty_like(g2)
    p_ref = torch.empty((B, 2, 64), device=DEV)
    K.conv_rows3x3(dy, wt, g_ref, stats=p_ref, ep_y=y_prev, ep_fcoef=fco)
    g = torch.full_like(g2, float("nan"))
    p = torch.full((B, 2, 64), float("nan"), device=DEV)
    K.conv_rows3x3(g2, wt, g, stats=p, ep_y=y_prev, ep_fcoef=fco, dy_y=y2, dy_coef=coef)
    assert torch.equal(g, g_ref)
    assert torch.equal(p, p_ref)
    # weight gradient, with a plain input and with an input that is transformed as well
    splits = K.conv_wgrad_splits(M, 64, 64, 3, 3, H, 56, 1, B)
    z = torch.empty_like(x_raw)
    K.bn_apply_fwd(x_raw, fco[0], fco[1], None, z, M, 64, True)
    s_ref = torch.empty((splits, 64 * 9 * 64), device=DEV)
    K.conv_wgrad(dy, z, hip._geom(z), s_ref, 64, 3, 3, 1, 1, 1, H, 56, splits)
    s_a = torch.full_like(s_ref, float("nan"))
    K.conv_wgrad(g2, z, hip._geom(z), s_a, 64, 3, 3, 1, 1, 1, H, 56, splits, dy_y=y2, dy_coef=coef)
    assert torch.equal(s_a, s_ref)
    s_b = torch.full_like(s_ref, float("nan"))
    K.conv_wgrad(g2, x_raw, hip._geom(x_raw), s_b, 64, 3, 3, 1, 1, 1, H, 56, splits, x_fcoef=fco, dy_y=y2, dy_coef=coef)
    assert torch.equal(s_b, s_ref)
    # shapes the kernel does not serve refuse the extra operands instead of ignoring them
    small = rand_nhwc(2, 64, 28, 28, seed=45)
    with pytest.raises(K.KernelError):
        sp = K.conv_wgrad_splits(2 * 28 * 28, 64, 64, 3, 3, 28, 28, 1, 2)
        K.conv_wgrad(small, small, hip._geom(small), torch.empty((sp, 64 * 9 * 64), device=DEV), 64, 3, 3, 1, 1, 1,
                     28, 28, sp, dy_y=small, dy_coef=coef)


@pytest.mark.parametrize("kind", ["bottleneck", "basic"])
def test_blocks_with_the_backward_apply_folded_into_the_row_kernels_are_bit_identical(kind):
    """A chain of stage-1 blocks with DDL_CONV_ROWS_BN_BWD on and off: every gradient is bit-identical, and the
    folded path is really taken (once per block whose incoming gradient is pre-masked by the next block)."""
    from distributeddeeplearning_amd.models.resnet import BasicBlock, Bottleneck, _block_spec
    from distributeddeeplearning_amd.ops import functional as ops

    hip = _hip()

    def make():
        torch.manual_seed(11)
        if kind == "bottleneck":
            ms = [Bottleneck(64, 64, 1, True, False), Bottleneck(256, 64, 1, False, False),
                  Bottleneck(256, 64, 1, False, False)]
        else:
            ms = [BasicBlock(64, 64, 1, False, False), BasicBlock(64, 64, 1, False, False),
                  BasicBlock(64, 64, 1, False, False)]
        for m in ms:
            m.to(DEV).train()
            for mod in m.modules():
                if hasattr(mod, "kernel"):
                    mod.weight.data = mod.weight.data.to(BF).contiguous(memory_format=CL)
        return ms

    x0 = torch.relu(rand_nhwc(4, 64, 56, 56, seed=71))
    gout = rand_nhwc(4, 256 if kind == "bottleneck" else 64, 56, 56, seed=72)
    used = {"n": 0}
    real = hip._unit_coef

    def counting(*a, **k):
        used["n"] += 1
        return real(*a, **k)

    def run(on):
        hip.reset_caches()
        ms = make()
        x = x0.detach().clone().requires_grad_(True)
        saved = hip._ROWS_BN_BWD
        hip._ROWS_BN_BWD = on
        hip._unit_coef = counting
        try:
            out = ops.residual_chain(x, [_block_spec(m.pairs(), m.downsample) for m in ms], True, 0.1, 1e-5)
            out.backward(gout)
            torch.cuda.synchronize()
        finally:
            hip._ROWS_BN_BWD = saved
            hip._unit_coef = real
        names = [n for m in ms for n, _ in m.named_parameters()]
        return x.grad, names, [p.grad.clone() for m in ms for p in m.parameters()]

    used["n"] = 0
    dx_b, names, g_b = run(0)
    base = used["n"]                   # (the linearity path of the bottlenecks uses the same helper)
    used["n"] = 0
    dx_a, _, g_a = run(1)
    assert used["n"] - base == (3 if kind == "bottleneck" else 2), (used, base)
    assert torch.equal(dx_a, dx_b)
    for n, a, b in zip(names, g_a, g_b):
        assert torch.equal(a, b), n


@pytest.mark.parametrize("M,N,Kd", [(64 * 196, 1024, 256), (65 * 196, 1024, 1024), (300, 256, 64)])
def test_gemm256_kernel_against_the_tile_kernel_and_fp32(M, N, Kd):
    """gemm256.hip (256 x 256 tile, 8 waves, LDS-DMA ring; opt-in for the wide 1x1 layers, DDL_GEMM256=1) called
    directly: outputs bit-identical to the 128 x 128 tile kernel (same reduction order), statistics table equal to
    rounding, both within bf16 accuracy of fp32; ragged last tile and last statistics row."""
    from distributeddeeplearning_amd.ops import hip_ops as hip
    from distributeddeeplearning_amd.ops import kernels as K

    K.require()
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn((M, Kd), generator=g, device=DEV).to(BF)
    w = (torch.randn((N, Kd), generator=g, device=DEV) / math.sqrt(Kd)).to(BF)
    rows = (M + 127) // 128
    y = torch.full((M, N), float("nan"), dtype=BF, device=DEV)
    st = torch.full((rows, 2, N), float("nan"), device=DEV)
    K.gemm256_fwd(x, w, y, st)
    ref = x.float() @ w.float().t()
    assert relerr(y, ref) < 1.5e-2 and rms_relerr(y, ref) < 4e-3
    yf = y.float()
    pad = rows * 128 - M
    yp = torch.cat([yf, torch.zeros((pad, N), device=DEV)]) if pad else yf
    assert torch.allclose(st[:, 0, :], yp.view(rows, 128, N).sum(1), rtol=1e-3, atol=1e-2)
    assert torch.allclose(st[:, 1, :], (yp * yp).view(rows, 128, N).sum(1), rtol=1e-3, atol=1e-2)
    # the tile kernel on the same operands (as a 1x1 conv over an [M, Kd, 1, 1] image)
    x4 = x.view(M, Kd, 1, 1).contiguous(memory_format=CL)
    y_t = torch.empty((M, N), dtype=BF, device=DEV)
    st_t = torch.empty((K.conv_tiles_m(M), 2, N), device=DEV)
    K.conv_igemm(x4, hip._geom(x4), w, y_t, N, 1, 1, 1, 0, 0, 1, 1, 1, stats=st_t)
    assert torch.equal(y, y_t)
    assert torch.allclose(st, st_t, rtol=1e-4, atol=1e-3)
    y2 = torch.full((M, N), float("nan"), dtype=BF, device=DEV)
    K.gemm256_fwd(x, w, y2)          # without statistics
    assert torch.equal(y2, y)
    with pytest.raises(K.KernelError):
        K.gemm256_fwd(x, w[:128].contiguous(), torch.empty((M, 128), dtype=BF, device=DEV))   # N % 256 != 0


def test_raw_addresses_handed_to_the_weight_gradient_stream_are_kept_alive():
    """The weight-gradient stream reads its operands after the autograd node has returned and released its saved
    tensors.  Every raw address among them must therefore come with its owner in ``keep`` (found as a 1-in-1000-steps
    run-to-run difference of ResNet-101: the saved BatchNorm coefficients behind ``x_fcoef`` were recycled by the main
    stream before the queued weight gradient had read them; profiles/r3_notes.md section 22).  One real training
    step of ResNet-18 (gradient arenas, side stream) with a spy on the launcher."""
    import distributeddeeplearning_amd.torch as hvd
    from distributeddeeplearning_amd.models import resnet18
    from distributeddeeplearning_amd.ops import functional as ops
    from distributeddeeplearning_amd.optim import FusedSGD

    hip = _hip()
    if hip._ROWS_BN_IN != 2:
        pytest.skip("the activated tensor is materialised: no raw coefficient address reaches the weight gradient")
    hvd.init()
    torch.manual_seed(13)
    model = resnet18(num_classes=64).to_compute(torch.device(DEV), BF).train()
    opt = hvd.DistributedOptimizer(FusedSGD(model.named_parameters(), lr=0.01, momentum=0.9),
                                   named_parameters=model.named_parameters())
    seen = []
    real = hip._conv_wgrad

    def spy(dy, x, w, stride, padding, prewaited=False, x_fcoef=None, dy_y=None, dy_coef=None, keep=()):
        if x_fcoef is not None:
            addr = x_fcoef if isinstance(x_fcoef, int) else x_fcoef.data_ptr()
            seen.append(any(torch.is_tensor(t) and t.data_ptr() <= addr < t.data_ptr() + t.numel() * t.element_size()
                            for t in keep))
        return real(dy, x, w, stride, padding, prewaited=prewaited, x_fcoef=x_fcoef, dy_y=dy_y, dy_coef=dy_coef, keep=keep)

    hip._conv_wgrad = spy
    try:
        x = torch.rand((4, 3, 224, 224), device=DEV).to(BF).contiguous(memory_format=CL)
        y = torch.randint(0, 64, (4,), device=DEV)
        opt.zero_grad()
        loss = ops.softmax_cross_entropy(model(x), y)
        loss.backward()
        opt.step()
        torch.cuda.synchronize()
    finally:
        hip._conv_wgrad = real
    assert len(seen) >= 2 and all(seen), seen     # the two stage-1 blocks of ResNet-18


def test_segmented_norms_are_deterministic_and_exact():
    """LARS pass 1 without float atomics: segments inside one 8192-element chunk, segments crossing one border,
    segments covering many chunks, an empty segment and zero padding between segments - against fp64, and bit for
    bit the same on every call."""
    from distributeddeeplearning_amd.ops import kernels as K

    K.require()
    g_ = torch.Generator(device=DEV).manual_seed(17)
    lens = [64, 64, 8000, 100, 9000, 0, 300000, 256, 8192 * 3, 5, 70000]
    starts, ends, pos = [], [], 0
    for n in lens:
        starts.append(pos)
        ends.append(pos + n)
        pos += (n + 63) // 64 * 64 + 64          # padded, with a gap
    total = pos
    w = torch.zeros(total, device=DEV)
    g = torch.zeros(total, device=DEV)
    for a, b in zip(starts, ends):
        w[a:b] = torch.randn(b - a, generator=g_, device=DEV)
        g[a:b] = torch.randn(b - a, generator=g_, device=DEV) * 0.1
    so = torch.tensor(starts, dtype=torch.int64, device=DEV)
    se = torch.tensor(ends, dtype=torch.int64, device=DEV)
    for gg in (g, g.to(BF)):
        wn, gn = K.segmented_norms(w, gg, so, se)
        ref_w = torch.stack([w[a:b].double().pow(2).sum().sqrt() for a, b in zip(starts, ends)])
        ref_g = torch.stack([gg[a:b].double().pow(2).sum().sqrt() for a, b in zip(starts, ends)])
        assert torch.allclose(wn.double(), ref_w, rtol=1e-5, atol=1e-6)
        assert torch.allclose(gn.double(), ref_g, rtol=1e-5, atol=1e-6)
        for _ in range(3):
            wn2, gn2 = K.segmented_norms(w, gg, so, se)
            assert torch.equal(wn2, wn) and torch.equal(gn2, gn)
    r1 = K.lars_trust_ratios(w, g, so, se, 0.001, 5e-5, 1e-9).clone()
    r2 = K.lars_trust_ratios(w, g, so, se, 0.001, 5e-5, 1e-9)
    assert torch.equal(r1, r2) and float(r1[5]) == 1.0       # (empty segment: ratio 1)


@pytest.mark.parametrize("shape", [(11, 56, 256, 64), (44, 28, 512, 128)])
def test_lin_fused_kernel_against_the_two_kernels_it_replaces(shape):
    """lin_fused.hip: the masked input gradient of a linear unit, its BatchNorm-backward sums and the raw
    weight-gradient product g^T z2 in one pass over g - against the two-segment tile kernel (bit-identical g2: same
    reduction order) and the split-K weight-gradient kernel (equal to fp32 rounding), and against fp32."""
    from distributeddeeplearning_amd.ops import hip_ops as hip
    from distributeddeeplearning_amd.ops import kernels as K

    K.require()
    B, H, C, C2 = shape             # (stage 1: one workgroup per pixel range; stage 2: two, one per half of the narrow channels)
    W = H
    M = B * H * W
    nwg = K.lin_fused_wgs(M, C, C2)
    if nwg == 0:
        pytest.skip("fused kernel switched off")
    g = rand_nhwc(B, C, H, W, seed=91)
    z2 = torch.relu(rand_nhwc(B, C2, H, W, seed=92))
    y_prev = rand_nhwc(B, C2, H, W, seed=93)
    gen = torch.Generator(device=DEV).manual_seed(94)
    b1 = (torch.randn((C2, C), generator=gen, device=DEV) / 16).to(BF)
    q = (torch.randn((C2, C2), generator=gen, device=DEV) / 8).to(BF)
    bias = torch.randn(C2, generator=gen, device=DEV) * 0.1
    fco = torch.stack([torch.rand(C2, generator=gen, device=DEV) + 0.5, torch.randn(C2, generator=gen, device=DEV) * 0.3]).contiguous()
    # the two kernels
    g2_ref = torch.empty_like(z2)
    T = K.conv_tiles_m_bwd(M, H, W, 1, 1)
    st_ref = torch.empty((T, 2, C2), device=DEV)
    K.conv_igemm(g, hip._geom(g), b1, g2_ref, C2, 1, 1, 1, 0, 0, 1, H, W, stats=st_ref, ep_y=y_prev, ep_fcoef=fco,
                 x2=z2, w2=q, c2=C2, bias=bias)
    splits = K.conv_wgrad_splits(M, C, C2, 1, 1)
    p_ref = torch.empty((splits, C * C2), device=DEV)
    K.conv_wgrad(g, z2, hip._geom(z2), p_ref, C, 1, 1, 1, 0, 0, H, W, splits)
    # the fused kernel
    g2 = torch.full_like(z2, float("nan"))
    st = torch.full((nwg, 2, C2), float("nan"), device=DEV)
    P = torch.full((nwg, C * C2), float("nan"), device=DEV)
    K.lin_fused(g, z2, b1, q, bias, y_prev, fco, g2, st, P, M, C, C2, nwg)
    assert torch.equal(g2, g2_ref)
    assert torch.allclose(st.sum(0), st_ref.sum(0), rtol=2e-4, atol=1e-2)
    assert torch.allclose(P.sum(0), p_ref.sum(0), rtol=1e-4, atol=1e-2)
    # fp32 references
    gm = g.permute(0, 2, 3, 1).reshape(M, C).float()
    zm = z2.permute(0, 2, 3, 1).reshape(M, C2).float()
    ym = y_prev.permute(0, 2, 3, 1).reshape(M, C2).float()
    full = gm @ b1.float().t() + zm @ q.float().t() + bias
    on = (ym * fco[0] + fco[1]) > 0
    ref = torch.where(on, full, torch.zeros_like(full))
    got = g2.permute(0, 2, 3, 1).reshape(M, C2).float()
    flips = (on != (got != 0)) & (full.abs() > 1e-3)          # (elements at the ReLU threshold may flip with the fma rounding)
    assert float(flips.float().mean()) < 1e-4
    assert rms_relerr(torch.where(flips, ref, got), ref) < 4e-3
    assert rms_relerr(P.sum(0).view(C, C2), gm.t() @ zm) < 1e-4
    # without the mask (projection shortcut's unit): the plain two-segment GEMM
    g2p_ref = torch.empty_like(z2)
    K.conv_igemm(g, hip._geom(g), b1, g2p_ref, C2, 1, 1, 1, 0, 0, 1, H, W, x2=z2, w2=q, c2=C2, bias=bias)
    g2p = torch.full_like(z2, float("nan"))
    Pp = torch.full((nwg, C * C2), float("nan"), device=DEV)
    K.lin_fused(g, z2, b1, q, bias, None, None, g2p, None, Pp, M, C, C2, nwg)
    assert torch.equal(g2p, g2p_ref) and torch.equal(Pp, P)
    # the Gram matrix and the column sums of z2 from the same tiles (where the shape has the registers for it)
    if K.lin_fused_gram_ok(C, C2):
        G = torch.full((nwg, C2 * C2), float("nan"), device=DEV)
        ZS = torch.full((nwg, 2, C2), float("nan"), device=DEV)
        g2g = torch.empty_like(z2)
        Pg = torch.empty_like(P)
        stg = torch.empty_like(st)
        K.lin_fused(g, z2, b1, q, bias, y_prev, fco, g2g, stg, Pg, M, C, C2, nwg, gram=G, colsum=ZS)
        assert torch.equal(g2g, g2) and torch.equal(Pg, P) and torch.equal(stg, st)
        assert rms_relerr(G.sum(0).view(C2, C2), zm.t() @ zm) < 1e-4
        assert torch.allclose(ZS.sum(0)[0], zm.sum(0), rtol=1e-4, atol=1e-2) and torch.equal(ZS[:, 0], ZS[:, 1])
    # deterministic
    P2 = torch.empty_like(P)
    st2 = torch.empty_like(st)
    K.lin_fused(g, z2, b1, q, bias, y_prev, fco, torch.empty_like(z2), st2, P2, M, C, C2, nwg)
    assert torch.equal(P2, P) and torch.equal(st2, st)


@pytest.mark.parametrize("stage", [1, 2])
def test_linear_units_with_the_fused_kernel_match_the_two_kernel_form(stage):
    """Three bottlenecks of stage 1 (56 x 56) or stage 2 (28 x 28) with the fused input-gradient / raw-product kernel on and off:
    every gradient agrees to the accuracy of the different summation orders (the slabs are cut differently)."""
    from distributeddeeplearning_amd.models.resnet import Bottleneck, _block_spec
    from distributeddeeplearning_amd.ops import functional as ops

    hip = _hip()
    from distributeddeeplearning_amd.ops import kernels as K
    B = 11 if stage == 1 else 44
    if (K.lin_fused_wgs(B * 56 * 56, 256, 64) if stage == 1 else K.lin_fused_wgs(B * 28 * 28, 512, 128)) == 0:
        pytest.skip("fused kernel switched off")

    def make():
        torch.manual_seed(19)
        if stage == 1:
            ms = [Bottleneck(64, 64, 1, True, False), Bottleneck(256, 64, 1, False, False), Bottleneck(256, 64, 1, False, False)]
        else:
            ms = [Bottleneck(256, 128, 2, True, False), Bottleneck(512, 128, 1, False, False), Bottleneck(512, 128, 1, False, False)]
        for m in ms:
            m.to(DEV).train()
            for mod in m.modules():
                if hasattr(mod, "kernel"):
                    mod.weight.data = mod.weight.data.to(BF).contiguous(memory_format=CL)
        return ms

    x0 = torch.relu(rand_nhwc(B, 64 if stage == 1 else 256, 56, 56, seed=95))
    gout = rand_nhwc(B, 256, 56, 56, seed=96) if stage == 1 else rand_nhwc(B, 512, 28, 28, seed=96)
    used = {"n": 0}
    real = K.lin_fused

    def counting(*a, **k):
        used["n"] += 1
        return real(*a, **k)

    def run(on, gram=False):
        hip.reset_caches()
        ms = make()
        x = x0.detach().clone().requires_grad_(True)
        saved = hip._LIN_FUSED["on"], hip._LIN_FUSED["gram"]
        hip._LIN_FUSED["on"], hip._LIN_FUSED["gram"] = on, gram
        K.lin_fused = counting
        try:
            out = ops.residual_chain(x, [_block_spec(m.pairs(), m.downsample) for m in ms], True, 0.1, 1e-5)
            out.backward(gout)
            torch.cuda.synchronize()
        finally:
            hip._LIN_FUSED["on"], hip._LIN_FUSED["gram"] = saved
            K.lin_fused = real
        names = [n for m in ms for n, _ in m.named_parameters()]
        return x.grad, names, [p.grad.float().clone() for m in ms for p in m.parameters()]

    dx_a, names, g_a = run(True)
    want = 3 if stage == 1 else 2          # the two blocks whose incoming gradient is pre-masked (+ stage 1: the projection's unit)
    assert used["n"] == want, used
    dx_b, _, g_b = run(False)
    assert used["n"] == want
    assert rms_relerr(dx_a, dx_b) < 2e-3
    for n, a, b in zip(names, g_a, g_b):
        assert rms_relerr(a, b) < 5e-3, (n, rms_relerr(a, b))
    if stage == 1:      # ... and with the Gram matrix / column sums from the same kernel (DDL_LIN_FUSED_GRAM=1)
        dx_c, _, g_c = run(True, gram=True)
        assert torch.equal(dx_c, dx_a)
        for n, a, c in zip(names, g_a, g_c):
            assert rms_relerr(c, a) < 5e-3, (n, rms_relerr(c, a))


_WGRAD_BIG_SNIPPET = r"""
import torch
from distributeddeeplearning_amd.ops import hip_ops as hip
def wide_weight_gradients():
    out = []
    for (B, C, K, hw, r) in ((64, 1024, 256, 14, 1), (64, 256, 256, 14, 3), (32, 512, 2048, 7, 1)):
        g = torch.Generator().manual_seed(100 + C + K + r)
        x = torch.randn(B, C, hw, hw, generator=g).to("cuda", torch.bfloat16).contiguous(memory_format=torch.channels_last)
        dy = torch.randn(B, K, hw, hw, generator=g).to("cuda", torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = torch.zeros(K, C, r, r, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
        out.append((hip._conv_wgrad(dy, x, w, 1, r // 2).float().cpu(), x.float().cpu(), dy.float().cpu(), r))
    torch.cuda.synchronize()
    return out
"""


def test_both_loop_forms_of_the_wide_weight_gradient_kernel_give_the_same_bits(tmp_path):
    """``conv_wgrad256_kernel`` with the next tile's fragments read under the current tile's MFMAs
    (``DDL_WGRAD_BIG_PIPE=1``, the default) and its plain loop (``=0``, in a child process: the switch is read once)
    accumulate in the same order: equal bits; and both match an fp32 convolution gradient."""
    import os
    import subprocess
    import sys

    ns = {}
    exec(_WGRAD_BIG_SNIPPET, ns)
    here = ns["wide_weight_gradients"]()
    for dw, x, dy, r in here:
        ref = torch.nn.grad.conv2d_weight(x, dw.shape, dy, stride=1, padding=r // 2)
        err = (dw - ref).abs().max().item() / ref.abs().max().item()
        assert err < 1e-2, err
    path = str(tmp_path / "plain.pt")
    code = _WGRAD_BIG_SNIPPET + "\ntorch.save([t[0] for t in wide_weight_gradients()], %r)\nprint('OK')\n" % path
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DDL_WGRAD_BIG_PIPE="0", PYTHONPATH=root)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300, cwd=root)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stderr[-3000:]
    plain = torch.load(path, weights_only=True)
    for (dw, _, _, _), other in zip(here, plain):
        assert torch.equal(dw, other)


@pytest.mark.parametrize("nbytes,value", [(16, 0), (4096 * 16 + 16, 0), (51_200_000, 0), (1 << 20, 0x5A), (1000, 0), (48 + 8, 7)])
def test_arena_fill_kernel_and_its_runtime_fallback(nbytes, value):
    """K16 (``ddl_memset_async``): 16-byte aligned spans are filled by the in-tree kernel, anything else by the
    runtime's memset; neither touches a byte outside the span."""
    from distributeddeeplearning_amd.ops import kernels as K

    guard = 64
    buf = torch.full((nbytes + 2 * guard,), 0xEE, dtype=torch.uint8, device=DEV)
    K.memset_async(buf[guard:guard + nbytes], value)
    torch.cuda.synchronize()
    assert bool((buf[guard:guard + nbytes] == value).all())
    assert bool((buf[:guard] == 0xEE).all()) and bool((buf[guard + nbytes:] == 0xEE).all())

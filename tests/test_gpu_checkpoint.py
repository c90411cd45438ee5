"""Checkpoint / resume on the path that is benchmarked: bf16 compute model,
fp32 master weights, fused optimizer, HIP kernels.  Reference flow: rank-0 checkpoints + resume
broadcast of the Keras flavour (``HorovodKeras/src/imagenet_keras_horovod.py:287-291,310-324``)."""
import io
import json
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _make(seed, dev, classes=32):
    from distributeddeeplearning_amd.models import resnet18
    from distributeddeeplearning_amd.optim import FusedSGD

    torch.manual_seed(seed)
    m = resnet18(num_classes=classes).to_compute(dev, torch.bfloat16).train()
    o = FusedSGD(m.named_parameters(), lr=0.05, momentum=0.9, weight_decay=5e-5)
    return m, o


def _step(m, o, x, y):
    from distributeddeeplearning_amd.ops import functional as ops

    o.zero_grad()
    loss = ops.softmax_cross_entropy(m(x), y)
    loss.backward()
    o.step()
    return loss.detach().float().clone()


def _snapshot(m, o):
    torch.cuda.synchronize()
    snap = {"master." + str(k): v.clone() for k, v in o.master.items()}
    snap.update({"momentum." + str(k): v.clone() for k, v in o.momentum_buf.items()})
    snap.update({"param." + str(k): v.clone() for k, v in o.arena.params.items()})
    snap.update({"buf." + k: v.clone() for k, v in m.named_buffers()})
    return snap


def test_resume_continues_bit_exactly_and_exports_a_stock_fp32_state_dict(tmp_path):
    from distributeddeeplearning_amd.models import resnet18
    from distributeddeeplearning_amd.ops import functional as ops
    from distributeddeeplearning_amd.utils import checkpoint

    assert ops.backend() != "aten"
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(11)
    xs = [torch.rand(8, 3, 64, 64, generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
          for _ in range(4)]
    ys = [torch.randint(0, 32, (8,), generator=g).to(dev) for _ in range(4)]

    m, o = _make(3, dev)
    for i in range(3):
        _step(m, o, xs[i], ys[i])
    torch.manual_seed(1234)                       # the generator state a resumed run must pick up
    path = checkpoint.save(str(tmp_path), 3, m, o)
    expect_draw = torch.rand(4)
    loss_a = _step(m, o, xs[3], ys[3])
    want = _snapshot(m, o)
    del m, o

    # (i) a FRESH model and optimizer (other seed: every tensor differs before the load)
    m2, o2 = _make(99, dev)
    _step(m2, o2, xs[0], ys[0])                   # its derived weight copies exist and are stale after the load
    torch.manual_seed(1)
    payload = checkpoint.load(path, m2, o2)
    assert payload["epoch"] == 3
    assert torch.equal(torch.rand(4), expect_draw)   # (iii) rng restored
    loss_b = _step(m2, o2, xs[3], ys[3])
    got = _snapshot(m2, o2)
    assert torch.equal(loss_a, loss_b), (float(loss_a), float(loss_b))
    assert set(got) == set(want)
    for k in want:
        assert torch.equal(got[k], want[k]), k

    # (ii) the saved model dict IS a torchvision-style fp32 state_dict
    sd = torch.load(path, weights_only=True)["model"]
    assert sd["conv1.weight"].shape == (64, 3, 7, 7) and sd["conv1.weight"].dtype == torch.float32
    assert sd["layer1.0.conv1.weight"].shape == (64, 64, 3, 3) and sd["fc.weight"].shape == (32, 512)
    assert sd["bn1.num_batches_tracked"].dtype == torch.int64 and int(sd["bn1.num_batches_tracked"]) == 3
    assert "layer2.0.downsample.0.weight" in sd and "layer2.0.downsample.1.running_var" in sd
    plain = resnet18(num_classes=32)              # the repo's plain fp32 CPU ResNet (stock torch ops)
    missing, unexpected = plain.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    plain.eval()
    m3, o3 = _make(5, dev)
    checkpoint.load(path, m3, o3)
    m3.eval()
    with torch.no_grad():
        ref = plain(xs[0].float().cpu().contiguous())
        out = m3(xs[0]).float().cpu()
    err = float((out - ref).abs().max() / ref.abs().max())
    assert err < 4e-2, err                        # bf16 activations through 18 layers vs fp32


def test_weights_only_checkpoint_seeds_the_masters(tmp_path):
    """A file without optimizer state (fine-tuning): the masters still come from the fp32 entries
    (checkpoint.load no longer couples that to an unrelated method)."""
    from distributeddeeplearning_amd.utils import checkpoint

    dev = torch.device("cuda", 0)
    m, o = _make(3, dev)
    path = checkpoint.save(str(tmp_path), 1, m, None)
    m2, o2 = _make(8, dev)
    checkpoint.load(path, m2, o2)
    torch.cuda.synchronize()
    for dt in o.master:
        assert torch.equal(o2.master[dt], o.arena.params[dt].float())      # saved without masters: bf16 -> fp32
        assert torch.equal(o2.arena.params[dt], o.arena.params[dt])


def test_two_rank_resume_through_the_native_engine(tmp_path):
    """2 processes on the one GPU: rank 0 loads the file, ``broadcast_parameters`` +
    ``broadcast_optimizer_state`` + master sync go through the native engine; afterwards both ranks
    hold rank 0's saved state bit for bit and stay identical through a further step."""
    sys.path.insert(0, HERE)
    from test_gpu_engine import _launch

    rc, res, out = _launch("resume", exports={"TEST_CKPT_DIR": str(tmp_path)})
    assert rc == 0, out[-3000:]
    assert set(res) == {0, 1}
    assert res[0]["epoch"] == res[1]["epoch"] == 2
    assert res[0]["after_resume"] == res[1]["after_resume"] == res[0]["saved"]
    assert res[0]["after_step"] == res[1]["after_step"]
    assert res[0]["files"] == ["checkpoint-2.pt"]
